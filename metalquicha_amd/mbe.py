"""The caller of the hot path: many-body-expansion fragment lists and energy assembly.

Host-side mirror (numpy/ints only) of the pieces of src/fragmentation the engine is driven by,
kept so that bench.py and the tests exercise the engine exactly the way metalquicha does:

  generate_mbe_term_list   src/mqc_driver.f90:511 -> combinations + distance screening
                           (src/fragmentation/common/mqc_frag_utils.f90:180-310) + largest-first
                           ordering (sort_fragments_by_size, mqc_frag_utils.f90:312)
  build_fragment           build_fragment_from_indices (no H-caps: monomers here are whole molecules)
  compute_mbe              bottom-up deltas, src/fragmentation/mbe/mqc_mbe.f90:37-118,503
  compute_mbe_coefficients src/fragmentation/mbe/mqc_mbe.f90:120-198
  partition_terms          static round-robin over the cost-sorted list, the FMO code's choice
                           (backends/libcint/mqc_libcint_fmo.f90:1806-1819); SURVEY.md 8e
  run_mbe                  do_fragment_work over the owned terms (one engine batch), energies
                           returned in a zero-padded vector ready for ONE all-reduce

Nothing here touches the GPU; the SCFs are run by methods.run_hip_scf_batch.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .basis import ANGSTROM_TO_BOHR, SYMBOL_TO_Z
from .methods import PhysicalFragment, ScfSettings, run_hip_scf_batch

BOHR_TO_ANGSTROM = 1.0 / ANGSTROM_TO_BOHR


@dataclass
class FragmentedSystem:
    element_numbers: np.ndarray                 # (n_atoms,)
    coordinates: np.ndarray                     # (3, n_atoms) Bohr
    monomers: List[np.ndarray]                  # atom indices of each monomer
    charges: Optional[List[int]] = None
    multiplicities: Optional[List[int]] = None

    @property
    def n_monomers(self) -> int:
        return len(self.monomers)


def read_xyz(path: str):
    with open(path) as f:
        lines = f.read().splitlines()
    n = int(lines[0].split()[0])
    sym, xyz = [], []
    for ln in lines[2:2 + n]:
        p = ln.split()
        sym.append(p[0]); xyz.append([float(p[1]), float(p[2]), float(p[3])])
    return sym, np.array(xyz)


def system_from_xyz(symbols: Sequence[str], xyz_angstrom: np.ndarray, monomers: Sequence[Sequence[int]]) -> FragmentedSystem:
    z = np.array([SYMBOL_TO_Z[s.lower()] for s in symbols], dtype=np.int32)
    return FragmentedSystem(z, (np.asarray(xyz_angstrom, dtype=float) * ANGSTROM_TO_BOHR).T.copy(),
                            [np.asarray(m, dtype=np.int64) for m in monomers])


def min_intermonomer_distance(system: FragmentedSystem, a: int, b: int) -> float:
    """Minimum interatomic distance between two monomers, in Angstrom."""
    xa = system.coordinates[:, system.monomers[a]].T
    xb = system.coordinates[:, system.monomers[b]].T
    d = np.linalg.norm(xa[:, None, :] - xb[None, :, :], axis=2)
    return float(d.min()) * BOHR_TO_ANGSTROM


def generate_mbe_term_list(system: FragmentedSystem, level: int, cutoffs: Optional[Dict[int, float]] = None) -> List[Tuple[int, ...]]:
    """All k-mers up to `level`; a term is dropped if it, or any of its k-subsets, exceeds the k-mer
    cutoff (minimum inter-monomer atomic distance, Angstrom).  Largest terms first."""
    n = system.n_monomers
    cutoffs = cutoffs or {}
    dist_cache: Dict[Tuple[int, int], float] = {}

    def dist(a, b):
        key = (a, b) if a < b else (b, a)
        if key not in dist_cache:
            dist_cache[key] = min_intermonomer_distance(system, *key)
        return dist_cache[key]

    def spread(term):
        return max(dist(a, b) for a, b in itertools.combinations(term, 2))

    terms: List[Tuple[int, ...]] = []
    for k in range(1, level + 1):
        for term in itertools.combinations(range(n), k):
            ok = True
            if k >= 2:
                for kk in range(2, k + 1):
                    cut = cutoffs.get(kk)
                    if cut is None:
                        continue
                    if any(spread(sub) > cut for sub in itertools.combinations(term, kk)):
                        ok = False
                        break
            if ok:
                terms.append(term)
    terms.sort(key=lambda t: -sum(len(system.monomers[m]) for m in t))     # stable: largest first
    return terms


def build_fragment(system: FragmentedSystem, term: Sequence[int]) -> PhysicalFragment:
    atoms = np.concatenate([system.monomers[m] for m in term])
    charge = 0 if system.charges is None else int(sum(system.charges[m] for m in term))
    return PhysicalFragment(system.element_numbers[atoms], system.coordinates[:, atoms].copy(), charge=charge)


_MBE_PLAN_CACHE: Dict[Tuple[int, int], list] = {}


def _mbe_plan(terms: Sequence[Tuple[int, ...]]):
    """Per order k (ascending): indices of the k-body terms and, per term, the indices of all its proper
    sub-terms.  Depends on the term list only, so it is cached (an MBE driver assembles the same list again
    and again)."""
    key = (len(terms), hash(tuple(map(tuple, terms))))
    plan = _MBE_PLAN_CACHE.get(key)
    if plan is not None:
        return plan
    lookup = {tuple(t): i for i, t in enumerate(terms)}
    by_order: Dict[int, List[int]] = {}
    for i, t in enumerate(terms):
        by_order.setdefault(len(t), []).append(i)
    plan = []
    for k in sorted(by_order):
        idx = np.asarray(by_order[k])
        subs = []
        for i in idx:
            t = terms[i]
            row = []
            for m in range(1, k):
                for sub in itertools.combinations(t, m):
                    j = lookup.get(sub)
                    if j is None:
                        raise KeyError("Subset not found in bottom-up MBE: %s of %s" % (sub, t))
                    row.append(j)
            subs.append(row)
        plan.append((k, idx, np.asarray(subs, dtype=np.int64).reshape(len(idx), -1)))
    if len(_MBE_PLAN_CACHE) >= 8:
        _MBE_PLAN_CACHE.clear()
    _MBE_PLAN_CACHE[key] = plan
    return plan


def compute_mbe(terms: Sequence[Tuple[int, ...]], energies: Sequence[float]):
    """Bottom-up many-body deltas; returns (total, per-order sums, delta per term)."""
    e = np.asarray(energies, dtype=np.float64)
    delta = np.zeros(len(terms))
    by_order: Dict[int, float] = {}
    for k, idx, subs in _mbe_plan(terms):
        d = e[idx] - (delta[subs].sum(axis=1) if subs.shape[1] else 0.0)
        delta[idx] = d
        by_order[k] = float(np.sum(d))
    return float(np.sum(delta)), by_order, delta


def compute_mbe_coefficients(terms: Sequence[Tuple[int, ...]]) -> np.ndarray:
    """E_total = sum_i c_i E_i.  For unscreened MBE-2 of N monomers: c_dimer = 1, c_monomer = 2 - N."""
    lookup = {tuple(t): i for i, t in enumerate(terms)}
    coef = np.zeros(len(terms))
    for t in terms:
        k = len(t)
        # E_total = sum over terms of delta(t); delta(t) = sum_{s subseteq t} (-1)^{|t|-|s|} E(s)
        for kk in range(1, k + 1):
            for sub in itertools.combinations(t, kk):
                coef[lookup[sub]] += (-1.0) ** (k - kk)
    return coef


def partition_terms(n_terms: int, rank: int, world: int) -> np.ndarray:
    """Static round-robin over the cost-sorted list: rank r owns terms r, r+world, ..."""
    return np.arange(rank, n_terms, world, dtype=np.int64)


# ---- cost-aware distribution ----------------------------------------------------------------------------------------
# The reference sorts fragments largest-first (sort_fragments_by_size, src/fragmentation/common/mqc_frag_utils.f90:312)
# and lets workers PULL them from a queue (queue_t, src/fragmentation/common/mqc_work_queue.f90:9-58; coordinator
# ...distribution_scheme.F90:402-758), so a rank that drew heavy fragments simply draws fewer.  Round-robin over the
# sorted list is only balanced when all terms of an order cost the same; with mixed lists (monomers next to dimers,
# two basis sets, GMBE intersections) it is not.  Two replacements, both ending in the same single all-reduce:
#   partition_terms_lpt   static, no communication: longest-processing-time-first over a cost model -- every rank
#                         computes the same assignment from the same list (makespan <= (4/3 - 1/(3 m)) optimum, and
#                         max load - mean load <= the largest single cost);
#   PullQueue             dynamic: an atomic counter on the job's rendezvous store (torch.distributed TCPStore.add);
#                         ranks draw chunks of the cost-sorted list until it is dry -- the reference's queue without a
#                         coordinator rank.  Each draw is one engine batch.
def term_costs(system: "FragmentedSystem", terms: Sequence[Tuple[int, ...]], basis_set: str, functional: str = "",
               density_fitting: bool = False) -> np.ndarray:
    """Relative cost of every term from its basis-function count n: the in-core exact-ERI path forms ~n^4 / 8 integrals
    and streams them every iteration (n^4), density fitting costs ~n^3, the quadrature adds (grid points ~ atoms) x n^2.
    A model for BALANCING -- only ratios between terms matter."""
    from .basis import build_flat_basis
    nao_of_z: Dict[int, int] = {}

    def nao(z: int) -> int:
        if z not in nao_of_z:
            fb = build_flat_basis(basis_set, [z])
            nao_of_z[z] = int(fb.nao)
        return nao_of_z[z]
    z_all = np.asarray(system.element_numbers)
    out = np.zeros(len(terms))
    for i, t in enumerate(terms):
        atoms = [int(a) for m in t for a in system.monomers[m]]
        n = float(sum(nao(int(z_all[a])) for a in atoms))
        c = n ** 3 if density_fitting else n ** 4 / 64.0
        if functional:
            c += 2.0e3 * len(atoms) * n * n        # ~17 k grid points per atom, ~8 n^2 flop each, against ~64 n^4 flop of J/K work per fragment
        out[i] = c
    return out


def partition_terms_lpt(costs: Sequence[float], rank: int, world: int) -> np.ndarray:
    """Longest-processing-time-first: terms by decreasing cost, each to the currently least loaded rank (ties: lowest
    rank).  Deterministic, so every rank derives the same assignment without talking; returns this rank's terms in
    decreasing cost."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    mine: List[int] = []
    for i in order:
        r = int(np.argmin(load))
        load[r] += costs[i]
        if r == rank:
            mine.append(int(i))
    return np.asarray(mine, dtype=np.int64)


def partition_loads(costs: Sequence[float], world: int, scheme: str = "lpt") -> np.ndarray:
    """Per-rank summed cost under a scheme ("lpt" | "round_robin"): what a balance check looks at."""
    costs = np.asarray(costs, dtype=np.float64)
    part = partition_terms_lpt if scheme == "lpt" else (lambda c, r, w: partition_terms(len(c), r, w))
    return np.array([float(np.sum(costs[part(costs, r, world)])) for r in range(world)])


class PullQueue:
    """Work queue over a shared atomic counter: `draw()` returns the next slice [lo, hi) of a list of n items, or None
    when it is dry.  `counter_add(k)` must atomically add k to the shared counter and return the NEW value
    (torch.distributed.TCPStore: lambda k: store.add(key, k)).  Guided chunks: a draw takes remaining / (2 world)
    items, at least `min_chunk` -- large batches while the list is long (the engine wants batches), small ones at the
    tail (balance).  Since the counter only says how many items are gone, every rank can replay the chunk boundaries."""

    def __init__(self, n_items: int, world: int, counter_add: Callable[[int], int], min_chunk: int = 1):
        self.n, self.world, self.add, self.min_chunk = int(n_items), max(1, int(world)), counter_add, max(1, int(min_chunk))

    def chunk_after(self, taken: int) -> int:
        left = self.n - taken
        return max(self.min_chunk, -(-left // (2 * self.world))) if left > 0 else 0

    def draw(self) -> Optional[Tuple[int, int]]:
        """Claims the next slice.  The counter is monotonic and every claim is one atomic add, so the slices of all ranks
        are disjoint and cover [0, n) whatever chunk sizes the ranks ask for; the size asked for follows the guided
        schedule at the position this rank last saw."""
        k = self.chunk_after(getattr(self, "_seen", 0))
        if k == 0:
            return None
        hi = int(self.add(k))
        lo = hi - k
        self._seen = hi
        if lo >= self.n:
            return None
        return lo, min(hi, self.n)


def run_mbe_pull(system: FragmentedSystem, settings: ScfSettings, terms: List[Tuple[int, ...]], queue: PullQueue,
                 costs: Optional[Sequence[float]] = None, evaluate: Optional[Callable] = None) -> MbeRun:
    """One rank's part of an MBE evaluation with a pull queue: draws slices of the cost-sorted term list, each slice ONE
    engine batch, until the queue is dry.  Energies / iterations are zero-padded over all terms for the one all-reduce
    that follows (as run_mbe)."""
    from .methods import run_hip_scf_groups
    evaluate = evaluate or (lambda groups: run_hip_scf_groups(settings, groups))
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable") if costs is not None else np.arange(len(terms))
    energies = np.zeros(len(terms)); iters = np.zeros(len(terms), dtype=np.int64)
    errors: List[str] = []
    owned: List[int] = []
    while True:
        sl = queue.draw()
        if sl is None:
            break
        idx = order[sl[0]:sl[1]]
        groups, positions = build_fragment_groups(system, [terms[i] for i in idx])
        recs = evaluate(groups)
        for pos, rec in zip(positions, recs):
            tix = idx[pos]
            ok = rec["has_error"] == 0
            energies[tix[ok]] = rec["e_total"][ok]
            iters[tix[ok]] = rec["iterations"][ok]
            for k in np.nonzero(~ok)[0]:
                errors.append("term %s: %s" % (terms[tix[k]], bytes(rec["message"][k]).split(b"\0", 1)[0].decode(errors="replace")))
        owned.extend(int(i) for i in idx)
    return MbeRun(terms, energies, iters, np.asarray(owned, dtype=np.int64), errors, None)


@dataclass
class MbeRun:
    terms: List[Tuple[int, ...]]
    energies: np.ndarray          # zero for terms this rank does not own
    iterations: np.ndarray
    owned: np.ndarray
    errors: List[str]
    gradient: Optional[np.ndarray] = None      # (n_atoms, 3) weighted sum of the owned fragments' gradients (zero-padded share)


def _group_layout(system: FragmentedSystem, term_list: Sequence[Tuple[int, ...]]):
    """Geometry-independent part of build_fragment_groups: per group (element sequence, atom index array
    (m, n_atoms), charges, positions in term_list).  Cached on the system per term list: a driver that evaluates
    the same expansion again (new coordinates or not) only regathers coordinates."""
    key = (len(term_list), hash(tuple(map(tuple, term_list))))
    cache = system.__dict__.setdefault("_layout_cache", {})
    if key in cache:
        return cache[key]
    by_shape: Dict[Tuple[int, ...], List[int]] = {}
    for pos, t in enumerate(term_list):
        by_shape.setdefault(tuple(len(system.monomers[m]) for m in t), []).append(pos)
    layout = []
    for shape, plist in by_shape.items():
        plist = np.asarray(plist)
        if len(set(shape)) == 1 and all(len(m) == shape[0] for m in system.monomers):
            mono = np.asarray(system.monomers)                                 # (n_monomers, atoms per monomer)
            atoms = mono[np.array([term_list[p] for p in plist])].reshape(len(plist), -1)
        else:
            atoms = np.array([np.concatenate([system.monomers[m] for m in term_list[p]]) for p in plist])   # (nt, na)
        zs = system.element_numbers[atoms]
        uniq, inverse = np.unique(zs, axis=0, return_inverse=True)
        inverse = np.asarray(inverse).reshape(-1)
        if system.charges is None:
            charges = np.zeros(len(plist), dtype=np.int32)
        else:
            ch = np.asarray(system.charges)
            charges = np.array([int(sum(ch[m] for m in term_list[p])) for p in plist], dtype=np.int32)
        for u in range(len(uniq)):
            sel = np.nonzero(inverse == u)[0]
            layout.append((uniq[u].astype(np.int32), np.ascontiguousarray(atoms[sel]), charges[sel], plist[sel]))
    if len(cache) >= 8:
        cache.clear()
    cache[key] = layout
    return layout


def build_fragment_groups(system: FragmentedSystem, term_list: Sequence[Tuple[int, ...]]):
    """The fragments of `term_list` as engine batches: terms of one order whose atoms carry the same element
    sequence form one FragmentGroup (coordinates gathered with one fancy index).  Returns (groups, positions)
    where positions[g][k] is the index into term_list of the k-th fragment of group g."""
    from .methods import FragmentGroup
    coords = np.ascontiguousarray(system.coordinates.T)           # (n_atoms, 3)
    groups, positions = [], []
    for z, atoms, charges, pos in _group_layout(system, term_list):
        groups.append(FragmentGroup(z, coords[atoms], charges))
        positions.append(pos)
    return groups, positions


def run_mbe(system: FragmentedSystem, settings: ScfSettings, level: int = 2,
            cutoffs: Optional[Dict[int, float]] = None, rank: int = 0, world: int = 1,
            terms: Optional[List[Tuple[int, ...]]] = None, want_gradient: bool = False,
            costs: Optional[Sequence[float]] = None) -> MbeRun:
    """`costs` (one per term, e.g. term_costs(...)): the static partition is longest-processing-time-first over them
    instead of round-robin -- for lists whose terms of one order do not cost the same."""
    from .methods import run_hip_scf_groups
    terms = terms if terms is not None else generate_mbe_term_list(system, level, cutoffs)
    owned = partition_terms(len(terms), rank, world) if costs is None else partition_terms_lpt(costs, rank, world)
    groups, positions = build_fragment_groups(system, [terms[i] for i in owned])
    owned_arr = np.asarray(owned)
    energies = np.zeros(len(terms))
    iters = np.zeros(len(terms), dtype=np.int64)
    errors = []
    grads_out: list = []
    gradient_total = None
    recs = run_hip_scf_groups(settings, groups, want_gradient=want_gradient, gradients_out=grads_out)
    if want_gradient:
        # weighted sum of the fragment gradients on the system's atoms (mqc_mbe.f90: the gradient twin of
        # E = sum_i c_i E_i; the reference all-reduces this 3 N vector next to the energies)
        coef = compute_mbe_coefficients(terms)
        total = np.zeros((len(system.element_numbers), 3))
        layout = _group_layout(system, [terms[i] for i in owned])
        for (z, atoms, charges, pos), rec, ga in zip(layout, recs, grads_out):
            tix = owned_arr[pos]
            for k in range(len(pos)):
                if rec["has_error"][k] or not rec["has_gradient"][k]:
                    continue
                np.add.at(total, atoms[k], coef[tix[k]] * ga[k])
        gradient_total = total
    for pos, rec in zip(positions, recs):
        tix = owned_arr[pos]
        ok = rec["has_error"] == 0
        energies[tix[ok]] = rec["e_total"][ok]
        iters[tix[ok]] = rec["iterations"][ok]
        for k in np.nonzero(~ok)[0]:
            msg = bytes(rec["message"][k]).split(b"\0", 1)[0].decode(errors="replace")
            errors.append("term %s: %s" % (terms[tix[k]], msg))
    return MbeRun(terms, energies, iters, owned, errors, gradient_total)


# ---------------------------------------------------------------------------------------------
W1_ANGSTROM = np.array([[0.0, 0.00000000009155, 0.10077199490609],
                        [0.0, 0.77250895271063, -0.46780199741728],
                        [0.0, -0.77250895280218, -0.46780199748881]])   # validation/inputs/sample_inputs/w1.xyz


def water_cluster(n_side: int = 4, spacing: float = 3.1, seed: int = 20260821) -> FragmentedSystem:
    """(H2O)_{n^3}: cubic lattice, `spacing` Angstrom, one rigid water (w1.xyz internal geometry) per
    site, uniformly random orientation from a seeded generator (the (H2O)64
    workload of BASELINE.json configs[2]; no such file exists in the reference)."""
    rng = np.random.default_rng(seed)
    mol = W1_ANGSTROM - W1_ANGSTROM.mean(axis=0)
    sym, xyz, monomers = [], [], []
    for ix in range(n_side):
        for iy in range(n_side):
            for iz in range(n_side):
                q = rng.normal(size=4); q /= np.linalg.norm(q)
                a, b, c, d = q
                R = np.array([[a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c)],
                              [2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b)],
                              [2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d]])
                pos = mol @ R.T + spacing * np.array([ix, iy, iz], dtype=float)
                base = len(sym)
                sym += ["O", "H", "H"]
                xyz.append(pos)
                monomers.append([base, base + 1, base + 2])
    return system_from_xyz(sym, np.vstack(xyz), monomers)


def write_xyz(path: str, system: FragmentedSystem, comment: str = ""):
    from .basis import SYMBOLS
    with open(path, "w") as f:
        f.write("%d\n%s\n" % (len(system.element_numbers), comment))
        for z, r in zip(system.element_numbers, system.coordinates.T * BOHR_TO_ANGSTROM):
            f.write("%-2s %18.12f %18.12f %18.12f\n" % (SYMBOLS[z], r[0], r[1], r[2]))
