"""Host-side mirror of the reference's FMO2 / EE-MBE caller (fragments embedded in the point charges of the others).

What `backends/libcint/mqc_libcint_fmo.f90` does around its per-fragment SCF, restated so that the embedded expansions
can be driven through the HIP engine the way `mbe.py` drives MBE -- the SCFs are the engine's, everything here is the
bookkeeping the reference's Fortran host code does:

* `run_fmo2` (:425-497): monomers iterated to self-consistency, then the pairs once;
* `calculate_monomers` (:1484-1564): a bare pass, then passes in the field of the previous pass's charges until the sum
  of monomer energies moves by less than `outer_tol`; every fragment of a pass is independent, so a pass is ONE batch
  call (`mqc_hip_scf_run_batch`), and across ranks a pass ends in one exchange (`exchange_monomers`, :1890-1948);
* `embedding_operator` (:1077-1160) with `esp = "ptc"`: every atom outside the fragment is a point charge
  q_A = Z_A - Mulliken population (`fragment_charges`, :2001-2021); the engine builds u = -sum_g q_g/|r - R_g| and adds it
  to H (ABI 3: `mqc_hip_molecule_t.point_charges`), returns tr(D u), u and the charges;
* `inner_scf` (:1992-1997): internal energy = E_scf - tr(D u);
* `nmer_term` (:1162-1274): a pair in the field of everything outside it, e_internal = E - tr(D u),
  e_resp = tr((D - D_I (+) D_J) u) for the "fmo" expansion; E_scf as it stands for the "mbe" (EE-MBE) expansion;
* `calculate_polymers` (:1566-1689) at level 2: dE_IJ = e_internal + e_resp - E_I - E_J, total = sum E_I + sum dE_IJ.

* `esp = "exact"` (the reference's default for FMO): fragments within `resppc` van der Waals sums (`near_fragments`,
  :1276-1316) give their bare nuclei as charges and their electrons through the exact Coulomb operator J[D_K] in the
  fragment's basis (`local_coulomb`, :1337-1406: the J build over the supersystem fragment + neighbour with only the
  neighbour's block of the density filled); here J comes from the engine's in-core integral and J kernels on the same
  supersystems, all (fragment, neighbour) pairs of a pass in one `mqc_hip_coulomb_batch` call per element sequence, and
  enters the SCF as `h_extra` (ABI 3); the far fragments stay Mulliken charges.

* `esp = "none"` is the plain many-body expansion through this driver; `far_field = "ignore"` drops the distant
  fragments from the field instead of approximating them (:1119-1124).

Scope: whole-molecule fragments (no severed bonds, caps or AFO projector), closed shells, Mulliken or ignored far field
-- CHELPG charges stay with the reference.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .mbe import FragmentedSystem
from .methods import FragmentGroup, ScfSettings


@dataclass
class EmbeddedJob:
    atoms: Tuple[int, ...]                   # atoms of the fragment / pair, in basis order
    field_atoms: Sequence[int] = ()          # outside atoms that act as point charges (empty = none)
    field_charges: Optional[np.ndarray] = None   # their weights: Mulliken charge (far) or bare nuclear charge (near)
    h_extra: Optional[np.ndarray] = None     # (n, n): the near fragments' exact Coulomb operator, or None


@dataclass
class EmbeddedResult:
    e_total: float = 0.0
    e_embedding: float = 0.0                 # tr(D u)
    iterations: int = 0
    density: Optional[np.ndarray] = None
    charges: Optional[np.ndarray] = None     # Mulliken, one per atom of the job
    u: Optional[np.ndarray] = None
    error: str = ""


Solver = Callable[[Sequence[EmbeddedJob]], List[EmbeddedResult]]
CoulombRequest = Tuple[Sequence[int], Sequence[int], np.ndarray]      # (atoms, neighbour's atoms, neighbour's density)
Coulomb = Callable[[Sequence[CoulombRequest]], List[np.ndarray]]      # -> J[D_K] in the basis of `atoms`, one per request

# Bondi's van der Waals radii with Rowland and Taylor's hydrogen, Angstrom, Z = 1..18 (src/core/mqc_elements.f90:58-60)
VDW_ANGSTROM = (1.10, 1.40, 1.81, 1.53, 1.92, 1.70, 1.55, 1.52, 1.47, 1.54, 2.27, 1.73, 1.84, 2.10, 1.80, 1.80, 1.75, 1.88)


def near_fragments(system: FragmentedSystem, group: Sequence[int], resppc: float) -> List[int]:
    """Fragments outside `group` treated exactly: closest atom pair within `resppc` sums of van der Waals radii; all of
    them when resppc < 0 (near_fragments / unitless_distance, mqc_libcint_fmo.f90:1276-1335)."""
    from .basis import ANGSTROM_TO_BOHR
    z = np.asarray(system.element_numbers); xyz = system.coordinates.T
    inside = [int(a) for g in group for a in system.monomers[g]]
    near = []
    for k in range(system.n_monomers):
        if k in group:
            continue
        if resppc < 0.0:
            near.append(k); continue
        other = [int(b) for b in system.monomers[k]]
        if max(int(z[a]) for a in inside + other) > len(VDW_ANGSTROM):
            raise ValueError("fmo: no van der Waals radius tabulated here for an element of fragment %d" % k)
        d = np.linalg.norm(xyz[inside][:, None, :] - xyz[other][None, :, :], axis=2)
        scale = (np.array([VDW_ANGSTROM[int(z[a]) - 1] for a in inside])[:, None]
                 + np.array([VDW_ANGSTROM[int(z[b]) - 1] for b in other])[None, :]) * ANGSTROM_TO_BOHR
        if float(np.min(d / scale)) <= resppc:
            near.append(k)
    return near


def fragment_separations(system: FragmentedSystem) -> np.ndarray:
    """(n_frag, n_frag) closest atom pair of every two fragments in sums of van der Waals radii -- near_fragments' measure
    for all pairs at once (the pair phase asks it for every n-mer)."""
    from .basis import ANGSTROM_TO_BOHR
    z = np.asarray(system.element_numbers); xyz = system.coordinates.T
    if int(np.max(z)) > len(VDW_ANGSTROM):
        raise ValueError("fmo: no van der Waals radius tabulated here for an element of the system")
    r = np.array([VDW_ANGSTROM[int(v) - 1] for v in z]) * ANGSTROM_TO_BOHR
    u = np.linalg.norm(xyz[:, None, :] - xyz[None, :, :], axis=2) / (r[:, None] + r[None, :])
    nf = system.n_monomers
    sep = np.zeros((nf, nf))
    idx = [np.asarray(m, dtype=np.int64) for m in system.monomers]
    for a in range(nf):
        rows = u[idx[a]]
        for b in range(a + 1, nf):
            sep[a, b] = sep[b, a] = float(np.min(rows[:, idx[b]]))
    return sep


def hip_cross_coulomb(system: FragmentedSystem, settings: ScfSettings) -> Coulomb:
    """J[D_K] of a neighbour's electrons in the basis of `atoms`, for all requests of a pass at once: the engine's
    in-core integral and J kernels on the supersystems atoms + neighbour with only the neighbour's block of the density
    filled, leading block of J (local_coulomb) -- ONE `mqc_hip_coulomb_batch` call per element sequence."""
    import ctypes as C
    from . import capi
    from .methods import _BAS_DTYPE, _MOL_DTYPE, _basis_record, _flat_basis_z
    z_all = np.asarray(system.element_numbers, dtype=np.int32)
    coords = np.ascontiguousarray(system.coordinates.T)

    def coulomb(requests: Sequence[CoulombRequest]) -> List[np.ndarray]:
        out: List[Optional[np.ndarray]] = [None] * len(requests)
        by_key: Dict[tuple, List[int]] = {}
        for r, (atoms, other, d_other) in enumerate(requests):
            by_key.setdefault((tuple(int(v) for v in z_all[list(atoms) + list(other)]), int(d_other.shape[0]), len(other)), []).append(r)
        lib = capi.load_library()
        ctx = capi.get_context(settings.device_rank)
        pieces = [(key, ks[lo:lo + 16384]) for key, ks in by_key.items() for lo in range(0, len(ks), 16384)]   # bounds the host arrays
        for (zseq, nk, n_src), rs in pieces:
            z = np.array(zseq, dtype=np.int32)
            fb = _flat_basis_z(settings.basis_set, z)
            n, m, na = fb.nao, len(rs), len(z)
            xyz = np.ascontiguousarray(np.stack([coords[list(requests[r][0]) + list(requests[r][1])] for r in rs]))
            D = np.zeros((m, n, n)); J = np.zeros((m, n, n))
            for pos, r in enumerate(rs):
                D[pos, n - nk:, n - nk:] = requests[r][2]
            mols = np.zeros(m, dtype=_MOL_DTYPE)
            mols["n_atoms"] = na; mols["atomic_numbers"] = z.ctypes.data
            mols["xyz"] = xyz.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(na * 3 * 8)
            mols["multiplicity"] = 1; mols["nelec"] = int(np.sum(z))
            bas = np.zeros(1, dtype=_BAS_DTYPE); bas[0] = _basis_record(fb, na)
            capi.check(lib.mqc_hip_coulomb_batch(ctx, m, mols.ctypes.data_as(C.POINTER(capi.Molecule)),
                                                 bas.ctypes.data_as(C.POINTER(capi.Basis)), n_src, capi.dptr(D), capi.dptr(J)))
            for pos, r in enumerate(rs):
                out[r] = J[pos, :n - nk, :n - nk].copy()
        return out      # type: ignore[return-value]

    return coulomb


def hip_solver(system: FragmentedSystem, settings: ScfSettings) -> Solver:
    """The product solver: every job of a pass in ONE engine call, grouped by (element sequence, field size)."""
    from .methods import run_hip_scf_groups
    coords = np.ascontiguousarray(system.coordinates.T)
    z_all = np.asarray(system.element_numbers)

    def solve(jobs: Sequence[EmbeddedJob]) -> List[EmbeddedResult]:
        by_key: Dict[tuple, List[int]] = {}
        for k, job in enumerate(jobs):
            by_key.setdefault((tuple(int(v) for v in z_all[list(job.atoms)]), len(job.field_atoms), job.h_extra is not None), []).append(k)
        groups, index = [], []
        for (zseq, npc, hx), ks in by_key.items():
            xyz = np.stack([coords[list(jobs[k].atoms)] for k in ks])
            g = FragmentGroup(np.array(zseq, dtype=np.int32), xyz, np.zeros(len(ks), dtype=np.int32))
            if npc:
                g.point_charge_xyz = np.stack([coords[np.asarray(jobs[k].field_atoms, dtype=np.int64)] for k in ks])
                g.point_charges = np.stack([np.asarray(jobs[k].field_charges, dtype=np.float64) for k in ks])
            if hx:
                g.h_extra = np.stack([jobs[k].h_extra for k in ks])
            groups.append(g); index.append(ks)
        extras_out: list = []
        recs = run_hip_scf_groups(settings, groups, extras=("density", "embedding_matrix", "mulliken_charges"),
                                  extras_out=extras_out)
        out = [EmbeddedResult() for _ in jobs]
        for ks, rec, ex in zip(index, recs, extras_out):
            for pos, k in enumerate(ks):
                r = out[k]
                if rec["has_error"][pos]:
                    r.error = bytes(rec["message"][pos]).split(b"\0", 1)[0].decode(errors="replace")
                    continue
                r.e_total = float(rec["e_total"][pos]); r.e_embedding = float(rec["e_embedding"][pos])
                r.iterations = int(rec["iterations"][pos])
                r.density = ex["density"][pos]; r.charges = ex["mulliken_charges"][pos]
                r.u = ex["embedding_matrix"][pos] if (len(jobs[k].field_atoms) or jobs[k].h_extra is not None) else None
        return out

    return solve


@dataclass
class FmoRun:
    energy: float
    monomer_energy: np.ndarray
    pair_sum: float
    response_sum: float
    outer_iterations: int
    converged: bool
    charges: np.ndarray                        # Mulliken charge of every atom of the system, in the settled field
    pair_corrections: Dict[Tuple[int, ...], float] = field(default_factory=dict)   # dE_S of every n-mer, |S| >= 2
    scf_iterations: int = 0
    errors: List[str] = field(default_factory=list)


def run_fmo2(system: FragmentedSystem, settings: ScfSettings, expansion: str = "fmo", max_outer: int = 50,
             outer_tol: float = 1.0e-7, rank: int = 0, world: int = 1,
             allreduce: Optional[Callable[[np.ndarray], np.ndarray]] = None, solver: Optional[Solver] = None,
             esp: str = "ptc", resppc: float = 2.0, coulomb: Optional[Coulomb] = None, level: int = 2,
             far_field: str = "mulliken") -> FmoRun:
    """FMO2 ("fmo") or electrostatically embedded MBE2 ("mbe") of whole-molecule fragments; the field of the others is
    Mulliken point charges (`esp = "ptc"`) or, for fragments within `resppc`, bare nuclei plus the exact Coulomb
    operator of their electrons (`esp = "exact"`, the reference's FMO default).

    With `world` > 1 every rank runs this on the same system, solves the fragments / pairs with index = rank (mod world)
    and `allreduce` (element-wise SUM over ranks of a float64 array) is the one exchange per pass.  `level` is the
    largest n-mer (2 = FMO2; at level = number of fragments the corrections telescope to the supermolecular energy)."""
    if expansion not in ("fmo", "mbe"):
        raise ValueError("expansion must be 'fmo' or 'mbe'")
    if world > 1 and allreduce is None:
        raise ValueError("several ranks need an allreduce")
    if esp not in ("ptc", "exact", "none"):
        raise ValueError("esp must be 'ptc', 'exact' or 'none'")
    if far_field not in ("mulliken", "ignore"):
        raise ValueError("far_field must be 'mulliken' or 'ignore' (CHELPG charges are not built)")
    solve = solver or hip_solver(system, settings)
    cutoff = resppc if esp == "exact" else 0.0                  # effective_resppc, :1032-1045
    exact = esp == "exact" and cutoff != 0.0
    if exact and coulomb is None:
        coulomb = hip_cross_coulomb(system, settings)
    z_all = np.asarray(system.element_numbers)
    share = allreduce if world > 1 else (lambda a: a)
    n_atoms, nfrag = len(system.element_numbers), system.n_monomers
    frags = [tuple(int(a) for a in m) for m in system.monomers]
    sep = fragment_separations(system) if (exact and cutoff >= 0.0) else None

    def near_of(group: Sequence[int]) -> List[int]:
        if not exact:
            return []
        if sep is None:                                         # resppc < 0: everything exact
            return [k for k in range(nfrag) if k not in group]
        close = np.min(sep[list(group)], axis=0) <= cutoff
        return [k for k in range(nfrag) if close[k] and k not in group]
    mine = [i for i in range(nfrag) if i % world == rank]
    errors: List[str] = []
    total_iters = 0

    # what a pass leaves behind, the same on every rank after the exchange
    e_total = np.zeros(nfrag); e_int = np.zeros(nfrag); q_all = np.zeros(n_atoms)
    dens: List[Optional[np.ndarray]] = [None] * nfrag
    nao = [0] * nfrag

    def embedded_jobs(groups: Sequence[Sequence[int]]) -> List[EmbeddedJob]:
        # embedding_operator (:1077-1160): near fragments = nuclei + exact J, the others = Mulliken charges; the
        # Coulomb operators of all (group, near fragment) pairs of the pass come from ONE batched request
        jobs, requests, owner = [], [], []
        for g, group in enumerate(groups):
            atoms = tuple(a for m in group for a in frags[m])
            outside = np.ones(n_atoms, dtype=bool); outside[list(atoms)] = False
            near = near_of(group)
            is_near = np.zeros(n_atoms, dtype=bool)
            for k in near:
                is_near[list(frags[k])] = True
            if far_field == "ignore":                  # distant atoms drop out entirely when they are being ignored (:1119-1124)
                outside &= is_near
            out = np.nonzero(outside)[0]
            w = np.where(is_near[out], z_all[out].astype(np.float64), q_all[out])
            jobs.append(EmbeddedJob(atoms, out, w, None))
            for k in near:
                requests.append((atoms, frags[k], dens[k])); owner.append(g)
        if requests:
            for g, j in zip(owner, coulomb(requests)):
                jobs[g].h_extra = j if jobs[g].h_extra is None else jobs[g].h_extra + j
        return jobs

    def any_failure() -> bool:
        # every rank must leave together: the count of failed SCFs is shared like the energies
        return bool(float(share(np.array([float(len(errors))]))[0]))

    def refused(outer_done: int, why: Optional[str] = None) -> FmoRun:
        # the reference returns an error and no energy (run_fmo2 :489-494, calculate_monomers :1560-1563,
        # calculate_polymers' early returns); here: NaN, not converged, and the reason on every rank
        mono_now = e_total.copy() if expansion == "mbe" else e_int.copy()
        if why:
            errors.append(why)
        return FmoRun(float("nan"), mono_now, 0.0, 0.0, outer_done, False, q_all.copy(), {}, total_iters,
                      errors or ["an SCF failed on another rank"])

    def monomer_pass(bare: bool):
        nonlocal total_iters
        monomer_jobs = [EmbeddedJob(frags[i]) for i in mine] if (bare or esp == "none") else embedded_jobs([[i] for i in mine])
        res = solve(monomer_jobs)
        new_e = np.zeros(nfrag); new_i = np.zeros(nfrag); new_q = np.zeros(n_atoms)
        for i, r in zip(mine, res):
            if r.error:
                errors.append("fragment %d: %s" % (i, r.error)); continue
            new_e[i] = r.e_total; new_i[i] = r.e_total - r.e_embedding
            new_q[list(frags[i])] = r.charges
            dens[i] = r.density; nao[i] = r.density.shape[0]
            total_iters += r.iterations
        e_total[:] = share(new_e); e_int[:] = share(new_i); q_all[:] = share(new_q)
        if world > 1 and (expansion == "fmo" or exact):
            # the pair phase (d_split) and the exact field need every monomer density: exchanged flattened
            sizes = share(np.array([float(nao[i]) if i in mine else 0.0 for i in range(nfrag)])).astype(int)
            flat = np.zeros(int(np.sum(sizes ** 2)))
            off = np.concatenate([[0], np.cumsum(sizes ** 2)])
            for i in mine:
                if dens[i] is not None:
                    flat[off[i]:off[i + 1]] = dens[i].reshape(-1)
            flat = share(flat)
            for i in range(nfrag):
                nao[i] = int(sizes[i]); dens[i] = flat[off[i]:off[i + 1]].reshape(sizes[i], sizes[i])

    monomer_pass(True)
    if any_failure():
        # a fragment without a density cannot field the next pass or the pair phase (solve_fragment's error return)
        return refused(0)
    e_prev = float(np.sum(e_int))
    converged, outer_done = False, 0
    change = 0.0
    if esp == "none":                                       # no field: the bare pass is the answer (:1523-1527)
        converged, outer_done, max_outer = True, 1, 0
    for outer in range(1, max_outer + 1):
        monomer_pass(False)
        if any_failure():
            return refused(outer)
        e_sum = float(np.sum(e_int))
        outer_done = outer
        change = abs(e_sum - e_prev)
        if change < outer_tol:
            converged = True
            break
        e_prev = e_sum
    if not converged:
        # calculate_monomers :1560-1563: an outer loop that has not settled is an error, not a total
        return refused(outer_done, "fmo: the outer SCF did not settle in %d passes; the monomer sum was still moving by %.3e Hartree"
                       % (max_outer, change))
    mono = e_total.copy() if expansion == "mbe" else e_int.copy()

    # every n-mer from pairs up to the level, one bag of independent tasks = ONE batch call (calculate_polymers,
    # :1566-1689): value_S = e_internal + e_resp; dE_S = value_S - sum of dE_T over the proper non-empty subsets T
    # (subtract_subsets, :1761-1778), with dE_{i} = E_i, so the response sits inside the recursion
    level = min(level, nfrag)
    terms = [t for size in range(2, level + 1) for t in itertools.combinations(range(nfrag), size)]
    my_terms = [t for k, t in enumerate(terms) if k % world == rank]
    nmer_jobs = ([EmbeddedJob(tuple(a for m in t for a in frags[m])) for t in my_terms] if esp == "none"
                 else embedded_jobs([list(t) for t in my_terms]))
    res = solve(nmer_jobs) if nmer_jobs else []
    value = np.zeros(len(terms)); resp = np.zeros(len(terms))
    index = {t: k for k, t in enumerate(terms)}
    for members, r in zip(my_terms, res):
        if r.error:
            errors.append("n-mer %s: %s" % (members, r.error)); continue
        total_iters += r.iterations
        e_internal, e_resp = r.e_total, 0.0
        if r.u is not None and expansion != "mbe":
            d_split = np.zeros_like(r.density)
            at = 0
            for m in members:
                d_split[at:at + nao[m], at:at + nao[m]] = dens[m]; at += nao[m]
            e_internal -= r.e_embedding
            e_resp = float(np.sum((r.density - d_split) * r.u))
        value[index[members]] = e_internal + e_resp
        resp[index[members]] = e_resp
    if any_failure():
        # a failed n-mer would enter as value 0 and shift the total by a whole pair energy: no total instead
        # (calculate_polymers returns on nmer_term's error, :1652-1653), and every rank learns of it
        return refused(outer_done)
    value = share(value); resp = share(resp)
    corr: Dict[Tuple[int, ...], float] = {(i,): float(mono[i]) for i in range(nfrag)}
    for members in terms:                                   # ordered by size: every subset is final before its supersets
        corr[members] = float(value[index[members]]) - sum(corr[t] for sub in range(1, len(members))
                                                           for t in itertools.combinations(members, sub))
    pair_sum = float(sum(corr[t] for t in terms))
    return FmoRun(float(np.sum(mono)) + pair_sum, mono, pair_sum, float(np.sum(resp)), outer_done, converged and not errors,
                  q_all.copy(), {t: corr[t] for t in terms}, total_iters, errors)
