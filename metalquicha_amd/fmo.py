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

Scope: whole-molecule fragments (no severed bonds, caps or AFO projector), closed shells, level 2, esp = "ptc" with
Mulliken charges -- the exact-ESP operator of near fragments (`local_coulomb`) and CHELPG stay with the reference.
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
    field_atoms: Tuple[int, ...] = ()        # outside atoms whose charges make the field (empty = in vacuum)


@dataclass
class EmbeddedResult:
    e_total: float = 0.0
    e_embedding: float = 0.0                 # tr(D u)
    iterations: int = 0
    density: Optional[np.ndarray] = None
    charges: Optional[np.ndarray] = None     # Mulliken, one per atom of the job
    u: Optional[np.ndarray] = None
    error: str = ""


Solver = Callable[[Sequence[EmbeddedJob], np.ndarray], List[EmbeddedResult]]


def hip_solver(system: FragmentedSystem, settings: ScfSettings) -> Solver:
    """The product solver: every job of a pass in ONE engine call, grouped by (element sequence, field size)."""
    from .methods import run_hip_scf_groups
    coords = np.ascontiguousarray(system.coordinates.T)
    z_all = np.asarray(system.element_numbers)

    def solve(jobs: Sequence[EmbeddedJob], q_all: np.ndarray) -> List[EmbeddedResult]:
        by_key: Dict[tuple, List[int]] = {}
        for k, job in enumerate(jobs):
            by_key.setdefault((tuple(int(v) for v in z_all[list(job.atoms)]), len(job.field_atoms)), []).append(k)
        groups, index = [], []
        for (zseq, npc), ks in by_key.items():
            xyz = np.stack([coords[list(jobs[k].atoms)] for k in ks])
            g = FragmentGroup(np.array(zseq, dtype=np.int32), xyz, np.zeros(len(ks), dtype=np.int32))
            if npc:
                g.point_charge_xyz = np.stack([coords[list(jobs[k].field_atoms)] for k in ks])
                g.point_charges = np.stack([q_all[list(jobs[k].field_atoms)] for k in ks])
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
                r.u = ex["embedding_matrix"][pos] if jobs[k].field_atoms else None
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
    pair_corrections: Dict[Tuple[int, int], float] = field(default_factory=dict)
    scf_iterations: int = 0
    errors: List[str] = field(default_factory=list)


def run_fmo2(system: FragmentedSystem, settings: ScfSettings, expansion: str = "fmo", max_outer: int = 50,
             outer_tol: float = 1.0e-7, rank: int = 0, world: int = 1,
             allreduce: Optional[Callable[[np.ndarray], np.ndarray]] = None, solver: Optional[Solver] = None) -> FmoRun:
    """FMO2 ("fmo") or electrostatically embedded MBE2 ("mbe") of whole-molecule fragments in Mulliken point charges.

    With `world` > 1 every rank runs this on the same system, solves the fragments / pairs with index = rank (mod world)
    and `allreduce` (element-wise SUM over ranks of a float64 array) is the one exchange per pass."""
    if expansion not in ("fmo", "mbe"):
        raise ValueError("expansion must be 'fmo' or 'mbe'")
    if world > 1 and allreduce is None:
        raise ValueError("several ranks need an allreduce")
    solve = solver or hip_solver(system, settings)
    share = allreduce if world > 1 else (lambda a: a)
    n_atoms, nfrag = len(system.element_numbers), system.n_monomers
    frags = [tuple(int(a) for a in m) for m in system.monomers]
    mine = [i for i in range(nfrag) if i % world == rank]
    errors: List[str] = []
    total_iters = 0

    # what a pass leaves behind, the same on every rank after the exchange
    e_total = np.zeros(nfrag); e_int = np.zeros(nfrag); q_all = np.zeros(n_atoms)
    dens: List[Optional[np.ndarray]] = [None] * nfrag
    nao = [0] * nfrag

    def monomer_pass(bare: bool):
        nonlocal total_iters
        jobs = [EmbeddedJob(frags[i], () if bare else tuple(a for a in range(n_atoms) if a not in set(frags[i]))) for i in mine]
        res = solve(jobs, q_all)
        new_e = np.zeros(nfrag); new_i = np.zeros(nfrag); new_q = np.zeros(n_atoms)
        for i, r in zip(mine, res):
            if r.error:
                errors.append("fragment %d: %s" % (i, r.error)); continue
            new_e[i] = r.e_total; new_i[i] = r.e_total - r.e_embedding
            new_q[list(frags[i])] = r.charges
            dens[i] = r.density; nao[i] = r.density.shape[0]
            total_iters += r.iterations
        e_total[:] = share(new_e); e_int[:] = share(new_i); q_all[:] = share(new_q)
        if world > 1 and expansion == "fmo":
            # the pair phase needs every monomer density (d_split): exchanged flattened, one slot per fragment
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
    e_prev = float(np.sum(e_int))
    converged, outer_done = False, 0
    for outer in range(1, max_outer + 1):
        monomer_pass(False)
        e_sum = float(np.sum(e_int))
        outer_done = outer
        if abs(e_sum - e_prev) < outer_tol:
            converged = True
            break
        e_prev = e_sum
    mono = e_total.copy() if expansion == "mbe" else e_int.copy()

    pairs = list(itertools.combinations(range(nfrag), 2))
    my_pairs = [p for t, p in enumerate(pairs) if t % world == rank]
    jobs = []
    for i, j in my_pairs:
        atoms = frags[i] + frags[j]
        inside = set(atoms)
        jobs.append(EmbeddedJob(atoms, tuple(a for a in range(n_atoms) if a not in inside)))
    res = solve(jobs, q_all) if jobs else []
    corr = np.zeros(len(pairs)); resp = np.zeros(len(pairs))
    index = {p: t for t, p in enumerate(pairs)}
    for (i, j), r in zip(my_pairs, res):
        if r.error:
            errors.append("pair (%d, %d): %s" % (i, j, r.error)); continue
        total_iters += r.iterations
        e_internal, e_resp = r.e_total, 0.0
        if r.u is not None and expansion != "mbe":
            ni = nao[i]
            d_split = np.zeros_like(r.density)
            d_split[:ni, :ni] = dens[i]; d_split[ni:, ni:] = dens[j]
            e_internal -= r.e_embedding
            e_resp = float(np.sum((r.density - d_split) * r.u))
        corr[index[(i, j)]] = e_internal + e_resp - mono[i] - mono[j]
        resp[index[(i, j)]] = e_resp
    corr = share(corr); resp = share(resp)
    pair_sum = float(np.sum(corr))
    return FmoRun(float(np.sum(mono)) + pair_sum, mono, pair_sum, float(np.sum(resp)), outer_done, converged and not errors,
                  q_all.copy(), {p: float(corr[t]) for p, t in index.items()}, total_iters, errors)
