"""Host-side mirror of the reference's GMBE caller (generalised many-body expansion over OVERLAPPING fragments).

What `mqc_driver.f90:449-503` and `src/fragmentation/gmbe/mqc_gmbe_utils.f90` do on the Fortran side, restated so that
BASELINE.json's configs[3] ((H2O)64 GMBE-2) can be driven through the HIP engine from Python the way `mbe.py` drives
MBE: primaries = all level-tuples of the (overlapping) base fragments, optionally distance-screened; subsystems and
their integer coefficients by the principle of inclusion-exclusion over cliques of primaries with a non-empty common
atom set (`gmbe_enumerate_pie_terms` / `dfs_pie_accumulate`, :533-772: depth-first, sign +1 for odd clique size and
-1 for even, equal atom sets merged, depth limited by `max_intersection_level`); total = sum_i c_i E(atom set i), the
gradient the same sum scattered onto the system's atoms.  The SCFs themselves go to the engine as batches grouped by
element sequence, exactly as in `mbe.run_mbe`.

Only non-covalent overlaps are handled here (a subsystem is the bare set of its atoms, neutral singlet unless the
per-atom charges say otherwise): hydrogen capping of cut bonds stays with the reference's `physical_fragment` code.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .mbe import FragmentedSystem, min_intermonomer_distance, partition_terms
from .methods import FragmentGroup, ScfSettings


def polymer_atoms(system: FragmentedSystem, polymer: Sequence[int]) -> Tuple[int, ...]:
    """Unique atoms of the base fragments of a polymer, in order of first appearance (compute_polymer_atoms, :263-318)."""
    seen, out = set(), []
    for m in polymer:
        for a in system.monomers[m]:
            a = int(a)
            if a not in seen:
                seen.add(a); out.append(a)
    return tuple(out)


def generate_primaries(system: FragmentedSystem, level: int, cutoffs: Optional[Dict[int, float]] = None) -> List[Tuple[int, ...]]:
    """GMBE(1): the base fragments; GMBE(N): all C(M, N) N-tuples, distance-screened like MBE n-mers
    (mqc_driver.f90:455-487), largest first."""
    m = system.n_monomers
    if level == 1:
        prim = [(i,) for i in range(m)]
    else:
        prim = []
        cut = (cutoffs or {}).get(level)
        for t in itertools.combinations(range(m), level):
            if cut is not None and max(min_intermonomer_distance(system, a, b) for a, b in itertools.combinations(t, 2)) > cut:
                continue
            prim.append(t)
    prim.sort(key=lambda t: -len(polymer_atoms(system, t)))
    return prim


def enumerate_pie_terms(primary_atom_sets: Sequence[Sequence[int]], max_k_level: int = 999) -> Tuple[List[Tuple[int, ...]], np.ndarray]:
    """-> (unique atom sets, integer coefficients).  Depth-first over cliques of primaries whose common atom set is not
    empty; a clique of size k contributes (-1)^(k+1) to the coefficient of its intersection (dfs_pie_accumulate)."""
    prim = [frozenset(int(a) for a in s) for s in primary_atom_sets]
    coef: Dict[frozenset, int] = {}
    order: List[frozenset] = []

    def visit(current: frozenset, size: int, candidates: List[int]):
        if not current:
            return
        if current not in coef:
            coef[current] = 0; order.append(current)
        coef[current] += 1 if size % 2 == 1 else -1
        if size >= max_k_level:
            return
        for pos, c in enumerate(candidates):
            new = current & prim[c]
            if not new:
                continue
            visit(new, size + 1, [d for d in candidates[pos + 1:] if new & prim[d]])

    for i in range(len(prim)):
        visit(prim[i], 1, list(range(i + 1, len(prim))))
    sets = [tuple(sorted(s)) for s in order]
    return sets, np.array([coef[s] for s in order], dtype=np.int64)


@dataclass
class GmbeRun:
    atom_sets: List[Tuple[int, ...]]
    coefficients: np.ndarray
    energies: np.ndarray            # zero for subsystems this rank does not own
    iterations: np.ndarray
    owned: np.ndarray
    errors: List[str]
    total: float                    # sum of c_i E_i over the owned subsystems (all-reduce over ranks for the GMBE energy)
    gradient: Optional[np.ndarray] = None


def _subsystem_groups(system: FragmentedSystem, atom_sets: Sequence[Tuple[int, ...]], atom_charges: Optional[np.ndarray]):
    coords = np.ascontiguousarray(system.coordinates.T)
    by_z: Dict[Tuple[int, ...], List[int]] = {}
    for pos, s in enumerate(atom_sets):
        by_z.setdefault(tuple(int(z) for z in system.element_numbers[list(s)]), []).append(pos)
    groups, positions, atoms_of = [], [], []
    for z, plist in by_z.items():
        idx = np.array([atom_sets[p] for p in plist])
        charges = np.zeros(len(plist), dtype=np.int32) if atom_charges is None else \
            np.array([int(round(float(np.sum(atom_charges[list(atom_sets[p])])))) for p in plist], dtype=np.int32)
        groups.append(FragmentGroup(np.array(z, dtype=np.int32), coords[idx], charges))
        positions.append(np.array(plist)); atoms_of.append(idx)
    return groups, positions, atoms_of


def run_gmbe(system: FragmentedSystem, settings: ScfSettings, level: int = 1, cutoffs: Optional[Dict[int, float]] = None,
             max_intersection_level: int = 999, rank: int = 0, world: int = 1, want_gradient: bool = False,
             atom_charges: Optional[np.ndarray] = None) -> GmbeRun:
    """One GMBE(level) evaluation of this rank's share (static round-robin over the subsystems, largest first)."""
    from .methods import run_hip_scf_groups
    primaries = generate_primaries(system, level, cutoffs)
    sets, coef = enumerate_pie_terms([polymer_atoms(system, p) for p in primaries], max_intersection_level)
    keep = [i for i in range(len(sets)) if coef[i] != 0]
    keep.sort(key=lambda i: -len(sets[i]))
    sets = [sets[i] for i in keep]; coef = coef[keep]
    owned = np.asarray(partition_terms(len(sets), rank, world))
    groups, positions, atoms_of = _subsystem_groups(system, [sets[i] for i in owned], atom_charges)
    energies = np.zeros(len(sets)); iters = np.zeros(len(sets), dtype=np.int64)
    errors: List[str] = []
    grads_out: list = []
    recs = run_hip_scf_groups(settings, groups, want_gradient=want_gradient, gradients_out=grads_out)
    grad = np.zeros((len(system.element_numbers), 3)) if want_gradient else None
    for g, (pos, rec) in enumerate(zip(positions, recs)):
        tix = owned[pos]
        ok = rec["has_error"] == 0
        energies[tix[ok]] = rec["e_total"][ok]
        iters[tix[ok]] = rec["iterations"][ok]
        for k in np.nonzero(~ok)[0]:
            errors.append("subsystem %s: %s" % (sets[tix[k]], bytes(rec["message"][k]).split(b"\0", 1)[0].decode(errors="replace")))
        if want_gradient:
            for k in np.nonzero(ok)[0]:
                if rec["has_gradient"][k]:
                    np.add.at(grad, atoms_of[g][k], coef[tix[k]] * grads_out[g][k])
    total = float(np.sum(coef[owned] * energies[owned]))
    return GmbeRun(sets, coef, energies, iters, owned, errors, total, grad)
