"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's FMO2 / EE-MBE driver with point-charge embedding.

Follows backends/libcint/mqc_libcint_fmo.f90 for non-covalent fragments (whole molecules, no caps, no AFO):
  run_fmo2 :425-497, calculate_monomers :1484-1564 (bare pass, then outer passes until the monomer sum moves by less
  than outer_tol), solve_fragment :1408-1482 / inner_scf :1950-1999 (energy = E_scf - tr(D u)), embedding_operator
  :1077-1160 with esp = "ptc" (every outside atom a Mulliken point charge, effective_resppc :1032-1045),
  nmer_term :1162-1274 (e_internal, e_resp), calculate_polymers :1566-1689 (level 2: dE_IJ = E'_IJ - E'_I - E'_J).
  esp = "exact": fragments within `resppc` (near_fragments :1276-1316, closest atom pair over the sum of the van der
  Waals radii, unitless_distance :1318-1335) give their bare nuclei as charges and their electrons through the exact
  Coulomb operator J[D_K] in the fragment's basis (local_coulomb :1337-1406); the others stay Mulliken charges.
Pinned by the reference's manifest rows "EE-MBE water trimer 6-31g (CPU)" = -227.9704573337 (deck eembe_water3.json:
expansion ee-mbe, level 2, Mulliken far field) and "FMO2 water trimer 6-31g (CPU)" = -227.9705411684 (deck
fmo_water3.json: esp exact, resppc 2.0) -- validation/validation_tests_cpu.json:2188-2198.
Nothing in the product path imports this file.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Callable, List, Sequence

import numpy as np

from . import scf_oracle as so


@dataclass
class FmoOracleResult:
    energy: float
    monomer_energy: np.ndarray
    pair_sum: float
    response_sum: float
    outer_iterations: int
    converged: bool
    charges: np.ndarray
    pair_corrections: dict = field(default_factory=dict)


# Bondi's van der Waals radii with Rowland and Taylor's hydrogen, Angstrom (src/core/mqc_elements.f90:58-60), Z = 1..18
VDW_ANGSTROM = [1.10, 1.40, 1.81, 1.53, 1.92, 1.70, 1.55, 1.52, 1.47, 1.54, 2.27, 1.73, 1.84, 2.10, 1.80, 1.80, 1.75, 1.88]
ANGSTROM_TO_BOHR = 1.8897261254578281


def near_fragments(z, xyz, frags, group, resppc):
    """Fragments outside `group` whose closest atom pair lies within `resppc` van der Waals sums (all of them when
    resppc < 0)."""
    near = []
    inside = [a for g in group for a in frags[g]]
    for k in range(len(frags)):
        if k in group:
            continue
        if resppc < 0.0:
            near.append(k); continue
        best = min(np.linalg.norm(xyz[a] - xyz[b]) / ((VDW_ANGSTROM[int(z[a]) - 1] + VDW_ANGSTROM[int(z[b]) - 1]) * ANGSTROM_TO_BOHR)
                   for a in inside for b in frags[k])
        if best <= resppc:
            near.append(k)
    return near


def run_fmo2(make_mol: Callable[[Sequence[int]], "so.OracleMol"], z: np.ndarray, xyz: np.ndarray,
             fragments: Sequence[Sequence[int]], expansion: str = "fmo", max_outer: int = 50, outer_tol: float = 1e-7,
             scf_max_iter: int = 100, e_tol: float = 1e-9, d_tol: float = 1e-7, esp: str = "ptc",
             resppc: float = 2.0, level: int = 2, far_field: str = "mulliken",
             scf_extra: Callable[[Sequence[int], "so.OracleMol"], dict] = None) -> FmoOracleResult:
    """make_mol(atom indices) -> OracleMol of those atoms; z (n_atoms,), xyz (n_atoms, 3) Bohr.

    scf_extra(atom indices, mol) -> extra keyword arguments of so.run_rhf for that fragment / n-mer (`aux` for a
    density-fitted run, `xc` for a Kohn-Sham one).  The reference's driver itself calls run_libcint_rhf without them
    (inner_scf :1950-1999): this is the same driver around the density-fitted Kohn-Sham SCF the cuEST path runs, the
    form BASELINE.json's configs[4] ("FMO-2 DF-RKS") names."""
    n_atoms, nfrag = len(z), len(fragments)
    frags = [list(map(int, f)) for f in fragments]
    mols = [make_mol(f) for f in frags]
    nelec = [int(sum(z[f])) for f in frags]
    cutoff = 0.0 if esp == "ptc" else resppc          # effective_resppc
    state: list = []
    extra = scf_extra or (lambda atoms, mol: {})

    def field_of(mol, group, q_all):
        inside = [a for g in group for a in frags[g]]
        near = near_fragments(z, xyz, frags, group, cutoff) if (esp == "exact" and cutoff != 0.0) else []
        near_atoms = set(a for k in near for a in frags[k])
        if esp == "none":
            return None
        out = [a for a in range(n_atoms) if a not in inside and (a in near_atoms or far_field != "ignore")]
        if not out and not near:
            return None
        w = np.array([float(z[a]) if a in near_atoms else q_all[a] for a in out])
        u = so.point_charge_potential(mol, xyz[out], w)
        n0 = mol.nao
        for k in near:
            # the electrons of a near fragment: J[D_K] over the supersystem group + K, group block (local_coulomb)
            sup = make_mol(inside + frags[k])
            eri = so.eri4(sup)
            u = u + np.einsum("ijkl,kl->ij", eri[:n0, :n0, n0:, n0:], state[k][2])
        return u

    def solve(i, q_all, bare):
        u = None if bare else field_of(mols[i], [i], q_all)
        r = so.run_rhf(mols[i], nelec[i], max_iter=scf_max_iter, e_tol=e_tol, d_tol=d_tol, guess="gwh", h_extra=u,
                       **extra(frags[i], mols[i]))
        S, _, _ = so.int1e(mols[i])
        e_int = r.energy - (float(np.sum(r.D * u)) if u is not None else 0.0)
        return r.energy, e_int, r.D, so.mulliken_charges(mols[i], r.D, S)

    q_all = np.zeros(n_atoms)
    state[:] = [solve(i, q_all, True) for i in range(nfrag)]
    e_prev = sum(s[1] for s in state)
    converged, outer_done = False, 0
    if esp == "none":
        converged, outer_done, max_outer = True, 1, 0
    for outer in range(1, max_outer + 1):
        for i in range(nfrag):
            q_all[frags[i]] = state[i][3]
        state[:] = [solve(i, q_all, False) for i in range(nfrag)]      # every fragment reads the previous pass
        e_sum = sum(s[1] for s in state)
        outer_done = outer
        if abs(e_sum - e_prev) < outer_tol:
            converged = True
            break
        e_prev = e_sum
    for i in range(nfrag):
        q_all[frags[i]] = state[i][3]
    mono = np.array([s[0] if expansion == "mbe" else s[1] for s in state])
    # every n-mer from pairs up to the level (calculate_polymers :1566-1689): value_S = e_internal + e_resp, then
    # dE_S = value_S - sum over proper non-empty subsets T of dE_T (subtract_subsets :1761-1778), dE_{i} = E_i
    level = min(level, nfrag)
    response_sum, corr = 0.0, {(i,): float(mono[i]) for i in range(nfrag)}
    for size in range(2, level + 1):
        for members in itertools.combinations(range(nfrag), size):
            atoms = [a for m in members for a in frags[m]]
            mol = make_mol(atoms)
            u = field_of(mol, list(members), q_all)
            r = so.run_rhf(mol, sum(nelec[m] for m in members), max_iter=scf_max_iter, e_tol=e_tol, d_tol=d_tol, guess="gwh", h_extra=u,
                           **extra(atoms, mol))
            e_internal, e_resp = r.energy, 0.0
            if u is not None and expansion != "mbe":
                d_split = np.zeros_like(r.D)
                at = 0
                for m in members:
                    nm = mols[m].nao
                    d_split[at:at + nm, at:at + nm] = state[m][2]; at += nm
                e_internal -= float(np.sum(r.D * u))
                e_resp = float(np.sum((r.D - d_split) * u))
            corr[members] = e_internal + e_resp
            response_sum += e_resp
    for size in range(2, level + 1):
        for members in itertools.combinations(range(nfrag), size):
            for sub in range(1, size):
                for t in itertools.combinations(members, sub):
                    corr[members] -= corr[t]
    pair_sum = float(sum(c for t, c in corr.items() if len(t) >= 2))
    return FmoOracleResult(float(np.sum(mono) + pair_sum), mono, pair_sum, response_sum, outer_done, converged, q_all.copy(),
                           {t: c for t, c in corr.items() if len(t) >= 2})
