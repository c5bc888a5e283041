"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's FMO2 / EE-MBE driver with point-charge embedding.

Follows backends/libcint/mqc_libcint_fmo.f90 for non-covalent fragments (whole molecules, no caps, no AFO):
  run_fmo2 :425-497, calculate_monomers :1484-1564 (bare pass, then outer passes until the monomer sum moves by less
  than outer_tol), solve_fragment :1408-1482 / inner_scf :1950-1999 (energy = E_scf - tr(D u)), embedding_operator
  :1077-1160 with esp = "ptc" (every outside atom a Mulliken point charge, effective_resppc :1032-1045),
  nmer_term :1162-1274 (e_internal, e_resp), calculate_polymers :1566-1689 (level 2: dE_IJ = E'_IJ - E'_I - E'_J).
Pinned by the reference's manifest row "EE-MBE water trimer 6-31g (CPU)" = -227.9704573337
(validation/validation_tests_cpu.json:2194-2198; deck eembe_water3.json: expansion ee-mbe, level 2, Mulliken far field).
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


def run_fmo2(make_mol: Callable[[Sequence[int]], "so.OracleMol"], z: np.ndarray, xyz: np.ndarray,
             fragments: Sequence[Sequence[int]], expansion: str = "fmo", max_outer: int = 50, outer_tol: float = 1e-7,
             scf_max_iter: int = 100, e_tol: float = 1e-9, d_tol: float = 1e-7) -> FmoOracleResult:
    """make_mol(atom indices) -> OracleMol of those atoms; z (n_atoms,), xyz (n_atoms, 3) Bohr."""
    n_atoms, nfrag = len(z), len(fragments)
    frags = [list(map(int, f)) for f in fragments]
    mols = [make_mol(f) for f in frags]
    nelec = [int(sum(z[f])) for f in frags]

    def field_of(mol, inside, q_all):
        out = [a for a in range(n_atoms) if a not in inside]
        if not out:
            return None
        return so.point_charge_potential(mol, xyz[out], q_all[out])

    def solve(i, q_all, bare):
        u = None if bare else field_of(mols[i], set(frags[i]), q_all)
        r = so.run_rhf(mols[i], nelec[i], max_iter=scf_max_iter, e_tol=e_tol, d_tol=d_tol, guess="gwh", h_extra=u)
        S, _, _ = so.int1e(mols[i])
        e_int = r.energy - (float(np.sum(r.D * u)) if u is not None else 0.0)
        return r.energy, e_int, r.D, so.mulliken_charges(mols[i], r.D, S)

    q_all = np.zeros(n_atoms)
    state = [solve(i, q_all, True) for i in range(nfrag)]
    e_prev = sum(s[1] for s in state)
    converged, outer_done = False, 0
    for outer in range(1, max_outer + 1):
        for i in range(nfrag):
            q_all[frags[i]] = state[i][3]
        state = [solve(i, q_all, False) for i in range(nfrag)]
        e_sum = sum(s[1] for s in state)
        outer_done = outer
        if abs(e_sum - e_prev) < outer_tol:
            converged = True
            break
        e_prev = e_sum
    for i in range(nfrag):
        q_all[frags[i]] = state[i][3]
    mono = np.array([s[0] if expansion == "mbe" else s[1] for s in state])
    pair_sum, response_sum, corr = 0.0, 0.0, {}
    for i, j in itertools.combinations(range(nfrag), 2):
        atoms = frags[i] + frags[j]
        mol = make_mol(atoms)
        u = field_of(mol, set(atoms), q_all)
        r = so.run_rhf(mol, nelec[i] + nelec[j], max_iter=scf_max_iter, e_tol=e_tol, d_tol=d_tol, guess="gwh", h_extra=u)
        e_internal, e_resp = r.energy, 0.0
        if u is not None and expansion != "mbe":
            ni = mols[i].nao
            d_split = np.zeros_like(r.D)
            d_split[:ni, :ni] = state[i][2]; d_split[ni:, ni:] = state[j][2]
            e_internal -= float(np.sum(r.D * u))
            e_resp = float(np.sum((r.D - d_split) * u))
        c = e_internal + e_resp - mono[i] - mono[j]
        corr[(i, j)] = c
        pair_sum += c; response_sum += e_resp
    return FmoOracleResult(float(np.sum(mono) + pair_sum), mono, pair_sum, response_sum, outer_done, converged, q_all.copy(), corr)
