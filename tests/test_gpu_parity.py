"""GPU parity tests: every stage of the hot path through the C ABI against the CPU oracle.

Tolerances: the reference's own direct-vs-in-core bound is 1e-11 elementwise
(test/test_mqc_libcint_direct.f90:139); total energies must agree to <= 1e-8 Eh per fragment
(BASELINE.json north_star) and the known-answer programs assert 1e-9.
"""
import json
import os

import numpy as np
import pytest

from metalquicha_amd import methods
from metalquicha_amd.basis import ANGSTROM_TO_BOHR, SYMBOL_TO_Z
from tests import stages
from oracle import scf_oracle as so
from oracle import xc_oracle
from tests.helpers import fragment_bohr, oracle_mol, water_at, synthetic_density, recorded_oracle, scf_record

pytestmark = pytest.mark.gpu

WATER = ([8, 1, 1], [[0.0, 0.0, -0.1364652], [0.0, 1.4304924, 1.0826636], [0.0, -1.4304924, 1.0826636]])


def test_backend_available():
    assert methods.hip_backend_available()


@pytest.mark.parametrize("basis", ["sto-3g-check_rhf", "cc-pvdz"])
def test_int1e_matches_oracle(basis):
    frag = fragment_bohr(*WATER)
    S, T, V = stages.int1e(basis, frag)
    So, To, Vo = so.int1e(oracle_mol(basis, frag))
    assert np.max(np.abs(S - So)) < 1e-12
    assert np.max(np.abs(T - To)) < 1e-11
    assert np.max(np.abs(V - Vo)) < 1e-10
    assert np.allclose(np.diag(S), 1.0, atol=1e-12)


@pytest.mark.parametrize("basis", ["sto-3g-check_rhf", "cc-pvdz"])
def test_eri_packed_matches_oracle(basis):
    frag = fragment_bohr(*WATER)
    M = stages.eri_packed(basis, frag)
    ref = stages.pack_eri(so.eri4(oracle_mol(basis, frag)))
    assert M.shape == ref.shape
    assert np.max(np.abs(M - ref)) < 1e-11
    assert np.max(np.abs(M - M.T)) == 0.0


def test_eri_schwarz_screening_only_drops_small():
    rng = np.random.default_rng(3)
    xyz = np.vstack([water_at(rng, [0, 0, 0]), water_at(rng, [12.0, 0, 0])])
    frag = fragment_bohr([8, 1, 1, 8, 1, 1], xyz)
    full = stages.eri_packed("sto-3g", frag)
    scr = stages.eri_packed("sto-3g", frag, schwarz_tol=1e-9)
    assert np.max(np.abs(full - scr)) < 1e-9
    assert np.count_nonzero(scr) < np.count_nonzero(full)


@pytest.mark.parametrize("basis", ["sto-3g-check_rhf", "cc-pvdz"])
def test_jk_incore_matches_oracle(basis):
    frag = fragment_bohr(*WATER)
    mol = oracle_mol(basis, frag)
    D = synthetic_density(mol.nao)
    J, K = stages.jk_incore(basis, frag, D)
    Jo, Ko = so.build_jk_incore(so.eri4(mol), D)
    assert np.max(np.abs(J - Jo)) < 1e-11
    assert np.max(np.abs(K - Ko)) < 1e-11


@pytest.mark.parametrize("n", [2, 7, 24, 25, 48, 86, 114, 140, 141, 200, 256])
def test_syev_matches_lapack(n):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)); A = 0.5 * (A + A.T)
    w, V = stages.syev(A)
    wr = np.linalg.eigvalsh(A)
    assert np.max(np.abs(w - wr)) < 1e-12 * max(1.0, np.max(np.abs(wr)))
    assert np.max(np.abs(V.T @ V - np.eye(n))) < 1e-12
    assert np.max(np.abs(A @ V - V * w[None, :])) < 1e-11


def test_diis_coefficients_match_reference_algorithm():
    rng = np.random.default_rng(11)
    for n in (2, 3, 5, 8):
        E = rng.normal(size=(n, 40)) * 1e-4
        B = E @ E.T
        c, ok = stages.diis_coefficients(B)
        d = so.Diis(n, 1, 40)
        for k in range(n):
            d.push(np.zeros(1), E[k])
        ref = d.coefficients()
        assert ok and ref is not None
        assert abs(np.sum(c) - 1.0) < 1e-10
        assert np.max(np.abs(c - ref[:n])) < 1e-9 * max(1.0, np.max(np.abs(ref[:n])))
    c, ok = stages.diis_coefficients(np.array([[1.0]]))
    assert not ok          # fewer than two vectors: no extrapolation


def test_check_rhf_goldens():
    """validation/check_rhf.f90:79-143: H2 -1.1167143251, H2O -74.9658162796 (1e-9)."""
    st = methods.ScfSettings(basis_set="sto-3g-check_rhf", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    h2 = methods.run_hip_scf(st, fragment_bohr([1, 1], [[0, 0, 0], [0, 0, 1.4]]))
    assert not h2.has_error, h2.error_message
    assert abs(h2.e_nuclear - 1.0 / 1.4) < 1e-12
    assert abs(h2.energy.scf - (-1.1167143251)) < 1e-9
    w = methods.run_hip_scf(st, fragment_bohr(*WATER))
    assert not w.has_error, w.error_message
    assert w.scf_status == methods.SCF_CONVERGED
    assert abs(w.energy.scf - (-74.9658162796)) < 1e-9
    # DIIS changes the iteration count and nothing else
    st0 = methods.ScfSettings(basis_set="sto-3g-check_rhf", energy_tol=1e-10, density_tol=1e-8, guess="gwh",
                              use_diis=False, max_iter=200)
    w0 = methods.run_hip_scf(st0, fragment_bohr(*WATER))
    assert abs(w0.energy.scf - w.energy.scf) < 1e-9
    assert w.scf_iterations < w0.scf_iterations


def test_check_df_exact_golden_and_oracle_iterations():
    """validation/check_df.f90:55-57: H2O/cc-pVDZ exact-ERI RHF -76.0220988827 (1e-9)."""
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-11, density_tol=1e-9, guess="gwh", max_iter=200)
    frag = fragment_bohr(*WATER)
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    assert abs(r.energy.scf - (-76.0220988827)) < 1e-9
    o = so.run_rhf(oracle_mol("cc-pvdz", frag), 10, 200, 1e-11, 1e-9)
    assert abs(r.energy.scf - o.energy) < 1e-10
    assert r.scf_iterations == o.iterations
    assert abs(r.homo - o.eps[4]) < 1e-8 and abs(r.lumo - o.eps[5]) < 1e-8


def test_batch_of_mixed_fragments_matches_oracle():
    """Monomers and dimers in ONE batch call; each fragment within 1e-8 Eh of the oracle."""
    rng = np.random.default_rng(20260821)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [5.6, 0.3, 0.2], [0.1, 5.9, -0.4])]
    frags = [fragment_bohr([8, 1, 1], w) for w in ws]
    frags += [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([ws[i], ws[j]])) for i, j in ((0, 1), (0, 2), (1, 2))]
    st = methods.ScfSettings(basis_set="sto-3g", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    res = methods.run_hip_scf_batch(st, frags)
    for f, r in zip(frags, res):
        assert not r.has_error, r.error_message
        o = so.run_rhf(oracle_mol("sto-3g", f), int(f.nelec), 100, 1e-10, 1e-8)
        assert abs(r.energy.scf - o.energy) < 1e-9, (r.energy.scf, o.energy)
        assert r.scf_iterations == o.iterations


def test_water_dimer_ccpvdz_matches_oracle():
    rng = np.random.default_rng(5)
    xyz = np.vstack([water_at(rng, [0, 0, 0]), water_at(rng, [5.5, 0.5, -0.3])])
    frag = fragment_bohr([8, 1, 1, 8, 1, 1], xyz)
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    o = so.run_rhf(oracle_mol("cc-pvdz", frag), 20, 100, 1e-10, 1e-8)
    assert abs(r.energy.scf - o.energy) < 1e-8
    assert r.scf_iterations == o.iterations


# ---- f shells: the wave-cooperative general kernel (def2-TZVP, BASELINE.json configs[3]) --------------------------
def test_def2_tzvp_stage_integrals_match_oracle():
    """def2-TZVP water (n = 43, one f shell on O): S, T, V and the packed ERI tensor against the oracle -- every class
    with an f shell goes through kern_eri_general.hip, the s/p/d classes through the register kernels."""
    frag = fragment_bohr(*WATER)
    mol = oracle_mol("def2-tzvp", frag)
    assert mol.nao == 43 and int(np.max(mol.sh_l)) == 3
    S, T, V = stages.int1e("def2-tzvp", frag)
    So, To, Vo = so.int1e(mol)
    assert np.max(np.abs(S - So)) < 1e-12
    assert np.max(np.abs(T - To)) < 1e-10
    assert np.max(np.abs(V - Vo)) < 1e-10
    M = stages.eri_packed("def2-tzvp", frag)
    ref = stages.pack_eri(so.eri4(mol))
    assert not np.any(np.isnan(M))                 # the stage entry poisons the tensor first: every element was written
    assert np.max(np.abs(M - ref)) < 1e-11
    assert np.max(np.abs(M - M.T)) == 0.0
    scr = stages.eri_packed("def2-tzvp", frag, schwarz_tol=1e-12)
    assert np.max(np.abs(scr - ref)) < 1e-11


def test_def2_tzvp_water_dimer_batch_matches_oracle():
    """A configs[3]-shaped batch: def2-TZVP monomers (n = 43) and a dimer (n = 86) in one call, RHF and B3LYP
    (XC grid kernel with f functions), against the oracle."""
    rng = np.random.default_rng(86)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [5.5, 0.3, -0.2])]
    frags = [fragment_bohr([8, 1, 1], ws[0]), fragment_bohr([8, 1, 1], ws[1]), fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack(ws))]
    for fn in ("", "b3lyp"):
        st = methods.ScfSettings(basis_set="def2-tzvp", functional=fn, energy_tol=1e-9, density_tol=1e-7, guess="gwh")
        res = methods.run_hip_scf_batch(st, frags)
        for f, r in zip(frags[1:], res[1:]):
            assert not r.has_error, r.error_message
            def oracle(f=f, fn=fn):
                mol = oracle_mol("def2-tzvp", f)
                return scf_record(so.run_rhf(mol, int(f.nelec), 100, 1e-9, 1e-7, xc=xc_oracle.XCOracle(mol, fn, 3) if fn else None))
            o = recorded_oracle("def2tzvp_water", f, "def2-tzvp|%s|grid3|1e-9|1e-7|gwh" % fn, oracle)
            assert abs(r.energy.scf - o["energy"]) < 1e-8, (fn, r.energy.scf, o["energy"])
            assert r.scf_iterations == o["iterations"]


# ---- unrestricted Hartree-Fock ---------------------------------------------------------------------------------
def _uhf_cases():
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest_subset.json")))["cases"]
    return [c for c in cases if c["method"] == "hf" and c["unrestricted"] and c["driver"] == "Energy" and not c["density_fitting"]]


@pytest.mark.parametrize("case", _uhf_cases(), ids=[c["name"] for c in _uhf_cases()])
def test_manifest_uhf_goldens(case):
    """validation_tests_cpu.json uhf/ rows (OH doublet in STO-3G, cc-pVDZ, def2-SVP; O2 triplet in cc-pVDZ), tolerance
    1e-9: energy against the golden, iteration count, <S^2> and both spins' orbital energies against the oracle."""
    z = [SYMBOL_TO_Z[s.lower()] for s in case["symbols"]]
    frag = fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=case["multiplicity"])
    st = methods.ScfSettings(basis_set=case["basis"], energy_tol=1e-10, density_tol=1e-7, guess="gwh", max_iter=case["maxiter"])
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    assert r.scf_status == methods.SCF_CONVERGED
    assert abs(r.energy.scf - case["expected_energy"]) < 1e-9
    o = so.run_uhf(oracle_mol(case["basis"], frag), int(frag.nelec), case["multiplicity"], case["maxiter"], 1e-10, 1e-7)
    assert abs(r.energy.scf - o.energy) < 1e-9
    assert r.scf_iterations == o.iterations
    assert (r.n_alpha, r.n_beta) == (o.n_alpha, o.n_beta)
    assert abs(r.s_squared - o.s_squared) < 1e-6
    assert np.max(np.abs(r.orbital_energies[: o.n_alpha + 1] - o.eps_a[: o.n_alpha + 1])) < 1e-6
    assert np.max(np.abs(r.orbital_energies_beta[: o.n_beta + 1] - o.eps_b[: o.n_beta + 1])) < 1e-6


def test_uhf_batch_and_closed_shell_limit():
    """A batch of OH radicals at different bond lengths in one call (UHF topology group) against the oracle; a closed-shell
    molecule run unrestricted lands on the restricted energy with <S^2> = 0."""
    frags = [fragment_bohr([8, 1], [[0, 0, 0], [0, 0, d]], multiplicity=2) for d in (1.70, 1.83, 1.95, 2.10)]
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    res = methods.run_hip_scf_batch(st, frags)
    for f, r in zip(frags, res):
        assert not r.has_error, r.error_message
        o = so.run_uhf(oracle_mol("cc-pvdz", f), 9, 2, 100, 1e-10, 1e-8)
        assert abs(r.energy.scf - o.energy) < 1e-8, (r.energy.scf, o.energy)
        assert r.scf_iterations == o.iterations
    w = fragment_bohr(*WATER)
    ru = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-10, density_tol=1e-8, guess="gwh", unrestricted=True), w)
    rr = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-10, density_tol=1e-8, guess="gwh"), w)
    assert not ru.has_error, ru.error_message
    assert abs(ru.energy.scf - rr.energy.scf) < 1e-9
    assert abs(ru.s_squared) < 1e-8


def _uks_cases():
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest_subset.json")))["cases"]
    return [c for c in cases if c["method"] == "dft" and c["unrestricted"] and c["driver"] == "Energy"
            and c["functional"] in xc_oracle.RESTRICTED_FUNCTIONALS and not c["density_fitting"]]


@pytest.mark.parametrize("case", _uks_cases(), ids=[c["name"] for c in _uks_cases()])
def test_manifest_uks_goldens(case):
    """validation_tests_cpu.json udft/ rows (CH3 doublet SVWN / PBE / B3LYP, O2 triplet PBE; cc-pVDZ, grid 3), tolerance
    1e-9: spin-polarised functionals (VWN5 with the spin stiffness, VWN-RPA, B88, LYP, PBE x and c with polarised
    PW92), the cross-spin gradient term, exchange scaled by the functional's fraction.  Energy against the golden;
    iteration count and <S^2> against the oracle (itself pinned to the same rows in tests/test_oracle_golden.py)."""
    z = [SYMBOL_TO_Z[s.lower()] for s in case["symbols"]]
    frag = fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=case["multiplicity"])
    st = methods.ScfSettings(basis_set=case["basis"], functional=case["functional"], grid_level=case["grid_level"],
                             energy_tol=1e-10, density_tol=1e-7, guess="gwh", max_iter=case["maxiter"])
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    assert r.scf_status == methods.SCF_CONVERGED
    assert abs(r.energy.scf - case["expected_energy"]) < 1e-9
    mol = oracle_mol(case["basis"], frag)
    o = so.run_uhf(mol, int(frag.nelec), case["multiplicity"], case["maxiter"], 1e-10, 1e-7,
                   xc=xc_oracle.XCOracle(mol, case["functional"], case["grid_level"]))
    assert abs(r.energy.scf - o.energy) < 1e-9
    assert r.scf_iterations == o.iterations
    assert abs(r.s_squared - o.s_squared) < 1e-6


def test_unrestricted_runs_on_the_direct_path():
    """UHF and UKS with the integrals formed on the fly (eri_mode = direct: what fragments above 116 functions take), the
    integrals formed once per spin density: the manifest's OH UHF and CH3 UKS-PBE energies to 1e-9."""
    oh = [c for c in _uhf_cases() if c["basis"] == "cc-pvdz" and c["symbols"] == ["O", "H"]][0]
    ch3 = [c for c in _uks_cases() if c["symbols"][0] == "C" and c["functional"] == "pbe"][0]
    for case in (oh, ch3):
        z = [SYMBOL_TO_Z[s_.lower()] for s_ in case["symbols"]]
        frag = fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=case["multiplicity"])
        st = methods.ScfSettings(basis_set=case["basis"], functional=case["functional"], eri_mode="direct", energy_tol=1e-10,
                                 density_tol=1e-7, guess="gwh", max_iter=case["maxiter"])
        r = methods.run_hip_scf(st, frag)
        assert not r.has_error, r.error_message
        assert abs(r.energy.scf - case["expected_energy"]) < 1e-9, (case["name"], r.energy.scf)


def test_unrestricted_density_fitting_matches_oracle():
    """UHF and UKS with density-fitted J / K -- the mode of the cuEST path's own unrestricted SCF (run_uks_scf,
    mqc_cuest_scf.f90:637-1009): J from the total density, K_s from the occupied orbitals of each spin.  OH (UHF) and
    CH3 (UKS-PBE) against the oracle with the same auxiliary set, iteration counts included; a closed-shell molecule run
    unrestricted lands on the restricted density-fitted energy."""
    oh = [c for c in _uhf_cases() if c["basis"] == "cc-pvdz" and c["symbols"] == ["O", "H"]][0]
    ch3 = [c for c in _uks_cases() if c["symbols"][0] == "C" and c["functional"] == "pbe"][0]
    for case, fn in ((oh, ""), (ch3, "pbe")):
        z = [SYMBOL_TO_Z[s_.lower()] for s_ in case["symbols"]]
        frag = fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=2)
        st = methods.ScfSettings(basis_set="cc-pvdz", functional=fn, density_fitting=True, aux_basis_set=AUX, energy_tol=1e-10,
                                 density_tol=1e-8, guess="gwh")
        r = methods.run_hip_scf(st, frag)
        assert not r.has_error, r.error_message

        def oracle(frag=frag, fn=fn):
            mol = oracle_mol("cc-pvdz", frag)
            return scf_record(so.run_uhf(mol, int(frag.nelec), 2, 100, 1e-10, 1e-8, aux=oracle_mol(AUX, frag),
                                         xc=xc_oracle.XCOracle(mol, fn, 3) if fn else None))
        o = recorded_oracle("unrestricted_df", frag, "cc-pvdz|%s|df:%s|grid3|1e-10|1e-8|gwh" % (fn, AUX), oracle)
        assert abs(r.energy.scf - o["energy"]) < 1e-8, (fn, r.energy.scf, o["energy"])
        assert r.scf_iterations == o["iterations"]
        assert abs(r.energy.scf - case["expected_energy"]) < 5e-4          # a fitting error away from the exact-ERI golden
    w = fragment_bohr(*WATER)
    ru = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", density_fitting=True, aux_basis_set=AUX, energy_tol=1e-10,
                                                 density_tol=1e-8, guess="gwh", unrestricted=True), w)
    rr = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", density_fitting=True, aux_basis_set=AUX, energy_tol=1e-10,
                                                 density_tol=1e-8, guess="gwh"), w)
    assert not ru.has_error and abs(ru.energy.scf - rr.energy.scf) < 1e-9 and abs(ru.s_squared) < 1e-8


def test_uks_closed_shell_limit_and_batch():
    """A closed-shell molecule run unrestricted lands on the restricted Kohn-Sham energy (the reference's own guard
    against a wrong spin stride, mqc_libcint_xc.F90:944-946); a batch of OH radicals in one call against the oracle."""
    w = fragment_bohr(*WATER)
    for fn in ("svwn", "b3lyp"):
        ru = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", functional=fn, energy_tol=1e-10, density_tol=1e-8,
                                                     guess="gwh", unrestricted=True), w)
        rr = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", functional=fn, energy_tol=1e-10, density_tol=1e-8,
                                                     guess="gwh"), w)
        assert not ru.has_error, ru.error_message
        assert abs(ru.energy.scf - rr.energy.scf) < 1e-9, (fn, ru.energy.scf, rr.energy.scf)
        assert abs(ru.s_squared) < 1e-8
    # methyl radicals (non-degenerate SOMO; OH's half-filled pi pair orients against the grid and two correct codes land
    # 5e-7 apart, as the reference's own manifest notes) at three C-H scalings in one call
    ch3 = [c for c in _uks_cases() if c["symbols"][0] == "C"][0]
    z = [SYMBOL_TO_Z[s_.lower()] for s_ in ch3["symbols"]]
    x0 = np.array(ch3["xyz_angstrom"]) * ANGSTROM_TO_BOHR
    frags = [fragment_bohr(z, x0[0] + (x0 - x0[0]) * k, multiplicity=2) for k in (0.96, 1.0, 1.05)]
    st = methods.ScfSettings(basis_set="cc-pvdz", functional="pbe", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    for f, r in zip(frags, methods.run_hip_scf_batch(st, frags)):
        assert not r.has_error, r.error_message
        def oracle(f=f):
            mol = oracle_mol("cc-pvdz", f)
            return scf_record(so.run_uhf(mol, 9, 2, 100, 1e-10, 1e-8, xc=xc_oracle.XCOracle(mol, "pbe", 3)))
        o = recorded_oracle("uks_pbe_methyl_batch", f, "cc-pvdz|pbe|uks|grid3|1e-10|1e-8|gwh", oracle)
        assert abs(r.energy.scf - o["energy"]) < 1e-8, (r.energy.scf, o["energy"])
        assert r.scf_iterations == o["iterations"]


# ---- superposed-atom guesses ------------------------------------------------------------------------------------
def test_sad_guess_reaches_the_same_states_in_fewer_cycles():
    """guess = sad / sac (mqc_libcint_atomic_guess.f90): free atoms solved by the engine itself (UHF at Hund's
    multiplicity, cached), spherically averaged blocks on the diagonal, the Hartree-Fock Fock matrix of that density
    as the starting Fock.  The converged energy does not depend on the guess: RHF, B3LYP, DF-RHF, direct SCF and UHF
    runs started from SAD land on the GWH-started energies (themselves pinned to the goldens) and need no more cycles."""
    w = fragment_bohr(*WATER)
    base = dict(basis_set="cc-pvdz", energy_tol=1e-10, density_tol=1e-8)
    for extra in (dict(), dict(functional="b3lyp"), dict(density_fitting=True, aux_basis_set="mqc-even-tempered-jkfit"),
                  dict(eri_mode="direct"), dict(unrestricted=True)):
        g = methods.run_hip_scf(methods.ScfSettings(guess="gwh", **base, **extra), w)
        a = methods.run_hip_scf(methods.ScfSettings(guess="sad", **base, **extra), w)
        assert not g.has_error and not a.has_error, (extra, g.error_message, a.error_message)
        assert abs(g.energy.scf - a.energy.scf) < 2e-9, (extra, g.energy.scf, a.energy.scf)
        assert a.scf_iterations <= g.scf_iterations, (extra, a.scf_iterations, g.scf_iterations)
    c = methods.run_hip_scf(methods.ScfSettings(guess="sac", **base), w)
    g = methods.run_hip_scf(methods.ScfSettings(guess="gwh", **base), w)
    assert not c.has_error and abs(c.energy.scf - g.energy.scf) < 2e-9
    # the manifest's UHF OH row was produced from the reference's auto = SAD start
    oh = [c_ for c_ in _uhf_cases() if c_["basis"] == "cc-pvdz" and c_["symbols"] == ["O", "H"]][0]
    z = [SYMBOL_TO_Z[s_.lower()] for s_ in oh["symbols"]]
    frag = fragment_bohr(z, np.array(oh["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=2)
    r = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", guess="sad", energy_tol=1e-10, density_tol=1e-7), frag)
    assert not r.has_error, r.error_message
    assert abs(r.energy.scf - oh["expected_energy"]) < 1e-9
    # unrestricted + sac is refused, not replaced
    r = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", guess="sac"), frag)
    assert r.has_error and "sac" in r.error_message.lower()


def test_sad_guess_in_a_batch_with_a_ghost_atom():
    """A batch of dimers started from SAD (one guess per topology, broadcast to the fragments) against the oracle's
    energies; a ghosted monomer (functions of the partner, no nuclei or electrons there) takes no density on the ghosts."""
    rng = np.random.default_rng(11)
    frags = []
    for k in range(6):
        frags.append(fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([water_at(rng, np.zeros(3)), water_at(rng, np.array([5.2 + 0.3 * k, 0.4, -0.2]))])))
    st = methods.ScfSettings(basis_set="cc-pvdz", guess="sad", energy_tol=1e-10, density_tol=1e-8)
    res = methods.run_hip_scf_batch(st, frags)
    for f, r in zip(frags[:2], res[:2]):
        assert not r.has_error, r.error_message
        o = so.run_rhf(oracle_mol("cc-pvdz", f), int(f.nelec), 100, 1e-10, 1e-8)
        assert abs(r.energy.scf - o.energy) < 1e-8
    gh = fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([water_at(rng, np.zeros(3)), water_at(rng, np.array([5.5, 0.0, 0.0]))]),
                       ghost=[False] * 3 + [True] * 3)
    a = methods.run_hip_scf(st, gh)
    g = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-10, density_tol=1e-8), gh)
    assert not a.has_error and not g.has_error, (a.error_message, g.error_message)
    assert abs(a.energy.scf - g.energy.scf) < 2e-9


def test_fragments_above_140_functions_run_from_global_memory():
    """n_ao = 147 ((H2O)21 in STO-3G: 63 atoms): the Fock matrix no longer fits the CU's LDS, the Jacobi rotations run
    on a copy in global memory (L2), the two-electron part on the direct path, the quadrature deals its output tiles to
    blockIdx.z groups.  RHF and B3LYP against the oracle, iteration counts included."""
    rng = np.random.default_rng(7)
    xs = [water_at(rng, np.array([5.6 * i, 5.6 * j, 5.6 * k])) for i in range(3) for j in range(3) for k in range(3)][:21]
    frag = fragment_bohr([8, 1, 1] * 21, np.vstack(xs))
    mol = oracle_mol("sto-3g", frag)
    assert mol.nao == 147
    eri_box = []

    def oracle(fn):
        if not eri_box:
            eri_box.append(so.eri4(mol))
        xc = xc_oracle.XCOracle(mol, fn, 1) if fn else None       # the coarsest grid: the oracle's quadrature sets the time
        return scf_record(so.run_rhf(mol, int(frag.nelec), 100, 1e-9, 1e-7, xc=xc, eri=eri_box[0]))

    for fn in ("", "b3lyp"):
        st = methods.ScfSettings(basis_set="sto-3g", functional=fn, grid_level=1, energy_tol=1e-9, density_tol=1e-7, guess="gwh")
        r = methods.run_hip_scf(st, frag)
        assert not r.has_error, r.error_message
        o = recorded_oracle("h2o21_sto3g", frag, "sto-3g|%s|grid1|1e-9|1e-7|gwh" % fn, lambda: oracle(fn))      # ~2 min live
        assert abs(r.energy.scf - o["energy"]) < 2e-8, (fn, r.energy.scf, o["energy"])
        assert r.scf_iterations == o["iterations"]
    r = methods.run_hip_scf(methods.ScfSettings(basis_set="sto-3g", density_fitting=True, aux_basis_set="mqc-even-tempered-jkfit"), frag)
    assert r.has_error and "140" in r.error_message


def test_refusals_match_reference_behaviour():
    st = methods.ScfSettings(basis_set="sto-3g", functional="tpss")
    r = methods.run_hip_scf(st, fragment_bohr(*WATER), want_gradient=True)
    assert r.has_error and not r.has_energy              # meta-GGA gradients: refused, not replaced by something else
    st = methods.ScfSettings(basis_set="sto-3g", max_iter=2, energy_tol=1e-12, density_tol=1e-12)
    r = methods.run_hip_scf(st, fragment_bohr(*WATER))
    assert r.scf_status == methods.SCF_NOT_CONVERGED and r.has_error     # not converged is an error ...
    st.allow_crap_scf = True
    r = methods.run_hip_scf(st, fragment_bohr(*WATER))
    assert r.scf_status == methods.SCF_NOT_CONVERGED and not r.has_error and r.has_energy   # ... unless allowed
    oh = methods.run_hip_scf(methods.ScfSettings(basis_set="sto-3g"), fragment_bohr([8, 1], [[0, 0, 0], [0, 0, 1.8]]))
    assert oh.has_error      # nine electrons cannot be paired into a singlet (parities disagree): a validation error


def _jk_from_packed(M, D):
    """numpy contraction of the engine's own packed tensor (size-independent property check)."""
    n = D.shape[0]
    idx = [(i, j) for i in range(n) for j in range(i + 1)]
    ii = np.array([p[0] for p in idx]); jj = np.array([p[1] for p in idx])
    full = np.zeros((n, n, n, n))
    full[ii[:, None], jj[:, None], ii[None, :], jj[None, :]] = M
    full[jj[:, None], ii[:, None], ii[None, :], jj[None, :]] = M
    full[ii[:, None], jj[:, None], jj[None, :], ii[None, :]] = M
    full[jj[:, None], ii[:, None], jj[None, :], ii[None, :]] = M
    return so.build_jk_incore(full, D)


def test_jk_incore_larger_fragments_all_kernel_variants():
    """n = 72 (water trimer, two k-chunks, D/K in global memory) against the packed tensor itself."""
    rng = np.random.default_rng(9)
    xyz = np.vstack([water_at(rng, c) for c in ([0, 0, 0], [5.6, 0.3, 0.2], [0.1, 5.9, -0.4])])
    frag = fragment_bohr([8, 1, 1] * 3, xyz)
    M = stages.eri_packed("cc-pvdz", frag)
    n = 72
    D = synthetic_density(n)
    J, K = stages.jk_incore("cc-pvdz", frag, D)
    Jr, Kr = _jk_from_packed(M, D)
    assert np.max(np.abs(J - Jr)) < 1e-10
    assert np.max(np.abs(K - Kr)) < 1e-10
    # linearity in the density: J[a D1 + b D2] = a J[D1] + b J[D2]
    D2 = synthetic_density(n)[::-1, ::-1].copy()
    J2, K2 = stages.jk_incore("cc-pvdz", frag, D2)
    J3, K3 = stages.jk_incore("cc-pvdz", frag, 0.3 * D - 1.7 * D2)
    assert np.max(np.abs(J3 - (0.3 * J - 1.7 * J2))) < 1e-10
    assert np.max(np.abs(K3 - (0.3 * K - 1.7 * K2))) < 1e-10


# ---- Kohn-Sham: XC quadrature kernel -----------------------------------------------------------
import json
import os

from metalquicha_amd.basis import ANGSTROM_TO_BOHR, SYMBOL_TO_Z
from oracle import xc_oracle

_CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest_subset.json")))["cases"]
_KS = [c for c in _CASES if c["method"] == "dft" and c["functional"] in xc_oracle.RESTRICTED_FUNCTIONALS and "grid 3" in c["name"]
       and not c["unrestricted"] and not c["density_fitting"]]
_RHF = [c for c in _CASES if c["method"] == "hf" and "mbe_level" not in c and c["driver"] == "Energy" and not c["unrestricted"]
        and not c["density_fitting"] and "*" not in c["basis"]]        # Pople star sets are Cartesian: refused, tested below


def _frag(c):
    z = [SYMBOL_TO_Z[s.lower()] for s in c["symbols"]]
    return fragment_bohr(z, np.array(c["xyz_angstrom"]) * ANGSTROM_TO_BOHR)


@pytest.mark.parametrize("case", _RHF, ids=[c["name"] for c in _RHF])
def test_manifest_rhf_goldens(case):
    st = methods.ScfSettings(basis_set=case["basis"], energy_tol=1e-10, density_tol=1e-7, guess="gwh", max_iter=case["maxiter"])
    r = methods.run_hip_scf(st, _frag(case))
    assert not r.has_error, r.error_message
    assert abs(r.energy.scf - case["expected_energy"]) < 1e-9


@pytest.mark.parametrize("case", _KS, ids=[c["name"] for c in _KS])
def test_manifest_kohn_sham_goldens(case):
    """SVWN / PBE / B3LYP / PBE0 on H2O and PBE on CH4, cc-pVDZ, grid level 3: reference tolerance 1e-9."""
    st = methods.ScfSettings(basis_set=case["basis"], functional=case["functional"], grid_level=case["grid_level"],
                             energy_tol=1e-10, density_tol=1e-7, guess="gwh", max_iter=case["maxiter"])
    frag = _frag(case)
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    assert r.scf_status == methods.SCF_CONVERGED
    assert abs(r.energy.scf - case["expected_energy"]) < 1e-8          # north-star bound
    mol = oracle_mol(case["basis"], frag)
    o = so.run_rhf(mol, int(frag.nelec), case["maxiter"], 1e-10, 1e-7, xc=xc_oracle.XCOracle(mol, case["functional"], case["grid_level"]))
    assert abs(r.energy.scf - o.energy) < 1e-9
    assert r.scf_iterations == o.iterations


def test_b3lyp_water_dimer_batch_matches_oracle():
    rng = np.random.default_rng(77)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [5.4, 0.4, -0.2])]
    frags = [fragment_bohr([8, 1, 1], ws[0]), fragment_bohr([8, 1, 1], ws[1]), fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack(ws))]
    st = methods.ScfSettings(basis_set="cc-pvdz", functional="b3lyp", energy_tol=1e-9, density_tol=1e-7, guess="gwh")
    res = methods.run_hip_scf_batch(st, frags)
    for f, r in zip(frags, res):
        assert not r.has_error, r.error_message
        def oracle(f=f):
            mol = oracle_mol("cc-pvdz", f)
            return scf_record(so.run_rhf(mol, int(f.nelec), 100, 1e-9, 1e-7, xc=xc_oracle.XCOracle(mol, "b3lyp", 3)))
        o = recorded_oracle("b3lyp_water_batch", f, "cc-pvdz|b3lyp|grid3|1e-9|1e-7|gwh", oracle)
        assert abs(r.energy.scf - o["energy"]) < 1e-8, (r.energy.scf, o["energy"])


_XC_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from metalquicha_amd import methods
from tests.helpers import fragment_bohr, water_at
rng = np.random.default_rng(4)
ws = [water_at(rng, c) for c in ([0, 0, 0], [5.5, 0.2, -0.3], [0.3, 5.6, 0.4])]
frags = [fragment_bohr([8, 1, 1], ws[0]), fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack(ws[:2])), fragment_bohr([8, 1, 1] * 3, np.vstack(ws))]
out = {}
for fn in ("svwn", "b3lyp"):
    st = methods.ScfSettings(basis_set="cc-pvdz", functional=fn, energy_tol=1e-9, density_tol=1e-7, guess="gwh")
    res = methods.run_hip_scf_batch(st, frags)
    out[fn] = {"e": [r.energy.scf for r in res], "it": [r.scf_iterations for r in res], "err": [r.error_message for r in res if r.has_error]}
print(json.dumps(out))
"""


def test_xc_kernel_variants_agree():
    """The split quadrature (default for n <= 64: density / functional / potential kernels; n = 24, 48 here, the tile
    kernel at n = 72) against the single tile kernel (MQC_HIP_XC_SPLIT=0), the pipelined kernel (MQC_HIP_XC_PIPE=1) and
    a run without the radial-value cache (MQC_HIP_XC_RADIAL_CACHE=0: exponentials in the kernel, tile kernel): same
    iteration counts, energies within the summation-order noise of the quadrature."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", _XC_CHILD, root], env=env, check=True, capture_output=True, text=True,
                             timeout=900).stdout.strip().splitlines()[-1]
        return json.loads(out)

    new = run({})
    others = [run({"MQC_HIP_XC_SPLIT": "0"}), run({"MQC_HIP_XC_SPLIT": "0", "MQC_HIP_XC_PIPE": "1"}), run({"MQC_HIP_XC_RADIAL_CACHE": "0"})]
    for fn in ("svwn", "b3lyp"):
        assert not new[fn]["err"], new[fn]["err"]
        for old in others:
            assert not old[fn]["err"], old[fn]["err"]
            assert new[fn]["it"] == old[fn]["it"]
            assert np.max(np.abs(np.array(new[fn]["e"]) - np.array(old[fn]["e"]))) < 1e-9


def test_unknown_functional_is_refused():
    r = methods.run_hip_scf(methods.ScfSettings(basis_set="sto-3g", functional="m06-l"), fragment_bohr(*WATER))
    assert r.has_error and "not available" in r.error_message


# ---- density fitting ----------------------------------------------------------------------------
AUX = "mqc-even-tempered-jkfit"    # this repo's own auxiliary set: DF parity is HIP vs oracle only


def _oracle_df(frag, basis, functional=""):
    mol = oracle_mol(basis, frag)
    aux = oracle_mol(AUX, frag)
    xc = xc_oracle.XCOracle(mol, functional, 3) if functional else None
    return so.run_rhf(mol, int(frag.nelec), 100, 1e-10, 1e-8, aux=aux, xc=xc)


@pytest.mark.parametrize("basis", ["sto-3g", "cc-pvdz"])
def test_df_rhf_matches_oracle(basis):
    frag = fragment_bohr(*WATER)
    st = methods.ScfSettings(basis_set=basis, density_fitting=True, aux_basis_set=AUX, energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    o = _oracle_df(frag, basis)
    assert abs(r.energy.scf - o.energy) < 1e-9
    assert r.scf_iterations == o.iterations
    exact = methods.run_hip_scf(methods.ScfSettings(basis_set=basis, energy_tol=1e-10, density_tol=1e-8, guess="gwh"), frag)
    assert 1e-7 < abs(r.energy.scf - exact.energy.scf) < 1e-3        # a fitting error, not zero and not large


def test_df_b3lyp_batch_matches_oracle():
    rng = np.random.default_rng(12)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [5.5, -0.3, 0.4])]
    frags = [fragment_bohr([8, 1, 1], ws[0]), fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack(ws))]
    st = methods.ScfSettings(basis_set="cc-pvdz", functional="b3lyp", density_fitting=True, aux_basis_set=AUX,
                             energy_tol=1e-9, density_tol=1e-7, guess="gwh")
    res = methods.run_hip_scf_batch(st, frags)
    for f, r in zip(frags, res):
        assert not r.has_error, r.error_message
        o = recorded_oracle("df_b3lyp_water_batch", f, "cc-pvdz|b3lyp|df:%s|grid3|1e-9|1e-7|gwh" % AUX,
                            lambda f=f: scf_record(_oracle_df(f, "cc-pvdz", "b3lyp")))
        assert abs(r.energy.scf - o["energy"]) < 1e-8, (r.energy.scf, o["energy"])


def test_df_rhf_with_the_orbital_basis_as_auxiliary_basis_matches_pinned_oracle():
    """The reference's own DF decks fit with the orbital basis (RHF H2O 6-31G*/6-31G*, CH4 6-31G**/6-31G**); the oracle
    reproduces those goldens (tests/test_oracle_golden.py, Cartesian d).  6-31G has no d shell, so the same kind of
    fit runs on the HIP engine: orbital = auxiliary = 6-31G, a poor fit with a large error, HIP == oracle to 1e-9."""
    from tests.helpers import W1_ANGSTROM
    frag = fragment_bohr([8, 1, 1], np.array(W1_ANGSTROM) * ANGSTROM_TO_BOHR)
    st = methods.ScfSettings(basis_set="6-31g", density_fitting=True, aux_basis_set="6-31g", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    mol = oracle_mol("6-31g", frag)
    o = so.run_rhf(mol, 10, 100, 1e-10, 1e-8, aux=oracle_mol("6-31g", frag))
    assert abs(r.energy.scf - o.energy) < 1e-9, (r.energy.scf, o.energy)
    assert r.scf_iterations == o.iterations
    exact = methods.run_hip_scf(methods.ScfSettings(basis_set="6-31g", energy_tol=1e-10, density_tol=1e-8, guess="gwh"), frag)
    assert abs(exact.energy.scf - (-75.984779843967)) < 1e-9            # manifest row RHF H2O 6-31g
    assert abs(r.energy.scf - exact.energy.scf) > 1e-3                   # a fit in a basis this small is far off


def test_df_near_singular_metric_takes_the_reference_eigen_cut(tmp_path, monkeypatch):
    """An auxiliary basis with a (numerically) duplicated shell: the metric (P|Q) has an eigenvalue far below the
    reference's 1e-10 cut (metric_inverse_sqrt, mqc_libcint_integrals.F90:992-1038).  The engine's Cholesky fit flags
    the fragment (pivot / ||L^-1||_F bound), the Jacobi eigen path builds J^{-1/2} over the kept eigenvalues, and the
    energy equals the oracle's (numpy eigh with the same cut) -- and the energy without the duplicate."""
    import json, os
    from metalquicha_amd import basis as basis_mod
    src = json.load(open(basis_mod.find_basis_file(AUX)))
    dup = json.loads(json.dumps(src))
    sh = json.loads(json.dumps(dup["elements"]["8"]["electron_shells"][0]))
    sh["exponents"] = ["%.16E" % (float(e) * (1.0 + 1.0e-9)) for e in sh["exponents"]]
    dup["elements"]["8"]["electron_shells"].append(sh)
    with open(tmp_path / "mqc-dup-jkfit.json", "w") as f:
        json.dump(dup, f)
    monkeypatch.setenv("MQC_BASIS_PATH", str(tmp_path))
    frag = fragment_bohr(*WATER)
    kw = dict(basis_set="cc-pvdz", density_fitting=True, energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    r = methods.run_hip_scf(methods.ScfSettings(aux_basis_set="mqc-dup-jkfit", **kw), frag)
    assert not r.has_error, r.error_message
    mol = oracle_mol("cc-pvdz", frag)
    aux = oracle_mol("mqc-dup-jkfit", frag)
    w = np.linalg.eigvalsh(so.eri2c(aux))
    assert w[0] < 1e-10 < w[1] * 1e3            # exactly the duplicated direction is (numerically) null
    o = so.run_rhf(mol, 10, 100, 1e-10, 1e-8, aux=aux)
    assert abs(r.energy.scf - o.energy) < 1e-8, (r.energy.scf, o.energy)
    clean = methods.run_hip_scf(methods.ScfSettings(aux_basis_set=AUX, **kw), frag)
    assert abs(r.energy.scf - clean.energy.scf) < 1e-7      # the duplicate adds nothing to the fit


def test_cartesian_basis_is_refused_like_the_cuest_driver_does():
    """6-31G* carries gto_cartesian d shells: load_basis refuses it on the GPU path (mqc_cuest_driver.f90:331-341)."""
    r = methods.run_hip_scf(methods.ScfSettings(basis_set="6-31g*"), fragment_bohr(*WATER))
    assert r.has_error and "Cartesian" in r.error_message


def test_df_without_aux_basis_file_is_an_error():
    st = methods.ScfSettings(basis_set="sto-3g", density_fitting=True, aux_basis_set="def2-universal-jkfit")
    r = methods.run_hip_scf(st, fragment_bohr(*WATER))
    assert r.has_error and "not found" in r.error_message


# ---- direct (integral-recomputing) Fock build ---------------------------------------------------
def test_direct_scf_matches_incore_and_oracle():
    """The reference's own direct-vs-in-core check (test_mqc_libcint_direct.f90): same energy, same
    iteration count; screening at 1e-11 changes nothing at the 1e-9 level."""
    frag = fragment_bohr(*WATER)
    kw = dict(basis_set="cc-pvdz", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    inc = methods.run_hip_scf(methods.ScfSettings(eri_mode="incore", **kw), frag)
    dire = methods.run_hip_scf(methods.ScfSettings(eri_mode="direct", **kw), frag)
    assert not dire.has_error, dire.error_message
    assert abs(dire.energy.scf - inc.energy.scf) < 1e-9
    assert dire.scf_iterations == inc.scf_iterations
    assert abs(dire.energy.scf - (-76.0220988827)) < 1e-9


def test_direct_scf_with_a_pure_functional_skips_exchange_and_matches_incore():
    """A functional without exact exchange on the direct path: the digest kernels leave after their Coulomb updates
    (exx = 0: K is never formed).  Same energy and iteration count as the in-core path, which forms K alongside J and
    weights it with zero; the hybrid next to it still digests the exchange."""
    frag = fragment_bohr(*WATER)
    for fun in ("pbe", "b3lyp"):
        kw = dict(basis_set="cc-pvdz", functional=fun, energy_tol=1e-10, density_tol=1e-8, guess="gwh")
        inc = methods.run_hip_scf(methods.ScfSettings(eri_mode="incore", **kw), frag)
        dire = methods.run_hip_scf(methods.ScfSettings(eri_mode="direct", **kw), frag)
        assert not inc.has_error and not dire.has_error, (inc.error_message, dire.error_message)
        assert abs(dire.energy.scf - inc.energy.scf) < 1e-9, (fun, dire.energy.scf, inc.energy.scf)
        assert dire.scf_iterations == inc.scf_iterations


def test_direct_scf_batch_with_far_apart_dimer_screens_and_agrees():
    rng = np.random.default_rng(31)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [14.0, 0.5, -0.3])]
    frags = [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack(ws)), fragment_bohr([8, 1, 1], ws[0])]
    kw = dict(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh")
    a = methods.run_hip_scf_batch(methods.ScfSettings(eri_mode="incore", **kw), frags)
    b = methods.run_hip_scf_batch(methods.ScfSettings(eri_mode="direct", **kw), frags)
    for x, y in zip(a, b):
        assert not y.has_error, y.error_message
        assert abs(x.energy.scf - y.energy.scf) < 1e-8


def test_auto_mode_goes_direct_for_large_fragments():
    """Benzene/cc-pVDZ has n_ao = 114 (in-core still fits); a water pentamer (n_ao = 120) does not and must
    run through the direct path under eri_mode auto; checked against DF within the fitting error."""
    rng = np.random.default_rng(8)
    cs = [[0, 0, 0], [5.6, 0.2, 0.1], [0.2, 5.7, -0.3], [5.5, 5.8, 0.2], [2.8, 2.9, 4.9]]
    xyz = np.vstack([water_at(rng, c) for c in cs])
    frag = fragment_bohr([8, 1, 1] * 5, xyz)
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-8, density_tol=1e-6, guess="gwh")
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    d = methods.run_hip_scf(methods.ScfSettings(basis_set="cc-pvdz", density_fitting=True, aux_basis_set=AUX,
                                                energy_tol=1e-8, density_tol=1e-6, guess="gwh"), frag)
    assert not d.has_error, d.error_message
    assert abs(r.energy.scf - d.energy.scf) < 5e-4
    assert r.scf_iterations == d.scf_iterations


# ---- chunked / pipelined batches ------------------------------------------------------------------
_CHUNK_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from metalquicha_amd import methods
from tests.helpers import fragment_bohr, water_at
rng = np.random.default_rng(77)
ws = [water_at(rng, [5.7 * i, 0.4 * (i % 3), -0.3 * (i % 2)]) for i in range(7)]
frags = [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([ws[i], ws[j]])) for i in range(7) for j in range(i + 1, 7)]
kw = dict(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh")
kw.update(json.loads(sys.argv[2]))
res = methods.run_hip_scf_batch(methods.ScfSettings(**kw), frags)
print(json.dumps({"e": [r.energy.scf for r in res], "it": [r.scf_iterations for r in res],
                  "err": [r.error_message for r in res if r.has_error]}))
"""


@pytest.mark.parametrize("extra", [{}, {"functional": "pbe"}, {"density_fitting": True, "aux_basis_set": "mqc-even-tempered-jkfit"},
                                   {"eri_mode": "direct"}], ids=["rhf", "pbe", "df", "direct"])
def test_chunked_two_slot_batches_match_the_single_chunk_result(extra):
    """21 dimers as ONE chunk vs cut into 5 chunks alternating between the two stream/pool slots
    (MQC_HIP_PIPELINE_CHUNKS; the same code path an over-budget batch takes).  Fragments do not interact,
    so iteration counts are identical and energies agree to the summation-order noise of the atomic
    accumulations (1e-11 Eh)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", _CHUNK_CHILD, root, json.dumps(extra)], env=env, check=True,
                             capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
        return json.loads(out)

    one = run({"MQC_HIP_PIPELINE_CHUNKS": "1"})
    five = run({"MQC_HIP_PIPELINE_CHUNKS": "5", "MQC_HIP_PIPELINE_MIN_FRAGMENTS": "2"})
    assert not one["err"] and not five["err"], (one["err"], five["err"])
    assert one["it"] == five["it"]
    assert np.max(np.abs(np.array(one["e"]) - np.array(five["e"]))) < 1e-11


# ---- block sharing between fragments with identical atoms ------------------------------------------
_SHARE_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from metalquicha_amd import methods
from tests.helpers import fragment_bohr, water_at
rng = np.random.default_rng(11)
ws = [water_at(rng, [5.9 * (i % 4), 5.9 * (i // 4), -0.4 * (i % 2)]) for i in range(13)]
# 78 dimers built from 13 monomers (every monomer geometry repeats bit for bit in 12 dimers: the engine shares atom
# sets that repeat at least 6 times) + 3 odd ones out
frags = [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([ws[i], ws[j]])) for i in range(13) for j in range(i + 1, 13)]
frags += [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([water_at(rng, [0, 29.0 + k, 0]), water_at(rng, [5.5, 29.0 + k, 1.0])])) for k in range(3)]
kw = dict(basis_set=sys.argv[2], energy_tol=1e-9, density_tol=1e-7, guess="gwh", eri_mode="incore")
res = methods.run_hip_scf_batch(methods.ScfSettings(**kw), frags)
print(json.dumps({"e": [r.energy.scf for r in res], "it": [r.scf_iterations for r in res],
                  "err": [r.error_message for r in res if r.has_error]}))
"""


@pytest.mark.parametrize("basis", ["sto-3g", "cc-pvdz"])
def test_block_sharing_gives_the_unshared_result(basis):
    """A batch whose dimers repeat their monomers bit for bit: the engine forms the intra-monomer integral
    blocks once per distinct monomer and copies them (kern_eri.hip, block sharing).  Against the same batch
    with MQC_HIP_NO_BLOCK_SHARING=1 and MQC_HIP_NO_TWIN_BLOCKS=1 (every fragment forms everything with the
    segmented kernels): same iteration counts, energies within summation-order noise."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", _SHARE_CHILD, root, basis], env=env, check=True,
                             capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
        return json.loads(out)

    shared = run({})
    plain = run({"MQC_HIP_NO_BLOCK_SHARING": "1", "MQC_HIP_NO_TWIN_BLOCKS": "1"})
    assert not shared["err"] and not plain["err"], (shared["err"], plain["err"])
    assert shared["it"] == plain["it"]
    assert np.max(np.abs(np.array(shared["e"]) - np.array(plain["e"]))) < 1e-10


# ---- the large-batch J/K variant (12 waves) -----------------------------------------------
def test_large_batch_jk_variant_matches_small_batch_variant_and_oracle():
    """66 water dimers in ONE batch (>= 64 fragments) store the tensor as triangular blocks of row pairs and take
    jk_tri_kernel (every stored element in both of its roles, K = Kh + Kh^T, diagonal elements pre-halved; with
    MQC_HIP_ERI_TRI=0: the tuned square kernel); the same dimers in three batches of 22 keep the square tensor and
    the generic 4-wave kernel.  Same iteration counts, energies within 1e-10; two of them are also checked against
    the CPU oracle.  The unscreened build exercises the zero fill the block padding needs, the screened one the
    copy of shared blocks in the triangular layout."""
    rng = np.random.default_rng(123)
    ws = [water_at(rng, [5.8 * (i % 4), 5.8 * (i // 4), 0.6 * (i % 3)]) for i in range(12)]
    pairs = [(i, j) for i in range(12) for j in range(i + 1, 12)]          # 66 dimers
    frags = [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([ws[i], ws[j]])) for i, j in pairs]
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh", eri_mode="incore")
    big = methods.run_hip_scf_batch(st, frags)
    small = []
    for k in range(0, 66, 22):
        small += methods.run_hip_scf_batch(st, frags[k:k + 22])
    for a, b in zip(big, small):
        assert not a.has_error and not b.has_error, (a.error_message, b.error_message)
        assert a.scf_iterations == b.scf_iterations
        assert abs(a.energy.scf - b.energy.scf) < 1e-10
    for k in (0, 37):
        o = so.run_rhf(oracle_mol("cc-pvdz", frags[k]), 20, 100, 1e-9, 1e-7)
        assert abs(big[k].energy.scf - o.energy) < 1e-8
        assert big[k].scf_iterations == o.iterations
    # the screened build bench.py uses (Schwarz 1e-12, together with twin blocks and block sharing): every
    # fragment within 1e-10 Eh of the unscreened result, same iteration counts
    scr = methods.run_hip_scf_batch(methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh",
                                                        eri_mode="incore", schwarz_tol=1e-12), frags)
    for a, b in zip(big, scr):
        assert not b.has_error, b.error_message
        assert a.scf_iterations == b.scf_iterations
        assert abs(a.energy.scf - b.energy.scf) < 1e-10


def test_triangular_tensor_at_another_fragment_size():
    """n = 40 (four hydrogen molecules, cc-pVDZ: npair = 820, blocks of 874 doubles): the triangular J/K path away from
    the water dimers' 48 functions.  70 fragments in one batch (triangular blocks, several workgroups per fragment with
    atomic flushes) against the same fragments in batches of 14 (square tensor): iteration counts equal, energies
    within 1e-10; one against the oracle."""
    rng = np.random.default_rng(2024)
    frags = []
    for _ in range(70):
        centres = np.array([[0.0, 0.0, 0.0], [4.2, 0.3, -0.2], [0.4, 4.4, 0.3], [4.0, 4.1, 0.5]]) + rng.normal(scale=0.25, size=(4, 3))
        xyz = []
        for c in centres:
            d = rng.normal(size=3); d *= 0.70 / np.linalg.norm(d)
            xyz += [c - d, c + d]
        frags.append(fragment_bohr([1] * 8, np.array(xyz)))
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh", eri_mode="incore", schwarz_tol=1e-12)
    big = methods.run_hip_scf_batch(st, frags)
    small = []
    for k in range(0, 70, 14):
        small += methods.run_hip_scf_batch(st, frags[k:k + 14])
    for a, b in zip(big, small):
        assert not a.has_error and not b.has_error, (a.error_message, b.error_message)
        assert a.scf_iterations == b.scf_iterations
        assert abs(a.energy.scf - b.energy.scf) < 1e-10
    o = so.run_rhf(oracle_mol("cc-pvdz", frags[5]), 8, 100, 1e-9, 1e-7)          # eight electrons
    assert abs(big[5].energy.scf - o.energy) < 1e-8 and big[5].scf_iterations == o.iterations


def test_small_batch_paths_match_the_large_batch_paths():
    """Batches of at most 16 fragments form their twin (ss|ss) / (ps|ss) entries one wave per (entry, fragment) with the
    lanes over the bra primitive pairs, fan the one-electron classes over four streams and skip the Schwarz bounds;
    larger ones run lane = fragment throughout.  The same four dimers alone and inside a batch of twenty: same
    iteration counts, energies within 1e-10; one of them against the oracle."""
    rng = np.random.default_rng(77)
    ws = [water_at(rng, [5.6 * (i % 3), 5.9 * (i // 3), 0.4 * (i % 2)]) for i in range(7)]
    pairs = [(i, j) for i in range(7) for j in range(i + 1, 7)][:20]
    frags = [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([ws[i], ws[j]])) for i, j in pairs]
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh", eri_mode="incore")
    big = methods.run_hip_scf_batch(st, frags)
    small = methods.run_hip_scf_batch(st, frags[:4])
    one = methods.run_hip_scf(st, frags[2])
    for a, b in zip(big[:4], small):
        assert not a.has_error and not b.has_error, (a.error_message, b.error_message)
        assert a.scf_iterations == b.scf_iterations
        assert abs(a.energy.scf - b.energy.scf) < 1e-10
    assert abs(one.energy.scf - big[2].energy.scf) < 1e-10 and one.scf_iterations == big[2].scf_iterations
    o = so.run_rhf(oracle_mol("cc-pvdz", frags[2]), 20, 100, 1e-9, 1e-7)
    assert abs(one.energy.scf - o.energy) < 1e-8 and one.scf_iterations == o.iterations


# ---- concurrent topology groups ---------------------------------------------------------------------
_LANES_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from metalquicha_amd import methods
from tests.helpers import fragment_bohr, water_at
rng = np.random.default_rng(5)
ws = [water_at(rng, [5.8 * i, 0.3 * (i % 2), 0.5 * (i % 3)]) for i in range(6)]
frags = [fragment_bohr([8, 1, 1], w) for w in ws]
frags += [fragment_bohr([8, 1, 1, 8, 1, 1], np.vstack([ws[i], ws[j]])) for i in range(6) for j in range(i + 1, 6)]
frags += [fragment_bohr([8, 1, 1] * 3, np.vstack([ws[0], ws[1], ws[2]]))]
kw = dict(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh")
kw.update(json.loads(sys.argv[2]))
res = methods.run_hip_scf_batch(methods.ScfSettings(**kw), frags)
print(json.dumps({"e": [r.energy.scf for r in res], "it": [r.scf_iterations for r in res],
                  "err": [r.error_message for r in res if r.has_error]}))
"""


@pytest.mark.parametrize("extra", [{}, {"functional": "b3lyp"}], ids=["rhf", "b3lyp"])
def test_concurrent_topology_groups_match_sequential_execution(extra):
    """Monomers, dimers and a trimer in one call = three topology groups; with MQC_HIP_CONCURRENT_GROUPS (default)
    two host threads drive two of them at a time on separate slots.  Same iteration counts and energies as the
    one-after-the-other execution."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", _LANES_CHILD, root, json.dumps(extra)], env=env, check=True,
                             capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
        return json.loads(out)

    both = run({})
    seq = run({"MQC_HIP_CONCURRENT_GROUPS": "0"})
    assert not both["err"] and not seq["err"], (both["err"], seq["err"])
    assert both["it"] == seq["it"]
    assert np.max(np.abs(np.array(both["e"]) - np.array(seq["e"]))) < 1e-10
