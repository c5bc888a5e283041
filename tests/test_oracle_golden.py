"""The oracle against every known-answer number the reference holds for the exact-ERI RHF path
(SURVEY.md section 8c).  CPU only; this is what pins the oracle before it is used as a checker."""
import numpy as np
import pytest

from oracle import scf_oracle as so
from tests.helpers import fragment_bohr, oracle_mol, W1_ANGSTROM
from metalquicha_amd.basis import ANGSTROM_TO_BOHR

WATER = ([8, 1, 1], [[0.0, 0.0, -0.1364652], [0.0, 1.4304924, 1.0826636], [0.0, -1.4304924, 1.0826636]])


def test_h2_sto3g_check_rhf():
    # validation/check_rhf.f90:79-83: -1.1167143251, E_nuc = 1/1.4, nao = 2
    mol = oracle_mol("sto-3g-check_rhf", fragment_bohr([1, 1], [[0, 0, 0], [0, 0, 1.4]]))
    assert mol.nao == 2
    r = so.run_rhf(mol, 2, 100, 1e-10, 1e-8)
    assert r.converged
    assert abs(r.nuclear - 1.0 / 1.4) < 1e-12
    assert abs(r.energy - (-1.1167143251)) < 1e-9


def test_h2_overlap_check_libcint():
    # validation/check_libcint.f90:33,123-127: S12(H2/STO-3G, R = 1.4) = 0.6593 +- 5e-5, diagonal 1
    S, _, _ = so.int1e(oracle_mol("sto-3g-check_rhf", fragment_bohr([1, 1], [[0, 0, 0], [0, 0, 1.4]])))
    assert abs(S[0, 1] - 0.6593) < 5e-5
    assert np.allclose(np.diag(S), 1.0, atol=1e-12)


def test_water_sto3g_check_rhf_and_diis_invariance():
    # validation/check_rhf.f90:102-143: nao = 7, -74.9658162796, DIIS == no DIIS to 1e-9, fewer iterations
    mol = oracle_mol("sto-3g-check_rhf", fragment_bohr(*WATER))
    assert mol.nao == 7
    r = so.run_rhf(mol, 10, 100, 1e-10, 1e-8)
    plain = so.run_rhf(mol, 10, 200, 1e-10, 1e-8, diis_vectors=0)
    assert r.converged and plain.converged
    assert abs(r.energy - (-74.9658162796)) < 1e-9
    assert abs(r.energy - plain.energy) < 1e-9
    assert r.iterations < plain.iterations


def test_water_ccpvdz_check_df_exact():
    # validation/check_df.f90:55-57: exact-ERI RHF/cc-pVDZ -76.0220988827 (1e-9); also pins the
    # cc-pVDZ H and O tables typed into metalquicha_amd/basis_data/make_basis_json.py
    mol = oracle_mol("cc-pvdz", fragment_bohr(*WATER))
    assert mol.nao == 24
    r = so.run_rhf(mol, 10, 200, 1e-11, 1e-9)
    assert r.converged
    assert abs(r.energy - (-76.0220988827)) < 1e-9


def test_water_sto3g_backends_example():
    # python/examples/backends.py:28-54: H2O STO-3G (8-decimal geometry) RHF -74.962005687948,
    # and the 4 A-separated dimer -149.922856255009 (the number an MBE(2) of it must reproduce)
    w = np.array([[0.0, 0.0, 0.10077199], [0.0, 0.77250895, -0.46780200], [0.0, -0.77250895, -0.46780200]])
    # (PySCF's built-in STO-3G = the 8-digit table of check_rhf.f90; the 10-digit BSE table moves
    #  this energy by 2.4e-8, which is how the two are told apart)
    bas = "sto-3g-check_rhf"
    r = so.run_rhf(oracle_mol(bas, fragment_bohr([8, 1, 1], w * ANGSTROM_TO_BOHR)), 10, 100, 1e-10, 1e-8)
    assert abs(r.energy - (-74.962005687948)) < 1e-9
    d = np.vstack([w, w + np.array([4.0, 0.0, 0.0])])
    r2 = so.run_rhf(oracle_mol(bas, fragment_bohr([8, 1, 1, 8, 1, 1], d * ANGSTROM_TO_BOHR)), 20, 100, 1e-10, 1e-8)
    assert abs(r2.energy - (-149.922856255009)) < 2e-9


def test_h2_manifest_first_case():
    # validation/validation_tests_cpu.json first case: H2 STO-3G r = 0.7414 A, -1.1166843872
    xyz = np.array([[0, 0, 0], [0, 0, 0.7414]]) * ANGSTROM_TO_BOHR
    r = so.run_rhf(oracle_mol("sto-3g", fragment_bohr([1, 1], xyz)), 2, 100, 1e-12, 1e-8)
    assert abs(r.energy - (-1.1166843872)) < 1e-9


def test_diis_reference_properties():
    # test/test_mqc_diis.f90:21-26: coefficients sum to 1, ring eviction, cached overlap = direct
    rng = np.random.default_rng(0)
    d = so.Diis(4, 9, 6)
    focks, errs = [], []
    for k in range(7):
        f, e = rng.normal(size=9), rng.normal(size=6) * 10.0 ** (-k)
        focks.append(f); errs.append(e)
        d.push(f, e)
        c = d.coefficients()
        if k == 0:
            assert c is None
            continue
        n = d.n_stored
        assert n == min(k + 1, 4)
        assert abs(np.sum(c[:n]) - 1.0) < 1e-10
        kept = errs[-n:]
        for a in range(n):
            for b in range(n):
                direct = float(np.dot(kept[a], kept[b]))
                cached = d.overlap[d.slot_of_age(a + 1) - 1, d.slot_of_age(b + 1) - 1]
                assert abs(direct - cached) < 1e-15 * max(1.0, abs(direct))
        ex, ok = d.extrapolate(focks[-1].copy())
        assert ok
        assert np.allclose(ex, sum(c[i] * focks[len(focks) - n + i] for i in range(n)), atol=1e-12)


def test_schwarz_bounds_bound_the_tensor():
    mol = oracle_mol("sto-3g", fragment_bohr(*WATER))
    eri = so.eri4(mol)
    q = so.schwarz(mol)
    off = list(mol.sh_aoff) + [mol.nao]
    for a in range(mol.nshell):
        for b in range(mol.nshell):
            for c in range(mol.nshell):
                for d in range(mol.nshell):
                    blk = eri[off[a]:off[a + 1], off[b]:off[b + 1], off[c]:off[c + 1], off[d]:off[d + 1]]
                    assert np.max(np.abs(blk)) <= q[a, b] * q[c, d] * (1 + 1e-10) + 1e-14


# ---- manifest fixtures (tests/golden/manifest_subset.json, harvested by harvest_manifest.py) ----
import json
import os

from metalquicha_amd.basis import SYMBOL_TO_Z
from oracle import grid_oracle, xc_oracle

_CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest_subset.json")))["cases"]
_RHF = [c for c in _CASES if c["method"] == "hf" and "mbe_level" not in c and c["driver"] == "Energy"
        and not c["unrestricted"] and not c["density_fitting"]]
_DF = [c for c in _CASES if c["method"] == "hf" and c["density_fitting"] and c["driver"] == "Energy" and not c["unrestricted"]]
_KS = [c for c in _CASES if c["method"] == "dft" and c["functional"] in xc_oracle.RESTRICTED_FUNCTIONALS and "grid 3" in c["name"]
       and not c["unrestricted"] and not c["density_fitting"]]


def _mol(c):
    z = [SYMBOL_TO_Z[s.lower()] for s in c["symbols"]]
    frag = fragment_bohr(z, np.array(c["xyz_angstrom"]) * ANGSTROM_TO_BOHR)
    return frag, oracle_mol(c["basis"], frag)


@pytest.mark.parametrize("case", _RHF, ids=[c["name"] for c in _RHF])
def test_manifest_rhf(case):
    """PySCF-referenced RHF energies (tolerance 1e-9): pins STO-3G, cc-pVDZ, 6-31G, 6-31G* / 6-31G** (Cartesian d),
    def2-SVP and def2-TZVP (f shells) for H, C, N, O as typed into metalquicha_amd/basis_data."""
    frag, mol = _mol(case)
    r = so.run_rhf(mol, int(frag.nelec), case["maxiter"], 1e-10, 1e-7)
    assert r.converged
    assert abs(r.energy - case["expected_energy"]) < 1e-9


@pytest.mark.parametrize("case", _DF, ids=[c["name"] for c in _DF])
def test_manifest_density_fitted_rhf(case):
    """The reference's density-fitted RHF goldens whose auxiliary basis this repository can supply: H2O 6-31G*/6-31G*
    -76.188111755038 and CH4 6-31G**/6-31G** -40.381603512964 (validation_tests_cpu.json, df-hf/).  This is what pins
    the oracle's DF leg -- three-centre (mu nu|P) and two-centre (P|Q) integrals, the J^{-1/2} fit with its 1e-10
    eigenvalue cut (mqc_libcint_integrals.F90:913-1038), DF-J and DF-K (mqc_libcint_rhf.f90:1576-1646) -- to numbers
    the reference produced; the HIP DF path is then compared with this oracle."""
    frag, mol = _mol(case)
    z = [SYMBOL_TO_Z[s.lower()] for s in case["symbols"]]
    aux = oracle_mol(case["aux_basis"], fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR))
    assert mol.cart and aux.cart          # Pople star sets: Cartesian d on both sides (they must share the angular form)
    r = so.run_rhf(mol, int(frag.nelec), case["maxiter"], 1e-10, 1e-7, aux=aux)
    assert r.converged
    assert abs(r.energy - case["expected_energy"]) < 1e-9


def test_grid_matches_reference_counts():
    """SURVEY.md section 9 (measured with the reference's own grid module): the check_rhf water at
    level 3 has 33 698 points and sum(w) = 17 026.5356."""
    pts, w, own = grid_oracle.build_grid(*WATER, 3)
    assert len(w) == 33698
    assert abs(w.sum() - 17026.5356) < 1e-3
    # a normalised Gaussian integrates to 1 on the atomic grid (test_mqc_dft_grid.f90:24-32 style)
    pts1, w1, _ = grid_oracle.build_grid([8], [[0.0, 0.0, 0.0]], 3)
    a = 1.3
    val = float(np.dot(w1, (a / np.pi) ** 1.5 * np.exp(-a * np.sum(pts1 ** 2, axis=1))))
    assert abs(val - 1.0) < 1e-10


@pytest.mark.parametrize("case", _KS, ids=[c["name"] for c in _KS])
def test_manifest_kohn_sham(case):
    """KS goldens (SVWN, PBE, B3LYP, PBE0, TPSS on H2O; PBE on CH4), tolerance 1e-9: pins the grid and the
    restated libxc functionals (lda_x, vwn5, vwn_rpa, b88, lyp, pbe x/c with pw_mod, tpss x/c with tau and the
    v_tau term of the potential)."""
    frag, mol = _mol(case)
    xc = xc_oracle.XCOracle(mol, case["functional"], case["grid_level"])
    r = so.run_rhf(mol, int(frag.nelec), case["maxiter"], 1e-10, 1e-7, xc=xc)
    assert r.converged
    assert abs(r.energy - case["expected_energy"]) < 1e-9
    assert abs(xc.n_electrons - frag.nelec) < 1e-4


_UKS = [c for c in _CASES if c["method"] == "dft" and c["functional"] in xc_oracle.RESTRICTED_FUNCTIONALS and c["unrestricted"]
        and c["driver"] == "Energy" and not c["density_fitting"]]


@pytest.mark.parametrize("case", _UKS, ids=[c["name"] for c in _UKS])
def test_manifest_unrestricted_kohn_sham(case):
    """UKS goldens (CH3 doublet SVWN / PBE / B3LYP / TPSS, O2 triplet PBE), tolerance 1e-9: pins the spin-polarised forms --
    TPSS exchange by spin scaling with tau_s, revPKZB correlation with C(zeta, xi) and the per-spin max[] terms --
    lda_x spin scaling, VWN5 with the spin stiffness, VWN-RPA's f(zeta) interpolation, B88 per spin, LYP for two
    spin densities, PBE exchange scaling and PBE correlation with phi(zeta) over polarised pw_mod -- and
    xc_add_potential_uks' cross-spin gradient term."""
    z = [SYMBOL_TO_Z[s.lower()] for s in case["symbols"]]
    frag = fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=case["multiplicity"])
    mol = oracle_mol(case["basis"], frag)
    xc = xc_oracle.XCOracle(mol, case["functional"], case["grid_level"])
    r = so.run_uhf(mol, int(frag.nelec), case["multiplicity"], case["maxiter"], 1e-10, 1e-7, xc=xc)
    assert r.converged
    assert abs(r.energy - case["expected_energy"]) < 1e-9
    assert abs(xc.n_electrons - frag.nelec) < 1e-4


def test_polarised_functionals_reduce_to_the_restricted_forms():
    """rho_a = rho_b: same energy density, v_rho_a = v_rho_b = v_rho, (v_aa + v_ab + v_bb) / 4 = v_sigma."""
    rng = np.random.default_rng(1)
    rho = 10.0 ** rng.uniform(-4, 1.0, 100)
    sig = (rho ** (4.0 / 3.0) * 10.0 ** rng.uniform(-2, 1, size=100)) ** 2
    for name in ("svwn", "pbe", "blyp", "b3lyp", "pbe0"):
        f, vr, vs = xc_oracle.eval_functional(name, rho, sig)
        fp, (vra, vrb, vaa, vab, vbb) = xc_oracle.eval_functional_pol(name, rho / 2, rho / 2, sig / 4, sig / 4, sig / 4)
        assert np.max(np.abs(f - fp) / np.abs(f)) < 1e-13
        assert np.max(np.abs(vr - vra) / np.abs(vr)) < 1e-12 and np.max(np.abs(vra - vrb)) == 0.0
        if name != "svwn":
            assert np.max(np.abs(vs - (vaa + vab + vbb) / 4.0) / (np.abs(vs) + 1e-300)) < 1e-10


def test_functional_derivatives_by_finite_differences():
    rng = np.random.default_rng(4)
    rho = 10.0 ** rng.uniform(-4, 1.5, size=200)
    sigma = (rho ** (4.0 / 3.0) * 10.0 ** rng.uniform(-2, 1, size=200)) ** 2
    for name in ("svwn", "pbe", "blyp", "b3lyp", "pbe0"):
        f, vr, vs = xc_oracle.eval_functional(name, rho, sigma)
        h = 1e-6
        fp, _, _ = xc_oracle.eval_functional(name, rho * (1 + h), sigma)
        fm, _, _ = xc_oracle.eval_functional(name, rho * (1 - h), sigma)
        assert np.max(np.abs((fp - fm) / (2 * h * rho) - vr) / (np.abs(vr) + 1e-8)) < 1e-6
        fp, _, _ = xc_oracle.eval_functional(name, rho, sigma * (1 + h))
        fm, _, _ = xc_oracle.eval_functional(name, rho, sigma * (1 - h))
        if name == "svwn":
            assert np.all(vs == 0.0)
            continue
        sig = np.abs(vs) * sigma > 1e-5 * np.abs(f)          # where the sigma derivative is resolvable by differences
        assert sig.sum() > 50
        assert np.max((np.abs((fp - fm) / (2 * h * sigma) - vs) / np.abs(vs))[sig]) < 1e-4


def test_meta_gga_derivatives_by_finite_differences():
    rng = np.random.default_rng(5)
    rho = 10.0 ** rng.uniform(-4, 1.5, size=300)
    tau_w_ratio = rng.uniform(0.02, 0.98, size=300)               # z = tau_W / tau stays inside (0, 1)
    sigma = (rho ** (4.0 / 3.0) * 10.0 ** rng.uniform(-2, 1, size=300)) ** 2
    tau = sigma / (8.0 * rho) / tau_w_ratio
    f, vr, vs, vt = xc_oracle.eval_functional_mgga("tpss", rho, sigma, tau)
    h = 1e-6
    for k, (x, v) in enumerate(((rho, vr), (sigma, vs), (tau, vt))):
        args_p = [rho, sigma, tau]; args_m = [rho, sigma, tau]
        args_p[k] = x * (1 + h); args_m[k] = x * (1 - h)
        fp = xc_oracle.eval_functional_mgga("tpss", *args_p)[0]
        fm = xc_oracle.eval_functional_mgga("tpss", *args_m)[0]
        num = (fp - fm) / (2 * h * x)
        big = np.abs(v) * x > 1e-5 * np.abs(f)
        assert big.sum() > 100
        assert np.max((np.abs(num - v) / np.abs(v))[big]) < 1e-4
    # the uniform-gas limit: no gradient, tau = tau_unif -> TPSS exchange is LDA exchange with F_x(p = 0, z = 0, alpha = 1) = 1
    r = np.array([0.3, 2.0]); tu = 0.3 * (3 * np.pi ** 2) ** (2 / 3) * r ** (5 / 3)
    V = [xc_oracle.DualN.var(x, i, 3) for i, x in enumerate((r, np.full(2, 1e-40), tu))]
    assert np.allclose(xc_oracle.mgga_x_tpss(*V).v, -0.75 * (3 / np.pi) ** (1 / 3) * r ** (4 / 3), rtol=1e-12)


def test_polarised_meta_gga_reduces_to_the_restricted_form():
    rng = np.random.default_rng(6)
    rho = 10.0 ** rng.uniform(-3, 1, size=100)
    sigma = (rho ** (4.0 / 3.0) * 10.0 ** rng.uniform(-2, 1, size=100)) ** 2
    tau = sigma / (8.0 * rho) / rng.uniform(0.05, 0.95, size=100)
    f0, vr, vs, vt = xc_oracle.eval_functional_mgga("tpss", rho, sigma, tau)
    f1, dv = xc_oracle.eval_functional_mgga_pol("tpss", rho / 2, rho / 2, sigma / 4, sigma / 4, sigma / 4, tau / 2, tau / 2)
    assert np.max(np.abs(f0 - f1) / np.abs(f0)) < 1e-13
    assert np.max(np.abs(dv[0] - vr)) < 1e-12 and np.max(np.abs(dv[1] - vr)) < 1e-12          # d/d rho_a = d/d rho at zeta = 0
    assert np.max(np.abs(dv[5] - vt)) < 1e-10 and np.max(np.abs(dv[6] - vt)) < 1e-10
    # sigma = saa + 2 sab + sbb: the restricted v_sigma is the common value of v_aa + v_bb + v_ab weighted by d sigma_xy / d sigma
    assert np.max(np.abs((dv[2] + dv[3] + dv[4]) / 4.0 - vs) / (np.abs(vs) + 1e-12)) < 1e-9


def test_recorded_fixtures_spot_check_against_the_live_oracle():
    """tests/golden/oracle_fixtures.json holds OUTPUTS of this repository's oracle for the slow GPU parity cases, keyed by
    their inputs -- a change to oracle/*.py does not change the key.  This test recomputes the cheapest recorded cases live
    (the three B3LYP / def2-TZVP water monomers of the GMBE-2 workload test: grid, functional, f shells, SCF driver all in
    play, 2-4 s each) and compares them with the file: an oracle that drifted shows up here, on the CPU, before the GPU
    suite compares the engine with stale numbers."""
    import json, os
    from tests import helpers, workload_cases as wc
    fixtures = json.load(open(helpers._FIXTURE_PATH))
    system = wc.gmbe_system()
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    checked = 0
    for m in system.monomers:
        atoms = [int(a) for a in m]
        f = helpers.fragment_bohr(z[atoms], xyz[atoms])
        key = helpers._fixture_key("gmbe2_b3lyp_def2tzvp", f, wc.GMBE_KEY)
        assert key in fixtures, "fixture missing: run tests/golden/record_oracle_fixtures.py gmbe"
        live = wc.gmbe_fragment_oracle(f)
        assert abs(live["energy"] - fixtures[key]["energy"]) < 1e-10, (live, fixtures[key])
        assert live["iterations"] == fixtures[key]["iterations"]
        checked += 1
    assert checked == 3
