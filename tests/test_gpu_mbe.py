"""GPU parity tests of the CALLER side of the hot path: fragment lists through
mbe.run_mbe -> one engine batch call -> compute_mbe, against the reference's MBE(2) golden and against
per-fragment oracle energies; plus BASELINE.json configs[1] (benzene) and a ghost-atom fragment.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from metalquicha_amd import mbe, methods
from metalquicha_amd.basis import ANGSTROM_TO_BOHR, SYMBOL_TO_Z
from oracle import scf_oracle as so
from oracle import xc_oracle
from tests.helpers import fragment_bohr, oracle_mol, water_at, recorded_oracle, scf_record

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest_subset.json")))["cases"]
AUX = "mqc-even-tempered-jkfit"


def test_mbe2_water_dimer_reference_golden():
    """validation_tests_cpu.json 'MBE(2) RHF gradient (H2O)2 cc-pvdz (CPU)': MBE(2) total -152.056781852646
    (tolerance 1e-9 in the reference suite), through run_mbe + compute_mbe + compute_mbe_coefficients."""
    case = [c for c in _CASES if c.get("mbe_level") == 2][0]
    system = mbe.system_from_xyz(case["symbols"], np.array(case["xyz_angstrom"]), case["fragments"])
    st = methods.ScfSettings(basis_set=case["basis"], energy_tol=1e-10, density_tol=1e-7, guess="gwh", max_iter=case["maxiter"])
    run = mbe.run_mbe(system, st, level=case["mbe_level"])
    assert not run.errors, run.errors
    assert sorted(len(t) for t in run.terms) == [1, 1, 2]
    total, by_order, _ = mbe.compute_mbe(run.terms, run.energies)
    assert abs(total - case["expected_energy"]) < 1e-9
    coef = mbe.compute_mbe_coefficients(run.terms)
    assert abs(float(np.dot(coef, run.energies)) - case["expected_energy"]) < 1e-9
    # the two-body correction is the interaction energy of the dimer: small and negative here
    assert -0.02 < by_order[2] < 0.0


_CLUSTER_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from metalquicha_amd import mbe, methods
system = mbe.water_cluster(2)
terms = mbe.generate_mbe_term_list(system, 2)
st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-8, density_tol=1e-6, schwarz_tol=float(sys.argv[2]))
out = {}
for world in (1, 2):
    e = np.zeros(len(terms)); it = np.zeros(len(terms))
    for rank in range(world):
        run = mbe.run_mbe(system, st, level=2, rank=rank, world=world, terms=terms)
        assert not run.errors, run.errors
        e += run.energies; it += run.iterations
    out["w%d" % world] = {"e": e.tolist(), "it": it.tolist()}
print(json.dumps(out))
"""


def _cluster_child(env_extra, schwarz):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", _CLUSTER_CHILD, ROOT, repr(schwarz)], env=env, check=True, capture_output=True,
                         text=True, timeout=900).stdout.strip().splitlines()[-1]
    return json.loads(out)


def test_mbe2_bench_shaped_cluster_matches_per_fragment_oracle():
    """(H2O)8 (the bench generator at side 2), MBE-2 RHF/cc-pVDZ with the bench settings: 8 monomers + 28 dimers in ONE
    batch call (two topology groups on two lanes, compactness re-ordering of the results, shared intra-monomer
    blocks, Schwarz 1e-12).  Every fragment energy within 1e-8 Eh of the oracle's, same iteration counts; the
    assembled MBE-2 total within 1e-8; block sharing off / round-robin over two ranks give the same numbers."""
    system = mbe.water_cluster(2)
    terms = mbe.generate_mbe_term_list(system, 2)
    assert len(terms) == 36
    got = _cluster_child({}, 1e-12)
    e = np.array(got["w1"]["e"]); it = np.array(got["w1"]["it"])
    eo = np.zeros(len(terms)); ito = np.zeros(len(terms))
    for k, t in enumerate(terms):
        frag = mbe.build_fragment(system, t)
        o = so.run_rhf(oracle_mol("cc-pvdz", frag), int(frag.nelec), 100, 1e-8, 1e-6)
        eo[k] = o.energy; ito[k] = o.iterations
    assert np.max(np.abs(e - eo)) < 1e-8, np.max(np.abs(e - eo))
    assert np.array_equal(it, ito)
    tot, _, _ = mbe.compute_mbe(terms, e)
    tot_o, _, _ = mbe.compute_mbe(terms, eo)
    assert abs(tot - tot_o) < 1e-8
    # round-robin partition over two ranks, energies summed as the all-reduce would: identical fragments, identical numbers
    assert np.max(np.abs(np.array(got["w2"]["e"]) - e)) < 1e-10
    assert np.array_equal(np.array(got["w2"]["it"]), it)
    plain = _cluster_child({"MQC_HIP_NO_BLOCK_SHARING": "1", "MQC_HIP_NO_TWIN_BLOCKS": "1", "MQC_HIP_CONCURRENT_GROUPS": "0"}, 0.0)
    assert np.max(np.abs(np.array(plain["w1"]["e"]) - e)) < 1e-9
    assert np.array_equal(np.array(plain["w1"]["it"]), it)


def _benzene():
    rcc, rch = 1.397, 1.084          # D6h, in the xy-plane
    sym, xyz = [], []
    for k in range(6):
        a = np.pi / 3 * k
        sym.append("C"); xyz.append([rcc * np.cos(a), rcc * np.sin(a), 0.0])
    for k in range(6):
        a = np.pi / 3 * k
        sym.append("H"); xyz.append([(rcc + rch) * np.cos(a), (rcc + rch) * np.sin(a), 0.0])
    return methods.PhysicalFragment.from_angstrom(sym, xyz)


def test_benzene_b3lyp_ccpvdz_df_matches_oracle():
    """BASELINE.json configs[1]: single benzene B3LYP/cc-pVDZ, density-fitted J/K, grid level 3, GWH guess,
    e_tol 1e-10 / d_tol 1e-8 (n_ao = 114, 21 occupied): energy within 1e-8 Eh of the oracle, same iteration count."""
    frag = _benzene()
    st = methods.ScfSettings(basis_set="cc-pvdz", functional="b3lyp", density_fitting=True, aux_basis_set=AUX,
                             energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    assert r.scf_status == methods.SCF_CONVERGED
    def oracle():
        mol = oracle_mol("cc-pvdz", frag); aux = oracle_mol(AUX, frag)
        assert mol.nao == 114
        return scf_record(so.run_rhf(mol, 42, 100, 1e-10, 1e-8, aux=aux, xc=xc_oracle.XCOracle(mol, "b3lyp", 3)))
    o = recorded_oracle("benzene_b3lyp_df", frag, "cc-pvdz|b3lyp|df:%s|grid3|1e-10|1e-8|gwh" % AUX, oracle)      # ~75 s live
    assert abs(r.energy.scf - o["energy"]) < 1e-8, (r.energy.scf, o["energy"])
    assert r.scf_iterations == o["iterations"]
    assert -232.4 < r.energy.scf < -232.1          # B3LYP/cc-pVDZ benzene, literature -232.26


@pytest.mark.parametrize("functional", ["", "b3lyp"], ids=["rhf", "b3lyp"])
def test_ghost_atom_fragment_matches_oracle(functional):
    """A counterpoise-style fragment: water A with the basis functions (and grid points) of a ghosted water B --
    ghost atoms carry basis functions, no nuclear charge and no electrons (mqc_libcint_integrals.F90:481-488,
    mqc_physical_fragment.f90:73-83)."""
    rng = np.random.default_rng(42)
    xyz = np.vstack([water_at(rng, [0, 0, 0]), water_at(rng, [5.4, 0.3, -0.2])])
    ghost = np.array([0, 0, 0, 1, 1, 1], dtype=bool)
    frag = fragment_bohr([8, 1, 1, 8, 1, 1], xyz, ghost=ghost)
    assert frag.nelec == 10
    st = methods.ScfSettings(basis_set="cc-pvdz", functional=functional, energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    mol = oracle_mol("cc-pvdz", frag)
    assert mol.nao == 48 and float(np.sum(mol.z)) == 10.0
    xc = xc_oracle.XCOracle(mol, functional, 3) if functional else None
    o = so.run_rhf(mol, 10, 100, 1e-10, 1e-8, xc=xc)
    assert abs(r.energy.scf - o.energy) < 1e-8, (r.energy.scf, o.energy)
    assert r.scf_iterations == o.iterations
    # the ghost basis lowers the monomer energy (basis-set superposition), by less than a few mEh
    real = fragment_bohr([8, 1, 1], xyz[:3])
    r0 = methods.run_hip_scf(st, real)
    assert 0.0 < r0.energy.scf - r.energy.scf < 5e-3


# ---- analytic gradients -----------------------------------------------------------
_GRAD = [c for c in _CASES if c.get("expected_gradient") and c["method"] == "hf" and not c["density_fitting"]
         and "*" not in c["basis"] and "mbe_level" not in c]


@pytest.mark.parametrize("case", _GRAD, ids=[c["name"] for c in _GRAD])
def test_manifest_hf_gradient_goldens(case):
    """validation_tests_cpu.json gradient/ rows the engine can run (RHF H2O STO-3G, RHF NH3 cc-pVDZ, UHF CH3 cc-pVDZ):
    energy to 1e-9, every gradient component to the manifest's own tolerance (1e-8), translational invariance."""
    z = [SYMBOL_TO_Z[s.lower()] for s in case["symbols"]]
    frag = fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=case["multiplicity"])
    st = methods.ScfSettings(basis_set=case["basis"], energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_gradient(frag)
    assert not r.has_error, r.error_message
    assert r.has_gradient and r.gradient.shape == (3, len(z))
    assert abs(r.energy.scf - case["expected_energy"]) < 1e-9
    ref = np.array(case["expected_gradient"]).T
    assert np.max(np.abs(r.gradient - ref)) < max(case["gradient_tolerance"], 2e-8), np.max(np.abs(r.gradient - ref))
    assert np.max(np.abs(r.gradient.sum(axis=1))) < 1e-9


from oracle import xc_oracle as _xco

_KSGRAD = [c for c in _CASES if c.get("expected_gradient") and c["method"] == "dft" and not c["density_fitting"]
           and c["functional"] in _xco.FUNCTIONALS and "*" not in c["basis"] and "mbe_level" not in c]


@pytest.mark.parametrize("case", _KSGRAD, ids=[c["name"] for c in _KSGRAD])
def test_manifest_kohn_sham_gradient_goldens(case):
    """validation_tests_cpu.json gradient/ rows for Kohn-Sham (SVWN, PBE, B3LYP on H2O cc-pVDZ; unrestricted PBE on
    CH3): the exchange-correlation gradient with moving functions (second derivatives of the basis for the GGAs),
    moving grid points and the Becke / Treutler partition derivatives, as xc_gradient builds it
    (mqc_libcint_gradient.f90:331-549).  Energy to 1e-9, components to the manifest's tolerance."""
    z = [SYMBOL_TO_Z[s.lower()] for s in case["symbols"]]
    frag = fragment_bohr(z, np.array(case["xyz_angstrom"]) * ANGSTROM_TO_BOHR, multiplicity=case["multiplicity"])
    st = methods.ScfSettings(basis_set=case["basis"], functional=case["functional"], grid_level=case["grid_level"],
                             energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_gradient(frag)
    assert not r.has_error, r.error_message
    assert r.has_gradient and r.gradient.shape == (3, len(z))
    assert abs(r.energy.scf - case["expected_energy"]) < 1e-9
    ref = np.array(case["expected_gradient"]).T
    assert np.max(np.abs(r.gradient - ref)) < max(case["gradient_tolerance"], 5e-8), (np.max(np.abs(r.gradient - ref)), r.gradient, ref)
    assert np.max(np.abs(r.gradient.sum(axis=1))) < 1e-7


def test_kohn_sham_gradient_matches_finite_differences():
    """Central differences of the engine's own B3LYP energy (the grid moves with the atoms, so the differences carry the
    full grid response) on a bent, asymmetric water."""
    xyz = np.array([[0.03, -0.02, -0.13], [0.10, 1.45, 1.05], [-0.05, -1.38, 1.12]])
    st = methods.ScfSettings(basis_set="cc-pvdz", functional="b3lyp", energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_gradient(fragment_bohr([8, 1, 1], xyz))
    assert not r.has_error, r.error_message
    h = 2e-3
    frags = []
    for a in range(3):
        for c in range(3):
            for sgn in (+1, -1):
                x = xyz.copy(); x[a, c] += sgn * h
                frags.append(fragment_bohr([8, 1, 1], x))
    e = [q.energy.scf for q in methods.run_hip_scf_batch(st, frags)]
    fd = np.zeros((3, 3))
    k = 0
    for a in range(3):
        for c in range(3):
            fd[c, a] = (e[k] - e[k + 1]) / (2 * h); k += 2
    assert np.max(np.abs(r.gradient - fd)) < 5e-6, (np.max(np.abs(r.gradient - fd)), r.gradient, fd)


def test_gradient_matches_finite_differences_of_the_oracle_energy():
    """check_gradient's procedure (validation/check_gradient.f90: central differences, bound 3.5e-8 Eh/a0) with the
    oracle as the energy function: a bent, asymmetric water in cc-pVDZ (d shells, every class of the gradient kernel)."""
    xyz = np.array([[0.03, -0.02, -0.13], [0.10, 1.45, 1.05], [-0.05, -1.38, 1.12]])
    frag = fragment_bohr([8, 1, 1], xyz)
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_gradient(frag)
    assert not r.has_error, r.error_message
    h = 1e-3
    fd = np.zeros((3, 3))
    for a in range(3):
        for c in range(3):
            e = []
            for sgn in (+1, -1):
                x = xyz.copy(); x[a, c] += sgn * h
                e.append(so.run_rhf(oracle_mol("cc-pvdz", fragment_bohr([8, 1, 1], x)), 10, 200, 1e-12, 1e-10).energy)
            fd[c, a] = (e[0] - e[1]) / (2 * h)
    assert np.max(np.abs(r.gradient - fd)) < 2e-6, np.max(np.abs(r.gradient - fd))       # O(h^2) error of the differences
    assert np.max(np.abs(r.gradient.sum(axis=1))) < 1e-9


def test_mbe2_water_dimer_gradient_golden():
    """'MBE(2) RHF gradient (H2O)2 cc-pvdz (CPU)': the 6 x 3 MBE(2) gradient (tolerance 1e-7) assembled from the
    fragment gradients with the MBE coefficients, fragments run as one batch."""
    case = [c for c in _CASES if c.get("mbe_level") == 2 and c.get("expected_gradient")][0]
    system = mbe.system_from_xyz(case["symbols"], np.array(case["xyz_angstrom"]), case["fragments"])
    st = methods.ScfSettings(basis_set=case["basis"], energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
    run = mbe.run_mbe(system, st, level=2, want_gradient=True)
    assert not run.errors, run.errors
    total, _, _ = mbe.compute_mbe(run.terms, run.energies)
    assert abs(total - case["expected_energy"]) < 1e-9
    ref = np.array(case["expected_gradient"])
    assert run.gradient is not None and run.gradient.shape == ref.shape
    assert np.max(np.abs(run.gradient - ref)) < case["gradient_tolerance"], np.max(np.abs(run.gradient - ref))


# ---- GMBE caller mirror ---------------------------------------------------------------------------------------------
def test_gmbe_over_overlapping_water_fragments():
    """GMBE(1) over two overlapping base fragments of a water trimer, E = E(w0 w1) + E(w1 w2) - E(w1), against the
    three SCFs run one by one; GMBE(2) over NON-overlapping monomers reproduces the MBE(2) total and gradient."""
    from metalquicha_amd import gmbe
    rng = np.random.default_rng(5)
    ws = [water_at(rng, c) for c in ([0, 0, 0], [5.4, 0.3, 0.1], [10.9, -0.2, 0.4])]
    xyz = np.vstack(ws)
    z = np.array([8, 1, 1] * 3, dtype=np.int32)
    st = methods.ScfSettings(basis_set="cc-pvdz", energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    ov = mbe.FragmentedSystem(z, xyz.T.copy(), [np.arange(0, 6), np.arange(3, 9)])
    run = gmbe.run_gmbe(ov, st, level=1)
    assert not run.errors, run.errors
    assert sorted(zip(map(len, run.atom_sets), run.coefficients.tolist())) == [(3, -1), (6, 1), (6, 1)]
    e = [methods.run_hip_scf(st, fragment_bohr(z[a], xyz[a])).energy.scf for a in (slice(0, 6), slice(3, 9), slice(3, 6))]
    assert abs(run.total - (e[0] + e[1] - e[2])) < 1e-9
    plain = mbe.FragmentedSystem(z, xyz.T.copy(), [np.arange(0, 3), np.arange(3, 6), np.arange(6, 9)])
    g2 = gmbe.run_gmbe(plain, st, level=2, want_gradient=True)
    m2 = mbe.run_mbe(plain, st, level=2, want_gradient=True)
    total, _, _ = mbe.compute_mbe(m2.terms, m2.energies)
    assert not g2.errors and abs(g2.total - total) < 1e-9
    assert np.max(np.abs(g2.gradient - m2.gradient)) < 1e-9


def test_finite_difference_hessian_through_one_batch():
    """calc_hessian = the reference's finite_difference_hessian (central differences of analytic gradients, 0.005 Bohr,
    symmetrised) with the 6 N + 1 geometries as one engine batch: symmetric, translationally invariant, six zero modes,
    and its projection on an arbitrary displacement equals the second difference of the ORACLE's energy."""
    xyz = np.array([[0.02, -0.01, -0.13], [0.05, 1.44, 1.06], [-0.03, -1.40, 1.10]])
    frag = fragment_bohr([8, 1, 1], xyz)
    st = methods.ScfSettings(basis_set="sto-3g", energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_hessian(frag)
    assert not r.has_error, r.error_message
    assert r.has_hessian and r.hessian.shape == (9, 9) and r.has_gradient and r.has_energy and r.has_dipole_derivatives
    H = r.hessian
    assert np.max(np.abs(H - H.T)) == 0.0
    assert np.max(np.abs(H.reshape(9, 3, 3).sum(axis=1))) < 2e-5            # sum over atoms B of H[., B] = 0
    w = np.linalg.eigvalsh(H)
    assert np.sum(np.abs(w) < 1e-10) == 3 and np.sum(np.abs(w) < 1e-2) == 6 and np.sum(w > 0.05) == 3           # translations + rotations (non-stationary point: small), 3 vibrations
    rng = np.random.default_rng(2)
    d = rng.normal(size=(3, 3)); d /= np.linalg.norm(d)
    h = 0.01
    e = [so.run_rhf(oracle_mol("sto-3g", fragment_bohr([8, 1, 1], xyz + s_ * h * d)), 10, 200, 1e-12, 1e-10).energy for s_ in (+1, 0, -1)]
    second = (e[0] - 2 * e[1] + e[2]) / h ** 2
    assert abs(d.reshape(-1) @ H @ d.reshape(-1) - second) < 2e-4, (d.reshape(-1) @ H @ d.reshape(-1), second)
    # the charge sum rule of the dipole derivatives: sum over atoms of d mu_k / d R_A,k' = charge * delta = 0
    assert np.max(np.abs(r.dipole_derivatives.reshape(3, 3, 3).sum(axis=1))) < 1e-4



def _oracle_fragment_energy(job):
    z, coords, nelec = job
    import numpy as _np
    from metalquicha_amd.methods import PhysicalFragment
    from oracle import scf_oracle as _so
    from tests.helpers import oracle_mol as _om
    frag = PhysicalFragment(_np.asarray(z), _np.asarray(coords))
    r = _so.run_rhf(_om("cc-pvdz", frag), int(nelec), 100, 1e-10, 1e-8)
    return float(r.energy), int(r.iterations)


def test_full_size_c3_workload_sample_matches_the_oracle():
    """BASELINE configs[2] at FULL size -- (H2O)64, MBE-2, RHF/cc-pVDZ, 2080 SCFs in one engine call with bench.py's
    settings (Schwarz screening at 1e-12, 1e-10 / 1e-8) -- with a sample of 12 dimers spread over the list and 4 monomers
    against the oracle (live, eight worker processes): energies to 1e-9 (observed 1e-11), equal iteration counts; plus
    the size-independent checks on the whole list: every fragment converged, c_dimer = 1 / c_monomer = 2 - N assembly
    equals the bottom-up deltas, and a rigid motion of the cluster leaves the MBE-2 energy alone (1e-8)."""
    import multiprocessing as mp
    system = mbe.water_cluster(4)
    terms = mbe.generate_mbe_term_list(system, 2)
    assert len(terms) == 2080
    st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-10, density_tol=1e-8, schwarz_tol=1e-12)
    run = mbe.run_mbe(system, st, level=2, terms=terms)
    assert not run.errors, run.errors[:3]
    total, by_order, _ = mbe.compute_mbe(terms, run.energies)
    coef = mbe.compute_mbe_coefficients(terms)
    assert abs(float(np.dot(coef, run.energies)) - total) < 1e-8
    assert -0.5 < by_order[2] < 0.0
    dimers = [i for i, t in enumerate(terms) if len(t) == 2]
    monos = [i for i, t in enumerate(terms) if len(t) == 1]
    sample = dimers[::len(dimers) // 12][:12] + monos[::16][:4]
    jobs = []
    for i in sample:
        frag = mbe.build_fragment(system, terms[i])
        jobs.append((frag.element_numbers.tolist(), frag.coordinates.tolist(), int(frag.nelec)))
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    try:
        with mp.get_context("spawn").Pool(8) as pool:
            out = pool.map(_oracle_fragment_energy, jobs, chunksize=1)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for i, (e, it) in zip(sample, out):
        assert abs(run.energies[i] - e) < 1e-9, (terms[i], run.energies[i], e)
        assert int(run.iterations[i]) == it, (terms[i], run.iterations[i], it)
    # rigid motion of the whole cluster
    rng = np.random.default_rng(4)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    moved = mbe.FragmentedSystem(system.element_numbers, q @ system.coordinates + rng.uniform(-3, 3, size=(3, 1)), system.monomers,
                                 system.charges, system.multiplicities)
    run2 = mbe.run_mbe(moved, st, level=2, terms=terms)
    assert not run2.errors
    assert abs(mbe.compute_mbe(terms, run2.energies)[0] - total) < 1e-8
