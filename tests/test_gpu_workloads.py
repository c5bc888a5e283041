"""GPU parity tests of BASELINE.json's larger configurations at reduced size, with the METHOD each one names:

  configs[3]  GMBE-2 B3LYP/def2-TZVP with the XC grid kernel   -> gmbe.run_gmbe(level=2) on (H2O)3, f shells, n = 43 / 86
  meta-GGA    TPSS, exact and density-fitted Coulomb              -> a water and a water dimer in one batch
  configs[4]  512-fragment FMO-2 DF-RKS                          -> fmo.run_fmo2 with density fitting AND a functional on
                                                                    (H2O)8 against the oracle's driver, and a c5-SHAPED
                                                                    27-fragment run checked through properties

The oracle side is oracle/fmo_oracle.py (pinned to the reference's FMO2 / EE-MBE manifest rows, tests/test_fmo_host.py)
around oracle/scf_oracle.run_rhf with `aux` and `xc` (pinned to the manifest's DF and Kohn-Sham rows,
tests/test_oracle_golden.py).  The slow oracle runs come from tests/golden/oracle_fixtures.json (recorded by
tests/golden/record_oracle_fixtures.py on the CPU; MQC_ORACLE_LIVE=1 runs them here instead)."""
import numpy as np
import pytest

from metalquicha_amd import fmo, gmbe, mbe, methods
from metalquicha_amd.methods import ScfSettings
from tests import workload_cases as wc
from tests.helpers import fragment_bohr, recorded_oracle

pytestmark = pytest.mark.gpu


def test_fmo2_df_rks_water8_matches_oracle():
    """configs[4]'s method -- FMO-2, density-fitted, Kohn-Sham (B3LYP) -- on (H2O)8 / cc-pVDZ: total to 2e-9 Eh, the
    same number of outer passes, monomer energies and Mulliken charges (mqc_libcint_fmo.f90:1484-1564, 1566-1689 around
    the DF-RKS SCF of the cuEST path)."""
    system = wc.fmo_df_rks_system()
    run = fmo.run_fmo2(system, wc.fmo_df_rks_settings(), expansion="fmo")
    assert not run.errors, run.errors
    ref = recorded_oracle("fmo2_df_b3lyp_water8", fragment_bohr(system.element_numbers, system.coordinates.T),
                          wc.FMO_DF_RKS_KEY, wc.fmo_df_rks_oracle)
    assert run.converged and run.outer_iterations == ref["iterations"]
    assert abs(run.energy - ref["energy"]) < 2e-9, (run.energy, ref["energy"])
    assert np.max(np.abs(run.monomer_energy - np.array(ref["monomer_energy"]))) < 1e-9
    assert abs(run.response_sum - ref["response_sum"]) < 1e-9
    assert np.max(np.abs(run.charges - np.array(ref["charges"]))) < 1e-6


def test_fmo2_df_rks_c5_shaped_properties():
    """configs[4] SHAPED: 27 fragments (3 x 3 x 3 waters), FMO-2 DF-RKS, every pass and the pair phase ONE batch each.
    Size-independent properties: every SCF converges, no failed fragment, the charges of each (neutral) fragment sum
    to zero, the pair corrections are small and attractive in sum, and a two-way rank split summed as the all-reduce
    would gives the one-rank energy."""
    system = mbe.water_cluster(3, seed=5)
    st = ScfSettings(basis_set="cc-pvdz", functional="b3lyp", density_fitting=True, aux_basis_set=wc.AUX,
                     energy_tol=1e-9, density_tol=1e-7, guess="gwh")
    one = fmo.run_fmo2(system, st, expansion="fmo")
    assert not one.errors, one.errors
    assert one.converged and 2 <= one.outer_iterations <= 20
    assert len(one.pair_corrections) == 27 * 26 // 2
    for m in system.monomers:
        assert abs(float(np.sum(one.charges[list(m)]))) < 1e-8
    corr = np.array(list(one.pair_corrections.values()))
    assert np.max(np.abs(corr)) < 0.05 and np.sum(corr) < 0.0
    assert -76.5 * 27 < one.energy < -76.3 * 27

    energy2 = _two_rank_replay(system, st)
    assert abs(energy2 - one.energy) < 1e-9, (energy2, one.energy)


def _two_rank_replay(system, st):
    """world = 2 on one GPU without a process group: both ranks run in lock step on threads and meet in an in-process
    element-wise SUM (exactly what `allreduce` must do); the engine context is process-wide and calls are serialised
    by a lock, so this exercises the rank split, the exchanges per pass and the n-mer sharing -- not concurrency."""
    import threading
    barrier = threading.Barrier(2)
    lock = threading.Lock()
    slots = [None, None]
    out = [None, None]
    base = fmo.hip_solver(system, st)

    def solver(jobs):
        with lock:
            return base(jobs)

    def make(rank):
        def allreduce(a):
            slots[rank] = np.array(a, dtype=np.float64, copy=True)
            barrier.wait()
            total = slots[0] + slots[1]
            barrier.wait()
            return total
        return allreduce

    def body(rank):
        try:
            out[rank] = fmo.run_fmo2(system, st, expansion="fmo", rank=rank, world=2, allreduce=make(rank), solver=solver)
        except BaseException as e:
            out[rank] = e
            barrier.abort()
    threads = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in out:
        assert isinstance(r, fmo.FmoRun), r
        assert not r.errors, r.errors
    assert out[0].energy == out[1].energy
    return out[0].energy


def test_gmbe2_b3lyp_def2_tzvp_matches_per_fragment_oracle():
    """configs[3]'s method -- GMBE(2), B3LYP, def2-TZVP (f shells on oxygen, n = 43 per water, 86 per pair), XC on the
    level-3 grid -- on (H2O)3: every subsystem energy against the oracle (1e-8), and the inclusion-exclusion total
    against the same sum of oracle energies (gmbe_enumerate_pie_terms, src/fragmentation/gmbe/mqc_gmbe_utils.f90)."""
    system = wc.gmbe_system()
    st = wc.gmbe_settings()
    run = gmbe.run_gmbe(system, st, level=2)
    assert not run.errors, run.errors
    assert sorted(len(s) for s in run.atom_sets) == [3, 3, 3, 6, 6, 6]
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    total = 0.0
    for atoms, c, e in zip(run.atom_sets, run.coefficients, run.energies):
        f = fragment_bohr(z[list(atoms)], xyz[list(atoms)])
        o = recorded_oracle("gmbe2_b3lyp_def2tzvp", f, wc.GMBE_KEY, lambda f=f: wc.gmbe_fragment_oracle(f))
        assert abs(e - o["energy"]) < 1e-8, (atoms, e, o["energy"])
        total += c * o["energy"]
    assert abs(run.total - total) < 3e-8
    # non-overlapping monomers: GMBE(2) is MBE(2) (c_dimer = 1, c_monomer = 2 - N)
    assert sorted(int(round(c)) for c in run.coefficients) == [-1, -1, -1, 1, 1, 1]


@pytest.mark.parametrize("df", [False, True], ids=["exact", "density-fitted"])
def test_tpss_batch_matches_oracle(df):
    """Meta-GGA leg (tau in the density, the v_tau term in the potential; mqc_libcint_xc.F90:1436-1448): TPSS on a
    water and a water dimer in one batch against the oracle, whose TPSS is pinned by the reference's own
    -76.422747225964 (tests/test_oracle_golden.py)."""
    frags = wc.tpss_fragments()
    res = methods.run_hip_scf_batch(wc.tpss_settings(df), frags)
    for f, r in zip(frags, res):
        assert not r.has_error, r.error_message
        o = recorded_oracle("tpss_water_batch", f, wc.TPSS_KEY % ("df:" + wc.AUX if df else "exact"),
                                    lambda f=f: wc.tpss_oracle(f, df))
        assert o["converged"]
        assert abs(r.energy.scf - o["energy"]) < 1e-8, (r.energy.scf, o["energy"])


def test_tpss_refusals_name_what_is_missing():
    w = wc.tpss_fragments()[0]
    r = methods.run_hip_scf(methods.ScfSettings(basis_set="sto-3g", functional="m06-l"), w)
    assert r.has_error and "not available" in r.error_message
    r = methods.run_hip_scf(methods.ScfSettings(basis_set="sto-3g", functional="tpss"), w, want_gradient=True)
    assert r.has_error and "gradient" in r.error_message


def test_density_fitted_gradient_matches_differences_of_the_oracle_energy():
    """The reference GPU backend's gradient is the density-fitted one (compute_scf_gradient, mqc_cuest_gradient.f90:91-175).
    Engine: Gam^P (mu nu|P)' - 1/2 gam (P|Q)' through the quartet gradient kernel with a unit s shell in the empty slots.
    Checked the way validation/check_gradient.f90 checks a gradient (its bound: 3.5e-8 Eh/a0), against differences of the
    ORACLE's density-fitted RHF energy (d shells in the orbital basis, up to f in the auxiliary basis)."""
    frag = fragment_bohr([8, 1, 1], wc.DF_GRAD_XYZ)
    st = ScfSettings(basis_set="cc-pvdz", density_fitting=True, aux_basis_set=wc.AUX, energy_tol=1e-12, density_tol=1e-10,
                     guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_gradient(frag)
    assert not r.has_error, r.error_message
    o = recorded_oracle("df_rhf_gradient_water", frag, wc.DF_GRAD_KEY, wc.df_gradient_oracle)
    fd = np.array(o["gradient"])
    assert np.max(np.abs(r.gradient - fd)) < 3.5e-8, (np.max(np.abs(r.gradient - fd)), r.gradient, fd)
    assert np.max(np.abs(r.gradient.sum(axis=1))) < 1e-9


@pytest.mark.parametrize("kind", ["df-b3lyp", "df-uhf", "df-pbe-dimer"])
def test_density_fitted_gradients_match_differences_of_the_engine_energy(kind):
    """Kohn-Sham (the exchange part of Gam scaled by the functional's exact-exchange fraction, none for PBE), unrestricted
    (spin densities in the exchange part) and a two-molecule batch member: central differences of the engine's own
    density-fitted energies, one batch."""
    if kind == "df-pbe-dimer":
        z = [8, 1, 1, 8, 1, 1]
        xyz = np.vstack([wc.DF_GRAD_XYZ, wc.DF_GRAD_XYZ[:, [1, 0, 2]] * 0.97 + np.array([5.1, 0.3, -0.4])])
        extra, mult, basis = dict(functional="pbe"), 1, "6-31g"
    elif kind == "df-uhf":
        z, xyz = [8, 1], np.array([[0.02, -0.01, 0.0], [0.11, 0.07, 1.83]])
        extra, mult, basis = dict(), 2, "cc-pvdz"
    else:
        z, xyz = [8, 1, 1], wc.DF_GRAD_XYZ
        extra, mult, basis = dict(functional="b3lyp"), 1, "cc-pvdz"
    st = ScfSettings(basis_set=basis, density_fitting=True, aux_basis_set=wc.AUX, energy_tol=1e-12, density_tol=1e-10,
                     guess="gwh", max_iter=200, **extra)
    r = methods.HFMethod(st).calc_gradient(fragment_bohr(z, xyz, multiplicity=mult))
    assert not r.has_error, r.error_message
    h = 2e-3
    frags = []
    na = len(z)
    for a in range(na):
        for c in range(3):
            for sgn in (+1, -1):
                x = xyz.copy(); x[a, c] += sgn * h
                frags.append(fragment_bohr(z, x, multiplicity=mult))
    e = [q.energy.scf for q in methods.run_hip_scf_batch(st, frags)]
    fd = np.zeros((3, na))
    k = 0
    for a in range(na):
        for c in range(3):
            fd[c, a] = (e[k] - e[k + 1]) / (2 * h); k += 2
    assert np.max(np.abs(r.gradient - fd)) < 5e-6, (np.max(np.abs(r.gradient - fd)), r.gradient, fd)
    assert np.max(np.abs(r.gradient.sum(axis=1))) < 1e-8


@pytest.mark.parametrize("functional", ["", "b3lyp"], ids=["rhf", "b3lyp"])
def test_density_fitting_with_f_orbital_shells_matches_oracle(functional):
    """configs[3]'s basis (def2-TZVP: an f shell on every oxygen) density-fitted, as the reference GPU backend runs it
    (mqc_cuest_integrals.f90:702-736,1636-1748): the three-centre classes with an f shell in the bra come from the
    general LDS kernel with a unit s shell in the fourth slot.  A water and a water dimer (n = 43 / 86) in one batch."""
    frags = wc.df_f_fragments()
    res = methods.run_hip_scf_batch(wc.df_f_settings(functional), frags)
    for f, r in zip(frags, res):
        assert not r.has_error, r.error_message
        o = recorded_oracle("df_f_orbitals", f, wc.DF_F_KEY % (functional or "rhf"), lambda f=f: wc.df_f_oracle(f, functional))
        assert o["converged"]
        assert abs(r.energy.scf - o["energy"]) < 1e-8, (r.energy.scf, o["energy"])


@pytest.mark.parametrize("df", [False, True], ids=["exact", "density-fitted"])
def test_f_shell_gradient_matches_differences_of_the_oracle_energy(df):
    """Gradients with f shells (def2-TZVP, configs[3]'s basis): CO puts an f shell on both centres, so every f class of the
    quartet gradient kernel runs, (f f|f d) and (f f|f f) with their ket columns in chunks.  check_gradient's procedure and
    bound (validation/check_gradient.f90: 3.5e-8 Eh/a0) with the oracle as the energy function."""
    frag = fragment_bohr(wc.F_GRAD_Z, wc.F_GRAD_XYZ)
    st = ScfSettings(basis_set="def2-tzvp", density_fitting=df, aux_basis_set=wc.AUX if df else "", energy_tol=1e-12,
                     density_tol=1e-10, guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_gradient(frag)
    assert not r.has_error, r.error_message
    o = recorded_oracle("f_shell_gradient_co", frag, wc.F_GRAD_KEY % ("df:" + wc.AUX if df else "exact"),
                        lambda: wc.f_gradient_oracle(df))
    fd = np.array(o["gradient"])
    assert np.max(np.abs(r.gradient - fd)) < 3.5e-8, (np.max(np.abs(r.gradient - fd)), r.gradient, fd)
    assert np.max(np.abs(r.gradient.sum(axis=1))) < 1e-9


def test_kohn_sham_f_shell_gradient_matches_differences_of_the_engine_energy():
    """configs[3]'s method with a gradient: B3LYP/def2-TZVP on CO (second derivatives of the f functions in the
    quadrature's gradient), central differences of the engine's own energies."""
    st = ScfSettings(basis_set="def2-tzvp", functional="b3lyp", energy_tol=1e-12, density_tol=1e-10, guess="gwh", max_iter=200)
    r = methods.HFMethod(st).calc_gradient(fragment_bohr(wc.F_GRAD_Z, wc.F_GRAD_XYZ))
    assert not r.has_error, r.error_message
    h = 2e-3
    frags = []
    for a in range(2):
        for c in range(3):
            for sgn in (+1, -1):
                x = wc.F_GRAD_XYZ.copy(); x[a, c] += sgn * h
                frags.append(fragment_bohr(wc.F_GRAD_Z, x))
    e = [q.energy.scf for q in methods.run_hip_scf_batch(st, frags)]
    fd = np.zeros((3, 2))
    k = 0
    for a in range(2):
        for c in range(3):
            fd[c, a] = (e[k] - e[k + 1]) / (2 * h); k += 2
    assert np.max(np.abs(r.gradient - fd)) < 5e-6, (np.max(np.abs(r.gradient - fd)), r.gradient, fd)
    assert np.max(np.abs(r.gradient.sum(axis=1))) < 1e-8
