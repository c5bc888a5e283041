"""GPU parity tests of the embedded callers: point charges in the one-electron Hamiltonian
through the C ABI (ABI 3), and FMO2 / EE-MBE of whole-molecule fragments through fmo.run_fmo2 -> engine batch calls,
against the oracle and the reference's golden energy for the EE-MBE water trimer."""
import numpy as np
import pytest

from metalquicha_amd import fmo, mbe, methods
from metalquicha_amd.methods import FragmentGroup, ScfSettings
from oracle import fmo_oracle, scf_oracle as so
from oracle import xc_oracle
from tests.helpers import EEMBE_W3_GOLDEN, FMO2_W3_GOLDEN, FMO3_W3_GOLDEN, oracle_cross_coulomb, oracle_make_mol, w3_system

pytestmark = pytest.mark.gpu

FRAGS = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def _settings(basis="6-31g", **kw):
    return ScfSettings(basis_set=basis, energy_tol=1e-9, density_tol=1e-7, guess="gwh", **kw)


@pytest.mark.parametrize("basis", ["6-31g", "cc-pvdz"])
def test_embedded_fragment_matches_oracle(basis):
    """One water and one water pair of the trimer in the field of made-up charges on the other atoms: energy, tr(D u),
    u itself, the density and the Mulliken charges (1e-9 / 1e-8 on matrices), and a second fragment of the same
    batch in vacuum-like zero charges giving the plain RHF energy."""
    system = w3_system()
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    rng = np.random.default_rng(5)
    make = oracle_make_mol(system, basis)
    for atoms in ([0, 1, 2], [0, 1, 2, 3, 4, 5]):
        out = [a for a in range(9) if a not in atoms]
        q = np.stack([rng.uniform(-0.8, 0.8, size=len(out)), np.zeros(len(out))])
        g = FragmentGroup(z[atoms].astype(np.int32), np.stack([xyz[atoms]] * 2), np.zeros(2, dtype=np.int32),
                          point_charge_xyz=np.stack([xyz[out]] * 2), point_charges=q)
        extras = []
        rec = methods.run_hip_scf_groups(_settings(basis), [g], extras=("density", "embedding_matrix", "mulliken_charges"),
                                         extras_out=extras)[0]
        assert not rec["has_error"].any(), rec["message"]
        mol = make(atoms)
        u = so.point_charge_potential(mol, xyz[out], q[0])
        ref = so.run_rhf(mol, int(np.sum(z[atoms])), e_tol=1e-9, d_tol=1e-7, h_extra=u)
        vac = so.run_rhf(mol, int(np.sum(z[atoms])), e_tol=1e-9, d_tol=1e-7)
        S, _, _ = so.int1e(mol)
        assert abs(rec["e_total"][0] - ref.energy) < 1e-9
        assert rec["iterations"][0] == ref.iterations
        assert abs(rec["e_embedding"][0] - float(np.sum(ref.D * u))) < 1e-9
        assert np.max(np.abs(extras[0]["embedding_matrix"][0] - u)) < 1e-11
        assert np.max(np.abs(extras[0]["density"][0] - ref.D)) < 1e-7
        assert np.max(np.abs(extras[0]["mulliken_charges"][0] - so.mulliken_charges(mol, ref.D, S))) < 1e-7
        assert abs(rec["e_total"][1] - vac.energy) < 1e-9 and abs(rec["e_embedding"][1]) < 1e-14
        assert abs(rec["e_nuclear"][0] - so.nuclear_repulsion(mol)) < 1e-10      # the charges stay out of E_nuc


def test_eembe_water_trimer_reference_golden():
    """validation_tests_cpu.json 'EE-MBE water trimer 6-31g (CPU)' = -227.9704573337: monomer passes and the pair
    phase as engine batches, Mulliken charges from the engine."""
    system = w3_system()
    run = fmo.run_fmo2(system, _settings(), expansion="mbe")
    assert not run.errors, run.errors
    assert run.converged
    assert abs(run.energy - EEMBE_W3_GOLDEN) < 1e-8
    ref = fmo_oracle.run_fmo2(oracle_make_mol(system, "6-31g"), np.asarray(system.element_numbers),
                              np.ascontiguousarray(system.coordinates.T), FRAGS, expansion="mbe")
    assert run.outer_iterations == ref.outer_iterations
    assert abs(run.energy - ref.energy) < 1e-9
    assert np.max(np.abs(run.charges - ref.charges)) < 1e-7


def test_fmo2_point_charge_water_trimer_matches_oracle():
    system = w3_system()
    run = fmo.run_fmo2(system, _settings(), expansion="fmo")
    assert not run.errors, run.errors
    ref = fmo_oracle.run_fmo2(oracle_make_mol(system, "6-31g"), np.asarray(system.element_numbers),
                              np.ascontiguousarray(system.coordinates.T), FRAGS, expansion="fmo")
    assert run.converged and run.outer_iterations == ref.outer_iterations
    assert abs(run.energy - ref.energy) < 1e-9
    assert abs(run.response_sum - ref.response_sum) < 1e-9
    for p, c in ref.pair_corrections.items():
        assert abs(run.pair_corrections[p] - c) < 1e-9


def test_extra_one_electron_operator_matches_oracle():
    """h_extra through the ABI (run_libcint_rhf's argument of the same name): a symmetric random operator on one of two
    fragments of a batch, with and without point charges beside it; energy, tr(D u), u = charges' part + h_extra."""
    system = w3_system()
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    mol = oracle_make_mol(system, "6-31g")([0, 1, 2])
    n = mol.nao
    rng = np.random.default_rng(3)
    hx = rng.normal(scale=0.02, size=(2, n, n)); hx = 0.5 * (hx + hx.transpose(0, 2, 1)); hx[1] = 0.0
    q = rng.uniform(-0.5, 0.5, size=6)
    for with_charges in (False, True):
        g = FragmentGroup(z[:3].astype(np.int32), np.stack([xyz[:3]] * 2), np.zeros(2, dtype=np.int32), h_extra=hx)
        if with_charges:
            g.point_charge_xyz = np.stack([xyz[3:]] * 2); g.point_charges = np.stack([q, q])
        extras = []
        rec = methods.run_hip_scf_groups(_settings(), [g], extras=("density", "embedding_matrix"), extras_out=extras)[0]
        assert not rec["has_error"].any(), rec["message"]
        upc = so.point_charge_potential(mol, xyz[3:], q) if with_charges else np.zeros((n, n))
        for f in range(2):
            u = upc + hx[f]
            ref = so.run_rhf(mol, 10, e_tol=1e-9, d_tol=1e-7, h_extra=u)
            assert abs(rec["e_total"][f] - ref.energy) < 1e-9
            assert abs(rec["e_embedding"][f] - float(np.sum(ref.D * u))) < 1e-9
            assert np.max(np.abs(extras[0]["embedding_matrix"][f] - u)) < 1e-11


def test_embedded_kohn_sham_and_density_fitted_fragments_match_oracle():
    """The field enters H only, so every two-electron path sees it: B3LYP with exact ERIs and density-fitted RHF of a
    water in the charges of its two neighbours (cc-pVDZ), against the oracle with the same h_extra."""
    from tests.helpers import fragment_bohr, oracle_mol
    system = w3_system()
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    q = np.array([-0.7, 0.35, 0.35, -0.6, 0.3, 0.3])
    frag = fragment_bohr(z[:3], xyz[:3])
    mol = oracle_mol("cc-pvdz", frag)
    u = so.point_charge_potential(mol, xyz[3:], q)
    g = FragmentGroup(z[:3].astype(np.int32), xyz[None, :3], np.zeros(1, dtype=np.int32),
                      point_charge_xyz=xyz[None, 3:], point_charges=q[None, :])
    aux = "mqc-even-tempered-jkfit"
    for kw, okw in ((dict(functional="b3lyp"), dict(xc=xc_oracle.XCOracle(mol, "b3lyp", 3))),
                    (dict(density_fitting=True, aux_basis_set=aux), dict(aux=oracle_mol(aux, frag)))):
        rec = methods.run_hip_scf_groups(_settings("cc-pvdz", **kw), [g])[0]
        assert not rec["has_error"][0], rec["message"][0]
        ref = so.run_rhf(mol, 10, e_tol=1e-9, d_tol=1e-7, h_extra=u, **okw)
        assert abs(rec["e_total"][0] - ref.energy) < 2e-9
        assert abs(rec["e_embedding"][0] - float(np.sum(ref.D * u))) < 1e-8


def test_cross_coulomb_of_a_neighbour_matches_oracle():
    """mqc_hip_coulomb_batch behind fmo.hip_cross_coulomb: monomer + neighbour and pair + neighbour supersystems of two
    element sequences and several geometries in one request list, against the oracle's four-centre integrals."""
    system = w3_system()
    st = _settings()
    make = oracle_make_mol(system, "6-31g")
    dens = {k: so.run_rhf(make(FRAGS[k]), 10).D for k in range(3)}
    requests = [([0, 1, 2], [3, 4, 5], dens[1]), ([0, 1, 2], [6, 7, 8], dens[2]), ([3, 4, 5], [0, 1, 2], dens[0]),
                ([0, 1, 2, 6, 7, 8], [3, 4, 5], dens[1]), ([0, 1, 2, 3, 4, 5], [6, 7, 8], dens[2])]
    got = fmo.hip_cross_coulomb(system, st)(requests)
    ref = oracle_cross_coulomb(system, "6-31g")(requests)
    for g, r in zip(got, ref):
        assert g.shape == r.shape and np.max(np.abs(g - r)) < 1e-11


def test_coulomb_batch_full_and_cross_modes():
    """The entry by itself: n_source_atoms = 0 gives the full Coulomb matrix of a full density for every fragment of
    the batch; n_source_atoms = 3 forms only the (leading | source) quartets and must give the same leading block for a
    density that lives on the source atoms (cc-pVDZ: d shells, three geometries, block sharing on)."""
    from tests import stages
    from tests.helpers import fragment_bohr, oracle_mol, synthetic_density, water_at
    rng = np.random.default_rng(21)
    w0 = water_at(rng, [0, 0, 0])
    frags = [fragment_bohr([8, 1, 1] * 2, np.vstack([w0, water_at(rng, c)])) for c in ([5.4, 0.2, 0.1], [0.3, 5.8, -0.2], [-0.5, 0.4, 6.1])]
    mols = [oracle_mol("cc-pvdz", f) for f in frags]
    n = mols[0].nao
    D = np.stack([synthetic_density(n)] * 3)
    J = stages.coulomb_batch("cc-pvdz", frags, D, 0)
    for f in range(3):
        Jo, _ = so.build_jk_incore(so.eri4(mols[f]), D[f])
        assert np.max(np.abs(J[f] - Jo)) < 1e-10
    nk = n // 2
    Dk = np.zeros_like(D); Dk[:, nk:, nk:] = D[:, nk:, nk:]
    Jc = stages.coulomb_batch("cc-pvdz", frags, Dk, 3)
    for f in range(3):
        Jo, _ = so.build_jk_incore(so.eri4(mols[f]), Dk[f])
        assert np.max(np.abs(Jc[f][:nk, :nk] - Jo[:nk, :nk])) < 1e-10


def test_fmo2_exact_esp_water_trimer_reference_golden():
    """validation_tests_cpu.json 'FMO2 water trimer 6-31g (CPU)' = -227.9705411684: the reference's default FMO2 (exact
    ESP of the near fragments, resppc 2.0) -- nuclei as charges, J[D_K] from the engine's J/K kernel as h_extra."""
    system = w3_system()
    run = fmo.run_fmo2(system, _settings(), expansion="fmo", esp="exact")
    assert not run.errors, run.errors
    assert run.converged
    assert abs(run.energy - FMO2_W3_GOLDEN) < 1e-8
    ref = fmo_oracle.run_fmo2(oracle_make_mol(system, "6-31g"), np.asarray(system.element_numbers),
                              np.ascontiguousarray(system.coordinates.T), FRAGS, expansion="fmo", esp="exact")
    assert run.outer_iterations == ref.outer_iterations
    assert abs(run.energy - ref.energy) < 1e-9
    assert abs(run.response_sum - ref.response_sum) < 1e-9
    # a tighter cutoff turns the far water of each end into charges: mixed field
    run = fmo.run_fmo2(system, _settings(), expansion="fmo", esp="exact", resppc=1.5)
    ref = fmo_oracle.run_fmo2(oracle_make_mol(system, "6-31g"), np.asarray(system.element_numbers),
                              np.ascontiguousarray(system.coordinates.T), FRAGS, expansion="fmo", esp="exact", resppc=1.5)
    assert abs(run.energy - ref.energy) < 1e-9


def test_level_three_reference_goldens():
    """'FMO3 / EE-MBE3 water trimer 6-31g, exact at full level (CPU)' = -227.970497639 for both expansions: monomers,
    pairs and the trimer in one pair-phase batch, corrections telescoping to the supermolecular energy."""
    system = w3_system()
    for kw in (dict(expansion="fmo", esp="exact"), dict(expansion="mbe")):
        run = fmo.run_fmo2(system, _settings(), level=3, **kw)
        assert not run.errors, run.errors
        assert abs(run.energy - FMO3_W3_GOLDEN) < 1e-8
        assert len(run.pair_corrections) == 4


def test_eembe_water_cluster_matches_oracle():
    """(H2O)8 of the bench's cluster builder, cc-pVDZ: 8 monomers x passes + 28 pairs, every pass one batch."""
    system = mbe.water_cluster(2, seed=11)
    frags = [list(map(int, m)) for m in system.monomers]
    run = fmo.run_fmo2(system, _settings("cc-pvdz"), expansion="mbe")
    assert not run.errors, run.errors
    from tests.helpers import fragment_bohr, recorded_oracle

    def oracle():
        r = fmo_oracle.run_fmo2(oracle_make_mol(system, "cc-pvdz"), np.asarray(system.element_numbers),
                                np.ascontiguousarray(system.coordinates.T), frags, expansion="mbe")
        return {"energy": float(r.energy), "iterations": int(r.outer_iterations), "converged": bool(r.converged),
                "monomer_energy": [float(v) for v in r.monomer_energy]}
    # the oracle's 54 s of CPU come from the committed fixture (MQC_ORACLE_LIVE=1 computes it here instead)
    ref = recorded_oracle("eembe2_water8", fragment_bohr(system.element_numbers, system.coordinates.T),
                          "cc-pvdz|ee-mbe2|ptc|mulliken|1e-9|1e-7|gwh|outer 1e-7", oracle)
    assert run.converged and run.outer_iterations == ref["iterations"]
    assert abs(run.energy - ref["energy"]) < 2e-9
    assert np.max(np.abs(run.monomer_energy - np.array(ref["monomer_energy"]))) < 1e-9


def test_embedded_gradient_is_refused():
    system = w3_system()
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    g = FragmentGroup(z[:3].astype(np.int32), xyz[None, :3], np.zeros(1, dtype=np.int32),
                      point_charge_xyz=xyz[None, 3:], point_charges=np.full((1, 6), 0.1))
    rec = methods.run_hip_scf_groups(_settings(), [g], want_gradient=True, gradients_out=[])[0]
    assert rec["has_error"][0] and b"point charges" in bytes(rec["message"][0])


@pytest.mark.parametrize("basis", ["cc-pvdz", "6-31g"])
def test_large_point_charge_field_far_table_matches_oracle(basis):
    """A field of 400 charges (a 512-fragment FMO run carries ~1500 per fragment): charges beyond an atom's far radius
    (p_min R^2 > 42) enter the same-centre shell pairs through the per-atom table of sum q d^{tuv}(1/|A - C|)
    (kern_int1e.hip, pc_far_table_kernel), the near ones and all two-centre pairs through the direct sum.  The embedding
    operator u must equal the oracle's point-charge potential to 1e-11 per element, the energy to 1e-9 -- for a water
    and for a water pair (s, p and d pairs on one centre and on two)."""
    system = w3_system()
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    rng = np.random.default_rng(17)
    make = oracle_make_mol(system, basis)
    for atoms in ([0, 1, 2], [0, 1, 2, 3, 4, 5]):
        centre = xyz[atoms].mean(axis=0)
        npc = 400
        direction = rng.normal(size=(npc, 3)); direction /= np.linalg.norm(direction, axis=1)[:, None]
        radius = np.concatenate([rng.uniform(4.0, 9.0, size=60), rng.uniform(9.0, 70.0, size=npc - 60)])     # near and far
        pts = centre + direction * radius[:, None]
        q = rng.uniform(-0.9, 0.9, size=npc)
        g = FragmentGroup(z[atoms].astype(np.int32), xyz[atoms][None], np.zeros(1, dtype=np.int32),
                          point_charge_xyz=pts[None], point_charges=q[None])
        extras = []
        rec = methods.run_hip_scf_groups(_settings(basis), [g], extras=("embedding_matrix",), extras_out=extras)[0]
        assert not rec["has_error"].any(), rec["message"]
        mol = make(atoms)
        u = so.point_charge_potential(mol, pts, q)
        assert np.max(np.abs(extras[0]["embedding_matrix"][0] - u)) < 1e-11
        ref = so.run_rhf(mol, int(np.sum(z[atoms])), e_tol=1e-9, d_tol=1e-7, h_extra=u)
        assert abs(rec["e_total"][0] - ref.energy) < 1e-9
        assert rec["iterations"][0] == ref.iterations
