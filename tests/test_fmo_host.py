"""FMO2 / EE-MBE with point-charge embedding, no GPU: the oracle against the reference's golden energy, the host driver
(metalquicha_amd/fmo.py) against the oracle through an oracle-backed solver, and its two-rank exchange over gloo."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from metalquicha_amd import fmo
from metalquicha_amd.methods import ScfSettings
from oracle import fmo_oracle, scf_oracle as so
from tests.helpers import EEMBE_W3_GOLDEN, FMO2_W3_GOLDEN, FMO3_W3_GOLDEN, oracle_cross_coulomb, oracle_fmo_solver, oracle_make_mol, w3_system

FRAGS = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def _oracle(expansion, **kw):
    system = w3_system()
    return fmo_oracle.run_fmo2(oracle_make_mol(system, "6-31g"), np.asarray(system.element_numbers),
                               np.ascontiguousarray(system.coordinates.T), FRAGS, expansion=expansion, **kw)


def test_oracle_reproduces_the_reference_fmo2_water_trimer():
    """The reference's default FMO2: exact ESP of near fragments (all three waters are within 2.0 van der Waals sums)."""
    r = _oracle("fmo", esp="exact")
    assert r.converged
    assert abs(r.energy - FMO2_W3_GOLDEN) < 1e-9
    assert abs(r.response_sum) > 1e-5


def test_level_three_telescopes_to_the_supermolecule():
    """Manifest rows 'FMO3 / EE-MBE3 water trimer 6-31g, exact at full level': at level = number of fragments the
    corrections telescope (response inside the recursion) and both expansions give the supermolecular RHF energy."""
    system = w3_system()
    for kw in (dict(expansion="fmo", esp="exact"), dict(expansion="mbe")):
        ref = _oracle(level=3, **kw)
        assert abs(ref.energy - FMO3_W3_GOLDEN) < 1e-9
        run = fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), level=3, solver=oracle_fmo_solver(system, "6-31g"),
                           coulomb=oracle_cross_coulomb(system, "6-31g"), **kw)
        assert abs(run.energy - ref.energy) < 1e-11
        assert abs(run.pair_corrections[(0, 1, 2)] - ref.pair_corrections[(0, 1, 2)]) < 1e-11
        assert sorted(run.pair_corrections) == [(0, 1), (0, 1, 2), (0, 2), (1, 2)]


def test_no_field_and_ignored_far_field():
    """esp = "none" is the plain MBE(2); a point-charge field whose distant fragments are ignored is no field at all
    (check_fmo.f90 separated_case: ignored == bare to 1e-12); with an exact near field and an ignored far one the
    driver still equals the oracle."""
    system = w3_system()
    make = oracle_make_mol(system, "6-31g")
    st = ScfSettings(basis_set="6-31g")
    solver, cc = oracle_fmo_solver(system, "6-31g"), oracle_cross_coulomb(system, "6-31g")
    e1 = [so.run_rhf(make(f), 10, e_tol=1e-9, d_tol=1e-7).energy for f in FRAGS]
    plain = sum(e1)
    for i in range(3):
        for j in range(i + 1, 3):
            plain += so.run_rhf(make(FRAGS[i] + FRAGS[j]), 20, e_tol=1e-9, d_tol=1e-7).energy - e1[i] - e1[j]
    bare = fmo.run_fmo2(system, st, expansion="mbe", esp="none", solver=solver)
    assert bare.converged and bare.outer_iterations == 1 and abs(bare.energy - plain) < 1e-11
    ignored = fmo.run_fmo2(system, st, expansion="fmo", esp="ptc", far_field="ignore", solver=solver)
    assert abs(ignored.energy - bare.energy) < 1e-12 and ignored.response_sum == 0.0
    assert abs(_oracle("mbe", esp="none").energy - plain) < 1e-11
    ref = _oracle("fmo", esp="exact", resppc=1.5, far_field="ignore")
    run = fmo.run_fmo2(system, st, expansion="fmo", esp="exact", resppc=1.5, far_field="ignore", solver=solver, coulomb=cc)
    assert abs(run.energy - ref.energy) < 1e-11
    assert abs(run.energy - _oracle("fmo", esp="exact", resppc=1.5).energy) > 1e-7      # the far charges do matter


def test_product_solver_fails_loudly_without_a_device():
    """No CPU fallback behind the embedded callers either: without a HIP device the default solver raises."""
    import pytest
    from metalquicha_amd import capi
    if capi.load_library().mqc_hip_backend_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.HipBackendError):
        fmo.run_fmo2(w3_system(), ScfSettings(basis_set="6-31g"))
    with pytest.raises(capi.HipBackendError):
        fmo.hip_cross_coulomb(w3_system(), ScfSettings(basis_set="6-31g"))([([0, 1, 2], [3, 4, 5], np.zeros((13, 13)))])


def test_failed_fragment_stops_before_the_pair_phase():
    """run_fmo2 refuses a total when a fragment SCF failed (mqc_libcint_fmo.f90:489-494): errors out, energy NaN."""
    system = w3_system()
    good = oracle_fmo_solver(system, "6-31g")

    def bad(jobs):
        out = good(jobs)
        out[0].error = "made to fail"
        return out
    run = fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), solver=bad)
    assert not run.converged and np.isnan(run.energy) and "made to fail" in run.errors[0]


def test_near_fragment_cutoff():
    system = w3_system()
    assert fmo.near_fragments(system, [0], 2.0) == [1, 2]          # O-O 5.8 A / 3.04 A = 1.91
    assert fmo.near_fragments(system, [0], 1.5) == [1]
    assert fmo.near_fragments(system, [0, 1], 0.5) == []
    assert fmo.near_fragments(system, [2], -1.0) == [0, 1]
    z = np.asarray(system.element_numbers); xyz = np.ascontiguousarray(system.coordinates.T)
    for cut in (2.0, 1.5, 0.5, -1.0):
        assert fmo_oracle.near_fragments(z, xyz, FRAGS, [0], cut) == fmo.near_fragments(system, [0], cut)


def test_host_driver_exact_esp_equals_the_oracle():
    system = w3_system()
    for resppc in (2.0, 1.5):                                          # 1.5: the far water of each end is a point charge
        ref = _oracle("fmo", esp="exact", resppc=resppc)
        run = fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), expansion="fmo", esp="exact", resppc=resppc,
                           solver=oracle_fmo_solver(system, "6-31g"), coulomb=oracle_cross_coulomb(system, "6-31g"))
        assert run.converged and run.outer_iterations == ref.outer_iterations
        assert abs(run.energy - ref.energy) < 1e-11
        assert abs(run.response_sum - ref.response_sum) < 1e-11


def test_oracle_reproduces_the_reference_eembe_water_trimer():
    r = _oracle("mbe")
    assert r.converged
    assert abs(r.energy - EEMBE_W3_GOLDEN) < 1e-9
    assert r.response_sum == 0.0                       # EE-MBE carries no response term (nmer_term, :1266-1272)


def test_point_charge_operator_is_the_nuclear_attraction_of_the_charges():
    system = w3_system()
    mol = oracle_make_mol(system, "6-31g")([0, 1, 2])
    _, _, V = so.int1e(mol)
    assert np.allclose(so.point_charge_potential(mol, mol.xyz, mol.z), V, atol=1e-13)
    S, _, _ = so.int1e(mol)
    r = so.run_rhf(mol, 10)
    q = so.mulliken_charges(mol, r.D, S)
    assert abs(np.sum(q)) < 1e-9 and q[0] < 0 < q[1]


def test_host_driver_equals_the_oracle_for_both_expansions():
    system = w3_system()
    for expansion in ("mbe", "fmo"):
        ref = _oracle(expansion)
        run = fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), expansion=expansion, solver=oracle_fmo_solver(system, "6-31g"))
        assert run.converged and run.outer_iterations == ref.outer_iterations
        assert abs(run.energy - ref.energy) < 1e-11
        assert abs(run.response_sum - ref.response_sum) < 1e-11
        assert np.allclose(run.charges, ref.charges, atol=1e-11)
        assert np.allclose(run.monomer_energy, ref.monomer_energy, atol=1e-11)
    assert abs(run.response_sum) > 1e-5                # the fmo expansion's response is not zero for three fragments


def test_two_rank_gloo_fmo_equals_serial(tmp_path):
    """world_size = 2 over gloo: fragments and pairs round-robin over the ranks, one all-reduce per pass
    (exchange_monomers, mqc_libcint_fmo.f90:1890-1948); both ranks must end on the serial energy."""
    script = tmp_path / "rank.py"
    script.write_text(
        "import sys\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "from metalquicha_amd import fmo\n"
        "from metalquicha_amd.methods import ScfSettings\n"
        "from tests.helpers import oracle_cross_coulomb, oracle_fmo_solver, w3_system\n"
        "dist.init_process_group('gloo', init_method='env://')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "def allreduce(a):\n"
        "    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy()); dist.all_reduce(t); return t.numpy()\n"
        "system = w3_system(); solver = oracle_fmo_solver(system, '6-31g'); cc = oracle_cross_coulomb(system, '6-31g')\n"
        "for expansion, esp in (('mbe', 'ptc'), ('fmo', 'ptc'), ('fmo', 'exact')):\n"
        "    par = fmo.run_fmo2(system, ScfSettings(basis_set='6-31g'), expansion=expansion, esp=esp, rank=r, world=w, allreduce=allreduce, solver=solver, coulomb=cc)\n"
        "    ser = fmo.run_fmo2(system, ScfSettings(basis_set='6-31g'), expansion=expansion, esp=esp, solver=solver, coulomb=cc)\n"
        "    assert abs(par.energy - ser.energy) < 1e-11, (par.energy, ser.energy)\n"
        "    assert abs(par.response_sum - ser.response_sum) < 1e-11\n"
        "print('rank', r, 'ok')\n"
        "dist.destroy_process_group()\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def _thread_ranks(world, body):
    """Runs body(rank, allreduce) on `world` threads with an element-wise SUM all-reduce between them (the exchange
    run_fmo2 needs, without a process group)."""
    import threading
    barrier = threading.Barrier(world)
    slots = [None] * world
    out = [None] * world

    def make_allreduce(rank):
        def allreduce(a):
            slots[rank] = np.array(a, dtype=np.float64, copy=True)
            barrier.wait()
            total = sum(slots[r] for r in range(world))
            barrier.wait()
            return total
        return allreduce

    def run(rank):
        try:
            out[rank] = body(rank, make_allreduce(rank))
        except BaseException as e:      # a rank that dies must not leave the other at the barrier forever
            out[rank] = e
            barrier.abort()
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    return out


def test_failed_pair_on_one_rank_refuses_the_total_on_every_rank():
    """A pair SCF that fails on rank 1 of 2 must not enter the total as a zero (that would shift it by a whole pair
    energy): every rank returns no energy, not converged, and an error -- calculate_polymers returns on nmer_term's
    error (mqc_libcint_fmo.f90:1652-1653)."""
    system = w3_system()
    good = oracle_fmo_solver(system, "6-31g")

    def body(rank, allreduce):
        def solver(jobs):
            out = good(jobs)
            if rank == 1:
                for job, r in zip(jobs, out):
                    if len(job.atoms) == 6:
                        r.error = "pair made to fail"
            return out
        return fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), expansion="fmo", rank=rank, world=2,
                            allreduce=allreduce, solver=solver)
    runs = _thread_ranks(2, body)
    for rank, run in enumerate(runs):
        assert isinstance(run, fmo.FmoRun), run
        assert np.isnan(run.energy) and not run.converged and run.errors, (rank, run)
    assert "pair made to fail" in runs[1].errors[0]
    serial = fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), expansion="fmo", solver=good)
    assert serial.converged and np.isfinite(serial.energy)


def test_failed_monomer_in_exact_esp_mode_is_refused_not_a_crash():
    """esp = "exact": a monomer that fails in the bare pass leaves no density; the next pass must not be built from it
    (it used to dereference None in the Coulomb requests)."""
    system = w3_system()
    good = oracle_fmo_solver(system, "6-31g")
    calls = []

    def solver(jobs):
        out = good(jobs)
        calls.append(len(jobs))
        if len(calls) == 1:
            out[1].error = "bare pass made to fail"
        return out
    run = fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), expansion="fmo", esp="exact", solver=solver,
                       coulomb=oracle_cross_coulomb(system, "6-31g"))
    assert np.isnan(run.energy) and not run.converged and "bare pass made to fail" in run.errors[0]
    assert calls == [3]                                  # nothing after the failed pass was attempted


def test_outer_loop_that_does_not_settle_is_an_error():
    """calculate_monomers (mqc_libcint_fmo.f90:1560-1563): max_outer passes without settling -> error, no pair phase."""
    system = w3_system()
    calls = []
    good = oracle_fmo_solver(system, "6-31g")

    def solver(jobs):
        calls.append(len(jobs))
        return good(jobs)
    run = fmo.run_fmo2(system, ScfSettings(basis_set="6-31g"), expansion="fmo", max_outer=1, outer_tol=1e-14, solver=solver)
    assert np.isnan(run.energy) and not run.converged
    assert "did not settle in 1 passes" in run.errors[-1]
    assert calls == [3, 3]                               # bare pass + one embedded pass, no n-mer batch
