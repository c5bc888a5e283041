"""Host logic that needs no GPU: basis reader, topology plumbing, MBE assembly, C-ABI exports,
the host build of the device integral templates, and the 2-rank gloo rehearsal of the N>1 path."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from metalquicha_amd import basis as basis_mod
from metalquicha_amd import capi, mbe, methods
from oracle import scf_oracle as so
from tests.helpers import fragment_bohr, oracle_mol


def test_basis_reader_splits_general_contractions_and_sp():
    o = basis_mod.read_element(basis_mod.find_basis_file("cc-pvdz"), 8)
    assert [s.l for s in o.shells] == [0, 0, 0, 1, 1, 2]          # 3s2p1d, one shell per row
    assert all(s.nprim == 9 for s in o.shells[:3])
    sto = basis_mod.read_element(basis_mod.find_basis_file("sto-3g"), 8)
    assert [s.l for s in sto.shells] == [0, 0, 1]                  # SP shell split on shared exponents
    assert np.array_equal(sto.shells[1].exps, sto.shells[2].exps)
    fb = basis_mod.build_flat_basis("cc-pvdz", [8, 1, 1])
    assert fb.nao == 24 and fb.nshell == 12 and list(fb.nshell_per_atom) == [6, 3, 3]
    with pytest.raises(basis_mod.BasisError):
        basis_mod.build_flat_basis("cc-pvdz", [3])                 # element not in the file
    with pytest.raises(basis_mod.BasisError):
        basis_mod.find_basis_file("no-such-basis")


def test_nelec_excludes_ghosts():
    f = methods.PhysicalFragment(np.array([8, 1, 1, 8, 1, 1]), np.zeros((3, 6)), ghost=[0, 0, 0, 1, 1, 1])
    assert f.nelec == 10


def test_capi_exports_every_declared_symbol():
    lib = capi.load_library()
    header = open(os.path.join(ROOT, "include", "mqc_hip.h")).read()
    declared = set(re.findall(r"\b(mqc_hip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.DECLARED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.mqc_hip_abi_version() == 3


def test_engine_fails_loudly_without_a_device():
    lib = capi.load_library()
    if lib.mqc_hip_backend_available():
        pytest.skip("a HIP device is present")
    h = ctypes.c_void_p()
    rc = lib.mqc_hip_context_get(0, ctypes.byref(h))
    assert rc == capi.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.mqc_hip_last_error()
    r = methods.run_hip_scf(methods.ScfSettings(), fragment_bohr([1, 1], [[0, 0, 0], [0, 0, 1.4]]))
    assert r.has_error and not r.has_energy


def test_struct_layouts_match_the_header():
    # sizes computed by the C compiler for the same declarations
    src = '#include "include/mqc_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu\\n",' \
          'sizeof(mqc_hip_molecule_t),sizeof(mqc_hip_basis_t),sizeof(mqc_hip_scf_options_t),' \
          'sizeof(mqc_hip_scf_result_t),sizeof(mqc_hip_stats_t));return 0;}'
    exe = os.path.join(ROOT, "tests", "host", "_sizes")
    subprocess.run(["gcc", "-x", "c", "-", "-I", ROOT, "-o", exe], input=src.encode(), cwd=ROOT, check=True)
    sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    os.remove(exe)
    assert sizes == [ctypes.sizeof(capi.Molecule), ctypes.sizeof(capi.Basis), ctypes.sizeof(capi.ScfOptions),
                     ctypes.sizeof(capi.ScfResult), ctypes.sizeof(capi.Stats)]


# ---- the gfx950 integral templates, compiled for the host by tests/host/build.sh --------------
CLASSES = [(0, 0, 0, 0), (1, 0, 0, 0), (1, 0, 1, 0), (1, 1, 0, 0), (1, 1, 1, 0), (1, 1, 1, 1), (2, 0, 0, 0),
           (2, 0, 1, 0), (2, 0, 1, 1), (2, 0, 2, 0), (2, 1, 0, 0), (2, 1, 1, 0), (2, 1, 1, 1), (2, 1, 2, 0),
           (2, 1, 2, 1), (2, 2, 0, 0), (2, 2, 1, 0), (2, 2, 1, 1), (2, 2, 2, 0), (2, 2, 2, 1), (2, 2, 2, 2)]


@pytest.fixture(scope="module")
def hostcheck():
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "host", "build.sh")], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(os.path.join(ROOT, "tests", "host", "libhostcheck.so"))


def test_device_boys_function_matches_oracle(hostcheck):
    dp = ctypes.POINTER(ctypes.c_double)
    for L in (0, 4, 8, 16):
        for T in (0.0, 1e-9, 0.03, 0.049999, 0.05, 0.77, 5.12345, 19.99, 33.3, 41.97, 42.0, 55.5, 300.0, 5000.0):
            F, G = np.zeros(L + 1), np.zeros(L + 1)
            hostcheck.hostcheck_boys(L, ctypes.c_double(T), F.ctypes.data_as(dp))
            so.lib().orc_boys(L, ctypes.c_double(T), G.ctypes.data_as(dp))
            assert np.max(np.abs(F - G) / np.maximum(np.abs(G), 1e-300)) < 2e-13, (L, T)


@pytest.mark.parametrize("cls", CLASSES)
def test_device_eri_class_matches_oracle(hostcheck, cls):
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    rng = np.random.default_rng(sum(c * 7 ** k for k, c in enumerate(cls)) + 1)
    l = np.array(cls, dtype=np.int32)
    nprim = rng.integers(1, 4, size=4).astype(np.int32)
    exps = rng.uniform(0.2, 4.0, size=int(nprim.sum())); exps[0] = 30.0
    coefs = rng.uniform(-1, 1, size=int(nprim.sum()))
    xyz = rng.uniform(-1.5, 1.5, size=(4, 3))
    mol = so.make_mol([1, 1, 1, 1], xyz, [1, 1, 1, 1], l, nprim, exps, coefs)
    eri = so.eri4(mol)
    off, ns = mol.sh_aoff, 2 * l + 1
    ref = eri[off[0]:off[0] + ns[0], off[1]:off[1] + ns[1], off[2]:off[2] + ns[2], off[3]:off[3] + ns[3]]
    fac = np.array([0.282094791773878143, 0.488602511902919921, 1.0, 1.0, 1.0])
    co = mol.coefs.copy(); o = 0
    for k in range(4):
        co[o:o + nprim[k]] *= fac[l[k]]; o += nprim[k]
    out = np.zeros(int(np.prod(ns)))
    rc = hostcheck.hostcheck_eri_block(l.ctypes.data_as(ip), nprim.ctypes.data_as(ip), exps.ctypes.data_as(dp),
                                       co.ctypes.data_as(dp), np.ascontiguousarray(xyz).ctypes.data_as(dp), out.ctypes.data_as(dp))
    assert rc == 0
    assert np.max(np.abs(out.reshape(ref.shape) - ref)) < 1e-13 * max(1.0, np.max(np.abs(ref)))


TWIN_CLASSES = [(0, 0, 0, 0), (1, 0, 0, 0), (1, 0, 1, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 0, 0, 0), (2, 0, 1, 0), (2, 1, 0, 0)]


@pytest.mark.parametrize("cls", TWIN_CLASSES)
def test_device_twin_block_equals_segmented_blocks(hostcheck, cls):
    """Twin s shells (two contractions over the same primitives): one pass over the primitive quartets must
    give exactly the blocks the segmented routine gives member by member; absent members stay zero."""
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    rng = np.random.default_rng(100 + sum(c * 5 ** k for k, c in enumerate(cls)))
    l = np.array(cls, dtype=np.int32)
    for trial in range(3):
        nprim = rng.integers(1, 6, size=4).astype(np.int32)
        n = int(nprim.sum())
        exps = rng.uniform(0.2, 6.0, size=n)
        c1, c2 = rng.uniform(-1, 1, size=n), rng.uniform(-1, 1, size=n)
        twin = (rng.integers(0, 2, size=4) if trial else np.ones(4)).astype(np.int32)
        xyz = rng.uniform(-1.5, 1.5, size=(4, 3))
        worst = ctypes.c_double(-1.0)
        rc = hostcheck.hostcheck_eri_twin(l.ctypes.data_as(ip), nprim.ctypes.data_as(ip), exps.ctypes.data_as(dp),
                                          c1.ctypes.data_as(dp), c2.ctypes.data_as(dp), twin.ctypes.data_as(ip),
                                          np.ascontiguousarray(xyz).ctypes.data_as(dp), ctypes.byref(worst))
        assert rc == 0
        assert 0.0 <= worst.value < 1e-13, (cls, trial, worst.value)


@pytest.mark.parametrize("zs,basis,expect_twins", [([8, 1, 1], "cc-pvdz", 1), ([8, 1, 1, 8, 1, 1], "cc-pvdz", 2),
                                                  ([6, 7, 8, 1], "cc-pvdz", 3), ([8, 1, 1], "sto-3g", 0)])
def test_topology_twin_cut_covers_every_quartet_once(hostcheck, zs, basis, expect_twins):
    """cc-pVDZ C/N/O: the 1s/2s functions are twins (same 9 primitives).  Twin entries expanded over their
    members plus the uncovered rest must be exactly the canonical quartet set of each class, nothing twice."""
    from tests import stages
    from tests.helpers import fragment_bohr
    rng = np.random.default_rng(3)
    frag = fragment_bohr(zs, rng.uniform(-3, 3, size=(len(zs), 3)))
    m = stages._marshal(basis, frag)
    ntw, nent, nbad = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
    rc = hostcheck.hostcheck_twin_cut(ctypes.byref(m.mol), ctypes.byref(m.bas), ctypes.byref(ntw), ctypes.byref(nent), ctypes.byref(nbad))
    assert rc == 0
    assert ntw.value == expect_twins
    assert nbad.value == 0
    assert (nent.value > 0) == (expect_twins > 0)


# ---- MBE assembly ----------------------------------------------------------------------------
def test_mbe_term_list_and_coefficients():
    system = mbe.water_cluster(2)
    assert system.n_monomers == 8
    terms = mbe.generate_mbe_term_list(system, 2)
    assert len(terms) == 8 + 28
    assert all(len(t) == 2 for t in terms[:28])                   # largest first
    c = mbe.compute_mbe_coefficients(terms)
    for t, ci in zip(terms, c):
        assert ci == (1.0 if len(t) == 2 else 2.0 - 8)            # c_dimer = 1, c_monomer = 2 - N
    rng = np.random.default_rng(1)
    e = rng.normal(size=len(terms))
    total, by_order, delta = mbe.compute_mbe(terms, e)
    assert abs(total - float(np.dot(c, e))) < 1e-12               # delta form == coefficient form
    perm = rng.permutation(len(terms))                            # order independence (test_mqc_mbe.f90:18-22)
    total2, _, _ = mbe.compute_mbe([terms[i] for i in perm], e[perm])
    assert abs(total - total2) < 1e-12


def test_mbe_distance_screening():
    system = mbe.water_cluster(3)
    all_terms = mbe.generate_mbe_term_list(system, 2)
    cut = mbe.generate_mbe_term_list(system, 2, cutoffs={2: 3.0})
    assert len(all_terms) == 27 + 27 * 26 // 2
    assert 27 < len(cut) < len(all_terms)
    for t in cut:
        if len(t) == 2:
            assert mbe.min_intermonomer_distance(system, *t) <= 3.0


def test_water_cluster_is_reproducible():
    a, b = mbe.water_cluster(4), mbe.water_cluster(4)
    assert a.n_monomers == 64 and np.array_equal(a.coordinates, b.coordinates)
    roh = np.linalg.norm(a.coordinates[:, 0] - a.coordinates[:, 1]) * mbe.BOHR_TO_ANGSTROM
    assert abs(roh - 0.9592) < 1e-3                               # rigid w1.xyz geometry


def test_partition_covers_every_term_once():
    for world in (1, 2, 3, 8):
        owned = np.concatenate([mbe.partition_terms(2080, r, world) for r in range(world)])
        assert sorted(owned.tolist()) == list(range(2080))


def test_two_rank_gloo_rehearsal_of_the_energy_reduction(tmp_path):
    """world_size = 2 over gloo: each rank fills the energies of ITS terms (stand-in values), one
    all-reduce of the zero-padded vector, MBE assembly on every rank; must equal the serial result."""
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "from metalquicha_amd import mbe\n"
        "dist.init_process_group('gloo', init_method='env://')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "system = mbe.water_cluster(2)\n"
        "terms = mbe.generate_mbe_term_list(system, 2)\n"
        "ref = np.array([-76.0 * len(t) - 1e-3 * sum(t) for t in terms])\n"
        "e = np.zeros(len(terms)); own = mbe.partition_terms(len(terms), r, w); e[own] = ref[own]\n"
        "buf = torch.from_numpy(e); dist.all_reduce(buf)\n"
        "tot, _, _ = mbe.compute_mbe(terms, buf.numpy())\n"
        "ser, _, _ = mbe.compute_mbe(terms, ref)\n"
        "assert abs(tot - ser) < 1e-10, (tot, ser)\n"
        "print('rank', r, 'ok', tot)\n"
        "dist.destroy_process_group()\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_fortran_drop_in_bridge_compiles_and_runs():
    """fortran/mqc_hip_bridge.f90 (module mqc_cuest_bridge: run_cuest_scf, run_cuest_scf_batch,
    cuest_backend_available) and fortran/mqc_hip_node_worker.f90 compiled with flang against stand-ins of the
    metalquicha modules they use, linked with libmqc_hip.so and RUN.  Without a GPU the engine has no fallback, so the
    programs must report the "no HIP device" error per fragment; on a GPU box the same executables produce energies
    (tests/test_gpu_fortran.py)."""
    flang = "/opt/rocm/lib/llvm/bin/flang"
    if not os.path.isfile(flang):
        pytest.skip("no flang in this image")
    out = subprocess.run(["bash", os.path.join(ROOT, "fortran", "check_bridge.sh")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "backend available (property of the binary) T" in out.stdout       # pure, like the reference's
    from metalquicha_amd import capi
    if not capi.load_library().mqc_hip_backend_available():
        assert "CHECK PASS no device: the call fails loudly" in out.stdout
        assert "SUMMARY no_device failures 0" in out.stdout
    env = dict(os.environ, MQC_FLAT_BASIS_PATH=os.path.join(ROOT, "fortran", "_build", "basis"))
    nw = subprocess.run([os.path.join(ROOT, "fortran", "_build", "check_node_worker")], capture_output=True, text=True, env=env, timeout=300)
    assert nw.returncode == 0, nw.stdout + nw.stderr
    assert "CHECK PASS every task has exactly one result" in nw.stdout and "SUMMARY failures 0" in nw.stdout
    src = open(os.path.join(ROOT, "fortran", "mqc_hip_bridge.f90")).read()
    assert "c_loc(aux)" in src and "settings%aux_basis_set" in src
    assert "pure function cuest_backend_available" in src                     # mqc_cuest_bridge.f90:20


def test_node_worker_patch_applies_to_the_reference_file(tmp_path):
    """fortran/patches/node_worker_batch.patch is a unified diff against the reference's distribution scheme; where the
    reference tree is present (this container, not the GPU box) it must apply cleanly to a scratch copy."""
    import shutil
    ref = "/root/reference/src/fragmentation/mbe/mqc_mbe_mpi_fragment_distribution_scheme.F90"
    patch = os.path.join(ROOT, "fortran", "patches", "node_worker_batch.patch")
    assert os.path.isfile(patch)
    if not os.path.isfile(ref) or shutil.which("patch") is None:
        pytest.skip("no reference tree (or no patch tool) here")
    work = tmp_path / "scheme.F90"
    shutil.copyfile(ref, work)
    out = subprocess.run(["patch", "--no-backup-if-mismatch", str(work), patch], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    text = work.read_text()
    assert "fragment_batch_t" in text and "outstanding_fifo_t" in text and "run_cuest_scf_batch" not in open(ref).read()


def test_fortran_iso_c_binding_module_links():
    """fortran/mqc_hip_c.f90 compiled with AMD flang and linked against libmqc_hip.so."""
    flang = "/opt/rocm/lib/llvm/bin/flang"
    if not os.path.isfile(flang):
        pytest.skip("no flang in this image")
    out = subprocess.run(["bash", os.path.join(ROOT, "fortran", "check_link.sh")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi version 3" in out.stdout


def test_vectorised_marshalling_layouts_match_ctypes():
    import ctypes as C
    assert methods._MOL_DTYPE.itemsize == C.sizeof(capi.Molecule)
    assert methods._BAS_DTYPE.itemsize == C.sizeof(capi.Basis)
    for name in methods._MOL_DTYPE.names:
        assert getattr(capi.Molecule, name).offset == methods._MOL_DTYPE.fields[name][1], name
    for name in methods._BAS_DTYPE.names:
        assert getattr(capi.Basis, name).offset == methods._BAS_DTYPE.fields[name][1], name


# ---- vectorised MBE marshalling ------------------------------------------------------------------
def test_fragment_groups_equal_fragment_by_fragment_construction():
    """build_fragment_groups gathers whole MBE orders with one fancy index; every fragment must be exactly
    what build_fragment makes for the same term (elements, coordinates bit for bit, charge)."""
    system = mbe.water_cluster(3)
    terms = mbe.generate_mbe_term_list(system, 2)
    groups, positions = mbe.build_fragment_groups(system, terms)
    seen = np.zeros(len(terms), dtype=int)
    for grp, pos in zip(groups, positions):
        assert grp.xyz.shape == (len(pos), len(grp.element_numbers), 3)
        for k in range(0, len(pos), 7):
            f = mbe.build_fragment(system, terms[pos[k]])
            assert np.array_equal(f.element_numbers, grp.element_numbers)
            assert np.array_equal(f.coordinates.T, grp.xyz[k])
            assert int(grp.charge[k]) == f.charge
        seen[pos] += 1
    assert np.all(seen == 1)          # every term in exactly one group


def test_result_record_view_matches_the_ctypes_struct():
    """methods._RES_DTYPE is a numpy view of capi.ScfResult: same size, same field offsets."""
    dt = methods._RES_DTYPE
    assert dt.itemsize == ctypes.sizeof(capi.ScfResult)
    for name, _ in capi.ScfResult._fields_:
        assert dt.fields[name][1] == getattr(capi.ScfResult, name).offset
    arr = (capi.ScfResult * 3)()
    arr[1].e_total = -1.5; arr[1].iterations = 7; arr[2].has_error = 1; arr[2].message = b"boom"
    rec = np.frombuffer(arr, dtype=dt)
    assert rec["e_total"][1] == -1.5 and rec["iterations"][1] == 7
    assert rec["has_error"][2] == 1 and bytes(rec["message"][2]).split(b"\0", 1)[0] == b"boom"


# ---- GMBE caller mirror (src/fragmentation/gmbe/mqc_gmbe_utils.f90) ----------------------------------------------
def test_gmbe_pie_terms_reduce_to_mbe_and_count_every_atom_once():
    from metalquicha_amd import gmbe
    # non-overlapping monomers, level 2: dimers +1, monomers -(M - 2): the MBE(2) formula
    mono = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (9, 10, 11)]
    prim = [tuple(sorted(mono[i] + mono[j])) for i in range(4) for j in range(i + 1, 4)]
    sets, coef = gmbe.enumerate_pie_terms(prim)
    table = dict(zip(sets, coef))
    assert all(table[p] == 1 for p in prim)
    assert all(table[m] == -2 for m in mono)
    assert len([c for c in coef if c != 0]) == 10
    # an overlapping chain at level 1
    sets, coef = gmbe.enumerate_pie_terms([(0, 1, 2, 3, 4, 5), (3, 4, 5, 6, 7, 8)])
    assert dict(zip(sets, coef)) == {(0, 1, 2, 3, 4, 5): 1, (3, 4, 5, 6, 7, 8): 1, (3, 4, 5): -1}
    # inclusion-exclusion: every atom of the union is counted exactly once, whatever the overlaps
    rng = np.random.default_rng(3)
    for _ in range(20):
        prim = [tuple(sorted(rng.choice(12, size=rng.integers(2, 7), replace=False).tolist())) for _ in range(rng.integers(2, 7))]
        prim = list(dict.fromkeys(prim))
        sets, coef = gmbe.enumerate_pie_terms(prim)
        union = set(a for p_ in prim for a in p_)
        for a in union:
            assert sum(int(c) for s_, c in zip(sets, coef) if a in s_) == 1
    # depth limit: pairs only
    sets, coef = gmbe.enumerate_pie_terms([(0, 1), (0, 2), (0, 3)], max_k_level=2)
    assert dict(zip(sets, coef))[(0,)] == -3


def test_gmbe_primaries_and_polymer_atoms():
    from metalquicha_amd import gmbe, mbe
    system = mbe.water_cluster(2)
    assert gmbe.generate_primaries(system, 1) == [(i,) for i in range(8)]
    assert len(gmbe.generate_primaries(system, 2)) == 28
    assert len(gmbe.generate_primaries(system, 2, {2: 4.0})) < 28
    # overlapping base fragments: shared atoms appear once
    ov = mbe.FragmentedSystem(system.element_numbers, system.coordinates, [np.arange(0, 6), np.arange(3, 9)])
    assert gmbe.polymer_atoms(ov, (0, 1)) == tuple(range(9))



# ---- cost-aware distribution of the term list (mbe.partition_terms_lpt, mbe.PullQueue) ---------------------------------
def test_lpt_partition_bounds_the_imbalance_of_a_mixed_list():
    """A mixed list -- a few large subsystems, many medium and small ones (monomers next to dimers next to def2-TZVP
    dimers) -- under static round-robin over the sorted list lands two large terms on one rank; longest-processing-time-
    first keeps max load - mean load below the largest single cost and the makespan within 4/3 of the lower bound."""
    from metalquicha_amd import mbe
    costs = np.array([100.0] * 5 + [10.0] * 20 + [1.0] * 40)
    for world in (2, 3, 4, 8):
        rr = mbe.partition_loads(costs, world, "round_robin")
        lpt = mbe.partition_loads(costs, world, "lpt")
        assert abs(rr.sum() - costs.sum()) < 1e-9 and abs(lpt.sum() - costs.sum()) < 1e-9
        lower = max(costs.sum() / world, costs.max())
        assert lpt.max() - lpt.mean() <= costs.max() + 1e-9
        assert lpt.max() <= (4.0 / 3.0) * lower + 1e-9
        assert lpt.max() <= rr.max() + 1e-9
        owned = [mbe.partition_terms_lpt(costs, r, world) for r in range(world)]
        assert sorted(int(i) for o in owned for i in o) == list(range(len(costs)))       # every term exactly once
    assert mbe.partition_loads(costs, 4, "round_robin").max() >= 1.25 * mbe.partition_loads(costs, 4, "lpt").max()      # 260 against 200
    # the model behind it: a cc-pVDZ water dimer (n = 48) costs 16 x a monomer (n = 24) on the in-core path
    system = mbe.water_cluster(2)
    terms = mbe.generate_mbe_term_list(system, 2)
    c = mbe.term_costs(system, terms, "cc-pvdz")
    assert abs(c[0] / c[-1] - 16.0) < 1e-9 and len(c) == 36
    assert mbe.term_costs(system, terms, "cc-pvdz", functional="b3lyp")[0] > c[0]


def test_pull_queue_slices_are_disjoint_and_cover_the_list():
    """The queue's claim is one atomic add on a shared counter: whatever chunk sizes the ranks ask for, the slices are
    disjoint and cover [0, n).  Guided chunks: large while the list is long, min_chunk at the tail."""
    import threading
    from metalquicha_amd import mbe
    n, world = 1000, 4
    lock = threading.Lock(); counter = [0]

    def add(k):
        with lock:
            counter[0] += k
            return counter[0]
    queues = [mbe.PullQueue(n, world, add, min_chunk=3) for _ in range(world)]
    got = [[] for _ in range(world)]

    def body(r):
        while True:
            sl = queues[r].draw()
            if sl is None:
                return
            got[r].append(sl)
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    slices = sorted(s for g in got for s in g)
    assert slices[0][0] == 0 and slices[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(slices, slices[1:]))
    sizes = [b - a for a, b in slices]
    assert max(sizes) == 125 and min(sizes[:-1]) >= 3


def test_two_rank_gloo_pull_queue_runs_every_term_once(tmp_path):
    """world_size = 2 over gloo: both ranks draw from ONE queue (atomic add on a TCPStore) with an injected solver that
    takes longer for costlier terms; every term is evaluated exactly once, the one all-reduce delivers the serial
    energies on both ranks, and the slower-fed rank is not starved."""
    script = tmp_path / "rank.py"
    script.write_text(
        "import sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "from metalquicha_amd import mbe\n"
        "from metalquicha_amd.methods import ScfSettings\n"
        "dist.init_process_group('gloo', init_method='env://')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "store = dist.TCPStore('127.0.0.1', 29541, w, r == 0)\n"
        "system = mbe.water_cluster(2)\n"
        "terms = mbe.generate_mbe_term_list(system, 2)\n"
        "costs = mbe.term_costs(system, terms, 'cc-pvdz')\n"
        "def fake(groups):\n"
        "    recs = []\n"
        "    for g in groups:\n"
        "        m = g.xyz.shape[0]\n"
        "        rec = np.zeros(m, dtype=[('has_error', 'i4'), ('e_total', 'f8'), ('iterations', 'i4'), ('message', 'S8')])\n"
        "        rec['e_total'] = -np.linalg.norm(g.xyz.reshape(m, -1), axis=1) - len(g.element_numbers)\n"
        "        rec['iterations'] = 7\n"
        "        time.sleep(0.002 * m * (len(g.element_numbers) / 3) ** 2 * (2 if r == 1 else 1))\n"
        "        recs.append(rec)\n"
        "    return recs\n"
        "q = mbe.PullQueue(len(terms), w, lambda k: store.add('mbe_next', k), min_chunk=2)\n"
        "run = mbe.run_mbe_pull(system, ScfSettings(basis_set='cc-pvdz'), terms, q, costs=costs, evaluate=fake)\n"
        "owned = torch.zeros(len(terms), dtype=torch.float64); owned[run.owned] = 1.0\n"
        "e = torch.from_numpy(run.energies.copy()); dist.all_reduce(e); dist.all_reduce(owned)\n"
        "assert bool((owned == 1.0).all()), owned\n"
        "groups, pos = mbe.build_fragment_groups(system, terms)\n"
        "ser = np.zeros(len(terms))\n"
        "for p, rec in zip(pos, fake(groups)): ser[p] = rec['e_total']\n"
        "assert np.max(np.abs(e.numpy() - ser)) < 1e-12\n"
        "assert 0 < len(run.owned) < len(terms)\n"
        "print('rank', r, 'ok', len(run.owned))\n"
        "dist.barrier(); dist.destroy_process_group()\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29539")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29539", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_no_launched_kernel_exceeds_the_scratch_bound():
    """ROCm reserves scratch per hardware queue for a full device of waves; the engine's HBM budget leaves
    n_queues x SCRATCH_BOUND_PER_LANE x 64 x (CUs x 32) alone (engine.cpp: scratch_reservation_bytes).  That promise holds
    only while no kernel the dispatchers launch carries more: read the private-segment sizes from the code objects of
    the built library (scripts/scratch_report.py) and check them against the routing rules of kern_eri.hip
    (class_is_general: four-centre pass classes of total angular momentum >= 7 and the (dd| Schwarz bounds go through
    the LDS kernel and are never launched)."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import scratch_report
    if not os.path.isfile("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump here")
    src = open(os.path.join(ROOT, "metalquicha_amd", "csrc", "engine.cpp")).read()
    bound = int(re.search(r"SCRATCH_BOUND_PER_LANE = (\d+);", src).group(1))
    ks = scratch_report.kernels()
    assert len(ks) > 150
    never_launched, worst = [], (0, "")
    for name, private, vgpr, sgpr, lds in ks:
        base, args = scratch_report.pretty(name)
        routed_to_lds_kernel = (base in ("eri_kernel", "eri_pass_kernel", "eri_twin_kernel", "eri_digest_kernel") and sum(args[:4]) >= 7) \
            or (base in ("schwarz_pass_kernel", "schwarz_kernel") and args[:2] == [2, 2])
        if routed_to_lds_kernel:
            never_launched.append((base, args, private))
            continue
        if private > worst[0]:
            worst = (private, "%s%s" % (base, args))
        assert private <= bound, (base, args, private)
    assert worst[0] > 0 and any(p > bound for _, _, p in never_launched)      # the rule is what keeps the big ones out
    # the LDS kernels that take those classes have no private segment at all
    for name, private, *_ in ks:
        if scratch_report.pretty(name)[0].startswith(("eri_general_kernel", "schwarz_general")):
            assert private == 0, name
