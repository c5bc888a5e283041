"""The boundary from the FORTRAN side, on the GPU: the prebuilt fortran/_build/check_bridge (module mqc_cuest_bridge of
fortran/mqc_hip_bridge.f90 linked with libmqc_hip.so, built by __graft_entry__.build() / fortran/check_bridge.sh) run as
a fresh child process -- the way hf_run / dft_run reach the engine -- must reproduce the reference's check_rhf golden
-74.9658162796 (validation/check_rhf.f90:142), serve the second call from its shell cache, deliver a gradient, take a
density-fitted run with its auxiliary basis, survive more (basis, element sequence) entries than cache slots, and run
run_cuest_scf_batch; fortran/_build/check_node_worker runs the batched worker protocol on real energies."""
import os
import re
import subprocess

import numpy as np
import pytest

from metalquicha_amd import methods
from tests.helpers import fragment_bohr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "fortran", "_build")


def _run(exe):
    path = os.path.join(BUILD, exe)
    if not os.path.isfile(path):
        if not os.path.isfile("/opt/rocm/lib/llvm/bin/flang"):
            pytest.skip("no flang here and no prebuilt %s" % exe)
        subprocess.check_call(["bash", os.path.join(ROOT, "fortran", "check_bridge.sh"), "--no-run"])
    env = dict(os.environ, MQC_FLAT_BASIS_PATH=os.path.join(BUILD, "basis"))
    return subprocess.run([path], capture_output=True, text=True, env=env, timeout=900)


def _energies(stdout):
    out = {}
    for line in stdout.splitlines():
        m = re.match(r"ENERGY (\S+)\s+(-?\d+\.\d+)", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def test_fortran_bridge_reproduces_check_rhf_and_uses_its_cache():
    out = _run("check_bridge")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device visible T" in out.stdout
    assert "CHECK FAIL" not in out.stdout and "SUMMARY failures 0" in out.stdout, out.stdout
    for needle in ("h2o sto-3g == check_rhf golden (1e-9)", "second call: basis reader not called",
                   "second call: no cache miss", "gradient rows sum to zero (1e-8)",
                   "cc-pvdz density-fitted runs (auxiliary basis through the bridge)",
                   "cache stress: evictions happened (more entries than slots)",
                   "cache stress: second cycle == first cycle (1e-10)", "batch: energies == single calls (1e-10)",
                   "batch: the fragment without a basis fails alone"):
        assert "CHECK PASS " + needle in out.stdout, needle
    e = _energies(out.stdout)
    assert abs(e["h2o_sto3g"] - (-74.9658162796)) < 1e-9
    # the same numbers through the Python mirror of the interface (ctypes into the same library)
    frag = fragment_bohr([8, 1, 1], [[0.0, 0.0, -0.1364652], [0.0, 1.4304924, 1.0826636], [0.0, -1.4304924, 1.0826636]])
    st = methods.ScfSettings(basis_set="cc-pvdz", density_fitting=True, aux_basis_set="mqc-even-tempered-jkfit",
                             energy_tol=1e-10, density_tol=1e-8, guess="gwh")
    r = methods.run_hip_scf(st, frag)
    assert not r.has_error, r.error_message
    assert abs(r.energy.scf - e["h2o_ccpvdz_df"]) < 1e-10


def test_fortran_batched_worker_protocol_on_real_energies():
    out = _run("check_node_worker")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CHECK FAIL" not in out.stdout and "SUMMARY failures 0" in out.stdout, out.stdout
    assert "CHECK PASS batched results == single calls, task by task (1e-10)" in out.stdout
