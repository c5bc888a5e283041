"""Latency of the literal drop-in: ONE fragment per mqc_hip_scf_run.  Prints ms per water dimer (RHF/cc-pVDZ, fresh
geometry per call) and per benzene SCF (B3LYP/cc-pVDZ, density-fitted), with the engine's own stage clocks.
Under `rocprofv3 --kernel-trace` the kernel timeline of a single call shows where the milliseconds go."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from metalquicha_amd import capi, mbe, methods

n_dimers = int(sys.argv[1]) if len(sys.argv) > 1 else 24
capi.get_context(0)
system = mbe.water_cluster(4)
terms = mbe.generate_mbe_term_list(system, 2)
st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-10, density_tol=1e-8, schwarz_tol=1e-12)
frags = [mbe.build_fragment(system, t) for t in [t for t in terms if len(t) == 2][:n_dimers]]
for f in frags[:2]:
    methods.run_hip_scf(st, f)
methods.get_stats()
t0 = time.perf_counter(); its = 0
for f in frags[2:]:
    r = methods.run_hip_scf(st, f); assert not r.has_error, r.error_message; its += r.scf_iterations
dt = time.perf_counter() - t0
s = methods.get_stats()
n = len(frags) - 2
print("dimer RHF/cc-pVDZ single call: %.3f ms per fragment, %.1f iterations each, %.0f it/s" % (1e3 * dt / n, its / n, its / dt))
print("  engine clocks per call (ms): setup %.3f int1e %.3f eri-enqueue %.3f scf-loop %.3f fetch %.3f total %.3f | kernels: eri %.3f jk %.3f scf_step %.3f"
      % tuple(1e3 * v / n for v in (s.t_setup, s.t_int1e, s.t_eri, s.t_fock, s.t_scf_step, s.t_total, s.eri_kernel_seconds, s.fock_kernel_seconds, s.scf_step_seconds)))

rcc, rch = 1.397, 1.084
sym = ["C"] * 6 + ["H"] * 6
xyz0 = np.array([[rcc * np.cos(np.pi / 3 * k), rcc * np.sin(np.pi / 3 * k), 0.0] for k in range(6)] +
                [[(rcc + rch) * np.cos(np.pi / 3 * k), (rcc + rch) * np.sin(np.pi / 3 * k), 0.0] for k in range(6)])
bst = methods.ScfSettings(basis_set="cc-pvdz", functional="b3lyp", density_fitting=True, aux_basis_set="mqc-even-tempered-jkfit",
                          energy_tol=1e-10, density_tol=1e-8, guess="gwh")
rng = np.random.default_rng(5)
for rep in range(3):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    fb = methods.PhysicalFragment.from_angstrom(sym, (xyz0 @ q.T).tolist())
    methods.get_stats()
    t0 = time.perf_counter(); rb = methods.run_hip_scf(bst, fb); dt = time.perf_counter() - t0
    s = methods.get_stats()
    assert not rb.has_error, rb.error_message
    print("benzene DF-B3LYP single call: %.2f ms, %d iterations | setup %.2f int1e %.2f eri-enqueue %.2f scf-loop %.2f fetch %.2f | kernels: df-build %.2f jk %.2f xc %.2f scf_step %.2f"
          % (1e3 * dt, rb.scf_iterations, 1e3 * s.t_setup, 1e3 * s.t_int1e, 1e3 * s.t_eri, 1e3 * s.t_fock, 1e3 * s.t_scf_step,
             1e3 * s.eri_kernel_seconds, 1e3 * s.fock_kernel_seconds, 1e3 * s.xc_kernel_seconds, 1e3 * s.scf_step_seconds))
capi.finalize()
