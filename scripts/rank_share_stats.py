#!/usr/bin/env python3
"""Measurement aid: where one rank's share of the (H2O)64 MBE-2 evaluation spends its time at world size W (engine
stage clocks of the last evaluation: host wall per stage and HIP-event kernel seconds)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalquicha_amd import capi, mbe, methods
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
system = mbe.water_cluster(4)
terms = mbe.generate_mbe_term_list(system, 2)
st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-8, density_tol=1e-6, schwarz_tol=1e-12)
for _ in range(2):
    mbe.run_mbe(system, st, level=2, rank=0, world=W, terms=terms)
methods.get_stats()          # reading resets the counters
t0 = time.perf_counter()
r = mbe.run_mbe(system, st, level=2, rank=0, world=W, terms=terms)
dt = time.perf_counter() - t0
s = methods.get_stats()
print("world %d share %d fragments: %.1f ms | host: setup %.1f int1e %.1f eri %.1f fock+loop %.1f total %.1f | kernels: eri %.1f jk %.1f (%d launches) scf_step %.1f | iterations %d"
      % (W, len(r.owned), 1e3 * dt, 1e3 * s.t_setup, 1e3 * s.t_int1e, 1e3 * s.t_eri, 1e3 * s.t_fock, 1e3 * s.t_total,
         1e3 * s.eri_kernel_seconds, 1e3 * s.fock_kernel_seconds, s.fock_launches, 1e3 * s.scf_step_seconds, s.scf_iterations_total))
