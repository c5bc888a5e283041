#!/usr/bin/env python3
"""Measurement aid: wall time of ONE rank's share of the (H2O)64 MBE-2 evaluation at world size W (static
round-robin partition), on a single GPU -- what each rank of an N-GPU job does per evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalquicha_amd import mbe, methods
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
system = mbe.water_cluster(4)
terms = mbe.generate_mbe_term_list(system, 2)
st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-8, density_tol=1e-6, schwarz_tol=1e-12)
for _ in range(2):
    mbe.run_mbe(system, st, level=2, rank=0, world=W, terms=terms)
t0 = time.perf_counter()
n = 5
for _ in range(n):
    r = mbe.run_mbe(system, st, level=2, rank=0, world=W, terms=terms)
dt = (time.perf_counter() - t0) / n
print("world %d: rank-0 share %d fragments, %.1f ms per evaluation (ideal %.1f ms at 131 ms for the whole job)"
      % (W, len(r.owned), 1e3 * dt, 131.0 / W))
