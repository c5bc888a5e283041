#!/usr/bin/env python3
"""Measurement aid: the engine's one-matrix eigen-solver (mqc_hip_syev: one workgroup, cyclic Jacobi) on random symmetric
matrices of the sizes the single-fragment path meets -- 48 (A and V in LDS), 114 (benzene: V rotated in global memory),
140, 200 (both in global memory)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import stages
for n in (48, 96, 114, 140, 200):
    rng = np.random.default_rng(n)
    a = rng.normal(size=(n, n)); a = 0.5 * (a + a.T)
    w, v = stages.syev(a)
    for _ in range(2): stages.syev(a)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): w, v = stages.syev(a)
    dt = (time.perf_counter() - t0) / reps
    err = np.max(np.abs(w - np.linalg.eigvalsh(a)))
    print("n = %3d: %.2f ms per decomposition (host round trip included), max |eigenvalue error| %.1e" % (n, 1e3 * dt, err))
