#!/usr/bin/env python3
"""Measurement aid: (H2O)8 cc-pVDZ (n = 192) RHF on the direct path (too wide for the in-core tensor), without and with
the incremental Fock build (MQC_HIP_DIRECT_INCREMENTAL=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalquicha_amd import mbe, methods
system = mbe.water_cluster(2)
frag = mbe.build_fragment(system, tuple(range(8)))
st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-10, density_tol=1e-8)
for rep in range(2):
    t0 = time.perf_counter()
    r = methods.run_hip_scf(st, frag)
    dt = time.perf_counter() - t0
    print("incremental=%s n_ao=%d E=%.10f iterations=%d %.2f s%s" % (os.environ.get("MQC_HIP_DIRECT_INCREMENTAL", "0"), 192, r.energy.scf, r.scf_iterations, dt,
                                                                  " error: " + r.error_message if r.has_error else ""))
