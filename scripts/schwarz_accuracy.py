"""Energy effect of the in-core Schwarz threshold on a spread of (H2O)64 dimers (GPU)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from metalquicha_amd import mbe, methods
system = mbe.water_cluster(4)
terms = [t for t in mbe.generate_mbe_term_list(system, 2) if len(t) == 2]
sel = terms[:: max(1, len(terms) // 256)][:256]
frags = [mbe.build_fragment(system, t) for t in sel]
ref = None
for tol in (0.0, 1e-14, 1e-13, 1e-12, 1e-11):
    st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-10, density_tol=1e-8, schwarz_tol=tol)
    methods.run_hip_scf_batch(st, frags)
    methods.get_stats()
    t = time.time(); res = methods.run_hip_scf_batch(st, frags); dt = time.time() - t
    s = methods.get_stats()
    e = np.array([r.energy.scf for r in res]); it = sum(r.scf_iterations for r in res)
    if ref is None: ref = e
    print("schwarz_tol %-7g  max|dE| %.2e  sum|dE| %.2e  iters %d  wall %.3f s  eri %.3f s" % (tol, np.max(np.abs(e - ref)), np.sum(np.abs(e - ref)), it, dt, s.t_eri), flush=True)
