import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
from metalquicha_amd import capi, mbe, methods
capi.get_context(0)
system = mbe.water_cluster(4); terms = mbe.generate_mbe_term_list(system, 2)
st = methods.ScfSettings(basis_set="cc-pvdz", guess="gwh", energy_tol=1e-10, density_tol=1e-8, schwarz_tol=1e-12)
dim = [t for t in terms if len(t) == 2][:26]
mono = [t for t in terms if len(t) == 1][:10]
def singles(tag, tl):
    frags = [mbe.build_fragment(system, t) for t in tl]
    for f in frags[:2]: methods.run_hip_scf(st, f)
    methods.get_stats()
    t0 = time.perf_counter()
    for f in frags[2:]: methods.run_hip_scf(st, f)
    dt = time.perf_counter() - t0
    s = methods.get_stats(); n = len(frags) - 2
    print(tag, "%.3f ms per fragment | setup %.3f int1e %.3f eri-enq %.3f scf-loop %.3f fetch %.3f total %.3f" % ((1e3*dt/n,) + tuple(1e3*v/n for v in (s.t_setup, s.t_int1e, s.t_eri, s.t_fock, s.t_scf_step, s.t_total))))
singles("fresh dimers  ", dim)
singles("fresh monomers", mono)
run = mbe.run_mbe(system, st, level=2, terms=terms)
singles("after big batch dimers  ", dim)
singles("after big batch monomers", mono)
capi.finalize()
