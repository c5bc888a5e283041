"""A BASELINE.json configs[4]-shaped run: 512 fragments, FMO2, density-fitted, point-charge (Mulliken) field -- here
(H2O)512 on the bench's lattice builder (8 x 8 x 8), cc-pVDZ + the even-tempered fitting set: 512 monomers per pass
(1533 charges each) and ALL 130 816 pairs (1530 charges each) as engine batches.
    python scripts/fmo_c5_probe.py [rhf|b3lyp] [n_side]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from metalquicha_amd import fmo, mbe, methods          # noqa: E402
from metalquicha_amd.methods import ScfSettings        # noqa: E402

functional = "" if (len(sys.argv) < 2 or sys.argv[1] == "rhf") else sys.argv[1]
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 8
system = mbe.water_cluster(n_side)
st = ScfSettings(basis_set="cc-pvdz", functional=functional, density_fitting=True, aux_basis_set="mqc-even-tempered-jkfit",
                 energy_tol=1e-9, density_tol=1e-7, guess="gwh")
clock = {"solve": 0.0, "calls": 0, "jobs": 0}
inner = fmo.hip_solver(system, st)


def solver(jobs):
    t0 = time.time()
    out = inner(jobs)
    dt = time.time() - t0
    clock["solve"] += dt; clock["calls"] += 1; clock["jobs"] += len(jobs)
    print("   batch of %6d fragments: %.2f s" % (len(jobs), dt), flush=True)
    return out


t = time.time()
run = fmo.run_fmo2(system, st, expansion="fmo", solver=solver)
dt = time.time() - t
st_ = methods.get_stats()
print("FMO2 %s/cc-pVDZ density-fitted, %d fragments, %d pairs: E = %.8f  outer passes %d  SCF iterations %d  errors %d" %
      (functional or "rhf", system.n_monomers, system.n_monomers * (system.n_monomers - 1) // 2, run.energy, run.outer_iterations,
       run.scf_iterations, len(run.errors)))
print("wall %.1f s (engine batch calls %.1f s in %d calls, %d SCFs; host bookkeeping %.1f s)  -> %.0f SCF iterations/s" %
      (dt, clock["solve"], clock["calls"], clock["jobs"], dt - clock["solve"], run.scf_iterations / dt))
if run.errors:
    print(run.errors[:3])
