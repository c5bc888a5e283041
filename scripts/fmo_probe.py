"""Times the embedded callers on the bench's cluster: (H2O)64 RHF/cc-pVDZ, Mulliken point-charge field (esp = "ptc").
EE-MBE2 and FMO2 through fmo.run_fmo2: 64 monomers per pass, 2016 pairs with 186 charges each in the pair phase.
    python scripts/fmo_probe.py [n_side] [exact]      -- "exact": FMO2 with the exact Coulomb field of the near fragments too"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from metalquicha_amd import fmo, mbe          # noqa: E402
from metalquicha_amd.methods import ScfSettings   # noqa: E402

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 4
system = mbe.water_cluster(n_side)
st = ScfSettings(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh")
plain = mbe.run_mbe(system, ScfSettings(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh"), level=2)
e_mbe = mbe.compute_mbe(plain.terms, plain.energies)[0]
for expansion in ("mbe", "fmo", "mbe", "fmo"):
    t = time.time()
    run = fmo.run_fmo2(system, st, expansion=expansion)
    dt = time.time() - t
    print("%-3s  E = %.10f  (plain MBE2 %.10f)  outer passes %d  SCF iterations %d  %.3f s  errors %d" %
          (expansion, run.energy, e_mbe, run.outer_iterations, run.scf_iterations, dt, len(run.errors)), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "exact":
    inner = fmo.hip_cross_coulomb(system, st)
    clock = {"coulomb": 0.0, "requests": 0}

    def timed_coulomb(requests):
        t0 = time.time()
        out = inner(requests)
        clock["coulomb"] += time.time() - t0; clock["requests"] += len(requests)
        return out

    for rep in range(2):
        clock["coulomb"] = 0.0; clock["requests"] = 0
        t = time.time()
        run = fmo.run_fmo2(system, st, expansion="fmo", esp="exact", coulomb=timed_coulomb)
        dt = time.time() - t
        print("           Coulomb batches: %d supersystems, %.3f s of the %.3f s" % (clock["requests"], clock["coulomb"], dt))
        near = [len(fmo.near_fragments(system, [i], 2.0)) for i in range(system.n_monomers)]
        print("fmo exact  E = %.10f  outer passes %d  SCF iterations %d  near fragments per monomer %.1f  %.3f s  errors %d" %
              (run.energy, run.outer_iterations, run.scf_iterations, sum(near) / len(near), dt, len(run.errors)), flush=True)
