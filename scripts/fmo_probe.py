"""Times the embedded callers on the bench's cluster: (H2O)64 RHF/cc-pVDZ, Mulliken point-charge field (esp = "ptc").
EE-MBE2 and FMO2 through fmo.run_fmo2: 64 monomers per pass, 2016 pairs with 186 charges each in the pair phase.
    python scripts/fmo_probe.py [n_side]"""
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
