"""FMO2 over several ranks (one process per GPU; here also usable with ranks sharing one GPU over gloo): every rank solves
the fragments / pairs with index = rank (mod world), one all-reduce per pass.  Launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/fmo_ranks.py [n_side] [backend]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402

from metalquicha_amd import fmo, mbe    # noqa: E402
from metalquicha_amd.methods import ScfSettings   # noqa: E402

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 4
backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
dist.init_process_group(backend, init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
local = int(os.environ.get("LOCAL_RANK", "0"))
device = local % max(torch.cuda.device_count(), 1)
on_gpu = backend == "nccl"
if on_gpu:
    torch.cuda.set_device(device)


def allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
    if on_gpu:
        t = t.cuda()
    dist.all_reduce(t)
    return t.cpu().numpy()


system = mbe.water_cluster(n_side)
st = ScfSettings(basis_set="cc-pvdz", energy_tol=1e-9, density_tol=1e-7, guess="gwh", device_rank=device)
fmo.run_fmo2(system, st, expansion="fmo", rank=rank, world=world, allreduce=allreduce)      # warm-up
dist.barrier()
t = time.time()
run = fmo.run_fmo2(system, st, expansion="fmo", rank=rank, world=world, allreduce=allreduce)
dist.barrier()
dt = time.time() - t
if rank == 0:
    print("FMO2 (point-charge field) of (H2O)%d over %d ranks (%s): E = %.10f  outer passes %d  %.3f s  errors %d" %
          (system.n_monomers, world, backend, run.energy, run.outer_iterations, dt, len(run.errors)), flush=True)
dist.destroy_process_group()
