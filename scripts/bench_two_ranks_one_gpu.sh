#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a one-GPU box: two ranks share device 0, gloo carries the gather.
# Usage (on the GPU box): bash scripts/bench_two_ranks_one_gpu.sh [tag]
set -e
export GPU_MAX_HW_QUEUES=16 MQC_BENCH_DEVICE=0
O=gpurun_out/${1:-two_ranks}
mkdir -p $O
for mode in weak strong; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
      bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --scaling $mode --no-secondary --no-cpu-baseline \
      > $O/bench_two_ranks_$mode.json 2> $O/bench_two_ranks_$mode.err
  python3 - <<PY
import json
d = json.loads(open("$O/bench_two_ranks_$mode.json").read().strip().splitlines()[-1])
print("$mode", d["n_gpus"], d["scaling"], "%.1f ms/step" % d["ms_per_step"], "%.0f it/s" % d["value"], d["config"]["fragments"], d["mbe2_energy_hartree"], d["rigid_motion_energy_spread"])
PY
done
python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > $O/bench_one_rank.json 2> $O/bench_one_rank.err
python3 - <<PY
import json
d = json.loads(open("$O/bench_one_rank.json").read().strip().splitlines()[-1])
print("one rank", d["n_gpus"], d["scaling"], "%.1f ms/step" % d["ms_per_step"], "%.0f it/s" % d["value"], d["config"]["fragments"], d["mbe2_energy_hartree"], d["rigid_motion_energy_spread"])
PY
