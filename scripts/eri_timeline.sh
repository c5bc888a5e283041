#!/bin/bash
# Kernel timeline of the integral stage (one MI355X): kernel trace of a short default bench, kept as CSV for
# scripts/eri_timeline.py.  Usage on the GPU box: bash scripts/eri_timeline.sh [tag]
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-tl}
O=gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_EXTRA > $O/trace.log 2>&1
find $O/trace -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
ls -la $O
