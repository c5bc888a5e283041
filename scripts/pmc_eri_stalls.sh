#!/bin/bash
# Issue / stall / instruction-cache counters of the integral-stage kernels (separate --pmc passes, kernel trace only).
# Usage on the GPU box: bash scripts/pmc_eri_stalls.sh [tag]
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-pmc_eri}
O=gpurun_out/$TAG
mkdir -p $O
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/a -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/b -- $B > $O/b.log 2>&1
python3 scripts/pmc_kernels.py $O/a eri_ schwarz > $O/a.txt
python3 scripts/pmc_kernels.py $O/b eri_ schwarz > $O/b.txt
find $O/a -name '*kernel_trace.csv' -exec cp {} $O/a_kernel_trace.csv \;
rm -rf $O/a $O/b
tail -2 $O/a.log
