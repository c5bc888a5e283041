#!/bin/bash
# Kernel timeline of one rank's share of an N-way split of the (H2O)64 MBE-2 evaluation (default N = 8).
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${2:-rs_tl}
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 scripts/rank_share_probe.py ${1:-8} > $O/trace.log 2>&1
find $O/trace -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
tail -2 $O/trace.log
