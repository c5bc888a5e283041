#!/bin/bash
# Kernel timeline of single benzene DF-B3LYP SCFs (BASELINE configs[1]) through mqc_hip_scf_run.
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-benz_tl}
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 scripts/single_call_probe.py 1 > $O/trace.log 2>&1
find $O/trace -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
tail -3 $O/trace.log
