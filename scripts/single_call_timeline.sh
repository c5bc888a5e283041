#!/bin/bash
# Kernel timeline of single water-dimer SCFs through mqc_hip_scf_run (the literal drop-in path).
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-sc_tl}
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 scripts/single_call_probe.py 2 > $O/trace.log 2>&1
find $O/trace -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
