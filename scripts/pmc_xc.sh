#!/bin/bash
# Issue / stall counters of the split-quadrature kernels (B3LYP bench, one evaluation), two --pmc passes.
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_xc}
mkdir -p $O
B="python3 bench.py --functional b3lyp --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"
C2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES"
C3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/x1 -- $B > $O/x1.log 2>&1
rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/x2 -- $B > $O/x2.log 2>&1
rocprofv3 --pmc $C3 --kernel-trace --output-format csv -d $O/x3 -- $B > $O/x3.log 2>&1
for d in x1 x2 x3; do python3 scripts/pmc_kernels.py $O/$d xc_ > $O/$d.txt; done
find $O/x1 -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/x1 $O/x2 $O/x3
cat $O/x1.txt $O/x2.txt $O/x3.txt
