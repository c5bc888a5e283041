#!/bin/bash
# Counters of the in-core J/K kernels: the triangular-tensor kernel and (MQC_HIP_ERI_TRI=0) the square one, same box.
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_jk}
mkdir -p $O
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"
C2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES"
rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/t1 -- $B > $O/t1.log 2>&1
rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/t2 -- $B > $O/t2.log 2>&1
if [ -z "$TRI_ONLY" ]; then export MQC_HIP_ERI_TRI=0
rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/s1 -- $B > $O/s1.log 2>&1
rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/s2 -- $B > $O/s2.log 2>&1; fi
for d in t1 t2 s1 s2; do [ -d $O/$d ] && python3 scripts/pmc_kernels.py $O/$d jk_t jk_incore_kernel\<1,\ true,\ 12 > $O/$d.txt && rm -rf $O/$d; done
cat $O/*.txt
