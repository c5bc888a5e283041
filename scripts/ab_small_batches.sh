#!/bin/bash
# A/B of the latency-bound regimes on ONE box: one rank's share of an 8-way / 4-way split and the single-call path,
# every libmqc_hip_<label>.so next to the product library against the product library.
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-absmall}
mkdir -p $O
LABELS=""
for f in $GRAFT_REPO_ROOT/metalquicha_amd/libmqc_hip_*.so; do [ -f "$f" ] && LABELS="$LABELS $(basename $f .so | sed 's/libmqc_hip_//')"; done
LABELS="$LABELS new"
libof() { if [ "$1" = new ]; then echo $GRAFT_REPO_ROOT/metalquicha_amd/libmqc_hip.so; else echo $GRAFT_REPO_ROOT/metalquicha_amd/libmqc_hip_$1.so; fi; }
for i in 1 2; do
  for l in $LABELS; do
    echo "== $l ($i)" >> $O/log.txt
    MQC_HIP_LIBRARY=$(libof $l) python scripts/rank_share_probe.py 8 >> $O/log.txt 2>&1
    MQC_HIP_LIBRARY=$(libof $l) python scripts/rank_share_probe.py 4 >> $O/log.txt 2>&1
    MQC_HIP_LIBRARY=$(libof $l) python scripts/single_call_probe.py 24 >> $O/log.txt 2>&1
  done
done
cat $O/log.txt
