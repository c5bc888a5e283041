#!/bin/bash
# A/B of the integral stage on ONE box: libmqc_hip_base.so (the build before a change, kept next to the
# product library) against libmqc_hip.so -- bench lines alternating, then kernel statistics of both.
# Usage (on the GPU box): bash scripts/ab_integral_stage.sh [tag]
set -e
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-ab}
O=gpurun_out/$TAG
mkdir -p $O
B="bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary"
# parity first: the integral and J/K tests of the suite against the oracle
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "eri or int1e or jk_incore or check_rhf or water_dimer or direct or def2 or df_rhf or block_sharing or mixed" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -3 $O/parity.log
# variants: every libmqc_hip_<label>.so next to the product library, then the product library itself ("new")
LABELS=""
for f in $GRAFT_REPO_ROOT/metalquicha_amd/libmqc_hip_*.so; do [ -f "$f" ] && LABELS="$LABELS $(basename $f .so | sed 's/libmqc_hip_//')"; done
LABELS="$LABELS new"
libof() { if [ "$1" = new ]; then echo $GRAFT_REPO_ROOT/metalquicha_amd/libmqc_hip.so; else echo $GRAFT_REPO_ROOT/metalquicha_amd/libmqc_hip_$1.so; fi; }
for i in 1 2; do
  for l in $LABELS; do
    MQC_HIP_LIBRARY=$(libof $l) python $B $BENCH_EXTRA > $O/bench_${l}_$i.json 2> $O/bench_${l}_$i.err
  done
done
for l in $LABELS; do
  MQC_HIP_LIBRARY=$(libof $l) rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$l -o $l -- python3 $B $BENCH_EXTRA > $O/prof_$l.log 2>&1
  find $O/prof_$l -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$l.csv \;
  rm -rf $O/prof_$l
done
python3 - <<EOF
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    st = d["roofline"]["stages"]
    print(f.split("/")[-1], "ms/step %.1f" % d["ms_per_step"], "eri %.1f ms" % (1e3 * st["eri"]["seconds"] / (d["steps"] + d["warmup"])),
          "jk %.1f ms" % (1e3 * st["jk"]["seconds"] / (d["steps"] + d["warmup"])), "E %.10f" % d["mbe2_energy_hartree"])
EOF
echo done
