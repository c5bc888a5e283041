#!/bin/bash
# Builds libmqc_hip.so for gfx950 in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libmqc_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-pass-failed"
mkdir -p _obj
[ -f lebedev_tables.inc ] || python3 gen_lebedev.py
HDRS="md_integrals.hpp engine.hpp eri_kernels.hpp ../../include/mqc_hip.h"
stale() { # obj src
  [ ! -f "$1" ] && return 0
  [ "$2" -nt "$1" ] && return 0
  for h in $HDRS; do [ "$h" -nt "$1" ] && return 0; done
  return 1
}
pids=()
# at most $(nproc) compilers at a time; a finished one (any, not the oldest) makes room for the next
NPAR=${MQC_BUILD_JOBS:-$(nproc)}
run() { while [ "$(jobs -rp | wc -l)" -ge "$NPAR" ]; do wait -n || exit 1; done; ( "$@" ) & pids+=($!); }
for g in 15 18 6 12 4 13 5 14 10 2 11 3 9 19 1 8 0 7 16 17; do
  o=_obj/kern_eri_inst_$g.o
  if stale "$o" kern_eri_inst.hip; then run hipcc $FLAGS -DERI_GROUP=$g -x hip -c kern_eri_inst.hip -o "$o"; fi
done
# kern_scf_wide.hip is kern_scf.hip compiled with 512 threads per fragment
[ kern_scf.hip -nt kern_scf_wide.hip ] && touch kern_scf_wide.hip
for f in kern_int1e.hip kern_eri.hip kern_eri_general.hip kern_grad.hip kern_fock.hip kern_scf.hip kern_scf_wide.hip kern_xc.hip kern_df.hip host_setup.cpp grid_host.cpp engine.cpp; do
  o=_obj/${f%.*}.o
  if stale "$o" "$f"; then run hipcc $FLAGS -x hip -c "$f" -o "$o"; fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT _obj/*.o
echo "built $OUT"
