#!/bin/bash
# Builds libmqc_hip.so for gfx950 in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libmqc_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-pass-failed"
mkdir -p _obj
pids=()
[ -f lebedev_tables.inc ] || python3 gen_lebedev.py
for f in kern_int1e.hip kern_eri.hip kern_fock.hip kern_scf.hip kern_xc.hip kern_df.hip host_setup.cpp grid_host.cpp engine.cpp; do
  o=_obj/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ md_integrals.hpp -nt "$o" ] || [ engine.hpp -nt "$o" ] || [ ../../include/mqc_hip.h -nt "$o" ]; then
    ( hipcc $FLAGS -x hip -c "$f" -o "$o" $EXTRA ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT _obj/*.o
echo "built $OUT"
