#!/bin/bash
# Measurement builds: the register / twin / pass class kernels (kern_eri_inst groups 0-6, 16, 17) recompiled with extra
# flags into metalquicha_amd/libmqc_hip_<label>.so; everything else comes from the product build's objects.
# Usage: bash scripts/build_variant.sh <label> <flags...>      e.g.  nostore -DMQC_ERI_NO_STORE
set -e
cd "$(dirname "$0")/../metalquicha_amd/csrc"
L=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-pass-failed"
rm -rf /tmp/obj_$L; cp -r _obj /tmp/obj_$L
for g in 0 1 2 3 4 5 6 16 17; do
  hipcc $FLAGS "$@" -DERI_GROUP=$g -x hip -c kern_eri_inst.hip -o /tmp/obj_$L/kern_eri_inst_$g.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmqc_hip_$L.so /tmp/obj_$L/*.o
echo "built libmqc_hip_$L.so"
