#!/bin/bash
# Instrumented build for measurements only: kern_xc.hip with in-kernel phase stamps (-DXC_STAMPS=1) linked with the
# regular objects into ../libmqc_hip_stamps.so (load it with MQC_HIP_LIBRARY=...; never shipped as the product library).
set -e
cd "$(dirname "$0")"
bash build.sh
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-pass-failed"
mkdir -p _obj_stamps
hipcc $FLAGS -DXC_STAMPS=1 -x hip -c kern_xc.hip -o _obj_stamps/kern_xc.o
OBJS=$(ls _obj/*.o | grep -v "/kern_xc.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmqc_hip_stamps.so $OBJS _obj_stamps/kern_xc.o
echo "built ../libmqc_hip_stamps.so"
