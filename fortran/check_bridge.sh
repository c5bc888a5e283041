#!/bin/bash
# Type-checks fortran/mqc_hip_bridge.f90 (module mqc_cuest_bridge, the drop-in) with AMD flang against interface
# stubs of the metalquicha modules it uses, and links it with libmqc_hip.so into a program that calls
# run_cuest_scf once (without a GPU the call returns the engine's "no HIP device" error through result%error).
set -e
cd "$(dirname "$0")"
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
mkdir -p _build
$FC -c stubs/reference_interface_stubs.f90 -J _build -o _build/stubs.o
$FC -c mqc_hip_c.f90 -J _build -o _build/mqc_hip_c.o
$FC -c mqc_hip_bridge.f90 -I _build -J _build -o _build/mqc_hip_bridge.o
$FC check_bridge.f90 _build/mqc_hip_bridge.o _build/mqc_hip_c.o _build/stubs.o -I _build -L ../metalquicha_amd -lmqc_hip \
    -Wl,-rpath,$(cd ../metalquicha_amd && pwd) -o _build/check_bridge
./_build/check_bridge
