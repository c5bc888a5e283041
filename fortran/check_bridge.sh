#!/bin/bash
# Builds the drop-in bridge (fortran/mqc_hip_bridge.f90, module mqc_cuest_bridge) with AMD flang against stand-ins of
# the metalquicha modules it uses, links it with libmqc_hip.so into fortran/_build/check_bridge, writes the flat basis
# files the stand-in reader reads, and (unless --no-run) runs the program: on a GPU box it reproduces the reference's
# check_rhf golden through the bridge; without a GPU the first call must return the engine's "no HIP device" error.
set -e
cd "$(dirname "$0")"
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
mkdir -p _build/basis
python3 make_flat_basis.py _build/basis
$FC -c stubs/reference_interface_stubs.f90 -J _build -o _build/stubs.o
$FC -c mqc_hip_c.f90 -J _build -o _build/mqc_hip_c.o
$FC -c mqc_hip_bridge.f90 -I _build -J _build -o _build/mqc_hip_bridge.o
$FC -c mqc_hip_node_worker.f90 -I _build -J _build -o _build/mqc_hip_node_worker.o
$FC check_bridge.f90 _build/mqc_hip_bridge.o _build/mqc_hip_c.o _build/stubs.o -I _build -L ../metalquicha_amd -lmqc_hip \
    -Wl,-rpath,$(cd ../metalquicha_amd && pwd) -o _build/check_bridge
$FC check_node_worker.f90 _build/mqc_hip_node_worker.o _build/mqc_hip_bridge.o _build/mqc_hip_c.o _build/stubs.o -I _build \
    -L ../metalquicha_amd -lmqc_hip -Wl,-rpath,$(cd ../metalquicha_amd && pwd) -o _build/check_node_worker
[ "$1" = "--no-run" ] && exit 0
MQC_FLAT_BASIS_PATH=$(pwd)/_build/basis ./_build/check_bridge
