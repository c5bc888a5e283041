#!/bin/bash
# Compiles the ISO_C_BINDING module with AMD flang and links a small program against
# metalquicha_amd/libmqc_hip.so.  Runs without a GPU (context_get then reports status 3).
set -e
cd "$(dirname "$0")"
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
mkdir -p _build
$FC -c mqc_hip_c.f90 -J _build -o _build/mqc_hip_c.o
$FC check_link.f90 _build/mqc_hip_c.o -I _build -L ../metalquicha_amd -lmqc_hip -Wl,-rpath,$(cd ../metalquicha_amd && pwd) -o _build/check_link
./_build/check_link
