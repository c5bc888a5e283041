#!/bin/bash
# test-only: host build of the __host__ __device__ integral templates (see check_device_math.cpp)
set -e
cd "$(dirname "$0")"
SRC=check_device_math.cpp
if [ ! -f libhostcheck.so ] || [ $SRC -nt libhostcheck.so ] || [ ../../metalquicha_amd/csrc/md_integrals.hpp -nt libhostcheck.so ] || [ ../../metalquicha_amd/csrc/host_setup.cpp -nt libhostcheck.so ]; then
  hipcc -O1 -std=c++17 -fPIC -shared -x hip --offload-host-only -Wno-pass-failed -Wno-unused-value $SRC ../../metalquicha_amd/csrc/host_setup.cpp -o libhostcheck.so
fi
echo "built tests/host/libhostcheck.so"
