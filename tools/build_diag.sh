#!/bin/bash
# Stamped diagnostic build of the kernel library (never loaded by default).
set -e
cd "$(dirname "$0")/../distributed_tensorflow_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function -Wno-pass-failed -ffp-contract=off \
  -DDTF_STAMPS -shared kernels/*.hip -o ../lib/libdtf_kernels_diag.so -ldl
