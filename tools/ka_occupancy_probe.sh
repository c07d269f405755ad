#!/bin/bash
# Builds csrc/pxr_ka.hip with the solve kernel capped at $1 (default 3) wavefronts per SIMD into
# tools/debug/libpixsfm_hip_occ$1.so (the other objects are those of the regular build: run `make -C
# pixel-perfect-sfm_amd/csrc` first) -- the input of tests/fuzz/ka_occupancy_probe.py.
set -e
W=${1:-3}
cd "$(dirname "$0")/../pixel-perfect-sfm_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -munsafe-fp-atomics"
mkdir -p ../../tools/debug
/opt/rocm/bin/hipcc $FL -DPXR_KA_WAVES=$W -c pxr_ka.hip -o /tmp/pxr_ka_occ$W.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/debug/libpixsfm_hip_occ$W.so \
  $(ls build/*.o | grep -v pxr_ka.hip.o) /tmp/pxr_ka_occ$W.o -ldl
echo "built tools/debug/libpixsfm_hip_occ$W.so"
