#!/bin/bash
# Builds csrc/pxr_ba_inner.hip with k_inner_gram compiled for $1 wavefronts per SIMD into tools/debug/libpixsfm_hip_gram$1.so
# (the other objects are those of the regular build: run `make -C pixel-perfect-sfm_amd/csrc` first); A/B with
#   PXR_HIP_LIB=tools/debug/libpixsfm_hip_gram$1.so python tools/_time_inner.py
set -e
W=${1:-2}
cd "$(dirname "$0")/../pixel-perfect-sfm_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -munsafe-fp-atomics"
mkdir -p ../../tools/debug
/opt/rocm/bin/hipcc $FL -DPXR_GRAM_WAVES=$W -c pxr_ba_inner.hip -o /tmp/pxr_inner_gram$W.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/debug/libpixsfm_hip_gram$W.so \
  $(ls build/*.o | grep -v pxr_ba_inner.hip.o) /tmp/pxr_inner_gram$W.o -ldl -lpthread
echo "built tools/debug/libpixsfm_hip_gram$W.so"
