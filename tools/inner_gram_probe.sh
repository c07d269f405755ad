#!/bin/bash
# Builds csrc/pxr_ba_inner.hip with extra -D flags ($2...) into tools/debug/libpixsfm_hip_gram_$1.so (the other objects are those
# of the regular build: run `make -C pixel-perfect-sfm_amd/csrc` first); A/B with
#   PXR_HIP_LIB=tools/debug/libpixsfm_hip_gram_$1.so python tools/_time_inner.py
set -e
TAG=$1; shift
cd "$(dirname "$0")/../pixel-perfect-sfm_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -munsafe-fp-atomics"
mkdir -p ../../tools/debug
/opt/rocm/bin/hipcc $FL "$@" -c pxr_ba_inner.hip -o /tmp/pxr_inner_gram_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/debug/libpixsfm_hip_gram_$TAG.so \
  $(ls build/*.o | grep -v pxr_ba_inner.hip.o) /tmp/pxr_inner_gram_$TAG.o -ldl -lpthread
echo "built tools/debug/libpixsfm_hip_gram_$TAG.so"
