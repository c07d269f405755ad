#!/bin/bash
# Builds ONE translation unit of csrc/ with extra -D flags into tools/debug/libpixsfm_hip_$TAG.so (the other objects are those of
# the regular build: run `make -C pixel-perfect-sfm_amd/csrc` first); A/B with PXR_HIP_LIB=tools/debug/libpixsfm_hip_$TAG.so.
#   tools/variant_build.sh <tag> <file.hip> [-DFLAG ...]
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../pixel-perfect-sfm_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -munsafe-fp-atomics"
mkdir -p ../../tools/debug
[ "$SRC" = pxr_ka.hip ] && FL="$FL -mllvm -amdgpu-spill-vgpr-to-agpr=0"     # (as in the Makefile)
/opt/rocm/bin/hipcc $FL "$@" -c $SRC -o /tmp/pxr_variant_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/debug/libpixsfm_hip_$TAG.so \
  $(ls build/*.o | grep -v "$SRC.o") /tmp/pxr_variant_$TAG.o -ldl -lpthread
echo "built tools/debug/libpixsfm_hip_$TAG.so"
