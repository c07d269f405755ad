#!/bin/bash
# Builds csrc/pxr_ka.hip with -DPXR_KA_PROFILE into tools/debug/libpixsfm_hip_kaprof.so (the other objects are those of the
# regular build): one workgroup of pxr_ka_solve (argument: its sub-problem index, default 0) prints how its wall time splits over the
# phases of the LM loop, when it started and how long it ran.
#   tools/ka_phase_probe.sh && PXR_HIP_LIB=tools/debug/libpixsfm_hip_kaprof.so python bench.py --no-cpu-baseline --no-costmap --lm-iters 0
set -e
cd "$(dirname "$0")/../pixel-perfect-sfm_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -munsafe-fp-atomics -mllvm -amdgpu-spill-vgpr-to-agpr=0"
mkdir -p ../../tools/debug
/opt/rocm/bin/hipcc $FL -DPXR_KA_PROFILE -DPXR_KA_PROFILE_BLOCK=${1:-0} -c pxr_ka.hip -o /tmp/pxr_ka_prof${1:-0}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/debug/libpixsfm_hip_kaprof${1:-}.so \
  $(ls build/*.o | grep -v pxr_ka.hip.o) /tmp/pxr_ka_prof${1:-0}.o -ldl
echo "built tools/debug/libpixsfm_hip_kaprof${1:-}.so"
