#!/bin/bash
# Builds csrc/pxr_ba_inner.hip with -DPXR_INNER_PROFILE into tools/debug/libpixsfm_hip_innerprof.so (the other objects are
# those of the regular build): two wavefronts of k_inner_packed print how their cycles split over the phases of a round.
#   tools/inner_phase_probe.sh && PXR_HIP_LIB=tools/debug/libpixsfm_hip_innerprof.so python bench.py --no-ka --no-costmap --no-cpu-baseline --no-api-e2e --steps 2 --warmup 1
set -e
cd "$(dirname "$0")/../pixel-perfect-sfm_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -munsafe-fp-atomics"
mkdir -p ../../tools/debug
/opt/rocm/bin/hipcc $FL -DPXR_INNER_PROFILE -c pxr_ba_inner.hip -o /tmp/pxr_inner_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/debug/libpixsfm_hip_innerprof.so \
  $(ls build/*.o | grep -v pxr_ba_inner.hip.o) /tmp/pxr_inner_prof.o -ldl -lpthread
echo "built tools/debug/libpixsfm_hip_innerprof.so"
