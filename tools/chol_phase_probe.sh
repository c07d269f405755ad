#!/bin/bash
# Phase times of k_chol_panel (workgroups 0 and 1 of the first panel): builds pxr_chol.hip with -DPXR_CHOL_PROFILE into a
# scratch copy of the library, runs ONE factorisation on the GPU box and restores the product library.
# usage: tools/chol_phase_probe.sh   (from the repository root, in the build container)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/pixel-perfect-sfm_amd/csrc
LIB=$ROOT/pixel-perfect-sfm_amd/pixsfm_amd/libpixsfm_hip.so
cp $LIB /tmp/libpixsfm_hip.so.keep
trap 'cp /tmp/libpixsfm_hip.so.keep $LIB' EXIT
(cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -munsafe-fp-atomics \
    -DPXR_CHOL_PROFILE -c -o /tmp/chol_prof.o pxr_chol.hip && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $LIB $(ls build/*.o | grep -v pxr_chol) /tmp/chol_prof.o -ldl)
/usr/local/graft/bin/gpurun --timeout 200 -- 'timeout 100 python tools/_time_chol.py 1593 1 < /dev/null 2>&1 | grep "chol \|n = " | sort | head -12'
