#!/bin/bash
# Builds timing variants of the one-launch render (fenerf_siren_f16w.hip FENERF_EXP_RAY_PHASE / FENERF_EXP_FUSED_MAXM):
#   libexp_RAY1.so  fences and barriers of the ray phases, no rays        libexp_RAY0.so  no ray phase
#   libexp_RAY3.so  every ray composited twice (marginal cost per ray)    libexp_M64.so   ray phases compiled for 2 N <= 64
# `bash tools/gpu_r4.sh fusionexp` times them.
set -e
cd "$(dirname "$0")/../../fenerf_amd/csrc"
build() { make -j16 OUT=../libexp_$1.so OBJDIR=build_$1 EXTRA="$2" > /dev/null; rm -rf build_$1; }
for v in ${@:-RAY0 RAY1 RAY3 M64}; do
  case $v in
    RAY0) build RAY0 -DFENERF_EXP_RAY_PHASE=0 ;;
    RAY1) build RAY1 -DFENERF_EXP_RAY_PHASE=1 ;;
    RAY3) build RAY3 -DFENERF_EXP_RAY_PHASE=3 ;;
    M64)  build M64 -DFENERF_EXP_FUSED_MAXM=64 ;;
  esac
done
ls -la ../libexp_*.so
