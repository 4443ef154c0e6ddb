#!/bin/bash
# Round 3 experiment: what bounds the fp32-class square weight-gradient kernel (siren_wgrad_sq_bf16_kernel, 5.4 of 8 TB/s)?
# Timing-only variants built from scratch copies of fenerf_siren_wgrad.hip (the product source is not touched; results are wrong on purpose):
#   Q_NOSTORE   the staging's 16-bit LDS stores removed (operands in LDS stay what the prologue wrote)
#   Q_NOSIN     x = the tape value instead of sin(2 pi (f' t + p'))
#   Q_NOMFMA    no MFMAs (accumulators untouched)
#   Q_NOFRAG    the B fragments of a group are not re-read (the first group's are reused)
# usage: bash tools/exp/wgrad_sq_variants.sh (here: hipcc cross-compiles), then on the GPU box
#   for v in "" Q_NOSTORE ...; do FENERF_LIB=$PWD/fenerf_amd/libexp_$v.so python tools/time_wgrad.py; done
set -e
cd "$(dirname "$0")/../../fenerf_amd/csrc"
make -j8 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc"
OTHERS=$(ls build/*.o | grep -v fenerf_siren_wgrad.o)
mk() {  # name, python patch (old -> new pairs read from stdin as a small script)
  python3 - "$1" <<PY
import sys
s = open("fenerf_siren_wgrad.hip").read()
$2
open("_exp_" + sys.argv[1] + ".hip", "w").write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -x hip -c _exp_$1.hip -o build/_exp_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared $OTHERS build/_exp_$1.o -o ../libexp_$1.so
  rm -f _exp_$1.hip build/_exp_$1.o
  echo "built ../libexp_$1.so"
}
mk Q_NOSTORE '
a = s.index("__device__ __forceinline__ void stage_split4"); b = s.index("struct Frag16")
body = s[a:b]
import re
q = chr(34)
body2 = re.sub(r"row0_m\[[^;]*\] = ([^;]*);", lambda m: "asm volatile(" + q + q + " :: " + q + "v" + q + "((unsigned)(" + m.group(1) + ")));", body)
assert body2 != body
s = s[:a] + body2 + s[b:]
' &
mk Q_NOSIN '
old = "const float x[4] = {sin2pi(__builtin_fmaf(f4.x, b.x, p4.x)), sin2pi(__builtin_fmaf(f4.y, b.y, p4.y)),\n                          sin2pi(__builtin_fmaf(f4.z, b.z, p4.z)), sin2pi(__builtin_fmaf(f4.w, b.w, p4.w))};"
assert old in s
s = s.replace(old, "const float x[4] = {__builtin_fmaf(f4.x, b.x, p4.x), __builtin_fmaf(f4.y, b.y, p4.y), __builtin_fmaf(f4.z, b.z, p4.z), __builtin_fmaf(f4.w, b.w, p4.w)};")
' &
mk Q_NOMFMA '
old = "#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)"
assert old in s
s = s.replace(old, "__device__ __forceinline__ f32x16 mfma_sink(const bf16x8& a, const bf16x8& b, f32x16 c) { asm volatile(\"\" :: \"v\"(a), \"v\"(b)); return c; }\n#define MFMA_BF16(a, b, c) mfma_sink((a), (b), (c))")
' &
mk Q_NOFRAG '
old = "bn[0] = frag(B_p, nt, 0); bn[1] = frag(B_p, nt, 1);\n          }\n#pragma unroll\n          for (int j = 0; j < HPG; ++j)"
assert old in s
s = s.replace(old, "bn[0] = bf[0]; bn[1] = bf[1]; (void)nt;\n          }\n#pragma unroll\n          for (int j = 0; j < HPG; ++j)")
' &
wait
