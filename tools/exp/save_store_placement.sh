#!/bin/bash
# Round-4 experiment (review item 3c): where in an epilogue piece should the forward-save kernel issue its tape store?
#   S_AFTER  behind the piece's two v_sin and the f16 split (shipped: in front of them)
#   S_PC1    in the SECOND piece of the row tile (pc == 1), behind its sines: the store leaves the registers half a k32-step later
set -e
cd /root/repo/fenerf_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc"
OTHERS=$(ls build/*.o | grep -v fenerf_siren_f16w.o)
mk() {  # name, python transform
  python3 - "$1" "$2" <<'PY'
import sys
name, mode = sys.argv[1], sys.argv[2]
s = open("fenerf_siren_f16w.hip").read()
blk_start = s.index("  if (SAVE && pc == 0) {\n    // register-dump position of this lane")
blk_end = s.index("  // x = sin(2 pi theta); carried as hi = rn_f16(16 x)")
blk = s[blk_start:blk_end]
s = s[:blk_start] + s[blk_end:]
if mode == "pc1":
    blk = blk.replace("if (SAVE && pc == 0) {", "if (SAVE && pc == 1) {")
anchor = '  asm volatile("" : "+v"(hp), "+v"(lp));\n'
assert s.count(anchor) == 1
s = s.replace(anchor, anchor + blk)
open(f"_exp_{name}.hip", "w").write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -x hip -c _exp_$1.hip -o build/_exp_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared $OTHERS build/_exp_$1.o -o ../libexp_$1.so
  rm -f _exp_$1.hip build/_exp_$1.o
  echo "built ../libexp_$1.so"
}
mk S_AFTER after &
mk S_PC1 pc1 &
wait
