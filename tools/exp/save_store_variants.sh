#!/bin/bash
# Round 3 experiment: what do the tape stores of the forward-save kernel (siren16w_kernel<.., SAVE>) cost, and why?
# Builds timing-only variants from scratch copies of fenerf_siren_f16w.hip (the product source is not touched):
#   S_L2        every tile dumps into the same two L2-resident tape blocks (store path without HBM writes)
#   S_NOSTORE   no tape stores at all
#   S_TEMPORAL  tape stores without the nt hint
#   S_WAIT2     the counted vmcnt waits of the stream loop allow two more operations in flight (UNSAFE, timing only): if un-acked
#               stores are what the waits trip over, this recovers the no-grad time
# usage: bash tools/exp/save_store_variants.sh   (here, CPU: hipcc cross-compiles), then on the GPU box:
#   for v in "" S_L2 S_NOSTORE S_TEMPORAL S_WAIT2; do FENERF_LIB=$PWD/fenerf_amd/libexp_$v.so python tools/time_bwd.py 196608; done
set -e
cd "$(dirname "$0")/../../fenerf_amd/csrc"
make -j8 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc"
OTHERS=$(ls build/*.o | grep -v fenerf_siren_f16w.o)
mk() {  # name, sed expression
  sed -e "$2" fenerf_siren_f16w.hip > _exp_$1.hip
  /opt/rocm/bin/hipcc $FLAGS -x hip -c _exp_$1.hip -o build/_exp_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared $OTHERS build/_exp_$1.o -o ../libexp_$1.so
  rm -f _exp_$1.hip build/_exp_$1.o
  echo "built ../libexp_$1.so"
}
mk S_L2 's/(size_t)(tile >> 1) \* L \* TL/(size_t)((tile >> 1) \& 1) * L * TL/' &
mk S_NOSTORE 's/asm volatile("global_store_dwordx4 %0, %1, %2 nt\\n\\ts_nop 1" : : "v"(voff), "v"(v), "s"(g_uniform) : "memory");/asm volatile("" : : "v"(voff), "v"(v), "s"(g_uniform) : "memory");/' &
mk S_TEMPORAL 's/global_store_dwordx4 %0, %1, %2 nt/global_store_dwordx4 %0, %1, %2/' &
mk S_WAIT2 's/WAIT_VMCNT(DPF - 3);/WAIT_VMCNT(DPF - 1);/' &
wait
