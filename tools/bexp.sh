#!/bin/bash
# timing variants of the 16-point backward chain kernel (make -C fenerf_amd/csrc bexp); kernel alone, one 131,072-point chunk
mkdir -p gpurun_out
for v in "" ${BEXPS:-BW_B1 BW_TAPE_T BW_INPHASE BW_NOEPI BW_NOMFMA BW_NOBARRIER BW_NODMA} b16; do
  lib=fenerf_amd/libfenerf_hip.so; k=b16w
  if [ "$v" = b16 ]; then k=b16; elif [ -n "$v" ]; then lib=fenerf_amd/libexp_$v.so; fi
  [ -f $lib ] || continue
  echo -n "variant ${v:-shipped}: "
  FENERF_BACKWARD_KERNEL=$k FENERF_LIB=$PWD/$lib timeout 200 python tools/bench_chain.py 2>&1 | tail -1
done > gpurun_out/bexp.log 2>&1
cat gpurun_out/bexp.log
