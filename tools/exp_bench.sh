#!/bin/bash
# timing-only A/B experiments on the shared-stream kernel (results of the EXP variants are numerically wrong on purpose)
# build the variants first: make -C fenerf_amd/csrc exp   (they are git-ignored but travel with gpurun)
for v in "" NOBARRIER NOWAIT NODMA NOLDSREAD; do
  if [ -z "$v" ]; then lib=fenerf_amd/libfenerf_hip.so; else lib=fenerf_amd/libexp_$v.so; fi
  echo -n "variant ${v:-baseline}: "
  FENERF_LIB=$PWD/$lib timeout 200 python bench.py --precision f16x3 --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('kernel_ms', round(d['roofline']['kernel_ms'],3), 'rays/s', int(d['value']))"
done
