#!/bin/bash
# Same-box A/B of forward-kernel variants: VARIANTS="A B" bash tools/exp/ab_forward_libs.sh  (fenerf_amd/libexp_<V>.so built from scratch
# copies of the kernel file, tools/exp/README.md).  shipped, the variants, shipped again: headline step, the SIREN kernel's hipEvent time,
# shader cycles per launch (box-independent to +- 0.03 %) and the clock the power manager granted -- from bench_detail.json of each run.
for v in "" ${VARIANTS:-} ""; do
  if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
  echo -n "${v:-shipped}: "
  FENERF_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gstep --no-gstep-ddp --no-gstep-b6 --no-f32 --no-sweep64 > /dev/null 2>&1
  python -c "
import json
j=json.load(open('bench_detail.json')); r=j['roofline']
print('ms/step %.4f kernel_ms %.4f cycles %.0f clock %.3f' % (j['ms_per_step'], r['kernel_ms'], r['cycles_per_launch'], r['clock_ghz_effective']))"
done
