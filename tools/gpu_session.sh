#!/bin/bash
# One gpurun call: GPU tests, smoke, bench, rocprof kernel trace.  Everything is logged under gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tests|bench|prof|gstep|pmc|all]'
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.log
nproc >> gpurun_out/device.log
if [[ $what == all || $what == tests ]]; then
  timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | tail -250 > gpurun_out/tests.log
  echo "pytest exit: $?" >> gpurun_out/tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit: $?" >> gpurun_out/smoke.log
fi
if [[ $what == all || $what == bench ]]; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1
  echo "bench exit: $?" >> gpurun_out/bench.log
fi
if [[ $what == all || $what == prof ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline) > gpurun_out/prof.log 2>&1
  echo "prof exit: $?" >> gpurun_out/prof.log
  find gpurun_out/prof -name "*stats*" | head >> gpurun_out/prof.log
  # keep only the small summaries (traces can be large)
  find gpurun_out/prof -type f ! -name "*stats*" -size +2M -delete
fi
if [[ $what == all || $what == bench || $what == bench16 ]]; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --precision f16x3 --no-cpu-baseline > gpurun_out/bench_f16x3.log 2>&1
  echo "bench16 exit: $?" >> gpurun_out/bench_f16x3.log
  tail -3 gpurun_out/bench_f16x3.log | cut -c1-1500
fi
if [[ $what == all || $what == gstep ]]; then
  timeout 600 python tools/bench_gstep.py --B 2 --size 64 > gpurun_out/gstep.log 2>&1
  echo "gstep exit: $?" >> gpurun_out/gstep.log
  rm -rf gpurun_out/prof_gstep
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gstep -o g -- python $GRAFT_REPO_ROOT/tools/bench_gstep.py --B 2 --size 64 --skip-eager --iters 3) > gpurun_out/prof_gstep.log 2>&1
  find gpurun_out/prof_gstep -type f ! -name "*stats*" -size +2M -delete
fi
if [[ $what == pmc ]]; then
  rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
  (cd /tmp && rocprofv3 -L 2>/dev/null | grep -iE "MFMA|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAVES|LDS_BANK|SQ_INSTS_VALU |SQ_ACTIVE_INST_VALU|SQ_WAIT_INST_ANY|SQ_WAIT_ANY" | head -60) > gpurun_out/pmc/counters_available.txt 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --precision ${PREC:-f32}) > gpurun_out/pmc/p$i.log 2>&1
    echo "pass $i ($set) exit $?" >> gpurun_out/pmc/summary.txt
  done
  find gpurun_out/pmc -type f -size +4M -delete
  python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/siren_pmc_summary.txt 2>&1
  cat gpurun_out/pmc/siren_pmc_summary.txt
fi
tail -5 gpurun_out/tests.log gpurun_out/smoke.log gpurun_out/bench.log 2>/dev/null
exit 0
