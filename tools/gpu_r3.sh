#!/bin/bash
# Round-3 GPU sessions.  usage: gpurun --timeout 1500 -- 'bash tools/gpu_r3.sh [tests] [newtests] [bench] [benchq] [prof] [pmc] [pmcgstep] [probe] [saveexp] [wgradexp] [soak] [gtimeline] [gstep]'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gstep --no-f32 --no-sweep64"
for what in "$@"; do
case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=10 > gpurun_out/tests_full.log 2>&1; grep -h 'parity\]\|dist\]' gpurun_out/tests_full.log | grep -v 'print(' > gpurun_out/tests_parity.log; tail -300 gpurun_out/tests_full.log > gpurun_out/tests.log
  echo "pytest exit: $?" >> gpurun_out/tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit: $?" >> gpurun_out/smoke.log
  tail -n 5 gpurun_out/tests.log; tail -n 4 gpurun_out/smoke.log ;;
newtests)   # the tests added this round, without -x, verbose: what they measure decides their asserts
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "${NEWTESTS:-all_rays or config5 or at_scale or rccl or ddp or repacked or chunked_backward or single_latent_generator or reference_checkpoint}" > gpurun_out/newtests_full.log 2>&1; grep -h 'parity\]\|dist\]' gpurun_out/newtests_full.log | grep -v 'print(' > gpurun_out/newtests_parity.log; tail -150 gpurun_out/newtests_full.log > gpurun_out/newtests.log
  cat gpurun_out/newtests_parity.log | cut -c1-400; grep -E "passed|failed|^E  |FAILED" gpurun_out/newtests.log | tail -40 ;;
bench)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1
  echo "bench exit: $?" >> gpurun_out/bench.log
  tail -2 gpurun_out/bench.log | cut -c1-6000 ;;
benchq)
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-b6 > gpurun_out/benchq.log 2>&1
  echo "benchq exit: $?" >> gpurun_out/benchq.log
  tail -2 gpurun_out/benchq.log | cut -c1-6000 ;;
prof)
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 3) > gpurun_out/prof.log 2>&1
  echo "prof exit: $?" >> gpurun_out/prof.log
  find gpurun_out/prof -type f ! -name "*stats*" -size +2M -delete
  find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs head -12 | cut -c1-200 ;;
pmc)
  rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 $Q) > gpurun_out/pmc/p$i.log 2>&1
    echo "pass $i ($set) exit $?" >> gpurun_out/pmc/summary.txt
  done
  find gpurun_out/pmc -type f -size +4M -delete
  python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/siren_pmc_summary.txt 2>&1
  cat gpurun_out/pmc/siren_pmc_summary.txt ;;
pmcgstep)   # HBM bytes of the generator-step kernels (default and AMP weight-gradient operands) and of the SPATIALSIRENGRID launch
  for gp in f32 amp; do
    GSTEP_ARGS="--B 1 --size 128 --grad-precision $gp" bash tools/pmc_gstep.sh > /dev/null 2>&1
    cp gpurun_out/pmc_gstep/gstep_pmc_summary.txt gpurun_out/pmc_gstep_$gp.txt
  done
  rm -rf gpurun_out/pmc_local; mkdir -p gpurun_out/pmc_local
  for set in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_local/$set -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_local.py --iters 2 --explicit) > gpurun_out/pmc_local/$set.log 2>&1
  done
  python - <<'PY' > gpurun_out/pmc_local.txt 2>&1
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_local/*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "siren_local_kernel" in n or "siren_kernel" in n or "film_prep" in n:
            agg[n.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,counter,avg_KiB_per_dispatch,bytes_per_point_at_196608_points,n")
for k in sorted(agg):
    for c, v in sorted(agg[k].items()):
        a = sum(v) / len(v)
        print(f"{k},{c},{a:.6g},{a * 1024 / 196608:.1f},{len(v)}")
PY
  find gpurun_out/pmc_local -type f -size +2M -delete
  head -20 gpurun_out/pmc_gstep_f32.txt; grep -E "wgrad|bwd16w" gpurun_out/pmc_gstep_amp.txt; cat gpurun_out/pmc_local.txt; tail -3 gpurun_out/pmc_local/FETCH_SIZE.log ;;
probe)
  ./tools/probe/tr_probe > gpurun_out/tr_probe.log 2>&1; head -40 gpurun_out/tr_probe.log ;;
saveexp)   # tools/exp/save_store_variants.sh: forward-save with the tape stores removed / redirected / re-hinted
  for v in "" S_L2 S_NOSTORE S_TEMPORAL S_WAIT2 ""; do
    lib=fenerf_amd/libfenerf_hip.so; [ -n "$v" ] && lib=fenerf_amd/libexp_$v.so
    [ -f $lib ] || continue
    echo -n "variant ${v:-shipped}: "
    FENERF_LIB=$PWD/$lib timeout 200 python tools/time_bwd.py 196608 2>&1 | grep -E "forward|chain" | tr '\n' ' '; echo
  done > gpurun_out/saveexp.log 2>&1
  cat gpurun_out/saveexp.log ;;
soak)      # determinism / race soak of the fused render and of the generator step (default and AMP-class)
  timeout 900 python tools/soak_render.py --iters ${SOAK_ITERS:-100} > gpurun_out/soak_render.log 2>&1; tail -12 gpurun_out/soak_render.log
  timeout 600 python tools/soak_gstep.py ${SOAK_STEPS:-300} > gpurun_out/soak_gstep.log 2>&1; tail -3 gpurun_out/soak_gstep.log ;;
gtimeline)   # per-launch timeline of one generator step (kernel trace only, no counters)
  rm -rf gpurun_out/gtl; mkdir -p gpurun_out/gtl
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gtl -o gtl -- python $GRAFT_REPO_ROOT/tools/bench_gstep.py --B 1 --size 128 --skip-eager --iters 4 ${GTL_ARGS:-}) > gpurun_out/gtl/run.log 2>&1
  python tools/gstep_timeline.py gpurun_out/gtl ${GTL_STEP:-4} > gpurun_out/gstep_timeline.txt 2>&1
  find gpurun_out/gtl -type f -size +2M -delete
  tail -45 gpurun_out/gstep_timeline.txt ;;
wgradexp)   # tools/exp/wgrad_sq_variants.sh built the libexp_Q_*.so here; time them on one backward chunk
  for v in "" Q_NOSTORE Q_NOSIN Q_NOMFMA Q_NOFRAG ""; do
    if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
    [ -f "$lib" ] || continue
    echo -n "variant ${v:-shipped}: "; FENERF_LIB=$lib timeout 200 python tools/time_wgrad.py 2>&1 | tail -1
  done > gpurun_out/wgradexp.log 2>&1
  cat gpurun_out/wgradexp.log ;;
gstep)
  timeout 600 python tools/chunk_sweep.py > gpurun_out/chunk_sweep.log 2>&1; cat gpurun_out/chunk_sweep.log ;;
esac
done
exit 0
