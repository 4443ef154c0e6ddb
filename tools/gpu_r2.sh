#!/bin/bash
# Round-2 GPU session: tests + smoke, A/B of the two f16x3 forward kernels, full bench.  Logs under gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_r2.sh [tests] [ab] [bench] [prof] [pmc]'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gstep --no-f32 --no-sweep64"
for what in "$@"; do
case $what in
tests)
  timeout 900 python -m pytest tests -m gpu -q -s -x 2>&1 | tail -150 > gpurun_out/tests.log
  echo "pytest exit: $?" >> gpurun_out/tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit: $?" >> gpurun_out/smoke.log
  tail -n 4 gpurun_out/tests.log; tail -n 4 gpurun_out/smoke.log ;;
ab)
  for k in f16w f16s f16w f16s; do
    FENERF_FORWARD_KERNEL=$k timeout 300 python bench.py --steps 20 --warmup 3 $Q 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$k', 'rays/s %.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.4f' % r['frac'])" 
  done > gpurun_out/ab.log 2>&1
  cat gpurun_out/ab.log ;;
bench)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1
  echo "bench exit: $?" >> gpurun_out/bench.log
  tail -2 gpurun_out/bench.log | cut -c1-3000 ;;
prof)
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 3) > gpurun_out/prof.log 2>&1
  echo "prof exit: $?" >> gpurun_out/prof.log
  find gpurun_out/prof -type f ! -name "*stats*" -size +2M -delete
  find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs head -8 ;;
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
wexp)
  # timing-only variants of the 16-point forward kernel (make -C fenerf_amd/csrc wexp)
  for v in "" ${WEXPS:-W_NODMA W_NOBARRIER W_NOLDSREAD W_NOEPI W_NOMFMA}; do
    if [ -z "$v" ]; then lib=fenerf_amd/libfenerf_hip.so; else lib=fenerf_amd/libexp_$v.so; fi
    [ -f $lib ] || continue
    echo -n "variant ${v:-baseline}: "
    FENERF_BENCH_ZERO_WEIGHTS=${ZERO:-} FENERF_LIB=$PWD/$lib timeout 200 python bench.py --steps 10 --warmup 2 $Q 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('kernel_ms', round(d['roofline']['kernel_ms'],4), 'rays/s', int(d['value']))"
  done > gpurun_out/wexp.log 2>&1
  cat gpurun_out/wexp.log ;;
dist2)
  # the driver's N>1 command on a 1-GPU box cannot get 2 GPUs; exercise the self-launch with the CPU rendezvous instead
  timeout 300 python bench.py --gpus 2 --dist-check --dist-backend gloo > gpurun_out/dist2.log 2>&1; tail -2 gpurun_out/dist2.log ;;
esac
done
exit 0
