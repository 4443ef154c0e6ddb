#!/bin/bash
# Round-4 GPU sessions.  usage: gpurun --timeout 1500 -- 'bash tools/gpu_r4.sh [tests] [newtests] [sine] [trigab] [bench] [benchq] [prof] [pmc] [gtimeline] [fusion] [fusionexp] [fusionprof] [ddp] [cuscale] [probe]'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gstep --no-gstep-ddp --no-f32 --no-sweep64"
parity() { grep -h 'parity\]\|dist\]' "$1" | grep -v 'print(' ; }
for what in "$@"; do
case $what in
tests)
  timeout 1800 python -m pytest tests -m gpu -q -s --maxfail=10 > gpurun_out/tests_full.log 2>&1; parity gpurun_out/tests_full.log > gpurun_out/tests_parity.log; tail -300 gpurun_out/tests_full.log > gpurun_out/tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit: $?" >> gpurun_out/smoke.log
  grep -E "passed|failed|^E  |FAILED" gpurun_out/tests.log | tail -30; tail -n 4 gpurun_out/smoke.log ;;
newtests)   # the tests selected by $NEWTESTS, without -x, verbose: what they measure decides their asserts
  timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "${NEWTESTS:-sine}" > gpurun_out/newtests_full.log 2>&1; parity gpurun_out/newtests_full.log > gpurun_out/newtests_parity.log; tail -150 gpurun_out/newtests_full.log > gpurun_out/newtests.log
  cut -c1-420 gpurun_out/newtests_parity.log; grep -E "passed|failed|^E  |FAILED" gpurun_out/newtests.log | tail -40 ;;
sine)       # the sine-domain tests on the shipped library and on the unreduced round-3 arithmetic (libexp_TRIG0.so): the second run
            # is EXPECTED to fail beyond 256 revolutions -- it documents the hole the reduction closes
  for v in "" TRIG0; do
    lib=$PWD/fenerf_amd/libfenerf_hip.so; [ -n "$v" ] && lib=$PWD/fenerf_amd/libexp_$v.so
    [ -f "$lib" ] || continue
    FENERF_LIB=$lib timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "sine_arguments" > gpurun_out/sine_${v:-shipped}_full.log 2>&1
    { echo "=== library: ${v:-shipped}"; parity gpurun_out/sine_${v:-shipped}_full.log | cut -c1-420; grep -E "passed|failed|^FAILED" gpurun_out/sine_${v:-shipped}_full.log | tail -40; } > gpurun_out/sine_${v:-shipped}.log
    cat gpurun_out/sine_${v:-shipped}.log | tail -60
  done ;;
trigab)     # cost of the range reduction: in-kernel cycle stamps of the forward kernel + the G-step, shipped (rndne + sub) vs fract vs none
  for rep in 1 2; do for v in "" TRIG0 TRIG1; do
    lib=$PWD/fenerf_amd/libfenerf_hip.so; [ -n "$v" ] && lib=$PWD/fenerf_amd/libexp_$v.so
    [ -f "$lib" ] || continue
    echo -n "${v:-shipped(rndne+sub)} run $rep: "
    FENERF_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep64 --no-gstep-b6 --no-gstep-ddp 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; f=j.get('f32',{}).get('roofline',{}); g=j.get('gstep',{})
        print('f16x3 ms/step %.4f kernel_ms %.4f cycles %.0f clock %.3f | f32 kernel_ms %.4f cycles %.0f | gstep ms %.3f amp %.3f' % (j['ms_per_step'], r['kernel_ms'], r['cycles_per_launch'], r['clock_ghz_effective'], f.get('kernel_ms',0), f.get('cycles_per_launch',0), g.get('ms',0), j.get('gstep_amp',{}).get('ms',0)))"
  done; done > gpurun_out/trigab.log 2>&1
  cat gpurun_out/trigab.log ;;
bench)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1
  echo "bench exit: $?" >> gpurun_out/bench.log
  tail -2 gpurun_out/bench.log | cut -c1-8000 ;;
benchq)
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-b6 > gpurun_out/benchq.log 2>&1
  echo "benchq exit: $?" >> gpurun_out/benchq.log
  tail -2 gpurun_out/benchq.log | cut -c1-8000 ;;
ddp)        # the driver's N > 1 command line at N = 1 with the distributed branch forced: RCCL + the gstep_ddp leg
  FENERF_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-f32 > gpurun_out/bench_ddp.log 2>&1
  echo "ddp exit: $?" >> gpurun_out/bench_ddp.log
  tail -2 gpurun_out/bench_ddp.log | cut -c1-8000 ;;
prof)
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 3) > gpurun_out/prof.log 2>&1
  echo "prof exit: $?" >> gpurun_out/prof.log
  find gpurun_out/prof -type f ! -name "*stats*" -size +2M -delete
  find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs head -14 | cut -c1-200 ;;
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
cuscale)    # do the backward kernels scale with the CUs they get?  (profiles/r04_gstep_overlap_why_not.md: running the chain beside the weight
            # gradients on disjoint CU sets only pays if they do NOT)  FENERF_EXP_NUM_CUS sizes every persistent launch for n CUs
  for n in 256 192 128 64; do
    echo "== FENERF_EXP_NUM_CUS=$n"
    FENERF_EXP_NUM_CUS=$n timeout 200 python tools/time_bwd.py 196608 2>&1 | grep -E "forward_save|chain"
    FENERF_EXP_NUM_CUS=$n timeout 200 python tools/time_wgrad.py 393216 2>&1 | tail -1
  done > gpurun_out/cuscale.log 2>&1
  cat gpurun_out/cuscale.log ;;
overlap)    # backward schedules of the generator step: serial chunks vs weight gradients beside the next chunk's chain
  timeout 900 python tools/overlap_sweep.py ${OVERLAP_ARGS:-} > gpurun_out/overlap_sweep.log 2>&1; cat gpurun_out/overlap_sweep.log | tail -45 ;;
ddptl)      # kernel timeline of a DDP generator step (world 1): reference wrapper vs prepare_for_ddp + recommended arguments
  for mode in reference tuned; do
    rm -rf gpurun_out/ddptl_$mode; mkdir -p gpurun_out/ddptl_$mode
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ddptl_$mode -o tl -- python $GRAFT_REPO_ROOT/tools/ddp_timeline.py $mode) > gpurun_out/ddptl_$mode/run.log 2>&1
    python tools/gstep_timeline.py gpurun_out/ddptl_$mode 4 > gpurun_out/ddp_timeline_$mode.txt 2>&1
    find gpurun_out/ddptl_$mode -type f -size +2M -delete
    echo "== $mode"; grep -E "one generator step|bwd16w|wgrad_sq|wgrad_thin|ccl|Ccl|grid_transpose|copyBuffer" gpurun_out/ddp_timeline_$mode.txt | head -40
  done ;;
probe)      # what v_sin_f32 / v_cos_f32 return for arguments of growing magnitude on this chip
  ./tools/probe/sin_domain_probe > gpurun_out/vsin_domain_probe.txt 2>&1; cat gpurun_out/vsin_domain_probe.txt ;;
fusion)      # one-launch render: the bit-for-bit test against the four launches, then the A/B (bench shape, 256^2 x 48+48, 4 x 64^2 x 24+24)
  timeout 900 python -m pytest tests -m gpu -q -s -k "one_launch_render or two_threads or e2e or staged_forward" --maxfail=5 > gpurun_out/fusion_tests.log 2>&1
  grep -E "passed|failed|^E  |FAILED|one-launch" gpurun_out/fusion_tests.log | tail -30
  timeout 300 python tools/render_fusion_ab.py > gpurun_out/fusion_ab.log 2>&1; tail -n 2 gpurun_out/fusion_ab.log
  timeout 300 python tools/render_fusion_ab.py --size 256 --steps 48 --iters 10 >> gpurun_out/fusion_ab.log 2>&1; tail -n 1 gpurun_out/fusion_ab.log
  timeout 300 python tools/render_fusion_ab.py --B 4 --size 64 --steps 24 --iters 40 >> gpurun_out/fusion_ab.log 2>&1; tail -n 1 gpurun_out/fusion_ab.log ;;
fusionexp)   # where the one-launch render spends its extra time: tools/exp/fused_ray_phase.sh variants (barriers only / no ray phase)
  for v in "" RAY1 RAY0 RAY3 M64; do
    lib=$PWD/fenerf_amd/libfenerf_hip.so; [ -n "$v" ] && lib=$PWD/fenerf_amd/libexp_$v.so
    [ -f $lib ] || continue
    echo -n "variant ${v:-shipped}: "; FENERF_LIB=$lib timeout 200 python tools/render_fusion_ab.py --rounds 3 --modes off,force 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print(j['ms_per_render'], j['launch_groups_of_one_render']['force']['ms'])"
  done 2>&1 | tee gpurun_out/fusionexp.log ;;
fusionprof)  # rocprofv3 kernel trace of the one-launch route: one kernel per render
  rm -rf gpurun_out/fprof; mkdir -p gpurun_out/fprof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fprof -o fprof -- python $GRAFT_REPO_ROOT/tools/render_fusion_ab.py --rounds 2 --iters 20 --modes force) > gpurun_out/fprof/run.log 2>&1
  find gpurun_out/fprof -name "*kernel_stats.csv" | head -1 | xargs -r head -8
  find gpurun_out/fprof -type f -size +2M -delete ;;
gtimeline)   # per-launch timeline of one generator step (kernel trace only, no counters)
  rm -rf gpurun_out/gtl; mkdir -p gpurun_out/gtl
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gtl -o gtl -- python $GRAFT_REPO_ROOT/tools/bench_gstep.py --B 1 --size 128 --skip-eager --iters 4 ${GTL_ARGS:-}) > gpurun_out/gtl/run.log 2>&1
  python tools/gstep_timeline.py gpurun_out/gtl ${GTL_STEP:-4} > gpurun_out/gstep_timeline.txt 2>&1
  find gpurun_out/gtl -type f -size +2M -delete
  tail -60 gpurun_out/gstep_timeline.txt ;;
esac
done
exit 0
