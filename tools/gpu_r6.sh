#!/bin/bash
# Round-6 GPU session script (run through gpurun): targeted tests first, then timing.  Everything goes to gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
which="${1:-all}"
if [ "$which" = "new" ] || [ "$which" = "all" ]; then
  timeout 1800 python -m pytest tests/test_gpu_parity.py -q -s -k "world_2 or bench_gpus_2 or bench_under_torch or mapping_network_native or generator_gradients_vs_reference or test_full_size or fp64_arbiter or f16x3c2 or test_siren_forward_vs_reference or test_forward_with_frequencies_vs_reference or far_beyond_the_init_range or style_generator3d or single_latent_generator_vs or rays_mode or single_latent_spatial or generator_step_through_ddp or reduced_precision" > gpurun_out/r6_new_tests.log 2>&1
  echo "new tests rc=$?"; tail -5 gpurun_out/r6_new_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r6_smoke.log
fi
if [ "$which" = "bench" ] || [ "$which" = "all" ]; then
  t0=$(date +%s)
  timeout 900 python bench.py > gpurun_out/r6_bench_default.log 2> gpurun_out/r6_bench_default.err
  echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; tail -1 gpurun_out/r6_bench_default.log | wc -c; tail -1 gpurun_out/r6_bench_default.log
  cp bench_detail.json gpurun_out/r6_bench_detail.json
  # the opt-in leg that is not part of the default command: the 6-image micro-batch with the exact-sparsity backward
  timeout 600 python bench.py --gstep-sparse-b6 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-ddp > gpurun_out/r6_bench_sparse_b6.log 2>&1
  python - <<PY
import json
d = json.load(open("bench_detail.json"))
keep = {k: {a: b for a, b in d[k].items() if a != "what"} for k in ("gstep", "gstep_sparse", "gstep_b6", "gstep_sparse_b6") if k in d}
json.dump(keep, open("gpurun_out/r6_bench_sparse_legs.json", "w"), indent=1)
print({k: round(v.get("ms", -1), 2) for k, v in keep.items()})
PY
  cp gpurun_out/r6_bench_detail.json bench_detail.json
fi
if [ "$which" = "full" ]; then
  timeout 3000 python -m pytest tests -q -s -m gpu > gpurun_out/r6_gpu_tests_full.log 2>&1
  echo "full gpu tests rc=$?"; tail -5 gpurun_out/r6_gpu_tests_full.log
fi
if [ "$which" = "fix1" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "(test_siren_forward_vs_reference or far_beyond_the_init_range or style_generator3d or test_full_size or world_2) and (f16x3c2 or world_2)" > gpurun_out/r6_fix1_tests.log 2>&1
  echo "fix1 tests rc=$?"; tail -3 gpurun_out/r6_fix1_tests.log
fi
if [ "$which" = "chain" ]; then     # after a chain-kernel edit: its parity tests, then the generator-step leg's per-kernel times
  timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "siren_backward or generator_gradient_end_to_end or chunked_backward or generator_gradients_vs_reference or fused_grid_scatter or deterministic or inversion or at_scale or grid_gradient_values or split_backward or render_backward_abi" > gpurun_out/r6_chain_tests.log 2>&1
  echo "chain tests rc=$?"; tail -3 gpurun_out/r6_chain_tests.log
  for i in 1 2; do
    timeout 600 python bench.py --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-ddp --no-gstep-b6 > gpurun_out/r6_bench_chain_$i.log 2>&1
    python - <<PY
import json
d = json.load(open("bench_detail.json"))
g = d["gstep"]
print("gstep ms", round(g["ms"], 3), {k["name"]: round(k["ms"], 3) for k in g["roofline"]["per_kernel"]}, "headline", round(d["value"]))
for k in ("gstep_tape16", "gstep_amp", "gstep_amp16"):
    if k in d and "ms" in d[k]: print(k, round(d[k]["ms"], 3), {q["name"]: round(q["ms"], 3) for q in d[k]["roofline"]["per_kernel"]})
PY
  done
fi
if [ "$which" = "pw" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -x -k "pointwise_siren_backward or spatial_siren_grid" > gpurun_out/r6_pw_tests.log 2>&1
  echo "pointwise tests rc=$?"; grep "parity\]" gpurun_out/r6_pw_tests.log | cut -c1-400; tail -30 gpurun_out/r6_pw_tests.log | cut -c1-300
fi
if [ "$which" = "evidence" ]; then   # the round's profiles: kernel stats of the default bench command, PMC passes, per-point backward timing
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 3) > gpurun_out/r6_bench_under_rocprofv3.log 2>&1
  echo "prof exit: $?"
  find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} gpurun_out/r6_bench_kernel_stats_default_command.csv
  cp bench_detail.json gpurun_out/r6_bench_detail_under_rocprofv3.json
  find gpurun_out/prof -type f -size +1M -delete
  head -8 gpurun_out/r6_bench_kernel_stats_default_command.csv | cut -c1-220
  bash tools/gpu_r6.sh pmcfwd > gpurun_out/r6_pmc_forward.log 2>&1      # (the round-3 script's pmc step no longer matches bench.py's flags)
  GSTEP_ARGS="--B 1 --size 128 --grad-precision f32" bash tools/pmc_gstep_mfma.sh > /dev/null 2>&1
  cp gpurun_out/pmc_gstep_mfma/summary.txt gpurun_out/r6_pmc_gstep_mfma_f32.txt; rm -rf gpurun_out/pmc_gstep_mfma
  bash tools/pmc_gstep_waits.sh > /dev/null 2>&1
  cp gpurun_out/pmc_gstep_waits/summary.txt gpurun_out/r6_pmc_gstep_waits.txt; rm -rf gpurun_out/pmc_gstep_waits
  grep "bwd16w\|^kernel" gpurun_out/r6_pmc_gstep_waits.txt | head; grep "^#" gpurun_out/r6_pmc_gstep_mfma_f32.txt
  for n in 65536 196608; do timeout 300 python tools/time_pointwise_backward.py $n 256 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r6_pointwise_backward_timing.txt
  cat gpurun_out/r6_pointwise_backward_timing.txt
fi
if [ "$which" = "pmcfwd" ]; then   # HBM traffic + pipe counters of the headline forward kernel: separate --pmc passes over the (shortened) default command
  rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
  QQ="--no-cpu-baseline --no-gstep --no-f32 --no-sweep64 --no-gstep-ddp --no-gstep-b6"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 $QQ) > gpurun_out/pmc/p$i.log 2>&1
    echo "pass $i ($set) exit $?"
  done
  find gpurun_out/pmc -type f -size +4M -delete
  python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/r6_pmc_siren16w_f16x3.txt 2>&1
  cat gpurun_out/r6_pmc_siren16w_f16x3.txt; tail -3 gpurun_out/pmc/p1.log
  rm -rf gpurun_out/pmc
fi
if [ "$which" = "dephase" ]; then   # A/B of the forward kernel's wave de-phasing (libexp_dephase{1,2}.so) against the shipped library, interleaved
  for rep in 1 2; do
  for v in "" dephase2 dephase1; do
    if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
    [ -f $lib ] || continue
    echo -n "${v:-shipped}: "
    FENERF_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gstep --no-f32 --no-sweep64 --no-gstep-ddp --no-gstep-b6 > /dev/null 2>&1
    python - <<PY
import json
d = json.load(open("bench_detail.json")); r = d["roofline"]
print("rays/s %.0f ms/step %.4f kernel_ms %.4f cycles %.0f clock %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r.get("cycles_per_launch", 0), r.get("clock_ghz_effective", 0)))
PY
  done; done
  FENERF_LIB=$PWD/fenerf_amd/libexp_dephase1.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "test_siren_forward_vs_reference or test_forward_with_frequencies_vs_reference or one_launch_render or test_full_size" 2>&1 | tail -3
fi
if [ "$which" = "halfab" ]; then   # same-box A/B of the wave-half copies: shipped vs chain with the run-time flag vs forward(-save) with the run-time flag
  for rep in 1 2; do
  for v in "" chain_runtime_flag forward_runtime_flag; do
    if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
    [ -f $lib ] || continue
    echo -n "${v:-shipped}: "
    FENERF_LIB=$lib timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-ddp --no-gstep-b6 > /dev/null 2>&1
    python - <<PY
import json
d = json.load(open("bench_detail.json")); r = d["roofline"]; g = d["gstep"]
print("rays/s %.0f kernel_ms %.4f cycles %.0f clock %.3f | gstep %.3f" % (d["value"], r["kernel_ms"], r.get("cycles_per_launch", 0), r.get("clock_ghz_effective", 0), g["ms"]),
      {k["name"]: round(k["ms"], 3) for k in g["roofline"]["per_kernel"]}, "| tape16 %.3f amp16 %.3f" % (d["gstep_tape16"]["ms"], d["gstep_amp16"]["ms"]))
PY
  done; done
fi
if [ "$which" = "b6" ]; then
  for i in 1 2 3; do
    timeout 600 python bench.py --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-ddp > /dev/null 2>&1
    python - <<PY
import json
d = json.load(open("bench_detail.json"))
print("gstep %.3f b6 %.3f ms (%.3f per image) peak %.1f GB" % (d["gstep"]["ms"], d["gstep_b6"]["ms"], d["gstep_b6"]["ms"] / 6, d["gstep_b6"]["peak_GB"]))
PY
  done
fi
if [ "$which" = "prioab" ]; then   # same-box A/B: static issue priority for waves 4-7 in the chain / in the square weight-gradient kernel
  for rep in 1 2; do
  for v in "" chain_setprio wgrad_setprio; do
    if [ -z "$v" ]; then lib=$PWD/fenerf_amd/libfenerf_hip.so; else lib=$PWD/fenerf_amd/libexp_$v.so; fi
    [ -f $lib ] || continue
    echo -n "${v:-shipped}: "
    FENERF_LIB=$lib timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-sweep64 --no-gstep-ddp --no-gstep-b6 > /dev/null 2>&1
    python - <<PY
import json
d = json.load(open("bench_detail.json")); g = d["gstep"]
print("gstep %.3f" % g["ms"], {k["name"]: round(k["ms"], 3) for k in g["roofline"]["per_kernel"]}, "| tape16 %.3f amp16 %.3f" % (d["gstep_tape16"]["ms"], d["gstep_amp16"]["ms"]))
PY
  done; done
fi
