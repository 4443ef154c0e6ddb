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
