#!/bin/bash
# Round-5 GPU session script (run through gpurun): targeted tests first, then timing.  Everything goes to gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
which="${1:-all}"
if [ "$which" = "tape16" ] || [ "$which" = "all" ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -x -k "16bit_tape or tape16" > gpurun_out/r5_tape16_tests.log 2>&1
  echo "tape16 tests rc=$?"; tail -3 gpurun_out/r5_tape16_tests.log
fi
if [ "$which" = "new" ] || [ "$which" = "all" ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -k "trained or inversion_reproduces or test_siren_forward_vs_reference or test_forward_with_frequencies_vs_reference or test_full_size or fp64_arbiter or test_config5 or far_beyond_the_init_range or mapping_network_native or split_backward or staged_forward_with_frequencies_vs_reference or forward_and_staged" > gpurun_out/r5_new_tests.log 2>&1
  echo "new tests rc=$?"; tail -3 gpurun_out/r5_new_tests.log
fi
if [ "$which" = "bench" ] || [ "$which" = "all" ]; then
  timeout 900 python bench.py --no-gstep-ddp --no-gstep-b6 --quick-cpu-baseline > gpurun_out/r5_bench_quick.log 2>&1
  echo "bench rc=$?"; tail -c 3000 gpurun_out/r5_bench_quick.log
fi
if [ "$which" = "glue" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -x -k "label_head or image_layout or mapping_network_native or integration_md or render_backward_abi or split_backward or gstep_e2e or generator_step" > gpurun_out/r5_glue_tests.log 2>&1
  echo "glue tests rc=$?"; tail -3 gpurun_out/r5_glue_tests.log
  for mode in bare gdp; do
    mkdir -p gpurun_out/ddptl_$mode
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ddptl_$mode -o tl -- python $GRAFT_REPO_ROOT/tools/ddp_timeline.py $mode) > gpurun_out/ddptl_$mode/run.log 2>&1
    python tools/gstep_timeline.py gpurun_out/ddptl_$mode 4 > gpurun_out/r5_ddp_timeline_${mode}_v2.txt 2>&1
    head -1 gpurun_out/r5_ddp_timeline_${mode}_v2.txt
    rm -rf gpurun_out/ddptl_$mode
  done
  timeout 900 python bench.py --no-cpu-baseline --no-f32 --no-sweep64 > gpurun_out/r5_bench_glue.log 2>&1
  echo "bench rc=$?"; tail -c 2500 gpurun_out/r5_bench_glue.log
fi
if [ "$which" = "glue2" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -x -k "generator_step_through_ddp or reduced_precision_forward_modes or split_backward or configs2" > gpurun_out/r5_glue2_tests.log 2>&1
  echo "glue2 tests rc=$?"; tail -3 gpurun_out/r5_glue2_tests.log
  timeout 900 python bench.py --no-cpu-baseline --no-f32 --no-sweep64 > gpurun_out/r5_bench_glue2.log 2>&1
  echo "bench rc=$?"; tail -c 600 gpurun_out/r5_bench_glue2.log
fi
if [ "$which" = "gpmc" ]; then      # HBM traffic (two tape formats) and matrix-pipe occupancy of the generator-step kernels at 1 x 128^2 x 24+24
  for gp in f32 tape16; do
    GSTEP_ARGS="--B 1 --size 128 --grad-precision $gp" bash tools/pmc_gstep.sh > /dev/null 2>&1
    cp gpurun_out/pmc_gstep/gstep_pmc_summary.txt gpurun_out/r5_pmc_gstep_traffic_$gp.txt
  done
  GSTEP_ARGS="--B 1 --size 128 --grad-precision f32" bash tools/pmc_gstep_mfma.sh > /dev/null 2>&1
  cp gpurun_out/pmc_gstep_mfma/summary.txt gpurun_out/r5_pmc_gstep_mfma_f32.txt
  rm -rf gpurun_out/pmc_gstep gpurun_out/pmc_gstep_mfma
  grep -v "^#" gpurun_out/r5_pmc_gstep_traffic_f32.txt | grep "siren\|kernel," | head -30
  grep "^#" gpurun_out/r5_pmc_gstep_mfma_f32.txt
fi
