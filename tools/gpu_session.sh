#!/bin/bash
# One gpurun call: GPU tests, smoke, bench, rocprof kernel trace.  Everything is logged under gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tests|bench|prof|all]'
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
tail -5 gpurun_out/tests.log gpurun_out/smoke.log gpurun_out/bench.log 2>/dev/null
exit 0
