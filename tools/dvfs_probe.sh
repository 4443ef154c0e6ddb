#!/bin/bash
# DVFS probe: the same forward-render loop on real and on all-zero weights, with rocm-smi sampled next to it.
# Same binary, same instruction stream: any difference in kernel time is the clock the power manager grants.
# usage (GPU box): bash tools/dvfs_probe.sh > gpurun_out/dvfs_probe.log
Q="--steps 1500 --warmup 3 --no-cpu-baseline --no-gstep --no-f32 --no-sweep64"
for z in "" 1; do
  for k in f16w f16s; do
    echo "=== zero_weights=${z:-0} forward_kernel=$k"
    FENERF_BENCH_ZERO_WEIGHTS=$z FENERF_FORWARD_KERNEL=$k python bench.py $Q > /tmp/dvfs_bench.log 2>&1 &
    pid=$!
    sleep 12
    for i in 1 2 3; do
      rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/^/    /'
      sleep 0.7
    done
    wait $pid
    tail -1 /tmp/dvfs_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('    kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_per_step', round(d['ms_per_step'],3))"
  done
done
