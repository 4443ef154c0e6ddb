#!/bin/bash
# Copies the round-6 evidence tools/gpu_r6.sh left under gpurun_out/ (scratch) into profiles/ (tracked), under the names DESIGN.md cites.
set -e
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/r6_bench_default.log                        $p/r06_bench_default_command.json.log
cp $g/r6_bench_detail.json                        $p/r06_bench_detail.json
[ -f $g/r6_bench_sparse_legs.json ] && cp $g/r6_bench_sparse_legs.json $p/r06_bench_sparse_legs.json
grep '^{"metric"' $g/r6_bench_under_rocprofv3.log | tail -1 > $p/r06_bench_default_command_under_rocprofv3.json.log
cp $g/r6_bench_detail_under_rocprofv3.json        $p/r06_bench_detail_under_rocprofv3.json
cp $g/r6_bench_kernel_stats_default_command.csv   $p/r06_bench_kernel_stats_default_command.csv
for f in pmc_gstep_mfma_f32 pmc_gstep_waits pmc_siren16w_f16x3 pointwise_backward_timing; do cp $g/r6_$f.txt $p/r06_$f.txt; done
[ -f $g/r6c_sparse_gstep_profile.txt ] && grep -v "amdgpu.ids" $g/r6c_sparse_gstep_profile.txt > $p/r06_sparse_gstep_profile.txt
grep "\[parity\]\|\[dist\]" $g/r6_gpu_tests_full.log | grep -v "print(" > $p/r06_gpu_tests_parity_lines.log
tail -1 $g/r6_gpu_tests_full.log >> $p/r06_gpu_tests_parity_lines.log
ls -la $p/r06_bench_* $p/r06_gpu_tests_parity_lines.log | awk '{print $5, $9}'
