#!/bin/bash
# HBM traffic of the generator-step kernels: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/bench_gstep.py
# usage: gpurun --timeout 900 -- 'bash tools/pmc_gstep.sh'
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_gstep; mkdir -p gpurun_out/pmc_gstep
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gstep/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_gstep.py ${GSTEP_ARGS:---B 2 --size 64} --skip-eager --iters 2) > gpurun_out/pmc_gstep/p$i.log 2>&1
  echo "pass $i ($set) exit $?" >> gpurun_out/pmc_gstep/summary.txt
done
python - <<'PY' > gpurun_out/pmc_gstep/gstep_pmc_summary.txt 2>&1
import csv, glob, os
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_gstep/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "fenerf::" not in name:
            continue
        short = name.split("fenerf::")[1].split("(")[0][:60]
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,counter,avg_KiB_per_dispatch,MB_per_dispatch,n")
for k in sorted(agg):
    for c, v in sorted(agg[k].items()):
        a = sum(v) / len(v)
        print(f"{k},{c},{a:.6g},{a * 1024 / 1e6:.1f},{len(v)}")
PY
find gpurun_out/pmc_gstep -type f -size +4M -delete
cat gpurun_out/pmc_gstep/gstep_pmc_summary.txt | head -60
