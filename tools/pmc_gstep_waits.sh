#!/bin/bash
# Per-wave time split (issuing / issue-stalled / parked) of the generator-step kernels: one rocprofv3 --pmc pass over tools/bench_gstep.py
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_gstep_waits; mkdir -p gpurun_out/pmc_gstep_waits
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gstep_waits/p1 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_gstep.py --B 1 --size 128 --skip-eager --iters 2) > gpurun_out/pmc_gstep_waits/p1.log 2>&1
python - <<'PY' > gpurun_out/pmc_gstep_waits/summary.txt 2>&1
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_gstep_waits/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "fenerf::" not in name:
            continue
        short = name.split("fenerf::")[1].split("(")[0][:60]
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,n,wave_cycles(quads),active,wait_inst,wait_any,wait_inst_lds,valu_per_mfma")
for k in sorted(agg):
    a = {c: sum(v) / len(v) for c, v in agg[k].items()}
    wc = a.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0:
        continue
    n = len(agg[k]["SQ_WAVE_CYCLES"])
    print(f"{k},{n},{wc:.4g},{a.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f},{a.get('SQ_WAIT_INST_ANY',0)/wc:.3f},{a.get('SQ_WAIT_ANY',0)/wc:.3f},"
          f"{a.get('SQ_WAIT_INST_LDS',0)/wc:.3f},{a.get('SQ_INSTS_VALU',0)/max(a.get('SQ_INSTS_MFMA',0),1):.2f}")
PY
find gpurun_out/pmc_gstep_waits -type f -size +4M -delete
cat gpurun_out/pmc_gstep_waits/summary.txt | head -40
