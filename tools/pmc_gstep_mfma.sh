#!/bin/bash
# MFMA-pipe utilisation of the generator-step kernels (one rocprofv3 --pmc pass over tools/bench_gstep.py)
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_gstep_mfma; mkdir -p gpurun_out/pmc_gstep_mfma
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gstep_mfma/p1 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_gstep.py ${GSTEP_ARGS:---B 2 --size 64} --skip-eager --iters 2) > gpurun_out/pmc_gstep_mfma/p1.log 2>&1
python - <<'PY' > gpurun_out/pmc_gstep_mfma/summary.txt 2>&1
import csv, glob
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_gstep_mfma/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "siren" not in name or "fenerf::" not in name:
            continue
        short = name.split("fenerf::")[1].split("(")[0][:60]
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,counter,avg_per_dispatch,n")
for k in sorted(agg):
    for c, v in sorted(agg[k].items()):
        print(f"{k},{c},{sum(v)/len(v):.6g},{len(v)}")
    a = agg[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a:
        busy = sum(a["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(a["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(a["GRBM_GUI_ACTIVE"]) / len(a["GRBM_GUI_ACTIVE"])
        # SQ_VALU_MFMA_BUSY_CYCLES = cycles summed over the 1024 SIMDs (it equals SQ_INSTS_MFMA x 32 for the bf16 / fp16 MFMAs);
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        print(f"# {k}: MFMA pipe utilisation = MFMA busy / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) = {busy / (gui / 8 * 1024):.3f}")
PY
find gpurun_out/pmc_gstep_mfma -type f -size +4M -delete
cat gpurun_out/pmc_gstep_mfma/summary.txt | head -70
