#!/bin/bash
# The shader clock the SIREN forward kernel actually runs at, on real and on all-zero weights (same instruction stream):
# GRBM_GUI_ACTIVE (cycles the chip was busy, summed over the 8 XCDs) / 8 / the dispatch's duration from the kernel trace.
# usage: gpurun --timeout 900 -- 'bash tools/clock_probe.sh'
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gstep --no-f32 --no-sweep64"
rm -rf gpurun_out/clock; mkdir -p gpurun_out/clock
for z in real zero; do
  ZW=""; [ $z = zero ] && ZW=1
  (cd /tmp && FENERF_BENCH_ZERO_WEIGHTS=$ZW timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/clock/$z -o c -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 10 --warmup 3 $Q) > gpurun_out/clock/$z.log 2>&1
done
python - <<'PY' | tee gpurun_out/clock/summary.txt
import csv, glob
for z in ("real", "zero"):
    trace = {}
    for f in glob.glob(f"gpurun_out/clock/{z}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "siren16w_kernel" in r["Kernel_Name"]:
                trace[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    cyc = {}
    for f in glob.glob(f"gpurun_out/clock/{z}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "siren16w_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cyc[r["Dispatch_Id"]] = float(r["Counter_Value"])
    ids = sorted(set(trace) & set(cyc), key=int)[6:]          # skip the warm-up launches
    if not ids:
        print(z, "no dispatches matched"); continue
    ghz = [cyc[i] / 8 / trace[i] / 1e9 for i in ids]
    ms = [trace[i] * 1e3 for i in ids]
    print(f"{z} weights: {len(ids)} launches, kernel {sum(ms)/len(ms):.4f} ms, busy cycles per XCD {sum(cyc[i] for i in ids)/len(ids)/8:.4g}, "
          f"shader clock {sum(ghz)/len(ghz):.3f} GHz (min {min(ghz):.3f}, max {max(ghz):.3f})")
PY
find gpurun_out/clock -type f -size +2M -delete
