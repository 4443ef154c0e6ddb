"""Per-launch timeline of ONE generator step from a rocprofv3 --kernel-trace CSV (tools/gpu_r3.sh gtimeline): every kernel of the last
complete step, or of step K counted from the first (steps are delimited by ray_setup_kernel launches) with its duration and the idle gap in front of it, then the totals per
kernel family.  python tools/gstep_timeline.py <dir with *kernel_trace.csv> [K]"""
import csv
import glob
import sys
from collections import OrderedDict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "ray_setup_kernel" in r[2]]
if len(marks) < 3:
    sys.exit("fewer than three steps in the trace")
lo, hi = (marks[-2], marks[-1]) if len(sys.argv) < 3 else (marks[int(sys.argv[2])], marks[int(sys.argv[2]) + 1])
step = rows[lo:hi]


def short(n):
    if "fenerf::" in n:
        n = n.split("fenerf::", 1)[1]
    n = n.split("(")[0]
    return n[:70]


print(f"# one generator step: {len(step)} launches, {(step[-1][1] - step[0][0]) / 1e6:.3f} ms from the first launch's start to the last one's end "
      f"(next step starts {(rows[hi][0] - step[-1][1]) / 1e3:.1f} us later)")
print("# start_us  dur_us  gap_us  kernel")
t0 = step[0][0]
prev = None
fam = OrderedDict()
gaps = 0.0
for s, e, n in step:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    gaps += max(gap, 0.0)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {short(n)}")
    k = short(n)
    d = fam.setdefault(k, [0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    prev = max(prev, e) if prev is not None else e
print(f"# idle between launches: {gaps / 1e3:.3f} ms")
print("# totals per kernel: launches, ms")
for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:5d} {d / 1e3:8.3f}  {k}")
