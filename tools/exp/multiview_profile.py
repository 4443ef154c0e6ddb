"""What a view of callers.render_multiview spends outside its kernels.  python tools/exp/multiview_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from fenerf_amd import callers, procedural as proc
dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, _ = bench.curriculum_generator(spec, sd, dev, "f16x3")
gen.eval()
for _ in range(2): callers.render_multiview(gen, cur, 0, dev)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    callers.render_multiview(gen, cur, 0, dev)
    torch.cuda.synchronize()
ev = sorted([e for e in prof.events() if e.device_type.name == "CUDA" and e.device_time_total > 0], key=lambda e: e.time_range.start)
gaps = []
for a, b in zip(ev, ev[1:]):
    g = b.time_range.start - a.time_range.end
    if g > 200: gaps.append((round(g / 1e3, 2), a.name[:40], "->", b.name[:40]))
print("device idle gaps > 0.2 ms:", *gaps, sep="\n  ")
tot = {}
for e in prof.events():
    if e.device_type.name == "CPU":
        a = tot.setdefault(e.name[:50], [0.0, 0]); a[0] += e.self_cpu_time_total / 1e3; a[1] += 1
print("self CPU by op:", sorted(((k, round(v[0], 2), v[1]) for k, v in tot.items()), key=lambda t: -t[1])[:12])
