"""Where the time of generator.staged_forward at 256 x 256 x 48+48 goes (BASELINE.json configs[4]): torch.profiler kernel table + host gaps.
    python tools/exp/staged_profile.py [size] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from fenerf_amd import curriculums
from fenerf_amd.generators import generators
from fenerf_amd.siren import siren

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
cur = curriculums.CelebA_double_semantic_texture_embedding_256_dim_96
gen = generators.DoubleImplicitGenerator3d(getattr(siren, cur["model"]), 256, 256, 22).to(dev)
gen.set_device(dev)
md = {**curriculums.extract_metadata(cur, 60000), "nerf_noise": 0, "psi": 0.7, "img_size": size, "num_steps": steps, "max_batch_size": 10 ** 9}
zg, za = torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)
def run():
    with torch.no_grad():
        return gen.staged_forward(zg, za, **md)
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("ms per call:", [round(t, 2) for t in ts])
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type.name == "CUDA" and e.device_time_total > 0]
agg = {}
for e in ev:
    a = agg.setdefault(e.name[:110], [0.0, 0]); a[0] += e.device_time_total / 1e3; a[1] += 1
print(f"device time (sum): {sum(v[0] for v in agg.values()):.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {v[0]:8.3f} ms x {v[1]:3d}  {k}")
ev.sort(key=lambda e: e.time_range.start)
t0, prev = ev[0].time_range.start, ev[0].time_range.end
for e in ev[1:]:
    if e.time_range.start - prev > 100:
        print(f"  +{(e.time_range.start - t0) / 1e3:7.3f} ms idle {e.time_range.start - prev:6.0f} us before {e.name[:70]}")
    prev = max(prev, e.time_range.end)
print(f"span {(prev - t0) / 1e3:.3f} ms")
cpu = sorted([e for e in prof.events() if e.device_type.name == "CPU"], key=lambda e: -e.cpu_time_total)[:10]
for e in cpu: print(f"  cpu {e.cpu_time_total / 1e3:8.3f} ms  {e.name[:90]}")
