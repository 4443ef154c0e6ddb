import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from fenerf_amd import curriculums
from fenerf_amd.generators import generators
from fenerf_amd.siren import siren
dev = torch.device("cuda:0")
cur = curriculums.CelebA_double_semantic_texture_embedding_256_dim_96
gen = generators.DoubleImplicitGenerator3d(getattr(siren, cur["model"]), 256, 256, 22).to(dev)
gen.set_device(dev)
md = {**curriculums.extract_metadata(cur, 60000), "nerf_noise": 0, "psi": 0.7, "img_size": 256, "num_steps": 48, "max_batch_size": 10 ** 9}
zg, za = torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)
def run():
    with torch.no_grad():
        return gen.staged_forward(zg, za, **md)
for _ in range(3): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(8):
        with record_function(f"CALL{i}"):
            run()
    torch.cuda.synchronize()
evs = prof.events()
calls = sorted([e for e in evs if e.name.startswith("CALL")], key=lambda e: e.time_range.start)
for c in calls:
    inside = [e for e in evs if e.device_type.name == "CPU" and e.time_range.start >= c.time_range.start and e.time_range.end <= c.time_range.end and not e.name.startswith("CALL")]
    top = sorted(inside, key=lambda e: -e.cpu_time_total)[:4]
    k = [e for e in evs if e.device_type.name == "CUDA" and "siren16w" in e.name and e.time_range.start >= c.time_range.start and e.time_range.start <= c.time_range.end + 100000]
    print(f"{c.name}: {c.cpu_time_total / 1e3:7.2f} ms; siren kernels {[round(e.device_time_total / 1e3, 2) for e in k]}; top cpu:", [(e.name[:28], round(e.cpu_time_total / 1e3, 2)) for e in top])
