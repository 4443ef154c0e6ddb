"""How does the torch-CPU oracle scale with intra-op threads on the GPU box's host?  (round 6: 256 threads ran 128x128x24+24 in 42 s,
8 threads of the build container in 9 s)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fenerf_amd import procedural as proc
from oracle import fenerf_oracle_torch as OT
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print(torch.__config__.parallel_info())
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
tsd = OT.state_to_torch(sd)
film = proc.film_params(spec, 1, seed=1000)
for th in (4, 8, 16, 32, 64, 128, 256):
    if th > os.cpu_count():
        break
    torch.set_num_threads(th)
    bench._oracle_torch_run(spec, tsd, film, 64, 12, False, 1, 2400000)
    t = min(bench._oracle_torch_run(spec, tsd, film, 64, 12, False, 2 + i, 2400000) for i in range(2))
    t2 = bench._oracle_torch_run(spec, tsd, film, 64, 24, True, 5, 2400000)
    print(f"threads {th}: 64x64x12 coarse {t:.3f} s ({4096 / t:.0f} rays/s); 64x64x24+24 {t2:.3f} s ({4096 / t2:.0f} rays/s)", flush=True)
