"""Where the sparse generator step's time goes (opt-in siren.sparse_backward): torch.profiler kernel table of a few steps at the bench's
shape (1 x 128 x 128 x 24+24, H = 256 + 96^3 grid, procedural density).   python tools/exp/sparse_gstep_profile.py [dense]"""
import functools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from fenerf_amd import procedural as proc
from fenerf_amd.generators import generators as G
from fenerf_amd.siren import siren as S_

dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
mod = S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=256, z_geo_dim=256, z_app_dim=256, output_dim=22)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
mod.load_state_dict(tsd, strict=False)
mod.sparse_backward = "dense" not in sys.argv
gen = G.DoubleImplicitGenerator3d(functools.partial(S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=256), 256, 256, 22)
gen.siren = mod
gen = gen.to(dev); gen.device = dev; gen.siren.device = dev
film = {k: torch.tensor(v, device=dev).requires_grad_(True) for k, v in proc.film_params(spec, 1, seed=5).items()}
kw = dict(img_size=128, fov=12, ray_start=0.88, ray_end=1.12, num_steps=24, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
          hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
w = torch.randn((1, 21, 128, 128), device=dev)
params = [p for n, p in mod.named_parameters() if "mapping_network" not in n]
def step():
    for p in params: p.grad = None
    px, _ = gen.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **kw)
    (px * w).sum().backward()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows if e.device_type.name != "CPU" or e.device_time_total) / 5e3
print(f"device time per step (sum of kernels): {sum(e.self_device_time_total for e in rows) / 5e3:.3f} ms")
for e in rows[:28]:
    if e.self_device_time_total > 0:
        print(f"{e.self_device_time_total / 5e3:8.3f} ms  x{e.count // 5:3d}  {e.key[:110]}")

# ---- timeline of ONE step: device intervals in start order with the idle gap before each (where the host-side sync / glue shows)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof1:
    step()
    torch.cuda.synchronize()
ev = sorted([e for e in prof1.events() if e.device_type.name == "CUDA" and e.device_time_total > 0], key=lambda e: e.time_range.start)
t0, prev_end, gaps = ev[0].time_range.start, ev[0].time_range.start, 0.0
print(f"timeline of one step ({len(ev)} device intervals); gaps > 15 us:")
for e in ev:
    gap = e.time_range.start - prev_end
    if gap > 15:
        gaps += gap
        print(f"  +{(e.time_range.start - t0) / 1e3:7.3f} ms  idle {gap:6.0f} us before {e.name[:80]}")
    prev_end = max(prev_end, e.time_range.end)
print(f"span {(prev_end - t0) / 1e3:.3f} ms, idle in gaps > 15 us: {gaps / 1e3:.3f} ms")

# ---- steady state: five back-to-back steps, device busy time (union of the intervals) against the span
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof5:
    for _ in range(5): step()
    torch.cuda.synchronize()
ev = sorted([e for e in prof5.events() if e.device_type.name == "CUDA" and e.device_time_total > 0], key=lambda e: e.time_range.start)
busy, cur_s, cur_e = 0.0, ev[0].time_range.start, ev[0].time_range.end
big = []
for e in ev[1:]:
    if e.time_range.start > cur_e:
        if e.time_range.start - cur_e > 40: big.append((e.time_range.start - cur_e, e.name[:60]))
        busy += cur_e - cur_s
        cur_s, cur_e = e.time_range.start, e.time_range.end
    else:
        cur_e = max(cur_e, e.time_range.end)
busy += cur_e - cur_s
span = cur_e - ev[0].time_range.start
print(f"five steps back to back: span {span / 5e3:.3f} ms per step, device busy {busy / 5e3:.3f} ms per step, idle {100 * (1 - busy / span):.1f} %")
from collections import Counter
c = Counter()
for g_, n_ in big: c[n_] += g_
for n_, g_ in c.most_common(8): print(f"   idle {g_ / 5e3:6.3f} ms per step before {n_}")
