"""Generator step (1 x 128 x 128 x 24+24, H = 256 + 96^3 grid) over backward schedules: serial chunks of several sizes, and the overlapped
schedule of siren/autograd.py (weight gradients of chunk i beside the chain of chunk i + 1) over chunk sizes and CU splits.
    python tools/overlap_sweep.py [--quick]          -> one line per configuration: ms per step, peak GB      (profiles/r04_gstep_overlap.md)"""
import functools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd import procedural as proc                      # noqa: E402
from fenerf_amd.generators import generators as G              # noqa: E402
from fenerf_amd.siren import autograd as SA, siren as S_       # noqa: E402

dev = torch.device("cuda", 0)
B, S, N, H = 1, 128, 24, 256
spec = proc.model_spec("texture", hidden_dim=H, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
mod = S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=H, z_geo_dim=256, z_app_dim=256, output_dim=22)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
mod.load_state_dict(tsd, strict=False)
mod.precision = "f16x3"
gen = G.DoubleImplicitGenerator3d(functools.partial(S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=H), 256, 256, 22)
gen.siren = mod
gen = gen.to(dev)
gen.device = dev
gen.siren.device = dev
film = {k: torch.tensor(v, device=dev).requires_grad_(True) for k, v in proc.film_params(spec, B, seed=5).items()}
kw = dict(img_size=S, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
          hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
w = torch.randn((B, 21, S, S), device=dev)
params = [p for n, p in mod.named_parameters() if "mapping_network" not in n]
bump = min(params, key=lambda t: t.numel())


def step():
    for p in params:
        p.grad = None
    with torch.no_grad():
        bump.add_(0)
    px, _ = gen.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **kw)
    (px * w).sum().backward()


def timed(iters=6):
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, torch.cuda.max_memory_allocated() / 2**30


quick = "--quick" in sys.argv
for _ in range(3):
    step()
print("# schedule, chunk points, chain CU share -> ms per generator step, peak GB")
for chunk in ((196608, 786432) if quick else (98304, 196608, 393216, 786432)):
    SA.OVERLAP_WGRAD, SA.BACKWARD_CHUNK_POINTS = False, chunk
    ms, gb = timed()
    print(f"serial      chunk {chunk:7d}                 : {ms:7.3f} ms  {gb:5.1f} GB", flush=True)
for chunk in ((131072, 196608) if quick else (65536, 98304, 131072, 196608, 262144)):
    for frac in ((0.75, 0.875) if quick else (0.5, 0.625, 0.75, 0.8125, 0.875, 0.9375)):
        SA.OVERLAP_WGRAD, SA.BACKWARD_CHUNK_POINTS, SA.CHAIN_CUS_FRACTION = True, chunk, frac
        ms, gb = timed()
        print(f"overlapped  chunk {chunk:7d}  chain CUs {frac:6.4f} : {ms:7.3f} ms  {gb:5.1f} GB", flush=True)
