"""Device time of every launch group of the fused render (fenerf_phase_timing) at the bench workload: python tools/time_render_phases.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd import _lib, native, procedural as proc                      # noqa: E402
from fenerf_amd.generators import volumetric_rendering as VR                 # noqa: E402

dev = "cuda:0"
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
nat = native.NativeModel(sd, spec, dev, "f16x3")
B, S, N = 1, 128, 24
film = proc.film_params(spec, B, seed=0)
tf = tuple(torch.as_tensor(film[k], device=dev) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
torch.manual_seed(0)
o, d, z, _, _ = VR.sample_rays(B, N, dev, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
u = torch.rand((B * S * S, N), device=dev)
opts = _lib.composite_opts("relu", 0.0, fill_mode="seg_padding_background", fill_color="black")
for _ in range(5):
    nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
iters = 30
with native.phase_timing() as t:
    for _ in range(iters):
        nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
print(" ".join(f"{k}={v / iters * 1e3:.1f}us/{t.calls[k] // iters}" for k, v in sorted(t.ms.items())))
