"""Forward render throughput by hidden width (128 x 128 rays, 24+24 samples, 96^3 grid, f16x3 and exact fp32): the headline's step at
H = 64 ... 256, incl. the widths instantiated in round 5 (96, 192).  usage: python tools/exp/width_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fenerf_amd import _lib, native, procedural as proc                   # noqa: E402
from fenerf_amd.generators import volumetric_rendering as VR            # noqa: E402

dev = torch.device("cuda", 0)
S, N, B = 128, 24, 1
opts = _lib.composite_opts("relu", 0.0, False, False, False, "seg_padding_background", "white")
for H in (64, 96, 128, 192, 256):
    spec = proc.model_spec("texture", hidden_dim=H, grid_size=96)
    sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
    film = proc.film_params(spec, B, seed=1)
    tf = tuple(torch.as_tensor(film[k], device=dev) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    torch.manual_seed(3)
    o, d, z, _, _ = VR.sample_rays(B, N, dev, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * S * S, N), device=dev)
    row = [f"H={H:3d}"]
    for prec in ("f16x3", "f32"):
        nat = native.NativeModel(sd, spec, dev, prec)
        for _ in range(3):
            nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        row.append(f"{prec}: {ms:.3f} ms/step = {B * S * S / ms / 1e3:.2f} M rays/s")
        nat.close()
    print("  ".join(row), flush=True)
