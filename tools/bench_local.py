"""SPATIALSIRENGRID timing / PMC target: the fused launch (per-point mapping network + FiLM-SIREN,
fenerf_siren_forward_local) on 196,608 points at H = 256, against torch mapping network + fenerf_siren_forward_pointwise (18 KB of
FiLM parameters per point through HBM).  python tools/bench_local.py [--points 196608] [--iters 5] [--explicit]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd.siren import siren as S          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=196608)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--explicit", action="store_true", help="also time the explicit per-point FiLM route")
a = ap.parse_args()
DEV = "cuda:0"
torch.manual_seed(3)
mod = S.SPATIALSIRENGRID(input_dim=3, z_dim=16, hidden_dim=256, output_dim=4).to(DEV).eval()
mod.device = torch.device(DEV)
g = torch.Generator(device=DEV).manual_seed(4)
pts = (torch.rand((1, a.points, 3), device=DEV, generator=g) - 0.5) * 0.24
dirs = torch.nn.functional.normalize(torch.randn((1, a.points, 3), device=DEV, generator=g), dim=-1)
lat = torch.randn((1, 32, 32, 32), device=DEV, generator=g)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters * 1e3


with torch.no_grad():
    sampled = mod.sample_local_latents(lat, mod.gridwarper(pts))
    local = mod.get_local_coordinates(pts, 32, preserve_y=False)
    nat = mod.native_local(DEV)
    ms = timed(lambda: nat.forward(local, dirs, sampled))
    print(f"fused launch: {ms:.3f} ms for {a.points} points = {a.points * 3.56e6 / ms / 1e9:.0f} TFLOP/s (exact fp32 MFMA), "
          f"{a.points * 168 / ms / 1e6:.1f} GB/s of compulsory HBM traffic (152 B in + 16 B out per point)")
    if a.explicit:
        def route():
            f, p = mod.mapping_network(sampled)
            return mod.forward_with_frequencies_phase_shifts(local, f, p, dirs)
        ms2 = timed(route)
        print(f"torch mapping network + explicit per-point FiLM (fenerf_siren_forward_pointwise): {ms2:.3f} ms")
