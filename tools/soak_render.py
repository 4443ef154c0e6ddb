"""Soak of the fused render (fenerf_render_forward: FiLM pre-pass in the coarse SIREN launch's prologue, coarse weights + resampling in one
wave per ray, fine SIREN, merge + composite): every shape is rendered `--iters` times on fresh FiLM parameters / rays and each render
twice -- the two must be bit-identical (a race between the workgroups that prepare the same image's FiLM block, or between the prologue's
stores and the LDS-DMA that reads them back, would show up as a difference) and within 1e-5 of the render whose FiLM pre-pass and resampling ran as their own
launches (stage-by-stage path through fenerf_siren_forward_rays / fenerf_composite / fenerf_resample / fenerf_merge_composite).  python tools/soak_render.py [--iters 100]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd import _lib, native, procedural as proc                      # noqa: E402
from fenerf_amd.generators import volumetric_rendering as VR                 # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=100)
a = ap.parse_args()
dev = "cuda:0"
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
opts = _lib.composite_opts("relu", 0.0, fill_mode="seg_padding_background", fill_color="black")
bad = 0
for precision in ("f16x3", "f32"):
    nat = native.NativeModel(sd, spec, dev, precision)
    for (B, S, N) in ((1, 128, 24), (4, 64, 24), (7, 17, 13), (24, 16, 12)):
        R = S * S
        worst = 0.0
        n_it = a.iters if precision == "f16x3" else max(5, a.iters // 10)
        for it in range(n_it):
            film = proc.film_params(spec, B, seed=100 + it)
            tf = tuple(torch.as_tensor(film[k], device=dev) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
            torch.manual_seed(it)
            o, d, z, _, _ = VR.sample_rays(B, N, dev, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
            u = torch.rand((B * R, N), device=dev)
            r1 = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True, want_weights=True)
            r2 = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True, want_weights=True)
            if not (torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1]) and torch.equal(r1[2], r2[2])):
                bad += 1
            if it % 10 == 0:      # the same render stage by stage (FiLM pre-pass as its own launch, resampling as its own launch)
                coarse = nat.siren_forward_rays(o, d, z, *tf)
                _, _, w, _ = native.composite(coarse, z, None, _lib.composite_opts("relu"))
                zf = native.resample(z.reshape(B * R, N), w.reshape(B * R, N), u)
                fine = nat.siren_forward_rays(o, d, zf.reshape(B, R, N), *tf)
                rgb = native.merge_composite(fine.reshape(B * R, N, -1), coarse.reshape(B * R, N, -1), zf, z.reshape(B * R, N), None, opts)[0]
                worst = max(worst, float((rgb.reshape(r1[0].shape) - r1[0]).abs().max()))
        print(f"{precision} B={B} {S}x{S} N={N}+{N}: {n_it} x 2 renders, fused vs stage-by-stage max|diff| {worst:.1e}", flush=True)
        if worst > 1e-5:
            bad += 1
print("soak_render:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(0 if bad == 0 else 1)
