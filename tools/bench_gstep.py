"""Generator-step benchmark (forward + backward through DoubleImplicitGenerator3d.forward_with_frequencies) on one GPU:
this package's native differentiable path vs the same maths as eager PyTorch ops (what the reference runs: ~60 ATen
launches per SIREN call recorded and replayed by autograd), in fp32 and under torch.cuda.amp.autocast like the
reference's training loop (train_double_latent_semantic.py:279).  Measurement tool only -- imports the test oracle for
the eager restatement.

    python tools/bench_gstep.py [--B 4] [--size 64] [--steps 24] [--iters 5]
"""
import argparse
import functools
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd import _lib, native, procedural as proc          # noqa: E402
from fenerf_amd.generators import generators as G                # noqa: E402
from fenerf_amd.generators import volumetric_rendering as VR     # noqa: E402
from fenerf_amd.siren import siren as S                          # noqa: E402
from oracle import fenerf_oracle_grad as OG                      # noqa: E402

DEV = "cuda:0"


def timed(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--grid", type=int, default=96)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--skip-eager", action="store_true")
    ap.add_argument("--grad-precision", choices=["f32", "tape16", "amp", "amp16"], default="f32", help="weight-gradient operands (siren.grad_precision)")
    a = ap.parse_args()
    B, S_, N, H = a.B, a.size, a.steps, a.H
    spec = proc.model_spec("texture", hidden_dim=H, grid_size=a.grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=4, sigma_gain=150.0, with_mapping=False)
    mod = S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=H, z_geo_dim=8, z_app_dim=8, output_dim=22)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
    mod.load_state_dict(tsd, strict=False)
    mod = mod.to(DEV)
    mod.grad_precision = a.grad_precision
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=H), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    film = proc.film_params(spec, B, seed=4)
    film_t = {k: torch.tensor(v, device=DEV).requires_grad_(True) for k, v in film.items()}
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
    w = torch.randn((B, 21, S_, S_), device=DEV)
    params = [p for n, p in mod.named_parameters() if "mapping_network" not in n]
    res = {"config": {"B": B, "img_size": S_, "num_steps": f"{N}+{N}", "H": H, "grid": a.grid, "points": B * S_ * S_ * 2 * N}}

    def native_step(repack):
        for p in params:
            p.grad = None
        if repack:
            with torch.no_grad():
                params[0].add_(0)        # bumps the version counter like optimizer.step(): weights are re-packed
        px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
        (px * w).sum().backward()

    def native_fwd_only():
        with torch.no_grad():
            gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)

    torch.cuda.reset_peak_memory_stats()
    res["native_fwd_bwd_ms"] = timed(lambda: native_step(False), a.iters)
    res["native_peak_GB"] = torch.cuda.max_memory_allocated() / 2**30
    res["native_fwd_bwd_with_repack_ms"] = timed(lambda: native_step(True), a.iters)
    res["native_nograd_render_ms"] = timed(native_fwd_only, a.iters)
    # inversion step (inverse_render_double_semantic.py:324-410): weights frozen, only the FiLM offsets take gradients
    for p in params:
        p.requires_grad_(False)
    res["native_inversion_fwd_bwd_ms"] = timed(lambda: native_step(False), a.iters)
    for p in params:
        p.requires_grad_(True)

    if not a.skip_eager:
        sdt = {k: torch.tensor(v, device=DEV).requires_grad_(True) for k, v in sd.items()}
        film_e = {k: torch.tensor(v, device=DEV).requires_grad_(True) for k, v in film.items()}
        R = S_ * S_

        def eager_step(amp):
            for t in list(sdt.values()) + list(film_e.values()):
                t.grad = None
            origins, dirs, z_vals, _, _ = VR.sample_rays(B, N, gen.device, kw["fov"], (S_, S_), kw["ray_start"], kw["ray_end"], kw["h_stddev"],
                                                         kw["v_stddev"], kw["h_mean"], kw["v_mean"], kw["sample_dist"], draws=gen.draws)
            noise_c = torch.randn((B * R, N), device=DEV); u = torch.rand((B * R, N), device=DEV)
            noise_f = torch.randn((B * R, 2 * N), device=DEV)
            z_c = z_vals.reshape(B, R, N)
            rd = dirs.unsqueeze(2).expand(-1, -1, N, -1).reshape(B, R * N, 3)
            args = (film_e["freq_geo"], film_e["phase_geo"], film_e["freq_app"], film_e["phase_app"])
            with torch.autocast("cuda", enabled=amp):
                pts_c = (origins.unsqueeze(2) + dirs.unsqueeze(2) * z_c.unsqueeze(-1)).reshape(B, R * N, 3)
                coarse = OG.siren_forward(sdt, spec, pts_c, rd, *args)
                with torch.no_grad():
                    _, _, w_c = OG.composite(coarse.detach().float().reshape(B * R, N, 22), z_c.reshape(B * R, N), noise_c, noise_std=0.2)
                    z_f = native.resample(z_c.reshape(B * R, N), w_c.contiguous(), u).reshape(B, R, N)   # (sample_pdf: not the point here)
                pts_f = (origins.unsqueeze(2) + dirs.unsqueeze(2) * z_f.unsqueeze(-1)).reshape(B, R * N, 3)
                fine = OG.siren_forward(sdt, spec, pts_f, rd, *args)
                rgb, _, _ = OG.merge_composite(fine.reshape(B * R, N, 22).float(), coarse.reshape(B * R, N, 22).float(), z_f.reshape(B * R, N),
                                               z_c.reshape(B * R, N), noise_f, noise_std=0.2)
            px = rgb.reshape(B, S_, S_, 21).permute(0, 3, 1, 2) * 2 - 1
            (px * w).sum().backward()

        def eager_inversion():
            for t in sdt.values():
                t.requires_grad_(False)
            try:
                eager_step(True)
            finally:
                for t in sdt.values():
                    t.requires_grad_(True)

        res["eager_amp_inversion_fwd_bwd_ms"] = timed(eager_inversion, max(2, a.iters // 2), warm=1)
        for amp, key in ((False, "eager_fp32_fwd_bwd_ms"), (True, "eager_amp_fwd_bwd_ms")):
            torch.cuda.reset_peak_memory_stats()
            try:
                res[key] = timed(lambda: eager_step(amp), max(2, a.iters // 2), warm=1)
                res[key.replace("_ms", "_peak_GB")] = torch.cuda.max_memory_allocated() / 2**30
            except torch.cuda.OutOfMemoryError as e:
                res[key] = "OOM"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
