"""generator(z_geo, z_app, **metadata) under torch.no_grad() -- the fake-image render of a discriminator step (train_double_latent_semantic.py:
300-310) -- against the bare fenerf_render_forward time of the same shape: what the Python layer adds.  python tools/exp/forward_api_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from fenerf_amd import _lib, procedural as proc
from fenerf_amd.generators import volumetric_rendering as VR

dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, curriculums = bench.curriculum_generator(spec, sd, dev, "f16x3")
for B, S, N in ((1, 128, 24), (6, 128, 24), (12, 64, 12)):
    md = {**curriculums.extract_metadata(cur, 60000), "img_size": S, "num_steps": N, "nerf_noise": 0.5}
    zg, za = torch.randn(B, 256, device=dev), torch.randn(B, 256, device=dev)
    def run():
        with torch.no_grad():
            return gen(zg, za, **md)
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    nat = gen.siren.native(dev)
    film = proc.film_params(spec, B, seed=1)
    tf = tuple(torch.as_tensor(film[k], device=dev) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    o, d, z, _, _ = VR.sample_rays(B, N, dev, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * S * S, N), device=dev)
    opts = _lib.composite_opts("relu", 0.0)
    for _ in range(3): nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"generator(z_geo, z_app) no-grad, {B} x {S}x{S} x {N}+{N}: {ms:.2f} ms per call = {B * S * S / ms / 1e3:.2f} M rays/s; bare fenerf_render_forward of the shape {dt / 10 * 1e3:.2f} ms", flush=True)
