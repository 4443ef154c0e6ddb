"""Host-clock breakdown of callers.render_multiview's own statements (5 views, 256 x 256 x 48+48), steady state.
    python tools/exp/multiview_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fenerf_amd import callers, procedural as proc
dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, _ = bench.curriculum_generator(spec, sd, dev, "f16x3")
gen.eval()
for _ in range(4): callers.render_multiview(gen, cur, 0, dev)
torch.cuda.synchronize()
for rep in range(3):
    T = {}
    def tick(name, t0):
        T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()
    t_all = time.perf_counter()
    t0 = time.perf_counter()
    kw = callers.multiview_kwargs(cur, 256, 2, False); h_mean = kw["h_mean"]; zg_dim, za_dim = callers._latent_dims(gen); t0 = tick("kwargs", t0)
    images, segmaps = [], []
    for a in (-0.5, -0.25, 0.0, 0.25, 0.5):
        kw["h_mean"] = a + h_mean
        torch.manual_seed(0); t0 = tick("manual_seed", t0)
        z_geo = torch.randn((1, zg_dim), device=dev); z_app = torch.randn((1, za_dim), device=dev); t0 = tick("randn", t0)
        with torch.no_grad():
            img, _ = gen.staged_forward(z_geo, z_app, **kw)
        t0 = tick("staged_forward", t0)
        images.append(img[:, -3:]); t0 = tick("append", t0)
        segmaps.append(callers.mask2color(img[:, :-3], dev) / 255.0); t0 = tick("mask2color", t0)
    r = torch.cat(images), torch.cat(segmaps); t0 = tick("cat", t0)
    print(f"manual copy of render_multiview: {(time.perf_counter() - t_all) * 1e3:.1f} ms;", {k: round(v, 2) for k, v in T.items()}, flush=True)
    t0 = time.perf_counter(); callers.render_multiview(gen, cur, 0, dev); print(f"callers.render_multiview: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
