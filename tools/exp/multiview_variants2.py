"""render_multiview's loop with one statement changed at a time.  python tools/exp/multiview_variants2.py"""
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
zg0, za0 = torch.randn((1, 256), device=dev), torch.randn((1, 256), device=dev)
def loop(redraw=True, yaw=True, seed=True, colours=True, hold=True, fresh_kw=True, reps=2):
    for rep in range(reps):
        kw = callers.multiview_kwargs(cur, 256, 2, False) if fresh_kw else KW
        h_mean = kw["h_mean"]
        images, segmaps, t_sf = [], [], 0.0
        torch.cuda.synchronize(); t_all = time.perf_counter()
        for a in (-0.5, -0.25, 0.0, 0.25, 0.5):
            kw["h_mean"] = (a if yaw else 0.0) + h_mean
            if seed: torch.manual_seed(0)
            zg, za = (torch.randn((1, 256), device=dev), torch.randn((1, 256), device=dev)) if redraw else (zg0, za0)
            t0 = time.perf_counter()
            with torch.no_grad(): img, _ = gen.staged_forward(zg, za, **kw)
            t_sf += time.perf_counter() - t0
            if hold: images.append(img[:, -3:])
            if colours: segmaps.append(callers.mask2color(img[:, :-3], dev) / 255.0)
        kw["h_mean"] = h_mean
        torch.cuda.synchronize(); ms = (time.perf_counter() - t_all) * 1e3
    return ms, t_sf * 1e3
KW = callers.multiview_kwargs(cur, 256, 2, False)
for _ in range(3): loop()
print("as written          : %.1f ms per call, staged_forward %.1f" % loop())
print("latents not redrawn : %.1f ms per call, staged_forward %.1f" % loop(redraw=False))
print("yaw fixed           : %.1f ms per call, staged_forward %.1f" % loop(yaw=False))
print("no manual_seed      : %.1f ms per call, staged_forward %.1f" % loop(seed=False))
print("no colours          : %.1f ms per call, staged_forward %.1f" % loop(colours=False))
print("results not held    : %.1f ms per call, staged_forward %.1f" % loop(hold=False))
print("one kwargs dict     : %.1f ms per call, staged_forward %.1f" % loop(fresh_kw=False))
print("as written          : %.1f ms per call, staged_forward %.1f" % loop())
print({k: (v if not torch.is_tensor(v) else "tensor") for k, v in KW.items()})
