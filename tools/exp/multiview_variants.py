"""Which statement of render_multiview's loop slows the NEXT staged_forward down?  python tools/exp/multiview_variants.py"""
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
kw = callers.multiview_kwargs(cur, 256, 2, False)
zg, za = torch.randn((1, 256), device=dev), torch.randn((1, 256), device=dev)
def loop(hold, colours, seed, n=10):
    kept, t_sf = [], 0.0
    torch.cuda.synchronize(); t_all = time.perf_counter()
    for _ in range(n):
        if seed: torch.manual_seed(0)
        t0 = time.perf_counter()
        with torch.no_grad(): img, _ = gen.staged_forward(zg, za, **kw)
        t_sf += time.perf_counter() - t0
        if hold: kept.append(img[:, -3:])
        if colours == "device": kept.append(callers.mask2color(img[:, :-3], dev))
        elif colours == "numpy": kept.append(callers.mask2color(img[:, :-3]))
    torch.cuda.synchronize()
    return (time.perf_counter() - t_all) / n * 1e3, t_sf / n * 1e3
for _ in range(3): loop(True, "device", True, 5)
for hold in (False, True):
    for colours in (None, "numpy", "device"):
        for seed in (False, True):
            a, b = loop(hold, colours, seed)
            print(f"hold {hold!s:5} colours {colours!s:6} seed {seed!s:5}: {a:6.2f} ms per view, staged_forward {b:6.2f}", flush=True)
