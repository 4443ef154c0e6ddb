"""staged_forward at 256 x 256 x 48+48 in segments, each closed by a device synchronize (host clock).  python tools/exp/staged_segments.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fenerf_amd import callers, native, procedural as proc
dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, _ = bench.curriculum_generator(spec, sd, dev, "f16x3")
gen.eval()
kw = callers.multiview_kwargs(cur, 256, 2, False)
zg, za = torch.randn((1, 256), device=dev), torch.randn((1, 256), device=dev)
def seg(T, name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3; return time.perf_counter()
for rep in range(3):
    T = {}
    for _ in range(5):
        t0 = time.perf_counter()
        with torch.no_grad():
            gen.generate_avg_frequencies(); t0 = seg(T, "avg", t0)
            rfg, rpg = gen.siren.geo_mapping_network(zg); rfa, rpa = gen.siren.app_mapping_network(za)
            psi = kw["psi"]
            fg = gen.avg_frequencies_geo + psi * (rfg - gen.avg_frequencies_geo); pg = gen.avg_phase_shifts_geo + psi * (rpg - gen.avg_phase_shifts_geo)
            fa = gen.avg_frequencies_app + psi * (rfa - gen.avg_frequencies_app); pa = gen.avg_phase_shifts_app + psi * (rpa - gen.avg_phase_shifts_app)
            t0 = seg(T, "mapping + truncation", t0)
            k2 = {k: v for k, v in kw.items() if k not in ("img_size", "fov", "ray_start", "ray_end", "num_steps", "h_stddev", "v_stddev", "h_mean", "v_mean", "psi",
                                                          "lock_view_dependence", "max_batch_size", "sample_dist", "hierarchical_sample")}
            px, depth, _, _, _ = gen._render((fg, pg, fa, pa), kw["img_size"], kw["fov"], kw["ray_start"], kw["ray_end"], kw["num_steps"], kw["h_stddev"], kw["v_stddev"],
                                             kw["h_mean"], kw["v_mean"], kw.get("hierarchical_sample", False), kw.get("sample_dist"), False, k2, use_fill=True, third=None)
            t0 = seg(T, "_render", t0)
            d = native.to_host(depth.reshape(1, 256, 256).contiguous()); p = native.to_host(gen._finish_scaled(px, 1, 256)); t0 = seg(T, "epilogue + to_host", t0)
    print({k: round(v / 5, 2) for k, v in T.items()}, "ms per image; sum", round(sum(T.values()) / 5, 2))
t0 = time.perf_counter()
for _ in range(5):
    with torch.no_grad(): gen.staged_forward(zg, za, **kw)
torch.cuda.synchronize(); print("staged_forward:", round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms per image")
import math
for a in (-0.5, -0.25, 0.0, 0.25, 0.5):
    k3 = {**kw, "h_mean": a + math.pi / 2}
    for _ in range(2):
        with torch.no_grad(): gen.staged_forward(zg, za, **k3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        with torch.no_grad(): gen.staged_forward(zg, za, **k3)
    torch.cuda.synchronize(); print(f"yaw offset {a:+.2f}: staged_forward {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per image")
t0 = time.perf_counter()
for rep in range(3):
    for a in (-0.5, -0.25, 0.0, 0.25, 0.5):
        with torch.no_grad(): gen.staged_forward(zg, za, **{**kw, "h_mean": a + math.pi / 2})
torch.cuda.synchronize(); print(f"alternating yaws: {(time.perf_counter() - t0) / 15 * 1e3:.2f} ms per image")
for hold in (False, True):
    for rep in range(2):
        kept = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            with torch.no_grad(): img, _ = gen.staged_forward(zg, za, **kw)
            if hold: kept.append(img[:, -3:])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        del kept
    print(f"results {'held in a list' if hold else 'dropped'}: {ms:.2f} ms per image")
for rep in range(2):
    kept = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        torch.manual_seed(0)
        with torch.no_grad(): img, _ = gen.staged_forward(zg, za, **kw)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"with torch.manual_seed(0) before every call: {ms:.2f} ms per image")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        with torch.no_grad(): img, _ = gen.staged_forward(zg, za, **kw)
        m = callers.mask2color(img[:, :-3], dev)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"with mask2color(device) after every call: {ms:.2f} ms per image")
