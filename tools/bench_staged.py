"""BASELINE.json configs[4]: generator.staged_forward at 256x256, 48+48 samples on one GPU (the inference / inversion path the
reference chunks by max_batch_size; 288 GB of HBM hold the whole image, so it is one fused render).
    python tools/bench_staged.py [--size 256] [--steps 48] [--iters 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fenerf_amd import curriculums
from fenerf_amd.generators import generators
from fenerf_amd.siren import siren

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=48)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
cur = curriculums.CelebA_double_semantic_texture_embedding_256_dim_96
gen = generators.DoubleImplicitGenerator3d(getattr(siren, cur["model"]), 256, 256, 22).to(dev)
gen.set_device(dev)
md = {**curriculums.extract_metadata(cur, 60000), "nerf_noise": 0, "psi": 0.7, "img_size": a.size, "num_steps": a.steps,
      "max_batch_size": 10 ** 9}
zg, za = torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)
def run():
    with torch.no_grad():
        return gen.staged_forward(zg, za, **md)
for _ in range(2):
    img, depth = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.iters * 1e3
print(json.dumps({"config": f"staged_forward {a.size}x{a.size}, {a.steps}+{a.steps} samples, batch 1", "ms_per_image": ms,
                  "rays_per_s": a.size * a.size / (ms * 1e-3), "points_per_image": a.size * a.size * 2 * a.steps,
                  "image_shape": list(img.shape), "peak_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))
