"""The generator step at the curriculum's early stages (curriculums.py:84-86: 4 images of 32 x 32 x 12+12, 6 of 64 x 64 x 12+12 per
micro-batch) against the sum of its SIREN kernels: what the host side and the small launches cost where the kernels are short.
    python tools/exp/gstep_small_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fenerf_amd import procedural as proc

dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
for B, S, N in ((4, 32, 12), (6, 64, 12), (1, 128, 24)):
    for sparse in (False, True):
        r = bench.gstep_leg(spec, sd, dev, B, S, N, "f16x3", iters=10, breakdown=not sparse, sparse=sparse)
        k = sum(q["ms"] for q in r["roofline"]["per_kernel"]) if "roofline" in r else float("nan")
        print(f"{B} x {S}x{S} x {N}+{N} ({B * S * S * 2 * N} points) {'sparse' if sparse else 'dense '}: {r['ms']:.2f} ms per step"
              + (f", SIREN kernels {k:.2f} ms" if not sparse else f", {100 * r['kept_frac']:.1f} % kept"), flush=True)
