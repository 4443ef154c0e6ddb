"""Does the dense generator step slow down after the shape changed (a curriculum stage switch)?  1 x 128 x 128 x 24+24, then 6 x 64 x 64 x
12+12, then the first shape again -- with and without torch.cuda.empty_cache() in between.   python tools/exp/gstep_shape_switch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fenerf_amd import procedural as proc

dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
def leg(B, S, N, tag):
    r = bench.gstep_leg(spec, sd, dev, B, S, N, "f16x3", iters=10, breakdown=True)
    k = sum(q["ms"] for q in r["roofline"]["per_kernel"])
    print(f"{tag}: {B} x {S}x{S} x {N}+{N}: {r['ms']:.2f} ms per step (SIREN kernels {k:.2f}); reserved {torch.cuda.memory_reserved() / 2**30:.1f} GB, "
          f"allocator retries {torch.cuda.memory_stats()['num_alloc_retries']}, segments {torch.cuda.memory_stats()['segment.all.current']}", flush=True)
leg(1, 128, 24, "fresh process")
leg(6, 64, 12, "other shape")
leg(1, 128, 24, "first shape again")
torch.cuda.empty_cache()
leg(1, 128, 24, "after empty_cache()")
