"""Dense generator step before and after sparse steps ran in the same process (allocator state).  python tools/exp/gstep_after_sparse.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fenerf_amd import procedural as proc

dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
def leg(B, S, N, sparse, tag):
    r = bench.gstep_leg(spec, sd, dev, B, S, N, "f16x3", iters=10, breakdown=not sparse, sparse=sparse)
    st = torch.cuda.memory_stats()
    print(f"{tag}: {B} x {S}x{S} x {N}+{N} {'sparse' if sparse else 'dense'}: {r['ms']:.2f} ms per step; reserved {torch.cuda.memory_reserved() / 2**30:.1f} GB, "
          f"allocator retries {st['num_alloc_retries']}, segments {st['segment.all.current']}, device mallocs so far {st['num_device_alloc']}", flush=True)
leg(1, 128, 24, False, "fresh")
leg(1, 128, 24, True, "sparse")
leg(1, 128, 24, False, "dense after sparse")
leg(6, 64, 12, True, "sparse, other shape")
leg(1, 128, 24, False, "dense after sparse at another shape")
