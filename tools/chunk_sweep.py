"""Generator step (bench.py's gstep leg, 1 x 128^2 x 24+24) against the size of the backward chunks (fenerf_amd/siren/autograd.py
BACKWARD_CHUNK_POINTS): time and peak memory.  python tools/chunk_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from fenerf_amd import procedural as proc                # noqa: E402
from fenerf_amd.siren import autograd as SA              # noqa: E402

spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
for chunk in (131072, 196608, 393216, 98304, 131072, 196608):
    SA.BACKWARD_CHUNK_POINTS = chunk
    r = bench.gstep_leg(spec, sd, torch.device("cuda:0"), 1, 128, 24, "f16x3", iters=16)
    print("chunk %7d: %.3f ms, peak %.2f GB" % (chunk, r["ms"], r["peak_GB"]), flush=True)
