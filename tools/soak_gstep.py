"""Soak: N generator steps (1 x 128^2 x 24+24, the shape of bench.py's gstep leg) back to back -- a hang or a rare race in the stream
loops of the 16-point kernels would show up as a timeout or a non-finite gradient.  python tools/soak_gstep.py [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from fenerf_amd import procedural as proc                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
t0 = time.time()
r = bench.gstep_leg(spec, sd, torch.device("cuda:0"), 1, 128, 24, "f16x3", iters=n)
print(f"{n} generator steps: {r['ms']:.3f} ms/step, peak {r['peak_GB']:.2f} GB, wall {time.time() - t0:.1f} s")
