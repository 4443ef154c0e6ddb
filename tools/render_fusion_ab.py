"""A/B of the two launch routes of fenerf_render_forward on an f16x3 model (include/fenerf.h fenerf_set_render_fusion): one launch per
hierarchical render against four.  Alternating blocks on the same box, wall clock over K back-to-back renders + the device time of the
launch groups of one render.  Measurement tool (no oracle).

    python tools/render_fusion_ab.py [--B 1] [--size 128] [--steps 24] [--iters 40] [--rounds 4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd import _lib, native, procedural as proc          # noqa: E402
from fenerf_amd.generators import volumetric_rendering as VR     # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--modes", default="off,auto")
    a = ap.parse_args()
    B, S, N = a.B, a.size, a.steps
    R = S * S
    spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
    sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, "f16x3")
    film = proc.film_params(spec, B, seed=1000)
    tf = tuple(torch.as_tensor(film[k], device=DEV) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    torch.manual_seed(1234)
    o, d, z, _, _ = VR.sample_rays(B, N, DEV, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * R, N), device=DEV)
    opts = _lib.composite_opts("relu", 0.0, fill_mode="seg_padding_background", fill_color="black")
    modes = a.modes.split(",")
    res = {m: [] for m in modes}
    groups = {}
    for m in modes:
        with native.render_fusion(m):
            for _ in range(5):
                nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
            with native.phase_timing() as t:
                nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
            groups[m] = {"calls": t.calls, "ms": {k: round(v, 4) for k, v in t.ms.items()}}
    for _ in range(a.rounds):
        for m in modes:
            with native.render_fusion(m):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
                torch.cuda.synchronize()
                res[m].append((time.perf_counter() - t0) / a.iters * 1e3)
    out = {"workload": f"{B} x {S}x{S} rays x {N}+{N} samples, H256 + 96^3 grid, f16x3", "iters": a.iters,
           "ms_per_render": {m: [round(x, 4) for x in v] for m, v in res.items()},
           "ms_per_render_best": {m: round(min(v), 4) for m, v in res.items()},
           "rays_per_s_best": {m: round(B * R / (min(v) * 1e-3)) for m, v in res.items()},
           "launch_groups_of_one_render": groups}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
