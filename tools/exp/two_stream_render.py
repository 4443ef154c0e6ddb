"""Consecutive independent renders (the headline's step) on ONE stream vs alternating between TWO: does the small composite / resample
launch of one image hide under the next image's SIREN kernel?  usage: python tools/exp/two_stream_render.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fenerf_amd import _lib, native, procedural as proc                   # noqa: E402
from fenerf_amd.generators import volumetric_rendering as VR            # noqa: E402

dev = torch.device("cuda", 0)
S, N, B = 128, 24, 1
opts = _lib.composite_opts("relu", 0.0, False, False, False, "seg_padding_background", "white")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
film = proc.film_params(spec, B, seed=1)
tf = tuple(torch.as_tensor(film[k], device=dev) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
torch.manual_seed(3)
o, d, z, _, _ = VR.sample_rays(B, N, dev, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
u = torch.rand((B * S * S, N), device=dev)
nat = native.NativeModel(sd, spec, dev, "f16x3")
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ref = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)[0].clone()


def run(n_streams, steps=40):
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if n_streams == 1:
            outs.append(nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)[0])
        else:
            with torch.cuda.stream(streams[i % 2]):
                outs.append(nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)[0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert all(torch.equal(x, ref) for x in outs[-4:])
    return ms


for s_ in streams:
    s_.wait_stream(torch.cuda.current_stream(dev))
for rep in range(3):
    a, b = run(1), run(2)
    print(f"rep {rep}: one stream {a:.3f} ms/step = {S * S / a / 1e3:.2f} M rays/s; two streams {b:.3f} ms/step = {S * S / b / 1e3:.2f} M rays/s ({a / b:.3f} x)", flush=True)
