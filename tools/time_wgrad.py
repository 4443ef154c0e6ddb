"""Times the weight-gradient stage alone on one backward chunk (A/B experiments on fenerf_siren_wgrad.hip): forward-save and the chain run
once, then fenerf_siren_param_grads is timed per launch group (fenerf_phase_timing).  python tools/time_wgrad.py [points] [amp]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd import native, procedural as proc     # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
amp = len(sys.argv) > 2 and sys.argv[2] == "amp"
B, H = 1, 256
spec = proc.model_spec("texture", hidden_dim=H, grid_size=96, z_dim=8)
sd = proc.make_state_dict(spec, seed=4, sigma_gain=150.0, with_mapping=False)
nat = native.NativeModel(sd, spec, "cuda:0", "f16x3", differentiable=True, wgrad_bf16_min_points=1 if amp else 0)
g = torch.Generator(device="cuda:0").manual_seed(0)
pts = (torch.rand((B, P, 3), device="cuda:0", generator=g) - 0.5) * 0.24
dirs = torch.randn((B, P, 3), device="cuda:0", generator=g)
film = {k: torch.tensor(v, device="cuda:0") for k, v in proc.film_params(spec, B, seed=4).items()}
args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
d_out = torch.randn_like(out)
d_t, d_e = nat.siren_backward(B, P, *args, out, d_out, tape)
iters = 8
for _ in range(2):
    nat.siren_param_grads(pts, dirs, *args, out, d_out, tape, tape_e, d_t)
with native.phase_timing() as t:
    for _ in range(iters):
        r = nat.siren_param_grads(pts, dirs, *args, out, d_out, tape, tape_e, d_t)
chk = float(sum(w.double().abs().sum() for w in r["geo_w"] + r["color_w"]))
gb = P * 10 * H * (4.0 if amp else 8.0) / 1e9
print(" ".join(f"{k}={v / iters:.4f}" for k, v in sorted(t.ms.items())) + f" ms per chunk of {P} points"
      f" | square jobs {gb / (t.ms['wgrad_square'] / iters * 1e-3) / 1e3:.2f} TB/s | sum|dW| {chk:.6e}")
