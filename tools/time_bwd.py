"""Times the backward-chain kernel alone (A/B experiments on fenerf_siren_bwd.hip): python tools/time_bwd.py [points_per_image]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fenerf_amd import native, procedural as proc

P = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
B, H = 2, 256
spec = proc.model_spec("texture", hidden_dim=H, grid_size=96, z_dim=8)
sd = proc.make_state_dict(spec, seed=4, sigma_gain=150.0, with_mapping=False)
nat = native.NativeModel(sd, spec, "cuda:0", "f16x3", differentiable=True)
g = torch.Generator(device="cuda:0").manual_seed(0)
pts = (torch.rand((B, P, 3), device="cuda:0", generator=g) - 0.5) * 0.24
dirs = torch.randn((B, P, 3), device="cuda:0", generator=g)
film = {k: torch.tensor(v, device="cuda:0") for k, v in proc.film_params(spec, B, seed=4).items()}
args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
d_out = torch.randn_like(out)
for name, fn in (("forward_save", lambda: nat.siren_forward_save(pts, dirs, *args)),
                 ("backward chain", lambda: nat.siren_backward(B, P, *args, out, d_out, tape))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name}: {ms:.3f} ms for {B * P} points = {B * P * 1603584 / ms / 1e9:.1f} TFLOP/s algorithmic")
