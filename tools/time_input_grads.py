"""Times fenerf_siren_input_grads alone (the extra pass over the d(theta) dump that yields d points / d view directions) beside the
chain kernel whose dump it reads: python tools/time_input_grads.py [points] [H] [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fenerf_amd import native, procedural as proc

P = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
B, DEV = 1, "cuda:0"
spec = proc.model_spec("texture", hidden_dim=H, grid_size=96, z_dim=8)
sd = proc.make_state_dict(spec, seed=4, sigma_gain=150.0, with_mapping=False)
nat = native.NativeModel(sd, spec, DEV, prec, differentiable=True)
g = torch.Generator(device=DEV).manual_seed(0)
pts = (torch.rand((B, P, 3), device=DEV, generator=g) - 0.5) * 0.24
dirs = torch.nn.functional.normalize(torch.randn((B, P, 3), device=DEV, generator=g), dim=-1)
film = {k: torch.tensor(v, device=DEV) for k, v in proc.film_params(spec, B, seed=4).items()}
args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
d_out = torch.randn_like(out)
d_grid = torch.zeros(tuple(nat.grid_shape) + (32,), device=DEV)
w0 = torch.tensor(sd["network.0.layer.weight"], device=DEV)
wc0 = torch.tensor(sd["color_layer_sine.0.layer.weight"], device=DEV)
dp, dd = torch.empty_like(pts), torch.empty_like(dirs)


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


d_t = nat.siren_backward_grid(B, P, *args, out, d_out, tape, pts, d_grid)
ms_chain = timed(lambda: nat.siren_backward_grid(B, P, *args, out, d_out, tape, pts, d_grid))
ms = timed(lambda: nat.siren_input_grads(pts, *args, d_t, w0, wc0, dp, dd))
ms_p = timed(lambda: nat.siren_input_grads(pts, *args, d_t, w0, wc0, dp, None))
ms_d = timed(lambda: nat.siren_input_grads(pts, *args, d_t, w0, wc0, None, dd))
n = B * P
alg = n * (2 * H * 4 + 12 + 24)            # two layers of the dump in, points in, two gradient rows out (the 8 grid corner lines per point come from L2)
print(f"{prec} H={H}, {n} points: chain {ms_chain:.3f} ms | input gradients {ms:.3f} ms (+{100 * ms / ms_chain:.1f} % of the chain) = "
      f"{alg / ms / 1e6:.0f} GB/s of {alg / 1e6:.0f} MB algorithmic ({alg / n:.0f} B/point) = {alg / ms / 1e6 / 8000:.2f} of the 8 TB/s HBM roof; "
      f"positions only {ms_p:.3f} ms, view directions only {ms_d:.3f} ms; + {8 * 128} B/point of grid corner lines from L2")
