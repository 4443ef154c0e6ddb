#!/usr/bin/env python
"""Fast GPU sanity check of one SIREN kernel variant vs the oracle (used before running the full suite on a new kernel)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fenerf_amd import native, procedural as proc
from oracle import fenerf_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
for kind, H, grid, npts, B in (("texture", 32, 5, 100, 2), ("baseline", 64, 0, 333, 1), ("texture", 128, 6, 1000, 2), ("texture", 256, 8, 5000, 1),
                               ("texture", 256, 8, 40000, 2)):
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=5, sigma_gain=200.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, "cuda:0", prec)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.12, 0.12, (B, npts, 3)).astype(np.float32)
    dirs = rng.normal(size=(B, npts, 3)).astype(np.float32)
    film = proc.film_params(spec, B, seed=5)
    tf = tuple(torch.as_tensor(film[k], device="cuda:0") for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    t0 = time.time()
    out = nat.siren_forward(torch.as_tensor(pts, device="cuda:0"), torch.as_tensor(dirs, device="cuda:0"), *tf)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out = out.cpu().numpy()
    n = min(npts, 600)
    ref = O.siren_forward(sd, spec, pts[:, :n], dirs[:, :n], film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"], dtype=np.float64)
    d = np.abs(out[:, :n] - ref)
    tail = np.abs(out[:, -n:] - O.siren_forward(sd, spec, pts[:, -n:], dirs[:, -n:], film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"], dtype=np.float64))
    print(f"{prec} {kind} H={H} B={B} P={npts}: {dt*1e3:.1f} ms  max|err| rgb {d[..., -4:-1].max():.2e} labels {d[..., :-4].max():.2e} "
          f"sigma rel {d[..., -1].max() / np.abs(ref[..., -1]).max():.2e} | tail rgb {tail[..., -4:-1].max():.2e}  finite={np.isfinite(out).all()}", flush=True)
