"""Times the backward chain kernel alone (fenerf_siren_backward) on one chunk of the generator step's backward:
131,072 points of one image, H=256 + 96^3 grid.  FENERF_LIB selects the build.

    python tools/bench_chain.py [--points 131072] [--iters 20]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd import native, procedural as proc   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=131072)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--determinism", type=int, default=0, help="repeat forward-save + chain this many times and compare bit for bit")
    a = ap.parse_args()
    spec = proc.model_spec("texture", hidden_dim=a.H, grid_size=96)
    sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, "cuda:0", precision="f16x3", differentiable=True)
    rng = np.random.default_rng(1)
    P = a.points
    pts = torch.tensor(rng.uniform(-0.12, 0.12, (1, P, 3)).astype(np.float32), device="cuda")
    dirs = torch.tensor(np.tile(np.array([0, 0, -1], np.float32), (1, P, 1)), device="cuda")
    film = {k: torch.tensor(v, device="cuda") for k, v in proc.film_params(spec, 1, seed=5).items()}
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
    g_out = torch.tensor(rng.normal(size=(1, P, spec["output_dim"])).astype(np.float32), device="cuda")
    for _ in range(3):
        d_t, d_e = nat.siren_backward(1, P, *args, out, g_out, tape)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(a.iters):
        e0.record()
        d_t, d_e = nat.siren_backward(1, P, *args, out, g_out, tape)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    if a.determinism:
        # the stream loop's synchronisation (counted vmcnt waits, barriers, LDS rings) has no data-dependent path: any race shows up
        # as run-to-run differences.  Forward-save and chain are repeated and compared bit for bit with the first run.
        LHP = (spec["n_geo"] + spec["n_color"]) * a.H * P
        film_grads = lambda dt: nat.siren_param_grads(pts, dirs, *args, out, g_out, tape, tape_e, dt, film_only=True)   # consumes the FiLM sums
        ref_t, ref_o = tape[:LHP].clone(), out.clone()
        ref_d, ref_e = d_t[:LHP].clone(), d_e.clone()
        ref_f = {k: v.clone() for k, v in film_grads(d_t).items()}
        bad = 0
        for i in range(a.determinism):
            o2, t2, _ = nat.siren_forward_save(pts, dirs, *args)
            d2, e2 = nat.siren_backward(1, P, *args, out, g_out, tape)
            f2 = film_grads(d2)
            same = torch.equal(o2, ref_o) and torch.equal(t2[:LHP], ref_t) and torch.equal(d2[:LHP], ref_d) and torch.equal(e2, ref_e) and \
                all(torch.equal(f2[k], ref_f[k]) for k in ref_f)
            bad += 0 if same else 1
        print(json.dumps({"determinism_runs": a.determinism, "runs_that_differ": bad}))
        sys.exit(1 if bad else 0)
    L, H = spec["n_geo"] + spec["n_color"], a.H
    n = L * H * P
    print(json.dumps({"points": P, "ms_median": float(np.median(ts)), "ms_min": float(np.min(ts)),
                      "d_t_abs_sum": float(d_t[:n].double().abs().sum()), "film_abs_sum": float(d_t[n:n + (P // 16) * L * 2 * H].double().abs().sum()),
                      "d_e_abs_sum": float(d_e.double().abs().sum()), "GBps_tape_plus_dtheta": 2 * n * 4 / (np.median(ts) * 1e-3) / 1e9}))


if __name__ == "__main__":
    main()
