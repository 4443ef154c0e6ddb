"""Two measurements that bound what the generator step's HBM streams could gain from the 256-MiB Infinity Cache
("what comes next"):

1. producer -> consumer hand-over through memory as a function of the buffer size: kernel A writes a buffer, kernel B reads it
   (torch elementwise ops, ping-pong between two buffers).  If the effective rate of small hand-overs is well above the
   large-buffer (HBM) rate, a layer-ordered backward whose dz buffers fit the cache would not pay HBM for them.
2. the generator step of bench.py (1 x 128^2 x 24+24) with the backward cut into chunks of different sizes: with the present
   tile-ordered chain kernel the d(theta) of a chunk is 9.2 KB per point, so only very small chunks would fit.

    python tools/mall_probe.py [--no-gstep]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def handover(nbytes, iters=40):
    n = nbytes // 4
    a = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        torch.mul(a, 1.0001, out=b); torch.mul(b, 0.9999, out=a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        torch.mul(a, 1.0001, out=b)
        torch.mul(b, 0.9999, out=a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (2 * iters)
    return {"MiB": nbytes / 2**20, "us_per_kernel": dt * 1e6, "read_plus_write_TBps": 2 * nbytes / dt / 1e12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-gstep", action="store_true")
    a = ap.parse_args()
    out = {"handover": [handover(m << 20) for m in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048)]}
    for r in out["handover"]:
        print("handover %6.0f MiB  %8.1f us/kernel  %6.2f TB/s (read + write)" % (r["MiB"], r["us_per_kernel"], r["read_plus_write_TBps"]), flush=True)
    if not a.no_gstep:
        import bench
        from fenerf_amd import procedural as proc
        from fenerf_amd.siren import autograd as SA
        spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
        sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
        out["gstep"] = []
        for chunk in (8192, 16384, 32768, 65536, 131072, 393216):
            SA.BACKWARD_CHUNK_POINTS = chunk
            r = bench.gstep_leg(spec, sd, torch.device("cuda:0"), 1, 128, 24, "f16x3", iters=5)
            out["gstep"].append({"chunk_points": chunk, "ms": r["ms"], "peak_GB": r["peak_GB"]})
            print("gstep chunk %7d points: %.2f ms, peak %.2f GB" % (chunk, r["ms"], r["peak_GB"]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
