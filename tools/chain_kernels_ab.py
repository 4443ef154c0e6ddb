"""A/B of the two bf16x3 chain kernels on identical inputs: runs itself once per kernel (the choice is per process,
FENERF_BACKWARD_KERNEL), compares the d(theta) dumps layer by layer, d(grid features) and the parameter gradients.
--ab forward: the two f16x3 forward-save kernels instead (FENERF_FORWARD_KERNEL=f16s vs the 16-point default): outputs, tape,
sampled grid features.

    python tools/chain_kernels_ab.py [--H 32] [--grid 5] [--B 2] [--P 96] [--ab backward|forward]
"""
import argparse
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(a, out_path):
    from fenerf_amd import native, procedural as proc
    spec = proc.model_spec("texture", hidden_dim=a.H, grid_size=a.grid, z_dim=8) if a.grid else proc.model_spec("baseline", hidden_dim=a.H, z_dim=8)
    sd = proc.make_state_dict(spec, seed=3, sigma_gain=20.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, "cuda:0", precision="f16x3", differentiable=True)
    rng = np.random.default_rng(7)
    B, P = a.B, a.P
    pts = torch.tensor(rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32), device="cuda")
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    dirs = torch.tensor(dirs, device="cuda")
    film = {k: torch.tensor(v, device="cuda") for k, v in proc.film_params(spec, B, seed=4).items()}
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
    g_out = rng.normal(size=(B, P, spec["output_dim"])).astype(np.float32)
    g_out[..., -1] *= 0.02
    g_out = torch.tensor(g_out, device="cuda")
    d_t, d_e = nat.siren_backward(B, P, *args, out, g_out, tape)
    grads = nat.siren_param_grads(pts, dirs, *args, out, g_out, tape, tape_e, d_t)
    torch.cuda.synchronize()
    H, L = spec["hidden_dim"], spec["n_geo"] + spec["n_color"]
    res = {"d_t": d_t[: L * H * B * P].cpu().numpy().reshape(B * P // 32, L, H // 8, 64, 4), "tape": tape[: L * H * B * P].cpu().numpy().reshape(B * P // 32, L, H // 8, 64, 4),
           "out": out.cpu().numpy()}
    if tape_e is not None:
        res["tape_e"] = tape_e.cpu().numpy()
    if d_e is not None:
        res["d_e"] = d_e.cpu().numpy()
    for k, v in grads.items():
        if isinstance(v, list):
            for i, t in enumerate(v):
                res[f"{k}.{i}"] = t.cpu().numpy()
        else:
            res[k] = v.cpu().numpy()
    np.savez(out_path, **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--grid", type=int, default=5)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--P", type=int, default=96)
    ap.add_argument("--tol", type=float, default=1e-4, help="both kernels round dz to bf16 pairs (the remainder differently): ~2e-5")
    ap.add_argument("--ab", default="backward", choices=["backward", "forward"])
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        run(a, a.child)
        return
    os.makedirs("gpurun_out", exist_ok=True)
    paths = {}
    var = "FENERF_BACKWARD_KERNEL" if a.ab == "backward" else "FENERF_FORWARD_KERNEL"
    names = ("b16", "b16w") if a.ab == "backward" else ("f16s", "f16w")
    for k in names:
        paths[k] = f"/tmp/dbg_{k}.npz"
        env = dict(os.environ)
        env[var] = k
        subprocess.run([sys.executable, os.path.abspath(__file__), "--H", str(a.H), "--grid", str(a.grid), "--B", str(a.B), "--P", str(a.P),
                        "--ab", a.ab, "--child", paths[k]], env=env, check=True, timeout=300)
    ref, got = np.load(paths[names[0]]), np.load(paths[names[1]])
    print("tape identical:", np.array_equal(ref["tape"], got["tape"]))
    if a.ab == "forward":
        worst = 0.0
        for k in ("out", "tape", "tape_e"):
            if k not in ref.files:
                continue
            sc = max(np.abs(ref[k]).max(), 1e-30)
            e = float(np.abs(ref[k] - got[k]).max() / sc)
            worst = max(worst, e)
            print(f"{k}: rel diff {e:.3e} (scale {sc:.3e})")
        print(f"worst relative difference between the two forward-save kernels: {worst:.3e}")
        sys.exit(0 if worst <= a.tol else 1)
    dr, dg = ref["d_t"], got["d_t"]
    T, L = dr.shape[0], dr.shape[1]
    for l in range(L - 1, -1, -1):
        e = np.abs(dr[:, l] - dg[:, l])
        sc = np.abs(dr[:, l]).max()
        print(f"layer {l}: max|d_t diff| {e.max():.3e} (scale {sc:.3e})")
        if e.max() > 1e-3 * sc:
            bad = np.argwhere(e > 1e-3 * sc)
            print("   bad entries:", len(bad), "of", e.size, " first:", bad[:6].tolist())
            print("   by tile:", np.bincount(bad[:, 0], minlength=T).tolist())
            print("   by group g=4nb+j:", np.bincount(bad[:, 1], minlength=dr.shape[2]).tolist())
            print("   by lane:", np.bincount(bad[:, 2], minlength=64).tolist())
            # is it a permutation?  look for the got-value in ref's tile
            t, g_, ln, i = bad[0]
            v = dg[t, l, g_, ln, i]
            where = np.argwhere(np.isclose(dr[t, l], v, rtol=1e-3, atol=0))
            print(f"   got[{t},{l},{g_},{ln},{i}] = {v:.5e}; ref has it at", where[:4].tolist(), " ref value there", dr[t, l, g_, ln, i])
    worst = 0.0
    for l in range(L):
        worst = max(worst, float(np.abs(dr[:, l] - dg[:, l]).max() / max(np.abs(dr[:, l]).max(), 1e-30)))
    for k in ref.files:
        if k in ("d_t", "tape"):
            continue
        sc = max(np.abs(ref[k]).max(), 1e-30)
        e = float(np.abs(ref[k] - got[k]).max() / sc)
        worst = max(worst, e)
        print(f"{k}: rel diff {e:.3e}")
    print(f"worst relative difference between the two chain kernels: {worst:.3e}")
    sys.exit(0 if worst <= a.tol else 1)


if __name__ == "__main__":
    main()
