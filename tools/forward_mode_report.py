"""Round 5: what the opt-in reduced-precision forwards (include/fenerf.h fenerf_model_set_forward_mode: "f16x2" = two fp16 MFMAs per
product everywhere, "f16x3c2" = three through the geometry trunk and two in the colour layers / heads) cost in accuracy, next to the
default f16x3 -- against every reference fixture with SIREN outputs, the end-to-end fixtures, all 16,384 rays of the bench image against
the numpy oracle, and the fp64 arbiter's flip count.  Measurement tool (imports the test helpers and the oracle); prints a markdown table.

    python tools/forward_mode_report.py > gpurun_out/forward_modes.md
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as TP                                      # noqa: E402
from conftest import load_golden, spec_from_golden                # noqa: E402
from fenerf_amd import _lib, native, procedural as proc           # noqa: E402
from fenerf_amd.generators import volumetric_rendering as VR      # noqa: E402

DEV = "cuda:0"
MODES = ["f16x3", "f16x3c2", "f16x2"]
T, N_ = TP.T, TP.N_


def siren_fixture(name):
    g = load_golden(name)
    spec, sd = TP._weights_for(name)
    film, tf = TP._film(g, spec)
    B, R, N = g["st_z_coarse"].shape[:3]
    pts = g["st_points"].reshape(B, R * N, 3)
    dirs = np.broadcast_to(g["st_dirs"][:, :, None, :], (B, R, N, 3)).reshape(B, R * N, 3)
    ref = g["st_siren_coarse"]
    row = {}
    for mode in MODES:
        nat = native.NativeModel(sd, spec, DEV, mode)
        out = N_(nat.siren_forward(T(pts), T(dirs), *tf))
        d = np.abs(out - ref)
        row[mode] = (d[..., -4:-1].max(), d[..., :-4].max() if out.shape[-1] > 4 else 0.0, d[..., -1].max() / max(np.abs(ref[..., -1]).max(), 0.1))
        nat.close()
    return row


def e2e_fixture(name):
    g = load_golden(name)
    spec = spec_from_golden(g)
    row = {}
    for mode in MODES:
        gen = TP._make_generator(g, dict(spec, z_dim=spec.get("z_dim", 16) if spec["hidden_dim"] == 32 else 256), mode)
        film, tf = TP._film(g, spec)
        hier = bool(g["meta_hier"])
        seq = [g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"]] + ([g["rand_u_fine"], g["rand_noise_fine"]] if hier else [])
        gen.draws = VR.RecordedDraws(seq)
        with torch.no_grad():
            px, _ = gen.forward_with_frequencies(tf[0], tf[2], tf[1], tf[3], img_size=int(g["meta_S"]), fov=12, ray_start=0.88, ray_end=1.12,
                                                 num_steps=int(g["meta_N"]), h_stddev=0.3, v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5,
                                                 hierarchical_sample=hier, sample_dist="gaussian", **TP.kwargs_from_golden(g))
        err = np.abs(N_(px) - g["pixels"]).max(axis=1)
        am = (N_(px)[:, :-3].argmax(1) != g["pixels"][:, :-3].argmax(1))
        row[mode] = (err.max(), int((err > 1e-3).sum()), err.size, int(am.sum()))
    return row


def all_rays():
    import __graft_entry__ as ge
    spec, sd = TP._full_weights()
    B, S_, N = 1, 128, 24
    R = S_ * S_
    film = proc.film_params(spec, B, seed=0)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    torch.manual_seed(0)
    o, d, z, _, _ = VR.sample_rays(B, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * R, N), device=DEV)
    opts = _lib.composite_opts("relu", fill_mode="seg_padding_background", fill_color="white")
    px32, dp32, z32 = TP._oracle_render_rays(sd, spec, args, N_(o), N_(d), N_(z), N_(u), "white")
    px64, dp64, z64 = TP._oracle_render_rays(sd, spec, args, N_(o), N_(d), N_(z), N_(u), "white", dtype=np.float64)
    flip32 = int((np.abs(z32 - z64).max(-1) > 1e-5).sum())
    rows = {}
    for mode in MODES:
        nat = native.NativeModel(sd, spec, DEV, mode)
        rgb, depth, _, _ = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
        rgb = N_(rgb)
        zs = ge.nat_sorted_z(nat, o, d, z, u, tf, opts)
        e = np.abs(rgb - px32).max(-1)
        thr = (rgb[..., 0] == 1) != (px32[..., 0] == 1)
        e = np.where(thr, 0.0, e)
        flips_o = int((np.abs(zs - z32).max(-1) > 1e-5).sum())
        flips64 = int((np.abs(zs - z64).max(-1) > 1e-5).sum())
        lab, r_lab = rgb[..., 1:-3], px32[..., 1:-3]
        decided = (np.sort(r_lab, -1)[..., -1] - np.sort(r_lab, -1)[..., -2] > 1e-6) & (px32[..., 0] != 1) & ~thr
        rows[mode] = (e.max(), int((e > 1e-3).sum()), int(thr.sum()), flips_o, flips64, int(((lab.argmax(-1) != r_lab.argmax(-1)) & decided).sum()))
        nat.close()
    return rows, flip32, R


def main():
    print("# Reduced-precision forward modes vs the default (round 5)\n")
    print("| fixture (SIREN outputs, coarse points, vs the reference's own) | " + " | ".join(f"{m}: rgb / labels / sigma rel" for m in MODES) + " |")
    print("|---|" + "---|" * len(MODES))
    for name in ("tiny_texture_fwd", "tiny_baseline_fwd", "h256_texture_16x16_n12", "h256_texture_16x16_n24_trained", "h256_baseline_8x8_n12",
                 "tiny_texture_fwd_trained", "tiny_texture_fwd_bigfilm", "h256_texture_8x8_n12_bigfilm"):
        r = siren_fixture(name)
        print(f"| {name} | " + " | ".join(f"{r[m][0]:.1e} / {r[m][1]:.1e} / {r[m][2]:.1e}" for m in MODES) + " |")
    print("\n| fixture (forward_with_frequencies, pixels vs the reference's) | " + " | ".join(f"{m}: max err, pixels > 1e-3, argmax mismatches" for m in MODES) + " |")
    print("|---|" + "---|" * len(MODES))
    for name in ("tiny_texture_fwd", "tiny_texture_fwd_nohier", "tiny_baseline_fwd", "h256_texture_16x16_n12", "h256_texture_16x16_n24_trained",
                 "h256_baseline_8x8_n12", "tiny_texture_fwd_trained"):
        r = e2e_fixture(name)
        print(f"| {name} | " + " | ".join(f"{r[m][0]:.1e}, {r[m][1]} of {r[m][2]}, {r[m][3]}" for m in MODES) + " |")
    rows, flip32, R = all_rays()
    print(f"\n128 x 128 x 24+24, H = 256 + 96^3 grid, all {R} rays against the fp32 numpy oracle (the reference's arithmetic; it flips {flip32} rays against its own fp64 evaluation):\n")
    print("| mode | max pixel err | rays > 1e-3 | rays on the fill threshold | rays resampled differently from the oracle | ... from fp64 | label argmax mismatches on decided rays |")
    print("|---|---|---|---|---|---|---|")
    for m in MODES:
        r = rows[m]
        print(f"| {m} | {r[0]:.2e} | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} |")


if __name__ == "__main__":
    main()
