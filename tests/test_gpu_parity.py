"""GPU parity tests (-m gpu): every call goes through the C-ABI (libfenerf_hip.so) on cuda:0 and is checked
against (a) golden vectors captured from the reference and (b) the numpy oracle on the same seeded inputs.

Tolerances (north_star: <= 1e-3 max-abs RGB vs the reference CPU path, exact argmax semantics):
  SIREN outputs (teacher-forced points)   rgb <= 1e-4, labels <= 1e-4 abs + 1e-4 rel, sigma <= 2e-4 * sigma_gain scale
  composite / resample / merge            <= 1e-5 (pure fp32 re-association)
  end to end                              <= 1e-3 on every pixel whose |weights_sum - 0.9| is not within rounding of the
                                          fill threshold; the count of excluded pixels is asserted small and reported.
"""
import ast
import functools
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import film_from_golden, kwargs_from_golden, load_golden, spec_from_golden, state_from_golden, weights_from_golden
from fenerf_amd import _lib, native, procedural as proc
from fenerf_amd.generators import generators as G
from fenerf_amd.generators import volumetric_rendering as VR
from fenerf_amd.siren import siren as S
from oracle import fenerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)


def N_(t):
    return t.detach().cpu().numpy()


PRECISIONS = ["f32", "f16x3"]   # exact fp32 MFMA / error-compensated fp16 MFMA (3 MFMAs per product)
# Round 6: the opt-in forward tier "f16x3c2" (fenerf_model_set_forward_mode: three fp16 MFMAs per product through the geometry trunk and the
# label / sigma head, two in the colour layers and the rgb head; + 7.6 % rays/s) goes through every no-grad forward test the default goes
# through, with its own measured bounds -- a tested product mode, not a report.  (A differentiable evaluation runs the default arithmetic:
# the gradient tests stay on PRECISIONS.)
FORWARD_PRECISIONS = PRECISIONS + ["f16x3c2"]


@functools.lru_cache(maxsize=None)
def _weights_for(name):
    g = load_golden(name)
    spec = spec_from_golden(g)
    return spec, weights_from_golden(g, spec, with_mapping=False)


@functools.lru_cache(maxsize=None)
def _native_for(name, precision="f32"):
    spec, sd = _weights_for(name)
    return native.NativeModel(sd, spec, DEV, precision), spec, sd


def _film(g, spec):
    f = film_from_golden(g, spec)
    return f, tuple(T(f[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))


def _report(tag, got, ref):
    d = np.abs(got - ref)
    print(f"[parity] {tag}: max|err| rgb {d[..., -4:-1].max():.3e}  sigma {d[..., -1].max():.3e}  "
          f"labels {d[..., :-4].max() if got.shape[-1] > 4 else 0:.3e}  (|sigma| max {np.abs(ref[..., -1]).max():.3g})")


# ---------------------------------------------------------------------------------------------------
# loaded native code is the product path
# ---------------------------------------------------------------------------------------------------
def test_native_library_is_loaded():
    l = _lib.lib()
    assert l.fenerf_abi_version() == _lib.ABI_VERSION == 2
    maps = open("/proc/self/maps").read()
    assert "libfenerf_hip.so" in maps


# ---------------------------------------------------------------------------------------------------
# a8-a12: SIREN kernel vs reference outputs (teacher-forced points)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
@pytest.mark.parametrize("name", ["tiny_texture_fwd", "tiny_baseline_fwd", "h256_texture_16x16_n12",
                                  "h256_texture_16x16_n24_trained", "h256_baseline_8x8_n12", "tiny_texture_fwd_trained",
                                  "h96_texture_8x8_n12", "h192_baseline_8x8_n12"])       # H = 96 / 192 (round 5): widths that are not powers of two
def test_siren_forward_vs_reference(name, precision):
    g = load_golden(name)
    nat, spec, sd = _native_for(name, precision)
    name = f"{name}[{precision}]"
    film, tf = _film(g, spec)
    B, R, N = g["st_z_coarse"].shape[:3]
    pts = g["st_points"].reshape(B, R * N, 3)
    dirs = np.broadcast_to(g["st_dirs"][:, :, None, :], (B, R, N, 3)).reshape(B, R * N, 3)
    out = N_(nat.siren_forward(T(pts), T(dirs), *tf))
    ref = g["st_siren_coarse"]
    _report(name + " coarse vs reference", out, ref)
    # Bounds = what is measured x 1.5 (round 5; round 4 asserted 1e-4 on rgb / labels against a measured 5e-7 / 6e-8): rgb <= 5.4e-7,
    # labels <= 6.0e-8, sigma <= 7.2e-6 x the fixture's largest |sigma| -- over all fixtures, both precisions, coarse and fine points.
    # tiny_texture_fwd_trained (round 5: weights 2.4 x beyond their init range after the reference's own Adam run -- larger pre-activations,
    # larger labels): measured rgb 1.64e-6, labels 6.9e-7, sigma 3.1e-6 x |sigma|max.
    b_rgb, b_lab = (7e-6, 2e-6) if "trained" in name and name.startswith("tiny") else (8.5e-7, 9e-8)     # (fine points: 4.4e-6 / 1.2e-6)
    if precision == "f16x3c2":
        # two MFMAs per product in the colour branch: rgb measured (round 6, all fixtures, coarse / fine / fp64) <= 2.91e-5, tiny trained
        # 3.40e-5 coarse / 5.74e-5 fine; x 1.5.  Labels and sigma are the default's, bit for bit -- asserted against the default's own output
        b_rgb = 8.7e-5 if "trained" in name and name.startswith("tiny") else 4.4e-5
        ref3 = N_(_native_for(name.split("[")[0], "f16x3")[0].siren_forward(T(pts), T(dirs), *tf))
        assert np.array_equal(out[..., :-4], ref3[..., :-4]) and np.array_equal(out[..., -1], ref3[..., -1]), "f16x3c2: labels and sigma bit-identical to f16x3"
    def close(got, want, tag):
        smax = float(np.abs(want[..., -1]).max())
        e_rgb, e_lab, e_sig = (float(np.abs(got[..., sl] - want[..., sl]).max()) for sl in (slice(-4, -1), slice(None, -4), slice(-1, None)))
        assert e_rgb <= b_rgb and e_lab <= b_lab and e_sig <= 1.1e-5 * max(smax, 0.1), (tag, e_rgb, e_lab, e_sig, smax)
    close(out, ref, "coarse")
    # fine points (explicit) too, and the fp64 oracle as a tighter arbiter
    fo = N_(nat.siren_forward(T(g["st_fine_points"]), T(dirs), *tf))
    _report(name + " fine vs reference", fo, g["st_siren_fine"])
    close(fo, g["st_siren_fine"], "fine")
    o64 = O.siren_forward(sd, spec, pts[:, :512], dirs[:, :512], film["freq_geo"], film["phase_geo"], film["freq_app"],
                          film["phase_app"], dtype=np.float64)
    _report(name + " vs fp64 oracle", out[:, :512], o64)
    close(out[:, :512], o64, "fp64")      # measured: rgb <= 4.5e-7, labels <= 5.5e-8, sigma <= 5.1e-6 x |sigma|max


@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
def test_siren_rays_mode_lock_view_and_ragged_tiles(precision):
    """points generated in-kernel from (o, d, z); lock_view_dependence; P not a multiple of the 32-point tile;
    tiles straddling image boundaries; tile-independence (bit-exact sub-batch)."""
    g = load_golden("tiny_texture_fwd")
    nat, spec, sd = _native_for("tiny_texture_fwd", precision)
    film, tf = _film(g, spec)
    B, R, N = g["st_z_coarse"].shape[:3]
    o, d, z = T(g["st_origins"]), T(g["st_dirs"]), T(g["st_z_coarse"][..., 0])
    out = N_(nat.siren_forward_rays(o, d, z, *tf))
    _report("rays mode vs reference", out.reshape(B, R * N, -1), g["st_siren_coarse"])
    np.testing.assert_allclose(out.reshape(B, R * N, -1)[..., -4:-1], g["st_siren_coarse"][..., -4:-1], atol=2e-4)
    # ragged: 5 rays x 6 samples = 30 points per image (tile of 32 straddles the two images)
    o5, d5, z5 = o[:, :5].contiguous(), d[:, :5].contiguous(), z[:, :5].contiguous()
    sub = N_(nat.siren_forward_rays(o5, d5, z5, *tf))
    assert np.array_equal(sub, out[:, :5]), "per-point results must not depend on tiling"
    # explicit points == rays with identical inputs
    pts = (o[:, :, None, :] + d[:, :, None, :] * z[..., None]).reshape(B, R * N, 3)
    dd = d[:, :, None, :].expand(B, R, N, 3).reshape(B, R * N, 3).contiguous()
    ep = N_(nat.siren_forward(pts.contiguous(), dd, *tf))
    np.testing.assert_allclose(ep, out.reshape(B, R * N, -1), atol=2e-5)
    # lock_view_dependence: dirs := (0,0,-1)   (generators.py:474-476)
    lk = N_(nat.siren_forward_rays(o, d, z, *tf, lock_view=True))
    lock_dirs = np.zeros((B, R * N, 3), np.float32)
    lock_dirs[..., -1] = -1
    ref = O.siren_forward(sd, spec, N_(pts), lock_dirs, film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    np.testing.assert_allclose(lk.reshape(B, R * N, -1)[..., -4:-1], ref[..., -4:-1], atol=2e-4)
    lk2 = N_(nat.siren_forward(pts.contiguous(), None, *tf))
    np.testing.assert_allclose(lk2, lk.reshape(B, R * N, -1), atol=2e-5)
    # sigma and labels do not depend on the view direction or the grid
    np.testing.assert_array_equal(lk[..., -1], out[..., -1])
    # empty input
    e = nat.siren_forward(torch.empty((2, 0, 3), device=DEV), torch.empty((2, 0, 3), device=DEV), *tf)
    assert e.shape == (2, 0, 22)


@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
def test_siren_single_latent_spatial_model(precision):
    spec = proc.model_spec("spatial", hidden_dim=64, z_dim=8)
    sd = proc.make_state_dict(spec, seed=21, sigma_gain=100.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, precision)
    rng = np.random.default_rng(3)
    pts = rng.uniform(-0.12, 0.12, (2, 77, 3)).astype(np.float32)
    dirs = rng.normal(size=(2, 77, 3)).astype(np.float32)
    f = proc.normal("f9", (2, 9 * 64), 0.4, 21)
    p = proc.normal("p9", (2, 9 * 64), 0.4, 21)
    out = N_(nat.siren_forward(T(pts), T(dirs), T(f[:, :512]), T(p[:, :512]), T(f[:, -64:]), T(p[:, -64:])))
    ref = O.siren_forward(sd, spec, pts, dirs, f, p)
    np.testing.assert_allclose(out[..., :3], ref[..., :3], atol=1e-4)
    np.testing.assert_allclose(out[..., 3], ref[..., 3], atol=2e-3, rtol=2e-4)


# ---------------------------------------------------------------------------------------------------
# a13: fancy_integration -- every flag combination the reference's own golden block covers
# ---------------------------------------------------------------------------------------------------
def test_composite_variants_vs_reference():
    g = load_golden("integration_variants")
    rs, z = T(g["rgb_sigma"]), T(g["z_vals"])
    worst = 0.0
    for i in range(int(g["n_variants"])):
        kw = ast.literal_eval(str(g[f"v{i}_kw"]))
        draws = VR.RecordedDraws([g[f"v{i}_noise"]])
        rgb, depth, third = VR.fancy_integration(rs, z, DEV, draws=draws, **kw)
        for a, b in ((rgb, g[f"v{i}_rgb"]), (depth, g[f"v{i}_depth"]), (third, g[f"v{i}_third"])):
            err = np.abs(N_(a) - b).max()
            worst = max(worst, err)
            assert N_(a).shape == b.shape and err < 1e-5, (kw, err)
    print(f"[parity] composite: {int(g['n_variants'])} variants, worst max|err| {worst:.3e}")
    rgb, depth, third = VR.fancy_integration(T(g["ewb_rgb_sigma"]), z, DEV, noise_std=0.0, clamp_mode="relu", fill_mode="eval_white_back")
    np.testing.assert_allclose(N_(rgb), g["ewb_rgb"], atol=1e-5)
    np.testing.assert_allclose(N_(third), g["ewb_third"], atol=1e-5)
    with pytest.raises(TypeError):
        VR.fancy_integration(rs, z, DEV, clamp_mode=None)
    with pytest.raises(RuntimeError):
        VR.fancy_integration(rs, z, DEV, clamp_mode="relu", fill_mode="debug")
    # C-ABI level error code for a missing clamp mode
    o = _lib.composite_opts("relu")
    o.clamp_mode = 0
    with pytest.raises(_lib.FenerfError) as ei:
        native.composite(rs, z.squeeze(-1), None, o)
    assert ei.value.code == _lib.E_CLAMP_MODE


def test_composite_max_samples_and_empty():
    """M = 1024 (sixteen samples per lane, FENERF_MAX_RAY_SAMPLES; the smallest of the 2- / 4- / 8- / 16-slot kernels that holds the ray
    runs), slot boundaries, M = 1, zero rays."""
    rng = np.random.default_rng(5)
    for M in (1024, 961, 770, 513, 512, 449, 257, 256, 193, 129, 128, 65, 64, 1):
        rs = rng.normal(size=(3, 7, M, 22)).astype(np.float32)
        rs[..., -1] *= 30
        z = np.sort(rng.uniform(0.88, 1.12, (3, 7, M, 1)).astype(np.float32), axis=2)
        opts = _lib.composite_opts("relu", last_back=(M % 2 == 0))
        rgb, depth, w, ws = native.composite(T(rs), T(z[..., 0]), None, opts)
        r_rgb, r_depth, r_w = O.fancy_integration(rs, z, clamp_mode="relu", last_back=(M % 2 == 0))
        np.testing.assert_allclose(N_(rgb), r_rgb, atol=2e-5)
        np.testing.assert_allclose(N_(depth), r_depth[..., 0], atol=2e-5)
        if M > 1:
            np.testing.assert_allclose(N_(w), r_w[..., 0], atol=1e-5)
        else:   # reference quirk: M == 1 composites to zero (empty deltas tensor, volumetric_rendering.py:23-25)
            assert (N_(w) == 0).all() and (N_(rgb) == 0).all()
    e = native.composite(torch.empty((0, 4, 22), device=DEV), torch.empty((0, 4), device=DEV), None, _lib.composite_opts("relu"))
    assert e[0].shape == (0, 21)
    with pytest.raises(_lib.FenerfError):
        native.composite(torch.zeros((1, 1025, 22), device=DEV), torch.zeros((1, 1025), device=DEV), None, _lib.composite_opts("relu"))


# ---------------------------------------------------------------------------------------------------
# a14 / a15: sample_pdf, resample, merge
# ---------------------------------------------------------------------------------------------------
def test_sample_pdf_and_resample_vs_reference():
    g = load_golden("sample_pdf_cases")
    for i in range(int(g["n_cases"])):
        draws = VR.RecordedDraws([g[f"c{i}_u"]])
        s = VR.sample_pdf(T(g[f"c{i}_bins"]), T(g[f"c{i}_weights"]), g[f"c{i}_u"].shape[1], det=False, draws=draws)
        np.testing.assert_allclose(N_(s), g[f"c{i}_samples"], atol=3e-6)
    draws = VR.RecordedDraws([g["edge_u"]])
    s = VR.sample_pdf(T(g["edge_bins"]), T(g["edge_weights"]), 5, draws=draws)
    np.testing.assert_allclose(N_(s), g["edge_samples"], atol=3e-6)
    for name in ("tiny_texture_fwd", "h256_texture_16x16_n24_trained"):
        g = load_golden(name)
        B, R, N = g["st_z_coarse"].shape[:3]
        zf = native.resample(T(g["st_z_coarse"].reshape(B * R, N)), T(g["st_coarse_weights"].reshape(B * R, N)), T(g["rand_u_fine"]))
        err = np.abs(N_(zf) - g["st_z_fine"]).max()
        print(f"[parity] resample {name}: max|err| {err:.3e}")
        # conditioning: z = b0 + (u-c0)/denom*(b1-b0); an fp32 rounding difference in the cdf (parallel scan vs the
        # reference's sequential cumsum / pairwise sum) is amplified by bin_width/denom, denom >= 1e-5 by the reference's
        # own clamp -> worst case 6e-8 * 0.01 / 1e-5 = 6e-5 on samples that land in (near-)empty bins.
        assert err < (3e-6 if "tiny" in name else 6e-5)
    with pytest.raises(_lib.FenerfError):
        native.resample(torch.zeros((4, 2), device=DEV), torch.zeros((4, 2), device=DEV), torch.zeros((4, 2), device=DEV))


@pytest.mark.parametrize("N", [129, 200, 256, 257, 400, 512])
def test_more_than_128_samples_per_pass(N):
    """The reference has no limit on num_steps; one wave per ray handles up to 512 + 512 (FENERF_MAX_RAY_SAMPLES): resample (four / eight
    64-sample slots), the rank-merge composite on 2 N <= 1024 samples, and the whole render, against the oracle (the backward at these
    sizes: test_composite_backward_vs_autograd, test_merge_composite_backward_vs_autograd)."""
    rng = np.random.default_rng(N)
    BR, C = 9, 22
    z_c = np.sort(rng.uniform(0.88, 1.12, (BR, N)).astype(np.float32), axis=1)
    w_c = (rng.random((BR, N)).astype(np.float32) ** 4)
    w_c[0, 40:90] = 0                                            # a stretch of empty bins
    u = rng.random((BR, N)).astype(np.float32)
    zf = native.resample(T(z_c), T(w_c), T(u))
    ref = O.fine_z_from_coarse(w_c.reshape(1, BR, N, 1), z_c.reshape(1, BR, N, 1), u).reshape(BR, N)
    # conditioning as in test_sample_pdf_and_resample_vs_reference: cdf rounding x bin_width / denom
    print(f"[parity] resample N={N}: max|err| {np.abs(N_(zf) - ref).max():.3e}")
    np.testing.assert_allclose(N_(zf), ref, atol=6e-5)
    assert np.abs(N_(zf) - ref).mean() < 1e-6
    # sample_pdf in its reference shape (bins [BR, K + 1], weights [BR, K]) with as many knots and draws
    bins = np.sort(rng.uniform(0.88, 1.12, (BR, N)).astype(np.float32), axis=1)
    wk = (rng.random((BR, N - 1)).astype(np.float32) ** 4)
    s = native.sample_pdf(T(bins), T(wk), T(u))
    rs_ = O.sample_pdf(bins, wk, u)
    np.testing.assert_allclose(N_(s), rs_, atol=6e-5)
    assert np.abs(N_(s) - rs_).mean() < 1e-6
    fine = rng.normal(size=(BR, N, C)).astype(np.float32); coarse = rng.normal(size=(BR, N, C)).astype(np.float32)
    fine[..., -1] *= 30; coarse[..., -1] *= 30
    z_f = N_(zf)
    opts = _lib.composite_opts("relu")
    rgb, depth, w, ws, zs = native.merge_composite(T(fine), T(coarse), T(z_f), T(z_c), None, opts)
    all_out, all_z = O.merge_sorted(fine[None], coarse[None], z_f[None, ..., None], z_c[None, ..., None])
    r_rgb, r_depth, r_w = O.fancy_integration(all_out, all_z, clamp_mode="relu")
    np.testing.assert_array_equal(N_(zs), all_z[0, ..., 0])
    np.testing.assert_allclose(N_(rgb), r_rgb[0], atol=2e-5)
    np.testing.assert_allclose(N_(depth), r_depth[0, ..., 0], atol=2e-5)
    np.testing.assert_allclose(N_(w), r_w[0, ..., 0], atol=1e-5)
    # the fused render (fenerf_render_forward: coarse SIREN, composite, resample, fine SIREN, merge) with N + N samples per ray
    spec = proc.model_spec("texture", hidden_dim=32, grid_size=5, z_dim=8)
    sd = proc.make_state_dict(spec, seed=2, sigma_gain=40.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, "f32")
    film = proc.film_params(spec, 1, seed=2)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    torch.manual_seed(N)
    o, d, z, _, _ = VR.sample_rays(1, N, DEV, 12, (3, 3), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    uu = torch.rand((9, N), device=DEV)
    rgb, depth, w, _ = nat.render(o, d, z, uu, None, None, *tf, opts, hierarchical=True, want_weights=True)
    oo, dd, zz, un = N_(o), N_(d), N_(z), N_(uu)
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    dexp = np.broadcast_to(dd[:, :, None, :], (1, 9, N, 3)).reshape(1, -1, 3)
    coarse_o = O.siren_forward(sd, spec, (oo[:, :, None, :] + dd[:, :, None, :] * zz[..., None]).reshape(1, -1, 3), dexp, *args).reshape(1, 9, N, -1)
    _, _, cw = O.fancy_integration(coarse_o, zz[..., None], clamp_mode="relu")
    zfo = O.fine_z_from_coarse(cw, zz[..., None], un)
    fine_o = O.siren_forward(sd, spec, (oo[:, :, None, :] + dd[:, :, None, :] * zfo).reshape(1, -1, 3), dexp, *args).reshape(1, 9, N, -1)
    ao, az = O.merge_sorted(fine_o, coarse_o, zfo, zz[..., None])
    r_rgb, r_depth, _ = O.fancy_integration(ao, az, clamp_mode="relu")
    err = np.abs(N_(rgb) - r_rgb).max()
    print(f"[parity] fused render with {N}+{N} samples per ray vs the oracle: max|err| {err:.2e}")
    assert err <= 1e-3 and np.abs(N_(depth) - r_depth[..., 0]).max() <= 1e-3


@pytest.mark.parametrize("name", ["tiny_texture_fwd", "tiny_baseline_fwd", "h256_texture_16x16_n24_trained", "tiny_texture_fwd_trained"])
def test_merge_composite_vs_reference(name):
    g = load_golden(name)
    B, R, N = g["st_z_coarse"].shape[:3]
    C = g["st_siren_coarse"].shape[-1]
    fine, coarse = g["st_siren_fine"].reshape(B * R, N, C), g["st_siren_coarse"].reshape(B * R, N, C)
    opts = _lib.composite_opts("relu")
    rgb, depth, w, ws, zs = native.merge_composite(T(fine), T(coarse), T(g["st_z_fine"]), T(g["st_z_coarse"].reshape(B * R, N)), None, opts)
    np.testing.assert_array_equal(N_(zs), g["st_all_z"].reshape(B * R, 2 * N))         # sort is exact
    np.testing.assert_allclose(N_(rgb), g["st_final_rgb"].reshape(B * R, -1), atol=1e-5)
    np.testing.assert_allclose(N_(depth), g["st_final_depth"].reshape(B * R), atol=1e-5)
    np.testing.assert_allclose(N_(w), g["st_final_third"].reshape(B * R, 2 * N), atol=1e-5)
    np.testing.assert_allclose(N_(ws), N_(w).sum(-1), atol=1e-5)
    assert (np.diff(N_(zs), axis=-1) >= 0).all()


# ---------------------------------------------------------------------------------------------------
# a15-a17: the generator API end to end, teacher-forced with the reference's recorded random draws
# ---------------------------------------------------------------------------------------------------
def _make_generator(g, spec, precision="f32"):
    H = spec["hidden_dim"]
    cls = {"texture": S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, "baseline": S.SIRENBASELINESEMANTICDISENTANGLE}[spec["kind"]]
    gen = G.DoubleImplicitGenerator3d(functools.partial(cls, hidden_dim=H), spec.get("z_dim", 256), spec.get("z_dim", 256), 22)
    sd = proc.make_state_dict(dict(spec, z_dim=spec.get("z_dim", 256), map_hidden=256), seed=int(g["meta_seed"]) if "meta_seed" in g else 3,
                              sigma_gain=float(g["meta_sigma_gain"]) if "meta_sigma_gain" in g else 300.0)
    st = state_from_golden(g)
    if st is not None:      # a state the reference's own Adam run produced (round 5): its render weights replace the procedural ones
        sd.update(st[0])
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    if "spatial_embeddings" in tsd:
        gen.siren.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
    gen.siren.load_state_dict(tsd, strict=True)
    gen = gen.to(DEV).eval()
    gen.siren.precision = precision
    gen.device = torch.device(DEV)
    gen.siren.device = gen.device
    return gen


# max |pixel error| of the end-to-end fixtures as measured in round 4 (both precisions, every box: the library is deterministic) -- the
# asserted bound is 1.5 x this, on top of north_star's 1e-3 (h256_texture_16x16_n24_trained is the closest to the bar: fp32 re-association at
# a synthetic |sigma| ~ 200 density scale, identical in the exact-fp32 kernel).  A fixture not listed asserts the bar alone.
E2E_MEASURED = {"tiny_texture_fwd_trained": 1.5e-5, "tiny_texture_fwd": 1.7e-6, "tiny_texture_fwd_nohier": 3.1e-6, "tiny_baseline_fwd": 1e-6, "h256_texture_16x16_n12": 2.0e-5,
                "h256_texture_16x16_n24_trained": 6.6e-4, "h256_baseline_8x8_n12": 1.2e-4, "tiny_texture_staged": 1.7e-6,
                "tiny_texture_staged_lock": 9e-7, "forward(z)": 3.1e-5, "staged_forward(z, psi=0.7)": 1.4e-6}


# the same for the opt-in "f16x3c2" forward (profiles/r05_forward_modes.md, second table)
E2E_MEASURED_C2 = {"tiny_texture_fwd": 2.9e-5, "tiny_texture_fwd_nohier": 3.1e-5, "tiny_baseline_fwd": 1e-6, "h256_texture_16x16_n12": 2.8e-5,
                   "h256_texture_16x16_n24_trained": 6.6e-4, "h256_baseline_8x8_n12": 1.2e-4, "tiny_texture_fwd_trained": 5.5e-5,
                   "h96_texture_8x8_n12": 2.7e-5, "h192_baseline_8x8_n12": 2.1e-5}


def _e2e_check(tag, px, ref_px, tol=1e-3, max_flips=0):
    """north_star's bar, asserted as measured: |RGB / label error| <= tol on every pixel except `max_flips` named threshold flips
    (0 for every committed fixture -- none shows one), and the label argmax (mask2color, train_double_latent_semantic.py:66-72)
    identical on every non-flipped pixel that the reference itself decides (near-ties included; only exact reference ties, a
    margin of a few fp32 ulps between its two best logits, are exempt and counted)."""
    err = np.abs(px - ref_px).max(axis=1)
    bad = err > tol
    am, am_ref = px[:, :-3].argmax(1), ref_px[:, :-3].argmax(1)
    top2 = np.sort(ref_px[:, :-3], axis=1)
    tie = (top2[:, -1] - top2[:, -2]) <= 1e-6
    mism = (am != am_ref) & ~bad
    print(f"[parity] {tag}: max|err| {err[~bad].max():.3e} on {int((~bad).sum())}/{bad.size} pixels; {int(bad.sum())} pixels differ by "
          f"more than {tol} (fill-threshold / resampling flips; allowed {max_flips}); label argmax mismatches on the other pixels: "
          f"{int(mism.sum())} (reference ties: {int(tie.sum())})")
    assert int(bad.sum()) <= max_flips, f"{int(bad.sum())} pixels off by more than {tol}, worst {err.max():.3e}"
    measured = (E2E_MEASURED_C2 if "[f16x3c2]" in tag else E2E_MEASURED).get(tag.split("[")[0])
    if measured is not None and max_flips == 0:
        assert err.max() <= min(tol, 1.5 * measured), f"{tag}: worst pixel {err.max():.3e}, measured {measured:.1e} in round 4"
    assert not (mism & ~tie).any(), "exact argmax semantics on every pixel the reference itself decides"
    return bad


@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
@pytest.mark.parametrize("name", ["tiny_texture_fwd", "tiny_texture_fwd_nohier", "tiny_baseline_fwd", "h256_texture_16x16_n12",
                                  "h256_texture_16x16_n24_trained", "h256_baseline_8x8_n12", "tiny_texture_fwd_trained",
                                  "h96_texture_8x8_n12", "h192_baseline_8x8_n12"])
def test_forward_with_frequencies_vs_reference(name, precision):
    g = load_golden(name)
    spec = spec_from_golden(g)
    gen = _make_generator(g, dict(spec, z_dim=spec.get("z_dim", 16) if spec["hidden_dim"] == 32 else 256), precision)
    name = f"{name}[{precision}]"
    film, tf = _film(g, spec)
    hier = bool(g["meta_hier"])
    seq = [g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"]]
    if hier:
        seq += [g["rand_u_fine"], g["rand_noise_fine"]]
    gen.draws = VR.RecordedDraws(seq)
    kw = kwargs_from_golden(g)
    with torch.no_grad():
        px, poses = gen.forward_with_frequencies(tf[0], tf[2], tf[1], tf[3], img_size=int(g["meta_S"]), fov=12, ray_start=0.88,
                                                 ray_end=1.12, num_steps=int(g["meta_N"]), h_stddev=0.3, v_stddev=0.155,
                                                 h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, hierarchical_sample=hier,
                                                 sample_dist="gaussian", **kw)
    assert not gen.draws.arrays, "all recorded draws consumed, in order"
    assert px.shape == g["pixels"].shape and px.is_cuda
    np.testing.assert_allclose(N_(poses), g["poses"], atol=1e-6)
    _e2e_check(name, N_(px), g["pixels"])


@pytest.mark.parametrize("name", ["tiny_texture_staged", "tiny_texture_staged_lock"])
def test_staged_forward_with_frequencies_vs_reference(name):
    g = load_golden(name)
    spec = spec_from_golden(g)
    gen = _make_generator(g, dict(spec, z_dim=16))
    film, tf = _film(g, spec)
    gen.draws = VR.RecordedDraws([g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"], g["rand_u_fine"],
                                  g["rand_noise_fine"]])
    kw = kwargs_from_golden(g)
    px, depth, third = gen.staged_forward_with_frequencies(tf[0], tf[2], tf[1], tf[3], img_size=int(g["meta_S"]), fov=12,
                                                           ray_start=0.88, ray_end=1.12, num_steps=int(g["meta_N"]), h_stddev=0.3,
                                                           v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5,
                                                           hierarchical_sample=True, sample_dist="gaussian", max_batch_size=1000, **kw)
    assert not px.is_cuda and px.shape == g["pixels"].shape and third.shape == g["third"].shape
    bad = _e2e_check(name, N_(px), g["pixels"])
    np.testing.assert_allclose(N_(depth)[~bad], g["depth"][~bad], atol=1e-4)
    terr = np.abs(N_(third) - g["third"]).max(axis=1)
    assert (terr[~bad] < 2e-3).all()


def test_forward_and_staged_forward_from_latents():
    """z -> mapping nets (PyTorch) -> render; staged_forward's truncation with the reference's avg frequencies."""
    g = load_golden("tiny_texture_z_full")
    spec = spec_from_golden(g)
    gen = _make_generator(dict(meta_seed=3, meta_sigma_gain=300.0), spec)
    zg, za = T(g["z_geo"]), T(g["z_app"])
    with torch.no_grad():
        fg, pg = gen.siren.geo_mapping_network(zg)
    np.testing.assert_allclose(N_(fg), g["map_freq_geo"], atol=5e-5)
    kw = dict(img_size=6, num_steps=6, hierarchical_sample=True, clamp_mode="relu", nerf_noise=0.0, fov=12, ray_start=0.88,
              ray_end=1.12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, sample_dist="gaussian")
    rd = {k[len("fwd_rand_"):]: v for k, v in g.items() if k.startswith("fwd_rand_")}
    gen.draws = VR.RecordedDraws([rd["u_jitter"], rd["r_theta"], rd["r_phi"], rd["noise_coarse"], rd["u_fine"], rd["noise_fine"]])
    with torch.no_grad():
        px, poses = gen(zg, za, **kw)
    np.testing.assert_allclose(N_(poses), g["fwd_poses"], atol=1e-6)
    _e2e_check("forward(z)", N_(px), g["fwd_pixels"])
    # staged_forward: first two draws are the 10000-z avg pass; replace its result by the reference's recorded means
    rd = {k[len("stg_rand_"):]: v for k, v in g.items() if k.startswith("stg_rand_")}
    gen.draws = VR.RecordedDraws([np.zeros((10000, 16), np.float32)] * 2 +
                                 [rd["u_jitter"], rd["r_theta"], rd["r_phi"], rd["noise_coarse"], rd["u_fine"], rd["noise_fine"]])
    orig = gen.generate_avg_frequencies

    def patched():
        orig()
        gen.avg_frequencies_geo, gen.avg_phase_shifts_geo = T(g["stg_avg_freq_geo"]), T(g["stg_avg_phase_geo"])
        gen.avg_frequencies_app, gen.avg_phase_shifts_app = T(g["stg_avg_freq_app"]), T(g["stg_avg_phase_app"])
    gen.generate_avg_frequencies = patched
    px, depth = gen.staged_forward(zg, za, psi=float(g["stg_psi"]), max_batch_size=97, fill_mode="seg_padding_background",
                                   fill_color="white", **kw)
    assert px.shape == g["stg_pixels"].shape and not px.is_cuda
    bad = _e2e_check("staged_forward(z, psi=0.7)", N_(px), g["stg_pixels"])
    np.testing.assert_allclose(N_(depth)[~bad], g["stg_depth"][~bad], atol=1e-4)


# ---------------------------------------------------------------------------------------------------
# BASELINE.json sizes: size-independent properties + oracle spot check on a ray subset
# ---------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def _full_weights():
    spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
    return spec, proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)


def _oracle_render_rays(sd, spec, args, oo, dd, zz, uu, fill_color, hier=True, slab=2048, dtype=np.float32):
    """numpy oracle of the fused render on explicit rays: oo / dd [B,R,3], zz [B,R,N], uu [B*R,N] -> (pixels [B,R,22], depth [B,R],
    the composited sample depths [B,R,M]).  dtype=np.float64: the same statements in double precision on the same fp32 inputs (the
    arbiter of test_resampling_flips_against_an_fp64_arbiter)."""
    B, R, N = zz.shape
    assert B == 1
    if np.dtype(dtype) != np.float32:
        oo, dd, zz = (a.astype(dtype) for a in (oo, dd, zz))
        uu = uu.astype(dtype) if uu is not None else None
    def one_slab(s0):
        sl = slice(s0, min(R, s0 + slab))
        o_, d_, z_ = oo[:, sl], dd[:, sl], zz[:, sl]
        n = z_.shape[1]
        pts = (o_[:, :, None, :] + d_[:, :, None, :] * z_[..., None]).reshape(B, -1, 3)
        dexp = np.broadcast_to(d_[:, :, None, :], (B, n, N, 3)).reshape(B, -1, 3)
        coarse = O.siren_forward(sd, spec, pts, dexp, *args, dtype=dtype).reshape(B, n, N, -1)
        if hier:
            _, _, cw = O.fancy_integration(coarse, z_[..., None], clamp_mode="relu")
            zf = O.fine_z_from_coarse(cw, z_[..., None], uu[sl])
            fine = O.siren_forward(sd, spec, (o_[:, :, None, :] + d_[:, :, None, :] * zf).reshape(B, -1, 3), dexp, *args, dtype=dtype).reshape(B, n, N, -1)
            ao, az = O.merge_sorted(fine, coarse, zf, z_[..., None])
        else:
            ao, az = coarse, z_[..., None]
        r_rgb, r_depth, r_w = O.fancy_integration(ao, az, clamp_mode="relu", fill_mode="seg_padding_background", fill_color=fill_color)
        return r_rgb, r_depth[..., 0], az[..., 0], r_w[..., 0].sum(-1)

    # The slabs are independent and every slab's arithmetic is what a serial walk would do; the oracle's time is numpy's single-threaded
    # sin / elementwise passes (2,816 sines per point), which release the GIL: walk the slabs on a few host threads (the GPU box has 256 cores).
    starts = list(range(0, R, slab))
    workers = max(1, min(len(starts), (os.cpu_count() or 8) // 8, 24))
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            parts = list(pool.map(one_slab, starts))
    else:
        parts = [one_slab(s0) for s0 in starts]
    px, dp, zs, ws = (np.concatenate([p[i] for p in parts], 1) for i in range(4))
    _oracle_render_rays.weights_sum = ws     # [B,R]: how close every ray is to the 0.9 fill threshold (volumetric_rendering.py:84)
    return px, dp, zs


def _check_render_vs_oracle(tag, nat, rays, tf, opts, rgb, depth, r_rgb, r_depth, r_z, hier=True, max_over=0, over_bound=1e-3, depth_bound=2e-3):
    """north_star's bar on a full render against the oracle, asserted as measured.  A ray is a RESAMPLING FLIP when the native
    pipeline and the oracle put a fine sample into different bins of the inverse-CDF (u within fp32 rounding of a knot; on background
    rays, whose coarse weights are ~1e-7 + the 1e-5 floor, the whole cdf moves with the rounding of those weights): both are valid
    evaluations of the same algorithm, the composited depths then differ by up to a bin while the pixel hardly moves.
      * pixels: at most `max_over` rays beyond 1e-3, none beyond `over_bound`;
      * depth: <= `depth_bound` on every ray with identical resampling;
        -- every call site passes its own (max_over, over_bound, depth_bound) = what was measured for that shape and precision x 1.5
        (round 5; the library is deterministic, so the same build measures the same numbers on every box: round 4's common 2 rays /
        2e-3 / 2e-3 would have let a 2 - 100 x regression through);
      * label argmax identical on every ray with identical resampling whose two best oracle logits are not tied."""
    import __graft_entry__ as ge
    o, d, z, u = rays
    err = np.abs(rgb - r_rgb).max(-1)
    if hier:
        zs = ge.nat_sorted_z(nat, o, d, z, u, tf, opts)
        flip = np.abs(zs - r_z).max(-1) > 1e-5
    else:
        flip = np.zeros(err.shape, bool)
    # FILL-THRESHOLD rays: seg_padding_background paints a ray whose weights_sum is below 0.9 (volumetric_rendering.py:84); a ray whose
    # oracle weights_sum is within 2e-5 (fp32 rounding of a 48 .. 96-term sum) of 0.9 may be painted by one evaluation and not by the other
    # -- a whole-pixel difference that is the reference's own discontinuity.  Such rays are excluded, counted and bounded: <= 2 per 16,384.
    r_ws = getattr(_oracle_render_rays, "weights_sum", None)
    thr = ((rgb[..., 0] == 1) != (r_rgb[..., 0] == 1))
    if thr.any():
        assert r_ws is not None and r_ws.shape == thr.shape and (np.abs(r_ws[thr] - 0.9) <= 2e-5).all(), "a fill decision differs away from the 0.9 threshold"
        assert int(thr.sum()) <= max(2, err.size // 8192)
        print(f"[parity] {tag}: {int(thr.sum())} ray(s) sit on the 0.9 fill threshold (oracle weights_sum {r_ws[thr].tolist()}) and are painted by one side only: excluded")
        keep = ~thr
        err, flip, rgb, r_rgb, depth, r_depth = err[keep][None], flip[keep][None], rgb[keep][None], r_rgb[keep][None], depth[keep][None], r_depth[keep][None]
    over = err > 1e-3
    derr = np.abs(depth - r_depth)
    print(f"[parity] {tag}: max|err| {err.max():.3e} over {err.size} rays; {int(over.sum())} rays > 1e-3 ({int((over & flip).sum())} of them "
          f"resampling flips); {int(flip.sum())} rays resample differently (their max pixel error {err[flip].max() if flip.any() else 0:.3e}, "
          f"depth error {derr[flip].max() if flip.any() else 0:.3e}); depth max|err| elsewhere {derr[~flip].max():.3e}")
    assert int(over.sum()) <= max_over and err.max() <= over_bound, f"{int(over.sum())} rays off by more than 1e-3 (worst {err.max():.3e})"
    assert derr[~flip].max() <= depth_bound, f"depth off by {derr[~flip].max():.3e} on a ray with identical resampling (bound {depth_bound:.1e})"   # (depth is not part of north_star's bar)
    lab, r_lab = rgb[..., 1:-3], r_rgb[..., 1:-3]
    top2 = np.sort(r_lab, axis=-1)
    decided = ((top2[..., -1] - top2[..., -2]) > 1e-6) & ~flip & (r_rgb[..., 0] != 1)
    assert (lab.argmax(-1) == r_lab.argmax(-1))[decided].all(), "exact argmax semantics on every ray the oracle itself decides"
    # the rays beyond 1e-3 as indices into the caller's rays (fill-threshold rays excluded above): the caller may have to account for each
    idx_all = np.flatnonzero(~thr.reshape(-1)) if thr.any() else np.arange(over.size)
    return idx_all[np.flatnonzero(over.reshape(-1))], zs if hier else None


# the rays of the bench image (seed 0) whose pixel differs from the fp32 oracle's by more than 1e-3, by index (measured on an MI355X; the
# library is deterministic).  f16x3: ray 9880, 1.309e-3 from the fp32 oracle -- a resampling flip (asserted against the fp64 arbiter inside the
# test: its merged sample depths are 2.7e-4 from fp64's, the fp32 oracle's own 1.15e-3; its pixel is 8.9e-4 from fp64's, the oracle's 2.2e-3)
EXPECTED_RAYS_BEYOND_1E3 = {"f32": (), "f16x3": (9880,), "f16x3c2": (9880,)}      # f16x3c2: sigma is f16x3's, bit for bit -> the same ray


@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
def test_full_size_128_24p24_properties_and_oracle_all_rays(precision):
    spec, sd = _full_weights()
    nat = native.NativeModel(sd, spec, DEV, precision)
    B, S_, N = 1, 128, 24
    R = S_ * S_
    film = proc.film_params(spec, B, seed=0)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    torch.manual_seed(0)
    o, d, z, pitch, yaw = VR.sample_rays(B, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * R, N), device=DEV)
    opts = _lib.composite_opts("relu", fill_mode="seg_padding_background", fill_color="white")
    rgb, depth, w, ws = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True, want_weights=True, want_wsum=True)
    rgb2, depth2, w2, _ = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True, want_weights=True)
    assert torch.equal(rgb, rgb2) and torch.equal(depth, depth2) and torch.equal(w, w2), "deterministic"
    rgb, depth, w, ws = N_(rgb), N_(depth), N_(w), N_(ws)
    assert np.isfinite(rgb).all() and np.isfinite(depth).all()
    np.testing.assert_allclose(w.sum(-1), ws, atol=1e-5)
    assert (w >= 0).all() and (ws <= 1 + 1e-4).all()
    filled = rgb[..., 0] == 1
    assert ((ws < 0.9) == filled).all(), "background channel set exactly on rays below the 0.9 threshold"
    assert (rgb[filled][:, 1:] == 1).all()
    assert ((depth >= -1e-6) & (depth <= 1.12 * (1 + 1e-4) + 0.02)).all()
    assert ((rgb[~filled][:, -3:] >= 0) & (rgb[~filled][:, -3:] <= 1 + 1e-4)).all()
    # sub-batch independence at full size: rows [5000, 5200) alone == the same rows of the full render (bit-exact)
    sl = slice(5000, 5200)
    r3, d3, w3, _ = nat.render(o[:, sl].contiguous(), d[:, sl].contiguous(), z[:, sl].contiguous(), u[sl].contiguous(), None, None, *tf,
                               opts, hierarchical=True, want_weights=True)
    assert np.array_equal(N_(r3), rgb[:, sl]) and np.array_equal(N_(w3), w[:, sl])
    # the oracle on ALL 16,384 rays of the bench workload (same inputs, plain end to end: coarse -> weights -> resample -> fine ->
    # merge -> composite), in slabs of 2,048 rays to bound the numpy activations; ~10 s on the GPU box's host cores
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    r_rgb, r_depth, r_z = _oracle_render_rays(sd, spec, args, N_(o), N_(d), N_(z), N_(u), "white")
    # measured (round 4, every box): f32 max 5.60e-4, 0 rays > 1e-3, depth 1.73e-3; f16x3 ONE ray at 1.309e-3 (a resampling flip: the fp64
    # arbiter below), depth 9.04e-4
    bounds = dict(f32=dict(max_over=0, over_bound=8.5e-4, depth_bound=2.6e-3), f16x3=dict(max_over=1, over_bound=2.0e-3, depth_bound=1.4e-3),
                  f16x3c2=dict(max_over=1, over_bound=2.0e-3, depth_bound=1.4e-3))[precision]      # f16x3c2: sigma is f16x3's -> the same ray, the same depths
    over_idx, zs = _check_render_vs_oracle(f"128x128 24+24 H=256 [{precision}] vs oracle on ALL {R} rays", nat, (o, d, z, u), tf, opts, rgb, depth, r_rgb,
                                           r_depth, r_z, **bounds)
    # Round 6: the allowance is not anonymous.  Every ray beyond north_star's 1e-3 must be (a) THE ray measured before -- the library is
    # deterministic: another index is another error -- and (b) a resampling flip under the fp64 arbiter, in THIS test: the oracle in fp64 on
    # that ray alone puts a merged sample > 1e-5 away from where the fp32 oracle or the native pipeline puts it (both are valid evaluations
    # of a discontinuous algorithm; the fp32 oracle itself is 2.9e-3 from fp64 on such rays, test_resampling_flips_against_an_fp64_arbiter)
    assert set(over_idx.tolist()) <= set(EXPECTED_RAYS_BEYOND_1E3[precision]), (over_idx.tolist(), EXPECTED_RAYS_BEYOND_1E3[precision])
    for i in over_idx.tolist():
        sl = slice(i, i + 1)
        p64, _, z64 = _oracle_render_rays(sd, spec, args, N_(o)[:, sl], N_(d)[:, sl], N_(z)[:, sl], N_(u)[sl], "white", dtype=np.float64)
        f_nat, f_32 = float(np.abs(zs[:, i] - z64[:, 0]).max()), float(np.abs(r_z[:, i] - z64[:, 0]).max())
        print(f"[parity] ray {i} beyond 1e-3 [{precision}]: merged sample depths differ from the fp64 arbiter's by {f_nat:.2e} (native) / {f_32:.2e} (fp32 oracle); "
              f"pixel error vs fp64 {np.abs(rgb[:, i] - p64[:, 0]).max():.2e} (native) / {np.abs(r_rgb[:, i] - p64[:, 0]).max():.2e} (fp32 oracle)")
        assert max(f_nat, f_32) > 1e-5, f"ray {i} is beyond 1e-3 without being a resampling flip under fp64"


@pytest.mark.parametrize("S_,N", [(128, 24), (64, 48)])
def test_resampling_flips_against_an_fp64_arbiter(S_, N):
    """(64 x 64 x 48+48, round 5: configs[4]'s sampling density -- its 3,100 differently-resampled rays of 65,536 get the same backing as
    configs[1]'s; asserted there with the generic factors of round 4 until a round has measured them.)
    DESIGN.md 2 says of a ray whose fine samples land in other bins than the fp32 oracle's: "both are valid evaluations of the same
    algorithm".  That needs an arbiter: the oracle in fp64 on the same fp32 inputs (rays, draws, weights, FiLM parameters).  On the bench
    image (all 16,384 rays, 128x128, 24+24, H = 256 + 96^3 grid, |sigma| ~ 2000) a ray FLIPS against fp64 when its merged sample depths
    differ from the fp64 ones by more than 1e-5.  Measured (round 4): the fp32 oracle -- the reference's own arithmetic -- flips 278 rays
    against fp64, the native pipeline 266 (237 of them the same rays); a flipped ray is a silhouette ray (a sample moved across the
    |sigma| ~ 2000 surface), its pixel moves by up to 2.9e-3 (oracle) / 2.5e-3 (native) and its depth by up to 0.09 / 0.11.  Asserted, for
    both precisions:
      * the native pipeline flips no more rays against fp64 than the fp32 oracle does (measured 261 / 257 against 278);
      * on rays that agree with fp64 in their sample positions: pixels <= 6e-5 (measured 3.1e-5 / 2.8e-5; the fp32 oracle 2.6e-5) and
        depth <= 1.7e-3 / 1.0e-3 (f32 / f16x3; measured 1.13e-3 / 6.2e-4) on ALL of them -- so every ray beyond north_star's 1e-3 is a flip;
      * on rays that flip: pixel and depth errors are those of the fp32 oracle on ITS flipped rays: max and mean pixel error within 1.2 x
        (measured 0.77 - 1.00 x), max depth error within 1.25 x (0.69 - 1.03 x), mean depth error within 1.5 x (0.97 - 1.28 x: different ray
        sets) -- the native result is as close to fp64 as the reference's arithmetic is (round 5: the factors were a common 1.5);
      * fill decisions identical to fp64 on every ray."""
    import __graft_entry__ as ge
    spec, sd = _full_weights()
    B = 1
    R = S_ * S_
    bench = (S_, N) == (128, 24)
    film = proc.film_params(spec, B, seed=0)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    torch.manual_seed(0 if bench else 3)
    o, d, z, _, _ = VR.sample_rays(B, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * R, N), device=DEV)
    opts = _lib.composite_opts("relu", fill_mode="seg_padding_background", fill_color="white")
    px32, dp32, z32 = _oracle_render_rays(sd, spec, args, N_(o), N_(d), N_(z), N_(u), "white")
    px64, dp64, z64 = _oracle_render_rays(sd, spec, args, N_(o), N_(d), N_(z), N_(u), "white", dtype=np.float64)
    ws64 = _oracle_render_rays.weights_sum
    flip32 = np.abs(z32 - z64).max(-1) > 1e-5
    e32, d32 = np.abs(px32 - px64).max(-1), np.abs(dp32 - dp64)
    assert flip32.any() and e32[~flip32].max() <= 1e-3
    print(f"[parity] fp64 arbiter {S_}x{S_} {N}+{N}, fp32 oracle (the reference's arithmetic): {int(flip32.sum())} of {R} rays resample differently from fp64; "
          f"pixel error on them max {e32[flip32].max():.2e} mean {e32[flip32].mean():.2e}, elsewhere {e32[~flip32].max():.2e}; depth error on them "
          f"max {d32[flip32].max():.2e} mean {d32[flip32].mean():.2e}, elsewhere {d32[~flip32].max():.2e}")
    for precision in FORWARD_PRECISIONS:
        nat = native.NativeModel(sd, spec, DEV, precision)
        rgb, depth, _, _ = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
        rgb, depth = N_(rgb), N_(depth)
        zs = ge.nat_sorted_z(nat, o, d, z, u, tf, opts)
        flip = np.abs(zs - z64).max(-1) > 1e-5
        both = flip & flip32
        err, derr = np.abs(rgb - px64).max(-1), np.abs(depth - dp64)
        err = np.where((rgb[..., 0] == 1) != (px64[..., 0] == 1), 0.0, err)        # rays ON the fill threshold: asserted separately below
        over = err > 1e-3
        print(f"[parity] fp64 arbiter {S_}x{S_} {N}+{N}, native {precision}: {int(flip.sum())} rays resample differently from fp64 ({int(both.sum())} of them are the fp32 "
              f"oracle's flips too); pixel error on them max {err[flip].max():.2e} mean {err[flip].mean():.2e}, elsewhere {err[~flip].max():.2e} "
              f"({int(over.sum())} rays > 1e-3, all of them flips: {bool((over & ~flip).sum() == 0)}); depth error on them max {derr[flip].max():.2e} "
              f"mean {derr[flip].mean():.2e}, elsewhere {derr[~flip].max():.2e}")
        if bench:
            assert int(flip.sum()) <= int(flip32.sum()), "the native pipeline resamples differently from fp64 more often than the reference's fp32 arithmetic does"
            assert err[~flip].max() <= 6e-5 and derr[~flip].max() <= dict(f32=1.7e-3, f16x3=1.0e-3, f16x3c2=1.0e-3)[precision] and not (over & ~flip).any()
            assert err[flip].max() <= 1.2 * e32[flip32].max() and err[flip].mean() <= 1.2 * e32[flip32].mean()
            assert derr[flip].max() <= 1.25 * d32[flip32].max() and derr[flip].mean() <= 1.5 * d32[flip32].mean()
        else:
            # 64 x 64 x 48+48 (measured, round 5): the fp32 oracle flips 189 rays against fp64, native f32 165 (156 shared); on flipped rays the
            # oracle's pixels are off by up to 3.0e-4 (mean 4.3e-6), the native ones by 5.1e-4 (7.8e-6) -- a handful of rays decide the maxima, and
            # every one of them is still inside north_star's 1e-3 --; elsewhere 1.1e-5 / 6.2e-6, depth 5.4e-4 / 1.4e-4
            assert int(flip.sum()) <= int(flip32.sum())
            assert err[~flip].max() <= 2e-5 and derr[~flip].max() <= 6e-4 and not over.any()
            assert err[flip].max() <= 1e-3 and err[flip].mean() <= 2.5 * e32[flip32].mean()
            assert derr[flip].max() <= 3 * d32[flip32].max() and derr[flip].mean() <= 2.5 * d32[flip32].mean()
        thr = (rgb[..., 0] == 1) != (px64[..., 0] == 1)
        assert int(thr.sum()) <= 2 and (np.abs(ws64[thr] - 0.9) <= 2e-5).all(), "fill decisions agree with fp64 on every ray that is not ON the 0.9 threshold"


# ---------------------------------------------------------------------------------------------------
# a1-a5: HIP ray setup (fenerf_ray_setup) vs the reference's recorded rays
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny_texture_fwd", "tiny_texture_fwd_nohier", "h256_texture_16x16_n24_trained"])
def test_ray_setup_kernel_vs_reference(name):
    g = load_golden(name)
    B, S_, N = int(g["meta_B"]), int(g["meta_S"]), int(g["meta_N"])
    draws = VR.RecordedDraws([g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"]])
    o, d, z, pitch, yaw = VR.sample_rays(B, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi * 0.5, np.pi * 0.5, "gaussian", draws=draws)
    assert not draws.arrays and o.is_cuda
    np.testing.assert_allclose(N_(o), g["st_origins"], atol=3e-7)
    np.testing.assert_allclose(N_(d), g["st_dirs"], atol=3e-7)
    np.testing.assert_allclose(N_(z), g["st_z_coarse"][..., 0], atol=2e-7)
    np.testing.assert_allclose(N_(torch.cat([pitch, yaw], -1)), g["poses"], atol=2e-7)
    # the PyTorch statement of the same math (what CPU tests pin) agrees with the kernel
    draws = VR.RecordedDraws([g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"]])
    o2, d2, z2, p2, y2 = VR.sample_rays(B, N, "cpu", 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi * 0.5, np.pi * 0.5, "gaussian", draws=draws)
    np.testing.assert_allclose(N_(d), d2.numpy(), atol=3e-7)
    np.testing.assert_allclose(N_(z), z2.numpy(), atol=2e-7)
    # extreme pitch is clamped like the reference (:220)
    u = torch.rand((1, 16, 3, 1), device=DEV)
    o3, d3, z3, p3, y3 = native.ray_setup(1, 4, 3, -9.5, 0.88, 1.12, u, torch.tensor([0.3], device=DEV), torch.tensor([-0.2], device=DEV))
    assert abs(float(p3) - 1e-5) < 1e-9 and np.isfinite(N_(d3)).all()


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: staged_forward 256x256, 48+48 samples (96 composited samples per ray = 2 per lane), and
# configs[0]: 64x64, 12 coarse samples, no hierarchical sampling
# ---------------------------------------------------------------------------------------------------
def test_config5_256_48p48_and_config1_64_12():
    spec, sd = _full_weights()
    nat = native.NativeModel(sd, spec, DEV, "f16x3")
    film = proc.film_params(spec, 1, seed=0)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    for (S_, N, hier) in ((256, 48, True), (64, 12, False)):
        R = S_ * S_
        torch.manual_seed(3)
        o, d, z, _, _ = VR.sample_rays(1, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
        u = torch.rand((R, N), device=DEV) if hier else None
        opts = _lib.composite_opts("relu", fill_mode="seg_padding_background", fill_color="black")
        rgb, depth, w, ws = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=hier, want_weights=True, want_wsum=True)
        rgb, depth, w, ws = N_(rgb), N_(depth), N_(w), N_(ws)
        M = 2 * N if hier else N
        assert rgb.shape == (1, R, 22) and w.shape == (1, R, M) and np.isfinite(rgb).all()
        np.testing.assert_allclose(w.sum(-1), ws, atol=2e-5)
        assert ((ws < 0.9) == (rgb[..., 0] == 1)).all()
        # oracle: EVERY ray of both images (round 4; the 256x256 one is 6.3 M points: ~80 s of numpy on the GPU box's host cores, in
        # slabs of 2,048 rays)
        idx = np.arange(R)
        ti = torch.as_tensor(idx, device=DEV)
        o_i, d_i, z_i, u_i = o[:, ti].contiguous(), d[:, ti].contiguous(), z[:, ti].contiguous(), (u[ti].contiguous() if hier else None)
        r_rgb, r_depth, r_z = _oracle_render_rays(sd, spec, args, N_(o_i), N_(d_i), N_(z_i), N_(u_i) if hier else None, "black", hier=hier)
        # measured (round 4): 256^2 x 48+48: max 7.37e-4, 0 rays > 1e-3, depth 5.2e-4; 64^2 x 12 coarse: max 1.67e-6, depth 2.1e-5
        bounds = dict(max_over=0, over_bound=1e-3, depth_bound=8e-4) if hier else dict(max_over=0, over_bound=3e-6, depth_bound=3.2e-5)
        _check_render_vs_oracle(f"{S_}x{S_} {N}{'+' + str(N) if hier else ''} H=256 f16x3 vs oracle on {len(idx)} rays", nat, (o_i, d_i, z_i, u_i), tf,
                                opts, rgb[:, idx], depth[:, idx], r_rgb, r_depth, r_z, hier=hier, **bounds)


def test_spatial_siren_grid_vs_reference():
    """SPATIALSIRENGRID vs the reference module's own forward(input, z, ray_directions) on the same weights: the
    latent-grid generator (PyTorch), then ONE native launch that evaluates the per-point mapping network and the FiLM-SIREN
    (fenerf_siren_forward_local; 70 points per image: ragged tiles).  Also the explicit per-point FiLM entry
    (fenerf_siren_forward_pointwise), on F32 and on F16X3 model handles."""
    from test_host_cpu import _spatial_grid_module
    g = load_golden("tiny_spatial_grid")
    H = int(g["meta_H"])
    mod = _spatial_grid_module(g).to(DEV).eval()
    mod.device = torch.device(DEV)
    with torch.no_grad():
        full = mod(T(g["points"]), T(g["z"]), T(g["dirs"]))                                             # the reference's forward, from z
        out = mod.forward_with_latent_grid(T(g["points"]), T(g["latent_grid"]), T(g["dirs"]))            # teacher-forced latent grid
        out2 = mod.forward_with_frequencies_phase_shifts(T(g["local_coords"]), T(g["freq"]), T(g["phase"]), T(g["dirs"]))
    assert full.shape == out.shape == g["out"].shape and out.is_cuda
    rel = lambda a: (np.abs(N_(a)[..., :3] - g["out"][..., :3]).max(), np.abs(N_(a)[..., 3] - g["out"][..., 3]).max() / max(1.0, np.abs(g["out"][..., 3]).max()))
    (e_rgb0, e_sig0), (e_rgb, e_sig), e2 = rel(full), rel(out), np.abs(N_(out2) - g["out"]).max()
    print(f"[parity] SPATIALSIRENGRID vs the reference: forward(z) rgb {e_rgb0:.2e} sigma rel {e_sig0:.2e}; given the latent grid rgb {e_rgb:.2e} "
          f"sigma rel {e_sig:.2e} (one launch: mapping network + SIREN); explicit per-point FiLM {e2:.2e}")
    assert e_rgb <= 5e-6 and e_sig <= 2e-5 and e_rgb0 <= 2e-5 and e_sig0 <= 5e-5 and e2 <= 1e-4
    # a [B, 9H] FiLM block (one per image) still takes the ordinary path and equals broadcasting it to every point
    f1, p1 = T(g["freq"][:, 0]), T(g["phase"][:, 0])
    with torch.no_grad():
        a = mod.forward_with_frequencies_phase_shifts(T(g["local_coords"]), f1, p1, T(g["dirs"]))
        b = mod.forward_with_frequencies_phase_shifts(T(g["local_coords"]), f1[:, None].expand(-1, 70, -1).contiguous(),
                                                      p1[:, None].expand(-1, 70, -1).contiguous(), T(g["dirs"]))
    assert torch.equal(a, b)
    # an F16X3 handle evaluates explicit per-point parameters on its resident exact-fp32 stream: same values as an F32 handle
    spec = mod._spec()
    sd = mod._state_numpy()
    args = (T(g["local_coords"]), T(g["dirs"]), T(g["freq"][..., :8 * H]), T(g["phase"][..., :8 * H]), T(g["freq"][..., -H:]), T(g["phase"][..., -H:]))
    r32 = native.NativeModel(sd, spec, DEV, "f32").siren_forward_pointwise(*args)
    n16 = native.NativeModel(sd, spec, DEV, "f16x3")
    assert torch.equal(n16.siren_forward_pointwise(*args), r32) and np.abs(N_(r32) - g["out"]).max() <= 1e-4
    n16.load_from_device({k: T(v) for k, v in sd.items()})         # a device-side re-pack does not refresh the fp32 stream: refused, loudly
    with pytest.raises(_lib.FenerfError, match="fenerf_model_update"):
        n16.siren_forward_pointwise(*args)
    n16.update(sd)
    assert torch.equal(n16.siren_forward_pointwise(*args), r32)


def test_spatial_siren_grid_at_h256_one_launch_vs_explicit_film():
    """The same at the paper's width (H = 256, 9 FiLM layers, 4,608 modulation values per point) and 20,000 points: the fused launch
    (local latents in, 152 B per point) against torch mapping network + fenerf_siren_forward_pointwise (18 KB of FiLM parameters per
    point through HBM) -- two independent routes to the same numbers."""
    torch.manual_seed(3)
    mod = S.SPATIALSIRENGRID(input_dim=3, z_dim=16, hidden_dim=256, output_dim=4).to(DEV).eval()
    mod.device = torch.device(DEV)
    B, P = 2, 10000
    g_ = torch.Generator(device=DEV).manual_seed(4)
    pts = (torch.rand((B, P, 3), device=DEV, generator=g_) - 0.5) * 0.24
    dirs = torch.nn.functional.normalize(torch.randn((B, P, 3), device=DEV, generator=g_), dim=-1)
    lat = torch.randn((B, 32, 32, 32), device=DEV, generator=g_)
    with torch.no_grad():
        fused = mod.forward_with_latent_grid(pts, lat, dirs)
        sampled = mod.sample_local_latents(lat, mod.gridwarper(pts))
        f, p = mod.mapping_network(sampled)
        local = mod.get_local_coordinates(pts, 32, preserve_y=False)
        explicit = mod.forward_with_frequencies_phase_shifts(local, f, p, dirs)
    e = (fused - explicit).abs()
    nat = mod.native_local(DEV)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    big = (torch.rand((1, 196608, 3), device=DEV, generator=g_) - 0.5) * 0.24
    lat_b = mod.sample_local_latents(lat[:1], mod.gridwarper(big))
    nat.forward(big, None, lat_b)
    ev[0].record()
    for _ in range(3):
        nat.forward(big, None, lat_b)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    print(f"[parity] SPATIALSIRENGRID H=256, {B * P} points: fused launch vs torch mapping network + explicit per-point FiLM: rgb {e[..., :3].max().item():.2e}, "
          f"sigma {e[..., 3].max().item():.2e} (|sigma| max {explicit[..., 3].abs().max().item():.2f}); fused launch on 196,608 points: {ms:.2f} ms = "
          f"{196608 * 3.56e6 / ms / 1e9:.0f} TFLOP/s of exact-fp32 MFMA work (3.56 MFLOP per point: 2.5 mapping network + 1.06 SIREN)")
    assert e[..., :3].max().item() <= 2e-5 and e[..., 3].max().item() <= 2e-5 * max(1.0, explicit[..., 3].abs().max().item())


def _curriculum_generator(precision="f16x3"):
    """The generator BASELINE.json names (curriculum CelebA_double_semantic_texture_embedding_256_dim_96: H=256 + 32x96^3 grid,
    two 256-d latents), random init like the reference's `generator = getattr(generators, ...)(...)` (train...py:150-160)."""
    from fenerf_amd import curriculums
    cur = curriculums.CelebA_double_semantic_texture_embedding_256_dim_96
    torch.manual_seed(11)
    gen = G.DoubleImplicitGenerator3d(getattr(S, cur["model"]), 256, 256, 22).to(DEV)
    gen.set_device(torch.device(DEV))
    gen.siren.precision = precision
    return gen, cur, curriculums


def test_one_model_handle_from_two_threads_and_streams():
    """include/fenerf.h: "a FenerfModel is immutable after create/update and may be used from several threads/streams".  Two
    host threads, each on its own HIP stream with its own workspaces, call siren_forward / render on the SAME FenerfModel
    concurrently; every result must equal the single-threaded one bit for bit (the per-device dynamic-LDS opt-in is taken
    under a lock, fenerf::ensure_dynamic_lds)."""
    import threading
    spec = proc.model_spec("texture", hidden_dim=64, grid_size=6, z_dim=8)
    sd = proc.make_state_dict(spec, seed=4, sigma_gain=50.0, with_mapping=False)
    film = proc.film_params(spec, 1, seed=4)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    S_, N = 24, 10
    torch.manual_seed(1)
    o, d, z, _, _ = VR.sample_rays(1, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((S_ * S_, N), device=DEV)
    opts = _lib.composite_opts("relu")
    for precision in PRECISIONS:
        nat = native.NativeModel(sd, spec, DEV, precision)        # fresh handle: the threads race for the first launch
        results, errors = {}, []

        def worker(tid):
            try:
                view = native.NativeModel.__new__(native.NativeModel)      # same FenerfModel*, private workspaces
                view.__dict__.update(nat.__dict__)
                view._ws = {}
                st = torch.cuda.Stream(device=DEV)
                outs = []
                with torch.cuda.stream(st):
                    for _ in range(6):
                        outs.append(view.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)[0])
                st.synchronize()
                results[tid] = outs
                view._h = None                                              # the handle belongs to `nat`
            except Exception as e:      # noqa: BLE001
                errors.append(e)

        torch.cuda.synchronize()
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        ref = nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)[0]
        torch.cuda.synchronize()
        for tid in (0, 1):
            assert all(torch.equal(r, ref) for r in results[tid]), (precision, tid)


@pytest.mark.parametrize("case", ["tiny_texture_fill_noise", "baseline_lock_view", "h256_bench_shape", "h256_48p48", "cannot_fuse"])
def test_one_launch_render_equals_the_four_launch_render(case):
    """fenerf_render_forward on an f16x3 model runs generators.py:479-519 as ONE launch when the shape allows it (ray groups of whole
    128-point tile groups: coarse tiles -> weights + resampling -> fine tiles -> merge + composite inside the workgroup;
    fenerf_siren_f16w.hip FUSED, include/fenerf.h fenerf_set_render_fusion).  Both routes run the same per-tile and per-ray code, so
    every output -- pixels, depth, weights, weights_sum -- must be IDENTICAL bit for bit; the launch-group record says which route ran."""
    cases = {
        # kind, H, grid, B, S, N, composite kwargs, noise, lock_view
        "tiny_texture_fill_noise": ("texture", 32, 8, 2, 8, 6, dict(fill_mode="seg_padding_background", fill_color="black", noise_std=0.3), True, False),
        "baseline_lock_view": ("baseline", 64, 0, 1, 24, 10, dict(white_back=True), False, True),
        "h256_bench_shape": ("texture", 256, 96, 1, 128, 24, dict(fill_mode="seg_padding_background", fill_color="black"), False, False),
        "h256_48p48": ("texture", 256, 96, 1, 64, 48, dict(last_back=True), False, False),
        "cannot_fuse": ("texture", 32, 8, 1, 10, 6, {}, False, False),          # 100 rays: not a multiple of the 64-ray group of N = 6
    }
    kind, H, grid, B, S_, N, okw, with_noise, lock = cases[case]
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8) if grid else proc.model_spec(kind, hidden_dim=H, z_dim=8)
    sd = proc.make_state_dict(spec, seed=6, sigma_gain=60.0, with_mapping=False)
    film = proc.film_params(spec, B, seed=6)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    R = S_ * S_
    torch.manual_seed(12)
    o, d, z, _, _ = VR.sample_rays(B, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * R, N), device=DEV)
    nc = torch.randn((B * R, N), device=DEV) if with_noise else None
    nf = torch.randn((B * R, 2 * N), device=DEV) if with_noise else None
    opts = _lib.composite_opts("relu", okw.pop("noise_std", 0.0), **okw)
    nat = native.NativeModel(sd, spec, DEV, "f16x3")
    outs, routes = {}, {}
    for mode in ("off", "force", "auto"):
        with native.render_fusion(mode), native.phase_timing() as t:
            outs[mode] = nat.render(o, d, z, u, nc, nf, *tf, opts, hierarchical=True, lock_view=lock, want_weights=True, want_wsum=True)
        routes[mode] = dict(t.calls)
    assert routes["off"].get("render_fused", 0) == 0 and routes["off"]["siren_forward"] == 2 and routes["off"]["composite"] == 2, routes
    if case == "cannot_fuse":
        assert routes["force"] == routes["off"] == routes["auto"], routes
    else:
        assert routes["force"] == {"render_fused": 1}, routes
        # auto: one launch when whole ray groups per workgroup do not lengthen the critical path (the bench shape and 48 + 48) AND the
        # library was built with the one-launch route enabled for AUTO (it is only where it has been measured to be no slower)
        assert routes["auto"] in ((({"render_fused": 1},) if case.startswith("h256") else ()) + (routes["off"],)), routes
    for mode in ("force", "auto"):
        for name, a, b_ in zip(("pixels", "depth", "weights", "weights_sum"), outs["off"], outs[mode]):
            assert torch.equal(a, b_), (case, mode, name, float((a - b_).abs().max()))
    # ... and an f32 model always takes the four launches
    nat32 = native.NativeModel(sd, spec, DEV, "f32")
    with native.render_fusion("force"), native.phase_timing() as t:
        nat32.render(o, d, z, u, nc, nf, *tf, opts, hierarchical=True, lock_view=lock)
    assert "render_fused" not in t.calls
    print(f"[parity] one-launch render == four-launch render bit for bit [{case}]: B {B}, {S_}x{S_} rays, {N}+{N} samples; routes {routes}")


def test_staged_forward_generator_call_at_configs4_256_48p48():
    """BASELINE.json configs[4] through the public method (generators.py:546-646), not only nat.render: 256x256 rays, 48+48
    samples, psi 0.7, the whole image as one fused render (max_batch_size ignored by design)."""
    gen, cur, curriculums = _curriculum_generator()
    gen.eval()
    md = {**curriculums.extract_metadata(cur, 60000), "nerf_noise": 0, "psi": 0.7, "img_size": 256, "num_steps": 48,
          "max_batch_size": 2400000, "lock_view_dependence": True, "h_stddev": 0, "v_stddev": 0}
    zg, za = torch.randn(1, 256, device=DEV), torch.randn(1, 256, device=DEV)
    outs = []
    for _ in range(2):
        torch.manual_seed(5)
        torch.cuda.reset_peak_memory_stats()
        with torch.no_grad():
            outs.append(gen.staged_forward(zg, za, **md))
    (px, depth), (px2, depth2) = outs
    assert tuple(px.shape) == (1, 22, 256, 256) and tuple(depth.shape) == (1, 256, 256) and not px.is_cuda
    assert torch.equal(px, px2) and torch.equal(depth, depth2), "same seed, same image (bit-exact)"
    px, depth = px.numpy(), depth.numpy()
    assert np.isfinite(px).all() and np.isfinite(depth).all()
    assert (px[:, -3:] >= -1 - 1e-5).all() and (px[:, -3:] <= 1 + 1e-5).all()   # '*2-1' epilogue: rgb in [-1, 1] (labels are logits)
    assert ((depth >= 0) & (depth <= 1.12 * 1.001 + 0.02)).all()
    filled = px[:, 0] == 1.0                                                     # seg_padding_background: channel 0 = 2*1-1
    print(f"[parity] staged_forward 256x256 48+48 through the generator call: {int(filled.sum())} background pixels, "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GB")
    # the second public inference method on the same latents: staged_forward_with_frequencies fed with the truncated FiLM
    # parameters staged_forward builds (generators.py:558-564) consumes the same six draws and must give the same image
    torch.manual_seed(5)
    gen.generate_avg_frequencies()                      # the two randn(10000, z) draws staged_forward makes first
    with torch.no_grad():
        rfg, rpg = gen.siren.geo_mapping_network(zg)
        rfa, rpa = gen.siren.app_mapping_network(za)
    trunc = lambda avg, raw: avg + 0.7 * (raw - avg)
    px3, depth3, third = gen.staged_forward_with_frequencies(
        trunc(gen.avg_frequencies_geo, rfg), trunc(gen.avg_frequencies_app, rfa), trunc(gen.avg_phase_shifts_geo, rpg),
        trunc(gen.avg_phase_shifts_app, rpa), **md)
    assert torch.equal(px3, outs[0][0]) and torch.equal(depth3, outs[0][1])
    assert tuple(third.shape) == (1, 96, 256, 256)      # seg_padding_background: per-sample weights


def test_generator_step_at_configs2_shape_B6_128_24p24():
    """BASELINE.json configs[2]: the reference's G-step micro-batch (batch 24 / batch_split 4 = 6 images, 128x128, 24+24,
    train_double_latent_semantic.py:402-446) through forward(z) + backward on the native differentiable path: finite gradients on
    every generator parameter, peak memory reported."""
    gen, cur, curriculums = _curriculum_generator()
    gen.train()
    md = {**curriculums.extract_metadata(cur, 60000), "img_size": 128, "num_steps": 24,
          "nerf_noise": max(0, 1.0 - 2500 / 5000.0)}         # the training loop sets it per step (train...py:276)
    B = 6
    zg, za = torch.randn(B, 256, device=DEV), torch.randn(B, 256, device=DEV)
    torch.cuda.reset_peak_memory_stats()
    px, poses = gen(zg, za, **md)
    assert tuple(px.shape) == (B, 21, 128, 128) and tuple(poses.shape) == (B, 2)
    w = torch.randn_like(px)
    (px * w).mean().backward()
    torch.cuda.synchronize()
    n, worst = 0, 0.0
    for name, p in gen.named_parameters():
        assert p.grad is not None, name
        assert torch.isfinite(p.grad).all(), name
        n += 1
        worst = max(worst, float(p.grad.abs().max()))
    assert worst > 0
    print(f"[parity] G-step at configs[2] micro-batch (6 x 128x128 x 24+24 = {B * 128 * 128 * 48} points): {n} parameter gradients "
          f"finite, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB")


class _PerImageDraws(VR.TorchDraws):
    """Every draw is generated image by image from a seed of (call number, global image id): image b of a batch sees exactly the draws
    it sees when rendered alone (`images` = the global ids of the batch being rendered)."""

    def __init__(self):
        self.images, self.calls = [0], 0

    def _blocks(self, shape, device, fn):
        self.calls += 1
        n = len(self.images)
        assert shape[0] % n == 0, (shape, n)
        out = []
        for b in self.images:
            g = torch.Generator(device=device).manual_seed(1000003 * self.calls + b)
            out.append(fn((shape[0] // n,) + tuple(shape[1:]), g))
        return torch.cat(out, 0)

    def rand(self, shape, device):
        return self._blocks(tuple(shape), device, lambda sh, g: torch.rand(sh, device=device, generator=g))

    def randn(self, shape, device):
        return self._blocks(tuple(shape), device, lambda sh, g: torch.randn(sh, device=device, generator=g))

    def normal_candidates(self, shape, device):
        return self.randn(shape, device)

    def coin(self):
        return 0.25


def test_generator_step_at_configs2_is_the_sum_of_its_six_images():
    """A size-independent property that pins VALUES at configs[2]'s full size: the micro-batch step (6 images in one forward_save /
    chain / weight-gradient pass each, 6 backward chunks of 786,432 points) must equal the six single-image steps -- pixels of image b bit for bit (the
    arithmetic of a sample does not depend on which workgroup evaluates it), every parameter gradient the sum of the six (fp32 sums in a
    different grouping: <= 5e-6 of the tensor's largest entry)."""
    gen, cur, curriculums = _curriculum_generator()
    gen.train()
    md = {**curriculums.extract_metadata(cur, 60000), "img_size": 128, "num_steps": 24, "nerf_noise": 0.5}
    B = 6
    g0 = torch.Generator(device=DEV).manual_seed(5)
    zg, za = torch.randn(B, 256, device=DEV, generator=g0), torch.randn(B, 256, device=DEV, generator=g0)
    w = torch.randn((B, 21, 128, 128), device=DEV, generator=g0)
    with torch.no_grad():       # the mapping networks are torch GEMMs (batch-size dependent rounding): one evaluation, sliced per image
        fg, pg = gen.siren.geo_mapping_network(zg)
        fa, pa = gen.siren.app_mapping_network(za)
    draws = _PerImageDraws()
    gen.draws = draws
    params = {k: p for k, p in gen.named_parameters() if "mapping_network" not in k}

    def step(ids):
        for p in params.values():
            p.grad = None
        draws.images, draws.calls = list(ids), 0
        film = [t[ids].clone().requires_grad_(True) for t in (fg, fa, pg, pa)]
        px, _ = gen.forward_with_frequencies(*film, **md)
        (px * w[ids]).sum().backward()
        grads = {k: p.grad.detach().clone() for k, p in params.items()}
        for name, t in zip(("film.freq_geo", "film.freq_app", "film.phase_geo", "film.phase_app"), film):
            grads[name] = t.grad.detach().clone()
        return px.detach().clone(), grads

    px6, g6 = step(list(range(B)))
    calls6 = draws.calls
    acc = None
    for b in range(B):
        px1, g1 = step([b])
        assert draws.calls == calls6
        assert torch.equal(px1[0], px6[b]), f"image {b}: pixels differ between the batch render and the single render"
        for k in [k for k in g1 if k.startswith("film.")]:      # per-image gradients: row b of the batch's
            full = torch.zeros_like(g6[k])
            full[b] = g1[k][0]
            g1[k] = full
        acc = g1 if acc is None else {k: acc[k] + g1[k] for k in acc}
    worst, wk = 0.0, None
    for k in g6:
        scale = float(g6[k].abs().max())
        assert scale > 0, k
        e = float((g6[k] - acc[k]).abs().max()) / scale
        if e > worst:
            worst, wk = e, k
    print(f"[parity] configs[2] micro-batch (6 x 128x128 x 24+24) = sum of its six single-image steps: pixels bit-identical per image, "
          f"worst gradient difference over {len(g6)} tensors {worst:.2e} ({wk})")
    assert worst <= 5e-6


# ---------------------------------------------------------------------------------------------------
# single-latent pi-GAN generator (ImplicitGenerator3d + SPATIALSIRENBASELINE, curriculum `CelebA`) end to end
# ---------------------------------------------------------------------------------------------------
def _make_spatial_generator(g, spec, precision):
    gen = G.ImplicitGenerator3d(functools.partial(S.SPATIALSIRENBASELINE, hidden_dim=spec["hidden_dim"]), spec["z_dim"], 4)
    sd = proc.make_state_dict(spec, seed=int(g["meta_seed"]), sigma_gain=float(g["meta_sigma_gain"]))
    gen.siren.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    gen = gen.to(DEV).eval()
    gen.siren.precision = precision
    gen.device = torch.device(DEV)
    gen.siren.device = gen.device
    return gen


@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
def test_single_latent_generator_vs_reference(precision):
    g = load_golden("tiny_spatial_fwd")
    spec = spec_from_golden(g)
    gen = _make_spatial_generator(g, spec, precision)
    f = proc.film_params(spec, int(g["meta_B"]), seed=int(g["meta_seed"]))
    freq = T(np.concatenate([f["freq_geo"], f["freq_app"]], -1))
    phase = T(np.concatenate([f["phase_geo"], f["phase_app"]], -1))
    common = dict(img_size=int(g["meta_S"]), fov=12, ray_start=0.88, ray_end=1.12, num_steps=int(g["meta_N"]), h_stddev=0.3,
                  v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, hierarchical_sample=True, sample_dist="gaussian")
    seq = [g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"], g["rand_u_fine"], g["rand_noise_fine"]]
    gen.draws = VR.RecordedDraws(seq)
    with torch.no_grad():
        px, poses = gen.forward_with_frequencies(freq, phase, **common, **kwargs_from_golden(g))
    assert px.shape == g["pixels"].shape == (2, 3, 8, 8)
    err = np.abs(N_(px) - g["pixels"]).max(axis=1)
    print(f"[parity] single-latent forward_with_frequencies[{precision}]: max|err| {err.max():.3e}")
    assert err.max() <= 1e-3
    np.testing.assert_allclose(N_(poses), g["poses"], atol=1e-6)
    # staged variant, eval_white_back fill (3-channel model), returns (pixels on device, depth.cpu()) -- two values
    g2 = load_golden("tiny_spatial_staged")
    gen.draws = VR.RecordedDraws([g2["rand_u_jitter"], g2["rand_r_theta"], g2["rand_r_phi"], g2["rand_noise_coarse"], g2["rand_u_fine"],
                                  g2["rand_noise_fine"]])
    res = gen.staged_forward_with_frequencies(freq, phase, **common, **kwargs_from_golden(g2))
    assert len(res) == 2 and res[0].is_cuda and not res[1].is_cuda
    err = np.abs(N_(res[0]) - g2["pixels"]).max(axis=1)
    bad = err > 1e-3
    print(f"[parity] single-latent staged_forward_with_frequencies[{precision}]: max|err| {err.max():.3e}, {int(bad.sum())} flips")
    assert int(bad.sum()) == 0
    np.testing.assert_allclose(N_(res[1])[~bad], g2["depth"][~bad], atol=1e-4)


# ---------------------------------------------------------------------------------------------------
# callers: multi-view render with mask2color, voxel-grid density evaluation
# ---------------------------------------------------------------------------------------------------
def test_callers_multiview_and_voxel_grid():
    from fenerf_amd import callers, curriculums
    g = load_golden("tiny_texture_z_full")
    spec = spec_from_golden(g)
    gen = _make_generator(dict(meta_seed=3, meta_sigma_gain=300.0), spec)
    cur = dict(curriculums.CelebA_double_semantic_texture_embedding_256_dim_96)
    imgs, segs = callers.render_multiview(gen, cur, seed=0, device=DEV, image_size=8, ray_step_multiplier=1, z_dim=16,
                                          face_angles=(-0.5, 0.0, 0.5))
    assert imgs.shape == (3, 3, 8, 8) and segs.shape == (3, 3, 8, 8) and not imgs.is_cuda
    assert float(imgs.min()) >= -1 - 1e-5 and float(imgs.max()) <= 1 + 1e-5 and float(segs.min()) >= 0 and float(segs.max()) <= 1
    assert not torch.equal(imgs[0], imgs[2]), "different yaw angles give different views"
    # voxel grid: sigma of the fused kernel == oracle on the same truncated film parameters
    N = 6
    torch.manual_seed(5)
    z = torch.randn((1, 16), device=DEV)
    torch.manual_seed(9)
    vol = callers.sample_generator(gen, z, voxel_resolution=N, cube_length=0.24, psi=0.5)
    assert vol.shape == (N, N, N)
    torch.manual_seed(9)
    avg = gen.generate_avg_frequencies()
    with torch.no_grad():
        rfg, rpg = gen.siren.geo_mapping_network(z)
        rfa, rpa = gen.siren.app_mapping_network(z)
    film = [N_(a + 0.5 * (r - a)) for a, r in zip(avg, (rfg, rpg, rfa, rpa))]
    sd = proc.make_state_dict(dict(spec, z_dim=16, map_hidden=256), seed=3, sigma_gain=300.0)
    samples, _, _ = callers.create_samples(N, (0, 0, 0), 0.24)
    lock = np.zeros((1, N ** 3, 3), np.float32)
    lock[..., -1] = -1
    ref = O.siren_forward(sd, spec, samples.numpy(), lock, film[0], film[1], film[2], film[3])
    np.testing.assert_allclose(vol.reshape(-1), ref[0, :, -1], atol=3e-3, rtol=2e-4)
    # latent interpolation: end frames are the two identities, middle frames differ, 'geo' keeps the appearance side fixed
    md = {**callers.multiview_kwargs(cur, image_size=8, ray_step_multiplier=1), "fill_mode": None, "sample_dist": None}
    z1, z2 = torch.randn((1, 16), device=DEV), torch.randn((1, 16), device=DEV)
    torch.manual_seed(1)
    frames, depth = callers.render_latent_interpolation(gen, z1, z2, z1, z2, md, n_frames=3, latent_type="both", psi=0.7)
    assert frames.shape == (3, 21, 8, 8) and depth.shape == (3, 8, 8) and not frames.is_cuda
    torch.manual_seed(1)
    gen.generate_avg_frequencies()
    with torch.no_grad():
        fg, pg = gen.siren.geo_mapping_network(z1)
        fa, pa = gen.siren.app_mapping_network(z1)
        tr = lambda a, r: a + 0.7 * (r - a)
        first, _, _ = gen.staged_forward_with_frequencies(tr(gen.avg_frequencies_geo, fg), tr(gen.avg_frequencies_app, fa),
                                                          tr(gen.avg_phase_shifts_geo, pg), tr(gen.avg_phase_shifts_app, pa), **md)
    assert torch.allclose(frames[0], first[0], atol=1e-5)
    assert not torch.allclose(frames[0], frames[1]) and not torch.allclose(frames[1], frames[2])


def _tiny_checkpoint_dir(tmp_path):
    """<tmp>/7000_generator.pth (the reference's pickled tiny generator) + <tmp>/7000_ema.pth (a torch_ema-layout pickle holding
    the same weights), i.e. what the reference's training loop leaves in its output directory (train...py:251, :524-526)."""
    import shutil
    from conftest import GOLDEN
    from fenerf_amd import compat, ema as ema_mod
    compat.install_aliases()
    path = str(tmp_path / "7000_generator.pth")
    shutil.copy(os.path.join(GOLDEN, "ref_generator_tiny.pth"), path)
    gen = torch.load(path, weights_only=False)
    torch.save(ema_mod.ExponentialMovingAverage(gen.parameters(), decay=0.999), str(tmp_path / "7000_ema.pth"))
    return path


def test_render_multiview_vs_reference_generate_img(tmp_path):
    """callers.load_generator + callers.render_multiview == the reference's generate_img over its five yaw angles
    (render_multiview_images_double_semantic.py:24-29, :66-83) on the same pickled generator, latents and draws."""
    import json
    g = load_golden("tiny_multiview")
    gen = callers_mod().load_generator(_tiny_checkpoint_dir(tmp_path), DEV)
    assert gen.softmax_label is False and not gen.training
    cur = {(int(k[4:]) if k.startswith("int:") else k): v for k, v in json.loads(str(g["curriculum_json"])).items()}
    six = [g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"], g["rand_u_fine"], g["rand_noise_fine"]]
    gen.draws = VR.RecordedDraws(([np.zeros((10000, 16), np.float32)] * 2 + six) * 5)
    orig = gen.generate_avg_frequencies

    def patched():
        orig()
        gen.avg_frequencies_geo, gen.avg_phase_shifts_geo = T(g["avg_freq_geo"]), T(g["avg_phase_geo"])
        gen.avg_frequencies_app, gen.avg_phase_shifts_app = T(g["avg_freq_app"]), T(g["avg_phase_app"])
        return gen.avg_frequencies_geo, gen.avg_phase_shifts_geo, gen.avg_frequencies_app, gen.avg_phase_shifts_app
    gen.generate_avg_frequencies = patched
    images, segmaps = callers_mod().render_multiview(gen, cur, int(g["seed"]), DEV, image_size=int(g["image_size"]),
                                                     ray_step_multiplier=int(g["ray_step_multiplier"]), lock_view_dependence=True,
                                                     latents=(g["z_geo"], g["z_app"]))
    assert not gen.draws.arrays, "all recorded draws consumed, in order"
    assert tuple(images.shape) == g["images"].shape and tuple(segmaps.shape) == g["segmaps"].shape
    err = np.abs(N_(images) - g["images"]).max()
    same = (N_(segmaps) == g["segmaps"]).all(axis=1).mean()
    print(f"[parity] render_multiview vs the reference's generate_img: images max|err| {err:.2e}, colour maps identical on "
          f"{same * 100:.1f} % of pixels")
    assert err <= 1e-3 and same == 1.0


def callers_mod():
    from fenerf_amd import callers
    return callers


def test_cli_front_ends_write_what_the_reference_scripts_write(tmp_path):
    """tools/render_multiview.py and tools/render_video_interpolation.py as commands (the reference's argparse surfaces) on a
    checkpoint directory laid out like the reference's: grids / frames / strips / video exist with the expected geometry."""
    import json
    import subprocess
    import sys
    from PIL import Image
    from conftest import ROOT
    g = load_golden("tiny_multiview")
    ckpt = _tiny_checkpoint_dir(tmp_path)
    cur = json.loads(str(g["curriculum_json"]))
    cur.update(output_dim=22, eval_last_back=False)
    cur_file = str(tmp_path / "tiny_curriculum.json")
    json.dump(cur, open(cur_file, "w"))
    out = str(tmp_path / "imgs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "render_multiview.py"), ckpt, "--curriculum", cur_file, "--seeds", "3", "4",
                        "--output_dir", out, "--image_size", "8", "--ray_step_multiplier", "2", "--lock_view_dependence"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for seed in (3, 4):
        for kind in ("RGB", "SEG"):
            im = np.asarray(Image.open(os.path.join(out, f"grid_{seed}_{kind}.png")))
            assert im.shape == (8 + 4, 5 * 10 + 2, 3)                                  # five 8x8 views, padding 2
    assert not np.array_equal(np.asarray(Image.open(os.path.join(out, "grid_3_RGB.png"))), np.asarray(Image.open(os.path.join(out, "grid_4_RGB.png"))))
    vids = str(tmp_path / "vids")
    base = [sys.executable, os.path.join(ROOT, "tools", "render_video_interpolation.py"), ckpt, "--curriculum", cur_file, "--seeds", "3",
            "--output_dir", vids, "--image_size", "8", "--ray_step_multiplier", "1", "--num_frames", "3", "--trajectory", "front",
            "--latent_type", "both", "--psi", "0.7"]
    r = subprocess.run(base, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = os.path.join(vids, "interpolation_both_3")
    for name in ("interp.png", "interp_seg.png", "interp_acc_map.png", "interp_depth_map.png"):
        assert np.asarray(Image.open(os.path.join(d, name))).shape[:2] == (8 + 4, 3 * 10 + 2), name
    for j in range(3):
        for kind in ("img", "label", "acc", "depth"):
            assert np.asarray(Image.open(os.path.join(d, "images", "both_front", f"{kind}_{j}.png"))).shape[:2] == (8, 8)
    r = subprocess.run(base + ["--save_with_video"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = open(os.path.join(d, "interp_both_3.avi"), "rb").read()
    assert raw[:4] == b"RIFF" and raw.count(b"00db") == 6                               # 3 frames [image | labels | blend | depth]


def test_inversion_and_shape_front_ends_as_commands(tmp_path):
    """tools/inverse_render.py and tools/extract_shapes.py as commands (the reference's argparse surfaces,
    inverse_render_double_semantic.py:132-169 / extract_double_semantic_shapes.py:90-97) on a checkpoint directory laid out like the
    reference's: previews, the eight-tensor checkpoint under the reference's keys, mious.npy, the reconstruction video; then the density
    volume of the inverted identity and of seeded identities as MRC maps, equal to the library-level callers' volumes."""
    import subprocess
    import sys
    from PIL import Image
    from conftest import ROOT
    from fenerf_amd import imageio_lite
    ckpt = _tiny_checkpoint_dir(tmp_path)
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (40, 32, 3), dtype=np.uint8)).save(str(tmp_path / "face.jpg"))
    lab = np.zeros((40, 32), np.uint8); lab[8:30, 6:26] = 1; lab[12:16, 10:14] = 4; lab[24:28, 12:20] = 12
    Image.fromarray(lab, "L").save(str(tmp_path / "face.png"))
    out = str(tmp_path / "inv")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "inverse_render.py"), "t", ckpt, "--image_path", str(tmp_path / "face.jpg"), "--seg_path",
           str(tmp_path / "face.png"), "--save_dir", out, "--image_size", "8", "--iteration", "21", "--lambda_seg", "1", "--lambda_img", "1",
           "--latent_normalize", "--no_center_crop", "--preview_size", "8", "--preview_steps", "6"]
    recon = ["--recon", "--trajectory", "rotation_linear", "--num_frames", "3", "--fill_color", "white"]
    r = subprocess.run(cmd + recon, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    for angle in (-0.5, 0, 0.5):
        for kind in ("img", "seg"):
            assert np.asarray(Image.open(os.path.join(out, f"0_{angle}_{kind}.jpg"))).shape == (8, 8, 3)
    assert not os.path.exists(os.path.join(out, "20_0_img.jpg"))                       # previews every 200 iterations, mIoU every 20
    mious = np.load(os.path.join(out, "mious.npy"))
    assert mious.shape == (2,) and ((0 <= mious) & (mious <= 1)).all()
    meta = torch.load(os.path.join(out, "freq_phase_offset_t.pth"), weights_only=False)
    assert sorted(meta) == sorted(["w_geo_frequencies", "w_geo_phase_shifts", "w_geo_frequency_offsets", "w_geo_phase_shift_offsets",
                                   "w_app_frequencies", "w_app_phase_shifts", "w_app_frequency_offsets", "w_app_phase_shift_offsets"])
    assert all(float(meta[k].abs().max()) > 0 for k in meta if "offset" in k)           # 21 Adam steps moved every offset tensor
    assert all(not meta[k].requires_grad for k in meta if "offset" in k)
    raw = open(os.path.join(out, "reconstructed_debug_rotation_linear_white.avi"), "rb").read()
    assert raw[:4] == b"RIFF" and raw.count(b"00db") == 2 * 3                          # three frames [image | labels | blend]
    line = [l for l in r.stdout.splitlines() if "loss" in l][0]
    print("[parity] tools/inverse_render.py on the tiny pickled generator:", line.strip())
    # an existing --checkpoint_path is reused (no optimisation) unless --load_checkpoint
    r2 = subprocess.run(cmd + ["--checkpoint_path", os.path.join(out, "freq_phase_offset_t.pth")], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and "loss" not in r2.stdout, (r2.stdout[-500:], r2.stderr[-2000:])
    # the same optimisation with the exact-sparsity backward picked per iteration (round 6): same draws, same losses to rounding
    out3 = str(tmp_path / "inv_sparse")          # (its own directory: the checkpoint of the first run is read again below)
    r3 = subprocess.run([out3 if a == out else a for a in cmd] + ["--sparse_backward", "auto"], capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, (r3.stdout[-500:], r3.stderr[-2000:])
    line3 = [l for l in r3.stdout.splitlines() if "loss" in l][0]
    print("[parity] tools/inverse_render.py --sparse_backward auto:", line3.strip())
    import re
    nums, nums3 = ([float(x) for x in re.findall(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", l.split(" -> /")[0])] for l in (line, line3))
    assert len(nums) == len(nums3) and len(nums) >= 1 and all(abs(a - b) <= 1e-4 * max(1.0, abs(a)) for a, b in zip(nums, nums3)), (line, line3)
    # shapes: the inverted identity, then two seeded ones
    shapes = str(tmp_path / "shapes")
    base = [sys.executable, os.path.join(ROOT, "tools", "extract_shapes.py"), ckpt, "--cube_size", "0.3", "--voxel_resolution", "12", "--output_dir", shapes]
    r = subprocess.run(base + ["--latent_path", os.path.join(out, "freq_phase_offset_t.pth"), "--seeds", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    vol, h = imageio_lite.read_mrc(os.path.join(shapes, "7.mrc"))
    assert vol.shape == (12, 12, 12) and h["mode"] == 2 and np.isfinite(vol).all() and vol.std() > 0
    cm = callers_mod()
    gen = cm.load_generator(ckpt, DEV, reset_render_options=False)
    fg, fa, pg, pa = cm.film_from_inversion(meta, DEV)
    ref = cm.sample_generator_wth_frequencies_phase_shifts(gen, dict(truncated_frequencies_geo=fg, truncated_frequencies_app=fa,
                                                                     truncated_phase_shifts_geo=pg, truncated_phase_shifts_app=pa),
                                                           cube_length=0.3, voxel_resolution=12)
    np.testing.assert_array_equal(vol, ref)
    r = subprocess.run(base + ["--seeds", "3", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    v3, v4 = (imageio_lite.read_mrc(os.path.join(shapes, f"{s}.mrc"))[0] for s in (3, 4))
    assert not np.array_equal(v3, v4)
    torch.manual_seed(3)
    z = torch.randn(1, gen.z_geo_dim, device=DEV)
    np.testing.assert_array_equal(v3, cm.sample_generator(gen, z, cube_length=0.3, voxel_resolution=12))


def test_single_latent_video_interpolation_loop_and_cli(tmp_path):
    """callers.render_latent_video / --interpolation_type video_latent_interpolation (render_video_interpolation_semantic.py:187-312,
    the ImplicitGenerator3d variant): frame j is staged_forward_with_frequencies on the truncated FiLM parameters interpolated at the
    trajectory's t; the command writes the per-frame PNGs, the strip and the video."""
    import functools
    import subprocess
    import json
    from PIL import Image
    from conftest import ROOT
    from fenerf_amd import curriculums
    torch.manual_seed(0)
    gen = G.ImplicitGenerator3d(functools.partial(S.SPATIALSIRENBASELINE, hidden_dim=32), 16, 4).to(DEV)
    gen.set_device(torch.device(DEV))
    gen.eval()
    cur = {0: {"batch_size": 1, "num_steps": 4, "img_size": 8}, "fov": 12, "ray_start": 0.88, "ray_end": 1.12, "h_stddev": 0.3,
           "v_stddev": 0.155, "h_mean": np.pi / 2, "v_mean": np.pi / 2, "sample_dist": "gaussian", "clamp_mode": "relu",
           "hierarchical_sample": True, "white_back": False, "last_back": False, "z_dist": "gaussian", "fill_mode": "weight"}
    opts = callers_mod().video_kwargs(cur, image_size=8, ray_step_multiplier=1, psi=0.7, num_frames=3, fov=12)
    traj = callers_mod().camera_trajectory_single("sphere", 3, 12)
    out = callers_mod().render_latent_video(gen, 5, opts, traj, latent_type="geo", psi=0.7, device=DEV)
    assert tuple(out["images"].shape) == (3, 3, 8, 8) and tuple(out["depth"].shape) == (3, 8, 8)
    torch.manual_seed(5)
    z1, z2 = torch.randn(1, 16, device=DEV), torch.randn(1, 16, device=DEV)
    with torch.no_grad():
        af, ap = gen.generate_avg_frequencies()
        (f1, p1), (f2, p2) = gen.siren.mapping_network(z1), gen.siren.mapping_network(z2)
        tr = lambda a, r: a + 0.7 * (r - a)
        t, pitch, yaw, fov = traj[0]          # same generator state as the loop's first frame: seed, two latents, the avg-frequency pass
        kw = {k: v for k, v in opts.items() if k != "num_frames"}
        kw.update(h_mean=float(yaw), v_mean=float(pitch), fov=float(fov), h_stddev=0, v_stddev=0)
        first, _ = gen.staged_forward_with_frequencies(tr(af, f1) * (1 - t) + tr(af, f2) * t, tr(ap, p1) * (1 - t) + tr(ap, p2) * t, **kw)
    assert torch.allclose(out["images"][0], first[0], atol=1e-5)
    non = callers_mod().render_latent_video(gen, 5, opts, traj, latent_type="non", psi=0.7, device=DEV)
    assert torch.allclose(non["images"][0], out["images"][0], atol=1e-5) and not torch.allclose(non["images"][2], out["images"][2], atol=1e-4)
    # the command on a pickled single-latent generator
    from fenerf_amd import ema as ema_mod
    ckpt = str(tmp_path / "100_generator.pth")
    gen.avg_frequencies = gen.avg_phase_shifts = None
    torch.save(gen.cpu(), ckpt)
    torch.save(ema_mod.ExponentialMovingAverage(gen.parameters(), decay=0.999), str(tmp_path / "100_ema.pth"))
    cur_file = str(tmp_path / "cur.json")
    json.dump({("int:0" if k == 0 else k): v for k, v in cur.items()}, open(cur_file, "w"))
    vids = str(tmp_path / "vids")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "render_video_interpolation.py"), ckpt, "--curriculum", cur_file,
                        "--interpolation_type", "video_latent_interpolation", "--seeds", "5", "--output_dir", vids, "--image_size", "8",
                        "--ray_step_multiplier", "1", "--num_frames", "3", "--trajectory", "sphere", "--psi", "0.7", "--save_with_video"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = os.path.join(vids, "interpolation_geo_5")
    assert np.asarray(Image.open(os.path.join(d, "video_latent_interpolation_img_0.png"))).shape[:2] == (8 + 4, 3 * 10 + 2)
    for j in range(3):
        assert np.asarray(Image.open(os.path.join(d, "images", "geo_sphere", f"img_{j}.png"))).shape[:2] == (8, 8)
    raw = open(os.path.join(d, "interp_geo_5.avi"), "rb").read()
    assert raw[:4] == b"RIFF" and raw.count(b"00db") == 6


def test_reference_checkpoint_renders_like_the_reference():
    """A pickled reference generator (whole nn.Module, the reference's checkpoint format) loaded through the import aliases
    renders, on the HIP path, what the reference rendered from it."""
    import os
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from fenerf_amd import compat; compat.install_aliases()\n"
        "from fenerf_amd.generators import volumetric_rendering as VR\n"
        "g = dict(np.load(%r))\n"
        "gen = torch.load(%r, weights_only=False).to('cuda:0').eval()\n"
        "gen.device = torch.device('cuda:0'); gen.siren.device = gen.device\n"
        "T = lambda a: torch.as_tensor(a, device='cuda:0')\n"
        "rd = {k[len('stg_rand_'):]: v for k, v in g.items() if k.startswith('stg_rand_')}\n"
        "gen.draws = VR.RecordedDraws([np.zeros((10000, 16), np.float32)] * 2 + [rd['u_jitter'], rd['r_theta'], rd['r_phi'], rd['noise_coarse'], rd['u_fine'], rd['noise_fine']])\n"
        "orig = gen.generate_avg_frequencies\n"
        "def patched():\n"
        "    orig()\n"
        "    gen.avg_frequencies_geo, gen.avg_phase_shifts_geo = T(g['stg_avg_freq_geo']), T(g['stg_avg_phase_geo'])\n"
        "    gen.avg_frequencies_app, gen.avg_phase_shifts_app = T(g['stg_avg_freq_app']), T(g['stg_avg_phase_app'])\n"
        "gen.generate_avg_frequencies = patched\n"
        "kw = dict(img_size=6, num_steps=6, hierarchical_sample=True, clamp_mode='relu', nerf_noise=0.0, fov=12, ray_start=0.88, ray_end=1.12,\n"
        "          h_stddev=0.3, v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, sample_dist='gaussian')\n"
        "px, depth = gen.staged_forward(T(g['z_geo']), T(g['z_app']), psi=float(g['stg_psi']), fill_mode='seg_padding_background', fill_color='white', **kw)\n"
        "err = np.abs(px.numpy() - g['stg_pixels']).max(axis=1)\n"
        "assert err.max() <= 1e-3, err.max()\n"
        "print('ok', float(err.max()))\n") % (ROOT, os.path.join(ROOT, "tests"), os.path.join(GOLDEN, "tiny_texture_z_full.npz"),
                                                                 os.path.join(GOLDEN, "ref_generator_tiny.pth"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    print("[parity] pickled reference generator -> HIP staged_forward:", r.stdout.strip())


# ---------------------------------------------------------------------------------------------------
# backward: composite gradient vs torch autograd of the fp64 restatement
# ---------------------------------------------------------------------------------------------------
def _grad_case(BR, N, C, seed, merge):
    rng = np.random.default_rng(seed)
    M = 2 * N if merge else N
    rows = rng.normal(size=(BR, M, C)).astype(np.float32)
    rows[..., -1] = rng.normal(size=(BR, M)).astype(np.float32) * 6 * (1 + (np.arange(BR) % 5))[:, None]
    z = np.sort(rng.uniform(0.88, 1.12, (BR, M)).astype(np.float32), -1)
    if merge:
        z = rng.permuted(z, axis=-1)     # fine / coarse halves are each unsorted relative to the union
        z[:, :N] = np.sort(z[:, :N], -1); z[:, N:] = np.sort(z[:, N:], -1)
    noise = rng.normal(size=(BR, M)).astype(np.float32)
    g = rng.normal(size=(BR, C - 1)).astype(np.float32)
    return rows, z, noise, g


@pytest.mark.parametrize("clamp,last_back,white,black,noise_std", [("relu", False, False, False, 0.0), ("softplus", False, False, False, 0.5),
                                                                    ("relu", True, False, False, 0.3), ("relu", False, True, False, 0.0),
                                                                    ("softplus", True, False, True, 0.2)])
@pytest.mark.parametrize("N", [1, 7, 24, 64, 100, 150, 256, 300, 512, 700, 1024])
def test_composite_backward_vs_autograd(N, clamp, last_back, white, black, noise_std):
    from oracle import fenerf_oracle_grad as OG
    rows, z, noise, g = _grad_case(37, N, 22, 5 + N, False)
    opts = _lib.composite_opts(clamp, last_back=last_back, white_back=white, black_back=black, noise_std=noise_std)
    got = N_(native.composite_backward(T(g), T(rows), T(z), opts, noise=T(noise)))
    r = torch.tensor(rows, dtype=torch.float64, requires_grad=True)
    rgb, _, _ = OG.composite(r, torch.tensor(z, dtype=torch.float64), torch.tensor(noise, dtype=torch.float64), noise_std=noise_std,
                             clamp_mode=clamp, last_back=last_back, white_back=white, black_back=black)
    (rgb * torch.tensor(g, dtype=torch.float64)).sum().backward()
    ref = r.grad.numpy()
    scale = max(1.0, np.abs(ref).max())
    err = np.abs(got - ref).max()
    print(f"[parity] composite backward N={N} {clamp} lb={last_back}: max|err| {err:.2e} (|grad| max {scale:.3g})")
    assert err <= 2e-5 * scale


@pytest.mark.parametrize("N", [12, 24, 64, 72, 128, 200, 300, 512])
def test_merge_composite_backward_vs_autograd(N):
    from oracle import fenerf_oracle_grad as OG
    rows, z, noise, g = _grad_case(29, N, 22, 40 + N, True)
    opts = _lib.composite_opts("relu", noise_std=0.4)
    df, dc = native.composite_backward(T(g), T(rows[:, :N]), T(z[:, :N]), opts, rows_b=T(rows[:, N:]), z_b=T(z[:, N:]), noise=T(noise))
    f = torch.tensor(rows[:, :N], dtype=torch.float64, requires_grad=True)
    c = torch.tensor(rows[:, N:], dtype=torch.float64, requires_grad=True)
    rgb, _, _ = OG.merge_composite(f, c, torch.tensor(z[:, :N], dtype=torch.float64), torch.tensor(z[:, N:], dtype=torch.float64),
                                   torch.tensor(noise, dtype=torch.float64), noise_std=0.4, clamp_mode="relu")
    (rgb * torch.tensor(g, dtype=torch.float64)).sum().backward()
    scale = max(1.0, f.grad.abs().max().item(), c.grad.abs().max().item())
    err = max(np.abs(N_(df) - f.grad.numpy()).max(), np.abs(N_(dc) - c.grad.numpy()).max())
    print(f"[parity] merge composite backward N={N}: max|err| {err:.2e} (|grad| max {scale:.3g})")
    assert err <= 2e-5 * scale


# ---------------------------------------------------------------------------------------------------
# backward: fused SIREN chain kernel + point-axis reductions vs torch autograd of the fp64 restatement
# ---------------------------------------------------------------------------------------------------
def _siren_module(kind, H, grid, seed=4, sigma_gain=30.0, precision="f16x3"):
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=seed, sigma_gain=sigma_gain, with_mapping=False)
    if kind == "spatial":
        mod = S.SPATIALSIRENBASELINE(hidden_dim=H, z_dim=8)
    else:
        cls = {"texture": S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, "baseline": S.SIRENBASELINESEMANTICDISENTANGLE}[kind]
        mod = cls(hidden_dim=H, z_geo_dim=8, z_app_dim=8, output_dim=22)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    if "spatial_embeddings" in tsd:
        mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
    mod.load_state_dict(tsd, strict=False)
    # "tape16" (round 5) = f16x3 kernels with the 16-bit tape between forward and backward (siren.grad_precision, FENERF_TAPE_U16)
    mod.precision = "f16x3" if precision == "tape16" else precision
    if precision == "tape16":
        mod.grad_precision = "tape16"
    return mod.to(DEV), spec, sd


def _rel_err(got, ref):
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12))


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
@pytest.mark.parametrize("kind,H,grid,B,P", [("texture", 32, 5, 2, 75), ("baseline", 64, 0, 1, 64), ("spatial", 32, 0, 2, 33),
                                             ("texture", 128, 4, 3, 130), ("texture", 256, 6, 2, 200),
                                             ("texture", 96, 5, 2, 75), ("baseline", 192, 0, 1, 64), ("spatial", 96, 0, 2, 33),      # round 5: H = 96 / 192
                                             # round 6: widths BETWEEN the instantiated ones run zero-padded at the next one (native.padded_hidden_dim)
                                             ("texture", 100, 5, 2, 75), ("baseline", 40, 0, 1, 64), ("spatial", 72, 0, 2, 33), ("texture", 250, 4, 1, 96)])
def test_siren_backward_vs_autograd(kind, H, grid, B, P, precision):
    from oracle import fenerf_oracle_grad as OG
    mod, spec, sd = _siren_module(kind, H, grid, precision=precision)
    rng = np.random.default_rng(7)
    pts = rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32)     # some points leave the grid box: zero padding
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, B, seed=4)
    if kind == "spatial":
        film["freq_app"] = proc.normal("film.freq_app", (B, H), 0.4, 4)
        film["phase_app"] = proc.normal("film.phase_app", (B, H), 0.4, 4)
    Cc = spec["output_dim"]
    g_out = rng.normal(size=(B, P, Cc)).astype(np.float32)
    g_out[..., -1] *= 0.02      # sigma is ~sigma_gain x larger than the other outputs; keep the contributions comparable

    film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
    if kind == "spatial":
        out = mod.forward_with_frequencies_phase_shifts(T(pts), torch.cat([film_t["freq_geo"], film_t["freq_app"]], -1),
                                                        torch.cat([film_t["phase_geo"], film_t["phase_app"]], -1), T(dirs))
    else:
        out = mod.forward_with_frequencies_phase_shifts(T(pts), film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"],
                                                        film_t["phase_app"], T(dirs))
    assert out.requires_grad
    (out * T(g_out)).sum().backward()

    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    sd64 = {k: t64(v).requires_grad_(True) for k, v in sd.items()}
    film64 = {k: t64(v).requires_grad_(True) for k, v in film.items()}
    ref = OG.siren_forward(sd64, spec, t64(pts), t64(dirs), film64["freq_geo"], film64["phase_geo"], film64["freq_app"], film64["phase_app"])
    (ref * t64(g_out)).sum().backward()
    fwd_err = np.abs(N_(out) - ref.detach().numpy())
    print(f"[parity] differentiable forward {kind} H={H}: max|err| rgb/labels {fwd_err[..., :-1].max():.2e} sigma {fwd_err[..., -1].max():.2e}")
    assert fwd_err[..., :-1].max() <= 1e-4 and fwd_err[..., -1].max() <= 2e-4 * 30

    worst = 0.0
    for k in film:
        e = _rel_err(N_(film_t[k].grad), film64[k].grad.numpy())
        worst = max(worst, e)
        assert e <= 2e-4, (k, e)
    named = dict(mod.named_parameters())
    for k, v in sd64.items():
        assert named[k].grad is not None, k
        e = _rel_err(N_(named[k].grad), v.grad.numpy())
        worst = max(worst, e)
        assert e <= 2e-4, (k, e)
    print(f"[parity] SIREN backward {precision} {kind} H={H} B={B} P={P}: worst relative error over {len(sd64) + len(film)} gradient tensors {worst:.2e}")


# max-norm relative error of d points / d view directions vs fp64 autograd: measured (profiles/r06_gpu_tests_parity_lines.log) x 1.5
INPUT_GRAD_BOUND = {"f32": 2.1e-5, "f16x3": 4.2e-5, "tape16": 1.9e-4}


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
@pytest.mark.parametrize("kind,H,grid,B,P", [("texture", 32, 5, 2, 75), ("baseline", 64, 0, 1, 64), ("spatial", 32, 0, 2, 33),
                                             ("texture", 256, 6, 2, 200), ("texture", 100, 5, 2, 75), ("baseline", 192, 0, 3, 160)])
def test_siren_input_gradients_vs_fp64_autograd(kind, H, grid, B, P, precision):
    """Gradients wrt the sample positions and view directions of forward_with_frequencies_phase_shifts (siren.py:1509-1530: layer 0,
    grid_sample's coordinate gradient :314-330, UniformBoxWarp :181-187, the colour layer's cat :1522) -- fenerf_siren_input_grads, an extra
    pass over the d(theta) dump -- against fp64 autograd of the oracle; asking for them changes no other gradient by a bit; the
    FiLM-only route (frozen weights) and a chunked backward give the same rows."""
    from oracle import fenerf_oracle_grad as OG
    from fenerf_amd.siren import autograd as SA
    mod, spec, sd = _siren_module(kind, H, grid, precision=precision)
    rng = np.random.default_rng(11)
    pts = rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32)     # some points leave the grid box: zero padding
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, B, seed=4)
    if kind == "spatial":
        film["freq_app"] = proc.normal("film.freq_app", (B, H), 0.4, 4)
        film["phase_app"] = proc.normal("film.phase_app", (B, H), 0.4, 4)
    g_out = rng.normal(size=(B, P, spec["output_dim"])).astype(np.float32)
    g_out[..., -1] *= 0.02

    def run(points_grad, dirs_grad, film_grad=True):
        mod.zero_grad(set_to_none=True)
        film_t = {k: T(v).requires_grad_(film_grad) for k, v in film.items()}
        p_t, d_t = T(pts).requires_grad_(points_grad), T(dirs).requires_grad_(dirs_grad)
        if kind == "spatial":
            out = mod.forward_with_frequencies_phase_shifts(p_t, torch.cat([film_t["freq_geo"], film_t["freq_app"]], -1),
                                                            torch.cat([film_t["phase_geo"], film_t["phase_app"]], -1), d_t)
        else:
            out = mod.forward_with_frequencies_phase_shifts(p_t, film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], d_t)
        (out * T(g_out)).sum().backward()
        others = {k: N_(v.grad) for k, v in film_t.items() if v.grad is not None}
        others.update({k: N_(v.grad) for k, v in mod.named_parameters() if v.grad is not None})
        return (N_(p_t.grad) if points_grad else None), (N_(d_t.grad) if dirs_grad else None), others

    gp, gd, with_inputs = run(True, True)
    _, _, without = run(False, False)
    differing = [k for k in without if not np.array_equal(with_inputs[k], without[k])]
    # (the grid gradient is a scatter of float atomics: equal to summation order only)
    assert set(with_inputs) == set(without) and set(differing) <= {"spatial_embeddings"}, differing
    if differing:
        assert _rel_err(with_inputs["spatial_embeddings"], without["spatial_embeddings"]) <= 1e-5

    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    sd64 = {k: t64(v) for k, v in sd.items()}
    p64, d64 = t64(pts).requires_grad_(True), t64(dirs).requires_grad_(True)
    ref = OG.siren_forward(sd64, spec, p64, d64, t64(film["freq_geo"]), t64(film["phase_geo"]), t64(film["freq_app"]), t64(film["phase_app"]))
    (ref * t64(g_out)).sum().backward()
    ep, ed = _rel_err(gp, p64.grad.numpy()), _rel_err(gd, d64.grad.numpy())
    print(f"[parity] SIREN input gradients {precision} {kind} H={H} B={B} P={P}: d points {ep:.2e} (max |ref| {np.abs(p64.grad.numpy()).max():.2e}), "
          f"d view directions {ed:.2e} vs fp64 autograd")
    assert ep <= INPUT_GRAD_BOUND[precision] and ed <= INPUT_GRAD_BOUND[precision], (ep, ed)

    # only the positions: (weights and FiLM parameters frozen: the backward is the FiLM-only one + the dump)
    for q in mod.parameters():
        q.requires_grad_(False)
    gp2, gd2, none = run(True, False, film_grad=False)
    assert gd2 is None and not none
    e2 = _rel_err(gp2, p64.grad.numpy())
    assert e2 <= INPUT_GRAD_BOUND[precision], e2
    gp3, gd3, _ = run(False, True, film_grad=False)
    assert gp3 is None and _rel_err(gd3, d64.grad.numpy()) <= INPUT_GRAD_BOUND[precision]
    for q in mod.parameters():
        q.requires_grad_(True)
    # a backward in several chunks fills the same rows
    old = SA.BACKWARD_CHUNK_POINTS
    try:
        SA.BACKWARD_CHUNK_POINTS = 128
        gp4, gd4, _ = run(True, True)
    finally:
        SA.BACKWARD_CHUNK_POINTS = old
    assert np.array_equal(gp4, gp) and np.array_equal(gd4, gd)


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
@pytest.mark.parametrize("name", ["tiny_texture_input_grad", "tiny_baseline_input_grad", "tiny_spatial_input_grad"])
def test_siren_input_gradients_vs_reference_autograd(name, precision):
    """input.grad / ray_directions.grad of the reference module's own forward_with_frequencies_phase_shifts (fixtures made by
    tools/make_golden.py::run_input_grad_case from the imported reference) against fenerf_siren_input_grads."""
    g = load_golden(name)
    spec = spec_from_golden(g)
    kind, H = spec["kind"], spec["hidden_dim"]
    mod, spec2, sd = _siren_module(kind, H, spec.get("grid_size", 0), seed=int(g["meta_seed"]), sigma_gain=float(g["meta_sigma_gain"]), precision=precision)
    assert abs(proc.checksum(proc.make_state_dict(spec, seed=int(g["meta_seed"]), sigma_gain=float(g["meta_sigma_gain"]))) - float(g["meta_weights_checksum"])) < 1e-9
    ref_sd = weights_from_golden(g, spec, with_mapping=False)
    assert all(np.array_equal(ref_sd[k], sd[k]) for k in sd)
    film = {k: T(v) for k, v in film_from_golden(g, spec).items()}
    pts, dirs = T(g["points"]).requires_grad_(True), T(g["dirs"]).requires_grad_(True)
    if kind == "spatial":
        out = mod.forward_with_frequencies_phase_shifts(pts, torch.cat([film["freq_geo"], film["freq_app"]], -1),
                                                        torch.cat([film["phase_geo"], film["phase_app"]], -1), dirs)
    else:
        out = mod.forward_with_frequencies_phase_shifts(pts, film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], dirs)
    fwd = np.abs(N_(out) - g["out"])
    (out * T(g["loss_w"])).sum().backward()
    ep, ed = _rel_err(N_(pts.grad), g["d_points"]), _rel_err(N_(dirs.grad), g["d_dirs"])
    print(f"[parity] {name} [{precision}] vs the reference's autograd: forward rgb/labels {fwd[..., :-1].max():.2e}, d points {ep:.2e}, d view directions {ed:.2e}")
    assert fwd[..., :-1].max() <= 2e-5
    assert ep <= INPUT_GRAD_BOUND[precision] * 2 and ed <= INPUT_GRAD_BOUND[precision] * 2, (ep, ed)      # (x 2: the reference's own fp32 autograd rounding)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_siren_input_gradients_at_scale(precision):
    """The input-gradient kernel at a size where every workgroup walks several units of its image (fenerf_siren_inputgrad.hip): 262,272 points
    in one chunk of two images, 4,098 tiles per image, every row against fp64 autograd."""
    from oracle import fenerf_oracle_grad as OG
    from fenerf_amd.siren import autograd as SA
    kind, H, grid, B, P = "texture", 32, 6, 2, 4098 * 32
    assert B * P <= SA.BACKWARD_CHUNK_POINTS
    mod, spec, sd = _siren_module(kind, H, grid, precision=precision)
    for q in mod.parameters():
        q.requires_grad_(False)
    rng = np.random.default_rng(13)
    pts = rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32)
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, B, seed=4)
    g_out = rng.normal(size=(B, P, spec["output_dim"])).astype(np.float32)
    g_out[..., -1] *= 0.02
    p_t, d_t = T(pts).requires_grad_(True), T(dirs).requires_grad_(True)
    out = mod.forward_with_frequencies_phase_shifts(p_t, T(film["freq_geo"]), T(film["freq_app"]), T(film["phase_geo"]), T(film["phase_app"]), d_t)
    (out * T(g_out)).sum().backward()
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    sd64 = {k: t64(v) for k, v in sd.items()}
    gp, gd = np.empty((B, P, 3)), np.empty((B, P, 3))
    for s0 in range(0, P, 32768):
        sl = slice(s0, min(P, s0 + 32768))
        p64, d64 = t64(pts[:, sl]).requires_grad_(True), t64(dirs[:, sl]).requires_grad_(True)
        ref = OG.siren_forward(sd64, spec, p64, d64, t64(film["freq_geo"]), t64(film["phase_geo"]), t64(film["freq_app"]), t64(film["phase_app"]))
        (ref * t64(g_out[:, sl])).sum().backward()
        gp[:, sl], gd[:, sl] = p64.grad.numpy(), d64.grad.numpy()
    # the trilinear gather's coordinate gradient jumps across voxel faces: a point whose grid coordinate lies within fp32 rounding of
    # a face may sit in the neighbouring cell in fp32 (the reference's own arithmetic, siren.py:314-330) and in fp64 -- compared apart
    gi = (pts.astype(np.float64) * (2 / 0.24) + 1) / 2 * (grid - 1)
    on_face = (np.abs(gi - np.round(gi)) < 2e-5).any(-1)
    assert on_face.sum() <= 64
    scale = np.abs(gp).max()
    ep = float(np.abs(N_(p_t.grad) - gp)[~on_face].max() / scale)
    ep_face = float(np.abs(N_(p_t.grad) - gp)[on_face].max() / scale) if on_face.any() else 0.0
    ed = _rel_err(N_(d_t.grad), gd)
    print(f"[parity] SIREN input gradients at scale [{precision}] H={H} B={B} P={P}: d points {ep:.2e} "
          f"({int(on_face.sum())} points within 2e-5 of a voxel face: {ep_face:.2e}), d view directions {ed:.2e} vs fp64 autograd")
    # yardstick for a maximum over 262,272 per-point quantities (the small cases above take it over 33 .. 480 points): the same maths as
    # torch fp32 autograd -- the reference's own arithmetic -- against the same fp64 values
    c32 = lambda a_: torch.tensor(np.asarray(a_), dtype=torch.float32, device=DEV)
    sd32 = {k: c32(v) for k, v in sd.items()}
    p32, d32 = c32(pts).requires_grad_(True), c32(dirs).requires_grad_(True)
    ref32 = OG.siren_forward(sd32, spec, p32, d32, c32(film["freq_geo"]), c32(film["phase_geo"]), c32(film["freq_app"]), c32(film["phase_app"]))
    (ref32 * c32(g_out)).sum().backward()
    ep32 = float(np.abs(N_(p32.grad) - gp)[~on_face].max() / scale)
    ed32 = _rel_err(N_(d32.grad), gd)
    med = float(np.median(np.abs(N_(p_t.grad) - gp)) / scale)
    print(f"[parity]   ... torch fp32 autograd of the same maths vs fp64: d points {ep32:.2e}, d view directions {ed32:.2e}; median |err| of d points {med:.1e}")
    # measured: d points 5.7e-5 (f32) / 5.4e-5 (f16x3) where torch fp32 autograd sits at 3.8e-5; d view directions 1.5e-5 / 1.2e-5
    assert ep <= max(INPUT_GRAD_BOUND[precision], 2 * ep32) and ed <= max(INPUT_GRAD_BOUND[precision], 2 * ed32) and ep_face <= 5e-3, (ep, ed, ep_face)
    assert med <= 2e-6


def test_siren_input_gradients_api_refuses_what_it_cannot_do():
    """fenerf_siren_input_grads: argument checks, and the bf16 dump of an AMP-class chunk is refused (Python: a NotImplementedError up front)."""
    import ctypes
    mod, spec, sd = _siren_module("texture", 32, 4, precision="f16x3")
    B, P = 1, 64
    nat = mod.native_differentiable(DEV)
    film = {k: T(v) for k, v in proc.film_params(spec, B, seed=4).items()}
    pts = torch.zeros(B, P, 3, device=DEV)
    l = _lib.lib()
    ws = torch.empty(int(l.fenerf_film_workspace_bytes(nat._h, B)), dtype=torch.uint8, device=DEV)
    d_t = torch.zeros(int(l.fenerf_siren_dtheta_floats(nat._h, B * P)), device=DEV)
    w0, wc0 = mod.network[0].layer.weight.detach().contiguous(), mod.color_layer_sine[0].layer.weight.detach().contiguous()
    out = torch.empty(B, P, 3, device=DEV)
    p = native._ptr
    args = lambda **kw: [kw.get("h", nat._h), B, kw.get("P", P), p(pts), p(film["freq_geo"]), p(film["phase_geo"]), p(film["freq_app"]), p(film["phase_app"]),
                         kw.get("d_t", p(d_t)), p(w0), p(wc0), kw.get("ld", wc0.shape[1]), kw.get("dp", p(out)), kw.get("dd", None),
                         ctypes.c_void_p(ws.data_ptr()), None]
    assert l.fenerf_siren_input_grads(*args()) == 0
    assert l.fenerf_siren_input_grads(*args(P=33)) == _lib.E_INVALID
    assert l.fenerf_siren_input_grads(*args(d_t=None)) == _lib.E_INVALID
    assert l.fenerf_siren_input_grads(*args(dp=None)) == _lib.E_INVALID          # neither output
    assert l.fenerf_siren_input_grads(*args(ld=10)) == _lib.E_INVALID
    assert l.fenerf_siren_input_grads(*args(P=0)) == 0
    plain = mod.native(DEV)                                                       # not differentiable
    assert l.fenerf_siren_input_grads(*args(h=plain._h)) == _lib.E_UNSUPPORTED
    mod.grad_precision = "amp"
    with pytest.raises(NotImplementedError):
        mod.forward_with_frequencies_phase_shifts(pts.clone().requires_grad_(True), film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"],
                                                  torch.zeros_like(pts))
    amp = mod.native_differentiable(DEV)
    if amp.wgrad_bf16_min_points:
        big_P = ((amp.wgrad_bf16_min_points + 31) // 32) * 32
        rc = l.fenerf_siren_input_grads(amp._h, 1, big_P, p(pts), p(film["freq_geo"]), p(film["phase_geo"]), p(film["freq_app"]), p(film["phase_app"]),
                                        p(d_t), p(w0), p(wc0), wc0.shape[1], p(out), None, ctypes.c_void_p(ws.data_ptr()), None)
        assert rc == _lib.E_UNSUPPORTED and b"bf16" in l.fenerf_last_error()


def test_hidden_width_between_the_instantiated_ones_is_the_padded_network_bit_for_bit():
    """The reference constructs any hidden width (siren.py:1451); the kernels are instantiated for 32 / 64 / 96 / 128 / 192 / 256.  Round 6:
    another width up to 256 runs at the next instantiated one with zero padding (native.padded_hidden_dim): padded features are
    sin(f' 0 + 0) = 0 and add exact zeros to every sum.  A 100-wide generator against the SAME network written out as a 128-wide module
    with zero rows / columns: hierarchical render under autograd -- pixels bit-identical, every gradient equal on the real entries (the
    padded module's gradients of padded entries are whatever they are) --, then an optimizer step on the 100-wide module (device-side re-pack
    through the padding) and a no-grad render against the oracle at the updated weights."""
    H, Hp = 100, 128
    mod, spec, sd = _siren_module("texture", H, 5, sigma_gain=150.0)
    big, spec_b, _ = _siren_module("texture", Hp, 5, sigma_gain=150.0)
    assert native.padded_hidden_dim(H) == Hp and native.padded_hidden_dim(33) == 64 and native.padded_hidden_dim(256) == 256
    with pytest.raises(ValueError):
        native.padded_hidden_dim(257)
    with torch.no_grad():
        for (n, p), (nb, pb) in zip(mod.named_parameters(), big.named_parameters()):
            assert n == nb
            if "mapping_network" in n or n == "spatial_embeddings":
                if n == "spatial_embeddings":
                    pb.copy_(p)
                continue
            pb.zero_()
            if p.dim() == 1:
                pb[:p.shape[0]] = p
            elif n.startswith("color_layer_sine.0."):          # [dirs | grid | x]: the hidden part is the trailing columns
                pb[:H, :35] = p[:, :35]
                pb[:H, 35:35 + H] = p[:, 35:]
            else:
                pb[:p.shape[0], :p.shape[1]] = p
    gens = []
    for m_, w in ((mod, H), (big, Hp)):
        gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=w), 8, 8, 22)
        gen.siren = m_
        gen = gen.to(DEV).train()
        gen.device = torch.device(DEV); gen.siren.device = gen.device
        gens.append(gen)
    B = 2
    film = proc.film_params(spec, B, seed=9)
    pad = lambda a, n: np.pad(a.reshape(B, n, H), ((0, 0), (0, 0), (0, Hp - H))).reshape(B, n * Hp)
    film_b = dict(freq_geo=pad(film["freq_geo"], 8), phase_geo=pad(film["phase_geo"], 8), freq_app=pad(film["freq_app"], 3), phase_app=pad(film["phase_app"], 3))
    kw = dict(img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
              hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.3)
    w = T(np.random.default_rng(2).normal(size=(B, 21, 8, 8)).astype(np.float32))
    res = []
    for gen, f in ((gens[0], film), (gens[1], film_b)):
        ft = {k: T(v).requires_grad_(True) for k, v in f.items()}
        torch.manual_seed(77)
        px, _ = gen.forward_with_frequencies(ft["freq_geo"], ft["freq_app"], ft["phase_geo"], ft["phase_app"], **kw)
        (px * w).sum().backward()
        res.append((N_(px), {k: N_(v.grad) for k, v in ft.items()}, {n: N_(p.grad) for n, p in gen.siren.named_parameters() if p.grad is not None}))
    (px_a, fg_a, pg_a), (px_b, fg_b, pg_b) = res
    assert np.array_equal(px_a, px_b), "pixels of the padded evaluation are those of the 128-wide network with zero rows, bit for bit"
    worst = 0.0
    for k, n in (("freq_geo", 8), ("phase_geo", 8), ("freq_app", 3), ("phase_app", 3)):
        assert fg_a[k].shape == (B, n * H)
        worst = max(worst, _rel_err(fg_a[k], fg_b[k].reshape(B, n, Hp)[..., :H].reshape(B, n * H)))
    for n, g in pg_a.items():
        gb = pg_b[n]
        assert g.shape == tuple(dict(mod.named_parameters())[n].shape)
        if n == "spatial_embeddings":
            ref = gb
        elif g.ndim == 1:
            ref = gb[:g.shape[0]]
        elif n.startswith("color_layer_sine.0."):
            ref = np.concatenate([gb[:H, :35], gb[:H, 35:35 + H]], 1)
        else:
            ref = gb[:g.shape[0], :g.shape[1]]
        worst = max(worst, _rel_err(g, ref))
    print(f"[parity] hidden_dim 100 (run at 128, zero-padded) vs the same network as a 128-wide module: pixels bit-identical; worst relative gradient "
          f"difference over {len(pg_a) + 4} tensors {worst:.1e}")
    assert worst <= 2e-6          # the same kernels on the same values; only the atomically scattered grid gradient may differ in the last bits
    # optimizer step -> device-side re-pack through the padding -> no-grad render vs the oracle at the new weights
    with torch.no_grad():       # an in-place update of every render parameter, 1e-3 of its gradient's scale (version counters bump like an optimizer's)
        for n, p in mod.named_parameters():
            if mod._is_render_param(n) and p.grad is not None:
                p.add_(p.grad / p.grad.abs().max().clamp_min(1e-30), alpha=-1e-3)
    sd2 = {n: N_(p) for n, p in mod.named_parameters() if mod._is_render_param(n)}
    rng = np.random.default_rng(3)
    pts = rng.uniform(-0.11, 0.11, (B, 64, 3)).astype(np.float32)
    dirs = rng.normal(size=(B, 64, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    for grad_mode in (False, True):
        with torch.set_grad_enabled(grad_mode):
            out = N_(mod.forward_with_frequencies_phase_shifts(T(pts), T(film["freq_geo"]), T(film["freq_app"]), T(film["phase_geo"]), T(film["phase_app"]), T(dirs)))
        ref = O.siren_forward(sd2, spec, pts, dirs, film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
        e = np.abs(out - ref)
        assert e[..., :-1].max() <= 2e-5 and e[..., -1].max() <= 1e-5 * max(float(np.abs(ref[..., -1]).max()), 1.0), (grad_mode, e[..., :-1].max(), e[..., -1].max())


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
def test_generator_gradient_end_to_end(precision):
    """g_loss.backward() through DoubleImplicitGenerator3d.forward (hierarchical 12+12, noise, last_back off): gradients of
    a pixel loss wrt z-mapped FiLM parameters and every render weight vs torch autograd of the fp64 restatement run on the
    SAME rays, resampled depths and noise (those are no_grad constants in the reference too)."""
    from oracle import fenerf_oracle_grad as OG
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0, precision=precision)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N = 2, 6, 12
    film = proc.film_params(spec, B, seed=4)
    film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
    torch.manual_seed(11)
    px, poses = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
    assert px.requires_grad and px.shape == (B, 21, S_, S_)
    w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    (px * w).sum().backward()

    # replay: same seed -> same draws; recompute the constants (rays, z, resampled z, noise) with the no-grad HIP pieces
    torch.manual_seed(11)
    R = S_ * S_
    origins, dirs, z_vals, _, _ = VR.sample_rays(B, N, gen.device, kw["fov"], (S_, S_), kw["ray_start"], kw["ray_end"], kw["h_stddev"],
                                                 kw["v_stddev"], kw["h_mean"], kw["v_mean"], kw["sample_dist"], draws=gen.draws)
    noise_c = gen.draws.randn((B, R, N, 1), gen.device); u = gen.draws.rand((B * R, N), gen.device)
    noise_f = gen.draws.randn((B, R, 2 * N, 1), gen.device)
    z_c = z_vals.reshape(B, R, N)
    nat = mod.native_differentiable(DEV)
    with torch.no_grad():
        pts_c = origins.unsqueeze(2) + dirs.unsqueeze(2) * z_c.unsqueeze(-1)
        rd = dirs.unsqueeze(2).expand(-1, -1, N, -1).reshape(B, R * N, 3)
        coarse = nat.siren_forward(pts_c.reshape(B, R * N, 3), rd, *(film_t[k] for k in ("freq_geo", "phase_geo", "freq_app", "phase_app")))
        _, _, w_c, _ = native.composite(coarse.reshape(B * R, N, 22), z_c.reshape(B * R, N), noise_c.reshape(B * R, N),
                                        _lib.composite_opts("relu", 0.2), want_wsum=False)
        z_f = native.resample(z_c.reshape(B * R, N), w_c, u).reshape(B, R, N)
        pts_f = origins.unsqueeze(2) + dirs.unsqueeze(2) * z_f.unsqueeze(-1)
    t64 = lambda a: torch.as_tensor(N_(a) if torch.is_tensor(a) else np.asarray(a), dtype=torch.float64)
    sd64 = {k: t64(v).requires_grad_(True) for k, v in sd.items()}
    film64 = {k: t64(v).requires_grad_(True) for k, v in film.items()}
    args = (film64["freq_geo"], film64["phase_geo"], film64["freq_app"], film64["phase_app"])
    c64 = OG.siren_forward(sd64, spec, t64(pts_c.reshape(B, R * N, 3)), t64(rd), *args)
    f64 = OG.siren_forward(sd64, spec, t64(pts_f.reshape(B, R * N, 3)), t64(rd), *args)
    rgb, _, _ = OG.merge_composite(f64.reshape(B * R, N, 22), c64.reshape(B * R, N, 22), t64(z_f.reshape(B * R, N)), t64(z_c.reshape(B * R, N)),
                                   t64(noise_f.reshape(B * R, 2 * N)), noise_std=0.2, clamp_mode="relu")
    ref_px = rgb.reshape(B, S_, S_, 21).permute(0, 3, 1, 2) * 2 - 1
    (ref_px * t64(w)).sum().backward()
    fe = np.abs(N_(px) - ref_px.detach().numpy()).max()
    print(f"[parity] differentiable generator forward: max|err| {fe:.2e}")
    assert fe <= 1e-3
    worst = 0.0
    for k in film:
        worst = max(worst, _rel_err(N_(film_t[k].grad), film64[k].grad.numpy()))
    named = dict(mod.named_parameters())
    for k, v in sd64.items():
        worst = max(worst, _rel_err(N_(named[k].grad), v.grad.numpy()))
    print(f"[parity] generator gradient end to end: worst relative error {worst:.2e}")
    assert worst <= 5e-4


def test_backward_refuses_weights_repacked_after_the_forward():
    """The backward kernels read the model's resident backward stream: a re-pack between a render's forward and its backward
    (optimizer step + another render of the same module before .backward()) is refused, not silently differentiated."""
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=50.0)
    rng = np.random.default_rng(2)
    pts = T(rng.uniform(-0.1, 0.1, (1, 64, 3)).astype(np.float32))
    dirs = T(rng.normal(size=(1, 64, 3)).astype(np.float32))
    film = {k: T(v) for k, v in proc.film_params(spec, 1, seed=4).items()}
    call = lambda: mod.forward_with_frequencies_phase_shifts(pts, film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], dirs)
    out = call()
    out2 = call()                                   # a second render of unchanged weights re-packs nothing
    out2.sum().backward()
    with torch.no_grad():
        next(iter(mod._render_params())).add_(1e-3)   # "optimizer.step()"
    call()                                          # ... and a new render: the streams are rebuilt on the device
    with pytest.raises(RuntimeError, match="re-packed"):
        out.sum().backward()
    # ... but a forced re-pack of UNCHANGED weights (train() / eval() round trip, invalidate_native()) between a forward and its
    # backward is recognised as such (content compare) and leaves the node valid
    for p_ in mod.parameters():
        p_.grad = None
    out3 = call()
    mod.eval(); mod.train(); mod.invalidate_native()
    out4 = call()
    out3.sum().backward()
    g3 = {k: N_(p_.grad).copy() for k, p_ in mod.named_parameters() if p_.grad is not None}
    for p_ in mod.parameters():
        p_.grad = None
    out4.sum().backward()
    # (equal up to the order of the grid scatter's float atomics)
    assert all(_rel_err(N_(p_.grad), g3[k]) <= 1e-5 for k, p_ in mod.named_parameters() if p_.grad is not None)
    # a write through param.data followed by the mode switch IS a change: refused again
    out5 = call()
    next(iter(mod._render_params())).data.mul_(1.001)
    mod.eval(); mod.train()
    call()
    with pytest.raises(RuntimeError, match="re-packed"):
        out5.sum().backward()


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
def test_chunked_backward_equals_one_pass(precision):
    """The backward runs chain + weight-gradient kernels per chunk of points (siren/autograd.py: bounded dtheta); chunk results
    add.  One launch over all 2 x 2 (pass, image) "images" (B > 1 kernels, per-image FiLM blocks) against 128-point chunks (4-5
    chunks per image here, ragged last chunk) and against chunks of whole images (1,700 points: 3 images + 1): every gradient
    equals the single-launch one up to fp32 summation order."""
    from fenerf_amd.siren import autograd as SA
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0, precision=precision)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N = 2, 7, 11                      # 539 points per image and pass -> padded to 544 = 4 x 128 + 32
    film = proc.film_params(spec, B, seed=4)
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
    results = []
    for chunk, overlap in ((1 << 30, False), (128, False), (1700, False), (128, True)):
        # (128, True): the two-stream schedule of round 4 -- weight gradients of chunk i beside the chain of chunk i + 1 under CU budgets
        # (OVERLAP_WGRAD; measured not to pay, profiles/r04_gstep_overlap.md, and off by default: kept exact here)
        SA_old, SA.BACKWARD_CHUNK_POINTS = SA.BACKWARD_CHUNK_POINTS, chunk
        ov_old, SA.OVERLAP_WGRAD = SA.OVERLAP_WGRAD, overlap
        try:
            film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
            for p_ in mod.parameters():
                p_.grad = None
            torch.manual_seed(11)
            px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
            w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            (px * w).sum().backward()
            grads = {k: N_(v.grad) for k, v in film_t.items()}
            grads.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
            results.append(grads)
        finally:
            SA.BACKWARD_CHUNK_POINTS, SA.OVERLAP_WGRAD = SA_old, ov_old
    one, many, grouped, overlapped = results
    assert one.keys() == many.keys() == grouped.keys() == overlapped.keys() and len(one) > 30
    worst = max(_rel_err(many[k], one[k]) for k in one)
    worst_g = max(_rel_err(grouped[k], one[k]) for k in one)
    worst_o = max(_rel_err(overlapped[k], one[k]) for k in one)
    print(f"[parity] chunked backward vs ONE launch over all images [{precision}]: worst relative difference over {len(one)} tensors "
          f"{worst:.1e} (128-point chunks), {worst_g:.1e} (whole-image chunks), {worst_o:.1e} (128-point chunks, two-stream schedule)")
    assert worst <= 2e-5 and worst_g <= 2e-5 and worst_o <= 2e-5


@pytest.mark.parametrize("precision", PRECISIONS)
def test_device_side_repack_matches_host_pack(precision):
    """optimizer.step() analogue: change the weights on the GPU, re-pack on the device (index-map gather +
    fenerf_model_load_packed) and compare forward outputs and gradients with a model packed on the host from the same values."""
    mod, spec, sd = _siren_module("texture", 64, 5, precision=precision)
    B, P = 2, 96
    rng = np.random.default_rng(3)
    pts = T(rng.uniform(-0.12, 0.12, (B, P, 3)).astype(np.float32))
    dirs = T(rng.normal(size=(B, P, 3)).astype(np.float32))
    film = {k: T(v) for k, v in proc.film_params(spec, B, seed=4).items()}
    args = (film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"])
    g_out = T(rng.normal(size=(B, P, 22)).astype(np.float32))

    def run(m):
        for p in m.parameters():
            p.grad = None
        out = m.forward_with_frequencies_phase_shifts(pts, *args, dirs)
        (out * g_out).sum().backward()
        return N_(out), {n: N_(p.grad) for n, p in m.named_parameters() if p.grad is not None}

    run(mod)                                         # creates the native model (host pack)
    nat = mod.native_differentiable(DEV)
    with torch.no_grad():
        for i, p in enumerate(mod._render_params()):
            p.mul_(1.0 + 0.01 * ((i % 5) - 2)).add_(1e-3)
    out_dev, g_dev = run(mod)                        # device-side re-pack of the same NativeModel
    assert mod.native_differentiable(DEV) is nat
    fresh, _, _ = _siren_module("texture", 64, 5, precision=precision)
    fresh.load_state_dict(mod.state_dict())
    out_host, g_host = run(fresh)                    # fresh model: host pack
    assert np.abs(out_dev - out_host).max() <= 1e-5 * max(1.0, np.abs(out_host).max())
    for k in g_host:
        assert _rel_err(g_dev[k], g_host[k]) <= 1e-5, k
    print("[parity] device-side re-pack == host pack: forward max|diff| %.2e" % np.abs(out_dev - out_host).max())


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("kind,H,grid", [("texture", 64, 5), ("baseline", 32, 0), ("spatial", 32, 0), ("texture", 256, 6), ("texture", 96, 5), ("baseline", 192, 0)])
def test_native_repack_is_the_torch_repack_bit_for_bit(kind, H, grid, precision):
    """fenerf_model_repack (row scales, gathers, fp16 / bf16 hi-lo splits in four kernels, written in place) against the same
    re-pack spelled out in torch ops (NativeModel._pack_on_device, itself pinned to the host packer on the CPU): forward
    stream, consts and backward stream must be identical bit patterns."""
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=21, sigma_gain=10.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, precision, differentiable=True)
    gen = torch.Generator(device="cpu").manual_seed(5)
    params = {k: (torch.from_numpy(v) * (1.0 + 0.05 * torch.randn(v.shape, generator=gen))).float().to(DEV) for k, v in sd.items()}
    nat.load_from_device(params)
    got = nat.export_packed()
    want = nat._pack_on_device(params)[:3]
    for name, g, w in zip(("stream", "consts", "bwd"), got, want):
        assert g.numel() == w.numel(), name
        assert torch.equal(g.view(torch.int32), w.view(torch.int32)), f"{name}: {(g.view(torch.int32) != w.view(torch.int32)).sum().item()} words differ"
    # and the host packer agrees (but for the fp32- vs fp64-folded label rows)
    host = native.NativeModel({k: N_(v) for k, v in params.items()}, spec, DEV, precision, differentiable=True).export_packed()
    for name, g, w in zip(("stream", "consts", "bwd"), got, host):
        frac = (g.view(torch.int32) != w.view(torch.int32)).float().mean().item()
        assert frac <= (0.05 if spec["n_label_layers"] > 1 else 0.0), (name, frac)
    print(f"[parity] native re-pack {kind} H={H} {precision}: {sum(t.numel() for t in got)} words identical to the torch spelling")


def test_inversion_film_only_gradients_and_loop():
    """Inversion (inverse_render_double_semantic.py:306-410): with the generator's weights frozen only the FiLM gradients are
    computed (the chain kernel's per-tile FiLM sums, gathered and reduced); they must equal the FiLM gradients of the full backward, and the optimisation loop must
    reduce its loss when the target is a render of the same generator at shifted FiLM parameters."""
    from fenerf_amd import callers
    torch.manual_seed(7)          # the module's mapping networks are randomly initialised
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0)
    B, P = 2, 128
    rng = np.random.default_rng(9)
    pts, dirs = T(rng.uniform(-0.12, 0.12, (B, P, 3)).astype(np.float32)), T(rng.normal(size=(B, P, 3)).astype(np.float32))
    g_out = T(rng.normal(size=(B, P, 22)).astype(np.float32))
    film_np = proc.film_params(spec, B, seed=4)

    def film_grads(freeze):
        for p in mod.parameters():
            p.requires_grad_(not freeze)
            p.grad = None
        ft = {k: T(v).requires_grad_(True) for k, v in film_np.items()}
        out = mod.forward_with_frequencies_phase_shifts(pts, ft["freq_geo"], ft["freq_app"], ft["phase_geo"], ft["phase_app"], dirs)
        (out * g_out).sum().backward()
        return {k: N_(v.grad) for k, v in ft.items()}

    full, only = film_grads(False), film_grads(True)
    assert all(p.grad is None for p in mod.parameters())
    for k in full:      # (round 5: the full backward takes its FREQUENCY gradients from the weight-gradient sums, the FiLM-only one from the chain's own)
        assert _rel_err(only[k], full[k]) <= (1e-5 if "phase" in k else 2e-4), k
    # a FiLM-sum budget smaller than one image (round-3 advisory: launches were unbounded): the image is walked in point ranges of 128
    # whose FiLM gradients add -- equal to the single launch up to fp32 summation order
    from fenerf_amd.siren import autograd as SA
    P_big = 640
    pts_b, dirs_b = T(rng.uniform(-0.12, 0.12, (B, P_big, 3)).astype(np.float32)), T(rng.normal(size=(B, P_big, 3)).astype(np.float32))
    g_b = T(rng.normal(size=(B, P_big, 22)).astype(np.float32))
    res = []
    for budget in (SA.FILM_SUMS_BUDGET_BYTES, 1):
        old, SA.FILM_SUMS_BUDGET_BYTES = SA.FILM_SUMS_BUDGET_BYTES, budget
        try:
            ft = {k: T(v).requires_grad_(True) for k, v in film_np.items()}
            out = mod.forward_with_frequencies_phase_shifts(pts_b, ft["freq_geo"], ft["freq_app"], ft["phase_geo"], ft["phase_app"], dirs_b)
            (out * g_b).sum().backward()
            res.append({k: N_(v.grad) for k, v in ft.items()})
        finally:
            SA.FILM_SUMS_BUDGET_BYTES = old
    worst = max(_rel_err(res[1][k], res[0][k]) for k in res[0])
    print(f"[parity] inversion with a FiLM-sum budget below one image (5 point ranges of 128 per image) vs one launch: {worst:.1e}")
    assert worst <= 2e-5

    torch.manual_seed(3)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren.spatial_embeddings = torch.nn.Parameter(mod.spatial_embeddings.detach().clone())
    gen.siren.load_state_dict(mod.state_dict(), strict=True)
    gen = gen.to(DEV).eval()
    for p in gen.parameters():
        p.requires_grad_(False)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    opts = dict(img_size=16, fov=12, ray_start=0.88, ray_end=1.12, num_steps=12, h_stddev=0, v_stddev=0, h_mean=np.pi / 2, v_mean=np.pi / 2,
                hierarchical_sample=True, sample_dist=None, clamp_mode="relu", nerf_noise=0, last_back=False)

    class FixedDraws(VR.TorchDraws):      # no stratified jitter / resampling randomness: the render is a function of the FiLM
        def rand(self, shape, device):    # parameters only, so the loss measures the optimisation, not per-render sampling noise
            return torch.full(shape, 0.5, device=device)
        def randn(self, shape, device):
            return torch.zeros(shape, device=device)
    gen.draws = FixedDraws()
    with torch.no_grad():     # target: a truncated sample (psi = 0.5 around the mean FiLM parameters the inversion starts from)
        zs = torch.randn(4000, 8, device=DEV)
        mfg, mpg = (t.mean(0, keepdim=True) for t in gen.siren.geo_mapping_network(zs))
        mfa, mpa = (t.mean(0, keepdim=True) for t in gen.siren.app_mapping_network(zs))
        fg, pg = gen.siren.geo_mapping_network(torch.randn(1, 8, device=DEV))
        fa, pa = gen.siren.app_mapping_network(torch.randn(1, 8, device=DEV))
        mix = lambda m, r: m + 0.5 * (r - m)
        target, _ = gen.forward_with_frequencies(mix(mfg, fg), mix(mfa, fa), mix(mpg, pg), mix(mpa, pa), **opts)
    res = callers.inverse_render(gen, target[:, -3:], target[:, :-3], opts, n_iterations=80, z_dim=8, latent_noise=0.0)
    first, last = np.mean(res["losses"][:5]), np.mean(res["losses"][-5:])
    print(f"[parity] inversion loop: loss {first:.4e} -> {last:.4e} over 80 native differentiable renders")
    assert last < 0.9 * first
    assert res["w_geo_frequency_offsets"].abs().max() > 0


@pytest.mark.parametrize("precision", PRECISIONS)
def test_inversion_reproduces_the_references_own_trajectory(precision):
    """tests/golden/tiny_texture_inversion.npz (round 5): the REFERENCE's inversion loop (inverse_render_double_semantic.py:306-410,
    restated on the reference's generator by tools/make_golden.py::run_inversion_case: Adam lr 1e-2 / weight_decay 1e-4 on the four
    offset tensors, StepLR(100, 0.75), annealed latent noise, the script's `options`, both MSE terms) on the frozen generator of
    tiny_texture_trained_state.npz -- weights the reference's own forward + autograd + Adam produced --, with every draw, the loss of every
    iteration and the four offset tensors after every iteration recorded.  callers.inverse_render on the same draws (30 native
    differentiable renders, FiLM-only backward) must walk the same trajectory: replaces round 4's "the loss falls by 10 %".
    Conditioning, measured with the reference itself in the build container: a 1e-5 relative change of the initial mean frequencies moves
    its losses by <= 2e-4 relative and its offsets by <= 7e-5 over the 30 iterations (1e-6: 5e-5 / 1.3e-5)."""
    from fenerf_amd import callers
    g = load_golden("tiny_texture_inversion")
    spec = spec_from_golden(g)
    gen = _make_generator(g, spec, precision)
    for p in gen.parameters():
        p.requires_grad_(False)
    n, S_, N = int(g["meta_iterations"]), int(g["meta_S"]), int(g["meta_N"])
    gen.draws = VR.RecordedDraws([g[k] for k in sorted(k for k in g if k.startswith("draw"))])
    # the script's options dict (:220-243) with its sizes scaled down, keys it never reads included
    options = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0, v_stddev=0,
                   h_mean=torch.tensor(np.pi / 2, device=DEV), v_mean=torch.tensor(np.pi / 2, device=DEV), hierarchical_sample=False,
                   sample_dist=None, clamp_mode="relu", nerf_noise=0, fade_steps=10000, z_app_lambda=0, z_geo_lambda=0, pos_lambda=0,
                   tok_interval=2000, tok_v=0.6, betas=(0, 0.9), fill_mode="eval_seg_padding_background")
    res = callers.inverse_render(gen, T(g["gt_image"]), T(g["gt_seg"]), options, n_iterations=n, z_dim=spec["z_dim"],
                                 n_mean_latents=int(g["meta_mean_latents"]), record_offsets=True)
    assert not gen.draws.arrays, "all recorded draws consumed, in order"
    for k in ("w_geo_frequencies", "w_geo_phase_shifts", "w_app_frequencies", "w_app_phase_shifts"):
        np.testing.assert_allclose(N_(res[k]), g[k], atol=2e-6)       # mean FiLM parameters over the recorded latents (torch mapping networks)
    losses = np.asarray(res["losses"])
    rel = np.abs(losses - g["losses"]) / g["losses"]
    off = np.zeros(n)
    for j, nm in enumerate(("geo_frequency", "geo_phase_shift", "app_frequency", "app_phase_shift")):
        ref = g[f"offsets_{nm}"]
        got = np.stack([h[j].numpy() for h in res["offset_history"]])
        assert got.shape == ref.shape
        off = np.maximum(off, np.abs(got - ref).reshape(n, -1).max(1))
    print(f"[parity] inversion trajectory {precision}: reference loss {g['losses'][0]:.5f} -> {g['losses'][-1]:.5f} over {n} iterations; native "
          f"loss within {rel[:20].max():.1e} relative over the first 20 iterations ({rel.max():.1e} over all), offsets within {off[:20].max():.1e} "
          f"/ {off.max():.1e} absolute (|offset| up to {max(np.abs(g['offsets_' + nm]).max() for nm in ('geo_frequency', 'geo_phase_shift', 'app_frequency', 'app_phase_shift')):.3f})")
    # the review's bar: 1e-3 relative on the loss over the first 20 iterations, 1e-3 absolute on the offsets.  Measured (round 5): loss within
    # 3.1e-5 (f32) / 4.4e-5 (f16x3) over ALL 30 iterations, offsets within 1.5e-5 / 2.1e-5 -- asserted at about twice that
    assert rel.max() <= 1e-4 and off.max() <= 5e-5
    assert losses[-1] < 0.5 * losses[0]


@pytest.mark.parametrize("kind,H,grid,B,P", [("texture", 32, 5, 2, 96), ("baseline", 64, 0, 1, 64), ("spatial", 32, 0, 2, 160), ("texture", 128, 4, 3, 160),
                                             ("texture", 256, 6, 2, 224), ("texture", 256, 6, 1, 4224), ("texture", 256, 6, 2, 2080),
                                             ("texture", 96, 5, 2, 224), ("texture", 192, 4, 2, 224)])
def test_16bit_tape_against_the_fp32_tape(kind, H, grid, B, P):
    """Round 5 (include/fenerf.h FENERF_TAPE_U16, siren.grad_precision = "tape16"): the tape between forward and backward holds frac(theta) as
    16-bit fixed point instead of the fp32 accumulators.  Same model, same inputs, both tapes:
      * the forward-save outputs are bit-identical (the tape is a by-product; the activations the forward carries on are the exact ones);
      * the tape is half as large;
      * every gradient agrees with the fp32-tape one to the quantisation's 1e-4 class -- the FiLM FREQUENCY gradients included, which the
        16-bit route derives from the weight-gradient partial sums and the weights (sum_p d theta (W x + b) = sum_k W G + b sum_p d theta)
        instead of the tape's accumulator; the phase gradients, which only see the quantised cosines, likewise;
      * FiLM-only backward passes (inversion) of a "tape16" module keep the fp32 tape: bit-identical to the default module's;
      * the C-ABI refuses the combinations that cannot work (exact-fp32 model, FiLM-only gradients from a 16-bit tape, no weights).
    Shapes: 16- and 128-point FiLM-sum units (P a multiple of 128 or not), several images, every H the kernels instantiate."""
    rng = np.random.default_rng(23)
    pts = T(rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32))
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs = T(dirs / np.linalg.norm(dirs, axis=-1, keepdims=True))
    g_out = rng.normal(size=(B, P, 4 if kind == "spatial" else 22)).astype(np.float32)
    g_out[..., -1] *= 0.02
    g_out = T(g_out)
    res = {}
    for gp in ("f32", "tape16"):
        mod, spec, sd = _siren_module(kind, H, grid, precision="f16x3")
        mod.grad_precision = gp
        film = proc.film_params(spec, B, seed=4)
        if kind == "spatial":
            film["freq_app"] = proc.normal("film.freq_app", (B, H), 0.4, 4)
            film["phase_app"] = proc.normal("film.phase_app", (B, H), 0.4, 4)
        ft = {k: T(v).requires_grad_(True) for k, v in film.items()}
        if kind == "spatial":
            out = mod.forward_with_frequencies_phase_shifts(pts, torch.cat([ft["freq_geo"], ft["freq_app"]], -1), torch.cat([ft["phase_geo"], ft["phase_app"]], -1), dirs)
        else:
            out = mod.forward_with_frequencies_phase_shifts(pts, ft["freq_geo"], ft["freq_app"], ft["phase_geo"], ft["phase_app"], dirs)
        nat = mod.native_differentiable(DEV)
        fmt = mod.tape_format(nat, film_only=False)
        assert fmt == (_lib.TAPE_U16 if gp == "tape16" else _lib.TAPE_F32_W)
        (out * g_out).sum().backward()
        g = {k: N_(v.grad) for k, v in ft.items()}
        g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
        res[gp] = (N_(out), g, nat.tape_words_per_point(fmt), nat.tape_floats(B * ((P + 31) // 32 * 32), fmt))
        if gp == "tape16":      # inversion on the same module: FiLM gradients only -> the fp32 tape, whatever grad_precision says
            assert mod.tape_format(nat, film_only=True) == _lib.TAPE_F32
    (o32, g32, w32, f32), (o16, g16, w16, f16) = res["f32"], res["tape16"]
    assert np.array_equal(o32, o16), "the forward-save outputs do not depend on the tape's format"
    assert 2 * w16 == w32 and 2 * f16 == f32
    assert g32.keys() == g16.keys()
    errs = {k: _rel_err(g16[k], g32[k]) for k in g32}
    worst = max(errs, key=errs.get)
    freq = max(errs[k] for k in errs if k.startswith("freq_"))
    print(f"[parity] 16-bit tape vs fp32 tape, {kind} H={H} B={B} P={P}: worst relative difference over {len(errs)} gradient tensors {errs[worst]:.2e} ({worst}); "
          f"FiLM frequency gradients (from the weight-gradient sums) {freq:.2e}; tape {w16 * 4} instead of {w32 * 4} bytes per point")
    assert errs[worst] <= 3e-4, errs


@pytest.mark.parametrize("kind,H,grid,B,P", [("texture", 32, 5, 2, 96), ("baseline", 64, 0, 1, 64), ("spatial", 32, 0, 2, 160), ("texture", 256, 6, 2, 224),
                                             ("texture", 256, 6, 1, 4224), ("texture", 96, 5, 2, 224), ("baseline", 192, 0, 2, 224)])     # H = 96 / 192: rows that
                                             # do not tile the reduction's workgroups (the round-5 first cut of that kernel was wrong there)
def test_frequency_gradients_from_the_weight_gradient_sums(kind, H, grid, B, P):
    """include/fenerf.h FENERF_TAPE_F32_W (round 5, the default of f16x3 models): the chain kernel no longer forms sum_p d theta * tape -- a
    multiply and a 16-lane butterfly per row tile, a fifth of its VALU instructions --; the FiLM frequency gradient sum_p d theta (W x + b) is
    formed by the weight-gradient reductions as sum_k W[n][k] G[n][k] + b[n] sum_p d theta[n] from the per-image partial sums G they hold
    anyway.  Against the rounds-2-4 route (siren.FREQ_FROM_WGRAD = False: FENERF_TAPE_F32) on the same model and inputs: the forward, the
    tape and EVERY other gradient are bit-identical (same kernels, same instructions: only the second FiLM sum is gone); the frequency
    gradients agree to the bf16x3 products' 2^-17 class."""
    rng = np.random.default_rng(29)
    pts = T(rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32))
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs = T(dirs / np.linalg.norm(dirs, axis=-1, keepdims=True))
    g_out = rng.normal(size=(B, P, 4 if kind == "spatial" else 22)).astype(np.float32)
    g_out[..., -1] *= 0.02
    g_out = T(g_out)
    res = {}
    for from_wgrad in (False, True):
        mod, spec, sd = _siren_module(kind, H, grid, precision="f16x3")
        mod.FREQ_FROM_WGRAD = from_wgrad
        film = proc.film_params(spec, B, seed=4)
        if kind == "spatial":
            film["freq_app"] = proc.normal("film.freq_app", (B, H), 0.4, 4)
            film["phase_app"] = proc.normal("film.phase_app", (B, H), 0.4, 4)
        ft = {k: T(v).requires_grad_(True) for k, v in film.items()}
        if kind == "spatial":
            out = mod.forward_with_frequencies_phase_shifts(pts, torch.cat([ft["freq_geo"], ft["freq_app"]], -1), torch.cat([ft["phase_geo"], ft["phase_app"]], -1), dirs)
        else:
            out = mod.forward_with_frequencies_phase_shifts(pts, ft["freq_geo"], ft["freq_app"], ft["phase_geo"], ft["phase_app"], dirs)
        assert mod.tape_format(mod.native_differentiable(DEV), film_only=False) == (_lib.TAPE_F32_W if from_wgrad else _lib.TAPE_F32)
        (out * g_out).sum().backward()
        g = {k: N_(v.grad) for k, v in ft.items()}
        g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
        res[from_wgrad] = (N_(out), g)
    (o0, g0), (o1, g1) = res[False], res[True]
    assert np.array_equal(o0, o1) and g0.keys() == g1.keys()
    worst = 0.0
    for k in g0:
        if k.startswith("freq_"):
            worst = max(worst, _rel_err(g1[k], g0[k]))
        elif k == "spatial_embeddings":
            assert _rel_err(g1[k], g0[k]) <= 1e-6, k        # float atomics: unordered sum
        else:
            assert np.array_equal(g0[k], g1[k]), (k, _rel_err(g1[k], g0[k]))
    print(f"[parity] FiLM frequency gradients from the weight-gradient sums vs from the chain's own sums, {kind} H={H} B={B} P={P}: {worst:.2e}; "
          f"every other gradient ({len(g0) - 2} tensors) bit-identical")
    assert worst <= 1e-4


def test_16bit_tape_api_refuses_what_cannot_work():
    mod, spec, sd = _siren_module("texture", 32, 5, precision="f32")
    nat32 = native.NativeModel(sd, spec, DEV, "f32", differentiable=True)
    nat16 = native.NativeModel(sd, spec, DEV, "f16x3", differentiable=True)
    B, P = 1, 64
    rng = np.random.default_rng(3)
    pts, dirs = T(rng.uniform(-0.1, 0.1, (B, P, 3)).astype(np.float32)), T(rng.normal(size=(B, P, 3)).astype(np.float32))
    film = proc.film_params(spec, B, seed=4)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    with pytest.raises(_lib.FenerfError, match="FENERF_PREC_F16X3"):
        nat32.siren_forward_save(pts, dirs, *tf, tape_format=_lib.TAPE_U16)
    out, tape, tape_e = nat16.siren_forward_save(pts, dirs, *tf, tape_format=_lib.TAPE_U16)
    d_out = torch.ones_like(out)
    d_grid = torch.zeros(tuple(nat16.grid_shape) + (32,), device=DEV)
    d_t = nat16.siren_backward_grid(B, P, *tf, out, d_out, tape, pts, d_grid, tape_format=_lib.TAPE_U16)
    with pytest.raises(_lib.FenerfError, match="weights"):
        nat16.siren_param_grads(pts, dirs, *tf, out, d_out, tape, tape_e, d_t, tape_format=_lib.TAPE_U16, weights=None)
    w = ([torch.zeros(32, 3, device=DEV)] + [torch.zeros(32, 32, device=DEV)] * 7, [torch.zeros(32, 3 + 32 + 32, device=DEV)] + [torch.zeros(32, 32, device=DEV)] * 2)
    with pytest.raises(_lib.FenerfError, match="FiLM-only"):
        nat16.siren_param_grads(pts, dirs, *tf, out, d_out, tape, tape_e, d_t, film_only=True, tape_format=_lib.TAPE_U16, weights=w)
    with pytest.raises(_lib.FenerfError, match="FiLM-only"):
        nat16.siren_param_grads(pts, dirs, *tf, out, d_out, tape, tape_e, d_t, film_only=True, tape_format=_lib.TAPE_F32_W, weights=w)
    with pytest.raises(_lib.FenerfError, match="unknown tape format"):
        nat16.siren_forward_save(pts, dirs, *tf, tape_format=7)


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
@pytest.mark.parametrize("case", ["aligned", "ragged_chunks", "film_only", "lock_view_baseline"])
def test_render_backward_abi_call_equals_the_python_orchestration(case, precision):
    """fenerf_render_forward_save / fenerf_render_backward (round 5) -- the differentiable hierarchical render as two C-ABI calls
    (FiLM pre-pass, sample points, both forward-save passes, weights + resampling, merged composite | composite backward, chunk plan, chain
    and weight-gradient launches per chunk, gradient sums in chunk order, the fold of the two passes' FiLM gradients, the grid gradient in
    the parameter's layout) against rounds 2-4's Python orchestration of the same kernels (generators/autograd.py::_hierarchical_forward,
    siren/autograd.py::chunked_backward): pixels and every gradient BIT-IDENTICAL.  Cases: whole tiles per image in one chunk; 539 points per
    image (padded to 544) in 128-point chunks and in whole-image chunks (several launches, FiLM rows added over point ranges); FiLM-only
    (inversion: frozen weights; f16x3 models walk their FiLM-sum budget); a model without a grid with a locked view direction."""
    from fenerf_amd.siren import autograd as SA
    from fenerf_amd.generators import autograd as GA
    kind, H, grid = ("baseline", 64, 0) if case == "lock_view_baseline" else ("texture", 32, 5)
    mod, spec, sd = _siren_module(kind, H, grid, sigma_gain=150.0, precision=precision)
    cls = S.SIRENBASELINESEMANTICDISENTANGLE if kind == "baseline" else S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE
    gen = G.DoubleImplicitGenerator3d(functools.partial(cls, hidden_dim=H), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N = (2, 8, 8) if case == "aligned" else (2, 7, 11)       # 512 | 539 (-> 544) points per image and pass
    film = proc.film_params(spec, B, seed=4)
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
              hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False,
              lock_view_dependence=case == "lock_view_baseline")
    if case == "film_only":
        for p_ in mod.parameters():
            p_.requires_grad_(False)
    settings = [(SA.BACKWARD_CHUNK_POINTS, SA.FILM_SUMS_BUDGET_BYTES)]
    if case == "ragged_chunks":
        settings = [(128, SA.FILM_SUMS_BUDGET_BYTES), (1700, SA.FILM_SUMS_BUDGET_BYTES)]
    if case == "film_only":
        settings.append((SA.BACKWARD_CHUNK_POINTS, 1))       # a FiLM-sum budget below one image: walked in 128-point ranges
    worst_name = None
    for chunk, budget in settings:
        res = []
        old = (SA.BACKWARD_CHUNK_POINTS, SA.FILM_SUMS_BUDGET_BYTES, GA.USE_RENDER_ABI)
        SA.BACKWARD_CHUNK_POINTS, SA.FILM_SUMS_BUDGET_BYTES = chunk, budget
        try:
            for abi in (False, True):
                GA.USE_RENDER_ABI = abi
                film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
                for p_ in mod.parameters():
                    p_.grad = None
                torch.manual_seed(11)
                with native.phase_timing() as t:
                    px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
                    w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
                    (px * w).sum().backward()
                g = {k: N_(v.grad) for k, v in film_t.items()}
                g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
                res.append((N_(px), g, dict(t.calls)))
        finally:
            SA.BACKWARD_CHUNK_POINTS, SA.FILM_SUMS_BUDGET_BYTES, GA.USE_RENDER_ABI = old
        (px_py, g_py, calls_py), (px_abi, g_abi, calls_abi) = res
        assert np.array_equal(px_py, px_abi), "pixels"
        assert g_py.keys() == g_abi.keys() and len(g_py) >= 4
        if case != "film_only":
            assert len(g_py) > 30
        for k in g_py:
            if k == "spatial_embeddings":      # the chain kernel scatters the grid gradient with float atomics: the sum's order varies from run to run
                assert _rel_err(g_abi[k], g_py[k]) <= 1e-6, (k, _rel_err(g_abi[k], g_py[k]))
            else:
                assert np.array_equal(g_py[k], g_abi[k]), (k, _rel_err(g_abi[k], g_py[k]))
        assert calls_abi.get("chain", 0) == calls_py.get("chain", 0) >= (2 if case == "ragged_chunks" else 1), (calls_py, calls_abi)
    print(f"[parity] fenerf_render_forward_save + fenerf_render_backward vs the Python orchestration [{case}, {precision}]: pixels and {len(g_py)} gradient "
          f"tensors bit-identical; chain launches per step {calls_abi.get('chain', 0)}")


def test_reduced_precision_forward_modes():
    """include/fenerf.h fenerf_model_set_forward_mode (round 5, opt-in): "f16x2" -- two fp16 MFMAs per product everywhere, the weights as one
    fp16 value -- and "f16x3c2" -- three terms through the geometry trunk and the label / sigma head, two in the colour layers and the rgb head.
    Neither meets the default's asserted bounds (they are reported beside it: bench.py legs, tools/forward_mode_report.py ->
    profiles/r05_forward_modes.md); what IS asserted: "f16x3c2" leaves sigma and the labels bit-identical to the default (so coarse weights,
    resampled depths and fill decisions are the default's) and moves rgb by no more than 2e-4; "f16x2" stays within 1e-3 on rgb / labels and 1e-3
    relative on sigma; a differentiable evaluation of a module set to either mode runs the default arithmetic; exact-fp32 handles refuse."""
    g = load_golden("h256_texture_16x16_n12")
    spec, sd = _weights_for("h256_texture_16x16_n12")
    film, tf = _film(g, spec)
    B, R, N = g["st_z_coarse"].shape[:3]
    pts = T(g["st_points"].reshape(B, R * N, 3))
    dirs = T(np.broadcast_to(g["st_dirs"][:, :, None, :], (B, R, N, 3)).reshape(B, R * N, 3).copy())
    outs = {}
    for mode in ("f16x3", "f16x3c2", "f16x2"):
        nat = native.NativeModel(sd, spec, DEV, mode)
        assert nat.precision == "f16x3" and nat.forward_mode == {"f16x3": 0, "f16x2": 1, "f16x3c2": 2}[mode]
        outs[mode] = N_(nat.siren_forward(pts, dirs, *tf))
        nat.close()
    ref = outs["f16x3"]
    assert np.array_equal(outs["f16x3c2"][..., :-4], ref[..., :-4]) and np.array_equal(outs["f16x3c2"][..., -1], ref[..., -1]), "labels and sigma of f16x3c2"
    e_c2 = np.abs(outs["f16x3c2"][..., -4:-1] - ref[..., -4:-1]).max()
    d2 = np.abs(outs["f16x2"] - ref)
    smax = np.abs(ref[..., -1]).max()
    print(f"[parity] reduced-precision forwards vs f16x3 (H=256 + 96^3, {B * R * N} points): f16x3c2 rgb {e_c2:.2e} (labels, sigma bit-identical); "
          f"f16x2 rgb {d2[..., -4:-1].max():.2e} labels {d2[..., :-4].max():.2e} sigma {d2[..., -1].max():.2e} (|sigma| max {smax:.3g})")
    assert 0 < e_c2 <= 2e-4
    assert 0 < d2[..., -4:-1].max() <= 1e-3 and d2[..., :-4].max() <= 1e-3 and d2[..., -1].max() <= 1e-3 * max(smax, 1.0)
    # The default's asserted bounds have teeth (round-4 review #5: "a deliberately perturbed build trips at least one bound"): against the
    # reference's own outputs of this fixture, with test_siren_forward_vs_reference's bounds (rgb 8.5e-7, labels 9e-8, sigma 1.1e-5 |sigma|max),
    # the default passes all three, one dropped compensation term in the colour branch (f16x3c2) trips the rgb bound and only it, one
    # dropped everywhere (f16x2) trips all three.
    want = g["st_siren_coarse"]
    wmax = max(float(np.abs(want[..., -1]).max()), 0.1)
    trips = {}
    for mode, o_ in outs.items():
        e = [float(np.abs(o_[..., sl] - want[..., sl]).max()) for sl in (slice(-4, -1), slice(None, -4), slice(-1, None))]
        trips[mode] = (e[0] > 8.5e-7, e[1] > 9e-8, e[2] > 1.1e-5 * wmax)
        print(f"[parity] forward mode {mode} vs the reference's outputs: rgb {e[0]:.2e} labels {e[1]:.2e} sigma {e[2]:.2e} -> bounds tripped (rgb, labels, sigma) {trips[mode]}")
    assert trips["f16x3"] == (False, False, False) and trips["f16x3c2"] == (True, False, False) and trips["f16x2"] == (True, True, True)
    with pytest.raises(_lib.FenerfError, match="FENERF_PREC_F16X3"):
        n32 = native.NativeModel(sd, spec, DEV, "f32")
        if _lib.lib().fenerf_model_set_forward_mode(n32._h, 1) < 0:
            raise _lib.FenerfError(-4, _lib.lib().fenerf_last_error().decode())
    mod, spec2, sd2 = _siren_module("texture", 32, 5, precision="f16x2")
    assert mod.native(DEV).forward_mode == 1 and mod.native_differentiable(DEV).forward_mode == 0


def test_single_latent_generator_gradient_nonhierarchical_locked_view():
    """ImplicitGenerator3d.forward with grad: hierarchical_sample=False (CompositeFunction), lock_view_dependence=True (the kernels
    substitute the constant view direction (0,0,-1), siren.py:1515 / generators.py:474-476), white_back -- gradients of a pixel
    loss wrt the mapped FiLM parameters and the render weights vs fp64 autograd on the same rays and noise."""
    from oracle import fenerf_oracle_grad as OG
    mod, spec, sd = _siren_module("spatial", 32, 0, sigma_gain=120.0)
    gen = G.ImplicitGenerator3d(functools.partial(S.SPATIALSIRENBASELINE, hidden_dim=32), 8, 4)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N, H = 2, 5, 9, 32        # R*N = 225: not a multiple of 32 -> the padded-tile path
    film = proc.film_params(spec, B, seed=4)
    film["freq_app"] = proc.normal("film.freq_app", (B, H), 0.4, 4)
    film["phase_app"] = proc.normal("film.phase_app", (B, H), 0.4, 4)
    freq = T(np.concatenate([film["freq_geo"], film["freq_app"]], -1)).requires_grad_(True)
    phase = T(np.concatenate([film["phase_geo"], film["phase_app"]], -1)).requires_grad_(True)
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=False, sample_dist="gaussian", clamp_mode="softplus", nerf_noise=0.3, white_back=True,
              lock_view_dependence=True)
    torch.manual_seed(21)
    px, _ = gen.forward_with_frequencies(freq, phase, **kw)
    assert px.requires_grad and px.shape == (B, 3, S_, S_)
    w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    (px * w).sum().backward()

    torch.manual_seed(21)
    R = S_ * S_
    origins, dirs, z_vals, _, _ = VR.sample_rays(B, N, gen.device, kw["fov"], (S_, S_), kw["ray_start"], kw["ray_end"], kw["h_stddev"],
                                                 kw["v_stddev"], kw["h_mean"], kw["v_mean"], kw["sample_dist"], draws=gen.draws)
    noise_f = gen.draws.randn((B, R, N, 1), gen.device)
    z_c = z_vals.reshape(B, R, N)
    pts = (origins.unsqueeze(2) + dirs.unsqueeze(2) * z_c.unsqueeze(-1)).reshape(B, R * N, 3)
    t64 = lambda a: torch.as_tensor(N_(a) if torch.is_tensor(a) else np.asarray(a), dtype=torch.float64)
    sd64 = {k: t64(v).requires_grad_(True) for k, v in sd.items()}
    f64, p64 = t64(freq).requires_grad_(True), t64(phase).requires_grad_(True)
    locked = torch.zeros((B, R * N, 3), dtype=torch.float64); locked[..., 2] = -1
    out = OG.siren_forward(sd64, spec, t64(pts), locked, f64[:, :8 * H], p64[:, :8 * H], f64[:, 8 * H:], p64[:, 8 * H:])
    rgb, _, _ = OG.composite(out.reshape(B * R, N, 4), t64(z_c.reshape(B * R, N)), t64(noise_f.reshape(B * R, N)), noise_std=0.3,
                             clamp_mode="softplus", white_back=True)
    ref_px = rgb.reshape(B, S_, S_, 3).permute(0, 3, 1, 2) * 2 - 1
    (ref_px * t64(w)).sum().backward()
    assert np.abs(N_(px) - ref_px.detach().numpy()).max() <= 1e-3
    worst = max(_rel_err(N_(freq.grad), f64.grad.numpy()), _rel_err(N_(phase.grad), p64.grad.numpy()))
    named = dict(mod.named_parameters())
    for k, v in sd64.items():
        worst = max(worst, _rel_err(N_(named[k].grad), v.grad.numpy()))
    print(f"[parity] single-latent generator gradient (no resampling, locked view, white_back): worst relative error {worst:.2e}")
    assert worst <= 5e-4


def test_part_forward_gradient_on_a_ray_subset():
    """generator.forward(..., grad_points=G) = part_forward (generators.py:858-910): a random subset of G rays is rendered
    with gradient, the rest without, and the pixels are scattered back.  Without resampling and noise every ray is independent
    of the draws, so the image must equal the plain render and the gradients must equal those of the plain differentiable
    render with the loss masked to the subset."""
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N, Gp = 2, 8, 12, 20
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.0, v_stddev=0.0, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=False, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.0)
    z = torch.randn(B, 8, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    w = torch.randn((B, 21, S_, S_), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))

    class Draws(VR.TorchDraws):           # fixed jitter and permutation so both renders see the same rays
        def __init__(self):
            self.g = torch.Generator(device=DEV).manual_seed(9)
            self.perm = torch.randperm(S_ * S_, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
        def rand(self, shape, device):
            return torch.full(shape, 0.5, device=device)
        def randn(self, shape, device):
            return torch.zeros(shape, device=device)
        def randperm(self, n, device):
            return self.perm
    gen.draws = Draws()

    def run(**extra):
        for p in gen.parameters():
            p.grad = None
        px, _ = gen(z, z, **kw, **extra)
        return px

    px_part = run(grad_points=Gp)
    mask = torch.zeros(S_ * S_, device=DEV)
    mask[gen.draws.perm[:Gp]] = 1
    (px_part * w).sum().backward()
    g_part = {n: N_(p.grad) for n, p in gen.named_parameters() if p.grad is not None}
    px_full = run()
    (px_full * w * mask.reshape(1, 1, S_, S_)).sum().backward()
    g_full = {n: N_(p.grad) for n, p in gen.named_parameters() if p.grad is not None}
    assert np.abs(N_(px_part) - N_(px_full)).max() <= 2e-6
    assert set(g_part) == set(g_full)
    worst = max(_rel_err(g_part[k], g_full[k]) for k in g_full if np.abs(g_full[k]).max() > 0)
    print(f"[parity] part_forward ({Gp} of {S_ * S_} rays differentiable): image max|diff| {np.abs(N_(px_part) - N_(px_full)).max():.1e}, "
          f"gradients vs masked full render worst relative diff {worst:.1e}")
    assert worst <= 2e-4
    with torch.no_grad():
        assert np.abs(N_(run(grad_points=Gp)) - N_(px_full)).max() <= 2e-6      # no-grad part_forward renders the same image
        # point_forward on explicit samples of every ray == the same image (per-point view directions, stage-by-stage path)
        o, dd, zz, _, _ = VR.sample_rays(B, N, gen.device, kw["fov"], (S_, S_), kw["ray_start"], kw["ray_end"], 0.0, 0.0, np.pi / 2,
                                         np.pi / 2, "gaussian", draws=gen.draws)
        pts = o.unsqueeze(2) + dd.unsqueeze(2) * zz.unsqueeze(-1)
        pf = gen.point_forward(pts, dd.unsqueeze(2).expand(-1, -1, N, -1), o, dd, zz.unsqueeze(-1), z, z, N, False, **{k: kw[k] for k in ("clamp_mode", "nerf_noise")})
        img = pf.reshape(B, S_, S_, -1).permute(0, 3, 1, 2) * 2 - 1
        assert np.abs(N_(img) - N_(px_full)).max() <= 2e-6


# Sizes that give every workgroup of the 16-point kernels more than one oct of tiles (256 CUs x 128 points): the stream wraps for
# the next tile, the ring slot counter and the tape-buffer parity carry over, the last oct is ragged -- and, last, the generator
# step's own shape: H = 256, 393,216 points of one image (one pass of the 128 x 128 x 24 image; the production chunk is both passes: 786,432).  Checked against
# torch autograd of the fp64 restatement (oracle/fenerf_oracle_grad.py, pinned to the reference's autograd), which walks the points
# in slabs of 32,768 (every gradient is a sum over points) to bound the host memory.
_FP64_BACKWARD_REFERENCE = {}


@pytest.mark.parametrize("precision,H,grid,B,P", [("f16x3", 32, 5, 1, 40000), ("f32", 32, 5, 1, 40000), ("tape16", 32, 5, 1, 40000),
                                                  ("f16x3", 64, 0, 2, 33024), ("tape16", 64, 0, 2, 33024),
                                                  ("f16x3", 256, 6, 1, 65536), ("f32", 256, 6, 1, 65536), ("amp", 256, 6, 1, 65536),
                                                  ("tape16", 256, 6, 1, 65536), ("amp16", 256, 6, 1, 65536),
                                                  ("f16x3", 256, 6, 1, 393216), ("amp", 256, 6, 1, 393216), ("tape16", 256, 6, 1, 393216),
                                                  ("f16x3", 96, 5, 1, 40000), ("f32", 96, 5, 1, 40000), ("tape16", 96, 5, 1, 40000),      # round 5: H = 96 / 192
                                                  ("f16x3", 192, 4, 1, 66000), ("amp", 192, 4, 1, 66000), ("amp16", 192, 4, 1, 66000)])
def test_siren_backward_at_scale_vs_fp64_autograd(precision, H, grid, B, P):
    # "amp" = f16x3 with the opt-in AMP-class weight-gradient operands (bf16, one MFMA per product; siren.grad_precision)
    # "tape16" = f16x3 with the opt-in 16-bit tape (frac(theta) as fixed point between forward and backward): the tier between the two
    # "amp16" = both (everything that passes through the dump OR the tape is then AMP class: the FiLM frequency gradients too)
    amp16 = precision == "amp16"
    amp, t16 = precision in ("amp", "amp16"), precision == "tape16"
    precision = "f16x3" if (amp or t16) else precision
    from oracle import fenerf_oracle_grad as OG
    from fenerf_amd.siren import autograd as SA
    kind = "texture" if grid else "baseline"
    mod, spec, sd = _siren_module(kind, H, grid, precision=precision)
    mod.grad_precision = "amp16" if amp16 else ("amp" if amp else ("tape16" if t16 else "f32"))
    rng = np.random.default_rng(17)
    pts = rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32)
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, B, seed=4)
    Cc = spec["output_dim"]
    g_out = rng.normal(size=(B, P, Cc)).astype(np.float32)
    g_out[..., -1] *= 0.02
    film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
    out = mod.forward_with_frequencies_phase_shifts(T(pts), film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], T(dirs))
    (out * T(g_out)).sum().backward()
    nchunks = -(-P // SA.BACKWARD_CHUNK_POINTS) * B if P > SA.BACKWARD_CHUNK_POINTS else -(-B // max(1, SA.BACKWARD_CHUNK_POINTS // P))

    # the fp64 reference depends on the shape only (weights, inputs and upstream gradient are seeded): computed once per shape, shared
    # by the precisions that are checked against it (65 s of host time at 393,216 points)
    key = (kind, H, grid, B, P)
    if key not in _FP64_BACKWARD_REFERENCE:
        t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
        sd64 = {k: t64(v).requires_grad_(True) for k, v in sd.items()}
        film64 = {k: t64(v).requires_grad_(True) for k, v in film.items()}
        ref_out = np.empty((B, P, Cc), np.float64)
        for s0 in range(0, P, 32768):
            sl = slice(s0, min(P, s0 + 32768))
            ref = OG.siren_forward(sd64, spec, t64(pts[:, sl]), t64(dirs[:, sl]), film64["freq_geo"], film64["phase_geo"], film64["freq_app"],
                                   film64["phase_app"])
            (ref * t64(g_out[:, sl])).sum().backward()           # .grad accumulates over the slabs
            ref_out[:, sl] = ref.detach().numpy()
        _FP64_BACKWARD_REFERENCE.clear()                         # one shape at a time: the parametrisation groups equal shapes
        _FP64_BACKWARD_REFERENCE[key] = (ref_out, {k: film64[k].grad.numpy() for k in film}, {k: v.grad.numpy() for k, v in sd64.items()})
    ref_out, film_ref, sd_ref = _FP64_BACKWARD_REFERENCE[key]
    fwd_err = float(np.abs(N_(out) - ref_out)[..., :-1].max())
    assert fwd_err <= 1e-4
    errs = {k: _rel_err(N_(film_t[k].grad), film_ref[k]) for k in film}
    named = dict(mod.named_parameters())
    errs.update({k: _rel_err(N_(named[k].grad), v) for k, v in sd_ref.items()})
    worst = max(errs, key=errs.get)
    print(f"[parity] SIREN backward at scale [{('amp16 (bf16 operands + 16-bit tape)' if amp16 else 'amp (bf16 weight-gradient operands)') if amp else ('tape16 (16-bit tape)' if t16 else precision)}] H={H} B={B} P={P} ({nchunks} "
          f"backward launch(es)): worst relative error over {len(errs)} gradient tensors {errs[worst]:.2e} ({worst}); forward max|err| {fwd_err:.1e}")
    # measured: f32 1.6e-5 .. 2.2e-5; f16x3 3.3e-5 .. 4.0e-5
    if t16:
        # the 16-bit tape: +-2^-17 rev = +-4.8e-5 rad on every recomputed activation and cosine, independent over points and layers --
        # 1.1e-4 in this metric by simulation (fp64 backward with quantised phases), on top of the 3.5e-5 of the bf16x3 products.  A tier
        # of its own: above the fp32 class asserted for the default (6e-5), 20 x below the AMP class.
        # measured (round 5): 1.68e-4 (H = 32, 40,000 points), 1.07e-4 (H = 64), 1.23e-4 / 1.18e-4 (H = 256, 65,536 / 393,216 points)
        assert errs[worst] <= 2.5e-4, (worst, errs[worst])
    elif not amp:
        assert errs[worst] <= (4e-5 if precision == "f32" else 6e-5), (worst, errs[worst])
    else:
        # AMP class, opt-in: the upstream gradient here is point-wise random, so every weight gradient is a pure noise sum and the
        # unbiased bf16 roundings (2^-9) show at full size whatever P is (measured 2.4e-3 .. 2.7e-3 at 65,536 and at 393,216 points);
        # what does not pass through the bf16 dump keeps the fp32 class
        through_dump = [k for k in errs if k.endswith("layer.weight") or (amp16 and k.startswith("freq_"))]
        rest = max(errs[k] for k in errs if k not in through_dump)
        print(f"[parity]   ... of which through the bf16 dump {max(errs[k] for k in through_dump):.2e}, everything else {rest:.2e}")
        assert max(errs[k] for k in through_dump) <= 6e-3 and rest <= (2.5e-4 if amp16 else 6e-5)


def test_grid_gradient_values_at_full_size_96cubed_grid():
    """The 96^3 grid-gradient scatter (113 MB of float atomics fused into the chain kernel, siren.py:314-330's grid_sample backward) VALUE-
    checked at the generator step's own size: bench model (H = 256 + 32 x 96^3 grid), the two passes of a 128 x 128 x 24 image = 786,432
    points.  The upstream gradient is non-zero only on a 2,048-ray slab (98,304 points over both passes), so the full-size backward must
    reproduce, voxel for voxel, the fp64 autograd gradient of that slab alone -- and leave every voxel the slab does not touch at exactly
    zero.  Run three times: as ONE chunk (786,432 points in one chain launch and one set of weight-gradient launches: 8.9 GB of
    d(theta), offsets beyond 2^32 bytes), as four serial 196,608-point chunks whose first boundary the slab straddles, and as those
    four chunks on the two-stream schedule (OVERLAP_WGRAD).  All other gradient tensors ride along."""
    from oracle import fenerf_oracle_grad as OG
    from fenerf_amd.siren import autograd as SA
    spec, sd = _full_weights()
    mod = S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=256, z_geo_dim=8, z_app_dim=8, output_dim=22)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
    mod.load_state_dict(tsd, strict=False)
    mod.precision = "f16x3"
    mod = mod.to(DEV)
    S_, N = 128, 24
    R = S_ * S_
    torch.manual_seed(0)
    o, d, z, _, _ = VR.sample_rays(1, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    zf = torch.sort(0.88 + 0.24 * torch.rand((1, R, N), device=DEV), -1)[0]             # stand-in fine depths (any depths do)
    pts = torch.cat([(o[:, :, None, :] + d[:, :, None, :] * zz[..., None]).reshape(1, R * N, 3) for zz in (z, zf)], 0)   # [2 passes, R N, 3]
    dirs = d[:, :, None, :].expand(1, R, N, 3).reshape(1, R * N, 3).expand(2, -1, -1).contiguous()
    film = proc.film_params(spec, 1, seed=0)
    film2 = {k: np.repeat(v, 2, 0) for k, v in film.items()}                          # both passes of one image share its FiLM block
    r0, r1 = 7168, 9216                                                                # rays of the slab: points [172,032, 221,184) of each pass
    assert r0 * N < 196608 < r1 * N
    rng = np.random.default_rng(5)
    g_slab = rng.normal(size=(2, (r1 - r0) * N, 22)).astype(np.float32)
    g_slab[..., -1] *= 1e-3                                                            # sigma is ~2000 x the other outputs in this model
    g_out = torch.zeros((2, R * N, 22), device=DEV)
    g_out[:, r0 * N:r1 * N] = T(g_slab)
    # fp64 autograd on the slab alone, in pieces of 8,192 points (gradients are sums over points)
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    sd64 = {k: t64(v).requires_grad_(True) for k, v in sd.items()}
    film64 = {k: t64(v).requires_grad_(True) for k, v in film2.items()}
    p_slab, d_slab = N_(pts[:, r0 * N:r1 * N]), N_(dirs[:, r0 * N:r1 * N])
    for s0 in range(0, p_slab.shape[1], 8192):
        sl = slice(s0, s0 + 8192)
        ref = OG.siren_forward(sd64, spec, t64(p_slab[:, sl]), t64(d_slab[:, sl]), film64["freq_geo"], film64["phase_geo"], film64["freq_app"], film64["phase_app"])
        (ref * t64(g_slab[:, sl])).sum().backward()
    g_ref = sd64["spatial_embeddings"].grad.numpy()
    touched_ref = np.abs(g_ref).max(1) > 0
    for chunk, overlap in ((2 * R * N, False), (196608, False), (196608, True)):
        old, SA.BACKWARD_CHUNK_POINTS = SA.BACKWARD_CHUNK_POINTS, chunk
        ov_old, SA.OVERLAP_WGRAD = SA.OVERLAP_WGRAD, overlap
        try:
            for p_ in mod.parameters():
                p_.grad = None
            film_t = {k: T(v).requires_grad_(True) for k, v in film2.items()}
            out = mod.forward_with_frequencies_phase_shifts(pts, film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], dirs)
            (out * g_out).sum().backward()
        finally:
            SA.BACKWARD_CHUNK_POINTS, SA.OVERLAP_WGRAD = old, ov_old
        g_nat = N_(mod.spatial_embeddings.grad)
        touched_nat = np.abs(g_nat).max(1) > 0
        e_grid = _rel_err(g_nat, g_ref)
        named = dict(mod.named_parameters())
        errs = {k: _rel_err(N_(film_t[k].grad), film64[k].grad.numpy()) for k in film2}
        errs.update({k: _rel_err(N_(named[k].grad), v.grad.numpy()) for k, v in sd64.items()})
        worst = max(errs, key=errs.get)
        print(f"[parity] 96^3 grid gradient at full size (786,432 points in backward chunks of {chunk}{', two-stream schedule' if overlap else ''}, upstream gradient on a 2,048-ray slab): "
              f"{int(touched_nat.sum())} voxels touched (fp64 autograd of the slab alone: {int(touched_ref.sum())}), relative error (max-norm) "
              f"{e_grid:.2e}; stray non-zero voxels {int((touched_nat & ~touched_ref).sum())}; worst of all {len(errs)} gradient tensors {errs[worst]:.2e} ({worst})")
        assert not (touched_nat & ~touched_ref).any(), "a voxel the slab does not touch received gradient"
        assert e_grid <= 6e-5 and errs[worst] <= 6e-5


@pytest.mark.parametrize("kind,H,grid,B,P", [("texture", 32, 5, 2, 96), ("baseline", 64, 0, 1, 64), ("texture", 128, 4, 3, 160),
                                             ("texture", 256, 6, 2, 224), ("texture", 256, 6, 1, 4224), ("texture", 96, 5, 2, 224), ("texture", 192, 4, 2, 224)])
def test_bf16_dump_layout_at_small_point_counts(kind, H, grid, B, P):
    """The opt-in AMP-class weight gradients (module.grad_precision = "amp": in backward chunks of >= AMP_MIN_POINTS points the chain
    kernel writes d theta and x = sin(2 pi theta) as bf16 and the square weight-gradient job multiplies them with one MFMA per product,
    fenerf_layout.h "bf16 dump").  AMP_MIN_POINTS = 1 forces that path here so that every hidden size, both FiLM-sum units and ragged
    tile counts walk its layout: the weight gradients agree with the default fp32-class path to a few 2^-9 -- a layout error would be
    O(1) -- while everything that does not pass through the dump (FiLM frequencies / phases, FiLM-layer biases, the grid, heads) is
    unchanged."""
    mod, spec, sd = _siren_module(kind, H, grid, precision="f16x3")
    rng = np.random.default_rng(23)
    pts = T(rng.uniform(-0.12, 0.12, (B, P, 3)).astype(np.float32))
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs = T(dirs / np.linalg.norm(dirs, axis=-1, keepdims=True))
    film = proc.film_params(spec, B, seed=4)
    g_out = rng.normal(size=(B, P, spec["output_dim"])).astype(np.float32)
    g_out[..., -1] *= 0.02
    res = {}
    mod.AMP_MIN_POINTS = 1
    mod.FREQ_FROM_WGRAD = False       # like with like: "amp" keeps the chain kernel's own frequency sums (FENERF_TAPE_F32), so the fp32-class leg
                                      # takes them there too (the default's route through the weight-gradient sums differs by 4e-6 .. 1e-5 in d_freq)
    for mode in ("fp32", "bf16"):
        mod.grad_precision = "amp" if mode == "bf16" else "f32"
        try:
            for p_ in mod.parameters():
                p_.grad = None
            film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
            out = mod.forward_with_frequencies_phase_shifts(pts, film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], dirs)
            (out * T(g_out)).sum().backward()
            torch.cuda.synchronize()
            res[mode] = {**{k: N_(v.grad) for k, v in film_t.items()}, **{k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None}}
        finally:
            mod.grad_precision = "f32"
    through_dump = [k for k in res["fp32"] if k.endswith("layer.weight")]          # square jobs + the two thin jobs that read d theta
    others = [k for k in res["fp32"] if k not in through_dump]
    e_dump = max(_rel_err(res["bf16"][k], res["fp32"][k]) for k in through_dump)
    e_rest = max(_rel_err(res["bf16"][k], res["fp32"][k]) for k in others)
    print(f"[parity] bf16 dump forced at {kind} H={H} B={B} P={P}: weight gradients vs the fp32 dump {e_dump:.2e} (2^-9 = 2.0e-3, no "
          f"averaging at this size), everything else {e_rest:.2e}")
    assert e_dump <= 8e-3 and e_rest <= 5e-6


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("H,B,P", [(32, 2, 96), (256, 1, 4224)])
def test_backward_with_fused_grid_scatter_equals_backward_then_scatter(H, B, P, precision):
    """fenerf_siren_backward_grid (f16x3: the chain kernel scatters d(grid features) into the channels-last gradient grid itself;
    f32: chain + scatter kernel over a scratch) == fenerf_siren_backward + fenerf_grid_backward, and d_t is the same dump."""
    spec = proc.model_spec("texture", hidden_dim=H, grid_size=7, z_dim=8)
    sd = proc.make_state_dict(spec, seed=11, sigma_gain=30.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, precision, differentiable=True)
    assert bool(_lib.lib().fenerf_siren_backward_fuses_grid(nat._h)) == (precision == "f16x3")
    rng = np.random.default_rng(3)
    pts = T(rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32))          # some points leave the box: zero padding
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs = T(dirs / np.linalg.norm(dirs, axis=-1, keepdims=True))
    film = {k: T(v) for k, v in proc.film_params(spec, B, seed=4).items()}
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
    g_out = T(rng.normal(size=(B, P, spec["output_dim"])).astype(np.float32))
    d_t0, d_e = nat.siren_backward(B, P, *args, out, g_out, tape)
    ref = torch.zeros((7, 7, 7, 32), device=DEV)
    with torch.cuda.device(nat.device):
        _lib.check(_lib.lib().fenerf_grid_backward(nat._h, B * P, native._ptr(pts.reshape(-1, 3).contiguous()), native._ptr(d_e), native._ptr(ref), None))
    got = torch.zeros((7, 7, 7, 32), device=DEV)
    d_t1 = nat.siren_backward_grid(B, P, *args, out, g_out, tape, pts, got)
    n = (spec["n_geo"] + spec["n_color"]) * H * B * P
    assert torch.equal(d_t0[:n], d_t1[:n])
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    print(f"[parity] fused grid scatter H={H} B={B} P={P} [{precision}]: relative difference {err:.2e} (float atomics in a different order)")
    assert err <= 5e-6 and ref.abs().max().item() > 0          # fp32 sums of up to a few hundred terms in arbitrary order
    # accumulates: a second call doubles the grid
    nat.siren_backward_grid(B, P, *args, out, g_out, tape, pts, got)
    assert (got - 2 * ref).abs().max().item() / ref.abs().max().item() <= 1e-5


def test_forward_save_and_chain_kernels_are_run_to_run_deterministic():
    """The stream loops of the 16-point kernels synchronise with counted vmcnt waits, workgroup barriers and LDS rings and have no
    data-dependent path: a race would show up as run-to-run differences.  40 repetitions of forward-save + chain (+ the FiLM
    gradients, which consume the workgroup-combined sums) on 33,024 points (several octs per workgroup, ragged last one), bit for bit."""
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_chain.py")
    r = subprocess.run([sys.executable, tool, "--determinism", "40", "--H", "64", "--points", "33024", "--iters", "2"], capture_output=True, text=True,
                       timeout=600)
    print("[parity] determinism:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]


@pytest.mark.parametrize("H,B,P", [(256, 1, 4224), (256, 2, 2080), (64, 3, 1024)])
def test_inversion_chain_without_the_dump_equals_the_full_backward(H, B, P):
    """fenerf_siren_backward_film + fenerf_siren_film_grads (inversion: no d(theta) dump, no d(grid features)) against
    fenerf_siren_backward + fenerf_siren_param_grads with NULL weight pointers: the FiLM sums are produced by the same instructions and
    gathered by the same kernels, so the four FiLM gradients must be BIT-identical -- with workgroup-combined sums (one image, or points per
    image a multiple of 128) and with per-wave sums (2,080 points per image).  An exact-fp32 model refuses (its FiLM sums come from the dump)."""
    spec = proc.model_spec("texture", hidden_dim=H, grid_size=8, z_dim=8)
    sd = proc.make_state_dict(spec, seed=6, sigma_gain=60.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, "f16x3", differentiable=True)
    g = torch.Generator(device=DEV).manual_seed(11)
    pts = (torch.rand((B, P, 3), device=DEV, generator=g) - 0.5) * 0.24
    dirs = torch.randn((B, P, 3), device=DEV, generator=g)
    film = {k: torch.tensor(v, device=DEV) for k, v in proc.film_params(spec, B, seed=2).items()}
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
    d_out = torch.randn(out.shape, device=DEV, generator=g)
    d_t, _ = nat.siren_backward(B, P, *args, out, d_out, tape)
    ref = nat.siren_param_grads(pts, dirs, *args, out, d_out, tape, tape_e, d_t, film_only=True)
    assert nat.film_only_native()
    sums = nat.siren_backward_film(B, P, *args, out, d_out, tape)
    got = nat.siren_film_grads(B, P, *args, sums)
    for k in ("d_freq_geo", "d_phase_geo", "d_freq_app", "d_phase_app"):
        assert torch.isfinite(got[k]).all() and float(ref[k].abs().max()) > 0
        assert torch.equal(got[k], ref[k]), k
    print(f"[parity] inversion chain without the dump H={H} B={B} P={P}: FiLM gradients bit-identical; FiLM sums {sums.numel() * 4 / 1e6:.2f} MB "
          f"instead of a {d_t.numel() * 4 / 1e6:.1f} MB dump")
    exact = native.NativeModel(sd, spec, DEV, "f32", differentiable=True)
    assert not exact.film_only_native()
    with pytest.raises(RuntimeError, match="F16X3"):
        exact.siren_backward_film(B, P, *args, out, d_out, tape)


def test_weight_gradient_kernels_are_run_to_run_deterministic():
    """fenerf_siren_param_grads sums in a fixed order (per-chunk partials, then one reduction: no atomics).  With the eight-wave square
    kernel (two waves per SIMD, double-buffered LDS image, one barrier per tile) and the thin jobs' VALU rows a missing barrier or a
    buffer reused too early would show as run-to-run differences: 12 repetitions at H = 256, two images of 2,112 points (66 tiles each,
    ragged chunks), every gradient tensor bit for bit -- default operands and the bf16 dump."""
    spec = proc.model_spec("texture", hidden_dim=256, grid_size=8, z_dim=8)
    sd = proc.make_state_dict(spec, seed=3, sigma_gain=40.0, with_mapping=False)
    B, P = 2, 2112
    g = torch.Generator(device=DEV).manual_seed(7)
    pts = (torch.rand((B, P, 3), device=DEV, generator=g) - 0.5) * 0.24
    dirs = torch.randn((B, P, 3), device=DEV, generator=g)
    film = {k: torch.tensor(v, device=DEV) for k, v in proc.film_params(spec, B, seed=4).items()}
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    for min_pts in (0, 1):
        nat = native.NativeModel(sd, spec, DEV, "f16x3", differentiable=True, wgrad_bf16_min_points=min_pts)
        out, tape, tape_e = nat.siren_forward_save(pts, dirs, *args)
        d_out = torch.randn(out.shape, device=DEV, generator=g)
        d_t, _ = nat.siren_backward(B, P, *args, out, d_out, tape)
        flat = lambda r: [t for k in sorted(r) for t in (r[k] if isinstance(r[k], list) else [r[k]])]
        ref = [t.clone() for t in flat(nat.siren_param_grads(pts, dirs, *args, out, d_out, tape, tape_e, d_t))]
        assert all(torch.isfinite(t).all() for t in ref) and sum(float(t.abs().sum()) for t in ref) > 0
        bad = 0
        for _ in range(12):
            got = flat(nat.siren_param_grads(pts, dirs, *args, out, d_out, tape, tape_e, d_t))
            bad += 0 if all(torch.equal(a, b) for a, b in zip(got, ref)) else 1
        print(f"[parity] weight-gradient determinism ({'bf16 dump' if min_pts else 'fp32-class operands'}): {bad} of 12 repetitions differ over {len(ref)} tensors")
        assert bad == 0
        nat.close()


def test_backward_api_rejects_bad_arguments():
    """Error behaviour of the differentiable entry points: a model created without the backward stream, point counts that
    are not whole tiles, a half-filled gradient struct -- negative status + message, never a launch."""
    import ctypes as C
    spec = proc.model_spec("texture", hidden_dim=32, grid_size=4, z_dim=8)
    sd = proc.make_state_dict(spec, seed=2, sigma_gain=10.0, with_mapping=False)
    plain = native.NativeModel(sd, spec, DEV, "f16x3")
    diff = native.NativeModel(sd, spec, DEV, "f16x3", differentiable=True)
    B, P = 1, 64
    pts = torch.zeros((B, P, 3), device=DEV)
    film = {k: T(v) for k, v in proc.film_params(spec, B, seed=2).items()}
    args = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    with pytest.raises(_lib.FenerfError, match="differentiable"):
        plain.siren_forward_save(pts, None, *args)
    with pytest.raises(_lib.FenerfError, match="multiple of 32"):
        diff.siren_forward_save(pts[:, :50].contiguous(), None, *args)
    out, tape, tape_e = diff.siren_forward_save(pts, None, *args)
    d_out = torch.ones_like(out)
    with pytest.raises(_lib.FenerfError, match="differentiable"):
        plain.siren_backward(B, P, *args, out, d_out, tape)
    d_t, d_e = diff.siren_backward(B, P, *args, out, d_out, tape)
    l = _lib.lib()
    g = _lib.FenerfSirenGrads()
    scratch = torch.zeros(1 << 16, device=DEV)
    for k in ("d_freq_geo", "d_phase_geo", "d_freq_app", "d_phase_app"):
        setattr(g, k, scratch.data_ptr())
    g.geo_w[0] = scratch.data_ptr()                     # some, but not all, weight buffers
    ws = torch.empty(l.fenerf_siren_grad_workspace_bytes(diff._h, B, P), dtype=torch.uint8, device=DEV)
    fws = torch.empty(l.fenerf_film_workspace_bytes(diff._h, B), dtype=torch.uint8, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = l.fenerf_siren_param_grads(diff._h, B, P, p(pts), None, p(args[0]), p(args[1]), p(args[2]), p(args[3]), p(out), p(d_out), p(tape),
                                    p(tape_e), p(d_t), C.byref(g), p(ws), p(fws), None)
    assert rc == _lib.E_INVALID and b"every weight / bias buffer or none" in l.fenerf_last_error()
    rc = l.fenerf_grid_backward(plain._h, -1, p(pts), p(d_e), p(scratch), None)
    assert rc == _lib.E_INVALID
    # device re-pack: maps that do not add up to the model's resident buffers are refused before any launch
    r = diff._repack_maps()
    clone = lambda x: type(x).from_buffer_copy(x)
    flat = torch.zeros(1 << 16, device=DEV)
    bad = clone(r)
    bad.n_stream_h16 = r.n_stream_h16 - 2
    assert l.fenerf_model_repack(diff._h, p(flat), flat.numel(), C.byref(bad), None, None) == _lib.E_INVALID
    assert b"forward stream" in l.fenerf_last_error()
    bad = clone(r)
    bad.bwd_b16 = None
    assert l.fenerf_model_repack(diff._h, p(flat), flat.numel(), C.byref(bad), None, None) == _lib.E_INVALID
    assert b"backward stream" in l.fenerf_last_error()
    assert l.fenerf_model_repack(diff._h, None, 0, C.byref(r), None, None) == _lib.E_INVALID
    rc = l.fenerf_model_export_packed(plain._h, None, 0, None, 0, p(scratch), 5, None)       # no backward stream to export
    assert rc == _lib.E_INVALID
    # AMP-class model: the d(theta) dump's format (fp32 | bf16) follows from the chunk's point count; a weight-gradient call that
    # describes the same buffer as a chunk of the other class is refused (round-3 advisory), the matching one runs
    amp = native.NativeModel(sd, spec, DEV, "f16x3", differentiable=True, wgrad_bf16_min_points=64)
    B2, P2 = 1, 96
    pts2 = torch.zeros((B2, P2, 3), device=DEV)
    out2, tape2, tape_e2 = amp.siren_forward_save(pts2, None, *args)
    d_out2 = torch.ones_like(out2)
    d_t2, _ = amp.siren_backward(B2, P2, *args, out2, d_out2, tape2)                           # 96 points >= 64: bf16 dump
    with pytest.raises(_lib.FenerfError, match="other format"):
        amp.siren_param_grads(pts2[:, :32].contiguous(), None, *args, out2[:, :32].contiguous(), d_out2[:, :32].contiguous(), tape2, tape_e2, d_t2)
    amp.siren_param_grads(pts2, None, *args, out2, d_out2, tape2, tape_e2, d_t2)
    torch.cuda.synchronize()


def test_generator_step_under_autocast_and_gradscaler():
    """The reference wraps the generator step in torch.cuda.amp.autocast and scales the loss (train_double_latent_semantic.py:
    279, :408-420): the mapping networks then emit fp16 FiLM parameters.  The native nodes cast to fp32 on entry
    (custom_fwd), so the step runs, gradients are finite fp32 and close to the non-autocast ones."""
    torch.manual_seed(5)
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    kw = dict(img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.1)
    z = torch.randn(2, 8, device=DEV)
    w = torch.randn((2, 21, 8, 8), device=DEV)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** -6)   # sigma_gain = 150 makes these gradients large: keep the fp16
                                                                  # FiLM gradients the mapping networks receive below 65504

    def step(amp):
        for p in gen.parameters():
            p.grad = None
        torch.manual_seed(11)
        with torch.autocast("cuda", enabled=amp):
            px, _ = gen(z, z, **kw)
            loss = (px.float() * w).sum()
        (scaler.scale(loss) if amp else loss).backward()
        inv = 1.0 / scaler.get_scale() if amp else 1.0
        return {n: N_(p.grad) * inv for n, p in gen.named_parameters() if p.grad is not None}

    g32, g16 = step(False), step(True)
    assert set(g32) == set(g16)
    for k in g32:
        assert np.isfinite(g16[k]).all() and g16[k].dtype == np.float32, k
    # Under autocast the mapping networks round the FiLM parameters to fp16; a SIREN with frequencies ~30 amplifies that
    # through 11 layers, so the two steps render visibly different samples of the same scene: the gradients must point the same
    # way, not agree digit by digit (the reference's own AMP step evaluates the whole SIREN in fp16).
    cos = {k: float((g16[k] * g32[k]).sum() / (np.linalg.norm(g16[k]) * np.linalg.norm(g32[k]) + 1e-30))
           for k in g32 if "mapping_network" not in k and g32[k].size >= 1024}
    print(f"[parity] generator step under autocast + GradScaler: cosine(grad_amp, grad_fp32) over render weights min {min(cos.values()):.3f}")
    assert min(cos.values()) >= 0.7


def test_generator_step_through_ddp(tmp_path):
    """The reference trains DistributedDataParallel(generator, find_unused_parameters=True) (train_double_latent_semantic.py:
    ~170): the native autograd nodes must feed DDP's gradient hooks like ordinary ops.  One process, world_size 1."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    torch.manual_seed(5)
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=50.0)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    kw = dict(img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.0)
    z = torch.randn(2, 8, device=DEV)
    w = torch.randn((2, 21, 8, 8), device=DEV)

    def grads(model):
        for p in gen.parameters():
            p.grad = None
        torch.manual_seed(11)
        px, _ = model(z, z, **kw)
        (px * w).sum().backward()
        return {n: N_(p.grad) for n, p in gen.named_parameters() if p.grad is not None}

    plain = grads(gen)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        assert dist.get_backend() == "nccl"
        probe = torch.full((4,), 3.0, device=DEV)
        dist.all_reduce(probe)                             # RCCL all-reduce on the GPU (world 1: identity)
        assert probe.tolist() == [3.0] * 4
        ddp = DDP(gen, device_ids=[0], find_unused_parameters=True)
        wrapped = grads(ddp)
        opt = torch.optim.Adam(ddp.parameters(), lr=1e-4)
        opt.step()                                         # parameters change in place -> the next render re-packs on the device
        again = grads(ddp)
        del ddp
        # the wrapper this package recommends (fenerf_amd.dist.prepare_for_ddp + RECOMMENDED_DDP_KWARGS): two-node backward, the grid
        # gradient reaches DDP before the weight-gradient kernels run; same gradients as the bare module at the same (stepped) weights
        from fenerf_amd import dist as fdist
        plain2 = grads(gen)
        kw_ddp = fdist.prepare_for_ddp(gen)
        assert gen.siren.split_backward and kw_ddp["find_unused_parameters"] is False
        ddp2 = DDP(gen, device_ids=[0], **kw_ddp)
        grads(ddp2)                                        # first step: DDP rebuilds its buckets in arrival order afterwards
        tuned = grads(ddp2)
        del ddp2
        # round 5: fenerf_amd.dist.GeneratorDataParallel -- the same gradients, the 8^3 grid reduced in place from its hook (async_numel lowered
        # so that this small model takes that path too), everything else as one flat buffer; with and without the two-node backward
        gdp = fdist.GeneratorDataParallel(gen, async_numel=32 * 8 ** 3)
        flat_split = grads(gdp)
        stats = dict(gdp.last_sync)
        fdist.prepare_for_ddp(gen, False)
        flat_single = grads(gdp)
        n_grads = len(flat_single)
        n_own = sum(1 for p in gen.parameters() if p.grad is not None and p.numel() >= gdp.async_numel)      # the grid and the mapping networks' 256 x 256 layers
        views = len({p.grad.untyped_storage().data_ptr() for p in gen.parameters() if p.grad is not None})
        gdp.detach_hooks()
    finally:
        if created:
            dist.destroy_process_group()
    assert set(plain) == set(wrapped)
    assert max(_rel_err(wrapped[k], plain[k]) for k in plain) <= 1e-5
    assert any(np.abs(again[k] - wrapped[k]).max() > 0 for k in plain)
    assert set(plain) == set(tuned) and max(_rel_err(tuned[k], plain2[k]) for k in plain2) <= 1e-5
    for got in (flat_split, flat_single):
        assert set(got) == set(plain2) and max(_rel_err(got[k], plain2[k]) for k in plain2) <= 1e-5
    assert stats["collectives"] == n_own + 1 and stats["flat_tensors"] == n_grads - n_own and stats["bytes"] == 4 * sum(v.size for v in plain2.values())
    assert n_own >= 1 and views == n_own + 1, "after the step every small gradient is a view of the one flat buffer; the large ones are their own"
    print("[parity] generator step through DistributedDataParallel over RCCL (backend nccl, world 1): gradients identical to the bare "
          f"module; optimizer step picked up; fenerf_amd.dist.GeneratorDataParallel: the same gradients from {stats['collectives']} collectives "
          f"({stats['flat_tensors']} tensors in the flat one)")


def test_bench_under_torch_distributed_run_initialises_rccl():
    """The driver's N > 1 command line at N = 1: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...` with
    FENERF_BENCH_FORCE_DIST=1, which makes bench.py take its N > 1 branch (init_process_group("nccl", device_id=...), barrier,
    max-over-ranks all-reduce, all-gathers) with a single rank -- so the first RCCL initialisation of this code base does not happen
    inside the driver's scaling run (reference: train_double_latent_semantic.py:58-63,148-150,584)."""
    import subprocess
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_command(1, ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-f32", "--no-gstep",
                                        "--no-gstep-b6", "--no-sweep64"], script=os.path.join(ROOT, "bench.py"))
    env = dict(os.environ, FENERF_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = r.stdout.rstrip("\n").splitlines()[-1]                 # the LAST stdout line is the compact contract line (round 6) ...
    assert len(line.encode()) < bench.COMPACT_LINE_LIMIT
    c = json.loads(line)
    assert c["n_gpus"] == 1 and c["n_ranks_seen"] == 1 and c["value"] > 1e6 and c["gstep_ddp"]["dist_backend"] == "nccl" and c["detail"] == bench.DETAIL_FILE
    d = json.load(open(os.path.join(ROOT, bench.DETAIL_FILE)))    # ... and every leg in full is in the detail file next to the script
    assert d["value"] == pytest.approx(c["value"], rel=1e-5)
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["dist_backend"] == "nccl"
    assert d["launcher"].startswith("torch.distributed.run") and "forced" in d["launcher"]
    assert d["value"] > 1e6 and len(d["rays_per_s_per_rank"]) == 1 and len(d["roofline"]["frac_per_rank"]) == 1
    # the generator step through DistributedDataParallel over RCCL (the one collective north_star names) is part of every line
    leg = d["gstep_ddp"]
    assert tuple(leg) == bench.GSTEP_DDP_KEYS, leg
    assert leg["dist_backend"] == "nccl" and leg["n_ranks"] == leg["n_ranks_seen"] == 1 and leg["batch_per_rank"] == 1
    assert leg["allreduce_bytes_largest_tensor"] == 32 * 96 ** 3 * 4 and 120e6 < leg["allreduce_bytes"] < 128e6      # 113 MB grid + MLP + mapping nets
    assert 0 < leg["ms_no_ddp"] < 40 and 0 < leg["ms"] < 40 and leg["ms_with_optimizer"] > 0
    print(f"[dist] bench.py under torch.distributed.run, RCCL process group at world 1: {d['value']:.3e} rays/s, n_ranks_seen 1; generator step "
          f"through DDP {leg['ms']:.2f} ms (bare module {leg['ms_no_ddp']:.2f} ms, + Adam {leg['ms_with_optimizer']:.2f} ms), "
          f"{leg['allreduce_bytes'] / 1e6:.1f} MB of gradients per all-reduce")


def _torchrun_two_ranks(script_args, timeout=900):
    import subprocess
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(bench._free_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FENERF_BENCH_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_world_2_on_one_gpu_native_two_stage_backward_through_both_wrappers():
    """World 2 with the real kernels (round-5 review, weak #9 / next #4): two processes on cuda:0 over gloo (RCCL refuses two ranks on one
    device; gloo is the reference's own backend, train_double_latent_semantic.py:63), the curriculum generator at 128 x 128 x 24+24 in each
    with its own latents (tools/world2_one_gpu.py).  Every parameter's gradient after a step through GeneratorDataParallel (two-stage native
    backward, grid all-reduce started from the hook; also on the single-node backward), through DistributedDataParallel(**RECOMMENDED_DDP_KWARGS)
    on the two-stage backward, and through the reference's own DDP(find_unused_parameters=True) is the mean of the two ranks' bare-module
    gradients -- one fp32 rounding of a two-term sum, so only the atomically scattered grid gradient (1e-8) may differ at all --, identical
    on both ranks; two micro-batches under no_sync() / micro_batch_sync give the mean of the per-rank sums."""
    from conftest import ROOT
    d = _torchrun_two_ranks([os.path.join(ROOT, "tools", "world2_one_gpu.py")])
    assert d["world"] == 2 and d["backend"] == "gloo" and d["points_per_image"] == 786432
    assert d["own_vs_mean"] > 1e-2, "the two ranks' own gradients differ (own latents): the mean is not either of them"
    legs = ("gdp_split", "gdp_split_2_micro_batches", "gdp_single_node_backward", "ddp_recommended_split", "ddp_recommended_split_2_micro_batches",
            "ddp_reference_wrapper", "gdp_sparse_backward", "gdp_sparse_backward_2_micro_batches", "gdp_split_and_sparse_backward",
            "ddp_reference_wrapper_sparse_backward")
    for k in legs:
        print(f"[dist] world 2 on one GPU (gloo), {k}: worst relative error vs the mean of the bare-module gradients {d[k]['worst_rel_err']:.1e} "
              f"({d[k]['worst']}; {d[k]['tensors']} tensors), identical on both ranks: {d[k]['identical_on_both_ranks']}")
        # measured 1.2e-6 .. 1.5e-6 (the mapping networks' first-layer gradients: the FiLM-parameter gradients feeding them carry the
        # atomically accumulated sums of the render backward -- run-to-run rounding, not the collective); every other tensor <= 1e-7
        assert d[k]["tensors"] == 53 and d[k]["worst_rel_err"] <= 4e-6 and d[k]["identical_on_both_ranks"], (k, d[k])
    assert d["gdp_split"]["collectives"] >= 2
    print(f"[dist] world 2 on one GPU: peak {d['peak_GB']:.1f} GB per rank")


def test_bench_gpus_2_on_one_device_over_gloo_prints_a_parseable_line():
    """`bench.py --gpus 2` as the driver's scaling run launches it (torch.distributed.run, one rank per process), with both ranks pinned to
    cuda:0 over gloo (--one-device --dist-backend gloo): the N > 1 branch end to end -- process group, barriers, max-over-ranks timing, the
    per-rank gathers, the DDP / GeneratorDataParallel generator-step leg with a real two-rank all-reduce of the 124 MB gradient set -- and the
    compact last line."""
    from conftest import ROOT
    import bench
    d = _torchrun_two_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--one-device", "--dist-backend", "gloo",
                             "--no-cpu-baseline", "--no-f32", "--no-gstep", "--no-gstep-b6", "--no-sweep64"])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "weak" and d["value"] > 1e6
    assert len(json.dumps(d)) < bench.COMPACT_LINE_LIMIT and d["roofline"]["frac"] > 0
    leg = d["gstep_ddp"]
    assert leg["n_ranks_seen"] == 2 and leg["dist_backend"] == "gloo" and leg["ms"] > leg["ms_no_ddp"] > 0 and leg["ms_generator_data_parallel"] > 0
    full = json.load(open(os.path.join(ROOT, bench.DETAIL_FILE)))
    assert len(full["rays_per_s_per_rank"]) == 2 and len(full["roofline"]["frac_per_rank"]) == 2 and "ONE device" in full["launcher"]
    assert full["gstep_ddp"]["n_ranks"] == 2 and full["gstep_ddp"]["allreduce_bytes_largest_tensor"] == 32 * 96 ** 3 * 4
    print(f"[dist] bench.py --gpus 2 on one device over gloo: {d['value']:.3e} rays/s over 2 ranks sharing the GPU; generator step through DDP "
          f"{leg['ms']:.1f} ms, GeneratorDataParallel {leg['ms_generator_data_parallel']:.1f} ms, bare {leg['ms_no_ddp']:.1f} ms (gloo stages "
          f"the all-reduce through the host)")


def test_weight_swaps_through_param_data_are_picked_up_at_mode_switch():
    """torch_ema's copy_to / restore write through param.data (no version bump).  The reference always follows them with
    generator.eval() / .train(); a mode switch therefore invalidates the packed weights.  Also: explicit invalidate_native()."""
    mod, spec, sd = _siren_module("texture", 32, 5)
    pts = T(np.random.default_rng(0).uniform(-0.1, 0.1, (1, 64, 3)).astype(np.float32))
    film = {k: T(v) for k, v in proc.film_params(spec, 1, seed=4).items()}
    args = (film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"])
    with torch.no_grad():
        a = N_(mod.forward_with_frequencies_phase_shifts(pts, *args, None))
        for p in mod._render_params():
            p.data.mul_(1.05)                       # invisible to the version counters
        stale = N_(mod.forward_with_frequencies_phase_shifts(pts, *args, None))
        mod.eval()
        b = N_(mod.forward_with_frequencies_phase_shifts(pts, *args, None))
        for p in mod._render_params():
            p.data.div_(1.05)
        mod.invalidate_native()
        c = N_(mod.forward_with_frequencies_phase_shifts(pts, *args, None))
    assert np.array_equal(stale, a)                 # documents the blind spot the mode switch / invalidate_native() closes
    assert np.abs(b - a).max() > 1e-3
    assert np.abs(c - a).max() <= 1e-4 * max(1.0, np.abs(a).max())


# worst relative error over all gradient tensors vs the reference's own autograd, measured on an MI355X x 1.5:
# texture 6.5e-5, trained 5.1e-5, spatial 2.1e-4, h96 1.45e-3, baseline 2.15e-3 (softplus + last_back cancellation), bigfilm 3.6e-3
GENERATOR_GRADIENT_BOUNDS = {"tiny_texture_grad": 1.0e-4, "tiny_texture_grad_trained": 8e-5, "tiny_spatial_grad": 3.2e-4, "h96_texture_grad": 2.2e-3,
                             "tiny_baseline_grad": 3.3e-3, "tiny_texture_grad_bigfilm": 5.5e-3}


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["tiny_texture_grad", "tiny_baseline_grad", "tiny_spatial_grad", "tiny_texture_grad_bigfilm",
                                  "tiny_texture_grad_trained",       # *_trained: at a state the reference's own Adam run produced (round 5)
                                  "h96_texture_grad"])               # hidden width 96 (round 5)
@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
def test_generator_gradients_vs_reference_autograd(name, precision, sparse):
    """tests/golden/tiny_*_grad.npz: gradients from the REFERENCE's own autograd through forward_with_frequencies (texture:
    hierarchical 8+8, noise, white_back; baseline: softplus, noise, last_back; single-latent: locked view direction).  The
    native differentiable path on the recorded draws must reproduce the pixels and every gradient (the reference ran fp32 on
    the CPU: its own rounding is ~1e-4 of the gradient scale)."""
    g = load_golden(name)
    spec = spec_from_golden(g)
    kind = spec["kind"]
    gen = (_make_spatial_generator if kind == "spatial" else _make_generator)(g, dict(spec, z_dim=spec.get("z_dim", 16)), precision)
    gen.train()
    if sparse:                 # round 6: the exact-sparsity backward (generators/autograd.py) against the same reference gradients, same bounds
        if kind == "spatial":
            pytest.skip("the per-point-modulated generator has its own autograd node")
        gen.siren.sparse_backward = True
    film, tf = _film(g, spec)
    if kind == "spatial":      # the golden's film helper draws the colour slice like the generator test above
        film = film_from_golden(g, spec)
        tf = [T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app")]
    tf = [t.clone().requires_grad_(True) for t in tf]           # freq_geo, phase_geo, freq_app, phase_app
    gen.draws = VR.RecordedDraws([g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"], g["rand_u_fine"],
                                  g["rand_noise_fine"]])
    kw = kwargs_from_golden(g)
    common = dict(img_size=int(g["meta_S"]), fov=12, ray_start=0.88, ray_end=1.12, num_steps=int(g["meta_N"]), h_stddev=0.3,
                  v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, hierarchical_sample=True, sample_dist="gaussian", **kw)
    if kind == "spatial":
        px, _ = gen.forward_with_frequencies(torch.cat([tf[0], tf[2]], -1), torch.cat([tf[1], tf[3]], -1), **common)
    else:
        px, _ = gen.forward_with_frequencies(tf[0], tf[2], tf[1], tf[3], **common)
    assert not gen.draws.arrays and px.requires_grad
    # *_bigfilm (round 4): FiLM phase shifts of +-300 revolutions, first-layer frequency x 4 -- the reference's fp32 radians carry an
    # argument rounding of 1.2e-4 .. 2.4e-4 rad there; its gradients sit 8e-4 from the fp64 restatement (tests/test_oracle_golden.py)
    big = name.endswith("_bigfilm")
    assert np.abs(N_(px) - g["pixels"]).max() <= (2e-3 if big else 1e-3)
    (px * T(g["loss_w"])).sum().backward()
    worst = 0.0
    for t, k in zip(tf, ("freq_geo", "phase_geo", "freq_app", "phase_app")):
        worst = max(worst, _rel_err(N_(t.grad), g["gfilm_" + k]))
    named = dict(gen.siren.named_parameters())
    n = 0
    for k in g:
        if k.startswith("gparam_"):
            worst = max(worst, _rel_err(N_(named[k[7:]].grad), g[k]))
            n += 1
    kept = ""
    if sparse:
        from fenerf_amd.generators import autograd as GA
        GA.SparseHierarchicalRenderFunction.verify()
        k_ = GA.SparseHierarchicalRenderFunction.last_kept
        kept = f" (sparse backward: {int(k_[0])} of {k_[1]} samples kept)"
        assert type(px.grad_fn).__name__ != "HierarchicalRenderFunctionBackward"
    print(f"[parity] generator gradients vs the reference's autograd {name}[{precision}]{kept}: worst relative error over {n + 4} tensors {worst:.2e}")
    # bound = the reference's own fp32 rounding: the fp64 restatement differs from these fixtures by 1.5e-4 (texture),
    # 4.8e-3 (baseline: softplus + last_back cancellation in final_layer.weight) and 1.8e-4 (single latent) on the CPU
    # Round 6: one bound per fixture = measured (profiles/r05_gpu_tests_parity_lines.log:317-328, both precisions within 15 % of each other) x 1.5
    # -- a regression of the texture fixtures from 6e-5 to 5e-3 used to stay green under the single 5e-3
    bound = GENERATOR_GRADIENT_BOUNDS[name]
    assert n == {"texture": 33, "baseline": 30, "spatial": 22}[kind] and worst <= bound, (worst, bound)


def test_amp_class_weight_gradients_against_the_references_own_autocast_step():
    """The opt-in AMP-class weight-gradient operands (siren.grad_precision = "amp") are labelled "the class of the
    reference's own training arithmetic".  tests/golden/tiny_texture_grad_autocast16.npz is that arithmetic: the reference's generator
    step under torch.autocast(float16) on the draws of tiny_texture_grad.npz (tools/make_golden.py::run_grad_autocast_case).  Measured
    against the reference's fp32 gradients, its own autocast step is off by 0.23 (median) .. 0.98 (worst) over the FiLM-layer weight
    gradients and by 0.03 in the pixels; this package's AMP mode (forced here at 576 points per pass: AMP_MIN_POINTS = 1) must be at
    least an order of magnitude closer on every one of those tensors, with fp32-class pixels."""
    g, ga = load_golden("tiny_texture_grad"), load_golden("tiny_texture_grad_autocast16")
    spec = spec_from_golden(g)
    gen = _make_generator(g, dict(spec, z_dim=16), "f16x3")
    gen.train()
    gen.siren.grad_precision, gen.siren.AMP_MIN_POINTS = "amp", 1
    film, tf = _film(g, spec)
    tf = [t.clone().requires_grad_(True) for t in tf]
    gen.draws = VR.RecordedDraws([g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"], g["rand_u_fine"],
                                  g["rand_noise_fine"]])
    common = dict(img_size=int(g["meta_S"]), fov=12, ray_start=0.88, ray_end=1.12, num_steps=int(g["meta_N"]), h_stddev=0.3,
                  v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, hierarchical_sample=True, sample_dist="gaussian", **kwargs_from_golden(g))
    px, _ = gen.forward_with_frequencies(tf[0], tf[2], tf[1], tf[3], **common)
    (px * T(g["loss_w"])).sum().backward()
    named = dict(gen.siren.named_parameters())
    keys = [k for k in g if k.startswith("gparam_") and k.endswith("layer.weight")]
    mine = {k: _rel_err(N_(named[k[7:]].grad), g[k]) for k in keys}
    theirs = {k: _rel_err(ga[k], g[k]) for k in keys}
    e_px, e_px_ref = np.abs(N_(px) - g["pixels"]).max(), np.abs(ga["pixels"] - g["pixels"]).max()
    print(f"[parity] AMP-class weight gradients vs the reference's fp32 autograd over {len(keys)} FiLM-layer weights: worst {max(mine.values()):.2e} "
          f"(pixels {e_px:.1e}); the reference's own autocast(float16) step: median {np.median(list(theirs.values())):.2f}, worst "
          f"{max(theirs.values()):.2f} (pixels {e_px_ref:.3f})")
    assert e_px <= 1e-3 and max(mine.values()) <= 1e-2
    assert all(mine[k] <= 0.1 * theirs[k] for k in keys)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_part_forward_vs_reference(precision):
    """tests/golden/tiny_texture_part_forward.npz: the reference's generator.forward(z_geo, z_app, grad_points=11, ...) --
    every draw including the randperm that picks the differentiable rays, the pixels, and the render-weight gradients of a
    fixed loss (only the picked rays carry gradient).  Same draw order, same pixels, same gradients."""
    g = load_golden("tiny_texture_part_forward")
    spec = spec_from_golden(g)
    gen = _make_generator(g, dict(spec, z_dim=16), precision)
    gen.train()
    draws = [g[k] for k in sorted(k for k in g if k.startswith("draw"))]
    gen.draws = VR.RecordedDraws(draws)
    kw = dict(img_size=int(g["meta_S"]), num_steps=int(g["meta_N"]), hierarchical_sample=True, clamp_mode="relu", nerf_noise=0.1,
              grad_points=int(g["meta_G"]), fov=12, ray_start=0.88, ray_end=1.12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi * 0.5,
              v_mean=np.pi * 0.5, sample_dist="gaussian")
    px, poses = gen(T(g["z_geo"]), T(g["z_app"]), **kw)
    assert not gen.draws.arrays, "all recorded draws consumed, in order"
    np.testing.assert_allclose(N_(poses), g["poses"], atol=1e-6)
    assert np.abs(N_(px) - g["pixels"]).max() <= 1e-3
    (px * T(g["loss_w"])).sum().backward()
    named = dict(gen.siren.named_parameters())
    worst, n = 0.0, 0
    for k in g:
        if k.startswith("gparam_"):
            worst = max(worst, _rel_err(N_(named[k[7:]].grad), g[k]))
            n += 1
    print(f"[parity] part_forward vs the reference [{precision}]: pixels max|err| {np.abs(N_(px) - g['pixels']).max():.1e}, "
          f"worst relative gradient error over {n} tensors {worst:.2e}")
    assert n == 33 and worst <= 2e-3


def test_merge_composite_and_render_beyond_128_samples():
    """72+72 and 128+128 samples per ray (e.g. --ray_step_multiplier 3 on a 24-step curriculum): the merge / composite kernels
    keep four samples per lane.  Merge vs the numpy oracle; a full hierarchical render runs end to end."""
    rng = np.random.default_rng(11)
    for N in (72, 128):
        BR = 9
        fine = rng.normal(size=(BR, N, 22)).astype(np.float32); fine[..., -1] *= 20
        coarse = rng.normal(size=(BR, N, 22)).astype(np.float32); coarse[..., -1] *= 20
        zf = np.sort(rng.uniform(0.88, 1.12, (BR, N)).astype(np.float32), -1)
        zc = np.sort(rng.uniform(0.88, 1.12, (BR, N)).astype(np.float32), -1)
        zf[:, 3] = zc[:, 5]                           # ties: stable order (fine first)
        zf.sort(-1)
        opts = _lib.composite_opts("relu", last_back=True)
        rgb, depth, w, ws, zs = native.merge_composite(T(fine), T(coarse), T(zf), T(zc), None, opts)
        mo, mz = O.merge_sorted(fine[None], coarse[None], zf[None, ..., None], zc[None, ..., None])
        r_rgb, r_depth, r_w = O.fancy_integration(mo, mz, clamp_mode="relu", last_back=True)
        np.testing.assert_array_equal(N_(zs), mz[0, ..., 0])
        np.testing.assert_allclose(N_(rgb), r_rgb[0], atol=3e-5)
        np.testing.assert_allclose(N_(w), r_w[0, ..., 0], atol=1e-5)
        np.testing.assert_allclose(N_(depth), r_depth[0, ..., 0], atol=3e-5)
    nat, spec, sd = _native_for("tiny_texture_fwd", "f16x3")
    g = load_golden("tiny_texture_fwd")
    film, tf = _film(g, spec)
    B, S_, N = int(g["meta_B"]), 4, 72
    o, d, z, _, _ = VR.sample_rays(B, N, torch.device(DEV), 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * S_ * S_, N), device=DEV)
    rgb, depth, _, _ = nat.render(o, d, z, u, None, None, *tf, _lib.composite_opts("relu"), hierarchical=True)
    assert rgb.shape == (B, S_ * S_, 21) and torch.isfinite(rgb).all() and torch.isfinite(depth).all()


# ---------------------------------------------------------------------------------------------------
# FiLM arguments beyond the init range (round 4): torch.sin in the reference's FiLMLayer (siren.py:113-123) takes fp32 radians of any
# magnitude; the GCN3 / Vega manuals define v_sin_f32 / v_cos_f32 on +-256 revolutions only (beyond: sin = 0, cos = 1, silently).  gfx950's
# instructions are full-range (fenerf_trig.h: compiler lowering, hardware sweep) and the kernels feed them unreduced revolutions; these
# tests are what pins that per kernel family -- they fail on a device (or a build) where the assumption breaks.  Bar: |native - fp64| <=
# ulp_fp32(largest sine argument) * 2 pi -- the error one fp32 rounding of the argument makes, which the reference's own fp32 radians
# carry too (the fp32 numpy oracle is reported beside).
# ---------------------------------------------------------------------------------------------------
REVS = [45, 120, 250, 257, 400, 1000]


def _big_film(spec, B, rev, seed=1):
    """FiLM parameters whose sine arguments reach ~rev revolutions in EVERY layer (phase shifts) and in layer 0 through its frequency too"""
    return proc.film_params(spec, B, seed=seed, phase_rev=float(rev), freq0_gain=max(1.0, rev / 10.0))


def _arg_ulp_2pi(rev_max):
    return float(np.spacing(np.float32(rev_max))) * 2 * np.pi


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("rev", REVS)
def test_siren_forward_with_sine_arguments_far_beyond_init(rev, precision):
    """f32 forward (siren_kernel) and f16x3 forward (siren16w_kernel), H = 256 + grid, 2,000 points (ragged tiles), vs the fp64 oracle."""
    spec = proc.model_spec("texture", hidden_dim=256, grid_size=16)
    sd = proc.make_state_dict(spec, seed=1, sigma_gain=1.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, precision)
    rng = np.random.default_rng(rev)
    B, P = 2, 1000
    pts = rng.uniform(-0.12, 0.12, (B, P, 3)).astype(np.float32)
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = _big_film(spec, B, rev)
    a = tuple(film[k] for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    out = N_(nat.siren_forward(T(pts), T(dirs), *(T(v) for v in a)))
    tap = []
    o64 = O.siren_forward(sd, spec, pts, dirs, *a, dtype=np.float64, rev_tap=tap)
    o32 = O.siren_forward(sd, spec, pts, dirs, *a)
    bound = _arg_ulp_2pi(max(tap))
    e, e32 = np.abs(out - o64), np.abs(o32 - o64)
    print(f"[parity] sine domain {precision} forward, arguments up to {max(tap):.0f} rev (layer 0) / {sorted(tap)[-2]:.0f} rev (others): "
          f"max|err| vs fp64 rgb {e[..., -4:-1].max():.2e} labels {e[..., :-4].max():.2e} sigma {e[..., -1].max():.2e}; fp32 numpy oracle "
          f"{e32.max():.2e}; bound ulp(arg) 2 pi = {bound:.2e}")
    assert np.isfinite(out).all() and e.max() <= bound


@pytest.mark.parametrize("rev", REVS)
def test_pointwise_and_local_kernels_with_sine_arguments_far_beyond_init(rev):
    """fenerf_siren_forward_pointwise (explicit per-point FiLM) and fenerf_siren_forward_local (per-point mapping network + SIREN in
    one launch): SPATIALSIRENGRID whose mapping network emits phase shifts of up to +-rev revolutions (its output bias), vs fp64."""
    torch.manual_seed(3)
    H = 64
    mod = S.SPATIALSIRENGRID(input_dim=3, z_dim=16, hidden_dim=H, output_dim=4).to(DEV).eval()
    mod.device = torch.device(DEV)
    last = mod.mapping_network.network[-1]
    half = last.bias.numel() // 2
    with torch.no_grad():
        last.bias[half:] += T(proc.uniform("map.bias.rev", (half,), -2 * np.pi * rev, 2 * np.pi * rev, seed=rev))
    B, P = 2, 700
    g_ = torch.Generator(device=DEV).manual_seed(4)
    pts = (torch.rand((B, P, 3), device=DEV, generator=g_) - 0.5) * 0.24
    dirs = torch.nn.functional.normalize(torch.randn((B, P, 3), device=DEV, generator=g_), dim=-1)
    lat = torch.randn((B, 32, 32, 32), device=DEV, generator=g_)
    with torch.no_grad():
        fused = mod.forward_with_latent_grid(pts, lat, dirs)
        sampled = mod.sample_local_latents(lat, mod.gridwarper(pts))
        f, p = mod.mapping_network(sampled)
        local = mod.get_local_coordinates(pts, 32, preserve_y=False)
        explicit = mod.forward_with_frequencies_phase_shifts(local, f, p, dirs)
    # fp64: the mapping network and the SIREN on the same fp32 inputs (sampled latents, local coordinates)
    sd = mod._state_numpy()
    spec = mod._spec()
    msd = {"m.network." + n: N_(q) for n, q in mod.mapping_network.network.named_parameters()}
    f64, p64 = O.mapping_network({k: v.astype(np.float64) for k, v in msd.items()}, "m", N_(sampled).astype(np.float64))
    tap = []
    o_fused = O.siren_forward(sd, spec, N_(local), N_(dirs), f64, p64, dtype=np.float64, rev_tap=tap)
    # the explicit route is given the fp32 FiLM tensors torch computed: its fp64 reference takes exactly those
    o_expl = O.siren_forward(sd, spec, N_(local), N_(dirs), N_(f), N_(p), dtype=np.float64)
    bound = _arg_ulp_2pi(max(tap))
    e_f, e_x = np.abs(N_(fused) - o_fused).max(), np.abs(N_(explicit) - o_expl).max()
    print(f"[parity] sine domain per-point FiLM, arguments up to {max(tap):.0f} rev: pointwise kernel vs fp64 {e_x:.2e}, one-launch local kernel "
          f"(phase shifts computed in-kernel in fp32: one more rounding of a ~{max(tap):.0f}-rev value) {e_f:.2e}; bound ulp(arg) 2 pi = {bound:.2e}")
    assert e_x <= bound and e_f <= 2 * bound


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("rev", REVS)
def test_siren_backward_with_sine_arguments_far_beyond_init(rev, precision):
    """Chain kernels (cos of the recomputed phase) and weight-gradient kernels (sin recomputed from the tape) -- fp32 and bf16x3
    families -- vs fp64 autograd of the restatement, FiLM arguments up to ~rev revolutions.  A gradient carries the argument error
    through cos / sin like the forward does: bound = the suite's fp32-class 2e-4 + 8 ulp(arg) 2 pi, relative (max-norm)."""
    from oracle import fenerf_oracle_grad as OG
    kind, H, grid, B, P = "texture", 64, 5, 2, 300
    mod, spec, sd = _siren_module(kind, H, grid, precision=precision, sigma_gain=1.0)
    rng = np.random.default_rng(rev)
    pts = rng.uniform(-0.12, 0.12, (B, P, 3)).astype(np.float32)
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = _big_film(spec, B, rev, seed=4)
    g_out = rng.normal(size=(B, P, spec["output_dim"])).astype(np.float32)
    film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
    out = mod.forward_with_frequencies_phase_shifts(T(pts), film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], T(dirs))
    (out * T(g_out)).sum().backward()
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    sd64 = {k: t64(v).requires_grad_(True) for k, v in sd.items()}
    film64 = {k: t64(v).requires_grad_(True) for k, v in film.items()}
    ref = OG.siren_forward(sd64, spec, t64(pts), t64(dirs), film64["freq_geo"], film64["phase_geo"], film64["freq_app"], film64["phase_app"])
    (ref * t64(g_out)).sum().backward()
    tap = []
    O.siren_forward(sd, spec, pts, dirs, film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"], dtype=np.float64, rev_tap=tap)
    bound = 2e-4 + 8 * _arg_ulp_2pi(max(tap))       # measured: 2.4e-4 .. 2.7e-4 at 62 rev (bound 3.9e-4), 2.8e-3 .. 3.3e-3 at 1,359 rev (6.3e-3)
    named = dict(mod.named_parameters())
    errs = {k: _rel_err(N_(film_t[k].grad), film64[k].grad.numpy()) for k in film}
    errs.update({k: _rel_err(N_(named[k].grad), v.grad.numpy()) for k, v in sd64.items()})
    worst = max(errs, key=errs.get)
    fe = np.abs(N_(out) - ref.detach().numpy()).max()
    print(f"[parity] sine domain {precision} backward, arguments up to {max(tap):.0f} rev: forward-save vs fp64 {fe:.2e}; worst relative gradient "
          f"error over {len(errs)} tensors {errs[worst]:.2e} ({worst}); bound {bound:.2e}")
    assert fe <= _arg_ulp_2pi(max(tap)) and errs[worst] <= bound


@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
@pytest.mark.parametrize("name", ["tiny_texture_fwd_bigfilm", "h256_texture_8x8_n12_bigfilm"])
def test_reference_fixtures_far_beyond_the_init_range(name, precision):
    """The reference's own outputs with FiLM phase shifts of +-300 revolutions in every layer and the first layer's frequency x 4
    (tools/make_golden.py, round 4; sine arguments 256 .. 320 revolutions: all beyond the hardware sine's documented domain): the
    SIREN kernels on the recorded coarse and fine points (teacher-forced), then generator.forward_with_frequencies on the recorded
    draws.  Tolerances = the fixture's distance from fp64 (tests/test_oracle_golden.py): rgb 5e-5, sigma 1e-4 x sigma_gain; end to
    end the bulk of the pixels."""
    g = load_golden(name)
    nat, spec, sd = _native_for(name, precision)
    film, tf = _film(g, spec)
    B, R, N = g["st_z_coarse"].shape[:3]
    pts = g["st_points"].reshape(B, R * N, 3)
    dirs = np.broadcast_to(g["st_dirs"][:, :, None, :], (B, R, N, 3)).reshape(B, R * N, 3)
    sig_tol = 1e-4 * float(g["meta_sigma_gain"])
    for tag, p_, ref in (("coarse", pts, g["st_siren_coarse"]), ("fine", g["st_fine_points"], g["st_siren_fine"])):
        out = N_(nat.siren_forward(T(p_), T(dirs), *tf))
        _report(f"{name}[{precision}] {tag} vs reference (sine arguments 256 .. 320 rev)", out, ref)
        np.testing.assert_allclose(out[..., -4:-1], ref[..., -4:-1], atol=5e-5)
        np.testing.assert_allclose(out[..., :-4], ref[..., :-4], atol=2e-6, rtol=1e-4)
        np.testing.assert_allclose(out[..., -1], ref[..., -1], atol=sig_tol, rtol=2e-4)
    gen = _make_generator(g, dict(spec, z_dim=16 if spec["hidden_dim"] == 32 else 256), precision)
    gen.draws = VR.RecordedDraws([g["rand_u_jitter"], g["rand_r_theta"], g["rand_r_phi"], g["rand_noise_coarse"], g["rand_u_fine"], g["rand_noise_fine"]])
    with torch.no_grad():
        px, poses = gen.forward_with_frequencies(tf[0], tf[2], tf[1], tf[3], img_size=int(g["meta_S"]), fov=12, ray_start=0.88, ray_end=1.12,
                                                 num_steps=int(g["meta_N"]), h_stddev=0.3, v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5,
                                                 hierarchical_sample=True, sample_dist="gaussian", **kwargs_from_golden(g))
    assert not gen.draws.arrays
    e = np.abs(N_(px) - g["pixels"]).max(axis=1)
    print(f"[parity] {name}[{precision}] forward_with_frequencies vs reference: median|err| {np.median(e):.2e} max {e.max():.2e}, {int((e > 2e-3).sum())} of "
          f"{e.size} pixels beyond 2e-3")
    # measured (round 4): tiny median 6.6e-6 / 6.9e-6, max 2.6e-5 / 2.7e-5; h256 median 6.0e-6 / 6.2e-6, max 5.1e-4 / 3.8e-4 (f32 / f16x3);
    # no pixel beyond 2e-3 in any of them (round 4 allowed 5 % of the pixels there and a 2e-4 median)
    # f16x3c2 (round 6): tiny median 1.46e-5 max 4.18e-5; h256 median 5.8e-6 max 3.82e-4
    b_med, b_max = (2.2e-5, 6.3e-5) if precision == "f16x3c2" else (1.1e-5, 4.5e-5)
    assert np.median(e) <= b_med and e.max() <= (8e-4 if spec["hidden_dim"] == 256 else b_max)


def test_integration_md_binding_renders():
    """INTEGRATION.md B executed as written on the GPU: model_from_siren builds a FenerfModel from a module with the reference's
    attribute names, render_forward drives fenerf_render_forward through the documented ctypes calls -- pixels and depth bit for bit
    those of the package's own binding on the same inputs (tests/test_host_cpu.py checks the struct mirrors on the CPU)."""
    from test_host_cpu import integration_md_binding
    ns = integration_md_binding()
    mod, spec, sd = _siren_module("texture", 64, 6, sigma_gain=300.0)
    h = ns["model_from_siren"](mod, precision=1)
    nat = native.NativeModel(sd, spec, DEV, "f16x3")
    B, S_, N = 2, 12, 12
    film = proc.film_params(spec, B, seed=3)
    tf = tuple(T(film[k]) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    torch.manual_seed(1)
    o, d, z, _, _ = VR.sample_rays(B, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * S_ * S_, N), device=DEV)
    nc, nf = torch.randn((B, S_ * S_, N), device=DEV), torch.randn((B, S_ * S_, 2 * N), device=DEV)
    mine = _lib.composite_opts("relu", 0.3, fill_mode="seg_padding_background", fill_color="white")
    opts = ns["Opts"](clamp_mode=1, noise_std=0.3, last_back=0, white_back=0, black_back=0, fill_mode=2, fill_value=1.0, fill_enabled=1)
    assert bytes(opts) == bytes(mine)
    px, dp = ns["render_forward"](h, o, d, z, u, nc, nf, *tf, opts)
    rgb, depth, _, _ = nat.render(o, d, z, u, nc, nf, *tf, mine, hierarchical=True)
    torch.cuda.synchronize()
    assert px.shape == rgb.shape and torch.equal(px, rgb) and torch.equal(dp, depth) and float(px.std()) > 0.01
    ns["_l"].fenerf_model_destroy.argtypes = [ctypes_void_p()]
    ns["_l"].fenerf_model_destroy(h)
    print("[parity] INTEGRATION.md B binding: model_from_siren + render_forward through the documented ctypes calls == the package's render, bit for bit")


def test_integration_md_binding_generator_step():
    """INTEGRATION.md B, the generator-step half (round 5), executed as written: render_forward_save / render_backward drive
    fenerf_render_forward_save / fenerf_render_backward through the documented ctypes calls on a module with the reference's attribute names --
    pixels and every gradient (FiLM parameters, every render parameter incl. the un-folded label head and the feature grid) against the
    package's own autograd path on the same inputs: the kernels' outputs bit for bit, the label head's un-fold to fp32 rounding."""
    from test_host_cpu import integration_md_binding
    ns = integration_md_binding()
    mod, spec, sd = _siren_module("texture", 64, 6, sigma_gain=300.0)
    mod.FREQ_FROM_WGRAD = False      # the binding passes FENERF_TAPE_F32 (tape format 0): compare with the package's route on the same format
    h = ns["model_from_siren"](mod, precision=1, differentiable=1)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=64), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N = 2, 8, 12
    R = S_ * S_
    film = proc.film_params(spec, B, seed=3)
    tf = [T(film[k]).requires_grad_(True) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app")]
    torch.manual_seed(1)
    o, d, z, _, _ = VR.sample_rays(B, N, DEV, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * R, N), device=DEV)
    nc, nf = torch.randn((B * R, N), device=DEV), torch.randn((B * R, 2 * N), device=DEV)
    g_px = torch.randn((B, R, 21), device=DEV)
    opts = ns["Opts"](clamp_mode=1, noise_std=0.3, last_back=0, white_back=0, black_back=0, fill_mode=0, fill_value=0.0, fill_enabled=1)
    px, dp, save = ns["render_forward_save"](h, o, d, z, u, nc, nf, *[t.detach() for t in tf], opts)
    film_g, param_g = ns["render_backward"](h, mod, save, z, nf, opts, g_px)
    # the package's own route: HierarchicalRenderFunction on the same inputs
    from fenerf_amd.generators.autograd import HierarchicalRenderFunction
    mine = _lib.composite_opts("relu", 0.3)
    assert bytes(opts) == bytes(mine)
    rgb, depth = HierarchicalRenderFunction.apply(mod, mine, mine, False, o, d, z, u, nc, nf, *tf, *mod._render_params())
    (rgb * g_px).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(px, rgb.detach()) and torch.equal(dp, depth)
    for t, gt in zip(tf, film_g):
        assert torch.equal(t.grad, gt)
    named = {k: p_ for k, p_ in mod.named_parameters() if "mapping_network" not in k}
    assert set(named) == set(param_g), sorted(set(named) ^ set(param_g))
    worst = 0.0
    for k, p_ in named.items():
        if k.startswith("label_layer_linear"):
            worst = max(worst, _rel_err(N_(param_g[k]), N_(p_.grad)))
        elif k == "spatial_embeddings":      # float atomics: unordered sum
            assert _rel_err(N_(param_g[k]), N_(p_.grad)) <= 1e-6, k
        else:
            assert torch.equal(param_g[k].reshape(p_.shape), p_.grad), k
    assert worst <= 1e-5, worst
    # the same backward as two calls (fenerf_render_backward_stage 1 / 2, keep_chunks 1 and 2): the grid gradient handed to on_grid_ready
    # is already the final one -- bit for bit what the call returns afterwards --, every other gradient equals the one-call backward's
    for keep in (1, 2):
        at_stage1 = []
        film_g2, param_g2 = ns["render_backward"](h, mod, save, z, nf, opts, g_px, on_grid_ready=lambda t_: at_stage1.append(t_.clone()), keep_chunks=keep)
        assert len(at_stage1) == 1 and torch.equal(at_stage1[0], param_g2["spatial_embeddings"]), "finished at stage 1"
        assert _rel_err(N_(param_g2["spatial_embeddings"]), N_(param_g["spatial_embeddings"])) <= 1e-6       # float atomics
        assert all(torch.equal(a_, b_) for a_, b_ in zip(film_g, film_g2))
        assert all(torch.equal(param_g[k], param_g2[k]) for k in param_g if k != "spatial_embeddings"), keep
    ns["_l"].fenerf_model_destroy.argtypes = [ctypes_void_p()]
    ns["_l"].fenerf_model_destroy(h)
    print(f"[parity] INTEGRATION.md B binding, generator step: render_forward_save + render_backward through the documented ctypes calls == the package's "
          f"autograd path (pixels, 4 FiLM gradients, {len(named) - 6} kernel-written parameter gradients bit for bit; the label head's un-fold {worst:.1e}); "
          f"the two-stage form (fenerf_render_backward_stage) hands over the finished grid gradient at stage 1 and returns the same gradients")


def ctypes_void_p():
    import ctypes
    return ctypes.c_void_p


@pytest.mark.parametrize("native_route", [True, False])
def test_spatial_siren_grid_gradients_vs_reference_autograd(native_route):
    """SURVEY 8 f4, backward (round 4): the reference's SPATIALSIRENGRID is an ordinary differentiable nn.Module (siren.py:413-477).
    tests/golden/tiny_spatial_grid.npz holds ITS OWN autograd gradients of sum(out * w): wrt the SIREN weights, the per-point mapping
    network, z (through the StyleGAN2-style latent-grid generator; that generator's 4.1 M parameter gradients as per-tensor norms) and,
    teacher-forced, the latent grid.  Here: the same module on the GPU under autograd, every one of them -- round 6: with the per-point-
    modulated SIREN on the NATIVE kernels (siren.autograd.PointwiseSirenFunction: forward-save, chain and weight-gradient jobs with one FiLM
    block per point), the per-point mapping network / latent sampling / StyleGenerator2D as PyTorch-ROCm autograd around it; `native=False`
    = the rounds-3-5 route (the SIREN as PyTorch-ROCm ops), which must still pass the same bounds.  The no-grad native launch agrees with
    the autograd route's forward values."""
    import json
    from test_host_cpu import _spatial_grid_module
    g = load_golden("tiny_spatial_grid")
    mod = _spatial_grid_module(g).to(DEV).train()
    mod.device = torch.device(DEV)
    mod.NATIVE_POINTWISE_BACKWARD = native_route
    z = T(g["z"]).requires_grad_(True)
    out = mod(T(g["points"]), z, T(g["dirs"]))
    assert out.requires_grad and out.is_cuda
    assert ("PointwiseSirenFunction" in str(out.grad_fn) or "Slice" in str(out.grad_fn)) == native_route, out.grad_fn
    with torch.no_grad():
        nat_out = mod(T(g["points"]), T(g["z"]), T(g["dirs"]))                  # one native launch (fenerf_siren_forward_local)
    e_fwd = max(np.abs(N_(out) - g["out"])[..., :3].max(), np.abs(N_(out) - N_(nat_out))[..., :3].max())
    (out * T(g["loss_w"])).sum().backward()
    errs = {"z": _rel_err(N_(z.grad), g["g_z"])}
    named = dict(mod.named_parameters())
    for k in g:
        if k.startswith("gw_"):
            assert named[k[3:]].grad is not None, k
            errs[k[3:]] = _rel_err(N_(named[k[3:]].grad), g[k])
    names = json.loads(str(g["g_generator_names"]))
    norms = np.array([float(named["grid_latent_network." + n].grad.double().norm()) for n in names])
    errs["latent-grid generator (per-tensor norms)"] = float(np.abs(norms / g["g_generator_norms"] - 1).max())
    lg = T(g["latent_grid"]).requires_grad_(True)
    out_l = mod.forward_with_latent_grid(T(g["points"]), lg, T(g["dirs"]))
    (out_l * T(g["loss_w"])).sum().backward()
    errs["latent_grid"] = _rel_err(N_(lg.grad), g["g_latent_grid"])
    # explicit per-point FiLM tensors that require grad (the reference's forward_with_frequencies_phase_shifts signature)
    f, p = T(g["freq"]).requires_grad_(True), T(g["phase"]).requires_grad_(True)
    out_x = mod.forward_with_frequencies_phase_shifts(T(g["local_coords"]), f, p, T(g["dirs"]))
    out_x.sum().backward()
    assert np.abs(N_(out_x) - g["out"]).max() <= 1e-4 and f.grad is not None and p.grad is not None and float(f.grad.abs().max()) > 0
    worst = max(errs, key=errs.get)
    print(f"[parity] SPATIALSIRENGRID gradients vs the reference's own autograd [{'native per-point backward' if native_route else 'PyTorch-ROCm SIREN'}]: worst "
          f"relative error over {len(errs)} tensors {errs[worst]:.2e} ({worst}); forward under autograd vs reference / vs the native launch rgb {e_fwd:.2e}")
    # measured: native 2.9e-4, PyTorch-ROCm route 2.0e-4 (both on z, through 4.1 M generator parameters); x 1.5 (rounds 4-5 asserted 2e-3)
    assert e_fwd <= 2e-5 and errs[worst] <= (4.4e-4 if native_route else 3e-4), errs


@pytest.mark.parametrize("H,B,P", [(32, 2, 96), (64, 1, 100), (96, 3, 160), (192, 1, 512), (256, 2, 2048)])
def test_pointwise_siren_backward_native_vs_fp64_autograd(H, B, P):
    """fenerf_siren_forward_save_pointwise / _backward_pointwise / _param_grads_pointwise (round 6): the per-point-modulated
    SIREN (siren.py:464-477 with [B, P, 9H] frequencies / phase shifts) through SPATIALSIRENGRID.forward_with_frequencies_phase_shifts under
    autograd, against fp64 autograd of the same statements -- the gradient of every SIREN weight and bias and of the per-point frequency /
    phase tensors themselves --, and against the PyTorch-ROCm route on the same inputs.  P = 100: padded to whole 32-point tiles."""
    torch.manual_seed(H + P)
    mod = S.SPATIALSIRENGRID(input_dim=3, z_dim=16, hidden_dim=H, output_dim=4).to(DEV).train()
    mod.device = torch.device(DEV)
    with torch.no_grad():
        mod.final_layer.weight.mul_(20.0)
    rng = np.random.default_rng(7)
    pts = T(rng.uniform(-1, 1, (B, P, 3)).astype(np.float32))
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs = T(dirs / np.linalg.norm(dirs, axis=-1, keepdims=True))
    f0 = T(rng.normal(0, 0.4, (B, P, 9 * H)).astype(np.float32))
    p0 = T(rng.normal(0, 0.4, (B, P, 9 * H)).astype(np.float32))
    w = T(rng.normal(size=(B, P, 4)).astype(np.float32))
    w[..., -1] *= 0.05
    res = {}
    for route in ("native", "torch"):
        mod.NATIVE_POINTWISE_BACKWARD = route == "native"
        mod.zero_grad(set_to_none=True)
        f, p = f0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
        out = mod.forward_with_frequencies_phase_shifts(pts, f, p, dirs)
        assert ("PointwiseSirenFunction" in str(out.grad_fn) or "Slice" in str(out.grad_fn)) == (route == "native"), out.grad_fn
        (out * w).sum().backward()
        res[route] = dict(out=N_(out), f=N_(f.grad), p=N_(p.grad), **{n: N_(q.grad) for n, q in mod.named_parameters() if q.grad is not None})
    # fp64 autograd of the reference's statements (siren.py:464-477)
    t64 = lambda t: t.detach().double().cpu()
    prm = {n: t64(q).requires_grad_(True) for n, q in mod.named_parameters() if mod._is_render_param(n)}
    f64, p64 = t64(f0).requires_grad_(True), t64(p0).requires_grad_(True)
    x = t64(pts) * (2 / 0.24)
    fr = f64 * 15 + 30
    for i in range(8):
        x = torch.sin(fr[..., i * H:(i + 1) * H] * torch.nn.functional.linear(x, prm[f"network.{i}.layer.weight"], prm[f"network.{i}.layer.bias"]) + p64[..., i * H:(i + 1) * H])
    sigma = torch.nn.functional.linear(x, prm["final_layer.weight"], prm["final_layer.bias"])
    c = torch.sin(fr[..., -H:] * torch.nn.functional.linear(torch.cat([t64(dirs), x], -1), prm["color_layer_sine.layer.weight"], prm["color_layer_sine.layer.bias"]) + p64[..., -H:])
    ref = torch.cat([torch.sigmoid(torch.nn.functional.linear(c, prm["color_layer_linear.0.weight"], prm["color_layer_linear.0.bias"])), sigma], -1)
    (ref * t64(w)).sum().backward()
    want = dict(out=ref.detach().numpy(), f=f64.grad.numpy(), p=p64.grad.numpy(), **{n: q.grad.numpy() for n, q in prm.items()})
    assert set(want) == set(res["native"]) == set(res["torch"])
    errs = {r: {k: _rel_err(res[r][k], want[k]) for k in want} for r in res}
    wn, wt = max(errs["native"], key=errs["native"].get), max(errs["torch"], key=errs["torch"].get)
    print(f"[parity] per-point-modulated SIREN backward H={H} B={B} P={P}: worst relative error vs fp64 autograd over {len(want)} tensors -- native "
          f"{errs['native'][wn]:.2e} ({wn}), PyTorch-ROCm route {errs['torch'][wt]:.2e} ({wt}); outputs {errs['native']['out']:.1e} / {errs['torch']['out']:.1e}")
    # measured (round 6): native 5.6e-5 / 5.2e-5 / 5.5e-5, the PyTorch-ROCm route 7.4e-5 / 7.3e-5 / 5.7e-5 (H = 32 / 64 / 256); x 1.5
    assert errs["native"][wn] <= 8.5e-5 and errs["native"]["out"] <= 4.5e-5, errs["native"]


def test_pointwise_siren_backward_at_scale_native_vs_pytorch_route():
    """The whole differentiable SPATIALSIRENGRID call at the size of a render pass (H = 256, 1 x 65,536 points: latent grid from z, per-point
    latents, per-point mapping network, per-point-modulated SIREN): the native route (PointwiseSirenFunction) against the PyTorch-ROCm route
    on the same inputs -- outputs and every gradient (SIREN weights, mapping network, z through StyleGenerator2D) --, its step time and
    peak memory beside the other's, and a second step after an optimizer update (the native model re-packs on the device)."""
    import time
    torch.manual_seed(3)
    mod = S.SPATIALSIRENGRID(input_dim=3, z_dim=64, hidden_dim=256, output_dim=4).to(DEV).train()
    mod.device = torch.device(DEV)
    P = 65536
    pts = torch.rand((1, P, 3), device=DEV) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn((1, P, 3), device=DEV), dim=-1)
    z0 = torch.randn((1, 64), device=DEV)
    w = torch.randn((1, P, 4), device=DEV) / P
    res, ms = {}, {}
    for route in ("native", "torch"):
        mod.NATIVE_POINTWISE_BACKWARD = route == "native"
        for it in range(3):
            mod.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = mod(pts, z, dirs)
            (out * w).sum().backward()
            torch.cuda.synchronize()
            ms[route] = (time.perf_counter() - t0) * 1e3
        res[route] = dict(out=N_(out), z=N_(z.grad), **{n: N_(q.grad) for n, q in mod.named_parameters() if q.grad is not None})
    assert set(res["native"]) == set(res["torch"]) and len(res["native"]) > 40
    errs = {k: _rel_err(res["native"][k], res["torch"][k]) for k in res["torch"]}
    worst = max(errs, key=errs.get)
    print(f"[parity] SPATIALSIRENGRID step at 65,536 points, H = 256: native vs PyTorch-ROCm route worst relative difference over {len(errs)} tensors {errs[worst]:.2e} "
          f"({worst}); outputs {errs['out']:.1e}; step {ms['native']:.1f} ms native / {ms['torch']:.1f} ms PyTorch-ROCm")
    # two fp32 evaluations, each ~5e-5 from fp64 (tests above); the worst tensor sits behind the StyleGAN2 latent-grid network, whose
    # PyTorch-ROCm backward accumulates atomically: 4.8e-5 / 5.5e-6 and 8.4e-5 / 7.4e-6 in two runs of the same library (round 6)
    assert errs[worst] <= 2e-4 and errs["out"] <= 2e-5, errs
    # an optimizer step, then the native route again: the device-side re-pack is picked up (pack_generation)
    mod.NATIVE_POINTWISE_BACKWARD = True
    opt = torch.optim.SGD([q for n, q in mod.named_parameters() if mod._is_render_param(n)], lr=1e-2)
    opt.step()
    mod.zero_grad(set_to_none=True)
    out2 = mod(pts, z0, dirs)
    (out2 * w).sum().backward()
    assert np.abs(N_(out2) - res["native"]["out"]).max() > 1e-6 and all(q.grad is not None for n, q in mod.named_parameters() if mod._is_render_param(n))


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
def test_split_backward_equals_the_single_node_backward(precision):
    """generators/autograd.py, round 4: with siren.split_backward the hierarchical render is two autograd nodes (render stage: composite
    backward + every chain + the grid gradient; weight stage: every weight-gradient launch) so that DistributedDataParallel can all-reduce
    the grid gradient beside the weight-gradient kernels.  Same kernels on the same chunks: pixels bit-identical, every gradient equal to
    the single-node backward's up to the order of the grid scatter's float atomics."""
    from fenerf_amd.siren import autograd as SA
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0, precision=precision)
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N = 2, 7, 11
    film = proc.film_params(spec, B, seed=4)
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
    res = []
    old, SA.BACKWARD_CHUNK_POINTS = SA.BACKWARD_CHUNK_POINTS, 256          # three point ranges per (pass, image): 12 backward chunks
    try:
        # keep: how many of the last chunks leave their dump for the weight stage (fenerf_render_backward_stage, round 5) -- one, the default
        # two, some, all of them
        for split, keep in ((False, None), (True, 1), (True, 2), (True, 5), (True, 100)):
            mod.split_backward = split
            if keep is not None:
                mod.split_keep_chunks = keep
            film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
            for p_ in mod.parameters():
                p_.grad = None
            torch.manual_seed(11)
            px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
            route, todo = [], [px.grad_fn]          # the autograd nodes behind the pixels (the NCHW / * 2 - 1 epilogue sits in front of the render node)
            while todo and len(route) < 64:
                f = todo.pop()
                route.append(type(f).__name__)
                todo.extend(g_ for g_, _ in f.next_functions if g_ is not None)
            assert any("HierarchicalRenderSplit" in r_ for r_ in route) == split, route      # the two-node route ran iff it was asked for
            w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            (px * w).sum().backward()
            g = {k: N_(v.grad) for k, v in film_t.items()}
            g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
            res.append((N_(px), g))
    finally:
        SA.BACKWARD_CHUNK_POINTS = old
        mod.split_backward = False
        if hasattr(mod, "split_keep_chunks"):
            del mod.split_keep_chunks
    px0, g0 = res[0]
    worst, exact = 0.0, 0
    for px1, g1 in res[1:]:
        assert np.array_equal(px0, px1) and g0.keys() == g1.keys() and len(g0) > 30
        worst = max(worst, max(_rel_err(g1[k], g0[k]) for k in g0))
        # every gradient but the atomically scattered grid's: the same kernels on the same chunks, summed in the same order -- bit for bit
        exact += all(np.array_equal(g1[k], g0[k]) for k in g0 if k != "spatial_embeddings")
    print(f"[parity] split (two-node) backward vs the single node [{precision}], 1 / 2 / 5 / all of 12 chunks kept: pixels bit-identical, worst relative "
          f"gradient difference over {len(g0)} tensors {worst:.1e}; all gradients but the grid's bit-identical in {exact} of {len(res) - 1} runs")
    assert worst <= 2e-6 and exact == len(res) - 1


@pytest.mark.parametrize("z_dim,hidden,out_dim,n_blocks,B", [(256, 256, 4096, 3, 1), (256, 256, 1536, 3, 6), (16, 256, 704, 3, 2), (32, 256, 64, 1, 5),
                                                             (8, 32, 40, 3, 64), (100, 300, 1000, 2, 3),
                                                             (514, 256, 96, 2, 2),       # z_dim > hidden, z_dim % 4 != 0: LDS buffer alignment (round-4 advisory)
                                                             (100, 301, 96, 2, 3), (7, 33, 10, 1, 2)])   # hidden > z_dim, hidden % 4 != 0: the launcher's LDS size (round-5 verdict)
def test_mapping_network_native_vs_torch(z_dim, hidden, out_dim, n_blocks, B):
    """CustomMappingNetwork (siren.py:82-102) at small batch runs as one native launch forward and three backward (fenerf_mapping.hip)
    instead of ~9 + ~30 ATen launches per network.  Same module, both routes: outputs and every weight / bias gradient against the
    PyTorch ops (nn.Sequential) on the same parameters -- fp32 sums in another order, nothing else."""
    torch.manual_seed(z_dim + out_dim + B)
    net = S.CustomMappingNetwork(z_dim, hidden, out_dim, n_blocks=n_blocks).to(DEV)
    z = torch.randn(B, z_dim, device=DEV)
    w = torch.randn(B, out_dim, device=DEV)
    assert net._native_ok(z)
    f, p = net(z)
    out = torch.cat([f, p], -1)
    assert out.requires_grad and "_MappingFunction" in str(f.grad_fn) and f.grad_fn is p.grad_fn, "the native route ran (both halves are its outputs)"
    (out * w).sum().backward()
    got = {n: N_(q.grad) for n, q in net.named_parameters()}
    for q in net.parameters():
        q.grad = None
    ref = net.network(z)
    (ref * w).sum().backward()
    e_out = _rel_err(N_(out), N_(ref))
    errs = {n: _rel_err(got[n], N_(q.grad)) for n, q in net.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"[parity] mapping network {z_dim}->{hidden}x{n_blocks + 1}->{out_dim}, batch {B}: native vs PyTorch ops output {e_out:.1e}, worst gradient {errs[worst]:.1e} ({worst})")
    assert e_out <= 2e-6 and errs[worst] <= 1e-5
    with torch.no_grad():                                      # inference: the same launch, no graph
        f2, p2 = net(z)
    assert torch.equal(torch.cat([f2, p2], -1), out.detach())
    big = torch.randn(net.NATIVE_MAX_BATCH + 1, z_dim, device=DEV)     # large batches stay rocBLAS GEMMs
    assert not net._native_ok(big) and torch.equal(torch.cat(net(big), -1), net.network(big))


@pytest.mark.parametrize("n_layers,H,n_lab", [(3, 256, 18), (2, 256, 18), (3, 32, 18), (2, 100, 1), (3, 96, 32), (1, 64, 18)])
def test_label_head_backward_native_vs_autograd(n_layers, H, n_lab):
    """fenerf_label_head_backward (round 5: one / two launches) against what it replaces -- torch autograd through the label head's fold
    (label_layer_linear, siren.py:1490-1494), in fp64 -- and against the torch products of rounds 2-4 (11 launches)."""
    from fenerf_amd.siren import autograd as SA
    g = torch.Generator(device="cpu").manual_seed(n_layers * 1000 + H + n_lab)
    dims = [H] * n_layers + [n_lab]
    params = [(torch.randn(dims[i + 1] if i == n_layers - 1 else H, H, generator=g).mul_(H ** -0.5).to(DEV), torch.randn(dims[i + 1] if i == n_layers - 1 else H, generator=g).mul_(0.3).to(DEV))
              for i in range(n_layers)]
    gA, gc = torch.randn(n_lab, H, generator=g).to(DEV), torch.randn(n_lab, generator=g).to(DEV)
    got = native.label_head_backward(params, gA, gc)
    p64 = [(W.double().requires_grad_(True), b.double().requires_grad_(True)) for W, b in params]
    A, c = SA._fold_label_head(p64)
    ((A * gA.double()).sum() + (c * gc.double()).sum()).backward()
    old = SA._fold_label_head_backward(params, gA, gc)
    worst, worst_old = 0.0, 0.0
    for (dW, db), (W64, b64), (oW, ob) in zip(got, p64, old):
        assert dW.shape == W64.shape and db.shape == b64.shape
        worst = max(worst, _rel_err(N_(dW), N_(W64.grad)), _rel_err(N_(db), N_(b64.grad)))
        worst_old = max(worst_old, _rel_err(N_(oW), N_(W64.grad)), _rel_err(N_(ob), N_(b64.grad)))
    print(f"[parity] label head backward, {n_layers} layers H={H} n_lab={n_lab}: native vs fp64 autograd {worst:.1e} (the torch products it replaces: {worst_old:.1e})")
    assert worst <= 2e-6


def test_label_head_backward_refuses_what_it_cannot_do():
    W, b = torch.zeros(40, 64, device=DEV), torch.zeros(40, device=DEV)
    with pytest.raises(_lib.FenerfError, match="n_lab <= 32"):
        native.label_head_backward([(torch.zeros(64, 64, device=DEV), torch.zeros(64, device=DEV)), (W, b)], torch.zeros(40, 64, device=DEV), torch.zeros(40, device=DEV))
    with pytest.raises(ValueError, match="layers must be"):
        native.label_head_backward([(torch.zeros(64, 32, device=DEV), torch.zeros(64, device=DEV)), (torch.zeros(18, 64, device=DEV), torch.zeros(18, device=DEV))],
                                   torch.zeros(18, 64, device=DEV), torch.zeros(18, device=DEV))


def test_image_layout_function_is_the_references_epilogue_bit_for_bit():
    """generators.py:519-521 -- reshape, permute(0, 3, 1, 2).contiguous(), * 2 - 1 -- as one launch each way (ImageLayoutFunction)"""
    from fenerf_amd.generators.autograd import ImageLayoutFunction
    torch.manual_seed(2)
    px = torch.rand(3, 16 * 16, 21, device=DEV, requires_grad=True)
    w = torch.randn(3, 21, 16, 16, device=DEV)
    out = ImageLayoutFunction.apply(px, 3, 16)
    ref = px.reshape(3, 16, 16, -1).permute(0, 3, 1, 2).contiguous() * 2 - 1
    assert out.is_contiguous() and torch.equal(out, ref)
    g, = torch.autograd.grad((out * w).sum(), px, retain_graph=True)
    g_ref, = torch.autograd.grad((ref * w).sum(), px, retain_graph=True)
    assert torch.equal(g, g_ref)
    g2, = torch.autograd.grad((out * w).permute(0, 1, 3, 2).sin().sum(), px)          # a non-contiguous incoming gradient
    g2_ref, = torch.autograd.grad((ref * w).permute(0, 1, 3, 2).sin().sum(), px)
    assert torch.equal(g2, g2_ref)


@pytest.mark.parametrize("precision", FORWARD_PRECISIONS)
def test_style_generator3d_vs_reference(precision):
    """StyleGenerator3d (generators.py:914-1294; round 5 -- the class round 4's review listed as absent): forward(z) and staged_forward(z)
    against the reference class's own outputs on recorded draws (tests/golden/tiny_style_generator.npz); staged_forward ignores psi and
    fill_color, set_device draws nothing, there are no average frequencies."""
    g = load_golden("tiny_style_generator")
    spec = spec_from_golden(g)
    gen = G.StyleGenerator3d(functools.partial(S.SPATIALSIRENBASELINE, hidden_dim=32), spec["z_dim"], spec["output_dim"])
    sd = proc.make_state_dict(dict(spec, map_hidden=256), seed=int(g["meta_seed"]), sigma_gain=float(g["meta_sigma_gain"]))
    gen.siren.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    gen = gen.to(DEV).eval()
    gen.siren.precision = precision
    gen.draws = VR.RecordedDraws([])               # set_device must not draw: an empty recording would raise
    gen.set_device(torch.device(DEV))
    assert not hasattr(gen, "avg_frequencies")
    with pytest.raises(AttributeError):
        gen.generate_avg_frequencies()
    z = T(g["z"])
    kw = dict(img_size=6, num_steps=6, hierarchical_sample=True, clamp_mode="relu", nerf_noise=0.5, white_back=True, fov=12, ray_start=0.88,
              ray_end=1.12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, sample_dist="gaussian")
    order = ("u_jitter", "r_theta", "r_phi", "noise_coarse", "u_fine", "noise_fine")
    gen.draws = VR.RecordedDraws([g["fwd_rand_" + k] for k in order])
    with torch.no_grad():
        px, poses = gen(z, **kw)
    np.testing.assert_allclose(N_(poses), g["fwd_poses"], atol=1e-6)
    e_f = np.abs(N_(px) - g["fwd_pixels"]).max()
    res = []
    for psi, colour in ((float(g["stg_psi"]), "white"), (1.0, "black")):        # both ignored by this class
        gen.draws = VR.RecordedDraws([g["stg_rand_" + k] for k in order])
        res.append(gen.staged_forward(z, psi=psi, max_batch_size=97, fill_mode="weight", fill_color=colour, **kw))
    (px_s, depth, third), (px_s2, depth2, third2) = res
    assert torch.equal(px_s, px_s2) and torch.equal(depth, depth2) and torch.equal(third, third2)
    assert px_s.shape == g["stg_pixels"].shape and depth.shape == g["stg_depth"].shape and third.shape == g["stg_third"].shape
    e_s, e_d, e_t = (np.abs(N_(a) - g[k]).max() for a, k in ((px_s, "stg_pixels"), (depth, "stg_depth"), (third, "stg_third")))
    print(f"[parity] StyleGenerator3d[{precision}] vs the reference class: forward(z) {e_f:.2e}, staged_forward(z) pixels {e_s:.2e} depth {e_d:.2e} "
          f"weights_sum {e_t:.2e}; psi / fill_color ignored, no average frequencies")
    b_px = 1.7e-5 if precision == "f16x3c2" else 2.6e-6         # f16x3c2: the colour branch's two-term products (measured 1.12e-5 / 1.13e-5)
    assert e_f <= b_px and e_s <= b_px and e_d <= 3e-6 and e_t <= 6e-7          # measured x 1.5 (1.7e-6 / 1.7e-6 / 2.0e-6 / 3.6e-7, both precisions)


def test_two_renders_in_one_graph_through_the_two_stage_backward():
    """generator called twice before ONE backward with siren.split_backward on: the render stage of the second render runs before the weight
    stage of the first (autograd orders by topology), so the two passes' two-stage workspaces are alive together -- the second gets its
    own (native.render_backward_stage) -- and the gradients are the sum of the two single-render backward passes'."""
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0, precision="f16x3")
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N = 1, 6, 8
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.0, last_back=False)
    films = [proc.film_params(spec, B, seed=s_) for s_ in (4, 9)]
    w = torch.randn((B, 21, S_, S_), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))

    def render(film, seed):
        torch.manual_seed(seed)
        px, _ = gen.forward_with_frequencies(T(film["freq_geo"]), T(film["freq_app"]), T(film["phase_geo"]), T(film["phase_app"]), **kw)
        return (px * w).sum()

    def grads():
        return {k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None}

    try:
        mod.split_backward = True
        singles = []
        for film, seed in zip(films, (11, 12)):
            for p_ in mod.parameters():
                p_.grad = None
            render(film, seed).backward()
            singles.append(grads())
        for p_ in mod.parameters():
            p_.grad = None
        (render(films[0], 11) + render(films[1], 12)).backward()
        both = grads()
    finally:
        mod.split_backward = False
    nat = mod.native_differentiable(torch.device(DEV))
    assert not getattr(nat, "_split_ws_busy", False), "the persistent two-stage workspace was released"
    worst = max(_rel_err(both[k], singles[0][k] + singles[1][k]) for k in both)
    print(f"[parity] two renders in one graph through the two-stage backward: gradients = the sum of the two single backward passes', {worst:.1e} over {len(both)} tensors")
    assert len(both) > 30 and worst <= 2e-6


def test_readme_python_example_runs_as_written():
    """README.md's python block: the curriculum generator's staged_forward, then the bare radiance field differentiated wrt its sample
    positions (surface normals) -- executed as it stands."""
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "README.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    assert blocks
    ns = {}
    exec(compile(blocks[0], "README.md", "exec"), ns)
    assert tuple(ns["img"].shape) == (1, 22, 128, 128) and tuple(ns["depth"].shape) == (1, 128, 128)
    n = ns["normals"]
    assert tuple(n.shape) == (1, 4096, 3) and bool(torch.isfinite(n).all()) and float((n.norm(dim=-1) - 1).abs().max()) < 1e-4
    # the same gradient with every weight frozen (the FiLM-only backward + the dump): equal up to summation order
    gen, pts = ns["gen"], ns["pts"]
    assert pts.grad is None
    print(f"[parity] README example: staged_forward {tuple(ns['img'].shape)}, normals of {n.shape[1]} points finite and unit length")


def test_fid_image_dump_two_ranks_on_one_gpu(tmp_path):
    """tools/dump_images.py (the reference's FID image dump, fid_evaluation.py:126-150 -- its multi-GPU forward path: shard by image
    id, no data-path collective) as the launcher runs it with two ranks, both on cuda:0 over gloo: the ids 0 .. 11 exist once each,
    128 x 128 JPEGs; and in process, rank 0 of 2's first batch equals generator.staged_forward on the same draws."""
    import json
    import subprocess
    from PIL import Image
    from conftest import ROOT
    from fenerf_amd import callers, curriculums
    sys.path.insert(0, ROOT)
    import bench
    g = load_golden("tiny_multiview")
    ckpt = _tiny_checkpoint_dir(tmp_path)
    cur = json.loads(str(g["curriculum_json"]))
    cur.update(output_dim=22, eval_last_back=False)
    cur_file = str(tmp_path / "tiny_curriculum.json")
    json.dump(cur, open(cur_file, "w"))
    out = str(tmp_path / "generated")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(bench._free_port()), os.path.join(ROOT, "tools", "dump_images.py"), ckpt, "--curriculum", cur_file, "--num_imgs", "10",
           "--output_dir", out, "--one_device", "--dist_backend", "gloo", "--seed", "5"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    names = sorted(os.listdir(out))
    assert names == [f"{i:05d}.jpg" for i in range(16)]        # two ranks x two whole batches of 4: ids 0 .. 15 (the reference overshoots the same way)
    ims = [np.asarray(Image.open(os.path.join(out, n))) for n in names]
    assert all(im.shape == (128, 128, 3) for im in ims) and not np.array_equal(ims[0], ims[1])
    assert "rank 0 of 2: 8 images" in r.stdout and "rank 1 of 2: 8 images" in r.stdout

    cur_i = {(int(k[4:]) if k.startswith("int:") else k): v for k, v in cur.items()}
    md = curriculums.extract_metadata(cur_i, 100000)
    md["nerf_noise"] = 0.0
    gen = callers.load_generator(ckpt, DEV, reset_render_options=False)
    got = []
    torch.manual_seed(5)
    callers.output_images_double(gen, md, 0, 2, str(tmp_path / "x"), num_imgs=1, save=lambda img, path: got.append(img.clone()))
    torch.manual_seed(5)
    fmd = callers.fid_dump_metadata(md)
    z_geo = torch.randn((4, gen.z_geo_dim), device=gen.device)
    z_app = torch.randn((4, gen.z_app_dim), device=gen.device)
    ref = gen.staged_forward(z_geo, z_app, **fmd)[0]
    assert len(got) == 4 and all(torch.equal(a, b[-3:]) for a, b in zip(got, ref))
    # the JPEG rank 0 wrote first is that image (8-bit, lossy: within a few grey levels on average)
    want = ((ref[0, -3:].clamp(-1, 1) + 1) / 2 * 255).permute(1, 2, 0).cpu().numpy()
    print(f"[parity] FID image dump, two ranks on one GPU: 16 JPEGs; rank 0's first file vs staged_forward on the same draws: mean |diff| {np.abs(ims[0] - want).mean():.2f} grey levels")
    assert np.abs(ims[0] - want).mean() < 4.0


def test_training_snapshots_under_autocast_with_the_ema_swap(tmp_path):
    """callers.training_snapshots = the sample-image block of the reference's training loop (train_double_latent_semantic.py:464-523):
    staged_forward under autocast, live and EMA weights (store / copy_to / eval ... restore through param.data-free copies), ten 5 x 5
    grids.  The EMA grids are the renders of the EMA weights, the live weights come back bit for bit, and the next render uses them."""
    import json
    from PIL import Image
    from fenerf_amd import callers, curriculums, ema as ema_mod
    g = load_golden("tiny_multiview")
    ckpt = _tiny_checkpoint_dir(tmp_path)
    cur = json.loads(str(g["curriculum_json"]))
    cur.update(output_dim=22, eval_last_back=False)
    cur_i = {(int(k[4:]) if k.startswith("int:") else k): v for k, v in cur.items()}
    md = curriculums.extract_metadata(cur_i, 100000)
    md["nerf_noise"] = 0.0
    gen = callers.load_generator(ckpt, DEV, use_ema=False, reset_render_options=False)
    gen.train()
    ema = ema_mod.ExponentialMovingAverage(gen.parameters(), decay=0.999)
    with torch.no_grad():
        for s_ in ema.shadow_params:             # an average that differs from the live weights
            s_.mul_(0.9)
    before = [p.detach().clone() for p in gen.parameters()]
    torch.manual_seed(0)
    zg, za = torch.randn(25, gen.z_geo_dim), torch.randn(25, gen.z_app_dim)          # on the CPU, as the training loop keeps them (:113-114)
    torch.manual_seed(1)
    paths = callers.training_snapshots(gen, ema, zg, za, md, str(tmp_path / "snap"), step=7000)
    names = [os.path.basename(p) for p in paths]
    assert names == [f"7000_{k}_{t}.png" for t in ("fixed", "tilted", "fixed_ema", "tilted_ema", "random") for k in ("seg", "img")]
    ims = {n: np.asarray(Image.open(p)) for n, p in zip(names, paths)}
    assert all(im.shape == (5 * 130 + 2, 5 * 130 + 2, 3) for im in ims.values())
    assert all(torch.equal(a, b) for a, b in zip(before, gen.parameters())) and not gen.training
    assert not np.array_equal(ims["7000_img_fixed.png"], ims["7000_img_fixed_ema.png"]) and not np.array_equal(ims["7000_img_fixed.png"], ims["7000_img_tilted.png"])
    # the grids are what staged_forward gives for those weights: live weights now (restored), EMA weights after copy_to
    def grid_of(**over):
        opts = dict(md, h_stddev=0, v_stddev=0, img_size=128, **over)
        with torch.no_grad(), torch.autocast("cuda"):
            px = gen.staged_forward(zg.to(DEV), za.to(DEV), **opts)[0]
        from fenerf_amd import imageio_lite
        return imageio_lite.to_uint8_hwc(imageio_lite.make_grid(px[:25, -3:], nrow=5, normalize=True))
    torch.manual_seed(1)                         # replay the block's draws: frontal, tilted, (EMA swap) frontal
    live, tilted = grid_of(), grid_of(h_mean=md["h_mean"] + 0.5)
    assert np.abs(live.astype(int) - ims["7000_img_fixed.png"].astype(int)).max() <= 1
    assert np.abs(tilted.astype(int) - ims["7000_img_tilted.png"].astype(int)).max() <= 1
    ema.copy_to(gen.parameters())
    gen.eval()
    avg = grid_of()
    d = np.abs(avg.astype(int) - ims["7000_img_fixed_ema.png"].astype(int)).max()
    print(f"[parity] training snapshots: 10 grids of 25 x 128 x 128 under autocast; live grids reproduced, EMA grid within {d} grey level(s)")
    assert d <= 1


@pytest.mark.parametrize("precision", PRECISIONS + ["tape16"])
@pytest.mark.parametrize("case", [dict(kind="texture", H=32, grid=5, B=2, S=7, N=11, kw=dict(clamp_mode="relu", nerf_noise=0.2, last_back=False)),
                                  dict(kind="texture", H=64, grid=6, B=1, S=9, N=12, kw=dict(clamp_mode="relu", nerf_noise=0.0, last_back=True, white_back=True)),
                                  dict(kind="baseline", H=32, grid=0, B=3, S=6, N=8, kw=dict(clamp_mode="softplus", nerf_noise=0.0, lock_view_dependence=True)),
                                  dict(kind="texture", H=256, grid=8, B=1, S=16, N=24, kw=dict(clamp_mode="relu", nerf_noise=0.0)),
                                  # without importance resampling (the reference's inversion renders): SparseSinglePassRenderFunction
                                  dict(kind="texture", H=64, grid=6, B=2, S=9, N=12, kw=dict(clamp_mode="relu", nerf_noise=0.3, hierarchical_sample=False)),
                                  dict(kind="texture", H=256, grid=8, B=1, S=16, N=24, kw=dict(clamp_mode="relu", nerf_noise=0.0, last_back=True,
                                                                                               hierarchical_sample=False, lock_view_dependence=True))])
def test_sparse_backward_equals_the_dense_backward(case, precision):
    """siren.sparse_backward (generators/autograd.py SparseHierarchicalRenderFunction): the backward runs only over the samples whose row
    of upstream gradients is not all zero -- under the relu clamp every sample with sigma + noise <= 0 has an all-zero row (weight 0,
    relu' = 0; volumetric_rendering.py:36-47), and torch autograd multiplies those zeros through the whole network.  Pixels bit-identical
    to the dense node's, every gradient equal up to the order of the sums; with the softplus clamp nothing is dropped and it still agrees."""
    from fenerf_amd.generators import autograd as GA
    kind, H = case["kind"], case["H"]
    mod, spec, sd = _siren_module(kind, H, case["grid"], sigma_gain=150.0, precision=precision)
    cls = {"texture": S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, "baseline": S.SIRENBASELINESEMANTICDISENTANGLE}[kind]
    gen = G.DoubleImplicitGenerator3d(functools.partial(cls, hidden_dim=H), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B, S_, N = case["B"], case["S"], case["N"]
    film = proc.film_params(spec, B, seed=4)
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian")
    kw.update(case["kw"])
    res = []
    try:
        for sparse in (False, True):
            mod.sparse_backward = sparse
            GA.SparseHierarchicalRenderFunction.last_kept = None
            film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
            for p_ in mod.parameters():
                p_.grad = None
            torch.manual_seed(11)
            px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
            w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            (px * w).sum().backward()
            g = {k: N_(v.grad) for k, v in film_t.items()}
            g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
            kept = GA.SparseHierarchicalRenderFunction.last_kept
            res.append((N_(px), g, None if kept is None else (int(kept[0]), kept[1])))
        GA.SparseHierarchicalRenderFunction.verify()      # the deferred check of the buffer bound (raises if a backward dropped samples)
    finally:
        mod.sparse_backward = False
    (px0, g0, k0), (px1, g1, k1) = res
    assert k0 is None and k1 is not None and np.array_equal(px0, px1) and g0.keys() == g1.keys() and len(g0) > 25
    errs = {k: _rel_err(g1[k], g0[k]) for k in g0}
    worst = max(errs, key=errs.get)
    print(f"[parity] sparse backward vs dense [{precision}] {kind} H={H} {case['kw']['clamp_mode']}{'' if kw['hierarchical_sample'] else ' one pass'}: {k1[0]} of {k1[1]} samples kept ({100 * k1[0] / k1[1]:.1f} %), "
          f"pixels bit-identical, worst relative gradient difference over {len(g0)} tensors {errs[worst]:.1e} ({worst})")
    if case["kw"]["clamp_mode"] == "softplus":
        assert k1[0] == k1[1]
    else:
        assert k1[0] < k1[1]
    # two fp32 evaluations of the same sums in different orders (the tape of a kept sample is re-evaluated from the same inputs)
    # measured 2.0e-7 .. 5.1e-7; 2.4e-6 on final_layer.bias (the sum of d sigma over all samples, heavy cancellation) of the one-pass H = 256 case
    assert errs[worst] <= 1e-5, (worst, errs[worst])


@pytest.mark.parametrize("N", [7, 32, 129])
def test_create_samples_on_the_device_equals_the_host(N):
    """callers.create_samples(device=...) -- the voxel centres of extract_double_semantic_shapes.py:13-35 built on the device instead of on
    the host + a 200-MB copy: int64 -> float, IEEE division, fmod, one multiply and one add, bit for bit the host's values."""
    from fenerf_amd import callers
    host, vo, vs = callers.create_samples(N, (0.01, -0.02, 0.03), 0.3)
    devs, vo2, vs2 = callers.create_samples(N, (0.01, -0.02, 0.03), 0.3, device=torch.device(DEV))
    assert devs.is_cuda and np.array_equal(vo, vo2) and vs == vs2
    assert torch.equal(devs.cpu(), host)


@pytest.mark.parametrize("B,R,N,C,images", [(1, 49, 11, 22, None), (3, 37, 7, 21, None), (4, 16, 24, 22, [2, 0]), (2, 300, 3, 4, [1]), (1, 1, 1, 5, None),
                                            (2, 16384, 12, 22, None)])        # 1,536 blocks of 256 samples per image: two pieces of the prefix-sum pass
def test_sparse_select_against_torch(B, R, N, C, images):
    """fenerf_sparse_select (include/fenerf.h) against the torch statements it replaced: per image the samples with a non-zero gradient row
    (NaN counts) in sample order, coarse pass first; their points origins + dirs * z, directions and rows; pad slots = the image's first
    sample with a zero row; counts; the overflow flag when cap is too small.  Odd / even C, P not a multiple of 256, an image list."""
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + R)
    P = R * N
    rnd = lambda *s: torch.randn(s, device=DEV, generator=g)
    d_c, d_f = rnd(B * R, N, C), rnd(B * R, N, C)
    for d in (d_c, d_f):
        d[torch.rand((B * R, N), device=DEV, generator=g) < 0.7] = 0.0           # 70 % all-zero rows
    d_c[0, 0, C - 1] = float("nan")                                               # a broken row is kept
    d_f.view(-1, C)[-1] = 0.0
    d_f.view(-1, C)[-1, 0] = -0.0                                                  # -0.0 is zero
    zc, zf, o, dr = rnd(B * R, N), rnd(B * R, N), rnd(B, R, 3), rnd(B, R, 3)
    ids = list(range(B)) if images is None else images
    d_all = torch.cat([d_c.reshape(B, P, C), d_f.reshape(B, P, C)], 1)
    keep = (d_all != 0).any(-1)
    z_all = torch.cat([zc.reshape(B, P), zf.reshape(B, P)], 1)
    want_counts = [int(keep[b].sum()) for b in ids]
    cap = max(32, (max(want_counts) + 31) // 32 * 32)
    idx = None if images is None else torch.tensor(images, dtype=torch.long, device=DEV)
    pts, rd, d_sel, counts = native.sparse_select(d_c, d_f, zc, zf, o, dr, cap, images=idx)
    assert pts.shape == (len(ids), cap, 3) and rd.shape == pts.shape and d_sel.shape == (len(ids), cap, C)
    assert counts.tolist() == want_counts + [0]
    for j, b in enumerate(ids):
        sel = torch.nonzero(keep[b]).flatten()
        sel = torch.cat([sel, sel.new_zeros(cap - sel.numel())])                  # pad slots: sample 0
        ray = (sel % P) // N
        want_pts = o[b, ray] + dr[b, ray] * z_all[b, sel].unsqueeze(-1)
        want_d = d_all[b, sel]
        want_d[want_counts[j]:] = 0
        assert torch.equal(pts[j], want_pts) and torch.equal(rd[j], dr[b, ray])
        assert torch.equal(torch.nan_to_num(d_sel[j], nan=7.0), torch.nan_to_num(want_d, nan=7.0))
    pts2, rd2, _, counts2 = native.sparse_select(d_c, d_f, zc, zf, o, dr, cap, want_dirs=False, images=idx)
    assert rd2 is None and torch.equal(pts2, pts) and torch.equal(counts2, counts)
    if max(want_counts) > 32:        # a cap below the fullest image's count: flag raised, the first `cap` kept samples delivered
        small = (max(want_counts) - 1) // 32 * 32
        pts3, _, d3, counts3 = native.sparse_select(d_c, d_f, zc, zf, o, dr, small, images=idx)
        assert counts3.tolist() == want_counts + [1]
        j = int(np.argmax(want_counts))
        assert torch.equal(pts3[j], pts[j, :small])


@pytest.mark.parametrize("film_only", [False, True], ids=["all_gradients", "film_only"])
def test_sparse_backward_in_launch_groups_of_similar_images(film_only, monkeypatch):
    """A batch whose images keep very different numbers of samples is walked in launch groups (generators/autograd.py plan_sparse_groups,
    fenerf_sparse_select's image list), each padded to its own fullest image.  Here the plan is forced to split (one 128-point workgroup
    per round, no cost per group): five images with a density bias each (FiLM phase of the last geometry layer shifted), so their kept
    counts differ; pixels bit-identical to the dense node, gradients equal to the order of the sums, FiLM gradient rows in image order."""
    from fenerf_amd.generators import autograd as GA
    plan = GA.plan_sparse_groups
    monkeypatch.setattr(GA, "plan_sparse_groups", lambda caps, n_cus: plan(caps, 1, 0.0))
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0, precision="f16x3")
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    B = 5
    film = proc.film_params(spec, B, seed=4)
    kw = dict(img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
              hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.0)
    if film_only:
        for p_ in mod.parameters():
            p_.requires_grad_(False)
    res = []
    try:
        for sparse in (False, True):
            mod.sparse_backward = sparse
            film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
            for p_ in mod.parameters():
                p_.grad = None
            torch.manual_seed(11)
            px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
            w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            (px * w).sum().backward()
            g = {k: N_(v.grad) for k, v in film_t.items()}
            g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
            res.append((N_(px), g))
        GA.SparseHierarchicalRenderFunction.verify()
        groups = GA.SparseHierarchicalRenderFunction.last_groups
        kept = GA.SparseHierarchicalRenderFunction.last_kept
    finally:
        mod.sparse_backward = False
        for p_ in mod.parameters():
            p_.requires_grad_(True)
    (px0, g0), (px1, g1) = res
    assert np.array_equal(px0, px1) and g0.keys() == g1.keys() and len(g0) == (4 if film_only else 37)
    errs = {k: _rel_err(g1[k], g0[k]) for k in g0}
    worst = max(errs, key=errs.get)
    print(f"[parity] sparse backward in launch groups [{'FiLM gradients only' if film_only else 'all gradients'}]: {len(groups)} groups {groups}, "
          f"{int(kept[0])} of {kept[1]} samples kept, pixels bit-identical, worst relative gradient difference over {len(g0)} tensors {errs[worst]:.1e} ({worst})")
    assert len(groups) >= 2 and sorted(b for g_, _ in groups for b in g_) == list(range(B))
    assert errs[worst] <= 2e-6, (worst, errs[worst])
    # per image: the FiLM gradient rows sit at their image's index (a permuted result would be far off row by row)
    for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"):
        for b in range(B):
            assert _rel_err(g1[k][b], g0[k][b]) <= 1e-5, (k, b)


@pytest.mark.parametrize("seed", range(10))
def test_sparse_backward_random_configurations(seed, monkeypatch):
    """Seeded random configurations of the differentiable render (batch, image size, samples per ray, hidden width, density gain, noise,
    last_back / white_back, locked view direction, precision tier, FiLM-only or all gradients, forced launch groups): the sparse node against
    the dense node -- pixels bit-identical, every gradient equal to the order of the sums.  Seeds 8 and 9 push the density bias far below
    zero: no sample of any image carries density (relu clamp), so nothing (seed 8) or only each ray's last sample (seed 9, last_back) is kept."""
    from fenerf_amd.generators import autograd as GA
    rng = np.random.default_rng(1000 + seed)
    H = int(rng.choice([32, 64, 96]))
    B, S_, N = int(rng.integers(1, 5)), int(rng.integers(3, 10)), int(rng.integers(4, 25))
    precision = str(rng.choice(PRECISIONS + ["tape16"]))
    kind = str(rng.choice(["texture", "baseline"]))
    mod, spec, sd = _siren_module(kind, H, 5 if kind == "texture" else 0, sigma_gain=float(rng.choice([1.0, 30.0, 400.0])), precision=precision)
    empty = seed in (8, 9)
    if empty:
        with torch.no_grad():
            mod.final_layer.bias.fill_(-1e5)
    cls = {"texture": S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, "baseline": S.SIRENBASELINESEMANTICDISENTANGLE}[kind]
    gen = G.DoubleImplicitGenerator3d(functools.partial(cls, hidden_dim=H), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    film = proc.film_params(spec, B, seed=seed)
    kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
              hierarchical_sample=bool(rng.integers(0, 3) > 0), sample_dist="gaussian", clamp_mode="relu", nerf_noise=float(rng.choice([0.0, 0.0, 0.3, 1.0])),
              last_back=(seed == 9) or (not empty and bool(rng.integers(0, 2))), white_back=bool(rng.integers(0, 2)),
              lock_view_dependence=bool(rng.integers(0, 2)))
    film_only = bool(rng.integers(0, 3) == 0)
    if bool(rng.integers(0, 2)):       # force the plan to split wherever padding would waste a 128-point unit
        plan = GA.plan_sparse_groups
        monkeypatch.setattr(GA, "plan_sparse_groups", lambda caps, n_cus: plan(caps, 1, 0.0))
    for p_ in mod.parameters():
        p_.requires_grad_(not film_only)
    res = []
    try:
        for sparse in (False, True):
            mod.sparse_backward = sparse
            film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
            for p_ in mod.parameters():
                p_.grad = None
            torch.manual_seed(11 + seed)
            px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
            w = torch.randn(px.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            (px * w).sum().backward()
            g = {k: N_(v.grad) for k, v in film_t.items()}
            g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
            res.append((N_(px), g))
        GA.SparseHierarchicalRenderFunction.verify()
        kept, groups = GA.SparseHierarchicalRenderFunction.last_kept, GA.SparseHierarchicalRenderFunction.last_groups
    finally:
        mod.sparse_backward = False
        for p_ in mod.parameters():
            p_.requires_grad_(True)
    (px0, g0), (px1, g1) = res
    assert np.array_equal(px0, px1) and g0.keys() == g1.keys() and len(g0) >= 4
    scale = max(float(np.abs(v).max()) for v in g0.values())
    errs = {k: float(np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-30)) for k in g0 if np.abs(g0[k]).max() > 1e-6 * scale}
    zero_ok = all(np.abs(g1[k]).max() <= 1e-5 * scale for k in g0 if k not in errs)       # a (numerically) zero gradient stays one
    worst = max(errs, key=errs.get) if errs else None
    print(f"[parity] sparse vs dense, random configuration {seed}: {kind} H={H} B={B} {S_}x{S_}x{N}+{N} [{precision}] noise {kw['nerf_noise']} "
          f"last_back {kw['last_back']} {'two passes' if kw['hierarchical_sample'] else 'one pass'} {'FiLM only' if film_only else 'all gradients'}, {int(kept[0])} of {kept[1]} samples kept in {len(groups)} group(s): "
          f"pixels bit-identical, worst relative gradient difference over {len(errs)} tensors {errs[worst] if worst else 0.0:.1e}")
    assert zero_ok and (not errs or errs[worst] <= 1e-5), (worst, errs.get(worst))        # measured <= 1.4e-6
    if seed == 8:
        assert int(kept[0]) == 0
    if seed == 9:
        assert int(kept[0]) == B * S_ * S_
    assert kept[1] == B * S_ * S_ * N * (2 if kw["hierarchical_sample"] else 1)


@pytest.mark.parametrize("clamp", ["relu", "softplus"])
def test_sparse_backward_auto_picks_the_cheaper_node(clamp, monkeypatch):
    """siren.sparse_backward = "auto" (generators/autograd.py sparse_auto_choice): the first step is a sparse one (nothing observed yet); it
    observes the fraction of the samples its buffers were sized for -- on the host, no wait.  Mostly empty space (relu clamp): sparse from
    then on.  Softplus clamp (every row non-zero, fraction 1): the dense node, with a sparse probe every SPARSE_AUTO_PROBE_EVERY-th step.
    Whatever node runs, pixels are bit-identical and gradients agree to the order of the sums."""
    from fenerf_amd.generators import autograd as GA
    monkeypatch.setattr(GA, "SPARSE_AUTO_PROBE_EVERY", 3)
    monkeypatch.setattr(GA, "SPARSE_AUTO_MIN_SAMPLES", 0)          # (this render is tiny: 3,072 samples)
    mod, spec, sd = _siren_module("texture", 32, 5, sigma_gain=150.0, precision="f16x3")
    gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=32), 8, 8, 22)
    gen.siren = mod
    gen = gen.to(DEV)
    gen.device = torch.device(DEV); gen.siren.device = gen.device
    film = proc.film_params(spec, 2, seed=4)
    kw = dict(img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=12, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
              hierarchical_sample=True, sample_dist="gaussian", clamp_mode=clamp, nerf_noise=0.0)

    def step():
        film_t = {k: T(v).requires_grad_(True) for k, v in film.items()}
        for p_ in mod.parameters():
            p_.grad = None
        torch.manual_seed(11)
        px, _ = gen.forward_with_frequencies(film_t["freq_geo"], film_t["freq_app"], film_t["phase_geo"], film_t["phase_app"], **kw)
        px.square().sum().backward()
        g = {k: N_(v.grad) for k, v in film_t.items()}
        g.update({k: N_(p_.grad) for k, p_ in mod.named_parameters() if p_.grad is not None})
        return N_(px), g

    px0, g0 = step()                                   # the dense node
    choices = []
    try:
        mod.sparse_backward = "auto"
        for _ in range(8):
            px, g = step()
            st = mod.__dict__["_sparse_auto"]
            choices.append(st["last"])
            assert np.array_equal(px, px0) and g.keys() == g0.keys()
            assert max(_rel_err(g[k], g0[k]) for k in g0) <= 2e-6
        GA.SparseHierarchicalRenderFunction.verify()
        with pytest.raises(ValueError):
            mod.sparse_backward = "sometimes"
            step()
        # below SPARSE_AUTO_MIN_SAMPLES the dense node whatever was observed (the sparse step's fixed cost is not paid back by short kernels)
        monkeypatch.setattr(GA, "SPARSE_AUTO_MIN_SAMPLES", 10 ** 9)
        mod.sparse_backward = "auto"
        px, g = step()
        assert mod.__dict__["_sparse_auto"]["last"] == "dense" and np.array_equal(px, px0)
    finally:
        mod.sparse_backward = False
    print(f"[parity] sparse_backward = 'auto' [{clamp}]: buffer fraction {st['fraction']:.3f}, nodes {choices}")
    if clamp == "relu":
        assert st["fraction"] < GA.SPARSE_AUTO_MAX_FRACTION and choices == ["sparse"] * 8      # measured: see the printed line
    else:
        assert st["fraction"] == 1.0 and choices == ["sparse", "dense", "dense", "probe", "dense", "dense", "probe", "dense"]
