#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference).

Runs only in the build container (the reference never travels).  Fixtures are data:
inputs (procedural-weight seeds, film params, every random draw the reference made,
recorded in call order) and the reference's outputs, per stage and end to end.

    python tools/make_golden.py            # writes tests/golden/
"""
import functools
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import  # noqa: E402
from fenerf_amd import procedural as proc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CURR = dict(fov=12, ray_start=0.88, ray_end=1.12, h_stddev=0.3, v_stddev=0.155,
            h_mean=np.pi * 0.5, v_mean=np.pi * 0.5, sample_dist="gaussian")


class DrawRecorder:
    """Records every torch.rand / torch.randn / torch.randperm call (shape, values) in order."""

    def __enter__(self):
        self.draws = []
        self._orig = (torch.rand, torch.randn, torch.randperm, torch.randn_like)

        def wrap(fn, kind):
            def inner(*a, **k):
                t = fn(*a, **k)
                self.draws.append((kind, t.detach().cpu().numpy().copy()))
                return t
            return inner

        torch.rand, torch.randn, torch.randperm = wrap(torch.rand, "rand"), wrap(torch.randn, "randn"), wrap(torch.randperm, "randperm")
        torch.randn_like = wrap(torch.randn_like, "randn")      # same generator consumption as randn(x.shape)
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn, torch.randperm, torch.randn_like = self._orig


class CallRecorder:
    """Wraps callables in a module namespace and records (args, result) per call."""

    def __init__(self, module, names):
        self.module, self.names = module, names
        self.calls = {n: [] for n in names}

    def __enter__(self):
        self._orig = {n: getattr(self.module, n) for n in self.names}
        for n in self.names:
            def mk(n, fn):
                def inner(*a, **k):
                    r = fn(*a, **k)
                    self.calls[n].append((a, k, r))
                    return r
                return inner
            setattr(self.module, n, mk(n, self._orig[n]))
        return self

    def __exit__(self, *exc):
        for n, fn in self._orig.items():
            setattr(self.module, n, fn)


def np_(t):
    return t.detach().cpu().numpy().copy() if torch.is_tensor(t) else t


def load_sd(module, sd):
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    if "spatial_embeddings" in tsd:
        module.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
    missing, unexpected = module.load_state_dict(tsd, strict=True)
    assert not missing and not unexpected


def load_state(name):
    """a state fixture written by run_trained_state -> (render weights {name: array}, raw FiLM parameters {name: array})"""
    st = np.load(os.path.join(OUT, name + ".npz"))
    return ({k[3:]: st[k] for k in st.files if k.startswith("sd_")}, {k[5:]: st[k] for k in st.files if k.startswith("film_")})


def build_ref_generator(refs, spec, seed, sigma_gain, state=None):
    """state: name of a state fixture whose render weights replace the procedural ones (the mapping networks stay procedural)"""
    siren_mod, gens, vr, cur = refs
    H = spec["hidden_dim"]
    if spec["kind"] == "texture":
        cls = functools.partial(siren_mod.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=H)
    elif spec["kind"] == "baseline":
        cls = functools.partial(siren_mod.SIRENBASELINESEMANTICDISENTANGLE, hidden_dim=H)
    else:
        cls = functools.partial(siren_mod.SPATIALSIRENBASELINE, hidden_dim=H)
    if spec["kind"] == "spatial":
        g = gens.ImplicitGenerator3d(cls, spec["z_dim"], spec["output_dim"])
    else:
        g = gens.DoubleImplicitGenerator3d(cls, spec["z_dim"], spec["z_dim"], spec["output_dim"])
    sd = proc.make_state_dict(spec, seed=seed, sigma_gain=sigma_gain)
    if state is not None:
        sd.update(load_state(state)[0])
    load_sd(g.siren, sd)
    g.eval()
    g.device = torch.device("cpu")
    g.siren.device = g.device
    return g, sd


def rand_dict_from_draws(draws, hierarchical):
    """appendix A.6 order -> oracle `rand` dict (theta/phi are pre-clamp angles built by host code)."""
    kinds = [k for k, _ in draws]
    vals = [v for _, v in draws]
    assert kinds[:3] == ["rand", "randn", "randn"], kinds
    rd = dict(u_jitter=vals[0], r_theta=vals[1], r_phi=vals[2], noise_coarse=vals[3])
    if hierarchical:
        assert kinds[3:6] == ["randn", "rand", "randn"], kinds
        rd.update(u_fine=vals[4], noise_fine=vals[5])
    return rd


def run_film_case(refs, name, spec, seed, sigma_gain, B, S, N, hier, kwargs, film_scale=1.0, stages=True, staged=False, film_kw=None, state=None):
    """forward_with_frequencies / staged_forward_with_frequencies with explicit film params.  film_kw: procedural.film_params'
    phase_rev / freq0_gain -- FiLM parameters far beyond the init range (sine arguments of hundreds of revolutions)."""
    siren_mod, gens, vr, cur = refs
    g, sd = build_ref_generator(refs, spec, seed, sigma_gain, state)
    film = proc.film_params(spec, B, seed=seed, scale=film_scale, **(film_kw or {})) if state is None else load_state(state)[1]
    tf = {k: torch.from_numpy(v) for k, v in film.items()}
    if spec["kind"] == "spatial":   # single latent: one [B, 9H] frequency / phase tensor; the colour layer uses the last H
        tf["freq_geo"] = torch.cat([tf["freq_geo"], tf["freq_app"]], -1)
        tf["phase_geo"] = torch.cat([tf["phase_geo"], tf["phase_app"]], -1)
    torch.manual_seed(1234 + seed)
    rec_names = ["fancy_integration", "sample_pdf", "transform_sampled_points", "get_initial_rays_trig"]
    siren_calls = []
    orig_fw = g.siren.forward_with_frequencies_phase_shifts

    def siren_rec(*a, **k):
        r = orig_fw(*a, **k)
        siren_calls.append((a, k, r))
        return r
    g.siren.forward_with_frequencies_phase_shifts = siren_rec
    common = dict(img_size=S, num_steps=N, hierarchical_sample=hier, fov=CURR["fov"], ray_start=CURR["ray_start"],
                  ray_end=CURR["ray_end"], h_stddev=CURR["h_stddev"], v_stddev=CURR["v_stddev"],
                  h_mean=CURR["h_mean"], v_mean=CURR["v_mean"], sample_dist=CURR["sample_dist"])
    common.update(kwargs)
    with torch.no_grad(), DrawRecorder() as dr, CallRecorder(gens, rec_names) as cr:
        if staged:
            if spec["kind"] == "spatial":
                res = g.staged_forward_with_frequencies(tf["freq_geo"], tf["phase_geo"], max_batch_size=1000, **common)
            else:
                res = g.staged_forward_with_frequencies(tf["freq_geo"], tf["freq_app"], tf["phase_geo"], tf["phase_app"],
                                                        max_batch_size=1000, **common)
        else:
            if spec["kind"] == "spatial":
                res = g.forward_with_frequencies(tf["freq_geo"], tf["phase_geo"], **common)
            else:
                res = g.forward_with_frequencies(tf["freq_geo"], tf["freq_app"], tf["phase_geo"], tf["phase_app"], **common)
    out = dict(meta_seed=seed, meta_sigma_gain=sigma_gain, meta_B=B, meta_S=S, meta_N=N, meta_hier=int(hier),
               meta_film_scale=film_scale, meta_staged=int(staged), meta_weights_checksum=proc.checksum(sd))
    if film_kw:
        out.update(meta_film_phase_rev=float(film_kw["phase_rev"]), meta_film_freq0_gain=float(film_kw["freq0_gain"]))
    if state is not None:
        out["meta_state"] = state
    for k, v in spec.items():
        out["spec_" + k] = v
    for k, v in kwargs.items():
        out["kw_" + k] = v
    rd = rand_dict_from_draws(dr.draws, hier)
    for k, v in rd.items():
        out["rand_" + k] = v
    out["pixels"] = np_(res[0])
    if staged:
        out["depth"] = np_(res[1])
        if len(res) > 2:
            out["third"] = np_(res[2])
    else:
        out["poses"] = np_(res[1])
    if stages:
        tsp = cr.calls["transform_sampled_points"][0][2]
        for nm, t in zip(["points", "z_coarse", "dirs", "origins", "pitch", "yaw"], tsp):
            out["st_" + nm] = np_(t)
        rays = cr.calls["get_initial_rays_trig"][0][2]
        out["st_points_cam"], out["st_z_cam"], out["st_dirs_cam"] = (np_(t) for t in rays)
        fi = cr.calls["fancy_integration"]
        if hier:
            out["st_coarse_weights"] = np_(fi[0][2][2])
            sp = cr.calls["sample_pdf"][0]
            out["st_pdf_bins"], out["st_pdf_weights"], out["st_z_fine"] = np_(sp[0][0]), np_(sp[0][1]), np_(sp[2])
        last = fi[-1]
        out["st_all_out"], out["st_all_z"] = np_(last[0][0]), np_(last[0][1])
        out["st_final_rgb"], out["st_final_depth"], out["st_final_third"] = (np_(t) for t in last[2])
        if not staged:
            out["st_siren_coarse"] = np_(siren_calls[0][2])
            if hier:
                out["st_siren_fine"] = np_(siren_calls[1][2])
                out["st_fine_points"] = np_(siren_calls[1][0][0])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: pixels {out['pixels'].shape}  -> {os.path.getsize(path) / 1024:.0f} KiB")
    return g, sd, out


def run_grad_case(refs, name, spec, seed, sigma_gain, B, S, N, kwargs, film_scale=1.0, film_kw=None, state=None):
    """The reference's OWN autograd through generator.forward_with_frequencies (what g_loss.backward() replays,
    train_double_latent_semantic.py:408-452): loss = sum(pixels * w) with a fixed w; gradients wrt the raw FiLM parameters
    and every render parameter, plus the draws and the pixels.  fp32 on the CPU (gradient noise ~1e-5 relative)."""
    siren_mod, gens, vr, cur = refs
    g, sd = build_ref_generator(refs, spec, seed, sigma_gain, state)
    film = proc.film_params(spec, B, seed=seed, scale=film_scale, **(film_kw or {})) if state is None else load_state(state)[1]
    tf = {k: torch.from_numpy(v).requires_grad_(True) for k, v in film.items()}
    torch.manual_seed(4321 + seed)
    common = dict(img_size=S, num_steps=N, hierarchical_sample=True, fov=CURR["fov"], ray_start=CURR["ray_start"],
                  ray_end=CURR["ray_end"], h_stddev=CURR["h_stddev"], v_stddev=CURR["v_stddev"],
                  h_mean=CURR["h_mean"], v_mean=CURR["v_mean"], sample_dist=CURR["sample_dist"])
    common.update(kwargs)
    with DrawRecorder() as dr:
        if spec["kind"] == "spatial":     # single latent: one [B, 9H] frequency / phase tensor (the colour layer uses the last H)
            freq = torch.cat([tf["freq_geo"], tf["freq_app"]], -1)
            phase = torch.cat([tf["phase_geo"], tf["phase_app"]], -1)
            px, poses = g.forward_with_frequencies(freq, phase, **common)
        else:
            px, poses = g.forward_with_frequencies(tf["freq_geo"], tf["freq_app"], tf["phase_geo"], tf["phase_app"], **common)
    assert float(px.detach().std()) > 0.05, "degenerate (empty) render: pick another seed"
    w = torch.from_numpy(np.random.default_rng(seed).normal(size=tuple(px.shape)).astype(np.float32))
    (px * w).sum().backward()
    out = dict(meta_seed=seed, meta_sigma_gain=sigma_gain, meta_B=B, meta_S=S, meta_N=N, meta_hier=1, meta_film_scale=film_scale,
               meta_weights_checksum=proc.checksum(sd))
    if film_kw:
        out.update(meta_film_phase_rev=float(film_kw["phase_rev"]), meta_film_freq0_gain=float(film_kw["freq0_gain"]))
    if state is not None:
        out["meta_state"] = state
    for k, v in spec.items():
        out["spec_" + k] = v
    for k, v in kwargs.items():
        out["kw_" + k] = v
    for k, v in rand_dict_from_draws(dr.draws, True).items():
        out["rand_" + k] = v
    out["pixels"], out["loss_w"] = np_(px), np_(w)
    for k, t in tf.items():
        out["gfilm_" + k] = np_(t.grad)
    for n, p in g.siren.named_parameters():
        if "mapping_network" not in n:
            assert p.grad is not None, n
            out["gparam_" + n] = np_(p.grad)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: pixels {out['pixels'].shape}, {sum(k.startswith('gparam_') for k in out)} parameter gradients -> {os.path.getsize(path) / 1024:.0f} KiB")


def run_input_grad_case(refs, name, spec, seed, sigma_gain, B, P):
    """The reference's OWN autograd wrt the radiance field's INPUTS: `input.grad` / `ray_directions.grad` of the SIREN module's
    forward_with_frequencies_phase_shifts (siren.py:1509-1530 / :1210-1229 / :227-244) for loss = sum(out * w) -- through layer 0, the box
    warp, grid_sample's coordinate gradient and the colour layer's cat.  (No reference loop asks for them -- the generators build their
    rays under no_grad, generators.py:465 -- but a caller of the bare module gets them; fenerf_siren_input_grads.)  fp32 on the CPU."""
    g, sd = build_ref_generator(refs, spec, seed, sigma_gain)
    film = proc.film_params(spec, B, seed=seed)
    tf = {k: torch.from_numpy(v) for k, v in film.items()}
    rng = np.random.default_rng(100 + seed)
    pts = torch.from_numpy(rng.uniform(-0.125, 0.125, (B, P, 3)).astype(np.float32)).requires_grad_(True)     # some leave the grid box (zeros padding)
    dirs = rng.normal(size=(B, P, 3)).astype(np.float32)
    dirs = torch.from_numpy(dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)).requires_grad_(True)
    if spec["kind"] == "spatial":
        out_ = g.siren.forward_with_frequencies_phase_shifts(pts, torch.cat([tf["freq_geo"], tf["freq_app"]], -1),
                                                             torch.cat([tf["phase_geo"], tf["phase_app"]], -1), ray_directions=dirs)
    else:
        out_ = g.siren.forward_with_frequencies_phase_shifts(pts, tf["freq_geo"], tf["freq_app"], tf["phase_geo"], tf["phase_app"], ray_directions=dirs)
    w = rng.normal(size=tuple(out_.shape)).astype(np.float32)
    w[..., -1] *= 0.02          # sigma is ~sigma_gain x larger than the other outputs
    (out_ * torch.from_numpy(w)).sum().backward()
    out = dict(meta_seed=seed, meta_sigma_gain=sigma_gain, meta_B=B, meta_P=P, meta_film_scale=1.0, meta_weights_checksum=proc.checksum(sd),
               points=np_(pts), dirs=np_(dirs), loss_w=w, out=np_(out_), d_points=np_(pts.grad), d_dirs=np_(dirs.grad))
    for k, v in spec.items():
        out["spec_" + k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: out {out['out'].shape}, max |d points| {np.abs(out['d_points']).max():.3g}, max |d dirs| {np.abs(out['d_dirs']).max():.3g} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def run_grad_autocast_case(refs, name, base, spec, seed, sigma_gain, B, S, N, kwargs, film_scale=1.0):
    """The same generator step as run_grad_case(`base`), run by the reference under torch.autocast(float16) -- the arithmetic its training
    loop uses (torch.cuda.amp.autocast, train_double_latent_semantic.py:402-446; here the CPU autocast of this torch build) -- on the same
    draws.  Records the pixels and the render-parameter gradients: how far the reference's OWN mixed-precision training arithmetic is from its
    fp32 gradients is the yardstick for this package's opt-in AMP-class weight gradients (tests/test_gpu_parity.py).  fp16 CPU kernels are
    not guaranteed bit-stable, so tests/test_golden_recipe.py compares this one fixture with a tolerance."""
    g, sd = build_ref_generator(refs, spec, seed, sigma_gain)
    film = proc.film_params(spec, B, seed=seed, scale=film_scale)
    tf = {k: torch.from_numpy(v).requires_grad_(True) for k, v in film.items()}
    torch.manual_seed(4321 + seed)
    common = dict(img_size=S, num_steps=N, hierarchical_sample=True, fov=CURR["fov"], ray_start=CURR["ray_start"],
                  ray_end=CURR["ray_end"], h_stddev=CURR["h_stddev"], v_stddev=CURR["v_stddev"],
                  h_mean=CURR["h_mean"], v_mean=CURR["v_mean"], sample_dist=CURR["sample_dist"])
    common.update(kwargs)
    with torch.autocast("cpu", dtype=torch.float16):
        px, poses = g.forward_with_frequencies(tf["freq_geo"], tf["freq_app"], tf["phase_geo"], tf["phase_app"], **common)
    w = torch.from_numpy(np.load(os.path.join(OUT, base + ".npz"))["loss_w"])
    (px.float() * w).sum().backward()
    out = dict(meta_base=base, pixels=np_(px.float()))
    for n, p in g.siren.named_parameters():
        if "mapping_network" not in n:
            out["gparam_" + n] = np_(p.grad.float())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    ref = np.load(os.path.join(OUT, base + ".npz"))
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    errs = [rel(out[k], ref[k]) for k in out if k.startswith("gparam_") and k.endswith("layer.weight")]
    print(f"{name}: reference under autocast(float16): pixels off by {np.abs(out['pixels'] - ref['pixels']).max():.3f}, FiLM-layer weight "
          f"gradients off by {np.median(errs):.2f} (median) .. {max(errs):.2f} (worst) relative to its own fp32 gradients")


def run_part_forward_case(refs, name):
    """generator.forward(..., grad_points=G) = part_forward (generators.py:459-461, :858-910): records every draw including
    the randperm that picks the differentiable rays, the pixels and the gradient of a fixed loss wrt z-mapped FiLM inputs'
    source (the render weights) -- only the G picked rays carry gradient."""
    siren_mod, gens, vr, cur = refs
    spec = proc.model_spec("texture", hidden_dim=32, grid_size=5, z_dim=16)
    seed = 2          # chosen so the mean-pose render is not empty (the relu density is dead for some procedural seeds)
    g, sd = build_ref_generator(refs, spec, seed=seed, sigma_gain=60.0)
    B, S, N, G = 2, 6, 6, 11
    zg = torch.from_numpy(proc.normal("z_geo", (B, 16), 1.0, seed))
    za = torch.from_numpy(proc.normal("z_app", (B, 16), 1.0, seed))
    kw = dict(img_size=S, num_steps=N, hierarchical_sample=True, clamp_mode="relu", nerf_noise=0.1, grad_points=G, **CURR)
    torch.manual_seed(77)
    with DrawRecorder() as dr:
        px, poses = g.forward(zg, za, **kw)
    assert float(px.detach().std()) > 0.05, "degenerate (empty) render: pick another seed"
    w = torch.from_numpy(np.random.default_rng(9).normal(size=tuple(px.shape)).astype(np.float32))
    (px * w).sum().backward()
    kinds = [k for k, _ in dr.draws]
    # forward() calls part_forward with sample_dist=None (generators.py:461): the camera sits at the mean pose, no angle draws
    assert kinds == ["rand", "randperm", "randn", "rand", "randn", "randn", "rand", "randn"], kinds
    out = dict(z_geo=np_(zg), z_app=np_(za), meta_weights_checksum=proc.checksum(sd), meta_B=B, meta_S=S, meta_N=N, meta_G=G,
               meta_seed=seed, meta_sigma_gain=60.0, pixels=np_(px), poses=np_(poses), loss_w=np_(w))
    for k, v in spec.items():
        out["spec_" + k] = v
    for i, (kind, v) in enumerate(dr.draws):
        out[f"draw{i:02d}_{kind}"] = v
    for n, p in g.siren.named_parameters():
        if "mapping_network" not in n and p.grad is not None:
            out["gparam_" + n] = np_(p.grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: part_forward with {G} of {S * S} rays differentiable, {len(dr.draws)} draws")


def run_integration_variants(refs, base, name):
    """fancy_integration over every flag combination on one fixed (all_out, all_z) block + noise."""
    siren_mod, gens, vr, cur = refs
    rs, z = torch.from_numpy(base["st_all_out"]), torch.from_numpy(base["st_all_z"])
    # sharpen sigma so weights_sum straddles the 0.9 fill threshold on a good share of rays
    rs = rs.clone()
    fac = torch.tensor([40.0, 1.0, 0.05, -1.0, 4.0]).repeat(rs.shape[1] // 5 + 1)[:rs.shape[1]]
    rs[..., -1] = rs[..., -1] * fac[None, :, None]
    out = dict(rgb_sigma=np_(rs), z_vals=np_(z))
    variants = []
    for clamp in ("relu", "softplus"):
        for flags in (dict(), dict(last_back=True), dict(white_back=True), dict(black_back=True),
                      dict(last_back=True, white_back=True)):
            variants.append(dict(clamp_mode=clamp, noise_std=0.0, **flags))
    variants.append(dict(clamp_mode="relu", noise_std=0.5))
    variants.append(dict(clamp_mode="softplus", noise_std=1.0, last_back=True))
    # 'debug' / 'weight_debug' assign a 22-vector into the 21-channel rgb_final and raise RuntimeError for
    # output_dim=22 models (volumetric_rendering.py:54,66) -> not representable as a vector; see test of the raise.
    for fm in ("weight", "seg_padding_background", "eval_seg_padding_background"):
        variants.append(dict(clamp_mode="relu", noise_std=0.0, fill_mode=fm))
    for fc in ("white", "grey", "light_grey", "black", "none"):
        variants.append(dict(clamp_mode="relu", noise_std=0.0, fill_mode="seg_padding_background", fill_color=fc))
        variants.append(dict(clamp_mode="relu", noise_std=0.0, fill_mode="eval_seg_padding_background", fill_color=fc,
                             white_back=True))
    out["n_variants"] = len(variants)
    for i, v in enumerate(variants):
        torch.manual_seed(77 + i)
        with DrawRecorder() as dr:
            r = vr.fancy_integration(rs.clone(), z.clone(), device="cpu", **v)
        out[f"v{i}_noise"] = dr.draws[0][1]
        out[f"v{i}_kw"] = repr(v)
        out[f"v{i}_rgb"], out[f"v{i}_depth"], out[f"v{i}_third"] = (np_(t).copy() for t in r)
    # 3-channel model path for eval_white_back
    rs4 = torch.cat([rs[..., -4:-1], rs[..., -1:]], -1).contiguous()
    r = vr.fancy_integration(rs4.clone(), z.clone(), device="cpu", clamp_mode="relu", noise_std=0.0, fill_mode="eval_white_back")
    out["ewb_rgb_sigma"] = np_(rs4)
    out["ewb_rgb"], out["ewb_depth"], out["ewb_third"] = (np_(t).copy() for t in r)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(variants)} variants -> {os.path.getsize(path) / 1024:.0f} KiB")


def run_sample_pdf_cases(refs, name):
    siren_mod, gens, vr, cur = refs
    out = {}
    g = torch.Generator().manual_seed(5)
    cases = []
    for i, (R, K, Ni) in enumerate([(40, 4, 6), (33, 22, 24), (8, 46, 48), (5, 1, 7)]):
        bins = torch.sort(torch.rand(R, K + 1, generator=g) * 0.24 + 0.88, dim=-1)[0]
        w = torch.rand(R, K, generator=g) ** 4
        if i == 1:
            w[::3] = 0.0            # all-zero rows -> uniform pdf after eps
            w[1::3, 5:] = 0.0       # zero-weight bins -> denom<eps branch
        w = w + 1e-5
        torch.manual_seed(100 + i)
        with DrawRecorder() as dr:
            s = vr.sample_pdf(bins, w, Ni, det=False)
        u = dr.draws[0][1]
        sd = vr.sample_pdf(bins, w, Ni, det=True)
        out[f"c{i}_bins"], out[f"c{i}_weights"], out[f"c{i}_u"] = np_(bins), np_(w), u
        out[f"c{i}_samples"], out[f"c{i}_samples_det"] = np_(s), np_(sd)
        cases.append(i)
    # edge: u exactly on cdf knots (0 and interior) -> searchsorted-left semantics
    bins = torch.linspace(0.9, 1.1, 5).repeat(3, 1)
    w = torch.tensor([[1.0, 1.0, 1.0, 1.0], [0.0, 1.0, 0.0, 1.0], [1.0, 0.0, 0.0, 0.0]]) + 1e-5
    orig = torch.rand
    u_edge = torch.tensor([[0.0, 0.25, 0.5, 0.75, 0.999999], [0.0, 1e-6, 0.5, 0.5000001, 0.9], [0.0, 0.3, 0.9, 0.5, 0.1]])
    torch.rand = lambda *a, **k: u_edge.clone()
    try:
        s = vr.sample_pdf(bins, w, 5, det=False)
    finally:
        torch.rand = orig
    out["edge_bins"], out["edge_weights"], out["edge_u"], out["edge_samples"] = np_(bins), np_(w), np_(u_edge), np_(s)
    out["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(cases)} cases + edge")


def run_camera_cases(refs, name):
    siren_mod, gens, vr, cur = refs
    out = {}
    modes = ["uniform", "normal", "gaussian", "spherical_uniform", "fixed"]
    for i, m in enumerate(modes):
        torch.manual_seed(9 + i)
        with DrawRecorder() as dr:
            o, phi, theta = vr.sample_camera_positions(device="cpu", n=6, r=1, horizontal_stddev=0.3, vertical_stddev=0.155,
                                                       horizontal_mean=np.pi * 0.5, vertical_mean=np.pi * 0.5, mode=m)
        out[f"m{i}_mode"] = m
        out[f"m{i}_draws"] = np.stack([d for _, d in dr.draws]) if dr.draws else np.zeros((0, 6, 1), np.float32)
        out[f"m{i}_origin"], out[f"m{i}_phi"], out[f"m{i}_theta"] = np_(o), np_(phi), np_(theta)
        fwd = vr.normalize_vecs(-o)
        out[f"m{i}_cam2world"] = np_(vr.create_cam2world_matrix(fwd, o, device="cpu"))
    # 'hybrid' (host-RNG coin, then uniform x2 or gaussian draws, :195-201) and 'truncated_gaussian' (first of four in-place
    # normal_() candidates inside +-2, :170-177 / :203-205): the coin and the candidate blocks are recorded as draws too
    import random as pyrandom
    k = 0
    for m, seeds in (("hybrid", (1, 2, 3, 4)), ("truncated_gaussian", (21, 22))):
        for sd_ in seeds:
            torch.manual_seed(40 + sd_)
            pyrandom.seed(sd_)
            rec = []
            orig_random, orig_normal = pyrandom.random, torch.Tensor.normal_

            def rec_random():
                v = orig_random()
                rec.append(np.float64(v))
                return v

            def rec_normal(self, *a, **kw):
                r = orig_normal(self, *a, **kw)
                rec.append(r.detach().cpu().numpy().copy())
                return r

            pyrandom.random, torch.Tensor.normal_ = rec_random, rec_normal
            try:
                with DrawRecorder() as dr:
                    o, phi, theta = vr.sample_camera_positions(device="cpu", n=5, r=1, horizontal_stddev=0.3, vertical_stddev=0.155,
                                                               horizontal_mean=np.pi * 0.5, vertical_mean=np.pi * 0.5, mode=m)
            finally:
                pyrandom.random, torch.Tensor.normal_ = orig_random, orig_normal
            draws = rec[:1] + [d for _, d in dr.draws] if m == "hybrid" else rec      # call order: coin, then the two tensors
            out[f"x{k}_mode"], out[f"x{k}_n_draws"] = m, len(draws)
            for j, d in enumerate(draws):
                out[f"x{k}_draw{j}"] = np.asarray(d)
            out[f"x{k}_origin"], out[f"x{k}_phi"], out[f"x{k}_theta"] = np_(o), np_(phi), np_(theta)
            k += 1
    out["n_extra_modes"] = k
    # extreme pitch (clamp) case
    o, phi, theta = vr.sample_camera_positions(device="cpu", n=2, horizontal_stddev=0, vertical_stddev=0,
                                               horizontal_mean=0.3, vertical_mean=-0.2, mode="fixed")
    out["clamp_origin"], out["clamp_phi"], out["clamp_theta"] = np_(o), np_(phi), np_(theta)
    # rays at several resolutions
    for j, (S, N) in enumerate([(4, 3), (8, 6), (5, 1)]):
        p, z, d = vr.get_initial_rays_trig(2, N, "cpu", 12, (S, S), 0.88, 1.12)
        out[f"r{j}_S"], out[f"r{j}_N"] = S, N
        out[f"r{j}_points"], out[f"r{j}_z"], out[f"r{j}_dirs"] = np_(p), np_(z), np_(d)
    out["n_modes"] = len(modes)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: camera + rays")


def run_mapping_and_full(refs, name):
    """z -> mapping nets -> forward / staged_forward (psi truncation with recorded avg draws)."""
    siren_mod, gens, vr, cur = refs
    spec = proc.model_spec("texture", hidden_dim=32, grid_size=8, z_dim=16)
    g, sd = build_ref_generator(refs, spec, seed=3, sigma_gain=300.0)
    B, S, N = 2, 6, 6
    zg = torch.from_numpy(proc.normal("z_geo", (B, 16), 1.0, 3))
    za = torch.from_numpy(proc.normal("z_app", (B, 16), 1.0, 3))
    out = dict(z_geo=np_(zg), z_app=np_(za), meta_weights_checksum=proc.checksum(sd))
    for k, v in spec.items():
        out["spec_" + k] = v
    with torch.no_grad():
        fg, pg = g.siren.geo_mapping_network(zg)
        fa, pa = g.siren.app_mapping_network(za)
    out.update(map_freq_geo=np_(fg), map_phase_geo=np_(pg), map_freq_app=np_(fa), map_phase_app=np_(pa))
    kw = dict(img_size=S, num_steps=N, hierarchical_sample=True, clamp_mode="relu", nerf_noise=0.0, **CURR)
    torch.manual_seed(42)
    with torch.no_grad(), DrawRecorder() as dr:
        px, poses = g.forward(zg, za, **kw)
    rd = rand_dict_from_draws(dr.draws, True)
    for k, v in rd.items():
        out["fwd_rand_" + k] = v
    out["fwd_pixels"], out["fwd_poses"] = np_(px), np_(poses)
    # staged_forward: first two draws are the 10000-z avg-frequency pass (generators.py:530-543)
    torch.manual_seed(43)
    with torch.no_grad(), DrawRecorder() as dr:
        px, depth = g.staged_forward(zg, za, psi=0.7, max_batch_size=97, fill_mode="seg_padding_background",
                                     fill_color="white", **kw)
    assert dr.draws[0][1].shape == (10000, 16) and dr.draws[1][1].shape == (10000, 16)
    out["stg_avg_freq_geo"], out["stg_avg_phase_geo"] = np_(g.avg_frequencies_geo), np_(g.avg_phase_shifts_geo)
    out["stg_avg_freq_app"], out["stg_avg_phase_app"] = np_(g.avg_frequencies_app), np_(g.avg_phase_shifts_app)
    rd = rand_dict_from_draws(dr.draws[2:], True)
    for k, v in rd.items():
        out["stg_rand_" + k] = v
    out["stg_pixels"], out["stg_depth"] = np_(px), np_(depth)
    out["stg_psi"] = 0.7
    # the reference's checkpoint format: the whole pickled nn.Module (train_double_latent_semantic.py:526); pins that
    # fenerf_amd.compat.install_aliases() lets such a file unpickle into this package's classes
    for attr in ("avg_frequencies_geo", "avg_phase_shifts_geo", "avg_frequencies_app", "avg_phase_shifts_app"):
        if hasattr(g, attr):
            delattr(g, attr)
    torch.save(g, os.path.join(OUT, "ref_generator_tiny.pth"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: mapping + forward + staged_forward")


def run_style_generator_case(refs, name="tiny_style_generator"):
    """StyleGenerator3d (generators.py:914-1294; round 5): forward(z) and staged_forward(z, psi) -- psi and fill_color are ignored by that
    class -- on the tiny single-latent model, every draw recorded; and the pickled module (ref_style_generator_tiny.pth)."""
    siren_mod, gens, vr, cur = refs
    spec = proc.model_spec("spatial", hidden_dim=32, z_dim=16)
    cls = functools.partial(siren_mod.SPATIALSIRENBASELINE, hidden_dim=32)
    g = gens.StyleGenerator3d(cls, spec["z_dim"], spec["output_dim"])
    sd = proc.make_state_dict(spec, seed=8, sigma_gain=300.0)
    load_sd(g.siren, sd)
    g.eval()
    g.set_device(torch.device("cpu"))
    assert not hasattr(g, "avg_frequencies")
    B, S, N = 2, 6, 6
    z = torch.from_numpy(proc.normal("z_style", (B, 16), 1.0, 8))
    out = dict(z=np_(z), meta_seed=8, meta_sigma_gain=300.0, meta_weights_checksum=proc.checksum(sd))
    for k, v in spec.items():
        out["spec_" + k] = v
    kw = dict(img_size=S, num_steps=N, hierarchical_sample=True, clamp_mode="relu", nerf_noise=0.5, white_back=True, **CURR)
    torch.manual_seed(52)
    with torch.no_grad(), DrawRecorder() as dr:
        px, poses = g.forward(z, **kw)
    for k, v in rand_dict_from_draws(dr.draws, True).items():
        out["fwd_rand_" + k] = v
    out["fwd_pixels"], out["fwd_poses"] = np_(px), np_(poses)
    torch.manual_seed(53)
    with torch.no_grad(), DrawRecorder() as dr:
        # fill_mode 'weight': the only family of modes this method survives (it reshapes fancy_integration's third output to channel_dim, :1086)
        px, depth, wsum = g.staged_forward(z, psi=0.3, max_batch_size=97, fill_mode="weight", fill_color="white", **kw)
    assert len(dr.draws) == 6, "no 10,000-latent average pass in this class"
    for k, v in rand_dict_from_draws(dr.draws, True).items():
        out["stg_rand_" + k] = v
    out["stg_pixels"], out["stg_depth"], out["stg_third"] = np_(px), np_(depth), np_(wsum)
    out["stg_psi"] = 0.3
    torch.save(g, os.path.join(OUT, "ref_style_generator_tiny.pth"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: StyleGenerator3d forward + staged_forward")


def run_caller_helpers(name="caller_helpers"):
    """mask2color (train_double_latent_semantic.py:36-72) and create_samples (extract_double_semantic_shapes.py:13-35):
    the two function bodies are executed straight from the reference sources (AST-extracted, so the scripts' heavy
    unrelated imports -- datasets, torch_ema, tensorboard, mrcfile ... -- are never triggered)."""
    import ast
    ns = {"torch": torch, "np": np}
    for fn, wanted in (("train_double_latent_semantic.py", {"COLOR_MAP", "mask2color"}), ("extract_double_semantic_shapes.py", {"create_samples"})):
        tree = ast.parse(open(os.path.join(ref_import.REFERENCE_ROOT, fn)).read())
        keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in wanted) or
                (isinstance(n, ast.Assign) and any(getattr(t, "id", None) in wanted for t in n.targets))]
        exec(compile(ast.Module(body=keep, type_ignores=[]), fn, "exec"), ns)
    out = {}
    g = torch.Generator().manual_seed(11)
    masks = torch.randn(3, 19, 6, 5, generator=g)
    masks[0, :, 0, 0] = 0.0          # exact tie -> argmax picks the first index
    masks[1, 7, 2, 2] = masks[1, 3, 2, 2] = masks[1].max() + 1.0
    out["masks"] = np_(masks)
    out["colors"] = np_(ns["mask2color"](masks))
    out["color_map"] = np.array([ns["COLOR_MAP"][k] for k in range(19)], dtype=np.float32)
    for N in (4, 5):
        s, vo, vs = ns["create_samples"](N, [0, 0, 0], 0.3)
        out[f"samples_{N}"], out[f"voxel_origin_{N}"], out[f"voxel_size_{N}"] = np_(s), np.asarray(vo, np.float64), np.float64(vs)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: mask2color + create_samples")


def run_inversion_helpers(name="inversion_helpers"):
    """The host pieces of inverse_render_double_semantic.py that fenerf_amd.callers restates -- mask2labels (:82-92), mIOU (:122-126),
    set_trajectory (:504-570) -- executed straight from the reference source (AST-extracted: the script's imports -- lpips, skvideo,
    torchvision, tensorboard -- and its module-level argparse / torch.load never run)."""
    import ast
    import math
    import types
    fn = "inverse_render_double_semantic.py"
    wanted = {"COLOR_MAP", "COLOR_MAP_COMPLETE", "mask2labels", "mIOU", "set_trajectory"}
    tree = ast.parse(open(os.path.join(ref_import.REFERENCE_ROOT, fn)).read())
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in wanted) or
            (isinstance(n, ast.Assign) and any(getattr(t, "id", None) in wanted for t in n.targets))]
    ns = {"torch": torch, "np": np, "math": math, "render_options": {"fov": 12}}
    exec(compile(ast.Module(body=keep, type_ignores=[]), fn, "exec"), ns)
    out = {}
    rng = np.random.default_rng(21)
    mask = rng.integers(0, 19, (6, 5)).astype(np.float32)
    out["mask"] = mask
    out["labels_18"] = ns["mask2labels"](mask)
    out["labels_19"] = ns["mask2labels"](mask, ns["COLOR_MAP_COMPLETE"])
    src = torch.tensor(rng.integers(0, 2, (2, 19, 6, 5)).astype(np.float32))
    tgt = torch.tensor(rng.integers(0, 2, (2, 19, 6, 5)).astype(np.float32))
    out["miou_source"], out["miou_target"], out["miou"] = np_(src), np_(tgt), np_(ns["mIOU"](src, tgt))
    names = ["front", "orbit", "non_rotation", "sphere", "inverse_sphere", "rotation_horizontal", "zoom", "rotation_linear"]
    out["trajectory_names"] = np.array(names)
    out["trajectory_num_frames"] = 7
    for n in names:
        out["trajectory_" + n] = np.array(ns["set_trajectory"](types.SimpleNamespace(trajectory=n, num_frames=7)), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: mask2labels + mIOU + set_trajectory of the inversion script")


def run_multiview_case(refs, name="tiny_multiview"):
    """generate_img of render_multiview_images_double_semantic.py:24-29 (AST-extracted, executed as is) over the script's
    five yaw angles (:68-83) on the tiny reference generator, with the options bag the script builds (:43-54) from a small
    curriculum: z from torch.manual_seed(seed) on the CPU generator, every draw recorded (identical for all five angles: the
    script re-seeds per angle)."""
    import ast
    siren_mod, gens, vr, cur = refs
    spec = proc.model_spec("texture", hidden_dim=32, grid_size=8, z_dim=16)
    g, sd = build_ref_generator(refs, spec, seed=3, sigma_gain=300.0)
    g.softmax_label = False
    ns = {"torch": torch, "np": np, "generator": g}
    for fn, wanted in (("train_double_latent_semantic.py", {"COLOR_MAP", "mask2color"}), ("render_multiview_images_double_semantic.py", {"generate_img"})):
        tree = ast.parse(open(os.path.join(ref_import.REFERENCE_ROOT, fn)).read())
        keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in wanted) or
                (isinstance(n, ast.Assign) and any(getattr(t, "id", None) in wanted for t in n.targets))]
        exec(compile(ast.Module(body=keep, type_ignores=[]), fn, "exec"), ns)
    curriculum = {0: dict(batch_size=4, num_steps=3, img_size=8), "fov": 12, "ray_start": 0.88, "ray_end": 1.12, "h_stddev": 0.3,
                  "v_stddev": 0.155, "h_mean": float(np.pi * 0.5), "v_mean": float(np.pi * 0.5), "sample_dist": "gaussian",
                  "hierarchical_sample": True, "clamp_mode": "relu", "fill_mode": "seg_padding_background", "white_back": False}
    image_size, mult, seed = 8, 2, 5
    c = dict(curriculum)                      # the script's lines :43-54
    c["num_steps"] = c[0]["num_steps"] * mult
    c["img_size"] = image_size
    c["psi"] = 0.7
    c["v_stddev"] = 0
    c["h_stddev"] = 0
    c["lock_view_dependence"] = True
    c["last_back"] = False
    c["nerf_noise"] = 0
    c = {k: v for k, v in c.items() if type(k) is str}
    face_angles = [a + c["h_mean"] for a in [-0.5, -0.25, 0., 0.25, 0.5]]
    images, segmaps, first = [], [], None
    for yaw in face_angles:
        c["h_mean"] = yaw
        torch.manual_seed(seed)
        z_geo = torch.randn((1, 16))
        z_app = torch.randn((1, 16))
        with DrawRecorder() as dr:
            img, segmap = ns["generate_img"](g, z_geo, z_app, **c)
        rec = dict(avg=[np_(g.avg_frequencies_geo), np_(g.avg_phase_shifts_geo), np_(g.avg_frequencies_app), np_(g.avg_phase_shifts_app)],
                   draws=list(dr.draws[2:]))
        if first is None:
            first = rec
        else:
            assert all(np.array_equal(a, b) for a, b in zip(first["avg"] + [d for _, d in first["draws"]], rec["avg"] + [d for _, d in rec["draws"]]))
        images.append(np_(img))
        segmaps.append(np_(segmap))
    out = dict(z_geo=np_(z_geo), z_app=np_(z_app), images=np.concatenate(images), segmaps=np.concatenate(segmaps), seed=seed,
               image_size=image_size, ray_step_multiplier=mult, curriculum_json=json_dumps_curriculum(curriculum),
               avg_freq_geo=first["avg"][0], avg_phase_geo=first["avg"][1], avg_freq_app=first["avg"][2], avg_phase_app=first["avg"][3])
    for k, v in rand_dict_from_draws(first["draws"], True).items():
        out["rand_" + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: images {out['images'].shape}, segmaps {out['segmaps'].shape}")


def run_trained_state(refs, name, spec, seed, teacher_seed, steps, B, S, N, lr=1e-4, lr_film=2e-3):
    """Weights and FiLM parameters an OPTIMISER produced (round 5): every other fixture sits on the init manifold (procedural weights in
    the reference initialisers' ranges, FiLM parameters in the mapping networks' init range, or hand-scaled excursions of those).
    Here the reference generator itself is trained in this container: `steps` Adam steps (betas (0, 0.9) like the curriculum's generator
    optimiser, train_double_latent_semantic.py:166; lr 1e-4 on every render weight, 8^3 feature grid included -- the curriculum's 2e-5 ..
    6e-5 over 100k+ steps is not affordable here, and a larger rate drives this tiny network into a regime where ANY two fp32 evaluations
    differ by O(1): with lr 2e-3 a 1e-6 relative change of one input moved the reference's own loss by 7 % -- and 2e-3 on the raw FiLM
    parameters), MSE between generator.forward_with_frequencies (hierarchical, fresh jitter / resampling draws per step, fixed pose) and
    the pixels of a second, denser procedural generator ("teacher") -- the reference's own forward, the reference's own autograd, torch's
    Adam.  The fixture holds the resulting state (sd_* / film_*), the loss curve and the target; run_film_case / run_grad_case(state=...)
    then record the reference's stage-wise outputs and gradients AT that state."""
    siren_mod, gens, vr, cur = refs
    common = dict(img_size=S, num_steps=N, hierarchical_sample=True, fov=CURR["fov"], ray_start=CURR["ray_start"], ray_end=CURR["ray_end"],
                  h_stddev=0.0, v_stddev=0.0, h_mean=CURR["h_mean"], v_mean=CURR["v_mean"], sample_dist=None, clamp_mode="relu", nerf_noise=0.0)
    teacher, _ = build_ref_generator(refs, spec, teacher_seed, 300.0)
    tfilm = {k: torch.from_numpy(v) for k, v in proc.film_params(spec, B, seed=teacher_seed).items()}
    torch.manual_seed(9000 + seed)
    with torch.no_grad():
        target, _ = teacher.forward_with_frequencies(tfilm["freq_geo"], tfilm["freq_app"], tfilm["phase_geo"], tfilm["phase_app"], **common)
    g, sd0 = build_ref_generator(refs, spec, seed, 60.0)
    film = {k: torch.from_numpy(v).requires_grad_(True) for k, v in proc.film_params(spec, B, seed=seed).items()}
    params = [p for n, p in g.siren.named_parameters() if "mapping_network" not in n]
    opt = torch.optim.Adam([dict(params=params, lr=lr), dict(params=list(film.values()), lr=lr_film)], betas=(0.0, 0.9))
    losses = []
    for i in range(steps):
        px, _ = g.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **common)
        loss = torch.nn.functional.mse_loss(px, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.mean(losses[-10:]) < 0.5 * np.mean(losses[:10]), (losses[0], losses[-1])
    out = dict(meta_seed=seed, meta_teacher_seed=teacher_seed, meta_steps=steps, meta_lr=lr, meta_lr_film=lr_film, meta_B=B, meta_S=S, meta_N=N,
               losses=np.asarray(losses, np.float32), target=np_(target))
    for k, v in spec.items():
        out["spec_" + k] = v
    moved = []
    for n, p in g.siren.named_parameters():
        if "mapping_network" not in n:
            out["sd_" + n] = np_(p)
            moved.append(float(np.abs(out["sd_" + n] - sd0[n]).max() / np.abs(sd0[n]).max()))
    f0 = proc.film_params(spec, B, seed=seed)
    for k, t in film.items():
        out["film_" + k] = np_(t)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {steps} Adam steps of the reference generator: loss {losses[0]:.4f} -> {losses[-1]:.4f}; weights moved by "
          f"{min(moved):.2f} .. {max(moved):.2f} of their init range, FiLM parameters by up to "
          f"{max(float(np.abs(out['film_' + k] - f0[k]).max()) for k in f0):.2f}")


def run_inversion_case(refs, name, spec, state, seed, n_iterations, n_mean_latents, S, N):
    """The reference's inversion loop (inverse_render_double_semantic.py:306-410) on the frozen trained generator of `state`, recorded:
    every draw in call order, the loss of every iteration and the four offset tensors after every iteration.  The loop is a script body
    there (not importable), so it is restated here statement by statement on the REFERENCE's generator, torch's Adam (lr 1e-2,
    weight_decay 1e-4) and StepLR(100, 0.75): mean FiLM parameters over `n_mean_latents` latents (10,000 there), init_psi 0, annealed
    latent noise 0.03 (n - i) / n, forward_with_frequencies with the script's `options` dict (its size entries scaled down),
    lambda_seg = lambda_img = 1 on the two MSE terms, no LPIPS term (lambda_percept 0: the package is absent), no norm term.  The target is
    a render of the same generator at truncated (psi 0.6) FiLM parameters of one random latent pair."""
    siren_mod, gens, vr, cur = refs
    g, sd = build_ref_generator(refs, spec, seed, 60.0, state)
    for p in g.parameters():
        p.requires_grad_(False)
    zd = spec["z_dim"]
    # the script's options (:220-243), sizes scaled to the tiny model; device tensors for h_mean / v_mean like there
    options = dict(img_size=S, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0, v_stddev=0,
                   h_mean=torch.tensor(np.pi / 2), v_mean=torch.tensor(np.pi / 2), hierarchical_sample=False, sample_dist=None,
                   clamp_mode="relu", nerf_noise=0, fade_steps=10000, z_app_lambda=0, z_geo_lambda=0, pos_lambda=0, tok_interval=2000,
                   tok_v=0.6, betas=(0, 0.9), fill_mode="eval_seg_padding_background")
    torch.manual_seed(2500 + seed)
    with torch.no_grad():      # target: psi-truncated sample of the same generator
        zs_g, zs_a = torch.randn((2000, zd)), torch.randn((2000, zd))
        mg, ma = g.siren.geo_mapping_network(zs_g), g.siren.app_mapping_network(zs_a)
        rg, ra = g.siren.geo_mapping_network(torch.randn((1, zd))), g.siren.app_mapping_network(torch.randn((1, zd)))
        mix = lambda m, r: m.mean(0, keepdim=True) + 0.6 * (r - m.mean(0, keepdim=True))
        target, _ = g.forward_with_frequencies(mix(mg[0], rg[0]), mix(ma[0], ra[0]), mix(mg[1], rg[1]), mix(ma[1], ra[1]), **options)
    gt_seg_18, gt_image = target[:, :-3].clone(), target[:, -3:].clone()
    losses, offs = [], []
    torch.manual_seed(2600 + seed)
    with DrawRecorder() as dr:
        z_geo = torch.randn((n_mean_latents, zd))
        rand_z_geo = torch.randn((1, zd))
        with torch.no_grad():
            geo_frequencies, geo_phase_shifts = g.siren.geo_mapping_network(z_geo)
            rand_geo_frequencies, rand_geo_phase_shifts = g.siren.geo_mapping_network(rand_z_geo)
        init_psi = 0.0
        w_geo_frequencies = geo_frequencies.mean(0, keepdim=True)
        w_geo_phase_shifts = geo_phase_shifts.mean(0, keepdim=True)
        w_geo_frequencies = w_geo_frequencies + init_psi * (rand_geo_frequencies - w_geo_frequencies)
        w_geo_phase_shifts = w_geo_phase_shifts + init_psi * (rand_geo_phase_shifts - w_geo_phase_shifts)
        w_geo_frequency_offsets = torch.zeros_like(w_geo_frequencies).requires_grad_()
        w_geo_phase_shift_offsets = torch.zeros_like(w_geo_phase_shifts).requires_grad_()
        z_app = torch.randn((n_mean_latents, zd))
        rand_z_app = torch.randn((1, zd))
        with torch.no_grad():
            app_frequencies, app_phase_shifts = g.siren.app_mapping_network(z_app)
            rand_app_frequencies, rand_app_phase_shifts = g.siren.app_mapping_network(rand_z_app)
        w_app_frequencies = app_frequencies.mean(0, keepdim=True)
        w_app_phase_shifts = app_phase_shifts.mean(0, keepdim=True)
        w_app_frequencies = w_app_frequencies + init_psi * (rand_app_frequencies - w_app_frequencies)
        w_app_phase_shifts = w_app_phase_shifts + init_psi * (rand_app_phase_shifts - w_app_phase_shifts)
        w_app_frequency_offsets = torch.zeros_like(w_app_frequencies).requires_grad_()
        w_app_phase_shift_offsets = torch.zeros_like(w_app_phase_shifts).requires_grad_()
        optimizer = torch.optim.Adam([w_geo_frequency_offsets, w_geo_phase_shift_offsets, w_app_frequency_offsets, w_app_phase_shift_offsets],
                                     lr=1e-2, weight_decay=1e-4)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, 100, gamma=0.75)
        for i in range(n_iterations):
            k = (n_iterations - i) / n_iterations
            noise_w_geo_frequencies = 0.03 * torch.randn_like(w_geo_frequencies) * k
            noise_w_geo_phase_shifts = 0.03 * torch.randn_like(w_geo_phase_shifts) * k
            noise_w_app_frequencies = 0.03 * torch.randn_like(w_app_frequencies) * k
            noise_w_app_phase_shifts = 0.03 * torch.randn_like(w_app_phase_shifts) * k
            frame, position = g.forward_with_frequencies(w_geo_frequencies + noise_w_geo_frequencies + w_geo_frequency_offsets,
                                                         w_app_frequencies + noise_w_app_frequencies + w_app_frequency_offsets,
                                                         w_geo_phase_shifts + noise_w_geo_phase_shifts + w_geo_phase_shift_offsets,
                                                         w_app_phase_shifts + noise_w_app_phase_shifts + w_app_phase_shift_offsets, **options)
            seg_loss = torch.nn.MSELoss(reduction="mean")(frame[:, :-3], gt_seg_18)
            img_loss = torch.nn.MSELoss(reduction="mean")(frame[:, -3:], gt_image)
            loss = 1.0 * seg_loss + 1.0 * img_loss
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            scheduler.step()
            losses.append(float(loss.detach()))
            offs.append([np_(t) for t in (w_geo_frequency_offsets, w_geo_phase_shift_offsets, w_app_frequency_offsets, w_app_phase_shift_offsets)])
    out = dict(meta_state=state, meta_seed=seed, meta_sigma_gain=60.0, meta_iterations=n_iterations, meta_mean_latents=n_mean_latents,
               meta_S=S, meta_N=N, meta_weights_checksum=proc.checksum(sd), gt_image=np_(gt_image), gt_seg=np_(gt_seg_18),
               losses=np.asarray(losses, np.float64),
               w_geo_frequencies=np_(w_geo_frequencies), w_geo_phase_shifts=np_(w_geo_phase_shifts),
               w_app_frequencies=np_(w_app_frequencies), w_app_phase_shifts=np_(w_app_phase_shifts))
    for j, nm in enumerate(("geo_frequency", "geo_phase_shift", "app_frequency", "app_phase_shift")):
        out[f"offsets_{nm}"] = np.stack([o[j] for o in offs])          # [iteration, 1, n]
    for k, v in spec.items():
        out["spec_" + k] = v
    for i, (kind, v) in enumerate(dr.draws):
        out[f"draw{i:04d}_{kind}"] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: the reference's inversion loop, {n_iterations} iterations, {len(dr.draws)} draws: loss {losses[0]:.5f} -> {losses[-1]:.5f}, "
          f"|offsets| up to {max(float(np.abs(o).max()) for o in offs[-1]):.3f}")


def json_dumps_curriculum(curriculum):
    import json
    return json.dumps({(f"int:{k}" if isinstance(k, int) else k): v for k, v in curriculum.items()}, sort_keys=True)


def run_spatial_grid_case(refs, name="tiny_spatial_grid"):
    """SPATIALSIRENGRID (siren.py:413-518): per-point FiLM modulation from a 2-D grid of local latents.  The reference module
    runs here as is (its StyleGenerator2D falls back to native torch ops); recorded: its render / mapping weights, the latent
    grid its generator produced for two z (an INPUT of the drop-in), points, directions and every stage of forward()."""
    siren_mod = refs[0]
    torch.manual_seed(31)
    H = 32
    ref = siren_mod.SPATIALSIRENGRID(input_dim=3, z_dim=16, hidden_dim=H, output_dim=4)
    # the latent-grid generator (4.1 M parameters: too large for a fixture) gets procedural weights the tests can re-create
    gshapes = {n_: tuple(p_.shape) for n_, p_ in ref.grid_latent_network.named_parameters()}
    with torch.no_grad():
        for n_, v in proc.latent_grid_state(gshapes, seed=31).items():
            dict(ref.grid_latent_network.named_parameters())[n_].copy_(torch.from_numpy(v))
    B, P = 2, 70
    z = torch.randn(B, 16)
    g = torch.Generator().manual_seed(32)
    pts = (torch.rand(B, P, 3, generator=g) - 0.5) * 0.24
    dirs = torch.nn.functional.normalize(torch.randn(B, P, 3, generator=g), dim=-1)
    out = {}
    with torch.no_grad():
        latent_grid = ref.grid_latent_network(z)
        input_grid = ref.gridwarper(pts)
        sampled = ref.sample_local_latents(latent_grid, input_grid)
        freq, phase = ref.mapping_network(sampled)
        local = ref.get_local_coordinates(global_coords=pts, local_grid_length=32, preserve_y=False)
        res = ref(pts, z, dirs)
        res2 = ref.forward_with_frequencies_phase_shifts(local, freq, phase, dirs, box_warp=False)
    assert torch.equal(res, res2)
    for n_, p_ in ref.named_parameters():
        if not n_.startswith("grid_latent_network"):
            out["w_" + n_] = np_(p_)
    import json
    full = ref.state_dict()
    out.update(z=np_(z), latent_grid=np_(latent_grid), points=np_(pts), dirs=np_(dirs), sampled_latent=np_(sampled), freq=np_(freq),
               phase=np_(phase), local_coords=np_(local), out=np_(res), meta_H=H,
               # the reference module's complete state dict as names + shapes (strict state-dict compatibility), the two blur kernels
               meta_state_dict=json.dumps({k: list(v.shape) for k, v in full.items()}, sort_keys=True),
               blur_kernel=np_(full["grid_latent_network.convs.0.blur.kernel"]))
    # round 4: the reference module under ITS OWN autograd (it is an ordinary differentiable nn.Module, siren.py:413-477) -- gradients of
    # sum(out * w) wrt the SIREN weights, the per-point mapping network, z (through the latent-grid generator) and, teacher-forced, the
    # latent grid itself.  A separate pass after everything above: nothing recorded so far changes.
    w = torch.from_numpy(np.random.default_rng(33).normal(size=tuple(res.shape)).astype(np.float32))
    zz = z.clone().requires_grad_(True)
    for p_ in ref.parameters():
        p_.grad = None
    res_g = ref(pts, zz, dirs)
    assert torch.equal(res_g.detach(), res)
    (res_g * w).sum().backward()
    out["loss_w"], out["g_z"] = np_(w), np_(zz.grad)
    for n_, p_ in ref.named_parameters():
        if not n_.startswith("grid_latent_network"):
            out["gw_" + n_] = np_(p_.grad)
    gl = {n_: p_.grad for n_, p_ in ref.grid_latent_network.named_parameters() if p_.grad is not None}
    out["g_generator_names"] = json.dumps(sorted(gl))
    out["g_generator_norms"] = np.array([float(gl[k].double().norm()) for k in sorted(gl)])     # 4.1 M values: their norms pin them
    lg = latent_grid.clone().requires_grad_(True)
    sampled_g = ref.sample_local_latents(lg, ref.gridwarper(pts))
    f_g, p_g = ref.mapping_network(sampled_g)
    res_l = ref.forward_with_frequencies_phase_shifts(local, f_g, p_g, dirs, box_warp=False)
    (res_l * w).sum().backward()
    out["g_latent_grid"] = np_(lg.grad)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: out {tuple(res.shape)}, per-point freq {tuple(freq.shape)}, {sum(k.startswith('gw_') for k in out)} weight gradients + z + latent grid")


def run_curriculums(refs, name="curriculums"):
    """tests/golden/curriculums.json: the three curriculum dicts the path is quoted on, as the reference's curriculums.py
    defines them (integer stage keys spelled "int:<step>", tuples as lists; `extract_metadata` etc. are behaviour, not data,
    and are pinned by tests/test_host_cpu.py through these values)."""
    import json
    cur = refs[3]
    out = {}
    for cname in ("CelebA", "CelebA_double_semantic", "CelebA_double_semantic_texture_embedding_256_dim_96"):
        d = getattr(cur, cname)
        out[cname] = {(f"int:{k}" if isinstance(k, int) else k): (list(v) if isinstance(v, tuple) else v) for k, v in d.items()}
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"{name}: {len(out)} curriculum dicts")


def main(out_dir=None):
    """Regenerates every fixture into `out_dir` (default tests/golden).  tests/test_golden_recipe.py calls this into a scratch
    directory when /root/reference is present and compares the result with the committed files bit for bit."""
    global OUT
    if out_dir is not None:
        OUT = out_dir
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    refs = ref_import.import_reference()
    run_curriculums(refs)
    relu = dict(clamp_mode="relu", nerf_noise=0.0)

    tiny = proc.model_spec("texture", hidden_dim=32, grid_size=8, z_dim=16)
    _, _, base = run_film_case(refs, "tiny_texture_fwd", tiny, seed=1, sigma_gain=300.0, B=2, S=8, N=6, hier=True, kwargs=relu)
    run_film_case(refs, "tiny_texture_fwd_nohier", tiny, seed=2, sigma_gain=300.0, B=2, S=8, N=6, hier=False,
                  kwargs=dict(clamp_mode="softplus", nerf_noise=0.5, last_back=True))
    run_film_case(refs, "tiny_texture_staged", tiny, seed=1, sigma_gain=300.0, B=2, S=8, N=6, hier=True, staged=True,
                  kwargs=dict(clamp_mode="relu", nerf_noise=0.0, fill_mode="seg_padding_background", fill_color="black"))
    run_film_case(refs, "tiny_texture_staged_lock", tiny, seed=4, sigma_gain=300.0, B=1, S=8, N=6, hier=True, staged=True,
                  kwargs=dict(clamp_mode="relu", nerf_noise=0.3, fill_mode="eval_seg_padding_background",
                              fill_color="grey", lock_view_dependence=True))
    run_integration_variants(refs, base, "integration_variants")
    run_sample_pdf_cases(refs, "sample_pdf_cases")
    run_camera_cases(refs, "camera_rays")
    run_mapping_and_full(refs, "tiny_texture_z_full")
    run_caller_helpers()
    run_inversion_helpers()
    run_multiview_case(refs)
    run_spatial_grid_case(refs)
    run_style_generator_case(refs)
    run_part_forward_case(refs, "tiny_texture_part_forward")
    run_grad_case(refs, "tiny_texture_grad", proc.model_spec("texture", hidden_dim=32, grid_size=5, z_dim=16), seed=3, sigma_gain=60.0,
                  B=2, S=6, N=8, kwargs=dict(clamp_mode="relu", nerf_noise=0.2, white_back=True))
    run_grad_autocast_case(refs, "tiny_texture_grad_autocast16", "tiny_texture_grad", proc.model_spec("texture", hidden_dim=32, grid_size=5, z_dim=16),
                           seed=3, sigma_gain=60.0, B=2, S=6, N=8, kwargs=dict(clamp_mode="relu", nerf_noise=0.2, white_back=True))
    run_grad_case(refs, "tiny_baseline_grad", proc.model_spec("baseline", hidden_dim=32, z_dim=16), seed=5, sigma_gain=60.0,
                  B=1, S=6, N=6, kwargs=dict(clamp_mode="softplus", nerf_noise=0.3, last_back=True))
    run_grad_case(refs, "tiny_spatial_grad", proc.model_spec("spatial", hidden_dim=32, z_dim=16), seed=8, sigma_gain=60.0,
                  B=2, S=5, N=7, kwargs=dict(clamp_mode="relu", nerf_noise=0.0, lock_view_dependence=True))

    # gradients wrt the SIREN's inputs (round 6): the reference module's own autograd, one fixture per model family
    run_input_grad_case(refs, "tiny_texture_input_grad", proc.model_spec("texture", hidden_dim=32, grid_size=5, z_dim=16), seed=3, sigma_gain=30.0, B=2, P=75)
    run_input_grad_case(refs, "tiny_baseline_input_grad", proc.model_spec("baseline", hidden_dim=32, z_dim=16), seed=5, sigma_gain=30.0, B=1, P=64)
    run_input_grad_case(refs, "tiny_spatial_input_grad", proc.model_spec("spatial", hidden_dim=32, z_dim=16), seed=8, sigma_gain=30.0, B=2, P=33)

    baseline = proc.model_spec("baseline", hidden_dim=32, z_dim=16)
    run_film_case(refs, "tiny_baseline_fwd", baseline, seed=5, sigma_gain=300.0, B=2, S=8, N=6, hier=True, kwargs=relu)

    # H=256 / 96^3 grid: output-level vectors only (weights are procedural; checksum pinned)
    full = proc.model_spec("texture", hidden_dim=256, grid_size=96)
    run_film_case(refs, "h256_texture_16x16_n12", full, seed=0, sigma_gain=1.0, B=1, S=16, N=12, hier=True, kwargs=relu)
    run_film_case(refs, "h256_texture_16x16_n24_trained", full, seed=0, sigma_gain=2000.0, B=1, S=16, N=24, hier=True, kwargs=relu)
    spatial = proc.model_spec("spatial", hidden_dim=32, z_dim=16)
    run_film_case(refs, "tiny_spatial_fwd", spatial, seed=8, sigma_gain=300.0, B=2, S=8, N=6, hier=True,
                  kwargs=dict(clamp_mode="relu", nerf_noise=0.0, last_back=True))
    run_film_case(refs, "tiny_spatial_staged", spatial, seed=8, sigma_gain=300.0, B=2, S=8, N=6, hier=True, staged=True,
                  kwargs=dict(clamp_mode="relu", nerf_noise=0.0, fill_mode="eval_white_back"))
    fullb = proc.model_spec("baseline", hidden_dim=256)
    run_film_case(refs, "h256_baseline_8x8_n12", fullb, seed=6, sigma_gain=2000.0, B=2, S=8, N=12, hier=True, kwargs=relu)

    # FiLM parameters far beyond the init range (round 4): phase shifts of up to +-300 revolutions in every layer, the first layer's
    # frequency x 4 -- what torch.sin in the reference's FiLMLayer takes in its stride (siren.py:113-123) and a hardware sine defined on
    # +-256 revolutions does not: pins the oracle (and through it the kernels' range reduction, fenerf_trig.h) to the reference there,
    # forward and backward.  (x 30 on the first layer was tried: the field then varies 30 x faster in space and the END-TO-END render is
    # only reproducible to ~1e-2 between any two fp32 evaluations -- the reference's own pixels sit 7e-3 from an fp64 evaluation --; the
    # per-kernel tests against fp64 in tests/test_gpu_parity.py cover first-layer arguments of up to 1,524 revolutions.)
    big = dict(phase_rev=300.0, freq0_gain=4.0)
    run_film_case(refs, "tiny_texture_fwd_bigfilm", tiny, seed=1, sigma_gain=300.0, B=2, S=8, N=6, hier=True, kwargs=relu, film_kw=big)
    run_film_case(refs, "h256_texture_8x8_n12_bigfilm", full, seed=0, sigma_gain=30.0, B=1, S=8, N=12, hier=True, kwargs=relu, film_kw=big)
    run_grad_case(refs, "tiny_texture_grad_bigfilm", proc.model_spec("texture", hidden_dim=32, grid_size=5, z_dim=16), seed=3, sigma_gain=60.0,
                  B=2, S=6, N=8, kwargs=dict(clamp_mode="relu", nerf_noise=0.2, white_back=True), film_kw=big)

    # Off the init manifold (round 5): a state the reference's own forward + autograd + Adam produced in this container, then the usual
    # stage-wise outputs and gradients AT that state, and the reference's inversion loop on the frozen trained generator.
    tiny8 = proc.model_spec("texture", hidden_dim=32, grid_size=8, z_dim=8)
    run_trained_state(refs, "tiny_texture_trained_state", tiny8, seed=1, teacher_seed=3, steps=400, B=2, S=8, N=6)
    run_film_case(refs, "tiny_texture_fwd_trained", tiny8, seed=1, sigma_gain=60.0, B=2, S=8, N=6, hier=True, kwargs=relu,
                  state="tiny_texture_trained_state")
    run_grad_case(refs, "tiny_texture_grad_trained", tiny8, seed=1, sigma_gain=60.0, B=2, S=8, N=6,
                  kwargs=dict(clamp_mode="relu", nerf_noise=0.2, white_back=True), state="tiny_texture_trained_state")
    # hidden widths beyond the powers of two (round 5; round-4 review, next-round #8): the reference's constructors take any width
    # (siren.py:1455); H = 96 with a 24^3 grid forward + the reference's autograd, H = 192 baseline forward
    h96 = proc.model_spec("texture", hidden_dim=96, grid_size=24, z_dim=16)
    run_film_case(refs, "h96_texture_8x8_n12", h96, seed=9, sigma_gain=300.0, B=2, S=8, N=12, hier=True, kwargs=relu)
    run_grad_case(refs, "h96_texture_grad", proc.model_spec("texture", hidden_dim=96, grid_size=6, z_dim=16), seed=17, sigma_gain=60.0,
                  B=2, S=6, N=8, kwargs=dict(clamp_mode="relu", nerf_noise=0.2, white_back=True))
    run_film_case(refs, "h192_baseline_8x8_n12", proc.model_spec("baseline", hidden_dim=192, z_dim=16), seed=10, sigma_gain=300.0, B=1, S=8, N=12,
                  hier=True, kwargs=relu)
    run_inversion_case(refs, "tiny_texture_inversion", tiny8, "tiny_texture_trained_state", seed=1, n_iterations=30, n_mean_latents=500,
                       S=8, N=12)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
