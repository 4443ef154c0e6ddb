"""CPU ORACLE, torch edition -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

The same path as oracle/fenerf_oracle.py (the numpy restatement that is pinned against the reference's own outputs), restated on torch
CPU tensors with the ATen statements the reference itself executes -- F.grid_sample, nn.functional.linear, torch.sin, cumprod,
searchsorted, sort / gather -- so that EVERY pass of the render (not only the BLAS GEMMs) runs on all host cores, as the reference's
CPU path does under torch.set_num_threads(all cores).  BASELINE.md §5 names this as the `cpu_baseline` of bench.py; the numpy oracle stays
the parity checker.  Only tests/ and bench.py's cpu_baseline leg import this module.

PARITY PINNING: tests/test_oracle_golden.py checks this module against the committed reference fixtures (tests/golden/*.npz, generated
from the imported reference by tools/make_golden.py) and against the numpy oracle on the same inputs.

Reference statements followed (file:line under /root/reference):
    generators/volumetric_rendering.py:109-131   get_initial_rays_trig
    generators/volumetric_rendering.py:133-168   perturb_points / transform_sampled_points
    generators/volumetric_rendering.py:220-248   camera origin, create_cam2world_matrix
    siren/siren.py:113-123, :314-330, :1509-1530 FiLMLayer, sample_from_3dgrid, forward_with_frequencies_phase_shifts
    generators/volumetric_rendering.py:18-106    fancy_integration
    generators/volumetric_rendering.py:259-300   sample_pdf
    generators/generators.py:546-646             staged_forward (point-chunked evaluation, max_batch_size)
All randomness is an input, as in the numpy oracle.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BOX_SCALE = 2 / 0.24                       # UniformBoxWarp(0.24), siren.py:181-187
FILL_COLORS = {"white": 1.0, "black": 0.0, "grey": 0.5, "light_grey": 0.81}


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def state_to_torch(sd):
    """{reference parameter name: numpy array} -> the same with torch tensors (done once, outside any timed region)"""
    return {k: _t(v).float() for k, v in sd.items()}


def normalize_vecs(v):                       # generators/math_utils_torch.py:16-20
    return v / torch.norm(v, dim=-1, keepdim=True)


def get_initial_rays_trig(n, num_steps, fov, resolution, ray_start, ray_end):
    W, H = resolution
    x, y = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="ij")
    x, y = x.T.flatten(), y.T.flatten()
    z = -torch.ones_like(x) / np.tan((2 * math.pi * fov / 360) / 2)
    rays_d_cam = normalize_vecs(torch.stack([x, y, z], -1))
    z_vals = torch.linspace(ray_start, ray_end, num_steps).reshape(1, num_steps, 1).repeat(W * H, 1, 1)
    points = rays_d_cam.unsqueeze(1).repeat(1, num_steps, 1) * z_vals
    return torch.stack(n * [points]), torch.stack(n * [z_vals]), torch.stack(n * [rays_d_cam])


def transform_sampled_points(points, z_vals, ray_directions, u_jitter, theta, phi):
    """theta / phi: the pre-clamp camera angles [n,1]; u_jitter the reference's torch.rand of z_vals' shape"""
    n, R, N, _ = points.shape
    dist = z_vals[:, :, 1:2, :] - z_vals[:, :, 0:1, :]
    offset = (u_jitter - 0.5) * dist
    z_vals = z_vals + offset
    points = points + offset * ray_directions.unsqueeze(2)
    phi = torch.clamp(phi, 1e-5, math.pi - 1e-5)
    origin = torch.zeros((n, 3))
    origin[:, 0:1] = torch.sin(phi) * torch.cos(theta)
    origin[:, 2:3] = torch.sin(phi) * torch.sin(theta)
    origin[:, 1:2] = torch.cos(phi)
    forward = normalize_vecs(normalize_vecs(-origin))
    up = torch.tensor([0.0, 1.0, 0.0]).expand_as(forward)
    left = normalize_vecs(torch.cross(up, forward, dim=-1))
    up = normalize_vecs(torch.cross(forward, left, dim=-1))
    rot = torch.eye(4).unsqueeze(0).repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((-left, up, -forward), dim=-1)
    tr = torch.eye(4).unsqueeze(0).repeat(n, 1, 1)
    tr[:, :3, 3] = origin
    c2w = tr @ rot
    ph = torch.ones((n, R, N, 4))
    ph[..., :3] = points
    tp = torch.bmm(c2w, ph.reshape(n, -1, 4).permute(0, 2, 1)).permute(0, 2, 1).reshape(n, R, N, 4)
    td = torch.bmm(c2w[..., :3, :3], ray_directions.reshape(n, -1, 3).permute(0, 2, 1)).permute(0, 2, 1).reshape(n, R, 3)
    ho = torch.zeros((n, R, 4))
    ho[..., 3] = 1
    to = torch.bmm(c2w, ho.permute(0, 2, 1)).permute(0, 2, 1).reshape(n, R, 4)[..., :3]
    return tp[..., :3], z_vals, td, to, phi, theta


def sample_from_3dgrid(coordinates, grid):
    B, P, _ = coordinates.shape
    s = F.grid_sample(grid.expand(B, -1, -1, -1, -1), coordinates.reshape(B, 1, 1, -1, 3), mode="bilinear", padding_mode="zeros",
                      align_corners=True)
    N, C, H, W, D = s.shape
    return s.permute(0, 4, 3, 2, 1).reshape(N, H * W * D, C)


def film_layer(x, w, b, freq, phase):
    x = F.linear(x, w, b)
    if x.shape[1] != freq.shape[1]:              # siren.py:119-122 (quirk A.7(i) kept: skipped when P == H)
        freq = freq.unsqueeze(1).expand_as(x)
        phase = phase.unsqueeze(1).expand_as(x)
    return torch.sin(freq * x + phase)


def siren_forward(sd, spec, points, ray_dirs, freq_geo, phase_geo, freq_app=None, phase_app=None):
    """points, ray_dirs [B,P,3]; raw frequencies / phases [B, n*H] -> [B,P,output_dim]; sd from state_to_torch"""
    H = spec["hidden_dim"]
    fg = freq_geo * 15 + 30
    x = points * BOX_SCALE
    feats = sample_from_3dgrid(x, sd["spatial_embeddings"]) if spec["grid_ch"] else None
    for i in range(spec["n_geo"]):
        x = film_layer(x, sd[f"network.{i}.layer.weight"], sd[f"network.{i}.layer.bias"], fg[..., i * H:(i + 1) * H], phase_geo[..., i * H:(i + 1) * H])
    sigma = F.linear(x, sd["final_layer.weight"], sd["final_layer.bias"])
    if spec["kind"] == "spatial":
        c = film_layer(torch.cat([ray_dirs, x], -1), sd["color_layer_sine.layer.weight"], sd["color_layer_sine.layer.bias"], fg[..., -H:], phase_geo[..., -H:])
        return torch.cat([torch.sigmoid(F.linear(c, sd["color_layer_linear.0.weight"], sd["color_layer_linear.0.bias"])), sigma], -1)
    fa = freq_app * 15 + 30
    labels = x
    for i in range(spec["n_label_layers"]):
        labels = F.linear(labels, sd[f"label_layer_linear.{i}.weight"], sd[f"label_layer_linear.{i}.bias"])
    c = torch.cat([ray_dirs, feats, x], -1) if feats is not None else torch.cat([ray_dirs, x], -1)
    for i in range(spec["n_color"]):
        c = film_layer(c, sd[f"color_layer_sine.{i}.layer.weight"], sd[f"color_layer_sine.{i}.layer.bias"], fa[..., i * H:(i + 1) * H], phase_app[..., i * H:(i + 1) * H])
    rgb = torch.sigmoid(F.linear(c, sd["color_layer_linear.0.weight"], sd["color_layer_linear.0.bias"]))
    return torch.cat([labels, rgb, sigma], -1)


def fancy_integration(rgb_sigma, z_vals, noise=None, noise_std=0.5, last_back=False, white_back=False, black_back=False, clamp_mode=None,
                      fill_mode=None, fill_color="black"):
    """noise = the reference's torch.randn(sigmas.shape) draw, un-scaled (None == zeros); fill modes: None, seg_padding_background,
    eval_seg_padding_background, eval_white_back, weight"""
    rgbs, sigmas = rgb_sigma[..., :-1], rgb_sigma[..., -1:]
    deltas = z_vals[:, :, 1:] - z_vals[:, :, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :, :1])], -2)
    nz = (noise * noise_std) if noise is not None else torch.zeros_like(sigmas)
    if clamp_mode == "softplus":
        alphas = 1 - torch.exp(-deltas * F.softplus(sigmas + nz))
    elif clamp_mode == "relu":
        alphas = 1 - torch.exp(-deltas * F.relu(sigmas + nz))
    else:
        raise TypeError("Need to choose clamp mode")
    shifted = torch.cat([torch.ones_like(alphas[:, :, :1]), 1 - alphas + 1e-10], -2)
    weights = alphas * torch.cumprod(shifted, -2)[:, :, :-1]
    weights_sum = weights.sum(2)
    if last_back:
        weights[:, :, -1] += (1 - weights_sum)
    rgb_final = torch.sum(weights * rgbs, -2)
    depth_final = torch.sum(weights * z_vals, -2)
    if white_back:
        rgb_final = rgb_final + 1 - weights_sum
    if black_back:
        rgb_final = rgb_final + (1 - weights_sum) * -1
    low = weights_sum.squeeze(-1) < 0.9
    if fill_mode == "weight":
        return rgb_final, depth_final, weights_sum.expand_as(rgb_final)
    if fill_mode in ("seg_padding_background", "eval_seg_padding_background"):
        rgb_final = torch.cat([torch.zeros_like(rgb_final[..., :1]), rgb_final], -1)
        if fill_color in FILL_COLORS:
            rgb_final[low] = torch.tensor([1.0] + [FILL_COLORS[fill_color]] * (rgb_final.shape[-1] - 1))
        if fill_mode == "seg_padding_background":
            return rgb_final, depth_final, weights
        return rgb_final, depth_final, weights_sum.expand_as(rgb_final)
    if fill_mode == "eval_white_back":
        rgb_final[low] = torch.ones(rgb_final.shape[-1])
        return rgb_final, depth_final, weights_sum.expand_as(rgb_final)
    if fill_mode is not None:
        raise NotImplementedError(f"fill_mode {fill_mode!r}: see the numpy oracle")
    return rgb_final, depth_final, weights


def sample_pdf(bins, weights, u, eps=1e-5):
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, weights.shape[1])
    inds_sampled = torch.stack([below, above], -1).view(u.shape[0], 2 * u.shape[1])
    cdf_g = torch.gather(cdf, 1, inds_sampled).view(u.shape[0], u.shape[1], 2)
    bins_g = torch.gather(bins, 1, inds_sampled).view(u.shape[0], u.shape[1], 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom[denom < eps] = 1
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def render_forward(sd, spec, film, img_size, fov, ray_start, ray_end, num_steps, rand, hierarchical_sample=True, lock_view_dependence=False,
                   clamp_mode="relu", nerf_noise=0.0, last_back=False, white_back=False, black_back=False, fill_mode=None, fill_color="black",
                   softmax_label=False, max_batch_size=50000, return_stages=False):
    """Same contract as fenerf_oracle.render_forward (inputs may be numpy or torch; outputs are torch CPU tensors); `sd` from
    state_to_torch.  The SIREN is evaluated in point chunks of `max_batch_size` per image like staged_forward (generators.py:584-590;
    quirk A.7(ii): the chunks slice points, not rays)."""
    with torch.no_grad():
        film = {k: _t(v).float() for k, v in film.items() if v is not None}
        B = film["freq_geo"].shape[0]
        S, N = img_size, num_steps
        R = S * S
        C = spec["output_dim"]
        pts_cam, z_vals, d_cam = get_initial_rays_trig(B, N, fov, (S, S), ray_start, ray_end)
        pts, z_vals, dirs, origins, pitch, yaw = transform_sampled_points(pts_cam, z_vals, d_cam, _t(rand["u_jitter"]).float(),
                                                                          _t(rand["theta"]).float(), _t(rand["phi"]).float())
        dirs_exp = dirs.unsqueeze(-2).expand(-1, -1, N, -1).reshape(B, R * N, 3)
        if lock_view_dependence:
            dirs_exp = torch.zeros_like(dirs_exp)
            dirs_exp[..., -1] = -1
        fa, pa = film.get("freq_app"), film.get("phase_app")

        def field(points):
            out = torch.zeros((B, points.shape[1], C))
            for b in range(B):
                head = 0
                while head < points.shape[1]:
                    tail = head + max_batch_size
                    out[b:b + 1, head:tail] = siren_forward(sd, spec, points[b:b + 1, head:tail], dirs_exp[b:b + 1, head:tail],
                                                            film["freq_geo"][b:b + 1], film["phase_geo"][b:b + 1],
                                                            None if fa is None else fa[b:b + 1], None if pa is None else pa[b:b + 1])
                    head += max_batch_size
            return out.reshape(B, R, N, C)

        noise = lambda k: None if rand.get(k) is None else _t(rand[k]).float()
        coarse = field(pts.reshape(B, R * N, 3))
        stages = dict(points=pts, z_coarse=z_vals, dirs=dirs, origins=origins, coarse=coarse)
        if hierarchical_sample:
            _, _, w = fancy_integration(coarse, z_vals, noise=noise("noise_coarse"), noise_std=nerf_noise, clamp_mode=clamp_mode)
            w = w.reshape(B * R, N) + 1e-5
            z = z_vals.reshape(B * R, N)
            z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
            fine_z = sample_pdf(z_mid, w[:, 1:-1], _t(rand["u_fine"]).float()).reshape(B, R, N, 1)
            fine_pts = origins.unsqueeze(2).contiguous() + dirs.unsqueeze(2).contiguous() * fine_z.expand(-1, -1, -1, 3).contiguous()
            fine = field(fine_pts.reshape(B, R * N, 3))
            all_out = torch.cat([fine, coarse], dim=-2)
            all_z = torch.cat([fine_z, z_vals], dim=-2)
            _, idx = torch.sort(all_z, dim=-2)
            all_z = torch.gather(all_z, -2, idx)
            all_out = torch.gather(all_out, -2, idx.expand(-1, -1, -1, C))
            stages.update(z_fine=fine_z, fine=fine, all_out=all_out, all_z=all_z)
            noise_final = noise("noise_fine")
        else:
            all_out, all_z = coarse, z_vals
            noise_final = noise("noise_fine") if rand.get("noise_fine") is not None else noise("noise_coarse")
        pixels, depth, third = fancy_integration(all_out, all_z, noise=noise_final, noise_std=nerf_noise, white_back=white_back, last_back=last_back,
                                                 black_back=black_back, clamp_mode=clamp_mode, fill_mode=fill_mode, fill_color=fill_color)
        if softmax_label:
            pixels = torch.cat([torch.nn.Softmax(dim=-1)(pixels[..., :-3]), pixels[..., -3:]], dim=-1)
        img = pixels.reshape((B, S, S, -1)).permute(0, 3, 1, 2).contiguous() * 2 - 1
        out = (img, depth.reshape(B, S, S), third)
        if return_stages:
            stages.update(pixels_flat=pixels, depth_flat=depth)
            return out + (stages,)
        return out
