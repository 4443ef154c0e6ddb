"""TEST INFRASTRUCTURE ONLY -- differentiable (torch autograd, CPU, float64 by default) restatement of the parts of the
hot path whose gradients the HIP backward kernels produce.  Only tests/ import it as the checker (and tools/bench_gstep.py as
the eager-PyTorch baseline it times the native path against); the product (fenerf_amd/) never does.

Pinned twice: its gradients reproduce the REFERENCE's own autograd gradients on tests/golden/tiny_texture_grad.npz
(tests/test_oracle_golden.py::test_grad_oracle_matches_reference_autograd; fixture made by tools/make_golden.py::run_grad_case
importing the reference), and its forward values are pinned against the numpy oracle (oracle/fenerf_oracle.py, itself pinned against reference golden
vectors) in tests/test_oracle_golden.py::test_grad_oracle_matches_numpy_oracle; the gradients are then whatever torch
autograd derives from that forward -- which is exactly what the reference's training loop gets
(train_double_latent_semantic.py: g_loss.backward() through generators.py:519 / volumetric_rendering.py:23-50).
"""
import torch
import torch.nn.functional as F


def composite(rows, z, noise=None, noise_std=0.0, clamp_mode="relu", last_back=False, white_back=False, black_back=False):
    """rows [BR,M,C], z [BR,M] (already sorted), noise [BR,M] un-scaled.  volumetric_rendering.py:23-50 (fill_mode None).
    -> rgb [BR,C-1], depth [BR], weights [BR,M]"""
    rgbs, sig = rows[..., :-1], rows[..., -1]
    deltas = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    x = sig + (noise * noise_std if noise is not None else 0.0)
    act = F.softplus(x) if clamp_mode == "softplus" else torch.relu(x)
    alphas = 1 - torch.exp(-deltas * act)
    if rows.shape[1] == 1:
        alphas = alphas * 0     # reference quirk: a single sample composites to zero (empty deltas broadcast)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    weights = alphas * torch.cumprod(shifted, -1)[:, :-1]
    wsum = weights.sum(-1, keepdim=True)
    if last_back:
        weights = torch.cat([weights[:, :-1], weights[:, -1:] + (1 - wsum)], -1)
    rgb = (weights[..., None] * rgbs).sum(1)
    depth = (weights * z).sum(1)
    if white_back:
        rgb = rgb + 1 - wsum
    if black_back:
        rgb = rgb - (1 - wsum)
    return rgb, depth, weights


def merge_composite(fine, coarse, z_fine, z_coarse, noise=None, **kw):
    """generators.py:500-520: cat fine|coarse, stable sort by depth, gather, composite."""
    rows = torch.cat([fine, coarse], 1)
    z = torch.cat([z_fine, z_coarse], 1)
    zs, idx = torch.sort(z, dim=1, stable=True)
    rows = torch.gather(rows, 1, idx[..., None].expand(-1, -1, rows.shape[-1]))
    return composite(rows, zs, noise, **kw)


BOX_SCALE = 2 / 0.24


def _film(x, w, b, freq, phase, taps=None):
    theta = freq[:, None, :] * (x @ w.T + b) + phase[:, None, :]
    if taps is not None:
        theta.retain_grad()
        taps.append(theta)
    return torch.sin(theta)


def siren_forward(sd, spec, points, ray_dirs, freq_geo, phase_geo, freq_app, phase_app, taps=None):
    """Differentiable restatement of siren.py:1509-1530 / :1210-1229 / :227-244.  sd: reference-named torch tensors
    (leaves with requires_grad as wanted); raw FiLM parameters split as (geo [B,n_geo*H], app [B,n_color*H]).
    points / ray_dirs [B,P,3] -> [B,P,output_dim].  taps: optional list that receives every FiLM layer's theta (retain_grad
    set), so tests can read dL/dtheta -- what fenerf_siren_backward writes."""
    H = spec["hidden_dim"]
    fg, fa = freq_geo * 15 + 30, freq_app * 15 + 30
    x = points * BOX_SCALE
    feats = None
    if spec["grid_ch"]:
        B, P = points.shape[:2]
        g = sd["spatial_embeddings"].expand(B, -1, -1, -1, -1)
        feats = F.grid_sample(g, x.reshape(B, 1, 1, P, 3), mode="bilinear", padding_mode="zeros", align_corners=True)
        feats = feats.reshape(B, -1, P).permute(0, 2, 1)
    for i in range(spec["n_geo"]):
        x = _film(x, sd[f"network.{i}.layer.weight"], sd[f"network.{i}.layer.bias"], fg[:, i * H:(i + 1) * H],
                  phase_geo[:, i * H:(i + 1) * H], taps)
    sigma = x @ sd["final_layer.weight"].T + sd["final_layer.bias"]
    if spec["kind"] == "spatial":
        c = _film(torch.cat([ray_dirs, x], -1), sd["color_layer_sine.layer.weight"], sd["color_layer_sine.layer.bias"], fa, phase_app, taps)
        rgb = torch.sigmoid(c @ sd["color_layer_linear.0.weight"].T + sd["color_layer_linear.0.bias"])
        return torch.cat([rgb, sigma], -1)
    labels = x
    for i in range(spec["n_label_layers"]):
        labels = labels @ sd[f"label_layer_linear.{i}.weight"].T + sd[f"label_layer_linear.{i}.bias"]
    c = torch.cat([ray_dirs, feats, x], -1) if feats is not None else torch.cat([ray_dirs, x], -1)
    for i in range(spec["n_color"]):
        c = _film(c, sd[f"color_layer_sine.{i}.layer.weight"], sd[f"color_layer_sine.{i}.layer.bias"], fa[:, i * H:(i + 1) * H],
                  phase_app[:, i * H:(i + 1) * H], taps)
    rgb = torch.sigmoid(c @ sd["color_layer_linear.0.weight"].T + sd["color_layer_linear.0.bias"])
    return torch.cat([labels, rgb, sigma], -1)
