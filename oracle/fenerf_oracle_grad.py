"""TEST INFRASTRUCTURE ONLY -- differentiable (torch autograd, CPU, float64 by default) restatement of the parts of the
hot path whose gradients the HIP backward kernels produce.  Only tests/ may import it; the product never does.

Forward values are pinned against the numpy oracle (oracle/fenerf_oracle.py, itself pinned against reference golden
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
