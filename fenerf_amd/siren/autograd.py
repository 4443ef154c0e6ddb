"""Differentiable evaluation of the fused SIREN kernel -- what the reference gets from torch autograd
over <siren>.forward_with_frequencies_phase_shifts (siren.py:1509-1530) in the generator step
(train_double_latent_semantic.py: g_loss.backward()) and in inversion (inverse_render_double_semantic.py).

Data path on the GPU, all native (include/fenerf.h):
    forward   fenerf_siren_forward_save   the fused MFMA kernel; keeps the pre-FiLM accumulators ("tape")
    backward  fenerf_siren_backward       one fused chain kernel: d_out -> dL/dtheta of every FiLM layer, the per-tile FiLM
                                          sums, d(grid features)
              fenerf_siren_param_grads    weight / bias / FiLM gradients (contractions over the point axis)
              fenerf_grid_backward        trilinear scatter into the feature-grid gradient
What is left to torch is the fold of the activation-free label head (tiny products) and handing the buffers to autograd.
"""
import torch


def _fold_label_head(label_params):
    """The label head is 2-3 Linear layers with no activation between them (siren.py:1490-1494) = one affine map.
    label_params = [(W1,b1), (W2,b2), ...] in application order -> (A [n_lab,H], c [n_lab])."""
    c = label_params[0][1]
    for W, b in label_params[1:]:
        c = W @ c + b
    # the matrix product from the OUTPUT side: every product (and every product of its backward) then has the n_lab = 18 rows
    # of the last layer as one dimension instead of being H x H x H -- rocBLAS spends 70 us on each of those
    A = label_params[-1][0]
    for W, _ in reversed(label_params[:-1]):
        A = A @ W
    return A, c


def assemble_param_grads(module, nat, params, r, points, d_e, need_params):
    """FenerfSirenGrads buffers (dict r) -> gradients in the order of `params` (module._render_params())."""
    roles = module._roles(params)
    n_lab = nat.spec["output_dim"] - 4
    grads = {}
    for (W, b), gw, gb in zip(roles["geo"] + roles["color"], r["geo_w"] + r["color_w"], r["geo_b"] + r["color_b"]):
        grads[id(W)], grads[id(b)] = gw, gb
    sw, sb = roles["sigma"]
    grads[id(sw)], grads[id(sb)] = r["head_w"][n_lab:n_lab + 1], r["head_b"][n_lab:n_lab + 1]
    if n_lab > 0:      # back through the fold of the activation-free label head (tiny H x H products)
        with torch.enable_grad():
            leaves = [(Wi.detach().requires_grad_(True), bi.detach().requires_grad_(True)) for Wi, bi in roles["label"]]
            A, c = _fold_label_head(leaves)
            flat = [t for pair in leaves for t in pair]
            g = torch.autograd.grad([A, c], flat, [r["head_w"][:n_lab], r["head_b"][:n_lab]], allow_unused=True)
        for (Wi, bi), gw, gb in zip(roles["label"], g[0::2], g[1::2]):
            grads[id(Wi)] = gw if gw is not None else torch.zeros_like(Wi)
            grads[id(bi)] = gb if gb is not None else torch.zeros_like(bi)
    rw, rb = roles["rgb"]
    grads[id(rw)], grads[id(rb)] = r["rgb_w"], r["rgb_b"]
    if roles["grid"] is not None:
        grads[id(roles["grid"])] = nat.grid_backward(points, d_e, roles["grid"].shape[2:]).contiguous()
    return tuple(grads[id(p)].reshape(p.shape) if need_params[i] else None for i, p in enumerate(params))


class SirenFunction(torch.autograd.Function):
    """out = siren(points, dirs; film params, weights).  Non-tensor arg `module` supplies the native model and roles."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # under autocast (the reference's training loop) inputs arrive as fp16
    def forward(ctx, module, points, dirs, fg, pg, fa, pa, *params):
        nat = module.native_differentiable(points.device)
        out, tape, tape_e = nat.siren_forward_save(points, dirs, fg, pg, fa, pa)
        ctx.module, ctx.nat = module, nat
        ctx.has_dirs = dirs is not None
        ctx.save_for_backward(points, dirs if dirs is not None else points.new_empty(0), fg, pg, fa, pa, out, tape,
                              tape_e if tape_e is not None else points.new_empty(0), *params)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_out):
        module, nat = ctx.module, ctx.nat
        points, dirs, fg, pg, fa, pa, out, tape, tape_e, *params = ctx.saved_tensors
        roles = module._roles(params)
        spec = nat.spec
        ng, C = spec["n_geo"], spec["output_dim"]
        n_lab = C - 4
        B, P = points.shape[0], points.shape[1]
        d_out = d_out.contiguous().float()
        need = ctx.needs_input_grad
        film_only = not any(need[7:])        # inversion: only the FiLM parameters are optimised
        d_t, d_e = nat.siren_backward(B, P, fg, pg, fa, pa, out, d_out, tape)
        r = nat.siren_param_grads(points, dirs if ctx.has_dirs else None, fg, pg, fa, pa, out, d_out, tape,
                                  tape_e if tape_e.numel() else None, d_t, film_only=film_only)
        film_grads = (r["d_freq_geo"] if need[3] else None, r["d_phase_geo"] if need[4] else None,
                      r["d_freq_app"] if need[5] else None, r["d_phase_app"] if need[6] else None)
        if film_only:
            return (None, None, None) + film_grads + (None,) * len(params)
        return (None, None, None) + film_grads + assemble_param_grads(module, nat, params, r, points, d_e, need[7:])


def siren_apply(module, points, dirs, fg, pg, fa, pa):
    """Differentiable SIREN evaluation.  The native path works on whole 32-point tiles per image: other point counts are
    padded here (with the last point; the pads get no gradient because their outputs are sliced away)."""
    if points.requires_grad or (dirs is not None and dirs.requires_grad):
        raise NotImplementedError("fenerf_amd: gradients wrt sample positions / view directions are not provided "
                                  "(the reference's training and inversion loops do not use them)")
    params = module._render_params()
    P = points.shape[1]
    pad = (-P) % 32
    if pad:
        points = torch.cat([points, points[:, -1:].expand(-1, pad, -1)], 1)
        if dirs is not None:
            dirs = torch.cat([dirs, dirs[:, -1:].expand(-1, pad, -1)], 1)
    out = SirenFunction.apply(module, points.contiguous(), dirs.contiguous() if dirs is not None else None, fg, pg, fa, pa, *params)
    return out[:, :P] if pad else out
