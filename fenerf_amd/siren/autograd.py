"""Differentiable evaluation of the fused SIREN kernel -- what the reference gets from torch autograd
over <siren>.forward_with_frequencies_phase_shifts (siren.py:1509-1530) in the generator step
(train_double_latent_semantic.py: g_loss.backward()) and in inversion (inverse_render_double_semantic.py).

Data path on the GPU, all native (include/fenerf.h):
    forward   fenerf_siren_forward_save   the fused fp32-MFMA kernel; keeps the pre-FiLM accumulators ("tape")
    backward  fenerf_siren_backward       one fused chain kernel: d_out -> dL/dtheta of every FiLM layer, d(grid features)
              fenerf_grid_backward        trilinear scatter into the feature-grid gradient
What remains are contractions over the point axis -- weight / bias / FiLM gradients -- which are plain library GEMMs and
reductions on (d_t, tape); they are issued here through torch (rocBLAS), layer by layer so the transient memory is two
[H, points] matrices.
"""
import torch


def _fold_label_head(label_params):
    """The label head is 2-3 Linear layers with no activation between them (siren.py:1490-1494) = one affine map.
    label_params = [(W1,b1), (W2,b2), ...] in application order -> (A [n_lab,H], c [n_lab])."""
    A, c = label_params[0]
    for W, b in label_params[1:]:
        c = W @ c + b
        A = W @ A
    return A, c


class SirenFunction(torch.autograd.Function):
    """out = siren(points, dirs; film params, weights).  Non-tensor arg `module` supplies the native model and roles."""

    @staticmethod
    def forward(ctx, module, points, dirs, fg, pg, fa, pa, *params):
        nat = module.native_differentiable(points.device)
        out, tape, tape_e = nat.siren_forward_save(points, dirs, fg, pg, fa, pa)
        ctx.module, ctx.nat = module, nat
        ctx.has_dirs = dirs is not None
        ctx.save_for_backward(points, dirs if dirs is not None else points.new_empty(0), fg, pg, fa, pa, out, tape,
                              tape_e if tape_e is not None else points.new_empty(0), *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        module, nat = ctx.module, ctx.nat
        points, dirs, fg, pg, fa, pa, out, tape, tape_e, *params = ctx.saved_tensors
        roles = module._roles(params)
        spec = nat.spec
        H, ng, nc, C = spec["hidden_dim"], spec["n_geo"], spec["n_color"], spec["output_dim"]
        L, n_lab = ng + nc, C - 4
        B, P = points.shape[0], points.shape[1]
        Pt = B * P
        d_out = d_out.contiguous().float()
        d_t, d_e = nat.siren_backward(B, P, fg, pg, fa, pa, out, d_out, tape)

        f = torch.cat([fg.reshape(B, ng, H), fa.reshape(B, nc, H)], 1) * 15 + 30        # [B,L,H]
        ph = torch.cat([pg.reshape(B, ng, H), pa.reshape(B, nc, H)], 1)
        layers = roles["geo"] + roles["color"]
        d_f = torch.empty_like(f)
        d_p = torch.empty_like(f)
        grads = {}
        q = (points.reshape(Pt, 3) * nat.box_scale)
        if ctx.has_dirs:
            dflat = dirs.reshape(Pt, 3)
        else:
            dflat = points.new_tensor([0.0, 0.0, -1.0]).expand(Pt, 3)
        d2 = d_out.reshape(Pt, C)
        x_prev = None                      # X_{l-1} as [H, Pt]
        for l in range(L):
            W, b = layers[l]
            fl = f[:, l].t().unsqueeze(-1)                              # [H,B,1]
            pl = ph[:, l].t().unsqueeze(-1)
            T = tape[l].view(H, B, P)
            D = d_t[l].view(H, B, P)
            Z = T + b.view(H, 1, 1)
            d_p[:, l] = D.sum(-1).t()
            d_f[:, l] = (D * Z).sum(-1).t()
            DZ = (D * fl).reshape(H, Pt)
            grads[id(b)] = DZ.sum(-1)
            if l == 0:
                dW = DZ @ q
            elif l == ng:
                parts = [DZ @ dflat]
                if tape_e.numel():
                    parts.append(DZ @ tape_e)
                parts.append(DZ @ x_prev.t())
                dW = torch.cat(parts, 1)
            else:
                dW = DZ @ x_prev.t()
            grads[id(W)] = dW
            x_l = torch.sin(fl * Z + pl).reshape(H, Pt)
            if l == ng - 1:                 # the heads read the trunk output
                sw, sb = roles["sigma"]
                grads[id(sw)] = d2[:, C - 1:].t() @ x_l.t()
                grads[id(sb)] = d2[:, C - 1].sum().reshape(1)
                if n_lab > 0:
                    dA = d2[:, :n_lab].t() @ x_l.t()
                    dc = d2[:, :n_lab].sum(0)
                    with torch.enable_grad():
                        leaves = [(Wi.detach().requires_grad_(True), bi.detach().requires_grad_(True)) for Wi, bi in roles["label"]]
                        A, c = _fold_label_head(leaves)
                        flat = [t for pair in leaves for t in pair]
                        g = torch.autograd.grad([A, c], flat, [dA, dc], allow_unused=True)
                    for (Wi, bi), gw, gb in zip(roles["label"], g[0::2], g[1::2]):
                        grads[id(Wi)] = gw if gw is not None else torch.zeros_like(Wi)
                        grads[id(bi)] = gb if gb is not None else torch.zeros_like(bi)
            if l == L - 1:
                rw, rb = roles["rgb"]
                s = out.reshape(Pt, C)[:, C - 4:C - 1]
                dpre = d2[:, C - 4:C - 1] * (s * (1 - s))
                grads[id(rw)] = dpre.t() @ x_l.t()
                grads[id(rb)] = dpre.sum(0)
            x_prev = x_l
        if roles["grid"] is not None:
            grads[id(roles["grid"])] = nat.grid_backward(points, d_e, roles["grid"].shape[2:]).contiguous()

        d_f = d_f * 15
        need = ctx.needs_input_grad
        g_fg = d_f[:, :ng].reshape(B, ng * H) if need[3] else None
        g_pg = d_p[:, :ng].reshape(B, ng * H) if need[4] else None
        g_fa = d_f[:, ng:].reshape(B, nc * H) if need[5] else None
        g_pa = d_p[:, ng:].reshape(B, nc * H) if need[6] else None
        g_params = tuple(grads[id(p)].reshape(p.shape) if need[7 + i] else None for i, p in enumerate(params))
        return (None, None, None, g_fg, g_pg, g_fa, g_pa) + g_params


def siren_apply(module, points, dirs, fg, pg, fa, pa):
    if points.requires_grad or (dirs is not None and dirs.requires_grad):
        raise NotImplementedError("fenerf_amd: gradients wrt sample positions / view directions are not provided "
                                  "(the reference's training and inversion loops do not use them)")
    params = module._render_params()
    return SirenFunction.apply(module, points, dirs, fg, pg, fa, pa, *params)
