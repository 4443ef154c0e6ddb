"""Differentiable final compositing: torch.autograd.Function wrappers whose forward AND backward are the
HIP kernels behind fenerf_composite / fenerf_merge_composite / fenerf_composite_backward (include/fenerf.h) -- what
torch autograd derives for fancy_integration (volumetric_rendering.py:23-50) after the cat / sort / gather of
generators.py:508-519.  Depth is returned detached (the reference's losses never read it); z values and noise are
constants of the graph, exactly as in the reference where they are produced under torch.no_grad()."""
import torch

from .. import native


class CompositeFunction(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # under autocast (the reference's training loop) inputs arrive as fp16
    def forward(ctx, rows, z, noise, opts):
        rgb, depth, _, _ = native.composite(rows, z, noise, opts, want_weights=False, want_wsum=False)
        ctx.opts = opts
        ctx.save_for_backward(rows, z, noise if noise is not None else rows.new_empty(0))
        ctx.mark_non_differentiable(depth)
        return rgb, depth

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_rgb, _g_depth):
        rows, z, noise = ctx.saved_tensors
        lead, M, C = rows.shape[:-2], rows.shape[-2], rows.shape[-1]
        d = native.composite_backward(g_rgb.reshape(-1, C - 1), rows.reshape(-1, M, C), z.reshape(-1, M), ctx.opts,
                                      noise=noise.reshape(-1, M) if noise.numel() else None)
        return d.reshape(*lead, M, C), None, None, None


class MergeCompositeFunction(torch.autograd.Function):
    """fine / coarse [BR,N,C] with their own depths [BR,N] -> (rgb [BR,C-1], depth [BR])."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # under autocast (the reference's training loop) inputs arrive as fp16
    def forward(ctx, fine, coarse, z_fine, z_coarse, noise, opts):
        rgb, depth, _, _, _ = native.merge_composite(fine, coarse, z_fine, z_coarse, noise, opts, want_weights=False,
                                                     want_wsum=False, want_z=False)
        ctx.opts = opts
        ctx.save_for_backward(fine, coarse, z_fine, z_coarse, noise if noise is not None else fine.new_empty(0))
        ctx.mark_non_differentiable(depth)
        return rgb, depth

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_rgb, _g_depth):
        fine, coarse, z_fine, z_coarse, noise = ctx.saved_tensors
        d_f, d_c = native.composite_backward(g_rgb, fine, z_fine, ctx.opts, rows_b=coarse, z_b=z_coarse,
                                             noise=noise if noise.numel() else None)
        return d_f, d_c, None, None, None, None
