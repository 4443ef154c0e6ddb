"""Differentiable final compositing: torch.autograd.Function wrappers whose forward AND backward are the
HIP kernels behind fenerf_composite / fenerf_merge_composite / fenerf_composite_backward (include/fenerf.h) -- what
torch autograd derives for fancy_integration (volumetric_rendering.py:23-50) after the cat / sort / gather of
generators.py:508-519.  Depth is returned detached (the reference's losses never read it); z values and noise are
constants of the graph, exactly as in the reference where they are produced under torch.no_grad()."""
import torch

from .. import _lib, native
from ..siren import autograd as _siren_autograd


class ImageLayoutFunction(torch.autograd.Function):
    """pixels [B, S*S, C] in [0, 1] -> the image the generators return, [B, C, S, S] * 2 - 1 (generators.py:519-521: reshape,
    permute(0, 3, 1, 2).contiguous(), * 2 - 1), as ONE launch forward and one backward instead of three and three (2 x exact in fp32, so
    the fused a + 2 b rounds once like the reference's (2 b) - 1: bit-identical).  Values only; not twice differentiable."""

    @staticmethod
    def forward(ctx, pixels, B, S):
        v = pixels.reshape(B, S, S, -1).permute(0, 3, 1, 2)
        out = torch.empty(v.shape, dtype=pixels.dtype, device=pixels.device)
        torch.add(torch.tensor(-1.0, dtype=pixels.dtype), v, alpha=2.0, out=out)       # a 0-dim CPU tensor is a scalar operand
        ctx.shape = pixels.shape
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        d = torch.empty((g.shape[0], g.shape[2], g.shape[3], g.shape[1]), dtype=g.dtype, device=g.device)
        torch.mul(g.permute(0, 2, 3, 1), 2.0, out=d)
        return d.reshape(ctx.shape), None, None


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


def _hierarchical_forward(ctx, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, params, film_only):
    """Forward of the hierarchical render node (both variants below): coarse forward-save -> (no-grad) coarse weights -> resampled depths
    -> fine forward-save -> merged composite; saves what the backward needs on ctx."""
    dev = origins.device
    nat = module.native_differentiable(dev)
    B, R, N = z_c.shape
    C, H = nat.C, nat.spec["hidden_dim"]
    L = nat.spec["n_geo"] + nat.spec["n_color"]
    P = R * N
    Pp = (P + 31) // 32 * 32                      # whole 32-point tiles per image; pad rows get no gradient
    G = nat.spec["grid_ch"]

    def samples(z):                               # [B,R,N] depths -> padded [B,Pp,3] points
        pts = (origins.unsqueeze(2) + dirs.unsqueeze(2) * z.unsqueeze(-1)).reshape(B, P, 3)   # generators.py:504
        return torch.cat([pts, pts[:, -1:].expand(-1, Pp - P, -1)], 1) if Pp != P else pts

    rd = None
    if not lock_view:
        rd = dirs.unsqueeze(2).expand(-1, -1, N, -1).reshape(B, P, 3)
        rd = (torch.cat([rd, rd[:, -1:].expand(-1, Pp - P, -1)], 1) if Pp != P else rd).contiguous()
    pts2 = torch.empty((2 * B, Pp, 3), dtype=torch.float32, device=dev)
    out2 = torch.empty((2 * B, Pp, C), dtype=torch.float32, device=dev)
    # the tape's format is fixed here: the 16-bit tape (siren.grad_precision = "tape16") for backward passes that take weight gradients,
    # the fp32 tape for FiLM-only ones (inversion); `film_only` as the backward will compute it
    fmt = ctx.tape_format = module.tape_format(nat, film_only=film_only)
    half = nat.tape_words_per_point(fmt) * B * Pp
    tape2 = torch.empty(half + nat.tape_floats(B * Pp, fmt), dtype=torch.float32, device=dev)   # pass 1 | pass 2 + slack
    tape_e2 = torch.empty((2 * B * Pp, 32), dtype=torch.float32, device=dev) if G else None
    pts2[:B] = samples(z_c)
    nat.siren_forward_save(pts2[:B], rd, fg, pg, fa, pa, out=out2[:B], tape=tape2[:half], tape_e=tape_e2[:B * Pp] if G else None, tape_format=fmt)
    coarse = out2[:B, :P].reshape(B * R, N, C)
    zc = z_c.reshape(B * R, N)
    _, _, w_c, _ = native.composite(coarse, zc, noise_c, copts, want_wsum=False)
    z_f = native.resample(zc, w_c, u)
    pts2[B:] = samples(z_f.reshape(B, R, N))
    nat.siren_forward_save(pts2[B:], rd, fg, pg, fa, pa, out=out2[B:], tape=tape2[half:], tape_e=tape_e2[B * Pp:] if G else None, tape_format=fmt)
    fine = out2[B:, :P].reshape(B * R, N, C)
    rgb, depth, _, _, _ = native.merge_composite(fine, coarse, z_f, zc, noise_f, opts, want_weights=False, want_wsum=False, want_z=False)
    ctx.module, ctx.nat, ctx.opts, ctx.dims = module, nat, opts, (B, R, N, P, Pp)
    ctx.pack_generation = nat.pack_generation
    empty = origins.new_empty(0)
    ctx.save_for_backward(pts2, rd if rd is not None else empty, fg, pg, fa, pa, out2, tape2, tape_e2 if G else empty, z_f, zc,
                          noise_f if noise_f is not None else empty, *params)
    ctx.mark_non_differentiable(depth)
    return rgb.reshape(B, R, C - 1), depth.reshape(B, R)


# The render runs through fenerf_render_forward_save / fenerf_render_backward (round 5: chunk planning, workspaces, launch order and gradient
# sums live behind the C-ABI; include/fenerf.h) -- and, in its two-node (DistributedDataParallel) form, fenerf_render_backward_stage 1 / 2.  The
# Python below it is the round-4 orchestration of the same kernels, kept for the two-stream experiment (siren/autograd.py OVERLAP_WGRAD) and
# as the reference the tests compare the library's paths with bit for bit.
USE_RENDER_ABI = True


class HierarchicalRenderFunction(torch.autograd.Function):
    """The whole differentiable hierarchical render of generators.py:479-519 as ONE autograd node: coarse SIREN pass ->
    (no-grad) coarse weights -> resampled depths -> fine SIREN pass -> merged composite.  Both passes write their tapes into
    the two halves of one buffer, so the backward is one composite-backward, ONE chain launch and ONE set of weight-gradient
    launches over 2B "images" (pass-major) -- instead of two of each plus 34 tensor additions when the passes are separate
    SirenFunction nodes.  Inputs: rays (origins / dirs [B,R,3], coarse depths z_c [B,R,N]), the caller's draws (u [B*R,N],
    noise_c / noise_f or None), composite options, raw FiLM parameters, then module._render_params().
    -> (rgb [B,R,C-1], depth [B,R])."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, *params):
        film_only = not any(ctx.needs_input_grad[14:])
        ctx.abi = USE_RENDER_ABI and not (_siren_autograd.OVERLAP_WGRAD)
        if not ctx.abi:
            return _hierarchical_forward(ctx, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, params, film_only=film_only)
        nat = module.native_differentiable(origins.device)
        B, R, N = z_c.shape
        ctx.tape_format = module.tape_format(nat, film_only=film_only)
        rgb, depth, save = nat.render_forward_save(origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, opts, lock_view=lock_view,
                                                   tape_format=ctx.tape_format)
        ctx.module, ctx.nat, ctx.opts, ctx.dims, ctx.lock_view = module, nat, opts, (B, R, N), lock_view
        ctx.pack_generation = nat.pack_generation
        ctx.save_for_backward(save, z_c, noise_f if noise_f is not None else origins.new_empty(0), *params)
        ctx.mark_non_differentiable(depth)
        return rgb, depth

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_rgb, _g_depth):
        module, nat, opts = ctx.module, ctx.nat, ctx.opts
        _siren_autograd.check_same_weights(ctx, nat)
        need = ctx.needs_input_grad
        if ctx.abi:
            B, R, N = ctx.dims
            save, z_c, noise_f, *params = ctx.saved_tensors
            film_only = not any(need[14:])
            r, g_grid = nat.render_backward(B, R, N, save, z_c, noise_f if noise_f.numel() else None, opts, g_rgb.contiguous().float(), film_only,
                                            lock_view=ctx.lock_view, tape_format=ctx.tape_format,
                                            weights=_siren_autograd.film_layer_weights(module, params) if ctx.tape_format else None,
                                            chunk_points=_siren_autograd.BACKWARD_CHUNK_POINTS, film_sums_budget_bytes=_siren_autograd.FILM_SUMS_BUDGET_BYTES)
            film_grads = tuple(r[k] if need[10 + i] else None for i, k in enumerate(("d_freq_geo", "d_phase_geo", "d_freq_app", "d_phase_app")))
            head = (None,) * 10
            if film_only:
                return head + film_grads + (None,) * len(params)
            return head + film_grads + _siren_autograd.assemble_param_grads(module, nat, params, r, None, None, need[14:], d_grid_ncdhw=g_grid)
        B, R, N, P, Pp = ctx.dims
        pts2, rd, fg, pg, fa, pa, out2, tape2, tape_e2, z_f, zc, noise_f, *params = ctx.saved_tensors
        C = nat.C
        fine, coarse = out2[B:, :P].reshape(B * R, N, C), out2[:B, :P].reshape(B * R, N, C)
        if Pp == P:     # whole tiles per image: the composite backward writes the chain's input directly (coarse | fine halves, pass-major)
            d_out2 = torch.empty_like(out2)
            native.composite_backward(g_rgb.reshape(B * R, C - 1), fine, z_f, opts, rows_b=coarse, z_b=zc, noise=noise_f if noise_f.numel() else None,
                                      out_a=d_out2[B:].view(B * R, N, C), out_b=d_out2[:B].view(B * R, N, C))
        else:
            d_f, d_c = native.composite_backward(g_rgb.reshape(B * R, C - 1), fine, z_f, opts, rows_b=coarse, z_b=zc,
                                                 noise=noise_f if noise_f.numel() else None)
            d_out2 = torch.zeros((2 * B, Pp, C), dtype=torch.float32, device=out2.device)
            d_out2[:B, :P] = d_c.reshape(B, P, C)
            d_out2[B:, :P] = d_f.reshape(B, P, C)
        film2 = [torch.cat([t, t]) for t in (fg, pg, fa, pa)]            # pass-major: image b' = pass * B + b
        rd2 = torch.cat([rd, rd]) if rd.numel() else None
        film_only = not any(need[14:])
        r, d_grid = _siren_autograd.chunked_backward(nat, 2 * B, Pp, film2, pts2, rd2, out2, d_out2, tape2,
                                                  tape_e2 if tape_e2.numel() else None, film_only, tape_format=ctx.tape_format,
                                                  weights=_siren_autograd.film_layer_weights(module, params) if ctx.tape_format else None)
        fold = lambda t, ok: (t[:B] + t[B:]) if ok else None
        film_grads = (fold(r["d_freq_geo"], need[10]), fold(r["d_phase_geo"], need[11]), fold(r["d_freq_app"], need[12]),
                      fold(r["d_phase_app"], need[13]))
        head = (None,) * 10
        if film_only:
            return head + film_grads + (None,) * len(params)
        return head + film_grads + _siren_autograd.assemble_param_grads(module, nat, params, r, pts2, d_grid, need[14:])


# ----------------------------------------------------------------------------------------------------------------------------------
# Exact sparsity of the backward pass (round 6; opt-in per module, `siren.sparse_backward = True`).  A sample whose row of upstream
# gradients is ALL ZERO contributes exact zeros to every gradient of the step -- and under the reference's relu clamp that is every
# sample with sigma + noise <= 0: its compositing weight alpha T is 0, so the colour / label channels take 0 * g, and relu' = 0 stops
# the density gradient (volumetric_rendering.py:36-47; torch autograd multiplies the same zeros through the whole SIREN).  In a density
# field that is mostly empty space that is most samples: 89 % of the coarse samples of the bench model (tools/exp/empty_fraction.py).
# So the forward here is the NO-GRAD render kept stage by stage (its 22 outputs per sample are all it saves: 69 MB instead of a 9-GB
# tape), and the backward runs composite-backward over all samples, keeps the samples with a non-zero gradient row (per image, padded
# to whole 32-point tiles with samples whose rows are zero anyway), and only THOSE go through forward-save, the chain and the
# weight-gradient kernels.  Same gradients as the dense node up to the order of the sums; nothing is approximated, nothing is skipped
# that the reference's arithmetic would not multiply by zero.  With another clamp mode (softplus) every row is non-zero and this is
# the dense backward plus a re-evaluation (`siren.sparse_backward = "auto"` then falls back to the dense node, below).  The backward does not wait for the device: the length of its buffers is a bound the FORWARD
# computes (samples with sigma + max|noise| std > 0, + one per ray with last_back: a row is non-zero only if alpha > 0), fetched
# asynchronously; which samples are kept is decided on the device from the rows themselves.
# ----------------------------------------------------------------------------------------------------------------------------------
class SparseHierarchicalRenderFunction(torch.autograd.Function):
    """HierarchicalRenderFunction's signature and results; see the block comment above."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, *params):
        nat = module.native_differentiable(origins.device)
        B, R, N = z_c.shape
        C = nat.C
        coarse = nat.siren_forward_rays(origins, dirs, z_c, fg, pg, fa, pa, lock_view=lock_view).reshape(B * R, N, C)
        zc = z_c.reshape(B * R, N)
        _, _, w_c, _ = native.composite(coarse, zc, noise_c, copts, want_wsum=False)
        z_f = native.resample(zc, w_c, u)
        fine = nat.siren_forward_rays(origins, dirs, z_f.reshape(B, R, N), fg, pg, fa, pa, lock_view=lock_view).reshape(B * R, N, C)
        rgb, depth, _, _, _ = native.merge_composite(fine, coarse, z_f, zc, noise_f, opts, want_weights=False, want_wsum=False, want_z=False)
        P = R * N
        _record_bound(ctx, opts, torch.cat([coarse[..., -1].reshape(B, P), fine[..., -1].reshape(B, P)], 1), noise_f, B, R)
        ctx.module, ctx.nat, ctx.opts, ctx.dims, ctx.lock_view = module, nat, opts, (B, R, N), lock_view
        ctx.pack_generation = nat.pack_generation
        ctx.save_for_backward(origins, dirs, zc, z_f, coarse, fine, noise_f if noise_f is not None else origins.new_empty(0), fg, pg, fa, pa, *params)
        ctx.mark_non_differentiable(depth)
        return rgb.reshape(B, R, C - 1), depth.reshape(B, R)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_rgb, _g_depth):
        module, nat, opts = ctx.module, ctx.nat, ctx.opts
        _siren_autograd.check_same_weights(ctx, nat)
        need = ctx.needs_input_grad
        B, R, N = ctx.dims
        origins, dirs, zc, z_f, coarse, fine, noise_f, fg, pg, fa, pa, *params = ctx.saved_tensors
        d_f, d_c = native.composite_backward(g_rgb.contiguous().float().reshape(B * R, nat.C - 1), fine, z_f, opts, rows_b=coarse, z_b=zc,
                                             noise=noise_f if noise_f.numel() else None)
        return _sparse_siren_backward(ctx, module, nat, need, B, R, N, 2, d_c, d_f, zc, z_f, origins, dirs, (fg, pg, fa, pa), params)


def _record_bound(ctx, opts, sig, noise, B, R):
    """How many samples of an image CAN carry a non-zero gradient row -- computed in the forward so that the backward never waits for the
    device: a row is non-zero only if alpha > 0, i.e. (relu clamp) sigma + noise * std > 0; the noise of a sample is the draw of its SORTED
    position, so the bound takes the largest |draw| of the render; `last_back` adds the ray's last sample (its colour row takes the residual
    weight whatever its density).  NaN densities count.  The softplus clamp keeps every sample.  sig [B, samples of the image]."""
    if opts.clamp_mode == _lib.CLAMP["relu"]:
        if noise is not None and opts.noise_std != 0:
            sig = sig + noise.abs().max() * opts.noise_std
        cap = (~(sig <= 0)).sum(1) + (R if opts.last_back else 0)                          # per image
        ctx.cap_host = torch.empty((B,), dtype=torch.long, pin_memory=True)
        ctx.cap_host.copy_(cap, non_blocking=True)
        ctx.cap_ready = torch.cuda.Event()
        ctx.cap_ready.record()
    else:
        ctx.cap_host, ctx.cap_ready = None, None


def _sparse_siren_backward(ctx, module, nat, need, B, R, N, passes, d_c, d_f, zc, z_f, origins, dirs, film, params):
    """The SIREN part of a sparse backward: d_c (/ d_f) [B*R, N, C] = gradients wrt the outputs of the pass(es) -> the autograd node's
    return tuple (10 Nones, four FiLM gradients, parameter gradients).  passes = 2: coarse | fine; 1: d_f = z_f = None."""
    fg, pg, fa, pa = film
    S = passes * R * N                                                          # samples per image
    dev = origins.device
    # the buffer lengths come from the forward's per-image bounds (that copy finished long ago), so nothing here waits for the device
    if ctx.cap_ready is not None:
        ctx.cap_ready.synchronize()
        caps = [min(S, int(c)) for c in ctx.cap_host.tolist()]
    else:
        caps = [S] * B
    # whole 32-point tiles; beyond 64 Ki samples whole blocks of 2,048 (<= 3 % more zero rows): the buffer lengths of consecutive steps then
    # repeat, and the caching allocator serves them from the blocks it has instead of growing by a new size every step (400 steps with 397
    # different lengths: 34 GB reserved for 0.6 GB allocated, tools/exp/sparse_soak.py)
    caps = [min(-(-S // 32) * 32, -(-c // 2048) * 2048) if c > 65536 else max(32, (c + 31) // 32 * 32) for c in caps]
    # images that keep similar numbers of samples share a launch group (padded to the group's fullest image); a batch of one dense and
    # five nearly empty images is not padded to six dense ones
    groups = plan_sparse_groups(caps, torch.cuda.get_device_properties(dev).multi_processor_count)
    sparse_auto_observe(module, sum(len(g) * c for g, c in groups) / (S * B))
    whole = len(groups) == 1
    perm = None if whole else torch.tensor([b for g, _ in groups for b in g], dtype=torch.long).pin_memory().to(dev, non_blocking=True)
    film_only = not any(need[14:])
    fmt = module.tape_format(nat, film_only=film_only)
    weights = _siren_autograd.film_layer_weights(module, params) if fmt else None
    total, d_grid, film_rows, kept, flags, first = None, None, [], [], [], 0
    for g, cap in groups:
        ids = None if whole else perm[first:first + len(g)]
        first += len(g)
        # kept samples (a row with a non-zero -- NaN != 0: a broken row is kept, not hidden) first, in sample order, coarse pass first; the
        # slots beyond an image's count repeat its first sample with a zero row (native.sparse_select: three launches)
        pts, rd, d_sel, counts = native.sparse_select(d_c, d_f, zc, z_f, origins, dirs, cap, want_dirs=not ctx.lock_view, images=ids)
        film_g = (fg, pg, fa, pa) if whole else tuple(t.index_select(0, ids) for t in (fg, pg, fa, pa))
        out, tape, tape_e = nat.siren_forward_save(pts, rd, *film_g, tape_format=fmt)
        r, d_grid = _siren_autograd.chunked_backward(nat, len(g), cap, film_g, pts, rd, out, d_sel, tape, tape_e, film_only, tape_format=fmt,
                                                  weights=weights, d_grid=d_grid)
        del out, tape, tape_e, d_sel
        film_rows.append([r[k] for k in _siren_autograd.FILM_KEYS])
        if not film_only:
            if total is None:
                total = r
            else:
                _siren_autograd._add_all(_siren_autograd._flat(total, _siren_autograd.FILM_KEYS), _siren_autograd._flat(r, _siren_autograd.FILM_KEYS))
        kept.append(counts[:len(g)].sum())
        flags.append(counts[len(g)])
    # (a count above its bound would mean the bound's argument is wrong: checked without waiting, reported by the next backward)
    cls = SparseHierarchicalRenderFunction
    cls._check_overflow(flags[0] if whole else torch.stack(flags).max())
    cls.last_kept = (kept[0] if whole else torch.stack(kept).sum(), S * B)            # for reports (a device scalar: read it after the step)
    cls.last_groups = [(list(g), c) for g, c in groups]
    if whole:
        film_g = film_rows[0]
    else:       # rows back into image order
        film_g = [torch.cat([rows[i] for rows in film_rows], 0).index_select(0, torch.argsort(perm)) for i in range(len(_siren_autograd.FILM_KEYS))]
    fr = dict(zip(_siren_autograd.FILM_KEYS, film_g))
    film_grads = tuple(fr[k] if need[10 + i] else None for i, k in enumerate(("d_freq_geo", "d_phase_geo", "d_freq_app", "d_phase_app")))
    head = (None,) * 10
    if film_only:
        return head + film_grads + (None,) * len(params)
    return head + film_grads + _siren_autograd.assemble_param_grads(module, nat, params, total, None, d_grid, need[14:])


class SparseSinglePassRenderFunction(torch.autograd.Function):
    """The render WITHOUT importance resampling (hierarchical_sample False: generators.py:479-483 + :519 on the coarse samples -- the
    reference's inversion renders, inverse_render_double_semantic.py:225-247) as one node with the exact-sparsity backward of the block
    comment above: SparseHierarchicalRenderFunction's signature (u and noise_c unused) and results."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, *params):
        nat = module.native_differentiable(origins.device)
        B, R, N = z_c.shape
        C = nat.C
        rows = nat.siren_forward_rays(origins, dirs, z_c, fg, pg, fa, pa, lock_view=lock_view).reshape(B * R, N, C)
        zc = z_c.reshape(B * R, N)
        rgb, depth, _, _ = native.composite(rows, zc, noise_f, opts, want_weights=False, want_wsum=False)
        _record_bound(ctx, opts, rows[..., -1].reshape(B, R * N), noise_f, B, R)
        ctx.module, ctx.nat, ctx.opts, ctx.dims, ctx.lock_view = module, nat, opts, (B, R, N), lock_view
        ctx.pack_generation = nat.pack_generation
        ctx.save_for_backward(origins, dirs, zc, rows, noise_f if noise_f is not None else origins.new_empty(0), fg, pg, fa, pa, *params)
        ctx.mark_non_differentiable(depth)
        return rgb.reshape(B, R, C - 1), depth.reshape(B, R)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_rgb, _g_depth):
        module, nat, opts = ctx.module, ctx.nat, ctx.opts
        _siren_autograd.check_same_weights(ctx, nat)
        B, R, N = ctx.dims
        origins, dirs, zc, rows, noise_f, fg, pg, fa, pa, *params = ctx.saved_tensors
        C = nat.C
        d = native.composite_backward(g_rgb.contiguous().float().reshape(B * R, C - 1), rows, zc, opts, noise=noise_f if noise_f.numel() else None)
        return _sparse_siren_backward(ctx, module, nat, ctx.needs_input_grad, B, R, N, 1, d, None, zc, None, origins, dirs, (fg, pg, fa, pa), params)


SparseHierarchicalRenderFunction.last_groups = None
SparseHierarchicalRenderFunction.last_kept = None
SparseHierarchicalRenderFunction._pending = None


def _check_overflow(overflowed):
    """Deferred assertion of the sparse backward's buffer bound: the flag of THIS call is copied to the host without waiting and read by the
    next call (or by SparseHierarchicalRenderFunction.verify()); a set flag means samples were dropped -- an error, never a silent result."""
    cls = SparseHierarchicalRenderFunction
    cls.verify(wait=False)
    flag = torch.empty((), dtype=torch.bool, pin_memory=True)
    flag.copy_(overflowed != 0, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    cls._pending = (flag, ev)


def _verify(wait=True):
    cls = SparseHierarchicalRenderFunction
    if cls._pending is None:
        return
    flag, ev = cls._pending
    if not wait and not ev.query():
        return
    ev.synchronize()
    cls._pending = None
    if bool(flag):
        raise RuntimeError("fenerf_amd: sparse backward: more samples carried a non-zero gradient row than the forward's bound allowed for; "
                           "the gradients of that backward pass are incomplete (please report; siren.sparse_backward = False avoids it)")


SparseHierarchicalRenderFunction._check_overflow = staticmethod(_check_overflow)
SparseHierarchicalRenderFunction.verify = staticmethod(_verify)


# `siren.sparse_backward = "auto"`: the sparse node while it pays, the dense node otherwise.  What the sparse backward costs is set by the
# length of its buffers -- the forward's bound on the samples with a non-zero row, as a fraction of all samples -- and that number is on
# the host when the backward starts (no wait).  Model of the two steps at configs[1] (bench.py legs gstep / gstep_sparse):
# dense = forward-save + chain + weight gradients over everything; sparse = a no-grad render + the same over the fraction f, i.e.
# sparse / dense ~ 0.35 + 0.93 f: break-even at f ~ 0.7.  While the last observed fraction is above SPARSE_AUTO_MAX_FRACTION the dense node
# runs and every SPARSE_AUTO_PROBE_EVERY-th step is a sparse one that observes again (a probe at f = 1 costs + 28 % of one step).
SPARSE_AUTO_MAX_FRACTION = 0.6
SPARSE_AUTO_PROBE_EVERY = 50
# ... and only for renders of at least this many samples: the sparse step has about twice the dense step's fixed cost (more launches, the
# wait for the forward's bound), which short kernels do not pay back -- 4 x 32 x 32 x 12+12 (98,304 samples, 13 % kept): 2.37 against 2.17 ms;
# 6 x 64 x 64 x 12+12 (589,824, 21 % kept): 6.51 against 9.55 ms (tools/exp/gstep_small_shapes.py)
SPARSE_AUTO_MIN_SAMPLES = 262144


def plan_sparse_groups(caps, n_cus=256, group_cost=0.6):
    """caps[b]: slots image b needs (a multiple of 32).  -> [(images, cap)]: a partition of the batch into launch groups, each padded to its
    fullest image, that minimises  sum over groups of (rounds of 128-point workgroups over the CUs + group_cost)  -- the SIREN kernels are
    persistent, one workgroup per CU, so a group's time goes with ceil(images * cap / (128 * CUs)); group_cost (in rounds) stands for what a
    further group adds beside its kernels (five more launches, one more sum over the weight gradients).  Optimal over partitions of the
    images sorted by cap (dynamic programme, O(B^2)); ties go to fewer groups."""
    n = len(caps)
    order = sorted(range(n), key=lambda b: (-caps[b], b))
    unit = 128 * max(1, n_cus)
    best, cut = [0.0] + [float("inf")] * n, [0] * (n + 1)
    for j in range(1, n + 1):
        for i in range(j):
            c = best[i] + -(-((j - i) * caps[order[i]]) // unit) + group_cost
            if c < best[j] - 1e-9:
                best[j], cut[j] = c, i
    groups, j = [], n
    while j > 0:
        i = cut[j]
        groups.append((sorted(order[i:j]), caps[order[i]]))
        j = i
    return groups[::-1]


def _sparse_auto_state(module):
    st = module.__dict__.get("_sparse_auto")
    if st is None:
        st = module.__dict__["_sparse_auto"] = {"fraction": None, "dense_steps": 0, "last": None}
    return st


def sparse_auto_observe(module, fraction):
    """called by the sparse backward: the fraction of the samples its buffers were sized for"""
    _sparse_auto_state(module)["fraction"] = float(fraction)


def sparse_auto_choice(module, samples=None):
    """True: this step's render is the sparse node.  (Nothing observed yet: sparse -- that step is the first observation.)
    samples: samples of the render (all images, all passes); below SPARSE_AUTO_MIN_SAMPLES the dense node."""
    st = _sparse_auto_state(module)
    if samples is not None and samples < SPARSE_AUTO_MIN_SAMPLES:
        st["last"] = "dense"
        return False
    f = st["fraction"]
    if f is None or f <= SPARSE_AUTO_MAX_FRACTION:
        st["dense_steps"], st["last"] = 0, "sparse"
        return True
    st["dense_steps"] += 1
    if st["dense_steps"] >= SPARSE_AUTO_PROBE_EVERY:
        st["dense_steps"], st["last"] = 0, "probe"
        return True
    st["last"] = "dense"
    return False


# ----------------------------------------------------------------------------------------------------------------------------------
# The same render as TWO autograd nodes (round 4): for DistributedDataParallel.  With one node every gradient of the step becomes
# available at the same instant -- the end of the backward -- so DDP's bucketed all-reduce (124 MB, 113 MB of it the 96^3 feature
# grid; train_double_latent_semantic.py:148-150) can only start when there is nothing left to hide it under.  But the grid gradient is
# FINAL as soon as the last chain launch has scattered into it, with all the weight-gradient launches (a quarter of the step) still to
# run.  Split: the render stage's backward runs the composite backward and every chunk's chain, and returns the grid gradient; the
# engine hands it to the parameter (AccumulateGrad nodes run before any other ready node), DDP's hook starts the all-reduce of the
# grid's bucket on its communication stream, and only then the weight stage's backward runs the weight-gradient kernels over the dumps
# the chains left.  Price: the d(theta) dumps of all chunks are alive together (as large as the tape).  Opt-in per module
# (`siren.split_backward = True`, fenerf_amd.dist.prepare_for_ddp) -- DDP must also be allowed to reduce buckets in the order the
# gradients arrive (find_unused_parameters=False lets it rebuild its buckets after the first step; with the reference's
# find_unused_parameters=True the grid's bucket stays last in line and nothing overlaps).  What the render stage leaves for the weight
# stage lives for one backward pass only (an engine callback drops it): torch.autograd.grad on a subset of the inputs is supported in
# the sense that it returns what it can and leaks nothing; it does not deliver weight gradients unless the weight stage's inputs are asked for.
# ----------------------------------------------------------------------------------------------------------------------------------
# d(theta) dumps the render stage of a split backward leaves for the weight stage: the last SPLIT_KEEP_CHUNKS backward chunks' (a chunk = one
# pass of a 128 x 128 x 24 image = 4.4 GB at H = 256).  Two chunks = 3.6 ms of weight-gradient kernels for the 113-MB grid all-reduce to run
# beside (a ring all-reduce of that size over 8 GPUs' xGMI links is 1 - 2 ms), 8.8 GB instead of all 12 chunks' 52 GB at configs[2]'s
# 6-image micro-batch (round-4 review: peak 108 GB against 56.6 one-node; measured round 5: 76.1 GB with four chunks kept).
SPLIT_KEEP_CHUNKS = 2


class _SplitState:
    """what the render stage's backward leaves for the weight stage's backward of the same render"""
    def __init__(self):
        self.work = None


class HierarchicalWeightStage(torch.autograd.Function):
    """Upstream node: owns the raw FiLM parameters and every render weight EXCEPT the grid.  forward: a token; backward: the
    weight-gradient kernels over the dumps of the render stage."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, state, module, fg, pg, fa, pa, *params_no_grid):
        ctx.state, ctx.module = state, module
        return fg.new_zeros(1)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, _g_token):
        w, ctx.state.work = ctx.state.work, None
        if w is None:
            raise RuntimeError("fenerf_amd: split backward: the render stage has not run (or ran twice) before the weight stage")
        module, nat, B = ctx.module, w["nat"], w["B"]
        need = ctx.needs_input_grad
        if w.get("abi"):        # fenerf_render_backward_stage(2): the kept chunks' weight gradients, the sums, the FiLM fold -- all in the library
            try:
                r = nat.render_backward_stage(2, w["keep"], B, w["R"], w["N"], w["save"], None, None, w["opts"], None, lock_view=w["lock_view"],
                                              tape_format=w["tape_format"], weights=w["weights"], chunk_points=w["chunk_points"], carry=w["carry"])
            finally:
                nat.release_split_workspace(w["carry"])
            film_grads = tuple(r[k] if need[2 + i] else None for i, k in enumerate(("d_freq_geo", "d_phase_geo", "d_freq_app", "d_phase_app")))
            params = w["params"]
            grid = module._roles(params)["grid"]
            all_grads = _siren_autograd.assemble_param_grads(module, nat, params, r, None, None, [True] * len(params))
            no_grid = [g for p_, g in zip(params, all_grads) if p_ is not grid]
            return (None, None) + film_grads + tuple(g if need[6 + i] else None for i, g in enumerate(no_grid))
        r = _siren_autograd.run_weight_grads(nat, 2 * B, w["Pp"], w["film2"], w["pts2"], w["rd2"], w["out2"], w["d_out2"], w["tape2"], w["tape_e2"],
                                             w["chunks"], w["dumps"], tape_format=w["tape_format"],
                                             weights=_siren_autograd.film_layer_weights(module, w["params"]) if w["tape_format"] else None,
                                             acc=w["acc"])
        fold = lambda t, ok: (t[:B] + t[B:]) if ok else None
        film_grads = (fold(r["d_freq_geo"], need[2]), fold(r["d_phase_geo"], need[3]), fold(r["d_freq_app"], need[4]), fold(r["d_phase_app"], need[5]))
        params = w["params"]                                    # module._render_params() order, grid included
        grid = module._roles(params)["grid"]
        all_grads = _siren_autograd.assemble_param_grads(module, nat, params, r, w["pts2"], None, [True] * len(params))
        no_grid = [g for p_, g in zip(params, all_grads) if p_ is not grid]
        return (None, None) + film_grads + tuple(g if need[6 + i] else None for i, g in enumerate(no_grid))


class HierarchicalRenderSplitFunction(torch.autograd.Function):
    """Downstream node: the render itself.  Differentiable inputs: the weight stage's token and the grid.  backward: composite backward,
    every chunk's chain (with the fused grid scatter), the grid gradient; the dumps go to the weight stage through `state`."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, state, token, grid, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa):
        ctx.state = state
        ctx.abi = USE_RENDER_ABI and not _siren_autograd.OVERLAP_WGRAD
        if not ctx.abi:
            return _hierarchical_forward(ctx, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, (),
                                         film_only=False)        # the split form exists for training steps: weight gradients are taken
        nat = module.native_differentiable(origins.device)
        B, R, N = z_c.shape
        ctx.tape_format = module.tape_format(nat, film_only=False)
        rgb, depth, save = nat.render_forward_save(origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa, opts, lock_view=lock_view,
                                                   tape_format=ctx.tape_format)
        ctx.module, ctx.nat, ctx.opts, ctx.dims, ctx.lock_view = module, nat, opts, (B, R, N), lock_view
        ctx.pack_generation = nat.pack_generation
        ctx.save_for_backward(save, z_c, noise_f if noise_f is not None else origins.new_empty(0))
        ctx.mark_non_differentiable(depth)
        return rgb, depth

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_rgb, _g_depth):
        module, nat, opts = ctx.module, ctx.nat, ctx.opts
        _siren_autograd.check_same_weights(ctx, nat)
        if ctx.abi:      # fenerf_render_backward_stage(1): composite backward, the early chunks whole, the kept chunks' chains, the grid gradient
            B, R, N = ctx.dims
            save, z_c, noise_f = ctx.saved_tensors
            keep = max(1, int(getattr(module, "split_keep_chunks", SPLIT_KEEP_CHUNKS)))
            params = module._render_params()
            weights = _siren_autograd.film_layer_weights(module, params) if ctx.tape_format else None
            g_grid, carry = nat.render_backward_stage(1, keep, B, R, N, save, z_c, noise_f if noise_f.numel() else None, opts, g_rgb.contiguous().float(),
                                                      lock_view=ctx.lock_view, tape_format=ctx.tape_format, weights=weights,
                                                      chunk_points=_siren_autograd.BACKWARD_CHUNK_POINTS)
            state = ctx.state
            state.work = dict(abi=True, nat=nat, B=B, R=R, N=N, save=save, opts=opts, lock_view=ctx.lock_view, tape_format=ctx.tape_format,
                              weights=weights, keep=keep, chunk_points=_siren_autograd.BACKWARD_CHUNK_POINTS, carry=carry, params=params)
            def drop():         # the engine has finished the pass: whatever the weight stage did not consume is dropped (see below)
                w_, state.work = state.work, None
                if w_ is not None:
                    nat.release_split_workspace(w_["carry"])
            torch.autograd.Variable._execution_engine.queue_callback(drop)
            g_token = torch.zeros(1, dtype=torch.float32, device=save.device) if ctx.needs_input_grad[1] else None
            if g_token is None:
                drop()
            return (None, g_token, g_grid if ctx.needs_input_grad[2] else None) + (None,) * 14
        B, R, N, P, Pp = ctx.dims
        pts2, rd, fg, pg, fa, pa, out2, tape2, tape_e2, z_f, zc, noise_f = ctx.saved_tensors
        C = nat.C
        fine, coarse = out2[B:, :P].reshape(B * R, N, C), out2[:B, :P].reshape(B * R, N, C)
        if Pp == P:
            d_out2 = torch.empty_like(out2)
            native.composite_backward(g_rgb.reshape(B * R, C - 1), fine, z_f, opts, rows_b=coarse, z_b=zc, noise=noise_f if noise_f.numel() else None,
                                      out_a=d_out2[B:].view(B * R, N, C), out_b=d_out2[:B].view(B * R, N, C))
        else:
            d_f, d_c = native.composite_backward(g_rgb.reshape(B * R, C - 1), fine, z_f, opts, rows_b=coarse, z_b=zc,
                                                 noise=noise_f if noise_f.numel() else None)
            d_out2 = torch.zeros((2 * B, Pp, C), dtype=torch.float32, device=out2.device)
            d_out2[:B, :P] = d_c.reshape(B, P, C)
            d_out2[B:, :P] = d_f.reshape(B, P, C)
        film2 = [torch.cat([t, t]) for t in (fg, pg, fa, pa)]            # pass-major: image b' = pass * B + b
        rd2 = torch.cat([rd, rd]) if rd.numel() else None
        chunks = _siren_autograd.plan_chunks(2 * B, Pp)
        # Bounded memory (round 5): only the LAST `split_keep_chunks` chunks' dumps are kept for the weight stage -- their weight-gradient
        # kernels are what the grid gradient's all-reduce runs beside; every earlier chunk takes its chain AND its weight gradients right
        # here, its dump freed before the next chain is launched (the one-node schedule), its gradients carried along as a running sum.
        # Same kernels on the same chunks, sums in chunk order: the gradients are those of the one-node backward.
        keep = max(1, int(getattr(module, "split_keep_chunks", SPLIT_KEEP_CHUNKS)))
        early, late = chunks[:max(0, len(chunks) - keep)], chunks[max(0, len(chunks) - keep):]
        tape_e = tape_e2 if tape_e2.numel() else None
        weights = _siren_autograd.film_layer_weights(module, module._render_params()) if ctx.tape_format else None
        acc, d_grid = _siren_autograd.GradSum(), None
        for c in early:
            d1, d_grid = _siren_autograd.run_chains(nat, 2 * B, Pp, film2, pts2, out2, d_out2, tape2, [c], tape_format=ctx.tape_format, d_grid=d_grid)
            _siren_autograd.run_weight_grads(nat, 2 * B, Pp, film2, pts2, rd2, out2, d_out2, tape2, tape_e, [c], d1, tape_format=ctx.tape_format,
                                             weights=weights, acc=acc, finish=False)
        dumps, d_grid = _siren_autograd.run_chains(nat, 2 * B, Pp, film2, pts2, out2, d_out2, tape2, late, tape_format=ctx.tape_format, d_grid=d_grid)
        state = ctx.state
        state.work = dict(nat=nat, B=B, Pp=Pp, film2=film2, pts2=pts2, rd2=rd2, out2=out2, d_out2=d_out2, tape2=tape2, tape_format=ctx.tape_format,
                          tape_e2=tape_e, chunks=late, dumps=dumps, acc=acc if early else None, params=module._render_params())
        # The dumps (as large as the tape) belong to THIS backward pass: if the weight stage does not consume them -- torch.autograd.grad
        # with only the grid as input, an exception between the two stages -- they are dropped when the engine finishes the pass, not
        # when the graph dies; a retained graph's next backward then starts from a clean state instead of overwriting a stale one.
        torch.autograd.Variable._execution_engine.queue_callback(lambda: setattr(state, "work", None))
        g_grid = nat.grid_gradient_ncdhw(d_grid).contiguous() if ctx.needs_input_grad[2] else None
        g_token = torch.zeros(1, dtype=torch.float32, device=out2.device) if ctx.needs_input_grad[1] else None
        if g_token is None:
            ctx.state.work = None          # nothing upstream will consume the dumps
        return (None, g_token, g_grid) + (None,) * 14


def hierarchical_render_split(module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f, fg, pg, fa, pa):
    """The two-node form of HierarchicalRenderFunction.apply(...) for a module with a feature grid (see the banner above)."""
    params = module._render_params()
    grid = module._roles(params)["grid"]
    state = _SplitState()
    token = HierarchicalWeightStage.apply(state, module, fg, pg, fa, pa, *[p_ for p_ in params if p_ is not grid])
    return HierarchicalRenderSplitFunction.apply(state, token, grid, module, opts, copts, lock_view, origins, dirs, z_c, u, noise_c, noise_f,
                                                 fg.detach(), pg.detach(), fa.detach(), pa.detach())
