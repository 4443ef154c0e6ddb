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
import typing

import torch

from .. import native


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


def _fold_label_head_backward(label_params, gA, gc):
    """Gradients of every (W_i, b_i) of the label head from the gradient (gA [n_lab, H], gc [n_lab]) of its fold A = W_{n-1} ... W_0,
    c = sum_i S_i b_i with S_i = W_{n-1} ... W_{i+1}.  Layer i sees y_i = R_i x + q_i (R_i = W_{i-1} ... W_0, q_i = W_{i-1} q_{i-1} + b_{i-1}),
    so  dW_i = S_i^T (gA R_i^T + gc (x) q_i) = S_i^T U_i + db_i (x) q_i,  db_i = S_i^T gc,  U_i = U_{i-1} W_{i-1}^T, U_0 = gA.
    Written out (11 launches for the three-layer head, every product with the n_lab rows as one dimension) instead of replaying the fold
    through torch.autograd.grad (27 launches of ~6 us at the end of every generator step)."""
    Ws = [W.detach() for W, _ in label_params]
    bs = [b.detach() for _, b in label_params]
    n = len(Ws)
    U, q = [gA], [None]
    for i in range(1, n):
        U.append(U[-1] @ Ws[i - 1].t())
        q.append(bs[i - 1] if q[-1] is None else torch.addmv(bs[i - 1], Ws[i - 1], q[-1]))
    out = [None] * n
    S = None                                      # S_{n-1} = identity
    for i in range(n - 1, -1, -1):
        if S is None:
            db = gc
            dW = U[i] if q[i] is None else torch.addr(U[i], gc, q[i])
        else:
            db = S.t() @ gc
            dW = S.t() @ U[i] if q[i] is None else torch.addmm(torch.outer(db, q[i]), S.t(), U[i])
        out[i] = (dW, db)
        if i > 0:
            S = Ws[i] if S is None else S @ Ws[i]
    return out


# fenerf_label_head_backward (one or two launches) instead of the torch products above (11 launches); False: A/B runs and the parity test
NATIVE_LABEL_HEAD_BACKWARD = True


def assemble_param_grads(module, nat, params, r, points, d_grid_cl, need_params, d_grid_ncdhw=None):
    """FenerfSirenGrads buffers (dict r) + the channels-last grid gradient chunked_backward accumulated -> gradients in the order of
    `params` (module._render_params()).  d_grid_cl None on a model with a grid: the grid's gradient is delivered elsewhere (split
    backward) and its slot is None here -- or it arrives in the parameter's own layout (d_grid_ncdhw: fenerf_render_backward)."""
    roles = module._roles(params)
    n_lab = nat.spec["output_dim"] - 4
    grads = {}
    for (W, b), gw, gb in zip(roles["geo"] + roles["color"], r["geo_w"] + r["color_w"], r["geo_b"] + r["color_b"]):
        grads[id(W)], grads[id(b)] = gw, gb
    sw, sb = roles["sigma"]
    grads[id(sw)], grads[id(sb)] = r["head_w"][n_lab:n_lab + 1], r["head_b"][n_lab:n_lab + 1]
    if n_lab > 0:      # back through the fold of the activation-free label head (skinny products: n_lab rows on one side of each)
        unfold = native.label_head_backward if (NATIVE_LABEL_HEAD_BACKWARD and r["head_w"].is_cuda) else _fold_label_head_backward
        for (Wi, bi), (gw, gb) in zip(roles["label"], unfold(roles["label"], r["head_w"][:n_lab], r["head_b"][:n_lab])):
            grads[id(Wi)], grads[id(bi)] = gw, gb
    rw, rb = roles["rgb"]
    grads[id(rw)], grads[id(rb)] = r["rgb_w"], r["rgb_b"]
    if roles["grid"] is not None and d_grid_ncdhw is not None:
        grads[id(roles["grid"])] = d_grid_ncdhw
    elif roles["grid"] is not None and d_grid_cl is not None:
        grads[id(roles["grid"])] = nat.grid_gradient_ncdhw(d_grid_cl).contiguous()
    return tuple(grads[id(p)].reshape(p.shape) if (need_params[i] and id(p) in grads) else None for i, p in enumerate(params))


# dtheta of a backward chunk: at most this many points (x L*H*4 B = 4.4 GB at L*H = 2816).  The chain kernel writes dL/dtheta of every
# FiLM layer (as large as the tape) only for the weight-gradient kernels to read it once, so it never needs to exist for more points than
# one chain launch's worth: peak memory of a generator step = tape + one chunk.  Rounds 2-3 kept the chunk at 196,608 points to stay
# under 12 GB; an MI355X has 288 GB, and the step at 1 x 128^2 x 24+24 measures 12.82 / 12.31 / 12.09 / 12.33 ms at 10.5 / 11.7 / 14.1 /
# 18.9 GB peak for chunks of 98,304 / 196,608 / 393,216 / 786,432 points (round 4, tools/overlap_sweep.py, one box): one pass of a
# 128 x 128 x 24 image per chunk.
BACKWARD_CHUNK_POINTS = 393216
# inversion (FiLM gradients only, no dump): bytes of per-tile FiLM sums one chain launch may allocate
FILM_SUMS_BUDGET_BYTES = 1 << 30
# Run the weight gradients of backward chunk i on a second stream BESIDE the chain of chunk i + 1 (CU budgets, fenerf_set_cu_budget).
# Built and measured in round 4, OFF: the backward kernels are limited by what the memory system sustains for their access patterns
# (4.0 - 5.8 TB/s), not by the CUs they occupy, so running two of them at once only makes both slower -- chain 1.12 -> 2.1 ms, square
# weight gradients 0.77 -> 1.3 - 1.95 ms, thin jobs 0.2 -> 0.73 ms per 196,608-point chunk, step 12.3 -> 12.8 ms at best and 17 - 35 ms
# with large chunks (profiles/r04_gstep_overlap.md: CU-scaling table, schedule sweep, kernel timeline).  tests/test_gpu_parity.py keeps
# the schedule exact (it must equal the serial one); tools/overlap_sweep.py reproduces the measurement.
OVERLAP_WGRAD = False
CHAIN_CUS_FRACTION = 0.75


FILM_KEYS = ("d_freq_geo", "d_phase_geo", "d_freq_app", "d_phase_app")


def _flat(r, skip=()):
    """the tensors of a siren_param_grads result in a fixed order (lists expanded), without the keys in `skip`"""
    out = []
    for k in sorted(r):
        if k in skip:
            continue
        out.extend(r[k] if isinstance(r[k], list) else [r[k]])
    return out


def _add_all(dst, src):
    if dst:
        torch._foreach_add_(dst, src)


def film_layer_weights(module, params):
    """(geometry FiLM-layer weights, colour FiLM-layer weights) of module._render_params(): what the 16-bit tape's frequency gradients need"""
    roles = module._roles(params)
    return [W for W, _ in roles["geo"]], [W for W, _ in roles["color"]]


class InputGrads(typing.NamedTuple):
    """what chunked_backward needs to also deliver the gradients wrt the SIREN's inputs (NativeModel.siren_input_grads)"""
    w_geo0: torch.Tensor                  # layer 0's nn.Linear weight [H, 3]
    w_color0: torch.Tensor                # colour layer 0's [H, 3 + G + H]
    d_points: typing.Optional[torch.Tensor]   # [nB, Pp, 3] to fill, or None
    d_dirs: typing.Optional[torch.Tensor]     # [nB, Pp, 3] to fill, or None
    only: bool = False                    # nothing else is wanted (with film_only): the weight- / FiLM-gradient launches are skipped


def chunked_backward(nat, nB, Pp, film, points, dirs, out, d_out, tape, tape_e, film_only, max_points=None, tape_format=0, weights=None,
                     input_grads=None, d_grid=None):
    """fenerf_siren_backward + fenerf_siren_param_grads over `nB` images of `Pp` points (a multiple of 32) in chunks of at most
    `max_points` points: tiles, tapes and outputs of an image range are contiguous, FiLM parameters are per image, and every gradient
    is a sum over points, so chunk results simply add.  A chunk is a run of WHOLE images while those fit (the curriculum's early
    stages: 6-12 images of 32x32x24 = 24,576 points per pass -- one launch of 8 images fills the 256 CUs, eight launches of 192
    workgroups do not); an image larger than `max_points` is split into point ranges of its own.  The gradient wrt the sampled grid
    features never exists as a tensor: every chunk scatters it into one channels-last gradient grid (fenerf_siren_backward_grid;
    inside the chain kernel for f16x3 models).
    tape_format / weights: the tape's format (_lib.TAPE_*) and, for the 16-bit tape, film_layer_weights(...).
    input_grads: None or an InputGrads: every chunk also fills its rows of the gradients wrt the sample positions / view directions
    from its d(theta) dump (NativeModel.siren_input_grads); with `only` set (and film_only) the weight- / FiLM-gradient launches are skipped.
    d_grid: None, or the channels-last gradient grid of an earlier call to add to (returned again).
    -> (grads dict like siren_param_grads with [nB]-leading FiLM gradients, d_grid_cl [D,H,W,32] or None)."""
    max_points = BACKWARD_CHUNK_POINTS if max_points is None else max_points
    max_points = max(128, max_points // 128 * 128)       # whole quads of 32-point tiles except in an image's last chunk
    LH = nat.tape_words_per_point(tape_format)          # fp32 words of tape per point
    G = nat.spec["grid_ch"]
    C = nat.C
    out, d_out = out.reshape(nB, Pp, C), d_out.reshape(nB, Pp, C)
    if d_grid is None:       # (given: the gradient grid an earlier call of the same backward pass started, scattered into further)
        d_grid = torch.zeros(tuple(nat.grid_shape) + (32,), dtype=torch.float32, device=out.device) if G and not film_only else None
    # (first image, images, first point, points) per launch
    if Pp <= max_points:
        per = max(1, max_points // Pp)
        chunks = [(b, min(per, nB - b), 0, Pp) for b in range(0, nB, per)]
    else:
        chunks = [(b, 1, s, min(max_points, Pp - s)) for b in range(nB) for s in range(0, Pp, max_points)]
    if film_only and nat.film_only_native() and input_grads is None:      # (the input gradients are read from the dump this route does not write)
        # Inversion on an f16x3 model: the chain writes only its per-tile FiLM sums (fenerf_siren_backward_film) -- no d(theta) dump.
        # A launch is bounded by the kernel's 32-bit tile arithmetic (2^24 points) and by FILM_SUMS_BUDGET_BYTES of FiLM sums (one
        # [L][2][H] block per 128 points, or per 16 points when an image is not a multiple of 128 points: 176 B / 1.4 KB per point at
        # H = 256); an image larger than that is walked in point ranges whose FiLM gradients add.
        sums_bytes_pp = 4.0 * nat.film_sums_floats(1, Pp) / Pp
        cap = int(max(128, min(1 << 24, FILM_SUMS_BUDGET_BYTES / sums_bytes_pp)) // 128 * 128)
        if Pp <= cap:
            per = max(1, cap // Pp)
            fchunks = [(b, min(per, nB - b), 0, Pp) for b in range(0, nB, per)]
        else:
            fchunks = [(b, 1, s, min(cap, Pp - s)) for b in range(nB) for s in range(0, Pp, cap)]
        rows, acc = {k: [] for k in FILM_KEYS}, None
        for b, nb, s, n in fchunks:
            film_c = tuple(t[b:b + nb] for t in film)
            g0 = b * Pp + s
            tape_c = tape[g0 * LH:(g0 + nb * n) * LH]
            sums = nat.siren_backward_film(nb, n, *film_c, out[b:b + nb, s:s + n], d_out[b:b + nb, s:s + n], tape_c)
            r = nat.siren_film_grads(nb, n, *film_c, sums)
            if s == 0:
                acc = [r[k] for k in FILM_KEYS]
                for k, t in zip(FILM_KEYS, acc):
                    rows[k].append(t)
            else:
                _add_all(acc, [r[k] for k in FILM_KEYS])
        return {k: (torch.cat(v, 0) if len(v) > 1 else v[0]) for k, v in rows.items()}, None
    # ---- chain + weight gradients per chunk.  With more than one chunk the weight gradients of chunk i run on a second stream BESIDE
    # the chain of chunk i + 1 (OVERLAP_WGRAD): the chain kernel is persistent, one workgroup per CU, and scales with the CUs it is
    # given; the weight-gradient kernels are HBM-bound and lose nothing on a quarter of the chip (profiles/r04_gstep_overlap.md) -- so the
    # chain is launched for CHAIN_CUS_FRACTION of the CUs and the weight-gradient grids are sized for the rest (fenerf_set_cu_budget).
    # The first chain and the last weight-gradient launch have the device to themselves.  Every d(theta) dump is kept alive until the
    # end of the loop and tied to the side stream (record_stream); scratch is per stream (NativeModel._workspace).
    dev = out.device
    inputs_only = bool(input_grads is not None and film_only and input_grads.only)
    overlap = OVERLAP_WGRAD and len(chunks) > 1 and not inputs_only
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev) if overlap else main
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    chain_cus = max(1, int(n_cus * CHAIN_CUS_FRACTION)) if overlap else 0
    wgrad_cus = max(1, n_cus - chain_cus) if overlap else 0
    if overlap:
        side.wait_stream(main)            # everything the weight gradients read (tape, outputs, upstream gradient) was produced on `main`
    total, film_rows = None, {k: [] for k in FILM_KEYS}
    acc_img = None           # FiLM gradients of the image whose point ranges are being walked
    keep = []
    for i, (b, nb, s, n) in enumerate(chunks):
        last = i == len(chunks) - 1
        film_c = tuple(t[b:b + nb] for t in film)
        g0 = b * Pp + s
        tape_c = tape[g0 * LH:(g0 + nb * n) * LH]
        out_c, d_out_c, pts_c = out[b:b + nb, s:s + n], d_out[b:b + nb, s:s + n], points[b:b + nb, s:s + n]
        with native.cu_budget(chain_cus if (overlap and i > 0) else 0):      # chain i runs beside the weight gradients of chunk i - 1
            if G and not film_only:
                d_t = nat.siren_backward_grid(nb, n, *film_c, out_c, d_out_c, tape_c, pts_c, d_grid, tape_format=tape_format)
            else:       # no grid, or inversion (only FiLM gradients wanted: nothing to scatter)
                d_t, _ = nat.siren_backward(nb, n, *film_c, out_c, d_out_c, tape_c, tape_format=tape_format)
        if input_grads is not None:       # a chunk is whole images or a point range of one image: its rows are contiguous
            dp, dd = input_grads.d_points, input_grads.d_dirs
            nat.siren_input_grads(pts_c, *film_c, d_t, input_grads.w_geo0, input_grads.w_color0, dp[b:b + nb, s:s + n] if dp is not None else None,
                                  dd[b:b + nb, s:s + n] if dd is not None else None)
        if overlap:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            d_t.record_stream(side)
            keep.append(d_t)
        if inputs_only:        # nothing but d points / d view directions was asked for: the dump has been read, no weight-gradient launch
            del d_t
            continue
        with torch.cuda.stream(side), native.cu_budget(wgrad_cus if (overlap and not last) else 0):
            r = nat.siren_param_grads(pts_c, dirs[b:b + nb, s:s + n] if dirs is not None else None, *film_c, out_c, d_out_c, tape_c,
                                      tape_e[g0:g0 + nb * n] if G else None, d_t, film_only=film_only, tape_format=tape_format, weights=weights)
            del d_t
            if total is None:
                total = {k: ([x for x in v] if isinstance(v, list) else v) for k, v in r.items() if k not in FILM_KEYS}
            else:
                _add_all(_flat(total), _flat(r, FILM_KEYS))      # one fused launch for all ~40 tensors
            if s == 0:
                acc_img = [r[k] for k in FILM_KEYS]
                for k, t in zip(FILM_KEYS, acc_img):
                    film_rows[k].append(t)
            else:
                _add_all(acc_img, [r[k] for k in FILM_KEYS])
            if last:
                for k, rows in film_rows.items():
                    total[k] = torch.cat(rows, 0) if len(rows) > 1 else rows[0]
    if inputs_only:
        return {k: None for k in FILM_KEYS}, None
    if overlap:
        main.wait_stream(side)
        for t in _flat(total):            # allocated on the side stream, consumed by autograd on `main`
            t.record_stream(main)
    return total, d_grid


_SIDE_STREAMS = {}


def _side_stream(dev):
    """one extra HIP stream per device for the overlapped weight gradients (created once)"""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def plan_chunks(nB, Pp, max_points=None):
    """(first image, images, first point, points) per backward launch -- the chunking of chunked_backward"""
    max_points = BACKWARD_CHUNK_POINTS if max_points is None else max_points
    max_points = max(128, max_points // 128 * 128)
    if Pp <= max_points:
        per = max(1, max_points // Pp)
        return [(b, min(per, nB - b), 0, Pp) for b in range(0, nB, per)]
    return [(b, 1, s, min(max_points, Pp - s)) for b in range(nB) for s in range(0, Pp, max_points)]


class GradSum:
    """The sum of per-chunk siren_param_grads results in chunk order, exactly as chunked_backward forms it: weight / bias gradients add,
    FiLM gradients are one row per (pass, image) -- written by the image's first point range, added to by its later ones."""

    def __init__(self):
        self.total, self.film_rows, self.acc_img = None, {k: [] for k in FILM_KEYS}, None

    def add(self, chunk, r):
        b, nb, s, n = chunk
        if self.total is None:
            self.total = {k: ([x for x in v] if isinstance(v, list) else v) for k, v in r.items() if k not in FILM_KEYS}
        else:
            _add_all(_flat(self.total), _flat(r, FILM_KEYS))
        if s == 0:
            self.acc_img = [r[k] for k in FILM_KEYS]
            for k, t in zip(FILM_KEYS, self.acc_img):
                self.film_rows[k].append(t)
        else:
            _add_all(self.acc_img, [r[k] for k in FILM_KEYS])

    def result(self):
        total = dict(self.total)
        for k, rows in self.film_rows.items():
            total[k] = torch.cat(rows, 0) if len(rows) > 1 else rows[0]
        return total


def run_chains(nat, nB, Pp, film, points, out, d_out, tape, chunks, tape_format=0, d_grid=None):
    """First half of a SPLIT backward (generators/autograd.py HierarchicalRenderSplitFunction): the chain launches of `chunks`, nothing else.
    -> ([d(theta) dump per chunk], d_grid_cl or None).  The dumps stay alive until run_weight_grads has consumed them (each as large as its
    chunk's tape: the price of handing the grid gradient -- final once the last chain has run -- to autograd / DistributedDataParallel
    BEFORE the weight-gradient kernels, so that its all-reduce runs beside them).  d_grid: the channels-last gradient grid to scatter into
    (a caller that runs some chunks' chains elsewhere passes the same one to every call); None = a fresh zeroed one."""
    LH = nat.tape_words_per_point(tape_format)
    G, C = nat.spec["grid_ch"], nat.C
    out, d_out = out.reshape(nB, Pp, C), d_out.reshape(nB, Pp, C)
    if d_grid is None and G:
        d_grid = torch.zeros(tuple(nat.grid_shape) + (32,), dtype=torch.float32, device=out.device)
    dumps = []
    for b, nb, s, n in chunks:
        film_c = tuple(t[b:b + nb] for t in film)
        g0 = b * Pp + s
        tape_c = tape[g0 * LH:(g0 + nb * n) * LH]
        if G:
            dumps.append(nat.siren_backward_grid(nb, n, *film_c, out[b:b + nb, s:s + n], d_out[b:b + nb, s:s + n], tape_c, points[b:b + nb, s:s + n], d_grid,
                                                 tape_format=tape_format))
        else:
            dumps.append(nat.siren_backward(nb, n, *film_c, out[b:b + nb, s:s + n], d_out[b:b + nb, s:s + n], tape_c, tape_format=tape_format)[0])
    return dumps, d_grid


def run_weight_grads(nat, nB, Pp, film, points, dirs, out, d_out, tape, tape_e, chunks, dumps, tape_format=0, weights=None, acc=None, finish=True):
    """Second half of a split backward: the weight-gradient launches of `chunks` over the dumps run_chains left, summed like
    chunked_backward does (acc: a GradSum that already holds earlier chunks' gradients).  Frees each dump after its chunk.
    -> grads dict (FiLM gradients with [nB] leading), or the GradSum itself when not `finish`."""
    LH = nat.tape_words_per_point(tape_format)
    G, C = nat.spec["grid_ch"], nat.C
    out, d_out = out.reshape(nB, Pp, C), d_out.reshape(nB, Pp, C)
    acc = GradSum() if acc is None else acc
    for i, (b, nb, s, n) in enumerate(chunks):
        film_c = tuple(t[b:b + nb] for t in film)
        g0 = b * Pp + s
        tape_c = tape[g0 * LH:(g0 + nb * n) * LH]
        d_t, dumps[i] = dumps[i], None
        r = nat.siren_param_grads(points[b:b + nb, s:s + n], dirs[b:b + nb, s:s + n] if dirs is not None else None, *film_c,
                                  out[b:b + nb, s:s + n], d_out[b:b + nb, s:s + n], tape_c, tape_e[g0:g0 + nb * n] if G else None, d_t,
                                  tape_format=tape_format, weights=weights)
        del d_t
        acc.add((b, nb, s, n), r)
    return acc.result() if finish else acc


def check_same_weights(ctx, nat):
    """The backward kernels read the model's resident backward stream, not a snapshot: if the model was re-packed between
    forward and backward (an optimizer step and a new render of the same module before .backward()), refuse instead of
    differentiating with the wrong weights."""
    if nat.pack_generation != ctx.pack_generation:
        raise RuntimeError("fenerf_amd: the generator's weights were re-packed between this render's forward and its backward "
                           "(parameters changed and the module rendered again before .backward()); call backward before the "
                           "next optimizer step + render")


class SirenFunction(torch.autograd.Function):
    """out = siren(points, dirs; film params, weights).  Non-tensor arg `module` supplies the native model and roles."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # under autocast (the reference's training loop) inputs arrive as fp16
    def forward(ctx, module, points, dirs, fg, pg, fa, pa, *params):
        nat = module.native_differentiable(points.device)
        # the tape's format is fixed at forward time: a backward that takes weight gradients may use the 16-bit tape (siren.grad_precision =
        # "tape16"), a FiLM-only one (inversion) needs the accumulators of the fp32 tape
        ctx.tape_format = module.tape_format(nat, film_only=not any(ctx.needs_input_grad[7:]))
        out, tape, tape_e = nat.siren_forward_save(points, dirs, fg, pg, fa, pa, tape_format=ctx.tape_format)
        ctx.module, ctx.nat = module, nat
        ctx.pack_generation = nat.pack_generation
        ctx.has_dirs = dirs is not None
        ctx.save_for_backward(points, dirs if dirs is not None else points.new_empty(0), fg, pg, fa, pa, out, tape,
                              tape_e if tape_e is not None else points.new_empty(0), *params)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_out):
        module, nat = ctx.module, ctx.nat
        check_same_weights(ctx, nat)
        points, dirs, fg, pg, fa, pa, out, tape, tape_e, *params = ctx.saved_tensors
        roles = module._roles(params)
        spec = nat.spec
        ng, C = spec["n_geo"], spec["output_dim"]
        n_lab = C - 4
        B, P = points.shape[0], points.shape[1]
        d_out = d_out.contiguous().float()
        need = ctx.needs_input_grad
        film_only = not any(need[7:])        # inversion: only the FiLM parameters are optimised
        # gradients wrt the sample positions / view directions (callers of the bare module; the generators build their rays under no_grad)
        d_points = torch.empty_like(points) if need[1] else None
        d_dirs = torch.empty_like(dirs) if (need[2] and ctx.has_dirs) else None
        input_grads = None
        if d_points is not None or d_dirs is not None:
            w_geo, w_col = film_layer_weights(module, params)
            input_grads = InputGrads(w_geo[0], w_col[0], d_points, d_dirs, only=film_only and not any(need[3:7]))
        r, d_grid = chunked_backward(nat, B, P, (fg, pg, fa, pa), points, dirs if ctx.has_dirs else None, out, d_out, tape,
                                     tape_e if tape_e.numel() else None, film_only, tape_format=ctx.tape_format,
                                     weights=film_layer_weights(module, params) if ctx.tape_format else None, input_grads=input_grads)
        film_grads = (r["d_freq_geo"] if need[3] else None, r["d_phase_geo"] if need[4] else None,
                      r["d_freq_app"] if need[5] else None, r["d_phase_app"] if need[6] else None)
        if film_only:
            return (None, d_points, d_dirs) + film_grads + (None,) * len(params)
        return (None, d_points, d_dirs) + film_grads + assemble_param_grads(module, nat, params, r, points, d_grid, need[7:])


class PointwiseSirenFunction(torch.autograd.Function):
    """out = siren(points, dirs; per-point film params [B,P,n*H], weights): SPATIALSIRENGRID.forward_with_frequencies_phase_shifts
    (siren.py:464-477) under autograd on the native kernels (round 6; include/fenerf.h fenerf_siren_*_pointwise).  Gradients wrt the
    per-point frequencies / phase shifts (for the per-point mapping network's own backward) and every SIREN weight and bias."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, points, dirs, fg, pg, fa, pa, *params):
        nat = module.native_pointwise_differentiable(points.device)
        out, tape = nat.siren_forward_save_pointwise(points, dirs, fg, pg, fa, pa)
        ctx.module, ctx.nat = module, nat
        ctx.pack_generation = nat.pack_generation
        ctx.has_dirs = dirs is not None
        ctx.save_for_backward(points, dirs if dirs is not None else points.new_empty(0), fg, pg, fa, pa, out, tape, *params)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_out):
        module, nat = ctx.module, ctx.nat
        check_same_weights(ctx, nat)
        points, dirs, fg, pg, fa, pa, out, tape, *params = ctx.saved_tensors
        need = ctx.needs_input_grad
        r = nat.siren_backward_pointwise(points, dirs if ctx.has_dirs else None, fg, pg, fa, pa, out, d_out.contiguous().float(), tape)
        film_grads = (r["d_freq_geo"] if need[3] else None, r["d_phase_geo"] if need[4] else None,
                      r["d_freq_app"] if need[5] else None, r["d_phase_app"] if need[6] else None)
        return (None, None, None) + film_grads + assemble_param_grads(module, nat, params, r, points, None, need[7:])


def siren_apply_pointwise(module, points, dirs, fg, pg, fa, pa):
    """Differentiable SIREN evaluation with per-point FiLM tensors [B,P,n*H]; pads to whole 32-point tiles like siren_apply."""
    if points.requires_grad or (dirs is not None and dirs.requires_grad):
        raise NotImplementedError("fenerf_amd: gradients wrt sample positions / view directions are not provided "
                                  "(the reference's training and inversion loops do not use them)")
    params = module._render_params()
    P = points.shape[1]
    pad = (-P) % 32
    if pad:
        rep = lambda t: torch.cat([t, t[:, -1:].expand(-1, pad, -1)], 1)
        points, fg, pg, fa, pa = rep(points), rep(fg), rep(pg), rep(fa), rep(pa)
        dirs = rep(dirs) if dirs is not None else None
    out = PointwiseSirenFunction.apply(module, points.contiguous(), dirs.contiguous() if dirs is not None else None, fg.contiguous(), pg.contiguous(),
                                       fa.contiguous(), pa.contiguous(), *params)
    return out[:, :P] if pad else out


def siren_apply(module, points, dirs, fg, pg, fa, pa):
    """Differentiable SIREN evaluation.  The native path works on whole 32-point tiles per image: other point counts are
    padded here (with the last point; the pads get no gradient because their outputs are sliced away).
    points / dirs that require grad get theirs too (fenerf_siren_input_grads: an extra pass over the fp32 d(theta) dump)."""
    if (points.requires_grad or (dirs is not None and dirs.requires_grad)) and module.grad_precision in ("amp", "amp16") \
            and module.precision != "f32":
        raise NotImplementedError("fenerf_amd: gradients wrt sample positions / view directions read the fp32 d(theta) dump; "
                                  "grad_precision 'amp' / 'amp16' writes it as bf16 -- use 'f32' or 'tape16'")
    params = module._render_params()
    P = points.shape[1]
    pad = (-P) % 32
    if pad:
        points = torch.cat([points, points[:, -1:].expand(-1, pad, -1)], 1)
        if dirs is not None:
            dirs = torch.cat([dirs, dirs[:, -1:].expand(-1, pad, -1)], 1)
    out = SirenFunction.apply(module, points.contiguous(), dirs.contiguous() if dirs is not None else None, fg, pg, fa, pa, *params)
    return out[:, :P] if pad else out
