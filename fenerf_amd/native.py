"""Torch-side plumbing over the C-ABI: device pointers, streams, workspaces.  No math lives here."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expected a contiguous fp32 device tensor"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def to_host(t):
    """t.cpu() through page-locked memory (torch's caching host allocator): the reference's staged_forward ends in `.cpu()` of a 5.8-MB
    pixel tensor (generators.py:640-646); into pageable memory that copy costs 0.2 ms or 45 ms depending on whether the destination pages
    were touched before (every other call at 256 x 256: 70 instead of 26 ms per image, tools/exp/staged_calls_profile.py) -- into pinned
    memory it is 0.2 ms every time.  Returns an ordinary CPU tensor (pinned), complete when the call returns."""
    if not t.is_cuda:
        return t.cpu()
    t = t.detach()
    out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    out.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return out



# Hidden widths the kernels are instantiated for (include/fenerf.h).  Any other width up to the largest is run at the next instantiated one
# with zero padding (round 6): padded rows / columns of every weight matrix, padded biases and padded FiLM phase shifts are zero, so a padded
# feature's activation is sin(f' * 0 + 0) = 0 and nothing it touches changes a sum (adding exact zeros) -- outputs and the gradients of the
# real entries are those of the unpadded network; gradients of padded entries are dropped.  The reference constructs any width
# (siren/siren.py:1451).
SUPPORTED_HIDDEN = (32, 64, 96, 128, 192, 256)


def padded_hidden_dim(H):
    for w in SUPPORTED_HIDDEN:
        if H <= w:
            return w
    raise ValueError(f"hidden_dim {H}: the kernels are instantiated up to {SUPPORTED_HIDDEN[-1]} (include/fenerf.h)")


def _pad_axes(name, n_label_layers):
    """which axes of render parameter `name` are hidden-width axes: (pad rows?, pad the LAST H columns?) -- the colour layer 0's input is
    [dirs | grid features | x]: its hidden part is the trailing H columns; the LAST label layer, final_layer and color_layer_linear map
    hidden features to outputs (columns only, biases untouched)"""
    last_label_weight = f"label_layer_linear.{n_label_layers - 1}.weight" if n_label_layers else None
    last_label_bias = f"label_layer_linear.{n_label_layers - 1}.bias" if n_label_layers else None
    if name == "spatial_embeddings":
        return False, False
    if name.endswith(".bias"):
        return (not (name.startswith("final_layer") or name.startswith("color_layer_linear") or name == last_label_bias)), False
    if name == "network.0.layer.weight":
        return True, False
    if name.startswith("final_layer") or name.startswith("color_layer_linear") or name == last_label_weight:
        return False, True
    return True, True


def _pad_param(name, t, H, Hp, n_label_layers, is_numpy):
    """zero-pad one reference-named render parameter from hidden width H to Hp (numpy array or torch tensor)"""
    rows, cols = _pad_axes(name, n_label_layers)
    d = Hp - H
    if t.ndim == 1:
        if not rows:
            return t
        return np.concatenate([t, np.zeros(d, t.dtype)]) if is_numpy else torch.nn.functional.pad(t, (0, d))
    if cols:
        t = np.concatenate([t, np.zeros((t.shape[0], d), t.dtype)], 1) if is_numpy else torch.nn.functional.pad(t, (0, d))
    if rows:
        t = np.concatenate([t, np.zeros((d, t.shape[1]), t.dtype)], 0) if is_numpy else torch.nn.functional.pad(t, (0, 0, 0, d))
    return t


class NativeModel:
    """Owns a FenerfModel* built from a reference-named state dict (numpy fp32 arrays)."""

    # precision strings that are a FENERF_PREC_F16X3 model with a reduced-precision NO-GRAD forward (include/fenerf.h
    # fenerf_model_set_forward_mode; round 5, opt-in): two fp16 MFMAs per product everywhere / in the colour layers and heads only
    FORWARD_MODES = {"f16x2": 1, "f16x3c2": 2}

    def __init__(self, sd, spec, device, precision="f32", differentiable=False, wgrad_bf16_min_points=0):
        self.logical_H = int(spec["hidden_dim"])                       # the module's width; self.spec carries the (padded) width the kernels run at
        spec = dict(spec, hidden_dim=padded_hidden_dim(self.logical_H))
        self.spec = dict(spec)
        sd = self._pad_state(sd)
        self.forward_mode = self.FORWARD_MODES.get(precision, 0)
        self.requested_precision = precision
        precision = "f16x3" if self.forward_mode else precision
        self.precision = precision
        self.differentiable = bool(differentiable)
        # > 0: AMP-class weight gradients (bf16 operands, include/fenerf.h FenerfModelDesc.wgrad_bf16_min_points); 0 = fp32 class
        self.wgrad_bf16_min_points = int(wgrad_bf16_min_points) if precision == "f16x3" else 0
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("fenerf_amd renders on the GPU only (there is no CPU path); got device %s" % device)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            d, keep = _lib.make_desc(sd, spec, precision, differentiable, self.wgrad_bf16_min_points)
            _lib.check(_lib.lib().fenerf_model_create(C.byref(d), C.byref(self._h)))
            if self.forward_mode:
                if _lib.lib().fenerf_model_set_forward_mode(self._h, self.forward_mode) < 0:
                    raise _lib.FenerfError(-1, _lib.lib().fenerf_last_error().decode())
        self.C = spec["output_dim"]
        self.grid_shape = tuple(int(v) for v in sd["spatial_embeddings"].shape[2:]) if spec.get("grid_ch") else None   # (D, H, W)
        self.box_scale = 2 / 0.24       # UniformBoxWarp(0.24), siren.py:181-187 (what _lib.make_desc sets)
        self._ws = {}
        self.pack_generation = 0      # bumped by every re-pack: autograd nodes check that forward and backward saw the same weights

    # ---- hidden widths between the instantiated ones: zero padding on the way in, slicing on the way out (padded_hidden_dim above)
    @property
    def padded(self):
        return getattr(self, "logical_H", self.spec["hidden_dim"]) != self.spec["hidden_dim"]

    def _pad_state(self, sd):
        if not self.padded:
            return sd
        H, Hp, nl = self.logical_H, self.spec["hidden_dim"], self.spec.get("n_label_layers", 0)
        return {k: _pad_param(k, np.asarray(v), H, Hp, nl, True) for k, v in sd.items()}

    def _pad_film(self, t, n):
        """[..., n * H] -> [..., n * Hp], zeros behind every layer's block"""
        H, Hp = self.logical_H, self.spec["hidden_dim"]
        if t.shape[-1] == n * Hp:
            return t
        return torch.nn.functional.pad(t.reshape(*t.shape[:-1], n, H), (0, Hp - H)).reshape(*t.shape[:-1], n * Hp)

    def _unpad_grads(self, res):
        """a gradient dict of siren_param_grads' layout at the padded width -> the module's width (views; the label head's rows stay)"""
        if not self.padded:
            return res
        H, Hp = self.logical_H, self.spec["hidden_dim"]
        ng, nc, G = self.spec["n_geo"], self.spec["n_color"], self.spec["grid_ch"]
        out = dict(res)
        for k, n in (("d_freq_geo", ng), ("d_phase_geo", ng), ("d_freq_app", nc), ("d_phase_app", nc)):
            if k in res:
                t = res[k]
                out[k] = t.reshape(*t.shape[:-1], n, Hp)[..., :H].reshape(*t.shape[:-1], n * H).contiguous()
        if "geo_w" in res:
            out["geo_w"] = [res["geo_w"][0][:H].contiguous()] + [w[:H, :H].contiguous() for w in res["geo_w"][1:]]
            out["color_w"] = [res["color_w"][0][:H, :3 + G + H].contiguous()] + [w[:H, :H].contiguous() for w in res["color_w"][1:]]
            out["geo_b"] = [b[:H].contiguous() for b in res["geo_b"]]
            out["color_b"] = [b[:H].contiguous() for b in res["color_b"]]
            out["head_w"] = res["head_w"][:, :H].contiguous()
            out["rgb_w"] = res["rgb_w"][:, :H].contiguous()
        return out

    def _pad_weights(self, weights):
        """film_layer_weights(...) at the module's width -> at the kernels' width (device copies)"""
        if weights is None or not self.padded:
            return weights
        H, Hp, G = self.logical_H, self.spec["hidden_dim"], self.spec["grid_ch"]
        pad2 = lambda w, cols: torch.nn.functional.pad(w.detach().float(), (0, cols - w.shape[1], 0, Hp - H))
        geo = [pad2(weights[0][0], 3)] + [pad2(w, Hp) for w in weights[0][1:]]
        col = [pad2(weights[1][0], 3 + G + Hp)] + [pad2(w, Hp) for w in weights[1][1:]]
        return geo, col

    def update(self, sd):
        sd = self._pad_state(sd)
        self.pack_generation += 1
        self._resident = None         # what a later load_from_device(maybe_unchanged=True) compares against is no longer what is resident
        with torch.cuda.device(self.device):
            d, keep = _lib.make_desc(sd, self.spec, self.precision, self.differentiable, self.wgrad_bf16_min_points)
            _lib.check(_lib.lib().fenerf_model_update(self._h, C.byref(d), _stream()))

    # ---- device-side packing (training): the fp32 streams are permutations of the parameters -------------------------
    def _canonical(self):
        """[(reference name, shape)] of the render parameters in the flat order of the index maps; the label head is its fold."""
        sp = self.spec
        H, ng, nc, G, n_lab = sp["hidden_dim"], sp["n_geo"], sp["n_color"], sp["grid_ch"], sp["output_dim"] - 4
        items = [("network.0.layer.weight", (H, 3)), ("network.0.layer.bias", (H,))]
        for i in range(1, ng):
            items += [(f"network.{i}.layer.weight", (H, H)), (f"network.{i}.layer.bias", (H,))]
        cname = (lambda i: "color_layer_sine.layer") if sp["kind"] == "spatial" else (lambda i: f"color_layer_sine.{i}.layer")
        items += [(cname(0) + ".weight", (H, 3 + G + H)), (cname(0) + ".bias", (H,))]
        for i in range(1, nc):
            items += [(cname(i) + ".weight", (H, H)), (cname(i) + ".bias", (H,))]
        if n_lab > 0:
            items += [("label_layer_linear.0.weight", (n_lab, H)), ("label_layer_linear.0.bias", (n_lab,))]
        items += [("final_layer.weight", (1, H)), ("final_layer.bias", (1,)), ("color_layer_linear.0.weight", (3, H)),
                  ("color_layer_linear.0.bias", (3,))]
        return items

    def _index_maps(self):
        """Packs index-valued weights once on the host: every stream element is then (1 + flat index) of its source
        parameter element, 0 for padding -- exact in fp32 below 2^24 elements.  -> dict of device index tensors."""
        if getattr(self, "_maps", None) is None:
            items = self._canonical()
            total = sum(int(np.prod(sh)) for _, sh in items)
            if total + 1 >= 1 << 24:
                raise RuntimeError("too many render parameters for fp32 index tagging")
            tag, off = {}, 1
            for name, sh in items:
                n = int(np.prod(sh))
                tag[name] = np.arange(off, off + n, dtype=np.float32).reshape(sh)
                off += n
            spec1 = dict(self.spec, n_label_layers=1 if self.spec["output_dim"] > 4 else 0)
            if self.spec["grid_ch"]:
                tag["spatial_embeddings"] = np.zeros((1, 32, 2, 2, 2), np.float32)
            to_idx = lambda a: torch.from_numpy(a.astype(np.int64)).to(self.device)
            blob, consts = _lib.pack_weights_host(tag, spec1, "f32")
            maps = dict(stream=to_idx(blob), consts=to_idx(consts))
            if self.differentiable:
                maps["bwd"] = to_idx(_lib.pack_backward_host(tag, spec1))
                if self.precision == "f16x3":     # bf16 (hi, lo) ring behind the fp32 rgb-head block; hi / lo alternate per entry
                    idx = _lib.pack_backward_index_map_bf16(tag, spec1).astype(np.int64)
                    maps["bwd_head"] = maps["bwd"][: (self.spec["hidden_dim"] // 32) * 256]
                    maps["bwd16_idx"] = to_idx(idx)
                    maps["bwd16_lo"] = to_idx((np.arange(idx.size) // 512) & 1).bool()
            if self.precision == "f16x3":
                codes = _lib.pack_index_map_f16(tag, spec1).astype(np.int64)
                maps["l0"] = maps["stream"][: (self.spec["hidden_dim"] // 32) * 256]      # fp32 layer-0 block, as in the f32 stream
                maps["h_idx"], maps["h_lo"] = to_idx(codes & 0x3FFFFFFF), to_idx(codes >> 30).bool()
            self._maps = maps
        return self._maps

    @staticmethod
    def _row_scale(W):
        """Power of two s with max|row| * s in [0.5, 1) (1 for an all-zero row) -- row_scales() of fenerf_pack.cpp."""
        m = W.abs().amax(1)
        _, ex = torch.frexp(m)
        return torch.where(m > 0, torch.ldexp(torch.ones_like(m), -ex), torch.ones_like(m))

    def _scaled_rows(self):
        """(film_w names, head_w names) -- the matrices whose rows get a power-of-two scale at f16x3 (row_scales() of fenerf_pack.cpp)"""
        sp = self.spec
        ng, nc, n_lab = sp["n_geo"], sp["n_color"], sp["output_dim"] - 4
        cname = (lambda i: "color_layer_sine.layer") if sp["kind"] == "spatial" else (lambda i: f"color_layer_sine.{i}.layer")
        film_w = [f"network.{i}.layer.weight" for i in range(1, ng)] + [cname(i) + ".weight" for i in range(nc)]
        head_w = (["label_layer_linear.0.weight"] if n_lab > 0 else []) + ["final_layer.weight", "color_layer_linear.0.weight"]
        return film_w, head_w

    def _repack_maps(self):
        """FenerfRepackMaps for fenerf_model_repack, built once: the index maps as int32 device arrays, plus (f16x3) the table of
        scaled rows, the per-element scale ids and the layout of the result scales behind the fp32 consts."""
        if getattr(self, "_rmaps", None) is None:
            maps = self._index_maps()
            sp = self.spec
            H, L, n_lab = sp["hidden_dim"], sp["n_geo"] + sp["n_color"], sp["output_dim"] - 4
            keep, r = {}, _lib.FenerfRepackMaps()

            def put(field, t, nfield=None):
                t = (t if torch.is_tensor(t) else torch.from_numpy(np.asarray(t))).to(self.device, torch.int32).contiguous()
                keep[field] = t
                setattr(r, field, t.data_ptr())
                if nfield:
                    setattr(r, nfield, t.numel())

            if self.precision == "f32":
                put("stream_f32", maps["stream"], "n_stream_f32")
                put("consts", maps["consts"], "n_consts")
                if self.differentiable:
                    put("bwd_f32", maps["bwd"], "n_bwd_f32")
            else:
                put("stream_f32", maps["l0"], "n_stream_f32")
                put("stream_h16", maps["h_idx"] | (maps["h_lo"].long() << 30), "n_stream_h16")
                put("consts", maps["consts"], "n_consts")
                if self.differentiable:
                    put("bwd_f32", maps["bwd_head"], "n_bwd_f32")
                    put("bwd_b16", maps["bwd16_idx"], "n_bwd_b16")
                film_w, head_w = self._scaled_rows()
                off, start = 1, {}
                for name, sh in self._canonical():
                    start[name] = (off, sh)
                    off += int(np.prod(sh))
                row_off, row_len, row_film, row0 = [], [], [], {}
                scale_id = np.zeros(off, np.int32)
                for name in film_w + head_w:
                    o, (rows, cols) = start[name]
                    row0[name] = len(row_off)
                    for q in range(rows):
                        scale_id[o + q * cols: o + (q + 1) * cols] = 1 + len(row_off)
                        row_off.append(o + q * cols); row_len.append(cols); row_film.append(1 if name in film_w else 0)
                put("row_off", row_off); put("row_len", row_len); put("row_film", row_film); put("scale_id", scale_id)
                r.n_rows = len(row_off)
                # result scales behind the fp32 consts: [L][H] per FiLM layer (layer 0 unscaled), head [32], rgb [4]
                tail = [-1] * H
                for name in film_w:
                    tail += [1 + row0[name] + n for n in range(H)]
                head = [0] * 32
                if n_lab > 0:
                    head[:n_lab] = [1 + row0["label_layer_linear.0.weight"] + q for q in range(n_lab)]
                head[n_lab] = 1 + row0["final_layer.weight"]
                tail += head + [1 + row0["color_layer_linear.0.weight"] + c for c in range(3)] + [-1]
                put("consts_tail", tail, "n_tail")
            self._rmaps = (r, keep)
        return self._rmaps[0]

    def _flat_params(self, params):
        """({name: fp32 device tensor} with the label head folded, the canonical flat vector behind a leading 0)"""
        sp = self.spec
        n_lab = sp["output_dim"] - 4
        dev = self.device
        p = {k: v.detach().to(dev, torch.float32) for k, v in params.items()}
        # autocast off: the reference's D-step / eval renders run under torch.cuda.amp.autocast (train_double_latent_semantic.py:
        # 279-290, 466-512); an fp16 fold of ~0.006-magnitude label weights lands in the subnormal range and would differ from
        # the fold the differentiable path does with autocast off (custom_fwd)
        with torch.autocast(dev.type, enabled=False):
            if n_lab > 0:       # fold the activation-free label head (siren.py:1490-1494) on the device
                nl = sp["n_label_layers"]
                c = p["label_layer_linear.0.bias"]
                for i in range(1, nl):
                    c = p[f"label_layer_linear.{i}.weight"] @ c + p[f"label_layer_linear.{i}.bias"]
                A = p[f"label_layer_linear.{nl - 1}.weight"]          # from the output side: n_lab-row products, not H x H x H
                for i in range(nl - 2, -1, -1):
                    A = A @ p[f"label_layer_linear.{i}.weight"]
                p["label_layer_linear.0.weight"], p["label_layer_linear.0.bias"] = A, c
            if self.padded:       # folded label head first (at the module's width), then every canonical item to the kernels' width
                H, Hp = self.logical_H, sp["hidden_dim"]
                for name, shape in self._canonical():
                    if tuple(p[name].shape) != tuple(shape):
                        p[name] = _pad_param(name, p[name], H, Hp, 1 if n_lab > 0 else 0, False)
            flat = torch.cat([torch.zeros(1, device=dev)] + [p[name].reshape(-1) for name, _ in self._canonical()])
        return p, flat

    @staticmethod
    def _grid_checksum(grid):
        """Digest of the grid's fp32 bit patterns: wrap-around int32 sums of runs of (up to) 1,024 consecutive words -- 27,648 sums for the
        96^3 grid.  One pass over 113 MB (~27 us), no temporaries, no sync, deterministic; any single changed word changes it, and edits
        that compensate each other (permuted or offsetting values written through .data) collide only inside one 4-KiB run instead of
        anywhere in the tensor.  (Accumulated in int64 a sum costs 129 us -- torch converts the tensor first -- which was 1 % of every
        generator step: tools/exp/checksum_probe.py.)"""
        words = grid.reshape(-1).view(torch.int32)
        run = 1024
        while words.numel() % run:
            run //= 2
        return words.view(-1, run).sum(1, dtype=torch.int32)

    def load_from_device(self, params, maybe_unchanged=False):
        """Re-pack from device-resident parameters {reference name: tensor} without touching the host: one concatenation,
        then fenerf_model_repack gathers / scales / splits straight into the model's resident streams.
        `maybe_unchanged`: the caller was told to re-pack without seeing a parameter version change (train() / eval() switch,
        invalidate_native()).  A differentiable model then compares the content with what is resident (the flat MLP vector and a
        checksum of the grid; one host sync) and skips an identical re-pack WITHOUT advancing pack_generation, so that autograd
        nodes between their forward and backward stay valid across a mode switch."""
        p, flat = self._flat_params(params)
        grid = p.get("spatial_embeddings")
        grid = grid.contiguous() if grid is not None else None
        if self.differentiable:
            last = getattr(self, "_resident", None)
            csum = self._grid_checksum(grid) if grid is not None else None
            if maybe_unchanged and last is not None and last[0].shape == flat.shape and torch.equal(last[0], flat) and \
                    (csum is None or (last[1] is not None and last[1].shape == csum.shape and torch.equal(last[1], csum))):
                return
            self._resident = None     # until the re-pack below has succeeded nothing is known to be resident
        r = self._repack_maps()
        self.pack_generation += 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().fenerf_model_repack(self._h, _ptr(flat), flat.numel(), C.byref(r), _ptr(grid), _stream()))
        if self.differentiable:
            self._resident = (flat, csum)

    def export_packed(self):
        """(stream, consts, backward stream or None) copies of the resident packed buffers (tests)."""
        n = lambda key: self._index_maps()[key].numel()
        if self.precision == "f32":
            ns, nc, nb = n("stream"), n("consts"), (n("bwd") if self.differentiable else 0)
        else:
            H, L = self.spec["hidden_dim"], self.spec["n_geo"] + self.spec["n_color"]
            ns, nc = n("l0") + n("h_idx") // 2, n("consts") + L * H + 36
            nb = n("bwd_head") + n("bwd16_idx") // 2 if self.differentiable else 0
        new = lambda k: torch.empty(k, dtype=torch.float32, device=self.device)
        stream, consts, bwd = new(ns), new(nc), (new(nb) if nb else None)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().fenerf_model_export_packed(self._h, _ptr(stream), ns, _ptr(consts), nc, _ptr(bwd), nb, _stream()))
        return stream, consts, bwd

    def _pack_on_device(self, params):
        """(stream, consts, backward stream or None, grid or None) as torch tensors on self.device: the re-pack of
        fenerf_model_repack spelled out in torch ops (gathers through the index maps; f16x3: after scaling rows and splitting
        into fp16 / bf16 hi, lo).  Not on the product path: the layout tests run it on the CPU against the host packer and on
        the GPU against the native re-pack, bit for bit."""
        maps = self._index_maps()
        sp = self.spec
        H, ng, nc, n_lab = sp["hidden_dim"], sp["n_geo"], sp["n_color"], sp["output_dim"] - 4
        dev = self.device
        p, flat = self._flat_params(params)
        items = self._canonical()
        zero = torch.zeros(1, device=dev)
        bwd = None
        if self.precision == "f32":
            stream, consts = flat[maps["stream"]], flat[maps["consts"]]
            if self.differentiable:
                bwd = flat[maps["bwd"]]
        else:
            film_w, head_w = self._scaled_rows()
            sc = {k: self._row_scale(p[k]) for k in film_w + head_w}
            scaled = lambda name, extra=1.0: (p[name] * (sc[name] * extra)[:, None]) if name in sc else p[name]
            flat_s = torch.cat([zero] + [scaled(name).reshape(-1) for name, _ in items])
            hi = flat_s.to(torch.float16)
            lo = (flat_s - hi.float()).to(torch.float16)
            halves = torch.where(maps["h_lo"], lo[maps["h_idx"]], hi[maps["h_idx"]])
            stream = torch.cat([flat[maps["l0"]], halves.view(torch.float32)])
            inv = [torch.ones(H, device=dev)] + [1.0 / (sc[k] * 16.0) for k in film_w]                       # [L][H], layer 0 unscaled
            head_inv = torch.full((32,), 1.0 / 16.0, device=dev)
            if n_lab > 0:
                head_inv[:n_lab] = 1.0 / (sc["label_layer_linear.0.weight"] * 16.0)
            head_inv[n_lab] = 1.0 / (sc["final_layer.weight"][0] * 16.0)
            rgb_inv = torch.cat([1.0 / (sc["color_layer_linear.0.weight"] * 16.0), torch.ones(1, device=dev)])
            consts = torch.cat([flat[maps["consts"]]] + inv + [head_inv, rgb_inv])
            if self.differentiable:     # backward stream: FiLM-layer rows scaled like the forward's (x 16), heads true; bf16 hi / lo
                flat_b = torch.cat([zero] + [(scaled(name, 16.0) if name in film_w else p[name]).reshape(-1) for name, _ in items])
                bhi = flat_b.to(torch.bfloat16)
                blo = (flat_b - bhi.float()).to(torch.bfloat16)
                bhalves = torch.where(maps["bwd16_lo"], blo[maps["bwd16_idx"]], bhi[maps["bwd16_idx"]])
                bwd = torch.cat([flat_b[maps["bwd_head"]], bhalves.view(torch.float32)])
        grid = p.get("spatial_embeddings")
        grid = grid.contiguous() if grid is not None else None
        return stream.contiguous(), consts.contiguous(), bwd, grid

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().fenerf_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _workspace(self, key, nbytes):
        """Persistent scratch per (purpose, HIP stream): calls issued on different streams (the overlapped backward of
        siren/autograd.py, two host threads) must not share FiLM / weight-gradient scratch."""
        key = (key, torch.cuda.current_stream(self.device).cuda_stream)
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = buf
        return buf

    def _film(self, B, fg, pg, fa, pa):
        H, ng, nc = self.spec["hidden_dim"], self.spec["n_geo"], self.spec["n_color"]
        dev = self.device
        fg, pg, fa, pa = (_f32(t, dev) for t in (fg, pg, fa, pa))
        if self.padded:
            fg, pg, fa, pa = self._pad_film(fg, ng), self._pad_film(pg, ng), self._pad_film(fa, nc), self._pad_film(pa, nc)
        for t, n in ((fg, ng), (pg, ng), (fa, nc), (pa, nc)):
            if tuple(t.shape) != (B, n * H):
                raise ValueError(f"film parameter of shape {tuple(t.shape)}, expected {(B, n * self.logical_H)}")
        return fg, pg, fa, pa

    def siren_forward(self, points, ray_dirs, fg, pg, fa, pa):
        """[B,P,3] points (+ per-point dirs or None) -> [B,P,C]   (siren.py:1509-1530)"""
        B, P = points.shape[0], points.shape[1]
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        points = _f32(points, self.device)
        ray_dirs = _f32(ray_dirs, self.device) if ray_dirs is not None else None
        out = torch.empty((B, P, self.C), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace("film", _lib.lib().fenerf_film_workspace_bytes(self._h, B))
            _lib.check(_lib.lib().fenerf_siren_forward(self._h, B, P, _ptr(points), _ptr(ray_dirs), _ptr(fg), _ptr(pg),
                                                       _ptr(fa), _ptr(pa), _ptr(out), C.c_void_p(ws.data_ptr()), _stream()))
        return out

    def siren_forward_pointwise(self, points, ray_dirs, fg, pg, fa, pa):
        """Per-point FiLM parameters (SPATIALSIRENGRID, siren.py:464-477): fg / pg [B,P,n_geo*H], fa / pa [B,P,n_color*H]
        -> [B,P,C].  The model must be precision 'f32'."""
        B, P = points.shape[0], points.shape[1]
        H, ng, nc = self.spec["hidden_dim"], self.spec["n_geo"], self.spec["n_color"]
        dev = self.device
        fg, pg, fa, pa = (_f32(t, dev) for t in (fg, pg, fa, pa))
        for t, n in ((fg, ng), (pg, ng), (fa, nc), (pa, nc)):
            if tuple(t.shape) != (B, P, n * H):
                raise ValueError(f"per-point film parameter of shape {tuple(t.shape)}, expected {(B, P, n * H)}")
        points = _f32(points, dev)
        ray_dirs = _f32(ray_dirs, dev) if ray_dirs is not None else None
        out = torch.empty((B, P, self.C), dtype=torch.float32, device=dev)
        # the FiLM pre-pass writes 2 L H floats per point (18 KB at H = 256): walk the points in slabs so that the scratch stays
        # below ~300 MB whatever P is
        slab = max(32, (1 << 28) // (8 * (ng + nc) * H) // 32 * 32)
        with torch.cuda.device(dev):
            for b in range(B):
                for s in range(0, P, slab):
                    n = min(slab, P - s)
                    ws = self._workspace("film_pw", _lib.lib().fenerf_film_workspace_bytes_pointwise(self._h, 1, n))
                    sl = lambda t: _ptr(t[b, s:s + n].contiguous()) if t is not None else None
                    _lib.check(_lib.lib().fenerf_siren_forward_pointwise(self._h, 1, n, sl(points), sl(ray_dirs), sl(fg), sl(pg), sl(fa),
                                                                         sl(pa), _ptr(out[b, s:s + n]), C.c_void_p(ws.data_ptr()), _stream()))
        return out

    # ---- per-point modulation under autograd (include/fenerf.h fenerf_siren_*_pointwise; SPATIALSIRENGRID, siren.py:464-477)
    def _film_pointwise(self, B, P, fg, pg, fa, pa):
        H, ng, nc = self.spec["hidden_dim"], self.spec["n_geo"], self.spec["n_color"]
        fg, pg, fa, pa = (_f32(t, self.device) for t in (fg, pg, fa, pa))
        if self.padded:
            fg, pg, fa, pa = self._pad_film(fg, ng), self._pad_film(pg, ng), self._pad_film(fa, nc), self._pad_film(pa, nc)
        for t, n in ((fg, ng), (pg, ng), (fa, nc), (pa, nc)):
            if tuple(t.shape) != (B, P, n * H):
                raise ValueError(f"per-point film parameter of shape {tuple(t.shape)}, expected {(B, P, n * H)}")
        return fg, pg, fa, pa

    def siren_forward_save_pointwise(self, points, ray_dirs, fg, pg, fa, pa):
        """Differentiable evaluation with one FiLM block per point: fg / pg [B,P,n_geo*H], fa / pa [B,P,n_color*H] -> (out [B,P,C], tape).
        The model must be precision 'f32', differentiable, without a feature grid; P a multiple of 32."""
        B, P = points.shape[0], points.shape[1]
        fg, pg, fa, pa = self._film_pointwise(B, P, fg, pg, fa, pa)
        points = _f32(points, self.device)
        ray_dirs = _f32(ray_dirs, self.device) if ray_dirs is not None else None
        out = torch.empty((B, P, self.C), dtype=torch.float32, device=self.device)
        tape = torch.empty((self.tape_floats(B * P),), dtype=torch.float32, device=self.device)
        l = _lib.lib()
        with torch.cuda.device(self.device):
            ws = self._workspace("film_pw", l.fenerf_film_workspace_bytes_pointwise(self._h, B, P))
            _lib.check(l.fenerf_siren_forward_save_pointwise(self._h, B, P, _ptr(points), _ptr(ray_dirs), _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa),
                                                             _ptr(out), _ptr(tape), C.c_void_p(ws.data_ptr()), _stream()))
        return out, tape

    def siren_backward_pointwise(self, points, ray_dirs, fg, pg, fa, pa, out, d_out, tape):
        """chain + weight gradients of the per-point-modulated SIREN -> dict like siren_param_grads, with d_freq_geo / d_phase_geo
        [B,P,n_geo*H] and d_freq_app / d_phase_app [B,P,n_color*H] (gradients wrt the RAW per-point parameters)."""
        sp = self.spec
        H, ng, nc = sp["hidden_dim"], sp["n_geo"], sp["n_color"]
        B, P = points.shape[0], points.shape[1]
        dev = self.device
        fg, pg, fa, pa = self._film_pointwise(B, P, fg, pg, fa, pa)
        out, d_out = _f32(out, dev), _f32(d_out, dev)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        res = dict(d_freq_geo=new(B, P, ng * H), d_phase_geo=new(B, P, ng * H), d_freq_app=new(B, P, nc * H), d_phase_app=new(B, P, nc * H),
                   geo_w=[new(H, 3)] + [new(H, H) for _ in range(ng - 1)], geo_b=[new(H) for _ in range(ng)],
                   color_w=[new(H, 3 + H)] + [new(H, H) for _ in range(nc - 1)], color_b=[new(H) for _ in range(nc)],
                   head_w=new(32, H), head_b=new(32), rgb_w=new(3, H), rgb_b=new(3))
        g = _lib.FenerfSirenGrads()
        for i in range(ng):
            g.geo_w[i], g.geo_b[i] = res["geo_w"][i].data_ptr(), res["geo_b"][i].data_ptr()
        for i in range(nc):
            g.color_w[i], g.color_b[i] = res["color_w"][i].data_ptr(), res["color_b"][i].data_ptr()
        for k in res:
            if not isinstance(res[k], list):
                setattr(g, k, res[k].data_ptr())
        l = _lib.lib()
        d_t = torch.empty((int(l.fenerf_siren_dtheta_floats(self._h, B * P)),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            fws = self._workspace("film_pw", l.fenerf_film_workspace_bytes_pointwise(self._h, B, P))
            ws = self._workspace("wgrad", l.fenerf_siren_grad_workspace_bytes(self._h, B, P))
            _lib.check(l.fenerf_siren_backward_pointwise(self._h, B, P, _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(out), _ptr(d_out), _ptr(tape),
                                                         _ptr(d_t), C.c_void_p(fws.data_ptr()), 0, _stream()))
            _lib.check(l.fenerf_siren_param_grads_pointwise(self._h, B, P, _ptr(_f32(points, dev)), _ptr(_f32(ray_dirs, dev)) if ray_dirs is not None else None,
                                                            _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(out), _ptr(d_out), _ptr(tape), _ptr(d_t),
                                                            C.byref(g), C.c_void_p(ws.data_ptr()), C.c_void_p(fws.data_ptr()), 1, _stream()))   # (prepared by the chain call above)
        return self._unpad_grads(res)

    def tape_floats(self, total_points, tape_format=0):
        """fp32 words of a tape for total_points points (L*H values per point + the slack the kernel's last workgroup may write; the
        16-bit tape, _lib.TAPE_U16, packs two values per word)"""
        return int(_lib.lib().fenerf_siren_tape_bytes(self._h, total_points, int(tape_format))) // 4

    def tape_words_per_point(self, tape_format=0):
        """fp32 words of tape per point (L*H, or half of it for the 16-bit tape): the stride of a point range inside a tape buffer"""
        LH = (self.spec["n_geo"] + self.spec["n_color"]) * self.spec["hidden_dim"]
        return LH // 2 if tape_format == _lib.TAPE_U16 else LH

    def dump_bytes_per_point(self, chunk_points=None, tape_format=0):
        """HBM bytes per (point x layer-feature) of the backward streams of a chunk (fenerf_siren_backward_stream_bytes_fmt):
        dict(chain_write, square_read, thin_read, tape)."""
        from .siren import autograd as _sa
        res = (C.c_double * 4)()
        _lib.check(_lib.lib().fenerf_siren_backward_stream_bytes_fmt(self._h, int(chunk_points or _sa.BACKWARD_CHUNK_POINTS), int(tape_format), res))
        return dict(chain_write=res[0], square_read=res[1], thin_read=res[2], tape_layer=res[3])

    def siren_forward_save(self, points, ray_dirs, fg, pg, fa, pa, out=None, tape=None, tape_e=None, tape_format=0):
        """Differentiable evaluation: like siren_forward, also returns the tape (pre-FiLM accumulators, tape_floats(B*P) floats
        of 32-point register dumps -- or, tape_format = _lib.TAPE_U16, the 16-bit phases of fenerf_layout.h "16-bit tape") and the
        sampled grid features [B*P,32] (None without a grid) that siren_backward consumes.
        out / tape / tape_e may be preallocated (contiguous views into larger buffers: several passes, one backward)."""
        B, P = points.shape[0], points.shape[1]
        H, L = self.spec["hidden_dim"], self.spec["n_geo"] + self.spec["n_color"]
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        points = _f32(points, self.device)
        ray_dirs = _f32(ray_dirs, self.device) if ray_dirs is not None else None
        if out is None:
            out = torch.empty((B, P, self.C), dtype=torch.float32, device=self.device)
        if tape is None:
            tape = torch.empty((self.tape_floats(B * P, tape_format),), dtype=torch.float32, device=self.device)
        if tape_e is None and self.spec["grid_ch"]:
            tape_e = torch.empty((B * P, 32), dtype=torch.float32, device=self.device)
        assert out.numel() == B * P * self.C and tape.numel() >= self.tape_words_per_point(tape_format) * B * P    # + the slack behind the last pass
        with torch.cuda.device(self.device):
            ws = self._workspace("film", _lib.lib().fenerf_film_workspace_bytes(self._h, B))
            _lib.check(_lib.lib().fenerf_siren_forward_save_fmt(self._h, B, P, _ptr(points), _ptr(ray_dirs), _ptr(fg), _ptr(pg), _ptr(fa),
                                                                _ptr(pa), _ptr(out), _ptr(tape), _ptr(tape_e), C.c_void_p(ws.data_ptr()),
                                                                int(tape_format), _stream()))
        return out, tape, tape_e

    def siren_backward(self, B, P, fg, pg, fa, pa, out, d_out, tape, tape_format=0):
        """-> (d_t = dL/dtheta per FiLM layer in the tape's layout + the per-tile FiLM sums (opaque, fenerf_siren_dtheta_floats),
        d_e [B*P,32] or None)"""
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        out, d_out = _f32(out, self.device), _f32(d_out, self.device)
        d_t = torch.empty((int(_lib.lib().fenerf_siren_dtheta_floats(self._h, B * P)),), dtype=torch.float32, device=self.device)
        d_e = torch.empty((B * P, 32), dtype=torch.float32, device=self.device) if self.spec["grid_ch"] else None
        with torch.cuda.device(self.device):
            ws = self._workspace("film", _lib.lib().fenerf_film_workspace_bytes(self._h, B))
            _lib.check(_lib.lib().fenerf_siren_backward_fmt(self._h, B, P, _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(out), _ptr(d_out),
                                                            _ptr(tape), int(tape_format), _ptr(d_t), _ptr(d_e), C.c_void_p(ws.data_ptr()), _stream()))
        return d_t, d_e

    def film_sums_floats(self, B, P):
        """floats of the per-tile FiLM sums fenerf_siren_backward_film writes for B images of P points"""
        return int(_lib.lib().fenerf_siren_film_sums_floats(self._h, int(B), int(P)))

    def film_only_native(self):
        """fenerf_siren_backward_film / fenerf_siren_film_grads exist for this model (f16x3 handles)"""
        return self.precision == "f16x3" and self.differentiable

    def siren_backward_film(self, B, P, fg, pg, fa, pa, out, d_out, tape):
        """Inversion: the chain without its d(theta) dump -> the per-tile FiLM sums (opaque; for siren_film_grads)."""
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        out, d_out = _f32(out, self.device), _f32(d_out, self.device)
        l = _lib.lib()
        sums = torch.empty((self.film_sums_floats(B, P),), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace("film", l.fenerf_film_workspace_bytes(self._h, B))
            _lib.check(l.fenerf_siren_backward_film(self._h, B, P, _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(out), _ptr(d_out), _ptr(tape),
                                                    _ptr(sums), C.c_void_p(ws.data_ptr()), _stream()))
        return sums

    def siren_film_grads(self, B, P, fg, pg, fa, pa, film_sums):
        """FiLM sums -> {d_freq_geo, d_phase_geo [B, n_geo*H], d_freq_app, d_phase_app [B, n_color*H]} (gradients wrt the RAW parameters)."""
        sp = self.spec
        H, ng, nc = sp["hidden_dim"], sp["n_geo"], sp["n_color"]
        dev = self.device
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        res = dict(d_freq_geo=new(B, ng * H), d_phase_geo=new(B, ng * H), d_freq_app=new(B, nc * H), d_phase_app=new(B, nc * H))
        g = _lib.FenerfSirenGrads()
        for k in res:
            setattr(g, k, res[k].data_ptr())
        l = _lib.lib()
        with torch.cuda.device(dev):
            fws = self._workspace("film", l.fenerf_film_workspace_bytes(self._h, B))
            ws = self._workspace("wgrad", l.fenerf_siren_grad_workspace_bytes(self._h, B, P))
            _lib.check(l.fenerf_siren_film_grads(self._h, B, P, _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(film_sums), C.byref(g),
                                                 C.c_void_p(ws.data_ptr()), C.c_void_p(fws.data_ptr()), _stream()))
        return self._unpad_grads(res)

    def siren_backward_grid(self, B, P, fg, pg, fa, pa, out, d_out, tape, points, d_grid_cl, tape_format=0):
        """siren_backward whose gradient wrt the sampled grid features is scattered (accumulated) straight into d_grid_cl
        [D,H,W,32] (zero-initialised by the caller before the first chunk) -> d_t.  f16x3 models scatter inside the chain kernel;
        others run the chain and the scatter kernel over a scratch d_e."""
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        out, d_out = _f32(out, self.device), _f32(d_out, self.device)
        points = _f32(points, self.device).reshape(B * P, 3)
        d_t = torch.empty((int(_lib.lib().fenerf_siren_dtheta_floats(self._h, B * P)),), dtype=torch.float32, device=self.device)
        fused = bool(_lib.lib().fenerf_siren_backward_fuses_grid(self._h))
        scratch = None if fused else torch.empty((B * P, 32), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace("film", _lib.lib().fenerf_film_workspace_bytes(self._h, B))
            _lib.check(_lib.lib().fenerf_siren_backward_grid_fmt(self._h, B, P, _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(out), _ptr(d_out),
                                                                 _ptr(tape), int(tape_format), _ptr(points), _ptr(d_t), _ptr(d_grid_cl),
                                                                 _ptr(scratch), C.c_void_p(ws.data_ptr()), _stream()))
        return d_t

    def siren_input_grads(self, points, fg, pg, fa, pa, d_t, w_geo0, w_color0, d_points=None, d_dirs=None):
        """Gradients wrt the SIREN's inputs from the fp32 d(theta) dump `d_t` of siren_backward / siren_backward_grid over the same
        (points, FiLM parameters): fills d_points / d_dirs [B,P,3] (contiguous fp32 device tensors; either may be None).
        w_geo0 [H,3] / w_color0 [H, 3+G+H]: the nn.Linear weights of layer 0 and of colour layer 0 at the module's width
        (include/fenerf.h fenerf_siren_input_grads; what autograd leaves in input.grad / ray_directions.grad, siren.py:1509-1530)."""
        B, P = points.shape[0], points.shape[1]
        dev = self.device
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        w0, wc0 = _f32(w_geo0, dev), _f32(w_color0, dev)
        if self.padded:       # zero rows for the padded features (their d theta is zero anyway); the x columns of colour layer 0 are not read
            d = self.spec["hidden_dim"] - self.logical_H
            w0, wc0 = torch.nn.functional.pad(w0, (0, 0, 0, d)), torch.nn.functional.pad(wc0, (0, 0, 0, d))
        l = _lib.lib()
        with torch.cuda.device(dev):
            fws = self._workspace("film", l.fenerf_film_workspace_bytes(self._h, B))
            _lib.check(l.fenerf_siren_input_grads(self._h, B, P, _ptr(_f32(points, dev)), _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(d_t), _ptr(w0),
                                                  _ptr(wc0), int(wc0.shape[1]), _ptr(d_points), _ptr(d_dirs), C.c_void_p(fws.data_ptr()), _stream()))
        return d_points, d_dirs

    def grid_gradient_ncdhw(self, d_grid_cl):
        """channels-last gradient grid [D,H,W,32] -> the parameter's layout [1,32,D,H,W]"""
        D, Hh, W = d_grid_cl.shape[:3]
        out = torch.empty((1, 32, D, Hh, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().fenerf_grid_gradient_ncdhw(self._h, _ptr(d_grid_cl), _ptr(out), _stream()))
        return out

    def siren_param_grads(self, points, ray_dirs, fg, pg, fa, pa, out, d_out, tape, tape_e, d_t, film_only=False, tape_format=0, weights=None):
        """(tape, d_t) -> dict of parameter gradients in nn.Linear layout: geo_w/geo_b/color_w/color_b lists, head_w [32,H]
        (folded label rows + sigma row), head_b [32], rgb_w [3,H], rgb_b [3], d_freq_geo / d_phase_geo [B,n_geo*H],
        d_freq_app / d_phase_app [B,n_color*H].  tape_format = _lib.TAPE_U16 needs `weights` = (geo weights, colour weights): the
        FiLM layers' nn.Linear weight tensors on the device (the frequency gradients are formed from the weight-gradient sums)."""
        sp = self.spec
        H, ng, nc, G = sp["hidden_dim"], sp["n_geo"], sp["n_color"], sp["grid_ch"]
        B, P = points.shape[0], points.shape[1]
        dev = self.device
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        res = dict(d_freq_geo=new(B, ng * H), d_phase_geo=new(B, ng * H), d_freq_app=new(B, nc * H), d_phase_app=new(B, nc * H))
        g = _lib.FenerfSirenGrads()
        if not film_only:
            res.update(geo_w=[new(H, 3)] + [new(H, H) for _ in range(ng - 1)], geo_b=[new(H) for _ in range(ng)],
                       color_w=[new(H, 3 + G + H)] + [new(H, H) for _ in range(nc - 1)], color_b=[new(H) for _ in range(nc)],
                       head_w=new(32, H), head_b=new(32), rgb_w=new(3, H), rgb_b=new(3))
            for i in range(ng):
                g.geo_w[i], g.geo_b[i] = res["geo_w"][i].data_ptr(), res["geo_b"][i].data_ptr()
            for i in range(nc):
                g.color_w[i], g.color_b[i] = res["color_w"][i].data_ptr(), res["color_b"][i].data_ptr()
        for k in res:
            if not isinstance(res[k], list):
                setattr(g, k, res[k].data_ptr())
        l = _lib.lib()
        with torch.cuda.device(dev):
            fws = self._workspace("film", l.fenerf_film_workspace_bytes(self._h, B))
            ws = self._workspace("wgrad", l.fenerf_siren_grad_workspace_bytes(self._h, B, P))
            wts, keep = None, []
            weights = self._pad_weights(weights)
            if weights is not None:
                wts = _lib.FenerfSirenGrads()
                for i, w in enumerate(weights[0]):
                    keep.append(_f32(w.detach(), dev)); wts.geo_w[i] = keep[-1].data_ptr()
                for i, w in enumerate(weights[1]):
                    keep.append(_f32(w.detach(), dev)); wts.color_w[i] = keep[-1].data_ptr()
            _lib.check(l.fenerf_siren_param_grads_fmt(self._h, B, P, _ptr(_f32(points, dev)), _ptr(_f32(ray_dirs, dev)) if ray_dirs is not None else None,
                                                      _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(out), _ptr(d_out), _ptr(tape), int(tape_format),
                                                      _ptr(tape_e), _ptr(d_t), C.byref(g), C.byref(wts) if wts is not None else None,
                                                      C.c_void_p(ws.data_ptr()), C.c_void_p(fws.data_ptr()), _stream()))
        return self._unpad_grads(res)

    def grid_backward(self, points, d_e, grid_shape):
        """Scatter d_e [Ptot,32] into the gradient of spatial_embeddings; returns it in the parameter's [1,32,D,H,W] shape."""
        D, Hh, W = grid_shape
        points = _f32(points, self.device).reshape(-1, 3)
        g = torch.zeros((D, Hh, W, 32), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().fenerf_grid_backward(self._h, points.shape[0], _ptr(points), _ptr(d_e), _ptr(g), _stream()))
            out = torch.empty((1, 32, D, Hh, W), dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().fenerf_grid_gradient_ncdhw(self._h, _ptr(g), _ptr(out), _stream()))
        return out

    def siren_forward_rays(self, origins, dirs, z, fg, pg, fa, pa, lock_view=False):
        """origins/dirs [B,R,3], z [B,R,N] -> [B,R,N,C]"""
        B, R, N = z.shape
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        origins, dirs, z = _f32(origins, self.device), _f32(dirs, self.device), _f32(z, self.device)
        out = torch.empty((B, R, N, self.C), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace("film", _lib.lib().fenerf_film_workspace_bytes(self._h, B))
            _lib.check(_lib.lib().fenerf_siren_forward_rays(self._h, B, R, N, _ptr(origins), _ptr(dirs), _ptr(z), int(lock_view),
                                                            _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), _ptr(out),
                                                            C.c_void_p(ws.data_ptr()), _stream()))
        return out

    def time_siren_rays(self, origins, dirs, z, fg, pg, fa, pa, iters=10):
        """Average duration (ms) of the SIREN kernel alone on these rays, hipEvent-timed on the current stream."""
        B, R, N = z.shape
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        origins, dirs, z = _f32(origins, self.device), _f32(dirs, self.device), _f32(z, self.device)
        out = torch.empty((B, R, N, self.C), dtype=torch.float32, device=self.device)
        ms = C.c_float(0)
        with torch.cuda.device(self.device):
            ws = self._workspace("film", _lib.lib().fenerf_film_workspace_bytes(self._h, B))
            _lib.check(_lib.lib().fenerf_siren_time_rays(self._h, B, R, N, _ptr(origins), _ptr(dirs), _ptr(z), _ptr(fg), _ptr(pg),
                                                         _ptr(fa), _ptr(pa), _ptr(out), C.c_void_p(ws.data_ptr()), int(iters),
                                                         C.byref(ms), _stream()))
        return float(ms.value)

    def clock_probe(self, origins, dirs, z, fg, pg, fa, pa, iters=10):
        """time_siren_rays with in-kernel clock stamps -> dict(kernel_ms, cycles_per_launch, clock_ghz, wall_clock_khz): the shader
        cycles a launch takes and the clock the power manager granted while it ran (fenerf_siren_clock_probe)."""
        B, R, N = z.shape
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        origins, dirs, z = _f32(origins, self.device), _f32(dirs, self.device), _f32(z, self.device)
        out = torch.empty((B, R, N, self.C), dtype=torch.float32, device=self.device)
        res = (C.c_double * 4)()
        with torch.cuda.device(self.device):
            ws = self._workspace("film", _lib.lib().fenerf_film_workspace_bytes(self._h, B))
            _lib.check(_lib.lib().fenerf_siren_clock_probe(self._h, B, R, N, _ptr(origins), _ptr(dirs), _ptr(z), _ptr(fg), _ptr(pg),
                                                           _ptr(fa), _ptr(pa), _ptr(out), C.c_void_p(ws.data_ptr()), int(iters), res,
                                                           _stream()))
        return dict(kernel_ms=res[0], cycles_per_launch=res[1], clock_ghz=res[2], wall_clock_khz=res[3])

    def executed_flop_per_point(self):
        """FLOPs the SIREN kernel issues on the matrix pipe per point (static MFMA count), next to the algorithmic 1,603,584"""
        return float(_lib.lib().fenerf_siren_executed_flop_per_point(self._h))

    def render_forward_save(self, origins, dirs, z_coarse, u, noise_coarse, noise_final, fg, pg, fa, pa, opts, lock_view=False, tape_format=0):
        """fenerf_render_forward_save: the differentiable hierarchical render's forward in ONE call -> (rgb [B,R,C-1], depth [B,R], save);
        `save` (opaque bytes on the device) is what render_backward needs besides z_coarse / noise_final / opts."""
        B, R, N = z_coarse.shape
        dev = self.device
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        origins, dirs, z_coarse, u = _f32(origins, dev), _f32(dirs, dev), _f32(z_coarse, dev), _f32(u, dev)
        noise_coarse = _f32(noise_coarse, dev) if noise_coarse is not None else None
        noise_final = _f32(noise_final, dev) if noise_final is not None else None
        l = _lib.lib()
        rgb = torch.empty((B, R, self.C - 1), dtype=torch.float32, device=dev)
        depth = torch.empty((B, R), dtype=torch.float32, device=dev)
        save = torch.empty((int(l.fenerf_render_save_bytes(self._h, B, R, N, int(tape_format), int(lock_view))),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(l.fenerf_render_forward_save(self._h, B, R, N, int(lock_view), _ptr(origins), _ptr(dirs), _ptr(z_coarse), _ptr(u), _ptr(noise_coarse),
                                                    _ptr(noise_final), _ptr(fg), _ptr(pg), _ptr(fa), _ptr(pa), C.byref(opts), _ptr(rgb), _ptr(depth),
                                                    C.c_void_p(save.data_ptr()), C.c_size_t(save.numel()), int(tape_format), _stream()))
        return rgb, depth, save

    def _render_grad_buffers(self, B, film_only):
        """-> (dict like siren_param_grads of freshly allocated gradient buffers, the FenerfSirenGrads pointing at them, d_grid or None)"""
        sp = self.spec
        H, ng, nc, G = sp["hidden_dim"], sp["n_geo"], sp["n_color"], sp["grid_ch"]
        dev = self.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        res = dict(d_freq_geo=new(B, ng * H), d_phase_geo=new(B, ng * H), d_freq_app=new(B, nc * H), d_phase_app=new(B, nc * H))
        g = _lib.FenerfSirenGrads()
        if not film_only:
            res.update(geo_w=[new(H, 3)] + [new(H, H) for _ in range(ng - 1)], geo_b=[new(H) for _ in range(ng)],
                       color_w=[new(H, 3 + G + H)] + [new(H, H) for _ in range(nc - 1)], color_b=[new(H) for _ in range(nc)],
                       head_w=new(32, H), head_b=new(32), rgb_w=new(3, H), rgb_b=new(3))
            for i in range(ng):
                g.geo_w[i], g.geo_b[i] = res["geo_w"][i].data_ptr(), res["geo_b"][i].data_ptr()
            for i in range(nc):
                g.color_w[i], g.color_b[i] = res["color_w"][i].data_ptr(), res["color_b"][i].data_ptr()
        for k in res:
            if not isinstance(res[k], list):
                setattr(g, k, res[k].data_ptr())
        d_grid = torch.empty((1, 32) + tuple(self.grid_shape), dtype=torch.float32, device=dev) if (G and not film_only) else None
        return res, g, d_grid

    def _film_weight_struct(self, weights):
        """film_layer_weights(...) -> (FenerfSirenGrads of their pointers or None, keep-alive list)"""
        if weights is None:
            return None, []
        weights = self._pad_weights(weights)
        wts, keep = _lib.FenerfSirenGrads(), []
        for i, w in enumerate(weights[0]):
            keep.append(_f32(w.detach(), self.device)); wts.geo_w[i] = keep[-1].data_ptr()
        for i, w in enumerate(weights[1]):
            keep.append(_f32(w.detach(), self.device)); wts.color_w[i] = keep[-1].data_ptr()
        return wts, keep

    def render_backward(self, B, R, N, save, z_coarse, noise_final, opts, g_rgb, film_only, lock_view=False, tape_format=0, weights=None,
                        chunk_points=0, film_sums_budget_bytes=0):
        """fenerf_render_backward: every gradient of the render in ONE call -> (dict like siren_param_grads -- FiLM gradients [B, n*H], both
        passes summed; weight / bias gradients unless film_only --, d_grid [1,32,D,H,W] or None)."""
        dev = self.device
        res, g, d_grid = self._render_grad_buffers(B, film_only)
        wts, keep = self._film_weight_struct(weights)
        l = _lib.lib()
        with torch.cuda.device(dev):
            ws = self._workspace("render_bwd", l.fenerf_render_backward_workspace_bytes(self._h, B, R, N, int(film_only), int(chunk_points),
                                                                                        int(film_sums_budget_bytes)))
            _lib.check(l.fenerf_render_backward(self._h, B, R, N, int(lock_view), C.c_void_p(save.data_ptr()), C.c_size_t(save.numel()), int(tape_format),
                                                _ptr(_f32(z_coarse, dev)), _ptr(_f32(noise_final, dev)) if noise_final is not None else None,
                                                C.byref(opts), _ptr(_f32(g_rgb, dev)), C.byref(g), _ptr(d_grid), C.byref(wts) if wts is not None else None,
                                                int(chunk_points), int(film_sums_budget_bytes), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), _stream()))
        return self._unpad_grads(res), d_grid

    def render_backward_stage(self, stage, keep_chunks, B, R, N, save, z_coarse, noise_final, opts, g_rgb, lock_view=False, tape_format=0,
                              weights=None, chunk_points=0, carry=None):
        """fenerf_render_backward_stage.  stage 1 (carry None): -> (d_grid [1,32,D,H,W] finished, carry) -- carry holds the gradient buffers
        and the workspace (own allocation: it must survive until stage 2, whatever else renders in between); stage 2 (carry from stage 1):
        -> the finished dict like siren_param_grads."""
        dev = self.device
        l = _lib.lib()
        wts, keep = self._film_weight_struct(weights)
        with torch.cuda.device(dev):
            if stage == 1:
                res, g, d_grid = self._render_grad_buffers(B, False)
                nbytes = int(l.fenerf_render_backward_split_workspace_bytes(self._h, B, R, N, int(chunk_points), int(keep_chunks)))
                # The workspace must survive until stage 2.  The model's persistent scratch when no other two-stage backward holds it (a
                # fresh 10-GB tensor per backward pass fragments the caching allocator between the forward's 54-GB save block and this:
                # + 1 ms per 6-image step); a second render of the same graph whose stage 1 runs before this one's stage 2 gets its own.
                cached = not getattr(self, "_split_ws_busy", False)
                if cached:
                    self._split_ws_busy = True
                    ws = self._workspace("render_bwd_split", nbytes)
                else:
                    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
                carry = dict(res=res, g=g, d_grid=d_grid, ws=ws, cached=cached)
            res, g, d_grid, ws = carry["res"], carry["g"], carry["d_grid"], carry["ws"]
            try:
                _lib.check(l.fenerf_render_backward_stage(self._h, int(stage), int(keep_chunks), B, R, N, int(lock_view), C.c_void_p(save.data_ptr()),
                                                          C.c_size_t(save.numel()), int(tape_format),
                                                          _ptr(_f32(z_coarse, dev)) if z_coarse is not None else None,
                                                          _ptr(_f32(noise_final, dev)) if noise_final is not None else None, C.byref(opts),
                                                          _ptr(_f32(g_rgb, dev)) if g_rgb is not None else None, C.byref(g), _ptr(d_grid),
                                                          C.byref(wts) if wts is not None else None, int(chunk_points), C.c_void_p(ws.data_ptr()),
                                                          C.c_size_t(ws.numel()), _stream()))
            except Exception:
                self.release_split_workspace(carry)      # a failed stage hands no carry back: the persistent scratch must not stay claimed
                raise
        if stage == 1:
            carry["d_grid"] = None          # finished: the caller's (autograd may take the tensor as the parameter's .grad without a copy)
            return d_grid, carry
        self.release_split_workspace(carry)
        return self._unpad_grads(res)

    def release_split_workspace(self, carry):
        """stage 2 has run -- or never will (the backward pass ended without it): the persistent two-stage scratch is free again"""
        if carry is not None and carry.get("cached"):
            carry["cached"] = False
            self._split_ws_busy = False

    def render(self, origins, dirs, z_coarse, u, noise_coarse, noise_final, fg, pg, fa, pa, opts, hierarchical=True,
               lock_view=False, want_weights=False, want_wsum=False):
        """The fused coarse->resample->fine->merge->composite pipeline (generators.py:479-519).
        Returns (rgb [B,R,C'], depth [B,R], weights [B,R,M] or None, wsum [B,R] or None)."""
        B, R, N = z_coarse.shape
        dev = self.device
        fg, pg, fa, pa = self._film(B, fg, pg, fa, pa)
        origins, dirs, z_coarse = _f32(origins, dev), _f32(dirs, dev), _f32(z_coarse, dev)
        u = _f32(u, dev) if u is not None else None
        noise_coarse = _f32(noise_coarse, dev) if noise_coarse is not None else None
        noise_final = _f32(noise_final, dev) if noise_final is not None else None
        M = 2 * N if hierarchical else N
        pad = opts.fill_mode in (_lib.FILL["seg_padding_background"], _lib.FILL["eval_seg_padding_background"])
        Cout = self.C if pad else self.C - 1
        rgb = torch.empty((B, R, Cout), dtype=torch.float32, device=dev)
        depth = torch.empty((B, R), dtype=torch.float32, device=dev)
        weights = torch.empty((B, R, M), dtype=torch.float32, device=dev) if want_weights else None
        wsum = torch.empty((B, R), dtype=torch.float32, device=dev) if want_wsum else None
        l = _lib.lib()
        with torch.cuda.device(dev):
            nbytes = l.fenerf_render_workspace_bytes(self._h, B, R, N, int(hierarchical))
            ws = self._workspace("render", nbytes)
            _lib.check(l.fenerf_render_forward(self._h, B, R, N, int(hierarchical), int(lock_view), _ptr(origins), _ptr(dirs),
                                               _ptr(z_coarse), _ptr(u), _ptr(noise_coarse), _ptr(noise_final), _ptr(fg),
                                               _ptr(pg), _ptr(fa), _ptr(pa), C.byref(opts), _ptr(rgb), _ptr(depth),
                                               _ptr(weights), _ptr(wsum), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()),
                                               _stream()))
        return rgb, depth, weights, wsum


class NativeLocalModel:
    """Owns a FenerfLocalModel*: SPATIALSIRENGRID's SIREN and its per-point mapping network packed into one fp32 stream; forward()
    evaluates both in one launch (fenerf_siren_forward_local).  sd / spec as NativeModel, mp = the mapping network's six tensors."""

    def __init__(self, sd, spec, mp, device):
        self.spec = dict(spec)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("fenerf_amd renders on the GPU only (there is no CPU path); got device %s" % device)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            d, keep = _lib.make_desc(sd, spec, "f32")
            md, keep2 = _lib.make_local_map_desc(mp)
            _lib.check(_lib.lib().fenerf_local_model_create(C.byref(d), C.byref(md), C.byref(self._h)))

    def forward(self, points, ray_dirs, latents):
        """points [B,P,3] local coordinates, ray_dirs [B,P,3] or None, latents [B,P,32] -> [B,P,4] = [rgb | sigma]"""
        B, P = points.shape[0], points.shape[1]
        dev = self.device
        points, latents = _f32(points, dev), _f32(latents, dev)
        ray_dirs = _f32(ray_dirs, dev) if ray_dirs is not None else None
        if tuple(latents.shape) != (B, P, 32):
            raise ValueError(f"local latents of shape {tuple(latents.shape)}, expected {(B, P, 32)}")
        out = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().fenerf_siren_forward_local(self._h, B * P, _ptr(points), _ptr(ray_dirs), _ptr(latents), _ptr(out), _stream()))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().fenerf_local_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class render_fusion:
    """with native.render_fusion("off" | "force" | "auto"): ...  -- how the calling thread's hierarchical renders are launched inside the
    block (include/fenerf.h fenerf_set_render_fusion): one launch when the shape allows it and it is balanced ("auto", the default), always
    four launches ("off"), one launch whenever the shape allows it ("force").  The results are the same bit for bit."""
    MODES = {"auto": _lib.FUSION_AUTO, "off": _lib.FUSION_OFF, "force": _lib.FUSION_FORCE}

    def __init__(self, mode):
        self.mode = self.MODES[mode]

    def __enter__(self):
        self._prev = _lib.lib().fenerf_set_render_fusion(self.mode)
        return self

    def __exit__(self, *exc):
        _lib.lib().fenerf_set_render_fusion(self._prev)
        return False


class cu_budget:
    """with native.cu_budget(n): ...  -- the calling thread's launches inside the block are sized for at most n compute units
    (include/fenerf.h fenerf_set_cu_budget; 0 / None = the whole device).  Workspace sizes are queried under the same setting."""

    def __init__(self, cus):
        self.cus = int(cus or 0)

    def __enter__(self):
        self._prev = _lib.lib().fenerf_set_cu_budget(self.cus)
        return self

    def __exit__(self, *exc):
        _lib.lib().fenerf_set_cu_budget(self._prev)
        return False


class phase_timing:
    """with native.phase_timing() as t: ...   ->  t.ms / t.calls: {phase name: device ms / launch groups} of everything the library
    launched inside the block (hipEvent pairs around each launch group, include/fenerf.h fenerf_phase_timing).  Synchronises on exit."""

    def __enter__(self):
        l = _lib.lib()
        ms, calls = (C.c_double * _lib.N_PHASES)(), (C.c_int * _lib.N_PHASES)()
        _lib.check(l.fenerf_phase_times(ms, calls, _lib.N_PHASES))      # drop whatever an earlier block left behind
        self._prev = l.fenerf_phase_timing(1)
        self.ms, self.calls = {}, {}
        return self

    def __exit__(self, *exc):
        l = _lib.lib()
        l.fenerf_phase_timing(self._prev)
        torch.cuda.synchronize()
        ms, calls = (C.c_double * _lib.N_PHASES)(), (C.c_int * _lib.N_PHASES)()
        _lib.check(l.fenerf_phase_times(ms, calls, _lib.N_PHASES))
        for i in range(_lib.N_PHASES):
            if calls[i]:
                name = l.fenerf_phase_name(i).decode()
                self.ms[name], self.calls[name] = float(ms[i]), int(calls[i])
        return False


def forward_kernel_name(nat):
    """Name of the no-grad forward SIREN kernel a NativeModel launches (bench / profile labels; fenerf_siren*.hip)."""
    H, g = nat.spec["hidden_dim"], "true" if nat.spec["grid_ch"] else "false"
    if nat.precision == "f32":
        return f"siren_kernel<{H}, {g}, false>"
    # <H, GRID, SAVE (0 no tape / 1 fp32 / 2 16-bit), FUSED (one-launch render), TERMS2 (0 f16x3 / 1 f16x2 / 2 f16x3c2)>: as rocprofv3 prints it
    return f"siren16w_kernel<{H}, {g}, 0, false, {getattr(nat, 'forward_mode', 0)}>"


# ----------------------------------------------------------------------
# CustomMappingNetwork as one launch forward / three backward (fenerf_mapping.hip; small batches)
# ----------------------------------------------------------------------
def _mapping_net(weights, biases):
    """[W_0 .. W_{L-1}], [b_0 .. b_{L-1}] contiguous fp32 device tensors in nn.Linear layout -> (FenerfMappingNet, keep-alive)"""
    L = len(weights)
    if not 2 <= L <= _lib.MAP_MAX_LAYERS:
        raise ValueError(f"mapping network with {L} linear layers (supported: 2 .. {_lib.MAP_MAX_LAYERS})")
    net = _lib.FenerfMappingNet()
    net.n_layers, net.z_dim, net.hidden, net.out_dim = L, weights[0].shape[1], weights[0].shape[0], weights[-1].shape[0]
    for l, (w, b) in enumerate(zip(weights, biases)):
        want = (net.out_dim if l == L - 1 else net.hidden, net.z_dim if l == 0 else net.hidden)
        if tuple(w.shape) != want or tuple(b.shape) != (want[0],):
            raise ValueError(f"mapping network layer {l}: weight {tuple(w.shape)} / bias {tuple(b.shape)}, expected {want} / {(want[0],)}")
        net.W[l], net.b[l] = w.data_ptr(), b.data_ptr()
    return net


def mapping_forward(weights, biases, z):
    """z [B, z_dim] -> (out [B, out_dim], acts [L-1, B, hidden]) in ONE launch (fenerf_mapping_forward; siren.py:82-102)"""
    dev = z.device
    weights, biases, z = [_f32(w, dev) for w in weights], [_f32(b, dev) for b in biases], _f32(z, dev)
    net = _mapping_net(weights, biases)
    B = z.shape[0]
    out = torch.empty((B, net.out_dim), dtype=torch.float32, device=dev)
    acts = torch.empty((net.n_layers - 1, B, net.hidden), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_mapping_forward(C.byref(net), B, _ptr(z), _ptr(acts), _ptr(out), _stream()))
    return out, acts


def mapping_backward(weights, biases, z, acts, d_out):
    """-> ([dW_l], [db_l]) of the parameters' shapes, three launches (fenerf_mapping_backward)"""
    dev = z.device
    weights, biases, z, d_out = [_f32(w, dev) for w in weights], [_f32(b, dev) for b in biases], _f32(z, dev), _f32(d_out, dev)
    net = _mapping_net(weights, biases)
    B = z.shape[0]
    dW, db = [torch.empty_like(w) for w in weights], [torch.empty_like(b) for b in biases]
    pw, pb = (C.c_void_p * len(dW))(*[t.data_ptr() for t in dW]), (C.c_void_p * len(db))(*[t.data_ptr() for t in db])
    l = _lib.lib()
    ws = torch.empty((max(1, int(l.fenerf_mapping_workspace_floats(C.byref(net), B))),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(l.fenerf_mapping_backward(C.byref(net), B, _ptr(z), _ptr(acts), _ptr(d_out), pw, pb, _ptr(ws), _stream()))
    return dW, db


def label_head_backward(label_params, g_head_w, g_head_b):
    """fenerf_label_head_backward: [(W_i, b_i)] of label_layer_linear in application order + the gradient of their fold (g_head_w [n_lab, H],
    g_head_b [n_lab]) -> [(dW_i, db_i)] of the parameters' shapes; one launch (two layers) or two (three)."""
    dev = g_head_w.device
    Ws, bs = [_f32(W.detach(), dev) for W, _ in label_params], [_f32(b.detach(), dev) for _, b in label_params]
    gA, gc = _f32(g_head_w, dev), _f32(g_head_b, dev)
    n, (n_lab, H) = len(Ws), gA.shape
    if tuple(Ws[-1].shape) != (n_lab, H) or any(tuple(W.shape) != (H, H) for W in Ws[:-1]):
        raise ValueError("label head: layers must be [H, H] ... [n_lab, H] with the fold's gradient [n_lab, H]")
    flat = torch.empty((sum(W.numel() + b.numel() for W, b in zip(Ws, bs)),), dtype=torch.float32, device=dev)     # one allocation: 2n views
    dW, db, off = [], [], 0
    for W, b in zip(Ws, bs):
        dW.append(flat[off:off + W.numel()].view_as(W)); off += W.numel()
        db.append(flat[off:off + b.numel()].view_as(b)); off += b.numel()
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    l = _lib.lib()
    ws = torch.empty((max(1, int(l.fenerf_label_head_workspace_floats(H))),), dtype=torch.float32, device=dev) if n == 3 else None
    with torch.cuda.device(dev):
        _lib.check(l.fenerf_label_head_backward(n, H, n_lab, arr(Ws), arr(bs), _ptr(gA), _ptr(gc), arr(dW), arr(db), _ptr(ws), _stream()))
    return list(zip(dW, db))


# ----------------------------------------------------------------------
# stand-alone ray-tail ops (no model needed)
# ----------------------------------------------------------------------
def ray_setup(B, img_size, N, z_cam, ray_start, ray_end, u_jitter, theta, phi):
    """u_jitter [B,R,N(,1)], theta/phi [B(,1)] device tensors -> (origins [B,R,3], dirs [B,R,3], z [B,R,N], pitch [B,1], yaw [B,1])"""
    dev = u_jitter.device
    R = img_size * img_size
    u, th, ph = _f32(u_jitter, dev).reshape(B, R, N), _f32(theta, dev).reshape(B), _f32(phi, dev).reshape(B)
    origins = torch.empty((B, R, 3), dtype=torch.float32, device=dev)
    dirs = torch.empty((B, R, 3), dtype=torch.float32, device=dev)
    z = torch.empty((B, R, N), dtype=torch.float32, device=dev)
    pitch = torch.empty((B, 1), dtype=torch.float32, device=dev)
    yaw = torch.empty((B, 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_ray_setup(B, img_size, N, float(z_cam), float(ray_start), float(ray_end), _ptr(u), _ptr(th),
                                               _ptr(ph), _ptr(origins), _ptr(dirs), _ptr(z), _ptr(pitch), _ptr(yaw), _stream()))
    return origins, dirs, z, pitch, yaw


def composite(rgb_sigma, z, noise, opts, want_weights=True, want_wsum=True):
    """fancy_integration on [..., M, C] / [..., M] device tensors -> (rgb, depth, weights, wsum)."""
    lead = rgb_sigma.shape[:-2]
    M, Cc = rgb_sigma.shape[-2], rgb_sigma.shape[-1]
    dev = rgb_sigma.device
    BR = int(np.prod(lead)) if len(lead) else 1
    rs, zz = _f32(rgb_sigma, dev).reshape(BR, M, Cc), _f32(z, dev).reshape(BR, M)
    nz = _f32(noise, dev).reshape(BR, M) if noise is not None else None
    pad = opts.fill_mode in (_lib.FILL["seg_padding_background"], _lib.FILL["eval_seg_padding_background"])
    Cout = Cc if pad else Cc - 1
    rgb = torch.empty((BR, Cout), dtype=torch.float32, device=dev)
    depth = torch.empty((BR,), dtype=torch.float32, device=dev)
    w = torch.empty((BR, M), dtype=torch.float32, device=dev) if want_weights else None
    ws = torch.empty((BR,), dtype=torch.float32, device=dev) if want_wsum else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_composite(BR, M, Cc, _ptr(rs), _ptr(zz), _ptr(nz), C.byref(opts), _ptr(rgb), _ptr(depth),
                                               _ptr(w), _ptr(ws), _stream()))
    return (rgb.reshape(*lead, Cout), depth.reshape(*lead), w.reshape(*lead, M) if w is not None else None,
            ws.reshape(*lead) if ws is not None else None)


def resample(z_coarse, coarse_weights, u):
    """[BR,N] x3 -> fine z [BR,N]   (generators.py:489-499 + sample_pdf)"""
    BR, N = z_coarse.shape
    dev = z_coarse.device
    zc, wc, uu = _f32(z_coarse, dev), _f32(coarse_weights, dev), _f32(u, dev)
    zf = torch.empty((BR, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_resample(BR, N, _ptr(zc), _ptr(wc), _ptr(uu), _ptr(zf), _stream()))
    return zf


def sample_pdf(bins, weights, u):
    """bins [R,K+1], weights [R,K], u [R,Ns] -> samples [R,Ns]   (sample_pdf, reference shape)"""
    R, K = weights.shape
    Ns = u.shape[1]
    dev = bins.device
    b, w, uu = _f32(bins, dev), _f32(weights, dev), _f32(u, dev)
    out = torch.empty((R, Ns), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_sample_pdf(R, K, Ns, _ptr(b), _ptr(w), _ptr(uu), _ptr(out), _stream()))
    return out


def merge_composite(fine, coarse, z_fine, z_coarse, noise, opts, want_weights=True, want_wsum=True, want_z=True):
    """[BR,N,C] x2, [BR,N] x2 -> (rgb, depth, weights[BR,2N], wsum, z_sorted[BR,2N])   (generators.py:508-519)"""
    BR, N, Cc = fine.shape
    dev = fine.device
    f, c, zf, zc = _f32(fine, dev), _f32(coarse, dev), _f32(z_fine, dev), _f32(z_coarse, dev)
    nz = _f32(noise, dev).reshape(BR, 2 * N) if noise is not None else None
    pad = opts.fill_mode in (_lib.FILL["seg_padding_background"], _lib.FILL["eval_seg_padding_background"])
    Cout = Cc if pad else Cc - 1
    rgb = torch.empty((BR, Cout), dtype=torch.float32, device=dev)
    depth = torch.empty((BR,), dtype=torch.float32, device=dev)
    w = torch.empty((BR, 2 * N), dtype=torch.float32, device=dev) if want_weights else None
    ws = torch.empty((BR,), dtype=torch.float32, device=dev) if want_wsum else None
    zs = torch.empty((BR, 2 * N), dtype=torch.float32, device=dev) if want_z else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_merge_composite(BR, N, Cc, _ptr(f), _ptr(c), _ptr(zf), _ptr(zc), _ptr(nz), C.byref(opts),
                                                     _ptr(rgb), _ptr(depth), _ptr(w), _ptr(ws), _ptr(zs), _stream()))
    return rgb, depth, w, ws, zs


def sparse_select(d_coarse, d_fine, z_coarse, z_fine, origins, dirs, cap, want_dirs=True, images=None):
    """Selection step of the exact-sparsity backward (fenerf_sparse_select, include/fenerf.h): d_coarse / d_fine [B*R, N, C] (the merged
    composite's backward), z_coarse / z_fine [B*R, N], origins / dirs [B, R, 3], cap = slots per image (a multiple of 32).
    -> pts [B, cap, 3], rd [B, cap, 3] or None, d_sel [B, cap, C], counts int32 [B + 1] (kept samples per image, then the overflow flag);
    three launches, nothing waits.  images: None, or an int64 device tensor [B'] of image indices -- the call then handles those B' images
    (outputs [B', ...]) of the inputs' B.  d_fine = z_fine = None: one pass (no importance resampling)."""
    Bin, R, _ = origins.shape
    BR, N, Cc = d_coarse.shape
    assert BR == Bin * R and cap >= 1 and (d_fine is None) == (z_fine is None) and (d_fine is None or d_fine.shape == d_coarse.shape)
    assert images is None or (images.is_cuda and images.dtype == torch.int64 and images.is_contiguous() and images.dim() == 1)
    B = Bin if images is None else images.numel()
    dev = d_coarse.device
    dc, df, zc, zf, o, d = (_f32(t, dev) if t is not None else None for t in (d_coarse, d_fine, z_coarse, z_fine, origins, dirs))
    pts = torch.empty((B, cap, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((B, cap, 3), dtype=torch.float32, device=dev) if want_dirs else None
    d_sel = torch.empty((B, cap, Cc), dtype=torch.float32, device=dev)
    counts = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    nbytes = _lib.lib().fenerf_sparse_select_workspace_bytes(B, R * N)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_sparse_select(B, R, N, Cc, cap, _ptr(dc), _ptr(df), _ptr(zc), _ptr(zf), _ptr(o), _ptr(d),
                                                   C.c_void_p(images.data_ptr()) if images is not None else None, _ptr(pts), _ptr(rd), _ptr(d_sel), C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, _stream()))
    return pts, rd, d_sel, counts


def composite_backward(g_rgb, rows_a, z_a, opts, rows_b=None, z_b=None, noise=None, out_a=None, out_b=None):
    """Gradient of the final composite wrt its input rows.  Non-merge: rows_a [BR,M,C], z_a [BR,M] -> d_rows_a.
    Merge (rows_b given): fine rows_a / coarse rows_b [BR,N,C], z_a / z_b [BR,N] -> (d_fine, d_coarse).
    out_a / out_b: contiguous fp32 device buffers of the rows' shapes to write into (views of a larger tensor: no copy afterwards)."""
    BR, N, Cc = rows_a.shape
    dev = rows_a.device
    merge = rows_b is not None
    ra, za, g = _f32(rows_a, dev), _f32(z_a, dev), _f32(g_rgb, dev).reshape(BR, Cc - 1)
    rb, zb = (_f32(rows_b, dev), _f32(z_b, dev)) if merge else (None, None)
    nz = _f32(noise, dev).reshape(BR, -1) if noise is not None else None
    for o, r in ((out_a, ra), (out_b, rb)):
        assert o is None or (o.is_cuda and o.dtype == torch.float32 and o.is_contiguous() and o.numel() == r.numel())
    da = out_a if out_a is not None else torch.empty_like(ra)
    db = (out_b if out_b is not None else torch.empty_like(rb)) if merge else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().fenerf_composite_backward(BR, N, Cc, int(merge), _ptr(ra), _ptr(rb), _ptr(za), _ptr(zb), _ptr(nz),
                                                        C.byref(opts), _ptr(g), _ptr(da), _ptr(db), _stream()))
    return (da, db) if merge else da
