"""ctypes binding of libfenerf_hip.so (the C-ABI of include/fenerf.h).

There is NO fallback: if the shared library is missing or an entry point fails, we raise.
The library itself is torch-free; device pointers come from torch tensors' data_ptr() and the
stream from torch.cuda.current_stream().cuda_stream.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FENERF_LIB", os.path.join(_HERE, "libfenerf_hip.so"))   # FENERF_LIB: kernel A/B experiments only

ABI_VERSION = 2
MAX_GEO, MAX_COLOR, MAX_LABEL = 8, 4, 3

OK, E_INVALID, E_HIP, E_NOMEM, E_UNSUPPORTED, E_CLAMP_MODE = 0, -1, -2, -3, -4, -5
CLAMP = {"relu": 1, "softplus": 2}
FILL = {None: 0, "weight": 1, "seg_padding_background": 2, "eval_seg_padding_background": 3, "eval_white_back": 4}
FILL_COLORS = {"white": 1.0, "black": 0.0, "grey": 0.5, "light_grey": 0.81}

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


class FenerfModelDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("hidden_dim", C.c_int32), ("n_geo", C.c_int32), ("n_color", C.c_int32),
        ("n_label_layers", C.c_int32), ("output_dim", C.c_int32), ("grid_ch", C.c_int32),
        ("grid_d", C.c_int32), ("grid_h", C.c_int32), ("grid_w", C.c_int32), ("box_scale", C.c_float),
        ("geo_w", _fp * MAX_GEO), ("geo_b", _fp * MAX_GEO),
        ("color_w", _fp * MAX_COLOR), ("color_b", _fp * MAX_COLOR),
        ("label_w", _fp * MAX_LABEL), ("label_b", _fp * MAX_LABEL),
        ("sigma_w", _fp), ("sigma_b", _fp), ("rgb_w", _fp), ("rgb_b", _fp), ("grid", _fp),
        ("precision", C.c_int32), ("differentiable", C.c_int32), ("wgrad_bf16_min_points", C.c_int32),
    ]


class FenerfLocalMapDesc(C.Structure):
    _fields_ = [("latent_dim", C.c_int32), ("map_hidden", C.c_int32), ("w0", _fp), ("b0", _fp), ("w1", _fp), ("b1", _fp),
                ("w2", _fp), ("b2", _fp)]


MAP_MAX_LAYERS = 8


class FenerfMappingNet(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("z_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32),
                ("W", _vp * MAP_MAX_LAYERS), ("b", _vp * MAP_MAX_LAYERS)]


class FenerfSirenGrads(C.Structure):
    _fields_ = [("geo_w", _vp * MAX_GEO), ("geo_b", _vp * MAX_GEO), ("color_w", _vp * MAX_COLOR), ("color_b", _vp * MAX_COLOR),
                ("head_w", _vp), ("head_b", _vp), ("rgb_w", _vp), ("rgb_b", _vp),
                ("d_freq_geo", _vp), ("d_phase_geo", _vp), ("d_freq_app", _vp), ("d_phase_app", _vp)]


class FenerfRepackMaps(C.Structure):
    _fields_ = [("stream_f32", _vp), ("n_stream_f32", C.c_size_t), ("stream_h16", _vp), ("n_stream_h16", C.c_size_t),
                ("consts", _vp), ("n_consts", C.c_size_t), ("consts_tail", _vp), ("n_tail", C.c_size_t),
                ("bwd_f32", _vp), ("n_bwd_f32", C.c_size_t), ("bwd_b16", _vp), ("n_bwd_b16", C.c_size_t),
                ("row_off", _vp), ("row_len", _vp), ("row_film", _vp), ("n_rows", C.c_int32), ("scale_id", _vp)]


class FenerfCompositeOpts(C.Structure):
    _fields_ = [("clamp_mode", C.c_int32), ("noise_std", C.c_float), ("last_back", C.c_int32),
                ("white_back", C.c_int32), ("black_back", C.c_int32), ("fill_mode", C.c_int32),
                ("fill_value", C.c_float), ("fill_enabled", C.c_int32)]


class FenerfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fenerf error {code}: {msg}")
        self.code = code


_i, _i64, _sz = C.c_int, C.c_int64, C.c_size_t
_SIGS = {
    "fenerf_last_error": (C.c_char_p, []),
    "fenerf_abi_version": (_i, []),
    "fenerf_set_cu_budget": (_i, [_i]),
    "fenerf_set_render_fusion": (_i, [_i]),
    "fenerf_mapping_forward": (_i, [C.POINTER(FenerfMappingNet), _i, _vp, _vp, _vp, _vp]),
    "fenerf_mapping_workspace_floats": (_sz, [C.POINTER(FenerfMappingNet), _i]),
    "fenerf_mapping_backward": (_i, [C.POINTER(FenerfMappingNet), _i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "fenerf_label_head_workspace_floats": (_sz, [_i]),
    "fenerf_label_head_backward": (_i, [_i, _i, _i, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "fenerf_struct_size": (C.c_long, [C.c_char_p]),
    "fenerf_struct_field_offset": (C.c_long, [C.c_char_p, C.c_char_p]),
    "fenerf_struct_field_name": (C.c_char_p, [C.c_char_p, _i]),
    "fenerf_pack_weights_host": (_i, [C.POINTER(FenerfModelDesc), C.POINTER(_fp), C.POINTER(_sz), C.POINTER(_fp), C.POINTER(_sz)]),
    "fenerf_free_host": (None, [_vp]),
    "fenerf_model_create": (_i, [C.POINTER(FenerfModelDesc), C.POINTER(_vp)]),
    "fenerf_model_update": (_i, [_vp, C.POINTER(FenerfModelDesc), _vp]),
    "fenerf_model_destroy": (None, [_vp]),
    "fenerf_pack_backward_host": (_i, [C.POINTER(FenerfModelDesc), C.POINTER(_fp), C.POINTER(_sz)]),
    "fenerf_pack_index_map_f16": (_i, [C.POINTER(FenerfModelDesc), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(_sz)]),
    "fenerf_pack_backward_index_map_bf16": (_i, [C.POINTER(FenerfModelDesc), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(_sz)]),
    "fenerf_model_load_packed": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp]),
    "fenerf_model_repack": (_i, [_vp, _vp, _sz, C.POINTER(FenerfRepackMaps), _vp, _vp]),
    "fenerf_model_export_packed": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    "fenerf_film_workspace_bytes": (_sz, [_vp, _i]),
    "fenerf_siren_forward": (_i, [_vp, _i, _i64] + [_vp] * 9),
    "fenerf_film_workspace_bytes_pointwise": (_sz, [_vp, _i, _i64]),
    "fenerf_siren_forward_pointwise": (_i, [_vp, _i, _i64] + [_vp] * 9),
    "fenerf_siren_forward_rays": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i] + [_vp] * 7),
    "fenerf_ray_setup": (_i, [_i, _i, _i, C.c_float, C.c_float, C.c_float] + [_vp] * 9),
    "fenerf_siren_time_rays": (_i, [_vp, _i, _i, _i] + [_vp] * 9 + [_i, C.POINTER(C.c_float), _vp]),
    "fenerf_local_model_create": (_i, [C.POINTER(FenerfModelDesc), C.POINTER(FenerfLocalMapDesc), C.POINTER(_vp)]),
    "fenerf_local_model_destroy": (None, [_vp]),
    "fenerf_siren_forward_local": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fenerf_pack_local_host": (_i, [C.POINTER(FenerfModelDesc), C.POINTER(FenerfLocalMapDesc), C.POINTER(_fp), C.POINTER(_sz),
                               C.POINTER(_fp), C.POINTER(_sz)]),
    "fenerf_siren_backward_stream_bytes": (_i, [_vp, _i64, C.POINTER(C.c_double)]),
    "fenerf_siren_clock_probe": (_i, [_vp, _i, _i, _i] + [_vp] * 9 + [_i, C.POINTER(C.c_double), _vp]),
    "fenerf_siren_executed_flop_per_point": (C.c_double, [_vp]),
    "fenerf_phase_timing": (_i, [_i]),
    "fenerf_phase_times": (_i, [C.POINTER(C.c_double), C.POINTER(_i), _i]),
    "fenerf_phase_name": (C.c_char_p, [_i]),
    "fenerf_composite": (_i, [_i64, _i, _i, _vp, _vp, _vp, C.POINTER(FenerfCompositeOpts), _vp, _vp, _vp, _vp, _vp]),
    "fenerf_resample": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp]),
    "fenerf_sample_pdf": (_i, [_i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "fenerf_merge_composite": (_i, [_i64, _i, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(FenerfCompositeOpts), _vp, _vp, _vp, _vp, _vp, _vp]),
    "fenerf_siren_tape_floats": (_sz, [_vp, _i64]),
    "fenerf_siren_dtheta_floats": (_sz, [_vp, _i64]),
    "fenerf_siren_forward_save": (_i, [_vp, _i, _i64] + [_vp] * 11),
    "fenerf_siren_backward": (_i, [_vp, _i, _i64] + [_vp] * 11),
    "fenerf_siren_grad_workspace_bytes": (_sz, [_vp, _i, _i64]),
    "fenerf_siren_param_grads": (_i, [_vp, _i, _i64] + [_vp] * 11 + [C.POINTER(FenerfSirenGrads)] + [_vp] * 3),
    "fenerf_siren_film_sums_floats": (_sz, [_vp, _i, _i64]),
    "fenerf_siren_backward_film": (_i, [_vp, _i, _i64] + [_vp] * 10),
    "fenerf_siren_film_grads": (_i, [_vp, _i, _i64] + [_vp] * 5 + [C.POINTER(FenerfSirenGrads)] + [_vp] * 3),
    "fenerf_grid_backward": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "fenerf_siren_backward_fuses_grid": (_i, [_vp]),
    "fenerf_siren_backward_grid": (_i, [_vp, _i, _i64] + [_vp] * 13),
    "fenerf_grid_gradient_ncdhw": (_i, [_vp, _vp, _vp, _vp]),
    "fenerf_siren_input_grads": (_i, [_vp, _i, _i64] + [_vp] * 8 + [_i] + [_vp] * 4),
    # round 5: the same four calls for a tape in `tape_format` (FENERF_TAPE_F32 | FENERF_TAPE_U16)
    "fenerf_siren_tape_bytes": (_sz, [_vp, _i64, _i]),
    "fenerf_siren_forward_save_fmt": (_i, [_vp, _i, _i64] + [_vp] * 10 + [_i, _vp]),
    "fenerf_siren_backward_fmt": (_i, [_vp, _i, _i64] + [_vp] * 7 + [_i] + [_vp] * 4),
    "fenerf_siren_backward_grid_fmt": (_i, [_vp, _i, _i64] + [_vp] * 7 + [_i] + [_vp] * 6),
    "fenerf_siren_forward_save_pointwise": (_i, [_vp, _i, _i64] + [_vp] * 10),
    "fenerf_siren_backward_pointwise": (_i, [_vp, _i, _i64] + [_vp] * 9 + [_i, _vp]),
    "fenerf_siren_param_grads_pointwise": (_i, [_vp, _i, _i64] + [_vp] * 10 + [C.POINTER(FenerfSirenGrads)] + [_vp] * 2 + [_i, _vp]),
    "fenerf_siren_param_grads_fmt": (_i, [_vp, _i, _i64] + [_vp] * 9 + [_i, _vp, _vp, C.POINTER(FenerfSirenGrads), C.POINTER(FenerfSirenGrads)] + [_vp] * 3),
    "fenerf_siren_backward_stream_bytes_fmt": (_i, [_vp, _i64, _i, C.POINTER(C.c_double)]),
    # round 5: the differentiable hierarchical render as two calls (SURVEY 8b fenerf_render_backward)
    "fenerf_model_set_forward_mode": (_i, [_vp, _i]),
    "fenerf_render_save_bytes": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "fenerf_render_forward_save": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 10 + [C.POINTER(FenerfCompositeOpts), _vp, _vp, _vp, _sz, _i, _vp]),
    "fenerf_render_backward_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i, _i64, _i64]),
    "fenerf_render_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _i, _vp, _vp, C.POINTER(FenerfCompositeOpts), _vp, C.POINTER(FenerfSirenGrads), _vp,
                                    C.POINTER(FenerfSirenGrads), _i64, _i64, _vp, _sz, _vp]),
    "fenerf_render_backward_split_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i64, _i]),
    "fenerf_render_backward_stage": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp, _vp, C.POINTER(FenerfCompositeOpts), _vp, C.POINTER(FenerfSirenGrads), _vp,
                                          C.POINTER(FenerfSirenGrads), _i64, _vp, _sz, _vp]),
    "fenerf_sparse_select_workspace_bytes": (C.c_size_t, [_i, _i64]),
    "fenerf_sparse_select": (_i, [_i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "fenerf_composite_backward": (_i, [_i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(FenerfCompositeOpts), _vp, _vp, _vp, _vp]),
    "fenerf_render_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "fenerf_render_forward": (_i, [_vp, _i, _i, _i, _i, _i] + [_vp] * 10 + [C.POINTER(FenerfCompositeOpts)] + [_vp] * 4 + [_vp, _sz, _vp]),
}
EXPORTS = tuple(_SIGS)
N_PHASES = 17        # include/fenerf.h FENERF_N_PHASES
FUSION_AUTO, FUSION_OFF, FUSION_FORCE = 0, 1, 2      # include/fenerf.h fenerf_set_render_fusion
TAPE_F32, TAPE_U16, TAPE_F32_W = 0, 1, 2             # include/fenerf.h FENERF_TAPE_*

_lib = None


def lib():
    """Loads the shared library (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or make -C fenerf_amd/csrc).  fenerf_amd has no CPU / PyTorch fallback for the render path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export what include/fenerf.h declares
            fn.restype, fn.argtypes = res, args
        if l.fenerf_abi_version() != ABI_VERSION:
            raise RuntimeError("libfenerf_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc):
    if rc != OK:
        raise FenerfError(rc, lib().fenerf_last_error().decode())


def _as_f32(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a, a.ctypes.data_as(_fp)


PRECISION = {"f32": 0, "f16x3": 1}


def make_desc(sd, spec, precision="f32", differentiable=False, wgrad_bf16_min_points=0):
    """Builds a FenerfModelDesc from a reference-named state dict of numpy arrays.
    Returns (desc, keepalive) -- keepalive holds the host arrays the desc points into."""
    keep = []

    def P(name):
        a, p = _as_f32(sd[name])
        keep.append(a)
        return p

    d = FenerfModelDesc()
    d.abi_version = ABI_VERSION
    d.hidden_dim, d.n_geo, d.n_color = spec["hidden_dim"], spec["n_geo"], spec["n_color"]
    d.n_label_layers, d.output_dim, d.grid_ch = spec["n_label_layers"], spec["output_dim"], spec["grid_ch"]
    d.box_scale = 2 / 0.24
    d.precision = PRECISION[precision]
    d.differentiable = int(bool(differentiable))
    d.wgrad_bf16_min_points = int(wgrad_bf16_min_points)
    for i in range(spec["n_geo"]):
        d.geo_w[i], d.geo_b[i] = P(f"network.{i}.layer.weight"), P(f"network.{i}.layer.bias")
    if spec["kind"] == "spatial":
        d.color_w[0], d.color_b[0] = P("color_layer_sine.layer.weight"), P("color_layer_sine.layer.bias")
    else:
        for i in range(spec["n_color"]):
            d.color_w[i], d.color_b[i] = P(f"color_layer_sine.{i}.layer.weight"), P(f"color_layer_sine.{i}.layer.bias")
    for i in range(spec["n_label_layers"]):
        d.label_w[i], d.label_b[i] = P(f"label_layer_linear.{i}.weight"), P(f"label_layer_linear.{i}.bias")
    d.sigma_w, d.sigma_b = P("final_layer.weight"), P("final_layer.bias")
    d.rgb_w, d.rgb_b = P("color_layer_linear.0.weight"), P("color_layer_linear.0.bias")
    if spec["grid_ch"]:
        g = sd["spatial_embeddings"]
        d.grid_d, d.grid_h, d.grid_w = int(g.shape[2]), int(g.shape[3]), int(g.shape[4])
        d.grid = P("spatial_embeddings")
    return d, keep


def make_local_map_desc(mp):
    """mp = {'0.weight', '0.bias', '2.weight', '2.bias', '4.weight', '4.bias'} of CustomMappingNetwork(32, 256, 2 L H, n_blocks=1).network
    (numpy) -> (FenerfLocalMapDesc, keepalive)"""
    keep = []

    def P(name):
        a, p = _as_f32(mp[name])
        keep.append(a)
        return p

    d = FenerfLocalMapDesc()
    d.map_hidden, d.latent_dim = (int(v) for v in np.asarray(mp["0.weight"]).shape)
    d.w0, d.b0, d.w1, d.b1, d.w2, d.b2 = P("0.weight"), P("0.bias"), P("2.weight"), P("2.bias"), P("4.weight"), P("4.bias")
    return d, keep


def pack_local_host(sd, spec, mp):
    """(blob, consts) numpy copies of the SPATIALSIRENGRID stream (mapping network + SIREN interleaved; CPU only, layout tests)."""
    d, keep = make_desc(sd, spec, "f32")
    md, keep2 = make_local_map_desc(mp)
    blob, consts = _fp(), _fp()
    nb, nc = _sz(), _sz()
    check(lib().fenerf_pack_local_host(C.byref(d), C.byref(md), C.byref(blob), C.byref(nb), C.byref(consts), C.byref(nc)))
    try:
        b = np.ctypeslib.as_array(blob, shape=(nb.value,)).copy()
        c = np.ctypeslib.as_array(consts, shape=(nc.value,)).copy()
    finally:
        lib().fenerf_free_host(blob)
        lib().fenerf_free_host(consts)
    return b, c


def pack_index_map_f16(sd, spec):
    """int32 codes of the f16x3 ring stream for index-valued weights: (1 + flat index) | (is_lo << 30), 0 = padding."""
    d, keep = make_desc(sd, spec, "f16x3")
    m, n = C.POINTER(C.c_int32)(), _sz()
    check(lib().fenerf_pack_index_map_f16(C.byref(d), C.byref(m), C.byref(n)))
    try:
        return np.ctypeslib.as_array(m, shape=(n.value,)).copy()
    finally:
        lib().fenerf_free_host(m)


def pack_backward_index_map_bf16(sd, spec):
    """int32 index (1 + flat index, 0 = padding) of every bf16 half of an f16x3 model's backward ring for index-valued weights."""
    d, keep = make_desc(sd, spec, "f16x3", True)
    m, n = C.POINTER(C.c_int32)(), _sz()
    check(lib().fenerf_pack_backward_index_map_bf16(C.byref(d), C.byref(m), C.byref(n)))
    try:
        return np.ctypeslib.as_array(m, shape=(n.value,)).copy()
    finally:
        lib().fenerf_free_host(m)


def pack_backward_host(sd, spec, precision="f32"):
    """numpy copy of the backward-chain stream (CPU only; layout tests and the device-side packing index map)."""
    d, keep = make_desc(sd, spec, precision, True)
    blob, nb = _fp(), _sz()
    check(lib().fenerf_pack_backward_host(C.byref(d), C.byref(blob), C.byref(nb)))
    try:
        return np.ctypeslib.as_array(blob, shape=(nb.value,)).copy()
    finally:
        lib().fenerf_free_host(blob)


def pack_weights_host(sd, spec, precision="f32"):
    """(blob, consts) numpy copies of the packed streaming layout (CPU only; layout tests)."""
    d, keep = make_desc(sd, spec, precision)
    blob, consts = _fp(), _fp()
    nb, nc = _sz(), _sz()
    check(lib().fenerf_pack_weights_host(C.byref(d), C.byref(blob), C.byref(nb), C.byref(consts), C.byref(nc)))
    try:
        b = np.ctypeslib.as_array(blob, shape=(nb.value,)).copy()
        c = np.ctypeslib.as_array(consts, shape=(nc.value,)).copy()
    finally:
        lib().fenerf_free_host(blob)
        lib().fenerf_free_host(consts)
    return b, c


def composite_opts(clamp_mode, noise_std=0.0, last_back=False, white_back=False, black_back=False, fill_mode=None,
                   fill_color="black"):
    """kwargs of fancy_integration -> FenerfCompositeOpts.  Mirrors the reference's error behaviour:
    an unknown clamp_mode is a TypeError (volumetric_rendering.py:34 raises a str), 'debug'/'weight_debug'
    raise RuntimeError for 21-channel outputs (:54, :66)."""
    if clamp_mode not in CLAMP:
        raise TypeError("exceptions must derive from BaseException")  # what `raise "Need to choose clamp mode"` does
    if fill_mode in ("debug", "weight_debug"):
        raise RuntimeError("shape mismatch: value tensor of shape [22] cannot be broadcast to indexing result "
                           "(reference fill_mode=%r is only shape-valid for 22 colour channels)" % fill_mode)
    if fill_mode not in FILL:
        fill_mode = None  # unknown strings fall through every elif in the reference -> plain return
    o = FenerfCompositeOpts()
    o.clamp_mode = CLAMP[clamp_mode]
    o.noise_std = float(noise_std)
    o.last_back, o.white_back, o.black_back = int(bool(last_back)), int(bool(white_back)), int(bool(black_back))
    o.fill_mode = FILL[fill_mode]
    o.fill_enabled = int(fill_color in FILL_COLORS)
    o.fill_value = FILL_COLORS.get(fill_color, 0.0)
    return o
