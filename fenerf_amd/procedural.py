"""Procedural (integer-hash) weights and inputs.

There is no network, hence no pretrained FENeRF checkpoint; every golden vector,
parity test and benchmark uses weights computed by this file so that the exact
same numbers can be re-created on the GPU box without shipping tensors.  No
torch RNG is involved: value i of tensor `name` is a splitmix64 hash of
(seed, crc(name), i) mapped to a uniform / normal variate.

Value ranges follow the reference initialisers (siren/siren.py:104-110
`frequency_init(25)`, :333-338 `modified_first_sine_init`, :82-95 mapping
network, :1546 grid N(0, 0.1^2)); `sigma_gain` optionally scales the density
head so alphas are O(0.1..1) ("trained-like") instead of ~1e-3 at init.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_u01(name: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in [0, 1), a pure function of (name, seed, stream, index)."""
    tag = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) ^ (tag << np.uint64(17)) ^ np.uint64(stream)],
                                dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base) & _M64
    z = _splitmix64(idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(name, shape, lo, hi, seed=0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * hash_u01(name, n, seed)).astype(np.float32).reshape(shape)


def normal(name, shape, std=1.0, seed=0):
    n = int(np.prod(shape))
    u1 = hash_u01(name, n, seed, stream=1)
    u2 = hash_u01(name, n, seed, stream=2)
    g = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (std * g).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------
# model specifications (mirror the reference class hierarchy we support)
# ---------------------------------------------------------------------------
def model_spec(kind="texture", hidden_dim=256, grid_size=96, output_dim=22, z_dim=256, map_hidden=256):
    """kind: 'texture'  -> TextureEmbeddingPiGAN*SEMANTICDISENTANGLE   (siren.py:1451)
             'baseline' -> SIRENBASELINESEMANTICDISENTANGLE             (siren.py:1163)
             'spatial'  -> SPATIALSIRENBASELINE (single latent, rgb+sigma, siren.py:189)"""
    H = hidden_dim
    if kind == "texture":
        return dict(kind=kind, hidden_dim=H, n_geo=8, n_color=3, grid_ch=32, grid_size=grid_size,
                    n_label_layers=3, output_dim=output_dim, color_in=H + 32 + 3, z_dim=z_dim, map_hidden=map_hidden)
    if kind == "baseline":
        return dict(kind=kind, hidden_dim=H, n_geo=8, n_color=3, grid_ch=0, grid_size=0,
                    n_label_layers=2, output_dim=output_dim, color_in=H + 3, z_dim=z_dim, map_hidden=map_hidden)
    if kind == "spatial":
        return dict(kind=kind, hidden_dim=H, n_geo=8, n_color=1, grid_ch=0, grid_size=0,
                    n_label_layers=0, output_dim=4, color_in=H + 3, z_dim=z_dim, map_hidden=map_hidden)
    raise ValueError(kind)


def _linear(sd, prefix, out_dim, in_dim, wbound, seed):
    sd[prefix + ".weight"] = uniform(prefix + ".weight", (out_dim, in_dim), -wbound, wbound, seed)
    b = 1.0 / np.sqrt(in_dim)  # nn.Linear default bias init
    sd[prefix + ".bias"] = uniform(prefix + ".bias", (out_dim,), -b, b, seed)


def _mapping(sd, prefix, z_dim, hid, out_dim, seed, n_blocks=3):
    dims = [(hid, z_dim)] + [(hid, hid)] * n_blocks + [(out_dim, hid)]
    for j, (o, i) in enumerate(dims):
        name = f"{prefix}.network.{2 * j}"
        std = np.sqrt(2.0 / (1 + 0.2 ** 2)) / np.sqrt(i)  # kaiming_normal_(a=0.2, fan_in)
        w = normal(name + ".weight", (o, i), std, seed)
        if j == len(dims) - 1:
            w = w * np.float32(0.25)
        sd[name + ".weight"] = w
        b = 1.0 / np.sqrt(i)
        sd[name + ".bias"] = uniform(name + ".bias", (o,), -b, b, seed)


def make_state_dict(spec, seed=0, sigma_gain=1.0, with_mapping=True):
    """Reference-named state_dict (numpy fp32) for `spec`; loadable with load_state_dict()
    into the reference siren class and into fenerf_amd.siren.siren classes alike."""
    H = spec["hidden_dim"]
    sd = {}
    fi = lambda n_in: float(np.sqrt(6.0 / n_in) / 25.0)
    _linear(sd, "network.0.layer", H, 3, 1.0 / 3.0, seed)
    for i in range(1, spec["n_geo"]):
        _linear(sd, f"network.{i}.layer", H, H, fi(H), seed)
    _linear(sd, "final_layer", 1, H, fi(H), seed)
    sd["final_layer.weight"] = sd["final_layer.weight"] * np.float32(sigma_gain)
    sd["final_layer.bias"] = sd["final_layer.bias"] * np.float32(sigma_gain)
    if spec["kind"] == "spatial":
        _linear(sd, "color_layer_sine.layer", H, spec["color_in"], fi(spec["color_in"]), seed)
    else:
        _linear(sd, "color_layer_sine.0.layer", H, spec["color_in"], fi(spec["color_in"]), seed)
        for i in range(1, spec["n_color"]):
            _linear(sd, f"color_layer_sine.{i}.layer", H, H, fi(H), seed)
    _linear(sd, "color_layer_linear.0", 3, H, fi(H), seed)
    n_lab = spec["output_dim"] - 4
    for i in range(spec["n_label_layers"]):
        last = i == spec["n_label_layers"] - 1
        _linear(sd, f"label_layer_linear.{i}", n_lab if last else H, H, fi(H), seed)
    if spec["grid_ch"]:
        g = spec["grid_size"]
        sd["spatial_embeddings"] = normal("spatial_embeddings", (1, spec["grid_ch"], g, g, g), 0.1, seed)
    if with_mapping:
        if spec["kind"] == "spatial":
            _mapping(sd, "mapping_network", spec["z_dim"], spec["map_hidden"], (spec["n_geo"] + 1) * H * 2, seed)
        else:
            _mapping(sd, "geo_mapping_network", spec["z_dim"], spec["map_hidden"], spec["n_geo"] * H * 2, seed)
            _mapping(sd, "app_mapping_network", spec["z_dim"], spec["map_hidden"], spec["n_color"] * H * 2, seed)
    return sd


def film_params(spec, batch, seed=0, scale=1.0, phase_rev=0.0, freq0_gain=1.0):
    """Raw (pre '*15+30') frequencies / phase shifts in the range the mapping nets emit
    at init (f = 15 f_raw + 30 in ~[13, 49]).

    Beyond init -- what inversion (inverse_render_double_semantic.py:370-410: Adam on unconstrained offsets) or a trained
    checkpoint may produce, and what torch.sin in the reference's FiLMLayer (siren.py:113-123) takes in its stride:
    `phase_rev` > 0 adds a uniform phase shift of up to +-phase_rev REVOLUTIONS (2 pi phase_rev radians) to every FiLM layer;
    `freq0_gain` multiplies the effective frequency 15 f + 30 of the first FiLM layer (whose pre-activation W0 x + b is O(1),
    so its sine argument reaches ~ freq0_gain * 6 revolutions; the deeper layers' pre-activations are O(0.1))."""
    H = spec["hidden_dim"]
    ng, nc = spec["n_geo"] * H, spec["n_color"] * H
    out = {}
    for name, n in (("freq_geo", ng), ("phase_geo", ng), ("freq_app", nc), ("phase_app", nc)):
        out[name] = normal(f"film.{name}", (batch, n), 0.4 * scale, seed)
    if phase_rev:
        for name, n in (("phase_geo", ng), ("phase_app", nc)):
            out[name] = (out[name] + uniform(f"film.{name}.rev", (batch, n), -2.0 * np.pi * phase_rev, 2.0 * np.pi * phase_rev, seed)).astype(np.float32)
    if freq0_gain != 1.0:
        f0 = out["freq_geo"][:, :H].astype(np.float64)
        out["freq_geo"][:, :H] = (((15.0 * f0 + 30.0) * freq0_gain - 30.0) / 15.0).astype(np.float32)
    return out


def checksum(sd) -> float:
    """Order-independent-ish checksum of a state dict (pins procedural weights in fixtures)."""
    tot = 0.0
    for k in sorted(sd):
        a = np.asarray(sd[k], dtype=np.float64).ravel()
        w = np.cos(np.arange(a.size, dtype=np.float64) * 0.37 + (zlib.crc32(k.encode()) % 1000))
        tot += float(np.dot(a, w))
    return tot


def latent_grid_state(shapes, seed=0):
    """Procedural parameters for the StyleGAN2-style latent-grid generator of SPATIALSIRENGRID (siren/latent_grid.py): `shapes` =
    {parameter name: shape} of its state dict (buffers excluded).  Magnitudes follow the reference initialisers (layers.py): N(0, 1)
    weights (the lr_mul = 0.01 mapping layers store weight / lr_mul), modulation biases around 1, small activation / output biases."""
    out = {}
    for name, shape in shapes.items():
        tag = "slg." + name
        if name.endswith("modulation.bias"):
            out[name] = (1.0 + normal(tag, shape, 0.1, seed)).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = normal(tag, shape, 0.1, seed)
        elif name.startswith("mapping_network") and name.endswith("weight"):
            out[name] = normal(tag, shape, 100.0, seed)
        else:
            out[name] = normal(tag, shape, 1.0, seed)
    return out
