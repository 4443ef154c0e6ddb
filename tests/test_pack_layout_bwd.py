"""CPU check of the backward-chain weight streams (fenerf_pack.cpp::pack_weights_bwd / pack_weights_bwd16) + the dataflow
of fenerf_siren_bwd.hip / the 32x32x16 reading of the bf16 stream (fenerf_siren_bwd16w.hip re-tiles it, tests/test_pack_layout_bwd16w.py): a numpy emulation of one wave -- same stream walk, same v_mfma_f32_32x32x2_f32 lane maps, same
stage order (rgb head^T, colour layers, the colour-layer-0 stage with head^T and the grid-feature body, trunk) -- run
on the blob the C packer produced and compared with dL/dtheta of every FiLM layer from torch fp64 autograd.
Both packings: exact fp32 rows, and the power-of-two row-scaled rows of FENERF_PREC_F16X3 models (dz' = dz / s)."""
import numpy as np
import pytest
import torch

from fenerf_amd import _lib, procedural as proc
from oracle import fenerf_oracle_grad as OG
from test_pack_layout import H_, M_, PF, mfma


def emulate_chain(blob, spec, theta, f_true, row_scale, d_out, out):
    """theta [L][H][32] (reference values), f_true [L][H], row_scale [L][H] (s_i of the packing; ones for fp32),
    d_out / out [32][C] -> (dtheta [L][H][32], d_e [32 points][32 channels])."""
    H, NB, KGX = spec["hidden_dim"], spec["hidden_dim"] // 32, spec["hidden_dim"] // 8
    pad = lambda n: (n + PF - 1) // PF * PF
    KGXP = pad(KGX)
    C0_KG = KGX + 4
    C0_KGP = pad(C0_KG)
    n_geo, n_color, C = spec["n_geo"], spec["n_color"], spec["output_dim"]
    n_lab, L = C - 4, spec["n_geo"] + spec["n_color"]
    entries = blob.reshape(-1, 64, 4).astype(np.float64)
    cur = [NB]                       # ring cursor; the rgb-head^T block occupies entries [0, NB)

    def next_entry():
        e = entries[cur[0]]
        cur[0] += 1
        return e

    dtheta = np.zeros((L, H, 32))
    slab = np.zeros((H // 8, 64, 4))

    def store(acc, layer, nb):       # acc[l][r] = dL/dx of feature 32nb + (r&3) + 8(r>>2) + 4h at point l&31
        for r in range(16):
            feat = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * H_
            dt = acc[:, r] * np.cos(theta[layer][feat, M_])
            dtheta[layer][feat, M_] = dt
            # the kernel multiplies by f'' * 2 pi = f / s (f16x3) or f (fp32): dz' = dz / s
            slab[nb * 4 + (r >> 2)][:, r & 3] = dt * f_true[layer][feat] / row_scale[layer][feat]

    def load_act():
        return slab.transpose(1, 0, 2).reshape(64, H // 2).copy()

    def mfma_x(acc, act):
        for kg in range(KGXP):
            w = next_entry()
            if kg < KGX:
                for i in range(4):
                    mfma(w[:, i], act[:, 4 * kg + i], acc)

    # rgb head
    s = out[M_, C - 4:C - 1]
    dpre = d_out[M_, C - 4:C - 1] * s * (1 - s)
    b0 = np.where(H_ == 1, dpre[:, 1], dpre[:, 0])
    b1 = np.where(H_ == 1, 0.0, dpre[:, 2])
    for nb in range(NB):
        w = entries[nb]
        acc = np.zeros((64, 16))
        mfma(w[:, 0], b0, acc)
        mfma(w[:, 1], b1, acc)
        store(acc, L - 1, nb)
    act = load_act()
    for l in range(L - 1, n_geo, -1):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            mfma_x(acc, act)
            store(acc, l - 1, nb)
        act = load_act()
    # colour layer 0 + heads (lane-half h multiplies head row 16h + s)
    dh = np.zeros((64, 16))
    for s_ in range(16):
        row = 16 * H_ + s_
        ch = np.where(row < n_lab, row, np.where(row == n_lab, C - 1, -1))
        dh[:, s_] = np.where(ch >= 0, d_out[M_, np.maximum(ch, 0)], 0.0)
    for nb in range(NB):
        acc = np.zeros((64, 16))
        for kg in range(C0_KGP):
            w = next_entry()
            if kg < KGX:
                for i in range(4):
                    mfma(w[:, i], act[:, 4 * kg + i], acc)
            elif kg < C0_KG:
                q = kg - KGX
                for i in range(4):
                    mfma(w[:, i], dh[:, 4 * q + i], acc)
        store(acc, n_geo - 1, nb)
    d_e = np.zeros((32, 32))
    if spec["grid_ch"]:
        acc = np.zeros((64, 16))
        mfma_x(acc, act)
        for r in range(16):
            d_e[M_, (r & 3) + 8 * (r >> 2) + 4 * H_] = acc[:, r]
    act = load_act()
    for l in range(n_geo - 1, 0, -1):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            mfma_x(acc, act)
            store(acc, l - 1, nb)
        act = load_act()
    assert cur[0] + PF == entries.shape[0], "backward stream must be consumed exactly (+ the PF tail pad)"
    return dtheta, d_e


PF16 = 8


def mfma16(a, b, acc):
    """v_mfma_f32_32x32x16_bf16: a[l][t] = A[i=l&31][k=8(l>>5)+t], b[l][t] = B[k=8(l>>5)+t][j=l&31], acc as mfma()."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for t in range(8):
        A[M_, 8 * H_ + t] = a[:, t]
        B[8 * H_ + t, M_] = b[:, t]
    D = A @ B
    for r in range(16):
        acc[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * H_, M_]


def emulate_chain16(blob, spec, theta, f_true, row_scale, d_out, out):
    """The bf16x3 chain (the stream of fenerf_siren_bwd16w.hip, read entry by entry) on its stream: fp32 rgb-head block, then [hi entry, lo entry] per k-step.
    The emulation multiplies (hi + lo) with exact dz: it checks layout and dataflow, not the split arithmetic."""
    H, NB, KS = spec["hidden_dim"], spec["hidden_dim"] // 32, spec["hidden_dim"] // 16
    pad = lambda n: (n + PF16 - 1) // PF16 * PF16
    EP, C0_EP = pad(2 * KS), pad(2 * (KS + 2))
    n_geo, n_color, C = spec["n_geo"], spec["n_color"], spec["output_dim"]
    n_lab, L = C - 4, spec["n_geo"] + spec["n_color"]
    head = blob[:NB * 256].reshape(NB, 64, 4).astype(np.float64)
    ring = (blob[NB * 256:].view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64).reshape(-1, 64, 8)
    cur = [0]

    def next_kstep():
        w = ring[cur[0]] + ring[cur[0] + 1]
        cur[0] += 2
        return w

    dtheta = np.zeros((L, H, 32))
    slab = np.zeros((KS, 64, 8))

    def store(acc, layer, nb):
        for r in range(16):
            feat = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * H_
            dt = acc[:, r] * np.cos(theta[layer][feat, M_])
            dtheta[layer][feat, M_] = dt
            slab[2 * nb + (r >> 3)][:, r & 7] = dt * f_true[layer][feat] / row_scale[layer][feat]

    def body(acc, act, ep, extra=None):
        for ks in range(ep // 2):
            w = next_kstep()
            if ks < KS:
                mfma16(w, act[ks], acc)
            elif extra is not None and ks - KS < len(extra):
                mfma16(w, extra[ks - KS], acc)

    s = out[M_, C - 4:C - 1]
    dpre = d_out[M_, C - 4:C - 1] * s * (1 - s)
    b0 = np.where(H_ == 1, dpre[:, 1], dpre[:, 0])
    b1 = np.where(H_ == 1, 0.0, dpre[:, 2])
    for nb in range(NB):
        acc = np.zeros((64, 16))
        mfma(head[nb][:, 0], b0, acc)
        mfma(head[nb][:, 1], b1, acc)
        store(acc, L - 1, nb)
    act = slab.copy()
    for l in range(L - 1, n_geo, -1):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            body(acc, act, EP)
            store(acc, l - 1, nb)
        act = slab.copy()
    dh = np.zeros((2, 64, 8))
    for s_ in range(2):
        for t in range(8):
            row = 16 * s_ + 8 * H_ + t
            ch = np.where(row < n_lab, row, np.where(row == n_lab, C - 1, -1))
            dh[s_][:, t] = np.where(ch >= 0, d_out[M_, np.maximum(ch, 0)], 0.0)
    for nb in range(NB):
        acc = np.zeros((64, 16))
        body(acc, act, C0_EP, dh)
        store(acc, n_geo - 1, nb)
    d_e = np.zeros((32, 32))
    if spec["grid_ch"]:
        acc = np.zeros((64, 16))
        body(acc, act, EP)
        for r in range(16):
            d_e[M_, (r & 3) + 8 * (r >> 2) + 4 * H_] = acc[:, r]
    act = slab.copy()
    for l in range(n_geo - 1, 0, -1):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            body(acc, act, EP)
            store(acc, l - 1, nb)
        act = slab.copy()
    assert cur[0] + PF16 == ring.shape[0], "bf16 backward stream must be consumed exactly (+ the tail pad)"
    return dtheta, d_e


def chain_case(kind, H, grid, precision):
    """One 32-point tile: the packed backward stream + what torch fp64 autograd says dL/dtheta of every FiLM layer is.
    -> dict(spec, sd, blob, theta [L][H][32], ref [L][H][32], f_true, row_scale [L][H], d_out, out [32][C])"""
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=13, sigma_gain=5.0, with_mapping=False)
    blob = _lib.pack_backward_host(sd, spec, precision)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-0.11, 0.11, (1, 32, 3))
    dirs = rng.normal(size=(1, 32, 3)); dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, 1, seed=13)
    if kind == "spatial":
        film["freq_app"] = proc.normal("film.freq_app", (1, H), 0.4, 13)
        film["phase_app"] = proc.normal("film.phase_app", (1, H), 0.4, 13)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    sd64 = {k: t(v) for k, v in sd.items()}
    if grid:
        sd64["spatial_embeddings"].requires_grad_(True)
    taps = []
    fl = {k: t(v).requires_grad_(True) for k, v in film.items()}      # makes every theta part of the autograd graph
    out = OG.siren_forward(sd64, spec, t(pts), t(dirs), fl["freq_geo"], fl["phase_geo"], fl["freq_app"], fl["phase_app"], taps)
    C, L = spec["output_dim"], spec["n_geo"] + spec["n_color"]
    d_out = rng.normal(size=(32, C))
    (out[0] * t(d_out)).sum().backward()
    theta = np.stack([tp.detach().numpy()[0].T for tp in taps])               # [L][H][32]
    ref = np.stack([tp.grad.numpy()[0].T for tp in taps])
    f_true = np.concatenate([np.float32(film["freq_geo"][0]) * np.float32(15) + np.float32(30),
                             np.float32(film["freq_app"][0]) * np.float32(15) + np.float32(30)]).astype(np.float64).reshape(L, H)
    row_scale = np.ones((L, H))
    if precision == "f16x3":              # s_i = 2^e_i * 16 with max|row| * 2^e_i in [0.5, 1)  (row_scales() of the packer)
        names = [f"network.{i}.layer.weight" for i in range(spec["n_geo"])] + \
                (["color_layer_sine.layer.weight"] if kind == "spatial" else [f"color_layer_sine.{i}.layer.weight" for i in range(spec["n_color"])])
        for l, n in enumerate(names):
            if l == 0:
                continue
            m = np.abs(sd[n].astype(np.float32)).max(1)
            _, ex = np.frexp(m)
            row_scale[l] = np.where(m > 0, np.ldexp(1.0, -ex), 1.0) * 16
    return dict(spec=spec, sd=sd, blob=blob, theta=theta, ref=ref, f_true=f_true, row_scale=row_scale, d_out=d_out, out=out.detach().numpy()[0])


def check_chain(case, got, d_e, precision, grid):
    spec, sd, ref, f_true = case["spec"], case["sd"], case["ref"], case["f_true"]
    # the fp32 stream holds fp32 values (the fp64-folded label head is rounded once): agreement to fp32 resolution;
    # the bf16 stream holds hi + lo = 16+ significant bits of each weight
    tol = 2e-5 if precision == "f16x3" else 5e-7
    np.testing.assert_allclose(got, ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=1e-5 if precision == "f32" else 1e-4)
    if grid:   # d(grid features): check through the grid gradient's total (sum over voxels per channel = sum_p d_e[p][c] * weights ...)
        # the sampled features enter colour layer 0 linearly: d_e[p] = W_c0[:, 3:35]^T dz_{n_geo}[p]
        W = sd["color_layer_sine.0.layer.weight"].astype(np.float64)[:, 3:35]
        dz = ref[spec["n_geo"]] * f_true[spec["n_geo"]][:, None]                 # [H][32]
        np.testing.assert_allclose(d_e, (W.T @ dz).T, atol=tol * max(1.0, np.abs(dz).max()), rtol=1e-5 if precision == "f32" else 1e-4)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("kind,H,grid", [("texture", 32, 5), ("baseline", 64, 0), ("spatial", 32, 0), ("texture", 96, 5), ("baseline", 192, 0)])
def test_backward_stream_and_chain_dataflow(kind, H, grid, precision):
    case = chain_case(kind, H, grid, precision)
    emu = emulate_chain16 if precision == "f16x3" else emulate_chain
    got, d_e = emu(case["blob"], case["spec"], case["theta"], case["f_true"], case["row_scale"], case["d_out"], case["out"])
    check_chain(case, got, d_e, precision, grid)
