"""CPU check of the dataflow of fenerf_siren_f16w.hip (16-point waves on v_mfma_f32_16x16x32_f16) on the blob the C packer
produced for the 32x32x16 kernel: numpy emulation of ONE wave's tile through the LDS-DMA re-tiling permutation, the 16x16x32
operand / result lane maps, the accumulator -> next-layer B operand identity, the grid / view-direction k-steps of colour
layer 0, the head row maps and the chunk / stage walk (padding, replicated head), compared with the fp64 oracle.
The MFMA lane maps themselves are re-checked on hardware by tests/test_gpu_parity.py."""
import numpy as np
import pytest

from fenerf_amd import _lib, procedural as proc
from oracle import fenerf_oracle as O
from test_pack_layout_f16 import split16

LANE = np.arange(64)
N_, G_ = LANE & 15, LANE >> 4          # MFMA column (point) / lane group
CH, NSLOT, DPF = 8, 8, 6


def row_of(rt, i):
    """weight row (inside a 32-row block) that A-operand row i of row tile rt carries (fenerf_siren_f16w.hip header)"""
    gi, r = i >> 2, i & 3
    return 16 * (gi >> 1) + 4 * (gi & 1) + 8 * rt + r


def mfma16w(a8, b8, acc):
    """v_mfma_f32_16x16x32_f16: a8[l][t] = A[i=l&15][k=8(l>>4)+t], b8[l][t] = B[k=8(l>>4)+t][n=l&15]; acc[l][r] += D[4(l>>4)+r][l&15]"""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for t in range(8):
        A[N_, 8 * G_ + t] = a8[:, t]
        B[8 * G_ + t, N_] = b8[:, t]
    D = A @ B
    for r in range(4):
        acc[:, r] += D[4 * G_ + r, N_]


def mfma32w(a, b, acc):
    """v_mfma_f32_16x16x4_f32: a[l] = A[i=l&15][k=l>>4], b[l] = B[k=l>>4][n=l&15]"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[N_, G_] = a
    B[G_, N_] = b
    D = A @ B
    for r in range(4):
        acc[:, r] += D[4 * G_ + r, N_]


def dma_operand(chunk, spl, rt, hl):
    """What wave (spl, rt, hl) of the workgroup DMAs out of an 8-entry chunk [8][64 pieces][8 halves]: lane (i, kg) <- old entry
    2 (2 spl + (kg >> 1)) + hl, piece 32 (kg & 1) + row(rt, i)."""
    out = np.zeros((64, 8))
    for l in range(64):
        i, kg = l & 15, l >> 4
        out[l] = chunk[2 * (2 * spl + (kg >> 1)) + hl][32 * (kg & 1) + row_of(rt, i)]
    return out


def emulate_tile_f16w(blob, consts, spec, pts, dirs, film, grid_cl):
    H, NB, KS = spec["hidden_dim"], spec["hidden_dim"] // 32, spec["hidden_dim"] // 32
    has_grid = spec["grid_ch"] > 0
    QB = (KS + 1) // 2
    C0_KS = KS + (1 if has_grid else 0) + 1
    C0_QB = (2 * (2 * KS + (2 if has_grid else 0) + 1) + CH - 1) // CH
    pad_stage = lambda chunks: (chunks + NSLOT - 1) // NSLOT * NSLOT
    n_geo, n_color, C = spec["n_geo"], spec["n_color"], spec["output_dim"]
    n_lab = C - 4
    L = n_geo + n_color
    l0 = blob[:NB * 256].astype(np.float64)
    ring = blob[NB * 256:].view(np.float16).reshape(-1, CH, 64, 8).astype(np.float64)     # [chunk][old entry][piece][slot]
    cur = [0]

    fg = film["freq_geo"][0].astype(np.float32) * np.float32(15) + np.float32(30)
    fa = film["freq_app"][0].astype(np.float32) * np.float32(15) + np.float32(30)
    f_all = np.concatenate([fg, fa]).astype(np.float64).reshape(L, H)
    p_all = np.concatenate([film["phase_geo"][0], film["phase_app"][0]]).astype(np.float64).reshape(L, H)
    bias = consts[36:36 + L * H].astype(np.float64).reshape(L, H)
    inv = consts[36 + L * H:36 + 2 * L * H].astype(np.float64).reshape(L, H)
    head_inv = consts[36 + 2 * L * H:36 + 2 * L * H + 32].astype(np.float64)
    rgb_inv = consts[36 + 2 * L * H + 32:36 + 2 * L * H + 36].astype(np.float64)
    fp, pp = f_all / (2 * np.pi) * inv, (f_all * bias + p_all) / (2 * np.pi)

    p = pts[N_].astype(np.float64)
    d = dirs[N_].astype(np.float64)
    q = p * (2 / 0.24)
    feat0 = 16 * (G_ >> 1) + 4 * (G_ & 1)

    def film_pack(acc, layer, nb):
        """acc[rt][l][r] -> (hi, lo) B operand of k32-step nb of the next layer: slot 4 rt + r"""
        hi = np.zeros((64, 8)); lo = np.zeros((64, 8))
        for rt in range(2):
            feat = 32 * nb + feat0[:, None] + 8 * rt + np.arange(4)[None, :]
            v = 16 * np.sin(2 * np.pi * (fp[layer][feat] * acc[rt] + pp[layer][feat]))
            hi[:, 4 * rt:4 * rt + 4], lo[:, 4 * rt:4 * rt + 4] = split16(v)
        return hi, lo

    def body(bops, n_chunks):
        """one n-block body: n_chunks chunks, k32-step sp multiplies bops[sp] (None = padding k-step) -> acc[rt]"""
        acc = [np.zeros((64, 4)), np.zeros((64, 4))]
        for qc in range(n_chunks):
            chunk = ring[cur[0]]
            cur[0] += 1
            for spl in range(2):
                sp = 2 * qc + spl
                if sp < len(bops) and bops[sp] is not None:
                    bh, bl = bops[sp]
                    for rt in range(2):
                        ahi, alo = dma_operand(chunk, spl, rt, 0), dma_operand(chunk, spl, rt, 1)
                        mfma16w(alo, bh, acc[rt]); mfma16w(ahi, bl, acc[rt]); mfma16w(ahi, bh, acc[rt])
                else:       # padding k-steps carry zero weights
                    for e in range(4 * spl, 4 * spl + 4):
                        assert not chunk[e].any()
        return acc

    stage_begin = [0]

    def end_stage():
        used = cur[0] - stage_begin[0]
        padded = pad_stage(used)
        assert not ring[cur[0]:stage_begin[0] + padded].any()
        cur[0] = stage_begin[0] + padded
        stage_begin[0] = cur[0]

    # grid features: lane (n, g) holds channels 16 (g & 1) + 8 (g >> 1) .. + 7 of its point
    e = np.zeros((64, 8))
    if has_grid:
        Dg, Hg, Wg = grid_cl.shape[:3]
        ix, iy, iz = (q[:, 0] + 1) / 2 * (Wg - 1), (q[:, 1] + 1) / 2 * (Hg - 1), (q[:, 2] + 1) / 2 * (Dg - 1)
        x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
        ch0 = 16 * (G_ & 1) + 8 * (G_ >> 1)
        for c in range(8):
            cz, cy, cx = c >> 2, (c >> 1) & 1, c & 1
            xi, yi, zi = x0 + cx, y0 + cy, z0 + cz
            wgt = (ix - x0 if cx else x0 + 1 - ix) * (iy - y0 if cy else y0 + 1 - iy) * (iz - z0 if cz else z0 + 1 - iz)
            ok = (xi >= 0) & (xi <= Wg - 1) & (yi >= 0) & (yi <= Hg - 1) & (zi >= 0) & (zi <= Dg - 1)
            for l in range(64):
                if ok[l]:
                    e[l] += grid_cl[int(zi[l]), int(yi[l]), int(xi[l]), ch0[l]:ch0[l] + 8] * wgt[l]
    # layer 0 on the fp32 MFMA (16x16x4: k = x, y, z, 0); weights from the 32x32x2 layer-0 block
    b0 = np.select([G_ == 0, G_ == 1, G_ == 2], [q[:, 0], q[:, 1], q[:, 2]], 0.0)
    x = []
    for nb in range(NB):
        acc = [np.zeros((64, 4)), np.zeros((64, 4))]
        for rt in range(2):
            a0 = np.array([l0[nb * 256 + ((G_[l] & 1) * 32 + row_of(rt, N_[l])) * 4 + (G_[l] >> 1)] for l in range(64)])
            mfma32w(a0, b0, acc[rt])
        x.append(film_pack(acc, 0, nb))
    assert (inv[0] == 1).all()
    for l in range(1, n_geo):
        y = [film_pack(body(x, QB), l, nb) for nb in range(NB)]
        end_stage()
        x = y
    # colour layer 0: [x | grid | dir]
    eh, el = split16(e * 16)
    dv = np.zeros((64, 8)); dv[G_ == 0, :3] = d[G_ == 0] * 16
    dh, dl = split16(dv)
    bops = list(x) + ([(eh, el)] if has_grid else []) + [(dh, dl)]
    assert len(bops) == C0_KS
    y = [film_pack(body(bops, C0_QB), n_geo, nb) for nb in range(NB)]
    end_stage()
    out = np.zeros((16, C))
    acc = body(x, QB)
    end_stage()
    for rt in range(2):
        for r in range(4):
            row = feat0 + 8 * rt + r
            for l in range(64):
                if row[l] <= n_lab:
                    ch = row[l] if row[l] < n_lab else C - 1
                    out[N_[l], ch] = acc[rt][l, r] * head_inv[row[l]] + consts[row[l]]
    x = y
    for c in range(1, n_color):
        y = [film_pack(body(x, QB), n_geo + c, nb) for nb in range(NB)]
        end_stage()
        x = y
    acc = body(x, QB)
    end_stage()
    for r in range(3):
        for l in range(16):      # lane group 0
            out[l, C - 4 + r] = 1 / (1 + np.exp(-(acc[0][l, r] * rgb_inv[r] + consts[32 + r])))
    assert cur[0] + DPF == ring.shape[0], "f16 stream must be consumed exactly (+ the replicated head)"
    assert np.array_equal(ring[cur[0]:], ring[:DPF])
    return out


@pytest.mark.parametrize("kind,H,grid", [("texture", 32, 5), ("texture", 64, 4), ("baseline", 32, 0), ("spatial", 32, 0),
                                         ("texture", 128, 4), ("texture", 256, 6), ("texture", 96, 5), ("baseline", 192, 0)])
def test_f16w_dataflow_on_the_packed_stream(kind, H, grid):
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=12, sigma_gain=500.0, with_mapping=False)
    blob, consts = _lib.pack_weights_host(sd, spec, "f16x3")
    rng = np.random.default_rng(1)
    pts = rng.uniform(-0.13, 0.13, (16, 3)).astype(np.float32)
    dirs = rng.normal(size=(16, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, 1, seed=12)
    if kind == "spatial":
        film["freq_app"] = proc.normal("film.freq_app", (1, H), 0.4, 12)
        film["phase_app"] = proc.normal("film.phase_app", (1, H), 0.4, 12)
    grid_cl = np.ascontiguousarray(sd["spatial_embeddings"][0].transpose(1, 2, 3, 0)).astype(np.float64) if grid else None
    got = emulate_tile_f16w(blob, consts, spec, pts, dirs, film, grid_cl)
    if kind == "spatial":
        fg = np.concatenate([film["freq_geo"], film["freq_app"]], -1)
        pg = np.concatenate([film["phase_geo"], film["phase_app"]], -1)
        ref = O.siren_forward(sd, spec, pts[None], dirs[None], fg, pg, dtype=np.float64)[0]
    else:
        ref = O.siren_forward(sd, spec, pts[None], dirs[None], film["freq_geo"], film["phase_geo"], film["freq_app"],
                              film["phase_app"], dtype=np.float64)[0]
    np.testing.assert_allclose(got[..., :-1], ref[..., :-1], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(got[..., -1], ref[..., -1], atol=1e-5 * max(1.0, np.abs(ref[..., -1]).max()), rtol=1e-5)
