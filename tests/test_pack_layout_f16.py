"""CPU check of the f16x3 weight packer + dataflow of fenerf_siren_f16.hip: numpy emulation of one wave
(v_mfma_f32_32x32x16_f16 lane maps, fp16 hi/lo splits, per-row power-of-two scales folded into the FiLM
frequencies) run on the blob the C packer produced, compared with the fp64 oracle."""
import numpy as np
import pytest

from fenerf_amd import _lib, procedural as proc
from oracle import fenerf_oracle as O
from test_pack_layout import H_, M_, PF, mfma


def mfma16(a8, b8, acc):
    """a8[l][t] = A[i=l&31][k=8(l>>5)+t], b8[l][t] = B[k=8(l>>5)+t][j=l&31]; acc[l][r] += D[row(r,l>>5)][l&31]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for t in range(8):
        A[M_, 8 * H_ + t] = a8[:, t]
        B[8 * H_ + t, M_] = b8[:, t]
    D = A @ B
    for r in range(16):
        acc[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * H_, M_]


def split16(v):
    v = np.asarray(v, dtype=np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def emulate_tile_f16(blob, consts, spec, pts, dirs, film, grid_cl):
    H, NB, KS = spec["hidden_dim"], spec["hidden_dim"] // 32, spec["hidden_dim"] // 16
    pad = lambda n: (n + PF - 1) // PF * PF
    EP = pad(2 * KS)
    has_grid = spec["grid_ch"] > 0
    C0_KS = KS + (2 if has_grid else 0) + 1
    C0_EP = pad(2 * C0_KS)
    n_geo, n_color, C = spec["n_geo"], spec["n_color"], spec["output_dim"]
    n_lab = C - 4
    L = n_geo + n_color
    l0 = blob[:NB * 256].reshape(NB, 64, 4).astype(np.float64)
    ring = blob[NB * 256:].view(np.float16).reshape(-1, 64, 8).astype(np.float64)
    cur = [0]

    def next_entry():
        e = ring[cur[0]]
        cur[0] += 1
        return e

    stage_begin = [0]

    def end_stage():   # every stage is padded to a whole number of ring revolutions (8 chunks x 8 entries)
        used = cur[0] - stage_begin[0]
        padded = (used + 63) // 64 * 64
        assert not ring[cur[0]:stage_begin[0] + padded].any()
        cur[0] = stage_begin[0] + padded
        stage_begin[0] = cur[0]

    fg = film["freq_geo"][0].astype(np.float32) * np.float32(15) + np.float32(30)
    fa = film["freq_app"][0].astype(np.float32) * np.float32(15) + np.float32(30)
    f_all = np.concatenate([fg, fa]).astype(np.float64).reshape(L, H)
    p_all = np.concatenate([film["phase_geo"][0], film["phase_app"][0]]).astype(np.float64).reshape(L, H)
    bias = consts[36:36 + L * H].astype(np.float64).reshape(L, H)
    inv = consts[36 + L * H:36 + 2 * L * H].astype(np.float64).reshape(L, H)
    head_inv = consts[36 + 2 * L * H:36 + 2 * L * H + 32].astype(np.float64)
    rgb_inv = consts[36 + 2 * L * H + 32:36 + 2 * L * H + 36].astype(np.float64)
    assert (inv[0] == 1).all()
    fp, pp = f_all / (2 * np.pi) * inv, (f_all * bias + p_all) / (2 * np.pi)

    p = pts[M_].astype(np.float64)
    d = dirs[M_].astype(np.float64)
    q = p * (2 / 0.24)
    slab_h = np.zeros((KS, 64, 8)); slab_l = np.zeros((KS, 64, 8))

    def film_store(acc, layer, nb):
        feat = 32 * nb + (np.arange(16) & 3)[None, :] + 8 * (np.arange(16) >> 2)[None, :] + 4 * H_[:, None]
        v = 16 * np.sin(2 * np.pi * (fp[layer][feat] * acc + pp[layer][feat]))
        for qq in range(2):
            hi, lo = split16(v[:, 8 * qq:8 * qq + 8])
            slab_h[2 * nb + qq], slab_l[2 * nb + qq] = hi, lo

    def kstep(acc, bh, bl):
        wh, wl = next_entry(), next_entry()
        mfma16(wh, bh, acc); mfma16(wh, bl, acc); mfma16(wl, bh, acc)

    def mfma_x(acc, xh, xl):
        for s in range(KS):
            kstep(acc, xh[s], xl[s])
        for _ in range(2 * KS, EP):
            next_entry()

    e = np.zeros((64, 16))
    if has_grid:
        Dg, Hg, Wg = grid_cl.shape[:3]
        ix, iy, iz = (q[:, 0] + 1) / 2 * (Wg - 1), (q[:, 1] + 1) / 2 * (Hg - 1), (q[:, 2] + 1) / 2 * (Dg - 1)
        x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
        for c in range(8):
            cz, cy, cx = c >> 2, (c >> 1) & 1, c & 1
            xi, yi, zi = x0 + cx, y0 + cy, z0 + cz
            wgt = (ix - x0 if cx else x0 + 1 - ix) * (iy - y0 if cy else y0 + 1 - iy) * (iz - z0 if cz else z0 + 1 - iz)
            ok = (xi >= 0) & (xi <= Wg - 1) & (yi >= 0) & (yi <= Hg - 1) & (zi >= 0) & (zi <= Dg - 1)
            for l in range(64):
                if ok[l]:
                    e[l] += grid_cl[int(zi[l]), int(yi[l]), int(xi[l]), 16 * H_[l]:16 * H_[l] + 16] * wgt[l]
    b0 = np.where(H_ == 1, q[:, 1], q[:, 0])
    b1 = np.where(H_ == 1, 0.0, q[:, 2])
    for nb in range(NB):
        acc = np.zeros((64, 16))
        mfma(l0[nb][:, 0], b0, acc)
        mfma(l0[nb][:, 1], b1, acc)
        film_store(acc, 0, nb)
    xh, xl = slab_h.copy(), slab_l.copy()
    for l in range(1, n_geo):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            mfma_x(acc, xh, xl)
            film_store(acc, l, nb)
        end_stage()
        xh, xl = slab_h.copy(), slab_l.copy()
    eh = [None, None]; el = [None, None]
    for j in range(2):
        eh[j], el[j] = split16(e[:, 8 * j:8 * j + 8] * 16)
    dv = np.zeros((64, 8)); dv[:, :3] = d * 16
    dh, dl = split16(dv)
    for nb in range(NB):
        acc = np.zeros((64, 16))
        for s in range(KS):
            kstep(acc, xh[s], xl[s])
        if has_grid:
            for j in range(2):
                kstep(acc, eh[j], el[j])
        kstep(acc, dh, dl)
        for _ in range(2 * C0_KS, C0_EP):
            next_entry()
        film_store(acc, n_geo, nb)
    end_stage()
    out = np.zeros((32, C))
    acc = np.zeros((64, 16))
    mfma_x(acc, xh, xl)
    end_stage()
    for r in range(16):
        row = (r & 3) + 8 * (r >> 2) + 4 * H_
        for l in range(64):
            if row[l] <= n_lab:
                ch = row[l] if row[l] < n_lab else C - 1
                out[M_[l], ch] = acc[l, r] * head_inv[row[l]] + consts[row[l]]
    xh, xl = slab_h.copy(), slab_l.copy()
    for c in range(1, n_color):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            mfma_x(acc, xh, xl)
            film_store(acc, n_geo + c, nb)
        end_stage()
        xh, xl = slab_h.copy(), slab_l.copy()
    acc = np.zeros((64, 16))
    mfma_x(acc, xh, xl)
    end_stage()
    for r in range(3):
        for l in range(32):
            out[l, C - 4 + r] = 1 / (1 + np.exp(-(acc[l, r] * rgb_inv[r] + consts[32 + r])))
    DPF_CH = 6 * 8   # the first 6 chunks are replicated after the end (prefetch of the next tile never wraps)
    assert cur[0] + DPF_CH == ring.shape[0], "f16 stream must be consumed exactly (+ the replicated head)"
    assert np.array_equal(ring[cur[0]:], ring[:DPF_CH])
    return out


@pytest.mark.parametrize("kind,H,grid", [("texture", 32, 5), ("texture", 64, 4), ("baseline", 32, 0), ("spatial", 32, 0),
                                         ("texture", 256, 6), ("texture", 96, 5), ("baseline", 192, 0)])
def test_packer_f16x3_and_kpermutation(kind, H, grid):
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=12, sigma_gain=500.0, with_mapping=False)
    blob, consts = _lib.pack_weights_host(sd, spec, "f16x3")
    rng = np.random.default_rng(1)
    pts = rng.uniform(-0.13, 0.13, (32, 3)).astype(np.float32)
    dirs = rng.normal(size=(32, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, 1, seed=12)
    if kind == "spatial":
        film["freq_app"] = proc.normal("film.freq_app", (1, H), 0.4, 12)
        film["phase_app"] = proc.normal("film.phase_app", (1, H), 0.4, 12)
    grid_cl = np.ascontiguousarray(sd["spatial_embeddings"][0].transpose(1, 2, 3, 0)).astype(np.float64) if grid else None
    got = emulate_tile_f16(blob, consts, spec, pts, dirs, film, grid_cl)
    if kind == "spatial":
        fg = np.concatenate([film["freq_geo"], film["freq_app"]], -1)
        pg = np.concatenate([film["phase_geo"], film["phase_app"]], -1)
        ref = O.siren_forward(sd, spec, pts[None], dirs[None], fg, pg, dtype=np.float64)[0]
    else:
        ref = O.siren_forward(sd, spec, pts[None], dirs[None], film["freq_geo"], film["phase_geo"], film["freq_app"],
                              film["phase_app"], dtype=np.float64)[0]
    # error-compensated fp16 products: fp32-class agreement with the fp64 oracle
    np.testing.assert_allclose(got[..., :-1], ref[..., :-1], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(got[..., -1], ref[..., -1], atol=1e-5 * max(1.0, np.abs(ref[..., -1]).max()), rtol=1e-5)
