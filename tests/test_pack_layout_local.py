"""CPU check of the SPATIALSIRENGRID stream (fenerf_pack.cpp pack_local_weights) + the dataflow of fenerf_siren_local.hip: a numpy
emulation of one wave -- same stream walk (mapping network bodies, then per FiLM layer and n-block [frequency | phase | layer] bodies),
same v_mfma_f32_32x32x2_f32 lane maps, same ring consumption order, same epilogue algebra -- is run on the blob the C packer produced
and compared with the plain statement of siren.py:440-477 (mapping network per point -> FiLM SIREN).  Pins packer + dataflow without a
GPU; the GPU test (tests/test_gpu_parity.py::test_spatial_siren_grid_vs_reference) then checks the kernel against the reference."""
import numpy as np
import pytest

from fenerf_amd import _lib, procedural as proc

PF = 8
LANE = np.arange(64)
M_, H_ = LANE & 31, LANE >> 5


def mfma(a, b, acc):
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[M_, H_] = a
    B[H_, M_] = b
    D = A @ B
    for r in range(16):
        acc[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * H_, M_]


def lrelu(v):
    return np.where(v > 0, v, 0.2 * v)


def local_weights(spec, seed=3):
    H, L = spec["hidden_dim"], spec["n_geo"] + spec["n_color"]
    mp = {"0.weight": proc.normal("lm.0.w", (256, 32), 0.25, seed), "0.bias": proc.normal("lm.0.b", (256,), 0.1, seed),
          "2.weight": proc.normal("lm.2.w", (256, 256), 0.09, seed), "2.bias": proc.normal("lm.2.b", (256,), 0.1, seed),
          "4.weight": proc.normal("lm.4.w", (2 * L * H, 256), 0.02, seed), "4.bias": proc.normal("lm.4.b", (2 * L * H,), 0.1, seed)}
    return mp


def reference(sd, spec, mp, pts, dirs, lat):
    """siren.py:455-477 in float64: mapping network per point, frequencies * 15 + 30, FiLM layers, sigma / colour / rgb heads."""
    H, n_geo = spec["hidden_dim"], spec["n_geo"]
    L = n_geo + spec["n_color"]
    f64 = lambda a: np.asarray(a, np.float64)
    h = lrelu(f64(lat) @ f64(mp["0.weight"]).T + f64(mp["0.bias"]))
    h = lrelu(h @ f64(mp["2.weight"]).T + f64(mp["2.bias"]))
    o = h @ f64(mp["4.weight"]).T + f64(mp["4.bias"])
    freq, phase = o[:, :L * H] * 15 + 30, o[:, L * H:]
    x = f64(pts) * (2 / 0.24)
    for l in range(n_geo):
        x = np.sin(freq[:, l * H:(l + 1) * H] * (x @ f64(sd[f"network.{l}.layer.weight"]).T + f64(sd[f"network.{l}.layer.bias"])) + phase[:, l * H:(l + 1) * H])
    sigma = x @ f64(sd["final_layer.weight"]).T + f64(sd["final_layer.bias"])
    c = np.concatenate([f64(dirs), x], -1)
    c = np.sin(freq[:, -H:] * (c @ f64(sd["color_layer_sine.layer.weight"]).T + f64(sd["color_layer_sine.layer.bias"])) + phase[:, -H:])
    rgb = 1 / (1 + np.exp(-(c @ f64(sd["color_layer_linear.0.weight"]).T + f64(sd["color_layer_linear.0.bias"]))))
    return np.concatenate([rgb, sigma], -1)


def emulate_tile(blob, consts, spec, pts, dirs, lat):
    H, NB, KGX = spec["hidden_dim"], spec["hidden_dim"] // 32, spec["hidden_dim"] // 8
    MH, NBM, KGM = 256, 8, 32
    pad = lambda n: (n + PF - 1) // PF * PF
    KGXP, KGCP = pad(KGX), pad(KGX + 1)
    n_geo, n_color = spec["n_geo"], spec["n_color"]
    L = n_geo + n_color
    entries = blob.reshape(-1, 64, 4).astype(np.float64)
    cur = [0]

    def next_entry():
        e = entries[cur[0]]
        cur[0] += 1
        return e

    cst = consts.astype(np.float64)
    b0, b1 = cst[:MH], cst[MH:2 * MH]
    b2f, b2p = cst[2 * MH:2 * MH + L * H].reshape(L, H), cst[2 * MH + L * H:2 * MH + 2 * L * H].reshape(L, H)
    bl = cst[2 * MH + 2 * L * H:2 * MH + 3 * L * H].reshape(L, H)
    bh = cst[2 * MH + 3 * L * H:]
    q = pts[M_].astype(np.float64) * (2 / 0.24)
    d = dirs[M_].astype(np.float64)
    lv = np.stack([lat[M_, 2 * s + H_] for s in range(16)], 1).astype(np.float64)    # [64 lanes][16 k-steps]
    slab = np.zeros((MH // 8, 64, 4))
    feat = lambda nb, j: 32 * nb + 8 * j + 4 * H_[:, None] + np.arange(4)[None, :]

    def load_act(width):
        return slab[:width // 8].transpose(1, 0, 2).reshape(64, width // 2).copy()

    def mfma_x(acc, act, n_real, n_pad):
        for kg in range(n_pad):
            w = next_entry()
            if kg < n_real:
                for i in range(4):
                    mfma(w[:, i], act[:, 4 * kg + i], acc)

    for nb in range(NBM):                                  # mapping layer 0
        acc = np.zeros((64, 16))
        mfma_x(acc, lv, 4, PF)
        for j in range(4):
            slab[nb * 4 + j] = lrelu(acc[:, 4 * j:4 * j + 4] + b0[feat(nb, j)])
    hid = load_act(MH)
    for nb in range(NBM):                                  # mapping layer 1
        acc = np.zeros((64, 16))
        mfma_x(acc, hid, KGM, KGM)
        for j in range(4):
            slab[nb * 4 + j] = lrelu(acc[:, 4 * j:4 * j + 4] + b1[feat(nb, j)])
    hid = load_act(MH)

    def film3(aF, aP, aZ, l, nb):
        for j in range(4):
            ft = feat(nb, j)
            f = (aF[:, 4 * j:4 * j + 4] + b2f[l][ft]) * 15 + 30
            theta = f * (aZ[:, 4 * j:4 * j + 4] + bl[l][ft]) + aP[:, 4 * j:4 * j + 4] + b2p[l][ft]
            slab[nb * 4 + j] = np.sin(theta)

    c0 = np.where(H_ == 1, q[:, 1], q[:, 0])
    c1 = np.where(H_ == 1, 0.0, q[:, 2])
    for nb in range(NB):                                   # FiLM layer 0
        aF, aP, aZ = np.zeros((64, 16)), np.zeros((64, 16)), np.zeros((64, 16))
        mfma_x(aF, hid, KGM, KGM)
        mfma_x(aP, hid, KGM, KGM)
        for kg in range(PF):
            w = next_entry()
            if kg == 0:
                mfma(w[:, 0], c0, aZ)
                mfma(w[:, 1], c1, aZ)
        film3(aF, aP, aZ, 0, nb)
    act = load_act(H)
    for l in range(1, n_geo):
        for nb in range(NB):
            aF, aP, aZ = np.zeros((64, 16)), np.zeros((64, 16)), np.zeros((64, 16))
            mfma_x(aF, hid, KGM, KGM)
            mfma_x(aP, hid, KGM, KGM)
            mfma_x(aZ, act, KGX, KGXP)
            film3(aF, aP, aZ, l, nb)
        act = load_act(H)
    out = np.zeros((32, 4))
    acc = np.zeros((64, 16))
    mfma_x(acc, act, KGX, KGXP)
    out[:, 3] = acc[:32, 0] + bh[0]                        # lanes of half 0, register 0 = row 0
    d0 = np.where(H_ == 1, d[:, 1], d[:, 0])
    d1 = np.where(H_ == 1, 0.0, d[:, 2])
    for c in range(n_color):
        l = n_geo + c
        for nb in range(NB):
            aF, aP, aZ = np.zeros((64, 16)), np.zeros((64, 16)), np.zeros((64, 16))
            mfma_x(aF, hid, KGM, KGM)
            mfma_x(aP, hid, KGM, KGM)
            if c == 0:
                for kg in range(KGCP):
                    w = next_entry()
                    if kg < KGX:
                        for i in range(4):
                            mfma(w[:, i], act[:, 4 * kg + i], aZ)
                    elif kg == KGX:
                        mfma(w[:, 0], d0, aZ)
                        mfma(w[:, 1], d1, aZ)
            else:
                mfma_x(aZ, act, KGX, KGXP)
            film3(aF, aP, aZ, l, nb)
        act = load_act(H)
    acc = np.zeros((64, 16))
    mfma_x(acc, act, KGX, KGXP)
    for r in range(3):
        out[:, r] = 1 / (1 + np.exp(-(acc[:32, r] + bh[4 + r])))
    assert cur[0] + PF == entries.shape[0], "the walk consumes the whole stream up to the tail pad"
    return out


@pytest.mark.parametrize("H,n_geo", [(32, 3), (64, 8), (256, 2), (96, 3)])
def test_local_stream_walk_matches_the_plain_statement(H, n_geo):
    spec = dict(proc.model_spec("spatial", hidden_dim=H, grid_size=0, output_dim=4), n_geo=n_geo)
    sd = proc.make_state_dict(spec, seed=5, sigma_gain=20.0, with_mapping=False)
    mp = local_weights(spec)
    blob, consts = _lib.pack_local_host(sd, spec, mp)
    rng = np.random.default_rng(2)
    pts = rng.uniform(-0.12, 0.12, (32, 3)).astype(np.float32)
    dirs = rng.normal(size=(32, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    lat = rng.normal(size=(32, 32)).astype(np.float32)
    got = emulate_tile(blob, consts, spec, pts, dirs, lat)
    ref = reference(sd, spec, mp, pts, dirs, lat)
    err = np.abs(got - ref).max()
    print(f"local stream H={H} n_geo={n_geo}: {blob.size * 4 / 1e6:.2f} MB, emulated wave vs plain statement max|err| {err:.2e}")
    assert err <= 1e-9


def test_local_packer_rejects_what_the_kernel_does_not_evaluate():
    spec = proc.model_spec("texture", hidden_dim=32, grid_size=4)
    sd = proc.make_state_dict(spec, seed=5, with_mapping=False)
    with pytest.raises(_lib.FenerfError):
        _lib.pack_local_host(sd, spec, local_weights(spec))          # feature grid / label head: not SPATIALSIRENGRID
    spec = dict(proc.model_spec("spatial", hidden_dim=32, grid_size=0, output_dim=4))
    sd = proc.make_state_dict(spec, seed=5, with_mapping=False)
    mp = local_weights(spec)
    mp["0.weight"] = mp["0.weight"][:, :16]
    with pytest.raises(_lib.FenerfError):
        _lib.pack_local_host(sd, spec, mp)                           # latent width other than 32
