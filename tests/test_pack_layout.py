"""CPU check of the host weight packer (fenerf_pack.cpp) + the K-permutation scheme of the SIREN kernel.

A numpy emulation of one wave of fenerf_siren.hip -- same stream walk, same MFMA 32x32x2 operand / result
lane maps (cdna_hip_programming.md §3), same ring consumption order -- is run on the blob the C packer
produced and compared with the oracle.  This pins packer + dataflow without a GPU; the MFMA lane maps
themselves are re-checked on hardware by tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

from fenerf_amd import _lib, procedural as proc
from oracle import fenerf_oracle as O

PF = 8
LANE = np.arange(64)
M_, H_ = LANE & 31, LANE >> 5


def mfma(a, b, acc):
    """v_mfma_f32_32x32x2_f32: a[l] = A[i=l&31][k=l>>5], b[l] = B[k=l>>5][j=l&31], acc[l][r] = D[row(r,l>>5)][l&31]."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[M_, H_] = a
    B[H_, M_] = b
    D = A @ B
    for r in range(16):
        acc[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * H_, M_]


def sin2pi(t):
    return np.sin(2 * np.pi * t)


def emulate_tile(blob, consts, spec, pts, dirs, film, grid_cl):
    H, NB, KGX = spec["hidden_dim"], spec["hidden_dim"] // 32, spec["hidden_dim"] // 8
    pad = lambda n: (n + PF - 1) // PF * PF
    KGXP = pad(KGX)
    has_grid = spec["grid_ch"] > 0
    C0_KG = KGX + (4 if has_grid else 0) + 1
    C0_KGP = pad(C0_KG)
    n_geo, n_color, C = spec["n_geo"], spec["n_color"], spec["output_dim"]
    n_lab = C - 4
    L = n_geo + n_color
    entries = blob.reshape(-1, 64, 4).astype(np.float64)
    cur = [NB]  # ring cursor (entry index); layer-0 block occupies entries [0, NB)

    def next_entry():
        e = entries[cur[0]]
        cur[0] += 1
        return e

    # FiLM pre-pass (film_prep_kernel), single image
    fg = film["freq_geo"][0].astype(np.float32) * np.float32(15) + np.float32(30)
    fa = film["freq_app"][0].astype(np.float32) * np.float32(15) + np.float32(30)
    f_all = np.concatenate([fg, fa]).astype(np.float64).reshape(L, H)
    p_all = np.concatenate([film["phase_geo"][0], film["phase_app"][0]]).astype(np.float64).reshape(L, H)
    bias = consts[36:36 + L * H].astype(np.float64).reshape(L, H)
    fp, pp = f_all / (2 * np.pi), (f_all * bias + p_all) / (2 * np.pi)

    p = pts[M_].astype(np.float64)
    d = dirs[M_].astype(np.float64)
    q = p * (2 / 0.24)

    slab = np.zeros((H // 8, 64, 4))

    def film_store(acc, layer, nb):
        for j in range(4):
            feat = 32 * nb + 8 * j + 4 * H_[:, None] + np.arange(4)[None, :]
            slab[nb * 4 + j] = sin2pi(fp[layer][feat] * acc[:, 4 * j:4 * j + 4] + pp[layer][feat])

    def load_act():
        return slab.transpose(1, 0, 2).reshape(64, H // 2).copy()   # in[l][4g+i] = slab[g][l][i]

    def mfma_x(acc, act, n_real, n_pad):
        for kg in range(n_pad):
            w = next_entry()
            if kg < n_real:
                for i in range(4):
                    mfma(w[:, i], act[:, 4 * kg + i], acc)

    # grid features: half h blends channels 16h..16h+15
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

    # layer 0
    b0 = np.where(H_ == 1, q[:, 1], q[:, 0])
    b1 = np.where(H_ == 1, 0.0, q[:, 2])
    for nb in range(NB):
        w = entries[nb]
        acc = np.zeros((64, 16))
        mfma(w[:, 0], b0, acc)
        mfma(w[:, 1], b1, acc)
        film_store(acc, 0, nb)
    act = load_act()
    for l in range(1, n_geo):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            mfma_x(acc, act, KGX, KGXP)
            film_store(acc, l, nb)
        act = load_act()
    # colour layer 0
    bd0 = np.where(H_ == 1, d[:, 1], d[:, 0])
    bd1 = np.where(H_ == 1, 0.0, d[:, 2])
    for nb in range(NB):
        acc = np.zeros((64, 16))
        for kg in range(C0_KGP):
            w = next_entry()
            if kg < KGX:
                for i in range(4):
                    mfma(w[:, i], act[:, 4 * kg + i], acc)
            elif has_grid and kg < KGX + 4:
                qq = kg - KGX
                for i in range(4):
                    mfma(w[:, i], e[:, 4 * qq + i], acc)
            elif kg == C0_KG - 1:
                mfma(w[:, 0], bd0, acc)
                mfma(w[:, 1], bd1, acc)
        film_store(acc, n_geo, nb)
    out = np.zeros((32, C))
    acc = np.zeros((64, 16))
    mfma_x(acc, act, KGX, KGXP)
    for r in range(16):
        row = (r & 3) + 8 * (r >> 2) + 4 * H_
        for l in range(64):
            if row[l] <= n_lab:
                ch = row[l] if row[l] < n_lab else C - 1
                out[M_[l], ch] = acc[l, r] + consts[row[l]]
    act = load_act()
    for c in range(1, n_color):
        for nb in range(NB):
            acc = np.zeros((64, 16))
            mfma_x(acc, act, KGX, KGXP)
            film_store(acc, n_geo + c, nb)
        act = load_act()
    acc = np.zeros((64, 16))
    mfma_x(acc, act, KGX, KGXP)
    for r in range(3):
        for l in range(32):
            out[l, C - 4 + r] = 1 / (1 + np.exp(-(acc[l, r] + consts[32 + r])))
    assert cur[0] + PF == entries.shape[0], "stream must be consumed exactly (+ the PF tail pad)"
    return out


@pytest.mark.parametrize("kind,H,grid", [("texture", 32, 5), ("texture", 64, 4), ("baseline", 32, 0), ("spatial", 32, 0),
                                         ("texture", 256, 6), ("texture", 96, 5), ("baseline", 192, 0)])
def test_packer_and_kpermutation(kind, H, grid):
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=11, sigma_gain=3.0, with_mapping=False)
    blob, consts = _lib.pack_weights_host(sd, spec)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.13, 0.13, (32, 3)).astype(np.float32)   # a few points fall outside the grid box
    dirs = rng.normal(size=(32, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    film = proc.film_params(spec, 1, seed=11)
    if kind == "spatial":   # single latent: the colour layer uses the 9th slice of the same mapping output
        film["freq_app"] = proc.normal("film.freq_app", (1, H), 0.4, 11)
        film["phase_app"] = proc.normal("film.phase_app", (1, H), 0.4, 11)
    grid_cl = np.ascontiguousarray(sd["spatial_embeddings"][0].transpose(1, 2, 3, 0)).astype(np.float64) if grid else None
    got = emulate_tile(blob, consts, spec, pts, dirs, film, grid_cl)
    if kind == "spatial":
        fg = np.concatenate([film["freq_geo"], film["freq_app"]], -1)
        pg = np.concatenate([film["phase_geo"], film["phase_app"]], -1)
        ref = O.siren_forward(sd, spec, pts[None], dirs[None], fg, pg, dtype=np.float64)[0]
    else:
        ref = O.siren_forward(sd, spec, pts[None], dirs[None], film["freq_geo"], film["phase_geo"], film["freq_app"],
                              film["phase_app"], dtype=np.float64)[0]
    np.testing.assert_allclose(got, ref, atol=2e-6, rtol=1e-6)


def test_library_exports_every_declared_symbol():
    """include/fenerf.h <-> libfenerf_hip.so: every declared entry point is exported (no compute calls)."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(_lib.__file__), "..", "include", "fenerf.h")).read()
    declared = set(re.findall(r"\b(fenerf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name)
    assert l.fenerf_abi_version() == _lib.ABI_VERSION


def test_desc_validation_errors():
    spec = proc.model_spec("texture", hidden_dim=32, grid_size=4, z_dim=8)
    sd = proc.make_state_dict(spec, seed=1, with_mapping=False)
    bad = dict(spec, hidden_dim=48)
    with pytest.raises(_lib.FenerfError) as ei:
        _lib.pack_weights_host(sd, bad)
    assert ei.value.code == _lib.E_UNSUPPORTED
    with pytest.raises(TypeError):
        _lib.composite_opts(None)
    with pytest.raises(RuntimeError):
        _lib.composite_opts("relu", fill_mode="debug")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_device_packing_reproduces_the_host_streams(precision):
    """Training re-packs on the GPU (NativeModel.load_from_device): a gather through index maps obtained by packing
    index-valued weights -- for f16x3 after scaling rows by their power-of-two scale and splitting into fp16 hi / lo.  Run
    here on the CPU, the same torch code must reproduce the host packer's forward stream, consts and backward-chain stream
    (bitwise, except what depends on the label-head fold, which the device does in fp32 and the host in fp64)."""
    import torch
    from fenerf_amd import native
    for kind, H, grid in [("texture", 64, 4), ("baseline", 32, 0), ("spatial", 32, 0)]:
        spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
        sd = proc.make_state_dict(spec, seed=1, sigma_gain=10.0, with_mapping=False)
        nm = object.__new__(native.NativeModel)            # no GPU here: only the torch-side packing is exercised
        nm.spec, nm.differentiable, nm.device, nm.precision, nm._maps, nm._h = dict(spec), True, torch.device("cpu"), precision, None, None
        stream, consts, bwd, _ = nm._pack_on_device({k: torch.from_numpy(v) for k, v in sd.items()})
        blob, hconsts = _lib.pack_weights_host(sd, spec, precision)
        hbwd = _lib.pack_backward_host(sd, spec, precision)
        assert stream.numel() == blob.size and consts.numel() == hconsts.size and bwd.numel() == hbwd.size
        fold = spec["n_label_layers"] > 1
        if precision == "f32":
            d = np.abs(stream.numpy() - blob)
            assert d.max() <= (1e-7 if fold else 0) and (d != 0).mean() <= 0.05
        else:
            diff = stream.numpy().view(np.uint16) != blob.view(np.uint16)
            assert diff.mean() <= (0.05 if fold else 0.0)        # only label-head rows (lo halves of a fp32- vs fp64-folded matrix)
            n0 = (H // 32) * 512     # halves of the fp32 layer-0 block
            assert not diff[:n0].any()
            err = np.abs(stream.numpy().view(np.float16)[n0:].astype(np.float64) - blob.view(np.float16)[n0:].astype(np.float64))
            assert err.max() <= 1e-3
        np.testing.assert_allclose(consts.numpy(), hconsts, rtol=0, atol=1e-7)
        if precision == "f32":
            np.testing.assert_allclose(bwd.numpy(), hbwd, rtol=1e-6, atol=1e-9)
        else:       # fp32 rgb-head block, then bf16 (hi, lo) entries of the row-scaled transposed weights
            nh = (H // 32) * 256
            np.testing.assert_array_equal(bwd.numpy()[:nh], hbwd[:nh])
            bf = lambda a: (a[nh:].view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
            dv, hv = bf(bwd.numpy()), bf(hbwd)
            assert (dv != hv).mean() <= (0.05 if fold else 0.0)
            assert np.abs(dv - hv).max() <= 1e-3 * max(1.0, np.abs(hv).max())


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("kind,H,grid", [("texture", 40, 4), ("baseline", 100, 0), ("spatial", 72, 0)])
def test_padded_hidden_width_packs_like_the_padded_state(kind, H, grid, precision):
    """Round 6: a hidden width between the instantiated ones runs zero-padded at the next one (native.padded_hidden_dim).  Host side
    (`_pad_state`: what fenerf_model_create / _update receive) and device side (`_flat_params` inside load_from_device) must produce the
    SAME streams: the device packing of the module's H-wide tensors == the host packer on the padded state dict, with the bounds of
    test_device_packing_reproduces_the_host_streams; and the padded entries of the host state are exact zeros in the right places."""
    import torch
    from fenerf_amd import native
    Hp = native.padded_hidden_dim(H)
    assert Hp > H
    spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
    sd = proc.make_state_dict(spec, seed=2, sigma_gain=10.0, with_mapping=False)
    nm = object.__new__(native.NativeModel)
    nm.logical_H = H
    nm.spec, nm.differentiable, nm.device, nm.precision, nm._maps, nm._h = dict(spec, hidden_dim=Hp), True, torch.device("cpu"), precision, None, None
    assert nm.padded
    psd = nm._pad_state(sd)
    G = spec["grid_ch"]
    for k, v in sd.items():
        if k == "spatial_embeddings":
            assert psd[k] is v or np.array_equal(psd[k], v)
            continue
        q = psd[k]
        if v.ndim == 1:
            assert np.array_equal(q[:v.shape[0]], v) and not q[v.shape[0]:].any()
        elif k.startswith("color_layer_sine.0.") or k == "color_layer_sine.layer.weight":      # [dirs | grid | x]: hidden columns are the trailing ones
            assert q.shape == (Hp, 3 + G + Hp) and np.array_equal(q[:H, :3 + G + H], v) and not q[H:].any() and not q[:, 3 + G + H:].any()
        else:
            assert np.array_equal(q[:v.shape[0], :v.shape[1]], v) and not q[v.shape[0]:].any() and not q[:, v.shape[1]:].any()
    pspec = dict(spec, hidden_dim=Hp)
    stream, consts, bwd, _ = nm._pack_on_device({k: torch.from_numpy(v) for k, v in sd.items()})
    blob, hconsts = _lib.pack_weights_host(psd, pspec, precision)
    hbwd = _lib.pack_backward_host(psd, pspec, precision)
    assert stream.numel() == blob.size and consts.numel() == hconsts.size and bwd.numel() == hbwd.size
    fold = spec["n_label_layers"] > 1
    if precision == "f32":
        d = np.abs(stream.numpy() - blob)
        assert d.max() <= (1e-7 if fold else 0) and (d != 0).mean() <= 0.05
        np.testing.assert_allclose(bwd.numpy(), hbwd, rtol=1e-6, atol=1e-9)
    else:
        diff = stream.numpy().view(np.uint16) != blob.view(np.uint16)
        assert diff.mean() <= (0.05 if fold else 0.0)
    np.testing.assert_allclose(consts.numpy(), hconsts, rtol=0, atol=1e-7)
    # FiLM tensors: zeros behind every layer's block, and gradients sliced back
    t = torch.arange(2 * 3 * H, dtype=torch.float32).reshape(2, 3 * H)
    q = nm._pad_film(t, 3)
    assert q.shape == (2, 3 * Hp) and torch.equal(q.reshape(2, 3, Hp)[..., :H].reshape(2, 3 * H), t) and not q.reshape(2, 3, Hp)[..., H:].any()
    ng, nc = spec["n_geo"], spec["n_color"]
    res = dict(d_freq_geo=torch.randn(2, ng * Hp), d_phase_geo=torch.randn(2, ng * Hp), d_freq_app=torch.randn(2, nc * Hp), d_phase_app=torch.randn(2, nc * Hp),
               geo_w=[torch.randn(Hp, 3)] + [torch.randn(Hp, Hp) for _ in range(ng - 1)], geo_b=[torch.randn(Hp) for _ in range(ng)],
               color_w=[torch.randn(Hp, 3 + G + Hp)] + [torch.randn(Hp, Hp) for _ in range(nc - 1)], color_b=[torch.randn(Hp) for _ in range(nc)],
               head_w=torch.randn(32, Hp), head_b=torch.randn(32), rgb_w=torch.randn(3, Hp), rgb_b=torch.randn(3))
    u = nm._unpad_grads(res)
    assert u["d_freq_geo"].shape == (2, ng * H) and torch.equal(u["d_freq_geo"].reshape(2, ng, H), res["d_freq_geo"].reshape(2, ng, Hp)[..., :H])
    assert u["geo_w"][0].shape == (H, 3) and u["geo_w"][1].shape == (H, H) and u["color_w"][0].shape == (H, 3 + G + H)
    assert torch.equal(u["color_w"][0], res["color_w"][0][:H, :3 + G + H]) and u["head_w"].shape == (32, H) and u["rgb_w"].shape == (3, H)
    assert u["head_b"] is res["head_b"] and u["geo_b"][0].shape == (H,)
    wts = nm._pad_weights(([torch.randn(H, 3)] + [torch.randn(H, H)] * (ng - 1), [torch.randn(H, 3 + G + H)] + [torch.randn(H, H)] * (nc - 1)))
    assert wts[0][0].shape == (Hp, 3) and wts[0][1].shape == (Hp, Hp) and wts[1][0].shape == (Hp, 3 + G + Hp) and not wts[1][0][H:].any()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_repack_maps_cover_the_packed_buffers(precision):
    """FenerfRepackMaps (what fenerf_model_repack consumes; the kernels themselves are compared bit for bit on the GPU): sizes must add
    up to the host packer's buffers, every index must address the canonical flat parameter vector, every scaled row must be
    covered exactly once by scale_id, and the result-scale table must point at real rows."""
    import torch
    from fenerf_amd import native
    for kind, H, grid in [("texture", 64, 4), ("baseline", 32, 0), ("spatial", 32, 0)]:
        spec = proc.model_spec(kind, hidden_dim=H, grid_size=grid, z_dim=8)
        sd = proc.make_state_dict(spec, seed=2, sigma_gain=10.0, with_mapping=False)
        nm = object.__new__(native.NativeModel)
        nm.spec, nm.differentiable, nm.device, nm.precision, nm._maps, nm._rmaps, nm._h = dict(spec), True, torch.device("cpu"), precision, None, None, None
        r = nm._repack_maps()
        keep = nm._rmaps[1]
        blob, consts = _lib.pack_weights_host(sd, spec, precision)
        bwd = _lib.pack_backward_host(sd, spec, precision)
        n_flat = 1 + sum(int(np.prod(sh)) for _, sh in nm._canonical())
        assert r.n_stream_f32 + r.n_stream_h16 // 2 == blob.size and r.n_stream_h16 % 2 == 0
        assert r.n_consts + r.n_tail == consts.size
        assert r.n_bwd_f32 + r.n_bwd_b16 // 2 == bwd.size and r.n_bwd_b16 % 2 == 0
        for k in ("stream_f32", "consts", "bwd_f32", "bwd_b16"):
            if k in keep:
                assert 0 <= int(keep[k].min()) and int(keep[k].max()) < n_flat, k
        if precision == "f16x3":
            idx = keep["stream_h16"] & 0x3FFFFFFF
            assert int(idx.max()) < n_flat and set(torch.unique(keep["stream_h16"] >> 30).tolist()) <= {0, 1}
            sid = keep["scale_id"].numpy()
            assert sid.size == n_flat and sid.max() == r.n_rows
            off, ln = keep["row_off"].numpy(), keep["row_len"].numpy()
            for q in (0, r.n_rows // 2, r.n_rows - 1):                       # a row's elements carry its id, its neighbours do not
                assert (sid[off[q]:off[q] + ln[q]] == q + 1).all() and (sid == q + 1).sum() == ln[q]
            tail = keep["consts_tail"].numpy()
            L = spec["n_geo"] + spec["n_color"]
            assert tail.size == L * H + 36 and tail.max() <= r.n_rows and tail.min() >= -1
            assert (tail[:H] == -1).all() and (tail[H:L * H] > 0).all() and tail[-1] == -1
        else:
            assert r.n_stream_h16 == 0 and r.n_tail == 0 and r.n_rows == 0
