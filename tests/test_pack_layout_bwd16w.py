"""CPU check of the dataflow of fenerf_siren_bwd16w.hip (backward chain on 16-point waves, v_mfma_f32_16x16x32_bf16) on the
blob the C packer produced for the 32x32x16 chain kernel: numpy emulation of the two waves that share a 32-point tile --
the LDS-DMA re-tiling permutation applied to the BACKWARD stream (whole-chunk bodies, no stage padding, one tail chunk),
the rgb-head^T block on the 16x16x4 fp32 MFMA, the accumulator -> next-stage B operand identity, the head k32-step of the
colour-layer-0 bodies, the grid-feature body, the d(theta) dump positions (tape layout of fenerf_layout.h) and the FiLM-sum
layout film_gather_kernel decodes -- compared with torch fp64 autograd.  The split arithmetic and the MFMA lane maps
themselves are checked on hardware (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from test_pack_layout_bwd import chain_case, check_chain
from test_pack_layout_f16w import G_, N_, dma_operand, mfma16w, mfma32w, row_of

CH = 8


def emulate_chain16w(blob, spec, theta, f_true, row_scale, d_out, out):
    H, NB, KS = spec["hidden_dim"], spec["hidden_dim"] // 32, spec["hidden_dim"] // 32
    pad = lambda n: (n + CH - 1) // CH * CH
    QB, C0_QB = pad(2 * (H // 16)) // CH, pad(2 * (H // 16 + 2)) // CH
    n_geo, n_color, C = spec["n_geo"], spec["n_color"], spec["output_dim"]
    n_lab, L = C - 4, n_geo + n_color
    head = blob[:NB * 256].reshape(NB, 64, 4).astype(np.float64)
    ring = (blob[NB * 256:].view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64).reshape(-1, CH, 64, 8)

    dump = np.zeros((L, H // 8, 64, 4))               # d(theta) of the 32-point tile in the tape's register-dump layout
    film = np.zeros((2, L, NB, 2, 16, 2))             # [tile16][layer][nb][rt][slot][s0, s1]
    d_e = np.zeros((32, 32))
    feat0 = 16 * (G_ >> 1) + 4 * (G_ & 1)
    for half in range(2):                              # the two waves of the tile
        pt = 16 * half + N_
        cur = [0]

        def body(bops, n_chunks):
            acc = [np.zeros((64, 4)), np.zeros((64, 4))]
            for qc in range(n_chunks):
                chunk = ring[cur[0]]
                cur[0] += 1
                for spl in range(2):
                    sp = 2 * qc + spl
                    if sp < len(bops):
                        for rt in range(2):
                            mfma16w(dma_operand(chunk, spl, rt, 0) + dma_operand(chunk, spl, rt, 1), bops[sp], acc[rt])
                    else:
                        assert not chunk[4 * spl:4 * spl + 4].any()          # padding k-steps carry zero weights
            return acc

        def epilogue(acc, layer, nb, y):
            for rt in range(2):
                feat = 32 * nb + feat0[:, None] + 8 * rt + np.arange(4)[None, :]          # [lane][r]
                th = theta[layer][feat, pt[:, None]]
                dt = acc[rt] * np.cos(th)
                # the kernel's store: float4 index (4 nb + 2 (g >> 1) + rt) * 64 + 32 (g & 1) + 16 half + n of the layer's block
                idx = (nb * 4 + 2 * (G_ >> 1) + rt) * 64 + 32 * (G_ & 1) + 16 * half + N_
                dump[layer].reshape(-1, 4)[idx] = dt
                y[nb][:, 4 * rt:4 * rt + 4] = dt * f_true[layer][feat] / row_scale[layer][feat]
                # FiLM sums over the 16 points of the row (= lane group); theta stands in for the tape value
                for g in range(4):
                    rows = np.flatnonzero(G_ == g)
                    film[half, layer, nb, rt, 4 * g:4 * g + 4, 0] = dt[rows].sum(0)
                    film[half, layer, nb, rt, 4 * g:4 * g + 4, 1] = (dt[rows] * th[rows]).sum(0)

        # rgb head^T on the fp32 MFMA: k = r, g, b, 0
        s = out[pt, C - 4:C - 1]
        dpre = d_out[pt, C - 4:C - 1] * s * (1 - s)
        b = np.select([G_ == 0, G_ == 1, G_ == 2], [dpre[:, 0], dpre[:, 1], dpre[:, 2]], 0.0)
        z = np.zeros((KS, 64, 8))
        for nb in range(NB):
            acc = [np.zeros((64, 4)), np.zeros((64, 4))]
            for rt in range(2):
                a = head[nb][32 * (G_ & 1) + np.array([row_of(rt, i) for i in N_]), G_ >> 1]
                mfma32w(a, b, acc[rt])
            epilogue(acc, L - 1, nb, z)
        # head rows as the B operand of the head k32-step: lane (n, kg) slot t = row 8 kg + t
        dh = np.zeros((64, 8))
        for t in range(8):
            row = 8 * G_ + t
            ch = np.where(row < n_lab, row, np.where(row == n_lab, C - 1, -1))
            dh[:, t] = np.where(ch >= 0, d_out[pt, np.maximum(ch, 0)], 0.0)
        for lo in range(L - 2, -1, -1):
            y = np.zeros((KS, 64, 8))
            if lo == n_geo - 1:
                for nb in range(NB):
                    epilogue(body(list(z) + [dh], C0_QB), lo, nb, y)
                if spec["grid_ch"]:
                    acc = body(list(z), QB)
                    for rt in range(2):
                        for r in range(4):
                            d_e[pt, feat0 + 8 * rt + r] = acc[rt][:, r]
            else:
                for nb in range(NB):
                    epilogue(body(list(z), QB), lo, nb, y)
            z = y
        assert cur[0] + 1 == ring.shape[0], "the backward stream must be consumed exactly (+ the tail chunk)"

    # decode the dump with the tape layout (fenerf_layout.h: element i of lane (m, half) in group g = feature 32 nb + 8 j + 4 half + i)
    dtheta = np.zeros((L, H, 32))
    for g4 in range(H // 8):
        for lane in range(64):
            m, hf = lane & 31, lane >> 5
            for i in range(4):
                dtheta[:, 32 * (g4 >> 2) + 8 * (g4 & 3) + 4 * hf + i, m] = dump[:, g4, lane, i]
    # decode the FiLM sums like film_gather_kernel
    sums = np.zeros((L, 2, H))
    for n in range(H):
        f = n & 31
        gq, rt, r = ((f >> 4) << 1) | ((f >> 2) & 1), (f >> 3) & 1, f & 3
        sums[:, :, n] = film[:, :, n >> 5, rt, 4 * gq + r, :].sum(0)
    return dtheta, d_e, sums


@pytest.mark.parametrize("kind,H,grid", [("texture", 32, 5), ("baseline", 64, 0), ("spatial", 32, 0), ("texture", 128, 4), ("texture", 96, 5), ("baseline", 192, 0)])
def test_backward16w_stream_walk_and_dataflow(kind, H, grid):
    case = chain_case(kind, H, grid, "f16x3")
    got, d_e, sums = emulate_chain16w(case["blob"], case["spec"], case["theta"], case["f_true"], case["row_scale"], case["d_out"], case["out"])
    check_chain(case, got, d_e, "f16x3", grid)
    ref, th = case["ref"], case["theta"]
    np.testing.assert_allclose(sums[:, 0], ref.sum(-1), atol=2e-5 * max(1.0, np.abs(ref).max()) * 32, rtol=1e-4)
    np.testing.assert_allclose(sums[:, 1], (ref * th).sum(-1), atol=2e-5 * max(1.0, np.abs(ref * th).max()) * 32, rtol=1e-4)
