// Streaming weight layout shared by the host packer (fenerf_pack.cpp) and the SIREN kernel
// (fenerf_siren.hip).  See DESIGN.md "SIREN kernel" for the derivation.
//
// The per-point network is evaluated TRANSPOSED on v_mfma_f32_32x32x2_f32: D[feature][point] +=
// W[feature][k] * X^T[k][point].  One wave owns 32 points (MFMA columns) and ALL H features of them.
// MFMA C/D layout (lane l, acc register r): column = l&31 (the point), row = (r&3) + 8*(r>>2) + 4*(l>>5).
// So after a layer, lane (point m, half h) holds, for n-block nb and register r, output feature
//      feat_of(16*nb + r, h) = 32*nb + (r&3) + 8*(r>>2) + 4*h .
// The next layer contracts over k in exactly that order: k-step s (s = 16*nb + r) multiplies
// B = {half 0: feature feat_of(s,0), half 1: feature feat_of(s,1)} -- the lane's OWN register --
// with A = W[n][feat_of(s, l>>5)].  The K permutation is applied to the weights once, on the host,
// so activations never leave the lane that produced them (no LDS transpose, no cross-lane traffic).
//
// An "entry" is one wave-wide float4 load: 64 lanes x 16 B = 1 KiB = the A operands of 4 consecutive
// k-steps for one 32-row n-block.  Bodies (one n-block of one stage) are padded to a multiple of
// FENERF_PF entries so the software prefetch ring has compile-time slot indices.
#pragma once

#define FENERF_PF 8            /* prefetch ring depth, entries */
#define FENERF_E_KSTEPS 16     /* 32 grid channels = 16 k-steps (half h holds channels 16h..16h+15) */

#ifdef __cplusplus
namespace fenerf {

constexpr int pad_pf(int n) { return (n + FENERF_PF - 1) / FENERF_PF * FENERF_PF; }
constexpr int feat_of(int s, int h) { return 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h; }

struct StreamShape {
  int H, NB, KGX, KGXP;  // features, n-blocks, k-groups of an H-wide input, padded
  int c0_kg, c0_kgp;     // colour-layer-0 body: x + grid + dir k-groups, padded
  int l0_entries;        // layer-0 block (plain loads, not in the ring)
  long long ring_entries; // total entries of the ring stream incl. the PF tail pad
};

inline StreamShape stream_shape(int H, int n_geo, int n_color, bool grid) {
  StreamShape s;
  s.H = H; s.NB = H / 32; s.KGX = H / 8; s.KGXP = pad_pf(s.KGX);
  s.c0_kg = s.KGX + (grid ? FENERF_E_KSTEPS / 4 : 0) + 1;
  s.c0_kgp = pad_pf(s.c0_kg);
  s.l0_entries = s.NB;
  long long e = 0;
  e += (long long)(n_geo - 1) * s.NB * s.KGXP;   // G1..G(n_geo-1)
  e += (long long)s.NB * s.c0_kgp;               // C0
  e += s.KGXP;                                   // HEAD (labels + sigma)
  e += (long long)(n_color - 1) * s.NB * s.KGXP; // C1..
  e += s.KGXP;                                   // RGB
  e += FENERF_PF;                                // tail pad (prefetch runs past the end)
  s.ring_entries = e;
  return s;
}

// ---------------------------------------------------------------------------------------------
// Backward chain (fenerf_siren_bwd.hip): dx_{l-1} = W_l^T dz_l runs on the same transposed fp32 MFMA with the same
// C/D -> B register identity, so its stream is the forward format applied to the TRANSPOSED matrices, in reverse
// layer order:  [rgb-head^T block: NB plain entries] | ring: C_{L-1}^T .. C_{n_geo+1}^T, the colour-layer-0 stage
// (NB bodies [W_c0[:, x-part]^T | head^T (16 k-steps: lane-half h multiplies head row 16h + s)], then one body
// W_c0[:, grid-part]^T -> d(grid features)), G_{n_geo-1}^T .. G_1^T, tail pad.
#define FENERF_HEAD_KSTEPS 16
struct BwdShape {
  int H, NB, KGX, KGXP;
  int c0_kg, c0_kgp;
  int ht_entries;
  long long ring_entries;
};

inline BwdShape bwd_stream_shape(int H, int n_geo, int n_color, bool grid) {
  BwdShape s;
  s.H = H; s.NB = H / 32; s.KGX = H / 8; s.KGXP = pad_pf(s.KGX);
  s.c0_kg = s.KGX + FENERF_HEAD_KSTEPS / 4;
  s.c0_kgp = pad_pf(s.c0_kg);
  s.ht_entries = s.NB;
  long long e = 0;
  e += (long long)(n_color - 1) * s.NB * s.KGXP;
  e += (long long)s.NB * s.c0_kgp + (grid ? s.KGXP : 0);
  e += (long long)(n_geo - 1) * s.NB * s.KGXP;
  e += FENERF_PF;
  s.ring_entries = e;
  return s;
}

// Tape (differentiable mode): the forward's pre-FiLM accumulators and the backward's dL/dtheta are kept as REGISTER DUMPS
// of the 32-point tiles, [tile][layer][g = 4 nb + j (H/8)][lane (64)][4 floats]: element i of lane (m, half) in group g is
// feature 32 nb + 8 j + 4 half + i of point 32 tile + m.  Every wave store / load is one contiguous 1-KiB transfer and a
// tile's whole tape is one contiguous L*H*128-byte run (vs 128-B pieces scattered over L*H rows for a feature-major
// matrix: measured 10x slower).  Point counts are padded to whole tiles by the caller (pad gradients are zero).
// FiLM sums (backward): per 32-point tile and FiLM layer the chain kernels also emit s0[n] = sum_p dtheta[n][p] and
// s1[n] = sum_p dtheta[n][p] * tape[n][p] (raw accumulator units), [tile][layer][2][H] floats, appended to the dtheta dump.
constexpr long long film_tile_floats(long long tiles, int L, int H) { return tiles * L * 2 * H; }
// bf16 dump (round 3; opt-in per model, FenerfModelDesc.wgrad_bf16_min_points: AMP-class weight gradients): the chain kernel writes,
// into the SAME (tile32, layer) block of H*128 bytes, first d theta_l and then x_l = sin(2 pi theta_l) -- which it has in registers
// anyway -- rounded to nearest-even bf16:
//     [which: d theta | x][nb (H/32)][16-point tile (2)][lane (64)][8 bf16],  lane (n, g) slot t = 4 rt + r:
//     feature 32 nb + 16 (g >> 1) + 4 (g & 1) + 8 rt + r of point 32 tile32 + 16 (tile & 1) + n
// (a 16-point wave's accumulator registers of one n-block, both row tiles, as one 16-byte store per lane).  The square
// weight-gradient job then reads 2 + 2 bytes per (point, feature) instead of 4 + 4 and never touches the tape.  Rounding errors are
// unbiased and independent over points, i.e. ~1e-3 of the NOISE FLOOR sqrt(sum_p (dtheta x)^2) of a gradient entry: measured 2.5e-3
// (max-norm relative) against fp64 autograd for point-wise random upstream gradients at 65,536 and at 393,216 points alike -- it does
// not average down with the point count, which is why this is an opt-in training precision and not the default
// (tests/test_gpu_parity.py::test_siren_backward_at_scale_vs_fp64_autograd).  FiLM sums and dz keep full precision (registers).
// 16-bit tape (round 5; opt-in per call, FENERF_TAPE_U16, FENERF_PREC_F16X3 models): all the backward needs from a FiLM layer is
// sin / cos of its phase theta = f'' t + p' (siren.py:113-123), i.e. frac(theta) in revolutions.  The forward-save kernel stores it as
// 16-bit fixed point, round to nearest -- +-2^-17 rev = +-4.8e-5 rad on every recomputed activation and cosine -- in the bf16 dump's
// piece layout: a (tile32, layer) block is H*64 bytes,
//     [nb (H/32)][16-point tile (2)][lane (64)][8 x u16],  lane (n, g) slot t = 4 rt + r: feature dump16_feature(nb, g, t) of point
//     32 tile32 + 16 (tile & 1) + n
// -- ONE 16-byte store per lane and n-block in the forward-save kernel (two before), ONE 1-KiB LDS-DMA per body in the chain kernel,
// 2 instead of 4 bytes per (point, feature) in the chain and in every weight-gradient job that recomputes activations.  What is lost:
// the raw accumulator t, which only the FiLM FREQUENCY gradient needs (sum_p d theta (W x + b)); it is derived from the weight-gradient
// partial sums instead (fenerf_siren_wgrad.hip), so inversion -- FiLM gradients only, no weight-gradient launch -- keeps the fp32 tape.
// Gradient accuracy: 1.1e-4 (max-norm relative, simulated and measured) on top of the 3.5e-5 of the bf16x3 products: between the fp32
// class (6e-5 asserted) and the AMP class (3e-3), hence a tier of its own (siren.grad_precision = "tape16"), never the default.
constexpr int dump16_feature(int nb, int g, int t) { return 32 * nb + 16 * (g >> 1) + 4 * (g & 1) + 8 * (t >> 2) + (t & 3); }
constexpr int tape_feature(int g, int half, int i) { return 32 * (g >> 2) + 8 * (g & 3) + 4 * half + i; }

// ---------------------------------------------------------------------------------------------
// f16x3 mode (error-compensated fp16 MFMA): v_mfma_f32_32x32x16_f16, k-step = 16 features.
// Each fp32 product w*x is evaluated as wh*xh + wh*xl + wl*xh with (h, l) = fp16 hi/lo splits of
// (w * 2^e_layer) and (x * 16); the result is exact to ~2^-22 relative (fp32 class), accumulated in fp32.
// An f16 entry = 64 lanes x 8 halves = the A operand of ONE k-step of one 32-row block; per k-step the
// stream holds [hi entry, lo entry].  Lane (m, h) holds, for k-step s16 and slot t (0..7), the feature
//      feat16_of(s16, h, t) = feat_of(16*(s16>>1) + 8*(s16&1) + t, h)
// i.e. accumulator registers r = 8*(s16&1)+t of n-block s16>>1 (same C/D layout as the fp32 MFMA).
constexpr int feat16_of(int s16, int h, int t) { return feat_of(16 * (s16 >> 1) + 8 * (s16 & 1) + t, h); }
constexpr float F16_ACT_SCALE = 16.f;   // activations are carried as x*16 so the lo halves stay normal fp16

// bf16x3 backward chain (fenerf_siren_bwd16w.hip, models created with FENERF_PREC_F16X3): the same stages on
// v_mfma_f32_32x32x16_bf16 with W'^T and dz each split into bf16 (hi, lo) and three MFMAs per k-step (wl*xh + wh*xl + wh*xh).
// [rgb-head^T block: NB plain fp32 entries, as above] | ring of bf16 entries (64 lanes x 8 bf16 = one A operand; per
// k-step [hi entry, lo entry], weights split hi = RNE(w), lo = RNE(w - hi)); k-step s16 of an H-wide input reads the features
// feat16_of(s16, h, t).  The colour-layer-0 bodies append 2 head k-steps (lane-half h, slot t multiplies head row
// 16 ks' + 8 h + t).  Bodies are padded to whole revolutions of the FENERF_PF16-entry register ring.
#ifndef FENERF_PF16
#define FENERF_PF16 8
#endif
constexpr int pad_pf16(int e) { return (e + FENERF_PF16 - 1) / FENERF_PF16 * FENERF_PF16; }
struct BwdShape16 {
  int H, NB, KS16, body_e, body_ep;   // k-steps of an H-wide input; entries per square body (2 per k-step), padded
  int c0_ks, c0_e, c0_ep;             // colour-layer-0 body: KS16 + 2 head k-steps
  int ht_entries;
  long long ring_entries;
};

inline BwdShape16 bwd_stream_shape16(int H, int n_geo, int n_color, bool grid) {
  BwdShape16 s;
  s.H = H; s.NB = H / 32; s.KS16 = H / 16;
  s.body_e = 2 * s.KS16; s.body_ep = pad_pf16(s.body_e);
  s.c0_ks = s.KS16 + 2; s.c0_e = 2 * s.c0_ks; s.c0_ep = pad_pf16(s.c0_e);
  s.ht_entries = s.NB;
  long long e = 0;
  e += (long long)(n_color - 1) * s.NB * s.body_ep;
  e += (long long)s.NB * s.c0_ep + (grid ? s.body_ep : 0);
  e += (long long)(n_geo - 1) * s.NB * s.body_ep;
  e += FENERF_PF16;
  s.ring_entries = e;
  return s;
}


// Workgroup-shared stream geometry (fenerf_siren_f16w.hip, fenerf_siren_bwd16w.hip): chunks of FENERF_CH entries travel through an LDS ring of
// FENERF_NSLOT slots, FENERF_DPF chunks ahead of the consumer.  Every STAGE (a layer's n-block bodies) is padded to a
// whole number of ring revolutions, so every stage starts at ring slot 0 and all slot indices are compile-time constants;
// the first FENERF_DPF chunks are replicated after the end, so the prefetch pointer never wraps inside a tile.
#define FENERF_CH 8
#define FENERF_NSLOT 8
#define FENERF_DPF 6
constexpr int pad_stage(int entries) { return (entries + FENERF_CH * FENERF_NSLOT - 1) / (FENERF_CH * FENERF_NSLOT) * (FENERF_CH * FENERF_NSLOT); }

struct StreamShape16 {
  int H, NB, KS16, body_e, body_ep;  // k-steps of an H-wide input; entries per square body, padded to whole chunks
  int c0_ks, c0_e, c0_ep;            // colour-layer-0 body
  int sq_stage_e, c0_stage_e, head_stage_e;   // entries per stage incl. stage padding
  int l0_entries;
  long long tile_entries;            // entries one tile consumes (= nchunk * FENERF_CH)
  long long ring_entries;            // tile_entries + replicated head (FENERF_DPF chunks)
};

inline StreamShape16 stream_shape16(int H, int n_geo, int n_color, bool grid) {
  StreamShape16 s;
  s.H = H; s.NB = H / 32; s.KS16 = H / 16;
  s.body_e = 2 * s.KS16; s.body_ep = pad_pf(s.body_e);
  s.c0_ks = s.KS16 + (grid ? 2 : 0) + 1;
  s.c0_e = 2 * s.c0_ks; s.c0_ep = pad_pf(s.c0_e);
  s.sq_stage_e = pad_stage(s.NB * s.body_ep);
  s.c0_stage_e = pad_stage(s.NB * s.c0_ep);
  s.head_stage_e = pad_stage(s.body_ep);
  s.l0_entries = s.NB;
  long long e = 0;
  e += (long long)(n_geo - 1) * s.sq_stage_e;   // G1..
  e += s.c0_stage_e;                             // C0
  e += s.head_stage_e;                           // HEAD (labels + sigma)
  e += (long long)(n_color - 1) * s.sq_stage_e; // C1..
  e += s.head_stage_e;                           // RGB
  s.tile_entries = e;
  s.ring_entries = e + FENERF_DPF * FENERF_CH;
  return s;
}

// consts layout (floats): [0,32) head bias by head row, [32,36) rgb bias, [36, 36 + L*H) FiLM-layer biases
constexpr int CONST_HEAD_BIAS = 0;
constexpr int CONST_RGB_BIAS = 32;
constexpr int CONST_FILM_BIAS = 36;
// f16x3 mode appends after the FiLM biases: [L] per-FiLM-layer result scale 1/(2^e * 16) (1 for layer 0),
// then head_inv_scale, rgb_inv_scale.

}  // namespace fenerf
#endif
