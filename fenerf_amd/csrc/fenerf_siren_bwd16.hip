// Backward chain of the FiLM-SIREN radiance field on the bf16 matrix pipe (models created with FENERF_PREC_F16X3).
// Same maths, stages, tape and outputs as siren_bwd_kernel (fenerf_siren_bwd.hip; reference siren/siren.py:1509-1530 under
// torch autograd), but dx_{l-1} = W'^T dz runs on v_mfma_f32_32x32x16_bf16 with both operands split into bf16 (hi, lo):
//     W'^T dz  ~=  wl*xh + wh*xl + wh*xh        (fp32 accumulate; the dropped wl*xl term is 2^-16 relative)
// -- three 32-cycle MFMAs per 16 features instead of eight 64-cycle fp32 MFMAs.  Weights are split on the host
// (hi = RNE, lo = RNE of the remainder: 16+ significant bits); dz is split in the epilogue (hi = truncation, lo = the exact
// remainder rounded: 16 bits, unbiased) and parked in the LDS slab as ready-made B operands.  Each wave streams its own
// copy of the weights through an 8-entry register ring (tools/probe/stream_probe.hip: a private L2 stream feeds one MFMA
// triple per ~180 cycles, 53 % of the matrix pipe, 2.8x the fp32 path).  Measurements and what bounds it: DESIGN.md 4.5.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_mfma32.h"
#include "fenerf_nt.h"

namespace fenerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float cos2pi16(float t) { return __builtin_amdgcn_cosf(t); }   // v_cos_f32, revolutions

// low half = bf16 truncation of v, high half = the (exact) remainder rounded to bf16.  Rounded, not truncated: a truncated
// remainder shrinks every dz by 2^-17 on average, and eleven layers of that bias showed as 8.5e-5 in the parameter gradients.
__device__ __forceinline__ unsigned split_pack(float v) {
  const unsigned vb = __builtin_bit_cast(unsigned, v);
  const float hi = __builtin_bit_cast(float, vb & 0xffff0000u);
  const unsigned rb = __builtin_bit_cast(unsigned, v - hi);
  return (vb >> 16) | ((rb + 0x8000u) & 0xffff0000u);
}
// 8 split-packed values -> the (hi x 8, lo x 8) operand pair
__device__ __forceinline__ void unpack8(const unsigned (&sp)[8], float4& hi, float4& lo) {
  uint4 h, l;
  h.x = __builtin_amdgcn_perm(sp[1], sp[0], 0x05040100u); l.x = __builtin_amdgcn_perm(sp[1], sp[0], 0x07060302u);
  h.y = __builtin_amdgcn_perm(sp[3], sp[2], 0x05040100u); l.y = __builtin_amdgcn_perm(sp[3], sp[2], 0x07060302u);
  h.z = __builtin_amdgcn_perm(sp[5], sp[4], 0x05040100u); l.z = __builtin_amdgcn_perm(sp[5], sp[4], 0x07060302u);
  h.w = __builtin_amdgcn_perm(sp[7], sp[6], 0x05040100u); l.w = __builtin_amdgcn_perm(sp[7], sp[6], 0x07060302u);
  hi = __builtin_bit_cast(float4, h);
  lo = __builtin_bit_cast(float4, l);
}

struct Ring16 {
  float4 w[FENERF_PF16];
  const float4* ptr;
};

struct Tape16 { float a[16]; };
__device__ __forceinline__ Tape16 tape_load16(const float4* tp, int nb) {
  Tape16 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = nt_load(tp + (nb * 4 + j) * 64);
    t.a[4 * j + 0] = v.x; t.a[4 * j + 1] = v.y; t.a[4 * j + 2] = v.z; t.a[4 * j + 3] = v.w;
  }
  return t;
}

// The epilogue of one accumulator register r of n-block nb.  dL/dtheta leaves in groups of four (one float4 of the
// register-dump layout), dL/dz in groups of eight: registers 8q..8q+7 are the lane's slots of k-step 2 nb + q.
// Slab layout (float4 units): [(2 s16 + {0: hi, 1: lo}) * 64 + lane].
struct BwdOct { float d[4]; unsigned sp[8]; };
// where an epilogue writes: the lane's LDS slab, the dtheta dump and the FiLM sums of (tile, layer) (fenerf_mfma32.h)
struct Sink16 { float4* slab; float4* dtp; float* ftp; int H; LaneBits lb; };
__device__ __forceinline__ void bwd_piece16(int r, const f32x16& acc, const FilmNB& fm, const Tape16& tn, int nb, const Sink16& k, BwdOct& q,
                                            FilmRed& R) {
  float4* const slab = k.slab; float4* const dtp = k.dtp;
  const float TWO_PI = 6.28318530717958647692f;
  const int j = r >> 2, i = r & 3;
  const float f = i == 0 ? fm.f[j].x : (i == 1 ? fm.f[j].y : (i == 2 ? fm.f[j].z : fm.f[j].w));
  const float p = i == 0 ? fm.p[j].x : (i == 1 ? fm.p[j].y : (i == 2 ? fm.p[j].z : fm.p[j].w));
  const float dt = acc[r] * cos2pi16(__builtin_fmaf(f, tn.a[r], p));
  q.d[i] = dt;
  R.v[0][r] = dt;
  R.v[1][r] = dt * tn.a[r];
  q.sp[r & 7] = split_pack(dt * (f * TWO_PI));
#ifndef EXP_B16_NOSTORE
#ifdef EXP_B16_STORE_L2
  if (i == 3) dtp[((nb & 1) * 4 + j) * 64] = make_float4(q.d[0], q.d[1], q.d[2], q.d[3]);
#else
  if (i == 3) nt_store(dtp + (nb * 4 + j) * 64, q.d[0], q.d[1], q.d[2], q.d[3]);
#endif
#endif
  if ((r & 7) == 7) {
    float4 hi, lo;
    unpack8(q.sp, hi, lo);
    const int s16 = 2 * nb + (r >> 3);
    slab[(2 * s16 + 0) * 64] = hi;
    slab[(2 * s16 + 1) * 64] = lo;
  }
}

__device__ __forceinline__ void bwd_store16(const f32x16& acc, const FilmNB& fm, const Tape16& tn, int nb, const Sink16& k) {
  BwdOct q;
  FilmRed R;
#pragma unroll
  for (int r = 0; r < 16; ++r) bwd_piece16(r, acc, fm, tn, nb, k, q, R);
#pragma unroll
  for (int c = 0; c < FILM_RED_CHUNKS; ++c) film_red_chunk(c, R, k.lb, k.ftp + 32 * nb, k.H);
}

// One of the 12 operand loads (4 tape float4 first -- they come from HBM --, then 8 FiLM float4) of n-block nb's epilogue.
__device__ __forceinline__ void prefetch_piece16(int i, FilmNB& fm, Tape16& tn, const float* fpl, const float* ppl, const float4* tp, int nb) {
  if (i < 4) {
#ifdef EXP_B16_NOTAPE
    const float4 v = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
#elif defined(EXP_B16_TAPE_L2)
    const float4 v = tp[((nb & 1) * 4 + i) * 64];
#else
    const float4 v = nt_load(tp + (nb * 4 + i) * 64);
#endif
    tn.a[4 * i + 0] = v.x; tn.a[4 * i + 1] = v.y; tn.a[4 * i + 2] = v.z; tn.a[4 * i + 3] = v.w;
  } else if (i < 8) fm.f[i - 4] = *reinterpret_cast<const float4*>(fpl + 32 * nb + 8 * (i - 4));
  else if (i < 12) fm.p[i - 8] = *reinterpret_cast<const float4*>(ppl + 32 * nb + 8 * (i - 8));
}

template <int KS>
struct Act16 { bf16x8 hi[KS], lo[KS]; };

template <int KS>
__device__ __forceinline__ void load_act16(Act16<KS>& x, const float4* slab) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    x.hi[s] = __builtin_bit_cast(bf16x8, slab[(2 * s + 0) * 64]);
    x.lo[s] = __builtin_bit_cast(bf16x8, slab[(2 * s + 1) * 64]);
  }
}

// acc += body * x over KS real k-steps (EP entries consumed), extra(ks, wh, wl) handles k-steps >= KS, piece(slot) is
// issued behind MFMA slot = 3 ks + j.
template <int KS, int EP, class EXTRA, class PIECE>
__device__ __forceinline__ void mfma16_x(f32x16& acc, const Act16<KS>& x, Ring16& ring, EXTRA extra, PIECE piece) {
  static_assert(EP % FENERF_PF16 == 0 && EP % 2 == 0, "bodies are padded to the ring depth");
#pragma unroll
  for (int ks = 0; ks < EP / 2; ++ks) {
    float4 wh, wl;
    RING_NEXT(ring, (2 * ks) % FENERF_PF16, wh);
    RING_NEXT(ring, (2 * ks + 1) % FENERF_PF16, wl);
    if (ks < KS) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, wh), al = __builtin_bit_cast(bf16x8, wl);
      // a dependent MFMA holds the wave's in-order issue port until its predecessor retires (32 cycles): work placed
      // BEHIND each one runs in that shadow, work placed after all three only in the last one's
      acc = MFMA_BF16(al, x.hi[ks], acc);
      piece(3 * ks + 0);
      acc = MFMA_BF16(ah, x.lo[ks], acc);
      piece(3 * ks + 1);
      acc = MFMA_BF16(ah, x.hi[ks], acc);
      piece(3 * ks + 2);
    } else {
      extra(ks - KS, wh, wl);
    }
    // pin (refill, MFMAs, piece) per k-step: otherwise the scheduler sinks the refills next to their use
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dz_l (registers) -> dz_{l-1}: one transposed square stage, software-pipelined over the n-blocks: behind the MFMAs of body
// nb run first the epilogue of body nb-1 (16 pieces), then -- into the FiLM / tape registers that epilogue has just
// finished with -- the operand loads of body nb's own epilogue (12 pieces), 28 pieces spread over the body's 3 KS MFMAs.
// (The vmcnt queue is in order: a tape load must land before the ring entries issued behind it are consumed, 8 k-steps
// later, whichever registers it targets -- a third operand set would buy no extra latency tolerance.)
// EP = entries per body (a square body, or the colour-layer-0 body with its two head k-steps, which extra(acc, s, wh, wl)
// multiplies); RELOAD: fetch the stage's output from the slab as the next stage's input.
template <int H, int EP, bool RELOAD, class EXTRA>
__device__ __forceinline__ void bwd_stage16(Act16<H / 16>& in, Ring16& ring, const float* fpl, const float* ppl, const float4* tp,
                                            const Sink16& k, EXTRA extra) {
  float4* const slab = k.slab;
  constexpr int NB = H / 32, KS = H / 16;
  constexpr int NPIECE = 28 + FILM_RED_CHUNKS;           // 16 epilogue + 12 operand loads + the FiLM-sum butterfly
  constexpr int PP = (NPIECE + 3 * KS - 1) / (3 * KS);   // pieces per MFMA slot
  FilmNB fm = film_load(fpl, ppl, 0);
  Tape16 tn = tape_load16(tp, 0);
  f32x16 acc_p = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto body = [&](int nb, auto has_prev) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    BwdOct q;
    FilmRed R;
    mfma16_x<KS, EP>(acc, in, ring, [&](int s_, float4 wh, float4 wl) { extra(acc, s_, wh, wl); }, [&](int slot) {
#ifdef EXP_B16_NOPIECE
      if (false) {
#else
      if (has_prev.value) {
#endif
#pragma unroll
        for (int i = slot * PP; i < (slot + 1) * PP; ++i) {
          if (i < 16) bwd_piece16(i, acc_p, fm, tn, nb - 1, k, q, R);
          else if (i < 28) prefetch_piece16(i - 16, fm, tn, fpl, ppl, tp, nb);
          else if (i < NPIECE) film_red_chunk(i - 28, R, k.lb, k.ftp + 32 * (nb - 1), k.H);
        }
      }
    });
    acc_p = acc;
  };
  body(0, std::false_type{});
#pragma unroll 1
  for (int nb = 1; nb < NB; ++nb) body(nb, std::true_type{});
  bwd_store16(acc_p, fm, tn, NB - 1, k);
  if (RELOAD) load_act16<H / 16>(in, slab);
}

template <int H>
__device__ __forceinline__ void bwd_square16(Act16<H / 16>& in, Ring16& ring, const float* fpl, const float* ppl, const float4* tp,
                                             const Sink16& k) {
  bwd_stage16<H, pad_pf16(2 * (H / 16)), true>(in, ring, fpl, ppl, tp, k, [](f32x16&, int, float4, float4) {});
}

template <int H, bool GRID>
__global__ __launch_bounds__(256, 1) void siren_bwd16_kernel(SirenBwdParams P, int n_geo, int n_color, int n_lab, int C) {
  constexpr int NB = H / 32, KS = H / 16, EP = pad_pf16(2 * KS), C0_EP = pad_pf16(2 * (KS + 2));
  constexpr int SLAB_F4 = (H / 8) * 64;
  extern __shared__ __attribute__((aligned(16))) float4 smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const LaneBits lbits = lane_bits(lane);
  float4* slab = smem + wave * SLAB_F4 + lane;
  const int L = n_geo + n_color;
  const float4* htw = reinterpret_cast<const float4*>(P.stream) + lane;
  const float4* ring_base = reinterpret_cast<const float4*>(P.stream + P.ring_offset_floats) + lane;

  const long long ntiles = (P.P + 31) / 32;
  const long long wstride = (long long)gridDim.x * 4;
  constexpr int tl = (H / 8) * 64;   // float4 units per (tile, layer)
#ifdef EXP_B16_TAPE_L2
  constexpr int tl_t = 0;
#else
  constexpr int tl_t = tl;
#endif
#ifdef EXP_B16_STORE_L2
  constexpr int tl_d = 0;
#else
  constexpr int tl_d = tl;
#endif
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += wstride) {
    long long pt = tile * 32 + m;
    const bool valid = pt < P.P;
    if (!valid) pt = P.P - 1;
    const long long img = pt / P.pts_per_image;
    const float* fpl = P.fp + (size_t)img * L * H + 4 * h;
    const float* ppl = P.pp + (size_t)img * L * H + 4 * h;
#ifdef EXP_B16_TAPE_L2
    const float4* tp = reinterpret_cast<const float4*>(P.tape) + (blockIdx.x * 4 + wave) * 512 + lane;
#else
    const float4* tp = reinterpret_cast<const float4*>(P.tape) + tile * L * (long long)tl + lane;
#endif
#ifdef EXP_B16_STORE_L2
    float4* dtp = reinterpret_cast<float4*>(P.d_t) + (blockIdx.x * 4 + wave) * 512 + lane;
#else
    float4* dtp = reinterpret_cast<float4*>(P.d_t) + tile * L * (long long)tl + lane;
#endif

    // FiLM sums of this tile: [layer][2][H]; after the butterfly lane i holds the sum of accumulator register i & 15
    float* ftp = P.film_tiles + tile * L * 2LL * H + film_lane_feature(lane);
    auto sink = [&](int layer) { return Sink16{slab, dtp + layer * tl_d, ftp + layer * 2 * H, H, lbits}; };

    Ring16 ring;
    ring.ptr = ring_base;
#pragma unroll
    for (int i = 0; i < FENERF_PF16; ++i) { ring.w[i] = *ring.ptr; ring.ptr += 64; }

    // gradient wrt the head rows this lane multiplies in head k-step s: row 16 s + 8 h + t  (rows [0,n_lab) labels, row n_lab sigma)
    bf16x8 dh_hi[2], dh_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned sp[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = 16 * s + 8 * h + t;
        const int ch = row < n_lab ? row : (row == n_lab ? C - 1 : -1);
        sp[t] = split_pack(ch >= 0 ? P.d_out[pt * C + ch] : 0.f);
      }
      float4 hi, lo;
      unpack8(sp, hi, lo);
      dh_hi[s] = __builtin_bit_cast(bf16x8, hi);
      dh_lo[s] = __builtin_bit_cast(bf16x8, lo);
    }
    // ---------------- rgb head (fp32 MFMA, two k-steps): d(pre-sigmoid) = d_rgb * s (1 - s);  dx_{L-1} = W_rgb^T d(pre) ----------------
    {
      float dpre[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float s = P.out[pt * C + (C - 4) + c];
        dpre[c] = P.d_out[pt * C + (C - 4) + c] * (s * (1.f - s));
      }
      const float b0 = h ? dpre[1] : dpre[0], b1 = h ? 0.f : dpre[2];
      const int l = L - 1;
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const float4 w = htw[nb * 64];
        const FilmNB fm = film_load(fpl + (size_t)l * H, ppl + (size_t)l * H, nb);
        const Tape16 tn = tape_load16(tp + l * tl_t, nb);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = MFMA(w.x, b0, acc);
        acc = MFMA(w.y, b1, acc);
        bwd_store16(acc, fm, tn, nb, sink(l));
      }
    }
    Act16<KS> in;
    load_act16<KS>(in, slab);

    // ---------------- colour layers L-1 .. n_geo+1 ----------------
#pragma unroll 1
    for (int l = L - 1; l > n_geo; --l)
      bwd_square16<H>(in, ring, fpl + (size_t)(l - 1) * H, ppl + (size_t)(l - 1) * H, tp + (l - 1) * tl_t, sink(l - 1));

    // ---------------- colour layer 0 + heads: dx_{n_geo-1} = W_c0[:, x]^T dz_{n_geo} + head^T d_head; d(grid feats) ----
    {
      const int l = n_geo - 1;
      // the slab is overwritten n-block by n-block while this stage's input lives in registers (in); it is re-read only
      // after the grid-feature body below, which still multiplies the old input
      bwd_stage16<H, C0_EP, false>(in, ring, fpl + (size_t)l * H, ppl + (size_t)l * H, tp + l * tl_t, sink(l),
                                   [&](f32x16& acc, int s_, float4 wh, float4 wl) {
        if (s_ < 2) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, wh), al = __builtin_bit_cast(bf16x8, wl);
          acc = MFMA_BF16(al, dh_hi[s_], acc);
          acc = MFMA_BF16(ah, dh_lo[s_], acc);
          acc = MFMA_BF16(ah, dh_hi[s_], acc);
        }
      });
      if (GRID) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        mfma16_x<KS, EP>(acc, in, ring, [](int, float4, float4) {}, [](int) {});
        if (valid) {
          float4* ep = reinterpret_cast<float4*>(P.d_e + pt * 32 + 4 * h);   // channels 8j + 4h + {0..3}
#pragma unroll
          for (int j = 0; j < 4; ++j) ep[2 * j] = make_float4(acc[4 * j + 0], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        }
      }
    }
    load_act16<KS>(in, slab);

    // ---------------- geometry trunk n_geo-1 .. 1 ----------------
#pragma unroll 1
    for (int l = n_geo - 1; l >= 1; --l)
      bwd_square16<H>(in, ring, fpl + (size_t)(l - 1) * H, ppl + (size_t)(l - 1) * H, tp + (l - 1) * tl_t, sink(l - 1));
    __builtin_amdgcn_wave_barrier();
  }
}

static int hip_fail_b16(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

template <int H, bool GRID>
static int launch_bwd16_t(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
  const size_t lds = (size_t)4 * ((H / 8) * 64) * sizeof(float4);
  auto kfn = siren_bwd16_kernel<H, GRID>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  const long long ntiles = (p.P + 31) / 32;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > m->num_cus) blocks = m->num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail_b16(e, "siren bf16 backward launch");
}

int launch_siren_backward16(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
  if (p.P <= 0) return FENERF_OK;
  if (bwd16w_enabled()) return launch_siren_backward16w(m, p, stream);   // 16-point waves, shared stream (fenerf_siren_bwd16w.hip)
  const bool g = m->grid_ch != 0;
  switch (m->H) {
    case 32: return g ? launch_bwd16_t<32, true>(m, p, stream) : launch_bwd16_t<32, false>(m, p, stream);
    case 64: return g ? launch_bwd16_t<64, true>(m, p, stream) : launch_bwd16_t<64, false>(m, p, stream);
    case 128: return g ? launch_bwd16_t<128, true>(m, p, stream) : launch_bwd16_t<128, false>(m, p, stream);
    case 256: return g ? launch_bwd16_t<256, true>(m, p, stream) : launch_bwd16_t<256, false>(m, p, stream);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

}  // namespace fenerf
