// Device helpers shared by the forward (fenerf_siren.hip) and backward (fenerf_siren_bwd.hip) fp32-MFMA SIREN kernels:
// the v_mfma_f32_32x32x2_f32 wrapper, the 8-deep register prefetch ring over the packed weight stream, FiLM parameter
// loads and the per-wave LDS activation slab.  Layout conventions: fenerf_layout.h.
#pragma once
#include <hip/hip_runtime.h>

#include "fenerf_layout.h"
#include "fenerf_trig.h"

namespace fenerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// sin2pi / cos2pi (revolutions, any finite magnitude): fenerf_trig.h

struct Ring {
  float4 w[FENERF_PF];
  const float4* ptr;  // per-lane cursor: next entry to fetch
};

// Consume the next ring entry (compile-time slot) and refill the slot with the entry PF ahead.
#define RING_NEXT(ring, slot, dst)      \
  do {                                  \
    (dst) = (ring).w[(slot)];           \
    (ring).w[(slot)] = *(ring).ptr;     \
    (ring).ptr += 64;                   \
  } while (0)

// acc += W_body[:, k-steps of an H-wide activation] * b  (NKG real entries, NKGP consumed)
template <int NIN, int NKG, int NKGP>
__device__ __forceinline__ void mfma_x(f32x16& acc, const float (&b)[NIN], Ring& ring) {
  static_assert(NKG * 4 == NIN, "k-groups must cover the activation");
  static_assert(NKGP % FENERF_PF == 0, "bodies are padded to the ring depth");
#pragma unroll
  for (int kg = 0; kg < NKGP; ++kg) {
    float4 w;
    RING_NEXT(ring, kg % FENERF_PF, w);
    if (kg < NKG) {
      acc = MFMA(w.x, b[4 * kg + 0], acc);
      acc = MFMA(w.y, b[4 * kg + 1], acc);
      acc = MFMA(w.z, b[4 * kg + 2], acc);
      acc = MFMA(w.w, b[4 * kg + 3], acc);
    }
    // pin the (refill, 4 x MFMA) order: without it the scheduler sinks the refill loads next to their use
    // (to shorten live ranges) and the prefetch distance collapses from PF entries to ~1.
    __builtin_amdgcn_sched_barrier(0);
  }
}

// mfma_x with a per-k-group hook: piece(kg) is issued behind k-group kg's four MFMAs.  A dependent fp32 MFMA occupies the
// matrix pipe for 64 cycles but the wave's issue port for 4, so ~50 issue cycles per MFMA are free for other work.
template <int NIN, int NKG, int NKGP, class PIECE>
__device__ __forceinline__ void mfma_x_p(f32x16& acc, const float (&b)[NIN], Ring& ring, PIECE piece) {
  static_assert(NKG * 4 == NIN, "k-groups must cover the activation");
  static_assert(NKGP % FENERF_PF == 0, "bodies are padded to the ring depth");
#pragma unroll
  for (int kg = 0; kg < NKGP; ++kg) {
    float4 w;
    RING_NEXT(ring, kg % FENERF_PF, w);
    if (kg < NKG) {
      acc = MFMA(w.x, b[4 * kg + 0], acc);
      acc = MFMA(w.y, b[4 * kg + 1], acc);
      acc = MFMA(w.z, b[4 * kg + 2], acc);
      acc = MFMA(w.w, b[4 * kg + 3], acc);
    }
    piece(kg);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// FiLM parameters of one n-block for this lane-half: features 32nb + 8j + 4h + {0..3}, j = 0..3.
// Loaded BEFORE the n-block's MFMAs so the L2 latency hides behind them.
struct FilmNB { float4 f[4], p[4]; };
__device__ __forceinline__ FilmNB film_load(const float* fpl, const float* ppl, int nb) {
  FilmNB fm;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    fm.f[j] = *reinterpret_cast<const float4*>(fpl + 32 * nb + 8 * j);
    fm.p[j] = *reinterpret_cast<const float4*>(ppl + 32 * nb + 8 * j);
  }
  return fm;
}

// FiLM epilogue of one n-block: out = sin(2 pi (f' acc + p')) -> this lane's LDS slab, groups nb*4 .. nb*4+3
__device__ __forceinline__ void film_store(const f32x16& acc, const FilmNB& fm, int nb, float4* slab /* + lane */) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 f = fm.f[j], p = fm.p[j];
    float4 o;
    o.x = sin2pi(__builtin_fmaf(f.x, acc[4 * j + 0], p.x));
    o.y = sin2pi(__builtin_fmaf(f.y, acc[4 * j + 1], p.y));
    o.z = sin2pi(__builtin_fmaf(f.z, acc[4 * j + 2], p.z));
    o.w = sin2pi(__builtin_fmaf(f.w, acc[4 * j + 3], p.w));
    slab[(nb * 4 + j) * 64] = o;
  }
}

// ---- FiLM-gradient sums in the chain kernels --------------------------------------------------------------------
// d_phase = sum_p dtheta and d_freq ~ sum_p dtheta * (W x + b) contract over the POINT axis, which is the lane axis of the
// chain kernels.  While dtheta and the tape value are in registers anyway, the 2 x 16 per-lane values of an n-block are
// summed over the 32 lanes of each wave half with a TRANSPOSING butterfly: at step k every lane keeps, of each register
// pair, the partial sum its lane bit k selects and receives the partner lane's, so the register count halves per step and
// after four steps lane i holds the 16-lane sum of register (i & 15); one cross-row exchange finishes it.  ~70 VALU ops
// per 16 registers (a plain 5-step DPP reduction of every register: 160), results spread over 16 lanes = one small store.
// The weight-gradient kernels then need neither the layer's own tape nor a FiLM pass (a third of their HBM traffic).
struct LaneBits { bool b0, b1, b2, b3; };   // lane & 1, 2, 4, 8
__device__ __forceinline__ LaneBits lane_bits(int lane) { return LaneBits{(lane & 1) != 0, (lane & 2) != 0, (lane & 4) != 0, (lane & 8) != 0}; }

__device__ __forceinline__ float lane_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)); }   // quad_perm:[1,0,3,2]
__device__ __forceinline__ float lane_xor2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)); }   // quad_perm:[2,3,0,1]
__device__ __forceinline__ float lane_xor4(float x) {   // row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3
  const int xi = __builtin_bit_cast(int, x);
  int t = __builtin_amdgcn_update_dpp(0, xi, 0x104, 0xf, 0x5, true);
  t = __builtin_amdgcn_update_dpp(t, xi, 0x114, 0xf, 0xa, true);
  return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float lane_xor8(float x) {   // row_shl:8 into banks 0, 1; row_shr:8 into banks 2, 3
  const int xi = __builtin_bit_cast(int, x);
  int t = __builtin_amdgcn_update_dpp(0, xi, 0x108, 0xf, 0x3, true);
  t = __builtin_amdgcn_update_dpp(t, xi, 0x118, 0xf, 0xc, true);
  return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float lane_xor16(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F)); }   // swizzle(SWAP,16)

template <class XCHG>
__device__ __forceinline__ float fold(bool bit, float a, float b, XCHG xchg) {
  const float keep = bit ? b : a, send = bit ? a : b;
  return keep + xchg(send);
}

// v[0][r] = dtheta, v[1][r] = dtheta * tape of accumulator register r; the butterfly in 8 chunks (so that it can be issued
// piecewise behind MFMAs).  ftp = film_tiles + ((tile * L + layer) * 2) * H + 32 * nb + feature offset of register (lane & 15)
// in this lane's half (fenerf_layout.h "FiLM sums").
constexpr int FILM_RED_CHUNKS = 8;
struct FilmRed { float v[2][16], w[2][8], x[2][4], y[2][2]; };
__device__ __forceinline__ void film_red_chunk(int c, FilmRed& R, const LaneBits& lb, float* ftp, int H) {
  if (c < 4) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int a = 2 * c; a < 2 * c + 2; ++a) R.w[s][a] = fold(lb.b0, R.v[s][2 * a], R.v[s][2 * a + 1], lane_xor1);
  } else if (c < 6) {
    const int s = c - 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) R.x[s][a] = fold(lb.b1, R.w[s][2 * a], R.w[s][2 * a + 1], lane_xor2);
  } else if (c == 6) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a) R.y[s][a] = fold(lb.b2, R.x[s][2 * a], R.x[s][2 * a + 1], lane_xor4);
  } else {
    float z[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      z[s] = fold(lb.b3, R.y[s][0], R.y[s][1], lane_xor8);   // lane i: 16-lane sum of register (i & 15)
      z[s] += lane_xor16(z[s]);
    }
    ftp[0] = z[0]; ftp[H] = z[1];     // lanes i and i ^ 16 hold the same sum and write it to the same address: no exec mask, no branch
  }
}
// feature offset inside an n-block of accumulator register (lane & 15) for the lane's half
__device__ __forceinline__ int film_lane_feature(int lane) { const int r = lane & 15; return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int NIN>
__device__ __forceinline__ void load_act(float (&in)[NIN], const float4* slab) {
#pragma unroll
  for (int g = 0; g < NIN / 4; ++g) {
    const float4 v = slab[g * 64];
    in[4 * g + 0] = v.x; in[4 * g + 1] = v.y; in[4 * g + 2] = v.z; in[4 * g + 3] = v.w;
  }
}

}  // namespace fenerf
