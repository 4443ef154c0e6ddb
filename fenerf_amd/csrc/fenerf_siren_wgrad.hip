// Parameter gradients of the FiLM-SIREN: the contractions over the point axis that follow the backward chain kernel
// (fenerf_siren_bwd.hip).  What torch autograd does in the reference with one addmm-backward + several elementwise /
// reduction kernels per layer (siren.py:113-123 FiLMLayer, :1509-1530) is here one MFMA kernel family that reads the two
// register-dump tapes (fenerf_layout.h "Tape") directly:
//
//     dL/dW_l [H x K_l] = sum_b diag(f_bl) * sum_{p in image b} dtheta_l[:, p] * in_l[:, p]^T
//     dL/dphase_bl = sum_p dtheta_l         dL/dfreq_bl = 15 sum_p dtheta_l * (W x + b)       dL/db_l = sum_b f_bl dL/dphase_bl
//
// with in_l = x_{l-1} = sin(2 pi (f' tape_{l-1} + p')) recomputed on the fly (bitwise the forward's activations), so no
// activation matrix and no f-scaled copy of dtheta ever exists in HBM.  A library GEMM is the wrong tool: M = N = 256 with
// K = 10^5..10^6 points gave 4 busy workgroups in rocBLAS (measured 33 TFLOP/s); here the point axis is split over
// workgroups (one image, one chunk of tiles each), every workgroup keeps a full 256x256 fp32 accumulator in AGPRs
// (v_mfma_f32_32x32x2_f32, 4 waves x 4x4 tiles) and writes one partial; a small second kernel sums the partials over chunks
// and images (applying f) straight into nn.Linear-layout gradient buffers -- deterministic, no atomics.
//
// Per 32-point tile a workgroup stages the operand rows through LDS ([feature][32 points], row stride 36 floats:
// conflict-free b128 reads): that is the transpose between the dump layout (lane = point) and the MFMA A/B layout
// (lane = feature, K = points).  FiLM sums fall out of the same staging: thread t owns row t.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_nt.h"
#include "fenerf_mfma32.h"

namespace fenerf {

constexpr int WG_LD = 36;   // LDS row stride (floats): 16-B aligned rows, conflict-free 128-bit reads across 8 rows

enum WgJob { WG_SQ = 0, WG_L0 = 1, WG_C0X = 2, WG_HEAD = 3, WG_RGB = 4 };

struct WgradParams {
  const float* tape; const float* d_t; const float* tape_e; const float* points; const float* dirs;
  const float* out; const float* d_out; const float* fp; const float* pp; const float* bias;
  const float* inv;            // f16x3 models: [L][H] result scale of the layer's GEMM (tape * inv = W x); nullptr = 1
  float box_scale;
  int B, L, n_geo, n_lab, C, H;
  long long P;                 // points per image (multiple of 32)
  int tiles_per_image, nchunk; // chunks per (job, image) of this launch
  int film_stride;             // chunk slots per (layer, image) in film_partial (>= every launch's nchunk)
  int layer0;                  // WG_SQ: first layer of the launch (blockIdx.z -> layer0 + z); other jobs: the layer
  float* partial;              // [z][b][chunk][MT*32][KT*32]
  float* film_partial;         // [L][b][chunk][H][2]: the chain kernel's per-tile FiLM sums gathered per chunk
  const float* film_tiles;     // [tiles][L][2][H] from the chain kernel (fenerf_layout.h "FiLM sums")
  int film16w;                 // the sums come from siren_bwd16w_kernel, register-dump order (fenerf_siren_bwd16w.hip): points per unit (16 / 128), 0 = no
  float* rowsum_partial;       // HEAD / RGB: [b][chunk][32]
  int bf16_dump;               // d_t holds the chain kernel's bf16 dump [d theta | x] (fenerf_layout.h "bf16 dump") instead of fp32 d theta
  int tape_u16;                // `tape` is the 16-bit tape (fenerf_layout.h "16-bit tape"): frac(theta) pieces instead of fp32 accumulators
  int freq_from_sums;          // the FiLM frequency gradients are derived from the weight-gradient partial sums (FENERF_TAPE_U16, FENERF_TAPE_F32_W)
  // Per-point modulation (round 6; SPATIALSIRENGRID under autograd, siren.py:440-477): fp / pp are [B*P][L][H].  The weight gradient of a
  // FiLM layer is then sum_p (f_l[p] (.) d theta_l[p]) x_{l-1}[p]^T -- the frequency cannot be pulled out of the sum over points -- so the
  // A side is scaled by the point's own 2 pi f' while it is staged, the B side's activations are recomputed with the point's own
  // (f', p'), the reductions apply no diag(f), and the FiLM-layer bias gradients are the row sums of the scaled A side.
  int film_per_point;
  float* bias_partial;         // film_per_point: [L][B][bias_stride][H] row sums of the scaled d theta (SQ: layers 1 .. L-1; L0: layer 0)
  int bias_stride;
};

// Stage one register-dump tile into LDS rows [H][WG_LD]; optional FiLM transform to activations.  The f' / p' rows are
// fetched (128-bit LDS reads) for the whole tile BEFORE the first write: LDS reads cannot be moved across LDS writes by the
// compiler, and one read-wait-write per element serialised the staging on LDS latency (measured: 35 % of the kernel).
// f_g / p_g (per-point modulation): this LANE's point's [H] rows of f' / p' in global memory instead of the image's rows in LDS;
// with !SIN (the d theta side) f_g scales the row by 2 pi f' = the point's own frequency (WgradParams::film_per_point).
template <int H, bool SIN, bool PW = false>
__device__ __forceinline__ void stage_dump(const float4 (&v)[H / 32], float* dst, int wave, int lane, const float* f_s, const float* p_s,
                                           const float* f_g = nullptr, const float* p_g = nullptr) {
  constexpr int NQ = H / 32;
  const float TWO_PI = 6.28318530717958647692f;
  const int m = lane & 31, half = lane >> 5;
  float4 f4[NQ], p4[NQ];
  if (SIN || PW) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = tape_feature(wave * NQ + q, half, 0);
      f4[q] = *reinterpret_cast<const float4*>((PW ? f_g : f_s) + row);
      if (SIN) p4[q] = *reinterpret_cast<const float4*>((PW ? p_g : p_s) + row);
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int row = tape_feature(wave * NQ + q, half, 0);
    float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
    if (SIN) {
      e[0] = sin2pi(__builtin_fmaf(f4[q].x, e[0], p4[q].x)); e[1] = sin2pi(__builtin_fmaf(f4[q].y, e[1], p4[q].y));
      e[2] = sin2pi(__builtin_fmaf(f4[q].z, e[2], p4[q].z)); e[3] = sin2pi(__builtin_fmaf(f4[q].w, e[3], p4[q].w));
    } else if (PW) {
      e[0] *= f4[q].x * TWO_PI; e[1] *= f4[q].y * TWO_PI; e[2] *= f4[q].z * TWO_PI; e[3] *= f4[q].w * TWO_PI;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[(row + i) * WG_LD + m] = e[i];
  }
}

template <int H>
__device__ __forceinline__ void load_dump(float4 (&v)[H / 32], const float4* src /* (tile, layer) base */, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < H / 32; ++q) v[q] = nt_load(src + (wave * (H / 32) + q) * 64 + lane);
}

// bf16 dump (fenerf_layout.h): the d theta half of a (tile32, layer) block is H*4 16-byte pieces; piece s = (nb, 16-point tile, lane
// (n, g)) holds 8 features of one point.  256 threads take pieces tid + 256 q.
template <int H>
struct Dump16 {
  static constexpr int PIECES = H * 4;                          // per operand half and 32-point tile
  static constexpr int PER_THREAD = (PIECES + 255) / 256;
};
template <int H>
__device__ __forceinline__ void load_dump16(uint4 (&v)[Dump16<H>::PER_THREAD], const char* half_base, int tid) {
#pragma unroll
  for (int q = 0; q < Dump16<H>::PER_THREAD; ++q) {
    const int s_ = tid + 256 * q;
    if (Dump16<H>::PIECES % 256 == 0 || s_ < Dump16<H>::PIECES)
      v[q] = __builtin_bit_cast(uint4, nt_load(reinterpret_cast<const float4*>(half_base) + s_));
  }
}
// -> fp32 rows [feature][WG_LD] (the exact-fp32 thin jobs): bf16 -> fp32 is a shift
template <int H>
__device__ __forceinline__ void stage_dump16_f32(const uint4 (&v)[Dump16<H>::PER_THREAD], float* dst, int tid) {
#pragma unroll
  for (int q = 0; q < Dump16<H>::PER_THREAD; ++q) {
    const int s_ = tid + 256 * q;
    if (Dump16<H>::PIECES % 256 != 0 && s_ >= Dump16<H>::PIECES) continue;
    const int nb = s_ >> 7, odd = (s_ >> 6) & 1, n = s_ & 15, g = (s_ >> 4) & 3;
    const unsigned w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const unsigned bits = (t & 1) ? (w[t >> 1] & 0xffff0000u) : (w[t >> 1] << 16);
      dst[dump16_feature(nb, g, t) * WG_LD + 16 * odd + n] = __builtin_bit_cast(float, bits);
    }
  }
}

// 16-bit tape (fenerf_layout.h): the same pieces hold frac(theta) as u16 -> activation rows x = sin(2 pi frac) (the HEAD / RGB jobs' B side)
template <int H>
__device__ __forceinline__ void stage_tape16_sin(const uint4 (&v)[Dump16<H>::PER_THREAD], float* dst, int tid) {
#pragma unroll
  for (int q = 0; q < Dump16<H>::PER_THREAD; ++q) {
    const int s_ = tid + 256 * q;
    if (Dump16<H>::PIECES % 256 != 0 && s_ >= Dump16<H>::PIECES) continue;
    const int nb = s_ >> 7, odd = (s_ >> 6) & 1, n = s_ & 15, g = (s_ >> 4) & 3;
    const unsigned w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float th = (float)((t & 1) ? (w[t >> 1] >> 16) : (w[t >> 1] & 0xffffu)) * (1.f / 65536.f);
      dst[dump16_feature(nb, g, t) * WG_LD + 16 * odd + n] = sin_rev_reduced(th);
    }
  }
}

// JOB: which operands.  MT x KT = output tiles (32x32) of the workgroup; WM x WK = tiles per wave.
template <int H, int JOB>
struct WgShape {
  static constexpr int NB = H / 32;
  static constexpr int MT = (JOB == WG_HEAD || JOB == WG_RGB) ? 1 : NB;
  static constexpr int KT = (JOB == WG_SQ || JOB == WG_HEAD || JOB == WG_RGB) ? NB : (JOB == WG_C0X ? 2 : 1);
  // Three-row operands (warped coordinates, view direction, rgb gradient rows) are contracted on the VALU -- thread = the H-side row,
  // three dot products over the tile's 32 points (~100 FMAs + 32 LDS reads per thread).  On the matrix pipe they are padded to a 32-row
  // tile: 128 exact-fp32 MFMAs = 2,048 cycles per wave and tile for 24,576 useful MACs.  L0 / RGB have no MFMA left; colour layer 0
  // keeps one column tile (the 32 grid features) instead of two.
  static constexpr bool V3 = (JOB == WG_L0 || JOB == WG_RGB || JOB == WG_C0X) && H <= 256;
  static constexpr bool V3_ONLY = V3 && JOB != WG_C0X;
  static constexpr bool V3_OWN_A = JOB != WG_RGB;                // the thread's own row comes from the A image (d theta); RGB: from B (x)
  static constexpr int V3_ROW0 = JOB == WG_C0X ? 32 : 0;         // first of the three rows in the other image
  static constexpr int KT_MFMA = (V3 && JOB == WG_C0X) ? 1 : KT;
  // wave grid: SQ 2x2 (big tiles); A-tall thin jobs 4x1; B-wide thin jobs 1x4
  static constexpr int WGM = (JOB == WG_SQ) ? 2 : ((MT >= 4) ? 4 : 1);
  static constexpr int WGK = 4 / WGM;
  static constexpr int WM = (MT + WGM - 1) / WGM;
  static constexpr int WK = (KT_MFMA + WGK - 1) / WGK;
  static constexpr bool A_DUMP = (JOB == WG_SQ || JOB == WG_L0 || JOB == WG_C0X);
  static constexpr bool B_DUMP = (JOB == WG_SQ || JOB == WG_HEAD || JOB == WG_RGB);
  static constexpr int A_ROWS = MT * 32, B_ROWS = KT * 32;
};

template <int H, int JOB, bool PW = false>     // PW: per-point FiLM parameters (WgradParams::film_per_point) -- its own instantiation: the per-image jobs keep their code
__device__ __forceinline__ void wgrad_job(const WgradParams& P, float* lds, int bz) {
  using S = WgShape<H, JOB>;
  constexpr int NQ = H / 32;                   // dump groups per wave
  float* A_s = lds;                                         // [A_ROWS][WG_LD]
  float* B_s = A_s + S::A_ROWS * WG_LD;                     // [B_ROWS][WG_LD]
  float* f_s = B_s + S::B_ROWS * WG_LD;                     // f', p' of the B-side layer
  float* p_s = f_s + H;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x, img = blockIdx.y;
  const int l = (JOB == WG_SQ) ? P.layer0 + bz : P.layer0;   // layer whose dtheta is the A side (SQ/L0/C0X)
  const int lb = (JOB == WG_SQ) ? l - 1 : ((JOB == WG_HEAD) ? P.n_geo - 1 : P.L - 1);   // B-side activation layer
  const int L = P.L, C = P.C;

  if (S::B_DUMP && !PW) for (int i = tid; i < H; i += 256) { f_s[i] = P.fp[((size_t)img * L + lb) * H + i]; p_s[i] = P.pp[((size_t)img * L + lb) * H + i]; }
  if (!S::A_DUMP) for (int i = tid; i < S::A_ROWS * WG_LD; i += 256) A_s[i] = 0.f;     // padded rows stay zero
  if (!S::B_DUMP) for (int i = tid; i < S::B_ROWS * WG_LD; i += 256) B_s[i] = 0.f;
  __syncthreads();

  const int t_per = (P.tiles_per_image + P.nchunk - 1) / P.nchunk;
  const int t0 = chunk * t_per, t1 = min(P.tiles_per_image, t0 + t_per);
  const long long tile_base = (long long)img * P.tiles_per_image;
  const long long tl = (long long)(H / 8) * 64;              // float4 per (tile, layer)
  const float4* tape4 = reinterpret_cast<const float4*>(P.tape);
  const float4* dt4 = reinterpret_cast<const float4*>(P.d_t);

  f32x16 acc[S::WM][S::WK];
#pragma unroll
  for (int a = 0; a < S::WM; ++a)
#pragma unroll
    for (int b = 0; b < S::WK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wm0 = (wave / S::WGK) * S::WM, wk0 = (wave % S::WGK) * S::WK;
  float s0 = 0.f;              // head row sums (thread = row)
  float sb = 0.f;              // film_per_point: row sum of the scaled d theta (thread = row): the FiLM layer's bias gradient
  float acc3[3] = {0.f, 0.f, 0.f};   // V3 jobs: this thread's three dot products

  float4 va[NQ], vb[NQ];
  uint4 va16[Dump16<H>::PER_THREAD], vb16[Dump16<H>::PER_THREAD];
  // The per-point side inputs of a tile (warped coordinates, grid features + view direction, output-gradient rows) are fetched into
  // registers ONE TILE AHEAD like the dumps (round 3: loaded inside the staging they put one HBM round trip per tile in front of the
  // barrier -- 24 tiles per workgroup, 20-40 % of its time).  Past the chunk's end the last tile is re-read (no branch around loads).
  float4 side4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float side_a[4] = {0.f, 0.f, 0.f, 0.f}, side_b = 0.f, side_c = 0.f;
  auto fetch = [&](int t) {
    const long long tile = tile_base + (t < t1 ? t : t1 - 1);
    const long long pt0 = tile * 32;
    if (S::A_DUMP) {
      if (P.bf16_dump) load_dump16<H>(va16, reinterpret_cast<const char*>(dt4 + (tile * L + l) * tl), tid);
      else load_dump<H>(va, dt4 + (tile * L + l) * tl, wave, lane);
    }
    if (S::B_DUMP) {
      if (P.tape_u16) load_dump16<H>(vb16, reinterpret_cast<const char*>(P.tape) + (tile * L + lb) * (long long)(H * 64), tid);
      else load_dump<H>(vb, tape4 + (tile * L + lb) * tl, wave, lane);
    }
    if (JOB == WG_L0) {
      if (tid < 96) side_b = P.points[(pt0 + (tid & 31)) * 3 + (tid >> 5)];
    }
    if (JOB == WG_C0X) {
      if (P.tape_e) side4 = *reinterpret_cast<const float4*>(P.tape_e + (pt0 + (tid >> 3)) * 32 + (tid & 7) * 4);
      if (tid < 96 && P.dirs) side_b = P.dirs[(pt0 + (tid & 31)) * 3 + (tid >> 5)];
    }
    if (JOB == WG_HEAD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = tid + 256 * q, r = i >> 5, m = i & 31;
        const int ch = r < P.n_lab ? r : (r == P.n_lab ? C - 1 : -1);
        side_a[q] = ch >= 0 ? P.d_out[(pt0 + m) * C + ch] : 0.f;
      }
    }
    if (JOB == WG_RGB) {
      if (tid < 96) {
        const int c = tid >> 5, m = tid & 31;
        side_b = P.out[(pt0 + m) * C + (C - 4) + c];
        side_c = P.d_out[(pt0 + m) * C + (C - 4) + c];
      }
    }
  };
  fetch(t0);
  for (int t = t0; t < t1; ++t) {
    // ---- stage the tile
    // per-point modulation: this lane's point of the tile being staged and its FiLM rows (layer l for the A side, lb for the B side)
    const size_t pw_pt = PW ? (size_t)((tile_base + t) * 32 + (lane & 31)) * L : 0;
    if (S::A_DUMP) {
      if (P.bf16_dump) stage_dump16_f32<H>(va16, A_s, tid);
      else stage_dump<H, false, PW>(va, A_s, wave, lane, nullptr, nullptr, PW ? P.fp + (pw_pt + l) * H : nullptr);
    }
    if (S::B_DUMP) {
      if (P.tape_u16) stage_tape16_sin<H>(vb16, B_s, tid);
      else stage_dump<H, true, PW>(vb, B_s, wave, lane, f_s, p_s, PW ? P.fp + (pw_pt + lb) * H : nullptr, PW ? P.pp + (pw_pt + lb) * H : nullptr);
    }
    if (JOB == WG_L0) {            // B rows 0..2 = warped coordinates
      if (tid < 96) B_s[(tid >> 5) * WG_LD + (tid & 31)] = side_b * P.box_scale;
    }
    if (JOB == WG_C0X) {           // B rows 0..31 = grid features, 32..34 = view direction
      if (P.tape_e) {
        const int m = tid >> 3, c4 = (tid & 7) * 4;
        B_s[(c4 + 0) * WG_LD + m] = side4.x; B_s[(c4 + 1) * WG_LD + m] = side4.y; B_s[(c4 + 2) * WG_LD + m] = side4.z; B_s[(c4 + 3) * WG_LD + m] = side4.w;
      }
      if (tid < 96) {
        const int c = tid >> 5, m = tid & 31;
        B_s[(32 + c) * WG_LD + m] = P.dirs ? side_b : (c == 2 ? -1.f : 0.f);
      }
    }
    if (JOB == WG_HEAD) {          // A row r = gradient wrt head row r: labels [0,n_lab), sigma n_lab
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = tid + 256 * q;
        A_s[(i >> 5) * WG_LD + (i & 31)] = side_a[q];
      }
    }
    if (JOB == WG_RGB) {
      if (tid < 96) A_s[(tid >> 5) * WG_LD + (tid & 31)] = side_c * (side_b * (1.f - side_b));
    }
    __syncthreads();
    fetch(t + 1);                          // next tile's global loads (dump and side inputs) fly behind this tile's MFMAs

    // ---- row sums (thread = row)
    if ((JOB == WG_SQ || JOB == WG_L0) && PW) {
      if (tid < H) {
        const float4* ar = reinterpret_cast<const float4*>(A_s + tid * WG_LD);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 a = ar[q]; sb += (a.x + a.y) + (a.z + a.w); }
      }
    }
    if (JOB == WG_HEAD || JOB == WG_RGB) {
      if (tid < 32) {
        const float4* ar = reinterpret_cast<const float4*>(A_s + tid * WG_LD);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 a = ar[q]; s0 += (a.x + a.y) + (a.z + a.w); }
      }
    }
    // ---- three-row operands on the VALU (WgShape::V3)
    if constexpr (S::V3) {
      if (tid < H) {
        const float4* own = reinterpret_cast<const float4*>((S::V3_OWN_A ? A_s : B_s) + tid * WG_LD);
        const float4* thin = reinterpret_cast<const float4*>((S::V3_OWN_A ? B_s : A_s) + S::V3_ROW0 * WG_LD);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 o = own[q];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float4 w = thin[c * (WG_LD / 4) + q];
            acc3[c] = __builtin_fmaf(o.x, w.x, acc3[c]); acc3[c] = __builtin_fmaf(o.y, w.y, acc3[c]);
            acc3[c] = __builtin_fmaf(o.z, w.z, acc3[c]); acc3[c] = __builtin_fmaf(o.w, w.w, acc3[c]);
          }
        }
      }
    }
    if constexpr (!S::V3_ONLY)
    // ---- MFMA: lane (i, kh) contracts points 16 kh + s, s = 0..15.  Operand fragments are read from LDS one
    //      (mt, kt) group ahead of their 16 MFMAs (1024 cycles), so no LDS latency is exposed.
    {
      const int i = lane & 31, kh = lane >> 5;
      auto frag = [&](const float* base, int tile_idx, float (&f)[16]) {
        const float4* r = reinterpret_cast<const float4*>(base + (tile_idx * 32 + i) * WG_LD + 16 * kh);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 v = r[q]; f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w; }
      };
      auto a_tile = [&](int mt) { return (wm0 + mt < S::MT) ? wm0 + mt : S::MT - 1; };   // waves beyond the tile grid recompute
      auto b_tile = [&](int kt) { return (wk0 + kt < S::KT_MFMA) ? wk0 + kt : S::KT_MFMA - 1; };   // the last tile (not stored)
      float a_cur[16], b_cur[16], a_nxt[16], b_nxt[16];
      frag(A_s, a_tile(0), a_cur);
      frag(B_s, b_tile(0), b_cur);
#pragma unroll
      for (int kt = 0; kt < S::WK; ++kt) {
#pragma unroll
        for (int mt = 0; mt < S::WM; ++mt) {
          const bool last_m = mt == S::WM - 1, last = last_m && kt == S::WK - 1;
          if (!last) frag(A_s, a_tile(last_m ? 0 : mt + 1), a_nxt);
          if (last_m && !last) frag(B_s, b_tile(kt + 1), b_nxt);
#pragma unroll
          for (int s = 0; s < 16; ++s) acc[mt][kt] = MFMA(a_cur[s], b_cur[s], acc[mt][kt]);
          __builtin_amdgcn_sched_barrier(0);   // pin (next reads, 16 MFMAs): keeps the register budget at 256 accumulators + 64
          if (!last) {
#pragma unroll
            for (int s = 0; s < 16; ++s) a_cur[s] = a_nxt[s];
          }
          if (last_m && !last) {
#pragma unroll
            for (int s = 0; s < 16; ++s) b_cur[s] = b_nxt[s];
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- partials
  {
    const int zi = (JOB == WG_SQ) ? bz : 0;
    float* out = P.partial + (((size_t)zi * P.B + img) * P.nchunk + chunk) * (size_t)(S::A_ROWS * S::B_ROWS);
    const int col = lane & 31, hh = lane >> 5;
    if constexpr (S::V3) {
      if (tid < H) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (S::V3_OWN_A) out[(size_t)tid * S::B_ROWS + S::V3_ROW0 + c] = acc3[c];   // [H rows][B_ROWS]: three columns (the reduction reads no padding)
          else out[(size_t)c * S::B_ROWS + tid] = acc3[c];                             // [32][H columns]: rows 0..2
        }
      }
    }
    if constexpr (!S::V3_ONLY) {
#pragma unroll
      for (int mt = 0; mt < S::WM; ++mt)
#pragma unroll
        for (int kt = 0; kt < S::WK; ++kt)
          if (wm0 + mt < S::MT && wk0 + kt < S::KT_MFMA)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (wm0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              out[(size_t)row * S::B_ROWS + (wk0 + kt) * 32 + col] = acc[mt][kt][r];
            }
    }
    if ((JOB == WG_HEAD || JOB == WG_RGB) && tid < 32) P.rowsum_partial[((size_t)img * P.nchunk + chunk) * 32 + tid] = s0;
    if ((JOB == WG_SQ || JOB == WG_L0) && PW && tid < H)
      P.bias_partial[(((size_t)l * P.B + img) * P.bias_stride + chunk) * H + tid] = sb;
  }
}

template <int H, int JOB, bool PW = false>
__global__ __launch_bounds__(256, 1) void siren_wgrad_kernel(WgradParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  wgrad_job<H, JOB, PW>(P, lds, blockIdx.z);
}

// The four thin jobs of one backward chunk in ONE launch (blockIdx.z = which): each is a chain of load -> stage -> barrier -> exact-fp32
// MFMAs per 32-point tile with one workgroup per CU and chunk, i.e. latency-bound (30-40 us each with the MFMAs removed); side by
// side their workgroups fill each other's waits.
struct ThinJobs { WgradParams j[4]; };
template <int H, bool PW = false>
__global__ __launch_bounds__(256, 1) void siren_wgrad_thin_kernel(ThinJobs T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  switch (blockIdx.z) {
    case 0: wgrad_job<H, WG_C0X, PW>(T.j[0], lds, 0); break;     // longest first
    case 1: wgrad_job<H, WG_HEAD, PW>(T.j[1], lds, 0); break;
    case 2: wgrad_job<H, WG_RGB, PW>(T.j[2], lds, 0); break;
    default: wgrad_job<H, WG_L0, PW>(T.j[3], lds, 0); break;
  }
}

// ------------------------------------------------------------------------------------------------
// Square weight-gradient job on the bf16 MFMA (v_mfma_f32_32x32x16_bf16), error-compensated like the forward's f16x3 GEMM:
// each fp32 operand is split into bf16 (hi, lo) -- hi = truncation, lo = the exact remainder rounded, together 16
// mantissa bits -- and a product is evaluated as ah*bh + ah*bl + al*bh (3 MFMAs at 16x the fp32 MFMA rate; dropped term
// and roundings ~2^-16 relative per product, unbiased, random over 10^5..10^6 points).  bf16 rather than fp16 because dtheta has
// no a-priori range.  Used for FENERF_PREC_F16X3 models; FENERF_PREC_F32 models keep the exact fp32 job above.
//
// LDS image: A_p / B_p rows [feature][hi: 32 points bf16 | lo: 32 points bf16 | pad], row stride WG_LD dwords (144 B) -- the
// lane's 8 consecutive points of a k-step are ONE 128-bit read per half, ready-made MFMA operands.  (Round 1 kept one
// split-packed dword per point and unpacked with v_perm_b32 at every fragment read: the B fragments are re-read for each of
// the wave's four row tiles, and the unpacking alone was 320 of the wave's 860 VALU instructions per tile on a kernel whose one
// wave per SIMD spent 60 % of its time issuing, profiles/r02_pmc_gstep_waits.txt.  Staging now writes the halves with 16-bit LDS
// stores -- the hi half is the upper half of the fp32 register as it is -- and splits with and / sub / v_cvt_pk_bf16_f32.)
// With the MFMA time cut 5x the kernel is bound by its two dump reads (dtheta_l, tape_{l-1}); the FiLM sums come from the
// chain kernel (film_gather_kernel).
// ------------------------------------------------------------------------------------------------
#ifndef FENERF_WGRAD_SETPRIO
#define FENERF_WGRAD_SETPRIO 0        // measured in round 6: no effect on this kernel (2.74 against 2.72-2.76 ms per step)
#endif
#ifndef FENERF_WGRAD_PAIR_STORES
#define FENERF_WGRAD_PAIR_STORES 1    // 0: rounds 2-5 (one 16-bit LDS store per value and half); A/B builds only
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
// four values of one point -> rows row .. row + 3 of an image: hi = bf16 truncation (the upper half of the fp32 word), lo = the
// exact remainder rounded to nearest (a truncated remainder would bias every product by 2^-17)
__device__ __forceinline__ void stage_split4(unsigned short* row0_m, const float (&v)[4]) {
  unsigned vb[4];
  float rem[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    vb[i] = __builtin_bit_cast(unsigned, v[i]);
    rem[i] = v[i] - __builtin_bit_cast(float, vb[i] & 0xffff0000u);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x2w rr = {rem[2 * j], rem[2 * j + 1]};
    const unsigned lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, bf16x2w));
    row0_m[(2 * j) * (2 * WG_LD)] = (unsigned short)(vb[2 * j] >> 16);
    row0_m[(2 * j + 1) * (2 * WG_LD)] = (unsigned short)(vb[2 * j + 1] >> 16);
    row0_m[(2 * j) * (2 * WG_LD) + 32] = (unsigned short)lo2;
    row0_m[(2 * j + 1) * (2 * WG_LD) + 32] = (unsigned short)(lo2 >> 16);
  }
}

// The same split for a PAIR of adjacent points (round 6): lanes m and m ^ 1 hold the same four rows of points m and m + 1.  They trade two
// values each (one DPP quad_perm), so that the even lane owns rows 0, 1 and the odd lane rows 2, 3 of BOTH points and writes them as
// whole dwords [even point | odd point]: 4 ds_write_b32 per lane instead of 8 ds_write_b16.  A 16-bit LDS store costs what a 32-bit one
// costs (address + data transfer, 64 B/clk at best), and with 512 of them per tile and workgroup the LDS -- 1,536 cycles of fragment
// reads + 2,048 of staging writes per tile against 3,072 cycles of MFMA work per SIMD -- was what bounded this kernel
// (SQ_WAIT_INST_LDS 21 %, matrix pipe 59 % busy, profiles/r06_pmc_gstep_waits.txt).  Same bits in the same places as stage_split4.
__device__ __forceinline__ float wg_lane_xor1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));   // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ void stage_split4_pair(unsigned* row0 /* dword 0 of row `row` of the image */, int m, const float (&v)[4]) {
  const bool odd = (m & 1) != 0;
  const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];        // what the partner lane's rows need from this point
  const float r0 = wg_lane_xor1(s0), r1 = wg_lane_xor1(s1);           // the partner point's values of THIS lane's rows
  const float o0 = odd ? v[2] : v[0], o1 = odd ? v[3] : v[1];        // this point's values of this lane's rows
  const float ev[2] = {odd ? r0 : o0, odd ? r1 : o1};                // even point (low half of the dword)
  const float od[2] = {odd ? o0 : r0, odd ? o1 : r1};                // odd point (high half)
  unsigned* dst = row0 + (odd ? 2 : 0) * WG_LD + (m >> 1);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned eb = __builtin_bit_cast(unsigned, ev[j]), ob = __builtin_bit_cast(unsigned, od[j]);
    const f32x2w rr = {ev[j] - __builtin_bit_cast(float, eb & 0xffff0000u), od[j] - __builtin_bit_cast(float, ob & 0xffff0000u)};
    dst[j * WG_LD] = __builtin_amdgcn_perm(ob, eb, 0x07060302u);                                          // hi: [od.hi16 | ev.hi16]
    dst[j * WG_LD + 16] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, bf16x2w));             // lo, 64 B behind
  }
}

struct Frag16 { bf16x8 hi, lo; };

// waves per workgroup of the kernel below (see its header)
template <int H> struct SqWaves { static constexpr int value = H >= 256 ? 8 : 4; };

// NW waves over the NB x NB output tiles.  H = 256: EIGHT waves (4 x 2; 2 x 4 tiles = 128 accumulator registers each, 246 in all), two
// per SIMD.  Rounds 1-2 ran four waves (2 x 2; 4 x 4 tiles = 256 accumulator AGPRs, one wave per SIMD): measured on one backward chunk,
// that kernel's time was its MFMA time PLUS its read time (no MFMAs: - 32 %; profiles/r03_wgrad_sq_experiment.txt) -- a single in-order
// wave per SIMD lets the matrix pipe and the memory pipe take turns.  With a second wave on the SIMD one multiplies while the other
// stages or waits: 1.00 -> 0.91 ms on noise gradients, 0.75 -> 0.68 ms per chunk inside the generator step (5.4 -> 5.9 TB/s), same LDS
// traffic (fragment reads per MFMA unchanged), bit-identical sums per output element (same k order).  H < 256: four waves.
// The LDS image is double-buffered: while the MFMAs of tile t read buffer t & 1, tile t + 1 (already in registers) is staged
// -- sin, split, LDS writes -- into the other buffer, half a dump group at a time BETWEEN the dependent MFMAs of each
// accumulator group, and every dump group's registers are refilled with tile t + 2 as soon as they have been staged: one
// barrier per tile, loads in flight for a whole tile period.  (Staged in a phase of its own, the kernel spent 30 % of a
// tile in staging and another 30 % waiting for loads that had only the MFMA phase to arrive.)
// T16 (round 5): the B side comes from the 16-bit tape -- 8-byte half-pieces (one point, the four features of a row tile), x =
// sin(2 pi u / 65536): no FiLM rows, no fma, half the bytes of that stream.
template <int H, int NW, bool T16>
__global__ __launch_bounds__(NW * 64, 1) void siren_wgrad_sq_bf16_kernel(WgradParams P) {
  constexpr int NB = H / 32, NG = H / 8;                  // output tiles per side; dump groups per tile
  constexpr int GPW = NG >= NW ? NG / NW : 1;             // dump groups staged per wave
  constexpr int WGM = NW / 2, WGK = 2;                    // wave grid over the NB x NB output tiles
  constexpr int WM = (NB + WGM - 1) / WGM, WK = (NB + 1) / 2;
  constexpr int NGROUP = WM * WK;                         // accumulator groups (6 dependent MFMAs each) per wave and tile
  constexpr int HPG = (2 * GPW + NGROUP - 1) / NGROUP;    // staging half-pieces (a dump group's dtheta or tape rows) per group
  constexpr int IMG = 2 * H * WG_LD;                      // dwords per buffer: [A rows | B rows]
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* img0 = reinterpret_cast<unsigned*>(lds);         // 2 x { A_p [H][WG_LD] split-packed dtheta_l, B_p [H][WG_LD] split-packed x_{l-1} }
  float* f_s = reinterpret_cast<float*>(img0 + 2 * IMG);
  float* p_s = f_s + H;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x, img = blockIdx.y;
  const int l = P.layer0 + blockIdx.z, lb = l - 1;
  const int L = P.L;
  for (int i = tid; i < H; i += NW * 64) { f_s[i] = P.fp[((size_t)img * L + lb) * H + i]; p_s[i] = P.pp[((size_t)img * L + lb) * H + i]; }
  __syncthreads();

  const int t_per = (P.tiles_per_image + P.nchunk - 1) / P.nchunk;
  const int t0 = chunk * t_per, t1 = min(P.tiles_per_image, t0 + t_per);
  const long long tile_base = (long long)img * P.tiles_per_image;
  const long long tl = (long long)NG * 64;
  const float4* tape4 = reinterpret_cast<const float4*>(P.tape);
  const float4* dt4 = reinterpret_cast<const float4*>(P.d_t);

  f32x16 acc[WM][WK];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wm0 = (wave / WGK) * WM, wk0 = (wave % WGK) * WK;
  const int m = lane & 31, half = lane >> 5;
  static_assert(NW * GPW == NG, "every wave stages GPW dump groups");
#if FENERF_WGRAD_SETPRIO
  if (NW == 8 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif

  float4 va[GPW], vb[GPW];                                  // dump group q of the tile being staged next: dtheta_l, tape_{l-1}
  uint2 vb16[GPW];                                          // T16: half-piece tid + NW * 64 * q of the tile's 16-bit tape block
  // No branches from here on: a branch inside the MFMA stream makes the compiler's vmcnt bookkeeping conservative (it
  // then waits for the refill loads it has just issued); past the chunk's end the last tile is re-read and never staged.
  auto fetch_q = [&](int t, int q) {
    const long long tile = tile_base + (t < t1 ? t : t1 - 1);
    const int g = wave * GPW + q;
    va[q] = nt_load(dt4 + (tile * L + l) * tl + g * 64 + lane);
    if constexpr (T16) {
      const float2 w = nt_load(reinterpret_cast<const float2*>(reinterpret_cast<const char*>(P.tape) + (tile * L + lb) * (long long)(H * 64)) + tid + NW * 64 * q);
      vb16[q] = __builtin_bit_cast(uint2, w);
    } else {
      vb[q] = nt_load(tape4 + (tile * L + lb) * tl + g * 64 + lane);
    }
  };
  // half-piece hp = 2 q + part of the tile in (va, vb) -> buffer dst: part 0 = dtheta rows, part 1 = x = sin(2 pi (f' tape + p')) rows.
  // f4 / p4 = the FiLM rows of the group, fetched from LDS ahead of time (LDS reads do not move across LDS writes).
  auto film_rows = [&](int hp, float4& f4, float4& p4) {
    if constexpr (T16) return;                              // the phase is in the tape: no FiLM rows
    const int row = tape_feature(wave * GPW + (hp >> 1), half, 0);
    f4 = *reinterpret_cast<const float4*>(f_s + row);
    p4 = *reinterpret_cast<const float4*>(p_s + row);
  };
  auto stage_half = [&](int hp, unsigned* dst, const float4& f4, const float4& p4) {
    const int q = hp >> 1;
    const int row = tape_feature(wave * GPW + q, half, 0);
    if ((hp & 1) == 0) {
      const float4 a = va[q];
      const float d[4] = {a.x, a.y, a.z, a.w};
#if FENERF_WGRAD_PAIR_STORES
      stage_split4_pair(dst + row * WG_LD, m, d);
#else
      stage_split4(reinterpret_cast<unsigned short*>(dst + row * WG_LD) + m, d);
#endif
    } else if constexpr (T16) {
      // half-piece hs = (16-byte piece s16 = (nb, 16-point tile, lane (n, g)), row tile rt): features dump16_feature(nb, g, 4 rt + r)
      const int hs = tid + NW * 64 * q, s16 = hs >> 1, rt = hs & 1;
      const int nb = s16 >> 7, odd = (s16 >> 6) & 1, n = s16 & 15, g = (s16 >> 4) & 3;
      const unsigned w0 = vb16[q].x, w1 = vb16[q].y;
      const float k = 1.f / 65536.f;
      const float x[4] = {sin_rev_reduced((float)(w0 & 0xffffu) * k), sin_rev_reduced((float)(w0 >> 16) * k),
                          sin_rev_reduced((float)(w1 & 0xffffu) * k), sin_rev_reduced((float)(w1 >> 16) * k)};
      stage_split4(reinterpret_cast<unsigned short*>(dst + H * WG_LD + dump16_feature(nb, g, 4 * rt) * WG_LD) + 16 * odd + n, x);
    } else {
      const float4 b = vb[q];
      const float x[4] = {sin2pi(__builtin_fmaf(f4.x, b.x, p4.x)), sin2pi(__builtin_fmaf(f4.y, b.y, p4.y)),
                          sin2pi(__builtin_fmaf(f4.z, b.z, p4.z)), sin2pi(__builtin_fmaf(f4.w, b.w, p4.w))};
#if FENERF_WGRAD_PAIR_STORES
      stage_split4_pair(dst + H * WG_LD + row * WG_LD, m, x);
#else
      stage_split4(reinterpret_cast<unsigned short*>(dst + H * WG_LD + row * WG_LD) + m, x);
#endif
    }
  };

  // ---- prologue: tile t0 staged into buffer 0, tile t0 + 1 on its way
#pragma unroll
  for (int q = 0; q < GPW; ++q) fetch_q(t0, q);
#pragma unroll
  for (int hp = 0; hp < 2 * GPW; ++hp) {
    float4 f4, p4;
    film_rows(hp, f4, p4);
    stage_half(hp, img0, f4, p4);
  }
#pragma unroll
  for (int q = 0; q < GPW; ++q) fetch_q(t0 + 1, q);
  __syncthreads();

  for (int t = t0; t < t1; ++t) {
    const unsigned* A_p = img0 + ((t - t0) & 1) * IMG;
    const unsigned* B_p = A_p + H * WG_LD;
    unsigned* nxt = img0 + (((t - t0) & 1) ^ 1) * IMG;   // after the last tile this stages a re-read tile nobody consumes
    // ---- MFMA: lane (i, kh) contracts points 16 ks + 8 kh + {0..7} in k-step ks (same order on both operands)
    {
      const int i = lane & 31, kh = lane >> 5;
      auto a_tile = [&](int mt) { return (wm0 + mt < NB) ? wm0 + mt : NB - 1; };   // waves beyond the tile grid recompute the
      auto b_tile = [&](int kt) { return (wk0 + kt < NB) ? wk0 + kt : NB - 1; };   // last tile (not stored)
      // Fragments are re-read from LDS per tile pair rather than cached (no room beside 256 accumulators), software-pipelined:
      // the fragments of group (mt, kt + 1) are fetched behind the first MFMA of group (mt, kt) -- the six MFMAs of a group are
      // dependent (same accumulator), so whatever sits between them is free.
      auto frag = [&](const unsigned* base, int tile_idx, int ks) {   // row = 9 x 16 B: [hi points 0..31 | lo points 0..31 | pad]
        const uint4* p = reinterpret_cast<const uint4*>(base + (tile_idx * 32 + i) * WG_LD);
        Frag16 f;
        f.hi = __builtin_bit_cast(bf16x8, p[2 * ks + kh]);
        f.lo = __builtin_bit_cast(bf16x8, p[4 + 2 * ks + kh]);
        return f;
      };
      Frag16 bf[2] = {frag(B_p, b_tile(0), 0), frag(B_p, b_tile(0), 1)};
#pragma unroll
      for (int mt = 0; mt < WM; ++mt) {
        const Frag16 af[2] = {frag(A_p, a_tile(mt), 0), frag(A_p, a_tile(mt), 1)};
#pragma unroll
        for (int kt = 0; kt < WK; ++kt) {
          const bool last = mt == WM - 1 && kt == WK - 1;
          const int g = mt * WK + kt;
          Frag16 bn[2];
          float4 f4[HPG], p4[HPG];
          // sched_barriers pin this order: left alone the scheduler hoists the unpack right behind the loads
          acc[mt][kt] = MFMA_BF16(af[0].lo, bf[0].hi, acc[mt][kt]);
          if (!last) {
            const int nt = b_tile(kt + 1 < WK ? kt + 1 : 0);
            bn[0] = frag(B_p, nt, 0); bn[1] = frag(B_p, nt, 1);
          }
#pragma unroll
          for (int j = 0; j < HPG; ++j)
            if (g * HPG + j < 2 * GPW) film_rows(g * HPG + j, f4[j], p4[j]);
          __builtin_amdgcn_sched_barrier(0);
          // The staging of this group (sin / split, 16-bit LDS stores, the refill loads) is spread behind ALL five remaining MFMAs:
          // issued as one block behind the second it left the matrix pipe idle for ~130 cycles per group (0.600 -> 0.561 ms per
          // launch, same box; the exact split -- 5..12 VALU per slot, the fragment reads spread too -- makes no difference).
          acc[mt][kt] = MFMA_BF16(af[0].hi, bf[0].lo, acc[mt][kt]);
#pragma unroll
          for (int j = 0; j < HPG; ++j) {
            const int hp = g * HPG + j;
            if (hp < 2 * GPW) {
              stage_half(hp, nxt, f4[j], p4[j]);
              if (hp & 1) fetch_q(t + 2, hp >> 1);          // both halves of dump group hp >> 1 are staged: refill its registers
            }
          }
          acc[mt][kt] = MFMA_BF16(af[0].hi, bf[0].hi, acc[mt][kt]);
          acc[mt][kt] = MFMA_BF16(af[1].lo, bf[1].hi, acc[mt][kt]);
          acc[mt][kt] = MFMA_BF16(af[1].hi, bf[1].lo, acc[mt][kt]);
          acc[mt][kt] = MFMA_BF16(af[1].hi, bf[1].hi, acc[mt][kt]);
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);    // up to 9 VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);    // up to 2 LDS writes
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // up to 1 global load
          }
          if (!last) { bf[0] = bn[0]; bf[1] = bn[1]; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();
  }

  // ---- partials (same layout as the fp32 job)
  float* out = P.partial + (((size_t)blockIdx.z * P.B + img) * P.nchunk + chunk) * (size_t)(H * H);
  const int col = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < WM; ++mt)
#pragma unroll
    for (int kt = 0; kt < WK; ++kt)
      if (wm0 + mt < NB && wk0 + kt < NB)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          out[(size_t)row * H + (wk0 + kt) * 32 + col] = acc[mt][kt][r];
        }
}

// ------------------------------------------------------------------------------------------------
// Square weight-gradient job on the chain kernel's bf16 dump (fenerf_layout.h "bf16 dump"; backward chunks of >=
// FENERF_BF16_DUMP_MIN_POINTS points): d theta_l and x_{l-1} arrive as bf16, a product is ONE v_mfma_f32_32x32x16_bf16, nothing is
// recomputed (no sin, no split) and the tape is not read: 2 + 2 bytes per (point, feature), the kernel is a pure two-stream reader.
// Same workgroup shape, partial layout and reduction as the kernels above.  LDS image per operand: [feature][32 points bf16 | pad],
// row stride 80 B (an odd multiple of 16 B: the ds_read_b128 fragment reads of 16 consecutive rows hit 16 different 16-byte bank
// groups); double-buffered; the transpose between the dump (lane = point, slots = features) and the MFMA operand (lane = feature,
// slots = points) is the staging's 16-bit LDS stores -- lower / upper half of a register as they are (ds_write_b16 / _d16_hi).
// ------------------------------------------------------------------------------------------------
constexpr int WD_LD = 40;   // u16 per image row
template <int H>
__global__ __launch_bounds__(256, 1) void siren_wgrad_sq_b16d_kernel(WgradParams P) {
  constexpr int NB = H / 32;
  constexpr int WGK = 2, WM = (NB + 1) / 2, WK = (NB + 1) / 2;
  constexpr int NGROUP = WM * WK;                         // accumulator groups (2 dependent MFMAs each) per wave and tile
  constexpr int NP = Dump16<H>::PER_THREAD;               // 16-byte pieces per thread, operand and tile
  constexpr int PPG = (2 * NP + NGROUP - 1) / NGROUP;     // pieces staged per group
  constexpr int OPER = H * WD_LD;                         // u16 per operand image
  constexpr long long TLB = (long long)H * 128;           // bytes of a (tile32, layer) dump block
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned short* img0 = reinterpret_cast<unsigned short*>(lds);     // 2 buffers x { A: d theta_l rows, B: x_{l-1} rows }

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x, img = blockIdx.y;
  const int l = P.layer0 + blockIdx.z, lb = l - 1;
  const int L = P.L;

  const int t_per = (P.tiles_per_image + P.nchunk - 1) / P.nchunk;
  const int t0 = chunk * t_per, t1 = min(P.tiles_per_image, t0 + t_per);
  const long long tile_base = (long long)img * P.tiles_per_image;
  const char* dump = reinterpret_cast<const char*>(P.d_t);

  f32x16 acc[WM][WK];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wm0 = (wave / WGK) * WM, wk0 = (wave % WGK) * WK;

  // Two register sets of dump pieces: while tile t multiplies out of LDS buffer (t - t0) & 1, tile t + 1 is staged from set
  // (t - t0) & 1 into the other buffer and every staged piece's registers are refilled with tile t + 3 -- loads stay in flight for
  // TWO tile periods (a tile is only 32 KiB per workgroup here: one period of ~1.5 us would not cover the loaded HBM latency).
  uint4 va[2][NP], vb[2][NP];
  // No branches inside a tile (fenerf_siren_wgrad.hip, the kernel above): past the chunk's end the last tile is re-read.
  auto fetch_q = [&](auto set_c, int t, int q) {
    constexpr int SET = decltype(set_c)::value;
    const long long tile = tile_base + (t < t1 ? t : t1 - 1);
    const int s_ = tid + 256 * q;
    if (Dump16<H>::PIECES % 256 == 0 || s_ < Dump16<H>::PIECES) {
      va[SET][q] = __builtin_bit_cast(uint4, nt_load(reinterpret_cast<const float4*>(dump + (tile * L + l) * TLB) + s_));
      vb[SET][q] = __builtin_bit_cast(uint4, nt_load(reinterpret_cast<const float4*>(dump + (tile * L + lb) * TLB + TLB / 2) + s_));
    }
  };
  // piece pc = 2 q + which (0 = d theta, 1 = x) of the tile in register set SET -> buffer dst
  auto stage_piece = [&](auto set_c, int pc, unsigned short* dst) {
    constexpr int SET = decltype(set_c)::value;
    const int q = pc >> 1;
    const int s_ = tid + 256 * q;
    if (Dump16<H>::PIECES % 256 != 0 && s_ >= Dump16<H>::PIECES) return;
    const int nb = s_ >> 7, odd = (s_ >> 6) & 1, n = s_ & 15, g = (s_ >> 4) & 3;
    const uint4 v = (pc & 1) ? vb[SET][q] : va[SET][q];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned short* col = dst + ((pc & 1) ? OPER : 0) + 16 * odd + n;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      col[dump16_feature(nb, g, t) * WD_LD] = (t & 1) ? (unsigned short)(w[t >> 1] >> 16) : (unsigned short)w[t >> 1];
  };
  const std::integral_constant<int, 0> set0;
  const std::integral_constant<int, 1> set1;

  // ---- prologue: tile t0 staged into buffer 0; tiles t0 + 1 / t0 + 2 on their way in sets 0 / 1
#pragma unroll
  for (int q = 0; q < NP; ++q) fetch_q(set0, t0, q);
#pragma unroll
  for (int pc = 0; pc < 2 * NP; ++pc) stage_piece(set0, pc, img0);
#pragma unroll
  for (int q = 0; q < NP; ++q) fetch_q(set0, t0 + 1, q);
#pragma unroll
  for (int q = 0; q < NP; ++q) fetch_q(set1, t0 + 2, q);
  __syncthreads();

  const int i = lane & 31, kh = lane >> 5;
  auto a_tile = [&](int mt) { return (wm0 + mt < NB) ? wm0 + mt : NB - 1; };   // waves beyond the tile grid recompute the
  auto b_tile = [&](int kt) { return (wk0 + kt < NB) ? wk0 + kt : NB - 1; };   // last tile (not stored)
  // tile t out of buffer PAR = (t - t0) & 1; stages tile t + 1 from register set PAR, refills it with tile t + 3
  auto tile_body = [&](auto par_c, int t) {
    constexpr int PAR = decltype(par_c)::value;
    const unsigned short* A_p = img0 + PAR * (2 * OPER);
    const unsigned short* B_p = A_p + OPER;
    unsigned short* nxt = img0 + (PAR ^ 1) * (2 * OPER);   // after the last tile this stages a re-read tile nobody consumes
    // lane (i, kh) of k-step ks holds points 16 ks + 8 kh .. + 7 of feature row i: one 16-byte read
    auto frag = [&](const unsigned short* base, int tile_idx, int ks) {
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + (tile_idx * 32 + i) * WD_LD + 16 * ks + 8 * kh));
    };
    bf16x8 bf[2] = {frag(B_p, b_tile(0), 0), frag(B_p, b_tile(0), 1)};
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
      const bf16x8 af[2] = {frag(A_p, a_tile(mt), 0), frag(A_p, a_tile(mt), 1)};
#pragma unroll
      for (int kt = 0; kt < WK; ++kt) {
        const bool last = mt == WM - 1 && kt == WK - 1;
        const int g = mt * WK + kt;
        bf16x8 bn[2];
        acc[mt][kt] = MFMA_BF16(af[0], bf[0], acc[mt][kt]);
        if (!last) {
          const int nt = b_tile(kt + 1 < WK ? kt + 1 : 0);
          bn[0] = frag(B_p, nt, 0); bn[1] = frag(B_p, nt, 1);
        }
#pragma unroll
        for (int j = 0; j < PPG; ++j) {
          const int pc = g * PPG + j;
          if (pc < 2 * NP) {
            stage_piece(par_c, pc, nxt);
            if (pc & 1) fetch_q(par_c, t + 3, pc >> 1);          // both operands' piece q are staged: refill its registers
          }
        }
        acc[mt][kt] = MFMA_BF16(af[1], bf[1], acc[mt][kt]);
        if (!last) { bf[0] = bn[0]; bf[1] = bn[1]; }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };
  int t = t0;
  for (; t + 1 < t1; t += 2) {
    tile_body(set0, t);
    tile_body(set1, t + 1);
  }
  if (t < t1) tile_body(set0, t);      // odd tile count: the pairs consumed an even number, so the tail has parity 0

  // ---- partials (same layout as the jobs above)
  float* out = P.partial + (((size_t)blockIdx.z * P.B + img) * P.nchunk + chunk) * (size_t)(H * H);
  const int col = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < WM; ++mt)
#pragma unroll
    for (int kt = 0; kt < WK; ++kt)
      if (wm0 + mt < NB && wk0 + kt < NB)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          out[(size_t)row * H + (wk0 + kt) * 32 + col] = acc[mt][kt][r];
        }
}

// FiLM sums: the chain kernel left s0 = sum_p dtheta, s1 = sum_p dtheta * tape per (tile, layer, feature); this gathers
// them per (layer, image, chunk of tiles) into film_partial -- with d theta / d f = W x + b = tape * inv + bias applied --
// for film_reduce_kernel.  Inversion (inverse_render_double_semantic.py:324-350 optimises only the FiLM frequencies /
// phases) needs nothing else from the weight-gradient stage.
__global__ __launch_bounds__(256) void film_gather_kernel(WgradParams P) {
  const int H = P.H, L = P.L;
  const int chunk = blockIdx.x, img = blockIdx.y, l = blockIdx.z;
  const int t_per = (P.tiles_per_image + P.nchunk - 1) / P.nchunk;
  const int t0 = chunk * t_per, t1 = min(P.tiles_per_image, t0 + t_per);
  const long long tile_base = (long long)img * P.tiles_per_image;
  for (int n = threadIdx.x; n < H; n += blockDim.x) {
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;     // two independent chains: the loads are latency-bound
    if (P.film16w) {
      // [unit][layer][nb][rt][slot = 4 g + r][s0, s1] with feature = 32 nb + 16 (g >> 1) + 4 (g & 1) + 8 rt + r; unit = 16 or 128 points
      const int f = n & 31, gq = ((f >> 4) << 1) | ((f >> 2) & 1), rt = (f >> 3) & 1, r = f & 3;
      const int idx = (n >> 5) * 64 + rt * 32 + (gq * 4 + r) * 2;
      const long long stride = (long long)L * 2 * H;
      // this chunk's units: 32-point tiles [t0, t1) hold two 16-point units each; 128-point units are split evenly over the chunks
      long long u0, u1, ubase;
      if (P.film16w == 16) { ubase = tile_base * 2; u0 = 2LL * t0; u1 = 2LL * t1; }
      else {
        const long long nu = (P.P + 127) / 128, per = (nu + P.nchunk - 1) / P.nchunk;
        ubase = (long long)img * nu; u0 = chunk * per; u1 = u0 + per < nu ? u0 + per : nu;
      }
      const float* p0 = P.film_tiles + ((ubase + u0) * L + l) * 2LL * H + idx;
      long long u = u0;
      for (; u + 2 <= u1; u += 2, p0 += 2 * stride) {
        const float2 x = *reinterpret_cast<const float2*>(p0), v = *reinterpret_cast<const float2*>(p0 + stride);
        a0 += x.x; a1 += x.y; b0 += v.x; b1 += v.y;
      }
      if (u < u1) { const float2 x = *reinterpret_cast<const float2*>(p0); a0 += x.x; a1 += x.y; }
    } else {
    int t = t0;
    for (; t + 2 <= t1; t += 2) {
      const float* p0 = P.film_tiles + ((tile_base + t) * L + l) * 2LL * H + n;
      const float* p1 = p0 + (long long)L * 2 * H;
      a0 += p0[0]; a1 += p0[H]; b0 += p1[0]; b1 += p1[H];
    }
    if (t < t1) { const float* p0 = P.film_tiles + ((tile_base + t) * L + l) * 2LL * H + n; a0 += p0[0]; a1 += p0[H]; }
    }
    const float s0 = a0 + b0, s1 = a1 + b1;
    const float iv = P.inv ? P.inv[(size_t)l * H + n] : 1.f, bb = P.bias[(size_t)l * H + n];
    float* fpart = P.film_partial + ((((size_t)l * P.B + img) * P.film_stride + chunk) * H + n) * 2;
    fpart[0] = s0; fpart[1] = __builtin_fmaf(s1, iv, s0 * bb);
  }
}

// dst[r][dst_col0 + c] = sum_b scale(b, r) * sum_chunk src[(b, chunk)][r][src_col0 + c]; scale = 2 pi f'[b][layer][r] or 1
// NC independent partial sums, fixed summation order (deterministic).  The thin jobs' reductions are latency-bound (few elements,
// up to 256 chunks each: 16 chains); the square job's is bandwidth-bound (64 MB: 4 chains, 14 us -- with 16 planes open per
// thread it drops to 20).
template <int NC>
__device__ __forceinline__ float sum_chunks(const float* src, size_t stride, int nchunk) {
  float s[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) s[j] = 0.f;
  int k = 0;
  for (; k + NC <= nchunk; k += NC) {
#pragma unroll
    for (int j = 0; j < NC; ++j) s[j] += src[(size_t)(k + j) * stride];
  }
  for (; k < nchunk; ++k) s[0] += src[(size_t)k * stride];
#pragma unroll
  for (int w = NC / 2; w >= 1; w >>= 1)
#pragma unroll
    for (int j = 0; j < w; ++j) s[j] += s[j + w];
  return s[0];
}

// all square jobs in one launch: blockIdx.y = layer - 1; destination by layer (nn.Linear layout, FenerfSirenGrads)
//
// Frequency gradients without the tape (FREQ; round 5, the 16-bit tape).  dL/dfreq_raw[b][l][n] = 15 sum_p d theta_l[n][p] Z_l[n][p] with
// Z = W_l x_{l-1} + b_l the layer's pre-activation, which the chain kernel forms from the fp32 tape's accumulator (its second FiLM sum).
// A tape of phases has no accumulator -- but the per-image partial sums of this very reduction are G_b[n][k] = sum_p d theta_l[n][p]
// x_{l-1}[k][p], and  sum_p d theta Z = sum_k W_l[n][k] G_b[n][k] + b_l[n] sum_p d theta_l[n][p]  exactly (the second sum is the phase
// gradient, already reduced by film_reduce_kernel).  So the thread of element (n, k) multiplies its image sum with W_l[n][k] and the
// row's threads add up (LDS tree, fixed order: deterministic): one extra fma and a row reduction on data that is in registers anyway.
// `w` = the FiLM layers' weights [dev], nn.Linear layout (geo_w / color_w fields).  Colour layer 0's view-direction and grid-feature
// columns and layer 0 come from the thin jobs' partials (film_freq_thin_kernel, behind the thin reduction).
template <bool FREQ>
__global__ __launch_bounds__(256) void wgrad_reduce_sq_kernel(FenerfSirenGrads g, FenerfSirenGrads w, const float* sq, int B, int nchunk, const float* fp,
                                                              const float* inv, const float* bias, int L, int H, int n_geo, int grid_ch,
                                                              int film_per_point = 0 /* the partials already carry the points' frequencies */) {
  __shared__ float red[256];
  const float TWO_PI = 6.28318530717958647692f;
  const int l = blockIdx.y + 1;
  const float* src = sq + (size_t)(l - 1) * B * nchunk * H * H;
  float* dst; const float* wl = nullptr; int ld, col0 = 0;
  if (l < n_geo) { dst = g.geo_w[l]; ld = H; if (FREQ) wl = w.geo_w[l]; }
  else if (l == n_geo) { dst = g.color_w[0]; ld = 3 + grid_ch + H; col0 = 3 + grid_ch; if (FREQ) wl = w.color_w[0]; }
  else { dst = g.color_w[l - n_geo]; ld = H; if (FREQ) wl = w.color_w[l - n_geo]; }
  const int n_color = L - n_geo;
  // one workgroup per output row r (H <= 256 columns: thread = column, the rest idle), so that the row sum of the frequency gradient is a
  // reduction over the whole workgroup whatever H is (round 5 first cut: 256 consecutive elements per workgroup and a tree over H
  // threads -- right only for H a power of two; H = 96 / 192 put rows across workgroups)
  const int r = blockIdx.x, c = threadIdx.x;
  const bool active = c < H;
  const float wv = (FREQ && active) ? wl[(size_t)r * ld + col0 + c] : 0.f;
  float sum = 0.f;
  for (int b = 0; b < B; ++b) {
    const float s = active ? sum_chunks<4>(src + ((size_t)b * nchunk * H + r) * H + c, (size_t)H * H, nchunk) : 0.f;
    sum += film_per_point ? s : s * (fp[((size_t)b * L + l) * H + r] * TWO_PI / (inv ? inv[(size_t)l * H + r] : 1.f));
    if (FREQ) {
      red[c] = s * wv;
      __syncthreads();
      for (int o = 128; o >= 1; o >>= 1) {
        if (c < o) red[c] += red[c + o];
        __syncthreads();
      }
      if (c == 0) {
        float* df = l < n_geo ? g.d_freq_geo + ((size_t)b * n_geo + l) * H + r : g.d_freq_app + ((size_t)b * n_color + (l - n_geo)) * H + r;
        const float* dp = l < n_geo ? g.d_phase_geo + ((size_t)b * n_geo + l) * H + r : g.d_phase_app + ((size_t)b * n_color + (l - n_geo)) * H + r;
        *df = 15.f * __builtin_fmaf(bias[(size_t)l * H + r], *dp, red[0]);
      }
      __syncthreads();
    }
  }
  if (active) dst[(size_t)r * ld + col0 + c] = sum;
}

// The thin jobs' share of the frequency gradients (16-bit tape; see wgrad_reduce_sq_kernel): layer 0 (three warped-coordinate columns,
// the L0 job's partials [B][nt][H x 32]) and colour layer 0's view-direction + grid-feature columns (the C0X job's [B][nt][H x 64]: grid
// features in columns 0 .. 31, view direction in 32 .. 34) -- added to what the square reduction left for that layer.  One wave per
// (row, image): lane = column, 16 load chains per lane over the chunks, a fixed-order shuffle reduction.
__global__ __launch_bounds__(64) void film_freq_thin_kernel(FenerfSirenGrads g, FenerfSirenGrads w, const float* p_l0, const float* p_c0, int nt,
                                                            const float* bias, int H, int n_geo, int n_color, int grid_ch) {
  const int r = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
  const int ldc = 3 + grid_ch + H;
  float v0 = 0.f, v1 = 0.f;
  if (c < 3) {
    v0 = w.geo_w[0][r * 3 + c] * sum_chunks<16>(p_l0 + ((size_t)b * nt * H + r) * 32 + c, (size_t)H * 32, nt);
  } else if (c < 6) {
    v1 = w.color_w[0][(size_t)r * ldc + (c - 3)] * sum_chunks<16>(p_c0 + ((size_t)b * nt * H + r) * 64 + 32 + (c - 3), (size_t)H * 64, nt);
  } else if (c < 6 + grid_ch) {
    v1 = w.color_w[0][(size_t)r * ldc + 3 + (c - 6)] * sum_chunks<16>(p_c0 + ((size_t)b * nt * H + r) * 64 + (c - 6), (size_t)H * 64, nt);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { v0 += __shfl_xor(v0, o, 64); v1 += __shfl_xor(v1, o, 64); }
  if (c == 0) {
    g.d_freq_geo[((size_t)b * n_geo) * H + r] = 15.f * __builtin_fmaf(bias[r], g.d_phase_geo[((size_t)b * n_geo) * H + r], v0);
    g.d_freq_app[((size_t)b * n_color) * H + r] += 15.f * v1;
  }
}

// FiLM sums: film_partial [L][B][nchunk][H][2] -> d_phase / d_freq [B][n*H] (geo | app split), d_bias[l][H] via pointers
__global__ void film_reduce_kernel(const float* part, int B, int L, int H, int n_geo, int nchunk0, int nchunk, int stride, const float* fp,
                                   const float* inv, float* d_freq_geo, float* d_phase_geo, float* d_freq_app, float* d_phase_app, FenerfSirenGrads g) {
  const float TWO_PI = 6.28318530717958647692f;
  const int n_color = L - n_geo;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L * H; i += gridDim.x * blockDim.x) {
    const int l = i / H, n = i % H;
    float db = 0.f;
    for (int b = 0; b < B; ++b) {
      const int nk = l == 0 ? nchunk0 : nchunk;
      const float2* p = reinterpret_cast<const float2*>(part) + (((size_t)l * B + b) * stride) * H + n;   // + k * H
      float2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;           // four independent chains: the loads are latency-bound
      int k = 0;
      for (; k + 4 <= nk; k += 4) {
        const float2 v0 = p[(size_t)k * H], v1 = p[(size_t)(k + 1) * H], v2 = p[(size_t)(k + 2) * H], v3 = p[(size_t)(k + 3) * H];
        a0.x += v0.x; a0.y += v0.y; a1.x += v1.x; a1.y += v1.y; a2.x += v2.x; a2.y += v2.y; a3.x += v3.x; a3.y += v3.y;
      }
      for (; k < nk; ++k) { const float2 v = p[(size_t)k * H]; a0.x += v.x; a0.y += v.y; }
      const float s0 = (a0.x + a1.x) + (a2.x + a3.x), s1 = (a0.y + a1.y) + (a2.y + a3.y);
      db += s0 * (fp[((size_t)b * L + l) * H + n] * TWO_PI / (inv ? inv[(size_t)l * H + n] : 1.f));
      if (l < n_geo) { d_phase_geo[((size_t)b * n_geo + l) * H + n] = s0; d_freq_geo[((size_t)b * n_geo + l) * H + n] = 15.f * s1; }
      else { d_phase_app[((size_t)b * n_color + (l - n_geo)) * H + n] = s0; d_freq_app[((size_t)b * n_color + (l - n_geo)) * H + n] = 15.f * s1; }
    }
    float* dst = l < n_geo ? g.geo_b[l] : g.color_b[l - n_geo];
    if (dst) dst[n] = db;
  }
}

// The thin jobs' reductions in ONE launch (blockIdx.y = which): five small matrices and two row sums, each a latency-bound sum over
// up to 256 chunk partials.  As seven launches they cost 5 x 17 + 2 x 9 us per backward chunk (5 x 23 with 4 chains); side by side 34 us.
struct ReduceMat { float* dst; const float* src; int dst_ld, dst_col0, src_rows, src_ld, src_col0, rows, cols, layer, film; };
struct ReduceSet { ReduceMat m[5]; const float* rs_src[2]; float* rs_dst[2]; int rs_rows[2]; int n_mat; };
__global__ __launch_bounds__(256) void wgrad_reduce_thin_kernel(ReduceSet J, int B, int nchunk, const float* fp, const float* inv, int L, int H) {
  const int job = blockIdx.y;
  if (job < J.n_mat) {
    const ReduceMat q = J.m[job];
    const float TWO_PI = 6.28318530717958647692f;
    const int total = q.rows * q.cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      const int r = i / q.cols, c = i % q.cols;
      float sum = 0.f;
      for (int b = 0; b < B; ++b) {
        const float s = sum_chunks<16>(q.src + ((size_t)b * nchunk * q.src_rows + r) * q.src_ld + q.src_col0 + c, (size_t)q.src_rows * q.src_ld, nchunk);
        sum += q.film ? s * (fp[((size_t)b * L + q.layer) * H + r] * TWO_PI / (inv ? inv[(size_t)q.layer * H + r] : 1.f)) : s;
      }
      q.dst[(size_t)r * q.dst_ld + q.dst_col0 + c] = sum;
    }
  } else if (blockIdx.x == 0) {
    __shared__ float red[8][32];
    const int k2 = job - J.n_mat;
    const float* part = J.rs_src[k2];
    const int r = threadIdx.x & 31, grp = threadIdx.x >> 5;
    float s = 0.f;
    for (int k = grp; k < B * nchunk; k += 8) s += part[(size_t)k * 32 + r];
    red[grp][r] = s;
    __syncthreads();
    if ((int)threadIdx.x < J.rs_rows[k2]) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
      J.rs_dst[k2][threadIdx.x] = t;
    }
  }
}

// Per-point modulation (WgradParams::film_per_point): the gradients wrt the RAW per-point FiLM parameters are the d(theta) dump itself,
//     dL/dphase[p][l][n] = d theta_l[n][p],      dL/dfreq[p][l][n] = 15 d theta_l[n][p] (W x + b)_l[n][p] = 15 d theta (tape + bias),
// moved from the register-dump layout (lane = point, four consecutive features per float4) to the caller's [point][n*H] rows (geometry |
// colour split, 16-byte pieces).  One workgroup per 32-point tile and layer range; what torch autograd leaves in `frequencies.grad` /
// `phase_shifts.grad` of SPATIALSIRENGRID.forward_with_frequencies_phase_shifts (siren.py:464-477, FiLMLayer :119-122 unbroadcast).
__global__ __launch_bounds__(256) void pointwise_film_grads_kernel(WgradParams P, float* d_freq_geo, float* d_phase_geo, float* d_freq_app, float* d_phase_app) {
  const int H = P.H, L = P.L, ng = P.n_geo, nc = L - ng;
  const long long tile = blockIdx.x;
  const int l = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, half = lane >> 5;
  const long long tl = (long long)(H / 8) * 64;
  const float4* dt4 = reinterpret_cast<const float4*>(P.d_t) + (tile * L + l) * tl;
  const float4* tp4 = reinterpret_cast<const float4*>(P.tape) + (tile * L + l) * tl;
  const long long pt = tile * 32 + m;
  float* df = l < ng ? d_freq_geo + (pt * ng + l) * H : d_freq_app + (pt * nc + (l - ng)) * H;
  float* dp = l < ng ? d_phase_geo + (pt * ng + l) * H : d_phase_app + (pt * nc + (l - ng)) * H;
  const float* bias = P.bias + (size_t)l * H;
  for (int grp = wave; grp < H / 8; grp += 4) {
    const int row = tape_feature(grp, half, 0);
    const float4 d = nt_load(dt4 + grp * 64 + lane), t = nt_load(tp4 + grp * 64 + lane);
    const float4 b = *reinterpret_cast<const float4*>(bias + row);
    *reinterpret_cast<float4*>(dp + row) = d;
    *reinterpret_cast<float4*>(df + row) = make_float4(15.f * (d.x * (t.x + b.x)), 15.f * (d.y * (t.y + b.y)), 15.f * (d.z * (t.z + b.z)), 15.f * (d.w * (t.w + b.w)));
  }
}

// film_per_point: FiLM-layer bias gradients = the chunk partials of the scaled d(theta) row sums, summed in a fixed order
__global__ __launch_bounds__(256) void pointwise_bias_reduce_kernel(FenerfSirenGrads g, const float* part, int B, int L, int H, int n_geo, int nchunk0, int nchunk, int stride) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L * H; i += gridDim.x * blockDim.x) {
    const int l = i / H, n = i % H;
    const int nk = l == 0 ? nchunk0 : nchunk;
    float db = 0.f;
    for (int b = 0; b < B; ++b) db += sum_chunks<4>(part + (((size_t)l * B + b) * stride) * H + n, (size_t)H, nk);
    float* dst = l < n_geo ? g.geo_b[l] : g.color_b[l - n_geo];
    if (dst) dst[n] = db;
  }
}

namespace {
int hipfail(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

template <int H, int JOB>
size_t wg_lds_bytes() {
  using S = WgShape<H, JOB>;
  return (size_t)((S::A_ROWS + S::B_ROWS) * WG_LD + 2 * H) * sizeof(float);
}

template <int H, int JOB, bool PW = false>
int launch_job(const WgradParams& p, int nz, hipStream_t st) {
  auto kfn = siren_wgrad_kernel<H, JOB, PW>;
  const size_t lds = wg_lds_bytes<H, JOB>();
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  hipLaunchKernelGGL(kfn, dim3(p.nchunk, p.B, nz), dim3(256), lds, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hipfail(e, "wgrad launch");
}

template <int H, bool T16>
int launch_sq_bf16(const WgradParams& p, int nz, hipStream_t st) {
  constexpr int NW = SqWaves<H>::value;
  auto kfn = siren_wgrad_sq_bf16_kernel<H, NW, T16>;
  const size_t lds = (size_t)(4 * H * WG_LD + 2 * H) * sizeof(float);     // two [A | B] images + FiLM rows
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  hipLaunchKernelGGL(kfn, dim3(p.nchunk, p.B, nz), dim3(NW * 64), lds, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hipfail(e, "wgrad bf16 launch");
}

template <int H>
int launch_sq_b16d(const WgradParams& p, int nz, hipStream_t st) {
  auto kfn = siren_wgrad_sq_b16d_kernel<H>;
  const size_t lds = (size_t)4 * H * WD_LD * sizeof(unsigned short);     // two [A | B] images
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  hipLaunchKernelGGL(kfn, dim3(p.nchunk, p.B, nz), dim3(256), lds, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hipfail(e, "wgrad bf16-dump launch");
}

}  // namespace

int wgrad_nchunk(const FenerfModel* m, int B, long long tiles_per_image) {
  // one workgroup per CU (110 KB LDS, the whole register file): size the grid of the (L-1)*B square jobs to whole rounds of the
  // machine -- 39 chunks gave 780 workgroups = 3.05 rounds on 256 CUs, i.e. a fourth round 5 % full.
  const long long jobs = (long long)(m->L - 1) * B;
  long long best = 1;
  double best_eff = 0.0;
  for (long long n = 1; n <= 64 && n <= tiles_per_image; ++n) {
    const long long wgs = jobs * n;
    const long long rounds = (wgs + launch_cus(m) - 1) / launch_cus(m);
    const long long t_per = (tiles_per_image + n - 1) / n;
    // time ~ rounds * tiles per chunk (+ a fixed per-workgroup cost of ~4 tiles: prologue, partial store, reduce traffic)
    const double eff = (double)tiles_per_image / (double)(rounds * (t_per + 4)) / (double)launch_cus(m) * (double)jobs;
    if (eff > best_eff) { best_eff = eff; best = n; }
  }
  return (int)best;
}

// thin jobs (layer 0, colour-layer-0 extras, heads) stream the tapes with almost no MFMA work
int wgrad_nchunk_thin(const FenerfModel* m, int B, long long tiles_per_image) {
  // two workgroups per CU (41-74 KB of LDS, <= 164 registers each): single-buffered, they need the second one to keep
  // loads in flight while the first stages and multiplies (measured 578 -> 460 us for the four jobs; three: 443 us, but the
  // partial reductions grow with the chunk count)
#ifndef FENERF_THIN_WGS
#define FENERF_THIN_WGS 2
#endif
  long long n = ((long long)FENERF_THIN_WGS * launch_cus(m) + B - 1) / B;
  if (n > tiles_per_image) n = tiles_per_image;
#ifndef FENERF_THIN_CAP
#define FENERF_THIN_CAP 256
#endif
  if (n > FENERF_THIN_CAP) n = FENERF_THIN_CAP;
  return (int)(n < 1 ? 1 : n);
}

// chunks of tiles per (layer, image) of the FiLM-sum gather
static int film_nchunk(long long tiles_per_image) { return (int)(tiles_per_image < 64 ? tiles_per_image : 64); }

size_t wgrad_workspace_bytes(const FenerfModel* m, int B, long long P) {
  const long long tiles = (P + 31) / 32;
  const int nc = wgrad_nchunk(m, B, tiles), nt = wgrad_nchunk_thin(m, B, tiles), ncm = film_nchunk(tiles) > nt ? film_nchunk(tiles) : nt;
  const size_t H = m->H;
  size_t sq = (size_t)(m->L - 1) * B * nc * H * H;         // square partials
  const size_t thin = (size_t)B * nt * H * 160;                 // the thin jobs reuse the buffer, side by side: [B][nt][H x 32 | H x 64 | 32 x H | 32 x H]
  if (thin > sq) sq = thin;
  size_t f = sq;
  f += (size_t)m->L * B * ncm * H * 2;                      // FiLM sums
  f += (size_t)2 * B * ncm * 32;                            // head / rgb row sums
  const int ncb = nc > nt ? nc : nt;
  f += (size_t)m->L * B * ncb * H;                          // per-point modulation: bias-gradient partials (fenerf_siren_param_grads_pointwise)
  return f * sizeof(float) + 1024;
}

template <int H>
static int param_grads_t(const FenerfModel* m, WgradParams p, const FenerfSirenGrads& g, float* ws, bool film_only, hipStream_t st,
                         const FenerfSirenGrads* weights) {
  const int L = m->L, ng = m->n_geo, B = p.B;
  const int nc = p.nchunk, nt = wgrad_nchunk_thin(m, B, p.tiles_per_image), nf = film_nchunk(p.tiles_per_image), ncm = nf > nt ? nf : nt;
  const int G = m->grid_ch;
  float* sq = ws;
  size_t sq_floats = (size_t)(L - 1) * B * nc * H * H;
  const size_t thin_floats = (size_t)B * nt * H * 160;
  if (thin_floats > sq_floats) sq_floats = thin_floats;
  float* film = sq + sq_floats;
  float* rows = film + (size_t)L * B * ncm * H * 2;
  p.film_partial = film; p.rowsum_partial = rows; p.film_stride = nf;
  p.bias_partial = rows + (size_t)2 * B * ncm * 32; p.bias_stride = nc > nt ? nc : nt;
  int rc;
  if (p.film_per_point) {   // the per-point FiLM gradients are the dump itself, re-laid; the bias gradients come from the jobs below
    PhaseScope ph(PH_WGRAD_FILM, st);
    hipLaunchKernelGGL(pointwise_film_grads_kernel, dim3((unsigned)((long long)B * p.tiles_per_image), L), dim3(256), 0, st, p, g.d_freq_geo, g.d_phase_geo,
                       g.d_freq_app, g.d_phase_app);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "per-point film gradients launch");
  } else
  {  // FiLM frequency / phase gradients and the FiLM-layer biases: gather the chain kernel's per-tile sums, reduce
    PhaseScope ph(PH_WGRAD_FILM, st);
    WgradParams pf = p;
    pf.nchunk = nf;
    hipLaunchKernelGGL(film_gather_kernel, dim3(nf, B, L), dim3(256), 0, st, pf);
    hipLaunchKernelGGL(film_reduce_kernel, dim3((L * H + 255) / 256), dim3(256), 0, st, film, B, L, H, ng, nf, nf, nf, p.fp, p.inv, g.d_freq_geo,
                       g.d_phase_geo, g.d_freq_app, g.d_phase_app, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hipfail(e, "film sums launch");
  }
  if (film_only) return FENERF_OK;
  // ---- square products dtheta_l x_{l-1}^T, l = 1..L-1, one launch
  p.partial = sq; p.layer0 = 1;
  {
    PhaseScope ph(PH_WGRAD_SQ, st);
    if ((rc = p.bf16_dump ? launch_sq_b16d<H>(p, L - 1, st)
                          : ((m->precision == FENERF_PREC_F16X3) ? (p.tape_u16 ? launch_sq_bf16<H, true>(p, L - 1, st) : launch_sq_bf16<H, false>(p, L - 1, st))
                                                                 : (p.film_per_point ? launch_job<H, WG_SQ, true>(p, L - 1, st) : launch_job<H, WG_SQ>(p, L - 1, st))))) return rc;
  }
  {
    PhaseScope ph(PH_WGRAD_SQ_REDUCE, st);
    if (p.freq_from_sums) hipLaunchKernelGGL(wgrad_reduce_sq_kernel<true>, dim3(H, L - 1), dim3(256), 0, st, g, *weights, sq, B, nc, p.fp, p.inv, p.bias, L, H, ng, G);
    else hipLaunchKernelGGL(wgrad_reduce_sq_kernel<false>, dim3(H, L - 1), dim3(256), 0, st, g, g, sq, B, nc, p.fp, p.inv, p.bias, L, H, ng, G, p.film_per_point);
  }
  // the thin jobs reuse the square partial buffer (stream-ordered after the reduction above), with their own chunking and side by
  // side: [H x 32 | H x 64 | 32 x H | 32 x H] per (image, chunk) -- so that ONE launch reduces all of them
  p.nchunk = nt;
  float* const p_l0 = sq;
  float* const p_c0 = p_l0 + (size_t)B * nt * H * 32;
  float* const p_hd = p_c0 + (size_t)B * nt * H * 64;
  float* const p_rgb = p_hd + (size_t)B * nt * 32 * H;
  float* const rows_rgb = rows + (size_t)B * ncm * 32;
  {
    PhaseScope ph(PH_WGRAD_THIN, st);
    ThinJobs T;
    T.j[0] = p; T.j[0].layer0 = ng; T.j[0].partial = p_c0;
    T.j[1] = p; T.j[1].layer0 = ng - 1; T.j[1].partial = p_hd; T.j[1].rowsum_partial = rows;
    T.j[2] = p; T.j[2].layer0 = L - 1; T.j[2].partial = p_rgb; T.j[2].rowsum_partial = rows_rgb;
    T.j[3] = p; T.j[3].layer0 = 0; T.j[3].partial = p_l0;
    auto kfn = p.film_per_point ? siren_wgrad_thin_kernel<H, true> : siren_wgrad_thin_kernel<H, false>;
    size_t lds = wg_lds_bytes<H, WG_C0X>();
    if (wg_lds_bytes<H, WG_HEAD>() > lds) lds = wg_lds_bytes<H, WG_HEAD>();
    if (wg_lds_bytes<H, WG_L0>() > lds) lds = wg_lds_bytes<H, WG_L0>();
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds))) return rc;
    hipLaunchKernelGGL(kfn, dim3(nt, B, 4), dim3(256), lds, st, T);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return hipfail(e2, "thin wgrad launch");
  }
  ReduceSet J;
  memset(&J, 0, sizeof(J));
  int nm = 0;
  auto mat = [&](float* dst, int dst_ld, int dst_col0, const float* src, int src_rows, int src_ld, int src_col0, int rws, int cols, int layer, int film) {
    J.m[nm++] = ReduceMat{dst, src, dst_ld, dst_col0, src_rows, src_ld, src_col0, rws, cols, layer, film};
  };
  const int fsc = p.film_per_point ? 0 : 1;    // diag(f) of the image in the reduction -- not with per-point frequencies (already in the partials)
  mat(g.geo_w[0], 3, 0, p_l0, H, 32, 0, H, 3, 0, fsc);
  mat(g.color_w[0], 3 + G + H, 0, p_c0, H, 64, 32, H, 3, ng, fsc);          // view direction columns
  if (G) mat(g.color_w[0], 3 + G + H, 3, p_c0, H, 64, 0, H, G, ng, fsc);    // grid feature columns
  mat(g.head_w, H, 0, p_hd, 32, H, 0, 32, H, 0, 0);
  mat(g.rgb_w, H, 0, p_rgb, 32, H, 0, 3, H, 0, 0);
  J.n_mat = nm;
  J.rs_src[0] = rows; J.rs_dst[0] = g.head_b; J.rs_rows[0] = 32;
  J.rs_src[1] = rows_rgb; J.rs_dst[1] = g.rgb_b; J.rs_rows[1] = 3;
  PhaseScope ph(PH_WGRAD_THIN_REDUCE, st);
  hipLaunchKernelGGL(wgrad_reduce_thin_kernel, dim3((32 * H + 255) / 256, nm + 2), dim3(256), 0, st, J, B, nt, p.fp, p.inv, L, H);
  if (p.freq_from_sums)      // the frequency gradients' thin-job share (reads the same partials; the square job reuses the buffer only in the next call)
    hipLaunchKernelGGL(film_freq_thin_kernel, dim3(H, B), dim3(64), 0, st, g, *weights, p_l0, p_c0, nt, p.bias, H, ng, L - ng, G);
  if (p.film_per_point)
    hipLaunchKernelGGL(pointwise_bias_reduce_kernel, dim3((L * H + 255) / 256), dim3(256), 0, st, g, p.bias_partial, B, L, H, ng, nt, nc, p.bias_stride);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hipfail(e, "wgrad reduce launch");
}

int launch_param_grads(const FenerfModel* m, int B, long long P, const float* points, const float* dirs, const float* fp, const float* pp,
                       const float* out, const float* d_out, const float* tape, const float* tape_e, const float* d_t,
                       const FenerfSirenGrads& g, bool film_only, void* workspace, void* stream, const float* film_tiles, int tape_format,
                       const FenerfSirenGrads* weights, int film_per_point) {
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.tape = tape; p.d_t = d_t; p.film_tiles = film_tiles ? film_tiles : d_t + (size_t)m->L * m->H * (size_t)B * (size_t)P; p.tape_e = tape_e; p.points = points; p.dirs = dirs; p.out = out; p.d_out = d_out;
  p.fp = fp; p.pp = pp; p.bias = m->d_consts + CONST_FILM_BIAS;
  p.inv = m->precision == FENERF_PREC_F16X3 ? m->d_consts + CONST_FILM_BIAS + (size_t)m->L * m->H : nullptr;
  p.box_scale = m->box_scale;
  p.B = B; p.L = m->L; p.n_geo = m->n_geo; p.n_lab = m->n_lab; p.C = m->C; p.H = m->H;
  p.P = P; p.tiles_per_image = (int)(P / 32);
  p.film16w = m->precision == FENERF_PREC_F16X3 ? bwd16w_film_unit((long long)B * P, P) : 0;
  p.bf16_dump = use_bf16_dump(m, (long long)B * P);
  p.tape_u16 = tape_format == FENERF_TAPE_U16;
  p.freq_from_sums = tape_format != FENERF_TAPE_F32;
  p.film_per_point = film_per_point;
  if (film_per_point && (m->precision != FENERF_PREC_F32 || film_only)) {
    set_error("per-point FiLM parameters: FENERF_PREC_F32 models, full backward only");
    return FENERF_E_UNSUPPORTED;
  }
  if (p.freq_from_sums && (film_only || !weights || m->precision != FENERF_PREC_F16X3)) {
    set_error("FENERF_TAPE_U16 / _F32_W: needs a FENERF_PREC_F16X3 model, a full (not FiLM-only) backward and the FiLM layers' weights");
    return FENERF_E_INVALID;
  }
  p.nchunk = wgrad_nchunk(m, B, p.tiles_per_image);
  float* ws = (float*)workspace;
  switch (m->H) {
    case 32: return param_grads_t<32>(m, p, g, ws, film_only, (hipStream_t)stream, weights);
    case 64: return param_grads_t<64>(m, p, g, ws, film_only, (hipStream_t)stream, weights);
    case 96: return param_grads_t<96>(m, p, g, ws, film_only, (hipStream_t)stream, weights);
    case 128: return param_grads_t<128>(m, p, g, ws, film_only, (hipStream_t)stream, weights);
    case 192: return param_grads_t<192>(m, p, g, ws, film_only, (hipStream_t)stream, weights);
    case 256: return param_grads_t<256>(m, p, g, ws, film_only, (hipStream_t)stream, weights);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

}  // namespace fenerf
