// Backward chain of the FiLM-SIREN radiance field for gfx950 (MI355X) -- the data-path half of what torch autograd
// derives for <siren>.forward_with_frequencies_phase_shifts (reference siren/siren.py:1509-1530, FiLMLayer :227-244)
// in the generator step and in inversion (train_double_latent_semantic.py: g_loss.backward(); inverse_render_double_semantic.py).
//
// Per layer l (x_l = sin(theta_l), theta_l = f_l (W_l x_{l-1} + b_l) + p_l):
//     dL/dtheta_l = dL/dx_l * cos(theta_l)          -> written to d_t[l]   (the tape's register-dump layout)
//     dL/dz_l     = dL/dtheta_l * f_l               -> the B operand of the next GEMM
//     dL/dx_{l-1} = W_l^T dL/dz_l                   -> transposed fp32 MFMA, same register identity as the forward
// One wave carries 32 points through the whole chain; activations are not recomputed: theta_l comes from the forward's
// saved pre-FiLM accumulators (tape, register dumps of the 32-point tiles).  The kernel also leaves the per-tile FiLM sums
// (sum_p dtheta, sum_p dtheta * tape: fenerf_mfma32.h "FiLM-gradient sums").  The contractions over the point axis that remain
// (weight gradients) are fenerf_siren_wgrad.hip.  FENERF_PREC_F32 models run this kernel; FENERF_PREC_F16X3 models the
// bf16x3 variant in fenerf_siren_bwd16w.hip.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_mfma32.h"
#include "fenerf_nt.h"

namespace fenerf {


struct TapeNB { float a[16]; };
// pre-FiLM accumulators of n-block nb for this lane from the forward's register dump (fenerf_layout.h "Tape"):
// tp = tape4 + ((tile*L + layer) * (H/8)) * 64 + lane
__device__ __forceinline__ TapeNB tape_load(const float4* tp, int nb) {
  TapeNB t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = nt_load(tp + (nb * 4 + j) * 64);
    t.a[4 * j + 0] = v.x; t.a[4 * j + 1] = v.y; t.a[4 * j + 2] = v.z; t.a[4 * j + 3] = v.w;
  }
  return t;
}

// where an epilogue writes: the lane's LDS slab, the dtheta dump and the FiLM sums of (tile, layer)
struct Sink { float4* slab; float4* dtp; float* ftp; int H; LaneBits lb; };

// The epilogue of one accumulator register (element r of n-block nb: dL/dx -> dL/dtheta, dL/dz, FiLM sums), and the stores
// of a finished group of four: cut this way the epilogue of n-block nb-1 is issued, piece by piece, behind the MFMAs of
// n-block nb -- run after its own body it serialised (last MFMA done -> 16 cos + stores -> next body's loads) for ~25 % of
// the kernel.
struct BwdQuad { float d[4], o[4]; };
__device__ __forceinline__ void bwd_piece(int r, const f32x16& acc, const FilmNB& fm, const TapeNB& tn, int nb, const Sink& k, BwdQuad& q,
                                          FilmRed& R) {
  const float TWO_PI = 6.28318530717958647692f;
  const int j = r >> 2, i = r & 3;
  const float f = i == 0 ? fm.f[j].x : (i == 1 ? fm.f[j].y : (i == 2 ? fm.f[j].z : fm.f[j].w));
  const float p = i == 0 ? fm.p[j].x : (i == 1 ? fm.p[j].y : (i == 2 ? fm.p[j].z : fm.p[j].w));
  const float dt = acc[r] * cos2pi(__builtin_fmaf(f, tn.a[r], p));
  q.d[i] = dt;
  q.o[i] = dt * (f * TWO_PI);
  R.v[0][r] = dt;
  R.v[1][r] = dt * tn.a[r];
  if (i == 3) {
    nt_store(k.dtp + (nb * 4 + j) * 64, q.d[0], q.d[1], q.d[2], q.d[3]);
    k.slab[(nb * 4 + j) * 64] = make_float4(q.o[0], q.o[1], q.o[2], q.o[3]);
  }
}

// acc = dL/dx of n-block nb: the whole epilogue at once (stages that are not software-pipelined)
__device__ __forceinline__ void bwd_store(const f32x16& acc, const FilmNB& fm, const TapeNB& tn, int nb, const Sink& k) {
  BwdQuad q;
  FilmRed R;
#pragma unroll
  for (int r = 0; r < 16; ++r) bwd_piece(r, acc, fm, tn, nb, k, q, R);
#pragma unroll
  for (int c = 0; c < FILM_RED_CHUNKS; ++c) film_red_chunk(c, R, k.lb, k.ftp + 32 * nb, k.H);
}

// One of the 12 loads (8 FiLM float4, 4 tape float4) of n-block nb's epilogue operands: issued one per k-group behind the
// MFMAs of the body before, instead of all twelve in front of the body's first MFMA.
__device__ __forceinline__ void prefetch_piece(int i, FilmNB& fm, TapeNB& tn, const float* fpl, const float* ppl, const float4* tp, int nb) {
  if (i < 4) fm.f[i] = *reinterpret_cast<const float4*>(fpl + 32 * nb + 8 * i);
  else if (i < 8) fm.p[i - 4] = *reinterpret_cast<const float4*>(ppl + 32 * nb + 8 * (i - 4));
  else if (i < 12) {
    const int j = i - 8;
    const float4 v = nt_load(tp + (nb * 4 + j) * 64);
    tn.a[4 * j + 0] = v.x; tn.a[4 * j + 1] = v.y; tn.a[4 * j + 2] = v.z; tn.a[4 * j + 3] = v.w;
  }
}

// dz_l (in registers) -> dz_{l-1}: one transposed square stage.  Software pipeline over the n-blocks: behind the MFMAs of
// body nb run the epilogue of body nb-1 (k-groups 0..15) and the operand loads of body nb+1 (k-groups 16..27).
template <int H>
__device__ __forceinline__ void bwd_square(float (&in)[H / 2], Ring& ring, const float* fpl, const float* ppl, const float4* tp,
                                           const Sink& k) {
  float4* const slab = k.slab;
  constexpr int NB = H / 32, KGX = H / 8, KGXP = pad_pf(KGX);
  constexpr int NPIECE = 28 + FILM_RED_CHUNKS;           // 16 epilogue + 12 operand loads + the FiLM-sum butterfly
  constexpr int PP = (NPIECE + KGX - 1) / KGX;           // pieces per k-group (H = 256: 2, done by k-group 18 of 32)
  FilmNB fm_c = film_load(fpl, ppl, 0), fm_n = fm_c, fm_p = fm_c;
  TapeNB tn_c = tape_load(tp, 0), tn_n = tn_c, tn_p = tn_c;
  f32x16 acc_p = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // PREV / NEXT are compile-time per copy of the body (first, middle, last): no branches inside the MFMA stream
  auto body = [&](int nb, auto has_prev, auto has_next) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    BwdQuad q;
    FilmRed R;
    mfma_x_p<H / 2, KGX, KGXP>(acc, in, ring, [&](int kg) {
      if (kg < KGX) {
#pragma unroll
        for (int i = kg * PP; i < (kg + 1) * PP; ++i) {
          if (i < 16) { if (has_prev.value) bwd_piece(i, acc_p, fm_p, tn_p, nb - 1, k, q, R); }
          else if (i < 28) { if (has_next.value) prefetch_piece(i - 16, fm_n, tn_n, fpl, ppl, tp, nb + 1); }
          else if (i < NPIECE) { if (has_prev.value) film_red_chunk(i - 28, R, k.lb, k.ftp + 32 * (nb - 1), k.H); }
        }
      }
    });
    acc_p = acc; fm_p = fm_c; tn_p = tn_c; fm_c = fm_n; tn_c = tn_n;
  };
  using T = std::true_type; using F = std::false_type;
  if (NB == 1) body(0, F{}, F{});
  else {
    body(0, F{}, T{});
#pragma unroll 1
    for (int nb = 1; nb < NB - 1; ++nb) body(nb, T{}, T{});
    body(NB - 1, T{}, F{});
  }
  bwd_store(acc_p, fm_p, tn_p, NB - 1, k);
  load_act<H / 2>(in, slab);
}

template <int H, bool GRID>
__global__ __launch_bounds__(256, 1) void siren_bwd_kernel(SirenBwdParams P, int n_geo, int n_color, int n_lab, int C) {
  constexpr int NB = H / 32, KGX = H / 8, KGXP = pad_pf(KGX);
  constexpr int C0_KG = KGX + FENERF_HEAD_KSTEPS / 4, C0_KGP = pad_pf(C0_KG);
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
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += wstride) {
    long long pt = tile * 32 + m;
    const bool valid = pt < P.P;
    if (!valid) pt = P.P - 1;
    // FiLM block of this lane: its image's, or -- per-point modulation (SPATIALSIRENGRID under autograd, round 6) -- its own point's.
    // The per-tile FiLM sums below then mix points with different frequencies and are ignored: the per-point FiLM gradients are the
    // d(theta) dump itself (pointwise_film_grads_kernel, fenerf_siren_wgrad.hip)
    const long long img = P.film_per_point ? pt : pt / P.pts_per_image;
    const float* fpl = P.fp + (size_t)img * L * H + 4 * h;
    const float* ppl = P.pp + (size_t)img * L * H + 4 * h;
    const float4* tp = reinterpret_cast<const float4*>(P.tape) + tile * L * (long long)tl + lane;   // + layer * tl
    float4* dtp = reinterpret_cast<float4*>(P.d_t) + tile * L * (long long)tl + lane;

    // FiLM sums of this tile: [layer][2][H]; after the butterfly lane i holds the sum of accumulator register i & 15
    float* ftp = P.film_tiles + tile * L * 2LL * H + film_lane_feature(lane);
    auto sink = [&](int layer) { return Sink{slab, dtp + layer * tl, ftp + layer * 2 * H, H, lbits}; };

    Ring ring;
    ring.ptr = ring_base;
#pragma unroll
    for (int i = 0; i < FENERF_PF; ++i) { ring.w[i] = *ring.ptr; ring.ptr += 64; }

    // gradient wrt the head rows this lane-half multiplies: row 16h + s  (rows [0,n_lab) labels, row n_lab sigma)
    float dh[FENERF_HEAD_KSTEPS];
#pragma unroll
    for (int s = 0; s < FENERF_HEAD_KSTEPS; ++s) {
      const int row = 16 * h + s;
      const int ch = row < n_lab ? row : (row == n_lab ? C - 1 : -1);
      dh[s] = ch >= 0 ? P.d_out[pt * C + ch] : 0.f;
    }
    // ---------------- rgb head: d(pre-sigmoid) = d_rgb * s (1 - s);  dx_{L-1} = W_rgb^T d(pre) ----------------
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
        const TapeNB tn = tape_load(tp + l * tl, nb);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = MFMA(w.x, b0, acc);
        acc = MFMA(w.y, b1, acc);
        bwd_store(acc, fm, tn, nb, sink(l));
      }
    }
    float in[H / 2];
    load_act<H / 2>(in, slab);

    // ---------------- colour layers L-1 .. n_geo+1 ----------------
#pragma unroll 1
    for (int l = L - 1; l > n_geo; --l)
      bwd_square<H>(in, ring, fpl + (size_t)(l - 1) * H, ppl + (size_t)(l - 1) * H, tp + (l - 1) * tl, sink(l - 1));

    // ---------------- colour layer 0 + heads: dx_{n_geo-1} = W_c0[:, x]^T dz_{n_geo} + head^T d_head; d(grid feats) ----
    {
      const int l = n_geo - 1;
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const FilmNB fm = film_load(fpl + (size_t)l * H, ppl + (size_t)l * H, nb);
        const TapeNB tn = tape_load(tp + l * tl, nb);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int kg = 0; kg < C0_KGP; ++kg) {
          float4 w;
          RING_NEXT(ring, kg % FENERF_PF, w);
          if (kg < KGX) {
            acc = MFMA(w.x, in[4 * kg + 0], acc);
            acc = MFMA(w.y, in[4 * kg + 1], acc);
            acc = MFMA(w.z, in[4 * kg + 2], acc);
            acc = MFMA(w.w, in[4 * kg + 3], acc);
          } else if (kg < C0_KG) {
            const int q = kg - KGX;
            acc = MFMA(w.x, dh[4 * q + 0], acc);
            acc = MFMA(w.y, dh[4 * q + 1], acc);
            acc = MFMA(w.z, dh[4 * q + 2], acc);
            acc = MFMA(w.w, dh[4 * q + 3], acc);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        bwd_store(acc, fm, tn, nb, sink(l));
      }
      if (GRID) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        mfma_x<H / 2, KGX, KGXP>(acc, in, ring);
        if (valid) {
          float4* ep = reinterpret_cast<float4*>(P.d_e + pt * 32 + 4 * h);   // channels 8j + 4h + {0..3}
#pragma unroll
          for (int j = 0; j < 4; ++j) ep[2 * j] = make_float4(acc[4 * j + 0], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        }
      }
    }
    load_act<H / 2>(in, slab);

    // ---------------- geometry trunk n_geo-1 .. 1 ----------------
#pragma unroll 1
    for (int l = n_geo - 1; l >= 1; --l)
      bwd_square<H>(in, ring, fpl + (size_t)(l - 1) * H, ppl + (size_t)(l - 1) * H, tp + (l - 1) * tl, sink(l - 1));
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// gradient wrt the 3-D feature grid: the transpose of the forward's trilinear gather (sample_from_3dgrid,
// siren.py:314-330; zeros padding, align_corners=True).  One thread per (point, channel): channels are contiguous in
// the channels-last gradient grid, so each corner is one 128-B line of hardware float atomics.
// ------------------------------------------------------------------------------------------------
__global__ void grid_backward_kernel(long long P, const float* points, float box_scale, const float* d_e, float* d_grid_cl,
                                     int gd, int gh, int gw) {
  const long long total = P * 32;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pt = i >> 5;
    const int ch = (int)(i & 31);
    const float g = d_e[i];
    const float qx = points[pt * 3 + 0] * box_scale, qy = points[pt * 3 + 1] * box_scale, qz = points[pt * 3 + 2] * box_scale;
    const float ix = ((qx + 1.f) / 2.f) * (float)(gw - 1);
    const float iy = ((qy + 1.f) / 2.f) * (float)(gh - 1);
    const float iz = ((qz + 1.f) / 2.f) * (float)(gd - 1);
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int cz = c >> 2, cy = (c >> 1) & 1, cx = c & 1;
      const float xi = x0 + cx, yi = y0 + cy, zi = z0 + cz;
      const float wx = cx ? (ix - x0) : (x0 + 1.f - ix);
      const float wy = cy ? (iy - y0) : (y0 + 1.f - iy);
      const float wz = cz ? (iz - z0) : (z0 + 1.f - iz);
      const bool ok = xi >= 0.f && xi <= (float)(gw - 1) && yi >= 0.f && yi <= (float)(gh - 1) && zi >= 0.f && zi <= (float)(gd - 1);
      if (ok) {
        const long long vox = ((long long)(int)zi * gh + (int)yi) * gw + (int)xi;
        unsafeAtomicAdd(d_grid_cl + vox * 32 + ch, g * (wx * wy * wz));
      }
    }
  }
}

static int hip_fail_b(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

int launch_grid_backward(const FenerfModel* m, long long P, const float* points, const float* d_e, float* d_grid_cl, void* stream) {
  if (P <= 0) return FENERF_OK;
  long long blocks = (P * 32 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(grid_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P, points, m->box_scale, d_e,
                     d_grid_cl, m->gd, m->gh, m->gw);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail_b(e, "grid_backward launch");
}

template <int H, bool GRID>
static int launch_bwd_t(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
  const size_t lds = (size_t)4 * ((H / 8) * 64) * sizeof(float4);
  auto kfn = siren_bwd_kernel<H, GRID>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  const long long ntiles = (p.P + 31) / 32;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > launch_cus(m)) blocks = launch_cus(m);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail_b(e, "siren backward launch");
}

int launch_siren_backward(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
  if (p.P <= 0) return FENERF_OK;
  const bool g = m->grid_ch != 0;
  switch (m->H) {
    case 32: return g ? launch_bwd_t<32, true>(m, p, stream) : launch_bwd_t<32, false>(m, p, stream);
    case 64: return g ? launch_bwd_t<64, true>(m, p, stream) : launch_bwd_t<64, false>(m, p, stream);
    case 96: return g ? launch_bwd_t<96, true>(m, p, stream) : launch_bwd_t<96, false>(m, p, stream);
    case 128: return g ? launch_bwd_t<128, true>(m, p, stream) : launch_bwd_t<128, false>(m, p, stream);
    case 192: return g ? launch_bwd_t<192, true>(m, p, stream) : launch_bwd_t<192, false>(m, p, stream);
    case 256: return g ? launch_bwd_t<256, true>(m, p, stream) : launch_bwd_t<256, false>(m, p, stream);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

}  // namespace fenerf
