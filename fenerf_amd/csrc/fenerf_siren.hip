// FiLM-SIREN radiance field, fused per 32-point tile, for gfx950 (MI355X).
//
// Replaces <siren>.forward_with_frequencies_phase_shifts (reference siren/siren.py:1509-1530, :1210-1229,
// :227-244) = UniformBoxWarp + trilinear 3-D feature-grid gather + 8 FiLM layers + sigma / label heads +
// 1-3 FiLM colour layers + sigmoid rgb head: ~60 separate ATen launches per call in the reference, with
// every activation round-tripping HBM.  Here one wave carries 32 points through the WHOLE network:
//
//   * transposed GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 == fmaf chain): D[feat][pt] += W[feat][k] X^T[k][pt];
//     MFMA columns are the wave's 32 points, so a layer's output registers ARE the next layer's B operand
//     (K order pre-permuted on the host, fenerf_layout.h) -- activations never move between lanes;
//   * weights are the only HBM/L2 stream: one contiguous fp32 stream in consumption order, read with 1 KiB
//     wave-wide float4 loads through an 8-deep register prefetch ring that runs across n-block, layer and
//     stage boundaries; all 256 CUs read the same 2.9 MB, so it lives in L2;
//   * FiLM epilogue fused on the accumulators: t = f'*acc + p' (revolutions, bias folded into p'), then v_sin_f32
//     (hardware sine of revolutions, 1.2e-7 max abs error measured);
//   * the layer output is parked in the wave's private LDS slab (32 KB at H=256, lane-major float4, conflict
//     free) while the old activations are still needed as B operands, then read back into the same registers;
//   * grid features: channels-last re-laid grid, lane-half h gathers channels 16h..16h+15 of its point's 8
//     corners (4 x 16 B each) at tile start, hidden behind layer 0..7 MFMAs, consumed by colour layer 0;
//   * 1 wave / SIMD (4 waves per CU, each its own tile stream), persistent over tiles, XCD-contiguous tile
//     ranges so neighbouring rays share an L2 for grid reads.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fenerf_film.h"
#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_mfma32.h"
#include "fenerf_nt.h"

namespace fenerf {

// differentiable mode: keep the n-block's pre-FiLM accumulators as a register dump (fenerf_layout.h "Tape"):
// tp = tape4 + ((tile*L + layer) * (H/8)) * 64 + lane; one contiguous 1-KiB wave store per 4 accumulator registers.
__device__ __forceinline__ void tape_store(const f32x16& acc, int nb, float4* tp) {
#pragma unroll
  for (int j = 0; j < 4; ++j) nt_store(tp + (nb * 4 + j) * 64, acc[4 * j + 0], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
}

// A square FiLM layer H -> H.
template <int H, bool SAVE>
__device__ __forceinline__ void square_layer(float (&in)[H / 2], Ring& ring, const float* fpl, const float* ppl,
                                             float4* slab, float4* tp) {
  constexpr int NB = H / 32, KGX = H / 8, KGXP = pad_pf(KGX);
#pragma unroll 1
  for (int nb = 0; nb < NB; ++nb) {
    const FilmNB fm = film_load(fpl, ppl, nb);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    mfma_x<H / 2, KGX, KGXP>(acc, in, ring);
    if (SAVE) tape_store(acc, nb, tp);
    film_store(acc, fm, nb, slab);
  }
  load_act<H / 2>(in, slab);
}

template <int H, bool GRID, bool SAVE>
__global__ __launch_bounds__(256, 1) void siren_kernel(SirenParams P, int n_geo, int n_color, int n_lab, int C) {
  constexpr int NB = H / 32, KGX = H / 8, KGXP = pad_pf(KGX);
  constexpr int C0_KG = KGX + (GRID ? FENERF_E_KSTEPS / 4 : 0) + 1, C0_KGP = pad_pf(C0_KG);
  constexpr int SLAB_F4 = (H / 8) * 64;        // activation slab per wave, float4 units
  extern __shared__ __attribute__((aligned(16))) float4 smem[];

  if (P.clk && threadIdx.x == 0) {   // fenerf_siren_clock_probe: shader-clock and wall-clock stamps of this workgroup's first instruction
    P.clk[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
    P.clk[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int stage_f4 = (32 * C + 3) / 4;
  float4* slab = smem + wave * (SLAB_F4 + stage_f4) + lane;
  float* stage = reinterpret_cast<float*>(smem + wave * (SLAB_F4 + stage_f4) + SLAB_F4);

  const int L = n_geo + n_color;
  const float4* l0w = reinterpret_cast<const float4*>(P.stream) + lane;
  const float4* ring_base = reinterpret_cast<const float4*>(P.stream + P.ring_offset_floats) + lane;

  // XCD-contiguous tile ranges: block b runs on XCD b % 8 (observed, speed only)
  const long long ntiles = (P.P + 31) / 32;
  const int nblk = gridDim.x;
  const int nx = nblk < 8 ? nblk : 8;
  const int x = blockIdx.x % nx, bi = blockIdx.x / nx;
  const int blocks_in_x = nblk / nx + (x < nblk % nx ? 1 : 0);
  const long long t_begin = ntiles * x / nx, t_end = ntiles * (x + 1) / nx;
  const int wstride = blocks_in_x * 4;

  if (P.raw_fg && t_begin + bi * 4 < t_end) {   // FiLM pre-pass in the launch (fenerf_film.h): the images of this workgroup's tiles
    long long p_last = t_end * 32 - 1;
    if (p_last >= P.P) p_last = P.P - 1;
    film_prep_prologue(P, (t_begin + bi * 4) * 32 / P.pts_per_image, p_last / P.pts_per_image, H, n_geo, n_color);
  }

  for (long long tile = t_begin + bi * 4 + wave; tile < t_end; tile += wstride) {
    // ---------------- this lane's point ----------------
    long long pt = tile * 32 + m;
    const bool valid = pt < P.P;
    if (!valid) pt = P.P - 1;
    // FiLM block of this lane: its image's, or -- per-point modulation (SPATIALSIRENGRID, siren.py:440-477: frequencies and
    // phase shifts come from a mapping network evaluated per sample point) -- its own point's
    const long long img = P.film_per_point ? pt : pt / P.pts_per_image;
    float px, py, pz, dx, dy, dz;
    if (P.points) {
      px = P.points[pt * 3 + 0]; py = P.points[pt * 3 + 1]; pz = P.points[pt * 3 + 2];
      if (P.pdirs) { dx = P.pdirs[pt * 3 + 0]; dy = P.pdirs[pt * 3 + 1]; dz = P.pdirs[pt * 3 + 2]; }
      else { dx = 0.f; dy = 0.f; dz = -1.f; }
    } else {
      const long long ray = pt / P.n_per_ray;
      const float zz = P.z[pt];
      const float ox = P.origins[ray * 3 + 0], oy = P.origins[ray * 3 + 1], oz = P.origins[ray * 3 + 2];
      dx = P.dirs[ray * 3 + 0]; dy = P.dirs[ray * 3 + 1]; dz = P.dirs[ray * 3 + 2];
      // generators.py:504: origins + dirs * z as separate mul and add (torch does not contract to fma)
      px = __fadd_rn(ox, __fmul_rn(dx, zz)); py = __fadd_rn(oy, __fmul_rn(dy, zz)); pz = __fadd_rn(oz, __fmul_rn(dz, zz));
      if (P.lock_view) { dx = 0.f; dy = 0.f; dz = -1.f; }
    }
    // UniformBoxWarp, siren.py:181-187
    const float qx = px * P.box_scale, qy = py * P.box_scale, qz = pz * P.box_scale;

    // ---------------- prime the weight ring ----------------
    Ring ring;
    ring.ptr = ring_base;
#pragma unroll
    for (int i = 0; i < FENERF_PF; ++i) { ring.w[i] = *ring.ptr; ring.ptr += 64; }

    // ---------------- grid features (sample_from_3dgrid, siren.py:314-330; grid_sample trilinear,
    //                  zeros padding, align_corners=True).  Lane-half h blends channels 16h..16h+15. ----------
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = 0.f;
    if (GRID) {
      const float ix = ((qx + 1.f) / 2.f) * (float)(P.gw - 1);
      const float iy = ((qy + 1.f) / 2.f) * (float)(P.gh - 1);
      const float iz = ((qz + 1.f) / 2.f) * (float)(P.gd - 1);
      const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int cz = c >> 2, cy = (c >> 1) & 1, cx = c & 1;
        const float xi = x0 + cx, yi = y0 + cy, zi = z0 + cz;
        const float wx = cx ? (ix - x0) : (x0 + 1.f - ix);
        const float wy = cy ? (iy - y0) : (y0 + 1.f - iy);
        const float wz = cz ? (iz - z0) : (z0 + 1.f - iz);
        const float wgt = wx * wy * wz;
        const bool ok = xi >= 0.f && xi <= (float)(P.gw - 1) && yi >= 0.f && yi <= (float)(P.gh - 1) && zi >= 0.f &&
                        zi <= (float)(P.gd - 1);
        if (ok) {
          const long long vox = ((long long)(int)zi * P.gh + (int)yi) * P.gw + (int)xi;
          const float4* g = reinterpret_cast<const float4*>(P.grid + vox * 32 + 16 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = g[q];
            e[4 * q + 0] += v.x * wgt; e[4 * q + 1] += v.y * wgt; e[4 * q + 2] += v.z * wgt; e[4 * q + 3] += v.w * wgt;
          }
        }
      }
    }

    const float* fpl = P.fp + (size_t)img * L * H + 4 * h;   // FiLM params of this lane's image, + half offset
    const float* ppl = P.pp + (size_t)img * L * H + 4 * h;
    float4* tp = SAVE ? reinterpret_cast<float4*>(P.tape) + tile * L * (long long)(H / 8) * 64 + lane : nullptr;   // + layer * tl
    constexpr int tl = (H / 8) * 64;
    if (SAVE && GRID && valid) {
      float4* ep = reinterpret_cast<float4*>(P.tape_e + pt * 32 + 16 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) ep[q] = make_float4(e[4 * q + 0], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]);
    }

    // ---------------- layer 0: 3 -> H.  k-steps (x|y), (z|0) ----------------
    {
      const float b0 = h ? qy : qx, b1 = h ? 0.f : qz;
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const float4 w = l0w[nb * 64];
        const FilmNB fm = film_load(fpl, ppl, nb);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = MFMA(w.x, b0, acc);
        acc = MFMA(w.y, b1, acc);
        if (SAVE) tape_store(acc, nb, tp);
        film_store(acc, fm, nb, slab);
      }
    }
    float in[H / 2];
    load_act<H / 2>(in, slab);

    // ---------------- geometry trunk G1 .. G(n_geo-1) ----------------
#pragma unroll 1
    for (int l = 1; l < n_geo; ++l)
      square_layer<H, SAVE>(in, ring, fpl + (size_t)l * H, ppl + (size_t)l * H, slab, SAVE ? tp + l * tl : nullptr);

    // ---------------- colour layer 0: [x | grid feats | dir] -> H ----------------
    {
      const float* f0 = fpl + (size_t)n_geo * H;
      const float* p0 = ppl + (size_t)n_geo * H;
      const float bd0 = h ? dy : dx, bd1 = h ? 0.f : dz;
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const FilmNB fm = film_load(f0, p0, nb);
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
          } else if (GRID && kg < KGX + FENERF_E_KSTEPS / 4) {
            const int q = kg - KGX;
            acc = MFMA(w.x, e[4 * q + 0], acc);
            acc = MFMA(w.y, e[4 * q + 1], acc);
            acc = MFMA(w.z, e[4 * q + 2], acc);
            acc = MFMA(w.w, e[4 * q + 3], acc);
          } else if (kg == C0_KG - 1) {
            acc = MFMA(w.x, bd0, acc);
            acc = MFMA(w.y, bd1, acc);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (SAVE) tape_store(acc, nb, tp + n_geo * tl);
        film_store(acc, fm, nb, slab);
      }
    }
    // ---------------- head: rows [0,n_lab) folded label head, row n_lab sigma (consumes x of the trunk) -------
    {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      mfma_x<H / 2, KGX, KGXP>(acc, in, ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row <= n_lab) {
          const int ch = row < n_lab ? row : C - 1;
          stage[m * C + ch] = acc[r] + P.consts[CONST_HEAD_BIAS + row];
        }
      }
    }
    load_act<H / 2>(in, slab);

    // ---------------- colour layers 1.. ----------------
#pragma unroll 1
    for (int c = 1; c < n_color; ++c)
      square_layer<H, SAVE>(in, ring, fpl + (size_t)(n_geo + c) * H, ppl + (size_t)(n_geo + c) * H, slab,
                            SAVE ? tp + (n_geo + c) * tl : nullptr);

    // ---------------- rgb head + sigmoid ----------------
    {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      mfma_x<H / 2, KGX, KGXP>(acc, in, ring);
      if (h == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float v = acc[r] + P.consts[CONST_RGB_BIAS + r];
          stage[m * C + (C - 4) + r] = 1.f / (1.f + __expf(-v));
        }
      }
    }
    // ---------------- coalesced write-out of the tile's [32][C] block ----------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const long long base = tile * 32 * C;
      const long long limit = P.P * C;
      for (int i = lane; i < 32 * C; i += 64)
        if (base + i < limit) P.out[base + i] = stage[i];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (P.clk) {                       // ... and of its last (the waves are independent: meet first)
    __syncthreads();
    if (threadIdx.x == 0) {
      P.clk[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
      P.clk[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// FiLM pre-pass: f' = (15 f + 30) / 2pi, p' = ((15 f + 30) b + p) / 2pi  (double, rounded once).
// The '*15 + 30' itself is done in fp32 with separate mul and add exactly like siren.py:1510-1511.
// ------------------------------------------------------------------------------------------------
__global__ void film_prep_kernel(long long B, int H, int n_geo, int n_color, const float* fg, const float* pg, const float* fa,
                                 const float* pa, const float* bias /* [L][H] */, const float* inv_scale /* [L][H] or null */,
                                 float* fp, float* pp, long long dup /* > 0: every value also at index i + dup */) {
  const int L = n_geo + n_color;
  const long long total = (long long)B * L * H;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % H);
    const int l = (int)((i / H) % L);
    const long long b = i / ((long long)H * L);
    float fr, ph;
    if (l < n_geo) { fr = fg[(b * n_geo + l) * H + n]; ph = pg[(b * n_geo + l) * H + n]; }
    else { fr = fa[(b * n_color + (l - n_geo)) * H + n]; ph = pa[(b * n_color + (l - n_geo)) * H + n]; }
    const float f = __fadd_rn(__fmul_rn(fr, 15.f), 30.f);
    const double inv2pi = 0.15915494309189533576888;
    // f16x3 mode: fold the (power-of-two, exact) result scale of the layer's GEMM into the frequency
    const double sc = inv_scale ? (double)inv_scale[l * H + n] : 1.0;
    const float fv = (float)((double)f * inv2pi * sc);
    const float pv = (float)(((double)f * (double)bias[l * H + n] + (double)ph) * inv2pi);
    fp[i] = fv;
    pp[i] = pv;
    if (dup > 0) { fp[i + dup] = fv; pp[i + dup] = pv; }
  }
}

// NCDHW <-> channels-last [D][H][W][32] (one 128-B line per voxel) through an LDS tile of 64 voxels x 32 channels: both the
// planar side (64 consecutive voxels of a channel) and the channels-last side (8 KiB contiguous) are read / written coalesced.
// TO_CL: spatial_embeddings -> the gather layout (model load / re-pack); !TO_CL: the gradient grid back to the parameter's
// layout (a strided torch copy of the 113 MB gradient took 0.5 ms per step).
template <bool TO_CL>
__global__ __launch_bounds__(256) void grid_transpose_kernel(const float* src, float* dst, long long vox) {
  __shared__ float tile[64][33];
  const long long v0 = (long long)blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = k * 256 + tid;
    if (TO_CL) { const int c = idx >> 6, v = idx & 63; if (v0 + v < vox) tile[v][c] = src[(long long)c * vox + v0 + v]; }
    else { const int v = idx >> 5, c = idx & 31; if (v0 + v < vox) tile[v][c] = src[(v0 + v) * 32 + c]; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = k * 256 + tid;
    if (TO_CL) { const int v = idx >> 5, c = idx & 31; if (v0 + v < vox) dst[(v0 + v) * 32 + c] = tile[v][c]; }
    else { const int c = idx >> 6, v = idx & 63; if (v0 + v < vox) dst[(long long)c * vox + v0 + v] = tile[v][c]; }
  }
}

static int hip_fail(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

int launch_film_prep(const FenerfModel* m, long long B, const float* fg, const float* pg, const float* fa, const float* pa,
                     float* fp, float* pp, void* stream, bool for_f32_kernel, bool twice) {
  const long long total = (long long)B * m->L * m->H;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  const bool f16 = m->precision == FENERF_PREC_F16X3 && !for_f32_kernel;
  const float* cst = (for_f32_kernel && m->precision == FENERF_PREC_F16X3) ? m->d_consts32 : m->d_consts;
  hipLaunchKernelGGL(film_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B, m->H, m->n_geo, m->n_color, fg,
                     pg, fa, pa, cst + CONST_FILM_BIAS, f16 ? m->d_consts + CONST_FILM_BIAS + (size_t)m->L * m->H : nullptr, fp, pp,
                     twice ? total : 0LL);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail(e, "film_prep launch");
}

int launch_grid_relayout(const float* src, float* dst, int C, int D, int Hh, int W, void* stream) {
  const long long vox = (long long)D * Hh * W;
  if (C != 32) { set_error("feature grid must have 32 channels"); return FENERF_E_UNSUPPORTED; }
  hipLaunchKernelGGL(grid_transpose_kernel<true>, dim3((unsigned)((vox + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src, dst, vox);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail(e, "grid_relayout launch");
}

int launch_grid_unlayout(const float* src_cl, float* dst_ncdhw, int D, int Hh, int W, void* stream) {
  const long long vox = (long long)D * Hh * W;
  hipLaunchKernelGGL(grid_transpose_kernel<false>, dim3((unsigned)((vox + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src_cl, dst_ncdhw, vox);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail(e, "grid_unlayout launch");
}

template <int H, bool GRID, bool SAVE>
static int launch_siren_t(const FenerfModel* m, const SirenParams& p, void* stream) {
  const int stage_f4 = (32 * m->C + 3) / 4;
  const size_t lds = (size_t)4 * ((H / 8) * 64 + stage_f4) * sizeof(float4);
  auto kfn = siren_kernel<H, GRID, SAVE>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  const long long ntiles = (p.P + 31) / 32;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > m->num_cus) blocks = m->num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail(e, "siren launch");
}

template <int H>
static int launch_siren_h(const FenerfModel* m, const SirenParams& p, void* stream) {
  const bool g = m->grid_ch != 0;
  if (p.tape) return g ? launch_siren_t<H, true, true>(m, p, stream) : launch_siren_t<H, false, true>(m, p, stream);
  return g ? launch_siren_t<H, true, false>(m, p, stream) : launch_siren_t<H, false, false>(m, p, stream);
}

int launch_siren(const FenerfModel* m, const SirenParams& p, void* stream) {
  if (p.P <= 0) return FENERF_OK;
  if (m->precision == FENERF_PREC_F16X3) return launch_siren16w(m, p, stream);
  return launch_siren_f32(m, p, stream);
}

int launch_siren_f32(const FenerfModel* m, const SirenParams& p, void* stream) {
  if (p.P <= 0) return FENERF_OK;
  switch (m->H) {
    case 32: return launch_siren_h<32>(m, p, stream);
    case 64: return launch_siren_h<64>(m, p, stream);
    case 96: return launch_siren_h<96>(m, p, stream);
    case 128: return launch_siren_h<128>(m, p, stream);
    case 192: return launch_siren_h<192>(m, p, stream);
    case 256: return launch_siren_h<256>(m, p, stream);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

}  // namespace fenerf
