// FiLM-SIREN radiance field, f16x3 mode: error-compensated fp16 MFMA (v_mfma_f32_32x32x16_f16) for gfx950.
//
// Same dataflow as fenerf_siren.hip (one wave = 32 points through the whole network, transposed GEMM, K order
// pre-permuted on the host, activations never leave their lane, weights = one contiguous L2-resident stream through
// an 8-deep register prefetch ring) but every fp32 product w*x of the dense layers is evaluated on the fp16 matrix
// pipe (16x the fp32 MFMA rate) as
//        wh*xh + wh*xl + wl*xh ,   (wh, wl) = fp16 hi/lo split of w * 2^e_row,  (xh, xl) = split of x * 16,
// accumulated in fp32: the dropped wl*xl term is 2^-22 relative, so the result is fp32-class (measured: rgb 3e-7,
// sigma 4e-6 relative vs fp64 -- identical to the exact-fp32 kernel) at 3/16 of the fp32 MFMA time.
//   * per-row power-of-two weight scales and the activation scale are folded into the FiLM frequency (f'' = f'/(2^e*16))
//     and the head epilogues: no extra instructions, lo halves stay in fp16's normal range;
//   * layer 0 (K = 3) stays on the exact fp32 MFMA;
//   * the FiLM epilogue emits 16*sin(2 pi t) and splits it into packed (hi, lo) halves = next layer's B operands.
#include <hip/hip_runtime.h>

#include "fenerf_internal.h"
#include "fenerf_layout.h"

namespace fenerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// 16*sin(2*pi*t).  v_sin_f32 takes its argument in revolutions and does its own range reduction; measured on
// MI355X (tools/probe/probe.hip): max abs error 1.2e-7 for |t| <= 45 revolutions, i.e. better than a degree-9
// polynomial evaluated in fp32 (2.1e-7) at one (quarter-rate) instruction instead of thirteen.
__device__ __forceinline__ float sin2pi_x16(float t) { return __builtin_amdgcn_sinf(t) * F16_ACT_SCALE; }

struct Ring16 {
  float4 w[FENERF_PF];
  const float4* ptr;
};

#define RING_NEXT(ring, slot, dst)      \
  do {                                  \
    (dst) = (ring).w[(slot)];           \
    (ring).w[(slot)] = *(ring).ptr;     \
    (ring).ptr += 64;                   \
  } while (0)

__device__ __forceinline__ half8 as_half8(const float4& v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ float4 as_float4(const half8& v) { return __builtin_bit_cast(float4, v); }

// split 8 fp32 values into packed fp16 (hi, lo): hi = rn(v), lo = rn(v - hi)  (v - hi is exact in fp32)
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)(v[t] - (float)h);
  }
}

// One accumulator takes all three product chains: dependent v_mfma_f32_32x32x16_f16 on the same accumulator
// issue back-to-back at 32 cycles (measured, tools/probe), so independent chains buy nothing.
struct Acc3 { f32x16 a; };
__device__ __forceinline__ void acc3_zero(Acc3& s) {
  const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  s.a = z;
}
__device__ __forceinline__ f32x16 acc3_sum(const Acc3& s) { return s.a; }

// one k-step (16 features): consumes [hi entry, lo entry] from the ring
#define KSTEP16(ring, slot0, acc, bh, bl)                  \
  do {                                                     \
    float4 _wh, _wl;                                       \
    RING_NEXT(ring, (slot0), _wh);                         \
    RING_NEXT(ring, (slot0) + 1, _wl);                     \
    (acc).a = MFMA16(as_half8(_wl), (bh), (acc).a);        \
    (acc).a = MFMA16(as_half8(_wh), (bl), (acc).a);        \
    (acc).a = MFMA16(as_half8(_wh), (bh), (acc).a);        \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

#define RING_SKIP(ring, slot)                              \
  do {                                                     \
    float4 _d;                                             \
    RING_NEXT(ring, (slot), _d);                           \
    (void)_d;                                              \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

// acc += W_body * x over an H-wide activation (KS k-steps, EP = padded entries of the body)
template <int KS, int EP>
__device__ __forceinline__ void mfma_x16(Acc3& acc, const half8 (&xh)[KS], const half8 (&xl)[KS], Ring16& ring) {
  static_assert(EP % FENERF_PF == 0 && EP >= 2 * KS, "bodies are padded to the ring depth");
#pragma unroll
  for (int s = 0; s < KS; ++s) KSTEP16(ring, (2 * s) % FENERF_PF, acc, xh[s], xl[s]);
#pragma unroll
  for (int e = 2 * KS; e < EP; ++e) RING_SKIP(ring, e % FENERF_PF);
}

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

// FiLM epilogue of one n-block: 16*sin(2 pi (f'' acc + p')) split into (hi, lo) halves -> k-steps 2nb, 2nb+1 of the slab
__device__ __forceinline__ void film_store16(const f32x16& acc, const FilmNB& fm, int nb, float4* slab /* + lane */) {
  float v[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 f = fm.f[j], p = fm.p[j];
    v[4 * j + 0] = sin2pi_x16(__builtin_fmaf(f.x, acc[4 * j + 0], p.x));
    v[4 * j + 1] = sin2pi_x16(__builtin_fmaf(f.y, acc[4 * j + 1], p.y));
    v[4 * j + 2] = sin2pi_x16(__builtin_fmaf(f.z, acc[4 * j + 2], p.z));
    v[4 * j + 3] = sin2pi_x16(__builtin_fmaf(f.w, acc[4 * j + 3], p.w));
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float w8[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) w8[t] = v[8 * q + t];
    half8 hi, lo;
    split8(w8, hi, lo);
    slab[(2 * (2 * nb + q) + 0) * 64] = as_float4(hi);
    slab[(2 * (2 * nb + q) + 1) * 64] = as_float4(lo);
  }
}

template <int KS>
__device__ __forceinline__ void load_act16(half8 (&xh)[KS], half8 (&xl)[KS], const float4* slab) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    xh[s] = as_half8(slab[(2 * s + 0) * 64]);
    xl[s] = as_half8(slab[(2 * s + 1) * 64]);
  }
}

template <int H>
__device__ __forceinline__ void square_layer16(half8 (&xh)[H / 16], half8 (&xl)[H / 16], Ring16& ring, const float* fpl,
                                               const float* ppl, float4* slab) {
  constexpr int NB = H / 32, KS = H / 16, EP = pad_pf(2 * KS);
#pragma unroll 1
  for (int nb = 0; nb < NB; ++nb) {
    const FilmNB fm = film_load(fpl, ppl, nb);
    Acc3 acc;
    acc3_zero(acc);
    mfma_x16<KS, EP>(acc, xh, xl, ring);
    film_store16(acc3_sum(acc), fm, nb, slab);
  }
  load_act16<KS>(xh, xl, slab);
}

template <int H, bool GRID>
__global__ __launch_bounds__(256, 1) void siren16_kernel(SirenParams P, int n_geo, int n_color, int n_lab, int C) {
  constexpr int NB = H / 32, KS = H / 16, EP = pad_pf(2 * KS);
  constexpr int C0_KS = KS + (GRID ? 2 : 0) + 1, C0_EP = pad_pf(2 * C0_KS);
  constexpr int SLAB_F4 = (H / 8) * 64;        // activation slab per wave: KS k-steps x (hi, lo) x 64 lanes, float4 units
  extern __shared__ __attribute__((aligned(16))) float4 smem[];

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

  for (long long tile = t_begin + bi * 4 + wave; tile < t_end; tile += wstride) {
    // ---------------- this lane's point ----------------
    long long pt = tile * 32 + m;
    const bool valid = pt < P.P;
    if (!valid) pt = P.P - 1;
    const long long img = pt / P.pts_per_image;
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
    Ring16 ring;
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

    // ---------------- layer 0: 3 -> H.  k-steps (x|y), (z|0) ----------------
    {
      const float b0 = h ? qy : qx, b1 = h ? 0.f : qz;
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const float4 w = l0w[nb * 64];
        const FilmNB fm = film_load(fpl, ppl, nb);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = MFMA32(w.x, b0, acc);
        acc = MFMA32(w.y, b1, acc);
        film_store16(acc, fm, nb, slab);
      }
    }
    half8 xh[KS], xl[KS];
    load_act16<KS>(xh, xl, slab);

    // ---------------- geometry trunk G1 .. G(n_geo-1) ----------------
#pragma unroll 1
    for (int l = 1; l < n_geo; ++l) square_layer16<H>(xh, xl, ring, fpl + (size_t)l * H, ppl + (size_t)l * H, slab);

    // ---------------- colour layer 0: [x | grid feats | dir] -> H ----------------
    {
      const float* f0 = fpl + (size_t)n_geo * H;
      const float* p0 = ppl + (size_t)n_geo * H;
      // extra B operands: grid feats (k-step j, slot t <-> this half's channel 8j + t) and the view direction, x16
      half8 eh[2], el[2], dh, dl;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float w8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) w8[t] = e[8 * j + t] * F16_ACT_SCALE;
        split8(w8, eh[j], el[j]);
      }
      {
        float w8[8] = {dx * F16_ACT_SCALE, dy * F16_ACT_SCALE, dz * F16_ACT_SCALE, 0.f, 0.f, 0.f, 0.f, 0.f};
        split8(w8, dh, dl);
      }
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const FilmNB fm = film_load(f0, p0, nb);
        Acc3 acc;
        acc3_zero(acc);
#pragma unroll
        for (int s = 0; s < KS; ++s) KSTEP16(ring, (2 * s) % FENERF_PF, acc, xh[s], xl[s]);
        if (GRID) {
#pragma unroll
          for (int j = 0; j < 2; ++j) KSTEP16(ring, (2 * (KS + j)) % FENERF_PF, acc, eh[j], el[j]);
        }
        KSTEP16(ring, (2 * (C0_KS - 1)) % FENERF_PF, acc, dh, dl);
#pragma unroll
        for (int en = 2 * C0_KS; en < C0_EP; ++en) RING_SKIP(ring, en % FENERF_PF);
        film_store16(acc3_sum(acc), fm, nb, slab);
      }
    }
    // ---------------- head: rows [0,n_lab) folded label head, row n_lab sigma (consumes x of the trunk) -------
    {
      Acc3 acc3;
      acc3_zero(acc3);
      mfma_x16<KS, EP>(acc3, xh, xl, ring);
      const f32x16 acc = acc3_sum(acc3);
      const float* head_inv = P.consts + CONST_FILM_BIAS + (size_t)2 * L * H;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row <= n_lab) {
          const int ch = row < n_lab ? row : C - 1;
          stage[m * C + ch] = acc[r] * head_inv[row] + P.consts[CONST_HEAD_BIAS + row];
        }
      }
    }
    load_act16<KS>(xh, xl, slab);

    // ---------------- colour layers 1.. ----------------
#pragma unroll 1
    for (int c = 1; c < n_color; ++c)
      square_layer16<H>(xh, xl, ring, fpl + (size_t)(n_geo + c) * H, ppl + (size_t)(n_geo + c) * H, slab);

    // ---------------- rgb head + sigmoid ----------------
    {
      Acc3 acc3;
      acc3_zero(acc3);
      mfma_x16<KS, EP>(acc3, xh, xl, ring);
      const f32x16 acc = acc3_sum(acc3);
      const float* rgb_inv = P.consts + CONST_FILM_BIAS + (size_t)2 * L * H + 32;
      if (h == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float v = acc[r] * rgb_inv[r] + P.consts[CONST_RGB_BIAS + r];
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
}


static int hip_fail16(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

template <int H, bool GRID>
static int launch_siren16_t(const FenerfModel* m, const SirenParams& p, void* stream) {
  const int stage_f4 = (32 * m->C + 3) / 4;
  const size_t lds = (size_t)4 * ((H / 8) * 64 + stage_f4) * sizeof(float4);
  static size_t configured = 0;
  auto kfn = siren16_kernel<H, GRID>;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_fail16(e, "hipFuncSetAttribute(max dynamic LDS)");
    configured = lds;
  }
  const long long ntiles = (p.P + 31) / 32;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > m->num_cus) blocks = m->num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail16(e, "siren16 launch");
}

int launch_siren16(const FenerfModel* m, const SirenParams& p, void* stream) {
  if (p.P <= 0) return FENERF_OK;
  const bool g = m->grid_ch != 0;
  switch (m->H) {
    case 32: return g ? launch_siren16_t<32, true>(m, p, stream) : launch_siren16_t<32, false>(m, p, stream);
    case 64: return g ? launch_siren16_t<64, true>(m, p, stream) : launch_siren16_t<64, false>(m, p, stream);
    case 128: return g ? launch_siren16_t<128, true>(m, p, stream) : launch_siren16_t<128, false>(m, p, stream);
    case 256: return g ? launch_siren16_t<256, true>(m, p, stream) : launch_siren16_t<256, false>(m, p, stream);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

}  // namespace fenerf
