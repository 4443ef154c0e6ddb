// SPATIALSIRENGRID (reference siren/siren.py:413-518), the whole per-point evaluation in ONE kernel for gfx950 (MI355X):
//
//     local latent (32 floats per sample point, sampled from the 2-D latent grid by the caller)
//       -> per-point mapping network 32 -> 256 -> 256 -> 2 L H   (siren.py:440: CustomMappingNetwork(32, 256, ., n_blocks=1))
//       -> FiLM-SIREN with THAT POINT's frequencies / phase shifts (siren.py:464-477)  -> [rgb | sigma]
//
// The reference materialises frequencies and phase shifts for every point ([B, P, 2 L H]: 18 KB per point at H = 256) between a
// torch Sequential and ~30 ATen ops; round 2's fenerf_siren_forward_pointwise still read them from HBM (7 GB per 128^2 x 24 pass).
// Here they exist only as MFMA accumulators: one wave carries 32 points through mapping network and SIREN with the transposed GEMM
// of fenerf_siren.hip (v_mfma_f32_32x32x2_f32 = exact fp32, D[feature][point] += W[feature][k] X^T[k][point], a layer's output
// registers ARE the next layer's B operand).  For n-block nb of FiLM layer l three accumulators are built from one weight stream
//     F = W2[freq rows of (l, nb)] h2      P = W2[phase rows of (l, nb)] h2      Z = W_l[rows of nb] x_{l-1}
// and meet in the epilogue  x_l = sin(2 pi (f' Z + p')),  f' = (15 (F + b2f) + 30) / 2 pi,  p' = ((15 (F + b2f) + 30) b_l + P + b2p) / 2 pi
// (the '* 15 + 30' in fp32 with separate mul and add like siren.py:465; the rest in fp64 rounded once, exactly film_prep_kernel).
// h2 (256 values per point) stays in registers for the whole tile, x_{l-1} too; outputs are parked in the wave's LDS slab.
// HBM traffic per point: 12 B position + 12 B direction + 128 B latent in, 16 B out; the 2.9 MB weight stream lives in L2.
// Work per point: 2.5 MFLOP mapping network + 1.1 MFLOP SIREN on the exact fp32 matrix pipe (157 TFLOP/s peak).
#include <hip/hip_runtime.h>

#include <new>
#include <string>
#include <vector>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_mfma32.h"

struct FenerfLocalModel {
  int H, n_geo, n_color, L;
  float box_scale;
  float* d_stream;     // packed ring stream (fenerf_pack.cpp pack_local_weights)
  float* d_consts;     // b0 [MH] | b1 [MH] | b2 [2 L H] | FiLM biases [L H] | sigma bias, pad [4] | rgb bias [4]
  int num_cus;
};

namespace fenerf {

constexpr int LOCAL_MH = 256;     // mapping network width (siren.py:440)
constexpr int LOCAL_ZL = 32;      // local latent channels

struct LocalParams {
  const float* stream; const float* consts;
  const float* points; const float* dirs; const float* latents;
  float* out;
  long long P;
  float box_scale;
  int n_geo, n_color;
};

// bias rows of one n-block for this lane-half: features 32 nb + 8 j + 4 h + {0..3}
struct Bias4 { float4 v[4]; };
__device__ __forceinline__ Bias4 bias_load(const float* b /* + 4 h */, int nb) {
  Bias4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.v[j] = *reinterpret_cast<const float4*>(b + 32 * nb + 8 * j);
  return r;
}
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }
__device__ __forceinline__ void lrelu_store(const f32x16& acc, const Bias4& b, int nb, float4* slab) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 o;
    o.x = lrelu(acc[4 * j + 0] + b.v[j].x); o.y = lrelu(acc[4 * j + 1] + b.v[j].y);
    o.z = lrelu(acc[4 * j + 2] + b.v[j].z); o.w = lrelu(acc[4 * j + 3] + b.v[j].w);
    slab[(nb * 4 + j) * 64] = o;
  }
}
// FiLM epilogue of one n-block from the three accumulators (header)
__device__ __forceinline__ float film1(float F, float Pp, float Z, float bf, float bp, float bl) {
  const float f = __fadd_rn(__fmul_rn(F + bf, 15.f), 30.f);
  const double inv2pi = 0.15915494309189533576888;
  const float fp = (float)((double)f * inv2pi);
  const float pp = (float)(((double)f * (double)bl + (double)(Pp + bp)) * inv2pi);
  return sin2pi(__builtin_fmaf(fp, Z, pp));
}
__device__ __forceinline__ void film3_store(const f32x16& aF, const f32x16& aP, const f32x16& aZ, const Bias4& bf, const Bias4& bp,
                                            const Bias4& bl, int nb, float4* slab) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 o;
    o.x = film1(aF[4 * j + 0], aP[4 * j + 0], aZ[4 * j + 0], bf.v[j].x, bp.v[j].x, bl.v[j].x);
    o.y = film1(aF[4 * j + 1], aP[4 * j + 1], aZ[4 * j + 1], bf.v[j].y, bp.v[j].y, bl.v[j].y);
    o.z = film1(aF[4 * j + 2], aP[4 * j + 2], aZ[4 * j + 2], bf.v[j].z, bp.v[j].z, bl.v[j].z);
    o.w = film1(aF[4 * j + 3], aP[4 * j + 3], aZ[4 * j + 3], bf.v[j].w, bp.v[j].w, bl.v[j].w);
    slab[(nb * 4 + j) * 64] = o;
  }
}

template <int H>
__global__ __launch_bounds__(256, 1) void siren_local_kernel(LocalParams P) {
  constexpr int MH = LOCAL_MH, NBM = MH / 32, KGM = MH / 8;              // mapping width: n-blocks, k-groups (already a multiple of PF)
  constexpr int NB = H / 32, KGX = H / 8, KGXP = pad_pf(KGX);
  constexpr int KGC = KGX + 1, KGCP = pad_pf(KGC);                       // first colour layer: x | dir
  constexpr int SLAB_F4 = (MH / 8) * 64;                                 // per-wave slab, float4 units (MH >= H)
  constexpr int C = 4;
  static_assert(H <= MH && KGM % FENERF_PF == 0, "slab and ring assumptions");
  extern __shared__ __attribute__((aligned(16))) float4 smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  constexpr int stage_f4 = (32 * C + 3) / 4;
  float4* slab = smem + wave * (SLAB_F4 + stage_f4) + lane;
  float* stage = reinterpret_cast<float*>(smem + wave * (SLAB_F4 + stage_f4) + SLAB_F4);

  const int n_geo = P.n_geo, n_color = P.n_color, L = n_geo + n_color;
  const float4* ring_base = reinterpret_cast<const float4*>(P.stream) + lane;
  const float* b0 = P.consts + 4 * h;
  const float* b1 = b0 + MH;
  const float* b2f = b1 + MH;                       // + l * H
  const float* b2p = b2f + (size_t)L * H;
  const float* bl = b2p + (size_t)L * H;            // FiLM-layer biases
  const float* bh = P.consts + 2 * MH + (size_t)3 * L * H;    // sigma bias [4] | rgb bias [4]

  const long long ntiles = (P.P + 31) / 32;
  const int nblk = gridDim.x;
  const int nx = nblk < 8 ? nblk : 8;
  const int x = blockIdx.x % nx, bi = blockIdx.x / nx;
  const int blocks_in_x = nblk / nx + (x < nblk % nx ? 1 : 0);
  const long long t_begin = ntiles * x / nx, t_end = ntiles * (x + 1) / nx;
  const int wstride = blocks_in_x * 4;

  for (long long tile = t_begin + bi * 4 + wave; tile < t_end; tile += wstride) {
    long long pt = tile * 32 + m;
    if (pt >= P.P) pt = P.P - 1;
    const float qx = P.points[pt * 3 + 0] * P.box_scale, qy = P.points[pt * 3 + 1] * P.box_scale, qz = P.points[pt * 3 + 2] * P.box_scale;
    float dx = 0.f, dy = 0.f, dz = -1.f;
    if (P.dirs) { dx = P.dirs[pt * 3 + 0]; dy = P.dirs[pt * 3 + 1]; dz = P.dirs[pt * 3 + 2]; }
    // this lane-half's latent channels: k-step s multiplies (channel 2 s | channel 2 s + 1)
    float lv[LOCAL_ZL / 2];
    {
      const float4* lp = reinterpret_cast<const float4*>(P.latents + pt * LOCAL_ZL);
#pragma unroll
      for (int q = 0; q < LOCAL_ZL / 4; ++q) {
        const float4 v = lp[q];
        lv[2 * q + 0] = h ? v.y : v.x;
        lv[2 * q + 1] = h ? v.w : v.z;
      }
    }
    Ring ring;
    ring.ptr = ring_base;
#pragma unroll
    for (int i = 0; i < FENERF_PF; ++i) { ring.w[i] = *ring.ptr; ring.ptr += 64; }

    // ---------------- mapping network, layer 0: 32 -> MH, LeakyReLU(0.2) ----------------
#pragma unroll 1
    for (int nb = 0; nb < NBM; ++nb) {
      const Bias4 bb = bias_load(b0, nb);
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kg = 0; kg < FENERF_PF; ++kg) {       // 16 k-steps = 4 entries, body padded to the ring depth
        float4 w;
        RING_NEXT(ring, kg, w);
        if (kg < LOCAL_ZL / 8) {
          acc = MFMA(w.x, lv[4 * kg + 0], acc);
          acc = MFMA(w.y, lv[4 * kg + 1], acc);
          acc = MFMA(w.z, lv[4 * kg + 2], acc);
          acc = MFMA(w.w, lv[4 * kg + 3], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      lrelu_store(acc, bb, nb, slab);
    }
    float hid[MH / 2];
    load_act<MH / 2>(hid, slab);
    // ---------------- mapping network, layer 1: MH -> MH, LeakyReLU(0.2) ----------------
#pragma unroll 1
    for (int nb = 0; nb < NBM; ++nb) {
      const Bias4 bb = bias_load(b1, nb);
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      mfma_x<MH / 2, KGM, KGM>(acc, hid, ring);
      lrelu_store(acc, bb, nb, slab);
    }
    load_act<MH / 2>(hid, slab);       // h2: the B operand of every frequency / phase-shift product of this tile

    // ---------------- FiLM layer 0: 3 -> H ----------------
    {
      const float c0 = h ? qy : qx, c1 = h ? 0.f : qz;
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const Bias4 vf = bias_load(b2f, nb), vp = bias_load(b2p, nb), vl = bias_load(bl, nb);
        f32x16 aF = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, aP = aF, aZ = aF;
        mfma_x<MH / 2, KGM, KGM>(aF, hid, ring);
        mfma_x<MH / 2, KGM, KGM>(aP, hid, ring);
#pragma unroll
        for (int kg = 0; kg < FENERF_PF; ++kg) {     // one real entry: k-steps (x | y), (z | 0)
          float4 w;
          RING_NEXT(ring, kg, w);
          if (kg == 0) {
            aZ = MFMA(w.x, c0, aZ);
            aZ = MFMA(w.y, c1, aZ);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        film3_store(aF, aP, aZ, vf, vp, vl, nb, slab);
      }
    }
    float in[H / 2];
    load_act<H / 2>(in, slab);
    // ---------------- FiLM layers 1 .. n_geo - 1 ----------------
#pragma unroll 1
    for (int l = 1; l < n_geo; ++l) {
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const Bias4 vf = bias_load(b2f + (size_t)l * H, nb), vp = bias_load(b2p + (size_t)l * H, nb), vl = bias_load(bl + (size_t)l * H, nb);
        f32x16 aF = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, aP = aF, aZ = aF;
        mfma_x<MH / 2, KGM, KGM>(aF, hid, ring);
        mfma_x<MH / 2, KGM, KGM>(aP, hid, ring);
        mfma_x<H / 2, KGX, KGXP>(aZ, in, ring);
        film3_store(aF, aP, aZ, vf, vp, vl, nb, slab);
      }
      load_act<H / 2>(in, slab);
    }
    // ---------------- sigma head on the trunk output (row 0) ----------------
    {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      mfma_x<H / 2, KGX, KGXP>(acc, in, ring);
      if (h == 0) stage[m * C + 3] = acc[0] + bh[0];
    }
    // ---------------- colour layers: the first takes [x | dir], FiLM parameters of layer n_geo + c ----------------
#pragma unroll 1
    for (int c = 0; c < n_color; ++c) {
      const int l = n_geo + c;
      const float d0 = h ? dy : dx, d1 = h ? 0.f : dz;
#pragma unroll 1
      for (int nb = 0; nb < NB; ++nb) {
        const Bias4 vf = bias_load(b2f + (size_t)l * H, nb), vp = bias_load(b2p + (size_t)l * H, nb), vl = bias_load(bl + (size_t)l * H, nb);
        f32x16 aF = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, aP = aF, aZ = aF;
        mfma_x<MH / 2, KGM, KGM>(aF, hid, ring);
        mfma_x<MH / 2, KGM, KGM>(aP, hid, ring);
        if (c == 0) {
#pragma unroll
          for (int kg = 0; kg < KGCP; ++kg) {
            float4 w;
            RING_NEXT(ring, kg % FENERF_PF, w);
            if (kg < KGX) {
              aZ = MFMA(w.x, in[4 * kg + 0], aZ);
              aZ = MFMA(w.y, in[4 * kg + 1], aZ);
              aZ = MFMA(w.z, in[4 * kg + 2], aZ);
              aZ = MFMA(w.w, in[4 * kg + 3], aZ);
            } else if (kg == KGX) {
              aZ = MFMA(w.x, d0, aZ);
              aZ = MFMA(w.y, d1, aZ);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          mfma_x<H / 2, KGX, KGXP>(aZ, in, ring);
        }
        film3_store(aF, aP, aZ, vf, vp, vl, nb, slab);
      }
      load_act<H / 2>(in, slab);
    }
    // ---------------- rgb head + sigmoid ----------------
    {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      mfma_x<H / 2, KGX, KGXP>(acc, in, ring);
      if (h == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) stage[m * C + r] = 1.f / (1.f + __expf(-(acc[r] + bh[4 + r])));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const long long base = tile * 32 * C, limit = P.P * C;
      for (int i = lane; i < 32 * C; i += 64)
        if (base + i < limit) P.out[base + i] = stage[i];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int H>
static int launch_local_t(const FenerfLocalModel* m, const LocalParams& p, void* stream) {
  const size_t lds = (size_t)4 * ((LOCAL_MH / 8) * 64 + (32 * 4 + 3) / 4) * sizeof(float4);
  auto kfn = siren_local_kernel<H>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  const long long ntiles = (p.P + 31) / 32;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > m->num_cus) blocks = m->num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("siren_local launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

}  // namespace fenerf

using namespace fenerf;

static int local_fail(int code, const std::string& msg) { set_error(msg); return code; }

extern "C" int fenerf_local_model_create(const FenerfModelDesc* d, const FenerfLocalMapDesc* mp, FenerfLocalModel** out) {
  if (!out) return local_fail(FENERF_E_INVALID, "out is NULL");
  *out = nullptr;
  std::vector<float> blob, consts;
  std::string err;
  int rc = pack_local_weights(d, mp, blob, consts, err);
  if (rc) return local_fail(rc, err);
  if ((rc = check_trig_domain())) return rc;
  FenerfLocalModel* m = new (std::nothrow) FenerfLocalModel();
  if (!m) return local_fail(FENERF_E_NOMEM, "out of host memory");
  m->H = d->hidden_dim; m->n_geo = d->n_geo; m->n_color = d->n_color; m->L = d->n_geo + d->n_color;
  m->box_scale = d->box_scale; m->d_stream = nullptr; m->d_consts = nullptr;
  int dev = 0;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
  if (e == hipSuccess) { m->num_cus = prop.multiProcessorCount; e = hipMalloc((void**)&m->d_stream, blob.size() * sizeof(float)); }
  if (e == hipSuccess) e = hipMalloc((void**)&m->d_consts, consts.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(m->d_stream, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_consts, consts.data(), consts.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    fenerf_local_model_destroy(m);
    return local_fail(FENERF_E_HIP, std::string("fenerf_local_model_create: ") + hipGetErrorString(e));
  }
  *out = m;
  return FENERF_OK;
}

extern "C" void fenerf_local_model_destroy(FenerfLocalModel* m) {
  if (!m) return;
  if (m->d_stream) (void)hipFree(m->d_stream);
  if (m->d_consts) (void)hipFree(m->d_consts);
  delete m;
}

extern "C" int fenerf_siren_forward_local(const FenerfLocalModel* m, int64_t total_points, const float* points, const float* ray_dirs,
                                          const float* latents, float* out, void* stream) {
  if (!m) return local_fail(FENERF_E_INVALID, "model is NULL");
  if (total_points < 0) return local_fail(FENERF_E_INVALID, "total_points < 0");
  if (total_points == 0) return FENERF_OK;
  if (!points || !latents || !out) return local_fail(FENERF_E_INVALID, "points / latents / out is NULL");
  LocalParams p;
  p.stream = m->d_stream; p.consts = m->d_consts;
  p.points = points; p.dirs = ray_dirs; p.latents = latents; p.out = out;
  p.P = total_points; p.box_scale = m->box_scale; p.n_geo = m->n_geo; p.n_color = m->n_color;
  PhaseScope ph(PH_SIREN, stream);
  switch (m->H) {
    case 32: return launch_local_t<32>(m, p, stream);
    case 64: return launch_local_t<64>(m, p, stream);
    case 96: return launch_local_t<96>(m, p, stream);
    case 128: return launch_local_t<128>(m, p, stream);
    case 192: return launch_local_t<192>(m, p, stream);
    case 256: return launch_local_t<256>(m, p, stream);
  }
  return local_fail(FENERF_E_UNSUPPORTED, "unsupported hidden_dim");
}
