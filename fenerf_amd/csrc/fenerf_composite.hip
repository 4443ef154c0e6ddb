// Ray-tail kernels for gfx950: alpha compositing (fancy_integration), inverse-CDF resampling (sample_pdf)
// and the sorted merge of coarse+fine samples -- one 64-lane wavefront per ray, lane = sample.
//
// reference: generators/volumetric_rendering.py:18-106 (fancy_integration), :259-300 (sample_pdf),
//            generators/generators.py:486-519 (resample orchestration, cat/sort/gather merge).
// The reference runs these as ~15-25 tiny ATen launches over [B,R,M,*] tensors and materialises the
// sorted [B,R,2N,22] tensor; here the exclusive transmittance product is a wavefront shuffle scan, the
// merge is a rank computation in LDS (stable, like torch.sort on CPU) and the sorted tensor never exists.
#include <hip/hip_runtime.h>

#include "fenerf_composite_ray.h"
#include "fenerf_internal.h"

namespace fenerf {

// ------------------------------------------------------------------------------------------------
// composite (+ optional merge of fine/coarse): one wave per ray
// ------------------------------------------------------------------------------------------------
// waves (= rays in flight) per workgroup: four, two for rays of more than 512 samples (their four LDS arrays are 16 KiB per wave)
template <int MAXM> struct RayWaves { static constexpr int n = MAXM > 512 ? 2 : 4; };

template <bool MERGE, int MAXM>
__global__ __launch_bounds__(64 * RayWaves<MAXM>::n) void composite_kernel(CompositeParams P) {
  constexpr int WPB = RayWaves<MAXM>::n;
  __shared__ float s_z[WPB][MAXM];      // z by source index (merge) / sorted z
  __shared__ float s_zs[WPB][MAXM + 1]; // sorted z
  __shared__ int s_ord[WPB][MAXM];      // sorted position -> source index
  __shared__ float s_w[WPB][MAXM];      // weights by sorted position

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long nwaves = (long long)gridDim.x * WPB;
  for (long long ray = (long long)blockIdx.x * WPB + wv; ray < P.BR; ray += nwaves) {
    composite_ray<MERGE, MAXM>(P, ray, lane, s_z[wv], s_zs[wv], s_ord[wv], s_w[wv]);
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// composite backward: d(loss)/d(rgb_final) -> d(loss)/d(rows) for the final fancy_integration of a render
// (what torch autograd derives for volumetric_rendering.py:23-50).  Same wave-per-ray layout; the suffix sum
// S_k = sum_{j>k} dw_j w_j is a reverse wavefront scan.  fill modes are not differentiated (generator.forward, the only
// differentiated caller, does not use them: generators.py:519); depth is not differentiated.
//   w_k = a_k T_k, T_k = prod_{j<k} u_j, u_j = 1 - a_j + 1e-10, a_k = 1 - exp(-delta_k act(sigma_k + noise))
//   dL/da_k = T_k dL/dw_k - S_k / u_k ;  dL/dsigma_k = dL/da_k * delta_k (1 - a_k) act'(.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_suffix_incl(float v, int lane) {   // inclusive suffix sum across lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(v, o, 64);
    if (lane + o < 64) v += t;
  }
  return v;
}

template <bool MERGE, int MAXM>
__global__ __launch_bounds__(64 * RayWaves<MAXM>::n) void composite_backward_kernel(CompositeParams P) {
  constexpr int SLOTS = MAXM / 64, WPB = RayWaves<MAXM>::n;
  __shared__ float s_z[WPB][MAXM];
  __shared__ float s_zs[WPB][MAXM + 1];
  __shared__ int s_ord[WPB][MAXM];
  // gradient rows of one ray, by SOURCE index, for a coalesced write-out (round 3: every lane writing its own 88-byte row channel by
  // channel cost 5x the bytes at the memory side -- PMC WRITE_SIZE 369 MB for 69 MB of gradients); rays of up to CB_STAGE_M samples
  constexpr int CB_STAGE_M = 64, CB_STAGE_C = 24;
  __shared__ float s_out[WPB][MAXM <= 256 ? CB_STAGE_M * CB_STAGE_C : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int M = P.M, C = P.C, N = P.N, nch = C - 1;
  const bool staged = MAXM <= 256 && M <= CB_STAGE_M && C <= CB_STAGE_C;
  const long long nwaves = (long long)gridDim.x * WPB;
  for (long long ray = (long long)blockIdx.x * WPB + wv; ray < P.BR; ray += nwaves) {
    // ---- sorted order (identical to the forward kernel)
    if (MERGE) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        if (i < M) s_z[wv][i] = i < N ? P.z_a[ray * N + i] : P.z_b[ray * N + (i - N)];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        if (i < M) {
          const float zi = s_z[wv][i];
          int rank = 0;
          for (int j = 0; j < M; ++j) {
            const float zj = s_z[wv][j];
            rank += (zj < zi || (zj == zi && j < i)) ? 1 : 0;
          }
          s_ord[wv][rank] = i;
          s_zs[wv][rank] = zi;
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        if (i < M) { s_ord[wv][i] = i; s_zs[wv][i] = P.z_a[ray * M + i]; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- forward quantities per sample + dL/dw'_k = sum_c g_c row_k[c]
    float alpha[SLOTS], tt[SLOTS], dact[SLOTS], delta[SLOTS], gw[SLOTS];
    const float* row[SLOTS];
    float* drow[SLOTS];
    const float* g = P.g_rgb + ray * (long long)nch;
    float gsum = 0.f;
    for (int c = lane; c < nch; c += 64) gsum += g[c];
    gsum = wave_sum(gsum);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int k = lane + 64 * s;
      alpha[s] = 0.f; tt[s] = 1.f; dact[s] = 0.f; delta[s] = 0.f; gw[s] = 0.f; row[s] = nullptr; drow[s] = nullptr;
      if (k < M) {
        const int src = s_ord[wv][k];
        if (MERGE) {
          const long long off = src < N ? (ray * N + src) * (long long)C : (ray * N + (src - N)) * (long long)C;
          row[s] = (src < N ? P.rows_a : P.rows_b) + off;
          drow[s] = (src < N ? P.d_rows_a : P.d_rows_b) + off;
        } else {
          row[s] = P.rows_a + (ray * M + k) * (long long)C;
          drow[s] = P.d_rows_a + (ray * M + k) * (long long)C;
        }
        delta[s] = (k == M - 1) ? 1e10f : (s_zs[wv][k + 1] - s_zs[wv][k]);
        float x = row[s][C - 1];
        if (P.noise) x = __fadd_rn(x, __fmul_rn(P.noise[ray * M + k], P.o.noise_std));
        float act;
        if (P.o.clamp_mode == FENERF_CLAMP_SOFTPLUS) { act = softplus_f(x); dact[s] = x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
        else { act = fmaxf(x, 0.f); dact[s] = x > 0.f ? 1.f : 0.f; }
        alpha[s] = M > 1 ? 1.f - expf(-delta[s] * act) : 0.f;
        tt[s] = 1.f - alpha[s] + 1e-10f;
        float acc = 0.f;
        for (int c = 0; c < nch; ++c) acc += g[c] * row[s][c];
        gw[s] = acc;
      }
    }
    // exclusive transmittance, weights (as forward)
    float T[SLOTS], w[SLOTS], wp[SLOTS];   // wp: weights actually used in the colour sum (after last_back)
    float carry = 1.f, wacc = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      T[s] = 1.f; w[s] = 0.f;
      if (64 * s < M) {
        const float inc = wave_scan_mul(tt[s], lane);
        float ex = __shfl_up(inc, 1, 64);
        if (lane == 0) ex = 1.f;
        T[s] = s == 0 ? ex : carry * ex;
        w[s] = alpha[s] * T[s];
        const float tot = __shfl(inc, 63, 64);
        carry = s == 0 ? tot : carry * tot;
      }
      wp[s] = w[s];
      wacc = s == 0 ? w[0] : wacc + w[s];
    }
    const float wsum = wave_sum(wacc);
    if (P.o.last_back) {
      // w'_last = w_last + 1 - sum_j w_j  ->  dL/dw_j = dL/dw'_j - dL/dw'_last
      const int ls = (M - 1) >> 6, ll = (M - 1) & 63;
      float gl = gw[0];
#pragma unroll
      for (int s = 1; s < SLOTS; ++s) gl = ls == s ? gw[s] : gl;
      const float g_last = __shfl(gl, ll, 64);
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (lane + 64 * s == M - 1) wp[s] += 1.f - wsum;
        gw[s] -= g_last;
      }
    }
    // rgb += (1 - wsum) for white_back, -= for black_back (wsum taken BEFORE the last_back adjustment,
    // volumetric_rendering.py:40-48): d/dw_j = -+ sum_c g_c
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (P.o.white_back) gw[s] -= gsum;
      if (P.o.black_back) gw[s] += gsum;
    }
    // S_k = sum_{j>k} gw_j w_j
    float S[SLOTS];
    float tail = 0.f;                          // sum over all later slots
#pragma unroll
    for (int s = SLOTS - 1; s >= 0; --s) {
      const float q = (lane + 64 * s < M) ? gw[s] * w[s] : 0.f;
      const float suf = wave_suffix_incl(q, lane);
      S[s] = (s == SLOTS - 1 ? suf : suf + tail) - q;
      const float tot = __shfl(suf, 0, 64);
      tail = s == SLOTS - 1 ? tot : tail + tot;
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int k = lane + 64 * s;
      if (k < M) {
        const float dalpha = T[s] * gw[s] - S[s] / tt[s];
        const float dsigma = (M > 1) ? dalpha * delta[s] * (1.f - alpha[s]) * dact[s] : 0.f;
        if (staged) {
          float* o = s_out[wv] + s_ord[wv][k] * C;
          for (int c = 0; c < nch; ++c) o[c] = wp[s] * g[c];
          o[C - 1] = dsigma;
        } else {
          for (int c = 0; c < nch; ++c) drow[s][c] = wp[s] * g[c];
          drow[s][C - 1] = dsigma;
        }
      }
    }
    if (staged) {
      // source index i < N: row i of d_rows_a (fine / the only input), else row i - N of d_rows_b (coarse): two contiguous runs
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int na = (MERGE ? N : M) * C;
      float* oa = P.d_rows_a + ray * (long long)na;
      for (int i = lane; i < na; i += 64) oa[i] = s_out[wv][i];
      if (MERGE) {
        float* ob = P.d_rows_b + ray * (long long)na;
        for (int i = lane; i < na; i += 64) ob[i] = s_out[wv][na + i];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// one launch of a wave-per-ray kernel instantiated for MAXM samples per ray
template <int MAXM, typename K, typename... A>
static void launch_rays(K kernel, long long rays, void* stream, A... args) {
  constexpr int WPB = RayWaves<MAXM>::n;
  long long blocks = (rays + WPB - 1) / WPB;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * WPB), 0, (hipStream_t)stream, args...);
}

// the smallest MAXM that holds the ray (a skipped slot contributes exact zeros: the result does not depend on it; fewer registers + LDS)
#define FENERF_BY_RAY_SAMPLES(M, KERNEL, FLAG, rays, stream, ...)                                    \
  do {                                                                                               \
    if ((M) <= 128) launch_rays<128>(KERNEL<FLAG, 128>, rays, stream, __VA_ARGS__);                  \
    else if ((M) <= 256) launch_rays<256>(KERNEL<FLAG, 256>, rays, stream, __VA_ARGS__);             \
    else if ((M) <= 512) launch_rays<512>(KERNEL<FLAG, 512>, rays, stream, __VA_ARGS__);             \
    else launch_rays<1024>(KERNEL<FLAG, 1024>, rays, stream, __VA_ARGS__);                           \
  } while (0)

int launch_composite_backward(const CompositeParams& p, bool merge, void* stream) {
  if (p.BR <= 0) return FENERF_OK;
  if (merge) FENERF_BY_RAY_SAMPLES(p.M, composite_backward_kernel, true, p.BR, stream, p);
  else FENERF_BY_RAY_SAMPLES(p.M, composite_backward_kernel, false, p.BR, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("composite_backward launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

// ------------------------------------------------------------------------------------------------
// importance resampling: generators.py:486-499 + sample_pdf (volumetric_rendering.py:259-300)
// ------------------------------------------------------------------------------------------------
// RAW = false: zc [BR,N], wc [BR,N] (coarse z / weights), K = N-2, draws N samples      (generators.py:486-499)
// RAW = true : zc = bins [BR,K+1], wc = weights [BR,K] as the caller passes them to sample_pdf, draws NS samples
// RS = 64-sample slots per lane: K + 1 <= 64 RS knots, NS <= 64 RS draws (2: the reference's curricula; 4: up to 256 samples; 8: up to 512)
template <bool RAW, int RS>
__global__ __launch_bounds__(256) void resample_kernel(long long BR, int K, int NS, const float* __restrict__ zc,
                                                       const float* __restrict__ wc, const float* __restrict__ u,
                                                       float* __restrict__ zf) {
  __shared__ float s_cdf[4][64 * RS];
  __shared__ float s_bin[4][64 * RS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int N = K + 2;          // RAW=false: coarse samples per ray; pdf bins = weights[:, 1:-1]; cdf has K+1 knots
  const long long nwaves = (long long)gridDim.x * 4;
  for (long long ray = (long long)blockIdx.x * 4 + wv; ray < BR; ray += nwaves) {
    float ww[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int j = lane + 64 * s;   // pdf bin j uses coarse weight j+1
      ww[s] = 0.f;
      if (RAW) {
        if (j < K) ww[s] = __fadd_rn(wc[ray * K + j], 1e-5f);                          // + eps (:273)
        if (j < K + 1) s_bin[wv][j] = zc[ray * (K + 1) + j];
      } else {
        if (j < K) ww[s] = __fadd_rn(__fadd_rn(wc[ray * N + j + 1], 1e-5f), 1e-5f);    // +1e-5 (generators.py:489) + eps (:273)
        if (j < K + 1) s_bin[wv][j] = 0.5f * (zc[ray * N + j] + zc[ray * N + j + 1]);  // z_vals_mid (generators.py:494)
      }
    }
    float wsum = ww[0];
#pragma unroll
    for (int s = 1; s < RS; ++s) wsum += ww[s];
    const float tot = wave_sum(wsum);
    if (lane == 0) s_cdf[wv][0] = 0.f;                                                 // (:276)
    float base = 0.f;
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      if (64 * s < K) {                                                                // wave-uniform
        const float inc = wave_scan_add(ww[s] / tot, lane);                            // pdf (:274), cdf = cumsum (:275)
        if (lane + 64 * s < K) s_cdf[wv][lane + 64 * s + 1] = s == 0 ? inc : base + inc;
        const float tots = __shfl(inc, 63, 64);
        base = s == 0 ? tots : base + tots;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int i = lane + 64 * s;
      if (i < NS) {
        const float ui = u[ray * NS + i];
        int inds = 0;                                   // torch.searchsorted(cdf, u) (left): #knots < u  (:286)
        for (int j = 0; j <= K; ++j) inds += s_cdf[wv][j] < ui ? 1 : 0;
        const int below = inds - 1 > 0 ? inds - 1 : 0;  // (:287-288)
        const int above = inds < K ? inds : K;
        const float c0 = s_cdf[wv][below], c1 = s_cdf[wv][above];
        const float b0 = s_bin[wv][below], b1 = s_bin[wv][above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;                 // (:295-296)
        zf[ray * NS + i] = b0 + (ui - c0) / denom * (b1 - b0);   // (:299)
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// ray setup: get_initial_rays_trig + perturb_points + camera pose -> cam2world -> world-space rays
// (volumetric_rendering.py:109-168, :220-248) in one launch instead of ~40 tiny ATen ops.
// One thread per ray; the per-image camera basis is recomputed per thread (a few dozen flops).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float torch_linspace(float start, float end, int steps, int i) {
  // torch.linspace kernel: step = (end-start)/(steps-1); i < steps/2 ? start + step*i : end - step*(steps-1-i)
  if (steps == 1) return start;
  const float step = __fdiv_rn(__fsub_rn(end, start), (float)(steps - 1));
  return i < steps / 2 ? __fadd_rn(start, __fmul_rn(step, (float)i)) : __fsub_rn(end, __fmul_rn(step, (float)(steps - 1 - i)));
}
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
  const float n = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
  x = __fdiv_rn(x, n); y = __fdiv_rn(y, n); z = __fdiv_rn(z, n);
}

__global__ __launch_bounds__(256) void ray_setup_kernel(int B, int S, int N, float z_cam, float ray_start, float ray_end,
                                                        const float* __restrict__ u_jitter, const float* __restrict__ theta_in,
                                                        const float* __restrict__ phi_in, float* __restrict__ origins,
                                                        float* __restrict__ dirs, float* __restrict__ z_out,
                                                        float* __restrict__ pitch, float* __restrict__ yaw) {
  const long long R = (long long)S * S;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)B * R) return;
  const int b = (int)(idx / R);
  const int r = (int)(idx % R);
  const int row = r / S, col = r % S;
  // camera-space direction: x = linspace(-1,1,W)[col], y = linspace(1,-1,H)[row], z = -1/tan(fov/2)   (:113-121)
  float dx = torch_linspace(-1.f, 1.f, S, col), dy = torch_linspace(1.f, -1.f, S, row), dz = z_cam;
  normalize3(dx, dy, dz);
  // camera origin on the unit sphere (:220-228)
  const float theta = theta_in[b];
  float phi = phi_in[b];
  phi = fminf(fmaxf(phi, 1e-5f), 3.14159265358979323846f - 1e-5f);
  const float sp = sinf(phi), cp = cosf(phi), st = sinf(theta), ct = cosf(theta);
  const float ox = __fmul_rn(sp, ct), oy = cp, oz = __fmul_rn(sp, st);
  // look-at basis (:230-248): forward = normalize(normalize(-o)), left = normalize(up x f), up' = normalize(f x left)
  float fx = -ox, fy = -oy, fz = -oz;
  normalize3(fx, fy, fz);
  normalize3(fx, fy, fz);
  float lx = __fsub_rn(__fmul_rn(1.f, fz), __fmul_rn(0.f, fy));   // cross((0,1,0), f)
  float ly = __fsub_rn(__fmul_rn(0.f, fx), __fmul_rn(0.f, fz));
  float lz = __fsub_rn(__fmul_rn(0.f, fy), __fmul_rn(1.f, fx));
  normalize3(lx, ly, lz);
  float ux = __fsub_rn(__fmul_rn(fy, lz), __fmul_rn(fz, ly));     // cross(f, left)
  float uy = __fsub_rn(__fmul_rn(fz, lx), __fmul_rn(fx, lz));
  float uz = __fsub_rn(__fmul_rn(fx, ly), __fmul_rn(fy, lx));
  normalize3(ux, uy, uz);
  // R = [-left | up | -f] (columns); world dir = R @ d_cam  (:162)
  const float wx = __fadd_rn(__fadd_rn(__fmul_rn(-lx, dx), __fmul_rn(ux, dy)), __fmul_rn(-fx, dz));
  const float wy = __fadd_rn(__fadd_rn(__fmul_rn(-ly, dx), __fmul_rn(uy, dy)), __fmul_rn(-fy, dz));
  const float wz = __fadd_rn(__fadd_rn(__fmul_rn(-lz, dx), __fmul_rn(uz, dy)), __fmul_rn(-fz, dz));
  dirs[idx * 3 + 0] = wx; dirs[idx * 3 + 1] = wy; dirs[idx * 3 + 2] = wz;
  origins[idx * 3 + 0] = ox; origins[idx * 3 + 1] = oy; origins[idx * 3 + 2] = oz;
  // stratified jitter (:133-139): z_k = linspace(start,end,N)[k] + (u - 0.5) * (z_1 - z_0)
  const float step = N > 1 ? __fsub_rn(torch_linspace(ray_start, ray_end, N, 1), torch_linspace(ray_start, ray_end, N, 0)) : 0.f;
  for (int k = 0; k < N; ++k) {
    const float u = u_jitter[idx * N + k];
    z_out[idx * N + k] = __fadd_rn(torch_linspace(ray_start, ray_end, N, k), __fmul_rn(__fsub_rn(u, 0.5f), step));
  }
  if (r == 0) { pitch[b] = phi; yaw[b] = theta; }
}

int launch_ray_setup(int B, int S, int N, float z_cam, float ray_start, float ray_end, const float* u_jitter,
                     const float* theta, const float* phi, float* origins, float* dirs, float* z, float* pitch, float* yaw,
                     void* stream) {
  const long long total = (long long)B * S * S;
  if (total <= 0) return FENERF_OK;
  hipLaunchKernelGGL(ray_setup_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, S, N, z_cam,
                     ray_start, ray_end, u_jitter, theta, phi, origins, dirs, z, pitch, yaw);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("ray_setup launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

static int hip_fail2(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

int launch_composite(const CompositeParams& p, bool merge, void* stream) {
  if (p.BR <= 0) return FENERF_OK;
  if (merge) FENERF_BY_RAY_SAMPLES(p.M, composite_kernel, true, p.BR, stream, p);   // same results for any MAXM: fenerf_composite_ray.h
  else FENERF_BY_RAY_SAMPLES(p.M, composite_kernel, false, p.BR, stream, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail2(e, "composite launch");
}

int launch_resample(long long BR, int N, const float* z, const float* w, const float* u, float* zf, void* stream) {
  if (BR <= 0) return FENERF_OK;
  long long blocks = (BR + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (N > 256) hipLaunchKernelGGL((resample_kernel<false, 8>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, BR, N - 2, N, z, w, u, zf);
  else if (N > 128) hipLaunchKernelGGL((resample_kernel<false, 4>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, BR, N - 2, N, z, w, u, zf);
  else hipLaunchKernelGGL((resample_kernel<false, 2>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, BR, N - 2, N, z, w, u, zf);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail2(e, "resample launch");
}

int launch_sample_pdf(long long BR, int K, int NS, const float* bins, const float* w, const float* u, float* out, void* stream) {
  if (BR <= 0) return FENERF_OK;
  long long blocks = (BR + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (K > 255 || NS > 256) hipLaunchKernelGGL((resample_kernel<true, 8>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, BR, K, NS, bins, w, u, out);
  else if (K > 127 || NS > 128) hipLaunchKernelGGL((resample_kernel<true, 4>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, BR, K, NS, bins, w, u, out);
  else hipLaunchKernelGGL((resample_kernel<true, 2>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, BR, K, NS, bins, w, u, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail2(e, "sample_pdf launch");
}

}  // namespace fenerf
