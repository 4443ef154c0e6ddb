// One ray of fancy_integration (+ the sorted merge of fine / coarse samples, + the inverse-CDF resampling of the coarse pass) on one
// 64-lane wavefront, lane = sample -- the body shared by composite_kernel (fenerf_composite.hip: one wave per ray over the whole chip) and
// by the fused render launch (fenerf_siren_f16w.hip: the waves of a workgroup composite the rays of the ray group whose SIREN tiles they
// have just evaluated).
//
// reference: generators/volumetric_rendering.py:18-106 (fancy_integration), :259-300 (sample_pdf),
//            generators/generators.py:486-519 (resample orchestration, cat/sort/gather merge).
#pragma once
#include <hip/hip_runtime.h>

#include "fenerf_internal.h"

namespace fenerf {

// MAXM (template parameter: 128 in the fused render launch, 128 ... 1024 in the stand-alone kernels): samples per ray handled by one wave;
// SLOTS = MAXM / 64 samples per lane: slot s of lane l is sample 64 s + l (slots past M are skipped, and a skipped slot contributes exact
// zeros: the result does not depend on MAXM).  The launchers pick the smallest that fits (LDS and registers).

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive scans across the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v *= t;
  }
  return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

__device__ __forceinline__ float softplus_f(float x) {  // F.softplus(beta=1, threshold=20)
  return x > 20.f ? x : log1pf(expf(x));
}

// s_z [MAXM] (z by source index / cdf), s_zs [MAXM + 1] (sorted z), s_ord [MAXM] (sorted position -> source index), s_w [MAXM] (weights by
// sorted position): this wave's LDS scratch.  P.M <= MAXM.
template <bool MERGE, int MAXM>
__device__ __forceinline__ void composite_ray(const CompositeParams& P, const long long ray, const int lane, float* s_z, float* s_zs,
                                              int* s_ord, float* s_w) {
  constexpr int SLOTS = MAXM / 64;
  const int M = P.M, C = P.C, N = P.N;
  float zk[SLOTS], sg[SLOTS];
  int src[SLOTS];
  // ---- sorted order
  if (MERGE) {
    // source index i < N: fine sample i, else coarse sample i-N   (cat([fine, coarse]), generators.py:508-509)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int i = lane + 64 * s;
      if (i < M) s_z[i] = i < N ? P.z_a[ray * N + i] : P.z_b[ray * N + (i - N)];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int i = lane + 64 * s;
      if (i < M) {
        const float zi = s_z[i];
        int rank = 0;
        for (int j = 0; j < M; ++j) {
          const float zj = s_z[j];
          rank += (zj < zi || (zj == zi && j < i)) ? 1 : 0;   // stable ascending (torch.sort, generators.py:510)
        }
        s_ord[rank] = i;
        s_zs[rank] = zi;
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int i = lane + 64 * s;
      if (i < M) { s_ord[i] = i; s_zs[i] = P.z_a[ray * M + i]; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- per-sample alpha (volumetric_rendering.py:23-34)
  float alpha[SLOTS], tt[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int k = lane + 64 * s;
    alpha[s] = 0.f; tt[s] = 1.f; zk[s] = 0.f; src[s] = 0; sg[s] = 0.f;
    if (k < M) {
      src[s] = s_ord[k];
      zk[s] = s_zs[k];
      const float* row = MERGE ? (src[s] < N ? P.rows_a + (ray * N + src[s]) * (long long)C
                                             : P.rows_b + (ray * N + (src[s] - N)) * (long long)C)
                               : P.rows_a + (ray * M + k) * (long long)C;
      sg[s] = row[C - 1];
      const float delta = (k == M - 1) ? 1e10f : (s_zs[k + 1] - zk[s]);
      float x = sg[s];
      if (P.noise) x = __fadd_rn(x, __fmul_rn(P.noise[ray * M + k], P.o.noise_std));
      const float act = P.o.clamp_mode == FENERF_CLAMP_SOFTPLUS ? softplus_f(x) : fmaxf(x, 0.f);
      // M == 1: the reference builds delta_inf from deltas[:, :, :1] of an EMPTY deltas tensor (:23-25), so every
      // per-sample tensor is empty and rgb / depth / weights_sum come out 0 -- reproduce that.
      alpha[s] = M > 1 ? 1.f - expf(-delta * act) : 0.f;
      tt[s] = 1.f - alpha[s] + 1e-10f;
    }
  }
  // ---- exclusive transmittance: T_k = prod_{j<k} (1 - alpha_j + 1e-10)   (cumprod, :36-37)
  // two-level: wavefront product scan inside a 64-sample slot, running product of the slots before it
  float w[SLOTS];
  float carry = 1.f, wacc = 0.f;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    w[s] = 0.f;
    if (64 * s < M) {                                  // wave-uniform
      const float inc = wave_scan_mul(tt[s], lane);
      float ex = __shfl_up(inc, 1, 64);
      if (lane == 0) ex = 1.f;
      w[s] = alpha[s] * (s == 0 ? ex : carry * ex);
      const float tot = __shfl(inc, 63, 64);
      carry = s == 0 ? tot : carry * tot;
    }
    wacc = s == 0 ? w[0] : wacc + w[s];
  }
  const float wsum = wave_sum(wacc);
  if (P.o.last_back) {   // weights[:, :, -1] += (1 - weights_sum)   (:40-41)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
      if (lane + 64 * s == M - 1) w[s] += 1.f - wsum;
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int k = lane + 64 * s;
    if (k < M) {
      s_w[k] = w[s];
      if (P.out_weights) P.out_weights[ray * M + k] = w[s];
      if (P.out_z) P.out_z[ray * M + k] = zk[s];
    }
  }
  if (P.out_wsum && lane == 0) P.out_wsum[ray] = wsum;
  if (!MERGE && P.z_fine) {
    // ---- fused importance resampling of the coarse pass (generators.py:486-499 + sample_pdf, volumetric_rendering.py:259-300):
    //      the arithmetic of resample_kernel<false, .> below on the weights in s_w and the depths in s_zs; s_z holds the cdf
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int K = M - 2;
    float ww[SLOTS];
    float wtot = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int j = lane + 64 * s;   // pdf bin j uses coarse weight j + 1
      ww[s] = j < K ? __fadd_rn(__fadd_rn(s_w[j + 1], 1e-5f), 1e-5f) : 0.f;
      wtot = s == 0 ? ww[0] : wtot + ww[s];
    }
    const float tot = wave_sum(wtot);
    if (lane == 0) s_z[0] = 0.f;
    float base = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (64 * s < K) {                                                                // wave-uniform
        const float inc = wave_scan_add(ww[s] / tot, lane);
        if (lane + 64 * s < K) s_z[lane + 64 * s + 1] = s == 0 ? inc : base + inc;
        const float tots = __shfl(inc, 63, 64);
        base = s == 0 ? tots : base + tots;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int i = lane + 64 * s;
      if (i < M) {
        const float ui = P.u[ray * M + i];
        int inds = 0;
        for (int j = 0; j <= K; ++j) inds += s_z[j] < ui ? 1 : 0;
        const int below = inds - 1 > 0 ? inds - 1 : 0;
        const int above = inds < K ? inds : K;
        const float c0 = s_z[below], c1 = s_z[above];
        const float b0 = 0.5f * (s_zs[below] + s_zs[below + 1]), b1 = 0.5f * (s_zs[above] + s_zs[above + 1]);   // z_vals_mid
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        P.z_fine[ray * M + i] = b0 + (ui - c0) / denom * (b1 - b0);
      }
    }
  }
  float dacc = w[0] * zk[0];
#pragma unroll
  for (int s = 1; s < SLOTS; ++s) dacc += w[s] * zk[s];
  const float depth = wave_sum(dacc);   // (:44)
  if (P.out_depth && lane == 0) P.out_depth[ray] = depth;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- colour / label accumulation: lane = (channel c = lane & 31 [+32 in a 2nd pass], sample parity)  (:43)
  if (!P.sigma_only && P.out_rgb) {
    const int nch = C - 1;
    const bool pad = P.o.fill_mode == FENERF_FILL_SEG_PADDING_BACKGROUND || P.o.fill_mode == FENERF_FILL_EVAL_SEG_PADDING_BACKGROUND;
    const bool low = wsum < 0.9f;
    const bool fill = low && ((pad && P.o.fill_enabled) || P.o.fill_mode == FENERF_FILL_EVAL_WHITE_BACK);
    float* orow = P.out_rgb + ray * (long long)P.out_ch;
    for (int c0 = 0; c0 < nch; c0 += 32) {
      const int c = c0 + (lane & 31), par = lane >> 5;
      float acc = 0.f;
      if (c < nch) {
        // four loads in flight per lane (the loop is a chain of dependent LDS read -> address -> global load otherwise); the
        // products are still added in sample order, so the result does not change
        int k = par;
        for (; k + 6 < M; k += 8) {
          float v[4], wk[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int kk = k + 2 * u;
            const int sidx = s_ord[kk];
            const float* row = MERGE ? (sidx < N ? P.rows_a + (ray * N + sidx) * (long long)C
                                                 : P.rows_b + (ray * N + (sidx - N)) * (long long)C)
                                     : P.rows_a + (ray * M + kk) * (long long)C;
            v[u] = row[c];
            wk[u] = s_w[kk];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += wk[u] * v[u];
        }
        for (; k < M; k += 2) {
          const int sidx = s_ord[k];
          const float* row = MERGE ? (sidx < N ? P.rows_a + (ray * N + sidx) * (long long)C
                                               : P.rows_b + (ray * N + (sidx - N)) * (long long)C)
                                   : P.rows_a + (ray * M + k) * (long long)C;
          acc += s_w[k] * row[c];
        }
      }
      acc += __shfl_xor(acc, 32, 64);
      if (P.o.white_back) acc = acc + 1.f - wsum;          // (:46-47)
      if (P.o.black_back) acc = acc + (1.f - wsum) * -1.f; // (:49-50)
      if (par == 0 && c < nch) {
        if (pad) orow[c + 1] = fill ? P.o.fill_value : acc;                       // (:71-83 / :85-97)
        else if (P.o.fill_mode == FENERF_FILL_EVAL_WHITE_BACK) orow[c] = fill ? 1.f : acc;  // (:99-102)
        else orow[c] = acc;
      }
    }
    if (pad && lane == 0) orow[0] = fill ? 1.f : 0.f;   // background channel prepended, 1 on filled rays
  }
}

}  // namespace fenerf
