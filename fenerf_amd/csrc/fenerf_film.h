// FiLM pre-pass as a kernel prologue (fenerf_render_forward): f' = (15 f + 30) / 2 pi [x the GEMM's result scale at f16x3],
// p' = ((15 f + 30) b + p) / 2 pi for images [img0, img1] -- the arithmetic of film_prep_kernel (fenerf_siren.hip), value for value.
// Every workgroup of the SIREN launch prepares the blocks of the images its own tiles belong to and then reads them back itself
// (LDS-DMA / plain loads): its own stores are visible to it after the fence + barrier; other workgroups that share an image write
// the same bytes.  The blocks stay in the workspace for the fine pass.
#pragma once
#include <hip/hip_runtime.h>

#include "fenerf_internal.h"

namespace fenerf {

__device__ __forceinline__ void film_prep_prologue(const SirenParams& P, long long img0, long long img1, int H, int n_geo, int n_color) {
  const int L = n_geo + n_color;
  float* fp = const_cast<float*>(P.fp);
  float* pp = const_cast<float*>(P.pp);
  if (img1 >= P.n_images) img1 = P.n_images - 1;
  const long long first = img0 * L * H, total = (img1 - img0 + 1) * (long long)L * H;
  for (long long k = threadIdx.x; k < total; k += blockDim.x) {
    const long long i = first + k;
    const int n = (int)(i % H);
    const int l = (int)((i / H) % L);
    const long long b = i / ((long long)H * L);
    float fr, ph;
    if (l < n_geo) { fr = P.raw_fg[(b * n_geo + l) * H + n]; ph = P.raw_pg[(b * n_geo + l) * H + n]; }
    else { fr = P.raw_fa[(b * n_color + (l - n_geo)) * H + n]; ph = P.raw_pa[(b * n_color + (l - n_geo)) * H + n]; }
    const float f = __fadd_rn(__fmul_rn(fr, 15.f), 30.f);
    const double inv2pi = 0.15915494309189533576888;
    const double sc = P.film_inv_scale ? (double)P.film_inv_scale[l * H + n] : 1.0;
    fp[i] = (float)((double)f * inv2pi * sc);
    pp[i] = (float)(((double)f * (double)P.film_bias[l * H + n] + (double)ph) * inv2pi);
  }
  __threadfence();
  __syncthreads();
}

}  // namespace fenerf
