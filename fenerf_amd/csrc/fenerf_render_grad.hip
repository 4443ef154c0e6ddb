// Small kernels of the one-call differentiable render (fenerf_render_forward_save / fenerf_render_backward, fenerf_api.cpp): what
// torch elementwise ops did around the native kernels of a generator step in rounds 1-4 -- the sample points of a pass
// (generators.py:468-476, :504: origins + dirs * z, padded to whole 32-point tiles per image), the sums of per-chunk gradients, the
// fold of the two passes' FiLM gradients -- so that a host that is not Python can take a generator step, and so that the Python
// host's step has no glue launches left between the library's.
#include <hip/hip_runtime.h>

#include "fenerf_internal.h"

namespace fenerf {

// pts [B][Pp][3] = origins[ray] + dirs[ray] * z[ray][k]  (mul, then add: the rounding of the reference's torch ops and of the forward
// kernel's rays mode); points P .. Pp - 1 of an image repeat its last point (their output rows take no gradient).  rd [B][Pp][3] = the
// ray's direction per point (generators.py:470-472), or nullptr.
__global__ void render_points_kernel(const float* origins, const float* dirs, const float* z, float* pts, float* rd, int R, int N, long long Pp,
                                     long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long b = i / Pp;
  long long p = i % Pp;
  const long long P = (long long)R * N;
  if (p >= P) p = P - 1;
  const long long ray = b * R + p / N;
  const float zz = z[b * P + p];
  const float ox = origins[ray * 3 + 0], oy = origins[ray * 3 + 1], oz = origins[ray * 3 + 2];
  const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  pts[i * 3 + 0] = __fadd_rn(ox, __fmul_rn(dx, zz));
  pts[i * 3 + 1] = __fadd_rn(oy, __fmul_rn(dy, zz));
  pts[i * 3 + 2] = __fadd_rn(oz, __fmul_rn(dz, zz));
  if (rd) { rd[i * 3 + 0] = dx; rd[i * 3 + 1] = dy; rd[i * 3 + 2] = dz; }
}

int launch_render_points(int B, int R, int N, long long Pp, const float* origins, const float* dirs, const float* z, float* pts, float* rd,
                         void* stream) {
  const long long total = (long long)B * Pp;
  hipLaunchKernelGGL(render_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, origins, dirs, z, pts, rd, R, N, Pp, total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("render points launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

// rows [nb][Pp][C] -> [nb][P][C] (or back, zero-filling the pad rows): the composite kernels work on unpadded rays
__global__ void pad_rows_kernel(const float* src, float* dst, long long P, long long Pp, int C, long long total, int to_padded) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long per = (to_padded ? Pp : P) * C;
  const long long b = i / per, r = i % per;
  if (to_padded) dst[i] = r < P * C ? src[b * P * C + r] : 0.f;
  else dst[i] = src[b * Pp * C + r];
}
int launch_pad_rows(const float* src, float* dst, long long nb, long long P, long long Pp, int C, bool to_padded, void* stream) {
  const long long total = nb * (to_padded ? Pp : P) * C;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, P, Pp, C, total, to_padded ? 1 : 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("pad rows launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

// dst_k += src_k for up to MULTI_ADD_MAX tensors in one launch (blockIdx.y = tensor): the sum of a later backward chunk's gradients into
// the first one's -- one fp32 add per element, in chunk order: what torch._foreach_add_ did
__global__ void multi_add_kernel(MultiAdd J) {
  const int k = blockIdx.y;
  float* d = J.dst[k];
  const float* s = J.src[k];
  const long long n = J.n[k];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) d[i] += s[i];
}
int launch_multi_add(const MultiAdd& J, void* stream) {
  if (J.count <= 0) return FENERF_OK;
  long long nmax = 0;
  for (int k = 0; k < J.count; ++k) nmax = J.n[k] > nmax ? J.n[k] : nmax;
  long long bx = (nmax + 255) / 256;
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(multi_add_kernel, dim3((unsigned)bx, (unsigned)J.count), dim3(256), 0, (hipStream_t)stream, J);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("multi add launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

// out[b][i] = a[b][i] + a[B + b][i]: the two passes of a hierarchical render share an image's FiLM parameters (four tensors, one launch)
__global__ void film_fold_kernel(FilmFold J) {
  const int k = blockIdx.y;
  const long long n = (long long)J.B * J.row[k];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    J.out[k][i] = J.in[k][i] + J.in[k][n + i];
}
int launch_film_fold(const FilmFold& J, void* stream) {
  long long nmax = 0;
  for (int k = 0; k < 4; ++k) nmax = (long long)J.B * J.row[k] > nmax ? (long long)J.B * J.row[k] : nmax;
  hipLaunchKernelGGL(film_fold_kernel, dim3((unsigned)((nmax + 255) / 256), 4), dim3(256), 0, (hipStream_t)stream, J);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("film fold launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

// ---- selection step of the exact-sparsity backward (fenerf_sparse_select; generators/autograd.py SparseHierarchicalRenderFunction) ----
// Sample s of image b: s < P the coarse pass's, else the fine pass's sample s - P.  A sample is kept when its row of upstream gradients
// is not all zero (NaN != 0: kept).  Kept samples go to slots 0 .. count - 1 of their image in sample order.
// Pass 1: one ballot per wave (mask of kept lanes) + one count per 256-sample block.
__global__ void __launch_bounds__(256) sparse_count_kernel(const float* d_coarse, const float* d_fine, const long long* images, long long P, int C,
                                                           unsigned long long* masks, int* block_counts, int* counts, int B) {
  const int b = blockIdx.y, t = threadIdx.x;
  const long long img = images ? images[b] : b;          // where image b of this call sits in the inputs
  const long long s = (long long)blockIdx.x * 256 + t;
  bool keep = false;
  if (s < (d_fine ? 2 * P : P)) {      // d_fine == nullptr: one pass (a render without importance resampling)
    const float* row = s < P ? d_coarse + (img * P + s) * C : d_fine + (img * P + (s - P)) * C;
    if ((C & 1) == 0) {        // even C (22 here): rows are 8-byte aligned
      const float2* r2 = reinterpret_cast<const float2*>(row);
      for (int c = 0; c < C / 2; ++c) { const float2 v = r2[c]; keep |= (v.x != 0.f) | (v.y != 0.f); }
    } else {
      for (int c = 0; c < C; ++c) keep |= (row[c] != 0.f);
    }
  }
  const unsigned long long m = __ballot(keep);
  __shared__ int wave_n[4];
  if ((t & 63) == 0) {
    masks[((long long)b * gridDim.x + blockIdx.x) * 4 + (t >> 6)] = m;
    wave_n[t >> 6] = __popcll(m);
  }
  __syncthreads();
  if (t == 0) {
    block_counts[(long long)b * gridDim.x + blockIdx.x] = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
    if (blockIdx.x == 0 && b == 0) counts[B] = 0;       // the overflow flag pass 2 may raise
  }
}

// Pass 2: one workgroup per image turns its block counts into their exclusive prefix sums in place (a running carry over pieces of 1024
// blocks: 49,152 samples at 128 x 128 x 24+24 are 3 pieces, 6.3 M at 256 x 256 x 48+48 are 24); counts[b] = the image's total, counts[B] = 1
// if some image kept more than cap.
__global__ void __launch_bounds__(1024) sparse_scan_kernel(int* block_counts, int nblk, long long cap, int* counts, int B) {
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  int* bc = block_counts + (long long)b * nblk;
  __shared__ int wave_sum[16];
  __shared__ int carry_s;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int j = base + t;
    const int v = j < nblk ? bc[j] : 0;
    int incl = v;                                           // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wave_sum[w] = incl;
    __syncthreads();
    int before = carry_s;
    for (int k = 0; k < w; ++k) before += wave_sum[k];
    if (j < nblk) bc[j] = before + incl - v;
    __syncthreads();
    if (t == 1023) carry_s = before + incl;
    __syncthreads();
  }
  if (t == 0) {
    counts[b] = carry_s;
    if (carry_s > cap) counts[B] = 1;
  }
}

// Pass 3: slot = kept samples of the image before this one; a kept sample writes its point (origins + dirs * z: mul, then add, as every
// other kernel of the path rounds it), its ray direction and its gradient row to its slot; slots count .. cap - 1 get sample 0's point and
// direction and a zero row.  Kept samples beyond cap are dropped (pass 2 raised the flag: the caller treats it as an error).
__global__ void __launch_bounds__(256) sparse_gather_kernel(const float* d_coarse, const float* d_fine, const float* z_coarse, const float* z_fine,
                                                            const float* origins, const float* dirs, const long long* images, int R, int N, int C,
                                                            long long cap, const unsigned long long* masks, const int* block_counts, float* pts, float* rd,
                                                            float* d_sel, const int* counts) {
  const int b = blockIdx.y, t = threadIdx.x, nblk = gridDim.x, blk = blockIdx.x;
  const long long img = images ? images[b] : b;
  const long long P = (long long)R * N;
  const int before = block_counts[(long long)b * nblk + blk];      // exclusive prefix (pass 2)
  const int all = counts[b];
  const int w = t >> 6, lane = t & 63;
  const unsigned long long* mw = masks + ((long long)b * nblk + blk) * 4;
  long long slot = before;
  for (int k = 0; k < w; ++k) slot += __popcll(mw[k]);
  const unsigned long long m = mw[w];
  slot += __popcll(m & ((1ull << lane) - 1ull));
  if (((m >> lane) & 1ull) && slot < cap) {
    const long long s = (long long)blk * 256 + t;
    const bool fine = s >= P;
    const long long q = fine ? s - P : s;
    const float* row = (fine ? d_fine : d_coarse) + (img * P + q) * C;
    const float zz = (fine ? z_fine : z_coarse)[img * P + q];
    const long long ray = img * R + q / N;
    const long long o = (long long)b * cap + slot;
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    pts[o * 3 + 0] = __fadd_rn(origins[ray * 3 + 0], __fmul_rn(dx, zz));
    pts[o * 3 + 1] = __fadd_rn(origins[ray * 3 + 1], __fmul_rn(dy, zz));
    pts[o * 3 + 2] = __fadd_rn(origins[ray * 3 + 2], __fmul_rn(dz, zz));
    if (rd) { rd[o * 3 + 0] = dx; rd[o * 3 + 1] = dy; rd[o * 3 + 2] = dz; }
    if ((C & 1) == 0) {
      const float2* r2 = reinterpret_cast<const float2*>(row);
      float2* o2 = reinterpret_cast<float2*>(d_sel + o * C);
      for (int c = 0; c < C / 2; ++c) o2[c] = r2[c];
    } else {
      for (int c = 0; c < C; ++c) d_sel[o * C + c] = row[c];
    }
  }
  // the pad slots of the image, shared out over its blocks
  const long long ray0 = img * R;
  for (long long sl = (long long)all + (long long)blk * 256 + t; sl < cap; sl += (long long)nblk * 256) {
    const long long o = (long long)b * cap + sl;
    const float zz = z_coarse[img * P];
    const float dx = dirs[ray0 * 3 + 0], dy = dirs[ray0 * 3 + 1], dz = dirs[ray0 * 3 + 2];
    pts[o * 3 + 0] = __fadd_rn(origins[ray0 * 3 + 0], __fmul_rn(dx, zz));
    pts[o * 3 + 1] = __fadd_rn(origins[ray0 * 3 + 1], __fmul_rn(dy, zz));
    pts[o * 3 + 2] = __fadd_rn(origins[ray0 * 3 + 2], __fmul_rn(dz, zz));
    if (rd) { rd[o * 3 + 0] = dx; rd[o * 3 + 1] = dy; rd[o * 3 + 2] = dz; }
    for (int c = 0; c < C; ++c) d_sel[o * C + c] = 0.f;
  }
}

size_t sparse_select_workspace_bytes(int B, long long P) {
  const long long nblk = (2 * P + 255) / 256;
  return (size_t)B * nblk * (4 * sizeof(unsigned long long) + sizeof(int));
}

int launch_sparse_select(int B, int R, int N, int C, long long cap, const float* d_coarse, const float* d_fine, const float* z_coarse,
                         const float* z_fine, const float* origins, const float* dirs, const long long* images, float* pts, float* rd, float* d_sel,
                         int* counts, void* workspace, void* stream) {
  const long long P = (long long)R * N;
  const long long nblk = ((d_fine ? 2 * P : P) + 255) / 256;
  unsigned long long* masks = (unsigned long long*)workspace;
  int* block_counts = (int*)(masks + (size_t)B * nblk * 4);
  const dim3 grid((unsigned)nblk, (unsigned)B);
  hipLaunchKernelGGL(sparse_count_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_coarse, d_fine, images, P, C, masks, block_counts, counts, B);
  hipLaunchKernelGGL(sparse_scan_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, block_counts, (int)nblk, cap, counts, B);
  hipLaunchKernelGGL(sparse_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_coarse, d_fine, z_coarse, z_fine, origins, dirs, images, R, N, C,
                     cap, masks, block_counts, pts, rd, d_sel, counts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("sparse select launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

}  // namespace fenerf
