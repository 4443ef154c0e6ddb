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

}  // namespace fenerf
