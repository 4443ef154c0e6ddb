// Device-side re-pack of the weight streams (fenerf_model_repack): training keeps the parameters on the GPU and changes
// them every optimizer step (train_double_latent_semantic.py: optimizer_G.step() / ema.update()); the packed streams are
// gathers of those parameters through fixed index maps -- for FENERF_PREC_F16X3 after scaling every row by its power-of-two
// scale and splitting into fp16 / bf16 (hi, lo).  Four small kernels write straight into the model's resident buffers;
// the same work issued as ~150 torch ops cost 1.5 ms of launch latency per step.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_trig.h"

namespace fenerf {

// ---- fenerf_trig.h: the kernels hand v_sin_f32 / v_cos_f32 unreduced revolutions; prove once per device that this device reduces them
__global__ void trig_domain_kernel(const float* t, float* out, int n) {
  const int i = threadIdx.x;
  if (i < n) { out[i] = sin2pi(t[i]); out[n + i] = cos2pi(t[i]); }
}
int check_trig_domain() {
  static std::mutex mu;
  static int verdict[64] = {0};       // per device: 0 = not checked, 1 = ok, -1 = failed
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { set_error(std::string("check_trig_domain: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64) return FENERF_OK;
  if (verdict[dev] == 0) {
    const int n = 8;
    const float h_t[n] = {257.25f, -257.25f, 1000.125f, -1000.125f, 70000.75f, -70000.75f, 3000000.25f, 0.375f};
    float h_o[2 * n];
    float *d_t = nullptr, *d_o = nullptr;
    // On a private non-blocking stream: nothing here touches the legacy null stream, so the check neither synchronises with nor is
    // ordered against the caller's streams.  (Model creation itself allocates device memory -- upload_model -- so it is never legal inside
    // a stream capture anyway; the check just does not add a null-stream dependency of its own.)
    hipStream_t st = nullptr;
    if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) == hipSuccess &&
        (e = hipMalloc((void**)&d_t, sizeof(h_t))) == hipSuccess && (e = hipMalloc((void**)&d_o, sizeof(h_o))) == hipSuccess &&
        (e = hipMemcpyAsync(d_t, h_t, sizeof(h_t), hipMemcpyHostToDevice, st)) == hipSuccess) {
      hipLaunchKernelGGL(trig_domain_kernel, dim3(1), dim3(64), 0, st, d_t, d_o, n);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(h_o, d_o, sizeof(h_o), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (d_t) (void)hipFree(d_t);
    if (d_o) (void)hipFree(d_o);
    if (st) (void)hipStreamDestroy(st);
    if (e != hipSuccess) { set_error(std::string("check_trig_domain: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
    verdict[dev] = 1;
    for (int i = 0; i < n; ++i) {
      const double x = (double)h_t[i], r = x - std::nearbyint(x);          // the fp32 argument, reduced exactly
      if (std::fabs((double)h_o[i] - std::sin(6.283185307179586476925 * r)) > 1e-6 ||
          std::fabs((double)h_o[n + i] - std::cos(6.283185307179586476925 * r)) > 1e-6) verdict[dev] = -1;
    }
  }
  if (verdict[dev] < 0) {
    set_error("this device's v_sin_f32 / v_cos_f32 do not reduce arguments beyond +-256 revolutions: rebuild libfenerf_hip.so with "
              "-DFENERF_TRIG_REDUCE=1 (fenerf_amd/csrc/fenerf_trig.h)");
    return FENERF_E_UNSUPPORTED;
  }
  return FENERF_OK;
}

// one wave per scaled row: s = power of two with max|row| * s in [0.5, 1) (1 for an all-zero row) -- row_scales() of
// fenerf_pack.cpp.  scale_fwd[1 + row] = s, scale_bwd[1 + row] = 16 s for FiLM-layer rows (the backward stream carries the
// forward's activation scale), 1 for head rows (true values); index 0 = unscaled elements.
__global__ void repack_row_scale_kernel(const float* flat, const int* row_off, const int* row_len, const int* row_film, int n_rows,
                                        float* scale_fwd, float* scale_bwd) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (row == 0 && lane == 0) { scale_fwd[0] = 1.f; scale_bwd[0] = 1.f; }
  if (row >= n_rows) return;
  const float* p = flat + row_off[row];
  float m = 0.f;
  for (int i = lane; i < row_len[row]; i += 64) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) {
    float s = 1.f;
    if (m > 0.f) { int e; (void)frexpf(m, &e); s = ldexpf(1.f, -e); }
    scale_fwd[1 + row] = s;
    scale_bwd[1 + row] = row_film[row] ? s * F16_ACT_SCALE : 1.f;
  }
}

__global__ void repack_gather_f32_kernel(const float* flat, const int* map, long long n, float* out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = flat[map[i]];
}

// fp16 (hi, lo) halves of the f16x3 forward ring: map = flat index | is_lo << 30 (fenerf_pack_index_map_f16)
__global__ void repack_gather_f16_kernel(const float* flat, const float* scale, const int* scale_id, const int* map, long long n, uint16_t* out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int code = map[i], idx = code & 0x3fffffff;
    const float w = flat[idx] * scale[scale_id[idx]];
    const _Float16 hi = (_Float16)w;
    const _Float16 v = (code >> 30) ? (_Float16)(w - (float)hi) : hi;
    out[i] = __builtin_bit_cast(uint16_t, v);
  }
}

__device__ __forceinline__ uint16_t bf16_rne_bits(float v) {
  const unsigned b = __builtin_bit_cast(unsigned, v);
  return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}

// bf16 (hi, lo) halves of the bf16x3 backward ring: hi / lo entries of 512 halves alternate (fenerf_pack_backward_index_map_bf16)
__global__ void repack_gather_bf16_kernel(const float* flat, const float* scale, const int* scale_id, const int* map, long long n, uint16_t* out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int idx = map[i];
    const float w = flat[idx] * scale[scale_id[idx]];
    const uint16_t hi = bf16_rne_bits(w);
    const float hf = __builtin_bit_cast(float, (unsigned)hi << 16);
    out[i] = ((i >> 9) & 1) ? bf16_rne_bits(w - hf) : hi;
  }
}

// f16x3 result scales behind the fp32 consts: k > 0 -> 1 / (16 scale[k]), k == 0 -> 1 / 16, k < 0 -> 1
__global__ void repack_tail_kernel(const float* scale_fwd, const int* tail, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = tail[i];
  out[i] = k > 0 ? 1.f / (scale_fwd[k] * F16_ACT_SCALE) : (k == 0 ? 1.f / F16_ACT_SCALE : 1.f);
}

static unsigned grid_for(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int launch_repack(FenerfModel* m, const float* flat, const FenerfRepackMaps* r, float* scale_fwd, float* scale_bwd, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool f16 = m->precision == FENERF_PREC_F16X3;
  if (f16) hipLaunchKernelGGL(repack_row_scale_kernel, dim3(r->n_rows > 0 ? r->n_rows : 1), dim3(64), 0, st, flat, r->row_off, r->row_len, r->row_film, r->n_rows, scale_fwd, scale_bwd);
  hipLaunchKernelGGL(repack_gather_f32_kernel, dim3(grid_for((long long)r->n_stream_f32)), dim3(256), 0, st, flat, r->stream_f32, (long long)r->n_stream_f32, m->d_stream);
  hipLaunchKernelGGL(repack_gather_f32_kernel, dim3(grid_for((long long)r->n_consts)), dim3(256), 0, st, flat, r->consts, (long long)r->n_consts, m->d_consts);
  if (f16) {
    hipLaunchKernelGGL(repack_gather_f16_kernel, dim3(grid_for((long long)r->n_stream_h16)), dim3(256), 0, st, flat, scale_fwd, r->scale_id, r->stream_h16,
                       (long long)r->n_stream_h16, reinterpret_cast<uint16_t*>(m->d_stream + r->n_stream_f32));
    hipLaunchKernelGGL(repack_tail_kernel, dim3((unsigned)((r->n_tail + 255) / 256)), dim3(256), 0, st, scale_fwd, r->consts_tail, (int)r->n_tail, m->d_consts + r->n_consts);
  }
  if (m->differentiable) {
    hipLaunchKernelGGL(repack_gather_f32_kernel, dim3(grid_for((long long)r->n_bwd_f32)), dim3(256), 0, st, flat, r->bwd_f32, (long long)r->n_bwd_f32, m->d_bwd_stream);
    if (f16)
      hipLaunchKernelGGL(repack_gather_bf16_kernel, dim3(grid_for((long long)r->n_bwd_b16)), dim3(256), 0, st, flat, scale_bwd, r->scale_id, r->bwd_b16,
                         (long long)r->n_bwd_b16, reinterpret_cast<uint16_t*>(m->d_bwd_stream + r->n_bwd_f32));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("repack launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

}  // namespace fenerf
