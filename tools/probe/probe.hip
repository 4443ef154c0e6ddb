// Hardware probes for kernel design decisions (not part of the product):
//   1. accuracy of v_sin_f32 (input in revolutions) vs double sin, with and without explicit range reduction
//   2. issue rate of v_mfma_f32_32x32x16_f16: same accumulator back-to-back vs 2 / 3 alternating accumulators
//   3. s_barrier cost with 4 waves per workgroup
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ void sin_probe(const float* t, float* out_hw, float* out_hw_red, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = t[i];
  out_hw[i] = __builtin_amdgcn_sinf(x);                 // v_sin_f32: sin(2 pi x)
  float r = x - __builtin_rintf(x);
  out_hw_red[i] = __builtin_amdgcn_sinf(r);
}

template <int NACC>
__global__ void mfma_rate(float* out, long long* cycles, int iters) {
  half8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (_Float16)(0.001f * (threadIdx.x + t)); b[t] = (_Float16)(0.002f * t); }
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u)
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NACC], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ void barrier_cost(long long* cycles, int iters) {
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
  const int n = 1 << 20;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)((i / (double)n - 0.5) * (i % 3 == 0 ? 1.0 : (i % 3 == 1 ? 20.0 : 90.0)));
  float *d_t, *d_a, *d_b;
  hipMalloc(&d_t, n * 4); hipMalloc(&d_a, n * 4); hipMalloc(&d_b, n * 4);
  hipMemcpy(d_t, h.data(), n * 4, hipMemcpyHostToDevice);
  sin_probe<<<n / 256, 256>>>(d_t, d_a, d_b, n);
  std::vector<float> a(n), b(n);
  hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost);
  double e1[3] = {0, 0, 0}, e2[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    double ref = sin(2.0 * M_PI * (double)h[i]);
    int c = i % 3;
    e1[c] = fmax(e1[c], fabs(a[i] - ref));
    e2[c] = fmax(e2[c], fabs(b[i] - ref));
  }
  printf("v_sin_f32 max abs err  |t|<=0.5: raw %.3e reduced %.3e ; |t|<=10: raw %.3e reduced %.3e ; |t|<=45: raw %.3e reduced %.3e\n",
         e1[0], e2[0], e1[1], e2[1], e1[2], e2[2]);

  float* d_o; long long* d_c; hipMalloc(&d_o, 256 * 1024 * 4); hipMalloc(&d_c, 8);
  long long c;
  const int iters = 2000;
  mfma_rate<1><<<1, 64>>>(d_o, d_c, iters); hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
  printf("mfma f16 32x32x16: 1 acc  (dependent)      : %.1f cycles / MFMA (s_memtime ticks)\n", (double)c / (iters * 12));
  mfma_rate<2><<<1, 64>>>(d_o, d_c, iters); hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
  printf("mfma f16 32x32x16: 2 acc  alternating      : %.1f cycles / MFMA\n", (double)c / (iters * 12));
  mfma_rate<3><<<1, 64>>>(d_o, d_c, iters); hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
  printf("mfma f16 32x32x16: 3 acc  alternating      : %.1f cycles / MFMA\n", (double)c / (iters * 12));
  barrier_cost<<<1, 256>>>(d_c, 10000); hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
  printf("s_barrier, 4 waves in lockstep             : %.1f cycles / barrier\n", (double)c / 10000);
  return 0;
}
