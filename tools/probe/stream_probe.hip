// Probe: how fast can one wave per SIMD feed v_mfma_f32_32x32x16_bf16 triples from a PRIVATE register-ring stream out of L2
// (the structure of the fp32 SIREN kernels, at the bf16x3 data rate: 2 KiB of A operands per 96 matrix cycles per wave)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PF>
__global__ __launch_bounds__(256, 1) void stream_mfma(const float4* stream, long long entries, int passes, float* out) {
  const int lane = threadIdx.x & 63;
  const float4* base = stream + lane;
  float4 ring[PF];
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 b = __builtin_bit_cast(bf16x8, make_float4(1.f, 2.f, 3.f, 4.f));
  for (int p = 0; p < passes; ++p) {
    const float4* ptr = base;
#pragma unroll
    for (int i = 0; i < PF; ++i) { ring[i] = *ptr; ptr += 64; }
    for (long long e = 0; e + PF <= entries; e += PF) {
#pragma unroll
      for (int i = 0; i < PF; i += 2) {
        const float4 hi = ring[i], lo = ring[i + 1];
        ring[i] = *ptr; ptr += 64;
        ring[i + 1] = *ptr; ptr += 64;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, lo), b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hi), b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hi), b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PF>
void run(const float4* d_stream, long long entries, float* d_out, const char* name) {
  const int passes = 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  stream_mfma<PF><<<256, 256>>>(d_stream, entries, 1, d_out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  stream_mfma<PF><<<256, 256>>>(d_stream, entries, passes, d_out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double mfma = (double)passes * (entries / 2) * 3;                       // per wave
  const double cyc = ms * 1e-3 * 2.4e9;
  printf("%s: %.3f ms, %.1f cycles per MFMA triple (96 = matrix-bound), stream %.1f TB/s aggregate\\n", name, ms,
         cyc / (mfma / 3), (double)passes * entries * 1024.0 * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  const long long entries = 2816;      // 2.9 MB stream (the backward stream of the H=256 model), + ring tail
  float4* d_stream; float* d_out;
  hipMalloc(&d_stream, (entries + 64) * 1024);
  hipMemset(d_stream, 0, (entries + 64) * 1024);
  hipMalloc(&d_out, 256 * 256 * 4);
  run<8>(d_stream, entries, d_out, "ring depth 8 ");
  run<16>(d_stream, entries, d_out, "ring depth 16");
  run<32>(d_stream, entries, d_out, "ring depth 32");
  return 0;
}
