// Non-temporal access to streams that are touched exactly once (tapes, dtheta dumps: GBs per pass), so that they do not
// evict what the kernels keep re-reading from L2 (the packed weight streams).  Measured on the bf16x3 chain kernel, same box:
// 1.62 -> 1.50 ms per 196,608 points.
#pragma once
#include <hip/hip_runtime.h>

namespace fenerf {

typedef float nfloat4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 nt_load(const float4* p) {
  const nfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const nfloat4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
typedef float nfloat2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 nt_load(const float2* p) {
  const nfloat2 v = __builtin_nontemporal_load(reinterpret_cast<const nfloat2*>(p));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void nt_store(float4* p, float a, float b, float c, float d) {
  const nfloat4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<nfloat4*>(p));
}

}  // namespace fenerf
