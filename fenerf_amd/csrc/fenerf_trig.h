// sin / cos of a FiLM phase carried in REVOLUTIONS (t = theta / 2 pi), for every SIREN kernel of the library.
//
// The reference evaluates torch.sin(freq * x + phase_shift) on fp32 radians of any magnitude (siren/siren.py:113-123); inversion
// (inverse_render_double_semantic.py:370-410: Adam on unconstrained frequency / phase offsets) and trained checkpoints are free to
// leave the init range.  v_sin_f32 / v_cos_f32 take revolutions and are only defined on [-256, +256]: beyond that the hardware
// returns sin = 0 / cos = 1 without any error.  So the argument is first reduced to [-0.5, +0.5] by  t - rint(t)  (v_rndne_f32 +
// v_sub_f32).  The reduction is EXACT in fp32 for every finite t (the difference of a float and the nearest integer has no more
// significant bits than the float: |t| < 2^23 is a Sterbenz-type subtraction, above that t is an integer and the result is 0,
// which is what fp32 radians can no longer resolve either), arguments with |t| <= 0.5 pass through bit for bit, and the same
// two instructions sit in front of the forward's sin, the chain kernels' cos and the weight-gradient kernels' recomputed sin,
// so a recomputed activation is still bitwise the forward's.
// (v_fract_f32 is one instruction cheaper but rounds: t + 1 for -1 < t < 0 loses up to 3e-8 revolutions = 1.9e-7 rad, and lands
// small negative arguments next to 1.0.  FENERF_TRIG_REDUCE = 1 builds it, 0 builds the unreduced round-3 kernels: A/B only,
// profiles/r04_trig_reduce_ab.txt.)
#pragma once
#include <hip/hip_runtime.h>

#ifndef FENERF_TRIG_REDUCE
#define FENERF_TRIG_REDUCE 2
#endif

namespace fenerf {

__device__ __forceinline__ float rev_reduce(float t) {
#if FENERF_TRIG_REDUCE == 2
  return t - __builtin_rintf(t);
#elif FENERF_TRIG_REDUCE == 1
  return __builtin_amdgcn_fractf(t);
#else
  return t;
#endif
}

// sin(2 pi t) / cos(2 pi t) on an argument that rev_reduce() has already brought into the hardware's domain
__device__ __forceinline__ float sin_rev_reduced(float r) { return __builtin_amdgcn_sinf(r); }
__device__ __forceinline__ float cos_rev_reduced(float r) { return __builtin_amdgcn_cosf(r); }

// sin(2 pi t), cos(2 pi t) for any finite t.  Measured on MI355X (tools/probe/probe.hip): max abs error of v_sin_f32 1.2e-7 on
// reduced arguments -- tighter than a degree-9 polynomial evaluated in fp32 (2.1e-7) and one quarter-rate instruction instead of
// thirteen.
__device__ __forceinline__ float sin2pi(float t) { return sin_rev_reduced(rev_reduce(t)); }
__device__ __forceinline__ float cos2pi(float t) { return cos_rev_reduced(rev_reduce(t)); }

}  // namespace fenerf
