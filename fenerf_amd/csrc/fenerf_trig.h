// sin / cos of a FiLM phase carried in REVOLUTIONS (t = theta / 2 pi), for every SIREN kernel of the library.
//
// The reference evaluates torch.sin(freq * x + phase_shift) on fp32 radians of any magnitude (siren/siren.py:113-123); inversion
// (inverse_render_double_semantic.py:370-410: Adam on unconstrained frequency / phase offsets) and trained checkpoints are free to
// leave the init range.  v_sin_f32 / v_cos_f32 take revolutions; the GFX9-family ISA manuals define them on [-256, +256] only ("out of
// range input results in float 0") and LLVM therefore puts a v_fract_f32 in front of them on this family.  So does every kernel
// here: the argument is reduced to [0, 1) by ONE full-rate v_fract_f32 -- the same instruction in front of the forward's sin, the
// chain kernels' cos and the weight-gradient kernels' recomputed sin, so a recomputed activation is still bitwise the forward's.
//   * exactness: t - floor(t) is exact in fp32 for t >= 0 and for t <= -1 (the result is no larger than |t| and a multiple of
//     ulp(t)); for -1 < t < 0 the result t + 1 is rounded to the fp32 grid of [0.5, 1): at most 2^-25 revolutions = 1.9e-7 rad,
//     the size of v_sin_f32's own error (1.2e-7) and below the rounding the reference's fp32 radians carry at |theta| >= 2 rad.
//   * cost, measured with the in-kernel cycle stamps (profiles/r04_trig_reduce_ab.txt): f16x3 forward 2,611,000 -> 2,615,000
//     cycles per launch (+0.16 %), exact-fp32 forward +0.9 %, generator step +1.2 %.
//   * FENERF_TRIG_REDUCE = 2 builds the EXACT two-instruction reduction t - rint(t) (v_rndne_f32 + v_sub_f32; |t| <= 0.5 passes
//     through bit for bit): +2.2 % / +1.4 % / +2.4 % on the same three -- measured, not shipped.  = 0 builds the unreduced round-3
//     arithmetic (A/B only).  On the MI355X boxes measured, v_sin_f32 / v_cos_f32 did NOT return zeros beyond 256 revolutions
//     (tools/probe/sin_domain_probe.hip, profiles/r04_vsin_domain_probe.txt) -- the reduction is kept because the documented
//     domain, not one stepping's behaviour, is what a library may rely on.
#pragma once
#include <hip/hip_runtime.h>

#ifndef FENERF_TRIG_REDUCE
#define FENERF_TRIG_REDUCE 1
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
