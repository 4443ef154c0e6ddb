// sin / cos of a FiLM phase carried in REVOLUTIONS (t = theta / 2 pi), for every SIREN kernel of the library.
//
// The reference evaluates torch.sin(freq * x + phase_shift) on fp32 radians of any magnitude (siren/siren.py:113-123); inversion
// (inverse_render_double_semantic.py:370-410: Adam on unconstrained frequency / phase offsets) and trained checkpoints are free to
// leave the init range.  v_sin_f32 / v_cos_f32 take revolutions.  The GCN3 / Vega ISA manuals define them on [-256, +256] only ("out
// of range input results in float 0") -- the round-3 review's concern.  On gfx950 that restriction is gone, and round 4 established it
// three ways before deciding what to ship:
//   * the compiler: hipcc --offload-arch=gfx950 lowers __sinf(x) to  v_mul_f32 (1 / 2 pi); v_sin_f32  with NO v_fract_f32 in between
//     (LLVM inserts one only on subtargets with FeatureTrigReducedRange; gfx950 is not one);
//   * the hardware: tools/probe/sin_domain_probe.hip sweeps |t| over [2^-2, 2^31) revolutions, 65,536 arguments per octave: max abs
//     error of the raw instruction 1.25e-7 up to 2^21 and exact zeros only where sin really is 0 (integers), not one spurious zero
//     (profiles/r04_vsin_domain_probe.txt);
//   * the kernels: every family (f32 / f16x3 forward, per-point and one-launch local kernels, both chain kernels, both weight-gradient
//     kernels) against fp64 with arguments of 45 .. 1,524 revolutions, on the unreduced build (profiles/
//     r04_sine_domain_tests_unreduced_library.txt) -- and tests/golden/*_bigfilm.npz pin the same to the reference's own torch.sin.
// So the argument goes to the instruction as it is (FENERF_TRIG_REDUCE = 0): no cost, and arguments in (-1, 0) keep their last bit.
// What guards the assumption instead of an instruction per sine:
//   * fenerf_model_create / fenerf_local_model_create run check_trig_domain() once per device (fenerf_repack.hip): a one-wave kernel
//     evaluates v_sin_f32 / v_cos_f32 at +-257.25, +-1000.125, +-70000.75 and 3e6 + 0.25 revolutions and the create call FAILS
//     (FENERF_E_UNSUPPORTED, message names this file) if any of them is off by more than 1e-6 -- no silent zeros on a device that
//     behaves like the old manuals;
//   * the -m gpu tests above run at every round end.
// Build options, measured with the in-kernel cycle stamps (profiles/r04_trig_reduce_ab.txt): FENERF_TRIG_REDUCE = 1 puts ONE
// v_fract_f32 in front of every sin / cos (f16x3 forward 2,611,000 -> 2,615,000 cycles per launch, +0.16 %; exact-fp32 forward +0.9 %;
// generator step +1.2 %; arguments in (-1, 0) are rounded to the grid of [0.5, 1): 2.6e-7 instead of 1.25e-7 max error); = 2 the
// exact two-instruction reduction t - rint(t) (+2.2 % / +1.4 % / +2.4 %).  Either makes the library independent of the instruction's
// domain; the same reduction then sits in front of the forward's sin, the chain kernels' cos and the weight-gradient kernels'
// recomputed sin, so a recomputed activation stays bitwise the forward's.
#pragma once
#include <hip/hip_runtime.h>

#ifndef FENERF_TRIG_REDUCE
#define FENERF_TRIG_REDUCE 0
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

// sin(2 pi t), cos(2 pi t) for any finite t.  Measured on MI355X: max abs error of v_sin_f32 1.25e-7 -- tighter than a degree-9
// polynomial evaluated in fp32 (2.1e-7) and one quarter-rate instruction instead of thirteen.
__device__ __forceinline__ float sin2pi(float t) { return sin_rev_reduced(rev_reduce(t)); }
__device__ __forceinline__ float cos2pi(float t) { return cos_rev_reduced(rev_reduce(t)); }

}  // namespace fenerf
