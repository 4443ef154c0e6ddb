// FiLM-SIREN radiance field, f16x3 mode, no-grad forward: 16-point waves, two waves per SIMD (gfx950 / MI355X).
//
// Every fp32 product of the dense layers is evaluated as wl*xh + wh*xl + wh*xh on the fp16 matrix pipe with fp32 accumulate
// (activations never leave their lane).  Round 1 ran this arithmetic, on the SAME packed weight stream, in a different execution
// shape (fenerf_siren_f16s.hip, retired in round 3: git history):
//
//   * that kernel ran one 32-point wave per SIMD at ~450 registers.  Its wave issues in order, so every
//     global_load_lds (100-185 cycles of issue each in a busy phase), every LDS wait and every barrier sits in the MFMA
//     stream: 51-53 % matrix-pipe utilisation.
//   * Here a workgroup is 8 waves = 2 per SIMD, each owning 16 points on v_mfma_f32_16x16x32_f16.  Activations halve to
//     64 (x) + 64 (y) registers per wave, everything fits 256 registers, and the SIMD's scheduler issues one wave's MFMAs
//     under the other's DMA / LDS / VALU / barrier time.  One workgroup barrier per two 8-KiB chunks; A operands are read
//     from the ring two k32-steps ahead, FiLM parameters one.  Measurements, the timing experiments (`make wexp`) and why
//     the kernel ends up 4-5 % (not 30 %) ahead of the 32-point one: DESIGN.md 4.1c, profiles/r02_siren16w_experiments.md.
//
// One stream, two consumers.  The packer lays an entry out for the 32x32x16 MFMA (lane (row, h), 8 halves = k slots of
// lane-half h).  A 16x16x32 A operand (16 rows x 32 k) is the union of halves of TWO consecutive k16 entries, so the
// LDS-DMA does the re-tiling: global_load_lds takes a per-lane global address, and each of the 8 waves fetches, per 8-KiB
// chunk, exactly one ready-made 1-KiB A operand (k32-step spl, row tile rt, hi/lo) by pointing its lanes at the right
// 16-byte pieces of the old entries.  Ring reads are then linear ds_read_b128 (conflict-free).  Derivation of the maps:
//
//   old entry (k16-step s16, hi|lo), piece 32 h + row, slot t  =  W[32 nb + row][feat16_of(s16, h, t)]
//   16x16x32 MFMA: A lane (i, kg) slot t = A[i][k(kg, t)], B lane (n, kg) slot t = B[k(kg, t)][n], C/D lane (n, g) reg r = D[4 g + r][n]
//   A operand (k32-step sp, row tile rt): lane (i = 4 gi + r, kg)  <-  old entry s16 = 2 sp + (kg >> 1), piece 32 (kg & 1) + row(rt, i)
//        row(rt, 4 gi + r) = 16 (gi >> 1) + 4 (gi & 1) + 8 rt + r
//   so lane group g ends n-block nb holding, in acc[rt][r], feature 32 nb + 16 (g >> 1) + 4 (g & 1) + 8 rt + r -- exactly the
//   features feat16_of(2 nb + (g >> 1), g & 1, 4 rt + r) the packer put into lane group g's k slots of k32-step nb: the
//   accumulators, FiLM'ed and split, ARE the next layer's B operand {acc[0][0..3], acc[1][0..3]}.
//
// Three uses of one kernel template, siren16w_kernel<H, GRID, SAVE, FUSED>:
//   <.., false, false>  the no-grad forward of fenerf_siren_forward / the two SIREN launches of fenerf_render_forward;
//   <.., 1 | 2, false>  forward-save (fenerf_siren_forward_save): the same tiles, every FiLM layer's phase also leaves as the tape --
//                       SAVE = 1: the raw fp32 accumulators (fenerf_layout.h "Tape"); SAVE = 2 (round 5): frac(theta) as 16-bit fixed
//                       point (fenerf_layout.h "16-bit tape"): half the bytes and half the store instructions;
//   <.., 0, false, 1|2> the opt-in reduced-precision forwards (round 5, fenerf_model_set_forward_mode): TERMS2 = 1 ("f16x2") drops the
//                       wl*xh term of every product -- weights as one fp16 (their lo halves are neither fetched by the LDS-DMA nor read from
//                       the ring nor multiplied) --; TERMS2 = 2 keeps three terms through the geometry trunk and the label / sigma
//                       head (sigma -- the output the inverse-CDF resampling and the 0.9 fill threshold are sensitive to -- and the labels
//                       are then the default's bit for bit) and two in the colour layers and the rgb head;
//   <.., false, true >  the whole hierarchical render in ONE launch (round 4, fenerf_set_render_fusion): ray groups of whole octs, the
//                       rays composited by the workgroup's own waves between its coarse and fine tiles -- see FuseArgs below and
//                       profiles/r04_render_one_launch.md for why it is selectable and not the default (6 % slower).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fenerf_composite_ray.h"
#include "fenerf_film.h"
#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_trig.h"


namespace fenerf {
namespace w16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2 __attribute__((ext_vector_type(2)));

#define MFMA16W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MFMA32W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int CH = FENERF_CH;        // old entries (KiB) per chunk = one A operand per wave
constexpr int DPF = FENERF_DPF;      // chunks in flight ahead of the chunk being consumed
constexpr int NSLOT = FENERF_NSLOT;  // LDS ring slots; every stage is a whole number of ring revolutions (packer)
constexpr int NWAVE = 8;
static_assert(CH == NWAVE, "one 1-KiB A operand per wave and chunk");
static_assert(NSLOT >= DPF + 2, "a slot is refilled two barriers after its last reader issued its reads");

__device__ __forceinline__ half8 as_half8(const float4& v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// LDS-DMA of one KiB: lane i's 16 bytes at g_lane  ->  lds_uniform + 16 i.  Inline asm on purpose: with the builtin hipcc
// tracks the DMA as a pending LDS write and drains the queue (vmcnt(0)) before every ring read.
__device__ __forceinline__ void glds_1k(const char* g_lane, unsigned lds_uniform) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(g_lane), "s"(lds_uniform)
      : "memory");
}
// saddr form: lane i's 16 bytes at g_uniform + voff  ->  lds_uniform + 16 i  (one VGPR of address instead of two)
__device__ __forceinline__ void glds_1k_s(const void* g_uniform, unsigned voff, unsigned lds_uniform) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(g_uniform), "s"(lds_uniform)
      : "memory");
}
// An opaque copy of a lane-derived value.  LICM hoists lane-only address arithmetic out of the tile loop, where it stays live
// through every layer (58 such registers at first count) until the allocator spills it INTO the stream loop -- and scratch
// traffic there would break the counted vmcnt waits.  Deriving addresses from a fresh opaque copy at each use site keeps
// them local.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// a wave-uniform pointer the compiler computed with vector instructions (64-bit multiplies) -> SGPRs, for "s" asm operands
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// Differentiable mode (forward-save): the raw accumulators of every FiLM layer leave as the tape (fenerf_layout.h "Tape": register
// dumps of 32-point tiles; wave w of the workgroup owns half (w & 1) of tile32 = tile16 >> 1).  One fire-and-forget 1-KiB wave
// store per (n-block, row tile): stores are not loads -- they only make the counted vmcnt waits stricter.  asm: uniform base
// in SGPRs + one VGPR of lane offset; the s_nop is the hazard slot behind a > 8-byte store whose data registers are
// overwritten next (the compiler does not look inside an asm).
template <int MODE> struct TapeW { const char* base; };   // (tile32, layer) block of the tape (uniform), or unused
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef FENERF_ST_POLICY
#define FENERF_ST_POLICY "nt"      // cache policy of the fire-and-forget tape / d(theta) stores (A/B builds: profiles/r06_store_policy_ab.txt)
#endif
__device__ __forceinline__ void st_f4_nt(const void* g_uniform, unsigned voff, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 " FENERF_ST_POLICY "\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(g_uniform) : "memory");
}
__device__ __forceinline__ void st_u4_nt(const void* g_uniform, unsigned voff, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 " FENERF_ST_POLICY "\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(g_uniform) : "memory");
}
// SAVE = 2: the four dwords (two 16-bit phases each) of an n-block that wait for the n-block's last epilogue piece: ONE 16-byte store
// per lane and n-block, [nb][16-point tile][lane][slot 4 rt + r] (the bf16 dump's layout, dump16_feature)
struct TapeQ { unsigned q[4]; };
// frac(theta) in revolutions -> round-to-nearest 16-bit fixed point in the low half of the result (65536 wraps to 0 = the same phase):
// fr * 2^16 + 2^23 rounds to an integer in the mantissa (RNE)
__device__ __forceinline__ unsigned phase_u16(float theta) {
  return __builtin_bit_cast(unsigned, __builtin_fmaf(__builtin_amdgcn_fractf(theta), 65536.f, 8388608.f));
}
#define LDS_FENCE() asm volatile("" ::: "memory")
#ifndef FENERF_WAVE_HALF_COPIES
#define FENERF_WAVE_HALF_COPIES 1     // 0: rounds 2-5 (the wave half is a run-time flag inside the stream loop); A/B builds only
#endif
#ifndef FENERF_EXP_DEPHASE
#define FENERF_EXP_DEPHASE 0          // 1: experiment, measured and dropped in round 6 (+ 8 % cycles): see chunk_step
#endif

struct WStream {
  bool skip;                   // f16x2: this wave's operand is a weight lo half -- never fetched (g_next still advances)
  unsigned long long g_next;   // global address of the next chunk to issue (uniform)
  unsigned voff;               // this lane's byte offset inside a chunk (the re-tiling permutation)
  unsigned ring_lds;           // LDS byte address of ring slot 0 + wave * 1024 (for M0)
  const char* ring_lane;       // generic pointer to ring slot 0 + lane * 16 (for the ds_reads)
  bool early;                  // waves 0-3: DMA at the top of a chunk step; waves 4-7: half a step later (ws_step)
};

__device__ __forceinline__ void ws_issue(WStream& w, int slot) {
  const unsigned m0 = w.ring_lds + (unsigned)slot * (CH * 1024);
  if (!w.skip)
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(w.voff), "s"(w.g_next), "s"(m0)
        : "memory");
  w.g_next += CH * 1024;
}

// A operands of one k32-step (both row tiles): ring slot layout = operand index (spl * 2 + rt) * 2 + hl, 1 KiB each
struct AK { float4 hi[2], lo[2]; };
__device__ __forceinline__ void ws_read_lo(AK& a, const WStream& w, int slot, int spl) {
  const float4* p = reinterpret_cast<const float4*>(w.ring_lane + slot * (CH * 1024) + spl * 4096);
  a.lo[0] = p[1 * 64]; a.lo[1] = p[3 * 64];
}
__device__ __forceinline__ void ws_read_hi(AK& a, const WStream& w, int slot, int spl) {
  const float4* p = reinterpret_cast<const float4*>(w.ring_lane + slot * (CH * 1024) + spl * 4096);
  a.hi[0] = p[0 * 64]; a.hi[1] = p[2 * 64];
}
template <bool LO = true>
__device__ __forceinline__ AK ws_read(const WStream& w, int slot, int spl) {
  AK a;
  if (LO) ws_read_lo(a, w, slot, spl);
  else { a.lo[0] = make_float4(0.f, 0.f, 0.f, 0.f); a.lo[1] = a.lo[0]; }
  ws_read_hi(a, w, slot, spl);
  return a;
}

// Top of pipeline step i of a stage: make the next chunks visible to every wave, then (waves 0-3) issue chunk i + D.
// The two waves of a SIMD (w and w + 4) issue their LDS-DMA half a chunk step apart -- waves 4-7 at the top of the chunk's
// second k32-step (ws_issue_late): an LDS-DMA blocks its wave's issue for 60-185 cycles, and issued by both waves at the same
// point of the same program those stalls coincide and the matrix pipe idles under both (+1.3 % on real weights, +2.8 % at
// full clock, profiles/r02_siren16w_experiments.md section 5).
__device__ __forceinline__ void ws_step(WStream& w, int i, bool early) {
  // ONE barrier per TWO chunk steps (even i; stages are whole ring revolutions, so the parity of i is the parity of the global
  // step).  Chunk i + 1 is read during step i and chunk i + 2 during the odd step i + 1, so barrier i publishes both: every wave
  // first waits for its own KiB of chunks <= i + 2 (chunks i + 1 .. i + D - 1 are in flight here: vmcnt(D - 3)).  The DMA of
  // step i (chunk i + D, into the slot of chunk i - 2) is issued BEHIND the barrier: every wave is then past step i - 1; the
  // DMA of the odd step i + 1 overwrites chunk i - 1, consumed by every wave before it arrived at barrier i.  (The barrier
  // costs 8 % of the kernel at full clock, profiles/r02_siren16w_experiments.md.)
  const bool sync = (i & 1) == 0;
  if (sync) {
    WAIT_VMCNT(DPF - 3);
    __builtin_amdgcn_s_barrier();
  }
  LDS_FENCE();
  if (early)
    ws_issue(w, (i + DPF) % NSLOT);
}
__device__ __forceinline__ void ws_issue_late(WStream& w, int i, bool early) {
  if (!early) ws_issue(w, (i + DPF) % NSLOT);
}

// 6 MFMAs of one k32-step, the two row tiles interleaved (dependent MFMAs are 2 apart): wl*xh + wh*xl + wh*xh  (LO = false: 4, no wl*xh)
template <bool LO = true>
__device__ __forceinline__ void kstep_mfma(f32x4 (&acc)[2], const AK& a, const half8& bh, const half8& bl) {
  if (LO) {
    acc[0] = MFMA16W(as_half8(a.lo[0]), bh, acc[0]);
    acc[1] = MFMA16W(as_half8(a.lo[1]), bh, acc[1]);
  }
  acc[0] = MFMA16W(as_half8(a.hi[0]), bl, acc[0]);
  acc[1] = MFMA16W(as_half8(a.hi[1]), bl, acc[1]);
  acc[0] = MFMA16W(as_half8(a.hi[0]), bh, acc[0]);
  acc[1] = MFMA16W(as_half8(a.hi[1]), bh, acc[1]);
}

// FiLM epilogue of two values: accumulator registers 2 pc, 2 pc + 1 of row tile rt of n-block nbp -> (hi, lo) halves
// slots 4 rt + 2 pc + {0, 1} of k32-step nbp of the layer's output.  film_f / film_p already point at this lane group's
// first feature (16 (g >> 1) + 4 (g & 1)).
struct FilmQ { float2 f, p; };
template <int PF4>   // PF4: float offset of p' behind f'' in the FiLM buffer
__device__ __forceinline__ FilmQ epi_load(int nbp, int piece, const float* film) {
  const int rt = piece >> 1, pc = piece & 1;
  FilmQ q;
  q.f = *reinterpret_cast<const float2*>(film + 32 * nbp + 8 * rt + 2 * pc);
  q.p = *reinterpret_cast<const float2*>(film + PF4 + 32 * nbp + 8 * rt + 2 * pc);
  return q;
}
template <int KS, int SAVE>
__device__ __forceinline__ void epi_compute(const f32x4 (&acc)[2], int nbp, int piece, const FilmQ& q, half8 (&yh)[KS],
                                            half8 (&yl)[KS], TapeW<SAVE> tw, int tile_odd, TapeQ& tq) {
  const int rt = piece >> 1, pc = piece & 1;
  const float2 f = q.f, p = q.p;
  if (SAVE == 1 && pc == 0) {
    // register-dump position of this lane: float4 index (4 nb + 2 (g >> 1) + rt) * 64 + 32 (g & 1) + 16 (tile & 1) + n
    const int lo = opaque((int)(threadIdx.x & 63));
    const unsigned toff = (unsigned)(((2 * (lo >> 5)) * 64 + ((lo >> 4) & 1) * 32 + 16 * tile_odd + (lo & 15)) * 16);
    st_f4_nt(tw.base + (nbp * 4 + rt) * 1024, toff, acc[rt]);
  }
  // x = sin(2 pi theta); carried as hi = rn_f16(16 x), lo = rn_f16(16 x - hi).  Written as fmas on x so that each half is ONE
  // v_fma_mix{lo,hi}_f16 (fp32 fma, one rounding to f16; 16 x and 16 x - hi are exact in fp32, so the values are those of the
  // mul / convert / subtract / convert spelling): 8 VALU per two values instead of 14.
  const float th0 = __builtin_fmaf(f.x, acc[rt][2 * pc + 0], p.x), th1 = __builtin_fmaf(f.y, acc[rt][2 * pc + 1], p.y);
  if (SAVE == 2) {
    tq.q[piece] = __builtin_amdgcn_perm(phase_u16(th1), phase_u16(th0), 0x05040100u);      // [u16 of value 2 pc + 1 | u16 of value 2 pc]
    if (piece == 3) {
      const int lo = opaque((int)(threadIdx.x & 63));
      const u32x4 v = {tq.q[0], tq.q[1], tq.q[2], tq.q[3]};
      st_u4_nt(tw.base + nbp * 2048, (unsigned)(1024 * tile_odd + 16 * lo), v);
    }
  }
  const float s0 = sin2pi(th0);
  const float s1 = sin2pi(th1);
  const _Float16 h0 = (_Float16)__builtin_fmaf(s0, F16_ACT_SCALE, 0.f), h1 = (_Float16)__builtin_fmaf(s1, F16_ACT_SCALE, 0.f);
  half2 hp = {h0, h1}, lp = {(_Float16)__builtin_fmaf(s0, F16_ACT_SCALE, -(float)h0), (_Float16)__builtin_fmaf(s1, F16_ACT_SCALE, -(float)h1)};
  // pinned here: without a use in this block the compiler sinks the whole epilogue behind the stage (the outputs are only
  // consumed after it), keeping 8 n-blocks of accumulators + FiLM values alive -- and spilling them into the stream loop
  asm volatile("" : "+v"(hp), "+v"(lp));
  const int s = 4 * rt + 2 * pc;
  yh[nbp][s] = hp[0]; yh[nbp][s + 1] = hp[1];
  yl[nbp][s] = lp[0]; yl[nbp][s + 1] = lp[1];
}
template <int KS, int PF4, int SAVE>
__device__ __forceinline__ void epi_piece(const f32x4 (&acc)[2], int nbp, int piece, const float* film, half8 (&yh)[KS],
                                          half8 (&yl)[KS], TapeW<SAVE> tw, int tile_odd, TapeQ& tq) {
  epi_compute<KS, SAVE>(acc, nbp, piece, epi_load<PF4>(nbp, piece, film), yh, yl, tw, tile_odd, tq);
}
template <int KS, int PF4, int SAVE>
__device__ __forceinline__ void epi_all(const f32x4 (&acc)[2], int nbp, const float* film, half8 (&yh)[KS], half8 (&yl)[KS],
                                        TapeW<SAVE> tw, int tile_odd) {
  TapeQ tq;
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) epi_piece<KS, PF4, SAVE>(acc, nbp, pc, film, yh, yl, tw, tile_odd, tq);
}

// Chunk step i of a stage: barrier (chunk i + 1 visible), then the chunk's two k32-steps.  bop(sp, bh, bl) supplies the B
// operands of k32-step sp (false = padding); piece(spl) is the epilogue work issued behind k32-step spl.  A operands are read
// from the ring TWO k32-steps (one chunk step) ahead of use, at the top of a k32-step: a.c / a.n hold chunk i's two k32-steps
// on entry and chunk i + 1's on exit.  (Reads issued one MFMA ahead of use, which is where the compiler sinks them if let,
// leave the whole LDS latency in front of every k32-step -- in both waves of the SIMD at once, they run in phase.)
struct APipe { AK c, n; };
struct FilmQ2 { FilmQ q[4]; };
// LOR: the ring reads include the weight lo halves (false only when no stage of the kernel multiplies them: the reads run one chunk step
// ahead, across stage boundaries); LOM: this stage multiplies them.
// PH (round 6): -1 = the wave's DMA half is a run-time flag (ws.early) and the epilogue pieces sit behind k32-step 1 in every wave;
// 0 / 1 = compile-time copy of the step for waves 0-3 / 4-7 (no branch per k32-step; the shipped build).  FENERF_EXP_DEPHASE == 1
// (experiment, round 6): waves 4-7 take their epilogue pieces behind k32-step 0 -- the two waves of a SIMD (w, w + 4) run the same program
// in phase, both with their VALU-heavy epilogue in the same half of a chunk step and bare MFMAs in the other, and the idea was that one
// wave's epilogue VALU would issue under its partner's bare MFMA burst.  Measured: 2,640,000 instead of 2,444,000 cycles per launch
// (+ 8 %): the pieces then consume FiLM values read from LDS in the same k32-step, and the partner's burst does not hide that round trip.
template <bool LOR = true, bool LOM = true, int PH = -1, class BOP, class LOADQ, class PIECE>
__device__ __forceinline__ void chunk_step(f32x4 (&acc)[2], APipe& a, WStream& ws, int i, int sp0, BOP bop, LOADQ loadq, PIECE piece) {
  const bool early = PH < 0 ? ws.early : PH == 0;
  constexpr int PSPL = (PH == 1 && FENERF_EXP_DEPHASE == 1) ? 0 : 1;     // the k32-step behind whose MFMAs the epilogue pieces are issued
  ws_step(ws, i, early);
  FilmQ2 fq;
#pragma unroll
  for (int spl = 0; spl < 2; ++spl) {
    if (spl == 1) ws_issue_late(ws, i, early);
    // FiLM parameters of the chunk's epilogue pieces: read at the top of k32-step 0 (BEFORE the A operands: LDS returns in
    // order, a read behind them could only be waited for together with them), used behind k32-step 1's MFMAs -- a read
    // issued in the k32-step that consumes it put one LDS round trip (~190 cycles of s_waitcnt per chunk step and wave,
    // SQ_WAIT_ANY) into every chunk step.
    if (spl == 0) loadq(fq);
    const AK nn = ws_read<LOR>(ws, (i + 1) % NSLOT, spl);
    __builtin_amdgcn_sched_barrier(0);   // the reads stay at the top of the k32-step
    half8 bh, bl;
    if (bop(sp0 + spl, bh, bl)) kstep_mfma<LOR && LOM>(acc, a.c, bh, bl);
    if (spl == PSPL) piece(fq);
    a.c = a.n;
    a.n = nn;
    __builtin_amdgcn_sched_barrier(0);   // keep every k32-step's MFMAs / epilogue pieces where they are written
  }
}
// Stage-padding chunk (no weights in it): keep the DMA / barrier cadence, fetch the next chunk's k32-steps.
template <bool LO = true, int PH = -1>
__device__ __forceinline__ void chunk_skip(APipe& a, WStream& ws, int i) {
  const bool early = PH < 0 ? ws.early : PH == 0;
  ws_step(ws, i, early);
  ws_issue_late(ws, i, early);
  a.c = ws_read<LO>(ws, (i + 1) % NSLOT, 0);
  a.n = ws_read<LO>(ws, (i + 1) % NSLOT, 1);
  __builtin_amdgcn_sched_barrier(0);
}

// epilogue pieces of the previous n-block handled by chunk qc of a body of QB chunks: 4 pieces over min(QB, 4) chunks
template <int QB>
__device__ __forceinline__ void piece_range(int qc, int& p0, int& p1) {
  constexpr int QBE = QB < 4 ? QB : 4;
  if (qc >= QBE) { p0 = p1 = 0; return; }
  p0 = 4 * qc / QBE;
  p1 = 4 * (qc + 1) / QBE;
}

// FUSED = the whole hierarchical render of generators.py:479-519 in this one launch (fenerf_render_forward).  The unit of work is then not
// an oct but a RAY GROUP: the fewest rays whose coarse samples fill whole octs (G = 128 / gcd(N, 128) rays = N / gcd(N, 128) octs; 16 rays =
// 3 octs at N = 24).  A workgroup evaluates the coarse octs of a group exactly like the plain kernel (outputs to the workspace), then its
// 8 waves take the group's rays one wave per ray through composite_ray -- coarse weights + inverse-CDF resampling, fine depths to the
// workspace --, evaluates the fine octs from those depths, and the waves merge + composite the group's rays into the output pixels.  The
// coarse / fine rows and the fine depths are written and read back by the SAME workgroup a few microseconds apart (L2), ordered by
// workgroup-scope fences + barriers; the weight ring is untouched by the ray phases (chunks 0 .. D-1 of the next tile land and wait), which
// use the per-wave LDS block that parks colour layer 0's extra operands during a tile.
template <bool FUSED> struct FuseArgs {};
template <> struct FuseArgs<true> {
  int rays_per_group, octs_per_group;
  long long groups, total_rays;
  float* z_fine;            // [B*R][N]  written by the coarse ray phase, read by the fine tiles and the final ray phase
  float* out_fine;          // [P][C]    fine rows (P.out takes the coarse rows)
  CompositeParams coarse;   // sigma_only + u / z_fine: weights and resampling     (generators.py:486-499)
  CompositeParams final_;   // merge of fine | coarse + fancy_integration          (generators.py:508-519)
};
// timing experiments only (tools/exp/fused_ray_phase.sh): 2 = the ray phases as shipped, 1 = their fences + barriers without the rays,
// 0 = no ray phase at all (the tiles of a fused launch in fused order; wrong pixels)
#ifndef FENERF_EXP_RAY_PHASE
#define FENERF_EXP_RAY_PHASE 2
#endif
#ifndef FENERF_EXP_FUSED_MAXM
#define FENERF_EXP_FUSED_MAXM 128
#endif
constexpr int FUSED_MAXM = FENERF_EXP_FUSED_MAXM;   // samples per ray (2 N) the ray phases handle: their LDS scratch is the 4-KiB colour-layer-0 block

template <int H, bool GRID, int SAVE, bool FUSED = false, int TERMS2 = 0>
__global__ __launch_bounds__(512, 2) void siren16w_kernel(SirenParams P, int n_geo, int n_color, int n_lab, int C, FuseArgs<FUSED> F) {
  static_assert(!(FUSED && SAVE != 0), "the fused render is the no-grad path");
  static_assert(TERMS2 == 0 || (SAVE == 0 && !FUSED), "the reduced-precision forwards are plain no-grad evaluations");
  constexpr bool LOR = TERMS2 != 1;      // weight lo halves are fetched and read from the ring at all
  constexpr bool LOC = TERMS2 == 0;      // ... and multiplied in the colour layers and the rgb head (geometry trunk, label / sigma head: whenever they are read)
  constexpr int NB = H / 32, KS = H / 32;                       // 32-row n-blocks; k32-steps of an H-wide input
  constexpr int QB = (KS + 1) / 2;                              // chunks per square n-block body
  constexpr int C0_KS = KS + (GRID ? 1 : 0) + 1;                // colour layer 0: x | grid | dir
  constexpr int C0_QB = (2 * (2 * KS + (GRID ? 2 : 0) + 1) + CH - 1) / CH;
  constexpr int SQ_CHUNKS = pad_stage(NB * QB * CH) / CH;
  constexpr int C0_CHUNKS = pad_stage(NB * C0_QB * CH) / CH;
  constexpr int HEAD_CHUNKS = pad_stage(QB * CH) / CH;
  extern __shared__ __attribute__((aligned(16))) float4 smem[];

  if (P.clk && threadIdx.x == 0) {   // fenerf_siren_clock_probe: shader-clock and wall-clock stamps of this workgroup's first instruction
    P.clk[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
    P.clk[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int L = n_geo + n_color;
  const int stage_f4 = (16 * C + 3) / 4;
  // LDS: [ring NSLOT x 8 KiB][film: 8 waves x 2 buffers x (f'' | p')][layer-0 weights NB KiB][head consts 72 floats][stage 8 x 16*C]
  char* lds = reinterpret_cast<char*>(smem);
  char* ring = lds;
  constexpr int FILM_F = H * 4 < 1024 ? 1024 : H * 4;   // LDS-DMA moves whole KiBs
  constexpr int FILM_BYTES = 2 * FILM_F;
  char* film_base = lds + NSLOT * CH * 1024 + wave * (2 * FILM_BYTES);
  float* l0_lds = reinterpret_cast<float*>(lds + NSLOT * CH * 1024 + NWAVE * 2 * FILM_BYTES);
  float* cst = l0_lds + NB * 256;                       // [0,32) head scale, [32,64) head bias, [64,68) rgb scale, [68,72) rgb bias
  float* stage = cst + 80 + wave * stage_f4 * 4;
  // colour layer 0's extra B operands [grid hi | grid lo | dir hi | dir lo], 64 lanes x 16 B each, parked here from the tile
  // prologue until the layer needs them (11 registers less through the trunk)
  float4* ext_wave = reinterpret_cast<float4*>(cst + 80 + NWAVE * stage_f4 * 4) + wave * 256;

  // ---- tile-invariant constants -> LDS
  for (int i = threadIdx.x; i < NB * 256; i += 512) l0_lds[i] = P.stream[i];
  if (threadIdx.x < 32) {
    cst[threadIdx.x] = P.consts[CONST_FILM_BIAS + (size_t)2 * L * H + threadIdx.x];
    cst[32 + threadIdx.x] = P.consts[CONST_HEAD_BIAS + threadIdx.x];
  } else if (threadIdx.x < 36) {
    cst[64 + threadIdx.x - 32] = P.consts[CONST_FILM_BIAS + (size_t)2 * L * H + threadIdx.x];
    cst[68 + threadIdx.x - 32] = P.consts[CONST_RGB_BIAS + threadIdx.x - 32];
  }
  // FUSED: the two CompositeParams of the ray phases live in LDS (behind the colour-layer-0 blocks), not in ~90 SGPRs across the tile body
  CompositeParams* ray_par = reinterpret_cast<CompositeParams*>(reinterpret_cast<float4*>(cst + 80 + NWAVE * stage_f4 * 4) + NWAVE * 256);
  if constexpr (FUSED) {
    constexpr int NW = (int)(sizeof(CompositeParams) / 4);
    static_assert(sizeof(CompositeParams) % 4 == 0 && NW <= 256, "copied one dword per thread");
    const int t = threadIdx.x & 255;
    const unsigned* src = reinterpret_cast<const unsigned*>(threadIdx.x < 256 ? &F.coarse : &F.final_);
    if (t < NW) reinterpret_cast<unsigned*>(ray_par + (threadIdx.x >> 8))[t] = src[t];
  }
  WAIT_VMCNT(0);
  __syncthreads();

  // ---- the DMA's re-tiling permutation (header): this wave fetches operand (spl, rt, hl) = wave bits of every chunk
  WStream ws;
  {
    const int spl = wave >> 2, rt = (wave >> 1) & 1, hl = wave & 1;
    const int gi = n >> 2, r = n & 3;
    const int row = 16 * (gi >> 1) + 4 * (gi & 1) + 8 * rt + r;
    const int e_old = 2 * (2 * spl + (g >> 1)) + hl;
    ws.voff = e_old * 1024 + ((g & 1) * 32 + row) * 16;
  }
  const unsigned long long g_stream = reinterpret_cast<unsigned long long>(P.stream + P.ring_offset_floats);
  ws.g_next = g_stream;
  ws.ring_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring) + wave * 1024);
  ws.ring_lane = ring + lane * 16;
  ws.early = wave < NWAVE / 2;
  ws.skip = TERMS2 == 1 && (wave & 1);       // f16x2: the lo operands are never fetched

  // work split: octs of 16-point tiles (one tile per wave) -- FUSED: ray groups of octs_per_group octs --, XCD-contiguous ranges
  const long long ntiles = (P.P + 15) / 16;
  const long long nocts = (ntiles + NWAVE - 1) / NWAVE;
  int opg = 1;
  long long nunits = nocts;
  if constexpr (FUSED) { opg = F.octs_per_group; nunits = F.groups; }
  const int nblk = gridDim.x;
  const int nx = nblk < 8 ? nblk : 8;
  const int xcd = blockIdx.x % nx, bi = blockIdx.x / nx;
  const int blocks_in_x = nblk / nx + (xcd < nblk % nx ? 1 : 0);
  const long long o_begin = nunits * xcd / nx, o_end = nunits * (xcd + 1) / nx;

  if (P.raw_fg && o_begin + bi < o_end) {
    // FiLM pre-pass in the launch (fenerf_film.h): the images of this workgroup's octs, before the first LDS-DMA of the stream is
    // issued (ordinary loads and stores: nothing of theirs is left in flight behind the fence + barrier that ends the prologue)
    const long long p_first = (o_begin + bi) * opg * (NWAVE * 16);
    long long p_last = o_end * opg * (NWAVE * 16) - 1;
    if (p_last >= P.P) p_last = P.P - 1;
    film_prep_prologue(P, p_first / P.pts_per_image, p_last / P.pts_per_image, H, n_geo, n_color);
  }

  // ---- prime the shared stream: chunks 0..D-1 in flight, first k32-step of chunk 0 in registers
#pragma unroll
  for (int i = 0; i < DPF; ++i) ws_issue(ws, i);
  WAIT_VMCNT(DPF - 1);
  __builtin_amdgcn_s_barrier();
  LDS_FENCE();
  // Static issue priority for the second-dispatched half: between the two waves of a SIMD the arbiter prefers the older one (waves 0-3),
  // so waves 4-7 lose VALU / LDS issue slots in every phase and arrive last at every barrier.  One s_setprio for the whole kernel (no
  // per-phase flips: priority around the MFMA bursts or around the epilogue costs 1.7-3 % more cycles): 2,715,000 -> 2,611,000 shader
  // cycles per launch (- 3.8 %), of which the power manager gives back half (2.03 -> 1.99 GHz): 1.367 -> 1.342 ms on the same box.
  if (wave >= NWAVE / 2) __builtin_amdgcn_s_setprio(1);
  APipe a_cur;
  a_cur.c = ws_read<LOR>(ws, 0, 0);
  a_cur.n = ws_read<LOR>(ws, 0, 1);


  for (long long unit = o_begin + bi; unit < o_end; unit += blocks_in_x) {
  const int nsub = FUSED ? 2 * opg : 1;       // FUSED: the group's coarse octs, then its fine octs
#pragma unroll 1
  for (int sub = 0; sub < nsub; ++sub) {
    const bool fine_pass = FUSED && sub >= opg;
    const long long oct = FUSED ? unit * opg + (fine_pass ? sub - opg : sub) : unit;
    const float* z_src = P.z;
    float* out_dst = P.out;
    if constexpr (FUSED) {
      if (fine_pass) { z_src = F.z_fine; out_dst = F.out_fine; }
    }
    // the previous tile ended by issuing the replicated head chunks nchunk..nchunk+D-1 (== this tile's chunks 0..D-1)
    ws.g_next = g_stream + (unsigned long long)DPF * (CH * 1024);
    const long long tile = oct * NWAVE + wave;
    // ---------------- this lane's point ----------------
    long long pt = tile * 16 + n;
    if (pt >= P.P) pt = P.P - 1;
    const long long img = __builtin_amdgcn_readfirstlane((int)(pt / P.pts_per_image));   // tiles do not straddle images (launcher)
    float px, py, pz, dx, dy, dz;
    if (P.points) {
      px = P.points[pt * 3 + 0]; py = P.points[pt * 3 + 1]; pz = P.points[pt * 3 + 2];
      if (P.pdirs) { dx = P.pdirs[pt * 3 + 0]; dy = P.pdirs[pt * 3 + 1]; dz = P.pdirs[pt * 3 + 2]; }
      else { dx = 0.f; dy = 0.f; dz = -1.f; }
    } else {
      const long long ray = pt / P.n_per_ray;
      const float zz = z_src[pt];
      const float ox = P.origins[ray * 3 + 0], oy = P.origins[ray * 3 + 1], oz = P.origins[ray * 3 + 2];
      dx = P.dirs[ray * 3 + 0]; dy = P.dirs[ray * 3 + 1]; dz = P.dirs[ray * 3 + 2];
      px = __fadd_rn(ox, __fmul_rn(dx, zz)); py = __fadd_rn(oy, __fmul_rn(dy, zz)); pz = __fadd_rn(oz, __fmul_rn(dz, zz));
      if (P.lock_view) { dx = 0.f; dy = 0.f; dz = -1.f; }
    }
    const float qx = px * P.box_scale, qy = py * P.box_scale, qz = pz * P.box_scale;
    // forward-save: this tile's half of the (tile32) tape block of FiLM layer `layer`; phantom tiles of the last oct (clamped
    // points) dump into the slack fenerf_siren_tape_floats keeps behind the last tile
    constexpr int TL = SAVE == 2 ? H * 64 : H * 128;       // bytes of a (tile32, layer) block of the tape
    const int tile_odd = (int)(tile & 1);
    const char* tape_tile = SAVE ? uniform_ptr(reinterpret_cast<const char*>(P.tape) + (size_t)(tile >> 1) * L * TL) : nullptr;
    auto tape_of = [&](int layer) {
      TapeW<SAVE> t{SAVE ? uniform_ptr(tape_tile + (size_t)layer * TL) : nullptr};
      if (SAVE) asm volatile("s_nop 4" ::: "memory");   // SGPRs written by v_readfirstlane feed a vector-memory address: 5 wait states
      return t;
    };

    // ---------------- FiLM parameters of layers 0 and 1 -> LDS (DMA), grid gather, then one drain ----------------
    const float* fp_img = P.fp + (size_t)img * L * H;
    const float* pp_img = P.pp + (size_t)img * L * H;
    auto film_issue = [&](int layer) {   // f'' (H floats) then p' (H floats) of `layer` into buffer layer & 1
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(film_base) + (layer & 1) * FILM_BYTES);
      const float* gf = fp_img + (size_t)layer * H;
      const float* gp = pp_img + (size_t)layer * H;
      const unsigned vo = (unsigned)opaque(lane) * 16;
      for (int off = 0; off < H * 4; off += 1024) {   // H=256: one KiB each; smaller H: lanes beyond H/4 read in-bounds pad
        glds_1k_s(reinterpret_cast<const char*>(gf) + off, vo, dst + off);
        glds_1k_s(reinterpret_cast<const char*>(gp) + off, vo, dst + FILM_F + off);
      }
    };
    film_issue(0);
    if (L > 1) film_issue(1);

    // grid features: lane (n, g) blends channels 16 (g & 1) + 8 (g >> 1) .. + 7 of its point's 8 corners (the k slots the
    // packer gave lane group g in the grid k32-step)
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = 0.f;
    if (GRID) {
      const int ch0 = 16 * (g & 1) + 8 * (g >> 1);
      const float ix = ((qx + 1.f) / 2.f) * (float)(P.gw - 1);
      const float iy = ((qy + 1.f) / 2.f) * (float)(P.gh - 1);
      const float iz = ((qz + 1.f) / 2.f) * (float)(P.gd - 1);
      const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int cz = c >> 2, cy = (c >> 1) & 1, cx = c & 1;
        const float xi = x0 + cx, yi = y0 + cy, zi = z0 + cz;
        const float wx = cx ? (ix - x0) : (x0 + 1.f - ix);
        const float wy = cy ? (iy - y0) : (y0 + 1.f - iy);
        const float wz = cz ? (iz - z0) : (z0 + 1.f - iz);
        const float wgt = wx * wy * wz;
        const bool ok = xi >= 0.f && xi <= (float)(P.gw - 1) && yi >= 0.f && yi <= (float)(P.gh - 1) && zi >= 0.f &&
                        zi <= (float)(P.gd - 1);
        if (ok) {
          const long long vox = ((long long)(int)zi * P.gh + (int)yi) * P.gw + (int)xi;
          const float4* gp = reinterpret_cast<const float4*>(P.grid + vox * 32 + ch0);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float4 v = gp[q];
            e[4 * q + 0] += v.x * wgt; e[4 * q + 1] += v.y * wgt; e[4 * q + 2] += v.z * wgt; e[4 * q + 3] += v.w * wgt;
          }
        }
      }
    }
    if (SAVE && GRID) {   // the sampled grid features, [P][32] (a clamped lane rewrites the last point's row with the same values)
      float4* ep = reinterpret_cast<float4*>(P.tape_e + pt * 32 + 16 * (g & 1) + 8 * (g >> 1));
      ep[0] = make_float4(e[0], e[1], e[2], e[3]);
      ep[1] = make_float4(e[4], e[5], e[6], e[7]);
    }
    {
      half8 eh, el, dh, dl;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = e[t] * F16_ACT_SCALE;
        const _Float16 hh = (_Float16)v;
        eh[t] = hh; el[t] = (_Float16)(v - (float)hh);
      }
      // the dir k16-step is lane group 0's (slots 0..2); groups 1-3 multiply zero padding
      const float d3[3] = {dx * F16_ACT_SCALE, dy * F16_ACT_SCALE, dz * F16_ACT_SCALE};
#pragma unroll
      for (int t = 0; t < 8; ++t) { dh[t] = (_Float16)0.f; dl[t] = (_Float16)0.f; }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float v = g == 0 ? d3[t] : 0.f;
        const _Float16 hh = (_Float16)v;
        dh[t] = hh; dl[t] = (_Float16)(v - (float)hh);
      }
      float4* ext = ext_wave + lane;
      ext[0] = __builtin_bit_cast(float4, eh); ext[64] = __builtin_bit_cast(float4, el);
      ext[128] = __builtin_bit_cast(float4, dh); ext[192] = __builtin_bit_cast(float4, dl);
    }
    WAIT_VMCNT(0);      // film 0/1 landed (own buffer, own reads: no barrier needed), point / grid loads done; once per tile
    LDS_FENCE();

    // f'' of FiLM layer `layer` at this lane group's first feature (p' follows FILM_F bytes behind); derived on the spot
    auto film_lane = [&](int layer) -> const float* {
      const int gq = opaque(lane) >> 4;
      return reinterpret_cast<const float*>(film_base + (layer & 1) * FILM_BYTES) + 16 * (gq >> 1) + 4 * (gq & 1);
    };

    half8 xh[KS], xl[KS];
    // ---------------- layer 0: 3 -> H on the exact fp32 MFMA (16x16x4: k = x, y, z, 0) ----------------
    {
      const float b = g == 0 ? qx : (g == 1 ? qy : (g == 2 ? qz : 0.f));
      const int gi = n >> 2, r = n & 3;
      const float* wl = l0_lds + ((g & 1) * 32 + 16 * (gi >> 1) + 4 * (gi & 1) + r) * 4 + (g >> 1);
      const TapeW<SAVE> tw0 = tape_of(0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          acc[rt] = MFMA32W(wl[nb * 256 + 8 * rt * 4], b, z4);
        }
        epi_all<KS, FILM_F / 4, SAVE>(acc, nb, film_lane(0), xh, xl, tw0, tile_odd);
      }
    }
    // ---------------- FiLM layers 1 .. L-1 ----------------
    auto film_layer = [&](auto ph_c, int l) {
      constexpr int PH = decltype(ph_c)::value;
      const float* ff = film_lane(l);
      const TapeW<SAVE> tw = tape_of(l);
      half8 yh[KS], yl[KS];
      if (l + 1 < L) film_issue(l + 1);
      if (SQ_CHUNKS < DPF + 2) WAIT_VMCNT(0);
      if (l == n_geo) {
        // ---------------- colour layer 0: [x | grid feats | dir] -> H, then the label/sigma head on the same x -------
        const float4* ext = ext_wave + opaque(lane);
        auto bop0 = [&](int sp, half8& bh, half8& bl) -> bool {
          if (sp < KS) { bh = xh[sp]; bl = xl[sp]; return true; }
          if (GRID && sp == KS) { bh = as_half8(ext[0]); bl = as_half8(ext[64]); return true; }
          if (sp == C0_KS - 1) { bh = as_half8(ext[128]); bl = as_half8(ext[192]); return true; }
          return false;
        };
        f32x4 acc_prev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        TapeQ tq;      // SAVE = 2: the packed phases of n-block nb - 1 between its first and last epilogue piece
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int qc = 0; qc < C0_QB; ++qc) {
            chunk_step<LOR, LOC, PH>(acc, a_cur, ws, nb * C0_QB + qc, 2 * qc, bop0, [&](FilmQ2& fq) {
              if (nb > 0) {
                int p0, p1;
                piece_range<C0_QB>(qc, p0, p1);
#pragma unroll
                for (int pc = 0; pc < 4; ++pc)
                  if (pc >= p0 && pc < p1) fq.q[pc] = epi_load<FILM_F / 4>(nb - 1, pc, ff);
              }
            }, [&](const FilmQ2& fq) {
              if (nb > 0) {
                int p0, p1;
                piece_range<C0_QB>(qc, p0, p1);
#pragma unroll
                for (int pc = 0; pc < 4; ++pc)
                  if (pc >= p0 && pc < p1) epi_compute<KS, SAVE>(acc_prev, nb - 1, pc, fq.q[pc], yh, yl, tw, tile_odd, tq);
              }
            });
          }
          acc_prev[0] = acc[0]; acc_prev[1] = acc[1];
        }
#pragma unroll
        for (int i = NB * C0_QB; i < C0_CHUNKS; ++i) chunk_skip<LOR, PH>(a_cur, ws, i);
        epi_all<KS, FILM_F / 4, SAVE>(acc_prev, NB - 1, ff, yh, yl, tw, tile_odd);
        // head on x (the trunk output), before x is overwritten with the colour-layer-0 activations
        {
          f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          auto bop = [&](int sp, half8& bh, half8& bl) -> bool {
            if (sp < KS) { bh = xh[sp]; bl = xl[sp]; return true; }
            return false;
          };
#pragma unroll
          // (the label / sigma head keeps every term the kernel reads: with TERMS2 = 2 sigma and the labels are those of the default, bit for bit)
          for (int qc = 0; qc < QB; ++qc) chunk_step<LOR, true, PH>(acc, a_cur, ws, qc, 2 * qc, bop, [](FilmQ2&) {}, [](const FilmQ2&) {});
#pragma unroll
          for (int i = QB; i < HEAD_CHUNKS; ++i) chunk_skip<LOR, PH>(a_cur, ws, i);
          const int lane_o = opaque(lane);
          const int n_o = lane_o & 15, f_o = 16 * (lane_o >> 5) + 4 * ((lane_o >> 4) & 1);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = f_o + 8 * rt + r;
              if (row <= n_lab) {
                const int ch = row < n_lab ? row : C - 1;
                stage[n_o * C + ch] = acc[rt][r] * cst[row] + cst[32 + row];
              }
            }
          }
        }
      } else {
        auto square_stage = [&](auto lom_c) {
        constexpr bool LOM = decltype(lom_c)::value;
        auto bop = [&](int sp, half8& bh, half8& bl) -> bool {
          if (sp < KS) { bh = xh[sp]; bl = xl[sp]; return true; }
          return false;
        };
        f32x4 acc_prev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        TapeQ tq;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int qc = 0; qc < QB; ++qc) {
            chunk_step<LOR, LOM, PH>(acc, a_cur, ws, nb * QB + qc, 2 * qc, bop, [&](FilmQ2& fq) {
              if (nb > 0) {
                int p0, p1;
                piece_range<QB>(qc, p0, p1);
#pragma unroll
                for (int pc = 0; pc < 4; ++pc)
                  if (pc >= p0 && pc < p1) fq.q[pc] = epi_load<FILM_F / 4>(nb - 1, pc, ff);
              }
            }, [&](const FilmQ2& fq) {
              if (nb > 0) {
                int p0, p1;
                piece_range<QB>(qc, p0, p1);
#pragma unroll
                for (int pc = 0; pc < 4; ++pc)
                  if (pc >= p0 && pc < p1) epi_compute<KS, SAVE>(acc_prev, nb - 1, pc, fq.q[pc], yh, yl, tw, tile_odd, tq);
              }
            });
          }
          acc_prev[0] = acc[0]; acc_prev[1] = acc[1];
        }
#pragma unroll
        for (int i = NB * QB; i < SQ_CHUNKS; ++i) chunk_skip<LOR, PH>(a_cur, ws, i);
        epi_all<KS, FILM_F / 4, SAVE>(acc_prev, NB - 1, ff, yh, yl, tw, tile_odd);
        };
        // TERMS2 = 2: three terms per product through the geometry trunk, two in the colour layers (one instantiation of the stage each; the
        // branch sits between stages, not inside an MFMA stream)
        if constexpr (TERMS2 == 2) {
          if (l < n_geo) square_stage(std::true_type{}); else square_stage(std::false_type{});
        } else {
          square_stage(std::integral_constant<bool, LOR>{});
        }
      }
#pragma unroll
      for (int k = 0; k < KS; ++k) { xh[k] = yh[k]; xl[k] = yl[k]; }
    };
#pragma unroll 1
    for (int l = 1; l < L; ++l) {
      // One compile-time copy of the layer per wave half (round 6): the DMA position of a wave (top of a chunk step / half a step later)
      // was a run-time flag tested in every k32-step -- two scalar branches per k32-step, one of them taken.  With the branch between
      // layers instead: 2,599,000 -> 2,444,000 shader cycles per launch (- 6.0 %), of which the power manager keeps two thirds
      // (1.95 -> 1.87 GHz): 5.75 -> 5.86 M rays/s on the same box (profiles/r06_forward_wave_half_copies.md).
#if FENERF_WAVE_HALF_COPIES
      if (ws.early) film_layer(std::integral_constant<int, 0>{}, l); else film_layer(std::integral_constant<int, 1>{}, l);
#else
      film_layer(std::integral_constant<int, -1>{}, l);
#endif
    }
    // ---------------- rgb head + sigmoid (rows 0..2 = row tile 0, lane group 0) ----------------
    {
      constexpr int PH = -1;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      auto bop = [&](int sp, half8& bh, half8& bl) -> bool {
        if (sp < KS) { bh = xh[sp]; bl = xl[sp]; return true; }
        return false;
      };
#pragma unroll
      for (int qc = 0; qc < QB; ++qc) chunk_step<LOR, LOC, PH>(acc, a_cur, ws, qc, 2 * qc, bop, [](FilmQ2&) {}, [](const FilmQ2&) {});
#pragma unroll
      for (int i = QB; i < HEAD_CHUNKS; ++i) chunk_skip<LOR, PH>(a_cur, ws, i);
      const int lane_o = opaque(lane);
      const int n_o = lane_o & 15;
      if ((lane_o >> 4) == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float v = acc[0][r] * cst[64 + r] + cst[68 + r];
          stage[n_o * C + (C - 4) + r] = 1.f / (1.f + __expf(-v));
        }
      }
    }
    // ---------------- coalesced write-out of the tile's [16][C] block ----------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const long long base = tile * 16 * C;
      const long long limit = P.P * C;
      for (int i = opaque(lane); i < 16 * C; i += 64)
        if (base + i < limit) out_dst[base + i] = stage[i];
    }
    WAIT_VMCNT(0);   // stores may retire out of order with the DMA loads: keep them out of the counted waits
    __builtin_amdgcn_wave_barrier();
    if constexpr (FUSED) {
      if (FENERF_EXP_RAY_PHASE > 0 && (sub == opg - 1 || sub == nsub - 1)) {
        // ---------------- ray phase: every wave has stored (and waited for) its rows of this pass; one wave per ray ----------------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float* sc = reinterpret_cast<float*>(ext_wave);
        float* s_z = sc;
        float* s_zs = sc + FUSED_MAXM;
        int* s_ord = reinterpret_cast<int*>(sc + 2 * FUSED_MAXM + 4);
        float* s_w = sc + 3 * FUSED_MAXM + 4;
        const int lane_r = opaque(lane);
        const long long ray0 = unit * F.rays_per_group;
        for (int r = wave; r < F.rays_per_group; r += NWAVE) {
          const long long ray = ray0 + r;
          if (FENERF_EXP_RAY_PHASE > 1 && ray < F.total_rays) {
#pragma unroll 1
            for (int rep = 0; rep < (FENERF_EXP_RAY_PHASE > 2 ? 2 : 1); ++rep) {   // > 2: every ray twice (marginal cost of a ray)
              if (!fine_pass) composite_ray<false, FUSED_MAXM>(ray_par[0], ray, lane_r, s_z, s_zs, s_ord, s_w);
              else composite_ray<true, FUSED_MAXM>(ray_par[1], ray, lane_r, s_z, s_zs, s_ord, s_w);
              __builtin_amdgcn_wave_barrier();
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (!fine_pass) {   // the fine depths of the group, for every wave's fine tiles
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        WAIT_VMCNT(0);      // nothing of the ray phase is in flight when the counted waits of the stream loop resume
      }
    }
  }
  }
  WAIT_VMCNT(0);     // no LDS-DMA may land after the workgroup has released its LDS
  __builtin_amdgcn_s_barrier();
  if (P.clk && threadIdx.x == 0) {   // ... and of its last
    P.clk[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
    P.clk[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

static int hip_fail16w(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

static size_t lds_bytes_16w(const FenerfModel* m, int H) {
  const int stage_f4 = (16 * m->C + 3) / 4;
  const size_t film_f = H * 4 < 1024 ? 1024 : H * 4;
  return (size_t)NSLOT * CH * 1024 + (size_t)NWAVE * 2 * (2 * film_f) + (size_t)(H / 32) * 1024 + 80 * 4 +
         (size_t)NWAVE * stage_f4 * 16 + (size_t)NWAVE * 4096;   // ring + FiLM buffers + layer-0 weights + head consts +
                                                                   // output staging + colour-layer-0 operands
}

// fenerf_render_forward in one launch (FUSED): `blocks` workgroups over F.groups ray groups
template <int H, bool GRID>
static int launch_fused_t(const FenerfModel* m, const SirenParams& p, const FuseArgs<true>& F, int blocks, void* stream) {
  const size_t lds = lds_bytes_16w(m, H) + (2 * sizeof(CompositeParams) + 255) / 256 * 256;
  auto kfn = siren16w_kernel<H, GRID, 0, true>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C, F);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail16w(e, "fused render launch");
}

template <int H, bool GRID, int SAVE, int TERMS2 = 0>
static int launch_t(const FenerfModel* m, const SirenParams& p, void* stream) {
  const size_t lds = lds_bytes_16w(m, H);
  auto kfn = siren16w_kernel<H, GRID, SAVE, false, TERMS2>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  const long long ntiles = (p.P + 15) / 16;
  long long blocks = (ntiles + NWAVE - 1) / NWAVE;
  if (blocks > m->num_cus) blocks = m->num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C, FuseArgs<false>{});
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail16w(e, "siren16w launch");
}

}  // namespace w16

// One launch over points whose tiles do not straddle images.
static int launch_siren16w_one(const FenerfModel* m, const SirenParams& q, void* stream) {
  const bool g = m->grid_ch != 0;
  if (q.tape && q.tape_format == FENERF_TAPE_U16) {   // forward-save with the 16-bit tape (round 5)
    switch (m->H) {
      case 32: return g ? w16::launch_t<32, true, 2>(m, q, stream) : w16::launch_t<32, false, 2>(m, q, stream);
      case 64: return g ? w16::launch_t<64, true, 2>(m, q, stream) : w16::launch_t<64, false, 2>(m, q, stream);
      case 96: return g ? w16::launch_t<96, true, 2>(m, q, stream) : w16::launch_t<96, false, 2>(m, q, stream);
      case 128: return g ? w16::launch_t<128, true, 2>(m, q, stream) : w16::launch_t<128, false, 2>(m, q, stream);
      case 192: return g ? w16::launch_t<192, true, 2>(m, q, stream) : w16::launch_t<192, false, 2>(m, q, stream);
      case 256: return g ? w16::launch_t<256, true, 2>(m, q, stream) : w16::launch_t<256, false, 2>(m, q, stream);
    }
  }
  if (q.tape) {   // forward-save: the same kernel also dumps the tape (and the sampled grid features)
    switch (m->H) {
      case 32: return g ? w16::launch_t<32, true, 1>(m, q, stream) : w16::launch_t<32, false, 1>(m, q, stream);
      case 64: return g ? w16::launch_t<64, true, 1>(m, q, stream) : w16::launch_t<64, false, 1>(m, q, stream);
      case 96: return g ? w16::launch_t<96, true, 1>(m, q, stream) : w16::launch_t<96, false, 1>(m, q, stream);
      case 128: return g ? w16::launch_t<128, true, 1>(m, q, stream) : w16::launch_t<128, false, 1>(m, q, stream);
      case 192: return g ? w16::launch_t<192, true, 1>(m, q, stream) : w16::launch_t<192, false, 1>(m, q, stream);
      case 256: return g ? w16::launch_t<256, true, 1>(m, q, stream) : w16::launch_t<256, false, 1>(m, q, stream);
    }
  }
  if (m->forward_mode == FENERF_FORWARD_F16X2) {
    switch (m->H) {
      case 32: return g ? w16::launch_t<32, true, 0, 1>(m, q, stream) : w16::launch_t<32, false, 0, 1>(m, q, stream);
      case 64: return g ? w16::launch_t<64, true, 0, 1>(m, q, stream) : w16::launch_t<64, false, 0, 1>(m, q, stream);
      case 96: return g ? w16::launch_t<96, true, 0, 1>(m, q, stream) : w16::launch_t<96, false, 0, 1>(m, q, stream);
      case 128: return g ? w16::launch_t<128, true, 0, 1>(m, q, stream) : w16::launch_t<128, false, 0, 1>(m, q, stream);
      case 192: return g ? w16::launch_t<192, true, 0, 1>(m, q, stream) : w16::launch_t<192, false, 0, 1>(m, q, stream);
      case 256: return g ? w16::launch_t<256, true, 0, 1>(m, q, stream) : w16::launch_t<256, false, 0, 1>(m, q, stream);
    }
  }
  if (m->forward_mode == FENERF_FORWARD_F16X3_COLOR_X2) {
    switch (m->H) {
      case 32: return g ? w16::launch_t<32, true, 0, 2>(m, q, stream) : w16::launch_t<32, false, 0, 2>(m, q, stream);
      case 64: return g ? w16::launch_t<64, true, 0, 2>(m, q, stream) : w16::launch_t<64, false, 0, 2>(m, q, stream);
      case 96: return g ? w16::launch_t<96, true, 0, 2>(m, q, stream) : w16::launch_t<96, false, 0, 2>(m, q, stream);
      case 128: return g ? w16::launch_t<128, true, 0, 2>(m, q, stream) : w16::launch_t<128, false, 0, 2>(m, q, stream);
      case 192: return g ? w16::launch_t<192, true, 0, 2>(m, q, stream) : w16::launch_t<192, false, 0, 2>(m, q, stream);
      case 256: return g ? w16::launch_t<256, true, 0, 2>(m, q, stream) : w16::launch_t<256, false, 0, 2>(m, q, stream);
    }
  }
  switch (m->H) {
    case 32: return g ? w16::launch_t<32, true, 0>(m, q, stream) : w16::launch_t<32, false, 0>(m, q, stream);
    case 64: return g ? w16::launch_t<64, true, 0>(m, q, stream) : w16::launch_t<64, false, 0>(m, q, stream);
    case 96: return g ? w16::launch_t<96, true, 0>(m, q, stream) : w16::launch_t<96, false, 0>(m, q, stream);
    case 128: return g ? w16::launch_t<128, true, 0>(m, q, stream) : w16::launch_t<128, false, 0>(m, q, stream);
    case 192: return g ? w16::launch_t<192, true, 0>(m, q, stream) : w16::launch_t<192, false, 0>(m, q, stream);
    case 256: return g ? w16::launch_t<256, true, 0>(m, q, stream) : w16::launch_t<256, false, 0>(m, q, stream);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

// Geometry of the one-launch render: G rays per group, octs per group, groups, workgroups; false when this (B, R, N) cannot run fused.
// `balanced_only`: refuse when the coarser unit of work would lengthen the critical path (whole groups per workgroup vs whole octs).
bool fused_render_plan(const FenerfModel* m, long long B, long long R, int N, bool balanced_only, FusedRenderPlan* plan) {
  if (m->precision != FENERF_PREC_F16X3 || N < 3 || 2 * N > w16::FUSED_MAXM) return false;
  if (m->forward_mode != FENERF_FORWARD_F16X3) return false;   // the fused kernel is instantiated for the full f16x3 arithmetic only: an opt-in
                                                               // reduced-precision model takes the four-launch route, which honours its mode
  int g = N, b = 128;
  while (b) { const int t = g % b; g = b; b = t; }          // gcd(N, 128)
  const int G = 128 / g, opg = N / g;
  if (R % G != 0) return false;                              // groups (hence tiles) do not straddle images
  const long long groups = B * R / G, nocts = groups * opg;
  const long long cus = launch_cus(m);
  const long long blocks_f = groups < cus ? groups : cus, blocks_p = nocts < cus ? nocts : cus;
  if (balanced_only && (groups + blocks_f - 1) / blocks_f * opg > (nocts + blocks_p - 1) / blocks_p) return false;
  plan->rays_per_group = G; plan->octs_per_group = opg; plan->groups = groups; plan->blocks = (int)blocks_f;
  return true;
}

int launch_render16w_fused(const FenerfModel* m, const SirenParams& p, const FusedRenderPlan& plan, float* z_fine, float* out_fine,
                           const CompositeParams& coarse, const CompositeParams& final_, void* stream) {
  w16::FuseArgs<true> F;
  F.rays_per_group = plan.rays_per_group; F.octs_per_group = plan.octs_per_group; F.groups = plan.groups;
  F.total_rays = coarse.BR;
  F.z_fine = z_fine; F.out_fine = out_fine; F.coarse = coarse; F.final_ = final_;
  const bool g = m->grid_ch != 0;
  switch (m->H) {
    case 32: return g ? w16::launch_fused_t<32, true>(m, p, F, plan.blocks, stream) : w16::launch_fused_t<32, false>(m, p, F, plan.blocks, stream);
    case 64: return g ? w16::launch_fused_t<64, true>(m, p, F, plan.blocks, stream) : w16::launch_fused_t<64, false>(m, p, F, plan.blocks, stream);
    case 96: return g ? w16::launch_fused_t<96, true>(m, p, F, plan.blocks, stream) : w16::launch_fused_t<96, false>(m, p, F, plan.blocks, stream);
    case 128: return g ? w16::launch_fused_t<128, true>(m, p, F, plan.blocks, stream) : w16::launch_fused_t<128, false>(m, p, F, plan.blocks, stream);
    case 192: return g ? w16::launch_fused_t<192, true>(m, p, F, plan.blocks, stream) : w16::launch_fused_t<192, false>(m, p, F, plan.blocks, stream);
    case 256: return g ? w16::launch_fused_t<256, true>(m, p, F, plan.blocks, stream) : w16::launch_fused_t<256, false>(m, p, F, plan.blocks, stream);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

// FENERF_PREC_F16X3 models, forward and forward-save.  Tiles must not straddle images (FiLM parameters are fetched per wave, the
// tape is laid out in 32-point tiles per image): when points-per-image is not a multiple of 32, launch image by image (each
// launch ends in a ragged tile).
int launch_siren16w(const FenerfModel* m, const SirenParams& p, void* stream) {
  if (p.P <= 0) return FENERF_OK;
  if (p.pts_per_image % 32 == 0 || p.P == p.pts_per_image) return launch_siren16w_one(m, p, stream);
  const long long nimg = p.P / p.pts_per_image;
  const int L = m->L, H = m->H;
  for (long long b = 0; b < nimg; ++b) {
    SirenParams q = p;
    q.P = p.pts_per_image;
    q.fp = p.fp + (size_t)b * L * H;
    q.pp = p.pp + (size_t)b * L * H;
    if (p.raw_fg) {   // FiLM pre-pass in the launch: this image's raw blocks
      q.raw_fg = p.raw_fg + (size_t)b * m->n_geo * H; q.raw_pg = p.raw_pg + (size_t)b * m->n_geo * H;
      q.raw_fa = p.raw_fa + (size_t)b * m->n_color * H; q.raw_pa = p.raw_pa + (size_t)b * m->n_color * H;
      q.n_images = 1;
    }
    q.out = p.out + (size_t)b * p.pts_per_image * m->C;
    if (p.points) {
      q.points = p.points + (size_t)b * p.pts_per_image * 3;
      if (p.pdirs) q.pdirs = p.pdirs + (size_t)b * p.pts_per_image * 3;
    } else {
      const long long rays = p.pts_per_image / p.n_per_ray;
      q.origins = p.origins + (size_t)b * rays * 3;
      q.dirs = p.dirs + (size_t)b * rays * 3;
      q.z = p.z + (size_t)b * p.pts_per_image;
    }
    int rc = launch_siren16w_one(m, q, stream);
    if (rc) return rc;
  }
  return FENERF_OK;
}

}  // namespace fenerf
