// FiLM-SIREN radiance field, f16x3 mode, workgroup-shared weight stream (gfx950 / MI355X).
//
// Arithmetic and per-wave dataflow are those of fenerf_siren_f16.hip (one wave = 32 points through the whole network
// on v_mfma_f32_32x32x16_f16 with error-compensated hi/lo fp16 operands; activations never leave their lane).
// What changes is how the weights reach the matrix pipe.  At fp16 MFMA speed each wave consumes its A operands at
// 21 B/clk; four waves streaming privately from L2 would need 85 B/clk/CU -- more than the L2 fabric delivers
// (~56 B/clk/CU) and the measured limiter of fenerf_siren_f16.hip (30 % of wave time in s_waitcnt vmcnt).  Here the
// four waves of a workgroup walk the weight stream in lockstep and share it through LDS:
//
//   * the stream is cut into 8-KiB chunks (8 entries = 4 k-steps x [hi, lo]); each wave DMAs a quarter of every chunk
//     straight into an LDS ring (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs), 8 chunks ahead;
//   * per chunk: counted `s_waitcnt vmcnt` (never 0) -> one raw s_barrier -> every wave ds_read_b128's the chunk's
//     A operands for the NEXT step while the MFMAs of the CURRENT step run from registers (double-buffered in VGPRs),
//     so neither the L2 latency nor the LDS latency nor the barrier skew ever sits in front of the matrix pipe;
//   * ring of 10 slots: a slot is re-filled two barriers after its last reader issued its reads (WAR-safe);
//   * the stream is cyclic over tiles (every tile uses the same 2.7 MB), so the prefetch never restarts;
//   * per-layer FiLM parameters arrive by the same LDS-DMA path (2 x 1 KiB per layer and wave) and are read back
//     with broadcast ds_reads -- no ordinary VMEM load sits in the main loop to make hipcc drain the DMA queue;
//   * the FiLM epilogue of n-block nb-1 (v_sin_f32 + hi/lo split, ~7 VALU per value) is software-pipelined into the
//     MFMA stream of n-block nb (n-block loop fully unrolled).  Its outputs cannot overwrite the inputs (every n-block
//     reads all of them): the first half of the layer's outputs waits in a 16-KiB per-wave LDS slab, the second half
//     in 64 VGPRs -- x (128) + y (128) in registers is what made hipcc spill the B operands to scratch.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_nt.h"

namespace fenerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int CH = FENERF_CH;        // entries (KiB) per chunk
constexpr int DPF = FENERF_DPF;      // chunks in flight ahead of the chunk being consumed (48 KiB per CU)
constexpr int NSLOT = FENERF_NSLOT;  // LDS ring slots; every stage is a whole number of ring revolutions (packer), so every
                                     // stage starts at slot 0 and slot indices / LDS offsets are compile-time constants
static_assert(CH == FENERF_PF, "bodies are padded to whole chunks by the packer");
static_assert(NSLOT >= DPF + 2, "a slot is refilled two barriers after its last reader issued its reads");

__device__ __forceinline__ half8 as_half8(const float4& v) { return __builtin_bit_cast(half8, v); }

// LDS-DMA of one KiB: lane i's 16 bytes at g + 16 i  ->  lds + 16 i   (lds wave-uniform, passed in M0).
// Inline asm on purpose: with the builtin, hipcc tracks the DMA as a pending LDS write and puts `s_waitcnt vmcnt(0)`
// in front of the next ds_read of the ring -- draining the 16-deep DMA queue every chunk.  Hidden in asm, the only
// waits are the counted ones below (cdna_hip_programming.md 5.7).  M0 is saved/restored inside the statement.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void glds_1k(const char* g_lane, unsigned lds_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g_lane), "s"(lds_uniform)
      : "memory");
}

#define SPLIT_DMA 1   // measured +1.5 % over issuing a chunk's two KiB back to back
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define LDS_FENCE() asm volatile("" ::: "memory")

// Per-wave view of the workgroup-shared weight stream.
struct WStream {
  unsigned long long g_next;   // global address of the next chunk to issue (uniform; + voff per lane)
  unsigned voff;               // this lane's byte offset inside a chunk: wave*2048 + lane*16
  unsigned ring_lds;           // LDS byte address of ring slot 0 + wave*2048 (for M0)
  const char* ring_lane;       // generic pointer to ring slot 0 + lane*16 (for the ds_reads)
};

// Issue this wave's quarter (2 KiB) of the next chunk into ring slot `slot` (compile-time after unrolling).
// Instruction diet: address = SGPR base + VGPR offset (saddr form); the instruction offset applies to BOTH the global
// and the LDS address (LDS_addr = M0 + inst_offset + lane*16), so one M0 write serves both KiB; no M0 save/restore
// (nothing else in this kernel uses M0) -- 4 instructions per chunk instead of 14.
__device__ __forceinline__ void ws_issue(WStream& w, int slot) {
  const unsigned m0 = w.ring_lds + (unsigned)slot * (CH * 1024);
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:1024"
      :
      : "v"(w.voff), "s"(w.g_next), "s"(m0)
      : "memory");
  w.g_next += CH * 1024;
}
// The same two KiB as two issues half a chunk step apart: a wave issues in order and the texture addresser takes a
// 64-lane dwordx4 for ~64 cycles, so the second of two back-to-back DMAs stalls the wave's MFMA issue behind it.
__device__ __forceinline__ void ws_issue_a(WStream& w, int slot) {
  const unsigned m0 = w.ring_lds + (unsigned)slot * (CH * 1024);
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(w.voff), "s"(w.g_next), "s"(m0)
      : "memory");
}
__device__ __forceinline__ void ws_issue_b(WStream& w) {   // M0 still holds the slot address (nothing else writes M0)
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(w.voff), "s"(w.g_next) : "memory");
  w.g_next += CH * 1024;
}

// A operands of one k-step (entries 2j = hi, 2j+1 = lo) of the chunk in ring slot `slot`
struct AK { float4 hi, lo; };
__device__ __forceinline__ AK ws_read_k(const WStream& w, int slot, int j) {
  const float4* p = reinterpret_cast<const float4*>(w.ring_lane + slot * (CH * 1024));
  AK a;
#ifdef EXP_NOLDSREAD
  a.hi = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f); a.lo = a.hi;
  asm volatile("" : "+v"(a.hi.x), "+v"(a.lo.y));
#else
  a.hi = p[(2 * j) * 64];
  a.lo = p[(2 * j + 1) * 64];
#endif
  return a;
}

struct AK2 { AK c, n; };   // A operands of the current and the next k-step

// Top of pipeline step i of a stage: issue chunk i+D, make chunk i+1 visible to every wave.
__device__ __forceinline__ void ws_step(WStream& w, int i) {
#ifndef EXP_NODMA
#ifdef SPLIT_DMA
  ws_issue_a(w, (i + DPF) % NSLOT);
#else
  ws_issue(w, (i + DPF) % NSLOT);
#endif
#endif
#ifndef EXP_NOWAIT
#ifdef SPLIT_DMA
  WAIT_VMCNT(2 * DPF - 3);            // newest first: a(i+D), then both halves of chunks i+D-1 .. i+2
#else
  WAIT_VMCNT(2 * (DPF - 1));          // this wave's quarter of the next chunk has landed (loads retire in order)
#endif
#endif
#ifndef EXP_NOBARRIER
  __builtin_amdgcn_s_barrier();       // ... and every other wave's quarter
#endif
  LDS_FENCE();
}

// split 4 fp32 values into packed fp16 (hi, lo)
__device__ __forceinline__ void split4(const float (&v)[4], half4& hi, half4& lo) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)(v[t] - (float)h);
  }
}
__device__ __forceinline__ void put4(half8& dst, const half4& v, int q /* 0: slots 0-3, 1: slots 4-7 */) {
  if (q == 0) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
  else        { dst[4] = v[0]; dst[5] = v[1]; dst[6] = v[2]; dst[7] = v[3]; }
}

// FiLM epilogue of one quarter (accumulator registers 4q..4q+3 of n-block nbp), cut into four pieces that are issued
// behind the four k-steps of a chunk step: each piece is <= 8 VALU slots, which fit under the 32 cycles the k-step's last
// MFMA is still executing (a wave issues in order, so VALU placed after a dependent MFMA chain overlaps only its tail).
//   piece 0: 16 sin(2 pi (f'' acc + p')) for values 0,1      piece 1: values 2,3
//   piece 2: hi = rn_f16(v)                                    piece 3: lo = rn_f16(v - hi) (v_fma_mixlo_f16); store
// Outputs are the (hi, lo) halves of k-step 2*nbp + (q>>1), slots 4*(q&1)..+3: the first NBL n-blocks' outputs wait
// in the wave's LDS slab (unit (2*ks + which) = 64 lanes x 16 B), the rest in the y registers.
struct EpiQ {
  float4 f, p;
  float v[4];
  half4 hi;
};
// Differentiable mode: keep the quarter's raw accumulators (register dump, fenerf_layout.h "Tape"; here in the scaled
// units of the f16x3 GEMM, which the FiLM frequencies f'' already absorb).  tp = tape4 + (tile*L + layer)*(H/8)*64 + lane,
// or nullptr.  One fire-and-forget 1-KiB wave store; stores only make the counted vmcnt waits stricter, never wrong
// (loads retire in order among themselves).
// No guard: a branch inside the MFMA stream makes the compiler's vmcnt bookkeeping conservative.  The phantom tiles that pad a
// workgroup's quad write into the (up to 3 tiles of) slack fenerf_siren_tape_floats includes.
template <bool ON> struct TapeDst { float4* p; };
template <bool ON>
__device__ __forceinline__ void tape_q(const f32x16& acc, int nbp, int q, TapeDst<ON> tp) {
  if (ON) nt_store(tp.p + (nbp * 4 + q) * 64, acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}
__device__ __forceinline__ void epi_load(EpiQ& e, int nbp, int q, const float* film_f, const float* film_p) {
  e.f = *reinterpret_cast<const float4*>(film_f + 32 * nbp + 8 * q);   // + 4*h folded into the pointer
  e.p = *reinterpret_cast<const float4*>(film_p + 32 * nbp + 8 * q);
}
__device__ __forceinline__ void epi_p0(EpiQ& e, const f32x16& acc, int q) {
  e.v[0] = __builtin_amdgcn_sinf(__builtin_fmaf(e.f.x, acc[4 * q + 0], e.p.x)) * F16_ACT_SCALE;
  e.v[1] = __builtin_amdgcn_sinf(__builtin_fmaf(e.f.y, acc[4 * q + 1], e.p.y)) * F16_ACT_SCALE;
}
__device__ __forceinline__ void epi_p1(EpiQ& e, const f32x16& acc, int q) {
  e.v[2] = __builtin_amdgcn_sinf(__builtin_fmaf(e.f.z, acc[4 * q + 2], e.p.z)) * F16_ACT_SCALE;
  e.v[3] = __builtin_amdgcn_sinf(__builtin_fmaf(e.f.w, acc[4 * q + 3], e.p.w)) * F16_ACT_SCALE;
}
__device__ __forceinline__ void epi_p2(EpiQ& e) {
#pragma unroll
  for (int t = 0; t < 4; ++t) e.hi[t] = (_Float16)e.v[t];
}
template <int KS, int NBL>
__device__ __forceinline__ void epi_p3(EpiQ& e, int nbp, int q, half8 (&yh)[KS], half8 (&yl)[KS], char* slab) {
  half4 lo;
#pragma unroll
  for (int t = 0; t < 4; ++t) lo[t] = (_Float16)(e.v[t] - (float)e.hi[t]);   // one v_fma_mixlo_f16: fp32 difference, rounded once
  const int ks = 2 * nbp + (q >> 1);
  if (nbp < NBL) {
    *reinterpret_cast<half4*>(slab + (2 * ks + 0) * 1024 + (q & 1) * 8) = e.hi;
    *reinterpret_cast<half4*>(slab + (2 * ks + 1) * 1024 + (q & 1) * 8) = lo;
  } else {
    put4(yh[ks], e.hi, q & 1);
    put4(yl[ks], lo, q & 1);
  }
}
// whole quarter at once (layer tails, layer 0, small-H bodies)
template <int KS, int NBL, bool ON = false>
__device__ __forceinline__ void epi_route(const f32x16& acc, int nbp, int q, const float* film_f, const float* film_p,
                                          half8 (&yh)[KS], half8 (&yl)[KS], char* slab, TapeDst<ON> tp = TapeDst<ON>{nullptr}) {
  tape_q(acc, nbp, q, tp);
  EpiQ e;
  epi_load(e, nbp, q, film_f, film_p);
  epi_p0(e, acc, q);
  epi_p1(e, acc, q);
  epi_p2(e);
  epi_p3<KS, NBL>(e, nbp, q, yh, yl, slab);
}
// layer 0 writes straight into x (nothing is reading it yet)
template <int KS, bool ON>
__device__ __forceinline__ void epi_quarter(const f32x16& acc, int nbp, int q, const float* film_f, const float* film_p,
                                            half8 (&yh)[KS], half8 (&yl)[KS], TapeDst<ON> tp) {
  epi_route<KS, 0>(acc, nbp, q, film_f, film_p, yh, yl, nullptr, tp);
}

// 3 MFMAs of one k-step: wl*xh + wh*xl + wh*xh
#define KSTEP_MFMA(acc, ak, bh, bl)                  \
  do {                                               \
    (acc) = MFMA16(as_half8((ak).lo), (bh), (acc));  \
    (acc) = MFMA16(as_half8((ak).hi), (bl), (acc));  \
    (acc) = MFMA16(as_half8((ak).hi), (bh), (acc));  \
  } while (0)

// Chunk step i of a stage: barrier (chunk i+1 visible), then 4 k-steps.  bop(k, bh, bl) supplies the B operands of
// k-step k (false = padding k-step); piece(j) is the epilogue piece issued behind k-step j.  a_cur holds the A operands of
// the chunk's first k-step on entry and of chunk i+1's first k-step on exit: A operands are read from the LDS ring one
// k-step (96 MFMA cycles) ahead of use.  Ring slot of chunk i = i % NSLOT (stages start at slot 0).
template <class BOP, class PIECE>
__device__ __forceinline__ void chunk_step(f32x16& acc, AK2& a, WStream& ws, int i, int k0, BOP bop, PIECE piece) {
  ws_step(ws, i);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#if defined(SPLIT_DMA) && !defined(EXP_NODMA)
    if (j == 2) ws_issue_b(ws);
#endif
    // A operands are fetched TWO k-steps (192 MFMA cycles) ahead of use: one k-step did not cover the loaded LDS latency
    const AK a_nn = (j < 2) ? ws_read_k(ws, i % NSLOT, j + 2) : ws_read_k(ws, (i + 1) % NSLOT, j - 2);
    half8 bh, bl;
    if (bop(k0 + j, bh, bl)) KSTEP_MFMA(acc, a.c, bh, bl);
    piece(j);
    a.c = a.n;
    a.n = a_nn;
    __builtin_amdgcn_sched_barrier(0);
  }
}
// Stage-padding chunk (no weights in it): keep the DMA / barrier cadence, fetch the next chunk's first two k-steps.
__device__ __forceinline__ void chunk_skip(AK2& a, WStream& ws, int i) {
  ws_step(ws, i);
#if defined(SPLIT_DMA) && !defined(EXP_NODMA)
  ws_issue_b(ws);
#endif
  a.c = ws_read_k(ws, (i + 1) % NSLOT, 0);
  a.n = ws_read_k(ws, (i + 1) % NSLOT, 1);
  __builtin_amdgcn_sched_barrier(0);
}

// Layer end: x <- outputs (first 2*NBL k-steps from the slab, the rest from y).
template <int KS, int NBL>
__device__ __forceinline__ void collect_act(half8 (&xh)[KS], half8 (&xl)[KS], const half8 (&yh)[KS], const half8 (&yl)[KS],
                                            const char* slab) {
  LDS_FENCE();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks < 2 * NBL) {
      xh[ks] = as_half8(*reinterpret_cast<const float4*>(slab + (2 * ks + 0) * 1024));
      xl[ks] = as_half8(*reinterpret_cast<const float4*>(slab + (2 * ks + 1) * 1024));
    } else {
      xh[ks] = yh[ks];
      xl[ks] = yl[ks];
    }
  }
}

// A square FiLM layer H -> H.  x: input activations (B operands); outputs replace x at the end.
template <int H, bool ON>
__device__ __forceinline__ void square_layer_s(half8 (&xh)[H / 16], half8 (&xl)[H / 16], WStream& ws, AK2& a_cur,
                                               const float* film_f, const float* film_p, char* slab, TapeDst<ON> tp) {
  constexpr int NB = H / 32, KS = H / 16, NBL = NB / 2;
  constexpr int QB = (2 * KS + CH - 1) / CH;          // chunks per n-block body (4 at H=256)
  constexpr int STAGE_CHUNKS = pad_stage(NB * QB * CH) / CH;
  constexpr int EQ = 4 / QB > 0 ? 4 / QB : 1;         // epilogue quarters per chunk
  half8 yh[KS], yl[KS];
  auto bop = [&](int k, half8& bh, half8& bl) -> bool {
    if (k < KS) { bh = xh[k]; bl = xl[k]; return true; }
    return false;
  };
  f32x16 acc_prev = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int qc = 0; qc < QB; ++qc) {
      // FiLM epilogue of the previous n-block, one quarter per chunk step (QB == 4), piece-wise behind the k-steps
      EpiQ eq;
      const bool fine = nb > 0 && EQ == 1;
      if (fine) { tape_q(acc_prev, nb - 1, qc, tp); epi_load(eq, nb - 1, qc, film_f, film_p); }
      chunk_step(acc, a_cur, ws, nb * QB + qc, 4 * qc, bop, [&](int j) {
        if (fine) {
          if (j == 0) epi_p0(eq, acc_prev, qc);
          else if (j == 1) epi_p1(eq, acc_prev, qc);
          else if (j == 2) epi_p2(eq);
          else epi_p3<KS, NBL>(eq, nb - 1, qc, yh, yl, slab);
        } else if (nb > 0 && j < EQ) {
          epi_route<KS, NBL>(acc_prev, nb - 1, qc * EQ + j, film_f, film_p, yh, yl, slab, tp);
        }
      });
    }
    acc_prev = acc;
  }
#pragma unroll
  for (int i = NB * QB; i < STAGE_CHUNKS; ++i) chunk_skip(a_cur, ws, i);
#pragma unroll
  for (int q = 0; q < 4; ++q) epi_route<KS, NBL>(acc_prev, NB - 1, q, film_f, film_p, yh, yl, slab, tp);
  collect_act<KS, NBL>(xh, xl, yh, yl, slab);
}

// A head body (labels+sigma, rgb): acc over the whole activation, no FiLM.
template <int H>
__device__ __forceinline__ void head_body_s(f32x16& acc, const half8 (&xh)[H / 16], const half8 (&xl)[H / 16], WStream& ws,
                                            AK2& a_cur) {
  constexpr int KS = H / 16;
  constexpr int QB = (2 * KS + CH - 1) / CH;
  constexpr int STAGE_CHUNKS = pad_stage(QB * CH) / CH;
  auto bop = [&](int k, half8& bh, half8& bl) -> bool {
    if (k < KS) { bh = xh[k]; bl = xl[k]; return true; }
    return false;
  };
#pragma unroll
  for (int qc = 0; qc < QB; ++qc) {
    chunk_step(acc, a_cur, ws, qc, 4 * qc, bop, [](int) {});
  }
#pragma unroll
  for (int i = QB; i < STAGE_CHUNKS; ++i) chunk_skip(a_cur, ws, i);
}

template <int H, bool GRID, bool SAVE>
__global__ __launch_bounds__(256, 1) void siren16s_kernel(SirenParams P, int n_geo, int n_color, int n_lab, int C) {
  constexpr int NB = H / 32, KS = H / 16;
  constexpr int C0_KS = KS + (GRID ? 2 : 0) + 1;
  constexpr int C0_QB = (2 * C0_KS + CH - 1) / CH;     // chunks of a colour-layer-0 body (5 at H=256 with grid)
  constexpr int SQ_CHUNKS = pad_stage(NB * ((2 * KS + CH - 1) / CH) * CH) / CH;   // chunks per square layer (>= 8)
  constexpr int NBL = NB / 2;                                 // n-blocks whose outputs wait in the LDS slab
  constexpr int SLAB_BYTES = NBL * 4 * 1024;                  // per wave: 2*NBL k-steps x (hi, lo) x 1 KiB
  extern __shared__ __attribute__((aligned(16))) float4 smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int L = n_geo + n_color;
  const int stage_f4 = (32 * C + 3) / 4;
  // LDS carve-up: [ring NSLOT*CH KiB][film: 4 waves x 2 buffers x (f 1 KiB... H floats, p H floats)][stage 4 x 32*C floats]
  char* lds = reinterpret_cast<char*>(smem);
  char* ring = lds;
  constexpr int FILM_F = H * 4 < 1024 ? 1024 : H * 4;   // LDS-DMA moves whole KiBs: pad the per-layer f'' / p' buffers
  constexpr int FILM_BYTES = 2 * FILM_F;                // f'' and p' of one layer
  char* film_base = lds + NSLOT * CH * 1024 + wave * (2 * FILM_BYTES);
  float* stage = reinterpret_cast<float*>(lds + NSLOT * CH * 1024 + 4 * 2 * FILM_BYTES) + wave * stage_f4 * 4;
  char* slab = lds + NSLOT * CH * 1024 + 4 * 2 * FILM_BYTES + 4 * stage_f4 * 16 + wave * SLAB_BYTES + lane * 16;

  const float4* l0w = reinterpret_cast<const float4*>(P.stream) + lane;

  WStream ws;
  const unsigned long long g_stream = reinterpret_cast<unsigned long long>(P.stream + P.ring_offset_floats);
  ws.g_next = g_stream;
  ws.voff = wave * 2048 + lane * 16;
  ws.ring_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring) + wave * 2048);
  ws.ring_lane = ring + lane * 16;

  // ---- prime the shared stream: chunks 0..D-1 in flight, first k-step of chunk 0 in registers
#pragma unroll
  for (int i = 0; i < DPF; ++i) ws_issue(ws, i);
  WAIT_VMCNT(2 * (DPF - 1));
  __builtin_amdgcn_s_barrier();
  LDS_FENCE();
  AK2 a_cur;
  a_cur.c = ws_read_k(ws, 0, 0);
  a_cur.n = ws_read_k(ws, 0, 1);

  // work split: quads of tiles (one tile per wave), XCD-contiguous ranges
  const long long ntiles = (P.P + 31) / 32;
  const long long nquads = (ntiles + 3) / 4;
  const int nblk = gridDim.x;
  const int nx = nblk < 8 ? nblk : 8;
  const int x = blockIdx.x % nx, bi = blockIdx.x / nx;
  const int blocks_in_x = nblk / nx + (x < nblk % nx ? 1 : 0);
  const long long q_begin = nquads * x / nx, q_end = nquads * (x + 1) / nx;

  for (long long quad = q_begin + bi; quad < q_end; quad += blocks_in_x) {
    // the previous tile ended by issuing the replicated head chunks nchunk..nchunk+D-1 (== this tile's chunks 0..D-1)
    ws.g_next = g_stream + (unsigned long long)DPF * (CH * 1024);
    const long long tile = quad * 4 + wave;
    // ---------------- this lane's point ----------------
    long long pt = tile * 32 + m;
    if (pt >= P.P) pt = P.P - 1;
    const long long img = __builtin_amdgcn_readfirstlane((int)(pt / P.pts_per_image));   // launcher guarantees tiles do not straddle images
    float px, py, pz, dx, dy, dz;
    if (P.points) {
      px = P.points[pt * 3 + 0]; py = P.points[pt * 3 + 1]; pz = P.points[pt * 3 + 2];
      if (P.pdirs) { dx = P.pdirs[pt * 3 + 0]; dy = P.pdirs[pt * 3 + 1]; dz = P.pdirs[pt * 3 + 2]; }
      else { dx = 0.f; dy = 0.f; dz = -1.f; }
    } else {
      const long long ray = pt / P.n_per_ray;
      const float zz = P.z[pt];
      const float ox = P.origins[ray * 3 + 0], oy = P.origins[ray * 3 + 1], oz = P.origins[ray * 3 + 2];
      dx = P.dirs[ray * 3 + 0]; dy = P.dirs[ray * 3 + 1]; dz = P.dirs[ray * 3 + 2];
      px = __fadd_rn(ox, __fmul_rn(dx, zz)); py = __fadd_rn(oy, __fmul_rn(dy, zz)); pz = __fadd_rn(oz, __fmul_rn(dz, zz));
      if (P.lock_view) { dx = 0.f; dy = 0.f; dz = -1.f; }
    }
    const float qx = px * P.box_scale, qy = py * P.box_scale, qz = pz * P.box_scale;

    // ---------------- FiLM parameters of layers 0 and 1 -> LDS (DMA), grid gather, then one drain ----------------
    const float* fp_img = P.fp + (size_t)img * L * H;
    const float* pp_img = P.pp + (size_t)img * L * H;
    auto film_issue = [&](int layer) {   // f'' (H floats) then p' (H floats) of `layer` into buffer layer&1
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(film_base) + (layer & 1) * FILM_BYTES);
      const char* gf = reinterpret_cast<const char*>(fp_img + (size_t)layer * H) + lane * 16;
      const char* gp = reinterpret_cast<const char*>(pp_img + (size_t)layer * H) + lane * 16;
      for (int off = 0; off < H * 4; off += 1024) {   // H=256: one KiB each; smaller H: lanes beyond H/4 read in-bounds pad
        glds_1k(gf + off, dst + off);
        glds_1k(gp + off, dst + FILM_F + off);
      }
    };
    film_issue(0);
    if (L > 1) film_issue(1);

    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = 0.f;
    if (GRID) {
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
          const float4* g = reinterpret_cast<const float4*>(P.grid + vox * 32 + 16 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = g[q];
            e[4 * q + 0] += v.x * wgt; e[4 * q + 1] += v.y * wgt; e[4 * q + 2] += v.z * wgt; e[4 * q + 3] += v.w * wgt;
          }
        }
      }
    }
    // quads are padded with phantom tiles (clamped points): their tape goes to the slack behind the last tile
    const TapeDst<SAVE> tp0{SAVE ? reinterpret_cast<float4*>(P.tape) + tile * L * (long long)((H / 8) * 64) + lane : nullptr};
    if (SAVE && GRID && tile * 32 + m < P.P) {
      float4* ep = reinterpret_cast<float4*>(P.tape_e + (tile * 32 + m) * 32 + 16 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) ep[q] = make_float4(e[4 * q + 0], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]);
    }
    WAIT_VMCNT(0);      // film 0/1 landed (own buffer, own reads: no barrier needed); once per tile
    LDS_FENCE();

    const float* film_f0 = reinterpret_cast<const float*>(film_base) + 4 * h;
    const float* film_p0 = film_f0 + FILM_F / 4;
    const float* film_f1 = reinterpret_cast<const float*>(film_base + FILM_BYTES) + 4 * h;
    const float* film_p1 = film_f1 + FILM_F / 4;

    half8 xh[KS], xl[KS];
    // ---------------- layer 0: 3 -> H on the exact fp32 MFMA.  k-steps (x|y), (z|0) ----------------
    {
      const float b0 = h ? qy : qx, b1 = h ? 0.f : qz;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float4 w = l0w[nb * 64];
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = MFMA32(w.x, b0, acc);
        acc = MFMA32(w.y, b1, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) epi_quarter<KS>(acc, nb, q, film_f0, film_p0, xh, xl, tp0);
      }
    }
    WAIT_VMCNT(0);      // l0w loads are ordinary loads: keep the compiler's own vmcnt bookkeeping out of the main loop
    // ---------------- geometry trunk G1 .. G(n_geo-1) ----------------
#pragma unroll 1
    for (int l = 1; l < n_geo + n_color; ++l) {
      const float* ff = (l & 1) ? film_f1 : film_f0;
      const float* fq = (l & 1) ? film_p1 : film_p0;
      const TapeDst<SAVE> tpl{SAVE ? tp0.p + (long long)l * ((H / 8) * 64) : nullptr};
      if (l == n_geo) {
        // ---------------- colour layer 0: [x | grid feats | dir] -> H, then the label/sigma head on the same x -------
        half8 eh[2], el[2], dh, dl;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float w4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) w4[t] = e[8 * j + 4 * q + t] * F16_ACT_SCALE;
            half4 hi, lo;
            split4(w4, hi, lo);
            put4(eh[j], hi, q); put4(el[j], lo, q);
          }
        }
        {
          float w4[4] = {dx * F16_ACT_SCALE, dy * F16_ACT_SCALE, dz * F16_ACT_SCALE, 0.f};
          half4 hi, lo, z4 = {0, 0, 0, 0};
          split4(w4, hi, lo);
          put4(dh, hi, 0); put4(dh, z4, 1); put4(dl, lo, 0); put4(dl, z4, 1);
        }
        if (l + 1 < L) film_issue(l + 1);
        if (SQ_CHUNKS < DPF) WAIT_VMCNT(0);
        auto bop0 = [&](int k, half8& bh, half8& bl) -> bool {
          if (k < KS) { bh = xh[k]; bl = xl[k]; return true; }
          if (GRID && k < KS + 2) { bh = eh[k - KS]; bl = el[k - KS]; return true; }
          if (k == C0_KS - 1) { bh = dh; bl = dl; return true; }
          return false;
        };
        half8 yh[KS], yl[KS];
        f32x16 acc_prev = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int qc = 0; qc < C0_QB; ++qc) {
            EpiQ eq;
            const bool fine = nb > 0 && qc < 4;
            if (fine) { tape_q(acc_prev, nb - 1, qc, tpl); epi_load(eq, nb - 1, qc, ff, fq); }
            chunk_step(acc, a_cur, ws, nb * C0_QB + qc, 4 * qc, bop0, [&](int j) {
              if (fine) {
                if (j == 0) epi_p0(eq, acc_prev, qc);
                else if (j == 1) epi_p1(eq, acc_prev, qc);
                else if (j == 2) epi_p2(eq);
                else epi_p3<KS, NBL>(eq, nb - 1, qc, yh, yl, slab);
              }
            });
            if (nb > 0 && C0_QB < 4 && qc == C0_QB - 1) {
#pragma unroll
              for (int q = C0_QB; q < 4; ++q) epi_route<KS, NBL>(acc_prev, nb - 1, q, ff, fq, yh, yl, slab, tpl);
            }
          }
          acc_prev = acc;
        }
#pragma unroll
        for (int i = NB * C0_QB; i < pad_stage(NB * C0_QB * CH) / CH; ++i) chunk_skip(a_cur, ws, i);
#pragma unroll
        for (int q = 0; q < 4; ++q) epi_route<KS, NBL>(acc_prev, NB - 1, q, ff, fq, yh, yl, slab, tpl);
        // head on x (the trunk output), before x is overwritten with the colour-layer-0 activations
        {
          f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          head_body_s<H>(acc, xh, xl, ws, a_cur);
          const float* head_inv = P.consts + CONST_FILM_BIAS + (size_t)2 * L * H;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row <= n_lab) {
              const int ch = row < n_lab ? row : C - 1;
              stage[m * C + ch] = acc[r] * head_inv[row] + P.consts[CONST_HEAD_BIAS + row];
            }
          }
          WAIT_VMCNT(0);   // consts were ordinary loads
        }
        collect_act<KS, NBL>(xh, xl, yh, yl, slab);
      } else {
        if (l + 1 < L) film_issue(l + 1);
        if (SQ_CHUNKS < DPF) WAIT_VMCNT(0);
        square_layer_s<H>(xh, xl, ws, a_cur, ff, fq, slab, tpl);
      }
    }
    // ---------------- rgb head + sigmoid ----------------
    {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      head_body_s<H>(acc, xh, xl, ws, a_cur);
      const float* rgb_inv = P.consts + CONST_FILM_BIAS + (size_t)2 * L * H + 32;
      if (h == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float v = acc[r] * rgb_inv[r] + P.consts[CONST_RGB_BIAS + r];
          stage[m * C + (C - 4) + r] = 1.f / (1.f + __expf(-v));
        }
      }
    }
    // ---------------- coalesced write-out of the tile's [32][C] block ----------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const long long base = tile * 32 * C;
      const long long limit = P.P * C;
      for (int i = lane; i < 32 * C; i += 64)
        if (base + i < limit) P.out[base + i] = stage[i];
    }
    WAIT_VMCNT(0);   // stores may retire out of order with the DMA loads: keep them out of the counted waits
    __builtin_amdgcn_wave_barrier();
  }
  WAIT_VMCNT(0);     // no LDS-DMA may land after the workgroup has released its LDS
  __builtin_amdgcn_s_barrier();
}

static int hip_fail16s(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

template <int H, bool GRID, bool SAVE>
static int launch_siren16s_t(const FenerfModel* m, const SirenParams& p, void* stream) {
  const int stage_f4 = (32 * m->C + 3) / 4;
  const size_t film_f = H * 4 < 1024 ? 1024 : H * 4;
  const size_t lds = (size_t)NSLOT * CH * 1024 + (size_t)4 * 2 * (2 * film_f) + (size_t)4 * stage_f4 * 16 +
                     (size_t)4 * (H / 64) * 4 * 1024;   // ring + FiLM buffers + output staging + activation slabs
  auto kfn = siren16s_kernel<H, GRID, SAVE>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  const long long ntiles = (p.P + 31) / 32;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > m->num_cus) blocks = m->num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail16s(e, "siren16s launch");
}

// Tiles must not straddle images (FiLM parameters are fetched per wave): when points-per-image is not a multiple of
// 32, launch image by image (each launch ends in a ragged tile).
int launch_siren16s(const FenerfModel* m, const SirenParams& p, void* stream) {
  if (p.P <= 0) return FENERF_OK;
  const bool g = m->grid_ch != 0;
  // all launches run the 16-point / 2-waves-per-SIMD kernel (fenerf_siren_f16w.hip, same stream, same tape layout);
  // FENERF_FORWARD_KERNEL=f16s forces this kernel (A/B timing only).
  static const bool force_s = [] { const char* v = getenv("FENERF_FORWARD_KERNEL"); return v && std::string(v) == "f16s"; }();
  auto one = [&](const SirenParams& q) -> int {
    const bool sv = q.tape != nullptr;
    if (!force_s) return launch_siren16w_one(m, q, stream);
    switch (m->H) {
      case 32: return sv ? (g ? launch_siren16s_t<32, true, true>(m, q, stream) : launch_siren16s_t<32, false, true>(m, q, stream))
                         : (g ? launch_siren16s_t<32, true, false>(m, q, stream) : launch_siren16s_t<32, false, false>(m, q, stream));
      case 64: return sv ? (g ? launch_siren16s_t<64, true, true>(m, q, stream) : launch_siren16s_t<64, false, true>(m, q, stream))
                         : (g ? launch_siren16s_t<64, true, false>(m, q, stream) : launch_siren16s_t<64, false, false>(m, q, stream));
      case 128: return sv ? (g ? launch_siren16s_t<128, true, true>(m, q, stream) : launch_siren16s_t<128, false, true>(m, q, stream))
                          : (g ? launch_siren16s_t<128, true, false>(m, q, stream) : launch_siren16s_t<128, false, false>(m, q, stream));
      case 256: return sv ? (g ? launch_siren16s_t<256, true, true>(m, q, stream) : launch_siren16s_t<256, false, true>(m, q, stream))
                          : (g ? launch_siren16s_t<256, true, false>(m, q, stream) : launch_siren16s_t<256, false, false>(m, q, stream));
    }
    set_error("unsupported hidden_dim");
    return FENERF_E_UNSUPPORTED;
  };
  if (p.pts_per_image % 32 == 0 || p.P == p.pts_per_image) return one(p);
  const long long nimg = p.P / p.pts_per_image;
  const int L = m->L, H = m->H;
  for (long long b = 0; b < nimg; ++b) {
    SirenParams q = p;
    q.P = p.pts_per_image;
    q.fp = p.fp + (size_t)b * L * H;
    q.pp = p.pp + (size_t)b * L * H;
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
    int rc = one(q);
    if (rc) return rc;
  }
  return FENERF_OK;
}

}  // namespace fenerf
