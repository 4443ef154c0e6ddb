// Backward chain of the FiLM-SIREN radiance field on the bf16 matrix pipe, 16-point waves, two waves per SIMD.
//
// The maths of the fp32 chain kernel (fenerf_siren_bwd.hip; reference: torch autograd through siren/siren.py:1509-1530) on bf16
// (hi, lo) operand splits -- three MFMAs per product, fp32 accumulate -- in the execution shape of the forward (fenerf_siren_f16w.hip):
//
//   * a workgroup is 8 waves = 2 per SIMD, each owning 16 points on v_mfma_f32_16x16x32_bf16; dz lives in registers as the next
//     stage's B operand (64 + 64 registers), never in LDS;
//   * ONE copy of the packed backward stream per workgroup travels through an LDS ring (8 slots x 8 KiB, 6 chunks ahead); every
//     wave's LDS-DMA fetches one ready-made 1-KiB A operand per chunk by pointing its lanes at the right 16-byte pieces of the
//     32x32x16 entries (same re-tiling map as the forward: the two streams have the same entry format and k order);
//   * the tape is read by LDS-DMA too (1 KiB per wave, n-block and row tile, two n-blocks ahead of its use), so every load of
//     the stream loop sits in ONE in-order queue whose depth at every wait is a compile-time number (ring_wait below);
//     d(theta) and the FiLM sums leave by fire-and-forget stores, which are not counted (they only make a wait stricter);
//   * the epilogue of n-block nb - 1 (cos, d theta, d z split into bf16 hi / lo, FiLM sums) is issued in four items behind the
//     MFMAs of n-block nb.
//
// (Round 1's private-stream kernel -- one 32-point wave per SIMD, every wave its own L2 stream through a register ring; retired in
// round 3, git history -- stalled on its tape loads: the in-order vmcnt queue put every HBM load in front of ring entries needed 4
// k-steps later.  Here the other wave of the SIMD fills those slots.  Measurements: DESIGN.md 4.5.)
//
// FiLM sums are emitted in register-dump order (film_gather_kernel, fenerf_siren_wgrad.hip, decodes it):
//   [unit][layer][nb][rt][slot = 4 g + r][s0, s1],  feature = 32 nb + 16 (g >> 1) + 4 (g & 1) + 8 rt + r,
// unit = the workgroup's oct of eight 16-point tiles (128 points) when an oct cannot straddle images (WGS: one image per launch,
// or points per image a multiple of 128) -- the eight waves' sums meet in LDS (three rotating 2-KiB buffers; the wave whose
// turn it is adds them in wave order, deterministic, one step of workgroup barriers after they were written) and leave as
// ONE 256-B store per n-block instead of eight: 23 MB instead of 184 MB per 131,072-point launch, written here and read by the
// gather -- otherwise unit = the 16-point tile, each wave storing its own sums.
//
// T16 (round 5, fenerf_layout.h "16-bit tape"): the tape holds frac(theta) as 16-bit fixed point in the bf16 dump's piece layout -- ONE
// 1-KiB tape DMA per body (both row tiles), theta is a convert and a multiply instead of the FiLM fma, and the second FiLM sum (sum of
// d theta * tape, the frequency gradient's raw material) is not formed: the weight-gradient stage derives the frequency gradient from
// its own partial sums.  The T16 instantiations live in a second translation unit (fenerf_siren_bwd16w_t16.hip includes this file with
// FENERF_BW16_T16 = 1) so that the two halves compile side by side.
//
// S1 = false (FENERF_TAPE_U16, FENERF_TAPE_F32_W): the second FiLM sum -- sum_p d theta * tape, the frequency gradient's raw material:
// a multiply and a 16-lane butterfly per row tile -- is not formed (a fifth of the kernel's VALU instructions); the weight-gradient stage
// derives the frequency gradient from its own partial sums (fenerf_siren_wgrad.hip).  FiLM-only launches (inversion) always run S1 = true.
//
// With P.d_grid_cl set (fenerf_siren_backward_grid) the gradient wrt the sampled grid features is not written to d_e: the kernel
// scatters it into the channels-last gradient grid itself (scatter_pairs below: float atomics, the arithmetic of
// grid_backward_kernel), two points x 32 channels per atomic instruction, one point pair per body of the following stage.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <type_traits>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_trig.h"

#ifndef FENERF_WAVE_HALF_COPIES
#define FENERF_WAVE_HALF_COPIES 1     // 0: the wave half as a run-time flag inside the stream loop (rounds 3-5); A/B builds only
#endif
// Timing ablations of the stage bodies (tools/exp/chain_ablations.sh; WRONG results): 1 = no FiLM sums (the B items: row butterfly, LDS
// hand-over, combine), 2 = no d(theta) stores, 4 = tape DMA from one L2-resident block instead of the tile's (same instruction, same queue)
#ifndef FENERF_CHAIN_SETPRIO
#define FENERF_CHAIN_SETPRIO 1        // round 6: chain 4.16-4.19 -> 4.08-4.11 ms per step in same-box A/B (tools/gpu_r6.sh prioab)
#endif
#ifndef FENERF_EXP_CHAIN_ABLATE
#define FENERF_EXP_CHAIN_ABLATE 0
#endif
#ifndef FENERF_BW16_T16
#define FENERF_BW16_T16 0      // tape mode of this translation unit: 0 = FENERF_TAPE_F32, 1 = FENERF_TAPE_U16, 2 = FENERF_TAPE_F32_W
#endif

namespace fenerf {
namespace bw16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int CH = FENERF_CH;        // entries (KiB) per chunk = one A operand per wave
constexpr int DPF = FENERF_DPF;      // chunks in flight ahead of the chunk being consumed
constexpr int NSLOT = FENERF_NSLOT;  // LDS ring slots
constexpr int NWAVE = 8;
static_assert(CH == NWAVE, "one 1-KiB A operand per wave and chunk");
static_assert(CH == FENERF_PF16, "bodies of the backward stream are whole chunks");
static_assert(NSLOT == 8 && NSLOT >= DPF + 2, "slot arithmetic below is & 7; a slot is refilled two barriers after its last reads");

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ int opaque(int v) {   // fenerf_siren_f16w.hip: keeps lane-derived address arithmetic local
  asm volatile("" : "+v"(v));
  return v;
}
// a wave-uniform pointer the compiler computed with vector instructions (64-bit multiplies) -> SGPRs, for the "s" operands below
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define LDS_FENCE() asm volatile("" ::: "memory")

// LDS-DMA of one KiB: lane i's 16 bytes at g_uniform + voff  ->  lds_uniform + 16 i
__device__ __forceinline__ void glds_1k_s(const void* g_uniform, unsigned voff, unsigned lds_uniform) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(g_uniform), "s"(lds_uniform)
      : "memory");
}
__device__ __forceinline__ void glds_1k_s_nt(const void* g_uniform, unsigned voff, unsigned lds_uniform) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1 nt"
      :
      : "v"(voff), "s"(g_uniform), "s"(lds_uniform)
      : "memory");
}
// fire-and-forget stores, uniform base + 32-bit lane offset.  The s_nop is the hazard slot the compiler would insert behind a
// store of more than 8 bytes whose data registers the next VALU instruction overwrites -- it does not look inside an asm.
#ifndef FENERF_ST_POLICY
#define FENERF_ST_POLICY "nt"      // cache policy of the fire-and-forget tape / d(theta) stores (A/B builds: profiles/r06_store_policy_ab.txt)
#endif
__device__ __forceinline__ void st_f4_nt(const void* g_uniform, unsigned voff, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 " FENERF_ST_POLICY "\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(g_uniform) : "memory");
}
__device__ __forceinline__ void st_u4_nt(const void* g_uniform, unsigned voff, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 " FENERF_ST_POLICY "\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(g_uniform) : "memory");
}
__device__ __forceinline__ void st_f2(const void* g_uniform, unsigned voff, const f32x2& v) {
  asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(voff), "v"(v), "s"(g_uniform) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// The in-order load queue.  Per chunk step a wave issues the ring DMA of chunk i + DPF, then one tape DMA per epilogue item
// E scheduled in that step -- items of a body: 0 = E(row tile 0): d theta store, tape DMA;  1 = B(0): FiLM-sum store;
// 2 = E(1);  3 = B(1); item k runs in chunk k * min(QB, 4) / 4.  Every body issues its two tape DMAs (the last body of the
// last stage re-fetches a block it does not need), so the number of LOADS behind any ring DMA is a compile-time number.
// At the top of step i chunk i + 1 must have landed; loads return in order, so everything issued after its DMA may still
// be in flight: the DMAs of chunks i + 2 .. i + DPF - 1 and the tape DMAs of steps i + 1 - DPF .. i - 1.  Steps before the
// stage's first count as zero, and the stores are not counted at all (they may retire out of order with the loads): a
// smaller number, or outstanding stores on top of it, only wait for more.
// ---------------------------------------------------------------------------------------------------------------------
// 16-bit tape (`t16`): ONE DMA per body -- both row tiles' phases are one 16-byte piece per lane -- issued by item 2 = E(1), i.e. only
// after BOTH row tiles of the block that sits in the target staging buffer (same parity, two n-blocks older) have been read: item 0's
// slot would overwrite the second row tile's half one or two chunk steps before E(1) reads it.
constexpr int item_chunk(int QB, int k) { return k * (QB < 4 ? QB : 4) / 4; }
constexpr int step_loads(int QB, int qc, bool t16 = false) {
  int n = 0;
  for (int k = (t16 ? 2 : 0); k < 4; k += 2)
    if (item_chunk(QB, k) == qc) n += 1;
  return n;
}
constexpr int ring_wait(int QB, int s, bool t16 = false) {
  int n = DPF - 2;
  for (int j = 1; j <= DPF - 1; ++j) {
    const int t = s - j;
    if (t >= 0) n += step_loads(QB, t % QB, t16);
  }
  return n;
}

// One barrier per TWO chunk steps (stages of an even number of steps): the barrier of even step i publishes chunks i + 1 and
// i + 2 (read during steps i and i + 1), so every wave first waits for its own KiB of chunks <= i + 2.
constexpr int ring_wait2(int QB, int s, bool t16 = false) {
  int n = DPF - 3;
  for (int j = 1; j <= DPF - 2; ++j) {
    const int t = s - j;
    if (t >= 0) n += step_loads(QB, t % QB, t16);
  }
  return n;
}

struct WStream {
  unsigned long long g_next;   // global address of the next chunk to issue (uniform)
  unsigned long long g_begin, g_end;
  unsigned voff;               // this lane's byte offset inside a chunk (the re-tiling permutation)
  unsigned ring_lds;           // LDS byte address of ring slot 0 + wave * 1024 (for M0)
  const char* ring_lane;       // ring slot 0 + lane * 16 (for the ds_reads)
  int cs;                      // ring slot of the chunk being consumed (uniform)
  bool early;                  // waves 0-3: DMA at the top of a chunk step; waves 4-7: half a step later
};

__device__ __forceinline__ void ws_issue(WStream& w, int slot) {
  const unsigned m0 = w.ring_lds + (unsigned)slot * (CH * 1024);
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(w.voff), "s"(w.g_next), "s"(m0)
      : "memory");
  const unsigned long long nx = w.g_next + CH * 1024;
  w.g_next = nx == w.g_end ? w.g_begin : nx;      // the stream restarts for the next tile
}

// A operands of one k32-step (both row tiles): ring slot layout = operand index (spl * 2 + rt) * 2 + hl, 1 KiB each
struct AK { float4 hi[2], lo[2]; };
__device__ __forceinline__ AK ws_read(const WStream& w, int slot, int spl) {
  AK a;
  const float4* p = reinterpret_cast<const float4*>(w.ring_lane + slot * (CH * 1024) + spl * 4096);
  a.lo[0] = p[1 * 64]; a.lo[1] = p[3 * 64];
  a.hi[0] = p[0 * 64]; a.hi[1] = p[2 * 64];
  return a;
}

__device__ __forceinline__ bf16x8 as_bf16x8(const float4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// 6 MFMAs of one k32-step, the two row tiles interleaved: wl*xh + wh*xl + wh*xh
__device__ __forceinline__ void kstep_mfma(f32x4 (&acc)[2], const AK& a, const bf16x8& bh, const bf16x8& bl) {
  acc[0] = MFMA16B(as_bf16x8(a.lo[0]), bh, acc[0]);
  acc[1] = MFMA16B(as_bf16x8(a.lo[1]), bh, acc[1]);
  acc[0] = MFMA16B(as_bf16x8(a.hi[0]), bl, acc[0]);
  acc[1] = MFMA16B(as_bf16x8(a.hi[1]), bl, acc[1]);
  acc[0] = MFMA16B(as_bf16x8(a.hi[0]), bh, acc[0]);
  acc[1] = MFMA16B(as_bf16x8(a.hi[1]), bh, acc[1]);
}

__device__ __forceinline__ float lane_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)); }   // quad_perm:[1,0,3,2]
__device__ __forceinline__ float lane_xor2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)); }   // quad_perm:[2,3,0,1]
__device__ __forceinline__ float lane_ror4(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, true)); }  // row_ror:4
__device__ __forceinline__ float lane_ror8(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, true)); }  // row_ror:8

// Sum of 4 per-lane values over the 16 lanes of a row (= the 16 points of the tile; a lane group IS a DPP row): two transposing
// steps (lane keeps, of each pair, the partial sum its lane bit selects and receives the partner's), then two rotations.
// Every lane n ends with the 16-point sum of value n & 3.
__device__ __forceinline__ float row_sum4(const f32x4& v, bool b0, bool b1) {
  const float k0 = b0 ? v[1] : v[0], s0 = b0 ? v[0] : v[1];
  const float k1 = b0 ? v[3] : v[2], s1 = b0 ? v[2] : v[3];
  const float w0 = k0 + lane_xor1(s0), w1 = k1 + lane_xor1(s1);
  const float k = b1 ? w1 : w0, s = b1 ? w0 : w1;
  float x = k + lane_xor2(s);
  x += lane_ror4(x);
  x += lane_ror8(x);
  return x;
}

// what an epilogue item needs from its stage
struct Sink {
  const float* film;        // LDS: f'' of the stage's FiLM layer at this lane group's first feature; p' FILM_F bytes behind
  const char* tape_lane;    // LDS: this wave's tape staging buffers + lane * 16
  const char* tape_base;    // global (uniform): tape of (tile32, layer), and of the next stage's FiLM layer
  const char* tape_next;
  const char* dt_base;      // global (uniform): d theta dump of (tile32, layer)
  const char* film_base;    // global (uniform): FiLM sums of (tile16, layer)
  unsigned toff;            // lane offset inside a (tile32, layer) dump block: the register-dump position of this lane
  unsigned ttoff;           // lane offset of the tape DMA: toff, or (16-bit tape) doff
  unsigned doff;            // bf16 dump: lane offset inside an n-block's 2 KiB = 1024 (tile & 1) + 16 lane
  unsigned foff;            // lane offset inside a FiLM-sum n-block
  bool b0, b1;              // lane & 1, lane & 2
};
struct EpiIn { float4 f, p, t; unsigned tu[2]; };   // t: fp32 tape values; tu (16-bit tape): the row tile's four phases, two per dword
struct EpiOut { f32x4 dt, dtt; unsigned pd[2], px[2]; };   // pd / px (bf16 dump): d theta and x = sin(2 pi theta) as bf16 pairs

template <int PF4, bool T16>
__device__ __forceinline__ EpiIn epi_read(const Sink& k, int nbp, int rt, int tbuf) {
  EpiIn q;
  q.f = *reinterpret_cast<const float4*>(k.film + 32 * nbp + 8 * rt);
  if (T16) {      // the lane's 16-byte piece of the n-block holds both row tiles: slots 4 rt .. 4 rt + 3 = 8 bytes
    const uint2 w = *reinterpret_cast<const uint2*>(k.tape_lane + (tbuf * 2) * 1024 + rt * 8);
    q.tu[0] = w.x; q.tu[1] = w.y;
  } else {
    q.p = *reinterpret_cast<const float4*>(k.film + PF4 + 32 * nbp + 8 * rt);
    q.t = *reinterpret_cast<const float4*>(k.tape_lane + (tbuf * 2 + rt) * 1024);
  }
  return q;
}

// E(rt) of n-block nbp: d theta = dx cos(2 pi theta), d z = d theta f'' 2 pi split into bf16 (hi = truncation, lo = the
// remainder rounded to nearest) -> slots 4 rt .. 4 rt + 3 of k32-step nbp of the next stage's B operand; the d theta store.
// BD (bf16 dump, header of this file): also x = sin(2 pi theta) -- bitwise the forward's activation -- and both rounded to bf16.
template <bool BD, bool T16, bool S1>
__device__ __forceinline__ EpiOut epi_compute(const f32x4& acc, const EpiIn& q, u32x4& yh, u32x4& yl, int rt) {
  [[maybe_unused]] const float TWO_PI = 6.28318530717958647692f;
  const float f[4] = {q.f.x, q.f.y, q.f.z, q.f.w};
  EpiOut o;
  if (!S1) o.dtt = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned hb[4];
  float rem[4];
  [[maybe_unused]] float xs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float th, tr = 0.f;
    if (T16) {    // frac(theta) as 16-bit fixed point: already reduced to [0, 1) revolutions
      const unsigned w = q.tu[r >> 1];
      th = (float)((r & 1) ? (w >> 16) : (w & 0xffffu)) * (1.f / 65536.f);
    } else {
      const float p[4] = {q.p.x, q.p.y, q.p.z, q.p.w}, t[4] = {q.t.x, q.t.y, q.t.z, q.t.w};
      tr = t[r];
      th = rev_reduce(__builtin_fmaf(f[r], tr, p[r]));      // fenerf_trig.h: the forward's reduction, bit for bit
    }
    const float dt = acc[r] * cos_rev_reduced(th);
    if (BD) xs[r] = sin_rev_reduced(th);
    o.dt[r] = dt;
    if (S1) o.dtt[r] = dt * tr;        // (!S1: not formed -- the frequency gradient comes from the weight-gradient partial sums)
    const float dz = dt * (f[r] * TWO_PI);
    hb[r] = __builtin_bit_cast(unsigned, dz);
    rem[r] = dz - __builtin_bit_cast(float, hb[r] & 0xffff0000u);
  }
  if (BD) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x2 dd = {o.dt[2 * j], o.dt[2 * j + 1]}, xx = {xs[2 * j], xs[2 * j + 1]};
      o.pd[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(dd, bf16x2));     // v_cvt_pk_bf16_f32: round to nearest even
      o.px[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(xx, bf16x2));
    }
  }
  unsigned h2[2], l2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    h2[j] = __builtin_amdgcn_perm(hb[2 * j + 1], hb[2 * j], 0x07060302u);
    const f32x2 rr = {rem[2 * j], rem[2 * j + 1]};
    l2[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, bf16x2));
  }
  // pinned here: without a use in this block the compiler sinks the epilogue behind the stage (fenerf_siren_f16w.hip)
  asm volatile("" : "+v"(h2[0]), "+v"(h2[1]), "+v"(l2[0]), "+v"(l2[1]));
  yh[2 * rt] = h2[0]; yh[2 * rt + 1] = h2[1];
  yl[2 * rt] = l2[0]; yl[2 * rt + 1] = l2[1];
  return o;
}

template <int H, bool GRID, bool WGS, bool BD, int TAPE>
__global__ __launch_bounds__(512, 2) void siren_bwd16w_kernel(SirenBwdParams P, int n_geo, int n_color, int n_lab, int C) {
  constexpr bool T16 = TAPE == 1;       // FENERF_TAPE_U16: 16-bit phases
  constexpr bool S1 = TAPE == 0;        // FENERF_TAPE_F32: the kernel forms sum_p d theta * tape (FiLM-only launches need it)
  constexpr int NB = H / 32, KS = H / 32;                       // 32-row n-blocks; k32-steps of an H-wide input
  constexpr int QB = pad_pf16(2 * (H / 16)) / CH;               // chunks per square body
  constexpr int C0_QB = pad_pf16(2 * (H / 16 + 2)) / CH;        // colour-layer-0 body: + one k32-step of head rows
  constexpr int FILM_F = H * 4 < 1024 ? 1024 : H * 4;           // LDS-DMA moves whole KiBs
  constexpr int FILM_BYTES = 2 * FILM_F;
  constexpr int TL = H * 128;                                   // bytes of one (tile32, layer) dump block
  constexpr int TLT = T16 ? H * 64 : H * 128;                   // ... of the tape
  // one barrier per two chunk steps where every stage has an even number of steps (H >= 128)
  constexpr bool B2 = (NB * QB) % 2 == 0 && (NB * C0_QB) % 2 == 0 && (!GRID || QB % 2 == 0);
  extern __shared__ __attribute__((aligned(16))) float4 smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int L = n_geo + n_color;
  // LDS: [ring NSLOT x 8 KiB][film: 8 waves x 2 buffers x (f'' | p')][tape staging: 8 waves x 4 KiB][head^T (rgb) NB KiB][head B operands 8 x 2 KiB]
  char* lds = reinterpret_cast<char*>(smem);
  char* ring = lds;
  char* film_base = lds + NSLOT * CH * 1024 + wave * (2 * FILM_BYTES);
  char* tape_stage = lds + NSLOT * CH * 1024 + NWAVE * 2 * FILM_BYTES + wave * 4096;
  float* ht_lds = reinterpret_cast<float*>(lds + NSLOT * CH * 1024 + NWAVE * 2 * FILM_BYTES + NWAVE * 4096);
  float4* ext_wave = reinterpret_cast<float4*>(ht_lds + NB * 256) + wave * 128;
  f32x2* fsum = reinterpret_cast<f32x2*>(reinterpret_cast<float4*>(ht_lds + NB * 256) + NWAVE * 128);   // WGS: [3 buffers][8 waves][rt][16 slots]
  float* pts_wave = reinterpret_cast<float*>(fsum + (WGS ? 3 * NWAVE * 32 : 0)) + wave * 48;             // fused grid scatter: this tile's 16 points

  for (int i = threadIdx.x; i < NB * 256; i += 512) ht_lds[i] = P.stream[i];
  wait_vmcnt<0>();
  __syncthreads();

  // ---- the DMA's re-tiling permutation (fenerf_siren_f16w.hip header): this wave fetches operand (spl, rt, hl) of every chunk
  WStream ws;
  {
    const int spl = wave >> 2, rt = (wave >> 1) & 1, hl = wave & 1;
    const int gi = n >> 2, r = n & 3;
    const int row = 16 * (gi >> 1) + 4 * (gi & 1) + 8 * rt + r;
    const int e_old = 2 * (2 * spl + (g >> 1)) + hl;
    ws.voff = e_old * 1024 + ((g & 1) * 32 + row) * 16;
  }
  const long long nchunk = (long long)(n_color - 1 + n_geo - 1) * NB * QB + (long long)NB * C0_QB + (GRID ? QB : 0);
  ws.g_begin = reinterpret_cast<unsigned long long>(P.stream + P.ring_offset_floats);
  ws.g_end = ws.g_begin + (unsigned long long)nchunk * (CH * 1024);
  ws.g_next = ws.g_begin;
  ws.ring_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring) + wave * 1024);
  ws.ring_lane = ring + lane * 16;
  ws.early = wave < NWAVE / 2;
  ws.cs = 0;

  // ---- prime the shared stream: chunks 0..D-1 in flight
#pragma unroll
  for (int i = 0; i < DPF; ++i) ws_issue(ws, i);
#if FENERF_CHAIN_SETPRIO
  // static issue priority for the second-dispatched half, as in the forward kernel (fenerf_siren_f16w.hip): between the two waves of a
  // SIMD the arbiter prefers the older one, so waves 4-7 arrive last at every barrier
  if (wave >= NWAVE / 2) __builtin_amdgcn_s_setprio(1);
#endif

  // work split: octs of 16-point tiles (one tile per wave), XCD-contiguous ranges
  const long long ntiles = (P.P + 15) / 16;
  const long long nocts = (ntiles + NWAVE - 1) / NWAVE;
  const int nblk = gridDim.x;
  const int nx = nblk < 8 ? nblk : 8;
  const int xcd = blockIdx.x % nx, bi = blockIdx.x / nx;
  const int blocks_in_x = nblk / nx + (xcd < nblk % nx ? 1 : 0);
  const long long o_begin = nocts * xcd / nx, o_end = nocts * (xcd + 1) / nx;

  // fenerf_siren_backward_film (inversion: only the FiLM sums are wanted): no d(theta) dump, no d(grid features)
  // (TAPE = 2 is only ever launched with a dump -- launch_siren_backward16w below sends FiLM-only launches to the TAPE = 0 kernel -- so the
  // default generator step carries no branch around its d(theta) stores: round 6)
  const bool dump = TAPE == 2 ? true : P.d_t != nullptr;
  int tpar = 0;   // parity of the tape staging buffers (advances per stage when NB is odd)
  for (long long oct = o_begin + bi; oct < o_end; oct += blocks_in_x) {
    // a wave past the last tile repeats the last tile: same loads, same values, same stores (no guard in the stream loop)
    long long tile = oct * NWAVE + wave;
    const float vf = tile < ntiles ? 1.f : 0.f;     // WGS: a wave that repeats the last tile adds nothing to the oct's sums
    if (tile >= ntiles) tile = ntiles - 1;
    const long long tile32 = tile >> 1;
    const long long pt = tile * 16 + n;        // P.P is a multiple of 32 (fenerf_siren_backward)
    const long long img = __builtin_amdgcn_readfirstlane((int)((tile * 16) / P.pts_per_image));
    const float* fp_img = P.fp + (size_t)img * L * H;
    const float* pp_img = P.pp + (size_t)img * L * H;
    const char* tape_tile = uniform_ptr(reinterpret_cast<const char*>(P.tape) + (size_t)tile32 * L * TLT);
    const char* dt_tile = uniform_ptr(reinterpret_cast<const char*>(P.d_t) + (size_t)tile32 * L * TL);
    const char* film_tile = uniform_ptr(reinterpret_cast<const char*>(P.film_tiles) + (size_t)(WGS ? oct : tile) * L * (2 * H * 4));
    // ---- WGS: per-wave sums -> LDS buffer kb; one step of barriers later the wave whose turn it is combines and stores them
    // this lane's slot of its wave's block in buffer 0, once per tile (round 6: the five address instructions per row tile it replaces were
    // 10 of a body's ~180 VALU instructions); buffer and row tile are uniform / immediate offsets
    int fs_idx;      // (an index, not a pointer: an opaque pointer would lose its address space and turn the ds_write into a flat store)
    {
      const int lo = opaque(lane);
      fs_idx = opaque(wave * 32 + (lo >> 4) * 4 + (lo & 3));
    }
    auto fs_write = [&](int kb, int rt, const f32x2& sm) {
      const f32x2 v = {sm[0] * vf, sm[1] * vf};
      fsum[fs_idx + kb * (NWAVE * 32) + rt * 16] = v;
    };
    auto fs_combine = [&](int kb, int who, const char* dst) {
      if (wave == who) {
        const int lo = opaque(lane);
        if (lo < 32) {
          const f32x2* src = fsum + kb * NWAVE * 32 + lo;
          f32x2 a = src[0];
#pragma unroll
          for (int w = 1; w < NWAVE; ++w) { const f32x2 b = src[w * 32]; a[0] += b[0]; a[1] += b[1]; }
          st_f2(dst, (unsigned)lo * 8, a);
        }
      }
    };
    // n-blocks whose sums sit in LDS, oldest first (at most two), and the rotating buffer / combiner-wave counters
    const char* pend_dst[2] = {nullptr, nullptr};
    int pend_kb[2] = {0, 0}, pend_who[2] = {0, 0}, npend = 0, kb_next = 0, who_next = 0;
    auto fs_push = [&](const char* dst) {
      pend_dst[npend] = dst; pend_kb[npend] = kb_next; pend_who[npend] = who_next;
      npend += 1;
      kb_next = kb_next == 2 ? 0 : kb_next + 1;
      who_next = (who_next + 1) & 7;
    };
    auto fs_pop = [&]() {
      if (npend > 0) {
        fs_combine(pend_kb[0], pend_who[0], pend_dst[0]);
        pend_dst[0] = pend_dst[1]; pend_kb[0] = pend_kb[1]; pend_who[0] = pend_who[1];
        npend -= 1;
      }
    };

    auto film_issue = [&](int layer) {   // f'' (H floats) then p' (H floats) of `layer` into buffer layer & 1
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(film_base) + (layer & 1) * FILM_BYTES);
      const float* gf = fp_img + (size_t)layer * H;
      const float* gp = pp_img + (size_t)layer * H;
      const unsigned vo = (unsigned)opaque(lane) * 16;
      for (int off = 0; off < H * 4; off += 1024) {
        glds_1k_s(reinterpret_cast<const char*>(gf) + off, vo, dst + off);
        glds_1k_s(reinterpret_cast<const char*>(gp) + off, vo, dst + FILM_F + off);
      }
    };
    auto film_lane = [&](int layer) -> const float* {
      const int gq = opaque(lane) >> 4;
      return reinterpret_cast<const float*>(film_base + (layer & 1) * FILM_BYTES) + 16 * (gq >> 1) + 4 * (gq & 1);
    };
    // register-dump position of this lane inside a (tile32, layer) block: float4 index (4 nb + 2 (g >> 1) + rt) * 64 + 32 (g & 1) + 16 (tile & 1) + n
    auto lane_toff = [&]() -> unsigned {
      const int lo = opaque(lane);
      const int nn = lo & 15, gq = lo >> 4;
      return (unsigned)(((2 * (gq >> 1)) * 64 + (gq & 1) * 32 + 16 * (int)(tile & 1) + nn) * 16);
    };
    // tape block (nb, rt) of the (tile32, layer) dump at `base` -> staging buffer.  `base` is in SGPRs long before (make_sink): an
    // SGPR written by v_readfirstlane must not feed a vector-memory address within 5 wait states, and nothing checks an asm.
    // 16-bit tape: the n-block's 2-KiB block holds both 16-point tiles' pieces; this wave's KiB (both row tiles) in one DMA, by E(1) only
    // (step_loads above: the staging buffer it lands in is read by E(0) AND E(1) of this body first)
    auto tape_issue = [&](const char* base, int nb, int rt, int tbuf, unsigned toff) {
      if constexpr (T16) {
        if (rt == 1) {
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(tape_stage) + (tbuf * 2) * 1024);
          glds_1k_s_nt(base + nb * 2048, toff, dst);
        }
      } else {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(tape_stage) + (tbuf * 2 + rt) * 1024);
        glds_1k_s_nt(base + (nb * 4 + rt) * 1024, toff, dst);
      }
    };
    auto make_sink = [&](int layer) -> Sink {
      Sink k;
      k.film = film_lane(layer);
      k.tape_lane = tape_stage + opaque(lane) * 16;
      k.tape_base = uniform_ptr(tape_tile + (size_t)layer * TLT);
      k.tape_next = uniform_ptr(tape_tile + (size_t)(layer > 0 ? layer - 1 : 0) * TLT);
      k.dt_base = uniform_ptr(dt_tile + (size_t)layer * TL);
      k.film_base = uniform_ptr(film_tile + (size_t)layer * (2 * H * 4));
      k.toff = lane_toff();
      const int lo = opaque(lane);
      k.doff = (unsigned)(1024 * (int)(tile & 1) + 16 * lo);
      k.ttoff = T16 ? k.doff : k.toff;
      k.foff = (unsigned)(((lo >> 4) * 4 + (lo & 3)) * 8);
      k.b0 = (lo & 1) != 0; k.b1 = (lo & 2) != 0;
      return k;
    };

    // bf16 dump of n-block nbp: both row tiles' d theta in ONE 16-byte store per lane (slot 4 rt + r), and their x likewise into the
    // second half of the (tile32, layer) block
    auto dump_bf16 = [&](const Sink& k, int nbp, const EpiOut (&e2)[2], bool with_x) {
      const u32x4 dd = {e2[0].pd[0], e2[0].pd[1], e2[1].pd[0], e2[1].pd[1]};
      st_u4_nt(k.dt_base + nbp * 2048, k.doff, dd);
      if (with_x) {
        const u32x4 xx = {e2[0].px[0], e2[0].px[1], e2[1].px[0], e2[1].px[1]};
        st_u4_nt(k.dt_base + TL / 2 + nbp * 2048, k.doff, xx);
      }
    };

    film_issue(L - 1);
    film_issue(L - 2);

    // ---------------- gradient wrt the head rows (labels, sigma) as the B operand of the head k32-step: lane (n, kg) slot t = row 8 kg + t
    {
      unsigned hb[8];
      float rem[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = 8 * g + t;
        const int ch = row < n_lab ? row : (row == n_lab ? C - 1 : -1);
        const float v = ch >= 0 ? P.d_out[pt * C + ch] : 0.f;
        hb[t] = __builtin_bit_cast(unsigned, v);
        rem[t] = v - __builtin_bit_cast(float, hb[t] & 0xffff0000u);
      }
      u32x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hi[j] = __builtin_amdgcn_perm(hb[2 * j + 1], hb[2 * j], 0x07060302u);
        const f32x2 rr = {rem[2 * j], rem[2 * j + 1]};
        lo[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, bf16x2));
      }
      float4* ext = ext_wave + lane;
      ext[0] = __builtin_bit_cast(float4, hi);
      ext[64] = __builtin_bit_cast(float4, lo);
    }
    if (GRID && P.d_grid_cl) {   // the tile's points, for the scatter of d(grid features) behind the colour-layer-0 stage
      if (lane < 48) pts_wave[lane] = P.points[tile * 48 + lane];
    }
    // ---------------- rgb head: d(pre-sigmoid) = d_rgb s (1 - s) as the B operand of the fp32 MFMA (k = r, g, b, 0)
    float brgb;
    {
      float dpre[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float s = P.out[pt * C + (C - 4) + c];
        dpre[c] = P.d_out[pt * C + (C - 4) + c] * (s * (1.f - s));
      }
      brgb = g == 0 ? dpre[0] : (g == 1 ? dpre[1] : (g == 2 ? dpre[2] : 0.f));
    }
    // the tape of layer L - 1 (all n-blocks) for the rgb stage's own epilogue: plain loads, once per tile
    float4 t_rgb[NB][T16 ? 1 : 2];      // 16-bit tape: one 16-byte piece per n-block (both row tiles)
    {
      if constexpr (T16) {
        const float4* tp = reinterpret_cast<const float4*>(tape_tile + (size_t)(L - 1) * TLT + 1024 * (int)(tile & 1) + 16 * opaque(lane));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) t_rgb[nb][0] = tp[nb * 128];
      } else {
        const float4* tp = reinterpret_cast<const float4*>(tape_tile + (size_t)(L - 1) * TLT + lane_toff());
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) t_rgb[nb][T16 ? 0 : rt] = tp[(nb * 4 + rt) * 64];
      }
    }
    // tape blocks the first pipelined stage (FiLM layer L - 2) finds in flight: n-block 0 (n-block 1 is issued by its body 0)
    {
      const char* tb = uniform_ptr(tape_tile + (size_t)(L - 2) * TLT);
      const unsigned to0 = T16 ? (unsigned)(1024 * (int)(tile & 1) + 16 * opaque(lane)) : lane_toff();
      asm volatile("s_nop 4" ::: "memory");
      tape_issue(tb, 0, 0, tpar & 1, to0);
      tape_issue(tb, 0, 1, tpar & 1, to0);      // (16-bit tape: this one is the DMA)
    }
    wait_vmcnt<0>();      // once per tile: film L-1 / L-2, the prologue loads, and the ring prefetch have landed
    __builtin_amdgcn_s_barrier();   // ... in every wave: ring chunks 0 .. D-1 of this tile are visible
    LDS_FENCE();

    u32x4 zh[KS], zl[KS];
    // ---------------- rgb head^T on the exact fp32 MFMA (16x16x4: k = r, g, b, 0) -> d theta_{L-1}, d z_{L-1} ----------------
    {
      const Sink k = make_sink(L - 1);
      const int gi = n >> 2, r = n & 3;
      const float* wl = ht_lds + ((g & 1) * 32 + 16 * (gi >> 1) + 4 * (gi & 1) + r) * 4 + (g >> 1);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        EpiOut o2[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          const f32x4 acc = MFMA32W(wl[nb * 256 + 8 * rt * 4], brgb, z4);
          EpiIn q;
          q.f = *reinterpret_cast<const float4*>(k.film + 32 * nb + 8 * rt);
          if constexpr (T16) {
            const uint4 w = __builtin_bit_cast(uint4, t_rgb[nb][0]);
            q.tu[0] = rt ? w.z : w.x; q.tu[1] = rt ? w.w : w.y;
          } else {
            q.p = *reinterpret_cast<const float4*>(k.film + FILM_F / 4 + 32 * nb + 8 * rt);
            q.t = t_rgb[nb][T16 ? 0 : rt];
          }
          o2[rt] = epi_compute<BD, T16, S1>(acc, q, zh[nb], zl[nb], rt);
          if (!BD && dump) st_f4_nt(k.dt_base + (nb * 4 + rt) * 1024, k.toff, o2[rt].dt);
          const f32x2 s = {row_sum4(o2[rt].dt, k.b0, k.b1), S1 ? row_sum4(o2[rt].dtt, k.b0, k.b1) : 0.f};
          if (WGS) fs_write(nb % 3, rt, s);
          else st_f2(k.film_base + nb * 256 + rt * 128, k.foff, s);
        }
        if (BD) dump_bf16(k, nb, o2, false);    // x of the last FiLM layer has no consumer in the dump (the rgb-head job reads the tape)
        if (WGS) {   // once per tile and n-block: every wave's sums of this n-block are in buffer nb % 3; buffer reuse is three barriers away
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes have completed before it arrives
          __builtin_amdgcn_s_barrier();
          LDS_FENCE();
          fs_combine(nb % 3, nb & 7, k.film_base + nb * 256);
        }
      }
      kb_next = NB % 3; who_next = NB & 7;
    }
    AK a_cur = ws_read(ws, ws.cs, 0);

    // Fused grid scatter (P.d_grid_cl != nullptr): the colour-layer-0 stage parks the tile's d(grid features) [16 points][32 channels]
    // in the head-operand area of LDS; one point pair (2 points x 32 channels = two whole 128-B lines per atomic instruction, 8
    // corners) is scattered per body of the following stage, so that the 64 atomic instructions of a tile never sit as one burst in
    // front of a ring wait (stores and atomics are not counted, they only make a wait stricter).  The scatter costs the kernel 50 us
    // per 131,072-point launch either way (201 M float atomics per step contend with its own HBM streams); the separate scatter
    // kernel it replaces took 115 us per such chunk plus the d_e round trip.
    int scat_next = 8;       // next point pair to scatter (8 = none pending)
    auto scatter_pairs = [&](int count) {
      const int lq = opaque(lane);
      const float* eblk = reinterpret_cast<const float*>(ext_wave);
      const int ch = lq & 31;
      const float gwf = (float)(P.gw - 1), ghf = (float)(P.gh - 1), gdf = (float)(P.gd - 1);
      for (int it = 0; it < count && scat_next < 8; ++it, ++scat_next) {
        const int pi = 2 * scat_next + (lq >> 5);
        const float gv = eblk[pi * 32 + ch];
        const float qx = pts_wave[pi * 3 + 0] * P.box_scale, qy = pts_wave[pi * 3 + 1] * P.box_scale, qz = pts_wave[pi * 3 + 2] * P.box_scale;
        const float ix = ((qx + 1.f) / 2.f) * gwf, iy = ((qy + 1.f) / 2.f) * ghf, iz = ((qz + 1.f) / 2.f) * gdf;
        const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int cz = c >> 2, cy = (c >> 1) & 1, cx = c & 1;
          const float xi = x0 + cx, yi = y0 + cy, zi = z0 + cz;
          const float wx = cx ? (ix - x0) : (x0 + 1.f - ix);
          const float wy = cy ? (iy - y0) : (y0 + 1.f - iy);
          const float wz = cz ? (iz - z0) : (z0 + 1.f - iz);
          const bool ok = xi >= 0.f && xi <= gwf && yi >= 0.f && yi <= ghf && zi >= 0.f && zi <= gdf;
          if (ok) {
            const long long vox = ((long long)(int)zi * P.gh + (int)yi) * P.gw + (int)xi;
            unsafeAtomicAdd(P.d_grid_cl + vox * 32 + ch, gv * (wx * wy * wz));
          }
        }
      }
    };

    // One stage: NBODY bodies of QBS chunks; bop(sp, bh, bl) supplies the B operand of k32-step sp (false = padding).  With
    // EPI the bodies carry the epilogue of FiLM layer `lo` (tape layer lo, d theta / FiLM sums of layer lo) into y.
    // ph_c (round 6): one compile-time copy of a stage per wave half -- PH = 0: waves 0-3 (ring DMA at the top of a chunk step), 1: waves
    // 4-7 (half a step later) -- instead of testing the run-time flag ws.early twice per k32-step (two scalar branches, one taken, in front
    // of every MFMA burst); the branch now sits between stages.  (fenerf_siren_f16w.hip: the same change, - 6 % shader cycles there.)
    auto run_stage = [&](auto ph_c, auto nbody_c, auto qbs_c, auto epi_c, int lo, auto bop, u32x4 (&yh)[KS], u32x4 (&yl)[KS], f32x4 (&acc_last)[2]) {
      constexpr int NBODY = decltype(nbody_c)::value, QBS = decltype(qbs_c)::value;
      constexpr bool EPI = decltype(epi_c)::value;
      constexpr int PH = decltype(ph_c)::value;             // -1 (FENERF_WAVE_HALF_COPIES = 0, A/B builds): the run-time flag, as in rounds 3-5
      const bool early = PH < 0 ? ws.early : PH == 0;
      const Sink k = make_sink(lo);
      f32x4 acc_prev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      EpiOut eo[2];
      static_for<0, NBODY>([&](auto nb_c) {
        constexpr int nb = decltype(nb_c)::value;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        static_for<0, QBS>([&](auto qc_c) {
          constexpr int qc = decltype(qc_c)::value;
          constexpr int s = nb * QBS + qc;
          // ---- top of the step: chunk s + 1 (B2: and s + 2) visible to every wave, then (waves 0-3) the DMA of chunk s + D.  The
          // DMA overwrites the slot of chunk s - 2, whose last reads every wave issued before the barrier (B2: of step s or s - 1)
          if constexpr (!B2) {
            wait_vmcnt<EPI ? ring_wait(QBS, s, T16) : DPF - 2>();
            __builtin_amdgcn_s_barrier();
          } else if constexpr ((s & 1) == 0) {
            wait_vmcnt<EPI ? ring_wait2(QBS, s, T16) : DPF - 3>();
            __builtin_amdgcn_s_barrier();
          }
          LDS_FENCE();
          if (early) ws_issue(ws, (ws.cs + DPF) & 7);
          // operands of the items of this chunk: FiLM parameters and tape from LDS, read before the A operands (LDS returns in order)
          EpiIn q[2];
          if constexpr (EPI && nb > 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
              if (item_chunk(QBS, 2 * rt) == qc) {
                // fp32 tape: the block was issued 2 QB steps ago at the same item position, with 2 QB loads behind it -- for 2 QB >= DPF that is
                // implied by the ring waits since.  16-bit tape: issued by E(1) of the body two back, i.e. at least QB + (QB - chunk of item 2)
                // steps ago (a longer colour-layer-0 body in between only adds), with at least the QB ring DMAs + 1 tape DMA of the body in
                // between behind it.
                if constexpr (!T16) { if constexpr (2 * QB < DPF) wait_vmcnt<2 * QB>(); }
                else { if constexpr (2 * QB - item_chunk(QB, 2) < DPF) wait_vmcnt<QB + 1>(); }
                q[rt] = epi_read<FILM_F / 4, T16>(k, nb - 1, rt, ((nb - 1) + tpar) & 1);
              }
          }
#pragma unroll
          for (int spl = 0; spl < 2; ++spl) {
            if (spl == 1 && !early) ws_issue(ws, (ws.cs + DPF) & 7);
            const AK nn = spl == 0 ? ws_read(ws, ws.cs, 1) : ws_read(ws, (ws.cs + 1) & 7, 0);
            __builtin_amdgcn_sched_barrier(0);   // the reads stay at the top of the k32-step
            bf16x8 bh, bl;
            if (bop(2 * qc + spl, bh, bl)) kstep_mfma(acc, a_cur, bh, bl);
            if constexpr (EPI) {
              if (spl == 1) {
                // WGS: the oldest n-block whose sums sit in LDS was completed (second B item or a stage-trailing epilogue) at least one
                // workgroup barrier ago -- bodies are >= 1 step, barriers at most 2 steps apart, this is the body's second chunk
                // (first of a one-chunk body), in front of the items that may complete another
                if constexpr (WGS && qc == (QBS > 1 ? 1 : 0)) fs_pop();
                if constexpr (GRID && qc == QBS - 1) {
                  if (scat_next < 8) scatter_pairs((8 + NBODY - 1) / NBODY);
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                  if (item_chunk(QBS, it) != qc) continue;
                  const int rt = it >> 1;
                  if ((it & 1) == 0) {
                    if constexpr (nb > 0) {
                      eo[rt] = epi_compute<BD, T16, S1>(acc_prev[rt], q[rt], yh[nb - 1], yl[nb - 1], rt);
                      if constexpr (!BD) { if (dump && !(FENERF_EXP_CHAIN_ABLATE & 2)) st_f4_nt(k.dt_base + ((nb - 1) * 4 + rt) * 1024, k.toff, eo[rt].dt); }
                      else if (rt == 1) dump_bf16(k, nb - 1, eo, true);
                    }
                    // the tape block two n-blocks ahead of its use: (lo, nb + 1), or the next stage's n-block 0; the last
                    // stage's last body re-fetches (0, 0) so that ring_wait's count holds
                    if constexpr (FENERF_EXP_CHAIN_ABLATE & 4) tape_issue(reinterpret_cast<const char*>(P.tape), 0, rt, ((nb + 1) + tpar) & 1, k.ttoff);
                    else if constexpr (nb + 1 < NBODY) tape_issue(k.tape_base, nb + 1, rt, ((nb + 1) + tpar) & 1, k.ttoff);
                    else tape_issue(k.tape_next, 0, rt, (NBODY + tpar) & 1, k.ttoff);
                  } else if constexpr (!(FENERF_EXP_CHAIN_ABLATE & 1)) {
                    if constexpr (nb > 0) {
                      const f32x2 sm = {row_sum4(eo[rt].dt, k.b0, k.b1), S1 ? row_sum4(eo[rt].dtt, k.b0, k.b1) : 0.f};
                      if (WGS) {
                        fs_write(kb_next, rt, sm);
                        if (rt == 1) fs_push(k.film_base + (nb - 1) * 256);
                      } else {
                        st_f2(k.film_base + (nb - 1) * 256 + rt * 128, k.foff, sm);
                      }
                    }
                  }
                }
              }
            }
            if constexpr (EPI) {
              if (spl == 1) {   // the items' VALU work between the six MFMAs of this k32-step rather than behind them (- 1 %)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                  __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                  __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                }
              }
            }
            a_cur = nn;
            __builtin_amdgcn_sched_barrier(0);   // keep every k32-step's MFMAs / epilogue items where they are written
          }
          ws.cs = (ws.cs + 1) & 7;
        });
        acc_prev[0] = acc[0]; acc_prev[1] = acc[1];
      });
      if constexpr (EPI) {
        // ---- the last n-block's epilogue, behind the stage
        EpiOut o2[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          constexpr int c0 = item_chunk(QBS, 0), c1 = item_chunk(QBS, 2);
          // (16-bit tape: the block was issued by E(1) of the body before the last: the rt = 1 count holds for both reads)
          if (rt == 0 && !T16) wait_vmcnt<2 * QBS - 1 - c0>(); else wait_vmcnt<2 * QBS - 1 - c1>();
          LDS_FENCE();
          const EpiIn q = epi_read<FILM_F / 4, T16>(k, NBODY - 1, rt, ((NBODY - 1) + tpar) & 1);
          o2[rt] = epi_compute<BD, T16, S1>(acc_prev[rt], q, yh[NBODY - 1], yl[NBODY - 1], rt);
          const EpiOut& o = o2[rt];
          if (!BD) { if (dump) st_f4_nt(k.dt_base + ((NBODY - 1) * 4 + rt) * 1024, k.toff, o.dt); }
          else if (rt == 1) dump_bf16(k, NBODY - 1, o2, true);
          const f32x2 sm = {row_sum4(o.dt, k.b0, k.b1), S1 ? row_sum4(o.dtt, k.b0, k.b1) : 0.f};
          if (WGS) {
            fs_write(kb_next, rt, sm);
            if (rt == 1) {
              fs_push(k.film_base + (NBODY - 1) * 256);
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // complete before the next stage's first barrier (in the bodies, later LDS reads are waited for first)
            }
          } else {
            st_f2(k.film_base + (NBODY - 1) * 256 + rt * 128, k.foff, sm);
          }
        }
      } else {
        acc_last[0] = acc_prev[0]; acc_last[1] = acc_prev[1];
      }
    };

    // ---------------- FiLM layers L-2 .. 0: colour layers, colour layer 0 (+ heads, + grid-feature gradient), trunk ----------------
    auto film_layer = [&](auto ph_c, int lo) {
      u32x4 yh[KS], yl[KS];
      f32x4 unused[2];
      if (lo >= 1) film_issue(lo - 1);
      if (NB * QB < DPF + 2) wait_vmcnt<0>();
      auto bop = [&](int sp, bf16x8& bh, bf16x8& bl) -> bool {
        if (sp < KS) { bh = as_bf16x8(zh[sp]); bl = as_bf16x8(zl[sp]); return true; }
        return false;
      };
      if (lo == n_geo - 1) {
        const float4* ext = ext_wave + opaque(lane);
        auto bop0 = [&](int sp, bf16x8& bh, bf16x8& bl) -> bool {
          if (sp < KS) { bh = as_bf16x8(zh[sp]); bl = as_bf16x8(zl[sp]); return true; }
          if (sp == KS) { bh = as_bf16x8(ext[0]); bl = as_bf16x8(ext[64]); return true; }
          return false;
        };
        run_stage(ph_c, std::integral_constant<int, NB>{}, std::integral_constant<int, C0_QB>{}, std::true_type{}, lo, bop0, yh, yl, unused);
        if (GRID) {
          // d(grid features) = W_c0[:, grid]^T dz_{n_geo}: one body on the stage's input, no epilogue
          f32x4 ge[2];
          run_stage(ph_c, std::integral_constant<int, 1>{}, std::integral_constant<int, QB>{}, std::false_type{}, lo, bop, yh, yl, ge);
          const int lq = opaque(lane);
          if (P.d_grid_cl) {
            // Fused scatter (grid_backward_kernel's arithmetic): the tile's [16 points][32 channels] block goes through the head-operand
            // area of LDS (free since the last colour-layer-0 body).  A wave that repeats the last tile must not scatter twice.
            float* eblk = reinterpret_cast<float*>(ext_wave);
            float4* ew = reinterpret_cast<float4*>(eblk + (lq & 15) * 32 + 16 * (lq >> 5) + 4 * ((lq >> 4) & 1));
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              const f32x4 v = ge[rt];
              ew[2 * rt] = make_float4(v[0], v[1], v[2], v[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (vf != 0.f) scat_next = 0;     // scattered by the bodies of the next stage (run_stage) / the tile's end
            __builtin_amdgcn_wave_barrier();
          } else if (P.d_e) {
            float* ep = P.d_e + pt * 32 + 16 * (lq >> 5) + 4 * ((lq >> 4) & 1);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              const f32x4 v = ge[rt];
              *reinterpret_cast<float4*>(ep + 8 * rt) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
      } else {
        run_stage(ph_c, std::integral_constant<int, NB>{}, std::integral_constant<int, QB>{}, std::true_type{}, lo, bop, yh, yl, unused);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) { zh[s] = yh[s]; zl[s] = yl[s]; }
      if (NB & 1) tpar ^= 1;
    };
#pragma unroll 1
    for (int lo = L - 2; lo >= 0; --lo) {
#if FENERF_WAVE_HALF_COPIES
      if (ws.early) film_layer(std::integral_constant<int, 0>{}, lo); else film_layer(std::integral_constant<int, 1>{}, lo);
#else
      film_layer(std::integral_constant<int, -1>{}, lo);
#endif
    }
    if (GRID) scatter_pairs(8);   // whatever no later stage picked up (a model without trunk layers behind colour layer 0)
    if (WGS) {   // the last two n-blocks' sums
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      LDS_FENCE();
      fs_pop();
      fs_pop();
    }
    wait_vmcnt<0>();   // stores may retire out of order with the DMA loads: keep them out of the counted waits
    __builtin_amdgcn_wave_barrier();
  }
  wait_vmcnt<0>();     // no LDS-DMA may land after the workgroup has released its LDS
  __builtin_amdgcn_s_barrier();
}

static int hip_fail16w(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return FENERF_E_HIP;
}

template <int H, bool GRID, bool WGS, bool BD>
static int launch_w(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
  constexpr int TAPE = FENERF_BW16_T16;
  const size_t film_f = H * 4 < 1024 ? 1024 : H * 4;
  const size_t lds = (size_t)NSLOT * CH * 1024 + (size_t)NWAVE * 2 * (2 * film_f) + (size_t)NWAVE * 4096 + (size_t)(H / 32) * 1024 +
                     (size_t)NWAVE * 2048 + (WGS ? (size_t)3 * NWAVE * 32 * 8 : 0) + (size_t)NWAVE * 48 * 4;   // ring + FiLM buffers + tape staging + rgb head^T + head B operands + FiLM-sum buffers + tile points
  auto kfn = siren_bwd16w_kernel<H, GRID, WGS, BD, TAPE>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  const long long ntiles = (p.P + 15) / 16;
  long long blocks = (ntiles + NWAVE - 1) / NWAVE;
  if (blocks > launch_cus(m)) blocks = launch_cus(m);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, p, m->n_geo, m->n_color, m->n_lab, m->C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : hip_fail16w(e, "siren bf16 backward (16-point waves) launch");
}
template <int H, bool GRID>
static int launch_t(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
  const bool wgs = bwd16w_film_unit(p.P, p.pts_per_image) == 128;
  if (p.bf16_dump) return wgs ? launch_w<H, GRID, true, true>(m, p, stream) : launch_w<H, GRID, false, true>(m, p, stream);
  return wgs ? launch_w<H, GRID, true, false>(m, p, stream) : launch_w<H, GRID, false, false>(m, p, stream);
}

}  // namespace bw16

}  // namespace fenerf

#if !FENERF_BW16_T16
// Test hook (not part of include/fenerf.h): the compile-time wait counts of the chain kernel's stream loop and the schedule they are
// derived from, so that tests/test_bwd16w_wait_counts.py can replay the in-order load queue on the CPU and check that no wait ever
// allows more loads in flight than were issued behind the chunk it waits for.  what: 0 = ring_wait(qb, step) (a barrier every step),
// 1 = ring_wait2(qb, step) (a barrier every second step), 2 = tape DMAs issued in chunk `step` of a body of qb chunks, 3 = DPF;
// 4 / 5 / 6 = the same three for the 16-bit tape (one tape DMA per body).
extern "C" int fenerf_internal_bwd16w_schedule(int what, int qb, int step) {
  using namespace fenerf::bw16;
  switch (what) {
    case 0: return ring_wait(qb, step);
    case 1: return ring_wait2(qb, step);
    case 2: return step_loads(qb, step);
    case 3: return DPF;
    case 4: return ring_wait(qb, step, true);
    case 5: return ring_wait2(qb, step, true);
    case 6: return step_loads(qb, step, true);
  }
  return -1;
}
#endif

namespace fenerf {
#if !FENERF_BW16_T16
// points per FiLM-sum unit of siren_bwd16w_kernel: the workgroup's 128 when an oct cannot straddle images, else the wave's 16
int bwd16w_film_unit(long long total_points, long long pts_per_image) {
  return (total_points == pts_per_image || pts_per_image % 128 == 0) ? 128 : 16;
}
int launch_siren_backward16w_t16(const FenerfModel* m, const SirenBwdParams& p, void* stream);   // fenerf_siren_bwd16w_t16.hip
int launch_siren_backward16w_w(const FenerfModel* m, const SirenBwdParams& p, void* stream);     // fenerf_siren_bwd16w_w.hip
#endif

#if FENERF_BW16_T16 == 1
int launch_siren_backward16w_t16(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
#elif FENERF_BW16_T16 == 2
int launch_siren_backward16w_w(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
#else
int launch_siren_backward16w(const FenerfModel* m, const SirenBwdParams& p, void* stream) {
  if (p.tape_format == FENERF_TAPE_U16) return launch_siren_backward16w_t16(m, p, stream);
  if (p.tape_format == FENERF_TAPE_F32_W && p.d_t) return launch_siren_backward16w_w(m, p, stream);     // (FiLM-only launches need the second sum)
#endif
  if (p.P <= 0) return FENERF_OK;
  const bool g = m->grid_ch != 0;
  switch (m->H) {
    case 32: return g ? bw16::launch_t<32, true>(m, p, stream) : bw16::launch_t<32, false>(m, p, stream);
    case 64: return g ? bw16::launch_t<64, true>(m, p, stream) : bw16::launch_t<64, false>(m, p, stream);
    case 96: return g ? bw16::launch_t<96, true>(m, p, stream) : bw16::launch_t<96, false>(m, p, stream);
    case 128: return g ? bw16::launch_t<128, true>(m, p, stream) : bw16::launch_t<128, false>(m, p, stream);
    case 192: return g ? bw16::launch_t<192, true>(m, p, stream) : bw16::launch_t<192, false>(m, p, stream);
    case 256: return g ? bw16::launch_t<256, true>(m, p, stream) : bw16::launch_t<256, false>(m, p, stream);
  }
  set_error("unsupported hidden_dim");
  return FENERF_E_UNSUPPORTED;
}

}  // namespace fenerf
