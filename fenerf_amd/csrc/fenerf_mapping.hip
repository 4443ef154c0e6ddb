// CustomMappingNetwork (siren/siren.py:82-102): z -> Linear, LeakyReLU(0.2) x (n_blocks + 1) -> Linear; the two halves of the output are
// the raw FiLM frequencies and phase shifts of one image.  Per image it is five matrix-vector products (z_dim -> 256 -> 256 -> 256 -> 256
// -> 2 n H): as PyTorch ops that is ~18 launches forward and ~60 backward for the two networks of DoubleImplicitGenerator3d, each a 5-us
// kernel at batch 1-6 -- 0.6 ms of a 13-ms generator step spent on 3 MFLOP (profiles/r04_ddp_step_timeline_*.txt).  Here: ONE launch
// forward, THREE backward per network, exact fp32 FMAs (fenerf_mapping_forward / fenerf_mapping_backward).  Used for small batches only
// (the 10,000-latent batch of generate_avg_frequencies, generators.py:530-543, stays a rocBLAS GEMM in PyTorch).
//
// forward   grid (B, S): every workgroup carries image b through the trunk (256-wide layers, activations in LDS; 16 waves, each wave
//           sixteen output rows at a time, lanes over the input: coalesced weight rows, shuffle reduction) and computes rows [s, s + 1) * out / S of the
//           last layer; workgroup s = 0 also stores the post-activation vectors the backward needs.  The trunk is recomputed S times
//           (3 x 65 k MACs) so that the 1 M MACs of the last layer spread over S workgroups.
// backward  (1) head_dx: partial[s][b][i] = sum_{j in range s} W_last[j][i] d_out[b][j]      grid (B, S), thread = input feature i
//           (2) deltas:  delta_last-1 = (sum_s partial) * lrelu'(act), then down the trunk: delta_{l-1}[i] = (sum_j W_l[j][i] delta_l[j])
//                        * lrelu'(act_{l-1}[i]); all deltas stored                            grid (B), thread = feature
//           (3) wgrad:   dW_l[j][i] = sum_b delta_l[b][j] x_{l-1}[b][i], db_l[j] = sum_b delta_l[b][j] for every layer in one launch
//                        (fixed summation order over b: deterministic)                       grid (row tiles, layers)
// LeakyReLU'(x) is taken from the sign of the stored POST-activation value (slope 0.2 > 0 keeps the sign; 0 -> slope, like torch).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "fenerf_internal.h"

namespace fenerf {

namespace {

constexpr int MAP_MAX_LAYERS = FENERF_MAP_MAX_LAYERS;
constexpr int MAP_THREADS = 512;           // 8 waves (256 registers each: the 16 x float4 row tile of matvec_rows must not spill)
constexpr int MAP_ROWS = 16;               // rows a wave has in flight at once (matvec_rows): 16 waves x 16 rows = a 256-row layer in ONE round of loads
constexpr int MAP_JSPLIT = MAP_THREADS / 256;
constexpr float LRELU_SLOPE = 0.2f;

struct MapParams {
  int n_layers;                 // linear layers (n_blocks + 2)
  int B, z_dim, hidden, out_dim, S;
  const float* W[MAP_MAX_LAYERS];
  const float* b[MAP_MAX_LAYERS];
  float* dW[MAP_MAX_LAYERS];
  float* db[MAP_MAX_LAYERS];
  const float* z;               // [B][z_dim]
  float* acts;                  // [n_layers - 1][B][hidden] post-activation outputs of the trunk layers
  float* out;                   // [B][out_dim]
  const float* d_out;           // [B][out_dim]
  float* partial;               // [S][B][hidden]
  float* delta;                 // [n_layers - 1][B][hidden] dL/d(pre-activation) of the trunk layers
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * LRELU_SLOPE; }

// The weights are cold every time (13 ms of other kernels' traffic since their last use), and a layer cannot start before the layer in
// front of it has finished: reading them layer by layer pays one HBM miss latency per round of loads, ~10 rounds per kernel.  So each
// kernel first touches every 128-byte line it is going to read -- all those misses overlap -- and the layers then run out of L2.
__device__ __forceinline__ float warm_lines(const float* __restrict__ p, size_t n_floats, float sink) {
  for (size_t off = (size_t)threadIdx.x * 32; off < n_floats; off += (size_t)MAP_THREADS * 32) sink += p[off];
  return sink;
}
__device__ __forceinline__ void warm_done(float sink, float* somewhere) {
  if (sink == 1.2345e30f) *somewhere = sink;      // never true: keeps the loads alive
}

// y[r] = sum_i W[r][i] x[i] + b[r] for rows [r0, r1).  A wave takes MAP_ROWS rows at a time; a lane reads 16 bytes of each row (float4:
// 64 lanes x 4 = a whole 256-wide row in ONE load per row), so a 256-row, 256-wide layer is one round of 16 loads per lane on 16 waves --
// a layer costs one memory latency, not sixteen.  (n_in not a multiple of 4: scalar loads, four rows at a time.)  Summation order per
// row: lane-strided partial sums, then a fixed butterfly over the lanes (deterministic).
template <class F>
__device__ __forceinline__ void matvec_rows(const float* __restrict__ W, const float* __restrict__ bias, const float* x, int n_in, int r0, int r1, F store) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = MAP_THREADS / 64;
  if ((n_in & 3) == 0) {
    for (int r = r0 + wave * MAP_ROWS; r < r1; r += NW * MAP_ROWS) {
      float acc[MAP_ROWS];
#pragma unroll
      for (int k = 0; k < MAP_ROWS; ++k) acc[k] = 0.f;
      const int nr = r1 - r < MAP_ROWS ? r1 - r : MAP_ROWS;      // wave-uniform
      const int myrow = lane >> 2;                               // the row whose total the transposing butterfly below leaves in this lane
      const float bl = myrow < nr ? bias[r + myrow] : 0.f;       // its bias, fetched with the weights (not behind the reduction)
      for (int c = 4 * lane; c < n_in; c += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(x + c);
        float4 wv[MAP_ROWS];
#pragma unroll
        for (int k = 0; k < MAP_ROWS; ++k)
          wv[k] = k < nr ? *reinterpret_cast<const float4*>(W + (size_t)(r + k) * n_in + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < MAP_ROWS; ++k)
          acc[k] = __builtin_fmaf(wv[k].w, xv.w, __builtin_fmaf(wv[k].z, xv.z, __builtin_fmaf(wv[k].y, xv.y, __builtin_fmaf(wv[k].x, xv.x, acc[k]))));
      }
      // 16 rows x 64 lane-partials -> 16 totals with 17 shuffles instead of 96: a transposing butterfly -- at each step a lane keeps the
      // half of its rows that its lane bit selects and adds the partner's partials of those rows; row k's total ends in lanes 4 k .. 4 k + 3
      static_assert(MAP_ROWS == 16, "the butterfly below is written for 16 rows");
      float a8[8], a4[4], a2[2];
      const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
      for (int k = 0; k < 8; ++k) a8[k] = (b5 ? acc[k + 8] : acc[k]) + __shfl_xor(b5 ? acc[k] : acc[k + 8], 32, 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) a4[k] = (b4 ? a8[k + 4] : a8[k]) + __shfl_xor(b4 ? a8[k] : a8[k + 4], 16, 64);
#pragma unroll
      for (int k = 0; k < 2; ++k) a2[k] = (b3 ? a4[k + 2] : a4[k]) + __shfl_xor(b3 ? a4[k] : a4[k + 2], 8, 64);
      float res = (b2 ? a2[1] : a2[0]) + __shfl_xor(b2 ? a2[0] : a2[1], 4, 64);
      res += __shfl_xor(res, 2, 64);
      res += __shfl_xor(res, 1, 64);
      if ((lane & 3) == 0 && myrow < nr) store(r + myrow, res + bl);
    }
    return;
  }
  constexpr int RS = 4;
  for (int r = r0 + wave * RS; r < r1; r += NW * RS) {
    float acc[RS];
#pragma unroll
    for (int k = 0; k < RS; ++k) acc[k] = 0.f;
    for (int i = lane; i < n_in; i += 64) {
      const float xi = x[i];
#pragma unroll
      for (int k = 0; k < RS; ++k)
        if (r + k < r1) acc[k] = __builtin_fmaf(W[(size_t)(r + k) * n_in + i], xi, acc[k]);
    }
    const float bl = (lane < RS && r + lane < r1) ? bias[r + lane] : 0.f;
    float res = 0.f;
#pragma unroll
    for (int k = 0; k < RS; ++k) {
      const float v = wave_sum_f(acc[k]);
      if (lane == k) res = v;
    }
    if (lane < RS && r + lane < r1) store(r + lane, res + bl);
  }
}

__global__ __launch_bounds__(MAP_THREADS) void mapping_forward_kernel(MapParams P) {
  extern __shared__ float lds[];          // two activation buffers of max(z_dim, hidden) floats, each a whole number of float4
  const int b = blockIdx.x, s = blockIdx.y;
  const int wmax = ((P.z_dim > P.hidden ? P.z_dim : P.hidden) + 3) & ~3;   // the buffers swap roles: both must take matvec_rows' 16-byte reads
  float* x = lds;
  float* y = lds + wmax;
  for (int i = threadIdx.x; i < P.z_dim; i += MAP_THREADS) x[i] = P.z[(size_t)b * P.z_dim + i];
  {
    float sink = 0.f;
    for (int l = 0; l + 1 < P.n_layers; ++l) sink = warm_lines(P.W[l], (size_t)P.hidden * (l == 0 ? P.z_dim : P.hidden), sink);
    const int w0 = (int)((long long)P.out_dim * s / P.S), w1 = (int)((long long)P.out_dim * (s + 1) / P.S);
    sink = warm_lines(P.W[P.n_layers - 1] + (size_t)w0 * P.hidden, (size_t)(w1 - w0) * P.hidden, sink);
    warm_done(sink, P.out);
  }
  __syncthreads();
  int n_in = P.z_dim;
  for (int l = 0; l + 1 < P.n_layers; ++l) {
    float* act = P.acts + ((size_t)l * P.B + b) * P.hidden;
    matvec_rows(P.W[l], P.b[l], x, n_in, 0, P.hidden, [&](int r, float v) {
      v = lrelu(v);
      y[r] = v;
      if (s == 0) act[r] = v;
    });
    __syncthreads();
    float* t = x; x = y; y = t;
    n_in = P.hidden;
  }
  const int L = P.n_layers - 1;
  const int r0 = (int)((long long)P.out_dim * s / P.S), r1 = (int)((long long)P.out_dim * (s + 1) / P.S);
  float* o = P.out + (size_t)b * P.out_dim;
  matvec_rows(P.W[L], P.b[L], x, n_in, r0, r1, [&](int r, float v) { o[r] = v; });
}

// dst[i] = sum_{j in [j0, j1)} W[j][i] g[j] for i < n (W row-major [.][n], g in memory or LDS): thread (i, q) sums the j of residue q
// modulo MAP_JSPLIT, eight loads in flight, then the MAP_JSPLIT partial sums of an i meet in LDS (fixed order: deterministic).
// red: MAP_THREADS floats of LDS.  Valid for n <= 256; larger n loops.  Ends with a __syncthreads().
template <class F>
__device__ __forceinline__ void matvec_cols(const float* __restrict__ W, const float* g, int n, int j0, int j1, float* red, F store) {
  for (int base = 0; base < n; base += 256) {
    const int i = base + (threadIdx.x & 255), q = threadIdx.x >> 8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (i < n) {
      int j = j0 + q;
      for (; j + 7 * MAP_JSPLIT < j1; j += 8 * MAP_JSPLIT) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_fmaf(W[(size_t)(j + k * MAP_JSPLIT) * n + i], g[j + k * MAP_JSPLIT], acc[k]);
      }
      for (; j < j1; j += MAP_JSPLIT) acc[0] = __builtin_fmaf(W[(size_t)j * n + i], g[j], acc[0]);
    }
    red[threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (q == 0 && i < n) {
      float v = red[threadIdx.x];
#pragma unroll
      for (int k = 1; k < MAP_JSPLIT; ++k) v += red[threadIdx.x + 256 * k];
      store(i, v);
    }
    __syncthreads();
  }
}

// (1) partial[s][b][i] = sum_{j in range s} W_last[j][i] d_out[b][j]
__global__ __launch_bounds__(MAP_THREADS) void mapping_head_dx_kernel(MapParams P) {
  __shared__ float red[MAP_THREADS];
  const int b = blockIdx.x, s = blockIdx.y;
  const int L = P.n_layers - 1;
  const int j0 = (int)((long long)P.out_dim * s / P.S), j1 = (int)((long long)P.out_dim * (s + 1) / P.S);
  float* dst = P.partial + ((size_t)s * P.B + b) * P.hidden;
  warm_done(warm_lines(P.W[L] + (size_t)j0 * P.hidden, (size_t)(j1 - j0) * P.hidden, 0.f), dst);
  matvec_cols(P.W[L], P.d_out + (size_t)b * P.out_dim, P.hidden, j0, j1, red, [&](int i, float v) { dst[i] = v; });
}

// (2) deltas of the trunk layers, last to first
__global__ __launch_bounds__(MAP_THREADS) void mapping_delta_kernel(MapParams P) {
  extern __shared__ float lds[];          // [hidden] delta of the layer above | [MAP_THREADS] reduction scratch
  float* dcur = lds;
  float* red = lds + P.hidden;
  const int b = blockIdx.x;
  const int T = P.n_layers - 1;           // trunk layers 0 .. T - 1
  {
    float sink = 0.f;
    for (int l = 1; l < T; ++l) sink = warm_lines(P.W[l], (size_t)P.hidden * P.hidden, sink);
    warm_done(sink, P.delta);
  }
  for (int i = threadIdx.x; i < P.hidden; i += MAP_THREADS) {
    float acc = 0.f;
    for (int s = 0; s < P.S; ++s) acc += P.partial[((size_t)s * P.B + b) * P.hidden + i];
    const float a = P.acts[((size_t)(T - 1) * P.B + b) * P.hidden + i];
    const float d = acc * (a > 0.f ? 1.f : LRELU_SLOPE);
    P.delta[((size_t)(T - 1) * P.B + b) * P.hidden + i] = d;
    dcur[i] = d;
  }
  __syncthreads();
  for (int l = T - 1; l >= 1; --l) {      // delta_{l-1} from delta_l through W_l [hidden][hidden]; results parked in the global delta
    float* dst = P.delta + ((size_t)(l - 1) * P.B + b) * P.hidden;      // array first (dcur is still being read), copied to LDS after
    const float* a = P.acts + ((size_t)(l - 1) * P.B + b) * P.hidden;
    matvec_cols(P.W[l], dcur, P.hidden, 0, P.hidden, red, [&](int i, float v) { dst[i] = v * (a[i] > 0.f ? 1.f : LRELU_SLOPE); });
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < P.hidden; i += MAP_THREADS) dcur[i] = dst[i];
    __syncthreads();
  }
}

// (3) dW_l[j][i] = sum_b delta_l[b][j] x_{l-1}[b][i];  db_l[j] = sum_b delta_l[b][j];  blockIdx.y = layer, one thread per (j, i)
__global__ __launch_bounds__(256) void mapping_wgrad_kernel(MapParams P) {
  const int l = blockIdx.y;
  const int T = P.n_layers - 1;
  const int n_out = l == T ? P.out_dim : P.hidden, n_in = l == 0 ? P.z_dim : P.hidden;
  const float* dl = l == T ? P.d_out : P.delta + (size_t)l * P.B * P.hidden;          // [B][n_out]
  const float* xin = l == 0 ? P.z : P.acts + (size_t)(l - 1) * P.B * P.hidden;        // [B][n_in]
  const long long total = (long long)n_out * n_in;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)gridDim.x * 256) {
    const int j = (int)(k / n_in), i = (int)(k % n_in);
    float acc = 0.f;
    for (int b = 0; b < P.B; ++b) acc = __builtin_fmaf(dl[(size_t)b * n_out + j], xin[(size_t)b * n_in + i], acc);
    P.dW[l][k] = acc;
    if (i == 0) {
      float sb = 0.f;
      for (int b = 0; b < P.B; ++b) sb += dl[(size_t)b * n_out + j];
      P.db[l][j] = sb;
    }
  }
}

int map_fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

int fill(MapParams& P, const FenerfMappingNet* net, int B) {
  if (!net) return map_fail(FENERF_E_INVALID, "mapping network description is NULL");
  if (net->n_layers < 2 || net->n_layers > MAP_MAX_LAYERS) return map_fail(FENERF_E_INVALID, "mapping network: n_layers must be in [2, FENERF_MAP_MAX_LAYERS]");
  if (B <= 0 || net->z_dim <= 0 || net->hidden <= 0 || net->out_dim <= 0) return map_fail(FENERF_E_INVALID, "mapping network: B, z_dim, hidden, out_dim must be > 0");
  if (net->hidden > 1024 || net->z_dim > 4096) return map_fail(FENERF_E_UNSUPPORTED, "mapping network: hidden <= 1024 and z_dim <= 4096");
  memset(&P, 0, sizeof(P));
  P.n_layers = net->n_layers; P.B = B; P.z_dim = net->z_dim; P.hidden = net->hidden; P.out_dim = net->out_dim;
  P.S = net->out_dim >= 2048 ? 16 : (net->out_dim >= 512 ? 8 : 1);
  for (int l = 0; l < net->n_layers; ++l) {
    if (!net->W[l] || !net->b[l]) return map_fail(FENERF_E_INVALID, "mapping network: weight / bias pointer is NULL");
    P.W[l] = net->W[l]; P.b[l] = net->b[l];
  }
  return FENERF_OK;
}

}  // namespace
}  // namespace fenerf

using namespace fenerf;

extern "C" size_t fenerf_mapping_workspace_floats(const FenerfMappingNet* net, int B) {
  if (!net || B <= 0) return 0;
  const size_t S = net->out_dim >= 2048 ? 16 : (net->out_dim >= 512 ? 8 : 1);
  return (S + (size_t)(net->n_layers - 1)) * (size_t)B * (size_t)net->hidden;        // partial [S][B][hidden] | delta [n_layers - 1][B][hidden]
}

extern "C" int fenerf_mapping_forward(const FenerfMappingNet* net, int B, const float* z, float* acts, float* out, void* stream) {
  MapParams P;
  int rc = fill(P, net, B);
  if (rc) return rc;
  if (!z || !acts || !out) return map_fail(FENERF_E_INVALID, "fenerf_mapping_forward: NULL pointer");
  P.z = z; P.acts = acts; P.out = out;
  const int wmax = ((P.z_dim > P.hidden ? P.z_dim : P.hidden) + 3) & ~3;     // as the kernel rounds it: two buffers of whole float4s
  PhaseScope ph(PH_OTHER, stream);
  hipLaunchKernelGGL(mapping_forward_kernel, dim3(B, P.S), dim3(MAP_THREADS), 2 * wmax * sizeof(float), (hipStream_t)stream, P);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : map_fail(FENERF_E_HIP, std::string("mapping forward launch: ") + hipGetErrorString(e));
}

extern "C" int fenerf_mapping_backward(const FenerfMappingNet* net, int B, const float* z, const float* acts, const float* d_out,
                                       float* const* dW, float* const* db, float* workspace, void* stream) {
  MapParams P;
  int rc = fill(P, net, B);
  if (rc) return rc;
  if (!z || !acts || !d_out || !dW || !db || !workspace) return map_fail(FENERF_E_INVALID, "fenerf_mapping_backward: NULL pointer");
  for (int l = 0; l < P.n_layers; ++l) {
    if (!dW[l] || !db[l]) return map_fail(FENERF_E_INVALID, "fenerf_mapping_backward: gradient pointer is NULL");
    P.dW[l] = dW[l]; P.db[l] = db[l];
  }
  P.z = z; P.acts = const_cast<float*>(acts); P.d_out = d_out;
  P.partial = workspace;
  P.delta = workspace + (size_t)P.S * B * P.hidden;
  hipStream_t st = (hipStream_t)stream;
  PhaseScope ph(PH_OTHER, stream);
  hipLaunchKernelGGL(mapping_head_dx_kernel, dim3(B, P.S), dim3(MAP_THREADS), 0, st, P);
  hipLaunchKernelGGL(mapping_delta_kernel, dim3(B), dim3(MAP_THREADS), (P.hidden + MAP_THREADS) * sizeof(float), st, P);
  const long long biggest = (long long)P.out_dim * P.hidden;
  int gx = (int)((biggest + 255) / 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(mapping_wgrad_kernel, dim3(gx, P.n_layers), dim3(256), 0, st, P);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : map_fail(FENERF_E_HIP, std::string("mapping backward launch: ") + hipGetErrorString(e));
}
