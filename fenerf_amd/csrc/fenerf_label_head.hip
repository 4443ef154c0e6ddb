// fenerf_label_head_backward: the gradients of label_layer_linear's 2-3 Linear layers from the gradient of their fold.
//
// The render kernels evaluate the activation-free label head (siren.py:1490-1494) as ONE affine map, A = W_{n-1} ... W_0 (n_lab x H),
// c = sum_i S_i b_i with S_i = W_{n-1} ... W_{i+1}, and fenerf_siren_param_grads / fenerf_render_backward return dL/dA, dL/dc.  What the
// reference's autograd leaves in .grad of every layer is
//     db_i = S_i^T gc,     dW_i = S_i^T (U_i + gc (x) q_i),     U_i = gA W_0^T ... W_{i-1}^T,   q_i = W_{i-1} q_{i-1} + b_{i-1}  (q_0 = 0)
// -- skinny products throughout: one side of each has the n_lab (18) rows of the last layer.  Rounds 2-4 ran them as 11 rocBLAS / ATen
// launches at the end of every generator step (15-25 us each: a 256 x 256 x 18 product does not fill one CU's worth of a GEMM tile);
// here they are two launches of H x {2, 3} one-wave workgroups (one for the two-layer head).  fp32 FMAs, sums over k in lane-strided
// order + a wave reduction: the same class as the GEMMs they replace, another summation order (tests: <= 2e-6 of the tensor's scale).
#include <hip/hip_runtime.h>

#include <string>

#include "fenerf_internal.h"

namespace fenerf {
namespace {

constexpr int LH_MAX_ROWS = 32;

struct LabelHeadJob {
  int n, H, nl;
  const float* W[FENERF_MAX_LABEL_LAYERS];
  const float* b[FENERF_MAX_LABEL_LAYERS];
  const float* gA;   // [nl][H]
  const float* gc;   // [nl]
  float* dW[FENERF_MAX_LABEL_LAYERS];
  float* db[FENERF_MAX_LABEL_LAYERS];
  float* U1;         // [nl][H]   gA W_0^T                 (three layers)
  float* S1;         // [nl][H]   W_2 W_1                  (three layers)
  float* q2;         // [H]       W_1 b_0 + b_1            (three layers)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// out[r][j] = sum_k A[r][k] * w_j[k] (+ gc[r] * q[j]),  w_j[k] = W[j][k] (row j: ROW) or W[k][j] (column j)
template <bool ROW>
__device__ void skinny(const float* A, const float* W, int H, int nl, int j, const float* gc, const float* q, float* out) {
  const int lane = threadIdx.x;
  float acc[LH_MAX_ROWS];
#pragma unroll
  for (int r = 0; r < LH_MAX_ROWS; ++r) acc[r] = 0.f;
  for (int k = lane; k < H; k += 64) {
    const float w = ROW ? W[(size_t)j * H + k] : W[(size_t)k * H + j];
#pragma unroll
    for (int r = 0; r < LH_MAX_ROWS; ++r)
      if (r < nl) acc[r] = fmaf(A[(size_t)r * H + k], w, acc[r]);
  }
  const float qj = q ? q[j] : 0.f;
#pragma unroll
  for (int r = 0; r < LH_MAX_ROWS; ++r) {
    if (r < nl) {          // uniform
      const float s = wave_sum(acc[r]);
      if (lane == 0) out[(size_t)r * H + j] = gc ? fmaf(gc[r], qj, s) : s;
    }
  }
}

// db[j] = sum_r S[r][j] gc[r];   dW[j][m] = sum_r S[r][j] U[r][m] (+ db[j] * q[m])
__device__ void outer(const float* S, const float* U, const float* gc, const float* q, int H, int nl, int j, float* dW, float* db) {
  const int lane = threadIdx.x;
  float s[LH_MAX_ROWS];
  float d = 0.f;
#pragma unroll
  for (int r = 0; r < LH_MAX_ROWS; ++r) {
    s[r] = r < nl ? S[(size_t)r * H + j] : 0.f;
    if (r < nl) d = fmaf(s[r], gc[r], d);
  }
  if (lane == 0) db[j] = d;
  for (int m = lane; m < H; m += 64) {
    float a = q ? d * q[m] : 0.f;
#pragma unroll
    for (int r = 0; r < LH_MAX_ROWS; ++r)
      if (r < nl) a = fmaf(s[r], U[(size_t)r * H + m], a);
    dW[(size_t)j * H + m] = a;
  }
}

// launch 1.  two layers: everything (blockIdx.y 0: dW_1 = gA W_0^T + gc (x) b_0, db_1 = gc;  1: dW_0, db_0 with S_0 = W_1).
//            three layers: the intermediates (0: U1 = gA W_0^T;  1: S1 = W_2 W_1;  2: q2 = W_1 b_0 + b_1).
__global__ __launch_bounds__(64) void label_head_stage1_kernel(LabelHeadJob J) {
  const int j = blockIdx.x, lane = threadIdx.x;
  if (J.n == 2) {
    if (blockIdx.y == 0) {
      skinny<true>(J.gA, J.W[0], J.H, J.nl, j, J.gc, J.b[0], J.dW[1]);
      if (j == 0 && lane < J.nl) J.db[1][lane] = J.gc[lane];
    } else {
      outer(J.W[1], J.gA, J.gc, nullptr, J.H, J.nl, j, J.dW[0], J.db[0]);
    }
    return;
  }
  if (blockIdx.y == 0) skinny<true>(J.gA, J.W[0], J.H, J.nl, j, nullptr, nullptr, J.U1);
  else if (blockIdx.y == 1) skinny<false>(J.W[2], J.W[1], J.H, J.nl, j, nullptr, nullptr, J.S1);
  else {
    float a = 0.f;
    for (int k = lane; k < J.H; k += 64) a = fmaf(J.W[1][(size_t)j * J.H + k], J.b[0][k], a);
    a = wave_sum(a);
    if (lane == 0) J.q2[j] = a + J.b[1][j];
  }
}

// launch 2 (three layers).  0: dW_2 = U1 W_1^T + gc (x) q2, db_2 = gc;  1: dW_1, db_1 with S_1 = W_2, U_1, q_1 = b_0;  2: dW_0, db_0 with S1, gA
__global__ __launch_bounds__(64) void label_head_stage2_kernel(LabelHeadJob J) {
  const int j = blockIdx.x, lane = threadIdx.x;
  if (blockIdx.y == 0) {
    skinny<true>(J.U1, J.W[1], J.H, J.nl, j, J.gc, J.q2, J.dW[2]);
    if (j == 0 && lane < J.nl) J.db[2][lane] = J.gc[lane];
  } else if (blockIdx.y == 1) {
    outer(J.W[2], J.U1, J.gc, J.b[0], J.H, J.nl, j, J.dW[1], J.db[1]);
  } else {
    outer(J.S1, J.gA, J.gc, nullptr, J.H, J.nl, j, J.dW[0], J.db[0]);
  }
}

int lh_fail(int code, const std::string& msg) { set_error(msg); return code; }

}  // namespace
}  // namespace fenerf

using namespace fenerf;

extern "C" size_t fenerf_label_head_workspace_floats(int H) { return H > 0 ? (size_t)(2 * LH_MAX_ROWS + 1) * (size_t)H : 0; }

extern "C" int fenerf_label_head_backward(int n_layers, int H, int n_lab, const float* const* W, const float* const* b, const float* g_head_w,
                                          const float* g_head_b, float* const* dW, float* const* db, float* workspace, void* stream) {
  if (n_layers < 1 || n_layers > FENERF_MAX_LABEL_LAYERS) return lh_fail(FENERF_E_INVALID, "fenerf_label_head_backward: n_layers must be 1..3");
  if (H < 1 || n_lab < 1 || n_lab > LH_MAX_ROWS) return lh_fail(FENERF_E_INVALID, "fenerf_label_head_backward: need H >= 1 and 1 <= n_lab <= 32");
  if (!W || !b || !g_head_w || !g_head_b || !dW || !db || (n_layers == 3 && !workspace))
    return lh_fail(FENERF_E_INVALID, "fenerf_label_head_backward: NULL pointer");
  LabelHeadJob J{};
  J.n = n_layers; J.H = H; J.nl = n_lab; J.gA = g_head_w; J.gc = g_head_b;
  for (int i = 0; i < n_layers; ++i) {
    if (!W[i] || !b[i] || !dW[i] || !db[i]) return lh_fail(FENERF_E_INVALID, "fenerf_label_head_backward: a layer's pointer is NULL");
    J.W[i] = W[i]; J.b[i] = b[i]; J.dW[i] = dW[i]; J.db[i] = db[i];
  }
  hipStream_t st = (hipStream_t)stream;
  PhaseScope ph(PH_OTHER, stream);
  if (n_layers == 1) {      // the fold IS the layer
    hipError_t e = hipMemcpyAsync(dW[0], g_head_w, sizeof(float) * (size_t)n_lab * H, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(db[0], g_head_b, sizeof(float) * (size_t)n_lab, hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? FENERF_OK : lh_fail(FENERF_E_HIP, std::string("label head backward copy: ") + hipGetErrorString(e));
  }
  if (n_layers == 3) {
    J.U1 = workspace; J.S1 = workspace + (size_t)LH_MAX_ROWS * H; J.q2 = workspace + (size_t)2 * LH_MAX_ROWS * H;
  }
  hipLaunchKernelGGL(label_head_stage1_kernel, dim3(H, n_layers), dim3(64), 0, st, J);
  if (n_layers == 3) hipLaunchKernelGGL(label_head_stage2_kernel, dim3(H, 3), dim3(64), 0, st, J);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FENERF_OK : lh_fail(FENERF_E_HIP, std::string("label head backward launch: ") + hipGetErrorString(e));
}
