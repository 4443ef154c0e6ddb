// Gradients wrt the SIREN's INPUTS -- sample positions and view directions -- from the d(theta) dump of the backward chain
// (fenerf_siren_input_grads, include/fenerf.h).  What torch autograd leaves in `input.grad` / `ray_directions.grad` of
// forward_with_frequencies_phase_shifts (siren.py:1509-1530) when a caller asks for them:
//     d input  = W_0^T dz_0  +  grid_sample's backward wrt its coordinates (sample_from_3dgrid, siren.py:314-330: trilinear, zeros
//                padding, align_corners=True) applied to d(shared_features) = W_c0[:, 3:3+G]^T dz_c0,        (:1513-1514, :1517-1520)
//     d points = d input * 2 / 0.24                                                                           (UniformBoxWarp, :181-187)
//     d dirs   = W_c0[:, 0:3]^T dz_c0                                                                         (the cat of :1522)
// with dz_l = dL/d(W_l x + b_l) = d theta_l * 2 pi f'_l / (GEMM result scale of the packed layer) -- the factor the weight-gradient
// reductions apply to the same dump (fenerf_siren_wgrad.hip, wgrad_reduce_thin_kernel).  The generator API never needs this (the
// reference computes its rays under torch.no_grad(), generators.py:465, :483); it exists for callers of the bare SIREN module.
// The chain kernels stay as they are: this is one extra HBM-bound pass over two layers of the dump (2 x 4 H bytes per point), exact
// fp32 FMAs, launched only when a caller asked for these gradients.
#include <hip/hip_runtime.h>

#include "fenerf_internal.h"
#include "fenerf_layout.h"
#include "fenerf_nt.h"

namespace fenerf {

struct InputGradParams {
  const float* d_t;        // fp32 d(theta) dump, [tile32][L][H/8][64 lanes][4]
  const float* fp;         // [B][L][H] f' (fenerf_film.h)
  const float* inv;        // [L][H] GEMM result scale (FENERF_PREC_F16X3) or nullptr
  const float* points;     // [P][3] as given to the forward (grid models)
  const float* w0;         // [H][3]   layer 0's nn.Linear weight
  const float* wc0;        // [H][ld]  colour layer 0's nn.Linear weight: columns [dirs 3 | grid features G | x H]
  int wc0_ld;
  const float* grid;       // channels-last [D][H][W][32] or nullptr
  int gd, gh, gw;
  float box_scale;
  long long ntiles, pts_per_image;
  int L, H, n_geo, G;
  float* d_points;         // [P][3] or nullptr
  float* d_dirs;           // [P][3] or nullptr
};

constexpr int IG_ROW = 40;     // floats per staged weight row: [w0 xyz, 0 | wc0 dirs xyz, 0 | wc0 grid features 32]
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(float z, float wx, float wy, v2f acc) { return __builtin_elementwise_fma(v2f{z, z}, v2f{wx, wy}, acc); }

// Workgroup = one image's tiles (blockIdx.y = image): the two weight blocks are staged ONCE, every row already multiplied by the image's
// dz factor (2 pi f' / GEMM result scale), so the inner loop is d theta * row.  A lane owns NP whole points: lanes 0-31 / 32-63 take two
// different 32-point tiles and fetch BOTH lane halves' 16-byte pieces of a feature group (8 consecutive features of the point), so
// every lane of the wave multiplies the same weight row (one broadcast LDS read per 4 weights and NP x 64 points), the sums need no
// cross-lane step, and the loop has no barrier.  Products are v_pk_fma_f32 pairs; the dump rows of the next feature group are in flight
// while the current one is multiplied.
template <int NP, bool GRID>
__global__ __launch_bounds__(256) void siren_input_grad_kernel(InputGradParams P) {
  extern __shared__ float lds[];
  const int H = P.H, L = P.L, G = GRID ? 32 : 0;
  float* Wt = lds;                        // [H][IG_ROW], scaled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, t = lane >> 5;
  const long long b = blockIdx.y;
  const float TWO_PI = 6.28318530717958647692f;
  for (int i = tid; i < H * IG_ROW; i += 256) {
    const int n = i / IG_ROW, c = i % IG_ROW;
    const int l = c < 4 ? 0 : P.n_geo;
    float v = 0.f;
    if (c < 3) v = P.w0[(size_t)n * 3 + c];
    else if (c >= 4 && c < 7) v = P.wc0[(size_t)n * P.wc0_ld + (c - 4)];
    else if (c >= 8 && c - 8 < G) v = P.wc0[(size_t)n * P.wc0_ld + 3 + (c - 8)];
    Wt[i] = v * (P.fp[((size_t)b * L + l) * H + n] * TWO_PI / (P.inv ? P.inv[(size_t)l * H + n] : 1.f));
  }
  __syncthreads();

  const long long tiles_img = P.pts_per_image / 32, tl = (long long)(H / 8) * 64;     // tl: float4s per (tile, layer)
  const long long units = (tiles_img + 2 * NP - 1) / (2 * NP);
  const float4* dt4 = reinterpret_cast<const float4*>(P.d_t);
  for (long long u = (long long)blockIdx.x * 4 + wave; u < units; u += (long long)gridDim.x * 4) {
    long long gt[NP];      // global tile of the lane's q-th point (clamped to the image's last tile when the unit runs past it)
    bool valid[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const long long ti = (u * NP + q) * 2 + t;
      valid[q] = ti < tiles_img;
      gt[q] = b * tiles_img + (valid[q] ? ti : tiles_img - 1);
    }
    v2f a0[NP][2], ad[NP][2], ae[NP][16];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      a0[q][0] = a0[q][1] = ad[q][0] = ad[q][1] = v2f{0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 16; ++c) ae[q][c] = v2f{0.f, 0.f};
    }
    // the dump rows of feature group g + 1 are in flight while group g is multiplied (two waves per SIMD do not cover an HBM round trip)
    const float4* p0[NP];
    const float4* pc[NP];
    float4 n0[NP][2], nc[NP][2];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      p0[q] = dt4 + (gt[q] * L + 0) * tl + m;
      pc[q] = dt4 + (gt[q] * L + P.n_geo) * tl + m;
      n0[q][0] = nt_load(p0[q]); n0[q][1] = nt_load(p0[q] + 32);
      nc[q][0] = nt_load(pc[q]); nc[q][1] = nt_load(pc[q] + 32);
    }
#pragma unroll 1
    for (int g = 0; g < H / 8; ++g) {
      float4 d0[NP][2], dc[NP][2];
#pragma unroll
      for (int q = 0; q < NP; ++q) { d0[q][0] = n0[q][0]; d0[q][1] = n0[q][1]; dc[q][0] = nc[q][0]; dc[q][1] = nc[q][1]; }
      if (g + 1 < H / 8) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          n0[q][0] = nt_load(p0[q] + (g + 1) * 64); n0[q][1] = nt_load(p0[q] + (g + 1) * 64 + 32);
          nc[q][0] = nt_load(pc[q] + (g + 1) * 64); nc[q][1] = nt_load(pc[q] + (g + 1) * 64 + 32);
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {       // feature 8 g + r = tape_feature(g, r >> 2, r & 3)
        const float4* wr = reinterpret_cast<const float4*>(Wt + (size_t)(8 * g + r) * IG_ROW);
        const float4 w0 = wr[0], wd = wr[1];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const float4 v0 = d0[q][r >> 2], vc = dc[q][r >> 2];
          const float z0 = (r & 3) == 0 ? v0.x : (r & 3) == 1 ? v0.y : (r & 3) == 2 ? v0.z : v0.w;
          const float zc = (r & 3) == 0 ? vc.x : (r & 3) == 1 ? vc.y : (r & 3) == 2 ? vc.z : vc.w;
          a0[q][0] = fma2(z0, w0.x, w0.y, a0[q][0]); a0[q][1] = fma2(z0, w0.z, w0.w, a0[q][1]);
          ad[q][0] = fma2(zc, wd.x, wd.y, ad[q][0]); ad[q][1] = fma2(zc, wd.z, wd.w, ad[q][1]);
        }
        if (GRID) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float4 we = wr[2 + k];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
              const float4 vc = dc[q][r >> 2];
              const float zc = (r & 3) == 0 ? vc.x : (r & 3) == 1 ? vc.y : (r & 3) == 2 ? vc.z : vc.w;
              ae[q][2 * k] = fma2(zc, we.x, we.y, ae[q][2 * k]); ae[q][2 * k + 1] = fma2(zc, we.z, we.w, ae[q][2 * k + 1]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const long long pt = gt[q] * 32 + m;
      float gx = 0.f, gy = 0.f, gz = 0.f;
      if (GRID && P.d_points) {      // grid_sample's backward wrt the coordinates: d ix = sum_corners (+-1) wy wz <d features, grid[corner]>
        const float qx = P.points[pt * 3 + 0] * P.box_scale, qy = P.points[pt * 3 + 1] * P.box_scale, qz = P.points[pt * 3 + 2] * P.box_scale;
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
          const bool ok = xi >= 0.f && xi <= (float)(P.gw - 1) && yi >= 0.f && yi <= (float)(P.gh - 1) && zi >= 0.f && zi <= (float)(P.gd - 1);
          if (ok) {
            const long long vox = ((long long)(int)zi * P.gh + (int)yi) * P.gw + (int)xi;
            const float4* gv = reinterpret_cast<const float4*>(P.grid + vox * 32);
            v2f s2 = v2f{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float4 v = gv[k];
              s2 = __builtin_elementwise_fma(ae[q][2 * k], v2f{v.x, v.y}, s2);
              s2 = __builtin_elementwise_fma(ae[q][2 * k + 1], v2f{v.z, v.w}, s2);
            }
            const float s = s2.x + s2.y;
            gx = fmaf(s, (cx ? 1.f : -1.f) * wy * wz, gx);
            gy = fmaf(s, wx * (cy ? 1.f : -1.f) * wz, gy);
            gz = fmaf(s, wx * wy * (cz ? 1.f : -1.f), gz);
          }
        }
        gx *= 0.5f * (float)(P.gw - 1); gy *= 0.5f * (float)(P.gh - 1); gz *= 0.5f * (float)(P.gd - 1);
      }
      if (valid[q]) {
        if (P.d_points) {
          P.d_points[pt * 3 + 0] = (a0[q][0].x + gx) * P.box_scale;
          P.d_points[pt * 3 + 1] = (a0[q][0].y + gy) * P.box_scale;
          P.d_points[pt * 3 + 2] = (a0[q][1].x + gz) * P.box_scale;
        }
        if (P.d_dirs) {
          P.d_dirs[pt * 3 + 0] = ad[q][0].x; P.d_dirs[pt * 3 + 1] = ad[q][0].y; P.d_dirs[pt * 3 + 2] = ad[q][1].x;
        }
      }
    }
  }
}

template <int NP, bool GRID>
static int launch_ig(const FenerfModel* m, int B, long long P, const InputGradParams& p, void* stream) {
  auto kfn = siren_input_grad_kernel<NP, GRID>;
  const size_t lds = (size_t)m->H * IG_ROW * sizeof(float);
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds)) return rc;
  // a workgroup's four waves walk units of 2 NP tiles of ONE image; enough workgroups per image to fill the device three deep
  const long long units = (P / 32 + 2 * NP - 1) / (2 * NP);
  long long bx = (units + 3) / 4, cap = (3LL * launch_cus(m) + B - 1) / B;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(kfn, dim3((unsigned)bx, (unsigned)B), dim3(256), lds, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("input gradient launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

int launch_siren_input_grads(const FenerfModel* m, int B, long long P, const float* points, const float* fp, const float* d_t, const float* w_geo0,
                             const float* w_color0, int w_color0_ld, float* d_points, float* d_dirs, void* stream) {
  InputGradParams p;
  p.d_t = d_t; p.fp = fp;
  p.inv = m->precision == FENERF_PREC_F16X3 ? m->d_consts + CONST_FILM_BIAS + (size_t)m->L * m->H : nullptr;
  p.points = points; p.w0 = w_geo0; p.wc0 = w_color0; p.wc0_ld = w_color0_ld;
  p.grid = m->grid_ch ? m->d_grid : nullptr; p.gd = m->gd; p.gh = m->gh; p.gw = m->gw; p.box_scale = m->box_scale;
  p.ntiles = (long long)B * P / 32; p.pts_per_image = P;
  p.L = m->L; p.H = m->H; p.n_geo = m->n_geo; p.G = m->grid_ch;
  p.d_points = d_points; p.d_dirs = d_dirs;
  // NP = 1: two points per lane halve the LDS reads per point, but at 222 registers leave two waves per SIMD where one point per lane
  // runs three -- measured 0.229 vs 0.213 ms at 393,216 points, 0.062 vs 0.056 ms at 65,536 (profiles/r06_input_grads_timing.txt)
  return m->grid_ch ? launch_ig<1, true>(m, B, P, p, stream) : launch_ig<1, false>(m, B, P, p, stream);
}

}  // namespace fenerf
