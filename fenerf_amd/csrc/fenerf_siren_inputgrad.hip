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

__global__ __launch_bounds__(256) void siren_input_grad_kernel(InputGradParams P) {
  extern __shared__ float lds[];
  const int H = P.H, L = P.L, G = P.G;
  float* Wt = lds;                        // [H][IG_ROW]
  float* sc = Wt + (size_t)H * IG_ROW;    // [2][H]: dz scale of layer 0 | colour layer 0 for the tile's image
  float* red = sc + 2 * H;                // [4 waves][IG_ROW][32]
  float* tot = red + 4 * IG_ROW * 32;     // [IG_ROW][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, half = lane >> 5;
  const float TWO_PI = 6.28318530717958647692f;

  for (int i = tid; i < H * IG_ROW; i += 256) {
    const int n = i / IG_ROW, c = i % IG_ROW;
    float v = 0.f;
    if (c < 3) v = P.w0[(size_t)n * 3 + c];
    else if (c >= 4 && c < 7) v = P.wc0[(size_t)n * P.wc0_ld + (c - 4)];
    else if (c >= 8 && c - 8 < G) v = P.wc0[(size_t)n * P.wc0_ld + 3 + (c - 8)];
    Wt[i] = v;
  }

  const long long tl = (long long)(H / 8) * 64;     // float4s per (tile, layer)
  for (long long tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    const long long b = tile * 32 / P.pts_per_image;
    __syncthreads();      // Wt staged (first trip) / the previous tile's readers of sc, red, tot are done
    for (int i = tid; i < 2 * H; i += 256) {
      const int l = i < H ? 0 : P.n_geo, n = i < H ? i : i - H;
      sc[i] = P.fp[((size_t)b * L + l) * H + n] * TWO_PI / (P.inv ? P.inv[(size_t)l * H + n] : 1.f);
    }
    __syncthreads();

    float a0[3] = {0.f, 0.f, 0.f}, ad[3] = {0.f, 0.f, 0.f}, ae[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) ae[c] = 0.f;
    const float4* d0p = reinterpret_cast<const float4*>(P.d_t) + (tile * L + 0) * tl;
    const float4* dcp = reinterpret_cast<const float4*>(P.d_t) + (tile * L + P.n_geo) * tl;
    for (int grp = wave; grp < H / 8; grp += 4) {
      const float4 d0 = nt_load(d0p + grp * 64 + lane), dc = nt_load(dcp + grp * 64 + lane);
      const int row = tape_feature(grp, half, 0);
      const float z0[4] = {d0.x * sc[row + 0], d0.y * sc[row + 1], d0.z * sc[row + 2], d0.w * sc[row + 3]};
      const float zc[4] = {dc.x * sc[H + row + 0], dc.y * sc[H + row + 1], dc.z * sc[H + row + 2], dc.w * sc[H + row + 3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4* wr = reinterpret_cast<const float4*>(Wt + (size_t)(row + i) * IG_ROW);
        const float4 w0 = wr[0], wd = wr[1];
        a0[0] = fmaf(z0[i], w0.x, a0[0]); a0[1] = fmaf(z0[i], w0.y, a0[1]); a0[2] = fmaf(z0[i], w0.z, a0[2]);
        ad[0] = fmaf(zc[i], wd.x, ad[0]); ad[1] = fmaf(zc[i], wd.y, ad[1]); ad[2] = fmaf(zc[i], wd.z, ad[2]);
        if (G) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float4 we = wr[2 + q];
            ae[4 * q + 0] = fmaf(zc[i], we.x, ae[4 * q + 0]); ae[4 * q + 1] = fmaf(zc[i], we.y, ae[4 * q + 1]);
            ae[4 * q + 2] = fmaf(zc[i], we.z, ae[4 * q + 2]); ae[4 * q + 3] = fmaf(zc[i], we.w, ae[4 * q + 3]);
          }
        }
      }
    }
    // the two lane halves hold the two halves of every feature group of the same point: fold, then one partial per wave
#pragma unroll
    for (int c = 0; c < 3; ++c) { a0[c] += __shfl_xor(a0[c], 32); ad[c] += __shfl_xor(ad[c], 32); }
#pragma unroll
    for (int c = 0; c < 32; ++c) ae[c] += __shfl_xor(ae[c], 32);
    if (half == 0) {
      float* r = red + (size_t)wave * IG_ROW * 32 + m;
#pragma unroll
      for (int c = 0; c < 3; ++c) { r[c * 32] = a0[c]; r[(4 + c) * 32] = ad[c]; }
      r[3 * 32] = 0.f; r[7 * 32] = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) r[(8 + c) * 32] = ae[c];
    }
    __syncthreads();
    for (int i = tid; i < IG_ROW * 32; i += 256)
      tot[i] = (red[i] + red[IG_ROW * 32 + i]) + (red[2 * IG_ROW * 32 + i] + red[3 * IG_ROW * 32 + i]);
    __syncthreads();

    // grid_sample's backward wrt the coordinates: thread = (corner, point); d ix = sum_corners (+-1) wy wz <d features, grid[corner]>
    float* gr = red;      // [8 corners][3][32], red is free again
    if (G && P.d_points) {
      const int corner = tid >> 5, pm = tid & 31;
      const long long pt = tile * 32 + pm;
      const float qx = P.points[pt * 3 + 0] * P.box_scale, qy = P.points[pt * 3 + 1] * P.box_scale, qz = P.points[pt * 3 + 2] * P.box_scale;
      const float ix = ((qx + 1.f) / 2.f) * (float)(P.gw - 1);
      const float iy = ((qy + 1.f) / 2.f) * (float)(P.gh - 1);
      const float iz = ((qz + 1.f) / 2.f) * (float)(P.gd - 1);
      const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
      const int cz = corner >> 2, cy = (corner >> 1) & 1, cx = corner & 1;
      const float xi = x0 + cx, yi = y0 + cy, zi = z0 + cz;
      const float wx = cx ? (ix - x0) : (x0 + 1.f - ix);
      const float wy = cy ? (iy - y0) : (y0 + 1.f - iy);
      const float wz = cz ? (iz - z0) : (z0 + 1.f - iz);
      const bool ok = xi >= 0.f && xi <= (float)(P.gw - 1) && yi >= 0.f && yi <= (float)(P.gh - 1) && zi >= 0.f && zi <= (float)(P.gd - 1);
      float s = 0.f;
      if (ok) {
        const long long vox = ((long long)(int)zi * P.gh + (int)yi) * P.gw + (int)xi;
        const float4* gv = reinterpret_cast<const float4*>(P.grid + vox * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 v = gv[q];
          s = fmaf(tot[(8 + 4 * q + 0) * 32 + pm], v.x, s); s = fmaf(tot[(8 + 4 * q + 1) * 32 + pm], v.y, s);
          s = fmaf(tot[(8 + 4 * q + 2) * 32 + pm], v.z, s); s = fmaf(tot[(8 + 4 * q + 3) * 32 + pm], v.w, s);
        }
      }
      gr[(corner * 3 + 0) * 32 + pm] = s * ((cx ? 1.f : -1.f) * wy * wz);
      gr[(corner * 3 + 1) * 32 + pm] = s * (wx * (cy ? 1.f : -1.f) * wz);
      gr[(corner * 3 + 2) * 32 + pm] = s * (wx * wy * (cz ? 1.f : -1.f));
    }
    __syncthreads();
    if (tid < 96) {
      const int k = tid >> 5, pm = tid & 31;
      const long long pt = tile * 32 + pm;
      if (P.d_points) {
        float g = 0.f;
        if (G) {
#pragma unroll
          for (int c = 0; c < 8; ++c) g += gr[(c * 3 + k) * 32 + pm];
          g *= 0.5f * (float)((k == 0 ? P.gw : k == 1 ? P.gh : P.gd) - 1);
        }
        P.d_points[pt * 3 + k] = (tot[k * 32 + pm] + g) * P.box_scale;
      }
      if (P.d_dirs) P.d_dirs[pt * 3 + k] = tot[(4 + k) * 32 + pm];
    }
  }
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
  const size_t lds = ((size_t)m->H * IG_ROW + 2 * (size_t)m->H + 5 * IG_ROW * 32) * sizeof(float);
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(siren_input_grad_kernel), lds)) return rc;
  long long blocks = 2LL * launch_cus(m);
  if (blocks > p.ntiles) blocks = p.ntiles;
  hipLaunchKernelGGL(siren_input_grad_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("input gradient launch: ") + hipGetErrorString(e)); return FENERF_E_HIP; }
  return FENERF_OK;
}

}  // namespace fenerf
