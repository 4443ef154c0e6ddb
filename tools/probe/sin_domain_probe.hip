// Hardware probe (not part of the product): what do v_sin_f32 / v_cos_f32 return on gfx950 for arguments of growing magnitude?
// The GCN3/Vega ISA manuals give a valid domain of [-256, +256] revolutions ("out of range input results in float 0"), and LLVM puts a
// v_fract_f32 in front of the instruction on the whole GFX9 family (FeatureTrigReducedRange).  This sweeps |t| in [2^k, 2^(k+1)) for
// k = -2 .. 30 and reports, per octave, the max abs error against double sin / cos of the SAME fp32 argument (exact reduction in
// double) of:   raw            v_sin_f32(t)
//               fract          v_sin_f32(v_fract_f32(t))                       (what fenerf_trig.h ships)
//               rndne          v_sin_f32(t - v_rndne_f32(t))                   (exact reduction, two instructions)
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/sin_domain_probe.hip -o tools/probe/sin_domain_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const float* t, float* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = t[i];
  const float fr = __builtin_amdgcn_fractf(x), rn = x - __builtin_rintf(x);
  o[0 * n + i] = __builtin_amdgcn_sinf(x);  o[1 * n + i] = __builtin_amdgcn_sinf(fr); o[2 * n + i] = __builtin_amdgcn_sinf(rn);
  o[3 * n + i] = __builtin_amdgcn_cosf(x);  o[4 * n + i] = __builtin_amdgcn_cosf(fr); o[5 * n + i] = __builtin_amdgcn_cosf(rn);
}

int main() {
  const int per = 1 << 16, K0 = -2, K1 = 30, noct = K1 - K0 + 1, n = per * noct;
  std::vector<float> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int k = 0; k < noct; ++k)
    for (int i = 0; i < per; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const double u = (double)(s >> 11) / 9007199254740992.0;
      const float mag = (float)ldexp(1.0 + u, K0 + k);
      h[k * per + i] = (i & 1) ? -mag : mag;
    }
  float *d_t, *d_o;
  hipMalloc(&d_t, n * 4); hipMalloc(&d_o, 6ull * n * 4);
  hipMemcpy(d_t, h.data(), n * 4, hipMemcpyHostToDevice);
  probe<<<(n + 255) / 256, 256>>>(d_t, d_o, n);
  std::vector<float> o(6ull * n);
  hipMemcpy(o.data(), d_o, 6ull * n * 4, hipMemcpyDeviceToHost);
  printf("# v_sin_f32 / v_cos_f32 on gfx950: max abs error per octave of |t| (revolutions) vs double sin / cos of the same fp32 argument\n");
  printf("# %-22s %10s %10s %10s | %10s %10s %10s | zeros returned by raw sin\n", "|t| in", "sin raw", "sin fract", "sin rndne", "cos raw", "cos fract", "cos rndne");
  for (int k = 0; k < noct; ++k) {
    double e[6] = {0, 0, 0, 0, 0, 0};
    int zeros = 0;
    for (int i = 0; i < per; ++i) {
      const int j = k * per + i;
      const double x = (double)h[j], r = x - nearbyint(x);
      const double rs = sin(2.0 * M_PI * r), rc = cos(2.0 * M_PI * r);
      for (int v = 0; v < 3; ++v) {
        e[v] = fmax(e[v], fabs((double)o[(size_t)v * n + j] - rs));
        e[3 + v] = fmax(e[3 + v], fabs((double)o[(size_t)(3 + v) * n + j] - rc));
      }
      zeros += (o[j] == 0.f && fabs(rs) > 1e-6);
    }
    printf("[2^%-3d, 2^%-3d) %8.3g  %10.2e %10.2e %10.2e | %10.2e %10.2e %10.2e | %d of %d\n", K0 + k, K0 + k + 1, ldexp(1.0, K0 + k), e[0], e[1], e[2], e[3], e[4],
           e[5], zeros, per);
  }
  return 0;
}
