// What does ds_read_b64_tr_b16 return?  LDS is filled with u16 element indices; every lane supplies an 8-byte-aligned address and gets
// 4 u16 back.  Pattern A: lane l reads at byte 8 l (its "own" elements 4 l .. 4 l + 3 if there were no transpose).  Pattern B: the
// [4 rows][16 cols] sub-tile image of cdna_hip_programming.md (lane l: row (l & 15) >> 2 ... see code).  Prints result[lane][j].
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o tools/probe/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned short* out, int pattern, int stride_elems) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr_elems;
  if (pattern == 0) addr_elems = 4 * l;                                   // contiguous 8 B per lane
  else if (pattern == 1) addr_elems = (l & 15) * stride_elems + 4 * (l >> 4);   // lane n of group g: row n (stride), cols 4 g .. 4 g + 3
  else addr_elems = ((l & 15) >> 2) * stride_elems + 4 * (l & 3) + 16 * (l >> 4);  // group g: 4 rows x 16 cols block at col 16 g; lane m -> row m >> 2, cols 4 (m & 3)
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds + addr_elems * 2;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}

int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * 2);
  std::vector<unsigned short> h(256);
  const int pats[5][2] = {{0, 0}, {1, 64}, {1, 40}, {2, 64}, {2, 16}};
  for (auto& p : pats) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p[0], p[1]);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("pattern %d stride %d (element indices returned, lane: j0 j1 j2 j3)\n", p[0], p[1]);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 1) ? "\n" : "   |");
  }
  hipFree(d);
  return 0;
}
