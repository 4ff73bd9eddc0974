// Micro-benchmark: the persistent kernel's per-tile arithmetic (update_tile<float,4,4,5>) in isolation -- no
// global or LDS traffic -- at one and two waves per SIMD.  Tells how much of the measured ~1500 cycles per
// 64-row tile and SIMD is the VALU work itself.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I muscle_synergies_amd/csrc tools/ubench/tile_rate.hip -o /tmp/tile_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "nmf_kernels.hpp"
using namespace hipnmf;

template <int K>
__global__ void __launch_bounds__(512) k_tile(float* out, int iters, float seed) {
  constexpr int G = 4, CH = 4;
  using C = Cfg<float, G, CH, K>;
  MatAddr<float, G, CH, K> ma(out, 1024, out, 1024, 1 << 20, 16);
  float h[K][CH], hht[K][K], accA[K][CH], accB[C::NB];
  for (int c = 0; c < K; ++c) {
    for (int cc = 0; cc < CH; ++cc) { h[c][cc] = seed + 0.01f * (c + cc + (threadIdx.x & 3)); accA[c][cc] = 0.f; }
    for (int c2 = 0; c2 < K; ++c2) hht[c][c2] = uniform(seed * (1.f + 0.1f * (c + c2)));
  }
  for (int i = 0; i < C::NB; ++i) accB[i] = 0.f;
  RowTile<float, G, CH, K> t;
  for (int cc = 0; cc < CH; ++cc)
    for (int r = 0; r < G; ++r) t.x[cc][r] = seed * 0.5f + 0.001f * (threadIdx.x + cc + r);
  for (int c = 0; c < K; ++c) t.w[c] = seed + 0.002f * (threadIdx.x + c);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc)
#pragma unroll
      for (int r = 0; r < G; ++r) asm volatile("" : "+v"(t.x[cc][r]));  // opaque: no hoisting of X H^T
    update_tile<float, G, CH, K>(t, ma, h, hht, accA, accB, 0.f, 0.f, true);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int c = 0; c < K; ++c) { s += t.w[c]; for (int cc = 0; cc < CH; ++cc) s += accA[c][cc]; }
  for (int i = 0; i < C::NB; ++i) s += accB[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int threads : {256, 512}) {
    k_tile<5><<<256, threads>>>(d, 100, 0.7f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_tile<5><<<256, threads>>>(d, iters, 0.7f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = (double)iters * threads / 64 / 4;
    printf("update_tile<float,4,4,5>, %d waves/SIMD: %.3f ms, %.1f ns per tile and SIMD (%.0f cycles at 2.0 GHz)\n",
           threads / 256, ms, ms * 1e6 / tiles_per_simd, ms * 1e6 / tiles_per_simd * 2.0);
  }
  return 0;
}
