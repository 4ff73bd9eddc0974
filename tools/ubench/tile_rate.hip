// Micro-benchmark: the persistent kernel's per-tile arithmetic (update_tile<float,4,4,5>) in isolation -- no
// global or LDS traffic -- at one and two waves per SIMD.  Tells how much of the measured ~1500 cycles per
// 64-row tile and SIMD is the VALU work itself.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I muscle_synergies_amd/csrc tools/ubench/tile_rate.hip -o /tmp/tile_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "nmf_kernels.hpp"
using namespace hipnmf;

// local copy of update_tile with parts removable (MODE bits: 1 no reduce-scatter, 2 no group broadcast,
// 4 no quotient, 8 no W^T W, 16 no numerator FMAs, 32 no denominator FMAs, 64 no W^T X FMAs)
template <int MODE, int G, int CH, int K>
__device__ __forceinline__ void tile_variant(RowTile<float, G, CH, K>& t, const MatAddr<float, G, CH, K>& ma,
                                             const float (&h)[K][CH], const float (&hht)[K][K], float (&accA)[K][CH],
                                             float (&accB)[Cfg<float, G, CH, K>::NB]) {
  const int g = ma.g;
  float pn[G][K];
#pragma unroll
  for (int r = 0; r < G; ++r)
#pragma unroll
    for (int c = 0; c < K; ++c) {
      float s = t.x[0][r] * h[c][0];
      if (!(MODE & 16)) {
#pragma unroll
        for (int cc = 1; cc < CH; ++cc) s = fma_(t.x[cc][r], h[c][cc], s);
      }
      pn[r][c] = s;
    }
  if (!(MODE & 1)) reduce_scatter<G / 2, float, G, K>(pn, g);
  float wn[K], den[K], num[K], quo[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    float d = t.w[0] * hht[0][c];
    if (!(MODE & 32)) {
#pragma unroll
      for (int c2 = 1; c2 < K; ++c2) d = fma_(t.w[c2], hht[c2][c], d);
    }
    den[c] = (d == 0.f) ? eps_val<float>() : d;
    num[c] = (MODE & 1) ? pn[0][c] + pn[1][c] + pn[2][c] + pn[3][c] : pn[0][c];
  }
  if (!(MODE & 4)) {
    quotients<K>(num, den, quo);
  } else {
#pragma unroll
    for (int c = 0; c < K; ++c) quo[c] = num[c] + den[c];
  }
#pragma unroll
  for (int c = 0; c < K; ++c) wn[c] = t.w[c] * quo[c];
#pragma unroll
  for (int c = 0; c < K; ++c) t.w[c] = wn[c];
  if constexpr ((MODE & 128) != 0) {
    // W^T X with the group's rows of the new W fetched from LDS (one ds_read_b128 per component) instead of
    // 4 x K DPP broadcasts
    extern __shared__ float stage[];  // [nw][K][64]
    float* mine = stage + (threadIdx.x / 64) * (K * 64);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < K; ++c) mine[c * 64 + lane] = wn[c];
    float wg[K][G];
#pragma unroll
    for (int c = 0; c < K; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(mine + c * 64 + (lane & ~3));
      wg[c][0] = v.x; wg[c][1] = v.y; wg[c][2] = v.z; wg[c][3] = v.w;
    }
#pragma unroll
    for (int r = 0; r < G; ++r)
#pragma unroll
      for (int c = 0; c < K; ++c)
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) accA[c][cc] = fma_(wg[c][r], t.x[cc][r], accA[c][cc]);
  } else
  static_for<G>([&](auto R) {
    constexpr int r = decltype(R)::value;
#pragma unroll
    for (int c = 0; c < K; ++c) {
      const float wr = (MODE & 2) ? wn[c] : group_bcast<G, r>(wn[c]);
      if (!(MODE & 64)) {
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) accA[c][cc] = fma_(wr, t.x[cc][r], accA[c][cc]);
      } else {
        accA[c][0] += wr;
      }
    }
  });
  if (!(MODE & 8)) {
    int idx = 0;
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int c2 = c; c2 < K; ++c2) {
        accB[idx] = fma_(wn[c], wn[c2], accB[idx]);
        ++idx;
      }
  }
}

template <int K, int MODE = -1>
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
    if constexpr (MODE < 0)
      update_tile<float, G, CH, K>(t, ma, h, hht, accA, accB, 0.f, 0.f, true);
    else
      tile_variant<MODE, G, CH, K>(t, ma, h, hht, accA, accB);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int c = 0; c < K; ++c) { s += t.w[c]; for (int cc = 0; cc < CH; ++cc) s += accA[c][cc]; }
  for (int i = 0; i < C::NB; ++i) s += accB[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Row-per-lane mapping (G = 1, CH = 16) with H and H H^T wave-uniform in SGPRs: no cross-lane traffic at all and
// every numerator / denominator FMA takes an SGPR operand.  80 accumulators per lane.
template <int K, int HH_SGPR>
__global__ void __launch_bounds__(512) k_tile_g1(float* out, int iters, float seed) {
  constexpr int M = 16, NB = K * (K + 1) / 2;
  float hs[K][M], hht[K][K];
  for (int c = 0; c < K; ++c) {
    // values the compiler cannot fold: read from memory, made wave-uniform with v_readfirstlane (-> SGPRs)
    for (int j = 0; j < M; ++j) hs[c][j] = uniform(out[c * M + j] + seed);
    for (int c2 = 0; c2 < K; ++c2) {
      const float v = out[128 + c * K + c2] + seed;
      hht[c][c2] = HH_SGPR ? uniform(v) : v;
    }
  }
  float accA[K][M], accB[NB], x[M], w[K];
  for (int c = 0; c < K; ++c)
    for (int j = 0; j < M; ++j) accA[c][j] = 0.f;
  for (int i = 0; i < NB; ++i) accB[i] = 0.f;
  for (int j = 0; j < M; ++j) x[j] = seed * 0.5f + 0.001f * (threadIdx.x + j);
  for (int c = 0; c < K; ++c) w[c] = seed + 0.002f * (threadIdx.x + c);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < M; ++j) asm volatile("" : "+v"(x[j]));
    float num[K], den[K], quo[K], wn[K];
#pragma unroll
    for (int c = 0; c < K; ++c) {
      float s = x[0] * hs[c][0];
#pragma unroll
      for (int j = 1; j < M; ++j) s = fma_(x[j], hs[c][j], s);
      num[c] = s;
      float dd = w[0] * hht[0][c];
#pragma unroll
      for (int c2 = 1; c2 < K; ++c2) dd = fma_(w[c2], hht[c2][c], dd);
      den[c] = (dd == 0.f) ? eps_val<float>() : dd;
    }
    quotients<K>(num, den, quo);
#pragma unroll
    for (int c = 0; c < K; ++c) wn[c] = w[c] * quo[c];
#pragma unroll
    for (int c = 0; c < K; ++c) w[c] = wn[c];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int j = 0; j < M; ++j) accA[c][j] = fma_(wn[c], x[j], accA[c][j]);
    int idx = 0;
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int c2 = c; c2 < K; ++c2) {
        accB[idx] = fma_(wn[c], wn[c2], accB[idx]);
        ++idx;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float sres = 0.f;
  for (int c = 0; c < K; ++c) { sres += w[c]; for (int j = 0; j < M; ++j) sres += accA[c][j]; }
  for (int i = 0; i < NB; ++i) sres += accB[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sres;
}

template <int HH_SGPR>
void run_g1(float* d, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int threads : {256, 512}) {
    k_tile_g1<5, HH_SGPR><<<256, threads>>>(d, 100, 0.7f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_tile_g1<5, HH_SGPR><<<256, threads>>>(d, iters, 0.7f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = (double)iters * threads / 64 / 4;
    printf("%-34s %d waves/SIMD: %7.1f ns per tile and SIMD\n", name, threads / 256, ms * 1e6 / tiles_per_simd);
  }
}

template <int MODE>
void run(float* d, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int threads : {256, 512}) {
    k_tile<5, MODE><<<256, threads, 8 * 5 * 64 * 4>>>(d, 100, 0.7f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_tile<5, MODE><<<256, threads, 8 * 5 * 64 * 4>>>(d, iters, 0.7f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = (double)iters * threads / 64 / 4;
    printf("%-34s %d waves/SIMD: %7.1f ns per tile and SIMD\n", name, threads / 256, ms * 1e6 / tiles_per_simd);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4 * 4);
  run<-1>(d, "update_tile (as shipped)");
  run<0>(d, "local copy, everything");
  run<1>(d, "no reduce-scatter");
  run<2>(d, "no group broadcast");
  run<4>(d, "no quotient");
  run<8>(d, "no W^T W");
  run<16>(d, "no numerator FMAs");
  run<32>(d, "no denominator FMAs");
  run<64>(d, "no W^T X FMAs");
  run<3>(d, "no reduce-scatter, no broadcast");
  run<127>(d, "nothing but the skeleton");
  run<128>(d, "W rows of the group via LDS");
  run_g1<1>(d, "row per lane, H and HHt in SGPRs");
  run_g1<0>(d, "row per lane, H in SGPRs");
  return 0;
}
