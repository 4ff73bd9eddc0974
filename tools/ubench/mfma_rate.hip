// Micro-benchmark: fp32 MFMA against fp32 VALU for the two T-long contractions of the mu iteration at the headline
// shape (16 channels, k = 5), no global or LDS traffic -- the shoot-out SURVEY.md section 7 step 4 asks for.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I muscle_synergies_amd/csrc tools/ubench/mfma_rate.hip -o /tmp/mfma_rate
//
// Part A  lane maps of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction) including the
//         CBSZ / ABID broadcast of one block's A operand, checked with exact integer data.
// Part B  issue rates: 4x4x1 and 16x16x4 alone (dependent chain / independent accumulators), and next to independent
//         v_fma_f32 streams (does the matrix pipe run beside the VALU?).
// Part C  the per-tile arithmetic of the row-per-lane mapping (lane = row, 16 channels in 16 VGPRs) with
//           num  = X H^T      on VALU (H in VGPRs, as shipped in round 1) | 4x4x1 MFMA (H in ONE VGPR, ABID = channel)
//           den  = W (H H^T)  on VALU (H H^T in SGPRs)                    | 4x4x1 MFMA
//           W^T X, W^T W      on VALU (95 accumulators)
//         and, for reference, the 16x16x4 formulations the verdict lists: X H^T as 4 MFMAs per 16 rows, and
//         W^T [X | W] accumulated in 16x16 tiles (operands assumed to be in the MFMA layout already: optimistic).
// Output unit of part C: ns per 64-row tile and SIMD, as tools/ubench/tile_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nmf_kernels.hpp"
using namespace hipnmf;
using f4 = float __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------ part A
template <int CBSZ, int ABID>
__global__ void k_layout(const float* a, const float* b, float* out) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) out[r * 64 + threadIdx.x] = acc[r];
}

template <int CBSZ, int ABID>
bool check_layout(float* da, float* db, float* dout) {
  std::vector<float> a(64), b(64), o(256);
  for (int l = 0; l < 64; ++l) { a[l] = (float)(1 + l); b[l] = (float)(100 + 7 * l); }
  CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
  k_layout<CBSZ, ABID><<<1, 64>>>(da, db, dout);
  CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
  // model: D_blk[i][j] = A_src[i] * B_blk[j]; lane l = 4*blk + j holds D_blk[i = reg][j]; A_src = block ABID of the
  // group of 2^CBSZ blocks that blk belongs to (CBSZ = 0: its own block)
  int bad = 0;
  for (int r = 0; r < 4; ++r)
    for (int l = 0; l < 64; ++l) {
      const int blk = l / 4;
      const int src = CBSZ == 0 ? blk : (blk / (1 << CBSZ)) * (1 << CBSZ) + ABID;
      const float want = a[4 * src + r] * b[l];
      if (o[r * 64 + l] != want) ++bad;
    }
  printf("  4x4x1_16b cbsz=%d abid=%2d : %s\n", CBSZ, ABID, bad ? "MISMATCH" : "lane map as modelled");
  return bad == 0;
}

// ------------------------------------------------------------------------------------------------ part B
// NACC independent accumulators, NV independent v_fma per MFMA
template <int SHAPE /*0: 4x4x1, 1: 16x16x4*/, int NACC, int NV>
__global__ void __launch_bounds__(512) k_rate(float* out, int iters, float seed) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{seed, seed, seed, seed};
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  float v[NV > 0 ? NV : 1];
  for (int i = 0; i < NV; ++i) v[i] = seed * i;
  for (int it = 0; it < iters; ++it) {
    static_for<8 * NACC>([&](auto I) {
      constexpr int idx = decltype(I)::value, i = idx % NACC;
      if constexpr (SHAPE == 0)
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, idx & 15, 0);
      else
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = fma_(v[q], a, b);
    });
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += v[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NACC, int NV>
void run_rate(float* d, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int threads : {256, 512}) {
    k_rate<SHAPE, NACC, NV><<<256, threads>>>(d, 10, 0.7f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_rate<SHAPE, NACC, NV><<<256, threads>>>(d, iters, 0.7f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)iters * 8 * NACC * (threads / 256);
    printf("  %-44s %d waves/SIMD: %6.2f ns per MFMA and SIMD (+%d v_fma each)\n", name, threads / 256,
           ms * 1e6 / mfma_per_simd, NV);
  }
}

// ------------------------------------------------------------------------------------------------ part C
// NUM: 0 VALU (H in VGPRs) | 1 MFMA c = 0..3, c = 4 on VALU (H row in SGPRs) | 2 MFMA for both component groups
// DEN: 0 VALU (H H^T in SGPRs) | 1 MFMA c = 0..3 + VALU c = 4 | 2 MFMA both groups
// ACC: 0 VALU 80 + 15 accumulators | 1 none (lower bound of everything else)
// SPLIT: number of independent MFMA accumulation chains for the numerator (1 or 2)
template <int K, int NUM, int DEN, int ACC, int SPLIT>
__global__ void __launch_bounds__(512) k_tile_mfma(float* out, int iters, float seed) {
  constexpr int M = 16, NB = K * (K + 1) / 2;
  const int lane = threadIdx.x & 63;
  float hv[K][M];       // NUM == 0: H in VGPRs
  float h4[M];          // NUM == 1: row 4 of H in SGPRs
  float hht[K][K];      // SGPRs
  float hq0, hq1;       // lane l: H[l%4][l/4] and H[4 + l%4][l/4]
  float hhq0, hhq1;     // lane l: HHt[c' = l/4][c = l%4] and HHt[c' = l/4][4 + l%4]
  for (int c = 0; c < K; ++c) {
    for (int j = 0; j < M; ++j) hv[c][j] = out[c * M + j] + seed;
    for (int c2 = 0; c2 < K; ++c2) hht[c][c2] = uniform(out[128 + c * K + c2] + seed);
  }
  for (int j = 0; j < M; ++j) h4[j] = uniform(out[(K - 1) * M + j] + seed);
  hq0 = out[(lane % 4) * M + lane / 4] + seed;
  hq1 = (4 + lane % 4 < K) ? out[(4 + lane % 4) * M + lane / 4] + seed : 0.f;
  hhq0 = (lane / 4 < K) ? out[128 + (lane / 4) * K + lane % 4] + seed : 0.f;
  hhq1 = (lane / 4 < K && 4 + lane % 4 < K) ? out[128 + (lane / 4) * K + 4 + lane % 4] + seed : 0.f;
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
    if constexpr (NUM == 0) {
#pragma unroll
      for (int c = 0; c < K; ++c) {
        float s = x[0] * hv[c][0];
#pragma unroll
        for (int j = 1; j < M; ++j) s = fma_(x[j], hv[c][j], s);
        num[c] = s;
      }
    } else {
      f4 n0[SPLIT], n1[SPLIT];
#pragma unroll
      for (int s = 0; s < SPLIT; ++s) n0[s] = n1[s] = f4{0.f, 0.f, 0.f, 0.f};
      static_for<M>([&](auto J) {
        constexpr int j = decltype(J)::value;
        n0[j % SPLIT] = __builtin_amdgcn_mfma_f32_4x4x1f32(hq0, x[j], n0[j % SPLIT], 4, j, 0);
        if constexpr (NUM == 2) n1[j % SPLIT] = __builtin_amdgcn_mfma_f32_4x4x1f32(hq1, x[j], n1[j % SPLIT], 4, j, 0);
      });
#pragma unroll
      for (int s = 1; s < SPLIT; ++s) { n0[0] += n0[s]; if constexpr (NUM == 2) n1[0] += n1[s]; }
#pragma unroll
      for (int c = 0; c < (K < 4 ? K : 4); ++c) num[c] = n0[0][c];
      if constexpr (NUM == 2) {
#pragma unroll
        for (int c = 4; c < K; ++c) num[c] = n1[0][c - 4];
      } else {
        static_assert(K <= 5, "NUM == 1 handles one component on the VALU");
        if constexpr (K == 5) {
          float s = x[0] * h4[0];
#pragma unroll
          for (int j = 1; j < M; ++j) s = fma_(x[j], h4[j], s);
          num[4] = s;
        }
      }
    }
    if constexpr (DEN == 0) {
#pragma unroll
      for (int c = 0; c < K; ++c) {
        float dd = w[0] * hht[0][c];
#pragma unroll
        for (int c2 = 1; c2 < K; ++c2) dd = fma_(w[c2], hht[c2][c], dd);
        den[c] = dd;
      }
    } else {
      f4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      static_for<K>([&](auto C2) {
        constexpr int c2 = decltype(C2)::value;
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(hhq0, w[c2], d0, 4, c2, 0);
        if constexpr (DEN == 2) d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(hhq1, w[c2], d1, 4, c2, 0);
      });
#pragma unroll
      for (int c = 0; c < (K < 4 ? K : 4); ++c) den[c] = d0[c];
      if constexpr (DEN == 2) {
#pragma unroll
        for (int c = 4; c < K; ++c) den[c] = d1[c - 4];
      } else if constexpr (K == 5) {
        float dd = w[0] * hht[0][4];
#pragma unroll
        for (int c2 = 1; c2 < K; ++c2) dd = fma_(w[c2], hht[c2][4], dd);
        den[4] = dd;
      }
    }
#pragma unroll
    for (int c = 0; c < K; ++c) den[c] = (den[c] == 0.f) ? eps_val<float>() : den[c];
    quotients<K>(num, den, quo);
#pragma unroll
    for (int c = 0; c < K; ++c) wn[c] = w[c] * quo[c];
#pragma unroll
    for (int c = 0; c < K; ++c) w[c] = wn[c];
    if constexpr (ACC == 0) {
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
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float sres = 0.f;
  for (int c = 0; c < K; ++c) { sres += w[c]; for (int j = 0; j < M; ++j) sres += accA[c][j]; }
  for (int i = 0; i < NB; ++i) sres += accB[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sres;
}

// v2 of the row-per-lane tile: everything that is an FMA over pairs of adjacent channels is a v_pk_fma_f32 (one
// wave alone issues a packed FMA every ~2.4 ns = 1.2 ns per FMA against 1.9 ns for v_fmac_f32: tools/ubench/
// vgpr_bank.hip), X H^T for components 0..3 (and optionally W H H^T) on the matrix pipe, component 4 packed.
//   NUMV: 0 all on VALU packed (H as 40 VGPR pairs) | 1 MFMA c < 4 + packed c = 4 (H row 4 as 8 VGPR pairs)
//   DENV: 0 VALU scalar with H H^T in SGPRs | 1 MFMA c < 4 + VALU c = 4
//   ASM:  1 broadcast of wn[c] through op_sel in inline assembly | 0 compiler splat (v_mov per component)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma_lo(f2 w, f2 x, f2 acc) {  // acc += {w.x, w.x} * x
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(x));
  return acc;
}
__device__ __forceinline__ f2 pk_fma_hi(f2 w, f2 x, f2 acc) {  // acc += {w.y, w.y} * x
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(x));
  return acc;
}
template <int NUMV, int DENV, int ASM>
__global__ void __launch_bounds__(512) k_tile_v2(float* out, int iters, float seed) {
  constexpr int K = 5, M = 16, NB = 15;
  const int lane = threadIdx.x & 63;
  f2 hp[K][8];      // NUMV == 0: all of H as pairs; NUMV == 1: only row 4 is used
  float hht[K][K];
  for (int c = 0; c < K; ++c) {
    for (int p = 0; p < 8; ++p) hp[c][p] = f2{out[c * M + 2 * p] + seed, out[c * M + 2 * p + 1] + seed};
    for (int c2 = 0; c2 < K; ++c2) hht[c][c2] = uniform(out[128 + c * K + c2] + seed);
  }
  const float hq0 = out[(lane % 4) * M + lane / 4] + seed;
  const float hhq0 = (lane / 4 < K) ? out[128 + (lane / 4) * K + lane % 4] + seed : 0.f;
  f2 accA[K][8];
  float accB[NB];
  f2 x2[8];
  float w[K];
  for (int c = 0; c < K; ++c)
    for (int p = 0; p < 8; ++p) accA[c][p] = f2{0.f, 0.f};
  for (int i = 0; i < NB; ++i) accB[i] = 0.f;
  for (int p = 0; p < 8; ++p) x2[p] = f2{seed * 0.5f + 0.001f * (threadIdx.x + p), seed * 0.4f + 0.002f * (threadIdx.x + p)};
  for (int c = 0; c < K; ++c) w[c] = seed + 0.002f * (threadIdx.x + c);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(x2[p]));
    float num[K], den[K], quo[K];
    if constexpr (NUMV == 0) {
#pragma unroll
      for (int c = 0; c < K; ++c) {
        f2 s2 = x2[0] * hp[c][0];
#pragma unroll
        for (int p = 1; p < 8; ++p) s2 = __builtin_elementwise_fma(x2[p], hp[c][p], s2);
        num[c] = s2.x + s2.y;
      }
    } else {
      f4 na = {0.f, 0.f, 0.f, 0.f}, nb = na;
      static_for<8>([&](auto P) {
        constexpr int p = decltype(P)::value;
        na = __builtin_amdgcn_mfma_f32_4x4x1f32(hq0, x2[p].x, na, 4, 2 * p, 0);
        nb = __builtin_amdgcn_mfma_f32_4x4x1f32(hq0, x2[p].y, nb, 4, 2 * p + 1, 0);
      });
      f2 s2 = x2[0] * hp[4][0];
#pragma unroll
      for (int p = 1; p < 8; ++p) s2 = __builtin_elementwise_fma(x2[p], hp[4][p], s2);
      const f4 n = na + nb;
#pragma unroll
      for (int c = 0; c < 4; ++c) num[c] = n[c];
      num[4] = s2.x + s2.y;
    }
    if constexpr (DENV == 0) {
#pragma unroll
      for (int c = 0; c < K; ++c) {
        float dd = w[0] * hht[0][c];
#pragma unroll
        for (int c2 = 1; c2 < K; ++c2) dd = fma_(w[c2], hht[c2][c], dd);
        den[c] = dd;
      }
    } else {
      f4 d0 = {0.f, 0.f, 0.f, 0.f};
      static_for<K>([&](auto C2) {
        constexpr int c2 = decltype(C2)::value;
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(hhq0, w[c2], d0, 4, c2, 0);
      });
#pragma unroll
      for (int c = 0; c < 4; ++c) den[c] = d0[c];
      float dd = w[0] * hht[0][4];
#pragma unroll
      for (int c2 = 1; c2 < K; ++c2) dd = fma_(w[c2], hht[c2][4], dd);
      den[4] = dd;
    }
#pragma unroll
    for (int c = 0; c < K; ++c) den[c] = (den[c] == 0.f) ? eps_val<float>() : den[c];
    quotients<K>(num, den, quo);
    f2 wn2[3];
    wn2[0] = f2{w[0] * quo[0], w[1] * quo[1]};
    wn2[1] = f2{w[2] * quo[2], w[3] * quo[3]};
    wn2[2] = f2{w[4] * quo[4], 0.f};
#pragma unroll
    for (int c = 0; c < K; ++c) w[c] = wn2[c / 2][c % 2];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if constexpr (ASM) {
          accA[c][p] = (c % 2) ? pk_fma_hi(wn2[c / 2], x2[p], accA[c][p]) : pk_fma_lo(wn2[c / 2], x2[p], accA[c][p]);
        } else {
          accA[c][p] = __builtin_elementwise_fma(f2{w[c], w[c]}, x2[p], accA[c][p]);
        }
      }
    int idx = 0;
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int c2 = c; c2 < K; ++c2) {
        accB[idx] = fma_(w[c], w[c2], accB[idx]);
        ++idx;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float sres = 0.f;
  for (int c = 0; c < K; ++c) { sres += w[c]; for (int p = 0; p < 8; ++p) sres += accA[c][p].x + accA[c][p].y; }
  for (int i = 0; i < NB; ++i) sres += accB[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sres;
}
template <int NUMV, int DENV, int ASM>
void run_tile_v2(float* d, const char* name);

// 16x16x4 formulations (operands assumed to be in MFMA layout already; 64 rows = 4 sub-tiles of 16 rows):
//  STEP1: X H^T = 4 MFMAs per 16 rows (A = X sub-tile, K = channels; B = H^T padded to 16 columns);
//  STEP2: W^T [X | W] = per 4 rows one MFMA for the X block and one for the W block (M = k padded to 16),
//         accumulated over the whole pass in two 16x16 accumulators;
//  the remaining VALU work (den, quotient, W update) is issued alongside in the padded D layout: each lane holds
//  4 (row, component) results of which only the lanes with component < k are useful, so per 16 rows the element-wise
//  part costs what 64 rows cost in the row-per-lane layout.
template <int K, int STEP1, int STEP2>
__global__ void __launch_bounds__(512) k_tile_16(float* out, int iters, float seed) {
  constexpr int M = 16, NB = K * (K + 1) / 2;
  float hb[4];          // B operand of step 1: H^T chunk s (lane: k = l/16 -> channel 4 (l/16) + s, column l%16 = c)
  float hht[K][K];
  for (int s = 0; s < 4; ++s) hb[s] = out[threadIdx.x % 64 + 64 * s] + seed;
  for (int c = 0; c < K; ++c)
    for (int c2 = 0; c2 < K; ++c2) hht[c][c2] = uniform(out[128 + c * K + c2] + seed);
  float x[M], w[K];
  float accA[K][M], accB[NB];
  for (int c = 0; c < K; ++c)
    for (int j = 0; j < M; ++j) accA[c][j] = 0.f;
  for (int i = 0; i < NB; ++i) accB[i] = 0.f;
  f4 a2x = {0.f, 0.f, 0.f, 0.f}, a2w = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < M; ++j) x[j] = seed * 0.5f + 0.001f * (threadIdx.x + j);
  for (int c = 0; c < K; ++c) w[c] = seed + 0.002f * (threadIdx.x + c);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < M; ++j) asm volatile("" : "+v"(x[j]));
    float num[K], den[K], quo[K], wn[K];
    if constexpr (STEP1) {
      // 4 sub-tiles of 16 rows x 4 k-steps; x[4 q + s] plays "sub-tile q, k-step s" (one VGPR each, as loaded by a
      // 16-byte row-major load in the A layout)
      f4 d[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d[q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) d[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[4 * q + s], hb[s], d[q], 0, 0, 0);
      }
      // padded D layout: lane (c = l%16, g = l/16), register r <-> row 4 g + r of the sub-tile: 16 results per lane and
      // tile; the first K "results" stand in for the lane's row in the VALU part below so that its cost is comparable
#pragma unroll
      for (int c = 0; c < K; ++c) num[c] = d[c % 4][c / 4 % 4] + d[(c + 1) % 4][(c + 2) % 4];
    } else {
#pragma unroll
      for (int c = 0; c < K; ++c) {
        float s = x[0] * hb[0];
#pragma unroll
        for (int j = 1; j < M; ++j) s = fma_(x[j], hb[j % 4], s);
        num[c] = s + (float)c;
      }
    }
#pragma unroll
    for (int c = 0; c < K; ++c) {
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
    if constexpr (STEP2) {
      // 64 rows = 16 k-steps of 4 rows; A = W^T chunk (lane: i = c, k = row), B = X chunk / W chunk
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        a2x = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[s % K], x[s], a2x, 0, 0, 0);
        a2w = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[s % K], wn[(s + 1) % K], a2w, 0, 0, 0);
      }
    } else {
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
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float sres = a2x[0] + a2x[1] + a2x[2] + a2x[3] + a2w[0] + a2w[1] + a2w[2] + a2w[3];
  for (int c = 0; c < K; ++c) { sres += w[c]; for (int j = 0; j < M; ++j) sres += accA[c][j]; }
  for (int i = 0; i < NB; ++i) sres += accB[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sres;
}

template <typename F>
void time_tile(F launch, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int threads : {256, 512}) {
    launch(threads, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(threads, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double tiles_per_simd = (double)iters * threads / 64 / 4;
    printf("  %-52s %d waves/SIMD: %7.1f ns per tile and SIMD\n", name, threads / 256, ms * 1e6 / tiles_per_simd);
  }
}

template <int K, int NUM, int DEN, int ACC, int SPLIT>
void run_tile(float* d, const char* name) {
  time_tile([&](int threads, int iters) { k_tile_mfma<K, NUM, DEN, ACC, SPLIT><<<256, threads>>>(d, iters, 0.7f); }, name);
}
template <int K, int S1, int S2>
void run_tile16(float* d, const char* name) {
  time_tile([&](int threads, int iters) { k_tile_16<K, S1, S2><<<256, threads>>>(d, iters, 0.7f); }, name);
}

template <int NUMV, int DENV, int ASM>
void run_tile_v2(float* d, const char* name) {
  time_tile([&](int threads, int iters) { k_tile_v2<NUMV, DENV, ASM><<<256, threads>>>(d, iters, 0.7f); }, name);
}

int main() {
  float* d;
  CK(hipMalloc(&d, 256 * 1024 * 4 * 4));
  CK(hipMemset(d, 0, 256 * 1024 * 4 * 4));
  printf("part A: lane maps\n");
  bool ok = true;
  ok &= check_layout<0, 0>(d, d + 64, d + 128);
  ok &= check_layout<4, 0>(d, d + 64, d + 128);
  ok &= check_layout<4, 5>(d, d + 64, d + 128);
  ok &= check_layout<4, 15>(d, d + 64, d + 128);
  ok &= check_layout<2, 3>(d, d + 64, d + 128);
  CK(hipMemset(d, 0, 256 * 1024 * 4 * 4));
  printf("part B: issue rates\n");
  run_rate<0, 1, 0>(d, "4x4x1_16b, one dependent chain");
  run_rate<0, 2, 0>(d, "4x4x1_16b, 2 accumulators");
  run_rate<0, 4, 0>(d, "4x4x1_16b, 4 accumulators");
  run_rate<0, 4, 1>(d, "4x4x1_16b, 4 accumulators");
  run_rate<0, 4, 2>(d, "4x4x1_16b, 4 accumulators");
  run_rate<0, 4, 4>(d, "4x4x1_16b, 4 accumulators");
  run_rate<0, 1, 2>(d, "4x4x1_16b, one dependent chain");
  run_rate<1, 1, 0>(d, "16x16x4, one dependent chain");
  run_rate<1, 4, 0>(d, "16x16x4, 4 accumulators");
  run_rate<1, 4, 4>(d, "16x16x4, 4 accumulators");
  run_rate<1, 4, 8>(d, "16x16x4, 4 accumulators");
  printf("part C: per-tile arithmetic, 16 channels, k = 5\n");
  run_tile<5, 0, 0, 0, 1>(d, "VALU everything (H in VGPRs; round-1 kernel)");
  run_tile<5, 1, 0, 0, 1>(d, "num: MFMA c<4 + VALU c=4");
  run_tile<5, 1, 0, 0, 2>(d, "num: MFMA c<4 (2 chains) + VALU c=4");
  run_tile<5, 2, 0, 0, 1>(d, "num: MFMA both groups");
  run_tile<5, 2, 0, 0, 2>(d, "num: MFMA both groups (2 chains)");
  run_tile<5, 1, 1, 0, 1>(d, "num + den: MFMA c<4, VALU c=4");
  run_tile<5, 2, 2, 0, 1>(d, "num + den: MFMA both groups");
  run_tile<5, 2, 2, 0, 2>(d, "num + den: MFMA both groups (2 chains)");
  run_tile<5, 0, 0, 1, 1>(d, "VALU, no W^T X / W^T W (step 1 only)");
  run_tile<5, 2, 2, 1, 1>(d, "num + den MFMA, no W^T X / W^T W (step 1 only)");
  run_tile_v2<0, 0, 0>(d, "v2: packed VALU everything (compiler splat)");
  run_tile_v2<0, 0, 1>(d, "v2: packed VALU everything (op_sel broadcast)");
  run_tile_v2<1, 0, 1>(d, "v2: num MFMA c<4 + packed c=4, den VALU");
  run_tile_v2<1, 1, 1>(d, "v2: num + den MFMA c<4, packed / VALU c=4");
  run_tile_v2<1, 1, 0>(d, "v2: same, compiler splat");
  run_tile<4, 0, 0, 0, 1>(d, "k=4: VALU everything");
  run_tile<4, 2, 2, 0, 1>(d, "k=4: num + den MFMA");
  run_tile16<5, 1, 0>(d, "16x16x4: X H^T on MFMA, rest VALU");
  run_tile16<5, 0, 1>(d, "16x16x4: W^T [X|W] on MFMA, rest VALU");
  run_tile16<5, 1, 1>(d, "16x16x4: both on MFMA, element-wise VALU");
  printf(ok ? "lane maps: OK\n" : "lane maps: MISMATCH\n");
  return ok ? 0 : 1;
}
