// nmf_rowlane.hpp -- the matrix-pipe instances of the solver (round 2): fp32, 9..16 channels, k = 1..8.
//   fit_rowlane_kernel<K>          batch path: one 512-thread workgroup per matrix, every iteration inside the kernel;
//                                  the library's default for k >= 6 (hipnmf_set_tuning variant 5 / HIPNMF_ROWLANE=1: any k)
//   slice_pass_rowlane_kernel<K>   the time-shard / row-sliced pass on channel-major X (hipnmf_shard_pass_f32, config #5)
//
// Arithmetic replaced: sklearn/decomposition/_nmf.py (1.7.2) _multiplicative_update_w (:526-631),
// _multiplicative_update_h (:634-728), _beta_divergence (:85-134), loop + stop rule (:731-893), reached from the
// reference at src/muscle_synergies/analysis.py:862-863.  sklearn notation: X (T x m) ~ W (T x k) H (k x m).
//
//   * X H^T and W (H H^T) run on the matrix pipe as v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per
//     instruction, K = 1): with the row-per-lane X register of channel j as the B operand (block b = rows 4b..4b+3)
//     and ONE register holding H for all 16 channels as the A operand (lane 4j + c <-> H[c][j]; CBSZ = 4 broadcasts
//     block ABID = j to all blocks), 16 instructions leave  sum_j H[c][j] X[row][j]  for c = 0..3 in the 4 accumulator
//     registers of the lane that owns the row -- the row-per-lane layout in and out, no transposes.  Components 4..7
//     take a second group.  Exact fp32 (an fmaf chain).
//   * W^T X / W^T W stay on the VALU (16 k + k (k + 1) / 2 accumulators per lane): with the rows on the lanes no MFMA
//     shape contracts over them (every f32 MFMA keeps the non-contracted index on the low lane bits).
//   * What this buys, measured (tools/ubench/mfma_rate.hip, tools/run_ksweep.sh, profiles/README.md): NOT arithmetic
//     speed -- the f32 matrix pipe has the VALU's FLOP rate and the two do not overlap, so at k = 5 the tile costs
//     ~15 % more time than the VALU form and the kernel runs 9.9 vs 10.15 M matrix-it/s -- but registers: H shrinks
//     from 16 k VGPRs to 2 (+2 for H H^T), so k = 6, 7, 8 fit two waves per SIMD (228 / 239 / 256 VGPRs, no scratch),
//     which the VALU form cannot (96..128 VGPRs of H): 8.9 / 8.1 / 7.3 M it/s against 7.0 / 5.3 / 4.1 M for the
//     channel-major fallback, 0.93 - 0.95 of their byte roofs.
//   * The batch kernel is bound by the rate at which the XCDs can re-read their Infinity-Cache-resident matrices
//     (7.2 - 8.1 TB/s, tools/ubench/mall_stream.hip), so the freed registers were also tried as storage -- the rows of
//     W that do not fit in LDS (NWR tiles per wave) and the first NXR tiles of X resident in VGPRs for the whole fit:
//     it loses (numbers at HIPNMF_RL_NXR below) and is compiled out by default.
//
// Everything else (LDS-resident W, SRD loads with hardware range checks, reduce-scatter of the per-lane sums, the
// iteration epilogue by wave 0, stop rule, residual / VAF statistics) follows nmf_kernels.hpp.
#pragma once
#include "nmf_kernels.hpp"
#include "nmf_rowlane_decl.hpp"

namespace hipnmf {

using f4 = float __attribute__((ext_vector_type(4)));

template <int ABID>
__device__ __forceinline__ f4 mfma_4x4_bcast(float a, float b, f4 c) {
  // D_blk[i][j] = C_blk[i][j] + A_{block ABID}[i] * B_blk[j]   (lane 4 blk + j holds D_blk[0..3][j])
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}

// Resident tiles per wave (compile-time: registers are indexed statically): NWR tiles of W that follow the LDS cache
// and the first NXR tiles of X can stay in VGPRs for the whole fit.  Measured at 16 x 10 000, k = 5, B = 2048
// (tools/run_variants.sh, profiles/README.md): (NXR, NWR) = (0, 0) 9.24 M matrix-it/s, (0, 5) 8.69 M, (1, 5) 8.83 M,
// (2, 4) 8.16 M; one wave per SIMD (HIPNMF_RL_THREADS=256) with (10, 10): 6.13 M -- the kernel needs 195 VGPRs before any resident tile (accumulators 95, MFMA results 24, one X tile
// 16, quotient temporaries), so more than ~3 tiles spill, and the wave-uniform register selects cost more issue
// slots than the 13 % of traffic they remove buys back.  Default: nothing resident.
#ifndef HIPNMF_RL_NXR
#define HIPNMF_RL_NXR 0
#endif
#ifndef HIPNMF_RL_NWR
#define HIPNMF_RL_NWR 0
#endif
#ifndef HIPNMF_RL_PF
#define HIPNMF_RL_PF 1  // X tiles in flight per wave in the streaming loop (1 or 2); 2: 9.13 vs 9.24 M matrix-it/s
#endif
template <int K>
constexpr int rl_nwr() {
  return (HIPNMF_RL_NWR);
}
template <int K>
constexpr int rl_nxr() {
  return (HIPNMF_RL_NXR);
}

// per-wave addressing state
template <int K>
struct RlCtx {
  rsrc_t xr, wr;
  unsigned xvoff;    // lane * (bytes per row of X)
  unsigned xrow_b;   // bytes per row of X
  unsigned woff;     // lane * 4
  unsigned ldw_b;    // bytes per component row of W
  int T, lane, wave, stride;  // stride = rows per workgroup-step
  float* lds_w;      // [K][lds_stride] component-major cache of rows [0, ilds * stride)
  int lds_stride;
  int ilds;          // tiles i < ilds of every wave have their W rows in LDS
};

// rows [wbase, wbase + 64): the lane's own row, 16 channels = four 16-byte loads; rows >= T read as zero
template <int K>
__device__ __forceinline__ void rl_load_x(const RlCtx<K>& cx, float (&x)[16], int wbase, bool valid) {
  const bool ok = valid && (wbase + cx.lane < cx.T);
  const unsigned v = ok ? cx.xvoff : OOB;
  const unsigned srow = (unsigned)wbase * cx.xrow_b;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    buf_load<float, 4>(cx.xr, v + 16u * (unsigned)q, srow, *reinterpret_cast<float(*)[4]>(&x[4 * q]));
}

template <int K>
__device__ __forceinline__ void rl_load_w_global(const RlCtx<K>& cx, float (&w)[K], int wbase) {
  const unsigned wv = (wbase + cx.lane < cx.T) ? cx.woff : OOB;
  const unsigned sbase = (unsigned)wbase * 4u;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    float tmp[1];
    buf_load<float, 1>(cx.wr, wv, sbase + (unsigned)c * cx.ldw_b, tmp);
    w[c] = tmp[0];
  }
}
template <int K>
__device__ __forceinline__ void rl_store_w_global(const RlCtx<K>& cx, const float (&w)[K], int wbase) {
  const unsigned wv = (wbase + cx.lane < cx.T) ? cx.woff : OOB;
  const unsigned sbase = (unsigned)wbase * 4u;
#pragma unroll
  for (int c = 0; c < K; ++c) buf_store<float>(cx.wr, wv, sbase + (unsigned)c * cx.ldw_b, w[c]);
}

// tile i of this wave is streamed: its X always, its W too when the rows live neither in LDS nor in registers
// (issued one tile ahead of their use, like X)
template <int K, int NWR>
__device__ __forceinline__ void rl_prefetch(const RlCtx<K>& cx, float (&x)[16], float (&wg)[K], int i, int wbase,
                                            bool valid) {
  rl_load_x<K>(cx, x, wbase, valid);
  if (valid && i >= cx.ilds + NWR) rl_load_w_global<K>(cx, wg, wbase);
}

// operands of the two MFMA products, rebuilt from LDS after every H update
struct RlHops {
  float hq0, hq1;    // lane l: H[l%4][l/4], H[4 + l%4][l/4]        (0 where the component does not exist)
  float hhq0, hhq1;  // lane l: HHt[l/4][l%4], HHt[l/4][4 + l%4]    (0 outside k x k)
};
template <int K>
__device__ __forceinline__ void rl_load_hops(const Smem<float, 1, 16, K>& s, int lane, RlHops& o, float (&hht)[K][K]) {
  const int lo = lane & 3, hi = lane >> 2;
  o.hq0 = (lo < K) ? s.H[lo * 16 + hi] : 0.f;
  o.hq1 = (4 + lo < K) ? s.H[(4 + lo) * 16 + hi] : 0.f;
  o.hhq0 = (hi < K && lo < K) ? s.HHt[hi * K + lo] : 0.f;
  o.hhq1 = (hi < K && 4 + lo < K) ? s.HHt[hi * K + 4 + lo] : 0.f;
#pragma unroll
  for (int c = 0; c < K; ++c)
#pragma unroll
    for (int c2 = 0; c2 < K; ++c2) hht[c][c2] = uniform(s.HHt[c * K + c2]);
}

// One 64-row tile: W update of the lane's row (_nmf.py:540-554, 615-631) and accumulation of W^T X / W^T W
// (:638-640).  DENV: denominator on the VALU with H H^T in SGPRs (25 FMAs at k = 5) instead of the matrix pipe.
#ifndef HIPNMF_RL_DEN_VALU
#define HIPNMF_RL_DEN_VALU 0
#endif
template <int K>
__device__ __forceinline__ void rl_update(const float (&x)[16], float (&w)[K], const RlHops& ho,
                                          const float (&hht)[K][K], float (&accA)[K][16],
                                          float (&accB)[K * (K + 1) / 2], float l1w, float l2w, bool update_h) {
  constexpr bool TWO = K > 4;
  // numerator X H^T: two accumulation chains per component group (even / odd channels)
  f4 n0a = {0.f, 0.f, 0.f, 0.f}, n0b = n0a, n1a = n0a, n1b = n0a;
  static_for<16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    if constexpr ((j & 1) == 0) {
      n0a = mfma_4x4_bcast<j>(ho.hq0, x[j], n0a);
      if constexpr (TWO) n1a = mfma_4x4_bcast<j>(ho.hq1, x[j], n1a);
    } else {
      n0b = mfma_4x4_bcast<j>(ho.hq0, x[j], n0b);
      if constexpr (TWO) n1b = mfma_4x4_bcast<j>(ho.hq1, x[j], n1b);
    }
  });
  float num[K], den[K], quo[K], wn[K];
  // denominator W (H H^T)
  if constexpr (HIPNMF_RL_DEN_VALU) {
#pragma unroll
    for (int c = 0; c < K; ++c) {
      float d = w[0] * hht[0][c];
#pragma unroll
      for (int c2 = 1; c2 < K; ++c2) d = fma_(w[c2], hht[c2][c], d);
      den[c] = d;
    }
  } else {
    f4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
    static_for<K>([&](auto C2) {
      constexpr int c2 = decltype(C2)::value;
      d0 = mfma_4x4_bcast<c2>(ho.hhq0, w[c2], d0);
      if constexpr (TWO) d1 = mfma_4x4_bcast<c2>(ho.hhq1, w[c2], d1);
    });
#pragma unroll
    for (int c = 0; c < K; ++c) den[c] = c < 4 ? d0[c & 3] : d1[c & 3];
  }
  const f4 n0 = n0a + n0b, n1 = n1a + n1b;
  if (l1w > 0.f || l2w > 0.f) {  // wave-uniform and rare: one scalar branch instead of 4 K selects per tile
#pragma unroll
    for (int c = 0; c < K; ++c) {
      float d = den[c];
      if (l1w > 0.f) d = d + l1w;
      if (l2w > 0.f) d = d + l2w * w[c];
      den[c] = d;
    }
  }
#pragma unroll
  for (int c = 0; c < K; ++c) {
    num[c] = c < 4 ? n0[c & 3] : n1[c & 3];
    den[c] = (den[c] == 0.f) ? eps_val<float>() : den[c];
  }
  quotients<K>(num, den, quo);
#pragma unroll
  for (int c = 0; c < K; ++c) wn[c] = w[c] * quo[c];
#pragma unroll
  for (int c = 0; c < K; ++c) w[c] = wn[c];
  if (update_h) {
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) accA[c][j] = fma_(wn[c], x[j], accA[c][j]);
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

// ---- Kullback-Leibler loss on the same mapping (round 3; _nmf.py:556-591, 642-684) ----------------------------------
// Both reconstructions W H of a tile run on the pipe as well: for channels 4 q .. 4 q + 3 of the lane's row,
//   WH[row][4 q + i] = sum_c H[c][4 q + i] W[row][c]   =   K instructions  D += A_{block (c, q)} x B  with B = the lane's w[c]
// and A broadcast from ONE register that holds H[c][4 q + i] at lane 4 (4 c + q) + i (a second register from c = 4 on);
// the result is row-per-lane again (4 channels per accumulator).  Q = X / max(WH, eps) stays on the VALU (16 quotients),
// Q H^T is the X H^T form with Q in X's place, W'^T Q' / colsum(W') accumulate on the VALU like W^T X.  What moved off the
// VALU per tile: 2 x 16 K FMAs of the reconstructions and 16 K of Q H^T.  Measured (B = 2048 x (16 x 10 000), M matrix-it/s,
// VALU instance -> this one): k = 3 6.59 -> 6.29, k = 5 5.24 -> 4.95, k = 8 3.05 -> 3.87 -- like the Frobenius flavour it pays
// from k = 6 on, where the VALU form no longer fits two waves per SIMD; the library picks it there.
struct RlKlOps {
  float hr0, hr1;  // lane l: H[c][4 q + l % 4] with 4 c + q = l / 4 (hr0) or 16 + l / 4 (hr1); 0 where c >= k
};
template <int K>
__device__ __forceinline__ void rl_load_klops(const Smem<float, 1, 16, K>& s, int lane, RlKlOps& o, float (&hsum)[K]) {
  const int i = lane & 3, p = lane >> 2;
  const int c0 = p >> 2, q0 = p & 3, c1 = 4 + (p >> 2);
  o.hr0 = (c0 < K) ? s.H[c0 * 16 + 4 * q0 + i] : 0.f;
  o.hr1 = (c1 < K) ? s.H[c1 * 16 + 4 * q0 + i] : 0.f;
#pragma unroll
  for (int c = 0; c < K; ++c) hsum[c] = uniform(s.HHt[c]);  // rowsum(H): compute_hsum / wave0_combine_and_update_h_kl
}
// W H of the lane's row: wh[q][i] = (W H)[row][4 q + i]
template <int K>
__device__ __forceinline__ void rl_reconstruct(const float (&w)[K], const RlKlOps& ko, f4 (&wh)[4]) {
  static_for<4>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    static_for<K>([&](auto Cc) {
      constexpr int c = decltype(Cc)::value;
      constexpr int p = 4 * c + q;
      if constexpr (p < 16)
        acc = mfma_4x4_bcast<p>(ko.hr0, w[c], acc);
      else
        acc = mfma_4x4_bcast<p - 16>(ko.hr1, w[c], acc);
    });
    wh[q] = acc;
  });
}
template <int K>
__device__ __forceinline__ void rl_update_kl(const float (&x)[16], float (&w)[K], const RlHops& ho, const RlKlOps& ko,
                                             const float (&hsum)[K], float (&accA)[K][16], float (&accB)[K * (K + 1) / 2],
                                             float l1w, float l2w, bool update_h) {
  constexpr bool TWO = K > 4;
  f4 wh[4];
  rl_reconstruct<K>(w, ko, wh);
  float q[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float r = wh[j >> 2][j & 3];
    q[j] = kl_quot(x[j], kl_floor(r));
  }
  // numerator Q H^T: the X H^T form of rl_update with Q in the place of X
  f4 n0a = {0.f, 0.f, 0.f, 0.f}, n0b = n0a, n1a = n0a, n1b = n0a;
  static_for<16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    if constexpr ((j & 1) == 0) {
      n0a = mfma_4x4_bcast<j>(ho.hq0, q[j], n0a);
      if constexpr (TWO) n1a = mfma_4x4_bcast<j>(ho.hq1, q[j], n1a);
    } else {
      n0b = mfma_4x4_bcast<j>(ho.hq0, q[j], n0b);
      if constexpr (TWO) n1b = mfma_4x4_bcast<j>(ho.hq1, q[j], n1b);
    }
  });
  const f4 n0 = n0a + n0b, n1 = n1a + n1b;
  float num[K], den[K], quo[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    float d = hsum[c];
    if (l1w > 0.f) d = d + l1w;
    if (l2w > 0.f) d = d + l2w * w[c];
    den[c] = (d == 0.f) ? eps_val<float>() : d;
    num[c] = c < 4 ? n0[c & 3] : n1[c & 3];
  }
  quotients<K>(num, den, quo);
#pragma unroll
  for (int c = 0; c < K; ++c) w[c] = w[c] * quo[c];
  if (update_h) {
    rl_reconstruct<K>(w, ko, wh);  // with the updated row (_nmf.py:642-684)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float r = wh[j >> 2][j & 3];
      q[j] = kl_quot(x[j], kl_floor(r));
    }
#pragma unroll
    for (int c = 0; c < K; ++c) {
#pragma unroll
      for (int j = 0; j < 16; ++j) accA[c][j] = fma_(w[c], q[j], accA[c][j]);
      accB[c] += w[c];
    }
  }
}

// W of tile i of this wave: LDS cache, resident registers, or global memory (wave-uniform choice)
template <int K, int NWR, bool PREFETCHED = false>
__device__ __forceinline__ void rl_get_w(const RlCtx<K>& cx, float (&w)[K], const float (&wres)[NWR > 0 ? NWR : 1][K],
                                         int i, int wbase, const float* wg = nullptr) {
  if (i < cx.ilds) {
    const float* p = cx.lds_w + wbase + cx.lane;
#pragma unroll
    for (int c = 0; c < K; ++c) w[c] = p[c * cx.lds_stride];
  } else if (NWR > 0 && i - cx.ilds < NWR) {
    const int q = i - cx.ilds;
    static_for<NWR>([&](auto Q) {
      constexpr int qq = decltype(Q)::value;
      if (q == qq) {
#pragma unroll
        for (int c = 0; c < K; ++c) w[c] = wres[qq][c];
      }
    });
  } else if constexpr (PREFETCHED) {
#pragma unroll
    for (int c = 0; c < K; ++c) w[c] = wg[c];
  } else {
    rl_load_w_global<K>(cx, w, wbase);
  }
}
template <int K, int NWR>
__device__ __forceinline__ void rl_put_w(const RlCtx<K>& cx, const float (&w)[K], float (&wres)[NWR > 0 ? NWR : 1][K],
                                         int i, int wbase) {
  if (i < cx.ilds) {
    float* p = cx.lds_w + wbase + cx.lane;
#pragma unroll
    for (int c = 0; c < K; ++c) p[c * cx.lds_stride] = w[c];
  } else if (NWR > 0 && i - cx.ilds < NWR) {
    const int q = i - cx.ilds;
    static_for<NWR>([&](auto Q) {
      constexpr int qq = decltype(Q)::value;
      if (q == qq) {
#pragma unroll
        for (int c = 0; c < K; ++c) wres[qq][c] = w[c];
      }
    });
  } else {
    rl_store_w_global<K>(cx, w, wbase);
  }
}

// residual of the lane's row: sse[j] += (x - w.h)^2, xsq[j] += x^2   (H in VGPRs: only inside residual passes)
template <int K, int LOSS = 0>
__device__ __forceinline__ void rl_resid(const float (&x)[16], const float (&w)[K], const float (&h)[K][16],
                                         float (&sse)[16], float (&xsq)[16], float& kl) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float rec = w[0] * h[0][j];
#pragma unroll
    for (int c = 1; c < K; ++c) rec = fma_(w[c], h[c][j], rec);
    const float d = x[j] - rec;
    sse[j] = fma_(d, d, sse[j]);
    xsq[j] = fma_(x[j], x[j], xsq[j]);
    if constexpr (LOSS == 1) {  // as resid_tile (nmf_kernels.hpp): x log(x / wh) - x + wh, branch-free
      const float whc = rec < eps_val<float>() ? eps_val<float>() : rec;
      const float xs = x[j] > eps_val<float>() ? x[j] : eps_val<float>();
      const float lg = fma_(x[j], log_(xs / whc), rec - x[j]);
      kl += (x[j] > eps_val<float>()) ? lg : rec;
    }
  }
}

#ifndef HIPNMF_RL_THREADS
#define HIPNMF_RL_THREADS 512  // 256: one wave per SIMD with up to 512 VGPRs (experiment: more resident tiles)
#endif
template <int K, int NXR = rl_nxr<K>(), int NWR = rl_nwr<K>(), int PF = (HIPNMF_RL_PF), int LOSS = 0>
__global__ void __launch_bounds__(HIPNMF_RL_THREADS) fit_rowlane_kernel(SolveArgs<float> a) {
  using C = Cfg<float, 1, 16, K>;
  constexpr int MP = 16, NB = C::NB;
  constexpr int NXA = NXR > 0 ? NXR : 1, NWA = NWR > 0 ? NWR : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nw = blockDim.x / WAVE;
  Smem<float, 1, 16, K> s(smem_raw, nw);
  const int b = blockIdx.x;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const float* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  float* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  float* __restrict__ Hb = a.H + (long long)b * K * a.m;
  int T = a.T;
  long long ldw = a.ldw;
  if (a.ragged) {  // packed batch of matrices with different numbers of rows: {T_b, row-major X offset, ldw, W offset}
    const long long* d = a.ragged + 4LL * b;
    T = (int)d[0];
    Xb = a.X + d[1];
    ldw = d[2];
    Wb = a.W + d[3];
  }
  const int m = a.m;
  const int row_end = ((T + WAVE - 1) / WAVE) * WAVE;
  const int stride = nw * WAVE;
  const int ntiles = row_end / WAVE;
  const int ntw = wave < ntiles ? (ntiles - wave + nw - 1) / nw : 0;  // tiles of this wave: rows (i nw + wave) 64 ...

  // W cache: rows [0, lds_rows) stay in LDS for the whole fit (lds_stride = launch-wide capacity)
  float* lds_w = reinterpret_cast<float*>(smem_raw + ((Smem<float, 1, 16, K>::bytes(nw) + 15) / 16) * 16);
  const int lds_stride = a.lds_rows;
  const int t_blk = (int)(((long long)T + blockDim.x - 1) / blockDim.x * blockDim.x);
  const int lds_rows = lds_stride < t_blk ? lds_stride : t_blk;

  RlCtx<K> cx;
  cx.xrow_b = (unsigned)(a.ldx * 4LL);
  cx.xr = make_rsrc(Xb, (unsigned)((long long)(T + 64) * a.ldx * 4LL));
  cx.wr = make_rsrc(Wb, (unsigned)((long long)K * ldw * 4LL));
  cx.xvoff = (unsigned)lane * cx.xrow_b;
  cx.woff = (unsigned)lane * 4u;
  cx.ldw_b = (unsigned)(ldw * 4LL);
  cx.T = T;
  cx.lane = lane;
  cx.wave = wave;
  cx.stride = stride;
  cx.lds_w = lds_w;
  cx.lds_stride = lds_stride;
  // tiles i < ilds of THIS wave (rows (i nw + wave) 64 ...) lie inside the cached rows [0, lds_rows), a multiple of 64
  cx.ilds = wave < lds_rows / WAVE ? (lds_rows / WAVE - wave + nw - 1) / nw : 0;
  // row base of tile i of this wave.  The index is made opaque so that the bases (and everything derived from them)
  // of the statically unrolled resident tiles are recomputed with two SALU instructions where they are used
  // instead of being hoisted out of the iteration loop into dozens of SGPRs (which then spill).
  auto tile_base = [&](int i) __attribute__((always_inline)) {
    asm volatile("" : "+s"(i));
    return (i * nw + wave) * WAVE;
  };

  for (int t0 = 0; t0 < lds_rows; t0 += blockDim.x) {  // row t0 + tid is owned by this thread in every pass
    const int t = t0 + threadIdx.x;
    if (t < lds_rows) {
#pragma unroll
      for (int c = 0; c < K; ++c) lds_w[c * lds_stride + t] = (t < T) ? Wb[(long long)c * ldw + t] : 0.f;
    }
  }
  // register-resident state: the first NXR tiles of X, and the NWR tiles of W that follow the LDS cache
  float xres[NXA][16], wres[NWA][K];
  static_for<NXA>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    if constexpr (NXR > 0) rl_load_x<K>(cx, xres[q], tile_base(q), q < ntw);
  });
  static_for<NWA>([&](auto Q) {
    constexpr int q = decltype(Q)::value;
    if constexpr (NWR > 0) {
      const int i = cx.ilds + q;
      if (i < ntw) {
        rl_load_w_global<K>(cx, wres[q], tile_base(i));
      } else {
#pragma unroll
        for (int c = 0; c < K; ++c) wres[q][c] = 0.f;
      }
    }
  });

  load_h_to_lds(s, Hb, m);
  __syncthreads();
  if constexpr (LOSS == 1)
    compute_hsum(s);
  else
    compute_hht(s);
  __syncthreads();
  RlHops ho;
  RlKlOps ko;
  float hht[K][K], hsum[K];
  // operands of the tile arithmetic, rebuilt from LDS after every H update (KL: s.HHt holds rowsum(H), not H H^T)
  auto load_ops = [&]() __attribute__((always_inline)) {
    if constexpr (LOSS == 1) {
      const int lo = lane & 3, hi = lane >> 2;
      ho.hq0 = (lo < K) ? s.H[lo * 16 + hi] : 0.f;
      ho.hq1 = (4 + lo < K) ? s.H[(4 + lo) * 16 + hi] : 0.f;
      ho.hhq0 = ho.hhq1 = 0.f;
      rl_load_klops<K>(s, lane, ko, hsum);
    } else {
      rl_load_hops<K>(s, lane, ho, hht);
    }
  };
  load_ops();
  auto update = [&](const float (&x)[16], float (&w)[K], float (&accA)[K][16], float (&accB)[NB], bool upd_)
                    __attribute__((always_inline)) {
    if constexpr (LOSS == 1)
      rl_update_kl<K>(x, w, ho, ko, hsum, accA, accB, a.l1w, a.l2w, upd_);
    else
      rl_update<K>(x, w, ho, hht, accA, accB, a.l1w, a.l2w, upd_);
  };

  // ||X - W H||_F^2 per column (+ sum X^2) of the whole matrix -> s.part[0 .. 2 MP); barriers inside
  // (always_inline: a real call would force every captured register array into scratch memory)
  auto block_resid = [&]() __attribute__((always_inline)) {
    float h[K][16], sse[16], xsq[16], kl = 0.f;
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) h[c][j] = s.H[c * MP + j];
#pragma unroll
    for (int j = 0; j < 16; ++j) sse[j] = xsq[j] = 0.f;
    static_for<NXA>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      if constexpr (NXR > 0) {
        if (q < ntw) {
          float w[K];
          rl_get_w<K, NWR>(cx, w, wres, q, tile_base(q));
          rl_resid<K, LOSS>(xres[q], w, h, sse, xsq, kl);
        }
      }
    });
    for (int i = NXR; i < ntw; ++i) {
      float x[16], w[K];
      rl_load_x<K>(cx, x, tile_base(i), true);
      rl_get_w<K, NWR>(cx, w, wres, i, tile_base(i));
      rl_resid<K, LOSS>(x, w, h, sse, xsq, kl);
    }
    float v[32];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      v[j] = sse[j];
      v[16 + j] = xsq[j];
    }
    wave_reduce_scatter<1, 32, float>(v, lane);  // lane l < 32: sum over the wave of value l
    if constexpr (LOSS == 1) {
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) kl += __shfl_xor(kl, off, WAVE);
    }
    __syncthreads();                             // s.part may still be read by a previous phase
    if (lane < 32) s.part[wave * 33 + lane] = v[0];
    if (LOSS == 1 && lane == 0) s.part[wave * 33 + 32] = kl;
    __syncthreads();
    if (threadIdx.x < 33) {
      float acc = s.part[threadIdx.x];
      for (int w2 = 1; w2 < nw; ++w2) acc += s.part[w2 * 33 + threadIdx.x];
      s.part[threadIdx.x] = acc;
    }
    __syncthreads();
  };
  auto error_from_part = [&]() __attribute__((always_inline)) -> float {
    if constexpr (LOSS == 1) {  // sqrt(2 KL(X || WH)) (_nmf.py:185-189)
      const float d = s.part[2 * MP];
      return sqrt_(2.f * (d > 0.f ? d : 0.f));
    }
    float tot = 0.f;
    for (int j = 0; j < MP; ++j) tot += s.part[j];
    return sqrt_(tot);
  };
  auto residual = [&]() __attribute__((always_inline)) -> float {
    block_resid();
    const float e = error_from_part();
    __syncthreads();  // the next iteration writes its wave records into s.part without another barrier
    return e;
  };

  float err0 = 0.f, prev = 0.f;
  if (a.tol > 0.f) {
    err0 = residual();
    prev = err0;
  }
  const bool upd = a.update_h != 0;
  int n_iter = 0;
  float xa[16], xb[16], wga[K], wgb[K];
  rl_prefetch<K, NWR>(cx, xa, wga, NXR, tile_base(NXR), NXR < ntw);
  for (int it = 1; it <= a.max_iter; ++it) {
    n_iter = it;
    float accA[K][16], accB[NB];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) accA[c][j] = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) accB[i] = 0.f;

    // tiles whose X lives in registers
    static_for<NXA>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      if constexpr (NXR > 0) {
        if (q < ntw) {
          float w[K];
          const int wb = tile_base(q);
          rl_get_w<K, NWR>(cx, w, wres, q, wb);
          update(xres[q], w, accA, accB, upd);
          rl_put_w<K, NWR>(cx, w, wres, q, wb);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    // streamed tiles: xa (and wga for the tail rows of W) holds, or has in flight, tile i
    int i = NXR;
    if constexpr (PF >= 2) {
      for (; i + 1 < ntw; i += 2) {
        float w[K];
        rl_prefetch<K, NWR>(cx, xb, wgb, i + 1, tile_base(i + 1), true);
        int wb = tile_base(i);
        rl_get_w<K, NWR, true>(cx, w, wres, i, wb, wga);
        update(xa, w, accA, accB, upd);
        rl_put_w<K, NWR>(cx, w, wres, i, wb);
        __builtin_amdgcn_sched_barrier(0);
        rl_prefetch<K, NWR>(cx, xa, wga, i + 2, tile_base(i + 2), i + 2 < ntw);
        wb = tile_base(i + 1);
        rl_get_w<K, NWR, true>(cx, w, wres, i + 1, wb, wgb);
        update(xb, w, accA, accB, upd);
        rl_put_w<K, NWR>(cx, w, wres, i + 1, wb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i < ntw) {
        float w[K];
        const int wb = tile_base(i);
        rl_get_w<K, NWR, true>(cx, w, wres, i, wb, wga);
        update(xa, w, accA, accB, upd);
        rl_put_w<K, NWR>(cx, w, wres, i, wb);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (; i < ntw; ++i) {
        float w[K];
        const int wb = tile_base(i);
        rl_get_w<K, NWR, true>(cx, w, wres, i, wb, wga);
        update(xa, w, accA, accB, upd);
        rl_put_w<K, NWR>(cx, w, wres, i, wb);
        rl_prefetch<K, NWR>(cx, xa, wga, i + 1, tile_base(i + 1), i + 1 < ntw);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // X does not depend on H (and the first streamed tile's W, if it streams at all, is final): start the next
    // iteration's first tile now, under the reduction
    if (it < a.max_iter) rl_prefetch<K, NWR>(cx, xa, wga, NXR, tile_base(NXR), NXR < ntw);
    if (upd) {
      // s.part was last read before the previous iteration's second barrier (or by a residual pass that ends with
      // a barrier), so the records can be written right away: two workgroup barriers per iteration
      wave_reduce_acc<float, 1, 16, K>(s.part + wave * C::NACC, accA, accB);
      __syncthreads();
      if (wave == 0) {
        if constexpr (LOSS == 1)
          wave0_combine_and_update_h_kl(s, nw, m, a.l1h, a.l2h);
        else
          wave0_combine_and_update_h(s, nw, m, a.l1h, a.l2h);
      }
      __syncthreads();
      load_ops();
    }
    if (a.tol > 0.f && (it % a.check_every) == 0) {
      const float err = residual();
      if ((prev - err) / err0 < a.tol) break;
      prev = err;
      rl_prefetch<K, NWR>(cx, xa, wga, NXR, tile_base(NXR), NXR < ntw);  // the residual pass used the buffers
    }
  }
  // reconstruction_err_ (_nmf.py:1628-1630) + per-column SSE / sum X^2 for VAF (analysis.py:654-662)
  block_resid();
  if (threadIdx.x == 0) {
    if (a.err_out) a.err_out[b] = error_from_part();
    if (a.n_iter_out) a.n_iter_out[b] = n_iter;
  }
  if ((int)threadIdx.x < m) {
    if (a.sse_col_out) a.sse_col_out[(long long)b * m + threadIdx.x] = s.part[threadIdx.x];
    if (a.xsq_col_out) a.xsq_col_out[(long long)b * m + threadIdx.x] = s.part[MP + threadIdx.x];
  }
  if (a.update_h) {
    for (int i2 = threadIdx.x; i2 < K * MP; i2 += blockDim.x) {
      const int c = i2 / MP, j = i2 % MP;
      if (j < m) Hb[c * m + j] = s.H[i2];
    }
  }
  for (int t0 = 0; t0 < lds_rows; t0 += blockDim.x) {  // write the cached rows of W back
    const int t = t0 + threadIdx.x;
    if (t < T && t < lds_rows) {
#pragma unroll
      for (int c = 0; c < K; ++c) Wb[(long long)c * ldw + t] = lds_w[c * lds_stride + t];
    }
  }
  static_for<NWA>([&](auto Q) {  // and the register-resident ones
    constexpr int q = decltype(Q)::value;
    if constexpr (NWR > 0) {
      const int i = cx.ilds + q;
      if (i < ntw) rl_store_w_global<K>(cx, wres[q], tile_base(i));
    }
  });
}


// =================================================================================================
// slice_pass_rowlane_kernel<K>: the time-shard / row-sliced pass (hipnmf_shard_pass_f32; BASELINE config #5) with
// the row-per-lane matrix-pipe arithmetic, reading the CHANNEL-MAJOR X and component-major W of the shard ABI in
// place.  Every lane owns FOUR consecutive rows: per channel one 16-byte load per lane = 1 KB contiguous per wave
// instruction (round 1's (G=4, CH=4) mapping touched four channels x 256 bytes per instruction, 26 streams of
// 256-byte pieces per workgroup, and reached 0.59 of the HBM line); W is read and written back the same way.  The
// four rows are then updated one after the other with rl_update.  Output: one record of [W^T X | W^T W] sums per
// slice in a.part, the same layout slice_pass_kernel<float, 4, 4, K> writes (reduce_slices / hupdate are unchanged).
// Requirements (checked by the caller): fp32, 9..16 channels, ldx % 4 == 0, T % 4 == 0, 16-byte aligned X and W,
// rows_per_slice % 256 == 0.
__device__ __forceinline__ void buf_store4(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[4]) {
  using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
  u32x4 u;
  __builtin_memcpy(&u, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
}

// Cache policy of the time-shard pass (buf_load: 2 = non-temporal).  A shard (2.6 GB of X per 25M rows) is read once per
// iteration and nothing of it survives in any cache until the next one.  tools/shard_bench.py, 2.5e7 rows, k = 5, per pass:
// default policy 0.541 ms (4.81 TB/s algorithmic), X non-temporal 0.526 ms (4.94), X and W non-temporal 0.537 ms.
#ifndef HIPNMF_SHARD_X_AUX
#define HIPNMF_SHARD_X_AUX 2
#endif
#ifndef HIPNMF_SHARD_W_AUX
#define HIPNMF_SHARD_W_AUX 0
#endif
template <int K>
__global__ void __launch_bounds__(256) slice_pass_rowlane_kernel(SolveArgs<float> a) {
  using C = Cfg<float, 1, 16, K>;
  constexpr int NB = C::NB, R = 4, STEP = WAVE * R;  // rows per wave-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nw = blockDim.x / WAVE;
  Smem<float, 1, 16, K> s(smem_raw, nw);
  const int b = blockIdx.y, sl = blockIdx.x;
  if (a.state && a.state[(long long)b * 8 + 3] != 0.f) return;  // matrix already converged
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const float* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  float* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  const float* __restrict__ Hb = a.H + (long long)b * K * a.m;
  const int m = a.m, T = a.T;
  const int row_begin = sl * a.rows_per_slice;
  int row_end = row_begin + a.rows_per_slice;
  if (row_end > T) row_end = T;

  load_h_to_lds(s, Hb, m);
  __syncthreads();
  compute_hht(s);
  __syncthreads();
  RlHops ho;
  float hht[K][K];
  rl_load_hops<K>(s, lane, ho, hht);
  float accA[K][16], accB[NB];
#pragma unroll
  for (int c = 0; c < K; ++c)
#pragma unroll
    for (int j = 0; j < 16; ++j) accA[c][j] = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i) accB[i] = 0.f;

  const rsrc_t xr = make_rsrc(Xb, (unsigned)((long long)m * a.ldx * 4LL));
  const rsrc_t wr = make_rsrc(Wb, (unsigned)((long long)K * a.ldw * 4LL));
  const unsigned ldx_b = (unsigned)(a.ldx * 4LL), ldw_b = (unsigned)(a.ldw * 4LL);
  const bool upd = a.update_h != 0;
  for (int wbase = row_begin + wave * STEP; wbase < row_end; wbase += nw * STEP) {
    // the lane's four rows wbase + 4 lane .. + 3 are inside the matrix together or not at all (T % 4 == 0)
    const unsigned v = (wbase + R * lane < row_end) ? (unsigned)(R * lane) * 4u : OOB;
    const unsigned sb = (unsigned)wbase * 4u;
    float x4[16][R], w4[K][R];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < m) {
        buf_load<float, R, (HIPNMF_SHARD_X_AUX)>(xr, v, sb + (unsigned)j * ldx_b, x4[j]);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) x4[j][r] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < K; ++c) buf_load<float, R, (HIPNMF_SHARD_W_AUX)>(wr, v, sb + (unsigned)c * ldw_b, w4[c]);
    static_for<R>([&](auto RR) {
      constexpr int r = decltype(RR)::value;
      float x[16], w[K];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = x4[j][r];
#pragma unroll
      for (int c = 0; c < K; ++c) w[c] = w4[c][r];
      rl_update<K>(x, w, ho, hht, accA, accB, a.l1w, a.l2w, upd);
#pragma unroll
      for (int c = 0; c < K; ++c) w4[c][r] = w[c];
    });
#pragma unroll
    for (int c = 0; c < K; ++c) buf_store4(wr, v, sb + (unsigned)c * ldw_b, w4[c]);
  }
  if (!upd) return;
  wave_reduce_acc<float, 1, 16, K>(s.part + wave * C::NACC, accA, accB);
  __syncthreads();
  float* __restrict__ out = a.part + ((long long)b * a.S + sl) * C::NACC;
  for (int i = threadIdx.x; i < C::NACC; i += blockDim.x) {
    float acc = s.part[i];
    for (int w2 = 1; w2 < nw; ++w2) acc += s.part[w2 * C::NACC + i];
    out[i] = acc;
  }
}

}  // namespace hipnmf
