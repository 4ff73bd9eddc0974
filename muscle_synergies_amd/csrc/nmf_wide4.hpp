// nmf_wide4.hpp -- the matrix-pipe solver for few components (fp32, n_components <= 8; 33..128 channels, and batches of narrower
// matrices where hipnmf_api.hip::wide_preferred measured it faster than the lane mappings: MP = 16 / 32): the same
// iteration as fit_wide_kernel (nmf_wide.hpp), with every contraction on v_mfma_f32_4x4x1_16b_f32 instead of
// v_mfma_f32_16x16x4_f32.
//
// Why a second formulation.  The 16x16x4 tile pads the components to 16: at k = 8 half of every MFMA multiplies zeros, and
// the matrix pipe is what fit_wide_kernel<float,64,16,..> is bound by at k <= 8 (profiles/r03_pmc_fit_wide_64_8.txt: 40 MFMAs
// = 1 280 pipe cycles per 16-row subtile, 56 % of all SIMD cycles; the kernel runs 18.3 G rows/s whether W streams from
// memory or sits in LDS entirely).  The 4x4x1 form (16 independent 4 x 4 outer products per instruction, K = 1) pads the
// components to a multiple of four only: 72 MFMAs of 8 cycles = 576 cycles per subtile at 64 channels, k = 8.
//
// Arithmetic replaced: sklearn/decomposition/_nmf.py (1.7.2) _multiplicative_update_w (:526-631), _multiplicative_update_h
// (:634-728), _beta_divergence (:85-134), loop + stop rule (:731-893), reached from the reference at
// src/muscle_synergies/analysis.py:862-863.  sklearn notation: X (T x m) ~ W (T x k) H (k x m).
//
// v_mfma_f32_4x4x1_16b_f32:  D_blk[i][j] += A_blk[i] B_blk[j]  for 16 blocks; lane 4 blk + i holds A_blk[i], lane 4 blk + j
// holds B_blk[j] and D_blk[0..3][j] (4 registers).  CBSZ = 4 broadcasts the A values of block ABID to all blocks.
// A wave owns 16-row subtiles.  KQ = KP / 4 component quads, lane = 16 rq + 4 p + jj:
//   numerator     lane (row r = 4 rq + jj, part p): B = X[r][16 cb + 4 p + e], one 16-byte LDS read per 16-channel block cb;
//                 A (per-lane registers, loaded once per iteration) = H[comp][same channel].  The four lanes of a block are
//                 four rows at the same channels, as the instruction requires.  Each lane ends with the partial sums over
//                 ITS quarter of the channels for all KP components; the components are named lane-relatively (register f <->
//                 component KQ ((p + f / KQ) mod 4) + f mod KQ -- free, the A registers are filled that way), so that three DPP
//                 row rotations per value reduce-scatter them: lane (r, p) ends with components KQ p .. KQ p + KQ - 1 of row r.
//   denominator   (H H^T) W^T the same way, the contraction split over the parts: lane (r, p) multiplies ITS KQ values of W.
//   W <- W * num / den   KQ values per lane: ONE load and ONE store of KQ floats per lane and subtile (512 contiguous bytes of
//                 the row-major W per subtile at k = 8), no value is computed twice
//   W^T X         lanes = channels: B = X[row s][lane] (16 dword reads from the stage), A = the new W of row s broadcast
//                 from block s (lane 4 s + i <-> W[s][4 cg + i], read back from a 16 x KP per-wave stage); D: lane = channel,
//                 register i <-> component 4 cg + i -- final layout, 4 KQ accumulators per 64 channels
//   W^T W         blocks = rows: A = B = that same W register; summed over the blocks once per pass
//   residual      lanes = channels: A = W[row][c] broadcast per row quad, B = H[c][lane]; per-column sums need no reduction
// X goes HBM -> registers (16-byte loads, whole rows, non-temporal) -> per-wave LDS stage -> both operand layouts, as in
// nmf_wide.hpp; W cache in LDS, padding rules, descriptors and the pass structure are that kernel's too.
#pragma once
#include "nmf_wide.hpp"

namespace hipnmf {

template <int MP, int KQ>
struct Wide4Cfg {
  static constexpr int KP = 4 * KQ;
  static constexpr int NH = (MP + 63) / 64;  // 64-lane groups of the lanes-are-channels layouts
  // lanes-are-channels layouts below 64 channels: LP lanes per row, NG = 64 / LP rows per instruction (CBSZ = log2(LP / 4)
  // broadcasts block ABID of each group of LP / 4 blocks)
  static constexpr int LP = MP <= 16 ? 16 : MP <= 32 ? 32 : 64;
  static constexpr int NG = 64 / LP;
  static constexpr int CB = LP == 16 ? 2 : LP == 32 ? 3 : 4;
  static constexpr int NCB = MP / 16;
  static constexpr int CPR = MP / 4;  // 16-byte pieces per row
  static constexpr int RPL = wide_pow2_floor(64 / CPR) > 16 ? 16 : wide_pow2_floor(64 / CPR);
  static constexpr int NLD = 16 / RPL;
  // row stride of the X stage = 16 (mod 64) words: the numerator's 16-byte reads (rows 4 rq + jj at channel offset 4 p) then
  // hit 16 distinct four-bank groups per quarter wave
  static constexpr int SX = ((MP + 47) / 64) * 64 + 16;
  static constexpr int SW = KP == 4 ? 4 : KP == 8 ? 12 : KP == 12 ? 12 : 20;  // row stride of the W stage (conflict-free dword reads)
  static constexpr int SH = MP + 4;                                         // row stride of H in LDS
  static constexpr int XS = 16 * SX, WS = 16 * SW;
  static constexpr int REC = KP * MP + KP * KP;  // per-wave record of [W^T X | W^T W]
  static constexpr int PERWAVE = (XS + WS > REC ? XS + WS : REC);
  static constexpr int COMMON = KP * SH + KP * KP + KP * MP + KP * KP + 2 * MP + 8;
  static_assert(MP % 16 == 0 && MP >= 16 && MP <= 128 && KQ >= 1 && KQ <= 4, "unsupported wide4 shape");
  static_assert(SX >= LP * NH, "the lanes-are-channels reads stay inside a stage row");
  __host__ __device__ static constexpr size_t smem_bytes(int nw) { return sizeof(float) * (size_t)(COMMON + nw * PERWAVE); }
};

using w4f4 = float __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__device__ __forceinline__ w4f4 w4_mfma(float a, float b, w4f4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0);
}
// value of the lane 4 D lanes below (mod 16) in the same row of 16 lanes: DPP row_ror
template <int D>
__device__ __forceinline__ float w4_from_part_below(float v) {
  static_assert(D >= 1 && D <= 3, "");
  return dpp_mov<0x120 + 4 * D>(v);
}
template <int N>
__device__ __forceinline__ void w4_store(rsrc_t r, unsigned voff, const float (&v)[N]) {
  if constexpr (N == 1) {
    buf_store<float>(r, voff, 0u, v[0]);
  } else if constexpr (N == 2) {
    using u32x2 = unsigned int __attribute__((ext_vector_type(2)));
    u32x2 u;
    __builtin_memcpy(&u, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(u, r, voff, 0u, 0);
  } else if constexpr (N == 3) {
    using u32x3 = unsigned int __attribute__((ext_vector_type(3)));
    u32x3 u;
    __builtin_memcpy(&u, &v, 12);
    __builtin_amdgcn_raw_buffer_store_b96(u, r, voff, 0u, 0);
  } else {
    using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
    u32x4 u;
    __builtin_memcpy(&u, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0u, 0);
  }
}

template <int MP, int KQ>
struct Wide4Tile {
  using C = Wide4Cfg<MP, KQ>;
  float xg[C::NLD][4];  // the subtile of X as loaded: piece (lane % CPR) of row n RPL + lane / CPR
  float w[KQ];          // W[row r][KQ p ..]
};

// NSET: register sets of subtile loads a wave keeps in flight beyond the one being worked on (see fit_wide_kernel)
// LOSS: 0 = Frobenius, 1 = Kullback-Leibler (beta_loss = 1; _nmf.py:556-591, 642-684; round 4): W *= ((X / WH) H^T) / rowsum(H),
// H *= (W'^T (X / W'H)) / colsum(W') with W' the updated W.  Both reconstructions run on the pipe in the numerator's own layout:
// lane (row r, part p) gathers its row's other component quads from the three lanes "below" (DPP row rotations) and multiplies
// by H^T read from a transposed copy in LDS (sA, idle during a pass), D: lane (r, p), register e <-> W H [r][16 cb + 4 p + e] --
// exactly where the lane's X values sit, so Q = X / max(W H, eps) IS the numerator's B operand; Q' goes over X in the stage and
// W'^T Q' is the Frobenius W^T X (every lanes-are-channels layout: 16 / 32 / 64 lanes per row; round 5: 17..32 channels too).
template <int MP, int KQ, int NW, int NSET, int LOSS = 0>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 8))) fit_wide4_kernel(WideArgs<float> a) {
  using C = Wide4Cfg<MP, KQ>;
  using Tile = Wide4Tile<MP, KQ>;
  constexpr int KP = C::KP, NH = C::NH, NCB = C::NCB, SX = C::SX, SW = C::SW, SH = C::SH, NLD = C::NLD, RPL = C::RPL,
                CPR = C::CPR, NT = NW * 64, LP = C::LP, NG = C::NG, CB = C::CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const sH = reinterpret_cast<float*>(smem_raw);  // [KP][SH]
  float* const sHHt = sH + KP * SH;                      // [KP][KP]
  float* const sA = sHHt + KP * KP;                      // [KP][MP]   W^T X summed over the waves
  float* const sB = sA + KP * MP;                        // [KP][KP]   W^T W   (directly behind sA: one index space)
  float* const sPart = sB + KP * KP;                     // [2 MP + 8] per-column sse | xsq of the residual pass
  float* const wv0 = sPart + 2 * MP + 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int ii = lane & 3, blk = lane >> 2, p = blk & 3, rq = blk >> 2, r = 4 * rq + ii;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const xs = wv0 + wave * C::PERWAVE;  // [16][SX] this wave's X stage
  float* const wst = xs + C::XS;              // [16][SW] this wave's W stage

  const int b = blockIdx.x;
  const float* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  float* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  float* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
  int T = a.T;
  if (a.ragged) {
    const long long* d = a.ragged + 4LL * b;
    T = (int)d[0];
    Xb = a.X + d[1];
    Wb = a.W + d[3];
  }
  const int m = a.m, k = a.k;  // (a.ks == KP: the host picks KQ = ks / 4)
  const int slice = blockIdx.y;
  if (a.mode != 0) {  // row-sliced mode (nmf_wide.hpp): a slice is a matrix of its own for everything row-local
    if (a.state && a.state[(long long)b * 8 + 3] != 0.0f) return;
    const int row_begin = slice * a.rows_per_slice;
    int rows = T - row_begin;
    if (rows > a.rows_per_slice) rows = a.rows_per_slice;
    if (rows <= 0) rows = 0;
    Xb += (long long)row_begin * a.ldx;
    Wb += (long long)row_begin * KP;
    T = rows;
  }
  const int ntiles = (T + 15) / 16;
  float* const wcache = wv0 + NW * C::PERWAVE;  // [lds_rows][KP]
  const int ncached = (a.lds_rows / 16 < ntiles) ? a.lds_rows / 16 : ntiles;
  const unsigned ldx_b = (unsigned)(a.ldx * 4LL);
  constexpr unsigned ldw_b = (unsigned)KP * 4u;

  // X: whole rows, 16-byte pieces; rows beyond the matrix are masked by the per-subtile descriptors (nmf_wide.hpp)
  const int xl_row = lane / CPR, xl_chunk = lane % CPR;
  const bool xl_active = lane < RPL * CPR;
  unsigned xvoff[NLD];
#pragma unroll
  for (int n = 0; n < NLD; ++n)
    xvoff[n] = (xl_active && xl_chunk < a.xchunks) ? (unsigned)(n * RPL + xl_row) * ldx_b + (unsigned)xl_chunk * 16u : OOB;
  float* const xs_put = xs + xl_row * SX + xl_chunk * 4;
  const unsigned wvoff = (unsigned)((r * KP + KQ * p) * 4);         // this lane's KQ values of its row
  float* const wc_lane = wcache + r * KP + KQ * p;
  const char* const xbase = reinterpret_cast<const char*>(Xb);
  char* const wbase = reinterpret_cast<char*>(Wb);
  auto x_rsrc = [&](int i) __attribute__((always_inline)) {
    const int rows = i < ntiles ? T - 16 * i : 0;
    return make_rsrc(xbase + (long long)(rows > 0 ? 16 * i : 0) * ldx_b, (unsigned)rows * ldx_b);
  };
  auto w_rsrc = [&](int i) __attribute__((always_inline)) {  // (subtiles cached in LDS: empty, their loads move nothing)
    const int rows = (i < ntiles && i >= ncached) ? T - 16 * i : 0;
    return make_rsrc(wbase + (long long)(rows > 0 ? 16 * i : 0) * ldw_b, (unsigned)rows * ldw_b);
  };
  auto issue = [&](Tile& t, int i) __attribute__((always_inline)) {
    const rsrc_t xr = x_rsrc(i);
    const rsrc_t wr = w_rsrc(i);
#pragma unroll
    for (int n = 0; n < NLD; ++n) buf_load<float, 4, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[n], 0u, t.xg[n]);
    buf_load<float, KQ>(wr, wvoff, 0u, t.w);
  };
  auto stage_x = [&](const Tile& t) __attribute__((always_inline)) {
    if (xl_active) {
#pragma unroll
      for (int n = 0; n < NLD; ++n) wide_lds_write<float, 4>(xs_put + n * RPL * SX, t.xg[n]);
    }
    wide_wave_lds_fence();
  };

  // ---- the cached rows of W -> LDS, H -> LDS (zero padded), H H^T ---------------------------------------------------
  for (int idx = tid; idx < ncached * 16 * KP; idx += NT) wcache[idx] = (idx < T * KP) ? Wb[idx] : 0.0f;
  for (int idx = tid; idx < KP * SH; idx += NT) {
    const int c = idx / SH, jj = idx % SH;
    sH[idx] = (c < k && jj < m) ? Hb[c * m + jj] : 0.0f;
  }
  __syncthreads();
  auto compute_hht_lds = [&]() __attribute__((always_inline)) {  // call between barriers
    if constexpr (LOSS == 1) {  // rowsum(H), the W update's denominator (_nmf.py:577-581), in sHHt[0 .. KP); H^T [MP][KP] in sA
      for (int c = tid; c < KP; c += NT) {
        float s = 0.0f;
        for (int jj = 0; jj < MP; ++jj) s += sH[c * SH + jj];
        sHHt[c] = s;
      }
      for (int idx = tid; idx < KP * MP; idx += NT) sA[idx] = sH[(idx % KP) * SH + idx / KP];
      return;
    }
    for (int idx = tid; idx < KP * KP; idx += NT) {
      const int c = idx / KP, c2 = idx % KP;
      float s = 0.0f;
      for (int jj = 0; jj < MP; ++jj) s = fma_(sH[c * SH + jj], sH[c2 * SH + jj], s);
      sHHt[idx] = s;
    }
  };
  compute_hht_lds();
  __syncthreads();

  // A operands that change once per iteration.  Register f = 4 cg + i of a numerator / denominator accumulator means
  // component comp_of(f) = KQ ((p + f / KQ) mod 4) + f mod KQ in this lane.
  int comp_a[KQ];  // component this lane's A value (i = lane & 3) stands for in quad cg
#pragma unroll
  for (int cg = 0; cg < KQ; ++cg) {
    const int f = 4 * cg + ii;
    comp_a[cg] = KQ * ((p + f / KQ) & 3) + f % KQ;
  }
  float hsum[KQ];        // KL: rowsum(H) of this lane's components KQ p + e
  float hA[KQ][NCB][4];  // H[comp_a[cg]][16 cb + 4 p + e]
  float hhA[KQ][KQ];     // HHt[comp_a[cg]][KQ p + e]
  auto load_operands = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) wide_lds_read<float, 4>(sH + comp_a[cg] * SH + 16 * cb + 4 * p, hA[cg][cb]);
      if constexpr (LOSS == 0) {
#pragma unroll
        for (int e = 0; e < KQ; ++e) hhA[cg][e] = sHHt[comp_a[cg] * KP + KQ * p + e];
      }
    }
    if constexpr (LOSS == 1) {
#pragma unroll
      for (int e = 0; e < KQ; ++e) hsum[e] = sHHt[KQ * p + e];
    }
  };
  load_operands();
  float csum[KQ];  // KL: this lane's share of colsum(W') (its row, its components)
  // KL: W H for the lane's row and channels 16 cb + 4 p .. + 3 from a fragment (lane (r, p): components KQ p + e): the other quads'
  // values come from the lanes D parts below (component KQ ((p - D) mod 4) + e), the matching H^T entries from sA
  auto wh_block4 = [&](const float (&w)[KQ], int cb) __attribute__((always_inline)) -> w4f4 {
    float wr[4][KQ];
#pragma unroll
    for (int e = 0; e < KQ; ++e) {
      wr[0][e] = w[e];
      wr[1][e] = w4_from_part_below<1>(w[e]);
      wr[2][e] = w4_from_part_below<2>(w[e]);
      wr[3][e] = w4_from_part_below<3>(w[e]);
    }
    w4f4 rec = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* ht = sA + (16 * cb + 4 * p + ii) * KP;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float hv[KQ];
      wide_lds_read<float, KQ>(ht + KQ * ((p - d) & 3), hv);
#pragma unroll
      for (int e = 0; e < KQ; ++e) rec = w4_mfma<0, 0>(hv[e], wr[d][e], rec);
    }
    return rec;
  };
  auto kl_quot4 = [&](float x, float wh) __attribute__((always_inline)) -> float {  // X / max(WH, EPSILON) (_nmf.py:574-575)
    return kl_quot(x, kl_floor(wh));  // (nmf_kernels.hpp: one v_max, the bare reciprocal)
  };

  w4f4 accA[NH][KQ], accB[KQ][KQ];
  const w4f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

  // W'^T X (KL: W'^T Q') of the subtile in the stage, with the new rows of W in `wst`; leaves the A operand in `wa`
  float wa[KQ];  // lane 4 blk + i <-> W'[row of blk][4 cg + i]
  auto accumulate_wtx = [&]() __attribute__((always_inline)) {
    // block blk stands for row blk (64 channel lanes: one row per instruction, broadcast from block s) or, with LP < 64 lanes
    // per row, for row NG (blk mod LP / 4) + blk / (LP / 4): NG rows per instruction, lanes [G LP, G LP + LP) row NG s + G,
    // CBSZ broadcasts block s of each group to the group's LP / 4 blocks
    const int wrow = NG * (blk % (LP / 4)) + blk / (LP / 4);
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) wa[cg] = wst[wrow * SW + 4 * cg + ii];
    // W^T X: lanes are channels, the row's W broadcast from block s
    // (all sixteen rows are requested before the first product: left to itself the compiler reads two, waits, multiplies, reads
    //  the next two ... and every wait exposes a full LDS round trip)
    const float* xcol = xs + lane;
    if constexpr (LP < 64) {
      float xc[16 / NG];
#pragma unroll
      for (int s = 0; s < 16 / NG; ++s) xc[s] = xs[(NG * s + lane / LP) * SX + lane % LP];
      __builtin_amdgcn_sched_barrier(0);
      static_for<16 / NG>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) accA[0][cg] = w4_mfma<CB, s>(wa[cg], xc[s], accA[0][cg]);
      });
    } else
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float xc[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) xc[s] = xcol[s * SX + 64 * h];
      __builtin_amdgcn_sched_barrier(0);
      static_for<16>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) accA[h][cg] = w4_mfma<4, s>(wa[cg], xc[s], accA[h][cg]);
      });
    }
  };

  // ---- one subtile: W update (_nmf.py:540-554, 615-631) and the sums of W^T X / W^T W (:638-640) ------------------
  auto update_subtile = [&](Tile& t, int i, int inext, bool upd) __attribute__((always_inline)) {
    stage_x(t);
    float wold[KQ];
    if (i < ncached) {
      wide_lds_read<float, KQ>(wc_lane + i * 16 * KP, wold);
    } else {
#pragma unroll
      for (int e = 0; e < KQ; ++e) wold[e] = t.w[e];
    }
    if (inext >= 0) issue(t, inext);
    if constexpr (LOSS == 1) {
      w4f4 num[KQ];
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) num[cg] = zero;
      const float* xrow = xs + r * SX + 4 * p;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const w4f4 wh = wh_block4(wold, cb);
        float xb[4];
        wide_lds_read<float, 4>(xrow + 16 * cb, xb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float q = kl_quot4(xb[e], wh[e]);
#pragma unroll
          for (int cg = 0; cg < KQ; ++cg) num[cg] = w4_mfma<0, 0>(hA[cg][cb][e], q, num[cg]);
        }
      }
      float nn[KQ], dd[KQ], qq[KQ], wn[KQ];
      {
        float nf[KP];
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg)
#pragma unroll
          for (int q = 0; q < 4; ++q) nf[4 * cg + q] = num[cg][q];
#pragma unroll
        for (int e = 0; e < KQ; ++e) {
          nn[e] = ((nf[e] + w4_from_part_below<1>(nf[KQ + e])) + w4_from_part_below<2>(nf[2 * KQ + e])) + w4_from_part_below<3>(nf[3 * KQ + e]);
          float d = hsum[e];
          if (a.l1w > 0.0f) d = d + a.l1w;
          if (a.l2w > 0.0f) d = d + a.l2w * wold[e];
          dd[e] = (d == 0.0f) ? eps_val<float>() : d;
        }
      }
      quotients<KQ>(nn, dd, qq);
#pragma unroll
      for (int e = 0; e < KQ; ++e) wn[e] = wold[e] * qq[e];
      if (i < ncached) {
        wide_lds_write<float, KQ>(wc_lane + i * 16 * KP, wn);
      } else {
        w4_store<KQ>(w_rsrc(i), wvoff, wn);
      }
      if (upd) {
        wide_lds_write<float, KQ>(wst + r * SW + KQ * p, wn);
#pragma unroll
        for (int e = 0; e < KQ; ++e) csum[e] += wn[e];
        // Q' = X / max(W' H, eps) with the updated rows, written over X in the stage (each lane replaces exactly what it read)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const w4f4 wh = wh_block4(wn, cb);
          float xb[4], qv[4];
          wide_lds_read<float, 4>(xrow + 16 * cb, xb);
#pragma unroll
          for (int e = 0; e < 4; ++e) qv[e] = kl_quot4(xb[e], wh[e]);
          wide_lds_write<float, 4>(xs + r * SX + 4 * p + 16 * cb, qv);
        }
        wide_wave_lds_fence();
        accumulate_wtx();  // W'^T Q'
      }
      wide_wave_lds_fence();
      return;
    }
    // numerator: partial sums over this lane's quarter of the channels, two chains per component quad
    w4f4 num[KQ], num2[KQ], den[KQ];
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) num[cg] = num2[cg] = den[cg] = zero;
    const float* xrow = xs + r * SX + 4 * p;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      float xb[4];
      wide_lds_read<float, 4>(xrow + 16 * cb, xb);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) {
          if (cb & 1)
            num2[cg] = w4_mfma<0, 0>(hA[cg][cb][e], xb[e], num2[cg]);
          else
            num[cg] = w4_mfma<0, 0>(hA[cg][cb][e], xb[e], num[cg]);
        }
    }
    // denominator: (H H^T) W^T, the contraction split over the parts
#pragma unroll
    for (int e = 0; e < KQ; ++e)
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) den[cg] = w4_mfma<0, 0>(hhA[cg][e], wold[e], den[cg]);
    // reduce-scatter over the four parts: register d KQ + e of the lane d parts below is this lane's component KQ p + e
    float nn[KQ], dd[KQ], qq[KQ], wn[KQ];
    {
      float nf[KP], df[KP];
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          nf[4 * cg + q] = num[cg][q] + num2[cg][q];
          df[4 * cg + q] = den[cg][q];
        }
#pragma unroll
      for (int e = 0; e < KQ; ++e) {
        nn[e] = ((nf[e] + w4_from_part_below<1>(nf[KQ + e])) + w4_from_part_below<2>(nf[2 * KQ + e])) + w4_from_part_below<3>(nf[3 * KQ + e]);
        float d = ((df[e] + w4_from_part_below<1>(df[KQ + e])) + w4_from_part_below<2>(df[2 * KQ + e])) + w4_from_part_below<3>(df[3 * KQ + e]);
        if (a.l1w > 0.0f) d = d + a.l1w;
        if (a.l2w > 0.0f) d = d + a.l2w * wold[e];
        dd[e] = (d == 0.0f) ? eps_val<float>() : d;
      }
    }
    quotients<KQ>(nn, dd, qq);
#pragma unroll
    for (int e = 0; e < KQ; ++e) wn[e] = wold[e] * qq[e];
    if (i < ncached) {
      wide_lds_write<float, KQ>(wc_lane + i * 16 * KP, wn);
    } else {
      w4_store<KQ>(w_rsrc(i), wvoff, wn);
    }
    if (upd) {
      wide_lds_write<float, KQ>(wst + r * SW + KQ * p, wn);
      wide_wave_lds_fence();
      accumulate_wtx();
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg)
#pragma unroll
        for (int cg2 = 0; cg2 < KQ; ++cg2) accB[cg][cg2] = w4_mfma<0, 0>(wa[cg], wa[cg2], accB[cg][cg2]);
    }
    wide_wave_lds_fence();  // the next subtile's stage writes stay behind this one's reads
  };

  // ---- ||X - W H||_F^2 per column and sum X^2 per column of the whole matrix -> sPart[0 .. 2 MP); barriers inside ----
  auto block_resid = [&]() __attribute__((always_inline)) {
    float sse[NH], xsq[NH];
    float kl = 0.0f;  // LOSS == 1: generalised KL divergence, element by element as x log(x / wh) - x + wh (_nmf.py:138-161)
    float hB[NH][KP];  // H[c][64 h + lane]  (LP < 64: H[c][lane mod LP], the same in every group of LP lanes)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      sse[h] = xsq[h] = 0.0f;
#pragma unroll
      for (int c = 0; c < KP; ++c) hB[h][c] = sH[c * SH + 64 * h + lane % LP];
    }
    // A operand: the lane's row of the subtile.  Instruction q covers the row quads NG q + G (G = lane / LP), broadcast from
    // block q of each group: lane G LP + 4 q + i carries row 4 (NG q + G) + i  (LP = 64: lane l < 16 carries row l)
    const int arow = (4 * (NG * ((lane % LP) >> 2) + lane / LP) + (lane & 3)) & 15;
    const unsigned arow_voff = (unsigned)((arow * KP) * 4);
    for (int i = wave; i < ntiles; i += NW) {
      float xg[NLD][4];
      float wrow[KP];  // W[row arow][..]
      {
        const rsrc_t xr = x_rsrc(i);
        const rsrc_t wr = w_rsrc(i);
#pragma unroll
        for (int n = 0; n < NLD; ++n) buf_load<float, 4, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[n], 0u, xg[n]);
        if (i < ncached) {
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            float t4[4];
            wide_lds_read<float, 4>(wcache + i * 16 * KP + arow * KP + 4 * q, t4);
#pragma unroll
            for (int e = 0; e < 4; ++e) wrow[4 * q + e] = t4[e];
          }
        } else {
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            float t4[4];
            buf_load<float, 4>(wr, arow_voff + 16u * q, 0u, t4);
#pragma unroll
            for (int e = 0; e < 4; ++e) wrow[4 * q + e] = t4[e];
          }
        }
      }
      if (xl_active) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) wide_lds_write<float, 4>(xs_put + n * RPL * SX, xg[n]);
      }
      wide_wave_lds_fence();
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        static_for<4 / NG>([&](auto Q_) {
          constexpr int q = decltype(Q_)::value;  // row quad NG q + G: rows 4 (NG q + G) .. + 3
          w4f4 rec = zero;
#pragma unroll
          for (int c = 0; c < KP; ++c) rec = w4_mfma<CB, q>(wrow[c], hB[h][c], rec);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xv = xs[(4 * (NG * q + lane / LP) + e) * SX + 64 * h + lane % LP];
            const float d = xv - rec[e];
            sse[h] = fma_(d, d, sse[h]);
            xsq[h] = fma_(xv, xv, xsq[h]);
            if constexpr (LOSS == 1) {  // branch-free, as in nmf_wide.hpp
              const float whv = rec[e];
              const float whc = whv < eps_val<float>() ? eps_val<float>() : whv;
              const float xs_ = xv > eps_val<float>() ? xv : eps_val<float>();
              const float lg = fma_(xv, log_(xs_ / whc), whv - xv);
              const float term = (xv > eps_val<float>()) ? lg : whv;
              // MP = 48 / 96: the lanes past the padded width hold no channel (their sse / xsq are dropped below)
              kl += (MP % LP == 0 || 64 * h + lane < MP) ? term : 0.0f;
            }
          }
        });
      }
      wide_wave_lds_fence();
    }
    if constexpr (LP < 64) {  // the groups of LP lanes hold the sums over their row quads for the same channels
      if constexpr (LP == 16) {
        sse[0] += __shfl_xor(sse[0], 16, WAVE);
        xsq[0] += __shfl_xor(xsq[0], 16, WAVE);
      }
      sse[0] += __shfl_xor(sse[0], 32, WAVE);
      xsq[0] += __shfl_xor(xsq[0], 32, WAVE);
    }
    float* rec = xs;  // [2][MP] (+ 1) record of this wave (the stage is idle now)
    if constexpr (LOSS == 1) {
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) kl += __shfl_xor(kl, off, WAVE);
      if (lane == 0) rec[2 * MP] = kl;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (64 * h + lane < MP) {
        rec[64 * h + lane] = sse[h];
        rec[MP + 64 * h + lane] = xsq[h];
      }
    __syncthreads();
    for (int idx = tid; idx < 2 * MP + (LOSS == 1 ? 1 : 0); idx += NT) {
      float s = wv0[idx];
      for (int w2 = 1; w2 < NW; ++w2) s += wv0[w2 * C::PERWAVE + idx];
      sPart[idx] = s;
    }
    __syncthreads();
  };
  auto error_from_part = [&]() __attribute__((always_inline)) -> float {
    if constexpr (LOSS == 1) {  // sqrt(2 KL(X || WH)) (_nmf.py:185-189)
      const float d = sPart[2 * MP];
      return sqrt_(2.0f * (d > 0.0f ? d : 0.0f));
    }
    float tot = 0.0f;
    for (int jj = 0; jj < m; ++jj) tot += sPart[jj];
    return sqrt_(tot);
  };

  // per-wave record [W^T X | W^T W] of a pass over the wave's stages (idle between passes); the waves' records are summed in fixed order
  auto write_record = [&]() __attribute__((always_inline)) {
    float* rec = xs;
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
      if constexpr (LP < 64) {  // lanes l, l + LP, ... hold the sums over the rows of their group for the same channel
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = accA[0][cg][q];
          if constexpr (LP == 16) v += __shfl_xor(v, 16, WAVE);
          v += __shfl_xor(v, 32, WAVE);
          if (lane < MP) rec[(4 * cg + q) * MP + lane] = v;
        }
      } else {
#pragma unroll
        for (int h = 0; h < NH; ++h)
          if (64 * h + lane < MP) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rec[(4 * cg + q) * MP + 64 * h + lane] = accA[h][cg][q];
          }
      }
      if constexpr (LOSS == 1) continue;  // (colsum(W') below instead of W^T W)
#pragma unroll
      for (int cg2 = 0; cg2 < KQ; ++cg2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // sum over the 16 blocks (= rows of the subtiles): two row rotations, two cross-row exchanges
          float v = accB[cg][cg2][q];
          v += dpp_mov<0x124>(v);
          v += dpp_mov<0x128>(v);
          v += __shfl_xor(v, 16, WAVE);
          v += __shfl_xor(v, 32, WAVE);
          if (lane < 4) rec[KP * MP + (4 * cg + q) * KP + 4 * cg2 + lane] = v;
        }
    }
    if constexpr (LOSS == 1) {  // colsum(W'): lanes (r, p) of one p hold the 16 rows: the quad (xor 1, 2), then the row quads (xor 16, 32)
#pragma unroll
      for (int e = 0; e < KQ; ++e) {
        float v = csum[e];
        v += __shfl_xor(v, 1, WAVE);
        v += __shfl_xor(v, 2, WAVE);
        v += __shfl_xor(v, 16, WAVE);
        v += __shfl_xor(v, 32, WAVE);
        if (ii == 0 && rq == 0) rec[KP * MP + KQ * p + e] = v;
      }
    }
  };

  if (a.mode == 2) {  // residual of the slice: per-column sums to global memory, summed over the slices by wide_resid_finalize_kernel
    block_resid();
    float* out = a.colpart + ((long long)b * a.S + slice) * (2 * MP);
    for (int idx = tid; idx < 2 * MP; idx += NT) out[idx] = sPart[idx];
    return;
  }
  if (a.mode == 1) {  // one update pass over the slice, its record to global memory (wide_hupdate_kernel sums the slices)
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
#pragma unroll
      for (int h = 0; h < NH; ++h) accA[h][cg] = zero;
#pragma unroll
      for (int cg2 = 0; cg2 < KQ; ++cg2) accB[cg][cg2] = zero;
    }
    const bool upd1 = a.update_h != 0;
    Tile t1;
    issue(t1, wave);
    for (int i = wave; i < ntiles; i += NW) {
      update_subtile(t1, i, i + NW, upd1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!upd1) return;
    write_record();
    __syncthreads();
    float* out = a.part + ((long long)b * a.S + slice) * C::REC;
    for (int idx = tid; idx < C::REC; idx += NT) {
      float sacc = wv0[idx];
      for (int w2 = 1; w2 < NW; ++w2) sacc += wv0[w2 * C::PERWAVE + idx];
      out[idx] = sacc;
    }
    return;
  }

  float err0 = 0.0f, prev = 0.0f;
  if (a.tol > 0.0f) {
    block_resid();
    err0 = error_from_part();
    prev = err0;
  }
  const bool upd = a.update_h != 0;
  int n_iter = 0;
  Tile ta, tb;
  __builtin_amdgcn_sched_barrier(0);
  issue(ta, wave);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NSET > 1) issue(tb, wave + NW);
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 1; it <= a.max_iter; ++it) {
    n_iter = it;
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
#pragma unroll
      for (int h = 0; h < NH; ++h) accA[h][cg] = zero;
#pragma unroll
      for (int cg2 = 0; cg2 < KQ; ++cg2) accB[cg][cg2] = zero;
      csum[cg] = 0.0f;
    }
    if constexpr (NSET > 1) {
      // pairs of subtiles in a loop without inner exits, then the odd one: with a conditional second half (or with the two
      // requests of a prologue swapped by the scheduler, hence the sched_barriers around them) the compiler's s_waitcnt
      // vmcnt must assume the worst path and waits for ALL loads in flight at the top of every round
      int i = wave;
      for (; i + NW < ntiles; i += 2 * NW) {
        update_subtile(ta, i, i + 2 * NW, upd);
        __builtin_amdgcn_sched_barrier(0);
        update_subtile(tb, i + NW, i + 3 * NW, upd);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i < ntiles) update_subtile(ta, i, -1, upd);  // (requests nothing: the next pass's first subtiles are requested below, in order)
      __builtin_amdgcn_sched_barrier(0);
    } else {
      for (int i = wave; i < ntiles; i += NW) {
        update_subtile(ta, i, i + NW, upd);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // X does not depend on H, and this wave's first rows of W are final: request the next pass's first subtiles now
    if (it < a.max_iter) {
      __builtin_amdgcn_sched_barrier(0);
      issue(ta, wave);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NSET > 1) issue(tb, wave + NW);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (upd) {
      write_record();
      __syncthreads();
      for (int idx = tid; idx < (LOSS == 1 ? KP * MP + KP : C::REC); idx += NT) {
        float s = wv0[idx];
        for (int w2 = 1; w2 < NW; ++w2) s += wv0[w2 * C::PERWAVE + idx];
        sA[idx] = s;  // sB follows sA
      }
      __syncthreads();
      // H *= (W^T X) / ((W^T W) H)   (_nmf.py:638-640, 701-728)
      constexpr int NHU = (KP * MP + NT - 1) / NT;
      float nh[NHU];
#pragma unroll
      for (int q = 0; q < NHU; ++q) {
        const int idx = tid + q * NT;
        const int c = idx / MP, jj = idx % MP;
        nh[q] = 0.0f;
        if (idx < KP * MP && c < k && jj < m) {
          float d;
          if constexpr (LOSS == 1) {  // H *= (W'^T Q') / colsum(W')   (_nmf.py:663-684; colsum 0 -> 1)
            d = sB[c];
            if (d == 0.0f) d = 1.0f;
          } else {
            d = sB[c * KP] * sH[jj];
            for (int c2 = 1; c2 < k; ++c2) d = fma_(sB[c * KP + c2], sH[c2 * SH + jj], d);
          }
          const float hold = sH[c * SH + jj];
          if (a.l1h > 0.0f) d = d + a.l1h;
          if (a.l2h > 0.0f) d = d + a.l2h * hold;
          d = (d == 0.0f) ? eps_val<float>() : d;
          nh[q] = hold * (sA[idx] / d);
          if constexpr (LOSS == 1) {
            if (nh[q] < 2.220446049250313e-16f) nh[q] = 0.0f;  // H[H < float64 eps] = 0 (_nmf.py:866-868)
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NHU; ++q) {
        const int idx = tid + q * NT;
        if (idx < KP * MP) sH[(idx / MP) * SH + idx % MP] = nh[q];
      }
      __syncthreads();
      compute_hht_lds();
      __syncthreads();
      load_operands();
    }
    if (a.tol > 0.0f && (it % a.check_every) == 0) {
      block_resid();
      const float err = error_from_part();
      if ((prev - err) / err0 < a.tol) break;
      prev = err;  // (the residual pass used the stages but not ta / tb: the requests above are still good)
    }
  }
  // reconstruction_err_ (_nmf.py:1628-1630) + per-column SSE / sum X^2 for VAF (analysis.py:654-662)
  block_resid();
  if (tid == 0) {
    if (a.err_out) a.err_out[b] = error_from_part();
    if (a.n_iter_out) a.n_iter_out[b] = n_iter;
  }
  for (int jj = tid; jj < m; jj += NT) {
    if (a.sse_col_out) a.sse_col_out[(long long)b * m + jj] = sPart[jj];
    if (a.xsq_col_out) a.xsq_col_out[(long long)b * m + jj] = sPart[MP + jj];
  }
  if (upd) {
    for (int idx = tid; idx < k * m; idx += NT) Hb[idx] = sH[(idx / m) * SH + idx % m];
  }
  // the cached rows of W back to global memory (block_resid above ended with a barrier: every wave's rows are final)
  for (int idx = tid; idx < ncached * 16 * KP && idx < T * KP; idx += NT) Wb[idx] = wcache[idx];
}

}  // namespace hipnmf
