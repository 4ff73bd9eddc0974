// nmf_wide4d.hpp -- float64 counterpart of fit_wide4_kernel (nmf_wide4.hpp): 1..128 channels (MP = 16 .. 128) with at most 8
// components on v_mfma_f64_4x4x4_4b_f64 instead of v_mfma_f64_16x16x4_f64.  float64 is what the reference's own calls carry
// (a DataFrame is float64 and scikit-learn keeps the dtype: src/muscle_synergies/analysis.py:862-863, sklearn/decomposition/
// _nmf.py:1638-1734), and the 16x16x4 tile pads the components to 16: at k = 8 half of every fp64 MFMA -- 64 pipe cycles
// each -- multiplies zeros (40 MFMAs = 2 560 cycles per 16-row subtile).  The 4x4x4 form (4 independent D(4x4) += A(4x4) B(4x4),
// 16 cycles) pads to a multiple of four: 72 MFMAs = 1 152 cycles at 64 channels, k = 8.
//
// Arithmetic replaced: _multiplicative_update_w (_nmf.py:526-631), _multiplicative_update_h (:634-728), _beta_divergence
// (:85-134), loop + stop rule (:731-893).  sklearn notation: X (T x m) ~ W (T x k) H (k x m).
//
// v_mfma_f64_4x4x4_4b_f64, probed with tools/ubench/mfma_f64_4x4x4_layout.hip: with lane = q0 + 4 q1 + 16 q2,
//   A_b[i][k]: i = q0, b = q1, k = q2;   B_b[k][j]: j = q0, b = q1, k = q2;   D_b[i][j]: j = q0, b = q1, i = q2  (one register each)
// -- the contracted index of the operands and the row index of the result share the high lane bits, so a result register
// is directly the B operand of a product that contracts over what it enumerates.  A wave owns 16-row subtiles:
//   numerator^T = H X^T   blocks = row quads: B lane (j, b, k) = X[row 4 b + j][channels of part k] (the lane's quarter of the
//                         row: 16-byte pieces 4 u + k), A lane (i, b, k) = H[4 cg + i][same channel] from per-lane registers;
//                         the sum over the four parts happens INSIDE the instruction (k is the contracted index): D lane
//                         (j, b, i) = numerator of component 4 cg + i for row 4 b + j -- no cross-lane reduction at all
//   denominator^T         A lane (i, b, k) = HHt[4 cg + i][4 cg' + k], B = the lane's own W value of quad cg' (its component index
//                         sits where B wants k)
//   W <- W * num / den    one value per lane and quad; loads / stores of 8 bytes per lane, 32 contiguous bytes per row and quad
//   W^T X                 k = rows: A lane (i, b, k) = W'[row 4 t + k][4 cg + i] (all blocks alike), B lane (j, b, k) = X[row 4 t + k]
//                         [16 q + 4 b + j]; D lane (j, b, i): component 4 cg + i, channel 16 q + 4 b + j -- final layout
//   W^T W                 blocks = row quads: A = B = W'[row 4 b + k][4 cg + i]; summed over the blocks once per pass
//   residual R = W H      k = components: A lane (i, b, k) = W[row 4 t + i][4 cg + k], B lane (j, b, k) = H[4 cg + k][16 q + 4 b + j]
// X staging, W cache, descriptors, pass structure and epilogue are fit_wide_kernel's (nmf_wide.hpp).
#pragma once
#include "nmf_wide.hpp"

namespace hipnmf {

template <int MP, int KQ>
struct Wide4dCfg {
  static constexpr int KP = 4 * KQ;
  static constexpr int NQ = MP / 16;  // 16-channel groups
  static constexpr int NU = MP / 8;   // 16-byte pieces per lane and row in the numerator (a quarter of the row)
  static constexpr int CPR = MP / 2;  // 16-byte pieces per row
  static constexpr int RPL = wide_pow2_floor(64 / CPR) > 16 ? 16 : wide_pow2_floor(64 / CPR);
  static constexpr int NLD = 16 / RPL;
  // row stride of the X stage (doubles) = 2 (mod 32): the numerator's 16-byte reads of 16 rows at one piece offset and the
  // 8-byte reads of W^T X then hit distinct banks
  static constexpr int SX = ((MP + 29) / 32) * 32 + 2;
  static constexpr int SW = KP + 2;  // row stride of the W stage
  static constexpr int SH = MP + 2;  // row stride of H in LDS
  static constexpr int XS = 16 * SX, WS = 16 * SW;
  static constexpr int REC = KP * MP + KP * KP;
  static constexpr int RREC = 8 * MP;  // residual record: 4 row slots x (sse | xsq) x MP
  static constexpr int PERWAVE = (XS + WS > REC ? (XS + WS > RREC ? XS + WS : RREC) : (REC > RREC ? REC : RREC));
  static constexpr int COMMON = KP * SH + KP * KP + KP * MP + KP * KP + 2 * MP + 8;
  static_assert(MP % 16 == 0 && MP >= 16 && MP <= 128 && (KQ == 1 || KQ == 2), "unsupported wide4d shape");
  static_assert(CPR <= 64, "a row must fit one load instruction");
  __host__ __device__ static constexpr size_t smem_bytes(int nw) { return sizeof(double) * (size_t)(COMMON + nw * PERWAVE); }
};

__device__ __forceinline__ double w4d_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int MP, int KQ>
struct Wide4dTile {
  using C = Wide4dCfg<MP, KQ>;
  double xg[C::NLD][2];  // the subtile of X as loaded: piece (lane % CPR) of row n RPL + lane / CPR
  double w[KQ];          // W[row 4 b + j][4 cg + i]
};

// WPE: waves per SIMD the instance is compiled for (2 = at most 256 registers: up to 64 channels; 1 beyond)
// LOSS: 0 = Frobenius, 1 = Kullback-Leibler (beta_loss = 1; _nmf.py:556-591, 642-684; round 5), as in fit_wide4_kernel: both
// reconstructions on the pipe in the numerator's own layout.  D lane (j, b, i) = (W H)[row 4 b + j][8 u + 2 i + e] needs
// A lane (i, b, k) = H[4 cg + k][8 u + 2 i + e] (a 16-byte read of H per quad and piece) and B lane (j, b, k) = W[row 4 b + j][4 cg + k]
// -- the lane's own W value of the quad; the lane's X values sit exactly there, so Q = X / max(W H, eps) is the numerator's B
// operand, Q' (with the updated rows) goes over X in the stage and W'^T Q' is the Frobenius W^T X.
template <int MP, int KQ, int NW, int NSET, int WPE = 2, int LOSS = 0>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE > 1 ? 8 : 1))) fit_wide4d_kernel(WideArgs<double> a) {
  using C = Wide4dCfg<MP, KQ>;
  using Tile = Wide4dTile<MP, KQ>;
  constexpr int KP = C::KP, NQ = C::NQ, NU = C::NU, SX = C::SX, SW = C::SW, SH = C::SH, NLD = C::NLD, RPL = C::RPL, CPR = C::CPR,
                NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* const sH = reinterpret_cast<double*>(smem_raw);  // [KP][SH]
  double* const sHHt = sH + KP * SH;                       // [KP][KP]
  double* const sA = sHHt + KP * KP;                       // [KP][MP]   W^T X summed over the waves
  double* const sB = sA + KP * MP;                         // [KP][KP]   W^T W   (directly behind sA: one index space)
  double* const sPart = sB + KP * KP;                      // [2 MP + 8] per-column sse | xsq of the residual pass
  double* const wv0 = sPart + 2 * MP + 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int q0 = lane & 3, q1 = (lane >> 2) & 3, q2 = lane >> 4;
  const int r = 4 * q1 + q0;  // the lane's row in the row-per-(j, b) layouts (numerator B operand, numerator / denominator result)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* const xs = wv0 + wave * C::PERWAVE;  // [16][SX] this wave's X stage
  double* const wst = xs + C::XS;              // [16][SW] this wave's W stage

  const int b = blockIdx.x;
  const double* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  double* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  double* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
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
    if (a.state && a.state[(long long)b * 8 + 3] != 0.0) return;
    const int row_begin = slice * a.rows_per_slice;
    int rows = T - row_begin;
    if (rows > a.rows_per_slice) rows = a.rows_per_slice;
    if (rows <= 0) rows = 0;
    Xb += (long long)row_begin * a.ldx;
    Wb += (long long)row_begin * KP;
    T = rows;
  }
  const int ntiles = (T + 15) / 16;
  double* const wcache = wv0 + NW * C::PERWAVE;  // [lds_rows][KP]
  const int ncached = (a.lds_rows / 16 < ntiles) ? a.lds_rows / 16 : ntiles;
  const unsigned ldx_b = (unsigned)(a.ldx * 8LL);
  constexpr unsigned ldw_b = (unsigned)KP * 8u;

  const int xl_row = lane / CPR, xl_chunk = lane % CPR;
  const bool xl_active = lane < RPL * CPR;
  unsigned xvoff[NLD];
#pragma unroll
  for (int n = 0; n < NLD; ++n)
    xvoff[n] = (xl_active && xl_chunk < a.xchunks) ? (unsigned)(n * RPL + xl_row) * ldx_b + (unsigned)xl_chunk * 16u : OOB;
  double* const xs_put = xs + xl_row * SX + xl_chunk * 2;
  // this lane's W values: row r, component 4 cg + q2
  unsigned wvoff[KQ];
#pragma unroll
  for (int cg = 0; cg < KQ; ++cg) wvoff[cg] = (unsigned)((r * KP + 4 * cg + q2) * 8);
  double* const wc_lane = wcache + r * KP + q2;
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
    for (int n = 0; n < NLD; ++n) buf_load<double, 2, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[n], 0u, t.xg[n]);
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
      double tmp[1];
      buf_load<double, 1>(wr, wvoff[cg], 0u, tmp);
      t.w[cg] = tmp[0];
    }
  };
  auto stage_x = [&](const Tile& t) __attribute__((always_inline)) {
    if (xl_active) {
#pragma unroll
      for (int n = 0; n < NLD; ++n) wide_lds_write<double, 2>(xs_put + n * RPL * SX, t.xg[n]);
    }
    wide_wave_lds_fence();
  };

  for (int idx = tid; idx < ncached * 16 * KP; idx += NT) wcache[idx] = (idx < T * KP) ? Wb[idx] : 0.0;
  for (int idx = tid; idx < KP * SH; idx += NT) {
    const int c = idx / SH, jj = idx % SH;
    sH[idx] = (c < k && jj < m) ? Hb[c * m + jj] : 0.0;
  }
  __syncthreads();
  auto compute_hht_lds = [&]() __attribute__((always_inline)) {  // call between barriers
    if constexpr (LOSS == 1) {  // rowsum(H), the W update's denominator (_nmf.py:577-581), in sHHt[0 .. KP)
      for (int c = tid; c < KP; c += NT) {
        double s = 0.0;
        for (int jj = 0; jj < MP; ++jj) s += sH[c * SH + jj];
        sHHt[c] = s;
      }
      return;
    }
    for (int idx = tid; idx < KP * KP; idx += NT) {
      const int c = idx / KP, c2 = idx % KP;
      double s = 0.0;
      for (int jj = 0; jj < MP; ++jj) s = fma_(sH[c * SH + jj], sH[c2 * SH + jj], s);
      sHHt[idx] = s;
    }
  };
  compute_hht_lds();
  __syncthreads();

  // A operands that change once per iteration: lane (i = q0, b, k = q2)
  double hA[KQ][NU][2];  // H[4 cg + q0][8 u + 2 q2 + e]
  double hhA[KQ][KQ];    // HHt[4 cg + q0][4 cg' + q2]
  double hsum[KQ];       // KL: rowsum(H) of the lane's components 4 cg + q2
  auto load_operands = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
#pragma unroll
      for (int u = 0; u < NU; ++u) wide_lds_read<double, 2>(sH + (4 * cg + q0) * SH + 8 * u + 2 * q2, hA[cg][u]);
      if constexpr (LOSS == 1) {
        hsum[cg] = sHHt[4 * cg + q2];
      } else {
#pragma unroll
        for (int cg2 = 0; cg2 < KQ; ++cg2) hhA[cg][cg2] = sHHt[(4 * cg + q0) * KP + 4 * cg2 + q2];
      }
    }
  };
  load_operands();

  double accA[NQ][KQ], accB[KQ][KQ];
  double csum[KQ];  // KL: this lane's share of colsum(W') (its row, its components)
  // KL: (W H)[row r][8 u + 2 q2 + e], e = 0, 1, from the lane's W values of the row (old or new)
  const double* const ht_lane = sH + q2 * SH + 2 * q0;  // H[4 cg + q2][8 u + 2 q0 + e] at + 4 cg SH + 8 u
  auto wh_piece = [&](const double (&w)[KQ], int u, double (&wh)[2]) __attribute__((always_inline)) {
    wh[0] = wh[1] = 0.0;
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
      double ht[2];
      wide_lds_read<double, 2>(ht_lane + 4 * cg * SH + 8 * u, ht);
#pragma unroll
      for (int e = 0; e < 2; ++e) wh[e] = w4d_mfma(ht[e], w[cg], wh[e]);
    }
  };
  // W'^T X (KL: W'^T Q') of the subtile in the stage with the new rows in `wst`: every LDS operand of the phase is requested
  // before its first product (see nmf_wide.hpp, HOIST)
  auto accumulate_wtx = [&]() __attribute__((always_inline)) {
    double wa[4][KQ];  // lane (i, b, k): W'[row 4 t + k][4 cg + i]
    double wb[KQ];     // lane (i, b, k): W'[row 4 b + k][4 cg + i]
    double xc[4][NQ];  // lane (j, b, k): X[row 4 t + k][16 q + 4 b + j]
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) wa[t4][cg] = wst[(4 * t4 + q2) * SW + 4 * cg + q0];
#pragma unroll
      for (int q = 0; q < NQ; ++q) xc[t4][q] = xs[(4 * t4 + q2) * SX + 16 * q + 4 * q1 + q0];
    }
    if constexpr (LOSS == 0) {
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) wb[cg] = wst[(4 * q1 + q2) * SW + 4 * cg + q0];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) accA[q][cg] = w4d_mfma(wa[t4][cg], xc[t4][q], accA[q][cg]);
    if constexpr (LOSS == 0) {
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg)
#pragma unroll
        for (int cg2 = 0; cg2 < KQ; ++cg2) accB[cg][cg2] = w4d_mfma(wb[cg], wb[cg2], accB[cg][cg2]);
    }
  };

  auto update_subtile = [&](Tile& t, int i, int inext, bool upd) __attribute__((always_inline)) {
    stage_x(t);
    double wold[KQ];
    if (i < ncached) {
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) wold[cg] = wc_lane[i * 16 * KP + 4 * cg];
    } else {
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) wold[cg] = t.w[cg];
    }
    if (inext >= 0) issue(t, inext);
    if constexpr (LOSS == 1) {
      const double* xrow = xs + r * SX + 2 * q2;
      double num[KQ];
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) num[cg] = 0.0;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double xb[2], wh[2];
        wide_lds_read<double, 2>(xrow + 8 * u, xb);
        wh_piece(wold, u, wh);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const double qv = xb[e] / kl_floor(wh[e]);  // X / max(WH, EPSILON) (_nmf.py:574-575)
#pragma unroll
          for (int cg = 0; cg < KQ; ++cg) num[cg] = w4d_mfma(hA[cg][u][e], qv, num[cg]);
        }
      }
      double wn[KQ];
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) {
        double d = hsum[cg];
        if (a.l1w > 0.0) d = d + a.l1w;
        if (a.l2w > 0.0) d = d + a.l2w * wold[cg];
        d = (d == 0.0) ? eps_val<double>() : d;
        wn[cg] = wold[cg] * (num[cg] / d);
      }
      if (i < ncached) {
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) wc_lane[i * 16 * KP + 4 * cg] = wn[cg];
      } else {
        const rsrc_t wr = w_rsrc(i);
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) buf_store<double>(wr, wvoff[cg], 0u, wn[cg]);
      }
      if (upd) {
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) {
          wst[r * SW + 4 * cg + q2] = wn[cg];
          csum[cg] += wn[cg];
        }
        // Q' = X / max(W' H, eps) with the updated rows, written over X in the stage (each lane replaces exactly what it read)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          double xb[2], wh[2], qv[2];
          wide_lds_read<double, 2>(xrow + 8 * u, xb);
          wh_piece(wn, u, wh);
#pragma unroll
          for (int e = 0; e < 2; ++e) qv[e] = xb[e] / kl_floor(wh[e]);
          wide_lds_write<double, 2>(xs + r * SX + 2 * q2 + 8 * u, qv);
        }
        wide_wave_lds_fence();
        accumulate_wtx();  // W'^T Q'
      }
      wide_wave_lds_fence();
      return;
    }
    // numerator (two chains per quad) and denominator
    double num[KQ], num2[KQ], den[KQ];
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) num[cg] = num2[cg] = den[cg] = 0.0;
    const double* xrow = xs + r * SX + 2 * q2;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      double xb[2];
      wide_lds_read<double, 2>(xrow + 8 * u, xb);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) {
          if (u & 1)
            num2[cg] = w4d_mfma(hA[cg][u][e], xb[e], num2[cg]);
          else
            num[cg] = w4d_mfma(hA[cg][u][e], xb[e], num[cg]);
        }
    }
#pragma unroll
    for (int cg2 = 0; cg2 < KQ; ++cg2)
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) den[cg] = w4d_mfma(hhA[cg][cg2], wold[cg2], den[cg]);
    double wn[KQ];
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
      double d = den[cg];
      if (a.l1w > 0.0) d = d + a.l1w;
      if (a.l2w > 0.0) d = d + a.l2w * wold[cg];
      d = (d == 0.0) ? eps_val<double>() : d;
      wn[cg] = wold[cg] * ((num[cg] + num2[cg]) / d);
    }
    if (i < ncached) {
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) wc_lane[i * 16 * KP + 4 * cg] = wn[cg];
    } else {
      const rsrc_t wr = w_rsrc(i);
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) buf_store<double>(wr, wvoff[cg], 0u, wn[cg]);
    }
    if (upd) {
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) wst[r * SW + 4 * cg + q2] = wn[cg];
      wide_wave_lds_fence();
      accumulate_wtx();
    }
    wide_wave_lds_fence();  // the next subtile's stage writes stay behind this one's reads
  };

  // ---- ||X - W H||_F^2 per column and sum X^2 per column of the whole matrix -> sPart[0 .. 2 MP); barriers inside ----
  auto block_resid = [&]() __attribute__((always_inline)) {
    double sse[NQ], xsq[NQ];
    double kl = 0.0;    // LOSS == 1: generalised KL divergence, element by element as x log(x / wh) - x + wh (_nmf.py:138-161)
    double hB[KQ][NQ];  // lane (j, b, k): H[4 cg + k][16 q + 4 b + j]
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      sse[q] = xsq[q] = 0.0;
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) hB[cg][q] = sH[(4 * cg + q2) * SH + 16 * q + 4 * q1 + q0];
    }
    for (int i = wave; i < ntiles; i += NW) {
      Tile t;
      issue(t, i);
      stage_x(t);
#pragma unroll
      for (int cg = 0; cg < KQ; ++cg) wst[r * SW + 4 * cg + q2] = (i < ncached) ? wc_lane[i * 16 * KP + 4 * cg] : t.w[cg];
      wide_wave_lds_fence();
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        double wr_[KQ];  // lane (i, b, k): W[row 4 t + i][4 cg + k]
#pragma unroll
        for (int cg = 0; cg < KQ; ++cg) wr_[cg] = wst[(4 * t4 + q0) * SW + 4 * cg + q2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          double rec = 0.0;
#pragma unroll
          for (int cg = 0; cg < KQ; ++cg) rec = w4d_mfma(wr_[cg], hB[cg][q], rec);
          // D lane (j, b, i): R[row 4 t + q2][16 q + 4 b + j]
          const double xv = xs[(4 * t4 + q2) * SX + 16 * q + 4 * q1 + q0];
          const double d = xv - rec;
          sse[q] = fma_(d, d, sse[q]);
          xsq[q] = fma_(xv, xv, xsq[q]);
          if constexpr (LOSS == 1) {  // branch-free, as in nmf_wide.hpp (padded rows / channels: x = wh = 0 -> 0)
            const double whc = rec < eps_val<double>() ? eps_val<double>() : rec;
            const double xs_ = xv > eps_val<double>() ? xv : eps_val<double>();
            const double lg = fma_(xv, log_(xs_ / whc), rec - xv);
            kl += (xv > eps_val<double>()) ? lg : rec;
          }
        }
      }
      wide_wave_lds_fence();
    }
    // lanes q2 = 0..3 hold partial sums (rows = q2 mod 4) of the same channels: record [q2][sse | xsq][MP], summed below
    double* rec = xs;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      rec[q2 * 2 * MP + 16 * q + 4 * q1 + q0] = sse[q];
      rec[q2 * 2 * MP + MP + 16 * q + 4 * q1 + q0] = xsq[q];
    }
    if constexpr (LOSS == 1) {
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) kl += __shfl_xor(kl, off, WAVE);
      if (lane == 0) rec[8 * MP] = kl;  // (behind the four row slots; PERWAVE >= 16 SX > 8 MP)
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * MP; idx += NT) {
      double s = 0.0;
      for (int w2 = 0; w2 < NW; ++w2)
        for (int qq = 0; qq < 4; ++qq) s += wv0[w2 * C::PERWAVE + qq * 2 * MP + idx];
      sPart[idx] = s;
    }
    if constexpr (LOSS == 1) {
      if (tid == 0) {
        double s = 0.0;
        for (int w2 = 0; w2 < NW; ++w2) s += wv0[w2 * C::PERWAVE + 8 * MP];
        sPart[2 * MP] = s;
      }
    }
    __syncthreads();
  };
  auto error_from_part = [&]() __attribute__((always_inline)) -> double {
    if constexpr (LOSS == 1) {  // sqrt(2 KL(X || WH)) (_nmf.py:185-189)
      const double d = sPart[2 * MP];
      return sqrt_(2.0 * (d > 0.0 ? d : 0.0));
    }
    double tot = 0.0;
    for (int jj = 0; jj < m; ++jj) tot += sPart[jj];
    return sqrt_(tot);
  };

  // per-wave record [W^T X | W^T W] of a pass over the wave's stages (idle between passes); the waves' records are summed in fixed order
  auto write_record = [&]() __attribute__((always_inline)) {
    double* rec = xs;
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) rec[(4 * cg + q2) * MP + 16 * q + 4 * q1 + q0] = accA[q][cg];
      if constexpr (LOSS == 1) {  // colsum(W') instead of W^T W: the lanes of one q2 hold the 16 rows of component 4 cg + q2
        double v = csum[cg];
        v += __shfl_xor(v, 1, WAVE);
        v += __shfl_xor(v, 2, WAVE);
        v += __shfl_xor(v, 4, WAVE);
        v += __shfl_xor(v, 8, WAVE);
        if (q0 == 0 && q1 == 0) rec[KP * MP + 4 * cg + q2] = v;
        continue;
      }
#pragma unroll
      for (int cg2 = 0; cg2 < KQ; ++cg2) {
        // accB lane (j, b, i): partial over the rows of quad b of W^T W[4 cg + i][4 cg2 + j]: sum over b (lane bits 2, 3)
        double v = accB[cg][cg2];
        v += __shfl_xor(v, 4, WAVE);
        v += __shfl_xor(v, 8, WAVE);
        if (q1 == 0) rec[KP * MP + (4 * cg + q2) * KP + 4 * cg2 + q0] = v;
      }
    }
  };

  if (a.mode == 2) {  // residual of the slice: per-column sums to global memory, summed over the slices by wide_resid_finalize_kernel
    block_resid();
    double* out = a.colpart + ((long long)b * a.S + slice) * (2 * MP);
    for (int idx = tid; idx < 2 * MP; idx += NT) out[idx] = sPart[idx];
    return;
  }
  if (a.mode == 1) {  // one update pass over the slice, its record to global memory (wide_hupdate_kernel sums the slices)
#pragma unroll
    for (int cg = 0; cg < KQ; ++cg) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) accA[q][cg] = 0.0;
#pragma unroll
      for (int cg2 = 0; cg2 < KQ; ++cg2) accB[cg][cg2] = 0.0;
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
    double* out = a.part + ((long long)b * a.S + slice) * C::REC;
    for (int idx = tid; idx < C::REC; idx += NT) {
      double sacc = wv0[idx];
      for (int w2 = 1; w2 < NW; ++w2) sacc += wv0[w2 * C::PERWAVE + idx];
      out[idx] = sacc;
    }
    return;
  }

  double err0 = 0.0, prev = 0.0;
  if (a.tol > 0.0) {
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
      for (int q = 0; q < NQ; ++q) accA[q][cg] = 0.0;
#pragma unroll
      for (int cg2 = 0; cg2 < KQ; ++cg2) accB[cg][cg2] = 0.0;
      csum[cg] = 0.0;
    }
    if constexpr (NSET > 1) {
      int i = wave;  // (pairs without inner exits, a tail that requests nothing: see fit_wide_kernel)
      for (; i + NW < ntiles; i += 2 * NW) {
        update_subtile(ta, i, i + 2 * NW, upd);
        __builtin_amdgcn_sched_barrier(0);
        update_subtile(tb, i + NW, i + 3 * NW, upd);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i < ntiles) update_subtile(ta, i, -1, upd);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      for (int i = wave; i < ntiles; i += NW) {
        update_subtile(ta, i, i + NW, upd);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
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
        double s = wv0[idx];
        for (int w2 = 1; w2 < NW; ++w2) s += wv0[w2 * C::PERWAVE + idx];
        sA[idx] = s;  // sB follows sA
      }
      __syncthreads();
      // H *= (W^T X) / ((W^T W) H)   (_nmf.py:638-640, 701-728)
      constexpr int NHU = (KP * MP + NT - 1) / NT;
      double nh[NHU];
#pragma unroll
      for (int q = 0; q < NHU; ++q) {
        const int idx = tid + q * NT;
        const int c = idx / MP, jj = idx % MP;
        nh[q] = 0.0;
        if (idx < KP * MP && c < k && jj < m) {
          double d;
          if constexpr (LOSS == 1) {  // H *= (W'^T Q') / colsum(W')   (_nmf.py:663-684; colsum 0 -> 1)
            d = sB[c];
            if (d == 0.0) d = 1.0;
          } else {
            d = sB[c * KP] * sH[jj];
            for (int c2 = 1; c2 < k; ++c2) d = fma_(sB[c * KP + c2], sH[c2 * SH + jj], d);
          }
          const double hold = sH[c * SH + jj];
          if (a.l1h > 0.0) d = d + a.l1h;
          if (a.l2h > 0.0) d = d + a.l2h * hold;
          d = (d == 0.0) ? eps_val<double>() : d;
          nh[q] = hold * (sA[idx] / d);
          if constexpr (LOSS == 1) {
            if (nh[q] < 2.220446049250313e-16) nh[q] = 0.0;  // H[H < float64 eps] = 0 (_nmf.py:866-868)
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
    if (a.tol > 0.0 && (it % a.check_every) == 0) {
      block_resid();
      const double err = error_from_part();
      if ((prev - err) / err0 < a.tol) break;
      prev = err;
    }
  }
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
  for (int idx = tid; idx < ncached * 16 * KP && idx < T * KP; idx += NT) Wb[idx] = wcache[idx];
}

}  // namespace hipnmf
