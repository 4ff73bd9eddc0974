// nmf_wide.hpp -- the wide-shape instances of the solver (round 3): every (n_features, n_components) the narrow
// lane mappings of nmf_kernels.hpp do not cover -- up to 128 channels (HD-EMG grids) and up to 32 components, fp32
// and fp64 -- with all four contractions of an iteration on the matrix pipe.
//   fit_wide_kernel<real, MP, KP, NW>   one NW-wave workgroup per matrix, every iteration inside the kernel
//
// Arithmetic replaced: sklearn/decomposition/_nmf.py (1.7.2) _multiplicative_update_w (:526-631),
// _multiplicative_update_h (:634-728), _beta_divergence (:85-134), loop + stop rule (:731-893), reached from the
// reference at src/muscle_synergies/analysis.py:862-863 (any 1 <= n <= max <= m passes its validation, :829-846).
// sklearn notation: X (T x m) ~ W (T x k) H (k x m).
//
// Formulation.  v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 compute D (16 x 16) += A (16 x 4) B (4 x 16) with the
// non-contracted index of either operand on the low four lane bits and the contracted one on the high two:
// A: lane (i, g) <-> A[i][g], B: lane (j, g) <-> B[g][j], D: lane (j, g), register r <-> D[4 g + r][j]  (fp64: D[g + 4 r][j];
// the A operands below are loaded with their rows permuted so that register r of lane (j, g) means row 4 g + r for
// both types).  A wave owns 16-row subtiles of the matrix; with l = lane, j = l & 15, g = l >> 4:
//   numerator^T   = H X^T          A = H (components x channels, from LDS), B = X^T: lane (j, g) holds row j of the
//                                  subtile, VEC consecutive channels per LDS read; D: lane (row j, g), reg r <-> comp 4g+r
//   denominator^T = (H H^T) W^T    B = W^T: register r of the row-per-lane W fragment (lane (row j, g) <-> comps 4g..4g+3,
//                                  one 16-byte load from the row-major W) IS the B operand of k-step r; A = H H^T
//   W <- W * num / den             elementwise in that D layout; stored back with one 16-byte store per lane
//   W^T X (k x m), W^T W (k x k)   contract over rows: A = W^T with lane (c, g), k-step s <-> W[row 4g+s][c], read back
//                                  transposed from a 1 KB per-wave LDS stage; B = X: lane (channel j, g) <-> X[row 4g+s][j]
//   residual  R^T-free form        R = W H per 16-channel block: A = H^T block, B = W^T fragment as above, D: lane
//                                  (row j, g), reg r <-> channel 4g+r of the block -- the layout X is read in
// X goes HBM -> registers (16-byte loads, whole rows, fully coalesced) -> a per-wave LDS stage (row-major, padded),
// from which both operand layouts are read; the two layouts of X are the transposition every formulation of the
// iteration needs once (X H^T contracts channels, W^T X contracts rows).  No workgroup barrier inside a pass: a
// wave's LDS operations execute in order and the stages are private to the wave.
//
// W cache.  Rows [0, lds_rows) of W live in LDS for the whole fit (row-major, ks values per row; every cached subtile is
// only ever touched by the wave that owns it, so no barrier is involved): they cost neither the read nor the write-back
// per iteration.  The write-back is what this kernel's traffic pays most for -- tools/ubench/wide_stream.hip: 256 B of X
// per row alone stream at 7.0 TB/s, with 32 B of W read and written back beside them the same rows take 1.49x as long.
//
// Padding: components are padded to KP (16 or 32) and channels to MP (a multiple of 16) with exact zeros, which the
// updates preserve (0 * 0 / EPSILON), so the padded problem's leading k x m block IS the unpadded iteration.
#pragma once
#include "nmf_kernels.hpp"

namespace hipnmf {

template <typename real>
struct WideArgs {
  const real* X;        // row-major [T][ldx], ldx * sizeof(real) % 16 == 0, 16-byte aligned
  long long x_bstride;  // elements between matrices
  long long ldx;
  real* W;              // row-major [T][ks], ks % 4 == 0 (components >= k are zero and stay zero)
  long long w_bstride;
  real* H;              // [B][k][m]
  real* err_out;        // [B] or nullptr
  int* n_iter_out;      // [B] or nullptr
  real* sse_col_out;    // [B][m] or nullptr
  real* xsq_col_out;    // [B][m] or nullptr
  const long long* ragged;  // [B][4] = {T_b, X offset, unused, W offset} (elements) or nullptr
  int T, m, k, ks, xchunks;  // xchunks: 16-byte pieces of a row of X that hold data (the rest of MP reads as zero)
  int max_iter, check_every, update_h;
  int lds_rows;  // rows [0, lds_rows) of W (a multiple of 16) stay in LDS for the whole fit; launch-wide capacity
  // row-sliced mode (few long matrices: grid (S, B), one launch per phase -- the wide counterpart of slice_pass / hupdate /
  // slice_resid of nmf_kernels.hpp):  mode 0 = the whole fit in one workgroup per matrix;  1 = ONE update pass over rows
  // [slice * rows_per_slice, ...) and the slice's record [W^T X | W^T W] to part[(b S + slice) REC];  2 = the residual pass
  // over the slice, per-column sse | xsq to colpart[(b S + slice) 2 MP]
  int mode, S, rows_per_slice;
  real* part;
  real* colpart;
  const real* state;  // [B][8]: entry 3 != 0 = matrix converged (its slices return at once), or nullptr
  real tol, l1w, l2w, l1h, l2h;
};

constexpr int wide_pow2_floor(int v) {
  int p = 1;
  while (2 * p <= v) p *= 2;
  return p;
}

template <typename real>
struct WideMma;
template <>
struct WideMma<float> {
  using acc = float __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc mma(float a, float b, acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // logical row carried by lane i of an A operand such that D's (g, r) means row 4 g + r
  static __device__ __forceinline__ int arow(int i) { return i; }
};
template <>
struct WideMma<double> {
  using acc = double __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc mma(double a, double b, acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // v_mfma_f64_16x16x4_f64 returns D[g + 4 r][j]: rows permuted on the way in give 4 g + r on the way out
  static __device__ __forceinline__ int arow(int i) { return 4 * (i & 3) + (i >> 2); }
};

template <typename real, int MP, int KP>
struct WideCfg {
  static constexpr int VEC = 16 / (int)sizeof(real);  // elements per 16-byte piece
  static constexpr int NKB = KP / 16, NCB = MP / 16;
  static constexpr int CPR = MP / VEC;                                  // pieces per row
  static constexpr int RPL = wide_pow2_floor(64 / CPR) > 16 ? 16 : wide_pow2_floor(64 / CPR);  // rows per load
  static constexpr int NLD = 16 / RPL;                                  // load instructions per subtile
  static constexpr int SX = MP + VEC;                                   // row stride of the X stage and of H in LDS
  static constexpr int SW = KP + 4;                                     // row stride of the W stage
  static constexpr int NS1 = MP / (4 * VEC);                            // LDS reads per subtile for H X^T
  static constexpr int XS = 16 * SX, WS = 16 * SW;
  static constexpr int REC = KP * MP + KP * KP;                         // per-wave record of [W^T X | W^T W]
  static constexpr int PERWAVE = (XS + WS > REC ? XS + WS : REC);
  static constexpr int COMMON = KP * SX + KP * KP + KP * MP + KP * KP + 2 * MP + 8;
  static_assert(MP % 16 == 0 && MP >= 16 && MP <= 256 && (KP == 16 || KP == 32), "unsupported wide shape");
  static_assert(CPR <= 64, "a row must fit one load instruction");
  __host__ __device__ static constexpr size_t smem_bytes(int nw) { return sizeof(real) * (size_t)(COMMON + nw * PERWAVE); }
};

// alignment the callers guarantee for an N-element piece: its size, capped at 16 bytes (12-byte pieces: 4)
template <typename real, int N>
constexpr size_t wide_piece_align() {
  constexpr size_t bytes = N * sizeof(real);
  return bytes >= 16 ? 16 : (bytes & (~bytes + 1));
}
template <typename real, int N>
__device__ __forceinline__ void wide_lds_read(const real* p, real (&out)[N]) {
  __builtin_memcpy(out, __builtin_assume_aligned(p, wide_piece_align<real, N>()), N * sizeof(real));
}
template <typename real, int N>
__device__ __forceinline__ void wide_lds_write(real* p, const real (&in)[N]) {
  __builtin_memcpy(__builtin_assume_aligned(p, wide_piece_align<real, N>()), in, N * sizeof(real));
}
// 16 / 32-byte buffer store of four consecutive elements
template <typename real>
__device__ __forceinline__ void wide_store4(rsrc_t r, unsigned voff, unsigned soff, const real (&v)[4]) {
  using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
  if constexpr (sizeof(real) == 4) {
    u32x4 u;
    __builtin_memcpy(&u, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
  } else {
    u32x4 u0, u1;
    __builtin_memcpy(&u0, &v[0], 16);
    __builtin_memcpy(&u1, &v[2], 16);
    __builtin_amdgcn_raw_buffer_store_b128(u0, r, voff, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b128(u1, r, voff + 16u, soff, 0);
  }
}
// orders a wave's LDS writes before its following LDS reads of other lanes' data (compiler-level; the LDS itself
// executes a wave's operations in order)
__device__ __forceinline__ void wide_wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Cache policy of the X stream (buf_load: 2 = non-temporal).  The matrices being worked on do not fit the Infinity Cache
// (2.56 MB of X each at 64 channels, two per CU), but their W does (320 KB each at k = 8): streamed non-temporal, X no
// longer evicts it and the read-modify-write of W stays on the die.  tools/ubench/wide_stream.hip, this traffic with no
// arithmetic: 4.97 TB/s with the default policy, 5.88 TB/s non-temporal (X alone: 6.3 -> 7.0 TB/s).
#ifndef HIPNMF_WIDE_X_AUX
#define HIPNMF_WIDE_X_AUX 2
#endif

template <typename real, int MP, int KP>
struct WideTile {
  using C = WideCfg<real, MP, KP>;
  real xg[C::NLD][C::VEC];  // the subtile of X as loaded: piece (lane % CPR) of row n RPL + lane / CPR
  real w[C::NKB][4];        // W fragment: row j, components 16 kb + 4 g .. + 3
};

// WPE: waves per SIMD the instance is compiled for.  2 caps the allocation at 256 registers, which also makes hipcc
// select the VGPR-destination form of the MFMAs (with 512 registers available it parks the accumulators in AGPRs and
// pays a v_accvgpr_read / _write per value the VALU touches: 46 of ~100 VALU instructions per subtile in the first build).
// NSET: register sets of subtile loads a wave keeps in flight (the subtile being worked on has been staged, so NSET
// further ones are on their way).
// LOSS: 0 = Frobenius (beta_loss = 2), 1 = Kullback-Leibler (beta_loss = 1; _nmf.py:556-591, 642-684): W *= ((X / WH) H^T) /
// rowsum(H), H *= (W^T (X / W'H)) / colsum(W') with W' the updated W; both reconstructions and both products on the pipe.
template <typename real, int MP, int KP, int NW, bool HREG, int WPE, int NSET, int LOSS = 0>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE > 1 ? 8 : 1)))
fit_wide_kernel(WideArgs<real> a) {
  using C = WideCfg<real, MP, KP>;
  using M = WideMma<real>;
  using acc = typename M::acc;
  using Tile = WideTile<real, MP, KP>;
  constexpr int VEC = C::VEC, NKB = C::NKB, NCB = C::NCB, SX = C::SX, SW = C::SW, NLD = C::NLD, RPL = C::RPL,
                CPR = C::CPR, NS1 = C::NS1, NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  real* const sH = reinterpret_cast<real*>(smem_raw);  // [KP][SX]
  real* const sHHt = sH + KP * SX;                     // [KP][KP]
  real* const sA = sHHt + KP * KP;                     // [KP][MP]   W^T X summed over the waves
  real* const sB = sA + KP * MP;                       // [KP][KP]   W^T W   (directly behind sA: one index space)
  real* const sPart = sB + KP * KP;                    // [2 MP + 8] per-column sse | xsq of the residual pass
  real* const wv0 = sPart + 2 * MP + 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  real* const xs = wv0 + wave * C::PERWAVE;  // [16][SX] this wave's X stage
  real* const wst = xs + C::XS;              // [16][SW] this wave's W stage

  const int b = blockIdx.x;
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  real* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
  int T = a.T;
  if (a.ragged) {
    const long long* d = a.ragged + 4LL * b;
    T = (int)d[0];
    Xb = a.X + d[1];
    Wb = a.W + d[3];
  }
  const int m = a.m, k = a.k, ks = a.ks;
  const int slice = blockIdx.y;
  if (a.mode != 0) {  // a slice is a matrix of its own for everything row-local
    if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
    const int row_begin = slice * a.rows_per_slice;
    int rows = T - row_begin;
    if (rows > a.rows_per_slice) rows = a.rows_per_slice;
    if (rows <= 0) rows = 0;
    Xb += (long long)row_begin * a.ldx;
    Wb += (long long)row_begin * ks;
    T = rows;
  }
  const int ntiles = (T + 15) / 16;
  real* const wcache = wv0 + NW * C::PERWAVE;  // [lds_rows][ks]
  const int ncached = (a.lds_rows / 16 < ntiles) ? a.lds_rows / 16 : ntiles;  // subtiles whose W lives in LDS
  const unsigned ldx_b = (unsigned)(a.ldx * (long long)sizeof(real));
  const unsigned ldw_b = (unsigned)ks * (unsigned)sizeof(real);

  // per-lane addressing.  Rows beyond the matrix are masked by the buffer descriptors, not by the lanes: the
  // descriptor of a subtile starts at its first row and ends with the matrix (scalar arithmetic), so a lane offset is
  // in range exactly when its row exists; pieces of a row beyond the data get the out-of-range sentinel once.
  const int xl_row = lane / CPR, xl_chunk = lane % CPR;
  const bool xl_active = lane < RPL * CPR;
  unsigned xvoff[NLD];
#pragma unroll
  for (int n = 0; n < NLD; ++n)
    xvoff[n] = (xl_active && xl_chunk < a.xchunks) ? (unsigned)(n * RPL + xl_row) * ldx_b + (unsigned)xl_chunk * 16u : OOB;
  real* const xs_put = xs + xl_row * SX + xl_chunk * VEC;
  unsigned wvoff[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
    wvoff[kb] = (16 * kb + 4 * g < ks) ? (unsigned)((j * ks + 16 * kb + 4 * g) * (int)sizeof(real)) : OOB;
  const int ar = M::arow(j);  // logical row this lane carries in an A operand
  real* const wc_lane = wcache + j * ks + 4 * g;  // this lane's fragment of cached subtile 0, component block 0
  // this lane's W fragment of subtile i: from / to the LDS cache or (already requested by issue) global memory
  auto get_w = [&](const Tile& t, int i, real (&w)[NKB][4]) __attribute__((always_inline)) {
    if (i < ncached) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (16 * kb + 4 * g < ks) {
          wide_lds_read<real, 4>(wc_lane + i * 16 * ks + 16 * kb, w[kb]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) w[kb][r] = (real)0;
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) w[kb][r] = t.w[kb][r];
    }
  };
  const char* const xbase = reinterpret_cast<const char*>(Xb);
  char* const wbase = reinterpret_cast<char*>(Wb);
  auto x_rsrc = [&](int i) __attribute__((always_inline)) {  // rows [16 i, T) of X; empty beyond the matrix
    const int rows = i < ntiles ? T - 16 * i : 0;
    return make_rsrc(xbase + (long long)(rows > 0 ? 16 * i : 0) * ldx_b, (unsigned)rows * ldx_b);
  };
  auto w_rsrc = [&](int i) __attribute__((always_inline)) {  // (subtiles cached in LDS: empty, their loads move nothing)
    const int rows = (i < ntiles && i >= ncached) ? T - 16 * i : 0;
    return make_rsrc(wbase + (long long)(rows > 0 ? 16 * i : 0) * ldw_b, (unsigned)rows * ldw_b);
  };

  auto issue = [&](Tile& t, int i) __attribute__((always_inline)) {  // loads of subtile i (i >= ntiles: nothing moves)
    const rsrc_t xr = x_rsrc(i);
    const rsrc_t wr = w_rsrc(i);
#pragma unroll
    for (int n = 0; n < NLD; ++n) buf_load<real, VEC, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[n], 0u, t.xg[n]);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) buf_load<real, 4>(wr, wvoff[kb], 0u, t.w[kb]);
  };
  auto stage_x = [&](const Tile& t) __attribute__((always_inline)) {
    if (xl_active) {
#pragma unroll
      for (int n = 0; n < NLD; ++n) wide_lds_write<real, VEC>(xs_put + n * RPL * SX, t.xg[n]);
    }
    wide_wave_lds_fence();
  };

  // ---- the cached rows of W -> LDS (the copy is the row-major image itself; rows past the matrix are zero) ---------
  for (int idx = tid; idx < ncached * 16 * ks; idx += NT) wcache[idx] = (idx < T * ks) ? Wb[idx] : (real)0;
  // ---- H -> LDS (zero padded), H H^T ----------------------------------------------------------------------------
  for (int idx = tid; idx < KP * SX; idx += NT) {
    const int c = idx / SX, jj = idx % SX;
    sH[idx] = (c < k && jj < m) ? Hb[c * m + jj] : (real)0;
  }
  __syncthreads();
  auto compute_hht_lds = [&]() __attribute__((always_inline)) {  // call between barriers
    if constexpr (LOSS == 1) {  // rowsum(H), the W update's denominator (_nmf.py:577-581), in sHHt[0 .. KP)
      for (int c = tid; c < KP; c += NT) {
        real s = (real)0;
        for (int jj = 0; jj < MP; ++jj) s += sH[c * SX + jj];
        sHHt[c] = s;
      }
      return;
    }
    for (int idx = tid; idx < KP * KP; idx += NT) {
      const int c = idx / KP, c2 = idx % KP;
      real s = (real)0;
      for (int jj = 0; jj < MP; ++jj) s = fma_(sH[c * SX + jj], sH[c2 * SX + jj], s);
      sHHt[idx] = s;
    }
  };
  compute_hht_lds();
  __syncthreads();

  // A operands that change once per iteration
  real hha[NKB][NKB][4];  // H H^T: lane (i, g), k-step r of input block kbi <-> HHt[16 kbo + arow(i)][16 kbi + 4 g + r]
  real hreg[HREG ? NKB : 1][HREG ? NS1 : 1][VEC];
  real hsum[NKB][4];  // KL: rowsum(H) of the components this lane holds in the D layout (16 kb + 4 g + r)
  auto load_operands = [&]() __attribute__((always_inline)) {
    if constexpr (LOSS == 1) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) wide_lds_read<real, 4>(sHHt + 16 * kb + 4 * g, hsum[kb]);
      return;
    }
#pragma unroll
    for (int kbo = 0; kbo < NKB; ++kbo)
#pragma unroll
      for (int kbi = 0; kbi < NKB; ++kbi) wide_lds_read<real, 4>(sHHt + (16 * kbo + ar) * KP + 16 * kbi + 4 * g, hha[kbo][kbi]);
    if constexpr (HREG) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s = 0; s < NS1; ++s) wide_lds_read<real, VEC>(sH + (16 * kb + ar) * SX + s * 4 * VEC + g * VEC, hreg[kb][s]);
    }
  };
  load_operands();

  acc accA[NKB][NCB], accB[NKB][NKB];
  real wsum[NKB][4];  // KL: this lane's share of colsum(W) (its row, the components of its D registers)
  const acc zero = {(real)0, (real)0, (real)0, (real)0};
  // W H for the lane's row and channel block cb: A = H^T block (lane (channel i, g), k-step r <-> H[16 kb + 4 g + r][16 cb +
  // arow(i)]), B = the W fragment; D: lane (row j, g), register r <-> channel 16 cb + 4 g + r
  auto wh_block = [&](const real (&w)[NKB][4], int cb) __attribute__((always_inline)) -> acc {
    acc rec = zero;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) rec = M::mma(sH[(16 * kb + 4 * g + r) * SX + 16 * cb + ar], w[kb][r], rec);
    return rec;
  };
  auto kl_quot = [&](real x, real wh) __attribute__((always_inline)) -> real {  // X / max(WH, EPSILON) (_nmf.py:574-575)
    const real d = kl_floor(wh);
    if constexpr (sizeof(real) == 4)
      return hipnmf::kl_quot(x, d);  // (nmf_kernels.hpp: the bare reciprocal)
    else
      return x / d;
  };

  // ---- one subtile: W update (_nmf.py:540-554, 615-631) and the sums of W^T X / W^T W (:638-640) ------------------
  auto update_subtile = [&](Tile& t, int i, int inext, bool upd) __attribute__((always_inline)) {
    stage_x(t);
    real wold[NKB][4];
    get_w(t, i, wold);
    if (inext >= 0) issue(t, inext);  // the registers of this subtile are free again: request the one PF steps ahead
    if constexpr (LOSS == 1) {
      // numerator^T = H Q^T with Q = X / max(W H, eps): Q comes out of wh_block's D layout (lane (row j, g), register r <->
      // channel 16 cb + 4 g + r), which is the B operand of k-step (cb, r); A: lane (i, g) <-> H[16 kb + arow(i)][16 cb + 4 g + r]
      acc num[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) num[kb] = zero;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const acc wh = wh_block(wold, cb);
        real xv[4];
        wide_lds_read<real, 4>(xs + j * SX + 16 * cb + 4 * g, xv);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          real ha[4];
          wide_lds_read<real, 4>(sH + (16 * kb + ar) * SX + 16 * cb + 4 * g, ha);
#pragma unroll
          for (int r = 0; r < 4; ++r) num[kb] = M::mma(ha[r], kl_quot(xv[r], wh[r]), num[kb]);
        }
      }
      real wn[NKB][4];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        real nn[4], dd[4], qq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          real d = hsum[kb][r];
          if (a.l1w > (real)0) d = d + a.l1w;
          if (a.l2w > (real)0) d = d + a.l2w * wold[kb][r];
          dd[r] = (d == (real)0) ? eps_val<real>() : d;
          nn[r] = num[kb][r];
        }
        quotients<4>(nn, dd, qq);
#pragma unroll
        for (int r = 0; r < 4; ++r) wn[kb][r] = wold[kb][r] * qq[r];
      }
      if (i < ncached) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
          if (16 * kb + 4 * g < ks) wide_lds_write<real, 4>(wc_lane + i * 16 * ks + 16 * kb, wn[kb]);
      } else {
        const rsrc_t wr = w_rsrc(i);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) wide_store4<real>(wr, wvoff[kb], 0u, wn[kb]);
      }
      if (upd) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          wide_lds_write<real, 4>(wst + j * SW + 16 * kb + 4 * g, wn[kb]);
#pragma unroll
          for (int r = 0; r < 4; ++r) wsum[kb][r] += wn[kb][r];
        }
        // Q' = X / max(W' H, eps) with the updated rows, written over X in the stage (each lane replaces exactly what it
        // read), then W'^T Q' like the Frobenius W^T X
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const acc wh = wh_block(wn, cb);
          real xv[4], qv[4];
          wide_lds_read<real, 4>(xs + j * SX + 16 * cb + 4 * g, xv);
#pragma unroll
          for (int r = 0; r < 4; ++r) qv[r] = kl_quot(xv[r], wh[r]);
          wide_lds_write<real, 4>(xs + j * SX + 16 * cb + 4 * g, qv);
        }
        wide_wave_lds_fence();
        real wa[NKB][4];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int s = 0; s < 4; ++s) wa[kb][s] = wst[(4 * g + s) * SW + 16 * kb + ar];
        const real* xcol = xs + 4 * g * SX + j;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const real xb = xcol[s * SX + 16 * cb];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) accA[kb][cb] = M::mma(wa[kb][s], xb, accA[kb][cb]);
          }
      }
      wide_wave_lds_fence();
      return;
    }
    // numerator^T = H X^T, two accumulation chains per component block
    acc num0[NKB], num1[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) num0[kb] = num1[kb] = zero;
    const real* xrow = xs + j * SX + g * VEC;
    const real* hrow = sH + ar * SX + g * VEC;
    if constexpr (!HREG) asm volatile("" : "+v"(hrow));  // keeps the H reads inside the subtile loop
    // HOIST: request every LDS operand of a phase before its first product.  Left to itself the compiler reads one, waits for
    // it, multiplies, reads the next ... and every wait exposes a full LDS round trip (8 per subtile); done where the extra
    // live registers (16 per phase at 64 fp32 channels) fit the instance's budget
    constexpr bool HOIST = MP * (int)(sizeof(real) / 4) <= 64;
    real xball[HOIST ? NS1 : 1][VEC];
    if constexpr (HOIST) {
#pragma unroll
      for (int s = 0; s < NS1; ++s) wide_lds_read<real, VEC>(xrow + s * 4 * VEC, xball[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < NS1; ++s) {
      real xb[VEC];
      if constexpr (HOIST) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) xb[e] = xball[s][e];
      } else {
        wide_lds_read<real, VEC>(xrow + s * 4 * VEC, xb);
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        real ha[VEC];
        if constexpr (HREG) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) ha[e] = hreg[kb][s][e];
        } else {
          wide_lds_read<real, VEC>(hrow + 16 * kb * SX + s * 4 * VEC, ha);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if ((s & 1) == 0)
            num0[kb] = M::mma(ha[e], xb[e], num0[kb]);
          else
            num1[kb] = M::mma(ha[e], xb[e], num1[kb]);
        }
      }
    }
    // denominator^T = (H H^T) W^T
    acc den[NKB];
#pragma unroll
    for (int kbo = 0; kbo < NKB; ++kbo) {
      den[kbo] = zero;
#pragma unroll
      for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
        for (int r = 0; r < 4; ++r) den[kbo] = M::mma(hha[kbo][kbi][r], wold[kbi][r], den[kbo]);
    }
    // W *= num / den   (regularisation :616-619, zero guard :620)
    real wn[NKB][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const acc nsum = num0[kb] + num1[kb];
      real nn[4], dd[4], qq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        real d = den[kb][r];
        if (a.l1w > (real)0) d = d + a.l1w;
        if (a.l2w > (real)0) d = d + a.l2w * wold[kb][r];
        dd[r] = (d == (real)0) ? eps_val<real>() : d;
        nn[r] = nsum[r];
      }
      quotients<4>(nn, dd, qq);
#pragma unroll
      for (int r = 0; r < 4; ++r) wn[kb][r] = wold[kb][r] * qq[r];
    }
    if (i < ncached) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        if (16 * kb + 4 * g < ks) wide_lds_write<real, 4>(wc_lane + i * 16 * ks + 16 * kb, wn[kb]);
    } else {
      const rsrc_t wr = w_rsrc(i);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) wide_store4<real>(wr, wvoff[kb], 0u, wn[kb]);
    }
    if (upd) {
      // transpose the new rows through the wave's W stage: A operand lane (c, g), k-step s <-> W[row 4 g + s][c]
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) wide_lds_write<real, 4>(wst + j * SW + 16 * kb + 4 * g, wn[kb]);
      wide_wave_lds_fence();
      real wa[NKB][4], wb[NKB][4];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          wa[kb][s] = wst[(4 * g + s) * SW + 16 * kb + ar];
          if constexpr (sizeof(real) == 8)
            wb[kb][s] = wst[(4 * g + s) * SW + 16 * kb + j];
          else
            wb[kb][s] = wa[kb][s];
        }
      // W^T X: B operand lane (channel j, g), k-step s <-> X[row 4 g + s][16 cb + j]
      const real* xcol = xs + 4 * g * SX + j;
      real xcall[HOIST ? NCB : 1][4];
      if constexpr (HOIST) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int s = 0; s < 4; ++s) xcall[cb][s] = xcol[s * SX + 16 * cb];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          real xb;
          if constexpr (HOIST)
            xb = xcall[cb][s];
          else
            xb = xcol[s * SX + 16 * cb];
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb) accA[kb][cb] = M::mma(wa[kb][s], xb, accA[kb][cb]);
        }
#pragma unroll
      for (int kbo = 0; kbo < NKB; ++kbo)
#pragma unroll
        for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
          for (int s = 0; s < 4; ++s) accB[kbo][kbi] = M::mma(wa[kbo][s], wb[kbi][s], accB[kbo][kbi]);
    }
    wide_wave_lds_fence();  // the next subtile's stage writes stay behind this one's reads
  };

  // ---- ||X - W H||_F^2 per column and sum X^2 per column of the whole matrix -> sPart[0 .. 2 MP); barriers inside ----
  auto block_resid = [&]() __attribute__((always_inline)) {
    real sse[NCB][4], xsq[NCB][4];
    real kl = (real)0;  // LOSS == 1: generalised KL divergence, element by element as x log(x / wh) - x + wh (_nmf.py:138-161)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) sse[cb][r] = xsq[cb][r] = (real)0;
    for (int i = wave; i < ntiles; i += NW) {
      Tile t;
      issue(t, i);
      stage_x(t);
      real wr_[NKB][4];
      get_w(t, i, wr_);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        // R block = W H[:, 16 cb ..]: A = H^T block (lane (channel i, g), k-step r <-> H[16 kb + 4 g + r][16 cb + arow(i)])
        const acc rec = wh_block(wr_, cb);
        real xv[4];
        wide_lds_read<real, 4>(xs + j * SX + 16 * cb + 4 * g, xv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const real d = xv[r] - rec[r];
          sse[cb][r] = fma_(d, d, sse[cb][r]);
          xsq[cb][r] = fma_(xv[r], xv[r], xsq[cb][r]);
          if constexpr (LOSS == 1) {  // branch-free, as in resid_tile (nmf_kernels.hpp)
            const real x = xv[r], whv = rec[r];
            const real whc = whv < eps_val<real>() ? eps_val<real>() : whv;
            const real xs_ = x > eps_val<real>() ? x : eps_val<real>();
            const real lg = fma_(x, log_(xs_ / whc), whv - x);
            kl += (x > eps_val<real>()) ? lg : whv;
          }
        }
      }
      wide_wave_lds_fence();
    }
    // lanes j = 0..15 of one g hold partial sums of the same channels: butterfly over the low four lane bits
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        real s1 = sse[cb][r], s2 = xsq[cb][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          s1 += __shfl_xor(s1, off, WAVE);
          s2 += __shfl_xor(s2, off, WAVE);
        }
        sse[cb][r] = s1;
        xsq[cb][r] = s2;
      }
    if constexpr (LOSS == 1) {
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) kl += __shfl_xor(kl, off, WAVE);
    }
    real* rec = xs;  // [2][MP] (+ 1) record of this wave (the stage is idle now)
    if (LOSS == 1 && lane == 0) rec[2 * MP] = kl;
    if (j == 0) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          rec[16 * cb + 4 * g + r] = sse[cb][r];
          rec[MP + 16 * cb + 4 * g + r] = xsq[cb][r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * MP + (LOSS == 1 ? 1 : 0); idx += NT) {
      real s = wv0[idx];
      for (int w2 = 1; w2 < NW; ++w2) s += wv0[w2 * C::PERWAVE + idx];
      sPart[idx] = s;
    }
    __syncthreads();
  };
  auto error_from_part = [&]() __attribute__((always_inline)) -> real {
    if constexpr (LOSS == 1) {  // sqrt(2 KL(X || WH)) (_nmf.py:185-189)
      const real d = sPart[2 * MP];
      return sqrt_((real)2 * (d > (real)0 ? d : (real)0));
    }
    real tot = (real)0;
    for (int jj = 0; jj < MP; ++jj) tot += sPart[jj];
    return sqrt_(tot);
  };

  if (a.mode == 2) {  // residual of the slice: per-column sums to global memory, summed over the slices by wide_resid_finalize_kernel
    block_resid();
    real* out = a.colpart + ((long long)b * a.S + slice) * (2 * MP);
    for (int idx = tid; idx < 2 * MP; idx += NT) out[idx] = sPart[idx];
    return;
  }
  if (a.mode == 1) {  // one update pass over the slice, its record to global memory (wide_hupdate_kernel sums the slices)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) accA[kb][cb] = zero;
#pragma unroll
      for (int kb2 = 0; kb2 < NKB; ++kb2) accB[kb][kb2] = zero;
    }
    const bool upd1 = a.update_h != 0;
    Tile t1;
    issue(t1, wave);
    for (int i = wave; i < ntiles; i += NW) {
      update_subtile(t1, i, i + NW, upd1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!upd1) return;
    real* rec = xs;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) rec[(16 * kb + 4 * g + r) * MP + 16 * cb + j] = accA[kb][cb][r];
#pragma unroll
      for (int kb2 = 0; kb2 < NKB; ++kb2)
#pragma unroll
        for (int r = 0; r < 4; ++r) rec[KP * MP + (16 * kb + 4 * g + r) * KP + 16 * kb2 + j] = accB[kb][kb2][r];
    }
    __syncthreads();
    real* out = a.part + ((long long)b * a.S + slice) * C::REC;
    for (int idx = tid; idx < C::REC; idx += NT) {
      real sacc = wv0[idx];
      for (int w2 = 1; w2 < NW; ++w2) sacc += wv0[w2 * C::PERWAVE + idx];
      out[idx] = sacc;
    }
    return;
  }

  real err0 = (real)0, prev = (real)0;
  if (a.tol > (real)0) {
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
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) accA[kb][cb] = zero;
#pragma unroll
      for (int kb2 = 0; kb2 < NKB; ++kb2) accB[kb][kb2] = zero;
#pragma unroll
      for (int r = 0; r < 4; ++r) wsum[kb][r] = (real)0;
    }
    if constexpr (NSET > 1) {
      // pairs of subtiles in a loop without inner exits, then the odd one (which requests nothing).  The compiler's s_waitcnt
      // vmcnt is only exact -- "this set's loads are in, the other set's may still fly" -- when every path into the loop has the
      // two sets' requests in the same order: a conditional second half, a tail that re-requests set a after set b, or a
      // prologue whose two requests the scheduler swapped (hence the sched_barriers around them) made it wait for ALL loads
      // in flight at the top of every round, i.e. a prefetch distance of half a round
      int i = wave;
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
    // X does not depend on H, and this wave's first rows of W are final: request the next pass's first subtiles now
    if (it < a.max_iter) {
      __builtin_amdgcn_sched_barrier(0);
      issue(ta, wave);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NSET > 1) issue(tb, wave + NW);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (upd) {
      // per-wave record [W^T X | W^T W] over the wave's stages (idle between passes), fixed-order sum over the waves
      real* rec = xs;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) rec[(16 * kb + 4 * g + r) * MP + 16 * cb + j] = accA[kb][cb][r];
        if constexpr (LOSS == 1) {  // colsum(W): the 16 lanes j of one g hold the rows, butterfly over the low lane bits
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            real sw = wsum[kb][r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) sw += __shfl_xor(sw, off, WAVE);
            if (j == 0) rec[KP * MP + 16 * kb + 4 * g + r] = sw;
          }
        } else {
#pragma unroll
          for (int kb2 = 0; kb2 < NKB; ++kb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) rec[KP * MP + (16 * kb + 4 * g + r) * KP + 16 * kb2 + j] = accB[kb][kb2][r];
        }
      }
      __syncthreads();
      for (int idx = tid; idx < (LOSS == 1 ? KP * MP + KP : C::REC); idx += NT) {
        real s = wv0[idx];
        for (int w2 = 1; w2 < NW; ++w2) s += wv0[w2 * C::PERWAVE + idx];
        sA[idx] = s;  // sB follows sA
      }
      __syncthreads();
      // H *= (W^T X) / ((W^T W) H)   (_nmf.py:638-640, 701-728)
      constexpr int NH = (KP * MP + NT - 1) / NT;
      real nh[NH];
#pragma unroll
      for (int q = 0; q < NH; ++q) {
        const int idx = tid + q * NT;
        const int c = idx / MP, jj = idx % MP;
        nh[q] = (real)0;
        if (idx < KP * MP && c < k && jj < m) {
          real d;
          if constexpr (LOSS == 1) {  // H *= (W^T Q') / colsum(W)   (_nmf.py:663-684; colsum 0 -> 1)
            d = sB[c];
            if (d == (real)0) d = (real)1;
          } else {
            d = sB[c * KP] * sH[jj];
            for (int c2 = 1; c2 < k; ++c2) d = fma_(sB[c * KP + c2], sH[c2 * SX + jj], d);
          }
          const real hold = sH[c * SX + jj];
          if (a.l1h > (real)0) d = d + a.l1h;
          if (a.l2h > (real)0) d = d + a.l2h * hold;
          d = (d == (real)0) ? eps_val<real>() : d;
          nh[q] = hold * (sA[idx] / d);
          if constexpr (LOSS == 1) {
            if (nh[q] < (real)2.220446049250313e-16) nh[q] = (real)0;  // H[H < float64 eps] = 0 (_nmf.py:866-868)
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NH; ++q) {
        const int idx = tid + q * NT;
        if (idx < KP * MP) sH[(idx / MP) * SX + idx % MP] = nh[q];
      }
      __syncthreads();
      compute_hht_lds();
      __syncthreads();
      load_operands();
    }
    if (a.tol > (real)0 && (it % a.check_every) == 0) {
      block_resid();
      const real err = error_from_part();
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
    for (int idx = tid; idx < k * m; idx += NT) Hb[idx] = sH[(idx / m) * SX + idx % m];
  }
  // the cached rows of W back to global memory (block_resid above ended with a barrier: every wave's rows are final)
  for (int idx = tid; idx < ncached * 16 * ks && idx < T * ks; idx += NT) Wb[idx] = wcache[idx];
}

// ---- row-sliced mode: the phases between the slice passes ----------------------------------------------------------------
template <typename real>
struct WideSliceArgs {
  real* H;            // [B][k][m]
  const real* part;   // [B][S][rec]      rec = KP MP + KP KP:  [W^T X | W^T W] per slice
  const real* colpart;  // [B][S][2 MP]   sse | xsq per slice
  real* state;        // [B][8]: err0, prev, err, done, checks so far
  real* err_out;
  int* n_iter_out;
  real* sse_col_out;
  real* xsq_col_out;
  int m, k, MP, KP, S, max_iter, check_every, it;  // it: 0 = error at init, 1 = stop-rule check, -1 = final outputs
  real tol, l1h, l2h;
  int kl;  // big_resid_finalize_kernel only: column records of 3 MP values, error = sqrt(2 KL)
};

// H *= (W^T X) / ((W^T W) H) from the slice records, summed in slice order (_nmf.py:638-640, 701-728).  One workgroup per matrix.
// out[idx] = sum over the S slice records of in[slice][idx], idx < n, in slice order: a thread per sum, sixteen records in
// flight per thread (a chain of S dependent loads cost 175 us per iteration at S = 157; a wave per sum with a butterfly
// 55 us: six dependent cross-lane steps per sum; this form ~15 us)
template <typename real>
__device__ __forceinline__ void wide_sum_slices(const real* __restrict__ in, int n, int S, real* __restrict__ out) {
  for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
    const real* p = in + idx;
    real s = (real)0;
    int q = 0;
    for (; q + 16 <= S; q += 16) {
      real v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(long long)(q + u) * n];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; q < S; ++q) s += p[(long long)q * n];
    out[idx] = s;
  }
}

template <typename real>
__global__ void __launch_bounds__(1024) wide_hupdate_kernel(WideSliceArgs<real> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_h_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  const int rec = a.KP * a.MP + a.KP * a.KP;
  real* sAB = reinterpret_cast<real*>(wide_h_smem);  // [rec]
  real* sHo = sAB + rec;                               // [k][m] the old H
  real* Hb = a.H + (long long)b * a.k * a.m;
  const real* pb = a.part + (long long)b * a.S * rec;
  wide_sum_slices<real>(pb, rec, a.S, sAB);
  for (int idx = tid; idx < a.k * a.m; idx += blockDim.x) sHo[idx] = Hb[idx];
  __syncthreads();
  const real* sB = sAB + a.KP * a.MP;
  for (int idx = tid; idx < a.k * a.m; idx += blockDim.x) {
    const int c = idx / a.m, jj = idx % a.m;
    real d = sB[c * a.KP] * sHo[jj];
    for (int c2 = 1; c2 < a.k; ++c2) d = fma_(sB[c * a.KP + c2], sHo[c2 * a.m + jj], d);
    const real hold = sHo[idx];
    if (a.l1h > (real)0) d = d + a.l1h;
    if (a.l2h > (real)0) d = d + a.l2h * hold;
    d = (d == (real)0) ? eps_val<real>() : d;
    Hb[idx] = hold * (sAB[c * a.MP + jj] / d);
  }
}

// per-column sums over the slices -> error, stop rule (_nmf.py:872-884), outputs.  One workgroup per matrix.
template <typename real>
__global__ void __launch_bounds__(1024) wide_resid_finalize_kernel(WideSliceArgs<real> a) {
  __shared__ real cols[2 * 256];  // (MP <= 256: inst_wide_f32_xl.hip)
  const int b = blockIdx.x, tid = threadIdx.x;
  real* st = a.state + (long long)b * 8;
  const bool done = st[3] != (real)0;
  if (done && a.it != -1) return;
  wide_sum_slices<real>(a.colpart + (long long)b * a.S * 2 * a.MP, 2 * a.MP, a.S, cols);
  __syncthreads();
  if (tid == 0) {
    real tot = (real)0;
    for (int jj = 0; jj < a.MP; ++jj) tot += cols[jj];
    const real err = sqrt_(tot);
    if (a.it == 0) {
      st[0] = err;
      st[1] = err;
    } else if (a.it == 1) {
      st[4] += (real)1;
      if ((st[1] - err) / st[0] < a.tol) {
        st[3] = (real)1;
        st[5] = st[4] * (real)a.check_every;  // n_iter_ of this matrix
      }
      st[1] = err;
    } else {
      if (a.err_out) a.err_out[b] = err;
      if (a.n_iter_out) a.n_iter_out[b] = done ? (int)st[5] : a.max_iter;
    }
    st[2] = err;
  }
  if (a.it == -1) {
    // (a converged matrix's slices returned at once from the last residual pass: its sums are those of the check that
    //  stopped it -- W and H have not changed since)
    for (int jj = tid; jj < a.m; jj += blockDim.x) {
      if (a.sse_col_out) a.sse_col_out[(long long)b * a.m + jj] = cols[jj];
      if (a.xsq_col_out) a.xsq_col_out[(long long)b * a.m + jj] = cols[a.MP + jj];
    }
  }
}

// ---- W between the caller's layout and the kernel's row-major [T][ks] rows (once per fit each way) -----------------
// ext_layout 0: row-major [T][k] (batch stride T k); 1: component-major [k][ld] (batch stride ext_bstride); with
// `desc` (device copy of the caller's ragged descriptors {T_b, -, ld_b, W offset}) the component-major matrices are
// packed.  `wdesc` ([B][4], entry 3 = offset of the matrix in `wide`) or wide_bstride locate the kernel-side copy.
template <typename real>
__global__ void wide_w_convert_kernel(real* __restrict__ ext, int ext_layout, long long ext_bstride, long long ext_ld,
                                      real* __restrict__ wide, long long wide_bstride, int ks, int T, int k, int dir,
                                      const long long* __restrict__ desc, const long long* __restrict__ wdesc) {
  const int b = blockIdx.y;
  real* e = ext + (long long)b * ext_bstride;
  real* w = wide + (long long)b * wide_bstride;
  long long ld = ext_ld;
  if (desc) {
    T = (int)desc[4LL * b + 0];
    ld = desc[4LL * b + 2];
    e = ext + desc[4LL * b + 3];
    w = wide + wdesc[4LL * b + 3];
  }
  const long long n = (long long)T * ks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / ks;
    const int c = (int)(i % ks);
    real* src = ext_layout == 0 ? e + t * k + c : e + (long long)c * ld + t;
    if (dir == 0)
      w[i] = c < k ? *src : (real)0;
    else if (c < k)
      *src = w[i];
  }
}

}  // namespace hipnmf
