// nmf_small.hpp -- fit_small_kernel<real, CH, K>: one WAVE per matrix for short recordings (T <= 256 rows).
//
// The reference's own matrices are short: its tutorial factorises 200 x 8 (docs/source/tutorials/Finding muscle
// synergies.ipynb: time_normalize(reduce_to=200) on 8 muscles, float64, find_synergies(df, 2, 3, max_iter=50_000) with
// the stop rule live, ~3500 iterations per rank), and gait-phase segments (project/segment.py) are of that size too.
// fit_persistent_kernel spends a 256-thread workgroup on such a matrix: four waves with one 64-row tile each, two
// workgroup barriers and a serial epilogue per iteration -- 2.6 us per iteration for one matrix, 187 M matrix-it/s for
// a batch of 16 384 (fp64).  Here the whole matrix lives in the registers of ONE wave (lane l owns rows l, l + 64,
// l + 128, l + 192 of X and W), H and the k x k products sit in a few hundred bytes of LDS private to the wave, and an
// iteration has no barrier and no global memory access at all:
//   W update   (_nmf.py:540-554, 615-631)   den = W (H H^T) with the rows of H H^T broadcast from LDS, then per
//                                           component: numerator X H^T with the row of H broadcast from LDS, quotient
//   H update   (_nmf.py:638-640, 701-728)   per-lane partial sums of W^T X / W^T W over the lane's rows, one
//                                           reduce-scatter over the wave, H and H H^T rebuilt by the same wave
//   stop rule  (_nmf.py:872-884), reconstruction_err_ / per-column SSE for VAF as in the other kernels.
// Layouts: channel-major X (x[j * ldx + t]) and component-major W, the engine's canonical ones; ragged batches too.
#pragma once
#include "nmf_kernels.hpp"
#include "nmf_small_decl.hpp"

namespace hipnmf {

// NT: 64-row tiles per matrix held in registers.  4 (n_samples <= 256, ~130 registers: four waves per SIMD) is the reference's
// own size; 6 / 8 / 10 / 12 / 16 (n_samples <= 384 / 512 / 640 / 768 / 1 024; up to 512 registers, one wave per SIMD) extend the same kernel to
// the batches of a few hundred to a thousand samples in between, where a 512-thread workgroup per matrix spends most of
// an iteration in its epilogue: 16 384 x (16 x 300), k = 5, fp32: 56 (workgroup per matrix) / 63 M (fit_wide4_kernel) matrix-it/s.
constexpr int SMALL_MAX_T = SMALL_NT * WAVE;

template <typename real, int CH, int K, int NT = SMALL_NT>
__global__ void __launch_bounds__(64) fit_small_kernel(SolveArgs<real> a) {
  using C = Cfg<real, 1, CH, K>;
  constexpr int NB = C::NB;
  // tiles per group in the W update (the temporaries of a group live at once).  float64, 16 channels, k = 5 walks groups of TWO:
  // with groups of four hipcc (ROCm 7.2.0) spills w[0][1] by halves -- high dword to an AGPR, low dword to scratch -- and at the
  // loop entry reloads only the scratch half, so the first W update of tile 0 reads an uninitialised register
  // (profiles/r04_small_f64_miscompile.md has the ISA).  A register-allocator defect, not UB here; tools/isa_split_spill_lint.py
  // looks for it in every kernel of every build (muscle_synergies_amd/build.py refuses the library if it fires).
#ifndef HIPNMF_SMALL_F64_16_5_TG
#define HIPNMF_SMALL_F64_16_5_TG 2
#endif
  constexpr int TG = (sizeof(real) == 8 && CH == 16 && K == 5) ? HIPNMF_SMALL_F64_16_5_TG : (NT % 4 == 0) ? 4 : 2;
  static_assert(NT % TG == 0, "whole groups of tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<real, 1, CH, K> s(smem_raw, 1);
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  real* __restrict__ Hb = a.H + (long long)b * K * a.m;
  int T = a.T;
  long long ldx = a.ldx, ldw = a.ldw;
  if (a.ragged) {
    const long long* d = a.ragged + 4LL * b;
    T = (int)d[0];
    Xb = a.X + d[1];
    ldx = ldw = d[2];
    Wb = a.W + d[3];
  }
  const int m = a.m;

  // the whole matrix into registers (rows >= T and channels >= m read as zero)
  real x[NT][CH], w[NT][K];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = t * WAVE + lane;
    const bool ok = row < T;
#pragma unroll
    for (int j = 0; j < CH; ++j) x[t][j] = (ok && j < m) ? Xb[(long long)j * ldx + row] : (real)0;
#pragma unroll
    for (int c = 0; c < K; ++c) w[t][c] = ok ? Wb[(long long)c * ldw + row] : (real)0;
  }
  load_h_to_lds(s, Hb, m);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): a single wave needs no barrier, LDS operations execute in order
  compute_hht(s);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);

  // ||X - W H||^2 per column (+ sum X^2) -> s.part[0 .. 2 CH): lanes j < CH hold column j's sums afterwards
  auto residual_cols = [&]() __attribute__((always_inline)) {
    real sse[CH], xsq[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) sse[j] = xsq[j] = (real)0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      real rec[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) rec[j] = (real)0;
#pragma unroll
      for (int c = 0; c < K; ++c) {
        const real* hp = s.H + c * CH;
#pragma unroll
        for (int j = 0; j < CH; ++j) rec[j] = fma_(w[t][c], hp[j], rec[j]);
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const real d = x[t][j] - rec[j];
        sse[j] = fma_(d, d, sse[j]);
        xsq[j] = fma_(x[t][j], x[t][j], xsq[j]);
      }
    }
    real v[2 * CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      v[j] = sse[j];
      v[CH + j] = xsq[j];
    }
    wave_reduce_scatter<1, 2 * CH, real>(v, lane);
    if (lane < 2 * CH) s.part[lane] = v[0];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
  };
  auto error_from_part = [&]() __attribute__((always_inline)) -> real {
    real tot = (real)0;
    for (int j = 0; j < CH; ++j) tot += s.part[j];
    return sqrt_(tot);
  };

  real err0 = (real)0, prev = (real)0;
  if (a.tol > (real)0) {
    residual_cols();
    err0 = error_from_part();
    prev = err0;
  }
  const bool upd = a.update_h != 0;
  const bool reg = a.l1w > (real)0 || a.l2w > (real)0;
  int n_iter = 0;
  for (int it = 1; it <= a.max_iter; ++it) {
    n_iter = it;
#pragma unroll
    for (int tg = 0; tg < NT; tg += TG) {
      // denominator W (H H^T): rows of H H^T broadcast from LDS
      real den[TG][K];
#pragma unroll
      for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int c = 0; c < K; ++c) den[t][c] = (real)0;
#pragma unroll
      for (int c2 = 0; c2 < K; ++c2) {
        real hrow[K];
#pragma unroll
        for (int c = 0; c < K; ++c) hrow[c] = s.HHt[c2 * K + c];
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
          for (int c = 0; c < K; ++c) den[t][c] = fma_(w[tg + t][c2], hrow[c], den[t][c]);
      }
      // numerator X H^T per component (row of H broadcast from LDS)
      real num[TG][K];
#pragma unroll
      for (int c = 0; c < K; ++c) {
        real hc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) hc[j] = s.H[c * CH + j];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          real acc = x[tg + t][0] * hc[0];
#pragma unroll
          for (int j = 1; j < CH; ++j) acc = fma_(x[tg + t][j], hc[j], acc);
          num[t][c] = acc;
        }
      }
      // regularisation (_nmf.py:616-619), zero guard (:620), quotient and update (:622-629)
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        real quo[K];
#pragma unroll
        for (int c = 0; c < K; ++c) {
          real d = den[t][c];
          if (reg) {
            if (a.l1w > (real)0) d = d + a.l1w;
            if (a.l2w > (real)0) d = d + a.l2w * w[tg + t][c];
          }
          den[t][c] = (d == (real)0) ? eps_val<real>() : d;
        }
        quotients<K>(num[t], den[t], quo);
#pragma unroll
        for (int c = 0; c < K; ++c) w[tg + t][c] = w[tg + t][c] * quo[c];
      }
    }
    if (upd) {
      // W^T X and W^T W: partial sums over the lane's rows, one reduce-scatter over the wave, H update by the wave
      real accA[K][CH], accB[NB];
#pragma unroll
      for (int c = 0; c < K; ++c)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          real acc = w[0][c] * x[0][j];
#pragma unroll
          for (int t = 1; t < NT; ++t) acc = fma_(w[t][c], x[t][j], acc);
          accA[c][j] = acc;
        }
      int idx = 0;
#pragma unroll
      for (int c = 0; c < K; ++c)
#pragma unroll
        for (int c2 = c; c2 < K; ++c2) {
          real acc = w[0][c] * w[0][c2];
#pragma unroll
          for (int t = 1; t < NT; ++t) acc = fma_(w[t][c], w[t][c2], acc);
          accB[idx++] = acc;
        }
      wave_reduce_acc<real, 1, CH, K>(s.part, accA, accB);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      wave0_combine_and_update_h(s, 1, m, a.l1h, a.l2h);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    if (a.tol > (real)0 && (it % a.check_every) == 0) {
      residual_cols();
      const real err = error_from_part();
      if ((prev - err) / err0 < a.tol) break;
      prev = err;
    }
  }
  // reconstruction_err_ (_nmf.py:1628-1630) + per-column SSE / sum X^2 for VAF (analysis.py:654-662)
  residual_cols();
  if (lane == 0) {
    if (a.err_out) a.err_out[b] = error_from_part();
    if (a.n_iter_out) a.n_iter_out[b] = n_iter;
  }
  if (lane < m) {
    if (a.sse_col_out) a.sse_col_out[(long long)b * m + lane] = s.part[lane];
    if (a.xsq_col_out) a.xsq_col_out[(long long)b * m + lane] = s.part[CH + lane];
  }
  if (upd) {
    for (int i = lane; i < K * CH; i += WAVE) {
      const int c = i / CH, j = i % CH;
      if (j < m) Hb[c * m + j] = s.H[i];
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = t * WAVE + lane;
    if (row < T) {
#pragma unroll
      for (int c = 0; c < K; ++c) Wb[(long long)c * ldw + row] = w[t][c];
    }
  }
}

}  // namespace hipnmf
