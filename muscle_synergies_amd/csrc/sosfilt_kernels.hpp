// sosfilt_kernels.hpp -- batched IIR filtering (cascaded second-order sections) on gfx950: the
// `digital_filter` / `linear_envelope` stage of the reference (src/muscle_synergies/analysis.py:252-432),
// i.e. scipy.signal.sosfilt (forward) and scipy.signal.sosfiltfilt (zero-lag: odd padding, steady-state
// initial conditions, forward then backward).  SURVEY.md section 8, row f-1.
//
// An IIR recursion is sequential in time, so the parallel axis is the series: one lane per (recording,
// channel) series, 64 series per wave, one wave per workgroup.  HBM is still streamed coalesced: a tile of
// 64 series x 64 samples is loaded row by row (64 lanes = 64 consecutive samples of one series), transposed
// through LDS (row stride 65 doubles: conflict-free both ways), filtered lane-per-series out of LDS, and
// stored row by row again.  The next tile's rows are in flight (registers) while the current tile is
// filtered.  All arithmetic is fp64 with every product and sum rounded separately, in scipy's order
// (direct form II transposed), so the fp64 result is bit-identical to scipy's for identical input.
//
// Zero-lag needs the whole forward output before the backward pass starts: it goes to an fp64 workspace
// [N][T + 2*edge] in HBM (written and re-read by the same wave, mostly from L2 for the tail).
#pragma once
#include <hip/hip_runtime.h>

namespace hipnmf {

constexpr int SOS_MAX_SECTIONS = 8;
constexpr int SOS_TT = 64;      // samples per tile (= lanes per wave)
constexpr int SOS_LD = 65;      // LDS row stride in doubles

struct SosArgs {
  const void* x;          // canonical channel-major [B][m][ld]
  long long bstride, ld;
  double* ws;             // [N][T + 2*edge] forward output over the extended signal (zero_lag only)
  void* y;                // [B][m][T]
  double sos[SOS_MAX_SECTIONS][6];
  double zi[SOS_MAX_SECTIONS][2];
  int T, m, N, edge, zero_lag, zero_center, rectify;
};

// one sample through the cascade; state z[s][0..1]; no fused multiply-adds (scipy's C loop has none)
template <int NS>
__device__ __forceinline__ double sos_step(double xc, double (&z)[NS][2], const double (&c)[NS][5]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const double xn = c[s][0] * xc + z[s][0];
    z[s][0] = c[s][1] * xc - c[s][3] * xn + z[s][1];
    z[s][1] = c[s][2] * xc - c[s][4] * xn;
    xc = xn;
  }
  return xc;
}

template <typename real, int NS>
__global__ void __launch_bounds__(64) sosfilt_kernel(SosArgs a) {
  __shared__ double tile[SOS_TT * SOS_LD];
  __shared__ double stat[64][3];  // per series of this wave: mean, first and last pre-processed sample
  const int lane = threadIdx.x;
  const int s0 = blockIdx.x * 64;
  const int T = a.T, edge = a.edge, L = T + 2 * edge, N = a.N;
  const int nrows = (N - s0 < 64) ? N - s0 : 64;  // series handled by this wave
  const real* __restrict__ xbase = static_cast<const real*>(a.x);
  auto series = [&](int r) -> const real* {  // wave-uniform
    const int s = s0 + r;
    return xbase + (long long)(s / a.m) * a.bstride + (long long)(s % a.m) * a.ld;
  };

  // ---- per-series mean (zero_center) and end samples ------------------------------------------------------
  for (int r = 0; r < nrows; ++r) {
    const real* __restrict__ xr = series(r);
    double mean = 0.0;
    if (a.zero_center) {
      double s = 0.0;
      for (int i = lane; i < T; i += 64) s += (double)xr[i];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
      mean = s / (double)T;
    }
    if (lane == 0) {
      double v0 = (double)xr[0] - mean, v1 = (double)xr[T - 1] - mean;
      if (a.rectify) {
        v0 = fabs(v0);
        v1 = fabs(v1);
      }
      stat[r][0] = mean;
      stat[r][1] = v0;
      stat[r][2] = v1;
    }
  }
  __syncthreads();

  double c[NS][5];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    c[s][0] = a.sos[s][0];
    c[s][1] = a.sos[s][1];
    c[s][2] = a.sos[s][2];
    c[s][3] = a.sos[s][4];
    c[s][4] = a.sos[s][5];
  }
  const int ntiles = (L + SOS_TT - 1) / SOS_TT;

  // ---- forward pass over the (odd-)extended signal ------------------------------------------------------------
  real pf[64];
  auto issue_fwd = [&](int k) {  // rows of tile k -> registers (raw samples; reflection applied on the index)
    const int i = k * SOS_TT + lane;
    int j = i - edge;
    if (j < 0) j = -j;
    if (j >= T) j = 2 * (T - 1) - j;
    const bool ok = i < L;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      pf[r] = (real)0;
      if (r < nrows && ok) pf[r] = series(r)[j];
    }
  };
  auto commit_fwd = [&](int k) {  // registers -> LDS as pre-processed fp64 (zero-centre, rectify, odd extension)
    const int i = k * SOS_TT + lane;
    const int j = i - edge;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      double v = (double)pf[r] - stat[r][0];
      if (a.rectify) v = fabs(v);
      if (j < 0)
        v = 2.0 * stat[r][1] - v;
      else if (j >= T)
        v = 2.0 * stat[r][2] - v;
      tile[r * SOS_LD + lane] = v;
    }
  };

  double z[NS][2];
  double ylast = 0.0;
  bool primed = false;
  issue_fwd(0);
  for (int k = 0; k < ntiles; ++k) {
    const int t0 = k * SOS_TT;
    __syncthreads();  // previous tile's row stores have read the LDS tile
    commit_fwd(k);
    __syncthreads();
    if (k + 1 < ntiles) issue_fwd(k + 1);
    const int nval = (L - t0 < SOS_TT) ? L - t0 : SOS_TT;
    if (!primed) {  // initial state: zi * ext[0] (sosfiltfilt) or zeros (sosfilt)
      const double x0 = tile[lane * SOS_LD];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        z[s][0] = a.zero_lag ? a.zi[s][0] * x0 : 0.0;
        z[s][1] = a.zero_lag ? a.zi[s][1] * x0 : 0.0;
      }
      primed = true;
    }
    if (nval == SOS_TT) {
#pragma unroll 8
      for (int n = 0; n < SOS_TT; ++n) {
        ylast = sos_step<NS>(tile[lane * SOS_LD + n], z, c);
        tile[lane * SOS_LD + n] = ylast;
      }
    } else {
      for (int n = 0; n < nval; ++n) {
        ylast = sos_step<NS>(tile[lane * SOS_LD + n], z, c);
        tile[lane * SOS_LD + n] = ylast;
      }
    }
    __syncthreads();
    const int i = t0 + lane;
    if (a.zero_lag) {
      if (i < L) {
#pragma unroll 8
        for (int r = 0; r < 64; ++r)
          if (r < nrows) a.ws[(long long)(s0 + r) * L + i] = tile[r * SOS_LD + lane];
      }
    } else if (i < T) {
      real* __restrict__ yb = static_cast<real*>(a.y);
#pragma unroll 8
      for (int r = 0; r < 64; ++r)
        if (r < nrows) yb[(long long)(s0 + r) * T + i] = (real)tile[r * SOS_LD + lane];
    }
  }
  if (!a.zero_lag) return;

  // ---- backward pass: the forward output reversed, initial state zi * y[L-1]; keep the central T samples ------
  __threadfence();  // this wave's own stores to ws are re-read below
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    z[s][0] = a.zi[s][0] * ylast;
    z[s][1] = a.zi[s][1] * ylast;
  }
  double pb[64];
  auto issue_bwd = [&](int k) {
    const int i = k * SOS_TT + lane;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      pb[r] = 0.0;
      if (r < nrows && i < L) pb[r] = a.ws[(long long)(s0 + r) * L + i];
    }
  };
  issue_bwd(ntiles - 1);
  for (int k = ntiles - 1; k >= 0; --k) {
    const int t0 = k * SOS_TT;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 64; ++r) tile[r * SOS_LD + lane] = pb[r];
    __syncthreads();
    if (k > 0) issue_bwd(k - 1);
    const int nval = (L - t0 < SOS_TT) ? L - t0 : SOS_TT;
    if (nval == SOS_TT) {
#pragma unroll 8
      for (int n = SOS_TT - 1; n >= 0; --n) tile[lane * SOS_LD + n] = sos_step<NS>(tile[lane * SOS_LD + n], z, c);
    } else {
      for (int n = nval - 1; n >= 0; --n) tile[lane * SOS_LD + n] = sos_step<NS>(tile[lane * SOS_LD + n], z, c);
    }
    __syncthreads();
    const int j = t0 + lane - edge;
    if (j >= 0 && j < T) {
      real* __restrict__ yb = static_cast<real*>(a.y);
#pragma unroll 8
      for (int r = 0; r < 64; ++r)
        if (r < nrows) yb[(long long)(s0 + r) * T + j] = (real)tile[r * SOS_LD + lane];
    }
  }
}

}  // namespace hipnmf
