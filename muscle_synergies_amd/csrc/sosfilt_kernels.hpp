// sosfilt_kernels.hpp -- batched IIR filtering (cascaded second-order sections) on gfx950: the
// `digital_filter` / `linear_envelope` stage of the reference (src/muscle_synergies/analysis.py:252-432),
// i.e. scipy.signal.sosfilt (forward) and scipy.signal.sosfiltfilt (zero-lag: odd padding, steady-state
// initial conditions, forward then backward).  SURVEY.md section 8, row f-1.
//
// An IIR recursion is sequential in time, so the parallel axis is the series: one lane per (recording,
// channel) series, S series per wave, one wave per workgroup.  HBM is still streamed coalesced: a tile of
// S series x 64 samples is loaded row by row (64 lanes = 64 consecutive samples of one series), transposed
// through LDS (row stride 65 doubles: conflict-free both ways), filtered lane-per-series out of LDS, and
// stored row by row again.  The recursion is bound by the latency of its dependent fp64 chain (about 40 cycles
// per sample and section), not by lane count; measured on MI355X for 16 384 series (1024 x 16 x
// 20 000 fp32, order 4, zero-lag): S = 64 (256 waves) 3.8 ms, S = 32 (512 waves) 3.15 ms, S = 16 (1024 waves)
// 4.1 ms -- with four single-wave workgroups per CU the cost per section doubles, so S = 32 it is.  The next tile's rows are in flight (registers) and the previous tile's rows are
// being stored (second LDS buffer) while the current tile is filtered.  All arithmetic is fp64 with every product and sum rounded separately, in scipy's order
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

// pre-processing of one raw sample in the precision of the samples (numpy / pandas compute the centring, the
// rectification and scipy's odd extension in the array's own dtype before sosfilt promotes to float64)
template <typename real>
__device__ __forceinline__ real sos_pre(real x, real mean, int rectify) {
  real v = x - mean;
  return rectify ? (real)fabs((double)v) : v;
}

// value of lane `r` (wave-uniform r) for every lane: v_readlane into an SGPR, no LDS round trip
__device__ __forceinline__ float lane_bcast(float v, int r) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r));
}
__device__ __forceinline__ double lane_bcast(double v, int r) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, r);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), r);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// per-series statistics: stat[s] = {mean (0 unless zero_center), first, last pre-processed sample}
template <typename real>
__global__ void __launch_bounds__(256) sos_stats_kernel(SosArgs a, double* __restrict__ stat) {
  __shared__ double scratch[4];
  const int s = blockIdx.x;
  const real* __restrict__ xr =
      static_cast<const real*>(a.x) + (long long)(s / a.m) * a.bstride + (long long)(s % a.m) * a.ld;
  double mean = 0.0;
  if (a.zero_center) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four loads in flight; fixed order of the sums
    int i = threadIdx.x;
    for (; i + 768 < a.T; i += 1024) {
      const real a0 = xr[i], a1 = xr[i + 256], a2 = xr[i + 512], a3 = xr[i + 768];
      s0 += (double)a0;
      s1 += (double)a1;
      s2 += (double)a2;
      s3 += (double)a3;
    }
    for (; i < a.T; i += 256) s0 += (double)xr[i];
    double acc = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
    __syncthreads();
    mean = (scratch[0] + scratch[1] + scratch[2] + scratch[3]) / (double)a.T;
  }
  if (threadIdx.x == 0) {
    const real mr = (real)mean;
    stat[3LL * s + 0] = (double)mr;
    stat[3LL * s + 1] = (double)sos_pre<real>(xr[0], mr, a.rectify);
    stat[3LL * s + 2] = (double)sos_pre<real>(xr[a.T - 1], mr, a.rectify);
  }
}

// filter `nval` consecutive samples of this lane's series in place in LDS, eight at a time; the next eight
// are read before the current eight are filtered (LDS latency hidden behind the dependent recursion);
// DIR = +1 forward in time, -1 backward
template <int NS, int DIR>
__device__ __forceinline__ double sos_run_tile(double* __restrict__ row, int nval, double (&z)[NS][2],
                                               const double (&c)[NS][5], double ylast) {
  if (nval == SOS_TT) {
    double cur[8], nxt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cur[e] = row[DIR > 0 ? e : SOS_TT - 1 - e];
#pragma unroll
    for (int q = 0; q < SOS_TT; q += 8) {
      if (q + 8 < SOS_TT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) nxt[e] = row[DIR > 0 ? q + 8 + e : SOS_TT - 1 - q - 8 - e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) cur[e] = sos_step<NS>(cur[e], z, c);
#pragma unroll
      for (int e = 0; e < 8; ++e) row[DIR > 0 ? q + e : SOS_TT - 1 - q - e] = cur[e];
      ylast = cur[7];
#pragma unroll
      for (int e = 0; e < 8; ++e) cur[e] = nxt[e];
    }
  } else {
    for (int n = 0; n < nval; ++n) {
      const int idx = DIR > 0 ? n : nval - 1 - n;
      ylast = sos_step<NS>(row[idx], z, c);
      row[idx] = ylast;
    }
  }
  return ylast;
}

constexpr int SOS_SERIES = 32;  // series per wave (S)
template <int S>
constexpr size_t sos_smem_bytes() {
  return sizeof(double) * 2 * S * SOS_LD;
}

// Per tile k (both passes): wait for the rows of tile k -> LDS buffer k%2; issue the row stores of tile k-1
// (other buffer) and then the row loads of tile k+1; run the recursion on tile k.  Every global access thus
// has a whole tile's recursion to complete before the wave waits on it, and the wait (vmcnt(0)) never covers
// an operation that was issued just before it.
template <typename real, int NS, int S>
__global__ void __launch_bounds__(64) sosfilt_kernel(SosArgs a, const double* __restrict__ stat_g) {
  constexpr int TILE = S * SOS_LD;  // doubles per LDS tile
  extern __shared__ __attribute__((aligned(16))) double sos_smem[];
  double* tile = sos_smem;  // [2][TILE]
  const int lane = threadIdx.x;
  const int s0 = blockIdx.x * S;
  const int T = a.T, edge = a.edge, L = T + 2 * edge, N = a.N;
  const int nrows = (N - s0 < S) ? N - s0 : S;  // series handled by this wave
  const real* __restrict__ xbase = static_cast<const real*>(a.x);
  // series s0 + r starts at xbase + off_r; the offsets advance by ld inside a recording and jump at its end
  const long long off0 = (long long)(s0 / a.m) * a.bstride + (long long)(s0 % a.m) * a.ld;
  const int ch0 = s0 % a.m;
  const long long jump = a.bstride - (long long)a.m * a.ld;
  // lane r < S keeps the statistics of series s0 + r: mean, first and last pre-processed sample
  const int sl = lane < nrows ? lane : nrows - 1;
  const real mean_l = (real)stat_g[3LL * (s0 + sl) + 0];
  const real first_l = (real)stat_g[3LL * (s0 + sl) + 1];
  const real last_l = (real)stat_g[3LL * (s0 + sl) + 2];
  const bool active = lane < S;  // lanes that own a series in the recursion

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
  double* __restrict__ myrow = tile + (active ? lane : 0) * SOS_LD;

  // ---- forward pass over the (odd-)extended signal ------------------------------------------------------------
  real pf[S];
  auto issue_fwd = [&](int k) {  // rows of tile k -> registers (raw samples; reflection applied on the index)
    const int i = k * SOS_TT + lane;
    int j = i - edge;
    if (j < 0) j = -j;
    if (j >= T) j = 2 * (T - 1) - j;
    j = j < 0 ? 0 : j;  // lanes past the end of the extended signal: any valid address, value unused
    // branch-free on purpose: a load inside a conditional makes hipcc wait for it at the join, which would
    // serialise the row loads; rows past the last series re-read the last valid row instead
    long long off = off0;
    int ch = ch0;
#pragma unroll
    for (int r = 0; r < S; ++r) {
      pf[r] = xbase[off + j];
      const bool more = r + 1 < nrows;
      long long step = a.ld;
      if (ch + 1 == a.m) step += jump;
      ch = more ? (ch + 1 == a.m ? 0 : ch + 1) : ch;
      off += more ? step : 0;
    }
  };
  auto commit_fwd = [&](int k, double* __restrict__ buf) {  // registers -> LDS: centre, rectify, odd extension
    const int i = k * SOS_TT + lane;
    const int j = i - edge;
    const bool left = j < 0, ext = left || j >= T;
    if (k * SOS_TT >= edge && k * SOS_TT + SOS_TT <= edge + T) {  // interior tile (wave-uniform): no extension
#pragma unroll
      for (int r = 0; r < S; ++r)
        buf[r * SOS_LD + lane] = (double)sos_pre<real>(pf[r], lane_bcast(mean_l, r), a.rectify);
    } else {
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const real v = sos_pre<real>(pf[r], lane_bcast(mean_l, r), a.rectify);
        const real e0 = lane_bcast(first_l, r), e1 = lane_bcast(last_l, r);
        const real end = left ? e0 : e1;
        const real refl = (real)2 * end - v;  // odd extension about the end sample
        buf[r * SOS_LD + lane] = (double)(ext ? refl : v);
      }
    }
  };
  auto store_y = [&](const double* __restrict__ buf, int j) {  // LDS rows -> y[series][j] where j is in range
    if (j < 0 || j >= T) return;
    real* __restrict__ yp = static_cast<real*>(a.y) + (long long)s0 * T + j;
    if (nrows == S) {
      double v[S];
#pragma unroll
      for (int r = 0; r < S; ++r) v[r] = buf[r * SOS_LD + lane];
#pragma unroll
      for (int r = 0; r < S; ++r) yp[(long long)r * T] = (real)v[r];
    } else {
      for (int r = 0; r < nrows; ++r) yp[(long long)r * T] = (real)buf[r * SOS_LD + lane];
    }
  };
  auto store_fwd = [&](int k, const double* __restrict__ buf) {  // filtered tile k of the forward pass
    const int i = k * SOS_TT + lane;
    if (a.zero_lag) {
      if (i < L) {  // ws has room for S rows per wave: no row guard
        double* __restrict__ wp = a.ws + (long long)s0 * L + i;
        double v[S];
#pragma unroll
        for (int r = 0; r < S; ++r) v[r] = buf[r * SOS_LD + lane];
#pragma unroll
        for (int r = 0; r < S; ++r) wp[(long long)r * L] = v[r];
      }
    } else {
      store_y(buf, i);
    }
  };

  double z[NS][2];
  double ylast = 0.0;
  issue_fwd(0);
  for (int k = 0; k < ntiles; ++k) {
    double* buf = tile + (k & 1) * TILE;
    __syncthreads();
    commit_fwd(k, buf);
    __syncthreads();
    if (k > 0) store_fwd(k - 1, tile + ((k - 1) & 1) * TILE);
    if (k + 1 < ntiles) issue_fwd(k + 1);
    const int t0 = k * SOS_TT;
    const int nval = (L - t0 < SOS_TT) ? L - t0 : SOS_TT;
    if (active) {
      double* __restrict__ row = myrow + (k & 1) * TILE;
      if (k == 0) {  // initial state: zi * ext[0] (sosfiltfilt) or zeros (sosfilt)
        const double x0 = row[0];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          z[s][0] = a.zero_lag ? a.zi[s][0] * x0 : 0.0;
          z[s][1] = a.zero_lag ? a.zi[s][1] * x0 : 0.0;
        }
      }
      ylast = sos_run_tile<NS, +1>(row, nval, z, c, ylast);
    }
  }
  __syncthreads();
  store_fwd(ntiles - 1, tile + ((ntiles - 1) & 1) * TILE);
  if (!a.zero_lag) return;

  // ---- backward pass: the forward output reversed, initial state zi * y[L-1]; keep the central T samples ------
  __threadfence();  // this wave's own stores to ws are re-read below
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    z[s][0] = a.zi[s][0] * ylast;
    z[s][1] = a.zi[s][1] * ylast;
  }
  double pb[S];
  auto issue_bwd = [&](int k) {
    int i = k * SOS_TT + lane;
    i = i < L ? i : L - 1;  // branch-free (see issue_fwd); ws holds S rows for every wave
    const double* __restrict__ wp = a.ws + (long long)s0 * L + i;
#pragma unroll
    for (int r = 0; r < S; ++r) pb[r] = wp[(long long)r * L];
  };
  issue_bwd(ntiles - 1);
  for (int k = ntiles - 1; k >= 0; --k) {
    double* buf = tile + (k & 1) * TILE;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < S; ++r) buf[r * SOS_LD + lane] = pb[r];
    __syncthreads();
    if (k + 1 < ntiles) store_y(tile + ((k + 1) & 1) * TILE, (k + 1) * SOS_TT + lane - edge);
    if (k > 0) issue_bwd(k - 1);
    const int t0 = k * SOS_TT;
    const int nval = (L - t0 < SOS_TT) ? L - t0 : SOS_TT;
    if (active) sos_run_tile<NS, -1>(myrow + (k & 1) * TILE, nval, z, c, 0.0);
  }
  __syncthreads();
  store_y(tile, lane - edge);
}

}  // namespace hipnmf
