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

#include <type_traits>

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

// mean of one series in fp64 with a FIXED order of the sums (256 threads; the value every thread returns is bit-identical whichever
// kernel calls this: the sequential and the time-parallel filter centre with the same number).  scratch: 4 doubles of LDS.
template <typename real>
__device__ __forceinline__ double series_mean(const real* __restrict__ xr, int T, double* scratch) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  constexpr int V = 16 / (int)sizeof(real);        // samples per 16-byte load
  int i;
  if ((reinterpret_cast<unsigned long long>(xr) & 15ull) == 0) {  // 16-byte loads, two in flight per thread
    struct alignas(16) Vec { real v[V]; };
    const Vec* __restrict__ xv = reinterpret_cast<const Vec*>(xr);
    const int nv = T / V;
    int q = threadIdx.x;
    for (; q + 256 < nv; q += 512) {
      const Vec a0 = xv[q], a1 = xv[q + 256];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        s0 += (double)a0.v[e];
        s1 += (double)a1.v[e];
      }
    }
    for (; q < nv; q += 256) {
      const Vec a0 = xv[q];
#pragma unroll
      for (int e = 0; e < V; ++e) s2 += (double)a0.v[e];
    }
    i = nv * V + threadIdx.x;
  } else {
    i = threadIdx.x;
    for (; i + 768 < T; i += 1024) {
      const real a0 = xr[i], a1 = xr[i + 256], a2 = xr[i + 512], a3 = xr[i + 768];
      s0 += (double)a0;
      s1 += (double)a1;
      s2 += (double)a2;
      s3 += (double)a3;
    }
  }
  for (; i < T; i += 256) s3 += (double)xr[i];
  double acc = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
  __syncthreads();
  return (scratch[0] + scratch[1] + scratch[2] + scratch[3]) / (double)T;
}

// per-series statistics: stat[s] = {mean (0 unless zero_center), first, last pre-processed sample}
template <typename real>
__global__ void __launch_bounds__(256) sos_stats_kernel(SosArgs a, double* __restrict__ stat) {
  __shared__ double scratch[4];
  const int s = blockIdx.x;
  const real* __restrict__ xr =
      static_cast<const real*>(a.x) + (long long)(s / a.m) * a.bstride + (long long)(s % a.m) * a.ld;
  double mean = 0.0;
  if (a.zero_center) mean = series_mean<real>(xr, a.T, scratch);
  if (threadIdx.x == 0) {
    const real mr = (real)mean;
    stat[3LL * s + 0] = (double)mr;
    stat[3LL * s + 1] = (double)sos_pre<real>(xr[0], mr, a.rectify);
    stat[3LL * s + 2] = (double)sos_pre<real>(xr[a.T - 1], mr, a.rectify);
  }
}

// =================================================================================================
// Round 2: sosfilt2_kernel -- the same filter, restructured around what bounds it.
//
// tools/ubench/sos_rate.hip: the recursion is bound by the LATENCY of its dependent fp64 chain (add -> mul -> sub ->
// add per sample and section, ~4.9 ns per dependent operation at one wave per SIMD), not by issue: 20.5 ns per
// sample for one section, 40.5 ns for two when one lane runs both sections in sequence, 29 ns when the sections sit
// on neighbouring lanes and the intermediate signal travels by DPP (row_shr:1) -- and then 29 ns for 4 sections too.
// And round 1's single wave spent ~40 % of its time not in the recursion at all but loading / converting / storing
// tiles.  So:
//   * LPS = 1, 2, 4 or 8 lanes per series (sections rounded up to a power of two; surplus lanes run the identity
//     section b = [1, 0, 0], a = [1, 0, 0], which is exact), S = 64 / LPS series per workgroup;
//   * section s of a series lags s samples behind section 0: at step n lane (series, s) filters sample n - s with
//     the output lane (series, s - 1) produced at step n - 1; the pipeline is drained at the end of every 64-sample
//     tile (LPS - 1 extra steps), so tiles stay independent units for the data mover;
//   * a SECOND wave in the workgroup moves the data: it loads the rows of tile k + 2, converts / centres / rectifies /
//     extends tile k + 1 into LDS and stores tile k - 1 while wave 0 runs the recursion on tile k (three LDS tiles,
//     one workgroup barrier per tile).
// Arithmetic per section and sample is unchanged (scipy's direct form II transposed, fp64, no fused multiply-adds), so
// the fp64 result is still bit-identical to scipy's.
template <int BANK_MASK>
__device__ __forceinline__ double sos_dpp_shr1(double keep, double src) {
  // lanes whose bank (lane % 4) is enabled take src of lane - 1 (inside their 16-lane row); the others keep `keep`
  const unsigned long long k = __builtin_bit_cast(unsigned long long, keep), v = __builtin_bit_cast(unsigned long long, src);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)k, (int)(unsigned)v, 0x111, 0xf, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(k >> 32), (int)(unsigned)(v >> 32), 0x111, 0xf, BANK_MASK, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// series per workgroup: 64 / LPS lanes are needed; one section per series (LPS = 1) still takes 32 series per
// workgroup (half of the recursion wave idle) so that 16 384 series give 512 workgroups, two per CU
template <int LPS>
constexpr int sos2_series() {
  return LPS == 1 ? 32 : 64 / LPS;
}
template <int LPS>
constexpr size_t sos2_smem_bytes() {
  return sizeof(double) * (3 * sos2_series<LPS>() * SOS_LD + 8);  // three tiles
}

// one step of the lane's section (no FMA contraction: scipy's C loop has none)
struct SosLane {
  double c0, c1, c2, c3, c4, z0, z1;
  __device__ __forceinline__ double step(double xin) {
#pragma clang fp contract(off)
    const double xn = c0 * xin + z0;
    z0 = c1 * xin - c3 * xn + z1;
    z1 = c2 * xin - c4 * xn;
    return xn;
  }
};

// recursion over one tile held in LDS row `row` (this lane's series), in place, in ascending position order (the
// backward pass hands over mirrored tiles).  sec = this lane's section (0 .. LPS-1).  EVERY lane stores its output at
// the position of the sample it has just filtered: section s overwrites sample j at step j + s, so the last section's
// value is the one that stays, and section 0 has read the input at step j or earlier -- no select on the store side,
// no dummy slot, and the store address is one per-lane base plus an immediate offset.  The recursion is bound by the
// instruction issue of a single wave (~2.2 ns per instruction): per step 9 fp64 operations, 2 DPP moves + 2 selects
// for the hand-off, one LDS store, 1/1 LDS load.
// The stores of consecutive recursion steps hit the SAME LDS position from different lanes (section s overwrites what
// section s - 1 stored one step earlier): their program order is the data flow.  hipcc pairs the stores of two steps into
// one ds_write2_b64, which performs the lower-address element of every lane first: in this ASCENDING pass that is step
// order, so the last section's value stays -- exact against scipy in every test and ~10 000 fuzz cases -- but it is the
// compiler's pairing, not the source, that decides (sosfilt3_kernel's descending pass came out reversed and needs the
// barrier).  A compiler barrier after every store pins the order in the source; measured cost 2.08 -> 2.35 ms (order 4
// zero-lag, 1024 x 16 x 20 000 fp32).  Round 4: ON by default -- the exact mode exists to be bit-identical to scipy, the
// time-parallel mode (sosfilt_scan.hpp) is the fast one, and a result that depends on an instruction pairing is not exact
// (ADVICE r03).  -DHIPNMF_SOS_FAST_STORES restores the unordered stores.
#ifndef HIPNMF_SOS_FAST_STORES
#define SOS_STORE_ORDER() asm volatile("" ::: "memory")
#else
#define SOS_STORE_ORDER()
#endif
template <int LPS>
__device__ __forceinline__ void sos2_run_tile(double* row, int sec, int nval, SosLane& f) {
  double xn = 0.0;
  // section 0 takes the sample from LDS, section s > 0 the output its left neighbour produced one step earlier
  // (DPP bank masks select groups of four consecutive lanes, not lanes modulo four: the choice needs a v_cndmask)
  auto input = [&](double lds_val) -> double {
    if constexpr (LPS == 1) return lds_val;
    const unsigned long long v = __builtin_bit_cast(unsigned long long, xn);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x111, 0xf, 0xf, true);
    const double sh = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    return sec == 0 ? lds_val : sh;
  };
  double* wrp = row - sec;  // wrp[n] = row[n - sec]: the sample this lane filters at step n (aliases row)
  auto masked_step = [&](int n) {  // prologue / epilogue / ragged tile: only lanes with 0 <= n - sec < nval are live
    const int j = n - sec;
    const int jr = n < nval ? n : nval - 1;
    const double xin = input(row[jr]);
    if (j >= 0 && j < nval) {
      xn = f.step(xin);
      wrp[n] = xn;
      SOS_STORE_ORDER();
    }
  };
  if (nval == SOS_TT) {
    int n = 0;
    for (; n < LPS - 1; ++n) masked_step(n);
    // main part: every lane is live.  Eight inputs are read ahead of the eight being filtered.
    double cur[8], nxt[8];
    constexpr int n_main_end = LPS - 1 + ((SOS_TT - (LPS - 1)) / 8) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) cur[e] = row[n + e];
    for (; n < n_main_end; n += 8) {
      if (n + 8 < n_main_end) {
#pragma unroll
        for (int e = 0; e < 8; ++e) nxt[e] = row[n + 8 + e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xn = f.step(input(cur[e]));
        wrp[n + e] = xn;
        SOS_STORE_ORDER();
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) cur[e] = nxt[e];
    }
    for (; n < SOS_TT + LPS - 1; ++n) masked_step(n);
  } else {
    for (int n = 0; n < nval + LPS - 1; ++n) masked_step(n);
  }
}

template <typename real, int LPS>
__global__ void __launch_bounds__(128) sosfilt2_kernel(SosArgs a, const double* __restrict__ stat_g, int ns) {
  constexpr int S = sos2_series<LPS>();  // series per workgroup
  constexpr int TILE = S * SOS_LD;   // doubles per LDS tile
  extern __shared__ __attribute__((aligned(16))) double sos_smem[];
  double* tile = sos_smem;                 // [3][TILE]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int s0 = blockIdx.x * S;
  const int T = a.T, edge = a.edge, L = T + 2 * edge, N = a.N;
  const int nrows = (N - s0 < S) ? N - s0 : S;
  const int ntiles = (L + SOS_TT - 1) / SOS_TT;
  auto nval_of = [&](int k) { return (L - k * SOS_TT < SOS_TT) ? L - k * SOS_TT : SOS_TT; };

  // ---- wave 1: data mover state ----------------------------------------------------------------------------------
  const real* __restrict__ xbase = static_cast<const real*>(a.x);
  const long long off0 = (long long)(s0 / a.m) * a.bstride + (long long)(s0 % a.m) * a.ld;
  const int ch0 = s0 % a.m;
  const long long jump = a.bstride - (long long)a.m * a.ld;
  const int sl = lane < nrows ? lane : nrows - 1;
  const real mean_l = (real)stat_g[3LL * (s0 + sl) + 0];
  const real first_l = (real)stat_g[3LL * (s0 + sl) + 1];
  const real last_l = (real)stat_g[3LL * (s0 + sl) + 2];
  // two register sets: the rows of a tile are requested two workgroup-steps before they are needed
  real pf[2][S];
  double pb[2][S];
  // forward output of the zero-lag filter: private to the kernel, so it is laid out tile by tile -- workgroup g,
  // tile k, row r, sample lane at ((g * ntiles + k) * S + r) * 64 + lane -- and every tile is one contiguous 16 KB
  // (S = 32) block written / read with 512-byte rows instead of S streams L * 8 bytes apart
  double* __restrict__ ws_wg = a.ws + (long long)blockIdx.x * ntiles * (S * SOS_TT);
  auto issue_fwd = [&](int k, auto P) {  // raw rows of tile k -> registers (reflection applied on the index)
    constexpr int p = decltype(P)::value;
    const int i = k * SOS_TT + lane;
    int j = i - edge;
    if (j < 0) j = -j;
    if (j >= T) j = 2 * (T - 1) - j;
    j = j < 0 ? 0 : j;
    long long off = off0;
    int ch = ch0;
#pragma unroll
    for (int r = 0; r < S; ++r) {  // branch-free: rows past the last series re-read the last valid row
      pf[p][r] = xbase[off + j];
      const bool more = r + 1 < nrows;
      long long step = a.ld;
      if (ch + 1 == a.m) step += jump;
      ch = more ? (ch + 1 == a.m ? 0 : ch + 1) : ch;
      off += more ? step : 0;
    }
  };
  auto commit_fwd = [&](int k, double* __restrict__ buf, auto P) {  // registers -> LDS: centre, rectify, odd extension
    constexpr int p = decltype(P)::value;
    const int i = k * SOS_TT + lane;
    const int j = i - edge;
    const bool left = j < 0, ext = left || j >= T;
    if (k * SOS_TT >= edge && k * SOS_TT + SOS_TT <= edge + T) {
#pragma unroll
      for (int r = 0; r < S; ++r)
        buf[r * SOS_LD + lane] = (double)sos_pre<real>(pf[p][r], lane_bcast(mean_l, r), a.rectify);
    } else {
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const real v = sos_pre<real>(pf[p][r], lane_bcast(mean_l, r), a.rectify);
        const real e0 = lane_bcast(first_l, r), e1 = lane_bcast(last_l, r);
        const real end = left ? e0 : e1;
        const real refl = (real)2 * end - v;
        buf[r * SOS_LD + lane] = (double)(ext ? refl : v);
      }
    }
  };
  auto store_y = [&](const double* __restrict__ buf, int j) {
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
  auto store_fwd = [&](int k, const double* __restrict__ buf) {
    if (a.zero_lag) {
      double* __restrict__ wp = ws_wg + (long long)k * (S * SOS_TT) + lane;
      double v[S];
#pragma unroll
      for (int r = 0; r < S; ++r) v[r] = buf[r * SOS_LD + lane];
#pragma unroll
      for (int r = 0; r < S; ++r) wp[r * SOS_TT] = v[r];
    } else {
      store_y(buf, k * SOS_TT + lane);
    }
  };
  auto issue_bwd = [&](int k, auto P) {
    constexpr int p = decltype(P)::value;
    const double* __restrict__ wp = ws_wg + (long long)k * (S * SOS_TT) + lane;
#pragma unroll
    for (int r = 0; r < S; ++r) pb[p][r] = wp[r * SOS_TT];
  };
  // The backward pass filters in descending time.  Its tiles are MIRRORED in LDS (sample offset l of tile k sits at
  // position nval - 1 - l), so that the recursion wave runs the very same ascending loop in both passes.
  auto write_bwd = [&](int k, double* __restrict__ buf, auto P) {
    constexpr int p = decltype(P)::value;
    const int nv = nval_of(k);
    const int pos = lane < nv ? nv - 1 - lane : lane;  // lanes past the end of the signal: unused positions
#pragma unroll
    for (int r = 0; r < S; ++r) buf[r * SOS_LD + pos] = pb[p][r];
  };
  auto store_y_mirrored = [&](int k, const double* __restrict__ buf) {
    const int nv = nval_of(k);
    const int j = k * SOS_TT + lane - edge;
    if (lane >= nv || j < 0 || j >= T) return;
    const int pos = nv - 1 - lane;
    real* __restrict__ yp = static_cast<real*>(a.y) + (long long)s0 * T + j;
    if (nrows == S) {
      double v[S];
#pragma unroll
      for (int r = 0; r < S; ++r) v[r] = buf[r * SOS_LD + pos];
#pragma unroll
      for (int r = 0; r < S; ++r) yp[(long long)r * T] = (real)v[r];
    } else {
      for (int r = 0; r < nrows; ++r) yp[(long long)r * T] = (real)buf[r * SOS_LD + pos];
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // ---- wave 0: recursion state -------------------------------------------------------------------------------------
  const int ser = (lane / LPS) < S ? lane / LPS : S - 1, sec = lane % LPS;  // surplus lanes shadow the last series
  SosLane f;
  double zi0 = 0.0, zi1 = 0.0;
  f.c0 = 1.0;
  f.c1 = f.c2 = f.c3 = f.c4 = 0.0;  // identity section for sec >= ns
#pragma unroll
  for (int q = 0; q < SOS_MAX_SECTIONS; ++q) {
    if (q < LPS && sec == q && q < ns) {
      f.c0 = a.sos[q][0];
      f.c1 = a.sos[q][1];
      f.c2 = a.sos[q][2];
      f.c3 = a.sos[q][4];
      f.c4 = a.sos[q][5];
      zi0 = a.zi[q][0];
      zi1 = a.zi[q][1];
    }
  }
  f.z0 = f.z1 = 0.0;

  // ---- forward pass over the (odd-)extended signal -----------------------------------------------------------------
  // tile t lives in register set t & 1 and LDS tile t % 3
  if (wave == 1) {
    issue_fwd(0, P0{});
    if (ntiles > 1) issue_fwd(1, P1{});
    commit_fwd(0, tile, P0{});
    if (ntiles > 2) issue_fwd(2, P0{});
  }
  __syncthreads();
  for (int k = 0; k <= ntiles; ++k) {
    if (wave == 0) {
      if (k < ntiles) {
        double* row = tile + (k % 3) * TILE + ser * SOS_LD;
        if (k == 0) {  // initial state: zi * ext[0] (sosfiltfilt) or zeros (sosfilt); surplus sections: zi = 0
          const double x0 = row[0];
          f.z0 = a.zero_lag ? zi0 * x0 : 0.0;
          f.z1 = a.zero_lag ? zi1 * x0 : 0.0;
        }
        sos2_run_tile<LPS>(row, sec, nval_of(k), f);
      }
    } else {
      auto mover = [&](auto P) {  // P = (k + 1) & 1: the set holding tile k + 1, refilled with tile k + 3
        if (k + 1 < ntiles) commit_fwd(k + 1, tile + ((k + 1) % 3) * TILE, P);
        if (k >= 1) store_fwd(k - 1, tile + ((k - 1) % 3) * TILE);
        if (k + 3 < ntiles) issue_fwd(k + 3, P);
      };
      if ((k + 1) & 1)
        mover(P1{});
      else
        mover(P0{});
    }
    __syncthreads();
  }
  if (!a.zero_lag) return;

  // ---- backward pass: the forward output reversed, initial state zi * y_fwd[L-1]; keep the central T samples ------
  // tile t lives in register set (ntiles - 1 - t) & 1
  __threadfence();  // the mover re-reads its own stores to ws
  if (wave == 1) {
    issue_bwd(ntiles - 1, P0{});
    if (ntiles > 1) issue_bwd(ntiles - 2, P1{});
    write_bwd(ntiles - 1, tile + ((ntiles - 1) % 3) * TILE, P0{});
    if (ntiles > 2) issue_bwd(ntiles - 3, P0{});
  }
  __syncthreads();
  for (int k = ntiles - 1; k >= -1; --k) {
    if (wave == 0) {
      if (k >= 0) {
        double* row = tile + (k % 3) * TILE + ser * SOS_LD;
        if (k == ntiles - 1) {
          const double yl = row[0];  // y_fwd[L-1] (mirrored tile)
          f.z0 = zi0 * yl;
          f.z1 = zi1 * yl;
        }
        sos2_run_tile<LPS>(row, sec, nval_of(k), f);
      }
    } else {
      auto mover = [&](auto P) {  // P = (ntiles - k) & 1: the set holding tile k - 1, refilled with tile k - 3
        if (k - 1 >= 0) write_bwd(k - 1, tile + ((k - 1) % 3) * TILE, P);
        if (k + 1 < ntiles) store_y_mirrored(k + 1, tile + ((k + 1) % 3) * TILE);
        if (k - 3 >= 0) issue_bwd(k - 3, P);
      };
      if ((ntiles - k) & 1)
        mover(P1{});
      else
        mover(P0{});
    }
    __syncthreads();
  }
}

}  // namespace hipnmf
