// envelope_kernels.hpp -- batched EMG envelope preprocessing on gfx950.
//
// Reference semantics (src/muscle_synergies/analysis.py): zero_center :230-249, rms :435-507
// (np.sqrt(np.convolve(x**2, ones(W)/W, "same"))), time_normalize :551-594 (scipy interp1d, linear, on
// linspace(0,1,T) -> linspace(0,1,n_out)), normalize :510-525 (divide by max |.| per column).
//
// One series = one channel of one recording.  The sliding-window mean is taken from an fp64 prefix sum of the
// squared (centred) samples, so every global access is coalesced and the window sum costs two loads per
// output regardless of W; sums are accumulated in fp64 for both fp32 and fp64 I/O.
// Kernels, in the order the launcher (hipnmf_envelope.hip) prefers them:
//   emg_wg_kernel     8 waves per series, samples and outputs in registers (full-length output, T <= ~20 K)
//   emg_wave_kernel   one wave per series, running prefix in an LDS ring, no barrier (any T, window <= 3584)
//   emg_fused_kernel  round 1: one 256-thread workgroup per series, tile + halo scanned in LDS (no window, long windows)
//   emg_prefix_kernel + emg_output_kernel   prefix array in HBM (windows that do not fit in LDS at all)
#pragma once
#include <hip/hip_runtime.h>

namespace hipnmf {

struct EnvArgs {
  const void* raw;     // canonical channel-major [B][m][ld]
  long long bstride, ld;
  double* prefix;      // workspace [B][m][T + 1]: prefix[i] = sum_{j < i} v_j^2
  double* chan_stat;   // workspace [B][m][2]: mean, max |out|
  void* out;           // [B][m][n_out]
  int T, m, window, zero_center, n_out, normalize;
  int resample_kind;   // HIPNMF_RESAMPLE_*: interp1d kind of the time normalisation (env_knot)
  // time normalisation table (env_resample_table_kernel -> emg_wave_kernel): per output q its left knot i0 and the
  // weight (xn - x0) / (x1 - x0) -- they depend on (T, n_out) only, not on the data
  const int* tab_i0;
  const double* tab_w;
};

// Time normalisation (analysis.py:551-594 -> scipy.interpolate.interp1d(linspace(0,1,T), y, kind=...) evaluated on
// linspace(0,1,n_out)): output q as  y = y[i0] + (y[i0+1] - y[i0]) * w  with 0 <= i0 <= T - 2.  Knots and abscissae as
// NumPy builds them (i * step, end point pinned to 1.0).  kind (HIPNMF_RESAMPLE_*):
//   0 linear     i0 = searchsorted(knots, xn, "left") - 1 clipped, w = (xn - x0) / (x1 - x0)   (interp1d._call_linear)
//   1 nearest    index = searchsorted(midpoints, xn, "left")  -- a tie goes DOWN (_call_nearest; midpoints = x[i]/2 + x[i+1]/2)
//   2 nearest-up                                  "right" -- a tie goes up
//   3 previous   last knot <= xn (_call_previousnext; also what kind="zero", the order-0 spline, evaluates to)
//   4 next       first knot >= xn
// The index kinds come out as (index, 0) or, for the last knot, (T - 2, 1); env_lerp returns the end values exactly.
__device__ __forceinline__ void env_knot(int T, int n_out, int q, int kind, int& i0, double& w) {
  const double step_out = (n_out > 1) ? 1.0 / (double)(n_out - 1) : 0.0;
  const double step_in = 1.0 / (double)(T - 1);  // T >= 2
  auto knot = [&](int i) { return (i == T - 1) ? 1.0 : (double)i * step_in; };
  const double xn = (q == n_out - 1 && n_out > 1) ? 1.0 : (double)q * step_out;
  int i = (int)floor(xn * (double)(T - 1));
  if (i > T - 2) i = T - 2;
  if (i < 0) i = 0;
  while (i > 0 && knot(i) >= xn) --i;  // i = searchsorted(side="left") - 1, clipped to [0, T - 2]
  while (i < T - 2 && knot(i + 1) < xn) ++i;
  if (kind == 0) {
    i0 = i;
    w = (xn - knot(i)) / (knot(i + 1) - knot(i));
    return;
  }
  int idx;  // the knot whose value is taken; knot(i) < xn <= knot(i + 1) here, except xn <= knot(0) at i = 0
  if (kind == 3) {
    idx = (knot(i + 1) <= xn) ? i + 1 : i;
  } else if (kind == 4) {
    idx = (knot(i) >= xn) ? i : i + 1;
  } else {
    const double mid = knot(i) / 2.0 + knot(i + 1) / 2.0;
    const bool up = kind == 2 ? (xn >= mid) : (xn > mid);
    idx = up ? i + 1 : i;
  }
  i0 = idx <= T - 2 ? idx : T - 2;
  w = idx <= T - 2 ? 0.0 : 1.0;
}
__device__ __forceinline__ double env_lerp(double y0, double y1, double w) {
  const double y = (y1 - y0) * w + y0;
  return w == 0.0 ? y0 : (w == 1.0 ? y1 : y);
}

// per-thread part of a strided sum over x[0 .. T): four independent accumulators so that four loads are in flight
// (a single running sum serialises on the load latency); fixed order: ((s0 + s1) + (s2 + s3))
template <typename real>
__device__ __forceinline__ double strided_sum(const real* __restrict__ x, int T, int first, int stride) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = first;
  if ((reinterpret_cast<unsigned long long>(x) & 15ull) == 0) {  // 16-byte loads, two in flight per thread
    constexpr int V = 16 / (int)sizeof(real);
    struct alignas(16) Vec { real v[V]; };
    const Vec* __restrict__ xv = reinterpret_cast<const Vec*>(x);
    const int nv = T / V;
    int q = first;
    for (; q + stride < nv; q += 2 * stride) {
      const Vec a0 = xv[q], a1 = xv[q + stride];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        s0 += (double)a0.v[e];
        s1 += (double)a1.v[e];
      }
    }
    for (; q < nv; q += stride) {
      const Vec a0 = xv[q];
#pragma unroll
      for (int e = 0; e < V; ++e) s2 += (double)a0.v[e];
    }
    for (i = nv * V + first; i < T; i += stride) s3 += (double)x[i];
    return (s0 + s1) + (s2 + s3);
  }
  for (; i + 3 * stride < T; i += 4 * stride) {
    const real a0 = x[i], a1 = x[i + stride], a2 = x[i + 2 * stride], a3 = x[i + 3 * stride];
    s0 += (double)a0;
    s1 += (double)a1;
    s2 += (double)a2;
    s3 += (double)a3;
  }
  for (; i < T; i += stride) s0 += (double)x[i];
  return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ double block_sum(double v, double* scratch /* [blockDim/64] */) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double tot = 0.0;
  for (int w = 0; w < nw; ++w) tot += scratch[w];  // fixed order
  return tot;
}

__device__ __forceinline__ double block_max(double v, double* scratch) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double tot = 0.0;
  for (int w = 0; w < nw; ++w) tot = fmax(tot, scratch[w]);
  return tot;
}

// Pass 1: per-channel mean (optional) and the prefix sums of the squared centred samples.
template <typename real>
__global__ void __launch_bounds__(256) emg_prefix_kernel(EnvArgs a) {
  __shared__ double scratch[8];
  __shared__ double wave_tot[4];
  const int ch = blockIdx.x, b = blockIdx.y;
  const real* __restrict__ x = static_cast<const real*>(a.raw) + (long long)b * a.bstride + (long long)ch * a.ld;
  double* __restrict__ ps = a.prefix + ((long long)b * a.m + ch) * ((long long)a.T + 1);
  double mean = 0.0;
  if (a.zero_center) {
    const double s = strided_sum<real>(x, a.T, threadIdx.x, blockDim.x);
    mean = block_sum(s, scratch) / (double)a.T;
  }
  if (threadIdx.x == 0) {
    a.chan_stat[((long long)b * a.m + ch) * 2] = mean;
    ps[0] = 0.0;
  }
  // blocked inclusive scan: tiles of blockDim.x * 4 consecutive samples, running carry in `base`
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double base = 0.0;
  const int tile = blockDim.x * 4;
  for (int t0 = 0; t0 < a.T; t0 += tile) {
    const int i0 = t0 + threadIdx.x * 4;
    double v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e;
      const double d = (i < a.T) ? (double)x[i] - mean : 0.0;
      v[e] = d * d;
    }
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    double incl = v[3];  // inclusive scan of the per-thread totals across the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double n = __shfl_up(incl, off, 64);
      if (lane >= off) incl += n;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    double woff = 0.0;
    for (int w = 0; w < wave; ++w) woff += wave_tot[w];
    const double excl = base + woff + incl - v[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e;
      if (i < a.T) ps[i + 1] = excl + v[e];
    }
    double tile_tot = 0.0;
    for (int w = 0; w < nw; ++w) tile_tot += wave_tot[w];
    base += tile_tot;
  }
}

// windowed RMS at sample i from the prefix sums: np.convolve(sq, ones(W)/W, "same")[i]
//   = (1/W) * sum_{j = i - (W-1 - (W-1)/2)}^{i + (W-1)/2} sq[j]   (zeros outside [0, T))
__device__ __forceinline__ double rms_at(const double* __restrict__ ps, int i, int T, int W) {
  const int hi = (W - 1) / 2, lo = (W - 1) - hi;
  int j0 = i - lo, j1 = i + hi + 1;  // [j0, j1)
  if (j0 < 0) j0 = 0;
  if (j1 > T) j1 = T;
  const double s = ps[j1] - ps[j0];
  return sqrt((s > 0.0 ? s : 0.0) / (double)W);
}

// Pass 2: RMS (or the plain centred signal when window == 0), optional linear time normalisation, optional
// max normalisation.  Writes out[b][ch][0 .. n_out).
template <typename real>
__global__ void __launch_bounds__(256) emg_output_kernel(EnvArgs a) {
  __shared__ double scratch[8];
  const int ch = blockIdx.x, b = blockIdx.y;
  const long long cidx = (long long)b * a.m + ch;
  const real* __restrict__ x = static_cast<const real*>(a.raw) + (long long)b * a.bstride + (long long)ch * a.ld;
  const double* __restrict__ ps = a.prefix + cidx * ((long long)a.T + 1);
  const double mean = a.chan_stat[cidx * 2];
  const int n_out = a.n_out > 0 ? a.n_out : a.T;
  real* __restrict__ o = static_cast<real*>(a.out) + cidx * (long long)n_out;
  auto value = [&](int i) -> double {
    if (a.window > 0) return rms_at(ps, i, a.T, a.window);
    return (double)x[i] - mean;
  };
  double vmax = 0.0;
  for (int q = threadIdx.x; q < n_out; q += blockDim.x) {
    double y;
    if (a.n_out > 0 && a.n_out != a.T) {
      // scipy interp1d(kind="linear") between linspace(0,1,T) knots, evaluated at q/(n_out-1)
      // knots as np.linspace builds them: i * step with the end point pinned to 1.0
      const double step_out = (n_out > 1) ? 1.0 / (double)(n_out - 1) : 0.0;
      const double xn = (q == n_out - 1 && n_out > 1) ? 1.0 : (double)q * step_out;
      if (a.T == 1) {
        y = value(0);
      } else {
        const double step_in = 1.0 / (double)(a.T - 1);
        auto knot = [&](int i) { return (i == a.T - 1) ? 1.0 : (double)i * step_in; };
        int i0;
        double wq;
        env_knot(a.T, n_out, q, a.resample_kind, i0, wq);
        const double x0 = knot(i0), x1 = knot(i0 + 1);
        const double y0 = value(i0), y1 = value(i0 + 1);
        y = a.resample_kind == 0 ? (y1 - y0) / (x1 - x0) * (xn - x0) + y0 : env_lerp(y0, y1, wq);
      }
    } else {
      y = value(q);
    }
    o[q] = (real)y;
    vmax = fmax(vmax, fabs(y));
  }
  if (a.normalize) {
    vmax = block_max(vmax, scratch);
    for (int q = threadIdx.x; q < n_out; q += blockDim.x) o[q] = (real)((double)o[q] / vmax);
  }
}

// =================================================================================================
// Fused version (default): no prefix array in HBM.
//   full-length output : tiles of ENV_TILE outputs; the squared centred samples of tile + halo go to LDS as
//                        fp64, are scanned there, and every output is a difference of two LDS prefix values;
//   resampled output   : the same tiles, but only the outputs whose left interpolation knot lies in the tile
//                        are produced (the full-length RMS never leaves LDS).
// Traffic per sample (fp32, W << tile): read x once for the mean (if centred), once (1 + W/tile) for the
// windows, write the output once (+ read/write it once more when normalising).
// =================================================================================================
#ifndef HIPNMF_ENV_TILE
#define HIPNMF_ENV_TILE 2048  // measured 1024 / 2048 / 4096 outputs per tile: 2.25 / 2.08 / 2.47 ms (1024 x 16 x 20 000 fp32)
#endif
constexpr int ENV_TILE = HIPNMF_ENV_TILE;

// exclusive block scan of n (<= capacity) fp64 values in LDS, in place: buf[i] <- sum_{j<i} buf[j]; buf[n] <- total
__device__ __forceinline__ void lds_exclusive_scan(double* buf, int n, double* wave_tot /* [blockDim/64] */) {
  const int nt = blockDim.x, per = (n + nt - 1) / nt;
  const int b0 = threadIdx.x * per;
  double run = 0.0;
  for (int i = b0; i < b0 + per && i < n; ++i) run += buf[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
  double incl = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  double base = incl - run;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  double tot = 0.0;
  for (int w = 0; w < nw; ++w) tot += wave_tot[w];
  for (int i = b0; i < b0 + per && i < n; ++i) {
    const double v = buf[i];
    buf[i] = base;
    base += v;
  }
  if (threadIdx.x == 0) buf[n] = tot;
  __syncthreads();
}

template <typename real>
__global__ void __launch_bounds__(256) emg_fused_kernel(EnvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char env_smem[];
  double* buf = reinterpret_cast<double*>(env_smem);  // [ENV_TILE + window + 1]
  __shared__ double scratch[8];
  const int ch = blockIdx.x, b = blockIdx.y;
  const long long cidx = (long long)b * a.m + ch;
  const real* __restrict__ x = static_cast<const real*>(a.raw) + (long long)b * a.bstride + (long long)ch * a.ld;
  const int T = a.T, W = a.window;
  const int n_out = a.n_out > 0 ? a.n_out : T;
  real* __restrict__ o = static_cast<real*>(a.out) + cidx * (long long)n_out;
  double mean = 0.0;
  if (a.zero_center) {
    const double s = strided_sum<real>(x, T, threadIdx.x, blockDim.x);
    mean = block_sum(s, scratch) / (double)T;
  }
  const int hi = W > 0 ? (W - 1) / 2 : 0, lo = W > 0 ? (W - 1) - hi : 0;
  double vmax = 0.0;
  const bool resample = a.n_out > 0 && a.n_out != T;
  const double step_out = (n_out > 1) ? 1.0 / (double)(n_out - 1) : 0.0;
  const double step_in = (T > 1) ? 1.0 / (double)(T - 1) : 0.0;
  auto knot = [&](int i) { return (i == T - 1) ? 1.0 : (double)i * step_in; };  // np.linspace(0, 1, T)[i]
  // raw samples of a tile (+ window halo) are requested one tile ahead, so that their latency hides behind the scan
  // and the output loop of the tile before (round 2; PFN values per thread)
  constexpr int PFN = (ENV_TILE + 1 + 1024 + 255) / 256;  // covers windows up to 1024 samples; longer ones load late
  real pf[PFN];
  const bool can_pf = W > 0 && ENV_TILE + W <= PFN * 256;
  auto issue = [&](int t0) {
#pragma unroll
    for (int q = 0; q < PFN; ++q) {
      int j = t0 - lo + threadIdx.x + q * 256;
      j = j < 0 ? 0 : (j >= T ? T - 1 : j);  // branch-free: out-of-range positions are masked when committed
      pf[q] = x[j];
    }
  };
  const double inv_w = W > 0 ? 1.0 / (double)W : 0.0;  // np.convolve(x^2, ones(W) / W): products by 1/W, no division
  if (can_pf) issue(0);
  for (int t0 = 0; t0 < T; t0 += ENV_TILE) {
    const int nout_t = (T - t0 < ENV_TILE) ? T - t0 : ENV_TILE;
    const int nval = nout_t + 1;  // one value past the tile: right neighbour for the interpolation
    if (W > 0) {
      const int nbuf = nval + W - 1;  // buf[e] <-> squared centred sample t0 - lo + e (0 outside [0, T))
      __syncthreads();
      if (can_pf) {
#pragma unroll
        for (int q = 0; q < PFN; ++q) {
          const int e = threadIdx.x + q * 256;
          const int j = t0 - lo + e;
          if (e < nbuf) {
            const double d = (j >= 0 && j < T) ? (double)pf[q] - mean : 0.0;
            buf[e] = d * d;
          }
        }
        if (t0 + ENV_TILE < T) issue(t0 + ENV_TILE);
      } else {
        for (int e = threadIdx.x; e < nbuf; e += blockDim.x) {
          const int j = t0 - lo + e;
          const double d = (j >= 0 && j < T) ? (double)x[j] - mean : 0.0;
          buf[e] = d * d;
        }
      }
      __syncthreads();
      lds_exclusive_scan(buf, nbuf, scratch);
    }
    auto value = [&](int i) -> double {  // i in [t0, t0 + nout_t]
      if (W > 0) {
        const double s = buf[i - t0 + W] - buf[i - t0];
        return sqrt((s > 0.0 ? s : 0.0) * inv_w);
      }
      return (i < T) ? (double)x[i] - mean : 0.0;
    };
    if (!resample) {
      if constexpr (sizeof(real) == 4) {
        if (W > 0) {
          // fp32 output: the fp64 window sum is rounded to float once and the root is taken in fp32 (v_sqrt_f32,
          // <= 1 ulp): within 1.5 ulp of rounding the fp64 root, at ~6 instructions per output instead of the ~25 of
          // the fp64 square root
          const float inv_wf = (float)inv_w;
          float vm = 0.f;
          for (int i = threadIdx.x; i < nout_t; i += blockDim.x) {
            const double sd = buf[i + W] - buf[i];
            const float y = __builtin_sqrtf(fmaxf((float)sd, 0.f) * inv_wf);
            o[t0 + i] = y;
            vm = fmaxf(vm, y);
          }
          vmax = fmax(vmax, (double)vm);
          continue;
        }
      }
      for (int i = threadIdx.x; i < nout_t; i += blockDim.x) {
        const double y = value(t0 + i);
        o[t0 + i] = (real)y;
        vmax = fmax(vmax, fabs(y));
      }
    } else {
      // scipy interp1d(kind="linear") from linspace(0,1,T) onto linspace(0,1,n_out): output q is produced by
      // the tile that holds its left knot i0
      for (int q = threadIdx.x; q < n_out; q += blockDim.x) {
        const double xn = (q == n_out - 1 && n_out > 1) ? 1.0 : (double)q * step_out;
        double y;
        if (T == 1) {
          if (t0 != 0) continue;
          y = value(0);
        } else {
          int i0;
          double wq;
          env_knot(T, n_out, q, a.resample_kind, i0, wq);
          if (i0 < t0 || i0 >= t0 + nout_t) continue;
          const double x0 = knot(i0), x1 = knot(i0 + 1);
          const double y0 = value(i0), y1 = value(i0 + 1);
          y = a.resample_kind == 0 ? (y1 - y0) / (x1 - x0) * (xn - x0) + y0 : env_lerp(y0, y1, wq);
        }
        o[q] = (real)y;
        vmax = fmax(vmax, fabs(y));
      }
    }
  }
  if (a.normalize) {
    vmax = block_max(vmax, scratch);  // barriers inside also order the writes of `o` above
    for (int q = threadIdx.x; q < n_out; q += blockDim.x) o[q] = (real)((double)o[q] / vmax);
  }
}

// =================================================================================================
// Wave-per-series version (round 2, default for 1 <= window and T >= 2): one WAVE owns a channel of a recording.
// PMC counters of emg_fused_kernel (profiles/README.md) showed it bound by instruction issue, not by memory: 77 VALU
// + 27 SALU instructions per sample and lane (index clamps of the halo loads, a runtime-length LDS scan, three
// workgroup barriers per tile, the halo of W - 1 samples squared twice).  Here
//   * a lane owns SPL CONSECUTIVE samples of a 64 SPL tile: the prefix over its samples is SPL register adds, the
//     prefix over the lanes six DPP steps, nothing of the scan goes through LDS and there is no barrier at all
//     (a single wave executes its LDS operations in order);
//   * the prefix is a RUNNING one (carried from tile to tile), so no halo is recomputed; the last `ring` > tile + W
//     prefix values live in an LDS ring (8 values per 9 slots: conflict-free for the blocked writes and for the
//     consecutive reads); every ENV_REBASE samples the carry is subtracted from the W + 1 live entries (exact by
//     Sterbenz' lemma) and reset, so a window sum carries an absolute error of at most ~eps times the sum of
//     ENV_REBASE + W squared samples (round 1's tiles: 2048 + W) -- invisible for real windows, but the root
//     amplifies it for windows of a few samples whose mean square is orders of magnitude below the signal's;
//   * outputs trail the inputs by (W - 1) / 2 + 1 samples and are produced 64 consecutive ones per instruction
//     (coalesced stores), each from two LDS reads.
// Same semantics as emg_fused_kernel (np.convolve "same" alignment, zero padding, NumPy's linspace knots, scipy's
// linear interp1d, division by the channel maximum); the summation order differs, results agree to ~1e-13 relative.
// =================================================================================================
constexpr int ENV_REBASE = 1 << 12;  // a multiple of every tile size

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double env_dpp_pull(double v) {  // lanes without a source, or in rows masked off, get 0
  const unsigned long long s = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)s, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(s >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double env_wave_inclusive_scan(double v) {
  v += env_dpp_pull<0x111, 0xf>(v);  // row_shr:1
  v += env_dpp_pull<0x112, 0xf>(v);  // row_shr:2
  v += env_dpp_pull<0x114, 0xf>(v);  // row_shr:4
  v += env_dpp_pull<0x118, 0xf>(v);  // row_shr:8  -> inclusive scan within each row of 16 lanes
  v += env_dpp_pull<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += env_dpp_pull<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ void env_wave_sync() {  // LDS operations of one wave execute in order: only the compiler needs telling
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
}

// State of one wave walking a series tile by tile (shared by emg_wave_kernel and emg_wg_kernel).
template <typename real, int SPL>
struct EnvWave {
  static constexpr int TILE = 64 * SPL, V = 16 / (int)sizeof(real), NV = SPL / V;
  static_assert(SPL == 4 || SPL == 8, "a lane's samples must stay inside one group of 8 ring entries");
  struct alignas(16) Vec { real v[V]; };
  const real* __restrict__ x;
  double* P;  // ring: prefix value j (sum of the squared centred samples before j) lives at slot(j)
  int T, W, lo, hi, mask, lane, first;  // first: the tile the walk started at (no prefix values exist before it)
  bool vec_ok;
  double mean, carry;

  __device__ __forceinline__ void init(const real* x_, double* P_, int ring, int T_, int W_, double mean_, int lane_, int first_) {
    x = x_;
    P = P_;
    T = T_;
    W = W_;
    hi = (W - 1) / 2;
    lo = (W - 1) - hi;
    mask = ring - 1;
    lane = lane_;
    first = first_;
    vec_ok = (reinterpret_cast<unsigned long long>(x_) & 15ull) == 0;
    mean = mean_;
    carry = 0.0;
  }
  __device__ __forceinline__ void set_mean(double mean_) { mean = mean_; }  // may follow init and the first loads: they do not need it
  __device__ __forceinline__ int slot(int j) const {
    const int s_ = j & mask;
    return s_ + (s_ >> 3);
  }
  // request the lane's SPL consecutive samples of the tile at t0 (zero past the end of the series)
  __device__ __forceinline__ void load(int t0, real (&v)[SPL]) const {
    const int j0 = t0 + lane * SPL;
    if (vec_ok && j0 + SPL <= T) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const Vec t = *reinterpret_cast<const Vec*>(x + j0 + q * V);
#pragma unroll
        for (int e = 0; e < V; ++e) v[q * V + e] = t.v[e];
      }
    } else {
#pragma unroll
      for (int c = 0; c < SPL; ++c) v[c] = (j0 + c < T) ? x[j0 + c] : (real)0;
    }
  }
  // prefix values of the tile at t0 (samples v) into the ring; afterwards P[j] is available for j < t0 + TILE
  template <typename F>
  __device__ __forceinline__ void tile_core(int t0, const real (&v)[SPL], F after_consume) {
    if (t0 > first && (t0 & (ENV_REBASE - 1)) == 0) {
      // re-base the running prefix: the live entries are P[t0 - W - 1 .. t0 - 1] (the time normalisation reads the
      // windows of two neighbouring samples, hence one entry more than W)
      for (int e = 1 + lane; e <= W + 1; e += 64) {
        const int j = t0 - e;
        if (j >= first) P[slot(j)] -= carry;
      }
      carry = 0.0;
      env_wave_sync();
    }
    // exclusive prefix over the lane's samples, then over the lanes
    double ex[SPL], run = 0.0;
    if (t0 + TILE <= T) {
#pragma unroll
      for (int c = 0; c < SPL; ++c) {
        const double d = (double)v[c] - mean;
        ex[c] = run;
        run += d * d;
      }
    } else {
      const int j0 = t0 + lane * SPL;
#pragma unroll
      for (int c = 0; c < SPL; ++c) {
        const double d = (j0 + c < T) ? (double)v[c] - mean : 0.0;
        ex[c] = run;
        run += d * d;
      }
    }
    after_consume();
    const double inc = env_wave_inclusive_scan(run);
    const double up = __shfl_up(inc, 1, 64);
    const double base = carry + (lane ? up : 0.0);
    {
      const int s0 = (t0 + lane * SPL) & mask;
      double* dst = P + s0 + (s0 >> 3);
#pragma unroll
      for (int c = 0; c < SPL; ++c) dst[c] = base + ex[c];
    }
    carry += __shfl(inc, 63, 64);
    env_wave_sync();
  }
  // sum of the squared centred samples in the window around i (np.convolve "same" alignment, zeros outside [0, T))
  __device__ __forceinline__ double window_sum(int i) const {
    const int jb = i - lo;
    return P[slot(i + hi + 1)] - P[slot(jb < 0 ? 0 : jb)];
  }
  // full tile of outputs emitted .. emitted + TILE (no clamping needed: emitted >= lo): all ring reads are issued
  // before the first use; the ring entry of slot j is 9 (j >> 3) + (j & 7) and j advances by 64 between a lane's outputs
  __device__ __forceinline__ void window_sums_full_tile(int emitted, double (&sd)[SPL]) const {
    const unsigned ja = (unsigned)(emitted + hi + 1 + lane), jb = (unsigned)(emitted - lo + lane);
    const unsigned ga = ja >> 3, gb = jb >> 3, gm = (unsigned)mask >> 3, ca = ja & 7u, cb = jb & 7u;
    double pa[SPL], pb[SPL];
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
      const unsigned ta = (ga + 8u * u) & gm, tb = (gb + 8u * u) & gm;  // 9 t + c as shifts (no v_mul_lo_u32)
      pa[u] = P[(ta << 3) + (ta + ca)];
      pb[u] = P[(tb << 3) + (tb + cb)];
    }
#pragma unroll
    for (int u = 0; u < SPL; ++u) sd[u] = pa[u] - pb[u];
  }
};

// y / vmax as the reference does it (float64 division): for fp32 outputs the correctly rounded fp32 quotient of two
// floats equals the fp64 quotient rounded to float (53 >= 2 * 24 + 2 bits: double rounding is innocuous), at a third
// of the cost; vmax is a maximum of floats there, hence exact in fp32
template <typename real>
__device__ __forceinline__ real env_scaled(real y, double vmax, float vmf) {
  if constexpr (sizeof(real) == 4) return y / vmf;
  else return (real)((double)y / vmax);
}

// scipy interp1d(kind="linear") from the knots linspace(0,1,T) onto linspace(0,1,n_out), knots built as NumPy builds
// them (i * step with the end point pinned to 1.0): left knot i0 = searchsorted(knots, xn, "left") - 1 clipped to
// [0, T - 2].  One thread per output; the table serves every series of the call.
__global__ void __launch_bounds__(256) env_resample_table_kernel(int T, int n_out, int kind, int* tab_i0, double* tab_w) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_out) return;
  int i0;
  double w;
  env_knot(T, n_out, q, kind, i0, w);
  tab_i0[q] = i0;
  tab_w[q] = w;
}

// Time-normalised output with left knot i0 and weight w = (xn - x0) / (x1 - x0) from the table: y = y0 + (y1 - y0) w
// (scipy evaluates (y1 - y0) / (x1 - x0) * (xn - x0) + y0: the same to an ulp of float64); both neighbours' windows
// must be in the ring.  For fp32 outputs the two roots are taken in fp32 like everywhere else in this file.
template <typename real, int SPL>
__device__ __forceinline__ double env_interp(const EnvWave<real, SPL>& wv, int i0, double w, double inv_w) {
  const double s0 = wv.window_sum(i0), s1 = wv.window_sum(i0 + 1);
  double y0, y1;
  if constexpr (sizeof(real) == 4) {
    const float inv_wf = (float)inv_w;
    y0 = (double)__builtin_amdgcn_sqrtf(fmaxf((float)s0, 0.f) * inv_wf);
    y1 = (double)__builtin_amdgcn_sqrtf(fmaxf((float)s1, 0.f) * inv_wf);
  } else {
    y0 = sqrt((s0 > 0.0 ? s0 : 0.0) * inv_w);
    y1 = sqrt((s1 > 0.0 ? s1 : 0.0) * inv_w);
  }
  return env_lerp(y0, y1, w);
}

template <typename real, int SPL>
__global__ void __launch_bounds__(64) emg_wave_kernel(EnvArgs a, int ring /* power of two > 64 SPL + window */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char env_smem[];
  using Wv = EnvWave<real, SPL>;
  using Vec = typename Wv::Vec;
  constexpr int TILE = Wv::TILE, V = Wv::V;
  const int lane = threadIdx.x;
  const int ch = blockIdx.x, b = blockIdx.y;
  const long long cidx = (long long)b * a.m + ch;
  const real* __restrict__ x = static_cast<const real*>(a.raw) + (long long)b * a.bstride + (long long)ch * a.ld;
  const int T = a.T, W = a.window;
  const int n_out = a.n_out > 0 ? a.n_out : T;
  real* __restrict__ o = static_cast<real*>(a.out) + cidx * (long long)n_out;

  Wv wv;
  wv.init(x, reinterpret_cast<double*>(env_smem), ring, T, W, 0.0, lane, 0);
  // Samples are requested TWO tiles ahead (two register sets, the loop below handles a pair of tiles per trip): with
  // one tile per wave in flight the 4096 waves of the chip keep 8 MB outstanding, which at 2-4 us of loaded memory
  // latency is 2.5-3 TB/s -- exactly what the time-normalised path measured.  The first two are requested here,
  // before the mean pass.
  real pfa[SPL], pfb[SPL];
  wv.load(0, pfa);
  wv.load(TILE, pfb);
  double mean = 0.0;
  if (a.zero_center) {
    // eight 16-byte loads in flight per lane; fixed summation order
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    int done = 0;
    if ((reinterpret_cast<unsigned long long>(x) & 15ull) == 0) {
      const Vec* __restrict__ xv = reinterpret_cast<const Vec*>(x);
      const int nv = T / V;
      int q0 = 0;
      for (; q0 + 8 * 64 <= nv; q0 += 8 * 64) {
        Vec t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = xv[q0 + q * 64 + lane];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[q] += (double)t[q].v[e];
      }
      for (int q = q0 + lane; q < nv; q += 64)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[0] += (double)xv[q].v[e];
      done = nv * V;
    }
    for (int i = done + lane; i < T; i += 64) acc[1] += (double)x[i];
    double s_ = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s_ += __shfl_xor(s_, off, 64);
    mean = s_ / (double)T;
  }

  wv.set_mean(mean);
  const int hi = wv.hi;
  const bool resample = a.n_out > 0 && a.n_out != T;
  const double inv_w = 1.0 / (double)W;  // np.convolve(x^2, ones(W) / W): products by 1/W, no division
  const float inv_wf = (float)inv_w;
  auto value = [&](int i) -> double {
    const double sd = wv.window_sum(i);
    return sqrt((sd > 0.0 ? sd : 0.0) * inv_w);
  };

  double vmax = 0.0;
  float vmaxf = 0.f;
  int emitted = 0;
  const int end_all = resample ? T - 1 : T;  // resampling: `emitted` counts left knots i0 in [0, T - 1)
  // time normalisation: the window of outputs qw .. qw + 63 and the one after it (table of env_resample_table_kernel)
  auto load_window = [&](int q0, int& i0v, double& wgt) __attribute__((always_inline)) {
    const int q = q0 + lane;
    const bool ok = resample && q < n_out;
    i0v = ok ? a.tab_i0[q] : 0x7fffffff;  // end marker: never inside a tile's range
    wgt = ok ? a.tab_w[q] : 0.0;
  };
  int qw = 0, i0_l, i0_n;
  double w_l, w_n;
  load_window(0, i0_l, w_l);
  load_window(64, i0_n, w_n);
  auto step = [&](int t0, real (&buf)[SPL]) __attribute__((always_inline)) {
    wv.tile_core(t0, buf, [&]() __attribute__((always_inline)) {
      if (t0 + 2 * TILE < T) wv.load(t0 + 2 * TILE, buf);  // the set is free again: request the tile after next
    });
    const int avail = t0 + TILE;  // P[j] is in the ring for j < avail (zeros past T: P[j] = P[T] there)
    if (!resample) {
      int end = avail - hi - 1;
      if (end > T) end = T;
      if (end - emitted == TILE && emitted >= wv.lo) {
        // steady state: a full tile of outputs, no clamping
        double sd[SPL];
        wv.window_sums_full_tile(emitted, sd);
        real* __restrict__ op = o + emitted + lane;
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
          if constexpr (sizeof(real) == 4) {
            // fp32 output: the fp64 window sum is rounded to float once, the root is the bare v_sqrt_f32 (1 ulp; the
            // correctly rounded one of the ragged tiles below pays ~18 instructions for the last half ulp)
            const float y = __builtin_amdgcn_sqrtf(fmaxf((float)sd[u], 0.f) * inv_wf);
            op[64 * u] = y;
            vmaxf = fmaxf(vmaxf, y);
          } else {
            const double y = sqrt((sd[u] > 0.0 ? sd[u] : 0.0) * inv_w);
            op[64 * u] = (real)y;
            vmax = fmax(vmax, y);
          }
        }
      } else {
        for (int i = emitted + lane; i < end; i += 64) {
          if constexpr (sizeof(real) == 4) {
            const float y = __builtin_sqrtf(fmaxf((float)wv.window_sum(i), 0.f) * inv_wf);
            o[i] = y;
            vmaxf = fmaxf(vmaxf, y);
          } else {
            const double y = value(i);
            o[i] = (real)y;
            vmax = fmax(vmax, y);
          }
        }
      }
      if (end > emitted) emitted = end;
    } else {
      // outputs are consumed in order: a window of 64 consecutive ones sits in registers (left knot, weight); a lane
      // fires when its left knot falls into this tile's range [emitted, end) of complete windows, and once the last
      // lane's knot is behind `end` the next window (requested when this one was installed) takes over
      int end = avail - hi - 2;
      if (end > T - 1) end = T - 1;
      if (end > emitted) {
        while (true) {
          if (i0_l >= emitted && i0_l < end) {
            const double y = env_interp(wv, i0_l, w_l, inv_w);
            o[qw + lane] = (real)y;
            vmax = fmax(vmax, fabs(y));
          }
          if (__builtin_amdgcn_readlane(i0_l, 63) >= end) break;  // later tiles' outputs (or the end marker) remain
          qw += 64;
          i0_l = i0_n;
          w_l = w_n;
          load_window(qw + 64, i0_n, w_n);
        }
        emitted = end;
      }
    }
    env_wave_sync();  // the next tile overwrites ring entries only after these reads
  };
  for (int t0 = 0; emitted < end_all; t0 += 2 * TILE) {
    step(t0, pfa);
    if (emitted >= end_all) break;
    step(t0 + TILE, pfb);
  }
  if (a.normalize) {
    vmax = fmax(vmax, (double)vmaxf);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the outputs were written by other lanes of this wave
    const float vmf = (float)vmax;
    int q0 = 0;
    if ((reinterpret_cast<unsigned long long>(o) & 15ull) == 0) {  // 16-byte accesses, four in flight per lane
      Vec* __restrict__ ov = reinterpret_cast<Vec*>(o);
      const int nv = n_out / V;
      int qv = 0;
      for (; qv + 4 * 64 <= nv; qv += 4 * 64) {
        Vec t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = ov[qv + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int e = 0; e < V; ++e) t[u].v[e] = env_scaled<real>(t[u].v[e], vmax, vmf);
          ov[qv + u * 64 + lane] = t[u];
        }
      }
      for (int q = qv + lane; q < nv; q += 64) {
        Vec t = ov[q];
#pragma unroll
        for (int e = 0; e < V; ++e) t.v[e] = env_scaled<real>(t.v[e], vmax, vmf);
        ov[q] = t;
      }
      q0 = nv * V;
    }
    for (int q = q0 + lane; q < n_out; q += 64) o[q] = env_scaled<real>(o[q], vmax, vmf);
  }
}

// =================================================================================================
// Workgroup-per-series version for the full-length output (no time normalisation) of series of up to ~20 K samples.
// (Time-normalised outputs stay with emg_wave_kernel: measured, the interpolation step per 256-sample tile costs this
// kernel more than the second read of the samples costs that one: 0.76 vs 0.56 ms.)
// emg_wave_kernel is bound by memory traffic there (measured 5.1 TB/s): raw read twice (mean, tiles), output written,
// read back and written again once the channel maximum is known = 5 sizeof(real) per sample.  Here the NW waves of a
// workgroup split the series into NW segments (each re-runs the prefix over the W samples before its segment) and
// keep BOTH their samples and their outputs in registers: all samples are requested at once (one memory latency per
// series), the mean comes from the same registers, and the envelope goes to memory once, already divided by the
// maximum: 2 sizeof(real) per sample, the algorithmic minimum.  LDS holds only the rings, so (float) two workgroups
// share a CU and one's memory wait / barriers overlap the other's arithmetic.
// =================================================================================================
// Instances (NW waves x MAXT tiles of 256 samples in registers): float 8 x 12 (two workgroups per CU, <= 128 VGPRs);
// double 8 x 6 for series of up to 8192 samples, and 16 x 7 -- one 1024-thread workgroup per CU, no overlap between
// workgroups, but 2 instead of 5 sizeof(double) of traffic per sample -- up to 20 480.
constexpr int ENV_WG_SPL = 4;

template <typename real, int NW, int MAXT>
__global__ void __launch_bounds__(64 * NW) emg_wg_kernel(EnvArgs a, int ring) {
  extern __shared__ __attribute__((aligned(16))) unsigned char env_smem[];
  constexpr int SPL = ENV_WG_SPL;
  using Wv = EnvWave<real, SPL>;
  constexpr int TILE = Wv::TILE;
  const int ring_entries = ring + ring / 8;
  double* rings = reinterpret_cast<double*>(env_smem);  // [NW][ring_entries]
  double* scratch = rings + (size_t)NW * ring_entries;  // [NW]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = blockIdx.x, b = blockIdx.y;
  const long long cidx = (long long)b * a.m + ch;
  const real* __restrict__ x = static_cast<const real*>(a.raw) + (long long)b * a.bstride + (long long)ch * a.ld;
  const int T = a.T, W = a.window;
  real* __restrict__ o = static_cast<real*>(a.out) + cidx * (long long)T;

  // segment of this wave: whole tiles; the walk starts far enough before it for the first window and ends far enough
  // behind it for the last one (at most MAXT tiles: the launcher checks)
  const int seg_len = ((T + NW - 1) / NW + TILE - 1) / TILE * TILE;
  const int seg_b = wave * seg_len;
  const int seg_e = (seg_b + seg_len < T) ? seg_b + seg_len : T;
  const int seg_end = seg_e;
  const int lag = (W - 1) / 2 + 1;  // output i needs the prefix up to i + lag
  int first = seg_b - (W + TILE - 1) / TILE * TILE;
  if (first < 0) first = 0;
  Wv wv;
  wv.init(x, rings + (size_t)wave * ring_entries, ring, T, W, 0.0, lane, first);
  int nt = (seg_b < seg_end) ? (seg_end + lag - first + TILE - 1) / TILE : 0;
  if (nt > MAXT) nt = MAXT;
  real xs[MAXT][SPL];  // samples, then (tile by tile, as they are consumed) outputs
#pragma unroll
  for (int k = 0; k < MAXT; ++k)
    if (k < nt) wv.load(first + k * TILE, xs[k]);
  if (a.zero_center) {
    double acc = 0.0;  // the wave's own samples (zeros past the end of the series), fixed order
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
      const int t0 = first + k * TILE;
      if (k < nt && t0 >= seg_b && t0 < seg_e) {
#pragma unroll
        for (int c = 0; c < SPL; ++c) acc += (double)xs[k][c];
      }
    }
    wv.set_mean(block_sum(acc, scratch) / (double)T);
  }
  const double inv_w = 1.0 / (double)W;
  const float inv_wf = (float)inv_w;
  double vmax = 0.0;
  float vmaxf = 0.f;
  int eb[MAXT], ee[MAXT];  // outputs eb[k] .. ee[k] were produced after tile k: output eb[k] + lane + 64 u sits in xs[k][u]
  int emitted = seg_b;
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    eb[k] = ee[k] = 0;
    if (k < nt) {
      const int t0 = first + k * TILE;
      wv.tile_core(t0, xs[k], []() {});
      int end = t0 + TILE - lag;
      if (end > seg_end) end = seg_end;
      if (end > emitted) {
        double sd[SPL];
        if (end - emitted == TILE && emitted >= wv.lo) {
          wv.window_sums_full_tile(emitted, sd);
        } else {
#pragma unroll
          for (int u = 0; u < SPL; ++u) {
            const int i = emitted + lane + 64 * u;
            sd[u] = (i < end) ? wv.window_sum(i) : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
          // fp32 output: the fp64 window sum is rounded to float once and the root is v_sqrt_f32 (1 ulp)
          if constexpr (sizeof(real) == 4) {
            const float y = __builtin_amdgcn_sqrtf(fmaxf((float)sd[u], 0.f) * inv_wf);
            xs[k][u] = y;
            vmaxf = fmaxf(vmaxf, y);  // lanes past `end` hold 0
          } else {
            const double y = sqrt((sd[u] > 0.0 ? sd[u] : 0.0) * inv_w);
            xs[k][u] = (real)y;
            vmax = fmax(vmax, y);
          }
        }
        eb[k] = emitted;
        ee[k] = end;
        emitted = end;
      }
      env_wave_sync();  // the next tile overwrites ring entries only after these reads
    }
  }
  const bool norm = a.normalize != 0;
  if (norm) vmax = block_max(fmax(vmax, (double)vmaxf), scratch);
  const float vmf = (float)vmax;
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
      const int i = eb[k] + lane + 64 * u;
      if (i < ee[k]) o[i] = norm ? env_scaled<real>(xs[k][u], vmax, vmf) : xs[k][u];
    }
  }
}

}  // namespace hipnmf
