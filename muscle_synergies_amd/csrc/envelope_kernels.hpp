// envelope_kernels.hpp -- batched EMG envelope preprocessing on gfx950.
//
// Reference semantics (src/muscle_synergies/analysis.py): zero_center :230-249, rms :435-507
// (np.sqrt(np.convolve(x**2, ones(W)/W, "same"))), time_normalize :551-594 (scipy interp1d, linear, on
// linspace(0,1,T) -> linspace(0,1,n_out)), normalize :510-525 (divide by max |.| per column).
//
// One workgroup per (channel, recording).  The sliding-window mean is taken from an fp64 prefix sum of the
// squared (centred) samples, so every global access is coalesced and the window sum costs two loads per
// output regardless of W; sums are accumulated in fp64 for both fp32 and fp64 I/O.
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
};

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
        int i0 = (int)floor(xn * (double)(a.T - 1));
        if (i0 > a.T - 2) i0 = a.T - 2;
        if (i0 < 0) i0 = 0;
        while (i0 > 0 && knot(i0) >= xn) --i0;           // searchsorted(side="left") - 1, clipped to >= 0
        while (i0 < a.T - 2 && knot(i0 + 1) < xn) ++i0;
        const double x0 = knot(i0), x1 = knot(i0 + 1);
        const double y0 = value(i0), y1 = value(i0 + 1);
        y = (y1 - y0) / (x1 - x0) * (xn - x0) + y0;
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
          int i0 = (int)floor(xn * (double)(T - 1));
          if (i0 > T - 2) i0 = T - 2;
          if (i0 < 0) i0 = 0;
          while (i0 > 0 && knot(i0) >= xn) --i0;  // searchsorted(side="left") - 1, clipped to >= 0
          while (i0 < T - 2 && knot(i0 + 1) < xn) ++i0;
          if (i0 < t0 || i0 >= t0 + nout_t) continue;
          const double x0 = knot(i0), x1 = knot(i0 + 1);
          const double y0 = value(i0), y1 = value(i0 + 1);
          y = (y1 - y0) / (x1 - x0) * (xn - x0) + y0;
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

}  // namespace hipnmf
