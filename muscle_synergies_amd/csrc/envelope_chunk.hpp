// envelope_chunk.hpp -- emg_chunk_kernel: the EMG envelope of a series that fits the LDS of one workgroup (round 4).
//
// Same stage of the reference as envelope_kernels.hpp (src/muscle_synergies/analysis.py: zero_center :230-249, rms :435-507,
// time_normalize :551-594, normalize :510-525), other structure.  emg_wg_kernel walks a series tile by tile: every 256-sample tile
// pays a DPP scan and two LDS round trips, strictly in sequence (7.5 us of latency per series against 3.7 us of memory time).
// Here one 256-thread workgroup owns a series and thread t owns the C CONSECUTIVE samples [C t, C t + C):
//   1. the series goes HBM -> registers -> LDS once (dword loads through a buffer resource: no alignment or length condition,
//      out-of-range lanes read 0), the mean comes from the same registers (fp64, fixed order);
//   2. every thread slides the window over its own chunk: R(p) = sum of the W squared centred samples ending at p,
//      R(p) = R(p - 1) + sq[p] - sq[p - W], kept RELATIVE to the thread's first position in C fp64 registers; both streams are
//      LDS reads at the thread's own addresses (C odd: the stride of the threads is conflict-free), samples outside [0, T) are
//      zeros as np.convolve(.., "same") pads them (the high word of the centred value is cleared: its square underflows to 0);
//   3. ONE scan over the 256 threads' totals (DPP inside a wave, four wave totals through LDS) gives every thread R before its
//      chunk -- a thread's total is exactly R(end) - R(start), whatever the window;
//   4. outputs y[i] = sqrt(R(i + hi) / W), hi = (W - 1) / 2, go back to LDS over the dead samples, and from there either to HBM
//      in coalesced order (divided by the channel maximum on the way) or through the time normalisation's table.
// HBM sees every sample once in and once out; the per-series latency chain is one load round trip, one scan, three barriers.
// C is odd (81 / 41 / 17: series of up to 256 C - hi samples) so that the natural LDS layout needs no padding; a front pad of W
// entries keeps the addresses of the leaving stream non-negative.  Two workgroups share a CU while 2 x (W + T + C) x
// sizeof(real) fit its 160 KB (float: T = 20 000 with W <= 279).  Sums are fp64 whatever the I/O type; the summation order
// differs from the other kernels', results agree to ~1e-13 relative (fp64) / an ulp of float (fp32).
#pragma once
#include "envelope_kernels.hpp"
#include "nmf_wide.hpp"  // rsrc_t, make_rsrc, buf_load, buf_store (nmf_kernels.hpp), wide_lds_write

namespace hipnmf {

constexpr int CHUNK_RED = 24;  // doubles of reduction scratch behind the samples (3 per wave, up to 8 waves)
#ifndef HIPNMF_CHUNK_LD_AUX
#define HIPNMF_CHUNK_LD_AUX 0
#endif
#ifndef HIPNMF_CHUNK_ST_AUX
#define HIPNMF_CHUNK_ST_AUX 0
#endif

template <typename real>
__device__ __forceinline__ void chunk_store(rsrc_t r, unsigned voff, unsigned soff, real v) {
  if constexpr (sizeof(real) == 4) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, (HIPNMF_CHUNK_ST_AUX));
  } else {
    using u32x2 = unsigned int __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, (HIPNMF_CHUNK_ST_AUX));
  }
}

// (double)x - mean, or a value whose square is exactly 0 when the position lies outside the series
template <typename real>
__device__ __forceinline__ double chunk_centred(real x, double mean, bool inside) {
  const double d = (double)x - mean;
  const unsigned long long u = __builtin_bit_cast(unsigned long long, d);
  const unsigned hi = inside ? (unsigned)(u >> 32) : 0u;  // exponent cleared: at most a subnormal is left, d * d == 0
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned)u);
}

// 16 bytes to / from a buffer resource (VEC instances: rows and lengths that are whole 16-byte pieces)
template <typename real>
__device__ __forceinline__ void chunk_store_vec(rsrc_t r, unsigned voff, unsigned soff, const real (&v)[16 / sizeof(real)]) {
  using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
  u32x4 u;
  __builtin_memcpy(&u, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, (HIPNMF_CHUNK_ST_AUX));
}

// VEC: the series (and, full length, the output) are moved in 16-byte pieces -- a quarter of the memory instructions; the host
// selects it when every row is 16-byte aligned and the lengths are whole pieces (the front pad is rounded up to a piece then).
template <typename real, int C, int NT, bool VEC = false>
__global__ void __launch_bounds__(NT, (NT >= 256 ? NT / 128 : 4)) emg_chunk_kernel(EnvArgs a, int nact /* threads that own a chunk */) {
  static_assert(C % 2 == 1, "odd chunk length: conflict-free LDS stride");
  static_assert(NT == 64 || NT == 256 || NT == 512, "one (short series: no workgroup-wide step is left), four or eight waves");
  constexpr int CHUNK_THREADS = NT, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char env_smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ch = blockIdx.x, b = blockIdx.y;
  const long long cidx = (long long)b * a.m + ch;
  const real* __restrict__ x = static_cast<const real*>(a.raw) + (long long)b * a.bstride + (long long)ch * a.ld;
  const int T = a.T, W = a.window, hi = (W - 1) / 2;
  const int n_out = a.n_out > 0 ? a.n_out : T;
  const bool resample = a.n_out > 0 && a.n_out != T;
  real* __restrict__ o = static_cast<real*>(a.out) + cidx * (long long)n_out;
  constexpr int V = 16 / (int)sizeof(real);
  const int NS = nact * C;                            // positions p in [0, NS): p < T samples, then zeros
  const int WP = VEC ? (W + V - 1) / V * V : W;       // front pad (VEC: whole pieces, so that position 0 is 16-byte aligned)
  const int NSA = VEC ? (NS + V - 1) / V * V : NS;
  real* __restrict__ xs = reinterpret_cast<real*>(env_smem) + WP;  // xs[p], p in [-WP, NSA)
  double* __restrict__ red = reinterpret_cast<double*>(env_smem + (((size_t)(WP + NSA) * sizeof(real) + 15) & ~(size_t)15));  // [3 NW]
  const rsrc_t xr = make_rsrc(x, (unsigned)((long long)T * (long long)sizeof(real)));
  const rsrc_t orr = make_rsrc(o, (unsigned)((long long)n_out * (long long)sizeof(real)));

  // the first window of table entries of the time normalisation: requested before anything else
  int i0_q = 0;
  double w_q = 0.0;
  if (resample && t < n_out) {
    i0_q = a.tab_i0[t];
    w_q = a.tab_w[t];
  }

  // ---- 1. HBM -> registers -> LDS; mean --------------------------------------------------------------------------------------
  double mean = 0.0;
  {
    double acc = 0.0;
    if constexpr (VEC) {
      constexpr int CV = (C + V - 1) / V;  // pieces per thread: piece t + NT k4 holds positions V (t + NT k4) .. + V - 1
      real v[CV][V];
#pragma unroll
      for (int k4 = 0; k4 < CV; ++k4)
        buf_load<real, V, (HIPNMF_CHUNK_LD_AUX)>(xr, (unsigned)t * 16u, (unsigned)(k4 * CHUNK_THREADS) * 16u, v[k4]);
#pragma unroll
      for (int k4 = 0; k4 < CV; ++k4) {
        const int p = (t + k4 * CHUNK_THREADS) * V;
        wide_lds_write<real, V>(xs + (p < NSA ? p : -V), v[k4]);  // (the front pad takes what lies behind the chunks)
#pragma unroll
        for (int e = 0; e < V; ++e) acc += (double)v[k4][e];  // (zeros past the end of the series)
      }
    } else {
      real v[C];
      const unsigned voff = (unsigned)t * (unsigned)sizeof(real);
#pragma unroll
      for (int k = 0; k < C; ++k) {
        real r1[1];
        buf_load<real, 1, (HIPNMF_CHUNK_LD_AUX)>(xr, voff, (unsigned)(k * CHUNK_THREADS) * (unsigned)sizeof(real), r1);
        v[k] = r1[0];
      }
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const int p = t + k * CHUNK_THREADS;
        xs[p < NS ? p : -1] = v[k];  // (xs[-1]: the front pad, W >= 1 -- a branch per store would serialise them)
        acc += (double)v[k];         // (zeros past the end of the series)
      }
    }
    if (a.zero_center) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) red[wave] = acc;
    }
    __syncthreads();
    if (a.zero_center) {
      double tot = red[0];
      if constexpr (NW >= 4) tot = (red[0] + red[1]) + (red[2] + red[3]);
      if constexpr (NW == 8) tot += (red[4] + red[5]) + (red[6] + red[7]);
      mean = tot / (double)T;
    }
  }

  // ---- 2. the window slides over the thread's chunk ------------------------------------------------------------------------
  const int p0 = C * t;
  const bool active = t < nact;
  double rel[C];
  double run = 0.0;
  if (active) {
    const real* __restrict__ xin = xs + p0;
    const real* __restrict__ xout = xs + p0 - W;
    const int n_in = T - p0;   // entering sample p0 + n is inside the series while n < n_in
    const int n_out0 = W - p0;  // leaving sample p0 + n - W while n >= n_out0 (it is below T for every active thread)
#pragma unroll
    for (int n = 0; n < C; ++n) {
      const double di = chunk_centred<real>(xin[n], mean, n < n_in);
      const double dl = chunk_centred<real>(xout[n], mean, n >= n_out0);
      run = __builtin_fma(di, di, run);
      run = __builtin_fma(-dl, dl, run);
      rel[n] = run;
    }
  }

  // ---- 3. R before the chunk: exclusive scan of the threads' totals ----------------------------------------------------------
  const double inc = env_wave_inclusive_scan(run);
  if (lane == 63) red[NW + wave] = inc;
  __syncthreads();  // (also: every read of the samples is done)
  double base = inc - run;
  for (int w2 = 0; w2 < wave; ++w2) base += red[NW + w2];

  // ---- 4. outputs over the dead samples: slot p holds output i = p - hi ------------------------------------------------------
  // (time normalisation: only the two neighbours of every output get their root, so the slots hold the window sums)
  const double inv_w = 1.0 / (double)W;  // np.convolve(x^2, ones(W) / W): products by 1/W, no division
  const float inv_wf = (float)inv_w;
  auto root = [&](real s) -> real {
    // fp32 output: the fp64 window sum is rounded to float once, the root is v_sqrt_f32 (1 ulp) -- as in emg_wg_kernel
    if constexpr (sizeof(real) == 4) return __builtin_amdgcn_sqrtf(fmaxf(s, 0.f) * inv_wf);
    else return sqrt((s > 0.0 ? s : 0.0) * inv_w);
  };
  real vmx = (real)0;
  if (active) {
    real* __restrict__ ys = xs + p0;
    const int n_lo = hi - p0;      // output index p0 + n - hi is >= 0 while n >= n_lo
    const int n_hi = T + hi - p0;  // ... and < T while n < n_hi
    if (resample) {
#pragma unroll
      for (int n = 0; n < C; ++n) ys[n] = (real)(base + rel[n]);
    } else {
#pragma unroll
      for (int n = 0; n < C; ++n) {
        const real y = root((real)(base + rel[n]));
        ys[n] = y;
        vmx = fmax(vmx, (n >= n_lo && n < n_hi) ? y : (real)0);
      }
    }
  }
  const real* __restrict__ yo = xs + hi;  // yo[i] = output i

  const bool norm = a.normalize != 0;
  if (!resample) {
    // ---- 5a. full length: LDS -> HBM in coalesced order, divided by the channel maximum ------------------------------------
    double vmax = (double)vmx;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) red[2 * NW + wave] = vmax;
    __syncthreads();
    vmax = red[2 * NW];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) vmax = fmax(vmax, red[2 * NW + w2]);
    const float vmf = (float)vmax;
    // y / vmax as the reference rounds it (env_scaled) costs ~10 instructions; with a reciprocal and one correction step
    // (q = y r, q += (y - q v) r) the quotient is the correctly rounded one up to rare off-by-an-ulp cases; that path is taken
    // when vmax is a normal number far from the ends of the range.  Batches of CB: the LDS reads of a batch are all requested
    // before the first is used (one read per trip, with a trip's branches around it, costs a full LDS latency per output).
    constexpr int CB = 8;
    const int kmax = (T + CHUNK_THREADS - 1) / CHUNK_THREADS;
    const unsigned voff = (unsigned)t * (unsigned)sizeof(real);
    auto copy_out = [&](auto scale) __attribute__((always_inline)) {
      if constexpr (VEC) {
        const int kmax4 = (T / V + CHUNK_THREADS - 1) / CHUNK_THREADS;
        constexpr int CB4 = CB / V > 0 ? CB / V : 1;
        for (int k0 = 0; k0 < kmax4; k0 += CB4) {
          real y[CB4][V];
          const real* __restrict__ src = yo + (t + k0 * CHUNK_THREADS) * V;
#pragma unroll
          for (int u = 0; u < CB4; ++u)
#pragma unroll
            for (int e = 0; e < V; ++e) y[u][e] = src[u * CHUNK_THREADS * V + e];
#pragma unroll
          for (int u = 0; u < CB4; ++u) {
#pragma unroll
            for (int e = 0; e < V; ++e) y[u][e] = scale(y[u][e]);
            chunk_store_vec<real>(orr, (unsigned)t * 16u, (unsigned)((k0 + u) * CHUNK_THREADS) * 16u, y[u]);
          }
        }
      } else {
        for (int k0 = 0; k0 < kmax; k0 += CB) {
          real y[CB];
          const real* __restrict__ src = yo + t + k0 * CHUNK_THREADS;  // (reads past NS: LDS reads cannot fault, the stores are dropped)
#pragma unroll
          for (int u = 0; u < CB; ++u) y[u] = src[u * CHUNK_THREADS];
#pragma unroll
          for (int u = 0; u < CB; ++u)
            chunk_store<real>(orr, voff, (unsigned)((k0 + u) * CHUNK_THREADS) * (unsigned)sizeof(real), scale(y[u]));
        }
      }
    };
    bool done = false;
    if constexpr (sizeof(real) == 4) {
      if (norm && vmf > 1e-30f && vmf < 1e30f) {
        const float rcp = __builtin_amdgcn_rcpf(vmf);
        copy_out([&](float y) {
          const float q = y * rcp;
          return __builtin_fmaf(__builtin_fmaf(-q, vmf, y), rcp, q);
        });
        done = true;
      }
    }
    if (!done) {
      if (norm) copy_out([&](real y) { return env_scaled<real>(y, vmax, vmf); });
      else copy_out([&](real y) { return y; });
    }
  } else {
    // ---- 5b. time normalisation (table of env_resample_table_kernel): output q from the outputs i0 and i0 + 1 -----------------
    __syncthreads();
    auto interp = [&](int i0, double w) -> double { return env_lerp((double)root(yo[i0]), (double)root(yo[i0 + 1]), w); };
    double vm = 0.0;
    for (int q = t; q < n_out; q += CHUNK_THREADS) {
      const int i0 = q == t ? i0_q : a.tab_i0[q];
      const double w = q == t ? w_q : a.tab_w[q];
      vm = fmax(vm, fabs(interp(i0, w)));
    }
    if (norm) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) vm = fmax(vm, __shfl_xor(vm, off, 64));
      if (lane == 0) red[2 * NW + wave] = vm;
      __syncthreads();
      vm = red[2 * NW];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) vm = fmax(vm, red[2 * NW + w2]);
    }
    const float vmf = (float)vm;
    for (int q = t; q < n_out; q += CHUNK_THREADS) {
      const int i0 = q == t ? i0_q : a.tab_i0[q];
      const double w = q == t ? w_q : a.tab_w[q];
      const real y = (real)interp(i0, w);
      o[q] = norm ? env_scaled<real>(y, vm, vmf) : y;
    }
  }
}

}  // namespace hipnmf
