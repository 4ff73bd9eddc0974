// sosfilt_scan.hpp -- the time-parallel mode of the IIR filter stage (hipnmf_sosfilt_params.mode = HIPNMF_SOSFILT_SCAN).
//
// Same stage of the reference as sosfilt_kernels.hpp (digital_filter / linear_envelope, src/muscle_synergies/analysis.py:252-432 ->
// scipy.signal.sosfiltfilt / sosfilt), other trade: sosfilt2_kernel reproduces scipy's sequential recurrence bit for bit and is
// bound by its 41 200-step dependent fp64 chain per series (0.16 of HBM, round 3); this kernel cuts every series into 256 chunks,
// one per thread of a workgroup, and runs them at once.  A cascade of second-order sections is a linear system
//     s[n] = A s[n-1] + B x[n],   y[n] = c s[n-1] + d x[n]          (s: the 2 * n_sections direct-form-II-transposed states)
// so for a chunk of C samples entered with state s0:  y = y0 + G s0  and  s_end = F + M s0, with y0 / F the chunk's response from
// a ZERO state, G[n] (C x 2NS) the output's response to a unit state and M = A^C.  Per direction:
//   1. every thread filters its chunk from a zero state (registers; the only dependent chain left is C = 79 samples long);
//   2. the chunk-end states are combined by a scan over the 256 threads with the constant matrices M^(2^j): Kogge-Stone inside a
//      wave (shuffles), the four waves' totals through LDS (scan_states);
//   3. y += G s_start, 2NS independent FMAs per sample with G read through the scalar cache.
// The whole (odd-extended) series lives in registers between the forward and the backward pass: HBM sees the samples once in and
// once out.  sosfiltfilt's backward pass starts at the last extended sample with the state zi * y[L-1]; the chunks are aligned to
// the front, so the tail of the last chunk is padded with that very constant and the backward recursion starts at the padded end
// in the filter's steady state for it -- which it then keeps (zi is by definition the fixed point of a constant input).
// Requirements (the host checks them and otherwise runs the sequential kernel): every series and its output 16-byte aligned with
// n_samples a multiple of 4 (fp32) / 2 (fp64) -- all staging traffic is 16-byte vectors -- and at most 20 480 extended samples.
// All arithmetic is fp64 with fused multiply-adds; the result agrees with scipy to rounding (1e-12 relative for the reference's
// 6 Hz low-pass at 2 kHz; the conditioning of the filter sets the constant), not bit for bit: tests/test_filters.py, <= 1e-10.
#pragma once
#include "nmf_kernels.hpp"  // rsrc_t, make_rsrc, buf_load, buf_store (sosfilt_chunk_kernel)
#include "sosfilt_kernels.hpp"

namespace hipnmf {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_CMAX = 80;  // samples per thread of the large instance: series of up to 256 * 80 = 20 480 extended samples
constexpr int SCAN_G_CAP = 128 * 2 * SOS_MAX_SECTIONS;            // doubles reserved for G in the table buffer
constexpr int SCAN_TAB_DOUBLES = SCAN_G_CAP + 9 * 4 * SOS_MAX_SECTIONS * SOS_MAX_SECTIONS;  // + M^(2^j), j = 0..8 (M^256: a whole block of 256 chunks)
constexpr int SCAN_STAGE_BYTES = 48 * 1024;
constexpr int SCAN_LOADS = 12;  // 16-byte loads in flight per thread while staging (a 48 KiB piece in one round)

// section coefficients {b0, b1, b2, a1, a2}; sections beyond `ns` pass their input through
template <int NSP>
__device__ __forceinline__ void scan_coeffs(const SosArgs& a, int ns, double (&c)[NSP][5]) {
#pragma unroll
  for (int s = 0; s < NSP; ++s) {
    const bool on = s < ns;
    c[s][0] = on ? a.sos[s][0] : 1.0;
    c[s][1] = on ? a.sos[s][1] : 0.0;
    c[s][2] = on ? a.sos[s][2] : 0.0;
    c[s][3] = on ? a.sos[s][4] : 0.0;
    c[s][4] = on ? a.sos[s][5] : 0.0;
  }
}

// one sample through the cascade (fused multiply-adds; two dependent operations per section on the state's critical path)
template <int NSP>
__device__ __forceinline__ double scan_step(double xc, double (&z)[NSP][2], const double (&c)[NSP][5]) {
#pragma unroll
  for (int s = 0; s < NSP; ++s) {
    const double u = __builtin_fma(c[s][1], xc, z[s][1]);  // off the critical path: does not wait for xn
    const double xn = __builtin_fma(c[s][0], xc, z[s][0]);
    z[s][0] = __builtin_fma(-c[s][3], xn, u);
    z[s][1] = __builtin_fma(c[s][2], xc, -c[s][4] * xn);
    xc = xn;
  }
  return xc;
}

// G[n][j] (n < C_run): output n steps after a unit state j, no input; M = A^C_run and its squarings M^2 .. M^128.
// tab: [SCAN_G_CAP] G row-major [n][NST], then [9][NST][NST] (M^(2^j), j = 0..8).  One workgroup of 256 threads.
template <int NSP>
__global__ void __launch_bounds__(256) sos_scan_tables_kernel(SosArgs a, int ns, int C_run, double* __restrict__ tab) {
  constexpr int NST = 2 * NSP;
  __shared__ double Ma[NST * NST], Mb[NST * NST];
  const int j = threadIdx.x;
  double c[NSP][5];
  scan_coeffs<NSP>(a, ns, c);
  if (j < NST) {
    double z[NSP][2];
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      z[s][0] = (j == 2 * s) ? 1.0 : 0.0;
      z[s][1] = (j == 2 * s + 1) ? 1.0 : 0.0;
    }
    for (int n = 0; n < C_run; ++n) tab[n * NST + j] = scan_step<NSP>(0.0, z, c);
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      Ma[(2 * s) * NST + j] = z[s][0];
      Ma[(2 * s + 1) * NST + j] = z[s][1];
    }
  }
  __syncthreads();
  double* Mp = tab + SCAN_G_CAP;
  for (int p = 0; p < 9; ++p) {
    if (j < NST * NST) Mp[p * NST * NST + j] = Ma[j];
    if (j < NST * NST) {
      const int r = j / NST, q = j % NST;
      double acc = 0.0;
      for (int k = 0; k < NST; ++k) acc = __builtin_fma(Ma[r * NST + k], Ma[k * NST + q], acc);
      Mb[j] = acc;
    }
    __syncthreads();
    if (j < NST * NST) Ma[j] = Mb[j];
    __syncthreads();
  }
}

// E += Mat * o  (Mat: NST x NST row-major, uniform -> scalar loads)
template <int NST>
__device__ __forceinline__ void scan_matvec_acc(const double* __restrict__ Mat, const double (&o)[NST], double (&E)[NST]) {
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    double acc = E[i];
#pragma unroll
    for (int k = 0; k < NST; ++k) acc = __builtin_fma(Mat[i * NST + k], o[k], acc);
    E[i] = acc;
  }
}

// P = Mat * o  (Mat uniform)
template <int NST>
__device__ __forceinline__ void scan_matvec(const double* __restrict__ Mat, const double (&o)[NST], double (&P)[NST]) {
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    double acc = Mat[i * NST] * o[0];
#pragma unroll
    for (int k = 1; k < NST; ++k) acc = __builtin_fma(Mat[i * NST + k], o[k], acc);
    P[i] = acc;
  }
}

// Chunk-end states F of the 256 chunks -> the state every chunk is ENTERED with.  Logical chunk order q = t (forward) or 255 - t
// (backward); chunk 0 is entered with s_init.  Three phases: (A) inclusive Kogge-Stone scan INSIDE each wave (64 chunks,
// shuffles, constant matrices M^(2^j)); (B) the state each wave is entered with from the waves' totals (Horner with M^64, at most
// three wave-uniform products); (C) lane l of a wave adds M^(l+1) times that state, the power built from the binary digits of
// l + 1.  xch: LDS [256][NST].
template <int NST, int NT = SCAN_THREADS>
__device__ __forceinline__ void scan_states(double (&E)[NST], const double (&s_init)[NST], bool reverse, const double* __restrict__ Mp,
                                            double* __restrict__ xch, double (&s_start)[NST]) {
  const int t = threadIdx.x, lane = t & 63;
  const int q = reverse ? NT - 1 - t : t;
  const int lq = q & 63, wq = q >> 6;
  if (q == 0) scan_matvec_acc<NST>(Mp, s_init, E);  // E_0 = F_0 + M s_init
#pragma unroll
  for (int j = 0; j < 6; ++j) {  // (A)
    const int d = 1 << j;
    const int src = reverse ? lane + d : lane - d;
    double o[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) o[i] = __shfl(E[i], src & 63, 64);
    if (lq >= d) scan_matvec_acc<NST>(Mp + j * NST * NST, o, E);
  }
  __syncthreads();
  if (lq == 63) {
#pragma unroll
    for (int i = 0; i < NST; ++i) xch[wq * NST + i] = E[i];  // total of logical wave wq
  }
  __syncthreads();
  if (wq > 0) {  // (B) + (C); wave-uniform
    double K[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) K[i] = xch[i];
    for (int u = 1; u < wq; ++u) {  // K <- W_u + M^64 K
      double Wu[NST];
#pragma unroll
      for (int i = 0; i < NST; ++i) Wu[i] = xch[u * NST + i];
      scan_matvec_acc<NST>(Mp + 6 * NST * NST, K, Wu);
#pragma unroll
      for (int i = 0; i < NST; ++i) K[i] = Wu[i];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {  // K <- M^(lq + 1) K, digit by digit
      double P[NST];
      scan_matvec<NST>(Mp + j * NST * NST, K, P);
      const bool on = ((lq + 1) >> j) & 1;
#pragma unroll
      for (int i = 0; i < NST; ++i) K[i] = on ? P[i] : K[i];
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) E[i] += K[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NST; ++i) xch[q * NST + i] = E[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NST; ++i) s_start[i] = q > 0 ? xch[(q - 1) * NST + i] : s_init[i];
}

// mean of one series in fp64, fixed order of the sums, sixteen 16-byte loads in flight per thread (series_mean of the sequential
// kernels keeps two: its order is pinned by the bit-exact mode).  scratch: 4 doubles of LDS.
template <typename real>
__device__ __forceinline__ double scan_mean(const real* __restrict__ xr, int T, double* scratch) {
  constexpr int V = 16 / (int)sizeof(real);
  constexpr int NV = 16;  // 16-byte loads in flight per thread
  double acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.0;
  int i;
  if ((reinterpret_cast<unsigned long long>(xr) & 15ull) == 0) {
    struct alignas(16) Vec { real v[V]; };
    const Vec* __restrict__ xv = reinterpret_cast<const Vec*>(xr);
    const int nv = T / V;
    int q = threadIdx.x;
    for (; q + (NV - 1) * SCAN_THREADS < nv; q += NV * SCAN_THREADS) {
      Vec b[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) b[u] = xv[q + u * SCAN_THREADS];
#pragma unroll
      for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[u & 7] += (double)b[u].v[e];
    }
    for (; q + 3 * SCAN_THREADS < nv; q += 4 * SCAN_THREADS) {
      Vec b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = xv[q + u * SCAN_THREADS];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[u] += (double)b[u].v[e];
    }
    for (; q < nv; q += SCAN_THREADS) {
      const Vec b0 = xv[q];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[0] += (double)b0.v[e];
    }
    i = nv * V + threadIdx.x;
  } else {
    i = threadIdx.x;
    for (; i + 7 * SCAN_THREADS < T; i += 8 * SCAN_THREADS) {
      real b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = xr[i + u * SCAN_THREADS];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += (double)b[u];
    }
  }
  for (; i < T; i += SCAN_THREADS) acc[1] += (double)xr[i];
  double sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = sum;
  __syncthreads();
  return ((scratch[0] + scratch[1]) + (scratch[2] + scratch[3])) / (double)T;
}

// One workgroup per series; thread t owns the extended samples [t * CMAX, (t + 1) * CMAX) in registers (CMAX a multiple of 4).  Dynamic LDS: the staging buffer (SCAN_STAGE_BYTES, coalesced
// HBM <-> per-thread chunks) + [256][NST] + 8 + CMAX * NST doubles.
template <typename real, int NSP, int CMAX>
__global__ void __launch_bounds__(SCAN_THREADS, (CMAX > 32 ? (NSP > 4 ? 1 : 2) : (NSP > 4 ? 2 : 4)))
sosfilt_scan_kernel(SosArgs a, const double* __restrict__ tab, int ns) {
  constexpr int NST = 2 * NSP;
  constexpr int C_run = CMAX;  // samples per thread: a compile-time constant (straight-line loops, no group branches)
  extern __shared__ __attribute__((aligned(16))) unsigned char scan_smem[];
  real* __restrict__ stage = reinterpret_cast<real*>(scan_smem);
  double* __restrict__ xch = reinterpret_cast<double*>(scan_smem + SCAN_STAGE_BYTES);
  double* __restrict__ misc = xch + SCAN_THREADS * NST;  // 8 doubles: mean scratch [0..3], broadcast value [4]
  double* __restrict__ Gl = misc + 8;                    // G [C_run][NST]: read by every thread at the same address (LDS broadcast)
  const int t = threadIdx.x;
  const int series = blockIdx.x;
  constexpr int V = 16 / (int)sizeof(real);  // samples per 16-byte vector
  struct alignas(16) Vec {
    real v[V];
  };
  // Positions p = delta + extended index: `delta` copies of the first extended sample in front make p and the raw index p - EO
  // (EO = edge + delta) congruent modulo V, so 16-byte vectors of x, of the staging rows and of y line up.  (The recursion then
  // starts `delta` samples early, in the steady state of that constant: the same argument as for the tail, see the header.)
  const int T = a.T, edge = a.edge;
  const int delta = (V - edge % V) % V, EO = edge + delta, L = T + 2 * edge + delta;
  const real* __restrict__ xr =
      static_cast<const real*>(a.x) + (long long)(series / a.m) * a.bstride + (long long)(series % a.m) * a.ld;
  real* __restrict__ yr = static_cast<real*>(a.y) + (long long)series * T;
  const double* __restrict__ G = tab;
  const double* __restrict__ Mp = tab + SCAN_G_CAP;

  double c[NSP][5];
  scan_coeffs<NSP>(a, ns, c);
  for (int i = t; i < C_run * NST; i += SCAN_THREADS) Gl[i] = G[i];  // (the first __syncthreads below orders it)

  // ---- HBM -> registers -> staging rows -> registers, 16-byte vectors.  Row t of the staging buffer = the chunk of thread t, SR
  // samples apart with SR * sizeof / 16 odd (the 16-byte chunk reads of the 16 lanes LDS serves at a time fall on distinct
  // banks); RP rows per piece.  Centring and rectification are applied between the registers and the staging buffer.
  constexpr int CV = C_run / V;  // vectors per row = vectors per thread (C_run is a multiple of 4)
  constexpr int SR = C_run + ((CV & 1) ? 0 : V);
  constexpr int RP = (SCAN_STAGE_BYTES / (SR * (int)sizeof(real))) < SCAN_THREADS ? (SCAN_STAGE_BYTES / (SR * (int)sizeof(real))) : SCAN_THREADS;
  constexpr bool ONE_ROUND = CV <= 20;  // the whole series fits the registers that are free before the chunks are: ONE round trip
  real mean = (real)0;
  real first = (real)0, last = (real)0;
  // value at position p outside the recording: the odd extension about the end samples (in the samples' precision, as scipy
  // builds it), its first value repeated in front, zeros past the end
  auto outside = [&](int p) -> real {
    int i = p - delta;
    i = i < 0 ? 0 : i;
    if (i >= T + 2 * edge) return (real)0;
    const int j = i - edge;
    const int jr = j < 0 ? -j : 2 * (T - 1) - j;
    return (real)2 * (j < 0 ? first : last) - sos_pre<real>(xr[jr], mean, a.rectify);
  };
  double v[CMAX];
#pragma unroll
  for (int n = 0; n < CMAX; ++n) v[n] = 0.0;
  auto read_chunk = [&](int r0) {
    const Vec* __restrict__ myrow = reinterpret_cast<const Vec*>(stage + (t - r0) * SR);
#pragma unroll
    for (int g = 0; g < CV; ++g) {
      const Vec x = myrow[g];
#pragma unroll
      for (int q = 0; q < V; ++q) v[g * V + q] = (double)x.v[q];
    }
  };
  if constexpr (ONE_ROUND) {
    // thread t loads the vectors w = t + 256 u of the whole series at once (coalesced); the mean comes from these registers
    Vec raw[CV];
#pragma unroll
    for (int u = 0; u < CV; ++u) {
      const int w = t + u * SCAN_THREADS, row = w / CV;
      const int j = row * C_run + (w - row * CV) * V - EO;
      raw[u] = *reinterpret_cast<const Vec*>(xr + ((j >= 0 && j < T) ? j : 0));  // (T is a multiple of V: inside or outside, never astride)
    }
    if (a.zero_center) {  // fp64 sum in a fixed order: the thread's vectors in order, wave tree, four waves
      double sum = 0.0;
#pragma unroll
      for (int u = 0; u < CV; ++u) {
        const int w = t + u * SCAN_THREADS, row = w / CV;
        const int j = row * C_run + (w - row * CV) * V - EO;
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < V; ++q) part += (double)raw[u].v[q];
        sum += (j >= 0 && j < T) ? part : 0.0;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
      if ((t & 63) == 0) misc[t >> 6] = sum;
      __syncthreads();
      mean = (real)(((misc[0] + misc[1]) + (misc[2] + misc[3])) / (double)T);
    }
    first = sos_pre<real>(xr[0], mean, a.rectify), last = sos_pre<real>(xr[T - 1], mean, a.rectify);
#pragma unroll
    for (int r0 = 0; r0 < SCAN_THREADS; r0 += RP) {
      const int rows = (RP < SCAN_THREADS - r0) ? RP : SCAN_THREADS - r0;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < CV; ++u) {
        const int w = t + u * SCAN_THREADS, row = w / CV;
        if (row >= r0 && row < r0 + rows) {
          const int col = (w - row * CV) * V, p = row * C_run + col, j = p - EO;
          Vec o;
          if (j >= 0 && j < T) {
#pragma unroll
            for (int q = 0; q < V; ++q) o.v[q] = sos_pre<real>(raw[u].v[q], mean, a.rectify);
          } else {
#pragma unroll
            for (int q = 0; q < V; ++q) o.v[q] = outside(p + q);
          }
          *reinterpret_cast<Vec*>(stage + (row - r0) * SR + col) = o;
        }
      }
      __syncthreads();
      if (t >= r0 && t < r0 + rows) read_chunk(r0);
    }
  } else {
    if (a.zero_center) mean = (real)scan_mean<real>(xr, T, misc);  // a first pass over the series; the second one hits the caches
    first = sos_pre<real>(xr[0], mean, a.rectify), last = sos_pre<real>(xr[T - 1], mean, a.rectify);
    for (int r0 = 0; r0 < SCAN_THREADS; r0 += RP) {
      const int rows = min(RP, SCAN_THREADS - r0);
      const int NW = rows * CV;  // vectors of this piece
      __syncthreads();
      for (int w0 = t; w0 < NW; w0 += SCAN_LOADS * SCAN_THREADS) {
        Vec raw[SCAN_LOADS];
#pragma unroll
        for (int u = 0; u < SCAN_LOADS; ++u) {  // all loads of the round go out before any of them is used
          const int w = w0 + u * SCAN_THREADS;
          const int row = w / CV;
          int j = (r0 + row) * C_run + (w - row * CV) * V - EO;
          j = (w < NW && j >= 0 && j < T) ? j : 0;  // outside the recording: any valid vector, replaced below
          raw[u] = *reinterpret_cast<const Vec*>(xr + j);
        }
#pragma unroll
        for (int u = 0; u < SCAN_LOADS; ++u) {
          const int w = w0 + u * SCAN_THREADS;
          if (w < NW) {
            const int row = w / CV;
            const int col = (w - row * CV) * V, p = (r0 + row) * C_run + col, j = p - EO;
            Vec o;
            if (j >= 0 && j < T) {
#pragma unroll
              for (int q = 0; q < V; ++q) o.v[q] = sos_pre<real>(raw[u].v[q], mean, a.rectify);
            } else {
#pragma unroll
              for (int q = 0; q < V; ++q) o.v[q] = outside(p + q);
            }
            *reinterpret_cast<Vec*>(stage + row * SR + col) = o;
          }
        }
      }
      __syncthreads();
      if (t >= r0 && t < r0 + rows) read_chunk(r0);
    }
  }

  // ---- forward: zero-state response, scan, correction --------------------------------------------------------------------
  double s_init[NST], s_start[NST], E[NST];
  if (a.zero_lag) {  // zi * ext[0] (sosfiltfilt)
    __syncthreads();
    if (t == 0) misc[4] = v[0];
    __syncthreads();
    const double x0 = misc[4];
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      s_init[2 * s] = s < ns ? a.zi[s][0] * x0 : 0.0;
      s_init[2 * s + 1] = s < ns ? a.zi[s][1] * x0 : 0.0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NST; ++i) s_init[i] = 0.0;
  }
  {
    double z[NSP][2];
#pragma unroll
    for (int s = 0; s < NSP; ++s) z[s][0] = z[s][1] = 0.0;
#pragma unroll
    for (int n0 = 0; n0 < CMAX; n0 += 4)
      {
#pragma unroll
        for (int n = n0; n < n0 + 4; ++n) v[n] = scan_step<NSP>(v[n], z, c);
      }
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      E[2 * s] = z[s][0];
      E[2 * s + 1] = z[s][1];
    }
  }
  scan_states<NST>(E, s_init, false, Mp, xch, s_start);
#pragma unroll
  for (int n0 = 0; n0 < CMAX; n0 += 4)
    {
#pragma unroll
      for (int n = n0; n < n0 + 4; ++n) {
        double acc = v[n];
#pragma unroll
        for (int i = 0; i < NST; ++i) acc = __builtin_fma(Gl[n * NST + i], s_start[i], acc);
        v[n] = acc;
      }
    }

  if (a.zero_lag) {
    // ---- backward over the forward output: start state zi * y[L-1]; the tail of the last chunk holds that constant ---------
    const int tL = (L - 1) / C_run, nL = (L - 1) - tL * C_run;  // (wave-uniform: one scalar division)
    __syncthreads();
    if (t == tL) {  // the owner's chunk through LDS: a register array cannot be indexed with a run-time value
      double* __restrict__ row = reinterpret_cast<double*>(stage);
#pragma unroll
      for (int n0 = 0; n0 < CMAX; n0 += 4)
        {
#pragma unroll
          for (int n = n0; n < n0 + 4; ++n) row[n] = v[n];
        }
    }
    __syncthreads();
    const double ylast = reinterpret_cast<const double*>(stage)[nL];
    if ((t + 1) * C_run > L) {
#pragma unroll
      for (int n = 0; n < CMAX; ++n) v[n] = (t * C_run + n >= L) ? ylast : v[n];  // (registers beyond C_run are never read)
    }
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      s_init[2 * s] = s < ns ? a.zi[s][0] * ylast : 0.0;
      s_init[2 * s + 1] = s < ns ? a.zi[s][1] * ylast : 0.0;
    }
    {
      double z[NSP][2];
#pragma unroll
      for (int s = 0; s < NSP; ++s) z[s][0] = z[s][1] = 0.0;
#pragma unroll
      for (int n0 = CMAX - 4; n0 >= 0; n0 -= 4)
        {
#pragma unroll
          for (int n = n0 + 3; n >= n0; --n) v[n] = scan_step<NSP>(v[n], z, c);
        }
#pragma unroll
      for (int s = 0; s < NSP; ++s) {
        E[2 * s] = z[s][0];
        E[2 * s + 1] = z[s][1];
      }
    }
    scan_states<NST>(E, s_init, true, Mp, xch, s_start);
    const double* __restrict__ Gr = Gl + (C_run - 1) * NST;  // sample n of a chunk is step C_run - 1 - n of the reversed walk
#pragma unroll
    for (int n0 = 0; n0 < CMAX; n0 += 4)
      {
#pragma unroll
        for (int n = n0; n < n0 + 4; ++n) {
          double acc = v[n];
#pragma unroll
          for (int i = 0; i < NST; ++i) acc = __builtin_fma(Gr[i - n * NST], s_start[i], acc);
          v[n] = acc;
        }
      }
  }

  // ---- registers -> staging rows -> HBM (the T samples of the recording), 16-byte vectors ---------------------------------
  for (int r0 = 0; r0 < SCAN_THREADS; r0 += RP) {
    const int rows = min(RP, SCAN_THREADS - r0);
    const int NW = rows * CV;
    __syncthreads();
    if (t >= r0 && t < r0 + rows) {
      Vec* __restrict__ myrow = reinterpret_cast<Vec*>(stage + (t - r0) * SR);
#pragma unroll
      for (int n0 = 0; n0 < CMAX; n0 += 4)
        {
#pragma unroll
          for (int g = 0; g < 4 / V; ++g) {
            Vec x;
#pragma unroll
            for (int q = 0; q < V; ++q) x.v[q] = (real)v[n0 + g * V + q];
            myrow[n0 / V + g] = x;
          }
        }
    }
    __syncthreads();
    for (int w = t; w < NW; w += SCAN_THREADS) {
      const int row = w / CV;  // (a compile-time divisor: multiply-high)
      const int col = (w - row * CV) * V, j = (r0 + row) * C_run + col - EO;
      if (j >= 0 && j < T) *reinterpret_cast<Vec*>(yr + j) = *reinterpret_cast<const Vec*>(stage + row * SR + col);
    }
  }
}

// =================================================================================================================================
// sosfilt_chunk_kernel (round 4, second version of the time-parallel mode): the same mathematics, the memory side rebuilt like
// emg_chunk_kernel (envelope_chunk.hpp).  sosfilt_scan_kernel moves the series HBM <-> registers through a 48 KiB staging buffer in
// two pieces per direction (four barriers each way, half the threads idle per piece) with 16-byte vectors (alignment and length
// conditions); alone on a CU a workgroup lives 24 us of which 7.4 us are arithmetic.  Here the WHOLE extended series sits in LDS in
// natural order and a thread's chunk length C is ODD (17 / 41 / 79), so the coalesced accesses (thread t <-> position t + 256 k)
// and the threads' own chunks (stride C) are both bank-conflict-free without padding; HBM is read and written with dword buffer
// accesses (no alignment condition, range-checked).  While the chunks are in registers the series buffer is dead: the scan's
// exchange area and the G table overlay it, so LDS = the series (80 120 B for 20 000 float samples of an order-4 zero-lag filter:
// two workgroups per CU).  The odd extension is built in LDS from the centred samples by the first / last `edge` threads.
// float and double up to 256 x 79 / 256 x 41 extended samples; anything else takes sosfilt_scan_kernel or the sequential kernels.
// NT = 64: one wave per series (short series: up to 64 x 41 extended samples; with 256 threads a 1 000-sample series kept 61 of
// them busy) -- the scan's cross-wave phases and the workgroup-wide steps fall away.  NT = 512: float64 series of up to 512 x 41
// extended samples with the CU's whole LDS for one series (20 000 doubles fit the 160 KB).
template <typename real, int NSP, int C, int NT = SCAN_THREADS>
__global__ void __launch_bounds__(NT, (C > 32 ? (NSP > 4 ? 1 : 2) : (NSP > 4 ? 2 : 4)))
sosfilt_chunk_kernel(SosArgs a, const double* __restrict__ tab, int ns, int region_bytes) {
  static_assert(C % 2 == 1, "odd chunk length: conflict-free LDS stride");
  constexpr int NST = 2 * NSP;
  extern __shared__ __attribute__((aligned(16))) unsigned char scan_smem[];
  real* __restrict__ xs = reinterpret_cast<real*>(scan_smem);               // [L] the extended series, later the output
  double* __restrict__ xch = reinterpret_cast<double*>(scan_smem);          // overlay: [256][NST]
  double* __restrict__ Gl = xch + NT * NST;                        // overlay: G [C][NST]
  double* __restrict__ row = Gl + C * NST;                                   // overlay: one chunk (ylast)
  double* __restrict__ misc = reinterpret_cast<double*>(scan_smem + region_bytes);  // [8] behind both
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int series = blockIdx.x;
  const int T = a.T, edge = a.edge, L = T + 2 * edge;
  const real* __restrict__ xr =
      static_cast<const real*>(a.x) + (long long)(series / a.m) * a.bstride + (long long)(series % a.m) * a.ld;
  real* __restrict__ yr = static_cast<real*>(a.y) + (long long)series * T;
  const rsrc_t xrs = make_rsrc(xr, (unsigned)((long long)T * (long long)sizeof(real)));
  const rsrc_t yrs = make_rsrc(yr, (unsigned)((long long)T * (long long)sizeof(real)));
  const double* __restrict__ G = tab;
  const double* __restrict__ Mp = tab + SCAN_G_CAP;
  const unsigned voff = (unsigned)t * (unsigned)sizeof(real);

#ifdef HIPNMF_SOS_TIMING  // development build (tools/sos_phase_timing.py): shader-clock stamps at the phase boundaries, written
  long long stamp[9];     // over the head of the output
  long long sub[4] = {0, 0, 0, 0};  // finer stamps inside phase 2 (written behind the nine phase durations)
#define SOS_SUB(i) sub[i] = (long long)__builtin_readcyclecounter()
  int n_stamp = 0;
#define SOS_STAMP() stamp[n_stamp++] = (long long)__builtin_readcyclecounter()
#else
#define SOS_STAMP() ((void)0)
#define SOS_SUB(i) ((void)0)
#endif
  SOS_STAMP();  // 0: start
  double c[NSP][5];
  scan_coeffs<NSP>(a, ns, c);

  // ---- HBM -> registers -> LDS (centred / rectified in the samples' precision, as the reference's array arithmetic) ------------
  double v[C];
  {
    real raw[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      real r1[1];
      buf_load<real, 1>(xrs, voff, (unsigned)(k * NT) * (unsigned)sizeof(real), r1);
      raw[k] = r1[0];
    }
#ifdef HIPNMF_SOS_TIMING
    __builtin_amdgcn_s_waitcnt(0);
#endif
    SOS_STAMP();  // 1: the samples have arrived
    real mean = (real)0;
    if (a.zero_center) {  // fp64 sum in a fixed order (zeros past the end of the series)
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < C; ++k) sum += (double)raw[k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
      if (lane == 0) misc[wave] = sum;
      __syncthreads();
      double tot = misc[0];
      if constexpr (NT >= 256) tot = (misc[0] + misc[1]) + (misc[2] + misc[3]);
      if constexpr (NT == 512) tot += (misc[4] + misc[5]) + (misc[6] + misc[7]);
      mean = (real)(tot / (double)T);
    }
    SOS_SUB(0);  // mean known
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const int j = t + k * NT;
      const real pv = sos_pre<real>(raw[k], mean, a.rectify);
      xs[j < T ? edge + j : L] = pv;  // (slot L: a dump slot behind the series -- a branch per store would serialise them)
    }
    __syncthreads();
    SOS_SUB(1);  // samples staged in LDS
    for (int e = t; e < edge; e += NT) {  // scipy's odd extension about the end samples, in the samples' precision
      xs[e] = (real)2 * xs[edge] - xs[2 * edge - e];
      xs[edge + T + e] = (real)2 * xs[edge + T - 1] - xs[edge + T - 2 - e];
    }
    __syncthreads();
    SOS_SUB(2);  // odd extension built
    const real* __restrict__ mine = xs + C * t;
    const int nval = L - C * t;  // positions of this chunk inside the extended series (zeros behind it)
#pragma unroll
    for (int n = 0; n < C; ++n) {
      const real xv = mine[n];  // (behind the series: whatever is there -- LDS reads cannot fault -- replaced by 0)
      v[n] = n < nval ? (double)xv : 0.0;
    }
  }
  const double x0 = (double)xs[0];
  __syncthreads();  // the series buffer is dead from here: the overlay takes it
  SOS_STAMP();  // 2: mean, staging through LDS, odd extension, chunk in registers

  // ---- forward: zero-state response, scan, correction --------------------------------------------------------------------
  double s_init[NST], s_start[NST], E[NST];
#pragma unroll
  for (int s = 0; s < NSP; ++s) {  // zi * ext[0] (sosfiltfilt); plain sosfilt starts at rest
    s_init[2 * s] = (a.zero_lag && s < ns) ? a.zi[s][0] * x0 : 0.0;
    s_init[2 * s + 1] = (a.zero_lag && s < ns) ? a.zi[s][1] * x0 : 0.0;
  }
  {
    double z[NSP][2];
#pragma unroll
    for (int s = 0; s < NSP; ++s) z[s][0] = z[s][1] = 0.0;
#pragma unroll
    for (int n = 0; n < C; ++n) v[n] = scan_step<NSP>(v[n], z, c);
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      E[2 * s] = z[s][0];
      E[2 * s + 1] = z[s][1];
    }
  }
  SOS_STAMP();  // 3: forward zero-state response
  scan_states<NST, NT>(E, s_init, false, Mp, xch, s_start);
  SOS_STAMP();  // 4: forward scan
#pragma unroll
  for (int n = 0; n < C; ++n) {
    double acc = v[n];
#pragma unroll
    // G through the scalar cache (a uniform address with compile-time offsets: s_load + an SGPR operand of the FMA).  Round 4 read
    // it from an LDS overlay: 4 C broadcast LDS reads per thread and direction held the correction at 3.9 k cycles per pass;
    // 1024 x 16 x 20 000 fp32 zero-lag: 0.898 -> 0.839 ms.
    for (int i = 0; i < NST; ++i) acc = __builtin_fma(G[n * NST + i], s_start[i], acc);
    v[n] = acc;
  }

  SOS_STAMP();  // 5: forward correction
  if (a.zero_lag) {
    // ---- backward over the forward output: start state zi * y[L-1]; the tail of the last chunk holds that constant ---------
    const int tL = (L - 1) / C, nL = (L - 1) - tL * C;  // (wave-uniform)
    if (t == tL) {  // the owner's chunk through LDS: a register array cannot be indexed with a run-time value
#pragma unroll
      for (int n = 0; n < C; ++n) row[n] = v[n];
    }
    __syncthreads();
    const double ylast = row[nL];
    if ((t + 1) * C > L) {
#pragma unroll
      for (int n = 0; n < C; ++n) v[n] = (t * C + n >= L) ? ylast : v[n];
    }
#pragma unroll
    for (int s = 0; s < NSP; ++s) {
      s_init[2 * s] = s < ns ? a.zi[s][0] * ylast : 0.0;
      s_init[2 * s + 1] = s < ns ? a.zi[s][1] * ylast : 0.0;
    }
    {
      double z[NSP][2];
#pragma unroll
      for (int s = 0; s < NSP; ++s) z[s][0] = z[s][1] = 0.0;
#pragma unroll
      for (int n = C - 1; n >= 0; --n) v[n] = scan_step<NSP>(v[n], z, c);
#pragma unroll
      for (int s = 0; s < NSP; ++s) {
        E[2 * s] = z[s][0];
        E[2 * s + 1] = z[s][1];
      }
    }
    scan_states<NST, NT>(E, s_init, true, Mp, xch, s_start);
#pragma unroll
    for (int n = 0; n < C; ++n) {
      double acc = v[n];
#pragma unroll
      for (int i = 0; i < NST; ++i) acc = __builtin_fma(G[(C - 1 - n) * NST + i], s_start[i], acc);  // sample n = step C - 1 - n of the reversed walk
      v[n] = acc;
    }
  }

  SOS_STAMP();  // 6: backward pass (zero-state response, scan, correction)
  // ---- registers -> LDS (natural order) -> HBM, the T samples of the recording -----------------------------------------------
  __syncthreads();  // the overlay is dead
  {
    real* __restrict__ mine = xs + C * t;
    const int nval = L - C * t;
#pragma unroll
    for (int n = 0; n < C; ++n) mine[n < nval ? n : L - C * t] = (real)v[n];  // (behind the series: the dump slot)
  }
  __syncthreads();
  SOS_STAMP();  // 7: output staged in LDS
  constexpr int CB = 8;
  const int kmax = (T + NT - 1) / NT;
  for (int k0 = 0; k0 < kmax; k0 += CB) {
    real y[CB];
    const real* __restrict__ src = xs + edge + t + k0 * NT;  // (reads past the series cannot fault; the stores are dropped)
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = src[u * NT];
#pragma unroll
    for (int u = 0; u < CB; ++u) buf_store<real>(yrs, voff, (unsigned)((k0 + u) * NT) * (unsigned)sizeof(real), y[u]);
  }
#ifdef HIPNMF_SOS_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  SOS_STAMP();  // 8: stores acknowledged
  __syncthreads();
  if (t == 0) {
    for (int i = 0; i < 8; ++i) yr[i] = (real)(stamp[i + 1] - stamp[i]);
    yr[8] = (real)(stamp[0] & 0xffffff);  // (low bits of the start time: order of the workgroups)
    yr[9] = (real)(sub[0] - stamp[1]);    // phase 2 in four parts: mean | staging stores + barrier | extension + barrier | chunk reads
    yr[10] = (real)(sub[1] - sub[0]);     // (the stamps are not scheduling barriers: the compiler moves the forward zero-state
    yr[11] = (real)(sub[2] - sub[1]);     //  recurrence up into the fourth part, next to the LDS reads it consumes)
    yr[12] = (real)(stamp[2] - sub[2]);
  }
#endif
#undef SOS_STAMP
#undef SOS_SUB
}

// =================================================================================================================================
// Long series (round 4): more than one workgroup's worth of samples -- a whole recording of minutes at 2 kHz is 10^5 .. 10^6
// samples, and the sequential kernels need 20 ms for ONE 16 x 200 000 frame (a dependent chain of 400 000 steps).  The chunk
// algebra nests: a block of 256 chunks entered with state s0 leaves the state F_blk + M^256 s0 behind, so per direction
//   state pass   every (series, block) workgroup filters its block from rest and stores F_blk          (sosfilt_block_kernel, full = 0)
//   block scan   one thread per series walks its blocks: start_b = s,  s <- F_b + M^256 s                (sos_block_scan_kernel)
//   full pass    every workgroup filters its block again, entered with start_b, and writes it out       (full = 1)
// forward over x (centred / rectified, odd extension at the recording's ends) into a workspace, backward over the workspace into y.
// Each pass is one chip-filling launch with a dependent chain of one block; the data are read four times and written twice
// instead of once each -- bandwidth that a single long frame does not miss and a batch of them trades for the chain.
template <typename real>
struct SosBlockArgs {
  const real* src;            // nullptr: the recording a.x (centred / rectified / extended); else the forward output [N][L]
  real* dst;                  // full pass: the forward output [N][L] (forward) or y [N][T] (backward, or forward of a causal filter)
  const double* stat;         // [N][3]: mean, first, last pre-processed sample (sos_stats_kernel)
  double* block_end;          // state pass: [N][NB][NST] state a block leaves behind when entered at rest
  const double* block_start;  // full pass: [N][NB][NST] state every block is entered with
  int L, NB, backward, full, dst_is_y;
};

template <typename real, int NSP>
__global__ void __launch_bounds__(64) sos_block_scan_kernel(SosArgs a, SosBlockArgs<real> k, const double* __restrict__ tab, int ns,
                                                            double* __restrict__ block_start) {
  constexpr int NST = 2 * NSP;
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.N) return;
  const double* __restrict__ M256 = tab + SCAN_G_CAP + 8 * NST * NST;
  double st[NST];
  double x0 = 0.0;
  if (a.zero_lag) {
    if (k.backward) {
      x0 = (double)k.src[(long long)s * k.L + k.L - 1];  // y_fwd[L - 1]
    } else {
      const real* __restrict__ xr = static_cast<const real*>(a.x) + (long long)(s / a.m) * a.bstride + (long long)(s % a.m) * a.ld;
      const real mean = (real)k.stat[3LL * s], first = (real)k.stat[3LL * s + 1];
      x0 = (double)(a.edge > 0 ? (real)2 * first - sos_pre<real>(xr[a.edge], mean, a.rectify) : first);  // ext[0]
    }
  }
#pragma unroll
  for (int q = 0; q < NSP; ++q) {
    st[2 * q] = (a.zero_lag && q < ns) ? a.zi[q][0] * x0 : 0.0;
    st[2 * q + 1] = (a.zero_lag && q < ns) ? a.zi[q][1] * x0 : 0.0;
  }
  for (int i = 0; i < k.NB; ++i) {
    const int b = k.backward ? k.NB - 1 - i : i;
    double* __restrict__ o = block_start + ((long long)s * k.NB + b) * NST;
    const double* __restrict__ e = k.block_end + ((long long)s * k.NB + b) * NST;
    double nx[NST];
#pragma unroll
    for (int r = 0; r < NST; ++r) {
      o[r] = st[r];
      double acc = e[r];
#pragma unroll
      for (int c2 = 0; c2 < NST; ++c2) acc = __builtin_fma(M256[r * NST + c2], st[c2], acc);
      nx[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < NST; ++r) st[r] = nx[r];
  }
}

// One workgroup per (series, block): grid.x = N * NB.  Dynamic LDS as sosfilt_chunk_kernel (the block in natural order, the
// overlay, 8 doubles behind).
template <typename real, int NSP, int C>
__global__ void __launch_bounds__(SCAN_THREADS, (C > 32 ? (NSP > 4 ? 1 : 2) : (NSP > 4 ? 2 : 4)))
sosfilt_block_kernel(SosArgs a, SosBlockArgs<real> k, const double* __restrict__ tab, int ns, int region_bytes) {
  static_assert(C % 2 == 1, "odd chunk length: conflict-free LDS stride");
  constexpr int NST = 2 * NSP, NT = SCAN_THREADS, LB = NT * C;
  extern __shared__ __attribute__((aligned(16))) unsigned char scan_smem[];
  real* __restrict__ xs = reinterpret_cast<real*>(scan_smem);
  double* __restrict__ xch = reinterpret_cast<double*>(scan_smem);
  double* __restrict__ Gl = xch + NT * NST;
  const int t = threadIdx.x;
  const int series = (int)(blockIdx.x / (unsigned)k.NB), b = (int)(blockIdx.x % (unsigned)k.NB);
  const int T = a.T, edge = a.edge, L = k.L;
  const int p0 = b * LB;
  const int nblk = (L - p0 < LB) ? L - p0 : LB;  // positions of the (extended) series in this block
  const double* __restrict__ G = tab;
  const double* __restrict__ Mp = tab + SCAN_G_CAP;
  const unsigned voff = (unsigned)t * (unsigned)sizeof(real);

  double c[NSP][5];
  scan_coeffs<NSP>(a, ns, c);
  constexpr int GN = (C * NST + NT - 1) / NT;
  double gpre[GN];
#pragma unroll
  for (int u = 0; u < GN; ++u) {
    const int i = t + u * NT;
    gpre[u] = (k.full && i < C * NST) ? G[i] : 0.0;
  }

  // ---- the block -> LDS ------------------------------------------------------------------------------------------------------
  double tailv = 0.0;  // value of the positions behind the series: 0 (forward), y_fwd[L - 1] (backward: the recursion then starts in its steady state)
  if (k.src) {
    const real* __restrict__ sr = k.src + (long long)series * L;
    const rsrc_t rs = make_rsrc(sr + p0, (unsigned)((long long)nblk * (long long)sizeof(real)));
    real raw[C];
#pragma unroll
    for (int q = 0; q < C; ++q) {
      real r1[1];
      buf_load<real, 1>(rs, voff, (unsigned)(q * NT) * (unsigned)sizeof(real), r1);
      raw[q] = r1[0];
    }
    if (k.backward) tailv = (double)sr[L - 1];
#pragma unroll
    for (int q = 0; q < C; ++q) {
      const int i = t + q * NT;
      xs[i < nblk ? i : LB] = raw[q];  // (slot LB: a dump slot behind the block)
    }
  } else {
    const real* __restrict__ xr = static_cast<const real*>(a.x) + (long long)(series / a.m) * a.bstride + (long long)(series % a.m) * a.ld;
    const rsrc_t rs = make_rsrc(xr, (unsigned)((long long)T * (long long)sizeof(real)));
    const real mean = (real)k.stat[3LL * series], first = (real)k.stat[3LL * series + 1], last = (real)k.stat[3LL * series + 2];
    real raw[C];
#pragma unroll
    for (int q = 0; q < C; ++q) {  // extended position p0 + i <-> sample p0 + i - edge (outside the recording: an offset the range check refuses)
      const long long j = (long long)p0 - edge + t + q * NT;
      real r1[1];
      buf_load<real, 1>(rs, (j >= 0 && j < T) ? (unsigned)((unsigned long long)j * sizeof(real)) : OOB, 0u, r1);
      raw[q] = r1[0];
    }
#pragma unroll
    for (int q = 0; q < C; ++q) {
      const int i = t + q * NT;
      xs[i < nblk ? i : LB] = sos_pre<real>(raw[q], mean, a.rectify);
    }
    __syncthreads();
    // scipy's odd extension about the recording's end samples, where this block holds a piece of it (straight from memory: the
    // reflected samples may belong to a neighbouring block)
    for (int e = t; e < edge; e += NT) {
      const int pf = e - p0, pt = edge + T + e - p0;
      if (pf >= 0 && pf < nblk) xs[pf] = (real)2 * first - sos_pre<real>(xr[edge - e], mean, a.rectify);
      if (pt >= 0 && pt < nblk) xs[pt] = (real)2 * last - sos_pre<real>(xr[T - 2 - e], mean, a.rectify);
    }
  }
  __syncthreads();
  double v[C];
  {
    const real* __restrict__ mine = xs + C * t;
    const int nval = nblk - C * t;
#pragma unroll
    for (int n = 0; n < C; ++n) {
      const real xv = mine[n];
      v[n] = n < nval ? (double)xv : tailv;
    }
  }
  __syncthreads();  // the block buffer is dead from here: the overlay takes it
  if (k.full) {
#pragma unroll
    for (int u = 0; u < GN; ++u) {
      const int i = t + u * NT;
      if (i < C * NST) Gl[i] = gpre[u];
    }
  }
  double s_init[NST], s_start[NST], E[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) s_init[i] = k.block_start ? k.block_start[((long long)series * k.NB + b) * NST + i] : 0.0;
  {
    double z[NSP][2];
#pragma unroll
    for (int q = 0; q < NSP; ++q) z[q][0] = z[q][1] = 0.0;
    if (k.backward) {
#pragma unroll
      for (int n = C - 1; n >= 0; --n) v[n] = scan_step<NSP>(v[n], z, c);
    } else {
#pragma unroll
      for (int n = 0; n < C; ++n) v[n] = scan_step<NSP>(v[n], z, c);
    }
#pragma unroll
    for (int q = 0; q < NSP; ++q) {
      E[2 * q] = z[q][0];
      E[2 * q + 1] = z[q][1];
    }
  }
  scan_states<NST, NT>(E, s_init, k.backward != 0, Mp, xch, s_start);
  if (!k.full) {  // the state the whole block leaves behind: the inclusive value of the last chunk of the walk
    if (t < NST) k.block_end[((long long)series * k.NB + b) * NST + t] = xch[(NT - 1) * NST + t];
    return;
  }
  if (k.backward) {
    const double* __restrict__ Gr = Gl + (C - 1) * NST;
#pragma unroll
    for (int n = 0; n < C; ++n) {
      double acc = v[n];
#pragma unroll
      for (int i = 0; i < NST; ++i) acc = __builtin_fma(Gr[i - n * NST], s_start[i], acc);
      v[n] = acc;
    }
  } else {
#pragma unroll
    for (int n = 0; n < C; ++n) {
      double acc = v[n];
#pragma unroll
      for (int i = 0; i < NST; ++i) acc = __builtin_fma(Gl[n * NST + i], s_start[i], acc);
      v[n] = acc;
    }
  }
  __syncthreads();  // the overlay is dead
  {
    real* __restrict__ mine = xs + C * t;
    const int nval = nblk - C * t;
#pragma unroll
    for (int n = 0; n < C; ++n) mine[n < nval ? n : LB - C * t] = (real)v[n];
  }
  __syncthreads();
  // block position i <-> extended position p0 + i; y keeps the positions [edge, edge + T)
  const int shift = k.dst_is_y ? edge : 0, len = k.dst_is_y ? T : L;
  real* __restrict__ dr = k.dst + (long long)series * len;
  const rsrc_t ds = make_rsrc(dr, (unsigned)((long long)len * (long long)sizeof(real)));
  constexpr int CB = 8;
  for (int k0 = 0; k0 < C; k0 += CB) {
    real y[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) y[u] = xs[(t + (k0 + u) * NT) < LB ? t + (k0 + u) * NT : LB];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int i = t + (k0 + u) * NT;
      const long long j = (long long)p0 + i - shift;  // (outside [0, len): an offset the range check refuses -- no branch around the store)
      const bool ok = k0 + u < C && i < nblk && j >= 0 && j < len;
      buf_store<real>(ds, ok ? (unsigned)((unsigned long long)j * sizeof(real)) : OOB, 0u, y[u]);
    }
  }
}

}  // namespace hipnmf
