// hipnmf_sosfilt.hip -- C ABI of the batched IIR filter stage (include/hip_nmf.h, row f-1 of SURVEY.md section 8):
// scipy.signal.sosfilt / sosfiltfilt as the reference's digital_filter / linear_envelope use them
// (src/muscle_synergies/analysis.py:252-432).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#include "hipnmf_internal.hpp"
#include "nmf_kernels.hpp"  // x_to_channel_major_kernel
#include "sosfilt_kernels.hpp"
#include "sosfilt_scan.hpp"

using namespace hipnmf;

namespace {

// scipy.signal.sosfiltfilt's default padlen: 3 * (2 n_sections + 1 - min(#(b2 == 0), #(a2 == 0)))
int default_padlen(const double* sos, int ns) {
  int zb = 0, za = 0;
  for (int s = 0; s < ns; ++s) {
    zb += sos[6 * s + 2] == 0.0;
    za += sos[6 * s + 5] == 0.0;
  }
  return 3 * (2 * ns + 1 - std::min(zb, za));
}

// scipy.signal.sosfilt_zi: steady-state step-response state of every section ((I - A^T) zi = B per section),
// scaled by the DC gain of the sections before it.  (scipy solves the 2 x 2 system with LAPACK; this closed form
// can differ from it in the last bit -- pass scipy's zi for bit parity.)
void steady_state_zi(const double* sos, int ns, double (*zi)[2]) {
  double scale = 1.0;
  for (int s = 0; s < ns; ++s) {
    const double b0 = sos[6 * s], b1 = sos[6 * s + 1], b2 = sos[6 * s + 2];
    const double a1 = sos[6 * s + 4], a2 = sos[6 * s + 5];
    const double B0 = b1 - a1 * b0, B1 = b2 - a2 * b0;
    // [[1 + a1, -1], [a2, 1]] zi = [B0, B1]
    const double det = (1.0 + a1) + a2;
    zi[s][0] = scale * ((B0 + B1) / det);
    zi[s][1] = scale * (((1.0 + a1) * B1 - a2 * B0) / det);
    scale *= (b0 + b1 + b2) / (1.0 + a1 + a2);
  }
}

// the sequential (bit-exact) kernel: LPS lanes per series (sections rounded up to a power of two), a second wave moves the data
template <typename real, int LPS>
hipError_t launch_v2(hipnmf_handle* h, const SosArgs& a, const double* stat, int ns, hipStream_t st) {
  constexpr int S = sos2_series<LPS>();
  constexpr size_t smem = sos2_smem_bytes<LPS>();
  const void* kern = reinterpret_cast<const void*>(&sosfilt2_kernel<real, LPS>);
  if (smem > 48 * 1024 && hipnmf_allow_full_lds(h, kern)) return hipErrorInvalidValue;
  HIPNMF_LAUNCH((sosfilt2_kernel<real, LPS>), dim3((a.N + S - 1) / S), dim3(128), smem, st, a, stat, ns);
  return hipSuccess;
}

// time-parallel mode (sosfilt_scan.hpp): tables for this filter and chunk length, then one workgroup per series
template <typename real, int NSP>
int launch_scan(hipnmf_handle* h, const SosArgs& a, int ns, int C_run, double* tab, hipStream_t st) {
  constexpr int NST = 2 * NSP;
  HIPNMF_LAUNCH((sos_scan_tables_kernel<NSP>), dim3(1), dim3(256), 0, st, a, ns, C_run, tab);
  const size_t smem = (size_t)SCAN_STAGE_BYTES + sizeof(double) * (SCAN_THREADS * NST + 8 + (size_t)C_run * NST);
  auto go = [&](auto kern) -> int {
    if (smem > 48 * 1024)
      if (int rc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(kern))) return rc;
    HIPNMF_LAUNCH(kern, dim3((unsigned)a.N), dim3(SCAN_THREADS), smem, st, a, (const double*)tab, ns);
    return HIPNMF_OK;
  };
  return C_run == 16 ? go(sosfilt_scan_kernel<real, NSP, 16>) : C_run == 40 ? go(sosfilt_scan_kernel<real, NSP, 40>)
                                                                             : go(sosfilt_scan_kernel<real, NSP, SCAN_CMAX>);
}

// second version of the time-parallel mode (sosfilt_chunk_kernel): the whole extended series in LDS, chunk length C odd; NT = 64:
// one wave per series (short series)
template <typename real, int NSP>
int launch_chunk_scan(hipnmf_handle* h, const SosArgs& a, int ns, int C, int NT, size_t region, double* tab, hipStream_t st) {
  HIPNMF_LAUNCH((sos_scan_tables_kernel<NSP>), dim3(1), dim3(256), 0, st, a, ns, C, tab);
  const size_t smem = region + 8 * sizeof(double);
  auto go = [&](auto kern) -> int {
    if (smem > 48 * 1024)
      if (int rc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(kern))) return rc;
    HIPNMF_LAUNCH(kern, dim3((unsigned)a.N), dim3(NT), smem, st, a, (const double*)tab, ns, (int)region);
    return HIPNMF_OK;
  };
  if (NT == 512) {
    if constexpr (sizeof(real) == 8 && NSP <= 4) {
      if (C == 41) return go(sosfilt_chunk_kernel<real, NSP, 41, 512>);
    }
    return HIPNMF_ERR_UNSUPPORTED;
  }
  if (NT == 64) {
    if constexpr (NSP <= 4) {
      if (C == 5) return go(sosfilt_chunk_kernel<real, NSP, 5, 64>);
      if (C == 9) return go(sosfilt_chunk_kernel<real, NSP, 9, 64>);
      if (C == 17) return go(sosfilt_chunk_kernel<real, NSP, 17, 64>);
      if (C == 25) return go(sosfilt_chunk_kernel<real, NSP, 25, 64>);
      if (C == 33) return go(sosfilt_chunk_kernel<real, NSP, 33, 64>);
      if (C == 41) return go(sosfilt_chunk_kernel<real, NSP, 41, 64>);
    }
    return HIPNMF_ERR_UNSUPPORTED;
  }
  if (C == 17) return go(sosfilt_chunk_kernel<real, NSP, 17, SCAN_THREADS>);
  if (C == 41) return go(sosfilt_chunk_kernel<real, NSP, 41, SCAN_THREADS>);
  if constexpr (NSP <= 4) {  // intermediate lengths (idle threads cost as much as busy ones); filters of more than four sections keep three
    if (C == 25) return go(sosfilt_chunk_kernel<real, NSP, 25, SCAN_THREADS>);
    if (C == 33) return go(sosfilt_chunk_kernel<real, NSP, 33, SCAN_THREADS>);
    if constexpr (sizeof(real) == 4) {
      if (C == 49) return go(sosfilt_chunk_kernel<real, NSP, 49, SCAN_THREADS>);
      if (C == 57) return go(sosfilt_chunk_kernel<real, NSP, 57, SCAN_THREADS>);
      if (C == 65) return go(sosfilt_chunk_kernel<real, NSP, 65, SCAN_THREADS>);
    }
  }
  if constexpr (sizeof(real) == 4) {
    if (C == 79) return go(sosfilt_chunk_kernel<real, NSP, 79, SCAN_THREADS>);
  }
  return HIPNMF_ERR_UNSUPPORTED;
}
// LDS of sosfilt_chunk_kernel without the 8 doubles behind it: the series + a dump slot, or the overlay, whichever is larger
inline size_t chunk_scan_region(size_t L, int C, int nt, int nsp, size_t sizeof_real) {
  const size_t series = (L + 1) * sizeof_real, overlay = sizeof(double) * ((size_t)nt * 2 * nsp + (size_t)C * 2 * nsp + (size_t)C);
  return (std::max(series, overlay) + 15) & ~(size_t)15;
}

// long series (sosfilt_block_kernel): state pass, block scan, full pass per direction
template <typename real, int NSP>
int launch_block_scan(hipnmf_handle* h, const SosArgs& a, int ns, long long L, int NB, size_t region, double* tab, const double* stat,
                      real* fwd, double* bend, double* bstart, real* y, hipStream_t st) {
  constexpr int C = sizeof(real) == 4 ? 79 : 41;
  HIPNMF_LAUNCH((sos_scan_tables_kernel<NSP>), dim3(1), dim3(256), 0, st, a, ns, C, tab);
  const size_t smem = region + 8 * sizeof(double);
  auto kern = sosfilt_block_kernel<real, NSP, C>;
  if (smem > 48 * 1024)
    if (int rc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(kern))) return rc;
  const dim3 grid((unsigned)((long long)a.N * NB)), gscan((unsigned)((a.N + 63) / 64));
  SosBlockArgs<real> k;
  k.stat = stat;
  k.L = (int)L;
  k.NB = NB;
  for (int dir = 0; dir < (a.zero_lag ? 2 : 1); ++dir) {
    k.backward = dir;
    k.src = dir ? fwd : nullptr;
    k.block_end = bend;
    k.block_start = nullptr;
    k.full = 0;
    k.dst = nullptr;
    k.dst_is_y = 0;
    HIPNMF_LAUNCH(kern, grid, dim3(SCAN_THREADS), smem, st, a, k, (const double*)tab, ns, (int)region);
    HIPNMF_LAUNCH((sos_block_scan_kernel<real, NSP>), gscan, dim3(64), 0, st, a, k, (const double*)tab, ns, bstart);
    k.block_start = bstart;
    k.full = 1;
    const bool last = dir == (a.zero_lag ? 1 : 0);
    k.dst = last ? y : fwd;
    k.dst_is_y = last ? 1 : 0;
    HIPNMF_LAUNCH(kern, grid, dim3(SCAN_THREADS), smem, st, a, k, (const double*)tab, ns, (int)region);
  }
  return HIPNMF_OK;
}

template <typename real>
int sosfilt_impl(hipnmf_handle* h, const hipnmf_sosfilt_params* p, const double* sos, const double* zi, const real* x,
                 real* y) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (!p) return fail(HIPNMF_ERR_BAD_ARG, "params is NULL");
  if (p->struct_size != (int32_t)sizeof(hipnmf_sosfilt_params))
    return fail(HIPNMF_ERR_BAD_ARG, "hipnmf_sosfilt_params.struct_size = %d, library expects %d", p->struct_size,
                (int)sizeof(hipnmf_sosfilt_params));
  if (!sos || !x || !y) return fail(HIPNMF_ERR_BAD_ARG, "sos (host), x and y (device) must be non-NULL");
  if (p->batch < 1 || p->n_samples < 1 || p->n_samples > 1000000000LL || p->n_channels < 1)
    return fail(HIPNMF_ERR_BAD_ARG, "bad shape: batch=%d n_samples=%lld n_channels=%d", p->batch,
                (long long)p->n_samples, p->n_channels);
  if ((long long)p->batch * p->n_channels > 2000000000LL) return fail(HIPNMF_ERR_BAD_ARG, "too many series");
  if (p->mode != HIPNMF_SOSFILT_EXACT && p->mode != HIPNMF_SOSFILT_SCAN)
    return fail(HIPNMF_ERR_BAD_ARG, "hipnmf_sosfilt_params.mode must be HIPNMF_SOSFILT_EXACT (0) or HIPNMF_SOSFILT_SCAN (1) (got %d)", p->mode);
  if (p->n_sections < 1) return fail(HIPNMF_ERR_BAD_ARG, "n_sections must be >= 1 (got %d)", p->n_sections);
  if (p->n_sections > SOS_MAX_SECTIONS)
    return fail(HIPNMF_ERR_UNSUPPORTED, "n_sections=%d outside the compiled kernel set (max %d)", p->n_sections,
                SOS_MAX_SECTIONS);
  for (int s = 0; s < p->n_sections; ++s) {
    if (sos[6 * s + 3] != 1.0) return fail(HIPNMF_ERR_BAD_ARG, "sos[:, 3] should be all ones");
    for (int q = 0; q < 6; ++q)
      if (!std::isfinite(sos[6 * s + q])) return fail(HIPNMF_ERR_BAD_ARG, "sos contains NaN or infinity");
  }
  if (p->x_layout != HIPNMF_X_ROW_MAJOR && p->x_layout != HIPNMF_X_CHANNEL_MAJOR)
    return fail(HIPNMF_ERR_BAD_ARG, "bad x_layout %d", p->x_layout);
  const long long min_ld = (p->x_layout == HIPNMF_X_ROW_MAJOR) ? p->n_channels : p->n_samples;
  if (p->ldx < min_ld) return fail(HIPNMF_ERR_BAD_ARG, "ldx=%lld smaller than %lld", (long long)p->ldx, min_ld);
  const int zero_lag = p->zero_lag ? 1 : 0;
  int edge = 0;
  if (zero_lag) {
    edge = p->padlen < 0 ? default_padlen(sos, p->n_sections) : p->padlen;
    if (p->n_samples <= edge)
      return fail(HIPNMF_ERR_BAD_ARG, "The length of the input vector x must be greater than padlen, which is %d.", edge);
  }
  HIP_TRY(hipSetDevice(h->device));
  const int B = p->batch, m = p->n_channels;
  const long long T = p->n_samples, L = T + 2LL * edge;
  const long long N = (long long)B * m;
  hipStream_t st = h->stream;

  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const bool inplace = p->x_layout == HIPNMF_X_CHANNEL_MAJOR;
  const size_t o_x = inplace ? 0 : carve(sizeof(real) * (size_t)N * T);
  // forward output of the zero-lag filter: SOS_SERIES rows per wave, so the last wave needs no row guards
  // forward output of the zero-lag filter: whole 64-sample tiles, 64 rows per workgroup-group (both kernels fit)
  // (sosfilt3_kernel -- section states checkpointed per tile, forward output recomputed in the backward pass: half the traffic --
  //  was measured slower, 2.40 vs 2.08 ms at 1024 x 16 x 20 000 fp32 order 4, because the filter is bound by the dependent fp64
  //  chain of the recursion, not by bytes; it and the round-1 sosfilt_kernel were removed in round 5: HISTORY.md)
  // workspace of the zero-lag sequential kernel: the forward output over the extended signal (whole 64-sample tiles, 64 rows per
  // workgroup group)
  const size_t ws_v2 = sizeof(double) * (size_t)round_up(N, 64) * (size_t)round_up(L, 64);
  // time-parallel mode: the whole extended series in the registers of one workgroup (256 chunks of at most SCAN_CMAX samples, whole
  // groups of four), all staging traffic in 16-byte vectors: series and outputs 16-byte aligned, n_samples a multiple of the vector
  constexpr int VS = 16 / (int)sizeof(real);
  const int scan_delta = (VS - edge % VS) % VS;  // front padding that aligns the extended positions with the raw samples
  const long long scan_need = (L + scan_delta + SCAN_THREADS - 1) / SCAN_THREADS;  // samples per thread; compiled: 16, 40, SCAN_CMAX
  const int scan_chunk = scan_need <= 16 ? 16 : scan_need <= 40 ? 40 : scan_need <= SCAN_CMAX ? SCAN_CMAX : SCAN_CMAX + 1;
  const bool scan_aligned = (T % VS) == 0 && (reinterpret_cast<uintptr_t>(y) % 16) == 0 &&
                            (!inplace || ((reinterpret_cast<uintptr_t>(x) % 16) == 0 && (p->ldx % VS) == 0 && (p->x_batch_stride % VS) == 0));
  // (the first version loads a float64 series in two passes; without the backward pass that loses to the sequential kernel --
  // 1.60 vs 1.49 ms at 1024 x 16 x 20 000 -- so causal float64 filters beyond the second version's range stay sequential)
  const bool scan_fits = p->mode == HIPNMF_SOSFILT_SCAN && scan_chunk <= SCAN_CMAX && scan_aligned &&
                         !(sizeof(real) == 8 && !zero_lag && scan_chunk > 40);
  // second version (sosfilt_chunk_kernel): no alignment or length condition; float up to 256 x 79, double up to 256 x 41 extended
  // samples; HIPNMF_SOS_CHUNK=0 leaves the mode to sosfilt_scan_kernel
  static const bool chunk_scan_ok = [] {
    const char* e = getenv("HIPNMF_SOS_CHUNK");
    return !(e && atoi(e) == 0);
  }();
  const int nsp_c = p->n_sections == 1 ? 1 : p->n_sections == 2 ? 2 : p->n_sections <= 4 ? 4 : 8;
  int chunk_c = 0, chunk_nt = SCAN_THREADS;
  static const bool chunk_one_ok = [] {
    const char* e = getenv("HIPNMF_SOS_CHUNK_64");
    return !(e && atoi(e) == 0);
  }();
  if (p->mode == HIPNMF_SOSFILT_SCAN && chunk_scan_ok && chunk_one_ok && nsp_c <= 4 && L <= 64LL * 41) {
    chunk_nt = 64;  // one wave per series
    chunk_c = L <= 64 * 5 ? 5 : L <= 64 * 9 ? 9 : L <= 64 * 17 ? 17 : L <= 64 * 25 ? 25 : L <= 64 * 33 ? 33 : 41;
  } else if (p->mode == HIPNMF_SOSFILT_SCAN && chunk_scan_ok)
  {
    const bool fine = nsp_c <= 4;  // (the intermediate lengths are compiled for up to four sections)
    const int sizes[] = {17, 25, 33, 41, 49, 57, 65, 79};
    for (int c : sizes) {
      if (!fine && c != 17 && c != 41 && c != 79) continue;
      if (sizeof(real) == 8 && c > 41) break;
      if (L <= 256LL * c) {
        chunk_c = c;
        break;
      }
    }
  }
  // float64 beyond 256 x 41 extended samples: eight waves and the CU's whole LDS for one series (HIPNMF_SOS_CHUNK_512=0: the first
  // version's two-pass loading)
  static const bool chunk_512_ok = [] {
    const char* e = getenv("HIPNMF_SOS_CHUNK_512");
    return !(e && atoi(e) == 0);
  }();
  if (sizeof(real) == 8 && chunk_c == 0 && p->mode == HIPNMF_SOSFILT_SCAN && chunk_scan_ok && chunk_512_ok && nsp_c <= 4 && L <= 512LL * 41 &&
      L > 16800) {  // (below ~80 % of the 20 992 positions the idle threads cost more than the two-pass loading: 14 000 samples 2.29 vs 2.21 ms)
    chunk_c = 41;
    chunk_nt = 512;
  }
  const size_t chunk_region = chunk_c ? chunk_scan_region((size_t)L, chunk_c, chunk_nt, nsp_c, sizeof(real)) : 0;
  const bool use_chunk_scan = chunk_c > 0 && chunk_region + 64 <= (size_t)h->lds_per_block;
  // long series: blocks of 256 chunks per workgroup, a scan over the blocks (sosfilt_block_kernel); HIPNMF_SOS_BLOCK=0: the
  // sequential kernels
  static const bool block_scan_ok = [] {
    const char* e = getenv("HIPNMF_SOS_BLOCK");
    return !(e && atoi(e) == 0);
  }();
  constexpr int BLK_C = sizeof(real) == 4 ? 79 : 41;
  const long long blk_len = 256LL * BLK_C;
  const long long blk_nb = (L + blk_len - 1) / blk_len;
  const size_t blk_region = chunk_scan_region((size_t)blk_len, BLK_C, SCAN_THREADS, nsp_c, sizeof(real));
  // the sequential kernels cost a dependent chain (measured ~130 / ~70 ns per sample and series, zero-lag / causal, whatever the
  // batch up to a chip-full of 16 384 series); the block scan moves the data six / three times at ~2.2 TB/s: whichever is less
  const double blk_exact_ms = (double)T * (zero_lag ? 130e-6 : 70e-6) * std::max(1.0, (double)N / 16384.0);
  const double blk_scan_ms = 0.06 + (zero_lag ? 6.0 : 3.0) * (double)N * (double)T * (double)sizeof(real) / 2.2e9;
  static const bool block_scan_force = [] {
    const char* e = getenv("HIPNMF_SOS_BLOCK");
    return e && atoi(e) == 2;
  }();
  const bool use_block_scan = p->mode == HIPNMF_SOSFILT_SCAN && block_scan_ok && !use_chunk_scan && !scan_fits && blk_nb > 1 &&
                              (blk_scan_ms < blk_exact_ms || block_scan_force) &&
                              (long long)T * (long long)sizeof(real) < (1LL << 31) && L * (long long)sizeof(real) < (1LL << 31) &&
                              N * blk_nb < (1LL << 31) && blk_region + 64 <= (size_t)h->lds_per_block;
  const size_t o_fwd = (use_block_scan && zero_lag) ? carve(sizeof(real) * (size_t)N * (size_t)L) : 0;
  const size_t o_bend = use_block_scan ? carve(sizeof(double) * (size_t)N * (size_t)blk_nb * 2 * nsp_c) : 0;
  const size_t o_bstart = use_block_scan ? carve(sizeof(double) * (size_t)N * (size_t)blk_nb * 2 * nsp_c) : 0;
  const size_t o_ws = (zero_lag && !scan_fits && !use_chunk_scan && !use_block_scan) ? carve(ws_v2) : 0;
  const size_t o_stat = carve(sizeof(double) * (size_t)N * 3);
  // time-parallel mode: the whole extended series in the registers of one workgroup (256 chunks of at most SCAN_CMAX samples);
  // longer series take the sequential kernel
  const int C_run = scan_fits ? scan_chunk : 0;
  const bool use_scan = scan_fits;
  const size_t o_tab = (use_scan || use_chunk_scan || use_block_scan) ? carve(sizeof(double) * SCAN_TAB_DOUBLES) : 0;
  int rc = hipnmf_ensure_ws(h, std::max<size_t>(off, 256));
  if (rc) return rc;
  char* ws = static_cast<char*>(h->ws);

  SosArgs a;
  if (inplace) {
    a.x = x;
    a.bstride = p->x_batch_stride;
    a.ld = p->ldx;
  } else {
    real* xc = reinterpret_cast<real*>(ws + o_x);
    dim3 blk(32, 8);
    dim3 grd((unsigned)((T + 31) / 32), (unsigned)((m + 31) / 32), (unsigned)B);
    HIPNMF_LAUNCH(x_to_channel_major_kernel<real>, grd, blk, 0, st, x, (long long)p->x_batch_stride,
                       (long long)p->ldx, (int)p->x_layout, xc, (long long)m * T, T, (int)T, m);
    a.x = xc;
    a.bstride = (long long)m * T;
    a.ld = T;
  }
  a.ws = reinterpret_cast<double*>(ws + o_ws);
  a.y = y;
  for (int s = 0; s < SOS_MAX_SECTIONS; ++s) {
    for (int q = 0; q < 6; ++q) a.sos[s][q] = s < p->n_sections ? sos[6 * s + q] : 0.0;
    a.zi[s][0] = a.zi[s][1] = 0.0;
  }
  if (zero_lag) {
    if (zi) {
      for (int s = 0; s < p->n_sections; ++s) {
        a.zi[s][0] = zi[2 * s];
        a.zi[s][1] = zi[2 * s + 1];
      }
    } else {
      steady_state_zi(sos, p->n_sections, a.zi);
    }
  }
  a.T = (int)T;
  a.m = m;
  a.N = (int)N;
  a.edge = edge;
  a.zero_lag = zero_lag;
  a.zero_center = p->zero_center ? 1 : 0;
  a.rectify = p->rectify ? 1 : 0;
  const bool async = h->async_mode != 0;
  if (!async) HIP_TRY(hipEventRecord(h->ev0, st));
  double* stat = reinterpret_cast<double*>(ws + o_stat);
  if (use_chunk_scan) {
    const int ns = p->n_sections;
    double* tab = reinterpret_cast<double*>(ws + o_tab);
    rc = ns == 1 ? launch_chunk_scan<real, 1>(h, a, ns, chunk_c, chunk_nt, chunk_region, tab, st)
         : ns == 2 ? launch_chunk_scan<real, 2>(h, a, ns, chunk_c, chunk_nt, chunk_region, tab, st)
         : ns <= 4 ? launch_chunk_scan<real, 4>(h, a, ns, chunk_c, chunk_nt, chunk_region, tab, st)
                   : launch_chunk_scan<real, 8>(h, a, ns, chunk_c, chunk_nt, chunk_region, tab, st);
    if (rc) return rc;
    if (chunk_nt != SCAN_THREADS)
      snprintf(h->last_kernel, sizeof(h->last_kernel), "sosfilt_chunk_kernel<%s,%d,%d,%d>", sizeof(real) == 4 ? "float" : "double", nsp_c, chunk_c, chunk_nt);
    else
      snprintf(h->last_kernel, sizeof(h->last_kernel), "sosfilt_chunk_kernel<%s,%d,%d>", sizeof(real) == 4 ? "float" : "double", nsp_c, chunk_c);
    HIP_TRY(hipGetLastError());
    if (!async) {
      HIP_TRY(hipEventRecord(h->ev1, st));
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return HIPNMF_OK;
  }
  if (use_block_scan) {
    const int ns = p->n_sections;
    double* tab = reinterpret_cast<double*>(ws + o_tab);
    HIPNMF_LAUNCH(sos_stats_kernel<real>, dim3((unsigned)N), dim3(256), 0, st, a, stat);
    real* fwd = reinterpret_cast<real*>(ws + o_fwd);
    double* bend = reinterpret_cast<double*>(ws + o_bend);
    double* bstart = reinterpret_cast<double*>(ws + o_bstart);
    rc = ns == 1 ? launch_block_scan<real, 1>(h, a, ns, L, (int)blk_nb, blk_region, tab, stat, fwd, bend, bstart, y, st)
         : ns == 2 ? launch_block_scan<real, 2>(h, a, ns, L, (int)blk_nb, blk_region, tab, stat, fwd, bend, bstart, y, st)
         : ns <= 4 ? launch_block_scan<real, 4>(h, a, ns, L, (int)blk_nb, blk_region, tab, stat, fwd, bend, bstart, y, st)
                   : launch_block_scan<real, 8>(h, a, ns, L, (int)blk_nb, blk_region, tab, stat, fwd, bend, bstart, y, st);
    if (rc) return rc;
    snprintf(h->last_kernel, sizeof(h->last_kernel), "sosfilt_block_kernel<%s,%d,%d>", sizeof(real) == 4 ? "float" : "double", nsp_c, (int)BLK_C);
    HIP_TRY(hipGetLastError());
    if (!async) {
      HIP_TRY(hipEventRecord(h->ev1, st));
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return HIPNMF_OK;
  }
  if (use_scan) {
    snprintf(h->last_kernel, sizeof(h->last_kernel), "sosfilt_scan_kernel<%s,%d,%d>", sizeof(real) == 4 ? "float" : "double", nsp_c, C_run);
    const int ns = p->n_sections;
    double* tab = reinterpret_cast<double*>(ws + o_tab);
    rc = ns == 1 ? launch_scan<real, 1>(h, a, ns, C_run, tab, st)
         : ns == 2 ? launch_scan<real, 2>(h, a, ns, C_run, tab, st)
         : ns <= 4 ? launch_scan<real, 4>(h, a, ns, C_run, tab, st)
                   : launch_scan<real, 8>(h, a, ns, C_run, tab, st);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    if (!async) {
      HIP_TRY(hipEventRecord(h->ev1, st));
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return HIPNMF_OK;
  }
  snprintf(h->last_kernel, sizeof(h->last_kernel), "sosfilt2_kernel<%s>", sizeof(real) == 4 ? "float" : "double");
  HIPNMF_LAUNCH(sos_stats_kernel<real>, dim3((unsigned)N), dim3(256), 0, st, a, stat);
  {
    const int ns = p->n_sections;
    if (ns == 1)
      HIP_TRY((launch_v2<real, 1>(h, a, stat, ns, st)));
    else if (ns == 2)
      HIP_TRY((launch_v2<real, 2>(h, a, stat, ns, st)));
    else if (ns <= 4)
      HIP_TRY((launch_v2<real, 4>(h, a, stat, ns, st)));
    else
      HIP_TRY((launch_v2<real, 8>(h, a, stat, ns, st)));
  }
  HIP_TRY(hipGetLastError());
  if (!async) {
    HIP_TRY(hipEventRecord(h->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return HIPNMF_OK;
}

}  // namespace

extern "C" {
int hipnmf_sosfilt_f32(hipnmf_handle* h, const hipnmf_sosfilt_params* p, const double* sos, const double* zi,
                       const float* x, float* y) {
  return sosfilt_impl<float>(h, p, sos, zi, x, y);
}
int hipnmf_sosfilt_f64(hipnmf_handle* h, const hipnmf_sosfilt_params* p, const double* sos, const double* zi,
                       const double* x, double* y) {
  return sosfilt_impl<double>(h, p, sos, zi, x, y);
}
}
