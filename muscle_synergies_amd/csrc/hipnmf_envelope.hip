// hipnmf_envelope.hip -- C ABI of the EMG envelope preprocessing (include/hip_nmf.h, row f-1 of SURVEY.md section 8).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "envelope_chunk.hpp"
#include "envelope_kernels.hpp"
#include "hipnmf_internal.hpp"
#include "nmf_kernels.hpp"  // x_to_channel_major_kernel

using namespace hipnmf;

#ifndef HIPNMF_ENV_CHUNK_MIN_T_DEFAULT
#define HIPNMF_ENV_CHUNK_MIN_T_DEFAULT 1280  // below: emg_wave_kernel (tools/envelope_chunk_sweep.sh)
#endif

namespace {

template <typename real>
int envelope_impl(hipnmf_handle* h, const hipnmf_envelope_params* p, const real* raw, real* out) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (!p) return fail(HIPNMF_ERR_BAD_ARG, "params is NULL");
  if (p->struct_size != (int32_t)sizeof(hipnmf_envelope_params))
    return fail(HIPNMF_ERR_BAD_ARG, "hipnmf_envelope_params.struct_size = %d, library expects %d", p->struct_size,
                (int)sizeof(hipnmf_envelope_params));
  if (!raw || !out) return fail(HIPNMF_ERR_BAD_ARG, "raw and out must be non-NULL device pointers");
  if (p->batch < 1 || p->n_samples < 1 || p->n_samples > 2000000000LL || p->n_channels < 1 || p->n_channels > 65535)
    return fail(HIPNMF_ERR_BAD_ARG, "bad shape: batch=%d n_samples=%lld n_channels=%d", p->batch,
                (long long)p->n_samples, p->n_channels);
  if (p->window < 0 || p->n_out < 0) return fail(HIPNMF_ERR_BAD_ARG, "window and n_out must be >= 0");
  if (p->resample_kind < HIPNMF_RESAMPLE_LINEAR || p->resample_kind > HIPNMF_RESAMPLE_NEXT)
    return fail(HIPNMF_ERR_BAD_ARG, "bad resample_kind %d", p->resample_kind);
  if (p->reserved0 != 0) return fail(HIPNMF_ERR_BAD_ARG, "hipnmf_envelope_params.reserved0 must be 0");
  if (p->x_layout != HIPNMF_X_ROW_MAJOR && p->x_layout != HIPNMF_X_CHANNEL_MAJOR)
    return fail(HIPNMF_ERR_BAD_ARG, "bad x_layout %d", p->x_layout);
  const long long min_ld = (p->x_layout == HIPNMF_X_ROW_MAJOR) ? p->n_channels : p->n_samples;
  if (p->ldx < min_ld) return fail(HIPNMF_ERR_BAD_ARG, "ldx=%lld smaller than %lld", (long long)p->ldx, min_ld);
  HIP_TRY(hipSetDevice(h->device));
  const int B = p->batch, m = p->n_channels;
  const long long T = p->n_samples;
  hipStream_t st = h->stream;

  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const bool inplace = p->x_layout == HIPNMF_X_CHANNEL_MAJOR;
  const size_t o_x = inplace ? 0 : carve(sizeof(real) * (size_t)B * m * T);
  // fused kernel (LDS prefix sums) unless the window does not fit in LDS beside a tile, or HIPNMF_ENV_FUSED=0
  static const bool fused_ok = [] {
    const char* e = getenv("HIPNMF_ENV_FUSED");
    return !(e && atoi(e) == 0);
  }();
  const size_t fused_lds = sizeof(double) * (size_t)(ENV_TILE + p->window + 16);
  // one wave per series with a running prefix in an LDS ring (emg_wave_kernel): the default whenever there is a window
  // that fits a 4096-entry ring beside a 512-sample tile; HIPNMF_ENV_WAVE=0 falls back to the workgroup kernels
  static const bool wave_ok = [] {
    const char* e = getenv("HIPNMF_ENV_WAVE");
    return !(e && atoi(e) == 0);
  }();
  static const int wave_spl = [] {  // samples per lane and tile: 8 (default) or 4 (HIPNMF_ENV_SPL=4: smaller ring, more waves per CU)
    const char* e = getenv("HIPNMF_ENV_SPL");
    return (e && atoi(e) == 4) ? 4 : 8;
  }();
  int ring = 512;
  while (ring < 64 * wave_spl + p->window + 1 && ring < 8192) ring *= 2;  // tile + window + 1 live prefix values
  const bool wave = wave_ok && p->window >= 1 && T >= 2 && ring <= 4096 && 64 * wave_spl + p->window + 1 <= ring;
  // full-length output of a series short enough for the registers of one workgroup (emg_wg_kernel: samples and
  // outputs stay on chip, 2 instead of up to 5 sizeof(real) of traffic per sample); HIPNMF_ENV_WG=0 leaves those to
  // emg_wave_kernel
  static const bool wg_ok = [] {
    const char* e = getenv("HIPNMF_ENV_WG");
    return !(e && atoi(e) == 0);
  }();
  constexpr int WG_TILE = 64 * ENV_WG_SPL;
  int ring4 = 512;
  while (ring4 < WG_TILE + p->window + 1 && ring4 < 8192) ring4 *= 2;
  // tiles a wave walks: those before its segment (first window), its segment, those behind it (last window)
  auto wg_tiles = [&](int nw) -> long long {
    const long long seg = ((T + nw - 1) / nw + WG_TILE - 1) / WG_TILE;
    return (p->window + WG_TILE - 1) / WG_TILE + seg + ((p->window - 1) / 2 + 1 + WG_TILE - 1) / WG_TILE;
  };
  // instance: float 8 waves x 12 tiles; double 8 x 6 while the series fits, else 16 x 7 (envelope_kernels.hpp).
  // Measured against emg_wave_kernel (tools/envelope_wg_sweep.sh): float wins from ~9 K samples on (the walk before /
  // behind a wave's segment costs two extra tiles per wave, and the wave kernel's working set still fits the Infinity
  // Cache below that); double moves twice the bytes and wins from 4 K samples on with 8 waves.
  int wg_nw = 0;
  if (sizeof(real) == 4) {
    if (T >= 9216 && wg_tiles(8) <= 12) wg_nw = 8;
  } else {
    if (T >= 4096 && wg_tiles(8) <= 6) wg_nw = 8;
    else if (T > 8192 && wg_tiles(16) <= 7) wg_nw = 16;
  }
  const size_t wg_lds = sizeof(double) * ((size_t)wg_nw * (ring4 + ring4 / 8) + (size_t)wg_nw);
  const bool wg = wave && wg_ok && wg_nw > 0 && (p->n_out == 0 || p->n_out == T) && WG_TILE + p->window + 1 <= ring4 &&
                  ring4 <= 4096 && wg_lds <= (size_t)h->lds_per_block;
  // a series that fits the LDS of one workgroup (emg_chunk_kernel: a thread owns C consecutive samples, one scan per series):
  // C = 81 / 41 / 17 by length; HIPNMF_ENV_CHUNK=0 leaves those to the kernels above, HIPNMF_ENV_CHUNK_MIN_T moves the lower end
  static const bool chunk_ok = [] {
    const char* e = getenv("HIPNMF_ENV_CHUNK");
    return !(e && atoi(e) == 0);
  }();
  static const long long chunk_min_t = [] {
    const char* e = getenv("HIPNMF_ENV_CHUNK_MIN_T");
    return e ? atoll(e) : (long long)HIPNMF_ENV_CHUNK_MIN_T_DEFAULT;
  }();
  static const long long chunk_min_t_one = [] {
    const char* e = getenv("HIPNMF_ENV_CHUNK_MIN_T_ONE");
    return e ? atoll(e) : 64LL;
  }();
  const long long chunk_span = T + (p->window - 1) / 2;  // positions the window's leading edge visits
  int chunk_c = 0, chunk_nt = 256;
  static const bool chunk_wide_ok = [] {
    const char* e = getenv("HIPNMF_ENV_CHUNK_512");
    return !(e && atoi(e) == 0);
  }();
  if (chunk_ok && p->window >= 1 && T >= 2 && (T >= chunk_min_t || (T + (p->window - 1) / 2 <= 64LL * 33 && T >= chunk_min_t_one))) {
    // the shortest compiled chunk that covers the span (idle threads cost as much as busy ones); double: up to 41 (LDS)
    constexpr int chunk_sizes[] = {9, 13, 17, 25, 33, 41, 49, 57, 65, 73, 81};
    for (int c : chunk_sizes)
      if (chunk_span <= 256LL * c && (sizeof(real) == 4 || c <= 41)) {
        chunk_c = c;
        break;
      }
    // float, the longest series: eight waves with half the chunk each (four waves per SIMD instead of two hide a little more
    // of the slide's latencies: 0.59 -> 0.55 / 0.337 -> 0.326 ms at 20 000 samples; no gain or a loss below ~17 000)
    if (sizeof(real) == 4 && chunk_c == 81 && chunk_wide_ok) {
      chunk_nt = 512;
      chunk_c = 41;
    }
    // short series: ONE wave per series (no workgroup-wide step is left, 4 KB of LDS: many series per CU) with the shortest chunk
    // that covers 64 of them; HIPNMF_ENV_CHUNK_64=0 keeps the four-wave instances / emg_wave_kernel
    static const bool chunk_one_ok = [] {
      const char* e = getenv("HIPNMF_ENV_CHUNK_64");
      return !(e && atoi(e) == 0);
    }();
    if (chunk_one_ok && chunk_span <= 64LL * 33) {
      constexpr int one_sizes[] = {5, 9, 13, 17, 25, 33};
      for (int c : one_sizes)
        if (chunk_span <= 64LL * c) {
          chunk_c = c;
          chunk_nt = 64;
          break;
        }
    }
    if (chunk_nt != 64 && T < chunk_min_t) chunk_c = 0;  // (below the four-wave instances' range only the one-wave ones apply)
    // double, 10 497 .. 20 992 positions: the same eight-wave instance with the whole CU's LDS for one series (20 000 doubles +
    // a window of up to ~280 samples fit 160 KB; anything more falls through to emg_wg_kernel by the LDS check below)
    if (sizeof(real) == 8 && chunk_c == 0 && chunk_wide_ok && chunk_span <= 512LL * 41) {
      chunk_nt = 512;
      chunk_c = 41;
    }
  }
  const int chunk_nact = chunk_c ? (int)((chunk_span + chunk_c - 1) / chunk_c) : 0;  // (<= chunk_nt)
  // 16-byte pieces (a quarter of the memory instructions) when every row of the canonical layout and of the output is aligned and
  // a whole number of pieces; HIPNMF_ENV_CHUNK_VEC=0 keeps the dword form
  static const bool chunk_vec_ok = [] {
    const char* e = getenv("HIPNMF_ENV_CHUNK_VEC");
    return !(e && atoi(e) == 0);
  }();
  constexpr int CV_ = 16 / (int)sizeof(real);
  const bool chunk_full = p->n_out == 0 || p->n_out == T;
  const bool chunk_vec = chunk_vec_ok && chunk_c > 0 && T % CV_ == 0 && (reinterpret_cast<uintptr_t>(out) % 16) == 0 &&
                         (!inplace || ((reinterpret_cast<uintptr_t>(raw) % 16) == 0 && p->ldx % CV_ == 0 && p->x_batch_stride % CV_ == 0));
  const size_t chunk_pad = chunk_vec ? ((size_t)p->window + CV_ - 1) / CV_ * CV_ : (size_t)p->window;
  const size_t chunk_ns = chunk_vec ? ((size_t)chunk_nact * chunk_c + CV_ - 1) / CV_ * CV_ : (size_t)chunk_nact * chunk_c;
  const size_t chunk_lds = ((sizeof(real) * (chunk_pad + chunk_ns) + 15) & ~(size_t)15) + CHUNK_RED * sizeof(double);
  // time-normalised output: only while two workgroups share a CU (measured: one alone loses to emg_wave_kernel, which overlaps
  // its single read stream across 16 waves per CU; the full-length output wins either way, its write stream is what counts)
  const bool chunk = chunk_c > 0 && chunk_lds <= (size_t)h->lds_per_block &&
                     ((p->n_out == 0 || p->n_out == T) || 2 * chunk_lds <= (size_t)h->lds_per_block);
  const bool fused = wave || (fused_ok && fused_lds <= 96 * 1024);
  const bool resample_tab = (wave || chunk) && p->n_out > 0 && p->n_out != T;
  const size_t o_ti = resample_tab ? carve(sizeof(int) * (size_t)p->n_out) : 0;
  const size_t o_tw = resample_tab ? carve(sizeof(double) * (size_t)p->n_out) : 0;
  const size_t o_ps = fused ? 0 : carve(sizeof(double) * (size_t)B * m * (T + 1));
  const size_t o_st = fused ? 0 : carve(sizeof(double) * (size_t)B * m * 2);
  int rc = hipnmf_ensure_ws(h, std::max<size_t>(off, 256));
  if (rc) return rc;
  char* ws = static_cast<char*>(h->ws);

  EnvArgs a;
  if (inplace) {
    a.raw = raw;
    a.bstride = p->x_batch_stride;
    a.ld = p->ldx;
  } else {
    real* xc = reinterpret_cast<real*>(ws + o_x);
    dim3 blk(32, 8);
    dim3 grd((unsigned)((T + 31) / 32), (unsigned)((m + 31) / 32), (unsigned)B);
    HIPNMF_LAUNCH(x_to_channel_major_kernel<real>, grd, blk, 0, st, raw, (long long)p->x_batch_stride,
                       (long long)p->ldx, (int)p->x_layout, xc, (long long)m * T, T, (int)T, m);
    a.raw = xc;
    a.bstride = (long long)m * T;
    a.ld = T;
  }
  a.prefix = reinterpret_cast<double*>(ws + o_ps);
  a.chan_stat = reinterpret_cast<double*>(ws + o_st);
  a.out = out;
  a.tab_i0 = reinterpret_cast<const int*>(ws + o_ti);
  a.tab_w = reinterpret_cast<const double*>(ws + o_tw);
  a.T = (int)T;
  a.m = m;
  a.window = p->window;
  a.zero_center = p->zero_center ? 1 : 0;
  a.n_out = p->n_out;
  a.resample_kind = p->resample_kind;
  a.normalize = p->normalize ? 1 : 0;
  const bool async = h->async_mode != 0;
  if (!async) HIP_TRY(hipEventRecord(h->ev0, st));
  if (resample_tab)
    HIPNMF_LAUNCH(env_resample_table_kernel, dim3((unsigned)((p->n_out + 255) / 256)), dim3(256), 0, st, (int)T, (int)p->n_out, (int)p->resample_kind,
                       reinterpret_cast<int*>(ws + o_ti), reinterpret_cast<double*>(ws + o_tw));
  if (chunk) {
    auto launch_chunk = [&](auto kern, const char* name) -> int {
      if (chunk_lds > 48 * 1024)
        if (int arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(kern))) return arc;
      HIPNMF_LAUNCH(kern, dim3(m, B), dim3(chunk_nt), chunk_lds, st, a, chunk_nact);
      snprintf(h->last_kernel, sizeof(h->last_kernel), "%s", name);
      return HIPNMF_OK;
    };
    int rcc = HIPNMF_ERR_UNSUPPORTED;
    char name[64];
    snprintf(name, sizeof(name), "emg_chunk_kernel<%s,%d,%d>%s", sizeof(real) == 4 ? "float" : "double", chunk_c, chunk_nt, chunk_vec ? "[vec]" : "");
#define HIPNMF_CHUNK_CASE(C_, NT_) \
  case C_:                         \
    rcc = chunk_vec ? launch_chunk(emg_chunk_kernel<real, C_, NT_, true>, name) : launch_chunk(emg_chunk_kernel<real, C_, NT_, false>, name); \
    break;
    if (chunk_nt == 256) {
      switch (chunk_c) {
        HIPNMF_CHUNK_CASE(9, 256) HIPNMF_CHUNK_CASE(13, 256) HIPNMF_CHUNK_CASE(17, 256) HIPNMF_CHUNK_CASE(25, 256) HIPNMF_CHUNK_CASE(33, 256)
        HIPNMF_CHUNK_CASE(41, 256)
        default:
          if constexpr (sizeof(real) == 4) {
            switch (chunk_c) { HIPNMF_CHUNK_CASE(49, 256) HIPNMF_CHUNK_CASE(57, 256) HIPNMF_CHUNK_CASE(65, 256) HIPNMF_CHUNK_CASE(73, 256) HIPNMF_CHUNK_CASE(81, 256) }
          }
      }
    } else if (chunk_nt == 64) {
      switch (chunk_c) {
        HIPNMF_CHUNK_CASE(5, 64) HIPNMF_CHUNK_CASE(9, 64) HIPNMF_CHUNK_CASE(13, 64) HIPNMF_CHUNK_CASE(17, 64) HIPNMF_CHUNK_CASE(25, 64) HIPNMF_CHUNK_CASE(33, 64)
      }
    } else {
      switch (chunk_c) { HIPNMF_CHUNK_CASE(41, 512) }
    }
#undef HIPNMF_CHUNK_CASE
    if (rcc) return rcc;
  } else if (wg) {
    snprintf(h->last_kernel, sizeof(h->last_kernel), "emg_wg_kernel");
    auto launch_wg = [&](auto kern) -> int {
      if (wg_lds > 48 * 1024)
        if (int arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(kern))) return arc;
      HIPNMF_LAUNCH(kern, dim3(m, B), dim3(64 * wg_nw), wg_lds, st, a, ring4);
      return HIPNMF_OK;
    };
    int rcw;
    if constexpr (sizeof(real) == 4)
      rcw = launch_wg(emg_wg_kernel<real, 8, 12>);
    else
      rcw = wg_nw == 8 ? launch_wg(emg_wg_kernel<real, 8, 6>) : launch_wg(emg_wg_kernel<real, 16, 7>);
    if (rcw) return rcw;
  } else if (wave) {
    snprintf(h->last_kernel, sizeof(h->last_kernel), "emg_wave_kernel");
    const size_t lds = sizeof(double) * (size_t)(ring + ring / 8);
    if (wave_spl == 4)
      HIPNMF_LAUNCH((emg_wave_kernel<real, 4>), dim3(m, B), dim3(64), lds, st, a, ring);
    else
      HIPNMF_LAUNCH((emg_wave_kernel<real, 8>), dim3(m, B), dim3(64), lds, st, a, ring);
  } else if (fused) {
    snprintf(h->last_kernel, sizeof(h->last_kernel), "emg_fused_kernel");
    if (fused_lds > 48 * 1024)
      if (int arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(emg_fused_kernel<real>))) return arc;
    HIPNMF_LAUNCH(emg_fused_kernel<real>, dim3(m, B), dim3(256), fused_lds, st, a);
  } else {
    snprintf(h->last_kernel, sizeof(h->last_kernel), "emg_prefix_kernel+emg_output_kernel");
    HIPNMF_LAUNCH(emg_prefix_kernel<real>, dim3(m, B), dim3(256), 0, st, a);
    HIPNMF_LAUNCH(emg_output_kernel<real>, dim3(m, B), dim3(256), 0, st, a);
  }
  HIP_TRY(hipGetLastError());
  if (!async) {
    HIP_TRY(hipEventRecord(h->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return HIPNMF_OK;
}

// ---- banded resampling operator (round 6: the spline kinds of time_normalize) -----------------------------------------------------
// interp1d(kind='quadratic' / 'cubic') is make_interp_spline (scipy/interpolate/_interpolate.py:397, _bsplines.py:1363-1580): a banded
// collocation solve over ALL samples of a channel, then an evaluation.  Both are linear in the samples and depend on (T, n_out, kind)
// only, and the solve's influence decays geometrically (cubic: 0.268 per sample, quadratic: 0.172): every output row is a short
// window of weights over the input.  The host builds that operator once per shape from scipy's own design matrices; this kernel
// applies it to every channel of every recording: out[b][j][r] = sum_i weights[r][i] * x[b][first[r] + i][j], fp64 accumulation.
template <typename real>
__global__ void __launch_bounds__(256) resample_weights_kernel(const real* __restrict__ x, long long bstride, long long ldx, int x_layout,
                                                                const int* __restrict__ first, const double* __restrict__ weights, int taps,
                                                                int n_out, int m, long long total, real* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // ((b * m) + j) * n_out + r: r fastest (coalesced stores)
  if (e >= total) return;
  const int r = (int)(e % n_out);
  const long long bj = e / n_out;
  const int j = (int)(bj % m);
  const long long b = bj / m;
  const real* __restrict__ xb = x + b * bstride;
  const long long step = x_layout == HIPNMF_X_ROW_MAJOR ? ldx : 1;
  const real* __restrict__ src = xb + (x_layout == HIPNMF_X_ROW_MAJOR ? (long long)j : (long long)j * ldx) + (long long)first[r] * step;
  const double* __restrict__ w = weights + (long long)r * taps;
  double acc0 = 0.0, acc1 = 0.0;  // two chains; a fixed order
  int i = 0;
  for (; i + 1 < taps; i += 2) {
    acc0 = __builtin_fma(w[i], (double)src[(long long)i * step], acc0);
    acc1 = __builtin_fma(w[i + 1], (double)src[(long long)(i + 1) * step], acc1);
  }
  if (i < taps) acc0 = __builtin_fma(w[i], (double)src[(long long)i * step], acc0);
  out[e] = (real)(acc0 + acc1);
}

template <typename real>
int resample_weights_impl(hipnmf_handle* h, const hipnmf_envelope_params* p, const real* x, const int32_t* first, const double* weights,
                          int32_t taps, real* out) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (!p || p->struct_size != (int32_t)sizeof(hipnmf_envelope_params))
    return fail(HIPNMF_ERR_BAD_ARG, "hipnmf_envelope_params.struct_size mismatch");
  if (!x || !first || !weights || !out) return fail(HIPNMF_ERR_BAD_ARG, "x, first, weights and out must be non-NULL device pointers");
  if (p->batch < 1 || p->n_samples < 1 || p->n_channels < 1 || p->n_out < 1 || taps < 1 || taps > p->n_samples)
    return fail(HIPNMF_ERR_BAD_ARG, "bad resampling problem (batch=%d, n_samples=%lld, n_channels=%d, n_out=%d, taps=%d)", p->batch,
                (long long)p->n_samples, p->n_channels, p->n_out, taps);
  if (p->x_layout != HIPNMF_X_ROW_MAJOR && p->x_layout != HIPNMF_X_CHANNEL_MAJOR) return fail(HIPNMF_ERR_BAD_ARG, "bad x_layout %d", p->x_layout);
  if (p->ldx < (p->x_layout == HIPNMF_X_ROW_MAJOR ? p->n_channels : p->n_samples)) return fail(HIPNMF_ERR_BAD_ARG, "ldx too small");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const long long total = (long long)p->batch * p->n_channels * p->n_out;
  const long long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(HIPNMF_ERR_UNSUPPORTED, "too many output samples for one launch");
  const bool async = h->async_mode != 0;
  if (!async) HIP_TRY(hipEventRecord(h->ev0, st));
  HIPNMF_LAUNCH(resample_weights_kernel<real>, dim3((unsigned)blocks), dim3(256), 0, st, x, (long long)p->x_batch_stride, (long long)p->ldx,
                     (int)p->x_layout, first, weights, (int)taps, (int)p->n_out, (int)p->n_channels, total, out);
  snprintf(h->last_kernel, sizeof(h->last_kernel), "resample_weights_kernel<%s>", sizeof(real) == 4 ? "float" : "double");
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
int hipnmf_resample_weights_f32(hipnmf_handle* h, const hipnmf_envelope_params* p, const float* x, const int32_t* first, const double* weights,
                                int32_t taps, float* out) {
  return resample_weights_impl<float>(h, p, x, first, weights, taps, out);
}
int hipnmf_resample_weights_f64(hipnmf_handle* h, const hipnmf_envelope_params* p, const double* x, const int32_t* first, const double* weights,
                                int32_t taps, double* out) {
  return resample_weights_impl<double>(h, p, x, first, weights, taps, out);
}
int hipnmf_emg_envelope_f32(hipnmf_handle* h, const hipnmf_envelope_params* p, const float* raw, float* out) {
  return envelope_impl<float>(h, p, raw, out);
}
int hipnmf_emg_envelope_f64(hipnmf_handle* h, const hipnmf_envelope_params* p, const double* raw, double* out) {
  return envelope_impl<double>(h, p, raw, out);
}
}
