// nmf_kernels.hpp -- CDNA4 (gfx950) kernels of the NMF multiplicative-update solver.
//
// Arithmetic replaced: sklearn/decomposition/_nmf.py (1.7.2) _multiplicative_update_w (:526-631),
// _multiplicative_update_h (:634-728), _beta_divergence (:85-134), loop + stop rule (:731-893), reached from
// the reference at src/muscle_synergies/analysis.py:862-863.  sklearn notation: X (T x m) ~ W (T x k) H (k x m).
//
// Mapping (wave64, no LDS in the streaming loop):
//   * a wave owns 64 consecutive rows (time samples) per step; lane l owns row  wbase + l  of W;
//   * G consecutive lanes form a group that shares G rows of X: lane g of the group holds CH channels
//     (columns g*CH .. g*CH+CH-1) of all G rows, loaded as CH vector loads of G elements from the
//     channel-major X (coalesced, 16 B per lane for fp32 G=4);
//   * X H^T : each lane forms partial dot products over its CH channels for the G rows, then a
//     reduce-scatter over the G lanes (DPP quad permutes, no LDS) leaves lane g with row g's numerator;
//   * W (H H^T), the division and the W update are row-per-lane;
//   * W^T X : the updated row is broadcast inside the group (DPP) and each lane accumulates K x CH sums
//     for its own channels; W^T W is accumulated row-per-lane (upper triangle);
//   * the T-long reductions finish with a butterfly over the wave and a fixed-order sum over waves/slices,
//     so results are bitwise reproducible for a given launch geometry.
//
// Kernels (all instantiated per (real, G, CH, K) in inst_*.hip through nmf_inst.hpp):
//   fit_persistent_kernel<.., LOSS>  one workgroup per matrix, every iteration inside the kernel, most rows of W
//                                    resident in LDS (the batch / headline path; LOSS = 1: Kullback-Leibler)
//   fit_coop_kernel                  S co-resident workgroups per matrix with a fence-free exchange of the
//                                    partial sums per iteration (few long matrices)
//   slice_pass / reduce_slices / hupdate / slice_resid / resid_finalize
//                                    one launch per phase over row slices (very long T; also the building blocks
//                                    of the multi-GPU time-sharded solver, where an all-reduce sits between them)
//   x_to_channel_major_kernel, w_convert_kernel   layout conversions, once per fit
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace hipnmf {

constexpr int WAVE = 64;
#ifndef HIPNMF_MAXNT
#define HIPNMF_MAXNT 512
#endif
#ifndef HIPNMF_WAVES_PER_EU
#define HIPNMF_OCC
#else
#define HIPNMF_OCC __attribute__((amdgpu_waves_per_eu(HIPNMF_WAVES_PER_EU)))
#endif

// EPSILON = np.finfo(np.float32).eps for fp32 *and* fp64 (_nmf.py:39)
template <typename real>
__device__ __forceinline__ real eps_val() {
  return (real)1.1920928955078125e-07;
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// Quotient of the W update.  fp32: v_rcp_f32 + one Newton step on the quotient (<= 1 ulp; 4 VALU ops instead
// of the ~10 of the IEEE expansion, 5 quotients per row); tiny denominators, whose reciprocal would overflow,
// take the exact path.  fp64 always divides exactly.
__device__ __forceinline__ float fast_div(float n, float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  const float q = n * r;
  const float e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
// x / wh of the Kullback-Leibler updates (_nmf.py:574-577, 660-663), 32 quotients per row of a 16-channel matrix and iteration:
// the bare reciprocal (1 ulp) times x, without the correction step of fast_div -- the quotient feeds sums of 16 / of all rows
// and the tolerance of the path is 1e-5 (north_star); wh >= EPSILON, so the reciprocal cannot overflow.  -DHIPNMF_KL_EXACT_Q
// restores fast_div.
// max(wh, EPSILON) of the Kullback-Leibler quotients (_nmf.py:574-575: WH_safe_X[WH_safe_X < EPSILON] = EPSILON) as ONE v_max
// (the comparison + select the ternary compiles to is two instructions, 32 times per row).  A NaN in wh would come out as
// EPSILON here instead of NaN -- it still reaches the factors through W and H themselves, which the update multiplies.
__device__ __forceinline__ float kl_floor(float wh) { return __builtin_fmaxf(wh, eps_val<float>()); }
__device__ __forceinline__ double kl_floor(double wh) { return __builtin_fmax(wh, eps_val<double>()); }
__device__ __forceinline__ float kl_quot(float x, float wh) {
#ifdef HIPNMF_KL_EXACT_Q
  return fast_div(x, wh);
#else
  return x * __builtin_amdgcn_rcpf(wh);
#endif
}
template <int K>
__device__ __forceinline__ void quotients(const float (&n)[K], const float (&d)[K], float (&q)[K]) {
#ifdef HIPNMF_EXACT_DIV
#pragma unroll
  for (int c = 0; c < K; ++c) q[c] = n[c] / d[c];
#else
  bool tiny = false;
#pragma unroll
  for (int c = 0; c < K; ++c) tiny = tiny || (d[c] < 1e-37f);
  if (__builtin_expect(__any(tiny), 0)) {  // one wave-uniform, practically never taken branch per tile
#pragma unroll
    for (int c = 0; c < K; ++c) q[c] = n[c] / d[c];
  } else {
#pragma unroll
    for (int c = 0; c < K; ++c) q[c] = fast_div(n[c], d[c]);
  }
#endif
}
template <int K>
__device__ __forceinline__ void quotients(const double (&n)[K], const double (&d)[K], double (&q)[K]) {
#pragma unroll
  for (int c = 0; c < K; ++c) q[c] = n[c] / d[c];
}
__device__ __forceinline__ float sqrt_(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ double sqrt_(double a) { return __builtin_sqrt(a); }

// ------------------------------------------------------------------------------------------------
// compile-time loop
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ------------------------------------------------------------------------------------------------
// cross-lane primitives
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  int i = __builtin_bit_cast(int, v);
#ifdef HIPNMF_SWIZZLE
  // quad permutes through the LDS crossbar (ds_swizzle, QDMode) instead of the VALU's DPP path
  if constexpr (CTRL < 0x100)
    i = __builtin_amdgcn_ds_swizzle(i, 0x8000 | CTRL);
  else
#endif
    i = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(float, i);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  long long ll = __builtin_bit_cast(long long, v);
  int lo = (int)(ll & 0xffffffffLL), hi = (int)(ll >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
  ll = ((long long)hi << 32) | (unsigned int)lo;
  return __builtin_bit_cast(double, ll);
}

// value of lane (l ^ H); H in {1, 2} stays inside a quad -> DPP quad_perm, else ds_bpermute
template <int H, typename real>
__device__ __forceinline__ real xor_lane(real v) {
  if constexpr (H == 1)
    return dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
  else if constexpr (H == 2)
    return dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
  else
    return __shfl_xor(v, H, WAVE);
}

// value held by lane R of this lane's G-lane group
template <int G, int R, typename real>
__device__ __forceinline__ real group_bcast(real v) {
  if constexpr (G == 1)
    return v;
  else if constexpr (G == 2)
    return dpp_mov<(R == 0 ? 0xA0 : 0xF5)>(v);  // quad_perm [0,0,2,2] / [1,1,3,3]
  else if constexpr (G == 4)
    return dpp_mov<R * 0x55>(v);  // quad_perm [R,R,R,R]
  else
    return __shfl(v, R, G);
}

template <typename real>
__device__ __forceinline__ real uniform(real v) {  // wave-uniform value -> SGPR
  if constexpr (sizeof(real) == 4) {
    return __builtin_bit_cast(real, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
  } else {
    long long ll = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readfirstlane((int)(ll & 0xffffffffLL));
    int hi = __builtin_amdgcn_readfirstlane((int)(ll >> 32));
    ll = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(real, ll);
  }
}

// ------------------------------------------------------------------------------------------------
// Buffer-resource (SRD) memory access: 32-bit per-lane offsets, base in SGPRs, and hardware range
// checking -- a lane whose offset is >= num_records reads 0 / drops its store, so ragged tails and
// padded channels need no branches around the loads (branches serialise them: one vmcnt(0) per load).
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned OOB = 0x80000000u;  // > any valid offset: matrices are limited to < 2 GiB each

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// AUX: cache policy bits of the instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1); 2 = non-temporal stream
template <typename real, int N, int AUX = 0>
__device__ __forceinline__ void buf_load(rsrc_t r, unsigned voff, unsigned soff, real (&out)[N]) {
  constexpr int BYTES = (int)sizeof(real) * N;
  static_assert(BYTES == 4 || BYTES == 8 || BYTES == 12 || BYTES == 16 || BYTES == 32, "unsupported vector width");
  if constexpr (BYTES == 12) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, AUX);
    __builtin_memcpy(&out, &v, 12);
  } else if constexpr (BYTES == 4) {
    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX);
    __builtin_memcpy(&out, &v, 4);
  } else if constexpr (BYTES == 8) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    __builtin_memcpy(&out, &v, 8);
  } else if constexpr (BYTES == 16) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    __builtin_memcpy(&out, &v, 16);
  } else {
    const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u, soff, AUX);
    __builtin_memcpy(&out, &v0, 16);
    __builtin_memcpy(reinterpret_cast<char*>(&out) + 16, &v1, 16);
  }
}

template <typename real>
__device__ __forceinline__ void buf_store(rsrc_t r, unsigned voff, unsigned soff, real v) {
  if constexpr (sizeof(real) == 4) {
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    __builtin_amdgcn_raw_buffer_store_b32(u, r, voff, soff, 0);
  } else {
    using u32x2 = unsigned int __attribute__((ext_vector_type(2)));
    u32x2 u;
    __builtin_memcpy(&u, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(u, r, voff, soff, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// Largest workgroup a kernel instance is compiled for.  512 threads = 2 waves/SIMD caps the allocation at 256
// VGPRs; instances whose live state cannot fit (fp64, m > 16 or large k) are compiled for 256 threads
// (1 wave/SIMD, up to 512 VGPRs) instead of spilling to scratch.
// The row-per-lane instance (G == 1, CH == 16: fp32, 9..16 channels, k <= 5) streams a ROW-MAJOR X: the lane's own
// row is 64 contiguous bytes (four 16-byte loads), nothing crosses lanes in the row loop, H and the K x 16
// accumulators live in VGPRs, and no tile is prefetched in software (two waves per SIMD cover the latency;
// 160 VGPRs of H + sums leave no room for a second tile).  Measured 9.6 vs 8.5 M matrix-it/s for the (G=4, CH=4)
// mapping on 2048 x (16 x 10 000), k = 5 (profiles/README.md).
template <int G, int CH>
constexpr bool x_row_major() {
  return G == 1 && (CH == 16 || CH == 8);  // (1, 8): 7..8 channels; fp32 any k <= 8, fp64 k <= 4
}
// Row-per-lane instances with k >= 4: the last HIPNMF_ROW_HLDS rows of H are re-read from LDS every tile (wave-
// uniform 16-byte broadcast reads, issued before the tile's X is waited for) instead of living in VGPRs; the 32
// registers this frees pay for a second tile in flight (HIPNMF_PF_ROW).
// Measured (B = 2048 x 200 iterations): all of H in VGPRs and one tile in flight 9.7 M matrix-it/s; 2 / 3 / 4 rows
// from LDS with two tiles in flight 7.4 / 8.5 / 7.8 M (29 / 6 / 0 spilled VGPRs) -- the per-tile LDS reads cost more
// than the second tile buys, so both knobs default to "off".
#ifndef HIPNMF_ROW_HLDS
#define HIPNMF_ROW_HLDS 0
#endif
#ifndef HIPNMF_PF_ROW
#define HIPNMF_PF_ROW 1
#endif
// Experiment (-DHIPNMF_ROW_TOUCH=n): row-major instances have no register to spare for a second tile, so the tile n
// steps ahead can be pulled towards the CU by a one-dword-per-lane load whose result is only consumed a tile later
// (64 lanes x 64-byte rows = every cache line of the tile).  Measured: n = 0 / 2 / 4 -> 9.9 / 9.4 / 7.8 M matrix-it/s:
// with ~7.5 TB/s of L2-side traffic the memory system is the wall now, extra requests only add to it.  Off.
#ifndef HIPNMF_ROW_TOUCH
#define HIPNMF_ROW_TOUCH 0
#endif
template <int G, int CH, int K>
constexpr int h_lds_rows() {
  return (x_row_major<G, CH>() && K >= 4 && (HIPNMF_ROW_HLDS) > 0) ? (HIPNMF_ROW_HLDS) : 0;
}
// H broadcast from LDS per tile instead of K x CH VGPRs per lane: the Frobenius kernels of the float64 (G=4, CH=8)
// mapping (17..32 channels), where H alone would take 2 K CH = 80..128 registers and the instance spilled up to
// 1.8 KB per lane (tools/quick_bench.py, 32 channels: k = 5 1.18 -> 1.53, k = 8 0.21 -> 0.39 M matrix-it/s).  The KL
// iteration (LOSS = 1) keeps H in registers.  -DHIPNMF_HLDS_CH8 extends it to every CH = 8 channel-major instance.
template <typename real, int G, int CH, int LOSS = 0>
constexpr bool h_in_lds() {
  if (LOSS != 0) return false;
#ifdef HIPNMF_HLDS_CH8
  if (CH >= 8 && !x_row_major<G, CH>()) return true;
#endif
  return sizeof(real) == 8 && G == 4 && CH == 8;
}

template <typename real, int G, int CH, int K>
constexpr int max_threads() {
  if constexpr (x_row_major<G, CH>() && K * CH * (int)(sizeof(real) / 4) <= (sizeof(real) == 4 ? 80 : 64))
    return HIPNMF_MAXNT;  // H + sums fit two waves per SIMD (fp32: k*CH <= 80; fp64 (1,8): k <= 4)
  constexpr int words = (int)(sizeof(real) / 4);
  constexpr int est = words * ((h_in_lds<real, G, CH>() ? 1 : 2) * K * CH + K * (K + 1) / 2 + 3 * G * CH + G * K + 40 +
                               (h_in_lds<real, G, CH>() ? 24 : 0));
  return est > 215 ? 256 : (HIPNMF_MAXNT);
}

template <typename real, int G, int CH, int K>
struct Cfg {
  static constexpr int MP = G * CH;            // padded channel count handled by a lane group
  static constexpr int NB = K * (K + 1) / 2;   // upper triangle of W^T W
  static constexpr int NACC = K * MP + NB;     // floats per wave partial
  // LDS record per wave: the update pass stores NACC sums, the residual pass 2*MP (sse | xsq)
  static constexpr int NREC = NACC > 2 * MP + 1 ? NACC : 2 * MP + 1;  // +1: KL divergence partial
};

// kernel arguments (canonical layouts: X channel-major with ldx % G == 0, W component-major)
template <typename real>
struct SolveArgs {
  const real* X;
  long long x_bstride, ldx;
  real* W;
  long long w_bstride, ldw;
  real* H;            // [B][k][m]
  real* part;         // [B][S][NACC]    per-slice partial sums (multi-slice path / shard path)
  real* sums;         // [B][k*m + k*k]  summed W^T X | W^T W (shard path), or nullptr
  real* colpart;      // [B][S][2*MP]    per-slice sse / xsq partials (multi-slice path)
  real* err_out;      // [B] or nullptr
  int* n_iter_out;    // [B] or nullptr
  real* sse_col_out;  // [B][m] or nullptr
  real* xsq_col_out;  // [B][m] or nullptr
  real* state;        // [B][8] err0, prev, err, done, checks-so-far   (multi-slice path stop rule)
  int T, m, max_iter, check_every, update_h, S, rows_per_slice, it;
  int lds_rows;       // persistent kernel: rows [0, lds_rows) of W live in LDS for the whole fit
  const long long* ragged;  // [B][4] = {T_b, X offset, leading dimension, W offset} (elements) or nullptr
  unsigned* sync;     // cooperative kernel: [B] arrival counters (zeroed before the launch) followed by one abort flag;
                      // sliced path with fuse_h: [B] arrival counters (the last slice to finish updates H)
  int coop_xcd;       // fit_coop_kernel<..., XCD = true> (the S workgroups of a matrix picked on ONE XCD, exchange through its L2): 2 = test hook
                      // (sync then continues with an 'updates began' word, [B][8] tickets, [B] target XCD + 1, [B][32] generation flags 32 words apart)
  int fuse_h;         // slice_pass_kernel: 1 = the last workgroup of a matrix sums the records and updates H itself
  real tol, l1w, l2w, l1h, l2h;
  unsigned gen_base;  // cooperative kernel: offset of the exchange's generation numbers (0; a test hook moves it next to the 32-bit wrap)
};

template <typename real, int G, int CH, int K>
struct RowTile {
  real x[CH][G];
  real w[K];
};

// Per-lane addressing state of one matrix (loop invariant): X is channel-major, W component-major.
//   X element (row t, channel j)  at byte  (j*ldx + t) * sizeof(real)
//   W element (row t, component c) at byte (c*ldw + t) * sizeof(real)
// The wave-uniform row base goes into the SGPR offset, the lane part into the VGPR offset.
template <typename real, int G, int CH, int K>
struct MatAddr {
  rsrc_t xr, wr;
  unsigned xoff[CH];  // (channel*ldx + lane's group row) * sizeof(real), or OOB for padded channels
  unsigned xrow_b;    // row-major instances: bytes between consecutive rows of X
  unsigned woff;      // lane * sizeof(real)
  unsigned ldw_b;     // ldw * sizeof(real)
  int T, lane, g;
  real* lds_w;        // [K][lds_rows] component-major W cache in LDS (persistent kernel), or nullptr
  int lds_rows;       // stride of the cache (rows)
  int lds_used;       // rows of this matrix that live in the cache
  const real* h_lds;  // LDS copy of H ([K][MP]), read per tile by the h_in_lds instances
  __device__ __forceinline__ MatAddr(const real* Xb, long long ldx, const real* Wb, long long ldw, int T_, int m,
                                     real* lds_w_ = nullptr, int lds_rows_ = 0) {
    lds_w = lds_w_;
    lds_rows = lds_rows_;
    lds_used = lds_rows_;
    h_lds = nullptr;
    lane = threadIdx.x & (WAVE - 1);
    g = lane % G;
    T = T_;
    xr = make_rsrc(Xb, (unsigned)((long long)m * ldx * (long long)sizeof(real)));
    wr = make_rsrc(Wb, (unsigned)((long long)K * ldw * (long long)sizeof(real)));
    xrow_b = 0;
    if constexpr (x_row_major<G, CH>()) {  // X row-major, row stride ldx (a multiple of 4 elements, >= CH)
      xr = make_rsrc(Xb, (unsigned)((long long)(T_ + 64) * ldx * (long long)sizeof(real)));
      xrow_b = (unsigned)(ldx * (long long)sizeof(real));
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) xoff[cc] = (unsigned)lane * xrow_b + (unsigned)(cc * (int)sizeof(real));
    } else
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      const int j = g * CH + cc;
      xoff[cc] = (j < m) ? (unsigned)(((long long)j * ldx + (lane - g)) * (long long)sizeof(real)) : OOB;
    }
    woff = (unsigned)(lane * (int)sizeof(real));
    ldw_b = (unsigned)(ldw * (long long)sizeof(real));
  }
};

// tile of the wave-step starting at row wbase (wave-uniform); rows >= T read as zero.
// WLDS: this step's rows of W are resident in LDS (each row is only ever touched by its owner lane,
// so no barrier is needed around these accesses).
// Cache policy of the row-major X stream (see buf_load).  Experiment -DHIPNMF_X_AUX=2: stream X non-temporal so that
// the rows of W that do not fit in LDS (re-read and re-written every iteration) stay in the XCD's L2.
#ifndef HIPNMF_X_AUX
#define HIPNMF_X_AUX 0
#endif
template <typename real, int G, int CH, int K, bool WLDS>
__device__ __forceinline__ void load_tile(RowTile<real, G, CH, K>& t, const MatAddr<real, G, CH, K>& ma, int wbase,
                                          bool in_range) {
  const unsigned sbase = (unsigned)wbase * (unsigned)sizeof(real);
  const bool grp_ok = in_range && (wbase + (ma.lane - ma.g) < ma.T);
  if constexpr (x_row_major<G, CH>()) {  // the lane's own row: CH consecutive values, 16 bytes per load
    constexpr int V = 16 / (int)sizeof(real);
    const unsigned srow = (unsigned)wbase * ma.xrow_b;
#pragma unroll
    for (int q = 0; q < CH / V; ++q)
      buf_load<real, V, (HIPNMF_X_AUX)>(ma.xr, grp_ok ? ma.xoff[q * V] : OOB, srow,
                                        *reinterpret_cast<real(*)[V]>(&t.x[q * V][0]));
  } else
#pragma unroll
  for (int cc = 0; cc < CH; ++cc) buf_load<real, G>(ma.xr, grp_ok ? ma.xoff[cc] : OOB, sbase, t.x[cc]);
  if constexpr (WLDS) {
    const real* p = ma.lds_w + (in_range ? wbase : 0) + ma.lane;
#pragma unroll
    for (int c = 0; c < K; ++c) t.w[c] = p[c * ma.lds_rows];
  } else {
    const bool row_ok = in_range && (wbase + ma.lane < ma.T);
    const unsigned wv = row_ok ? ma.woff : OOB;
#pragma unroll
    for (int c = 0; c < K; ++c) {
      real tmp[1];
      buf_load<real, 1>(ma.wr, wv, sbase + (unsigned)c * ma.ldw_b, tmp);
      t.w[c] = tmp[0];
    }
  }
}

template <typename real, int G, int CH, int K, bool WLDS>
__device__ __forceinline__ void store_w(const RowTile<real, G, CH, K>& t, const MatAddr<real, G, CH, K>& ma, int wbase) {
  if constexpr (WLDS) {
    real* p = ma.lds_w + wbase + ma.lane;
#pragma unroll
    for (int c = 0; c < K; ++c) p[c * ma.lds_rows] = t.w[c];
  } else {
    const unsigned sbase = (unsigned)wbase * (unsigned)sizeof(real);
    const unsigned wv = (wbase + ma.lane < ma.T) ? ma.woff : OOB;
#pragma unroll
    for (int c = 0; c < K; ++c) buf_store<real>(ma.wr, wv, sbase + (unsigned)c * ma.ldw_b, t.w[c]);
  }
}

// reduce-scatter of pn[G][K] over the G lanes of a group: afterwards pn[0][*] of lane g = sum over the
// group's lanes of (their) pn[g][*].
template <int HSZ, typename real, int G, int K>
__device__ __forceinline__ void reduce_scatter(real (&pn)[G][K], int g) {
  if constexpr (HSZ >= 1) {
    const bool up = (g & HSZ) != 0;
#pragma unroll
    for (int i = 0; i < HSZ; ++i)
#pragma unroll
      for (int c = 0; c < K; ++c) {
        const real lo = pn[i][c], hi = pn[i + HSZ][c];
        const real send = up ? lo : hi;
        const real keep = up ? hi : lo;
        pn[i][c] = keep + xor_lane<HSZ>(send);
      }
    reduce_scatter<HSZ / 2, real, G, K>(pn, g);
  }
}

// One step: W-update of the lane's row and accumulation of W^T X (lane's channels) and W^T W (own row).
// WSTAGE (rows whose W lives in the LDS cache, G > 1): the new row goes to the cache right here (instead of
// store_w afterwards) and the G rows of the lane's group come straight back as one vector LDS read per
// component, replacing the G x K DPP broadcasts of the register path -- a DPP move costs ~3 plain VALU
// instructions on gfx950 and the W^T X accumulation stalls on each of them.  Experiment (-DHIPNMF_WSTAGE=1):
// in isolation it saves 12 % of the tile arithmetic (tools/ubench/tile_rate.hip: 537 -> 474 ns per tile and
// SIMD), inside the kernel the extra LDS round trip and 7 more spilled VGPRs make it slower (8.27 vs 8.54 M
// matrix-it/s at B = 2048), so the DPP path stays the default.
#ifndef HIPNMF_WSTAGE
#define HIPNMF_WSTAGE 0
#endif
template <typename real, int G, int CH, int K, bool WSTAGE = false>
__device__ __forceinline__ void update_tile(RowTile<real, G, CH, K>& t, const MatAddr<real, G, CH, K>& ma,
                                            const real (&h)[K][CH], const real (&hht)[K][K],
                                            real (&accA)[K][CH], real (&accB)[Cfg<real, G, CH, K>::NB],
                                            real l1w, real l2w, bool update_h, int wbase = 0) {
  const int g = ma.g;
  // numerator X H^T (_nmf.py:543): partial over this lane's channels, for each of the G rows
  real pn[G][K];
  if constexpr (h_in_lds<real, G, CH>()) {
    constexpr int MP = G * CH;
    const real* hp = ma.h_lds + g * CH;
    asm volatile("" : "+v"(hp));  // opaque per tile: keeps the K*CH broadcast reads inside the row loop
#pragma unroll
    for (int c = 0; c < K; ++c) {
      real hc[CH];
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) hc[cc] = hp[c * MP + cc];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        real s = t.x[0][r] * hc[0];
#pragma unroll
        for (int cc = 1; cc < CH; ++cc) s = fma_(t.x[cc][r], hc[cc], s);
        pn[r][c] = s;
      }
    }
  } else {
    constexpr int RL = h_lds_rows<G, CH, K>();
#pragma unroll
    for (int r = 0; r < G; ++r)
#pragma unroll
      for (int c = 0; c < K - RL; ++c) {
        real s = t.x[0][r] * h[c][0];
#pragma unroll
        for (int cc = 1; cc < CH; ++cc) s = fma_(t.x[cc][r], h[c][cc], s);
        pn[r][c] = s;
      }
    if constexpr (RL > 0) {  // G == 1: rows K-RL .. K-1 of H from LDS
      const real* hp = ma.h_lds;
      asm volatile("" : "+v"(hp));  // opaque per tile: keeps the reads inside the row loop
#pragma unroll
      for (int c = K - RL; c < K; ++c) {
        real hc[CH];
        __builtin_memcpy(hc, __builtin_assume_aligned(hp + c * (G * CH), 16), sizeof(hc));
        real s = t.x[0][0] * hc[0];
#pragma unroll
        for (int cc = 1; cc < CH; ++cc) s = fma_(t.x[cc][0], hc[cc], s);
        pn[0][c] = s;
      }
    }
  }
  reduce_scatter<G / 2, real, G, K>(pn, g);

  // denominator W (H H^T) (_nmf.py:553-554), regularisation (:616-619), zero guard (:620), update (:622-629)
  real wn[K], den[K], num[K], quo[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    real d = t.w[0] * hht[0][c];
#pragma unroll
    for (int c2 = 1; c2 < K; ++c2) d = fma_(t.w[c2], hht[c2][c], d);
    // regularisation (_nmf.py:616-619 adds the terms only when positive): added unconditionally -- the coefficients are >= 0
    // (validated by the host) and d + 0, d + 0 * w leave d as it is, while the conditional form costs a select per term and row
    d = d + l1w;
    d = d + l2w * t.w[c];
    den[c] = (d == (real)0) ? eps_val<real>() : d;
    num[c] = pn[0][c];
  }
  quotients<K>(num, den, quo);
#pragma unroll
  for (int c = 0; c < K; ++c) wn[c] = t.w[c] * quo[c];
#pragma unroll
  for (int c = 0; c < K; ++c) t.w[c] = wn[c];

  if constexpr (WSTAGE) {  // store_w<WLDS = true>, done here so that the group can read the rows back
    real* p = ma.lds_w + wbase + ma.lane;
#pragma unroll
    for (int c = 0; c < K; ++c) p[c * ma.lds_rows] = wn[c];
  }
  if (update_h) {
    if constexpr (WSTAGE && G > 1) {
      // W^T X (_nmf.py:639): the group's G consecutive rows of every component, one aligned vector read each
      // (same wave wrote them just above; LDS operations of a wave execute in order)
      const real* q = ma.lds_w + wbase + (ma.lane - g);
      real wg[K][G];
#pragma unroll
      for (int c = 0; c < K; ++c)
        __builtin_memcpy(wg[c], __builtin_assume_aligned(q + c * ma.lds_rows, sizeof(real) * G), sizeof(real) * G);
#pragma unroll
      for (int r = 0; r < G; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c)
#pragma unroll
          for (int cc = 0; cc < CH; ++cc) accA[c][cc] = fma_(wg[c][r], t.x[cc][r], accA[c][cc]);
    } else {
      // W^T X (_nmf.py:639): rows of the group broadcast lane by lane
      static_for<G>([&](auto R) {
        constexpr int r = decltype(R)::value;
#pragma unroll
        for (int c = 0; c < K; ++c) {
          const real wr = group_bcast<G, r>(wn[c]);
#pragma unroll
          for (int cc = 0; cc < CH; ++cc) accA[c][cc] = fma_(wr, t.x[cc][r], accA[c][cc]);
        }
      });
    }
    // W^T W (first factor of multi_dot, _nmf.py:640), upper triangle, own row
    int idx = 0;
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int c2 = c; c2 < K; ++c2) {
        accB[idx] = fma_(wn[c], wn[c2], accB[idx]);
        ++idx;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Kullback-Leibler loss (beta_loss = 1).  Same lane mapping as update_tile:
//   W *= ((X / WH) H^T) / rowsum(H)          (_nmf.py:556-591, 615-631), WH clamped at EPSILON (:574-575)
//   H *= (W^T (X / WH)) / colsum(W)          (_nmf.py:642-684, 701-728) -- WH recomputed with the new W
// hsum[c] = sum_j H[c][j] arrives in hht[0][c]; accB[0..K) accumulates colsum(W) (the rest stays 0).
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void update_tile_kl(RowTile<real, G, CH, K>& t, const MatAddr<real, G, CH, K>& ma,
                                               const real (&h)[K][CH], const real (&hht)[K][K],
                                               real (&accA)[K][CH], real (&accB)[Cfg<real, G, CH, K>::NB],
                                               real l1w, real l2w, bool update_h) {
  const int g = ma.g;
#ifndef HIPNMF_KL_NO_PK
  if constexpr (G == 1 && sizeof(real) == 4 && CH == 8) {
    // Row-per-lane fp32, 7..8 channels (round 6): the four K x CH products of the iteration -- W H, Q H^T, W' H, W'^T Q' -- as
    // v_pk_fma_f32 over channel PAIRS (the scalar factor of a pair is an op_sel splat, no extra move): half the FMA instructions per
    // row and one packed multiply per pair of quotients.  Explicit two-element vectors: -fno-slp-vectorize stays on (left to itself
    // the vectoriser packs the Frobenius tile with ~40 registers of shuffles, profiles/README.md round 1).  Measured on one box
    // (profiles/r06_kl_narrow_pk_ab.log, 2048+ matrices): 8 x 4 at 2 500 rows 41.3 -> 44.6 M matrix-it/s; 16 x 5 at 10 000 rows
    // 6.49 -> 5.21 (the pairs cost 27 more spilled registers at the 256-register cap and a packed FMA has no rate advantage once
    // two waves share the SIMD): so CH == 8 only.
    // The sums over a row's channels add the even and the odd channels' partial sums at the end: another fixed order.
    using f2 = float __attribute__((ext_vector_type(2)));
    constexpr int P = CH / 2;
    auto hv = [&](int c, int j) __attribute__((always_inline)) -> f2 { return f2{h[c][2 * j], h[c][2 * j + 1]}; };
    auto quot = [&](const f2 (&rec)[P], f2 (&qo)[P]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const f2 r = f2{__builtin_amdgcn_rcpf(kl_floor(rec[j].x)), __builtin_amdgcn_rcpf(kl_floor(rec[j].y))};
        qo[j] = f2{t.x[2 * j][0], t.x[2 * j + 1][0]} * r;
      }
    };
    f2 rec[P], q2[P];
#pragma unroll
    for (int j = 0; j < P; ++j) rec[j] = f2{t.w[0], t.w[0]} * hv(0, j);
#pragma unroll
    for (int c = 1; c < K; ++c)
#pragma unroll
      for (int j = 0; j < P; ++j) rec[j] = __builtin_elementwise_fma(f2{t.w[c], t.w[c]}, hv(c, j), rec[j]);
    quot(rec, q2);
    real wn[K], den[K], num[K], quo[K];
#pragma unroll
    for (int c = 0; c < K; ++c) {
      f2 s2 = q2[0] * hv(c, 0);
#pragma unroll
      for (int j = 1; j < P; ++j) s2 = __builtin_elementwise_fma(q2[j], hv(c, j), s2);
      num[c] = s2.x + s2.y;
      real d = hht[0][c];
      d = d + l1w;
      d = d + l2w * t.w[c];
      den[c] = (d == (real)0) ? eps_val<real>() : d;
    }
    quotients<K>(num, den, quo);
#pragma unroll
    for (int c = 0; c < K; ++c) wn[c] = t.w[c] * quo[c];
#pragma unroll
    for (int c = 0; c < K; ++c) t.w[c] = wn[c];
    if (update_h) {
#pragma unroll
      for (int j = 0; j < P; ++j) rec[j] = f2{wn[0], wn[0]} * hv(0, j);
#pragma unroll
      for (int c = 1; c < K; ++c)
#pragma unroll
        for (int j = 0; j < P; ++j) rec[j] = __builtin_elementwise_fma(f2{wn[c], wn[c]}, hv(c, j), rec[j]);
      quot(rec, q2);
#pragma unroll
      for (int c = 0; c < K; ++c)
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const f2 a2 = __builtin_elementwise_fma(f2{wn[c], wn[c]}, q2[j], f2{accA[c][2 * j], accA[c][2 * j + 1]});
          accA[c][2 * j] = a2.x;
          accA[c][2 * j + 1] = a2.y;
        }
#pragma unroll
      for (int c = 0; c < K; ++c) accB[c] += wn[c];
    }
    return;
  }
#endif
  // q = X / max(WH, EPSILON) for the group's G rows restricted to this lane's channels
  real q[CH][G];
  static_for<G>([&](auto R) {
    constexpr int r = decltype(R)::value;
    real wr[K];
#pragma unroll
    for (int c = 0; c < K; ++c) wr[c] = group_bcast<G, r>(t.w[c]);
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      real rec = wr[0] * h[0][cc];
#pragma unroll
      for (int c = 1; c < K; ++c) rec = fma_(wr[c], h[c][cc], rec);
      rec = kl_floor(rec);
      if constexpr (sizeof(real) == 4)
        q[cc][r] = kl_quot(t.x[cc][r], rec);
      else
        q[cc][r] = t.x[cc][r] / rec;
    }
  });
  real pn[G][K];
#pragma unroll
  for (int r = 0; r < G; ++r)
#pragma unroll
    for (int c = 0; c < K; ++c) {
      real s = q[0][r] * h[c][0];
#pragma unroll
      for (int cc = 1; cc < CH; ++cc) s = fma_(q[cc][r], h[c][cc], s);
      pn[r][c] = s;
    }
  reduce_scatter<G / 2, real, G, K>(pn, g);
  real wn[K], den[K], num[K], quo[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    real d = hht[0][c];
    // regularisation (_nmf.py:616-619 adds the terms only when positive): added unconditionally -- the coefficients are >= 0
    // (validated by the host) and d + 0, d + 0 * w leave d as it is, while the conditional form costs a select per term and row
    d = d + l1w;
    d = d + l2w * t.w[c];
    den[c] = (d == (real)0) ? eps_val<real>() : d;
    num[c] = pn[0][c];
  }
  quotients<K>(num, den, quo);
#pragma unroll
  for (int c = 0; c < K; ++c) wn[c] = t.w[c] * quo[c];
#pragma unroll
  for (int c = 0; c < K; ++c) t.w[c] = wn[c];
  if (update_h) {
    static_for<G>([&](auto R) {
      constexpr int r = decltype(R)::value;
      real wr[K];
#pragma unroll
      for (int c = 0; c < K; ++c) wr[c] = group_bcast<G, r>(wn[c]);
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        real rec = wr[0] * h[0][cc];
#pragma unroll
        for (int c = 1; c < K; ++c) rec = fma_(wr[c], h[c][cc], rec);
        rec = kl_floor(rec);
        real qq;
        if constexpr (sizeof(real) == 4)
          qq = kl_quot(t.x[cc][r], rec);
        else
          qq = t.x[cc][r] / rec;
#pragma unroll
        for (int c = 0; c < K; ++c) accA[c][cc] = fma_(wr[c], qq, accA[c][cc]);
      }
    });
#pragma unroll
    for (int c = 0; c < K; ++c) accB[c] += wn[c];
  }
}

// residual of the group's rows restricted to this lane's channels: sse += (x - w.h)^2, xsq += x^2
// LOSS == 1 additionally accumulates the generalised KL divergence (_nmf.py:140-161, 185-189), element by
// element as x log(x / wh) - x + wh (each term >= 0) with zeros of X skipped and wh clamped at EPSILON.
__device__ __forceinline__ float log_(float a) { return ::logf(a); }
__device__ __forceinline__ double log_(double a) { return ::log(a); }
template <typename real, int G, int CH, int K, int LOSS = 0>
__device__ __forceinline__ void resid_tile(const RowTile<real, G, CH, K>& t, const MatAddr<real, G, CH, K>& ma,
                                           const real (&h)[K][CH], real (&sse)[CH], real (&xsq)[CH], real& kl) {
  if constexpr (h_in_lds<real, G, CH, LOSS>()) {
    constexpr int MP = G * CH;
    const real* hp = ma.h_lds + ma.g * CH;
    asm volatile("" : "+v"(hp));
    real hl[K][CH];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) hl[c][cc] = hp[c * MP + cc];
    static_for<G>([&](auto R) {
      constexpr int r = decltype(R)::value;
      real wr[K];
#pragma unroll
      for (int c = 0; c < K; ++c) wr[c] = group_bcast<G, r>(t.w[c]);
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        real rec = wr[0] * hl[0][cc];
#pragma unroll
        for (int c = 1; c < K; ++c) rec = fma_(wr[c], hl[c][cc], rec);
        const real d = t.x[cc][r] - rec;
        sse[cc] = fma_(d, d, sse[cc]);
        xsq[cc] = fma_(t.x[cc][r], t.x[cc][r], xsq[cc]);
      }
    });
    return;
  }
  constexpr int RL = h_lds_rows<G, CH, K>();
  real hx[RL > 0 ? RL : 1][CH];  // the rows of H that are not kept in registers
  if constexpr (RL > 0) {
    const real* hp = ma.h_lds;
#pragma unroll
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) hx[q][cc] = hp[(K - RL + q) * (G * CH) + cc];
  }
  static_for<G>([&](auto R) {
    constexpr int r = decltype(R)::value;
    real wr[K];
#pragma unroll
    for (int c = 0; c < K; ++c) wr[c] = group_bcast<G, r>(t.w[c]);
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      real rec = wr[0] * (RL >= K ? hx[0][cc] : h[0][cc]);
#pragma unroll
      for (int c = 1; c < K; ++c) rec = fma_(wr[c], (c >= K - RL) ? hx[c - (K - RL) < 0 ? 0 : c - (K - RL)][cc] : h[c][cc], rec);
      const real d = t.x[cc][r] - rec;
      sse[cc] = fma_(d, d, sse[cc]);
      xsq[cc] = fma_(t.x[cc][r], t.x[cc][r], xsq[cc]);
      if constexpr (LOSS == 1) {
        // X log(X / WH) - X + WH where X > EPSILON, WH elsewhere (_nmf.py:138-155).  Branch-free: the logarithm is
        // evaluated for every element (on max(X, EPSILON), which the select below discards where it differs) instead
        // of inside a divergent region -- 32 inlined double-precision logarithms per tile under `if (x > eps)` made
        // the float64 (4, 8) instance spill inside divergent control flow and fault (tools/repro/case69.py)
        const real x = t.x[cc][r];
        const real whc = rec < eps_val<real>() ? eps_val<real>() : rec;
        const real xs = x > eps_val<real>() ? x : eps_val<real>();
        const real lg = fma_(x, log_(xs / whc), rec - x);
        kl += (x > eps_val<real>()) ? lg : rec;
      }
    }
  });
}

// ------------------------------------------------------------------------------------------------
// LDS carve-up shared by all solver kernels
template <typename real, int G, int CH, int K>
struct Smem {
  using C = Cfg<real, G, CH, K>;
  real* H;     // [K][MP]
  real* HHt;   // [K][K]
  real* A;     // [K][MP]
  real* B;     // [K][K]
  real* part;  // [NW][NACC]  (also used as [NW][2*MP] by the residual)
  real* scal;  // [8]
  __device__ __forceinline__ Smem(unsigned char* raw, int nw) {
    H = reinterpret_cast<real*>(raw);
    HHt = H + K * C::MP;
    A = HHt + K * K;
    B = A + K * C::MP;
    part = B + K * K;
    scal = part + nw * C::NREC;
  }
  __host__ __device__ static size_t bytes(int nw) { return sizeof(real) * (size_t)(2 * K * C::MP + 2 * K * K + nw * C::NREC + 8); }
};

template <typename real, int G, int CH, int K>
__device__ __forceinline__ void load_h_to_lds(Smem<real, G, CH, K>& s, const real* __restrict__ Hb, int m) {
  constexpr int MP = G * CH;
  for (int i = threadIdx.x; i < K * MP; i += blockDim.x) {
    const int c = i / MP, j = i % MP;
    s.H[i] = (j < m) ? Hb[c * m + j] : (real)0;
  }
}

// H H^T from LDS H (_nmf.py:553); call between barriers
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void compute_hht(Smem<real, G, CH, K>& s) {
  constexpr int MP = G * CH;
  for (int i = threadIdx.x; i < K * K; i += blockDim.x) {
    const int c = i / K, c2 = i % K;
    real acc = (real)0;
#pragma unroll
    for (int j = 0; j < MP; ++j) acc = fma_(s.H[c * MP + j], s.H[c2 * MP + j], acc);
    s.HHt[i] = acc;
  }
}

template <typename real, int G, int CH, int K, int LOSS = 0>
__device__ __forceinline__ void load_h_regs(const Smem<real, G, CH, K>& s, int g, real (&h)[K][CH], real (&hht)[K][K]) {
  constexpr int MP = G * CH;
  if constexpr (!h_in_lds<real, G, CH, LOSS>()) {
#pragma unroll
    for (int c = 0; c < K - h_lds_rows<G, CH, K>(); ++c)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) h[c][cc] = s.H[c * MP + g * CH + cc];
#pragma unroll
    for (int c = K - h_lds_rows<G, CH, K>(); c < K; ++c)  // read from LDS per tile (never used from registers)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) h[c][cc] = (real)0;
  }
#pragma unroll
  for (int c = 0; c < K; ++c)
#pragma unroll
    for (int c2 = 0; c2 < K; ++c2) hht[c][c2] = uniform(s.HHt[c * K + c2]);
}

// streaming pass over rows [row_begin, row_end) (row_begin % 64 == 0): W update + accumulation.
// Software pipeline: PF tiles per wave are in flight (loads issued PF-1 steps ahead of their use).
#ifndef HIPNMF_PF
#define HIPNMF_PF 2      // tiles in flight per wave when W streams from global memory (21 VGPRs per tile)
#endif
#ifndef HIPNMF_PF_LDS
#define HIPNMF_PF_LDS 2  // same for the rows whose W lives in LDS (16 VGPRs per tile)
#endif
template <bool WLDS, int G = 4, int CH = 4>
struct PipeDepth {
  static constexpr int value = x_row_major<G, CH>() ? (HIPNMF_PF_ROW) : (WLDS ? HIPNMF_PF_LDS : HIPNMF_PF);
};
// the tiles a wave keeps in flight (an alias: an array bound with template arguments confuses the parser in a
// parameter list)
template <typename real, int G, int CH, int K, bool WLDS>
using TileBuf = RowTile<real, G, CH, K>[PipeDepth<WLDS, G, CH>::value];

// issue the loads of this wave's first PF tiles of rows [row_begin, row_end)
template <typename real, int G, int CH, int K, bool WLDS>
__device__ __forceinline__ void prefetch_head(TileBuf<real, G, CH, K, WLDS>& tiles,
                                              const MatAddr<real, G, CH, K>& ma, int row_begin, int row_end) {
  constexpr int PF = PipeDepth<WLDS, G, CH>::value;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int stride = (blockDim.x / WAVE) * WAVE;
  const int wbase = row_begin + wave * WAVE;
#pragma unroll
  for (int p = 0; p < PF; ++p)
    load_tile<real, G, CH, K, WLDS>(tiles[p], ma, wbase + p * stride, wbase + p * stride < row_end);
}

// PRELOADED: `tiles` already holds (or has in flight) the first PF tiles -- the persistent kernel issues them
// before the reduction / H-update phase of the previous iteration so the pipeline never starts cold.
template <typename real, int G, int CH, int K, bool WLDS = false, bool PRELOADED = false, int LOSS = 0>
__device__ __forceinline__ void rows_update_pass(const MatAddr<real, G, CH, K>& ma, int row_begin, int row_end,
                                                 const real (&h)[K][CH], const real (&hht)[K][K],
                                                 real (&accA)[K][CH], real (&accB)[Cfg<real, G, CH, K>::NB], real l1w,
                                                 real l2w, bool update_h,
                                                 TileBuf<real, G, CH, K, WLDS>& tiles) {
  constexpr int PF = PipeDepth<WLDS, G, CH>::value;
  // readfirstlane makes the wave id (hence every row base / SGPR offset) provably wave-uniform; without it
  // hipcc wraps each buffer access in a waterfall loop
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int stride = (blockDim.x / WAVE) * WAVE;
  int wbase = row_begin + wave * WAVE;  // wave-uniform
  if (wbase >= row_end) return;
  const int nsteps = (row_end - wbase + stride - 1) / stride;  // wave-uniform trip count
  const int nfull = nsteps / PF, rem = nsteps - nfull * PF;
  if constexpr (!PRELOADED) prefetch_head<real, G, CH, K, WLDS>(tiles, ma, row_begin, row_end);
  unsigned touch_prev = 0, touch_sink = 0;  // cache-line touches of the tiles further ahead (row-major instances)
  // Main loop: PF tiles per trip, no exit in the middle (a single back edge keeps hipcc's s_waitcnt vmcnt
  // counting exact, so the tiles loaded PF-1 steps ahead really stay in flight).
  for (int grp = 0; grp < nfull; ++grp) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      constexpr bool STAGE = WLDS && LOSS == 0 && (HIPNMF_WSTAGE != 0) && !h_in_lds<real, G, CH, LOSS>();
      if constexpr (LOSS == 1)
        update_tile_kl<real, G, CH, K>(tiles[p], ma, h, hht, accA, accB, l1w, l2w, update_h);
      else
        update_tile<real, G, CH, K, STAGE>(tiles[p], ma, h, hht, accA, accB, l1w, l2w, update_h, wbase);
      if constexpr (!STAGE) store_w<real, G, CH, K, WLDS>(tiles[p], ma, wbase);
      const int nb = wbase + PF * stride;
      load_tile<real, G, CH, K, WLDS>(tiles[p], ma, nb, nb < row_end);
      if constexpr (x_row_major<G, CH>() && (HIPNMF_ROW_TOUCH) > 0) {
        const int tb = nb + (HIPNMF_ROW_TOUCH) * stride;
        touch_sink ^= touch_prev;  // consumes the touch issued one tile ago: never a wait on a fresh load
        const bool ok = tb < row_end && tb + ma.lane < ma.T;
        touch_prev = __builtin_amdgcn_raw_buffer_load_b32(ma.xr, ok ? ma.xoff[0] : OOB, (unsigned)tb * ma.xrow_b, 0);
      }
      wbase += stride;
#ifndef HIPNMF_NO_TILE_BARRIER
      __builtin_amdgcn_sched_barrier(0);  // keep the tiles' arithmetic from being interleaved (VGPR pressure)
#endif
    }
  }
#pragma unroll
  for (int p = 0; p < PF - 1; ++p) {
    if (p < rem) {  // wave-uniform
      constexpr bool STAGE = WLDS && LOSS == 0 && (HIPNMF_WSTAGE != 0) && !h_in_lds<real, G, CH, LOSS>();
      if constexpr (LOSS == 1)
        update_tile_kl<real, G, CH, K>(tiles[p], ma, h, hht, accA, accB, l1w, l2w, update_h);
      else
        update_tile<real, G, CH, K, STAGE>(tiles[p], ma, h, hht, accA, accB, l1w, l2w, update_h, wbase);
      if constexpr (!STAGE) store_w<real, G, CH, K, WLDS>(tiles[p], ma, wbase);
      wbase += stride;
    }
  }
  if constexpr (x_row_major<G, CH>() && (HIPNMF_ROW_TOUCH) > 0) {
    touch_sink ^= touch_prev;
    asm volatile("" ::"v"(touch_sink));  // keeps the touch loads alive
  }
}

template <typename real, int G, int CH, int K, bool WLDS = false, int LOSS = 0>
__device__ __forceinline__ void rows_resid_pass(const MatAddr<real, G, CH, K>& ma, int row_begin, int row_end,
                                                const real (&h)[K][CH], real (&sse)[CH], real (&xsq)[CH], real& kl) {
  // readfirstlane makes the wave id (hence every row base / SGPR offset) provably wave-uniform; without it
  // hipcc wraps each buffer access in a waterfall loop
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int stride = (blockDim.x / WAVE) * WAVE;
  int wbase = row_begin + wave * WAVE;
  if (wbase >= row_end) return;
  RowTile<real, G, CH, K> ta, tb;
  load_tile<real, G, CH, K, WLDS>(ta, ma, wbase, true);
  while (true) {
    load_tile<real, G, CH, K, WLDS>(tb, ma, wbase + stride, wbase + stride < row_end);
    resid_tile<real, G, CH, K, LOSS>(ta, ma, h, sse, xsq, kl);
    wbase += stride;
    if (wbase >= row_end) break;
    load_tile<real, G, CH, K, WLDS>(ta, ma, wbase + stride, wbase + stride < row_end);
    resid_tile<real, G, CH, K, LOSS>(tb, ma, h, sse, xsq, kl);
    wbase += stride;
    if (wbase >= row_end) break;
  }
}

// Reduce-scatter of N per-lane values over the 64 lanes of a wave: afterwards v[0] of lane l holds the sum over
// all lanes of value l and v[1] that of value 64 + l.  Each of the six stages halves the number of live
// registers (the lanes whose bit is set keep the odd element of a pair and hand the even one to their partner),
// so the whole reduction costs ~N cross-lane moves instead of 6 N.  Fixed association order.
template <int MASK, int N, typename real>
__device__ __forceinline__ void wave_reduce_scatter(real (&v)[N], int lane) {
  if constexpr (MASK <= 32) {
    constexpr int NH = (N + 1) / 2;
    const bool up = (lane & MASK) != 0;
    real nxt[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const real a = v[2 * i], b = (2 * i + 1 < N) ? v[2 * i + 1] : (real)0;
      const real keep = up ? b : a, send = up ? a : b;
      real got;
      if constexpr (MASK <= 2)
        got = xor_lane<MASK>(send);
      else
        got = __shfl_xor(send, MASK, WAVE);
      nxt[i] = keep + got;
    }
    real (&w)[NH] = nxt;
    wave_reduce_scatter<MASK * 2, NH, real>(w, lane);
#pragma unroll
    for (int i = 0; i < NH; ++i) v[i] = nxt[i];
  }
}

// wave butterfly of the accumulators, then one record per wave in LDS
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void wave_reduce_acc(real* __restrict__ rec /* [NACC] of this wave */, real (&accA)[K][CH],
                                                real (&accB)[Cfg<real, G, CH, K>::NB]) {
  using C = Cfg<real, G, CH, K>;
  const int lane = threadIdx.x & (WAVE - 1);
  if constexpr (x_row_major<G, CH>()) {  // row-per-lane mapping: K*CH + NB (= 95 for k = 5, m = 16) values per lane
    real v[C::NACC];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) v[c * C::MP + cc] = accA[c][cc];
#pragma unroll
    for (int i = 0; i < C::NB; ++i) v[K * C::MP + i] = accB[i];
    wave_reduce_scatter<1, C::NACC, real>(v, lane);
    constexpr int NOUT = (C::NACC + WAVE - 1) / WAVE;
#pragma unroll
    for (int q = 0; q < NOUT; ++q)
      if (q * WAVE + lane < C::NACC) rec[q * WAVE + lane] = v[q];
    return;
  }
  // inside a 16-lane row: DPP rotations (VALU only); across the four rows: two ds_bpermute stages
#pragma unroll
  for (int c = 0; c < K; ++c)
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      real v = accA[c][cc];
      if constexpr (G == 1) {
        v += xor_lane<1>(v);
        v += xor_lane<2>(v);
      } else if constexpr (G == 2) {
        v += xor_lane<2>(v);
      }
      if constexpr (G <= 4) v += dpp_mov<0x124>(v);  // row_ror:4
      if constexpr (G <= 8) v += dpp_mov<0x128>(v);  // row_ror:8
      v += __shfl_xor(v, 16, WAVE);
      v += __shfl_xor(v, 32, WAVE);
      accA[c][cc] = v;
    }
#pragma unroll
  for (int i = 0; i < C::NB; ++i) {
    real v = accB[i];
    v += xor_lane<1>(v);
    v += xor_lane<2>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x128>(v);
    v += __shfl_xor(v, 16, WAVE);
    v += __shfl_xor(v, 32, WAVE);
    accB[i] = v;
  }
  if (lane < G) {
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) rec[c * C::MP + lane * CH + cc] = accA[c][cc];
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < C::NB; ++i) rec[K * C::MP + i] = accB[i];
  }
}

// H update from LDS A (= W^T X, [K][MP]) and B (= W^T W, full [K][K]) (_nmf.py:638-640, 701-728).
// Barriers inside; every thread of the block must call it.  Leaves new H and H H^T in LDS.
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void h_update_lds(Smem<real, G, CH, K>& s, int m, real l1h, real l2h) {
  constexpr int MP = G * CH;
  real newh = (real)0;
  const int i = threadIdx.x;
  const bool active = i < K * MP;
  if (active) {
    const int c = i / MP, j = i % MP;
    if (j < m) {
      real d = s.B[c * K + 0] * s.H[0 * MP + j];
#pragma unroll
      for (int c2 = 1; c2 < K; ++c2) d = fma_(s.B[c * K + c2], s.H[c2 * MP + j], d);
      const real hold = s.H[i];
      if (l1h > (real)0) d = d + l1h;
      if (l2h > (real)0) d = d + l2h * hold;
      d = (d == (real)0) ? eps_val<real>() : d;
      newh = hold * (s.A[i] / d);
    }
  }
  __syncthreads();
  if (active) s.H[i] = newh;
  __syncthreads();
  compute_hht(s);
  __syncthreads();
}

// KL: rowsum(H) (the W-update denominator, _nmf.py:577-581) into s.HHt[0..K); call between barriers
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void compute_hsum(Smem<real, G, CH, K>& s) {
  constexpr int MP = G * CH;
  if (threadIdx.x < K) {
    real acc = (real)0;
#pragma unroll
    for (int j = 0; j < MP; ++j) acc += s.H[threadIdx.x * MP + j];
    s.HHt[threadIdx.x] = acc;
  }
}

// KL epilogue (wave 0 alone): records hold W^T (X / WH) [K][MP] followed by colsum(W) [K].
//   H *= A / colsum(W)   (_nmf.py:663-684, 701-728; colsum 0 -> 1, then regularisation, then 0 -> EPSILON),
//   H[H < float64 eps] = 0 (_nmf.py:866-868), rowsum(H) for the next W update.
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void wave0_combine_and_update_h_kl(Smem<real, G, CH, K>& s, int nw, int m, real l1h, real l2h) {
  using C = Cfg<real, G, CH, K>;
  constexpr int MP = C::MP;
  const int lane = threadIdx.x & (WAVE - 1);
  for (int i = lane; i < K * MP + K; i += WAVE) {
    real acc = s.part[i];
    for (int w = 1; w < nw; ++w) acc += s.part[w * C::NACC + i];
    if (i < K * MP)
      s.A[i] = acc;
    else
      s.B[i - K * MP] = acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  for (int i = lane; i < K * MP; i += WAVE) {
    const int c = i / MP, j = i % MP;
    real newh = (real)0;
    if (j < m) {
      real d = s.B[c];
      if (d == (real)0) d = (real)1;
      const real hold = s.H[i];
      if (l1h > (real)0) d = d + l1h;
      if (l2h > (real)0) d = d + l2h * hold;
      d = (d == (real)0) ? eps_val<real>() : d;
      newh = hold * (s.A[i] / d);
      if (newh < (real)2.220446049250313e-16) newh = (real)0;
    }
    s.H[i] = newh;  // element i is read and written by the same lane only
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane < K) {
    real acc = (real)0;
#pragma unroll
    for (int j = 0; j < MP; ++j) acc += s.H[lane * MP + j];
    s.HHt[lane] = acc;
  }
}

// Persistent-kernel epilogue of an iteration, executed by wave 0 alone between two workgroup barriers:
// sum the wave records (fixed order) -> W^T X, W^T W; H update (_nmf.py:638-640, 701-728); H H^T.
// LDS accesses of one wave execute in order, so only compiler/LDS-counter fences are needed inside.
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void wave0_combine_and_update_h(Smem<real, G, CH, K>& s, int nw, int m, real l1h, real l2h) {
  using C = Cfg<real, G, CH, K>;
  constexpr int MP = C::MP;
  const int lane = threadIdx.x & (WAVE - 1);
  for (int i = lane; i < C::NACC; i += WAVE) {
    real acc = s.part[i];
    for (int w = 1; w < nw; ++w) acc += s.part[w * C::NACC + i];
    if (i < K * MP) {
      s.A[i] = acc;
    } else {
      int idx = i - K * MP, c = 0;
      while (idx >= K - c) {
        idx -= K - c;
        ++c;
      }
      const int c2 = c + idx;
      s.B[c * K + c2] = acc;
      s.B[c2 * K + c] = acc;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  constexpr int NH = (K * MP + WAVE - 1) / WAVE;
  real newh[NH];
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    const int i = lane + q * WAVE;
    newh[q] = (real)0;
    if (i < K * MP) {
      const int c = i / MP, j = i % MP;
      if (j < m) {
        real d = s.B[c * K + 0] * s.H[0 * MP + j];
#pragma unroll
        for (int c2 = 1; c2 < K; ++c2) d = fma_(s.B[c * K + c2], s.H[c2 * MP + j], d);
        const real hold = s.H[i];
        if (l1h > (real)0) d = d + l1h;
        if (l2h > (real)0) d = d + l2h * hold;
        d = (d == (real)0) ? eps_val<real>() : d;
        newh[q] = hold * (s.A[i] / d);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    const int i = lane + q * WAVE;
    if (i < K * MP) s.H[i] = newh[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane < K * K) {
    const int c = lane / K, c2 = lane % K;
    real acc = (real)0;
#pragma unroll
    for (int j = 0; j < MP; ++j) acc = fma_(s.H[c * MP + j], s.H[c2 * MP + j], acc);
    s.HHt[lane] = acc;
  }
}

// sum of wave records -> LDS A / B (fixed order over waves)
template <typename real, int G, int CH, int K>
__device__ __forceinline__ void combine_wave_records(Smem<real, G, CH, K>& s, int nw) {
  using C = Cfg<real, G, CH, K>;
  for (int i = threadIdx.x; i < C::NACC; i += blockDim.x) {
    real acc = s.part[i];
    for (int w = 1; w < nw; ++w) acc += s.part[w * C::NACC + i];
    if (i < K * C::MP) {
      s.A[i] = acc;
    } else {
      // unpack the upper triangle index
      int idx = i - K * C::MP, c = 0;
      while (idx >= K - c) {
        idx -= K - c;
        ++c;
      }
      const int c2 = c + idx;
      s.B[c * K + c2] = acc;
      s.B[c2 * K + c] = acc;
    }
  }
}

// block-wide residual over rows [row_begin,row_end): returns per-column sse/xsq in LDS part[0 .. 2*MP)
// (sums over the block's waves, fixed order) and, for LOSS == 1, the KL divergence in part[2*MP].  Barriers inside.
template <typename real, int G, int CH, int K, int LOSS = 0>
__device__ __forceinline__ void block_residual(Smem<real, G, CH, K>& s, const MatAddr<real, G, CH, K>& ma, int row_begin,
                                               int row_end, const real (&h)[K][CH]) {
  constexpr int MP = G * CH;
  constexpr int NR = 2 * MP + (LOSS == 1 ? 1 : 0);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int nw = blockDim.x / WAVE;
  real sse[CH], xsq[CH], kl = (real)0;
#pragma unroll
  for (int cc = 0; cc < CH; ++cc) sse[cc] = xsq[cc] = (real)0;
  if (ma.lds_used > 0) {
    const int split = row_end < ma.lds_used ? row_end : ma.lds_used;
    rows_resid_pass<real, G, CH, K, true, LOSS>(ma, row_begin, split, h, sse, xsq, kl);
    rows_resid_pass<real, G, CH, K, false, LOSS>(ma, split, row_end, h, sse, xsq, kl);
  } else {
    rows_resid_pass<real, G, CH, K, false, LOSS>(ma, row_begin, row_end, h, sse, xsq, kl);
  }
#pragma unroll
  for (int off = G; off < WAVE; off <<= 1)
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      sse[cc] += __shfl_xor(sse[cc], off, WAVE);
      xsq[cc] += __shfl_xor(xsq[cc], off, WAVE);
    }
  if constexpr (LOSS == 1) {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) kl += __shfl_xor(kl, off, WAVE);
  }
  __syncthreads();  // part may still be read by a previous phase
  real* rec = s.part + wave * NR;
  if (lane < G) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      rec[lane * CH + cc] = sse[cc];
      rec[MP + lane * CH + cc] = xsq[cc];
    }
  }
  if constexpr (LOSS == 1) {
    if (lane == 0) rec[2 * MP] = kl;
  }
  __syncthreads();
  // thread i reads part[i + w*NR] and rewrites part[i]: no other thread touches part[i]
  if (threadIdx.x < NR) {
    real acc = s.part[threadIdx.x];
    for (int w = 1; w < nw; ++w) acc += s.part[w * NR + threadIdx.x];
    s.part[threadIdx.x] = acc;
  }
  __syncthreads();
}

// =================================================================================================
// Kernel 1: one workgroup per matrix, all iterations inside the kernel (batch mode, S == 1).
// =================================================================================================
// LOSS: 0 = Frobenius (beta_loss = 2), 1 = Kullback-Leibler (beta_loss = 1; update_tile_kl)
template <typename real, int G, int CH, int K, int LOSS = 0>
__global__ void __launch_bounds__((max_threads<real, G, CH, K>())) HIPNMF_OCC fit_persistent_kernel(SolveArgs<real> a) {
  using C = Cfg<real, G, CH, K>;
  constexpr int MP = C::MP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nw = blockDim.x / WAVE;
  Smem<real, G, CH, K> s(smem_raw, nw);
  const int b = blockIdx.x;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int g = lane % G;
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  real* __restrict__ Hb = a.H + (long long)b * K * a.m;
  int T = a.T;
  long long ldx = a.ldx, ldw = a.ldw;
  if (a.ragged) {  // packed batch of matrices with different numbers of rows (trials of unequal length)
    const long long* d = a.ragged + 4LL * b;
    T = (int)d[0];
    Xb = a.X + d[1];
    ldw = d[2];
    if constexpr (!x_row_major<G, CH>()) ldx = d[2];  // row-major instances: rows of a.ldx (= MP) values, d[1] counts them
    Wb = a.W + d[3];
  }
  const int m = a.m;
  const int row_end = ((T + WAVE - 1) / WAVE) * WAVE;
  // W cache: rows [0, lds_rows) stay in LDS for the whole fit (a multiple of blockDim.x; lds_stride is the
  // launch-wide capacity, lds_rows what this matrix uses of it)
  real* lds_w = reinterpret_cast<real*>(smem_raw + ((Smem<real, G, CH, K>::bytes(nw) + 15) / 16) * 16);
  const int lds_stride = a.lds_rows;
  const int t_blk = (int)(((long long)T + blockDim.x - 1) / blockDim.x * blockDim.x);
  const int lds_rows = lds_stride < t_blk ? lds_stride : t_blk;
  MatAddr<real, G, CH, K> ma(Xb, ldx, Wb, ldw, T, m, lds_w, lds_stride);
  ma.h_lds = s.H;
  ma.lds_used = lds_rows;
  for (int t0 = 0; t0 < lds_rows; t0 += blockDim.x) {  // row t0 + tid is owned by this thread in every pass
    const int t = t0 + threadIdx.x;
    if (t < lds_rows) {  // (lds_rows is a multiple of 64, not necessarily of the workgroup size)
#pragma unroll
      for (int c = 0; c < K; ++c) lds_w[c * lds_stride + t] = (t < T) ? Wb[(long long)c * ldw + t] : (real)0;
    }
  }

  load_h_to_lds(s, Hb, m);
  __syncthreads();
  if constexpr (LOSS == 1)
    compute_hsum(s);
  else
    compute_hht(s);
  __syncthreads();
  real h[K][CH], hht[K][K];
  load_h_regs<real, G, CH, K, LOSS>(s, g, h, hht);

  // reconstruction error from the block sums in s.part: ||X - WH||_F, or sqrt(2 KL(X || WH)) (_nmf.py:185-189)
  auto error_from_part = [&]() -> real {
    if constexpr (LOSS == 1) {
      const real d = s.part[2 * MP];
      return sqrt_((real)2 * (d > (real)0 ? d : (real)0));
    } else {
      real tot = (real)0;
      for (int j = 0; j < MP; ++j) tot += s.part[j];
      return sqrt_(tot);
    }
  };
  auto residual = [&]() -> real {
    block_residual<real, G, CH, K, LOSS>(s, ma, 0, row_end, h);
    const real e = error_from_part();
    __syncthreads();  // the next iteration writes its wave records into s.part without another barrier
    return e;
  };

  real err0 = (real)0, prev = (real)0;
  if (a.tol > (real)0) {
    err0 = residual();
    prev = err0;
  }
  int n_iter = 0;
#ifdef HIPNMF_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
#endif
  RowTile<real, G, CH, K> tiles_lds[PipeDepth<true, G, CH>::value];
  if (lds_rows > 0) prefetch_head<real, G, CH, K, true>(tiles_lds, ma, 0, lds_rows);
  for (int it = 1; it <= a.max_iter; ++it) {
    n_iter = it;
    real accA[K][CH], accB[C::NB];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) accA[c][cc] = (real)0;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) accB[i] = (real)0;
#ifdef HIPNMF_TIMING
    const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
    if (lds_rows > 0)
      rows_update_pass<real, G, CH, K, true, true, LOSS>(ma, 0, lds_rows, h, hht, accA, accB, a.l1w, a.l2w,
                                                         a.update_h != 0, tiles_lds);
    {
      RowTile<real, G, CH, K> tiles_glb[PipeDepth<false, G, CH>::value];
      rows_update_pass<real, G, CH, K, false, false, LOSS>(ma, lds_rows, row_end, h, hht, accA, accB, a.l1w, a.l2w,
                                                           a.update_h != 0, tiles_glb);
    }
    // X does not depend on H: start streaming the next iteration's first tiles now, under the reduction
    if (lds_rows > 0 && it < a.max_iter) prefetch_head<real, G, CH, K, true>(tiles_lds, ma, 0, lds_rows);
    if (a.update_h) {
      // s.part was last read before the previous iteration's second barrier (or by a residual pass that ends
      // with a barrier), so the records can be written right away: two workgroup barriers per iteration
#ifdef HIPNMF_TIMING
      const unsigned long long tm1 = __builtin_readcyclecounter();
#endif
      wave_reduce_acc<real, G, CH, K>(s.part + wave * C::NACC, accA, accB);
#ifdef HIPNMF_TIMING
      const unsigned long long tm2 = __builtin_readcyclecounter();
#endif
      __syncthreads();
#ifdef HIPNMF_TIMING
      const unsigned long long tm3 = __builtin_readcyclecounter();
#endif
      if (wave == 0) {
        if constexpr (LOSS == 1)
          wave0_combine_and_update_h_kl(s, nw, m, a.l1h, a.l2h);
        else
          wave0_combine_and_update_h(s, nw, m, a.l1h, a.l2h);
      }
#ifdef HIPNMF_TIMING
      const unsigned long long tm4 = __builtin_readcyclecounter();
#endif
      __syncthreads();
#ifdef HIPNMF_TIMING
      const unsigned long long tm5 = __builtin_readcyclecounter();
#endif
      load_h_regs<real, G, CH, K, LOSS>(s, g, h, hht);
#ifdef HIPNMF_TIMING
      const unsigned long long tm6 = __builtin_readcyclecounter();
      tacc[0] += tm1 - tm0; tacc[1] += tm2 - tm1; tacc[2] += tm3 - tm2; tacc[3] += tm4 - tm3; tacc[4] += tm5 - tm4; tacc[5] += tm6 - tm5;
#endif
    }
    if (a.tol > (real)0 && (it % a.check_every) == 0) {
      const real err = residual();
      if ((prev - err) / err0 < a.tol) break;
      prev = err;
    }
  }
  // reconstruction_err_ (_nmf.py:1628-1630) + per-column SSE / sum X^2 for VAF (analysis.py:654-662)
  block_residual<real, G, CH, K, LOSS>(s, ma, 0, row_end, h);
  if (threadIdx.x == 0) {
    if (a.err_out) a.err_out[b] = error_from_part();
    if (a.n_iter_out) a.n_iter_out[b] = n_iter;
  }
  if (threadIdx.x < m) {
    if (a.sse_col_out) a.sse_col_out[(long long)b * m + threadIdx.x] = s.part[threadIdx.x];
    if (a.xsq_col_out) a.xsq_col_out[(long long)b * m + threadIdx.x] = s.part[MP + threadIdx.x];
  }
#ifdef HIPNMF_TIMING
  // development aid (overwrites the outputs): average cycles per iteration -- sse_col_out[b][0..5] = the six
  // phases as seen by wave 0, xsq_col_out[b][w] = row-pass cycles of wave w, xsq_col_out[b][8 + w] = its wait at
  // the first barrier
  __syncthreads();
  if (lane == 0 && m >= 16) {
    if (wave == 0)
      for (int q = 0; q < 6; ++q) a.sse_col_out[(long long)b * m + q] = (real)((double)tacc[q] / (double)n_iter);
    if (wave < 8) {
      a.xsq_col_out[(long long)b * m + wave] = (real)((double)tacc[0] / (double)n_iter);
      a.xsq_col_out[(long long)b * m + 8 + wave] = (real)((double)tacc[2] / (double)n_iter);
    }
  }
#endif
  if (a.update_h) {
    for (int i = threadIdx.x; i < K * MP; i += blockDim.x) {
      const int c = i / MP, j = i % MP;
      if (j < m) Hb[c * m + j] = s.H[i];
    }
  }
  for (int t0 = 0; t0 < lds_rows; t0 += blockDim.x) {  // write the cached rows of W back
    const int t = t0 + threadIdx.x;
    if (t < T && t < lds_rows) {
#pragma unroll
      for (int c = 0; c < K; ++c) Wb[(long long)c * ldw + t] = lds_w[c * lds_stride + t];
    }
  }
}

// =================================================================================================
// Kernel 1b: cooperative multi-workgroup fit (few matrices, each too long for one workgroup to be fast:
// BASELINE config #2, one 16 x 10 000 matrix).  grid = (S, B), launched with hipLaunchCooperativeKernel so
// that all S x B workgroups are co-resident.  Slice s of matrix b owns rows [s*rows_per_slice, ...) for the
// whole fit: its rows of W never leave LDS, its rows of X stay in L2.  Per iteration: W update + partial
// sums of the slice -> one record in global memory -> barrier among the S workgroups of the matrix
// (arrival counter in global memory) -> every workgroup adds the S records in the same fixed order and
// updates its own copy of H.  One exchange of NACC numbers per workgroup and iteration, no kernel launch.
// =================================================================================================
#ifndef HIPNMF_COOP_SPIN_LIMIT
#define HIPNMF_COOP_SPIN_LIMIT (1u << 22)  // ~1 s of polling before a barrier gives up and raises the abort flag
#endif

// barrier number `n` (1, 2, ...) among the `S` workgroups that share `counter`; false after an abort
__device__ __forceinline__ bool coop_barrier(unsigned* counter, unsigned* abort_flag, unsigned S, unsigned n) {
  __shared__ int ok_sh;
  // Every storing wave waits for its own record stores to be acknowledged BEFORE the workgroup barrier: on gfx950
  // __syncthreads() alone is an s_barrier without a vmcnt wait (the sc1 stores could still be in flight when lane 0
  // bumps the counter, and a peer would then sum a stale record).  This is the publish form the hardware guide
  // lists as validated: sc1 stores -> per-wave vmcnt(0) -> workgroup barrier -> one lane's agent-scope atomic add;
  // consumers poll with sc1 loads, pass a workgroup barrier and read the records with sc1 loads only.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    // No release / acquire fences: on a multi-XCD part they mean an L2 write-back / invalidate per barrier
    // (measured ~8 us).  The records exchanged around the barrier are written and read with device-scope
    // relaxed atomics (sc1: straight to / from the coherence point), every wave has waited for its stores to be
    // acknowledged before the barrier above, and the counter itself is only ever touched atomically.
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = S * n;
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0) {
        if (spins > HIPNMF_COOP_SPIN_LIMIT) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = 0;
          break;
        }
      }
    }
    ok_sh = ok;
  }
  __syncthreads();
  return ok_sh != 0;
}

// Head count of the same-XCD mode: all or nothing.  Every slice arrives at `counter`; the arrival that completes the
// count and a slice that gives up waiting race for ONE state word (compare-and-swap 0 -> 1 "go" / 0 -> 2 "abort"), and
// every slice follows whatever that word holds -- the S slices always take the same branch, however a time-out and the
// last arrival interleave.  The wait is short (the workgroups of a cooperative launch start together): a head count
// that fails costs milliseconds, not the ~1 s of the in-fit barriers.
// Scope of the same-XCD flavour's record / flag stores.  Workgroup scope is a global_store with sc0 (coherent in the
// XCD's L2, which is where the peers read it with L1-bypassing loads); -DHIPNMF_COOP_XCD_SCOPE=__HIP_MEMORY_SCOPE_WAVEFRONT
// gives round 2's store without cache-policy bits (measured: no difference, 6.3 us per iteration either way).
#ifndef HIPNMF_COOP_XCD_SCOPE
#define HIPNMF_COOP_XCD_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
#ifndef HIPNMF_COOP_HEAD_SPIN_LIMIT
#define HIPNMF_COOP_HEAD_SPIN_LIMIT (1u << 16)
#endif
__device__ __forceinline__ bool coop_head_count(unsigned* counter, unsigned* state, unsigned* abort_flag, unsigned S) {
  __shared__ int go_sh;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned expected = 0u;
    if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == S)
      __hip_atomic_compare_exchange_strong(state, &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned st, spins = 0;
    while ((st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > HIPNMF_COOP_HEAD_SPIN_LIMIT) {
        expected = 0u;
        __hip_atomic_compare_exchange_strong(state, &expected, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (st != 1u) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    go_sh = st == 1u;
  }
  __syncthreads();
  return go_sh != 0;
}

// Same-XCD flavour (a.coop_xcd): every workgroup of the matrix runs on one XCD, whose L2 all of them share.  Records
// are PLAIN stores (written through this CU's L1 into that L2, where the line stays), every storing wave waits for
// them (vmcnt(0)), then the workgroup barrier, then ONE lane publishes the workgroup's generation number the same way;
// the peers poll the S generation words and read the records with L1-bypassing loads (sc1: served by the L2).  No
// atomic, no write-through to memory, no fabric round trip.
__device__ __forceinline__ bool coop_barrier_xcd(unsigned* flags /* [32 * 32] words: workgroup i owns word 32 i */, unsigned* abort_flag,
                                                 unsigned S, unsigned sl, unsigned n) {
  __shared__ int ok_xcd_sh;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < WAVE) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(flags + 32u * sl, n, __ATOMIC_RELAXED, HIPNMF_COOP_XCD_SCOPE);  // a plain store into the shared L2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int ok = 1;
    unsigned spins = 0;
    const unsigned* mine = flags + 32u * (threadIdx.x < S ? threadIdx.x : 0u);
    while (true) {
      const unsigned v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: L2-served
      if (__all((int)(v >= n))) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0) {
        if (spins > HIPNMF_COOP_SPIN_LIMIT) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__any((int)(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))) {
          ok = 0;
          break;
        }
      }
    }
    if (threadIdx.x == 0) ok_xcd_sh = ok;
  }
  __syncthreads();
  return ok_xcd_sh != 0;
}

// record exchanged between workgroups: device-scope relaxed atomic store / load (no cache holds a stale copy)
template <typename real>
__device__ __forceinline__ void coop_store(real* p, real v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename real>
__device__ __forceinline__ real coop_load(const real* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename real>
__device__ __forceinline__ real block_sum_slices(const real* __restrict__ in, long long stride, int S, int nout,
                                                 real* scratch);
// the same with device-scope atomic loads, four records in flight per thread (cooperative kernel)
template <typename real>
__device__ __forceinline__ void coop_sum_records_atomic(const real* __restrict__ in, int S, int nout, real* scratch,
                                                        real* __restrict__ out) {
  for (int o0 = 0; o0 < nout; o0 += blockDim.x) {
    const int n = (nout - o0 < (int)blockDim.x) ? nout - o0 : (int)blockDim.x;
    const int nq = (int)blockDim.x / n > 0 ? (int)blockDim.x / n : 1;
    const int o = threadIdx.x % n, q = threadIdx.x / n;
    real acc = (real)0;
    if (q < nq) {
      const real* p = in + o0 + o;
      int sl = q;
      for (; sl + 3 * nq < S; sl += 4 * nq) {
        const real v0 = coop_load(p + (long long)sl * nout), v1 = coop_load(p + (long long)(sl + nq) * nout);
        const real v2 = coop_load(p + (long long)(sl + 2 * nq) * nout), v3 = coop_load(p + (long long)(sl + 3 * nq) * nout);
        acc += v0;
        acc += v1;
        acc += v2;
        acc += v3;
      }
      for (; sl < S; sl += nq) acc += coop_load(p + (long long)sl * nout);
    }
    __syncthreads();
    if (q < nq) scratch[q * n + o] = acc;
    __syncthreads();
    if ((int)threadIdx.x < n) {
      real tot = (real)0;
      for (int i = 0; i < nq; ++i) tot += scratch[i * n + threadIdx.x];
      out[o0 + threadIdx.x] = tot;
    }
    __syncthreads();
  }
}

// Records travel as 8-byte granules {32 value bits, generation}, each written with ONE store (plain in the same-XCD
// flavour, device-scope write-through otherwise: 8-byte stores arrive untorn) -- a float is one granule, a double two
// (low and high word, each with the generation).  The consumer needs neither a barrier nor a flag: it polls the
// granules it sums (L1-bypassing loads) until their generation matches, which saves the publish / poll round trip of
// the flag barrier (~0.8 us per iteration).  Same summation order as coop_sum_records_atomic.  False after an abort.
// Generation number of iteration `it`: never 0 (the cleared state of the record buffers) and different for any two
// iterations less than 2^32 - 1 apart, wherever the sequence starts -- in particular two iterations apart, which is the
// age of the granules an exchange overwrites.
__device__ __forceinline__ unsigned coop_generation(unsigned base, int it) {
  return (unsigned)(((unsigned long long)base + (unsigned long long)(unsigned)it) % 0xffffffffull) + 1u;
}
__device__ __forceinline__ unsigned long long coop_poll_granule(const unsigned long long* p, unsigned long long g, unsigned gen,
                                                                unsigned* abort_flag, bool& ok) {
  unsigned spins = 0;
  while ((unsigned)(g >> 32) != gen) {  // not published yet: poll this one
    __builtin_amdgcn_s_sleep(1);
    g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((++spins & 1023u) == 0) {
      if (spins > HIPNMF_COOP_SPIN_LIMIT) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        ok = false;
        break;
      }
    }
  }
  return g;
}
template <typename real>
__device__ __forceinline__ bool coop_sum_records_tagged(const unsigned long long* __restrict__ in, int S, int nout, unsigned gen,
                                                        real* scratch, real* __restrict__ out, unsigned* abort_flag) {
  constexpr int GP = (int)sizeof(real) / 4;  // granules per element
  __shared__ int ok_tag_sh;
  if (threadIdx.x == 0) ok_tag_sh = 1;
  for (int o0 = 0; o0 < nout; o0 += blockDim.x) {
    const int n = (nout - o0 < (int)blockDim.x) ? nout - o0 : (int)blockDim.x;
    const int nq = (int)blockDim.x / n > 0 ? (int)blockDim.x / n : 1;
    const int o = threadIdx.x % n, q = threadIdx.x / n;
    real acc = (real)0;
    bool ok = true;
    if (q < nq) {
      const unsigned long long* p = in + (long long)(o0 + o) * GP;
      const long long stride = (long long)nout * GP;
      for (int s0 = q; s0 < S && ok; s0 += 8 * nq) {
        unsigned long long g[8][GP];
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // all of this thread's granules of the chunk in flight
          const int sl = s0 + u * nq;
#pragma unroll
          for (int w = 0; w < GP; ++w)
            g[u][w] = sl < S ? __hip_atomic_load(p + sl * stride + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sl = s0 + u * nq;
          if (sl < S) {
#pragma unroll
            for (int w = 0; w < GP; ++w) g[u][w] = coop_poll_granule(p + sl * stride + w, g[u][w], gen, abort_flag, ok);
            if constexpr (GP == 1)
              acc += (real)__uint_as_float((unsigned)g[u][0]);
            else
              acc += (real)__longlong_as_double((long long)(((g[u][GP - 1] & 0xffffffffull) << 32) | (g[u][0] & 0xffffffffull)));
          }
        }
      }
    }
    if (!ok) ok_tag_sh = 0;
    __syncthreads();
    if (q < nq) scratch[q * n + o] = acc;
    __syncthreads();
    if ((int)threadIdx.x < n) {
      real tot = (real)0;
      for (int i2 = 0; i2 < nq; ++i2) tot += scratch[i2 * n + threadIdx.x];
      out[o0 + threadIdx.x] = tot;
    }
    __syncthreads();
  }
  return ok_tag_sh != 0;
}

// out[o] = sum over the S records of in[record][o], o < nout (fixed order; nout may exceed the workgroup size)
template <typename real>
__device__ __forceinline__ void coop_sum_records(const real* __restrict__ in, int S, int nout, real* scratch,
                                                 real* __restrict__ out) {
  for (int o0 = 0; o0 < nout; o0 += blockDim.x) {
    const int n = (nout - o0 < (int)blockDim.x) ? nout - o0 : (int)blockDim.x;
    const real t = block_sum_slices<real>(in + o0, nout, S, n, scratch);
    if ((int)threadIdx.x < n) out[o0 + threadIdx.x] = t;
    __syncthreads();
  }
}

// XCD: the same-XCD exchange (a compile-time flavour: with both protocols in one body the register allocation of the
// tile loop suffered -- 14 -> 60 spilled registers, 7.4 -> 9.2 us per iteration)
template <typename real, int G, int CH, int K, bool XCD = false>
__global__ void __launch_bounds__((max_threads<real, G, CH, K>())) HIPNMF_OCC fit_coop_kernel(SolveArgs<real> a) {
  using C = Cfg<real, G, CH, K>;
  constexpr int MP = C::MP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nw = blockDim.x / WAVE;
  Smem<real, G, CH, K> s(smem_raw, nw);
  const int b = blockIdx.y, S = a.S;
  int sl = blockIdx.x;
  unsigned* abort_flag = a.sync + gridDim.y;
  unsigned* xflags = nullptr;
  if constexpr (XCD) {
    // Same-XCD mode: the grid holds 8 S workgroups per matrix; those that run on the XCD of the matrix's workgroup 0
    // draw a ticket, the first S of them are the slices, everybody else leaves.  Placement (observed: workgroups
    // are dealt round-robin over the XCDs) only decides whether S workgroups turn up -- the head count below is
    // taken with the ordinary device-scope barrier BEFORE anything is updated; if it fails the launch aborts with
    // W and H untouched and the host runs the ordinary cooperative kernel.
    __shared__ int sl_sh;
    unsigned* tickets = a.sync + gridDim.y + 2 + 8 * b;        // [B][8] (after the abort flag and the 'updates began' word)
    unsigned* target = a.sync + gridDim.y + 2 + 8 * gridDim.y + b;  // [B]: XCD id + 1 of the matrix
    xflags = a.sync + gridDim.y + 2 + 9 * gridDim.y + 32 * 32 * b;  // [B][32 * 32]
    if (threadIdx.x == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 7u;
      if (blockIdx.x == 0) __hip_atomic_store(target, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned tgt = 0, spins = 0;
      while ((tgt = __hip_atomic_load(target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && ++spins < HIPNMF_COOP_SPIN_LIMIT)
        __builtin_amdgcn_s_sleep(1);
      int mine = -1;
      if (tgt == xcc + 1u) {
        const unsigned t = __hip_atomic_fetch_add(tickets + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < (unsigned)(a.coop_xcd == 2 ? S - 1 : S)) mine = (int)t;  // 2: test hook, one slice stays away
      }
      sl_sh = mine;
    }
    __syncthreads();
    sl = sl_sh;
    if (sl < 0) return;
    unsigned* head_state = a.sync + gridDim.y + 2 + 9 * gridDim.y + 32 * 32 * gridDim.y + b;  // [B] behind the flags
    if (!coop_head_count(a.sync + b, head_state, abort_flag, (unsigned)S)) return;  // S slices present, or nothing happens
    if (sl == 0 && threadIdx.x == 0)  // from here on W and H change: the host may no longer fall back silently
      __hip_atomic_store(abort_flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int g = lane % G;
  const int m = a.m;
  const int row_begin = sl * a.rows_per_slice;  // multiple of blockDim.x
  int T = a.T - row_begin;                      // rows of this slice (slice-local indexing from here on)
  if (T > a.rows_per_slice) T = a.rows_per_slice;
  const real* __restrict__ Xb =
      a.X + (long long)b * a.x_bstride + (x_row_major<G, CH>() ? (long long)row_begin * a.ldx : (long long)row_begin);
  real* __restrict__ Wb = a.W + (long long)b * a.w_bstride + row_begin;
  real* __restrict__ Hb = a.H + (long long)b * K * m;
  unsigned* counter = a.sync + b;
  unsigned nbar = XCD ? 1u : 0u;  // the head count was barrier 1
  // records: device-scope atomics through the fabric, or (same-XCD mode) plain stores kept in the shared L2
  auto put = [&](real* p_, real v_) __attribute__((always_inline)) {
    if constexpr (XCD)  // a plain global store in the ISA (no sc bits), but single-copy atomic and not the compiler's to split or sink
      __hip_atomic_store(p_, v_, __ATOMIC_RELAXED, HIPNMF_COOP_XCD_SCOPE);
    else
      coop_store(p_, v_);
  };
  auto barrier = [&]() __attribute__((always_inline)) -> bool {
    ++nbar;
    if constexpr (XCD)
      return coop_barrier_xcd(xflags, abort_flag, (unsigned)S, (unsigned)sl, nbar);
    else
      return coop_barrier(counter, abort_flag, (unsigned)S, nbar);
  };
  const int row_end = ((T + WAVE - 1) / WAVE) * WAVE;
  // rows [0, lds_rows) of the slice live in LDS for the whole fit (all of them when the slice fits), the rest
  // of W streams from global memory as in the persistent kernel
  real* lds_w = reinterpret_cast<real*>(smem_raw + ((Smem<real, G, CH, K>::bytes(nw) + 15) / 16) * 16);
  const int lds_rows = a.lds_rows < row_end ? a.lds_rows : ((row_end + (int)blockDim.x - 1) / (int)blockDim.x) * (int)blockDim.x;
  real* scratch = lds_w + (long long)K * a.lds_rows;  // [blockDim.x] for the cross-workgroup sums
  const int lds_stride = a.lds_rows;
  MatAddr<real, G, CH, K> ma(Xb, a.ldx, Wb, a.ldw, T, m, lds_w, lds_stride);
  ma.h_lds = s.H;
  ma.lds_used = lds_rows;
  for (int t0 = 0; t0 < lds_rows; t0 += blockDim.x) {
    const int t = t0 + threadIdx.x;
#pragma unroll
    for (int c = 0; c < K; ++c) lds_w[c * lds_stride + t] = (t < T) ? Wb[(long long)c * a.ldw + t] : (real)0;
  }
  load_h_to_lds(s, Hb, m);
  __syncthreads();
  compute_hht(s);
  __syncthreads();
  real h[K][CH], hht[K][K];
  load_h_regs(s, g, h, hht);

  real* __restrict__ gpart = a.part + (long long)b * S * 2 * C::NACC * 2;  // [2][S][NACC] elements of 8 bytes per 32 value bits ({value bits, generation} granules), alternating per exchange
  real* __restrict__ gcol = a.colpart + (long long)b * S * 2 * (2 * MP);  // [2][S][2*MP]
  unsigned nres = 0;
  bool alive = true;
  // ||X - WH||_F^2 per column over all slices -> s.part[0 .. 2*MP) in every workgroup (same order everywhere)
  auto residual_all = [&]() {
    block_residual<real, G, CH, K>(s, ma, 0, row_end, h);
    real* mine = gcol + ((long long)(nres & 1) * S + sl) * (2 * MP);
    if (threadIdx.x < 2 * MP) put(mine + threadIdx.x, s.part[threadIdx.x]);
    alive = barrier() && alive;
    coop_sum_records_atomic<real>(gcol + (long long)(nres & 1) * S * (2 * MP), S, 2 * MP, scratch, s.part);
    ++nres;
  };
  auto total_err = [&]() -> real {
    real tot = (real)0;
    for (int j = 0; j < MP; ++j) tot += s.part[j];
    return sqrt_(tot);
  };

  real err0 = (real)0, prev = (real)0;
  if (a.tol > (real)0) {
    residual_all();
    err0 = total_err();
    prev = err0;
    __syncthreads();
  }
  int n_iter = 0;
#ifdef HIPNMF_TIMING  // development build (tools/coop_phase_timing.py): shader-clock cycles per phase of an iteration, summed
  unsigned long long cacc[6] = {0, 0, 0, 0, 0, 0};
#define COOP_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define COOP_STAMP(v) ((void)0)
#endif
  RowTile<real, G, CH, K> tiles_lds[PipeDepth<true, G, CH>::value];
  prefetch_head<real, G, CH, K, true>(tiles_lds, ma, 0, lds_rows);
  for (int it = 1; it <= a.max_iter && alive; ++it) {
    n_iter = it;
    COOP_STAMP(c0);
    real accA[K][CH], accB[C::NB];
#pragma unroll
    for (int c = 0; c < K; ++c)
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) accA[c][cc] = (real)0;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) accB[i] = (real)0;
    rows_update_pass<real, G, CH, K, true, true>(ma, 0, lds_rows, h, hht, accA, accB, a.l1w, a.l2w, a.update_h != 0,
                                                 tiles_lds);
    if (lds_rows < row_end) {
      RowTile<real, G, CH, K> tiles_glb[PipeDepth<false, G, CH>::value];
      rows_update_pass<real, G, CH, K, false, false>(ma, lds_rows, row_end, h, hht, accA, accB, a.l1w, a.l2w,
                                                     a.update_h != 0, tiles_glb);
    }
    if (it < a.max_iter) prefetch_head<real, G, CH, K, true>(tiles_lds, ma, 0, lds_rows);
    if (a.update_h) {
      COOP_STAMP(c1);  // row pass done (this wave)
      wave_reduce_acc<real, G, CH, K>(s.part + wave * C::NACC, accA, accB);
      __syncthreads();
      COOP_STAMP(c2);  // wave records in LDS, workgroup barrier passed
      {
        // tagged granules {32 value bits, generation}: ONE 8-byte store each (plain into the shared L2, or written through
        // to the coherence point in the device-scope flavour), no barrier, no flag
        constexpr int GP = (int)sizeof(real) / 4;
        // (gpart is 256-byte aligned workspace and every granule index is a whole number of 8-byte elements)
        static_assert(sizeof(unsigned long long) == 8 && alignof(unsigned long long) == 8, "granules are 8-byte objects");
        unsigned long long* g64 = reinterpret_cast<unsigned long long*>(gpart) + (long long)(it & 1) * S * C::NACC * GP;
        const unsigned gen = coop_generation(a.gen_base, it);
        const unsigned long long tag = (unsigned long long)gen << 32;
        for (int i = threadIdx.x; i < C::NACC; i += blockDim.x) {
          real acc = s.part[i];
          for (int w = 1; w < nw; ++w) acc += s.part[w * C::NACC + i];
          unsigned long long bits;
          if constexpr (GP == 1)
            bits = (unsigned long long)__float_as_uint((float)acc);
          else
            bits = (unsigned long long)__double_as_longlong((double)acc);
#pragma unroll
          for (int w = 0; w < GP; ++w) {
            const unsigned long long gran = tag | ((bits >> (32 * w)) & 0xffffffffull);
            unsigned long long* dst = g64 + ((long long)sl * C::NACC + i) * GP + w;
            // ONE 8-byte store either way (formally an atomic: the tag and the value can neither tear nor be merged, split
            // or sunk by the compiler).  Workgroup scope = a plain global_store_dwordx2 that stays in the XCD's L2, agent
            // scope = the same store written through to the coherence point.
            if constexpr (XCD)
              __hip_atomic_store(dst, gran, __ATOMIC_RELAXED, HIPNMF_COOP_XCD_SCOPE);
            else
              __hip_atomic_store(dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        __syncthreads();  // s.part is about to be overwritten by the sums
        COOP_STAMP(c3);  // this workgroup's record published (tagged granules)
        alive = coop_sum_records_tagged<real>(g64, S, C::NACC, gen, scratch, s.part, abort_flag) && alive;
        COOP_STAMP(c4);  // every slice's record polled and summed
#ifdef HIPNMF_TIMING
        cacc[0] += c1 - c0; cacc[1] += c2 - c1; cacc[2] += c3 - c2; cacc[3] += c4 - c3;
#endif
      }
      COOP_STAMP(c5);
      if (wave == 0) wave0_combine_and_update_h(s, 1, m, a.l1h, a.l2h);
      __syncthreads();
      COOP_STAMP(c6);  // H updated, workgroup barrier passed
      load_h_regs(s, g, h, hht);
      COOP_STAMP(c7);
#ifdef HIPNMF_TIMING
      cacc[4] += c6 - c5; cacc[5] += c7 - c6;
#endif
    }
    if (a.tol > (real)0 && (it % a.check_every) == 0) {
      residual_all();
      const real err = total_err();
      __syncthreads();
      if ((prev - err) / err0 < a.tol) break;
      prev = err;
    }
  }
  residual_all();
  if (sl == 0) {
    if (threadIdx.x == 0) {
      if (a.err_out) a.err_out[b] = total_err();
      if (a.n_iter_out) a.n_iter_out[b] = n_iter;
    }
    if (threadIdx.x < m) {
      if (a.sse_col_out) a.sse_col_out[(long long)b * m + threadIdx.x] = s.part[threadIdx.x];
      if (a.xsq_col_out) a.xsq_col_out[(long long)b * m + threadIdx.x] = s.part[MP + threadIdx.x];
    }
#ifdef HIPNMF_TIMING
    // (overwrites the outputs) average cycles per iteration of the six phases as seen by wave 0 of slice 0, then by its last wave
    __syncthreads();
    if (lane == 0 && m >= 12 && (wave == 0 || wave == nw - 1) && a.sse_col_out && a.xsq_col_out) {
      real* dst = wave == 0 ? a.sse_col_out : a.xsq_col_out;
      for (int q = 0; q < 6; ++q) dst[(long long)b * m + q] = (real)((double)cacc[q] / (double)n_iter);
    }
#endif
    if (a.update_h) {
      for (int i = threadIdx.x; i < K * MP; i += blockDim.x) {
        const int c = i / MP, j = i % MP;
        if (j < m) Hb[c * m + j] = s.H[i];
      }
    }
  }
  for (int t0 = 0; t0 < lds_rows; t0 += blockDim.x) {  // this slice's rows of W back to global memory
    const int t = t0 + threadIdx.x;
    if (t < T) {
#pragma unroll
      for (int c = 0; c < K; ++c) Wb[(long long)c * a.ldw + t] = lds_w[c * lds_stride + t];
    }
  }
#undef COOP_STAMP
}

// =================================================================================================
// Kernels 2..5: multi-slice path (few matrices / very long T; also the time-sharded building blocks).
//   grid = (S, B); slice s owns rows [s*rows_per_slice, (s+1)*rows_per_slice)
// =================================================================================================
template <typename real, int G, int CH, int K>
__global__ void __launch_bounds__((max_threads<real, G, CH, K>())) HIPNMF_OCC slice_pass_kernel(SolveArgs<real> a) {
  using C = Cfg<real, G, CH, K>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nw = blockDim.x / WAVE;
  Smem<real, G, CH, K> s(smem_raw, nw);
  const int b = blockIdx.y, sl = blockIdx.x;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;  // matrix already converged
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int g = lane % G;
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  const real* __restrict__ Hb = a.H + (long long)b * K * a.m;
  const int row_begin = sl * a.rows_per_slice;
  int row_end = row_begin + a.rows_per_slice;
  const int t_pad = ((a.T + WAVE - 1) / WAVE) * WAVE;
  if (row_end > t_pad) row_end = t_pad;

  load_h_to_lds(s, Hb, a.m);
  __syncthreads();
  compute_hht(s);
  __syncthreads();
  real h[K][CH], hht[K][K];
  load_h_regs(s, g, h, hht);
  real accA[K][CH], accB[C::NB];
#pragma unroll
  for (int c = 0; c < K; ++c)
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) accA[c][cc] = (real)0;
#pragma unroll
  for (int i = 0; i < C::NB; ++i) accB[i] = (real)0;
  MatAddr<real, G, CH, K> ma(Xb, a.ldx, Wb, a.ldw, a.T, a.m);
  ma.h_lds = s.H;
  RowTile<real, G, CH, K> tiles_glb[PipeDepth<false, G, CH>::value];
  rows_update_pass<real, G, CH, K, false, false>(ma, row_begin, row_end, h, hht, accA, accB, a.l1w, a.l2w,
                                                 a.update_h != 0, tiles_glb);
  if (!a.update_h) return;
  wave_reduce_acc<real, G, CH, K>(s.part + wave * C::NACC, accA, accB);
  __syncthreads();
  real* __restrict__ out = a.part + ((long long)b * a.S + sl) * C::NACC;
  if (!a.fuse_h) {
    for (int i = threadIdx.x; i < C::NACC; i += blockDim.x) {
      real acc = s.part[i];
      for (int w = 1; w < nw; ++w) acc += s.part[w * C::NACC + i];
      out[i] = acc;
    }
    return;
  }
  // Single-GPU sliced path: one launch per iteration.  The workgroup that arrives last at the matrix's counter
  // owns the H update ("last block" reduction).  As in the cooperative kernel the exchange is fence-free: the
  // records travel as device-scope relaxed atomics and the ticket is an atomic, so no workgroup has to write
  // its XCD's L2 (full of freshly updated W) back just to publish 95 numbers.  The records are still added in
  // slice order, so the result does not depend on which workgroup happens to be last.
  for (int i = threadIdx.x; i < C::NACC; i += blockDim.x) {
    real acc = s.part[i];
    for (int w = 1; w < nw; ++w) acc += s.part[w * C::NACC + i];
    coop_store(out + i, acc);
  }
  __shared__ int is_last;
  __shared__ real scratch[HIPNMF_MAXNT];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores are acknowledged (see coop_barrier)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(a.sync + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = ticket == (unsigned)a.S - 1u;
    if (is_last) __hip_atomic_store(a.sync + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
  }
  __syncthreads();
  if (!is_last) return;
  coop_sum_records_atomic<real>(a.part + (long long)b * a.S * C::NACC, a.S, C::NACC, scratch, s.part);
  for (int i = threadIdx.x; i < C::NACC; i += blockDim.x) {
    const real acc = s.part[i];
    if (i < K * C::MP) {
      s.A[i] = acc;
    } else {
      int idx = i - K * C::MP, c = 0;
      while (idx >= K - c) {
        idx -= K - c;
        ++c;
      }
      const int c2 = c + idx;
      s.B[c * K + c2] = acc;
      s.B[c2 * K + c] = acc;
    }
  }
  __syncthreads();
  h_update_lds(s, a.m, a.l1h, a.l2h);  // s.H still holds the H this launch started from
  real* __restrict__ Hout = a.H + (long long)b * K * a.m;
  for (int i = threadIdx.x; i < K * C::MP; i += blockDim.x) {
    const int c = i / C::MP, j = i % C::MP;
    if (j < a.m) Hout[c * a.m + j] = s.H[i];
  }
}

// Fixed-order parallel sum of S slice records of `nout` values each (record stride `stride`):
// thread (q, o) adds slices q, q+nq, q+2nq, ... of output o; the nq partial sums are then added in order.
// scratch: blockDim.x values of LDS.  Result for output o is returned to threads o < nout (others get 0).
template <typename real>
__device__ __forceinline__ real block_sum_slices(const real* __restrict__ in, long long stride, int S, int nout,
                                                 real* scratch) {
  const int nq = blockDim.x / nout > 0 ? blockDim.x / nout : 1;
  const int o = threadIdx.x % nout, q = threadIdx.x / nout;
  real acc = (real)0;
  if (q < nq) {
    // eight records in flight per thread, added in the order of the plain loop (a thread's ~S / nq loads were strictly
    // serial: 39 us for 1018 slices on the time-shard path, profiles/r02_c_kernel_stats_bench_config5.csv)
    const real* p = in + o;
    int sl = q;
    for (; sl + 7 * nq < S; sl += 8 * nq) {
      real v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long long)(sl + u * nq) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sl < S; sl += nq) acc += p[(long long)sl * stride];
  }
  __syncthreads();
  if (q < nq) scratch[q * nout + o] = acc;
  __syncthreads();
  real tot = (real)0;
  if (threadIdx.x < nout)
    for (int i = 0; i < nq; ++i) tot += scratch[i * nout + threadIdx.x];
  return tot;
}

// sums[b] = [ W^T X (k x m) | W^T W (k x k, full) ] = fixed-order sum of the slice records
template <typename real, int G, int CH, int K>
__global__ void reduce_slices_kernel(SolveArgs<real> a) {
  using C = Cfg<real, G, CH, K>;
  __shared__ real scratch[1024];
  __shared__ real tot[C::NACC];
  const int b = blockIdx.x, m = a.m;
  const real* __restrict__ in = a.part + (long long)b * a.S * C::NACC;
  real* __restrict__ out = a.sums + (long long)b * (K * m + K * K);
  const real t = block_sum_slices<real>(in, C::NACC, a.S, C::NACC, scratch);
  if (threadIdx.x < C::NACC) tot[threadIdx.x] = t;
  __syncthreads();
  for (int i = threadIdx.x; i < K * m + K * K; i += blockDim.x) {
    int src;
    if (i < K * m) {
      src = (i / m) * C::MP + (i % m);
    } else {
      int c = (i - K * m) / K, c2 = (i - K * m) % K;
      if (c > c2) {
        const int tmp = c;
        c = c2;
        c2 = tmp;
      }
      src = K * C::MP + c * K - c * (c - 1) / 2 + (c2 - c);
    }
    out[i] = tot[src];
  }
}

// H update of matrix b; one block per matrix.  a.sums != nullptr: from the already summed
// [W^T X | W^T W] record (time-shard path, after the caller's all-reduce); otherwise straight from the S slice
// records of slice_pass_kernel (fixed-order sum, saves the reduce launch on the single-GPU sliced path).
template <typename real, int G, int CH, int K>
__global__ void hupdate_kernel(SolveArgs<real> a) {
  using C = Cfg<real, G, CH, K>;
  constexpr int MP = G * CH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem<real, G, CH, K> s(smem_raw, 1);
  const int b = blockIdx.x, m = a.m;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  real* __restrict__ Hb = a.H + (long long)b * K * m;
  load_h_to_lds(s, Hb, m);
  if (a.sums) {
    const real* __restrict__ in = a.sums + (long long)b * (K * m + K * K);
    for (int i = threadIdx.x; i < K * MP; i += blockDim.x) {
      const int c = i / MP, j = i % MP;
      s.A[i] = (j < m) ? in[c * m + j] : (real)0;
    }
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) s.B[i] = in[K * m + i];
  } else {
    const real* __restrict__ in = a.part + (long long)b * a.S * C::NACC;
    __shared__ real scratch[1024];
    const real acc = block_sum_slices<real>(in, C::NACC, a.S, C::NACC, scratch);
    {
      const int i = threadIdx.x;
      if (i >= C::NACC) {
      } else if (i < K * MP) {
        s.A[i] = acc;
      } else {
        int idx = i - K * MP, c = 0;
        while (idx >= K - c) {
          idx -= K - c;
          ++c;
        }
        const int c2 = c + idx;
        s.B[c * K + c2] = acc;
        s.B[c2 * K + c] = acc;
      }
    }
  }
  __syncthreads();
  h_update_lds(s, m, a.l1h, a.l2h);
  for (int i = threadIdx.x; i < K * MP; i += blockDim.x) {
    const int c = i / MP, j = i % MP;
    if (j < m) Hb[c * m + j] = s.H[i];
  }
}

// per-slice residual partials: colpart[b][s][0..MP) = sse, [MP..2MP) = xsq
template <typename real, int G, int CH, int K>
__global__ void __launch_bounds__((max_threads<real, G, CH, K>())) slice_resid_kernel(SolveArgs<real> a) {
  constexpr int MP = G * CH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nw = blockDim.x / WAVE;
  Smem<real, G, CH, K> s(smem_raw, nw);
  const int b = blockIdx.y, sl = blockIdx.x;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0 && a.it >= 0) return;
  const int lane = threadIdx.x & (WAVE - 1);
  const int g = lane % G;
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  const real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  const real* __restrict__ Hb = a.H + (long long)b * K * a.m;
  const int row_begin = sl * a.rows_per_slice;
  int row_end = row_begin + a.rows_per_slice;
  const int t_pad = ((a.T + WAVE - 1) / WAVE) * WAVE;
  if (row_end > t_pad) row_end = t_pad;
  load_h_to_lds(s, Hb, a.m);
  __syncthreads();
  real h[K][CH];
#pragma unroll
  for (int c = 0; c < K; ++c)
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) h[c][cc] = s.H[c * MP + g * CH + cc];
  MatAddr<real, G, CH, K> ma(Xb, a.ldx, Wb, a.ldw, a.T, a.m);
  ma.h_lds = s.H;
  block_residual<real, G, CH, K>(s, ma, row_begin, row_end, h);
  real* __restrict__ out = a.colpart + ((long long)b * a.S + sl) * (2 * MP);
  if (threadIdx.x < 2 * MP) out[threadIdx.x] = s.part[threadIdx.x];
}

// combine slice residuals; a.it < 0: final (write outputs), a.it == 0: error at init, a.it > 0: stop test
template <typename real, int G, int CH, int K>
__global__ void resid_finalize_kernel(SolveArgs<real> a) {
  constexpr int MP = G * CH;
  __shared__ real col[2 * MP];
  const int b = blockIdx.x, m = a.m;
  real* st = a.state ? a.state + (long long)b * 8 : nullptr;
  if (st && st[3] != (real)0 && a.it >= 0) return;
  const real* __restrict__ in = a.colpart + (long long)b * a.S * (2 * MP);
  __shared__ real scratch[1024];
  const real colsum = block_sum_slices<real>(in, 2 * MP, a.S, 2 * MP, scratch);
  if (threadIdx.x < 2 * MP) col[threadIdx.x] = colsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    real tot = (real)0;
    for (int j = 0; j < MP; ++j) tot += col[j];
    const real err = sqrt_(tot);
    if (a.it < 0) {
      if (a.err_out) a.err_out[b] = err;
      if (a.n_iter_out && (!st || st[3] == (real)0)) a.n_iter_out[b] = a.max_iter;
    } else if (a.it == 0) {
      st[0] = err;
      st[1] = err;
      st[2] = err;
      st[3] = (real)0;
      st[4] = (real)0;
    } else {
      // the iteration number is kept on the device so that a replayed hipGraph needs no new arguments
      const int it_now = ((int)st[4] + 1) * a.check_every;
      st[4] = st[4] + (real)1;
      st[2] = err;
      if ((st[1] - err) / st[0] < a.tol) {
        st[3] = (real)1;
        if (a.n_iter_out) a.n_iter_out[b] = it_now;
      }
      st[1] = err;
    }
  }
  if (a.it < 0 && threadIdx.x < m) {
    if (a.sse_col_out) a.sse_col_out[(long long)b * m + threadIdx.x] = col[threadIdx.x];
    if (a.xsq_col_out) a.xsq_col_out[(long long)b * m + threadIdx.x] = col[MP + threadIdx.x];
  }
}

// =================================================================================================
// layout conversion (once per fit, not per iteration)
// =================================================================================================
// any X layout -> row-major [T][ld_out] with the channels zero-padded to ld_out (the row-per-lane instance's layout)
template <typename real>
__global__ void x_to_row_major_kernel(const real* __restrict__ in, long long in_bstride, long long ld_in,
                                      int in_layout, real* __restrict__ out, long long out_bstride, int ld_out,
                                      int T, int m) {
  __shared__ real tile[32][33];
  const int b = blockIdx.z;
  const real* __restrict__ ib = in + (long long)b * in_bstride;
  real* __restrict__ ob = out + (long long)b * out_bstride;
  const int t0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  if (in_layout == 0) {  // already row-major: strided copy with channel padding
    for (int tt = threadIdx.y; tt < 32; tt += blockDim.y) {
      const int t = t0 + tt, j = j0 + threadIdx.x;
      if (t < T && j < ld_out) ob[(long long)t * ld_out + j] = (j < m) ? ib[(long long)t * ld_in + j] : (real)0;
    }
    return;
  }
  for (int jj = threadIdx.y; jj < 32; jj += blockDim.y) {  // channel-major in: coalesced along t
    const int j = j0 + jj, t = t0 + threadIdx.x;
    tile[jj][threadIdx.x] = (j < m && t < T) ? ib[(long long)j * ld_in + t] : (real)0;
  }
  __syncthreads();
  for (int tt = threadIdx.y; tt < 32; tt += blockDim.y) {
    const int t = t0 + tt, j = j0 + threadIdx.x;
    if (t < T && j < ld_out) ob[(long long)t * ld_out + j] = tile[threadIdx.x][tt];
  }
}

// X row-major [T][ldx_in] -> channel-major [m][ldx_out] (zero padded to ldx_out)
template <typename real>
__global__ void x_to_channel_major_kernel(const real* __restrict__ in, long long in_bstride, long long ld_in,
                                          int in_layout, real* __restrict__ out, long long out_bstride,
                                          long long ld_out, int T, int m) {
  __shared__ real tile[32][33];
  const int b = blockIdx.z;
  const real* __restrict__ ib = in + (long long)b * in_bstride;
  real* __restrict__ ob = out + (long long)b * out_bstride;
  const int t0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  if (in_layout == 1) {  // already channel-major: strided copy with padding
    for (int jj = threadIdx.y; jj < 32; jj += blockDim.y) {
      const int j = j0 + jj, t = t0 + threadIdx.x;
      if (j < m && t < ld_out) ob[(long long)j * ld_out + t] = (t < T) ? ib[(long long)j * ld_in + t] : (real)0;
    }
    return;
  }
  for (int tt = threadIdx.y; tt < 32; tt += blockDim.y) {
    const int t = t0 + tt, j = j0 + threadIdx.x;
    tile[tt][threadIdx.x] = (t < T && j < m) ? ib[(long long)t * ld_in + j] : (real)0;
  }
  __syncthreads();
  for (int jj = threadIdx.y; jj < 32; jj += blockDim.y) {
    const int j = j0 + jj, t = t0 + threadIdx.x;
    if (j < m && t < ld_out) ob[(long long)j * ld_out + t] = tile[threadIdx.x][jj];
  }
}

// W row-major [T][k] <-> component-major [k][ldw]; dir 0: rm -> cm (zero pad), 1: cm -> rm
template <typename real>
__global__ void w_convert_kernel(real* __restrict__ rm, real* __restrict__ cm, long long ldw, int T, int k, int dir) {
  const int b = blockIdx.y;
  real* __restrict__ r = rm + (long long)b * T * k;
  real* __restrict__ c = cm + (long long)b * k * ldw;
  const long long n = (long long)k * ldw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int comp = (int)(i / ldw);
    const long long t = i % ldw;
    if (dir == 0)
      c[i] = (t < T) ? r[t * k + comp] : (real)0;
    else if (t < T)
      r[t * k + comp] = c[i];
  }
}

}  // namespace hipnmf
