// hipnmf_wide.hip -- host driver of the wide-shape kernels (nmf_wide.hpp): n_features up to 128 and n_components up
// to 32 (float64 with more than 16 components: up to 64 channels), fp32 and fp64, reached from hipnmf_fit_batched_* / hipnmf_fit_ragged_* for every shape outside the narrow
// kernel set of hipnmf_api.hip (n_features > 32 or n_components > 8).  The reference accepts any
// 1 <= n_components <= n_features (src/muscle_synergies/analysis.py:829-846, :862-863); sklearn's solver is shape-
// agnostic (sklearn/decomposition/_nmf.py:540-554, 638-640).
//
// Layout canonicalisation, once per fit: X row-major with 16-byte aligned rows (a caller's C-order X with
// n_features * sizeof(real) % 16 == 0 is streamed in place), W row-major with rows of ks = round_up(k, 4) values
// (a caller's row-major W with k % 4 == 0 is updated in place).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hipnmf_internal.hpp"
#include "nmf_wide4_inst.hpp"
#include "nmf_wide_inst.hpp"
#include "nmf_big.hpp"  // (the one-pass kernel of nmf_big1.hpp is instantiated in inst_big1_f32.hip and reached through big1_kernel_f32)

using namespace hipnmf;

namespace hipnmf {
static int wide_mp(int m) { return m <= 16 ? 16 : m <= 32 ? 32 : m <= 48 ? 48 : m <= 64 ? 64 : m <= 96 ? 96 : m <= 128 ? 128 : 0; }
const WideKernel<float>* wide_kernel_f32(int m, int k, int nw) {
  // 129..256 channels with at most 16 components (HD-EMG grids), fp32: the one-pass kernel still holds its accumulators in one
  // wave (inst_wide_f32_xl.hip; one wave per SIMD).  HIPNMF_WIDE_XL=0: the two-pass general-shape kernels instead.
  static const bool xl = [] {
    const char* e = getenv("HIPNMF_WIDE_XL");
    return !(e && atoi(e) == 0);
  }();
  if (xl && m > 128 && m <= 256 && k <= 16 && nw == 4) return wide_kernel_f32_xl(m <= 160 ? 160 : m <= 192 ? 192 : 256);
  const int MP = wide_mp(m);
  if (!MP || k > 32) return nullptr;
  if (k > 16) return nw == 4 ? wide_kernel_f32_k32(MP < 32 ? 32 : MP) : nullptr;
  return MP <= 48 ? wide_kernel_f32_lo(MP, 16, nw) : wide_kernel_f32_hi(MP, 16, nw);
}
const WideKernel<double>* wide_kernel_f64(int m, int k, int nw) {
  const int MP = wide_mp(m);
  if (!MP || k > 32) return nullptr;
  if (k > 16) return nw == 4 ? wide_kernel_f64_k32(MP < 32 ? 32 : MP) : nullptr;
  return MP <= 48 ? wide_kernel_f64_lo(MP, 16, nw) : wide_kernel_f64_hi(MP, 16, nw);
}
// fp32, 33..128 channels, at most 8 components: the v_mfma_f32_4x4x1 formulation (nmf_wide4.hpp), which pads the components to
// a multiple of 4 instead of 16
const WideKernel<float>* wide4_kernel_f32(int m, int k, int nw) {
  // (9..12 components as three component quads were built and measured too: 222..234 registers, two waves per SIMD, and no
  //  faster than the 16x16x4 kernel -- 4096 x (64 x 2 500), k = 12: 6.71 vs 6.95 M matrix-it/s; 48 channels, k = 10: 7.64 vs 8.23)
  if (m > 128 || k > 8) return nullptr;
  const int MP = m <= 16 ? 16 : m <= 32 ? 32 : m <= 48 ? 48 : m <= 64 ? 64 : m <= 96 ? 96 : 128;
  const int KQ = k <= 4 ? 1 : 2;
  if (MP == 16) return wide4_kernel_f32_16(KQ, nw);
  if (MP == 32) return wide4_kernel_f32_32(KQ, nw);
  return MP <= 64 ? wide4_kernel_f32_lo(MP, KQ, nw) : wide4_kernel_f32_hi(MP, KQ, nw);
}
}  // namespace hipnmf

namespace {
template <typename real>
const WideKernel<real>* pick4(int, int, int) {
  return nullptr;
}
template <>
const WideKernel<float>* pick4<float>(int m, int k, int nw) {
  return wide4_kernel_f32(m, k, nw);
}
// float64, 33..128 channels, at most 8 components: v_mfma_f64_4x4x4 (nmf_wide4d.hpp); 4 or 8 waves up to 64 channels, 4 beyond
template <>
const WideKernel<double>* pick4<double>(int m, int k, int nw) {
  if (m > 128 || k > 8) return nullptr;
  if (m > 64) return wide4d_kernel_f64_hi(m <= 96 ? 96 : 128, k <= 4 ? 1 : 2, 4);  // one wave per SIMD, 256 threads
  return wide4d_kernel_f64(m <= 16 ? 16 : m <= 32 ? 32 : m <= 48 ? 48 : 64, k <= 4 ? 1 : 2, nw == 4 ? 4 : 8);
}
template <typename real>
const WideKernel<real>* pick(int m, int k, int nw);
template <>
const WideKernel<float>* pick<float>(int m, int k, int nw) {
  return wide_kernel_f32(m, k, nw);
}
template <>
const WideKernel<double>* pick<double>(int m, int k, int nw) {
  return wide_kernel_f64(m, k, nw);
}
}  // namespace

// The one-pass update + record kernel of the general shapes (nmf_big1.hpp; round 5), both losses: fp32 everywhere, float64 up to
// 256 channels x 32 components.  HIPNMF_BIG1=0 keeps the two-pass pair big_pass_w_kernel + big_records_kernel, which remains the
// path of the instances that do not fit LDS: float64 beyond that, Kullback-Leibler fp32 with 48 / 64 padded components on more
// than 256 channels and float64 with 32 on more than 128.
static bool big1_enabled() {
  static const bool on = [] {
    const char* e = getenv("HIPNMF_BIG1");
    return !(e && atoi(e) == 0);
  }();
  return on;
}
template <typename real>
static const Big1Kernel<real>* big1_table(int KPb, int MPb);
template <>
const Big1Kernel<float>* big1_table<float>(int KPb, int MPb) {
  return big1_kernel_f32(KPb, MPb);
}
template <>
const Big1Kernel<double>* big1_table<double>(int KPb, int MPb) {
  return big1_kernel_f64(KPb, MPb);
}
template <typename real>
static const Big1Kernel<real>* pick_big1(hipnmf_handle* h, int KPb, int MPb, bool kl) {
  if (!big1_enabled()) return nullptr;
  const Big1Kernel<real>* b1 = big1_table<real>(KPb, MPb);
  if (!b1) return nullptr;
  if (kl) return (b1->fn_kl && b1->smem_kl <= (size_t)h->lds_per_block) ? b1 : nullptr;  // (instances whose two layouts of H do not fit: two-pass)
  return b1->smem <= (size_t)h->lds_per_block ? b1 : nullptr;
}

// Slices per matrix for the one-pass kernel: one workgroup fits a CU, so the launch runs in ceil(B S / CUs) waves of workgroups of
// round_up(T / S, 64) rows each (+ ~96 rows' worth of prologue and record); the cheapest S, the smaller one on a tie.
// 64 x (512 x 10 000), k = 32 on 256 CUs: S = 4 (one wave of 2 560-row slices) 30.4 ms per 50 iterations, S = 8 31.9, S = 2 51.9.
static long long big1_slices(int num_cu, int B, long long T) {
  const long long s_max = std::max<long long>(1, std::min<long long>((T + 63) / 64, (4LL * num_cu + B - 1) / B));
  long long best = 1, best_cost = -1;
  for (long long s = 1; s <= s_max; ++s) {
    const long long rows = round_up((T + s - 1) / s, 64);
    const long long cost = (((long long)B * s + num_cu - 1) / num_cu) * (rows + 96);
    if (best_cost < 0 || cost < best_cost) best = s, best_cost = cost;
  }
  return best;
}

// Fitted on MI355X, ms per 100 iterations (tools/probes/kl_long_ab.sh, kl_long_narrow_ab.sh; profiles/r05_kl_long_matrices_ab.log):
//  * one workgroup per matrix, whatever the batch up to one matrix per CU: (0.35 + 0.0265 m) [float64: 0.3 + 0.06 m] per 1 000 rows on the
//    4x4 kernels; the lane mappings' own kernels are faster on few channels (fit_batched_impl passes their rate);
//  * row slices: 1.0 (10 us per iteration of launches) + 0.015 per slice of a matrix (the H update sums the slices' records) +
//    0.005 [x 1.9] per row of a slice and wave of workgroups, x 1.6 at 128 channels.
// Checked against: fp32 1 x (64 x 100 000), k = 8: 183.5 -> 8.0; 32 x (64 x 5 000): 9.5 -> 3.8; 64 x (32 x 2 500): 2.8 -> 3.3 (not taken);
// 1 x (16 x 10 000), k = 5 on the lane mappings: 3.6 -> 3.3 (not taken), k = 8: 5.6 -> 3.3; 1 x (4 x 5 000), k = 2: 0.5 -> 2.1 (not taken);
// float64 1 x (128 x 5 000), k = 6: 53.8 -> 4.0; 128 x (128 x 10 000): 108.5 -> 60.7; 128 x (64 x 2 500), k = 8: 10.9 -> 11.5 (not taken).
// `max_slices` > 0: the caller's cap on slices per matrix (hipnmf_set_tuning) -- the launch applies it, so the estimate must too (round-5
// advisor finding: max_slices = 2..N used to be priced as if the uncapped slice count ran).
bool hipnmf_kl_row_sliced_wins(bool f64, int m, long long T, int B, int num_cu, double t_one_per_krow, int max_slices) {
  if (B > num_cu) return false;
  const hipnmf_route_table& rt = hipnmf_routes();
  if (t_one_per_krow < 0) t_one_per_krow = f64 ? rt.kl_one_f64_a + rt.kl_one_f64_b * m : rt.kl_one_f32_a + rt.kl_one_f32_b * m;
  const double t_one = (double)T * 1e-3 * t_one_per_krow;
  long long S = big1_slices(num_cu, B, T);
  if (max_slices > 0) S = std::min<long long>(S, max_slices);
  const long long rows = round_up((T + S - 1) / S, 64);
  const double waves = (double)(((long long)B * S + num_cu - 1) / num_cu);
  const double mp64 = (double)round_up(m, 64) / 64.0;
  const double t_rows = rt.kl_sliced_launches + rt.kl_sliced_per_slice * (double)S +
                        rt.kl_sliced_per_row * (f64 ? rt.kl_sliced_f64_factor : 1.0) * (1.0 + rt.kl_sliced_wide_factor * (mp64 - 1.0)) * (double)rows * waves;
  return t_rows < rt.kl_sliced_margin * t_one;
}

// `p` has passed validate() of hipnmf_api.hip.  `ragged`: host copy of the caller's descriptors or nullptr.
template <typename real>
int hipnmf_fit_wide(hipnmf_handle* h, const hipnmf_problem* p, const real* X, real* W, real* H, real* err_out,
                    int32_t* n_iter_out, real* sse_col_out, real* xsq_col_out, const int64_t* ragged) {
  const int B = p->batch, m = p->n_features, k = p->n_components;
  const int Br = std::max(B, h->path_batch_hint);  // the batch the ROUTE is chosen for (hipnmf_set_batch_hint: chunks of one batch)
  const long long T = p->n_samples;
  // Instance: 512 threads (one workgroup per CU, two waves per SIMD) with every byte of LDS the stages leave as W cache,
  // wherever that instance exists; 256 threads (two workgroups per CU, small cache) otherwise, or on request
  // (hipnmf_set_tuning threads = 256 / 512; HIPNMF_LDS_W=0 turns the cache off).
  // Measured (tools/quick_bench.py, 1024 x T = 10 000, 100 iterations, TB/s algorithmic, 512 vs 256 threads): 64 ch k = 8
  // 5.70 / 5.83, 64 ch k = 16 6.31 / 6.12, 128 ch k = 16 5.57 / 5.39, 32 ch k = 12 5.76 / 5.94 -- the bigger cache pays
  // once W is a third of the traffic or the stages of two workgroups leave no room for a cache at all.
  const int ks_ = (int)round_up(k, 4);
  const bool kl = p->loss == HIPNMF_LOSS_KL;  // the Kullback-Leibler flavour exists as the 256-thread instance
  const bool want512 = !kl && (h->threads == 512 || (h->threads == 0 && ks_ >= 12 && m > 48));
  const WideKernel<real>* wk = want512 ? pick<real>(m, k, 8) : nullptr;
  if (!wk) wk = pick<real>(m, k, 4);
  // beyond the instances of nmf_wide.hpp -- more than 128 channels or 32 components, or float64 with more than 16 components on
  // more than 64 channels (its LDS footprint) -- the general-shape kernels of nmf_big.hpp take over (HIPNMF_FORCE_BIG=1: always)
  static const bool force_big = [] {
    const char* e = getenv("HIPNMF_FORCE_BIG");
    return e && atoi(e) != 0;
  }();
  // Kullback-Leibler has no row-sliced form on the kernels of nmf_wide.hpp / nmf_wide4.hpp: a few long matrices would each sit on ONE
  // workgroup.  The one-pass general-shape kernel is row-sliced by construction (components padded to 16, channels to 64) and fills
  // the chip with them: chosen where its cost model wins (round 5; tools/probes/kl_long_ab.sh, ms per 100 iterations, one workgroup per
  // matrix -> row slices: fp32 1 x (64 x 100 000), k = 8: 183.5 -> 8.0; 32 x (64 x 5 000): 9.5 -> 3.8; 64 x (32 x 2 500): 2.8 -> 3.3;
  // float64 1 x (128 x 5 000), k = 6: 53.8 -> 4.0; 128 x (128 x 10 000): 108.5 -> 60.7; 128 x (64 x 2 500), k = 8: 10.9 -> 11.5).
  // (cost model: hipnmf_kl_row_sliced_wins above; HIPNMF_KL_SLICED=0: never)
  bool kl_sliced = false;
  if (kl && !ragged && wk && h->variant == 0 && h->max_slices != 1 && !force_big) {  // (hipnmf_set_tuning(max_slices = 1): one workgroup per matrix)
    static const bool kl_sliced_env = [] {
      const char* e = getenv("HIPNMF_KL_SLICED");
      return !(e && e[0] == '0');
    }();
    kl_sliced = kl_sliced_env && hipnmf_kl_row_sliced_wins(sizeof(real) == 8, m, T, Br, h->num_cu, -1.0, h->max_slices) &&
                pick_big1<real>(h, (int)round_up(k, 16), (int)round_up(m, 16), true) != nullptr;
  }
  const bool big = !wk || wk->smem > (size_t)h->lds_per_block || force_big || kl_sliced;
  if (big) {
    if (m > HIPNMF_MAX_FEATURES || k > HIPNMF_MAX_COMPONENTS)
      return fail(HIPNMF_ERR_UNSUPPORTED, "no kernel for n_features=%d (max %d) n_components=%d (max %d)", m, HIPNMF_MAX_FEATURES, k,
                  HIPNMF_MAX_COMPONENTS);
    if (ragged) {
      // trials of unequal length on the general-shape kernels: one chip-filling row-sliced fit per trial (the slices of ONE
      // matrix already fill the chip there, so nothing is lost against a common launch).  The padding rows of a packed matrix
      // are zero by the entry point's contract and inert under the updates (0 * x / eps), so trial b is the uniform problem
      // with n_samples = ld_b.
      if (p->x_layout != HIPNMF_X_CHANNEL_MAJOR || p->w_layout != HIPNMF_W_COMPONENT_MAJOR)
        return fail(HIPNMF_ERR_UNSUPPORTED, "ragged batches use the packed native layouts (channel-major X, component-major W)");
      float total_ms = 0.0f;
      for (int b = 0; b < B; ++b) {
        const int64_t* d = ragged + 4 * (size_t)b;
        if (d[0] < 1 || d[0] > p->n_samples || d[2] < d[0] || d[1] < 0 || d[3] < 0 || (d[2] % 4) != 0)
          return fail(HIPNMF_ERR_BAD_ARG, "bad ragged descriptor for matrix %d (T=%lld, xoff=%lld, ld=%lld, woff=%lld)", b,
                      (long long)d[0], (long long)d[1], (long long)d[2], (long long)d[3]);
        hipnmf_problem q = *p;
        q.batch = 1;
        q.n_samples = d[2];
        q.ldx = d[2];
        q.x_batch_stride = (int64_t)m * d[2];
        const int rc = hipnmf_fit_wide<real>(h, &q, X + d[1], W + d[3], H + (size_t)b * k * m, err_out ? err_out + b : nullptr,
                                             n_iter_out ? n_iter_out + b : nullptr, sse_col_out ? sse_col_out + (size_t)b * m : nullptr,
                                             xsq_col_out ? xsq_col_out + (size_t)b * m : nullptr, nullptr);
        if (rc) return rc;
        total_ms += h->last_ms;
      }
      h->last_ms = total_ms;
      return HIPNMF_OK;
    }
  }
  if (h->variant == 3 || h->variant == 5 || h->variant == 6)
    return fail(HIPNMF_ERR_UNSUPPORTED, "tuning variant %d does not exist for wide shapes (n_features=%d, n_components=%d)",
                h->variant, m, k);
  constexpr int VEC = 16 / (int)sizeof(real);
  const int KPb = (int)round_up(k, 16), MPb = (int)round_up(m, 16);  // the general-shape kernels' padded sizes
  const int ks = big ? KPb : (int)round_up(k, 4);                    // row length of the kernel-side W
  hipStream_t st = h->stream;

  if (ragged) {
    if (p->x_layout != HIPNMF_X_CHANNEL_MAJOR || p->w_layout != HIPNMF_W_COMPONENT_MAJOR)
      return fail(HIPNMF_ERR_UNSUPPORTED, "ragged batches use the packed native layouts (channel-major X, component-major W)");
    for (int b = 0; b < B; ++b) {
      const int64_t* d = ragged + 4 * (size_t)b;
      if (d[0] < 1 || d[0] > p->n_samples || d[2] < d[0] || d[1] < 0 || d[3] < 0)
        return fail(HIPNMF_ERR_BAD_ARG, "bad ragged descriptor for matrix %d (T=%lld, xoff=%lld, ld=%lld, woff=%lld)", b,
                    (long long)d[0], (long long)d[1], (long long)d[2], (long long)d[3]);
    }
  }

  // ---- canonical layouts ----------------------------------------------------------------------------------------
  const bool aligned = (reinterpret_cast<uintptr_t>(X) % 16) == 0 && ((p->x_batch_stride * (long long)sizeof(real)) % 16) == 0;
  const bool x_inplace = !ragged && p->x_layout == HIPNMF_X_ROW_MAJOR && (m % VEC) == 0 && (p->ldx % VEC) == 0 && aligned;
  const long long ldx_c = x_inplace ? p->ldx : round_up(m, 4);
  const bool w_inplace = !ragged && p->w_layout == HIPNMF_W_ROW_MAJOR && ks == k && (reinterpret_cast<uintptr_t>(W) % 16) == 0;
  if ((T + 16) * ldx_c * (long long)sizeof(real) >= (1LL << 31) || (T + 16) * (long long)ks * (long long)sizeof(real) >= (1LL << 31))
    return fail(HIPNMF_ERR_UNSUPPORTED, "one matrix needs >= 2 GiB of X or W; the engine addresses < 2 GiB per matrix");

  // ragged: distinct source matrices are converted once (the restarts of one trial share theirs)
  struct Src {
    long long xoff, T, ld, roff;
  };
  std::vector<Src> srcs;
  std::vector<long long> kdesc;  // kernel descriptors {T_b, X offset, -, W offset}
  long long ragged_x_elems = 0, ragged_w_elems = 0;
  if (ragged) {
    kdesc.resize(4 * (size_t)B);
    for (int b = 0; b < B; ++b) {
      const int64_t* d = ragged + 4 * (size_t)b;
      size_t idx = srcs.size();
      for (size_t q = 0; q < srcs.size(); ++q)
        if (srcs[q].xoff == d[1] && srcs[q].T == d[0] && srcs[q].ld == d[2]) {
          idx = q;
          break;
        }
      if (idx == srcs.size()) {
        srcs.push_back({(long long)d[1], (long long)d[0], (long long)d[2], ragged_x_elems});
        ragged_x_elems += (long long)d[0] * ldx_c;
      }
      kdesc[4 * (size_t)b + 0] = d[0];
      kdesc[4 * (size_t)b + 1] = srcs[idx].roff;
      kdesc[4 * (size_t)b + 2] = 0;
      kdesc[4 * (size_t)b + 3] = ragged_w_elems;
      ragged_w_elems += (long long)d[0] * ks;
    }
  }

  const int KPb0 = (int)round_up(k, 16), MPb0 = (int)round_up(m, 16);
  // ---- path: one workgroup per matrix, or row slices over the whole chip (few long matrices: the reference's own
  // single-DataFrame call)
  const WideKernel<real>* wk4 = big ? nullptr : pick<real>(m, k, 4);
  int S = 1;
  long long rps = 0;
  bool sliced = false;
  if (big) {  // always row slices: enough of them to fill the chip, whole 64-row sweeps of a workgroup
    const long long want = std::max<long long>(1, (2LL * h->num_cu + Br - 1) / Br);
    long long s_try = std::min<long long>(want, (T + 63) / 64);
    if (pick_big1<real>(h, KPb0, MPb0, kl)) s_try = big1_slices(h->num_cu, Br, T);  // one 512-thread workgroup per CU
    if (h->max_slices > 0) s_try = std::min<long long>(s_try, h->max_slices);
    rps = round_up((T + s_try - 1) / s_try, 64);
    S = (int)((T + rps - 1) / rps);
    sliced = true;
    if (B > 65535) return fail(HIPNMF_ERR_UNSUPPORTED, "batch=%d: the general-shape path takes at most 65535 matrices", B);
  } else if (!ragged && !kl && wk4 && wk4->smem <= (size_t)h->lds_per_block && h->variant != 1 && h->variant != 4 && B <= 65535) {
    const long long target = std::min<long long>(128, std::max<long long>(1, 2LL * h->num_cu / Br));  // (the H update sums S records per launch)
    long long s_try = std::min<long long>(target, (T + 127) / 128);  // at least two subtiles per wave and slice
    if (h->max_slices > 0) s_try = std::min<long long>(s_try, h->max_slices);
    s_try = std::max<long long>(s_try, 1);
    rps = round_up((T + s_try - 1) / s_try, 16);
    S = (int)((T + rps - 1) / rps);
    // fitted to tools/quick_bench.py --batch 1 --m 64 --k 8 (us per iteration, one workgroup / sliced): T = 1 000 20 / 12.7,
    // 4 000 71 / 11, 10 000 171 / 20.6, 100 000 1 692 / 28.7; 8 x (128 x 20 000), k = 16: 480 / 39
    const double unit = 1.1e-6 * (double)wk4->MP / 64.0;  // one round of four 16-row subtiles on a workgroup that has its CU to itself
    const double t_pers = 3e-6 + (double)((Br + 2 * h->num_cu - 1) / (2 * h->num_cu)) * (double)((T + 63) / 64) * unit * 4.0 / (double)wk->NW;
    const double t_sliced = 11e-6 + (double)((rps + 63) / 64) * unit;  // two graph-replayed launches: the pass, the record sums + H update
    sliced = h->variant == 2 ? S >= 1 : (S >= 2 && t_sliced < 0.85 * t_pers);
  } else if (h->variant == 2) {
    return fail(HIPNMF_ERR_UNSUPPORTED, "the row-sliced wide path handles uniform Frobenius batches only");
  }
  if (sliced && !big) {
    wk = wk4;
    // the 256-thread instance of the 4x4 kernels where one exists (at most 8 components, Frobenius): same slice records
    // (KP = 4 or 8 rows instead of 16), same phases (HIPNMF_WIDE4_SLICED=0: the 16x16x4 kernel)
    static const bool sliced4 = [] {
      const char* e = getenv("HIPNMF_WIDE4_SLICED");
      const char* e4 = getenv("HIPNMF_WIDE4");
      return !(e && e[0] == '0') && !(e4 && e4[0] == '0');
    }();
    // up to 32 channels (reached from fit_batched_impl's routing of small batches since round 5): float64 yes, fp32 no difference
    // (tools/probes/sliced4_narrow_ab.sh, ms per 200 iterations, 16x16x4 -> 4x4: float64 32 x (32 x 3 000), k = 8: 3.3 -> 2.7; 1 x 10 000 rows: 4.0 -> 3.0;
    //  2 x (24 x 30 000), k = 7: 5.3 -> 3.7; k = 3: 3.1 -> 2.0; fp32 32 x (32 x 3 000), k = 8: 2.2 -> 2.3; 100 x 10 000 rows: 10.2 -> 9.4).
    // HIPNMF_WIDE4_SLICED_NARROW=0 / 1: never / for fp32 too
    static const int sliced4_narrow_env = [] {
      const char* e = getenv("HIPNMF_WIDE4_SLICED_NARROW");
      return e ? atoi(e) : -1;
    }();
    const bool sliced4_narrow = sliced4_narrow_env < 0 ? sizeof(real) == 8 : sliced4_narrow_env != 0;
    const WideKernel<real>* w4s = (sliced4 && !kl && (m > HIPNMF_NARROW_MAX_FEATURES || sliced4_narrow)) ? pick4<real>(m, k, 4) : nullptr;
    if (w4s && w4s->NW == 4 && w4s->smem <= (size_t)h->lds_per_block) wk = w4s;
  }
  // one workgroup per matrix, at most 8 components: the 4x4x1 / 4x4x4 formulation (HIPNMF_WIDE4=0: the 16x16x4 one); the
  // Kullback-Leibler loss on its 256-thread instances (fp32 33..128 channels: round 4; fp32 up to 32 channels and float64: round 5),
  // Frobenius on all of them
  if (!sliced) {
    static const bool use4 = [] {
      const char* e = getenv("HIPNMF_WIDE4");
      return !(e && e[0] == '0');
    }();
    // waves per workgroup: 12 (three per SIMD: the instance stays within 168 registers up to 64 channels) where it exists,
    // unless the caller asks for 4 / 8 (threads = 256 / 512).  tools/quick_bench.py, M matrix-it/s with 4 / 8 / 12 waves:
    // 1024 x (64 x 10 000), k = 8: 1.80 / 1.84 / 1.93; 4096 x (64 x 2 500), k = 8: 7.89 / 8.69 / 9.29; 48 ch, k = 6: 9.25 / 10.15 / 10.68;
    // k <= 4 (one component quad, at the X stream's rate with 8 waves already: 11.4 M at 64 x 2 500): 8
    const WideKernel<real>* w4 = nullptr;
    if (use4) {
      // Kullback-Leibler: 4 waves, two workgroups per CU, for chip-filling batches; 8 waves where a CU gets one matrix at most.
      // tools/probes/kl_waves_ab.sh, ms per 100 iterations, 4 -> 8 waves: 1 x (32 x 2 500), k = 8, float64: 7.5 -> 5.7; 16 x (24 x 1 000), k = 6:
      // 3.2 -> 2.5; 4096 x (64 x 2 500), k = 8, fp32: 83.4 -> 81.9; 8192 x (32 x 300): 12.3 -> 15.7; float64 8192 x (32 x 128): 13.1 -> 15.5
      // (HIPNMF_KL_WAVES=4 / 8 pins it)
      static const int kl_waves_env = [] {
        const char* e = getenv("HIPNMF_KL_WAVES");
        return e ? atoi(e) : 0;
      }();
      const int kl_waves = kl_waves_env ? kl_waves_env : (Br <= h->num_cu ? 8 : 4);
      const int want = kl ? (h->threads == 256 ? 4 : kl_waves) : h->threads == 256 ? 4 : h->threads == 512 ? 8 : (h->threads == 768 || k > 4) ? 12 : 8;
      w4 = pick4<real>(m, k, want);
      if (!w4 && want == 12) w4 = pick4<real>(m, k, 8);
      if (kl && !(w4 && w4->fn_kl && w4->smem <= (size_t)h->lds_per_block)) w4 = pick4<real>(m, k, 4);
      if (kl && !(w4 && w4->fn_kl)) w4 = nullptr;
    }
    if (w4 && w4->smem <= (size_t)h->lds_per_block) wk = w4;
  }
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t x_elems = (size_t)T * (size_t)ldx_c, w_elems = (size_t)T * (size_t)ks;  // per matrix (uniform batches)
  const size_t o_x = x_inplace ? 0 : carve(sizeof(real) * (ragged ? (size_t)ragged_x_elems + 64 : (size_t)B * x_elems + 64));
  const size_t o_w = w_inplace ? 0 : carve(sizeof(real) * (ragged ? (size_t)ragged_w_elems + 64 : (size_t)B * w_elems + 64));
  const size_t o_kdesc = ragged ? carve(sizeof(long long) * 4 * (size_t)B) : 0;
  const size_t o_cdesc = ragged ? carve(sizeof(long long) * 4 * (size_t)B) : 0;
  const int recKP = big ? KPb : wk->KP, recMP = big ? MPb : wk->MP;  // (`wk` is not consulted on the general-shape path)
  const size_t rec = (size_t)recKP * recMP + (size_t)recKP * recKP;
  const size_t o_part = sliced ? carve(sizeof(real) * (size_t)B * S * rec) : 0;
  const size_t o_col = sliced ? carve(sizeof(real) * (size_t)B * S * 3 * recMP) : 0;  // (3: sse | xsq | KL per column of the general-shape path)
  const size_t o_state = sliced ? carve(sizeof(real) * (size_t)B * 8) : 0;
  const size_t o_hht = big ? carve(sizeof(real) * (size_t)B * KPb * KPb) : 0;
  const int n_hblk = (m + 63) / 64;
  const size_t o_hhtp = big ? carve(sizeof(real) * (size_t)B * n_hblk * KPb * KPb) : 0;
  int rc = hipnmf_ensure_ws(h, std::max<size_t>(off, 256));
  if (rc) return rc;
  char* ws = static_cast<char*>(h->ws);

  WideArgs<real> a;
  std::memset(&a, 0, sizeof(a));
  if (x_inplace) {
    a.X = X;
    a.x_bstride = p->x_batch_stride;
    a.ldx = p->ldx;
  } else {
    real* xc = reinterpret_cast<real*>(ws + o_x);
    dim3 blk(32, 8);
    if (ragged) {
      for (const Src& r : srcs) {
        dim3 grd((unsigned)((r.T + 31) / 32), (unsigned)((ldx_c + 31) / 32), 1u);
        HIPNMF_LAUNCH(x_to_row_major_kernel<real>, grd, blk, 0, st, X + r.xoff, 0LL, r.ld, (int)HIPNMF_X_CHANNEL_MAJOR,
                           xc + r.roff, 0LL, (int)ldx_c, (int)r.T, m);
      }
    } else {
      for (int b0 = 0; b0 < B; b0 += 65535) {  // the batch rides on grid.z (HIP limit 65535)
        const unsigned nb = (unsigned)std::min(65535, B - b0);
        dim3 grd((unsigned)((T + 31) / 32), (unsigned)((ldx_c + 31) / 32), nb);
        HIPNMF_LAUNCH(x_to_row_major_kernel<real>, grd, blk, 0, st, X + (long long)b0 * p->x_batch_stride,
                           (long long)p->x_batch_stride, (long long)p->ldx, (int)p->x_layout, xc + (size_t)b0 * x_elems,
                           (long long)x_elems, (int)ldx_c, (int)T, m);
      }
    }
    a.X = xc;
    a.x_bstride = (long long)x_elems;
    a.ldx = ldx_c;
  }
  const long long* d_cdesc = nullptr;  // device copies: caller's descriptors, kernel descriptors
  const long long* d_kdesc = nullptr;
  if (ragged) {
    HIP_TRY(hipMemcpyAsync(ws + o_cdesc, ragged, sizeof(long long) * 4 * (size_t)B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ws + o_kdesc, kdesc.data(), sizeof(long long) * 4 * (size_t)B, hipMemcpyHostToDevice, st));
    d_cdesc = reinterpret_cast<const long long*>(ws + o_cdesc);
    d_kdesc = reinterpret_cast<const long long*>(ws + o_kdesc);
    a.ragged = d_kdesc;
  }
  real* wc = w_inplace ? W : reinterpret_cast<real*>(ws + o_w);
  auto convert_w = [&](int dir) {
    const long long n = T * ks;
    for (int b0 = 0; b0 < B; b0 += 65535) {  // the batch rides on grid.y
      const int nb = std::min(65535, B - b0);
      dim3 grd((unsigned)std::min<long long>((n + 255) / 256, 1024), (unsigned)nb);
      if (ragged)
        HIPNMF_LAUNCH(wide_w_convert_kernel<real>, grd, dim3(256), 0, st, W, 1, 0LL, 0LL, wc, 0LL, ks, (int)T, k, dir,
                           d_cdesc + 4LL * b0, d_kdesc + 4LL * b0);
      else if (p->w_layout == HIPNMF_W_ROW_MAJOR)
        HIPNMF_LAUNCH(wide_w_convert_kernel<real>, grd, dim3(256), 0, st, W + (long long)b0 * T * k, 0, (long long)T * k, 0LL,
                           wc + (size_t)b0 * w_elems, (long long)w_elems, ks, (int)T, k, dir, (const long long*)nullptr,
                           (const long long*)nullptr);
      else
        HIPNMF_LAUNCH(wide_w_convert_kernel<real>, grd, dim3(256), 0, st, W + (long long)b0 * T * k, 1, (long long)T * k,
                           (long long)T, wc + (size_t)b0 * w_elems, (long long)w_elems, ks, (int)T, k, dir,
                           (const long long*)nullptr, (const long long*)nullptr);
    }
  };
  if (!w_inplace) convert_w(0);
  a.W = wc;
  a.w_bstride = (long long)w_elems;
  a.H = H;
  a.err_out = err_out;
  a.n_iter_out = n_iter_out;
  a.sse_col_out = sse_col_out;
  a.xsq_col_out = xsq_col_out;
  a.T = (int)T;
  a.m = m;
  a.k = k;
  a.ks = ks;
  a.xchunks = (m + VEC - 1) / VEC;
  a.max_iter = p->max_iter;
  a.check_every = p->check_every;
  a.update_h = p->update_h ? 1 : 0;
  a.tol = (real)p->tol;
  a.l1w = (real)p->l1_reg_W;
  a.l2w = (real)p->l2_reg_W;
  a.l1h = (real)p->l1_reg_H;
  a.l2h = (real)p->l2_reg_H;

  // ---- driver of the row-sliced paths: `enqueue(n, check, emit)` queues n iterations (+ one stop-rule evaluation), `residual(it,
  // emit)` one residual evaluation (it: 0 at init, 1 a check, -1 the final outputs); every launch goes through an emitter --
  // straight onto the stream, or into the replayed chain of kernel nodes (hipnmf_kernel_chain in hipnmf_internal.hpp says why
  // not a stream capture).  The done flags live on the device (state[b][3]).
  const bool stop_rule = p->tol > 0;
  real* d_state = sliced ? reinterpret_cast<real*>(ws + o_state) : nullptr;
  auto direct = [&](auto fn, dim3 g, dim3 blk, size_t sm, const auto& args) { HIPNMF_LAUNCH(fn, g, blk, sm, st, args); };
  auto drive = [&](auto&& enqueue, auto&& residual) -> int {
    std::vector<real> host_state((size_t)B * 8);
    auto all_converged = [&](bool* done) -> int {
      HIP_TRY(hipMemcpyAsync(host_state.data(), d_state, sizeof(real) * (size_t)B * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      *done = true;
      for (int b = 0; b < B; ++b) *done = *done && host_state[(size_t)b * 8 + 3] != (real)0;
      return HIPNMF_OK;
    };
    if (stop_rule) residual(0, direct);
    const int chunk = stop_rule ? p->check_every : std::min(p->max_iter, 64);
    int it_done = 0;
    bool converged = false;
    if (h->use_graph && p->max_iter >= 2 * chunk) {
      hipnmf_kernel_chain chain;
      hipError_t ge = hipSuccess;
      enqueue(chunk, stop_rule, [&](auto fn, dim3 g, dim3 blk, size_t sm, const auto& args) {
        if (ge == hipSuccess) ge = chain.add(reinterpret_cast<const void*>(fn), g, blk, sm, args);
      });
      if (ge == hipSuccess) ge = chain.instantiate();
      int graph_rc = HIPNMF_OK;
      while (ge == hipSuccess && !converged && it_done + chunk <= p->max_iter) {
        ge = chain.launch(st);
        if (ge != hipSuccess) break;
        it_done += chunk;
        if (stop_rule) {
          graph_rc = all_converged(&converged);
          if (graph_rc) break;
        }
      }
      if (ge == hipSuccess && !graph_rc) ge = hipStreamSynchronize(st);
      if (graph_rc) return graph_rc;
      if (ge != hipSuccess) {
        (void)hipGetLastError();
        if (it_done > 0)
          return fail(HIPNMF_ERR_HIP, "hipGraph replay failed after %d iterations: %s", it_done, hipGetErrorString(ge));
      }
    }
    while (!converged && it_done < p->max_iter) {
      const int n = std::min(chunk, p->max_iter - it_done);
      const bool check = stop_rule && n == chunk;
      enqueue(n, check, direct);
      it_done += n;
      if (check) {
        int crc = all_converged(&converged);
        if (crc) return crc;
      }
    }
    residual(-1, direct);
    return HIPNMF_OK;
  };
  WideSliceArgs<real> sa;
  std::memset(&sa, 0, sizeof(sa));
  if (sliced) {
    HIP_TRY(hipMemsetAsync(d_state, 0, sizeof(real) * (size_t)B * 8, st));  // zeroed: nothing is done yet (also read without a stop rule)
    sa.H = H;
    sa.part = reinterpret_cast<real*>(ws + o_part);
    sa.colpart = reinterpret_cast<real*>(ws + o_col);
    sa.state = d_state;
    sa.err_out = err_out;
    sa.n_iter_out = n_iter_out;
    sa.sse_col_out = sse_col_out;
    sa.xsq_col_out = xsq_col_out;
    sa.m = m;
    sa.k = k;
    sa.MP = recMP;
    sa.KP = recKP;
    sa.S = S;
    sa.max_iter = p->max_iter;
    sa.check_every = p->check_every;
    sa.tol = (real)p->tol;
    sa.l1h = (real)p->l1_reg_H;
    sa.l2h = (real)p->l2_reg_H;
    sa.kl = (big && kl) ? 1 : 0;
  }

  if (big) {
    // ---- general shapes (nmf_big.hpp): four launches per iteration over row slices, H staged in LDS in blocks of CBH channels
    HIP_TRY(hipEventRecord(h->ev0, st));
    h->last_path = 2;
    snprintf(h->last_kernel, sizeof(h->last_kernel), "big_pass_w_kernel<%s,%d>[sliced]", sizeof(real) == 4 ? "float" : "double", KPb);
    BigArgs<real> ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.X = a.X;
    ba.x_bstride = a.x_bstride;
    ba.ldx = a.ldx;
    ba.W = wc;
    ba.w_bstride = (long long)w_elems;
    ba.H = H;
    ba.HHt = reinterpret_cast<real*>(ws + o_hht);
    ba.part = reinterpret_cast<real*>(ws + o_part);
    ba.colpart = reinterpret_cast<real*>(ws + o_col);
    ba.state = d_state;
    ba.T = (int)T;
    ba.m = m;
    ba.k = k;
    ba.KP = KPb;
    ba.MP = MPb;
    ba.xchunks = a.xchunks;
    ba.S = S;
    ba.rows_per_slice = (int)rps;
    ba.l1w = (real)p->l1_reg_W;
    ba.l2w = (real)p->l2_reg_W;
    ba.kl = kl ? 1 : 0;
    ba.update_h = a.update_h;
    const Big1Kernel<real>* b1 = pick_big1<real>(h, KPb, MPb, kl);
    const auto b1fn = b1 ? (kl ? b1->fn_kl : b1->fn) : nullptr;
    const size_t b1smem = b1 ? (kl ? b1->smem_kl : b1->smem) : 0;
    if (b1) snprintf(h->last_kernel, sizeof(h->last_kernel), "%s[sliced]", kl ? b1->name_kl : b1->name);
    ba.hht_part = reinterpret_cast<real*>(ws + o_hhtp);
    ba.n_hblk = n_hblk;
    // H in LDS: all of it when [KP][MP + 4] (+ H H^T, + the residual's column accumulators) fits 96 KiB, else blocks of channels
    const size_t fixed = sizeof(real) * ((size_t)KPb * (KPb + 4) + 12 * (size_t)MPb);
    const size_t cap = 96 * 1024;
    int cbh = MPb;
    while (cbh > 16 && fixed + sizeof(real) * (size_t)KPb * (cbh + 4) > cap) cbh = (int)round_up(cbh / 2, 16);
    ba.CBH = cbh;
    const size_t smem_w = sizeof(real) * ((size_t)KPb * (cbh + 4) + (size_t)KPb * (KPb + 4));
    const size_t smem_r = sizeof(real) * ((size_t)KPb * (cbh + 4) + 4 * (size_t)(kl ? 3 : 2) * MPb);
    const size_t smem_rec = sizeof(real) * std::max<size_t>(4 * (16 * (size_t)(KPb + 4) + 16 * (size_t)(BIG_CB + 4)) + (kl ? (size_t)KPb * (BIG_CB + 4) : 0),
                                                            4 * (size_t)KPb * BIG_CB);  // (stages [+ the block of H of the Kullback-Leibler quotient] | reduction buffer)
    const size_t smem_h = sizeof(real) * ((size_t)k * k + 128 * (size_t)k);
    BigHArgs<real> hb;
    hb.H = H;
    hb.part = ba.part;
    hb.state = d_state;
    hb.m = m;
    hb.k = k;
    hb.S = S;
    hb.rec = KPb * MPb + KPb * KPb;
    hb.ldA = MPb;
    hb.offB = KPb * MPb;
    hb.ldB = KPb;
    hb.l1h = (real)p->l1_reg_H;
    hb.l2h = (real)p->l2_reg_H;
    hb.kl = kl ? 1 : 0;
    hb.hht_part = b1 ? reinterpret_cast<real*>(ws + o_hhtp) : nullptr;  // one-pass kernel: the H update leaves the next H H^T behind
    hb.KP = KPb;
    const dim3 gslice(S, B), grec(S, B, (MPb + BIG_CB - 1) / BIG_CB + 1), ghup(B, (m + 63) / 64);
    auto with_kp = [&](auto&& f) {  // the padded component count is a compile-time parameter of three of the kernels
      switch (KPb) {
        case 16: f(std::integral_constant<int, 16>{}); break;
        case 32: f(std::integral_constant<int, 32>{}); break;
        case 48: f(std::integral_constant<int, 48>{}); break;
        default: f(std::integral_constant<int, 64>{}); break;
      }
    };
    int arc = HIPNMF_OK;
    with_kp([&](auto kp) {
      constexpr int KP = decltype(kp)::value;
      for (const void* fn : {reinterpret_cast<const void*>(big_pass_w_kernel<real, KP>), reinterpret_cast<const void*>(big_records_kernel<real, KP>),
                             reinterpret_cast<const void*>(big_resid_kernel<real, KP>)})
        if (!arc) arc = hipnmf_allow_full_lds(h, fn);
    });
    if (!arc && b1 && b1smem > 48 * 1024) arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(b1fn));
    if (!arc && smem_h > 48 * 1024) arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(big_hupdate_kernel<real>));
    if (arc) return arc;
    const Big1Kernel<real>* b1r = pick_big1<real>(h, KPb, MPb, false);  // its residual kernel serves both losses
    {  // only the kernels this call will launch count (round-5 advisor finding: the two-pass pair's footprint used to refuse calls
       // that the one-pass kernel -- smaller LDS, a residual kernel with none -- serves entirely)
      size_t need = smem_h;
      if (b1) need = std::max(need, b1smem);
      else need = std::max(need, std::max(smem_w, smem_rec));
      if (!b1r) need = std::max(need, smem_r);
      if (need > (size_t)h->lds_per_block)
        return fail(HIPNMF_ERR_UNSUPPORTED, "n_features=%d n_components=%d needs more LDS per workgroup than this device has (%d bytes)", m, k,
                    h->lds_per_block);
    }
    auto residual = [&](int it, auto&& emit) {
      if (b1r)
        emit(b1r->resid, gslice, dim3(512), (size_t)0, ba);
      else
        with_kp([&](auto kp) { emit(big_resid_kernel<real, decltype(kp)::value>, gslice, dim3(256), smem_r, ba); });
      WideSliceArgs<real> f = sa;
      f.it = it;
      emit(big_resid_finalize_kernel<real>, dim3(B), dim3(1024), sizeof(real) * 3 * (size_t)MPb, f);
    };
    auto enqueue = [&](int n, bool check, auto&& emit) {
      for (int i = 0; i < n; ++i) {
        if (b1) {  // one pass: the W update and the slice's record together (H H^T comes with the H update)
          emit(b1fn, gslice, dim3(512), b1smem, ba);
        } else {
          emit(big_hht_kernel<real>, dim3(B, KPb), dim3(256), (size_t)0, ba);
          with_kp([&](auto kp) {
            constexpr int KP = decltype(kp)::value;
            emit(big_pass_w_kernel<real, KP>, gslice, dim3(256), smem_w, ba);
            if (a.update_h) emit(big_records_kernel<real, KP>, grec, dim3(256), smem_rec, ba);
          });
        }
        if (a.update_h) emit(big_hupdate_kernel<real>, ghup, dim3(256), smem_h, hb);
      }
      if (check) residual(1, emit);
    };
    if (b1) HIPNMF_LAUNCH(big_hht_part_kernel<real>, ghup, dim3(256), sizeof(real) * 64 * (size_t)k, st, hb);  // H H^T of the initial H
    rc = drive(enqueue, residual);
    if (rc) return rc;
  } else {
  // W cache: whole 16-row subtiles in what LDS is left (512 threads: of the whole CU; 256 threads: of half of it, so that
  // two workgroups stay resident); none in the sliced mode (a slice lives for one pass)
  size_t smem = wk->smem;
  if (!sliced) {
    const size_t lds_cap = (h->lds_budget > 0 ? (size_t)h->lds_budget : (size_t)h->lds_per_block) / (wk->NW >= 8 ? 1 : 2);
    long long rows = 0;
    if (h->use_lds_w && lds_cap > smem + 1024) rows = (long long)((lds_cap - smem - 256) / (sizeof(real) * (size_t)ks)) / 16 * 16;
    rows = std::min<long long>(rows, round_up(T, 16));
    a.lds_rows = (int)rows;
    smem += sizeof(real) * (size_t)ks * (size_t)rows;
  }
  HIP_TRY(hipEventRecord(h->ev0, st));
  const auto kern = kl ? wk->fn_kl : wk->fn;
  if (!kern) return fail(HIPNMF_ERR_UNSUPPORTED, "no Kullback-Leibler instance of %s", wk->name);
  if (smem > 48 * 1024 && (rc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(kern)))) return rc;
  if (!sliced) {
    h->last_path = 1;
    snprintf(h->last_kernel, sizeof(h->last_kernel), "%s", kl ? wk->name_kl : wk->name);
    HIPNMF_LAUNCH(kern, dim3(B), dim3(wk->NW * 64), smem, st, a);
  } else {
    // ---- row-sliced: per iteration one pass over all slices (mode 1) and the H update; the stop rule's residual (mode 2 +
    // finalize) every check_every iterations with the done flags on the device; chunks of iterations replayed as a hipGraph
    h->last_path = 2;
    snprintf(h->last_kernel, sizeof(h->last_kernel), "%s[sliced]", wk->name);
    WideArgs<real> pa = a, ra = a;
    pa.mode = 1;
    ra.mode = 2;
    pa.S = ra.S = S;
    pa.rows_per_slice = ra.rows_per_slice = (int)rps;
    pa.part = reinterpret_cast<real*>(ws + o_part);
    ra.colpart = reinterpret_cast<real*>(ws + o_col);
    pa.state = ra.state = d_state;
    const dim3 grid(B, S), block(wk->NW * 64);
    const size_t hsmem = sizeof(real) * (rec + (size_t)k * m);
    if (hsmem > 48 * 1024 && (rc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(wide_hupdate_kernel<real>)))) return rc;
    auto residual = [&](int it, auto&& emit) {
      emit(kern, grid, block, smem, ra);
      WideSliceArgs<real> f = sa;
      f.it = it;
      emit(wide_resid_finalize_kernel<real>, dim3(B), dim3(1024), (size_t)0, f);
    };
    auto enqueue = [&](int n, bool check, auto&& emit) {
      for (int i = 0; i < n; ++i) {
        emit(kern, grid, block, smem, pa);
        if (a.update_h) emit(wide_hupdate_kernel<real>, dim3(B), dim3(1024), hsmem, sa);
      }
      if (check) residual(1, emit);
    };
    rc = drive(enqueue, residual);
    if (rc) return rc;
  }
  }
  HIP_TRY(hipEventRecord(h->ev1, st));
  if (!w_inplace) convert_w(1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  return HIPNMF_OK;
}

// ---- time-shard building blocks for wide shapes (hipnmf_shard_pass / _hupdate / _residual beyond 32 channels / 8 components) on
// the general-shape kernels.  Layout contract of these shapes (include/hip_nmf.h): X row-major [T][ldx] with 16-byte aligned rows,
// W row-major [T][KP], KP = round_up(n_components, 16), columns >= n_components zero.  op: 0 pass, 1 H update, 2 residual.
template <typename real>
int hipnmf_shard_wide(hipnmf_handle* h, const hipnmf_problem* p, int op, const real* X, real* W, real* H, real* sums,
                      real* sse_col, real* xsq_col) {
  constexpr int VEC = 16 / (int)sizeof(real);
  const int B = p->batch, m = p->n_features, k = p->n_components;
  const long long T = p->n_samples;
  const int KPb = (int)round_up(k, 16), MPb = (int)round_up(m, 16);
  if (op != 1) {
    if (p->x_layout != HIPNMF_X_ROW_MAJOR || p->w_layout != HIPNMF_W_ROW_MAJOR_PAD16 || (p->ldx % VEC) != 0 ||
        (reinterpret_cast<uintptr_t>(X) % 16) != 0 || ((p->x_batch_stride * (long long)sizeof(real)) % 16) != 0 ||
        (reinterpret_cast<uintptr_t>(W) % 16) != 0)
      return fail(HIPNMF_ERR_UNSUPPORTED, "shard entry points beyond 32 channels / 8 components (and the Kullback-Leibler loss) need row-major X "
                                          "with 16-byte aligned rows (ldx %% %d == 0) and w_layout = HIPNMF_W_ROW_MAJOR_PAD16: W [n_samples][%d], "
                                          "components padded to 16 with zeros", VEC, KPb);
    if (T >= (1LL << 31)) return fail(HIPNMF_ERR_UNSUPPORTED, "n_samples must be < 2^31 per shard");
  }
  if (B > 65535) return fail(HIPNMF_ERR_UNSUPPORTED, "batch=%d: at most 65535 matrices", B);
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const long long want = std::max<long long>(1, (2LL * h->num_cu + B - 1) / B);
  long long s_try = std::min<long long>(want, (T + 63) / 64);
  if (pick_big1<real>(h, KPb, MPb, p->loss == HIPNMF_LOSS_KL)) s_try = big1_slices(h->num_cu, B, T);
  if (h->max_slices > 0) s_try = std::min<long long>(s_try, h->max_slices);
  const long long rps = round_up((T + s_try - 1) / s_try, 64);
  const int S = (int)((T + rps - 1) / rps);
  const size_t rec = (size_t)KPb * MPb + (size_t)KPb * KPb;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t o_part = carve(sizeof(real) * (size_t)B * S * rec);
  const bool kl = p->loss == HIPNMF_LOSS_KL;
  const size_t o_col = carve(sizeof(real) * (size_t)B * S * 3 * MPb);
  const size_t o_hht = carve(sizeof(real) * (size_t)B * KPb * KPb);
  const int n_hblk = (m + 63) / 64;
  const size_t o_hhtp = carve(sizeof(real) * (size_t)B * n_hblk * KPb * KPb);
  int rc = hipnmf_ensure_ws(h, off);
  if (rc) return rc;
  char* ws = static_cast<char*>(h->ws);
  BigArgs<real> ba;
  std::memset(&ba, 0, sizeof(ba));
  ba.X = X;
  ba.x_bstride = p->x_batch_stride;
  ba.ldx = p->ldx;
  ba.W = W;
  ba.w_bstride = T * (long long)KPb;
  ba.H = H;
  ba.HHt = reinterpret_cast<real*>(ws + o_hht);
  ba.part = reinterpret_cast<real*>(ws + o_part);
  ba.colpart = reinterpret_cast<real*>(ws + o_col);
  ba.T = (int)T;
  ba.m = m;
  ba.k = k;
  ba.KP = KPb;
  ba.MP = MPb;
  ba.xchunks = (m + VEC - 1) / VEC;
  ba.S = S;
  ba.rows_per_slice = (int)rps;
  ba.l1w = (real)p->l1_reg_W;
  ba.l2w = (real)p->l2_reg_W;
  ba.kl = kl ? 1 : 0;
  ba.update_h = p->update_h ? 1 : 0;
  ba.hht_part = reinterpret_cast<real*>(ws + o_hhtp);
  ba.n_hblk = n_hblk;
  const Big1Kernel<real>* b1 = pick_big1<real>(h, KPb, MPb, kl);
  const size_t fixed = sizeof(real) * ((size_t)KPb * (KPb + 4) + 12 * (size_t)MPb);
  int cbh = MPb;
  while (cbh > 16 && fixed + sizeof(real) * (size_t)KPb * (cbh + 4) > 96 * 1024) cbh = (int)round_up(cbh / 2, 16);
  ba.CBH = cbh;
  const size_t smem_w = sizeof(real) * ((size_t)KPb * (cbh + 4) + (size_t)KPb * (KPb + 4));
  const size_t smem_r = sizeof(real) * ((size_t)KPb * (cbh + 4) + 4 * (size_t)(kl ? 3 : 2) * MPb);
  const size_t smem_rec = sizeof(real) * std::max<size_t>(4 * (16 * (size_t)(KPb + 4) + 16 * (size_t)(BIG_CB + 4)) + (kl ? (size_t)KPb * (BIG_CB + 4) : 0),
                                                          4 * (size_t)KPb * BIG_CB);
  const dim3 gslice(S, B), grec(S, B, (MPb + BIG_CB - 1) / BIG_CB + 1);
  auto with_kp = [&](auto&& f) {
    switch (KPb) {
      case 16: f(std::integral_constant<int, 16>{}); break;
      case 32: f(std::integral_constant<int, 32>{}); break;
      case 48: f(std::integral_constant<int, 48>{}); break;
      default: f(std::integral_constant<int, 64>{}); break;
    }
  };
  const bool async = h->async_mode != 0;
  if (!async) HIP_TRY(hipEventRecord(h->ev0, st));
  int arc = HIPNMF_OK;
  if (op == 0) {
    if (b1) {
      const auto b1fn = kl ? b1->fn_kl : b1->fn;
      const size_t b1smem = kl ? b1->smem_kl : b1->smem;
      if (b1smem > 48 * 1024) arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(b1fn));
      if (arc) return arc;
      BigHArgs<real> hp;
      std::memset(&hp, 0, sizeof(hp));
      hp.H = H;
      hp.m = m;
      hp.k = k;
      hp.KP = KPb;
      hp.hht_part = reinterpret_cast<real*>(ws + o_hhtp);
      hp.kl = kl ? 1 : 0;
      HIPNMF_LAUNCH(big_hht_part_kernel<real>, dim3(B, n_hblk), dim3(256), sizeof(real) * 64 * (size_t)k, st, hp);
      HIPNMF_LAUNCH(b1fn, gslice, dim3(512), b1smem, st, ba);
    } else {
      HIPNMF_LAUNCH(big_hht_kernel<real>, dim3(B, KPb), dim3(256), 0, st, ba);
      if (std::max(smem_w, smem_rec) > (size_t)h->lds_per_block)
        return fail(HIPNMF_ERR_UNSUPPORTED, "n_features=%d n_components=%d needs more LDS per workgroup than this device has", m, k);
      with_kp([&](auto kp) {
        constexpr int KP = decltype(kp)::value;
        if (!arc) arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(big_pass_w_kernel<real, KP>));
        if (!arc) arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(big_records_kernel<real, KP>));
        if (arc) return;
        HIPNMF_LAUNCH((big_pass_w_kernel<real, KP>), gslice, dim3(256), smem_w, st, ba);
        if (p->update_h) HIPNMF_LAUNCH((big_records_kernel<real, KP>), grec, dim3(256), smem_rec, st, ba);
      });
      if (arc) return arc;
    }
    if (p->update_h) HIPNMF_LAUNCH(big_pack_sums_kernel<real>, dim3(B), dim3(256), 0, st, ba, sums);
  } else if (op == 1) {
    BigHArgs<real> hb;
    hb.H = H;
    hb.part = sums;
    hb.state = nullptr;
    hb.m = m;
    hb.k = k;
    hb.S = 1;
    hb.rec = k * m + k * k;
    hb.ldA = m;
    hb.offB = k * m;
    hb.ldB = k;
    hb.l1h = (real)p->l1_reg_H;
    hb.l2h = (real)p->l2_reg_H;
    hb.kl = kl ? 1 : 0;
    hb.hht_part = nullptr;
    hb.KP = KPb;
    const size_t smem_h = sizeof(real) * ((size_t)k * k + 128 * (size_t)k);
    if (smem_h > (size_t)h->lds_per_block)
      return fail(HIPNMF_ERR_UNSUPPORTED, "n_components=%d needs more LDS per workgroup than this device has", k);
    if (smem_h > 48 * 1024 && (arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(big_hupdate_kernel<real>)))) return arc;
    HIPNMF_LAUNCH(big_hupdate_kernel<real>, dim3(B, (m + 63) / 64), dim3(256), smem_h, st, hb);
  } else if (const Big1Kernel<real>* b1r = pick_big1<real>(h, KPb, MPb, false)) {
    HIPNMF_LAUNCH(b1r->resid, gslice, dim3(512), 0, st, ba);
    HIPNMF_LAUNCH(big_colsum_kernel<real>, dim3(B), dim3(256), 0, st, ba, sse_col, xsq_col);
  } else {
    with_kp([&](auto kp) {
      constexpr int KP = decltype(kp)::value;
      if (!arc && smem_r > (size_t)h->lds_per_block) arc = fail(HIPNMF_ERR_UNSUPPORTED, "n_features=%d needs more LDS per workgroup than this device has", m);
      if (!arc) arc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(big_resid_kernel<real, KP>));
      if (arc) return;
      HIPNMF_LAUNCH((big_resid_kernel<real, KP>), gslice, dim3(256), smem_r, st, ba);
    });
    if (arc) return arc;
    HIPNMF_LAUNCH(big_colsum_kernel<real>, dim3(B), dim3(256), 0, st, ba, sse_col, xsq_col);
  }
  HIP_TRY(hipGetLastError());
  if (!async) {
    HIP_TRY(hipEventRecord(h->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return HIPNMF_OK;
}
template int hipnmf_shard_wide<float>(hipnmf_handle*, const hipnmf_problem*, int, const float*, float*, float*, float*, float*, float*);
template int hipnmf_shard_wide<double>(hipnmf_handle*, const hipnmf_problem*, int, const double*, double*, double*, double*, double*,
                                       double*);

template int hipnmf_fit_wide<float>(hipnmf_handle*, const hipnmf_problem*, const float*, float*, float*, float*, int32_t*,
                                    float*, float*, const int64_t*);
template int hipnmf_fit_wide<double>(hipnmf_handle*, const hipnmf_problem*, const double*, double*, double*, double*, int32_t*,
                                     double*, double*, const int64_t*);
