// hipnmf_api.hip -- C ABI of libhip_nmf.so (see include/hip_nmf.h for the contract).
//
// Host-side driver of the CDNA4 kernels in nmf_kernels.hpp: argument validation, layout canonicalisation
// (once per fit), path selection (one persistent workgroup per matrix vs. row-sliced launches), the
// stop-rule bookkeeping of sklearn's _fit_multiplicative_update (_nmf.py:826-884) and HIP-event timing.
// There is deliberately no CPU implementation here: without a GPU every compute call fails.
#include "../../include/hip_nmf.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "hipnmf_internal.hpp"
#include "nmf_inst.hpp"
#include "nmf_rowlane_decl.hpp"
#include "nmf_small_decl.hpp"

using namespace hipnmf;

thread_local std::string g_last_error;

// ---- hipnmf_route_table: defaults + HIPNMF_ROUTES ------------------------------------------------------------------------------
namespace {
struct RouteField {
  const char* name;
  double hipnmf_route_table::*field;
};
#define RF(n) {#n, &hipnmf_route_table::n}
const RouteField kRouteFields[] = {
    RF(f32_16ch_wide_max_rows), RF(f32_16ch_k7_wide_max_rows), RF(f64_16ch_wide_max_rows), RF(f32_32ch_wide_max_rows),
    RF(f32_32ch_k5_wide_max_rows), RF(kl_f32_32ch_short_max_rows), RF(wide_min_batch_cus), RF(small_long_min_batch_cus_f32),
    RF(small_long_min_batch_cus_f64), RF(pers_s_per_row_rowmajor), RF(pers_s_per_row), RF(pers_s_fixed), RF(sliced_s_launches),
    RF(sliced_s_per_row), RF(kl_one_f32_a), RF(kl_one_f32_b), RF(kl_one_f64_a), RF(kl_one_f64_b), RF(kl_lane_f32_k5_per_ch),
    RF(kl_lane_f32_per_ch), RF(kl_lane_f64_per_ch), RF(kl_sliced_launches), RF(kl_sliced_per_slice), RF(kl_sliced_per_row),
    RF(kl_sliced_f64_factor), RF(kl_sliced_wide_factor), RF(kl_sliced_margin)};
#undef RF
hipnmf_route_table make_routes() {
  hipnmf_route_table t;
  const char* e = getenv("HIPNMF_ROUTES");
  if (!e) return t;
  std::string spec(e);
  size_t pos = 0;
  while (pos < spec.size()) {
    size_t end = spec.find(',', pos);
    if (end == std::string::npos) end = spec.size();
    const std::string item = spec.substr(pos, end - pos);
    pos = end + 1;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) continue;
    const std::string name = item.substr(0, eq);
    char* tail = nullptr;
    const double v = strtod(item.c_str() + eq + 1, &tail);
    bool known = false;
    for (const RouteField& f : kRouteFields)
      if (name == f.name && tail != item.c_str() + eq + 1 && std::isfinite(v)) {
        t.*(f.field) = v;
        known = true;
      }
    if (!known) fprintf(stderr, "libhip_nmf: HIPNMF_ROUTES: ignoring '%s'\n", item.c_str());
  }
  return t;
}
}  // namespace

const hipnmf_route_table& hipnmf_routes() {
  static const hipnmf_route_table t = make_routes();
  return t;
}

extern "C" const char* hipnmf_routes_describe(void) {
  static const std::string text = [] {
    std::string s;
    const hipnmf_route_table& t = hipnmf_routes();
    char buf[96];
    for (const RouteField& f : kRouteFields) {
      snprintf(buf, sizeof(buf), "%s%s=%.9g", s.empty() ? "" : ",", f.name, t.*(f.field));
      s += buf;
    }
    return s;
  }();
  return text.c_str();
}

// ---- hipnmf_first_use (hipnmf_internal.hpp) --------------------------------------------------------------------------------------
namespace {
constexpr size_t kSeenSlots = 1 << 14;  // ~1 500 kernel instances x devices; open addressing, never deleted from
std::atomic<uintptr_t> g_seen[kSeenSlots];
std::mutex g_first_use_mu;
inline size_t seen_slot(uintptr_t key) { return (size_t)((key * 0x9E3779B97F4A7C15ull) >> 50) & (kSeenSlots - 1); }
bool seen_lookup(uintptr_t key) {
  size_t i = seen_slot(key);
  for (size_t probe = 0; probe < kSeenSlots; ++probe, i = (i + 1) & (kSeenSlots - 1)) {
    const uintptr_t v = g_seen[i].load(std::memory_order_acquire);
    if (v == key) return true;
    if (v == 0) return false;
  }
  return false;
}
void seen_insert(uintptr_t key) {
  size_t i = seen_slot(key);
  for (size_t probe = 0; probe < kSeenSlots; ++probe, i = (i + 1) & (kSeenSlots - 1)) {
    uintptr_t v = g_seen[i].load(std::memory_order_acquire);
    if (v == key) return;
    if (v == 0 && g_seen[i].compare_exchange_strong(v, key, std::memory_order_acq_rel)) return;
    if (v == key) return;
  }  // (a full table only means every later launch of that function takes the lock)
}
const bool g_first_use_lock = [] {  // HIPNMF_FIRST_LAUNCH_LOCK=0: no lock (the A/B of tools/probes/cold_start_threads.py)
  const char* e = getenv("HIPNMF_FIRST_LAUNCH_LOCK");
  return !(e && e[0] == '0');
}();
}  // namespace

hipnmf_first_use::hipnmf_first_use(const void* fn) : key_(nullptr), locked_(false) {
  if (!g_first_use_lock) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uintptr_t key = reinterpret_cast<uintptr_t>(fn) ^ ((uintptr_t)(dev + 1) << 56);
  if (seen_lookup(key)) return;
  g_first_use_mu.lock();
  locked_ = true;
  key_ = reinterpret_cast<const void*>(key);
}
hipnmf_first_use::~hipnmf_first_use() {
  if (!locked_) return;
  seen_insert(reinterpret_cast<uintptr_t>(key_));
  g_first_use_mu.unlock();
}

int hipnmf_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

namespace {


// `row_major_ok`: the caller can hand X over (or convert it) in row-major order; then fp32 problems with 9..16
// channels and k <= 5 take the row-per-lane instance (HIPNMF_G1C16=0 keeps the channel-major (G=4, CH=4) one)
template <typename real>
const KernelSet<real>* select_kernels(int m, int k, bool row_major_ok);

template <>
const KernelSet<float>* select_kernels<float>(int m, int k, bool row_major_ok) {
  static const bool g2c8 = [] {
    const char* e = getenv("HIPNMF_G2C8");
    return e && atoi(e) != 0;
  }();
  static const bool g1c16 = [] {
    const char* e = getenv("HIPNMF_G1C16");
    return !(e && atoi(e) == 0);
  }();
  if (g1c16 && row_major_ok && m > 8 && m <= 16 && k <= 5) return kernels_f32_g1c16(k);
  // 7..8 channels (rows padded to 8): +10 % at m = 8, k = 4; at 5..6 channels the padding costs more than it buys
  if (g1c16 && row_major_ok && m > 6 && m <= 8) return kernels_f32_g1c8(k);
  if (m <= 4) return kernels_f32_g1c4(k);
  if (m <= 8) return kernels_f32_g2c4(k);
  if (m <= 16) return g2c8 ? kernels_f32_g2c8(k) : kernels_f32_g4c4(k);
  if (m <= 32) return kernels_f32_g4c8(k);
  return nullptr;
}
template <>
const KernelSet<double>* select_kernels<double>(int m, int k, bool row_major_ok) {
  static const bool g1 = [] {
    const char* e = getenv("HIPNMF_G1C16");
    return !(e && atoi(e) == 0);
  }();
  // fp64 with 7..8 channels (the reference's own recordings: 8 muscles, float64 frames) and k <= 4: row per lane
  if (g1 && row_major_ok && m > 6 && m <= 8 && k <= 4) return kernels_f64_g1c8(k);
  if (m <= 4) return kernels_f64_g1c4(k);
  if (m <= 8) return kernels_f64_g2c4(k);
  if (m <= 16) return kernels_f64_g4c4(k);
  if (m <= 32) return kernels_f64_g4c8(k);
  return nullptr;
}

}  // namespace


int hipnmf_ensure_ws(hipnmf_handle* h, size_t bytes) {
  if (bytes <= h->ws_bytes) return HIPNMF_OK;
  if (h->ws) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_bytes = 0;
  }
  HIP_TRY(hipMalloc(&h->ws, bytes));
  h->ws_bytes = bytes;
  return HIPNMF_OK;
}

int hipnmf_ensure_aux(hipnmf_handle* h, size_t bytes) {
  if (bytes <= h->aux_bytes) return HIPNMF_OK;
  if (h->aux) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(h->aux));
    h->aux = nullptr;
    h->aux_bytes = 0;
  }
  HIP_TRY(hipMalloc(&h->aux, bytes));
  h->aux_bytes = bytes;
  return HIPNMF_OK;
}

int hipnmf_copy(hipnmf_handle* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (!bytes) return HIPNMF_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HIPNMF_OK;
}

int hipnmf_allow_full_lds(hipnmf_handle* h, const void* fn) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({fn, h->device})) return HIPNMF_OK;
  hipFuncAttributes fa;
  hipnmf_first_use first_use_guard_(fn);  // (looking the function up loads its code object, like a first launch: same lock)
  HIP_TRY(hipFuncGetAttributes(&fa, fn));  // static + dynamic LDS must fit the CU: the runtime rejects more (invalid argument)
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_per_block - (int)fa.sharedSizeBytes));
  done.insert({fn, h->device});
  return HIPNMF_OK;
}

namespace {

// Cooperative grids need ALL their workgroups resident at once.  hipLaunchCooperativeKernel checks the grid against what the
// device could hold, not against what other launches hold at that moment: two cooperative grids of this process (two host
// threads, one handle each) that each got part of the chip would spin on each other until the exchange's time-out.  So every
// cooperative launch reserves its workgroups out of the device's CUs (one workgroup per CU: each takes most of the CU's LDS)
// until its stream has drained, and waits while there is no room.  Ordinary kernels of other threads only delay a
// cooperative grid (they finish by themselves); other PROCESSES on the same GPU are beyond a mutex: the time-out and its
// error code stay the last line of defence there.
class CoopBudget {
 public:
  void acquire(int device, int need, int capacity) {
    std::unique_lock<std::mutex> lock(mu_);
    need = std::min(need, capacity);
    cv_.wait(lock, [&] { return used_[slot(device)] + need <= capacity; });
    used_[slot(device)] += need;
  }
  void release(int device, int need, int capacity) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      used_[slot(device)] -= std::min(need, capacity);
    }
    cv_.notify_all();
  }

 private:
  static int slot(int device) { return device & 63; }
  std::mutex mu_;
  std::condition_variable cv_;
  int used_[64] = {0};
};
CoopBudget g_coop_budget;
struct CoopReservation {
  int device = 0, need = 0, capacity = 0;
  bool held = false;
  hipStream_t stream = nullptr;  // the stream the cooperative grid is launched on
  void take(int dev, int n, int cap, hipStream_t st) {
    g_coop_budget.acquire(dev, n, cap);
    device = dev, need = n, capacity = cap, held = true, stream = st;
  }
  void drop() {
    if (held) g_coop_budget.release(device, need, capacity);
    held = false;
  }
  // An early error return between the launch and the call's own synchronisation must not hand the CUs to the next cooperative
  // launch while this grid may still be running (round-4 advisor finding): drain the stream first, best effort.
  ~CoopReservation() {
    if (held && stream) (void)hipStreamSynchronize(stream);
    drop();
  }
};

int validate(const hipnmf_problem* p, bool shard, bool ragged = false) {
  if (!p) return fail(HIPNMF_ERR_BAD_ARG, "problem is NULL");
  if (p->struct_size != (int32_t)sizeof(hipnmf_problem))
    return fail(HIPNMF_ERR_BAD_ARG, "hipnmf_problem.struct_size = %d, library expects %d", p->struct_size,
                (int)sizeof(hipnmf_problem));
  if (p->batch < 1) return fail(HIPNMF_ERR_BAD_ARG, "batch must be >= 1 (got %d)", p->batch);
  if (p->n_samples < 1 || p->n_samples > 2000000000LL)
    return fail(HIPNMF_ERR_BAD_ARG, "n_samples must be in [1, 2e9] (got %lld)", (long long)p->n_samples);
  if (p->n_features < 1) return fail(HIPNMF_ERR_BAD_ARG, "n_features must be >= 1 (got %d)", p->n_features);
  if (p->n_components < 1) return fail(HIPNMF_ERR_BAD_ARG, "n_components must be >= 1 (got %d)", p->n_components);
  if (p->n_features > HIPNMF_MAX_FEATURES || p->n_components > HIPNMF_MAX_COMPONENTS)
    return fail(HIPNMF_ERR_UNSUPPORTED, "shape outside the compiled kernel set: n_features=%d (max %d), n_components=%d (max %d)",
                p->n_features, HIPNMF_MAX_FEATURES, p->n_components, HIPNMF_MAX_COMPONENTS);
  if (p->x_layout != HIPNMF_X_ROW_MAJOR && p->x_layout != HIPNMF_X_CHANNEL_MAJOR)
    return fail(HIPNMF_ERR_BAD_ARG, "bad x_layout %d", p->x_layout);
  if (p->w_layout != HIPNMF_W_ROW_MAJOR && p->w_layout != HIPNMF_W_COMPONENT_MAJOR && p->w_layout != HIPNMF_W_ROW_MAJOR_PAD16)
    return fail(HIPNMF_ERR_BAD_ARG, "bad w_layout %d", p->w_layout);
  if (p->w_layout == HIPNMF_W_ROW_MAJOR_PAD16 && !shard)
    return fail(HIPNMF_ERR_UNSUPPORTED, "w_layout = HIPNMF_W_ROW_MAJOR_PAD16 is a layout of the hipnmf_shard_* / hipnmf_fit_tsharded_* entry points only");
  if (p->loss != HIPNMF_LOSS_FROBENIUS && p->loss != HIPNMF_LOSS_KL)
    return fail(HIPNMF_ERR_BAD_ARG, "bad loss %d", p->loss);
  if (!ragged) {  // the ragged entry points take leading dimension and offsets per matrix from the descriptors
    const long long min_ld = (p->x_layout == HIPNMF_X_ROW_MAJOR) ? p->n_features : p->n_samples;
    if (p->ldx < min_ld) return fail(HIPNMF_ERR_BAD_ARG, "ldx=%lld smaller than %lld", (long long)p->ldx, min_ld);
    if (p->batch > 1 && p->x_batch_stride < 1) return fail(HIPNMF_ERR_BAD_ARG, "x_batch_stride must be >= 1");
  }
  if (!shard) {
    if (p->max_iter < 1) return fail(HIPNMF_ERR_BAD_ARG, "max_iter must be >= 1 (got %d)", p->max_iter);
    if (p->check_every < 1) return fail(HIPNMF_ERR_BAD_ARG, "check_every must be >= 1 (got %d)", p->check_every);
    if (!(p->tol >= 0)) return fail(HIPNMF_ERR_BAD_ARG, "tol must be >= 0");
  }
  if (!(p->l1_reg_W >= 0) || !(p->l1_reg_H >= 0) || !(p->l2_reg_W >= 0) || !(p->l2_reg_H >= 0))
    return fail(HIPNMF_ERR_BAD_ARG, "regularisation terms must be >= 0");
  return HIPNMF_OK;
}

// One matrix is addressed through a 32-bit buffer resource with 0x80000000 as the out-of-range sentinel:
// X (and W) of a single matrix must stay below 2 GiB.  Longer recordings are time-sharded (tsharded.py).
template <typename real>
int check_matrix_bytes(const hipnmf_problem* p) {
  const long long ld = std::max<long long>(p->x_layout == HIPNMF_X_CHANNEL_MAJOR ? p->ldx : 0, p->n_samples + 64);
  const long long xbytes = (long long)p->n_features * ld * (long long)sizeof(real);
  const long long wbytes = (long long)p->n_components * (p->n_samples + 64) * (long long)sizeof(real);
  if (xbytes >= (1LL << 31) || wbytes >= (1LL << 31))
    return fail(HIPNMF_ERR_UNSUPPORTED,
                "one matrix needs %lld bytes of X and %lld of W; the engine addresses < 2 GiB per matrix -- "
                "shard the time axis (hipnmf_shard_* / muscle_synergies_amd.tsharded)",
                xbytes, wbytes);
  return HIPNMF_OK;
}

template <typename real>
void launch(typename KernelSet<real>::Fn fn, dim3 grid, dim3 block, size_t smem, hipStream_t st,
            const SolveArgs<real>& a) {
  HIPNMF_LAUNCH(fn, grid, block, smem, st, a);
}

// geometry of the row-sliced path
struct SliceGeom {
  int threads, S, rows_per_slice;
};

SliceGeom slice_geometry(const hipnmf_handle* h, long long T, int B) {
  SliceGeom g;
  g.threads = (h->threads == 512 && h->slice_threads_ok512) ? 512 : 256;
  const long long quantum = g.threads;  // 64 rows per wave-step x waves
  const long long t_pad = round_up(T, 64);
  const long long target_wgs = h->max_slices > 0 ? (long long)h->max_slices : 4LL * h->num_cu;
  long long rps = round_up(std::max<long long>(2 * quantum, (t_pad * B + target_wgs - 1) / target_wgs), quantum);
  long long S = (t_pad + rps - 1) / rps;
  const long long cap = h->max_slices > 0 ? h->max_slices : 4096;
  if (S > cap) {
    rps = round_up((t_pad + cap - 1) / cap, quantum);
    S = (t_pad + rps - 1) / rps;
  }
  g.S = (int)S;
  g.rows_per_slice = (int)rps;
  return g;
}

// Narrow shapes the matrix-pipe instances of nmf_wide.hpp handle better than the lane mappings.  Measured with
// tools/quick_bench.py, 1024 x (m x 10 000), HIPNMF_FORCE_WIDE=0 / 1, M matrix-it/s (lane mapping -> matrix pipe):
//   float64, 17..32 channels: the (G=4, CH=8) instances spill from k = 7 on (0.5-0.7 KB per lane): 32 ch k = 8 0.39 -> 1.40,
//   32 ch k = 7 0.51 -> 1.40, 24 ch k = 7 0.54 -> 1.52, 17 ch k = 8 0.44 -> 1.54; k = 6: 1.24 -> 1.42 (32 ch), 1.39 -> 1.53 (24 ch);
//   k = 5: 1.53 -> 1.49 (32 ch), 1.86 -> 1.55 (24 ch): the lane mapping stays;
//   float32, 17..32 channels: k = 8 2.23 -> 2.59 (32 ch), 2.46 -> 2.81 (20 ch); k = 7 2.83 -> 2.61, k = 6 3.24 -> 2.63: stays.
// Since fit_wide4_kernel / fit_wide4d_kernel (4x4x1 / 4x4x4 tiles, at most 8 components; 17..32 channels: MP = 32, two rows per
// W^T X instruction in fp32), same measurement, lane mapping -> 4x4 tiles:
//   float32 2048 x (32 x 10 000): k = 8 2.31 -> 3.27 (16x16x4: 2.78), k = 7 3.12 -> 3.34, k = 6 3.58 -> 3.33; 24 ch k = 8 2.42 -> 3.87;
//           8192 x (32 x 2 500):  k = 8 8.60 -> 14.9, k = 6 12.1 -> 15.3, k = 5 14.5 -> 15.7; 24 ch k = 7 10.8 -> 16.3
//           => k >= 7 always, k = 5, 6 up to 5 000 rows (k <= 4: the lane mapping runs at 7.9 TB/s algorithmic already)
//   float64 1024 x (32 x 10 000): k = 8 0.40 -> 1.87 (16x16x4: 1.42), k = 6 1.30 -> 1.69, k = 5 1.68 -> 1.87, k = 4 1.85 -> 2.18;
//           4096 x (32 x 2 500):  k = 8 2.19 -> 8.22, k = 5 7.60 -> 7.78, k = 4 8.65 -> 11.2; 24 ch k = 6 6.67 -> 8.87   => every k
// Short matrices -- the reference's own sizes: a few hundred to a few thousand samples -- favour the 4x4 kernels further: the
// lane mappings walk 64-row tiles per wave and pay a heavier per-iteration epilogue.  16 384 matrices, M matrix-it/s:
//   float32 32 ch k = 4: T = 300 49 -> 129, 600 42 -> 79, 1 200 30 -> 44, 2 400 19.2 -> 18.6; 24 ch k = 3: 49 -> 139, 43 -> 86, 27 -> 49, 18.7 -> 20.0
//           (up to 16 channels the 32-channel instance pads too much: 16 ch k = 5 T = 600 49 -> 43: the lane mappings stay)
//   float64 16 ch k = 5: T = 300 39 -> 66, 600 31 -> 40, 1 200 23.0 -> 23.9, 2 400 15.4 -> 12.4; k = 7: 25 -> 65, 20 -> 40, 14.6 -> 23.5, 9.3 -> 12.3,
//           5 000: 5.2 -> 5.4; k = 8: 21 -> 65, ..., 10 000: 2.13 -> 2.67; 8 ch k = 4: 65 -> 161, 60 -> 97, 49 -> 55, 31 -> 20
// Only where one workgroup per matrix is the path anyway (many matrices, or a ragged batch) and for the Frobenius loss.
// Smallest batch for which one WAVE per matrix (fit_small_kernel with 8 / 12 / 16 tiles, one wave per SIMD) beats a workgroup
// per matrix.  tools/quick_bench.py --variant 0 / 6, M matrix-it/s (the other kernel / one wave per matrix): fp32 16 x 400, k = 5:
// B = 256 55 / 41, 512 55 / 79, 768 57 / 117, 1 024 - / 152; 8 x 700, k = 4: 256 89 / 57, 512 91 / 106; fp64 8 x 400, k = 4: 512 129 / 100, 768 120 / 147
template <typename real>
int small_long_min_batch(const hipnmf_handle* h) {
  const hipnmf_route_table& rt = hipnmf_routes();
  return (int)((sizeof(real) == 8 ? rt.small_long_min_batch_cus_f64 : rt.small_long_min_batch_cus_f32) * h->num_cu);
}

template <typename real>
bool wide_preferred(int m, int k, const hipnmf_problem* p, const hipnmf_handle* h, bool ragged) {
  if (m > HIPNMF_NARROW_MAX_FEATURES || h->variant != 0) return false;
  // float64 Kullback-Leibler beyond 8 channels: whatever the batch (the lane mappings have no multi-workgroup form for this loss,
  // and one workgroup of the 4x4x4 kernel beats one of theirs: 1 x (32 x 2 500), k = 8: 49 -> 15 ms per 200 iterations; 16 x (24 x 1 000),
  // k = 6: 0.20 -> 0.50 M matrix-it/s; 1 x (16 x 10 000), k = 5: 32.2 -> 29.6 ms; float32: 16 x (24 x 2 500), k = 6: 0.55 -> 0.41, so not there)
  if (p->loss != HIPNMF_LOSS_FROBENIUS && sizeof(real) == 8 && m > 8) return true;
  // Kullback-Leibler, few long matrices: the row-sliced one-pass kernel behind hipnmf_fit_wide (hipnmf_kl_row_sliced_wins).  The lane
  // mappings' one workgroup per matrix runs at the 4x4 kernels' rate on 17..32 channels (1 x (32 x 2 500), k = 8, fp32: 7.6 vs 7.7 ms per 200
  // iterations) and faster below (tools/probes/kl_long_narrow_ab.sh, ms per 1 000 rows and 100 iterations: fp32 up to 16 channels 0.0225 m with
  // k <= 5, 0.035 m on fit_rowlane_kernel; float64 up to 8 channels 0.05 m)
  const hipnmf_route_table& rt = hipnmf_routes();
  const double kl_one = sizeof(real) == 4 ? (m <= 16 ? (k <= 5 ? rt.kl_lane_f32_k5_per_ch : rt.kl_lane_f32_per_ch) * m : -1.0)
                                          : (m <= 8 ? rt.kl_lane_f64_per_ch * m : -1.0);
  if (p->loss != HIPNMF_LOSS_FROBENIUS && !ragged && h->max_slices != 1 &&
      hipnmf_kl_row_sliced_wins(sizeof(real) == 8, m, p->n_samples, std::max(p->batch, h->path_batch_hint), h->num_cu, kl_one, h->max_slices)) {
    static const bool kl_sliced_env = [] {
      const char* e = getenv("HIPNMF_KL_SLICED");
      return !(e && e[0] == '0');
    }();
    if (kl_sliced_env) return true;
  }
  const int fill = (int)(rt.wide_min_batch_cus * h->num_cu);
  if (!(ragged || p->batch >= fill || h->path_batch_hint >= fill)) return false;
  const long long T = p->n_samples;
  if (p->loss != HIPNMF_LOSS_FROBENIUS) {
    // Kullback-Leibler (round 5): fit_wide4_kernel<32, 2, 4, 1, 1> keeps both reconstructions on the matrix pipe, the lane mappings
    // keep them on the VALU.  tools/quick_bench.py --loss kullback-leibler, 4096 x (m x 2 500), M matrix-it/s lane mapping -> 4x4x1:
    // 32 ch k = 8: 6.64 -> 9.15; 24 ch k = 6: 8.58 -> 9.59; 32 ch k = 5: 9.55 -> 9.34; 20 ch k = 4 (T = 5 000): 6.68 -> 6.38;
    // up to 16 channels: 16 ch k = 8: 14.3 -> 14.2, k = 5 (T = 10 000): 6.3 -> 3.5
    // float64 (fit_wide4d_kernel<.., LOSS = 1>, v_mfma_f64_4x4x4): 2048 x (m x 2 500): 32 ch k = 8: 0.71 -> 4.15; 24 ch k = 6: 1.11 -> 4.50;
    // 16 ch k = 5: 4.84 -> 7.35; 4096 x (32 x 1 000), k = 4: 5.6 -> 13.6
    // 16 ch k = 5: 4.84 -> 7.35; 4096 x (32 x 1 000), k = 4: 5.6 -> 13.6; 8192 x (32 x 128), k = 8: 4.3 -> 62.5; 4096 x (12 x 1 000), k = 3: 13.5 -> 24.9;
    // up to 8 channels (one row per lane): 8 ch k = 4: 17.6 -> 10.7; 4 ch k = 2: 57 -> 27
    if (sizeof(real) == 8) return m > 8;
    // short matrices, any k (tools/probes/kl32_short_ab.sh): 8192 x (32 x 300), k = 5: 32.2 -> 68.6; k = 4: 40.9 -> 94.6; 8192 x (20 x 300), k = 3: 48.4 -> 104.7;
    // 4096 x (32 x 1 000), k = 5: 18.2 -> 21.1; 20 ch k = 3: 26.9 -> 33.7; 2 500 rows, k = 5: 9.4 -> 8.6
    return m > 16 && (k >= 6 || T <= rt.kl_f32_32ch_short_max_rows);
  }
  if (m <= 16) {  // beyond the reach of fit_small_kernel (one wave per matrix: n_samples <= 256, and up to 1 024 for some shapes
                  // when the batch gives every SIMD a wave -- then that kernel is the fastest of the three: inst_small_long.hpp)
    if (T <= 256) {
      // fit_small_kernel's own range -- where it exists: float64 with more than 6 components has no instance, and a workgroup per
      // such matrix is slow (16 x 128, k = 8: 25 -> 102 M matrix-it/s).  float64 with 9..16 channels (instances since round 3):
      // 16 384 x (16 x 200), k = 5: workgroup 40 / 4x4 kernel 78 / one wave per matrix 146; k = 3: - / 188 / 294; 16 x 250, k = 6: - / 71 / 115;
      // but 12 x 128, k = 4: 268 / 212 (half of the 256 rows are padding)
      if (sizeof(real) != 8) return false;
      return k > 6 || (m > 8 && T <= 128);
    }
    int nt = 0;
    const int small_min = small_long_min_batch<real>(h);
    if (!ragged && small_kernel_long<real>(m, k, T, &nt) && (p->batch >= small_min || h->path_batch_hint >= small_min)) return false;
    // float32 (four rows per W^T X instruction): 16 ch k = 5: T = 300 55.8 -> 62.8, 600 49.4 -> 50.7, 1 200 42 -> 35; k = 8: 46 -> 61, 41 -> 49, 33.7 -> 34.1;
    // 8 ch k = 4: 101 -> 177, 94 -> 124, 79 -> 75; 12 ch k = 3: 94 -> 179, 84 -> 118, 72 -> 71
    return sizeof(real) == 8 ? (k >= 7 || T <= rt.f64_16ch_wide_max_rows) : T <= (k >= 7 ? rt.f32_16ch_k7_wide_max_rows : rt.f32_16ch_wide_max_rows);
  }
  if (sizeof(real) == 8) return true;
  return k >= 7 || T <= (k >= 5 ? rt.f32_32ch_k5_wide_max_rows : rt.f32_32ch_wide_max_rows);
}

template <typename real>
int fit_batched_impl(hipnmf_handle* h, const hipnmf_problem* p, const real* X, real* W, real* H, real* err_out,
                     int32_t* n_iter_out, real* sse_col_out, real* xsq_col_out, const int64_t* ragged = nullptr) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  int rc = validate(p, false, ragged != nullptr);
  if (rc) return rc;
  if (!X || !W || !H) return fail(HIPNMF_ERR_BAD_ARG, "X, W and H must be non-NULL device pointers");
  HIP_TRY(hipSetDevice(h->device));
  const int B = p->batch, m = p->n_features, k = p->n_components;
  const long long T = p->n_samples;
  static const int force_wide = [] {  // HIPNMF_FORCE_WIDE=1: every shape on the matrix-pipe instances, -1: only the shapes the lane mappings cannot hold (measurement / tests)
    const char* e = getenv("HIPNMF_FORCE_WIDE");
    return e ? atoi(e) : 0;
  }();
  if (m > HIPNMF_NARROW_MAX_FEATURES || k > HIPNMF_NARROW_MAX_COMPONENTS || force_wide == 1 || (force_wide != -1 && wide_preferred<real>(m, k, p, h, ragged != nullptr)))
    return hipnmf_fit_wide<real>(h, p, X, W, H, err_out, n_iter_out, sse_col_out, xsq_col_out, ragged);
  const KernelSet<real>* ks = select_kernels<real>(m, k, true);
  if (ks && ks->row_major && (T + 64) * (long long)ks->MP * (long long)sizeof(real) >= (1LL << 31))
    ks = select_kernels<real>(m, k, false);  // rows padded to MP channels would not fit the 32-bit addressing
  if (!ks) return fail(HIPNMF_ERR_UNSUPPORTED, "no kernel for n_features=%d n_components=%d", m, k);
  rc = check_matrix_bytes<real>(p);
  if (rc) return rc;
  hipStream_t st = h->stream;

  // ---- path selection ---------------------------------------------------------------------------
  // persistent: one workgroup per matrix, ~1.2 ns per row-iteration per workgroup, num_cu in flight;
  // sliced: ~3 launches (~7 us) per iteration, rows spread over the whole chip.
  const int Bsel = h->path_batch_hint > B ? h->path_batch_hint : B;  // rank sweep on a compacted sub-batch: path and geometry of the full batch
  SliceGeom sg = slice_geometry(h, T, Bsel);
  // per-iteration cost models fitted to tools/config2_bench.py on MI355X (k = 5, m = 16, fp32):
  // persistent 2.7 ns per row of one matrix, num_cu matrices at a time; sliced 9.5 us of launches +
  // 0.021 ns per row of the whole batch; cooperative (further down) 3.7-5.5 us + 2.4 us per workgroup-step of rows
  const double waves = (double)((Bsel + h->num_cu - 1) / h->num_cu);
  const hipnmf_route_table& rt = hipnmf_routes();
  const double t_pers = waves * ((double)T * (ks->row_major ? rt.pers_s_per_row_rowmajor : rt.pers_s_per_row) + rt.pers_s_fixed);  // row-per-lane: 20.8 us / 10 000 rows
  const double t_sliced = rt.sliced_s_launches + (double)Bsel * (double)T * rt.sliced_s_per_row;
  bool persistent;
  if (h->variant == 1 || h->variant == 4 || h->variant == 5 || h->variant == 6)
    persistent = true;
  else if (h->variant == 2)
    persistent = false;
  else
    persistent = t_pers <= t_sliced;
  if (!persistent && sg.S == 1 && h->variant != 2) persistent = true;
  // grid.y carries the batch on the sliced / cooperative launches (HIP limit 65535): larger batches always have
  // more matrices than CUs and take the one-workgroup-per-matrix path (grid.x = batch)
  if (B > 65535) {
    if (h->variant == 2 || h->variant == 3)
      return fail(HIPNMF_ERR_UNSUPPORTED, "batch=%d: the row-sliced / cooperative paths take at most 65535 matrices", B);
    persistent = true;
  }
  const bool kl = p->loss == HIPNMF_LOSS_KL;
  if (kl) {
    // the KL iteration exists as the one-workgroup-per-matrix kernel only (every batch size, any T < 2 GiB)
    persistent = true;
    if (!ks->fit_persistent_kl)
      return fail(HIPNMF_ERR_UNSUPPORTED, "no Kullback-Leibler kernel in the selected instance (G=%d, CH=%d)", ks->G, ks->CH);
  }
  if (ragged) {
    persistent = true;  // one workgroup per matrix handles any mix of lengths
    if (p->x_layout != HIPNMF_X_CHANNEL_MAJOR || p->w_layout != HIPNMF_W_COMPONENT_MAJOR)
      return fail(HIPNMF_ERR_UNSUPPORTED, "ragged batches use the packed native layouts (channel-major X, component-major W)");
    if ((reinterpret_cast<uintptr_t>(X) % 16) != 0) return fail(HIPNMF_ERR_BAD_ARG, "X must be 16-byte aligned");
    for (int b = 0; b < B; ++b) {
      const int64_t* d = ragged + 4 * (size_t)b;
      if (d[0] < 1 || d[0] > p->n_samples || d[2] < d[0] || (d[2] % 4) != 0 || (d[1] % 4) != 0 || d[1] < 0 || d[3] < 0)
        return fail(HIPNMF_ERR_BAD_ARG, "bad ragged descriptor for matrix %d (T=%lld, xoff=%lld, ld=%lld, woff=%lld)", b,
                    (long long)d[0], (long long)d[1], (long long)d[2], (long long)d[3]);
      if ((long long)std::max(m, k) * (d[2] + 64) * (long long)sizeof(real) >= (1LL << 31))
        return fail(HIPNMF_ERR_UNSUPPORTED, "ragged matrix %d: ld=%lld needs >= 2 GiB of 32-bit addressing per matrix", b,
                    (long long)d[2]);
    }
  }

  // Short recordings (the reference's own matrices are 200 x 8 after time_normalize): one WAVE per matrix, everything
  // in registers, no barrier and no memory traffic inside an iteration (nmf_small.hpp).  Chosen automatically when it
  // applies (HIPNMF_SMALL=0 disables it; hipnmf_set_tuning variant 6 insists on it).
  bool use_small = false;
  SmallFn<real> small_fn = nullptr;
  int small_nt = 0;
  {
    static const bool small_env = [] {
      const char* e = getenv("HIPNMF_SMALL");
      return !(e && atoi(e) == 0);
    }();
    bool small_ok = !kl && T <= 256 && small_kernel<real>(m, k) != nullptr && !(sizeof(real) == 8 && k > 6);
    if (small_ok) {
      small_fn = small_kernel<real>(m, k);
      small_nt = SMALL_NT;
    } else if (!kl && T > 256) {
      // up to 8 / 12 / 16 tiles in registers (n_samples <= 512 / 768 / 1 024): for batches that give most SIMDs a wave (a lone
      // long-ish matrix is better served by a workgroup: small_long_min_batch), or on request (variant 6)
      int nt = 0;
      SmallFn<real> f = small_kernel_long<real>(m, k, T, &nt);
      if (f && (h->variant == 6 || Bsel >= small_long_min_batch<real>(h))) {
        small_ok = true;
        small_fn = f;
        small_nt = nt;
      }
    }
    if (h->variant == 6 && !small_ok)
      return fail(HIPNMF_ERR_UNSUPPORTED, "fit_small_kernel needs n_samples <= 256 (<= 512 / 768 / 1 024 for some shapes), the Frobenius loss and "
                  "n_features <= 16 (float64: at most 6 components; n_samples=%lld, n_features=%d, n_components=%d)", T, m, k);
    use_small = small_ok && (h->variant == 6 || (h->variant == 0 && small_env));
    if (use_small) {
      persistent = true;
      ks = select_kernels<real>(m, k, false);  // the channel-major family drives the layout handling below
    }
  }

  // cooperative multi-workgroup fit: few matrices, each long enough to keep several workgroups busy, the slice
  // of W of every workgroup resident in LDS.  S workgroups per matrix, S * B <= number of CUs (co-residency is
  // what hipLaunchCooperativeKernel guarantees; it refuses the launch otherwise and the sliced path runs).
  int coop_S = 0, coop_threads = 0;
  long long coop_rps = 0, coop_lds_rows = 0;
  size_t coop_smem = 0;
  // Measured on MI355X (tools/config2_bench.py, profiles/README.md): with release / acquire fences the barrier
  // costs ~8 us (L2 write-back / invalidate on a multi-XCD part); with the fence-free exchange the kernel uses
  // (device-scope relaxed atomics for the records and the counter) an iteration of one 16 x 10 000 matrix takes
  // 7.1 us against 10.2 us for the sliced path and 27 us for one persistent workgroup.
  if (!use_small && !ragged && !kl && ks->fit_coop && (h->variant == 3 || (h->variant == 0 && h->use_coop)) && Bsel <= h->num_cu / 2 && B <= 65535) {
    int threads = std::min(h->threads > 0 ? h->threads : 512, ks->max_threads);
    const long long t_pad = round_up(T, 64);
    while (threads > 64 && t_pad < 2LL * threads) threads /= 2;  // at least two workgroup-steps of rows in total
    long long S = std::min<long long>(h->num_cu / Bsel, (t_pad + threads - 1) / threads);
    if (h->max_slices > 0) S = std::min<long long>(S, h->max_slices);
    if (S >= 2) {
      const long long rps = round_up((T + S - 1) / S, threads);
      S = (T + rps - 1) / rps;
      const size_t base = (ks->smem_bytes(threads / 64) + 15) / 16 * 16 + sizeof(real) * (size_t)threads;
      const size_t lds_cap = h->lds_budget > 0 ? (size_t)h->lds_budget : (size_t)h->lds_per_block;
      // as many whole workgroup-steps of the slice's W as fit in LDS; the rest streams from global memory
      long long lds_rows = lds_cap > base ? (long long)((lds_cap - base) / (sizeof(real) * (size_t)k)) / threads * threads : 0;
      lds_rows = std::min(lds_rows, rps);
      const size_t smem = base + sizeof(real) * (size_t)k * (size_t)lds_rows;
      // fixed part: records travel as {value bits, generation} granules (no barrier round trip): 4.5 us, 3.7 us when the
      // one matrix's workgroups share an XCD (tools/config2_bench.py)
      const double t_fixed = B == 1 ? 3.7e-6 : 4.5e-6;
      const double t_coop = t_fixed + 2.4e-6 * (double)(rps / threads) + 0.022e-6 * (double)S;
      const bool wins = h->variant == 3 || t_coop < std::min(t_pers, t_sliced);
      if (S >= 2 && lds_rows >= threads && wins) {
        coop_S = (int)S;
        coop_threads = threads;
        coop_rps = rps;
        coop_lds_rows = lds_rows;
        coop_smem = smem;
      }
    }
  }
  if (h->variant == 3 && coop_S == 0)
    return fail(HIPNMF_ERR_UNSUPPORTED, "cooperative path not applicable (batch=%d, n_samples=%lld)", B, T);
  const bool coop = coop_S > 0;
  // float64 beyond 8 channels and the choice above is one workgroup per matrix (a batch below half the CUs that neither the
  // cooperative nor the one-wave form serves): a workgroup of the 4x4x4 kernel beats one of the (G = 4) lane mappings.
  // tools/quick_bench.py --dtype float64, ms per 200 iterations (tools/probes/f64_small_batch_ab.sh): 100 x (32 x 600), k = 8: 9.4 -> 1.8;
  // 32 x (32 x 3 000), k = 8: 27.8 -> 3.3 ([sliced]); 100 x (32 x 10 000), k = 8: 119.5 -> 17.3; 100 x (24 x 3 000), k = 6: 10.0 -> 6.6;
  // 32 x (32 x 3 000), k = 3: 5.7 -> 3.1, 100 of them: 5.9 -> 6.4; 60 x (12 x 300), k = 4: 1.0 -> 0.4; 100 x (16 x 600), k = 5: 1.6 -> 1.4.
  // Up to 16 channels only short matrices: 100 x (16 x 3 000), k = 5: 3.7 -> 4.8; 100 x (12 x 10 000), k = 4: 7.8 -> 12.6
  // ... and since the row-sliced path runs on the 4x4x4 kernel for these widths too (hipnmf_wide.hip), the cooperative and row-sliced forms
  // of the lane mappings lose as well: 17..32 channels at every k, up to 16 channels with 7 / 8 components, up to 200 000 rows
  // (tools/probes/f64_coop_ab.sh, ms per 100 iterations, lane mappings -> matrix pipe: 1 x (32 x 10 000), k = 8: 3.4 -> 1.5; 2 x (24 x 30 000), k = 6:
  // 3.5 -> 1.8; 40 x (32 x 5 000), k = 4: 3.2 -> 1.7; 1 x (32 x 100 000), k = 4: 2.2 -> 1.9; 2 x (16 x 30 000), k = 8: 3.1 -> 1.6; but 10^6 rows: 20 x 5
  // 10.6 -> 13.4; up to 16 channels with k <= 6: 1 x (16 x 3 000), k = 5: 0.9 -> 1.0, 10^6 rows: 5.4 -> 12.1, 1 x (12 x 3 000), k = 4: 0.7 -> 0.9)
  const bool f64_pipe = sizeof(real) == 8 && (m > 16 || k >= 7) && T <= 200000;
  bool to_wide = sizeof(real) == 8 && m > 8 && (f64_pipe || (T <= 1000 && persistent && !coop));
  // float32, 17..32 channels (the (G = 4, CH = 8) mappings), same finding at a smaller scale (tools/probes/f32_small_batch_ab.sh, ms per 200
  // iterations, lane mappings -> matrix pipe): 100 x (32 x 600), k = 8: 2.5 -> 1.2, k = 4: 1.2 -> 0.8; 60 x (20 x 300), k = 3: 1.0 -> 0.5;
  // 32 x (32 x 3 000), k = 8: 6.2 -> 2.3, k = 6 (24 ch): 4.3 -> 2.2; 100 x (32 x 10 000), k = 8: 18.5 -> 10.2; cooperative form, k = 8:
  // 1 x 3 000 rows 2.7 -> 2.1, 8 x 10 000 rows 3.6 -> 2.6.  Not: k <= 5 beyond 1 000 rows (100 x (32 x 3 000), k = 4: 3.0 -> 4.1) and the
  // cooperative form up to 6 components (1 x (24 x 10 000), k = 6: 2.4 -> 3.1)
  if (sizeof(real) == 4 && m > 16) to_wide = (k >= 8 && T <= 20000) || (persistent && !coop && (k >= 6 || T <= 1000));  // (k = 7, cooperative: 2.0 -> 2.1, 2.7 -> 2.7)
  if (to_wide && !use_small && !kl && !ragged && h->variant == 0 && force_wide != -1)
    return hipnmf_fit_wide<real>(h, p, X, W, H, err_out, n_iter_out, sse_col_out, xsq_col_out, ragged);

  // fp32, 9..16 channels, any k <= 8, Frobenius, one workgroup per matrix: fit_rowlane_kernel (nmf_rowlane.hpp).
  // It streams the same row-major X as the (G=1, CH=16) instance, whose KernelSet supplies the layout handling
  // and the LDS carve-up (HIPNMF_ROWLANE=0 keeps round 1's fit_persistent_kernel).
  bool use_rowlane = false;
  if constexpr (std::is_same<real, float>::value) {
    static const int rl_env = [] {  // -1: not set
      const char* e = getenv("HIPNMF_ROWLANE");
      return e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    const bool rl_ok = persistent && !coop && !use_small && m > 8 && m <= 16 && rowlane_kernel(k) != nullptr &&
                       (T + 64) * 16LL * (long long)sizeof(real) < (1LL << 31);
    // Default policy (tools/rank_sweep_bench.py, profiles/README.md): k >= 6, where round 1 had to fall back to the
    // channel-major (G=4, CH=4) mapping; at k <= 5 round 1's VALU instance is still ahead (9.54 vs 9.24 M it/s).
    // HIPNMF_ROWLANE=1 / 0 forces it on / off for every k; variant 5 / 4 of hipnmf_set_tuning does the same per handle.
    // Kullback-Leibler: the same policy.  Measured (tools/quick_bench.py --loss kullback-leibler, B = 2048, M matrix-it/s,
    // VALU instance -> matrix-pipe flavour with both W H reconstructions on the pipe): k = 3 6.59 -> 6.29, k = 5 5.24 -> 4.95,
    // k = 8 3.05 -> 3.87: the f32 pipe has the VALU's FLOP rate and the two do not overlap, so moving 3 x 16 k FMAs per row
    // across buys nothing until the VALU form runs out of registers.
    const bool want = h->variant == 5 || (h->variant != 4 && (rl_env > 0 || (rl_env < 0 && k >= 6)));
    if (h->variant == 5 && !rl_ok)
      return fail(HIPNMF_ERR_UNSUPPORTED, "fit_rowlane_kernel needs fp32, 9..16 channels and the "
                  "one-workgroup-per-matrix path (n_features=%d, loss=%d)", m, (int)p->loss);
    if (want && rl_ok) {
      ks = kernels_f32_g1c16(k);
      use_rowlane = true;
    }
  } else {
    if (h->variant == 5) return fail(HIPNMF_ERR_UNSUPPORTED, "fit_rowlane_kernel is an fp32 kernel");
  }

  // Ragged batch on a row-major instance: every distinct packed channel-major matrix (restarts of one trial share
  // theirs) is converted once into row-major rows of MP values; the kernel then gets {T, row-major offset, ldw, woff}.
  struct RaggedSrc {
    long long xoff, T, ld, roff;
  };
  std::vector<RaggedSrc> rsrc;
  std::vector<long long> rdesc;
  long long ragged_x_elems = 0;
  if (ragged && ks->row_major) {
    rdesc.resize(4 * (size_t)B);
    for (int b = 0; b < B; ++b) {
      const int64_t* d = ragged + 4 * (size_t)b;
      size_t idx = rsrc.size();
      for (size_t q = 0; q < rsrc.size(); ++q)
        if (rsrc[q].xoff == d[1] && rsrc[q].T == d[0] && rsrc[q].ld == d[2]) {
          idx = q;
          break;
        }
      if (idx == rsrc.size()) {
        rsrc.push_back({(long long)d[1], (long long)d[0], (long long)d[2], ragged_x_elems});
        ragged_x_elems += (long long)d[0] * ks->MP;
      }
      rdesc[4 * (size_t)b + 0] = d[0];
      rdesc[4 * (size_t)b + 1] = rsrc[idx].roff;
      rdesc[4 * (size_t)b + 2] = d[2];
      rdesc[4 * (size_t)b + 3] = d[3];
    }
  }

  // ---- workspace carve-up -----------------------------------------------------------------------
  const size_t o_desc_bytes = ragged ? sizeof(long long) * 4 * (size_t)B : 0;
  // canonical X: channel-major with an aligned leading dimension, or -- for the row-per-lane instance -- row-major
  // with rows of MP (= 16) values; a caller's X that already has that shape is streamed in place
  const bool aligned = (reinterpret_cast<uintptr_t>(X) % 16) == 0 && ((p->x_batch_stride * (long long)sizeof(real)) % 16) == 0;
  bool x_inplace;
  if (ks->row_major)
    x_inplace = !ragged && p->x_layout == HIPNMF_X_ROW_MAJOR && m == ks->MP && (p->ldx % 4) == 0 && aligned &&
                (long long)(T + 64) * p->ldx * (long long)sizeof(real) < (1LL << 31);
  else
    x_inplace = ragged || (p->x_layout == HIPNMF_X_CHANNEL_MAJOR && (p->ldx % ks->G) == 0 && (T % ks->G) == 0 && aligned);
  const bool w_inplace = p->w_layout == HIPNMF_W_COMPONENT_MAJOR;
  const long long ldx_c = x_inplace ? p->ldx : (ks->row_major ? (long long)ks->MP : round_up(T, 64));
  const long long ldw_c = w_inplace ? T : round_up(T, 64);
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t x_elems = ks->row_major ? (size_t)T * (size_t)ldx_c : (size_t)m * (size_t)ldx_c;  // per matrix
  const size_t o_x = x_inplace ? 0
                               : carve(sizeof(real) * (ragged ? (size_t)ragged_x_elems + 64 : (size_t)B * x_elems));
  const size_t o_w = w_inplace ? 0 : carve(sizeof(real) * (size_t)B * k * ldw_c);
  const size_t o_desc = ragged ? carve(o_desc_bytes) : 0;
  size_t o_part = 0, o_sums = 0, o_col = 0, o_state = 0;
  size_t o_cpart = 0, o_ccol = 0, o_sync = 0;
  if (coop) {
    o_cpart = carve(2 * sizeof(real) * (size_t)B * 2 * coop_S * ks->NACC);  // x 2: float records travel as 8-byte {value, generation} granules
    o_ccol = carve(sizeof(real) * (size_t)B * 2 * coop_S * 2 * ks->MP);
    o_sync = carve(sizeof(unsigned) * ((size_t)B + 2 + 10 * (size_t)B + 1024 * (size_t)B));  // + same-XCD mode: 'updates began', tickets, target, flags
  }
  size_t o_fsync = 0;
  if (!persistent) o_fsync = carve(sizeof(unsigned) * (size_t)B);
  if (!persistent || coop) {
    o_part = carve(sizeof(real) * (size_t)B * sg.S * ks->NACC);
    o_sums = carve(sizeof(real) * (size_t)B * (k * m + k * k));
    o_col = carve(sizeof(real) * (size_t)B * sg.S * 2 * ks->MP);
    o_state = carve(sizeof(real) * (size_t)B * 8);
  }
  rc = hipnmf_ensure_ws(h, std::max<size_t>(off, 256));
  if (rc) return rc;
  char* ws = static_cast<char*>(h->ws);

  SolveArgs<real> a;
  std::memset(&a, 0, sizeof(a));
  if (x_inplace) {
    a.X = X;
    a.x_bstride = p->x_batch_stride;
    a.ldx = p->ldx;
  } else {
    real* xc = reinterpret_cast<real*>(ws + o_x);
    dim3 blk(32, 8);
    if (ks->row_major && ragged) {
      for (const RaggedSrc& r : rsrc) {
        dim3 grd((unsigned)((r.T + 31) / 32), (unsigned)((ldx_c + 31) / 32), 1u);
        HIPNMF_LAUNCH(x_to_row_major_kernel<real>, grd, blk, 0, st, X + r.xoff, 0LL, r.ld, (int)HIPNMF_X_CHANNEL_MAJOR,
                           xc + r.roff, 0LL, (int)ldx_c, (int)r.T, m);
      }
    } else {
      for (int b0 = 0; b0 < B; b0 += 65535) {  // the batch rides on grid.z (HIP limit 65535)
        const unsigned nb = (unsigned)std::min(65535, B - b0);
        const real* xin = X + (long long)b0 * p->x_batch_stride;
        real* xout = xc + (size_t)b0 * x_elems;
        if (ks->row_major) {
          dim3 grd((unsigned)((T + 31) / 32), (unsigned)((ldx_c + 31) / 32), nb);
          HIPNMF_LAUNCH(x_to_row_major_kernel<real>, grd, blk, 0, st, xin, (long long)p->x_batch_stride,
                             (long long)p->ldx, (int)p->x_layout, xout, (long long)x_elems, (int)ldx_c, (int)T, m);
        } else {
          dim3 grd((unsigned)((ldx_c + 31) / 32), (unsigned)((m + 31) / 32), nb);
          HIPNMF_LAUNCH(x_to_channel_major_kernel<real>, grd, blk, 0, st, xin, (long long)p->x_batch_stride,
                             (long long)p->ldx, (int)p->x_layout, xout, (long long)x_elems, ldx_c, (int)T, m);
        }
      }
    }
    a.X = xc;
    a.x_bstride = (long long)x_elems;
    a.ldx = ldx_c;
  }
  if (w_inplace) {
    a.W = W;
    a.w_bstride = (long long)k * T;
    a.ldw = T;
  } else {
    real* wc = reinterpret_cast<real*>(ws + o_w);
    const long long n = (long long)k * ldw_c;
    for (int b0 = 0; b0 < B; b0 += 65535) {  // the batch rides on grid.y
      dim3 grd((unsigned)std::min<long long>((n + 255) / 256, 1024), (unsigned)std::min(65535, B - b0));
      HIPNMF_LAUNCH(w_convert_kernel<real>, grd, dim3(256), 0, st, W + (long long)b0 * T * k,
                         wc + (long long)b0 * k * ldw_c, ldw_c, (int)T, k, 0);
    }
    a.W = wc;
    a.w_bstride = (long long)k * ldw_c;
    a.ldw = ldw_c;
  }
  if (ragged) {  // descriptors to the device (pageable host memory: the copy is complete on return)
    const void* src = ks->row_major ? static_cast<const void*>(rdesc.data()) : static_cast<const void*>(ragged);
    HIP_TRY(hipMemcpyAsync(ws + o_desc, src, o_desc_bytes, hipMemcpyHostToDevice, st));
    a.ragged = reinterpret_cast<const long long*>(ws + o_desc);
  }
  a.H = H;
  a.err_out = err_out;
  a.n_iter_out = n_iter_out;
  a.sse_col_out = sse_col_out;
  a.xsq_col_out = xsq_col_out;
  a.T = (int)T;
  a.m = m;
  a.max_iter = p->max_iter;
  a.check_every = p->check_every;
  a.update_h = p->update_h ? 1 : 0;
  a.tol = (real)p->tol;
  a.l1w = (real)p->l1_reg_W;
  a.l2w = (real)p->l2_reg_W;
  a.l1h = (real)p->l1_reg_H;
  a.l2h = (real)p->l2_reg_H;
  a.S = 1;
  a.rows_per_slice = (int)round_up(T, 64);

  HIP_TRY(hipEventRecord(h->ev0, st));
  bool coop_done = false, coop_xcd_used = false;
  CoopReservation coop_cus;  // held until the stream has drained (released on every return path)
  if (coop) {
    SolveArgs<real> c = a;
    c.S = coop_S;
    c.rows_per_slice = (int)coop_rps;
    c.lds_rows = (int)coop_lds_rows;
    c.part = reinterpret_cast<real*>(ws + o_cpart);
    c.colpart = reinterpret_cast<real*>(ws + o_ccol);
    c.sync = reinterpret_cast<unsigned*>(ws + o_sync);
    // Same-XCD mode (one matrix: BASELINE config #2): the S workgroups are picked on ONE XCD out of a grid of 8 S and
    // exchange their records through that XCD's L2 (fit_coop_kernel, coop_barrier_xcd): 7.4 -> 6.5 us per iteration.
    // HIPNMF_COOP_XCD=0 keeps the device-scope exchange.
    static const int coop_xcd_env = [] {  // 0 off, 1 on (default), 2 test hook: one slice stays away from the head count
      const char* e = getenv("HIPNMF_COOP_XCD");
      return e ? atoi(e) : 1;
    }();
    c.coop_xcd = (coop_xcd_env > 0 && !h->coop_xcd_failed && B == 1 && coop_S >= 2 && coop_S <= 32 && 8 * coop_S <= h->num_cu) ? coop_xcd_env : 0;
    static const unsigned gen_base_env = [] {  // test hook: start the exchange's generation numbers next to the 32-bit wrap
      const char* e = getenv("HIPNMF_COOP_GEN_BASE");
      return e ? (unsigned)strtoul(e, nullptr, 0) : 0u;
    }();
    c.gen_base = gen_base_env;
    const void* kern = reinterpret_cast<const void*>(c.coop_xcd ? ks->fit_coop_xcd : ks->fit_coop);
    if (coop_smem > 48 * 1024 && (rc = hipnmf_allow_full_lds(h, kern))) return rc;
    HIP_TRY(hipMemsetAsync(c.sync, 0, sizeof(unsigned) * ((size_t)B + 2 + 10 * (size_t)B + 1024 * (size_t)B), st));
    // records travel as {value bits, generation} granules: those of a previous fit must not look fresh
    HIP_TRY(hipMemsetAsync(c.part, 0, 2 * sizeof(real) * (size_t)B * 2 * coop_S * ks->NACC, st));
    void* args[] = {&c};
    coop_cus.take(h->device, (c.coop_xcd ? 8 * coop_S : coop_S) * B, h->num_cu, st);
    // An ORDINARY launch since round 6.  hipLaunchCooperativeKernel bought nothing the kernel uses -- no grid-wide sync object (the
    // exchange is the kernel's own tagged-granule protocol with a time-out and an abort flag), and co-residency is arranged here:
    // at most one workgroup per CU (S B <= CUs, each takes most of a CU's LDS) out of a budget shared by the process's handles
    // (CoopBudget) -- and it cost a process: on this runtime (ROCm 7.x CLR / ROCr) cooperative launches issued from two host
    // threads (two streams) make the HSA runtime's shutdown, run from libamdhip64's exit handler, dereference freed memory:
    // SIGSEGV at exit() after every call had returned correct results (profiles/r06_abort_hunt.md: 2 threads 3 / 3, native
    // backtrace inside libhsa-runtime64.so <- libamdhip64.so <- __run_exit_handlers; one thread, or HIPNMF_COOP=0: never).
    // HIPNMF_COOP_LAUNCH=1 restores the cooperative launch API (A/B and the regression probe tools/probes/exit_bt.sh).
    static const bool coop_api = [] {
      const char* e = getenv("HIPNMF_COOP_LAUNCH");
      return e && atoi(e) != 0;
    }();
    const dim3 cgrid(c.coop_xcd ? 8 * coop_S : coop_S, B);
    hipError_t e;
    {
      hipnmf_first_use first_use_guard_(kern);
      e = coop_api ? hipLaunchCooperativeKernel(kern, cgrid, dim3(coop_threads), args, (unsigned)coop_smem, st)
                   : hipLaunchKernel(kern, cgrid, dim3(coop_threads), args, coop_smem, st);
    }
    coop_xcd_used = c.coop_xcd != 0;
    if (e == hipSuccess) {
      coop_done = true;
      h->last_path = 3;
      snprintf(h->last_kernel, sizeof(h->last_kernel), "fit_coop_kernel<%s,%d,%d,%d%s>",
               sizeof(real) == 4 ? "float" : "double", ks->G, ks->CH, ks->K, c.coop_xcd ? ",xcd" : "");
    } else {
      coop_cus.drop();
      (void)hipGetLastError();  // not launchable as a cooperative grid on this device: use the regular paths
      if (h->variant == 3)
        return fail(HIPNMF_ERR_HIP, "launch of the cooperative multi-workgroup kernel failed: %s", hipGetErrorString(e));
    }
  }
  if (coop_done) {
    // nothing else to enqueue
  } else if (use_small) {
    h->last_path = 1;
    if (small_nt == SMALL_NT)
      snprintf(h->last_kernel, sizeof(h->last_kernel), "fit_small_kernel<%s,%d,%d>", sizeof(real) == 4 ? "float" : "double",
               m <= 8 ? 8 : 16, k);
    else
      snprintf(h->last_kernel, sizeof(h->last_kernel), "fit_small_kernel<%s,%d,%d,%d>", sizeof(real) == 4 ? "float" : "double",
               m <= 8 ? 8 : 16, k, small_nt);
    HIPNMF_LAUNCH(small_fn, dim3(B), dim3(64), small_smem_bytes<real>(m, k), st, a);
  } else if (persistent) {
    h->last_path = 1;
    int threads = h->threads > 0 ? h->threads : 512;
    threads = std::min(threads, use_rowlane ? 512 : ks->max_threads);  // fit_rowlane_kernel: 512 for every k
    // a wave covers 64 rows per step: do not launch waves that would never get a row
    const long long t_pad = round_up(T, 64);
    while (threads > 256 && t_pad <= threads / 2) threads /= 2;
    const int nw = threads / 64;
    // W cache in LDS: as many whole workgroup-steps (threads rows each) as fit beside the reduction scratch
    const size_t base = (ks->smem_bytes(nw) + 15) / 16 * 16;
    const size_t lds_cap = h->lds_budget > 0 ? (size_t)h->lds_budget : (size_t)h->lds_per_block;
    long long lds_rows = 0;
    if (lds_cap > base + 1024 && h->use_lds_w) {
      // whole 64-row wave tiles (round 2 cached whole workgroup-steps of `threads` rows: 7680 instead of 8000 of the
      // headline's 10 000 rows; a tile is cached or not per wave, so the granule is the wave's)
      lds_rows = (long long)((lds_cap - base) / (sizeof(real) * (size_t)k)) / 64 * 64;
      lds_rows = std::min<long long>(lds_rows, t_pad);
    }
    a.lds_rows = (int)lds_rows;
    const size_t smem = base + sizeof(real) * (size_t)k * (size_t)lds_rows;
    auto kern = kl ? ks->fit_persistent_kl : ks->fit_persistent;
    snprintf(h->last_kernel, sizeof(h->last_kernel), "fit_persistent_kernel<%s,%d,%d,%d,%d>",
             sizeof(real) == 4 ? "float" : "double", ks->G, ks->CH, ks->K, kl ? 1 : 0);
    if constexpr (std::is_same<real, float>::value) {
      if (use_rowlane) {
        kern = kl ? rowlane_kernel_kl(k) : rowlane_kernel(k);
        snprintf(h->last_kernel, sizeof(h->last_kernel), "%s%s", rowlane_kernel_name(k), kl ? "[kl]" : "");
      }
    }
    if (smem > 48 * 1024 && (rc = hipnmf_allow_full_lds(h, reinterpret_cast<const void*>(kern)))) return rc;
    launch<real>(kern, dim3(B), dim3(threads), smem, st, a);
  } else {
    h->last_path = 2;
    snprintf(h->last_kernel, sizeof(h->last_kernel), "slice_pass_kernel<%s,%d,%d,%d>",
             sizeof(real) == 4 ? "float" : "double", ks->G, ks->CH, ks->K);
    a.S = sg.S;
    a.rows_per_slice = sg.rows_per_slice;
    a.part = reinterpret_cast<real*>(ws + o_part);
    a.sums = nullptr;
    a.colpart = reinterpret_cast<real*>(ws + o_col);
    const bool stop_rule = p->tol > 0;
    a.state = stop_rule ? reinterpret_cast<real*>(ws + o_state) : nullptr;
    // one launch per iteration: the last slice of a matrix to finish does the H update (HIPNMF_FUSE_H=0: two launches)
    a.fuse_h = (h->use_fuse_h && a.update_h && sg.threads >= 256) ? 1 : 0;
    a.sync = reinterpret_cast<unsigned*>(ws + o_fsync);
    if (a.fuse_h) HIP_TRY(hipMemsetAsync(a.sync, 0, sizeof(unsigned) * (size_t)B, st));
    const int nt = sg.threads, nw = nt / 64;
    const size_t smem = ks->smem_bytes(nw), smem1 = ks->smem_bytes(1);
    const dim3 grid2(sg.S, B);
    std::vector<real> host_state;
    if (stop_rule) {
      HIP_TRY(hipMemsetAsync(a.state, 0, sizeof(real) * (size_t)B * 8, st));
      a.it = 0;
      launch<real>(ks->slice_resid, grid2, dim3(nt), smem, st, a);
      launch<real>(ks->resid_finalize, dim3(B), dim3(1024), 0, st, a);
      host_state.resize((size_t)B * 8);
    }
    // `n` iterations, optionally followed by one stop-rule evaluation (sklearn: every check_every-th iteration), handed to
    // `emit` launch by launch: straight onto the stream, or into the replayed chain
    auto enqueue = [&](int n, bool check, auto&& emit) {
      for (int i = 0; i < n; ++i) {
        emit(ks->slice_pass, grid2, dim3(nt), smem, a);
        if (a.update_h && !a.fuse_h) emit(ks->hupdate, dim3(B), dim3(1024), smem1, a);  // sums the records
      }
      if (check) {
        a.it = 1;
        emit(ks->slice_resid, grid2, dim3(nt), smem, a);
        emit(ks->resid_finalize, dim3(B), dim3(1024), (size_t)0, a);
      }
    };
    auto direct = [&](typename KernelSet<real>::Fn fn, dim3 g, dim3 blk, size_t sm, const SolveArgs<real>& args) {
      launch<real>(fn, g, blk, sm, st, args);
    };
    auto all_converged = [&](bool* done) -> int {
      HIP_TRY(hipMemcpyAsync(host_state.data(), a.state, sizeof(real) * (size_t)B * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      *done = true;
      for (int b = 0; b < B; ++b) *done = *done && host_state[(size_t)b * 8 + 3] != (real)0;
      return HIPNMF_OK;
    };
    // The launches of one chunk are built once into a hipGraph and replayed: this path is launch-bound (2 small kernels
    // per iteration), and a replay costs ~1.5 us per kernel instead of ~4 us of host work.  An explicit kernel-node chain,
    // not a stream capture (hipnmf_kernel_chain in hipnmf_internal.hpp says why).
    const int chunk = stop_rule ? p->check_every : std::min(p->max_iter, 64);
    int it_done = 0;
    bool converged = false;
    if (h->use_graph && p->max_iter >= 2 * chunk) {
      hipnmf_kernel_chain chain;
      hipError_t ge = hipSuccess;
      enqueue(chunk, stop_rule, [&](typename KernelSet<real>::Fn fn, dim3 g, dim3 blk, size_t sm, const SolveArgs<real>& args) {
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
      if (ge == hipSuccess && !graph_rc) ge = hipStreamSynchronize(st);  // the chain (and its argument copies) dies with this scope
      if (graph_rc) return graph_rc;
      if (ge != hipSuccess) {
        (void)hipGetLastError();
        if (it_done > 0)  // part of the replay ran: the state on the device is no longer a whole number of chunks
          return fail(HIPNMF_ERR_HIP, "hipGraph replay failed after %d iterations: %s", it_done, hipGetErrorString(ge));
      }
    }
    while (!converged && it_done < p->max_iter) {
      const int n = std::min(chunk, p->max_iter - it_done);
      const bool check = stop_rule && n == chunk;
      enqueue(n, check, direct);
      it_done += n;
      if (check) {
        rc = all_converged(&converged);
        if (rc) return rc;
      }
    }
    a.it = -1;
    launch<real>(ks->slice_resid, grid2, dim3(nt), smem, st, a);
    launch<real>(ks->resid_finalize, dim3(B), dim3(1024), 0, st, a);
  }
  HIP_TRY(hipEventRecord(h->ev1, st));
  if (!w_inplace) {
    real* wc = reinterpret_cast<real*>(ws + o_w);
    const long long n = (long long)k * ldw_c;
    for (int b0 = 0; b0 < B; b0 += 65535) {
      dim3 grd((unsigned)std::min<long long>((n + 255) / 256, 1024), (unsigned)std::min(65535, B - b0));
      HIPNMF_LAUNCH(w_convert_kernel<real>, grd, dim3(256), 0, st, W + (long long)b0 * T * k,
                         wc + (long long)b0 * k * ldw_c, ldw_c, (int)T, k, 1);
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  coop_cus.drop();
  HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  if (coop_done) {  // a barrier that gave up (workgroups not co-resident after all) leaves the abort flag set
    unsigned flags2[2] = {0, 0};  // abort flag, 'updates began' word of the same-XCD mode
    if ((rc = hipnmf_copy(h, flags2, reinterpret_cast<unsigned*>(ws + o_sync) + B, sizeof(flags2), hipMemcpyDeviceToHost))) return rc;
    if (flags2[0] && coop_xcd_used && !flags2[1]) {
      // same-XCD mode: fewer than S workgroups turned up on the chosen XCD, the head count timed out before anything was
      // updated (W, H and the layout round trip of W are value-preserving): run again with the device-scope exchange,
      // and do not try the same-XCD mode again on this handle
      h->coop_xcd_failed = true;
      coop_cus.drop();
      return fit_batched_impl<real>(h, p, X, W, H, err_out, n_iter_out, sse_col_out, xsq_col_out, ragged);
    }
    if (flags2[0]) return fail(HIPNMF_ERR_HIP, "cooperative fit: a grid barrier timed out (results are invalid)");
  }
  return HIPNMF_OK;
}

// ---- shard building blocks ------------------------------------------------------------------------
// beyond the narrow lane mappings (32 channels / 8 components), for the Kullback-Leibler loss whatever the shape, and for any
// caller that names the general-shape kernels' W layout (HIPNMF_W_ROW_MAJOR_PAD16: e.g. the squared-error residual of a
// Kullback-Leibler fit of a narrow recording), the building blocks run on the general-shape kernels (hipnmf_wide.hip).
// Plain HIPNMF_W_ROW_MAJOR ([T][k], what hipnmf_fit_batched_* means by it) never selects them: a [T][k] buffer read with a
// stride of round_up(k, 16) would be an out-of-bounds access the library cannot detect (round-4 advisor finding).
inline bool shard_is_wide(const hipnmf_problem* p) {
  return p && p->struct_size == (int32_t)sizeof(hipnmf_problem) &&
         (p->n_features > HIPNMF_NARROW_MAX_FEATURES || p->n_components > HIPNMF_NARROW_MAX_COMPONENTS || p->loss == HIPNMF_LOSS_KL ||
          p->w_layout == HIPNMF_W_ROW_MAJOR_PAD16);
}
template <typename real>
int shard_common(hipnmf_handle* h, const hipnmf_problem* p, const KernelSet<real>** ks_out, SolveArgs<real>* a,
                 const real* X, const real* W, const real* H, SliceGeom* sg, bool need_part, bool need_col) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  int rc = validate(p, true);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  const int B = p->batch, m = p->n_features, k = p->n_components;
  const long long T = p->n_samples;
  const KernelSet<real>* ks = select_kernels<real>(m, k, false);  // the shard ABI is channel-major
  if (!ks) return fail(HIPNMF_ERR_UNSUPPORTED, "no kernel for n_features=%d n_components=%d", m, k);
  if (p->w_layout != HIPNMF_W_COMPONENT_MAJOR)
    return fail(HIPNMF_ERR_UNSUPPORTED, "shard entry points need w_layout = HIPNMF_W_COMPONENT_MAJOR");
  rc = check_matrix_bytes<real>(p);
  if (rc) return rc;
  if (X) {
    if (p->x_layout != HIPNMF_X_CHANNEL_MAJOR || (p->ldx % 4) != 0 || (T % ks->G) != 0 ||
        (reinterpret_cast<uintptr_t>(X) % 16) != 0 || ((p->x_batch_stride * (long long)sizeof(real)) % 16) != 0)
      return fail(HIPNMF_ERR_UNSUPPORTED,
                  "shard entry points need channel-major X, 16-byte aligned, ldx %% 4 == 0 and n_samples %% %d == 0",
                  ks->G);
  }
  *sg = slice_geometry(h, T, B);
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t o_part = need_part ? carve(sizeof(real) * (size_t)B * sg->S * ks->NACC) : 0;
  const size_t o_col = need_col ? carve(sizeof(real) * (size_t)B * sg->S * 2 * ks->MP) : 0;
  rc = hipnmf_ensure_ws(h, std::max<size_t>(off, 256));
  if (rc) return rc;
  char* ws = static_cast<char*>(h->ws);
  std::memset(a, 0, sizeof(*a));
  a->X = X;
  a->x_bstride = p->x_batch_stride;
  a->ldx = p->ldx;
  a->W = const_cast<real*>(W);
  a->w_bstride = (long long)k * T;
  a->ldw = T;
  a->H = const_cast<real*>(H);
  a->T = (int)T;
  a->m = m;
  a->update_h = p->update_h ? 1 : 0;
  a->l1w = (real)p->l1_reg_W;
  a->l2w = (real)p->l2_reg_W;
  a->l1h = (real)p->l1_reg_H;
  a->l2h = (real)p->l2_reg_H;
  a->S = sg->S;
  a->rows_per_slice = sg->rows_per_slice;
  a->max_iter = 1;
  a->check_every = 1;
  if (need_part) a->part = reinterpret_cast<real*>(ws + o_part);
  if (need_col) a->colpart = reinterpret_cast<real*>(ws + o_col);
  *ks_out = ks;
  return HIPNMF_OK;
}

template <typename real>
int shard_pass_impl(hipnmf_handle* h, const hipnmf_problem* p, const real* X, real* W, const real* H, real* sums) {
  if (!X || !W || !H) return fail(HIPNMF_ERR_BAD_ARG, "X, W and H must be non-NULL device pointers");
  if (p && p->update_h && !sums) return fail(HIPNMF_ERR_BAD_ARG, "sums must be non-NULL when update_h != 0");
  if (h && shard_is_wide(p)) {
    if (int vrc = validate(p, true)) return vrc;
    return hipnmf_shard_wide<real>(h, p, 0, X, W, const_cast<real*>(H), sums, nullptr, nullptr);
  }
  const KernelSet<real>* ks = nullptr;
  SolveArgs<real> a;
  SliceGeom sg;
  int rc = shard_common<real>(h, p, &ks, &a, X, W, H, &sg, true, false);
  if (rc) return rc;
  a.sums = sums;
  hipStream_t st = h->stream;
  const bool async = h->async_mode != 0;
  if (!async) HIP_TRY(hipEventRecord(h->ev0, st));
  // fp32, 9..16 channels: the row-per-lane matrix-pipe pass (four rows per lane, 1 KB per load instruction);
  // HIPNMF_SHARD_ROWLANE=0 keeps slice_pass_kernel<float, 4, 4, K>.  Same slice records either way.
  bool rl_pass = false;
  if constexpr (std::is_same<real, float>::value) {
    static const bool rl_env = [] {
      const char* e = getenv("HIPNMF_SHARD_ROWLANE");
      return !(e && atoi(e) == 0);
    }();
    rl_pass = rl_env && p->n_features > 8 && p->n_features <= 16 && ks->MP == 16 && slice_pass_rowlane(p->n_components) &&
              (p->ldx % 4) == 0 && (p->n_samples % 4) == 0 && (sg.rows_per_slice % 256) == 0 &&
              (reinterpret_cast<uintptr_t>(W) % 16) == 0;
    if (rl_pass) {
      const KernelSet<float>* k16 = kernels_f32_g1c16(p->n_components);
      HIPNMF_LAUNCH(slice_pass_rowlane(p->n_components), dim3(sg.S, p->batch), dim3(256), k16->smem_bytes(4), st, a);
    }
  }
  if (!rl_pass)
    launch<real>(ks->slice_pass, dim3(sg.S, p->batch), dim3(sg.threads), ks->smem_bytes(sg.threads / 64), st, a);
  if (a.update_h) launch<real>(ks->reduce_slices, dim3(p->batch), dim3(1024), 0, st, a);
  HIP_TRY(hipGetLastError());
  if (!async) {
    HIP_TRY(hipEventRecord(h->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return HIPNMF_OK;
}

template <typename real>
int shard_hupdate_impl(hipnmf_handle* h, const hipnmf_problem* p, real* H, const real* sums) {
  if (!H || !sums) return fail(HIPNMF_ERR_BAD_ARG, "H and sums must be non-NULL device pointers");
  if (h && shard_is_wide(p)) {
    if (int vrc = validate(p, true)) return vrc;
    return hipnmf_shard_wide<real>(h, p, 1, nullptr, nullptr, H, const_cast<real*>(sums), nullptr, nullptr);
  }
  const KernelSet<real>* ks = nullptr;
  SolveArgs<real> a;
  SliceGeom sg;
  int rc = shard_common<real>(h, p, &ks, &a, nullptr, nullptr, H, &sg, false, false);
  if (rc) return rc;
  a.sums = const_cast<real*>(sums);
  hipStream_t st = h->stream;
  const bool async = h->async_mode != 0;
  if (!async) HIP_TRY(hipEventRecord(h->ev0, st));
  launch<real>(ks->hupdate, dim3(p->batch), dim3(1024), ks->smem_bytes(1), st, a);
  HIP_TRY(hipGetLastError());
  if (!async) {
    HIP_TRY(hipEventRecord(h->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return HIPNMF_OK;
}

template <typename real>
int shard_residual_impl(hipnmf_handle* h, const hipnmf_problem* p, const real* X, const real* W, const real* H,
                        real* sse_col, real* xsq_col) {
  if (!X || !W || !H || !sse_col) return fail(HIPNMF_ERR_BAD_ARG, "X, W, H and sse_col must be non-NULL");
  if (h && shard_is_wide(p)) {
    if (int vrc = validate(p, true)) return vrc;
    return hipnmf_shard_wide<real>(h, p, 2, X, const_cast<real*>(W), const_cast<real*>(H), nullptr, sse_col, xsq_col);
  }
  const KernelSet<real>* ks = nullptr;
  SolveArgs<real> a;
  SliceGeom sg;
  int rc = shard_common<real>(h, p, &ks, &a, X, W, H, &sg, false, true);
  if (rc) return rc;
  a.sse_col_out = sse_col;
  a.xsq_col_out = xsq_col;
  a.it = -1;
  hipStream_t st = h->stream;
  const bool async = h->async_mode != 0;
  if (!async) HIP_TRY(hipEventRecord(h->ev0, st));
  launch<real>(ks->slice_resid, dim3(sg.S, p->batch), dim3(sg.threads), ks->smem_bytes(sg.threads / 64), st, a);
  launch<real>(ks->resid_finalize, dim3(p->batch), dim3(1024), 0, st, a);
  HIP_TRY(hipGetLastError());
  if (!async) {
    HIP_TRY(hipEventRecord(h->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return HIPNMF_OK;
}

// The whole time-sharded fit on this rank's rows (the loop muscle_synergies_amd/tsharded.py::fit_tsharded drives from
// Python, for hosts that are not Python): sklearn's iteration order and stop rule (_nmf.py:826-884) with the two sums
// that span all rows -- [W^T X | W^T W] once per iteration, the per-column residual at every stop-rule check -- handed
// to the caller's all-reduce.  Everything is enqueued on the handle's stream; the host waits only where the stop rule
// needs the residual.
template <typename real>
int tsharded_impl(hipnmf_handle* h, const hipnmf_problem* p, const real* X, real* W, real* H, hipnmf_allreduce_fn allreduce,
                  void* user, real* err_out, int32_t* n_iter_out, real* sse_col_out, real* xsq_col_out) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (!p) return fail(HIPNMF_ERR_BAD_ARG, "problem is NULL");
  if (!X || !W || !H) return fail(HIPNMF_ERR_BAD_ARG, "X, W and H must be non-NULL device pointers");
  if (p->struct_size != (int32_t)sizeof(hipnmf_problem))
    return fail(HIPNMF_ERR_BAD_ARG, "hipnmf_problem.struct_size = %d, library expects %d", p->struct_size, (int)sizeof(hipnmf_problem));
  if (p->max_iter < 1) return fail(HIPNMF_ERR_BAD_ARG, "max_iter must be >= 1 (got %d)", p->max_iter);
  if (p->batch < 1 || p->n_features < 1 || p->n_components < 1) return fail(HIPNMF_ERR_BAD_ARG, "bad shape");
  HIP_TRY(hipSetDevice(h->device));
  const int B = p->batch, m = p->n_features, k = p->n_components;
  const size_t n_sums = (size_t)B * ((size_t)k * m + (size_t)k * k), n_res = (size_t)B * 2 * m;
  const int check_every = p->check_every > 0 ? p->check_every : 10;
  const bool stop_rule = p->tol > 0;
  hipStream_t st = h->stream;

  // [sums | sse, xsq packed per matrix as the all-reduce wants them: one buffer, one call] in the handle's second scratch
  if (int arc = hipnmf_ensure_aux(h, sizeof(real) * (n_sums + 2 * n_res))) return arc;
  real* dbuf = static_cast<real*>(h->aux);
  real* sums = dbuf;
  real* packed = dbuf + n_sums;        // [B][2 m]: sse | xsq
  real* cols = dbuf + n_sums + n_res;  // scratch the residual kernels write: sse [B][m], xsq [B][m]
  std::vector<real> host(n_res);
  std::vector<double> err(B, 0.0), err_init(B, 0.0), prev(B, 0.0);
  const int saved_async = h->async_mode;
  h->async_mode = 1;  // the building blocks only enqueue; this function decides where the host waits
  int rc = HIPNMF_OK, n_iter = 0;
  auto reduce = [&](real* buf, size_t count) -> int {
    if (!allreduce) return HIPNMF_OK;
    const int e = allreduce(buf, count, (int)sizeof(real), (void*)st, user);
    return e == 0 ? HIPNMF_OK : fail(HIPNMF_ERR_HIP, "the all-reduce callback returned %d", e);
  };
  // global ||X - W H||_F per matrix (and the per-column sums for VAF), in the arithmetic type like the other paths
  // (Kullback-Leibler: the residual returns the divergence per column, err = sqrt(2 KL) (_nmf.py:185-189); `pp` with the loss
  // switched to Frobenius gives the squared-error columns of VAF once at the end)
  const bool kl = p->loss == HIPNMF_LOSS_KL;
  auto global_error = [&](const hipnmf_problem* pp = nullptr) -> int {
    if (!pp) pp = p;
    int r = shard_residual_impl<real>(h, pp, X, W, H, cols, cols + (size_t)B * m);
    if (r) return r;
    for (int b = 0; b < B && !r; ++b) {  // pack [sse_b | xsq_b]
      if (hipMemcpyAsync(packed + (size_t)b * 2 * m, cols + (size_t)b * m, sizeof(real) * m, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync(packed + (size_t)b * 2 * m + m, cols + (size_t)B * m + (size_t)b * m, sizeof(real) * m, hipMemcpyDeviceToDevice, st) != hipSuccess)
        r = fail(HIPNMF_ERR_HIP, "hipMemcpyAsync failed");
    }
    if (r) return r;
    r = reduce(packed, n_res);
    if (r) return r;
    if (hipMemcpyAsync(host.data(), packed, sizeof(real) * n_res, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return fail(HIPNMF_ERR_HIP, "reading the residual back failed: %s", hipGetErrorString(hipGetLastError()));
    if (pp != p) return HIPNMF_OK;  // (the extra pass for VAF: the error stays the divergence's)
    for (int b = 0; b < B; ++b) {
      real tot = (real)0;
      for (int j = 0; j < m; ++j) tot += host[(size_t)b * 2 * m + j];
      err[b] = kl ? (double)(real)std::sqrt(2.0 * std::max((double)tot, 0.0)) : (double)(real)std::sqrt((double)tot);
    }
    return HIPNMF_OK;
  };
  // reconstruction_err_ / n_iter_ per matrix and the global per-column sums (for VAF) to the caller's device buffers
  auto write_outputs = [&]() -> int {
    std::vector<real> e(B);
    std::vector<int32_t> it(B, n_iter);
    for (int b = 0; b < B; ++b) e[b] = (real)err[b];
    if (err_out) HIP_TRY(hipMemcpyAsync(err_out, e.data(), sizeof(real) * B, hipMemcpyHostToDevice, st));
    if (n_iter_out) HIP_TRY(hipMemcpyAsync(n_iter_out, it.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, st));
    for (int b = 0; b < B; ++b) {
      if (sse_col_out)
        HIP_TRY(hipMemcpyAsync(sse_col_out + (size_t)b * m, packed + (size_t)b * 2 * m, sizeof(real) * m, hipMemcpyDeviceToDevice, st));
      if (xsq_col_out)
        HIP_TRY(hipMemcpyAsync(xsq_col_out + (size_t)b * m, packed + (size_t)b * 2 * m + m, sizeof(real) * m, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));  // e / it are host temporaries
    return HIPNMF_OK;
  };
  do {
    if (stop_rule) {
      if ((rc = global_error())) break;
      err_init = err;
      prev = err;
    }
    for (n_iter = 1; n_iter <= p->max_iter; ++n_iter) {
      if ((rc = shard_pass_impl<real>(h, p, X, W, H, sums))) break;
      if (p->update_h) {
        if ((rc = reduce(sums, n_sums))) break;
        if ((rc = shard_hupdate_impl<real>(h, p, H, sums))) break;
      }
      if (stop_rule && n_iter % check_every == 0) {
        if ((rc = global_error())) break;
        bool all_done = true;  // every matrix of the (small) batch must have converged; B == 1: sklearn's rule
        for (int b = 0; b < B; ++b) all_done = all_done && ((real)((prev[b] - err[b]) / err_init[b]) < (real)p->tol);
        if (all_done) break;
        prev = err;
      }
    }
    if (rc) break;
    if (n_iter > p->max_iter) n_iter = p->max_iter;
    if ((rc = global_error())) break;
    if (kl && (sse_col_out || xsq_col_out)) {
      hipnmf_problem q = *p;
      q.loss = HIPNMF_LOSS_FROBENIUS;
      if ((rc = global_error(&q))) break;
    }
    rc = write_outputs();
  } while (false);
  h->async_mode = saved_async;
  (void)hipStreamSynchronize(st);
  return rc;
}

// Rank sweep (BASELINE.json config #4; find_synergies(df, k_min, k_max) for a batch of trials, analysis.py:848-914):
// one batched fit per rank from init='random' starting points drawn on the device, VAF per trial and rank
// (analysis.py:654-662), and the smallest rank whose VAF reaches the threshold.  Host-side loop over the library's own
// entry points; the per-rank results are read back (B (m + 2) numbers per rank) for the VAF table.
//
// stop_at_threshold: the Rabbi et al. protocol proper -- a trial whose VAF has reached the threshold is not fitted at
// the higher ranks.  After every rank the still-unexplained trials are compacted (their X gathered into workspace, their
// starting points drawn with the ORIGINAL trial indices as keys, the solver path chosen as for the full batch), so
// every fit that does run is bit-identical to the one the compute-all mode runs and `selected` cannot differ.
template <typename real>
__global__ void gather_matrices_kernel(const real* __restrict__ src, long long src_bstride, const int* __restrict__ index,
                                       real* __restrict__ dst, long long dst_bstride, long long span) {
  const real* s = src + (long long)index[blockIdx.y] * src_bstride;
  real* d = dst + (long long)blockIdx.y * dst_bstride;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += (long long)gridDim.x * blockDim.x) d[i] = s[i];
}

template <typename real>
int rank_sweep_impl(hipnmf_handle* h, const hipnmf_problem* p, int k_min, int k_max, double vaf_threshold, uint64_t seed,
                    int first_matrix, const real* X, real* W_ws, real* H_out, real* vaf_out, int32_t* selected_out,
                    real* err_out, int32_t* n_iter_out, bool stop_at_threshold) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (!p || p->struct_size != (int32_t)sizeof(hipnmf_problem)) return fail(HIPNMF_ERR_BAD_ARG, "bad hipnmf_problem");
  if (!X || !W_ws || !H_out || !vaf_out) return fail(HIPNMF_ERR_BAD_ARG, "X, W_ws, H_out and vaf_out must be non-NULL");
  if (k_min < 1 || k_max < k_min || k_max > p->n_features)
    return fail(HIPNMF_ERR_BAD_ARG, "invalid number of components: need 1 <= k_min <= k_max <= n_features (got %d..%d, %d features)",
                k_min, k_max, p->n_features);
  if (p->n_features > HIPNMF_MAX_FEATURES || k_max > HIPNMF_MAX_COMPONENTS)
    return fail(HIPNMF_ERR_UNSUPPORTED, "rank %d with %d features is outside the compiled kernel set (max %d x %d)", k_max,
                p->n_features, HIPNMF_MAX_COMPONENTS, HIPNMF_MAX_FEATURES);
  if (p->batch < 1) return fail(HIPNMF_ERR_BAD_ARG, "batch must be >= 1");
  HIP_TRY(hipSetDevice(h->device));
  const int B = p->batch, m = p->n_features, nk = k_max - k_min + 1;
  const long long T = p->n_samples;
  // one matrix of the caller's X as a flat span (gathered as it lies: same layout and leading dimension; the last row ends after
  // its own elements, not after a whole leading dimension -- a strided view's buffer may end there)
  const long long rows = p->x_layout == HIPNMF_X_ROW_MAJOR ? T : (long long)m, row_len = p->x_layout == HIPNMF_X_ROW_MAJOR ? (long long)m : T;
  const long long span = (rows - 1) * p->ldx + row_len;
  const long long cstride = round_up(span, 16 / (long long)sizeof(real));
  // scratch (the handle's second buffer; `ws` belongs to the fits): [2][B][m] sse | xsq, [B] err, [B] n_iter, the index list and
  // the compacted H; the compacted X is added when the first compaction happens, sized for the trials still active then
  size_t aoff = 0;
  auto acarve = [&](size_t bytes) {
    size_t o = aoff;
    aoff += (bytes + 255) / 256 * 256;
    return o;
  };
  const size_t o_cols = acarve(sizeof(real) * ((size_t)2 * B * m + B)), o_it = acarve(sizeof(int32_t) * (size_t)B);
  const size_t o_index = acarve(sizeof(int) * (size_t)B), o_hc = acarve(sizeof(real) * (size_t)B * k_max * m);
  const size_t o_xc = aoff;
  size_t xc_matrices = 0;
  if (int arc = hipnmf_ensure_aux(h, aoff)) return arc;
  real *cols = nullptr, *d_err = nullptr, *xc = nullptr, *hc = nullptr;
  int32_t* d_it = nullptr;
  int* d_index = nullptr;
  auto place = [&]() {  // (the buffer may have moved when it grew: nothing of a previous rank is live across that)
    char* base = static_cast<char*>(h->aux);
    cols = reinterpret_cast<real*>(base + o_cols);
    d_err = cols + (size_t)2 * B * m;
    d_it = reinterpret_cast<int32_t*>(base + o_it);
    d_index = reinterpret_cast<int*>(base + o_index);
    hc = reinterpret_cast<real*>(base + o_hc);
    xc = reinterpret_cast<real*>(base + o_xc);
  };
  place();
  const real nan = std::numeric_limits<real>::quiet_NaN();
  std::vector<real> hs((size_t)2 * B * m), vaf((size_t)B * nk, nan), herr((size_t)B * nk, nan), e(B);
  std::vector<int32_t> hit((size_t)B * nk, 0), it(B), sel(B, -1);
  std::vector<int> active(B);
  for (int b = 0; b < B; ++b) active[b] = b;
  int rc = HIPNMF_OK;
  size_t h_off = 0;
  const int saved_hint = h->path_batch_hint;
  h->path_batch_hint = std::max(B, saved_hint);
  for (int k = k_min; k <= k_max && !rc; ++k) {
    real* Hk = H_out + h_off;
    h_off += (size_t)B * k * m;
    const int nA = (int)active.size();
    const bool compact = stop_at_threshold && nA < B;
    if (compact && hipMemsetAsync(Hk, 0, sizeof(real) * (size_t)B * k * m, h->stream) != hipSuccess) {
      rc = fail(HIPNMF_ERR_HIP, "hipMemsetAsync failed");
      break;
    }
    if (nA == 0) continue;  // every trial is explained: the remaining ranks report NaN / 0 iterations / zero components
    hipnmf_problem q = *p;
    q.n_components = k;
    q.batch = nA;
    const real* Xk = X;
    real* Hfit = Hk;
    if (compact) {
      if (!xc_matrices) {
        xc_matrices = (size_t)nA;  // the active set only shrinks
        if ((rc = hipnmf_ensure_aux(h, o_xc + sizeof(real) * xc_matrices * (size_t)cstride))) break;
        place();
      }
      if (hipMemcpyAsync(d_index, active.data(), sizeof(int) * (size_t)nA, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
        rc = fail(HIPNMF_ERR_HIP, "copying the trial list failed");
        break;
      }
      for (int b0 = 0; b0 < nA; b0 += 65535) {
        dim3 grd((unsigned)std::min<long long>((span + 255) / 256, 256), (unsigned)std::min(65535, nA - b0));
        HIPNMF_LAUNCH(gather_matrices_kernel<real>, grd, dim3(256), 0, h->stream, X, (long long)p->x_batch_stride, d_index + b0,
                           xc + (size_t)b0 * cstride, cstride, span);
      }
      q.x_batch_stride = cstride;
      Xk = xc;
      Hfit = hc;
    }
    rc = hipnmf_random_init_indexed<real>(h, &q, seed + (uint64_t)k, first_matrix, compact ? d_index : nullptr, Xk, W_ws, Hfit);
    if (rc) break;
    rc = fit_batched_impl<real>(h, &q, Xk, W_ws, Hfit, d_err, d_it, cols, cols + (size_t)nA * m);
    if (rc) break;
    if (hipMemcpyAsync(hs.data(), cols, sizeof(real) * (size_t)2 * nA * m, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipMemcpyAsync(e.data(), d_err, sizeof(real) * nA, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipMemcpyAsync(it.data(), d_it, sizeof(int32_t) * nA, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
      rc = fail(HIPNMF_ERR_HIP, "reading the results of rank %d back failed", k);
      break;
    }
    const int ki = k - k_min;
    std::vector<int> still;
    for (int a = 0; a < nA; ++a) {
      const int b = active[a];
      if (compact &&  // components of the trials fitted at this rank to their own slots (the others stay zero)
          hipMemcpyAsync(Hk + (size_t)b * k * m, hc + (size_t)a * k * m, sizeof(real) * (size_t)k * m, hipMemcpyDeviceToDevice,
                         h->stream) != hipSuccess) {
        rc = fail(HIPNMF_ERR_HIP, "hipMemcpyAsync failed");
        break;
      }
      real sse = (real)0, xsq = (real)0;
      for (int j = 0; j < m; ++j) {
        sse += hs[(size_t)a * m + j];
        xsq += hs[(size_t)nA * m + (size_t)a * m + j];
      }
      const real v = (real)1 - sse / xsq;  // VAF over all muscles (analysis.py:660-662)
      vaf[(size_t)b * nk + ki] = v;
      herr[(size_t)b * nk + ki] = e[a];
      hit[(size_t)b * nk + ki] = it[a];
      if (sel[b] < 0 && (double)v >= vaf_threshold) sel[b] = k;
      if (!stop_at_threshold || sel[b] < 0) still.push_back(b);
    }
    if (stop_at_threshold) active.swap(still);
  }
  h->path_batch_hint = saved_hint;
  if (!rc) {
    hipStream_t st = h->stream;
    if (hipMemcpyAsync(vaf_out, vaf.data(), sizeof(real) * vaf.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
        (selected_out && hipMemcpyAsync(selected_out, sel.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, st) != hipSuccess) ||
        (err_out && hipMemcpyAsync(err_out, herr.data(), sizeof(real) * herr.size(), hipMemcpyHostToDevice, st) != hipSuccess) ||
        (n_iter_out && hipMemcpyAsync(n_iter_out, hit.data(), sizeof(int32_t) * hit.size(), hipMemcpyHostToDevice, st) != hipSuccess))
      rc = fail(HIPNMF_ERR_HIP, "writing the sweep results failed");
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess && !rc) rc = fail(HIPNMF_ERR_HIP, "hipStreamSynchronize failed");  // (host vectors above are read until here)
  return rc;
}

}  // namespace

// =================================================================================================
extern "C" {

int hipnmf_rank_sweep_f32(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                          uint64_t seed, int32_t first_matrix, const float* X, float* W_ws, float* H_out, float* vaf_out,
                          int32_t* selected_out, float* err_out, int32_t* n_iter_out) {
  return rank_sweep_impl<float>(h, p, k_min, k_max, vaf_threshold, seed, first_matrix, X, W_ws, H_out, vaf_out, selected_out,
                                err_out, n_iter_out, false);
}
int hipnmf_rank_sweep_stop_f32(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                               uint64_t seed, int32_t first_matrix, const float* X, float* W_ws, float* H_out, float* vaf_out,
                               int32_t* selected_out, float* err_out, int32_t* n_iter_out) {
  return rank_sweep_impl<float>(h, p, k_min, k_max, vaf_threshold, seed, first_matrix, X, W_ws, H_out, vaf_out, selected_out,
                                err_out, n_iter_out, true);
}
int hipnmf_rank_sweep_f64(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                          uint64_t seed, int32_t first_matrix, const double* X, double* W_ws, double* H_out, double* vaf_out,
                          int32_t* selected_out, double* err_out, int32_t* n_iter_out) {
  return rank_sweep_impl<double>(h, p, k_min, k_max, vaf_threshold, seed, first_matrix, X, W_ws, H_out, vaf_out, selected_out,
                                 err_out, n_iter_out, false);
}
int hipnmf_rank_sweep_stop_f64(hipnmf_handle* h, const hipnmf_problem* p, int32_t k_min, int32_t k_max, double vaf_threshold,
                               uint64_t seed, int32_t first_matrix, const double* X, double* W_ws, double* H_out, double* vaf_out,
                               int32_t* selected_out, double* err_out, int32_t* n_iter_out) {
  return rank_sweep_impl<double>(h, p, k_min, k_max, vaf_threshold, seed, first_matrix, X, W_ws, H_out, vaf_out, selected_out,
                                 err_out, n_iter_out, true);
}

int hipnmf_fit_tsharded_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, float* W, float* H,
                            hipnmf_allreduce_fn allreduce, void* user, float* err_out, int32_t* n_iter_out,
                            float* sse_col_out, float* xsq_col_out) {
  return tsharded_impl<float>(h, p, X, W, H, allreduce, user, err_out, n_iter_out, sse_col_out, xsq_col_out);
}
int hipnmf_fit_tsharded_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* W, double* H,
                            hipnmf_allreduce_fn allreduce, void* user, double* err_out, int32_t* n_iter_out,
                            double* sse_col_out, double* xsq_col_out) {
  return tsharded_impl<double>(h, p, X, W, H, allreduce, user, err_out, n_iter_out, sse_col_out, xsq_col_out);
}

int hipnmf_version(void) { return HIPNMF_VERSION; }

const char* hipnmf_last_error(void) { return g_last_error.c_str(); }

int hipnmf_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(HIPNMF_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  return n;
}

// Set by an exit handler registered at the first hipnmf_create -- i.e. AFTER the HIP runtime registered its own, so (exit handlers
// run last-in-first-out) it fires BEFORE the runtime starts to unload.  A handle destroyed after that point (a host that frees
// its objects from static destructors or from a garbage collector running at interpreter teardown) is released on the host side only:
// calling hipStreamSynchronize / hipFree into a runtime that is being torn down is how a process aborts at exit.
static std::atomic<bool> g_process_exiting{false};
static void hipnmf_mark_exiting() { g_process_exiting.store(true, std::memory_order_release); }

int hipnmf_create(int device, hipnmf_handle** out) {
  if (!out) return fail(HIPNMF_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n < 1)
    return fail(HIPNMF_ERR_NO_DEVICE, "no ROCm device available (%s); libhip_nmf has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= n) return fail(HIPNMF_ERR_BAD_ARG, "device %d out of range [0, %d)", device, n);
  HIP_TRY(hipSetDevice(device));
  static std::once_flag exit_hook_once;
  std::call_once(exit_hook_once, [] { std::atexit(hipnmf_mark_exiting); });
  hipnmf_handle* h = new hipnmf_handle();
  h->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
    h->num_cu = prop.multiProcessorCount;
    if (prop.maxSharedMemoryPerMultiProcessor >= 65536) h->lds_per_block = (int)prop.maxSharedMemoryPerMultiProcessor;
  }
  if (const char* e = getenv("HIPNMF_LDS_W")) h->use_lds_w = atoi(e) != 0;
  if (const char* e = getenv("HIPNMF_LDS_BUDGET")) h->lds_budget = atoi(e);
  if (const char* e = getenv("HIPNMF_GRAPH")) h->use_graph = atoi(e) != 0;
  if (const char* e = getenv("HIPNMF_COOP")) h->use_coop = atoi(e) != 0;
  if (const char* e = getenv("HIPNMF_FUSE_H")) h->use_fuse_h = atoi(e) != 0;
  if (const char* e = getenv("HIPNMF_SLICE512")) h->slice_threads_ok512 = atoi(e) != 0;
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    delete h;
    return fail(HIPNMF_ERR_HIP, "stream/event creation failed");
  }
  h->stream = h->own_stream;
  *out = h;
  return HIPNMF_OK;
}

int hipnmf_destroy(hipnmf_handle* h) {
  if (!h) return HIPNMF_OK;
  if (g_process_exiting.load(std::memory_order_acquire)) {  // the device memory goes with the process
    delete h;
    return HIPNMF_OK;
  }
  const hipError_t se = hipSetDevice(h->device);
  if (se == hipErrorDeinitialized || se == hipErrorContextIsDestroyed || se == hipErrorNotInitialized || se == hipErrorNoDevice ||
      se == hipErrorInvalidContext) {
    (void)hipGetLastError();
    delete h;  // the runtime is gone (or was never up in this process, e.g. after a fork): nothing of ours is left on the device
    return HIPNMF_OK;
  }
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  if (h->ws) (void)hipFree(h->ws);
  if (h->aux) (void)hipFree(h->aux);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  (void)hipGetLastError();
  delete h;
  return HIPNMF_OK;
}

int hipnmf_set_batch_hint(hipnmf_handle* h, int batch) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (batch < 0) return fail(HIPNMF_ERR_BAD_ARG, "batch hint must be >= 0 (got %d)", batch);
  h->path_batch_hint = batch;
  return HIPNMF_OK;
}

int hipnmf_set_stream(hipnmf_handle* h, void* hip_stream) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (hip_stream == HIPNMF_STREAM_NULL)
    h->stream = nullptr;  // legacy default stream
  else
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return HIPNMF_OK;
}

size_t hipnmf_workspace_bytes(const hipnmf_problem* p, int elem_size) {
  if (!p || (elem_size != 4 && elem_size != 8)) return 0;
  const long long ld = round_up(p->n_samples, 64);
  const int m_pad = p->n_features <= 8 ? 8 : (p->n_features <= 16 ? 16 : p->n_features);  // row-major copies pad the rows
  size_t bytes = (size_t)elem_size * (size_t)p->batch * (size_t)(m_pad + p->n_components) * (size_t)ld;
  bytes += (size_t)elem_size * (size_t)p->batch * 4096 * 2;  // slice partials upper bound
  if (p->n_features > HIPNMF_NARROW_MAX_FEATURES || p->n_components > HIPNMF_NARROW_MAX_COMPONENTS) {
    // wide / general shapes: row-major copies with the components padded to 16, and when the fit is row-sliced (at most
    // max(batch, 2 x 256 CUs) slices in all) one [W^T X | W^T W] record and three column sums per slice, H H^T per matrix
    const size_t KP = (size_t)round_up(p->n_components, 16), MP = (size_t)round_up(p->n_features, 16);
    const size_t slices = std::max<size_t>((size_t)p->batch, 512);
    bytes = (size_t)elem_size * ((size_t)p->batch * (MP + KP) * (size_t)ld + slices * (KP * MP + KP * KP + 3 * MP) + (size_t)p->batch * KP * KP + 4096);
  }
  return bytes;
}

const char* hipnmf_last_kernel(hipnmf_handle* h) { return h ? h->last_kernel : ""; }

int hipnmf_last_kernel_ms(hipnmf_handle* h, float* ms) {
  if (!h || !ms) return fail(HIPNMF_ERR_BAD_ARG, "NULL argument");
  *ms = h->last_ms;
  return HIPNMF_OK;
}

int hipnmf_set_async(hipnmf_handle* h, int enable) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  h->async_mode = enable ? 1 : 0;
  return HIPNMF_OK;
}

int hipnmf_set_tuning(hipnmf_handle* h, int threads, int max_slices, int variant) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (threads != 0 && threads != 256 && threads != 512 && threads != 768 && threads != 1024)
    return fail(HIPNMF_ERR_BAD_ARG, "threads must be 0, 256, 512, 768 or 1024");
  if (max_slices < 0 || variant < 0 || variant > 6) return fail(HIPNMF_ERR_BAD_ARG, "bad tuning value");
  h->threads = threads;
  h->max_slices = max_slices;
  h->variant = variant;
  return HIPNMF_OK;
}

int hipnmf_fit_batched_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, float* W, float* H,
                           float* err_out, int32_t* n_iter_out, float* sse_col_out, float* xsq_col_out) {
  return fit_batched_impl<float>(h, p, X, W, H, err_out, n_iter_out, sse_col_out, xsq_col_out);
}
int hipnmf_fit_batched_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* W, double* H,
                           double* err_out, int32_t* n_iter_out, double* sse_col_out, double* xsq_col_out) {
  return fit_batched_impl<double>(h, p, X, W, H, err_out, n_iter_out, sse_col_out, xsq_col_out);
}

int hipnmf_fit_ragged_f32(hipnmf_handle* h, const hipnmf_problem* p, const int64_t* desc, const float* X, float* W,
                          float* H, float* err_out, int32_t* n_iter_out, float* sse_col_out, float* xsq_col_out) {
  if (!desc) return fail(HIPNMF_ERR_BAD_ARG, "desc is NULL");
  return fit_batched_impl<float>(h, p, X, W, H, err_out, n_iter_out, sse_col_out, xsq_col_out, desc);
}
int hipnmf_fit_ragged_f64(hipnmf_handle* h, const hipnmf_problem* p, const int64_t* desc, const double* X, double* W,
                          double* H, double* err_out, int32_t* n_iter_out, double* sse_col_out, double* xsq_col_out) {
  if (!desc) return fail(HIPNMF_ERR_BAD_ARG, "desc is NULL");
  return fit_batched_impl<double>(h, p, X, W, H, err_out, n_iter_out, sse_col_out, xsq_col_out, desc);
}

int hipnmf_shard_pass_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, float* W, const float* H,
                          float* sums) {
  return shard_pass_impl<float>(h, p, X, W, H, sums);
}
int hipnmf_shard_hupdate_f32(hipnmf_handle* h, const hipnmf_problem* p, float* H, const float* sums) {
  return shard_hupdate_impl<float>(h, p, H, sums);
}
int hipnmf_shard_residual_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, const float* W,
                              const float* H, float* sse_col, float* xsq_col) {
  return shard_residual_impl<float>(h, p, X, W, H, sse_col, xsq_col);
}
int hipnmf_shard_pass_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* W, const double* H,
                          double* sums) {
  return shard_pass_impl<double>(h, p, X, W, H, sums);
}
int hipnmf_shard_hupdate_f64(hipnmf_handle* h, const hipnmf_problem* p, double* H, const double* sums) {
  return shard_hupdate_impl<double>(h, p, H, sums);
}
int hipnmf_shard_residual_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, const double* W,
                              const double* H, double* sse_col, double* xsq_col) {
  return shard_residual_impl<double>(h, p, X, W, H, sse_col, xsq_col);
}

}  // extern "C"
