// hipnmf_internal.hpp -- shared by the translation units that implement the C ABI (not installed).
#pragma once
#include "../../include/hip_nmf.h"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

struct hipnmf_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  void* aux = nullptr;  // second grow-only scratch: entry points that drive fit_batched_impl (which owns `ws`) keep theirs here
  size_t aux_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0.f;
  int threads = 0;     // 0 = default
  int max_slices = 0;  // 0 = default
  int variant = 0;     // 0 auto, 1 force persistent, 2 force sliced, 3 force cooperative
  int last_path = 0;   // path the last fit took: 1 persistent, 2 sliced, 3 cooperative
  bool coop_xcd_failed = false;  // the same-XCD cooperative mode found fewer workgroups than slices once: not tried again
  char last_kernel[96] = {0};  // instance name of the solver kernel the last fit launched (hipnmf_last_kernel)
  int num_cu = 256;
  int lds_per_block = 65536;  // hipDeviceProp_t::maxSharedMemoryPerMultiProcessor (160 KiB on MI355X)
  int lds_budget = 0;         // override (bytes), 0 = all of it
  int use_lds_w = 1;
  int use_graph = 1;
  int use_fuse_h = 0;  // HIPNMF_FUSE_H=1: sliced path, H update by the last slice of the pass (one launch per
                       // iteration).  Measured: with release/acquire fences 12.6 us per iteration (every workgroup
                       // writes its XCD's L2 back), fence-free 9.5 us vs 10.2 us for two launches on one 16 x 10 000
                       // matrix but slower for many slices (60.7 vs 50.6 us at T = 1e6): off by default
  int slice_threads_ok512 = 0;  // experiment: let hipnmf_set_tuning(threads=512) also apply to the sliced kernels
  int use_coop = 1;    // HIPNMF_COOP=0: never pick the cooperative kernel automatically
  int async_mode = 0;
  int path_batch_hint = 0;  // > 0: choose the solver path as for a batch of this size (rank sweep on a compacted sub-batch:
                            // every trial is fitted by the kernel the full batch would have used)
};

// sets the thread-local error text returned by hipnmf_last_error() and returns `code`
int hipnmf_fail(int code, const char* fmt, ...);
// grow-only workspace of the handle
int hipnmf_ensure_ws(hipnmf_handle* h, size_t bytes);

#define fail hipnmf_fail
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(HIPNMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                       \
  } while (0)

inline long long round_up(long long v, long long q) { return (v + q - 1) / q * q; }

// wide shapes (n_features > 32 or n_components > 8; hipnmf_wide.hip): the whole fit of a validated problem
template <typename real>
int hipnmf_fit_wide(hipnmf_handle* h, const hipnmf_problem* p, const real* X, real* W, real* H, real* err_out,
                    int32_t* n_iter_out, real* sse_col_out, real* xsq_col_out, const int64_t* ragged);
// hipnmf_shard_pass / _hupdate / _residual beyond the narrow lane mappings (hipnmf_wide.hip; op: 0 pass, 1 H update, 2 residual)
template <typename real>
int hipnmf_shard_wide(hipnmf_handle* h, const hipnmf_problem* p, int op, const real* X, real* W, real* H, real* sums,
                      real* sse_col, real* xsq_col);
// hipnmf_random_init_* for a compacted sub-batch: matrix b draws the numbers of matrix first_matrix + index[b] (hipnmf_init.hip)
template <typename real>
int hipnmf_random_init_indexed(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int first_matrix, const int* index,
                               const real* X, real* W, real* H);
// grow-only second scratch buffer of the handle (rank sweep, time-sharded fit): no hipMalloc / hipFree per call -- hipFree waits for
// every stream of the device, i.e. for the other host threads' work
int hipnmf_ensure_aux(hipnmf_handle* h, size_t bytes);
// Host <-> device copy ordered on the handle's stream and waited for.  Never the synchronous hipMemcpy: that is a legacy-(null-)stream
// operation, serialises against every blocking stream of the process and -- see hipnmf_kernel_chain -- breaks other threads' captures.
int hipnmf_copy(hipnmf_handle* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function, not of a launch: raised once per (function, device) to
// the whole LDS of the CU instead of per call with the call's own size (round 3 did that from concurrent threads)
int hipnmf_allow_full_lds(hipnmf_handle* h, const void* fn);

// ---- first use of a kernel function: under a process-wide lock -----------------------------------------------------------------
// The HIP runtime loads a code object lazily, at the first launch of one of its functions on a device.  On this runtime (ROCm 7.x
// CLR) two host threads that make such FIRST launches at the same time can crash inside hipLaunchKernel (SIGSEGV at address 0,
// 3 of 30 fresh processes whose worker threads start fitting at once -- the rank range of find_synergies; never once every kernel
// had been launched before, which is why the thread stress with its sequential reference pass never saw it:
// profiles/r06_abort_hunt.md).  Every launch of the library goes through HIPNMF_LAUNCH: a lock-free look-up of (function, device)
// in a table of functions already launched, and only on a miss the process-wide mutex around the launch.
class hipnmf_first_use {
 public:
  explicit hipnmf_first_use(const void* fn);
  ~hipnmf_first_use();
  hipnmf_first_use(const hipnmf_first_use&) = delete;
  hipnmf_first_use& operator=(const hipnmf_first_use&) = delete;

 private:
  const void* key_;
  bool locked_;
};
#define HIPNMF_LAUNCH(kern, grid, block, smem, stream, ...)                  \
  do {                                                                        \
    hipnmf_first_use first_use_guard_(reinterpret_cast<const void*>(kern));  \
    hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__);        \
  } while (0)

// Replayed launch sequences (the row-sliced paths: two small kernels per iteration, launch-bound) are hipGraphs built EXPLICITLY
// -- hipGraphCreate + hipGraphAddKernelNode, a linear chain -- and never by stream capture.  Root cause of round 3's "operation
// failed due to a previous error during capture" (profiles/r04_threads_root_cause.md): on this runtime (ROCm 7.x CLR) ANY
// legacy-stream call of ANY thread of the process -- a plain hipMemcpy of the host application, torch's null-stream traffic, this
// library's own hipMemcpy in another handle's call -- returns hipErrorStreamCaptureImplicit while some stream is capturing AND
// invalidates that capture, even a hipStreamCaptureModeThreadLocal capture on a hipStreamNonBlocking stream; hipStreamEndCapture
// then returns hipErrorStreamCaptureInvalidated WITHOUT leaving capture mode, so every later call on that stream fails too.  A
// library cannot know what the other threads of its host do, so it must not open captures at all.  Building the graph node by node
// puts no stream into capture mode: nothing another thread does can invalidate it, and nothing it does can fail because of us.
class hipnmf_kernel_chain {
 public:
  hipnmf_kernel_chain() = default;
  hipnmf_kernel_chain(const hipnmf_kernel_chain&) = delete;
  hipnmf_kernel_chain& operator=(const hipnmf_kernel_chain&) = delete;
  ~hipnmf_kernel_chain() {
    if (exec_) (void)hipGraphExecDestroy(exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
  }
  // appends kernel `fn`(args) after the previous node; the argument struct is copied (and kept until the chain dies)
  template <typename Args>
  hipError_t add(const void* fn, dim3 grid, dim3 block, size_t smem, const Args& args) {
    hipError_t e = hipSuccess;
    if (!graph_ && (e = hipGraphCreate(&graph_, 0)) != hipSuccess) return e;
    slots_.emplace_back(new Slot);
    Slot& s = *slots_.back();
    s.bytes.reset(new (std::align_val_t(16)) unsigned char[sizeof(Args)]);
    std::memcpy(s.bytes.get(), &args, sizeof(Args));
    s.params[0] = s.bytes.get();
    hipKernelNodeParams np;
    std::memset(&np, 0, sizeof(np));
    np.func = const_cast<void*>(fn);
    np.gridDim = grid;
    np.blockDim = block;
    np.sharedMemBytes = (unsigned)smem;
    np.kernelParams = s.params;
    np.extra = nullptr;
    hipGraphNode_t node = nullptr;
    hipnmf_first_use first_use_guard_(fn);  // (resolving the function for the node loads its code object like a launch does)
    e = hipGraphAddKernelNode(&node, graph_, n_nodes_ ? &last_ : nullptr, n_nodes_ ? 1 : 0, &np);
    if (e != hipSuccess) return e;
    last_ = node;
    ++n_nodes_;
    return hipSuccess;
  }
  hipError_t instantiate() { return hipGraphInstantiate(&exec_, graph_, nullptr, nullptr, 0); }
  hipError_t launch(hipStream_t st) { return hipGraphLaunch(exec_, st); }
  int nodes() const { return n_nodes_; }

 private:
  struct AlignedDelete {
    void operator()(unsigned char* p) const { ::operator delete[](p, std::align_val_t(16)); }
  };
  struct Slot {
    std::unique_ptr<unsigned char[], AlignedDelete> bytes;
    void* params[1] = {nullptr};
  };
  hipGraph_t graph_ = nullptr;
  hipGraphExec_t exec_ = nullptr;
  hipGraphNode_t last_ = nullptr;
  int n_nodes_ = 0;
  std::vector<std::unique_ptr<Slot>> slots_;
};


// ---- the fitted routing constants, in ONE table ---------------------------------------------------------------------------------
// Which kernel family serves a call is decided by rules whose thresholds were MEASURED on one box type (MI355X, 256 CUs; the A/B
// behind each rule is cited where the rule is applied: hipnmf_api.hip wide_preferred / fit_batched_impl, hipnmf_wide.hip).  They
// are data, not code: every one lives here with its default, and HIPNMF_ROUTES="name=value,name=value" (read once per process)
// overrides any of them -- the string tools/calibrate_routes.py prints after re-deriving the crossovers on the box at hand.
struct hipnmf_route_table {
  // -- narrow shapes (<= 32 channels, <= 8 components), chip-filling or ragged batches: lane mappings (fit_persistent_kernel,
  //    fit_rowlane_kernel) vs the 4x4 matrix-pipe kernels (fit_wide4_kernel / fit_wide4d_kernel); "wide up to this many rows"
  double f32_16ch_wide_max_rows = 600;        // fp32, <= 16 channels, k <= 6
  double f32_16ch_k7_wide_max_rows = 1200;    // fp32, <= 16 channels, k = 7, 8
  double f64_16ch_wide_max_rows = 1200;       // float64, <= 16 channels, k <= 6 (k = 7, 8: always the matrix pipe)
  double f32_32ch_wide_max_rows = 2400;       // fp32, 17..32 channels, k <= 4
  double f32_32ch_k5_wide_max_rows = 3000;    // fp32, 17..32 channels, k = 5, 6 (k = 7, 8: always); 5000 until round 6: tools/calibrate_routes.py
                                              // measured the crossover at 2 900 rows (5 000 rows: 14.2 ms lanes, 16.3 ms matrix pipe; profiles/r06_calibrate_routes.log)
  double kl_f32_32ch_short_max_rows = 1500;   // Kullback-Leibler fp32, 17..32 channels, k <= 5: matrix pipe up to this many rows
  double wide_min_batch_cus = 0.5;            // "chip-filling": at least this many matrices per CU
  double small_long_min_batch_cus_f32 = 2;    // one wave per matrix beyond 256 rows: matrices per CU from which it wins
  double small_long_min_batch_cus_f64 = 3;
  // -- one workgroup per matrix vs row slices vs the cooperative kernel, lane mappings (seconds; tools/config2_bench.py)
  double pers_s_per_row_rowmajor = 2.0e-9, pers_s_per_row = 2.7e-9, pers_s_fixed = 1e-6;
  double sliced_s_launches = 9.5e-6, sliced_s_per_row = 0.021e-9;
  // -- Kullback-Leibler on few long matrices: one workgroup per matrix vs the row-sliced one-pass kernel (ms per 100 iterations;
  //    tools/probes/kl_long_ab.sh): one = (a + b m) per 1 000 rows; sliced = launches + per_slice S + per_row rows waves
  double kl_one_f32_a = 0.35, kl_one_f32_b = 0.0265, kl_one_f64_a = 0.3, kl_one_f64_b = 0.06;
  double kl_lane_f32_k5_per_ch = 0.0225, kl_lane_f32_per_ch = 0.035, kl_lane_f64_per_ch = 0.05;  // the lane mappings' own rates
  double kl_sliced_launches = 1.0, kl_sliced_per_slice = 0.015, kl_sliced_per_row = 0.005, kl_sliced_f64_factor = 1.9,
         kl_sliced_wide_factor = 0.6, kl_sliced_margin = 0.9;
};
// the process's table: the defaults above with HIPNMF_ROUTES applied (unknown names are reported once on stderr and ignored)
const hipnmf_route_table& hipnmf_routes();
// ("name=value" pairs of the table in force: hipnmf_routes_describe, include/hip_nmf.h)

constexpr int HIPNMF_NARROW_MAX_FEATURES = 32, HIPNMF_NARROW_MAX_COMPONENTS = 8;  // nmf_kernels.hpp lane mappings
constexpr int HIPNMF_MAX_FEATURES = 512, HIPNMF_MAX_COMPONENTS = 64;              // nmf_big.hpp (nmf_wide.hpp: 128 / 32)

// Kullback-Leibler, few long matrices: does the row-sliced one-pass general-shape kernel (nmf_big1.hpp; components padded to 16, channels
// to 64) beat one workgroup per matrix (the only form the other families have for this loss)?  hipnmf_wide.hip; `t_one_per_krow`: ms per
// 1 000 rows and 100 iterations of the one-workgroup kernel the call would otherwise run (< 0: the 4x4 kernels' fitted rate).
bool hipnmf_kl_row_sliced_wins(bool f64, int m, long long T, int B, int num_cu, double t_one_per_krow, int max_slices = 0);

// X layout canonicalisation kernel (nmf_kernels.hpp) reused by the envelope entry point
