// hipnmf_internal.hpp -- shared by the translation units that implement the C ABI (not installed).
#pragma once
#include "../../include/hip_nmf.h"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>

struct hipnmf_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
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
// hipnmf_random_init_* for a compacted sub-batch: matrix b draws the numbers of matrix first_matrix + index[b] (hipnmf_init.hip)
template <typename real>
int hipnmf_random_init_indexed(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int first_matrix, const int* index,
                               const real* X, real* W, real* H);
// Stream captures (the graph-replayed row-sliced paths) are serialised across the host threads of a process.  A precaution, not a
// cure: three host threads driving the wide row-sliced path at once still ended with "operation failed due to a previous error
// during capture" (tools/repro/rank_threads_long_matrix.py with REPRO_UNLIMITED=1; the cooperative kernel and the narrow sliced path
// ran fine side by side), which is why the Python host fits the ranks of long or wide frames in a loop (analysis.py).
std::mutex& hipnmf_capture_mutex();

constexpr int HIPNMF_NARROW_MAX_FEATURES = 32, HIPNMF_NARROW_MAX_COMPONENTS = 8;  // nmf_kernels.hpp lane mappings
constexpr int HIPNMF_MAX_FEATURES = 128, HIPNMF_MAX_COMPONENTS = 32;              // nmf_wide.hpp

// X layout canonicalisation kernel (nmf_kernels.hpp) reused by the envelope entry point
