// hipnmf_diag.hip -- measurement aids exported beside the solver (include/hip_nmf.h, "diagnostics"): the rate at which the
// memory system serves the headline kernel's access pattern, measured in the run that quotes it (bench.py roofline.memory).
#include <cstdint>

#include "hipnmf_internal.hpp"

namespace {

using rsrc_t = __amdgpu_buffer_rsrc_t;
using u4 = unsigned __attribute__((ext_vector_type(4)));

// One workgroup per region (a matrix of the batch): its eight waves walk the region `passes` times in 1 KB pieces, U 16-byte
// loads in flight per lane -- what fit_persistent_kernel does to its X, minus the arithmetic (tools/ubench/mall_stream.hip is
// the stand-alone original).  150 KB of dynamic LDS keep it at one workgroup per CU like the solver.
template <int U>
__global__ void __launch_bounds__(512) diag_stream_kernel(const char* base, size_t region, int passes, unsigned* out) {
  extern __shared__ char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63, nw = blockDim.x / 64;
  const char* p = base + (size_t)blockIdx.x * region;
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, (int)region, 0x00020000);
  u4 acc = {0, 0, 0, 0};
  const unsigned stride = nw * 1024u, end = (unsigned)region, lane_off = lane * 16u;
  for (int it = 0; it < passes; ++it) {
    unsigned off = wave * 1024u;
    for (; off + (U - 1) * stride < end; off += U * stride) {
      u4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, off + u * stride, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    for (; off < end; off += stride) acc ^= __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, off, 0);
    asm volatile("" : "+v"(acc));
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;  // keeps the loads alive
  if (threadIdx.x == 0 && lds[threadIdx.x] == 77) out[0] = 2;
}

}  // namespace

extern "C" int hipnmf_diag_stream_gbs(hipnmf_handle* h, int64_t region_bytes, int32_t regions, int32_t passes, double* gbs_out) {
  if (!h || !gbs_out) return fail(HIPNMF_ERR_BAD_ARG, "NULL argument");
  if (region_bytes < 4096 || region_bytes >= (1LL << 31) || (region_bytes % 1024) || regions < 1 || passes < 1)
    return fail(HIPNMF_ERR_BAD_ARG, "region_bytes must be a multiple of 1024 in [4096, 2 GiB), regions and passes >= 1");
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = (size_t)regions * (size_t)region_bytes;
  int rc = hipnmf_ensure_ws(h, bytes + 4 * (size_t)regions + 256);
  if (rc) return rc;
  char* d = static_cast<char*>(h->ws);
  unsigned* out = reinterpret_cast<unsigned*>(d + (bytes + 255) / 256 * 256);
  hipStream_t st = h->stream;
  HIP_TRY(hipMemsetAsync(d, 1, bytes, st));
  const void* kern = reinterpret_cast<const void*>(diag_stream_kernel<2>);
  if ((rc = hipnmf_allow_full_lds(h, kern))) return rc;
  const size_t lds = 150 * 1024 <= (size_t)h->lds_per_block ? 150 * 1024 : (size_t)h->lds_per_block;
  HIPNMF_LAUNCH(diag_stream_kernel<2>, dim3(regions), dim3(512), lds, st, (const char*)d, (size_t)region_bytes, 2, out);  // warm-up
  HIP_TRY(hipEventRecord(h->ev0, st));
  HIPNMF_LAUNCH(diag_stream_kernel<2>, dim3(regions), dim3(512), lds, st, (const char*)d, (size_t)region_bytes, (int)passes, out);
  HIP_TRY(hipEventRecord(h->ev1, st));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *gbs_out = (double)bytes * (double)passes / ((double)ms * 1e-3) / 1e9;
  return HIPNMF_OK;
}
