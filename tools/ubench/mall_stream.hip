// Micro-benchmark: what the memory system delivers for the persistent kernel's access pattern -- one workgroup per
// CU (forced by 150 KB of dynamic LDS), each streaming ITS OWN region (a matrix: 640 KB of X, or 733 KB with the
// un-cached rows of W) `passes` times, B regions in total (B = 4096 -> 2.6 GB, far beyond the Infinity Cache; the 256
// regions being streamed at any time, 164 MB, fit in it).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mall_stream.hip -o tools/ubench/bin/mall_stream
// Knobs: loads in flight per wave (U x 1 KB), the part of each region loaded with the default cache policy (the rest is
// loaded non-temporal, so that it does not evict the first part from the XCD's 4 MB L2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using rsrc_t = __amdgpu_buffer_rsrc_t;
using u4 = unsigned __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__device__ __forceinline__ void sweep(rsrc_t r, unsigned begin, unsigned end, unsigned lane_off, unsigned stride, u4& acc) {
  // each wave walks its own 1 KB pieces: piece p of the wave at begin + (p * nwaves + wave) KB
  unsigned off = begin;
  for (; off + (U - 1) * stride < end; off += U * stride) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, off + u * stride, NT ? 2 : 0);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  for (; off < end; off += stride) acc ^= __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, off, NT ? 2 : 0);
}

template <int U>
__global__ void __launch_bounds__(512) k_stream(const char* base, size_t region, unsigned keep_bytes, int passes, unsigned* out) {
  extern __shared__ char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63, nw = blockDim.x / 64;
  const char* p = base + (size_t)blockIdx.x * region;
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, (int)region, 0x00020000);
  u4 acc = {0, 0, 0, 0};
  const unsigned stride = nw * 1024u;
  for (int it = 0; it < passes; ++it) {
    sweep<U, false>(r, wave * 1024u, keep_bytes, lane * 16u, stride, acc);
    sweep<U, true>(r, keep_bytes + wave * 1024u, (unsigned)region, lane * 16u, stride, acc);
    asm volatile("" : "+v"(acc));
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;  // keeps the loads alive
  if (threadIdx.x == 0 && lds[threadIdx.x] == 77) out[0] = 2;
}

template <int U>
void run(const char* d, unsigned* out, int B, size_t region, unsigned keep, int passes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k_stream<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  k_stream<U><<<B, 512, 150 * 1024>>>(d, region, keep, 2, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_stream<U><<<B, 512, 150 * 1024>>>(d, region, keep, passes, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  B=%5d region=%4zu KB default-policy part=%4u KB loads in flight per wave=%d : %7.2f ms  %6.2f TB/s\n", B,
         region / 1024, keep / 1024, U, ms, (double)B * region * passes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int passes = argc > 2 ? atoi(argv[2]) : 100;
  const size_t region_max = 768 * 1024;
  char* d;
  unsigned* out;
  CK(hipMalloc(&d, (size_t)B * region_max));
  CK(hipMalloc(&out, B * 4));
  CK(hipMemset(d, 1, (size_t)B * region_max));
  for (size_t region : {(size_t)640 * 1024, (size_t)736 * 1024, (size_t)320 * 1024}) {
    run<1>(d, out, B, region, (unsigned)region, passes);
    run<2>(d, out, B, region, (unsigned)region, passes);
    run<4>(d, out, B, region, (unsigned)region, passes);
    run<8>(d, out, B, region, (unsigned)region, passes);
  }
  printf("part of each region kept out of the non-temporal stream (L2 share per CU: 128 KB)\n");
  for (unsigned keep_kb : {0u, 64u, 96u, 128u, 192u}) run<4>(d, out, B, 640 * 1024, keep_kb * 1024, passes);
  printf("256 regions only (everything resident)\n");
  run<4>(d, out, 256, 640 * 1024, 640 * 1024, passes * 4);
  run<8>(d, out, 256, 640 * 1024, 640 * 1024, passes * 4);
  return 0;
}
