// Micro-benchmark: what HBM delivers for the wide-shape kernel's traffic with no arithmetic (round 3).
// One 256-thread workgroup per matrix (as fit_wide_kernel), B matrices far beyond the Infinity Cache; every wave walks
// its 16-row subtiles: reads the subtile of X (16 rows x XB bytes, whole rows, 16 bytes per lane), reads the 16 rows of
// W (WB bytes each) and -- mode 2 -- writes them back.  Modes: 0 = X only, 1 = X + W read, 2 = X + W read + W write.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wide_stream.hip -o tools/ubench/bin/wide_stream
//   wide_stream [B] [T] [XB] [WB] [passes] [dynamic LDS bytes per workgroup: limits the workgroups per CU]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using rsrc_t = __amdgpu_buffer_rsrc_t;
using u4 = unsigned __attribute__((ext_vector_type(4)));

template <int U, int MODE, int NX, int XAUX>
__global__ void __launch_bounds__(256) k_wide_stream(const char* X, char* W, int T, int xb, int wb, int passes, unsigned* out) {
  extern __shared__ char lds_[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63;
  const size_t xbytes = (size_t)T * xb, wbytes = (size_t)T * wb;
  rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(X + blockIdx.x * xbytes), 0, (int)xbytes, 0x00020000);
  rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(W + blockIdx.x * wbytes, 0, (int)wbytes, 0x00020000);
  const int xtile = 16 * xb, wtile = 16 * wb;  // bytes per subtile
  const unsigned woff = lane * 16u < (unsigned)wtile ? lane * 16u : 0x80000000u;
  const int ntiles = T / 16;
  u4 acc = {0, 0, 0, 0};
  for (int it = 0; it < passes; ++it) {
    for (int i = wave; i < ntiles; i += 4 * U) {
      u4 v[U][NX], w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ii = i + 4 * u;
        {
          _Pragma("unroll") for (int n = 0; n < NX; ++n) v[u][n] = __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16u + n * 1024u, (unsigned)ii * xtile, XAUX);
          if (MODE >= 1) w[u] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff, (unsigned)ii * wtile, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ii = i + 4 * u;
        {
          _Pragma("unroll") for (int n = 0; n < NX; ++n) acc ^= v[u][n];
          if (MODE >= 1) acc ^= w[u];
          if (MODE >= 2) __builtin_amdgcn_raw_buffer_store_b128(w[u] + acc[0] * 0u, wr, woff, (unsigned)ii * wtile, 0);
        }
      }
    }
    asm volatile("" : "+v"(acc));
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;
  if (threadIdx.x == 0 && lds_[0] == 77) out[0] = 2;
}

static int g_lds = 0;
template <int U, int MODE, int NX, int XAUX = 0>
void run_(const char* x, char* w, unsigned* out, int B, int T, int xb, int wb, int passes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k_wide_stream<U, MODE, NX, XAUX>, hipFuncAttributeMaxDynamicSharedMemorySize, g_lds > 1024 ? g_lds : 1024));
  k_wide_stream<U, MODE, NX, XAUX><<<B, 256, g_lds>>>(x, w, T, xb, wb, 1, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_wide_stream<U, MODE, NX, XAUX><<<B, 256, g_lds>>>(x, w, T, xb, wb, passes, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)B * T * (xb + (MODE >= 1 ? wb : 0) + (MODE >= 2 ? wb : 0)) * passes;
  printf("  B=%5d T=%d X %d B/row W %d B/row mode %d (%s) X cache policy %d subtiles in flight per wave=%d : %7.2f ms  %6.2f TB/s\n", B, T, xb, wb, MODE,
         MODE == 0 ? "X only" : MODE == 1 ? "X + W read" : "X + W read + W write", XAUX, U, ms, bytes / (ms * 1e-3) / 1e12);
}

template <int U, int MODE>
void run(const char* x, char* w, unsigned* out, int B, int T, int xb, int wb, int passes) {
  if (xb == 256) {
    run_<U, MODE, 4>(x, w, out, B, T, xb, wb, passes);
    if (MODE == 2 || U == 2) { run_<U, MODE, 4, 2>(x, w, out, B, T, xb, wb, passes); run_<U, MODE, 4, 1>(x, w, out, B, T, xb, wb, passes); run_<U, MODE, 4, 3>(x, w, out, B, T, xb, wb, passes); run_<U, MODE, 4, 17>(x, w, out, B, T, xb, wb, passes); }
  }
  else if (xb == 512) run_<U, MODE, 8>(x, w, out, B, T, xb, wb, passes);
  else if (xb == 128) run_<U, MODE, 2>(x, w, out, B, T, xb, wb, passes);
  else printf("X row bytes must be 128, 256 or 512\n");
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024;
  const int T = argc > 2 ? atoi(argv[2]) : 10000;
  const int xb = argc > 3 ? atoi(argv[3]) : 256;
  const int wb = argc > 4 ? atoi(argv[4]) : 32;
  const int passes = argc > 5 ? atoi(argv[5]) : 20;
  g_lds = argc > 6 ? atoi(argv[6]) : 0;
  printf("dynamic LDS per workgroup: %d bytes\n", g_lds);
  char *x, *w;
  unsigned* out;
  CK(hipMalloc(&x, (size_t)B * T * xb));
  CK(hipMalloc(&w, (size_t)B * T * wb));
  CK(hipMalloc(&out, B * 4));
  CK(hipMemset(x, 1, (size_t)B * T * xb));
  CK(hipMemset(w, 1, (size_t)B * T * wb));
  run<1, 0>(x, w, out, B, T, xb, wb, passes);
  run<2, 0>(x, w, out, B, T, xb, wb, passes);
  run<4, 0>(x, w, out, B, T, xb, wb, passes);
  run<2, 1>(x, w, out, B, T, xb, wb, passes);
  run<1, 2>(x, w, out, B, T, xb, wb, passes);
  run<2, 2>(x, w, out, B, T, xb, wb, passes);
  run<4, 2>(x, w, out, B, T, xb, wb, passes);
  return 0;
}
