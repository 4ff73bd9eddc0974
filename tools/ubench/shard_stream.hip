// Micro-benchmark: the memory system's ceiling for the time-shard pass (config #5) -- channel-major X (16 rows of T
// floats), component-major W (5 rows of T floats, read AND written back), T = 2.5e7, no arithmetic to speak of.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/shard_stream.hip -o tools/ubench/bin/shard_stream
// Per row of the matrix: 64 B of X read, 20 B of W read, 20 B of W written = 104 B (the algorithmic bytes of the pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using f4 = float __attribute__((ext_vector_type(4)));

// MODE 0: read X and W, write W; 1: read only; 2: read X, write W without reading it
template <int MODE, int U>
__global__ void __launch_bounds__(256) k(const float* __restrict__ X, float* __restrict__ W, long long T, long long rps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * rps, r1 = (r0 + rps < T) ? r0 + rps : T;
  f4 keep = {0.f, 0.f, 0.f, 0.f};
  for (long long base = r0 + wave * 256LL * U; base < r1; base += 4LL * 256 * U) {
    f4 x[U][16], w[U][5];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long t = base + 256LL * u + 4 * lane;
      if (t < r1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[u][j] = *reinterpret_cast<const f4*>(X + j * T + t);
        if (MODE != 2) {
#pragma unroll
          for (int c = 0; c < 5; ++c) w[u][c] = *reinterpret_cast<const f4*>(W + c * T + t);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long t = base + 256LL * u + 4 * lane;
      if (t < r1) {
        f4 sx = x[u][0];
#pragma unroll
        for (int j = 1; j < 16; ++j) sx += x[u][j];
        if (MODE == 1) {
          keep += sx;
#pragma unroll
          for (int c = 0; c < 5; ++c) keep += w[u][c];
        } else {
#pragma unroll
          for (int c = 0; c < 5; ++c) *reinterpret_cast<f4*>(W + c * T + t) = (MODE == 2 ? sx : w[u][c] + sx * 1e-9f);
        }
      }
    }
  }
  if (MODE == 1 && keep[0] + keep[1] + keep[2] + keep[3] == 12345.f) W[0] = 1.f;
}

template <int MODE, int U>
void run(const float* X, float* W, long long T, int wgs, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  long long rps = ((T + wgs - 1) / wgs + 1023) / 1024 * 1024;
  const int grid = (int)((T + rps - 1) / rps);
  k<MODE, U><<<grid, 256>>>(X, W, T, rps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) k<MODE, U><<<grid, 256>>>(X, W, T, rps);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double bytes = (MODE == 0 ? 104.0 : (MODE == 1 ? 84.0 : 84.0)) * T;
  printf("  %-40s U=%d %5d workgroups: %7.3f ms  %6.2f TB/s moved (%5.2f TB/s on the pass's 104 B per row)\n", name, U, grid, ms,
         bytes / (ms * 1e-3) / 1e12, 104.0 * T / (ms * 1e-3) / 1e12);
}

int main() {
  const long long T = 25000000;
  float *X, *W;
  CK(hipMalloc(&X, 16 * T * 4));
  CK(hipMalloc(&W, 5 * T * 4));
  CK(hipMemset(X, 0, 16 * T * 4));
  CK(hipMemset(W, 0, 5 * T * 4));
  for (int wgs : {512, 1024, 2048, 4096}) {
    run<0, 1>(X, W, T, wgs, "read X + W, write W");
    run<0, 2>(X, W, T, wgs, "read X + W, write W");
  }
  run<1, 1>(X, W, T, 1024, "read X + W only");
  run<1, 2>(X, W, T, 2048, "read X + W only");
  run<2, 1>(X, W, T, 1024, "read X, write W (no W read)");
  return 0;
}
