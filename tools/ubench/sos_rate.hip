// Micro-benchmark: the fp64 second-order-section recursion of sosfilt_kernels.hpp in registers (no memory), one wave
// per SIMD as in the kernel (512 waves for 16 384 series).
//   A  one lane per series, NS sections in sequence per sample (round 1)          : 9 NS fp64 instructions per sample
//   B  NS lanes per series, one section each, the intermediate handed to the next lane by DPP row_shr:1 (round 2):
//      9 fp64 + 2 DPP instructions per sample whatever NS
//   C  as B with part of the wave masked off (EXEC): measures whether idle 16-lane groups are skipped
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/sos_rate.hip -o tools/ubench/bin/sos_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NS>
__device__ __forceinline__ double step_seq(double xc, double (&z)[NS][2], const double (&c)[NS][5]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const double xn = c[s][0] * xc + z[s][0];
    z[s][0] = c[s][1] * xc - c[s][3] * xn + z[s][1];
    z[s][1] = c[s][2] * xc - c[s][4] * xn;
    xc = xn;
  }
  return xc;
}

template <int BANK_MASK>
__device__ __forceinline__ double dpp_shr1_keep(double keep, double src) {  // lanes enabled by BANK_MASK take lane-1's src
  const unsigned long long k = __builtin_bit_cast(unsigned long long, keep), s = __builtin_bit_cast(unsigned long long, src);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)k, (int)(unsigned)s, 0x111, 0xf, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(k >> 32), (int)(unsigned)(s >> 32), 0x111, 0xf, BANK_MASK, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int NS>
__global__ void __launch_bounds__(64) k_seq(double* out, int n, double seed) {
  double c[NS][5], z[NS][2];
  for (int s = 0; s < NS; ++s) {
    for (int q = 0; q < 5; ++q) c[s][q] = seed * 0.01 * (1 + q + s);
    z[s][0] = z[s][1] = 0.0;
  }
  double x = seed + threadIdx.x * 1e-3, acc = 0.0;
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      asm volatile("" : "+v"(x));
      acc += step_seq<NS>(x, z, c);
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int LPS>
__global__ void __launch_bounds__(64) k_lanes(double* out, int n, double seed, int active = 64) {
#pragma clang fp contract(off)
  if ((int)threadIdx.x >= active) return;  // C: does a partly filled wave run its fp64 passes faster?
  constexpr int BM = LPS == 2 ? 0xA : (LPS == 4 ? 0xE : 0xF);
  const int s = threadIdx.x % LPS;
  double c0 = seed * 0.01 * (1 + s), c1 = seed * 0.02 * (1 + s), c2 = seed * 0.03, c3 = seed * 0.04, c4 = seed * 0.05;
  double z0 = 0.0, z1 = 0.0, xn = 0.0, acc = 0.0;
  double x = seed + threadIdx.x * 1e-3;
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      asm volatile("" : "+v"(x));
      const double xin = LPS > 1 ? dpp_shr1_keep<BM>(x, xn) : x;
      xn = c0 * xin + z0;
      z0 = c1 * xin - c3 * xn + z1;
      z1 = c2 * xin - c4 * xn;
      acc += xn;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <typename F>
void run(F launch, const char* name, int ns) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n = 200000;
  for (int waves : {512, 1024, 2048}) {
    launch(waves, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(waves, n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-36s NS=%d %4d waves: %7.2f ns per sample (all sections) and wave\n", name, ns, waves, ms * 1e6 / n);
  }
}

int main() {
  double* d;
  CK(hipMalloc(&d, 2048 * 64 * 8));
  run([&](int w, int n) { k_seq<1><<<w, 64>>>(d, n, 0.7); }, "A one lane per series", 1);
  run([&](int w, int n) { k_seq<2><<<w, 64>>>(d, n, 0.7); }, "A one lane per series", 2);
  run([&](int w, int n) { k_seq<4><<<w, 64>>>(d, n, 0.7); }, "A one lane per series", 4);
  run([&](int w, int n) { k_lanes<1><<<w, 64>>>(d, n, 0.7); }, "B one lane per section", 1);
  run([&](int w, int n) { k_lanes<2><<<w, 64>>>(d, n, 0.7); }, "B one lane per section", 2);
  run([&](int w, int n) { k_lanes<4><<<w, 64>>>(d, n, 0.7); }, "B one lane per section", 4);
  run([&](int w, int n) { k_lanes<2><<<w, 64>>>(d, n, 0.7, 32); }, "C as B, 32 of 64 lanes active", 2);
  run([&](int w, int n) { k_lanes<2><<<w, 64>>>(d, n, 0.7, 16); }, "C as B, 16 of 64 lanes active", 2);
  return 0;
}
