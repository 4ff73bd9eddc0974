// Micro-benchmark: issue rate of scalar vs packed fp32 VALU ops on gfx950 (one to four waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x[8]; f2 y[8]; float z[8], w[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 1.f}; z[i] = x[i] * 0.37f + a; w[i] = x[i] * 0.11f + b; }
  f2 av{a, a * 1.0001f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);                       // v_fma_f32
        if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], av, f2{b, b});      // v_pk_fma_f32
        if (MODE == 2) x[i] = __builtin_amdgcn_rcpf(x[i]) + a;                    // v_rcp_f32 + add
        if (MODE == 3) { int v = __builtin_bit_cast(int, x[i]); v = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, true); x[i] = __builtin_bit_cast(float, v) + a; }  // dpp mov + add
        if (MODE == 4) x[i] = (x[i] > a) ? x[i] * b : a;                           // cmp+cndmask+mul
        if (MODE == 5) x[i] = x[i] / (a + x[i]);                                  // IEEE div
        if (MODE == 6) x[i] = __builtin_fmaf(z[i], w[(i + u) & 7], x[i]);          // v_fmac with 3 VGPR operands
        if (MODE == 7) x[i] = (threadIdx.x & 2) ? z[i] : x[(i + 1) & 7];           // v_cndmask (mask in SGPR pair)
        if (MODE == 8) x[i] = __builtin_fmaf(z[i], a, x[i]);                       // v_fmac VGPR x SGPR + VGPR
      }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y + z[i] + w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char* name, int opsPerInner) {
  float* d; CHECK(hipMalloc(&d, 256 * 1024 * 4 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wavesPerSimd : {1, 2, 4}) {
    int threads = 256 * wavesPerSimd; if (threads > 1024) threads = 1024;
    int blocks = 256 * (wavesPerSimd * 256 / threads);
    k<MODE><<<blocks, threads>>>(d, 10, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = (double)iters * 64;  // 8x8 inner statements
    double waves_per_simd = (double)blocks * threads / 64 / 1024;
    double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.3f ns per statement per SIMD (%.2f cycles @2.4GHz)\n", name, wavesPerSimd, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
  }
  return 0;
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<2>("rcp+add", 2); run<3>("dppmov+add", 2); run<4>("cmp+cnd+mul", 3); run<5>("ieee_div", 10); run<6>("fmac_3vgpr", 1); run<7>("cndmask", 1); run<8>("fmac_vsv", 1);
  return 0;
}
