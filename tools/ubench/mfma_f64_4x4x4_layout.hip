// mfma_f64_4x4x4_layout.hip -- lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of D(4x4) += A(4x4) B(4x4)), probed: for every
// pair (la, lb) one launch-lane sets A = 1 in lane la and B = 1 in lane lb (all else 0) and reports which lane of D becomes 1.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_f64_4x4x4_layout.hip -o tools/ubench/bin/mfma_f64_4x4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {  // grid (64, 64): blockIdx.x = la, blockIdx.y = lb
  const int lane = threadIdx.x;
  const double a = lane == (int)blockIdx.x ? 1.0 : 0.0;
  const double b = lane == (int)blockIdx.y ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  if (d != 0.0) out[blockIdx.x * 64 + blockIdx.y] = lane;
}
int main() {
  int* d_out;
  std::vector<int> h(64 * 64, -1);
  hipMalloc(&d_out, sizeof(int) * 64 * 64);
  hipMemcpy(d_out, h.data(), sizeof(int) * 64 * 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, d_out);
  hipMemcpy(h.data(), d_out, sizeof(int) * 64 * 64, hipMemcpyDeviceToHost);
  // for every A lane: the B lanes it meets and where the product lands
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
