// waitcnt_probe.hip -- what makes hipcc emit s_waitcnt vmcnt(0) instead of vmcnt(N) in a loop with two register sets of loads in
// flight: compile with  hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S [-DTAIL] [-DBR]  and grep s_waitcnt.
// As written (prologue requests unordered) the top of the loop waits vmcnt(3..0); with __builtin_amdgcn_sched_barrier(0) between the two
// prologue requests (-DORDERED) it waits vmcnt(7..4); a tail that requests set a again (-DTAIL) brings vmcnt(3..0) back.
#include <hip/hip_runtime.h>
using rsrc_t = __amdgpu_buffer_rsrc_t;
using u4 = unsigned __attribute__((ext_vector_type(4)));
__device__ inline rsrc_t mk(const void* p, unsigned n) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, 0x00020000); }
extern "C" __global__ void k(const float* X, float* out, int n, int T, int iters) {
  __shared__ float s[64 * 4 * 4];
  __shared__ float red[64];
  unsigned voff = threadIdx.x * 16u;
  u4 a[4], b[4];
  float acc = 0;
  auto rs = [&](int i) { const int rows = i < n ? T - 16 * i : 0; return mk((const char*)X + (rows > 0 ? 16 * i : 0) * 1024, (unsigned)rows * 1024u); };
  auto issue = [&](u4 (&t)[4], int i) { rsrc_t r = rs(i);
    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + q * 1024u, 0u, 2); };
  auto work = [&](u4 (&t)[4], int i, int inext) {
    for (int q = 0; q < 4; ++q) *(u4*)&s[(q * 64 + threadIdx.x) * 4] = t[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    issue(t, inext);
    for (int q = 0; q < 16; ++q) acc += s[(q * 16 + threadIdx.x * 7) & 1023];
#ifdef BR
    if (__any(acc < 1e-30f)) acc = acc / (acc + 1.0f); else acc = acc * __builtin_amdgcn_rcpf(acc + 1.0f);
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
#ifdef ORDERED
  issue(a, 0); __builtin_amdgcn_sched_barrier(0); issue(b, 1); __builtin_amdgcn_sched_barrier(0);
#else
  issue(a, 0); issue(b, 1);
#endif
  for (int it = 0; it < iters; ++it) {
    int i = 0;
    for (; i + 1 < n; i += 2) { work(a, i, i + 2); __builtin_amdgcn_sched_barrier(0); work(b, i + 1, i + 3); __builtin_amdgcn_sched_barrier(0); }
#ifdef TAIL
    if (i < n) work(a, i, i + 2);
#endif
    if (it + 1 < iters) { issue(a, 0); issue(b, 1); }
    red[threadIdx.x] = acc;
    __syncthreads();
    acc = red[(threadIdx.x + 1) & 63];
    __syncthreads();
  }
  out[threadIdx.x] = acc;
}
