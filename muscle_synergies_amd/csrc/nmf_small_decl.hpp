// nmf_small_decl.hpp -- host-side view of fit_small_kernel (nmf_small.hpp, instantiated in inst_small*.hip)
#pragma once
#include "nmf_kernels.hpp"

namespace hipnmf {
constexpr int SMALL_NT = 4;  // 64-row tiles of the default instances (n_samples <= 256)
template <typename real>
using SmallFn = void (*)(SolveArgs<real>);
// fit_small_kernel<real, CH, K, 4> for CH = 8 (m <= 8) or 16 (m <= 16, fp32 only), n_samples <= 256; nullptr when not compiled
template <typename real>
SmallFn<real> small_kernel(int m, int K);
// the instances with more tiles in registers: the smallest NT in {8, 12, 16} with 64 NT >= n_samples that is compiled for
// the shape (fp32: NT = 16 up to 5 components, 12 beyond; fp64: 8 channels, up to 6 components, NT <= 12); *nt_out = NT
template <typename real>
SmallFn<real> small_kernel_long(int m, int K, long long n_samples, int* nt_out);
template <typename real>
size_t small_smem_bytes(int m, int K);
}  // namespace hipnmf
