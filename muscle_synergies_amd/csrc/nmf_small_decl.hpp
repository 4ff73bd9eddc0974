// nmf_small_decl.hpp -- host-side view of fit_small_kernel (nmf_small.hpp, instantiated in inst_small.hip)
#pragma once
#include "nmf_kernels.hpp"

namespace hipnmf {
template <typename real>
using SmallFn = void (*)(SolveArgs<real>);
// fit_small_kernel<real, CH, K> for CH = 8 (m <= 8) or 16 (m <= 16, fp32 only); nullptr when not compiled
template <typename real>
SmallFn<real> small_kernel(int m, int K);
template <typename real>
size_t small_smem_bytes(int m, int K);
}  // namespace hipnmf
