// nmf_rowlane_decl.hpp -- host-side view of the kernels of nmf_rowlane.hpp (instantiated in inst_f32_rowlane.hip)
#pragma once
#include "nmf_kernels.hpp"

namespace hipnmf {
using RowLaneFn = void (*)(SolveArgs<float>);
RowLaneFn rowlane_kernel(int K);        // fit_rowlane_kernel<K>, nullptr outside 1..8
const char* rowlane_kernel_name(int K); // "fit_rowlane_kernel<K,NXR,NWR,PF>"
RowLaneFn rowlane_kernel_kl(int K);     // fit_rowlane_kernel<K, 0, 0, 1, LOSS = 1>: Kullback-Leibler loss, nullptr outside 1..8
RowLaneFn slice_pass_rowlane(int K);    // slice_pass_rowlane_kernel<K> (time-shard pass on channel-major X), nullptr outside 1..8
}  // namespace hipnmf
