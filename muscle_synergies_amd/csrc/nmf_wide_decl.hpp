// nmf_wide_decl.hpp -- host-side view of the wide-shape kernels of nmf_wide.hpp (instantiated in inst_wide_*.hip)
#pragma once
#include <cstddef>

#include "nmf_wide.hpp"

namespace hipnmf {
template <typename real>
struct WideKernel {
  void (*fn)(WideArgs<real>);
  void (*fn_kl)(WideArgs<real>);  // Kullback-Leibler flavour (256-thread instances only), nullptr otherwise
  size_t smem;  // dynamic LDS bytes
  int MP, KP, NW;
  const char* name;     // "fit_wide_kernel<real,MP,KP,NW,HREG,WPE,NSET,0>", as rocprofv3 prints the instance
  const char* name_kl;  // the Kullback-Leibler flavour's
};
// smallest compiled instance that holds n_features x n_components, nullptr beyond 128 channels / 16 components;
// nw = 8: the 512-thread instance (one workgroup per CU, the rest of LDS as W cache), nullptr where it does not exist
const WideKernel<float>* wide_kernel_f32(int n_features, int n_components, int nw);
const WideKernel<double>* wide_kernel_f64(int n_features, int n_components, int nw);
}  // namespace hipnmf
