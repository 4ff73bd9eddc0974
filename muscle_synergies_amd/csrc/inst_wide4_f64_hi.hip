// fit_wide4d_kernel<MP, KQ, 4>, MP = 96, 128: one wave per SIMD (nmf_wide4d.hpp)
#include "nmf_wide4_inst.hpp"
namespace hipnmf {
const WideKernel<double>* wide4d_kernel_f64_hi(int MP, int KQ, int NW) {
  static const WideKernel<double> t[2][2] = {{make_wide4d_kernel<96, 1, 4>(), make_wide4d_kernel<96, 2, 4>()},
                                             {make_wide4d_kernel<128, 1, 4>(), make_wide4d_kernel<128, 2, 4>()}};
  if ((KQ != 1 && KQ != 2) || NW != 4 || (MP != 96 && MP != 128)) return nullptr;
  return &t[MP == 128][KQ - 1];
}
}  // namespace hipnmf
