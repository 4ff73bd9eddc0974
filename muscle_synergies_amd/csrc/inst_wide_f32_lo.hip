// fit_wide_kernel<float, MP, 16, NW>, MP = 16, 32, 48, NW = 4 / 8 (nmf_wide.hpp)
#include "nmf_wide_inst.hpp"
namespace hipnmf {
const WideKernel<float>* wide_kernel_f32_lo(int MP, int KP, int NW) {
  static const char* const names[6] = {"fit_wide_kernel<float,16,16,4>", "fit_wide_kernel<float,32,16,4>", "fit_wide_kernel<float,48,16,4>",
                                       "fit_wide_kernel<float,16,16,8>", "fit_wide_kernel<float,32,16,8>", "fit_wide_kernel<float,48,16,8>"};
  return wide_table_lookup<float, 16, 32, 48>(MP, KP, NW, names);
}
}  // namespace hipnmf
