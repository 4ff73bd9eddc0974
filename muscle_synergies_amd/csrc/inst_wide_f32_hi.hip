// fit_wide_kernel<float, MP, 16, NW>, MP = 64, 96, 128, NW = 4 / 8 (nmf_wide.hpp)
#include "nmf_wide_inst.hpp"
namespace hipnmf {
const WideKernel<float>* wide_kernel_f32_hi(int MP, int KP, int NW) {
  static const char* const names[6] = {"fit_wide_kernel<float,64,16,4>", "fit_wide_kernel<float,96,16,4>", "fit_wide_kernel<float,128,16,4>",
                                       "fit_wide_kernel<float,64,16,8>", "fit_wide_kernel<float,96,16,8>", "fit_wide_kernel<float,128,16,8>"};
  return wide_table_lookup<float, 64, 96, 128>(MP, KP, NW, names);
}
}  // namespace hipnmf
