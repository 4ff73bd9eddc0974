// fit_wide_kernel<double, MP, 16, 4>, MP = 64, 96, 128 (nmf_wide.hpp)
#include "nmf_wide_inst.hpp"
namespace hipnmf {
const WideKernel<double>* wide_kernel_f64_hi(int MP, int KP) {
  static const WideKernel<double> tbl[3] = {make_wide_kernel<double, 64, 16, 4>("fit_wide_kernel<double,64,16,4>"),
                                          make_wide_kernel<double, 96, 16, 4>("fit_wide_kernel<double,96,16,4>"),
                                          make_wide_kernel<double, 128, 16, 4>("fit_wide_kernel<double,128,16,4>")};
  if (KP != 16) return nullptr;
  return MP == 64 ? &tbl[0] : MP == 96 ? &tbl[1] : MP == 128 ? &tbl[2] : nullptr;
}
}  // namespace hipnmf
