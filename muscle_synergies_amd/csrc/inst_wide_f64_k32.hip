// fit_wide_kernel<double, MP, 32, 4>: 17..32 components, MP = 32, 48, 64, 96, 128 (nmf_wide.hpp)
#include "nmf_wide_inst.hpp"
namespace hipnmf {
const WideKernel<double>* wide_kernel_f64_k32(int MP) {
  static const WideKernel<double> tbl[5] = {make_wide_kernel<double, 32, 32, 4>("fit_wide_kernel<double,32,32,4>"),
                                          make_wide_kernel<double, 48, 32, 4>("fit_wide_kernel<double,48,32,4>"),
                                          make_wide_kernel<double, 64, 32, 4>("fit_wide_kernel<double,64,32,4>"),
                                          make_wide_kernel<double, 96, 32, 4>("fit_wide_kernel<double,96,32,4>"),
                                          make_wide_kernel<double, 128, 32, 4>("fit_wide_kernel<double,128,32,4>")};
  return MP == 32 ? &tbl[0] : MP == 48 ? &tbl[1] : MP == 64 ? &tbl[2] : MP == 96 ? &tbl[3] : MP == 128 ? &tbl[4] : nullptr;
}
}  // namespace hipnmf
