// fit_wide_kernel<float, MP, 16, 4>, MP = 16, 32, 48 (nmf_wide.hpp)
#include "nmf_wide_inst.hpp"
namespace hipnmf {
const WideKernel<float>* wide_kernel_f32_lo(int MP, int KP) {
  static const WideKernel<float> tbl[3] = {make_wide_kernel<float, 16, 16, 4>("fit_wide_kernel<float,16,16,4>"),
                                          make_wide_kernel<float, 32, 16, 4>("fit_wide_kernel<float,32,16,4>"),
                                          make_wide_kernel<float, 48, 16, 4>("fit_wide_kernel<float,48,16,4>")};
  if (KP != 16) return nullptr;
  return MP == 16 ? &tbl[0] : MP == 32 ? &tbl[1] : MP == 48 ? &tbl[2] : nullptr;
}
}  // namespace hipnmf
