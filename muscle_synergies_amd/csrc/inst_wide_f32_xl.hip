// fit_wide_kernel<float, MP, 16, 4>, MP = 160, 192, 256: HD-EMG grids of up to 256 channels with at most 16 components keep the
// one-pass formulation (X read once per iteration; the accumulators of [W^T X | W^T W] still fit a wave: 16 x 256 / 64 = 64
// registers) instead of the two passes of nmf_big.hpp
#include "nmf_wide_inst.hpp"
namespace hipnmf {
const WideKernel<float>* wide_kernel_f32_xl(int MP) {
  static const WideKernel<float> t[3] = {make_wide_kernel<float, 160, 16, 4>(), make_wide_kernel<float, 192, 16, 4>(),
                                         make_wide_kernel<float, 256, 16, 4>()};
  return MP == 160 ? &t[0] : MP == 192 ? &t[1] : MP == 256 ? &t[2] : nullptr;
}
}  // namespace hipnmf
