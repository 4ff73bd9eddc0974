// fit_wide4_kernel<32 / 16, KQ, NW>: up to 32 channels (nmf_wide4.hpp)
#include "nmf_wide4_inst.hpp"
namespace hipnmf {
const WideKernel<float>* wide4_kernel_f32_32(int KQ, int NW) {
  static const WideKernel<float> t[2][3] = {{make_wide4_kernel<32, 1, 4>(), make_wide4_kernel<32, 1, 8>(), make_wide4_kernel12<32, 1>()},
                                            {make_wide4_kernel<32, 2, 4>(), make_wide4_kernel<32, 2, 8>(), make_wide4_kernel12<32, 2>()}};
  if ((KQ != 1 && KQ != 2) || (NW != 4 && NW != 8 && NW != 12)) return nullptr;
  return &t[KQ - 1][NW / 4 - 1];
}
const WideKernel<float>* wide4_kernel_f32_16(int KQ, int NW) {
  static const WideKernel<float> t[2][3] = {{make_wide4_kernel<16, 1, 4>(), make_wide4_kernel<16, 1, 8>(), make_wide4_kernel12<16, 1>()},
                                            {make_wide4_kernel<16, 2, 4>(), make_wide4_kernel<16, 2, 8>(), make_wide4_kernel12<16, 2>()}};
  if ((KQ != 1 && KQ != 2) || (NW != 4 && NW != 8 && NW != 12)) return nullptr;
  return &t[KQ - 1][NW / 4 - 1];
}
}  // namespace hipnmf
