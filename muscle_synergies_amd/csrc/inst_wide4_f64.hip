// fit_wide4d_kernel<MP, KQ, NW>, MP = 16, 32, 48, 64 (nmf_wide4d.hpp)
#include "nmf_wide4_inst.hpp"
namespace hipnmf {
const WideKernel<double>* wide4d_kernel_f64(int MP, int KQ, int NW) {
  static const WideKernel<double> t[4][2][2] = {
      {{make_wide4d_kernel<16, 1, 4>(), make_wide4d_kernel<16, 1, 8>()}, {make_wide4d_kernel<16, 2, 4>(), make_wide4d_kernel<16, 2, 8>()}},
      {{make_wide4d_kernel<32, 1, 4>(), make_wide4d_kernel<32, 1, 8>()}, {make_wide4d_kernel<32, 2, 4>(), make_wide4d_kernel<32, 2, 8>()}},
      {{make_wide4d_kernel<48, 1, 4>(), make_wide4d_kernel<48, 1, 8>()}, {make_wide4d_kernel<48, 2, 4>(), make_wide4d_kernel<48, 2, 8>()}},
      {{make_wide4d_kernel<64, 1, 4>(), make_wide4d_kernel<64, 1, 8>()}, {make_wide4d_kernel<64, 2, 4>(), make_wide4d_kernel<64, 2, 8>()}}};
  // (12 / 16 waves, which 16 / 32 channels would fit -- 105..177 registers -- were measured: 8192 x (16 x 600), k = 5: 38.7 (8 waves) / 36.7
  //  (12); k = 4: 77.4 / 68.4 (16); 32 x 600, k = 3: 47.3 / 43.6 (16) M matrix-it/s: not compiled)
  if ((KQ != 1 && KQ != 2) || (NW != 4 && NW != 8) || (MP != 16 && MP != 32 && MP != 48 && MP != 64)) return nullptr;
  return &t[MP == 16 ? 0 : MP == 32 ? 1 : MP == 48 ? 2 : 3][KQ - 1][NW == 8];
}
}  // namespace hipnmf
