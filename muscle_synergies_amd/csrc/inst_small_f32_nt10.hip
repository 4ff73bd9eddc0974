// fit_small_kernel<float, 8 / 16, K <= 6, 10>: one wave per matrix, n_samples <= 640 (nmf_small.hpp)
#include "inst_small_long.hpp"
namespace hipnmf {
#define T10(CH) {fit_small_kernel<float, CH, 1, 10>, fit_small_kernel<float, CH, 2, 10>, fit_small_kernel<float, CH, 3, 10>, \
                 fit_small_kernel<float, CH, 4, 10>, fit_small_kernel<float, CH, 5, 10>, fit_small_kernel<float, CH, 6, 10>}
SmallFn<float> small_f32_nt10(int CH, int K) {
  static const SmallFn<float> t8[6] = T10(8);
  static const SmallFn<float> t16[6] = T10(16);
  if (K < 1 || K > 6) return nullptr;
  return CH == 8 ? t8[K - 1] : t16[K - 1];
}
}  // namespace hipnmf
