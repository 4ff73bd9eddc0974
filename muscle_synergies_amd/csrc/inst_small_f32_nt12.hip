// fit_small_kernel<float, 8 / 16, K <= 6, 12>: one wave per matrix, n_samples <= 768 (nmf_small.hpp)
#include "inst_small_long.hpp"
namespace hipnmf {
#define T12(CH) {fit_small_kernel<float, CH, 1, 12>, fit_small_kernel<float, CH, 2, 12>, fit_small_kernel<float, CH, 3, 12>, \
                 fit_small_kernel<float, CH, 4, 12>, fit_small_kernel<float, CH, 5, 12>, fit_small_kernel<float, CH, 6, 12>}
SmallFn<float> small_f32_nt12(int CH, int K) {
  static const SmallFn<float> t8[6] = T12(8);
  static const SmallFn<float> t16[6] = T12(16);
  if (K < 1 || K > 6) return nullptr;
  return CH == 8 ? t8[K - 1] : t16[K - 1];
}
}  // namespace hipnmf
