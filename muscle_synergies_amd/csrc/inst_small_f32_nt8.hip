// fit_small_kernel<float, 8 / 16, K, 8>: one wave per matrix, n_samples <= 512 (nmf_small.hpp)
#include "inst_small_long.hpp"
namespace hipnmf {
#define T8(CH) {fit_small_kernel<float, CH, 1, 8>, fit_small_kernel<float, CH, 2, 8>, fit_small_kernel<float, CH, 3, 8>, fit_small_kernel<float, CH, 4, 8>, \
                fit_small_kernel<float, CH, 5, 8>, fit_small_kernel<float, CH, 6, 8>, fit_small_kernel<float, CH, 7, 8>, fit_small_kernel<float, CH, 8, 8>}
SmallFn<float> small_f32_nt8(int CH, int K) {
  static const SmallFn<float> t8[8] = T8(8);
  static const SmallFn<float> t16[8] = T8(16);
  if (K < 1 || K > 8) return nullptr;
  return CH == 8 ? t8[K - 1] : t16[K - 1];
}
}  // namespace hipnmf
