// fit_small_kernel<float, 8 / 16, K, 6> and <double, 8, K <= 6, 6>: one wave per matrix, n_samples <= 384 (nmf_small.hpp)
#include "inst_small_long.hpp"
namespace hipnmf {
#define T6(CH) {fit_small_kernel<float, CH, 1, 6>, fit_small_kernel<float, CH, 2, 6>, fit_small_kernel<float, CH, 3, 6>, fit_small_kernel<float, CH, 4, 6>, \
                fit_small_kernel<float, CH, 5, 6>, fit_small_kernel<float, CH, 6, 6>, fit_small_kernel<float, CH, 7, 6>, fit_small_kernel<float, CH, 8, 6>}
SmallFn<float> small_f32_nt6(int CH, int K) {
  static const SmallFn<float> t8[8] = T6(8);
  static const SmallFn<float> t16[8] = T6(16);
  if (K < 1 || K > 8) return nullptr;
  return CH == 8 ? t8[K - 1] : t16[K - 1];
}
SmallFn<double> small_f64_nt6(int K) {
  static const SmallFn<double> t[6] = {fit_small_kernel<double, 8, 1, 6>, fit_small_kernel<double, 8, 2, 6>, fit_small_kernel<double, 8, 3, 6>,
                                       fit_small_kernel<double, 8, 4, 6>, fit_small_kernel<double, 8, 5, 6>, fit_small_kernel<double, 8, 6, 6>};
  return (K < 1 || K > 6) ? nullptr : t[K - 1];
}
}  // namespace hipnmf
