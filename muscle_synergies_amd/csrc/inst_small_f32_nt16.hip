// fit_small_kernel<float, 8, K <= 5, 16> / <float, 16, K <= 3, 16>: one wave per matrix, n_samples <= 1024 (nmf_small.hpp)
#include "inst_small_long.hpp"
namespace hipnmf {
SmallFn<float> small_f32_nt16(int CH, int K) {
  static const SmallFn<float> t8[5] = {fit_small_kernel<float, 8, 1, 16>, fit_small_kernel<float, 8, 2, 16>, fit_small_kernel<float, 8, 3, 16>,
                                       fit_small_kernel<float, 8, 4, 16>, fit_small_kernel<float, 8, 5, 16>};
  static const SmallFn<float> t16[3] = {fit_small_kernel<float, 16, 1, 16>, fit_small_kernel<float, 16, 2, 16>, fit_small_kernel<float, 16, 3, 16>};
  if (K < 1) return nullptr;
  if (CH == 8) return K <= 5 ? t8[K - 1] : nullptr;
  return K <= 3 ? t16[K - 1] : nullptr;
}
}  // namespace hipnmf
