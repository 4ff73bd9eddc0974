// fit_small_kernel<double, 8, K <= 6, 8> / <double, 8, K <= 3, 12>: one wave per matrix, n_samples <= 512 / 768 (nmf_small.hpp)
#include "inst_small_long.hpp"
namespace hipnmf {
SmallFn<double> small_f64_nt8(int K) {
  static const SmallFn<double> t[6] = {fit_small_kernel<double, 8, 1, 8>, fit_small_kernel<double, 8, 2, 8>, fit_small_kernel<double, 8, 3, 8>,
                                       fit_small_kernel<double, 8, 4, 8>, fit_small_kernel<double, 8, 5, 8>, fit_small_kernel<double, 8, 6, 8>};
  return (K < 1 || K > 6) ? nullptr : t[K - 1];
}
SmallFn<double> small_f64_nt12(int K) {
  static const SmallFn<double> t[3] = {fit_small_kernel<double, 8, 1, 12>, fit_small_kernel<double, 8, 2, 12>, fit_small_kernel<double, 8, 3, 12>};
  return (K < 1 || K > 3) ? nullptr : t[K - 1];
}
}  // namespace hipnmf
