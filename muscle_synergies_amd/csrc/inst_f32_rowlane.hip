// fit_rowlane_kernel<K>, K = 1..8: the fp32 / 9..16 channel batch kernel (nmf_rowlane.hpp)
#include "nmf_rowlane.hpp"

#include <cstdio>
namespace hipnmf {
RowLaneFn rowlane_kernel(int K) {
  static const RowLaneFn tbl[8] = {fit_rowlane_kernel<1>, fit_rowlane_kernel<2>, fit_rowlane_kernel<3>,
                                   fit_rowlane_kernel<4>, fit_rowlane_kernel<5>, fit_rowlane_kernel<6>,
                                   fit_rowlane_kernel<7>, fit_rowlane_kernel<8>};
  return (K >= 1 && K <= 8) ? tbl[K - 1] : nullptr;
}
RowLaneFn rowlane_kernel_kl(int K) {  // Kullback-Leibler flavour (both reconstructions on the matrix pipe)
  static const RowLaneFn tbl[8] = {
      fit_rowlane_kernel<1, 0, 0, 1, 1>, fit_rowlane_kernel<2, 0, 0, 1, 1>, fit_rowlane_kernel<3, 0, 0, 1, 1>,
      fit_rowlane_kernel<4, 0, 0, 1, 1>, fit_rowlane_kernel<5, 0, 0, 1, 1>, fit_rowlane_kernel<6, 0, 0, 1, 1>,
      fit_rowlane_kernel<7, 0, 0, 1, 1>, fit_rowlane_kernel<8, 0, 0, 1, 1>};
  return (K >= 1 && K <= 8) ? tbl[K - 1] : nullptr;
}
RowLaneFn slice_pass_rowlane(int K) {
  static const RowLaneFn tbl[8] = {slice_pass_rowlane_kernel<1>, slice_pass_rowlane_kernel<2>, slice_pass_rowlane_kernel<3>,
                                   slice_pass_rowlane_kernel<4>, slice_pass_rowlane_kernel<5>, slice_pass_rowlane_kernel<6>,
                                   slice_pass_rowlane_kernel<7>, slice_pass_rowlane_kernel<8>};
  return (K >= 1 && K <= 8) ? tbl[K - 1] : nullptr;
}
template <int K>
static void fill_name(char (&buf)[64]) {
  snprintf(buf, sizeof(buf), "fit_rowlane_kernel<%d,%d,%d,%d>", K, rl_nxr<K>(), rl_nwr<K>(), (int)(HIPNMF_RL_PF));
}
const char* rowlane_kernel_name(int K) {
  static char names[8][64];
  static const bool init = [] {
    fill_name<1>(names[0]); fill_name<2>(names[1]); fill_name<3>(names[2]); fill_name<4>(names[3]);
    fill_name<5>(names[4]); fill_name<6>(names[5]); fill_name<7>(names[6]); fill_name<8>(names[7]);
    return true;
  }();
  (void)init;
  return (K >= 1 && K <= 8) ? names[K - 1] : "";
}
}  // namespace hipnmf
