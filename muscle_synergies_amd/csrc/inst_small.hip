// fit_small_kernel<real, CH, K>: one wave per short matrix (nmf_small.hpp); fp32 with 8 / 16 channels, fp64 with 8
#include "inst_small_long.hpp"
namespace hipnmf {
#define SMALL_TABLE(REAL, CH)                                                                                        \
  {fit_small_kernel<REAL, CH, 1>, fit_small_kernel<REAL, CH, 2>, fit_small_kernel<REAL, CH, 3>,                     \
   fit_small_kernel<REAL, CH, 4>, fit_small_kernel<REAL, CH, 5>, fit_small_kernel<REAL, CH, 6>,                     \
   fit_small_kernel<REAL, CH, 7>, fit_small_kernel<REAL, CH, 8>}
template <>
SmallFn<float> small_kernel<float>(int m, int K) {
  static const SmallFn<float> t8[8] = SMALL_TABLE(float, 8);
  static const SmallFn<float> t16[8] = SMALL_TABLE(float, 16);
  if (K < 1 || K > 8 || m < 1 || m > 16) return nullptr;
  return m <= 8 ? t8[K - 1] : t16[K - 1];
}
template <>
SmallFn<double> small_kernel<double>(int m, int K) {
  static const SmallFn<double> t8[8] = SMALL_TABLE(double, 8);
  // 9..16 channels in float64 (round 3), up to 6 components.  The k = 5 instance walks its W update in groups of two tiles
  // instead of four (nmf_small.hpp: hipcc's split-spill defect, profiles/r04_small_f64_miscompile.md); the build lints every
  // kernel's ISA for that defect and tests/test_gpu_small_long.py checks EVERY compiled (dtype, channels, k, tiles) instance
  // against the oracle.
  static const SmallFn<double> t16[6] = {fit_small_kernel<double, 16, 1>, fit_small_kernel<double, 16, 2>, fit_small_kernel<double, 16, 3>,
                                         fit_small_kernel<double, 16, 4>, fit_small_kernel<double, 16, 5>, fit_small_kernel<double, 16, 6>};
  if (K < 1 || K > 8 || m < 1 || m > 16) return nullptr;
  if (m > 8) return K <= 6 ? t16[K - 1] : nullptr;
  return t8[K - 1];
}
// the smallest NT in {6, 8, 10, 12, 16} that holds n_samples rows and is compiled for the shape (inst_small_long.hpp)
template <>
SmallFn<float> small_kernel_long<float>(int m, int K, long long T, int* nt_out) {
  if (K < 1 || K > 8 || m < 1 || m > 16 || T > 1024) return nullptr;
  const int CH = m <= 8 ? 8 : 16;
  static const int nts[5] = {6, 8, 10, 12, 16};
  for (int nt : nts) {
    if (T > 64LL * nt) continue;
    SmallFn<float> f = nt == 6 ? small_f32_nt6(CH, K) : nt == 8 ? small_f32_nt8(CH, K) : nt == 10 ? small_f32_nt10(CH, K)
                     : nt == 12 ? small_f32_nt12(CH, K) : small_f32_nt16(CH, K);
    if (f) {
      *nt_out = nt;
      return f;
    }
  }
  return nullptr;
}
template <>
SmallFn<double> small_kernel_long<double>(int m, int K, long long T, int* nt_out) {
  if (K < 1 || K > 6 || m < 1 || m > 8 || T > 768) return nullptr;
  static const int nts[3] = {6, 8, 12};
  for (int nt : nts) {
    if (T > 64LL * nt) continue;
    SmallFn<double> f = nt == 6 ? small_f64_nt6(K) : nt == 8 ? small_f64_nt8(K) : small_f64_nt12(K);
    if (f) {
      *nt_out = nt;
      return f;
    }
  }
  return nullptr;
}
template <typename real>
static size_t smem_of(int m, int K) {
  const int CH = m <= 8 ? 8 : 16;
  const int nacc = K * CH + K * (K + 1) / 2;
  const int nrec = nacc > 2 * CH + 1 ? nacc : 2 * CH + 1;
  return sizeof(real) * (size_t)(2 * K * CH + 2 * K * K + nrec + 8);  // = Smem<real, 1, CH, K>::bytes(1)
}
template <>
size_t small_smem_bytes<float>(int m, int K) { return smem_of<float>(m, K); }
template <>
size_t small_smem_bytes<double>(int m, int K) { return smem_of<double>(m, K); }
}  // namespace hipnmf
