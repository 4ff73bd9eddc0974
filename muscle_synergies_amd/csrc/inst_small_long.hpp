// inst_small_long.hpp -- tables of fit_small_kernel<real, CH, K, NT> for NT > 4 (one translation unit per NT and dtype).
// Which (CH, K, NT) exist is what measured faster than a workgroup per matrix AND than the 4x4 matrix-pipe kernels, 16 384
// matrices, M matrix-it/s (workgroup / this kernel / fit_wide4[d]_kernel):
//   fp32 16 ch: NT = 8 (n_samples <= 512): k = 5 T = 300 56 / 157 / 63, T = 500 54 / 155 / 54; k = 8 T = 500 46 / 90 / 53; 12 ch k = 7 48 / 107 / 54
//               NT = 12 (<= 768): k = 5 T = 700 49 / 100 / 48; k = 6 51 / 68 / 47; k = 8 40 / 28 / 46 (the registers run out)      => k <= 6
//               NT = 16 (<= 1 024): k = 2 T = 1 000 92 / 144 / 78; k = 3 76 / 81 / 76; k = 5 44 / 23 / 40                          => k <= 3
//   fp32  8 ch: NT = 8: k = 4 T = 500 99 / 454 / 135; NT = 12: k = 4 T = 700 92 / 209 / 111; NT = 16: k = 4 T = 1 000 84 / 165 / 83 => k <= 5
//   fp64  8 ch: NT = 8: k = 4 T = 500 65 / 189 / 113; k = 6 33 / 54 / 47; 5 ch k = 3 T = 400 61 / 274 / 137;
//               NT = 12: k = 2 T = 700 98 / 269 / 91; k = 4 59 / 57 / 87                                                           => k <= 3
//   6 and 10 tiles (n_samples <= 384 / 640) cut the padding of the sizes in between: 16 ch k = 5 T = 300 157 (8 tiles) -> 197 (6 tiles),
//   T = 600 ~117 (12 tiles) -> 128 (10 tiles); 8 ch k = 4 T = 350 548; fp64 8 ch k = 4 T = 300 243
#pragma once
#include "nmf_small.hpp"
namespace hipnmf {
SmallFn<float> small_f32_nt6(int CH, int K);    // K <= 8
SmallFn<float> small_f32_nt10(int CH, int K);   // K <= 6
SmallFn<double> small_f64_nt6(int K);           // CH = 8, K <= 6
SmallFn<float> small_f32_nt8(int CH, int K);    // K <= 8
SmallFn<float> small_f32_nt12(int CH, int K);   // K <= 6
SmallFn<float> small_f32_nt16(int CH, int K);   // CH = 16: K <= 3, CH = 8: K <= 5
SmallFn<double> small_f64_nt8(int K);           // CH = 8, K <= 6
SmallFn<double> small_f64_nt12(int K);          // CH = 8, K <= 3
}  // namespace hipnmf
