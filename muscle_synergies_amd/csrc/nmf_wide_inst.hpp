// nmf_wide_inst.hpp -- instantiation helper of the wide-shape kernels (one translation unit per dtype and half of the
// channel paddings, so that the build compiles them in parallel)
#pragma once
#include "nmf_wide_decl.hpp"

namespace hipnmf {
// H operands of H X^T in registers when they take at most 16 VGPRs, re-read from LDS per subtile otherwise
template <typename real, int MP, int KP>
constexpr bool wide_hreg() {
  return (KP / 16) * (MP / 4) * (int)(sizeof(real) / 4) <= 16;
}
// two waves per SIMD wherever the live state fits 256 registers (fp32 up to 128 channels, fp64 up to 48)
template <typename real, int MP, int KP>
constexpr int wide_wpe() {
  return (KP / 16) * MP * (int)(sizeof(real) / 4) <= 128 ? 2 : 1;
}
// register sets of loads in flight per wave: two while a set is at most 20 registers
#ifndef HIPNMF_WIDE_NSET
#define HIPNMF_WIDE_NSET 0
#endif
template <typename real, int MP, int KP>
constexpr int wide_nset() {
  if (HIPNMF_WIDE_NSET) return HIPNMF_WIDE_NSET;
  return (MP / 4 + 4 * (KP / 16)) * (int)(sizeof(real) / 4) <= 20 ? 2 : 1;
}
template <typename real, int MP, int KP, int NW>
WideKernel<real> make_wide_kernel(const char* name) {
  WideKernel<real> w;
  w.fn = fit_wide_kernel<real, MP, KP, NW, wide_hreg<real, MP, KP>(), wide_wpe<real, MP, KP>(), wide_nset<real, MP, KP>()>;
  w.smem = WideCfg<real, MP, KP>::smem_bytes(NW);
  w.MP = MP;
  w.KP = KP;
  w.NW = NW;
  w.name = name;
  return w;
}
const WideKernel<float>* wide_kernel_f32_lo(int MP, int KP);
const WideKernel<float>* wide_kernel_f32_hi(int MP, int KP);
const WideKernel<double>* wide_kernel_f64_lo(int MP, int KP);
const WideKernel<double>* wide_kernel_f64_hi(int MP, int KP);
}  // namespace hipnmf
