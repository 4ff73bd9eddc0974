// nmf_wide_inst.hpp -- instantiation helper of the wide-shape kernels (one translation unit per dtype and half of the
// channel paddings, so that the build compiles them in parallel)
#pragma once
#include <cstdio>

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
  if (KP == 32 && sizeof(real) == 8) return 1;  // (32 channels would spill 6 registers at two waves per SIMD)
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
// name as rocprofv3 --kernel-trace prints the instance (minus namespace and argument list): what hipnmf_last_kernel reports
// and what profiles/traffic.json is keyed on
template <typename real, int MP, int KP, int NW, bool HREG, int WPE, int NSET, int LOSS>
const char* wide_kernel_name() {
  static char buf[96];
  static const bool once = [] {
    snprintf(buf, sizeof(buf), "fit_wide_kernel<%s,%d,%d,%d,%s,%d,%d,%d>", sizeof(real) == 4 ? "float" : "double", MP, KP, NW,
             HREG ? "true" : "false", WPE, NSET, LOSS);
    return true;
  }();
  (void)once;
  return buf;
}
template <typename real, int MP, int KP, int NW>
WideKernel<real> make_wide_kernel(const char* = nullptr) {
  WideKernel<real> w;
  constexpr bool HREG = wide_hreg<real, MP, KP>();
  constexpr int WPE = wide_wpe<real, MP, KP>(), NSET = wide_nset<real, MP, KP>();
  w.fn = fit_wide_kernel<real, MP, KP, NW, HREG, WPE, NSET>;
  w.fn_kl = nullptr;
  w.name_kl = "";
  if constexpr (NW == 4) {  // (H is re-read from LDS in the KL flavour: no register copy, one set of loads in flight)
    w.fn_kl = fit_wide_kernel<real, MP, KP, NW, false, WPE, 1, 1>;
    w.name_kl = wide_kernel_name<real, MP, KP, NW, false, WPE, 1, 1>();
  }
  w.smem = WideCfg<real, MP, KP>::smem_bytes(NW);
  w.MP = MP;
  w.KP = KP;
  w.NW = NW;
  w.name = wide_kernel_name<real, MP, KP, NW, HREG, WPE, NSET, 0>();
  return w;
}
const WideKernel<float>* wide_kernel_f32_lo(int MP, int KP, int NW);
const WideKernel<float>* wide_kernel_f32_hi(int MP, int KP, int NW);
const WideKernel<double>* wide_kernel_f64_lo(int MP, int KP, int NW);
const WideKernel<double>* wide_kernel_f64_hi(int MP, int KP, int NW);
// 17..32 components (KP = 32): 256-thread instances for 32, 48, 64, 96, 128 channels
const WideKernel<float>* wide_kernel_f32_k32(int MP);
// 129..256 channels, at most 16 components, fp32 (MP = 160, 192, 256): inst_wide_f32_xl.hip
const WideKernel<float>* wide_kernel_f32_xl(int MP);
const WideKernel<double>* wide_kernel_f64_k32(int MP);
// the 512-thread instance exists only where the kernel is compiled for two waves per SIMD (256 registers)
template <typename real, int MP>
WideKernel<real> make_wide_kernel8(const char* name) {
  if constexpr (wide_wpe<real, MP, 16>() == 2) {
    return make_wide_kernel<real, MP, 16, 8>(name);
  } else {
    WideKernel<real> w{};
    return w;
  }
}
// the table of one translation unit: three channel paddings x {4, 8} waves (8 only for the two-waves-per-SIMD instances)
template <typename real, int MPA, int MPB, int MPC>
const WideKernel<real>* wide_table_lookup(int MP, int KP, int NW, const char* const (&names)[6]) {
  static const WideKernel<real> t4[3] = {make_wide_kernel<real, MPA, 16, 4>(names[0]), make_wide_kernel<real, MPB, 16, 4>(names[1]),
                                         make_wide_kernel<real, MPC, 16, 4>(names[2])};
  static const WideKernel<real> t8[3] = {make_wide_kernel8<real, MPA>(names[3]), make_wide_kernel8<real, MPB>(names[4]),
                                         make_wide_kernel8<real, MPC>(names[5])};
  static const bool ok8[3] = {wide_wpe<real, MPA, 16>() == 2, wide_wpe<real, MPB, 16>() == 2, wide_wpe<real, MPC, 16>() == 2};
  if (KP != 16 || (NW != 4 && NW != 8)) return nullptr;
  const int q = MP == MPA ? 0 : MP == MPB ? 1 : MP == MPC ? 2 : -1;
  if (q < 0) return nullptr;
  if (NW == 8) return ok8[q] ? &t8[q] : nullptr;
  return &t4[q];
}
}  // namespace hipnmf
