// nmf_wide4_inst.hpp -- instantiation helper of fit_wide4_kernel (nmf_wide4.hpp): fp32, n_components <= 8, one translation
// unit per pair of channel paddings
#pragma once
#include <cstdio>

#include "nmf_wide4.hpp"
#include "nmf_wide4d.hpp"
#include "nmf_wide_decl.hpp"

namespace hipnmf {
// two register sets of loads in flight while a set is at most 20 registers (up to 64 channels)
template <int MP, int KQ>
constexpr int wide4_nset() {
  return Wide4Cfg<MP, KQ>::NLD * 4 + KQ <= 20 ? 2 : 1;
}
template <int MP, int KQ, int NW, int NSET>
const char* wide4_kernel_name() {
  static char buf[96];
  static const bool once = [] {
    snprintf(buf, sizeof(buf), "fit_wide4_kernel<%d,%d,%d,%d,0>", MP, KQ, NW, NSET);  // (as rocprofv3 prints it: LOSS = 0)
    return true;
  }();
  (void)once;
  return buf;
}
template <int MP, int KQ, int NW>
WideKernel<float> make_wide4_kernel() {
  WideKernel<float> w;
  constexpr int NSET = wide4_nset<MP, KQ>();
  w.fn = fit_wide4_kernel<MP, KQ, NW, NSET>;
  w.fn_kl = nullptr;
  w.name_kl = "";
  if constexpr (NW == 4 || NW == 8) {  // the Kullback-Leibler flavour: the 256- and 512-thread instances
    w.fn_kl = fit_wide4_kernel<MP, KQ, NW, 1, 1>;
    static char kl_name[96];
    snprintf(kl_name, sizeof(kl_name), "fit_wide4_kernel<%d,%d,%d,1,1>", MP, KQ, NW);
    w.name_kl = kl_name;
  }
  w.smem = Wide4Cfg<MP, KQ>::smem_bytes(NW);
  w.MP = MP;
  w.KP = 4 * KQ;
  w.NW = NW;
  w.name = wide4_kernel_name<MP, KQ, NW, NSET>();
  return w;
}
// the 768-thread instance (three waves per SIMD) exists where the kernel stays within 168 registers: up to 64 channels
template <int MP, int KQ>
WideKernel<float> make_wide4_kernel12() {
  if constexpr (MP <= 64) {
    return make_wide4_kernel<MP, KQ, 12>();
  } else {
    WideKernel<float> w{};
    return w;
  }
}
// the table of one translation unit: two channel paddings x KQ = 1, 2 x {4, 8, 12} waves
template <int MPA, int MPB>
const WideKernel<float>* wide4_table_lookup(int MP, int KQ, int NW) {
  static const WideKernel<float> t[2][2][3] = {
      {{make_wide4_kernel<MPA, 1, 4>(), make_wide4_kernel<MPA, 1, 8>(), make_wide4_kernel12<MPA, 1>()},
       {make_wide4_kernel<MPA, 2, 4>(), make_wide4_kernel<MPA, 2, 8>(), make_wide4_kernel12<MPA, 2>()}},
      {{make_wide4_kernel<MPB, 1, 4>(), make_wide4_kernel<MPB, 1, 8>(), make_wide4_kernel12<MPB, 1>()},
       {make_wide4_kernel<MPB, 2, 4>(), make_wide4_kernel<MPB, 2, 8>(), make_wide4_kernel12<MPB, 2>()}}};
  if ((KQ != 1 && KQ != 2) || (NW != 4 && NW != 8 && NW != 12)) return nullptr;
  const int q = MP == MPA ? 0 : MP == MPB ? 1 : -1;
  if (q < 0) return nullptr;
  const WideKernel<float>* w = &t[q][KQ - 1][NW / 4 - 1];
  return w->fn ? w : nullptr;
}
// float64 (nmf_wide4d.hpp): 48 / 64 channels, KQ = 1, 2, {4, 8} waves
#ifndef HIPNMF_WIDE4D_NSET
#define HIPNMF_WIDE4D_NSET 1
#endif
template <int MP, int KQ, int NW, int NSET, int WPE>
const char* wide4d_kernel_name() {
  static char buf[96];
  static const bool once = [] {
    snprintf(buf, sizeof(buf), "fit_wide4d_kernel<%d,%d,%d,%d,%d,0>", MP, KQ, NW, NSET, WPE);  // (as rocprofv3 prints it: LOSS = 0)
    return true;
  }();
  (void)once;
  return buf;
}
template <int MP, int KQ, int NW>
WideKernel<double> make_wide4d_kernel() {
  WideKernel<double> w;
  constexpr int NSET = HIPNMF_WIDE4D_NSET;
  constexpr int WPE = MP <= 64 ? 2 : 1;
  w.fn = fit_wide4d_kernel<MP, KQ, NW, NSET, WPE>;
  w.fn_kl = nullptr;
  w.name_kl = "";
  if constexpr (NW == 4 || NW == 8) {  // the Kullback-Leibler flavour (round 5)
    w.fn_kl = fit_wide4d_kernel<MP, KQ, NW, 1, WPE, 1>;
    static char kl_name[96];
    snprintf(kl_name, sizeof(kl_name), "fit_wide4d_kernel<%d,%d,%d,1,%d,1>", MP, KQ, NW, WPE);
    w.name_kl = kl_name;
  }
  w.smem = Wide4dCfg<MP, KQ>::smem_bytes(NW);
  w.MP = MP;
  w.KP = 4 * KQ;
  w.NW = NW;
  w.name = wide4d_kernel_name<MP, KQ, NW, NSET, WPE>();
  return w;
}
const WideKernel<double>* wide4d_kernel_f64(int MP, int KQ, int NW);     // MP = 48, 64
const WideKernel<double>* wide4d_kernel_f64_hi(int MP, int KQ, int NW);  // MP = 96, 128: one wave per SIMD, 256 threads
const WideKernel<float>* wide4_kernel_f32_lo(int MP, int KQ, int NW);  // MP = 48, 64
const WideKernel<float>* wide4_kernel_f32_32(int KQ, int NW);          // MP = 32 (17..32 channels)
const WideKernel<float>* wide4_kernel_f32_16(int KQ, int NW);          // MP = 16 (up to 16 channels)
const WideKernel<float>* wide4_kernel_f32_hi(int MP, int KQ, int NW);  // MP = 96, 128
}  // namespace hipnmf
