// big1_pass_kernel<double, KP, NQ, RS, HL, NST, LOSS> (nmf_big1.hpp): the one-pass general-shape kernel in float64 -- what a
// DataFrame carries.  Registers and LDS hold half as many values as in fp32, so the instances stop at 256 channels (eight waves x
// 32 channels, NQ = 2): beyond, and where the Kullback-Leibler flavour's two operand layouts of H do not fit, float64 keeps the
// two-pass pair of nmf_big.hpp.  H always lives in LDS here (HL): KP x CW doubles would be 2 KP CW / 64 registers per lane.
#include "nmf_big1.hpp"
namespace hipnmf {
namespace {
template <int KP, int NQ, int RS, int NST = 2>
Big1Kernel<double> make_big1d(const char* name) {
  return Big1Kernel<double>{big1_pass_kernel<double, KP, NQ, RS, true, NST>, Big1Cfg<double, KP, NQ, RS, true, NST>::smem_bytes(), KP, NQ, RS, name,
                            big1_resid_kernel<double, KP, NQ>, nullptr, 0, ""};
}
template <int KP, int NQ, int RS, int NST = 2>
void add_kl(Big1Kernel<double>& k, const char* name) {
  k.fn_kl = big1_pass_kernel<double, KP, NQ, RS, true, NST, 1>;
  k.smem_kl = Big1Cfg<double, KP, NQ, RS, true, NST, 1>::smem_bytes();
  k.name_kl = name;
}
}  // namespace
const Big1Kernel<double>* big1_kernel_f64(int KP, int MP) {
  // 16 / 32 padded components; 48 / 64 do not fit (the eight waves' partial numerators alone are 98 / 131 KB in float64)
  static Big1Kernel<double> t[2][2] = {
      {make_big1d<16, 1, 4>("big1_pass_kernel<double,16,1,4,true,2,0>"), make_big1d<16, 2, 4>("big1_pass_kernel<double,16,2,4,true,2,0>")},
      {make_big1d<32, 1, 2>("big1_pass_kernel<double,32,1,2,true,2,0>"), make_big1d<32, 2, 2>("big1_pass_kernel<double,32,2,2,true,2,0>")}};
  static const bool once = [] {
    add_kl<16, 1, 4>(t[0][0], "big1_pass_kernel<double,16,1,4,true,2,1>");
    add_kl<16, 2, 4>(t[0][1], "big1_pass_kernel<double,16,2,4,true,2,1>");
    add_kl<32, 1, 2>(t[1][0], "big1_pass_kernel<double,32,1,2,true,2,1>");
    return true;
  }();
  (void)once;
  if ((KP != 16 && KP != 32) || MP > 256) return nullptr;
  return &t[KP / 16 - 1][MP <= 128 ? 0 : 1];
}
}  // namespace hipnmf
