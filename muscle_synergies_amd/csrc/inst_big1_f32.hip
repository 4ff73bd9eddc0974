// big1_pass_kernel<float, KP, NQ, RS, HL, NST, LOSS> (nmf_big1.hpp): the one-pass update + record kernel of the general shapes,
// fp32.  KP = 16 / 32 / 48 / 64 padded components; NQ = 1 / 2 / 4: eight waves x 16 NQ channels cover 128 / 256 / 512 channels;
// RS subtiles of 16 rows per round (the X block of a round sits in 4 NQ RS registers per lane); HL: the wave's block of H in LDS.
// Kullback-Leibler instances (LOSS = 1) exist where both operand layouts of H fit LDS beside the exchange areas: everything but
// 48 / 64 components on more than 256 channels (those keep the two-pass pair of nmf_big.hpp).
#include "nmf_big1.hpp"
namespace hipnmf {
namespace {
template <int KP, int NQ, int RS, bool HL = false, int NST = 2>
Big1Kernel<float> make_big1(const char* name) {
  return Big1Kernel<float>{big1_pass_kernel<float, KP, NQ, RS, HL, NST>, Big1Cfg<float, KP, NQ, RS, HL, NST>::smem_bytes(), KP, NQ, RS, name,
                           big1_resid_kernel<float, KP, NQ>, nullptr, 0, ""};
}
template <int KP, int NQ, int RS, int NST = 2>
void add_kl(Big1Kernel<float>& k, const char* name) {
  k.fn_kl = big1_pass_kernel<float, KP, NQ, RS, true, NST, 1>;
  k.smem_kl = Big1Cfg<float, KP, NQ, RS, true, NST, 1>::smem_bytes();
  k.name_kl = name;
}
}  // namespace
const Big1Kernel<float>* big1_kernel_f32(int KP, int MP) {
  static Big1Kernel<float> t[4][3] = {
      {make_big1<16, 1, 4>("big1_pass_kernel<float,16,1,4,false,2,0>"), make_big1<16, 2, 4>("big1_pass_kernel<float,16,2,4,false,2,0>"),
       make_big1<16, 4, 4>("big1_pass_kernel<float,16,4,4,false,2,0>")},
      {make_big1<32, 1, 4>("big1_pass_kernel<float,32,1,4,false,2,0>"), make_big1<32, 2, 4>("big1_pass_kernel<float,32,2,4,false,2,0>"),
       make_big1<32, 4, 4, true>("big1_pass_kernel<float,32,4,4,true,2,0>")},  // (H in registers fits since the loop carries no branch: 250 VGPRs, same rate)
      {make_big1<48, 1, 4>("big1_pass_kernel<float,48,1,4,false,2,0>"), make_big1<48, 2, 4>("big1_pass_kernel<float,48,2,4,false,2,0>"),
       make_big1<48, 4, 2, true, 1>("big1_pass_kernel<float,48,4,2,true,1,0>")},
      {make_big1<64, 1, 4>("big1_pass_kernel<float,64,1,4,false,2,0>"), make_big1<64, 2, 2>("big1_pass_kernel<float,64,2,2,false,2,0>"),
       make_big1<64, 4, 2>("big1_pass_kernel<float,64,4,2,false,2,0>")}};
  static const bool once = [] {
    add_kl<16, 1, 4>(t[0][0], "big1_pass_kernel<float,16,1,4,true,2,1>");
    add_kl<16, 2, 4>(t[0][1], "big1_pass_kernel<float,16,2,4,true,2,1>");
    add_kl<16, 4, 4>(t[0][2], "big1_pass_kernel<float,16,4,4,true,2,1>");
    add_kl<32, 1, 4>(t[1][0], "big1_pass_kernel<float,32,1,4,true,2,1>");
    add_kl<32, 2, 4>(t[1][1], "big1_pass_kernel<float,32,2,4,true,2,1>");
    add_kl<32, 4, 4>(t[1][2], "big1_pass_kernel<float,32,4,4,true,2,1>");
    add_kl<48, 1, 2>(t[2][0], "big1_pass_kernel<float,48,1,2,true,2,1>");
    add_kl<48, 2, 2>(t[2][1], "big1_pass_kernel<float,48,2,2,true,2,1>");
    add_kl<64, 1, 2>(t[3][0], "big1_pass_kernel<float,64,1,2,true,2,1>");
    add_kl<64, 2, 2>(t[3][1], "big1_pass_kernel<float,64,2,2,true,2,1>");
    return true;
  }();
  (void)once;
  if (KP < 16 || KP > 64 || KP % 16 || MP > 512) return nullptr;
  return &t[KP / 16 - 1][MP <= 128 ? 0 : MP <= 256 ? 1 : 2];
}
}  // namespace hipnmf
