// fit_wide4_kernel<MP, KQ, NW>, MP = 48, 64 (nmf_wide4.hpp)
#include "nmf_wide4_inst.hpp"
namespace hipnmf {
const WideKernel<float>* wide4_kernel_f32_lo(int MP, int KQ, int NW) { return wide4_table_lookup<48, 64>(MP, KQ, NW); }
}  // namespace hipnmf
