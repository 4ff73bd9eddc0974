// fit_wide4_kernel<MP, KQ, NW>, MP = 96, 128 (nmf_wide4.hpp)
#include "nmf_wide4_inst.hpp"
namespace hipnmf {
const WideKernel<float>* wide4_kernel_f32_hi(int MP, int KQ, int NW) { return wide4_table_lookup<96, 128>(MP, KQ, NW); }
}  // namespace hipnmf
