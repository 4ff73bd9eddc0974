#include "nmf_inst.hpp"
namespace hipnmf {
HIPNMF_DEFINE_TABLE(float, f32_g1c8, 1, 8)
}
