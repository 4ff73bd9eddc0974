#include "nmf_inst.hpp"
namespace hipnmf {
HIPNMF_DEFINE_TABLE(float, f32_g4c8, 4, 8)
}
