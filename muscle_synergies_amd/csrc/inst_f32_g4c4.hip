#include "nmf_inst.hpp"
namespace hipnmf {
HIPNMF_DEFINE_TABLE(float, f32_g4c4, 4, 4)
}
