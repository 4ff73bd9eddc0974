#include "nmf_inst.hpp"
namespace hipnmf {
HIPNMF_DEFINE_TABLE(float, f32_g2c8, 2, 8)
}
