#include "nmf_inst.hpp"
namespace hipnmf {
HIPNMF_DEFINE_TABLE(double, f64_g2c4, 2, 4)
}
