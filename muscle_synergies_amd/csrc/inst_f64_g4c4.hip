#include "nmf_inst.hpp"
namespace hipnmf {
HIPNMF_DEFINE_TABLE(double, f64_g4c4, 4, 4)
}
