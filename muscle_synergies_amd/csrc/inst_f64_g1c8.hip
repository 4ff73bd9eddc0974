#include "nmf_inst.hpp"
namespace hipnmf {
HIPNMF_DEFINE_TABLE(double, f64_g1c8, 1, 8)
}
