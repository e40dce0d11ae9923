// Emit-time Montgomery expansion kernels (Em::M32) for 2 limbs per spread (num_bits_lookup = 8); see hsw_expand.hpp.
#include "hsw_expand.hpp"
namespace hsw {
template hipError_t launch_expand_m32<2>(const ExpandParams &, hipStream_t);
}
