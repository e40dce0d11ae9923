// halo2-internals expansion kernels for 16 limbs per spread (num_bits_lookup = 1); see hsw_expand.hpp.
#include "hsw_expand.hpp"
namespace hsw {
template hipError_t launch_expand_L_internals_wide<16>(const ExpandParams &, hipStream_t);
}
