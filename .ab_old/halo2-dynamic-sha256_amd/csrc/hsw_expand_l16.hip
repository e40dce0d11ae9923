// Expansion kernels for 16 limb(s) per spread (num_bits_lookup = 1); see hsw_expand.hpp.
#include "hsw_expand.hpp"
namespace hsw {
template hipError_t launch_expand_L<16>(const ExpandParams &, int, hipStream_t);
}
