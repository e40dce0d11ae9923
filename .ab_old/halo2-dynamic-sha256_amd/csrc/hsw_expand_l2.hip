// Expansion kernels for 2 limb(s) per spread (num_bits_lookup = 8); see hsw_expand.hpp.
#include "hsw_expand.hpp"
namespace hsw {
template hipError_t launch_expand_L<2>(const ExpandParams &, int, hipStream_t);
}
