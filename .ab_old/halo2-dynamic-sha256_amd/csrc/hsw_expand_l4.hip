// Expansion kernels for 4 limb(s) per spread (num_bits_lookup = 4); see hsw_expand.hpp.
#include "hsw_expand.hpp"
namespace hsw {
template hipError_t launch_expand_L<4>(const ExpandParams &, int, hipStream_t);
}
