// Expansion kernels for 8 limb(s) per spread (num_bits_lookup = 2); see hsw_expand.hpp.
#include "hsw_expand.hpp"
namespace hsw {
template hipError_t launch_expand_L<8>(const ExpandParams &, int, hipStream_t);
}
