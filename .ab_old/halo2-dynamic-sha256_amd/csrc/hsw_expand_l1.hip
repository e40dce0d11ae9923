// Expansion kernels for 1 limb(s) per spread (num_bits_lookup = 16); see hsw_expand.hpp.
#include "hsw_expand.hpp"
namespace hsw {
template hipError_t launch_expand_L<1>(const ExpandParams &, int, hipStream_t);
}
