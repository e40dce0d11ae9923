// The small-batch kernel (hsw_small.hpp) for the reference's 8-bit spread table, in a translation unit of its own.
#include "hsw_small.hpp"
namespace hsw {
hipError_t launch_small(const ExpandParams &p, const SmallFrames *frames, int limbs, hipStream_t stream) {
    if (limbs != 2) return hipErrorInvalidValue;
    return launch_small_L<2>(p, frames, stream);
}
}  // namespace hsw
