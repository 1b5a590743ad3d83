// tip_fused.hip — fused execution plan (paper configuration): placeholder until the kernel lands.
#include "tip_internal.h"

namespace tip {

bool fused_supported(const Dims&, int) { return false; }
size_t fused_packed_floats(const Dims&) { return 0; }
void fused_pack(const Dims&, const float* const*, float*) {}
hipError_t launch_fused_encoder(const Dims&, const float*, const float*, const float*, const float*, float, float*, int,
                                int, int, hipStream_t) {
    return hipErrorNotSupported;
}

}  // namespace tip
