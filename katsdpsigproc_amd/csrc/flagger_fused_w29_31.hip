// Fused flagger for median windows of 29, 31 channels (see flagger_fused_kernel.h).
#include "flagger_fused_kernel.h"

int ksp_fused_launch_w29_31(int width, int device, hipStream_t s, const FusedParams &p,
                            hipEvent_t ev0, hipEvent_t ev1)
{
    switch (width) {
    case 29: return launch_fused<64, 29>(device, s, p, ev0, ev1);
    case 31: return launch_fused<64, 31>(device, s, p, ev0, ev1);
    default:
        ksp_set_error("fused flagger: width %d is not compiled here", width);
        return (int)hipErrorInvalidValue;
    }
}
