// Fused flagger for median windows of 19, 21 channels (see flagger_fused_kernel.h).
#include "flagger_fused_kernel.h"

int ksp_fused_launch_w19_21(int width, int device, hipStream_t s, const FusedParams &p,
                          hipEvent_t ev0, hipEvent_t ev1)
{
    switch (width) {
    case 19: return launch_fused<64, 19>(device, s, p, ev0, ev1);
    case 21: return launch_fused<64, 21>(device, s, p, ev0, ev1);
    default:
        ksp_set_error("fused flagger: width %d is not compiled here", width);
        return (int)hipErrorInvalidValue;
    }
}
