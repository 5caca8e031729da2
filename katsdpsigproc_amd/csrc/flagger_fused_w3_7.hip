// Fused flagger for median windows of 3, 5, 7 channels (see flagger_fused_kernel.h).
#include "flagger_fused_kernel.h"

int ksp_fused_launch_w3_7(int width, int device, hipStream_t s, const FusedParams &p,
                          hipEvent_t ev0, hipEvent_t ev1)
{
    switch (width) {
    case 3: return launch_fused<64, 3>(device, s, p, ev0, ev1);
    case 5: return launch_fused<64, 5>(device, s, p, ev0, ev1);
    case 7: return launch_fused<64, 7>(device, s, p, ev0, ev1);
    default:
        ksp_set_error("fused flagger: width %d is not compiled here", width);
        return (int)hipErrorInvalidValue;
    }
}
