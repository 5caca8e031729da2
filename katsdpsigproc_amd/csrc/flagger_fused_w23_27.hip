// Fused flagger for median windows of 23, 25, 27 channels (see flagger_fused_kernel.h).
#include "flagger_fused_kernel.h"

int ksp_fused_launch_w23_27(int width, int device, hipStream_t s, const FusedParams &p,
                            hipEvent_t ev0, hipEvent_t ev1)
{
    switch (width) {
    case 23: return launch_fused<64, 23>(device, s, p, ev0, ev1);
    case 25: return launch_fused<64, 25>(device, s, p, ev0, ev1);
    case 27: return launch_fused<64, 27>(device, s, p, ev0, ev1);
    default:
        ksp_set_error("fused flagger: width %d is not compiled here", width);
        return (int)hipErrorInvalidValue;
    }
}
