// Persistent ring kernel for the fused flagger at 4096 channels (ring_kernel.h): the
// instantiations and the entry points flagger_fused.hip dispatches to.
#include <hip/hip_ext.h>

#include <atomic>

#include "ring_kernel.h"

bool ksp_ring_supported(const FusedParams &p, int width) { return ring_supported(p, width); }

int ksp_ring_launch(int width, int device, hipStream_t s, const FusedParams &p, int n_cu,
                    hipEvent_t ev0, hipEvent_t ev1)
{
    switch (width) {
    case 13: return launch_ring<13>(device, s, p, n_cu, ev0, ev1);
    default: break;
    }
    ksp_set_error("ksp_ring_launch: width %d not instantiated", width);
    return (int)hipErrorNotSupported;
}
