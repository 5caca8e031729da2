// Portability vocabulary for run-time compiled kernels (HIP, gfx950).
//
// The reference writes its kernels once for CUDA and OpenCL behind a set of macros
// (reference port.mako:19-116). Kernels handed to accel.build() here may use the same
// words, so that code written for the reference keeps its spelling; only the HIP
// meanings are defined, there is no second backend behind them.
#pragma once
#include <hip/hip_runtime.h>

#define KERNEL extern "C" __global__
#define DEVICE_FN __device__
#define GLOBAL
#define LOCAL
#define LOCAL_DECL __shared__
#define RESTRICT __restrict__
#define REQD_WORK_GROUP_SIZE(x, y, z) __launch_bounds__((x) * (y) * (z))
#define BARRIER() __syncthreads()
#define SHUFFLE_AVAILABLE 1
#define KSP_SIMD_GROUP_SIZE 64
// hiprtc compiles without the host's <math.h>: the two constants kernels reach for
#ifndef INFINITY
#define INFINITY (__builtin_inff())
#endif
#ifndef NAN
#define NAN (__builtin_nanf(""))
#endif

__device__ static inline unsigned get_local_id(int dim)
{
    return dim == 0 ? threadIdx.x : dim == 1 ? threadIdx.y : threadIdx.z;
}
__device__ static inline unsigned get_group_id(int dim)
{
    return dim == 0 ? blockIdx.x : dim == 1 ? blockIdx.y : blockIdx.z;
}
__device__ static inline unsigned get_local_size(int dim)
{
    return dim == 0 ? blockDim.x : dim == 1 ? blockDim.y : blockDim.z;
}
__device__ static inline unsigned get_num_groups(int dim)
{
    return dim == 0 ? gridDim.x : dim == 1 ? gridDim.y : gridDim.z;
}
__device__ static inline unsigned get_global_id(int dim)
{
    return get_group_id(dim) * get_local_size(dim) + get_local_id(dim);
}
__device__ static inline unsigned get_global_size(int dim)
{
    return get_num_groups(dim) * get_local_size(dim);
}
__device__ static inline float as_float(unsigned x) { return __uint_as_float(x); }
__device__ static inline unsigned as_uint(float x) { return __float_as_uint(x); }
__device__ static inline int as_int(float x) { return __float_as_int(x); }
