// 2-D transpose for gfx950 (stands in for reference transpose.mako:44-73).
//
// One 256-thread workgroup moves a 64x64-element tile through LDS. Global reads
// and writes are both row-contiguous: the vector path moves V elements (16 bytes
// for 4-byte types) per lane so that a wave touches whole 256-byte row segments;
// the tile is stored with a one-word row pad so the transposed LDS read is
// conflict-free. Workgroups are enumerated along the *output* rows first so that
// consecutive blocks (which land on different XCDs) write neighbouring lines.
// HBM-bound: 2 * elem_size bytes per element.
#include "ksp_common.h"

template <typename T, int V>
__global__ __launch_bounds__(256) void transpose_kernel(T *__restrict__ dst,
                                                        const T *__restrict__ src, int in_rows,
                                                        int in_cols, int out_stride, int in_stride)
{
    constexpr int TILE = 64;
    __shared__ T tile[TILE][TILE + 1];
    const int t = threadIdx.x;
    // blockIdx.x walks tiles down the input rows (= along output rows' columns)
    const int tile_r = blockIdx.x * TILE;  // first input row of the tile
    const int tile_c = blockIdx.y * TILE;  // first input column of the tile

    constexpr int LANES_PER_ROW = TILE / V;        // lanes covering one tile row
    constexpr int ROWS_PER_PASS = 256 / LANES_PER_ROW;
    const int lr = t / LANES_PER_ROW;
    const int lc = (t % LANES_PER_ROW) * V;

#pragma unroll
    for (int p = 0; p < TILE / ROWS_PER_PASS; p++) {
        int r = p * ROWS_PER_PASS + lr;
        int gr = tile_r + r, gc = tile_c + lc;
        if (gr < in_rows) {
            if (V > 1 && gc + V <= in_cols) {
                T v[V];
                __builtin_memcpy(v, __builtin_assume_aligned(
                                        src + (size_t)gr * in_stride + gc, sizeof(T) * V),
                                 sizeof(T) * V);
#pragma unroll
                for (int i = 0; i < V; i++) tile[r][lc + i] = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < V; i++)
                    if (gc + i < in_cols) tile[r][lc + i] = src[(size_t)gr * in_stride + gc + i];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < TILE / ROWS_PER_PASS; p++) {
        int r = p * ROWS_PER_PASS + lr;  // output row within the tile = input column
        int gr = tile_c + r, gc = tile_r + lc;
        if (gr < in_cols) {
            if (V > 1 && gc + V <= in_rows) {
                T v[V];
#pragma unroll
                for (int i = 0; i < V; i++) v[i] = tile[lc + i][r];
                __builtin_memcpy(__builtin_assume_aligned(dst + (size_t)gr * out_stride + gc,
                                                          sizeof(T) * V),
                                 v, sizeof(T) * V);
            } else {
#pragma unroll
                for (int i = 0; i < V; i++)
                    if (gc + i < in_rows) dst[(size_t)gr * out_stride + gc + i] = tile[lc + i][r];
            }
        }
    }
}

template <typename T, int V>
static int launch_transpose(hipStream_t stream, void *dst, const void *src, int in_rows,
                            int in_cols, int out_stride, int in_stride)
{
    dim3 grid(ksp_divup(in_rows, 64), ksp_divup(in_cols, 64));
    hipLaunchKernelGGL((transpose_kernel<T, V>), grid, dim3(256), 0, stream, (T *)dst,
                       (const T *)src, in_rows, in_cols, out_stride, in_stride);
    KSP_LAUNCH_CHECK();
    return 0;
}

struct alignas(8) ksp_b8 { uint32_t a, b; };
struct alignas(16) ksp_b16 { uint32_t a, b, c, d; };

extern "C" int ksp_transpose(int device, void *stream, void *dst, const void *src, int in_rows,
                             int in_cols, int out_stride, int in_stride, int elem_size)
{
    KSP_REQUIRE(in_rows >= 0 && in_cols >= 0, "negative shape");
    KSP_REQUIRE(in_stride >= in_cols && out_stride >= in_rows, "stride smaller than row");
    KSP_REQUIRE(dst != nullptr && src != nullptr, "NULL buffer");
    if (in_rows == 0 || in_cols == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    // The vector path needs 16-byte (or V*elem) aligned row starts on both sides.
    auto aligned = [&](int v) {
        size_t bytes = (size_t)v * elem_size;
        return ((uintptr_t)dst % bytes == 0) && ((uintptr_t)src % bytes == 0) &&
               ((size_t)in_stride * elem_size % bytes == 0) &&
               ((size_t)out_stride * elem_size % bytes == 0);
    };
    switch (elem_size) {
    case 1:
        return aligned(4) ? launch_transpose<uint8_t, 4>(s, dst, src, in_rows, in_cols, out_stride, in_stride)
                          : launch_transpose<uint8_t, 1>(s, dst, src, in_rows, in_cols, out_stride, in_stride);
    case 2:
        return aligned(4) ? launch_transpose<uint16_t, 4>(s, dst, src, in_rows, in_cols, out_stride, in_stride)
                          : launch_transpose<uint16_t, 1>(s, dst, src, in_rows, in_cols, out_stride, in_stride);
    case 4:
        return aligned(4) ? launch_transpose<uint32_t, 4>(s, dst, src, in_rows, in_cols, out_stride, in_stride)
                          : launch_transpose<uint32_t, 1>(s, dst, src, in_rows, in_cols, out_stride, in_stride);
    case 8:
        return aligned(2) ? launch_transpose<ksp_b8, 2>(s, dst, src, in_rows, in_cols, out_stride, in_stride)
                          : launch_transpose<ksp_b8, 1>(s, dst, src, in_rows, in_cols, out_stride, in_stride);
    case 16:
        return launch_transpose<ksp_b16, 1>(s, dst, src, in_rows, in_cols, out_stride, in_stride);
    default:
        ksp_set_error("ksp_transpose: unsupported element size %d", elem_size);
        return (int)hipErrorInvalidValue;
    }
}
