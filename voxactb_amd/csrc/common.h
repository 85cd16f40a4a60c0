// Shared device/host helpers for the gfx950 kernels (internal; the public C ABI is include/voxactb_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/voxactb_hip.h"

#define VXB_OK 0
#define VXB_EARG (-1)
#define VXB_ESIZE (-2)
#define VXB_EWS (-3)
#define VXB_ELAUNCH (-4)

#define VXB_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return VXB_ELAUNCH; \
    } while (0)

static inline int vxb_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
