// Shared device/host helpers for libossid_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ossid_hip.h"

#define OSSID_ABI_VERSION 1

static inline int ossid_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? OSSID_OK : OSSID_ELAUNCH;
}

// 64-lane wave reductions (xor butterfly; every lane ends with the result)
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
