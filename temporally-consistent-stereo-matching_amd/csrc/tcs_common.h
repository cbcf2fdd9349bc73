// Shared helpers for the gfx950 kernels of libtcs_mi355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tcs_mi355.h"

#define TCS_WAVE 64

static inline hipStream_t tcs_stream(tcs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int tcs_launch_status() {
    return hipGetLastError() == hipSuccess ? TCS_OK : TCS_ELAUNCH;
}

static inline int tcs_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// wave-level reductions over the 64 lanes
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
