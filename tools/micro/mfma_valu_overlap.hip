// Diagnostic: do MFMA work and VALU work of DIFFERENT waves on the same SIMD overlap?  Block = 8 waves: waves 0-3 run
// an MFMA loop, waves 4-7 a VALU (v_fma_f32) loop; timed alone and together, one block per CU.
// Also: MFMA + VALU interleaved inside ONE wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int valu_iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (mode == 2) {            // same wave: 1 MFMA + 6 FMAs interleaved
        if (wave < 4) {
            half8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
            f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f;
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                    x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f);
                    x3 = fmaf(x3, 1.0001f, 0.5f); x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (int i = 0; i < 16; ++i) s += acc[i];
            s += x0 + x1 + x2 + x3 + x4 + x5;
        }
    } else if (wave < 4) {
        if (mfma_iters > 0) {
            half8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
            f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            for (int i = 0; i < 16; ++i) s += acc[i];
        }
    } else {
        float x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
        for (int it = 0; it < valu_iters; ++it)
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
        for (int i = 0; i < 8; ++i) s += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float run(int mi, int vi, int mode) {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 512>>>(out, 10, 10, mode); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<<<256, 512>>>(out, mi, vi, mode); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipFree(out); return ms;
}

int main() {
    const int MI = 20000, VI = 20000;       // 160k MFMAs (x32 cycles = 5.1M) ; 960k FMAs per VALU wave (x4 cycles = 3.8M)
    printf("MFMA waves alone            : %.3f ms\n", run(MI, 0, 0));
    printf("VALU waves alone            : %.3f ms\n", run(0, VI, 0));
    printf("MFMA waves + VALU waves     : %.3f ms   (sum if exclusive, max if they overlap)\n", run(MI, VI, 0));
    printf("one wave: MFMA + 6 FMA each : %.3f ms   (160k MFMAs + 960k FMAs in the same wave)\n", run(MI, 0, 2));
    return 0;
}
