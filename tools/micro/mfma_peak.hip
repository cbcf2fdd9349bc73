// Diagnostic: sustained v_mfma_f32_32x32x16_f16 rate on this box (operands in registers), for 1..4 waves per SIMD,
// with 1 or 2 independent accumulators per wave.  Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 8 * NACC;
    const double flops = mfmas * 32.0 * 32 * 16 * 2;
    // cycles per MFMA per SIMD if the clock were 2.4 GHz
    printf("acc=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD => %.2f GHz if 32 cycles each)\n", NACC, blocks_per_cu, ms,
           flops / ms / 1e9, ms * 1e6 / (mfmas / 1024.0), 32.0 / (ms * 1e6 / (mfmas / 1024.0)));
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) { run<1>(w, 20000); run<2>(w, 10000); }
    return 0;
}
