// Which MFMA shape should the S16 convolution's K loop use?  Both loops below do the work of one 32(cout) x 32(pixel) wave tile
// of k_conv_s16 per 32 input channels with the f16x3 split — 3 x 32x32x32 products, operands re-read from LDS by ds_read_b128
// exactly as in the kernel (8 fragment reads per 32 channels) — on RANDOM fp16 data (the clock this part holds under an MFMA
// loop depends on the data, MI355X_MICROARCH.md "DVFS give-back"):
//     A: 6 x v_mfma_f32_32x32x16_f16 (two 16-channel k-steps x 3 products)                       -> what k_conv_s16 issues today
//     B: 12 x v_mfma_f32_16x16x32_f16 (4 sub-tiles of 16x16, 3 products of K = 32 each)
// Reported per variant: TFLOP/s of f16 MFMA work, ns per 32-channel step per wave, and the in-kernel clock
// (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups).  2 waves per SIMD (8-wave workgroups, one per CU),
// launches long enough (tens of ms) for the clock to settle.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_shapes.hip -o mfma_shapes ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int WAVES = 8, LDS_HALVES = 32 * 1024;        // 64 KiB of random fp16 per workgroup

__device__ __forceinline__ half8 lds_read(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }

template <int SHAPE>
__global__ __launch_bounds__(64 * WAVES) void k_loop(const _Float16* __restrict__ src, float* __restrict__ out, int steps,
                                                     unsigned long long* __restrict__ stamps) {
    extern __shared__ _Float16 lds[];
    for (int i = threadIdx.x; i < LDS_HALVES; i += 64 * WAVES) lds[i] = src[(size_t)blockIdx.x % 4 * LDS_HALVES + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // eight fragment images of 1 KiB per 32-channel step (a_hi/a_lo x 2 k-steps, b_hi/b_lo x 2), offset per wave so that waves
    // read different data; a lane reads 16 bytes at lane * 16 of each image (conflict-free, like the kernel's operand images)
    const _Float16* base = lds + ((wave * 8 * 512) & (LDS_HALVES - 1)) + lane * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int s = 0; s < steps; ++s) {
            const _Float16* p = base + ((s * 4096) & (LDS_HALVES - 8 * 512 - 1) & ~511);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 ah = lds_read(p + (ks * 4 + 0) * 512), al = lds_read(p + (ks * 4 + 1) * 512);
                const half8 bh = lds_read(p + (ks * 4 + 2) * 512), bl = lds_read(p + (ks * 4 + 3) * 512);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) sum += acc[i];
    } else {
        f32x4 acc[4];
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
        for (int s = 0; s < steps; ++s) {
            const _Float16* p = base + ((s * 4096) & (LDS_HALVES - 8 * 512 - 1) & ~511);
            // per 32 channels: A operands for the two cout halves (hi|lo, hi|hi', lo|..: three K = 32 operand pairs per half are
            // built from four 16-byte fragments each in the real kernel; here the same 8 reads feed the same 12 MFMAs)
            half8 f[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = lds_read(p + i * 512);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const half8 a0 = f[t & 1], a1 = f[2 + (t & 1)], b0 = f[4 + (t >> 1)], b1 = f[6 + (t >> 1)];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[t], 0, 0, 0);
            }
        }
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) sum += acc[t][i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static int run(const char* name, const _Float16* src, float* out, unsigned long long* stamps, int blocks, int steps) {
    const size_t lds = LDS_HALVES * 2;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(k_loop<SHAPE>, dim3(blocks), dim3(64 * WAVES), lds, 0, src, out, steps, stamps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_loop<SHAPE>, dim3(blocks), dim3(64 * WAVES), lds, 0, src, out, steps, stamps);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flops = (double)blocks * WAVES * steps * 3.0 * 2.0 * 32 * 32 * 32;
    printf("%-28s %7.2f ms  %7.1f TFLOP/s f16-MFMA  %6.1f ns per 32-channel step per wave  in-kernel clock %.2f GHz (median)\n", name, ms,
           flops / ms / 1e9, ms * 1e6 / steps, ghz[ghz.size() / 2]);
    return 0;
}

int main() {
    const int blocks = 256, steps = 40000;
    std::vector<_Float16> host(4 * LDS_HALVES);
    unsigned s = 12345u;
    for (auto& v : host) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f); }
    _Float16* src; float* out; unsigned long long* stamps;
    CHECK(hipMalloc(&src, host.size() * 2));
    CHECK(hipMemcpy(src, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * WAVES * 4));
    CHECK(hipMalloc(&stamps, (size_t)2 * blocks * 8));
    printf("f16x3 wave tile 32x32 per 32 channels, operands from LDS (8 ds_read_b128 per step), random data, %d workgroups x %d waves\n", blocks, WAVES);
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0>("A: 6 x mfma 32x32x16", src, out, stamps, blocks, steps)) return 1;
        if (run<1>("B: 12 x mfma 16x16x32", src, out, stamps, blocks, steps)) return 1;
    }
    return 0;
}
