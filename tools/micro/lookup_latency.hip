// Latency budget of the corr-lookup kernel at the single-sequence size (BASELINE configs[1]: 19,200 pixels, 5.9 MB per launch).
//
// k_corr_lookup is three dependent memory round trips long: coordinate load -> 10 tap loads -> 9 stores, on a grid of 1,200
// single-wave workgroups.  This micro-benchmark measures what each hop costs on THIS part with the SAME grid and the same
// footprints (77 KB coordinate plane written by the preceding kernel, 23 MB pyramid, 2.76 MB output), by building the
// kernel up one hop at a time:
//     K0  nothing (first instruction -> last instruction)                      : dispatch ramp of the grid
//     K1  + coordinate load                                                     : + one load round trip
//     K2  + 10 tap loads whose addresses depend on the coordinate               : + a second, dependent round trip
//     K3  + 9 coalesced 256-B stores per wave, waited for (s_waitcnt vmcnt(0))  : + store acknowledgement
// Every variant stamps s_memrealtime (100 MHz) at its first instruction and after its last memory operation completed;
// the span of a launch = max(end) - min(start) over all workgroups, as bench.py measures the real kernel ("in_kernel").
// A "producer" kernel that rewrites the coordinate plane runs before every launch, as the blend kernel does in the frame.
// Build: hipcc -O3 --offload-arch=gfx950 lookup_latency.hip -o lookup_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int H = 120, W = 160, NPIX = H * W, GROUPS = NPIX / 64;      // 300 groups of 64 pixels

__global__ void k_producer(float* coords, int n, float t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) coords[i] = (float)(i % W) - (4.f + 30.f * (float)(i / W) / H + 2.f * __sinf((i % W) / 17.f + t));
}

template <int HOPS>
__global__ __launch_bounds__(64) void k_probe(const float* __restrict__ coords, const float* __restrict__ pyr0, const float* __restrict__ pyr1,
                                              const float* __restrict__ pyr2, const float* __restrict__ pyr3, float* __restrict__ out,
                                              unsigned long long* __restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x;
    const int level = blockIdx.x & 3, grp = blockIdx.x >> 2;
    const int p = grp * 64 + lane;
    const int h = p / W, w1 = p - h * W;
    float acc = 0.f;
    if (HOPS >= 1) {
        float x = coords[p] * (1.0f / (float)(1 << level));
        x = fminf(fmaxf(x, -1048576.f), 1048576.f);
        const float x0 = floorf(x), fr = x - x0;
        acc = fr;
        if (HOPS >= 2) {
            const int Wl = W >> level, q = w1 >> level, j0 = (int)x0 - 4;
            const float* pl = level == 0 ? pyr0 : (level == 1 ? pyr1 : (level == 2 ? pyr2 : pyr3));
            const float* base = pl + ((size_t)h * Wl) * W + w1;
            float v[10];
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const int j = j0 + t;
                const bool ok = j >= 0 && j < Wl;
                int d = q - j;
                d = d < 0 ? d + Wl : (d >= Wl ? d - Wl : d);
                d = ok ? d : 0;
                const float val = base[(size_t)d * W];
                v[t] = ok ? val : 0.f;
            }
            if (HOPS >= 3) {
                float* o = out + (size_t)level * 9 * NPIX + p;
#pragma unroll
                for (int t = 0; t < 9; ++t) o[(size_t)t * NPIX] = (1.f - fr) * v[t] + fr * v[t + 1];
            } else {
#pragma unroll
                for (int t = 0; t < 10; ++t) acc += v[t];
            }
        }
    }
    if (HOPS < 3 && acc == 123456.789f) out[p] = acc;            // keeps the loads alive without a store on the timed path
    __builtin_amdgcn_s_waitcnt(0);                                // every load has returned / every store is acknowledged
    if (lane == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int HOPS>
static int run(const char* what, float* coords, float** pyr, float* out, unsigned long long* stamps, hipStream_t s) {
    const int blocks = GROUPS * 4, reps = 200;
    std::vector<unsigned long long> host(2 * blocks);
    std::vector<double> spans;
    for (int r = 0; r < reps + 10; ++r) {
        hipLaunchKernelGGL(k_producer, dim3((NPIX + 255) / 256), dim3(256), 0, s, coords, NPIX, 0.01f * r);
        hipLaunchKernelGGL(k_probe<HOPS>, dim3(blocks), dim3(64), 0, s, coords, pyr[0], pyr[1], pyr[2], pyr[3], out, stamps);
        CHECK(hipMemcpyAsync(host.data(), stamps, host.size() * 8, hipMemcpyDeviceToHost, s));
        CHECK(hipStreamSynchronize(s));
        if (r < 10) continue;
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < blocks; ++b) { lo = std::min(lo, host[2 * b]); hi = std::max(hi, host[2 * b + 1]); }
        spans.push_back((hi - lo) * 0.01);                        // 100 MHz -> us
    }
    std::sort(spans.begin(), spans.end());
    printf("%-44s span us: min %.2f  median %.2f  p90 %.2f\n", what, spans.front(), spans[spans.size() / 2], spans[spans.size() * 9 / 10]);
    return 0;
}

int main() {
    float *coords, *out, *pyr[4];
    unsigned long long* stamps;
    CHECK(hipMalloc(&coords, NPIX * 4));
    CHECK(hipMalloc(&out, (size_t)36 * NPIX * 4));
    for (int i = 0; i < 4; ++i) {
        const size_t n = (size_t)H * (W >> i) * W;
        CHECK(hipMalloc(&pyr[i], n * 4));
        CHECK(hipMemset(pyr[i], 0, n * 4));
    }
    CHECK(hipMalloc(&stamps, (size_t)2 * GROUPS * 4 * 8));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    printf("corr-lookup latency budget, %d pixels, %d single-wave workgroups (algorithmic bytes 5,913,600 -> 0.74 us at 8 TB/s)\n", NPIX, GROUPS * 4);
    if (run<0>("K0 dispatch ramp (no memory operation)", coords, pyr, out, stamps, s)) return 1;
    if (run<1>("K1 + coordinate load", coords, pyr, out, stamps, s)) return 1;
    if (run<2>("K2 + 10 dependent tap loads", coords, pyr, out, stamps, s)) return 1;
    if (run<3>("K3 + 9 stores, acknowledged (= the kernel)", coords, pyr, out, stamps, s)) return 1;
    return 0;
}
