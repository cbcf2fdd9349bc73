// What does the SIZE of a kernel's argument block cost per launch inside a replayed HIP graph?
//
// k_conv_s16 takes a 400-byte argument struct; with HIP_FORCE_DEV_KERNARG=0 (arguments in host memory) the frame is 2.5 ms slower,
// 1.5 us per launch (profiles/r03_ab_logs.txt), so the argument fetch is on the launch's critical path.  The first 16 dwords arrive in
// SGPRs at wave launch (-amdgpu-kernarg-preload-count=16); everything behind them is an s_load from a cold address at the head of every
// wave.  This micro-benchmark chains 400 launches of one kernel in a graph (each depends on the previous one through a buffer, as
// the layers of the loop do) and reports us per launch for
//     K_small  64-byte arguments, all preloaded
//     K_big    448-byte arguments, the kernel needs a field at the END before it can do anything
//     K_big0   448-byte arguments, the kernel only needs fields of the first 64 bytes
// on a grid of 300 x 256 threads (a 1/4-scale layer).
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 kernarg_latency.hip -o kernarg_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct Small { float* buf; int n; int pad[13]; };                       // 64 bytes
struct Big { float* buf; int n; int pad[13]; int more[95]; int last; };   // 448 bytes

__global__ __launch_bounds__(256) void k_small(Small a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.buf[i] += 1.0f;
}
__global__ __launch_bounds__(256) void k_big(Big a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n + a.last) a.buf[i] += 1.0f;          // a.last (= 0) sits at byte 444: an s_load behind the preloaded 64 bytes
}
__global__ __launch_bounds__(256) void k_big0(Big a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.buf[i] += 1.0f;
}

template <class F>
static int timed(const char* name, F launch, hipStream_t s) {
    const int N = 400;
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) launch(s);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a, s));
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("%-8s %7.3f us per launch (best of 5 replays of %d chained launches)\n", name, best * 1e3f / N, N);
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    return 0;
}

int main() {
    const int n = 300 * 256;
    float* buf; CHECK(hipMalloc(&buf, n * sizeof(float))); CHECK(hipMemset(buf, 0, n * sizeof(float)));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    Small sa{}; sa.buf = buf; sa.n = n;
    Big ba{}; ba.buf = buf; ba.n = n; ba.last = 0;
    for (int round = 0; round < 2; ++round) {
        if (timed("K_small", [&](hipStream_t st) { hipLaunchKernelGGL(k_small, dim3(300), dim3(256), 0, st, sa); }, s)) return 1;
        if (timed("K_big", [&](hipStream_t st) { hipLaunchKernelGGL(k_big, dim3(300), dim3(256), 0, st, ba); }, s)) return 1;
        if (timed("K_big0", [&](hipStream_t st) { hipLaunchKernelGGL(k_big0, dim3(300), dim3(256), 0, st, ba); }, s)) return 1;
    }
    return 0;
}
