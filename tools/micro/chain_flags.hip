// What does a layer boundary cost as a LAUNCH boundary, and what as a per-tile dependency inside one resident launch?
//
// the refinement iteration is 53 launches on one dependency chain at ~6.4 us of launch each; the one structure that has
// not been priced on this part is a multi-layer RESIDENT kernel whose workgroups walk layer by layer over their own tile and wait on arrival
// counters of the producer tiles they read (not a grid barrier: round 2 priced that at 4-7 us, no better than a launch boundary).
//
// Model: T tiles (one workgroup of 256 threads each), L layers.  In layer l workgroup t reads tiles t-1, t, t+1 of buffer l-1 (16 KiB each: a
// 4 x 32-pixel x 32-channel fp32 tile), spends `work` iterations of dependent FMAs on them (the K loop's stand-in), writes tile t of buffer l.
//   mode G   L launches in a replayed HIP graph, one layer each (what the frame does today)
//   mode R1  one launch; plain stores, release fence (buffer_wbl2 sc1), counter += 1; consumers poll the three counters, acquire fence (buffer_inv sc1)
//   mode R2  one launch; write-through stores (agent-scope atomic stores: sc1), s_waitcnt, counter += 1; consumers poll, then read with sc1 loads
//   mode R3  one launch, XCD-LOCAL handoff: tiles are dealt to the XCDs in contiguous strips (workgroup w runs on XCD w % 8 and owns tile (w % 8) * T/8 + w / 8), so
//            the neighbours of an interior tile were produced on the same XCD: plain stores (they KEEP the line in the XCD's L2; sc1 stores drop it) + s_waitcnt,
//            counters as L2 atomics (workgroup scope), consumers read with sc1 loads (L1 bypassed, L2-served: MI355X_MICROARCH.md) — no trip to the memory
//            side.  Only the two tiles at each strip border store through (sc1).  This mode relies on the OBSERVED placement (blocks b and b + 8 share an XCD),
//            which HIP does not promise: it prices the hand-off, it is not a protocol to ship; the checksum says whether the placement held.
//            (A first version polled and read with sc0 loads: those hit the L1 like plain loads, every wait timed out.)
// Spins are BOUNDED (a timeout sets an error word and falls through): a protocol bug is a wrong checksum, never a hung GPU.  The three modes must
// produce the same checksum — a stale read across XCDs shows up there.
// Build: hipcc -O3 --offload-arch=gfx950 chain_flags.hip -o chain_flags ; run on the GPU box: chain_flags [tiles] [layers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int TILE_U64 = 2048;          // 16 KiB per tile as 8-byte words: 256 threads x 8 words
constexpr int PER_THREAD = TILE_U64 / 256;

struct Args {
    unsigned long long* buf;            // [L + 1][T][TILE_U64]
    unsigned* cnt;                      // [L + 1][T] arrival counters (monotonic over launches)
    unsigned* err;                      // != 0: a spin timed out
    const int* salt_dev;                // (graph mode: arguments are baked into the graph, the salt is read from memory)
    int T, L, work;
    int salt;                           // changes per repetition: a value read stale from the previous repetition changes the checksum
    unsigned epoch;                     // counters reach `epoch` when a tile of this launch has arrived
};

__device__ __forceinline__ unsigned long long mix(unsigned long long a, unsigned long long b, unsigned long long c, int work, int salt) {
    unsigned long long v = a * 3 + b * 5 + c * 7 + 1 + (unsigned long long)salt;
    float f = (float)(v & 1023);
    for (int i = 0; i < work; ++i) f = fmaf(f, 1.0000001f, 0.5f);       // dependent chain: ~4 cycles per iteration
    return v + (unsigned long long)((int)f & 1);
}

template <int MODE>   // 0: plain loads / stores (one layer per launch); 1: plain + fences; 2: sc1 stores and loads; 3: XCD-local (sc0 loads), sc1 at strip borders
__device__ __forceinline__ void layer(const Args& a, int l, int t) {
    const int tid = threadIdx.x;
    const unsigned long long* in = a.buf + (size_t)(l - 1) * a.T * TILE_U64;
    unsigned long long* out = a.buf + (size_t)l * a.T * TILE_U64 + (size_t)t * TILE_U64;
    const int tl = t > 0 ? t - 1 : t, tr = t + 1 < a.T ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
        const int k = j * 256 + tid;
        unsigned long long x, y, z;
        if (MODE == 2) {
            x = __hip_atomic_load(in + (size_t)tl * TILE_U64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            y = __hip_atomic_load(in + (size_t)t * TILE_U64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z = __hip_atomic_load(in + (size_t)tr * TILE_U64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 3) {
            x = __hip_atomic_load(in + (size_t)tl * TILE_U64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1: L1 bypassed, served by the L2
            y = __hip_atomic_load(in + (size_t)t * TILE_U64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z = __hip_atomic_load(in + (size_t)tr * TILE_U64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            x = in[(size_t)tl * TILE_U64 + k]; y = in[(size_t)t * TILE_U64 + k]; z = in[(size_t)tr * TILE_U64 + k];
        }
        const unsigned long long v = mix(x, y, z, a.work, a.salt);
        const bool border = MODE == 3 && (t % (a.T / 8) == 0 || t % (a.T / 8) == a.T / 8 - 1);     // read by another XCD: write through
        if (MODE == 2 || border) __hip_atomic_store(out + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[k] = v;
    }
}

__global__ __launch_bounds__(256) void k_one_layer(Args a, int l) { a.salt = *a.salt_dev; layer<0>(a, l, blockIdx.x); }

template <int MODE>
__global__ __launch_bounds__(256) void k_resident(Args a) {
    const int t = MODE == 3 ? (int)(blockIdx.x & 7) * (a.T / 8) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    for (int l = 1; l <= a.L; ++l) {
        if (l > 1) {
            // wait for the three producer tiles of layer l-1 (threads 0..2 poll one counter each, bounded)
            if (threadIdx.x < 3) {
                const int q = t - 1 + (int)threadIdx.x;
                if (q >= 0 && q < a.T) {
                    const unsigned* c = a.cnt + (size_t)(l - 1) * a.T + q;
                    int spins = 0;
                    const int S_ = a.T / 8;
                    const bool q_border = q % S_ == 0 || q % S_ == S_ - 1;              // border tiles signal (and store) through the memory side
                    const bool local = MODE == 3 && q / S_ == t / S_ && !q_border;
                    (void)local;
                    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.epoch) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 15)) { atomicOr(a.err, 1u); break; }      // ~5 ms: a lost hand-off costs a wrong checksum, not minutes
                    }
                }
            }
            __syncthreads();
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // buffer_inv sc1: drop what this XCD's caches hold of the buffers
        }
        layer<MODE>(a, l, t);
        if (l < a.L) {
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // buffer_wbl2 sc1 + wait
            else __builtin_amdgcn_s_waitcnt(0);                                    // the write-through stores have been acknowledged
            __syncthreads();
            if (threadIdx.x == 0) {
                const bool border = MODE == 3 && (t % (a.T / 8) == 0 || t % (a.T / 8) == a.T / 8 - 1);
                if (MODE == 3 && !border) __hip_atomic_fetch_add(a.cnt + (size_t)l * a.T + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // L2 atomic
                else __hip_atomic_fetch_add(a.cnt + (size_t)l * a.T + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

static int run(const char* name, int mode, Args a, hipStream_t s, unsigned long long* h_sum) {
    const int REP = 20;
    hipGraph_t g; hipGraphExec_t ge;
    static unsigned epoch = 0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    if (mode == 0) {
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int l = 1; l <= a.L; ++l) hipLaunchKernelGGL(k_one_layer, dim3(a.T), dim3(256), 0, s, a, l);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < REP + 3; ++r) {
            CHECK(hipMemcpyAsync(const_cast<int*>(a.salt_dev), &r, 4, hipMemcpyHostToDevice, s));
            CHECK(hipStreamSynchronize(s));
            CHECK(hipEventRecord(e0, s));
            CHECK(hipGraphLaunch(ge, s));
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3 && ms < best) best = ms;
        }
    } else {
        for (int r = 0; r < REP + 3; ++r) {
            a.epoch = ++epoch;
            a.salt = r;
            CHECK(hipEventRecord(e0, s));
            if (mode == 1) hipLaunchKernelGGL(k_resident<1>, dim3(a.T), dim3(256), 0, s, a);
            else if (mode == 2) hipLaunchKernelGGL(k_resident<2>, dim3(a.T), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(k_resident<3>, dim3(a.T), dim3(256), 0, s, a);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3 && ms < best) best = ms;
        }
    }
    // checksum of the last layer
    std::vector<unsigned long long> h((size_t)a.T * TILE_U64);
    CHECK(hipMemcpy(h.data(), a.buf + (size_t)a.L * a.T * TILE_U64, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (auto v : h) sum = sum * 1099511628211ull + v;
    unsigned err = 0;
    CHECK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
    *h_sum = sum;
    printf("  %-44s %8.2f us total  %6.2f us per layer   checksum %016llx%s\n", name, best * 1e3f, best * 1e3f / a.L, sum, err ? "  SPIN TIMEOUT" : "");
    return 0;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 600, L = argc > 2 ? atoi(argv[2]) : 8;
    hipStream_t s; CHECK(hipStreamCreate(&s));
    Args a; a.T = T; a.L = L; a.epoch = 0;
    CHECK(hipMalloc(&a.buf, (size_t)(L + 1) * T * TILE_U64 * 8));
    CHECK(hipMalloc(&a.cnt, (size_t)(L + 1) * T * 4));
    CHECK(hipMalloc(&a.err, 4));
    int* salt_dev; CHECK(hipMalloc(&salt_dev, 4)); a.salt_dev = salt_dev; a.salt = 0;
    CHECK(hipMemset(a.cnt, 0, (size_t)(L + 1) * T * 4));
    CHECK(hipMemset(a.err, 0, 4));
    std::vector<unsigned long long> h0((size_t)T * TILE_U64);
    for (size_t i = 0; i < h0.size(); ++i) h0[i] = i * 2654435761ull + 12345;
    for (int work : {0, 40, 160}) {                     // dependent FMAs per element (8 elements per thread): 0 / ~8 / ~25 us per layer
        a.work = work;
        printf("tiles %d (x 256 threads), layers %d, work %d FMAs per element chain:\n", T, L, work);
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 1 && getenv("CHAIN_SKIP_R1")) { s1 = s0; continue; }
            CHECK(hipMemset(a.buf, 0, (size_t)(L + 1) * T * TILE_U64 * 8));
            CHECK(hipMemcpy(a.buf, h0.data(), h0.size() * 8, hipMemcpyHostToDevice));
            const char* names[4] = {"G : one launch per layer (graph replay)", "R1: resident, plain stores + agent fences", "R2: resident, sc1 stores / loads",
                                    "R3: resident, XCD-local strips (sc0 loads)"};
            if (run(names[mode], mode, a, s, mode == 0 ? &s0 : (mode == 1 ? &s1 : (mode == 2 ? &s2 : &s3)))) return 1;
        }
        printf("  checksums %s\n", (s0 == s1 && s0 == s2 && s0 == s3) ? "AGREE" : "DIFFER (a stale read or a protocol bug)");
    }
    return 0;
}
