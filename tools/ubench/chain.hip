// Microbenchmark (developer tool, round 4): what does a greedy placement iteration cost at the least?  A chain of short
// DEPENDENT kernels (as cf_place's loop is), each doing N dependent memory round trips per wave on a buffer no cache holds:
//   mode 0 plain 8-byte load, 1 agent-scope (sc1) load, 2 returning 64-bit atomicAdd, 3 64-bit atomicCAS
// and optionally a "last workgroup done" tail (release fence + one counter + M more trips by the last arriver).
// Prints us per launch for grids G x B; slope over N = cost of one round trip, intercept = the launch itself.
// hipcc --offload-arch=gfx950 -O3 chain.hip -o chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void init(unsigned long long* buf, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = mix(i) % n;
}

template <int MODE>
__device__ __forceinline__ uint64_t trip(unsigned long long* buf, uint64_t at, uint64_t n) {
    if (MODE == 0) return buf[at];
    if (MODE == 1) return __hip_atomic_load(&buf[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) return atomicAdd(&buf[at], 0ull);
    const unsigned long long cur = buf[at];      // (not used: the CAS below is what is timed; compare value unlikely to match)
    (void)cur;
    return atomicCAS(&buf[at], 0xFFFFFFFFFFFFFFFFull, 0ull);
}

template <int MODE>
__global__ void __launch_bounds__(1024) chain(unsigned long long* buf, uint64_t n, int trips, int tail_trips, unsigned int* done, unsigned long long* sink, uint64_t salt, int lanes, int fence_all) {
    const int lane = threadIdx.x & 63;
    uint64_t at = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) ^ salt) % n;
    uint64_t acc = 0;
    if (lane < lanes) {
        for (int t = 0; t < trips; ++t) { at = trip<MODE>(buf, at, n) % n; acc += at; }
    }
    if (tail_trips >= 0) {
        __shared__ int last;
        if (fence_all) __threadfence();
        __syncthreads();      // every wave's memory operations are done (s_waitcnt vmcnt(0) in front of the barrier)
        if (threadIdx.x == 0) { if (fence_all == 0) __threadfence(); last = atomicAdd(done, 1u) == gridDim.x - 1; }
        __syncthreads();
        if (last) {
            if (threadIdx.x == 0) *done = 0;
            for (int t = 0; t < tail_trips; ++t) { at = __hip_atomic_load(&buf[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % n; acc += at; __syncthreads(); }
        }
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}

static int g_fence_all = 0, g_graph = 0;
template <int MODE>
static void run(const char* name, int G, int B, int trips, int tail, int lanes, unsigned long long* buf, uint64_t n, unsigned int* done, unsigned long long* sink) {
    const int reps = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(chain<MODE>, dim3(G), dim3(B), 0, 0, buf, n, trips, tail, done, sink, (uint64_t)i, lanes, g_fence_all);
    if (g_graph) {
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t gr; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(chain<MODE>, dim3(G), dim3(B), 0, st, buf, n, trips, tail, done, sink, (uint64_t)(i + 77), lanes, g_fence_all);
        hipStreamEndCapture(st, &gr);
        hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int i = 0; i < reps / 200; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(b, st); hipEventSynchronize(b);
        hipGraphExecDestroy(ge); hipGraphDestroy(gr); hipStreamDestroy(st);
    } else {
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(chain<MODE>, dim3(G), dim3(B), 0, 0, buf, n, trips, tail, done, sink, (uint64_t)(i + 77), lanes, g_fence_all);
    hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s%-10s G %4d x B %4d lanes %2d trips %2d tail %2d : %7.2f us per launch\n", g_graph ? "graph " : "eager ", name, G, B, lanes, trips, tail, ms * 1e3 / reps);
    fflush(stdout);
}

int main() {
    const uint64_t n = (uint64_t)1 << 28;      // 2 GiB of 8-byte words: no cache holds it
    unsigned long long *buf, *sink; unsigned int* done;
    hipMalloc(&buf, n * 8); hipMalloc(&sink, 64); hipMalloc(&done, 64); hipMemset(done, 0, 64);
    hipLaunchKernelGGL(init, dim3(4096), dim3(256), 0, 0, buf, n);
    hipDeviceSynchronize();
    for (g_graph = 0; g_graph < 2; ++g_graph) {
        for (int G : {1, 16, 32, 64}) for (int trips : {0, 4}) run<0>("load", G, 1024, trips, -1, 1, buf, n, done, sink);
        for (int G : {64, 256}) for (int trips : {0, 4}) run<0>("load", G, 256, trips, -1, 1, buf, n, done, sink);
        for (g_fence_all = 0; g_fence_all < 2; ++g_fence_all)
            for (int G : {16, 32, 64, 128}) for (int tail : {0, 4}) run<2>(g_fence_all ? "tail/fall" : "tail/f0", G, 1024, 4, tail, 1, buf, n, done, sink);
        g_fence_all = 0;
        for (int G : {64, 256}) for (int tail : {0, 4}) run<2>("tail/f0", G, 256, 4, tail, 1, buf, n, done, sink);
        for (int G : {16, 64}) run<2>("tail/f0/64", G, 1024, 4, 4, 64, buf, n, done, sink);
    }
    return 0;
}
