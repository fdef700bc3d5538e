// Microbenchmark (developer tool, round 6): is a dependent memory round trip cheaper when every workgroup of a launch sits on ONE XCD?
// MI355X has 8 XCDs with an L2 each; L2s are not coherent with each other, so what crosses workgroups inside a launch goes past them
// (agent scope, sc1: 1.1 us per atomic under the placement kernel's load).  A stream created with hipExtStreamCreateWithCUMask can
// confine its kernels to the 32 CUs of one XCD; there the L2 IS the point of coherence for everything the launch does, and
// workgroup-scope operations (performed in that L2) are enough on this hardware.
//   xcd_local <mask stride>      bits 0, s, 2 s, ... of the CU mask are set (32 of them); prints which XCDs the workgroups ran on
//                                 (HW_REG_XCC_ID) and us per launch / per dependent round trip for loads and atomics of either scope
// hipcc --offload-arch=gfx950 -O3 xcd_local.hip -o xcd_local
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void init(unsigned long long* buf, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = mix(i) % n;
}
__global__ void where(unsigned int* xcc) {
    unsigned int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = id;
}
// MODE 0 plain load, 1 agent-scope load, 2 agent-scope returning add, 3 workgroup-scope load (sc0), 4 workgroup-scope returning add
template <int MODE>
__device__ __forceinline__ uint64_t trip(unsigned long long* buf, uint64_t at) {
    if (MODE == 0) return buf[at];
    if (MODE == 1) return __hip_atomic_load(&buf[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) return __hip_atomic_fetch_add(&buf[at], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 3) return __hip_atomic_load(&buf[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_fetch_add(&buf[at], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int MODE>
__global__ void __launch_bounds__(1024) chain(unsigned long long* buf, uint64_t n, int trips, unsigned long long* sink, uint64_t salt, int lanes) {
    const int lane = threadIdx.x & 63;
    uint64_t at = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) ^ salt) % n, acc = 0;
    if (lane < lanes) for (int t = 0; t < trips; ++t) { at = trip<MODE>(buf, at) % n; acc += at; }
    if (acc == 0x123456789ull) sink[0] = acc;
}
template <int MODE>
static double run(hipStream_t st, int G, int B, int trips, int lanes, unsigned long long* buf, uint64_t n, unsigned long long* sink) {
    const int reps = 1000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(chain<MODE>, dim3(G), dim3(B), 0, st, buf, n, trips, sink, (uint64_t)i, lanes);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(chain<MODE>, dim3(G), dim3(B), 0, st, buf, n, trips, sink, (uint64_t)(i + 100), lanes);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return 1e3 * ms / reps;
}
int main(int argc, char** argv) {
    const int stride = argc > 1 ? atoi(argv[1]) : 8;
    const uint64_t n = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 27);      // 1 GiB of 8-byte words by default: no cache holds it
    unsigned long long *buf, *sink; unsigned int* xcc;
    hipMalloc(&buf, n * 8); hipMalloc(&sink, 8); hipMalloc(&xcc, 4096 * 4);
    hipLaunchKernelGGL(init, dim3(4096), dim3(256), 0, 0, buf, n); hipDeviceSynchronize();
    hipStream_t plain, masked;
    hipStreamCreate(&plain);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int set = 0;
    for (int i = 0; i < 256 && set < 32; i += stride) { mask[i >> 5] |= 1u << (i & 31); ++set; }
    hipError_t e = hipExtStreamCreateWithCUMask(&masked, 8, mask);
    printf("hipExtStreamCreateWithCUMask(stride %d, %d CUs): %s\n", stride, set, hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    for (int which = 0; which < 2; ++which) {
        hipStream_t st = which ? masked : plain;
        hipMemsetAsync(xcc, 0xFF, 4096 * 4, st);
        hipLaunchKernelGGL(where, dim3(512), dim3(64), 0, st, xcc); hipStreamSynchronize(st);
        std::vector<unsigned int> h(512); hipMemcpy(h.data(), xcc, 512 * 4, hipMemcpyDeviceToHost);
        int cnt[16] = {0}; for (auto v : h) if ((v & 15) < 16) ++cnt[v & 15];
        printf("%s stream: workgroups per XCC id:", which ? "masked" : "plain ");
        for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
        printf("\n");
    }
    const char* names[5] = {"plain load", "agent load", "agent add rtn", "wg-scope load", "wg-scope add rtn"};
    for (int which = 0; which < 2; ++which) {
        hipStream_t st = which ? masked : plain;
        for (int G : {32, 128}) for (int lanes : {64}) {
            printf("%s stream, %3d workgroups x 1024 threads:\n", which ? "masked" : "plain ", G);
            for (int mode = 0; mode < 5; ++mode) {
                double t1, t9;
                switch (mode) {
                    case 0: t1 = run<0>(st, G, 1024, 1, lanes, buf, n, sink); t9 = run<0>(st, G, 1024, 9, lanes, buf, n, sink); break;
                    case 1: t1 = run<1>(st, G, 1024, 1, lanes, buf, n, sink); t9 = run<1>(st, G, 1024, 9, lanes, buf, n, sink); break;
                    case 2: t1 = run<2>(st, G, 1024, 1, lanes, buf, n, sink); t9 = run<2>(st, G, 1024, 9, lanes, buf, n, sink); break;
                    case 3: t1 = run<3>(st, G, 1024, 1, lanes, buf, n, sink); t9 = run<3>(st, G, 1024, 9, lanes, buf, n, sink); break;
                    default: t1 = run<4>(st, G, 1024, 1, lanes, buf, n, sink); t9 = run<4>(st, G, 1024, 9, lanes, buf, n, sink); break;
                }
                printf("   %-18s launch with 1 trip %6.2f us, 9 trips %6.2f us -> %5.2f us per round trip, %5.2f us per empty launch\n", names[mode], t1, t9, (t9 - t1) / 8.0, t1 - (t9 - t1) / 8.0);
            }
        }
    }
    // the same on a SMALL array that one L2 holds (4 MiB): what a dependent step costs when its line is in the XCD's own L2
    const uint64_t n2 = 1ull << 18;      // 2 MiB
    hipLaunchKernelGGL(init, dim3(1024), dim3(256), 0, 0, buf, n2); hipDeviceSynchronize();
    for (int which = 0; which < 2; ++which) {
        hipStream_t st = which ? masked : plain;
        printf("%s stream, 32 workgroups, 2 MiB array:\n", which ? "masked" : "plain ");
        printf("   agent load %5.2f  agent add %5.2f  wg load %5.2f  wg add %5.2f us per round trip\n",
               (run<1>(st, 32, 1024, 9, 64, buf, n2, sink) - run<1>(st, 32, 1024, 1, 64, buf, n2, sink)) / 8, (run<2>(st, 32, 1024, 9, 64, buf, n2, sink) - run<2>(st, 32, 1024, 1, 64, buf, n2, sink)) / 8,
               (run<3>(st, 32, 1024, 9, 64, buf, n2, sink) - run<3>(st, 32, 1024, 1, 64, buf, n2, sink)) / 8, (run<4>(st, 32, 1024, 9, 64, buf, n2, sink) - run<4>(st, 32, 1024, 1, 64, buf, n2, sink)) / 8);
    }
    return 0;
}
