// Microbenchmark (developer tool): throughput of random LDS atomics on gfx950, one 1024-thread workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(int iters, int slots, unsigned long long* out) {
    unsigned long long* t64 = (unsigned long long*)lds;
    uint32_t* t32 = (uint32_t*)lds;
    for (int i = threadIdx.x; i < slots * 2; i += blockDim.x) t32[i] = 0;
    __syncthreads();
    uint32_t s = mix(blockIdx.x * 1024 + threadIdx.x + 1);
    unsigned long long acc = 0;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t h = (uint32_t)(((unsigned long long)mix(s) * (unsigned long long)slots) >> 32);
        if (MODE == 0) atomicAdd(&t64[h], 1ull);                        // ds_add_u64
        if (MODE == 1) atomicAdd(&t32[h], 1u);                          // ds_add_u32
        if (MODE == 2) acc += atomicAdd(&t64[h], 1ull);                 // ds_add_rtn_u64
        if (MODE == 3) acc += atomicAdd(&t32[h], 1u);                   // ds_add_rtn_u32
        if (MODE == 4) acc += atomicCAS(&t64[h], 0ull, (unsigned long long)s | 1);   // ds_cmpst_rtn_b64
        if (MODE == 5) acc += atomicCAS(&t32[h], 0u, s | 1);            // ds_cmpst_rtn_b32
        if (MODE == 6) acc += t64[h];                                   // ds_read_b64
        if (MODE == 7) { t32[h] += 1; }                                 // non-atomic read-modify-write b32
        if (MODE == 8) acc += h;                                        // no LDS: loop overhead
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + t64[1];
}
template <int MODE> void run(const char* name, unsigned long long* d) {
    const int iters = 4096, slots = 19200, grid = 256, block = 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, slots * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), slots * 8, 0, iters, slots, d);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), slots * 8, 0, iters, slots, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)iters * grid * block;
    printf("%-22s %8.3f ms  %7.2f Gops/s chip  %6.2f cycles/op/CU @2.4GHz\n", name, ms, ops / ms / 1e6, ms * 1e-3 * 2.4e9 / (ops / grid));
}
int main() {
    unsigned long long* d; hipMalloc(&d, 256 * 8);
    run<8>("loop only", d); run<0>("ds_add_u64", d); run<1>("ds_add_u32", d); run<2>("ds_add_rtn_u64", d); run<3>("ds_add_rtn_u32", d);
    run<4>("ds_cmpst_rtn_b64", d); run<5>("ds_cmpst_rtn_b32", d); run<6>("ds_read_b64", d); run<7>("rmw b32 (no atomic)", d);
    return 0;
}
