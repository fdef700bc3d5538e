// Microbenchmark (developer tool): what does a CU sustain on the instruction mix of
// cf_dist_kernel's sketch sweep — per pair: v_sub, v_lshr, v_mul_u32_u24, v_mad_u32_u24, v_lshr, v_lshl, v_lshl, v_and,
// (ds_add_rtn_u32), v_bfe, v_cmp = 11 vector instructions + one returning LDS atomic on a random 8-bit counter, four pairs per
// step, plus SALU instructions per step — at 1, 2, 4 waves per SIMD?  Prints wave-instructions per cycle per CU (VALU
// ceiling: 4 SIMDs x 1 wave64 instruction per 2 cycles = 2.0) next to what the kernel itself reaches
// (SQ_INSTS_VALU / (kernel time x clock x CUs)).  hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

// SALU per step: 3 x S dependent scalar instructions on a uniform value (the kernel runs 15 - 60 scalar instructions per step)
template <int ATOMIC, int S>
__global__ void k(int steps, uint32_t ig, uint32_t sk_shift, uint32_t min_cov_m1, unsigned long long* out) {
    uint32_t* sk = (uint32_t*)lds;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) sk[i] = 0;
    __syncthreads();
    uint32_t x0 = blockIdx.x * 1024u + threadIdx.x * 2654435761u, acc = 0;
    uint32_t su = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x + 1));
    for (int s = 0; s < steps; ++s) {
        uint32_t raw[4], old[4], sft[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x0 = (x0 ^ (x0 << 7)) + 0x9E3779B9u; raw[u] = x0 ^ (x0 >> 9); }       // stands in for the global load (4 full-rate VALU per pair)
#pragma unroll
        for (int j = 0; j < S; ++j) su = (su ^ (su >> 3)) + 0x9E37u + (uint32_t)j;           // s_lshr_b32 + s_xor_b32 + s_add_i32 (a dependent chain, as the kernel's scalar code mostly is)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t q = raw[u] - (ig << 24);
            const uint32_t h = (raw[u] & 0xFFFFFFu) * 0x9E3779u + (q >> 24) * 0x5BD1E9u;
            const uint32_t idx = h >> sk_shift;
            sft[u] = idx << 3;
            const uint32_t inc = 1u << (sft[u] & 31u);
            if (ATOMIC) old[u] = atomicAdd(&sk[idx >> 2], inc); else old[u] = idx + inc;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t seen = __builtin_amdgcn_ubfe(old[u], sft[u], 8u);
            if (seen >= min_cov_m1 + 250u) acc += seen;      // (never taken with 8-bit counters below 250)
        }
        ig = (ig + su) & 0xFFu;
    }
    if (acc == 0x12345u || threadIdx.x == 0) out[blockIdx.x] = acc + su;
}

template <int ATOMIC, int S>
void run(const char* name, int block, unsigned long long* d, int per_cu = 1) {
    const int steps = 20000, grid = 256 * per_cu;      // per_cu workgroups per CU: per_cu x block / 256 waves per SIMD
    const int lds_bytes = per_cu == 1 ? 100 * 1024 : 72 * 1024;      // > 80 KiB: one workgroup per CU; 72 KiB: two
    hipFuncSetAttribute((const void*)k<ATOMIC, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<ATOMIC, S>), dim3(grid), dim3(block), lds_bytes, 0, steps, 3u, 16u, 3u, d);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<ATOMIC, S>), dim3(grid), dim3(block), lds_bytes, 0, steps, 3u, 16u, 3u, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves = (double)block / 64 * per_cu, valu = 4 * (4 + 11) + 3, pairs = 4.0 * steps * 64 * waves;      // per CU
    printf("%-34s waves/SIMD %g  %8.3f ms  %6.1f ns/step/wave  VALU wave-instr/cycle/CU @2.4GHz %.3f  pairs/s chip %.3e\n", name, waves / 4, ms,
           ms * 1e6 / steps, valu * steps * waves / (ms * 1e-3 * 2.4e9), pairs * grid / (ms * 1e-3));
}
int main() {
    unsigned long long* d; hipMalloc(&d, 512 * 8);
    for (int block : {256, 512, 1024}) {
        run<0, 0>("VALU only", block, d);
        run<1, 0>("VALU + 4 ds_add_rtn_u32", block, d);
        run<1, 5>("VALU + 4 ds_add_rtn + 15 SALU", block, d);
        run<1, 11>("VALU + 4 ds_add_rtn + 33 SALU", block, d);
        run<1, 21>("VALU + 4 ds_add_rtn + 63 SALU", block, d);
    }
    run<0, 0>("VALU only (2 x 1024)", 1024, d, 2);
    run<1, 5>("VALU + 4 ds_add_rtn + 15 SALU (2x)", 1024, d, 2);
    return 0;
}
