// cf_checksum.hip — order-independent checksums of the results that are resident in HBM (A1 table, A2 rare set, A3 clouds,
// A6 unique k-mers), computed where the data is.  The figures are the ones the plain-C oracle reports
// (oracle/c/cf_oracle_mt.c: cfo_table_mix / cfo_key_mix / cfo_cloud_mix, sums mod 2^64), so a full-size parity check — 1.3e9
// table entries, 6.5e8 cloud entries at BASELINE configs[3] — compares every element without copying tens of GB to the host
// and hashing them there.  Reference: distance_based_kmer_recruitment.py:39-63 (table), :66-82 (rare set), :145-148
// (unique k-mers), read_kmer_cloud.py:17-40 (clouds).
#include "cf_common.h"

namespace {

__device__ __forceinline__ void cf_sum_out(unsigned long long s, unsigned long long c, unsigned long long* __restrict__ out) {
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_down(s, (unsigned)d); c += __shfl_down(c, (unsigned)d); }
    if ((threadIdx.x & 63) == 0 && c) { atomicAdd(out, s); atomicAdd(out + 1, c); }
}

// A1: every occupied slot (key, pres, multi) of the table — dense array or open-addressed, the scan is the same
__global__ void __launch_bounds__(256)
cf_cs_table_kernel(const cf_slot* __restrict__ tab, uint64_t cap, unsigned long long* __restrict__ out) {
    unsigned long long s = 0, c = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) {
        const cf_slot sl = tab[i];
        if (sl.key & CF_OCC) {
            const unsigned long long key = sl.key & ~CF_OCC, pres = sl.val & 0xFFFFFFFFull, multi = sl.val >> 32;
            s += cf_mix64(cf_mix64(cf_mix64(key + 0x7AB1Eull) ^ pres) ^ (multi << 1));
            ++c;
        }
    }
    cf_sum_out(s, c, out);
}

// A2 / A6: the k-mers of the set (bits == nullptr) or those whose bit is set in the unique bitmap
__global__ void __launch_bounds__(256)
cf_cs_kmers_kernel(const unsigned long long* __restrict__ kmers, int64_t n, const uint32_t* __restrict__ bits, unsigned long long* __restrict__ out) {
    unsigned long long s = 0, c = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (bits && !((bits[i >> 5] >> (i & 31)) & 1u)) continue;
        s += cf_mix64(kmers[i] ^ 0xABCDEFull);
        ++c;
    }
    cf_sum_out(s, c, out);
}

// A3: one wave per unit, lanes over the unit's entries
__global__ void __launch_bounds__(256)
cf_cs_clouds_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t n_units, unsigned long long* __restrict__ out) {
    unsigned long long s = 0, c = 0;
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t u = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < n_units; u += n_waves) {
        const int64_t b = cloud_ptr[u], e = cloud_ptr[u + 1];
        const unsigned long long hu = cf_mix64((unsigned long long)u + 0x51EDull);
        for (int64_t i = b + lane; i < e; i += 64) { s += cf_mix64(hu ^ (unsigned long long)(uint32_t)entries[i]); ++c; }
    }
    cf_sum_out(s, c, out);
}

}  // namespace

extern "C" int cf_checksum(cf_ctx* ctx, int32_t what, uint64_t* sum, int64_t* n_items) {
    if (!ctx || !sum) return -22;
    *sum = 0;
    if (n_items) *n_items = 0;
    CF_HIP(hipSetDevice(ctx->device));
    unsigned long long* d_out = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_out, 2, "checksum"));
    int rc = 0;
    unsigned long long h[2] = {0, 0};
    const int grid_max = std::max(1, ctx->n_cu) * 16;
    do {
        if (hipMemsetAsync(d_out, 0, 16, ctx->stream) != hipSuccess) { rc = cf_fail(ctx, -5, "cf_checksum: memset"); break; }
        if (what == CF_CHECKSUM_TABLE) {
            if (!ctx->d_table) { rc = cf_fail(ctx, -22, "cf_checksum: no table (call cf_count_kmers first)"); break; }
            hipLaunchKernelGGL(cf_cs_table_kernel, dim3((unsigned)cf_grid_for((int64_t)ctx->table_cap, 256, grid_max)), dim3(256), 0, ctx->stream,
                               (const cf_slot*)ctx->d_table, (uint64_t)ctx->table_cap, d_out);
        } else if (what == CF_CHECKSUM_KMERS || what == CF_CHECKSUM_UNIQUE) {
            if (what == CF_CHECKSUM_UNIQUE && !ctx->d_unique_bits) { rc = cf_fail(ctx, -22, "cf_checksum: no k-mer set"); break; }
            if (ctx->n_kmers > 0)
                hipLaunchKernelGGL(cf_cs_kmers_kernel, dim3((unsigned)cf_grid_for(ctx->n_kmers, 256, grid_max)), dim3(256), 0, ctx->stream,
                                   (const unsigned long long*)ctx->d_kmers, ctx->n_kmers,
                                   what == CF_CHECKSUM_UNIQUE ? (const uint32_t*)ctx->d_unique_bits : (const uint32_t*)nullptr, d_out);
        } else if (what == CF_CHECKSUM_CLOUDS) {
            if (!ctx->have_clouds) { rc = cf_fail(ctx, -22, "cf_checksum: no clouds built"); break; }
            if (ctx->n_units > 0)
                hipLaunchKernelGGL(cf_cs_clouds_kernel, dim3((unsigned)cf_grid_for(ctx->n_units, 4, grid_max)), dim3(256), 0, ctx->stream,
                                   (const int64_t*)ctx->d_cloud_ptr, (const int32_t*)ctx->d_entries, ctx->n_units, d_out);
        } else { rc = cf_fail(ctx, -22, "cf_checksum: unknown selector"); break; }
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cf_fail(ctx, -5, "cf_checksum: kernel");
    } while (0);
    cf_release_t(ctx, d_out, 2);
    if (rc) return rc;
    *sum = (uint64_t)h[0];
    if (n_items) *n_items = (int64_t)h[1];
    return 0;
}
