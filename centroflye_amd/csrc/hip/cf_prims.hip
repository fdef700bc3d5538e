// cf_prims.hip — device-wide exclusive scan and LSD radix sort (gfx950, wave64).
// Auxiliary primitives of the pipeline: CSR offsets (clouds, postings) and the ascending
// order of the rare k-mer set (reference: the k-mer file is written sorted,
// scripts/distance_based_kmer_recruitment.py:160-164).
#include "cf_common.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

// inclusive scan of one value per thread over a 256-thread block; returns exclusive prefix,
// *total = block sum.  lds: at least 8 int64.
__device__ __forceinline__ int64_t cf_block_exclusive(int64_t v, int64_t* lds, int64_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int64_t inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        int64_t o = __shfl_up(inc, (unsigned)d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int64_t wave_off = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int64_t s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

template <class TIn>
__global__ void __launch_bounds__(SCAN_THREADS)
cf_scan_local(const TIn* __restrict__ in, int64_t* __restrict__ out, int64_t* __restrict__ block_sums, int64_t n) {
    int64_t* lds = (int64_t*)cf_lds;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS];
    int64_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? (int64_t)in[base + i] : 0;
        sum += v[i];
    }
    int64_t total;
    int64_t pre = cf_block_exclusive(sum, lds, &total);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = pre;
        pre += v[i];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SCAN_THREADS)
cf_scan_add(int64_t* __restrict__ out, const int64_t* __restrict__ block_off, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    const int64_t off = block_off[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) out[base + i] += off;
}

template <class TIn>
static int scan_impl(cf_ctx* ctx, const TIn* d_in, int64_t* d_out, int64_t n, int64_t* total) {
    if (n <= 0) { if (total) *total = 0; return 0; }
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    int64_t* d_sums = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_sums, (size_t)nb + 1, "scan block sums"));
    hipLaunchKernelGGL((cf_scan_local<TIn>), dim3((unsigned)nb), dim3(SCAN_THREADS), 64, ctx->stream, d_in, d_out, d_sums, n);
    CF_KERNEL_CHECK("cf_scan_local");
    int rc = 0;
    int64_t tot = 0;
    if (nb > 1) {
        rc = scan_impl<int64_t>(ctx, d_sums, d_sums, nb, &tot);
        if (rc == 0) {
            hipLaunchKernelGGL(cf_scan_add, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, d_out, d_sums, n);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("launch of cf_scan_add: ") + hipGetErrorString(e));
        }
    } else {
        hipError_t e = hipMemcpyAsync(&tot, d_sums, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("scan total copy: ") + hipGetErrorString(e));
    }
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("scan sync: ") + hipGetErrorString(e));
    }
    cf_release_t(ctx, d_sums, (size_t)nb + 1);
    if (total) *total = tot;
    return rc;
}

int cf_scan_exclusive_i64(cf_ctx* ctx, const int64_t* d_in, int64_t* d_out, int64_t n, int64_t* total) {
    return scan_impl<int64_t>(ctx, d_in, d_out, n, total);
}
int cf_scan_exclusive_u32_to_i64(cf_ctx* ctx, const uint32_t* d_in, int64_t* d_out, int64_t n, int64_t* total) {
    return scan_impl<uint32_t>(ctx, d_in, d_out, n, total);
}

// ------------------------------------------------------------------ radix sort
#define RS_THREADS 256
#define RS_ITEMS 8
#define RS_TILE (RS_THREADS * RS_ITEMS)

__global__ void __launch_bounds__(RS_THREADS)
cf_radix_hist(const unsigned long long* __restrict__ in, uint32_t* __restrict__ hist, int64_t n, int shift, int ntiles) {
    uint32_t* h = (uint32_t*)cf_lds;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(in[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(RS_THREADS)
cf_radix_scatter(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                 const int64_t* __restrict__ offs, int64_t n, int shift, int ntiles) {
    int64_t* run = (int64_t*)cf_lds;                       // 256 running offsets, one per digit
    uint32_t* wcount = (uint32_t*)(cf_lds + 256 * 8);      // [4][256] per-wave digit counts
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    run[t] = offs[(int64_t)t * ntiles + blockIdx.x];
    for (int w = 0; w < 4; ++w) wcount[w * 256 + t] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int round = 0; round < RS_ITEMS; ++round) {
        const int64_t idx = base + (int64_t)round * RS_THREADS + t;
        const bool valid = idx < n;
        const unsigned long long key = valid ? in[idx] : 0ull;
        const uint32_t digit = (uint32_t)(key >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int bit = (digit >> b) & 1;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcount[wave * 256 + digit] = (uint32_t)__popcll(peers);
        __syncthreads();
        int64_t pos = 0;
        if (valid) {
            pos = run[digit] + rank;
            for (int w = 0; w < wave; ++w) pos += wcount[w * 256 + digit];
        }
        __syncthreads();
        {
            uint32_t s = 0;
            for (int w = 0; w < 4; ++w) { s += wcount[w * 256 + t]; wcount[w * 256 + t] = 0; }
            run[t] += s;
        }
        __syncthreads();
        if (valid) out[pos] = key;
    }
}

int cf_radix_sort_u64(cf_ctx* ctx, unsigned long long* d_keys, unsigned long long* d_tmp, int64_t n, int bits) {
    return cf_radix_sort_u64_any(ctx, d_keys, d_tmp, n, bits, nullptr);
}

// the same; with `result` the sorted keys stay in whichever of the two buffers the last pass wrote (no copy back) and
// *result tells which
int cf_radix_sort_u64_any(cf_ctx* ctx, unsigned long long* d_keys, unsigned long long* d_tmp, int64_t n, int bits, unsigned long long** result) {
    if (result) *result = d_keys;
    if (n <= 1) return 0;
    const int ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
    const int64_t nh = (int64_t)ntiles * 256;
    uint32_t* d_hist = nullptr;
    int64_t* d_offs = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_hist, (size_t)nh, "radix histogram"));
    int rc = cf_alloc_t(ctx, &d_offs, (size_t)nh, "radix offsets");
    unsigned long long* src = d_keys;
    unsigned long long* dst = d_tmp;
    for (int shift = 0; rc == 0 && shift < bits; shift += 8) {
        hipLaunchKernelGGL(cf_radix_hist, dim3((unsigned)ntiles), dim3(RS_THREADS), 256 * 4, ctx->stream, src, d_hist, n, shift, ntiles);
        rc = cf_scan_exclusive_u32_to_i64(ctx, d_hist, d_offs, nh, nullptr);
        if (rc) break;
        hipLaunchKernelGGL(cf_radix_scatter, dim3((unsigned)ntiles), dim3(RS_THREADS), 256 * 8 + 4 * 256 * 4, ctx->stream, src, dst, d_offs, n, shift, ntiles);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("radix launch: ") + hipGetErrorString(e)); break; }
        std::swap(src, dst);
    }
    if (rc == 0 && result) *result = src;
    if (rc == 0 && src != d_keys && !result) {
        hipError_t e = hipMemcpyAsync(d_keys, src, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("radix copy back: ") + hipGetErrorString(e));
    }
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("radix sync: ") + hipGetErrorString(e));
    }
    if (d_offs) cf_release_t(ctx, d_offs, (size_t)nh);
    cf_release_t(ctx, d_hist, (size_t)nh);
    return rc;
}

// ------------------------------------------------------------------ radix sort of 16-byte records by 32-bit fields
// LSD passes of 8 bits over the fields the caller lists (least significant first): the same histogram / scan / stable
// ballot-ranked scatter as above, with a record = four 32-bit words and the digit taken from word `word`.
struct __attribute__((aligned(16))) cf_rec16 { uint32_t w[4]; };

__global__ void __launch_bounds__(RS_THREADS)
cf_rec16_hist(const cf_rec16* __restrict__ in, uint32_t* __restrict__ hist, int64_t n, int word, int shift, int ntiles) {
    uint32_t* h = (uint32_t*)cf_lds;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(((const uint32_t*)(in + idx))[word] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(RS_THREADS)
cf_rec16_scatter(const cf_rec16* __restrict__ in, cf_rec16* __restrict__ out, const int64_t* __restrict__ offs, int64_t n, int word, int shift, int ntiles) {
    int64_t* run = (int64_t*)cf_lds;                       // 256 running offsets, one per digit
    uint32_t* wcount = (uint32_t*)(cf_lds + 256 * 8);      // [4][256] per-wave digit counts
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    run[t] = offs[(int64_t)t * ntiles + blockIdx.x];
    for (int w = 0; w < 4; ++w) wcount[w * 256 + t] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int round = 0; round < RS_ITEMS; ++round) {
        const int64_t idx = base + (int64_t)round * RS_THREADS + t;
        const bool valid = idx < n;
        cf_rec16 rec{{0u, 0u, 0u, 0u}};
        if (valid) rec = in[idx];
        const uint32_t digit = (rec.w[word] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int bit = (digit >> b) & 1;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcount[wave * 256 + digit] = (uint32_t)__popcll(peers);
        __syncthreads();
        int64_t pos = 0;
        if (valid) {
            pos = run[digit] + rank;
            for (int w = 0; w < wave; ++w) pos += wcount[w * 256 + digit];
        }
        __syncthreads();
        {
            uint32_t sum = 0;
            for (int w = 0; w < 4; ++w) { sum += wcount[w * 256 + t]; wcount[w * 256 + t] = 0; }
            run[t] += sum;
        }
        __syncthreads();
        if (valid) out[pos] = rec;
    }
}

// d_recs: n records of 16 bytes; d_tmp: scratch of the same size; fields: n_fields (word index, significant bits), least
// significant field first.  The sorted records end in d_recs.
int cf_radix_sort_rec16(cf_ctx* ctx, void* d_recs, void* d_tmp, int64_t n, const int* words, const int* bits, int n_fields) {
    if (n <= 1) return 0;
    const int ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
    const int64_t nh = (int64_t)ntiles * 256;
    uint32_t* d_hist = nullptr;
    int64_t* d_offs = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_hist, (size_t)nh, "radix histogram"));
    int rc = cf_alloc_t(ctx, &d_offs, (size_t)nh, "radix offsets");
    cf_rec16* src = (cf_rec16*)d_recs;
    cf_rec16* dst = (cf_rec16*)d_tmp;
    for (int f = 0; rc == 0 && f < n_fields; ++f) {
        for (int shift = 0; rc == 0 && shift < bits[f]; shift += 8) {
            hipLaunchKernelGGL(cf_rec16_hist, dim3((unsigned)ntiles), dim3(RS_THREADS), 256 * 4, ctx->stream, (const cf_rec16*)src, d_hist, n, words[f], shift, ntiles);
            rc = cf_scan_exclusive_u32_to_i64(ctx, d_hist, d_offs, nh, nullptr);
            if (rc) break;
            hipLaunchKernelGGL(cf_rec16_scatter, dim3((unsigned)ntiles), dim3(RS_THREADS), 256 * 8 + 4 * 256 * 4, ctx->stream, (const cf_rec16*)src, dst, (const int64_t*)d_offs, n, words[f], shift, ntiles);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("record sort launch: ") + hipGetErrorString(e)); break; }
            std::swap(src, dst);
        }
    }
    if (rc == 0 && src != (cf_rec16*)d_recs) {
        hipError_t e = hipMemcpyAsync(d_recs, src, (size_t)n * 16, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("record sort copy back: ") + hipGetErrorString(e));
    }
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("record sort sync: ") + hipGetErrorString(e));
    }
    if (d_offs) cf_release_t(ctx, d_offs, (size_t)nh);
    cf_release_t(ctx, d_hist, (size_t)nh);
    return rc;
}
