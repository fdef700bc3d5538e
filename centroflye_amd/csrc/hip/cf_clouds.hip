// cf_clouds.hip — k-mer set installation, A3 per-unit clouds, A4 multiplicity filter, unique mask.
//
// Reference:
//   A3 scripts/read_kmer_cloud.py:17-40: for each HOR unit of each read, the SET of k-mers of the
//      unit's de-gapped row (windows inside the unit only) that belong to the given k-mer set.
//   A4 scripts/read_kmer_cloud.py:43-54: keep k-mers present in >= min_mult (<= max_mult) clouds.
//
// Device design: one 256-thread workgroup per unit.  The unit's bases are staged through LDS
// (coalesced byte loads, sliding window), every window is looked up in an open-addressed HBM
// table of the k-mer set (key -> rank), hits are de-duplicated in an LDS set, compacted and
// bitonic-sorted in LDS, and written as one CSR row.  Two launches: sizes, then rows.
#include "cf_common.h"

void cf_free_kmers(cf_ctx* c);
void cf_free_clouds(cf_ctx* c);
void cf_free_gview(cf_ctx* c);

#define CL_THREADS 256
#define CL_TILE_W 8
#define CL_EMPTY 0xFFFFFFFFu
struct alignas(16) cf_u32x4_cl { uint32_t x, y, z, w; };

// Hashes of the lookup: one 32-bit multiply for the fold and one for each of the two words (cf_mix64 — two 64 x 64
// multiplies, sixteen quarter-rate 32-bit ones — was a quarter of the cloud kernel's instructions).  h1 picks the slot,
// h2 the prefilter word (its high bits) and the two bits tested in it (its low 10 bits).
struct cf_lut_hash { uint32_t h1, h2; };
__device__ __forceinline__ cf_lut_hash cf_lut_hash_of(unsigned long long code) {
    const uint32_t lo = (uint32_t)code, hi = (uint32_t)(code >> 32);
    uint32_t f = lo ^ (hi * 0x85EBCA6Bu) ^ (hi >> 7);
    f ^= f >> 15;
    uint32_t h1 = f * 0x9E3779B1u, h2 = (f ^ 0x5BD1E995u) * 0xC2B2AE35u;
    h1 ^= h1 >> 16; h2 ^= h2 >> 13;
    return cf_lut_hash{h1, h2};
}
// The set holds a small share of all k-mers (1 window in 9 of a HOR read is a rare k-mer): a Bloom word — 8 bits per
// lookup slot (16 MB for 7.5 M k-mers: Infinity-Cache resident), two bits per k-mer inside ONE 32-bit word, 1.5 % false
// positives (round 2, one bit: 6 %) — answers most windows without touching the lookup table.
__device__ __forceinline__ uint64_t cf_lut_pre_word(cf_lut_hash h, uint64_t pre_words_mask) { return ((uint64_t)(h.h2 >> 10) | ((uint64_t)(h.h1 >> 20) << 22)) & pre_words_mask; }
__device__ __forceinline__ uint32_t cf_lut_pre_bits(uint32_t h2) { return (1u << (h2 & 31u)) | (1u << ((h2 >> 5) & 31u)); }

// One 16-byte slot {k-mer | CF_OCC, rank} per entry: key and rank arrive with ONE random HBM access (round 2 kept the ranks in
// an array of their own: a second, dependent round trip and as many 64-byte sectors again for every window that is in the set)
__global__ void __launch_bounds__(256)
cf_lut_build_kernel(const unsigned long long* __restrict__ kmers, int64_t n, cf_slot* __restrict__ lut, uint64_t mask, uint32_t* __restrict__ pre,
                    uint64_t pre_words_mask, unsigned int* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long want = kmers[i] | CF_OCC;
        const cf_lut_hash hh = cf_lut_hash_of(kmers[i]);
        uint64_t h = hh.h1 & mask;
        atomicOr(&pre[cf_lut_pre_word(hh, pre_words_mask)], cf_lut_pre_bits(hh.h2));
        bool done = false;
        for (uint64_t probe = 0; probe <= mask; ++probe) {
            const unsigned long long cur = atomicCAS(&lut[h].key, 0ull, want);
            if (cur == 0ull) { lut[h].val = (unsigned long long)i; done = true; break; }
            if (cur == want) { atomicOr(flags, 2u); done = true; break; }  // duplicate in the set
            h = (h + 1) & mask;
        }
        if (!done) atomicOr(flags, 1u);
        if (i > 0 && kmers[i - 1] >= kmers[i]) atomicOr(flags, 4u);      // not sorted ascending
    }
}

// mode 0: sizes[u] = |cloud(u)|; mode 1: entries[cloud_ptr[u] ...] = sorted cloud; mode 2 (one pass over the reads instead
// of two): sizes[u] AND the sorted cloud at entries[u * row_stride ...] of a scratch buffer, compacted afterwards
__global__ void __launch_bounds__(CL_THREADS)
cf_cloud_kernel(const uint8_t* __restrict__ bases, const int64_t* __restrict__ unit_start, const int64_t* __restrict__ unit_end,
                int64_t n_units, int k, int CL_SET /* LDS set slots, power of two */, const cf_slot* __restrict__ lut,
                uint64_t lut_mask, const uint32_t* __restrict__ lut_pre, uint64_t lut_pre_mask /* words - 1 */, int mode, int64_t row_stride, uint32_t* __restrict__ sizes, const int64_t* __restrict__ cloud_ptr,
                int32_t* __restrict__ entries, unsigned int* __restrict__ flags) {
    uint32_t* set = (uint32_t*)cf_lds;                         // CL_SET
    uint32_t* list = set + CL_SET;                             // CL_SET
    uint8_t* stage = (uint8_t*)(list + CL_SET);                // CL_THREADS * CL_TILE_W + 64
    unsigned int* counters = (unsigned int*)(stage + CL_THREADS * CL_TILE_W + 64);
    const int t = threadIdx.x;
    const unsigned long long kmask = (1ull << (2 * k)) - 1ull;
    const int tile = CL_THREADS * CL_TILE_W;
    for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int64_t b0 = unit_start[u], b1 = unit_end[u];
        const int64_t n_win = b1 - b0 - k + 1;
        for (int s = t; s < CL_SET; s += CL_THREADS) set[s] = CL_EMPTY;
        if (t == 0) { counters[0] = 0; counters[1] = 0; counters[2] = 0; }
        __syncthreads();
        uint32_t mine_new = 0;      // set slots this thread created (summed per wave: one LDS atomic instead of one per slot)
        for (int64_t w0 = 0; w0 < n_win; w0 += tile) {
            const int64_t nb = min((int64_t)tile + k - 1, b1 - b0 - w0);
            for (int64_t i = t; i < nb; i += CL_THREADS) stage[i] = bases[b0 + w0 + i];
            __syncthreads();
            const int64_t my0 = (int64_t)t * CL_TILE_W;
            const int64_t my_n = min((int64_t)CL_TILE_W, n_win - w0 - my0);
            if (my_n > 0) {
                // the unit's row is upper-cased first (reference read_kmer_cloud.py:25): the code ignores the case, and a window
                // holding anything but A, C, G, T (either case) can match no k-mer of the 2-bit set: skipped.
                // The thread's CL_TILE_W windows go through the lookup TOGETHER: all prefilter words are requested before the
                // first is looked at, then the table slots of the windows that passed, probe chains in lockstep — two or three
                // HBM round trips per tile instead of one or two per window, 8 - 10 in a row (19.4 -> 18.4 ms at 0.99 Gb: the kernel is
                // bound by the rate of random prefilter words, 5.5e10 per second, not by a thread's chain of them).
                unsigned long long codes[CL_TILE_W];
                uint32_t live = 0;           // bit i: window i exists and holds only A, C, G, T
                {
                    unsigned long long code = 0;
                    int run = 0;             // valid bases in a row, ending at the current one
                    for (int j = 0; j < k - 1; ++j) { const uint32_t c = stage[my0 + j]; code = (code << 2) | cf_base2(c); run = cf_is_acgt_nocase(c) ? run + 1 : 0; }
#pragma unroll
                    for (int i = 0; i < CL_TILE_W; ++i) {
                        codes[i] = 0;
                        if (i < my_n) {
                            const uint32_t c = stage[my0 + i + k - 1];
                            code = ((code << 2) | cf_base2(c)) & kmask;
                            run = cf_is_acgt_nocase(c) ? run + 1 : 0;
                            codes[i] = code;
                            if (run >= k) live |= 1u << i;
                        }
                    }
                }
                cf_lut_hash hh[CL_TILE_W];
                uint32_t pw[CL_TILE_W];
#pragma unroll
                for (int i = 0; i < CL_TILE_W; ++i) {
                    hh[i] = cf_lut_hash_of(codes[i]);
                    pw[i] = ((live >> i) & 1u) ? lut_pre[cf_lut_pre_word(hh[i], lut_pre_mask)] : 0u;
                }
                uint32_t act = 0;            // windows still walking their probe chain
                uint64_t hs[CL_TILE_W];
#pragma unroll
                for (int i = 0; i < CL_TILE_W; ++i) {
                    const uint32_t need = cf_lut_pre_bits(hh[i].h2);
                    if ((pw[i] & need) == need) act |= 1u << i;      // (a window that is not live has pw = 0)
                    hs[i] = hh[i].h1 & lut_mask;
                }
                uint32_t found = 0, idx[CL_TILE_W];
                for (uint64_t probe = 0; act && probe <= lut_mask; ++probe) {
                    cf_slot cur[CL_TILE_W];
#pragma unroll
                    for (int i = 0; i < CL_TILE_W; ++i) { cur[i].key = 0ull; cur[i].val = 0ull; if ((act >> i) & 1u) cur[i] = lut[hs[i]]; }
#pragma unroll
                    for (int i = 0; i < CL_TILE_W; ++i) {
                        if (!((act >> i) & 1u)) continue;
                        if (cur[i].key == (codes[i] | CF_OCC)) { idx[i] = (uint32_t)cur[i].val; found |= 1u << i; act &= ~(1u << i); }
                        else if (cur[i].key == 0ull) act &= ~(1u << i);
                        else hs[i] = (hs[i] + 1) & lut_mask;
                    }
                }
#pragma unroll
                for (int i = 0; i < CL_TILE_W; ++i) {
                    if (!((found >> i) & 1u)) continue;
                    uint32_t h = cf_mix32(idx[i]) & (CL_SET - 1);
                    bool done = false;
                    for (int probe = 0; probe < CL_SET; ++probe) {
                        uint32_t cur = set[h];
                        if (cur == CL_EMPTY) {
                            cur = atomicCAS(&set[h], CL_EMPTY, idx[i]);
                            if (cur == CL_EMPTY) { ++mine_new; done = true; break; }
                        }
                        if (cur == idx[i]) { done = true; break; }
                        h = (h + 1) & (CL_SET - 1);
                    }
                    if (!done) counters[1] = 1;
                }
            }
            __syncthreads();
        }
        for (int d = 32; d >= 1; d >>= 1) mine_new += __shfl_down(mine_new, (unsigned)d);
        if ((t & 63) == 0 && mine_new) atomicAdd(&counters[0], mine_new);
        __syncthreads();
        const uint32_t cnt = counters[0];
        if (counters[1] || cnt > CL_SET * 3 / 4) {
            if (t == 0) atomicOr(flags, 1u);
        } else if (mode == 0) {
            if (t == 0) sizes[u] = cnt;
        } else if (cnt > 0) {
            if (mode == 2 && t == 0) sizes[u] = cnt;
            for (int s0 = 0; s0 < CL_SET; s0 += CL_THREADS) {      // (uniform trip count: a ballot and ONE LDS atomic per wave and round)
                const uint32_t v = set[s0 + t];
                const unsigned long long m = __ballot(v != CL_EMPTY);
                if (m) {
                    uint32_t base = 0;
                    if ((t & 63) == 0) base = atomicAdd(&counters[2], (uint32_t)__popcll(m));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (v != CL_EMPTY) list[base + (uint32_t)__popcll(m & ((1ull << (t & 63)) - 1ull))] = v;
                }
            }
            uint32_t n2 = 1;
            while (n2 < cnt) n2 <<= 1;
            __syncthreads();
            if (cnt <= 2 * CL_THREADS) {
                // a cloud of a few hundred ranks, all different: an entry's place is the number of smaller ones — the whole list read
                // once per thread (every lane the same address: a broadcast), two barriers; the bitonic network below takes 36 rounds
                // with a barrier each for 256 entries (18.0 -> 17.2 ms; walking the list in registers with v_readlane instead: 21.6 ms)
                for (uint32_t s = cnt + t; s < ((cnt + 3u) & ~3u); s += CL_THREADS) list[s] = CL_EMPTY;
                __syncthreads();
                const uint32_t m0 = (uint32_t)t < cnt ? list[t] : CL_EMPTY, m1 = (uint32_t)t + CL_THREADS < cnt ? list[t + CL_THREADS] : CL_EMPTY;
                uint32_t r0 = 0, r1 = 0;
                for (uint32_t s = 0; s < cnt; s += 4) {
                    const cf_u32x4_cl q = *(const cf_u32x4_cl*)(list + s);
                    r0 += (uint32_t)(q.x < m0) + (uint32_t)(q.y < m0) + (uint32_t)(q.z < m0) + (uint32_t)(q.w < m0);
                    if (cnt > CL_THREADS) r1 += (uint32_t)(q.x < m1) + (uint32_t)(q.y < m1) + (uint32_t)(q.z < m1) + (uint32_t)(q.w < m1);
                }
                __syncthreads();
                if ((uint32_t)t < cnt) list[r0] = m0;
                if ((uint32_t)t + CL_THREADS < cnt) list[r1] = m1;
                n2 = 0;      // (no network)
            } else
                for (uint32_t s = cnt + t; s < n2; s += CL_THREADS) list[s] = CL_EMPTY;
            for (uint32_t size = 2; size <= n2; size <<= 1) {
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    __syncthreads();
                    for (uint32_t i = t; i < (n2 >> 1); i += CL_THREADS) {
                        const uint32_t j = i & (stride - 1);
                        const uint32_t lo = 2 * i - j, hi = lo + stride;
                        const uint32_t a = list[lo], b = list[hi];
                        const bool up = (lo & size) == 0;
                        if ((a > b) == up) { list[lo] = b; list[hi] = a; }
                    }
                }
            }
            __syncthreads();
            const int64_t o = mode == 2 ? u * row_stride : cloud_ptr[u];
            for (uint32_t s = t; s < cnt; s += CL_THREADS) entries[o + s] = (int32_t)list[s];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
cf_mult_hist_kernel(const int32_t* __restrict__ entries, int64_t n, uint32_t* __restrict__ mult) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&mult[entries[i]], 1u);
}

// one wave per unit; mode 0: sizes, mode 1: ordered compaction
__global__ void __launch_bounds__(256)
cf_mult_filter_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t n_units,
                      const uint32_t* __restrict__ mult, uint32_t min_mult, uint32_t max_mult, int mode,
                      uint32_t* __restrict__ sizes, const int64_t* __restrict__ new_ptr, int32_t* __restrict__ new_entries) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < n_units; u += n_waves) {
        const int64_t e0 = cloud_ptr[u], e1 = cloud_ptr[u + 1];
        uint32_t kept = 0;
        const int64_t rounds = (e1 - e0 + 63) / 64;
        for (int64_t rd = 0; rd < rounds; ++rd) {
            const int64_t e = e0 + rd * 64 + lane;
            bool keep = false;
            int32_t x = 0;
            if (e < e1) {
                x = entries[e];
                const uint32_t m = mult[x];
                keep = m >= min_mult && (max_mult == 0 || m <= max_mult);
            }
            const unsigned long long b = __ballot(keep);
            if (mode == 1 && keep) new_entries[new_ptr[u] + kept + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = x;
            kept += (uint32_t)__popcll(b);
        }
        if (mode == 0 && lane == 0) sizes[u] = kept;
    }
}

__global__ void __launch_bounds__(256)
cf_bits_expand_kernel(const uint32_t* __restrict__ bits, int64_t n, uint8_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (bits[i >> 5] >> (i & 31)) & 1u;
}
__global__ void __launch_bounds__(256)
cf_bits_or_kernel(uint32_t* __restrict__ bits, int64_t n, const uint8_t* __restrict__ in) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (in[i]) atomicOr(&bits[i >> 5], 1u << (i & 31));
}
__global__ void __launch_bounds__(256)
cf_bits_count_kernel(const uint32_t* __restrict__ bits, int64_t words, unsigned long long* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) c += (unsigned long long)__popc(bits[i]);
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, (unsigned)d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// Build the lookup table and the unique bitmap for ctx->d_kmers (already on the device, sorted).
int cf_install_kmers(cf_ctx* ctx, int32_t k) {
    const int64_t n = ctx->n_kmers;
    ctx->set_k = k;
    const uint64_t cap2 = cf_pow2_ceil((uint64_t)std::max<int64_t>(2 * n, 1024));      // load <= 0.5: what the prefilter is sized from
    // Round 6: a SPARSER table (load <= 0.125 up to 1.7e7 k-mers, <= 0.25 up to 1.3e8) — every workgroup of cf_cloud_kernel waits for the
    // slowest of its 2 048 look-ups, whose probe chain is what a load of 0.5 makes long: 16.0 -> 13.8 (x 2) -> 13.25 ms (x 4) at 0.99 Gb;
    // the prefilter keeps its size (it is what has to stay cache resident).  "lut_shift" 0 .. 3 forces the factor.
    const int lut_shift = ctx->lut_shift >= 0 ? ctx->lut_shift : (cap2 <= (1ull << 25) ? 2 : cap2 <= (1ull << 28) ? 1 : 0);
    ctx->lut_cap = cap2 << lut_shift;
    if (ctx->lut_cap > (1ull << 32)) return cf_fail(ctx, -34, "k-mer set too large for the lookup table's 32-bit hash");
    CF_TRY(cf_alloc_t(ctx, &ctx->d_lut, (size_t)ctx->lut_cap, "k-mer lookup table"));
    ctx->lut_pre_words = cap2 * 8 / 32;
    CF_TRY(cf_alloc_t(ctx, &ctx->d_lut_pre, (size_t)ctx->lut_pre_words, "k-mer lookup prefilter"));
    CF_HIP(hipMemsetAsync(ctx->d_lut_pre, 0, (size_t)ctx->lut_pre_words * 4, ctx->stream));
    ctx->unique_words = (n + 31) / 32 + 1;
    CF_TRY(cf_alloc_t(ctx, &ctx->d_unique_bits, (size_t)ctx->unique_words, "unique bitmap"));
    CF_HIP(hipMemsetAsync(ctx->d_lut, 0, (size_t)ctx->lut_cap * sizeof(cf_slot), ctx->stream));
    CF_HIP(hipMemsetAsync(ctx->d_unique_bits, 0, (size_t)ctx->unique_words * 4, ctx->stream));
    ctx->stats.n_unique = 0;
    unsigned int* d_flags = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_flags, 4, "lut flags"));      // (the last early return: everything below releases it)
    unsigned int flags = 0;
    int rc = 0;
    hipError_t e = hipMemsetAsync(d_flags, 0, 16, ctx->stream);
    if (e == hipSuccess && n) {
        const int grid = cf_grid_for(n, 256, std::max(1, ctx->n_cu) * 8);
        hipLaunchKernelGGL(cf_lut_build_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const unsigned long long*)ctx->d_kmers,
                           n, ctx->d_lut, (uint64_t)(ctx->lut_cap - 1), ctx->d_lut_pre, (uint64_t)(ctx->lut_pre_words - 1), d_flags);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("k-mer lookup build: ") + hipGetErrorString(e));
    cf_release_t(ctx, d_flags, 4);
    if (rc) return rc;
    if (flags & 1u) return cf_fail(ctx, -34, "k-mer lookup table overflow");
    if (flags & 6u) return cf_fail(ctx, -22, "k-mer set must be sorted ascending and unique");
    cf_free_clouds(ctx);
    return 0;
}

// popcount of the unique bitmap -> stats.n_unique
int cf_refresh_unique_count(cf_ctx* ctx) {
    if (!ctx->d_unique_bits) { ctx->stats.n_unique = 0; return 0; }
    unsigned long long* d_u = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_u, 2, "unique count"));
    unsigned long long hu = 0;
    hipError_t e = hipMemsetAsync(d_u, 0, 16, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cf_bits_count_kernel, dim3((unsigned)cf_grid_for(ctx->unique_words, 256, std::max(1, ctx->n_cu) * 8)), dim3(256), 0,
                           ctx->stream, (const uint32_t*)ctx->d_unique_bits, ctx->unique_words, d_u);
        e = hipMemcpy(&hu, d_u, 8, hipMemcpyDeviceToHost);
    }
    cf_release_t(ctx, d_u, 2);
    if (e != hipSuccess) return cf_fail(ctx, -5, std::string("unique count: ") + hipGetErrorString(e));
    ctx->stats.n_unique = (int64_t)hu;
    return 0;
}

extern "C" {

int cf_set_kmers(cf_ctx* ctx, const uint64_t* kmers, int64_t n, int32_t k) {
    if (!ctx) return -22;
    if (k < 1 || k > 31) return cf_fail(ctx, -22, "k must be in [1, 31]");
    if (n < 0 || n >= (int64_t)1 << 31) return cf_fail(ctx, -22, "bad k-mer count");
    CF_HIP(hipSetDevice(ctx->device));
    cf_free_kmers(ctx);
    CF_TRY(cf_alloc_t(ctx, &ctx->d_kmers, (size_t)n, "k-mer set"));
    ctx->n_kmers = n;
    if (n) CF_HIP(hipMemcpy(ctx->d_kmers, kmers, (size_t)n * 8, hipMemcpyDefault));
    return cf_install_kmers(ctx, k);
}

}  // extern "C" (reopened below)

// rows of the scratch buffer (fixed stride) -> CSR (one wave per unit)
__global__ void __launch_bounds__(256)
cf_cloud_compact_kernel(const int32_t* __restrict__ rows, int64_t row_stride, const int64_t* __restrict__ cloud_ptr, int64_t n_units, int32_t* __restrict__ entries) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < n_units; u += n_waves) {
        const int64_t o = cloud_ptr[u], n = cloud_ptr[u + 1] - o;
        for (int64_t i = lane; i < n; i += 64) entries[o + i] = rows[u * row_stride + i];
    }
}

extern "C" {

// One attempt with an LDS set of set_slots entries per unit; returns 1 when some unit's cloud did not fit.
static int build_clouds_attempt(cf_ctx* ctx, int set_slots, int64_t* n_entries) {
    cf_free_clouds(ctx);
    const int64_t U = ctx->n_units;
    uint32_t* d_sizes = nullptr;
    int32_t* d_rows = nullptr;
    unsigned int* d_flags = nullptr;
    CF_TRY(cf_alloc_t(ctx, &ctx->d_cloud_ptr, (size_t)U + 1, "cloud_ptr"));
    CF_TRY(cf_alloc_t(ctx, &d_sizes, (size_t)U + 1, "cloud sizes"));
    int rc = cf_alloc_t(ctx, &d_flags, 4, "cloud flags");
    const size_t lds = (size_t)set_slots * 8 + CL_THREADS * CL_TILE_W + 64 + 16;
    const int grid = (int)std::min<int64_t>(std::max<int64_t>(U, 1), (int64_t)std::max(1, ctx->n_cu) * 32);
    int64_t total = 0, row_stride = 0;
    unsigned int flags = 0;
    do {
        if (rc) break;
        hipError_t e = hipMemsetAsync(d_flags, 0, 16, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_sizes, 0, (size_t)(U + 1) * 4, ctx->stream);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)cf_cloud_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_build_clouds setup: ") + hipGetErrorString(e)); break; }
        // one pass: every unit's sorted cloud goes to a fixed-stride row of a scratch buffer (a cloud that fits the LDS set
        // has at most 3/4 of its slots), sizes are scanned, rows are compacted into the CSR
        // (nor more entries than the longest unit has windows: the retry with the 64 KiB set does not quadruple the scratch)
        row_stride = std::min<int64_t>((int64_t)set_slots * 3 / 4, std::max<int64_t>(1, ctx->max_unit_len - ctx->set_k + 1));
        if ((rc = cf_alloc_t(ctx, &d_rows, (size_t)(U * row_stride + 1), "cloud rows scratch"))) break;
        if (U) {
            hipLaunchKernelGGL(cf_cloud_kernel, dim3((unsigned)grid), dim3(CL_THREADS), lds, ctx->stream, (const uint8_t*)ctx->d_bases,
                               (const int64_t*)ctx->d_unit_start, (const int64_t*)ctx->d_unit_end, U, ctx->set_k, set_slots,
                               (const cf_slot*)ctx->d_lut, (uint64_t)(ctx->lut_cap - 1), (const uint32_t*)ctx->d_lut_pre, (uint64_t)(ctx->lut_pre_words - 1),
                               2, row_stride, d_sizes, (const int64_t*)nullptr, d_rows, d_flags);
            e = hipGetLastError();
            if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_cloud_kernel: ") + hipGetErrorString(e)); break; }
        }
        if ((rc = cf_scan_exclusive_u32_to_i64(ctx, d_sizes, ctx->d_cloud_ptr, U + 1, &total))) break;
        if (hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "cloud flags copy"); break; }
        if (flags & 1u) { rc = 1; break; }
        ctx->n_entries = total;
        if ((rc = cf_alloc_t(ctx, &ctx->d_entries, (size_t)total, "cloud entries"))) break;
        if (U && total) {
            hipLaunchKernelGGL(cf_cloud_compact_kernel, dim3((unsigned)cf_grid_for(U * 64, 256, std::max(1, ctx->n_cu) * 8)), dim3(256), 0, ctx->stream,
                               (const int32_t*)d_rows, row_stride, (const int64_t*)ctx->d_cloud_ptr, U, ctx->d_entries);
            e = hipGetLastError();
            if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_cloud_compact_kernel: ") + hipGetErrorString(e)); break; }
        }
        e = hipEventRecord(ctx->ev1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_build_clouds sync: ") + hipGetErrorString(e)); break; }
        (void)hipEventElapsedTime(&ctx->times.clouds_ms, ctx->ev0, ctx->ev1);
    } while (0);
    if (d_flags) cf_release_t(ctx, d_flags, 4);
    if (d_rows) cf_release_t(ctx, d_rows, (size_t)(U * row_stride + 1));
    cf_release_t(ctx, d_sizes, (size_t)U + 1);
    if (rc) { cf_free_clouds(ctx); return rc; }
    ctx->have_clouds = true;
    ctx->stats.n_cloud_entries = total;
    if (n_entries) *n_entries = total;
    return 0;
}

int cf_build_clouds(cf_ctx* ctx, int64_t* n_entries) {
    if (!ctx) return -22;
    if (!ctx->d_unit_ptr) return cf_fail(ctx, -22, "cf_build_clouds: no reads loaded");
    if (!ctx->d_lut) return cf_fail(ctx, -22, "cf_build_clouds: no k-mer set installed");
    CF_HIP(hipSetDevice(ctx->device));
    CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    // small LDS set first (16 KiB: many workgroups per CU); clouds of > 1536 k-mers get the 64 KiB set
    int rc = build_clouds_attempt(ctx, 2048, n_entries);
    if (rc == 1) rc = build_clouds_attempt(ctx, 8192, n_entries);
    if (rc == 1) return cf_fail(ctx, -34, "a unit holds more distinct set k-mers than the LDS cloud set (6144)");
    if (rc) return rc;
    CF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    CF_HIP(hipEventSynchronize(ctx->ev1));
    CF_HIP(hipEventElapsedTime(&ctx->times.clouds_ms, ctx->ev0, ctx->ev1));
    return 0;
}

int cf_filter_clouds(cf_ctx* ctx, uint32_t min_mult, uint32_t max_mult, int64_t* n_entries) {
    if (!ctx) return -22;
    if (!ctx->have_clouds) return cf_fail(ctx, -22, "cf_filter_clouds: no clouds built");
    CF_HIP(hipSetDevice(ctx->device));
    CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    const int64_t U = ctx->n_units, N = ctx->n_entries, K = ctx->n_kmers;
    uint32_t *d_mult = nullptr, *d_sizes = nullptr;
    int64_t* d_new_ptr = nullptr;
    int32_t* d_new_entries = nullptr;
    int64_t total = 0;
    int rc = 0;
    const int grid_e = cf_grid_for(N, 256, std::max(1, ctx->n_cu) * 8);
    const int grid_u = cf_grid_for(U * 64, 256, std::max(1, ctx->n_cu) * 8);
    do {
        if ((rc = cf_alloc_t(ctx, &d_mult, (size_t)K + 1, "k-mer multiplicities"))) break;
        if ((rc = cf_alloc_t(ctx, &d_sizes, (size_t)U + 1, "filtered sizes"))) break;
        if ((rc = cf_alloc_t(ctx, &d_new_ptr, (size_t)U + 1, "filtered cloud_ptr"))) break;
        hipError_t e = hipMemsetAsync(d_mult, 0, (size_t)(K + 1) * 4, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_sizes, 0, (size_t)(U + 1) * 4, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, "cf_filter_clouds memset"); break; }
        if (N) hipLaunchKernelGGL(cf_mult_hist_kernel, dim3((unsigned)grid_e), dim3(256), 0, ctx->stream, (const int32_t*)ctx->d_entries, N, d_mult);
        if (U) hipLaunchKernelGGL(cf_mult_filter_kernel, dim3((unsigned)grid_u), dim3(256), 0, ctx->stream, (const int64_t*)ctx->d_cloud_ptr,
                                  (const int32_t*)ctx->d_entries, U, (const uint32_t*)d_mult, min_mult, max_mult, 0, d_sizes,
                                  (const int64_t*)nullptr, (int32_t*)nullptr);
        if ((rc = cf_scan_exclusive_u32_to_i64(ctx, d_sizes, d_new_ptr, U + 1, &total))) break;
        if ((rc = cf_alloc_t(ctx, &d_new_entries, (size_t)total, "filtered entries"))) break;
        if (U && total)
            hipLaunchKernelGGL(cf_mult_filter_kernel, dim3((unsigned)grid_u), dim3(256), 0, ctx->stream, (const int64_t*)ctx->d_cloud_ptr,
                               (const int32_t*)ctx->d_entries, U, (const uint32_t*)d_mult, min_mult, max_mult, 1, d_sizes,
                               (const int64_t*)d_new_ptr, d_new_entries);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_filter_clouds: ") + hipGetErrorString(e)); break; }
        (void)hipEventElapsedTime(&ctx->times.filter_ms, ctx->ev0, ctx->ev1);
    } while (0);
    if (d_sizes) cf_release_t(ctx, d_sizes, (size_t)U + 1);
    if (d_mult) cf_release_t(ctx, d_mult, (size_t)K + 1);
    if (rc) {
        if (d_new_entries) cf_release_t(ctx, d_new_entries, (size_t)total);
        if (d_new_ptr) cf_release_t(ctx, d_new_ptr, (size_t)U + 1);
        return rc;
    }
    cf_free_gview(ctx);
    cf_release_t(ctx, ctx->d_cloud_ptr, (size_t)U + 1);
    cf_release_t(ctx, ctx->d_entries, (size_t)N);
    ctx->d_cloud_ptr = d_new_ptr;
    ctx->d_entries = d_new_entries;
    ctx->n_entries = total;
    ctx->stats.n_cloud_entries = total;
    if (n_entries) *n_entries = total;
    return 0;
}

int cf_get_unique_mask(cf_ctx* ctx, uint8_t* mask) {
    if (!ctx) return -22;
    if (!ctx->d_unique_bits) return cf_fail(ctx, -22, "cf_get_unique_mask: no k-mer set");
    const int64_t n = ctx->n_kmers;
    if (!n) return 0;
    CF_HIP(hipSetDevice(ctx->device));
    uint8_t* d_tmp = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_tmp, (size_t)n, "unique mask bytes"));
    const int grid = cf_grid_for(n, 256, std::max(1, ctx->n_cu) * 8);
    hipLaunchKernelGGL(cf_bits_expand_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->d_unique_bits, n, d_tmp);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(mask, d_tmp, (size_t)n, hipMemcpyDefault);
    cf_release_t(ctx, d_tmp, (size_t)n);
    if (e != hipSuccess) return cf_fail(ctx, -5, std::string("cf_get_unique_mask: ") + hipGetErrorString(e));
    return 0;
}

int cf_or_unique_mask(cf_ctx* ctx, const uint8_t* mask) {
    if (!ctx) return -22;
    if (!ctx->d_unique_bits) return cf_fail(ctx, -22, "cf_or_unique_mask: no k-mer set");
    const int64_t n = ctx->n_kmers;
    if (!n) return 0;
    CF_HIP(hipSetDevice(ctx->device));
    uint8_t* d_tmp = nullptr;
    unsigned long long* d_cnt = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_tmp, (size_t)n, "unique mask bytes"));
    int rc = cf_alloc_t(ctx, &d_cnt, 2, "unique count");
    unsigned long long h = 0;
    if (rc == 0) {
        hipError_t e = hipMemcpy(d_tmp, mask, (size_t)n, hipMemcpyDefault);
        if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 16, ctx->stream);
        if (e == hipSuccess) {
            const int grid = cf_grid_for(n, 256, std::max(1, ctx->n_cu) * 8);
            hipLaunchKernelGGL(cf_bits_or_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->d_unique_bits, n, (const uint8_t*)d_tmp);
            hipLaunchKernelGGL(cf_bits_count_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->d_unique_bits,
                               ctx->unique_words, d_cnt);
            e = hipMemcpy(&h, d_cnt, 8, hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("cf_or_unique_mask: ") + hipGetErrorString(e));
        else ctx->stats.n_unique = (int64_t)h;
    }
    if (d_cnt) cf_release_t(ctx, d_cnt, 2);
    cf_release_t(ctx, d_tmp, (size_t)n);
    return rc;
}

int cf_reset_unique(cf_ctx* ctx) {
    if (!ctx) return -22;
    if (!ctx->d_unique_bits) return 0;
    CF_HIP(hipSetDevice(ctx->device));
    CF_HIP(hipMemsetAsync(ctx->d_unique_bits, 0, (size_t)ctx->unique_words * 4, ctx->stream));
    CF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stats.n_unique = 0;
    return 0;
}

}  // extern "C"
