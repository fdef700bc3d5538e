// cf_count.hip — A1 (presence / multi-occurrence table) and A2 (rare window select).
//
// Reference: scripts/distance_based_kmer_recruitment.py:39-63 builds, read by read, a dict of
// per-read k-mer multiplicities and from it `all_kmers[x]` = number of reads containing x,
// dropping x once it occurs >= 2x inside more than max_nonuniq reads.  Closed form used here
// (order independent): pres[x] = #reads containing x, multi[x] = #reads where x occurs >= 2x;
// the result is {x: pres[x] | multi[x] <= max_nonuniq}; :66-82 then keeps lo <= pres <= hi.
//
// Device design (MI355X):
//   * work item = (read, hash class p of P): the item streams the whole read (coalesced byte
//     loads staged through LDS as a sliding window), keeps the k-mers whose class is p and
//     de-duplicates them in an LDS open-addressed set (one 64-bit CAS per window).  All
//     occurrences of a k-mer inside a read fall into the same item, so the per-read
//     statistics (present / occurs twice) are exact without any cross-item traffic.
//   * the set is then flushed into the HBM table: one 16-byte slot touch per distinct
//     (read, k-mer): a 64-bit CAS to claim the key and ONE 64-bit atomic add carrying
//     pres (+1, low half) and multi (+1 if seen twice, high half).
//   * A2 is a table scan with wave-ballot compaction, then the LSD radix sort of cf_prims.
#include "cf_common.h"

void cf_free_table(cf_ctx* c);
void cf_free_kmers(cf_ctx* c);
int cf_install_kmers(cf_ctx* ctx, int32_t k);  // cf_clouds.hip: builds the lookup table for ctx->d_kmers

#define CNT_THREADS 256
#define CNT_DUP (1ull << 62)

struct cf_count_item {
    int32_t read;
    int32_t cls;
    int32_t n_cls;
    int32_t pad;
};

__device__ __forceinline__ void cf_table_add(cf_slot* __restrict__ table, uint64_t mask, unsigned long long key,
                                             unsigned long long inc, unsigned int* __restrict__ flags) {
    const unsigned long long want = key | CF_OCC;
    uint64_t h = cf_mix64(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        unsigned long long cur = table[h].key;
        if (cur == 0ull) cur = atomicCAS(&table[h].key, 0ull, want);
        if (cur == 0ull || cur == want) {
            atomicAdd(&table[h].val, inc);
            return;
        }
        h = (h + 1) & mask;
    }
    atomicOr(flags, 1u);  // table full
}

// Hash of a window for the LDS stage only (class of the k-mer from the high half, set slot from the low bits): one
// 32-bit multiply — every window of every (read, class) item pays it, the 64-bit mixer cost 6.  The HBM table keeps
// cf_mix64.
__device__ __forceinline__ uint32_t cf_window_hash(unsigned long long code) {
    uint32_t x = (uint32_t)code ^ (uint32_t)(code >> 29) ^ (uint32_t)(code >> 58);
    x *= 0x9E3779B1u;
    return x ^ (x >> 16);
}

// the same for CNT_FLUSH keys of one thread at once: the key loads, then the claims, then the adds are each issued back
// to back, so a thread waits for three HBM round trips per batch instead of two or three per KEY (the flush of an
// item's LDS set used to be the longest phase of the kernel: ~9 dependent atomic chains per thread)
#define CNT_FLUSH 8
__device__ __forceinline__ void cf_table_add_batch(cf_slot* __restrict__ table, uint64_t mask, const unsigned long long (&key)[CNT_FLUSH],
                                                   const unsigned long long (&inc)[CNT_FLUSH], unsigned int live, unsigned int* __restrict__ flags) {
    uint64_t h[CNT_FLUSH];
    unsigned long long cur[CNT_FLUSH];
#pragma unroll
    for (int j = 0; j < CNT_FLUSH; ++j) { h[j] = cf_mix64(key[j]) & mask; cur[j] = ((live >> j) & 1u) ? table[h[j]].key : 1ull; }
#pragma unroll
    for (int j = 0; j < CNT_FLUSH; ++j) if (((live >> j) & 1u) && cur[j] == 0ull) cur[j] = atomicCAS(&table[h[j]].key, 0ull, key[j] | CF_OCC);
#pragma unroll
    for (int j = 0; j < CNT_FLUSH; ++j) {
        if (!((live >> j) & 1u)) continue;
        if (cur[j] == 0ull || cur[j] == (key[j] | CF_OCC)) { atomicAdd(&table[h[j]].val, inc[j]); continue; }
        // home slot holds another key: the general probe loop from the next slot on
        const unsigned long long want = key[j] | CF_OCC;
        uint64_t g = (h[j] + 1) & mask;
        bool done = false;
        for (uint64_t probe = 0; probe < mask; ++probe) {
            unsigned long long c = table[g].key;
            if (c == 0ull) c = atomicCAS(&table[g].key, 0ull, want);
            if (c == 0ull || c == want) { atomicAdd(&table[g].val, inc[j]); done = true; break; }
            g = (g + 1) & mask;
        }
        if (!done) atomicOr(flags, 1u);  // table full
    }
}

__global__ void __launch_bounds__(CNT_THREADS)
cf_count_kernel(const uint8_t* __restrict__ bases, const int64_t* __restrict__ read_off,
                const cf_count_item* __restrict__ items, int n_items, int k, int slots, int tile_w,
                cf_slot* __restrict__ table, uint64_t tmask, unsigned long long* __restrict__ n_rk,
                unsigned int* __restrict__ flags) {
    unsigned long long* set = (unsigned long long*)cf_lds;            // slots x 8 B
    uint8_t* stage = cf_lds + (size_t)slots * 8;                       // tile + k - 1 bases
    unsigned int* counters = (unsigned int*)(stage + ((CNT_THREADS * tile_w + 64 + 15) & ~15));
    const int t = threadIdx.x;
    const uint32_t smask = (uint32_t)slots - 1u;
    const unsigned long long kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const int tile = CNT_THREADS * tile_w;

    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const cf_count_item item = items[it];
        const int64_t r0 = read_off[item.read], r1 = read_off[item.read + 1];
        const int64_t n_win = r1 - r0 - k + 1;
        for (int s = t; s < slots; s += CNT_THREADS) set[s] = 0ull;
        if (t == 0) { counters[0] = 0; counters[1] = 0; }
        __syncthreads();
        for (int64_t w0 = 0; w0 < n_win; w0 += tile) {
            // stage bases [w0, w0 + tile + k - 1) of the read
            const int64_t nb = min((int64_t)tile + k - 1, r1 - r0 - w0);
            for (int64_t i = t; i < nb; i += CNT_THREADS) stage[i] = bases[r0 + w0 + i];
            __syncthreads();
            const int64_t my0 = (int64_t)t * tile_w;
            const int64_t my_n = min((int64_t)tile_w, n_win - w0 - my0);
            if (my_n > 0) {
                unsigned long long code = 0;
                int run = 0;      // upper-case A, C, G, T in a row, ending at the current base: a window holding anything else has no
                                  // 2-bit code and is skipped (the reference counts it as a k-mer of its own: cfh_exotic_summary)
                for (int j = 0; j < k - 1; ++j) { const uint32_t c = stage[my0 + j]; code = (code << 2) | cf_base2(c); run = cf_is_acgt(c) ? run + 1 : 0; }
                for (int64_t i = 0; i < my_n; ++i) {
                    const uint32_t c = stage[my0 + i + k - 1];
                    code = ((code << 2) | cf_base2(c)) & kmask;
                    run = cf_is_acgt(c) ? run + 1 : 0;
                    if (run < k) continue;
                    const uint32_t hh = cf_window_hash(code);
                    if ((int)(((hh >> 16) * ((uint32_t)item.n_cls & 0xFFFFu)) >> 16) != item.cls) continue;   // class of the k-mer (n_cls < 65536)
                    const unsigned long long want = code | CF_OCC;
                    uint32_t h = hh & smask;
                    bool done = false;
                    for (int probe = 0; probe < slots; ++probe) {
                        unsigned long long cur = set[h];
                        if (cur == 0ull) {
                            cur = atomicCAS(&set[h], 0ull, want);
                            if (cur == 0ull) { atomicAdd(&counters[0], 1u); done = true; break; }
                        }
                        if ((cur & ~CNT_DUP) == want) {
                            if (!(cur & CNT_DUP)) atomicOr(&set[h], CNT_DUP);
                            done = true;
                            break;
                        }
                        h = (h + 1) & smask;
                    }
                    if (!done) counters[1] = 1;  // LDS set full
                }
            }
            __syncthreads();
        }
        if (counters[1]) {
            if (t == 0) atomicOr(flags, 2u);
        } else {
            for (int s0 = t; s0 < slots; s0 += CNT_THREADS * CNT_FLUSH) {
                unsigned long long key[CNT_FLUSH], inc[CNT_FLUSH];
                unsigned int live = 0;
#pragma unroll
                for (int j = 0; j < CNT_FLUSH; ++j) {
                    const int s = s0 + j * CNT_THREADS;
                    const unsigned long long v = s < slots ? set[s] : 0ull;
                    key[j] = v & ~(CF_OCC | CNT_DUP);
                    inc[j] = 1ull + ((v & CNT_DUP) ? (1ull << 32) : 0ull);
                    live |= (v != 0ull ? 1u : 0u) << j;
                }
                if (live) cf_table_add_batch(table, tmask, key, inc, live, flags);
            }
            if (t == 0) atomicAdd(n_rk, (unsigned long long)counters[0]);
        }
        __syncthreads();
    }
}

// Occurrence counting (reference scripts/better_consensus_unit_reconstruction.py:127-135:
// kmer_counts_reads[kmer] += 1 for every window of every read).  Same work items as A1; the LDS set carries a count
// per key, so a k-mer that occurs thousands of times in a read (HOR k-mers) costs ONE 64-bit HBM atomic per item.
__global__ void __launch_bounds__(CNT_THREADS)
cf_occ_kernel(const uint8_t* __restrict__ bases, const int64_t* __restrict__ read_off, const cf_count_item* __restrict__ items,
              int n_items, int k, int slots, int tile_w, cf_slot* __restrict__ table, uint64_t tmask, unsigned int* __restrict__ flags) {
    unsigned long long* set = (unsigned long long*)cf_lds;            // slots x 8 B keys
    unsigned int* cnt = (unsigned int*)(cf_lds + (size_t)slots * 8);   // slots x 4 B occurrence counts
    uint8_t* stage = cf_lds + (size_t)slots * 12;
    unsigned int* counters = (unsigned int*)(stage + ((CNT_THREADS * tile_w + 64 + 15) & ~15));
    const int t = threadIdx.x;
    const uint32_t smask = (uint32_t)slots - 1u;
    const unsigned long long kmask = (1ull << (2 * k)) - 1ull;
    const int tile = CNT_THREADS * tile_w;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const cf_count_item item = items[it];
        const int64_t r0 = read_off[item.read], r1 = read_off[item.read + 1];
        const int64_t n_win = r1 - r0 - k + 1;
        for (int s = t; s < slots; s += CNT_THREADS) { set[s] = 0ull; cnt[s] = 0u; }
        if (t == 0) counters[1] = 0;
        __syncthreads();
        for (int64_t w0 = 0; w0 < n_win; w0 += tile) {
            const int64_t nb = min((int64_t)tile + k - 1, r1 - r0 - w0);
            for (int64_t i = t; i < nb; i += CNT_THREADS) stage[i] = bases[r0 + w0 + i];
            __syncthreads();
            const int64_t my0 = (int64_t)t * tile_w;
            const int64_t my_n = min((int64_t)tile_w, n_win - w0 - my0);
            if (my_n > 0) {
                unsigned long long code = 0;
                int run = 0;      // upper-case A, C, G, T in a row, ending at the current base: a window holding anything else has no
                                  // 2-bit code and is skipped (the reference counts it as a k-mer of its own: cfh_exotic_summary)
                for (int j = 0; j < k - 1; ++j) { const uint32_t c = stage[my0 + j]; code = (code << 2) | cf_base2(c); run = cf_is_acgt(c) ? run + 1 : 0; }
                for (int64_t i = 0; i < my_n; ++i) {
                    const uint32_t c = stage[my0 + i + k - 1];
                    code = ((code << 2) | cf_base2(c)) & kmask;
                    run = cf_is_acgt(c) ? run + 1 : 0;
                    if (run < k) continue;
                    const uint32_t hh = cf_window_hash(code);
                    if ((int)(((hh >> 16) * ((uint32_t)item.n_cls & 0xFFFFu)) >> 16) != item.cls) continue;   // class of the k-mer (n_cls < 65536)
                    const unsigned long long want = code | CF_OCC;
                    uint32_t h = hh & smask;
                    bool done = false;
                    for (int probe = 0; probe < slots; ++probe) {
                        unsigned long long cur = set[h];
                        if (cur == 0ull) cur = atomicCAS(&set[h], 0ull, want);
                        if (cur == 0ull || cur == want) { atomicAdd(&cnt[h], 1u); done = true; break; }
                        h = (h + 1) & smask;
                    }
                    if (!done) counters[1] = 1;  // LDS set full
                }
            }
            __syncthreads();
        }
        if (counters[1]) {
            if (t == 0) atomicOr(flags, 2u);
        } else {
            for (int s = t; s < slots; s += CNT_THREADS) {
                const unsigned long long v = set[s];
                if (v) cf_table_add(table, tmask, v & ~CF_OCC, (unsigned long long)cnt[s], flags);
            }
        }
        __syncthreads();
    }
}

// Table scan with compaction.  Every workgroup owns one contiguous chunk of the table.
// mode 0: count {occupied, kept, selected} and write the workgroup's selected count to block_io[blockIdx];
// mode 1: block_io holds the exclusive scan of those counts; selected entries are written at
//         block_io[blockIdx] + (LDS cursor), wave-ballot aggregated — no global atomics at all.
__global__ void __launch_bounds__(256)
cf_select_kernel(const cf_slot* __restrict__ table, uint64_t cap, uint32_t max_nonuniq, uint32_t lo, uint32_t hi,
                 int mode, unsigned long long* __restrict__ counts, int64_t* __restrict__ block_io,
                 unsigned long long* __restrict__ out, uint32_t* __restrict__ out_pres, uint32_t* __restrict__ out_multi,
                 int pred, unsigned long long t_val, unsigned long long t_key) {
    // pred 0: multi <= max_nonuniq && lo <= pres <= hi (A2).  pred 1 (occurrence table, val = 64-bit count):
    // val > t_val || (val == t_val && key >= t_key); counts[1] then accumulates the maximum count instead of "kept".
    unsigned int* cursor = (unsigned int*)cf_lds;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) cursor[0] = 0;
    __syncthreads();
    unsigned long long occ = 0, kept = 0, sel = 0;
    const uint64_t chunk = ((cap + gridDim.x - 1) / gridDim.x + 255) & ~255ull;
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk, c1 = min(cap, c0 + chunk);
    const unsigned long long base = mode == 1 ? (unsigned long long)block_io[blockIdx.x] : 0ull;
    for (uint64_t i0 = c0; i0 < c1; i0 += 256) {
        const uint64_t i = i0 + threadIdx.x;
        bool s = false;
        cf_slot sl; sl.key = 0; sl.val = 0;
        if (i < c1) {
            sl = table[i];
            if (sl.key) {
                const uint32_t pres = (uint32_t)sl.val, multi = (uint32_t)(sl.val >> 32);
                ++occ;
                if (pred == 1) {
                    kept = max(kept, sl.val);
                    if (sl.val > t_val || (sl.val == t_val && (sl.key & ~CF_OCC) >= t_key)) { s = true; ++sel; }
                } else if (multi <= max_nonuniq) {
                    ++kept;
                    if (pres >= lo && pres <= hi) { s = true; ++sel; }
                }
            }
        }
        if (mode == 1) {
            const unsigned long long m = __ballot(s);
            if (m) {
                unsigned int off = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) off = atomicAdd(&cursor[0], (unsigned int)__popcll(m));
                off = __shfl(off, leader);
                if (s) {
                    const unsigned long long o = base + off + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                    out[o] = sl.key & ~CF_OCC;
                    if (out_pres) { out_pres[o] = (uint32_t)sl.val; out_multi[o] = (uint32_t)(sl.val >> 32); }
                }
            }
        }
    }
    if (mode == 0) {
        for (int d = 32; d >= 1; d >>= 1) {
            occ += __shfl_down(occ, (unsigned)d);
            { const unsigned long long o = __shfl_down(kept, (unsigned)d); kept = pred == 1 ? max(kept, o) : kept + o; }
            sel += __shfl_down(sel, (unsigned)d);
        }
        if (lane == 0) {
            if (occ) atomicAdd(&counts[0], occ);
            if (kept) { if (pred == 1) atomicMax(&counts[1], kept); else atomicAdd(&counts[1], kept); }
            if (sel) { atomicAdd(&counts[2], sel); atomicAdd(&cursor[0], (unsigned int)sel); }
        }
        __syncthreads();
        if (threadIdx.x == 0) block_io[blockIdx.x] = (int64_t)cursor[0];
    }
}

__global__ void __launch_bounds__(256)
cf_table_merge_kernel(cf_slot* __restrict__ table, uint64_t tmask, const unsigned long long* __restrict__ keys,
                      const uint32_t* __restrict__ pres, const uint32_t* __restrict__ multi, int64_t n,
                      unsigned int* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        cf_table_add(table, tmask, keys[i], (unsigned long long)pres[i] | ((unsigned long long)multi[i] << 32), flags);
}

// Count, then compact the table entries passing (multi <= max_nonuniq, lo <= pres <= hi) into d_keys (and
// d_pres / d_multi when given).  counts_out: {occupied, kept, selected}.  Buffers are allocated by the caller
// through `alloc(n_selected)` once the count is known.
template <class Alloc>
static int table_compact(cf_ctx* ctx, uint32_t max_nonuniq, uint32_t lo, uint32_t hi, unsigned long long counts_out[3], Alloc alloc,
                         unsigned long long** d_keys, uint32_t** d_pres, uint32_t** d_multi, int pred = 0,
                         unsigned long long t_val = 0, unsigned long long t_key = 0) {
    const int grid = std::max(1, ctx->n_cu) * 8;
    unsigned long long* d_cnt = nullptr;
    int64_t* d_blk = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_cnt, 4, "select counters"));
    int rc = cf_alloc_t(ctx, &d_blk, (size_t)grid + 1, "select block counts");
    unsigned long long h[4] = {0, 0, 0, 0};
    do {
        if (rc) break;
        if (hipMemsetAsync(d_cnt, 0, 32, ctx->stream) != hipSuccess || hipMemsetAsync(d_blk, 0, (size_t)(grid + 1) * 8, ctx->stream) != hipSuccess) { rc = cf_fail(ctx, -5, "select memset"); break; }
        hipLaunchKernelGGL(cf_select_kernel, dim3((unsigned)grid), dim3(256), 16, ctx->stream, (const cf_slot*)ctx->d_table, (uint64_t)ctx->table_cap,
                           max_nonuniq, lo, hi, 0, d_cnt, d_blk, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, pred, t_val, t_key);
        if (hipMemcpy(h, d_cnt, 32, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "select count"); break; }
        counts_out[0] = h[0]; counts_out[1] = h[1]; counts_out[2] = h[2];
        if ((rc = alloc((int64_t)h[2]))) break;
        if (h[2] == 0 || !*d_keys) break;
        if ((rc = cf_scan_exclusive_i64(ctx, d_blk, d_blk, grid + 1, nullptr))) break;
        hipLaunchKernelGGL(cf_select_kernel, dim3((unsigned)grid), dim3(256), 16, ctx->stream, (const cf_slot*)ctx->d_table, (uint64_t)ctx->table_cap,
                           max_nonuniq, lo, hi, 1, d_cnt, d_blk, *d_keys, d_pres ? *d_pres : (uint32_t*)nullptr, d_multi ? *d_multi : (uint32_t*)nullptr, pred, t_val, t_key);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_select_kernel: ") + hipGetErrorString(e)); break; }
    } while (0);
    if (d_blk) cf_release_t(ctx, d_blk, (size_t)grid + 1);
    cf_release_t(ctx, d_cnt, 4);
    return rc;
}

int cf_count_sorted(cf_ctx* ctx, int32_t k, int64_t read_lo, int64_t read_hi, int64_t n_w, int occ);   // cf_count2.hip

int cf_table_ensure(cf_ctx* ctx, uint64_t want_cap) {
    ctx->table_dense = false;
    // a table no larger than the current allocation uses a prefix of it (steps of the sharded path alternate between
    // the local count table and the merged table of owned keys: no 10-GB free + malloc per step)
    if (ctx->d_table && want_cap <= ctx->table_alloc) { ctx->table_cap = want_cap; return 0; }
    cf_free_table(ctx);
    CF_TRY(cf_alloc(ctx, (void**)&ctx->d_table, (size_t)want_cap * sizeof(cf_slot), "k-mer table"));
    ctx->table_cap = want_cap; ctx->table_alloc = want_cap;
    return 0;
}

extern "C" {

}  // extern "C" (reopened below)

// mode 0: presence / multi-occurrence table (A1); mode 1: occurrence counts
static int count_impl(cf_ctx* ctx, int32_t k, int64_t read_lo, int64_t read_hi, int mode) {
    if (!ctx) return -22;
    if (!ctx->d_read_off) return cf_fail(ctx, -22, "cf_count_kmers: no reads loaded");
    if (k < 1 || k > 31) return cf_fail(ctx, -22, "k must be in [1, 31]");
    CF_HIP(hipSetDevice(ctx->device));
    if (read_lo < 0) read_lo = 0;
    if (read_hi > ctx->n_reads) read_hi = ctx->n_reads;
    int64_t n_w = 0;
    for (int64_t r = read_lo; r < read_hi; ++r) {
        const int64_t len = ctx->h_read_off[(size_t)r + 1] - ctx->h_read_off[(size_t)r];
        if (len >= k) n_w += len - k + 1;
    }
    uint64_t cap = cf_pow2_ceil((uint64_t)std::max<int64_t>(n_w, 1024));
    if (2 * k < 62 && cap > (2ull << (2 * k))) cap = cf_pow2_ceil(2ull << (2 * k));
    ctx->k = k;
    ctx->stats.n_windows = n_w;
    if (mode == 1 && ctx->has_exotic)
        return cf_fail(ctx, -22, "cf_count_occurrences: reads with symbols other than upper-case A, C, G, T are not supported");
    if (ctx->count_mode) {      // by sort and reduce; 1 = does not apply here (long k, too many reads, a crowded bucket)
        const int rc2 = cf_count_sorted(ctx, k, read_lo, read_hi, n_w, mode);
        if (rc2 <= 0) return rc2;
    }
    // (the atomic-table path below skips windows with symbols other than upper-case A, C, G, T as well: round 3, ADVICE)
    const int slots = ctx->count_slots;
    int shrink = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        CF_TRY(cf_table_ensure(ctx, cap));
        // items: each read is split into n_cls hash classes so that a class fits the LDS set
        const int64_t per_cls = std::max<int64_t>(32, ((int64_t)slots * 3 / 8) >> shrink);
        std::vector<cf_count_item> items;
        for (int64_t r = read_lo; r < read_hi; ++r) {
            const int64_t len = ctx->h_read_off[(size_t)r + 1] - ctx->h_read_off[(size_t)r];
            if (len < k) continue;
            const int64_t nw = len - k + 1;
            const int n_cls = (int)((nw + per_cls - 1) / per_cls);
            if (n_cls > 65535) return cf_fail(ctx, -34, "cf_count_kmers: a read needs more than 65535 hash classes (raise count_slots)");
            for (int c = 0; c < n_cls; ++c) items.push_back({(int32_t)r, c, n_cls, 0});
        }
        CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        CF_HIP(hipMemsetAsync(ctx->d_table, 0, (size_t)cap * sizeof(cf_slot), ctx->stream));
        cf_count_item* d_items = nullptr;
        unsigned long long* d_cnt = nullptr;
        CF_TRY(cf_alloc_t(ctx, &d_items, items.size(), "count items"));
        int rc = cf_alloc_t(ctx, &d_cnt, 2, "count counters");
        unsigned long long h_cnt[2] = {0, 0};
        if (rc == 0) {
            hipError_t e = hipMemsetAsync(d_cnt, 0, 16, ctx->stream);
            if (e == hipSuccess && !items.empty())
                e = hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(cf_count_item), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && !items.empty()) {
                const int tile_w = ctx->count_tile;
                const size_t lds = (size_t)slots * (mode == 1 ? 12 : 8) + (size_t)((CNT_THREADS * tile_w + 64 + 15) & ~15) + 16;
                const int grid = (int)std::min<int64_t>((int64_t)items.size(), (int64_t)std::max(1, ctx->n_cu) * 8);
                if (lds > 64 * 1024)
                    e = mode == 1 ? hipFuncSetAttribute((const void*)cf_occ_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                  : hipFuncSetAttribute((const void*)cf_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) {
                    (void)hipEventRecord(ctx->ev2, ctx->stream);
                    if (mode == 1)
                        hipLaunchKernelGGL(cf_occ_kernel, dim3((unsigned)grid), dim3(CNT_THREADS), lds, ctx->stream,
                                           (const uint8_t*)ctx->d_bases, (const int64_t*)ctx->d_read_off, (const cf_count_item*)d_items,
                                           (int)items.size(), (int)k, slots, tile_w, ctx->d_table, (uint64_t)(cap - 1), (unsigned int*)(d_cnt + 1));
                    else
                        hipLaunchKernelGGL(cf_count_kernel, dim3((unsigned)grid), dim3(CNT_THREADS), lds, ctx->stream,
                                           (const uint8_t*)ctx->d_bases, (const int64_t*)ctx->d_read_off, (const cf_count_item*)d_items,
                                           (int)items.size(), (int)k, slots, tile_w, ctx->d_table, (uint64_t)(cap - 1), d_cnt,
                                           (unsigned int*)(d_cnt + 1));
                    e = hipGetLastError();
                    (void)hipEventRecord(ctx->ev3, ctx->stream);
                }
            }
            if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 16, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
            if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
            if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("cf_count_kmers: ") + hipGetErrorString(e));
        }
        if (d_cnt) cf_release_t(ctx, d_cnt, 2);
        cf_release_t(ctx, d_items, items.size());
        if (rc) return rc;
        const unsigned int flags = (unsigned int)h_cnt[1];
        if (flags & 1u) { cap *= 2; continue; }            // HBM table full: retry larger
        if (flags & 2u) { ++shrink; continue; }            // an LDS set overflowed: more classes per read
        ctx->stats.n_read_kmers = (int64_t)h_cnt[0];
        (void)hipEventElapsedTime(&ctx->times.count_ms, ctx->ev0, ctx->ev1);
        if (!items.empty()) (void)hipEventElapsedTime(&ctx->times.count_kernel_ms, ctx->ev2, ctx->ev3);
        return 0;
    }
    return cf_fail(ctx, -34, "cf_count_kmers: k-mer table kept overflowing");
}

// histogram of the occupied slots by floor(log2(val)) (val >= 1): 64 bins, per-block in LDS, one global add per bin and block
__global__ void __launch_bounds__(256)
cf_val_log2_hist_kernel(const cf_slot* __restrict__ table, uint64_t cap, unsigned long long* __restrict__ hist) {
    unsigned int* h = (unsigned int*)cf_lds;      // 64 bins
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) {
        const cf_slot s = table[i];
        if (s.key != 0ull && s.val != 0ull) atomicAdd(&h[63 - __clzll((long long)s.val)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
__global__ void __launch_bounds__(256)
cf_pack_slots_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, int64_t n, cf_slot* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        cf_slot s; s.key = keys[i] | CF_OCC; s.val = (unsigned long long)lo[i] | ((unsigned long long)hi[i] << 32);
        out[i] = s;
    }
}

// the n k-mers with the largest (count, k-mer) of an occurrence table, sorted descending
// (reference better_consensus_unit_reconstruction.py:156-167: heapq.nlargest(n, counts, key=lambda kmer: (counts[kmer], kmer)))
static int top_exact(cf_ctx* ctx, int64_t n, uint64_t* keys_out, uint64_t* counts_out, int64_t* n_out);

// The exact thresholds below cost ~80 scans of the table, and the occurrence table is sized for every window (16 GiB at
// 1 Gb).  So the table is scanned twice first: a histogram of floor(log2(count)) finds the power of two 2^b with
// #(count >= 2^b) >= n, a compaction copies those entries — a superset of the answer, usually a few thousand — into a
// small dense table, and the threshold searches run on that (255 -> ~20 ms for the top 6 165 of 1.7e8 k-mers).
static int top_impl(cf_ctx* ctx, int64_t n, uint64_t* keys_out, uint64_t* counts_out, int64_t* n_out) {
    if (n <= 0 || !keys_out || ctx->table_cap < 4096) return top_exact(ctx, n, keys_out, counts_out, n_out);
    unsigned long long* d_hist = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_hist, 64, "count magnitude histogram"));
    unsigned long long hist[64];
    hipError_t e = hipMemsetAsync(d_hist, 0, 64 * 8, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cf_val_log2_hist_kernel, dim3((unsigned)(std::max(1, ctx->n_cu) * 8)), dim3(256), 256, ctx->stream, (const cf_slot*)ctx->d_table, (uint64_t)ctx->table_cap, d_hist);
        e = hipMemcpy(hist, d_hist, 64 * 8, hipMemcpyDeviceToHost);
    }
    cf_release_t(ctx, d_hist, 64);
    if (e != hipSuccess) return cf_fail(ctx, -5, std::string("cf_top_kmers histogram: ") + hipGetErrorString(e));
    int b = 63;
    unsigned long long above = 0, total = 0;
    for (int i = 0; i < 64; ++i) total += hist[i];
    while (b > 0 && above + hist[b] < (unsigned long long)n) { above += hist[b]; --b; }
    const unsigned long long n_cand = above + hist[b];
    if (b == 0 || n_cand > total / 4 || n_cand < (unsigned long long)std::min<int64_t>(n, (int64_t)total)) return top_exact(ctx, n, keys_out, counts_out, n_out);      // no real reduction: search the table itself
    // compact count >= 2^b into a small dense table and search there
    unsigned long long c[3] = {0, 0, 0};
    unsigned long long* d_keys = nullptr; uint32_t *d_lo = nullptr, *d_hi = nullptr;
    int64_t m_sel = 0;
    auto alloc = [&](int64_t m) -> int {
        m_sel = m;
        CF_TRY(cf_alloc_t(ctx, &d_keys, (size_t)m, "top candidates keys"));
        CF_TRY(cf_alloc_t(ctx, &d_lo, (size_t)m, "top candidates lo"));
        return cf_alloc_t(ctx, &d_hi, (size_t)m, "top candidates hi");
    };
    int rc = table_compact(ctx, 0, 0, 0, c, alloc, &d_keys, &d_lo, &d_hi, 1, (1ull << b) - 1ull, ~0ull);
    cf_slot* d_small = nullptr;
    if (rc == 0 && (unsigned long long)m_sel != n_cand) rc = cf_fail(ctx, -5, "cf_top_kmers: internal error, candidate count mismatch");
    if (rc == 0) rc = cf_alloc_t(ctx, &d_small, (size_t)m_sel, "top candidates");
    if (rc == 0) {
        hipLaunchKernelGGL(cf_pack_slots_kernel, dim3((unsigned)cf_grid_for(m_sel, 256, std::max(1, ctx->n_cu) * 8)), dim3(256), 0, ctx->stream,
                           (const unsigned long long*)d_keys, (const uint32_t*)d_lo, (const uint32_t*)d_hi, m_sel, d_small);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cf_fail(ctx, -5, "cf_top_kmers: candidate pack");
    }
    if (d_hi) cf_release_t(ctx, d_hi, (size_t)m_sel);
    if (d_lo) cf_release_t(ctx, d_lo, (size_t)m_sel);
    if (d_keys) cf_release_t(ctx, d_keys, (size_t)m_sel);
    if (rc == 0) {
        cf_slot* const big = ctx->d_table; const uint64_t big_cap = ctx->table_cap;
        ctx->d_table = d_small; ctx->table_cap = (uint64_t)m_sel;
        rc = top_exact(ctx, n, keys_out, counts_out, n_out);
        ctx->d_table = big; ctx->table_cap = big_cap;
    }
    if (d_small) cf_release_t(ctx, d_small, (size_t)m_sel);
    return rc;
}

static int top_exact(cf_ctx* ctx, int64_t n, uint64_t* keys_out, uint64_t* counts_out, int64_t* n_out) {
    unsigned long long c[3] = {0, 0, 0};
    auto none = [&](int64_t) -> int { return 0; };
    unsigned long long* nokeys = nullptr;
    // pred 1 with (t_val, t_key): val > t_val || (val == t_val && key >= t_key)
    auto count_ge = [&](unsigned long long t, unsigned long long kt, unsigned long long* cnt, unsigned long long* mx) -> int {
        int rc = table_compact(ctx, 0, 0, 0, c, none, &nokeys, nullptr, nullptr, 1, t, kt);
        *cnt = c[2]; if (mx) *mx = c[1];
        return rc;
    };
    unsigned long long total = 0, vmax = 0, cnt = 0;
    CF_TRY(count_ge(0, ~0ull, &total, &vmax));              // val > 0: every occupied slot
    if (n > (int64_t)total) n = (int64_t)total;
    *n_out = n;
    if (n == 0 || !keys_out) return 0;
    // largest T with #(val >= T) >= n   (#(val >= T) = #(val > T - 1))
    unsigned long long lo = 1, hi = vmax;
    while (lo < hi) {
        const unsigned long long mid = lo + (hi - lo + 1) / 2;
        CF_TRY(count_ge(mid - 1, ~0ull, &cnt, nullptr));
        if ((int64_t)cnt >= n) lo = mid; else hi = mid - 1;
    }
    const unsigned long long T = lo;
    unsigned long long above = 0;
    CF_TRY(count_ge(T, ~0ull, &above, nullptr));             // val > T
    const int64_t need = n - (int64_t)above;                // ties at val == T: the `need` largest k-mers
    unsigned long long klo = 0, khi = (ctx->k >= 32) ? ~0ull : ((1ull << (2 * ctx->k)) - 1ull);
    while (klo < khi) {                                      // largest kt with #(val == T && key >= kt) >= need
        const unsigned long long mid = klo + (khi - klo + 1) / 2;
        CF_TRY(count_ge(T, mid, &cnt, nullptr));
        if ((int64_t)cnt - (int64_t)above >= need) klo = mid; else khi = mid - 1;
    }
    unsigned long long* d_keys = nullptr; uint32_t *d_lo = nullptr, *d_hi = nullptr;
    int64_t n_sel = 0;
    auto alloc = [&](int64_t m) -> int {
        n_sel = m;
        CF_TRY(cf_alloc_t(ctx, &d_keys, (size_t)m, "top keys"));
        CF_TRY(cf_alloc_t(ctx, &d_lo, (size_t)m, "top counts lo"));
        return cf_alloc_t(ctx, &d_hi, (size_t)m, "top counts hi");
    };
    int rc = table_compact(ctx, 0, 0, 0, c, alloc, &d_keys, &d_lo, &d_hi, 1, T, klo);
    std::vector<unsigned long long> hk((size_t)n_sel);
    std::vector<uint32_t> hl((size_t)n_sel), hh((size_t)n_sel);
    if (rc == 0 && n_sel != n) rc = cf_fail(ctx, -5, "cf_top_kmers: internal error, selection size mismatch");
    if (rc == 0) {
        hipError_t e = hipMemcpy(hk.data(), d_keys, (size_t)n_sel * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hl.data(), d_lo, (size_t)n_sel * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hh.data(), d_hi, (size_t)n_sel * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("cf_top_kmers copy: ") + hipGetErrorString(e));
    }
    if (d_hi) cf_release_t(ctx, d_hi, (size_t)n_sel);
    if (d_lo) cf_release_t(ctx, d_lo, (size_t)n_sel);
    if (d_keys) cf_release_t(ctx, d_keys, (size_t)n_sel);
    if (rc) return rc;
    std::vector<int64_t> idx((size_t)n);
    for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = i;
    auto val = [&](int64_t i) { return (unsigned long long)hl[(size_t)i] | ((unsigned long long)hh[(size_t)i] << 32); };
    std::sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return val(a) != val(b) ? val(a) > val(b) : hk[(size_t)a] > hk[(size_t)b]; });
    for (int64_t i = 0; i < n; ++i) { keys_out[i] = hk[(size_t)idx[(size_t)i]]; counts_out[i] = val(idx[(size_t)i]); }
    return 0;
}

extern "C" {

int cf_count_kmers(cf_ctx* ctx, int32_t k, int64_t read_lo, int64_t read_hi) { return count_impl(ctx, k, read_lo, read_hi, 0); }

int cf_count_occurrences(cf_ctx* ctx, int32_t k, int64_t read_lo, int64_t read_hi) { return count_impl(ctx, k, read_lo, read_hi, 1); }

int cf_top_kmers(cf_ctx* ctx, int64_t n, uint64_t* keys_out, uint64_t* counts_out, int64_t* n_out) {
    if (!ctx || !n_out) return -22;
    if (!ctx->d_table) return cf_fail(ctx, -22, "cf_top_kmers: no table (call cf_count_occurrences first)");
    if (n < 0) return cf_fail(ctx, -22, "cf_top_kmers: n < 0");
    CF_HIP(hipSetDevice(ctx->device));
    return top_impl(ctx, n, keys_out, counts_out, n_out);
}

int cf_reset_table(cf_ctx* ctx, int32_t k, int64_t expected_keys) {
    if (!ctx) return -22;
    if (k < 1 || k > 31) return cf_fail(ctx, -22, "k must be in [1, 31]");
    CF_HIP(hipSetDevice(ctx->device));
    const uint64_t cap = cf_pow2_ceil((uint64_t)std::max<int64_t>(2 * expected_keys, 1024));
    CF_TRY(cf_table_ensure(ctx, cap));
    CF_HIP(hipMemsetAsync(ctx->d_table, 0, (size_t)cap * sizeof(cf_slot), ctx->stream));
    CF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->k = k;
    return 0;
}

int cf_get_table(cf_ctx* ctx, uint64_t* keys, uint32_t* pres, uint32_t* multi, int64_t cap, int64_t* n_out) {
    if (!ctx || !n_out) return -22;
    if (!ctx->d_table) return cf_fail(ctx, -22, "cf_get_table: no table");
    CF_HIP(hipSetDevice(ctx->device));
    unsigned long long *d_keys = nullptr;
    uint32_t *d_pres = nullptr, *d_multi = nullptr;
    unsigned long long counts[3] = {0, 0, 0};
    int64_t n_alloc = 0;
    auto alloc = [&](int64_t n) -> int {
        *n_out = n;
        if (!keys) return 0;   // size query
        if (n > cap) return cf_fail(ctx, -22, "cf_get_table: buffer too small");
        n_alloc = n;
        CF_TRY(cf_alloc_t(ctx, &d_keys, (size_t)n, "dump keys"));
        CF_TRY(cf_alloc_t(ctx, &d_pres, (size_t)n, "dump pres"));
        return cf_alloc_t(ctx, &d_multi, (size_t)n, "dump multi");
    };
    int rc = table_compact(ctx, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu, counts, alloc, &d_keys, &d_pres, &d_multi);
    if (rc == 0 && keys && n_alloc) {
        hipError_t e = hipMemcpy(keys, d_keys, (size_t)n_alloc * 8, hipMemcpyDefault);
        if (e == hipSuccess) e = hipMemcpy(pres, d_pres, (size_t)n_alloc * 4, hipMemcpyDefault);
        if (e == hipSuccess) e = hipMemcpy(multi, d_multi, (size_t)n_alloc * 4, hipMemcpyDefault);
        if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("cf_get_table copy: ") + hipGetErrorString(e));
    }
    if (d_multi) cf_release_t(ctx, d_multi, (size_t)n_alloc);
    if (d_pres) cf_release_t(ctx, d_pres, (size_t)n_alloc);
    if (d_keys) cf_release_t(ctx, d_keys, (size_t)n_alloc);
    return rc;
}

int cf_merge_table(cf_ctx* ctx, const uint64_t* keys, const uint32_t* pres, const uint32_t* multi, int64_t n) {
    if (!ctx) return -22;
    if (!ctx->d_table) return cf_fail(ctx, -22, "cf_merge_table: no table (call cf_count_kmers first)");
    if (ctx->table_dense) return cf_fail(ctx, -22, "cf_merge_table: the table of cf_count_kmers is a dense array (call cf_reset_table first)");
    if (n <= 0) return 0;
    CF_HIP(hipSetDevice(ctx->device));
    unsigned long long* d_keys = nullptr; uint32_t *d_pres = nullptr, *d_multi = nullptr; unsigned int* d_flags = nullptr;
    int rc = 0;
    unsigned int flags = 0;
    do {
        if ((rc = cf_alloc_t(ctx, &d_keys, (size_t)n, "merge keys"))) break;
        if ((rc = cf_alloc_t(ctx, &d_pres, (size_t)n, "merge pres"))) break;
        if ((rc = cf_alloc_t(ctx, &d_multi, (size_t)n, "merge multi"))) break;
        if ((rc = cf_alloc_t(ctx, &d_flags, 4, "merge flags"))) break;
        hipError_t e = hipMemcpy(d_keys, keys, (size_t)n * 8, hipMemcpyDefault);
        if (e == hipSuccess) e = hipMemcpy(d_pres, pres, (size_t)n * 4, hipMemcpyDefault);
        if (e == hipSuccess) e = hipMemcpy(d_multi, multi, (size_t)n * 4, hipMemcpyDefault);
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 16, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_merge_table copy: ") + hipGetErrorString(e)); break; }
        const int grid = cf_grid_for(n, 256, std::max(1, ctx->n_cu) * 8);
        hipLaunchKernelGGL(cf_table_merge_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->d_table,
                           (uint64_t)(ctx->table_cap - 1), (const unsigned long long*)d_keys, (const uint32_t*)d_pres,
                           (const uint32_t*)d_multi, n, d_flags);
        if (hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "cf_merge_table sync"); break; }
        if (flags & 1u) rc = cf_fail(ctx, -34, "cf_merge_table: table full");
    } while (0);
    if (d_flags) cf_release_t(ctx, d_flags, 4);
    if (d_multi) cf_release_t(ctx, d_multi, (size_t)n);
    if (d_pres) cf_release_t(ctx, d_pres, (size_t)n);
    if (d_keys) cf_release_t(ctx, d_keys, (size_t)n);
    return rc;
}

int cf_select_rare(cf_ctx* ctx, int32_t max_nonuniq, uint32_t lo, uint32_t hi, int64_t* n_out) {
    if (!ctx) return -22;
    if (!ctx->d_table) return cf_fail(ctx, -22, "cf_select_rare: no table (call cf_count_kmers first)");
    CF_HIP(hipSetDevice(ctx->device));
    const uint32_t mn = max_nonuniq < 0 ? 0u : (uint32_t)max_nonuniq;
    if (max_nonuniq < 0) hi = 0, lo = 1;  // multi <= negative is never true: empty set
    CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    unsigned long long *d_keys = nullptr, *d_tmp = nullptr;
    unsigned long long counts[3] = {0, 0, 0};
    int64_t n_sel = 0;
    auto alloc = [&](int64_t n) -> int {
        if (n >= (int64_t)1 << 31) return cf_fail(ctx, -34, "more than 2^31 selected k-mers");
        n_sel = n;
        CF_TRY(cf_alloc_t(ctx, &d_keys, (size_t)n, "selected k-mers"));
        return cf_alloc_t(ctx, &d_tmp, (size_t)n, "sort scratch");
    };
    int rc = table_compact(ctx, mn, lo, hi, counts, alloc, &d_keys, nullptr, nullptr);
    if (rc == 0) {
        ctx->stats.n_distinct = (int64_t)counts[0];
        ctx->stats.n_kept = (int64_t)counts[1];
        if (n_sel) rc = cf_radix_sort_u64(ctx, d_keys, d_tmp, n_sel, 2 * ctx->k);
    }
    if (d_tmp) cf_release_t(ctx, d_tmp, (size_t)n_sel);
    if (rc) { if (d_keys) cf_release_t(ctx, d_keys, (size_t)n_sel); return rc; }
    cf_free_kmers(ctx);
    ctx->d_kmers = d_keys;
    ctx->n_kmers = n_sel;
    CF_TRY(cf_install_kmers(ctx, ctx->k));
    CF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    CF_HIP(hipEventSynchronize(ctx->ev1));
    CF_HIP(hipEventElapsedTime(&ctx->times.select_ms, ctx->ev0, ctx->ev1));
    if (n_out) *n_out = n_sel;
    return 0;
}

}  // extern "C"
