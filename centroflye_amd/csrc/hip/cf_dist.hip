// cf_dist.hip — A5 (k-mer pair / unit-distance histogram) fused with A6 (edge filter).
//
// Reference: scripts/distance_based_kmer_recruitment.py:85-128 counts, for every read, every
// pair of unit clouds d units apart (min_d <= d <= max_d) and every ordered pair (a in C_i,
// b in C_{i+d}, a != b), cnt[d][a][b] += 1; :131-149 keeps (d, a, b, cnt) with cnt >= min_cov
// and cnt / sum_d' cnt[d'][a][b] >= 0.8 (true division), and marks a and b as selected.
//
// Device design (the dominant kernel of the path; HBM/L2 streaming + LDS atomics, no MFMA):
//   * dist_cnt[d][a] is owned by the first k-mer a in the reference; so is it here: one workgroup owns one a at a time
//     (ticket queues in the locality order of the first k-mers).  Before the launch two builder kernels cut a's partner
//     entries — for every posting (a unit holding a) the clouds of the later units of that read, ONE contiguous CSR range —
//     into item records of <= 256 entries in HBM; a wave reads its records with one load and both sweeps run on them.
//   * sweep 1 only COUNTS every (b, d) pair in 4-bit counters (8-bit ones for min_cov > 9, or when a 4-bit one wrapped) laid over the
//     table's LDS (most pairs never reach min_cov) and marks hash(b) of the pairs that can; sweep 2 queues the pairs of marked b and inserts them, 64 at a time, into an exact
//     open-addressed LDS table (6-byte slots [d | b] + 15-bit count in 4-key buckets; 8-byte slots / region layouts for
//     large k-mer sets).  The slot hash depends on b only, so all d of one b share a probe chain and sum_d cnt is a chain walk:
//     both filters run in LDS, and the lane that evaluates a selected slot writes the edge row itself, into a chunk of the
//     output the workgroup reserved earlier.
//   * a table that would overflow is split by a second hash of b into 2, 4, ... partitions, each processed (re-streamed)
//     on its own: exact, no HBM spill.
//   * first k-mers partition across GPUs (a % n_parts == part) with no reduction.
#include "cf_common.h"

#include <cstdlib>

void cf_free_edges(cf_ctx* c);
int cf_refresh_unique_count(cf_ctx* ctx);  // cf_clouds.hip

#define DIST_STAGE_CAP 1024              /* selected slots whose rows do not fit the workgroup's output chunk, staged per table pass (u16 slot indices) */
#define DIST_STACK 112
#define DIST_UNROLL 4
#define DIST_ITEM (64u * DIST_UNROLL)    /* cloud entries one wave takes per step: DIST_UNROLL consecutive ones per lane */
#define DIST_BM_BITS 65536u              /* bitmap over hash(b): k-mers that may have a selected edge */
#define DIST_HOT_CAP 2048u               /* slots the insert path can hand to the filter per pass (more: the filter scans the table) */
#define DIST_EDGE_CHUNK 8192ull          /* edge rows a workgroup reserves per global atomic */
#define DIST_OVQ 160u                    /* per-wave list of inserts whose first probe did not finish (run through the probe loop down to < 32 after every drain: 31 + the 128 a drain can park) */
#define DIST_LDS_HEAD (DIST_BM_BITS / 8 + 128)  /* bitmap + sh (32 words): the fixed head of the kernel's LDS */
#define DIST_CNT_MASK 0x7FFFFFu          /* 23-bit count */
#define DIST_SEL_BIT (1ull << 23)        /* slot selected by the A6 filter */

#ifndef cf_ballot
// the mask of a condition straight from the compare that made it (HIP's __ballot takes an int: the condition goes through a 0 / 1 vector
// register and a second compare)
__device__ __forceinline__ unsigned long long cf_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#endif

// One posting (= one unit g holding the first k-mer): its partner entries — the clouds of the units g+min_d ..
// min(read end, g+max_d) — are ONE contiguous CSR range [e0, e0 + len); ig is the index of g inside its read.
struct alignas(16) cf_dist_rec { int64_t e0; uint32_t len; uint32_t ig; };

__global__ void __launch_bounds__(256)
cf_post_hist_kernel(const int32_t* __restrict__ entries, int64_t e0, int64_t e1, uint32_t part, uint32_t n_parts, uint32_t* __restrict__ cnt) {
    // postings are only needed for the first k-mers of this partition (a % n_parts == part): at N GPUs every rank
    // scans all clouds but keeps 1/N of the postings
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e1; i += stride) {
        const uint32_t x = (uint32_t)entries[i];
        if (n_parts == 1u || x % n_parts == part) atomicAdd(&cnt[x], 1u);
    }
}

__global__ void __launch_bounds__(256)
cf_post_fill_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t u0, int64_t u1, uint32_t part, uint32_t n_parts,
                    const int64_t* __restrict__ post_ptr, uint32_t* __restrict__ cursor, int32_t* __restrict__ post,
                    uint32_t* __restrict__ first_unit) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = u0 + wave; u < u1; u += n_waves) {
        const int64_t a = cloud_ptr[u], b = cloud_ptr[u + 1];
        for (int64_t e = a + lane; e < b; e += 64) {
            const int32_t x = entries[e];
            if (n_parts != 1u && (uint32_t)x % n_parts != part) continue;
            post[post_ptr[x] + atomicAdd(&cursor[x], 1u)] = (int32_t)u;
            if (first_unit[x] > (uint32_t)u) atomicMin(&first_unit[x], (uint32_t)u);
        }
    }
}

// Postings by SORT (one GPU, every first k-mer: n_parts == 1).  The postings are the transpose of the cloud CSR; round 2 built them
// with one device-scope returning atomic per cloud entry (a histogram pass and a fill pass: 12 ms, 40 GB of traffic for 0.45 GB of
// output).  Here every cloud entry becomes one record [unit : high | rank : low kb bits], the records are radix-sorted on the
// rank bits (stable: the units of a k-mer stay in ascending order), and the run boundaries give the counts and the first unit.
__global__ void __launch_bounds__(256)
cf_post_recs_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t u0, int64_t u1, int64_t e0, int kb,
                    unsigned long long* __restrict__ recs) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = u0 + wave; u < u1; u += n_waves) {
        const int64_t a = cloud_ptr[u], b = cloud_ptr[u + 1];
        for (int64_t e = a + lane; e < b; e += 64) recs[e - e0] = ((unsigned long long)u << kb) | (unsigned long long)(uint32_t)entries[e];
    }
}
// The same for ONE partition of the first k-mers (n_parts > 1: a rank of a multi-GPU run scans all clouds and keeps the entries
// x % n_parts == part, 1 / n_parts of them).  The survivors are compacted in cloud order — per unit a count, a scan, then the fill; one
// wave per unit, ballots inside — so that the stable sort leaves the units of a k-mer ascending, and the sort key is x / n_parts:
// fewer bits, often one radix pass fewer.  (Before: one returning device-scope atomic per kept entry in a histogram pass and again
// in a fill pass, 26 ms per step for a rank of 8 against 19 ms for the whole single-GPU job.)
__global__ void __launch_bounds__(256)
cf_post_ucount_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t u0, int64_t u1, uint32_t part, uint32_t n_parts,
                      uint32_t* __restrict__ ucnt) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = u0 + wave; u < u1; u += n_waves) {
        const int64_t a = cloud_ptr[u], b = cloud_ptr[u + 1];
        uint32_t c = 0;
        for (int64_t e = a + lane; e < b; e += 64) c += (uint32_t)((uint32_t)entries[e] % n_parts == part);
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, (unsigned)d);
        if (lane == 0) ucnt[u - u0] = c;
    }
}
__global__ void __launch_bounds__(256)
cf_post_recs_part_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t u0, int64_t u1, uint32_t part, uint32_t n_parts,
                         const int64_t* __restrict__ uoff, int kb, unsigned long long* __restrict__ recs) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = u0 + wave; u < u1; u += n_waves) {
        const int64_t a = cloud_ptr[u], b = cloud_ptr[u + 1];
        int64_t o = uoff[u - u0];
        for (int64_t e0 = a; e0 < b; e0 += 64) {      // (uniform trip count: a ballot inside)
            const int64_t e = e0 + lane;
            const uint32_t x = e < b ? (uint32_t)entries[e] : 0u;
            const bool keep = e < b && x % n_parts == part;
            const unsigned long long m = cf_ballot(keep);
            if (keep) recs[o + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)u << kb) | (unsigned long long)(x / n_parts);
            o += __popcll(m);
        }
    }
}
// sorted records -> post[] (the units), run starts (rs[x]) / ends (re[x]) of every rank x that occurs, and its first unit
__global__ void __launch_bounds__(256)
cf_post_bounds_kernel(const unsigned long long* __restrict__ recs, int64_t n, int kb, uint32_t part, uint32_t n_parts, int32_t* __restrict__ post,
                      uint32_t* __restrict__ rs, uint32_t* __restrict__ re, uint32_t* __restrict__ first_unit) {
    // (the sort key is the rank itself, or — one partition — rank / n_parts: rank = key * n_parts + part)
    const unsigned long long mask = (1ull << kb) - 1ull;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long r = recs[i];
        const uint32_t j = (uint32_t)(r & mask), u = (uint32_t)(r >> kb), x = j * n_parts + part;
        post[i] = (int32_t)u;
        if (i == 0 || (uint32_t)(recs[i - 1] & mask) != j) { rs[x] = (uint32_t)i; first_unit[x] = u; }
        if (i == n - 1 || (uint32_t)(recs[i + 1] & mask) != j) re[x] = (uint32_t)(i + 1);
    }
}
__global__ void __launch_bounds__(256)
cf_post_counts_kernel(const uint32_t* __restrict__ rs, uint32_t* __restrict__ re_to_cnt, int64_t n_kmers) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_kmers; x += stride) re_to_cnt[x] = re_to_cnt[x] - rs[x];      // (both 0 for a rank that does not occur)
}

__global__ void __launch_bounds__(256)
cf_unit_rend_kernel(const int64_t* __restrict__ unit_ptr, const int64_t* __restrict__ cloud_ptr, int64_t n_reads, int32_t min_d, int32_t max_d,
                    int32_t* __restrict__ rend, int32_t* __restrict__ rbeg, cf_dist_rec* __restrict__ urange) {
    // per unit: its read's unit range and, precomputed once per launch, the partner range a posting at this unit has
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
        const int64_t a = unit_ptr[r], b = unit_ptr[r + 1];
        for (int64_t u = a; u < b; ++u) {
            rend[u] = (int32_t)b; rbeg[u] = (int32_t)a;
            const int64_t jlo = u + min_d, jhi = min(b - 1, u + (int64_t)max_d);
            cf_dist_rec x{0, 0u, (uint32_t)(u - a)};
            if (jhi >= jlo) { x.e0 = cloud_ptr[jlo]; x.len = (uint32_t)(cloud_ptr[jhi + 1] - x.e0); }
            urange[u] = x;
        }
    }
}

// per cloud entry the index of its unit inside its read (one wave per unit): as a 16-bit side array (wide table
// layout) or, mod 2^(32 - b_bits), packed above the rank (narrow layouts: one 4-byte load per pair emission)
__global__ void __launch_bounds__(256)
cf_entry_unit_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ rbeg, const int32_t* __restrict__ entries, int64_t n_units,
                     uint16_t* __restrict__ entry_i, uint32_t* __restrict__ packed, int b_bits, uint8_t* __restrict__ entry_i8) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < n_units; u += n_waves) {
        const uint32_t i = (uint32_t)(u - rbeg[u]);
        if (packed) { for (int64_t e = cloud_ptr[u] + lane; e < cloud_ptr[u + 1]; e += 64) packed[e] = (i << b_bits) | (uint32_t)entries[e]; }      // (the shift drops all but the low 32 - b_bits bits of i)
        else if (entry_i8) { for (int64_t e = cloud_ptr[u] + lane; e < cloud_ptr[u + 1]; e += 64) entry_i8[e] = (uint8_t)i; }
        else { for (int64_t e = cloud_ptr[u] + lane; e < cloud_ptr[u + 1]; e += 64) entry_i[e] = (uint16_t)i; }
    }
}

// keys (first posting unit << 32 | a) of the first k-mers of this partition that have postings
// A wave takes spans of 64 x 16 consecutive k-mers: it counts what it keeps of a span, reserves the output with ONE atomic and then
// writes (the keys are sorted afterwards: their order here is free).  An atomic per 64 k-mers on the one counter was 6.9 ms of an
// emulated rank's 23 ms of set-up (3.7e7 k-mers: 570 000 adds to one address, one after the other).
__global__ void __launch_bounds__(256)
cf_order_keys_kernel(const uint32_t* __restrict__ pcnt, const uint32_t* __restrict__ first_unit, int64_t n_kmers, int part, int n_parts, int ab,
                     unsigned long long* __restrict__ keys, unsigned long long* __restrict__ n_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int kRounds = 16;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int64_t s0 = wave * (64 * kRounds); s0 < n_kmers; s0 += n_waves * (64 * kRounds)) {      // (wave-uniform: ballots inside)
        uint32_t takes = 0, total = 0;      // bit r: this lane keeps its k-mer of round r
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int64_t a = s0 + (int64_t)r * 64 + lane;
            const bool take = a < n_kmers && (a % n_parts) == part && pcnt[a] > 0;
            takes |= (uint32_t)take << r;
            total += (uint32_t)__popcll(cf_ballot(take));
        }
        if (total == 0u) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_out, (unsigned long long)total);
        base = __shfl(base, 0);
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const bool take = (takes >> r) & 1u;
            const unsigned long long m = cf_ballot(take);
            if (take) {
                const int64_t a = s0 + (int64_t)r * 64 + lane;
                keys[base + (unsigned long long)__popcll(m & lt)] = ((unsigned long long)first_unit[a] << ab) | (unsigned long long)a;
            }
            base += (unsigned long long)__popcll(m);
        }
    }
}
__global__ void __launch_bounds__(256)
cf_order_extract_kernel(const unsigned long long* __restrict__ keys, int64_t n, int ab, int32_t* __restrict__ order) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) order[i] = (int32_t)(keys[i] & ((1ull << ab) - 1ull));
}

// One item = up to DIST_ITEM consecutive partner entries of one posting: what one wave takes per step.
//   e  index of its first entry in the cloud-entry arrays (32 bits: cf_dist_edges refuses more than 2^32 - DIST_ITEM entries)
//   m  [entries of the item (1 .. DIST_ITEM) : 16 | unit index of the posting inside its read, mod 65536 : 16]
struct alignas(8) cf_dist_item { uint32_t e, m; };
// One first k-mer of the launch, in processing order: its rank, its number of items and where its item records start.
// The records of a first k-mer are laid out per WAVE of the sweeping workgroup (W waves): item j belongs to wave j % W and is its
// record number j / W; wave w's records are the contiguous run [ibase + w * per, ...), per = ceil(n_items / W).
// (-DCF_DIST_ITEMS_BLOCKED=1, measured and not kept: wave w sweeps the items [w * per, (w + 1) * per), the records lie in item order
// and the fill kernel's stores are nearly coalesced — the set-up gains 0.6 ms of 18 and the kernel loses 15 of 263: dealt round robin
// the waves of a workgroup read neighbouring 1 KB runs of the entry stream at the same time.)
#ifndef CF_DIST_ITEMS_BLOCKED
#define CF_DIST_ITEMS_BLOCKED 0
#endif
struct alignas(16) cf_dist_head { uint32_t a, n_items; unsigned long long ibase; uint32_t n_entries, pad0, pad1, pad2; };   // n_entries: partner entries (capped at 2^30 - 1)

struct cf_dist_args {
    const int64_t* post_ptr;
    const int32_t* post;
    const int64_t* cloud_ptr;
    const int32_t* entries;
    const int32_t* unit_rend;      // one past the last unit of the unit's read
    const int32_t* unit_rbeg;      // first unit of the unit's read
    const cf_dist_rec* urange;     // per unit: the partner range of a posting at this unit
    const uint16_t* entry_i;       // wide layout: per cloud entry the index of its unit inside its read
    const uint8_t* entry_i8;       // region layout: the same index mod 256
    uint32_t reg_shift;            // region layout: log2 of the number of table regions
    const uint32_t* packed;        // narrow layout: per cloud entry [unit index inside its read mod 256 : 8 | rank : 24]
    int64_t n_kmers;
    int32_t part, n_parts;
    int32_t min_d, max_d;       // min_d already clamped to >= 1
    uint32_t min_cov;
    double thr;
    uint32_t thr_num, thr_den;     // thr as an exact fraction when it is the literal 0.8 (4 / 5), else 0 / 0: the dominance test is then integer
    int32_t slots;
    uint32_t fill_limit;
    uint32_t est_limit;            // emissions one partition is expected to hold (fill_limit / expected distinct share)
    int32_t sketch;                // 1: count first in 8-bit counters, build the exact table only for k-mers that can pass min_cov
    uint32_t sk_counters, sk_shift, sk_mask;   // counters (a power of two that fits the LDS that is dead during the sketch sweep), 32 - log2 of it, and it - 1
    uint32_t sk_fbits, sk_fsh, sk_wshift, sk_fmask, sk_guard, sk_bytes;      // a counter's bits (8 or 4), log2 of that, log2 of the counters per word, its largest value, the value from which an add takes itself back (4-bit counters), bytes of the array
    const cf_dist_head* heads;     // the first k-mers of the launch in processing order (cf_items_fill_kernel) ...
    const cf_dist_item* items;     // ... and the item records of their sweeps, laid out per wave of a workgroup of blockDim.x threads
    uint32_t stage_cap;            // <= DIST_STAGE_CAP
    uint32_t hot_cap;              // cap on the filter's list of hot slots (tests: a small one forces the in-scan evaluation)
    uint32_t hot_entries;          // first k-mers with more partner entries than this do not keep the list at all (it would overflow: the filter scans)
    uint32_t* edges;
    unsigned long long edge_cap;
    const int32_t* order;          // first k-mers of this partition, sorted by their first posting (locality)
    int64_t n_order;
    unsigned long long edge_chunk; // edge rows a workgroup reserves per global atomic (DIST_EDGE_CHUNK; tests: small)
    unsigned long long* holes;     // (start, rows) of the unused rest of every workgroup's last chunk of the edge output
    unsigned long long* counters;  // [0] edge rows reserved (chunks) [6] selected edges [7] holes [1] partner entries swept [2] spilled a [3] (host: self pairs) [4] error flags [5] passes; [16 + 16 x] queue head x
    uint32_t* unique_bits;
};

// The 4-byte entry stream of the narrow / region26 layouts: 16 bytes per lane from entry s_e (wave-uniform) on.  Through a BUFFER
// load (round 5): the array's descriptor sits in four scalar registers, the item's byte offset is the instruction's scalar offset and
// the lane's share the constant 16 x lane — no vector instruction per fetch.  As a global load through a pointer the compiler folded
// the lane's offset into a 64-bit pointer per lane and added the item's offset with a v_lshl_add_u64 whose DESTINATION pair it then
// reused for the loaded data: the loop head had to wait for every load in flight before it could write the address
// (s_waitcnt vmcnt(0) once per D + 1 steps).  The scalar offset is 32 bits of bytes: streams of 2^30 entries and more
// take the layouts that stream rank and unit index apart (cf_dist_edges).  (The host emulator of tests/emu defines CF_NO_BUFFER_LOAD.)
typedef uint32_t cf_raw4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) cf_run4 { uint32_t x, y, z, w; };
__device__ __forceinline__ void cf_load_packed4(const cf_dist_args& A, uint32_t s_e, uint32_t l4, uint32_t& x, uint32_t& y, uint32_t& z, uint32_t& w) {
#if !defined(CF_NO_BUFFER_LOAD)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A.packed, 0, (int)0xFFFFFFFFu, 0x00020000);
    const cf_raw4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(l4 * 4u), (int)(s_e << 2), 0);
    x = v.x; y = v.y; z = v.z; w = v.w;
#else
    const uint32_t* p = A.packed + s_e;
    const cf_run4 r = *(const cf_run4*)(p + l4);
    x = r.x; y = r.y; z = r.z; w = r.w;
#endif
}

// ---------------------------------------------------------------------------------------------------
// The (b, d) table lives in LDS and is organised in 32-byte buckets read with two ds_read_b128: a probe
// inspects a whole bucket with straight-line code, so a wave does not run a per-lane probe loop in the
// common case.  A key lives in the first bucket, starting at home(b), that held a match or an empty slot
// when it was inserted; slots of a bucket fill in ascending order and never empty, so every (b, .) key sits
// between home(b) and the first bucket that still has an empty slot, and a key is never inserted twice.
// Two layouts with one interface:
//   cf_tab_wide    4 slots of 64 bits  [b:32 | d:8 | sel:1 | cnt:23]            any k-mer set size (cf_tab_wide16: [b:32 | d:16 | sel:1 | cnt:15] for max_d > 255)
//   cf_tab_narrow  8 keys of 32 bits   [d:8 | b:24] + 8 x 16-bit [sel:1 | cnt-1:15]  (6 bytes per slot: a third
//                  more slots in the same LDS, half as many full buckets, 32-bit compares) when the set has
//                  < 2^24 - 1 k-mers and no k-mer has more than 32767 postings.  A claimed slot counts 1 with
//                  its count field still 0 (most pairs are seen once: no second atomic for them); the cloud
//                  entries are pre-packed [unit index mod 256 : 8 | b : 24], so ONE 4-byte load and one
//                  subtraction give the key; hashing uses 24-bit multiplies (full rate, v_mul_u32_u24).
// b is a dense rank, so one odd multiplier spreads it; the home bucket comes from the high bits of the product.
struct alignas(16) cf_u64x2 { unsigned long long x, y; };
struct alignas(16) cf_u32x4 { uint32_t x, y, z, w; };

// DB = bits of the distance field: 8 (count 23 bits) for max_d <= 255, 16 (count 15 bits) beyond
template <int DB>
struct cf_tab_wide_t {
    static constexpr uint32_t kCntBits = 31 - DB, kCntMask = (1u << kCntBits) - 1u, kDMask = (1u << DB) - 1u, kDShift = 32 - DB;
    static constexpr uint32_t kSlotBytes = 8, kPerBucket = 4;
    struct bucket { cf_u64x2 lo, hi; };
    struct raw { uint32_t b, i; };
    typedef unsigned long long qitem;   // deferred insert: [b:32 | d:8 | bucket to look at next:24]
    static __device__ __forceinline__ qitem q_make(uint32_t b, uint32_t dd, uint32_t bk) { return ((unsigned long long)b << 32) | ((unsigned long long)dd << kDShift) | bk; }
    static __device__ __forceinline__ void q_take(qitem q, uint32_t n_buckets, uint32_t& b, uint32_t& dd, uint32_t& bk) { b = (uint32_t)(q >> 32); dd = ((uint32_t)q >> kDShift) & kDMask; bk = (uint32_t)q & ((1u << kDShift) - 1u); }
    __device__ __forceinline__ qitem q_of(uint32_t b, uint32_t dd, uint32_t n_buckets) const { return q_make(b, dd, home(hash(b), n_buckets)); }
    static __device__ __forceinline__ uint32_t next(uint32_t bk, uint32_t, uint32_t n_buckets) { return bk + 1 == n_buckets ? 0u : bk + 1; }
    __device__ __forceinline__ void configure(const cf_dist_args&, uint32_t) {}
    unsigned long long* tab;
    __device__ __forceinline__ void init(unsigned char* lds, uint32_t) { tab = (unsigned long long*)lds; }
    __device__ __forceinline__ void clear(uint32_t slots, uint32_t t, uint32_t nt) const {
        const cf_u64x2 z{0ull, 0ull};
        for (uint32_t s = t; s < (slots >> 1); s += nt) ((cf_u64x2*)tab)[s] = z;
    }
    // streaming side: one cloud entry -> (b, d)
    // DIST_UNROLL consecutive entries from e on; entries whose bit in ok is clear are not touched in memory
    struct __attribute__((packed, aligned(4))) run4 { uint32_t x, y, z, w; };
    struct __attribute__((packed, aligned(2))) run4h { uint16_t x, y, z, w; };
    // s_e: index of the item's first entry (wave-uniform: the base address is scalar), l4 = 4 * lane
    static __device__ __forceinline__ void load_run(const cf_dist_args& A, uint32_t s_e, uint32_t l4, uint32_t ok, raw (&out)[DIST_UNROLL]) {
        static_assert(DIST_UNROLL == 4, "one 16-byte and one 8-byte load per lane");
        const int32_t* pe = A.entries + s_e;
        const uint16_t* pi = A.entry_i + s_e;
        if (ok == (1u << DIST_UNROLL) - 1u) {      // the whole run lies inside the posting's range (all items but the last of a posting)
            const run4 r = *(const run4*)(pe + l4);
            const run4h h = *(const run4h*)(pi + l4);
            out[0] = raw{r.x, h.x}; out[1] = raw{r.y, h.y}; out[2] = raw{r.z, h.z}; out[3] = raw{r.w, h.w};
            return;
        }
#pragma unroll
        for (int u = 0; u < DIST_UNROLL; ++u) { const uint32_t x = ((ok >> u) & 1u) ? l4 + (uint32_t)u : 0u; out[u] = raw{(uint32_t)pe[x], (uint32_t)pi[x]}; }
    }
    static __device__ __forceinline__ void decode(const raw& r, uint32_t ig, uint32_t, uint32_t& b, uint32_t& dd, uint32_t& qk, uint32_t& lo) { b = r.b; dd = (r.i - ig) & 0xFFFFu; qk = 0u; lo = r.b; }
    __device__ __forceinline__ qitem q_push(uint32_t b, uint32_t dd, uint32_t, uint32_t n_buckets) const { return q_of(b, dd, n_buckets); }   // unit indices are kept mod 65536 and d <= max_d < 65536: the 16-bit difference IS d
    // 24 x 24-bit multiplies only (full rate; a 32-bit multiply or a multiply-high is quarter rate): the low 24 bits of b as in
    // the narrow layout, the high 8 bits through a second multiplier
    static __device__ __forceinline__ uint32_t hash(uint32_t b) { return (b & 0xFFFFFFu) * 0x9E3779u + (b >> 24) * 0x85EBCBu; }
    static __device__ __forceinline__ uint32_t home(uint32_t h, uint32_t n_buckets) { return ((h >> 16) * (n_buckets & 0xFFFFu)) >> 16; }
    static __device__ __forceinline__ uint32_t bm_bit(uint32_t b) { return b & (DIST_BM_BITS - 1u); }      // (k-mer ranks: their low bits are as good as a hash of them, and cost nothing)
    static __device__ __forceinline__ uint32_t sk_hash(uint32_t b, uint32_t dd, uint32_t) { return hash(b) + dd * 0x5BD1E9u; }      // the sketch's counter of the pair (b, d): its high bits
    __device__ __forceinline__ bucket read(uint32_t bk) const { return bucket{*(const cf_u64x2*)&tab[4 * bk], *(const cf_u64x2*)&tab[4 * bk + 2]}; }
    static __device__ __forceinline__ bool is(unsigned long long v, uint32_t b, uint32_t dd) { return (uint32_t)(v >> 32) == b && ((uint32_t)v >> kDShift) == dd; }
    // branch-free: one bit per slot, then find-first-set (nested ?: chains compile to a cascade of exec-mask branches)
    static __device__ __forceinline__ int match(const bucket& k, uint32_t b, uint32_t dd) {   // dd >= 1: an empty slot never matches
        const uint32_t m = (uint32_t)is(k.lo.x, b, dd) | ((uint32_t)is(k.lo.y, b, dd) << 1) | ((uint32_t)is(k.hi.x, b, dd) << 2) | ((uint32_t)is(k.hi.y, b, dd) << 3);
        return __ffs((int)m) - 1;
    }
    static __device__ __forceinline__ int empty(const bucket& k) {
        const uint32_t m = (uint32_t)(k.lo.x == 0ull) | ((uint32_t)(k.lo.y == 0ull) << 1) | ((uint32_t)(k.hi.x == 0ull) << 2) | ((uint32_t)(k.hi.y == 0ull) << 3);
        return __ffs((int)m) - 1;
    }
    // counts one more occurrence of the key in slot i of bucket bk; returns its count after the add
    __device__ __forceinline__ uint32_t add(uint32_t bk, int i) const { return ((uint32_t)atomicAdd(&tab[4 * bk + i], 1ull) & kCntMask) + 1u; }
    static constexpr uint32_t kSlotsPerBucket = 4;
    static constexpr bool kProbe1 = false;      // (64-bit slots: the drain keeps its match / claim branches)
    __device__ __forceinline__ void probe1(bool, uint32_t, uint32_t, uint32_t, bool&, bool&, uint32_t&) const {}
    __device__ __forceinline__ void probe2(bool, uint32_t, uint32_t, uint32_t, bool, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, bool&, bool&, uint32_t&, bool&, bool&, uint32_t&) const {}
    __device__ __forceinline__ uint32_t key_of(uint32_t, uint32_t) const { return 0u; }
    // claim slot i of bucket bk for (b, dd): 0 = claimed (count 1), 1 = the same key got there first (counted), 2 = another key
    __device__ __forceinline__ unsigned long long claim_issue(uint32_t bk, int i, uint32_t b, uint32_t dd) const {
        return atomicCAS(&tab[4 * bk + i], 0ull, ((unsigned long long)b << 32) | ((unsigned long long)dd << kDShift) | 1ull);
    }
    __device__ __forceinline__ int claim_finish(unsigned long long old, uint32_t bk, int i, uint32_t b, uint32_t dd, uint32_t& cnt) const {
        cnt = 1u;
        if (old == 0ull) return 0;
        if (is(old, b, dd)) { cnt = add(bk, i); return 1; }
        return 2;
    }
    // filter side: slot s -> (b, dd, cnt) or false when empty
    __device__ __forceinline__ bool get(uint32_t s, uint32_t& b, uint32_t& dd, uint32_t& cnt) const {
        const unsigned long long v = tab[s];
        b = (uint32_t)(v >> 32); dd = ((uint32_t)v >> kDShift) & kDMask; cnt = (uint32_t)v & kCntMask;
        return v != 0ull;
    }
    __device__ __forceinline__ unsigned long long total_of(uint32_t b, uint32_t n_buckets) const {
        unsigned long long total = 0;
        uint32_t bk = home(hash(b), n_buckets);
        for (uint32_t probe = 0; probe < n_buckets; ++probe) {
            const bucket k = read(bk);
            if ((uint32_t)(k.lo.x >> 32) == b && k.lo.x) total += k.lo.x & (unsigned long long)kCntMask;
            if ((uint32_t)(k.lo.y >> 32) == b && k.lo.y) total += k.lo.y & (unsigned long long)kCntMask;
            if ((uint32_t)(k.hi.x >> 32) == b && k.hi.x) total += k.hi.x & (unsigned long long)kCntMask;
            if ((uint32_t)(k.hi.y >> 32) == b && k.hi.y) total += k.hi.y & (unsigned long long)kCntMask;
            if (empty(k) >= 0) break;
            bk = bk + 1 == n_buckets ? 0u : bk + 1;
        }
        return total;
    }
    // filter side: f(slot, b, dd, cnt, sum over d of cnt(b, .)) for the occupied slots of bucket bk whose count is at least min_cov
    template <class F>
    __device__ __forceinline__ void for_counts_at_least(uint32_t bk, uint32_t n_buckets, uint32_t min_cov, F&& f) const {
        const bucket k = read(bk);
        const unsigned long long v[4] = {k.lo.x, k.lo.y, k.hi.x, k.hi.y};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t cnt = (uint32_t)v[i] & kCntMask;
            if (v[i] != 0ull && cnt >= min_cov) f(4u * bk + (uint32_t)i, (uint32_t)(v[i] >> 32), ((uint32_t)v[i] >> kDShift) & kDMask, cnt, total_of((uint32_t)(v[i] >> 32), n_buckets));
        }
    }
    // filter, two steps: the slots of bucket bk whose count reaches min_cov as a bit mask ...
    static constexpr uint32_t kScanGroup = 4;      // slots per step of the scan (one bucket)
    static __device__ __forceinline__ uint32_t slot_of_bit(uint32_t bit) { return bit; }      // hot_mask: bit i = slot i of the group
    __device__ __forceinline__ uint32_t hot_mask(uint32_t bk, uint32_t min_cov) const {
        const bucket k = read(bk);
        const unsigned long long v[4] = {k.lo.x, k.lo.y, k.hi.x, k.hi.y};
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) m |= (uint32_t)(v[i] != 0ull && ((uint32_t)v[i] & kCntMask) >= min_cov) << i;
        return m;
    }
    // ... and one such slot evaluated: f(slot, b, dd, cnt, sum over d of cnt(b, .))
    template <class F>
    __device__ __forceinline__ void eval_slot(uint32_t s, uint32_t n_buckets, uint32_t min_cov, F&& f) const {
        const unsigned long long v = tab[s];
        const uint32_t cnt = (uint32_t)v & kCntMask;
        if (v != 0ull && cnt >= min_cov) f(s, (uint32_t)(v >> 32), ((uint32_t)v >> kDShift) & kDMask, cnt, total_of((uint32_t)(v >> 32), n_buckets));
    }
    template <class F>
    __device__ __forceinline__ void for_marked(uint32_t bk, F&& f) const {
        const bucket k = read(bk);
        const unsigned long long v[4] = {k.lo.x, k.lo.y, k.hi.x, k.hi.y};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((v[i] >> kCntBits) & 1ull) f(4u * bk + (uint32_t)i, (uint32_t)(v[i] >> 32), ((uint32_t)v[i] >> kDShift) & kDMask, (uint32_t)v[i] & kCntMask);
    }
    __device__ __forceinline__ void mark(uint32_t s) const { atomicOr(&tab[s], 1ull << kCntBits); }
};
typedef cf_tab_wide_t<8> cf_tab_wide;
typedef cf_tab_wide_t<16> cf_tab_wide16;

// DB = bits of the distance field (5 .. 8): the key is [d : DB | b : 32 - DB].  d never exceeds min(max_d, units of the longest
// read - 1), so sets of up to 2^27 - 2 k-mers keep the 6-byte slots when the reads are short enough in units (multi-GPU
// runs sweep the union of all ranks' k-mers: 6e7 at 8 x 50 000 reads).  Hashes use the low 24 bits of b only.
#ifndef CF_NARROW_PB
#define CF_NARROW_PB 4      /* keys per bucket of the 6-byte-slot layout: 4 (one 16-byte read of keys; 453 -> 441 ms) or 8 */
#endif
// Two probes of the 6-byte-slot tables side by side (the drain of the table sweep with CF_DIST_DRAIN2: two queued inserts per lane).
// The same straight line as probe1 below, written stage by stage for both inserts — both bucket reads, both slot searches, both
// claims, both count adds — so that the two chains of LDS round trips are in flight together.  Two inserts of one lane may meet in
// one bucket: they then behave like two lanes that do (the second claim of the same key sees the key and counts; another key's parks).
// Round 6, measured and switched off (CF_DIST_PROBE_TWICE): an insert that finds its bucket full without its key (or loses the slot it
// wanted to another key) gets a SECOND straight-line probe at once — of the next bucket of its chain, or the same one again — before it is
// parked.  On cenX-shaped reads 4 % of the inserts park (2.2 trips of the probe loop each, 28 lanes per run: a run takes 4 400 cycles —
// every LDS round trip under this kernel's load is 350-650 cycles; profiles/r06_dist_ab_cenx.log).
#ifndef CF_DIST_PROBE_TWICE
#define CF_DIST_PROBE_TWICE 0      /* measured: 117.5 (every lane) / 118.0 ms (parked lanes only) against 108.0 without on cenX-shaped reads, 249 against 240.6 on the bench's: the second probe's LDS round trips cost every drain more than the parked inserts' loop costs the few */
#endif
template <class Tab>
__device__ __forceinline__ void cf_probe2(const Tab& T, bool act0, uint32_t bk0, uint32_t key0, uint32_t b0, bool act1, uint32_t bk1, uint32_t key1, uint32_t b1, uint32_t n_buckets, uint32_t min_cov,
                                          bool& made0, bool& park0, uint32_t& hot0, bool& made1, bool& park1, uint32_t& hot1) {
    constexpr int PB_ = (int)Tab::kPerBucket;
    bool full0 = false, full1 = false;      // the bucket had neither the key nor a free slot (else a parked insert lost its slot to another key)
    auto stage = [&](bool a0, uint32_t k0b, bool a1, uint32_t k1b, bool& md0, bool& pk0, uint32_t& ht0, bool& md1, bool& pk1, uint32_t& ht1) {
        const typename Tab::bucket k0 = T.read(k0b), k1 = T.read(k1b);
        uint32_t em0 = (uint32_t)PB_, em1 = (uint32_t)PB_;
#pragma unroll
        for (int j = PB_ - 1; j >= 0; --j) { em0 = k0.k[j] == Tab::kEmpty ? (uint32_t)j : em0; em1 = k1.k[j] == Tab::kEmpty ? (uint32_t)j : em1; }
        uint32_t sl0 = em0, sl1 = em1;
        bool mt0 = false, mt1 = false;
#pragma unroll
        for (int j = PB_ - 1; j >= 0; --j) {
            const bool e0 = k0.k[j] == key0, e1 = k1.k[j] == key1;
            sl0 = e0 ? (uint32_t)j : sl0; mt0 |= e0; sl1 = e1 ? (uint32_t)j : sl1; mt1 |= e1;
        }
        mt0 &= a0; mt1 &= a1;
        const bool claim0 = a0 && !mt0 && em0 < (uint32_t)PB_, claim1 = a1 && !mt1 && em1 < (uint32_t)PB_;
        const uint32_t s0 = (uint32_t)PB_ * k0b + (sl0 & (uint32_t)(PB_ - 1)), s1 = (uint32_t)PB_ * k1b + (sl1 & (uint32_t)(PB_ - 1));
        uint32_t old0 = key0, old1 = key1;
        if (claim0) old0 = atomicCAS(&T.keys[s0], Tab::kEmpty, key0);
        if (claim1) old1 = atomicCAS(&T.keys[s1], Tab::kEmpty, key1);
        const bool matched0 = mt0 || (claim0 && old0 == key0), fresh0 = claim0 && old0 == Tab::kEmpty;
        const bool matched1 = mt1 || (claim1 && old1 == key1), fresh1 = claim1 && old1 == Tab::kEmpty;
        const uint32_t sh0 = (s0 & 1u) * 16u, sh1 = (s1 & 1u) * 16u;
        const uint32_t was0 = atomicAdd(&T.cnt32[s0 >> 1], matched0 ? 1u << sh0 : 0u);
        const uint32_t was1 = atomicAdd(&T.cnt32[s1 >> 1], matched1 ? 1u << sh1 : 0u);
        const uint32_t cnt0 = matched0 ? ((was0 >> sh0) & 0x7FFFu) + 2u : 1u, cnt1 = matched1 ? ((was1 >> sh1) & 0x7FFFu) + 2u : 1u;
        md0 = fresh0; pk0 = a0 && !(matched0 || fresh0); ht0 = ((matched0 || fresh0) && cnt0 == min_cov) ? s0 : 0xFFFFFFFFu;
        md1 = fresh1; pk1 = a1 && !(matched1 || fresh1); ht1 = ((matched1 || fresh1) && cnt1 == min_cov) ? s1 : 0xFFFFFFFFu;
        full0 = a0 && !mt0 && em0 >= (uint32_t)PB_; full1 = a1 && !mt1 && em1 >= (uint32_t)PB_;
    };
    stage(act0, bk0, act1, bk1, made0, park0, hot0, made1, park1, hot1);
#if CF_DIST_PROBE_TWICE
    if (park0 | park1) {      // (only the lanes with a parked insert: the others' LDS operations would double the drain's)
        bool md0 = false, pk0 = false, md1 = false, pk1 = false;
        uint32_t ht0 = 0xFFFFFFFFu, ht1 = 0xFFFFFFFFu;
        stage(park0, full0 ? T.next(bk0, b0, n_buckets) : bk0, park1, full1 ? T.next(bk1, b1, n_buckets) : bk1, md0, pk0, ht0, md1, pk1, ht1);
        made0 |= md0; made1 |= md1; hot0 = min(hot0, ht0); hot1 = min(hot1, ht1); park0 = pk0; park1 = pk1;      // (an insert has at most one of the two hot slots: the other is all ones)
    }
#endif
}

template <int DB, int PB = CF_NARROW_PB>
struct cf_tab_narrow_t {
    static_assert(PB == 4 || PB == 8, "a bucket is one or two 16-byte reads of keys");
    static constexpr uint32_t kBBits = 32 - DB, kBMask = (1u << kBBits) - 1u;
    static constexpr uint32_t kSlotBytes = 6, kPerBucket = PB, kAll = (1u << PB) - 1u;
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    struct bucket { uint32_t k[PB]; };
    struct counts { uint32_t w[PB / 2]; };      // PB x 16-bit [sel:1 | count - 1 : 15]
    struct raw { uint32_t v; };
    typedef uint32_t qitem;             // deferred insert: the key; probing restarts at the home bucket
    static __device__ __forceinline__ qitem q_make(uint32_t b, uint32_t dd, uint32_t) { return key_of(b, dd); }
    static __device__ __forceinline__ void q_take(qitem q, uint32_t n_buckets, uint32_t& b, uint32_t& dd, uint32_t& bk) { b = q & kBMask; dd = q >> kBBits; bk = home(hash(b), n_buckets); }
    __device__ __forceinline__ qitem q_of(uint32_t b, uint32_t dd, uint32_t) const { return key_of(b, dd); }
    static __device__ __forceinline__ uint32_t next(uint32_t bk, uint32_t, uint32_t n_buckets) { return bk + 1 == n_buckets ? 0u : bk + 1; }
    __device__ __forceinline__ void configure(const cf_dist_args&, uint32_t) {}
    uint32_t* keys;     // slots x 32-bit [d : DB | b : 32 - DB]
    uint32_t* cnt32;    // slots x 16-bit [sel:1 | count - 1 : 15], two per word
    __device__ __forceinline__ void init(unsigned char* lds, uint32_t slots) { keys = (uint32_t*)lds; cnt32 = keys + slots; }
    __device__ __forceinline__ void clear(uint32_t slots, uint32_t t, uint32_t nt) const {
        const cf_u32x4 e{kEmpty, kEmpty, kEmpty, kEmpty}, z{0u, 0u, 0u, 0u};
        for (uint32_t s = t; s < (slots >> 2); s += nt) ((cf_u32x4*)keys)[s] = e;
        for (uint32_t s = t; s < (slots >> 3); s += nt) ((cf_u32x4*)cnt32)[s] = z;
    }
    // DIST_UNROLL (= 4) consecutive entries from e on with ONE 16-byte load (4-byte aligned); the packed array is
    // padded by DIST_ITEM entries, so a run that starts inside the array may be read whole whatever ok says
    struct __attribute__((packed, aligned(4))) run4 { uint32_t x, y, z, w; };
    static __device__ __forceinline__ void load_run(const cf_dist_args& A, uint32_t s_e, uint32_t l4, uint32_t, raw (&out)[DIST_UNROLL]) {
        static_assert(DIST_UNROLL == 4, "one 16-byte load per lane");
        cf_load_packed4(A, s_e, l4, out[0].v, out[1].v, out[2].v, out[3].v);
    }
    // the unit index is kept mod 2^DB and 1 <= d < 2^DB, so the DB-bit difference IS d; no borrow reaches b
    static __device__ __forceinline__ void decode(const raw& r, uint32_t ig, uint32_t, uint32_t& b, uint32_t& dd, uint32_t& qk, uint32_t& lo) { const uint32_t q = r.v - (ig << kBBits); b = q & kBMask; dd = q >> kBBits; qk = q; lo = r.v; }      // (lo: a word whose low 16 bits are b's — the bitmap's index — that costs nothing to make)
    // (the queued insert IS the difference decode() made: [d | b]; rebuilt from b and d it cost three more instructions per entry)
    __device__ __forceinline__ qitem q_push(uint32_t, uint32_t, uint32_t qk, uint32_t) const { return qk; }
    static __device__ __forceinline__ uint32_t hash(uint32_t b) { return (b & 0xFFFFFFu) * 0x9E3779u; }               // 24 x 24 -> low 32 bits
    static __device__ __forceinline__ uint32_t home(uint32_t h, uint32_t n_buckets) { return ((h >> 16) * (n_buckets & 0xFFFFu)) >> 16; }
    static __device__ __forceinline__ uint32_t bm_bit(uint32_t b) { return b & (DIST_BM_BITS - 1u); }      // (k-mer ranks: their low bits are as good as a hash of them, and cost nothing)
    // the sketch's counter of the pair (b, d), from the decoded difference q = [d | b] itself: one multiply-add (b's low 24 bits times the
    // constant, plus q: d lands in the counter index's high bits, so the pairs of one b never share a counter)
    static __device__ __forceinline__ uint32_t sk_hash(uint32_t, uint32_t, uint32_t qk) { return (qk & 0xFFFFFFu) * 0x9E3779u + qk; }
    __device__ __forceinline__ bucket read(uint32_t bk) const {
        bucket r;
#pragma unroll
        for (int q = 0; q < PB / 4; ++q) { const cf_u32x4 v = *(const cf_u32x4*)&keys[PB * bk + 4 * q]; r.k[4 * q] = v.x; r.k[4 * q + 1] = v.y; r.k[4 * q + 2] = v.z; r.k[4 * q + 3] = v.w; }
        return r;
    }
    __device__ __forceinline__ counts read_counts(uint32_t bk) const {
        counts c;
        if constexpr (PB == 8) { const cf_u32x4 v = *(const cf_u32x4*)&cnt32[4 * bk]; c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w; }
        else { const unsigned long long v = *(const unsigned long long*)&cnt32[2 * bk]; c.w[0] = (uint32_t)v; c.w[1] = (uint32_t)(v >> 32); }
        return c;
    }
    static __device__ __forceinline__ uint32_t field(const counts& c, int j) { return (c.w[j >> 1] >> ((j & 1) * 16)) & 0x7FFFu; }      // count - 1 of slot j
    static __device__ __forceinline__ uint32_t key_of(uint32_t b, uint32_t dd) { return (dd << kBBits) | b; }
    // branch-free: one bit per slot, then find-first-set (nested ?: chains compile to a cascade of exec-mask branches)
    static __device__ __forceinline__ uint32_t ne_bit(uint32_t k, uint32_t q) { return min(k ^ q, 1u); }
    static __device__ __forceinline__ int first_equal(const bucket& k, uint32_t q) {
        uint32_t ne = 0;
#pragma unroll
        for (int j = 0; j < PB; ++j) ne |= ne_bit(k.k[j], q) << j;
        return __ffs((int)(ne ^ kAll)) - 1;
    }
    static __device__ __forceinline__ int match(const bucket& k, uint32_t b, uint32_t dd) { return first_equal(k, key_of(b, dd)); }
    static __device__ __forceinline__ int empty(const bucket& k) { return first_equal(k, kEmpty); }
    // counts one more occurrence of the key in slot i of bucket bk; returns its count after the add (the field holds count - 1)
    __device__ __forceinline__ uint32_t add(uint32_t bk, int i) const {
        const uint32_t s = PB * bk + (uint32_t)i, sh_ = (s & 1u) * 16u;
        return ((atomicAdd(&cnt32[s >> 1], 1u << sh_) >> sh_) & 0x7FFFu) + 2u;
    }
    static constexpr uint32_t kSlotsPerBucket = PB;
    __device__ __forceinline__ uint32_t claim_issue(uint32_t bk, int i, uint32_t b, uint32_t dd) const { return atomicCAS(&keys[PB * bk + i], kEmpty, key_of(b, dd)); }
    __device__ __forceinline__ int claim_finish(uint32_t old, uint32_t bk, int i, uint32_t b, uint32_t dd, uint32_t& cnt) const {
        cnt = 1u;
        if (old == kEmpty) return 0;                        // claimed: the zero count field already means "seen once"
        if (old == key_of(b, dd)) { cnt = add(bk, i); return 1; }  // the same key was claimed by someone else: count it
        return 2;
    }
    // ONE probe of the key's home bucket in straight-line code (the drain of the table sweep): match -> count it; no match and a
    // free slot -> claim the first one (the zero count field of a fresh slot already means "seen once"; the same key claimed by
    // another lane a moment ago counts as a match); anything else -> park.  Flags are integers made by selects, the count add is
    // issued by every lane (adding 0 where there is nothing to count): as nested ifs that set booleans on divergent paths the
    // drain compiled to ~115 scalar instructions of exec-mask bookkeeping per 64 inserts (profiles/r04_dist_phase_insts.md:
    // the inserts were 0.27 of the kernel's 0.58 scalar instructions per pair).  EVERY lane of the wave runs it (active = the lane has
    // an insert): under `if (lane < n)` the flags crossed the join in vector registers and came back through compares.
    static constexpr bool kProbe1 = true;
    __device__ __forceinline__ void probe1(bool active, uint32_t bk, uint32_t key, uint32_t min_cov, bool& made, bool& park, uint32_t& hot_slot) const {
        const bucket k = read(bk);
        // the slot of the key, else the first empty one, else PB: two chains of selects over the slots (a compare and a v_cndmask each;
        // as bit masks + find-first-set the two searches took 26 instructions instead of 16)
        uint32_t em = (uint32_t)PB;
#pragma unroll
        for (int j = PB - 1; j >= 0; --j) em = k.k[j] == kEmpty ? (uint32_t)j : em;
        uint32_t sl = em;
        bool mt = false;
#pragma unroll
        for (int j = PB - 1; j >= 0; --j) { const bool e = k.k[j] == key; sl = e ? (uint32_t)j : sl; mt |= e; }
        mt &= active;      // (a lane without an insert — only the last drain of a sweep has any — reads some bucket and changes nothing)
        const bool claim = active && !mt && em < (uint32_t)PB;
        const uint32_t s = PB * bk + (sl & (uint32_t)(PB - 1));      // (nothing to do in a full bucket without the key: slot 0's address for the adds of 0 below)
        uint32_t old = key;
        if (claim) old = atomicCAS(&keys[s], kEmpty, key);
        const bool matched = mt || (claim && old == key);
        const bool fresh = claim && old == kEmpty;
        const uint32_t sh_ = (s & 1u) * 16u;
        const uint32_t was = atomicAdd(&cnt32[s >> 1], matched ? 1u << sh_ : 0u);
        const uint32_t cnt = matched ? ((was >> sh_) & 0x7FFFu) + 2u : 1u;
        made = fresh;
        park = active && !(matched || fresh);
        hot_slot = ((matched || fresh) && cnt == min_cov) ? s : 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void probe2(bool act0, uint32_t bk0, uint32_t key0, uint32_t b0, bool act1, uint32_t bk1, uint32_t key1, uint32_t b1, uint32_t n_buckets, uint32_t min_cov,
                                           bool& made0, bool& park0, uint32_t& hot0, bool& made1, bool& park1, uint32_t& hot1) const {
        cf_probe2(*this, act0, bk0, key0, b0, act1, bk1, key1, b1, n_buckets, min_cov, made0, park0, hot0, made1, park1, hot1);
    }
    __device__ __forceinline__ bool get(uint32_t s, uint32_t& b, uint32_t& dd, uint32_t& cnt) const {
        const uint32_t q = keys[s];
        b = q & kBMask; dd = q >> kBBits; cnt = ((cnt32[s >> 1] >> ((s & 1u) * 16u)) & 0x7FFFu) + 1u;
        return q != kEmpty;
    }
    // sum of the counts of the keys (b, .) among the PB keys of a bucket
    static __device__ __forceinline__ uint32_t sum_of(const bucket& k, const counts& c, uint32_t b) {
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if ((k.k[j] & kBMask) == b && k.k[j] != kEmpty) total += field(c, j) + 1u;
        return total;
    }
    __device__ __forceinline__ unsigned long long total_of(uint32_t b, uint32_t n_buckets) const {
        unsigned long long total = 0;
        uint32_t bk = home(hash(b), n_buckets);
        for (uint32_t probe = 0; probe < n_buckets; ++probe) {
            const bucket k = read(bk);
            total += sum_of(k, read_counts(bk), b);
            if (k.k[PB - 1] == kEmpty) break;       // slots fill in ascending order: an empty last slot = the chain ends here
            bk = bk + 1 == n_buckets ? 0u : bk + 1;
        }
        return total;
    }
    template <class F>
    __device__ __forceinline__ void for_counts_at_least(uint32_t bk, uint32_t n_buckets, uint32_t min_cov, F&& f) const {
        const counts c = read_counts(bk);
        const uint32_t need = min_cov ? min_cov - 1u : 0u;     // on the stored field (count - 1)
        bool any = false;
#pragma unroll
        for (int j = 0; j < PB; ++j) any |= field(c, j) >= need;
        if (!any) return;
        const bucket k = read(bk);
        const bool open_bucket = k.k[PB - 1] == kEmpty;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const uint32_t cnt = field(c, i) + 1u;
            if (cnt >= min_cov && k.k[i] != kEmpty) {
                const uint32_t b = k.k[i] & kBMask;
                // usual case: b lives in its home bucket and the bucket is not full, so all (b, .) keys are in the
                // registers already — no chain walk through LDS
                const unsigned long long total = (open_bucket && home(hash(b), n_buckets) == bk) ? (unsigned long long)sum_of(k, c, b) : total_of(b, n_buckets);
                f((uint32_t)PB * bk + (uint32_t)i, b, k.k[i] >> kBBits, cnt, total);
            }
        }
    }
    // filter, two steps: the slots of bucket bk whose count field reaches min_cov - 1 as a bit mask (an empty slot has the
    // field 0: with min_cov <= 1 it is in the mask and eval_slot drops it) ...
    static constexpr uint32_t kScanGroup = 8;      // slots per step of the scan: 16 bytes of count fields, whatever the bucket size
    // two 16-bit fields per word compared at once: (field & 0x7FFF) + (0x8000 - need) has bit 15 set iff the field reaches need
    // (no carry leaves a half: both terms are <= 0x8000).  Bit 2j of the result = slot 2j of the group, bit 16 + 2j = slot 2j + 1.
    __device__ __forceinline__ uint32_t hot_mask(uint32_t g, uint32_t min_cov) const {
        const cf_u32x4 v = *(const cf_u32x4*)&cnt32[4 * g];
        const uint32_t need = min(min_cov ? min_cov - 1u : 0u, 0x8000u);
        const uint32_t add = 0x80008000u - need * 0x00010001u;
        const uint32_t y0 = ((v.x & 0x7FFF7FFFu) + add) & 0x80008000u, y1 = ((v.y & 0x7FFF7FFFu) + add) & 0x80008000u;
        const uint32_t y2 = ((v.z & 0x7FFF7FFFu) + add) & 0x80008000u, y3 = ((v.w & 0x7FFF7FFFu) + add) & 0x80008000u;
        return (y0 >> 15) | (y1 >> 13) | (y2 >> 11) | (y3 >> 9);
    }
    static __device__ __forceinline__ uint32_t slot_of_bit(uint32_t bit) { return (bit & 15u) + (bit >> 4); }
    // ... and one such slot evaluated: f(slot, b, dd, cnt, sum over d of cnt(b, .)).  Usual case: b lives in its home bucket
    // and the bucket is not full, so all (b, .) keys are among the keys just read — no chain walk through LDS.
    template <class F>
    __device__ __forceinline__ void eval_slot(uint32_t s, uint32_t n_buckets, uint32_t min_cov, F&& f) const {
        const uint32_t bk = s / (uint32_t)PB, i = s % (uint32_t)PB;
        const bucket k = read(bk);
        const counts c = read_counts(bk);
        uint32_t mine = k.k[0], mine_f = field(c, 0);
#pragma unroll
        for (int j = 1; j < PB; ++j) { if (i == (uint32_t)j) { mine = k.k[j]; mine_f = field(c, j); } }
        const uint32_t cnt = mine_f + 1u;
        if (mine == kEmpty || cnt < min_cov) return;
        const uint32_t b = mine & kBMask;
        const unsigned long long total = (k.k[PB - 1] == kEmpty && home(hash(b), n_buckets) == bk) ? (unsigned long long)sum_of(k, c, b) : total_of(b, n_buckets);
        f(s, b, mine >> kBBits, cnt, total);
    }
    template <class F>
    __device__ __forceinline__ void for_marked(uint32_t bk, F&& f) const {
        const counts c = read_counts(bk);
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < PB / 2; ++j) any |= c.w[j];
        if (!(any & 0x80008000u)) return;       // no selected slot in the bucket
        const bucket k = read(bk);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const uint32_t h = c.w[i >> 1] >> ((i & 1) * 16);
            if (h & 0x8000u) f((uint32_t)PB * bk + (uint32_t)i, k.k[i] & kBMask, k.k[i] >> kBBits, (h & 0x7FFFu) + 1u);
        }
    }
    __device__ __forceinline__ void mark(uint32_t s) const { atomicOr(&cnt32[s >> 1], 0x8000u << ((s & 1u) * 16u)); }
};
typedef cf_tab_narrow_t<8> cf_tab_narrow;

// The 6-byte slots for k-mer sets of up to 2^27 - 16 k-mers whatever the reads' lengths (distances up to 255): the 32-bit key
// [d : 8 | b >> S : 24] leaves out the low S <= 3 bits of the rank, and the table is cut into 2^S REGIONS of equal size —
// a key lives in the region of its rank's low bits (home bucket, probe chain and the chain walk for the totals all stay
// inside the region), so (region, key) identifies (b, d).  Ranks are dense, so the regions fill evenly.  The cloud entries
// are read as they are (the CSR's 32-bit ranks) plus one byte per entry for the unit index mod 256; a queued insert is
// 64 bits (b, d).  Everything else — 4 keys per bucket, 16-bit count fields, the filter — is the 6-byte layout's.
struct cf_tab_region {
    static constexpr int PB = 4;
    static constexpr uint32_t kBBits = 24, kBMask = (1u << kBBits) - 1u;
    static constexpr uint32_t kSlotBytes = 6, kPerBucket = PB, kAll = (1u << PB) - 1u;
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    struct bucket { uint32_t k[PB]; };
    struct counts { uint32_t w[PB / 2]; };
    struct raw { uint32_t b, i; };
    typedef unsigned long long qitem;   // deferred insert: [b : 32 | d : 32]
    uint32_t* keys;
    uint32_t* cnt32;
    uint32_t S, nb_r;                   // log2(regions), buckets per region
    __device__ __forceinline__ void init(unsigned char* lds, uint32_t slots) { keys = (uint32_t*)lds; cnt32 = keys + slots; S = 0; nb_r = slots / PB; }
    __device__ __forceinline__ void configure(const cf_dist_args& A, uint32_t n_buckets) { S = A.reg_shift; nb_r = n_buckets >> A.reg_shift; }
    __device__ __forceinline__ void clear(uint32_t slots, uint32_t t, uint32_t nt) const {
        const cf_u32x4 e{kEmpty, kEmpty, kEmpty, kEmpty}, z{0u, 0u, 0u, 0u};
        for (uint32_t s = t; s < (slots >> 2); s += nt) ((cf_u32x4*)keys)[s] = e;
        for (uint32_t s = t; s < (slots >> 3); s += nt) ((cf_u32x4*)cnt32)[s] = z;
    }
    struct __attribute__((packed, aligned(4))) run4 { uint32_t x, y, z, w; };
    static __device__ __forceinline__ void load_run(const cf_dist_args& A, uint32_t s_e, uint32_t l4, uint32_t ok, raw (&out)[DIST_UNROLL]) {
        static_assert(DIST_UNROLL == 4, "one 16-byte and one 8-byte load per lane");
        const int32_t* pe = A.entries + s_e;
        if (ok == (1u << DIST_UNROLL) - 1u) {      // the whole run lies inside the posting's range
            const run4 r = *(const run4*)(pe + l4);
            // the 4 index bytes start at any byte address: two ALIGNED dwords around them and a funnel shift (a dword load
            // from a misaligned address takes the slow path of the memory pipeline: the sketch sweep ran 39 % longer with it);
            // the misalignment is the item's (wave-uniform)
            struct __attribute__((packed, aligned(4))) pair2 { uint32_t lo, hi; };
            const uint8_t* pb = A.entry_i8 + (s_e & ~3u);
            const pair2 w = *(const pair2*)(pb + l4);
            const uint32_t sh = (s_e & 3u) * 8u;
            const uint32_t iw = (uint32_t)((((unsigned long long)w.hi << 32) | w.lo) >> sh);
            out[0] = raw{r.x, iw & 0xFFu}; out[1] = raw{r.y, (iw >> 8) & 0xFFu}; out[2] = raw{r.z, (iw >> 16) & 0xFFu}; out[3] = raw{r.w, iw >> 24};
            return;
        }
        const uint8_t* pi = A.entry_i8 + s_e;
#pragma unroll
        for (int u = 0; u < DIST_UNROLL; ++u) { const uint32_t x = ((ok >> u) & 1u) ? l4 + (uint32_t)u : 0u; out[u] = raw{(uint32_t)pe[x], (uint32_t)pi[x]}; }
    }
    static __device__ __forceinline__ void decode(const raw& r, uint32_t ig, uint32_t, uint32_t& b, uint32_t& dd, uint32_t& qk, uint32_t& lo) { b = r.b; dd = (r.i - ig) & 0xFFu; qk = 0u; lo = r.b; }      // unit indices mod 256, d <= 255
    __device__ __forceinline__ qitem q_push(uint32_t b, uint32_t dd, uint32_t, uint32_t n_buckets) const { return q_of(b, dd, n_buckets); }
    static __device__ __forceinline__ uint32_t hash(uint32_t b) { return (b & 0xFFFFFFu) * 0x9E3779u; }      // sketch and bitmap: any function of b will do (ranks that differ above bit 23 share counters and bits)
    static __device__ __forceinline__ uint32_t bm_bit(uint32_t b) { return b & (DIST_BM_BITS - 1u); }      // (k-mer ranks: their low bits are as good as a hash of them, and cost nothing)
    static __device__ __forceinline__ uint32_t sk_hash(uint32_t b, uint32_t dd, uint32_t) { return hash(b) + dd * 0x5BD1E9u; }      // the sketch's counter of the pair (b, d): its high bits
    __device__ __forceinline__ uint32_t key_of(uint32_t b, uint32_t dd) const { return (dd << kBBits) | (b >> S); }
    __device__ __forceinline__ uint32_t region_base(uint32_t b) const { return (b & ((1u << S) - 1u)) * nb_r; }
    __device__ __forceinline__ uint32_t home_of(uint32_t b) const { return region_base(b) + (((((b >> S) * 0x9E3779u) >> 16) * (nb_r & 0xFFFFu)) >> 16); }
    __device__ __forceinline__ uint32_t next(uint32_t bk, uint32_t b, uint32_t) const { const uint32_t r0 = region_base(b); return bk + 1 == r0 + nb_r ? r0 : bk + 1; }
    __device__ __forceinline__ uint32_t region_of(uint32_t bk) const {      // bk / nb_r for at most 8 regions
        uint32_t r = 0;
        for (uint32_t x = nb_r; x <= bk; x += nb_r) ++r;
        return r;
    }
    __device__ __forceinline__ uint32_t b_of(uint32_t key, uint32_t bk) const { return ((key & kBMask) << S) | region_of(bk); }
    __device__ __forceinline__ qitem q_of(uint32_t b, uint32_t dd, uint32_t) const { return ((unsigned long long)b << 32) | dd; }
    __device__ __forceinline__ void q_take(qitem q, uint32_t, uint32_t& b, uint32_t& dd, uint32_t& bk) const { b = (uint32_t)(q >> 32); dd = (uint32_t)q; bk = home_of(b); }
    __device__ __forceinline__ bucket read(uint32_t bk) const {
        bucket r;
        const cf_u32x4 v = *(const cf_u32x4*)&keys[PB * bk];
        r.k[0] = v.x; r.k[1] = v.y; r.k[2] = v.z; r.k[3] = v.w;
        return r;
    }
    __device__ __forceinline__ counts read_counts(uint32_t bk) const {
        counts c;
        const unsigned long long v = *(const unsigned long long*)&cnt32[2 * bk];
        c.w[0] = (uint32_t)v; c.w[1] = (uint32_t)(v >> 32);
        return c;
    }
    static __device__ __forceinline__ uint32_t field(const counts& c, int j) { return (c.w[j >> 1] >> ((j & 1) * 16)) & 0x7FFFu; }
    static __device__ __forceinline__ uint32_t ne_bit(uint32_t k, uint32_t q) { return min(k ^ q, 1u); }
    static __device__ __forceinline__ int first_equal(const bucket& k, uint32_t q) {
        uint32_t ne = 0;
#pragma unroll
        for (int j = 0; j < PB; ++j) ne |= ne_bit(k.k[j], q) << j;
        return __ffs((int)(ne ^ kAll)) - 1;
    }
    __device__ __forceinline__ int match(const bucket& k, uint32_t b, uint32_t dd) const { return first_equal(k, key_of(b, dd)); }
    static __device__ __forceinline__ int empty(const bucket& k) { return first_equal(k, kEmpty); }
    // counts one more occurrence of the key in slot i of bucket bk; returns its count after the add (the field holds count - 1)
    __device__ __forceinline__ uint32_t add(uint32_t bk, int i) const {
        const uint32_t s = PB * bk + (uint32_t)i, sh_ = (s & 1u) * 16u;
        return ((atomicAdd(&cnt32[s >> 1], 1u << sh_) >> sh_) & 0x7FFFu) + 2u;
    }
    static constexpr uint32_t kSlotsPerBucket = PB;
    __device__ __forceinline__ uint32_t claim_issue(uint32_t bk, int i, uint32_t b, uint32_t dd) const { return atomicCAS(&keys[PB * bk + i], kEmpty, key_of(b, dd)); }
    __device__ __forceinline__ int claim_finish(uint32_t old, uint32_t bk, int i, uint32_t b, uint32_t dd, uint32_t& cnt) const {
        cnt = 1u;
        if (old == kEmpty) return 0;
        if (old == key_of(b, dd)) { cnt = add(bk, i); return 1; }
        return 2;
    }
    static constexpr bool kProbe1 = true;      // (cf_tab_narrow_t::probe1: the drain's one probe in straight-line code)
    __device__ __forceinline__ void probe1(bool active, uint32_t bk, uint32_t key, uint32_t min_cov, bool& made, bool& park, uint32_t& hot_slot) const {
        const bucket k = read(bk);
        // the slot of the key, else the first empty one, else PB: two chains of selects over the slots (a compare and a v_cndmask each;
        // as bit masks + find-first-set the two searches took 26 instructions instead of 16)
        uint32_t em = (uint32_t)PB;
#pragma unroll
        for (int j = PB - 1; j >= 0; --j) em = k.k[j] == kEmpty ? (uint32_t)j : em;
        uint32_t sl = em;
        bool mt = false;
#pragma unroll
        for (int j = PB - 1; j >= 0; --j) { const bool e = k.k[j] == key; sl = e ? (uint32_t)j : sl; mt |= e; }
        mt &= active;      // (a lane without an insert — only the last drain of a sweep has any — reads some bucket and changes nothing)
        const bool claim = active && !mt && em < (uint32_t)PB;
        const uint32_t s = PB * bk + (sl & (uint32_t)(PB - 1));      // (nothing to do in a full bucket without the key: slot 0's address for the adds of 0 below)
        uint32_t old = key;
        if (claim) old = atomicCAS(&keys[s], kEmpty, key);
        const bool matched = mt || (claim && old == key);
        const bool fresh = claim && old == kEmpty;
        const uint32_t sh_ = (s & 1u) * 16u;
        const uint32_t was = atomicAdd(&cnt32[s >> 1], matched ? 1u << sh_ : 0u);
        const uint32_t cnt = matched ? ((was >> sh_) & 0x7FFFu) + 2u : 1u;
        made = fresh;
        park = active && !(matched || fresh);
        hot_slot = ((matched || fresh) && cnt == min_cov) ? s : 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void probe2(bool act0, uint32_t bk0, uint32_t key0, uint32_t b0, bool act1, uint32_t bk1, uint32_t key1, uint32_t b1, uint32_t n_buckets, uint32_t min_cov,
                                           bool& made0, bool& park0, uint32_t& hot0, bool& made1, bool& park1, uint32_t& hot1) const {
        cf_probe2(*this, act0, bk0, key0, b0, act1, bk1, key1, b1, n_buckets, min_cov, made0, park0, hot0, made1, park1, hot1);
    }
    __device__ __forceinline__ bool get(uint32_t s, uint32_t& b, uint32_t& dd, uint32_t& cnt) const {
        const uint32_t q = keys[s];
        b = b_of(q, s / (uint32_t)PB); dd = q >> kBBits; cnt = ((cnt32[s >> 1] >> ((s & 1u) * 16u)) & 0x7FFFu) + 1u;
        return q != kEmpty;
    }
    // sum of the counts of the keys (b, .) among the keys of a bucket of b's region
    __device__ __forceinline__ uint32_t sum_of(const bucket& k, const counts& c, uint32_t b) const {
        const uint32_t bq = b >> S;
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if ((k.k[j] & kBMask) == bq && k.k[j] != kEmpty) total += field(c, j) + 1u;
        return total;
    }
    __device__ __forceinline__ unsigned long long total_of(uint32_t b, uint32_t n_buckets) const {
        unsigned long long total = 0;
        uint32_t bk = home_of(b);
        for (uint32_t probe = 0; probe < nb_r; ++probe) {
            const bucket k = read(bk);
            total += sum_of(k, read_counts(bk), b);
            if (k.k[PB - 1] == kEmpty) break;
            bk = next(bk, b, n_buckets);
        }
        return total;
    }
    template <class F>
    __device__ __forceinline__ void for_counts_at_least(uint32_t bk, uint32_t n_buckets, uint32_t min_cov, F&& f) const {
        const counts c = read_counts(bk);
        const uint32_t need = min_cov ? min_cov - 1u : 0u;
        bool any = false;
#pragma unroll
        for (int j = 0; j < PB; ++j) any |= field(c, j) >= need;
        if (!any) return;
        const bucket k = read(bk);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const uint32_t cnt = field(c, i) + 1u;
            if (cnt >= min_cov && k.k[i] != kEmpty) {
                const uint32_t b = b_of(k.k[i], bk);
                const unsigned long long total = (k.k[PB - 1] == kEmpty && home_of(b) == bk) ? (unsigned long long)sum_of(k, c, b) : total_of(b, n_buckets);
                f((uint32_t)PB * bk + (uint32_t)i, b, k.k[i] >> kBBits, cnt, total);
            }
        }
    }
    static constexpr uint32_t kScanGroup = 8;
    // two 16-bit fields per word compared at once: (field & 0x7FFF) + (0x8000 - need) has bit 15 set iff the field reaches need
    // (no carry leaves a half: both terms are <= 0x8000).  Bit 2j of the result = slot 2j of the group, bit 16 + 2j = slot 2j + 1.
    __device__ __forceinline__ uint32_t hot_mask(uint32_t g, uint32_t min_cov) const {
        const cf_u32x4 v = *(const cf_u32x4*)&cnt32[4 * g];
        const uint32_t need = min(min_cov ? min_cov - 1u : 0u, 0x8000u);
        const uint32_t add = 0x80008000u - need * 0x00010001u;
        const uint32_t y0 = ((v.x & 0x7FFF7FFFu) + add) & 0x80008000u, y1 = ((v.y & 0x7FFF7FFFu) + add) & 0x80008000u;
        const uint32_t y2 = ((v.z & 0x7FFF7FFFu) + add) & 0x80008000u, y3 = ((v.w & 0x7FFF7FFFu) + add) & 0x80008000u;
        return (y0 >> 15) | (y1 >> 13) | (y2 >> 11) | (y3 >> 9);
    }
    static __device__ __forceinline__ uint32_t slot_of_bit(uint32_t bit) { return (bit & 15u) + (bit >> 4); }
    template <class F>
    __device__ __forceinline__ void eval_slot(uint32_t s, uint32_t n_buckets, uint32_t min_cov, F&& f) const {
        const uint32_t bk = s / (uint32_t)PB, i = s % (uint32_t)PB;
        const bucket k = read(bk);
        const counts c = read_counts(bk);
        uint32_t mine = k.k[0], mine_f = field(c, 0);
#pragma unroll
        for (int j = 1; j < PB; ++j) { if (i == (uint32_t)j) { mine = k.k[j]; mine_f = field(c, j); } }
        const uint32_t cnt = mine_f + 1u;
        if (mine == kEmpty || cnt < min_cov) return;
        const uint32_t b = b_of(mine, bk);
        const unsigned long long total = (k.k[PB - 1] == kEmpty && home_of(b) == bk) ? (unsigned long long)sum_of(k, c, b) : total_of(b, n_buckets);
        f(s, b, mine >> kBBits, cnt, total);
    }
    template <class F>
    __device__ __forceinline__ void for_marked(uint32_t bk, F&& f) const {
        const counts c = read_counts(bk);
        if (!((c.w[0] | c.w[1]) & 0x80008000u)) return;
        const bucket k = read(bk);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const uint32_t h = c.w[i >> 1] >> ((i & 1) * 16);
            if (h & 0x8000u) f((uint32_t)PB * bk + (uint32_t)i, b_of(k.k[i], bk), k.k[i] >> kBBits, (h & 0x7FFFu) + 1u);
        }
    }
    __device__ __forceinline__ void mark(uint32_t s) const { atomicOr(&cnt32[s >> 1], 0x8000u << ((s & 1u) * 16u)); }
};

// The region layout's table behind a FOUR-byte stream, for k-mer sets of 2^24 .. 2^26 ranks whose reads have at most 128 units
// (8 x 50 000 reads sharded over 8 GPUs sweep a union of 3.7e7 k-mers: tools/rank_emulation.py).  26 bits of rank leave 6 for
// the unit index — one bit short of the 7 that distances up to 127 need, which is why round 3 streamed rank and index apart
// (a 16-byte and an 8-byte load per lane and step, a funnel shift, 64-bit queue items: 450 ms against 320 at 50 000 reads).
// The partner entries of a posting are sorted by unit, so inside an item of 256 entries "d >= 64" begins at ONE position,
// t64, that the item record carries next to the posting's unit index ([entries : 16 | t64 : 9 | unit index : 7]):
// d = ((i - ig) mod 64) + 64 * [position >= t64].  Table, keys, regions, filter: cf_tab_region's.
struct cf_tab_region26 : cf_tab_region {
    struct raw { uint32_t v; };
    struct __attribute__((packed, aligned(4))) run4 { uint32_t x, y, z, w; };
    static __device__ __forceinline__ void load_run(const cf_dist_args& A, uint32_t s_e, uint32_t l4, uint32_t, raw (&out)[DIST_UNROLL]) {
        static_assert(DIST_UNROLL == 4, "one 16-byte load per lane");
        cf_load_packed4(A, s_e, l4, out[0].v, out[1].v, out[2].v, out[3].v);
    }
    static __device__ __forceinline__ void decode(const raw& r, uint32_t igf, uint32_t pos, uint32_t& b, uint32_t& dd, uint32_t& qk, uint32_t& lo) {
        qk = 0u; lo = r.v;
        b = r.v & 0x3FFFFFFu;
        dd = (((r.v >> 26) - igf) & 63u) + (pos >= (igf >> 7) ? 64u : 0u);      // (igf = t64 << 7 | unit index mod 128: its low 6 bits are all the subtraction keeps)
    }
};


// lanes below this one whose bit is set in a ballot: v_mbcnt_lo + v_mbcnt_hi (2 instructions; popcount(mask & lanemask_lt) is 4)
// bit (off mod 32) of w in ONE instruction: v_bfe_u32 takes the low five bits of its offset operand, which the compiler does not use
// (it masks the offset first, or shifts and masks); the host emulator (tests/emu) defines the macro of the same name
#ifndef cf_bit_of
__device__ __forceinline__ uint32_t cf_bit_of(uint32_t w, uint32_t off) { uint32_t r; asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "v"(off)); return r; }
#endif
__device__ __forceinline__ uint32_t cf_rank_in(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

#define DIST_QCAP 384                    /* deferred inserts per wave: fewer than 128 (64 without CF_DIST_DRAIN2) left by the drains before a step + at most 4 x 64 pushed by it */
#define DIST_QSTRIDE DIST_QCAP
// Diagnostic builds only (tools/dist_ablation.sh): -DCF_DIST_ABL=n removes the kernel's phases from the END — 1 no filter / rows, 2 also no
// inserts (the drains drop their queue), 3 no pushes, 4 no table sweep, 5 no sketch arithmetic, 6 no sketch sweep, 7 no clears — so that
// the instruction counters of successive builds differ by ONE phase (results are wrong, what runs before the cut is unchanged)
#ifndef CF_DIST_ABL
#define CF_DIST_ABL 0
#endif
// Round 5: three changes to the insert path of the table sweep, each behind a switch for A/B builds (tools/build_variant.sh):
//  CF_DIST_HOTBLK  the slots whose count reaches min_cov go to the filter's list through blocks of 64 entries that a WAVE reserves with
//                  one LDS atomic per 64 entries (cursor and block end live in scalar registers) — it was one returning LDS atomic,
//                  a wait and a readfirstlane per DRAIN (nearly every drain of 64 inserts brings some slot to min_cov);
//  CF_DIST_FILLRD  the table's fill level is read INSIDE the drain (the load is issued with the queue read and looked at after the
//                  bucket came back: no round trip of its own) instead of twice in front of every drain;
//  CF_DIST_DRAIN2  a drain takes up to 128 queued inserts, two per lane: two independent chains (queue word, bucket, claim, count)
//                  in flight per lane — the inserts are bound by that chain of LDS round trips at four waves per SIMD, not by
//                  instruction issue (profiles/r04_dist_phase_insts_after_cuts.md).
#ifndef CF_DIST_HOTBLK
#define CF_DIST_HOTBLK 0      /* measured (gpurun_out/c1_dist_ab.log): 257.4 ms with the blocks, 251.3 without — the unused entries of every wave's last block lengthen the filter's list by up to 64 x waves entries */
#endif
#ifndef CF_DIST_PUSH_FLAT
#define CF_DIST_PUSH_FLAT 0
#endif
#ifndef CF_DIST_FILLRD
#define CF_DIST_FILLRD 1
#endif
#ifndef CF_DIST_DRAIN2
#define CF_DIST_DRAIN2 1
#endif
#ifndef CF_DIST_OLD_DRAIN
#define CF_DIST_OLD_DRAIN 0      /* 1: the drain's probe as nested match / claim branches (rounds 2-3), for A/B runs */
#endif
#define DIST_FULL_BIT 0x80000000u        /* sh[0]: the table is physically full (the pass is void and will be split) */

// general insert: walk buckets from bk; claims the first empty slot with a CAS when the key is absent.
// Returns 1 when a new key was created; hot := the slot when this occurrence brought its count to exactly min_cov.
template <class Tab>
__device__ __forceinline__ uint32_t cf_dist_insert(const Tab& T, uint32_t n_buckets, uint32_t bk, uint32_t b, uint32_t dd, uint32_t* sh, uint32_t min_cov, uint32_t& hot, unsigned long long* trips = nullptr) {
    for (uint32_t tries = 0; tries < 9 * n_buckets; ++tries) {
        if (trips) ++*trips;
        const typename Tab::bucket k = T.read(bk);
        const int m = T.match(k, b, dd);
        if (m >= 0) { if (T.add(bk, m) == min_cov) hot = Tab::kSlotsPerBucket * bk + (uint32_t)m; return 0u; }
        const int e = Tab::empty(k);
        if (e >= 0) {
            uint32_t cnt;
            const int st = T.claim_finish(T.claim_issue(bk, e, b, dd), bk, e, b, dd, cnt);
            if (st != 2 && cnt == min_cov) hot = Tab::kSlotsPerBucket * bk + (uint32_t)e;
            if (st == 0) return 1u;
            if (st == 1) return 0u;
            continue;   // another key took the slot: look at the same bucket again
        }
        bk = T.next(bk, b, n_buckets);
    }
    atomicOr(&sh[0], DIST_FULL_BIT);
    return 0u;
}

// Diagnostic build only (-DCF_DIST_STAMPS, tools/dist_stamps.py): per-phase shader-clock sums of thread 0 of every
// workgroup into counters[8..15]; the shipped library compiles these to nothing.
#if defined(CF_DIST_STAMPS)
#define CF_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)
#else
#define CF_STAMP(i) do { } while (0)
#endif

// Round 3: the work lists of the sweeps are built ONCE per launch, in HBM, by two small kernels (one wave per first k-mer),
// instead of once per first k-mer inside cf_dist_kernel.  There every workgroup loaded its postings and their partner ranges
// (two dependent round trips), scanned the item counts, searched the posting of every item and wrote the records to LDS behind
// three barriers: 8 % of the kernel's time, 6.5 KB of its LDS, and every step of both sweeps first had to find its posting
// (round 2: 25 vector + 40 scalar instructions per step).  Now a wave reads its 8-byte records with one coalesced load.
// Both kernels give every first k-mer a GROUP of 16 lanes (most have fewer than 16 postings): four first k-mers per wave are in
// flight at once — the work per first k-mer is a chain of dependent loads (order -> post_ptr -> post -> urange), so the kernels are
// bound by how many chains run side by side.
// cf_items_count_kernel also counts the partner entries that are the first k-mer itself: the pairs of its postings (u, v) in one
// read with min_d <= v - u <= max_d (the reference skips a == b, distance_based_kmer_recruitment.py:118; the sweeps count such an
// entry like any other and the host subtracts this sum).
__global__ void __launch_bounds__(256)
cf_items_count_kernel(const int32_t* __restrict__ order, int64_t n_order, const int64_t* __restrict__ post_ptr, const int32_t* __restrict__ post,
                      const cf_dist_rec* __restrict__ urange, const int32_t* __restrict__ rbeg, int32_t min_d, int32_t max_d, uint32_t nw, int sorted_post,
                      uint32_t* __restrict__ n_items, uint32_t* __restrict__ n_alloc, unsigned long long* __restrict__ self_pairs) {
    const int gl = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t n_grp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    unsigned long long self = 0;
    // (the first two links of an iteration's chain — order -> post_ptr — are fetched one iteration ahead)
    int64_t p0_n = 0, p1_n = 0;
    { const int32_t a0 = grp < n_order ? order[grp] : 0; if (grp < n_order) { p0_n = post_ptr[a0]; p1_n = post_ptr[a0 + 1]; } }
    for (int64_t i0 = 0; i0 < n_order; i0 += n_grp) {      // (uniform trip count: shuffles inside)
        const int64_t i = i0 + grp;
        const bool on = i < n_order;
        const int64_t p0 = p0_n, p1 = p1_n;
        { const int64_t in = i + n_grp; p0_n = 0; p1_n = 0; if (in < n_order) { const int32_t an = order[in]; p0_n = post_ptr[an]; p1_n = post_ptr[an + 1]; } }
        unsigned long long c = 0;
        int64_t np_max = p1 - p0;      // the longest posting list among the wave's four groups bounds the loops
        for (int d = 16; d <= 32; d <<= 1) np_max = max(np_max, __shfl_xor(np_max, d));
        for (int64_t x0 = 0; x0 < np_max; x0 += 16) {
            const bool hx = p0 + x0 + gl < p1;
            const int32_t ux = hx ? post[p0 + x0 + gl] : 0;
            cf_dist_rec rx_{0, 0u, 0u};
            if (hx) rx_ = urange[ux];
            const int32_t rx = hx ? ux - (int32_t)rx_.ig : -1;      // first unit of the posting's read (ig = the unit's index in its read): ONE gather per posting
            if (hx) c += (rx_.len + DIST_ITEM - 1u) / DIST_ITEM;
            if (np_max < 2) continue;
            if (sorted_post && np_max <= 16) {
                // (round 6) the postings of a k-mer are in ascending unit order (when they were made by the sort: not by the atomics' fill pass), so two of them in ONE read stand next to each other: when
                // no posting of the wave's four first k-mers has its successor in its own read — the usual case: a rare k-mer occurs once per
                // read — there is no pair to count, and the 16 rounds of shuffles below are skipped (with the mask-and-shift of cf_items_fill_kernel: set-up 17.9 -> 17.2 ms)
                const int32_t r_next = __shfl(rx, (gl + 1) & 15, 16);
                const bool adj = hx && gl < 15 && p0 + gl + 1 < p1 && r_next == rx;
                if (!cf_ballot(adj)) continue;
            }
            for (int64_t y0 = 0; y0 < np_max; y0 += 16) {
                const bool hy = p0 + y0 + gl < p1;
                int32_t uy_l = ux, ry_l = hx ? rx : -2;      // (the same 16 postings: nearly every first k-mer has at most 16)
                if (y0 != x0) { uy_l = hy ? post[p0 + y0 + gl] : 0; ry_l = hy ? uy_l - (int32_t)urange[uy_l].ig : -2; }
                for (int j = 0; j < 16; ++j) {
                    const int32_t uy = __shfl(uy_l, j, 16), ry = __shfl(ry_l, j, 16);
                    const int32_t d = uy - ux;
                    self += (unsigned long long)(hx && ry == rx && d >= min_d && d <= max_d);
                }
            }
        }
        for (int d = 8; d >= 1; d >>= 1) c += __shfl_down(c, (unsigned)d, 16);
        if (on && gl == 0) {
            const uint32_t n = (uint32_t)min(c, 0xFFFFFFFFull - 2048ull);
            n_items[i] = n;
            n_alloc[i] = ((n + nw - 1u) / nw) * nw;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) self += __shfl_down(self, (unsigned)d);
    if ((threadIdx.x & 63) == 0 && self) atomicAdd(self_pairs, self);
}

__global__ void __launch_bounds__(256)
cf_items_fill_kernel(const int32_t* __restrict__ order, int64_t n_order, const int64_t* __restrict__ post_ptr, const int32_t* __restrict__ post,
                     const cf_dist_rec* __restrict__ urange, uint32_t nw, const uint32_t* __restrict__ n_items, const int64_t* __restrict__ ibase,
                     cf_dist_head* __restrict__ heads, cf_dist_item* __restrict__ items, const int64_t* __restrict__ cloud_ptr64, int64_t n_units) {
    // cloud_ptr64 != null: the 26-bit stream (cf_tab_region26): the item record's low half is [t64 : 9 | unit index mod 128 : 7]
    const int gl = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t n_grp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const bool nw_pow2 = (nw & (nw - 1u)) == 0u;
    const uint32_t nw_log2 = 31u - (uint32_t)__clz((int)max(nw, 1u));
    int32_t a_n = 0; int64_t p0_n = 0, p1_n = 0, ib_n = 0; uint32_t n_n = 0;      // (fetched one iteration ahead, as in cf_items_count_kernel)
    if (grp < n_order) { a_n = order[grp]; p0_n = post_ptr[a_n]; p1_n = post_ptr[a_n + 1]; n_n = n_items[grp]; ib_n = ibase[grp]; }
    for (int64_t i0 = 0; i0 < n_order; i0 += n_grp) {      // (uniform trip count: shuffles inside)
        const int64_t i = i0 + grp;
        const bool on = i < n_order;
        const int32_t a = a_n;
        const int64_t p0 = p0_n, p1 = p1_n;
        const uint32_t n = n_n, per = (n + nw - 1u) / nw;
        const int64_t ib = ib_n;
        { const int64_t in = i + n_grp; a_n = 0; p0_n = 0; p1_n = 0; n_n = 0; ib_n = 0;
          if (in < n_order) { a_n = order[in]; p0_n = post_ptr[a_n]; p1_n = post_ptr[a_n + 1]; n_n = n_items[in]; ib_n = ibase[in]; } }
        cf_dist_item* out = items + ib;
        int64_t np_max = p1 - p0;
        for (int d = 16; d <= 32; d <<= 1) np_max = max(np_max, __shfl_xor(np_max, d));
        uint32_t j0 = 0;      // items of the postings before this chunk (group-uniform)
        unsigned long long ne = 0;
        for (int64_t q0 = 0; q0 < np_max; q0 += 16) {      // 16 postings at a time, one per lane of the group
            cf_dist_rec r{0, 0u, 0u};
            int64_t e64 = 0;      // (26-bit stream) the first partner entry whose unit lies 64 or more behind the posting's
            if (p0 + q0 + gl < p1) {
                const int32_t ux = post[p0 + q0 + gl];
                r = urange[ux];
                if (cloud_ptr64) { const int64_t c64 = (int64_t)ux + 64 <= n_units ? cloud_ptr64[(int64_t)ux + 64] : r.e0 + (int64_t)r.len; e64 = min(max(c64, r.e0), r.e0 + (int64_t)r.len); }
            }
            const uint32_t c = (r.len + DIST_ITEM - 1u) / DIST_ITEM;
            uint32_t inc = c;
            for (int d = 1; d < 16; d <<= 1) { const uint32_t o = __shfl_up(inc, (unsigned)d, 16); if (gl >= d) inc += o; }
            const uint32_t jb = j0 + inc - c;      // the lane's first item
            for (uint32_t x = 0; x < c; ++x) {
                const uint32_t j = jb + x, off = x * DIST_ITEM;
                uint32_t low = r.ig & 0xFFFFu;
                if (cloud_ptr64) low = ((uint32_t)min(max(e64 - (r.e0 + (int64_t)off), (int64_t)0), (int64_t)DIST_ITEM) << 7) | (r.ig & 127u);
                // (round 6: the waves of a workgroup are a power of two in every launch shape the library picks — a mask and a shift
                // instead of two divisions by a run-time number per item record, 6.9e8 of them at 50 000 reads)
                const uint32_t jw = nw_pow2 ? (j & (nw - 1u)) : j % nw, jq = nw_pow2 ? (j >> nw_log2) : j / nw;
                out[CF_DIST_ITEMS_BLOCKED ? (size_t)j : (size_t)jw * per + jq] = cf_dist_item{(uint32_t)r.e0 + off, (min(r.len - off, DIST_ITEM) << 16) | low};
            }
            j0 += (uint32_t)__shfl((int)inc, 15, 16);
            unsigned long long l = r.len;
            for (int d = 8; d >= 1; d >>= 1) l += __shfl_down(l, (unsigned)d, 16);
            ne += (unsigned long long)__shfl((long long)l, 0, 16);
        }
        if (on && gl == 0) heads[i] = cf_dist_head{(uint32_t)a, n, (unsigned long long)ib, (uint32_t)min(ne, 0x3FFFFFFFull), 0u, 0u, 0u};
    }
}

// Sweeps one wave's items: recs = the wave's records in HBM, mine = how many, my0 = the first 64 of them already in registers
// (lane j: the j-th; loaded at the top of the first k-mer's iteration and used by both sweeps).  A step gets its record with
// two v_readlane into SGPRs — the base address of the step's global load is scalar, the per-lane offset is the constant
// 16 * lane.  Software pipeline: D loads in flight per lane.
// pre(final) runs at ONE site before every step and once more (final = true) after the wave's last step: the table sweep drains
// its insert queue there (one copy of that code in the loop instead of one per push site).
// body(bb, dd, qk, lo, ok, len): the decoded entries of a step; ok = per-lane bit mask of the entries that exist (only the last item of
// a posting has lanes past its end: len < DIST_ITEM, wave-uniform, says whether ok needs looking at); returns true to stop the wave.
// (Round 6, measured and removed: the waves of a workgroup SHARING their items — per list one LDS word [items taken from its end | items its
// owner claimed], the owner's claim in flight under the step before, a wave out of items taking half of what the fullest list has left with
// one record load.  Thread 0's stamps had put 22 % of the kernel on cenX-shaped reads (9 % on the bench's) into waiting for the other waves
// at the end of a sweep; with sharing: 114-116 ms against 109.9 without on those reads, 259-271 against 243.3 on the bench's, whatever the
// thresholds (profiles/r06_dist_ab_steal.log) — the wait is not idle hardware: a wave that is through leaves its issue slots and LDS
// cycles to the others, and a thief's first step waits for two round trips to memory.)
template <class Tab, int D, class Pre, class Body>
__device__ __forceinline__ void cf_dist_sweep(const cf_dist_args& A, const cf_dist_item* recs, uint32_t mine, const cf_dist_item& my0, Pre&& pre, Body&& body) {
    const uint32_t lane = threadIdx.x & 63u, l4 = lane * DIST_UNROLL;
    auto ok_of = [&](uint32_t m) -> uint32_t {      // per-lane mask of the entries of an item that exist
        uint32_t ok = (1u << DIST_UNROLL) - 1u;
        if ((m >> 16) < DIST_ITEM) {      // (wave-uniform) the last item of a posting
            ok = 0;
#pragma unroll
            for (int u = 0; u < DIST_UNROLL; ++u) ok |= (uint32_t)(l4 + (uint32_t)u < (m >> 16)) << u;
        }
        return ok;
    };
    // Round 6: pre(false) in front of every step and pre(true) ONCE behind the loop over the windows (rounds 3-5 computed "is this the
    // wave's last step" inside the loop and passed it to the one call site): 251.5 -> 243.3 ms on the bench's reads, 115.4 -> 109.9 on
    // cenX-shaped ones — found while measuring the work sharing above, whose loop had this shape.
    for (uint32_t j0 = 0; j0 < mine; j0 += 64u) {      // (one window whenever the wave has at most 64 items)
        const uint32_t cnt = min(64u, mine - j0);
        cf_dist_item my = my0;
        if (j0) { my = cf_dist_item{0u, 0u}; if (lane < cnt) my = recs[j0 + lane]; }
        // D loads in flight per lane WHILE a step is worked on: D + 1 register sets, the loop unrolled D + 1 times so that every set has
        // fixed registers — at position p the step's data is set p, and the load of the step D ahead goes into set (p + D) mod (D + 1), the
        // one the previous position has just finished with.  (Rounds 3-4 kept D sets, copied the step's set aside and re-filled it at once:
        // the compiler made the copy a chain of v_mov at the END of the unrolled body, behind an s_waitcnt vmcnt(0) — every wave waited
        // out the loads it had just issued, every D steps, whatever D was; round 5, read off the ISA.)
        constexpr int NS = D + 1;
        typename Tab::raw ring[NS][DIST_UNROLL];
#define CF_DIST_FETCH(J, SET) {                                                                                 \
            const uint32_t s_e = (uint32_t)__builtin_amdgcn_readlane((int)my.e, (int)(J));                      \
            const uint32_t fm_ = (uint32_t)__builtin_amdgcn_readlane((int)my.m, (int)(J));                      \
            Tab::load_run(A, s_e, l4, ok_of(fm_), ring[SET]);                                                   \
        }
#pragma unroll
        for (int d = 0; d < D; ++d) CF_DIST_FETCH(min((uint32_t)d, cnt - 1u), d)      // (unconditional, see below)
        // the steps of this window: a lambda, so that "the window's items are through" leaves the unrolled loop by a plain return — with a flag
        // tested at the loop head the compiler saw a path from every exit back into the loop and made the head wait for every load
        // in flight (s_waitcnt vmcnt(0) in front of the first fetch of every D + 1 steps)
        const bool stop = [&]() -> bool {
            uint32_t j = 0;
            for (;;) {
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    // The fetch of the step D ahead comes FIRST, in front of pre(): the table sweep's drain runs there, and with a register
                    // set of the ring free at that point the compiler used it for the drain's temporaries — a write to a register that an
                    // earlier load had as its destination, which its wait-count pass could only make safe by waiting for every load in
                    // flight (s_waitcnt vmcnt(0) in every drain).  With all D + 1 sets spoken for the drain gets registers of its own.
                    // Unconditional: a wave past its last item fetches that one again — with a fetch under a condition the number of
                    // loads in flight differs between the paths into the loop head and the compiler waits for ALL of them there.
                    CF_DIST_FETCH(min(j + (uint32_t)D, cnt - 1u), (p + D) % NS)
                    pre(false);
                    if (j == cnt) return false;
                    const uint32_t cm = (uint32_t)__builtin_amdgcn_readlane((int)my.m, (int)j);
                    uint32_t bb[DIST_UNROLL], dd_[DIST_UNROLL], qq_[DIST_UNROLL], lo_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) Tab::decode(ring[p][u], cm & 0xFFFFu, l4 + (uint32_t)u, bb[u], dd_[u], qq_[u], lo_[u]);
                    const uint32_t len = cm >> 16;      // entries of the item (wave-uniform); < DIST_ITEM only for the last item of a posting
                    if (body(bb, dd_, qq_, lo_, len < DIST_ITEM ? ok_of(cm) : (1u << DIST_UNROLL) - 1u, len)) return true;
                    ++j;
                }
            }
        }();
#undef CF_DIST_FETCH
        if (stop) return;
    }
    pre(true);      // after the wave's last step: the table sweep empties its queues
}

#ifndef CF_DIST_SKETCH_BY_B
#define CF_DIST_SKETCH_BY_B 0
#endif
#ifndef CF_DIST_PF_A
#define CF_DIST_PF_A 1      /* loads in flight per lane while a step of the sketch sweep is worked on (round 5, with the pipeline that really keeps them in flight: 1 / 2 / 3 = 250.6 / 251.7 / 257.6 ms) */
#endif
#ifndef CF_DIST_PF_B
#define CF_DIST_PF_B 1      /* ... and of the table sweep (1 / 2 / 3 = 250.7 / 251.7 / 255.5 ms; both at 1: 250.1) */
#endif

// registers per lane: the default bound (1024 threads, one workgroup per CU) gives 128 = four waves per SIMD; diagnostic builds
// ask for (640, 2) = five waves per SIMD (96 registers) or (768, 2) = six (80)
#ifndef CF_DIST_LB_THREADS
#define CF_DIST_LB_THREADS 1024
#define CF_DIST_LB_BLOCKS 1
#endif
template <class Tab>
__global__ void __launch_bounds__(CF_DIST_LB_THREADS, CF_DIST_LB_BLOCKS) cf_dist_kernel(cf_dist_args A) {
    // LDS: [bitmap | sh] at FIXED offsets (the bitmap at 0: its word address is the hash bits themselves, no base to add), then
    // [table | edge stage | partition stack | insert queues].  The table group is dead while the sketch sweep runs, so its
    // 8-bit counters (sk) lie over ALL of it.  (The item records of the sweeps are in HBM: cf_items_fill_kernel.)
    const int t = threadIdx.x, lane = t & 63, nt = blockDim.x;
    uint32_t* bm = (uint32_t*)cf_lds_at(0u);                          // DIST_BM_BITS bits: hash(b) of the k-mers b that may have a selected edge
    uint32_t* sh = bm + DIST_BM_BITS / 32;                     // [0] keys in table | DIST_FULL_BIT [1] first k-mer [2] sp [3] P [4] idx [5,6] queue ticket [7] E of pass [8] selected [9,10] edge base [11] hot-list cursor [12] items of the first k-mer [13] a counter of the sketch wrapped [14,15] where its item records start
    unsigned char* lds_tab = cf_lds_at(DIST_LDS_HEAD);
    Tab T;
    T.init(lds_tab, (uint32_t)A.slots);
    T.configure(A, (uint32_t)A.slots / Tab::kPerBucket);
    uint32_t* sk = (uint32_t*)lds_tab;
    uint16_t* stage = (uint16_t*)(lds_tab + (size_t)A.slots * Tab::kSlotBytes);   // slot indices of the selected edges of a pass
    // per-wave queue of pending inserts: candidates are compacted here and inserted 64 at a time by a full wave.
    // Only its own wave touches it: LDS operations of one wave execute in order, and wavefront-scope fences (no
    // instructions) keep the compiler from moving the queue accesses across the drain; a volatile pointer would turn
    // every push into a FLAT store followed by a full vmcnt wait (it did: 4 per step in the first version).
    uint32_t* stack = (uint32_t*)(stage + DIST_STAGE_CAP + 8);                   // (P, idx) pairs
    typename Tab::qitem* wq0 = (typename Tab::qitem*)(stack + 2 * DIST_STACK);
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6)), nw = (uint32_t)nt >> 6;      // (the wave's number in a scalar register: so are the addresses of its queues)
    typename Tab::qitem* wq = wq0 + (size_t)wv * DIST_QSTRIDE;
    // inserts whose FIRST probe did not finish (bucket full, or another key took the slot it wanted): parked here and run
    // through the probe loop 32 .. 64 at a time (round 2 ran that loop inside every drain: most drains went around twice for
    // one or two of their 64 lanes)
    typename Tab::qitem* ovq = wq0 + (size_t)(nt >> 6) * DIST_QSTRIDE + (size_t)wv * DIST_OVQ;
    uint16_t* hotl = (uint16_t*)(wq0 + (size_t)(nt >> 6) * (DIST_QSTRIDE + DIST_OVQ));      // DIST_HOT_CAP slots whose count reached min_cov during the inserts of the pass
    const uint32_t slots = (uint32_t)A.slots, n_buckets = slots / Tab::kPerBucket;   // slots is a multiple of 8
    unsigned long long acc_E = 0, acc_spill = 0, acc_pass = 0, acc_edges = 0;  // flushed once per workgroup (thread 0)
#if defined(CF_DIST_DIAG_COUNT)      /* diagnostic build: [0] loop trips of parked inserts [1] overflow calls [2] parked inserts run [3] drains [4] inserts drained */
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};      /* [5] shader cycles inside the parked-insert runs [6] inside the drains [7] inside the table sweep, per wave (lane 0) */
#define CF_DG(X) X
#define CF_DG_PTR (&dg[0])
#else
#define CF_DG(X)
#define CF_DG_PTR nullptr
#endif
#if defined(CF_DIST_STAMPS)
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t = __builtin_amdgcn_s_memtime();
#endif

    // 8 ticket queues over 8 contiguous ranges of the locality-sorted first k-mers: workgroups with equal blockIdx % 8
    // (observed to share an XCD, i.e. an L2) drain one range; idle ones steal
    // The ticket of the next first k-mer is taken in two halves: pop_issue() sends the atomic on this workgroup's own queue and
    // returns at once, pop_finish() — a phase later — looks at what came back and, only when that queue is exhausted, tries the
    // others.  (Round 2 waited for the returning atomic right where it was issued, at the top of every iteration: a round trip
    // to the memory side with the whole workgroup behind thread 0 at the next barrier, 9 % of the kernel.)
    const int64_t q_per = (A.n_order + 7) / 8;
    auto pop_issue = [&]() -> unsigned long long { return atomicAdd(&A.counters[16 + 16 * (blockIdx.x & 7)], 1ull); };
    auto pop_finish = [&](unsigned long long q) -> long long {
        const int x0 = (int)(blockIdx.x & 7);
        const int64_t lo0 = x0 * q_per, hi0 = min((int64_t)(x0 + 1) * q_per, A.n_order);
        if (lo0 + (int64_t)q < hi0) return lo0 + (int64_t)q;
        for (int s8 = 1; s8 < 8; ++s8) {
            const int x = (int)((blockIdx.x + s8) & 7);
            const int64_t lo = x * q_per, hi = min((int64_t)(x + 1) * q_per, A.n_order);
            if (lo >= hi) continue;
            const unsigned long long q2 = atomicAdd(&A.counters[16 + 16 * x], 1ull);
            if (lo + (int64_t)q2 < hi) return lo + (int64_t)q2;
        }
        return -1;
    };
    // Thread 0 fetches the NEXT first k-mer's head (ticket -> heads[]: two dependent round trips) while the workgroup works on
    // the current one; the values wait in its registers until the loop comes around.
    long long nx_idx = -1;
    cf_dist_head nx_head{0u, 0u, 0ull, 0u, 0u, 0u, 0u};
    unsigned long long nx_q = 0;
    if (!cf_lds_base_ok()) { if (t == 0) atomicOr(&A.counters[4], 2ull); return; }      // (cf_common.h: cf_lds_at)
    // Barriers.  A first k-mer whose table needs one pass — nearly all — meets the workgroup SEVEN times: [top] the previous
    // one's edge rows are written, [sketch cleared], [sketch swept], [table cleared], [table swept], [filtered], [rows reserved].
    // Round 2 and the first half of round 3 had fifteen (the head of the next first k-mer published between two barriers of
    // its own, the partition stack read between two more at the top of every pass and again to find it empty, a barrier
    // between reading the number of selected edges and reusing its word): with ~10 steps per wave and sweep the waves arrive
    // skewed at every one of them, and thread 0's stamps put a third of the kernel into those waits.  Now the next first
    // k-mer's head goes into words of its own (sh[16..22]) in front of an existing barrier — the last pass's reservation —
    // and the default pass (one partition) is set up in front of the sketch's barrier.
    auto publish_next = [&]() {      // thread 0
        sh[16] = (uint32_t)(unsigned long long)nx_idx; sh[17] = (uint32_t)((unsigned long long)nx_idx >> 32);
        sh[18] = nx_head.a; sh[19] = nx_head.n_items; sh[20] = (uint32_t)nx_head.ibase; sh[21] = (uint32_t)(nx_head.ibase >> 32);
        sh[22] = nx_head.n_entries;      // partner entries (sizes the passes when every b is marked)
    };
    auto pop_pass = [&]() {          // thread 0: the next partition of the stack becomes the pass
        const uint32_t sp = sh[2] - 1; sh[2] = sp; sh[3] = stack[2 * sp]; sh[4] = stack[2 * sp + 1]; sh[0] = 0; sh[7] = 0; sh[8] = 0; sh[11] = 0; sh[30] = 0;
    };
    if (t == 0) {
        nx_idx = pop_finish(pop_issue());
        if (nx_idx >= 0) nx_head = A.heads[nx_idx];
        publish_next();
        sh[26] = 0; sh[27] = 0; sh[28] = 0; sh[29] = 0;      // no chunk of the edge output yet
    }
    __syncthreads();

    while (true) {
        const int64_t ai = (int64_t)(((unsigned long long)sh[17] << 32) | sh[16]);
        if (ai < 0) break;
        const uint32_t a = sh[18], n_items = sh[19], n_ent_a = sh[22];
        // Round 6: the list of slots that reach min_cov (one returning LDS add, a wait and a readfirstlane per drain that has one — on
        // cenX-shaped reads every drain has a dozen) is not kept for a first k-mer that would overflow it anyway: its filter scans the count
        // fields, as it did all along once the list was full (12 of 108 ms there)
        const bool use_hot = n_ent_a <= A.hot_entries;
        // this wave's item records: one coalesced load, in flight while the sketch is cleared; both sweeps run on them
        const uint32_t per_w = (n_items + nw - 1u) / nw;
        const uint32_t mine = CF_DIST_ITEMS_BLOCKED ? (n_items > wv * per_w ? min(per_w, n_items - wv * per_w) : 0u) : (wv < n_items ? (n_items - wv + nw - 1u) / nw : 0u);
        const cf_dist_item* recs = A.items + ((((unsigned long long)sh[21] << 32) | sh[20]) + (unsigned long long)wv * per_w);
        cf_dist_item my0 = cf_dist_item{0u, 0u};
        if ((uint32_t)lane < min(mine, 64u)) my0 = recs[lane];
        if (t == 0) nx_q = pop_issue();                               // next: the ticket (looked at after phase A)
        __syncthreads();      // [top] the rows of the previous first k-mer are out of the table, and everyone has this one's head
        CF_STAMP(0);   // queue pop
        if (n_items == 0u) {
            if (t == 0) { nx_idx = pop_finish(nx_q); if (nx_idx >= 0) nx_head = A.heads[nx_idx]; publish_next(); }
            __syncthreads();
            continue;
        }
        CF_STAMP(1);   // prologue
        // ---- phase A (min_cov >= 2): which k-mers b can have a selected edge at all?  Most (b, d) pairs of a are seen
        // once or twice and can never reach min_cov, but an exact table would have to hold them all.  So first every
        // pair is only COUNTED, in an array of 8-bit counters indexed by hash(b, d) that fills the table's LDS: a
        // counter is never below the count of a pair that maps to it (collisions add), and the add whose returned
        // value shows the counter reaching min_cov marks hash(b) in a bitmap — the last add of a pair with
        // cnt >= min_cov always does.  Phase B then builds the exact table for the marked b only (all their d, so the
        // totals are exact too); unmarked b have no pair with cnt >= min_cov and cannot be selected.  A counter about
        // to wrap (255 -> 0) is seen by the add that wraps it: the first k-mer then falls back to "every b marked".
        bool mark_all = !A.sketch;
        const uint32_t min_cov_m1 = A.min_cov - 1u;      // (the sketch runs with min_cov >= 2)
        // the pass that nearly every first k-mer gets by with: one partition, every marked b in it (set up here, in front of a
        // barrier that is there anyway; "every b marked" below replaces it)
        if (t == 0) { sh[13] = 0; sh[31] = 0; sh[2] = 0; sh[3] = 1; sh[4] = 0; sh[0] = 0; sh[7] = 0; sh[8] = 0; sh[11] = 0; sh[30] = 0; }
        if (A.sketch) {
            // (round 6) the counters' width for THIS attempt: 4-bit counters that wrap — a burst of 16 adds on one counter before the first
            // of them can take itself back: a pair counted 15 times plus a collision — send the first k-mer through the sketch sweep once
            // more with bytes (half as many counters), not straight to "every b marked" with its 4-8 table passes; the marks made so far stay
            // (a mark too many only lets pairs into the exact table).  The two attempts report their wrap in words of their own (sh[13],
            // sh[31]): nobody resets a word that another thread may still be reading.
            uint32_t m_fbits = A.sk_fbits, m_fsh = A.sk_fsh, m_wshift = A.sk_wshift, m_fmask = A.sk_fmask, m_guard = A.sk_guard, m_shift = A.sk_shift, m_flag = 13u;
            for (;;) {
            {
                const cf_u32x4 z{0u, 0u, 0u, 0u};
                if (CF_DIST_ABL < 7) {
                for (uint32_t s = (uint32_t)t; s < (A.sk_bytes >> 4); s += (uint32_t)nt) ((cf_u32x4*)sk)[s] = z;
                if (m_flag == 13u) for (uint32_t s = (uint32_t)t; s < DIST_BM_BITS / 128; s += (uint32_t)nt) ((cf_u32x4*)bm)[s] = z;
                }
                __syncthreads();      // [sketch cleared]
            }
            {
                if (CF_DIST_ABL < 6) cf_dist_sweep<Tab, CF_DIST_PF_A>(A, recs, mine, my0, [](bool) {}, [&](const uint32_t (&bb)[DIST_UNROLL], const uint32_t (&dd_)[DIST_UNROLL], const uint32_t (&qq_)[DIST_UNROLL], const uint32_t (&lo_)[DIST_UNROLL], uint32_t ok, uint32_t len) -> bool {
                    if (CF_DIST_ABL >= 5) { if (bb[0] == 0xFFFFFFF1u && dd_[0] == 77u) sh[13] = 1u; return false; }      // (the loads stay: their data is looked at)
                    // (entries equal to a are counted too: the sketch may only over-count, and the table sweep drops them)
                    uint32_t old_[DIST_UNROLL], sft_[DIST_UNROLL], inc_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) {
                        // (Round 6, measured, off: a counter per b — the low bits of the rank, i.e. of the raw stream word: no decode, no multiply;
                        // Σ_d cnt(a, b, ·) bounds every cnt(a, b, d).  Its collisions NEST with the bitmap's (two counters per bit), and on reads
                        // whose rare k-mers are copy-specific it lets fewer pairs through — 0.355 of the pairs inserted instead of 0.394 on
                        // cenX-shaped reads, 6 100 keys per first k-mer instead of 7 900: 108.0 -> 101.9 ms — but a k-mer that recurs along the
                        // array has pairs at every distance, is marked as a whole, and brings a key per distance into ONE probe chain:
                        // var_len 1 on the same reads 7.75 -> 9.2 ms with 6 x the split tables.  profiles/r06_dist_ab_cenx.log)
#if CF_DIST_SKETCH_BY_B
                        const uint32_t idx = lo_[u] & A.sk_mask;
#else
                        const uint32_t idx = Tab::sk_hash(bb[u], dd_[u], qq_[u]) >> m_shift;
#endif
                        // Round 6: counters of FOUR bits where min_cov allows it (twice as many in the same LDS: on samples of first k-mers 15 % fewer
                        // keys reach the exact table on cenX-shaped reads — 6 800 instead of 7 950 — and 8 % fewer pairs are inserted).  A counter
                        // only has to tell "min_cov - 1 or more": an add that sees 12 or more takes itself back (the counter stays where it is and
                        // every later add still sees >= min_cov - 1), and the add that would wrap a field — it sees 15 — is seen doing it, like the
                        // 255 of the 8-bit counters: the first k-mer falls back to "every b marked".  A carry into the next field only over-counts.
                        sft_[u] = idx << m_fsh;      // (only its low 5 bits are used: the shift and the bit-field extract take them mod 32)
                        inc_[u] = 1u << (sft_[u] & 31u);
                        old_[u] = idx >> m_wshift;
                    }
                    if (len < DIST_ITEM) {      // (wave-uniform) lanes past the end of the posting's range add nothing
#pragma unroll
                        for (int u = 0; u < DIST_UNROLL; ++u) if (!((ok >> u) & 1u)) inc_[u] = 0u;
                    }
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) old_[u] = atomicAdd(&sk[old_[u]], inc_[u]);     // all counter adds of the step back to back
                    uint32_t seen_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) seen_[u] = __builtin_amdgcn_ubfe(old_[u], sft_[u], m_fbits);     // occurrences before this one (v_bfe_u32 takes the offset mod 32)
                    // ONE test per step: few adds reach min_cov (each of the four branches of round 2 cost a compare, three scalar
                    // exec-mask instructions and a jump)
                    static_assert(DIST_UNROLL == 4, "max of four");
                    const uint32_t seen_max = max(max(seen_[0], seen_[1]), max(seen_[2], seen_[3]));
                    if (seen_max >= min_cov_m1) {
                        // The bit is set without looking first (a read, a wait, a test and a second exec mask per entry: the atomic costs the
                        // LDS what the read did), and lanes past the end of a posting's range (inc 0: they read a counter without adding)
                        // are not told apart: a bit too many in the bitmap only lets entries into the exact table that are then not
                        // selected, and a wrap seen by such a lane only sends the first k-mer down the every-b-marked path.
#pragma unroll
                        for (int u = 0; u < DIST_UNROLL; ++u) {
                            if (seen_[u] >= min_cov_m1) {
                                atomicOr(&bm[Tab::bm_bit(lo_[u]) >> 5], 1u << (lo_[u] & 31u));
                            }
                        }
                        if (seen_max >= m_guard) {      // (4-bit counters only: the guard of the 8-bit ones is beyond their range)
#pragma unroll
                            for (int u = 0; u < DIST_UNROLL; ++u)
                                if (seen_[u] >= m_guard && ((ok >> u) & 1u)) atomicSub(&sk[sft_[u] >> 5], 1u << (sft_[u] & 31u));      // (idx << fsh >> 5 = idx >> wshift: the counter's word; the increment is made again: kept in a register it made the kernel spill)
                        }
                        if (seen_max == m_fmask) sh[m_flag] = 1u;
                    }
                    return false;
                });
                CF_STAMP(6);   // sketch sweep (clear + wave 0's own items)
            }
            __syncthreads();      // [sketch swept]
            CF_STAMP(7);      // (stamp 7: thread 0's wait for the other waves at the end of a sweep)
            if (!sh[m_flag]) break;
            if (m_fbits == 8u) { mark_all = true; break; }
            m_fbits = 8u; m_fsh = 3u; m_wshift = 2u; m_fmask = 255u; m_guard = 0x7FFFFFFFu; m_shift = A.sk_shift + 1u; m_flag = 31u;
            }
        }
        if (t == 0) { nx_idx = pop_finish(nx_q); if (nx_idx >= 0) nx_head = A.heads[nx_idx]; }      // next: its head (published with the last pass's reservation)
        if (mark_all) {
            const cf_u32x4 ones{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            for (uint32_t s = (uint32_t)t; s < DIST_BM_BITS / 128; s += (uint32_t)nt) ((cf_u32x4*)bm)[s] = ones;
            if (t == 0) {
                // number of partitions of the exact (b, d) table: every pair may need a slot when all b are marked
                // (upper bound from the emission count); with the bitmap only a fraction does — one, set up above
                uint32_t P0 = 1;
                while (P0 < 64u && (unsigned long long)n_ent_a > (unsigned long long)A.est_limit * P0) P0 <<= 1;
                for (uint32_t i = 0; i < P0; ++i) { stack[2 * i] = P0; stack[2 * i + 1] = i; }
                sh[2] = P0;
                pop_pass();
            }
        }
        bool spilled = false;
        if (CF_DIST_ABL < 7) T.clear(slots, (uint32_t)t, (uint32_t)nt);      // (the sketch is dead: its counters lay over the table)
        __syncthreads();      // [table cleared] (and, when every b is marked, the bitmap filled and the passes set up)
        while (true) {
            const uint32_t P = sh[3], pidx = sh[4], pmask = P - 1u;   // P is a power of two; P == 1: every b belongs to the pass
            uint32_t my_e = 0, s_e = 0;      // partner entries swept by this lane (split passes) / by this wave (whole passes)
            CF_STAMP(2);   // pop partition + clear table
            // ---- phase B: sweep again; pairs of this partition whose b is marked go to the wave's queue and are
            // inserted into the exact table 64 at a time by a full wave
            {
                uint32_t qtail = 0, otail = 0;      // wave-uniform: queued inserts / parked inserts of this wave
                bool too_full = sh[0] > A.fill_limit;      // (another list of this pass may already have filled the table)
                // pops the last N (<= 64) queued inserts, one per lane, and gives each ONE probe of its home bucket in
                // straight-line code: match -> count it; empty slot -> claim it; bucket full or the slot lost to another key ->
                // parked in the overflow list, which goes through the probe loop once it holds 32
                // slots whose count just reached min_cov go to the filter's list (one LDS atomic per wave and drain): the filter then
                // evaluates that list instead of scanning every slot of the table for counts >= min_cov (round 2: 3 scan rounds per
                // first k-mer, 7 % of the kernel)
#if CF_DIST_HOTBLK
                uint32_t hcur = 0, hend = 0;      // wave-uniform: this wave's block of the hot list [hcur, hend)
#define CF_DIST_HOT(SLOT) {                                                                                   \
                    const unsigned long long hm_ = cf_ballot((SLOT) != 0xFFFFFFFFu);                           \
                    if (hm_) {                                                                                \
                        const uint32_t hn_ = (uint32_t)__popcll(hm_), room_ = hend - hcur;                    \
                        uint32_t hp_ = hcur + cf_rank_in(hm_);                                                \
                        if (hn_ > room_) {      /* (wave-uniform) the rest goes to a new block of 64: entries the wave never fills stay 0xFFFF */ \
                            uint32_t nb_ = 0;                                                                 \
                            if (lane == 0) nb_ = atomicAdd(&sh[11], 64u);                                     \
                            nb_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb_);                         \
                            if (nb_ + (uint32_t)lane < DIST_HOT_CAP) hotl[nb_ + (uint32_t)lane] = (uint16_t)0xFFFFu; \
                            hp_ = hp_ < hend ? hp_ : nb_ + (hp_ - hend);                                      \
                            hcur = nb_ + (hn_ - room_); hend = nb_ + 64u;                                     \
                        } else hcur += hn_;                                                                   \
                        if ((SLOT) != 0xFFFFFFFFu && hp_ < DIST_HOT_CAP) hotl[hp_] = (uint16_t)(SLOT);        \
                    }                                                                                         \
                }
#elif defined(CF_DIST_DIAG_NOHOT)
#define CF_DIST_HOT(SLOT) { if ((SLOT) == 0xFFFFFFF0u) sh[11] = 1u; }      /* (diagnostic build: no hot list; the filter scans) */
#else
#define CF_DIST_HOT(SLOT) {                                                                                   \
                    const unsigned long long hm_ = cf_ballot((SLOT) != 0xFFFFFFFFu);                           \
                    if (hm_ && use_hot) {                                                                     \
                        uint32_t hb_ = 0;                                                                     \
                        if (lane == 0) hb_ = atomicAdd(&sh[11], (uint32_t)__popcll(hm_));                     \
                        hb_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb_) + cf_rank_in(hm_);          \
                        if ((SLOT) != 0xFFFFFFFFu && hb_ < DIST_HOT_CAP) hotl[hb_] = (uint16_t)(SLOT);        \
                    }                                                                                         \
                }
#endif
#define CF_DIST_OVERFLOW(N) {                                                                                 \
                    const uint32_t m_ = (N); otail -= m_; CF_DG(if (lane == 0) { ++dg[1]; dg[2] += m_; })      \
                    uint32_t omade_ = 0, ohot_ = 0xFFFFFFFFu;                                                 \
                    if ((uint32_t)lane < m_) {                                                                \
                        uint32_t xb, xd, xk;                                                                  \
                        T.q_take(ovq[otail + (uint32_t)lane], n_buckets, xb, xd, xk);                        \
                        omade_ = cf_dist_insert(T, n_buckets, xk, xb, xd, sh, A.min_cov, ohot_, CF_DG_PTR);   \
                    }                                                                                         \
                    const uint32_t onew_ = (uint32_t)__popcll(cf_ballot(omade_ != 0u));                        \
                    if (onew_ && lane == 0) atomicAdd(&sh[0], onew_);                                         \
                    CF_DIST_HOT(ohot_)                                                                        \
                }
#if CF_DIST_FILLRD
#define CF_DIST_FILL_ISSUE const uint32_t fl_ = sh[0];
#define CF_DIST_FILL_LOOK too_full = too_full || (uint32_t)__builtin_amdgcn_readfirstlane((int)fl_) > A.fill_limit;
#else
#define CF_DIST_FILL_ISSUE
#define CF_DIST_FILL_LOOK
#endif
#define CF_DIST_DRAIN(N) {                                                                                    \
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                    \
                    __builtin_amdgcn_wave_barrier();                                                          \
                    const uint32_t n_ = (N); qtail -= n_; CF_DG(if (lane == 0) { ++dg[3]; dg[4] += n_; })     \
                    bool made_ = false, park_ = false;                                                        \
                    uint32_t hot_ = 0xFFFFFFFFu;                                                              \
                    typename Tab::qitem it_ = 0;                                                              \
                    CF_DIST_FILL_ISSUE                                                                        \
                    if constexpr (Tab::kProbe1 && !CF_DIST_OLD_DRAIN && CF_DIST_DRAIN2) {                     \
                        /* two queued inserts per lane, their probes side by side (lanes past n_: stale words of the queue) */ \
                        bool made2_ = false, park2_ = false;                                                  \
                        uint32_t hot2_ = 0xFFFFFFFFu;                                                         \
                        uint32_t xb, xd, xk, yb, yd, yk;                                                      \
                        it_ = wq[qtail + (uint32_t)lane];                                                     \
                        const typename Tab::qitem it2_ = wq[qtail + 64u + (uint32_t)lane];                    \
                        T.q_take(it_, n_buckets, xb, xd, xk);                                                 \
                        T.q_take(it2_, n_buckets, yb, yd, yk);                                                \
                        T.probe2((uint32_t)lane < n_, xk, T.key_of(xb, xd), xb, (uint32_t)lane + 64u < n_, yk, T.key_of(yb, yd), yb, n_buckets, A.min_cov, made_, park_, hot_, made2_, park2_, hot2_); \
                        const uint32_t new_ = (uint32_t)__popcll(cf_ballot(made_)) + (uint32_t)__popcll(cf_ballot(made2_)); \
                        if (new_ && lane == 0) atomicAdd(&sh[0], new_);                                       \
                        const unsigned long long pm_ = cf_ballot(park_), pm2_ = cf_ballot(park2_);              \
                        if (pm_ | pm2_) {                                                                     \
                            if (park_) ovq[otail + cf_rank_in(pm_)] = it_;                                    \
                            otail += (uint32_t)__popcll(pm_);                                                 \
                            if (park2_) ovq[otail + cf_rank_in(pm2_)] = it2_;                                 \
                            otail += (uint32_t)__popcll(pm2_);                                                \
                        }                                                                                     \
                        CF_DIST_HOT(hot_)                                                                     \
                        CF_DIST_HOT(hot2_)                                                                    \
                    } else {                                                                                  \
                    if constexpr (Tab::kProbe1 && !CF_DIST_OLD_DRAIN) {                                       \
                        uint32_t xb, xd, xk;                                                                  \
                        it_ = wq[qtail + (uint32_t)lane];      /* (lanes past n_: a stale word of the queue) */ \
                        T.q_take(it_, n_buckets, xb, xd, xk);                                                 \
                        T.probe1((uint32_t)lane < n_, xk, T.key_of(xb, xd), A.min_cov, made_, park_, hot_);   \
                    } else if ((uint32_t)lane < n_) {                                                         \
                        uint32_t xb, xd, xk;                                                                  \
                        it_ = wq[qtail + (uint32_t)lane];                                                     \
                        T.q_take(it_, n_buckets, xb, xd, xk);                                                 \
                        const typename Tab::bucket k_ = T.read(xk);                                           \
                        const int mt_ = T.match(k_, xb, xd);                                                  \
                        if (mt_ >= 0) { if (T.add(xk, mt_) == A.min_cov) hot_ = Tab::kSlotsPerBucket * xk + (uint32_t)mt_; } \
                        else {                                                                                \
                            const int em_ = Tab::empty(k_);                                                   \
                            park_ = true;                                                                     \
                            if (em_ >= 0) {                                                                   \
                                uint32_t cn_;                                                                 \
                                const int st_ = T.claim_finish(T.claim_issue(xk, em_, xb, xd), xk, em_, xb, xd, cn_); \
                                made_ = st_ == 0; park_ = st_ == 2;                                           \
                                if (st_ != 2 && cn_ == A.min_cov) hot_ = Tab::kSlotsPerBucket * xk + (uint32_t)em_; \
                            }                                                                                 \
                        }                                                                                     \
                    }                                                                                         \
                    const uint32_t new_ = (uint32_t)__popcll(cf_ballot(made_));                                \
                    if (new_ && lane == 0) atomicAdd(&sh[0], new_);                                           \
                    const unsigned long long pm_ = cf_ballot(park_);                                           \
                    if (pm_) {                                                                                \
                        if (park_) ovq[otail + cf_rank_in(pm_)] = it_;                                        \
                        otail += (uint32_t)__popcll(pm_);                                                     \
                    }                                                                                         \
                    CF_DIST_HOT(hot_)                                                                         \
                    }                                                                                         \
                    CF_DIST_FILL_LOOK                                                                         \
                    __builtin_amdgcn_wave_barrier();                                                          \
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                    \
                }
                CF_DG(const unsigned long long t2_ = __builtin_amdgcn_s_memtime();)
                if (CF_DIST_ABL < 4) cf_dist_sweep<Tab, CF_DIST_PF_B>(A, recs, mine, my0, [&](bool final) {
                    if (CF_DIST_ABL >= 2) { qtail = 0; return; }
                    // a step pushes at most 4 x 64 inserts: the queue is brought below 64 first; after the wave's last step
                    // both lists are emptied (a pass whose table got too full is void and drops them)
                    constexpr uint32_t kDrain = (Tab::kProbe1 && !CF_DIST_OLD_DRAIN && CF_DIST_DRAIN2) ? 128u : 64u;      // queued inserts per drain
                    const uint32_t lim = final ? 1u : kDrain;
                    if (qtail < lim && !(final && otail > 0u)) return;      // nothing to drain: no look at the fill level either
                    for (;;) {
#if CF_DIST_FILLRD
                        if (too_full) break;      // (seen by the last drain; the limit leaves room for the drains of all waves that run meanwhile)
#else
                        if (sh[0] > A.fill_limit) { too_full = true; break; }
#endif
                        if (qtail >= lim) { CF_DG(const unsigned long long t1_ = __builtin_amdgcn_s_memtime();) CF_DIST_DRAIN(min(qtail, kDrain)) CF_DG(if (lane == 0) dg[6] += __builtin_amdgcn_s_memtime() - t1_;) }
                        else if (!(final && otail > 0u)) break;
#if defined(CF_DIST_DIAG_NOPARK)
                        otail = 0;      // (diagnostic build: parked inserts are dropped — timing only, results are wrong)
#endif
                        while (otail >= 32u || (final && qtail == 0u && otail > 0u)) { CF_DG(const unsigned long long t0_ = __builtin_amdgcn_s_memtime();) CF_DIST_OVERFLOW(min(otail, 64u)) CF_DG(if (lane == 0) dg[5] += __builtin_amdgcn_s_memtime() - t0_;) }
                    }
                }, [&](const uint32_t (&bb)[DIST_UNROLL], const uint32_t (&dd_)[DIST_UNROLL], const uint32_t (&qq_)[DIST_UNROLL], const uint32_t (&lo_)[DIST_UNROLL], uint32_t ok, uint32_t len) -> bool {
                    if (too_full) return true;     // (wave-uniform, set by the drains: the fill only changes there) the pass will be split
                    // Entries equal to a itself are NOT told apart here (round 2 spent 13 vector instructions per step on it): they are
                    // rare (a k-mer twice in one read), the filter never selects a slot whose b is a, and the emission count of the
                    // launch has them subtracted on the host (cf_self_pairs_kernel counts them from the posting lists).
                    uint32_t w_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) w_[u] = bm[Tab::bm_bit(lo_[u]) >> 5];
                    // Entries that do not exist (only the last item of a posting has lanes past its end) or belong to another partition
                    // (only a first k-mer whose table was split) get an empty bitmap word, in two wave-uniform branches that are nearly
                    // never taken: the common path tests one bit per entry and nothing else.
                    if (len < DIST_ITEM) {
#pragma unroll
                        for (int u = 0; u < DIST_UNROLL; ++u) if (!((ok >> u) & 1u)) w_[u] = 0u;
                    }
                    if (pmask) {
                        uint32_t live = len < DIST_ITEM ? ok : (1u << DIST_UNROLL) - 1u;
#pragma unroll
                        for (int u = 0; u < DIST_UNROLL; ++u) {
                            const uint32_t hb = Tab::hash(bb[u]);
                            if ((((hb ^ (hb >> 15)) >> 3) & pmask) != pidx) { live &= ~(1u << u); w_[u] = 0u; }
                        }
                        my_e += (uint32_t)__popcll((unsigned long long)live);
                    }
                    s_e += pmask ? 0u : len;      // entries swept, counted on the scalar unit (an `else` here made the compiler keep both
                                                  // counters in a scratch array picked by index: a scratch load + store per step)
                    bool c_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) c_[u] = cf_bit_of(w_[u], lo_[u]) != 0u;
                    if (CF_DIST_ABL >= 3) { if (c_[0] && c_[1] && c_[2] && c_[3] && bb[0] == 0xFFFFFFF1u) sh[13] = 1u; return false; }
#if CF_DIST_PUSH_FLAT
                    // no test for "no candidate in the wave" (with 15 % of the entries candidates there always is one: the test was a
                    // compare of the ballot and a branch per entry slot), rank and queue word made by every lane, only the store under the mask
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) {
                        const unsigned long long cm = cf_ballot(c_[u]);
                        const uint32_t at = qtail + cf_rank_in(cm);
                        const typename Tab::qitem qv = T.q_push(bb[u], dd_[u], qq_[u], n_buckets);
                        if (c_[u]) wq[at] = qv;
                        qtail += (uint32_t)__popcll(cm);
                    }
#else
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) {
                        const unsigned long long cm = cf_ballot(c_[u]);
                        if (cm) {
                            if (c_[u]) wq[qtail + cf_rank_in(cm)] = T.q_push(bb[u], dd_[u], qq_[u], n_buckets);
                            qtail += (uint32_t)__popcll(cm);
                        }
                    }
#endif
                    // (measured and not kept: every lane storing — candidates at their rank, the others into a dump word of their own —
                    // to save the four scalar instructions per entry of the skip branches and the exec save / restore: 325 vs 320 ms,
                    // the LDS takes four full-wave writes per step instead of 15 % of the lanes)
                    return false;
                });
#undef CF_DIST_DRAIN
#undef CF_DIST_OVERFLOW
#undef CF_DIST_HOT
#undef CF_DIST_FILL_ISSUE
#undef CF_DIST_FILL_LOOK
                CF_DG(if (lane == 0) dg[7] += __builtin_amdgcn_s_memtime() - t2_;)
                CF_STAMP(3);   // table sweep + inserts (wave 0's own items)
            }
            __syncthreads();      // [table swept]
            CF_STAMP(7);
            if (sh[0] > A.fill_limit) {  // overflow: split this partition in two, go on with one of the halves
                __syncthreads();      // (everyone has seen the fill level before thread 0 resets it)
                if (t == 0) {
                    uint32_t sp = sh[2];
                    if (P >= (1u << 20) || sp + 2 > DIST_STACK) { atomicOr(&A.counters[4], 1ull); }      // (the launch fails: -34)
                    else { stack[2 * sp] = 2 * P; stack[2 * sp + 1] = pidx; stack[2 * sp + 2] = 2 * P; stack[2 * sp + 3] = pidx + P; sh[2] = sp + 2; }
                    if (sh[2]) pop_pass(); else { publish_next(); sh[3] = 0; }      // (nothing left only after that error: on to the next first k-mer)
                }
                spilled = true;
                T.clear(slots, (uint32_t)t, (uint32_t)nt);
                __syncthreads();
                if (sh[3] == 0u) break;
                continue;
            }
            // ---- table of the pass complete: count emissions, filter in LDS (selected slots are marked and
            // staged), reserve the edge range with ONE global atomic, then write
            for (int d = 32; d >= 1; d >>= 1) my_e += __shfl_down(my_e, (unsigned)d);
            if (lane == 0 && (my_e + s_e)) atomicAdd(&sh[7], my_e + s_e);
            if (t == 0 && sh[2] == 0u) publish_next();      // the last pass: the next first k-mer's head rides on the filter's barrier
            // Filter in two steps.  (1) A group of slots per thread and round: the slots whose count reaches min_cov (few: the
            // table is sparse and most pairs stay below) are compacted into a list that lies over the insert queues, dead
            // by now.  (2) The list is evaluated one slot per thread with all lanes busy: sum over d from the bucket's
            // registers or a chain walk, the double division, mark + stage.  (Evaluating inside the bucket scan ran the
            // division code at 8 unrolled sites per round with 4 % of the lanes active: 9 100 cycles per first k-mer.)
            // The slots to evaluate: the list the insert path made (slots whose count reached min_cov) — or, when that list
            // overflowed / min_cov < 2 / a test asks for it, a scan of the whole table into a list over the insert queues.
            uint16_t* hot = hotl;
            uint32_t hot_cap = min(A.hot_cap, DIST_HOT_CAP);
            const bool from_inserts = use_hot && A.min_cov >= 2u && sh[11] <= hot_cap;      // (uniform: nothing changes sh[11] after the sweep's barrier)
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (!from_inserts) {
            if (use_hot) {      // (a first k-mer that kept no list — round 6 — finds the cursor as the pass's set-up left it, 0: the inserts only move it when they keep the list; two barriers less)
            __syncthreads();
            if (t == 0) sh[11] = 0;
            __syncthreads();
            }
            hot = (uint16_t*)wq0;
            hot_cap = min(A.hot_cap, (uint32_t)((size_t)(nt >> 6) * DIST_QCAP * sizeof(typename Tab::qitem) / 2));
            const uint32_t n_groups = slots / Tab::kScanGroup;
            for (uint32_t g0 = 0; g0 < n_groups; g0 += (uint32_t)nt) {      // (uniform trip count: ballots inside)
                const uint32_t bk = g0 + (uint32_t)t;                       // a group of kScanGroup consecutive slots
                const uint32_t m = bk < n_groups ? T.hot_mask(bk, A.min_cov) : 0u;
                const uint32_t cnt = (uint32_t)__popc(m);
                if (cf_ballot(cnt != 0u)) {
                    // positions in the list: lanes hand in their k-th hot slot in round k (most lanes have none, few have
                    // two: one or two rounds, against eight ballots — one per slot of the group — before)
                    uint32_t total = 0;
                    for (uint32_t k = 0; k < Tab::kScanGroup; ++k) { const unsigned long long bk_ = cf_ballot(cnt > k); if (!bk_) break; total += (uint32_t)__popcll(bk_); }
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&sh[11], total);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    uint32_t mm = m;
                    for (uint32_t k = 0; k < Tab::kScanGroup; ++k) {
                        const unsigned long long bk_ = cf_ballot(cnt > k);
                        if (!bk_) break;
                        if (cnt > k) {
                            const uint32_t bit = (uint32_t)__ffs((int)mm) - 1u;
                            mm &= mm - 1u;
                            const uint32_t pos = base + cf_rank_in(bk_);
                            if (pos < hot_cap) hot[pos] = (uint16_t)(Tab::kScanGroup * bk + Tab::slot_of_bit(bit));
                        }
                        base += (uint32_t)__popcll(bk_);
                    }
                }
            }
            __syncthreads();
            }
            const uint32_t n_hot = sh[11];
            // cnt / total >= thr as Python evaluates it (distance_based_kmer_recruitment.py:143-144: true division of two ints, compared
            // with a double).  For the default threshold — the literal 0.8, the double just above 4 / 5 — the test is exactly
            // 5 cnt >= 4 total: a quotient below 4 / 5 lies at least 1 / (5 total) below it, far more than an ulp for any total < 2^40,
            // and 4 / 5 itself rounds to the literal.  Other thresholds take the double division (about 40 instructions).
            auto dominant = [&](uint32_t cnt, unsigned long long total) -> bool {
                if (A.thr_den) return (unsigned long long)A.thr_den * cnt >= (unsigned long long)A.thr_num * total;
                return ((double)cnt / (double)total) >= A.thr;
            };
            // A selected edge takes its row of the output right here: the workgroup's chunk (DIST_EDGE_CHUNK rows reserved with
            // one global atomic; base in sh[28,29], rows in sh[27], cursor sh[26]) hands out rows through an LDS atomic per wave
            // and round, and the lane that evaluated the slot writes the row from its registers.  Only an edge whose row does not
            // fit the rest of the chunk is marked and staged (count sh[30]); those are written behind a barrier, after thread 0 has
            // reserved the next chunk.  (The first half of round 3 staged EVERY selected slot, met at a barrier for thread 0 to
            // hand out the rows and read the slots again: the write phase was 9 % of the kernel.)
            const unsigned long long ch_base = ((unsigned long long)sh[29] << 32) | sh[28];
            const uint32_t ch_rows = sh[27];
            auto put_row = [&](unsigned long long o, uint32_t b, uint32_t dd, uint32_t cnt) {
                if (o < A.edge_cap) *(cf_u32x4*)(A.edges + 4 * o) = cf_u32x4{dd, a, b, cnt};
                const uint32_t bit = 1u << (b & 31);
                if (!(A.unique_bits[b >> 5] & bit)) atomicOr(&A.unique_bits[b >> 5], bit);
            };
            // the same in two halves (round 6): the row is stored and the word of b's unique bit REQUESTED; the bit is looked at — and set
            // when it is not — a round of the filter later.  A first k-mer of cenX-shaped reads has ~3 500 selected edges: four rounds
            // of the list per thread, each of which waited for its own round trip to the unique mask (18 % of the kernel by thread 0's stamps).
            auto set_unique = [&](uint32_t w, uint32_t b) { const uint32_t bit = 1u << (b & 31); if (!(w & bit)) atomicOr(&A.unique_bits[b >> 5], bit); };
            // called by all lanes of a wave together; true for the lanes whose edge got its row here (the others': behind the barrier below)
            auto keep = [&](bool sel, uint32_t s, uint32_t b, uint32_t dd, uint32_t cnt) -> bool {
                const unsigned long long m = cf_ballot(sel);
                if (!m) return false;
                uint32_t row0 = 0;
                if (lane == 0) { const uint32_t n = (uint32_t)__popcll(m); atomicAdd(&sh[8], n); row0 = atomicAdd(&sh[26], n); }
                const uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)row0) + (uint32_t)__popcll(m & lt);
                const bool late = sel && row >= ch_rows;
                const unsigned long long o = ch_base + row;
                if (sel && !late && o < A.edge_cap) *(cf_u32x4*)(A.edges + 4 * o) = cf_u32x4{dd, a, b, cnt};
                const unsigned long long ml = cf_ballot(late);
                if (ml) {
                    uint32_t p0 = 0;
                    if (lane == 0) p0 = atomicAdd(&sh[30], (uint32_t)__popcll(ml));
                    const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)p0) + (uint32_t)__popcll(ml & lt);
                    if (late) { T.mark(s); if (pos < A.stage_cap) stage[pos] = (uint16_t)s; }
                }
                return sel && !late;
            };
            if (CF_DIST_ABL >= 1) { }
            else if (n_hot <= hot_cap) {
                bool p_on = false;      // the previous round's edge of this lane: its unique-mask word is on its way
                uint32_t p_w = 0, p_b = 0;
                for (uint32_t i0 = 0; i0 < n_hot; i0 += (uint32_t)nt) {
                    const uint32_t i = i0 + (uint32_t)t;
                    bool sel = false;
                    uint32_t s = 0, eb = 0, ed = 0, ec = 0;
                    if (i < n_hot && (s = hot[i]) != 0xFFFFu) {      // (0xFFFF: an entry of a wave's block of the list that the wave did not fill)
                        T.eval_slot(s, n_buckets, A.min_cov, [&](uint32_t, uint32_t b, uint32_t dd, uint32_t cnt, unsigned long long total) { sel = b != a && dominant(cnt, total); eb = b; ed = dd; ec = cnt; });
                    }
                    const bool wrote = keep(sel, s, eb, ed, ec);
                    uint32_t w_new = 0;
                    if (wrote) w_new = A.unique_bits[eb >> 5];
                    if (p_on) set_unique(p_w, p_b);
                    p_on = wrote; p_w = w_new; p_b = eb;
                }
                if (p_on) set_unique(p_w, p_b);
            } else {        // more than the list holds (never seen with the sketch): evaluate inside the bucket scan, a row per atomic
                for (uint32_t bk = (uint32_t)t; bk < n_buckets; bk += (uint32_t)nt)
                    T.for_counts_at_least(bk, n_buckets, A.min_cov, [&](uint32_t s, uint32_t b, uint32_t dd, uint32_t cnt, unsigned long long total) {
                        if (b != a && dominant(cnt, total)) {
                            atomicAdd(&sh[8], 1u);
                            const uint32_t row = atomicAdd(&sh[26], 1u);
                            if (row < ch_rows) put_row(ch_base + row, b, dd, cnt);
                            else {
                                T.mark(s);
                                const uint32_t pos = atomicAdd(&sh[30], 1u);
                                if (pos < A.stage_cap) stage[pos] = (uint16_t)s;
                            }
                        }
                    });
            }
            __syncthreads();      // [filtered]
            CF_STAMP(4);   // filter
            const uint32_t n_sel = sh[8], n_late = sh[30];
            const uint32_t sp_left = sh[2];      // partitions still on the stack (thread 0 changes the word only behind the next barrier)
            if (t == 0) {
                acc_E += sh[7]; ++acc_pass;
                if (n_sel) { acc_edges += n_sel; atomicOr(&A.unique_bits[a >> 5], 1u << (a & 31)); }      // (fire and forget: a load to test the bit first would stall thread 0)
            }
            if (n_late) {      // (uniform; once per DIST_EDGE_CHUNK rows) the chunk is full: the late rows open the next one
                __syncthreads();      // everyone has read the chunk and the counts
                if (t == 0) {
                    const unsigned long long take = ((unsigned long long)n_late + A.edge_chunk - 1ull) / A.edge_chunk * A.edge_chunk;
                    const unsigned long long nb = atomicAdd(&A.counters[0], take);
                    sh[28] = (uint32_t)nb; sh[29] = (uint32_t)(nb >> 32); sh[27] = (uint32_t)take; sh[26] = n_late;
                    sh[30] = 0; sh[13] = 0;      // ([13]: cursor of the marked-slot sweep below; the sketch's flag is long read)
                }
                __syncthreads();      // [rows reserved]
                const unsigned long long nbase = ((unsigned long long)sh[29] << 32) | sh[28];
                if (n_late <= A.stage_cap) {           // the staged slot list
                    for (uint32_t i = (uint32_t)t; i < n_late; i += (uint32_t)nt) {
                        uint32_t b, dd, cnt;
                        if (T.get(stage[i], b, dd, cnt)) put_row(nbase + i, b, dd, cnt);
                    }
                } else {                               // more late edges than the stage holds: sweep the marked slots, a bucket per thread
                    for (uint32_t bk = (uint32_t)t; bk < n_buckets; bk += (uint32_t)nt)
                        T.for_marked(bk, [&](uint32_t, uint32_t b, uint32_t dd, uint32_t cnt) { put_row(nbase + atomicAdd(&sh[13], 1u), b, dd, cnt); });
                }
            }
            CF_STAMP(5);   // reserve + write edges
            if (sp_left == 0u) break;
            __syncthreads();      // (more partitions) the rows are out of the table before it is cleared for the next one
            if (t == 0) pop_pass();
            T.clear(slots, (uint32_t)t, (uint32_t)nt);
            __syncthreads();
        }
        if (spilled && t == 0) ++acc_spill;
    }
    if (t == 0) {
        if (acc_E) atomicAdd(&A.counters[1], acc_E);
        if (acc_spill) atomicAdd(&A.counters[2], acc_spill);
        if (acc_pass) atomicAdd(&A.counters[5], acc_pass);
        if (acc_edges) atomicAdd(&A.counters[6], acc_edges);
        if (sh[27] > sh[26]) {      // the unused rest of the last chunk: a hole for cf_edge_compact_kernel
            const unsigned long long h = atomicAdd(&A.counters[7], 1ull), cb = ((unsigned long long)sh[29] << 32) | sh[28];
            A.holes[2 * h] = cb + sh[26]; A.holes[2 * h + 1] = (unsigned long long)(sh[27] - sh[26]);
        }
#if defined(CF_DIST_STAMPS)
        for (int i = 0; i < 8; ++i) atomicAdd(&A.counters[8 + i], stamp_acc[i]);
#endif
    }
#if defined(CF_DIST_DIAG_COUNT)
    for (int i = 0; i < 8; ++i) if (dg[i]) atomicAdd(&A.counters[8 + i], dg[i]);
#endif
}

__global__ void __launch_bounds__(256)
cf_max_u32_kernel(const uint32_t* __restrict__ v, int64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = max(m, v[i]);
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_down(m, (unsigned)d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// Closes the holes of the chunked edge output: row k of the move list goes from the k-th valid row at or above n_valid to the k-th
// hole row below it (segment lists with inclusive prefixes, built on the host from the <= grid holes of the kernel).
__global__ void __launch_bounds__(256)
cf_edge_compact_kernel(uint32_t* __restrict__ edges, const unsigned long long* __restrict__ src_start, const unsigned long long* __restrict__ src_pre, int n_src,
                       const unsigned long long* __restrict__ dst_start, const unsigned long long* __restrict__ dst_pre, int n_dst, unsigned long long n_move) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_move; k += stride) {
        int lo = 0, hi = n_src - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (src_pre[mid] > k) hi = mid; else lo = mid + 1; }
        const unsigned long long from = src_start[lo] + (k - (lo ? src_pre[lo - 1] : 0ull));
        lo = 0; hi = n_dst - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (dst_pre[mid] > k) hi = mid; else lo = mid + 1; }
        const unsigned long long to = dst_start[lo] + (k - (lo ? dst_pre[lo - 1] : 0ull));
        *(cf_u32x4*)(edges + 4 * to) = *(const cf_u32x4*)(edges + 4 * from);
    }
}

// order-independent checksum of stored edges: sum over rows of mix(d, a, b, cnt) mod 2^64 (the oracle's edge checksum:
// tests compare all 3e9 edges of a full-size run without copying 50 GB to the host)
__global__ void __launch_bounds__(256)
cf_edge_checksum_kernel(const uint32_t* __restrict__ edges, int64_t n, unsigned long long* __restrict__ out) {
    unsigned long long s = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const cf_u32x4 e = *(const cf_u32x4*)(edges + 4 * i);
        s += cf_mix64(cf_mix64(cf_mix64(cf_mix64((unsigned long long)e.x + 0x9E37ull) ^ (unsigned long long)e.y) ^ ((unsigned long long)e.z << 1)) ^ ((unsigned long long)e.w << 2));
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, (unsigned)d);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

extern "C" {

int cf_edges_checksum(cf_ctx* ctx, int64_t n, uint64_t* out) {
    if (!ctx || !out) return -22;
    *out = 0;
    n = std::min(n, ctx->n_edges_stored);
    if (n <= 0) return 0;
    CF_HIP(hipSetDevice(ctx->device));
    unsigned long long* d_sum = nullptr;
    CF_TRY(cf_alloc_t(ctx, &d_sum, 1, "edge checksum"));
    int rc = 0;
    unsigned long long h = 0;
    if (hipMemsetAsync(d_sum, 0, 8, ctx->stream) != hipSuccess) rc = cf_fail(ctx, -5, "edge checksum memset");
    if (!rc) {
        hipLaunchKernelGGL(cf_edge_checksum_kernel, dim3((unsigned)cf_grid_for(n, 256, std::max(1, ctx->n_cu) * 16)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)ctx->d_edges, n, d_sum);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cf_fail(ctx, -5, "edge checksum kernel");
    }
    cf_release_t(ctx, d_sum, 1);
    if (!rc) *out = (uint64_t)h;
    return rc;
}

int cf_dist_edges(cf_ctx* ctx, int64_t min_n, int64_t max_n, int32_t min_d, int32_t max_d, uint32_t min_cov,
                  double rel_threshold, int32_t part, int32_t n_parts, int64_t edge_cap, int64_t* n_edges) {
    if (!ctx) return -22;
    if (!ctx->have_clouds && !ctx->have_gview) return cf_fail(ctx, -22, "cf_dist_edges: no clouds built");
    // the clouds the stage works on: the all-gathered ones of every rank (multi-GPU) or the local ones
    const bool gv = ctx->have_gview;
    const int64_t* v_unit_ptr = gv ? ctx->g_unit_ptr : ctx->d_unit_ptr;
    const int64_t* v_cloud_ptr = gv ? ctx->g_cloud_ptr : ctx->d_cloud_ptr;
    const int32_t* v_entries = gv ? ctx->g_entries_d : ctx->d_entries;
    const std::vector<int64_t>& v_h_unit_ptr = gv ? ctx->g_h_unit_ptr : ctx->h_unit_ptr;
    const int64_t v_n_entries = gv ? ctx->g_entries : ctx->n_entries;
    if (n_parts < 1 || part < 0 || part >= n_parts) return cf_fail(ctx, -22, "cf_dist_edges: bad partition");
    if (max_d > 65535) return cf_fail(ctx, -22, "cf_dist_edges: max_d > 65535 does not fit the 16-bit distance field");
    if (v_n_entries >= ((int64_t)1 << 32) - 8 * (int64_t)DIST_ITEM) return cf_fail(ctx, -34, "cf_dist_edges: more than 2^32 cloud entries (the sweeps address them with 32-bit indices)");
    if (edge_cap < 0) edge_cap = 0;
    const int64_t R = gv ? ctx->g_reads : ctx->n_reads, U = gv ? ctx->g_units : ctx->n_units, K = ctx->n_kmers;
    // (no limit on the units of a read: unit indices travel mod 256 / mod 65536 and a difference <= max_d is exact)
    CF_HIP(hipSetDevice(ctx->device));
    CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    // Python slice semantics of itertools.islice(items, min_n, max_n) for non-negative bounds
    if (min_n < 0) min_n = 0;
    if (max_n > R) max_n = R;
    if (max_n < min_n) max_n = min_n;
    if (min_n > R) min_n = R;
    const int64_t u0 = v_h_unit_ptr[(size_t)min_n], u1 = v_h_unit_ptr[(size_t)max_n];
    const int32_t min_d_eff = min_d < 1 ? 1 : min_d;  // kmer_clouds[:-0] is empty: d = 0 emits nothing

    uint32_t *d_pcnt = nullptr, *d_cursor = nullptr, *d_first = nullptr;
    unsigned long long *d_okeys = nullptr, *d_otmp = nullptr;
    int32_t* d_order = nullptr;
    int64_t n_order = 0, n_a_alloc = 0, n_item_slots = 0;
    uint32_t *d_icnt = nullptr, *d_ialloc = nullptr;
    int64_t* d_ibase = nullptr;
    cf_dist_head* d_heads = nullptr;
    cf_dist_item* d_items = nullptr;
    unsigned long long* d_holes = nullptr;
    size_t n_holes_alloc = 0;
    int64_t n_stored = 0;
    int64_t* d_post_ptr = nullptr;
    int32_t *d_post = nullptr, *d_rend = nullptr, *d_rbeg = nullptr;
    uint16_t* d_entry_i = nullptr;
    uint32_t* d_packed = nullptr;
    cf_dist_rec* d_urange = nullptr;
    bool narrow = false, wide16 = false, region = false, region26 = false;
    int narrow_db = 8, reg_shift = 0;
    uint8_t* d_entry_i8 = nullptr;
    unsigned long long* d_cnt = nullptr;
    int64_t n_post = 0;
    int rc = 0;
    unsigned long long h_cnt[8] = {0};
    const size_t n_cnt = 8 + 16 * 9;   // counters + 8 queue heads on their own cache lines
    const int max_blocks = std::max(1, ctx->n_cu) * 8;
    do {
        cf_free_edges(ctx);      // (the buffer of this call is allocated below, once the launch shape is known)
        if ((rc = cf_alloc_t(ctx, &d_pcnt, (size_t)K + 1, "posting counts"))) break;
        if ((rc = cf_alloc_t(ctx, &d_cursor, (size_t)K + 1, "posting cursors"))) break;
        if ((rc = cf_alloc_t(ctx, &d_post_ptr, (size_t)K + 1, "posting offsets"))) break;
        if ((rc = cf_alloc_t(ctx, &d_rend, (size_t)U + 1, "unit read ends"))) break;
        if ((rc = cf_alloc_t(ctx, &d_rbeg, (size_t)U + 1, "unit read begins"))) break;
        if ((rc = cf_alloc_t(ctx, &d_urange, (size_t)U + 1, "unit partner ranges"))) break;
        if ((rc = cf_alloc_t(ctx, &d_cnt, n_cnt, "dist counters"))) break;
        if ((rc = cf_alloc_t(ctx, &d_first, (size_t)K + 1, "first posting units"))) break;
        hipError_t e = hipMemsetAsync(d_pcnt, 0, (size_t)(K + 1) * 4, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_cursor, 0, (size_t)(K + 1) * 4, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, n_cnt * 8, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_first, 0xFF, (size_t)(K + 1) * 4, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, "cf_dist_edges memset"); break; }
        int64_t e0 = 0, e1 = 0;
        {
            int64_t tmp[2] = {0, 0};
            if (hipMemcpy(&tmp[0], v_cloud_ptr + u0, 8, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(&tmp[1], v_cloud_ptr + u1, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "cloud_ptr read"); break; }
            e0 = tmp[0]; e1 = tmp[1];
        }
        const bool by_sort = !ctx->dist_post_atomics && e1 > e0 && (e1 - e0) < ((int64_t)1 << 32) && U < ((int64_t)1 << 31);
        if (by_sort) {
            int64_t n = e1 - e0;
            const int64_t K_part = (K + n_parts - 1) / n_parts;      // sort keys: the rank, or rank / n_parts for one partition
            int kb = 1; while (kb < 32 && ((int64_t)1 << kb) < std::max<int64_t>(K_part, 2)) ++kb;
            kb = (kb + 7) & ~7;      // whole 8-bit digits: the last radix pass must not reach into the unit bits
            unsigned long long *d_recs = nullptr, *d_rtmp = nullptr, *d_sorted = nullptr;
            uint32_t* d_ucnt = nullptr; int64_t* d_uoff = nullptr;
            const unsigned g_units = (unsigned)cf_grid_for((u1 - u0) * 64, 256, max_blocks);
            if (n_parts > 1) {       // what this partition keeps of every unit's cloud, and where its records go
                if ((rc = cf_alloc_t(ctx, &d_ucnt, (size_t)(u1 - u0) + 1, "kept entries per unit"))) break;
                if ((rc = cf_alloc_t(ctx, &d_uoff, (size_t)(u1 - u0) + 1, "record offsets per unit"))) { cf_release_t(ctx, d_ucnt, (size_t)(u1 - u0) + 1); break; }
                if (hipMemsetAsync(d_ucnt + (u1 - u0), 0, 4, ctx->stream) != hipSuccess) rc = cf_fail(ctx, -5, "cf_dist_edges memset");
                hipLaunchKernelGGL(cf_post_ucount_kernel, dim3(g_units), dim3(256), 0, ctx->stream, v_cloud_ptr, v_entries, u0, u1, (uint32_t)part, (uint32_t)n_parts, d_ucnt);
                if (!rc) rc = cf_scan_exclusive_u32_to_i64(ctx, d_ucnt, d_uoff, (u1 - u0) + 1, &n);
            }
            if (!rc && (rc = cf_alloc_t(ctx, &d_recs, (size_t)n + 1, "posting records")) == 0 &&
                (rc = cf_alloc_t(ctx, &d_rtmp, (size_t)n + 1, "posting records (sort)")) != 0) { cf_release_t(ctx, d_recs, (size_t)n + 1); d_recs = nullptr; }
            if (!rc) {
                if (n_parts > 1)
                    hipLaunchKernelGGL(cf_post_recs_part_kernel, dim3(g_units), dim3(256), 0, ctx->stream,
                                       v_cloud_ptr, v_entries, u0, u1, (uint32_t)part, (uint32_t)n_parts, (const int64_t*)d_uoff, kb, d_recs);
                else
                    hipLaunchKernelGGL(cf_post_recs_kernel, dim3(g_units), dim3(256), 0, ctx->stream, v_cloud_ptr, v_entries, u0, u1, e0, kb, d_recs);
                if (n) rc = cf_radix_sort_u64_any(ctx, d_recs, d_rtmp, n, kb, &d_sorted);
                if (!rc) rc = cf_alloc_t(ctx, &d_post, (size_t)n, "postings");
                if (!rc) {
                    n_post = n;
                    if (n) hipLaunchKernelGGL(cf_post_bounds_kernel, dim3((unsigned)cf_grid_for(n, 256, max_blocks)), dim3(256), 0, ctx->stream,
                                       (const unsigned long long*)d_sorted, n, kb, (uint32_t)part, (uint32_t)n_parts, d_post, d_cursor, d_pcnt, d_first);      // (d_cursor = run starts, d_pcnt = run ends)
                    hipLaunchKernelGGL(cf_post_counts_kernel, dim3((unsigned)cf_grid_for(K, 256, max_blocks)), dim3(256), 0, ctx->stream, (const uint32_t*)d_cursor, d_pcnt, K);
                    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = cf_fail(ctx, -5, "postings by sort");
                }
                cf_release_t(ctx, d_rtmp, (size_t)n + 1);
                cf_release_t(ctx, d_recs, (size_t)n + 1);
            }
            if (d_uoff) cf_release_t(ctx, d_uoff, (size_t)(u1 - u0) + 1);
            if (d_ucnt) cf_release_t(ctx, d_ucnt, (size_t)(u1 - u0) + 1);
            if (rc) break;
            int64_t n_chk = 0;
            if ((rc = cf_scan_exclusive_u32_to_i64(ctx, d_pcnt, d_post_ptr, K + 1, &n_chk))) break;
            if (n_chk != n_post) { rc = cf_fail(ctx, -5, "cf_dist_edges: internal error, posting counts do not add up"); break; }
        } else {
        if (e1 > e0)
            hipLaunchKernelGGL(cf_post_hist_kernel, dim3((unsigned)cf_grid_for(e1 - e0, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               v_entries, e0, e1, (uint32_t)part, (uint32_t)n_parts, d_pcnt);
        if ((rc = cf_scan_exclusive_u32_to_i64(ctx, d_pcnt, d_post_ptr, K + 1, &n_post))) break;
        if ((rc = cf_alloc_t(ctx, &d_post, (size_t)n_post, "postings"))) break;
        if (u1 > u0 && n_post)
            hipLaunchKernelGGL(cf_post_fill_kernel, dim3((unsigned)cf_grid_for((u1 - u0) * 64, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               v_cloud_ptr, v_entries, u0, u1, (uint32_t)part, (uint32_t)n_parts, (const int64_t*)d_post_ptr, d_cursor, d_post, d_first);
        }
        if (R)
            hipLaunchKernelGGL(cf_unit_rend_kernel, dim3((unsigned)cf_grid_for(R, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               v_unit_ptr, v_cloud_ptr, R, min_d_eff, max_d, d_rend, d_rbeg, d_urange);
        // table layout: 6-byte slots (32-bit keys, 16-bit counts) whenever ranks fit 24 bits and counts 15 bits
        uint32_t max_post = 0;
        if (K) {
            hipLaunchKernelGGL(cf_max_u32_kernel, dim3((unsigned)cf_grid_for(K, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const uint32_t*)d_pcnt, K, (uint32_t*)(d_cnt + 7));
            if (hipMemcpy(&max_post, d_cnt + 7, 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "max postings"); break; }
        }
        // 6-byte slots [d : DB | b : 32 - DB] need the k-mer ranks in 32 - DB bits and every distance that can occur in DB
        // bits: d <= min(max_d, units of the longest read - 1)
        int64_t u_max = 0;
        for (int64_t r = min_n; r < max_n; ++r) u_max = std::max(u_max, v_h_unit_ptr[(size_t)r + 1] - v_h_unit_ptr[(size_t)r]);
        const int64_t d_need = std::max<int64_t>(1, std::min<int64_t>(max_d, u_max - 1));
        int need_bits = 1; while (((int64_t)1 << need_bits) <= d_need) ++need_bits;
        wide16 = d_need > 255;         // 16-bit distance field: [b:32 | d:16 | sel:1 | cnt:15]
        int kb = 24; while (kb < 32 && K >= ((int64_t)1 << kb) - 1) ++kb;       // ranks 0 .. K - 1 and the all-ones key stays free
        narrow_db = 32 - kb;                                                       // the widest distance field the ranks leave
        if (ctx->dist_dbits) narrow_db = std::min(narrow_db, ctx->dist_dbits);
        // (the 4-byte stream is read with buffer loads whose scalar byte offset has 32 bits: 2^30 entries and more take the other layouts)
        const bool stream4_ok = v_n_entries + (int64_t)DIST_ITEM < ((int64_t)1 << 30);
        narrow = !wide16 && !ctx->dist_wide && narrow_db >= 5 && narrow_db >= need_bits && max_post <= 32767u && stream4_ok;
        if (ctx->dist_dbits && !narrow) { rc = cf_fail(ctx, -22, "cf_dist_edges: dist_dbits does not fit this input (distances or k-mer ranks need more bits)"); break; }
        // ranks that do not fit next to the distance: the 6-byte slots in 2^S table regions (the key drops the rank's low S bits)
        if (!narrow || ctx->dist_regions) {
            reg_shift = 0;
            while (reg_shift < 3 && ((K - 1) >> reg_shift) > ((int64_t)1 << 24) - 2) ++reg_shift;
            if (ctx->dist_regions) { int want = 0; while ((1 << want) < ctx->dist_regions) ++want; reg_shift = std::max(reg_shift, want); }
            region = !wide16 && !ctx->dist_wide && !ctx->dist_dbits && reg_shift <= 3 && ((K - 1) >> reg_shift) <= ((int64_t)1 << 24) - 2 && max_post <= 32767u && K < ((int64_t)1 << 31);
            if (ctx->dist_regions && !region) { rc = cf_fail(ctx, -22, "cf_dist_edges: dist_regions does not fit this input"); break; }
            if (region) narrow = false;
            // up to 2^26 ranks and distances up to 127: the region table behind the 4-byte stream [unit index mod 64 : 6 | rank : 26]
            region26 = region && !ctx->dist_region_bytes && K <= ((int64_t)1 << 26) && d_need <= 127 && stream4_ok;
        }
        if (max_post >= (1u << 23) || (wide16 && max_post > 32767u)) { rc = cf_fail(ctx, -34, wide16 ? "cf_dist_edges: distances above 255 with a k-mer of more than 32767 postings" : "cf_dist_edges: a k-mer has more than 2^23 postings"); break; }
        if (narrow || region26) { if ((rc = cf_alloc_t(ctx, &d_packed, (size_t)v_n_entries + DIST_ITEM, "packed cloud entries"))) break; }
        else if (region) { if ((rc = cf_alloc_t(ctx, &d_entry_i8, (size_t)v_n_entries + 4 * DIST_ITEM, "entry unit indices (bytes)"))) break; }
        else if ((rc = cf_alloc_t(ctx, &d_entry_i, (size_t)v_n_entries + 1, "entry unit indices"))) break;
        if (U && v_n_entries)
            hipLaunchKernelGGL(cf_entry_unit_kernel, dim3((unsigned)cf_grid_for(U * 64, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               v_cloud_ptr, (const int32_t*)d_rbeg, v_entries, U, d_entry_i, d_packed, region26 ? 26 : 32 - narrow_db, d_entry_i8);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ctx->ev2, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("postings: ") + hipGetErrorString(e)); break; }

        cf_dist_args A;
        A.post_ptr = d_post_ptr; A.post = d_post; A.cloud_ptr = v_cloud_ptr; A.entries = v_entries; A.unit_rend = d_rend; A.unit_rbeg = d_rbeg; A.urange = d_urange; A.entry_i = d_entry_i; A.packed = d_packed; A.entry_i8 = d_entry_i8; A.reg_shift = (uint32_t)reg_shift;
        A.n_kmers = K; A.part = part; A.n_parts = n_parts; A.min_d = min_d_eff; A.max_d = max_d; A.min_cov = min_cov; A.thr = rel_threshold;
        A.thr_num = (rel_threshold == 0.8 && ctx->dist_int_thr) ? 4u : 0u; A.thr_den = A.thr_num ? 5u : 0u;
        A.stage_cap = (uint32_t)std::min(ctx->dist_stage, DIST_STAGE_CAP);
        A.hot_cap = ctx->dist_hot_cap > 0 ? (uint32_t)ctx->dist_hot_cap : 0xFFFFFFFFu;
        A.hot_entries = ctx->dist_hot_entries >= 0 ? (uint32_t)ctx->dist_hot_entries : 0xFFFFFFFFu;
        const uint32_t slot_bytes = (narrow || region) ? cf_tab_narrow::kSlotBytes : cf_tab_wide::kSlotBytes;
        // launch shape: two 512-thread workgroups per CU (80 KiB of LDS each) overlap each other's latency-bound phases
        // and win when a first k-mer has few pair emissions; with many (long reads, high coverage) the halved table and
        // sketch cost more than the overlap gains, and one 1024-thread workgroup with the whole LDS wins (measured:
        // 20 k emissions per first k-mer: 480 vs 716 ms; 53 k: 437 vs 274 ms; 160 k: 2097 vs 739 ms)
        // pair emissions per first k-mer, ESTIMATED on the host (round 6; rounds 2-5 summed the partner ranges of every posting on the device —
        // a kernel of random gathers, 1.05 ms of every launch, and a blocking copy — for a figure that only picks the launch shape): the
        // partner units of every unit of the reads in range, exactly, times the mean cloud size, times the mean postings per unit
        double per_first = 0.0;
        if (n_post && u1 > u0) {
            unsigned long long partner_units = 0;
            for (int64_t r = min_n; r < max_n; ++r) {
                const int64_t nu = v_h_unit_ptr[(size_t)r + 1] - v_h_unit_ptr[(size_t)r];
                // unit i of a read of nu units has max(0, min(nu - 1, i + max_d) - (i + min_d) + 1) partner units
                const int64_t span = (int64_t)max_d - min_d_eff + 1;
                if (nu <= min_d_eff || span <= 0) continue;
                const int64_t full = std::max<int64_t>(0, nu - max_d);           // units with the whole span behind them
                const int64_t tail = nu - min_d_eff - full;                      // the rest: nu - min_d - i partner units each, down to 1
                partner_units += (unsigned long long)(full * span) + (unsigned long long)(tail > 0 ? tail * (tail + 1) / 2 : 0);
            }
            const double cloud = (double)(e1 - e0) / (double)(u1 - u0), post_per_unit = (double)n_post / (double)(u1 - u0);
            per_first = (double)partner_units * cloud * post_per_unit / (double)std::max<int64_t>(1, (K - part + n_parts - 1) / n_parts);
        }
        int wgs = ctx->dist_wgs, block = ctx->dist_block;
        if (wgs == 0) wgs = (block > 512 || per_first > 32768.0) ? 1 : 2;
        if (block == 0) block = wgs == 1 ? 1024 : 512;
        // LDS: everything but the table is fixed; dist_slots (the table budget in 8-byte units) defaults to all the rest
        const size_t qitem_bytes = narrow ? sizeof(cf_tab_narrow::qitem) : sizeof(cf_tab_wide::qitem);      // (the region layouts queue 64-bit items, like the wide one)
        const size_t lds_fixed = DIST_LDS_HEAD + (size_t)(2 * DIST_STACK) * 4 + DIST_STAGE_CAP * 2 + 16 + (size_t)(block / 64) * (DIST_QSTRIDE + DIST_OVQ) * qitem_bytes + DIST_HOT_CAP * 2;
        const int64_t budget8 = ((int64_t)160 * 1024 / wgs - (int64_t)lds_fixed) / 8;
        if (budget8 < 256) { rc = cf_fail(ctx, -22, "cf_dist_edges: dist_wgs leaves no LDS for the table"); break; }
        if (ctx->dist_slots > budget8) { rc = cf_fail(ctx, -22, "cf_dist_edges: dist_slots does not fit the 160 KiB LDS next to the work lists"); break; }
        const int64_t slots8 = ctx->dist_slots ? ctx->dist_slots : budget8;
        A.slots = (int32_t)((slots8 * 8 / slot_bytes) & ~(region ? (int64_t)(32 << reg_shift) - 1 : 7ll));      // (regions: equal parts of whole 8-slot groups)
        if (A.slots >= 0xFFFF) { rc = cf_fail(ctx, -22, "cf_dist_edges: more than 65534 table slots (16-bit slot indices; 0xFFFF marks an unused entry of the hot list)"); break; }
        A.fill_limit = (uint32_t)((int64_t)A.slots * ctx->dist_fill_pct / 100);   // checked once per wave step: leave slack below the physical size
        {   //  the fill level a drain looks at lags by one drain: every wave may put 2 x 128 fresh keys behind the limit before it
            // stops — the limit leaves that room below the physical size where the table is large enough for it (tables that tests force down
            // to a few hundred slots keep their limit: a pass that fills up there is void and split, as it always was, and all the pairs of ONE
            // b must fit one partition)
            const int64_t slack = (int64_t)(block / 64) * 2 * 128;
            if (slack * 4 <= (int64_t)A.slots) A.fill_limit = (uint32_t)std::min<int64_t>(A.fill_limit, (int64_t)A.slots - slack);
        }
        A.est_limit = (uint32_t)((int64_t)A.fill_limit * 100 / ctx->dist_est_pct);
        A.counters = d_cnt; A.unique_bits = ctx->d_unique_bits;
        const size_t lds = (size_t)A.slots * slot_bytes + lds_fixed;
        A.sketch = (ctx->dist_sketch && min_cov >= 2 && min_cov <= 200) ? 1 : 0;
        A.sk_shift = 32; A.sk_counters = 1;
        const size_t sk_room = (size_t)A.slots * slot_bytes + DIST_STAGE_CAP * 2 + 16 + (size_t)(block / 64) * (DIST_QSTRIDE + DIST_OVQ) * qitem_bytes + 2 * DIST_STACK * 4 + DIST_HOT_CAP * 2;   // table + stage + stack + queues + hot list: all dead while the sketch runs
        while (A.sk_shift > 8 && (size_t)A.sk_counters * 2 <= sk_room) { A.sk_counters *= 2; --A.sk_shift; }
        if (A.sk_counters < 16) A.sketch = 0;
        A.sk_bytes = A.sk_counters;
        // 4-bit counters when min_cov - 1 fits well below their guard ("dist_sketch_bits" 8 keeps the bytes)
        const bool sk4 = A.sketch && min_cov <= 9 && ctx->dist_sketch_bits != 8 && A.sk_shift > 9;
        if (sk4) { A.sk_counters *= 2; --A.sk_shift; A.sk_fbits = 4; A.sk_fsh = 2; A.sk_wshift = 3; A.sk_fmask = 15; A.sk_guard = 12; }
        else { A.sk_fbits = 8; A.sk_fsh = 3; A.sk_wshift = 2; A.sk_fmask = 255; A.sk_guard = 0x7FFFFFFFu; }
        A.sk_mask = A.sk_counters - 1u;
        const int per_cu = std::max(1, std::min((int)((160 * 1024) / lds), 2048 / block));
        // locality order of the first k-mers: sort (first posting unit, a); k-mers without postings drop out
        n_a_alloc = (K > part) ? (K - part + n_parts - 1) / n_parts : 0;
        if ((rc = cf_alloc_t(ctx, &d_okeys, (size_t)n_a_alloc, "order keys"))) break;
        if ((rc = cf_alloc_t(ctx, &d_otmp, (size_t)n_a_alloc, "order scratch"))) break;
        if ((rc = cf_alloc_t(ctx, &d_order, (size_t)n_a_alloc, "order"))) break;
        int ab = 1, ub = 1;      // bits of a rank, bits of a unit number
        while (ab < 32 && ((int64_t)1 << ab) < K) ++ab;
        while (ub < 32 && ((int64_t)1 << ub) < std::max<int64_t>(U, 2)) ++ub;
        if (K) {
            hipLaunchKernelGGL(cf_order_keys_kernel, dim3((unsigned)cf_grid_for(K, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const uint32_t*)d_pcnt, (const uint32_t*)d_first, K, (int)part, (int)n_parts, ab, d_okeys, d_cnt + 6);
            unsigned long long h_n = 0;
            if (hipMemcpy(&h_n, d_cnt + 6, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "order count"); break; }
            n_order = (int64_t)h_n;
            if ((rc = cf_radix_sort_u64(ctx, d_okeys, d_otmp, n_order, ab + ub))) break;      // (round 6: the key's own bits — 6 radix passes at 50 000 reads; rounds 2-5 sorted all 64)
            if (n_order)
                hipLaunchKernelGGL(cf_order_extract_kernel, dim3((unsigned)cf_grid_for(n_order, 256, max_blocks)), dim3(256), 0, ctx->stream,
                                   (const unsigned long long*)d_okeys, n_order, ab, d_order);
        }
        A.order = d_order; A.n_order = n_order;
        // the work lists of the sweeps (heads + item records, laid out for workgroups of `block` threads): count, scan, fill
        if (n_order) {
            const uint32_t nw = (uint32_t)block / 64u;
            const int g_items = cf_grid_for(n_order * 16, 256, max_blocks);
            if ((rc = cf_alloc_t(ctx, &d_icnt, (size_t)n_order + 1, "item counts"))) break;
            if ((rc = cf_alloc_t(ctx, &d_ialloc, (size_t)n_order + 1, "item slots"))) break;
            if ((rc = cf_alloc_t(ctx, &d_ibase, (size_t)n_order + 1, "item bases"))) break;
            if ((rc = cf_alloc_t(ctx, &d_heads, (size_t)n_order, "first k-mer heads"))) break;
            hipLaunchKernelGGL(cf_items_count_kernel, dim3((unsigned)g_items), dim3(256), 0, ctx->stream, (const int32_t*)d_order, n_order, (const int64_t*)d_post_ptr,
                               (const int32_t*)d_post, (const cf_dist_rec*)d_urange, (const int32_t*)d_rbeg, min_d_eff, max_d, nw, by_sort ? 1 : 0, d_icnt, d_ialloc, d_cnt + 3);
            if ((rc = cf_scan_exclusive_u32_to_i64(ctx, d_ialloc, d_ibase, n_order, &n_item_slots))) break;
            if ((rc = cf_alloc_t(ctx, &d_items, (size_t)n_item_slots + 64, "item records"))) break;
            hipLaunchKernelGGL(cf_items_fill_kernel, dim3((unsigned)g_items), dim3(256), 0, ctx->stream, (const int32_t*)d_order, n_order, (const int64_t*)d_post_ptr,
                               (const int32_t*)d_post, (const cf_dist_rec*)d_urange, nw, (const uint32_t*)d_icnt, (const int64_t*)d_ibase, d_heads, d_items,
                               region26 ? v_cloud_ptr : (const int64_t*)nullptr, U);
        }
        A.heads = d_heads; A.items = d_items;
        const int64_t n_a = n_order;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_a, (int64_t)std::max(1, ctx->n_cu) * per_cu));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ctx->ev2, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("order: ") + hipGetErrorString(e)); break; }
        void (*kern)(cf_dist_args) = region26 ? cf_dist_kernel<cf_tab_region26> : region ? cf_dist_kernel<cf_tab_region> : !narrow ? (wide16 ? cf_dist_kernel<cf_tab_wide16> : cf_dist_kernel<cf_tab_wide>)
                                   : narrow_db == 8 ? cf_dist_kernel<cf_tab_narrow_t<8>> : narrow_db == 7 ? cf_dist_kernel<cf_tab_narrow_t<7>>
                                   : narrow_db == 6 ? cf_dist_kernel<cf_tab_narrow_t<6>> : cf_dist_kernel<cf_tab_narrow_t<5>>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("dist LDS attribute: ") + hipGetErrorString(e)); break; }
        // the edge output: edge_cap rows + the unused rest of one chunk per workgroup (the holes; closed below), so that every
        // selected edge is in memory whenever edge_cap >= their number
        {
            A.edge_chunk = ctx->dist_edge_chunk > 0 ? (unsigned long long)ctx->dist_edge_chunk : DIST_EDGE_CHUNK;
            const int64_t rows = edge_cap > 0 ? edge_cap + (int64_t)grid * (int64_t)A.edge_chunk : 0;
            if ((rc = cf_alloc_t(ctx, &ctx->d_edges, (size_t)rows * 4, "edges"))) break;
            ctx->edge_cap = rows;
            A.edges = ctx->d_edges; A.edge_cap = (unsigned long long)rows;
        }
        n_holes_alloc = (size_t)grid + 1;
        if ((rc = cf_alloc_t(ctx, &d_holes, 2 * n_holes_alloc, "edge holes"))) break;
        A.holes = d_holes;
        if (hipMemsetAsync(d_cnt + 6, 0, 16, ctx->stream) != hipSuccess) { rc = cf_fail(ctx, -5, "counter reset"); break; }      // [6] selected edges, [7] holes
        if (n_a > 0 && max_d >= min_d_eff && n_post > 0) {
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)block), lds, ctx->stream, A);
            e = hipGetLastError();
            if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_dist_kernel: ") + hipGetErrorString(e)); break; }
        }
        e = hipEventRecord(ctx->ev3, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 64, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_dist_edges: ") + hipGetErrorString(e)); break; }
#if defined(CF_DIST_DIAG_COUNT)
        {
            unsigned long long st[8];
            if (hipMemcpy(st, d_cnt + 8, 64, hipMemcpyDeviceToHost) == hipSuccess)
                std::fprintf(stderr, "[cf_dist diag] parked-insert loop trips=%llu overflow calls=%llu parked inserts run=%llu drains=%llu inserts drained=%llu cycles(sum over waves): parked runs=%llu drains=%llu table sweeps=%llu; waves=%d passes=%llu\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], grid * (block / 64), h_cnt[5]);
        }
#endif
#if defined(CF_DIST_STAMPS)
        {
            unsigned long long st[8];
            if (hipMemcpy(st, d_cnt + 8, 64, hipMemcpyDeviceToHost) == hipSuccess)
                std::fprintf(stderr, "[cf_dist stamps] pop=%llu prologue=%llu sketch=%llu clear=%llu stream=%llu filter=%llu write=%llu sweep_end_wait=%llu (shader cycles summed over %d workgroups; passes=%llu)\n",
                             st[0], st[1], st[6], st[2], st[3], st[4], st[5], st[7], grid, h_cnt[5]);
        }
#endif
        if (h_cnt[4] & 2ull) { rc = cf_fail(ctx, -5, "cf_dist_edges: the kernel's dynamic LDS does not begin at address 0"); break; }
        if (h_cnt[4]) { rc = cf_fail(ctx, -34, "cf_dist_edges: (b,d) table could not be partitioned far enough"); break; }
        // close the holes of the chunked edge output: the valid rows at or above n_valid move into the holes below it
        {
            const unsigned long long a_cap = std::min<unsigned long long>(h_cnt[0], A.edge_cap);      // rows that exist in memory
            std::vector<unsigned long long> hh(2 * (size_t)h_cnt[7]);
            if (!hh.empty() && (hipMemcpyAsync(hh.data(), d_holes, hh.size() * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) { rc = cf_fail(ctx, -5, "edge holes copy"); break; }
            std::vector<std::pair<unsigned long long, unsigned long long>> holes;      // [start, end) clipped to the rows in memory, sorted
            for (size_t i = 0; i + 1 < hh.size(); i += 2) {
                const unsigned long long s0 = std::min(hh[i], a_cap), s1 = std::min(hh[i] + hh[i + 1], a_cap);
                if (s1 > s0) holes.emplace_back(s0, s1);
            }
            std::sort(holes.begin(), holes.end());
            unsigned long long hole_rows = 0;
            for (auto& h : holes) hole_rows += h.second - h.first;
            const unsigned long long n_valid = a_cap - hole_rows;
            n_stored = (int64_t)std::min<unsigned long long>(n_valid, (unsigned long long)edge_cap);
            std::vector<unsigned long long> dst_start, dst_pre, src_start, src_pre;
            unsigned long long dsum = 0, ssum = 0, at = n_valid;
            for (auto& h : holes) {
                if (h.first < n_valid) { const unsigned long long e1 = std::min(h.second, n_valid); dsum += e1 - h.first; dst_start.push_back(h.first); dst_pre.push_back(dsum); }
                if (h.second > n_valid) {      // valid rows between `at` and this hole's part above n_valid
                    const unsigned long long h0 = std::max(h.first, n_valid);
                    if (h0 > at) { ssum += h0 - at; src_start.push_back(at); src_pre.push_back(ssum); }
                    at = h.second;
                }
            }
            if (a_cap > at) { ssum += a_cap - at; src_start.push_back(at); src_pre.push_back(ssum); }
            if (dsum != ssum) { rc = cf_fail(ctx, -5, "cf_dist_edges: internal error, edge holes do not match the rows above them"); break; }
            if (dsum) {
                unsigned long long* d_seg = nullptr;
                const size_t n_seg = src_start.size() * 2 + dst_start.size() * 2;
                if ((rc = cf_alloc_t(ctx, &d_seg, n_seg, "edge move lists"))) break;
                std::vector<unsigned long long> pack;
                pack.insert(pack.end(), src_start.begin(), src_start.end()); pack.insert(pack.end(), src_pre.begin(), src_pre.end());
                pack.insert(pack.end(), dst_start.begin(), dst_start.end()); pack.insert(pack.end(), dst_pre.begin(), dst_pre.end());
                hipError_t e2 = hipMemcpyAsync(d_seg, pack.data(), n_seg * 8, hipMemcpyHostToDevice, ctx->stream);
                if (e2 == hipSuccess) {
                    hipLaunchKernelGGL(cf_edge_compact_kernel, dim3((unsigned)cf_grid_for((int64_t)dsum, 256, max_blocks)), dim3(256), 0, ctx->stream, ctx->d_edges,
                                       (const unsigned long long*)d_seg, (const unsigned long long*)(d_seg + src_start.size()), (int)src_start.size(),
                                       (const unsigned long long*)(d_seg + 2 * src_start.size()), (const unsigned long long*)(d_seg + 2 * src_start.size() + dst_start.size()), (int)dst_start.size(), dsum);
                    e2 = hipGetLastError();
                }
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
                cf_release_t(ctx, d_seg, n_seg);
                if (e2 != hipSuccess) { rc = cf_fail(ctx, -5, std::string("edge compaction: ") + hipGetErrorString(e2)); break; }
            }
        }
        // (the stage ends behind the compaction of the chunked edge output, not behind the kernel)
        if (hipEventRecord(ctx->ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ctx->ev1) != hipSuccess) { rc = cf_fail(ctx, -5, "cf_dist_edges: final event"); break; }
        (void)hipEventElapsedTime(&ctx->times.dist_ms, ctx->ev0, ctx->ev1);
        (void)hipEventElapsedTime(&ctx->times.postings_ms, ctx->ev0, ctx->ev2);
        (void)hipEventElapsedTime(&ctx->times.dist_kernel_ms, ctx->ev2, ctx->ev3);
    } while (0);
    if (d_holes) cf_release_t(ctx, d_holes, 2 * n_holes_alloc);
    if (d_items) cf_release_t(ctx, d_items, (size_t)n_item_slots + 64);
    if (d_heads) cf_release_t(ctx, d_heads, (size_t)n_order);
    if (d_ibase) cf_release_t(ctx, d_ibase, (size_t)n_order + 1);
    if (d_ialloc) cf_release_t(ctx, d_ialloc, (size_t)n_order + 1);
    if (d_icnt) cf_release_t(ctx, d_icnt, (size_t)n_order + 1);
    if (d_order) cf_release_t(ctx, d_order, (size_t)n_a_alloc);
    if (d_otmp) cf_release_t(ctx, d_otmp, (size_t)n_a_alloc);
    if (d_okeys) cf_release_t(ctx, d_okeys, (size_t)n_a_alloc);
    if (d_first) cf_release_t(ctx, d_first, (size_t)K + 1);
    if (d_cnt) cf_release_t(ctx, d_cnt, n_cnt);
    if (d_entry_i) cf_release_t(ctx, d_entry_i, (size_t)v_n_entries + 1);
    if (d_packed) cf_release_t(ctx, d_packed, (size_t)v_n_entries + DIST_ITEM);
    if (d_entry_i8) cf_release_t(ctx, d_entry_i8, (size_t)v_n_entries + 4 * DIST_ITEM);
    if (d_urange) cf_release_t(ctx, d_urange, (size_t)U + 1);
    if (d_rbeg) cf_release_t(ctx, d_rbeg, (size_t)U + 1);
    if (d_rend) cf_release_t(ctx, d_rend, (size_t)U + 1);
    if (d_post) cf_release_t(ctx, d_post, (size_t)n_post);
    if (d_post_ptr) cf_release_t(ctx, d_post_ptr, (size_t)K + 1);
    if (d_cursor) cf_release_t(ctx, d_cursor, (size_t)K + 1);
    if (d_pcnt) cf_release_t(ctx, d_pcnt, (size_t)K + 1);
    if (rc) return rc;
    ctx->stats.n_edges = (int64_t)h_cnt[6];
    ctx->stats.n_emissions = (int64_t)h_cnt[1] - (int64_t)h_cnt[3];      // partner entries swept, less those that are the first k-mer itself (a != b in the reference)
    ctx->stats.n_spilled = (int64_t)h_cnt[2];
    ctx->stats.n_dist_passes = (int64_t)h_cnt[5];
    ctx->n_edges_stored = n_stored;      // every selected edge when edge_cap allowed it, else the valid rows below the cap
    CF_TRY(cf_refresh_unique_count(ctx));
    if (n_edges) *n_edges = (int64_t)h_cnt[6];
    return 0;
}

}  // extern "C"
