// cf_common.h — shared declarations of the gfx950 device pipeline (libcfhip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "cfhip.h"

// Every kernel uses one dynamic-LDS window and carves it up itself.
extern __shared__ __attribute__((aligned(16))) unsigned char cf_lds[];

#define CF_WAVE 64

// The dynamic LDS window begins at address 0 of the workgroup's LDS (no kernel of this library has static LDS).  A pointer
// made from that NUMBER lets the compiler fold an array's offset into the offset field of the ds_ instruction; through the
// symbol cf_lds it puts a `v_add_u32 v, 0, v` in front of every access (the symbol's address is assigned after instruction
// selection): four of the 48 vector instructions of cf_dist_kernel's sketch step.  cf_lds_base_ok() is the kernel's check
// of the premise.  (The host emulator of tests/emu defines both itself.)
#ifndef cf_lds_at
typedef __attribute__((address_space(3))) unsigned char cf_lds_u8;
__device__ __forceinline__ unsigned char* cf_lds_at(uint32_t off) { return (unsigned char*)((cf_lds_u8*)(uintptr_t)(off + 16u) - 16); }
__device__ __forceinline__ bool cf_lds_base_ok() { return (uint32_t)(uintptr_t)(cf_lds_u8*)cf_lds == 0u; }
#endif

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ uint64_t cf_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint32_t cf_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// ASCII base -> 2-bit code with A<C<G<T (A 0x41, C 0x43, G 0x47, T 0x54); bit 5 (the case) plays no part
__device__ __forceinline__ uint32_t cf_base2(uint32_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }
// upper-case A, C, G or T?  (x = c - 'A': bits 0, 2, 6, 19 of the mask)  A window holding anything else has no 2-bit code:
// the device paths skip it (what the reference does with such windows: centroflye_amd/_host.py exotic_summary)
__device__ __forceinline__ bool cf_is_acgt(uint32_t c) { const uint32_t x = c - 0x41u; return x < 20u && ((0x80045u >> x) & 1u); }
__device__ __forceinline__ bool cf_is_acgt_nocase(uint32_t c) { return cf_is_acgt(c & 0xDFu); }

// bit 63 marks an occupied slot in every k-mer keyed table (k <= 31 -> key < 2^62)
#define CF_OCC (1ull << 63)

// 16-byte slot of the HBM k-mer table: key | CF_OCC, then pres (low 32) | multi (high 32)
struct alignas(16) cf_slot {
    unsigned long long key;
    unsigned long long val;
};

// ---------------------------------------------------------------- host side
struct cf_devbuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct cf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    std::string err;
    int n_cu = 0;
    int64_t hbm_total = 0;
    size_t live = 0;
    // freed device blocks are kept and reused (hipMalloc / hipFree of the per-call work buffers — hundreds of MB each —
    // cost up to tens of ms per step and vary from box to box): size -> block, and the true size of every block handed out
    std::multimap<size_t, void*> pool;
    std::unordered_map<void*, size_t> block_bytes;
    size_t pooled = 0;
    size_t pool_max = (size_t)128 << 30;      // bytes kept for reuse (cf_create: 85 % of the device's memory)

    // reads / units
    uint8_t* d_bases = nullptr;
    int64_t* d_read_off = nullptr;
    int64_t* d_unit_ptr = nullptr;
    int64_t* d_unit_start = nullptr;
    int64_t* d_unit_end = nullptr;
    std::vector<int64_t> h_read_off, h_unit_ptr;
    int64_t max_unit_len = 0;     // bases of the longest unit (bounds the entries of a cloud)
    int64_t n_reads = 0, n_bases = 0, n_units = 0;
    bool has_exotic = false;     // some base is not upper-case A, C, G, T

    // A1 table
    cf_slot* d_table = nullptr;
    uint64_t table_cap = 0;      // slots in use (a power of two)
    uint64_t table_alloc = 0;    // slots allocated (>= table_cap: a smaller table reuses a larger allocation)
    bool table_dense = false;    // the table is a dense array of table_cap occupied slots (cf_count2.hip): scan it, do not probe it
    int k = 0;

    // k-mer set + lookup table
    unsigned long long* d_kmers = nullptr;
    int64_t n_kmers = 0;
    int set_k = 0;
    cf_slot* d_lut = nullptr;        // lookup table k-mer -> rank: lut_cap slots {k-mer | CF_OCC, rank}
    uint64_t lut_cap = 0;
    uint32_t* d_lut_pre = nullptr;   // 8 bits per lookup slot: Bloom words tested before the lookup table (most windows are not in the set)
    uint64_t lut_pre_words = 0;

    // clouds
    int64_t* d_cloud_ptr = nullptr;  // U+1
    int32_t* d_entries = nullptr;
    int64_t n_entries = 0;
    bool have_clouds = false;

    // multi-GPU (cf_exchange.hip): the transport, and the all-gathered clouds of every rank's reads — what the distance
    // stage works on when present (units in rank-major order; reads and bases stay those of the local shard)
    struct cf_comm* comm = nullptr;
    bool have_gview = false;
    int64_t g_reads = 0, g_units = 0, g_entries = 0;
    int64_t* g_unit_ptr = nullptr;    // g_reads + 1
    int64_t* g_cloud_ptr = nullptr;   // g_units + 1
    int32_t* g_entries_d = nullptr;   // g_entries
    std::vector<int64_t> g_h_unit_ptr;
    int64_t exchange_bytes = 0;       // bytes this rank sent in the last cf_exchange_table

    // edges / unique bitmap
    uint32_t* d_edges = nullptr;  // n x 4
    int64_t edge_cap = 0, n_edges_stored = 0;
    uint32_t* d_unique_bits = nullptr;
    int64_t unique_words = 0;

    // host <-> device copies of the caller's (pageable) buffers go through pinned staging slots, one per copy thread
    // (cf_api.hip: cf_copy_h2d / cf_copy_d2h)
    static constexpr int kCopyThreads = 16;      // slots; CF_COPY_THREADS (1 .. 16, default 16) picks how many are used
    void* pin_slot[kCopyThreads] = {nullptr};
    hipStream_t pin_stream[kCopyThreads] = {nullptr};
    size_t pin_bytes = 0;
    int copy_threads = 16;       // CF_COPY_THREADS (1 .. 16), read once by cf_create (round 5: 16 staged threads move 54 GB/s D2H, 8 moved 34 - 40)

    cf_stats stats{};
    cf_times times{};

    // knobs
    int dist_block = 0;      // threads per workgroup; 0 = chosen with dist_wgs
    int dist_wgs = 0;        // workgroups per CU the LDS is split between; 0 = 2 x 512 threads or 1 x 1024, by the pair emissions per first k-mer
    int dist_slots = 0;      // LDS budget of the (b,d) table in 8-byte units; 0 = all that is left next to the work lists
    int dist_wide = 0;       // 1 forces the 8-byte-slot table layout (tests)
    int dist_post_atomics = 0;   // 1 builds the postings with the histogram + fill passes of atomics instead of the sort (tests, A/B runs)
    int dist_hot_cap = 0;    // > 0: cap on the filter's hot-slot list (tests)
    int lut_shift = -1;      // the k-mer lookup table of A3 has (2 x k-mers, rounded up to a power of two) << lut_shift slots; -1 = by the set's size
    int dist_hot_entries = 32768;   // first k-mers with more partner entries keep no hot-slot list (it would overflow: ~5 % of the pairs' keys reach min_cov); -1: always keep it
    int dist_regions = 0;    // 1, 2, 4, 8: force the region layout of the 6-byte slots with at least that many regions (tests), 0 = only when the ranks need it
    int dist_region_bytes = 0; // 1: the region layout streams rank and unit index apart (round 3) even where the 4-byte stream of cf_tab_region26 applies (tests)
    int dist_dbits = 0;      // 5 .. 8: upper limit of the distance-field bits of the 6-byte-slot layout (tests), 0 = as many as the k-mer ranks leave
    int dist_fill_pct = 70;  // a (b,d) table pass is split when more than this share of the slots is in use
    int dist_edge_chunk = 0; // > 0: edge rows a workgroup reserves per global atomic (tests: small chunks cross often), 0 = 8192
    int dist_int_thr = 1;    // 1: rel_threshold == 0.8 is tested as 5 cnt >= 4 total; 0: always the double division (tests)
    int dist_sketch = 1;     // 0: every (b,d) pair goes to the exact table (no counting sketch first)
    int dist_sketch_bits = 0;   // bits of a sketch counter: 0 = 4 when min_cov <= 9, else 8; 8 forces bytes
    int dist_est_pct = 80;   // expected distinct (b,d) keys per 100 pair emissions: sizes the initial number of table partitions
    int dist_stage = 2048;   // edges of a pass whose rows do not fit the rest of the workgroup's output chunk, staged in LDS (more: a sweep of the marked slots)
    int place_chunk = 2, place_grid = 0;     // cloud entries per wave step of that kernel (1 .. 64), its workgroups (0 = one per CU)
    int place_fused = 1;         // 1: the greedy iteration is arg-max + (pick, add, score updates) = 2 kernels; 0: 3 kernels with an event list
    int place_mode = 2;          // 2: per-read score regions, one kernel per greedy iteration (cf_place2.hip); 1: the hash-map path of rounds 1-3 (cf_place.hip)
    int place_block = 0;         // cf_place2: threads per workgroup of the iteration kernel (128 .. 1024, a multiple of 128; 0 = 1024)
    int place_row_words = 0;     // cf_place2: 32-bit words of a posting row (32 or 64); 0 = the smaller one that holds the longest posting list
    int place_long_rescans = 2;  // cf_place2: long rescans (reads with more than four candidate rows) per greedy iteration above which the run goes to the hash-map path (-1: at the first look, tests)
    int place_cmap_bits = 0;     // cf_place2: log2 of the first capacity of the contig's overflow map (0 = entries / 8, at least 2^21; tests force tiny maps that have to grow)
    int place_slots_per_unit = 0; // cf_place2: score-region slots per unit of a read (0 = 48); doubled-up automatically when a region fills
    int place_l3 = 0;            // cf_place2: third level of the arg-max (best candidate per group of 64-read blocks): 1 = on; 0 / 2 = off (measured: no gain at 500 000 reads)
    int place_l3_shift = 0;      // cf_place2: log2 of the blocks per group (0 = 6: groups of 64 blocks = 4 096 reads; tests use small groups at small read sets)
    int count_mode = 1;          // 1: sort and reduce (cf_count2.hip) when it applies; 0: the atomic table of round 1 (cf_count.hip)
    int count_bits = 0;          // bucket bits of the sort-and-reduce path; 0 = from the number of windows (tests force small / large values)
    int count_slots = 4096;
    int count_tile = 16;
    int64_t comm_round_bytes = (int64_t)256 << 20;   // bytes per pair and round of the multi-GPU exchanges (tests force small rounds)
    int comm_self_p2p = 0;       // 1: a rank's message to itself goes through ncclSend / ncclRecv too (tests: the p2p path on one GPU)
};

int cf_fail(cf_ctx* ctx, int code, const std::string& msg);
// copies between device memory and memory of the CALLER (pageable): staged through pinned slots by several threads; `host`
// may also be a device pointer (the multi-GPU exchange hands device buffers to some getters): then one plain copy
int cf_copy_h2d(cf_ctx* ctx, void* dev, const void* host, size_t bytes);
int cf_copy_d2h(cf_ctx* ctx, void* host, const void* dev, size_t bytes);
int cf_alloc(cf_ctx* ctx, void** p, size_t bytes, const char* what);
void cf_release(cf_ctx* ctx, void* p, size_t bytes);

template <class T>
inline int cf_alloc_t(cf_ctx* ctx, T** p, size_t n, const char* what) {
    return cf_alloc(ctx, (void**)p, n * sizeof(T), what);
}
template <class T>
inline void cf_release_t(cf_ctx* ctx, T*& p, size_t n) {
    cf_release(ctx, (void*)p, n * sizeof(T));
    p = nullptr;
}

#define CF_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return cf_fail(ctx, -5, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)
#define CF_TRY(expr)                  \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != 0) return rc_;     \
    } while (0)
#define CF_KERNEL_CHECK(name)                                                                 \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess)                                                                 \
            return cf_fail(ctx, -5, std::string("launch of ") + name + ": " + hipGetErrorString(e_)); \
    } while (0)

static inline uint64_t cf_pow2_ceil(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
static inline int cf_grid_for(int64_t items, int per_block, int max_blocks) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

extern "C" int cf_comm_free(cf_ctx* ctx);   // cf_exchange.hip

// primitives (cf_prims.hip)
int cf_scan_exclusive_i64(cf_ctx* ctx, const int64_t* d_in, int64_t* d_out, int64_t n, int64_t* total);
int cf_scan_exclusive_u32_to_i64(cf_ctx* ctx, const uint32_t* d_in, int64_t* d_out, int64_t n, int64_t* total);
// LSD radix sort of 64-bit keys on `bits` low bits; result lands in d_keys (d_tmp is scratch)
int cf_radix_sort_u64(cf_ctx* ctx, unsigned long long* d_keys, unsigned long long* d_tmp, int64_t n, int bits);
int cf_radix_sort_u64_any(cf_ctx* ctx, unsigned long long* d_keys, unsigned long long* d_tmp, int64_t n, int bits, unsigned long long** result);
int cf_radix_sort_rec16(cf_ctx* ctx, void* d_recs, void* d_tmp, int64_t n, const int* words, const int* bits, int n_fields);
