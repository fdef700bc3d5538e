// cf_dist.hip — A5 (k-mer pair / unit-distance histogram) fused with A6 (edge filter).
//
// Reference: scripts/distance_based_kmer_recruitment.py:85-128 counts, for every read, every
// pair of unit clouds d units apart (min_d <= d <= max_d) and every ordered pair (a in C_i,
// b in C_{i+d}, a != b), cnt[d][a][b] += 1; :131-149 keeps (d, a, b, cnt) with cnt >= min_cov
// and cnt / sum_d' cnt[d'][a][b] >= 0.8 (true division), and marks a and b as selected.
//
// Device design (the dominant kernel of the path; HBM/L2 streaming + LDS atomics, no MFMA):
//   * dist_cnt[d][a] is owned by the first k-mer a in the reference; so is it here: one
//     workgroup owns one a at a time (dynamic queue), walks a's posting list
//     (the (read, unit) clouds that contain a), streams the later unit clouds of those reads
//     (contiguous int32 CSR ranges, coalesced, 4 B per pair emission) and counts (b, d) in an
//     LDS open-addressed table of 64-bit slots [b:32 | d:8 | cnt:24] with one LDS atomic per
//     emission.  The slot hash depends on b only, so all d of one b share a probe chain and
//     sum_d cnt is a chain walk — both filters run in LDS and only selected edges reach HBM.
//   * a table that would overflow is split by a second hash of b into 2, 4, ... partitions,
//     each processed (re-streamed) on its own: exact, no HBM spill.
//   * first k-mers partition across GPUs (a % n_parts == part) with no reduction.
#include "cf_common.h"

#include <cstdlib>

void cf_free_edges(cf_ctx* c);
int cf_refresh_unique_count(cf_ctx* ctx);  // cf_clouds.hip

#define DIST_NP_CAP 256
#define DIST_STAGE_CAP 2048              /* selected slots staged per table pass (u16 slot indices) */
#define DIST_STACK 112
#define DIST_UNROLL 4
#define DIST_CNT_MASK 0x7FFFFFu          /* 23-bit count */
#define DIST_SEL_BIT (1ull << 23)        /* slot selected by the A6 filter */

__global__ void __launch_bounds__(256)
cf_post_hist_kernel(const int32_t* __restrict__ entries, int64_t e0, int64_t e1, uint32_t* __restrict__ cnt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e1; i += stride) atomicAdd(&cnt[entries[i]], 1u);
}

__global__ void __launch_bounds__(256)
cf_post_fill_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t u0, int64_t u1,
                    const int64_t* __restrict__ post_ptr, uint32_t* __restrict__ cursor, int32_t* __restrict__ post,
                    uint32_t* __restrict__ first_unit) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = u0 + wave; u < u1; u += n_waves) {
        const int64_t a = cloud_ptr[u], b = cloud_ptr[u + 1];
        for (int64_t e = a + lane; e < b; e += 64) {
            const int32_t x = entries[e];
            post[post_ptr[x] + atomicAdd(&cursor[x], 1u)] = (int32_t)u;
            if (first_unit[x] > (uint32_t)u) atomicMin(&first_unit[x], (uint32_t)u);
        }
    }
}

__global__ void __launch_bounds__(256)
cf_unit_rend_kernel(const int64_t* __restrict__ unit_ptr, int64_t n_reads, int32_t* __restrict__ rend, int32_t* __restrict__ rbeg) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
        const int64_t a = unit_ptr[r], b = unit_ptr[r + 1];
        for (int64_t u = a; u < b; ++u) { rend[u] = (int32_t)b; rbeg[u] = (int32_t)a; }
    }
}

// per cloud entry the index of its unit inside its read (one wave per unit)
__global__ void __launch_bounds__(256)
cf_entry_unit_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ rbeg, int64_t n_units, uint16_t* __restrict__ entry_i) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < n_units; u += n_waves) {
        const uint16_t i = (uint16_t)(u - rbeg[u]);
        for (int64_t e = cloud_ptr[u] + lane; e < cloud_ptr[u + 1]; e += 64) entry_i[e] = i;
    }
}

// keys (first posting unit << 32 | a) of the first k-mers of this partition that have postings
__global__ void __launch_bounds__(256)
cf_order_keys_kernel(const uint32_t* __restrict__ pcnt, const uint32_t* __restrict__ first_unit, int64_t n_kmers, int part, int n_parts,
                     unsigned long long* __restrict__ keys, unsigned long long* __restrict__ n_out) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (n_kmers + stride - 1) / stride;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t a = rd * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool take = a < n_kmers && (a % n_parts) == part && pcnt[a] > 0;
        const unsigned long long m = __ballot(take);
        if (m) {
            unsigned long long base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(n_out, (unsigned long long)__popcll(m));
            base = __shfl(base, leader);
            if (take) keys[base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)first_unit[a] << 32) | (unsigned long long)a;
        }
    }
}
__global__ void __launch_bounds__(256)
cf_order_extract_kernel(const unsigned long long* __restrict__ keys, int64_t n, int32_t* __restrict__ order) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) order[i] = (int32_t)(keys[i] & 0xFFFFFFFFull);
}

struct cf_dist_args {
    const int64_t* post_ptr;
    const int32_t* post;
    const int64_t* cloud_ptr;
    const int32_t* entries;
    const int32_t* unit_rend;      // one past the last unit of the unit's read
    const int32_t* unit_rbeg;      // first unit of the unit's read
    const uint16_t* entry_i;       // per cloud entry: index of its unit inside its read
    int64_t n_kmers;
    int32_t part, n_parts;
    int32_t min_d, max_d;       // min_d already clamped to >= 1
    uint32_t min_cov;
    double thr;
    int32_t slots;
    uint32_t fill_limit;
    uint32_t stage_cap;            // <= DIST_STAGE_CAP
    uint32_t* edges;
    unsigned long long edge_cap;
    const int32_t* order;          // first k-mers of this partition, sorted by their first posting (locality)
    int64_t n_order;
    unsigned long long* counters;  // [0] edges [1] emissions [2] spilled a [4] error flags [5] passes; [16 + 16 x] queue head x
    uint32_t* unique_bits;
};

// ---------------------------------------------------------------------------------------------------
// The (b, d) table lives in LDS and is organised in 32-byte buckets read with two ds_read_b128: a probe
// inspects a whole bucket with straight-line code, so a wave does not run a per-lane probe loop in the
// common case.  A key lives in the first bucket, starting at home(b), that held a match or an empty slot
// when it was inserted; slots of a bucket fill in ascending order and never empty, so every (b, .) key sits
// between home(b) and the first bucket that still has an empty slot, and a key is never inserted twice.
// Two layouts with one interface:
//   cf_tab_wide    4 slots of 64 bits  [b:32 | d:8 | sel:1 | cnt:23]            any k-mer set size
//   cf_tab_narrow  8 keys of 32 bits   [d:8 | b:24] + 8 x 16-bit [sel:1 | cnt:15]  (6 bytes per slot: a third
//                  more slots in the same LDS, half as many full buckets, 32-bit compares) when the set has
//                  < 2^24 - 1 k-mers and no k-mer has more than 32767 postings
// b is a dense rank, so one odd multiplier spreads it; the home bucket is the high product.
struct alignas(16) cf_u64x2 { unsigned long long x, y; };
struct alignas(16) cf_u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t cf_dist_hash(uint32_t b) { return b * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t cf_dist_home(uint32_t b, uint32_t n_buckets) {
    return (uint32_t)(((unsigned long long)cf_dist_hash(b) * (unsigned long long)n_buckets) >> 32);
}

struct cf_tab_wide {
    static constexpr uint32_t kSlotBytes = 8, kPerBucket = 4;
    struct bucket { cf_u64x2 lo, hi; };
    unsigned long long* tab;
    __device__ __forceinline__ void init(unsigned char* lds, uint32_t) { tab = (unsigned long long*)lds; }
    __device__ __forceinline__ void clear(uint32_t slots, uint32_t t, uint32_t nt) const {
        const cf_u64x2 z{0ull, 0ull};
        for (uint32_t s = t; s < (slots >> 1); s += nt) ((cf_u64x2*)tab)[s] = z;
    }
    __device__ __forceinline__ bucket read(uint32_t bk) const { return bucket{*(const cf_u64x2*)&tab[4 * bk], *(const cf_u64x2*)&tab[4 * bk + 2]}; }
    static __device__ __forceinline__ bool is(unsigned long long v, uint32_t b, uint32_t dd) { return (uint32_t)(v >> 32) == b && ((uint32_t)v >> 24) == dd; }
    // branch-free: one bit per slot, then find-first-set (nested ?: chains compile to a cascade of exec-mask branches)
    static __device__ __forceinline__ int match(const bucket& k, uint32_t b, uint32_t dd) {   // dd >= 1: an empty slot never matches
        const uint32_t m = (uint32_t)is(k.lo.x, b, dd) | ((uint32_t)is(k.lo.y, b, dd) << 1) | ((uint32_t)is(k.hi.x, b, dd) << 2) | ((uint32_t)is(k.hi.y, b, dd) << 3);
        return __ffs((int)m) - 1;
    }
    static __device__ __forceinline__ int empty(const bucket& k) {
        const uint32_t m = (uint32_t)(k.lo.x == 0ull) | ((uint32_t)(k.lo.y == 0ull) << 1) | ((uint32_t)(k.hi.x == 0ull) << 2) | ((uint32_t)(k.hi.y == 0ull) << 3);
        return __ffs((int)m) - 1;
    }
    __device__ __forceinline__ void add(uint32_t bk, int i) const { atomicAdd(&tab[4 * bk + i], 1ull); }
    // claim slot i of bucket bk for (b, dd): 0 = claimed (count 1), 1 = the same key got there first (counted), 2 = another key
    __device__ __forceinline__ unsigned long long claim_issue(uint32_t bk, int i, uint32_t b, uint32_t dd) const {
        return atomicCAS(&tab[4 * bk + i], 0ull, ((unsigned long long)b << 32) | ((unsigned long long)dd << 24) | 1ull);
    }
    __device__ __forceinline__ int claim_finish(unsigned long long old, uint32_t bk, int i, uint32_t b, uint32_t dd) const {
        if (old == 0ull) return 0;
        if (is(old, b, dd)) { add(bk, i); return 1; }
        return 2;
    }
    // filter side: slot s -> (b, dd, cnt) or false when empty
    __device__ __forceinline__ bool get(uint32_t s, uint32_t& b, uint32_t& dd, uint32_t& cnt) const {
        const unsigned long long v = tab[s];
        b = (uint32_t)(v >> 32); dd = ((uint32_t)v >> 24) & 0xFFu; cnt = (uint32_t)v & 0x7FFFFFu;
        return v != 0ull;
    }
    __device__ __forceinline__ unsigned long long total_of(uint32_t b, uint32_t n_buckets) const {
        unsigned long long total = 0;
        uint32_t bk = cf_dist_home(b, n_buckets);
        for (uint32_t probe = 0; probe < n_buckets; ++probe) {
            const bucket k = read(bk);
            if ((uint32_t)(k.lo.x >> 32) == b && k.lo.x) total += k.lo.x & 0x7FFFFFull;
            if ((uint32_t)(k.lo.y >> 32) == b && k.lo.y) total += k.lo.y & 0x7FFFFFull;
            if ((uint32_t)(k.hi.x >> 32) == b && k.hi.x) total += k.hi.x & 0x7FFFFFull;
            if ((uint32_t)(k.hi.y >> 32) == b && k.hi.y) total += k.hi.y & 0x7FFFFFull;
            if (empty(k) >= 0) break;
            bk = bk + 1 == n_buckets ? 0u : bk + 1;
        }
        return total;
    }
    __device__ __forceinline__ void mark(uint32_t s) const { atomicOr(&tab[s], 1ull << 23); }
    __device__ __forceinline__ bool marked(uint32_t s) const { return (tab[s] >> 23) & 1ull; }
};

struct cf_tab_narrow {
    static constexpr uint32_t kSlotBytes = 6, kPerBucket = 8;
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    struct bucket { cf_u32x4 lo, hi; };
    uint32_t* keys;     // slots x 32-bit [d:8 | b:24]
    uint32_t* cnt32;    // slots x 16-bit counts, two per word
    __device__ __forceinline__ void init(unsigned char* lds, uint32_t slots) { keys = (uint32_t*)lds; cnt32 = keys + slots; }
    __device__ __forceinline__ void clear(uint32_t slots, uint32_t t, uint32_t nt) const {
        const cf_u32x4 e{kEmpty, kEmpty, kEmpty, kEmpty}, z{0u, 0u, 0u, 0u};
        for (uint32_t s = t; s < (slots >> 2); s += nt) ((cf_u32x4*)keys)[s] = e;
        for (uint32_t s = t; s < (slots >> 3); s += nt) ((cf_u32x4*)cnt32)[s] = z;
    }
    __device__ __forceinline__ bucket read(uint32_t bk) const { return bucket{*(const cf_u32x4*)&keys[8 * bk], *(const cf_u32x4*)&keys[8 * bk + 4]}; }
    static __device__ __forceinline__ uint32_t key_of(uint32_t b, uint32_t dd) { return (dd << 24) | b; }
    // branch-free: one bit per slot, then find-first-set (nested ?: chains compile to a cascade of exec-mask branches)
    // bit i set <=> slot i differs from q: min(k ^ q, 1) stays in vector registers (no compare -> SGPR -> select hazard)
    static __device__ __forceinline__ uint32_t ne_bit(uint32_t k, uint32_t q) { return min(k ^ q, 1u); }
    static __device__ __forceinline__ uint32_t eq_mask(const bucket& k, uint32_t q) {
        const uint32_t ne = ne_bit(k.lo.x, q) | (ne_bit(k.lo.y, q) << 1) | (ne_bit(k.lo.z, q) << 2) | (ne_bit(k.lo.w, q) << 3)
                          | (ne_bit(k.hi.x, q) << 4) | (ne_bit(k.hi.y, q) << 5) | (ne_bit(k.hi.z, q) << 6) | (ne_bit(k.hi.w, q) << 7);
        return ne ^ 0xFFu;
    }
    static __device__ __forceinline__ int match(const bucket& k, uint32_t b, uint32_t dd) { return __ffs((int)eq_mask(k, key_of(b, dd))) - 1; }
    static __device__ __forceinline__ int empty(const bucket& k) { return __ffs((int)eq_mask(k, kEmpty)) - 1; }
    __device__ __forceinline__ void add(uint32_t bk, int i) const { const uint32_t s = 8 * bk + (uint32_t)i; atomicAdd(&cnt32[s >> 1], 1u << ((s & 1u) * 16u)); }
    __device__ __forceinline__ uint32_t claim_issue(uint32_t bk, int i, uint32_t b, uint32_t dd) const { return atomicCAS(&keys[8 * bk + i], kEmpty, key_of(b, dd)); }
    __device__ __forceinline__ int claim_finish(uint32_t old, uint32_t bk, int i, uint32_t b, uint32_t dd) const {
        const bool mine = old == kEmpty, same = old == key_of(b, dd);
        if (mine | same) add(bk, i);          // the claimed key, or the same key claimed by someone else: count it
        return mine ? 0 : same ? 1 : 2;
    }
    __device__ __forceinline__ bool get(uint32_t s, uint32_t& b, uint32_t& dd, uint32_t& cnt) const {
        const uint32_t q = keys[s];
        b = q & 0xFFFFFFu; dd = q >> 24; cnt = (cnt32[s >> 1] >> ((s & 1u) * 16u)) & 0x7FFFu;
        return q != kEmpty;
    }
    __device__ __forceinline__ unsigned long long total_of(uint32_t b, uint32_t n_buckets) const {
        unsigned long long total = 0;
        uint32_t bk = cf_dist_home(b, n_buckets);
        for (uint32_t probe = 0; probe < n_buckets; ++probe) {
            const bucket k = read(bk);
            const cf_u32x4 c = *(const cf_u32x4*)&cnt32[4 * bk];   // the 8 counts of the bucket
            if ((k.lo.x & 0xFFFFFFu) == b && k.lo.x != kEmpty) total += c.x & 0x7FFFu;
            if ((k.lo.y & 0xFFFFFFu) == b && k.lo.y != kEmpty) total += (c.x >> 16) & 0x7FFFu;
            if ((k.lo.z & 0xFFFFFFu) == b && k.lo.z != kEmpty) total += c.y & 0x7FFFu;
            if ((k.lo.w & 0xFFFFFFu) == b && k.lo.w != kEmpty) total += (c.y >> 16) & 0x7FFFu;
            if ((k.hi.x & 0xFFFFFFu) == b && k.hi.x != kEmpty) total += c.z & 0x7FFFu;
            if ((k.hi.y & 0xFFFFFFu) == b && k.hi.y != kEmpty) total += (c.z >> 16) & 0x7FFFu;
            if ((k.hi.z & 0xFFFFFFu) == b && k.hi.z != kEmpty) total += c.w & 0x7FFFu;
            if ((k.hi.w & 0xFFFFFFu) == b && k.hi.w != kEmpty) total += (c.w >> 16) & 0x7FFFu;
            if (empty(k) >= 0) break;
            bk = bk + 1 == n_buckets ? 0u : bk + 1;
        }
        return total;
    }
    __device__ __forceinline__ void mark(uint32_t s) const { atomicOr(&cnt32[s >> 1], 0x8000u << ((s & 1u) * 16u)); }
    __device__ __forceinline__ bool marked(uint32_t s) const { return (cnt32[s >> 1] >> ((s & 1u) * 16u + 15u)) & 1u; }
};

// general insert: walk buckets from bk; claims the first empty slot with a CAS when the key is absent.
// Returns 1 when a new key was created.
template <class Tab>
__device__ __forceinline__ uint32_t cf_dist_insert(const Tab& T, uint32_t n_buckets, uint32_t bk, uint32_t b, uint32_t dd, uint32_t* sh) {
    for (uint32_t tries = 0; tries < 9 * n_buckets; ++tries) {
        const typename Tab::bucket k = T.read(bk);
        const int m = Tab::match(k, b, dd);
        if (m >= 0) { T.add(bk, m); return 0u; }
        const int e = Tab::empty(k);
        if (e >= 0) {
            const int st = T.claim_finish(T.claim_issue(bk, e, b, dd), bk, e, b, dd);
            if (st == 0) return 1u;
            if (st == 1) return 0u;
            continue;   // another key took the slot: look at the same bucket again
        }
        bk = bk + 1 == n_buckets ? 0u : bk + 1;
    }
    sh[1] = 1;   // table physically full: the pass is void and will be split
    return 0u;
}

// Partner ranges of the postings [c0, c0 + np) of one first k-mer -> LDS (pE0, pig) and the exclusive prefix of
// their lengths (pre[0..np]).  Called by all threads of the workgroup.
__device__ __forceinline__ void cf_dist_setup(const cf_dist_args& A, int64_t c0, int np, int64_t* pE0, int32_t* pig, uint32_t* pre) {
    const int t = threadIdx.x, nt = blockDim.x;
    for (int p = t; p < np; p += nt) {
        const int32_t g = A.post[c0 + p];
        const int32_t jlo = g + A.min_d;
        const int32_t jhi = min(A.unit_rend[g] - 1, g + A.max_d);
        int64_t E0 = 0, E1 = 0;
        if (jhi >= jlo) { E0 = A.cloud_ptr[jlo]; E1 = A.cloud_ptr[jhi + 1]; }
        pE0[p] = E0; pig[p] = g - A.unit_rbeg[g];
        pre[p + 1] = (uint32_t)(E1 - E0);
    }
    __syncthreads();
    if (t < 64) {   // wave 0: inclusive scan of np <= DIST_NP_CAP (= 4 x 64) lengths, 4 per lane
        uint32_t v[DIST_NP_CAP / 64], sum = 0;
#pragma unroll
        for (int i = 0; i < DIST_NP_CAP / 64; ++i) { const int p = t * (DIST_NP_CAP / 64) + i; v[i] = p < np ? pre[p + 1] : 0u; sum += v[i]; v[i] = sum; }
        uint32_t inc = sum;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, (unsigned)d); if (t >= d) inc += o; }
        const uint32_t base = inc - sum;
#pragma unroll
        for (int i = 0; i < DIST_NP_CAP / 64; ++i) { const int p = t * (DIST_NP_CAP / 64) + i; if (p < np) pre[p + 1] = base + v[i]; }
        if (t == 0) pre[0] = 0;
    }
    __syncthreads();
}

// Diagnostic build only (-DCF_DIST_STAMPS, tools/dist_stamps.py): per-phase shader-clock sums of thread 0 of every
// workgroup into counters[8..15]; the shipped library compiles these to nothing.
#if defined(CF_DIST_STAMPS)
#define CF_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)
#else
#define CF_STAMP(i) do { } while (0)
#endif

template <class Tab>
__global__ void cf_dist_kernel(cf_dist_args A) {
    Tab T;
    T.init(cf_lds, (uint32_t)A.slots);
    int64_t* pE0 = (int64_t*)(cf_lds + (size_t)A.slots * Tab::kSlotBytes);  // first partner entry of each posting
    int32_t* pig = (int32_t*)(pE0 + DIST_NP_CAP);              // unit index of the posting inside its read
    uint32_t* pre = (uint32_t*)(pig + DIST_NP_CAP);            // prefix of partner-entry counts (NP_CAP + 1)
    uint32_t* stack = pre + DIST_NP_CAP + 1;                   // (P, idx) pairs
    uint32_t* sh = stack + 2 * DIST_STACK;                     // [0] keys in table [1] overflow [2] sp [3] P [4] idx [5,6] queue ticket [7] E of pass [8] selected [9,10] edge base [11] stream cursor
    uint16_t* stage = (uint16_t*)(sh + 16);                    // slot indices of the selected edges of a pass
    const int t = threadIdx.x, lane = t & 63, nt = blockDim.x;
    const uint32_t slots = (uint32_t)A.slots, n_buckets = slots / Tab::kPerBucket;   // slots is a multiple of 8
    unsigned long long acc_E = 0, acc_spill = 0, acc_pass = 0;  // flushed once per workgroup (thread 0)
#if defined(CF_DIST_STAMPS)
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t = __builtin_amdgcn_s_memtime();
#endif

    while (true) {
        __syncthreads();
        if (t == 0) {
            // 8 ticket queues over 8 contiguous ranges of the locality-sorted first k-mers: workgroups with
            // equal blockIdx % 8 (observed to share an XCD, i.e. an L2) drain one range; idle ones steal
            long long idx = -1;
            const int64_t per = (A.n_order + 7) / 8;
            for (int s8 = 0; s8 < 8 && idx < 0; ++s8) {
                const int x = (int)((blockIdx.x + s8) & 7);
                const int64_t lo = x * per, hi = min((int64_t)(x + 1) * per, A.n_order);
                if (lo >= hi) continue;
                const unsigned long long q = atomicAdd(&A.counters[16 + 16 * x], 1ull);
                if (lo + (int64_t)q < hi) idx = lo + (int64_t)q;
            }
            sh[5] = (uint32_t)(unsigned long long)idx; sh[6] = (uint32_t)((unsigned long long)idx >> 32);
        }
        __syncthreads();
        const int64_t ai = (int64_t)(((unsigned long long)sh[6] << 32) | sh[5]);
        if (ai < 0) break;
        CF_STAMP(0);   // queue pop
        const uint32_t a = (uint32_t)A.order[ai];
        const int64_t pp0 = A.post_ptr[a], pp1 = A.post_ptr[a + 1];
        if (pp1 == pp0) continue;
        // upper bound of the emissions of a -> initial number of partitions of its (b, d) table.
        // Usual case (<= DIST_NP_CAP postings): the partner ranges are set up ONCE and reused by every pass.
        const bool one_chunk = (pp1 - pp0) <= DIST_NP_CAP;
        if (t == 0) sh[7] = 0;
        __syncthreads();
        if (one_chunk) {
            cf_dist_setup(A, pp0, (int)(pp1 - pp0), pE0, pig, pre);
            if (t == 0) sh[7] = pre[pp1 - pp0];
        } else {
            unsigned long long em = 0;
            for (int64_t p = pp0 + t; p < pp1; p += nt) {
                const int32_t g = A.post[p];
                const int32_t jlo = g + A.min_d, jhi = min(A.unit_rend[g] - 1, g + A.max_d);
                if (jhi >= jlo) em += (unsigned long long)(A.cloud_ptr[jhi + 1] - A.cloud_ptr[jlo]);
            }
            for (int d = 32; d >= 1; d >>= 1) em += __shfl_down(em, (unsigned)d);
            if (lane == 0 && em) atomicAdd(&sh[7], (uint32_t)min(em, 0x3FFFFFFFull));
        }
        __syncthreads();
        if (t == 0) {
            uint32_t P0 = 1;
            while (P0 < 64u && (unsigned long long)sh[7] * 4ull > (unsigned long long)A.fill_limit * 5ull * P0) P0 <<= 1;
            for (uint32_t i = 0; i < P0; ++i) { stack[2 * i] = P0; stack[2 * i + 1] = i; }
            sh[2] = P0;
        }
        CF_STAMP(1);   // prologue: posting ranges, estimate
        bool spilled = false;
        while (true) {
            CF_STAMP(5);   // reserve + write edges of the previous pass
            __syncthreads();
            const uint32_t sp_now = sh[2];
            __syncthreads();  // everyone has read the stack pointer before thread 0 pops
            if (sp_now == 0) break;
            if (t == 0) { const uint32_t sp = sh[2] - 1; sh[2] = sp; sh[3] = stack[2 * sp]; sh[4] = stack[2 * sp + 1]; sh[0] = 0; sh[1] = 0; sh[7] = 0; sh[8] = 0; }
            T.clear(slots, (uint32_t)t, (uint32_t)nt);
            __syncthreads();
            const uint32_t P = sh[3], pidx = sh[4];
            uint32_t my_e = 0;
            CF_STAMP(2);   // pop partition + clear table
            // ---- stream the partner clouds of every posting of a, in chunks of DIST_NP_CAP postings.
            // The units g+min_d .. min(read end, g+max_d) of a posting are ONE contiguous range of the
            // CSR; the ranges of all postings are concatenated into a flat index space that the waves
            // sweep with coalesced loads (entry rank + the entry's unit index inside its read).
            for (int64_t c0 = pp0; c0 < pp1; c0 += DIST_NP_CAP) {
                const int np = (int)min((int64_t)DIST_NP_CAP, pp1 - c0);
                if (!one_chunk) cf_dist_setup(A, c0, np, pE0, pig, pre);
                if (t == 0) sh[11] = 0;   // shared cursor over the flat entry range of this chunk
                __syncthreads();
                const uint32_t total = pre[np];
                int p_cur = 0;
                uint32_t r_lo = 0, r_hi = pre[1];          // flat range of posting p_cur, cached in registers
                int64_t r_e0 = pE0[0];
                int32_t r_ig = pig[0];
                // software pipeline: the global loads of step i+1 are issued before the LDS work of step i
                uint32_t nb_[DIST_UNROLL], nd_[DIST_UNROLL];
#define CF_DIST_FETCH(F0)                                                                                     \
                _Pragma("unroll") for (int u = 0; u < DIST_UNROLL; ++u) {                                     \
                    const uint32_t f = (F0) + (uint32_t)u * 64u + (uint32_t)lane;                             \
                    nb_[u] = a; nd_[u] = 0;              /* a itself is never counted: skip marker */         \
                    if (f < total) {                                                                          \
                        if (f >= r_hi) {                 /* monotone: f only grows */                         \
                            while (pre[p_cur + 1] <= f) ++p_cur;                                              \
                            r_lo = pre[p_cur]; r_hi = pre[p_cur + 1]; r_e0 = pE0[p_cur]; r_ig = pig[p_cur];   \
                        }                                                                                     \
                        const int64_t e = r_e0 + (int64_t)(f - r_lo);                                         \
                        nb_[u] = (uint32_t)A.entries[e];                                                      \
                        nd_[u] = (uint32_t)((int32_t)A.entry_i[e] - r_ig);                                    \
                    }                                                                                         \
                }
                // waves pull 64 x DIST_UNROLL consecutive flat entries at a time from a shared cursor: the cost of
                // an entry varies (new key, full bucket), a static split leaves waves idle at the closing barrier
#define CF_DIST_GRAB(VAR) { uint32_t g_ = 0; if (lane == 0) g_ = atomicAdd(&sh[11], 64u * DIST_UNROLL); VAR = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_); }
                uint32_t f0, f1;
                CF_DIST_GRAB(f0)
                if (f0 < total) { CF_DIST_FETCH(f0) }
                while (f0 < total) {
                    if (sh[1] || sh[0] > A.fill_limit) break;
                    uint32_t bb[DIST_UNROLL], dd_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) { bb[u] = nb_[u]; dd_[u] = nd_[u]; }
                    CF_DIST_GRAB(f1)
                    if (f1 < total) { CF_DIST_FETCH(f1) }
                    f0 = f1;
                    // bucket reads of all unrolled emissions first (independent LDS reads in flight), then resolve
                    uint32_t bk_[DIST_UNROLL], live = 0;
                    typename Tab::bucket kb_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) {
                        const uint32_t b = bb[u];
                        const uint32_t hb = cf_dist_hash(b);
                        if (b != a && (P == 1 || (((hb ^ (hb >> 15)) >> 3) & (P - 1)) == pidx)) live |= 1u << u;
                        bk_[u] = (uint32_t)(((unsigned long long)hb * (unsigned long long)n_buckets) >> 32);
                        kb_[u] = T.read(bk_[u]);
                    }
                    // (1) matches -> fire-and-forget adds, new keys -> first empty slot of the bucket already in
                    // registers; (2) all CASes of the step issued back to back; (3) leftovers (bucket full, slot
                    // lost to another key) take the general probe loop, one leftover per lane and round.
                    uint32_t fresh = 0, failm = 0;   // bit u: created a new key / needs the general path
                    int cand_[DIST_UNROLL];          // slot to claim, or -1
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) {
                        cand_[u] = -1;
                        if (!((live >> u) & 1u)) continue;
                        ++my_e;
                        const int m = Tab::match(kb_[u], bb[u], dd_[u]);
                        if (m >= 0) { T.add(bk_[u], m); continue; }              // common case: the pair was seen before
                        const int e = Tab::empty(kb_[u]);
                        if (e >= 0) cand_[u] = e;
                        else { failm |= 1u << u; bk_[u] = bk_[u] + 1 == n_buckets ? 0u : bk_[u] + 1; }
                    }
                    decltype(T.claim_issue(0u, 0, 0u, 0u)) old_[DIST_UNROLL];
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) {
                        old_[u] = 0;
                        if (cand_[u] >= 0) old_[u] = T.claim_issue(bk_[u], cand_[u], bb[u], dd_[u]);
                    }
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) {
                        if (cand_[u] < 0) continue;
                        const int st = T.claim_finish(old_[u], bk_[u], cand_[u], bb[u], dd_[u]);
                        if (st == 0) fresh |= 1u << u;
                        else if (st == 2) failm |= 1u << u;
                    }
                    while (__any(failm != 0u)) {
                        if (failm != 0u) {
                            const int u = __ffs((int)failm) - 1;
                            failm &= failm - 1u;
                            uint32_t xb = bb[0], xd = dd_[0], xk = bk_[0];
#pragma unroll
                            for (int v = 1; v < DIST_UNROLL; ++v) if (u == v) { xb = bb[v]; xd = dd_[v]; xk = bk_[v]; }
                            fresh |= cf_dist_insert(T, n_buckets, xk, xb, xd, sh) << u;
                        }
                    }
                    // fill level: one fire-and-forget LDS atomic per wave and step; read back at the next step
                    uint32_t wave_new = 0;
#pragma unroll
                    for (int u = 0; u < DIST_UNROLL; ++u) wave_new += (uint32_t)__popcll(__ballot((fresh >> u) & 1u));
                    if (wave_new && lane == 0) atomicAdd(&sh[0], wave_new);
                }
                __syncthreads();
            }
            CF_STAMP(3);   // stream + insert
            if (sh[1] || sh[0] > A.fill_limit) {  // overflow: split this partition in two
                if (t == 0) {
                    uint32_t sp = sh[2];
                    if (P >= (1u << 20) || sp + 2 > DIST_STACK) { atomicOr(&A.counters[4], 1ull); }
                    else { stack[2 * sp] = 2 * P; stack[2 * sp + 1] = pidx; stack[2 * sp + 2] = 2 * P; stack[2 * sp + 3] = pidx + P; sh[2] = sp + 2; }
                }
                spilled = true;
                continue;
            }
            // ---- table of the pass complete: count emissions, filter in LDS (selected slots are marked and
            // staged), reserve the edge range with ONE global atomic, then write
            for (int d = 32; d >= 1; d >>= 1) my_e += __shfl_down(my_e, (unsigned)d);
            if (lane == 0 && my_e) atomicAdd(&sh[7], my_e);
            const uint32_t rounds = (slots + nt - 1) / nt;
            for (uint32_t rd = 0; rd < rounds; ++rd) {
                const uint32_t s = rd * nt + t;
                uint32_t b, dd, cnt;
                if (s < slots && T.get(s, b, dd, cnt) && cnt >= A.min_cov) {
                    const unsigned long long total = T.total_of(b, n_buckets);
                    if (((double)cnt / (double)total) >= A.thr) {
                        T.mark(s);
                        const uint32_t pos = atomicAdd(&sh[8], 1u);
                        if (pos < A.stage_cap) stage[pos] = (uint16_t)s;
                    }
                }
            }
            __syncthreads();
            CF_STAMP(4);   // filter
            const uint32_t n_sel = sh[8];
            __syncthreads();  // everyone has read the count before thread 0 reuses the word as a cursor
            if (t == 0) {
                acc_E += sh[7]; ++acc_pass;
                if (n_sel) {
                    const unsigned long long base = atomicAdd(&A.counters[0], (unsigned long long)n_sel);
                    sh[9] = (uint32_t)base; sh[10] = (uint32_t)(base >> 32);
                    const uint32_t bit = 1u << (a & 31);
                    if (!(A.unique_bits[a >> 5] & bit)) atomicOr(&A.unique_bits[a >> 5], bit);
                }
                sh[8] = 0;
            }
            __syncthreads();
            if (n_sel) {
                const unsigned long long base = ((unsigned long long)sh[10] << 32) | sh[9];
                const bool staged = n_sel <= A.stage_cap;   // else: sweep the table for the marked slots
                const uint32_t n_iter = staged ? n_sel : rounds * (uint32_t)nt;
                for (uint32_t i0 = 0; i0 < n_iter; i0 += nt) {
                    const uint32_t i = i0 + t;
                    uint32_t b = 0, dd = 0, cnt = 0, s = 0;
                    bool sel = false;
                    if (staged) { if (i < n_sel) { s = stage[i]; sel = T.get(s, b, dd, cnt); } }
                    else if (i < slots) { s = i; sel = T.marked(s) && T.get(s, b, dd, cnt); }
                    unsigned long long o = base + i;
                    if (!staged) {   // order of the sweep: wave-aggregated cursor
                        const unsigned long long m = __ballot(sel);
                        uint32_t off = 0;
                        if (m) { const int leader = __ffsll((long long)m) - 1; if (lane == leader) off = atomicAdd(&sh[8], (uint32_t)__popcll(m)); off = __shfl(off, leader); }
                        o = base + off + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                    }
                    if (sel) {
                        if (o < A.edge_cap) { uint32_t* E = A.edges + 4 * o; E[0] = dd; E[1] = a; E[2] = b; E[3] = cnt; }
                        const uint32_t bit = 1u << (b & 31);
                        if (!(A.unique_bits[b >> 5] & bit)) atomicOr(&A.unique_bits[b >> 5], bit);
                    }
                }
            }
        }
        if (spilled && t == 0) ++acc_spill;
    }
    if (t == 0) {
        if (acc_E) atomicAdd(&A.counters[1], acc_E);
        if (acc_spill) atomicAdd(&A.counters[2], acc_spill);
        if (acc_pass) atomicAdd(&A.counters[5], acc_pass);
#if defined(CF_DIST_STAMPS)
        for (int i = 0; i < 8; ++i) atomicAdd(&A.counters[8 + i], stamp_acc[i]);
#endif
    }
}

__global__ void __launch_bounds__(256)
cf_max_u32_kernel(const uint32_t* __restrict__ v, int64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = max(m, v[i]);
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_down(m, (unsigned)d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

extern "C" {

int cf_dist_edges(cf_ctx* ctx, int64_t min_n, int64_t max_n, int32_t min_d, int32_t max_d, uint32_t min_cov,
                  double rel_threshold, int32_t part, int32_t n_parts, int64_t edge_cap, int64_t* n_edges) {
    if (!ctx) return -22;
    if (!ctx->have_clouds) return cf_fail(ctx, -22, "cf_dist_edges: no clouds built");
    if (n_parts < 1 || part < 0 || part >= n_parts) return cf_fail(ctx, -22, "cf_dist_edges: bad partition");
    if (max_d > 255) return cf_fail(ctx, -22, "cf_dist_edges: max_d > 255 does not fit the 8-bit distance field");
    if (edge_cap < 0) edge_cap = 0;
    for (int64_t r = 0; r < ctx->n_reads; ++r)
        if (ctx->h_unit_ptr[(size_t)r + 1] - ctx->h_unit_ptr[(size_t)r] > 65535)
            return cf_fail(ctx, -22, "cf_dist_edges: a read has more than 65535 units (16-bit unit index per cloud entry)");
    CF_HIP(hipSetDevice(ctx->device));
    CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    const int64_t R = ctx->n_reads, U = ctx->n_units, K = ctx->n_kmers;
    // Python slice semantics of itertools.islice(items, min_n, max_n) for non-negative bounds
    if (min_n < 0) min_n = 0;
    if (max_n > R) max_n = R;
    if (max_n < min_n) max_n = min_n;
    if (min_n > R) min_n = R;
    const int64_t u0 = ctx->h_unit_ptr[(size_t)min_n], u1 = ctx->h_unit_ptr[(size_t)max_n];
    const int32_t min_d_eff = min_d < 1 ? 1 : min_d;  // kmer_clouds[:-0] is empty: d = 0 emits nothing

    uint32_t *d_pcnt = nullptr, *d_cursor = nullptr, *d_first = nullptr;
    unsigned long long *d_okeys = nullptr, *d_otmp = nullptr;
    int32_t* d_order = nullptr;
    int64_t n_order = 0, n_a_alloc = 0;
    int64_t* d_post_ptr = nullptr;
    int32_t *d_post = nullptr, *d_rend = nullptr, *d_rbeg = nullptr;
    uint16_t* d_entry_i = nullptr;
    unsigned long long* d_cnt = nullptr;
    int64_t n_post = 0;
    int rc = 0;
    unsigned long long h_cnt[8] = {0};
    const size_t n_cnt = 8 + 16 * 9;   // counters + 8 queue heads on their own cache lines
    const int max_blocks = std::max(1, ctx->n_cu) * 8;
    do {
        cf_free_edges(ctx);
        if ((rc = cf_alloc_t(ctx, &ctx->d_edges, (size_t)edge_cap * 4, "edges"))) break;
        ctx->edge_cap = edge_cap;
        if ((rc = cf_alloc_t(ctx, &d_pcnt, (size_t)K + 1, "posting counts"))) break;
        if ((rc = cf_alloc_t(ctx, &d_cursor, (size_t)K + 1, "posting cursors"))) break;
        if ((rc = cf_alloc_t(ctx, &d_post_ptr, (size_t)K + 1, "posting offsets"))) break;
        if ((rc = cf_alloc_t(ctx, &d_rend, (size_t)U + 1, "unit read ends"))) break;
        if ((rc = cf_alloc_t(ctx, &d_rbeg, (size_t)U + 1, "unit read begins"))) break;
        if ((rc = cf_alloc_t(ctx, &d_entry_i, (size_t)ctx->n_entries + 1, "entry unit indices"))) break;
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
            if (hipMemcpy(&tmp[0], ctx->d_cloud_ptr + u0, 8, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(&tmp[1], ctx->d_cloud_ptr + u1, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "cloud_ptr read"); break; }
            e0 = tmp[0]; e1 = tmp[1];
        }
        if (e1 > e0)
            hipLaunchKernelGGL(cf_post_hist_kernel, dim3((unsigned)cf_grid_for(e1 - e0, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const int32_t*)ctx->d_entries, e0, e1, d_pcnt);
        if ((rc = cf_scan_exclusive_u32_to_i64(ctx, d_pcnt, d_post_ptr, K + 1, &n_post))) break;
        if ((rc = cf_alloc_t(ctx, &d_post, (size_t)n_post, "postings"))) break;
        if (u1 > u0 && n_post)
            hipLaunchKernelGGL(cf_post_fill_kernel, dim3((unsigned)cf_grid_for((u1 - u0) * 64, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const int64_t*)ctx->d_cloud_ptr, (const int32_t*)ctx->d_entries, u0, u1, (const int64_t*)d_post_ptr, d_cursor, d_post, d_first);
        if (R)
            hipLaunchKernelGGL(cf_unit_rend_kernel, dim3((unsigned)cf_grid_for(R, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const int64_t*)ctx->d_unit_ptr, R, d_rend, d_rbeg);
        if (U && ctx->n_entries)
            hipLaunchKernelGGL(cf_entry_unit_kernel, dim3((unsigned)cf_grid_for(U * 64, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const int64_t*)ctx->d_cloud_ptr, (const int32_t*)d_rbeg, U, d_entry_i);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ctx->ev2, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("postings: ") + hipGetErrorString(e)); break; }

        cf_dist_args A;
        A.post_ptr = d_post_ptr; A.post = d_post; A.cloud_ptr = ctx->d_cloud_ptr; A.entries = ctx->d_entries; A.unit_rend = d_rend; A.unit_rbeg = d_rbeg; A.entry_i = d_entry_i;
        A.n_kmers = K; A.part = part; A.n_parts = n_parts; A.min_d = min_d_eff; A.max_d = max_d; A.min_cov = min_cov; A.thr = rel_threshold;
        A.stage_cap = (uint32_t)std::min(ctx->dist_stage, DIST_STAGE_CAP);
        // table layout: 6-byte slots (32-bit keys, 16-bit counts) whenever ranks fit 24 bits and counts 15 bits
        uint32_t max_post = 0;
        if (K) {
            hipLaunchKernelGGL(cf_max_u32_kernel, dim3((unsigned)cf_grid_for(K, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const uint32_t*)d_pcnt, K, (uint32_t*)(d_cnt + 7));
            if (hipMemcpy(&max_post, d_cnt + 7, 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "max postings"); break; }
        }
        const bool narrow = !ctx->dist_wide && K < ((int64_t)1 << 24) - 1 && max_post <= 32767u;
        if (max_post >= (1u << 23)) { rc = cf_fail(ctx, -34, "cf_dist_edges: a k-mer has more than 2^23 postings"); break; }
        const uint32_t slot_bytes = narrow ? cf_tab_narrow::kSlotBytes : cf_tab_wide::kSlotBytes;
        A.slots = (int32_t)(((int64_t)ctx->dist_slots * 8 / slot_bytes) & ~7ll);   // dist_slots is the LDS budget in 8-byte slots
        A.fill_limit = (uint32_t)((int64_t)A.slots * ctx->dist_fill_pct / 100);   // checked once per wave step: leave slack below the physical size
        A.edges = ctx->d_edges; A.edge_cap = (unsigned long long)edge_cap; A.counters = d_cnt; A.unique_bits = ctx->d_unique_bits;
        const size_t lds = (size_t)A.slots * slot_bytes + (size_t)(4 * DIST_NP_CAP + 1 + 2 * DIST_STACK + 16) * 4 + DIST_STAGE_CAP * 2 + 16;
        if (lds > 160 * 1024) { rc = cf_fail(ctx, -22, "cf_dist_edges: LDS request exceeds 160 KiB"); break; }
        const int per_cu = std::max(1, std::min((int)((160 * 1024) / lds), 2048 / ctx->dist_block));
        // locality order of the first k-mers: sort (first posting unit, a); k-mers without postings drop out
        n_a_alloc = (K > part) ? (K - part + n_parts - 1) / n_parts : 0;
        if ((rc = cf_alloc_t(ctx, &d_okeys, (size_t)n_a_alloc, "order keys"))) break;
        if ((rc = cf_alloc_t(ctx, &d_otmp, (size_t)n_a_alloc, "order scratch"))) break;
        if ((rc = cf_alloc_t(ctx, &d_order, (size_t)n_a_alloc, "order"))) break;
        if (K) {
            hipLaunchKernelGGL(cf_order_keys_kernel, dim3((unsigned)cf_grid_for(K, 256, max_blocks)), dim3(256), 0, ctx->stream,
                               (const uint32_t*)d_pcnt, (const uint32_t*)d_first, K, (int)part, (int)n_parts, d_okeys, d_cnt + 6);
            unsigned long long h_n = 0;
            if (hipMemcpy(&h_n, d_cnt + 6, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = cf_fail(ctx, -5, "order count"); break; }
            n_order = (int64_t)h_n;
            if ((rc = cf_radix_sort_u64(ctx, d_okeys, d_otmp, n_order, 63))) break;
            if (n_order)
                hipLaunchKernelGGL(cf_order_extract_kernel, dim3((unsigned)cf_grid_for(n_order, 256, max_blocks)), dim3(256), 0, ctx->stream,
                                   (const unsigned long long*)d_okeys, n_order, d_order);
        }
        A.order = d_order; A.n_order = n_order;
        const int64_t n_a = n_order;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_a, (int64_t)std::max(1, ctx->n_cu) * per_cu));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ctx->ev2, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("order: ") + hipGetErrorString(e)); break; }
        e = narrow ? hipFuncSetAttribute((const void*)cf_dist_kernel<cf_tab_narrow>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                   : hipFuncSetAttribute((const void*)cf_dist_kernel<cf_tab_wide>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("dist LDS attribute: ") + hipGetErrorString(e)); break; }
        if (n_a > 0 && max_d >= min_d_eff && n_post > 0) {
            if (narrow) hipLaunchKernelGGL((cf_dist_kernel<cf_tab_narrow>), dim3((unsigned)grid), dim3((unsigned)ctx->dist_block), lds, ctx->stream, A);
            else hipLaunchKernelGGL((cf_dist_kernel<cf_tab_wide>), dim3((unsigned)grid), dim3((unsigned)ctx->dist_block), lds, ctx->stream, A);
            e = hipGetLastError();
            if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_dist_kernel: ") + hipGetErrorString(e)); break; }
        }
        e = hipEventRecord(ctx->ev3, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt, 64, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_dist_edges: ") + hipGetErrorString(e)); break; }
#if defined(CF_DIST_STAMPS)
        {
            unsigned long long st[8];
            if (hipMemcpy(st, d_cnt + 8, 64, hipMemcpyDeviceToHost) == hipSuccess)
                std::fprintf(stderr, "[cf_dist stamps] pop=%llu prologue=%llu clear=%llu stream=%llu filter=%llu write=%llu (shader cycles summed over %d workgroups; passes=%llu)\n",
                             st[0], st[1], st[2], st[3], st[4], st[5], grid, h_cnt[5]);
        }
#endif
        if (h_cnt[4]) { rc = cf_fail(ctx, -34, "cf_dist_edges: (b,d) table could not be partitioned far enough"); break; }
        (void)hipEventElapsedTime(&ctx->times.dist_ms, ctx->ev0, ctx->ev1);
        (void)hipEventElapsedTime(&ctx->times.postings_ms, ctx->ev0, ctx->ev2);
        (void)hipEventElapsedTime(&ctx->times.dist_kernel_ms, ctx->ev2, ctx->ev3);
    } while (0);
    if (d_order) cf_release_t(ctx, d_order, (size_t)n_a_alloc);
    if (d_otmp) cf_release_t(ctx, d_otmp, (size_t)n_a_alloc);
    if (d_okeys) cf_release_t(ctx, d_okeys, (size_t)n_a_alloc);
    if (d_first) cf_release_t(ctx, d_first, (size_t)K + 1);
    if (d_cnt) cf_release_t(ctx, d_cnt, n_cnt);
    if (d_entry_i) cf_release_t(ctx, d_entry_i, (size_t)ctx->n_entries + 1);
    if (d_rbeg) cf_release_t(ctx, d_rbeg, (size_t)U + 1);
    if (d_rend) cf_release_t(ctx, d_rend, (size_t)U + 1);
    if (d_post) cf_release_t(ctx, d_post, (size_t)n_post);
    if (d_post_ptr) cf_release_t(ctx, d_post_ptr, (size_t)K + 1);
    if (d_cursor) cf_release_t(ctx, d_cursor, (size_t)K + 1);
    if (d_pcnt) cf_release_t(ctx, d_pcnt, (size_t)K + 1);
    if (rc) return rc;
    ctx->stats.n_edges = (int64_t)h_cnt[0];
    ctx->stats.n_emissions = (int64_t)h_cnt[1];
    ctx->stats.n_spilled = (int64_t)h_cnt[2];
    ctx->stats.n_dist_passes = (int64_t)h_cnt[5];
    ctx->n_edges_stored = std::min<int64_t>((int64_t)h_cnt[0], edge_cap);
    CF_TRY(cf_refresh_unique_count(ctx));
    if (n_edges) *n_edges = (int64_t)h_cnt[0];
    return 0;
}

}  // extern "C"
