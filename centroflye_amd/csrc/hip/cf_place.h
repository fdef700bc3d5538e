// cf_place.h — what both placement paths (cf_place.hip: hash map + seen set, rounds 1-3; cf_place2.hip: per-read row
// regions, round 4) share: the contig map, the candidate key and its reductions.
#pragma once
#include "cf_common.h"

#define PL_THREADS 256

struct cf_cand {
    uint32_t s0, s1, off, rank;  // rank: smaller id wins
    uint32_t read;
    uint32_t valid;
};

__device__ __forceinline__ bool cf_cand_better(const cf_cand& a, const cf_cand& b) {
    // is a strictly better than b?
    if (!a.valid) return false;
    if (!b.valid) return true;
    if (a.s0 != b.s0) return a.s0 > b.s0;
    if (a.s1 != b.s1) return a.s1 > b.s1;
    if (a.off != b.off) return a.off > b.off;
    return a.rank < b.rank;
}

struct cf_place_state {
    // clouds
    const int64_t* unit_ptr;
    const int64_t* cloud_ptr;
    const int32_t* entries;
    const int32_t* unit2read;
    const int32_t* entry_unit;      // unit of every cloud entry (the fused iteration kernel walks a read's entries, not its units)
    // contig map
    unsigned long long* ckeys; uint32_t* ccnt; uint64_t cmask;
    uint8_t* freq_flag;
    // postings of the stage
    const int64_t* post_ptr; const unsigned long long* post_ri;   // posting = read << 32 | unit index inside the read
    // score map: key (read<<32|off)|OCC, value s0 << 32 | s1 in ONE word: the lane whose add is the last one on an entry gets
    // the entry's final state back from that add
    unsigned long long* skeys; unsigned long long* s01; uint64_t smask;
    uint32_t* qflag;   // one byte per score slot (4 per word): some view of the entry qualified; the arg-max scans these bytes only
    // seen set of (score slot << 32 | unit index)
    unsigned long long* seen; uint64_t seen_mask;
    // events (kmer << 32 | pos)
    unsigned long long* events; unsigned long long* n_events;  // n_events[0] = count
    // control: [0] done, [1] n_out, [2] error flags (1: the contig's map is full, 2: a score region / the score map, 4: ...), [3] thr, [4] score-map entries, [5] contig-map entries, [6] cf_place2: rescans of reads with more than four candidate rows
    unsigned int* ctl;
    const uint8_t* used_in; uint8_t* used;
    const int32_t* id_rank;
    cf_cand* block_best; cf_cand* best;
    uint8_t* dirty;        // per arg-max block: an entry of its slice of the score map changed since the block's cached best was computed
    uint32_t slice_shift;  // log2(score slots per arg-max block)
    int64_t* out_read; int64_t* out_pos; int32_t* out_s0; int32_t* out_s1;
    uint32_t thr, min_unit, min_inters, min_prop;
};

// The contig's (k-mer, position) map is open-addressed.  Round 5 (tools/fuzz_place.py: 500 reads placed in 183 s): a map that had
// filled up was noticed only by an add that had probed EVERY slot — 131 072 compare-and-swaps per add, for up to 512 greedy iterations
// until the host looked at the flag.  Now the claims are counted ([5] of the control words; the map lives as long as the contig, so the
// count is never reset) and the flag goes up at half load, and no add probes more than CF_CONTIG_PROBES slots (at half load the longest
// run of an open-addressed table of 10^9 slots is about a hundred).
#define CF_CONTIG_PROBES 512

// ---- add one read at a position: thread-block grid over units of the read
__device__ __forceinline__ void cf_contig_add(const cf_place_state& S, uint32_t x, uint32_t q) {
    const unsigned long long want = (((unsigned long long)q << 32) | x) | CF_OCC;
    uint64_t h = cf_mix64(want) & S.cmask;
    for (uint64_t probe = 0; probe <= min(S.cmask, (uint64_t)CF_CONTIG_PROBES); ++probe) {
        const unsigned long long cur = atomicCAS(&S.ckeys[h], 0ull, want);      // (the claim IS the look: one round trip, not load + claim)
        if (cur == 0ull || cur == want) {
            if (cur == 0ull && atomicAdd(&S.ctl[5], 1u) > (unsigned int)(S.cmask >> 1)) atomicOr(&S.ctl[2], 1u);      // load > 0.5: the host starts over with a larger map
            const uint32_t c = atomicAdd(&S.ccnt[h], 1u) + 1u;
            if (c == S.thr) {
                S.freq_flag[x] = 1;
                const unsigned long long p = atomicAdd(S.n_events, 1ull);
                S.events[p] = ((unsigned long long)x << 32) | q;
            }
            return;
        }
        h = (h + 1) & S.cmask;
    }
    atomicOr(&S.ctl[2], 1u);
}

// The same add for the fused iteration kernel: returns true when (x, q) just became frequent instead of queueing an event.
__device__ __forceinline__ bool cf_contig_add_hit(const cf_place_state& S, uint32_t x, uint32_t q) {
    const unsigned long long want = (((unsigned long long)q << 32) | x) | CF_OCC;
    uint64_t h = cf_mix64(want) & S.cmask;
    for (uint64_t probe = 0; probe <= min(S.cmask, (uint64_t)CF_CONTIG_PROBES); ++probe) {
        const unsigned long long cur = atomicCAS(&S.ckeys[h], 0ull, want);
        if (cur == 0ull || cur == want) {
            if (cur == 0ull && atomicAdd(&S.ctl[5], 1u) > (unsigned int)(S.cmask >> 1)) atomicOr(&S.ctl[2], 1u);
            const uint32_t c = atomicAdd(&S.ccnt[h], 1u) + 1u;
            if (c == S.thr) { S.freq_flag[x] = 1; return true; }
            return false;
        }
        h = (h + 1) & S.cmask;
    }
    atomicOr(&S.ctl[2], 1u);
    return false;
}


// A candidate as ONE lexicographic key: hi = (s0 << 32 | s1) + 1 (0 = no candidate), lo = offset << 32 | ~rank — larger is
// better (read_placer.py:63-78: larger (s0, s1), then the larger offset, then the smaller read id), plus the read it names.
// Every comparison-and-replace below is written as branch-free selects on these three scalars.  Round 3: the round-2 form —
// `if (cf_cand_better(o, mine)) mine = o;` on the six-field struct inside the shuffle loop — was MISCOMPILED by hipcc 7.2 at
// -O3 for gfx950: after the d = 32 step the wave's lane kept the OLD `read` next to the new (s0, s1, offset, rank) whenever it
// took its partner's candidate there and none afterwards (the generated code parks (read, valid) of the not-taken side in a
// register pair and copies it back over the taken one; profiles/r03_place_miscompile.md has the ISA).  The 50 000-read
// placement test against the C placer caught it: a read was placed with another read's score.  tests/test_gpu_parity.py
// (cf_selftest_argmax) pins the reduction against the host on adversarial candidate sets.
struct cf_key { unsigned long long hi, lo; uint32_t read; };
__device__ __forceinline__ cf_key cf_key_of(const cf_cand& c) {
    cf_key k;
    k.hi = c.valid ? ((((unsigned long long)c.s0 << 32) | c.s1) + 1ull) : 0ull;
    k.lo = c.valid ? (((unsigned long long)c.off << 32) | (unsigned long long)(~c.rank)) : 0ull;
    k.read = c.valid ? c.read : 0u;
    return k;
}
__device__ __forceinline__ cf_cand cf_cand_of(const cf_key& k) {
    cf_cand c;
    c.valid = k.hi != 0ull ? 1u : 0u;
    c.s0 = c.valid ? (uint32_t)((k.hi - 1ull) >> 32) : 0u; c.s1 = c.valid ? (uint32_t)(k.hi - 1ull) : 0u;
    c.off = (uint32_t)(k.lo >> 32); c.rank = c.valid ? ~(uint32_t)k.lo : 0u; c.read = k.read;
    return c;
}
// mine := the better of (mine, o), by selects
__device__ __forceinline__ void cf_key_take(cf_key& mine, const cf_key& o) {
    const bool bt = o.hi > mine.hi || (o.hi == mine.hi && o.lo > mine.lo);
    mine.hi = bt ? o.hi : mine.hi;
    mine.lo = bt ? o.lo : mine.lo;
    mine.read = bt ? o.read : mine.read;
}

// best candidate of the workgroup (every thread gets it); uses the first 16 x 24 bytes of LDS
__device__ __forceinline__ cf_key cf_block_best_key(cf_key mine) {
    unsigned long long* shk = (unsigned long long*)cf_lds;      // per wave: hi, lo, read
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int d = 32; d >= 1; d >>= 1) {      // butterfly: all lanes end up with the wave's best
        cf_key o;
        o.hi = __shfl_xor(mine.hi, d); o.lo = __shfl_xor(mine.lo, d); o.read = (uint32_t)__shfl_xor((int)mine.read, d);
        cf_key_take(mine, o);
    }
    if (lane == 0) { shk[3 * wave] = mine.hi; shk[3 * wave + 1] = mine.lo; shk[3 * wave + 2] = (unsigned long long)mine.read; }
    __syncthreads();
    cf_key best{shk[0], shk[1], (uint32_t)shk[2]};
    for (int w = 1; w < nw; ++w) { const cf_key o{shk[3 * w], shk[3 * w + 1], (uint32_t)shk[3 * w + 2]}; cf_key_take(best, o); }
    __syncthreads();
    return best;
}
__device__ __forceinline__ cf_cand cf_block_best(const cf_key& mine) { return cf_cand_of(cf_block_best_key(mine)); }

namespace {
struct Bufs {
    cf_ctx* ctx;
    std::vector<std::pair<void*, size_t>> owned;
    template <class T> int get(T** p, size_t n, const char* what) {
        int rc = cf_alloc_t(ctx, p, n, what);
        if (rc == 0) owned.emplace_back((void*)*p, n * sizeof(T));
        return rc;
    }
    void release_all() { for (auto it = owned.rbegin(); it != owned.rend(); ++it) cf_release(ctx, it->first, it->second); owned.clear(); }
    ~Bufs() { release_all(); }
};
}  // namespace

// cf_place2.hip
bool cf_place2_fits(const cf_ctx* ctx);
int cf_place2_run(cf_ctx* ctx, const uint8_t* cls, const int32_t* id_rank, int32_t min_freq, int32_t min_unit, int32_t min_inters, int32_t min_prop,
                  std::vector<int64_t>& o_read, std::vector<int64_t>& o_pos, std::vector<int32_t>& o_s0, std::vector<int32_t>& o_s1);
