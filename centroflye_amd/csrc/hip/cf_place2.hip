// cf_place2.hip — A8 + A9 (cloud contig + greedy read placement), round 4: ONE kernel per greedy iteration.
//
// Reference (same semantics as cf_place.hip, which keeps the round 1-3 path as `place_mode` 1):
//   scripts/cloud_contig.py:26-41   add_read: count[(pos, k-mer)] += 1; the pair is reported when the count EQUALS the threshold
//   scripts/cloud_contig.py:87-95   update_mapping_scores: a reported (k-mer, q) adds 1 to scores[read][q - i][i] for every
//                                   posting (read, i) of the k-mer with q >= i
//   scripts/read_placer.py:42-94    seed (over-inclusive, :54-57), arg-max over (s0, s1, offset, smaller id) among entries
//                                   with s0 >= min_unit, s0 * min_prop <= s1, s1 >= min_inters (:63-78), None tail (:79-84)
//
// What rounds 1-3 measured: 24.6 us per greedy iteration, 25.0 of them INSIDE two kernels whose waves walk
// a chain of ten dependent HBM round trips (candidates, unit_ptr, cloud_ptr, entries, contig claim, contig count, posting
// range, postings, score-map claim, seen-set claim, score add) and then scan flag bytes of a 16 M-slot hash map.
// Round 4 prices the iteration in dependent steps (tools/ubench/chain.hip; the in-kernel per-wave timelines of
// -DCF_PL2_STAMPS2): a load 0.45 - 0.65 us, a device-scope atomic 1.1 - 1.2 us whether it returns or not, a dependent
// launch 1.6 - 3.3 us, "last workgroup done" 0.8 us; a wave's memory operations complete IN ORDER (one counter), so a
// load issued behind an atomic waits with it; and everything waits for the slowest lane.
//
//   iteration kernel (G workgroups; PW = 32 or 64 lanes per cloud entry of the read being laid down, one posting per lane):
//     1 the winner record {read, offset, first entry, entries}          (written by the previous launch's tail)
//     2 the entry {k-mer, unit index}                                    (one 8-byte array built per run)
//     3 the k-mer's CONTIG RECORD (its positions and counts, 32 bytes)  ||  its POSTING ROW (packed postings + count)
//       -> the event is decided from the record as loaded (no other lane touches the pair in this launch); the add is owed
//     4 the region record of the posting's read                          (only lanes of entries that raised an event)
//     5 the row's home bucket of 4 headers
//     6 ONE round of independent atomics: cell, count, claim (new rows), hot / dirty bits, the owed contig add;  drain
//   tail, by the workgroup that arrives last at a counter (one lane per dirty read, packed densely):
//     7 dirty-read bitmap  8 region records  9 hot bits + the anchor row  10 other hot rows, in lockstep over the wave
//     11 the RB records of the touched blocks of 64 reads || the L2 records of the others -> winner record for the next launch.
//   50 000 reads: 15.8 us per iteration (6.3 lay-down + 0.8 arrive + 7 tail + launch gap), 0.79 s in all (round 3: 1.25 - 1.35 s).
//
// Scores.  scores[read][offset][unit] is kept as the reference keeps it — one counter per (read, offset, unit) — in a
// REGION PER READ: H_r slots (a power of two >= place_slots_per_unit x units) in buckets of 4, slot = one 64-bit header
// [offset + 1 : 32 | hits taken through the bucket path : 32] + a row of 16-bit cells, one per unit.  s1 is the sum of
// the cells, s0 the number of non-zero ones: nothing else is stored.  A read's rows are enumerable (the round 1-3 hash map
// was not: it needed flag bytes and slice scans for the arg-max).
// tools/place_stats.py on the bench's 50 000 reads: 87 % of all hits land on the ONE row per read that finally wins, the
// rest is ~125 rows per read with 1.3 hits each, scattered over all offsets.  So:
//   * the row that the last rescan of a read found strongest is its ANCHOR {offset, slot}: a hit on the anchor offset
//     goes to the known slot without looking (a fire-and-forget cell add);
//   * any other hit looks at the offset's home bucket, then claims / finds its row and adds count and cell in one round
//     (taken back if the claim loses: counters are only READ after every workgroup has arrived, so a transient +1 is
//     invisible);
//   * a row is "hot" from max(1, min_inters) hits on: its bit in a per-slot bitmap is set by the adder that sees the count
//     there (a transient excess can only set it early); only hot rows can qualify, so the rescan of a read looks at its hot
//     rows only (1 - 2 of ~125), and only hits on hot rows mark the read dirty.
// Contig.  clouds[pos][k-mer] BY K-MER: four [position + 1 | count] words per k-mer (a genomic k-mer sits at one or two
// positions; a fifth position continues in the (position, k-mer) hash map of rounds 1-3).
// Arg-max.  RB[read] = best qualifying row of the read, L2[block of 64 reads] = best of the block; the tail rescans the
// dirty reads, re-reduces their blocks, reduces all blocks.  (s0 * min_prop <= s1 is not monotone; nothing here assumes
// it is: every dirty read is recomputed from its counters.)
#include "cf_place.h"

#include <cstdlib>

#define PL2_B 1024          // threads per workgroup of the iteration kernel (the tail wants 16 waves)
#define PL2_LIST 2048       // dirty blocks of 8 reads per round of the tail
#define PL2_CRES 512        // touched blocks of 64 reads whose new records go through LDS to the final reduction
#define PL2_HEAVY 512       // dirty reads with more than four candidate rows that a tail hands to whole WAVES (round 6)
#define PL2_HSCR 512        // hot slots of one such read that a wave gathers per pass (LDS scratch over the final reduction's array)
#ifndef PL2_DW
#define PL2_DW 8            // dirty-bitmap words per thread and round of the tail's scan (16: 50 000 reads + 3.5 %, 500 000 - 3.3 %)
#endif
#define PL2_G3MAX 1024      // groups of the third level that the tail keeps bitmaps and per-wave lists for (more groups: the level is not used)
#define PL2_REFRESH 2       // third-level records a sweeping wave brings up to date per tail (a 64-lane reduction each)
#ifndef PL2_SW
#define PL2_SW 4            // block records per lane and round of the tail's sweep over all blocks of 64 reads
#endif

// Uw = 32-bit words of a slot's cell row: two 16-bit counters per word, rounded up to a multiple of 4 (rows are read 16 bytes at a time)
struct alignas(16) cf_pl2_rinfo { unsigned long long slot_base, cell_base; uint32_t hmask, Uw, anchor_off1, anchor_slot; };
// a candidate: hi = (s0 << 32 | s1) + 1 (0 = none), lo = offset << 32 | ~id rank (cf_place.h: cf_key), ext = first entry << 24 | entries
struct alignas(32) cf_pl2_rec { unsigned long long hi, lo, ext; uint32_t read, pad; };

// The contig (cloud_contig.py:26-41: clouds[pos][k-mer] += 1) BY K-MER: a genomic k-mer sits at one or two contig positions, so
// its record holds four [position + 1 : 32 | count : 32] words that the lane laying a (position, k-mer) pair down loads with
// ONE access (together with the k-mer's posting row) and then updates with one add or one claim: two round trips whatever the
// fill — the open-addressed (position, k-mer) map of rounds 1-3 cost a round trip per PROBE, and the slowest lane of ~1 500 sets
// the pace of a greedy iteration.  A k-mer with more than four positions continues in that map (cf_contig_add_hit).
struct alignas(32) cf_pl2_crec { unsigned long long s[4]; };

struct cf_pl2 {
    cf_place_state C;                // the contig map (cf_contig_add_hit) + ctl: [0] done, [1] n_out, [2] error flags, [3] arrivals
    const uint2* ent;                // per cloud entry: {k-mer rank, unit index inside its read}
    const int64_t* read_e;           // R + 1: first cloud entry of a read
    cf_pl2_crec* crec;               // per k-mer: the contig positions it has been laid at, with their counts
    const uint32_t* prow;            // per k-mer: a row of 32 or 64 words: postings [read : 32 - ib | unit index : ib], the last word = their number
    uint32_t ib;
    cf_pl2_rinfo* rinfo;
    unsigned long long* hdr; uint32_t* cells;      // per slot: header [offset + 1 : 32 | hits taken before the row became an anchor : 32], cell row
    uint32_t* hotbits;               // one bit per slot: the row has reached hot_thr hits (it may qualify: rescans look at it)
    uint32_t* dirty; uint32_t n_dirty_words;
    cf_pl2_rec* RB; cf_pl2_rec* L2;      // best candidate per read, per 64 reads
    cf_pl2_rec* L3; uint32_t* l3_stale;  // (large read sets) best candidate per GROUP of 2^g3s blocks, and one bit per group: its record is out of date
    uint32_t n3, g3s;                    // groups (0: the level is not used), log2 of the blocks per group
    uint32_t n_reads, n2;
    cf_pl2_rec* win;                 // the read the next launch lays down (hi == 0: none, the stage is over)
    uint32_t hot_thr;
    unsigned long long* stamps;      // -DCF_PL2_STAMPS: time per phase, summed by the last workgroup's thread 0
    unsigned long long* trace; unsigned long long* ttrace; uint32_t trace_iter;      // -DCF_PL2_STAMPS2: every wave's clock at 8 points of ONE iteration
};

// what crosses workgroups inside ONE launch is written by device-scope atomics and read with these (sc1: past the
// non-coherent L2 of the reader's XCD); the emulator of tests/emu defines them as plain accesses
#ifndef cf_ld_agent
__device__ __forceinline__ uint32_t cf_ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long cf_ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a workgroup barrier that waits for this wave's LDS traffic only (__syncthreads also waits for its global stores to be acknowledged)
__device__ __forceinline__ void cf_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// every memory operation of this wave has been performed (device-scope atomics: at the memory side)
__device__ __forceinline__ void cf_drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// a record comes out of memory as ONE 32-byte value: handed a reference into memory, hipcc 7.2 turned the field-wise selects of
// pl2_take into a select of the two ADDRESSES followed by flat loads, with the running best parked in scratch
struct alignas(32) cf_pl2_raw { unsigned long long a, b, c, d; };
__device__ __forceinline__ cf_pl2_rec pl2_load(const cf_pl2_rec* p) {
    const cf_pl2_raw v = *(const cf_pl2_raw*)p;
    cf_pl2_rec r; r.hi = v.a; r.lo = v.b; r.ext = v.c; r.read = (uint32_t)v.d; r.pad = 0u;
    return r;
}
__device__ __forceinline__ void pl2_store(cf_pl2_rec* p, const cf_pl2_rec& r) {
    cf_pl2_raw v; v.a = r.hi; v.b = r.lo; v.c = r.ext; v.d = (unsigned long long)r.read;
    *(cf_pl2_raw*)p = v;
}
__device__ __forceinline__ void pl2_take(cf_pl2_rec& m, const cf_pl2_rec o) {      // m := the better of (m, o), by selects (cf_place.h: why)
    const bool bt = o.hi > m.hi || (o.hi == m.hi && o.lo > m.lo);
    m.hi = bt ? o.hi : m.hi; m.lo = bt ? o.lo : m.lo; m.ext = bt ? o.ext : m.ext; m.read = bt ? o.read : m.read;
}
__device__ __forceinline__ cf_pl2_rec pl2_shfl_xor(const cf_pl2_rec& m, int d) {
    cf_pl2_rec o;
    o.hi = __shfl_xor(m.hi, d); o.lo = __shfl_xor(m.lo, d); o.ext = __shfl_xor(m.ext, d); o.read = (uint32_t)__shfl_xor((int)m.read, d); o.pad = 0;
    return o;
}

// lay (k-mer x, position q) down; true when the pair's count has just reached the threshold (it becomes "frequent").
// `seen` = the k-mer's record as device-scope loads of THIS launch returned it.  No other lane lays the same pair down in this
// launch (a k-mer occurs once per unit cloud, units of a read land on different positions) and earlier launches are complete, so
// the count seen for the pair IS its count: the add need not return — the returning atomic was the longest single step of the
// iteration (1.1 us under this kernel's load against 0.6 for the loads).  Only a pair that is not in the record yet goes through
// claims whose results are looked at.
__device__ __forceinline__ cf_pl2_crec pl2_crec_load(const cf_pl2& S, uint32_t x) {
    const unsigned long long* p = S.crec[x].s;
    return cf_pl2_crec{{cf_ld_agent(p), cf_ld_agent(p + 1), cf_ld_agent(p + 2), cf_ld_agent(p + 3)}};
}
// decide: which word of the record (0 .. 3, or -1: the pair is new) and whether the add makes the pair frequent.  No memory
// operation: a wave's memory operations complete IN ORDER (one counter), so anything asked for after an atomic also waits the
// atomic's 1.1 us — the add itself is committed at the end of the wave's work on the entry, next to the score atomics.
__device__ __forceinline__ bool pl2_contig_decide(const cf_pl2& S, uint32_t q, const cf_pl2_crec& seen, int& word) {
    // (no arrays indexed by a variable: hipcc parked `seen` in scratch for that)
    const unsigned long long q1 = (unsigned long long)(q + 1u);
    const unsigned long long s0 = seen.s[0], s1 = seen.s[1], s2 = seen.s[2], s3 = seen.s[3];
    const bool m0 = (s0 >> 32) == q1, m1 = (s1 >> 32) == q1, m2 = (s2 >> 32) == q1, m3 = (s3 >> 32) == q1;
    word = m0 ? 0 : m1 ? 1 : m2 ? 2 : m3 ? 3 : -1;
    const uint32_t cnt = (uint32_t)(m0 ? s0 : m1 ? s1 : m2 ? s2 : s3);
    if (word >= 0) return cnt + 1u == S.C.thr;
    // a new pair is frequent at once only with a threshold of 1 — and then only if a word of the record is free for it (else
    // it goes to the overflow map, whose add tells)
    return false;
}
// commit; returns true only for what decide could not know: a new pair under a threshold of 1, or a pair of the overflow map
__device__ __forceinline__ bool pl2_contig_commit(const cf_pl2& S, uint32_t x, uint32_t q, const cf_pl2_crec& seen, int word) {
    unsigned long long* rec = S.crec[x].s;
    if (word >= 0) { atomicAdd(rec + word, 1ull); return false; }
    const unsigned long long q1 = (unsigned long long)(q + 1u);
    // the first free word, decided by the claim itself (two units of the read may bring the same k-mer)
    uint32_t free_mask = (seen.s[0] == 0ull ? 1u : 0u) | (seen.s[1] == 0ull ? 2u : 0u) | (seen.s[2] == 0ull ? 4u : 0u) | (seen.s[3] == 0ull ? 8u : 0u);
    while (free_mask) {
        const int j = __ffs((int)free_mask) - 1;
        free_mask &= free_mask - 1u;
        if (atomicCAS(rec + j, 0ull, (q1 << 32) | 1ull) == 0ull) {
            if (S.C.thr == 1u) { S.C.freq_flag[x] = 1; return true; }
            return false;
        }
    }
    return cf_contig_add_hit(S.C, x, q);
}
__device__ __forceinline__ bool pl2_contig_add(const cf_pl2& S, uint32_t x, uint32_t q, const cf_pl2_crec& seen) {
    int word;
    const bool hit = pl2_contig_decide(S, q, seen, word);
    if (hit) S.C.freq_flag[x] = 1;
    return pl2_contig_commit(S, x, q, seen, word) || hit;
}

// In-kernel timelines (-DCF_PL2_STAMPS2, one clock reading per wave and phase): a load costs 0.45 - 0.65 us under this kernel's
// load, a device-scope atomic 1.1 - 1.2 us whether or not it returns, and an iteration waits for its SLOWEST lane.  Open
// addressing with a claim per probe made that lane walk 3 - 4 slots in most busy iterations.  So a hit first LOOKS — the
// offset's home bucket of 4 headers, one 32-byte access — and then sends ONE round of independent atomics to the slot it
// chose: the cell, the count (header's low half) and, for a new row, the claim of the key (CAS on the high half).  When
// the claim loses to another offset's, count and cell are taken back and the next free slot of the bucket is tried.
// Counters are therefore only exact once every workgroup has arrived — which is when the tail reads the cells.  The count
// is read at once, by the adder, to learn that a row is hot: a transient excess can only make a row hot EARLY (rescans
// look at it and find that it does not qualify), never late — the add that brings the true count to hot_thr returns at
// least hot_thr - 1.
// (`valid`: the lane has a posting to apply; `commit`: the lane also owes the contig its add — issued here, AFTER the loads of
// the wave's hit chains, because a wave's memory operations complete in order and a load behind an atomic waits 1.1 us with it)
__device__ __forceinline__ void pl2_hit(const cf_pl2& S, bool valid, uint32_t r, uint32_t i, uint32_t off, unsigned long long* commit) {
    cf_pl2_rinfo ri{};
    if (valid) ri = S.rinfo[r];
    const uint32_t off1 = off + 1u;
    const uint32_t inc = 1u << ((i & 1u) * 16u);
    const bool anchor = valid && off1 == ri.anchor_off1;      // the read's strongest row at its last rescan: the slot is known, the row is hot
    bool walk = valid && !anchor;
    const uint32_t n_bk = (ri.hmask + 1u) >> 2;
    uint32_t bk = cf_mix32(off1) & (n_bk - 1u);
    unsigned long long h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    if (walk) { const unsigned long long* hp = S.hdr + ri.slot_base + 4u * bk; h0 = cf_ld_agent(hp); h1 = cf_ld_agent(hp + 1); h2 = cf_ld_agent(hp + 2); h3 = cf_ld_agent(hp + 3); }
    uint32_t* h32 = (uint32_t*)(S.hdr + ri.slot_base);      // [2 h] hits, [2 h + 1] offset + 1
    // ---- atomics from here on
    uint32_t walked = 0;
    bool first = true;
    while (walk) {
        if (!first) { const unsigned long long* hp = S.hdr + ri.slot_base + 4u * bk; h0 = cf_ld_agent(hp); h1 = cf_ld_agent(hp + 1); h2 = cf_ld_agent(hp + 2); h3 = cf_ld_agent(hp + 3); }
        const uint32_t k0 = (uint32_t)(h0 >> 32), k1 = (uint32_t)(h1 >> 32), k2 = (uint32_t)(h2 >> 32), k3 = (uint32_t)(h3 >> 32);
        int j = k0 == off1 ? 0 : k1 == off1 ? 1 : k2 == off1 ? 2 : k3 == off1 ? 3 : -1;
        const uint32_t seen_cnt = (uint32_t)(j == 0 ? h0 : j == 1 ? h1 : j == 2 ? h2 : h3);
        uint32_t free_mask = (k0 == 0u ? 1u : 0u) | (k1 == 0u ? 2u : 0u) | (k2 == 0u ? 4u : 0u) | (k3 == 0u ? 8u : 0u);
        bool claim = false;
        for (;;) {
            if (j < 0) {      // the offset has no row yet (as far as this lane has seen): the next free slot of the bucket
                if (!free_mask) break;      // full of other offsets: the next bucket
                j = __ffs((int)free_mask) - 1;
                free_mask &= free_mask - 1u;
                claim = true;
            }
            const uint32_t h = 4u * bk + (uint32_t)j;
            uint32_t* cell = &S.cells[ri.cell_base + (unsigned long long)h * ri.Uw + (i >> 1)];
            const uint32_t gs = (uint32_t)ri.slot_base + h;
            const uint32_t was = atomicAdd(&h32[2 * h], 1u);
            const uint32_t cur = claim ? atomicCAS(&h32[2 * h + 1], 0u, off1) : off1;
            atomicAdd(cell, inc);
            // the row was hot already when this lane looked: its marks go out with the adds (they are idempotent)
            const bool early = !claim && seen_cnt + 1u >= S.hot_thr;
            if (early) { atomicOr(&S.hotbits[gs >> 5], 1u << (gs & 31)); atomicOr(&S.dirty[r >> 5], 1u << (r & 31)); }
            if (first && commit) { atomicAdd(commit, 1ull); commit = nullptr; }
            first = false;
            if (cur == 0u || cur == off1) {
                if (was + 1u >= S.hot_thr && !early) { atomicOr(&S.hotbits[gs >> 5], 1u << (gs & 31)); atomicOr(&S.dirty[r >> 5], 1u << (r & 31)); }
                walk = false;
                break;
            }
            atomicSub(&h32[2 * h], 1u); atomicSub(cell, inc);      // another offset took the slot meanwhile
            j = -1;
        }
        if (!walk) break;
        first = false;
        bk = (bk + 1u) & (n_bk - 1u);
        if (++walked >= n_bk) { atomicOr(&S.C.ctl[2], 2u); break; }      // the region is full: the host starts over with larger regions
    }
    if (anchor) {      // s1 of an anchor row = the sum of its cells
        atomicAdd(&S.cells[ri.cell_base + (unsigned long long)ri.anchor_slot * ri.Uw + (i >> 1)], inc);
        atomicOr(&S.dirty[r >> 5], 1u << (r & 31));
    }
    if (commit) atomicAdd(commit, 1ull);
}

// one hot row of a read seen by one lane: s0 = non-zero cells, s1 = their sum (header and cells are asked for together)
__device__ __forceinline__ void pl2_row(const cf_pl2& S, const cf_pl2_rinfo& ri, uint32_t slot, uint32_t rank, uint32_t r,
                                        cf_pl2_rec& best, unsigned long long& anchor, uint32_t& anchor_off1) {
    const unsigned long long hd = cf_ld_agent(&S.hdr[ri.slot_base + slot]);
    const unsigned long long* row = (const unsigned long long*)(S.cells + ri.cell_base + (unsigned long long)slot * ri.Uw);
    uint32_t s0 = 0, s1 = 0;
    for (uint32_t q = 0; q < ri.Uw / 2; q += 2) {      // (Uw is a multiple of 4 words)
        const unsigned long long c0 = cf_ld_agent(row + q), c1 = cf_ld_agent(row + q + 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t a = (uint32_t)(c0 >> (16 * t)) & 0xFFFFu, b = (uint32_t)(c1 >> (16 * t)) & 0xFFFFu;
            s0 += (a != 0u) + (b != 0u); s1 += a + b;
        }
    }
    const uint32_t off = (uint32_t)(hd >> 32) - 1u;
    const unsigned long long a = (((unsigned long long)s1 << 32) | slot) + 1ull;
    anchor_off1 = a > anchor ? (uint32_t)(hd >> 32) : anchor_off1;
    anchor = a > anchor ? a : anchor;
    if (s0 >= S.C.min_unit && (unsigned long long)s0 * S.C.min_prop <= s1 && s1 >= S.C.min_inters) {
        const cf_pl2_rec c{(((unsigned long long)s0 << 32) | s1) + 1ull, ((unsigned long long)off << 32) | (unsigned long long)(~rank), best.ext, r, 0u};
        pl2_take(best, c);
    }
}

// ---- one dirty read by one LANE: recomputed from its hot rows (RB, anchor); its block of 64 is marked (LDS)
__device__ __forceinline__ void pl2_rescan_read(const cf_pl2& S, uint32_t r, uint32_t* bb2, uint32_t* list2, uint32_t* n_list2, uint32_t* bb3, uint32_t* heavy, uint32_t* n_heavy) {
    const cf_pl2_rinfo ri = S.rinfo[r];
    const bool used = S.C.used[r] != 0;
    const uint32_t rank = (uint32_t)S.C.id_rank[r];
    const int64_t e0 = S.read_e[r], e1 = S.read_e[r + 1];
    cf_pl2_rec rec{0ull, 0ull, ((unsigned long long)e0 << 24) | (unsigned long long)(e1 - e0), r, 0u};
    unsigned long long anchor = 0ull;      // (s1 << 32 | slot) + 1 of the strongest hot row
    uint32_t anchor_off1 = 0u;
    if (!used) {
        // Lanes of a wave rescan different reads: whatever ONE lane does costs the wave a round trip.  So a lane first gathers its
        // hot slots from the bitmap words (registers only), then the wave looks at everybody's first hot row, then everybody's
        // second, ...: as many round trips as the read with most hot rows has — not one per bitmap word that holds a bit of
        // some lane, which made one pass of a busy tail take 11 us.
        const unsigned long long* hb = (const unsigned long long*)S.hotbits + (ri.slot_base >> 6);      // (regions begin at multiples of 64 slots)
        const uint32_t n_w = (ri.hmask + 1u) >> 6;
        const uint32_t a_slot = ri.anchor_off1 ? ri.anchor_slot : 0xFFFFFFFFu;
        uint32_t p0 = 0xFFFFFFFFu, p1 = 0xFFFFFFFFu, p2 = 0xFFFFFFFFu, p3 = 0xFFFFFFFFu;
        uint32_t np = 0;      // hot rows besides the anchor (the first four are kept)
        for (uint32_t w0 = 0; w0 < n_w; w0 += 16) {
            unsigned long long bits[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) bits[j] = w0 + (uint32_t)j < n_w ? cf_ld_agent(hb + w0 + j) : 0ull;
            // the anchor is hot and, mostly, the read's only hot row: its header and cells are asked for with the bits
            if (w0 == 0 && a_slot != 0xFFFFFFFFu) pl2_row(S, ri, a_slot, rank, r, rec, anchor, anchor_off1);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                for (unsigned long long b = bits[j]; b; b &= b - 1ull) {
                    const uint32_t slot = (w0 + (uint32_t)j) * 64u + (uint32_t)(__ffsll((long long)b) - 1);
                    if (slot == a_slot) continue;
                    p0 = np == 0 ? slot : p0; p1 = np == 1 ? slot : p1; p2 = np == 2 ? slot : p2; p3 = np == 3 ? slot : p3;
                    ++np;
                }
        }
        if (np > 4) {
            // Round 6: a read with MANY candidate rows — thin coverage, k-mers that are not unique to one place of the array — goes to a list
            // that whole waves work off behind the lane pass (pl2_rescan_read_wave: a lane per ROW); one lane walking them one after the
            // other was 254 us of a 270-us iteration on such reads (round 5), and the run was handed to the hash-map path.
            // (still counted: a run that has more than a couple of them per greedy iteration is faster on the hash-map path, cf_place2_run)
            atomicAdd(&S.C.ctl[6], 1u);
            const uint32_t hp = atomicAdd(n_heavy, 1u);
            if (hp < (uint32_t)PL2_HEAVY) { heavy[hp] = r; return; }
        }
        if (np > 0) pl2_row(S, ri, p0, rank, r, rec, anchor, anchor_off1);
        if (np > 1) pl2_row(S, ri, p1, rank, r, rec, anchor, anchor_off1);
        if (np > 2) pl2_row(S, ri, p2, rank, r, rec, anchor, anchor_off1);
        if (np > 3) pl2_row(S, ri, p3, rank, r, rec, anchor, anchor_off1);
        if (np > 4) {      // more than four: the rest, word by word (never seen at the default thresholds on reads of unique k-mers)
            uint32_t seen = 0;
            for (uint32_t w = 0; w < n_w; ++w)
                for (unsigned long long b = cf_ld_agent(hb + w); b; b &= b - 1ull) {
                    const uint32_t slot = w * 64u + (uint32_t)(__ffsll((long long)b) - 1);
                    if (slot == a_slot) continue;
                    if (seen++ >= 4) pl2_row(S, ri, slot, rank, r, rec, anchor, anchor_off1);
                }
        }
    }
    pl2_store(&S.RB[r], rec);
    if (anchor) { S.rinfo[r].anchor_slot = (uint32_t)(anchor - 1ull); S.rinfo[r].anchor_off1 = anchor_off1; }
    const uint32_t i2 = r >> 6, bit = 1u << (i2 & 31);
    if (!(atomicOr(&bb2[i2 >> 5], bit) & bit)) {
        list2[atomicAdd(n_list2, 1u)] = i2;
        if (S.n3) { const uint32_t g = i2 >> S.g3s; atomicOr(&bb3[g >> 5], 1u << (g & 31)); }      // the block's group of the third level is touched
    }
}

// ---- one dirty read by one WAVE (round 6): its hot slots are gathered from the bitmap words into an LDS scratch (lanes over the words, a
// scan of their counts), then a lane per ROW — a round trip per 64 rows instead of one per row; best candidate and anchor by butterflies.
// Same outcome as the lane version: the best qualifying row by (s0, s1, offset, id) and the anchor = the hot row with the largest (s1, slot).
__device__ __forceinline__ void pl2_rescan_read_wave(const cf_pl2& S, uint32_t r, uint32_t* bb2, uint32_t* list2, uint32_t* n_list2, uint32_t* bb3, uint32_t* scratch) {
    const int lane = threadIdx.x & 63;
    const cf_pl2_rinfo ri = S.rinfo[r];
    const bool used = S.C.used[r] != 0;
    const uint32_t rank = (uint32_t)S.C.id_rank[r];
    const int64_t e0 = S.read_e[r], e1 = S.read_e[r + 1];
    cf_pl2_rec rec{0ull, 0ull, ((unsigned long long)e0 << 24) | (unsigned long long)(e1 - e0), r, 0u};
    unsigned long long anchor = 0ull;
    uint32_t anchor_off1 = 0u;
    if (!used) {
        const unsigned long long* hb = (const unsigned long long*)S.hotbits + (ri.slot_base >> 6);
        const uint32_t n_w = (ri.hmask + 1u) >> 6;
        for (uint32_t skip = 0;; skip += (uint32_t)PL2_HSCR) {      // (one pass unless the read has more than PL2_HSCR hot rows)
            uint32_t total = 0;      // hot slots of the read (wave-uniform)
            for (uint32_t w0 = 0; w0 < n_w; w0 += 64u) {
                const uint32_t w = w0 + (uint32_t)lane;
                const unsigned long long bits = w < n_w ? cf_ld_agent(hb + w) : 0ull;
                const uint32_t c = (uint32_t)__popcll(bits);
                uint32_t inc = c;
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, (unsigned)d); if (lane >= d) inc += o; }
                uint32_t at = total + inc - c;
                for (unsigned long long b = bits; b; b &= b - 1ull, ++at)
                    if (at >= skip && at < skip + (uint32_t)PL2_HSCR) scratch[at - skip] = w * 64u + (uint32_t)(__ffsll((long long)b) - 1);
                total += (uint32_t)__shfl((int)inc, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_here = total > skip ? min(total - skip, (uint32_t)PL2_HSCR) : 0u;
            for (uint32_t i = (uint32_t)lane; i < n_here; i += 64u) pl2_row(S, ri, scratch[i], rank, r, rec, anchor, anchor_off1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (total <= skip + (uint32_t)PL2_HSCR) break;
        }
        for (int d = 1; d <= 32; d <<= 1) {
            pl2_take(rec, pl2_shfl_xor(rec, d));
            const unsigned long long oa = __shfl_xor(anchor, d);
            const uint32_t oo = (uint32_t)__shfl_xor((int)anchor_off1, d);
            anchor_off1 = oa > anchor ? oo : anchor_off1;
            anchor = oa > anchor ? oa : anchor;
        }
        rec.ext = ((unsigned long long)e0 << 24) | (unsigned long long)(e1 - e0); rec.read = r;      // (a read without a qualifying row keeps its own entry range)
    }
    if (lane == 0) {
        pl2_store(&S.RB[r], rec);
        if (anchor) { S.rinfo[r].anchor_slot = (uint32_t)(anchor - 1ull); S.rinfo[r].anchor_off1 = anchor_off1; }
        const uint32_t i2 = r >> 6, bit = 1u << (i2 & 31);
        if (!(atomicOr(&bb2[i2 >> 5], bit) & bit)) {
            list2[atomicAdd(n_list2, 1u)] = i2;
            if (S.n3) { const uint32_t g = i2 >> S.g3s; atomicOr(&bb3[g >> 5], 1u << (g & 31)); }
        }
    }
}

#ifdef CF_PL2_STAMPS
#define PL2_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&S.stamps[i], t_ - t_last); t_last = t_; } } while (0)
#else
#define PL2_STAMP(i) do { } while (0)
#endif

// ---- the tail: recompute the dirty reads, then the blocks of 64 reads that hold one, reduce the blocks, publish the next
// winner.  One workgroup.  A busy iteration dirties a thousand reads scattered over the read numbers: they are packed into
// a dense list so that every lane of the workgroup has one (rescans by blocks of 8 left most lanes of a pass without work
// and needed eight passes).
// out_idx: the line of the output this tail's winner gets = the number of reads this stage has placed so far, which the HOST knows (one
// per launch): it comes as a kernel argument — round 4 kept it in memory and the lane that publishes the winner paid a dependent load for it
// placed: the read this launch has just laid down (its RB entry and its block are redone here: it is used now), or 0xFFFFFFFF — round 6:
// it comes in a register; rounds 4-5 had the PREVIOUS tail mark it in the dirty bitmap with a device-scope atomic, the last memory operation
// of every launch (1.1 us under this kernel's load, whether or not it returns)
__device__ void pl2_tail(const cf_pl2& S, uint32_t out_idx, uint32_t placed, unsigned long long t_last = 0ull) {
    (void)t_last;
#ifdef CF_PL2_STAMPS2
    const bool ttr = S.C.ctl[1] == S.trace_iter && (threadIdx.x & 63) == 0;
    unsigned long long* ttw = S.ttrace + (threadIdx.x >> 6) * 8;
#define PL2_TTRACE(i) do { if (ttr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ttw[i] = wall_clock64(); } } while (0)
#else
#define PL2_TTRACE(i) do { } while (0)
#endif
    PL2_TTRACE(0);
    uint32_t* lds32 = (uint32_t*)cf_lds;
    uint32_t* n_list = lds32; uint32_t* more = lds32 + 1; uint32_t* n_list2 = lds32 + 2; uint32_t* n_heavy = lds32 + 3;
    uint32_t* list = lds32 + 4;
    uint32_t* list2 = list + PL2_LIST;
    unsigned long long* red = (unsigned long long*)(list2 + PL2_LIST);      // the sweeping lanes' bests: up to 16 waves x 64 lanes x {hi, lo, ext, read}
    unsigned long long* cres = red + 4 * 1024;                                 // new records of the touched blocks of 64 (fused last round)
    uint32_t* bb2 = (uint32_t*)(cres + 4 * PL2_CRES);
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const uint32_t n_b2words = (S.n2 + 31u) >> 5;
    // third level (S.n3 > 0): bb3 = groups with a touched block, st3 = the groups whose record is out of date (a copy of S.l3_stale,
    // written back at the end), glist = per sweeping wave the groups whose blocks it has to look at one by one
    uint32_t* bb3 = bb2 + n_b2words;
    uint32_t* st3 = bb3 + PL2_G3MAX / 32;
    //  st3 is READ-ONLY during the sweep: the waves decide from it which groups' records they may load, and a bit cleared
    // by the wave that has just stored a mended record — with no barrier between that store and the other waves' loads — let them read
    // the old record.  The sweep's changes go to st3n, which is what is written back.
    uint32_t* st3n = st3 + PL2_G3MAX / 32;
    uint16_t* glist = (uint16_t*)(st3n + PL2_G3MAX / 32);      // 16 waves x (PL2_G3MAX / 8) entries
    uint32_t* heavy = (uint32_t*)(glist + 16 * (PL2_G3MAX / 8));      // PL2_HEAVY dirty reads with many candidate rows: a wave each (their scratch: `red`, free until the sweep)
    const uint32_t n_b3words = (S.n3 + 31u) >> 5;
    for (uint32_t i = tid; i < n_b2words; i += nthr) bb2[i] = 0u;
    for (uint32_t i = tid; i < n_b3words; i += nthr) { bb3[i] = 0u; const uint32_t w3 = S.l3_stale[i]; st3[i] = w3; st3n[i] = w3; }
    bool again = true;
    bool first_round = true;
    while (again) {
        if (tid == 0) { *n_list = 0u; *more = 0u; *n_list2 = 0u; *n_heavy = 0u; if (first_round && placed != 0xFFFFFFFFu) { list[0] = placed; *n_list = 1u; } }
        first_round = false;
        __syncthreads();
        // every workgroup has arrived: plain traffic on the bitmap.  A thread takes PL2_DW words per round, ALL loaded before the first is
        // looked at: the bitmap is one bit per read, and a round trip per pair of words made this scan 8 dependent round trips at 500 000
        // reads (one at 50 000).
        for (uint32_t w0 = tid; w0 < S.n_dirty_words; w0 += (uint32_t)PL2_DW * nthr) {
            uint32_t bw[PL2_DW];
#pragma unroll
            for (int t = 0; t < PL2_DW; ++t) { const uint32_t wi = w0 + (uint32_t)t * nthr; bw[t] = wi < S.n_dirty_words ? cf_ld_agent(&S.dirty[wi]) : 0u; }
#pragma unroll
            for (int t = 0; t < PL2_DW; ++t) {
                uint32_t bits = bw[t];
                const uint32_t wi = w0 + (uint32_t)t * nthr;
                if (!bits) continue;
                const uint32_t cnt = (uint32_t)__popc(bits);
                const uint32_t k0 = atomicAdd(n_list, cnt);
                if (k0 + cnt <= PL2_LIST) {
                    for (uint32_t k = k0; bits; bits &= bits - 1u) list[k++] = wi * 32u + (uint32_t)(__ffs((int)bits) - 1);
                    S.dirty[wi] = 0u;
                } else {      // (what does not fit the list stays in the bitmap for the next round)
                    for (uint32_t k = k0; k < k0 + cnt && k < PL2_LIST; ++k) list[k] = 0xFFFFFFFFu;
                    *more = 1u;
                }
            }
        }
        PL2_TTRACE(1);
        cf_barrier_lds();      // (the rescans read the list, which is in LDS: the stores that cleared the bitmap need not have landed)
        PL2_TTRACE(2);
        PL2_STAMP(3);
        const uint32_t n = min(*n_list, (uint32_t)PL2_LIST);
        again = *more != 0u;
        for (uint32_t k = tid; k < n; k += nthr) {
            const uint32_t r = list[k];
            if (r < S.n_reads && r != 0xFFFFFFFFu) pl2_rescan_read(S, r, bb2, list2, n_list2, bb3, heavy, n_heavy);
        }
        PL2_TTRACE(3);
        __syncthreads();
        {   // the reads the lanes handed over: a wave each (nearly always none: no second barrier then)
            const uint32_t nh = min(*n_heavy, (uint32_t)PL2_HEAVY);
            if (nh) {
                for (uint32_t h = (uint32_t)wave; h < nh; h += (uint32_t)nw) pl2_rescan_read_wave(S, heavy[h], bb2, list2, n_list2, bb3, (uint32_t*)red + (size_t)wave * PL2_HSCR);
                __syncthreads();
            }
        }
        PL2_TTRACE(4);
        PL2_STAMP(4);
        // Blocks of 64 reads with a rescanned read: 8 lanes per block, 8 RB records per lane.  In the last round (almost always the
        // only one) the blocks that were NOT touched are reduced at the same time by the first waves — they skip the touched
        // ones, whose new records come through LDS — so the two levels cost one round trip and one barrier, not two of each.
        const uint32_t m = *n_list2;
        const bool fused = !again && m <= PL2_CRES;
        const int nfw = (int)min((uint32_t)(S.n2 > 2048u ? nw - nw / 4 : nw / 2), (S.n2 + 255u) >> 8);      // waves of the sweep over all blocks (three quarters of the workgroup for large read sets: the touched blocks need few)
        const int w_first = fused ? nfw : 0, w_n = nw - w_first;
        if (wave >= w_first)
            for (uint32_t k = (uint32_t)(wave - w_first) * 8u; k < m; k += (uint32_t)w_n * 8u) {
                const uint32_t kk = k + ((uint32_t)lane >> 3);
                const bool have = kk < m;
                const uint32_t i2 = have ? list2[kk] : 0u, r0 = i2 * 64u + ((uint32_t)lane & 7u) * 8u;
                cf_pl2_rec rec{0ull, 0ull, 0ull, 0u, 0u};
                if (have) {
                    cf_pl2_rec o[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) o[t] = r0 + (uint32_t)t < S.n_reads ? pl2_load(&S.RB[r0 + (uint32_t)t]) : cf_pl2_rec{0ull, 0ull, 0ull, 0u, 0u};
#pragma unroll
                    for (int t = 0; t < 8; ++t) pl2_take(rec, o[t]);
                }
                pl2_take(rec, pl2_shfl_xor(rec, 1)); pl2_take(rec, pl2_shfl_xor(rec, 2)); pl2_take(rec, pl2_shfl_xor(rec, 4));
                if (have && (lane & 7) == 0) {
                    pl2_store(&S.L2[i2], rec);
                    if (fused) { cres[4 * kk] = rec.hi; cres[4 * kk + 1] = rec.lo; cres[4 * kk + 2] = rec.ext; cres[4 * kk + 3] = rec.read; }
                    else atomicAnd(&bb2[i2 >> 5], ~(1u << (i2 & 31)));
                }
            }
        if (again || !fused) { __syncthreads(); PL2_STAMP(5); }
        if (!again) {
            // all blocks of 64: lanes take 4 records at a time (the loads of a batch are in flight together)
            if (wave < nfw && S.n3 && fused) {
                // Third level (read sets of more than 2 048 blocks; round 5: the sweep over ALL blocks made the tail grow with the read set —
                // 7 800 records of 32 bytes through one CU at 500 000 reads, three rounds of loads).  L3[g] = the best candidate of the blocks
                // of group g, kept LAZILY: a group with a block touched in this tail is marked out of date (its blocks' new records are
                // being written right now) and is looked at block by block — as are the groups still out of date from earlier tails —,
                // every other group is ONE record.  A sweeping wave brings up to PL2_REFRESH out-of-date groups that were NOT touched
                // this time up to date (all their block records are in its lanes: one 64-lane reduction each).  A busy tail touches
                // every group and sweeps like before; the light tails that follow (most are: a handful of dirty reads) mend the level
                // a few groups at a time and then read n3 + 64 x (touched groups) records instead of n2.
                cf_pl2_rec mine{0ull, 0ull, 0ull, 0u, 0u};
                const cf_pl2_rec none{0ull, 0ull, 0ull, 0u, 0u};
                const uint32_t gb = 1u << S.g3s;      // blocks per group (<= 64: a lane per block)
                // (a) this wave's groups that need their blocks: a wave-private list (lane 0 writes, the wave reads)
                uint16_t* gl = glist + (size_t)wave * (PL2_G3MAX / 8);
                uint32_t n_gl = 0;
                for (uint32_t g = (uint32_t)wave; g < S.n3; g += (uint32_t)nfw) {      // (wave-uniform)
                    const uint32_t bt = 1u << (g & 31);
                    if ((st3[g >> 5] | bb3[g >> 5]) & bt) { if (lane == 0) gl[n_gl] = (uint16_t)g; ++n_gl; }
                }
                // (b) groups that are up to date and untouched: one record per lane (all in flight together)
                {
                    cf_pl2_rec o[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {      // (two per lane: the host uses the level only when n3 <= 128 x the sweeping waves)
                        const uint32_t g = (uint32_t)(wave * 64 + lane) + (uint32_t)t * (uint32_t)nfw * 64u;
                        const bool kt = g < S.n3 && !(((st3[g >> 5] | bb3[g >> 5]) >> (g & 31)) & 1u);
                        o[t] = kt ? pl2_load(&S.L3[g]) : none;
                    }
                    // (c) block by block, PL2_SW groups in flight
                    uint32_t refreshed = 0;
                    for (uint32_t i0 = 0; i0 < n_gl; i0 += (uint32_t)PL2_SW) {
                        cf_pl2_rec q[PL2_SW];
                        uint32_t gq[PL2_SW];
#pragma unroll
                        for (int t = 0; t < PL2_SW; ++t) {
                            gq[t] = i0 + (uint32_t)t < n_gl ? (uint32_t)gl[i0 + (uint32_t)t] : 0xFFFFFFFFu;
                            const uint32_t b = (gq[t] << S.g3s) + (uint32_t)lane;
                            const bool kt = gq[t] != 0xFFFFFFFFu && (uint32_t)lane < gb && b < S.n2 && !((bb2[b >> 5] >> (b & 31)) & 1u);
                            q[t] = kt ? pl2_load(&S.L2[b]) : none;
                        }
#pragma unroll
                        for (int t = 0; t < PL2_SW; ++t) {
                            pl2_take(mine, q[t]);
                            if (gq[t] == 0xFFFFFFFFu) continue;      // (wave-uniform)
                            const uint32_t g = gq[t], bt = 1u << (g & 31);
                            if ((bb3[g >> 5] & bt)) { if (lane == 0) atomicOr(&st3n[g >> 5], bt); }      // touched now: out of date from here on
                            else if (refreshed < (uint32_t)PL2_REFRESH) {      // out of date, untouched: every block record of the group is in q[t]
                                cf_pl2_rec w = q[t];
                                for (int d = 1; d <= 32; d <<= 1) pl2_take(w, pl2_shfl_xor(w, d));
                                if (lane == 0) { pl2_store(&S.L3[g], w); atomicAnd(&st3n[g >> 5], ~bt); }      // (up to date from the NEXT launch on)
                                ++refreshed;
                            }
                        }
                    }
                    pl2_take(mine, o[0]); pl2_take(mine, o[1]);
                }
                const uint32_t at = 4u * (uint32_t)(wave * 64 + lane);
                red[at] = mine.hi; red[at + 1] = mine.lo; red[at + 2] = mine.ext; red[at + 3] = mine.read;
            } else if (wave < nfw) {
                if (S.n3) for (uint32_t i = (uint32_t)(wave * 64 + lane); i < n_b3words; i += (uint32_t)nfw * 64u) st3n[i] = 0xFFFFFFFFu;      // (a tail whose touched blocks went through several rounds: every group out of date)
                cf_pl2_rec mine{0ull, 0ull, 0ull, 0u, 0u};
                const uint32_t stride = (uint32_t)nfw * 64u;
                for (uint32_t b = (uint32_t)(wave * 64 + lane); b < S.n2; b += (uint32_t)PL2_SW * stride) {      // (PL2_SW loads in flight per lane)
                    const cf_pl2_rec none{0ull, 0ull, 0ull, 0u, 0u};
                    cf_pl2_rec o[PL2_SW];
#pragma unroll
                    for (int t = 0; t < PL2_SW; ++t) {
                        const uint32_t bt = b + (uint32_t)t * stride;
                        const bool kt = bt < S.n2 && (!fused || !((bb2[bt >> 5] >> (bt & 31)) & 1u));
                        o[t] = kt ? pl2_load(&S.L2[bt]) : none;
                    }
#pragma unroll
                    for (int t = 0; t < PL2_SW; ++t) pl2_take(mine, o[t]);
                }
                // every lane's best goes to LDS as it is: ONE reduction over the wave's lanes, in the last wave standing, instead of one
                // per sweeping wave and another over their results (a 64-lane reduction of a record is 42 lane permutes)
                const uint32_t at = 4u * (uint32_t)(wave * 64 + lane);
                red[at] = mine.hi; red[at + 1] = mine.lo; red[at + 2] = mine.ext; red[at + 3] = mine.read;
            }
            PL2_TTRACE(5);
            cf_barrier_lds();      // (what the last wave reads — the sweeping lanes' bests, the touched blocks' new records — is in LDS: the stores of the
                                   // new block records to memory need not have been acknowledged; round 4 waited for them here, in every tail)
            PL2_TTRACE(6);
            PL2_STAMP(6);
            if (wave == 0) {      // the sweeping lanes' results and the touched blocks' new records
                cf_pl2_rec w{0ull, 0ull, 0ull, 0u, 0u};
                for (int k = 0; k < nfw; ++k) { const uint32_t at = 4u * (uint32_t)(k * 64 + lane); pl2_take(w, cf_pl2_rec{red[at], red[at + 1], red[at + 2], (uint32_t)red[at + 3], 0u}); }
                if (fused)
                    for (uint32_t k = (uint32_t)lane; k < m; k += 64u) pl2_take(w, cf_pl2_rec{cres[4 * k], cres[4 * k + 1], cres[4 * k + 2], (uint32_t)cres[4 * k + 3], 0u});
                {   // 64 lanes -> 1: the KEY (hi, lo) goes through the butterfly (4 dwords per step instead of 7), the payload comes from the lane that holds it
                    unsigned long long khi = w.hi, klo = w.lo;
                    for (int d = 1; d <= 32; d <<= 1) {
                        const unsigned long long ohi = __shfl_xor(khi, d), olo = __shfl_xor(klo, d);
                        const bool bt = ohi > khi || (ohi == khi && olo > klo);
                        khi = bt ? ohi : khi; klo = bt ? olo : klo;
                    }
                    const unsigned long long mine_m = __ballot(w.hi == khi && w.lo == klo);      // (equal keys are equal candidates: lo holds the read's id rank; all-empty: every lane)
                    const int src = __ffsll((long long)mine_m) - 1;
                    w.ext = __shfl(w.ext, src); w.read = (uint32_t)__shfl((int)w.read, src); w.hi = khi; w.lo = klo;
                }
                if (lane == 0) {
                    pl2_store(S.win, w);
                    if (!w.hi) S.C.ctl[0] = 1u;
                    else {
                        const unsigned int o = out_idx;
                        S.C.ctl[1] = o + 1u;      // (the host reads the number of placements from here)
                        S.C.out_read[o] = (int64_t)w.read; S.C.out_pos[o] = (int64_t)(w.lo >> 32);
                        S.C.out_s0[o] = (int32_t)((w.hi - 1ull) >> 32); S.C.out_s1[o] = (int32_t)(uint32_t)(w.hi - 1ull);
                        S.C.used[w.read] = 1;      // (its RB entry and its block are redone by the next launch's tail, which gets the read as `placed`)
                    }
                    S.C.ctl[3] = 0u;
                }
            }
            if (S.n3 && wave == 1) for (uint32_t i = (uint32_t)lane; i < n_b3words; i += 64u) S.l3_stale[i] = st3n[i];      // (behind the sweep's barrier: the bits are final)
        }
    }
    PL2_TTRACE(7);
    PL2_STAMP(7);
}

static size_t pl2_lds_bytes(uint32_t n2) { return (16 + (size_t)PL2_LIST * 8 + 1024 * 32 + (size_t)PL2_CRES * 32 + (size_t)((n2 + 31) / 32) * 4 + 3 * (PL2_G3MAX / 32) * 4 + 16 * (PL2_G3MAX / 8) * 2 + (size_t)PL2_HEAVY * 4 + 16); }

__global__ void __launch_bounds__(PL2_B)
cf_pl2_tail_kernel(cf_pl2 S) {
#ifdef CF_PL2_STAMPS
    pl2_tail(S, 0u, 0xFFFFFFFFu, wall_clock64());
#else
    pl2_tail(S, 0u, 0xFFFFFFFFu);
#endif
}

// ---- one greedy iteration: lay the winner onto the contig, apply the events it raises, then (last workgroup) pick the next.
// PW = 32-bit words of a posting row (32 or 64: the smallest that holds the longest posting list; longer lists continue in the CSR arrays)
template <int PW>
__global__ void __launch_bounds__(PL2_B)
cf_pl2_iter_kernel(cf_pl2 S, uint32_t it /* greedy iteration of the stage, 0-based: its winner was published as line `it` */) {
#ifdef CF_PL2_STAMPS
    unsigned long long t_last = wall_clock64();
#endif
    const cf_pl2_rec w = pl2_load(S.win);
    if (!w.hi) return;      // the stage is over (launches are enqueued ahead)
#ifdef CF_PL2_STAMPS
    const unsigned long long t_win = wall_clock64();
#endif
#ifdef CF_PL2_STAMPS2
    const bool tr = S.C.ctl[1] == S.trace_iter && (threadIdx.x & 63) == 0;
    unsigned long long* trw = S.trace + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8;
#define PL2_TRACE(i) do { if (tr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); trw[i] = wall_clock64(); } } while (0)
    if (tr) { trw[0] = t_last; trw[1] = t_win; }
#else
#define PL2_TRACE(i) do { } while (0)
#endif
    // PW lanes per cloud entry of the read, ONE posting per lane: a lane's hit is a chain of dependent steps (region record,
    // bucket, atomics) and two postings in one lane ran their chains one after the other — the slowest lanes of an iteration
    const uint32_t off = (uint32_t)(w.lo >> 32), n_e = (uint32_t)(w.ext & 0xFFFFFFull);
    const unsigned long long e0 = w.ext >> 24;
    const int sub = threadIdx.x & (PW - 1);
    const uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) / PW, n_grp = (gridDim.x * blockDim.x) / PW;
    const uint32_t imask = (1u << S.ib) - 1u;
    PL2_TRACE(2);
    for (uint32_t g = grp; g < n_e; g += n_grp) {
        const uint2 en = S.ent[e0 + g];
        const uint32_t x = en.x, q = off + en.y;
        const uint32_t pw = S.prow[(unsigned long long)x * PW + sub];      // this lane's posting (the row's last word: the k-mer's posting count)
        int hit = 0;
        unsigned long long* owed = nullptr;      // lane 0: the contig word its add goes to (committed behind the hit chains' loads)
        if (sub == 0) {
            const cf_pl2_crec seen = pl2_crec_load(S, x);
            int word;
            hit = pl2_contig_decide(S, q, seen, word) ? 1 : 0;
            if (word >= 0) owed = S.crec[x].s + word;
            else hit = pl2_contig_commit(S, x, q, seen, word) ? 1 : 0;      // a new pair: claimed at once (frequent at once only with a threshold of 1)
        }
        const uint32_t n_post = (uint32_t)__shfl((int)pw, PW - 1, PW);
        const uint32_t pr = pw >> S.ib, pi = pw & imask;
        const bool pv = (uint32_t)sub < min(n_post, (uint32_t)(PW - 1)) && q >= pi;
        hit = __shfl(hit, 0, PW);
        if (!hit) { if (owed) atomicAdd(owed, 1ull); continue; }
        if (pv || owed) pl2_hit(S, pv, pr, pi, q - pi, owed);
        if (n_post > (uint32_t)(PW - 1)) {      // a k-mer with more postings than a row holds: the rest from the CSR lists
            const int64_t p0 = S.C.post_ptr[x];
            for (uint32_t p = (uint32_t)(PW - 1) + (uint32_t)sub; p < n_post; p += PW) {
                const unsigned long long prr = S.C.post_ri[p0 + p];
                const uint32_t r2 = (uint32_t)(prr >> 32), i2 = (uint32_t)prr;
                if (q >= i2) pl2_hit(S, true, r2, i2, q - i2, nullptr);
            }
        }
        if (sub == 0) S.C.freq_flag[x] = 1;      // (a store: behind the loads, like the atomics)
    }
    // arrive; the last workgroup runs the tail.  Everything the tail reads from this phase was written by device-scope
    // atomics: once a wave's counter of outstanding memory operations is zero they have been performed.
#ifdef CF_PL2_STAMPS
    const unsigned long long t_body = wall_clock64();
#endif
    PL2_TRACE(3);
    cf_drain_vm();
    PL2_TRACE(4);
    __syncthreads();
    PL2_TRACE(5);
#ifdef CF_PL2_STAMPS
    const unsigned long long t_loop = wall_clock64();
#endif
    uint32_t* last = (uint32_t*)cf_lds;
    if (threadIdx.x == 0) *last = atomicAdd(&S.C.ctl[3], 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    PL2_TRACE(6);
    if (!*last) return;
    __syncthreads();
#ifdef CF_PL2_STAMPS
    if (threadIdx.x == 0) { atomicAdd(&S.stamps[1], t_loop - t_last); atomicAdd(&S.stamps[8], t_win - t_last); atomicAdd(&S.stamps[9], t_body - t_win); atomicAdd(&S.stamps[10], t_loop - t_body); t_last = t_loop; atomicAdd(&S.stamps[0], 1ull); }
    PL2_STAMP(2);
    pl2_tail(S, it + 1u, w.read, t_last);
#else
    pl2_tail(S, it + 1u, w.read);
#endif
}

// ---- set-up kernels
__global__ void __launch_bounds__(256)
cf_pl2_ent_kernel(const int64_t* __restrict__ unit_ptr, const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, int64_t n_reads,
                  uint2* __restrict__ ent, int64_t* __restrict__ read_e) {      // one wave per read
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r <= n_reads; r += n_waves) {
        const int64_t u0 = unit_ptr[min(r, n_reads)], u1 = r < n_reads ? unit_ptr[r + 1] : u0;
        if (lane == 0) read_e[r] = cloud_ptr[u0];
        for (int64_t u = u0; u < u1; ++u)
            for (int64_t e = cloud_ptr[u] + lane; e < cloud_ptr[u + 1]; e += 64) ent[e] = make_uint2((uint32_t)entries[e], (uint32_t)(u - u0));
    }
}

// lay read r at `pos` (prefix reads, read_placer.py:35-40): 16 lanes do nothing useful here, one lane per entry
__global__ void __launch_bounds__(256)
cf_pl2_add_kernel(cf_pl2 S, int64_t r, uint32_t pos) {
    const int64_t e0 = S.read_e[r], e1 = S.read_e[r + 1];
    for (int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (int64_t)gridDim.x * blockDim.x) {
        const uint2 en = S.ent[e];
        (void)pl2_contig_add(S, en.x, pos + en.y, pl2_crec_load(S, en.x));
    }
}

// postings of one class of reads: mode 0 counts, mode 1 fills (read << 32 | unit index)
__global__ void __launch_bounds__(256)
cf_pl2_post_kernel(cf_pl2 S, const uint8_t* __restrict__ cls, int want_cls, int64_t n_reads, int mode, uint32_t* __restrict__ cnt, unsigned long long* __restrict__ post) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_reads; r += n_waves) {
        if (cls[r] != want_cls) continue;
        for (int64_t e = S.read_e[r] + lane; e < S.read_e[r + 1]; e += 64) {
            const uint2 en = S.ent[e];
            if (mode == 0) atomicAdd(&cnt[en.x], 1u);
            else post[S.C.post_ptr[en.x] + atomicAdd(&cnt[en.x], 1u)] = ((unsigned long long)(uint32_t)r << 32) | en.y;
        }
    }
}

__global__ void __launch_bounds__(256)
cf_pl2_prow_kernel(const int64_t* __restrict__ post_ptr, const unsigned long long* __restrict__ post, int64_t n_kmers, uint32_t ib, int pw, uint32_t* __restrict__ prow,
                   uint32_t* __restrict__ max_post) {
    const int sub = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, n_grp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int ppl = pw / 16;
    uint32_t longest = 0;
    for (int64_t x = grp; x < n_kmers; x += n_grp) {
        const int64_t p0 = post_ptr[x], n = post_ptr[x + 1] - p0;
        longest = max(longest, (uint32_t)min(n, (int64_t)0xFFFFFFFFll));
        if (!prow) continue;
        for (int k = 0; k < ppl; ++k) {
            const int64_t p = (int64_t)ppl * sub + k;
            uint32_t w = 0u;
            if (p < n && p < pw - 1) { const unsigned long long pr = post[p0 + p]; w = ((uint32_t)(pr >> 32) << ib) | (uint32_t)pr; }
            if (p == pw - 1) w = (uint32_t)n;
            prow[x * pw + p] = w;
        }
    }
    if (max_post && longest) atomicMax(max_post, longest);
}

// seed of a stage (read_placer.py:54-57): every (k-mer, position) of the contig whose k-mer is frequent anywhere is applied once
__global__ void __launch_bounds__(256)
cf_pl2_seed_kernel(cf_pl2 S, int64_t n_kmers) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t s = t0; s < (uint64_t)n_kmers * 4ull; s += stride) {      // the k-mer records, one lane per position word
        const uint32_t x = (uint32_t)(s >> 2);
        const unsigned long long k = S.crec[x].s[s & 3];
        if (!k || !S.C.freq_flag[x]) continue;
        const uint32_t q = (uint32_t)(k >> 32) - 1u;
        for (int64_t p = S.C.post_ptr[x]; p < S.C.post_ptr[x + 1]; ++p) {
            const unsigned long long pr = S.C.post_ri[p];
            const uint32_t r = (uint32_t)(pr >> 32), i = (uint32_t)pr;
            if (q >= i) pl2_hit(S, true, r, i, q - i, nullptr);
        }
    }
    for (uint64_t s = t0; s <= S.C.cmask; s += stride) {      // k-mers with more than four positions
        const unsigned long long k = S.C.ckeys[s];
        if (!k) continue;
        const uint32_t x = (uint32_t)k, q = (uint32_t)((k & ~CF_OCC) >> 32);
        if (!S.C.freq_flag[x]) continue;
        for (int64_t p = S.C.post_ptr[x]; p < S.C.post_ptr[x + 1]; ++p) {
            const unsigned long long pr = S.C.post_ri[p];
            const uint32_t r = (uint32_t)(pr >> 32), i = (uint32_t)pr;
            if (q >= i) pl2_hit(S, true, r, i, q - i, nullptr);
        }
    }
}

// One attempt with regions of `slots_per_unit` x units slots per read; returns 1 when a region or the contig map overflowed.
static int pl2_attempt(cf_ctx* ctx, const uint8_t* cls, const int32_t* id_rank, int32_t min_freq, int32_t min_unit, int32_t min_inters, int32_t min_prop,
                       int slots_per_unit, int cmap_grow, std::vector<int64_t>& o_read, std::vector<int64_t>& o_pos, std::vector<int32_t>& o_s0, std::vector<int32_t>& o_s1) {
    const int64_t R = ctx->n_reads, N = ctx->n_entries, K = ctx->n_kmers;
    const std::vector<int64_t>& up = ctx->h_unit_ptr;
    Bufs B{ctx, {}};
    cf_pl2 S;
    std::memset(&S, 0, sizeof S);
    S.C.thr = (uint32_t)std::max(1, min_freq); S.C.min_unit = (uint32_t)std::max(0, min_unit);
    S.C.min_inters = (uint32_t)std::max(0, min_inters); S.C.min_prop = (uint32_t)std::max(0, min_prop);
    S.hot_thr = std::max(1u, S.C.min_inters);
    S.n_reads = (uint32_t)R; S.n2 = (uint32_t)((R + 63) / 64); S.n_dirty_words = (uint32_t)((R + 31) / 32);
    int64_t max_u = 1;
    for (int64_t r = 0; r < R; ++r) max_u = std::max(max_u, up[(size_t)r + 1] - up[(size_t)r]);
    S.ib = 1; while ((1ll << S.ib) < max_u) ++S.ib;
    uint2* d_ent = nullptr; int64_t* d_read_e = nullptr; uint8_t* d_cls = nullptr; uint32_t* d_pcnt = nullptr; int64_t* d_post_ptr = nullptr;
    unsigned long long* d_post = nullptr; uint32_t* d_prow = nullptr; int32_t* d_rank = nullptr;
    // only the fifth and later positions of a k-mer land here; `cmap_grow`: times four per attempt that filled it (k-mers that are NOT
    // unique to one place of the array — small read sets, thin coverage — have many positions)
    const uint64_t ccap = (ctx->place_cmap_bits > 0 ? 1ull << ctx->place_cmap_bits : cf_pow2_ceil((uint64_t)std::max<int64_t>(N / 8, 1 << 21))) << (2 * cmap_grow);      // (2^21 slots = 25 MB: small read sets, whose k-mers are the least unique, start with room)
    {
        size_t free_b = 0, total_b = 0;
        if (ccap >= (1ull << 33) || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && ccap * 12ull > (unsigned long long)free_b + (unsigned long long)ctx->pooled))
            return cf_fail(ctx, -34, "cf_place_reads: the contig's overflow map would not fit the device");
    }
    CF_TRY(B.get(&d_ent, (size_t)N + 1, "entry records"));
    CF_TRY(B.get(&d_read_e, (size_t)R + 2, "read entry offsets"));
    CF_TRY(B.get(&d_cls, (size_t)R + 1, "classes"));
    CF_TRY(B.get(&d_rank, (size_t)R + 1, "id ranks"));
    CF_TRY(B.get(&S.C.used, (size_t)R + 1, "used flags"));
    CF_TRY(B.get(&S.C.ckeys, (size_t)ccap, "contig keys"));
    CF_TRY(B.get(&S.C.ccnt, (size_t)ccap, "contig counts"));
    CF_TRY(B.get(&S.crec, (size_t)K + 1, "contig records by k-mer"));
    CF_TRY(B.get(&S.C.freq_flag, (size_t)K + 1, "frequent flags"));
    CF_TRY(B.get(&d_pcnt, (size_t)K + 1, "stage posting counts"));
    CF_TRY(B.get(&d_post_ptr, (size_t)K + 2, "stage posting offsets"));
    CF_TRY(B.get(&d_post, (size_t)N + 1, "stage postings"));
    uint32_t* d_maxp = nullptr;
    CF_TRY(B.get(&d_maxp, 4, "longest posting list"));
    CF_TRY(B.get(&S.rinfo, (size_t)R + 1, "read regions"));
    CF_TRY(B.get(&S.dirty, (size_t)S.n_dirty_words + 1, "dirty bits"));
    CF_TRY(B.get(&S.RB, (size_t)S.n2 * 64 + 64, "read candidates"));
    CF_TRY(B.get(&S.L2, (size_t)S.n2 + 1, "candidates per 64 reads"));
    {
        // the third level (VERDICT round 4 asked for it for read sets of more than 10^5 reads): only when the tail's per-wave lists and the
        // two-records-per-lane pass hold its groups
        const int blk = ctx->place_block > 0 ? ctx->place_block : PL2_B;
        const uint32_t nw_ = (uint32_t)blk / 64u;
        const uint32_t nfw_ = std::max(1u, std::min(S.n2 > 2048u ? nw_ - nw_ / 4u : nw_ / 2u, (S.n2 + 255u) >> 8));
        S.g3s = ctx->place_l3_shift > 0 ? (uint32_t)ctx->place_l3_shift : 6u;
        uint32_t n3 = (S.n2 + (1u << S.g3s) - 1u) >> S.g3s;
        // measured (profiles/r05_place_l3.log, r05_place_phases.log): at 500 000 reads 10.69 - 10.81 s without the level, 10.77 - 11.13 s with
        // it — the sweep over all blocks is not what makes the tail grow with the read set (every dependent step gets slower against the
        // larger arrays: dirty list + 1.1 us, rescans + 1.3, blocks + 2.1, publish + 0.4 per iteration), and the level's own reductions and
        // lists cost what its fewer loads save.  So it is OFF unless asked for (place_l3 = 1); the tests keep it exact.
        const bool want = ctx->place_l3 == 1;
        if (!want || n3 > PL2_G3MAX || n3 > 128u * nfw_) n3 = 0;
        S.n3 = n3;
        CF_TRY(B.get(&S.L3, (size_t)n3 + 1, "candidates per group of blocks"));
        CF_TRY(B.get(&S.l3_stale, (size_t)(PL2_G3MAX / 32), "out-of-date groups"));
    }
    CF_TRY(B.get(&S.win, 1, "winner"));
    CF_TRY(B.get(&S.C.ctl, 8, "control"));
    CF_TRY(B.get(&S.stamps, 16, "phase stamps"));
    CF_TRY(B.get(&S.trace, 4096 * 16 * 8, "wave trace"));
    CF_TRY(B.get(&S.ttrace, 16 * 8, "tail trace"));
    S.trace_iter = std::getenv("CF_PL2_TRACE") ? (uint32_t)std::atoll(std::getenv("CF_PL2_TRACE")) : 0xFFFFFFFFu;
    CF_TRY(B.get(&S.C.out_read, (size_t)R + 1, "out_read"));
    CF_TRY(B.get(&S.C.out_pos, (size_t)R + 1, "out_pos"));
    CF_TRY(B.get(&S.C.out_s0, (size_t)R + 1, "out_s0"));
    CF_TRY(B.get(&S.C.out_s1, (size_t)R + 1, "out_s1"));
    S.C.cmask = ccap - 1; S.C.id_rank = d_rank; S.C.post_ptr = d_post_ptr; S.C.post_ri = d_post;
    S.ent = d_ent; S.read_e = d_read_e;
    hipStream_t st = ctx->stream;
    const int n_blocks = std::max(1, ctx->n_cu) * 4;
    CF_HIP(hipMemcpyAsync(d_cls, cls, (size_t)R, hipMemcpyHostToDevice, st));
    CF_HIP(hipMemcpyAsync(d_rank, id_rank, (size_t)R * 4, hipMemcpyHostToDevice, st));
    CF_HIP(hipMemsetAsync(S.C.used, 0, (size_t)R + 1, st));
    CF_HIP(hipMemsetAsync(S.C.ckeys, 0, (size_t)ccap * 8, st));
    CF_HIP(hipMemsetAsync(S.C.ccnt, 0, (size_t)ccap * 4, st));
    CF_HIP(hipMemsetAsync(S.crec, 0, ((size_t)K + 1) * sizeof(cf_pl2_crec), st));
    CF_HIP(hipMemsetAsync(S.C.freq_flag, 0, (size_t)K + 1, st));
    CF_HIP(hipMemsetAsync(S.C.ctl, 0, 32, st));
    CF_HIP(hipMemsetAsync(S.stamps, 0, 128, st));
    CF_HIP(hipMemsetAsync(S.trace, 0, (size_t)4096 * 16 * 8 * 8, st));
    CF_HIP(hipMemsetAsync(S.ttrace, 0, 16 * 8 * 8, st));
    hipLaunchKernelGGL(cf_pl2_ent_kernel, dim3((unsigned)cf_grid_for((R + 1) * 64, 256, n_blocks)), dim3(256), 0, st, (const int64_t*)ctx->d_unit_ptr,
                       (const int64_t*)ctx->d_cloud_ptr, (const int32_t*)ctx->d_entries, R, d_ent, d_read_e);
    CF_KERNEL_CHECK("cf_pl2_ent_kernel");
    o_read.clear(); o_pos.clear(); o_s0.clear(); o_s1.clear();
    for (int64_t r = 0; r < R; ++r) {      // prefix reads at position 0, in record order (read_placer.py:35-40)
        if (cls[r] != 0) continue;
        hipLaunchKernelGGL(cf_pl2_add_kernel, dim3(8), dim3(256), 0, st, S, r, 0u);
        o_read.push_back(r); o_pos.push_back(0); o_s0.push_back(-1); o_s1.push_back(-1);
    }
    CF_KERNEL_CHECK("cf_pl2_add_kernel");
    const size_t lds = pl2_lds_bytes(S.n2);
    // (one posting per lane: a read of 1 500 entries with 64-word rows is 1 500 waves = 94 workgroups; 96 / 128 / 192 measured 16.5 / 15.8 / 15.8 us per iteration)
    const int grid = ctx->place_grid > 0 ? ctx->place_grid : 128;
    const int block = ctx->place_block > 0 ? ctx->place_block : PL2_B;
    std::vector<cf_pl2_rinfo> h_ri((size_t)R + 1);
    for (int stage_cls = 1; stage_cls <= 2; ++stage_cls) {
        std::vector<int64_t> stage_reads;
        for (int64_t r = 0; r < R; ++r) if (cls[r] == stage_cls) stage_reads.push_back(r);
        if (stage_reads.empty()) continue;
        // postings of the stage: CSR lists, then the packed rows
        CF_HIP(hipMemsetAsync(d_pcnt, 0, (size_t)(K + 1) * 4, st));
        const int g_reads = cf_grid_for(std::max<int64_t>(R, 1) * 64, 256, n_blocks);
        hipLaunchKernelGGL(cf_pl2_post_kernel, dim3((unsigned)g_reads), dim3(256), 0, st, S, (const uint8_t*)d_cls, stage_cls, R, 0, d_pcnt, (unsigned long long*)nullptr);
        int64_t n_post = 0;
        CF_TRY(cf_scan_exclusive_u32_to_i64(ctx, d_pcnt, d_post_ptr, K + 1, &n_post));
        CF_HIP(hipMemsetAsync(d_pcnt, 0, (size_t)(K + 1) * 4, st));
        hipLaunchKernelGGL(cf_pl2_post_kernel, dim3((unsigned)g_reads), dim3(256), 0, st, S, (const uint8_t*)d_cls, stage_cls, R, 1, d_pcnt, d_post);
        // posting rows: as wide as the longest posting list needs (32 words hold 31 postings, 64 words 63; beyond: the CSR arrays)
        Bufs PB{ctx, {}};
        int pw = 32;
        if (K) {
            const unsigned g_k = (unsigned)cf_grid_for(K * 16, 256, n_blocks * 4);
            unsigned int h_maxp = 0;
            CF_HIP(hipMemsetAsync(d_maxp, 0, 4, st));
            hipLaunchKernelGGL(cf_pl2_prow_kernel, dim3(g_k), dim3(256), 0, st, (const int64_t*)d_post_ptr, (const unsigned long long*)d_post, K, S.ib, 32, (uint32_t*)nullptr, d_maxp);
            CF_HIP(hipMemcpyAsync(&h_maxp, d_maxp, 4, hipMemcpyDeviceToHost, st));
            CF_HIP(hipStreamSynchronize(st));
            pw = ctx->place_row_words ? ctx->place_row_words : (h_maxp > 31 ? 64 : 32);
            CF_TRY(PB.get(&d_prow, (size_t)(K + 1) * (size_t)pw, "posting rows"));
            S.prow = d_prow;
            hipLaunchKernelGGL(cf_pl2_prow_kernel, dim3(g_k), dim3(256), 0, st, (const int64_t*)d_post_ptr, (const unsigned long long*)d_post, K, S.ib, pw, d_prow, (uint32_t*)nullptr);
            if (std::getenv("CF_DEBUG")) std::fprintf(stderr, "[cf_place2] stage %d: longest posting list %u -> rows of %d words\n", stage_cls, h_maxp, pw);
        }
        CF_KERNEL_CHECK("placement postings");
        // regions of this stage's reads, then the seed.  The seed changes nothing but the scores: when it fills a region (a
        // late stage's few reads meet every position of a long contig) only this step is repeated with larger regions.
        Bufs SB{ctx, {}};      // the stage's score memory
        unsigned int h_ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int spu = slots_per_unit, tries = 0;; spu *= 4, ++tries) {
            if (tries == 6) return cf_fail(ctx, -34, "cf_place_reads: the seed of a stage kept overflowing the score regions");
            unsigned long long n_slots = 0, n_cells = 0;
            std::memset(h_ri.data(), 0, h_ri.size() * sizeof(cf_pl2_rinfo));
            for (int64_t r : stage_reads) {
                const uint64_t U = (uint64_t)(up[(size_t)r + 1] - up[(size_t)r]);
                if (!U) continue;
                const uint64_t H = cf_pow2_ceil(std::max<uint64_t>(64, U * (uint64_t)spu));
                cf_pl2_rinfo& q = h_ri[(size_t)r];
                const uint64_t Uw = ((U + 1) / 2 + 3) / 4 * 4;      // two 16-bit cells per word, rows of whole 16-byte pieces
                q.slot_base = n_slots; q.cell_base = n_cells; q.hmask = (uint32_t)(H - 1); q.Uw = (uint32_t)Uw;
                n_slots += H; n_cells += H * Uw;
            }
            SB.release_all();
            //  sizes are known before anything is allocated: regions that would need more than 2^32 slots or more
            // memory than the device has left give up with -34 — which the caller answers with the hash-map path — instead of
            // running into a failed hipMalloc (-12) after a x4 growth
            if (n_slots >= (1ull << 32)) return cf_fail(ctx, -34, "cf_place_reads: more than 2^32 score slots");
            {
                const unsigned long long need = n_slots * 8ull + n_cells * 4ull + n_slots / 8ull + 4096ull;
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (unsigned long long)free_b + (unsigned long long)ctx->pooled)
                    return cf_fail(ctx, -34, "cf_place_reads: the score regions would need " + std::to_string(need >> 20) + " MiB, more than the device has free");
            }
            CF_TRY(SB.get(&S.hdr, (size_t)n_slots + 1, "score headers"));
            CF_TRY(SB.get(&S.cells, (size_t)n_cells + 4, "score cells"));
            CF_TRY(SB.get(&S.hotbits, (size_t)n_slots / 32 + 16, "hot-row bits"));
            CF_HIP(hipMemcpyAsync(S.rinfo, h_ri.data(), (size_t)R * sizeof(cf_pl2_rinfo), hipMemcpyHostToDevice, st));
            CF_HIP(hipMemsetAsync(S.hdr, 0, (size_t)n_slots * 8, st));
            CF_HIP(hipMemsetAsync(S.cells, 0, (size_t)n_cells * 4, st));
            CF_HIP(hipMemsetAsync(S.hotbits, 0, ((size_t)n_slots / 32 + 16) * 4, st));
            CF_HIP(hipMemsetAsync(S.dirty, 0, (size_t)S.n_dirty_words * 4, st));
            CF_HIP(hipMemsetAsync(S.RB, 0, (size_t)S.n2 * 64 * sizeof(cf_pl2_rec), st));
            CF_HIP(hipMemsetAsync(S.L2, 0, (size_t)S.n2 * sizeof(cf_pl2_rec), st));
            CF_HIP(hipMemsetAsync(S.L3, 0, ((size_t)S.n3 + 1) * sizeof(cf_pl2_rec), st));
            CF_HIP(hipMemsetAsync(S.l3_stale, 0xFF, (size_t)(PL2_G3MAX / 32) * 4, st));      // every group out of date: the first tails sweep block by block and mend it
            CF_HIP(hipMemsetAsync(S.win, 0, sizeof(cf_pl2_rec), st));
            CF_HIP(hipMemsetAsync(S.C.ctl, 0, 8, st));         // done = 0, n_out = 0 (error flags kept)
            CF_HIP(hipMemsetAsync(S.C.ctl + 3, 0, 4, st));
            CF_HIP(hipMemsetAsync(S.C.ctl + 6, 0, 4, st));     // long rescans of this stage
            hipLaunchKernelGGL(cf_pl2_seed_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, S, K);
            CF_KERNEL_CHECK("cf_pl2_seed_kernel");
            CF_HIP(hipMemcpyAsync(h_ctl, S.C.ctl, 16, hipMemcpyDeviceToHost, st));
            CF_HIP(hipStreamSynchronize(st));
            if (std::getenv("CF_DEBUG")) std::fprintf(stderr, "[cf_place2] stage %d: %lld reads, %llu slots, %llu cells, %d slots per unit, flags after the seed %u\n", stage_cls, (long long)stage_reads.size(), n_slots, n_cells, spu, h_ctl[2]);
            if (h_ctl[2] & 1u) return 2;      // the contig's overflow map: another attempt with a larger one
            if (!(h_ctl[2] & 2u)) break;
            CF_HIP(hipMemsetAsync(S.C.ctl + 2, 0, 4, st));
        }
        hipLaunchKernelGGL(cf_pl2_tail_kernel, dim3(1), dim3(PL2_B), lds, st, S);
        CF_KERNEL_CHECK("cf_pl2_tail_kernel");
        const int64_t n_iter = (int64_t)stage_reads.size();
        unsigned long long long_seen = 0, iters_seen = 0;
        for (int64_t it = 0; it < n_iter; ++it) {
            if (pw == 32) hipLaunchKernelGGL(cf_pl2_iter_kernel<32>, dim3((unsigned)grid), dim3((unsigned)block), lds, st, S, (uint32_t)it);
            else hipLaunchKernelGGL(cf_pl2_iter_kernel<64>, dim3((unsigned)grid), dim3((unsigned)block), lds, st, S, (uint32_t)it);
            if ((it & 511) == 511 || it == 63 || it + 1 == n_iter) {
                CF_KERNEL_CHECK("placement iteration");
                CF_HIP(hipMemcpyAsync(h_ctl, S.C.ctl, 32, hipMemcpyDeviceToHost, st));
                CF_HIP(hipStreamSynchronize(st));
                // Round 5 (tools/fuzz_place.py, thin coverage: 15 000 reads over 45 000 units): where the k-mers are NOT unique to one place of
                // the array a read meets the contig at many offsets, many of its score rows are candidate rows, and the lane that rescans
                // the read walks them one by one — 254 us of a 270 us iteration, against 39 us per iteration on the hash-map path.  More than
                // two such rescans per iteration so far: this path gives the run up (place_mode 3 keeps it)
                // (counted per window between two looks: the contig of a stage's first iterations is short, the rows come later)
                const unsigned long long win_long = (unsigned long long)h_ctl[6] - long_seen, win_iters = (unsigned long long)(it + 1) - iters_seen;
                long_seen = h_ctl[6]; iters_seen = (unsigned long long)(it + 1);
                if (ctx->place_mode != 3 && it + 1 < n_iter && (ctx->place_long_rescans < 0 || win_long > (unsigned long long)ctx->place_long_rescans * win_iters))
                    return cf_fail(ctx, -34, "cf_place_reads: reads with many candidate score rows (" + std::to_string(win_long) + " long rescans in the last " + std::to_string(win_iters) + " of " + std::to_string(it + 1) + " iterations)");
                if (std::getenv("CF_DEBUG") && ((it & 8191) == 8191 || h_ctl[2] || h_ctl[0])) std::fprintf(stderr, "[cf_place2] stage %d iter %lld/%lld ctl=%u,%u,%u\n", stage_cls, (long long)it, (long long)n_iter, h_ctl[0], h_ctl[1], h_ctl[2]);
                if (h_ctl[2]) return (h_ctl[2] & 1u) ? 2 : 1;
                if (h_ctl[0]) break;
            }
        }
        const unsigned int n_out = h_ctl[1];
        std::vector<int64_t> t_read(n_out), t_pos(n_out);
        std::vector<int32_t> t_s0(n_out), t_s1(n_out);
        if (n_out) {
            CF_HIP(hipMemcpy(t_read.data(), S.C.out_read, (size_t)n_out * 8, hipMemcpyDeviceToHost));
            CF_HIP(hipMemcpy(t_pos.data(), S.C.out_pos, (size_t)n_out * 8, hipMemcpyDeviceToHost));
            CF_HIP(hipMemcpy(t_s0.data(), S.C.out_s0, (size_t)n_out * 4, hipMemcpyDeviceToHost));
            CF_HIP(hipMemcpy(t_s1.data(), S.C.out_s1, (size_t)n_out * 4, hipMemcpyDeviceToHost));
        }
        std::vector<uint8_t> placed((size_t)R, 0);
        for (unsigned int i = 0; i < n_out; ++i) {
            o_read.push_back(t_read[i]); o_pos.push_back(t_pos[i]); o_s0.push_back(t_s0[i]); o_s1.push_back(t_s1[i]);
            placed[(size_t)t_read[i]] = 1;
        }
        std::vector<int64_t> rest;      // None tail of the stage, ordered by read id (the reference's order is set-iteration order)
        for (int64_t r : stage_reads) if (!placed[(size_t)r]) rest.push_back(r);
        std::sort(rest.begin(), rest.end(), [&](int64_t a, int64_t b) { return id_rank[a] < id_rank[b]; });
        for (int64_t r : rest) { o_read.push_back(r); o_pos.push_back(-1); o_s0.push_back(-1); o_s1.push_back(-1); }
    }
    unsigned int h_ctl[4] = {0, 0, 0, 0};
    CF_HIP(hipMemcpy(h_ctl, S.C.ctl, 16, hipMemcpyDeviceToHost));
#ifdef CF_PL2_STAMPS
    {
        unsigned long long h_st[16];
        CF_HIP(hipMemcpy(h_st, S.stamps, 128, hipMemcpyDeviceToHost));
        const char* nm[8] = {"iterations", "lay down + events", "arrive", "dirty list", "rescans", "blocks", "block reduce", "publish"};
        std::fprintf(stderr, "[cf_place2 stamps] %llu iterations; us per iteration (100 MHz clock):", h_st[0]);
        for (int i = 1; i < 8; ++i) std::fprintf(stderr, "  %s %.2f", nm[i], h_st[0] ? (double)h_st[i] / 100.0 / (double)h_st[0] : 0.0);
        std::fprintf(stderr, "\n   of the lay-down phase (thread 0 of the last workgroup): winner record %.2f, its entries %.2f, drain + barrier %.2f\n",
                     h_st[0] ? (double)h_st[8] / 100.0 / (double)h_st[0] : 0.0, h_st[0] ? (double)h_st[9] / 100.0 / (double)h_st[0] : 0.0, h_st[0] ? (double)h_st[10] / 100.0 / (double)h_st[0] : 0.0);
#ifdef CF_PL2_STAMPS2
        if (S.trace_iter != 0xFFFFFFFFu) {
            std::vector<unsigned long long> tr((size_t)grid * 16 * 8);
            CF_HIP(hipMemcpy(tr.data(), S.trace, tr.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < tr.size(); i += 8) if (tr[i] && tr[i] < t0) t0 = tr[i];
            std::fprintf(stderr, "[cf_place2 trace] iteration %u: per wave, us since the first wave's entry: entry, winner, tags+barrier, loop, drain, barrier, arrive\n", S.trace_iter);
            for (int wg = 0; wg < grid; ++wg)
                for (int wv = 0; wv < 16; ++wv) {
                    const unsigned long long* t = &tr[((size_t)wg * 16 + wv) * 8];
                    if (!t[0]) continue;
                    std::fprintf(stderr, "  wg %3d wave %2d:", wg, wv);
                    for (int k = 0; k < 7; ++k) std::fprintf(stderr, " %6.2f", t[k] ? (double)(t[k] - t0) / 100.0 : -1.0);
                    std::fprintf(stderr, "\n");
                }
            unsigned long long tt[16 * 8];
            CF_HIP(hipMemcpy(tt, S.ttrace, sizeof tt, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[cf_place2 tail trace] per wave, us since the first wave's entry of the launch: tail entry, dirty list done, barrier, rescans done, barrier, blocks + sweep done, barrier, end\n");
            for (int wv = 0; wv < 16; ++wv) {
                if (!tt[wv * 8]) continue;
                std::fprintf(stderr, "  tail wave %2d:", wv);
                for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %6.2f", tt[wv * 8 + k] ? (double)(tt[wv * 8 + k] - t0) / 100.0 : -1.0);
                std::fprintf(stderr, "\n");
            }
        }
#endif
        if (h_st[14]) std::fprintf(stderr, "   one round trip at a time (first wave of every workgroup, %llu samples): entry record %.2f us, posting row + contig record %.2f, contig claim / add %.2f\n",
                                   h_st[14], (double)h_st[11] / 100.0 / (double)h_st[14], (double)h_st[12] / 100.0 / (double)h_st[14], (double)h_st[13] / 100.0 / (double)h_st[14]);
    }
#endif
    if (h_ctl[2]) return (h_ctl[2] & 1u) ? 2 : 1;
    return 0;
}

// can this read set take the round-4 path?  (packed postings [read | unit index] in 32 bits, entry offsets in 40 + 24 bits)
bool cf_place2_fits(const cf_ctx* ctx) {
    int64_t max_u = 1;
    for (int64_t r = 0; r < ctx->n_reads; ++r) max_u = std::max(max_u, ctx->h_unit_ptr[(size_t)r + 1] - ctx->h_unit_ptr[(size_t)r]);
    int ib = 1; while ((1ll << ib) < max_u) ++ib;
    // the winner record packs a read's cloud-entry count into 24 bits (ext = e0 << 24 | n): a cloud has at most one entry per
    // window of its unit, so units x longest unit bounds the entries of any read (such a read — a sequence of
    // megabases, not a read — takes the hash-map path instead of being truncated)
    if (max_u * std::max<int64_t>(ctx->max_unit_len, 1) >= (1ll << 24)) return false;
    // (16-bit cells: a cell counts the events of one unit's k-mers at one offset, at most two per k-mer — seed and threshold)
    return ib < 32 && 2 * ctx->max_unit_len < 65536 && ctx->n_reads <= (1ll << (32 - ib)) && ctx->n_entries < (1ll << 40) && (int64_t)ctx->h_unit_ptr.size() == ctx->n_reads + 1;
}

int cf_place2_run(cf_ctx* ctx, const uint8_t* cls, const int32_t* id_rank, int32_t min_freq, int32_t min_unit, int32_t min_inters, int32_t min_prop,
                  std::vector<int64_t>& o_read, std::vector<int64_t>& o_pos, std::vector<int32_t>& o_s0, std::vector<int32_t>& o_s1) {
    // an attempt ends with 1 when a score region filled up (the next one gets four times the slots per unit) and with 2 when the contig's
    // overflow map did (round 5: that flag used to enlarge the REGIONS, six times over, and then give the run to the hash-map path)
    int spu = ctx->place_slots_per_unit > 0 ? ctx->place_slots_per_unit : 48;
    int rc = 1, cmap_grow = 0;
    for (int attempt = 0; attempt < 8 && (rc == 1 || rc == 2); ++attempt) {
        rc = pl2_attempt(ctx, cls, id_rank, min_freq, min_unit, min_inters, min_prop, spu, cmap_grow, o_read, o_pos, o_s0, o_s1);
        if (std::getenv("CF_DEBUG")) std::fprintf(stderr, "[cf_place2] attempt %d rc=%d slots per unit %d, contig map x 4^%d\n", attempt, rc, spu, cmap_grow);
        if (rc == 1) spu *= 4;
        if (rc == 2) ++cmap_grow;
    }
    if (rc == 1 || rc == 2) return cf_fail(ctx, -34, rc == 1 ? "cf_place_reads: score regions kept overflowing" : "cf_place_reads: the contig's overflow map kept overflowing");
    return rc;
}
