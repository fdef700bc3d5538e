// cf_place.hip — A8 + A9: cloud contig and greedy read placement, device resident.
//
// Reference:
//   scripts/cloud_contig.py:26-41  CloudContig.add_read: count[(pos, kmer)] += 1 for every k-mer of
//       every unit cloud; exactly when a count EQUALS max(1, min_cloud_kmer_freq) the pair
//       becomes "frequent" and is reported.
//   scripts/cloud_contig.py:87-95  update_mapping_scores: each reported (kmer, q) adds 1 to
//       scores[read][q - i][i] for every posting (read, i) of kmer with q >= i.
//   scripts/read_placer.py:35-94   prefix reads at position 0; per stage (internal, suffix):
//       postings over the stage's reads, seed = every (kmer, pos) with kmer frequent anywhere and
//       pos any position the k-mer was ever added at (:54-57, over-inclusive on purpose), then
//       repeatedly place the arg-max read over (s0 = #units hit, s1 = total hits, offset, then
//       smaller r_id) among entries with s0 >= min_unit, s0 * min_prop <= s1, s1 >= min_inters;
//       if none qualifies all remaining reads of the stage are written as None.
//
// Device design: everything lives in HBM hash tables — contig (pos,kmer)->count, score
// (read,offset)->(s0,s1), a seen-set of (score slot, unit) for s0.  One iteration = 3 small
// kernels enqueued back to back with NO host round trip: apply events -> block arg-max over
// cached slices -> pick the winner (recorded device-side) and add the chosen read (emits the
// next events).  The host only polls a done flag every few hundred iterations.
#include "cf_place.h"

#include <cstdlib>

// mode 0: add read `fixed_read` at position 0 (prefix reads); mode 1: add the read in S.best
__global__ void __launch_bounds__(PL_THREADS)
cf_place_add_kernel(cf_place_state S, int mode, int64_t fixed_read) {
    if (mode == 1 && (S.ctl[0] || !S.best->valid)) return;
    const int64_t r = mode == 0 ? fixed_read : (int64_t)S.best->read;
    const uint32_t p = mode == 0 ? 0u : S.best->off;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t u0 = S.unit_ptr[r], u1 = S.unit_ptr[r + 1];
    for (int64_t u = u0 + wave; u < u1; u += n_waves) {
        const uint32_t q = p + (uint32_t)(u - u0);
        for (int64_t e = S.cloud_ptr[u] + lane; e < S.cloud_ptr[u + 1]; e += 64) cf_contig_add(S, (uint32_t)S.entries[e], q);
    }
}

// seed of a stage: every (kmer, pos) of the contig whose k-mer is frequent anywhere
__global__ void __launch_bounds__(PL_THREADS)
cf_place_seed_kernel(cf_place_state S) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= S.cmask; i += stride) {
        const unsigned long long k = S.ckeys[i];
        if (!k) continue;
        const uint32_t x = (uint32_t)k, q = (uint32_t)((k & ~CF_OCC) >> 32);
        if (S.freq_flag[x]) {
            const unsigned long long p = atomicAdd(S.n_events, 1ull);
            S.events[p] = ((unsigned long long)x << 32) | q;
        }
    }
}

// one hit: k-mer frequent at contig position q, posting (read r, unit i) -> scores[r][q - i][i] += 1 (reference cloud_contig.py:90-94)
__device__ __forceinline__ void cf_score_hit(const cf_place_state& S, uint32_t r, uint32_t i, uint32_t q) {
    if (q < i) return;
    const uint32_t off = q - i;
    const unsigned long long want = (((unsigned long long)r << 32) | off) | CF_OCC;
    uint64_t h = cf_mix64(want) & S.smask;
    bool ok = false;
    for (uint64_t probe = 0; probe <= S.smask && probe < 4096; ++probe) {   // a long probe = table too full: grow and restart
        const unsigned long long cur = atomicCAS(&S.skeys[h], 0ull, want);
        if (cur == 0ull && atomicAdd(&S.ctl[4], 1u) > (unsigned int)(S.smask >> 1)) atomicOr(&S.ctl[2], 2u);   // load > 0.5
        if (cur == 0ull || cur == want) { ok = true; break; }
        h = (h + 1) & S.smask;
    }
    if (!ok) { atomicOr(&S.ctl[2], 2u); return; }
    // first hit of unit i at this (read, offset)?
    const unsigned long long sk = ((((unsigned long long)h) << 24) ^ ((unsigned long long)i)) | CF_OCC;  // slot < 2^38, i < 2^24
    uint64_t hs = cf_mix64(sk) & S.seen_mask;
    bool fresh = false, placed = false;
    for (uint64_t probe = 0; probe <= S.seen_mask && probe < 4096; ++probe) {
        const unsigned long long cur = atomicCAS(&S.seen[hs], 0ull, sk);
        if (cur == 0ull) { fresh = true; placed = true; break; }
        if (cur == sk) { placed = true; break; }
        hs = (hs + 1) & S.seen_mask;
    }
    if (!placed) atomicOr(&S.ctl[2], 4u);
    // s1 += 1, s0 += fresh in one add; what comes back plus the increment is the entry as this lane left it: the lane whose
    // add is the last one on the entry sees its final state, so a finally-qualifying entry is always flagged; flags can be
    // stale-true (the arg-max re-checks and clears them), never stale-false
    const unsigned long long inc = 1ull | ((unsigned long long)(fresh ? 1u : 0u) << 32);
    const unsigned long long was = atomicAdd(&S.s01[h], inc), v = was + inc;
    const uint32_t v0 = (uint32_t)(v >> 32), v1 = (uint32_t)v, w0 = (uint32_t)(was >> 32), w1 = (uint32_t)was;
    const bool now_q = v0 >= S.min_unit && (unsigned long long)v0 * S.min_prop <= v1 && v1 >= S.min_inters;
    const bool was_q = w0 >= S.min_unit && (unsigned long long)w0 * S.min_prop <= w1 && w1 >= S.min_inters;
    if (now_q) ((uint8_t*)S.qflag)[h] = 1;
    // the slice's cached candidate may be this entry with its OLD values: any change of an entry that qualified before or
    // qualifies now makes the cache stale (s0 * min_prop <= s1 is not monotone: a first hit in a new unit can take an entry out
    // of the candidates, and round 2 marked the slice only when the entry qualified AFTER the add)
    if (now_q || was_q) S.dirty[h >> S.slice_shift] = 1;
}

// apply the pending events to the scores: one wave per event, lanes over the k-mer's postings
__global__ void __launch_bounds__(PL_THREADS)
cf_place_update_kernel(cf_place_state S) {
    if (S.ctl[0]) return;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t n_ev = (int64_t)S.n_events[0];
    for (int64_t ev = wave; ev < n_ev; ev += n_waves) {
        const unsigned long long E = S.events[ev];
        const uint32_t x = (uint32_t)(E >> 32), q = (uint32_t)E;
        const int64_t p0 = S.post_ptr[x], p1 = S.post_ptr[x + 1];
        for (int64_t pp = p0 + lane; pp < p1; pp += 64) {
            const unsigned long long ri = S.post_ri[pp];
            cf_score_hit(S, (uint32_t)(ri >> 32), (uint32_t)ri, q);
        }
    }
}


__global__ void __launch_bounds__(PL_THREADS)
cf_place_argmax_kernel(cf_place_state S) {
    // Block b owns one contiguous slice of the score map and caches its best candidate in block_best[b].  The cache
    // stays valid until an entry of the slice changes (the update kernel marks the slice dirty) or the cached read gets
    // placed; only then is the slice scanned again — most slices are untouched by one greedy iteration.
    if (S.ctl[0]) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) S.n_events[0] = 0ull;   // consumed by the update kernel before this one; refilled by the next
    uint32_t* rescan = (uint32_t*)(cf_lds + 16 * sizeof(cf_cand));
    if (threadIdx.x == 0) {      // (a slice whose cached candidate is the read just placed was marked by the kernel that placed it)
        const bool r = S.dirty[blockIdx.x] != 0;
        if (r) S.dirty[blockIdx.x] = 0;
        *rescan = r ? 1u : 0u;
    }
    __syncthreads();
    if (!*rescan) return;
    cf_key mine{0ull, 0ull, 0u};
    {
        const uint64_t per = (1ull << S.slice_shift) >> 4;   // 16-byte quads (16 flag bytes) per block
        const uint64_t q0 = (uint64_t)blockIdx.x * per;
        struct alignas(16) quad { uint32_t w[4]; };
        for (uint64_t q = q0 + threadIdx.x; q < q0 + per; q += blockDim.x) {
            const quad fq = ((const quad*)S.qflag)[q];
            if (!(fq.w[0] | fq.w[1] | fq.w[2] | fq.w[3])) continue;
#pragma unroll
            for (int wi = 0; wi < 4; ++wi) {
                const uint32_t f = fq.w[wi];
                if (!f) continue;
                for (int j = 0; j < 4; ++j) {
                    if (!((f >> (8 * j)) & 0xFFu)) continue;
                    const uint64_t i = 16 * q + 4 * (uint64_t)wi + (uint64_t)j;
                    const unsigned long long k = S.skeys[i];
                    const uint32_t r = (uint32_t)((k & ~CF_OCC) >> 32), off = (uint32_t)k;
                    const unsigned long long v01 = S.s01[i];
                    const uint32_t v0 = (uint32_t)(v01 >> 32), v1 = (uint32_t)v01;
                    if (!(v0 >= S.min_unit && (unsigned long long)v0 * S.min_prop <= v1 && v1 >= S.min_inters)) { ((uint8_t*)S.qflag)[i] = 0; continue; }   // stale flag
                    if (S.used[r]) { ((uint8_t*)S.qflag)[i] = 0; continue; }   // a placed read is never a candidate again: drop its entry from later scans
                    cf_cand c; c.s0 = v0; c.s1 = v1; c.off = off; c.rank = (uint32_t)S.id_rank[r]; c.read = r; c.valid = 1;
                    cf_key_take(mine, cf_key_of(c));
                }
            }
        }
    }
    const cf_cand b = cf_block_best(mine);
    if (threadIdx.x == 0) S.block_best[blockIdx.x] = b;
}

// Every block reduces the block candidates to the same winner (a few KB, cheaper than one more kernel in the dependent
// chain of the greedy iteration); block 0 records the placement; then all blocks lay the read onto the contig, which
// emits the next events (the event list was reset by the arg-max kernel, after the update kernel had consumed it).
__global__ void __launch_bounds__(PL_THREADS)
cf_place_pick_add_kernel(cf_place_state S, int n_cand) {
    if (S.ctl[0]) return;
    cf_key mine{0ull, 0ull, 0u};
    for (int i = threadIdx.x; i < n_cand; i += blockDim.x) cf_key_take(mine, cf_key_of(S.block_best[i]));
    const cf_cand b = cf_block_best(mine);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *S.best = b;
        if (!b.valid) S.ctl[0] = 1;
        else {
            const unsigned int o = S.ctl[1]++;
            S.out_read[o] = (int64_t)b.read; S.out_pos[o] = (int64_t)b.off; S.out_s0[o] = (int32_t)b.s0; S.out_s1[o] = (int32_t)b.s1;
            S.used[b.read] = 1;
        }
    }
    if (!b.valid) return;
    if (blockIdx.x == gridDim.x - 1)      // (see cf_place_pick_add_update_kernel)
        for (int i = threadIdx.x; i < n_cand; i += blockDim.x) { const cf_cand c = S.block_best[i]; if (c.valid && c.read == b.read) S.dirty[i] = 1; }
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t u0 = S.unit_ptr[b.read], u1 = S.unit_ptr[b.read + 1];
    for (int64_t u = u0 + wave; u < u1; u += n_waves) {
        const uint32_t q = b.off + (uint32_t)(u - u0);
        for (int64_t e = S.cloud_ptr[u] + lane; e < S.cloud_ptr[u + 1]; e += 64) cf_contig_add(S, (uint32_t)S.entries[e], q);
    }
}

// pick + add + score updates in ONE kernel: the wave whose add makes a (k-mer, position) pair frequent applies that event
// to the scores itself (all 64 lanes over the k-mer's postings) instead of queueing it for a separate update kernel — one
// kernel boundary less in the dependent chain of a greedy iteration.  The read's cloud entries are one contiguous CSR
// range; a wave takes PL_CHUNK (ctx->place_chunk, default 2) of them at a time, so that the events of a read spread over a
// thousand waves: what a wave does per event is a chain of dependent HBM round trips, and the iteration takes as long as the
// wave with the most events (50 000 reads: 8 entries per wave and 128 workgroups 1.36 s, 2 and 256: 1.22 s; three kernels 1.61 s).
__global__ void __launch_bounds__(PL_THREADS)
cf_place_pick_add_update_kernel(cf_place_state S, int n_cand, int PL_CHUNK) {
    if (S.ctl[0]) return;
    cf_key mine{0ull, 0ull, 0u};
    for (int i = threadIdx.x; i < n_cand; i += blockDim.x) cf_key_take(mine, cf_key_of(S.block_best[i]));
    const cf_cand b = cf_block_best(mine);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *S.best = b;
        if (!b.valid) S.ctl[0] = 1;
        else {
            const unsigned int o = S.ctl[1]++;
            S.out_read[o] = (int64_t)b.read; S.out_pos[o] = (int64_t)b.off; S.out_s0[o] = (int32_t)b.s0; S.out_s1[o] = (int32_t)b.s1;
            S.used[b.read] = 1;
        }
    }
    if (!b.valid) return;
    // the slices whose cached candidate names the read placed now have to be scanned again (the arg-max kernel used to find
    // that out itself: its cached candidate, then used[] of that read — two dependent loads at the head of every block)
    if (blockIdx.x == gridDim.x - 1)
        for (int i = threadIdx.x; i < n_cand; i += blockDim.x) { const cf_cand c = S.block_best[i]; if (c.valid && c.read == b.read) S.dirty[i] = 1; }
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t u0 = S.unit_ptr[b.read], u1 = S.unit_ptr[b.read + 1];
    const int64_t e0 = S.cloud_ptr[u0], e1 = S.cloud_ptr[u1];
    const int64_t n_chunks = (e1 - e0 + PL_CHUNK - 1) / PL_CHUNK;
    for (int64_t c = wave; c < n_chunks; c += n_waves) {
        const int64_t e = e0 + c * PL_CHUNK + lane;
        const bool act = lane < PL_CHUNK && e < e1;
        uint32_t x = 0, q = 0;
        bool hit = false;
        if (act) {
            x = (uint32_t)S.entries[e];
            q = b.off + (uint32_t)((int64_t)S.entry_unit[e] - u0);      // (a table instead of a binary search over cloud_ptr: four dependent loads less in every wave's chain)
            hit = cf_contig_add_hit(S, x, q);
        }
        unsigned long long m = __ballot(hit);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const uint32_t xx = (uint32_t)__shfl((int)x, l), qq = (uint32_t)__shfl((int)q, l);
            const int64_t p0 = S.post_ptr[xx], p1 = S.post_ptr[xx + 1];
            for (int64_t pp = p0 + lane; pp < p1; pp += 64) {
                const unsigned long long ri = S.post_ri[pp];
                cf_score_hit(S, (uint32_t)(ri >> 32), (uint32_t)ri, qq);
            }
        }
    }
}


// (Also measured and not kept: the score-entry probe and the seen-set probe of a hit in lockstep — the seen key made of
// (read, offset, unit) instead of the score slot, so that both slot loads and both claims are in flight together: 1.203 vs
// 1.207 s per 50 000 reads, and it limits reads, units and units per read to 24, 24 and 15 bits.)
// (Round 2 tried the greedy loop of a stage as ONE persistent launch — flag scan per workgroup, grid barrier, pick + add +
// score updates by the waves that raise the events, grid barrier: 58 us per placed read against 32 us for the three
// kernels below at 50 000 reads (two agent-scope barriers under load cost more than three dependent launches whose
// grids fit their work), and entries updated by other XCDs' atomics were read stale through this XCD's L2 by the plain
// loads of the scan.  Removed; the kernel boundary is the cheapest correct grid-wide barrier this loop has.)

__global__ void __launch_bounds__(256)
cf_entry2unit_kernel(const int64_t* __restrict__ cloud_ptr, int64_t n_units, int32_t* __restrict__ e2u) {      // one wave per unit
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < n_units; u += n_waves)
        for (int64_t e = cloud_ptr[u] + lane; e < cloud_ptr[u + 1]; e += 64) e2u[e] = (int32_t)u;
}

__global__ void __launch_bounds__(256)
cf_unit2read_kernel(const int64_t* __restrict__ unit_ptr, int64_t n_reads, int32_t* __restrict__ u2r) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride)
        for (int64_t u = unit_ptr[r]; u < unit_ptr[r + 1]; ++u) u2r[u] = (int32_t)r;
}

// postings over the reads of one class: mode 0 histogram, mode 1 fill
__global__ void __launch_bounds__(256)
cf_place_post_kernel(const int64_t* __restrict__ cloud_ptr, const int32_t* __restrict__ entries, const int32_t* __restrict__ u2r,
                     const uint8_t* __restrict__ cls, int want_cls, int64_t n_units, int mode, uint32_t* __restrict__ cnt,
                     const int64_t* __restrict__ post_ptr, unsigned long long* __restrict__ post, const int64_t* __restrict__ unit_ptr) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < n_units; u += n_waves) {
        const int32_t r = u2r[u];
        if (cls[r] != want_cls) continue;
        const unsigned long long ri = mode == 0 ? 0ull : (((unsigned long long)(uint32_t)r << 32) | (unsigned long long)(uint32_t)(u - unit_ptr[r]));
        for (int64_t e = cloud_ptr[u] + lane; e < cloud_ptr[u + 1]; e += 64) {
            const int32_t x = entries[e];
            if (mode == 0) atomicAdd(&cnt[x], 1u);
            else post[post_ptr[x] + atomicAdd(&cnt[x], 1u)] = ri;
        }
    }
}


// One attempt with given table sizes; returns 1 if a table overflowed (caller retries larger).
static int place_attempt(cf_ctx* ctx, const uint8_t* cls, const int32_t* id_rank, int32_t min_freq, int32_t min_unit,
                         int32_t min_inters, int32_t min_prop, uint64_t score_cap, uint64_t seen_cap,
                         std::vector<int64_t>& o_read, std::vector<int64_t>& o_pos, std::vector<int32_t>& o_s0, std::vector<int32_t>& o_s1) {
    const int64_t R = ctx->n_reads, U = ctx->n_units, N = ctx->n_entries, K = ctx->n_kmers;
    Bufs B{ctx, {}};
    cf_place_state S;
    std::memset(&S, 0, sizeof S);
    S.unit_ptr = ctx->d_unit_ptr; S.cloud_ptr = ctx->d_cloud_ptr; S.entries = ctx->d_entries;
    S.thr = (uint32_t)std::max(1, min_freq); S.min_unit = (uint32_t)std::max(0, min_unit);
    S.min_inters = (uint32_t)std::max(0, min_inters); S.min_prop = (uint32_t)std::max(0, min_prop);
    int32_t *d_u2r = nullptr, *d_rank = nullptr, *d_e2u = nullptr;
    unsigned long long* d_post = nullptr;
    uint8_t *d_cls = nullptr, *d_used = nullptr;
    uint32_t* d_pcnt = nullptr;
    int64_t* d_post_ptr = nullptr;
    const uint64_t ccap = cf_pow2_ceil((uint64_t)std::max<int64_t>(2 * N, 1024));
    CF_TRY(B.get(&d_u2r, (size_t)U + 1, "unit2read"));
    CF_TRY(B.get(&d_e2u, (size_t)N + 1, "entry2unit"));
    CF_TRY(B.get(&d_cls, (size_t)R + 1, "classes"));
    CF_TRY(B.get(&d_used, (size_t)R + 1, "used flags"));
    CF_TRY(B.get(&d_rank, (size_t)R + 1, "id ranks"));
    CF_TRY(B.get(&S.ckeys, (size_t)ccap, "contig keys"));
    CF_TRY(B.get(&S.ccnt, (size_t)ccap, "contig counts"));
    CF_TRY(B.get(&S.freq_flag, (size_t)K + 1, "frequent flags"));
    CF_TRY(B.get(&d_pcnt, (size_t)K + 1, "stage posting counts"));
    CF_TRY(B.get(&d_post_ptr, (size_t)K + 1, "stage posting offsets"));
    CF_TRY(B.get(&d_post, (size_t)N + 1, "stage postings"));
    CF_TRY(B.get(&S.skeys, (size_t)score_cap, "score keys"));
    CF_TRY(B.get(&S.s01, (size_t)score_cap, "score s0 | s1"));
    CF_TRY(B.get(&S.qflag, (size_t)score_cap / 4 + 1, "score flags"));
    CF_TRY(B.get(&S.seen, (size_t)seen_cap, "seen set"));
    CF_TRY(B.get(&S.events, (size_t)N + 1, "events"));
    CF_TRY(B.get(&S.n_events, 2, "event count"));
    CF_TRY(B.get(&S.ctl, 8, "control"));
    const int n_blocks = std::max(1, ctx->n_cu) * 4;
    // arg-max blocks: a power of two, each owning a contiguous slice of at least 16 score slots
    int n_am = 1;
    // (1 024 slices at 256 CUs: every workgroup of the iteration kernel reduces all cached candidates first, and a dirty slice is
    // scanned by one block — 2 048 slices 1.288 s per 50 000 reads, 1 024: 1.229 s, 512: 1.302 s, 256: 1.536 s)
    while (2 * n_am <= n_blocks && (uint64_t)(2 * n_am) * 16 <= score_cap) n_am *= 2;
    S.slice_shift = 0;
    while ((score_cap >> S.slice_shift) > (uint64_t)n_am) ++S.slice_shift;
    CF_TRY(B.get(&S.block_best, (size_t)n_am, "block candidates"));
    CF_TRY(B.get(&S.dirty, (size_t)n_am + 16, "slice dirty flags"));
    CF_TRY(B.get(&S.best, 1, "best candidate"));
    CF_TRY(B.get(&S.out_read, (size_t)R + 1, "out_read"));
    CF_TRY(B.get(&S.out_pos, (size_t)R + 1, "out_pos"));
    CF_TRY(B.get(&S.out_s0, (size_t)R + 1, "out_s0"));
    CF_TRY(B.get(&S.out_s1, (size_t)R + 1, "out_s1"));
    S.unit2read = d_u2r; S.entry_unit = d_e2u; S.cmask = ccap - 1; S.smask = score_cap - 1; S.seen_mask = seen_cap - 1;
    S.used = d_used; S.id_rank = d_rank; S.post_ptr = d_post_ptr; S.post_ri = d_post;
    hipStream_t st = ctx->stream;
    CF_HIP(hipMemcpyAsync(d_cls, cls, (size_t)R, hipMemcpyHostToDevice, st));
    CF_HIP(hipMemcpyAsync(d_rank, id_rank, (size_t)R * 4, hipMemcpyHostToDevice, st));
    CF_HIP(hipMemsetAsync(d_used, 0, (size_t)R + 1, st));
    CF_HIP(hipMemsetAsync(S.ckeys, 0, (size_t)ccap * 8, st));
    CF_HIP(hipMemsetAsync(S.ccnt, 0, (size_t)ccap * 4, st));
    CF_HIP(hipMemsetAsync(S.freq_flag, 0, (size_t)K + 1, st));
    CF_HIP(hipMemsetAsync(S.n_events, 0, 16, st));
    CF_HIP(hipMemsetAsync(S.ctl, 0, 32, st));
    const int g_units = cf_grid_for(std::max<int64_t>(U, 1) * 64, 256, n_blocks);
    if (U) hipLaunchKernelGGL(cf_entry2unit_kernel, dim3((unsigned)g_units), dim3(256), 0, st, (const int64_t*)ctx->d_cloud_ptr, U, d_e2u);
    if (R) hipLaunchKernelGGL(cf_unit2read_kernel, dim3((unsigned)cf_grid_for(R, 256, n_blocks)), dim3(256), 0, st, (const int64_t*)ctx->d_unit_ptr, R, d_u2r);
    // prefix reads at position 0, in record order (reference read_placer.py:35-40)
    o_read.clear(); o_pos.clear(); o_s0.clear(); o_s1.clear();
    for (int64_t r = 0; r < R; ++r) {
        if (cls[r] != 0) continue;
        hipLaunchKernelGGL(cf_place_add_kernel, dim3(8), dim3(PL_THREADS), 0, st, S, 0, r);
        o_read.push_back(r); o_pos.push_back(0); o_s0.push_back(-1); o_s1.push_back(-1);
    }
    CF_KERNEL_CHECK("cf_place_add_kernel");
    for (int stage_cls = 1; stage_cls <= 2; ++stage_cls) {
        std::vector<int64_t> stage_reads;
        for (int64_t r = 0; r < R; ++r) if (cls[r] == stage_cls) stage_reads.push_back(r);
        if (stage_reads.empty()) continue;
        // postings of the stage
        CF_HIP(hipMemsetAsync(d_pcnt, 0, (size_t)(K + 1) * 4, st));
        hipLaunchKernelGGL(cf_place_post_kernel, dim3((unsigned)g_units), dim3(256), 0, st, (const int64_t*)ctx->d_cloud_ptr, (const int32_t*)ctx->d_entries,
                           (const int32_t*)d_u2r, (const uint8_t*)d_cls, stage_cls, U, 0, d_pcnt, (const int64_t*)nullptr, (unsigned long long*)nullptr, (const int64_t*)ctx->d_unit_ptr);
        int64_t n_post = 0;
        CF_TRY(cf_scan_exclusive_u32_to_i64(ctx, d_pcnt, d_post_ptr, K + 1, &n_post));
        CF_HIP(hipMemsetAsync(d_pcnt, 0, (size_t)(K + 1) * 4, st));
        hipLaunchKernelGGL(cf_place_post_kernel, dim3((unsigned)g_units), dim3(256), 0, st, (const int64_t*)ctx->d_cloud_ptr, (const int32_t*)ctx->d_entries,
                           (const int32_t*)d_u2r, (const uint8_t*)d_cls, stage_cls, U, 1, d_pcnt, (const int64_t*)d_post_ptr, d_post, (const int64_t*)ctx->d_unit_ptr);
        // fresh scores, seed events
        CF_HIP(hipMemsetAsync(S.skeys, 0, (size_t)score_cap * 8, st));
        CF_HIP(hipMemsetAsync(S.s01, 0, (size_t)score_cap * 8, st));
        CF_HIP(hipMemsetAsync(S.qflag, 0, (size_t)score_cap + 4, st));
        CF_HIP(hipMemsetAsync(S.dirty, 1, (size_t)n_am, st));
        CF_HIP(hipMemsetAsync(S.block_best, 0, (size_t)n_am * sizeof(cf_cand), st));
        CF_HIP(hipMemsetAsync(S.seen, 0, (size_t)seen_cap * 8, st));
        CF_HIP(hipMemsetAsync(S.n_events, 0, 16, st));
        CF_HIP(hipMemsetAsync(S.ctl, 0, 8, st));  // done = 0, n_out = 0 (error flags kept)
        CF_HIP(hipMemsetAsync(S.ctl + 4, 0, 4, st));  // score-map entry count of this stage
        hipLaunchKernelGGL(cf_place_seed_kernel, dim3((unsigned)n_blocks), dim3(PL_THREADS), 0, st, S);
        CF_KERNEL_CHECK("cf_place_seed_kernel");
        unsigned int h_ctl[4] = {0, 0, 0, 0};
        const int64_t n_iter = (int64_t)stage_reads.size();
        const bool fused = ctx->place_fused != 0;
        if (fused) hipLaunchKernelGGL(cf_place_update_kernel, dim3((unsigned)n_blocks), dim3(PL_THREADS), 0, st, S);      // the seed events; later ones are applied by the waves that raise them
        // Diagnostic (CF_PLACE_DEBUG="iteration,read,read"): before that greedy iteration of the first stage, print every score
        // entry of the two reads and the cached slice candidates that name them
        int64_t dbg_it = -1, dbg_r[2] = {-1, -1};
        if (const char* dv = std::getenv("CF_PLACE_DEBUG")) { long long a_ = -1, b_ = -1, c_ = -1; if (std::sscanf(dv, "%lld,%lld,%lld", &a_, &b_, &c_) == 3 && stage_cls == 1) { dbg_it = a_; dbg_r[0] = b_; dbg_r[1] = c_; } }
        for (int64_t it = 0; it < n_iter; ++it) {
            if (dbg_it >= 0 && (it == dbg_it || it == dbg_it + 1)) {
                CF_HIP(hipStreamSynchronize(st));
                std::vector<unsigned long long> hk((size_t)score_cap), hv((size_t)score_cap);
                std::vector<uint8_t> hf((size_t)score_cap);
                std::vector<cf_cand> hb((size_t)n_am);
                std::vector<uint8_t> hd((size_t)n_am), hu((size_t)R);
                cf_cand hbest;
                CF_HIP(hipMemcpy(hk.data(), S.skeys, (size_t)score_cap * 8, hipMemcpyDeviceToHost));
                CF_HIP(hipMemcpy(hv.data(), S.s01, (size_t)score_cap * 8, hipMemcpyDeviceToHost));
                CF_HIP(hipMemcpy(hf.data(), S.qflag, (size_t)score_cap, hipMemcpyDeviceToHost));
                CF_HIP(hipMemcpy(hb.data(), S.block_best, (size_t)n_am * sizeof(cf_cand), hipMemcpyDeviceToHost));
                CF_HIP(hipMemcpy(hd.data(), S.dirty, (size_t)n_am, hipMemcpyDeviceToHost));
                CF_HIP(hipMemcpy(hu.data(), d_used, (size_t)R, hipMemcpyDeviceToHost));
                CF_HIP(hipMemcpy(&hbest, S.best, sizeof hbest, hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[cf_place debug] before iteration %lld: score_cap=%llu n_am=%d slice_shift=%u last best: read=%u off=%u s0=%u s1=%u rank=%u valid=%u\n", (long long)it, (unsigned long long)score_cap, n_am, S.slice_shift, hbest.read, hbest.off, hbest.s0, hbest.s1, hbest.rank, hbest.valid);
                for (uint64_t i = 0; i < score_cap; ++i) {
                    if (!hk[i]) continue;
                    const int64_t r = (int64_t)((hk[i] & ~CF_OCC) >> 32);
                    if (r != dbg_r[0] && r != dbg_r[1]) continue;
                    const uint32_t v0 = (uint32_t)(hv[i] >> 32), v1 = (uint32_t)hv[i];
                    if (v1 >= 20) std::fprintf(stderr, "  slot %llu (slice %llu dirty %d) read %lld used %d off %u s0 %u s1 %u flag %d\n", (unsigned long long)i, (unsigned long long)(i >> S.slice_shift), (int)hd[i >> S.slice_shift], (long long)r, (int)hu[(size_t)r], (uint32_t)hk[i], v0, v1, (int)hf[i]);
                }
                for (int b = 0; b < n_am; ++b) if (hb[b].valid && ((int64_t)hb[b].read == dbg_r[0] || (int64_t)hb[b].read == dbg_r[1]))
                    std::fprintf(stderr, "  cached slice %d: read %u off %u s0 %u s1 %u rank %u\n", b, hb[b].read, hb[b].off, hb[b].s0, hb[b].s1, hb[b].rank);
            }
            if (!fused) hipLaunchKernelGGL(cf_place_update_kernel, dim3((unsigned)n_blocks), dim3(PL_THREADS), 0, st, S);
            hipLaunchKernelGGL(cf_place_argmax_kernel, dim3((unsigned)n_am), dim3(PL_THREADS), 16 * sizeof(cf_cand) + 16, st, S);
            if (fused) hipLaunchKernelGGL(cf_place_pick_add_update_kernel, dim3((unsigned)(ctx->place_grid > 0 ? ctx->place_grid : std::max(8, ctx->n_cu))), dim3(PL_THREADS), 16 * sizeof(cf_cand), st, S, n_am, ctx->place_chunk);
            else hipLaunchKernelGGL(cf_place_pick_add_kernel, dim3(8), dim3(PL_THREADS), 16 * sizeof(cf_cand), st, S, n_am);
            if ((it & 255) == 255 || it + 1 == n_iter) {
                CF_KERNEL_CHECK("placement iteration");
                CF_HIP(hipMemcpyAsync(h_ctl, S.ctl, 16, hipMemcpyDeviceToHost, st));
                CF_HIP(hipStreamSynchronize(st));
                if (std::getenv("CF_DEBUG") && ((it & 8191) == 8191 || h_ctl[2] || h_ctl[0])) std::fprintf(stderr, "[cf_place] stage %d iter %lld/%lld ctl=%u,%u,%u\n", stage_cls, (long long)it, (long long)n_iter, h_ctl[0], h_ctl[1], h_ctl[2]);
                if (h_ctl[2]) return 1;
                if (h_ctl[0]) break;
            }
        }
        const unsigned int n_out = h_ctl[1];
        std::vector<int64_t> t_read(n_out), t_pos(n_out);
        std::vector<int32_t> t_s0(n_out), t_s1(n_out);
        if (n_out) {
            CF_HIP(hipMemcpy(t_read.data(), S.out_read, (size_t)n_out * 8, hipMemcpyDeviceToHost));
            CF_HIP(hipMemcpy(t_pos.data(), S.out_pos, (size_t)n_out * 8, hipMemcpyDeviceToHost));
            CF_HIP(hipMemcpy(t_s0.data(), S.out_s0, (size_t)n_out * 4, hipMemcpyDeviceToHost));
            CF_HIP(hipMemcpy(t_s1.data(), S.out_s1, (size_t)n_out * 4, hipMemcpyDeviceToHost));
        }
        std::vector<uint8_t> placed((size_t)R, 0);
        for (unsigned int i = 0; i < n_out; ++i) {
            o_read.push_back(t_read[i]); o_pos.push_back(t_pos[i]); o_s0.push_back(t_s0[i]); o_s1.push_back(t_s1[i]);
            placed[(size_t)t_read[i]] = 1;
        }
        // None tail of the stage, ordered by read id (the reference's order is set-iteration order)
        std::vector<int64_t> rest;
        for (int64_t r : stage_reads) if (!placed[(size_t)r]) rest.push_back(r);
        std::sort(rest.begin(), rest.end(), [&](int64_t a, int64_t b) { return id_rank[a] < id_rank[b]; });
        for (int64_t r : rest) { o_read.push_back(r); o_pos.push_back(-1); o_s0.push_back(-1); o_s1.push_back(-1); }
    }
    unsigned int h_ctl[4] = {0, 0, 0, 0};
    CF_HIP(hipMemcpy(h_ctl, S.ctl, 16, hipMemcpyDeviceToHost));
    if (h_ctl[2]) return 1;
    return 0;
}

// the placement's candidate reduction on n given candidates (tests): thread t takes the candidates t, t + 256, ...
__global__ void __launch_bounds__(PL_THREADS)
cf_selftest_argmax_kernel(const uint32_t* __restrict__ cands, int64_t n, uint32_t* __restrict__ out) {
    cf_key mine{0ull, 0ull, 0u};
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        cf_cand c; c.s0 = cands[5 * i]; c.s1 = cands[5 * i + 1]; c.off = cands[5 * i + 2]; c.rank = cands[5 * i + 3]; c.valid = cands[5 * i + 4] ? 1u : 0u; c.read = (uint32_t)i;
        cf_key_take(mine, cf_key_of(c));
    }
    const cf_cand b = cf_block_best(mine);
    if (threadIdx.x == 0) { out[0] = b.s0; out[1] = b.s1; out[2] = b.off; out[3] = b.rank; out[4] = b.read; out[5] = b.valid; }
}

extern "C" {

int cf_selftest_argmax(cf_ctx* ctx, const uint32_t* cands, int64_t n, uint32_t* out6) {
    if (!ctx || !out6 || n < 0 || (n && !cands)) return -22;
    CF_HIP(hipSetDevice(ctx->device));
    Bufs B{ctx, {}};
    uint32_t *d_c = nullptr, *d_o = nullptr;
    CF_TRY(B.get(&d_c, (size_t)5 * (size_t)std::max<int64_t>(n, 1), "selftest candidates"));
    CF_TRY(B.get(&d_o, 8, "selftest winner"));
    if (n) CF_HIP(hipMemcpyAsync(d_c, cands, (size_t)n * 20, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(cf_selftest_argmax_kernel, dim3(1), dim3(PL_THREADS), 16 * sizeof(cf_cand), ctx->stream, (const uint32_t*)d_c, n, d_o);
    CF_KERNEL_CHECK("cf_selftest_argmax_kernel");
    CF_HIP(hipMemcpyAsync(out6, d_o, 24, hipMemcpyDeviceToHost, ctx->stream));
    CF_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int cf_place_reads(cf_ctx* ctx, const uint8_t* cls, const int32_t* id_rank, int32_t min_cloud_kmer_freq,
                   int32_t min_unit, int32_t min_inters, int32_t min_prop,
                   int64_t* out_read, int64_t* out_pos, int32_t* out_s0, int32_t* out_s1) {
    if (!ctx) return -22;
    if (!ctx->have_clouds) return cf_fail(ctx, -22, "cf_place_reads: no clouds built");
    if (!cls || !id_rank || !out_read || !out_pos || !out_s0 || !out_s1) return cf_fail(ctx, -22, "cf_place_reads: null argument");
    CF_HIP(hipSetDevice(ctx->device));
    CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    const int64_t R = ctx->n_reads;
    for (int64_t r = 0; r < R; ++r) if (cls[r] > 2) return cf_fail(ctx, -22, "cf_place_reads: class must be 0, 1 or 2");
    // round 4: per-read score regions, one kernel per greedy iteration (cf_place2.hip).  Its rescans look at the rows that have
    // max(1, min_inters) hits: with a threshold that nearly every row reaches (--min-inters 1 .. 3) that is every row of every read a
    // placement touches — measured at 10 000 reads, min_inters 1 / 2 / 3 / 5: 2 256 / 619 / 267 / 164 ms against 220 for the hash-map
    // path (35 s with --min-cloud-kmer-freq 1 on top: tools/place_sweep_check.py) — so small thresholds take the rounds 1-3 path.
    // (place_mode 3 forces the regions whatever the threshold: tests.)
    if ((ctx->place_mode == 3 || (ctx->place_mode == 2 && min_inters >= 4)) && cf_place2_fits(ctx)) {
        std::vector<int64_t> o_read, o_pos;
        std::vector<int32_t> o_s0, o_s1;
        const int rc2 = cf_place2_run(ctx, cls, id_rank, min_cloud_kmer_freq, min_unit, min_inters, min_prop, o_read, o_pos, o_s0, o_s1);
        // -34: the score regions kept overflowing (reads that meet every offset of a long contig: thin coverage with thresholds of 1)
        // or would need more than 2^32 slots — the hash-map path below has no such limit and takes over (place_mode 3: the error stands)
        // -12: the regions were sized within the free memory and an allocation failed all the same (fragmentation): same answer
        if (rc2 != 0 && ((rc2 != -34 && rc2 != -12) || ctx->place_mode == 3)) return rc2;
        if (rc2 == 0) {
            if ((int64_t)o_read.size() != R) return cf_fail(ctx, -5, "cf_place_reads: internal error, output count != reads");
            for (int64_t i = 0; i < R; ++i) { out_read[i] = o_read[(size_t)i]; out_pos[i] = o_pos[(size_t)i]; out_s0[i] = o_s0[(size_t)i]; out_s1[i] = o_s1[(size_t)i]; }
            CF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
            CF_HIP(hipEventSynchronize(ctx->ev1));
            CF_HIP(hipEventElapsedTime(&ctx->times.place_ms, ctx->ev0, ctx->ev1));
            return 0;
        }
        if (std::getenv("CF_DEBUG")) std::fprintf(stderr, "[cf_place_reads] the region path gave up (%s): hash-map path\n", ctx->err.c_str());
    }
    uint64_t score_cap = cf_pow2_ceil((uint64_t)std::max<int64_t>(256 * R, 1 << 14));
    uint64_t seen_cap = cf_pow2_ceil((uint64_t)std::max<int64_t>(8 * ctx->n_entries, 1 << 14));
    std::vector<int64_t> o_read, o_pos;
    std::vector<int32_t> o_s0, o_s1;
    int rc = 1;
    for (int attempt = 0; attempt < 6 && rc == 1; ++attempt) {
        rc = place_attempt(ctx, cls, id_rank, min_cloud_kmer_freq, min_unit, min_inters, min_prop, score_cap, seen_cap, o_read, o_pos, o_s0, o_s1);
        if (std::getenv("CF_DEBUG")) std::fprintf(stderr, "[cf_place] attempt %d rc=%d score_cap=%llu seen_cap=%llu entries=%lld reads=%lld\n", attempt, rc, (unsigned long long)score_cap, (unsigned long long)seen_cap, (long long)ctx->n_entries, (long long)R);
        if (rc == 1) { score_cap *= 4; seen_cap *= 4; }
    }
    if (rc == 1) return cf_fail(ctx, -34, "cf_place_reads: score tables kept overflowing");
    if (rc) return rc;
    if ((int64_t)o_read.size() != R) return cf_fail(ctx, -5, "cf_place_reads: internal error, output count != reads");
    for (int64_t i = 0; i < R; ++i) { out_read[i] = o_read[(size_t)i]; out_pos[i] = o_pos[(size_t)i]; out_s0[i] = o_s0[(size_t)i]; out_s1[i] = o_s1[(size_t)i]; }
    CF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    CF_HIP(hipEventSynchronize(ctx->ev1));
    CF_HIP(hipEventElapsedTime(&ctx->times.place_ms, ctx->ev0, ctx->ev1));
    return 0;
}

}  // extern "C"
