// cf_count2.hip — A1 without a single global atomic per k-mer: sort and reduce.
//
// Reference: scripts/distance_based_kmer_recruitment.py:39-63 (closed form: pres[x] = #reads containing x,
// multi[x] = #reads where x occurs >= 2x; see cf_count.hip).
//
// Round 1 built the presence table with one 64-bit CAS + one 64-bit add per distinct (read, k-mer) into a 16 GiB open
// addressed table: 8e8 x 2 device-scope atomics on random lines run at ~2e10 per second whatever their locality
// (DESIGN §3.1), i.e. 80 ms per Gbase at 2 % of the HBM roofline.  Here nothing is updated in place:
//   1. every window becomes ONE 8-byte record [k-mer : 2k | read : RB]                       (no per-read work yet)
//   2. the records are partitioned by a hash of the k-mer with stable LSD radix passes (<= 9 bits each; the first pass
//      makes its records straight from the bases), so all records of a k-mer end up in one bucket, in read order
//   3. one streaming pass reduces every bucket in LDS: a record looks back along its read's run of the bucket (the sort is
//      stable, so the records of one read stand together) and is the first, the second or a later occurrence of its k-mer in
//      that read; first ones add to pres, second ones to multi in a per-bucket table keyed by k-mer
//   4. the result is a DENSE array of table slots {key | OCC, pres | multi << 32}: every consumer that scans the table
//      (A2 select, cf_get_table, the multi-GPU bucketing) reads it like a table without holes.
// All traffic is streamed: N_b x 2 + 8 N_w x 5 + 16 K_dist bytes (~45 GB per Gbase).  Needs 2k + bits(reads) <= 64 and
// reads < 2^27; otherwise the table path of cf_count.hip runs.  cf_count_occurrences takes the same passes with records that
// are the k-mer alone and its own reduce kernel (cf_c2_reduce_occ_kernel below).
#include "cf_common.h"

void cf_free_table(cf_ctx* c);

#define C2_THREADS 256
#define C2_ITEMS 16                         /* windows (pass 1) / records (later passes) per thread and tile */
#define C2_TILE (C2_THREADS * C2_ITEMS)
#define C2_MAXBITS 9                        /* radix bits per pass */
#ifndef C2_RTHREADS
#define C2_RTHREADS 512                     /* threads of a reduce workgroup */
#endif
#ifndef C2_RTILE
#define C2_RTILE 2048                       /* records per reduce tile */
#endif
#ifndef C2_TAB
#define C2_TAB 2048                         /* k-mers per bucket table */
#endif
#define C2_DUP (1ull << 63)

struct cf_c2_tile { int32_t read; int32_t chunk; };      // pass-1 tile: windows [chunk * C2_TILE, ...) of a read

// Hashes of this file: ONE 32-bit multiply each (a 64-bit mixer is two 64 x 64 multiplies = eight quarter-rate 32-bit
// ones, and the reduce evaluates three hashes per record).  A k-mer is folded to 32 bits first; the bucket takes the top
// `bits` (<= 27) bits of the product, the two LDS tables use other multipliers.
__device__ __forceinline__ uint32_t cf_c2_fold(unsigned long long x) { const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32); return lo ^ (hi * 0x85EBCA6Bu) ^ (hi >> 7); }
__device__ __forceinline__ uint32_t cf_c2_bucket(unsigned long long kmer, int bits) { uint32_t h = cf_c2_fold(kmer); h ^= h >> 15; h *= 0x9E3779B1u; h ^= h >> 13; h *= 0xC2B2AE35u; return h >> (32 - bits); }
__device__ __forceinline__ uint32_t cf_c2_hash_set(unsigned long long rec) { uint32_t h = cf_c2_fold(rec); h ^= h >> 16; h *= 0x7FEB352Du; return h ^ (h >> 15); }
__device__ __forceinline__ uint32_t cf_c2_hash_tab(unsigned long long kmer) { uint32_t h = cf_c2_fold(kmer); h ^= h >> 14; h *= 0x846CA68Bu; return h ^ (h >> 16); }

// ---- stable ranking of one round (one record per thread) by digit, as in cf_radix_scatter (cf_prims.hip)
struct cf_c2_rank { uint32_t rank, count; };
template <int NB>
__device__ __forceinline__ cf_c2_rank cf_c2_wave_rank(uint32_t digit, bool valid, int lane) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int bit = (digit >> b) & 1;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return cf_c2_rank{(uint32_t)__popcll(peers & ((1ull << lane) - 1ull)), (uint32_t)__popcll(peers)};
}

// windows of a pass-1 tile -> records, 16 consecutive windows per thread (rolled); calls f(j, record, valid) for j < 16.
// The tile's bases come in as aligned 32-bit words (`a` = the tile's first base's offset in its word, uniform) and are
// handled four to a word: 2-bit codes ((w >> 1) ^ (w >> 2)) & 0x03030303, folded to one byte per word; the alphabet test
// is a byte lookup of the code (v_perm_b32 into "ACGT") compared with the word itself.  Round 2 walked the thread's 46 bases
// one at a time (~50 instructions per window of ~80).
template <class F>
__device__ __forceinline__ void cf_c2_tile_records(const uint8_t* __restrict__ bases, int64_t n_bases, const int64_t* __restrict__ read_off, cf_c2_tile tl, int k, int rb,
                                                   uint8_t* stage, F&& f) {
    const int t = threadIdx.x;
    const int64_t r0 = read_off[tl.read], r1 = read_off[tl.read + 1];
    const int64_t n_win = r1 - r0 - k + 1, w0 = (int64_t)tl.chunk * C2_TILE;
    const int64_t nb = min((int64_t)C2_TILE + k - 1, r1 - r0 - w0);
    const int64_t g0 = r0 + w0;
    const uint32_t a = (uint32_t)(g0 & 3);
    const int64_t gw = g0 - a;                                 // (the array begins on a word boundary)
    const int n_dw = (int)((a + nb + 3) >> 2);
    uint32_t* st32 = (uint32_t*)stage;
    for (int d = t; d < n_dw; d += C2_THREADS) {
        uint32_t v = 0;
        if (gw + 4 * (int64_t)d + 4 <= n_bases) v = *(const uint32_t*)(bases + gw + 4 * (int64_t)d);
        else for (int q = 0; q < 4; ++q) { const int64_t p = gw + 4 * (int64_t)d + q; if (p < n_bases) v |= (uint32_t)bases[p] << (8 * q); }
        st32[d] = v;
    }
    __syncthreads();
    const unsigned long long kmask = (1ull << (2 * k)) - 1ull;
    const int64_t my0 = (int64_t)t * C2_ITEMS;
    const int64_t my_n = min((int64_t)C2_ITEMS, n_win - w0 - my0);
    // the thread's 16 + k - 1 <= 46 bases start `a` bytes into the 13 words at my0: three 16-byte LDS reads + one word (lane
    // stride 16 bytes: conflict free), shifted into place two words at a time
    struct alignas(16) q4 { uint32_t x, y, z, w; };
    const q4 qa = *(const q4*)(stage + my0), qb = *(const q4*)(stage + my0 + 16), qc = *(const q4*)(stage + my0 + 32);
    const uint32_t raw[13] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w, st32[(my0 >> 2) + 12]};
    uint32_t w32[12], c8[12], diff = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        w32[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], a);
        const uint32_t c = ((w32[i] >> 1) ^ (w32[i] >> 2)) & 0x03030303u;                // codes of the word's four bases, one per byte
        const uint32_t back = __builtin_amdgcn_perm(0u, 0x54474341u, c);                  // code -> 'A', 'C', 'G', 'T'
        diff |= (back ^ w32[i]) & (i == 11 ? 0x0000FFFFu : 0xFFFFFFFFu);                  // (bases 46, 47 belong to no window of this thread)
        c8[i] = ((c << 6) | (c >> 4) | (c >> 14) | (c >> 24)) & 0xFFu;                   // first base highest
    }
    const uint32_t W0 = (c8[0] << 24) | (c8[1] << 16) | (c8[2] << 8) | c8[3], W1 = (c8[4] << 24) | (c8[5] << 16) | (c8[6] << 8) | c8[7],
                   W2 = (c8[8] << 24) | (c8[9] << 16) | (c8[10] << 8) | c8[11];
    // codes of bases 0 .. 13 in hi (28 bits), 14 .. 45 in lo (64 bits); a window = a 128-bit shift
    const unsigned long long hi = W0 >> 4, lo = ((((unsigned long long)W1 << 32) | W2) >> 4) | ((unsigned long long)W0 << 60);
    unsigned long long bad = 0;             // bit i: base i is not upper-case A, C, G, T (A1 does not upper-case: reference :47-53)
    if (diff) {
        for (int i = 0; i < 46; ++i) {
            const uint32_t ch = (w32[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            bad |= (unsigned long long)!cf_is_acgt(ch) << i;
        }
    }
    const unsigned long long wmask = (1ull << k) - 1ull;
#pragma unroll
    for (int j = 0; j < C2_ITEMS; ++j) {
        const bool valid = j < my_n && ((bad >> j) & wmask) == 0ull;      // a window with another symbol has no code: counted on the host side
        const int s2 = 2 * (46 - j - k);                 // bits to drop behind the window [j, j + k): 0 .. 90
        unsigned long long code = s2 >= 64 ? hi >> (s2 - 64) : (s2 ? (lo >> s2) | (hi << (64 - s2)) : lo);
        code &= kmask;
        f(j, (code << rb) | ((unsigned long long)(uint32_t)tl.read & ((1ull << rb) - 1ull)), valid);      // (rb = 0: occurrence mode, the record is the k-mer alone)
    }
}

// pass 1, histogram: digit counts of every tile -> hist[tile * D + digit] (tile-major: a tile writes and reads ONE run of D
// counters; digit-major — round 2 — made every counter its own 64-byte sector on the way out and on the way back in: 20 GB of
// the stage's 91 GB of HBM traffic)
__global__ void __launch_bounds__(C2_THREADS)
cf_c2_hist1_kernel(const uint8_t* __restrict__ bases, int64_t n_bases, const int64_t* __restrict__ read_off, const cf_c2_tile* __restrict__ tiles, int n_tiles,
                   int k, int rb, int bits, int shift, int nb, uint32_t* __restrict__ hist) {
    uint32_t* h = (uint32_t*)cf_lds;                          // 1 << nb counters
    uint8_t* stage = cf_lds + ((((size_t)4 << nb) + 15) & ~(size_t)15);
    const uint32_t mask = (1u << nb) - 1u;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int d = threadIdx.x; d < (1 << nb); d += C2_THREADS) h[d] = 0;
        __syncthreads();
        cf_c2_tile_records(bases, n_bases, read_off, tiles[tile], k, rb, stage, [&](int, unsigned long long rec, bool valid) {
            if (valid) atomicAdd(&h[(cf_c2_bucket(rec >> rb, bits) >> shift) & mask], 1u);
        });
        __syncthreads();
        for (int d = threadIdx.x; d < (1 << nb); d += C2_THREADS) hist[((int64_t)tile << nb) + d] = h[d];
        __syncthreads();
    }
}

// Stable scatter of one tile.  The tile's order is wave-major: wave w holds the records [w * 64 * C2_ITEMS, ...) of the
// tile, C2_ITEMS rounds of 64.  A wave ranks its own records by itself — per round a ballot match per digit bit, across
// rounds a running count per digit in the wave's private LDS row — so the workgroup meets only twice per tile: to turn
// the rows into exclusive offsets across waves, and before the rows are reused.
//   rank_[j]  rank of the thread's j-th record among the records of its wave with the same digit (filled by cf_c2_rank_round)
template <int NB>
__device__ __forceinline__ uint32_t cf_c2_rank_round(uint32_t digit, bool valid, uint32_t* wrow) {     // wrow: this wave's 1 << NB counters
    const int lane = threadIdx.x & 63;
    const cf_c2_rank rk = cf_c2_wave_rank<NB>(digit, valid, lane);
    uint32_t before = 0;
    if (valid) before = wrow[digit];                       // records of earlier rounds (only this wave writes the row; LDS ops of a wave are in order)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (valid && rk.rank == 0) wrow[digit] = before + rk.count;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return before + rk.rank;
}
// after all rounds: wcount rows -> exclusive prefix over the waves (in place), dstart[d] = first position of digit d in
// the tile sorted by digit (exclusive scan of the tile's digit counts; 1 << NB <= 2 * C2_THREADS)
template <int NB, int ROWS = C2_THREADS / 64>
__device__ __forceinline__ void cf_c2_tile_bases(uint32_t* dstart, uint32_t* wcount, uint32_t* scan_tmp) {
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t tot[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int d = 2 * t + h;
        if (d < (1 << NB)) {
            uint32_t s = 0;
            for (int w = 0; w < ROWS; ++w) { const uint32_t c = wcount[w * (1 << NB) + d]; wcount[w * (1 << NB) + d] = s; s += c; }
            tot[h] = s;
        }
    }
    // exclusive scan of tot over the threads (2 digits each)
    const int lane = t & 63, wave = t >> 6;
    uint32_t inc = tot[0] + tot[1];
    const uint32_t mine = inc;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
    if (lane == 63) scan_tmp[wave] = inc;
    __syncthreads();
    uint32_t off = inc - mine;
    for (int w = 0; w < wave; ++w) off += scan_tmp[w];
    if (2 * t < (1 << NB)) dstart[2 * t] = off;
    if (2 * t + 1 < (1 << NB)) dstart[2 * t + 1] = off + tot[0];
    if (t == C2_THREADS - 1) scan_tmp[7] = off + mine;          // records of the tile
    __syncthreads();
}
// records staged in digit order -> global: consecutive threads write consecutive addresses inside a digit's run
template <int NB>
__device__ __forceinline__ void cf_c2_copy_out(const unsigned long long* stage_recs, uint32_t n_tile, const uint32_t* dstart, const int64_t* gbase,
                                               int rb, int bits, int shift, unsigned long long* __restrict__ out) {
    const uint32_t mask = (1u << NB) - 1u;
    for (uint32_t i = threadIdx.x; i < n_tile; i += C2_THREADS) {
        const unsigned long long rec = stage_recs[i];
        const uint32_t d = (cf_c2_bucket(rec >> rb, bits) >> shift) & mask;
        out[gbase[d] + (int64_t)(i - dstart[d])] = rec;
    }
}

// pass 1, scatter: offs = exclusive scan of hist in (digit, tile) order (cf_c2_col* below).  The windows of a tile belong to ONE read, so their order inside the tile
// is free: a record's rank among the tile's records with its digit is what a returning LDS add hands out (the ballot
// ranking of the later passes, which keeps the order, costs ~45 instructions per record); the digit is kept next to the
// staged record for the copy-out.
// LDS: gbase int64[D] | cnt u32[D] | dstart u32[D] | scan_tmp u32[8] | staged records u64[C2_TILE] | their digits u16[C2_TILE] | bases
template <int NB>
__global__ void __launch_bounds__(C2_THREADS)
cf_c2_scatter1_kernel(const uint8_t* __restrict__ bases, int64_t n_bases, const int64_t* __restrict__ read_off, const cf_c2_tile* __restrict__ tiles, int n_tiles,
                      int k, int rb, int bits, int shift, const int64_t* __restrict__ offs, unsigned long long* __restrict__ out) {
    constexpr int D = 1 << NB;
    int64_t* gbase = (int64_t*)cf_lds;
    uint32_t* cnt = (uint32_t*)(gbase + D);
    uint32_t* dstart = cnt + D;
    uint32_t* scan_tmp = dstart + D;
    unsigned long long* srec = (unsigned long long*)(((uintptr_t)(scan_tmp + 8) + 15) & ~(uintptr_t)15);
    uint16_t* sdig = (uint16_t*)(srec + C2_TILE);
    uint8_t* stage = (uint8_t*)(sdig + C2_TILE);
    const uint32_t mask = D - 1u;
    for (int d = threadIdx.x; d < D; d += C2_THREADS) cnt[d] = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int d = threadIdx.x; d < D; d += C2_THREADS) gbase[d] = offs[(int64_t)tile * D + d];
        unsigned long long rec_[C2_ITEMS];
        uint32_t rd_[C2_ITEMS];       // rank | digit << 16
        uint32_t ok = 0;
        cf_c2_tile_records(bases, n_bases, read_off, tiles[tile], k, rb, stage, [&](int j, unsigned long long rec, bool valid) {      // (its barrier orders the zeroed counters)
            const uint32_t d = (cf_c2_bucket(rec >> rb, bits) >> shift) & mask;
            rec_[j] = rec; ok |= (uint32_t)valid << j;
            rd_[j] = (valid ? atomicAdd(&cnt[d], 1u) : 0u) | (d << 16);
        });
        cf_c2_tile_bases<NB, 1>(dstart, cnt, scan_tmp);       // (leaves the counters at 0 for the next tile)
#pragma unroll
        for (int j = 0; j < C2_ITEMS; ++j)
            if ((ok >> j) & 1u) { const uint32_t d = rd_[j] >> 16, at = dstart[d] + (rd_[j] & 0xFFFFu); srec[at] = rec_[j]; sdig[at] = (uint16_t)d; }
        const uint32_t n_tile = scan_tmp[7];      // records of the tile: its windows of plain A, C, G, T
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_tile; i += C2_THREADS) {      // consecutive threads write consecutive addresses inside a digit's run
            const uint32_t d = sdig[i];
            out[gbase[d] + (int64_t)(i - dstart[d])] = srec[i];
        }
        __syncthreads();
    }
}

// later passes: records -> records
__global__ void __launch_bounds__(C2_THREADS)
cf_c2_hist_kernel(const unsigned long long* __restrict__ in, int64_t n, int n_tiles, int rb, int bits, int shift, int nb, uint32_t* __restrict__ hist) {
    uint32_t* h = (uint32_t*)cf_lds;
    const uint32_t mask = (1u << nb) - 1u;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int d = threadIdx.x; d < (1 << nb); d += C2_THREADS) h[d] = 0;
        __syncthreads();
        const int64_t base = (int64_t)tile * C2_TILE;
#pragma unroll
        for (int j = 0; j < C2_ITEMS; ++j) {
            const int64_t i = base + (int64_t)j * C2_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&h[(cf_c2_bucket(in[i] >> rb, bits) >> shift) & mask], 1u);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < (1 << nb); d += C2_THREADS) hist[((int64_t)tile << nb) + d] = h[d];
        __syncthreads();
    }
}

template <int NB>
__global__ void __launch_bounds__(C2_THREADS)
cf_c2_scatter_kernel(const unsigned long long* __restrict__ in, int64_t n, int n_tiles, int rb, int bits, int shift, const int64_t* __restrict__ offs,
                     unsigned long long* __restrict__ out) {
    constexpr int D = 1 << NB;
    int64_t* gbase = (int64_t*)cf_lds;
    uint32_t* wcount = (uint32_t*)(gbase + D);
    uint32_t* dstart = wcount + (C2_THREADS / 64) * D;
    uint32_t* scan_tmp = dstart + D;
    unsigned long long* srec = (unsigned long long*)(((uintptr_t)(scan_tmp + 8) + 15) & ~(uintptr_t)15);
    const uint32_t mask = D - 1u;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = threadIdx.x; d < (C2_THREADS / 64) * D; d += C2_THREADS) wcount[d] = 0;
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int d = threadIdx.x; d < D; d += C2_THREADS) gbase[d] = offs[(int64_t)tile * D + d];
        // wave w takes the records [w * 64 * C2_ITEMS, ...) of the tile in rounds of 64: array order = (wave, round, lane)
        const int64_t base = (int64_t)tile * C2_TILE + (int64_t)wave * 64 * C2_ITEMS + lane;
        unsigned long long rec_[C2_ITEMS];
        uint32_t rank_[C2_ITEMS];
#pragma unroll
        for (int j = 0; j < C2_ITEMS; ++j) { const int64_t i = base + (int64_t)j * 64; rec_[j] = i < n ? in[i] : 0ull; }
#pragma unroll
        for (int j = 0; j < C2_ITEMS; ++j)
            rank_[j] = cf_c2_rank_round<NB>((cf_c2_bucket(rec_[j] >> rb, bits) >> shift) & mask, base + (int64_t)j * 64 < n, wcount + wave * D);
        cf_c2_tile_bases<NB>(dstart, wcount, scan_tmp);
#pragma unroll
        for (int j = 0; j < C2_ITEMS; ++j)
            if (base + (int64_t)j * 64 < n) { const uint32_t d = (cf_c2_bucket(rec_[j] >> rb, bits) >> shift) & mask; srec[dstart[d] + wcount[wave * D + d] + rank_[j]] = rec_[j]; }
        const uint32_t n_tile = (uint32_t)min((int64_t)C2_TILE, n - (int64_t)tile * C2_TILE);
        __syncthreads();
        cf_c2_copy_out<NB>(srec, n_tile, dstart, gbase, rb, bits, shift, out);
        __syncthreads();
        for (int d = threadIdx.x; d < (C2_THREADS / 64) * D; d += C2_THREADS) wcount[d] = 0;
    }
}

// ---- offsets of a pass: offs[tile][d] = records with a smaller digit + records of digit d in earlier tiles = the exclusive scan
// of hist in (digit, tile) order, computed on the tile-major arrays by columns: (1) column sums of chunks of C2_SCAN_CHUNK
// tiles, (2) one workgroup turns them into the chunks' bases (column totals, scan over the digits, running sums down the
// chunks), (3) every chunk walks its tiles again.  All accesses are runs of D counters; the loads of a walk do not depend on
// its running sum and are issued eight at a time.
#define C2_SCAN_CHUNK 512
#define C2_SCAN_THREADS 512                 /* >= 1 << C2_MAXBITS: a thread per digit */
__global__ void __launch_bounds__(C2_SCAN_THREADS)
cf_c2_colsum_kernel(const uint32_t* __restrict__ hist, int n_tiles, int D, uint32_t* __restrict__ part) {
    const int d = threadIdx.x;
    if (d >= D) return;
    const int t0 = blockIdx.x * C2_SCAN_CHUNK, t1 = min(n_tiles, t0 + C2_SCAN_CHUNK);
    uint32_t acc = 0;
    for (int t = t0; t < t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t + u < t1 ? hist[(int64_t)(t + u) * D + d] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    part[(int64_t)blockIdx.x * D + d] = acc;
}
__global__ void __launch_bounds__(C2_SCAN_THREADS)
cf_c2_colbase_kernel(const uint32_t* __restrict__ part, int n_chunks, int D, int64_t* __restrict__ base, int64_t* __restrict__ total_out) {
    long long* sh = (long long*)cf_lds;                       // C2_SCAN_THREADS / 64 wave totals
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    long long tot = 0;
    if (d < D)
        for (int c = 0; c < n_chunks; c += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c + u < n_chunks ? part[(int64_t)(c + u) * D + d] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += v[u];
        }
    long long inc = tot;                                       // inclusive scan over the digits
    for (int s = 1; s < 64; s <<= 1) { const long long o = __shfl_up(inc, (unsigned)s); if (lane >= s) inc += o; }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    long long run = inc - tot;
    for (int w = 0; w < wave; ++w) run += sh[w];
    if (d == D - 1) *total_out = run + tot;
    if (d < D)
        for (int c = 0; c < n_chunks; c += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c + u < n_chunks ? part[(int64_t)(c + u) * D + d] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (c + u < n_chunks) { base[(int64_t)(c + u) * D + d] = run; run += v[u]; }
        }
}
__global__ void __launch_bounds__(C2_SCAN_THREADS)
cf_c2_coloffs_kernel(const uint32_t* __restrict__ hist, int n_tiles, int D, const int64_t* __restrict__ base, int64_t* __restrict__ offs) {
    const int d = threadIdx.x;
    if (d >= D) return;
    const int t0 = blockIdx.x * C2_SCAN_CHUNK, t1 = min(n_tiles, t0 + C2_SCAN_CHUNK);
    long long run = base[(int64_t)blockIdx.x * D + d];
    for (int t = t0; t < t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t + u < t1 ? hist[(int64_t)(t + u) * D + d] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (t + u < t1) { offs[(int64_t)(t + u) * D + d] = run; run += v[u]; }
    }
}

// ---- the reduce.  Records are sorted by bucket; inside a bucket they are in emission order, so the records of one
// k-mer come in read order.  Every workgroup takes a contiguous range of buckets, finds its first record by binary
// search and streams on from there.
// counters: [0] slots reserved (chunks) [1] sum of pres (= N_rk) [2] flags (1: a bucket holds more k-mers than the LDS
// table) [3] k-mers written
// first record of every chunk of `per` buckets (binary search on the bucket-sorted records; one thread per chunk)
__global__ void __launch_bounds__(256)
cf_c2_chunk_starts_kernel(const unsigned long long* __restrict__ recs, int64_t n, int rb, int bits, int64_t per, int64_t n_chunks, int64_t* __restrict__ starts) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chunks) return;
    const int64_t b0 = c * per;
    int64_t lo = 0, hi = n;
    if (c == n_chunks) lo = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if ((int64_t)cf_c2_bucket(recs[m] >> rb, bits) < b0) lo = m + 1; else hi = m; }
    starts[c] = lo;
}

// Buckets are handed out in chunks of `per` through a ticket (counters[4]): a bucket of a repeat's consensus k-mer has 40 x the
// records of an average one, and with a fixed range per workgroup the unlucky ones took three times the mean.
#if defined(CF_C2_STAMPS)
#define C2_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_t; st_t = now_; } } while (0)
#else
#define C2_STAMP(i) do { } while (0)
#endif
// The dense table is written in chunks a workgroup reserves with one global atomic each (one reservation per bucket
// cost a round trip of ~2 us per ~4000 records); the part of a chunk it does not fill is zeroed: empty slots, which every
// consumer of the table skips.
//
// What a record adds is decided by LOOKING BACK: the sort is stable, so inside a bucket the records of one read are a
// contiguous run (a handful of records: a read has 2e4 windows for 2.6e5 buckets), and a record is the first / the second /
// a later occurrence of its k-mer in its read according to how many equal records stand before it in that run:
//   first  -> pres of the k-mer + 1;   second -> multi + 1;   later -> nothing.
// The tile is staged in LDS for that; a run that began in an earlier tile is followed into the record array itself (down to
// the bucket's first record).  Rounds 1-2 kept a (k-mer, read) hash set per tile and a "last read" word per k-mer instead:
// a CAS chain, a flag, a clean-up store and an atomicMax per record more than this.  bstart[b] = first record of bucket b
// (cf_c2_chunk_starts_kernel with one bucket per chunk), so a tile knows its length without hashing its records.
__global__ void __launch_bounds__(C2_RTHREADS)
cf_c2_reduce_kernel(const unsigned long long* __restrict__ recs, int64_t n, int rb, int bits, cf_slot* __restrict__ out, unsigned long long out_cap,
                    unsigned long long chunk, int64_t per, int64_t n_chunks, const int64_t* __restrict__ bstart, unsigned long long* __restrict__ counters) {
    unsigned long long* srec = (unsigned long long*)cf_lds;                    // the tile's records
    unsigned long long* tkey = srec + C2_RTILE;                                 // C2_TAB x (k-mer + 1); 0 = empty
    uint32_t* tpres = (uint32_t*)(tkey + C2_TAB);
    uint32_t* tmulti = tpres + C2_TAB;
    unsigned long long* sh64 = (unsigned long long*)(tmulti + C2_TAB);          // [0] ticket / flush base [1] next free slot of the chunk [2] end of the chunk
    uint32_t* sh = (uint32_t*)(sh64 + 3);                                       // [1] k-mers in the table
    const int t = threadIdx.x, lane = t & 63;
    const unsigned long long read_mask = (1ull << rb) - 1ull;
    const int64_t n_buckets = (int64_t)1 << bits;
    constexpr int RJ = C2_RTILE / C2_RTHREADS;
    if (t == 0) { sh64[1] = 0; sh64[2] = 0; }
    unsigned long long n_dist = 0;      // thread 0: k-mers written by this workgroup
    unsigned long long psum_all = 0;    // sum of pres of the k-mers this thread flushed
#if defined(CF_C2_STAMPS)
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
    __syncthreads();
    if (t == 0) sh64[0] = atomicAdd(&counters[4], 1ull);
    __syncthreads();
    const int64_t ck = (int64_t)sh64[0];
    C2_STAMP(0);   // ticket
    if (ck >= n_chunks) break;
    const int64_t b0 = ck * per, b1 = min(n_buckets, b0 + per);
    // ld[] holds the records [ld_pos, ld_pos + C2_RTILE): the loads of the NEXT tile (the buckets of a chunk follow one another
    // in the array) are issued as soon as this tile is staged and land while its table phase and the bucket's flush run
    unsigned long long ld[RJ];
    int64_t ld_pos = -1;
    for (int64_t b = b0; b < b1; ++b) {
        const int64_t bpos = bstart[b], bend = bstart[b + 1];
        if (bpos >= bend) continue;      // (uniform) an empty bucket
        for (int s = t; s < C2_TAB; s += C2_RTHREADS) { tkey[s] = 0ull; tpres[s] = 0; tmulti[s] = 0; }
        if (t == 0) sh[1] = 0;
        C2_STAMP(1);   // table clear
        for (int64_t pos = bpos; pos < bend; pos += C2_RTILE) {
            const uint32_t got = (uint32_t)min((int64_t)C2_RTILE, bend - pos);
            if (ld_pos != pos) {
#pragma unroll
                for (int j = 0; j < RJ; ++j) {      // all loads of the tile in flight before the first is used
                    const int64_t i = pos + (int64_t)j * C2_RTHREADS + t;
                    ld[j] = i < n ? recs[i] : ~0ull;
                }
            }
            unsigned long long rec[RJ];
#pragma unroll
            for (int j = 0; j < RJ; ++j) { rec[j] = ld[j]; if ((uint32_t)(j * C2_RTHREADS + t) < got) srec[j * C2_RTHREADS + t] = rec[j]; }
            __syncthreads();            // the tile is staged (and the table clear of a bucket's first tile is done)
            {
#pragma unroll
                for (int j = 0; j < RJ; ++j) {
                    const int64_t i = pos + (int64_t)got + (int64_t)j * C2_RTHREADS + t;
                    ld[j] = i < n ? recs[i] : ~0ull;
                }
                ld_pos = pos + (int64_t)got;
            }
            C2_STAMP(2);   // loads + staging
            // ---- occurrences of the record's k-mer earlier in its read's run: 0, 1 or "2 or more"
            uint32_t seen[RJ], act = 0;
            unsigned long long prev[RJ];
#pragma unroll
            for (int j = 0; j < RJ; ++j) {          // the step that settles almost every record — the record before it is of another read — for all four at once
                const uint32_t x = (uint32_t)(j * C2_RTHREADS + t);
                seen[j] = 0; prev[j] = ~0ull;
                if (x < got) { act |= 1u << j; if (x > 0u) prev[j] = srec[x - 1u]; else if (pos > bpos) prev[j] = recs[pos - 1]; }
            }
            uint32_t deep = 0;
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                if (!((act >> j) & 1u)) continue;
                if (prev[j] != ~0ull && ((prev[j] ^ rec[j]) & read_mask) == 0ull) { seen[j] = prev[j] == rec[j] ? 1u : 0u; deep |= 1u << j; }
            }
            if (deep) {
#pragma unroll
                for (int j = 0; j < RJ; ++j) {
                    if (!((deep >> j) & 1u)) continue;
                    for (int64_t y = (int64_t)(j * C2_RTHREADS + t) - 2; pos + y >= bpos && seen[j] < 2u; --y) {
                        const unsigned long long p = y >= 0 ? srec[y] : recs[pos + y];
                        if (((p ^ rec[j]) & read_mask) != 0ull) break;      // the run of the read begins here
                        if (p == rec[j]) ++seen[j];
                    }
                }
            }
            C2_STAMP(3);   // look back
            // ---- first and second occurrences: the k-mer's table entry (probe chains of the thread's records in lockstep: their
            // dependent LDS round trips overlap instead of adding up), then the add
            uint32_t th[RJ], tact = 0, made = 0;
#pragma unroll
            for (int j = 0; j < RJ; ++j) { th[j] = cf_c2_hash_tab(rec[j] >> rb) & (C2_TAB - 1); if (((act >> j) & 1u) && seen[j] < 2u) tact |= 1u << j; }
            const uint32_t want = tact;
            for (int probe = 0; probe < C2_TAB && __any(tact != 0u); ++probe) {
                unsigned long long cur[RJ];
#pragma unroll
                for (int j = 0; j < RJ; ++j) cur[j] = ((tact >> j) & 1u) ? tkey[th[j]] : 1ull;
#pragma unroll
                for (int j = 0; j < RJ; ++j)
                    if (((tact >> j) & 1u) && cur[j] == 0ull) { cur[j] = atomicCAS(&tkey[th[j]], 0ull, (rec[j] >> rb) + 1ull); if (cur[j] == 0ull) { ++made; cur[j] = (rec[j] >> rb) + 1ull; } }
#pragma unroll
                for (int j = 0; j < RJ; ++j) {
                    if (!((tact >> j) & 1u)) continue;
                    if (cur[j] == (rec[j] >> rb) + 1ull) tact &= ~(1u << j);
                    else th[j] = (th[j] + 1) & (C2_TAB - 1);
                }
            }
            if (tact) atomicOr(&counters[2], 1ull);       // the bucket holds more k-mers than the table: the caller falls back
#pragma unroll
            for (int j = 0; j < RJ; ++j)
                if (((want & ~tact) >> j) & 1u) atomicAdd(seen[j] ? &tmulti[th[j]] : &tpres[th[j]], 1u);
            for (int d = 32; d >= 1; d >>= 1) made += __shfl_down(made, (unsigned)d);
            if (lane == 0 && made) atomicAdd(&sh[1], made);
            C2_STAMP(4);   // table phase
            __syncthreads();            // the staged tile may be overwritten; (last tile) the table is complete
            C2_STAMP(5);   // wait for the others
        }
        // ---- bucket done: its k-mers go to the table, into the workgroup's current chunk or a fresh one
        const uint32_t n_k = sh[1];
        const unsigned long long c_free = sh64[1], c_end = sh64[2];
        __syncthreads();            // everyone has the count and the chunk before thread 0 changes them
        const bool fresh = n_k > c_end - c_free;
        if (fresh) for (unsigned long long o = c_free + t; o < c_end; o += C2_RTHREADS) if (o < out_cap) { cf_slot z; z.key = 0ull; z.val = 0ull; out[o] = z; }
        if (t == 0) {
            unsigned long long at = c_free;
            if (fresh) { at = atomicAdd(&counters[0], chunk); sh64[2] = at + chunk; }
            sh64[1] = at + n_k;
            sh64[0] = at;            // (the ticket word is free during the flush: base of this bucket)
            sh[1] = 0;
            n_dist += n_k;
        }
        __syncthreads();
        if (n_k) {
            const unsigned long long base = sh64[0];
            uint32_t psum = 0;
            for (int s0 = 0; s0 < C2_TAB; s0 += C2_RTHREADS) {
                const int s = s0 + t;
                const bool occ = tkey[s] != 0ull;
                const unsigned long long m = __ballot(occ);
                uint32_t off = 0;
                if (m) {
                    const int leader = __ffsll((long long)m) - 1;
                    if (lane == leader) off = atomicAdd(&sh[1], (uint32_t)__popcll(m));
                    off = (uint32_t)__shfl((int)off, leader);
                }
                if (occ) {
                    const unsigned long long o = base + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    psum += tpres[s];
                    if (o < out_cap) { cf_slot sl; sl.key = (tkey[s] - 1ull) | CF_OCC; sl.val = (unsigned long long)tpres[s] | ((unsigned long long)tmulti[s] << 32); out[o] = sl; }
                }
            }
            psum_all += psum;       // (one global atomic per wave at the END: 2 million adds to one word take 11 ns each, in a row)
        }
        __syncthreads();
        C2_STAMP(6);   // flush
    }
    }
    for (unsigned long long o = sh64[1] + t; o < sh64[2]; o += C2_RTHREADS) if (o < out_cap) { cf_slot z; z.key = 0ull; z.val = 0ull; out[o] = z; }
    if (t == 0 && n_dist) atomicAdd(&counters[3], n_dist);
    for (int d = 32; d >= 1; d >>= 1) psum_all += __shfl_down(psum_all, (unsigned)d);
    if (lane == 0 && psum_all) atomicAdd(&counters[1], psum_all);
#if defined(CF_C2_STAMPS)
    if (t == 0) for (int i = 0; i < 8; ++i) atomicAdd(&counters[8 + i], st_acc[i]);
#endif
}

// Occurrence counts (reference better_consensus_unit_reconstruction.py:129-137: counts[kmer] += 1 for every
// window): the same bucketed record stream with records that are the k-mer alone (rb = 0), reduced by a per-bucket LDS table
// k-mer -> 64-bit count.  No (k-mer, read) set, no per-read state.  A round of 64 records that are all the same k-mer — the
// HOR's consensus k-mers fill whole tiles — is ONE add of 64 by the first lane (64 adds to one LDS word would serialise).
__global__ void __launch_bounds__(C2_RTHREADS)
cf_c2_reduce_occ_kernel(const unsigned long long* __restrict__ recs, int64_t n, int bits, cf_slot* __restrict__ out, unsigned long long out_cap,
                        unsigned long long chunk, int64_t per, int64_t n_chunks, const int64_t* __restrict__ starts, unsigned long long* __restrict__ counters) {
    unsigned long long* tkey = (unsigned long long*)cf_lds;                    // C2_TAB x (k-mer + 1); 0 = empty
    unsigned long long* tcnt = tkey + C2_TAB;
    unsigned long long* sh64 = tcnt + C2_TAB;                                   // [0] position / flush base [1] next free slot of the chunk [2] end of the chunk
    uint32_t* sh = (uint32_t*)(sh64 + 3);                                       // [0] records of the tile in the bucket [1] k-mers in the table
    const int t = threadIdx.x, lane = t & 63;
    const int64_t n_buckets = (int64_t)1 << bits;
    constexpr int RJ = C2_RTILE / C2_RTHREADS;
    if (t == 0) { sh64[1] = 0; sh64[2] = 0; }
    unsigned long long n_dist = 0;
    while (true) {
        __syncthreads();
        if (t == 0) sh64[0] = atomicAdd(&counters[4], 1ull);
        __syncthreads();
        const int64_t ck = (int64_t)sh64[0];
        if (ck >= n_chunks) break;
        const int64_t b0 = ck * per, b1 = min(n_buckets, b0 + per);
        int64_t pos = starts[ck];
        for (int64_t b = b0; b < b1; ++b) {
            for (int s = t; s < C2_TAB; s += C2_RTHREADS) { tkey[s] = 0ull; tcnt[s] = 0ull; }
            if (t == 0) sh[1] = 0;
            __syncthreads();
            bool more = true;
            while (more) {
                if (t == 0) sh[0] = 0;
                __syncthreads();
                unsigned long long rec[RJ];
                uint32_t th[RJ], act = 0, mine = 0, made = 0;
#pragma unroll
                for (int j = 0; j < RJ; ++j) {
                    const int64_t i = pos + (int64_t)j * C2_RTHREADS + t;
                    const unsigned long long v = i < n ? recs[i] : ~0ull;
                    rec[j] = v; th[j] = 0;
                    if (i < n && (int64_t)cf_c2_bucket(v, bits) == b) { ++mine; act |= 1u << j; th[j] = cf_c2_hash_tab(v) & (C2_TAB - 1); }
                }
                const uint32_t in_bucket = act;
                for (int probe = 0; probe < C2_TAB && __any(act != 0u); ++probe) {      // the k-mer's table entry, probe chains in lockstep
                    unsigned long long cur[RJ];
#pragma unroll
                    for (int j = 0; j < RJ; ++j) cur[j] = ((act >> j) & 1u) ? tkey[th[j]] : 1ull;
#pragma unroll
                    for (int j = 0; j < RJ; ++j)
                        if (((act >> j) & 1u) && cur[j] == 0ull) { cur[j] = atomicCAS(&tkey[th[j]], 0ull, rec[j] + 1ull); if (cur[j] == 0ull) { ++made; cur[j] = rec[j] + 1ull; } }
#pragma unroll
                    for (int j = 0; j < RJ; ++j) {
                        if (!((act >> j) & 1u)) continue;
                        if (cur[j] == rec[j] + 1ull) act &= ~(1u << j);
                        else th[j] = (th[j] + 1) & (C2_TAB - 1);
                    }
                }
                if (act) atomicOr(&counters[2], 1ull);       // the bucket holds more k-mers than the table: the caller falls back
#pragma unroll
                for (int j = 0; j < RJ; ++j) {                // (uniform loop: the ballots inside see the whole wave)
                    const bool have = ((in_bucket >> j) & 1u) && !((act >> j) & 1u);
                    const unsigned long long m = __ballot(have);
                    if (!m) continue;
                    const int leader = __ffsll((long long)m) - 1;
                    const unsigned long long k0 = (unsigned long long)__shfl((long long)rec[j], leader);
                    if (m == ~0ull && __all(rec[j] == k0)) { if (lane == 0) atomicAdd(&tcnt[th[j]], 64ull); }
                    else if (have) atomicAdd(&tcnt[th[j]], 1ull);
                }
                for (int d = 32; d >= 1; d >>= 1) { mine += __shfl_down(mine, (unsigned)d); made += __shfl_down(made, (unsigned)d); }
                if (lane == 0) { if (mine) atomicAdd(&sh[0], mine); if (made) atomicAdd(&sh[1], made); }
                __syncthreads();
                const uint32_t got = sh[0];
                pos += got;
                more = got == C2_RTILE && pos < n;                  // a full tile: the bucket may go on
                __syncthreads();
            }
            // ---- bucket done: its k-mers go to the table, into the workgroup's current chunk or a fresh one (as in cf_c2_reduce_kernel)
            const uint32_t n_k = sh[1];
            const unsigned long long c_free = sh64[1], c_end = sh64[2];
            __syncthreads();
            const bool fresh = n_k > c_end - c_free;
            if (fresh) for (unsigned long long o = c_free + t; o < c_end; o += C2_RTHREADS) if (o < out_cap) { cf_slot z; z.key = 0ull; z.val = 0ull; out[o] = z; }
            if (t == 0) {
                unsigned long long at = c_free;
                if (fresh) { at = atomicAdd(&counters[0], chunk); sh64[2] = at + chunk; }
                sh64[1] = at + n_k;
                sh64[0] = at;
                sh[1] = 0;
                n_dist += n_k;
            }
            __syncthreads();
            if (n_k) {
                const unsigned long long base = sh64[0];
                for (int s0 = 0; s0 < C2_TAB; s0 += C2_RTHREADS) {
                    const int s = s0 + t;
                    const bool occ = tkey[s] != 0ull;
                    const unsigned long long m = __ballot(occ);
                    uint32_t off = 0;
                    if (m) {
                        const int leader = __ffsll((long long)m) - 1;
                        if (lane == leader) off = atomicAdd(&sh[1], (uint32_t)__popcll(m));
                        off = (uint32_t)__shfl((int)off, leader);
                    }
                    if (occ) {
                        const unsigned long long o = base + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (o < out_cap) { cf_slot sl; sl.key = (tkey[s] - 1ull) | CF_OCC; sl.val = tcnt[s]; out[o] = sl; }
                    }
                }
            }
            __syncthreads();
        }
    }
    for (unsigned long long o = sh64[1] + t; o < sh64[2]; o += C2_RTHREADS) if (o < out_cap) { cf_slot z; z.key = 0ull; z.val = 0ull; out[o] = z; }
    if (t == 0 && n_dist) atomicAdd(&counters[3], n_dist);
}

namespace {
struct Bufs2 {
    cf_ctx* ctx;
    std::vector<std::pair<void*, size_t>> v;
    explicit Bufs2(cf_ctx* c) : ctx(c) {}
    ~Bufs2() { for (auto it = v.rbegin(); it != v.rend(); ++it) cf_release(ctx, it->first, it->second); }
    template <class T> int get(T** p, size_t n, const char* what) {
        int rc = cf_alloc_t(ctx, p, n, what);
        if (rc == 0) v.emplace_back((void*)*p, n * sizeof(T));
        return rc;
    }
    void keep(void* p) { for (auto& e : v) if (e.first == p) e.first = nullptr; }
};

template <class K>
int launch_nb(int nb, K&& k) {      // radix bits of a pass -> template instance
    switch (nb) {
        case 1: k(std::integral_constant<int, 1>()); break; case 2: k(std::integral_constant<int, 2>()); break;
        case 3: k(std::integral_constant<int, 3>()); break; case 4: k(std::integral_constant<int, 4>()); break;
        case 5: k(std::integral_constant<int, 5>()); break; case 6: k(std::integral_constant<int, 6>()); break;
        case 7: k(std::integral_constant<int, 7>()); break; case 8: k(std::integral_constant<int, 8>()); break;
        case 9: k(std::integral_constant<int, 9>()); break; default: return -22;
    }
    return 0;
}
}  // namespace

// Returns 0 on success, 1 when the sort-and-reduce path does not apply (caller falls back to the table path), < 0 on error.
// occ = 0: presence / multi table (A1); occ = 1: occurrence counts (records without the read id)
int cf_count_sorted(cf_ctx* ctx, int32_t k, int64_t read_lo, int64_t read_hi, int64_t n_w, int occ) {
    const int64_t R = ctx->n_reads;
    int rb = 1;
    while (((int64_t)1 << rb) < std::max<int64_t>(R, 2)) ++rb;
    if (occ) rb = 0;
    if (2 * k + rb > 62 || rb > 26) return 1;
    // tiles of pass 1
    std::vector<cf_c2_tile> tiles;
    int64_t max_win = 0;      // windows of the longest read
    for (int64_t r = read_lo; r < read_hi; ++r) {
        const int64_t len = ctx->h_read_off[(size_t)r + 1] - ctx->h_read_off[(size_t)r];
        if (len < k) continue;
        const int64_t nw = len - k + 1;
        max_win = std::max(max_win, nw);
        for (int64_t c = 0; c * C2_TILE < nw; ++c) tiles.push_back(cf_c2_tile{(int32_t)r, (int32_t)c});
    }
    if (tiles.size() >= (size_t)1 << 31) return 1;
    const int n_tiles1 = (int)tiles.size();
    // buckets: ~2000 records each, so that a bucket's k-mers fit the LDS table with room to spare — but not at the price
    // of one more pass over all records when up to 5000 per bucket do without it
    int bits = 1, bits_min = 1;
    while (bits < 27 && (n_w >> bits) > 2000) ++bits;
    while (bits_min < 27 && (n_w >> bits_min) > 5000) ++bits_min;
    const int passes_min = (bits_min + C2_MAXBITS - 1) / C2_MAXBITS;
    if ((bits + C2_MAXBITS - 1) / C2_MAXBITS > passes_min) bits = passes_min * C2_MAXBITS;
    if (ctx->count_bits) bits = ctx->count_bits;
    const int n_pass = (bits + C2_MAXBITS - 1) / C2_MAXBITS;
    // the reduce looks back along a read's run of records inside a bucket (~ the read's windows >> bits of them): a few very
    // long sequences in a large input would make those runs thousands long — not reads; the table path takes such input
    if (!occ && !ctx->count_bits && n_w > ((int64_t)1 << 24) && (max_win >> bits) > 256) return 1;
    CF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    cf_free_table(ctx);
    Bufs2 tmp(ctx);
    unsigned long long *d_a = nullptr, *d_b = nullptr;
    cf_c2_tile* d_tiles = nullptr;
    unsigned long long* d_cnt = nullptr;
    CF_TRY(tmp.get(&d_a, (size_t)std::max<int64_t>(n_w, 2), "count records"));
    CF_TRY(tmp.get(&d_b, (size_t)std::max<int64_t>(n_w, 2), "count records (pong)"));
    CF_TRY(tmp.get(&d_tiles, tiles.size() + 1, "count tiles"));
    CF_TRY(tmp.get(&d_cnt, 16, "count counters"));
    if (n_tiles1) CF_HIP(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(cf_c2_tile), hipMemcpyHostToDevice, ctx->stream));
    CF_HIP(hipMemsetAsync(d_cnt, 0, 64, ctx->stream));
    const int max_grid = std::max(1, ctx->n_cu) * 8;
    unsigned long long *src = nullptr, *dst = d_a;
    int64_t n_rec = n_w;
    (void)hipEventRecord(ctx->ev2, ctx->stream);
    for (int p = 0; p < n_pass && n_w > 0; ++p) {
        // LSD: low digit first; stable passes leave the array sorted by the whole bucket number
        const int shift = p * C2_MAXBITS, nb = std::min(C2_MAXBITS, bits - shift);
        const int n_tiles = p == 0 ? n_tiles1 : (int)std::max<int64_t>(1, (n_rec + C2_TILE - 1) / C2_TILE);
        const int64_t nh = (int64_t)n_tiles << nb;
        uint32_t* d_hist = nullptr;
        int64_t* d_offs = nullptr;
        Bufs2 pass(ctx);
        CF_TRY(pass.get(&d_hist, (size_t)nh + 1, "count histogram"));
        CF_TRY(pass.get(&d_offs, (size_t)nh + 1, "count offsets"));
        const int grid = std::min(n_tiles, max_grid);
        const size_t lds_h = ((((size_t)4 << nb) + 15) & ~(size_t)15) + (p == 0 ? C2_TILE + 64 : 0);
        const size_t lds_s = p == 0 ? ((size_t)8 << nb) + 2 * ((size_t)4 << nb) + 32 + 16 + (size_t)C2_TILE * (8 + 2) + C2_TILE + 64
                                    : ((size_t)8 << nb) + (size_t)(C2_THREADS / 64 + 1) * ((size_t)4 << nb) + 32 + 16 + (size_t)C2_TILE * 8;
        if (p == 0)
            hipLaunchKernelGGL(cf_c2_hist1_kernel, dim3((unsigned)grid), dim3(C2_THREADS), lds_h, ctx->stream, (const uint8_t*)ctx->d_bases, ctx->n_bases,
                               (const int64_t*)ctx->d_read_off, (const cf_c2_tile*)d_tiles, n_tiles, (int)k, rb, bits, shift, nb, d_hist);
        else
            hipLaunchKernelGGL(cf_c2_hist_kernel, dim3((unsigned)grid), dim3(C2_THREADS), lds_h, ctx->stream, (const unsigned long long*)src, n_rec, n_tiles, rb, bits,
                               shift, nb, d_hist);
        CF_KERNEL_CHECK("cf_c2_hist");
        int64_t n_made = 0;
        {
            const int D = 1 << nb, n_chunks = (n_tiles + C2_SCAN_CHUNK - 1) / C2_SCAN_CHUNK;
            uint32_t* d_part = nullptr;
            int64_t* d_base = nullptr;
            CF_TRY(pass.get(&d_part, (size_t)n_chunks * D + 1, "count column sums"));
            CF_TRY(pass.get(&d_base, (size_t)n_chunks * D + 2, "count column bases"));
            int64_t* d_total = d_base + (size_t)n_chunks * D;
            hipLaunchKernelGGL(cf_c2_colsum_kernel, dim3((unsigned)n_chunks), dim3(C2_SCAN_THREADS), 0, ctx->stream, (const uint32_t*)d_hist, n_tiles, D, d_part);
            hipLaunchKernelGGL(cf_c2_colbase_kernel, dim3(1), dim3(C2_SCAN_THREADS), 64, ctx->stream, (const uint32_t*)d_part, n_chunks, D, d_base, d_total);
            hipLaunchKernelGGL(cf_c2_coloffs_kernel, dim3((unsigned)n_chunks), dim3(C2_SCAN_THREADS), 0, ctx->stream, (const uint32_t*)d_hist, n_tiles, D, (const int64_t*)d_base, d_offs);
            CF_KERNEL_CHECK("cf_c2_col*");
            CF_HIP(hipMemcpyAsync(&n_made, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
            CF_HIP(hipStreamSynchronize(ctx->stream));
        }
        if (p == 0) n_rec = n_made;         // windows holding other symbols than A, C, G, T make no record
        const int rc = launch_nb(nb, [&](auto NB) {
            if (p == 0)
                hipLaunchKernelGGL((cf_c2_scatter1_kernel<decltype(NB)::value>), dim3((unsigned)grid), dim3(C2_THREADS), lds_s, ctx->stream, (const uint8_t*)ctx->d_bases, ctx->n_bases,
                                   (const int64_t*)ctx->d_read_off, (const cf_c2_tile*)d_tiles, n_tiles, (int)k, rb, bits, shift, (const int64_t*)d_offs, dst);
            else
                hipLaunchKernelGGL((cf_c2_scatter_kernel<decltype(NB)::value>), dim3((unsigned)grid), dim3(C2_THREADS), lds_s, ctx->stream, (const unsigned long long*)src,
                                   n_rec, n_tiles, rb, bits, shift, (const int64_t*)d_offs, dst);
        });
        if (rc) return cf_fail(ctx, rc, "cf_count_kmers: bad radix width");
        CF_KERNEL_CHECK("cf_c2_scatter");
        CF_HIP(hipStreamSynchronize(ctx->stream));
        src = dst;
        dst = (src == d_a) ? d_b : d_a;
    }
    // reduce into the free buffer (2 records = 1 slot); when more k-mers turn up than fit, an exact-size table is taken
    unsigned long long h_cnt[4] = {0, 0, 0, 0};
    cf_slot* d_out = (cf_slot*)dst;
    unsigned long long out_cap = (unsigned long long)std::max<int64_t>(n_w, 2) / 2;
    size_t out_bytes = (size_t)std::max<int64_t>(n_w, 2) * 8;
    const size_t lds_r = (size_t)C2_RTILE * 8 + (size_t)C2_TAB * (8 + 4 + 4) + 24 + 16;
    cf_slot* d_exact = nullptr;
    const int64_t n_buckets = (int64_t)1 << bits;
    const int64_t per = std::max<int64_t>(1, n_buckets / ((int64_t)std::max(1, ctx->n_cu) * 64));     // ~16 k chunks
    const int64_t n_chunks = (n_buckets + per - 1) / per;
    // first record of every chunk (occurrence counts) / of every bucket (presence table: its tiles take their lengths from it)
    const int64_t per_s = occ ? per : 1, n_starts = occ ? n_chunks : n_buckets;
    int64_t* d_starts = nullptr;
    CF_TRY(tmp.get(&d_starts, (size_t)n_starts + 2, "count bucket starts"));
    if (n_w > 0) {
        hipLaunchKernelGGL(cf_c2_chunk_starts_kernel, dim3((unsigned)((n_starts + 256) / 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)src, n_rec, rb, bits,
                           per_s, n_starts, d_starts);
        CF_KERNEL_CHECK("cf_c2_chunk_starts_kernel");
    }
    for (int attempt = 0; attempt < 2 && n_w > 0; ++attempt) {
        CF_HIP(hipMemsetAsync(d_cnt, 0, 128, ctx->stream));
        const int grid = (int)std::min<int64_t>(n_chunks, (int64_t)std::max(1, ctx->n_cu) * 2);
        const unsigned long long chunk = (unsigned long long)std::min<int64_t>(32768, std::max<int64_t>(C2_TAB, n_w / grid / 4));
        if (occ)
            hipLaunchKernelGGL(cf_c2_reduce_occ_kernel, dim3((unsigned)grid), dim3(C2_RTHREADS), (size_t)C2_TAB * 16 + 24 + 16, ctx->stream, (const unsigned long long*)src, n_rec, bits, d_out,
                               out_cap, chunk, per, n_chunks, (const int64_t*)d_starts, d_cnt);
        else
            hipLaunchKernelGGL(cf_c2_reduce_kernel, dim3((unsigned)grid), dim3(C2_RTHREADS), lds_r, ctx->stream, (const unsigned long long*)src, n_rec, rb, bits, d_out,
                               out_cap, chunk, per, n_chunks, (const int64_t*)d_starts, d_cnt);
        CF_KERNEL_CHECK("cf_c2_reduce_kernel");
        CF_HIP(hipMemcpy(h_cnt, d_cnt, 32, hipMemcpyDeviceToHost));
#if defined(CF_C2_STAMPS)
        { unsigned long long st[8]; if (hipMemcpy(st, d_cnt + 8, 64, hipMemcpyDeviceToHost) == hipSuccess)
            std::fprintf(stderr, "[cf_c2 stamps] ticket=%llu clear=%llu stage=%llu lookback=%llu table=%llu wait=%llu flush=%llu (shader cycles over %d workgroups)\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6], grid); }
#endif
        if (h_cnt[2] & 1ull) { if (d_exact) cf_release(ctx, d_exact, out_bytes); return 1; }     // a bucket with too many k-mers for the LDS table: table path
        if (h_cnt[0] <= out_cap) break;
        if (attempt == 1) { if (d_exact) cf_release(ctx, d_exact, out_bytes); return cf_fail(ctx, -5, "cf_count_kmers: reduce output overflow"); }
        // (chunks are handed out dynamically: the next run may leave other holes) what was written + two chunks per workgroup
        out_cap = h_cnt[3] + 2ull * chunk * (unsigned long long)grid; out_bytes = (size_t)out_cap * sizeof(cf_slot);
        CF_TRY(cf_alloc(ctx, (void**)&d_exact, out_bytes, "k-mer table (dense)"));
        d_out = d_exact;
    }
    (void)hipEventRecord(ctx->ev3, ctx->stream);
    // hand the dense table to the context
    if (n_w > 0) {
        if (!d_exact) tmp.keep(dst);
        ctx->d_table = d_out;
        ctx->table_alloc = out_bytes / sizeof(cf_slot);
    } else {
        CF_TRY(cf_alloc(ctx, (void**)&ctx->d_table, 16 * sizeof(cf_slot), "k-mer table (dense)"));
        ctx->table_alloc = 16;
    }
    ctx->table_cap = h_cnt[0];
    ctx->table_dense = true;
    ctx->k = k;
    ctx->stats.n_windows = n_w;
    ctx->stats.n_read_kmers = occ ? (int64_t)h_cnt[3] : (int64_t)h_cnt[1];      // (occurrence mode: the distinct k-mers)
    CF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    CF_HIP(hipEventSynchronize(ctx->ev1));
    (void)hipEventElapsedTime(&ctx->times.count_ms, ctx->ev0, ctx->ev1);
    (void)hipEventElapsedTime(&ctx->times.count_kernel_ms, ctx->ev2, ctx->ev3);
    return 0;
}
