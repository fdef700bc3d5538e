// cf_recruit.hip — read recruitment, the stage before the hot path.
//
// Reference: scripts/read_recruitment/rr.cpp:73-90 — for every read, edlibAlign(unit, read) and
// edlibAlign(revcomp(unit), read) in mode HW (the unit may start and end anywhere inside the read at no cost) with the
// threshold as k; the read is kept when either distance is within k.  edlib (vendored there) implements Myers'
// bit-vector algorithm in Hyyro's block formulation; this is that published algorithm laid out for a wavefront:
//
//   * one WAVE per (read, strand); lane i owns block i of the unit (64 rows: vertical deltas Pv / Mv in two 64-bit
//     registers, the block's match masks for A, C, G, T in eight more).  DXZ1 (2055 bp) = 33 blocks; up to 64.
//   * the column dependency between blocks (the horizontal delta of a block's last row feeds the block below) makes
//     the wave a systolic array: at step t lane i works on text column t - i.  The text character and the horizontal
//     delta travel down the lanes together in one register (one wave shift per step); lane 0 takes its character from
//     a 64-byte chunk the whole wave loaded with one coalesced access.
//   * mode HW: the delta into block 0 is always 0; the answer is the minimum over all columns of the bottom row,
//     tracked by the last lane from the delta at the unit's true last row.
// No LDS, no atomics besides the work ticket; bound by vector instructions (about 45 per text character and strand).
#include "cf_common.h"

#define RR_THREADS 256
#define RR_MAX_BLOCKS 64

struct cf_rr_args {
    const uint8_t* reads;
    const int64_t* read_off;
    int64_t n_items;                     // 2 x reads: item = 2 * read + strand
    const unsigned long long* peq;       // [2 strands][4 bases][RR_MAX_BLOCKS]
    int32_t m, nb, k;
    unsigned long long* ticket;
    int32_t* out;                        // [reads][2]
};

__device__ __forceinline__ uint32_t cf_rr_code(uint32_t c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }

__global__ void __launch_bounds__(RR_THREADS)
cf_rr_kernel(cf_rr_args A) {
    const int lane = threadIdx.x & 63;
    const int last_row = lane == A.nb - 1 ? (A.m - 1) & 63 : 63;
    while (true) {
        unsigned long long it = 0;
        if (lane == 0) it = atomicAdd(A.ticket, 1ull);
        it = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(it >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
        if ((int64_t)it >= A.n_items) break;
        const int64_t r = (int64_t)(it >> 1);
        const int strand = (int)(it & 1ull);
        const int64_t t0 = A.read_off[r], L = A.read_off[r + 1] - t0;
        const unsigned long long* pq = A.peq + (size_t)strand * 4 * RR_MAX_BLOCKS + lane;
        const unsigned long long pA = pq[0], pC = pq[RR_MAX_BLOCKS], pG = pq[2 * RR_MAX_BLOCKS], pT = pq[3 * RR_MAX_BLOCKS];
        unsigned long long pv = ~0ull, mv = 0ull;
        uint32_t pack = 4u | (1u << 8);          // [code : 8 | horizontal delta + 1 : 8] handed to the lane below
        int32_t score = A.m, best = A.m;         // before any text character the bottom row holds m
        const int64_t n_steps = L + A.nb - 1;
        for (int64_t base = 0; base < n_steps; base += 64) {
            const int64_t at = base + lane;
            const uint32_t chunk = at < L ? cf_rr_code(A.reads[t0 + at]) : 4u;
            const int lim = (int)min((int64_t)64, n_steps - base);
            for (int s = 0; s < lim; ++s) {
                const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)chunk, s);
                // lane i takes what lane i - 1 produced in the previous step: a DPP wave shift (one VALU move, no LDS crossbar)
                uint32_t in = (uint32_t)__builtin_amdgcn_update_dpp((int)pack, (int)pack, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                if (lane == 0) in = c0 | (1u << 8);                  // HW: the row above the unit is all zeros
                const uint32_t code = in & 0xFFu;
                const int hin = (int)(in >> 8) - 1;
                const int64_t j = base + s - lane;
                int hout = 0;
                if (lane < A.nb && j >= 0 && j < L) {
                    unsigned long long eq = code == 0u ? pA : code == 1u ? pC : code == 2u ? pG : code == 3u ? pT : 0ull;
                    const unsigned long long neg = hin < 0 ? 1ull : 0ull, posb = hin > 0 ? 1ull : 0ull;
                    const unsigned long long xv = eq | mv;
                    eq |= neg;
                    const unsigned long long xh = (((eq & pv) + pv) ^ pv) | eq;
                    unsigned long long ph = mv | ~(xh | pv);
                    unsigned long long mh = pv & xh;
                    hout = (int)((ph >> last_row) & 1ull) - (int)((mh >> last_row) & 1ull);
                    ph = (ph << 1) | posb;
                    mh = (mh << 1) | neg;
                    pv = mh | ~(xv | ph);
                    mv = ph & xv;
                    score += hout;                                   // only the last lane's score is the bottom row
                    best = min(best, score);
                }
                pack = code | ((uint32_t)(hout + 1) << 8);
            }
        }
        if (lane == A.nb - 1) A.out[2 * r + strand] = (A.k >= 0 && best > A.k) ? -1 : best;
    }
}

extern "C" int cf_rr_distances(cf_ctx* ctx, const uint8_t* unit, int32_t unit_len, const uint8_t* reads, const int64_t* read_off,
                               int64_t n_reads, int32_t threshold, int32_t* dist_fwd, int32_t* dist_rc) {
    if (!ctx) return -22;
    if (!unit || unit_len < 1 || unit_len > 64 * RR_MAX_BLOCKS) return cf_fail(ctx, -22, "cf_rr_distances: the unit must have 1 .. 4096 bases");
    if (n_reads < 0 || (n_reads && (!read_off || !dist_fwd || !dist_rc))) return cf_fail(ctx, -22, "cf_rr_distances: bad arguments");
    // (round 5, tools/fuzz_rr.py: a batch of EMPTY reads has no bytes — `reads` may then be null; the offsets are looked at before they are trusted)
    if (n_reads) {
        if (read_off[0] < 0) return cf_fail(ctx, -22, "cf_rr_distances: negative read offset");
        for (int64_t r = 0; r < n_reads; ++r) if (read_off[r + 1] < read_off[r]) return cf_fail(ctx, -22, "cf_rr_distances: read offsets must not decrease");
        if (read_off[n_reads] > 0 && !reads) return cf_fail(ctx, -22, "cf_rr_distances: bad arguments (no read bytes)");
    }
    // match masks of the unit and of its reverse complement (the reference asserts upper-case ACGT, rr.cpp:11-26)
    std::vector<unsigned long long> peq((size_t)2 * 4 * RR_MAX_BLOCKS, 0ull);
    for (int32_t i = 0; i < unit_len; ++i) {
        const uint8_t c = unit[i];
        const int v = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
        if (v < 0) return cf_fail(ctx, -22, "cf_rr_distances: the unit has a character outside upper-case ACGT");
        peq[(size_t)(0 * 4 + v) * RR_MAX_BLOCKS + (size_t)(i >> 6)] |= 1ull << (i & 63);
        const int32_t ri = unit_len - 1 - i;               // position of the complement in the reverse strand
        peq[(size_t)(1 * 4 + (3 - v)) * RR_MAX_BLOCKS + (size_t)(ri >> 6)] |= 1ull << (ri & 63);
    }
    if (n_reads == 0) return 0;
    CF_HIP(hipSetDevice(ctx->device));
    const int64_t n_bytes = read_off[n_reads];
    uint8_t* d_reads = nullptr; int64_t* d_off = nullptr; unsigned long long *d_peq = nullptr, *d_ticket = nullptr; int32_t* d_out = nullptr;
    int rc = 0;
    std::vector<int32_t> h_out((size_t)2 * n_reads);
    do {
        if ((rc = cf_alloc_t(ctx, &d_reads, (size_t)std::max<int64_t>(n_bytes, 1), "rr reads"))) break;
        if ((rc = cf_alloc_t(ctx, &d_off, (size_t)n_reads + 1, "rr offsets"))) break;
        if ((rc = cf_alloc_t(ctx, &d_peq, peq.size(), "rr match masks"))) break;
        if ((rc = cf_alloc_t(ctx, &d_ticket, 1, "rr ticket"))) break;
        if ((rc = cf_alloc_t(ctx, &d_out, (size_t)2 * n_reads, "rr distances"))) break;
        hipError_t e = hipSuccess;
        if (n_bytes) e = hipMemcpyAsync(d_reads, reads, (size_t)n_bytes, hipMemcpyDefault, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_off, read_off, (size_t)(n_reads + 1) * 8, hipMemcpyDefault, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_peq, peq.data(), peq.size() * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_ticket, 0, 8, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev0, ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_rr_distances copy: ") + hipGetErrorString(e)); break; }
        cf_rr_args A;
        A.reads = d_reads; A.read_off = d_off; A.n_items = 2 * n_reads; A.peq = d_peq; A.m = unit_len; A.nb = (unit_len + 63) / 64;
        A.k = threshold; A.ticket = d_ticket; A.out = d_out;
        const int64_t waves = 2 * n_reads;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((waves + 3) / 4, (int64_t)std::max(1, ctx->n_cu) * 8));
        hipLaunchKernelGGL(cf_rr_kernel, dim3((unsigned)grid), dim3(RR_THREADS), 0, ctx->stream, A);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = cf_fail(ctx, -5, std::string("cf_rr_kernel: ") + hipGetErrorString(e)); break; }
        (void)hipEventElapsedTime(&ctx->times.rr_kernel_ms, ctx->ev0, ctx->ev1);
    } while (0);
    if (d_out) cf_release_t(ctx, d_out, (size_t)2 * n_reads);
    if (d_ticket) cf_release_t(ctx, d_ticket, 1);
    if (d_peq) cf_release_t(ctx, d_peq, peq.size());
    if (d_off) cf_release_t(ctx, d_off, (size_t)n_reads + 1);
    if (d_reads) cf_release_t(ctx, d_reads, (size_t)std::max<int64_t>(n_bytes, 1));
    if (rc) return rc;
    for (int64_t r = 0; r < n_reads; ++r) { dist_fwd[r] = h_out[(size_t)2 * r]; dist_rc[r] = h_out[(size_t)2 * r + 1]; }
    return 0;
}
