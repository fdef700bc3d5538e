/* TEST INFRASTRUCTURE — plain-C restatement of the read-recruitment test of the reference
 * (scripts/read_recruitment/rr.cpp:73-90): a read is recruited when the edit distance between the HOR unit (forward or
 * reverse complement) and SOME substring of the read (edlib's EDLIB_MODE_HW: gaps before and after the unit inside the
 * read are free) is at most the threshold.
 *
 * The algorithm is the published one the reference's dependency (vendored edlib, Sosic & Sikic 2017) implements:
 * Myers' bit-vector edit distance (Myers 1999) in Hyyro's block formulation — the unit is cut into 64-row blocks, each
 * keeps the vertical deltas of its rows as two bit vectors (Pv: +1, Mv: -1) and passes the horizontal delta of its
 * last row (-1, 0, +1) to the block below.  HW mode = the top row is all zeros (horizontal delta into block 0 is
 * always 0) and the answer is the minimum of the bottom row.  Only tests/, smoke() and bench cpu legs may use this.
 * Pinned against the reference's own code (oracle/_ref/librr_ref.so) and tests/golden/rr_vectors.json. */
#include <stdint.h>
#include <stdlib.h>

typedef unsigned long long u64;

/* one block, one text character: returns the horizontal delta at row `last` of the block (0..63) */
static inline int block_step(u64* pv, u64* mv, u64 eq, int hin, int last) {
    const u64 neg = hin < 0 ? 1ull : 0ull, posb = hin > 0 ? 1ull : 0ull;
    const u64 Pv = *pv, Mv = *mv;
    const u64 xv = eq | Mv;
    eq |= neg;
    const u64 xh = (((eq & Pv) + Pv) ^ Pv) | eq;
    u64 ph = Mv | ~(xh | Pv);
    u64 mh = Pv & xh;
    const int hout = (int)((ph >> last) & 1ull) - (int)((mh >> last) & 1ull);
    ph = (ph << 1) | posb;
    mh = (mh << 1) | neg;
    *pv = mh | ~(xv | ph);
    *mv = ph & xv;
    return hout;
}

/* minimum over all substrings of text of the edit distance to pattern; -1 when it exceeds k (k < 0: no limit) */
int cfo_rr_distance(const unsigned char* pattern, int m, const unsigned char* text, long long n, int k) {
    if (m <= 0) return 0;
    const int nb = (m + 63) / 64;
    u64* peq = (u64*)calloc((size_t)256 * nb, sizeof(u64));
    u64* pv = (u64*)malloc((size_t)nb * sizeof(u64));
    u64* mv = (u64*)calloc((size_t)nb, sizeof(u64));
    if (!peq || !pv || !mv) { free(peq); free(pv); free(mv); return -2; }
    for (int i = 0; i < m; ++i) peq[(size_t)pattern[i] * nb + i / 64] |= 1ull << (i % 64);
    for (int b = 0; b < nb; ++b) pv[b] = ~0ull;
    long long score = m, best = m;      /* before any text character the bottom row holds m */
    for (long long j = 0; j < n; ++j) {
        const u64* eq = peq + (size_t)text[j] * nb;
        int h = 0;                      /* HW: row 0 is all zeros */
        for (int b = 0; b < nb; ++b) h = block_step(&pv[b], &mv[b], eq[b], h, b == nb - 1 ? (m - 1) % 64 : 63);
        score += h;
        if (score < best) best = score;
    }
    free(peq); free(pv); free(mv);
    return (k >= 0 && best > k) ? -1 : (int)best;
}

/* the reference's complement (rr.cpp:11-26: A<->T, C<->G, upper case only) then reversal; 0 on success */
int cfo_rr_revcomp(const unsigned char* s, int m, unsigned char* out) {
    for (int i = 0; i < m; ++i) {
        unsigned char c;
        switch (s[m - 1 - i]) { case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'G': c = 'C'; break; case 'C': c = 'G'; break; default: return -1; }
        out[i] = c;
    }
    return 0;
}
