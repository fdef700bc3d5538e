/*
 * tools/place_stats.c — developer tool: replays a finished greedy placement (reference read_placer.py:42-94,
 * cloud_contig.py:26-41, :87-95) in its own order and reports the shape of the work one greedy iteration does:
 * entries laid down, events raised, postings visited, score rows touched and how far their offsets lie from the
 * offset the read is finally placed at.  No arg-max: the order is given.  Not part of the product or the oracle.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
typedef struct { uint64_t* k; uint32_t* v; uint64_t cap, n; } map_t;
static void map_init(map_t* m, uint64_t cap) { m->cap = cap; m->n = 0; m->k = calloc(cap, 8); m->v = calloc(cap, 4); }
static int64_t map_slot(map_t* m, uint64_t key, int* fresh) {
    if ((m->n + 1) * 2 > m->cap) {
        map_t b; map_init(&b, m->cap * 2);
        for (uint64_t s = 0; s < m->cap; ++s) if (m->k[s]) { uint64_t h = mix64(m->k[s] - 1) & (b.cap - 1); while (b.k[h]) h = (h + 1) & (b.cap - 1); b.k[h] = m->k[s]; b.v[h] = m->v[s]; }
        b.n = m->n; free(m->k); free(m->v); *m = b;
    }
    uint64_t h = mix64(key) & (m->cap - 1);
    while (m->k[h] && m->k[h] != key + 1) h = (h + 1) & (m->cap - 1);
    *fresh = 0;
    if (!m->k[h]) { m->k[h] = key + 1; m->v[h] = 0; m->n++; *fresh = 1; }
    return (int64_t)h;
}

#define NB 24
static void hadd(int64_t* h, int64_t v) { int b = 0; while ((1ll << b) <= v && b < NB - 1) ++b; h[b]++; }      /* bucket b: [2^(b-1), 2^b) ; 0 -> 0 */
static void hprint(FILE* f, const char* name, const int64_t* h) {
    fprintf(f, "  \"%s\": [", name);
    int last = NB - 1; while (last > 0 && !h[last]) --last;
    for (int b = 0; b <= last; ++b) fprintf(f, "%s%lld", b ? ", " : "", (long long)h[b]);
    fprintf(f, "],\n");
}

/* order[t] = read placed t-th over all stages (prefix first), pos[t] its offset (-1 = None), cls per read */
int place_stats(int64_t n_reads, int64_t n_kmers, const uint8_t* cls, const int64_t* unit_ptr, const int64_t* cloud_ptr, const int32_t* entries,
                const int64_t* order, const int64_t* pos, int thr, const char* out_path) {
    FILE* f = fopen(out_path, "w");
    if (!f) return -1;
    map_t cnt; map_init(&cnt, 1 << 20);
    uint8_t* freq = calloc(n_kmers + 1, 1);
    int64_t* final_off = malloc(8 * n_reads);
    for (int64_t r = 0; r < n_reads; ++r) final_off[r] = -1;
    for (int64_t t = 0; t < n_reads; ++t) final_off[order[t]] = pos[t];
    int64_t h_entries[NB] = {0}, h_events[NB] = {0}, h_hits[NB] = {0}, h_post[NB] = {0}, h_dirty[NB] = {0}, h_units[NB] = {0}, h_rows_per_read[NB] = {0};
    int64_t n_add_first = 0, n_add_event = 0, n_add_later = 0, n_add_before = 0, n_dup_x = 0, n_iter = 0, n_seed = 0, n_seed_hits = 0;
    int64_t d_hits[65] = {0}, d_rows[65] = {0}, d_first[65] = {0};      /* offset - final offset, clamped to [-32, 32]; weighted by hits / by rows; relative to the first-touched offset */
    int64_t far_hits = 0, far_rows = 0, n_rows = 0, n_hits_total = 0, max_post = 0, max_units = 0, n_nonmono = 0;
    int64_t t = 0;
    /* prefix reads */
    for (; t < n_reads && cls[order[t]] == 0; ++t) {
        const int64_t r = order[t];
        for (int64_t u = unit_ptr[r]; u < unit_ptr[r + 1]; ++u)
            for (int64_t e = cloud_ptr[u]; e < cloud_ptr[u + 1]; ++e) {
                int fresh; const int64_t s = map_slot(&cnt, ((uint64_t)(u - unit_ptr[r]) << 32) | (uint32_t)entries[e], &fresh);
                if (++cnt.v[s] == (uint32_t)thr) freq[entries[e]] = 1;
            }
    }
    int64_t* pptr = malloc(8 * (n_kmers + 2));
    int64_t* stamp = malloc(8 * n_reads);
    int64_t* first_off = malloc(8 * n_reads);
    for (int stage = 1; stage <= 2; ++stage) {
        memset(pptr, 0, 8 * (n_kmers + 2));
        int64_t n_stage = 0;
        for (int64_t r = 0; r < n_reads; ++r) {
            stamp[r] = -1; first_off[r] = -1;
            if (cls[r] != stage) continue;
            ++n_stage;
            hadd(h_units, unit_ptr[r + 1] - unit_ptr[r]);
            if (unit_ptr[r + 1] - unit_ptr[r] > max_units) max_units = unit_ptr[r + 1] - unit_ptr[r];
            for (int64_t e = cloud_ptr[unit_ptr[r]]; e < cloud_ptr[unit_ptr[r + 1]]; ++e) pptr[entries[e] + 2]++;
        }
        if (!n_stage) continue;
        for (int64_t x = 0; x < n_kmers; ++x) { if (pptr[x + 2]) hadd(h_post, pptr[x + 2]); if (pptr[x + 2] > max_post) max_post = pptr[x + 2]; pptr[x + 2] += pptr[x + 1]; }
        const int64_t n_post = pptr[n_kmers + 1];
        int64_t* post_r = malloc(8 * (n_post + 1)); int64_t* post_i = malloc(8 * (n_post + 1));
        for (int64_t r = 0; r < n_reads; ++r) {
            if (cls[r] != stage) continue;
            for (int64_t u = unit_ptr[r]; u < unit_ptr[r + 1]; ++u)
                for (int64_t e = cloud_ptr[u]; e < cloud_ptr[u + 1]; ++e) { const int64_t at = pptr[entries[e] + 1]++; post_r[at] = r; post_i[at] = u - unit_ptr[r]; }
        }
        map_t score; map_init(&score, 1 << 20);      /* (read, off) -> s1 ; s0 through `seen` */
        map_t seen; map_init(&seen, 1 << 20);
        int64_t* row_s0 = NULL; int64_t row_cap = 0;
        /* seed: every (x, q) in the contig with x frequent */
        for (uint64_t s = 0; s < cnt.cap; ++s) {
            if (!cnt.k[s]) continue;
            const uint64_t key = cnt.k[s] - 1; const int64_t x = (int64_t)(uint32_t)key, q = (int64_t)(key >> 32);
            if (!freq[x]) continue;
            ++n_seed;
            for (int64_t p = pptr[x]; p < pptr[x + 1]; ++p) {
                const int64_t r = post_r[p], i = post_i[p];
                if (q < i) continue;
                int fresh; const int64_t sl = map_slot(&score, ((uint64_t)r << 32) | (uint64_t)(q - i), &fresh);
                score.v[sl]++; ++n_seed_hits;
                if (first_off[r] < 0) first_off[r] = q - i;
            }
        }
        (void)row_s0; (void)row_cap;
        for (; t < n_reads && cls[order[t]] == stage; ++t) {
            const int64_t r = order[t], off = pos[t];
            if (off < 0) continue;
            ++n_iter;
            int64_t n_e = 0, n_ev = 0, n_h = 0, n_dirty = 0;
            for (int64_t u = unit_ptr[r]; u < unit_ptr[r + 1]; ++u) {
                const int64_t q = off + (u - unit_ptr[r]);
                for (int64_t e = cloud_ptr[u]; e < cloud_ptr[u + 1]; ++e) {
                    const int64_t x = entries[e];
                    ++n_e;
                    int fresh; const int64_t s = map_slot(&cnt, ((uint64_t)q << 32) | (uint64_t)x, &fresh);
                    const uint32_t c = ++cnt.v[s];
                    if (c == 1) ++n_add_first; else if (c < (uint32_t)thr) ++n_add_before; else if (c > (uint32_t)thr) ++n_add_later;
                    if (c != (uint32_t)thr) continue;
                    ++n_add_event; ++n_ev; freq[x] = 1;
                    for (int64_t p = pptr[x]; p < pptr[x + 1]; ++p) {
                        const int64_t r2 = post_r[p], i2 = post_i[p];
                        if (q < i2) continue;
                        int fr2; const int64_t sl = map_slot(&score, ((uint64_t)r2 << 32) | (uint64_t)(q - i2), &fr2);
                        score.v[sl]++; ++n_h;
                        if (first_off[r2] < 0) first_off[r2] = q - i2;
                        if (stamp[r2] != t) { stamp[r2] = t; ++n_dirty; }
                    }
                }
            }
            hadd(h_entries, n_e); hadd(h_events, n_ev); hadd(h_hits, n_h); hadd(h_dirty, n_dirty);
            n_hits_total += n_h;
        }
        /* rows of the stage: offsets relative to the final / first-touched offset of their read */
        int64_t* rows_of = calloc(n_reads, 8);
        for (uint64_t s = 0; s < score.cap; ++s) {
            if (!score.k[s]) continue;
            const uint64_t key = score.k[s] - 1; const int64_t r = (int64_t)(key >> 32), o = (int64_t)(uint32_t)key;
            ++n_rows; rows_of[r]++;
            if (final_off[r] >= 0) {
                int64_t d = o - final_off[r];
                if (d < -32 || d > 32) { far_hits += score.v[s]; ++far_rows; } else { d_hits[d + 32] += score.v[s]; d_rows[d + 32]++; }
            }
            int64_t d1 = o - first_off[r]; if (d1 < -32) d1 = -32; if (d1 > 32) d1 = 32; d_first[d1 + 32] += score.v[s];
        }
        for (int64_t r = 0; r < n_reads; ++r) if (cls[r] == stage) hadd(h_rows_per_read, rows_of[r]);
        free(rows_of); free(post_r); free(post_i); free(score.k); free(score.v); free(seen.k); free(seen.v);
    }
    /* k-mers occurring more than once inside one read */
    {
        int64_t* last = malloc(8 * (n_kmers + 1));
        for (int64_t x = 0; x <= n_kmers; ++x) last[x] = -1;
        for (int64_t r = 0; r < n_reads; ++r)
            for (int64_t e = cloud_ptr[unit_ptr[r]]; e < cloud_ptr[unit_ptr[r + 1]]; ++e) { if (last[entries[e]] == r) ++n_dup_x; last[entries[e]] = r; }
        free(last);
    }
    fprintf(f, "{\n  \"reads\": %lld, \"kmers\": %lld, \"entries\": %lld, \"iterations\": %lld,\n", (long long)n_reads, (long long)n_kmers, (long long)cloud_ptr[unit_ptr[n_reads]], (long long)n_iter);
    fprintf(f, "  \"histograms_are\": \"bucket b counts values in [2^(b-1), 2^b), bucket 0 = value 0\",\n");
    hprint(f, "entries_per_iteration", h_entries); hprint(f, "events_per_iteration", h_events); hprint(f, "hits_per_iteration", h_hits);
    hprint(f, "reads_hit_per_iteration", h_dirty); hprint(f, "postings_per_kmer", h_post); hprint(f, "units_per_read", h_units); hprint(f, "score_rows_per_read", h_rows_per_read);
    fprintf(f, "  \"max_postings_per_kmer\": %lld, \"max_units_per_read\": %lld,\n", (long long)max_post, (long long)max_units);
    fprintf(f, "  \"contig_adds\": {\"first\": %lld, \"below_thr\": %lld, \"event\": %lld, \"later\": %lld},\n", (long long)n_add_first, (long long)n_add_before, (long long)n_add_event, (long long)n_add_later);
    fprintf(f, "  \"seed_events\": %lld, \"seed_hits\": %lld, \"hits\": %lld, \"score_rows\": %lld, \"entries_with_kmer_repeated_in_read\": %lld, \"nonmono\": %lld,\n", (long long)n_seed, (long long)n_seed_hits, (long long)n_hits_total, (long long)n_rows, (long long)n_dup_x, (long long)n_nonmono);
    fprintf(f, "  \"rows_beyond_32_of_final\": %lld, \"hits_beyond_32_of_final\": %lld,\n", (long long)far_rows, (long long)far_hits);
    fprintf(f, "  \"hits_by_offset_minus_final\": ["); for (int i = 0; i < 65; ++i) fprintf(f, "%s%lld", i ? ", " : "", (long long)d_hits[i]); fprintf(f, "],\n");
    fprintf(f, "  \"rows_by_offset_minus_final\": ["); for (int i = 0; i < 65; ++i) fprintf(f, "%s%lld", i ? ", " : "", (long long)d_rows[i]); fprintf(f, "],\n");
    fprintf(f, "  \"hits_by_offset_minus_first_touched\": ["); for (int i = 0; i < 65; ++i) fprintf(f, "%s%lld", i ? ", " : "", (long long)d_first[i]); fprintf(f, "]\n}\n");
    fclose(f);
    free(cnt.k); free(cnt.v); free(freq); free(final_off); free(pptr); free(stamp); free(first_off);
    return 0;
}
