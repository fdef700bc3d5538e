/*
 * cf_oracle_place.c — plain-C restatement of the greedy cloud-contig placement (A8 + A9) on CSR clouds of integer
 * k-mer indices, for sizes oracle/placer.py (pure Python) cannot reach.  TEST INFRASTRUCTURE ONLY: the checker of the
 * `-m gpu` placement test at thousands of reads.  Nothing in centroflye_amd/ links or loads it.
 *
 * Parity: pinned through oracle/placer.py — tests/test_oracle_golden.py checks this file against it on every fixture
 * (identical lines), and oracle/placer.py against the reference's own read_positions.csv goldens.
 *
 * Reference functions restated (scripts/ of the reference):
 *   CloudContig.add_read          cloud_contig.py:26-41   count(p+i, x) += 1; an event (x, p+i) when it EQUALS the threshold
 *   update_mapping_scores         cloud_contig.py:87-95   for each event (x, q), posting (r, i) of x with q >= i: scores[r][q-i][i] += 1
 *   ReadPlacer.add_prefix_reads   read_placer.py:35-40
 *   ReadPlacer.add_reads          read_placer.py:42-94    seed = every (x, q) with x frequent and q in kmer_positions[x]
 *                                                         (:54-57); best = max (s0, s1, offset), then smallest r_id (:63-78);
 *                                                         none qualifies -> the rest are "None" (:79-84)
 *   ReadPlacer.run                read_placer.py:96-128   prefix reads, internal stage, suffix stage
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

/* open-addressed u64 -> u32 map (key + 1 stored; 0 = empty), grows by doubling */
typedef struct { uint64_t* k; uint32_t* v; uint64_t cap, n; } map_t;
static int map_init(map_t* m, uint64_t cap) {
    m->cap = cap; m->n = 0;
    m->k = (uint64_t*)calloc(cap, 8); m->v = (uint32_t*)calloc(cap, 4);
    return (m->k && m->v) ? 0 : -1;
}
static void map_free(map_t* m) { free(m->k); free(m->v); m->k = NULL; m->v = NULL; }
static int map_grow(map_t* m);
/* returns the slot of key, inserting it with value `init` when absent (*fresh = 1); -1 on out of memory */
static int64_t map_slot(map_t* m, uint64_t key, uint32_t init, int* fresh) {
    if ((m->n + 1) * 2 > m->cap && map_grow(m)) return -1;
    uint64_t h = mix64(key) & (m->cap - 1);
    while (m->k[h] && m->k[h] != key + 1) h = (h + 1) & (m->cap - 1);
    *fresh = 0;
    if (!m->k[h]) { m->k[h] = key + 1; m->v[h] = init; m->n++; *fresh = 1; }
    return (int64_t)h;
}
static int map_grow(map_t* m) {
    map_t b;
    if (map_init(&b, m->cap * 2)) { map_free(&b); return -1; }
    for (uint64_t s = 0; s < m->cap; ++s)
        if (m->k[s]) {
            uint64_t h = mix64(m->k[s] - 1) & (b.cap - 1);
            while (b.k[h]) h = (h + 1) & (b.cap - 1);
            b.k[h] = m->k[s]; b.v[h] = m->v[s];
        }
    b.n = m->n;
    map_free(m);
    *m = b;
    return 0;
}

typedef struct { int64_t* v; int64_t n, cap; } veci;
static int vpush(veci* a, int64_t x) {
    if (a->n == a->cap) {
        int64_t nc = a->cap ? a->cap * 2 : 1024;
        int64_t* nv = (int64_t*)realloc(a->v, 8 * (size_t)nc);
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x;
    return 0;
}

typedef struct {
    int thr;
    map_t count;          /* (pos << 32 | kmer) -> multiplicity */
    uint8_t* freq;        /* per k-mer: frequent at some position */
    int64_t* pos_head;    /* per k-mer: head of the list of positions it was ever added at (index into pos_q / pos_next) */
    veci pos_q, pos_next;
} contig_t;

/* lay read r at `position`; appends the events (kmer, pos) to ev (pairs) */
static int contig_add(contig_t* c, int64_t r, int64_t position, const int64_t* unit_ptr, const int64_t* cloud_ptr, const int32_t* entries, veci* ev) {
    for (int64_t u = unit_ptr[r]; u < unit_ptr[r + 1]; ++u) {
        const int64_t q = position + (u - unit_ptr[r]);
        for (int64_t e = cloud_ptr[u]; e < cloud_ptr[u + 1]; ++e) {
            const int64_t x = entries[e];
            int fresh;
            const int64_t s = map_slot(&c->count, ((uint64_t)q << 32) | (uint64_t)x, 0, &fresh);
            if (s < 0) return -2;
            if (fresh) {      /* kmer_positions[x].add(q) */
                if (vpush(&c->pos_q, q) || vpush(&c->pos_next, c->pos_head[x])) return -2;
                c->pos_head[x] = c->pos_q.n - 1;
            }
            if (++c->count.v[s] == (uint32_t)c->thr) {
                c->freq[x] = 1;
                if (vpush(ev, x) || vpush(ev, q)) return -2;
            }
        }
    }
    return 0;
}

/* read_placer.py:63-78: larger (s0, s1, offset) wins, then the smaller read id */
static int entry_better(int64_t en, int64_t best, const int64_t* s0, const int64_t* s1, const int64_t* s_off, const int64_t* s_read, const int32_t* id_rank) {
    const int64_t v0 = s0[en], v1 = s1[en], b0 = s0[best], b1 = s1[best], bo = s_off[best], o = s_off[en];
    return v0 != b0 ? v0 > b0 : v1 != b1 ? v1 > b1 : o != bo ? o > bo : id_rank[s_read[en]] < id_rank[s_read[best]];
}

/*
 * classes[r]: 0 prefix, 1 internal, 2 suffix; id_rank[r]: rank of the read id in ascending string order.
 * Outputs in the order the reference writes read_positions.csv (each stage's None block by id_rank):
 * out_read[i], out_pos[i] (-1 = None), out_s0[i] (-1 marks a prefix line "r_id 0"), out_s1[i].
 * Returns 0 or -2 (out of memory).
 */
int cfo_place_reads(int64_t n_reads, int64_t n_kmers, const uint8_t* classes, const int32_t* id_rank, const int64_t* unit_ptr,
                    const int64_t* cloud_ptr, const int32_t* entries, int min_cloud_kmer_freq, int min_unit, int min_inters, int min_prop,
                    int64_t* out_read, int64_t* out_pos, int32_t* out_s0, int32_t* out_s1) {
    int rc = 0;
    int64_t n_out = 0;
    contig_t C;
    memset(&C, 0, sizeof C);
    C.thr = min_cloud_kmer_freq < 1 ? 1 : min_cloud_kmer_freq;
    C.freq = (uint8_t*)calloc((size_t)n_kmers + 1, 1);
    C.pos_head = (int64_t*)malloc(8 * (size_t)(n_kmers + 1));
    veci ev = {0, 0, 0};
    int64_t* pptr = (int64_t*)malloc(8 * (size_t)(n_kmers + 2));
    uint8_t* unused = (uint8_t*)calloc((size_t)n_reads + 1, 1);
    int64_t *post_r = NULL, *post_i = NULL;
    if (!C.freq || !C.pos_head || !pptr || !unused || map_init(&C.count, 1 << 16)) { rc = -2; goto done; }
    for (int64_t x = 0; x <= n_kmers; ++x) C.pos_head[x] = -1;
    for (int64_t r = 0; r < n_reads; ++r)
        if (classes[r] == 0) {
            ev.n = 0;
            if ((rc = contig_add(&C, r, 0, unit_ptr, cloud_ptr, entries, &ev))) goto done;
            out_read[n_out] = r; out_pos[n_out] = 0; out_s0[n_out] = -1; out_s1[n_out] = 0; ++n_out;
        }
    for (int cls = 1; cls <= 2; ++cls) {
        /* postings of this stage's reads: k-mer -> (read, unit index inside the read) */
        int64_t n_post = 0, n_unused = 0;
        memset(pptr, 0, 8 * (size_t)(n_kmers + 2));
        for (int64_t r = 0; r < n_reads; ++r) {
            unused[r] = classes[r] == cls;
            if (!unused[r]) continue;
            ++n_unused;
            for (int64_t e = cloud_ptr[unit_ptr[r]]; e < cloud_ptr[unit_ptr[r + 1]]; ++e) pptr[entries[e] + 2]++;
        }
        for (int64_t x = 0; x < n_kmers; ++x) pptr[x + 2] += pptr[x + 1];
        n_post = pptr[n_kmers + 1];
        free(post_r); free(post_i);
        post_r = (int64_t*)malloc(8 * (size_t)(n_post + 1)); post_i = (int64_t*)malloc(8 * (size_t)(n_post + 1));
        if (!post_r || !post_i) { rc = -2; goto done; }
        for (int64_t r = 0; r < n_reads; ++r) {
            if (!unused[r]) continue;
            for (int64_t u = unit_ptr[r]; u < unit_ptr[r + 1]; ++u)
                for (int64_t e = cloud_ptr[u]; e < cloud_ptr[u + 1]; ++e) {
                    const int64_t at = pptr[entries[e] + 1]++;
                    post_r[at] = r; post_i[at] = u - unit_ptr[r];
                }
        }
        /* after the fill pptr[x + 1] = end of x = start of x + 1, so x's postings are [pptr[x], pptr[x + 1]) */
        map_t score = {0, 0, 0, 0}, seen = {0, 0, 0, 0};         /* (read << 32 | offset) -> entry index; (entry << 16 | unit) -> present */
        veci s_read = {0, 0, 0}, s_off = {0, 0, 0}, s0 = {0, 0, 0}, s1 = {0, 0, 0};
        if (map_init(&score, 1 << 16) || map_init(&seen, 1 << 16)) { rc = -2; map_free(&score); map_free(&seen); goto done; }
        /* seed: every (x, q) with x frequent and q in kmer_positions[x] */
        ev.n = 0;
        for (int64_t x = 0; x < n_kmers && !rc; ++x)
            if (C.freq[x])
                for (int64_t p = C.pos_head[x]; p >= 0; p = C.pos_next.v[p])
                    if (vpush(&ev, x) || vpush(&ev, C.pos_q.v[p])) { rc = -2; break; }
        while (!rc && n_unused > 0) {
            for (int64_t j = 0; j + 1 < ev.n && !rc; j += 2) {
                const int64_t x = ev.v[j], q = ev.v[j + 1];
                for (int64_t p = pptr[x]; p < pptr[x + 1]; ++p) {
                    const int64_t r = post_r[p], i = post_i[p];
                    if (q < i) continue;
                    int fresh;
                    const int64_t sl = map_slot(&score, ((uint64_t)r << 32) | (uint64_t)(q - i), (uint32_t)s1.n, &fresh);
                    if (sl < 0) { rc = -2; break; }
                    if (fresh && (vpush(&s_read, r) || vpush(&s_off, q - i) || vpush(&s0, 0) || vpush(&s1, 0))) { rc = -2; break; }
                    const int64_t en = score.v[sl];
                    s1.v[en]++;
                    const int64_t ss = map_slot(&seen, ((uint64_t)en << 16) | (uint64_t)i, 1, &fresh);
                    if (ss < 0) { rc = -2; break; }
                    if (fresh) s0.v[en]++;
                }
            }
            if (rc) break;
            /* arg-max over the live entries; (s0, s1, offset, then smaller id rank) is a strict total order over entries of
             * unused reads (two entries of one read differ in their offset), so the scan may be split over threads: every
             * thread keeps its best, the bests are merged with the same comparison — the result does not depend on the split */
            int64_t best = -1;
#pragma omp parallel if (s1.n > 200000) num_threads(omp_get_max_threads() > 32 ? 32 : omp_get_max_threads())
            {
                int64_t lb = -1;
#pragma omp for schedule(static) nowait
                for (int64_t en = 0; en < s1.n; ++en) {
                    const int64_t r = s_read.v[en];
                    if (!unused[r]) continue;
                    const int64_t v0 = s0.v[en], v1 = s1.v[en];
                    if (!(v0 >= min_unit && v0 * min_prop <= v1 && v1 >= min_inters)) continue;
                    if (lb < 0 || entry_better(en, lb, s0.v, s1.v, s_off.v, s_read.v, id_rank)) lb = en;
                }
#pragma omp critical
                if (lb >= 0 && (best < 0 || entry_better(lb, best, s0.v, s1.v, s_off.v, s_read.v, id_rank))) best = lb;
            }
            if (best < 0) break;
            const int64_t r = s_read.v[best];
            out_read[n_out] = r; out_pos[n_out] = s_off.v[best]; out_s0[n_out] = (int32_t)s0.v[best]; out_s1[n_out] = (int32_t)s1.v[best]; ++n_out;
            ev.n = 0;
            rc = contig_add(&C, r, s_off.v[best], unit_ptr, cloud_ptr, entries, &ev);
            unused[r] = 0; --n_unused;
        }
        if (!rc && n_unused > 0) {      /* the None block, by id rank */
            int64_t* by_rank = (int64_t*)malloc(8 * (size_t)(n_reads + 1));
            if (!by_rank) rc = -2;
            else {
                for (int64_t r = 0; r < n_reads; ++r) by_rank[r] = -1;
                for (int64_t r = 0; r < n_reads; ++r) if (unused[r]) by_rank[id_rank[r]] = r;
                for (int64_t k = 0; k < n_reads; ++k)
                    if (by_rank[k] >= 0) { out_read[n_out] = by_rank[k]; out_pos[n_out] = -1; out_s0[n_out] = 0; out_s1[n_out] = 0; ++n_out; }
                free(by_rank);
            }
        }
        map_free(&score); map_free(&seen);
        free(s_read.v); free(s_off.v); free(s0.v); free(s1.v);
        if (rc) goto done;
    }
done:
    map_free(&C.count);
    free(C.freq); free(C.pos_head); free(C.pos_q.v); free(C.pos_next.v);
    free(ev.v); free(pptr); free(unused); free(post_r); free(post_i);
    return rc;
}
