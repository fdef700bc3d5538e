/*
 * cf_oracle_mt.c — OpenMP restatement of stage 2 of the reference (A1-A6) on packed reads: the same results as
 * cf_oracle.c (which stays the single-thread statement of record), organised so that every phase runs on all host
 * cores.  TEST INFRASTRUCTURE ONLY: the checker of the full-size `-m gpu` parity tests (BASELINE configs[0] / [1]) and
 * the "all cores" leg of bench.py's cpu_baseline.  Nothing in centroflye_amd/ links or loads it.
 *
 * Parity: pinned through cf_oracle.c — tests/test_oracle_golden.py checks this file against it on every fixture (all
 * counters, rare set, clouds, edges, unique mask), and cf_oracle.c against oracle/recruit.py and the reference goldens.
 *
 * Reference functions restated (scripts/ of the reference):
 *   A1 distance_based_kmer_recruitment.py:39-63   per read: sort the windows, one record (k-mer, seen twice) per distinct
 *                                                 k-mer; records are partitioned by a hash of the k-mer, each partition
 *                                                 is sorted and run-length reduced: pres = records, multi = flagged ones
 *   A2 distance_based_kmer_recruitment.py:66-82   rare window (integer bounds lo..hi from the caller)
 *   A3 read_kmer_cloud.py:17-40                   per-unit clouds, units in parallel
 *   A5 distance_based_kmer_recruitment.py:85-128  (a, b, d) histogram, first k-mers a in parallel (dist_cnt[d][a] is a's own)
 *   A6 distance_based_kmer_recruitment.py:131-149 min-coverage + dominance filter
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC (oracle/c/Makefile)
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t n_bases, n_windows, n_read_kmers, n_distinct, n_kept, n_rare, n_units, n_cloud_entries,
        n_emissions, n_edges, n_unique;
    uint64_t edge_checksum, rare_checksum, cloud_checksum;
} cfo_result;

uint64_t cfo_edge_mix(uint64_t d, uint64_t a, uint64_t b, uint64_t cnt);   /* cf_oracle.c */
uint64_t cfo_cloud_mix(uint64_t unit, uint64_t entry);
uint64_t cfo_key_mix(uint64_t key);

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
/* checksum of one table entry (key, pres, multi): the full-table comparison of BASELINE configs[1] */
uint64_t cfo_table_mix(uint64_t key, uint64_t pres, uint64_t multi) { return mix64(mix64(mix64(key + 0x7AB1E) ^ pres) ^ (multi << 1)); }

static int code_of(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return x < y ? -1 : x > y;
}
static int64_t windows(const uint8_t* bases, int64_t b0, int64_t b1, int k, uint64_t* out) {
    int64_t n = b1 - b0 - k + 1, i;
    uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1), code = 0;
    if (n <= 0) return 0;
    for (i = 0; i < b1 - b0; ++i) {
        int c = code_of(bases[b0 + i]);
        if (c < 0) return -1;
        code = ((code << 2) | (uint64_t)c) & mask;
        if (i >= k - 1) out[i - k + 1] = code;
    }
    return n;
}
static int64_t lower_bound(const uint64_t* a, int64_t n, uint64_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (a[m] < x) lo = m + 1; else hi = m; }
    return lo;
}

#define NB_LOG 12
#define NB (1 << NB_LOG)     /* partitions of the A1 records */
#define DUP (1ULL << 63)

typedef struct { uint64_t* v; int64_t n, cap; } vec64;
static int push64(vec64* a, uint64_t x) {
    if (a->n == a->cap) {
        int64_t nc = a->cap ? a->cap * 2 : 1 << 16;
        uint64_t* nv = (uint64_t*)realloc(a->v, 8 * (size_t)nc);
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x;
    return 0;
}

/*
 * State kept between the two halves of stage 2: cfo_mt_prepare runs A1-A3 once (all threads), cfo_mt_dist_part runs
 * A5 + A6 for the first k-mers a with a % n_parts == part (dist_cnt[d][a] is a's own dict in the reference,
 * distance_based_kmer_recruitment.py:108-113, so any subset of the a's is an independent piece of the same result).
 * The full-size parity test and bench.py's cpu_baseline use the partitioned form on the benchmark's own reads.
 */
typedef struct {
    int64_t n_reads, n_rare, U, n_ent;
    int64_t* unit_ptr;       /* copy, n_reads + 1 */
    uint64_t* rare;
    int64_t* cptr;
    int32_t* ent;
    /* postings of the reads [post_min_n, post_max_n), built on first use */
    int64_t post_min_n, post_max_n;
    int64_t* pptr; int32_t* post; int32_t* rend;
} cfo_mt_state;

static double now_s(void) { return omp_get_wtime(); }

void cfo_mt_free(void* h) {
    cfo_mt_state* st = (cfo_mt_state*)h;
    if (!st) return;
    free(st->unit_ptr); free(st->rare); free(st->cptr); free(st->ent); free(st->pptr); free(st->post); free(st->rend);
    free(st);
}

/*
 * A1 + A2 (+ A3 unless stop_after == 1) on n_threads threads (<= 0: all).  secs (optional, 2 doubles): A1 + A2, A3.
 * Returns the state (NULL on failure, *rc_out = -1 non-ACGT, -2 out of memory).
 */
void* cfo_mt_prepare(const uint8_t* bases, const int64_t* read_off, int64_t n_reads, const int64_t* unit_ptr,
                     const int64_t* unit_start, const int64_t* unit_end, int k, int max_nonuniq, uint32_t lo, uint32_t hi,
                     int n_threads, int stop_after, cfo_result* res, uint64_t* table_checksum, double* secs, int* rc_out) {
    int rc = 0;
    int64_t r, max_len = 0, n_w = 0;
    const double t_begin = now_s();
    memset(res, 0, sizeof *res);
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    cfo_mt_state* st = (cfo_mt_state*)calloc(1, sizeof *st);
    if (!st) { *rc_out = -2; return NULL; }
    st->n_reads = n_reads;
    st->post_min_n = st->post_max_n = -1;
    for (r = 0; r < n_reads; ++r) {
        int64_t len = read_off[r + 1] - read_off[r];
        if (len > max_len) max_len = len;
        if (len >= k) n_w += len - k + 1;
    }
    res->n_bases = read_off[n_reads];
    res->n_windows = n_w;
    const int T = n_threads;
    /* ---- A1: per-thread record lists, then per-thread counting sort into NB partitions, then partitions in parallel */
    vec64* recs = (vec64*)calloc((size_t)T, sizeof(vec64));
    int64_t* cnt = (int64_t*)calloc((size_t)T * NB, 8);       /* [thread][partition] */
    int64_t* offs = (int64_t*)calloc((size_t)T * NB, 8);      /* start of the partition inside the thread's sorted records */
    uint64_t** parts = (uint64_t**)calloc((size_t)T, sizeof(uint64_t*));
    vec64* rare_parts = (vec64*)calloc(NB, sizeof(vec64));
    uint64_t* rare = NULL;
    int64_t n_rare = 0;
    if (!recs || !cnt || !offs || !parts || !rare_parts) { rc = -2; goto done; }
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        uint64_t* buf = (uint64_t*)malloc(8 * (size_t)(max_len + 1));
        int bad = buf ? 0 : -2;
#pragma omp for schedule(dynamic, 16)
        for (int64_t rr = 0; rr < n_reads; ++rr) {
            if (bad) continue;
            int64_t n = windows(bases, read_off[rr], read_off[rr + 1], k, buf), i;
            if (n < 0) { bad = -1; continue; }
            qsort(buf, (size_t)n, 8, cmp_u64);
            for (i = 0; i < n;) {
                int64_t j = i + 1;
                while (j < n && buf[j] == buf[i]) ++j;
                if (push64(&recs[t], buf[i] | (j - i > 1 ? DUP : 0))) { bad = -2; break; }
                i = j;
            }
        }
        free(buf);
        if (bad) {
#pragma omp critical
            rc = bad;
        }
    }
    if (rc) goto done;
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        const vec64* a = &recs[t];
        int64_t* c = cnt + (size_t)t * NB;
        int64_t i, s = 0;
        for (i = 0; i < a->n; ++i) c[mix64(a->v[i] & ~DUP) >> (64 - NB_LOG)]++;
        uint64_t* out = (uint64_t*)malloc(8 * (size_t)(a->n + 1));
        int64_t* pos = (int64_t*)malloc(8 * NB);
        if (!out || !pos) {
#pragma omp critical
            rc = -2;
        } else {
            for (i = 0; i < NB; ++i) { pos[i] = s; offs[(size_t)t * NB + i] = s; s += c[i]; }
            for (i = 0; i < a->n; ++i) out[pos[mix64(a->v[i] & ~DUP) >> (64 - NB_LOG)]++] = a->v[i];
        }
        free(pos);
        parts[t] = out;
        free(recs[t].v); recs[t].v = NULL;
    }
    if (rc) goto done;
    {
        int64_t n_rk = 0, n_distinct = 0, n_kept = 0;
        uint64_t tchk = 0;
        for (int t = 0; t < T; ++t) n_rk += recs[t].n;
        res->n_read_kmers = n_rk;
#pragma omp parallel for schedule(dynamic, 4) num_threads(T) reduction(+ : n_distinct, n_kept, tchk)
        for (int b = 0; b < NB; ++b) {
            int64_t n = 0, i;
            for (int t = 0; t < T; ++t) n += cnt[(size_t)t * NB + b];
            uint64_t* v = (uint64_t*)malloc(8 * (size_t)(n + 1));
            if (!v) {
#pragma omp critical
                rc = -2;
                continue;
            }
            n = 0;
            for (int t = 0; t < T; ++t) {
                memcpy(v + n, parts[t] + offs[(size_t)t * NB + b], 8 * (size_t)cnt[(size_t)t * NB + b]);
                n += cnt[(size_t)t * NB + b];
            }
            /* sort by k-mer (the flag bit is the top bit: mask it by sorting on key << 1 | flag) */
            for (i = 0; i < n; ++i) v[i] = (v[i] << 1) | (v[i] >> 63);
            qsort(v, (size_t)n, 8, cmp_u64);
            for (i = 0; i < n;) {
                int64_t j = i;
                uint32_t pres = 0, multi = 0;
                const uint64_t key = v[i] >> 1;
                while (j < n && (v[j] >> 1) == key) { ++pres; multi += (uint32_t)(v[j] & 1); ++j; }
                ++n_distinct;
                tchk += cfo_table_mix(key, pres, multi);
                if (max_nonuniq >= 0 && multi <= (uint32_t)max_nonuniq) {
                    ++n_kept;
                    if (pres >= lo && pres <= hi && push64(&rare_parts[b], key)) {
#pragma omp critical
                        rc = -2;
                    }
                }
                i = j;
            }
            free(v);
        }
        if (rc) goto done;
        res->n_distinct = n_distinct;
        res->n_kept = n_kept;
        if (table_checksum) *table_checksum = tchk;
    }
    for (int b = 0; b < NB; ++b) n_rare += rare_parts[b].n;
    rare = (uint64_t*)malloc(8 * (size_t)(n_rare + 1));
    if (!rare) { rc = -2; goto done; }
    n_rare = 0;
    for (int b = 0; b < NB; ++b) { memcpy(rare + n_rare, rare_parts[b].v, 8 * (size_t)rare_parts[b].n); n_rare += rare_parts[b].n; }
    qsort(rare, (size_t)n_rare, 8, cmp_u64);
    res->n_rare = n_rare;
    for (int64_t i = 0; i < n_rare; ++i) res->rare_checksum += cfo_key_mix(rare[i]);
    st->rare = rare; st->n_rare = n_rare; rare = NULL;
    /* the A1 work arrays are not needed any more */
    for (int t = 0; t < T; ++t) { free(parts[t]); parts[t] = NULL; }
    if (secs) secs[0] = now_s() - t_begin;
    if (stop_after == 1) goto done;
    /* ---- A3: units in parallel, each cloud in its own block, then one CSR */
    {
        const double t_a3 = now_s();
        const int64_t U = unit_ptr[n_reads];
        const uint64_t* rk = st->rare;
        res->n_units = U;
        st->U = U;
        int32_t** uc = (int32_t**)calloc((size_t)U + 1, sizeof(int32_t*));
        int64_t* cptr = (int64_t*)calloc((size_t)U + 1, 8);
        int32_t* ent = NULL;
        st->unit_ptr = (int64_t*)malloc(8 * (size_t)(n_reads + 1));
        if (!uc || !cptr || !st->unit_ptr) { rc = -2; goto done3; }
        memcpy(st->unit_ptr, unit_ptr, 8 * (size_t)(n_reads + 1));
#pragma omp parallel num_threads(T)
        {
            uint64_t* buf = (uint64_t*)malloc(8 * (size_t)(max_len + 1));
            int bad = buf ? 0 : -2;
#pragma omp for schedule(dynamic, 64)
            for (int64_t u = 0; u < U; ++u) {
                if (bad) continue;
                int64_t n = windows(bases, unit_start[u], unit_end[u], k, buf), i, m = 0;
                if (n < 0) { bad = -1; continue; }
                int32_t* e = (int32_t*)malloc(4 * (size_t)(n + 1));
                if (!e) { bad = -2; continue; }
                for (i = 0; i < n; ++i) {
                    int64_t p = lower_bound(rk, n_rare, buf[i]);
                    if (p < n_rare && rk[p] == buf[i]) e[m++] = (int32_t)p;
                }
                qsort(e, (size_t)m, 4, cmp_i32);
                int64_t w = 0;
                for (i = 0; i < m; ++i) if (i == 0 || e[i] != e[i - 1]) e[w++] = e[i];
                uc[u] = e;
                cptr[u + 1] = w;
            }
            free(buf);
            if (bad) {
#pragma omp critical
                rc = bad;
            }
        }
        if (rc) goto done3;
        for (int64_t u = 0; u < U; ++u) cptr[u + 1] += cptr[u];
        const int64_t n_ent = cptr[U];
        ent = (int32_t*)malloc(4 * (size_t)(n_ent + 1));
        if (!ent) { rc = -2; goto done3; }
        {
            uint64_t cchk = 0;
#pragma omp parallel for schedule(static) num_threads(T) reduction(+ : cchk)
            for (int64_t u = 0; u < U; ++u) {
                memcpy(ent + cptr[u], uc[u], 4 * (size_t)(cptr[u + 1] - cptr[u]));
                for (int64_t i = cptr[u]; i < cptr[u + 1]; ++i) cchk += cfo_cloud_mix((uint64_t)u, (uint64_t)ent[i]);
                free(uc[u]); uc[u] = NULL;
            }
            res->cloud_checksum = cchk;
        }
        res->n_cloud_entries = n_ent;
        st->cptr = cptr; st->ent = ent; st->n_ent = n_ent;
        cptr = NULL; ent = NULL;
        if (secs) secs[1] = now_s() - t_a3;
    done3:
        if (uc) for (int64_t u = 0; u < U; ++u) free(uc[u]);
        free(uc); free(cptr); free(ent);
    }
done:
    if (recs) for (int t = 0; t < T; ++t) free(recs[t].v);
    if (parts) for (int t = 0; t < T; ++t) free(parts[t]);
    if (rare_parts) for (int b = 0; b < NB; ++b) free(rare_parts[b].v);
    free(recs); free(cnt); free(offs); free(parts); free(rare_parts); free(rare);
    *rc_out = rc;
    if (rc) { cfo_mt_free(st); return NULL; }
    return st;
}

/* counters of one A5 + A6 run over a partition of the first k-mers */
typedef struct {
    int64_t n_emissions, n_edges, n_unique, n_first_kmers;
    uint64_t edge_checksum;
    double secs;             /* A5 + A6 alone (postings excluded: they are built once per read range) */
    double secs_postings;    /* 0 when the postings of this read range were already there */
} cfo_part_result;

/*
 * A5 + A6 for the first k-mers a % n_parts == part on n_threads threads (<= 0: all).  unique_out (optional, n_rare
 * bytes, ZEROED by the caller or carrying earlier partitions): the k-mers a, b of this partition's edges are set to 1.
 * edges_out (optional): (d, a, b, cnt) rows in no particular order.  Returns 0, -2 (memory), -3 (edges_cap too small).
 */
int cfo_mt_dist_part(void* h, int64_t min_n, int64_t max_n, int min_d, int max_d, uint32_t min_cov, double rel_threshold,
                     int part, int n_parts, int n_threads, cfo_part_result* out, uint8_t* unique_out, uint32_t* edges_out,
                     int64_t edges_cap) {
    cfo_mt_state* st = (cfo_mt_state*)h;
    int rc = 0;
    memset(out, 0, sizeof *out);
    if (!st || !st->cptr || n_parts < 1 || part < 0 || part >= n_parts) return -22;
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    const int T = n_threads;
    const int64_t n_reads = st->n_reads, n_rare = st->n_rare, U = st->U;
    const int64_t* cptr = st->cptr;
    const int32_t* ent = st->ent;
    if (min_n < 0) min_n = 0;
    if (max_n > n_reads) max_n = n_reads;
    if (max_n < min_n) max_n = min_n;
    if (min_d < 1) min_d = 1;
    if (st->post_min_n != min_n || st->post_max_n != max_n) {
        const double t0 = now_s();
        const int64_t u_lo = st->unit_ptr[min_n], u_hi = st->unit_ptr[max_n];
        free(st->pptr); free(st->post); free(st->rend);
        st->pptr = (int64_t*)calloc((size_t)n_rare + 2, 8);
        st->post = (int32_t*)malloc(4 * (size_t)(cptr[u_hi] - cptr[u_lo] + 1));
        st->rend = (int32_t*)malloc(4 * (size_t)(U + 1));
        st->post_min_n = st->post_max_n = -1;
        if (!st->pptr || !st->post || !st->rend) return -2;
        int64_t* pptr = st->pptr;
        for (int64_t e = cptr[u_lo]; e < cptr[u_hi]; ++e) pptr[ent[e] + 2]++;
        for (int64_t i = 0; i < n_rare; ++i) pptr[i + 2] += pptr[i + 1];
        for (int64_t u = u_lo; u < u_hi; ++u)
            for (int64_t e = cptr[u]; e < cptr[u + 1]; ++e) st->post[pptr[ent[e] + 1]++] = (int32_t)u;
        for (int64_t r = 0; r < n_reads; ++r)
            for (int64_t u = st->unit_ptr[r]; u < st->unit_ptr[r + 1]; ++u) st->rend[u] = (int32_t)st->unit_ptr[r + 1];
        st->post_min_n = min_n; st->post_max_n = max_n;
        out->secs_postings = now_s() - t0;
    }
    const int64_t* pptr = st->pptr;
    const int32_t* post = st->post;
    const int32_t* rend = st->rend;
    uint8_t* uniq = (uint8_t*)calloc((size_t)n_rare + 1, 1);
    if (!uniq) return -2;
    const double t1 = now_s();
    int64_t n_em = 0, n_edges = 0, n_first = 0;
    uint64_t echk = 0;
#pragma omp parallel num_threads(T) reduction(+ : n_em, echk, n_first)
    {
        uint64_t hcap = 1 << 12;
        uint64_t* hk = (uint64_t*)malloc(8 * hcap);
        uint32_t* hv = (uint32_t*)malloc(4 * hcap);
        int bad = (hk && hv) ? 0 : -2;
#pragma omp for schedule(dynamic, 64)
        for (int64_t a = part; a < n_rare; a += n_parts) {
            if (bad) continue;
            const int64_t p0 = pptr[a], p1 = pptr[a + 1];
            if (p0 == p1) continue;
            int64_t em = 0;
            for (int64_t p = p0; p < p1; ++p) {
                int32_t g = post[p], jlo = g + min_d, jhi = rend[g] - 1 < g + max_d ? rend[g] - 1 : g + max_d;
                if (jhi >= jlo) em += cptr[jhi + 1] - cptr[jlo];
            }
            if (!em) continue;
            ++n_first;
            uint64_t need = 16;
            while (need < (uint64_t)em * 2) need <<= 1;
            if (need > hcap) {
                free(hk); free(hv);
                hcap = need;
                hk = (uint64_t*)malloc(8 * hcap); hv = (uint32_t*)malloc(4 * hcap);
                if (!hk || !hv) { bad = -2; continue; }
            }
            memset(hk, 0, 8 * need);
            for (int64_t p = p0; p < p1; ++p) {
                int32_t g = post[p], jlo = g + min_d, jhi = rend[g] - 1 < g + max_d ? rend[g] - 1 : g + max_d;
                for (int32_t j = jlo; j <= jhi; ++j) {
                    const uint64_t d = (uint64_t)(j - g);
                    for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) {
                        const uint64_t b = (uint64_t)ent[e];
                        if ((int64_t)b == a) continue;
                        ++n_em;
                        const uint64_t key = ((b << 17) | d) + 1;
                        uint64_t hh = mix64(key) & (need - 1);
                        while (hk[hh] && hk[hh] != key) hh = (hh + 1) & (need - 1);
                        if (!hk[hh]) { hk[hh] = key; hv[hh] = 0; }
                        hv[hh]++;
                    }
                }
            }
            for (uint64_t s = 0; s < need; ++s) {
                if (!hk[s] || hv[s] < min_cov) continue;
                const uint64_t b = (hk[s] - 1) >> 17, d = (hk[s] - 1) & 0x1FFFF;
                uint64_t total = 0;
                for (int dd = min_d; dd <= max_d; ++dd) {
                    const uint64_t key = ((b << 17) | (uint64_t)dd) + 1;
                    uint64_t hh = mix64(key) & (need - 1);
                    while (hk[hh] && hk[hh] != key) hh = (hh + 1) & (need - 1);
                    if (hk[hh]) total += hv[hh];
                }
                if ((double)hv[s] / (double)total >= rel_threshold) {
                    int64_t at;
#pragma omp atomic capture
                    at = n_edges++;
                    if (edges_out) {
                        if (at >= edges_cap) { bad = -3; break; }
                        uint32_t* E = edges_out + 4 * at;
                        E[0] = (uint32_t)d; E[1] = (uint32_t)a; E[2] = (uint32_t)b; E[3] = hv[s];
                    }
                    echk += cfo_edge_mix(d, (uint64_t)a, b, hv[s]);
#pragma omp atomic write
                    uniq[a] = 1;
#pragma omp atomic write
                    uniq[b] = 1;
                }
            }
        }
        free(hk); free(hv);
        if (bad) {
#pragma omp critical
            rc = bad;
        }
    }
    out->secs = now_s() - t1;
    out->n_emissions = n_em;
    out->n_edges = n_edges;
    out->edge_checksum = echk;
    out->n_first_kmers = n_first;
    for (int64_t i = 0; i < n_rare; ++i) out->n_unique += uniq[i];
    if (unique_out && !rc)
        for (int64_t i = 0; i < n_rare; ++i) unique_out[i] |= uniq[i];
    free(uniq);
    return rc;
}

/* read-only views of the state (tests compare them with the device's arrays) */
int64_t cfo_mt_n_rare(const void* h) { return ((const cfo_mt_state*)h)->n_rare; }
int cfo_mt_get(const void* h, uint64_t* rare_out, int64_t* cloud_ptr_out, int32_t* entries_out) {
    const cfo_mt_state* st = (const cfo_mt_state*)h;
    if (rare_out) memcpy(rare_out, st->rare, 8 * (size_t)st->n_rare);
    if (cloud_ptr_out) { if (!st->cptr) return -22; memcpy(cloud_ptr_out, st->cptr, 8 * (size_t)(st->U + 1)); }
    if (entries_out) { if (!st->ent) return -22; memcpy(entries_out, st->ent, 4 * (size_t)st->n_ent); }
    return 0;
}

/*
 * Whole stage 2 on n_threads threads (<= 0: all).  stop_after = 1 ends after A2 (BASELINE configs[1]: count + rare filter).
 * Optional outputs as in cfo_stage2, plus table_checksum (sum of cfo_table_mix over every distinct k-mer).
 * Returns 0, or -1 (non-ACGT), -2 (out of memory), -3 (an output buffer is too small).
 */
int cfo_stage2_mt(const uint8_t* bases, const int64_t* read_off, int64_t n_reads, const int64_t* unit_ptr,
                  const int64_t* unit_start, const int64_t* unit_end, int k, int max_nonuniq, uint32_t lo, uint32_t hi,
                  int64_t min_n, int64_t max_n, int min_d, int max_d, uint32_t min_cov, double rel_threshold,
                  cfo_result* res, uint64_t* rare_out, int64_t rare_cap, int64_t* cloud_ptr_out, int32_t* entries_out,
                  int64_t entries_cap, uint32_t* edges_out, int64_t edges_cap, uint8_t* unique_out,
                  int n_threads, int stop_after, uint64_t* table_checksum) {
    int rc = 0;
    cfo_mt_state* st = (cfo_mt_state*)cfo_mt_prepare(bases, read_off, n_reads, unit_ptr, unit_start, unit_end, k, max_nonuniq, lo, hi,
                                                     n_threads, stop_after, res, table_checksum, NULL, &rc);
    if (!st) return rc;
    if (rare_out) {
        if (rare_cap < st->n_rare) { rc = -3; goto done; }
        memcpy(rare_out, st->rare, 8 * (size_t)st->n_rare);
    }
    if (stop_after == 1) goto done;
    if (cloud_ptr_out) memcpy(cloud_ptr_out, st->cptr, 8 * (size_t)(st->U + 1));
    if (entries_out) {
        if (entries_cap < st->n_ent) { rc = -3; goto done; }
        memcpy(entries_out, st->ent, 4 * (size_t)st->n_ent);
    }
    if (stop_after == 2) goto done;
    {
        cfo_part_result pr;
        if (unique_out) memset(unique_out, 0, (size_t)st->n_rare);
        rc = cfo_mt_dist_part(st, min_n, max_n, min_d, max_d, min_cov, rel_threshold, 0, 1, n_threads, &pr, unique_out, edges_out, edges_cap);
        res->n_emissions = pr.n_emissions;
        res->n_edges = pr.n_edges;
        res->edge_checksum = pr.edge_checksum;
        res->n_unique = pr.n_unique;
    }
done:
    cfo_mt_free(st);
    return rc;
}
