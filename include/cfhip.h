/*
 * cfhip.h — C ABI of libcfhip.so: the MI355X (gfx950) device pipeline of centroflye_amd.
 *
 * This is the drop-in boundary of the hot path.  The reference has no FFI:
 * its stage scripts are pure Python, so the entry points below are what a ctypes binding
 * inside the reference's own functions would call.  Each entry point cites the reference
 * code it replaces (paths relative to the reference repository root):
 *
 *   cf_load_reads      hand-over of what scripts/ncrf_parser.py:61-118 / :28-59 produce
 *   cf_count_kmers     scripts/distance_based_kmer_recruitment.py:39-63  (A1)
 *   cf_select_rare     scripts/distance_based_kmer_recruitment.py:66-82  (A2)
 *   cf_set_kmers       scripts/read_placer.py:20-27 (genomic k-mer set given from a file)
 *   cf_build_clouds    scripts/read_kmer_cloud.py:17-40                  (A3)
 *   cf_filter_clouds   scripts/read_kmer_cloud.py:43-54                  (A4)
 *   cf_dist_edges      scripts/distance_based_kmer_recruitment.py:85-128 (A5) fused with
 *                      :131-149 (A6)
 *   cf_place_reads     scripts/cloud_contig.py:26-41, :87-95 (A8) and
 *                      scripts/read_placer.py:35-94 (A9)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 or a negative
 * errno-style code and never throws or aborts; cf_last_error() gives the message; the
 * caller allocates every output buffer; the opaque context owns all device memory; one
 * context per process and device, calls serialised by the caller; one HIP stream inside.
 * Host pointers are borrowed for the duration of the call only.
 *
 * k-mers are 2-bit packed, A=0 C=1 G=2 T=3, first base in the most significant position, so
 * unsigned integer order equals the string order the reference sorts by.  Forward strand
 * only — the reference never canonicalises.  1 <= k <= 31.
 */
#ifndef CFHIP_H
#define CFHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cf_ctx cf_ctx;

/* Work counters; the counters of the reference's algorithm (n_bases ... n_unique) are identical for the oracle and
 * the HIP path on one input; table_capacity, n_spilled, hbm_bytes_live, n_dist_passes and n_edges_stored describe the device run
 * (n_dist_passes and n_spilled depend on how workgroups interleave and may differ by a few between two runs). */
typedef struct cf_stats {
    int64_t n_reads, n_bases, n_units;
    int64_t n_windows;      /* N_w  = sum max(0, len - k + 1)                       */
    int64_t n_read_kmers;   /* N_rk = sum over reads of #distinct k-mers             */
    int64_t n_distinct;     /* K_dist = distinct k-mers seen                         */
    int64_t n_kept;         /* k-mers surviving the multi-occurrence cut             */
    int64_t n_kmers;        /* size of the current k-mer set (rare / genomic)        */
    int64_t n_cloud_entries;/* N_ce                                                  */
    int64_t n_emissions;    /* E = pair emissions of the last cf_dist_edges          */
    int64_t n_edges;        /* selected edges of the last cf_dist_edges              */
    int64_t n_unique;       /* selected ("unique") k-mers so far                     */
    int64_t table_capacity; /* slots of the HBM k-mer table                          */
    int64_t n_spilled;      /* first k-mers whose (b,d) table had to be partitioned  */
    int64_t hbm_bytes_live; /* device memory currently owned by the context          */
    int64_t n_dist_passes;  /* (first k-mer, partition) table passes of the last cf_dist_edges */
    int64_t n_edges_stored; /* edge rows of the last cf_dist_edges held on the device: n_edges when edge_cap allowed it, else at most edge_cap */
} cf_stats;

/* Device time (HIP events on the context's stream) of the last call of each stage. */
typedef struct cf_times {
    float load_ms, count_ms, select_ms, clouds_ms, filter_ms, postings_ms, dist_ms, place_ms;
    float dist_kernel_ms;   /* the cf_dist_edges main kernel alone                          */
    float count_kernel_ms;  /* the cf_count_kmers main kernel alone                         */
    float rr_kernel_ms;     /* the cf_rr_distances kernel alone                             */
} cf_times;

int  cf_create(int device, cf_ctx** out);
void cf_destroy(cf_ctx* ctx);
const char* cf_last_error(const cf_ctx* ctx);
int  cf_device_info(cf_ctx* ctx, char* name, int name_len, int64_t* hbm_bytes, int32_t* n_cu);

/* Reads: ASCII bases, read_off[R+1]; units: unit_ptr[R+1] indexes global units, unit_start/unit_end are absolute
 * offsets into bases.  Symbols other than upper-case A, C, G, T are allowed: cf_count_kmers skips the windows that hold
 * one (the reference counts those as k-mers of their own, distance_based_kmer_recruitment.py:47-53 — the host side-path
 * cfh_exotic_summary of cfhost.h keeps that count and tells whether any of them could matter downstream), and
 * cf_build_clouds upper-cases a, c, g, t first, as read_kmer_cloud.py:25 does. */
int cf_load_reads(cf_ctx* ctx, const uint8_t* bases, const int64_t* read_off, int64_t n_reads,
                  const int64_t* unit_ptr, const int64_t* unit_start, const int64_t* unit_end);
/* Replace the unit table only (n_motif change). */
int cf_load_units(cf_ctx* ctx, const int64_t* unit_ptr, const int64_t* unit_start, const int64_t* unit_end);

/* A1: presence / multi-occurrence table over the reads [read_lo, read_hi) (the whole set when
 * read_lo = 0, read_hi >= R). */
int cf_count_kmers(cf_ctx* ctx, int32_t k, int64_t read_lo, int64_t read_hi);
/* SURVEY §8(f) rank 2 (scripts/better_consensus_unit_reconstruction.py:127-135, :156-167): table of total
 * OCCURRENCE counts (every window of every read counts; val is one 64-bit count, cf_get_table returns its low /
 * high halves in pres / multi), and the n k-mers with the largest (count, k-mer), sorted descending
 * (size-query with keys_out == NULL: n_out = min(n, distinct)). */
int cf_count_occurrences(cf_ctx* ctx, int32_t k, int64_t read_lo, int64_t read_hi);
int cf_top_kmers(cf_ctx* ctx, int64_t n, uint64_t* keys_out, uint64_t* counts_out, int64_t* n_out);
/* Start an empty table for keys of length k sized for about expected_keys distinct k-mers (owner side of
 * the multi-GPU exchange, followed by cf_merge_table). */
int cf_reset_table(cf_ctx* ctx, int32_t k, int64_t expected_keys);
/* Dump the occupied slots, unordered (tests, multi-GPU merge): size-query with keys == NULL. */
int cf_get_table(cf_ctx* ctx, uint64_t* keys, uint32_t* pres, uint32_t* multi, int64_t cap, int64_t* n_out);
/* Add (key, pres, multi) triples into the table (owner-side merge of the multi-GPU exchange). */
int cf_merge_table(cf_ctx* ctx, const uint64_t* keys, const uint32_t* pres, const uint32_t* multi, int64_t n);

/* A2: k-mer set := { x : multi[x] <= max_nonuniq, lo <= pres[x] <= hi }, sorted ascending. */
int cf_select_rare(cf_ctx* ctx, int32_t max_nonuniq, uint32_t lo, uint32_t hi, int64_t* n_out);
/* Install a k-mer set (sorted ascending, unique). */
int cf_set_kmers(cf_ctx* ctx, const uint64_t* kmers, int64_t n, int32_t k);
int cf_get_kmers(cf_ctx* ctx, uint64_t* out, int64_t cap);

/* A3: per-unit clouds of the current k-mer set as CSR (entries = indices into the set, sorted
 * unique inside each unit). */
int cf_build_clouds(cf_ctx* ctx, int64_t* n_entries);
/* A4: keep k-mers present in [min_mult, max_mult] clouds overall (max_mult = 0: no upper bound). */
int cf_filter_clouds(cf_ctx* ctx, uint32_t min_mult, uint32_t max_mult, int64_t* n_entries);
int cf_get_clouds(cf_ctx* ctx, int64_t* cloud_ptr /* U+1 */, int32_t* entries, int64_t cap);
/* Install clouds computed elsewhere (multi-GPU all-gather of per-shard clouds). */
int cf_set_clouds(cf_ctx* ctx, const int64_t* cloud_ptr, const int32_t* entries, int64_t n_entries);

/* A5+A6: for first k-mers a with a % n_parts == part: histogram over (b, d) of the reads
 * [min_n, max_n), then keep (d, a, b, cnt) with cnt >= min_cov and
 * (double)cnt / (double)sum_d cnt >= rel_threshold.  Up to edge_cap edges are stored on the
 * device (all are counted); the unique-k-mer bitmap accumulates across calls until
 * cf_reset_unique(). */
int cf_dist_edges(cf_ctx* ctx, int64_t min_n, int64_t max_n, int32_t min_d, int32_t max_d,
                  uint32_t min_cov, double rel_threshold, int32_t part, int32_t n_parts,
                  int64_t edge_cap, int64_t* n_edges);
int cf_get_edges(cf_ctx* ctx, uint32_t* out /* n x 4: d, a, b, cnt */, int64_t cap);   /* the first min(cap, stored) edges */
/* Sort the stored edges by (d, a, b) on the device (they are written in the order workgroups finish; the reference's file
 * order, distance_based_kmer_recruitment.py:165-171, is the insertion order of its dicts and is not reproduced).  Only
 * meaningful when every selected edge was stored (edge_cap >= n_edges of the last cf_dist_edges call). */
int cf_sort_edges(cf_ctx* ctx);
/* Order-independent checksum of the first min(n, stored) edges, computed on the device: the sum over rows (d, a, b, cnt) of
 * mix(mix(mix(mix(d + 0x9E37) ^ a) ^ (b << 1)) ^ (cnt << 2)) mod 2^64 with the 64-bit finaliser of MurmurHash3 as mix (the
 * same figure the oracle reports; full-size parity checks compare every selected edge of
 * distance_based_kmer_recruitment.py:131-149 without copying tens of GB to the host). */
int cf_edges_checksum(cf_ctx* ctx, int64_t n, uint64_t* out);
/* Order-independent checksums of the other resident results, computed on the device (sums mod 2^64 of the oracle's per-element
 * mixes, oracle/c/cf_oracle_mt.c; mix = the 64-bit finaliser of MurmurHash3): what =
 *   CF_CHECKSUM_TABLE   every (key, pres, multi) of the A1 table (distance_based_kmer_recruitment.py:39-63):
 *                       mix(mix(mix(key + 0x7AB1E) ^ pres) ^ (multi << 1));
 *   CF_CHECKSUM_KMERS   the installed k-mer set, after cf_select_rare the rare set (:66-82): mix(kmer ^ 0xABCDEF);
 *   CF_CHECKSUM_CLOUDS  every (unit, entry) of the cloud CSR (read_kmer_cloud.py:17-40): mix(mix(unit + 0x51ED) ^ entry);
 *   CF_CHECKSUM_UNIQUE  the k-mers of the set whose unique bit is set (:145-148), same mix as CF_CHECKSUM_KMERS.
 * n_items (may be NULL): how many elements went into the sum.  Full-size parity checks (BASELINE configs[3]: 1.3e9 table
 * entries, 6.5e8 cloud entries) compare these figures with a committed oracle record instead of copying the arrays back. */
enum { CF_CHECKSUM_TABLE = 0, CF_CHECKSUM_KMERS = 1, CF_CHECKSUM_CLOUDS = 2, CF_CHECKSUM_UNIQUE = 3 };
int cf_checksum(cf_ctx* ctx, int32_t what, uint64_t* sum, int64_t* n_items);
int cf_get_unique_mask(cf_ctx* ctx, uint8_t* mask /* n_kmers bytes of 0/1 */);
int cf_or_unique_mask(cf_ctx* ctx, const uint8_t* mask);
int cf_reset_unique(cf_ctx* ctx);

/* A8+A9: greedy placement.  cls[r]: 0 prefix, 1 internal, 2 suffix; id_rank[r] = rank of the
 * read id in ascending string order (tie-break).  Outputs, in the order the reference writes
 * the file: out_read[i], out_pos[i] (-1 = None), out_s0[i], out_s1[i] for i < R. */
int cf_place_reads(cf_ctx* ctx, const uint8_t* cls, const int32_t* id_rank, int32_t min_cloud_kmer_freq,
                   int32_t min_unit, int32_t min_inters, int32_t min_prop,
                   int64_t* out_read, int64_t* out_pos, int32_t* out_s0, int32_t* out_s1);

int cf_get_stats(cf_ctx* ctx, cf_stats* out);
int cf_get_times(cf_ctx* ctx, cf_times* out);

/* Read recruitment, the stage before the path (reference scripts/read_recruitment/rr.cpp:73-90:
 * edlibAlign(unit, read) and edlibAlign(revcomp(unit), read), mode HW, k = threshold; a read is kept when either result
 * is not -1).  Needs only a context.  unit: 1 .. 4096 upper-case ACGT; reads: any bytes back to back (matching is
 * literal, as in edlib), read_off[n_reads + 1].  dist_fwd / dist_rc [n_reads]: minimum edit distance between the unit /
 * its reverse complement and a substring of the read, -1 when above threshold (threshold < 0: no limit). */
int cf_rr_distances(cf_ctx* ctx, const uint8_t* unit, int32_t unit_len, const uint8_t* reads, const int64_t* read_off,
                    int64_t n_reads, int32_t threshold, int32_t* dist_fwd, int32_t* dist_rc);

/* Multi-GPU (new design — the reference has no distributed code, scripts/ is single-process): one
 * process per GPU, reads sharded across ranks, RCCL over xGMI inside the library on the context's stream.
 *   cf_comm_init         joins the communicator: rank 0 publishes the RCCL unique id at `rendezvous` (a path all ranks
 *                        share), the others read it.  world = 1 is allowed (every collective degenerates).
 *   cf_exchange_table    after cf_count_kmers on the local shard: (key, pres, multi) records are bucketed by
 *                        owner = hash(key) % world on the device and exchanged with one all-to-all (ncclSend/ncclRecv
 *                        pairs in rounds of <= 256 MB); the table then holds the exact global counts of the OWNED keys
 *                        (what distance_based_kmer_recruitment.py:39-63 computes over all reads), so cf_select_rare
 *                        selects the owned rare k-mers.  bytes_sent: bytes this rank sent to its peers.
 *   cf_allgather_kmers   all-gather of the owned rare lists; every rank installs the sorted union (= :66-82's set).
 *   cf_allgather_clouds  after cf_build_clouds on the local shard: all-gather of every rank's per-unit clouds
 *                        (read_kmer_cloud.py:34-40 over all reads, units in rank order); cf_dist_edges then works on
 *                        them, each rank on the first k-mers a % n_parts == part, with no reduction.
 *   cf_allreduce_unique  OR of the selected-k-mer masks of all ranks (filter_dist_tuples' set, :145-148).
 *   cf_comm_allreduce_i64  host values summed (op 0) or maximised (op 1) over ranks: counters, timing, barrier. */
int cf_comm_init(cf_ctx* ctx, int32_t rank, int32_t world, const char* rendezvous);
int cf_comm_free(cf_ctx* ctx);
int cf_comm_info(cf_ctx* ctx, int32_t* rank, int32_t* world);
int cf_comm_allreduce_i64(cf_ctx* ctx, int64_t* vals, int64_t n, int32_t op);
int cf_exchange_table(cf_ctx* ctx, int64_t* bytes_sent);
int cf_allgather_kmers(cf_ctx* ctx, int64_t* n_out);
int cf_allgather_clouds(cf_ctx* ctx, int64_t* n_entries);
int cf_allreduce_unique(cf_ctx* ctx, int64_t* n_unique);

/* Tuning knobs (defaults are chosen for gfx950): name in {"dist_block" (threads per workgroup, 0 = auto), "dist_wgs"
 * (workgroups per CU the LDS is split between, 0 = auto: by the pair emissions per first k-mer), "dist_slots" (LDS budget of the (b,d) table in 8-byte units, 0 = all that
 * is left), "dist_sketch" (0: every pair goes to the exact table), "dist_fill_pct", "dist_est_pct", "dist_stage", "dist_edge_chunk" (edge rows a workgroup reserves in the output per global atomic, 0 = 8192; tests use small chunks), "dist_int_thr" (0: the dominance test always divides in doubles; 1, the default: the literal 0.8 is tested as 5 cnt >= 4 total, which is the same predicate),
 * "dist_wide", "dist_post_atomics" (1: postings by a histogram and a fill pass of atomics instead of the sort), "dist_hot_cap" (tests: a small cap on the filter's hot-slot list forces the evaluation inside the bucket scan), "dist_sketch_bits" (bits of a counter of the distance stage's counting sketch: 0, the default: 4 when min_cov <= 9 — twice the counters in the same LDS —, else 8; 8 forces bytes), "lut_shift" (the k-mer lookup table of cf_build_clouds gets (2 x k-mers rounded up to a power of two) << lut_shift slots; -1, the default: 2 for sets of up to 1.7e7 k-mers, 1 up to 1.3e8, else 0), "dist_hot_entries" (default 32768: first k-mers with more partner entries than this keep no list of hot slots during their inserts — it would overflow — and their filter scans the count fields; -1: always keep it), "dist_regions" (1, 2, 4, 8: force the region layout of the 6-byte slots, which k-mer sets of 2^24 .. 2^27 ranks with long reads take by themselves), "dist_dbits" (5 .. 8: cap on the distance-field bits of the 6-byte table slots [d | b]; 0 = 32 minus the bits the k-mer ranks need), "place_mode" (2, the default: per-read score regions and one kernel per greedy iteration, cf_place2.hip — for min_inters >= 4; smaller thresholds make nearly every score row a candidate row and take path 1; 3: the regions whatever the threshold; 1: the hash-map path of rounds 1-3, cf_place.hip), "place_grid" / "place_block" (workgroups and threads per workgroup of the iteration kernel, 0 = 128 x 1024), "place_row_words" (32 or 64 words per posting row, 0 = by the longest posting list), "place_slots_per_unit" (score-region slots per unit of a read, 0 = 48; grown automatically when a region fills), "place_l3" / "place_l3_shift" (1: the third level of the placement arg-max, groups of 2^shift blocks of 64 reads kept lazily — built in round 5, measured neutral at 500 000 reads, off by default), "dist_region_bytes" (1: the region layout streams rank and unit index apart, as k-mer sets beyond 2^26 ranks or reads beyond 128 units do by themselves), "place_long_rescans" (default 2: a run of the region path whose reads average more than this many rescans of reads with more than four candidate score rows per greedy iteration — k-mers that are not unique to one place of the array, thin coverage — is handed to the hash-map path; -1: at the first look, tests), "place_cmap_bits" (log2 of the first capacity of the contig's overflow map — the positions of a k-mer beyond its fourth —, 0 = cloud entries / 8, at least 2^21; grown automatically, times four, when it passes half load), "place_chunk", "place_fused" (place_mode 1: cloud entries per wave step; 1: score updates applied by the waves that lay a read onto the contig, 0: through an event list and a third kernel per greedy iteration), "count_mode" (1: A1 by sort and reduce, 0: the atomic table), "count_bits" (bucket bits of the former, 0 = auto), "count_slots", "count_tile", "comm_round_bytes" (bytes per pair of ranks and round of the multi-GPU exchanges, default 2^28; tests force many rounds), "comm_self_p2p" (1: the message a rank sends to itself goes through ncclSend / ncclRecv like every other one, so that a one-GPU box runs the whole p2p path)}.  Results never depend on them (tests/test_gpu_parity.py). */
int cf_set_param(cf_ctx* ctx, const char* name, int64_t value);

/* Self-tests of the device primitives against host results (used by tests/ only). */
int cf_selftest_sort(cf_ctx* ctx, const uint64_t* keys, int64_t n, int32_t bits, uint64_t* out);
int cf_selftest_scan(cf_ctx* ctx, const int64_t* in, int64_t n, int64_t* out);
/* the arg-max of the greedy placement (read_placer.py:63-78: larger (s0, s1), then the larger offset, then the smaller id rank)
 * over n candidates given as rows (s0, s1, offset, rank, valid); out6 = (s0, s1, offset, rank, index of the winner, valid) */
int cf_selftest_argmax(cf_ctx* ctx, const uint32_t* cands, int64_t n, uint32_t* out6);

#ifdef __cplusplus
}
#endif
#endif
