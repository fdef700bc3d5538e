/*
 * cfhost.h — host-side C ABI of centroflye_amd (libcfhost.so, plain C++17, no GPU).
 *
 * Two jobs, both on the INPUT side of the hot path:
 *
 *  1. NCRF report ingestion: parse the text report, keep the longest >= min_record_len
 *     alignment per read, orient '-' records, split alignments into HOR units, classify
 *     reads, and pack everything into flat arrays the device library (cfhip.h) consumes.
 *     Replaces scripts/ncrf_parser.py:61-118 (NCRF_Report.__init__), :28-59
 *     (NCRF_Record.get_motif_alignments) and :120-145 (classify) of the reference.
 *
 *  2. A deterministic synthetic generator of HOR arrays + ONT-like reads that writes
 *     NCRF-format reports (or packs directly).  The reference has no read simulator
 *     (scripts/simulate_tandem_repeat.py:15-55 only makes the genome); NCRF itself is an
 *     external binary that is not available, so reports are synthesised from the known
 *     true alignment.
 *
 * Conventions: every function returns 0 on success or a negative code and writes a
 * message into the caller's err buffer; no function throws or aborts; returned pointers
 * are borrowed from the pack and live until cfh_pack_free().
 */
#ifndef CFHOST_H
#define CFHOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cfh_pack cfh_pack;

/* Parameters of the synthetic generator. */
typedef struct cfh_synth_params {
    uint64_t seed;
    int32_t unit_len;        /* HOR unit length (2055 = len of supplementary_data/DXZ1_rc.fasta) */
    int32_t monomer_len;     /* 171 */
    double  monomer_div;     /* divergence of each monomer from the ancestor (0.25) */
    int64_t n_units;         /* M copies in the array */
    int64_t flank;           /* random flank length on each side (200000) */
    double  unit_div;        /* per-copy point-substitution rate (0.01) */
    int64_t n_reads;         /* number of emitted reads (reads with < min_aligned are rejected) */
    double  mean_len;        /* log-normal mean (20000) */
    double  sigma;           /* log-normal sigma (0.5) */
    int64_t min_len, max_len;/* clamp (6000, 200000) */
    double  p_del, p_sub, p_ins; /* per-base read errors (0.02, 0.02, 0.015) */
    int64_t min_aligned;     /* 5000 */
    int32_t n_prefix, n_suffix; /* forced prefix / suffix reads (8, 8) */
    int64_t prefix_threshold;/* 50000 */
    double  p_split;         /* probability that a read's alignment is reported as two records */
    int32_t n_threads;       /* worker threads (0 = hardware concurrency) */
    int32_t var_len;         /* bases replaced per copy-specific variant (1 = point substitution) */
    int32_t cand_offset, cand_stride; /* simulate read candidates offset, offset+stride, ... (rank sharding) */
} cfh_synth_params;

void cfh_synth_defaults(cfh_synth_params* p);

/* Generate.  report_path may be NULL (no text written).  If out != NULL a pack is built
 * through exactly the same ingestion code the parser uses.  keep_rows: keep the oriented
 * alignment rows (needed for n_motif != 1 and for record-level API parity). */
int cfh_synth(const cfh_synth_params* p, const char* report_path, int keep_rows,
              cfh_pack** out, char* err, int errlen);

/* Parse an NCRF report (format: SURVEY.md Appendix B; reference ncrf_parser.py:66-77). */
int cfh_parse_report(const char* path, int64_t min_record_len, int keep_rows, int n_threads,
                     cfh_pack** out, char* err, int errlen);

void cfh_pack_free(cfh_pack* p);

/* Binary cache of a pack (both stage scripts read the same report).  source_id[4] identifies what
 * the pack was made from, as the caller sees it — by convention {size of the report, its mtime in ns, min_record_len,
 * keep_rows}; cfh_pack_load refuses (-61) a file whose four numbers differ, that is truncated or damaged. */
int cfh_pack_save(const cfh_pack* p, const char* path, const int64_t* source_id, char* err, int errlen);
int cfh_pack_load(const char* path, const int64_t* source_id, cfh_pack** out, char* err, int errlen);

/* Shape. */
int64_t cfh_n_reads(const cfh_pack* p);      /* kept records, in first-insertion order */
int64_t cfh_n_bases(const cfh_pack* p);      /* N_b = sum of de-gapped aligned lengths */
int64_t cfh_n_seen(const cfh_pack* p);       /* distinct read ids seen (kept + discarded) */
int32_t cfh_non_acgt(const cfh_pack* p);     /* 1 if any kept base is outside {A,C,G,T} */
/* The k-mer windows of reads [read_lo, read_hi) that hold a symbol other than upper-case A, C, G, T — the ones the device
 * path has no 2-bit code for and skips — counted as the reference counts every window (as strings of the raw row,
 * scripts/distance_based_kmer_recruitment.py:39-63): out[0] distinct such k-mers, out[1] sum over reads of distinct ones,
 * out[2] of them with multi <= max_nonuniq, out[3] of those with lo <= pres <= hi (they are "rare" k-mers of the
 * reference), out[4] of those holding no lower-case letter: only these could match a window of an upper-cased unit
 * (read_kmer_cloud.py:25) and reach the outputs; the others cannot.  The caller refuses the input when out[4] > 0. */
int cfh_exotic_summary(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, int64_t read_lo, int64_t read_hi, int64_t out[5]);
/* The same windows one by one — rows of 5 int64 {h1, h2: two independent 63-bit hashes of the window's text, pres, multi, 1 if it
 * holds no lower-case letter} — for a caller that must add the counts of several read shards before it can tell which
 * windows are rare (centroflye_amd/sharded.py).  Returns the number of distinct windows (or < 0); fills at most cap rows. */
int64_t cfh_exotic_list(const cfh_pack* p, int32_t k, int64_t read_lo, int64_t read_hi, int64_t* rows, int64_t cap);
/* The windows out[4] of cfh_exotic_summary counts, as text: rare and free of lower-case letters — the k-mers with an N (or another
 * upper-case symbol) that the reference selects, finds again in the upper-cased units (read_kmer_cloud.py:25) and writes
 * (distance_based_kmer_recruitment.py:47-53, :160-164).  k bytes each, ascending, at most cap written; returns their number (or < 0).
 * The caller carries them beside the 2-bit set (centroflye_amd/kmers.py: KmerSet.extra). */
int64_t cfh_exotic_rare(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, char* out, int64_t cap);
/* The rare windows that hold a lower-case letter (the rest of cfh_exotic_summary's out[3]): members of the set the reference's
 * get_rare_kmers returns (distance_based_kmer_recruitment.py:66-82), never of a cloud (read_kmer_cloud.py:25 upper-cases the unit). */
int64_t cfh_exotic_rare_lower(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, char* out, int64_t cap);
/* Every such window that the reference's table keeps (twice in at most max_nonuniq reads, :56-62), lower case included: text (k bytes
 * each, ascending) and the number of reads holding it — the keys of get_kmer_freqs_from_ncrf_report's mapping that have no 2-bit code. */
int64_t cfh_exotic_kept(const cfh_pack* p, int32_t k, int32_t max_nonuniq, char* out, int64_t* pres, int64_t cap);

/* Flat arrays. */
const uint8_t* cfh_bases(const cfh_pack* p);     /* ASCII, de-gapped oriented r_al, length N_b */
const int64_t* cfh_read_off(const cfh_pack* p);  /* R+1 offsets into bases */
const char*    cfh_ids(const cfh_pack* p);       /* concatenated read ids */
const int64_t* cfh_id_off(const cfh_pack* p);    /* R+1 offsets into ids */
/* per record: r_len, r_al_len, r_st, r_en (flipped for '-'), strand (0 '+', 1 '-'), n_alignments,
 * alignment-row length (columns), motif id */
const int64_t* cfh_meta(const cfh_pack* p);      /* R x 8 */
int32_t        cfh_n_motifs(const cfh_pack* p);
const char*    cfh_motif(const cfh_pack* p, int32_t motif_id, int64_t* len);
/* discarded read ids (seen but never kept), '\n'-joined */
const char*    cfh_discarded(const cfh_pack* p, int64_t* len);

/* Unit split for n_motif = n (reference ncrf_parser.py:28-59).  n == 1 is always available;
 * other n need keep_rows.  unit_ptr: R+1; unit_start/unit_end: absolute offsets into bases
 * (de-gapped); unit_col: per unit the [start,end) alignment columns (2 values per unit). */
int cfh_units(cfh_pack* p, int32_t n, int64_t* n_units,
              const int64_t** unit_ptr, const int64_t** unit_start, const int64_t** unit_end,
              const int64_t** unit_col, char* err, int errlen);

/* Read classes (reference ncrf_parser.py:120-145): 0 prefix, 1 internal, 2 suffix. */
int cfh_classify(const cfh_pack* p, int64_t large_threshold, int64_t small_threshold,
                 uint8_t* cls_out);

/* Oriented alignment rows of record r (which: 0 = r_al, 1 = m_al); NULL if rows not kept. */
const char* cfh_row(const cfh_pack* p, int64_t r, int32_t which, int64_t* len);

/* Text writers for the on-disk outputs of the path (reference
 * distance_based_kmer_recruitment.py:158-171): k-mers as strings, one per line, in the
 * given order; edges as "d kmer_a kmer_b cnt". kmers are 2-bit packed (A<C<G<T), first
 * base in the most significant position. */
int cfh_write_kmers(const char* path, const uint64_t* kmers, int64_t n, int32_t k,
                    char* err, int errlen);
int cfh_write_edges(const char* path, int append, const uint64_t* rare_kmers, int32_t k,
                    const uint32_t* edges /* n x 4: d,a,b,cnt */, int64_t n,
                    char* err, int errlen);
/* Per-position read-unit export, the consumer of read_positions.csv (reference scripts/eltr_polisher.py:53-66
 * ELTR_Polisher.map_pos2read, :68-97 export_read_units, :45-51 default max_pos).  rec[i] / pos[i]: record index and
 * position of the i-th placed read in read_positions.csv order ("None" lines left out by the caller).  max_pos < 0
 * means infinity (the end of the right-most placed read).  Writes outdir/pos_P/read_units.fasta (">gen_pos=P|r_id=ID|
 * r_pos=i" + the upper-case de-gapped unit, in placement order) and outdir/pos_P/median_read_unit.fasta (the first
 * header in sorted order whose unit has the high-median length).  SURVEY.md §8(f) rank 3. */
int cfh_export_read_units(cfh_pack* p, const int64_t* rec, const int64_t* pos, int64_t n_placed, int64_t min_pos,
                          int64_t max_pos, const char* outdir, int n_threads, int64_t* n_positions,
                          int64_t* n_units_written, char* err, int errlen);

/* Read a k-mer text file (one per line) into 2-bit codes; returns count via n_out, fills out
 * if non-NULL (size-query then fill). All k-mers must have length k and be ACGT. */
int cfh_read_kmers(const char* path, int32_t k, uint64_t* out, int64_t cap, int64_t* n_out,
                   char* err, int errlen);

#ifdef __cplusplus
}
#endif
#endif
