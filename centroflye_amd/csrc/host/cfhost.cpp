// cfhost.cpp — host-side ingestion of NCRF reports and the synthetic read generator.
// C ABI in include/cfhost.h.  Plain C++17; no GPU code here.
//
// Reference behaviour restated (never copied) from:
//   scripts/ncrf_parser.py:61-118  record selection / orientation
//   scripts/ncrf_parser.py:28-59   unit split (regex "base([-]*)" x len(motif) x n)
//   scripts/ncrf_parser.py:120-145 classify
//   scripts/utils/bio.py:27-29     RC
#include "cfhost.h"

#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/types.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// os.makedirs that tolerates an existing directory (reference utils/os_utils.py:29-34)
static int mkdir_p(const char* path) {
    std::string cur;
    const std::string full(path);
    for (size_t i = 0; i <= full.size(); ++i) {
        if (i == full.size() || full[i] == '/') {
            if (!cur.empty() && ::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return -1;
        }
        if (i < full.size()) cur += full[i];
    }
    return 0;
}

static bool write_file(const std::string& path, const std::string& data) {
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) return false;
    const bool ok = std::fwrite(data.data(), 1, data.size(), f) == data.size();
    return std::fclose(f) == 0 && ok;
}

void set_err(char* err, int errlen, const std::string& msg) {
    if (err && errlen > 0) {
        std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
    }
}

// ---------------------------------------------------------------- small helpers
inline char rc_char(char c) {
    // translate 'ATGCatgc-' -> 'TACGtacg-'; everything else passes through
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G';
        case 'a': return 't'; case 't': return 'a'; case 'g': return 'c'; case 'c': return 'g';
        default: return c;
    }
}
struct ByteTables {      // per-byte lookups for the row passes
    unsigned char rc[256], up[256], other[256];      // other: 1 for anything that is not A/C/G/T
    ByteTables() {
        for (int i = 0; i < 256; ++i) {
            rc[i] = (unsigned char)rc_char((char)i);
            up[i] = (unsigned char)((i >= 'a' && i <= 'z') ? i - 32 : i);
            other[i] = !(i == 'A' || i == 'C' || i == 'G' || i == 'T');
        }
    }
};
const ByteTables BT;
void rc_inplace(std::string& s) {
    const size_t n = s.size();
    unsigned char* d = (unsigned char*)&s[0];
    for (size_t i = 0, j = n; i < j; ) {
        --j;
        const unsigned char a = BT.rc[d[i]], b = BT.rc[d[j]];
        d[i] = b; d[j] = a;
        ++i;
    }
}
inline char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

// One alignment as it stands in the file (read orientation for '-').
struct FileRecord {
    std::string r_id;
    int64_t r_len = 0, r_al_len = 0, r_st = 0, r_en = 0;
    std::string r_al, m_al, motif;
    char strand = '+';
    int64_t m_al_len = 0, score = 0;
};

// CFH_TIMING=1 in the environment prints the phases of the parser to stderr.
struct PhaseClock {
    bool on = std::getenv("CFH_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[cfhost] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// Result of the heavy per-record work (orientation + de-gap + unit split for n = 1).
struct Staged {
    std::vector<int64_t> col_bounds; // unit boundaries as alignment columns (n = 1)
    std::vector<int64_t> pos_bounds; // same boundaries in de-gapped read coordinates
    int64_t ncols = 0;
    bool non_acgt = false;
};

// Unit split: leftmost non-overlapping matches of motif*n in the de-gapped, upper-cased
// motif row; a match ends after the gap columns that follow its last base.
void split_units(size_t r_cols, const char* m_al, size_t m_cols, const std::string& motif,
                 int n, std::vector<int64_t>& col_bounds) {
    col_bounds.clear();
    const int64_t ncols = (int64_t)m_cols;
    static thread_local std::vector<unsigned char> s;     // the motif row without gaps, upper case
    static thread_local std::vector<uint32_t> cols;       // and the column of each of its letters
    s.resize((size_t)ncols + 1);
    cols.resize((size_t)ncols + 1);
    size_t ns = 0;
    const unsigned char* m = (const unsigned char*)m_al;
    for (int64_t c = 0; c < ncols; ++c) {
        s[ns] = BT.up[m[c]]; cols[ns] = (uint32_t)c;
        ns += m[c] != '-';
    }
    std::string pat;
    for (int i = 0; i < n; ++i) pat += motif;
    if (pat.empty()) return;
    const size_t L = pat.size();
    std::vector<int64_t> starts;
    int64_t last_end = -1;
    size_t pos = 0;
    while (pos + L <= ns) {
        size_t p = pos;
        if (std::memcmp(s.data() + pos, pat.data(), L) != 0) {      // copies usually follow one another directly
            const void* f = memmem(s.data() + pos, ns - pos, pat.data(), L);
            if (!f) break;
            p = (size_t)((const unsigned char*)f - s.data());
        }
        int64_t st = cols[p];
        int64_t en = (int64_t)cols[p + L - 1] + 1;
        while (en < ncols && m_al[(size_t)en] == '-') ++en;
        starts.push_back(st);
        last_end = en;
        pos = p + L;
    }
    if (starts.empty()) return;
    const double cut = (double)motif.size() * 0.2;
    if ((double)starts[0] > cut) col_bounds.push_back(0);
    for (auto v : starts) col_bounds.push_back(v);
    col_bounds.push_back(last_end);
    if ((double)last_end < (double)r_cols - cut) col_bounds.push_back((int64_t)r_cols);
}
void split_units(const std::string& r_al, const std::string& m_al, const std::string& motif,
                 int n, std::vector<int64_t>& col_bounds) {
    split_units(r_al.size(), m_al.data(), m_al.size(), motif, n, col_bounds);
}

void cols_to_pos(const char* r, size_t r_cols, const std::vector<int64_t>& col_bounds,
                 std::vector<int64_t>& pos_bounds) {
    pos_bounds.resize(col_bounds.size());
    int64_t c = 0, cnt = 0;
    for (size_t bi = 0; bi < col_bounds.size(); ++bi) {      // boundaries ascend
        const int64_t to = std::min<int64_t>(col_bounds[bi], (int64_t)r_cols);
        int64_t add = 0;
        for (; c < to; ++c) add += r[c] != '-';
        cnt += add;
        pos_bounds[bi] = cnt;
    }
}
void cols_to_pos(const std::string& r_al, const std::vector<int64_t>& col_bounds,
                 std::vector<int64_t>& pos_bounds) {
    cols_to_pos(r_al.data(), r_al.size(), col_bounds, pos_bounds);
}

inline int64_t count_bases(const char* r, size_t n) {      // letters of a row that are not gap columns
    int64_t g = 0;
    for (size_t c = 0; c < n; ++c) g += r[c] != '-';
    return g;
}

// Reverse complement of [b, b + n) into out (resized).
void rc_copy(const char* b, size_t n, std::string& out) {
    out.resize(n);
    unsigned char* d = (unsigned char*)&out[0];
    const unsigned char* sb = (const unsigned char*)b;
    for (size_t i = 0; i < n; ++i) d[i] = BT.rc[sb[n - 1 - i]];
}

struct AlnKey {  // (r_st, r_en, strand) tuple ordering as Python sorts it
    int64_t st, en; char strand;
    bool operator<(const AlnKey& o) const {
        if (st != o.st) return st < o.st;
        if (en != o.en) return en < o.en;
        return strand < o.strand;
    }
};

struct SeenRead {
    int64_t read_len = 0;   // read_lens[r_id]: value of the LAST alignment seen
    AlnKey first{}, last{}; // min / max of positions_all_alignments[r_id]
    int64_t n_aln = 0;
    int64_t rec = -1;       // index into records (kept), -1 if none
};

}  // namespace

// The flat base array: anonymous pages that nobody has written yet (each staging thread is the first to touch the part it
// fills, so the page faults of a gigabyte are spread over the threads instead of being taken by one zero-fill).
struct RawBytes {
    char* p = nullptr;
    size_t n = 0;
    RawBytes() = default;
    RawBytes(const RawBytes&) = delete;
    RawBytes& operator=(const RawBytes&) = delete;
    ~RawBytes() { release(); }
    void release() { if (p) munmap(p, std::max<size_t>(n, 1)); p = nullptr; n = 0; }
    void resize(size_t bytes) {
        release();
        void* m = mmap(nullptr, std::max<size_t>(bytes, 1), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(m, std::max<size_t>(bytes, 1), MADV_HUGEPAGE);
        p = (char*)m; n = bytes;
    }
    const char* data() const { return p; }
    char* data() { return p; }
    size_t size() const { return n; }
    char operator[](size_t i) const { return p[i]; }
};

struct cfh_pack {
    // kept records
    std::vector<int64_t> meta;      // R x 8
    std::string ids;
    std::vector<int64_t> id_off{0};
    RawBytes bases;
    std::vector<int64_t> read_off{0};
    std::vector<std::string> rows_r, rows_m;
    bool keep_rows = false;
    bool non_acgt = false;
    std::vector<std::string> motifs;
    std::unordered_map<std::string, int32_t> motif_ids;
    // n = 1 units
    std::vector<int64_t> u1_ptr{0}, u1_start, u1_end, u1_col;
    // cache for other n
    std::map<int32_t, std::array<std::vector<int64_t>, 4>> units_n;
    // all reads seen
    std::unordered_map<std::string, int64_t> seen_idx;
    std::vector<SeenRead> seen;
    std::vector<int64_t> rec_seen;  // kept record -> seen index
    std::string discarded;

    int64_t n_reads() const { return (int64_t)read_off.size() - 1; }
};

namespace {

// Light-weight header view of a file record used for the keep decision.
struct Header {
    std::string r_id;
    int64_t r_len, r_al_len, r_st, r_en;
    char strand;
};

// Sequential dictionary logic of NCRF_Report.__init__ (ncrf_parser.py:88-111).
// Returns: -1 skip, otherwise the kept-record slot this alignment (re)fills.
struct Selector {
    cfh_pack* P;
    int64_t min_record_len;
    std::vector<int64_t> rec_al_len;  // r_al_len of the currently kept record
    int64_t offer(const Header& h) {
        auto it = P->seen_idx.find(h.r_id);
        int64_t si;
        if (it == P->seen_idx.end()) {
            si = (int64_t)P->seen.size();
            P->seen_idx.emplace(h.r_id, si);
            P->seen.emplace_back();
        } else si = it->second;
        SeenRead& s = P->seen[si];
        AlnKey key{h.r_st, h.r_en, h.strand};
        if (s.n_aln == 0) { s.first = key; s.last = key; }
        else { if (key < s.first) s.first = key; if (s.last < key) s.last = key; }
        s.n_aln++;
        s.read_len = h.r_len;
        if (s.rec < 0 || rec_al_len[s.rec] < h.r_al_len) {
            if (h.r_al_len < min_record_len) return -1;
            if (s.rec < 0) {
                s.rec = (int64_t)rec_al_len.size();
                rec_al_len.push_back(h.r_al_len);
                P->rec_seen.push_back(si);
            } else rec_al_len[s.rec] = h.r_al_len;
            return s.rec;
        }
        return -1;
    }
};

struct Winner {  // final content of one kept slot
    FileRecord fr;
    bool set = false;
    const char *a1b = nullptr, *a1e = nullptr, *a2b = nullptr, *a2e = nullptr;   // the two alignment rows where they stand in the mapped report (copied by the worker that stages the record)
};

// Every kept record: rows oriented to the read's forward strand, gap columns dropped into the flat base array, unit
// boundaries for n = 1.  Three parallel sweeps with a prefix sum between them, so that every record writes its bases and
// units where they finally stand (no per-record staging copies): (a) bases per record, (b) orientation + bases + unit
// boundaries, (c) the unit arrays.
void finalize_pack(cfh_pack* P, std::vector<Winner>& winners, int n_threads) {
    const int64_t R = (int64_t)winners.size();
    std::vector<Staged> staged((size_t)R);
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, R));
    auto sweep = [&](auto&& body) {
        std::atomic<int64_t> next{0};
        auto work = [&]() {
            while (true) {
                const int64_t i0 = next.fetch_add(16);
                if (i0 >= R) break;
                for (int64_t i = i0; i < std::min(R, i0 + 16); ++i) body(i);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    };
    for (auto& w : winners)          // records that come with their rows (the generator) are viewed like mapped ones
        if (!w.a1b) { w.a1b = w.fr.r_al.data(); w.a1e = w.a1b + w.fr.r_al.size(); w.a2b = w.fr.m_al.data(); w.a2e = w.a2b + w.fr.m_al.size(); }
    PhaseClock clk;
    std::vector<int64_t> b_off((size_t)R + 1, 0), u_off((size_t)R + 1, 0);
    sweep([&](int64_t i) { const Winner& w = winners[(size_t)i]; b_off[(size_t)i + 1] = count_bases(w.a1b, (size_t)(w.a1e - w.a1b)); });
    for (int64_t i = 0; i < R; ++i) b_off[(size_t)i + 1] += b_off[(size_t)i];
    clk.lap("  count bases");
    P->bases.resize((size_t)b_off[(size_t)R]);
    clk.lap("  allocate bases");
    if (P->keep_rows) { P->rows_r.resize((size_t)R); P->rows_m.resize((size_t)R); }
    sweep([&](int64_t i) {
        Winner& w = winners[(size_t)i];
        Staged& st = staged[(size_t)i];
        static thread_local std::string tr, tm;
        const bool minus = w.fr.strand == '-';
        const char *r = w.a1b, *m = w.a2b;
        const size_t rn = (size_t)(w.a1e - w.a1b), mn = (size_t)(w.a2e - w.a2b);
        if (minus) { rc_copy(w.a1b, rn, tr); rc_copy(w.a2b, mn, tm); r = tr.data(); m = tm.data(); }
        st.ncols = (int64_t)rn;
        unsigned char* b = (unsigned char*)P->bases.data() + b_off[(size_t)i];
        unsigned other = 0;
        for (size_t c = 0; c < rn; ++c) {         // gap columns are skipped: a base is written only where it stays
            const unsigned char ch = (unsigned char)r[c];
            if (ch != '-') { *b++ = ch; other |= BT.other[ch]; }
        }
        st.non_acgt = other != 0;
        split_units(rn, m, mn, w.fr.motif, 1, st.col_bounds);
        cols_to_pos(r, rn, st.col_bounds, st.pos_bounds);
        u_off[(size_t)i + 1] = st.pos_bounds.empty() ? 0 : (int64_t)st.pos_bounds.size() - 1;
        if (P->keep_rows) {
            if (minus) { P->rows_r[(size_t)i] = tr; P->rows_m[(size_t)i] = tm; }
            else { P->rows_r[(size_t)i].assign(r, rn); P->rows_m[(size_t)i].assign(m, mn); }
        }
        std::string().swap(w.fr.r_al); std::string().swap(w.fr.m_al);
        w.a1b = w.a1e = w.a2b = w.a2e = nullptr;
    });
    clk.lap("  orient + bases + units");
    for (int64_t i = 0; i < R; ++i) u_off[(size_t)i + 1] += u_off[(size_t)i];
    const int64_t nu = u_off[(size_t)R];
    P->u1_start.resize((size_t)nu); P->u1_end.resize((size_t)nu); P->u1_col.resize((size_t)nu * 2);
    P->read_off = b_off;
    P->u1_ptr = u_off;
    P->meta.resize((size_t)R * 8);
    // small sequential part: ids, motif ids (first-seen order), flags
    std::vector<int32_t> mids((size_t)R);
    for (int64_t i = 0; i < R; ++i) {
        FileRecord& fr = winners[(size_t)i].fr;
        P->ids += fr.r_id;
        P->id_off.push_back((int64_t)P->ids.size());
        auto it = P->motif_ids.find(fr.motif);
        if (it == P->motif_ids.end()) { mids[(size_t)i] = (int32_t)P->motifs.size(); P->motif_ids.emplace(fr.motif, mids[(size_t)i]); P->motifs.push_back(fr.motif); }
        else mids[(size_t)i] = it->second;
        if (staged[(size_t)i].non_acgt) P->non_acgt = true;
    }
    sweep([&](int64_t i) {
        const FileRecord& fr = winners[(size_t)i].fr;
        const Staged& st = staged[(size_t)i];
        const int64_t base0 = b_off[(size_t)i];
        int64_t r_st = fr.r_st, r_en = fr.r_en;
        if (fr.strand == '-') { r_st = fr.r_len - fr.r_en; r_en = fr.r_len - fr.r_st; }
        int64_t* m = &P->meta[(size_t)i * 8];
        m[0] = fr.r_len; m[1] = fr.r_al_len; m[2] = r_st; m[3] = r_en;
        m[4] = fr.strand == '-' ? 1 : 0; m[5] = P->seen[(size_t)P->rec_seen[(size_t)i]].n_aln;
        m[6] = st.ncols; m[7] = mids[(size_t)i];
        int64_t u = u_off[(size_t)i];
        for (size_t b = 0; b + 1 < st.pos_bounds.size(); ++b, ++u) {
            P->u1_start[(size_t)u] = base0 + st.pos_bounds[b];
            P->u1_end[(size_t)u] = base0 + st.pos_bounds[b + 1];
            P->u1_col[(size_t)u * 2] = st.col_bounds[b];
            P->u1_col[(size_t)u * 2 + 1] = st.col_bounds[b + 1];
        }
    });
    clk.lap("  unit arrays + meta");
    // discarded reads: seen but never kept (order is not defined by the reference: it goes through set())
    for (auto& kv : P->seen_idx) {
        if (P->seen[(size_t)kv.second].rec < 0) { P->discarded += kv.first; P->discarded.push_back('\n'); }
    }
}

// ---------------------------------------------------------------- text parsing
bool parse_int(const char*& p, const char* end, int64_t& v) {
    if (p >= end || *p < '0' || *p > '9') return false;
    v = 0;
    while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); ++p; }
    return true;
}
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; }
bool skip_ws1(const char*& p, const char* end) {  // \s+
    if (p >= end || !is_ws(*p)) return false;
    while (p < end && is_ws(*p)) ++p;
    return true;
}

// ^([^ ]+)\s+(\d+)\s+(\d+)bp\s+(\d+)-(\d+)\s+(.+)$
static bool parse_first_from(const char* b, const char* p, const char* e, FileRecord& fr, const char*& al_b, const char*& al_e);
bool parse_first(const char* b, const char* e, FileRecord& fr, const char*& al_b, const char*& al_e) {
    const char* p = b;
    while (p < e && *p != ' ') ++p;
    if (p == b) return false;
    // [^ ]+ is greedy but must be followed by \s+ and the rest of the pattern: the id ends at the first blank — or, when the rest does not
    // match from there, at the LAST tab (or other white space that is not a blank) inside that stretch from which it does: the regex
    // backtracks (round 5, tools/fuzz_parser_vs_reference.py: a tab between the id and the read length).  NCRF itself pads with blanks.
    if (parse_first_from(b, p, e, fr, al_b, al_e)) return true;
    for (const char* q = p - 1; q > b; --q)
        if (is_ws(*q) && parse_first_from(b, q, e, fr, al_b, al_e)) return true;
    return false;
}
static bool parse_first_from(const char* b, const char* p, const char* e, FileRecord& fr, const char*& al_b, const char*& al_e) {
    fr.r_id.assign(b, p);
    if (!skip_ws1(p, e)) return false;
    if (!parse_int(p, e, fr.r_len)) return false;
    if (!skip_ws1(p, e)) return false;
    if (!parse_int(p, e, fr.r_al_len)) return false;
    if (e - p < 2 || p[0] != 'b' || p[1] != 'p') return false;
    p += 2;
    if (!skip_ws1(p, e)) return false;
    if (!parse_int(p, e, fr.r_st)) return false;
    if (p >= e || *p != '-') return false;
    ++p;
    if (!parse_int(p, e, fr.r_en)) return false;
    if (!skip_ws1(p, e)) return false;
    if (p >= e) return false;
    al_b = p; al_e = e;
    return true;
}
// ^([^+-]+)([+-])\s+(\d+)bp\s+score=(\d+)\s+(.+)$
bool parse_second(const char* b, const char* e, FileRecord& fr, const char*& al_b, const char*& al_e) {
    const char* p = b;
    while (p < e && *p != '+' && *p != '-') ++p;
    if (p == b || p >= e) return false;
    fr.motif.assign(b, p);
    fr.strand = *p++;
    if (!skip_ws1(p, e)) return false;
    if (!parse_int(p, e, fr.m_al_len)) return false;
    if (e - p < 2 || p[0] != 'b' || p[1] != 'p') return false;
    p += 2;
    if (!skip_ws1(p, e)) return false;
    if (e - p < 6 || std::strncmp(p, "score=", 6) != 0) return false;
    p += 6;
    if (!parse_int(p, e, fr.score)) return false;
    if (!skip_ws1(p, e)) return false;
    if (p >= e) return false;
    al_b = p; al_e = e;
    return true;
}

// ---------------------------------------------------------------- PRNG
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed, uint64_t stream = 0) {
        uint64_t x = seed * 0xD1342543DE82EF95ull + stream * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        for (auto& v : s) v = splitmix(x);
    }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? (uint64_t)(uniform() * (double)n) % n : 0; }
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    char base() { return "ACGT"[next() >> 62]; }
    char other(char c) {
        while (true) { char b = base(); if (b != c) return b; }
    }
};

struct Genome {
    std::string seq;      // flank + array + flank
    std::string motif;    // consensus HOR unit
    int64_t a0 = 0, a1 = 0;
};

Genome make_genome(const cfh_synth_params& sp) {
    Genome g;
    Rng rng(sp.seed, 1);
    std::string anc((size_t)sp.monomer_len, 'A');
    for (auto& c : anc) c = rng.base();
    while ((int64_t)g.motif.size() + sp.monomer_len <= sp.unit_len) {
        std::string m = anc;
        for (auto& c : m) if (rng.uniform() < sp.monomer_div) c = rng.other(c);
        g.motif += m;
    }
    while ((int64_t)g.motif.size() < sp.unit_len) g.motif.push_back(rng.base());
    g.seq.reserve((size_t)(2 * sp.flank + sp.n_units * sp.unit_len));
    for (int64_t i = 0; i < sp.flank; ++i) g.seq.push_back(rng.base());
    g.a0 = (int64_t)g.seq.size();
    for (int64_t u = 0; u < sp.n_units; ++u) {
        Rng ur(sp.seed, 1000 + (uint64_t)u);
        // copy-specific variants: at each chosen site replace var_len consecutive bases (the first
        // always changes).  var_len = 1 is the point-substitution model of
        // simulate_tandem_repeat.py:15-30; longer variants keep copy-specific k-mers unique when
        // the array has 10^4-10^5 copies (only 3 * unit_len distinct point substitutions exist).
        const int64_t ulen = (int64_t)g.motif.size();
        const int vl = sp.var_len > 1 ? sp.var_len : 1;
        const size_t u0 = g.seq.size();
        g.seq.append(g.motif);
        for (int64_t i = 0; i < ulen; ++i) {
            if (!(ur.uniform() < sp.unit_div)) continue;
            g.seq[u0 + (size_t)i] = ur.other(g.motif[(size_t)i]);
            for (int j = 1; j < vl && i + j < ulen; ++j) g.seq[u0 + (size_t)(i + j)] = ur.base();
        }
    }
    g.a1 = (int64_t)g.seq.size();
    Rng fr(sp.seed, 2);
    for (int64_t i = 0; i < sp.flank; ++i) g.seq.push_back(fr.base());
    return g;
}

// Simulate candidate read c; fills recs with 0 (rejected), 1 or 2 file records.
void simulate_read(const cfh_synth_params& sp, const Genome& g, int64_t c, std::vector<FileRecord>& recs) {
    recs.clear();
    Rng rng(sp.seed, 100000 + (uint64_t)c);
    const int64_t G = (int64_t)g.seq.size();
    int64_t len, start;
    if (c < sp.n_prefix) {
        int64_t delta = 1000 + (int64_t)rng.below(2000);
        int64_t inside = 8000 + (int64_t)rng.below(22000);
        inside = std::min(inside, g.a1 - g.a0);
        start = g.a0 - sp.prefix_threshold - delta;
        len = sp.prefix_threshold + delta + inside;
    } else if (c < sp.n_prefix + sp.n_suffix) {
        int64_t delta = 1000 + (int64_t)rng.below(2000);
        int64_t inside = 8000 + (int64_t)rng.below(22000);
        inside = std::min(inside, g.a1 - g.a0);
        start = g.a1 - inside;
        len = sp.prefix_threshold + delta + inside;
    } else {
        double mu = std::log(sp.mean_len) - 0.5 * sp.sigma * sp.sigma;
        double l = std::exp(mu + sp.sigma * rng.normal());
        len = (int64_t)l;
        len = std::max(sp.min_len, std::min(sp.max_len, len));
        len = std::min(len, G);
        start = (int64_t)rng.below((uint64_t)(G - len + 1));
    }
    if (start < 0) start = 0;
    if (start + len > G) len = G - start;
    const bool minus = (rng.next() >> 63) != 0;
    // overlap with the array
    const int64_t o0 = std::max(start, g.a0), o1 = std::min(start + len, g.a1);
    if (o1 - o0 < sp.min_aligned / 2) return;  // cannot reach min_aligned
    // walk the genome segment in forward orientation
    std::string r_al, m_al;
    r_al.reserve((size_t)((o1 - o0) * 11 / 10));
    m_al.reserve((size_t)((o1 - o0) * 11 / 10));
    int64_t before = 0, aligned = 0, after = 0;  // read bases before / inside / after the array
    const int64_t ulen = (int64_t)g.motif.size();
    for (int64_t p = start; p < start + len; ++p) {
        const bool in_arr = p >= g.a0 && p < g.a1;
        const char gb = g.seq[(size_t)p];
        const char mb = in_arr ? g.motif[(size_t)((p - g.a0) % ulen)] : 0;
        if (rng.uniform() < sp.p_del) {
            if (in_arr) { r_al.push_back('-'); m_al.push_back(mb); }
        } else {
            char rb = rng.uniform() < sp.p_sub ? rng.other(gb) : gb;
            if (in_arr) { r_al.push_back(rb); m_al.push_back(mb); ++aligned; }
            else if (p < g.a0) ++before; else ++after;
        }
        if (rng.uniform() < sp.p_ins) {
            char ib = rng.base();
            // insertions at the very edge of the array belong to the flank part
            if (in_arr && p + 1 < g.a1 && p + 1 < start + len) { r_al.push_back(ib); m_al.push_back('-'); ++aligned; }
            else if (p < g.a0) ++before; else ++after;
        }
    }
    // trim alignment so it neither starts nor ends with a gap column in either row
    size_t lo = 0, hi = r_al.size();
    while (lo < hi && (r_al[lo] == '-' || m_al[lo] == '-')) { if (r_al[lo] != '-') { --aligned; ++before; } ++lo; }
    while (hi > lo && (r_al[hi - 1] == '-' || m_al[hi - 1] == '-')) { if (r_al[hi - 1] != '-') { --aligned; ++after; } --hi; }
    r_al = r_al.substr(lo, hi - lo);
    m_al = m_al.substr(lo, hi - lo);
    if (aligned < sp.min_aligned) return;
    const int64_t r_len = before + aligned + after;

    // describe as one or two records (forward coordinates first)
    struct Piece { size_t c0, c1; int64_t st, en; };
    std::vector<Piece> pieces;
    bool split = sp.p_split > 0 && rng.uniform() < sp.p_split && r_al.size() > 4000;
    if (split) {
        size_t cut = r_al.size() / 5 + (size_t)rng.below((uint64_t)(r_al.size() * 3 / 5));
        while (cut < r_al.size() && (r_al[cut] == '-' || m_al[cut] == '-' || r_al[cut - 1] == '-' || m_al[cut - 1] == '-')) ++cut;
        if (cut + 10 >= r_al.size()) split = false;
        else {
            int64_t n1 = 0;
            for (size_t i = 0; i < cut; ++i) n1 += r_al[i] != '-';
            pieces.push_back({0, cut, before, before + n1});
            pieces.push_back({cut, r_al.size(), before + n1, before + aligned});
        }
    }
    if (!split) pieces.push_back({0, r_al.size(), before, before + aligned});
    char idbuf[64];
    std::snprintf(idbuf, sizeof idbuf, "read_%08lld", (long long)c);
    for (auto& pc : pieces) {
        FileRecord fr;
        fr.r_id = idbuf;
        fr.r_len = r_len;
        fr.r_al = r_al.substr(pc.c0, pc.c1 - pc.c0);
        fr.m_al = m_al.substr(pc.c0, pc.c1 - pc.c0);
        fr.r_al_len = pc.en - pc.st;
        fr.m_al_len = 0;
        for (char ch : fr.m_al) fr.m_al_len += ch != '-';
        fr.motif = g.motif;
        fr.score = fr.r_al_len;
        if (minus) {
            fr.strand = '-';
            fr.r_st = r_len - pc.en; fr.r_en = r_len - pc.st;
            rc_inplace(fr.r_al); rc_inplace(fr.m_al);
        } else {
            fr.strand = '+';
            fr.r_st = pc.st; fr.r_en = pc.en;
        }
        recs.push_back(std::move(fr));
    }
}

void write_record(FILE* f, const FileRecord& fr) {
    std::fprintf(f, "%s %lld %lldbp %lld-%lld %s\n", fr.r_id.c_str(), (long long)fr.r_len,
                 (long long)fr.r_al_len, (long long)fr.r_st, (long long)fr.r_en, fr.r_al.c_str());
    std::fprintf(f, "%s%c %lldbp score=%lld %s\n\n", fr.motif.c_str(), fr.strand,
                 (long long)fr.m_al_len, (long long)fr.score, fr.m_al.c_str());
}

}  // namespace

// ================================================================== C ABI
static void decode_kmer(uint64_t code, int k, char* out) {
    for (int i = k - 1; i >= 0; --i) { out[i] = "ACGT"[code & 3]; code >>= 2; }
}

// Text files written in record order by many threads: the records [0, n) are cut into tasks, workers render tasks into
// the buffers of a ring and the calling thread writes the buffers to the file in task order (rendering 150 GB of edge
// lines — BASELINE configs[1] — is the work; the file system takes whole buffers).
template <class Render>      // render(lo, hi, std::vector<char>& out) appends the text of records [lo, hi)
static int write_ordered(int fd, int64_t n, int64_t per_task, Render&& render) {
    const int64_t n_tasks = (n + per_task - 1) / per_task;
    auto put = [&](const std::vector<char>& buf) -> bool {
        size_t off = 0;
        while (off < buf.size()) {
            const ssize_t w = ::write(fd, buf.data() + off, buf.size() - off);
            if (w < 0) { if (errno == EINTR) continue; return false; }
            off += (size_t)w;
        }
        return true;
    };
    if (n_tasks <= 1) {
        std::vector<char> buf;
        if (n > 0) render((int64_t)0, n, buf);
        return put(buf) ? 0 : -5;
    }
    const int nt = (int)std::min<int64_t>(std::min<int64_t>(n_tasks, 32), std::max(1u, std::thread::hardware_concurrency()));
    const int64_t S = std::min<int64_t>(n_tasks, 2 * (int64_t)nt);
    struct Slot { std::vector<char> buf; std::atomic<int64_t> turn{0}, full{0}; };     // turn: the task that may render into the slot; full: task + 1 once rendered
    std::vector<Slot> ring((size_t)S);
    for (int64_t i = 0; i < S; ++i) ring[(size_t)i].turn = i;
    std::atomic<int64_t> next{0};
    std::atomic<bool> stop{false};
    auto work = [&]() {
        while (!stop.load(std::memory_order_relaxed)) {
            const int64_t t = next.fetch_add(1);
            if (t >= n_tasks) break;
            Slot& sl = ring[(size_t)(t % S)];
            while (sl.turn.load(std::memory_order_acquire) != t) { if (stop.load(std::memory_order_relaxed)) return; std::this_thread::yield(); }
            sl.buf.clear();
            render(t * per_task, std::min(n, (t + 1) * per_task), sl.buf);
            sl.full.store(t + 1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(work);
    int rc = 0;
    for (int64_t t = 0; t < n_tasks; ++t) {
        Slot& sl = ring[(size_t)(t % S)];
        while (sl.full.load(std::memory_order_acquire) != t + 1) std::this_thread::yield();
        if (!put(sl.buf)) { rc = -5; stop = true; break; }
        sl.turn.store(t + S, std::memory_order_release);
    }
    for (auto& x : th) x.join();
    return rc;
}

static inline char* put_u32(char* p, uint32_t v) {      // decimal, no padding
    char tmp[10];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

namespace {
const char kPackMagic[8] = {'C', 'F', 'P', 'A', 'C', 'K', '0', '2'};
struct PackWriter {
    FILE* f; bool ok = true;
    void raw(const void* p, size_t n) { if (ok && n && std::fwrite(p, 1, n, f) != n) ok = false; }
    void u64(uint64_t v) { raw(&v, 8); }
    template <class T> void vec(const std::vector<T>& v) { u64(v.size()); raw(v.data(), v.size() * sizeof(T)); }
    void str(const std::string& v) { u64(v.size()); raw(v.data(), v.size()); }
    void strs(const std::vector<std::string>& v) { u64(v.size()); for (auto& x : v) str(x); }
};
struct PackReader {
    const unsigned char* p; const unsigned char* e; bool ok = true;
    bool raw(void* d, size_t n) { if (!ok || (size_t)(e - p) < n) { ok = false; return false; } std::memcpy(d, p, n); p += n; return true; }
    uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
    template <class T> void vec(std::vector<T>& v) { const uint64_t n = u64(); if (!ok || n > (uint64_t)(e - p) / sizeof(T)) { ok = false; return; } v.resize((size_t)n); raw(v.data(), (size_t)n * sizeof(T)); }
    void str(std::string& v) { const uint64_t n = u64(); if (!ok || n > (uint64_t)(e - p)) { ok = false; return; } v.assign((const char*)p, (size_t)n); p += n; }
    void strs(std::vector<std::string>& v) { const uint64_t n = u64(); if (!ok || n > (uint64_t)(e - p)) { ok = false; return; } v.resize((size_t)n); for (auto& x : v) str(x); }
};
struct SeenFlat { int64_t read_len, f_st, f_en, l_st, l_en, n_aln, rec; int64_t f_strand, l_strand; };
// checksum of a byte range by all threads: 1 MiB blocks, a multiplicative hash over the 8-byte words of each, block hashes
// combined with their index (a flipped bit anywhere in a gigabyte of bases must not pass for a valid cache)
uint64_t pack_checksum(const unsigned char* p, size_t n) {
    const size_t B = (size_t)1 << 20, nb = (n + B - 1) / B;
    const int T = (int)std::max<size_t>(1, std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), nb));
    std::vector<uint64_t> part((size_t)T, 0);
    std::atomic<size_t> next{0};
    auto work = [&](int t) {
        uint64_t acc = 0;
        while (true) {
            const size_t b = next.fetch_add(1);
            if (b >= nb) break;
            const size_t lo = b * B, hi = std::min(n, lo + B);
            uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)b;
            size_t i = lo;
            for (; i + 8 <= hi; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001B3ull; h ^= h >> 29; }
            for (; i < hi; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
            acc += h * (2 * (uint64_t)b + 1);
        }
        part[(size_t)t] = acc;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    uint64_t sum = 0;
    for (uint64_t v : part) sum += v;
    return sum;
}
}  // namespace

extern "C" {

void cfh_synth_defaults(cfh_synth_params* p) {
    std::memset(p, 0, sizeof *p);
    p->seed = 1; p->unit_len = 2055; p->monomer_len = 171; p->monomer_div = 0.25;
    p->n_units = 300; p->flank = 200000; p->unit_div = 0.01; p->n_reads = 1000;
    p->mean_len = 20000; p->sigma = 0.5; p->min_len = 6000; p->max_len = 200000;
    p->p_del = 0.02; p->p_sub = 0.02; p->p_ins = 0.015; p->min_aligned = 5000;
    p->n_prefix = 8; p->n_suffix = 8; p->prefix_threshold = 50000; p->p_split = 0.0;
    p->n_threads = 0; p->var_len = 1; p->cand_offset = 0; p->cand_stride = 1;
}

int cfh_synth(const cfh_synth_params* sp, const char* report_path, int keep_rows,
              cfh_pack** out, char* err, int errlen) {
    try {
        if (sp->unit_len <= 0 || sp->n_units <= 0 || sp->n_reads < 0 || sp->monomer_len <= 0) {
            set_err(err, errlen, "cfh_synth: bad parameters"); return -22;
        }
        Genome g = make_genome(*sp);
        FILE* f = nullptr;
        if (report_path) {
            f = std::fopen(report_path, "w");
            if (!f) { set_err(err, errlen, std::string("cfh_synth: cannot open ") + report_path); return -2; }
            std::fprintf(f, "# synthetic NCRF-format report: seed=%llu unit_len=%d n_units=%lld\n",
                         (unsigned long long)sp->seed, sp->unit_len, (long long)sp->n_units);
        }
        std::unique_ptr<cfh_pack> P;
        Selector sel{nullptr, 5000, {}};  // NCRF_Report's default min_record_len
        std::vector<Winner> winners;
        if (out) { P.reset(new cfh_pack()); P->keep_rows = keep_rows != 0; sel.P = P.get(); }
        int nt = sp->n_threads > 0 ? sp->n_threads : (int)std::max(1u, std::thread::hardware_concurrency());
        int64_t emitted = 0, cand = 0;
        const int64_t batch = 256;
        int64_t guard = 0;
        while (emitted < sp->n_reads) {
            std::vector<std::vector<FileRecord>> res((size_t)batch);
            std::atomic<int64_t> next{0};
            auto work = [&]() {
                while (true) {
                    int64_t i = next.fetch_add(1);
                    if (i >= batch) break;
                    simulate_read(*sp, g, (int64_t)sp->cand_offset + (cand + i) * (int64_t)(sp->cand_stride > 0 ? sp->cand_stride : 1), res[(size_t)i]);
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(work);
            work();
            for (auto& t : th) t.join();
            int64_t got = 0;
            for (int64_t i = 0; i < batch && emitted < sp->n_reads; ++i) {
                auto& rs = res[(size_t)i];
                if (rs.empty()) continue;
                ++got;
                for (auto& fr : rs) {
                    if (f) write_record(f, fr);
                    if (P) {
                        Header h{fr.r_id, fr.r_len, fr.r_al_len, fr.r_st, fr.r_en, fr.strand};
                        int64_t slot = sel.offer(h);
                        if (slot >= 0) {
                            if ((int64_t)winners.size() <= slot) winners.resize((size_t)slot + 1);
                            winners[(size_t)slot].fr = std::move(fr);
                            winners[(size_t)slot].set = true;
                        }
                    }
                }
                ++emitted;
            }
            cand += batch;
            guard = got ? 0 : guard + 1;
            if (guard > 2000) { if (f) std::fclose(f); set_err(err, errlen, "cfh_synth: no read reaches min_aligned"); return -34; }
        }
        if (f) std::fclose(f);
        if (P) { finalize_pack(P.get(), winners, nt); *out = P.release(); }
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errlen, std::string("cfh_synth: ") + e.what());
        return -12;
    }
}

// The report is mapped and read by all threads: (1) every thread finds the content lines (not blank, not '#') of its
// slice of the file; (2) content lines pair up into records in file order and every thread parses the two HEADERS of its
// share of the records (the alignment rows stay where they are); (3) the dictionary logic of NCRF_Report.__init__ — which
// alignment of a read is kept, in which order reads first appear — runs sequentially over the headers alone (no row is
// copied for an alignment that loses); (4) the kept records are oriented, de-gapped and split into units by all threads
// (finalize_pack), straight from the mapped rows.
int cfh_parse_report(const char* path, int64_t min_record_len, int keep_rows, int n_threads,
                     cfh_pack** out, char* err, int errlen) {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    bool mapped = false;
    std::string fallback;
    try {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { set_err(err, errlen, std::string("cannot open NCRF report ") + path); return -2; }
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { data = (const char*)m; size = (size_t)st.st_size; mapped = true; }
        }
        if (!mapped) {          // a pipe, an empty file: read it whole
            char buf[1 << 16];
            ssize_t n;
            while ((n = ::read(fd, buf, sizeof buf)) > 0) fallback.append(buf, (size_t)n);
            data = fallback.data(); size = fallback.size();
        }
        PhaseClock clk;
        if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
        size_t min_slice = 1 << 20;      // bytes of report per scanning thread at least (CFH_PARSE_MIN_SLICE: smaller in tests)
        if (const char* e = std::getenv("CFH_PARSE_MIN_SLICE")) min_slice = (size_t)std::max(1L, std::atol(e));
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, size / min_slice + 1));
        // (1) content lines per slice; a slice starts behind the first newline at or after its nominal start
        struct Line { const char *b, *e; int64_t lineno; };
        std::vector<std::vector<Line>> lines((size_t)T);
        std::vector<int64_t> n_lines((size_t)T, 0);      // all lines of the slice (for line numbers in messages)
        auto slice_begin = [&](int t) -> size_t {
            if (t == 0) return 0;
            if (t >= T) return size;
            size_t p = size / (size_t)T * (size_t)t;
            const void* nl = p < size ? std::memchr(data + p - 1, '\n', size - p + 1) : nullptr;     // (p - 1: a slice that starts right behind a newline keeps its first line)
            return nl ? (size_t)((const char*)nl - data) + 1 : size;
        };
        auto scan = [&](int t) {
            size_t p = slice_begin(t);
            const size_t end = slice_begin(t + 1);
            int64_t ln = 0;
            while (p < end) {
                const void* nl = std::memchr(data + p, '\n', size - p);
                const size_t q = nl ? (size_t)((const char*)nl - data) : size;
                ++ln;
                const char* b = data + p; const char* e = data + q;
                while (b < e && is_ws(*b)) ++b;
                while (e > b && is_ws(e[-1])) --e;
                if (b != e && *b != '#') lines[(size_t)t].push_back(Line{b, e, ln});
                p = q + 1;
            }
            n_lines[(size_t)t] = ln;
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(scan, t);
            scan(0);
            for (auto& x : th) x.join();
        }
        clk.lap("line scan");
        std::vector<Line> all;
        {
            size_t tot = 0;
            for (auto& v : lines) tot += v.size();
            all.reserve(tot);
            int64_t base = 0;
            for (int t = 0; t < T; ++t) { for (auto& l : lines[(size_t)t]) all.push_back(Line{l.b, l.e, base + l.lineno}); base += n_lines[(size_t)t]; lines[(size_t)t] = std::vector<Line>(); }
        }
        auto cleanup = [&]() { if (mapped) munmap((void*)data, size); if (fd >= 0) ::close(fd); fd = -1; mapped = false; };
        const int64_t n_rec = (int64_t)all.size() / 2;
        // (2) headers of every record
        struct Parsed { FileRecord fr; const char *a1b, *a1e, *a2b, *a2e; bool ok; };
        std::vector<Parsed> recs((size_t)n_rec);
        {
            std::atomic<int64_t> next{0};
            auto work = [&]() {
                while (true) {
                    const int64_t i0 = next.fetch_add(64);
                    if (i0 >= n_rec) break;
                    for (int64_t i = i0; i < std::min(n_rec, i0 + 64); ++i) {
                        Parsed& r = recs[(size_t)i];
                        const Line &l1 = all[(size_t)2 * i], &l2 = all[(size_t)2 * i + 1];
                        r.ok = parse_first(l1.b, l1.e, r.fr, r.a1b, r.a1e) && parse_second(l2.b, l2.e, r.fr, r.a2b, r.a2e);
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(work);
            work();
            for (auto& x : th) x.join();
        }
        clk.lap("headers");
        for (int64_t i = 0; i < n_rec; ++i)
            if (!recs[(size_t)i].ok) {
                const int64_t ln = all[(size_t)2 * i + 1].lineno;
                cleanup();
                set_err(err, errlen, "malformed NCRF record ending at line " + std::to_string(ln) + " of " + path);
                return -22;
            }
        if (all.size() & 1) { cleanup(); set_err(err, errlen, std::string("odd number of record lines in ") + path); return -22; }
        // (3) which alignment of which read is kept: sequential, headers only
        std::unique_ptr<cfh_pack> P(new cfh_pack());
        P->keep_rows = keep_rows != 0;
        Selector sel{P.get(), min_record_len, {}};
        std::vector<int64_t> keep_idx;          // slot -> record
        for (int64_t i = 0; i < n_rec; ++i) {
            const FileRecord& fr = recs[(size_t)i].fr;
            Header h{fr.r_id, fr.r_len, fr.r_al_len, fr.r_st, fr.r_en, fr.strand};
            const int64_t slot = sel.offer(h);
            if (slot >= 0) {
                if ((int64_t)keep_idx.size() <= slot) keep_idx.resize((size_t)slot + 1, -1);
                keep_idx[(size_t)slot] = i;
            }
        }
        clk.lap("selection");
        // (4) stage the kept records from the mapped rows
        std::vector<Winner> winners(keep_idx.size());
        for (size_t sl = 0; sl < keep_idx.size(); ++sl) {
            Parsed& r = recs[(size_t)keep_idx[sl]];
            Winner& w = winners[sl];
            w.fr = std::move(r.fr); w.set = true;
            w.a1b = r.a1b; w.a1e = r.a1e; w.a2b = r.a2b; w.a2e = r.a2e;
        }
        recs = std::vector<Parsed>();
        finalize_pack(P.get(), winners, n_threads);
        clk.lap("staging + assembly");
        cleanup();
        *out = P.release();
        return 0;
    } catch (const std::exception& e) {
        if (mapped) munmap((void*)data, size);
        if (fd >= 0) ::close(fd);
        set_err(err, errlen, std::string("cfh_parse_report: ") + e.what());
        return -12;
    }
}

void cfh_pack_free(cfh_pack* p) { delete p; }

// ---------------------------------------------------------------- binary cache of a pack
// Both stage scripts parse the same report; the second one can load what the first one packed.  One file: a header with
// the identity of the source report as the caller states it (size, mtime, min_record_len, keep_rows), then every array of
// the pack as (count, bytes).  A file that does not match, is truncated or was written by another version is refused
// (-61): the caller parses the report instead.

int cfh_pack_save(const cfh_pack* p, const char* path, const int64_t* source_id, char* err, int errlen) {
    try {
        const std::string tmp = std::string(path) + ".tmp" + std::to_string((long long)getpid());
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) { set_err(err, errlen, std::string("cannot open ") + tmp); return -2; }
        PackWriter w{f};
        w.raw(kPackMagic, 8);
        for (int i = 0; i < 4; ++i) w.u64((uint64_t)source_id[i]);
        w.u64(p->keep_rows ? 1 : 0); w.u64(p->non_acgt ? 1 : 0);
        w.vec(p->meta); w.str(p->ids); w.vec(p->id_off); w.vec(p->read_off);
        w.u64(p->bases.size()); w.raw(p->bases.data(), p->bases.size());
        w.strs(p->motifs);
        w.vec(p->u1_ptr); w.vec(p->u1_start); w.vec(p->u1_end); w.vec(p->u1_col);
        std::vector<SeenFlat> sf(p->seen.size());
        for (size_t i = 0; i < sf.size(); ++i) {
            const SeenRead& s = p->seen[i];
            sf[i] = SeenFlat{s.read_len, s.first.st, s.first.en, s.last.st, s.last.en, s.n_aln, s.rec, (int64_t)s.first.strand, (int64_t)s.last.strand};
        }
        w.vec(sf); w.vec(p->rec_seen); w.str(p->discarded);
        if (p->keep_rows) { w.strs(p->rows_r); w.strs(p->rows_m); }
        bool ok = w.ok;
        if (std::fclose(f) != 0 || !ok) { std::remove(tmp.c_str()); set_err(err, errlen, std::string("write failed: ") + tmp); return -5; }
        {       // checksum of what was written (from the page cache, by all threads), then the end marker: a truncated file is not a cache
            const int fd = ::open(tmp.c_str(), O_RDWR);
            struct stat st;
            ok = fd >= 0 && fstat(fd, &st) == 0;
            if (ok) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                ok = m != MAP_FAILED;
                if (ok) {
                    const uint64_t sum = pack_checksum((const unsigned char*)m, (size_t)st.st_size);
                    munmap(m, (size_t)st.st_size);
                    unsigned char tail[16];
                    std::memcpy(tail, &sum, 8); std::memcpy(tail + 8, kPackMagic, 8);
                    ok = ::lseek(fd, 0, SEEK_END) >= 0 && ::write(fd, tail, 16) == 16;
                }
            }
            if (fd >= 0 && ::close(fd) != 0) ok = false;
            if (!ok) { std::remove(tmp.c_str()); set_err(err, errlen, std::string("write failed: ") + tmp); return -5; }
        }
        if (std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); set_err(err, errlen, std::string("cannot rename to ") + path); return -5; }
        return 0;
    } catch (const std::exception& e) { set_err(err, errlen, std::string("cfh_pack_save: ") + e.what()); return -12; }
}

int cfh_pack_load(const char* path, const int64_t* source_id, cfh_pack** out, char* err, int errlen) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { set_err(err, errlen, std::string("cannot open ") + path); return -2; }
    struct stat st;
    void* m = MAP_FAILED;
    size_t size = 0;
    if (fstat(fd, &st) == 0 && st.st_size >= 32) { size = (size_t)st.st_size; m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0); }
    ::close(fd);
    if (m == MAP_FAILED) { set_err(err, errlen, std::string("not a pack cache: ") + path); return -61; }
    int rc = -61;
    try {
        PackReader r{(const unsigned char*)m, (const unsigned char*)m + size};
        char magic[8];
        std::unique_ptr<cfh_pack> P(new cfh_pack());
        bool good = r.raw(magic, 8) && std::memcmp(magic, kPackMagic, 8) == 0 && std::memcmp((const char*)m + size - 8, kPackMagic, 8) == 0;
        for (int i = 0; good && i < 4; ++i) good = (int64_t)r.u64() == source_id[i];
        if (good) { uint64_t want; std::memcpy(&want, (const char*)m + size - 16, 8); good = pack_checksum((const unsigned char*)m, size - 16) == want; }
        if (good) {
            P->keep_rows = r.u64() != 0; P->non_acgt = r.u64() != 0;
            r.vec(P->meta); r.str(P->ids); r.vec(P->id_off); r.vec(P->read_off);
            const uint64_t nb = r.u64();
            if (r.ok && nb <= (uint64_t)(r.e - r.p)) {
                P->bases.resize((size_t)nb);
                // the gigabyte of bases: copied by all threads (each the first to touch its part of the new pages)
                const int T = (int)std::max<size_t>(1, std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), (size_t)nb / (8 << 20) + 1));
                std::vector<std::thread> th;
                const unsigned char* src = r.p;
                for (int t = 0; t < T; ++t) {
                    const size_t lo = (size_t)nb * (size_t)t / (size_t)T, hi = (size_t)nb * (size_t)(t + 1) / (size_t)T;
                    th.emplace_back([=, &P]() { if (hi > lo) std::memcpy(P->bases.data() + lo, src + lo, hi - lo); });
                }
                for (auto& x : th) x.join();
                r.p += nb;
            } else r.ok = false;
            r.strs(P->motifs);
            r.vec(P->u1_ptr); r.vec(P->u1_start); r.vec(P->u1_end); r.vec(P->u1_col);
            std::vector<SeenFlat> sf;
            r.vec(sf); r.vec(P->rec_seen); r.str(P->discarded);
            if (P->keep_rows) { r.strs(P->rows_r); r.strs(P->rows_m); }
            const int64_t R = (int64_t)P->read_off.size() - 1;
            good = r.ok && (size_t)(r.e - r.p) == 16 && R >= 0 && (int64_t)P->meta.size() == R * 8 && (int64_t)P->id_off.size() == R + 1 &&
                   (int64_t)P->u1_ptr.size() == R + 1 && (int64_t)P->rec_seen.size() == R && (!P->keep_rows || ((int64_t)P->rows_r.size() == R && (int64_t)P->rows_m.size() == R)) &&
                   (R == 0 || ((uint64_t)P->read_off[(size_t)R] == nb && (uint64_t)P->id_off[(size_t)R] == P->ids.size())) &&
                   P->u1_start.size() == P->u1_end.size() && P->u1_col.size() == 2 * P->u1_start.size() && (P->u1_ptr.empty() || (uint64_t)P->u1_ptr.back() == P->u1_start.size());
            if (good) {
                P->seen.resize(sf.size());
                for (size_t i = 0; i < sf.size(); ++i) {
                    SeenRead& s = P->seen[i];
                    s.read_len = sf[i].read_len; s.first = AlnKey{sf[i].f_st, sf[i].f_en, (char)sf[i].f_strand}; s.last = AlnKey{sf[i].l_st, sf[i].l_en, (char)sf[i].l_strand};
                    s.n_aln = sf[i].n_aln; s.rec = sf[i].rec;
                }
                for (int64_t i = 0; good && i < R; ++i) good = P->rec_seen[(size_t)i] >= 0 && P->rec_seen[(size_t)i] < (int64_t)P->seen.size() && P->meta[(size_t)i * 8 + 7] >= 0 && P->meta[(size_t)i * 8 + 7] < (int64_t)P->motifs.size();
                // offsets ascend and stay inside their arrays (a damaged file must not turn into reads out of bounds later)
                auto ascending = [](const std::vector<int64_t>& v, int64_t last) {
                    if (v.empty() || v[0] != 0 || v.back() != last) return false;
                    for (size_t i = 1; i < v.size(); ++i) if (v[i] < v[i - 1]) return false;
                    return true;
                };
                good = good && ascending(P->read_off, (int64_t)nb) && ascending(P->id_off, (int64_t)P->ids.size()) && ascending(P->u1_ptr, (int64_t)P->u1_start.size());
                for (int64_t i = 0; good && i < R; ++i) {
                    const int64_t b0 = P->read_off[(size_t)i], b1 = P->read_off[(size_t)i + 1], ncols = P->meta[(size_t)i * 8 + 6];
                    for (int64_t u = P->u1_ptr[(size_t)i]; good && u < P->u1_ptr[(size_t)i + 1]; ++u)
                        good = P->u1_start[(size_t)u] >= b0 && P->u1_end[(size_t)u] >= P->u1_start[(size_t)u] && P->u1_end[(size_t)u] <= b1 &&
                               P->u1_col[(size_t)u * 2] >= 0 && P->u1_col[(size_t)u * 2 + 1] >= P->u1_col[(size_t)u * 2] && P->u1_col[(size_t)u * 2 + 1] <= ncols;
                    if (good && P->keep_rows) good = (int64_t)P->rows_r[(size_t)i].size() == ncols && (int64_t)P->rows_m[(size_t)i].size() == ncols;
                }
                for (size_t i = 0; i < P->motifs.size(); ++i) P->motif_ids.emplace(P->motifs[i], (int32_t)i);
            }
        }
        if (good) { *out = P.release(); rc = 0; }
        else set_err(err, errlen, std::string("pack cache does not match its report or is damaged: ") + path);
    } catch (const std::exception& e) { set_err(err, errlen, std::string("cfh_pack_load: ") + e.what()); rc = -12; }
    munmap(m, size);
    return rc;
}

int64_t cfh_n_reads(const cfh_pack* p) { return p->n_reads(); }
int64_t cfh_n_bases(const cfh_pack* p) { return (int64_t)p->bases.size(); }
int64_t cfh_n_seen(const cfh_pack* p) { return (int64_t)p->seen.size(); }
int32_t cfh_non_acgt(const cfh_pack* p) { return p->non_acgt ? 1 : 0; }

// The windows the device path skips (reference scripts/distance_based_kmer_recruitment.py:39-63 counts EVERY window of the raw,
// not upper-cased row as a string): those holding a symbol other than upper-case A, C, G, T.  They are rare (N calls,
// soft-masked stretches), so a dictionary keyed by the window's text is enough.
static int exotic_summary_impl(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, int64_t read_lo, int64_t read_hi, int64_t out[5]);
// (nothing may cross the C ABI: the maps below can throw bad_alloc on a read set full of N)
int cfh_exotic_summary(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, int64_t read_lo, int64_t read_hi, int64_t out[5]) {
    try {
        return exotic_summary_impl(p, k, max_nonuniq, lo, hi, read_lo, read_hi, out);
    } catch (const std::bad_alloc&) {
        return -12;
    } catch (...) {
        return -5;
    }
}
// k-mer (raw window text) -> (reads holding it, reads holding it twice) over reads [read_lo, read_hi)
static void exotic_windows(const cfh_pack* p, int32_t k, int64_t read_lo, int64_t read_hi, std::unordered_map<std::string, std::pair<uint32_t, uint32_t>>& all, int64_t& n_pairs) {
    const int64_t R = (int64_t)p->read_off.size() - 1;
    if (read_lo < 0) read_lo = 0;
    if (read_hi > R) read_hi = R;
    n_pairs = 0;
    std::unordered_map<std::string, uint32_t> mine;
    for (int64_t r = read_lo; r < read_hi; ++r) {
        const int64_t b0 = p->read_off[(size_t)r], len = p->read_off[(size_t)r + 1] - b0;
        if (len < k) continue;
        mine.clear();
        int64_t next_w = 0;                  // windows below this start were taken already
        for (int64_t i = 0; i < len; ++i) {
            const char c = p->bases[(size_t)(b0 + i)];
            if (c == 'A' || c == 'C' || c == 'G' || c == 'T') continue;
            const int64_t w_lo = std::max<int64_t>(next_w, i - k + 1), w_hi = std::min<int64_t>(i, len - k);
            for (int64_t w = w_lo; w <= w_hi; ++w) ++mine[std::string(p->bases.data() + (b0 + w), (size_t)k)];
            if (w_hi + 1 > next_w) next_w = w_hi + 1;
        }
        for (const auto& kv : mine) {
            auto& e = all[kv.first];
            ++e.first;
            if (kv.second > 1) ++e.second;
            ++n_pairs;
        }
    }
}
static int exotic_summary_impl(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, int64_t read_lo, int64_t read_hi, int64_t out[5]) {
    if (!p || !out || k < 1) return -22;
    std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> all;
    int64_t n_pairs = 0;
    exotic_windows(p, k, read_lo, read_hi, all, n_pairs);
    int64_t n_kept = 0, n_rare = 0, n_block = 0;
    for (const auto& kv : all) {
        if (max_nonuniq < 0 || kv.second.second > (uint32_t)max_nonuniq) continue;
        ++n_kept;
        if (kv.second.first < lo || kv.second.first > hi) continue;
        ++n_rare;
        bool lower = false;
        for (char c : kv.first) lower |= (c >= 'a' && c <= 'z');
        if (!lower) ++n_block;               // equals its own upper-case form: could match a window of an upper-cased unit (read_kmer_cloud.py:25)
    }
    out[0] = (int64_t)all.size(); out[1] = n_pairs; out[2] = n_kept; out[3] = n_rare; out[4] = n_block;
    return 0;
}
// The same windows one by one, for a caller that has to ADD the counts of several read shards before it can tell which are rare
// (centroflye_amd/sharded.py): rows of 5 int64 {h1, h2 (two independent 63-bit hashes of the window's text), pres, multi,
// 1 if the window holds no lower-case letter}.  Returns the number of distinct windows; fills at most `cap` rows.
int64_t cfh_exotic_list(const cfh_pack* p, int32_t k, int64_t read_lo, int64_t read_hi, int64_t* rows, int64_t cap) {
    try {
        if (!p || k < 1 || (cap > 0 && !rows)) return -22;
        std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> all;
        int64_t n_pairs = 0;
        exotic_windows(p, k, read_lo, read_hi, all, n_pairs);
        int64_t n = 0;
        for (const auto& kv : all) {
            if (n < cap) {
                uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x9E3779B97F4A7C15ull;
                bool lower = false;
                for (unsigned char c : kv.first) { h1 = (h1 ^ c) * 0x100000001b3ull; h2 = (h2 + c) * 0xff51afd7ed558ccdull; h2 ^= h2 >> 29; lower |= (c >= 'a' && c <= 'z'); }
                int64_t* o = rows + 5 * n;
                o[0] = (int64_t)(h1 >> 1); o[1] = (int64_t)(h2 >> 1); o[2] = kv.second.first; o[3] = kv.second.second; o[4] = lower ? 0 : 1;
            }
            ++n;
        }
        return n;
    } catch (const std::bad_alloc&) {
        return -12;
    } catch (...) {
        return -5;
    }
}
// The windows of cfh_exotic_summary's out[4] themselves: rare (pres in [lo, hi], multi <= max_nonuniq) and free of lower-case letters,
// as text, in ascending order — the k-mers the caller carries beside the 2-bit set (k bytes each, at most cap of them written).
static int64_t exotic_rare_impl(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, char* out, int64_t cap, bool want_lower);
int64_t cfh_exotic_rare(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, char* out, int64_t cap) {
    return exotic_rare_impl(p, k, max_nonuniq, lo, hi, out, cap, false);
}
// The rare windows that DO hold a lower-case letter: the reference's get_rare_kmers returns them too (:66-82 on the raw text of :47-53); they
// can never equal a window of an upper-cased unit (read_kmer_cloud.py:25), so no output depends on them — members of the returned set only.
int64_t cfh_exotic_rare_lower(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, char* out, int64_t cap) {
    return exotic_rare_impl(p, k, max_nonuniq, lo, hi, out, cap, true);
}
static int64_t exotic_rare_impl(const cfh_pack* p, int32_t k, int32_t max_nonuniq, uint32_t lo, uint32_t hi, char* out, int64_t cap, bool want_lower) {
    try {
        if (!p || k < 1 || (cap > 0 && !out)) return -22;
        std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> all;
        int64_t n_pairs = 0;
        exotic_windows(p, k, 0, (int64_t)p->read_off.size() - 1, all, n_pairs);
        std::vector<const std::string*> keep;
        for (const auto& kv : all) {
            if (max_nonuniq < 0 || kv.second.second > (uint32_t)max_nonuniq || kv.second.first < lo || kv.second.first > hi) continue;
            bool lower = false;
            for (char c : kv.first) lower |= (c >= 'a' && c <= 'z');
            if (lower == want_lower) keep.push_back(&kv.first);
        }
        std::sort(keep.begin(), keep.end(), [](const std::string* a, const std::string* b) { return *a < *b; });
        for (int64_t i = 0; i < (int64_t)keep.size() && i < cap; ++i) std::memcpy(out + i * k, keep[(size_t)i]->data(), (size_t)k);
        return (int64_t)keep.size();
    } catch (const std::bad_alloc&) {
        return -12;
    } catch (...) {
        return -5;
    }
}
// Every such window the reference's table keeps (twice in at most max_nonuniq reads; distance_based_kmer_recruitment.py:39-63), as
// text in ascending order with the number of reads that hold it: the keys the 2-bit table of cf_count_kmers cannot have.
int64_t cfh_exotic_kept(const cfh_pack* p, int32_t k, int32_t max_nonuniq, char* out, int64_t* pres, int64_t cap) {
    try {
        if (!p || k < 1 || (cap > 0 && (!out || !pres))) return -22;
        std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> all;
        int64_t n_pairs = 0;
        exotic_windows(p, k, 0, (int64_t)p->read_off.size() - 1, all, n_pairs);
        std::vector<const std::pair<const std::string, std::pair<uint32_t, uint32_t>>*> keep;
        for (const auto& kv : all)
            if (max_nonuniq >= 0 && kv.second.second <= (uint32_t)max_nonuniq) keep.push_back(&kv);
        std::sort(keep.begin(), keep.end(), [](auto* a, auto* b) { return a->first < b->first; });
        for (int64_t i = 0; i < (int64_t)keep.size() && i < cap; ++i) {
            std::memcpy(out + i * k, keep[(size_t)i]->first.data(), (size_t)k);
            pres[i] = (int64_t)keep[(size_t)i]->second.first;
        }
        return (int64_t)keep.size();
    } catch (const std::bad_alloc&) {
        return -12;
    } catch (...) {
        return -5;
    }
}
const uint8_t* cfh_bases(const cfh_pack* p) { return (const uint8_t*)p->bases.data(); }
const int64_t* cfh_read_off(const cfh_pack* p) { return p->read_off.data(); }
const char* cfh_ids(const cfh_pack* p) { return p->ids.data(); }
const int64_t* cfh_id_off(const cfh_pack* p) { return p->id_off.data(); }
const int64_t* cfh_meta(const cfh_pack* p) { return p->meta.data(); }
int32_t cfh_n_motifs(const cfh_pack* p) { return (int32_t)p->motifs.size(); }
const char* cfh_motif(const cfh_pack* p, int32_t id, int64_t* len) {
    if (id < 0 || id >= (int32_t)p->motifs.size()) { if (len) *len = 0; return nullptr; }
    if (len) *len = (int64_t)p->motifs[(size_t)id].size();
    return p->motifs[(size_t)id].data();
}
const char* cfh_discarded(const cfh_pack* p, int64_t* len) {
    if (len) *len = (int64_t)p->discarded.size();
    return p->discarded.data();
}

int cfh_units(cfh_pack* p, int32_t n, int64_t* n_units, const int64_t** unit_ptr,
              const int64_t** unit_start, const int64_t** unit_end, const int64_t** unit_col,
              char* err, int errlen) {
    try {
        if (n < 1) { set_err(err, errlen, "cfh_units: n must be >= 1"); return -22; }
        if (n == 1) {
            *n_units = (int64_t)p->u1_start.size();
            *unit_ptr = p->u1_ptr.data(); *unit_start = p->u1_start.data();
            *unit_end = p->u1_end.data(); *unit_col = p->u1_col.data();
            return 0;
        }
        auto it = p->units_n.find(n);
        if (it == p->units_n.end()) {
            if (!p->keep_rows) { set_err(err, errlen, "cfh_units: n != 1 needs a pack built with keep_rows"); return -22; }
            std::array<std::vector<int64_t>, 4> a;
            a[0].push_back(0);
            std::vector<int64_t> cb, pb;
            const int64_t R = p->n_reads();
            for (int64_t r = 0; r < R; ++r) {
                const std::string& motif = p->motifs[(size_t)p->meta[(size_t)r * 8 + 7]];
                split_units(p->rows_r[(size_t)r], p->rows_m[(size_t)r], motif, n, cb);
                cols_to_pos(p->rows_r[(size_t)r], cb, pb);
                const int64_t base0 = p->read_off[(size_t)r];
                for (size_t b = 0; b + 1 < pb.size(); ++b) {
                    a[1].push_back(base0 + pb[b]); a[2].push_back(base0 + pb[b + 1]);
                    a[3].push_back(cb[b]); a[3].push_back(cb[b + 1]);
                }
                a[0].push_back((int64_t)a[1].size());
            }
            it = p->units_n.emplace(n, std::move(a)).first;
        }
        *n_units = (int64_t)it->second[1].size();
        *unit_ptr = it->second[0].data(); *unit_start = it->second[1].data();
        *unit_end = it->second[2].data(); *unit_col = it->second[3].data();
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errlen, std::string("cfh_units: ") + e.what());
        return -12;
    }
}

int cfh_classify(const cfh_pack* p, int64_t large, int64_t small, uint8_t* cls) {
    const int64_t R = p->n_reads();
    for (int64_t r = 0; r < R; ++r) {
        const SeenRead& s = p->seen[(size_t)p->rec_seen[(size_t)r]];
        const int64_t* m = &p->meta[(size_t)r * 8];
        const int64_t r_len = s.read_len;
        int64_t left, right;
        if (m[4] == 0) { left = s.first.st; right = s.last.en; }
        else { left = r_len - s.last.en; right = r_len - s.first.st; }
        if (left > large && right > r_len - small && right == m[3]) cls[r] = 0;
        else if (right < r_len - large && left < small && left == m[2]) cls[r] = 2;
        else cls[r] = 1;
    }
    return 0;
}

const char* cfh_row(const cfh_pack* p, int64_t r, int32_t which, int64_t* len) {
    if (!p->keep_rows || r < 0 || r >= p->n_reads()) { if (len) *len = 0; return nullptr; }
    const std::string& s = which ? p->rows_m[(size_t)r] : p->rows_r[(size_t)r];
    if (len) *len = (int64_t)s.size();
    return s.data();
}


int cfh_write_kmers(const char* path, const uint64_t* kmers, int64_t n, int32_t k, char* err, int errlen) {
    if (k < 1 || k > 32) { set_err(err, errlen, "cfh_write_kmers: k out of range"); return -22; }
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { set_err(err, errlen, std::string("cannot open ") + path); return -2; }
    int rc = write_ordered(fd, n, (int64_t)1 << 17, [&](int64_t lo, int64_t hi, std::vector<char>& out) {
        out.resize((size_t)(hi - lo) * (size_t)(k + 1));
        char* p = out.data();
        for (int64_t i = lo; i < hi; ++i) { decode_kmer(kmers[i], k, p); p[k] = '\n'; p += k + 1; }
    });
    if (::close(fd) != 0 && rc == 0) rc = -5;
    if (rc) set_err(err, errlen, std::string("write failed: ") + path);
    return rc;
}

int cfh_write_edges(const char* path, int append, const uint64_t* rare, int32_t k,
                    const uint32_t* edges, int64_t n, char* err, int errlen) {
    if (k < 1 || k > 32) { set_err(err, errlen, "cfh_write_edges: k out of range"); return -22; }
    const int fd = ::open(path, O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0666);
    if (fd < 0) { set_err(err, errlen, std::string("cannot open ") + path); return -2; }
    int rc = write_ordered(fd, n, (int64_t)1 << 16, [&](int64_t lo, int64_t hi, std::vector<char>& out) {
        out.resize((size_t)(hi - lo) * (size_t)(2 * k + 24));      // "d a b cnt\n": two k-mers, two numbers of <= 10 digits, 4 separators
        char* p = out.data();
        for (int64_t i = lo; i < hi; ++i) {
            const uint32_t* e = edges + 4 * i;
            p = put_u32(p, e[0]); *p++ = ' ';
            decode_kmer(rare[e[1]], k, p); p += k; *p++ = ' ';
            decode_kmer(rare[e[2]], k, p); p += k; *p++ = ' ';
            p = put_u32(p, e[3]); *p++ = '\n';
        }
        out.resize((size_t)(p - out.data()));
    });
    if (::close(fd) != 0 && rc == 0) rc = -5;
    if (rc) set_err(err, errlen, std::string("write failed: ") + path);
    return rc;
}

// 2-bit code of A, C, G, T; 4 for every other byte
static const struct Code2Table { unsigned char t[256]; Code2Table() { for (int i = 0; i < 256; ++i) t[i] = 4; t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3; } unsigned char operator[](unsigned char c) const { return t[c]; } } kCode2;

int cfh_read_kmers(const char* path, int32_t k, uint64_t* out, int64_t cap, int64_t* n_out,
                   char* err, int errlen) {
    if (k < 1 || k > 32) { set_err(err, errlen, "cfh_read_kmers: k out of range"); return -22; }
    // the file is mapped and scanned line by line with memchr (the caller asks twice — count, then fill: getline made each pass of
    // an 80 MB file of 4 million k-mers half a second)
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { set_err(err, errlen, std::string("cannot open ") + path); return -2; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); set_err(err, errlen, std::string("cannot stat ") + path); return -2; }
    const size_t size = (size_t)st.st_size;
    const char* data = nullptr;
    void* m = MAP_FAILED;
    std::vector<char> held;
    if (size) {
        m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) data = (const char*)m;
        else {      // (not mappable: a pipe, an odd file system) read it
            held.resize(size);
            size_t got = 0;
            while (got < size) { const ssize_t r = read(fd, held.data() + got, size - got); if (r <= 0) break; got += (size_t)r; }
            if (got != size) { close(fd); set_err(err, errlen, std::string("cannot read ") + path); return -5; }
            data = held.data();
        }
    }
    close(fd);
    int64_t cnt = 0;
    const char* p = data; const char* const end = data + size;
    while (p < end) {
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        const char* b = p; const char* e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        while (b < e && is_ws(*b)) ++b;
        while (e > b && is_ws(e[-1])) --e;
        // the reference keeps every stripped line (read_placer.py:23-25); a line that is not a k-long ACGT word has no 2-bit code:
        // skipped here (the k-long ones without a lower-case letter are read as text by kmers.exotic_lines)
        if (e - b != k) continue;
        uint64_t code = 0; unsigned bad = 0;
        for (const char* q = b; q < e; ++q) {      // (table driven, no branch per symbol)
            const unsigned v = kCode2[(unsigned char)*q];
            bad |= v;
            code = (code << 2) | (uint64_t)(v & 3u);
        }
        if (bad & 4u) continue;
        if (out && cnt < cap) out[cnt] = code;
        ++cnt;
    }
    if (m != MAP_FAILED) munmap(m, size);
    *n_out = cnt;
    return 0;
}

// Per-position read-unit export (reference scripts/eltr_polisher.py:53-66 ELTR_Polisher.map_pos2read and :68-97
// export_read_units; the max_pos default of :45-51).  Placed reads come in read_positions.csv order (that is the
// order of the reference's read_placement dict and therefore of every pos2read list and FASTA file).
int cfh_export_read_units(cfh_pack* p, const int64_t* rec, const int64_t* pos, int64_t n_placed, int64_t min_pos,
                          int64_t max_pos, const char* outdir, int n_threads, int64_t* n_positions,
                          int64_t* n_units_written, char* err, int errlen) {
    try {
        const int64_t R = p->n_reads();
        for (int64_t i = 0; i < n_placed; ++i)
            if (rec[i] < 0 || rec[i] >= R || pos[i] < 0) { set_err(err, errlen, "cfh_export_read_units: bad record index or position"); return -22; }
        auto n_units_of = [&](int64_t r) { return p->u1_ptr[(size_t)r + 1] - p->u1_ptr[(size_t)r]; };
        if (max_pos < 0) {   // math.inf: the end of the right-most placed read (:45-51)
            max_pos = 0;
            for (int64_t i = 0; i < n_placed; ++i) max_pos = std::max(max_pos, pos[i] + n_units_of(rec[i]));
        }
        struct Item { int64_t gpos, rec, unit; };
        std::vector<Item> items;
        for (int64_t i = 0; i < n_placed; ++i) {
            const int64_t n = n_units_of(rec[i]), q = pos[i];
            if (q > max_pos) continue;
            int64_t lo = 1, hi = n - 1;                       // inner units only (:61) ...
            if (q == min_pos || q + n == max_pos) { lo = 0; hi = n; }   // ... unless the read touches an end (:58-59)
            for (int64_t u = lo; u < hi; ++u)
                if (min_pos <= q + u && q + u <= max_pos) items.push_back({q + u, rec[i], u});
        }
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.gpos < b.gpos; });
        std::vector<size_t> starts;
        for (size_t i = 0; i < items.size(); ++i) if (i == 0 || items[i].gpos != items[i - 1].gpos) starts.push_back(i);
        starts.push_back(items.size());
        if (mkdir_p(outdir) != 0) { set_err(err, errlen, std::string("cannot create ") + outdir); return -2; }
        const size_t n_pos = starts.size() - 1;
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        auto work = [&]() {
            std::string units, med, dir, key;
            std::vector<std::pair<std::string, size_t>> keys;   // (header, item index)
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= n_pos || failed.load()) break;
                const int64_t gp = items[starts[j]].gpos;
                dir = std::string(outdir) + "/pos_" + std::to_string(gp);
                if (mkdir_p(dir.c_str()) != 0) { failed = 1; break; }
                units.clear(); keys.clear();
                std::vector<int64_t> lens;
                for (size_t i = starts[j]; i < starts[j + 1]; ++i) {
                    const Item& it = items[i];
                    const int64_t u = p->u1_ptr[(size_t)it.rec] + it.unit;
                    const int64_t b0 = p->u1_start[(size_t)u], b1 = p->u1_end[(size_t)u];
                    key = "gen_pos=" + std::to_string(gp) + "|r_id=" +
                          p->ids.substr((size_t)p->id_off[(size_t)it.rec], (size_t)(p->id_off[(size_t)it.rec + 1] - p->id_off[(size_t)it.rec])) +
                          "|r_pos=" + std::to_string(it.unit);
                    units += '>'; units += key; units += '\n';
                    const size_t at = units.size();
                    units.append(p->bases.data() + b0, (size_t)(b1 - b0));
                    for (size_t c = at; c < units.size(); ++c) units[c] = (char)std::toupper((unsigned char)units[c]);
                    units += '\n';
                    lens.push_back(b1 - b0);
                    keys.emplace_back(key, i);
                }
                // statistics.median_high (:82), then the first header in sorted order with that length (:84-89)
                std::vector<int64_t> sl(lens);
                std::sort(sl.begin(), sl.end());
                const int64_t med_len = sl[sl.size() / 2];
                std::sort(keys.begin(), keys.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
                med.clear();
                for (const auto& kv : keys) {
                    const Item& it = items[kv.second];
                    const int64_t u = p->u1_ptr[(size_t)it.rec] + it.unit;
                    const int64_t b0 = p->u1_start[(size_t)u], b1 = p->u1_end[(size_t)u];
                    if (b1 - b0 != med_len) continue;
                    med += '>'; med += kv.first; med += '\n';
                    const size_t at = med.size();
                    med.append(p->bases.data() + b0, (size_t)(b1 - b0));
                    for (size_t c = at; c < med.size(); ++c) med[c] = (char)std::toupper((unsigned char)med[c]);
                    med += '\n';
                    break;
                }
                if (!write_file(dir + "/read_units.fasta", units) || !write_file(dir + "/median_read_unit.fasta", med)) { failed = 1; break; }
            }
        };
        int nt = n_threads > 0 ? n_threads : (int)std::max(1u, std::thread::hardware_concurrency());
        nt = (int)std::min<size_t>((size_t)nt, std::max<size_t>(n_pos, 1));
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
        if (failed.load()) { set_err(err, errlen, std::string("cfh_export_read_units: cannot write under ") + outdir); return -5; }
        if (n_positions) *n_positions = (int64_t)n_pos;
        if (n_units_written) *n_units_written = (int64_t)items.size();
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errlen, std::string("cfh_export_read_units: ") + e.what());
        return -12;
    }
}

}  // extern "C"
