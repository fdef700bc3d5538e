// cf_comm_emu.cpp — file-based transport behind cf_comm for the HOST-EMULATED build of the kernels.
// TEST INFRASTRUCTURE ONLY (tests/emu/build_emu.sh links it in place of centroflye_amd/csrc/hip/cf_comm_rccl.hip): the
// world-size-2 CPU test runs two processes of the emulated library; "device" memory is host memory there, and every
// message is a file `m<seq>_<from>_<to>` in the rendezvous DIRECTORY (written as tmp + rename, removed by its reader).
// Both ends of a pair derive the same messages from the pair's byte counts, so per-pair sequence numbers pair them up.
#include "cf_comm.h"

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <thread>

namespace {

struct emu_comm : cf_comm {
    std::string dir;
    std::vector<uint64_t> sseq, rseq;      // messages sent to / received from each peer so far: both ends of a pair count alike

    std::string name(uint64_t s, int from, int to) const { return dir + "/m" + std::to_string(s) + "_" + std::to_string(from) + "_" + std::to_string(to); }

    int put(const std::string& fn, const void* p, int64_t n, std::string& err) {
        const std::string tmp = fn + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) { err = "emu comm: cannot create " + tmp; return -5; }
        const bool ok = n == 0 || std::fwrite(p, 1, (size_t)n, f) == (size_t)n;
        std::fclose(f);
        if (!ok || std::rename(tmp.c_str(), fn.c_str()) != 0) { err = "emu comm: cannot write " + fn; return -5; }
        return 0;
    }
    int get(const std::string& fn, void* p, int64_t n, std::string& err) {
        for (int i = 0; i < 24000; ++i) {     // 2 minutes
            struct stat st;
            if (stat(fn.c_str(), &st) == 0) {
                if ((int64_t)st.st_size != n) { err = "emu comm: " + fn + " has " + std::to_string((long long)st.st_size) + " bytes, expected " + std::to_string((long long)n); return -5; }
                FILE* f = std::fopen(fn.c_str(), "rb");
                if (!f) { err = "emu comm: cannot open " + fn; return -5; }
                const bool ok = n == 0 || std::fread(p, 1, (size_t)n, f) == (size_t)n;
                std::fclose(f);
                if (!ok) { err = "emu comm: short read of " + fn; return -5; }
                std::remove(fn.c_str());
                return 0;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
        err = "emu comm: timed out waiting for " + fn;
        return -5;
    }

    // one round of the shared loop in cf_comm.h (p == rank appears only with self_p2p: the file goes to ourselves)
    int exchange_round(const char* const* sp, const int64_t* ns, char* const* rp, const int64_t* nr, hipStream_t, std::string& err) override {
        if (sseq.empty()) { sseq.assign((size_t)world, 0); rseq.assign((size_t)world, 0); }
        for (int p = 0; p < world; ++p)
            if (ns[p]) { int rc = put(name(sseq[(size_t)p]++, rank, p), sp[p], ns[p], err); if (rc) return rc; }
        for (int p = 0; p < world; ++p)
            if (nr[p]) { int rc = get(name(rseq[(size_t)p]++, p, rank), rp[p], nr[p], err); if (rc) return rc; }
        return 0;
    }
    int allgather(const void* send, void* recv, int64_t bytes, hipStream_t st, std::string& err) override {
        std::vector<int64_t> soff((size_t)world, 0), sb((size_t)world, bytes), roff((size_t)world), rb((size_t)world, bytes);
        for (int p = 0; p < world; ++p) roff[(size_t)p] = (int64_t)p * bytes;
        return alltoallv(send, soff.data(), sb.data(), recv, roff.data(), rb.data(), st, err);
    }
    int allreduce(void* buf, int64_t count, cf_comm_dtype dt, cf_comm_op op, hipStream_t st, std::string& err) override {
        const int64_t bytes = count * (dt == CF_COMM_U8 ? 1 : 8);
        std::vector<char> all((size_t)(bytes * world));
        int rc = allgather(buf, all.data(), bytes, st, err);
        if (rc) return rc;
        for (int64_t i = 0; i < count; ++i) {
            if (dt == CF_COMM_U8) {
                unsigned v = op == CF_COMM_SUM ? 0u : 0u;
                for (int p = 0; p < world; ++p) { const unsigned x = (unsigned char)all[(size_t)(p * bytes + i)]; v = op == CF_COMM_SUM ? v + x : std::max(v, x); }
                ((unsigned char*)buf)[i] = (unsigned char)v;
            } else {
                int64_t v = op == CF_COMM_SUM ? 0 : INT64_MIN;
                for (int p = 0; p < world; ++p) { int64_t x; std::memcpy(&x, &all[(size_t)(p * bytes + i * 8)], 8); v = op == CF_COMM_SUM ? v + x : std::max(v, x); }
                ((int64_t*)buf)[i] = v;
            }
        }
        return 0;
    }
};

}  // namespace

cf_comm* cf_comm_open(int, int rank, int world, const char* rendezvous, std::string& err) {
    if (world < 1 || rank < 0 || rank >= world) { err = "cf_comm_init: bad rank / world"; return nullptr; }
    struct stat st;
    if (world > 1 && rendezvous) (void)mkdir(rendezvous, 0700);      // every rank may try; EEXIST is fine
    if (world > 1 && (!rendezvous || stat(rendezvous, &st) != 0 || !S_ISDIR(st.st_mode))) { err = "emu comm: the rendezvous must be a directory"; return nullptr; }
    emu_comm* c = new emu_comm();
    c->rank = rank; c->world = world; c->dir = rendezvous ? rendezvous : "";
    return c;
}
