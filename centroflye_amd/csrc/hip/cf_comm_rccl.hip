// cf_comm_rccl.hip — cf_comm over RCCL (xGMI inside a node): one process per GPU.
//
// New design (the reference has no collective).  xGMI is point to point (7 links per GPU), so the two
// variable-size exchanges — the owner-bucketed k-mer table (all-to-all) and the rare lists / clouds (all-gather of
// different sizes) — are ncclSend / ncclRecv pairs inside one ncclGroup: every peer's link carries its own message at the
// same time, nothing funnels through a ring.  Messages are cut into rounds of at most round_bytes per pair by the
// loop in cf_comm.h (shared with the CPU suite's file transport); one round = one ncclGroup here.
// RCCL is loaded with dlopen at cf_comm_init: a single-GPU process never maps it.
#include "cf_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <thread>

namespace {

struct rccl_api {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

template <class F>
bool sym(void* so, const char* name, F& f, std::string& err) {
    f = (F)dlsym(so, name);
    if (!f) { err = std::string("librccl: missing symbol ") + name; return false; }
    return true;
}

bool load_rccl(rccl_api& a, std::string& err) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        a.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.so) break;
    }
    if (!a.so) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    return sym(a.so, "ncclGetUniqueId", a.GetUniqueId, err) && sym(a.so, "ncclCommInitRank", a.CommInitRank, err) &&
           sym(a.so, "ncclCommDestroy", a.CommDestroy, err) && sym(a.so, "ncclGroupStart", a.GroupStart, err) &&
           sym(a.so, "ncclGroupEnd", a.GroupEnd, err) && sym(a.so, "ncclSend", a.Send, err) && sym(a.so, "ncclRecv", a.Recv, err) &&
           sym(a.so, "ncclAllGather", a.AllGather, err) && sym(a.so, "ncclAllReduce", a.AllReduce, err) &&
           sym(a.so, "ncclGetErrorString", a.GetErrorString, err);
}

struct rccl_comm : cf_comm {
    rccl_api api;
    ncclComm_t comm = nullptr;

    ~rccl_comm() override {
        if (comm) (void)api.CommDestroy(comm);
        // the library stays mapped: unloading RCCL while HIP is alive is not worth the risk
    }
    int fail(ncclResult_t r, const char* what, std::string& err) {
        err = std::string(what) + ": " + api.GetErrorString(r);
        return -5;
    }
    int sync(hipStream_t s, const char* what, std::string& err) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return -5; }
        return 0;
    }

    int exchange_round(const char* const* sp, const int64_t* ns, char* const* rp, const int64_t* nr, hipStream_t stream, std::string& err) override {
        ncclResult_t r = api.GroupStart();
        if (r != ncclSuccess) return fail(r, "ncclGroupStart", err);
        for (int p = 0; p < world && r == ncclSuccess; ++p) {      // (p == rank: only with self_p2p — a send to and a receive from the calling rank)
            if (ns[p]) r = api.Send(sp[p], (size_t)ns[p], ncclUint8, p, comm, stream);
            if (nr[p] && r == ncclSuccess) r = api.Recv(rp[p], (size_t)nr[p], ncclUint8, p, comm, stream);
        }
        const ncclResult_t r2 = api.GroupEnd();
        if (r != ncclSuccess) return fail(r, "ncclSend/ncclRecv", err);
        if (r2 != ncclSuccess) return fail(r2, "ncclGroupEnd", err);
        return 0;
    }

    int allgather(const void* send, void* recv, int64_t bytes, hipStream_t stream, std::string& err) override {
        const ncclResult_t r = api.AllGather(send, recv, (size_t)bytes, ncclUint8, comm, stream);
        if (r != ncclSuccess) return fail(r, "ncclAllGather", err);
        return sync(stream, "allgather", err);
    }

    int allreduce(void* buf, int64_t count, cf_comm_dtype dt, cf_comm_op op, hipStream_t stream, std::string& err) override {
        const ncclDataType_t t = dt == CF_COMM_U8 ? ncclUint8 : ncclInt64;
        const ncclRedOp_t o = op == CF_COMM_SUM ? ncclSum : ncclMax;
        const int64_t per = std::max<int64_t>(1, dt == CF_COMM_U8 ? round_bytes : round_bytes / 8);
        for (int64_t c0 = 0; c0 < count; c0 += per) {
            char* p = (char*)buf + c0 * (dt == CF_COMM_U8 ? 1 : 8);
            const ncclResult_t r = api.AllReduce(p, p, (size_t)std::min(per, count - c0), t, o, comm, stream);
            if (r != ncclSuccess) return fail(r, "ncclAllReduce", err);
        }
        return sync(stream, "allreduce", err);
    }
};

}  // namespace

// What rank 0 publishes: a magic word, the time it was written (ns since the epoch) and the id.
struct rdv_record { uint64_t magic, written_ns, nonce; ncclUniqueId id; };
static const uint64_t RDV_MAGIC = 0x63666364763034ull;      // "cfcdv04"
// the launch's nonce: every rank of ONE launch sees the same CF_COMM_NONCE — a token the launcher exports per launch (bench.py,
// scripts/distance_based_kmer_recruitment.py with CF_GPUS); centroflye_amd/sharded.py derives one from the common parent's pid and start
// time when the ranks share a parent and leaves it unset otherwise; a C-ABI user who reuses rendezvous names exports a fresh one
// per launch.  0 when it is not set: no nonce check, the age rule alone
static uint64_t launch_nonce() {
    const char* e = std::getenv("CF_COMM_NONCE");
    if (!e || !*e) return 0;
    uint64_t h = 1469598103934665603ull;      // FNV-1a
    for (; *e; ++e) { h ^= (unsigned char)*e; h *= 1099511628211ull; }
    return h ? h : 1;
}
static uint64_t wall_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); }
// when this process started (ns since the epoch), from /proc/self/stat; 0 when it cannot be told
static uint64_t process_start_ns() {
    FILE* f = std::fopen("/proc/self/stat", "r");
    if (!f) return 0;
    char buf[2048];
    const size_t n = std::fread(buf, 1, sizeof buf - 1, f);
    std::fclose(f);
    buf[n] = 0;
    const char* p = std::strrchr(buf, ')');
    if (!p) return 0;
    unsigned long long ticks = 0;
    int field = 2;      // the field after ") " is number 3 (state); starttime is number 22
    for (++p; *p && field < 22; ++p) if (*p == ' ') { ++field; if (field == 22) { ticks = std::strtoull(p + 1, nullptr, 10); break; } }
    f = std::fopen("/proc/uptime", "r");
    double up = 0;
    if (!f || std::fscanf(f, "%lf", &up) != 1) { if (f) std::fclose(f); return 0; }
    std::fclose(f);
    const long hz = sysconf(_SC_CLK_TCK);
    if (hz <= 0 || ticks == 0) return 0;
    const double age = up - (double)ticks / (double)hz;      // seconds since this process started
    return wall_ns() - (uint64_t)(std::max(age, 0.0) * 1e9);
}

// rank 0 writes {magic, time, ncclUniqueId} to `<rendezvous>.<n>` for the n-th communicator of this process (tmp + rename;
// a leftover of that name is removed first), the others poll for it (2 minutes) and ignore a file written more than
// CF_RDV_SLACK_S before they started or carrying another launch's nonce (a leftover of a crashed launch under a reused name:
// the time test alone lets a file of the last two minutes through); rank 0 removes the file once
// ncclCommInitRank has returned (every rank has joined by then).
static const double CF_RDV_SLACK_S = 120.0;
cf_comm* cf_comm_open(int device, int rank, int world, const char* rendezvous, std::string& err) {
    static int generation = 0;      // ranks open their communicators in the same order
    const int gen = generation++;
    if (world < 1 || rank < 0 || rank >= world) { err = "cf_comm_init: bad rank / world"; return nullptr; }
    if (world > 1 && (!rendezvous || !*rendezvous)) { err = "cf_comm_init: a rendezvous path is needed for world > 1"; return nullptr; }
    rccl_comm* c = new (std::nothrow) rccl_comm();
    if (!c) { err = "out of memory"; return nullptr; }
    c->rank = rank; c->world = world;
    if (!load_rccl(c->api, err)) { delete c; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { err = "cf_comm_init: hipSetDevice"; delete c; return nullptr; }
    // RCCL prints a version banner on stdout when a communicator comes up; the caller's stdout belongs to the caller (the
    // benchmark prints ONE line there): while RCCL initialises, file descriptor 1 points at stderr.
    struct StdoutToStderr {
        int saved = -1;
        StdoutToStderr() { std::fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
        ~StdoutToStderr() { std::fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
    } quiet;
    const std::string path = (rendezvous ? std::string(rendezvous) : std::string()) + "." + std::to_string(gen);
    rdv_record rec;
    std::memset(&rec, 0, sizeof rec);
    if (rank == 0) {
        if (world > 1) (void)std::remove(path.c_str());      // a leftover of an earlier launch under the same name
        const ncclResult_t r = c->api.GetUniqueId(&rec.id);
        if (r != ncclSuccess) { c->fail(r, "ncclGetUniqueId", err); delete c; return nullptr; }
        if (world > 1) {
            rec.magic = RDV_MAGIC; rec.written_ns = wall_ns(); rec.nonce = launch_nonce();
            const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
            FILE* f = std::fopen(tmp.c_str(), "wb");
            const bool ok = f && std::fwrite(&rec, sizeof rec, 1, f) == 1;
            if (f) std::fclose(f);
            if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) { err = std::string("cf_comm_init: cannot write ") + path; delete c; return nullptr; }
        }
    } else {
        const uint64_t born = process_start_ns(), nonce = launch_nonce();
        bool got = false, stale = false;
        // CF_RDV_TIMEOUT_S: how long to wait for rank 0's record (default 120 s; tests of the stale-file rule use a short one)
        const char* to_env = std::getenv("CF_RDV_TIMEOUT_S");
        const int polls = std::max(1, (int)((to_env && *to_env ? std::atof(to_env) : 120.0) * 20.0));
        for (int i = 0; i < polls && !got; ++i) {
            FILE* f = std::fopen(path.c_str(), "rb");
            if (f) {
                got = std::fread(&rec, sizeof rec, 1, f) == 1 && rec.magic == RDV_MAGIC;
                std::fclose(f);
                if (got && born && rec.written_ns + (uint64_t)(CF_RDV_SLACK_S * 1e9) < born) { got = false; stale = true; }      // older than this launch: wait for rank 0 to replace it
                if (got && nonce && rec.nonce != nonce) { got = false; stale = true; }      // another launch's
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
        if (!got) { err = std::string("cf_comm_init: timed out waiting for ") + path + (stale ? " (only a stale file of an earlier launch was there)" : ""); delete c; return nullptr; }
    }
    const ncclResult_t r = c->api.CommInitRank(&c->comm, world, rec.id, rank);
    if (rank == 0 && world > 1) (void)std::remove(path.c_str());      // every rank has read it
    if (r != ncclSuccess) { c->fail(r, "ncclCommInitRank", err); c->comm = nullptr; delete c; return nullptr; }
    return c;
}
