// cf_comm_rccl.hip — cf_comm over RCCL (xGMI inside a node): one process per GPU.
//
// New design (the reference has no collective).  xGMI is point to point (7 links per GPU), so the two
// variable-size exchanges — the owner-bucketed k-mer table (all-to-all) and the rare lists / clouds (all-gather of
// different sizes) — are ncclSend / ncclRecv pairs inside one ncclGroup: every peer's link carries its own message at the
// same time, nothing funnels through a ring.  Messages are cut into rounds of at most CF_COMM_ROUND bytes per pair
// (round 1 finding: a 1.26 GB all_to_all lost rows on this RCCL build while 480 MB arrived whole; both sides of a pair
// derive the rounds from the pair's byte count alone, so no agreement between ranks is needed).
// RCCL is loaded with dlopen at cf_comm_init: a single-GPU process never maps it.
#include "cf_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <thread>

static const int64_t CF_COMM_ROUND = (int64_t)256 << 20;

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

    int alltoallv(const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff,
                  const int64_t* rbytes, hipStream_t stream, std::string& err) override {
        if (sbytes[rank] != rbytes[rank]) { err = "alltoallv: self message sizes differ"; return -22; }
        if (sbytes[rank]) {
            const hipError_t e = hipMemcpyAsync((char*)recv + roff[rank], (const char*)send + soff[rank], (size_t)sbytes[rank], hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) { err = std::string("alltoallv self copy: ") + hipGetErrorString(e); return -5; }
        }
        int64_t most = 0;
        for (int p = 0; p < world; ++p) if (p != rank) most = std::max(most, std::max(sbytes[p], rbytes[p]));
        for (int64_t r0 = 0; r0 < most; r0 += CF_COMM_ROUND) {
            ncclResult_t r = api.GroupStart();
            if (r != ncclSuccess) return fail(r, "ncclGroupStart", err);
            for (int p = 0; p < world && r == ncclSuccess; ++p) {
                if (p == rank) continue;
                const int64_t ns = std::min(std::max<int64_t>(sbytes[p] - r0, 0), CF_COMM_ROUND);
                const int64_t nr = std::min(std::max<int64_t>(rbytes[p] - r0, 0), CF_COMM_ROUND);
                if (ns) r = api.Send((const char*)send + soff[p] + r0, (size_t)ns, ncclUint8, p, comm, stream);
                if (nr && r == ncclSuccess) r = api.Recv((char*)recv + roff[p] + r0, (size_t)nr, ncclUint8, p, comm, stream);
            }
            const ncclResult_t r2 = api.GroupEnd();
            if (r != ncclSuccess) return fail(r, "ncclSend/ncclRecv", err);
            if (r2 != ncclSuccess) return fail(r2, "ncclGroupEnd", err);
        }
        return sync(stream, "alltoallv", err);
    }

    int allgather(const void* send, void* recv, int64_t bytes, hipStream_t stream, std::string& err) override {
        const ncclResult_t r = api.AllGather(send, recv, (size_t)bytes, ncclUint8, comm, stream);
        if (r != ncclSuccess) return fail(r, "ncclAllGather", err);
        return sync(stream, "allgather", err);
    }

    int allreduce(void* buf, int64_t count, cf_comm_dtype dt, cf_comm_op op, hipStream_t stream, std::string& err) override {
        const ncclDataType_t t = dt == CF_COMM_U8 ? ncclUint8 : ncclInt64;
        const ncclRedOp_t o = op == CF_COMM_SUM ? ncclSum : ncclMax;
        const int64_t per = dt == CF_COMM_U8 ? CF_COMM_ROUND : CF_COMM_ROUND / 8;
        for (int64_t c0 = 0; c0 < count; c0 += per) {
            char* p = (char*)buf + c0 * (dt == CF_COMM_U8 ? 1 : 8);
            const ncclResult_t r = api.AllReduce(p, p, (size_t)std::min(per, count - c0), t, o, comm, stream);
            if (r != ncclSuccess) return fail(r, "ncclAllReduce", err);
        }
        return sync(stream, "allreduce", err);
    }
};

}  // namespace

// rank 0 writes the ncclUniqueId to `rendezvous` (tmp + rename), the others poll for it (2 minutes)
cf_comm* cf_comm_open(int device, int rank, int world, const char* rendezvous, std::string& err) {
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
    ncclUniqueId id;
    std::memset(&id, 0, sizeof id);
    if (rank == 0) {
        const ncclResult_t r = c->api.GetUniqueId(&id);
        if (r != ncclSuccess) { c->fail(r, "ncclGetUniqueId", err); delete c; return nullptr; }
        if (world > 1) {
            const std::string tmp = std::string(rendezvous) + ".tmp" + std::to_string((long)getpid());
            FILE* f = std::fopen(tmp.c_str(), "wb");
            const bool ok = f && std::fwrite(&id, sizeof id, 1, f) == 1;
            if (f) std::fclose(f);
            if (!ok || std::rename(tmp.c_str(), rendezvous) != 0) { err = std::string("cf_comm_init: cannot write ") + rendezvous; delete c; return nullptr; }
        }
    } else {
        bool got = false;
        for (int i = 0; i < 2400 && !got; ++i) {
            FILE* f = std::fopen(rendezvous, "rb");
            if (f) { got = std::fread(&id, sizeof id, 1, f) == 1; std::fclose(f); }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
        if (!got) { err = std::string("cf_comm_init: timed out waiting for ") + rendezvous; delete c; return nullptr; }
    }
    const ncclResult_t r = c->api.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { c->fail(r, "ncclCommInitRank", err); c->comm = nullptr; delete c; return nullptr; }
    return c;
}
