// cf_comm.h — the transport under the multi-GPU entry points of libcfhip.so (cf_exchange.hip).
//
// The reference has no distributed code; the message shapes follow SURVEY.md §8(e).  One process per
// GPU.  The product build links cf_comm_rccl.hip: RCCL over xGMI (ncclSend / ncclRecv groups, ncclAllGather,
// ncclAllReduce), loaded with dlopen so that a single-GPU run never pays for it.  The CPU test-suite links
// tests/emu/cf_comm_emu.cpp instead (a file-based transport between processes running the host-emulated kernels);
// nothing in the package can select it.
#pragma once
#include "cf_common.h"

enum cf_comm_dtype { CF_COMM_U8 = 0, CF_COMM_I64 = 1 };
enum cf_comm_op { CF_COMM_SUM = 0, CF_COMM_MAX = 1 };

struct cf_comm {
    int rank = 0, world = 1;
    // Messages between a pair of ranks are cut into rounds of at most round_bytes (round 1 finding: a single 1.26 GB
    // all-to-all message lost rows on this RCCL build while 480 MB arrived whole).  Both sides of a pair derive the rounds
    // from the pair's byte count alone, so ranks need no agreement.  Settable (cf_set_param "comm_round_bytes") so that
    // the tests can force many rounds on small data: the loop lives HERE, above the transport, and the CPU suite's file
    // transport runs it too.
    int64_t round_bytes = (int64_t)256 << 20;
    // 1: the message a rank sends to itself goes through the transport's send / receive like every other one (RCCL allows
    // ncclSend / ncclRecv to the calling rank inside a group) instead of a device-to-device copy: a one-GPU box can then run
    // the whole p2p path (cf_set_param "comm_self_p2p", tests).
    bool self_p2p = false;
    virtual ~cf_comm() {}
    // ONE round: for every peer p with ns[p] / nr[p] > 0 send ns[p] bytes from sp[p] and receive nr[p] bytes at rp[p]
    // (device pointers; p == rank only when self_p2p).  Returns when the data has arrived.
    virtual int exchange_round(const char* const* sp, const int64_t* ns, char* const* rp, const int64_t* nr, hipStream_t stream, std::string& err) = 0;
    // For every peer p: send bytes [soff[p], soff[p] + sbytes[p]) of `send` to p and receive rbytes[p] bytes from p at
    // recv + roff[p].  Device pointers; rbytes[p] here equals sbytes[rank] on p.  Returns when the data has arrived.
    int alltoallv(const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff,
                  const int64_t* rbytes, hipStream_t stream, std::string& err) {
        if (sbytes[rank] != rbytes[rank]) { err = "alltoallv: self message sizes differ"; return -22; }
        if (!self_p2p && sbytes[rank]) {
            const hipError_t e = hipMemcpyAsync((char*)recv + roff[rank], (const char*)send + soff[rank], (size_t)sbytes[rank], hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) { err = std::string("alltoallv self copy: ") + hipGetErrorString(e); return -5; }
        }
        const int64_t round = std::max<int64_t>(round_bytes, 16);
        int64_t most = 0;
        for (int p = 0; p < world; ++p) if (p != rank || self_p2p) most = std::max(most, std::max(sbytes[p], rbytes[p]));
        std::vector<const char*> sp((size_t)world);
        std::vector<char*> rp((size_t)world);
        std::vector<int64_t> ns((size_t)world), nr((size_t)world);
        for (int64_t r0 = 0; r0 < most; r0 += round) {
            for (int p = 0; p < world; ++p) {
                const bool on = p != rank || self_p2p;
                ns[(size_t)p] = on ? std::min(std::max<int64_t>(sbytes[p] - r0, 0), round) : 0;
                nr[(size_t)p] = on ? std::min(std::max<int64_t>(rbytes[p] - r0, 0), round) : 0;
                sp[(size_t)p] = (const char*)send + soff[p] + r0;
                rp[(size_t)p] = (char*)recv + roff[p] + r0;
            }
            const int rc = exchange_round(sp.data(), ns.data(), rp.data(), nr.data(), stream, err);
            if (rc) return rc;
        }
        const hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { err = std::string("alltoallv: ") + hipGetErrorString(e); return -5; }
        return 0;
    }
    // recv[p * bytes .. ) = send of rank p (device pointers, the same `bytes` on every rank)
    virtual int allgather(const void* send, void* recv, int64_t bytes, hipStream_t stream, std::string& err) = 0;
    // in place on a device buffer
    virtual int allreduce(void* buf, int64_t count, cf_comm_dtype dt, cf_comm_op op, hipStream_t stream, std::string& err) = 0;
};

// rendezvous: a path every rank can read and write (rank 0 publishes what the others need to join).  It must be unique
// per launch (centroflye_amd/sharded.py derives it from the launcher's pid + start time); the n-th communicator a process
// opens uses `<rendezvous>.<n>`, rank 0 removes a leftover of that name before publishing and removes its own file once
// every rank has joined, and the others ignore files written long before they started.
cf_comm* cf_comm_open(int device, int rank, int world, const char* rendezvous, std::string& err);
