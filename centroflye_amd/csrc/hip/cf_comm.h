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
    virtual ~cf_comm() {}
    // For every peer p: send bytes [soff[p], soff[p] + sbytes[p]) of `send` to p and receive rbytes[p] bytes from p at
    // recv + roff[p].  Device pointers; rbytes[p] here equals sbytes[rank] on p.  Returns when the data has arrived.
    virtual int alltoallv(const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff,
                          const int64_t* rbytes, hipStream_t stream, std::string& err) = 0;
    // recv[p * bytes .. ) = send of rank p (device pointers, the same `bytes` on every rank)
    virtual int allgather(const void* send, void* recv, int64_t bytes, hipStream_t stream, std::string& err) = 0;
    // in place on a device buffer
    virtual int allreduce(void* buf, int64_t count, cf_comm_dtype dt, cf_comm_op op, hipStream_t stream, std::string& err) = 0;
};

// rendezvous: a path every rank can read and write (rank 0 publishes what the others need to join)
cf_comm* cf_comm_open(int device, int rank, int world, const char* rendezvous, std::string& err);
