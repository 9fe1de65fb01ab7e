// transport.h -- what moves bytes between the ranks of a sharded run (internal to libnbody_hip.so).
//
// The reference has no counterpart (its only parallelism is rayon inside one process, src/manual/barnes_hut.rs:160-170,
// 258); the exchanges are this build's own.  The force paths issue exactly three kinds
// of operation, all stream-ordered and all enqueued without waiting for the device:
//   all_gather  every rank's part of a buffer ends up on every rank (positions, counts, the small tables of the halo
//               exchange), in place: rank r's part already sits at buf + r * bytes
//   send/recv   grouped point-to-point messages (partial sums back to their owners, migrants, tree nodes)
// Two implementations, chosen by the id nbody_comm_init receives:
//   rccl  ncclAllGather / ncclSend / ncclRecv: one process per GPU over xGMI (transport_rccl.cpp)
//   ipc   ranks that share ONE device (processes, or threads of one process): staged through hipIpc-shared windows,
//         "data ready" / "window free" flags in host shared memory set and polled by one-wave kernels
//         (transport_ipc.hip).  It exists so that the production step -- the same nbody_step_by / nbody_steps, the same
//         counts, offsets, streams and events -- can run with G real ranks on a one-GPU box.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace nbody {

enum { kTransportRccl = 0, kTransportIpc = 1 };

class Transport {
public:
    virtual ~Transport() = default;
    virtual const char* name() const = 0;
    virtual int rank() const = 0;
    virtual int world() const = 0;
    // 0 or an NBODY_ERR_* code; error() has the text
    virtual int all_gather(void* buf, size_t bytes_per_rank, hipStream_t s) = 0;
    virtual int group_begin() = 0;
    virtual int send(const void* p, size_t bytes, int peer, hipStream_t s) = 0;
    virtual int recv(void* p, size_t bytes, int peer, hipStream_t s) = 0;
    virtual int group_end() = 0;
    // blocking host-side all-gather of one small blob per rank (<= 256 bytes): agreement checks at nbody_comm_init
    virtual int host_all_gather(const void* mine, void* all, size_t bytes) = 0;
    // anything that went wrong behind the host's back (a device-side wait that ran out of time, a peer that gave up)
    virtual int check() = 0;
    const std::string& error() const { return err_; }

protected:
    std::string err_;
};

// rank 0 makes the id (128 bytes, shipped to the other ranks out of band); every rank then creates its end
int transport_make_id(int kind, void* id128, std::string* err);
int transport_id_kind(const void* id128);   // kTransportRccl | kTransportIpc
Transport* transport_create(const void* id128, int rank, int world, int device, std::string* err, int* code);

Transport* transport_create_rccl(const void* id128, int rank, int world, std::string* err, int* code);
Transport* transport_create_ipc(const void* id128, int rank, int world, int device, std::string* err, int* code);
int transport_make_id_rccl(void* id128, std::string* err);
int transport_make_id_ipc(void* id128, std::string* err);

}  // namespace nbody
