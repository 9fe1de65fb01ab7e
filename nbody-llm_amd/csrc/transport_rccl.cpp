// transport_rccl.cpp -- the exchanges over RCCL (one process per GPU, xGMI): ncclAllGather in place, grouped
// ncclSend / ncclRecv.  Everything is enqueued on the caller's stream; nothing here waits for the device except the
// agreement check of nbody_comm_init (host_all_gather).
#include "transport.h"
#include "../../include/nbody_hip.h"

#include <rccl/rccl.h>

#include <cstring>
#include <vector>

namespace nbody {

namespace {

class RcclTransport final : public Transport {
public:
    RcclTransport(ncclComm_t c, int rank, int world) : comm_(c), rank_(rank), world_(world) {}
    ~RcclTransport() override {
        if (d_blob_) (void)hipFree(d_blob_);
        if (comm_) (void)ncclCommDestroy(comm_);
    }
    const char* name() const override { return "rccl"; }
    int rank() const override { return rank_; }
    int world() const override { return world_; }

    int all_gather(void* buf, size_t bytes, hipStream_t s) override {
        char* base = static_cast<char*>(buf);
        return nccl(ncclAllGather(base + size_t(rank_) * bytes, base, bytes, ncclChar, comm_, s), "ncclAllGather");
    }
    int group_begin() override { return nccl(ncclGroupStart(), "ncclGroupStart"); }
    int send(const void* p, size_t bytes, int peer, hipStream_t s) override { return nccl(ncclSend(p, bytes, ncclChar, peer, comm_, s), "ncclSend"); }
    int recv(void* p, size_t bytes, int peer, hipStream_t s) override { return nccl(ncclRecv(p, bytes, ncclChar, peer, comm_, s), "ncclRecv"); }
    int group_end() override { return nccl(ncclGroupEnd(), "ncclGroupEnd"); }

    int host_all_gather(const void* mine, void* all, size_t bytes) override {
        if (bytes > 256) { err_ = "host_all_gather: blob larger than 256 bytes"; return NBODY_ERR_INVALID; }
        if (!d_blob_ && hipMalloc(&d_blob_, 256 * size_t(world_)) != hipSuccess) { err_ = "hipMalloc (agreement blob)"; return NBODY_ERR_HIP; }
        char* base = static_cast<char*>(d_blob_);
        if (hipMemcpy(base + size_t(rank_) * bytes, mine, bytes, hipMemcpyHostToDevice) != hipSuccess) { err_ = "hipMemcpy (agreement blob)"; return NBODY_ERR_HIP; }
        int rc = nccl(ncclAllGather(base + size_t(rank_) * bytes, base, bytes, ncclChar, comm_, nullptr), "ncclAllGather (agreement)");
        if (rc) return rc;
        if (hipStreamSynchronize(nullptr) != hipSuccess || hipMemcpy(all, base, bytes * size_t(world_), hipMemcpyDeviceToHost) != hipSuccess) {
            err_ = "agreement all-gather: device error";
            return NBODY_ERR_HIP;
        }
        return NBODY_OK;
    }
    int check() override {
        ncclResult_t async = ncclSuccess;
        if (ncclCommGetAsyncError(comm_, &async) != ncclSuccess || async != ncclSuccess) {
            err_ = std::string("RCCL reports an asynchronous error: ") + ncclGetErrorString(async);
            return NBODY_ERR_COMM;
        }
        return NBODY_OK;
    }

private:
    int nccl(ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return NBODY_OK;
        err_ = std::string(what) + ": " + ncclGetErrorString(r);
        return NBODY_ERR_COMM;
    }
    ncclComm_t comm_ = nullptr;
    int rank_, world_;
    void* d_blob_ = nullptr;
};

}  // namespace

int transport_make_id_rccl(void* id128, std::string* err) {
    static_assert(sizeof(ncclUniqueId) <= NBODY_COMM_ID_BYTES, "ncclUniqueId larger than NBODY_COMM_ID_BYTES");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { if (err) *err = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return NBODY_ERR_COMM; }
    std::memset(id128, 0, NBODY_COMM_ID_BYTES);
    std::memcpy(id128, &id, sizeof(id));
    return NBODY_OK;
}

Transport* transport_create_rccl(const void* id128, int rank, int world, std::string* err, int* code) {
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) {
        if (err) *err = std::string("ncclCommInitRank(&h->comm, h->cfg.world_size, id, h->cfg.rank): ") + ncclGetErrorString(r);
        if (code) *code = NBODY_ERR_COMM;
        return nullptr;
    }
    return new RcclTransport(comm, rank, world);
}

}  // namespace nbody
