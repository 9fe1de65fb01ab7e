// nbody_let.h -- Barnes-Hut over spatial shards with a halo exchange (NbodyConfig.shard_mode = NBODY_SHARD_SPATIAL):
// internal entry points behind include/nbody_hip.h; kernels in kernels_let.hip.
#pragma once
#include "nbody_handle.h"

namespace nbody {
namespace let {

struct State;

int create(NbodyHandle* h);                 // after the single-segment Shard of the handle exists
void destroy(NbodyHandle* h);
int upload(NbodyHandle* h, const void* aos, size_t n, size_t stride);   // every rank passes the full vector
int download_ids(NbodyHandle* h, int32_t* ids, size_t cap, size_t* n_out);
int count_global(NbodyHandle* h, size_t* n_out);
int clone_state(NbodyHandle* src, NbodyHandle* dst);    // ids, ownership bounds (the communicator is re-attached by nbody_comm_init)
int add_point(NbodyHandle* h, const void* particle);    // collective
int remove_point(NbodyHandle* h, size_t index);         // collective
int step(NbodyHandle* h, float dt);         // one step with the RCCL exchanges
int update_forces(NbodyHandle* h);
int stats(NbodyHandle* h, NbodyLetStats* out);
int reset_stats(NbodyHandle* h);
// offsets and record counts of the variable-size rounds, from the all-gathered count matrix (host arithmetic)
size_t exchange_layout(const int* m, int G, int me, long long clamp, bool packed_send, size_t send_stride, size_t* out_at, size_t* n_out,
                       size_t* in_at, size_t* n_in);
int check_flags(NbodyHandle* h);            // turns the device's sticky flags into an error code (synchronises)
// one-process emulation of G ranks (tests): the phases between the exchanges, and the exchanges as copies
int debug_phase(NbodyHandle* h, int phase, float dt);
int debug_exchange(NbodyHandle* h, NbodyHandle* peer, int which);

}  // namespace let
}  // namespace nbody
