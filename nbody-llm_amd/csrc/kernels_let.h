// kernels_let.h -- launchers and wire structs of the spatial-shard (halo exchange) Barnes-Hut path; see kernels_let.hip.
#pragma once
#include "kernels.h"

namespace nbody {
namespace let {

constexpr int kLevels = 21;     // levels of the device build's keys
constexpr int kMaxRanks = 16;

enum { kFlagDeep = 1, kFlagNodeCapLocal = 2, kFlagBigGroup = 4 /* kernels_tree.hip: > 4096 bodies share 16 levels */, kFlagCapacity = 8, kFlagNodeCap = 16,
       kFlagMigSpill = 32 };   // more migrants between some pair of ranks than the size its message was posted with (the
                               // step's migrant round is then made again with the exact sizes: nbody_let.cpp)

struct Migrant {                // a body on its way to the rank that owns its key range (64 bytes)
    float4 pos, vel, acc;
    int id, pad[3];
};

constexpr int kBoxDigits = 2;   // a rank describes where its bodies are by one bounding box per child cell, this many levels
constexpr int kBoxes = 1 << (3 * kBoxDigits);   // below the deepest cell that holds all of them (a key range is not a box: a
                                // range that straddles the boundary of two big cells has a bounding box that spans both)

struct EndInfo {                // what every rank tells the others after its sort (all-gathered)
    unsigned long long first_key, last_key;   // of its sorted bodies
    int n_bodies, pad;
    long long weight;                         // sum of its bodies' weights (body_weight of the last walk's visit counts)
    float lo[3], hi[3];                       // bounding box of all its bodies (quick reject)
    float box_lo[kBoxes][3], box_hi[kBoxes][3];   // of its bodies in each child cell; lo > hi: none there
};

struct Contrib {                // [r][d]: what a rank adds to the cell of depth d on rank r's LAST body's path (40 bytes)
    double m, mx, my, mz;       // sums over its bodies inside that cell
    int cnt;                    // how many of them
    int base_after;             // r != self: index in its slice of the first node behind them (its node count if none);
                                // r == self: index in its slice of the cell itself, -1 if the cell is not its own
};

struct RoundB {                 // what every rank tells the others after its scans (all-gathered)
    int n_nodes, flags, pad[2];
    unsigned long long new_bound[kMaxRanks];   // [j]: the key at global sorted position j N / G if this rank holds it, else 0:
                                               // next step's ownership bounds (the world's G-quantiles, redrawn every step)
    Contrib c[kMaxRanks][kLevels];
};

struct LetRecord {              // an exported node (32 bytes): the record with its GLOBAL skip link; b.z (NodeB::hot, which the
    float4 a, b;                // plain walk does not read) carries its global index (its place in the world's pre-order)
};

void launch_classify(hipStream_t s, const Shard& sh, int n_upper, const float center[3], float width, const unsigned long long* bounds,
                     int G, int me, unsigned char* dest_of, Migrant* send, int* send_count, int* send_off, int* cursor, bool after_drift);
void launch_append(hipStream_t s, const Shard& sh, const Migrant* recv, int n_in, int G, int* flags, int* new_count, int* send_count);
// The migrant round with message sizes fixed BEFORE the counts are known to the host (pred[a * G + b] = records the message
// from rank a to rank b is posted with; the same matrix on every rank): k_let_spec raises kFlagMigSpill on every rank alike
// if some pair has more, launch_slot_migrants copies the packed emigrants into one slot per destination, and
// launch_append_slots takes the immigrants out of the receive slots (actual counts from the all-gathered matrix) unless
// the flag is up.  slots_in_upper = sum of the predicted sizes of this rank's incoming messages.
void launch_spec_check(hipStream_t s, const int* matrix, const int* pred, int G, int* flags);
void launch_slot_migrants(hipStream_t s, const Migrant* packed, int n_packed_upper, const int* send_off, const int* pred, int G, int me, Migrant* slots);
void launch_append_slots(hipStream_t s, const Shard& sh, const Migrant* slots_in, int slots_in_upper, const int* matrix, const int* pred, int G, int me,
                         int* flags, int* new_count, int* send_count);
// everything the host wants to know once per step, gathered into one block for one copy:
// report[0..G*G) export counts | [G*G..2 G*G) migrant counts | offsets[G+1] | tree_info[3] | flags[4]
void launch_report(hipStream_t s, const int* let_matrix, const int* mig_matrix, const int* offsets, const int* tree_info, const int* flags, int G,
                   int* report);
inline int report_ints(int G) { return 2 * G * G + (G + 1) + 3 + 4; }
void launch_ends(hipStream_t s, const Shard& sh, int n_upper, const unsigned long long* sorted_keys, const int* sorted_ids, int* box_ord,
                 unsigned long long* weight_sum, EndInfo* mine);
void launch_edges(hipStream_t s, const EndInfo* ends, int G, int me, int* edge);
void launch_contrib(hipStream_t s, const Shard& sh, const TreeDevWork& w, const int* info, const EndInfo* ends, const int* edge, int G, int me,
                    RoundB* mine, bool balance_by_work, const int* own_flags);
void launch_offsets(hipStream_t s, const RoundB* rb, int G, int* offsets, int* out_flags, unsigned long long* bounds);
void launch_finalize(hipStream_t s, const RoundB* rb, const EndInfo* ends, int G, int me, float width, float4* slice, const int* offsets, int* top_index,
                     float4* top_nodes);
// the export lists, each in node order, one after the other in `send` (list r at list_first[r], let_count[r] long);
// node_mask [local_cap], block_n [pack_blocks(local_cap) * kMaxRanks] hold what launch_pack needs to write them again into a
// bigger buffer (records beyond send_cap are dropped: the counts tell)
void launch_flags_and_pack(hipStream_t s, int local_cap, const int* info, const int* edge, const float4* slice, const float4* top_nodes,
                           const int* offsets, const int* top_index, const EndInfo* ends, int G, int me, float theta2, const int* parent,
                           const unsigned char* depth, unsigned int* upper_ok, int2* link, unsigned int* node_mask, int* block_n, int* let_count,
                           int* list_first, LetRecord* send, size_t send_cap, bool prune);
void launch_pack(hipStream_t s, int local_cap, const int* info, const float4* slice, const float4* top_nodes, const int* offsets, const int* top_index,
                 const unsigned char* depth, const unsigned int* node_mask, const int* block_first, const int* list_first, int G, int me, LetRecord* send,
                 size_t send_cap);
size_t pack_blocks(int local_cap);
// the array the walk runs over: the rank's slice and the staged imports (in_n[q] records from rank q, one rank after the
// other) in global-index order, links translated to positions; split = {0, nodes held}.  staged_upper >= the imports' total.
void launch_assemble(hipStream_t s, const float4* slice, int local_cap, const LetRecord* staged, int staged_upper, const int* in_n, const int* info,
                     const int* offsets, const int* top_index, const float4* top_nodes, int G, int me, void* layout, int* split, float4* held,
                     int n_split, int* seg_first /* n_split > 1: [n_split + 1] where the walk's segments start (global cuts) */);
size_t layout_bytes();

}  // namespace let
}  // namespace nbody
