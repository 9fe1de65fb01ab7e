/* nbody_hip.h -- C ABI of the MI355X (gfx950) N-body force-and-integrate engine.
 *
 * The reference (alxn3/nbody-llm) has no FFI for this path: its operator interface is the Rust
 * trait `Simulation<F, D, P, I>` (src/shared.rs:80-97) implemented by
 * `BruteForceSimulation` (src/manual/brute_force.rs:28-103) and `BarnesHutSimulation`
 * (src/manual/barnes_hut.rs:205-285).  Every entry point below names the trait method (or
 * reference function) it stands in for; INTEGRATION.md shows the Rust `extern "C"` block and the
 * `impl Simulation` a maintainer would add on the reference side.
 *
 * Conventions
 *   - every call returns int: 0 = NBODY_OK, < 0 = error (nbody_last_error() has the text);
 *     no exception or panic crosses the boundary;
 *   - a handle is used from one thread at a time (the reference calls its simulation from one
 *     thread: src/main.rs:119-122, src/vis.rs:537-553);
 *   - the library owns all device memory; host buffers are caller-owned and only touched
 *     during the call;
 *   - bodies cross the boundary as `PointParticle<F,3>` records (src/shared.rs:151-158,
 *     #[repr(C)]): 10 scalars {pos[3], vel[3], acc[3], mass} -- F = f32: 40 bytes (NbodyConfig.dtype =
 *     NBODY_F32), F = f64: 80 bytes (NBODY_F64; the reference's own driver runs f64, src/main.rs:52-105);
 *   - there is no CPU fallback: without a HIP device nbody_create fails with NBODY_ERR_NO_DEVICE.
 */
#ifndef NBODY_HIP_H
#define NBODY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_ABI_VERSION 4

typedef struct NbodyHandle NbodyHandle;

enum {
    NBODY_OK = 0,
    NBODY_ERR_INVALID = -1,     /* bad argument / call not valid in this state */
    NBODY_ERR_HIP = -2,         /* a HIP runtime call failed */
    NBODY_ERR_CAPACITY = -3,    /* more bodies than the handle was created for */
    NBODY_ERR_TREE_DEPTH = -4,  /* octree deeper than NBODY_MAX_TREE_DEPTH (coincident bodies): the
                                   reference recurses without bound here (barnes_hut.rs:143-183) */
    NBODY_ERR_COMM = -5,        /* RCCL failure */
    NBODY_ERR_NO_DEVICE = -6    /* no usable HIP device */
};

#define NBODY_MAX_TREE_DEPTH 192

/* which reference solver the handle stands in for */
enum { NBODY_BRUTE_FORCE = 0,  /* src/manual/brute_force.rs */
       NBODY_BARNES_HUT = 1 }; /* src/manual/barnes_hut.rs  */

/* arithmetic of the force kernels */
enum { NBODY_MATH_STRICT = 0, /* sqrt, (d*d)*d, g/r^3, no FMA contraction, partners in ascending
                                 index order: the reference's rounding sequence */
       NBODY_MATH_FAST = 1 }; /* v_rsq_f32, FMA, partner range split over waves: <=1e-5 relative */

/* where the Barnes-Hut octree is built */
enum { NBODY_TREE_HOST = 0,    /* host, every step (north_star; barnes_hut.rs:143-183 bit for bit) */
       NBODY_TREE_DEVICE = 1,  /* device: same cells and links, centre-of-mass
                                  sums in a different order (f64 prefix sums), so node counts may differ
                                  by a few parts in 1e4; <= 42 levels (a deeper step is built on the host).
                                  Single-shard handles enqueue their steps without any read-back */
       NBODY_TREE_AUTO = 2 };  /* NBODY_MATH_FAST -> device, NBODY_MATH_STRICT -> host (the bit-exact path);
                                  what the host-side mirrors pass by default */

/* Barnes-Hut leaf semantics */
enum {
    NBODY_LEAF_REFERENCE = 0, /* src/manual/barnes_hut.rs:185-203: a leaf failing the opening test contributes 0 */
    NBODY_LEAF_DIRECT = 1     /* the walk of src/llm/barnes_hut.rs:915-997 on the same tree: a node closer than
                                 r2 < 1e-10 is skipped whole (how a body skips itself), a leaf failing the opening
                                 test is evaluated directly; force = d * (g*mass * (1/sqrt(r2+eps2))^3) */
};

/* the reference's `F: Float` (src/shared.rs:12-44) */
enum { NBODY_F32 = 0,  /* PointParticle<f32,3>: every path of this library */
       NBODY_F64 = 1 }; /* PointParticle<f64,3> (80-byte records).  Brute force: the strict kernel.  Barnes-Hut, NBODY_MATH_STRICT: the
                          reference's nested sums on the host-built tree (NBODY_TREE_HOST, also what AUTO means then): positions,
                          velocities, accelerations and node counts bit-equal to the reference's rounding sequence in f64 (oracle/:
                          the same templated restatement); with NBODY_TREE_DEVICE the tree is built on the device (same cells, centres
                          of mass to the last bits: node counts within 1e-6), ~4x the steps per second.  NBODY_MATH_FAST: one running
                          sum per lane (FMA, 1/sqrt) over a split node range, device build under AUTO: accelerations to 1e-12.
                          Worlds of several ranks: index-block shards (strict results bit-equal to one shard; the tree is built on the
                          host in strict math, by every rank on the device from the gathered positions in fast math) */

/* how the bodies are dealt to the shards of a multi-GPU run */
enum { NBODY_SHARD_INDEX = 0,   /* contiguous index blocks of the vector; positions all-gathered every step (every method) */
       NBODY_SHARD_SPATIAL = 1 }; /* Barnes-Hut, fast math, device build: ownership by Morton-key range (bodies that cross a
                                   boundary migrate), every rank builds only its slice of the tree and receives from each
                                   partner the nodes its own bodies can reach ("halo" / locally essential tree): configs[4] */

typedef struct NbodyConfig {
    uint32_t struct_size;  /* = sizeof(NbodyConfig) */
    int32_t method;        /* NBODY_BRUTE_FORCE | NBODY_BARNES_HUT */
    int32_t math_mode;     /* NBODY_MATH_STRICT | NBODY_MATH_FAST */
    int32_t leaf_mode;     /* NBODY_LEAF_REFERENCE (default) | NBODY_LEAF_DIRECT; Barnes-Hut only */
    int32_t device;        /* HIP device ordinal; -1 = LOCAL_RANK env or 0 */
    int32_t rank;          /* this process's shard, 0 <= rank < world_size */
    int32_t world_size;    /* number of shards (one process per GPU); 1 = single GPU */
    int32_t host_threads;  /* octree-build threads (the reference's `-t`, src/main.rs:34-35); 0 = all */
    uint64_t capacity;     /* max bodies over ALL shards (add_point may grow up to this) */
    int32_t tree_build;    /* NBODY_TREE_HOST | NBODY_TREE_DEVICE | NBODY_TREE_AUTO (Barnes-Hut only) */
    int32_t dtype;         /* NBODY_F32 (0, the default) | NBODY_F64 */
    int32_t shard_mode;    /* NBODY_SHARD_INDEX (0, the default) | NBODY_SHARD_SPATIAL; struct_size may also be the 48 bytes of */
    int32_t reserved;      /* ABI versions <= 2, which end before this field */
} NbodyConfig;

typedef struct NbodyStats {
    uint64_t steps;               /* step_by calls completed */
    uint64_t interactions;        /* bf: sum of n_own*(n_total-1) directed pairs; bh: accepted nodes */
    uint64_t node_visits;         /* bh: opening tests evaluated (0 for bf) */
    uint64_t tree_nodes;          /* bh: nodes in the last tree built */
    uint64_t force_launches;      /* dominant-kernel launches timed since the last nbody_reset_stats */
    uint64_t force_kernel_interactions; /* directed interactions those launches evaluated (bf: the
                                     symmetric kernel leaves ~2 % to a small companion kernel) */
    double force_kernel_ms;       /* sum of their HIP-event durations (needs nbody_set_profiling(h,1)) */
    double tree_build_ms;         /* bh: host wall time in the octree build, summed */
    double tree_copy_ms;          /* bh: host wall time in D2H positions + H2D nodes, summed */
    double exchange_ms;           /* multi-GPU: host wall time blocked in the exchange (0 when async) */
} NbodyStats;

/* ---- lifecycle: Simulation::new / Clone / drop ------------------------------------------- */
/* Simulation::new(points, integrator, bounds) (shared.rs:84): bodies and bounds arrive through
 * nbody_upload / nbody_set_bounds; the integrator is the reference's LeapFrogIntegrator
 * (shared.rs:106-149), the only one it ships.  Settings start at SimulationSettings::default()
 * (shared.rs:69-78): g=1, g_soft=0, dt=1e-3, theta2=0.5. */
int nbody_create(const NbodyConfig* cfg, NbodyHandle** out);
void nbody_destroy(NbodyHandle* h);
/* `Clone` supertrait (shared.rs:80; BH clone drops the tree, barnes_hut.rs:113-135; the visualiser's reset depends on it,
 * vis.rs:217-220).  Any handle; the clone of a sharded handle has no communicator: call nbody_comm_init on it. */
int nbody_clone(const NbodyHandle* h, NbodyHandle** out);

/* ---- state in / out ------------------------------------------------------------------------ */
/* Replaces the body vector (the `points: Vec<P>` argument of Simulation::new).  In a sharded run
 * every rank passes the same full vector; the library keeps its own index block. */
int nbody_upload(NbodyHandle* h, const void* aos, size_t n, size_t stride_bytes);
/* Simulation::get_points (shared.rs:93).  Writes this rank's bodies (all of them when
 * world_size == 1) in vector order; *n_out = how many. */
int nbody_download(NbodyHandle* h, void* aos, size_t cap, size_t stride_bytes, size_t* n_out);
/* get_points().len() of this rank's block. */
int nbody_count(NbodyHandle* h, size_t* n_out);
/* Sum of nbody_count over all ranks as of the last exchange (== nbody_count when world_size == 1). */
int nbody_count_global(NbodyHandle* h, size_t* n_out);
/* Simulation::add_point = Vec::push (brute_force.rs:92-94).  In a sharded world a COLLECTIVE call (every rank passes the
 * same particle): the vector is the concatenation of the ranks' blocks, so the body goes to the end of the last rank's
 * block (NBODY_ERR_CAPACITY when that block is full); with NBODY_SHARD_SPATIAL to the rank that owns its key range, with
 * the next free index as its place in the vector. */
int nbody_add_point(NbodyHandle* h, const void* particle);
/* Simulation::remove_point = Vec::swap_remove (brute_force.rs:96-98): the world's last body takes the place of body `index`
 * (an index into the concatenated vector of all ranks; collective in a sharded world -- the body travels between ranks if
 * they differ).  NBODY_SHARD_SPATIAL: `index` counts the bodies in the order of their indices in the vector, as
 * nbody_download_ids reports them. */
int nbody_remove_point(NbodyHandle* h, size_t index);

/* ---- settings: Simulation::settings / settings_mut (shared.rs:95-96) ----------------------- */
int nbody_set_settings(NbodyHandle* h, float g, float g_soft, float dt, float theta2);
int nbody_get_settings(const NbodyHandle* h, float* g, float* g_soft, float* dt, float* theta2);
/* Bounds::new(center, width) (shared.rs:236-243). */
int nbody_set_bounds(NbodyHandle* h, const float center[3], float width);
/* The same for F = f64 (SimulationSettings<f64>, Bounds<f64, 3>).  Either set works on either kind of handle: the
 * f32 entry points widen exactly, the f64 ones round to f32 on an f32 handle. */
int nbody_set_settings_f64(NbodyHandle* h, double g, double g_soft, double dt, double theta2);
int nbody_get_settings_f64(const NbodyHandle* h, double* g, double* g_soft, double* dt, double* theta2);
int nbody_set_bounds_f64(NbodyHandle* h, const double center[3], double width);

/* ---- stepping ------------------------------------------------------------------------------- */
/* Simulation::init (brute_force.rs:47-50, barnes_hut.rs:229-236): elapsed = 0. */
int nbody_init(NbodyHandle* h);
/* Simulation::step_by(dt) (brute_force.rs:84-90, barnes_hut.rs:265-271): half drift, retain
 * in-bounds bodies, forces, kick + half drift, elapsed += dt.  dt may be negative. */
int nbody_step_by(NbodyHandle* h, float dt);
int nbody_step_by_f64(NbodyHandle* h, double dt);
/* k x Simulation::step() (shared.rs:86-88) with no host synchronisation in between (brute
 * force); returns after enqueueing.  Use nbody_sync before reading a host clock. */
int nbody_steps(NbodyHandle* h, int k);
/* Simulation::update_forces (brute_force.rs:64-82, barnes_hut.rs:250-263). */
int nbody_update_forces(NbodyHandle* h);
/* Simulation::elapsed (shared.rs:94). */
int nbody_elapsed(const NbodyHandle* h, float* out);
int nbody_elapsed_f64(const NbodyHandle* h, double* out);
/* Blocks until everything enqueued on the handle's stream has finished. */
int nbody_sync(NbodyHandle* h);

/* ---- diagnostics (no reference counterpart) -------------------------------------------------- */
int nbody_set_profiling(NbodyHandle* h, int on); /* HIP events around the force-kernel launches: 0 off, 1 every launch, k > 1 every k-th (an
                                                    event pair costs the stream ~11 us; the statistics then cover the bracketed launches) */
int nbody_stats(NbodyHandle* h, NbodyStats* out);
int nbody_reset_stats(NbodyHandle* h);
/* f64 kinetic and potential energy of this rank's view (world_size == 1: the whole system),
 * evaluated on the device: KE = sum 1/2 m v^2, PE = -g sum_{i<j} m_i m_j / sqrt(r^2 + g_soft^2). */
int nbody_energy(NbodyHandle* h, double* kinetic, double* potential);
/* Linearised octree of the last Barnes-Hut force pass: per node {com xyz, mass}, width, skip
 * index (first node after the subtree, depth-first pre-order).  Arrays may be NULL to count. */
int nbody_tree_export(NbodyHandle* h, float* com_mass, float* width, int32_t* skip, size_t cap, size_t* n_nodes);
int nbody_tree_export_f64(NbodyHandle* h, double* com_mass, double* width, int32_t* skip, size_t cap, size_t* n_nodes); /* f64 handles */
/* The cells of that octree for drawing: what the reference's Barnes-Hut Renderable walks (node.bounds.min() / .max() of
 * every node, barnes_hut.rs:322-343).  Per node, pre-order: {min xyz, max xyz} as f32 (the renderer casts to f32 anyway,
 * :331-333; an f64 handle's boxes are computed in double first) and its depth (root = 0).  Arrays may be NULL to count. */
int nbody_tree_export_cells(NbodyHandle* h, float* min_max6, int32_t* depth, size_t cap, size_t* n_nodes);
const char* nbody_last_error(const NbodyHandle* h); /* h may be NULL: last create/clone error */

/* ---- launch-shape and scheme knobs of one handle (no reference counterpart) --------------------------------- */
/* Per handle; the library exports no mutable globals.  Names (csrc/kernels.h struct Tuning): cross_sym, sym_packed,
 * bf_fast_variant, sym_wpb, sym_rounds, sym_reduce_split, cross_slots, cross_ipt, cross_wpb, bh_walk_split, bh_walk_order,
 * bh_reduce_split, tree_max_tie; the environment switches NBODY_CROSS_SYM, NBODY_SYM_PACKED, NBODY_BF_VARIANT, NBODY_SYM_WPB
 * and NBODY_BH_SPLIT preset them at nbody_create.  cross_sym, sym_packed and bf_fast_variant are part of what the ranks of a
 * world agree on at nbody_comm_init and cannot change afterwards.  bh_walk_variant, bh_walk_lds_block, bh_hot_cap,
 * bh_walk_debug and sym_debug select experimental walks and in-kernel stamps that only the tuning build carries
 * (libnbody_hip_tuning.so, `make -C nbody-llm_amd/csrc tuning`): the release library refuses them. */
int nbody_set_tuning(NbodyHandle* h, const char* name, int value);
int nbody_get_tuning(const NbodyHandle* h, const char* name, int* value);
int nbody_is_tuning_build(void);

/* ---- multi-GPU (no reference counterpart) ------------------------ */
#define NBODY_COMM_ID_BYTES 128
/* rank 0 calls nbody_comm_unique_id and ships the bytes to the other ranks out of band; every
 * rank then calls nbody_comm_init on its handle (collective).  The id names the transport of the exchanges:
 *   nbody_comm_unique_id  RCCL (one process per GPU, xGMI) -- or, with NBODY_TRANSPORT=ipc in the environment, the same
 *                         as nbody_comm_local_id;
 *   nbody_comm_local_id   ranks that share ONE device (processes, or threads of one process, on the same host): payloads
 *                         staged through hipIpc-shared windows, flags in host shared memory (csrc/transport_ipc.hip).
 *                         RCCL refuses two ranks on one device; this is how the multi-rank step runs on a one-GPU box.
 * nbody_comm_init also checks that every rank was created alike (ABI version, method, arithmetic, sharding, capacity,
 * exchange scheme): a rank that differs gets NBODY_ERR_COMM there instead of a hang in the first exchange. */
int nbody_comm_unique_id(void* id_bytes);
int nbody_comm_local_id(void* id_bytes);
int nbody_comm_init(NbodyHandle* h, const void* id_bytes);
/* "rccl", "ipc" or "none" */
int nbody_comm_transport(const NbodyHandle* h, char* out, size_t cap);
/* first global index and length of this rank's block at upload time */
int nbody_local_range(const NbodyHandle* h, size_t* first, size_t* count);
/* NBODY_SHARD_SPATIAL: a rank's bodies are not an index block (and change as bodies migrate).  This gives, in the
 * order nbody_download writes the bodies, each one's index in the uploaded vector: scattering all ranks' bodies by it
 * restores the reference's vector order (Vec::retain keeps relative order, so the indices stay ascending there). */
int nbody_download_ids(NbodyHandle* h, int32_t* ids, size_t cap, size_t* n_out);
typedef struct NbodyLetStats {
    uint64_t steps;             /* force passes counted */
    uint64_t bodies_migrated;   /* bodies this rank sent to another rank */
    uint64_t nodes_local;       /* nodes of this rank's slice, summed over the passes */
    uint64_t nodes_global;      /* nodes of the whole tree, summed */
    uint64_t nodes_sent;        /* node records this rank exported, summed over partners and passes */
    uint64_t nodes_received;    /* node records it imported */
    uint64_t bytes_sent;        /* bytes of all four exchanges this rank sent (migrants, end info, spanning-cell tables, nodes) */
    uint64_t bytes_allgather_equivalent; /* what the index-block scheme sends per rank for the same passes: 16 B per own body */
    double phase_ms[5];         /* device time of the five phases between the exchanges, summed over the passes made with
                                 * nbody_set_profiling(h, 1): drift + retain + pick migrants | take them in, keys, sort |
                                 * emit the slice + spanning-cell table | finish + flag + pack the export | walk + kick */
    uint64_t host_syncs;        /* host synchronisations inside the passes (steady state: one per pass, where the export counts are read) */
    uint64_t migrant_respills;  /* passes whose migrants did not fit the sizes their messages were posted with and made the round twice */
    uint64_t node_array_peak_bytes;  /* most bytes of node records this rank held at once: its own slice + what it imported */
    uint64_t node_array_bytes;       /* bytes of every buffer of node records this rank has allocated: its slice, the array the walk runs over
                                        (slice + imports, in global-index order), the export lists, the staged imports -- sized from the
                                        rank's own capacity, not from the world's */
} NbodyLetStats;
int nbody_let_stats(NbodyHandle* h, NbodyLetStats* out);

/* ---- synthetic initial conditions (host side; what src/main.rs:52-89 does for the disc) ----- */
/* Plummer sphere, G = M = 1, Henon units, equal masses 1/n, centre of mass at rest at the origin. */
int nbody_ic_plummer(void* aos, size_t n, size_t stride_bytes, uint64_t seed);
/* The reference's self-gravitating disc: 1 unit-mass star + n disc bodies (src/main.rs:52-89);
 * writes n+1 records. */
int nbody_ic_disc(void* aos, size_t n_disc, size_t stride_bytes, uint64_t seed);
/* the same sets as PointParticle<f64,3> records (80 bytes), unrounded */
int nbody_ic_plummer_f64(void* aos, size_t n, size_t stride_bytes, uint64_t seed);
int nbody_ic_disc_f64(void* aos, size_t n_disc, size_t stride_bytes, uint64_t seed);

/* ---- test hooks: one sharded step with the exchange done by the caller ------------------------- */
/* G handles of one process (rank r of world G, same device) stand in for G GPUs: step_begin on
 * each, import every peer's segment into each (the all-gather of positions), step_forces on each,
 * import every peer's partial sums into each (the ncclSend/ncclRecv round of the symmetric scheme
 * across shards; a no-op for force passes that exchange nothing), step_end on each --
 * nbody_step_by with the RCCL transfers replaced by device-to-device copies. */
int nbody_debug_step_begin(NbodyHandle* h, float dt);
int nbody_debug_import_segment(NbodyHandle* h, NbodyHandle* peer);
int nbody_debug_step_forces(NbodyHandle* h, float dt);
int nbody_debug_import_partials(NbodyHandle* h, NbodyHandle* peer);
int nbody_debug_step_end(NbodyHandle* h, float dt);
/* the same for NBODY_SHARD_SPATIAL handles: phases 0..4 of a step (drift + retain + pick migrants | take migrants in +
 * sort | emit the slice + spanning-cell table | finish the spanning cells + pick and pack the nodes partners need |
 * place imports + walk + kick), and between them the four exchanges as copies from `peer` (which = 0 migrants,
 * 1 end info, 2 spanning-cell tables, 3 nodes).  prune = 0 in nbody_debug_let_set_prune exports every private node. */
int nbody_debug_let_phase(NbodyHandle* h, int phase, float dt);
int nbody_debug_let_exchange(NbodyHandle* h, NbodyHandle* peer, int which);
/* by_work = 1 (default): the redrawn bounds give every rank the same share of the last walk's node visits; 0: of the bodies */
int nbody_debug_let_set_balance(NbodyHandle* h, int by_work);
/* the key-range bounds the next classification will use ([world_size + 1]; redrawn every step at the world's quantiles) */
int nbody_debug_let_bounds(NbodyHandle* h, unsigned long long* out);
int nbody_debug_let_set_prune(NbodyHandle* h, int prune);

/* ---- host-only entry (no device needed): the plan of the symmetric scheme across shards ---------- */
/* Which pairs between shards `rank` evaluates (rows {shard, first chunk, last chunk, first own set,
 * last own set} of 64-body chunks and 64*ipt-body sets) and which ranks send it partial sums. */
int nbody_host_cross_plan(int rank, int world, int seg_cap, int n_own, int* ipt, int* n_sets, int* n_parts, int* parts,
                          int* n_recv, int* recv_from);

/* ---- host-only entry (no device needed): message layout of the spatial step's variable-size rounds ---- */
/* matrix[r * world + q] = records rank r sends to rank q (all-gathered, the same on every rank).  For `rank`: out_at / n_out
 * [world] = record offset (in its packed send buffer, or q * send_stride when !packed_send) and count of the message to
 * each rank; in_at / n_in [world] = offset in its receive buffer and count of the message from each rank; counts are
 * clamped to `clamp` on BOTH sides of every pair (the byte counts of a send and of the receive that meets it are the
 * same expression). */
int nbody_host_exchange_layout(const int* matrix, int world, int rank, long long clamp, int packed_send, size_t send_stride,
                               size_t* out_at, size_t* n_out, size_t* in_at, size_t* n_in, size_t* total_in);

/* ---- host-only entry (no device needed): the launch shapes the library derives from a body count ---- */
/* With the default knobs: the fast Barnes-Hut walk's bodies per lane and node-range segments for `n_bodies` walked bodies at
 * opening angle theta2 (kernels.h walk_plan), and the symmetric brute-force kernel's bodies per lane of a resident set
 * (sym_bodies_per_lane).  What tools and tests read the plan with; out[0..2] = {walk bodies per lane, walk segments, sym
 * bodies per lane}. */
int nbody_host_launch_plan(size_t n_bodies, float theta2, int fast_math, int out[3]);

/* ---- host-only entry (no device needed): the octree build alone ------------------------------ */
/* BarnesHutSimulation::build_tree (barnes_hut.rs:143-183) + linearisation, as the Barnes-Hut step
 * runs it.  pos4 = n records {x,y,z,m}.  Output arrays hold `cap` nodes (com_mass: 4 floats per
 * node) and may be NULL to count; `order` receives the n body ids in depth-first leaf order. */
int nbody_host_build_tree(const float* pos4, size_t n, const float center[3], float width, int threads,
                          float* com_mass, float* node_width, int32_t* skip, int32_t* leaf_body, int32_t* order,
                          size_t cap, size_t* n_nodes);

int nbody_abi_version(void);
int nbody_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_H */
