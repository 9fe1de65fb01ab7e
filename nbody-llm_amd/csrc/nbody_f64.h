// nbody_f64.h -- F = f64 handles (NbodyConfig.dtype == NBODY_F64): internal entry points behind include/nbody_hip.h.
#pragma once
#include "nbody_handle.h"

namespace nbody64 {

int create(NbodyHandle* h);                                   // after the stream (and, for Barnes-Hut, pool + counters) exist
void destroy(NbodyHandle* h);
int clone_state(NbodyHandle* src, NbodyHandle* dst);
int upload(NbodyHandle* h, const void* aos, size_t n, size_t stride);
int download(NbodyHandle* h, void* aos, size_t cap, size_t stride, size_t* n_out);
int count(NbodyHandle* h, size_t* n_out);
int count_global(NbodyHandle* h, size_t* n_out);
int add_point(NbodyHandle* h, const void* particle);
int remove_point(NbodyHandle* h, size_t index);
int set_settings(NbodyHandle* h, double g, double g_soft, double dt, double theta2);
int get_settings(const NbodyHandle* h, double* g, double* g_soft, double* dt, double* theta2);
int set_bounds(NbodyHandle* h, const double center[3], double width);
void get_bounds(const NbodyHandle* h, double center[3], double* width);
int init(NbodyHandle* h);
int step_by(NbodyHandle* h, double dt);
int steps(NbodyHandle* h, int k);
int update_forces(NbodyHandle* h);
double elapsed(const NbodyHandle* h);
int stats(NbodyHandle* h, NbodyStats* out);
int reset_stats(NbodyHandle* h);
int energy(NbodyHandle* h, double* kinetic, double* potential);
int tree_export(NbodyHandle* h, double* com_mass, double* width, int32_t* skip, size_t cap, size_t* n_nodes);

}  // namespace nbody64
