// vba_launch.h -- host-side launchers of the kernels (one per .hip file), all asynchronous on `s`.
#pragma once
#include <hip/hip_runtime.h>

#include "vba_device.h"

namespace vba {

// vba_obs.hip
void launch_obs_residual(const DevView& V, double* abs_out, hipStream_t s);
void launch_select(const DevView& V, bool with_digit0, hipStream_t s);
void launch_select_warm(const DevView& V, hipStream_t s);
void launch_select_finish(const DevView& V, hipStream_t s);
void launch_obs_accumulate(const DevView& V, hipStream_t s);
void launch_trial(const DevView& V, hipStream_t s);
void launch_clear_hist(const DevView& V, int which, hipStream_t s);
void launch_reset_calls(const DevView& V, hipStream_t s);
void launch_set_counts(const DevView& V, int w, int n, int m, hipStream_t s);
void launch_broadcast_states(const DevView& V, int n, double lamda, hipStream_t s);
void launch_debug_project(const DevView& V, int w, int m, double* est, double* J, double* wt, hipStream_t s);

void launch_sh_clear_miss(const DevView& V, hipStream_t s);
void launch_sh_front(const DevView& V, const double* gathered, int ranks, int slot_len, double* bucket_out, int do_fold, int do_resolve, hipStream_t s);

// vba_dyn.hip
void launch_dynamics(const DevView& V, hipStream_t s);
void launch_assemble(const DevView& V, int fuse_init_solve, hipStream_t s);

// vba_long.hip (both: nothing to do, and nothing launched, for a handle without long edges)
void launch_long_factor(const DevView& V, hipStream_t s);
void launch_long_trial(const DevView& V, hipStream_t s);

// vba_solve.hip
void launch_solve(const DevView& V, int initialize, hipStream_t s);
#ifdef VBA_RESIDENT_STAMPS
void fetch_kstamps(unsigned long long* out);     // diagnostic builds: see vba_solve.hip
void fetch_ostamps(unsigned long long* out);     // ... and vba_obs.hip
#endif
hipError_t configure_solver_device();      // per device, from vba_create
bool solve_forms_blocks(const DevView& V);  // the chunk kernel forms its blocks itself: no k_assemble for this call
void launch_decide(const DevView& V, const double* trial_all, int ranks, hipStream_t s);

// vba_shard.hip
void launch_shard_pack(const DevView& V, double* partial_out, hipStream_t s);
void launch_shard_reduce(const DevView& V, const double* partial_all, int ranks, hipStream_t s, int with_sum = 1);
void launch_shard_trial_sum(const DevView& V, double* trial_local, hipStream_t s);

}  // namespace vba
