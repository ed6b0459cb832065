/*
 * vinsat_ba.h -- C ABI of libvinsat_ba.so: the VINSat bundle-adjustment iteration on MI355X.
 *
 * The reference exposes this path as ONE Python function (there is no FFI layer in it):
 *
 *   BA(iter, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics,
 *      confidences, Sigma, V, lamda_init, poses_gt_eci, initialize=False)
 *        -> (states_new, velocities, lamda_init, last_hessian)
 *   reference: estimation/BA/BA_filtering.py:4-98, called from estimation/od_pipe.py:1038,1040.
 *
 * The entry points below are what a binding for that function needs: the arguments that stay constant
 * over the 20 iterations of a window are uploaded once (observations, per-pose constants), and
 * vba_iterate() is one call of BA().  Plain pointers and sizes only; all host buffers are caller
 * owned, row-major, fp64 unless stated; every call returns 0 on success or a VBA_E* code.
 *
 * A handle owns its device memory and (unless vba_set_stream is used) its HIP stream.  Calls are
 * synchronous on return unless documented otherwise.  One handle per host thread.
 *
 * A handle can hold W independent windows ("batched windows": the reference's outer loop over
 * sequences, estimation/od_pipe.py:1069-1077); every kernel launch then covers all of them.
 */
#ifndef VINSAT_BA_H
#define VINSAT_BA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vba_context* vba_handle;

enum {
    VBA_OK = 0,
    VBA_EINVAL = 1,     /* bad argument (null pointer, size out of range, unsorted input ...) */
    VBA_ENODEV = 2,     /* no usable HIP device */
    VBA_EHIP = 3,       /* a HIP runtime call failed; see vba_last_error() */
    VBA_ESTATE = 4,     /* call order violated (e.g. iterate before upload) */
    VBA_ENOMEM = 5
};

/* bits of the `flags` word returned by vba_iterate (reference behaviour is unchanged by them) */
enum {
    VBA_FLAG_LAMBDA_EXHAUSTED = 1u, /* "lamda too large": no trial improved, last trial kept (BA_filtering.py:75-77) */
    VBA_FLAG_NONFINITE = 2u,        /* NaN/Inf met in the solve or the residuals */
    VBA_FLAG_ZERO_PIVOT = 4u,       /* a diagonal block was numerically singular */
    VBA_FLAG_HOST_CHANGED = 1u << 30 /* vba_iterate_resident only: a watched host buffer (vba_set_host_watch) no longer holds the bytes
                                       that were uploaded -- the result was computed from the OLD window: upload again and repeat the call */
};

/* selectors for vba_debug_fetch: intermediates of the LAST vba_iterate call, window 0 unless stated */
enum {
    VBA_DBG_EST = 0,        /* [m,2]   reprojection at the input states (BA_utils.py:30-43), input order */
    VBA_DBG_WEIGHT = 1,     /* [m]     final robust weight w_k (BA_filtering.py:24-25), input order */
    VBA_DBG_H = 2,          /* [n,6,6] per-pose sum of w J^T J, scaled as the reference (BA_filtering.py:32-36) */
    VBA_DBG_B = 3,          /* [n,6]   per-pose sum of w J^T r (BA_filtering.py:44) */
    VBA_DBG_PHI = 4,        /* [n,6,6] d x_hat_i / d x_i of the orbit propagation (BA_utils.py:73-87) */
    VBA_DBG_RPRED = 5,      /* [n-1,7] dynamics residuals (BA_utils.py:476) */
    VBA_DBG_QGRAD = 6,      /* [n,3]   attitude gradient, rotation slots (BA_utils.py:520) */
    VBA_DBG_HQ = 7,         /* [n,3,3,3] attitude Newton blocks (sub, diag, super) rot-rot (BA_utils.py:521-523) */
    VBA_DBG_BANDS = 8,      /* [n,3,9,9] undamped block-tridiagonal system (sub, diag, super) (BA_filtering.py:54) */
    VBA_DBG_RHS = 9,        /* [n,9]   JTr (BA_filtering.py:48) */
    VBA_DBG_DPOSE = 10,     /* [n,9]   solution of the last LM trial (BA_filtering.py:55) */
    VBA_DBG_SCALARS = 11,   /* [8]     c_obs, w_max, init_residual, last trial residual, lamda32 of last trial, sigma, alpha, n_trials */
    VBA_DBG_JG = 12         /* [m,2,6] reprojection Jacobian (BA_utils.py:44-48), input order */
};

/* library / device */
int vba_version(void);
/* 1 if the library was built with -DVBA_VARIANTS (make VARIANTS=1 -> libvinsat_ba_variants.so): the solver variants that were
 * measured slower and are kept for comparison only -- three windows per wavefront (vba_set_solver -3), one window per wavefront
 * forming its own blocks, one cyclic-reduction level in front (VBA_OPT_FUSION bit 4), the solve as one grid of waiting blocks
 * (bits 5, 6).  The default build does not carry them: those settings return VBA_EINVAL. */
int vba_has_variants(void);
const char* vba_last_error(void);
int vba_device_count(int* count);

/* Create a context on HIP device `device` able to hold `windows` windows of at most n_max poses and
 * m_max observations each.  Replaces nothing in the reference (it allocates per call). */
int vba_create(int device, int windows, int n_max, int64_t m_max, vba_handle* out);
/* The same with the kernel set named by the caller.  A handle runs one of two kernel sets, fixed at creation because the
 * device memory differs (bin buckets of the carried keys exist in latency mode only):
 *   mode 1  latency mode: few kernels per call (select inside the accumulation, accept test folded into the next call, the
 *           chunk elimination forms its own blocks, the trial kernel forms the step), 64 lanes per pose -- what a handle
 *           whose windows cannot fill the chip by themselves wants;
 *   mode 0  bandwidth mode: streaming kernels with few registers and many windows per launch;
 *   mode -1 (what vba_create passes) chooses by window count and window size -- latency mode up to 38 (50 000 / m_max)^0.7 windows
 *           (exponent 0.46 for windows of more than 50 000 rows; at most 192),
 *           the crossover of the measured sweeps over W = 1 .. 4096 windows of 5 000 / 20 000 / 50 000 / 200 000 rows (bench.py
 *           "batched_sweep").
 * Both modes give the same results to rounding (and the same bits for equal lanes per pose / solver settings). */
int vba_create_mode(int device, int windows, int n_max, int64_t m_max, int mode, vba_handle* out);
/* *mode receives the kernel set of the handle (0 / 1 as above), *chunk the solver partition in use (0: sequential walk). */
int vba_get_mode(vba_handle h, int* mode, int* chunk);
int vba_destroy(vba_handle h);

/* external != 0: run all work of this handle on the caller's HIP stream `hip_stream` (a hipStream_t; NULL is the
 * legacy default stream), e.g. the stream RCCL collectives are ordered against.  external == 0: back to the
 * handle's own stream. */
int vba_set_stream(vba_handle h, void* hip_stream, int external);

/* Choice of the block-tridiagonal solve: chunk = 0 one wavefront walks the whole pose chain (work optimal, used
 * when many windows are batched); chunk in [2,60] cuts the chain into chunks of that many poses that are
 * eliminated in parallel plus a reduced system over the separators; chunk = -1 restores the default
 * (0 for more than 1023 windows -- the walk is a latency chain of ~2.2 ms at 500 poses that only so many windows amortise, see the
 * sweep in DESIGN.md section 3 --; bandwidth-mode handles: chunks of 12 poses (fewer separators: throughput); otherwise chunks of 8 poses
 * (n_max / 3 for chains shorter than 24; more beyond 1032 poses; several latency-mode windows: up to 12 where that keeps the chunk
 * elimination of all windows in one round of the chip, 1024 two-wave blocks) and the reduced
 * system by cyclic reduction, see vba_set_solver2; two levels of ~n^(1/3) beyond 7700 poses).  With chunk = 0 a
 * wavefront walks one window; chunk = -3 makes three windows of equal pose count share a wavefront (no faster on
 * MI355X at any batch size measured, kept for comparison), chunk = -2 forbids it.  All variants agree to rounding. */
int vba_set_solver(vba_handle h, int chunk);
/* Explicit partition: chunks of `chunk` poses; the reduced system over their separators is
 *   chunk2 = 0       walked by one wavefront,
 *   chunk2 in [2,60] cut again into chunks of `chunk2` separators (two levels),
 *   chunk2 = -1      solved by block cyclic reduction (log2 of the separator count levels): inside one workgroup with
 *                    the whole reduced system in LDS up to 23 separators; from 24 on the first level runs as its own
 *                    kernel on one CU per separator pair and the workgroup continues with the halved system
 *                    (at most 128 separators, i.e. chunk >= ceil(n_max / 129)). */
int vba_set_solver2(vba_handle h, int chunk, int chunk2);

/* Orbit integrator of the dynamics factor.  0 (default): one-second RK4 steps, the reference's CPU branch `predict`
 * (BA_utils.py:73-87) -- the parity target.  A gap of more than 64 s between two poses (time_idx of vba_upload_window; a knot
 * every 1000 s makes every later-pass window hold such gaps, od_pipe.py:213-221) is propagated PARALLEL IN TIME: the same
 * one-second steps in ~sqrt(gap) chunks side by side, their start states found by a parareal iteration whose fixed point is the
 * serial chain (stopped when every chunk's end state meets the next chunk's start state to 2^-48 relative), the transition matrix
 * as the ordered product of the chunks' matrices -- equal to the serial chain to rounding (csrc/vba_long.hip).  1: the coarse schedule of `propagate_orbit_dynamics_skip`
 * (BA_utils.py:52-71: steps of 100 s plus one remainder step) that the reference itself switches to when it sees a
 * GPU (`predict_gpu`, BA_filtering.py:16-17); results differ from mode 0 by the integration error. */
int vba_set_integrator(vba_handle h, int hop100);



/* ---- settings a caller of BA() never needs: ONE entry point.  Every option keeps results identical to rounding (most: bit for
 * bit); defaults are chosen from the handle's geometry by measured sweeps (docs/NOTEBOOK.md).  A value an
 * option does not take returns VBA_EINVAL.  The tests and the A/B tools are the users of these. */
enum {
    VBA_OPT_ACCUMULATE_LANES = 1,   /* lanes per pose of the accumulation (4, 8, 16, 32, 64; 0 = from the geometry, ~12 rows per lane):
                                       the shape of its reduction tree -- equal settings give equal bits */
    VBA_OPT_TRIAL_TILES = 2,        /* tiles of 256 rows per observation block of the plain latency-mode trial kernel (0 = automatic, 1, 2,
                                       4, 8): one pass of bin reservations per block; block sums stay per tile: bit-identical for every value */
    VBA_OPT_KEY_CARRY = 3,          /* 1 (default): an accepted trial leaves the next call's |r| keys, histogram and sum |r| behind (it is
                                       evaluated at exactly the states the next call starts from, BA_filtering.py:61-66 / :12-21); 0: every
                                       call recomputes them (same bits) */
    VBA_OPT_WARM_SELECT = 4,        /* exact lower median (torch.median, BA_filtering.py:23) of carried keys: 1 (default) from the warm bins
                                       the producing trial binned them into (latency mode: the bucket of one bin, ranked inside the
                                       accumulation, which also evaluates the accept test of the call in front; else one compaction pass), a
                                       rank outside the bins repeats the select with the exact digits (vba_warm_select_misses); 0: exact digit
                                       passes and a decide launch per call; 2: every warm select misses (test); 3: the select stays a kernel */
    VBA_OPT_WARM_SHIFT = 5,         /* log2 of a warm bin's width in bit patterns (52 = a binade; defaults 44 / 43 latency, 46 bandwidth
                                       mode); the median is exact for every width */
    VBA_OPT_BUCKET_CAP = 6,         /* test knob: keys a bin bucket holds (8 .. allocated; 0 = default): a fuller bin is a miss */
    VBA_OPT_FUSION = 7,             /* bit mask: 0 the trial kernel forms the step of each pose (latency mode), 1 the chunk elimination
                                       forms the blocks of its chunk (latency mode), 2 the sequential walk forms its blocks (bandwidth mode),
                                       3 full-phase assembly in uniform passes; bits 4 .. 6 (one cyclic-reduction level in front; the solve as
                                       one grid of waiting blocks: measured slower) exist in the comparison build only (vba_has_variants).
                                       Default 15, latency-mode handles of many rows 14 / 12 */
    VBA_OPT_CHUNK_WAVES = 8,        /* partitioned solve: 2 (default) = a chunk is eliminated from both ends by two waves, 1 = one wave
                                       left to right (another rounding, ~1e-9 on the step; each deterministic) */
    VBA_OPT_PIVOTING = 9,           /* 0 (default): 9x9 blocks eliminated without row exchanges, every pivot checked against the diagonal
                                       entry it started from, a failed check repeats that solve with pivoting (vba_solver_fallbacks);
                                       1: pivot from the start */
    VBA_OPT_PIPELINE = 10,          /* 1 (default, one-window handles): vba_iterate_resident returns call k once its accept test is known
                                       and has enqueued call k + 1 speculatively (a wrong guess is dropped with the carried keys); bits of
                                       vba_step; vba_debug_fetch refuses after a pipelined call */
    VBA_OPT_SCHEDULE_GRAPH = 11,    /* 1 (default, latency mode): vba_run_schedule replays the launches of its first pass as a hipGraph while
                                       nothing that goes into them has changed (per-call kernel arguments compared exactly); a capture that
                                       fails switches it off for the handle and the pass is launched kernel by kernel; also VBA_NO_GRAPH */
    VBA_OPT_CHAIN_PROFILE = 12      /* 1: vba_run_schedule records HIP events at the class boundaries of every call (~1 us each, no graph
                                       replay meanwhile): vba_chain_profile */
};
int vba_set_option(vba_handle h, int option, int value);

/* BA_reg (BA_filtering.py:100-210): the BA call with a propagated-covariance prior per pose.
 * vba_upload_prior: states_prior [n,10] (arguments states_prior / velocity_prior of BA_reg: positions and the velocity
 * columns are used), hessian_state [n,6,6] (argument hessian_state_t: information matrix over [position, velocity]).
 * The reference's hessian_rot_t has no effect on its result: the rotation term of prior_gpu (BA_utils.py:626) is
 * quat_coeff (1 - |q_p^T G(q_p) H_rot G(q)^T q|) with G(q)^T q = 0 identically, i.e. a constant -- it is reproduced
 * as that constant (1 per pose in the initial residual mean, 100 per pose in every trial mean, as the reference
 * passes its coefficients) and takes no matrix.
 * vba_set_prior(h, 1): the following vba_step / vba_iterate / vba_run_schedule calls are BA_reg calls (the prior is
 * inactive in landmark-only calls, BA_utils.py:609-612, but still counts in the residual means); 0 (default): BA. */
int vba_upload_prior(vba_handle h, int window, int n, const double* states_prior, const double* hessian_state);
int vba_set_prior(vba_handle h, int on);


/* Carried-key selects that fell outside the warm bins and were repeated with the exact digits (VBA_OPT_WARM_SELECT). */
int vba_warm_select_misses(vba_handle h, int* count);



/* Solves repeated with row pivoting after a failed pivot check (VBA_OPT_PIVOTING). */
int vba_solver_fallbacks(vba_handle h, int* count);

/* Observation rows of window `window`: landmarks_xyz [m,3] (ECI km), landmarks (uv) [m,2] px,
 * confidences [m], ii [m] pose index of each row (BA arguments landmarks_xyz, landmarks, confidences, ii:
 * BA_filtering.py:4; ii is int64 as at BA_filtering.py:35).  n is the number of poses the indices refer to.
 * Uploading with a pose count different from the window's current one starts a new window: the other upload
 * (pose constants / observations) and the states must follow before the next step.
 * The rows are packed into pinned memory and sent with one asynchronous copy on the handle's stream (ordered in
 * front of the kernels that read them); the caller's arrays may be reused as soon as the call returns. */
int vba_upload_observations(vba_handle h, int window, int n, int64_t m, const double* landmarks_xyz,
                            const double* landmarks_uv, const double* confidences, const int64_t* ii);

/* Per-pose constants of window `window`: intrinsics [n,4] = fx,fy,cx,cy; cumrot_last [n,4] =
 * imu_meas[0,:,-1,6:10], the attitude increment over the gap that follows each pose (the only part of
 * imu_meas the reference's CPU path reads, BA_utils.py:295); time_idx [n] seconds, strictly increasing. */
int vba_upload_window(vba_handle h, int window, int n, const double* intrinsics, const double* cumrot_last,
                      const int64_t* time_idx);

/* Device-resident state of a window: [n,10] = p(3) q(4, scalar last) v(3); lamda is the LM damping.
 * window == -1 in vba_set_states gives every window the same states (equal pose counts required). */
int vba_set_states(vba_handle h, int window, const double* states, double lamda);
int vba_get_states(vba_handle h, int window, double* states, double* lamda, double* last_hessian /*[81] or NULL*/,
                   int* n_trials /*or NULL*/, unsigned* flags /*or NULL*/);

/* The same for every window of the handle at once -- the batch dimension of BA()'s `states [bsz, n, 10]` (BA_filtering.py:14):
 * states [W][n_max][10] (rows beyond a window's pose count are ignored / left untouched), lamda [W], last_hessian [W][81],
 * n_trials [W], flags [W] (each of the last three may be NULL).  One copy each way instead of W. */
int vba_set_states_all(vba_handle h, const double* states, const double* lamda);
int vba_get_states_all(vba_handle h, double* states, double* lamda, double* last_hessian, int* n_trials, unsigned* flags);

/* One BA() call (BA_filtering.py:4-98) on EVERY window of the handle, states and lamda device resident:
 * states <- states_new, lamda <- lamda_out.  `iter` selects alpha and Sigma (BA_filtering.py:22,26);
 * `initialize` != 0 is the landmark-only phase (BA_utils.py:463-466).  Synchronous. */
int vba_step(vba_handle h, int iter, int initialize);

/* ncalls consecutive BA() calls on every window -- the driver's loop `for iter in range(20): BA(iter, ...)`
 * (od_pipe.py:1036-1040) -- issued as one host call: iters[c] / inits[c] are the `iter` and `initialize` arguments
 * of call c.  The calls are chained on the device (a window that needs further LM trials in some call stalls there
 * and is finished by the host before the rest is re-issued); the results are bit-identical to ncalls vba_step
 * calls.  *trials_total (optional) receives the number of LM trials issued.  Synchronous. */
int vba_run_schedule(vba_handle h, int ncalls, const int* iters, const int* inits, int* trials_total);

/* Convenience: set_states(window 0) + step + get_states(window 0); the exact shape of one BA() call. */
int vba_iterate(vba_handle h, int iter, int initialize, double lamda_in, const double* states_in,
                double* states_out, double* lamda_out, double* last_hessian, int* n_trials, unsigned* flags);

/* The same call when the states argument IS the result of the previous vba_iterate / vba_iterate_resident /
 * vba_run_schedule on this handle (the driver loop `states = BA(iter, states, ...)`, od_pipe.py:1036-1040) and lamda_in
 * the lamda it returned: nothing is uploaded, the device-resident states and damping are used, and the call starts
 * from the keys the last accepted trial left behind (VBA_OPT_KEY_CARRY).  Same bits as vba_iterate. */
int vba_iterate_resident(vba_handle h, int iter, int initialize, double* states_out, double* lamda_out,
                         double* last_hessian, int* n_trials, unsigned* flags);

/* vba_iterate as the FIRST call of a driver loop whose following calls are vba_iterate_resident (a new window, or states the caller
 * changed): the states go up and the call is served like a resident one -- returned as soon as its accept test is known, the next
 * call enqueued behind it (VBA_OPT_PIPELINE); the watched host buffers are compared as in a resident call (VBA_FLAG_HOST_CHANGED).
 * Same bits as vba_iterate.  A caller that replaces the states before every call
 * should use vba_iterate: there the speculated call would be waited for and dropped each time. */
int vba_iterate_open(vba_handle h, int iter, int initialize, double lamda_in, const double* states_in, double* states_out,
                     double* lamda_out, double* last_hessian, int* n_trials, unsigned* flags);

/* Host buffers of the caller whose content was uploaded (e.g. the ndarray arguments ii / time_idx of BA()) and that the caller
 * might edit in place: every vba_iterate_resident compares `live` with the reference `copy` (bytes each, both must stay valid;
 * 8 slots -- one per array argument of BA() --, live == NULL clears one) while the device works -- from 64 kB of watched bytes on a helper
 * thread of the handle, beside the enqueue of the speculated call -- and reports a difference as VBA_FLAG_HOST_CHANGED. */
int vba_set_host_watch(vba_handle h, int slot, const void* live, const void* copy, int64_t bytes);
/* Speculated calls of the pipelined driver loop that were used / dropped (VBA_OPT_PIPELINE). */
int vba_pipeline_stats(vba_handle h, int* hits, int* discards);

/* Copy an intermediate of the last step of `window` to host memory; *count receives the number of
 * doubles written (capacity is checked). */
int vba_debug_fetch(vba_handle h, int window, int what, double* out, int64_t capacity, int64_t* count);

/* Timing of the last vba_step measured with HIP events on the handle's stream, milliseconds. */
int vba_last_step_ms(vba_handle h, float* ms);

/* Same work as vba_step, with every kernel class bracketed by HIP events on the handle's stream (first LM
 * trial only).  ms[VBA_NKERNELS] receives the durations in the order of the VBA_K_* enum; a class that did
 * not run (dynamics in the landmark-only phase) reports 0.  VBA_K_BEGIN has no kernel any more: it is the
 * interval between two back-to-back event records, i.e. the measurement overhead contained in every class.
 * In the landmark-only phase the first trial's solve rides in the assemble class (k_assemble<true>). */
enum { VBA_K_BEGIN = 0, VBA_K_RESIDUAL, VBA_K_SELECT, VBA_K_ACCUMULATE, VBA_K_DYNAMICS, VBA_K_ASSEMBLE, VBA_K_SOLVE,
       VBA_K_TRIAL, VBA_K_DECIDE, VBA_NKERNELS };
int vba_step_profiled(vba_handle h, int iter, int initialize, float* ms);

/* Graphs captured / replays of vba_run_schedule so far (VBA_OPT_SCHEDULE_GRAPH). */
int vba_schedule_graph_stats(vba_handle h, int* captures, int* replays);

/* Class times of the CHAINED schedule (VBA_OPT_CHAIN_PROFILE on): ms[3] = summed milliseconds of the classes {accumulate (with
 * the select / accept test folded into it, or launched in front of it), solve, trial}, launches[3] = how many intervals each sum
 * holds (a landmark-only call whose step the trial kernel forms has no solve interval); reset != 0 clears the sums. */
int vba_chain_profile(vba_handle h, double* ms, int64_t* launches, int reset);

/* ---- observation-sharded multi-GPU operation (one rank per GPU, window 0 only) -------------------------
 * Each rank uploads its slice of the observation rows and the full per-pose constants.  One BA() call is
 * the sequence   stage1 -> all-gather -> stage2 -> all-gather -> stage3 -> all-gather -> stage4 [-> stage3 ...]
 * where the exchanges are done by the caller (RCCL through torch.distributed) on DEVICE buffers; all
 * stage calls are asynchronous on the handle's stream except stage4.
 */
/* number of doubles each rank contributes to the second exchange: per-pose blocks + gradient + scalars */
int64_t vba_sh_partial_count(int n);
/* stage 1: residuals of the local rows at the resident states; writes 2*m_local |r| values to d_abs_local.
 * m_total = number of observation rows over all ranks (fixes the median rank and the residual means). */
int vba_sh_stage1(vba_handle h, int iter, int initialize, int64_t m_total, double* d_abs_local);
/* stage 2: exact lower median over the gathered |r| of all ranks, robust weights and the local per-pose
 * accumulation; writes vba_sh_partial_count(n) doubles to d_partial_local.  d_abs_all holds count_all values of
 * which 2*m_total are keys; slots that pad unequal shards must hold +inf (they sort above every key). */
int vba_sh_stage2(vba_handle h, const double* d_abs_all, int64_t count_all, double* d_partial_local);
/* stage 3: reduce the R gathered partials in rank order, build + solve the system (every rank redundantly),
 * retract, and evaluate the local part of the trial residual into d_trial_local[0..1].  d_partial_all == NULL
 * runs another LM trial (next damping) on the system already built. */
int vba_sh_stage3(vba_handle h, const double* d_partial_all, int ranks, double* d_trial_local);
/* stage 4: accept test on the gathered trial sums (2 doubles per rank); *done = 1 when the LM loop ended
 * (states/lamda updated).  Synchronous. */
int vba_sh_stage4(vba_handle h, const double* d_trial_all, int ranks, int* done);
/* The same protocol with the three exchanges issued by the LIBRARY: RCCL all-gathers on the handle's stream between the
 * stage kernels -- one host call per BA() call and one host synchronisation per LM trial instead of four stage calls and
 * three collectives dispatched by the caller.  RCCL is resolved at run time from `rccl_path` (a process that also uses
 * torch.distributed names the copy it has loaded already; /opt/rocm/lib/librccl.so otherwise); libvinsat_ba.so itself does
 * not link it.
 *   vba_sh_unique_id: rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means;
 *   vba_sh_comm_init: every rank of the window joins (ncclCommInitRank: collective, returns when all have);
 *   vba_sh_call:      one BA() call (BA_filtering.py:4-98) on the sharded window (protocol below); this rank's rows
 *                     (vba_upload_observations) must number at most ceil(m_total / ranks);
 *   vba_sh_comm_destroy: leaves the communicator (vba_destroy does it as well).
 * Protocol of the library-issued form (vba_sh_set_protocol; default 1):
 *   1  carried keys.  An accepted trial is evaluated at the states the next call starts from, so the trial kernel of every rank
 *      leaves the |r| keys of ITS rows behind in per-bin buckets, with their warm histogram (VBA_OPT_WARM_SELECT) and block sums,
 *      written straight into the exchange buffer.  Per call: all-gather A [histogram | block sums] (~12 kB per rank at 500 / 50k;
 *      the next call's first kernel evaluates the accept test on it and resolves the bin of the global median from the summed
 *      histograms), all-gather B [this rank's bucket of that bin] (<= 8 kB), all-gather C [per-pose normal equations, written
 *      by the accumulation in place].  No pass over all keys, no glue launches; the calls of a schedule are chained on the device
 *      and every rank enqueues the same collectives whether a call runs or skips, one host synchronisation per schedule.  A call
 *      whose select misses (median outside the binned range, a bucket overflowed) is repeated with protocol 0's front; a trial
 *      that is not cleanly accepted is finished by the ordinary LM loop with the trial sums gathered per round.  All decisions
 *      are taken on gathered data: every rank takes the same.  Needs a latency-mode handle with its default kernel fusion,
 *      created for exactly m_max = ceil(m_total / ranks) rows on every rank (the exchange buffers follow the handle's geometry);
 *      any other handle takes protocol 0.
 *   0  the round-3 protocol: every call recomputes its keys and gathers all of them (16 B per observation). */
int vba_sh_unique_id(const char* rccl_path, void* id128);
int vba_sh_comm_init(vba_handle h, const char* rccl_path, const void* id128, int nranks, int rank);
int vba_sh_call(vba_handle h, int iter, int initialize, int64_t m_total, int* n_trials);
/* ncalls consecutive BA() calls on the sharded window as one host call (the driver's loop od_pipe.py:1036-1040), chained on the
 * device in protocol 1; *trials_total: LM rounds issued.  vba_sh_call is the schedule of one call. */
int vba_sh_run_schedule(vba_handle h, int ncalls, const int* iters, const int* inits, int64_t m_total, int* trials_total);
int vba_sh_set_protocol(vba_handle h, int carried_keys);
/* bytes this rank contributes to the first exchange of a call; calls repeated after a missed select; calls finished by the LM loop */
int vba_sh_stats(vba_handle h, int64_t* bytes_first_exchange, int64_t* fallbacks_miss, int64_t* fallbacks_lm);
int vba_sh_comm_destroy(vba_handle h);

/* ---- host-side helpers of the driver around BA() (streaming_version, od_pipe.py:911-1062).  No device is
 * involved and no handle is needed: the serial recurrences of the driver's data preparation, which as interpreted loops cost more
 * than the BA calls they sit between.
 *   vba_host_orbit_chain:   `steps` one-second RK4 steps of the J2 orbit dynamics from x0 = [p (km), v (km/s)] -- the dead
 *                           reckoning across the gap between two batches (propagate_dynamics_init, BA_utils.py:114-129; RK4 :901-912;
 *                           driver od_pipe.py:1011, 1052); out[k] = the state after k + 1 steps.
 *   vba_host_quat_chain:    running Hamilton product (scalar last, BA_utils.py:992-1000) out[k] = q0 (x) r[0] (x) ... (x) r[k]
 *                           (q0 NULL: out[k] = r[0] (x) ... (x) r[k]) -- the attitude of the same dead reckoning
 *                           (propagate_rotation_dynamics_init, BA_utils.py:105-112).  Every product and sum is rounded on its own in
 *                           the order of the reference's expression: the bits of the array code.
 *   vba_host_gap_rotations: the attitude increment accumulated over the gap after each pose, cum[i] = r[t_i] (x) ... (x)
 *                           r[t_{i+1} - 1] with r[N,4] the per-second increments exp(dt * omega) and cum[T - 1] the identity
 *                           (precompute_cum_rotations, BA_utils.py:278-288, of which only [..., -1] is read, :295; driver
 *                           od_pipe.py:945-961).  Same rounding rule. */
int vba_host_orbit_chain(const double* x0 /*[6]*/, int steps, double* out /*[steps,6]*/);
int vba_host_quat_chain(const double* q0 /*[4] or NULL*/, const double* r /*[K,4]*/, int K, double* out /*[K,4]*/);
int vba_host_gap_rotations(const double* r /*[N,4]*/, int64_t N, const int64_t* time_idx /*[T]*/, int T, double* cum /*[T,4]*/);

/* ---- the per-row part of the driver's data preparation ON THE DEVICE (no handle; synchronous; a process-wide workspace).
 * For every detection row det[k] = [frame (s), lon (deg), lat (deg), u, v, confidence] (sim/nadir_sim.py:236, 256) with pose index
 * ii[k] into the T ground-truth poses (pos_gt [T,3] ECI km, rot_gt [T,9] row-major camera->inertial rotation):
 *   xyz[k]  = the landmark in ECI km at the frame's second (latlon_to_eci, BA_utils.py:1238-1251 with the ellipsoid of :1221-1236 and the
 *             Greenwich angle of :1172-1218),
 *   proj[k] = its reprojection at that ground-truth pose (landmark_project, BA_utils.py:30-43; depth clamped at 0.1 km, :13),
 *   mask[k] = the driver's outlier test (od_pipe.py:930): 0 < proj < (4700, 2600), |proj - uv| < 1000 px, confidence > 0.8.
 * Agrees with the host path (vinsat_amd/od_pipe.py without a device, which is bit-identical to the reference's arrays) to rounding:
 * the device library's sin / cos are not the host's. */
int vba_prepare_rows(int device, int64_t M, const double* det /*[M,6]*/, const int64_t* ii /*[M]*/, int T, const double* pos_gt /*[T,3]*/,
                     const double* rot_gt /*[T,9]*/, const double* intrinsics /*[4] fx fy cx cy*/, double* xyz /*[M,3]*/, double* proj /*[M,2]*/,
                     unsigned char* mask /*[M]*/);

/* ---- free-landmark Schur-complement BA: ADD-ON, PARITY UNPINNED ------------------------------------------
 * The reference keeps its landmarks fixed (BA_filtering.py:32-37) and has nothing to marginalise; this mode is the
 * variant BASELINE.json's north_star describes on top of it and has NO counterpart in the reference.  Unknowns: 6 per
 * pose (position, rotation; velocities untouched) and 3 per landmark, the landmarks held by a catalogue prior
 * N(X0, sigma_prior^2 I); weights = confidences (the reference's alpha = 2 case).  One call of vba_schur_iterate builds the
 * normal equations at the resident state, marginalises the landmarks (3x3 inversions), factorises the dense reduced
 * camera system on the matrix cores (blocked Cholesky), back-substitutes, and keeps the step if the cost
 * sum w |r|^2 + |X - X0|^2 / sigma^2 went down.  Validated against oracle/schur_oracle.py (this repository's own CPU
 * restatement) only.  It shares nothing with a vba_handle and never runs inside vba_iterate.
 *
 * Structure arrays (built by the host once per window, vinsat_amd/schur.py): rows sorted by landmark with CSR lm_ptr[L+1];
 * row_pose / row_lm [m]; the rows of every pose as CSR pose_ptr[n+1] -> pose_rows[m]; and, for every 6x6 block (i >= j) of
 * the reduced system that two poses sharing a landmark touch (every diagonal block included), the list of row pairs
 * (pair_k of pose i, pair_k2 of pose j, same landmark) as CSR blk_ptr[nblk+1], with blk_i / blk_j [nblk]. */
typedef struct vba_schur_context* vba_schur_handle;
const char* vba_schur_last_error(void);
int vba_schur_create(int device, int n, int64_t m, int L, int nblk, int64_t npairs, vba_schur_handle* out);
int vba_schur_destroy(vba_schur_handle h);
int vba_schur_upload(vba_schur_handle h, const int* lm_ptr, const int* row_pose, const int* row_lm, const double* row_u,
                     const double* row_v, const double* row_w, const int* pose_ptr, const int* pose_rows, const int* blk_i,
                     const int* blk_j, const int* blk_ptr, const int* pair_k, const int* pair_k2, const double* intrinsics /*[n,4]*/,
                     const double* X0 /*[L,3] catalogue positions*/, double sigma_prior);
int vba_schur_set_state(vba_schur_handle h, const double* states /*[n,10]*/, const double* landmarks /*[L,3]*/);
int vba_schur_get_state(vba_schur_handle h, double* states, double* landmarks);
/* one LM trial at damping lamda (added to every diagonal entry of B and C); *accepted = the cost went down and the state moved.
 * A reduced camera system that is not positive definite at this damping is a REJECTED trial (*accepted = 0, *cost_after =
 * *cost_before, state untouched; vba_schur_last_info tells the failing row): the caller raises lamda as after any rejection. */
int vba_schur_iterate(vba_schur_handle h, double lamda, double* cost_before, double* cost_after, int* accepted);
/* 0 if the last factorisation went through, else 1 + the row of the reduced system at which it met a non-positive pivot */
int vba_schur_last_info(vba_schur_handle h, int* info);
/* HIP-event times of the last iterate: build (blocks + Schur complement), factor (Cholesky), solve (substitutions + update) */
int vba_schur_last_ms(vba_schur_handle h, float* build_ms, float* factor_ms, float* solve_ms);
/* what = 0: the step of the last iterate [6 n + 3 L]; 1: its Cholesky factor, dense lower triangular [6n, 6n] */
int vba_schur_debug_fetch(vba_schur_handle h, int what, double* out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* VINSAT_BA_H */
