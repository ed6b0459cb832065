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
 * forming its own blocks, one cyclic-reduction level in front (vba_set_fusion bit 4), the solve as one grid of waiting blocks
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
 * (BA_utils.py:73-87) -- the parity target.  1: the coarse schedule of `propagate_orbit_dynamics_skip`
 * (BA_utils.py:52-71: steps of 100 s plus one remainder step) that the reference itself switches to when it sees a
 * GPU (`predict_gpu`, BA_filtering.py:16-17); results differ from mode 0 by the integration error. */
int vba_set_integrator(vba_handle h, int hop100);

/* Lanes per pose of the per-pose accumulation kernel (4, 8, 16, 32 or 64; 0 = choose from the handle geometry:
 * ~12 observations per lane, more lanes when few windows leave the GPU idle).  The value fixes the shape of the
 * reduction tree, i.e. results are bit-reproducible for equal settings. */
int vba_set_accumulate_lanes(vba_handle h, int lanes);

/* Tiles of 256 observation rows per block of the latency-mode trial kernel where it runs in its plain geometry (vba_set_fusion
 * bit 0 off): 0 = automatic (by the number of tiles of the handle: 4 from 600 tiles, 2 from 150), 1, 2, 4 or 8.  A block's keys
 * share one pass of bin reservations, so a window of 10^6 keys does not queue ~2000 returning atomics on each of the central
 * bins.  A performance knob only: block sums stay per tile, results are bit-identical for every value. */
int vba_set_trial_tiles(vba_handle h, int tiles);

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

/* Carried keys (default on).  The trial residual of an accepted LM trial (BA_filtering.py:61-66) is evaluated at
 * exactly the states the next BA call starts from (BA_filtering.py:12-21), so the trial kernel also leaves the |r|
 * keys, their histogram (see vba_set_warm_select) and sum |r| of the next call on the device, and a call that follows another one
 * without vba_set_states / uploads in between starts at the median select instead of re-reading every observation.
 * on == 0: every call recomputes them (same bits; for comparison). */
int vba_set_key_carry(vba_handle h, int on);

/* Warm select (default: on).  The exact lower median of the 2m keys |r| (torch.median, BA_filtering.py:23) is found by
 * radix select.  On carried keys the trial that produced them has already binned them into 2046 narrow bins around the
 * median of its own call (consecutive calls move the median by a factor 0.3 .. 2.5), so ONE pass over the keys -- the
 * compaction of the bin that holds the wanted rank -- replaces the two digit passes; the short list is ranked exactly as
 * before.  If the wanted rank falls outside the binned range the call repeats its select with the exact digits (a
 * "miss": counted by vba_warm_select_misses, same result either way).  In a chained schedule (vba_run_schedule) the
 * warm pass of call c + 1 also evaluates the LM accept test of call c in its prologue, which removes the decide launch
 * from the chain.  on == 0: every select takes the exact digit passes and every accept test its own launch (same bits).
 * on == 2 (test knob): every warm select reports a miss, i.e. every carried call takes the repeat path.
 * Latency mode (vba_create_mode) goes one step further: the trial kernel drops every key into the bucket of its warm
 * bin (capacity ~6x the densest bin; a longer bin is a miss), so the bin of the wanted rank needs no pass over the keys --
 * the accumulation kernel resolves the histogram, ranks that bucket and evaluates the folded accept test in its own
 * prologue, and a chained landmark-only call is two kernels.  on == 3: keep the select as its own kernel (comparison). */
int vba_set_warm_select(vba_handle h, int on);
/* Tuning / test knob: log2 of the width of a warm bin in bit patterns (52 = one binade).  Defaults: 44 (1/256 binade; 43 for
 * more than 300 000 keys) in latency mode, where the bin of the median must be a short bucket; 49 (1/8 binade) for handles
 * in bandwidth mode, where a block's histogram flush costs one global atomic per bin it touched and the few per cent of
 * the keys in the median's bin are compacted by one pass.  The median is exact for every width. */
int vba_set_warm_shift(vba_handle h, int shift);
int vba_warm_select_misses(vba_handle h, int* count);
/* Test knob: capacity of a bin bucket (latency mode), 8 .. the allocated one; 0 restores the default.  A bin that holds more
 * keys than that overflows -- its bucket is incomplete -- and a call whose median falls into it takes the miss path. */
int vba_set_bucket_cap(vba_handle h, int cap);

/* Kernel fusion, a bit mask (bits 0 and 1: latency mode); same results to rounding.
 *   bit 0: the trial kernel forms the step of each pose itself (landmark-only phase: the 6x6 solve; full phase: the
 *          recovery of the partitioned solve) -- no recovery launch and, in the landmark-only phase, no assembly + solve launch;
 *   bit 1: the chunk elimination forms the blocks of its chunk in LDS itself -- no assembly launch in the full phase, the
 *          bands never go through memory.  With the generic formation it gained nothing (rounds 1, 2); formed by column
 *          (asm_form_columns, a row of 16 lanes per pose row) it takes 1.1 us off the average call: default since round 3.
 *   bit 2: (bandwidth mode, sequential driver) the solve of the full phase forms each block from the per-pose inputs
 *          itself -- no assembly launch, the bands never go through memory.  Bit-exact.  With one window per wavefront
 *          (k_solve_forming) it measured slower than assembly + walk (the walk was bound by instruction issue); since the
 *          walk packs four windows into a wavefront (k_solve_quad) it is the faster form and the default.
 *   bit 3: the full-phase assembly forms each pose row with one wave in seven uniform passes (vba_asm_fast.h) instead of one
 *          entry per thread.  Bit-exact; 0.8 us off the average call of a single window (the per-entry form is a serial
 *          ~600 instructions per thread there), on par at 4096 windows (1.77 vs 1.85 ms).
 *   bit 4: only ONE cyclic-reduction level of the reduced system runs on its own CUs in front of the one-workgroup kernel
 *          (k_cr_level0) instead of two (k_cr_level01, default).  Same bits; 0.9 us per call slower.  Comparison / tests.
 *   bit 5: (latency mode, partitioned solve with two split-off levels) chunk elimination and the two cyclic-reduction levels
 *          run as ONE grid (k_solve_resident): the consumer blocks are resident from the start and wait for their producers
 *          on flags in device memory -- bounded: a consumer that gives up flags its window and the call returns VBA_ESTATE.
 *   bit 6: ... and the one-workgroup tail as well (one launch for the whole solve).
 *          Same bits as three launches.  Measured SLOWER (C3, one window: +2.5 and +6.5 us per call): a hop over a flag is two
 *          round trips to device memory plus the write-back / invalidate of the per-XCD L2s, 4.7 us from the last producer's
 *          last store to the consumer's first load, against ~3.2 us for a kernel boundary.  Comparison / tests only.
 *   bits 4 .. 6 exist in the comparison build only (vba_has_variants); the default build answers VBA_EINVAL.
 * Default: 15 (bits 0 .. 3) for one window of fewer than 150 000 rows and 1500 poses (a bigger one: 14, with the trial kernel's
 * observation blocks of several tiles, vba_set_trial_tiles -- C4 17.4 against 15.4 k it/s, C5 12.2 against 10.1; the pipelined
 * vba_iterate_resident works with either) and for bandwidth mode; latency-mode handles of several windows drop the fusions that
 * trade instructions for launches once launches are no longer what a call costs -- 15 up to 175 000 rows per launch, 14 (the trial
 * kernel reads a step that its own launch formed) up to 450 000, 12 (the assembly is a launch as well) beyond: from the round-4
 * sweep, DESIGN.md section 3.  Measured on MI355X (C3, one window): bit 0 takes 2.7 us off the average call once the step of a pose is
 * formed by 16 lanes together (formed redundantly by every thread it was 6 us SLOWER: instruction issue of a single wave
 * is the time in this mode); bit 1, see above (48.0 against 49.1 us per call).  All masks are covered by the parity tests. */
int vba_set_fusion(vba_handle h, int mask);

/* Partitioned solve: waves per chunk.  2 (default): every chunk of 4 or more blocks is eliminated from both ends by two
 * waves that meet at its middle block -- half the dependent block steps of the kernel that is the longest of a
 * full-phase call in latency mode.  1: one wave walks the chunk left to right.  The two orders round differently
 * (~1e-9 relative on the step, the order of the difference to the reference's dense LU); each is deterministic. */
int vba_set_chunk_waves(vba_handle h, int waves);

/* Row pivoting inside the 9x9 diagonal blocks.  always == 0 (default): the blocks are eliminated without row
 * exchanges (the damped normal equations are positive definite up to a ~1e-6 non-symmetric term) while every pivot
 * is checked against the diagonal entry it started from; a failed check repeats that solve with pivoting, so the
 * result is never taken from an unchecked elimination.  always != 0: pivot from the start.
 * vba_solver_fallbacks reports how many solves were repeated (diagnostic). */
int vba_set_pivoting(vba_handle h, int always);
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
 * from the keys the last accepted trial left behind (vba_set_key_carry).  Same bits as vba_iterate. */
int vba_iterate_resident(vba_handle h, int iter, int initialize, double* states_out, double* lamda_out,
                         double* last_hessian, int* n_trials, unsigned* flags);

/* vba_iterate as the FIRST call of a driver loop whose following calls are vba_iterate_resident (a new window, or states the caller
 * changed): the states go up and the call is served like a resident one -- returned as soon as its accept test is known, the next
 * call enqueued behind it (vba_set_pipeline); the watched host buffers are compared as in a resident call (VBA_FLAG_HOST_CHANGED).
 * Same bits as vba_iterate.  A caller that replaces the states before every call
 * should use vba_iterate: there the speculated call would be waited for and dropped each time. */
int vba_iterate_open(vba_handle h, int iter, int initialize, double lamda_in, const double* states_in, double* states_out,
                     double* lamda_out, double* last_hessian, int* n_trials, unsigned* flags);

/* Pipelining of the driver loop (default on; handles of one window).  vba_iterate_resident returns call k as soon as its
 * accept test is known and has by then enqueued call k + 1 speculatively (iter + 1 / the same phase until the caller has
 * been seen doing something else after that iter), so the device works through the caller's host-side turnaround; a call
 * that was not asked for after all is waited for and dropped, at the price of the carried keys.  Results are bit-identical
 * to vba_step.  The call speculated behind a pipelined call reuses its scratch (maximum weight, step, trial states), so
 * vba_debug_fetch refuses (VBA_ESTATE) after a pipelined call: switch the pipeline off to inspect intermediates.
 * vba_pipeline_stats: speculated calls that were used / dropped. */
int vba_set_pipeline(vba_handle h, int on);
/* Host buffers of the caller whose content was uploaded (e.g. the ndarray arguments ii / time_idx of BA()) and that the caller
 * might edit in place: every vba_iterate_resident compares `live` with the reference `copy` (bytes each, both must stay valid;
 * 8 slots -- one per array argument of BA() --, live == NULL clears one) while the device works -- from 64 kB of watched bytes on a helper
 * thread of the handle, beside the enqueue of the speculated call -- and reports a difference as VBA_FLAG_HOST_CHANGED. */
int vba_set_host_watch(vba_handle h, int slot, const void* live, const void* copy, int64_t bytes);
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

/* vba_run_schedule on a latency-mode handle captures the launches of its first pass (~70 dependent kernels for the driver's 20 calls)
 * as a hipGraph and replays it while nothing that goes into those launches has changed (the per-call kernel arguments are compared by
 * hash, the schedule and the host-side switches by value): 45.2 -> 42.6 us per call at C3.  Same kernels, same arguments, same bits;
 * calls that stall (rejected trial, missed select) are finished by the host afterwards as without a graph.  Default on; off = kernel by
 * kernel (comparison, debugging; also the environment variable VBA_NO_GRAPH).  A capture or launch that fails once switches it off for
 * the handle.  vba_schedule_graph_stats: graphs captured / replays so far. */
int vba_set_schedule_graph(vba_handle h, int on);
int vba_schedule_graph_stats(vba_handle h, int* captures, int* replays);

/* Class times of the CHAINED schedule (what vba_run_schedule really runs, as opposed to vba_step_profiled's serialised one).
 * on != 0: every vba_run_schedule records HIP events on the handle's stream at three boundaries of each call -- in front of
 * the call's first kernel, behind its accumulation (+ assembly, where that is a launch), behind its solve kernels, behind its
 * trial kernel -- and, if every call of the schedule was accepted at its first trial, adds the three intervals to the sums.
 * vba_chain_profile: ms[3] = summed milliseconds of the classes {accumulate (with the select / accept test folded into it, or
 * launched in front of it), solve, trial}, launches[3] = how many intervals each sum holds (a landmark-only call whose step the
 * trial kernel forms has no solve interval); reset != 0 clears the sums.  The markers cost ~1 us each on the chain: use a
 * profiled schedule for class times, an unprofiled one for throughput. */
int vba_set_chain_profile(vba_handle h, int on);
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
 *      leaves the |r| keys of ITS rows behind in per-bin buckets, with their warm histogram (vba_set_warm_select) and block sums,
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
