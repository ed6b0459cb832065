// vba_decide.h -- the LM accept test (BA_filtering.py:51, 66-79) as a device function of a 256-thread block.
//
// Two callers evaluate it with the same arithmetic (fixed-order sums over the same block partials, so the same bits):
//   * k_decide (vba_solve.hip): its own launch, one block per window -- every trial of a call-by-call step, the later
//     trials of any call, the last call of a chained schedule;
//   * the first kernel of the NEXT call of a chained schedule (k_select_warm, vba_obs.hip): every block of that kernel
//     re-evaluates the test of the call in front of it in its prologue and simply goes on if the first trial was accepted
//     -- the kernel boundary, the launch and the single-block tail of a separate decide kernel are gone from the chain.
//     A trial that is not cleanly accepted (rejected, pivot check failed, non-finite) leaves everything untouched: the
//     window stalls there and the host finishes that call with k_decide and the ordinary loop.
//
// Everything the test reads was written by earlier kernels and is written by nobody while it is evaluated (call parity,
// see WinScalars): the blocks of the folding kernel cannot disagree.
#pragma once

#include "vba_device.h"

namespace vba {

struct DecideOut {
    int accept, stop;
    unsigned flags;             // flags of the decided call so far (pivot check, non-finite step, ...)
    double lam_next;            // damping after this trial: lam * 10 (BA_filtering.py:72)
    double lam_out;             // what the call returns when it stops here: clamp(lam_next * 0.01, 1e-4, 1e-1) (:79)
    double residual, init_residual;
    double sum_pred;            // sqrt(Sigma) * sum |r_pred| at the input states (recorded at the first trial)
    double sum_next;            // sum |r_obs| at the trial states (carried keys)
};

// What a thread reads for the test: its share of the block partials and the scalars of the decided call.  Loaded apart from
// the evaluation so that a caller can have the loads in flight while it does something else (k_obs_accumulate).
struct DecideIn {
    double s_pred, s_trial, s_next, s_prior, lam_in, so;
    unsigned flags;
};

// pc: parity of the decided call; prm: ITS per-call constants
__device__ __forceinline__ DecideIn decide_load(const DevView& V, int w, int pc, const StepParams& prm, int ranks) {
    const WinScalars& sc = V.sc[w];
    const int t = threadIdx.x;
    const bool reg = V.reg && !prm.initialize;
    DecideIn in;
    in.s_pred = in.s_trial = in.s_next = in.s_prior = 0.0;
    if (!prm.initialize) {
        const double* pp = V.part_pred + ((size_t)w * 2 + pc) * V.pred_stride;
        for (int b = t; b < V.nblk_pred + V.nblk_long; b += 256) in.s_pred += pp[b];    // (the long edges' slots behind the blocks')
        if (reg) {
            const double* pq = V.part_prior + ((size_t)w * 2 + pc) * V.pred_stride;
            for (int b = t; b < V.nblk_pred; b += 256) in.s_prior += pq[b];
        }
    }
    if (ranks == 0) {
        const double* pt = V.part_trial + (size_t)w * V.trial_stride;
        // (the long edges' slots behind the blocks': written by k_long_trial, which has nothing to do in a landmark-only call)
        for (int b = t; b < V.nblk_obs + V.nblk_dyn + (prm.initialize ? 0 : V.nblk_long); b += 256) in.s_trial += pt[b];
    }
    if (V.emit) {
        const double* pn = V.part_next + (size_t)w * V.nblk_obs;
        for (int b = t; b < V.nblk_obs; b += 256) in.s_next += pn[b];
    }
    in.lam_in = sc.lam[pc];
    in.so = sc.sum_in[pc];
    in.flags = sc.fl[pc];
    return in;
}

// n_trials_in / init_prev: trials already counted and the initial residual recorded by the first of them.
// red: [5][4] doubles of LDS.  All 256 threads return the same values.
__device__ __forceinline__ DecideOut decide_finish(const DevView& V, int w, const DecideIn& in, const StepParams& prm, int n_trials_in,
                                                   double init_prev, const double* trial_all, int ranks, double (*red)[4]) {
    const int t = threadIdx.x;
    const int n = V.n[w], m = V.m[w];
    const bool reg = V.reg && !prm.initialize;
    const double lam_in = in.lam_in, so = in.so;
    const unsigned flags = in.flags;
    double sh_trial = 0.0;
    if (ranks > 0) {        // sharded: rank-ordered sum of the gathered per-rank sums; every rank holds the same dynamics part
        sh_trial = trial_all[1];
        for (int q = 0; q < ranks; ++q) sh_trial += trial_all[2 * q];
    }
    const double v4[4] = {wave_sum(in.s_pred), wave_sum(in.s_trial), wave_sum(in.s_next), wave_sum(in.s_prior)};
    __syncthreads();        // red may still be read from a previous use
    if ((t & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[q][t >> 6] = v4[q];
    }
    __syncthreads();
    double tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tot[q] = ((red[q][0] + red[q][1]) + red[q][2]) + red[q][3];
    DecideOut o;
    o.flags = flags;
    const double M = V.m_total ? (double)V.m_total : (double)m;
    // BA_reg: the prior adds 7 entries per pose to both means (6 zeros per pose in the landmark-only phase); the 7th is
    // prior_gpu's constant rotation residual quat_coeff: 1 in the initial residual, 100 in every trial, as the reference
    // passes its coefficients (BA_filtering.py:121, 163 vs :175, 178-180)
    const double denom = 2.0 * M + (prm.initialize ? 6.0 : 7.0) * (double)(n - 1) +
                         (V.reg ? (prm.initialize ? 6.0 : 7.0) * (double)n : 0.0);
    o.sum_pred = prm.initialize ? 0.0 : tot[0] * prm.sqrt_sigma;
    o.init_residual = init_prev;
    if (n_trials_in == 0)   // mean |[r_obs ; sqrt(Sigma) r_pred]| with UNweighted r_obs (BA_filtering.py:51)
        o.init_residual = (so + o.sum_pred + (reg ? tot[3] + 1.0 * (double)n : 0.0)) / denom;
    const double S = (ranks > 0 ? sh_trial : tot[1]) + (reg ? 100.0 * (double)n : 0.0);
    o.residual = S / denom;
    o.lam_next = lam_in * 10.0;
    o.accept = o.residual < o.init_residual;
    o.stop = o.accept || o.lam_next > 1e4;
    o.lam_out = fmax(fmin(1e-1, o.lam_next * 0.01), 1e-4);
    o.sum_next = tot[2];
    return o;
}

__device__ __forceinline__ DecideOut decide_eval(const DevView& V, int w, int pc, const StepParams& prm, int n_trials_in,
                                                 double init_prev, const double* trial_all, int ranks, double (*red)[4]) {
    const DecideIn in = decide_load(V, w, pc, prm, ranks);
    return decide_finish(V, w, in, prm, n_trials_in, init_prev, trial_all, ranks, red);
}

}  // namespace vba
