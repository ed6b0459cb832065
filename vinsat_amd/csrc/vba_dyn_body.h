// vba_dyn_body.h -- the dynamics factor of one 256-thread block (A4 + A5), shared by k_dynamics (its own launch on a
// second stream when many windows are batched) and by k_obs_accumulate (few windows: the dynamics blocks ride in the
// accumulation's grid, which runs about as long and is just as register-hungry -- no second stream, no cross-stream
// join in front of the assembly: that join cost 5.5 us on average, 21 us at the 90th percentile).
//
// The reference differentiates the RK4 chain with reverse-mode autograd into a dense [6(n-1), 9n] Jacobian
// (BA_utils.py:506); here every pose carries its six tangent vectors forward through the same RK4 steps
// (8 lanes per pose: lanes 0-5 one tangent each, lane 6 the attitude term), so only the 6x6 block that is
// actually non-zero is ever produced.
#pragma once

#include "vba_device.h"

namespace vba {

// Besides the factor itself the block leaves the fixed-order sums of |r_pred| (and, BA_reg, of |r_prior|) over its poses
// behind: the accept test adds up a handful of block partials instead of walking the pose arrays.  The partials are kept per
// call parity: in a chained schedule the accept test of call c is evaluated in the prologue of the accumulation of call
// c + 1, whose grid also carries the blocks that form the factor of call c + 1.
__device__ __forceinline__ void dynamics_block(const DevView& V, int w, int block) {
    __shared__ double dred[4];
    const int n = V.n[w];
    const int gid = block * 256 + threadIdx.x;
    const int i = gid / kDynLanes, c = gid % kDynLanes;
    const size_t pb = (size_t)w * V.n_max + (i < n ? i : 0);
    const double* st = V.states + pb * 10;
    double s_pred = 0.0, s_prior = 0.0;
    if (i < n && c < 6) {
        const int steps = V.steps[pb];
        // the last pose's propagation is discarded by the reference (BA_utils.py:476); a long edge (marked by a negative count) is
        // k_long_factor's: transition matrix, prediction, residual and its block sum
        if (i < n - 1 && (steps > 0 || V.hop)) {
            double x[6] = {st[0], st[1], st[2], st[7], st[8], st[9]};
            double t[6] = {0, 0, 0, 0, 0, 0};
            t[c] = 1.0;
            propagate_gap<true>(x, t, abs(steps), V.hop);
            double* Phi = V.Phi + pb * 36;
#pragma unroll
            for (int r = 0; r < 6; ++r) Phi[6 * r + c] = t[r];
            if (c == 0) {
                const double* sn = st + 10;
                double* xh = V.xhat + pb * 6;
                double* ro = V.rorb + pb * 6;
#pragma unroll
                for (int r = 0; r < 6; ++r) xh[r] = x[r];
                ro[0] = x[0] - sn[0];
                ro[1] = x[1] - sn[1];
                ro[2] = x[2] - sn[2];
                ro[3] = (x[3] - sn[7]) * kVelCoeff;
                ro[4] = (x[4] - sn[8]) * kVelCoeff;
                ro[5] = (x[5] - sn[9]) * kVelCoeff;
                s_pred = fabs(ro[0]) + fabs(ro[1]) + fabs(ro[2]) + fabs(ro[3]) + fabs(ro[4]) + fabs(ro[5]);
            }
        }
    } else if (i < n && c == 6) {
        const double* qp = i > 0 ? st - 10 + 3 : nullptr;
        const double* cp = i > 0 ? V.cumrot + (pb - 1) * 4 : nullptr;
        const double* qn = i < n - 1 ? st + 10 + 3 : nullptr;
        double f, qg[3], Hd[9], Hu[9], Hl[9];
        attitude_term(qp, cp, st + 3, V.cumrot + pb * 4, qn, f, qg, Hd, Hu, Hl);
        V.fatt[pb] = f;
        if (i < n - 1) s_pred = fabs(f);
#pragma unroll
        for (int k = 0; k < 3; ++k) V.qgrad[pb * 3 + k] = qg[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            V.Hd[pb * 9 + k] = Hd[k];
            V.Hu[pb * 9 + k] = Hu[k];
            V.Hl[pb * 9 + k] = Hl[k];
        }
    } else if (i < n && V.reg) {    // lane 7: sum |r_prior| at the input states (BA_filtering.py:163)
        double r6[6];
        prior_residual(V.prior_H + pb * 36, V.prior_x + pb * 6, st, r6);
        s_prior = fabs(r6[0]) + fabs(r6[1]) + fabs(r6[2]) + fabs(r6[3]) + fabs(r6[4]) + fabs(r6[5]);
    }
    const double tp = block_sum<256>(s_pred, dred);
    if (threadIdx.x == 0) V.part_pred[((size_t)w * 2 + V.par) * V.pred_stride + block] = tp;
    if (V.reg) {
        const double tq = block_sum<256>(s_prior, dred);
        if (threadIdx.x == 0) V.part_prior[((size_t)w * 2 + V.par) * V.pred_stride + block] = tq;
    }
}

}  // namespace vba
