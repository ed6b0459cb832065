// vba_dyn.hip -- pose-chain kernels: orbit-dynamics factor with its state-transition matrix (A4), attitude
// Newton term (A5) and block-tridiagonal assembly (A6).
//
// The reference differentiates the RK4 chain with reverse-mode autograd into a dense [6(n-1), 9n] Jacobian
// (BA_utils.py:506); here every pose carries its six tangent vectors forward through the same RK4 steps
// (8 lanes per pose: lanes 0-5 one tangent each, lane 6 the attitude term), so only the 6x6 block that is
// actually non-zero is ever produced.
#include "vba_asm.h"
#include "vba_asm_fast.h"
#include "vba_device.h"
#include "vba_dyn_body.h"
#include "vba_launch.h"
#include "vba_step.h"

namespace vba {

__global__ __launch_bounds__(256) void k_dynamics(DevView V) {
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    dynamics_block(V, w, blockIdx.x);
}

// Batched windows: TWO lanes per pose, three tangents each, 128 poses per block.  With thousands of windows there are
// poses enough to fill the chip without spreading one pose over eight lanes, and the eight-lane form pays for that
// spread: the base trajectory is integrated by all six tangent lanes, and the attitude term and the prior are divergent
// sections that a wave runs for 8 poses at a time (here for 32).  Per tangent the arithmetic is that of dynamics_block
// (rk4_step_multi); the block sums of |r_pred| / |r_prior| are formed in the order of the eight-lane layout (a wave_sum
// over the 64 slots of 8 poses, four of them added in sequence), so the accept test sees the same bits in both modes.
__global__ __launch_bounds__(256) void k_dynamics_pair(DevView V) {
    __shared__ double slot[128 * 3];        // per pose: sum |r_orbit|, |f_att|, sum |r_prior|
    __shared__ double vw[2][16];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int n = V.n[w];
    const int lp = threadIdx.x >> 1, half = threadIdx.x & 1;
    const int i = blockIdx.x * 128 + lp;
    const size_t pb = (size_t)w * V.n_max + (i < n ? i : 0);
    const double* st = V.states + pb * 10;
    double s_orb = 0.0, s_att = 0.0, s_pri = 0.0;
    if (i < n) {
        if (i < n - 1) {            // the last pose's propagation is discarded by the reference (BA_utils.py:476)
            double x[6] = {st[0], st[1], st[2], st[7], st[8], st[9]};
            double t[3][6];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 6; ++r) t[j][r] = (r == 3 * half + j) ? 1.0 : 0.0;
            const int steps = V.steps[pb];
            const bool mine = steps > 0 || V.hop;      // (a long edge is k_long_factor's: transition matrix, prediction, residual, block sum)
            if (mine) propagate_gap_multi<3>(x, t, abs(steps), V.hop);
            double* Phi = V.Phi + pb * 36;
            if (mine) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 6; ++r) Phi[6 * r + 3 * half + j] = t[j][r];
            }
            if (half == 1 && mine) {
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
                s_orb = fabs(ro[0]) + fabs(ro[1]) + fabs(ro[2]) + fabs(ro[3]) + fabs(ro[4]) + fabs(ro[5]);
            }
        }
        if (half == 0) {
            const double* qp = i > 0 ? st - 10 + 3 : nullptr;
            const double* cp = i > 0 ? V.cumrot + (pb - 1) * 4 : nullptr;
            const double* qn = i < n - 1 ? st + 10 + 3 : nullptr;
            double f, qg[3], Hd[9], Hu[9], Hl[9];
            attitude_term(qp, cp, st + 3, V.cumrot + pb * 4, qn, f, qg, Hd, Hu, Hl);
            V.fatt[pb] = f;
            if (i < n - 1) s_att = fabs(f);
#pragma unroll
            for (int k = 0; k < 3; ++k) V.qgrad[pb * 3 + k] = qg[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                V.Hd[pb * 9 + k] = Hd[k];
                V.Hu[pb * 9 + k] = Hu[k];
                V.Hl[pb * 9 + k] = Hl[k];
            }
        } else if (V.reg) {         // sum |r_prior| at the input states (BA_filtering.py:163)
            double r6[6];
            prior_residual(V.prior_H + pb * 36, V.prior_x + pb * 6, st, r6);
            s_pri = fabs(r6[0]) + fabs(r6[1]) + fabs(r6[2]) + fabs(r6[3]) + fabs(r6[4]) + fabs(r6[5]);
        }
    }
    if (half == 0) slot[lp * 3 + 1] = s_att;
    else { slot[lp * 3 + 0] = s_orb; slot[lp * 3 + 2] = s_pri; }
    __syncthreads();
    // the sums of dynamics_block: 32 poses per partial, 8 slots per pose (slot 0 the orbit residual, 6 the attitude
    // residual, 7 the prior), a butterfly over the 64 slots of 8 poses, then those four in sequence
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vwave = wv * 4 + q;
        const int pl = vwave * 8 + (lane >> 3), c = lane & 7;
        const double a = c == 0 ? slot[pl * 3] : (c == 6 ? slot[pl * 3 + 1] : 0.0);
        const double sa = wave_sum(a);
        if (lane == 0) vw[0][vwave] = sa;
        if (V.reg) {
            const double sb2 = wave_sum(c == 7 ? slot[pl * 3 + 2] : 0.0);
            if (lane == 0) vw[1][vwave] = sb2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int vb = blockIdx.x * 4 + threadIdx.x;
        if (vb < V.nblk_pred) {
            double tp = 0.0, tq = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { tp += vw[0][threadIdx.x * 4 + k]; tq += vw[1][threadIdx.x * 4 + k]; }
            V.part_pred[((size_t)w * 2 + V.par) * V.pred_stride + vb] = tp;
            if (V.reg) V.part_prior[((size_t)w * 2 + V.par) * V.pred_stride + vb] = tq;
        }
    }
}

// Block-tridiagonal assembly (BA_filtering.py:40-48): 3 x 81 band entries + 9 right-hand-side entries per pose.
// A block of 256 threads takes kAsmPoses consecutive poses (4 for a few windows: more blocks, shorter; 16 when
// batched windows fill the chip anyway: the decode of an entry and the halo slot are amortised over more poses): the per-pose inputs (141 doubles each, plus the
// transition matrix of the pose in front) are staged once in LDS with coalesced loads, then every thread forms
// entries from LDS and the block writes its 252 * kAsmPoses outputs contiguously.
// FUSE (landmark-only phase, first LM trial, unpivoted path): the system is block diagonal and, inside a pose,
// the velocity rows carry only the damping, so the step of a pose is a 6x6 solve of (H_i / w_max + lamda I) x = b_i /
// w_max.  One thread per pose does it in registers straight from the staged inputs, retracts and writes the trial
// state -- the separate solve and recover launches of that trial are not needed.  The pivots are checked exactly as
// in the chain solver; a failed check hands the window to the pivoted kernels.
// REG (BA_reg, full phase only): the per-pose prior is staged and added as well.
template <bool FUSE, int kAsmPoses, bool REG>
__global__ __launch_bounds__(256) void k_assemble(DevView V) {
    constexpr int kAsmIn = kAsmBase + (REG ? kAsmPrior : 0);
    __shared__ double in[(kAsmPoses + 1) * kAsmIn];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int n = V.n[w];
    const int i0 = blockIdx.x * kAsmPoses;
    if (i0 >= n) return;
    const StepParams& prm = V.prm;
    const size_t sb = (size_t)w * V.n_max;
    const bool dyn = !prm.initialize;
    // REG: the host launches it only for full-phase BA_reg calls (BA_utils.py:609-612)
    // slot 0 = pose i0-1 (only Phi and rorb are used), slots 1..kAsmPoses = poses i0 ..
    // (few windows: every load in flight at once; many: the plain loop -- other blocks hide the round trips and it is
    // the faster one there, 1.77 against 1.87 ms at 4096 windows)
    if constexpr (kAsmPoses <= 4) asm_stage_all<REG, kAsmPoses + 1, 256>(V, w, n, dyn, i0 - 1, in, threadIdx.x);
    else asm_stage<REG>(V, w, n, dyn, i0 - 1, kAsmPoses + 1, in, threadIdx.x, 256);
    __syncthreads();
    const double inv_wmax = 1.0 / bits_f64(V.sc[w].wmax_bits[V.par]);
    const int cnt = min(kAsmPoses, n - i0);
    // thread t forms entry t of every pose of the block: which band / row / column it is (and with that every index
    // into the staged inputs) is decoded once, and for a fixed pose the 252 threads write consecutive addresses
    if (threadIdx.x < 252) {
        const int t = threadIdx.x;
        const bool is_rhs = t >= 243;
        const int which = t / 81, a = is_rhs ? t - 243 : (t % 81) / 9, b = t % 9;
#pragma unroll
        for (int p = 0; p < kAsmPoses; ++p) {
            if (p >= cnt) continue;
            const int i = i0 + p;
            const double* me = in + (p + 1) * kAsmIn;
            const double* pv = in + p * kAsmIn;
            const AsmRow R = asm_row<REG>(me, pv, i, n, dyn, prm.sigma, inv_wmax);
            // landmark-only phase: the off-diagonal blocks are zero and nobody reads them (k_solve_blockdiag takes
            // the diagonal block only; vba_debug_fetch reports them as zeros)
            // FUSE: the step is formed below from the staged inputs, so the diagonal blocks and right-hand sides are not
            // written at all (872 -> 152 bytes per pose); a later trial of the call -- the rare one -- has the host run the
            // plain assembly first.  Only the last pose's diagonal block leaves (last_hessian).
            if (FUSE) {
                if (which == 1 && !is_rhs && i == n - 1) V.lastD[(size_t)w * 81 + (t - 81)] = band_entry(R, which, a, b);
                continue;
            }
            if (is_rhs) V.rhs[(sb + i) * 9 + a] = rhs_entry(R, a);
            else if (dyn || which == 1) {
                const double e = band_entry(R, which, a, b);
                V.bands[(sb + i) * 243 + t] = e;
                if (which == 1 && i == n - 1) V.lastD[(size_t)w * 81 + (t - 81)] = e;      // BA_filtering.py:97
            }
        }
    }
    if (FUSE) {
        WinScalars& sc = V.sc[w];
        const double lam32 = (double)(float)sc.lam[V.par];      // torch.eye() is float32 (BA_filtering.py:54)
        if (blockIdx.x == 0 && threadIdx.x == 0) sc.lam32 = lam32;
        // 16 lanes per pose: lane c < 7 owns column c of the 6x6 system [A | b] (vba_step.h); lane 0 retracts and stores
        bool badpiv = false, badnum = false;
        if ((int)threadIdx.x < kAsmPoses * 16) {
            const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15, gbase = (threadIdx.x & 63) & ~15;
            const bool live = grp < cnt;
            const double* me = in + ((live ? grp : 0) + 1) * kAsmIn;
            double d9[9];
            if (!step_blockdiag6_group(me, me + 21, inv_wmax, lam32, l16, gbase, d9) && live) badpiv = true;
            if (live && l16 == 0) {
                const int i = i0 + grp;
                double o[10];
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    badnum |= !(fabs(d9[r]) <= 1.79e308);
                    V.dpose[(sb + i) * 9 + r] = d9[r];
                }
                retract(V.states + (sb + i) * 10, d9, o);
#pragma unroll
                for (int r = 0; r < 10; ++r) V.states_new[(sb + i) * 10 + r] = o[r];
                if (V.host_states) {        // (one-window handles: sb == 0; a pipelined call reads its result from host memory)
#pragma unroll
                    for (int r = 0; r < 10; ++r) V.host_states[((size_t)V.par * V.n_max + i) * 10 + r] = o[r];
                }
            }
        }
        {
            const unsigned long long bp = __ballot(badpiv), bn = __ballot(badnum);
            if ((threadIdx.x & 63) == 0 && (bp || bn)) atomicOr(&sc.fl[V.par], (bp ? (8u | 16u) : 0u) | (bn ? 2u : 0u));
        }
    }
}

// Landmark-only phase, first LM trial, unpivoted path, MANY windows: the step of every pose straight from its per-pose sums,
// one THREAD per pose (step_blockdiag6, vba_step.h: the 6x6 Gauss-Jordan in registers, then the retraction).  With millions of
// poses per launch what counts is instructions per pose: the 16-lanes-per-pose form of the latency mode (k_assemble<FUSE>)
// spends a wave on 4 poses and runs the retraction in one lane of sixteen.  Same arithmetic per element, same bits.
__global__ __launch_bounds__(256) void k_init_step(DevView V) {
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int n = V.n[w];
    if ((int)blockIdx.x * 256 >= n) return;
    WinScalars& sc = V.sc[w];
    const size_t sb = (size_t)w * V.n_max;
    const double lam32 = (double)(float)sc.lam[V.par];      // torch.eye() is float32 (BA_filtering.py:54)
    if (blockIdx.x == 0 && threadIdx.x == 0) sc.lam32 = lam32;
    const double inv_wmax = 1.0 / bits_f64(sc.wmax_bits[V.par]);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    bool badpiv = false, badnum = false;
    if (live) {
        double H[21], b[6], d9[9], o[10];
        const double* Hg = V.Hraw + (sb + i) * 21;
        const double* bg = V.braw + (sb + i) * 6;
#pragma unroll
        for (int q = 0; q < 21; ++q) H[q] = Hg[q];
#pragma unroll
        for (int q = 0; q < 6; ++q) b[q] = bg[q];
        if (!step_blockdiag6(H, b, inv_wmax, lam32, d9)) badpiv = true;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            badnum |= !(fabs(d9[r]) <= 1.79e308);
            V.dpose[(sb + i) * 9 + r] = d9[r];
        }
        retract(V.states + (sb + i) * 10, d9, o);
#pragma unroll
        for (int r = 0; r < 10; ++r) V.states_new[(sb + i) * 10 + r] = o[r];
        if (i == n - 1) {       // the last pose's diagonal block leaves for last_hessian (BA_filtering.py:97; band_entry, no dynamics)
            for (int e = 0; e < 81; ++e) {
                const int a = e / 9, c = e % 9;
                V.lastD[(size_t)w * 81 + e] = (a < 6 && c < 6) ? Hg[sym6(a, c)] * inv_wmax : 0.0;
            }
        }
    }
    const unsigned long long bp = __ballot(badpiv), bn = __ballot(badnum);
    if ((threadIdx.x & 63) == 0 && (bp || bn)) atomicOr(&sc.fl[V.par], (bp ? (8u | 16u) : 0u) | (bn ? 2u : 0u));
}

// Full-phase assembly, one wave per pose in seven uniform passes (vba_asm_fast.h): same staging, same entries to the bit,
// about a third of the instructions of k_assemble's per-entry form -- which was VALU-bound, not bandwidth-bound.
template <int kAsmPoses, bool REG>
__global__ __launch_bounds__(256) void k_assemble_rows(DevView V) {
    constexpr int kAsmIn = kAsmBase + (REG ? kAsmPrior : 0);
    __shared__ double in[(kAsmPoses + 1) * kAsmIn];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int n = V.n[w];
    const int i0 = blockIdx.x * kAsmPoses;
    if (i0 >= n) return;
    const size_t sb = (size_t)w * V.n_max;
    if constexpr (kAsmPoses <= 4) asm_stage_all<REG, kAsmPoses + 1, 256>(V, w, n, true, i0 - 1, in, threadIdx.x);
    else asm_stage<REG>(V, w, n, true, i0 - 1, kAsmPoses + 1, in, threadIdx.x, 256);
    __syncthreads();
    const double inv_wmax = 1.0 / bits_f64(V.sc[w].wmax_bits[V.par]);
    const double sigma = V.prm.sigma;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const AsmLanes g = asm_lanes(lane);
    const int cnt = min(kAsmPoses, n - i0);
    for (int p = wave; p < cnt; p += 4) {
        const int i = i0 + p;
        double* bands = V.bands + (sb + i) * 243;
        double* rhs = V.rhs + (sb + i) * 9;
        double* lastD = i == n - 1 ? V.lastD + (size_t)w * 81 : nullptr;      // BA_filtering.py:97
        asm_form_row<REG>(g, in + (size_t)(p + 1) * kAsmIn, in + (size_t)p * kAsmIn, i < n - 1, i > 0, sigma, inv_wmax, lane,
                          [&](int e, double v) {
                              if (e < 243) {
                                  bands[e] = v;
                                  if (lastD && e >= 81 && e < 162) lastD[e - 81] = v;
                              } else {
                                  rhs[e - 243] = v;
                              }
                          });
    }
}

void launch_dynamics(const DevView& V, hipStream_t s) {
    if (!V.lat) {
        hipLaunchKernelGGL(k_dynamics_pair, dim3((V.n_max + 127) / 128, V.W), dim3(256), 0, s, V);
    } else {
        const int nb = (V.n_max * kDynLanes + 255) / 256;
        hipLaunchKernelGGL(k_dynamics, dim3(nb, V.W), dim3(256), 0, s, V);
    }
    launch_long_factor(V, s);
}

void launch_assemble(const DevView& V, int fuse_init_solve, hipStream_t s) {
#ifndef VBA_ASM_BATCHED
#define VBA_ASM_BATCHED 16
#endif
    const bool reg = V.reg && !V.prm.initialize;
    if (fuse_init_solve && !V.lat) {     // (few windows: the 16-lanes-per-pose form below, latency)
        hipLaunchKernelGGL(k_init_step, dim3((V.n_max + 255) / 256, V.W), dim3(256), 0, s, V);
        return;
    }
    const bool rows = !V.prm.initialize && !fuse_init_solve && V.asm_rows;     // full phase: the uniform-pass form
    if (!V.lat) {
        constexpr int P = VBA_ASM_BATCHED;
        const dim3 g((V.n_max + P - 1) / P, V.W);
        if (rows) {
            if (reg) hipLaunchKernelGGL((k_assemble_rows<P, true>), g, dim3(256), 0, s, V);
            else hipLaunchKernelGGL((k_assemble_rows<P, false>), g, dim3(256), 0, s, V);
        } else if (fuse_init_solve) hipLaunchKernelGGL((k_assemble<true, P, false>), g, dim3(256), 0, s, V);
        else if (reg) hipLaunchKernelGGL((k_assemble<false, P, true>), g, dim3(256), 0, s, V);
        else hipLaunchKernelGGL((k_assemble<false, P, false>), g, dim3(256), 0, s, V);
    } else {
        const dim3 g((V.n_max + 3) / 4, V.W);
        if (rows) {
            if (reg) hipLaunchKernelGGL((k_assemble_rows<4, true>), g, dim3(256), 0, s, V);
            else hipLaunchKernelGGL((k_assemble_rows<4, false>), g, dim3(256), 0, s, V);
        } else if (fuse_init_solve) hipLaunchKernelGGL((k_assemble<true, 4, false>), g, dim3(256), 0, s, V);
        else if (reg) hipLaunchKernelGGL((k_assemble<false, 4, true>), g, dim3(256), 0, s, V);
        else hipLaunchKernelGGL((k_assemble<false, 4, false>), g, dim3(256), 0, s, V);
    }
}

}  // namespace vba
