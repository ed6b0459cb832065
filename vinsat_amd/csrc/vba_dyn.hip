// vba_dyn.hip -- pose-chain kernels: orbit-dynamics factor with its state-transition matrix (A4), attitude
// Newton term (A5) and block-tridiagonal assembly (A6).
//
// The reference differentiates the RK4 chain with reverse-mode autograd into a dense [6(n-1), 9n] Jacobian
// (BA_utils.py:506); here every pose carries its six tangent vectors forward through the same RK4 steps
// (8 lanes per pose: lanes 0-5 one tangent each, lane 6 the attitude term), so only the 6x6 block that is
// actually non-zero is ever produced.
#include "vba_device.h"
#include "vba_launch.h"

namespace vba {

__global__ __launch_bounds__(256) void k_dynamics(DevView V) {
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int n = V.n[w];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int i = gid / kDynLanes, c = gid % kDynLanes;
    if (i >= n) return;
    const size_t pb = (size_t)w * V.n_max + i;
    const double* st = V.states + pb * 10;
    if (c < 6) {
        if (i >= n - 1) return;     // the last pose's propagation is discarded by the reference (BA_utils.py:476)
        double x[6] = {st[0], st[1], st[2], st[7], st[8], st[9]};
        double t[6] = {0, 0, 0, 0, 0, 0};
        t[c] = 1.0;
        const int steps = V.steps[pb];
        propagate_gap<true>(x, t, steps, V.hop);
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
        }
    } else if (c == 6) {
        const double* qp = i > 0 ? st - 10 + 3 : nullptr;
        const double* cp = i > 0 ? V.cumrot + (pb - 1) * 4 : nullptr;
        const double* qn = i < n - 1 ? st + 10 + 3 : nullptr;
        double f, qg[3], Hd[9], Hu[9], Hl[9];
        attitude_term(qp, cp, st + 3, V.cumrot + pb * 4, qn, f, qg, Hd, Hu, Hl);
        V.fatt[pb] = f;
#pragma unroll
        for (int k = 0; k < 3; ++k) V.qgrad[pb * 3 + k] = qg[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            V.Hd[pb * 9 + k] = Hd[k];
            V.Hu[pb * 9 + k] = Hu[k];
            V.Hl[pb * 9 + k] = Hl[k];
        }
    }
}

// Block-tridiagonal assembly (BA_filtering.py:40-48): 3 x 81 band entries + 9 right-hand-side entries per pose.
// A block of 256 threads takes kAsmPoses consecutive poses: the per-pose inputs (141 doubles each, plus the
// transition matrix of the pose in front) are staged once in LDS with coalesced loads, then every thread forms
// entries from LDS and the block writes its 252 * kAsmPoses outputs contiguously.
constexpr int kAsmPoses = 4;
constexpr int kAsmIn = 21 + 6 + 36 + 6 + 3 + 27;     // Hraw, braw, Phi, rorb, qgrad, Hd|Hu|Hl

__global__ __launch_bounds__(256) void k_assemble(DevView V) {
    __shared__ double in[(kAsmPoses + 1) * kAsmIn];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int n = V.n[w];
    const int i0 = blockIdx.x * kAsmPoses;
    if (i0 >= n) return;
    const StepParams& prm = V.prm;
    const size_t sb = (size_t)w * V.n_max;
    const bool dyn = !prm.initialize;
    // slot 0 = pose i0-1 (only Phi and rorb are used), slots 1..kAsmPoses = poses i0 ..
    for (int e = threadIdx.x; e < (kAsmPoses + 1) * kAsmIn; e += 256) {
        const int slot = e / kAsmIn, q = e % kAsmIn;
        const int i = i0 - 1 + slot;
        double v = 0.0;
        if (i >= 0 && i < n) {
            const size_t pb = sb + i;
            if (q < 21) v = V.Hraw[pb * 21 + q];
            else if (q < 27) v = V.braw[pb * 6 + (q - 21)];
            else if (dyn) {
                if (q < 63) v = V.Phi[pb * 36 + (q - 27)];
                else if (q < 69) v = V.rorb[pb * 6 + (q - 63)];
                else if (q < 72) v = V.qgrad[pb * 3 + (q - 69)];
                else if (q < 81) v = V.Hd[pb * 9 + (q - 72)];
                else if (q < 90) v = V.Hu[pb * 9 + (q - 81)];
                else v = V.Hl[pb * 9 + (q - 90)];
            }
        }
        in[e] = v;
    }
    __syncthreads();
    const double inv_wmax = 1.0 / bits_f64(V.sc[w].wmax_bits);
    const int cnt = min(kAsmPoses, n - i0);
    for (int e = threadIdx.x; e < cnt * 252; e += 256) {
        const int p = e / 252, t = e % 252;
        const int i = i0 + p;
        const double* me = in + (p + 1) * kAsmIn;
        const double* pv = in + p * kAsmIn;
        AsmRow R;
        R.Hraw = me;
        R.braw = me + 21;
        R.inv_wmax = inv_wmax;
        R.sigma = dyn ? prm.sigma : 0.0;
        R.Phi_i = (dyn && i < n - 1) ? me + 27 : nullptr;
        R.Phi_im1 = (dyn && i > 0) ? pv + 27 : nullptr;
        R.rorb_i = (dyn && i < n - 1) ? me + 63 : nullptr;
        R.rorb_im1 = (dyn && i > 0) ? pv + 63 : nullptr;
        R.qgrad = me + 69;
        R.Hd = me + 72;
        R.Hu = me + 81;
        R.Hl = me + 90;
        if (t < 243) V.bands[(sb + i) * 243 + t] = band_entry(R, t / 81, (t % 81) / 9, t % 9);
        else V.rhs[(sb + i) * 9 + (t - 243)] = rhs_entry(R, t - 243);
    }
}

void launch_dynamics(const DevView& V, hipStream_t s) {
    const int nb = (V.n_max * kDynLanes + 255) / 256;
    hipLaunchKernelGGL(k_dynamics, dim3(nb, V.W), dim3(256), 0, s, V);
}

void launch_assemble(const DevView& V, hipStream_t s) {
    hipLaunchKernelGGL(k_assemble, dim3((V.n_max + kAsmPoses - 1) / kAsmPoses, V.W), dim3(256), 0, s, V);
}

}  // namespace vba
