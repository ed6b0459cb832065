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
        for (int q = 0; q < steps; ++q) rk4_step<true>(x, t);
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

// One block per pose row: 3 x 81 band entries + 9 right-hand-side entries, one thread each (BA_filtering.py:40-48).
__global__ __launch_bounds__(256) void k_assemble(DevView V) {
    const int w = blockIdx.y;
    const int n = V.n[w];
    const int i = blockIdx.x;
    if (i >= n) return;
    const StepParams& prm = V.prm;
    const size_t pb = (size_t)w * V.n_max + i;
    const bool dyn = !prm.initialize;
    AsmRow R;
    R.Hraw = V.Hraw + pb * 21;
    R.braw = V.braw + pb * 6;
    R.inv_wmax = 1.0 / bits_f64(V.sc[w].wmax_bits);
    R.sigma = dyn ? prm.sigma : 0.0;
    R.Phi_i = (dyn && i < n - 1) ? V.Phi + pb * 36 : nullptr;
    R.Phi_im1 = (dyn && i > 0) ? V.Phi + (pb - 1) * 36 : nullptr;
    R.rorb_i = (dyn && i < n - 1) ? V.rorb + pb * 6 : nullptr;
    R.rorb_im1 = (dyn && i > 0) ? V.rorb + (pb - 1) * 6 : nullptr;
    R.qgrad = V.qgrad + pb * 3;
    R.Hd = V.Hd + pb * 9;
    R.Hu = V.Hu + pb * 9;
    R.Hl = V.Hl + pb * 9;
    const int t = threadIdx.x;
    if (t < 243) {
        const int which = t / 81, e = t % 81;
        V.bands[pb * 243 + t] = band_entry(R, which, e / 9, e % 9);
    } else if (t < 252) {
        V.rhs[pb * 9 + (t - 243)] = rhs_entry(R, t - 243);
    }
}

void launch_dynamics(const DevView& V, hipStream_t s) {
    const int nb = (V.n_max * kDynLanes + 255) / 256;
    hipLaunchKernelGGL(k_dynamics, dim3(nb, V.W), dim3(256), 0, s, V);
}

void launch_assemble(const DevView& V, hipStream_t s) {
    hipLaunchKernelGGL(k_assemble, dim3(V.n_max, V.W), dim3(256), 0, s, V);
}

}  // namespace vba
