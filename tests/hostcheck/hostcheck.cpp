// Host build of vinsat_amd/csrc/vba_math.h for the CPU test-suite (formula checks without a GPU).
// Test infrastructure only: nothing in the product loads this.
#include "../../vinsat_amd/csrc/vba_math.h"
#include <cstdint>
using namespace vba;

extern "C" {

void hc_project(int64_t m, const double* states, const double* K, const double* xyz, const int64_t* ii,
                double* est, double* J) {
    for (int64_t k = 0; k < m; ++k) {
        PoseCam pc;
        pose_camera(states + 10 * ii[k], K + 4 * ii[k], pc);
        double cam[3], d;
        project(pc, xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], est[2 * k], est[2 * k + 1], cam, d);
        if (J) project_jacobian(pc, cam, d, J + 12 * k);
    }
}

void hc_weights(int64_t m, const double* r, double c, double alpha, double* w) {
    RobustParams rp;
    rp.c = c; rp.inv_c = 1.0 / c; rp.inv_c2 = 1.0 / (c * c); rp.am2 = fabs(alpha - 2); rp.inv_am2 = 1.0 / rp.am2; rp.expo = alpha / 2 - 1; rp.alpha_is_2 = alpha == 2.0; rp.expo_is_mhalf = rp.expo == -0.5;
    for (int64_t k = 0; k < m; ++k) w[k] = robust_weight_raw(rp, r[2 * k], r[2 * k + 1]);
}

void hc_orbit(int n, const double* states, const int64_t* steps, double* xhat, double* Phi) {
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < 6; ++c) {
            double x[6] = {states[10 * i], states[10 * i + 1], states[10 * i + 2], states[10 * i + 7], states[10 * i + 8], states[10 * i + 9]};
            double t[6] = {0, 0, 0, 0, 0, 0};
            t[c] = 1.0;
            for (int64_t s = 0; s < steps[i]; ++s) rk4_step<true>(x, t);
            for (int r = 0; r < 6; ++r) Phi[36 * i + 6 * r + c] = t[r];
            if (c == 0) for (int r = 0; r < 6; ++r) xhat[6 * i + r] = x[r];
        }
    }
}

void hc_orbit_hop(int n, const double* x6, const int64_t* steps, double* xhat, double* Phi) {
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 6; ++c) {
            double x[6], t[6] = {0, 0, 0, 0, 0, 0};
            for (int r = 0; r < 6; ++r) x[r] = x6[6 * i + r];
            t[c] = 1.0;
            propagate_gap<true>(x, t, (int)steps[i], 1);
            for (int r = 0; r < 6; ++r) Phi[36 * i + 6 * r + c] = t[r];
            if (c == 0) for (int r = 0; r < 6; ++r) xhat[6 * i + r] = x[r];
        }
}

void hc_orbit_fwd(int n, const double* states, const int64_t* steps, double* xhat) {
    for (int i = 0; i < n; ++i) {
        double x[6] = {states[10 * i], states[10 * i + 1], states[10 * i + 2], states[10 * i + 7], states[10 * i + 8], states[10 * i + 9]};
        for (int64_t s = 0; s < steps[i]; ++s) rk4_step<false>(x, nullptr);
        for (int r = 0; r < 6; ++r) xhat[6 * i + r] = x[r];
    }
}

void hc_attitude(int n, const double* states, const double* cumrot, double* f, double* qgrad, double* Hd, double* Hu, double* Hl) {
    for (int i = 0; i < n; ++i) {
        const double* qp = i > 0 ? states + 10 * (i - 1) + 3 : nullptr;
        const double* cp = i > 0 ? cumrot + 4 * (i - 1) : nullptr;
        const double* qn = i < n - 1 ? states + 10 * (i + 1) + 3 : nullptr;
        attitude_term(qp, cp, states + 10 * i + 3, cumrot + 4 * i, qn, f[i], qgrad + 3 * i, Hd + 9 * i, Hu + 9 * i, Hl + 9 * i);
    }
}

void hc_assemble(int n, const double* Hraw, const double* braw, double inv_wmax, double sigma, const double* Phi,
                 const double* rorb, const double* qgrad, const double* Hd, const double* Hu, const double* Hl,
                 double* bands, double* rhs) {
    for (int i = 0; i < n; ++i) {
        AsmRow R{};
        R.Hraw = Hraw + 21 * i; R.braw = braw + 6 * i; R.inv_wmax = inv_wmax; R.sigma = sigma;
        const bool dyn = sigma != 0.0;
        R.Phi_i = (dyn && i < n - 1) ? Phi + 36 * i : nullptr;
        R.Phi_im1 = (dyn && i > 0) ? Phi + 36 * (i - 1) : nullptr;
        R.rorb_i = (dyn && i < n - 1) ? rorb + 6 * i : nullptr;
        R.rorb_im1 = (dyn && i > 0) ? rorb + 6 * (i - 1) : nullptr;
        R.qgrad = qgrad + 3 * i; R.Hd = Hd + 9 * i; R.Hu = Hu + 9 * i; R.Hl = Hl + 9 * i;
        for (int w = 0; w < 3; ++w)
            for (int a = 0; a < 9; ++a)
                for (int b = 0; b < 9; ++b) bands[((i * 3 + w) * 9 + a) * 9 + b] = band_entry(R, w, a, b);
        for (int a = 0; a < 9; ++a) rhs[9 * i + a] = rhs_entry(R, a);
    }
}

void hc_retract(int n, const double* states, const double* dpose, double* out) {
    for (int i = 0; i < n; ++i) retract(states + 10 * i, dpose + 9 * i, out + 10 * i);
}

}
