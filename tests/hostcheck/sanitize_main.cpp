// AddressSanitizer / UBSan driver of the host build of vinsat_amd/csrc/vba_math.h (CPU only, test infrastructure).
// Every entry of hostcheck.cpp runs over exactly-sized heap buffers of a small synthetic window that includes the edge
// cases of the arithmetic (depth below the clamp, zero residuals at alpha = 2, a zero step in the retraction, the first /
// last pose of the attitude chain, hop and one-second integrators, the BA_reg prior); any out-of-bounds access, use of
// an uninitialised stack slot that ASan can see, signed overflow, bad shift or misaligned access aborts with a report.
#include "hostcheck.cpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

static double urand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (double)(s >> 8) / (double)(1u << 24);
}

int main() {
    unsigned seed = 12345u;
    const int n = 7;
    const int64_t m = 41;
    std::vector<double> states((size_t)n * 10), K((size_t)n * 4), cumrot((size_t)n * 4), xyz((size_t)m * 3);
    std::vector<int64_t> ii((size_t)m), steps((size_t)n);
    for (int i = 0; i < n; ++i) {
        double* s = &states[(size_t)i * 10];
        s[0] = 6978.0 + 10.0 * urand(seed); s[1] = 50.0 * urand(seed); s[2] = 40.0 * i;
        double q[4] = {urand(seed) - 0.5, urand(seed) - 0.5, urand(seed) - 0.5, 1.0 + urand(seed)};
        const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; ++k) s[3 + k] = q[k] / nq;
        s[7] = 0.1 * urand(seed); s[8] = 0.2; s[9] = 7.5;
        K[(size_t)i * 4] = 3547.85; K[(size_t)i * 4 + 1] = 3547.85; K[(size_t)i * 4 + 2] = 2304.0; K[(size_t)i * 4 + 3] = 1296.0;
        cumrot[(size_t)i * 4] = 1e-3 * urand(seed); cumrot[(size_t)i * 4 + 1] = 2e-3; cumrot[(size_t)i * 4 + 2] = 0.0;
        cumrot[(size_t)i * 4 + 3] = 1.0;
        steps[(size_t)i] = i == 2 ? 237 : (i == 4 ? 100 : 1 + i);      // above, at and below the 100 s hop
    }
    for (int64_t k = 0; k < m; ++k) {
        ii[(size_t)k] = k % n;
        const double* s = &states[(size_t)(k % n) * 10];
        // most points in front of the camera, some behind it or closer than the 0.1 km clamp
        const double depth = (k % 9 == 0) ? 0.05 : ((k % 13 == 0) ? -300.0 : 500.0 + 100.0 * urand(seed));
        xyz[(size_t)k * 3] = s[0] + 30.0 * (urand(seed) - 0.5);
        xyz[(size_t)k * 3 + 1] = s[1] + 30.0 * (urand(seed) - 0.5);
        xyz[(size_t)k * 3 + 2] = s[2] + depth;
    }
    std::vector<double> est((size_t)m * 2), J((size_t)m * 12);
    hc_project(m, states.data(), K.data(), xyz.data(), ii.data(), est.data(), J.data());
    hc_project(m, states.data(), K.data(), xyz.data(), ii.data(), est.data(), nullptr);

    std::vector<double> r((size_t)m * 2), w((size_t)m);
    for (size_t k = 0; k < r.size(); ++k) r[k] = 6.0 * (urand(seed) - 0.5);
    r[0] = r[1] = 0.0;
    const double alphas[] = {2.0, 1.6, 1.2, 1.0};
    for (double a : alphas) hc_weights(m, r.data(), 0.7, a, w.data());

    std::vector<double> xhat((size_t)n * 6), Phi((size_t)n * 36), x6((size_t)n * 6);
    hc_orbit(n, states.data(), steps.data(), xhat.data(), Phi.data());
    hc_orbit_fwd(n, states.data(), steps.data(), xhat.data());
    for (int i = 0; i < n; ++i) {
        const double* s = &states[(size_t)i * 10];
        const double v[6] = {s[0], s[1], s[2], s[7], s[8], s[9]};
        for (int c = 0; c < 6; ++c) x6[(size_t)i * 6 + c] = v[c];
    }
    hc_orbit_hop(n, x6.data(), steps.data(), xhat.data(), Phi.data());

    std::vector<double> f((size_t)n), qgrad((size_t)n * 3), Hd((size_t)n * 9), Hu((size_t)n * 9), Hl((size_t)n * 9);
    hc_attitude(n, states.data(), cumrot.data(), f.data(), qgrad.data(), Hd.data(), Hu.data(), Hl.data());

    std::vector<double> Hraw((size_t)n * 21), braw((size_t)n * 6), rorb((size_t)n * 6), bands((size_t)n * 243), rhs((size_t)n * 9);
    for (double& v : Hraw) v = urand(seed);
    for (double& v : braw) v = urand(seed) - 0.5;
    for (double& v : rorb) v = urand(seed) - 0.5;
    const double sigmas[] = {0.0, 1e4};
    for (double sg : sigmas)
        hc_assemble(n, Hraw.data(), braw.data(), 0.5, sg, Phi.data(), rorb.data(), qgrad.data(), Hd.data(), Hu.data(), Hl.data(),
                    bands.data(), rhs.data());

    std::vector<double> dpose((size_t)n * 9, 0.0), out((size_t)n * 10);
    for (int i = 1; i < n; ++i)
        for (int c = 0; c < 9; ++c) dpose[(size_t)i * 9 + c] = 1e-2 * (urand(seed) - 0.5);       // pose 0: zero step (identity branch)
    hc_retract(n, states.data(), dpose.data(), out.data());

    // BA_reg prior helpers
    std::vector<double> H(36), xp(6), r6(6);
    for (double& v : H) v = urand(seed);
    for (double& v : xp) v = urand(seed);
    prior_residual(H.data(), xp.data(), states.data(), r6.data());
    double acc = 0.0;
    for (int a = 0; a < 6; ++a) {
        acc += prior_htr(H.data(), r6.data(), a);
        for (int b = 0; b < 6; ++b) acc += prior_hth(H.data(), a, b);
    }
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b)
            if (sym6(a, b) < 0 || sym6(a, b) > 20) return 2;

    double chk = acc;
    for (double v : est) chk += v;
    for (double v : out) chk += v;
    for (double v : bands) chk += v;
    std::printf("sanitize_main ok %.17g\n", chk);
    return chk == chk ? 0 : 3;
}
