// vba_host.hip -- host-side helpers of the driver around BA() (the reference's streaming_version,
// od_pipe.py:911-1062).  No device is involved: these are the SERIAL recurrences of the driver's data preparation, which as
// Python loops cost more than the BA calls they sit between (a 935 s gap: 126 ms of interpreted RK4 steps in front of ~4 ms of
// BA calls) -- the per-second attitude increments accumulated over each gap (od_pipe.py:945-961, precompute_cum_rotations
// BA_utils.py:278-288) and the dead reckoning across the gap between two batches (propagate_dynamics_init, BA_utils.py:114-129).
#include <cstring>

#include <hip/hip_runtime.h>

#include "../../include/vinsat_ba.h"
#include "vba_math.h"

namespace {

// Hamilton product, scalar last, every product and sum rounded on its own in the order of the NumPy expression
// (vinsat_amd/quat.py:qmul = BA_utils.py:992-1000): the bits of the array code it replaces.
inline void qmul_exact(const double* a, const double* b, double* o) {
#pragma clang fp contract(off)
    const double x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    const double x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    const double w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    const double x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    const double y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    const double z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

}  // namespace

extern "C" {

int vba_host_orbit_chain(const double* x0, int steps, double* out) {
    if (!x0 || !out || steps < 0) return VBA_EINVAL;
    double x[6];
    std::memcpy(x, x0, sizeof(x));
    for (int k = 0; k < steps; ++k) {
        vba::rk4_step<false>(x, nullptr, 1.0);
        std::memcpy(out + (size_t)k * 6, x, sizeof(x));
    }
    return VBA_OK;
}

int vba_host_quat_chain(const double* q0, const double* r, int K, double* out) {
    if (!r || !out || K < 0) return VBA_EINVAL;
    double q[4];
    int k = 0;
    if (q0) std::memcpy(q, q0, sizeof(q));
    else if (K > 0) { std::memcpy(q, r, sizeof(q)); std::memcpy(out, q, sizeof(q)); k = 1; }
    for (; k < K; ++k) {
        double t[4];
        qmul_exact(q, r + (size_t)k * 4, t);
        std::memcpy(q, t, sizeof(q));
        std::memcpy(out + (size_t)k * 4, q, sizeof(q));
    }
    return VBA_OK;
}

int vba_host_gap_rotations(const double* r, int64_t N, const int64_t* time_idx, int T, double* cum) {
    if (!r || !time_idx || !cum || T < 1) return VBA_EINVAL;
    for (int i = 0; i < T; ++i) {
        double* c = cum + (size_t)i * 4;
        c[0] = c[1] = c[2] = 0.0;
        c[3] = 1.0;
        if (i == T - 1) break;
        const int64_t t0 = time_idx[i], gap = time_idx[i + 1] - t0;
        if (gap < 1 || t0 < 0 || t0 + gap > N) return VBA_EINVAL;
        double q[4];
        std::memcpy(q, r + (size_t)t0 * 4, sizeof(q));
        for (int64_t j = 1; j < gap; ++j) {
            double t[4];
            qmul_exact(q, r + (size_t)(t0 + j) * 4, t);
            std::memcpy(q, t, sizeof(q));
        }
        std::memcpy(c, q, sizeof(q));
    }
    return VBA_OK;
}

}  // extern "C"
