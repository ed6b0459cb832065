// vba_prep.hip -- the per-row part of the driver's data preparation on the device (reference od_pipe.py:924-930).
//
// Before its first BA() call the reference's driver turns every detection row [frame, lon, lat, u, v, conf] into an inertial landmark
// position (read_detections / latlon_to_eci, BA_utils.py:1221-1251, 1172-1218), reprojects it at the ground-truth pose of its frame
// (landmark_project, BA_utils.py:30-43) and masks outliers (od_pipe.py:930).  On the host that is ~9 ms of array code for a
// 50 000-row sequence against ~1 ms for the 20 BA() calls that follow; here it is one thread per row: 112 B of traffic and six
// transcendentals per row.  The models and constants are those of vinsat_amd/frames.py (= the reference's); sin / cos come from
// the device library instead of the host's, so positions agree with the host path to rounding (~1e-16 relative), not bit for bit
// -- the host path (vinsat_amd.od_pipe.prepare_window without a device) stays the one that is compared bit for bit with the
// reference's arrays.
#include <mutex>

#include <hip/hip_runtime.h>

#include "../../include/vinsat_ba.h"
#include "vba_math.h"

namespace vba {

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kDeg = kPi / 180.0;                  // numpy.deg2rad multiplies by this constant
constexpr double kThetaG0Deg = 280.16;                // BA_utils.py:1172-1218
constexpr double kOmegaEarthDegPerSec = 360.0 / 86164.100352;
constexpr double kAEarth = 6378.137, kBEarth = 6356.752;    // km, BA_utils.py:1178-1180

__global__ __launch_bounds__(256) void k_prepare_rows(int64_t M, const double* __restrict__ det /*[M,6]*/, const int64_t* __restrict__ ii, int T,
                                                      const double* __restrict__ pos_gt /*[T,3]*/, const double* __restrict__ rot_gt /*[T,9]*/,
                                                      double fx, double fy, double cx, double cy, double ecc2, double axis_ratio_sq,
                                                      double* __restrict__ xyz, double* __restrict__ proj, unsigned char* __restrict__ mask) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    const double frame = det[6 * k], lon = det[6 * k + 1] * kDeg, lat = det[6 * k + 2] * kDeg;
    const double u = det[6 * k + 3], v = det[6 * k + 4], conf = det[6 * k + 5];
    // geodetic (altitude 0) -> Earth-fixed, km
    const double sl = sin(lat), cl = cos(lat);
    const double n_phi = kAEarth / sqrt(1.0 - ecc2 * (sl * sl));
    const double ring = n_phi * cl;
    const double xe = ring * cos(lon), ye = ring * sin(lon), ze = (axis_ratio_sq * n_phi) * sl;
    // Earth-fixed -> inertial at the frame's second
    const double th = (kThetaG0Deg + kOmegaEarthDegPerSec * frame) * kDeg;
    const double c = cos(th), s = sin(th);
    const double X = xe * c - ye * s, Y = xe * s + ye * c, Z = ze;
    xyz[3 * k] = X; xyz[3 * k + 1] = Y; xyz[3 * k + 2] = Z;
    // reprojection at the ground-truth pose of the row's frame: p_c = R^T (X - t)
    int64_t i = ii[k];
    i = i < 0 ? 0 : (i >= T ? T - 1 : i);
    const double* R = rot_gt + 9 * i;
    const double dx = X - pos_gt[3 * i], dy = Y - pos_gt[3 * i + 1], dz = Z - pos_gt[3 * i + 2];
    const double px = R[0] * dx + R[3] * dy + R[6] * dz;
    const double py = R[1] * dx + R[4] * dy + R[7] * dz;
    const double pz = fmax(R[2] * dx + R[5] * dy + R[8] * dz, kZMin);
    const double pu = fx * px / pz + cx, pv = fy * py / pz + cy;
    proj[2 * k] = pu; proj[2 * k + 1] = pv;
    const double eu = pu - u, ev = pv - v;
    mask[k] = (pu > 0.0 && pv > 0.0 && pu < 4700.0 && pv < 2600.0 && sqrt(eu * eu + ev * ev) < 1000.0 && conf > 0.8) ? 1 : 0;    // od_pipe.py:930
}

// one grow-only workspace per process (the driver prepares one sequence at a time per device)
struct Workspace {
    std::mutex m;
    int device = -1;
    char* base = nullptr;
    size_t size = 0;
    hipStream_t stream = nullptr;
} g_ws;

}  // namespace

}  // namespace vba

extern "C" int vba_prepare_rows(int device, int64_t M, const double* det, const int64_t* ii, int T, const double* pos_gt, const double* rot_gt,
                                const double* intrinsics, double* xyz, double* proj, unsigned char* mask) {
    using namespace vba;
    if (M < 0 || T < 1 || !det || !ii || !pos_gt || !rot_gt || !intrinsics || !xyz || !proj || !mask) return VBA_EINVAL;
    if (M == 0) return VBA_OK;
    std::lock_guard<std::mutex> lk(g_ws.m);
    if (hipSetDevice(device) != hipSuccess) return VBA_ENODEV;
    const size_t rows = (size_t)M;
    const size_t need = rows * (6 + 1 + 3 + 2) * 8 + rows + (size_t)T * 12 * 8 + 4096;
    if (g_ws.device != device || g_ws.size < need) {
        if (g_ws.base) (void)hipFree(g_ws.base);
        g_ws.base = nullptr;
        g_ws.size = 0;
        if (!g_ws.stream || g_ws.device != device) {
            if (g_ws.stream) (void)hipStreamDestroy(g_ws.stream);
            if (hipStreamCreateWithFlags(&g_ws.stream, hipStreamNonBlocking) != hipSuccess) { g_ws.stream = nullptr; return VBA_EHIP; }
        }
        if (hipMalloc(&g_ws.base, need + need / 2) != hipSuccess) return VBA_ENOMEM;
        g_ws.size = need + need / 2;
        g_ws.device = device;
    }
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    char* p = g_ws.base;
    double* d_det = reinterpret_cast<double*>(p); p += al(rows * 48);
    int64_t* d_ii = reinterpret_cast<int64_t*>(p); p += al(rows * 8);
    double* d_pos = reinterpret_cast<double*>(p); p += al((size_t)T * 24);
    double* d_rot = reinterpret_cast<double*>(p); p += al((size_t)T * 72);
    double* d_xyz = reinterpret_cast<double*>(p); p += al(rows * 24);
    double* d_proj = reinterpret_cast<double*>(p); p += al(rows * 16);
    unsigned char* d_mask = reinterpret_cast<unsigned char*>(p);
    hipStream_t s = g_ws.stream;
    const double ecc = sqrt(1.0 - (kBEarth * kBEarth) / (kAEarth * kAEarth));      // the eccentricity enters squared AFTER its square root was taken (frames.py)
    bool ok = hipMemcpyAsync(d_det, det, rows * 48, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_ii, ii, rows * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_pos, pos_gt, (size_t)T * 24, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_rot, rot_gt, (size_t)T * 72, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_prepare_rows, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, M, d_det, d_ii, T, d_pos, d_rot, intrinsics[0], intrinsics[1],
                           intrinsics[2], intrinsics[3], ecc * ecc, (kBEarth * kBEarth) / (kAEarth * kAEarth), d_xyz, d_proj, d_mask);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(xyz, d_xyz, rows * 24, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(proj, d_proj, rows * 16, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(mask, d_mask, rows, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipStreamSynchronize(s) == hipSuccess;
    }
    if (!ok) { (void)hipGetLastError(); return VBA_EHIP; }
    return VBA_OK;
}
