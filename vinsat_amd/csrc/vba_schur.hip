// vba_schur.hip -- free-landmark bundle adjustment with a Schur-complement solve (ADD-ON, PARITY UNPINNED).
//
// The reference keeps its landmarks fixed (estimation/BA/BA_filtering.py:32-37): every observation touches only the
// 6x6 block of its own pose and there is nothing to marginalise.  This file is the variant BASELINE.json's north_star
// describes on top of that: the landmarks become unknowns (3 each, held by a catalogue prior N(X0, sigma^2 I)), the
// normal equations
//        [ B   E ] [dc]   [v]        B: 6x6 per pose,   C: 3x3 per landmark,   E: 6x3 per observation
//        [ E^T C ] [dl] = [w]
// are reduced to the cameras,  S = B - E C^-1 E^T,  g = v - E C^-1 w,  the dense reduced camera system S dc = g is
// factorised on the matrix cores (blocked Cholesky, v_mfma_f64_16x16x4), and the landmarks follow by back substitution
// dl = C^-1 (w - E^T dc).  It has NO counterpart in the reference; it is checked against this repository's own CPU
// restatement (oracle/schur_oracle.py) only and never runs inside vba_iterate / BA().
//
// Reprojection and its pose Jacobian are the reference's (vba_math.h: BA_utils.py:30-49); the landmark Jacobian is
// d uv / d X = A R^T = -(translation part of the pose Jacobian).  Weights are the observation confidences (the
// reference's alpha = 2 case, BA_filtering.py:22-25, where the robust weight is constant).
//
// Kernels (all reductions in a fixed order -- no float atomics):
//   k_lm_blocks     thread per landmark: C_l, w_l over its rows, C_l^-1, then E_k and Y_k = E_k C_l^-1 per row
//   k_pose_blocks   16 lanes per pose: B_i, v_i over its rows, g_i = v_i - sum Y_k w_l; diagonal block of S
//   k_pair_blocks   wave per (i, j) block of S: S_ij -= sum over the rows pairs sharing a landmark of Y_k E_k'^T
//   k_potrf64       64x64 diagonal block: Cholesky factor and its inverse (one workgroup, LDS)
//   k_gemm_abt      64x64x64 tiles on the matrix cores: panel = A inv(L)^T (MODE 0), trailing S_IJ -= L_I L_J^T (MODE 1)
//   k_trsv_step     one tile step of the forward / backward substitution (stored inverse diagonal factors), a launch per step
//   k_lm_update     dl, new landmarks, new poses (retraction as BA_filtering.py:56-60)
//   k_cost          sum w |r|^2 + prior, block partials
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/vinsat_ba.h"
#include "vba_math.h"

namespace {

using namespace vba;

typedef double vf4 __attribute__((ext_vector_type(4)));

constexpr int kT = 64;              // tile of the blocked Cholesky
constexpr int kPanel = 4;           // tile columns per panel: the trailing update runs with K = kPanel * kT = 256 (round 4)

struct SchurView {
    int n, L;
    int64_t m;
    int N, Npad, nb;                // reduced system: N = 6 n, padded to a multiple of kT, nb tiles per side
    // rows sorted by landmark: CSR lm_ptr[L+1]
    const int* lm_ptr;
    const int* row_pose;            // [m]
    const int* row_lm;              // [m]
    const double *row_u, *row_v, *row_w;    // [m] measurement, weight
    // rows of a pose: CSR pose_ptr[n+1] -> pose_rows[m] (indices into the landmark-sorted rows)
    const int* pose_ptr;
    const int* pose_rows;
    // blocks (i >= j) of S that receive pair products: CSR blk_ptr[nblk+1] -> (pair_k, pair_k2); blk_i, blk_j
    int nblk;
    const int *blk_i, *blk_j, *blk_ptr, *pair_k, *pair_k2;
    const double* intr;             // [n][4]
    const double* X0;               // [L][3] catalogue positions
    double inv_sigma2;              // 1 / sigma_prior^2
    double lamda;
    // state
    const double* states;           // [n][10]
    const double* X;                // [L][3]
    double* states_new;
    double* X_new;
    // work
    double* Cinv;                   // [L][6] symmetric inverse (00,01,02,11,12,22)
    double* wl;                     // [L][3]
    double* E;                      // [m][18] row major 6x3
    double* Y;                      // [m][18]
    double* S;                      // [Npad][Npad] row major, lower triangle used
    double* g;                      // [Npad] right-hand side, then the step of the poses
    double* ybuf;                   // [Npad] intermediate of the two substitutions
    double* invL;                   // [nb][kT*kT] inverse of the diagonal Cholesky factors
    double* dl;                     // [L][3]
    double* part;                   // cost partials
    int npart;
};

// row k of the landmark-sorted rows: residual, A (d uv / d p_c), camera point, for pose / landmark state given
struct RowGeom {
    double ru, rv, a00, a02, a11, a12, cam[3];
    PoseCam pc;
};

__device__ __forceinline__ void row_geometry(const SchurView& V, const double* states, const double* X, int k, RowGeom& q) {
    const int i = V.row_pose[k], l = V.row_lm[k];
    pose_camera(states + (size_t)i * 10, V.intr + (size_t)i * 4, q.pc);
    double u, v, d;
    project(q.pc, X[3 * l], X[3 * l + 1], X[3 * l + 2], u, v, q.cam, d);
    q.ru = V.row_u[k] - u;
    q.rv = V.row_v[k] - v;
    const double live = q.cam[2] > kZMin ? 1.0 : 0.0;
    q.a00 = q.pc.fx * d;
    q.a11 = q.pc.fy * d;
    q.a02 = -(q.a00 * (q.cam[0] * d * live));
    q.a12 = -(q.a11 * (q.cam[1] * d * live));
}

// J_l = A R^T (2x3): row u -> jl[0..2], row v -> jl[3..5]
__device__ __forceinline__ void landmark_jacobian(const RowGeom& q, double* jl) {
    const double* R = q.pc.R;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        jl[c] = q.a00 * R[3 * c] + q.a02 * R[3 * c + 2];
        jl[3 + c] = q.a11 * R[3 * c + 1] + q.a12 * R[3 * c + 2];
    }
}

// J_c = [-J_l | 2 A hat(p_c)] (2x6): row u -> jc[0..5], row v -> jc[6..11]
__device__ __forceinline__ void pose_jacobian(const RowGeom& q, const double* jl, double* jc) {
    const double x = q.cam[0], y = q.cam[1], z = q.cam[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) { jc[c] = -jl[c]; jc[6 + c] = -jl[3 + c]; }
    jc[3] = 2.0 * (-q.a02 * y);
    jc[4] = 2.0 * (-q.a00 * z + q.a02 * x);
    jc[5] = 2.0 * (q.a00 * y);
    jc[9] = 2.0 * (q.a11 * z - q.a12 * y);
    jc[10] = 2.0 * (q.a12 * x);
    jc[11] = 2.0 * (-q.a11 * x);
}

// ------------------------------------------------------------------------------------------------ landmark side
__global__ __launch_bounds__(256) void k_lm_blocks(SchurView V) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= V.L) return;
    const int beg = V.lm_ptr[l], end = V.lm_ptr[l + 1];
    double C[6] = {0, 0, 0, 0, 0, 0}, w3[3] = {0, 0, 0};
    for (int k = beg; k < end; ++k) {
        RowGeom q;
        row_geometry(V, V.states, V.X, k, q);
        double jl[6];
        landmark_jacobian(q, jl);
        const double w = V.row_w[k];
        int e = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = a; b < 3; ++b) { C[e] = fma(w * jl[a], jl[b], fma(w * jl[3 + a], jl[3 + b], C[e])); ++e; }
            w3[a] = fma(w * jl[a], q.ru, fma(w * jl[3 + a], q.rv, w3[a]));
        }
    }
    const double dg = V.inv_sigma2 + V.lamda;
    C[0] += dg; C[3] += dg; C[5] += dg;
#pragma unroll
    for (int a = 0; a < 3; ++a) w3[a] -= V.inv_sigma2 * (V.X[3 * l + a] - V.X0[3 * l + a]);
    // inverse of the symmetric 3x3 by cofactors
    const double c00 = C[0], c01 = C[1], c02 = C[2], c11 = C[3], c12 = C[4], c22 = C[5];
    const double m00 = c11 * c22 - c12 * c12, m01 = c02 * c12 - c01 * c22, m02 = c01 * c12 - c02 * c11;
    const double det = c00 * m00 + c01 * m01 + c02 * m02;
    const double id = 1.0 / det;
    double Ci[6] = {m00 * id, m01 * id, m02 * id, (c00 * c22 - c02 * c02) * id, (c01 * c02 - c00 * c12) * id, (c00 * c11 - c01 * c01) * id};
#pragma unroll
    for (int e = 0; e < 6; ++e) V.Cinv[(size_t)l * 6 + e] = Ci[e];
#pragma unroll
    for (int a = 0; a < 3; ++a) V.wl[(size_t)l * 3 + a] = w3[a];
    const double Cm[3][3] = {{Ci[0], Ci[1], Ci[2]}, {Ci[1], Ci[3], Ci[4]}, {Ci[2], Ci[4], Ci[5]}};
    for (int k = beg; k < end; ++k) {
        RowGeom q;
        row_geometry(V, V.states, V.X, k, q);
        double jl[6], jc[12];
        landmark_jacobian(q, jl);
        pose_jacobian(q, jl, jc);
        const double w = V.row_w[k];
        double* Ek = V.E + (size_t)k * 18;
        double* Yk = V.Y + (size_t)k * 18;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double e3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) e3[c] = w * (jc[a] * jl[c] + jc[6 + a] * jl[3 + c]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Ek[3 * a + c] = e3[c];
                Yk[3 * a + c] = e3[0] * Cm[0][c] + e3[1] * Cm[1][c] + e3[2] * Cm[2][c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ pose side
// 16 lanes per pose: lane t < 16 strides over the pose's rows, 21 + 6 + 6 partial sums, xor butterfly over the 16 lanes
__global__ __launch_bounds__(256) void k_pose_blocks(SchurView V) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    double acc[33];
#pragma unroll
    for (int q = 0; q < 33; ++q) acc[q] = 0.0;
    if (i < V.n) {
        const int beg = V.pose_ptr[i], end = V.pose_ptr[i + 1];
        for (int r = beg + l16; r < end; r += 16) {
            const int k = V.pose_rows[r];
            RowGeom q;
            row_geometry(V, V.states, V.X, k, q);
            double jl[6], jc[12];
            landmark_jacobian(q, jl);
            pose_jacobian(q, jl, jc);
            const double w = V.row_w[k];
            int e = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const double ja = w * jc[a], jb = w * jc[6 + a];
#pragma unroll
                for (int b = a; b < 6; ++b) { acc[e] = fma(ja, jc[b], fma(jb, jc[6 + b], acc[e])); ++e; }
                acc[21 + a] = fma(ja, q.ru, fma(jb, q.rv, acc[21 + a]));
            }
            // g_i -= Y_k w_l
            const double* Yk = V.Y + (size_t)k * 18;
            const double* wl = V.wl + (size_t)V.row_lm[k] * 3;
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[27 + a] = fma(Yk[3 * a], wl[0], fma(Yk[3 * a + 1], wl[1], fma(Yk[3 * a + 2], wl[2], acc[27 + a])));
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 33; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
    }
    if (i < V.n && l16 == 0) {
        double* Sd = V.S + (size_t)(6 * i) * V.Npad + 6 * i;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) Sd[(size_t)a * V.Npad + b] = acc[sym6(b, a)] + (a == b ? V.lamda : 0.0);
#pragma unroll
        for (int a = 0; a < 6; ++a) V.g[6 * i + a] = acc[21 + a] - acc[27 + a];
    }
}

// S_ij -= sum_{(k, k')} Y_k E_k'^T over the row pairs (k of pose i, k' of pose j) that share a landmark; one wave per block,
// lane t < 36 owns entry (t / 6, t % 6); the pair list is walked in a fixed order
__global__ __launch_bounds__(64) void k_pair_blocks(SchurView V) {
    const int bq = blockIdx.x;
    if (bq >= V.nblk) return;
    const int i = V.blk_i[bq], j = V.blk_j[bq];
    const int t = threadIdx.x;
    const int a = t / 6, b = t % 6;
    double s = 0.0;
    if (t < 36) {
        for (int p = V.blk_ptr[bq]; p < V.blk_ptr[bq + 1]; ++p) {
            const double* Yk = V.Y + (size_t)V.pair_k[p] * 18 + 3 * a;
            const double* Ek = V.E + (size_t)V.pair_k2[p] * 18 + 3 * b;
            s = fma(Yk[0], Ek[0], fma(Yk[1], Ek[1], fma(Yk[2], Ek[2], s)));
        }
        if (i != j || b <= a) V.S[(size_t)(6 * i + a) * V.Npad + 6 * j + b] -= s;
    }
}

// padding rows of the reduced system (N up to the next multiple of the tile): identity, so the factorisation runs on whole tiles
__global__ void k_pad_identity(SchurView V) {
    const int r = V.N + blockIdx.x * 64 + threadIdx.x;      // (grid: enough blocks of 64 for the padding rows)
    if (r < V.Npad) V.S[(size_t)r * V.Npad + r] = 1.0;
}

// ------------------------------------------------------------------------------------------------ dense Cholesky
// Diagonal tile kb: A_kk = L L^T in place (lower), inverse of L into invL[kb] (row major, lower).  ONE WAVE, the tile in
// registers: lane r holds row r (64 doubles); column j is scaled and the rank-1 update of the rows below runs with the
// column entries broadcast from their lanes (v_readlane) -- no LDS, no barrier (the LDS version spent its 130 us in 192
// workgroup barriers).  The inverse (needed so that the panel solve is a matrix product for the matrix cores) follows the
// same way: lane c owns column c of L^-1, the entries L[r][k] are broadcast.
__device__ __forceinline__ double bcast64(double v, int lane) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(64) void k_potrf64(SchurView V, int kb, int* info) {
    const int r = threadIdx.x;
    double* At = V.S + (size_t)(kb * kT) * V.Npad + kb * kT;
    double a[kT];
#pragma unroll
    for (int c = 0; c < kT; ++c) a[c] = At[(size_t)r * V.Npad + c];     // (the upper part is never used)
    bool bad = false;
#pragma unroll
    for (int j = 0; j < kT; ++j) {
        const double d = bcast64(a[j], j);
        if (!(d > 0.0)) bad = true;
        const double sd = sqrt(d > 0.0 ? d : 1.0), isd = 1.0 / sd;
        a[j] = r == j ? sd : a[j] * isd;            // column j: L[r][j] for r > j (rows above j hold garbage there, never read)
#pragma unroll
        for (int c = j + 1; c < kT; ++c) a[c] = fma(-a[j], bcast64(a[j], c), a[c]);     // A[r][c] -= L[r][j] L[c][j]; used for r >= c
    }
    if (bad && r == 0) *info = kb * kT + 1;
#pragma unroll
    for (int c = 0; c < kT; ++c)
        if (c <= r) At[(size_t)r * V.Npad + c] = a[c];
    // inverse: lane c owns column c of X = L^-1 (x[k] = X[k][c]); X[rr][c] = (delta - sum_{k < rr} L[rr][k] X[k][c]) / L[rr][rr]
    double x[kT];
#pragma unroll
    for (int rr = 0; rr < kT; ++rr) {
        double s = rr == r ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) s = fma(-bcast64(a[k], rr), x[k], s);
        x[rr] = s / bcast64(a[rr], rr);
    }
    double* out = V.invL + (size_t)kb * kT * kT;
#pragma unroll
    for (int rr = 0; rr < kT; ++rr) out[(size_t)rr * kT + r] = r <= rr ? x[rr] : 0.0;
}

// The same tile factorisation BLOCKED (round 4): four block columns of 16.  Only the 16 x 16 diagonal blocks are factorised
// and inverted element by element (one wave, row per lane, 16 steps each instead of 64); the panel below a diagonal block, the
// update of the tile's remaining blocks and the assembly of the full inverse (block lower triangular: X_jj = W_j,
// X_ij = -W_i sum_k L_ik X_kj) are 16 x 16 x 16 products on the matrix cores, the tile living in LDS.  256 threads; the
// chain of tile factorisations is the critical path of the whole Cholesky once the trailing update runs at K = 256 beside it.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// acc += A B^T (bt = true: b feeds B[lr][k]) or A B (bt = false: b feeds B[k][lr]); 16 x 16 blocks in LDS with leading dimension ld
__device__ __forceinline__ vf4 mm16(vf4 acc, const double* A, const double* B, int ld, bool bt, int lr, int lk) {
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        const int k = 4 * sidx + lk;
        const double a = A[lr * ld + k];
        const double b = bt ? B[lr * ld + k] : B[k * ld + lr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_potrf64b(SchurView V, int kb, int* info) {
    constexpr int LD = kT + 1;
    __shared__ double T[kT * LD];           // the tile, then its factor (lower)
    __shared__ double X[kT * LD];           // its inverse (lower)
    __shared__ double Y[4][16 * 17];        // per wave: an intermediate product
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, lr = lane & 15, lk = lane >> 4;
    double* At = V.S + (size_t)(kb * kT) * V.Npad + kb * kT;
    for (int e = t; e < kT * kT; e += 256) {
        const int r = e >> 6, c = e & 63;
        T[r * LD + c] = At[(size_t)r * V.Npad + c];
        X[r * LD + c] = 0.0;
    }
    __syncthreads();
    for (int jb = 0; jb < 4; ++jb) {
        const int o = 16 * jb;
        if (wv == 0) {      // diagonal block: factor and inverse, row per lane (lanes 16 .. 63 ride along on an identity)
            const int r = lane;
            double a[16], x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = r < 16 ? T[(o + r) * LD + o + c] : (c == (r & 15) ? 1.0 : 0.0);
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double d = bcast64(a[j], j);
                if (!(d > 0.0)) bad = true;
                const double sd = sqrt(d > 0.0 ? d : 1.0), isd = 1.0 / sd;
                a[j] = r == j ? sd : a[j] * isd;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) a[c] = fma(-a[j], bcast64(a[j], c), a[c]);
            }
            if (bad && r == 0) *info = kb * kT + o + 1;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                double sacc = rr == r ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < rr; ++k) sacc = fma(-bcast64(a[k], rr), x[k], sacc);
                x[rr] = sacc / bcast64(a[rr], rr);
            }
            if (r < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    T[(o + r) * LD + o + c] = c <= r ? a[c] : 0.0;
                    X[(o + c) * LD + o + r] = r <= c ? x[c] : 0.0;      // x[c] = X[c][r] of this block (lane r owns column r)
                }
            }
        }
        __syncthreads();
        // panel below: L_i,jb = A_i,jb W^T, one row tile per wave
        if (wv < 3 - jb) {
            const int i = jb + 1 + wv;
            double* Aij = T + (16 * i) * LD + o;
            vf4 acc = (vf4){0.0, 0.0, 0.0, 0.0};
            acc = mm16(acc, Aij, X + o * LD + o, LD, true, lr, lk);
            wave_lds_sync();        // (all operands of this wave are read before it overwrites its own tile)
#pragma unroll
            for (int q = 0; q < 4; ++q) Aij[(lk + 4 * q) * LD + lr] = acc[q];
        }
        __syncthreads();
        // the tile's remaining blocks: A_ik -= L_i,jb L_k,jb^T for jb < k <= i
        {
            int q = 0;
            for (int i = jb + 1; i < 4; ++i)
                for (int k = jb + 1; k <= i; ++k, ++q) {
                    if ((q & 3) != wv) continue;
                    vf4 acc = (vf4){0.0, 0.0, 0.0, 0.0};
                    acc = mm16(acc, T + (16 * i) * LD + o, T + (16 * k) * LD + o, LD, true, lr, lk);
                    double* C = T + (16 * i) * LD + 16 * k;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) C[(lk + 4 * qq) * LD + lr] -= acc[qq];
                }
        }
        __syncthreads();
    }
    // the inverse below its diagonal blocks, by distance from the diagonal: X_ij = -W_i (sum_{k = j}^{i-1} L_ik X_kj)
    for (int dist = 1; dist < 4; ++dist) {
        if (wv < 4 - dist) {
            const int j = wv, i = j + dist;
            vf4 acc = (vf4){0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; ++k) acc = mm16(acc, T + (16 * i) * LD + 16 * k, X + (16 * k) * LD + 16 * j, LD, false, lr, lk);
#pragma unroll
            for (int q = 0; q < 4; ++q) Y[wv][(lk + 4 * q) * 17 + lr] = acc[q];
            wave_lds_sync();
            vf4 acc2 = (vf4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int k = 4 * sidx + lk;
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(16 * i + lr) * LD + 16 * i + k], Y[wv][k * 17 + lr], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) X[(16 * i + lk + 4 * q) * LD + 16 * j + lr] = -acc2[q];
        }
        __syncthreads();
    }
    double* out = V.invL + (size_t)kb * kT * kT;
    for (int e = t; e < kT * kT; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (c <= r) At[(size_t)r * V.Npad + c] = T[r * LD + c];
        out[e] = c <= r ? X[r * LD + c] : 0.0;
    }
}

// One 64x64 tile of  out = alpha * (C + sign * A B^T)  on the matrix cores.  A, B: 64x64 row major with leading dimension
// lda / ldb.  256 threads = 4 waves, wave wv owns the 32x32 quadrant (wv >> 1, wv & 1) = 2x2 MFMA tiles of 16x16, K = 64
// in 16 steps of 4.  v_mfma_f64_16x16x4: lane l feeds A[l & 15][4 s + (l >> 4)] and B^T[4 s + (l >> 4)][l & 15] = B[l & 15][..]
// and owns C[(l >> 4) + 4 i][l & 15], i = 0..3.  Both operand tiles are staged in LDS first (which also makes the in-place
// panel form safe).
//   MODE 0 (panel):    S_{I,kb} <- S_{I,kb} invL_kb^T              grid = tiles I > kb
//   MODE 1 (trailing): S_{I,J}  -= S_{I,kb} S_{J,kb}^T             grid = tiles kb < J <= I
//   MODE 2 (inside a panel): as MODE 1 for the tile columns kb < J <= jhi only, grid = (rows I >= kb + 1, those columns)
template <int MODE>
__global__ __launch_bounds__(256) void k_gemm_abt(SchurView V, int kb, int jhi = 0) {
    __shared__ double As[kT][kT + 1];
    __shared__ double Bs[kT][kT + 1];
    const int t = threadIdx.x;
    int I, J;
    if (MODE == 0) {
        I = kb + 1 + blockIdx.x;
        J = kb;
    } else if (MODE == 2) {
        I = kb + 1 + blockIdx.x;
        J = kb + 1 + blockIdx.y;
        if (J > jhi || I < J) return;
    } else {        // blockIdx.x enumerates the lower triangle of the (nb - kb - 1)^2 trailing tiles
        const int q = blockIdx.x;
        int r = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        while (r * (r + 1) / 2 > q) --r;
        const int c = q - r * (r + 1) / 2;
        I = kb + 1 + r;
        J = kb + 1 + c;
    }
    const double* Ap = V.S + (size_t)(I * kT) * V.Npad + kb * kT;
    const double* Bp = MODE == 0 ? V.invL + (size_t)kb * kT * kT : V.S + (size_t)(J * kT) * V.Npad + kb * kT;
    const int ldb = MODE == 0 ? kT : V.Npad;
    for (int e = t; e < kT * kT; e += 256) {
        const int r = e / kT, c = e % kT;
        As[r][c] = Ap[(size_t)r * V.Npad + c];
        Bs[r][c] = Bp[(size_t)r * ldb + c];
    }
    __syncthreads();
    const int lane = t & 63, wv = t >> 6;
    const int r0 = (wv >> 1) * 32, c0 = (wv & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    vf4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (vf4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int s = 0; s < kT / 4; ++s) {
        const int k = 4 * s + lk;
        const double a0 = As[r0 + lr][k], a1 = As[r0 + 16 + lr][k];
        const double b0 = Bs[c0 + lr][k], b1 = Bs[c0 + 16 + lr][k];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    double* Cp = MODE == 0 ? V.S + (size_t)(I * kT) * V.Npad + kb * kT : V.S + (size_t)(I * kT) * V.Npad + J * kT;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + 16 * x + lk + 4 * i, c = c0 + 16 * y + lr;
                double* p = Cp + (size_t)r * V.Npad + c;
                if (MODE == 0) *p = acc[x][y][i];
                else if (I != J || c <= r) *p -= acc[x][y][i];
            }
}

// Trailing update of a whole PANEL of kPanel tile columns (round 4): S_{I,J} -= sum_{k in panel} S_{I,k} S_{J,k}^T with K = 256 per
// launch.  With K = 64 every 64 x 64 tile of the trailing matrix was read and written once per 64 columns -- 4 flop per byte,
// the update ran at the speed of memory (13.9 TFLOP/s = 3.5 TB/s); here a block owns 128 x 128 of C and walks K = 256 in
// chunks of 32 through LDS (16 x the arithmetic per byte of C, 2 x per byte of the operands); the next chunk's operands are
// requested from memory before the current chunk's matrix operations are issued.  Grid: (128-row blocks from the first trailing one, 128-column
// blocks [cj_lo, cj_hi]); blocks above the diagonal leave at once, diagonal blocks store their lower triangle.
#ifndef VBA_SYRK_CHUNK
#define VBA_SYRK_CHUNK 16
#endif
constexpr int kSyrkChunk = VBA_SYRK_CHUNK;
// (four waves per SIMD: two blocks per compute unit, one computes while the other stages, waits at a barrier or updates C)
__global__ __launch_bounds__(512, 4) void k_syrk_panel(SchurView V, int kb0, int b0, int cj_lo) {
    __shared__ double As[128][kSyrkChunk + 1];
    __shared__ double Bs[128][kSyrkChunk + 1];
    const int bi = b0 + (int)blockIdx.x, bj = cj_lo + (int)blockIdx.y;
    if (bi < bj) return;
    // 512 threads = 8 waves, wave wv owns 64 rows x 32 columns of the block (4 x 2 matrix-core tiles, 8 accumulators: two waves
    // per SIMD fit, one covers the other's LDS round trips and barriers -- with 64 x 64 per wave and one wave per SIMD the
    // matrix pipe was busy a quarter of the time)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int r0 = (wv >> 2) * 64, c0 = (wv & 3) * 32;
    const size_t ld = (size_t)V.Npad;
    const double* Ap = V.S + (size_t)(bi * 128) * ld + (size_t)kb0 * kT;
    const double* Bp = V.S + (size_t)(bj * 128) * ld + (size_t)kb0 * kT;
    constexpr int K = kPanel * kT, NK = K / kSyrkChunk;
    // staging: 128 rows x kSyrkChunk columns per operand and chunk = 64 kSyrkChunk double2 / 512 threads = kSyrkChunk / 8 each
    constexpr int NS = kSyrkChunk / 8, C2 = kSyrkChunk / 2;       // double2 per thread; double2 per row
    double2 pa[NS], pb[NS];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = t + 512 * i, row = e / C2, c2 = (e % C2) * 2;
            pa[i] = *reinterpret_cast<const double2*>(Ap + (size_t)row * ld + kc * kSyrkChunk + c2);
            pb[i] = *reinterpret_cast<const double2*>(Bp + (size_t)row * ld + kc * kSyrkChunk + c2);
        }
    };
    vf4 acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (vf4){0.0, 0.0, 0.0, 0.0};
    fetch(0);
    for (int kc = 0; kc < NK; ++kc) {
        __syncthreads();            // the matrix operations of the previous chunk have read their operands
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = t + 512 * i, row = e / C2, c2 = (e % C2) * 2;
            As[row][c2] = pa[i].x; As[row][c2 + 1] = pa[i].y;
            Bs[row][c2] = pb[i].x; Bs[row][c2 + 1] = pb[i].y;
        }
        __syncthreads();
        if (kc + 1 < NK) fetch(kc + 1);
#pragma unroll
        for (int sidx = 0; sidx < kSyrkChunk / 4; ++sidx) {
            const int k = 4 * sidx + lk;
            double a[4], b[2];
#pragma unroll
            for (int x = 0; x < 4; ++x) a[x] = As[r0 + 16 * x + lr][k];
#pragma unroll
            for (int y = 0; y < 2; ++y) b[y] = Bs[c0 + 16 * y + lr][k];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
        }
    }
    double* Cp = V.S + (size_t)(bi * 128) * ld + (size_t)bj * 128;
    const bool diag = bi == bj;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + 16 * x + lk + 4 * i, c = c0 + 16 * y + lr;
                if (!diag || c <= r) Cp[(size_t)r * ld + c] -= acc[x][y][i];
            }
}

// One tile step of the forward (L y = g) / backward (L^T x = y) substitution, a launch per step (as the factorisation):
// block 0 forms the step's 64 solution entries from the stored inverse diagonal factor, every other block forms them too
// (a 64x64 product, cheaper than a hand-off) and updates ONE tile of the right-hand side with them.  Tiles further than bw
// from the diagonal are zero (the factor of a banded matrix keeps its band) and get no block.
//   forward:  y_kb = invL_kb g_kb (-> ybuf),  g_I    -= L_{I,kb}   y_kb   for kb < I <= kb + bw
//   backward: x_kb = invL_kb^T ybuf_kb (-> g), ybuf_J -= L_{kb,J}^T x_kb   for kb - bw <= J < kb
__global__ __launch_bounds__(256) void k_trsv_step(SchurView V, int kb, int backward) {
    __shared__ double xb[kT];
    __shared__ double gb[kT];
    __shared__ double part[4][kT];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const double* Li = V.invL + (size_t)kb * kT * kT;
    const double* rhs = backward ? V.ybuf : V.g;
    if (t < kT) gb[t] = rhs[kb * kT + t];
    __syncthreads();
    for (int e = wv; e < kT; e += 4) {      // entry e: a 64-term dot product by one wave
        double sdot = backward ? Li[(size_t)lane * kT + e] * gb[lane] : Li[(size_t)e * kT + lane] * gb[lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
        if (lane == 0) xb[e] = sdot;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (t < kT) (backward ? V.g : V.ybuf)[kb * kT + t] = xb[t];
        return;
    }
    if (!backward) {
        const int I = kb + (int)blockIdx.x;
        // rows of the tile over the waves, lanes over its columns (64 consecutive doubles per load)
        for (int rr = wv; rr < kT; rr += 4) {
            const size_t row = (size_t)I * kT + rr;
            double sdot = V.S[row * V.Npad + kb * kT + lane] * xb[lane];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
            if (lane == 0) V.g[row] -= sdot;
        }
    } else {
        const int J = kb - (int)blockIdx.x;
        // lanes over the tile's columns (= entries of ybuf_J), the 64 rows split over the waves, partial sums through LDS
        double sdot = 0.0;
        for (int c = wv; c < kT; c += 4) sdot = fma(V.S[(size_t)(kb * kT + c) * V.Npad + J * kT + lane], xb[c], sdot);
        part[wv][lane] = sdot;
        __syncthreads();
        if (wv == 0) V.ybuf[J * kT + lane] -= ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    }
}

// ------------------------------------------------------------------------------------------------ update / cost
__global__ __launch_bounds__(256) void k_lm_update(SchurView V) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id < V.L) {
        const int l = id;
        double s[3] = {V.wl[3 * l], V.wl[3 * l + 1], V.wl[3 * l + 2]};
        for (int k = V.lm_ptr[l]; k < V.lm_ptr[l + 1]; ++k) {
            const double* Ek = V.E + (size_t)k * 18;
            const double* dc = V.g + 6 * V.row_pose[k];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int a = 0; a < 6; ++a) s[c] -= Ek[3 * a + c] * dc[a];
        }
        const double* Ci = V.Cinv + (size_t)l * 6;
        const double d0 = Ci[0] * s[0] + Ci[1] * s[1] + Ci[2] * s[2];
        const double d1 = Ci[1] * s[0] + Ci[3] * s[1] + Ci[4] * s[2];
        const double d2 = Ci[2] * s[0] + Ci[4] * s[1] + Ci[5] * s[2];
        V.dl[3 * l] = d0; V.dl[3 * l + 1] = d1; V.dl[3 * l + 2] = d2;
        V.X_new[3 * l] = V.X[3 * l] + d0;
        V.X_new[3 * l + 1] = V.X[3 * l + 1] + d1;
        V.X_new[3 * l + 2] = V.X[3 * l + 2] + d2;
    }
    if (id < V.n) {
        const double* dc = V.g + 6 * id;
        const double d9[9] = {dc[0], dc[1], dc[2], dc[3], dc[4], dc[5], 0.0, 0.0, 0.0};
        double o[10];
        retract(V.states + (size_t)id * 10, d9, o);
#pragma unroll
        for (int r = 0; r < 10; ++r) V.states_new[(size_t)id * 10 + r] = o[r];
    }
}

// cost at (states, X) given as arguments: sum w (ru^2 + rv^2) over rows + inv_sigma2 |X - X0|^2 over landmarks
__global__ __launch_bounds__(256) void k_cost(SchurView V, const double* states, const double* X) {
    __shared__ double red[4];
    const int id = blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (id < V.m) {
        RowGeom q;
        row_geometry(V, states, X, id, q);
        s = V.row_w[id] * (q.ru * q.ru + q.rv * q.rv);
    }
    if (id < V.L) {
        const double dx = X[3 * id] - V.X0[3 * id], dy = X[3 * id + 1] - V.X0[3 * id + 1], dz = X[3 * id + 2] - V.X0[3 * id + 2];
        s += V.inv_sigma2 * (dx * dx + dy * dy + dz * dz);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) V.part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

thread_local std::string g_serr;
int sfail(int code, const std::string& msg) { g_serr = msg; return code; }

#define SCHK(expr)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return sfail(VBA_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct vba_schur_context {
    int device = 0;
    SchurView V{};
    char* arena = nullptr;
    hipStream_t stream = nullptr, aux = nullptr;    // aux: the bulk of a panel's trailing update beside the next panel's factorisation
    hipEvent_t ev[5] = {};
    hipEvent_t ev_chain = nullptr, ev_bulk = nullptr;
    int classic = 0;                // VBA_SCHUR_CLASSIC=1: the round-2 tile-by-tile factorisation (comparison)
    int* d_info = nullptr;
    int last_info = 0;              // 0, or 1 + the row at which the last factorisation met a non-positive pivot
    double* S0 = nullptr;       // states buffers
    double* S1 = nullptr;
    double* X0buf = nullptr;
    double* X1buf = nullptr;
    std::vector<double> h_part;
    bool uploaded = false, have_state = false;
    float ms[4] = {0, 0, 0, 0};
    int64_t npairs = 0;
    int bw = 1 << 30;           // tile bandwidth of the reduced system (max |I - J| over its non-zero tiles), from the block list
};

extern "C" {

const char* vba_schur_last_error(void) { return g_serr.c_str(); }

int vba_schur_create(int device, int n, int64_t m, int L, int nblk, int64_t npairs, vba_schur_handle* out) {
    if (!out) return sfail(VBA_EINVAL, "null out");
    *out = nullptr;
    if (n < 1 || m < 1 || L < 1 || nblk < n || npairs < m) return sfail(VBA_EINVAL, "sizes out of range");
    if (m > INT32_MAX || npairs > INT32_MAX) return sfail(VBA_EINVAL, "m and npairs must fit 32-bit indices (CSR and pair lists are int32)");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) return sfail(VBA_ENODEV, "no HIP device visible");
    if (device < 0 || device >= cnt) return sfail(VBA_EINVAL, "device index out of range");
    SCHK(hipSetDevice(device));
    vba_schur_context* h = new (std::nothrow) vba_schur_context();
    if (!h) return sfail(VBA_ENOMEM, "host allocation failed");
    h->device = device;
    SchurView& V = h->V;
    V.n = n; V.m = m; V.L = L; V.nblk = nblk;
    V.N = 6 * n;
    V.nb = (V.N + kT * kPanel - 1) / (kT * kPanel) * kPanel;    // whole panels (the padding rows are an identity block)
    V.Npad = V.nb * kT;
    V.npart = (int)((std::max<int64_t>(m, L) + 255) / 256);
    h->npairs = npairs;
    size_t bytes = 0;
    auto need = [&](size_t b) { size_t o = bytes; bytes += (b + 255) & ~size_t(255); return o; };
    const size_t o_lmptr = need((size_t)(L + 1) * 4), o_rpose = need(m * 4), o_rlm = need(m * 4), o_u = need(m * 8), o_v = need(m * 8), o_w = need(m * 8);
    const size_t o_pptr = need((size_t)(n + 1) * 4), o_prows = need(m * 4);
    const size_t o_bi = need((size_t)nblk * 4), o_bj = need((size_t)nblk * 4), o_bptr = need((size_t)(nblk + 1) * 4), o_pk = need(npairs * 4), o_pk2 = need(npairs * 4);
    const size_t o_intr = need((size_t)n * 32), o_X0 = need((size_t)L * 24);
    const size_t o_s0 = need((size_t)n * 80), o_s1 = need((size_t)n * 80), o_x0 = need((size_t)L * 24), o_x1 = need((size_t)L * 24);
    const size_t o_ci = need((size_t)L * 48), o_wl = need((size_t)L * 24), o_E = need(m * 144), o_Y = need(m * 144);
    const size_t o_S = need((size_t)V.Npad * V.Npad * 8), o_g = need((size_t)V.Npad * 8), o_y = need((size_t)V.Npad * 8), o_iL = need((size_t)V.nb * kT * kT * 8);
    const size_t o_dl = need((size_t)L * 24), o_part = need((size_t)V.npart * 8), o_info = need(256);
    if (hipMalloc(&h->arena, bytes) != hipSuccess) { delete h; return sfail(VBA_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed"); }
    if (hipMemset(h->arena, 0, bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { hipFree(h->arena); delete h; return sfail(VBA_EHIP, "hipMemset failed"); }
    char* A = h->arena;
    V.lm_ptr = (int*)(A + o_lmptr); V.row_pose = (int*)(A + o_rpose); V.row_lm = (int*)(A + o_rlm);
    V.row_u = (double*)(A + o_u); V.row_v = (double*)(A + o_v); V.row_w = (double*)(A + o_w);
    V.pose_ptr = (int*)(A + o_pptr); V.pose_rows = (int*)(A + o_prows);
    V.blk_i = (int*)(A + o_bi); V.blk_j = (int*)(A + o_bj); V.blk_ptr = (int*)(A + o_bptr); V.pair_k = (int*)(A + o_pk); V.pair_k2 = (int*)(A + o_pk2);
    V.intr = (double*)(A + o_intr); V.X0 = (double*)(A + o_X0);
    h->S0 = (double*)(A + o_s0); h->S1 = (double*)(A + o_s1); h->X0buf = (double*)(A + o_x0); h->X1buf = (double*)(A + o_x1);
    V.Cinv = (double*)(A + o_ci); V.wl = (double*)(A + o_wl); V.E = (double*)(A + o_E); V.Y = (double*)(A + o_Y);
    V.S = (double*)(A + o_S); V.g = (double*)(A + o_g); V.ybuf = (double*)(A + o_y); V.invL = (double*)(A + o_iL);
    V.dl = (double*)(A + o_dl); V.part = (double*)(A + o_part); h->d_info = (int*)(A + o_info);
    h->h_part.resize(V.npart);
    bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_bulk, hipEventDisableTiming) == hipSuccess;
    if (const char* e = std::getenv("VBA_SCHUR_CLASSIC")) h->classic = std::atoi(e);
    for (int k = 0; k < 5 && ok; ++k) ok = hipEventCreate(&h->ev[k]) == hipSuccess;
    if (!ok) { vba_schur_destroy(h); return sfail(VBA_EHIP, "stream / event creation failed"); }
    *out = h;
    return VBA_OK;
}

int vba_schur_destroy(vba_schur_handle h) {
    if (!h) return VBA_OK;
    hipSetDevice(h->device);
    if (h->stream) { hipStreamSynchronize(h->stream); hipStreamDestroy(h->stream); }
    if (h->aux) { hipStreamSynchronize(h->aux); hipStreamDestroy(h->aux); }
    if (h->ev_chain) hipEventDestroy(h->ev_chain);
    if (h->ev_bulk) hipEventDestroy(h->ev_bulk);
    for (hipEvent_t e : h->ev) if (e) hipEventDestroy(e);
    if (h->arena) hipFree(h->arena);
    delete h;
    return VBA_OK;
}

int vba_schur_upload(vba_schur_handle h, const int* lm_ptr, const int* row_pose, const int* row_lm, const double* row_u,
                     const double* row_v, const double* row_w, const int* pose_ptr, const int* pose_rows, const int* blk_i,
                     const int* blk_j, const int* blk_ptr, const int* pair_k, const int* pair_k2, const double* intrinsics,
                     const double* X0, double sigma_prior) {
    if (!h) return sfail(VBA_EINVAL, "null handle");
    if (!lm_ptr || !row_pose || !row_lm || !row_u || !row_v || !row_w || !pose_ptr || !pose_rows || !blk_i || !blk_j || !blk_ptr ||
        !pair_k || !pair_k2 || !intrinsics || !X0)
        return sfail(VBA_EINVAL, "null array");
    if (!(sigma_prior > 0.0)) return sfail(VBA_EINVAL, "sigma_prior must be positive");
    const SchurView& V = h->V;
    // the structure arrays index each other: check them on the host before a kernel trusts them
    if (lm_ptr[0] != 0 || lm_ptr[V.L] != V.m || pose_ptr[0] != 0 || pose_ptr[V.n] != V.m || blk_ptr[0] != 0 || blk_ptr[V.nblk] != h->npairs)
        return sfail(VBA_EINVAL, "CSR pointers do not span their arrays");
    for (int l = 0; l < V.L; ++l) if (lm_ptr[l + 1] < lm_ptr[l]) return sfail(VBA_EINVAL, "lm_ptr not monotone");
    for (int i = 0; i < V.n; ++i) if (pose_ptr[i + 1] < pose_ptr[i]) return sfail(VBA_EINVAL, "pose_ptr not monotone");
    for (int b = 0; b < V.nblk; ++b) {
        if (blk_ptr[b + 1] < blk_ptr[b]) return sfail(VBA_EINVAL, "blk_ptr not monotone");
        if (blk_i[b] < 0 || blk_i[b] >= V.n || blk_j[b] < 0 || blk_j[b] > blk_i[b]) return sfail(VBA_EINVAL, "block index outside the lower triangle");
    }
    for (int64_t k = 0; k < V.m; ++k) {
        if (row_pose[k] < 0 || row_pose[k] >= V.n || row_lm[k] < 0 || row_lm[k] >= V.L || pose_rows[k] < 0 || pose_rows[k] >= V.m)
            return sfail(VBA_EINVAL, "row index out of range");
    }
    for (int64_t p = 0; p < h->npairs; ++p)
        if (pair_k[p] < 0 || pair_k[p] >= V.m || pair_k2[p] < 0 || pair_k2[p] >= V.m) return sfail(VBA_EINVAL, "pair index out of range");
    SCHK(hipSetDevice(h->device));
    auto up = [&](const void* dst, const void* src, size_t bytes) { return hipMemcpy(const_cast<void*>(dst), src, bytes, hipMemcpyHostToDevice); };
    SCHK(up(V.lm_ptr, lm_ptr, (size_t)(V.L + 1) * 4)); SCHK(up(V.row_pose, row_pose, V.m * 4)); SCHK(up(V.row_lm, row_lm, V.m * 4));
    SCHK(up(V.row_u, row_u, V.m * 8)); SCHK(up(V.row_v, row_v, V.m * 8)); SCHK(up(V.row_w, row_w, V.m * 8));
    SCHK(up(V.pose_ptr, pose_ptr, (size_t)(V.n + 1) * 4)); SCHK(up(V.pose_rows, pose_rows, V.m * 4));
    SCHK(up(V.blk_i, blk_i, (size_t)V.nblk * 4)); SCHK(up(V.blk_j, blk_j, (size_t)V.nblk * 4)); SCHK(up(V.blk_ptr, blk_ptr, (size_t)(V.nblk + 1) * 4));
    SCHK(up(V.pair_k, pair_k, h->npairs * 4)); SCHK(up(V.pair_k2, pair_k2, h->npairs * 4));
    SCHK(up(V.intr, intrinsics, (size_t)V.n * 32)); SCHK(up(V.X0, X0, (size_t)V.L * 24));
    h->V.inv_sigma2 = 1.0 / (sigma_prior * sigma_prior);
    h->bw = 0;
    for (int b = 0; b < V.nblk; ++b) h->bw = std::max(h->bw, (6 * blk_i[b] + 5) / kT - (6 * blk_j[b]) / kT);
    h->uploaded = true;
    return VBA_OK;
}

int vba_schur_set_state(vba_schur_handle h, const double* states, const double* landmarks) {
    if (!h || !states || !landmarks) return sfail(VBA_EINVAL, "null argument");
    SCHK(hipSetDevice(h->device));
    SCHK(hipStreamSynchronize(h->stream));
    SCHK(hipMemcpy(h->S0, states, (size_t)h->V.n * 80, hipMemcpyHostToDevice));
    SCHK(hipMemcpy(h->X0buf, landmarks, (size_t)h->V.L * 24, hipMemcpyHostToDevice));
    h->have_state = true;
    return VBA_OK;
}

int vba_schur_get_state(vba_schur_handle h, double* states, double* landmarks) {
    if (!h) return sfail(VBA_EINVAL, "null handle");
    if (!h->have_state) return sfail(VBA_ESTATE, "no state set");
    SCHK(hipSetDevice(h->device));
    SCHK(hipStreamSynchronize(h->stream));
    if (states) SCHK(hipMemcpy(states, h->S0, (size_t)h->V.n * 80, hipMemcpyDeviceToHost));
    if (landmarks) SCHK(hipMemcpy(landmarks, h->X0buf, (size_t)h->V.L * 24, hipMemcpyDeviceToHost));
    return VBA_OK;
}

static int schur_cost(vba_schur_handle h, const double* states, const double* X, double* cost) {
    SchurView V = h->V;
    hipLaunchKernelGGL(k_cost, dim3(V.npart), dim3(256), 0, h->stream, V, states, X);
    SCHK(hipGetLastError());
    SCHK(hipMemcpyAsync(h->h_part.data(), V.part, (size_t)V.npart * 8, hipMemcpyDeviceToHost, h->stream));
    SCHK(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (int b = 0; b < V.npart; ++b) s += h->h_part[b];      // fixed order
    *cost = s;
    return VBA_OK;
}

int vba_schur_iterate(vba_schur_handle h, double lamda, double* cost_before, double* cost_after, int* accepted) {
    if (!h || !cost_before || !cost_after || !accepted) return sfail(VBA_EINVAL, "null argument");
    if (!h->uploaded || !h->have_state) return sfail(VBA_ESTATE, "upload the problem and set the state first");
    if (!(lamda >= 0.0)) return sfail(VBA_EINVAL, "lamda must be >= 0");
    SCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    SchurView V = h->V;
    V.lamda = lamda;
    V.states = h->S0; V.X = h->X0buf; V.states_new = h->S1; V.X_new = h->X1buf;
    if (int rc = schur_cost(h, V.states, V.X, cost_before)) return rc;
    SCHK(hipEventRecord(h->ev[0], s));
    SCHK(hipMemsetAsync(V.S, 0, (size_t)V.Npad * V.Npad * 8, s));
    SCHK(hipMemsetAsync(V.g, 0, (size_t)V.Npad * 8, s));
    SCHK(hipMemsetAsync(h->d_info, 0, 4, s));
    hipLaunchKernelGGL(k_lm_blocks, dim3((V.L + 255) / 256), dim3(256), 0, s, V);
    hipLaunchKernelGGL(k_pose_blocks, dim3((V.n + 15) / 16), dim3(256), 0, s, V);
    hipLaunchKernelGGL(k_pair_blocks, dim3(V.nblk), dim3(64), 0, s, V);
    if (V.Npad > V.N) hipLaunchKernelGGL(k_pad_identity, dim3((V.Npad - V.N + 63) / 64), dim3(64), 0, s, V);
    SCHK(hipEventRecord(h->ev[1], s));
    if (h->classic == 1) {  // round 2: one tile column at a time, trailing update with K = 64 (VBA_SCHUR_CLASSIC=2: panels, round-2 tile kernel)
        for (int kb = 0; kb < V.nb; ++kb) {
            hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(64), 0, s, V, kb, h->d_info);
            const int rest = V.nb - kb - 1;
            if (rest > 0) {
                hipLaunchKernelGGL((k_gemm_abt<0>), dim3(rest), dim3(256), 0, s, V, kb, 0);
                hipLaunchKernelGGL((k_gemm_abt<1>), dim3(rest * (rest + 1) / 2), dim3(256), 0, s, V, kb, 0);
            }
        }
    } else {
        // Right-looking by PANELS of kPanel tile columns with look-ahead.  The panel is factorised tile column by tile column
        // (diagonal tile in one wave's registers, panel rows by its inverse, the panel's remaining columns updated with K = 64:
        // skinny, cheap); then the trailing matrix gets ONE update with K = 256.  That update is split: the 256 columns the NEXT
        // panel consists of first, on this stream -- the next panel's factorisation (a chain of small dependent launches) then
        // starts at once and runs beside the bulk of the update on the second stream.
        const int np = V.nb / kPanel, nB = V.nb / 2;            // panels; 128-row blocks
        bool bulk_pending = false;
        for (int p = 0; p < np; ++p) {
            const int kb0 = p * kPanel, klast = kb0 + kPanel - 1;
            for (int kb = kb0; kb <= klast; ++kb) {
                if (h->classic == 2) hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(64), 0, s, V, kb, h->d_info);
                else hipLaunchKernelGGL(k_potrf64b, dim3(1), dim3(256), 0, s, V, kb, h->d_info);
                const int rest = V.nb - kb - 1;
                if (rest > 0) hipLaunchKernelGGL((k_gemm_abt<0>), dim3(rest), dim3(256), 0, s, V, kb, 0);
                if (kb < klast) hipLaunchKernelGGL((k_gemm_abt<2>), dim3(rest, klast - kb), dim3(256), 0, s, V, kb, klast);
            }
            const int b0 = (kb0 + kPanel) / 2;                  // first 128-block of the trailing matrix
            if (b0 >= nB) break;
            SCHK(hipEventRecord(h->ev_chain, s));
            if (bulk_pending) SCHK(hipStreamWaitEvent(s, h->ev_bulk, 0));      // the previous bulk update wrote the columns of this one
            hipLaunchKernelGGL(k_syrk_panel, dim3(nB - b0, std::min(2, nB - b0)), dim3(512), 0, s, V, kb0, b0, b0);     // next panel's columns
            bulk_pending = false;
            if (nB - b0 > 2) {
                SCHK(hipStreamWaitEvent(h->aux, h->ev_chain, 0));
                hipLaunchKernelGGL(k_syrk_panel, dim3(nB - b0 - 2, nB - b0 - 2), dim3(512), 0, h->aux, V, kb0, b0 + 2, b0 + 2);
                SCHK(hipEventRecord(h->ev_bulk, h->aux));
                bulk_pending = true;
            }
        }
        if (bulk_pending) SCHK(hipStreamWaitEvent(s, h->ev_bulk, 0));
    }
    SCHK(hipEventRecord(h->ev[2], s));
    for (int kb = 0; kb < V.nb; ++kb)
        hipLaunchKernelGGL(k_trsv_step, dim3(1 + std::min(h->bw, V.nb - 1 - kb)), dim3(256), 0, s, V, kb, 0);
    for (int kb = V.nb - 1; kb >= 0; --kb)
        hipLaunchKernelGGL(k_trsv_step, dim3(1 + std::min(h->bw, kb)), dim3(256), 0, s, V, kb, 1);
    hipLaunchKernelGGL(k_lm_update, dim3((std::max(V.L, V.n) + 255) / 256), dim3(256), 0, s, V);
    SCHK(hipEventRecord(h->ev[3], s));
    SCHK(hipGetLastError());
    int info = 0;
    SCHK(hipMemcpyAsync(&info, h->d_info, 4, hipMemcpyDeviceToHost, s));
    SCHK(hipStreamSynchronize(s));
    for (int k = 0; k < 3; ++k) (void)hipEventElapsedTime(&h->ms[k], h->ev[k], h->ev[k + 1]);
    h->last_info = info;
    if (info != 0) {
        // Not positive definite at this damping: the ordinary LM answer to a failed factorisation is a rejected trial (the
        // caller raises lamda), not an error.  The state is untouched; vba_schur_last_info reports the failing row.
        *cost_after = *cost_before;
        *accepted = 0;
        return VBA_OK;
    }
    if (int rc = schur_cost(h, V.states_new, V.X_new, cost_after)) return rc;
    *accepted = (*cost_after < *cost_before) ? 1 : 0;
    if (*accepted) { std::swap(h->S0, h->S1); std::swap(h->X0buf, h->X1buf); }
    return VBA_OK;
}

int vba_schur_last_info(vba_schur_handle h, int* info) {
    if (!h || !info) return sfail(VBA_EINVAL, "null argument");
    *info = h->last_info;
    return VBA_OK;
}

int vba_schur_last_ms(vba_schur_handle h, float* build_ms, float* factor_ms, float* solve_ms) {
    if (!h) return sfail(VBA_EINVAL, "null handle");
    if (build_ms) *build_ms = h->ms[0];
    if (factor_ms) *factor_ms = h->ms[1];
    if (solve_ms) *solve_ms = h->ms[2];
    return VBA_OK;
}

// what: 0 = step of the last iteration [6 n + 3 L] (dc then dl), 1 = lower triangle of the Cholesky factor as a dense
// [6n][6n] row-major matrix (upper part zero)
int vba_schur_debug_fetch(vba_schur_handle h, int what, double* out, int64_t capacity) {
    if (!h || !out) return sfail(VBA_EINVAL, "null argument");
    SCHK(hipSetDevice(h->device));
    SCHK(hipStreamSynchronize(h->stream));
    const SchurView& V = h->V;
    if (what == 0) {
        if (capacity < (int64_t)V.N + 3 * (int64_t)V.L) return sfail(VBA_EINVAL, "buffer too small");
        SCHK(hipMemcpy(out, V.g, (size_t)V.N * 8, hipMemcpyDeviceToHost));
        SCHK(hipMemcpy(out + V.N, V.dl, (size_t)V.L * 24, hipMemcpyDeviceToHost));
        return VBA_OK;
    }
    if (what == 1) {
        if (capacity < (int64_t)V.N * V.N) return sfail(VBA_EINVAL, "buffer too small");
        std::vector<double> tmp((size_t)V.Npad * V.Npad);
        SCHK(hipMemcpy(tmp.data(), V.S, tmp.size() * 8, hipMemcpyDeviceToHost));
        for (int r = 0; r < V.N; ++r)
            for (int c = 0; c < V.N; ++c) out[(size_t)r * V.N + c] = c <= r ? tmp[(size_t)r * V.Npad + c] : 0.0;
        return VBA_OK;
    }
    return sfail(VBA_EINVAL, "unknown selector");
}

}  // extern "C"
