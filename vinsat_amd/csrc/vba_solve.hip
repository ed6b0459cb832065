// vba_solve.hip -- damped block-tridiagonal solve (A7), retraction (A8) and the LM accept test.
//
// The reference forms the (9n)^2 matrix densely and calls LU (BA_filtering.py:54-55).  The matrix is exactly
// block tridiagonal in 9x9 blocks and not symmetric, so the solve here is a block elimination along the
// pose chain with partial pivoting inside each 9x9 diagonal block:
//
//   forward :  D'_i = D_i + fp32(lamda) I - L_i X_{i-1},  y_i = g_i - L_i z_{i-1},
//              [X_i | z_i] = D'_i^{-1} [U_i | y_i]           (Gauss-Jordan, row pivoting)
//   backward:  x_{n-1} = z_{n-1},  x_i = z_i - X_i x_{i+1}
//
// One wavefront per window walks the chain.  Lane c of the wave owns COLUMN c of the 9 x 19 working matrix
// [D' | U | y] in 9 registers, so the pivot search is lane-local and a pivot step is 8 broadcasts (v_readlane)
// plus 8 FMAs per lane.  The lanes that end a step holding X_i are exactly the ones that need it as the
// D' columns of step i+1, so the two column groups swap roles every step and nothing is shuffled.
#include "vba_device.h"
#include "vba_launch.h"

namespace vba {

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long b = f64_bits(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane);
    return bits_f64(((unsigned long long)hi << 32) | lo);
}

// One forward step.  DB = first lane of the D' column group (0 or 9); the U group starts at 9 - DB.
template <int DB>
__device__ __forceinline__ void forward_step(const double* blk /*LDS: L,D,U (81 each), g(9)*/, double lam32, double (&a)[9],
                                             int lane, bool& zero_pivot) {
    constexpr int UB = 9 - DB;
    const bool isD = lane >= DB && lane < DB + 9;
    const bool isU = lane >= UB && lane < UB + 9;
    const bool isY = lane == 18;
    const int cc = isD ? lane - DB : (isU ? lane - UB : 0);
    double xp[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) xp[j] = (isD || isY) ? a[j] : 0.0;
    const double* base = isD ? blk + 81 + cc : (isU ? blk + 162 + cc : blk + 243);
    const int stride = isY ? 1 : 9;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        double v = (isD || isU || isY) ? base[r * stride] : 0.0;
        if (isD && r == cc) v += lam32;
#pragma unroll
        for (int j = 0; j < 9; ++j) v -= blk[r * 9 + j] * xp[j];   // L_i[r][j], broadcast read
        a[r] = v;
    }
    // Gauss-Jordan with partial pivoting on the D' columns
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        constexpr int dummy = 0;
        (void)dummy;
        const int pl = DB + k;
        double best = fabs(a[k]);
        int p = k;
#pragma unroll
        for (int r = k + 1; r < 9; ++r) {
            const double v = fabs(a[r]);
            if (v > best) { best = v; p = r; }
        }
        p = __builtin_amdgcn_readlane(p, pl);
#pragma unroll
        for (int r = k + 1; r < 9; ++r) {
            if (p == r) { const double t = a[k]; a[k] = a[r]; a[r] = t; }
        }
        const double piv = readlane_f64(a[k], pl);
        if (piv == 0.0) zero_pivot = true;
        a[k] = a[k] / piv;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (r != k) {
                const double f = readlane_f64(a[r], pl);
                a[r] -= f * a[k];
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_solve(DevView V) {
    __shared__ double blk[2][256];
    const int w = blockIdx.x;
    WinScalars& sc = V.sc[w];
    if (sc.done) return;
    const int n = V.n[w];
    const int lane = threadIdx.x;
    const StepParams prm = *V.prm;
    const size_t sb = (size_t)w * V.n_max;

    if (sc.n_trials == 0) {
        // init_residual = mean |[r_obs ; sqrt(Sigma) r_pred]| with UNweighted r_obs (BA_filtering.py:51)
        double so;
        if (V.m_total == 0) {
            double s = 0.0;
            const double* pi = V.part_init + (size_t)w * V.nblk_obs;
            for (int b = lane; b < V.nblk_obs; b += 64) s += pi[b];
            so = wave_sum(s);
        } else {
            so = sc.sum_abs_robs;
        }
        double sp = 0.0;
        if (!prm.initialize) {
            for (int i = lane; i < n - 1; i += 64) {
                const double* ro = V.rorb + (sb + i) * 6;
                sp += fabs(ro[0]) + fabs(ro[1]) + fabs(ro[2]) + fabs(ro[3]) + fabs(ro[4]) + fabs(ro[5]) + fabs(V.fatt[sb + i]);
            }
            sp = wave_sum(sp) * prm.sqrt_sigma;
        }
        const double M = V.m_total ? (double)V.m_total : (double)V.m[w];
        const double denom = 2.0 * M + (prm.initialize ? 6.0 : 7.0) * (double)(n - 1);
        if (lane == 0) {
            sc.sum_abs_robs = so;
            sc.sum_abs_rpred = sp;
            sc.init_residual = (so + sp) / denom;
        }
    }

    const double lam32 = (double)(float)sc.lamda;      // torch.eye() is float32 (BA_filtering.py:54)
    if (lane == 0) sc.lam32 = lam32;

    // ------------------------------------------------------------------ forward sweep
    double a[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = 0.0;
    bool zero_pivot = false;
    double pre[4];
    auto fetch = [&](int i) {
        const double* src = V.bands + (sb + i) * 243;
        const double* rsrc = V.rhs + (sb + i) * 9;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q;
            pre[q] = e < 243 ? src[e] : (e < 252 ? rsrc[e - 243] : 0.0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) blk[buf][lane + 64 * q] = pre[q];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const int buf = i & 1;
        if (i + 1 < n) fetch(i + 1);
        if (buf == 0) forward_step<0>(blk[0], lam32, a, lane, zero_pivot);
        else forward_step<9>(blk[1], lam32, a, lane, zero_pivot);
        // columns of X_i sit in the U group of this step, z_i in lane 18
        const int ub = buf == 0 ? 9 : 0;
        if (lane >= ub && lane < ub + 9) {
            double* X = V.Xs + (sb + i) * 81 + (lane - ub);
#pragma unroll
            for (int r = 0; r < 9; ++r) X[r * 9] = a[r];
        } else if (lane == 18) {
            double* z = V.zs + (sb + i) * 9;
#pragma unroll
            for (int r = 0; r < 9; ++r) z[r] = a[r];
        }
        if (i + 1 < n) stash(buf ^ 1);
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();

    // ------------------------------------------------------------------ backward sweep (lane r = row r)
    const int r = lane < 9 ? lane : 0;
    double x = V.zs[(sb + n - 1) * 9 + r];
    if (lane < 9) V.dpose[(sb + n - 1) * 9 + r] = x;
    double Xrow[9], zr;
    auto fetch_row = [&](int i) {
        const double* X = V.Xs + (sb + i) * 81 + r * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) Xrow[j] = X[j];
        zr = V.zs[(sb + i) * 9 + r];
    };
    if (n > 1) fetch_row(n - 2);
    for (int i = n - 2; i >= 0; --i) {
        double cur[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) cur[j] = Xrow[j];
        double v = zr;
        if (i > 0) fetch_row(i - 1);
#pragma unroll
        for (int j = 0; j < 9; ++j) v -= cur[j] * readlane_f64(x, j);
        x = v;
        if (lane < 9) V.dpose[(sb + i) * 9 + r] = x;
    }
    __threadfence_block();
    __syncthreads();

    // ------------------------------------------------------------------ retraction (BA_filtering.py:56-60)
    bool bad = false;
    for (int i = lane; i < n; i += 64) {
        const double* dp = V.dpose + (sb + i) * 9;
        double d9[9], o[10];
#pragma unroll
        for (int j = 0; j < 9; ++j) { d9[j] = dp[j]; bad |= !(fabs(d9[j]) <= 1.79e308); }
        retract(V.states + (sb + i) * 10, d9, o);
        double* sn = V.states_new + (sb + i) * 10;
#pragma unroll
        for (int j = 0; j < 10; ++j) sn[j] = o[j];
    }
    const unsigned long long anybad = __ballot(bad);
    if (lane == 0) {
        unsigned f = sc.flags;
        if (anybad) f |= 2u;
        if (zero_pivot) f |= 4u;
        sc.flags = f;
    }
}

// LM accept test (BA_filtering.py:66-79).  ranks == 0: sum this window's block partials; ranks > 0: the
// observation part is the rank-ordered sum of the gathered per-rank sums (sharded mode).
__global__ __launch_bounds__(64) void k_decide(DevView V, const double* trial_all, int ranks) {
    const int w = blockIdx.x;
    WinScalars& sc = V.sc[w];
    if (sc.done) return;
    const int n = V.n[w];
    const int lane = threadIdx.x;
    const StepParams prm = *V.prm;
    const size_t sb = (size_t)w * V.n_max;
    double S;
    if (ranks > 0) {
        S = trial_all[1];
        for (int q = 0; q < ranks; ++q) S += trial_all[2 * q];
    } else {
        const double* pt = V.part_trial + (size_t)w * (V.nblk_obs + V.nblk_dyn);
        double s = 0.0;
        for (int b = lane; b < V.nblk_obs + V.nblk_dyn; b += 64) s += pt[b];
        S = wave_sum(s);
    }
    const double M = V.m_total ? (double)V.m_total : (double)V.m[w];
    const double denom = 2.0 * M + (prm.initialize ? 6.0 : 7.0) * (double)(n - 1);
    const double residual = S / denom;
    const double lam = sc.lamda * 10.0;
    const bool accept = residual < sc.init_residual;
    const bool stop = accept || lam > 1e4;
    if (stop) {
        const double* s_new = V.states_new + sb * 10;
        double* s_cur = V.states + sb * 10;
        for (int k = lane; k < n * 10; k += 64) s_cur[k] = s_new[k];
        const double* D = V.bands + (sb + n - 1) * 243 + 81;
        for (int k = lane; k < 81; k += 64) sc.last_hessian[k] = D[k] + ((k / 9 == k % 9) ? sc.lam32 : 0.0);
    }
    if (lane == 0) {
        sc.trial_residual = residual;
        sc.n_trials += 1;
        if (stop) {
            sc.done = 1;
            if (!accept) sc.flags |= 1u;
            if (!(residual == residual)) sc.flags |= 2u;
            sc.lamda = fmax(fmin(1e-1, lam * 0.01), 1e-4);
        } else {
            sc.lamda = lam;
        }
    }
}

void launch_solve(const DevView& V, hipStream_t s) { hipLaunchKernelGGL(k_solve, dim3(V.W), dim3(64), 0, s, V); }

void launch_decide(const DevView& V, const double* trial_all, int ranks, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(V.W), dim3(64), 0, s, V, trial_all, ranks);
}

}  // namespace vba
