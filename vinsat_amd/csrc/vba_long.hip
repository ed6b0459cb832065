// vba_long.hip -- long gaps of the pose chain (A4 over hundreds of seconds), propagated PARALLEL IN TIME.
//
// The reference walks a gap of s seconds as s dependent RK4 steps of 1 s (propagate_orbit_dynamics, BA_utils.py:73-87) and
// differentiates that chain (predict, :457-509).  A window of a later pass of a sequence holds gaps of up to 1000 s
// (read_detections inserts a knot every 1000 s, od_pipe.py:213-221): on one lane that is a chain of ~10^5 dependent
// instructions per edge, per LM trial and once more (six times, with a tangent each) for the factor -- 1.28 ms per BA call
// at a 945 s gap in round 4, thirty times the 500-pose headline window.  Only the 6-vector state is a recurrence, and even
// that recurrence parallelises: the gap is cut into P ~ sqrt(s) chunks of L ~ sqrt(s) steps and the chunk-start states U_j are
// found by the parareal iteration (Lions, Maday, Turinici 2001) whose FINE propagator is the reference's own chain of 1 s
// steps and whose coarse propagator G is ONE RK4 step over the chunk:
//     U_0 = x0,  U_{j+1} <- F_j(U_j) + (G_j(U_j') - G_j(U_j))        (U' = the new iterate, swept left to right)
// The fixed point is U_{j+1} = F_j(U_j), the serial chain itself (after k iterations the first k chunks are the serial chain
// bit for bit).  The coarse step over ~27 s is wrong by ~1e-10 relative, so an iteration contracts the error by ~1e-8: the
// first one leaves ~1e-15 relative, and the fine pass of the second finds every chunk landing on the next chunk's start state
// to 2^-48 relative -- the chain's defect -- and stops there (one sweep and two fine passes up to ~1000 s, two or three
// sweeps at 3000 - 6000 s; prototype with convergence table: tools/parareal_prototype.py).  What comes out differs from the serial
// walk by rounding (<= 1e-15 relative measured, the size of the difference between this library's rsq-based acceleration and
// the reference's sqrt / division), is a function of the states and the step count only, and is the same in every kernel set
// -- both kernels below call the same long_states().
// Cost of a 935 s edge: 35 coarse steps + 2 * 27 fine steps + a 35-step matrix-vector recurrence instead of 935 fine steps.
//
// The transition matrix of a long edge is the ORDERED PRODUCT of the transition matrices of G <= 128 sub-chunks of ~10 steps (each
// from its start state on the converged chain, six tangent lanes per sub-chunk as in dynamics_block), multiplied pairwise in a
// fixed tree.  The start states are a by-product of the last fine pass -- and of the TRIAL kernel's: an accepted trial is evaluated
// at exactly the states the next call starts from, so its chain is carried (DevView::long_pool, the carried-keys argument applied
// to dynamics) and the next call's factor is the tangent pass and the product alone.
//
// Edges of at most kLongGap steps never come here: their arithmetic (and the bits of every window without a long gap) is
// what it was.  With the hop integrator (predict_gpu's <= 100 s steps) no edge is long.
#include "vba_device.h"
#include "vba_launch.h"

namespace vba {

namespace {

__device__ __forceinline__ double readlane_f64(double v, int lane /*wave-uniform*/) {
    const unsigned long long b = f64_bits(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return bits_f64(((unsigned long long)hi << 32) | lo);
}

// the coarse propagator: one RK4 step of length h (hh = h / 2, h6 = h / 6 precomputed: an IEEE division per step would sit on
// the critical path of the sweep)
__device__ __forceinline__ void rk4_coarse(const double* x, double h, double hh, double h6, double* o) {
    double k1[3], k2[3], k3[3], k4[3], p[3], v2[3], v3[3], v4[3];
    accel_jvp(x, nullptr, k1, nullptr, false);
#pragma unroll
    for (int i = 0; i < 3; ++i) { p[i] = x[i] + hh * x[3 + i]; v2[i] = x[3 + i] + hh * k1[i]; }
    accel_jvp(p, nullptr, k2, nullptr, false);
#pragma unroll
    for (int i = 0; i < 3; ++i) { p[i] = x[i] + hh * v2[i]; v3[i] = x[3 + i] + hh * k2[i]; }
    accel_jvp(p, nullptr, k3, nullptr, false);
#pragma unroll
    for (int i = 0; i < 3; ++i) { p[i] = x[i] + h * v3[i]; v4[i] = x[3 + i] + h * k3[i]; }
    accel_jvp(p, nullptr, k4, nullptr, false);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o[i] = x[i] + h6 * (x[3 + i] + 2 * v2[i] + 2 * v3[i] + v4[i]);
        o[3 + i] = x[3 + i] + h6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
}

// The chain is converged when every chunk's fine end state and the next chunk's start state agree to kLongTight, relative (the
// defects of up to 64 chunks add up along the gap: 2^-48 each keeps the end state within ~1e-13) -- or, a few units of rounding above
// that, when another sweep no longer reduces the defect (a chunk of ~50 fine steps carries ~1e-15 of rounding of its own).
constexpr double kLongTight = 0x1p-48, kLongLoose = 0x1p-45;

// One wavefront.  x0: the state at the start of the gap (the same in every lane), s: its steps.  Lane j < P ends up with the
// converged chunk-start state U_j in `U`; xh (every lane) = U_P, the state at the end of the gap.  Everything lives in registers:
// a lane keeps what concerns its chunk (its start state U, its end state N = the next chunk's start).
// The correction sweep is LINEARISED: with A_j the Jacobian of the coarse step at U_j (every lane forms its own, six tangents
// through one RK4 step) the parareal update G_j(U_j') - G_j(U_j) becomes A_j c_j and the sweep the affine recurrence
//     c_0 = 0,  c_{j+1} = (F_j(U_j) - U_{j+1}) + A_j c_j,   U_j' = U_j + c_j
// -- a matrix-vector product and six v_readlane per chunk instead of a nonlinear RK4 step (66 instead of 250 instructions on the
// serial path).  What the linearisation drops is second order in c (~1e-8 relative after the coarse chain: 1e-16), and the first
// k chunks are still the serial chain bit for bit after k sweeps (c_1 = F_0 - U_1 exactly, and so on).
// The iteration ends when the DEFECT of the chain is below tolerance: every chunk's fine propagation from its start state
// lands on the next chunk's start state, i.e. the U_j are the serial chain up to that tolerance -- checked right behind the
// fine pass, so a converged iterate costs no sweep of its own (one sweep, two fine passes for gaps up to ~1000 s).  Whatever the
// sweep does, that check is what the result answers to.
// subs: receives the states at the sub-chunk starts of the LAST fine pass -- subs[g * 6 .. ], g in chain order.
__device__ __forceinline__ void long_states(const double* x0, int s, const LongPlan pl, int lane, double* U /*[6]*/, double* xh /*[6]*/,
                                            double* subs) {
    const int P = pl.P;
    const int last_len = s - (P - 1) * pl.L;
    const int my_len = lane < P - 1 ? pl.L : (lane == P - 1 ? last_len : 0);
    const double hL = (double)pl.L, hhL = 0.5 * hL, h6L = hL / 6.0;
    const double hT = (double)last_len, hhT = 0.5 * hT, h6T = hT / 6.0;
    double F[6], N[6], u[6];
    double worst_prev = 1.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) { u[c] = x0[c]; U[c] = x0[c]; N[c] = x0[c]; }
    // the coarse chain (serial, the same in every lane)
    for (int j = 0; j < P; ++j) {
        const bool tail = j == P - 1;
        double g[6];
        rk4_coarse(u, tail ? hT : hL, tail ? hhT : hhL, tail ? h6T : h6L, g);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (lane == j) N[c] = g[c];
            if (lane == j + 1) U[c] = g[c];
            u[c] = g[c];
        }
    }
    for (int it = 0; it <= P; ++it) {
        // fine: every chunk from its current start state, in parallel
#pragma unroll
        for (int c = 0; c < 6; ++c) F[c] = U[c];
        {
            double* rec = it > 0 ? subs + (size_t)lane * pl.nsubL * 6 : nullptr;    // (the pass behind the coarse chain is never the last)
            int until = 0;
            for (int q = 0; q < my_len; ++q) {
                if (rec && q == until) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) rec[c] = F[c];
                    rec += 6;
                    until += pl.sub;
                }
                rk4_step<false>(F, nullptr, 1.0);
            }
        }
        if (it > 0) {       // (the coarse chain alone is never close enough)
            const double np = fmax(fmax(fabs(N[0]), fabs(N[1])), fabs(N[2])), nv = fmax(fmax(fabs(N[3]), fabs(N[4])), fabs(N[5]));
            const double dp = fmax(fmax(fabs(F[0] - N[0]), fabs(F[1] - N[1])), fabs(F[2] - N[2]));
            const double dv = fmax(fmax(fabs(F[3] - N[3]), fabs(F[4] - N[4])), fabs(F[5] - N[5]));
            const double rel = lane < P ? fmax(dp / np, dv / nv) : 0.0;
            if (__any(rel != rel)) break;               // non-finite states: nothing to converge to (the residual carries the NaN on)
            const double worst = wave_max(rel);
            if (worst <= kLongTight || (it >= 2 && worst <= kLongLoose && worst > 0.25 * worst_prev)) break;
            worst_prev = worst;
        }
        // the Jacobian of this lane's coarse step at its start state: A[r][k] = T[k][r] (two passes of three tangents: the registers)
        double T[6][6];
        {
            const double h = lane == P - 1 ? hT : hL;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                double xs[6], t3[3][6];
#pragma unroll
                for (int c = 0; c < 6; ++c) xs[c] = U[c];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int c = 0; c < 6; ++c) t3[k][c] = (c == 3 * half + k) ? 1.0 : 0.0;
                rk4_step_multi<3>(xs, t3, h);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int c = 0; c < 6; ++c) T[3 * half + k][c] = t3[k][c];
            }
        }
        double d[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) d[c] = F[c] - N[c];
        // sweep: c_{j+1} = d_j + A_j c_j; every lane evaluates it with its own A and d, lane j's value is the one that counts
        double cc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j = 0; j < P; ++j) {
            double cn[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double a = d[r];
#pragma unroll
                for (int k = 0; k < 6; ++k) a = fma(T[k][r], cc[k], a);
                cn[r] = readlane_f64(a, j);
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (lane == j) N[c] += cn[c];
                if (lane == j + 1) U[c] += cn[c];
                cc[c] = cn[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) xh[c] = readlane_f64(N[c], P - 1);
}

}  // namespace

// A slot of the pool, in states of six doubles: [0] x_hat, [1] the start state the chain belongs to, [2] spare, [3, 3 + G) the
// sub-chunk start states, then kLongSplit partial products of the transition matrix (36 doubles each).
constexpr int kLongHead = 3;
constexpr int kLongSplit = 8;           // workgroups that share the tangent pass of one edge (16 sub-chunks each)

// where the chain of long edge k of window w lives for the call of parity `par` (every long edge has a slot: an edge the window's
// pool has no room for is not marked long, vba_upload_window)
__device__ __forceinline__ double* long_slot(const DevView& V, int w, int par, int k) {
    const int off = V.long_off[(size_t)w * kLongCap + k];
    return V.long_pool + (((size_t)w * 2 + par) * V.long_pool_cap + off) * 6;
}

// The orbit residual of the long edges at the TRIAL states (BA_filtering.py:63-67): sqrt(Sigma) sum |[x_hat - p', 100 (v_hat - v')]|
// of edge long_idx[k] into slot nblk_obs + nblk_dyn + k of part_trial (k_trial's pose-chain lanes left that edge's orbit part out
// and kept its attitude part).  One wavefront per long edge; gated as k_trial is.  The chain it walked is left for the next call
// (slot of the parity that reads it): x_hat, then the sub-chunk start states.
__global__ __launch_bounds__(64) void k_long_trial(DevView V) {
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    if (V.sc[w].done) return;
    const int k = blockIdx.x, lane = threadIdx.x;
    double* slot = V.part_trial + (size_t)w * V.trial_stride + V.nblk_obs + V.nblk_dyn + k;
    if (k >= V.n_long[w]) {
        if (lane == 0) *slot = 0.0;
        return;
    }
    const int i = V.long_idx[(size_t)w * kLongCap + k];
    const size_t sb = (size_t)w * V.n_max;
    const double* st = V.states_new + (sb + i) * 10;
    const double* sn = st + 10;
    const double x0[6] = {st[0], st[1], st[2], st[7], st[8], st[9]};
    const int s = -V.steps[sb + i];
    double* carry = long_slot(V, w, V.par ^ 1, k);
    double U[6], x[6];
    long_states(x0, s, long_plan(s), lane, U, x, carry + kLongHead * 6);
    if (lane == 0) {
        // header: x_hat, then the start state this chain belongs to (what the reader checks)
#pragma unroll
        for (int c = 0; c < 6; ++c) { carry[c] = x[c]; carry[6 + c] = x0[c]; }
        const double r = fabs(x[0] - sn[0]) + fabs(x[1] - sn[1]) + fabs(x[2] - sn[2]) +
                         fabs((x[3] - sn[7]) * kVelCoeff) + fabs((x[4] - sn[8]) * kVelCoeff) + fabs((x[5] - sn[9]) * kVelCoeff);
        *slot = r * V.prm.sqrt_sigma;
    }
}

// The chain of the long edges at the INPUT states of a call, where the pool does not hold it already.  The slot of the call's
// parity was written by the previous call's trial kernel; IF that chain started from the very bits of this call's state (the header
// says which state it belongs to; a chain is a function of that state and the step count alone, and an upload of the window clears
// the pool) there is nothing to do -- otherwise (fresh states from the host, a landmark-only call in front) one wavefront finds it.
__global__ __launch_bounds__(64) void k_long_chain(DevView V) {
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= V.n_long[w]) return;
    const int i = V.long_idx[(size_t)w * kLongCap + k];
    const size_t pb = (size_t)w * V.n_max + i;
    const double* st = V.states + pb * 10;
    const double x0[6] = {st[0], st[1], st[2], st[7], st[8], st[9]};
    double* chain = long_slot(V, w, V.par, k);
    bool same = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) same = same && f64_bits(chain[6 + c]) == f64_bits(x0[c]);
    if (same) return;       // (every lane takes the same decision from the same six words)
    const int s = -V.steps[pb];
    double U[6], x[6];
    long_states(x0, s, long_plan(s), lane, U, x, chain + kLongHead * 6);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) { chain[c] = x[c]; chain[6 + c] = x0[c]; }
    }
}

// The transition matrices of the sub-chunks of the long edges and their ordered product, first part.  kLongSplit workgroups of 128
// per long edge (on as many compute units: sixteen waves of fp64 arithmetic on ONE compute unit took 38 us for what is 6 us of
// instructions per wave), behind k_long_chain: a workgroup takes 16 consecutive sub-chunks -- 8 lanes each, 6 tangents as in
// dynamics_block, ~10 steps from the sub-chunk's start state in the pool -- and multiplies their matrices in order, pairwise, in
// place: its partial product M_{g0 + 15} ... M_{g0} goes behind the chain in the edge's slot.
__global__ __launch_bounds__(128) void k_long_tangent(DevView V) {
    __shared__ double M[16][36];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int k = blockIdx.x / kLongSplit, part = blockIdx.x % kLongSplit, tid = threadIdx.x;
    if (k >= V.n_long[w]) return;
    const int i = V.long_idx[(size_t)w * kLongCap + k];
    const size_t pb = (size_t)w * V.n_max + i;
    const int s = -V.steps[pb];
    const LongPlan pl = long_plan(s);
    const int g0 = part * 16;
    if (g0 >= pl.G) return;
    const int cnt = pl.G - g0 < 16 ? pl.G - g0 : 16;
    double* chain = long_slot(V, w, V.par, k);
    {
        const int g = g0 + (tid >> 3), c = tid & 7;
        if (g < pl.G && c < 6) {
            const int j = g / pl.nsubL < pl.P - 1 ? g / pl.nsubL : pl.P - 1;      // the chunk of this sub-chunk (the tail chunk may hold fewer)
            const int q = g - j * pl.nsubL;
            const int clen = j < pl.P - 1 ? pl.L : s - (pl.P - 1) * pl.L;
            const int len = clen - q * pl.sub < pl.sub ? clen - q * pl.sub : pl.sub;
            const double* sg = chain + (size_t)(kLongHead + g) * 6;
            double x[6], t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 6; ++r) x[r] = sg[r];
            t[c] = 1.0;
            for (int e = 0; e < len; ++e) rk4_step<true>(x, t, 1.0);
#pragma unroll
            for (int r = 0; r < 6; ++r) M[tid >> 3][6 * r + c] = t[r];
        }
    }
    __syncthreads();
    // ordered product, in place: at stride d the matrix at 2 a d takes M[2 a d + d] * M[2 a d] (a matrix without a partner stays)
    for (int d = 1; d < cnt; d <<= 1) {
        const int pairs = (cnt + 2 * d - 1) / (2 * d);
        double v[3];
        int n = 0;
        for (int e = tid; e < pairs * 36; e += 128, ++n) {
            const int a = e / 36, rc = e % 36, r = rc / 6, c = rc % 6;
            const int lo = 2 * a * d, hi = lo + d;
            if (hi < cnt) {
                double acc = M[hi][6 * r] * M[lo][c];
#pragma unroll
                for (int q = 1; q < 6; ++q) acc = fma(M[hi][6 * r + q], M[lo][6 * q + c], acc);
                v[n] = acc;
            } else {
                v[n] = M[lo][rc];
            }
        }
        __syncthreads();
        n = 0;
        for (int e = tid; e < pairs * 36; e += 128, ++n) M[2 * (e / 36) * d][e % 36] = v[n];
        __syncthreads();
    }
    if (tid < 36) chain[(size_t)(kLongHead + pl.G) * 6 + (size_t)part * 36 + tid] = M[0][tid];
}

// ... second part: the partial products of an edge multiplied in order -- Phi = M_{G-1} ... M_1 M_0 --, and with x_hat from the
// chain's header the prediction, the residual r_orbit of pose long_idx[k] and sum |r_orbit| into slot nblk_pred + k of part_pred
// (parity of the call).  One wavefront per long edge.
__global__ __launch_bounds__(64) void k_long_finish(DevView V) {
    __shared__ double M[kLongSplit][36];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int k = blockIdx.x, tid = threadIdx.x;
    double* slot = V.part_pred + ((size_t)w * 2 + V.par) * V.pred_stride + V.nblk_pred + k;
    if (k >= V.n_long[w]) {
        if (tid == 0) *slot = 0.0;
        return;
    }
    const int i = V.long_idx[(size_t)w * kLongCap + k];
    const size_t pb = (size_t)w * V.n_max + i;
    const double* st = V.states + pb * 10;
    const int s = -V.steps[pb];
    const LongPlan pl = long_plan(s);
    const double* chain = long_slot(V, w, V.par, k);
    const int cnt = (pl.G + 15) / 16;
    for (int e = tid; e < cnt * 36; e += 64) (&M[0][0])[e] = chain[(size_t)(kLongHead + pl.G) * 6 + e];
    if (tid == 0) {
        const double* x = chain;
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
        *slot = fabs(ro[0]) + fabs(ro[1]) + fabs(ro[2]) + fabs(ro[3]) + fabs(ro[4]) + fabs(ro[5]);
    }
    __syncthreads();
    for (int d = 1; d < cnt; d <<= 1) {
        const int pairs = (cnt + 2 * d - 1) / (2 * d);
        double v[3];
        int n = 0;
        for (int e = tid; e < pairs * 36; e += 64, ++n) {
            const int a = e / 36, rc = e % 36, r = rc / 6, c = rc % 6;
            const int lo = 2 * a * d, hi = lo + d;
            if (hi < cnt) {
                double acc = M[hi][6 * r] * M[lo][c];
#pragma unroll
                for (int q = 1; q < 6; ++q) acc = fma(M[hi][6 * r + q], M[lo][6 * q + c], acc);
                v[n] = acc;
            } else {
                v[n] = M[lo][rc];
            }
        }
        __syncthreads();
        n = 0;
        for (int e = tid; e < pairs * 36; e += 64, ++n) M[2 * (e / 36) * d][e % 36] = v[n];
        __syncthreads();
    }
    if (tid < 36) V.Phi[pb * 36 + tid] = M[0][tid];
}

void launch_long_factor(const DevView& V, hipStream_t s) {
    if (V.nblk_long <= 0 || V.hop) return;
    hipLaunchKernelGGL(k_long_chain, dim3(V.nblk_long, V.W), dim3(64), 0, s, V);
    hipLaunchKernelGGL(k_long_tangent, dim3(V.nblk_long * kLongSplit, V.W), dim3(128), 0, s, V);
    hipLaunchKernelGGL(k_long_finish, dim3(V.nblk_long, V.W), dim3(64), 0, s, V);
}

void launch_long_trial(const DevView& V, hipStream_t s) {
    if (V.nblk_long <= 0 || V.hop || V.prm.initialize) return;
    hipLaunchKernelGGL(k_long_trial, dim3(V.nblk_long, V.W), dim3(64), 0, s, V);
}

}  // namespace vba
