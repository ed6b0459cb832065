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
// A wavefront walks a chain.  Lane c owns COLUMN c of the working matrix [D' | U | right-hand sides] in 9
// registers, so the pivot search is lane-local and a pivot step is 8 broadcasts (v_readlane) plus 8 FMAs
// per lane; extra right-hand-side columns ride along in otherwise idle lanes.  The lanes that end a step
// holding X_i are the ones that need it as the D' columns of step i+1, so the two column groups swap roles
// every step and nothing is shuffled.
//
// Two drivers share that step:
//   * k_solve          one wave per window walks all n blocks (work-optimal; used when many windows are
//                      batched, the windows supply the parallelism);
//   * k_solve_chunks / k_solve_reduced / k_solve_recover
//                      the chain is cut into P chunks separated by single "separator" blocks.  Every chunk is
//                      eliminated by its own wave with 19 right-hand sides (g and the couplings to its two
//                      separators), a reduced block-tridiagonal system over the P-1 separators is solved by one
//                      wave, and the interiors are recovered in parallel: ~ n/P + P sequential block steps
//                      instead of n.  Default of the latency mode: k_solve_chunks_ts (two waves per chunk, meeting in the
//                      middle), the reduced system by block cyclic reduction -- k_cr_level01 (first two levels, one
//                      workgroup per four separators) and k_solve_reduced_cr (the rest in one workgroup) -- and the
//                      recovery inside the trial kernel (vba_step.h).
#include <atomic>

#include "vba_asm.h"
#include "vba_asm_fast.h"
#include "vba_decide.h"
#include "vba_device.h"
#include "vba_launch.h"
#include "vba_step.h"

namespace vba {

typedef double vf4 __attribute__((ext_vector_type(4)));     // accumulator of v_mfma_f64_16x16x4

// Diagnostic builds (-DVBA_RESIDENT_STAMPS; tools/tail_stamps.py): 100 MHz wall-clock stamps of one thread along the
// single-window solve kernels, fetched with vba_debug_fetch(h, 0, 101, ...).
#ifdef VBA_RESIDENT_STAMPS
__device__ unsigned long long g_kstamps[128];
#define VBA_KSTAMP(on, slot) do { if (on) g_kstamps[slot] = wall_clock64(); } while (0)
void fetch_kstamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kstamps), sizeof(g_kstamps)); }
#else
#define VBA_KSTAMP(on, slot) do {} while (0)
#endif

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long b = f64_bits(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane);
    return bits_f64(((unsigned long long)hi << 32) | lo);
}

// Lane roles of a forward step.  DB = first lane of the D' group (0 or 9), the U group starts at 9 - DB,
// right-hand-side columns sit in lanes [18, 18 + NRHS).
template <int DB, int NRHS>
struct Roles {
    static constexpr int UB = 9 - DB;
    __device__ static bool isD(int lane) { return lane >= DB && lane < DB + 9; }
    __device__ static bool isU(int lane) { return lane >= UB && lane < UB + 9; }
    __device__ static bool isR(int lane) { return lane >= 18 && lane < 18 + NRHS; }
};

// 1/x to ~1 ulp from v_rcp_f64's 24-bit seed (the full IEEE division sequence is 3x longer and sits on the
// critical path of every pivot)
__device__ __forceinline__ double fast_rcp(double x) {
    // 1/x = r / (1 - e) with e = 1 - x r ~ 4e-8: r (1 + e + e^2) is exact to e^3, three dependent operations after the
    // seed instead of the four of two Newton steps
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    const double t = fma(e, e, e);
    r = fma(r, t, r);
    return r;
}

// a[] enters holding this lane's column of [X_{i-1} | z_{i-1}] (or zeros), base[] this lane's column of
// [D_i + lam I | U_i | rhs_i]; on exit a[] holds the column of [I | X_i | z_i].  Lmat (LDS, row major 9x9) is
// L_i, or null for the first block of a chain.
//
// Per pivot the dependent chain is: lane-local tree search for the largest |entry| of the pivot column ->
// reciprocal of that entry (computed by every lane on its own candidate, only the pivot lane's is used) ->
// two broadcasts (row index, reciprocal) -> scale -> rank-1 update.  The row swap and the broadcasts of the
// eight multipliers run beside the reciprocal.
// PIVOT = false is the fast path: the damped normal equations are symmetric positive definite up to a ~1e-6
// relative non-symmetric term, for which elimination without row exchanges is as stable as Cholesky; every
// pivot is checked against the diagonal entry it started from and a failed check (`bad`) makes the host repeat
// the solve with PIVOT = true (row pivoting inside the 9x9 block).
// SPARSE_L: L_i of the assembled system has the pattern [pp 0 pv; 0 rr 0; vp 0 vv] over (position, rotation,
// velocity) (the orbit factor does not touch the rotation slots and the attitude term touches nothing else),
// so 45 instead of 81 multiply-adds; not valid for the reduced system.
// GROUPED: the wave holds several independent chains side by side (19 lanes each, `lane` is the lane inside the
// group, `gbase` the group's first lane); broadcasts then come from the group's own pivot lane through the LDS
// crossbar (ds_bpermute) instead of v_readlane, and the pivot row index is a per-lane value.
template <bool GROUPED>
__device__ __forceinline__ double bcast_f64(double v, int src) {
    if (GROUPED) return __shfl(v, src, kWave);
    return readlane_f64(v, src);
}
template <bool GROUPED>
__device__ __forceinline__ int bcast_i32(int v, int src) {
    if (GROUPED) return __shfl(v, src, kWave);
    return __builtin_amdgcn_readlane(v, src);
}

template <int DB, int NRHS, bool PIVOT, bool SPARSE_L, bool GROUPED = false>
__device__ __forceinline__ void forward_step(const double* Lmat, const double (&base)[9], double (&a)[9], int lane,
                                             bool& bad, int gbase = 0) {
    using R = Roles<DB, NRHS>;
    const bool carry = R::isD(lane) || R::isR(lane);
    double xp[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) xp[j] = carry ? a[j] : 0.0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        double v = base[r];
        if (Lmat) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bool rot_r = (r >= 3 && r < 6), rot_j = (j >= 3 && j < 6);
                if (!SPARSE_L || rot_r == rot_j) v = fma(-Lmat[r * 9 + j], xp[j], v);   // broadcast LDS read
            }
        }
        a[r] = v;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int pl = DB + k;
        double inv;
        if (PIVOT) {
            // lane-local tree search for the largest |entry| of the pivot column, reciprocal of that entry
            // computed by every lane on its own candidate, then two broadcasts (row index, reciprocal)
            const int cnt = 9 - k;
            double cv[9], cs[9];
            int ci[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                if (r < cnt) { cs[r] = a[k + r]; cv[r] = fabs(cs[r]); ci[r] = k + r; }
            }
#pragma unroll
            for (int step = 1; step < 9; step *= 2) {
#pragma unroll
                for (int r = 0; r < 9; r += 2 * step) {
                    if (r + step < cnt) {
                        const bool take = cv[r + step] > cv[r];       // strict: the lowest row wins a tie
                        cv[r] = take ? cv[r + step] : cv[r];
                        cs[r] = take ? cs[r + step] : cs[r];
                        ci[r] = take ? ci[r + step] : ci[r];
                    }
                }
            }
            const double inv_l = fast_rcp(cs[0]);
            const int p = bcast_i32<GROUPED>(ci[0], gbase + pl);
            inv = bcast_f64<GROUPED>(inv_l, gbase + pl);
            if (!(fabs(inv) <= 1.79e308)) bad = true;
            const double ak = a[k];
            double nk = ak;
#pragma unroll
            for (int r = k + 1; r < 9; ++r) {     // row swap k <-> p (p is wave uniform), branch free
                const bool sel = (p == r);
                const double ar = a[r];
                nk = sel ? ar : nk;
                a[r] = sel ? ak : ar;
            }
            a[k] = nk;
        } else {
            // pivot on the diagonal; it must stay a healthy fraction of the diagonal entry it started from
            if (lane == pl && !(a[k] > 1e-10 * base[k])) bad = true;
            inv = bcast_f64<GROUPED>(fast_rcp(a[k]), gbase + pl);
        }
        double f[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) f[r] = (r != k) ? bcast_f64<GROUPED>(a[r], gbase + pl) : 0.0;
        a[k] = a[k] * inv;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (r != k) a[r] = fma(-f[r], a[k], a[r]);
        }
    }
}

// The unpivoted elimination with row broadcasts (DPP) instead of v_readlane: every ROW of 16 lanes holds the nine columns of
// D in its lanes 0..8 (the same values in all rows: each row pivots its own copy) and seven of the nineteen columns of
// [L | U | g] in lanes 9..15 (rows 0..2; row 3 idles).  A pivot is then reciprocal -> one v_mov_b64_dpp -> scale -> eight
// v_fmac_f64_dpp (dpp_rank1_9) per lane instead of twenty v_readlane through scalar registers + nine multiply-adds -- the
// same operations per entry in the same order as forward_step<0, 10, false, false>, so the same bits.
template <int K = 0>
__device__ __forceinline__ void cr_pivots_dpp(const double (&base)[9], double (&a)[9], int c, bool& bad) {
    if constexpr (K < 9) {
        bad = bad | ((c == K) & !(a[K] > 1e-10 * base[K]));
        const double inv = bcast_row16<K>(fast_rcp(a[K]));
        a[K] = a[K] * inv;
        dpp_rank1_9<K>(a);
        cr_pivots_dpp<K + 1>(base, a, c, bad);
    }
}

// A failed pivot check: with row pivoting it is a numerically singular block (flag 4, result kept as in the
// reference); without it the host is asked to repeat this solve with pivoting (internal flag 8) and the window
// stays on the pivoted kernels for the rest of the call (internal flag 16).  The choice is per window, so what one
// window of a batch needs never changes the arithmetic of another.
// V.pivot: 0 = only the unpivoted kernels are launched, 1 = only the pivoted ones and they take every window,
// 2 = both are launched and each takes the windows whose sticky bit matches.
template <bool PIVOT>
__device__ __forceinline__ bool solver_mine(const DevView& V, const WinScalars& sc) {
    if (V.pivot == 1) return PIVOT;
    return ((sc.fl[V.par] & 16u) != 0) == PIVOT;
}

template <bool PIVOT>
__device__ __forceinline__ void report_pivot(bool bad, WinScalars& sc, int lane, int par) {
    const unsigned long long any = __ballot(bad);
    if (lane == 0 && any) atomicOr(&sc.fl[par], PIVOT ? 4u : (8u | 16u));
}

// ================================================================================================== sequential
// Walks blocks [0, n) of a block-tridiagonal system stored as bands[n][3][81], rhs[n][9]; damping lam32 is
// added to the diagonal.  Writes the solution to x[n][9].  Xs/zs: scratch [n][81], [n][9] in global memory.
// block sources: entry e of block i, e in [0,243) = sub|diag|super row major, [243,252) = right-hand side
struct BandSource {
    const double* bands;
    const double* rhs;
    __device__ double operator()(int i, int e) const { return e < 243 ? bands[(size_t)i * 243 + e] : rhs[(size_t)i * 9 + (e - 243)]; }
};

// A source may stage its own inputs beside the walk (RawSource): prefetch(i) starts the loads of what block i needs,
// commit(i) puts them where operator() finds them; the barrier of the walk's step orders the two.  No-ops otherwise.
template <class S> __device__ __forceinline__ auto src_prefetch(const S& s, int i, int) -> decltype(s.prefetch(i), void()) { s.prefetch(i); }
template <class S> __device__ __forceinline__ void src_prefetch(const S&, int, long) {}
template <class S> __device__ __forceinline__ auto src_commit(const S& s, int i, int) -> decltype(s.commit(i), void()) { s.commit(i); }
template <class S> __device__ __forceinline__ void src_commit(const S&, int, long) {}
// a source that COMPUTES its entries takes them behind the elimination step (nothing of it is live across the step)
template <class S> constexpr auto src_late(int) -> decltype(S::kLateFetch) { return S::kLateFetch; }
template <class S> constexpr bool src_late(long) { return false; }

#ifdef VBA_STAMPS
#define VBA_STAMP(k) do { if (stamps && lane == 0) stamps[k] = clock64(); } while (0)
#else
#define VBA_STAMP(k) do { } while (0)
#endif

template <bool PIVOT, bool SPARSE_L, class Src>
__device__ __forceinline__ void chain_solve(const Src& src, int n, double lam32, double* Xs, double* zs, double* x_out,
                                            double (*blk)[256], int lane, bool& zero_pivot, long long* stamps = nullptr) {
    VBA_STAMP(0);
    double a[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = 0.0;
    double pre[4];
    auto fetch = [&](int i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q;
            pre[q] = e < 252 ? src(i, e) : 0.0;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) blk[buf][lane + 64 * q] = pre[q];
    };
    auto load_base = [&](const double* b, int db, double (&base)[9]) {
        const int ub = 9 - db;
        const bool isD = lane >= db && lane < db + 9, isU = lane >= ub && lane < ub + 9, isY = lane == 18;
        const int cc = isD ? lane - db : (isU ? lane - ub : 0);
        const double* p = isD ? b + 81 + cc : (isU ? b + 162 + cc : b + 243);
        const int stride = isY ? 1 : 9;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = (isD || isU || isY) ? p[r * stride] : 0.0;
            if (isD && r == cc) v += lam32;
            base[r] = v;
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const int buf = i & 1;
        constexpr bool late = src_late<Src>(0);
        src_prefetch(src, i + 2, 0);
        if (!late && i + 1 < n) fetch(i + 1);
        double base[9];
        if (buf == 0) {
            load_base(blk[0], 0, base);
            forward_step<0, 1, PIVOT, SPARSE_L>(i > 0 ? blk[0] : nullptr, base, a, lane, zero_pivot);
        } else {
            load_base(blk[1], 9, base);
            forward_step<9, 1, PIVOT, SPARSE_L>(blk[1], base, a, lane, zero_pivot);
        }
        const int ub = buf == 0 ? 9 : 0;    // X_i sits in the U group of this step, z_i in lane 18
        if (lane >= ub && lane < ub + 9) {
            double* X = Xs + (size_t)i * 81 + (lane - ub);
#pragma unroll
            for (int r = 0; r < 9; ++r) X[r * 9] = a[r];
        } else if (lane == 18) {
            double* z = zs + (size_t)i * 9;
#pragma unroll
            for (int r = 0; r < 9; ++r) z[r] = a[r];
        }
        if constexpr (late) {
            if (i + 1 < n) src.form(i + 1, blk[buf ^ 1]);       // the whole block in uniform passes, straight into the other buffer
        } else {
            if (i + 1 < n) stash(buf ^ 1);
        }
        src_commit(src, i + 2, 0);
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    VBA_STAMP(1);
    // backward sweep, lane r = row r
    const int r = lane < 9 ? lane : 0;
    double x = zs[(size_t)(n - 1) * 9 + r];
    if (lane < 9) x_out[(size_t)(n - 1) * 9 + r] = x;
    double Xrow[9], zr = 0.0;
    auto fetch_row = [&](int i) {
        const double* X = Xs + (size_t)i * 81 + r * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) Xrow[j] = X[j];
        zr = zs[(size_t)i * 9 + r];
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
        if (lane < 9) x_out[(size_t)i * 9 + r] = x;
    }
    __threadfence_block();
    __syncthreads();
    VBA_STAMP(2);
}

// retraction of poses [lane, lane+64, ...) (BA_filtering.py:56-60); returns true if a non-finite step was seen
__device__ __forceinline__ bool retract_range(const DevView& V, size_t sb, int n, int first, int stride) {
    bool bad = false;
    for (int i = first; i < n; i += stride) {
        const double* dp = V.dpose + (sb + i) * 9;
        double d9[9], o[10];
#pragma unroll
        for (int j = 0; j < 9; ++j) { d9[j] = dp[j]; bad |= !(fabs(d9[j]) <= 1.79e308); }
        retract(V.states + (sb + i) * 10, d9, o);
        double* sn = V.states_new + (sb + i) * 10;
#pragma unroll
        for (int j = 0; j < 10; ++j) sn[j] = o[j];
    }
    return bad;
}

template <bool PIVOT>
__global__ __launch_bounds__(64) void k_solve(DevView V) {
    __shared__ double blk[2][256];
    const int w = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n = V.n[w];
    const int lane = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const double lam32 = (double)(float)sc.lam[V.par];      // torch.eye() is float32 (BA_filtering.py:54)
    if (lane == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    bool badp = false;
    const BandSource src{V.bands + sb * 243, V.rhs + sb * 9};
    chain_solve<PIVOT, true>(src, n, lam32, V.Xs + sb * 81, V.zs + sb * 9, V.dpose + sb * 9, blk, lane, badp);
    report_pivot<PIVOT>(badp, sc, lane, V.par);
    const bool bad = retract_range(V, sb, n, lane, 64);
    const unsigned long long anybad = __ballot(bad);
    if (lane == 0 && anybad) atomicOr(&sc.fl[V.par], 2u);
}

#ifdef VBA_VARIANTS   // measured dead ends kept for comparison builds (make VARIANTS=1): k_solve_forming, k_solve_packed
// Batched windows, full phase (VBA_OPT_FUSION bit 2): the walk forms the blocks itself.  The assembly kernel wrote 2 kB
// per pose that this kernel read straight back -- 8 GB per call at 4096 windows of 500 poses; here the wave keeps the
// inputs of three consecutive poses (141 doubles each, BA_reg 183) in an LDS ring, loads the next pose's while it
// eliminates, and every lane forms the four entries of the next block it used to load.  Same band_entry / rhs_entry,
// so the same system to the bit; ~200 more instructions per block step in a kernel that is issue-bound at four waves
// per SIMD, against a whole launch and its traffic.
template <bool REG>
struct RawSource {
    static constexpr int kIn = kAsmBase + (REG ? kAsmPrior : 0);
    static constexpr int kPer = (kIn + 63) / 64;
    static constexpr bool kLateFetch = true;
    const DevView& V;
    size_t sb;
    int n, lane;
    double sigma, inv_wmax;
    double* ring;               // [3][kIn]: pose i lives in slot i % 3
    AsmLanes lanes;
    mutable double hold[kPer];
    __device__ void prefetch(int i) const {
        if (i >= n) return;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int q = lane + 64 * k;
            hold[k] = q < kIn ? asm_input<REG>(V, sb + i, q, true) : 0.0;
        }
    }
    __device__ void commit(int i) const {
        if (i >= n) return;
        double* slot = ring + (i % 3) * kIn;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int q = lane + 64 * k;
            if (q < kIn) slot[q] = hold[k];
        }
    }
    __device__ double operator()(int i, int e) const {
        const AsmRow R = asm_row<REG>(ring + (i % 3) * kIn, ring + ((i + 2) % 3) * kIn, i, n, true, sigma, inv_wmax);
        if (e >= 243) return rhs_entry(R, e - 243);
        return band_entry(R, e / 81, (e % 81) / 9, e % 9);
    }
    // all 252 entries of block i into out (LDS), seven uniform passes of the wave (vba_asm_fast.h)
    __device__ void form(int i, double* out) const {
        asm_form_row<REG>(lanes, ring + (i % 3) * kIn, ring + ((i + 2) % 3) * kIn, i < n - 1, i > 0, sigma, inv_wmax, lane,
                          [&](int e, double v) { out[e] = v; });
    }
};

template <bool PIVOT, bool REG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_solve_forming(DevView V) {
    __shared__ double blk[2][256];
    __shared__ double ring[3 * RawSource<REG>::kIn];
    const int w = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n = V.n[w];
    const int lane = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const double lam32 = (double)(float)sc.lam[V.par];      // torch.eye() is float32 (BA_filtering.py:54)
    if (lane == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    bool badp = false;
    const RawSource<REG> src{V, sb, n, lane, V.prm.sigma, 1.0 / bits_f64(sc.wmax_bits[V.par]), ring, asm_lanes(lane), {}};
    src.prefetch(0); src.commit(0);
    src.prefetch(1); src.commit(1);
    for (int e = lane; e < 512; e += 64) (&blk[0][0])[e] = 0.0;     // (entries 252 .. 255 of a buffer are never formed)
    __syncthreads();
    chain_solve<PIVOT, true>(src, n, lam32, V.Xs + sb * 81, V.zs + sb * 9, V.dpose + sb * 9, blk, lane, badp);
    // the ring still holds poses n-3 .. n-1: the last diagonal block leaves for last_hessian (BA_filtering.py:97)
    for (int e = 81 + lane; e < 162; e += 64) V.lastD[(size_t)w * 81 + (e - 81)] = src(n - 1, e);
    report_pivot<PIVOT>(badp, sc, lane, V.par);
    const bool bad = retract_range(V, sb, n, lane, 64);
    const unsigned long long anybad = __ballot(bad);
    if (lane == 0 && anybad) atomicOr(&sc.fl[V.par], 2u);
}

// ---------------------------------------------------------------------------------------------- packed
// Many batched windows: three chains per wavefront (19 lanes each: 9 D' + 9 U + 1 right-hand side), so the ~550
// instructions of a block step serve three windows.  All windows of the handle must have the same pose count
// (checked by the host); windows that are already done ride along without storing.
constexpr int kPack = 3;

template <bool PIVOT>
__global__ __launch_bounds__(64) void k_solve_packed(DevView V) {
    __shared__ double blk[2][kPack][256];
    const int lane = threadIdx.x;
    const int g = lane / 19 < kPack ? lane / 19 : kPack - 1;
    const bool lane_ok = lane < 19 * kPack;
    const int ll = lane_ok ? lane - 19 * g : 19;          // 19 = no role
    const int gbase = 19 * g;
    const int w0 = blockIdx.x * kPack;
    const int wg = min(w0 + g, V.W - 1);
    const bool w_ok = w0 + g < V.W;
    const int n = V.n[w0];
    WinScalars& sc = V.sc[wg];
    const bool active = lane_ok && w_ok && !sc.done && VBA_WINDOW_RUNS(V, wg) && solver_mine<PIVOT>(V, sc);
    const size_t sb = (size_t)wg * V.n_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    if (active && ll == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    double a[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = 0.0;
    bool badp = false;
    double pre[kPack][4];
    auto fetch = [&](int i) {
#pragma unroll
        for (int q3 = 0; q3 < kPack; ++q3) {
            const size_t s3 = (size_t)min(w0 + q3, V.W - 1) * V.n_max + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = lane + 64 * q;
                pre[q3][q] = e < 243 ? V.bands[s3 * 243 + e] : (e < 252 ? V.rhs[s3 * 9 + (e - 243)] : 0.0);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q3 = 0; q3 < kPack; ++q3)
#pragma unroll
            for (int q = 0; q < 4; ++q) blk[buf][q3][lane + 64 * q] = pre[q3][q];
    };
    auto load_base = [&](const double* b, int db, double (&base)[9]) {
        const int ub = 9 - db;
        const bool isD = ll >= db && ll < db + 9, isU = ll >= ub && ll < ub + 9, isY = ll == 18;
        const int cc = isD ? ll - db : (isU ? ll - ub : 0);
        const double* p = isD ? b + 81 + cc : (isU ? b + 162 + cc : b + 243);
        const int stride = isY ? 1 : 9;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = (isD || isU || isY) ? p[r * stride] : 0.0;
            if (isD && r == cc) v += lam32;
            base[r] = v;
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const int buf = i & 1;
        if (i + 1 < n) fetch(i + 1);
        double base[9];
        const double* mine = blk[buf][g];
        if (buf == 0) {
            load_base(mine, 0, base);
            forward_step<0, 1, PIVOT, true, true>(i > 0 ? mine : nullptr, base, a, ll, badp, gbase);
        } else {
            load_base(mine, 9, base);
            forward_step<9, 1, PIVOT, true, true>(mine, base, a, ll, badp, gbase);
        }
        const int ub = buf == 0 ? 9 : 0;
        if (active) {
            if (ll >= ub && ll < ub + 9) {
                double* X = V.Xs + (sb + i) * 81 + (ll - ub);
#pragma unroll
                for (int r = 0; r < 9; ++r) X[r * 9] = a[r];
            } else if (ll == 18) {
                double* z = V.zs + (sb + i) * 9;
#pragma unroll
                for (int r = 0; r < 9; ++r) z[r] = a[r];
            }
        }
        if (i + 1 < n) stash(buf ^ 1);
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // backward sweep: lane ll < 9 of each group owns row ll
    const int r = ll < 9 ? ll : 0;
    double x = V.zs[(sb + n - 1) * 9 + r];
    if (active && ll < 9) V.dpose[(sb + n - 1) * 9 + r] = x;
    double Xrow[9], zr = 0.0;
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
        for (int j = 0; j < 9; ++j) v -= cur[j] * __shfl(x, gbase + j, kWave);
        x = v;
        if (active && ll < 9) V.dpose[(sb + i) * 9 + r] = x;
    }
    __threadfence_block();
    __syncthreads();
    // flags + retraction, one window after the other with the whole wave
    const unsigned long long badmask = __ballot(badp && active);
#pragma unroll
    for (int q3 = 0; q3 < kPack; ++q3) {
        const int w = w0 + q3;
        if (w >= V.W) break;
        WinScalars& s3 = V.sc[w];
        if (s3.done || !VBA_WINDOW_RUNS(V, w) || !solver_mine<PIVOT>(V, s3)) continue;
        const unsigned long long gm = ((1ull << 19) - 1ull) << (19 * q3);
        if (lane == 0 && (badmask & gm)) atomicOr(&s3.fl[V.par], PIVOT ? 4u : 8u);
        const bool bad = retract_range(V, (size_t)w * V.n_max, n, lane, 64);
        const unsigned long long anybad = __ballot(bad);
        if (lane == 0 && anybad) atomicOr(&s3.fl[V.par], 2u);
    }
}

#endif  // VBA_VARIANTS

// ---------------------------------------------------------------------------------------------- quad
// Many batched windows: FOUR chains per wavefront, one per row of 16 lanes.  The walk of one window per wave is bound by
// instruction issue (four waves per SIMD, each ~450 instructions per block step of which 19 lanes do anything), and most of
// those instructions are the v_readlane pairs that broadcast a pivot column -- which serve one window however many lanes are
// idle.  gfx90a and later can broadcast a lane inside every row of 16 with one DPP move (row_newbcast), so here lane c < 9
// of a row owns column c of D' (registers A) AND column c of U (registers B), lane 9 the right-hand side (in A): a pivot is
// 20 DPP moves + 18 multiply-adds for four windows instead of 20 readlanes + 9 multiply-adds for one.  The update
// D' = D - L X_prev is lane-local (X_prev's column c is this lane's B).  Same operations in the same order per entry as
// forward_step, so the same bits as k_solve.  Windows may differ in length; rows are independent (nothing crosses a row).
constexpr int kQuad = 4;

// One block step of four chains.  c = lane inside the row.  In: A = this lane's column of [I | z_{i-1}] (lane 9: z),
// B = column of X_{i-1}; baseA = column of [D_i + lam I | rhs_i], baseB = column of U_i.  Out: A = [I | z_i], B = X_i.
template <bool PIVOT, int K = 0>
__device__ __forceinline__ void quad_pivots(const double (&baseA)[9], double (&A)[9], double (&B)[9], int c, bool& bad) {
    if constexpr (K < 9) {
        double inv;
        if (PIVOT) {
            const int cnt = 9 - K;
            double cv[9], cs[9];
            int ci[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                if (r < cnt) { cs[r] = A[K + r]; cv[r] = fabs(cs[r]); ci[r] = K + r; }
            }
#pragma unroll
            for (int step = 1; step < 9; step *= 2) {
#pragma unroll
                for (int r = 0; r < 9; r += 2 * step) {
                    if (r + step < cnt) {
                        const bool take = cv[r + step] > cv[r];       // strict: the lowest row wins a tie
                        cv[r] = take ? cv[r + step] : cv[r];
                        cs[r] = take ? cs[r + step] : cs[r];
                        ci[r] = take ? ci[r + step] : ci[r];
                    }
                }
            }
            const double inv_l = fast_rcp(cs[0]);
            const int p = bcast_row16_i32<K>(ci[0]);
            inv = bcast_row16<K>(inv_l);
            if (!(fabs(inv) <= 1.79e308)) bad = true;
            const double ak = A[K], bk = B[K];
            double nk = ak, mk = bk;
#pragma unroll
            for (int r = K + 1; r < 9; ++r) {     // row swap K <-> p (p is uniform over the row of lanes), branch free
                const bool sel = (p == r);
                const double ar = A[r], br = B[r];
                nk = sel ? ar : nk;
                mk = sel ? br : mk;
                A[r] = sel ? ak : ar;
                B[r] = sel ? bk : br;
            }
            A[K] = nk;
            B[K] = mk;
        } else {
            bad = bad | ((c == K) & !(A[K] > 1e-10 * baseA[K]));
            inv = bcast_row16<K>(fast_rcp(A[K]));
        }
#ifdef VBA_QUAD_PLAIN_UPDATE
        double f[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) f[r] = (r != K) ? bcast_row16<K>(A[r]) : 0.0;
        A[K] = A[K] * inv;
        B[K] = B[K] * inv;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (r != K) {
                A[r] = fma(-f[r], A[K], A[r]);
                B[r] = fma(-f[r], B[K], B[r]);
            }
        }
#else
        A[K] = A[K] * inv;
        B[K] = B[K] * inv;
        dpp_rank1_9<K>(A, B);       // a_r = fma(-a_r[pivot lane], a_K, a_r), likewise b_r: the broadcast rides in the multiply-add
#endif
        quad_pivots<PIVOT, K + 1>(baseA, A, B, c, bad);
    }
}

// FORM (VBA_OPT_FUSION bit 2, default): the walk forms its blocks itself from the per-pose inputs (asm_form_row, the uniform
// passes of vba_asm_fast.h: the same system to the bit) -- no assembly launch, and the bands (2 kB per pose written and read
// back) never go through memory: per pose 0.8 kB of inputs instead.  The inputs of three consecutive poses of each window
// live in an LDS ring, the loads run kFwdDepth poses ahead in registers.
// The block is ONE wave: LDS operations of a wave execute in order, so what one lane wrote is there for the lane that reads
// it in a later instruction -- no s_barrier, and above all no s_waitcnt vmcnt(0), which __syncthreads() carries and which
// would make every block step wait for the loads it has just issued for four steps ahead.  The fence keeps the compiler
// from moving LDS accesses across.
__device__ __forceinline__ void quad_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool PIVOT, bool FORM, bool REG>
__global__ __launch_bounds__(64) void k_solve_quad(DevView V) {
    constexpr int kIn = kAsmBase + (REG ? kAsmPrior : 0);
    constexpr int kPerIn = FORM ? (kQuad * kIn + 63) / 64 : 1;
    __shared__ double blk[2][kQuad][256];
    __shared__ double ring[FORM ? 3 : 1][FORM ? kPerIn * 64 : 1];       // [slot][window q4 at q4 * kIn]; padded to whole passes of the wave
    const int lane = threadIdx.x;
    const int row = lane >> 4, c = lane & 15;
    const int w0 = blockIdx.x * kQuad;
    const int w = min(w0 + row, V.W - 1);
    WinScalars& sc = V.sc[w];
    const bool active = w0 + row < V.W && !sc.done && VBA_WINDOW_RUNS(V, w) && solver_mine<PIVOT>(V, sc);
    const int n = active ? V.n[w] : 0;
    int nmax = n;       // the longest chain of the four
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, kWave));
    nmax = __builtin_amdgcn_readfirstlane(nmax);        // (uniform: the loop bounds below are scalar branches)
    if (nmax == 0) return;
    const size_t sb = (size_t)w * V.n_max;
    const double lam32 = (double)(float)sc.lam[V.par];      // torch.eye() is float32 (BA_filtering.py:54)
    if (active && c == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    double A[9], B[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) A[j] = B[j] = 0.0;
    bool badp = false;
    // The 252 entries of block i of each of the four windows, coalesced: 16 loads per lane.  One wave per SIMD has nobody to
    // hide a round trip to memory behind, so the loads run kFwdDepth block steps ahead of their use (a ring of register
    // sets; the loop is unrolled by the depth so that the ring index is static).
#ifndef VBA_Q_FWD
#define VBA_Q_FWD 2
#endif
#ifndef VBA_Q_BWD
#define VBA_Q_BWD 8
#endif
#ifndef VBA_QX
#define VBA_QX 0
#endif
    constexpr int kFwdDepth = VBA_Q_FWD, kBwdDepth = VBA_Q_BWD;
    int nq[kQuad];
#pragma unroll
    for (int q4 = 0; q4 < kQuad; ++q4) nq[q4] = __shfl(n, 16 * q4, kWave);
    double pre[FORM ? 1 : kFwdDepth][kQuad][4];
    auto fetch = [&](int i, double (&dst)[kQuad][4]) {
#pragma unroll
        for (int q4 = 0; q4 < kQuad; ++q4) {
            const int wq = min(w0 + q4, V.W - 1);
            const bool have = i < nq[q4];
            const size_t s4 = (size_t)wq * V.n_max + (have ? i : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = lane + 64 * q;        // (q < 3: a band entry; q == 3: lanes 51 .. 59 the right-hand side)
                const double* src = e < 243 ? V.bands + s4 * 243 + e : V.rhs + s4 * 9 + min(e - 243, 8);
                const double v = *src;
                dst[q4][q] = (have && e < 252) ? v : 0.0;
            }
        }
    };
    auto stash = [&](int buf, const double (&src)[kQuad][4]) {
#pragma unroll
        for (int q4 = 0; q4 < kQuad; ++q4)
#pragma unroll
            for (int q = 0; q < 4; ++q) blk[buf][q4][lane + 64 * q] = src[q4][q];
    };
    // FORM: the inputs of pose i of the four windows (kIn doubles each), kPerIn loads per lane
    double hold[kFwdDepth][kPerIn];
    const double inv_wmax4[1] = {1.0 / bits_f64(sc.wmax_bits[V.par])};        // (of this lane's own window)
    // where input q = lane + 64 k of the staged layout (vba_asm.h: asm_input) lives: base pointer of pose 0 and stride per
    // pose, decoded ONCE -- a per-element `if (q < 21) ... else if (q < 27) ...` ladder inside the walk is a few thousand
    // basic blocks with a wait for memory at every join
    const double* inbase[kPerIn];
    int instride[kPerIn], inn[kPerIn], inprior[kPerIn];
#pragma unroll
    for (int k = 0; k < kPerIn; ++k) {
        const int q = lane + 64 * k;
        const int q4 = min(q / kIn, kQuad - 1), e = q - q4 * kIn;
        const size_t pb = (size_t)min(w0 + q4, V.W - 1) * V.n_max;
        const double* bp = V.Hraw + pb * 21 + e;
        int st = 21;
        if (e >= 21) { bp = V.braw + pb * 6 + (e - 21); st = 6; }
        if (e >= 27) { bp = V.Phi + pb * 36 + (e - 27); st = 36; }
        if (e >= 63) { bp = V.rorb + pb * 6 + (e - 63); st = 6; }
        if (e >= 69) { bp = V.qgrad + pb * 3 + (e - 69); st = 3; }
        if (e >= 72) { bp = V.Hd + pb * 9 + (e - 72); st = 9; }
        if (e >= 81) { bp = V.Hu + pb * 9 + (e - 81); st = 9; }
        if (e >= 90) { bp = V.Hl + pb * 9 + (e - 90); st = 9; }
        if (REG && e >= 99) { bp = V.prior_H + pb * 36 + (e - 99); st = 36; }
        inprior[k] = (REG && e >= 135 && e < kIn) ? (e - 135) : -1;
        if (REG && e >= 135) { bp = V.prior_H + pb * 36; st = 36; }
        inbase[k] = bp;
        instride[k] = st;
        const int nn = q4 == 0 ? nq[0] : (q4 == 1 ? nq[1] : (q4 == 2 ? nq[2] : nq[3]));
        inn[k] = (q < kQuad * kIn) ? nn : 0;
    }
    auto in_fetch = [&](int i, double (&dst)[kPerIn]) {
#pragma unroll
        for (int k = 0; k < kPerIn; ++k) {
            // (unconditional load from a clamped pose index, then a select: no branch)
            const int ic = max(min(i, inn[k] - 1), 0);
            const double v = inbase[k][(size_t)ic * instride[k]];
            dst[k] = i < inn[k] ? v : 0.0;
        }
        if (REG) {      // the staged prior residual is a computed value: component e - 135 of H [p_prior - p ; v_prior - v]
#pragma unroll
            for (int k = 0; k < kPerIn; ++k) {
                if (inprior[k] >= 0 && i < inn[k]) {
                    const int q4 = (lane + 64 * k) / kIn;
                    dst[k] = asm_input<REG>(V, (size_t)min(w0 + q4, V.W - 1) * V.n_max + i, 135 + inprior[k], true);
                }
            }
        }
    };
    auto in_commit = [&](int i, const double (&src)[kPerIn]) {
        double* slot = ring[i % 3];
#pragma unroll
        for (int k = 0; k < kPerIn; ++k) slot[lane + 64 * k] = src[k];
    };
    // (The formation below is asm_form_columns of vba_asm_fast.h written out in place: called as the function, with the column
    // of L_j handed back in registers and stored afterwards, this kernel took 3.5 ms instead of 2.3 -- fewer instructions,
    // worse order, and one wave per SIMD has nothing to cover that with.  The two are the same arithmetic: the fusion tests
    // compare them bit for bit.)
    // FORM: block j of this lane's window, column-wise -- the lane's column of [D_j | rhs_j] (nextA, undamped) and of U_j
    // (nextB) straight into registers, its column of L_j into blk[buf] (every lane of the row reads all of L_j).  Entry by
    // entry the operations of band_entry / rhs_entry (vba_math.h) in their order, so the same system to the bit; what
    // differs between the lanes (rotation column or not, which Phi column, right-hand side) is data.
    double nextA[9], nextB[9], lastDcol[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) lastDcol[a] = 0.0;
    const bool is_col = c < 9, is_rhs = c == 9;
    const bool rotc = c >= 3 && c < 6, nonrot = is_col && !rotc;
    const int pcl = c < 3 ? c : (nonrot ? c - 3 : 0);          // column of Phi / row of F that this lane's state slot maps to
    const int crl = rotc ? c - 3 : 0;
    const double fvc = c < 3 ? -1.0 : -kVelCoeff;
    const double Dcl = pcl < 3 ? 1.0 : kVelCoeff;
    auto form = [&](int j, int buf) {
        const double* me = ring[j % 3] + row * kIn;
        const double* pv = ring[(j + 2) % 3] + row * kIn;
        double* Lout = blk[buf][row];
        const bool live = j < n, has_next = j < n - 1, has_prev = j > 0 && live;
        const double sigma = V.prm.sigma, iw = inv_wmax4[0];       // (inv_wmax4[0]: this lane's own window, see below)
        const double fs[2] = {vba_mul(-1.0, sigma), vba_mul(-kVelCoeff, sigma)};
        // Every LDS read below is UNCONDITIONAL -- the address is selected, one load is made, the value is selected.
        // (`cond ? lds[i] : 0` compiles to a masked load in a basic block of its own with a full wait behind it; a few dozen
        // of those per block step were most of this kernel's time.)
        auto ld = [](const double* p) { return *p; };
        // the lane's second factor of the J_f^T Sigma J_f sums: its column of E_j = D Phi_j, or r_orb (right-hand side)
        double X[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double Dr = r < 3 ? 1.0 : kVelCoeff;
            const double xv = ld(me + (is_rhs ? 63 + r : 27 + 6 * r + pcl));
            const double e = vba_mul(Dr, xv);
            X[r] = nonrot ? e : (is_rhs ? xv : 0.0);
        }
        double Xp[6];
        if (REG) {
#pragma unroll
            for (int k = 0; k < 6; ++k) Xp[k] = ld(me + (is_rhs ? 135 + k : 99 + k * 6 + pcl));
        }
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const bool rota = a >= 3 && a < 6;
            const int pa = a < 3 ? a : a - 3;           // (non-rotation a)
            double v = 0.0;
            if (a < 6) {
                const int idx = is_rhs ? 21 + a : sym6(a, c < 6 ? c : 0);
                const double h = vba_mul(ld(me + idx), iw);
                v = (is_rhs || c < 6) ? h : 0.0;
            }
            {
                double sdyn = 0.0;
                if (!rota) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const double Dr = r < 3 ? 1.0 : kVelCoeff;
                        sdyn = fma(vba_mul(vba_mul(Dr, ld(me + 27 + 6 * r + pa)), sigma), X[r], sdyn);
                    }
                }
                const double t = vba_add(v, is_rhs ? -sdyn : sdyn);
                v = has_next ? t : v;
            }
            if (!rota) {
                const double fsa = fs[a < 3 ? 0 : 1], fva = a < 3 ? -1.0 : -kVelCoeff;
                const double zr = ld(pv + 63 + pa);
                const double z = is_rhs ? -zr : fva;
                const double t = fma(fsa, z, v);
                v = (has_prev && (is_rhs || c == a)) ? t : v;
            } else {
                const double yv = ld(me + (is_rhs ? 69 + (a - 3) : 72 + 3 * (a - 3) + crl));
                const double t = fma(sigma, is_rhs ? -yv : yv, v);
                v = (is_rhs || rotc) ? t : v;
            }
            if (REG && !rota) {
                double sp = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) sp = fma(ld(me + 99 + k * 6 + pa), Xp[k], sp);
                const double t = vba_add(v, sp);
                v = (is_rhs || nonrot) ? t : v;
            }
            nextA[a] = (live && c < 10) ? v : 0.0;
            // super-diagonal column and sub-diagonal column
            double u, l;
            if (!rota) {
                const double fsa = fs[a < 3 ? 0 : 1];
                const double eu = vba_mul(Dcl, ld(me + 27 + 6 * pcl + pa));              // E_entry(Phi_j, F_row(c), a)
                u = vba_mul(vba_mul(eu, sigma), fvc);
                const double Dra = pa < 3 ? 1.0 : kVelCoeff;
                const double el = vba_mul(Dra, ld(pv + 27 + 6 * pa + pcl));              // E_entry(Phi_{j-1}, F_row(a), c)
                l = vba_mul(fsa, el);
                u = nonrot ? u : 0.0;
                l = nonrot ? l : 0.0;
            } else {
                u = vba_mul(sigma, ld(me + 81 + 3 * (a - 3) + crl));
                l = vba_mul(sigma, ld(me + 90 + 3 * (a - 3) + crl));
                u = rotc ? u : 0.0;
                l = rotc ? l : 0.0;
            }
            nextB[a] = (has_next && is_col) ? u : 0.0;
            Lout[is_col ? a * 9 + c : 96 + 9 * (c - 9) + a] = has_prev ? l : 0.0;      // (lanes without a column: a spare slot each)
            lastDcol[a] = j == n - 1 ? nextA[a] : lastDcol[a];      // last_hessian (BA_filtering.py:97): kept, stored after the walk
        }
#ifdef VBA_DEBUG_FORM
        if (active && live) {
            const size_t pp = sb + j;
#pragma unroll
            for (int a = 0; a < 9; ++a) {
                if (is_col) { V.bands[pp * 243 + 81 + a * 9 + c] = nextA[a]; V.bands[pp * 243 + 162 + a * 9 + c] = nextB[a]; V.bands[pp * 243 + a * 9 + c] = has_prev ? Lout[a * 9 + c] : 0.0; }
                if (is_rhs) V.rhs[pp * 9 + a] = nextA[a];
            }
        }
#endif
    };
    if (FORM) {
        for (int e = lane; e < 3 * kPerIn * 64; e += 64) (&ring[0][0])[e] = 0.0;
        for (int e = lane; e < 2 * kQuad * 256; e += 64) (&blk[0][0][0])[e] = 0.0;      // (only the sub-diagonal block goes through LDS)
        quad_sync();
        double first[2][kPerIn];
        in_fetch(0, first[0]);
        in_fetch(1, first[1]);
#pragma unroll
        for (int k = 0; k < kFwdDepth; ++k) in_fetch(2 + k, hold[(2 + k) % kFwdDepth]);
        in_commit(0, first[0]);
        in_commit(1, first[1]);
        quad_sync();
        form(0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < kFwdDepth; ++k) fetch(k, pre[k]);       // (beyond the end of a chain: zeros, no access)
        stash(0, pre[0]);
        fetch(kFwdDepth, pre[0]);
    }
    quad_sync();
    for (int i0 = 0; i0 < nmax; i0 += kFwdDepth) {
#pragma unroll
        for (int k = 0; k < kFwdDepth; ++k) {
            const int i = i0 + k;
            if (i >= nmax) break;
            const int buf = i & 1;
            const double* b = blk[buf][row];
            // this lane's column of [D_i + lam I | rhs_i] and of U_i
            double baseA[9], baseB[9];
            if (FORM) {
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    baseA[r] = r == c ? nextA[r] + lam32 : nextA[r];
                    baseB[r] = nextB[r];
                }
                // Block i + 1 is formed HERE, in front of the elimination of block i and in the same basic block (nothing
                // below branches): its LDS reads and multiply-adds are independent of the pivot chain, and interleaved with it
                // they fill the chain's bubbles -- behind the elimination they were a second latency-bound phase.  Beyond the
                // end of every chain it forms zeros / harmless values (no branch to skip it).
                const int kh = (k + 2) % kFwdDepth;     // (static once the loop is unrolled)
                if (!(VBA_QX & 2)) form(i + 1, buf ^ 1);
                in_commit(i + 2, hold[kh]);
                in_fetch(i + 2 + kFwdDepth, hold[kh]);
            } else {
                const double* pa = c < 9 ? b + 81 + c : b + 243;
                const int stride = c < 9 ? 9 : 1;
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    double v = c < 10 ? pa[r * stride] : 0.0;
                    if (r == c) v += lam32;
                    baseA[r] = v;
                    baseB[r] = c < 9 ? b[162 + r * 9 + c] : 0.0;
                }
            }
            // D' = D - L X_{i-1}, y = g - L z_{i-1}: lane local (sparse L: [pp 0 pv; 0 rr 0; vp 0 vv])
            double xp[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) xp[j] = c < 9 ? B[j] : (c == 9 ? A[j] : 0.0);
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                double v = baseA[r];
                if (FORM || i > 0) {        // (FORM: the sub-diagonal block of block 0 is formed as zeros)
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const bool rot_r = (r >= 3 && r < 6), rot_j = (j >= 3 && j < 6);
                        if (rot_r == rot_j) v = fma(-b[r * 9 + j], xp[j], v);      // LDS read, one address per row of lanes
                    }
                }
                A[r] = v;
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) B[r] = baseB[r];
            if (!(VBA_QX & 1)) quad_pivots<PIVOT>(baseA, A, B, c, badp);
            {   // column c of X_i, or z_i.  No branch: a lane with nothing to store writes into the chunk-solution scratch of its
                // window (unused by this driver) instead
                const bool st_ok = active && i < n && c < 10;
                double* dst = c < 9 ? V.Xs + (sb + i) * 81 + c : V.zs + (sb + i) * 9;
                dst = st_ok ? dst : V.csol + sb * 171 + (size_t)lane * 9;
                const int stride = st_ok && c < 9 ? 9 : 1;
#pragma unroll
                for (int r = 0; r < 9; ++r) dst[r * stride] = c < 9 ? B[r] : A[r];
            }
            if (!FORM) {
                // block i + 1 goes to the other buffer (its loads were issued kFwdDepth steps ago); its register set takes the
                // loads of block i + 1 + kFwdDepth
                const int kn = (k + 1) % kFwdDepth;
                if (i + 1 < nmax) {
                    stash(buf ^ 1, pre[kn]);
                    fetch(i + 1 + kFwdDepth, pre[kn]);
                }
            }
            quad_sync();
        }
    }
    quad_sync();
    if (FORM && active && is_col) {
#pragma unroll
        for (int a = 0; a < 9; ++a) V.lastD[(size_t)w * 81 + a * 9 + c] = lastDcol[a];
    }
    // backward sweep: lane c < 9 of a row of lanes owns row c of its window; the rows of X run kBwdDepth steps ahead
    const int r = c < 9 ? c : 0;
    const bool mine = active && c < 9;
    double x = 0.0;
    double Xring[kBwdDepth][10];
    auto fetch_row = [&](int i, double (&dst)[10]) {
#pragma unroll
        for (int j = 0; j < 10; ++j) dst[j] = 0.0;
        if (mine && i >= 0 && i < n) {
            const double* X = V.Xs + (sb + i) * 81 + r * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) dst[j] = X[j];
            dst[9] = V.zs[(sb + i) * 9 + r];
        }
    };
#pragma unroll
    for (int k = 0; k < kBwdDepth; ++k) fetch_row(nmax - 1 - k, Xring[k]);
    for (int i0 = nmax - 1; i0 >= 0; i0 -= kBwdDepth) {
#pragma unroll
        for (int k = 0; k < kBwdDepth; ++k) {
            const int i = i0 - k;
            if (i < 0) break;
            double cur[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) cur[j] = Xring[k][j];
            fetch_row(i - kBwdDepth, Xring[k]);
            double v = cur[9];
            const double xb[9] = {bcast_row16<0>(x), bcast_row16<1>(x), bcast_row16<2>(x), bcast_row16<3>(x), bcast_row16<4>(x),
                                  bcast_row16<5>(x), bcast_row16<6>(x), bcast_row16<7>(x), bcast_row16<8>(x)};
            if (i < n - 1) {
#pragma unroll
                for (int j = 0; j < 9; ++j) v -= cur[j] * xb[j];
            }
            if (i < n) {        // (i == n - 1: x = z, the last block of this chain)
                x = v;
                if (mine) V.dpose[(sb + i) * 9 + r] = x;
            }
        }
    }
    quad_sync();
    // flags + retraction, every row of lanes for its own window
    const unsigned long long badmask = __ballot(badp && active);
    const unsigned long long rowmask = 0xffffull << (16 * row);
    if (active && c == 0 && (badmask & rowmask)) atomicOr(&sc.fl[V.par], PIVOT ? 4u : (8u | 16u));
    bool bad = false;
    if (active) bad = retract_range(V, sb, n, c, 16);
    const unsigned long long anybad = __ballot(bad);
    if (active && c == 0 && (anybad & rowmask)) atomicOr(&sc.fl[V.par], 2u);
}

// ================================================================================================== partitioned
// chunk c of a window covers blocks [c s, min((c+1) s, n)); its last block is a separator unless c is the last
// chunk.  s >= 2, so every chunk has at least one interior block.
__device__ __forceinline__ void chunk_range(int c, int s, int n, int& a, int& b, bool& has_sep) {
    a = c * s;
    const int end = min((c + 1) * s, n);        // exclusive
    has_sep = end < n;
    b = has_sep ? end - 2 : end - 1;            // last interior block
}

// Reduced system over the separators of a chain given by `Inner`: row q couples separators q-1, q, q+1 (block
// j = (q+1) s - 1 of the inner chain):
//   sub = -L_j Vhat_{j-1},  diag = D_j - L_j What_{j-1} - U_j Vhat_{j+1},  super = -U_j What_{j+1},
//   rhs = g_j - L_j yhat_{j-1} - U_j yhat_{j+1};   the products were left in cL / cR by the chunk waves.
// It is again a block source, so the same chunk elimination can be applied to it (second level).
template <class Inner>
struct ReducedSource {
    Inner inner;
    const double* cL;       // [ns][9][19]
    const double* cR;
    int s;
    __device__ double operator()(int q, int e) const {
        const int j = (q + 1) * s - 1;
        const double* l = cL + (size_t)q * 171;         // [row][19 columns]: row-major like the bands, so that a
        const double* r_ = cR + (size_t)q * 171;        // consumer walking e reads runs of 9 contiguous doubles
        if (e >= 243) {
            const int r = e - 243;
            return inner(j, e) - l[r * 19] - r_[r * 19];
        }
        const int which = e / 81, r = (e % 81) / 9, cc = e % 9;
        if (which == 0) return -l[r * 19 + 1 + cc];
        if (which == 2) return -r_[r * 19 + 10 + cc];
        return inner(j, e) - l[r * 19 + 10 + cc] - r_[r * 19 + 1 + cc];
    }
};

// Eliminates the interior of chunk c of a chain of n blocks with 19 right-hand sides: column 0 = g, 1..9 = L_a
// (coupling to the left separator), 10..18 = U_b (coupling to the right separator).  csol[i][col][r] receives
// T^{-1} of them for every interior block i; cL[c] / cR[c-1] receive L_j / U_j times the solutions next to the
// chunk's two separators (what the reduced system needs).
template <bool PIVOT, bool SPARSE_L, class Src>
__device__ __forceinline__ void chunk_eliminate(const Src& src, int n, int s, int c, double lam32, double* csol, double* cL,
                                                double* cR, double* smem, int lane, bool& zero_pivot) {
    int a0, b0;
    bool has_sep;
    chunk_range(c, s, n, a0, b0, has_sep);
    const int len = b0 - a0 + 1;
    double (*blk)[256] = reinterpret_cast<double (*)[256]>(smem);           // [2][256]
    double* Xb = smem + 512;                                                 // [s][81]
    double* Zb = Xb + (size_t)s * 81;                                        // [s][19][9]
    double* Cm = Zb + (size_t)s * 171;                                       // [2][81]: L of the right separator, U of the left one
    {   // (the loads of a lane before its first store)
        double cm[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = lane + 64 * q;
            double v = 0.0;
            if (e < 81) { if (has_sep) v = src(b0 + 1, e); }
            else if (e < 162 && c > 0) v = src(a0 - 1, 162 + (e - 81));
            cm[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = lane + 64 * q;
            if (e < 162) Cm[e] = cm[q];
        }
    }
    double a[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = 0.0;
    double pre[4];
    auto fetch = [&](int i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q;
            pre[q] = e < 252 ? src(i, e) : 0.0;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) blk[buf][lane + 64 * q] = pre[q];
    };
    // lanes: D/U groups in 0..17 (alternating), V = 18..26, W = 27..35, y = 36  -> rhs column order in Zb: y, V, W
    const bool isV = lane >= 18 && lane < 27, isW = lane >= 27 && lane < 36, isY = lane == 36;
    const int zcol = isY ? 0 : (isV ? 1 + (lane - 18) : (isW ? 10 + (lane - 27) : 0));
    // one LDS address per lane and role (selecting among loaded values would make every lane load all five)
    auto load_base = [&](const double* b, int db, bool first, bool last, double (&base)[9]) {
        const int ub = 9 - db;
        const bool isD = lane >= db && lane < db + 9, isU = lane >= ub && lane < ub + 9;
        const int cc = isD ? lane - db : (isU ? lane - ub : (isV ? lane - 18 : (isW ? lane - 27 : 0)));
        const bool ok = isD || isU || isY || (isV && first) || (isW && last);
        const int off = isD ? 81 + cc : ((isU || isW) ? 162 + cc : (isY ? 243 : cc));
        const int stride = isY ? 1 : 9;
        const double* p = b + (ok ? off : 0);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = p[r * stride];
            v = ok ? v : 0.0;
            if (isD && r == cc) v += lam32;
            base[r] = v;
        }
    };
    fetch(a0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < len; ++t) {
        const int buf = t & 1;
        if (t + 1 < len) fetch(a0 + t + 1);
        double base[9];
        if (buf == 0) {
            load_base(blk[0], 0, t == 0, t == len - 1, base);
            forward_step<0, 19, PIVOT, SPARSE_L>(t > 0 ? blk[0] : nullptr, base, a, lane, zero_pivot);
        } else {
            load_base(blk[1], 9, false, t == len - 1, base);
            forward_step<9, 19, PIVOT, SPARSE_L>(blk[1], base, a, lane, zero_pivot);
        }
        const int ub = buf == 0 ? 9 : 0;
        if (lane >= ub && lane < ub + 9) {
#pragma unroll
            for (int r = 0; r < 9; ++r) Xb[(size_t)t * 81 + r * 9 + (lane - ub)] = a[r];
        } else if (isV || isW || isY) {
#pragma unroll
            for (int r = 0; r < 9; ++r) Zb[((size_t)t * 19 + zcol) * 9 + r] = a[r];
        }
        if (t + 1 < len) stash(buf ^ 1);
        __syncthreads();
    }
    // Backward sweep for the 19 right-hand sides on the matrix cores: x_t (9 x 19) = Z_t - X_t x_{t+1}.
    // v_mfma_f64_16x16x4 leaves C[(l >> 4) + 4 i][l & 15] in register i of lane l, and wants B[4 s + (l >> 4)][l & 15] in
    // k-step s: register s of the previous result IS the B operand of k-step s, so x never moves between steps.  Two
    // column tiles (columns 0..15, 16..18), three k-steps (rows 9..11 of x stay zero), one LDS read per A element.
    const int lr = lane & 15, lk = lane >> 4;
    auto load_Z = [&](int t, int tile) {
        vf4 z;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lk + 4 * i, col = 16 * tile + lr;
            const bool ok = row < 9 && col < 19;
            const double v = Zb[ok ? ((size_t)t * 19 + col) * 9 + row : 0];
            z[i] = ok ? v : 0.0;
        }
        return z;
    };
    auto mul_sub = [&](const double* M, double sign, vf4& acc0, vf4& acc1, const vf4& b0, const vf4& b1) {
        // acc += sign * M (9 x 9, row major in LDS) * b
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int k = 4 * st + lk;
            const bool ok = lr < 9 && k < 9;
            const double m = M[ok ? lr * 9 + k : 0];
            const double am = ok ? sign * m : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, b0[st], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, b1[st], acc1, 0, 0, 0);
        }
    };
    auto store_cols = [&](double* dst, size_t col_stride, size_t row_stride, const vf4& v0, const vf4& v1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lk + 4 * i;
            if (row < 9) {
                dst[(size_t)lr * col_stride + (size_t)row * row_stride] = v0[i];
                if (lr < 3) dst[(size_t)(16 + lr) * col_stride + (size_t)row * row_stride] = v1[i];
            }
        }
    };
    double* out = csol + (size_t)a0 * 171;
    vf4 x0 = load_Z(len - 1, 0), x1 = load_Z(len - 1, 1);
    store_cols(out + (size_t)(len - 1) * 171, 9, 1, x0, x1);
    // contribution of this chunk to its right separator j = b+1:  L_j [yhat_b | Vhat_b | What_b]   ([row][19 columns])
    if (has_sep) {
        vf4 p0 = {0.0, 0.0, 0.0, 0.0}, p1 = {0.0, 0.0, 0.0, 0.0};
        mul_sub(Cm, 1.0, p0, p1, x0, x1);
        store_cols(cL + (size_t)c * 171, 1, 19, p0, p1);
    }
    for (int t = len - 2; t >= 0; --t) {
        vf4 n0 = load_Z(t, 0), n1 = load_Z(t, 1);
        mul_sub(Xb + (size_t)t * 81, -1.0, n0, n1, x0, x1);
        x0 = n0;
        x1 = n1;
        store_cols(out + (size_t)t * 171, 9, 1, x0, x1);
    }
    // contribution to the left separator j = a-1:  U_j [yhat_a | Vhat_a | What_a]
    if (c > 0) {
        vf4 p0 = {0.0, 0.0, 0.0, 0.0}, p1 = {0.0, 0.0, 0.0, 0.0};
        mul_sub(Cm + 81, 1.0, p0, p1, x0, x1);
        store_cols(cR + (size_t)(c - 1) * 171, 1, 19, p0, p1);
    }
}

// The same elimination by TWO waves per chunk that meet in the middle.  The elimination of a chunk is a chain of dependent
// block steps (~2 us each on a single wave) and in latency mode that chain IS the time of the kernel: wave 0 eliminates
// blocks a .. m-1 left to right, wave 1 blocks b .. m+1 right to left -- the same step on the mirrored chain (sub and super
// diagonal swap roles; the coupling to the right separator enters at its first block the way the left one enters wave 0's)
// -- then wave 0 solves block m with both neighbours folded in,
//     (D_m - L_m X_{m-1} - U_m X'_{m+1}) x_m = g_m - L_m z_{m-1} - U_m z'_{m+1}      (19 right-hand sides),
// and both waves substitute outwards from x_m on the matrix cores.  Half the dependent steps (7 interior blocks: 3 + 1
// + the two substitutions side by side instead of 7 + 6).  Chunks with fewer than 3 interior blocks take the one-wave
// path.  tid: 0 .. 127.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__host__ __device__ constexpr int twosided_half(int s) { return (s + 1) / 2; }
__host__ __device__ constexpr int twosided_region(int s) { return 512 + twosided_half(s) * (81 + 171); }
__host__ __device__ constexpr int twosided_lds_doubles(int s) {
    const int two = 2 * twosided_region(s) + 162 + 256 + 171, one = 512 + s * 252 + 162;
    return two > one ? two : one;
}

template <bool PIVOT, bool SPARSE_L, class Src>
__device__ __forceinline__ void chunk_eliminate_twosided(const Src& src, int n, int s, int c, double lam32, double* csol, double* cL,
                                                         double* cR, double* smem, int tid, bool& zero_pivot) {
    int a0, b0;
    bool has_sep;
    chunk_range(c, s, n, a0, b0, has_sep);
    const int len = b0 - a0 + 1;
    const int side = tid >> 6, lane = tid & 63;
    if (len < 3) {      // (uniform over the workgroup) nothing to share: one wave, the other leaves
        if (side == 0) chunk_eliminate<PIVOT, SPARSE_L>(src, n, s, c, lam32, csol, cL, cR, smem, lane, zero_pivot);
        return;
    }
    const int lenL = len / 2, lenR = len - 1 - lenL, m = a0 + lenL;
    const int lenS = side ? lenR : lenL;
    const int hs = twosided_half(s), RS = twosided_region(s);
    double* reg0 = smem;
    double* reg1 = smem + RS;
    double* reg = side ? reg1 : reg0;
    double (*blk)[256] = reinterpret_cast<double (*)[256]>(reg);            // [2][256]
    double* Xb = reg + 512;                                                  // [hs][81]
    double* Zb = Xb + (size_t)hs * 81;                                       // [hs][19][9]
    double* Cm = smem + 2 * (size_t)RS;                                      // [2][81]: L of the right separator, U of the left one
    double* blkM = Cm + 162;                                                 // block m
    double* xm = blkM + 256;                                                 // [19][9] its solution
    // entry e of real block i as this side's sweep sees it
    auto entry = [&](int i, int e) {
        const int ee = side ? (e < 81 ? e + 162 : ((e >= 162 && e < 243) ? e - 162 : e)) : e;
        return src(i, ee);
    };
    auto block_of = [&](int t) { return side ? b0 - t : a0 + t; };
    {   // (both loads of a lane before its first store)
        double cm[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            cm[q] = e >= 81 ? 0.0 : (side ? (has_sep ? src(b0 + 1, e) : 0.0) : (c > 0 ? src(a0 - 1, 162 + e) : 0.0));
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            if (e < 81) Cm[side ? e : 81 + e] = cm[q];
        }
    }
    double mid[4] = {0.0, 0.0, 0.0, 0.0};
    if (side == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q;
            mid[q] = e < 252 ? src(m, e) : 0.0;
        }
    }
    double a[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = 0.0;
    double pre[4];
    auto fetch = [&](int t) {
        const int i = block_of(t);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q;
            pre[q] = e < 252 ? entry(i, e) : 0.0;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) blk[buf][lane + 64 * q] = pre[q];
    };
    // lanes as in chunk_eliminate: D/U groups in 0..17 (alternating), V = 18..26, W = 27..35, y = 36.  "V" is the coupling
    // that enters at the sweep's FIRST block: the left separator for wave 0, the right one for wave 1 (columns swapped
    // back when the results are stored); the W columns stay zero during the sweeps.
    const bool isV = lane >= 18 && lane < 27, isW = lane >= 27 && lane < 36, isY = lane == 36;
    const int zcol = isY ? 0 : (isV ? 1 + (lane - 18) : (isW ? 10 + (lane - 27) : 0));
    auto load_base = [&](const double* b, int db, bool first, double (&base)[9]) {
        const int ub = 9 - db;
        const bool isD = lane >= db && lane < db + 9, isU = lane >= ub && lane < ub + 9;
        const int cc = isD ? lane - db : (isU ? lane - ub : (isV ? lane - 18 : 0));
        const bool ok = isD || isU || isY || (isV && first);
        const int off = isD ? 81 + cc : (isU ? 162 + cc : (isY ? 243 : cc));
        const int stride = isY ? 1 : 9;
        const double* p = b + (ok ? off : 0);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = p[r * stride];
            v = ok ? v : 0.0;
            if (isD && r == cc) v += lam32;
            base[r] = v;
        }
    };
    // Row layout of the unpivoted path (cr_pivots_dpp): every row of 16 lanes holds D' in lanes 0..8 (the same values in
    // all four rows) and seven of the 28 columns [U | y V W] in lanes 9..15; a pivot broadcasts inside the row (DPP).  What
    // the alternating lane groups of forward_step got for free -- X_{t-1}'s column c already sitting in the lane that forms
    // D'_t's column c -- comes from Xb in LDS here (written for the outward substitution anyway).
#ifndef VBA_CHUNK_READLANE
    constexpr bool kRows = !PIVOT;
#else
    constexpr bool kRows = false;
#endif
    const int rrow = lane >> 4, rc = lane & 15;
    const int ro = rrow * 7 + (rc - 9);                 // column of [U | y V W] of a lane with rc >= 9
    const bool rD = rc < 9, rU = !rD && ro < 9, rR = !rD && ro >= 9;
    const int rz = rR ? ro - 9 : 0;                     // column of Zb: 0 = y, 1..9 = V, 10..18 = W
    auto rows_step = [&](const double* b, const double* Lmat, const double* Xprev, bool first) {
        // base: this lane's column of [D + lam I | U | y | V (first block only)]
        const bool ok = rD || rU || (rR && (rz == 0 || (first && rz < 10)));
        const int off = rD ? 81 + rc : (rU ? 162 + ro : (rz == 0 ? 243 : rz - 1));
        const int stride = (rR && rz == 0) ? 1 : 9;
        const double* p = b + (ok ? off : 0);
        double base[9], xp[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = p[r * stride];
            v = ok ? v : 0.0;
            if (rD && r == rc) v += lam32;
            base[r] = v;
        }
        if (Lmat) {
            // carried column: X_{t-1}[:, c] for the D lanes (from LDS), this lane's own z_{t-1} for the right-hand sides
            const double* xs = Xprev + (rD ? rc : 0);
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double xv = xs[j * 9];
                xp[j] = rD ? xv : (rR ? a[j] : 0.0);
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                double v = base[r];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const bool rot_r = (r >= 3 && r < 6), rot_j = (j >= 3 && j < 6);
                    if (!SPARSE_L || rot_r == rot_j) v = fma(-Lmat[r * 9 + j], xp[j], v);
                }
                a[r] = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 9; ++r) a[r] = base[r];
        }
        bool mybad = false;
        cr_pivots_dpp<0>(base, a, rc, mybad);
        zero_pivot = zero_pivot | (rD & mybad);
    };
    fetch(0);
    stash(0);
    wave_sync_lds();
    for (int t = 0; t < lenS; ++t) {
        const int buf = t & 1;
        if (t + 1 < lenS) fetch(t + 1);
        if constexpr (kRows) {
            rows_step(blk[buf], t > 0 ? blk[buf] : nullptr, Xb + (size_t)(t > 0 ? t - 1 : 0) * 81, t == 0);
            if (rU) {
#pragma unroll
                for (int r = 0; r < 9; ++r) Xb[(size_t)t * 81 + r * 9 + ro] = a[r];
            } else if (rR) {
#pragma unroll
                for (int r = 0; r < 9; ++r) Zb[((size_t)t * 19 + rz) * 9 + r] = a[r];
            }
            if (t + 1 < lenS) stash(buf ^ 1);
            wave_sync_lds();
            VBA_KSTAMP(tid == 0 && c == 30, 35 + t);
            continue;
        }
        double base[9];
        if (buf == 0) {
            load_base(blk[0], 0, t == 0, base);
            forward_step<0, 19, PIVOT, SPARSE_L>(t > 0 ? blk[0] : nullptr, base, a, lane, zero_pivot);
        } else {
            load_base(blk[1], 9, false, base);
            forward_step<9, 19, PIVOT, SPARSE_L>(blk[1], base, a, lane, zero_pivot);
        }
        const int ub = buf == 0 ? 9 : 0;
        if (lane >= ub && lane < ub + 9) {
#pragma unroll
            for (int r = 0; r < 9; ++r) Xb[(size_t)t * 81 + r * 9 + (lane - ub)] = a[r];
        } else if (isV || isW || isY) {
#pragma unroll
            for (int r = 0; r < 9; ++r) Zb[((size_t)t * 19 + zcol) * 9 + r] = a[r];
        }
        if (t + 1 < lenS) stash(buf ^ 1);
        wave_sync_lds();
    }
    if (side == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) blkM[lane + 64 * q] = mid[q];
    }
    __syncthreads();
    VBA_KSTAMP(tid == 0 && c == 30, 40);
    // block m: both neighbours folded in, then the same Gauss-Jordan step on [M | 19 right-hand sides]
    if (kRows && side == 0) {
        const double* XL = reg0 + 512 + (size_t)(lenL - 1) * 81;
        const double* ZL = reg0 + 512 + (size_t)hs * 81 + (size_t)(lenL - 1) * 171;
        const double* XR = reg1 + 512 + (size_t)(lenR - 1) * 81;
        const double* ZR = reg1 + 512 + (size_t)hs * 81 + (size_t)(lenR - 1) * 171;
        // column of the left / right sweep's results this lane folds in (wave 1 keeps the right coupling in ITS columns 1..9)
        const int cl = rz, cr = rz == 0 ? 0 : (rz < 10 ? rz + 9 : rz - 9);
        const double* pl = rD ? XL + rc : ZL + (size_t)(rR ? cl : 0) * 9;
        const double* pr = rD ? XR + rc : ZR + (size_t)(rR ? cr : 0) * 9;
        const int st = rD ? 9 : 1;
        double base[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = 0.0;
            const double dv = blkM[81 + r * 9 + (rD ? rc : 0)], yv = blkM[243 + r];
            if (rD) v = dv + (r == rc ? lam32 : 0.0);
            else if (rR && rz == 0) v = yv;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bool rot_r = (r >= 3 && r < 6), rot_j = (j >= 3 && j < 6);
                if (!SPARSE_L || rot_r == rot_j) {
                    v -= blkM[r * 9 + j] * pl[j * st];
                    v -= blkM[162 + r * 9 + j] * pr[j * st];
                }
            }
            base[r] = (rD || rR) ? v : 0.0;
            a[r] = base[r];
        }
        bool mybad = false;
        cr_pivots_dpp<0>(base, a, rc, mybad);
        zero_pivot = zero_pivot | (rD & mybad);
        if (rR) {
#pragma unroll
            for (int r = 0; r < 9; ++r) xm[(size_t)rz * 9 + r] = a[r];
        }
    } else if (side == 0) {
        const double* XL = reg0 + 512 + (size_t)(lenL - 1) * 81;
        const double* ZL = reg0 + 512 + (size_t)hs * 81 + (size_t)(lenL - 1) * 171;
        const double* XR = reg1 + 512 + (size_t)(lenR - 1) * 81;
        const double* ZR = reg1 + 512 + (size_t)hs * 81 + (size_t)(lenR - 1) * 171;
        const bool isD = lane < 9, isR = isV || isW || isY;
        // column of the left / right sweep's results this lane folds in (wave 1 keeps the right coupling in ITS columns 1..9)
        const int cl = isY ? 0 : (isV ? 1 + (lane - 18) : (isW ? 10 + (lane - 27) : 0));
        const int cr = isY ? 0 : (isV ? 10 + (lane - 18) : (isW ? 1 + (lane - 27) : 0));
        const double* pl = isD ? XL + lane : ZL + (size_t)cl * 9;
        const double* pr = isD ? XR + lane : ZR + (size_t)cr * 9;
        const int st = isD ? 9 : 1;
        double base[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = 0.0;
            if (isD) v = blkM[81 + r * 9 + lane] + (r == lane ? lam32 : 0.0);
            else if (isY) v = blkM[243 + r];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bool rot_r = (r >= 3 && r < 6), rot_j = (j >= 3 && j < 6);
                if (!SPARSE_L || rot_r == rot_j) {
                    v -= blkM[r * 9 + j] * pl[j * st];
                    v -= blkM[162 + r * 9 + j] * pr[j * st];
                }
            }
            base[r] = (isD || isR) ? v : 0.0;
        }
        forward_step<0, 19, PIVOT, SPARSE_L>(nullptr, base, a, lane, zero_pivot);
        if (isR) {
#pragma unroll
            for (int r = 0; r < 9; ++r) xm[(size_t)zcol * 9 + r] = a[r];
        }
    }
    __syncthreads();
    VBA_KSTAMP(tid == 0 && c == 30, 41);
    // outward substitution on the matrix cores (see chunk_eliminate): x_t = Z_t - X_t x_{t+1}
    const int lr = lane & 15, lk = lane >> 4;
    auto colperm = [&](int col) { return side ? (col == 0 ? 0 : (col < 10 ? col + 9 : col - 9)) : col; };
    auto load_cols = [&](const double* Z, int tile, bool perm) {
        vf4 z;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lk + 4 * i, col = 16 * tile + lr;
            const bool ok = row < 9 && col < 19;
            const double v = Z[ok ? (size_t)(perm ? colperm(col) : col) * 9 + row : 0];
            z[i] = ok ? v : 0.0;
        }
        return z;
    };
    auto mul_sub = [&](const double* M, double sign, vf4& acc0, vf4& acc1, const vf4& b0v, const vf4& b1v) {
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int k = 4 * st + lk;
            const bool ok = lr < 9 && k < 9;
            const double mm = M[ok ? lr * 9 + k : 0];
            const double am = ok ? sign * mm : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, b0v[st], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, b1v[st], acc1, 0, 0, 0);
        }
    };
    // the real column a value of this side's column `col` belongs to
    // (every lane stores every time: one without an entry repeats its own first one -- row lk < 4 of the first tile always
    // exists -- instead of opening a branch region per store)
    auto store_cols = [&](double* dst, size_t col_stride, size_t row_stride, const vf4& v0, const vf4& v1) {
        const size_t c0 = (size_t)colperm(lr) * col_stride, c1 = (size_t)colperm(lr < 3 ? 16 + lr : 0) * col_stride;
        const size_t home = c0 + (size_t)lk * row_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lk + 4 * i;
            const bool ok0 = row < 9, ok1 = ok0 && lr < 3;
            dst[ok0 ? c0 + (size_t)row * row_stride : home] = ok0 ? v0[i] : v0[0];
            dst[ok1 ? c1 + (size_t)row * row_stride : home] = ok1 ? v1[i] : v0[0];
        }
    };
    vf4 x0 = load_cols(xm, 0, true), x1 = load_cols(xm, 1, true);
    if (side == 0) store_cols(csol + (size_t)m * 171, 9, 1, x0, x1);
    for (int t = lenS - 1; t >= 0; --t) {
        vf4 n0 = load_cols(Zb + (size_t)t * 171, 0, false), n1 = load_cols(Zb + (size_t)t * 171, 1, false);
        mul_sub(Xb + (size_t)t * 81, -1.0, n0, n1, x0, x1);
        x0 = n0;
        x1 = n1;
        store_cols(csol + (size_t)block_of(t) * 171, 9, 1, x0, x1);
    }
    VBA_KSTAMP(tid == 0 && c == 30, 42);
    // x is now the solution next to this side's separator: its contribution to that row of the reduced system
    if (side == 0 ? c > 0 : has_sep) {
        vf4 p0 = {0.0, 0.0, 0.0, 0.0}, p1 = {0.0, 0.0, 0.0, 0.0};
        mul_sub(side ? Cm : Cm + 81, 1.0, p0, p1, x0, x1);
        store_cols(side ? cL + (size_t)c * 171 : cR + (size_t)(c - 1) * 171, 1, 19, p0, p1);
    }
}

// number of separators of a chain of n blocks cut into chunks of s
__device__ __forceinline__ int n_separators(int n, int s) { return (n + s - 1) / s - 1; }

// level 1: chunks of the window's own chain
template <bool PIVOT>
__global__ __launch_bounds__(64) void k_solve_chunks(DevView V, int s) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int w = blockIdx.y, c = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n = V.n[w];
    if (c * s >= n) return;
    const int lane = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    if (c == 0 && lane == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    bool bad = false;
    const BandSource src{V.bands + sb * 243, V.rhs + sb * 9};
    chunk_eliminate<PIVOT, true>(src, n, s, c, lam32, V.csol + sb * 171, V.cL + rb * 171, V.cR + rb * 171, smem, lane, bad);
    report_pivot<PIVOT>(bad, sc, lane, V.par);
}

template <bool PIVOT>
__global__ __launch_bounds__(128) void k_solve_chunks_ts(DevView V, int s) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int w = blockIdx.y, c = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n = V.n[w];
    if (c * s >= n) return;
    const int tid = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    if (c == 0 && tid == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    bool bad = false;
    const BandSource src{V.bands + sb * 243, V.rhs + sb * 9};
    chunk_eliminate_twosided<PIVOT, true>(src, n, s, c, lam32, V.csol + sb * 171, V.cL + rb * 171, V.cR + rb * 171, smem, tid, bad);
    report_pivot<PIVOT>(bad, sc, tid & 63, V.par);
}

// Latency mode: the chunk's wave(s) build the blocks of the chunk themselves (no assembly launch, no round trip of the
// bands through memory).  The 256 threads of the block stage the per-pose inputs of the chunk and of its two
// neighbours in LDS and form the (at most s + 1) blocks  a - 1 .. b + 1  there; the first wave then eliminates the chunk
// exactly as k_solve_chunks does, reading blocks from LDS.  What the later kernels need from the system itself -- the
// diagonal block and right-hand side of the chunk's right separator (reduced system), the last pose's diagonal block
// (last_hessian) -- is written out on the way.
struct LdsBlockSource {
    const double* blocks;   // [count][252]
    int first;              // pose index of blocks[0]
    __device__ double operator()(int i, int e) const { return blocks[(size_t)(i - first) * 252 + e]; }
};

template <bool PIVOT, bool REG>
__global__ __launch_bounds__(256) void k_solve_chunks_fused(DevView V, int s) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int kAsmIn = kAsmBase + (REG ? kAsmPrior : 0);
    const int w = blockIdx.y, c = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n = V.n[w];
    if (c * s >= n) return;
    const int tid = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    if (c == 0 && tid == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    int a0, b0;
    bool has_sep;
    chunk_range(c, s, n, a0, b0, has_sep);
    const int j0 = a0 > 0 ? a0 - 1 : 0, j1 = has_sep ? b0 + 1 : b0;         // blocks formed here
    const int nblk = j1 - j0 + 1;
    double* elim = smem;                                                     // scratch of chunk_eliminate
    double* blocks = smem + (512 + (size_t)s * 252 + 162);                   // [s + 1][252]
    double* in = blocks + (size_t)(s + 1) * 252;                             // [s + 2][kAsmIn]: poses j0 - 1 .. j1
    asm_stage<REG>(V, w, n, true, j0 - 1, nblk + 1, in, tid, 256);
    __syncthreads();
    const double inv_wmax = 1.0 / bits_f64(sc.wmax_bits[V.par]);
    // thread t forms entry t of every block of the chunk: which band / row / column it is is decoded once (as in k_assemble)
    if (tid < 252) {
        const int e = tid;
        const bool is_rhs = e >= 243;
        const int which = e / 81, a = is_rhs ? e - 243 : (e % 81) / 9, b = e % 9;
        for (int q = 0; q < nblk; ++q) {
            const int i = j0 + q;
            const AsmRow R = asm_row<REG>(in + (size_t)(q + 1) * kAsmIn, in + (size_t)q * kAsmIn, i, n, true, V.prm.sigma, inv_wmax);
            double v;
            if (is_rhs) {
                v = rhs_entry(R, a);
                if (has_sep && i == j1) V.rhs[(sb + i) * 9 + a] = v;
            } else {
                v = band_entry(R, which, a, b);
                if (which == 1) {
                    if (has_sep && i == j1) V.bands[(sb + i) * 243 + e] = v;
                    if (i == n - 1) V.lastD[(size_t)w * 81 + (e - 81)] = v;
                }
            }
            blocks[(size_t)q * 252 + e] = v;
        }
    }
    __syncthreads();
    if (tid >= 64) return;      // the elimination is one wave's work (its barriers count the surviving wave only)
    bool bad = false;
    const LdsBlockSource src{blocks, j0};
    chunk_eliminate<PIVOT, true>(src, n, s, c, lam32, V.csol + sb * 171, V.cL + rb * 171, V.cR + rb * 171, elim, tid, bad);
    report_pivot<PIVOT>(bad, sc, tid, V.par);
}

// The same with the two-sided elimination and the uniform-pass row former (vba_asm_fast.h): the four waves of the block form
// the chunk's (at most s + 1) blocks in LDS -- one wave per pose row, seven uniform passes -- then waves 0 and 1 eliminate
// from both ends.  No assembly launch in the full phase of the latency mode.
__host__ __device__ constexpr int twosided_fused_lds_doubles(int s, bool reg) {
    return twosided_lds_doubles(s) + (s + 1) * 252 + (s + 2) * (kAsmBase + (reg ? kAsmPrior : 0));
}

template <bool PIVOT, bool REG>
__device__ __forceinline__ void chunks_ts_fused_body(const DevView& V, int s, int w, int c, double* smem) {
    constexpr int kAsmIn = kAsmBase + (REG ? kAsmPrior : 0);
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n = V.n[w];
    if (c * s >= n) return;
    const int tid = threadIdx.x;
    VBA_KSTAMP(tid == 0 && c == 30, 32);
    const size_t sb = (size_t)w * V.n_max;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    // (requested here, in front of the staging: behind the barrier below this load would be a round trip to memory of its own)
    const unsigned long long wmax_bits = sc.wmax_bits[V.par];
    if (c == 0 && tid == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    int a0, b0;
    bool has_sep;
    chunk_range(c, s, n, a0, b0, has_sep);
    const int j0 = a0 > 0 ? a0 - 1 : 0, j1 = has_sep ? b0 + 1 : b0;         // blocks formed here
    const int nblk = j1 - j0 + 1;
    double* elim = smem;                                                     // scratch of the elimination
    double* blocks = smem + twosided_lds_doubles(s);                         // [s + 1][252]
    double* in = blocks + (size_t)(s + 1) * 252;                             // [s + 2][kAsmIn]: poses j0 - 1 .. j1
    // staging: four loads per thread in flight at a time (the address is selected, never the load: vba_asm.h)
    {
        const int total = (nblk + 1) * kAsmIn;
        for (int e0 = 0; e0 < total; e0 += 4 * 256) {
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + tid + 256 * k;
                const int slot = e / kAsmIn, q = e - slot * kAsmIn;
                const int i = j0 - 1 + slot;
                const bool ok = e < total && i >= 0 && i < n;
                v[k] = asm_input_nobranch<REG>(V, sb + (ok ? i : 0), ok ? q : 0);
                v[k] = ok ? v[k] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + tid + 256 * k;
                if (e < total) in[e] = v[k];
            }
        }
    }
    __syncthreads();
    VBA_KSTAMP(tid == 0 && c == 30, 33);
    const double inv_wmax = 1.0 / bits_f64(wmax_bits);
    {
        // formation by column: a row of 16 lanes per pose row, sixteen pose rows per pass of the workgroup
        const int lane = tid & 63, wave = tid >> 6, row = lane >> 4, cc = lane & 15;
        const AsmColLane cl = asm_col_lane(cc);
        for (int q0 = 0; q0 < nblk; q0 += 16) {
            const int q = q0 + wave * 4 + row;
            const bool have = q < nblk;
            const int qq = have ? q : 0;
            const int i = j0 + qq;
            double* blk = blocks + (size_t)qq * 252;
            const bool sep = have && has_sep && i == j1, last = have && i == n - 1;
            double A[9], B[9], Lc[9];
            VBA_KSTAMP(tid == 0 && c == 30, 48);
            asm_form_columns<REG>(cl, cc, in + (size_t)(qq + 1) * kAsmIn, in + (size_t)qq * kAsmIn, true, i < n - 1, i > 0, V.prm.sigma, inv_wmax, A, B, Lc);
#ifdef VBA_RESIDENT_STAMPS
            if (tid == 0 && c == 30) g_kstamps[49] = (unsigned long long)(A[0] + A[8] + B[4] + Lc[7] != 12345.0);
            VBA_KSTAMP(tid == 0 && c == 30, 50);
#endif
            {
                // one destination and one predicate per lane and array, decided once (as nested branches inside the unrolled
                // loop this was some forty basic blocks).  What later kernels read from memory: the right separator's
                // diagonal block and right-hand side (reduced system), the last pose's diagonal block (last_hessian)
                const bool isc = have && cc < 9, isr = have && cc == 9;
                double* pA = blk + (cc < 9 ? 81 + cc : 243);
                const int stA = cc < 9 ? 9 : 1;
                double* gS = cc < 9 ? V.bands + (sb + i) * 243 + 81 + cc : V.rhs + (sb + i) * 9;
                double* gL = V.lastD + (size_t)w * 81 + (cc < 9 ? cc : 0);
                const bool wS = sep && (isc || isr), wL = last && isc;
                // LDS: every lane stores, the lanes without a column into a dump word of the elimination's scratch (idle until
                // the barrier below) -- a predicated store inside the unrolled loop is a branch region of its own, and
                // forty-five of them cost more than the formation itself (1.3 against 0.6 us)
                double* dump = elim + (tid & 63);
                double* pL = isc ? blk + cc : dump;
                double* pD = (isc || isr) ? pA : dump;
                double* pU = isc ? blk + 162 + cc : dump;
                const int sC = isc ? 9 : 0, sD = (isc || isr) ? stA : 0;
#pragma unroll
                for (int a9 = 0; a9 < 9; ++a9) {
                    pL[a9 * sC] = Lc[a9];
                    pD[a9 * sD] = A[a9];
                    pU[a9 * sC] = B[a9];
                }
                if (wS) {
#pragma unroll
                    for (int a9 = 0; a9 < 9; ++a9) gS[a9 * stA] = A[a9];
                }
                if (wL) {
#pragma unroll
                    for (int a9 = 0; a9 < 9; ++a9) gL[a9 * 9] = A[a9];
                }
            }
            VBA_KSTAMP(tid == 0 && c == 30, 51);
        }
    }
    __syncthreads();
    VBA_KSTAMP(tid == 0 && c == 30, 34);
    if (tid >= 128) return;     // the elimination is two waves' work (its barriers count the surviving waves only)
    bool bad = false;
    const LdsBlockSource src{blocks, j0};
    chunk_eliminate_twosided<PIVOT, true>(src, n, s, c, lam32, V.csol + sb * 171, V.cL + rb * 171, V.cR + rb * 171, elim, tid, bad);
    VBA_KSTAMP(tid == 0 && c == 30, 47);
    report_pivot<PIVOT>(bad, sc, tid & 63, V.par);
}

template <bool PIVOT, bool REG>
__global__ __launch_bounds__(256) void k_solve_chunks_ts_fused(DevView V, int s) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    chunks_ts_fused_body<PIVOT, REG>(V, s, blockIdx.y, blockIdx.x, smem);
}

// level 2: the reduced system over the level-1 separators is itself cut into chunks of s2
template <bool PIVOT>
__global__ __launch_bounds__(64) void k_solve_chunks2(DevView V, int s, int s2) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int w = blockIdx.y, c = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n1 = n_separators(V.n[w], s);
    if (n1 <= 0 || c * s2 >= n1) return;
    const int lane = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    bool bad = false;
    const ReducedSource<BandSource> src{BandSource{V.bands + sb * 243, V.rhs + sb * 9}, V.cL + rb * 171, V.cR + rb * 171, s};
    chunk_eliminate<PIVOT, false>(src, n1, s2, c, lam32, V.csol2 + rb * 171, V.cL2 + rb * 171, V.cR2 + rb * 171, smem, lane, bad);
    report_pivot<PIVOT>(bad, sc, lane, V.par);
}

// Solves the last reduced block-tridiagonal system (one wave per window): over the level-1 separators (s2 == 0)
// or over the level-2 separators.
template <bool PIVOT>
__global__ __launch_bounds__(64) void k_solve_reduced(DevView V, int s, int s2) {
    __shared__ double blk[2][256];
    const int w = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n1 = n_separators(V.n[w], s);
    if (n1 <= 0) return;
    const int lane = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    const ReducedSource<BandSource> src1{BandSource{V.bands + sb * 243, V.rhs + sb * 9}, V.cL + rb * 171, V.cR + rb * 171, s};
    bool zero_pivot = false;
    if (s2 == 0) {
        chain_solve<PIVOT, false>(src1, n1, lam32, V.rXs + rb * 81, V.rzs + rb * 9, V.rx + rb * 9, blk, lane, zero_pivot);
    } else {
        const int n2 = n_separators(n1, s2);
        if (n2 > 0) {
            const ReducedSource<ReducedSource<BandSource>> src2{src1, V.cL2 + rb * 171, V.cR2 + rb * 171, s2};
            chain_solve<PIVOT, false>(src2, n2, lam32, V.rXs + rb * 81, V.rzs + rb * 9, V.rx2 + rb * 9, blk, lane, zero_pivot);
        }
    }
    report_pivot<PIVOT>(zero_pivot, sc, lane, V.par);
}

// The reduced system over the separators by block cyclic reduction inside ONE workgroup (16 waves, the whole system
// in LDS): log2(n1) levels of "every other block eliminated in parallel" instead of n1 sequential block steps.
//   level with stride h, active blocks k = r h - 1 (r = 1, 2, ...):
//     A  odd r:   [PL | PU | Pg]_k = D_k^{-1} [L_k | U_k | g_k]                   (one Gauss-Jordan per wave, in place)
//     B  even r:  D_j -= L_j PU_{j-h} + U_j PL_{j+h},  g_j -= L_j Pg_{j-h} + U_j Pg_{j+h},
//                 L_j  = -L_j PL_{j-h},  U_j = -U_j PU_{j+h}                        (couples j to j -+ 2h from now on)
//   back substitution, coarsest level first:  x_k = Pg_k - PL_k x_{k-h} - PU_k x_{k+h}.
// Schur complements of the (damped, near-SPD) system stay near-SPD, so the unpivoted path applies with the same
// per-pivot check; PIVOT exchanges rows inside a block as everywhere else.
// LDS: n1 blocks of 252 doubles [L | D | U | g]; x overwrites g.  n1 <= kCrMax.
constexpr int kCrMax = 64;
constexpr int kFusedChunkMax = 28;  // largest chunk whose blocks, staged inputs and elimination scratch fit 160 KiB of LDS
constexpr int kCrThreads = 1024;
constexpr int kCrSplitMin = 24;     // from this many separators on, the first level runs as its own multi-CU kernel (re-measured with
                                    // the two-wave chunks: 56.2 us per call against 59.1 with all levels in the one workgroup)

// Per-lane geometry of the two block operations (depends on the lane only, built once per kernel).
struct CrLanes {
    // elimination (A): lane -> column of [D | L | U | g]; one address per lane and role (selecting among loaded
    // VALUES would make every lane load every alternative)
    int grp, own, ownst;
    // fold (B) on the matrix cores:
    //   out (9 x 28: new D | L | U | g) = init - [L_j | U_j] (9 x 18) * Bm (18 x 28),
    //   Bm rows 0..8  = [PU | PL | 0  | Pg] of the left neighbour,  rows 9..17 = [PL | 0 | PU | Pg] of the right one.
    // v_mfma_f64_16x16x4: lane l feeds A[l & 15][4 s + (l >> 4)] and B[4 s + (l >> 4)][l & 15] of k-step s and owns
    // C[(l >> 4) + 4 i][l & 15], i = 0..3; two column tiles, five k-steps.  One LDS read per operand element instead
    // of 162 broadcast reads per lane (the VALU form was LDS-bandwidth-bound).
    int lr, lk;
    int offA[5], offB0[5], offB1[5];    // -1: structural zero
    bool hiB[5];                        // the B element comes from the right neighbour
};

__device__ __forceinline__ CrLanes cr_lanes(int lane) {
    CrLanes g;
    g.grp = lane < 9 ? 0 : (lane < 18 ? 1 : (lane < 27 ? 2 : (lane == 27 ? 3 : 4)));
    const int c = g.grp < 3 ? lane - 9 * g.grp : 0;
    g.own = g.grp == 0 ? 81 + c : (g.grp == 1 ? c : (g.grp == 2 ? 162 + c : 243));
    g.ownst = g.grp == 3 ? 1 : 9;
    g.lr = lane & 15;
    g.lk = lane >> 4;
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int k = 4 * st + g.lk;
        g.offA[st] = (g.lr < 9 && k < 18) ? (k < 9 ? g.lr * 9 + k : 162 + g.lr * 9 + (k - 9)) : -1;
        const bool lo = k < 9;
        const int q = lo ? k : k - 9;
        g.hiB[st] = !lo;
        auto bm = [&](int col) -> int {
            if (k >= 18) return -1;
            if (col < 9) return lo ? 162 + q * 9 + col : q * 9 + col;
            if (col < 18) return lo ? q * 9 + col - 9 : -1;
            if (col < 27) return lo ? -1 : 162 + q * 9 + col - 18;
            if (col == 27) return 243 + q;
            return -1;
        };
        g.offB0[st] = bm(g.lr);
        g.offB1[st] = bm(16 + g.lr);
    }
    return g;
}

// (cr_pivots_dpp, above forward_step's users: D replicated per row of 16 lanes, seven other columns per row)
__device__ __forceinline__ void cr_eliminate_dpp(double* B, int lane, bool& bad) {
    const int row = lane >> 4, c = lane & 15;
    const int o = row * 7 + (c - 9);                    // column of [L | U | g] of a lane with c >= 9
    const bool isD = c < 9, isX = !isD && o < 19;
    const int off = isD ? 81 + c : (o < 9 ? o : (o < 18 ? 162 + (o - 9) : 243));
    const int st = (isX && o == 18) ? 1 : 9;
    const double* p = B + ((isD || isX) ? off : 0);
    double a[9], base[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const double v = p[r * st];
        a[r] = (isD || isX) ? v : 0.0;
        base[r] = a[r];
    }
    bool mybad = false;
    cr_pivots_dpp<0>(base, a, c, mybad);
    bad = bad | (isD & mybad);
    if (isX) {
        double* q = B + off;
#pragma unroll
        for (int r = 0; r < 9; ++r) q[r * st] = a[r];
    }
}

// A: [PL | PU | Pg] = D^{-1} [L | U | g] of block B (LDS, 252 doubles), in place; one wave.
template <bool PIVOT>
__device__ __forceinline__ void cr_eliminate(double* B, const CrLanes& g, int lane, bool& bad) {
#ifndef VBA_CR_READLANE
    if constexpr (!PIVOT) {
        cr_eliminate_dpp(B, lane, bad);
        return;
    }
#endif
    double base[9], a[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const double v = B[g.own + r * g.ownst];
        base[r] = g.grp < 4 ? v : 0.0;
        a[r] = 0.0;
    }
    forward_step<0, 10, PIVOT, false>(nullptr, base, a, lane, bad);
    if (g.grp >= 1 && g.grp <= 3) {
#pragma unroll
        for (int r = 0; r < 9; ++r) B[g.own + r * g.ownst] = a[r];
    }
}

// B: fold the eliminated neighbours Pm (left) and Pp (right, if has_p) into block Bj, in place; one wave.
__device__ __forceinline__ void cr_fold(double* Bj, const double* Pm, const double* Pp, bool has_p, const CrLanes& g) {
    vf4 acc0, acc1;
    VBA_KSTAMP(threadIdx.x == 0 && gridDim.y == 1 && blockDim.x == 1024, 79);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = g.lk + 4 * i;
        const bool rv = row < 9;
        const double d0 = Bj[(rv && g.lr < 9) ? 81 + row * 9 + g.lr : 0];
        const double g0 = Bj[(rv && g.lr == 11) ? 243 + row : 0];
        acc0[i] = (rv && g.lr < 9) ? d0 : 0.0;
        acc1[i] = (rv && g.lr == 11) ? g0 : 0.0;      // column 27 = 16 + 11
    }
    // every LDS operand first (15 reads in flight together), then the chain of matrix operations: left to itself the
    // compiler reads each k-step's operands right in front of its two MFMAs -- five LDS round trips one after the other
    double am[5], b0[5], b1[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const double a0 = Bj[g.offA[st] >= 0 ? g.offA[st] : 0];
        am[st] = g.offA[st] >= 0 ? -a0 : 0.0;
        const double* P = g.hiB[st] ? Pp : Pm;
        const bool okp = !g.hiB[st] || has_p;
        const double v0 = P[g.offB0[st] >= 0 ? g.offB0[st] : 0], v1 = P[g.offB1[st] >= 0 ? g.offB1[st] : 0];
        b0[st] = (okp && g.offB0[st] >= 0) ? v0 : 0.0;
        b1[st] = (okp && g.offB1[st] >= 0) ? v1 : 0.0;
    }
#ifndef VBA_FOLD_INTERLEAVED
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef VBA_RESIDENT_STAMPS
    const bool fson = threadIdx.x == 0 && gridDim.y == 1 && blockDim.x == 1024;
    VBA_KSTAMP(fson, 80);
    if (fson) g_kstamps[81] = (unsigned long long)(am[0] + b0[0] + b1[4] + am[4] != 12345.0);   // (forces the operands)
    VBA_KSTAMP(fson, 82);
#endif
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[st], b0[st], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[st], b1[st], acc1, 0, 0, 0);
    }
#ifdef VBA_RESIDENT_STAMPS
    if (fson) g_kstamps[83] = (unsigned long long)(acc0[0] + acc1[3] != 12345.0);               // (forces the results)
    VBA_KSTAMP(fson, 84);
#endif
    // every operand has been read (the LDS operations of a wave execute in order): replace the block.  One destination per
    // lane and tile, decided by arithmetic -- as nested branches this tail was twenty basic blocks
    const int c1 = 16 + g.lr;
    const int col0 = g.lr < 9 ? 81 + g.lr : g.lr - 9;                                              // D | L columns 0..6
    const int col1 = c1 < 18 ? c1 - 9 : (c1 < 27 ? 162 + (c1 - 18) : 243);                         // L columns 7, 8 | U | g
    const int st1 = c1 == 27 ? 1 : 9;
    const bool has1 = c1 <= 27;
    // ... and every lane stores every time: a lane without an entry repeats its own first one (row g.lk < 4 of tile 0 always
    // exists) -- a predicated store is a branch region of its own
    const int home = col0 + g.lk * 9;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = g.lk + 4 * i;
        const bool ok0 = row < 9, ok1 = ok0 && has1;
        Bj[ok0 ? col0 + row * 9 : home] = ok0 ? acc0[i] : acc0[0];
        Bj[ok1 ? col1 + row * st1 : home] = ok1 ? acc1[i] : acc0[0];
    }
#ifdef VBA_RESIDENT_STAMPS
    VBA_KSTAMP(fson, 85);
#endif
}

// Blocks q0 + u * stride (u < NB) of the reduced system (see ReducedSource) into LDS at dst + u * dst_stride * 252.  A wave takes
// whole blocks and a lane the same (row, column) of L, D and U, so the index arithmetic is done once per three
// entries and nothing diverges (walking the 252 entries of a block through the generic source costs more in integer
// divisions and branches than in loads); all loads are issued before the first store.
// between(): called when the loads have been issued and before the first store waits for them (lane geometry and the like)
template <int NB, class Between>
__device__ __forceinline__ void cr_fill(const DevView& V, int w, int s, int n1, double lam32, int q0, int stride, double* dst, int dst_stride, int lane,
                                        Between&& between) {
    const size_t sb = (size_t)w * V.n_max, rb = (size_t)w * V.p_max;
    const double* bands = V.bands + sb * 243;
    const double* rhs = V.rhs + sb * 9;
    const double* cL = V.cL + rb * 171;
    const double* cR = V.cR + rb * 171;
    const int r0 = lane / 9, c0 = lane % 9, r1 = (lane + 64) / 9, c1 = (lane + 64) % 9;
    double lv[NB][2], lw[NB][2], rv[NB][2], rw[NB][2], dv[NB][2], gv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int q = q0 + u * stride;
        const bool in = q >= 0 && q < n1;
        const size_t j = in ? (size_t)(q + 1) * s - 1 : 0;
        const double* l = cL + (size_t)(in ? q : 0) * 171;
        const double* r_ = cR + (size_t)(in ? q : 0) * 171;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = it ? r1 : r0, cc = it ? c1 : c0;
            const bool ok = in && lane + 64 * it < 81;
            lv[u][it] = ok ? l[r * 19 + 1 + cc] : 0.0;
            lw[u][it] = ok ? l[r * 19 + 10 + cc] : 0.0;
            rv[u][it] = ok ? r_[r * 19 + 1 + cc] : 0.0;
            rw[u][it] = ok ? r_[r * 19 + 10 + cc] : 0.0;
            dv[u][it] = ok ? bands[j * 243 + 81 + lane + 64 * it] : 0.0;
        }
        gv[u] = (in && lane < 9) ? rhs[j * 9 + lane] - l[lane * 19] - r_[lane * 19] : 0.0;
    }
    between();
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int q = q0 + u * stride;
        if (q >= 0 && q < n1) {
            double* B = dst + (size_t)u * dst_stride * 252;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = lane + 64 * it;
                if (i < 81) {
                    const int r = it ? r1 : r0, cc = it ? c1 : c0;
                    B[i] = q == 0 ? 0.0 : -lv[u][it];                       // no neighbour on that side
                    B[81 + i] = dv[u][it] - lw[u][it] - rv[u][it] + (r == cc ? lam32 : 0.0);
                    B[162 + i] = q == n1 - 1 ? 0.0 : -rw[u][it];
                }
            }
            if (lane < 9) B[243 + lane] = gv[u];
        }
    }
}

#ifdef VBA_VARIANTS   // one cyclic-reduction level in front instead of two (VBA_OPT_FUSION bit 4): 0.9 us per call slower, comparison builds
// First level of the cyclic reduction as its own kernel, one wave (one CU) per pair of separators: wave t builds the
// blocks 2t, 2t+1, 2t+2, eliminates the two even ones (each even block is eliminated by both of its odd neighbours'
// waves: redundant work instead of communication), folds them into block 2t+1 and leaves
//   red[t] = the folded block 2t+1 (252 doubles) and  P[2t] = [PL | PU | Pg] of block 2t (for the back substitution)
// in global memory.  The 31 eliminations + folds of a 62-separator system then run on 31 CUs instead of sharing the
// four SIMDs of one.
// Two waves: the two eliminations are independent, so wave 1 builds and eliminates block 2t+2 beside wave 0's 2t.
template <bool PIVOT>
__global__ __launch_bounds__(128) void k_cr_level0(DevView V, int s) {
    __shared__ __attribute__((aligned(16))) double blk[3 * 252];
    const int w = blockIdx.y, t = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n1 = n_separators(V.n[w], s);
    if (n1 < kCrSplitMin || n1 > 2 * kCrMax || 2 * t >= n1) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    CrLanes g;
    if (wv == 0) cr_fill<2>(V, w, s, n1, lam32, 2 * t, 1, blk, 1, lane, [&]() { g = cr_lanes(lane); });
    else cr_fill<1>(V, w, s, n1, lam32, 2 * t + 2, 1, blk + 504, 1, lane, [&]() { g = cr_lanes(lane); });
    __syncthreads();
    bool bad = false;
    const bool has_j = 2 * t + 1 < n1, has_p = 2 * t + 2 < n1;
    if (wv == 0) cr_eliminate<PIVOT>(blk, g, lane, bad);
    else if (has_p) cr_eliminate<PIVOT>(blk + 504, g, lane, bad);
    __syncthreads();
    if (wv == 0) {
        double* P = V.csol2 + (rb + 2 * t) * 171;           // scratch of the two-level driver, unused in this mode
        for (int e = lane; e < 171; e += 64) P[e] = e < 81 ? blk[e] : blk[81 + e];       // PL | PU | Pg
        if (has_j) {
            cr_fold(blk + 252, blk, blk + 504, has_p, g);
            double* R = V.cL2 + rb * 171 + (size_t)t * 252;
            for (int e = lane; e < 252; e += 64) R[e] = blk[252 + e];
        }
    }
    report_pivot<PIVOT>(bad, sc, lane, V.par);
}

#endif  // VBA_VARIANTS

// The first TWO levels on their own CUs: four waves per group of four separators.  Group t builds the seven blocks
// 4t .. 4t+6, eliminates the even ones (four waves side by side), folds them into 4t+1, 4t+3, 4t+5, eliminates 4t+1 and
// 4t+5 (level 1: every other odd block) and folds those into 4t+3.  It leaves
//   red2[t] = the twice-folded block 4t+3 (252 doubles),  P[4t], P[4t+2] (level 0) and P[4t+1] (level 1) = [PL | PU | Pg]
// in global memory; what it shares with its neighbour groups (blocks 4t+4 .. 4t+6) is computed by both: redundant work
// instead of communication.  In the one-workgroup kernel the first level ran 16 eliminations on the four SIMDs of one CU
// (5.8 us, issue-bound at four waves per SIMD) -- here they are spread over 16 CUs and that kernel starts from 15 blocks.
template <bool PIVOT>
__device__ __forceinline__ void cr_level01_body(const DevView& V, int s, int w, int t, double* blk) {
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n1 = n_separators(V.n[w], s);
    if (n1 < kCrSplitMin || n1 > 4 * kCrMax || 4 * t >= n1) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    const int q0 = 4 * t;
    CrLanes g;
    cr_fill<2>(V, w, s, n1, lam32, q0 + wv, 4, blk + (size_t)wv * 252, 4, lane, [&]() { g = cr_lanes(lane); });     // blocks wv, wv + 4
    __syncthreads();
    bool bad = false;
    auto store_P = [&](int u) {         // [PL | PU | Pg] of block q0 + u
        double* P = V.csol2 + (rb + q0 + u) * 171;
        const double* B = blk + (size_t)u * 252;
        for (int e = lane; e < 171; e += 64) P[e] = e < 81 ? B[e] : B[81 + e];
    };
    // level 0: the even blocks
    if (q0 + 2 * wv < n1) cr_eliminate<PIVOT>(blk + (size_t)(2 * wv) * 252, g, lane, bad);
    __syncthreads();
    if (wv < 3) {
        const int u = 2 * wv + 1;
        if (q0 + u < n1) cr_fold(blk + (size_t)u * 252, blk + (size_t)(u - 1) * 252, blk + (size_t)(u + 1) * 252, q0 + u + 1 < n1, g);
    } else {
        store_P(0);
        if (q0 + 2 < n1) store_P(2);
    }
    __syncthreads();
    // level 1: blocks 4t+1 and 4t+5
    if (wv < 2) {
        const int u = 4 * wv + 1;
        if (q0 + u < n1) cr_eliminate<PIVOT>(blk + (size_t)u * 252, g, lane, bad);
    }
    __syncthreads();
    if (wv == 0) {
        if (q0 + 3 < n1) {
            cr_fold(blk + 3 * 252, blk + 1 * 252, blk + 5 * 252, q0 + 5 < n1, g);
            wave_sync_lds();
            double* R = V.cL2 + rb * 171 + (size_t)t * 252;
            for (int e = lane; e < 252; e += 64) R[e] = blk[3 * 252 + e];
        }
    } else if (wv == 1) {
        if (q0 + 1 < n1) store_P(1);
    }
    report_pivot<PIVOT>(bad, sc, lane, V.par);
}

template <bool PIVOT>
__global__ __launch_bounds__(256) void k_cr_level01(DevView V, int s) {
    __shared__ __attribute__((aligned(16))) double blk[7 * 252];
    cr_level01_body<PIVOT>(V, s, blockIdx.y, blockIdx.x, blk);
}

#ifdef VBA_VARIANTS   // three levels in front (VBA_CR_LEVELS=3): measured 0.45 us per call SLOWER than two, comparison builds
// The first THREE levels on their own CUs (round 4): eight waves per group of eight separators.  Group t builds the fifteen
// blocks 8t .. 8t+14, eliminates the even ones (eight waves side by side), folds them into the odd ones, eliminates 8t+1, 8t+5,
// 8t+9, 8t+13 and folds those into 8t+3, 8t+7, 8t+11, eliminates 8t+3 and 8t+11 and folds them into 8t+7.  It leaves
//   red3[t] = the three times folded block 8t+7 (252 doubles),  P[8t], P[8t+2], P[8t+4], P[8t+6] (level 0), P[8t+1], P[8t+5]
//   (level 1) and P[8t+3] (level 2) = [PL | PU | Pg]
// in global memory; what it shares with the next group (blocks 8t+8 .. 8t+14) is computed by both.  The one-workgroup kernel then
// starts from n / 8 blocks (7 instead of 15 at 62 separators): its fill shrinks and its first level of eight eliminations, two
// per SIMD, is gone.  Same eliminations and folds in another place: the bits of two levels in front.
template <bool PIVOT>
__device__ __forceinline__ void cr_level012_body(const DevView& V, int s, int w, int t, double* blk) {
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n1 = n_separators(V.n[w], s);
    if (n1 < kCrSplitMin || n1 > 8 * kCrMax || 8 * t >= n1) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;     // 8 waves
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    const int q0 = 8 * t;
    CrLanes g;
    cr_fill<2>(V, w, s, n1, lam32, q0 + wv, 8, blk + (size_t)wv * 252, 8, lane, [&]() { g = cr_lanes(lane); });     // blocks wv, wv + 8
    __syncthreads();
    bool bad = false;
    auto store_P = [&](int u) {         // [PL | PU | Pg] of block q0 + u
        double* P = V.csol2 + (rb + q0 + u) * 171;
        const double* B = blk + (size_t)u * 252;
        for (int e = lane; e < 171; e += 64) P[e] = e < 81 ? B[e] : B[81 + e];
    };
    auto fold = [&](int u, int h) {     // the eliminated blocks u - h and u + h into block u
        if (q0 + u < n1) cr_fold(blk + (size_t)u * 252, blk + (size_t)(u - h) * 252, blk + (size_t)(u + h) * 252, q0 + u + h < n1, g);
    };
    // level 0: the even blocks 0, 2, ..., 14
    if (q0 + 2 * wv < n1) cr_eliminate<PIVOT>(blk + (size_t)(2 * wv) * 252, g, lane, bad);
    __syncthreads();
    if (wv < 7) fold(2 * wv + 1, 1);
    else {
        for (int u = 0; u < 8; u += 2) if (q0 + u < n1) store_P(u);
    }
    __syncthreads();
    // level 1: blocks 1, 5, 9, 13
    if (wv < 4 && q0 + 4 * wv + 1 < n1) cr_eliminate<PIVOT>(blk + (size_t)(4 * wv + 1) * 252, g, lane, bad);
    __syncthreads();
    if (wv < 3) fold(4 * wv + 3, 2);
    else if (wv == 3) { if (q0 + 1 < n1) store_P(1); }
    else if (wv == 4) { if (q0 + 5 < n1) store_P(5); }
    __syncthreads();
    // level 2: blocks 3 and 11
    if (wv < 2 && q0 + 8 * wv + 3 < n1) cr_eliminate<PIVOT>(blk + (size_t)(8 * wv + 3) * 252, g, lane, bad);
    __syncthreads();
    if (wv == 0) {
        if (q0 + 7 < n1) {
            fold(7, 4);
            wave_sync_lds();
            double* R = V.cL2 + rb * 171 + (size_t)t * 252;
            for (int e = lane; e < 252; e += 64) R[e] = blk[7 * 252 + e];
        }
    } else if (wv == 1) {
        if (q0 + 3 < n1) store_P(3);
    }
    report_pivot<PIVOT>(bad, sc, lane, V.par);
}

template <bool PIVOT>
__global__ __launch_bounds__(512) void k_cr_level012(DevView V, int s) {
    __shared__ __attribute__((aligned(16))) double blk[16 * 252];
    cr_level012_body<PIVOT>(V, s, blockIdx.y, blockIdx.x, blk);
}
#endif  // VBA_VARIANTS

// PRE: the first level has been done by k_cr_level0; this kernel continues with the n1 / 2 folded blocks and finishes
// with the back substitution of the level-0 blocks.
// PRE 2: the first two levels have been done by k_cr_level01; the system solved here is over the separators 4b + 3.
template <bool PIVOT, int PRE, int kCrThreads>
__device__ __forceinline__ void reduced_cr_body(const DevView& V, int s, int w, double* smem) {
    VBA_SKIP_CALL(V, w);
    int tsi = 0;
    const bool tson = threadIdx.x == 0;
    VBA_KSTAMP(tson, tsi++);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n0 = n_separators(V.n[w], s);             // separators of the window
    if (n0 <= 0) return;
    if (PRE ? (n0 < kCrSplitMin || n0 > (PRE == 3 ? 8 : (PRE == 2 ? 4 : 2)) * kCrMax) : (n0 >= kCrSplitMin || n0 > kCrMax)) return;   // the other variant's window
    const int n1 = PRE == 3 ? n0 / 8 : (PRE == 2 ? n0 / 4 : (PRE ? n0 / 2 : n0));             // blocks of the system solved here
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kCrThreads / 64;
    const size_t rb = (size_t)w * V.p_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    CrLanes g;
    if (PRE) {
        // all loads of a thread before its first store: a copy loop waits for every element in turn (up to 16 dependent
        // round trips here -- a third of this kernel's time when it was written that way)
        const double* R = V.cL2 + rb * 171;
        constexpr int kFill = (kCrMax * 252 + kCrThreads - 1) / kCrThreads;
        double v[kFill];
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
            const int idx = tid + k * kCrThreads;
            v[k] = idx < n1 * 252 ? R[idx] : 0.0;
        }
        g = cr_lanes(lane);     // (while the loads are in flight)
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
            const int idx = tid + k * kCrThreads;
            if (idx < n1 * 252) smem[idx] = v[k];
        }
    } else {
        cr_fill<kCrMax / NW>(V, w, s, n1, lam32, wave, NW, smem + (size_t)wave * 252, NW, lane, [&]() { g = cr_lanes(lane); });   // blocks wave, wave + NW, ...
    }
    __syncthreads();
    VBA_KSTAMP(tson, tsi++);
    bool bad = false;
    int h = 1, lv = 0;                          // h = 1 << lv (shifts: a division by a run-time h is ~40 instructions per level)
    for (;; h <<= 1, ++lv) {
        const int cnt = n1 >> lv;               // active blocks of this level
        const int nel = (cnt + 1) / 2;
#pragma nounroll
        for (int t = wave; t < nel; t += NW)    // A: eliminate the odd-ranked blocks
            cr_eliminate<PIVOT>(smem + (size_t)((2 * t + 1) * h - 1) * 252, g, lane, bad);
        __syncthreads();
        VBA_KSTAMP(tson, tsi++);
        if (cnt <= 1) break;
        const int nk = cnt / 2;
#pragma nounroll
        for (int t = wave; t < nk; t += NW) {   // B: fold the eliminated neighbours into the even-ranked blocks
            const int j = (2 * t + 2) * h - 1;
            const bool has_p = j + h < n1;
            cr_fold(smem + (size_t)j * 252, smem + (size_t)(j - h) * 252, smem + (size_t)(has_p ? j + h : j) * 252, has_p, g);
        }
        VBA_KSTAMP(tson, 64 + tsi);
        __syncthreads();
        VBA_KSTAMP(tson, tsi++);
    }
    // back substitution: the level that ended the loop has a single block with no active neighbour (x = Pg)
    for (; h >= 1; h >>= 1, --lv) {
        const int cnt = n1 >> lv;
        const int nel = (cnt + 1) / 2;
        for (int idx = tid; idx < nel * 9; idx += kCrThreads) {
            const int t = idx / 9, r = idx % 9;
            const int k = (2 * t + 1) * h - 1;
            double* B = smem + (size_t)k * 252;
            double x = B[243 + r];
            if (k - h >= 0) {
                const double* xm = smem + (size_t)(k - h) * 252 + 243;
#pragma unroll
                for (int q = 0; q < 9; ++q) x -= B[r * 9 + q] * xm[q];
            }
            if (k + h < n1) {
                const double* xp = smem + (size_t)(k + h) * 252 + 243;
#pragma unroll
                for (int q = 0; q < 9; ++q) x -= B[162 + r * 9 + q] * xp[q];
            }
            B[243 + r] = x;     // read only by this thread at this level (the neighbours belong to coarser levels)
        }
        __syncthreads();
        VBA_KSTAMP(tson, tsi++);
    }
    if (PRE == 3) {
        // separators 8b+7 are the blocks solved here.  8b+3 come from the level-2 eliminations, x = Pg - PL x_{q-4} - PU x_{q+4};
        // then 4b+1 from level 1 (neighbours q -+ 2), then the even ones from level 0 (neighbours q -+ 1)
        double* x2 = smem + (size_t)n1 * 252;                   // [ceil(n0 / 8)][9]
        double* x1 = x2 + (size_t)((n0 + 7) / 8) * 9;           // [ceil(n0 / 4)][9]
        auto x_odd = [&](int q) -> const double* {              // solution of an odd separator
            return (q & 7) == 7 ? smem + (size_t)(q >> 3) * 252 + 243 : ((q & 7) == 3 ? x2 + (size_t)(q >> 3) * 9 : x1 + (size_t)(q >> 2) * 9);
        };
        auto solve_from = [&](int q, int r, int h) {            // row r of separator q from its P rows and the solutions h away
            const double* P = V.csol2 + (rb + q) * 171;
            double x = P[162 + r];
            if (q - h >= 0) {
                const double* xm = x_odd(q - h);
#pragma unroll
                for (int k = 0; k < 9; ++k) x -= P[r * 9 + k] * xm[k];
            }
            if (q + h < n0) {
                const double* xp = x_odd(q + h);
#pragma unroll
                for (int k = 0; k < 9; ++k) x -= P[81 + r * 9 + k] * xp[k];
            }
            return x;
        };
        for (int idx = tid; idx < ((n0 + 7) / 8) * 9; idx += kCrThreads) {
            const int q = 8 * (idx / 9) + 3, r = idx % 9;
            if (q < n0) x2[(size_t)(q >> 3) * 9 + r] = solve_from(q, r, 4);
        }
        __syncthreads();
        for (int idx = tid; idx < ((n0 + 3) / 4) * 9; idx += kCrThreads) {
            const int q = 4 * (idx / 9) + 1, r = idx % 9;
            if (q < n0) x1[(size_t)(q >> 2) * 9 + r] = solve_from(q, r, 2);
        }
        __syncthreads();
        for (int idx = tid; idx < n0 * 9; idx += kCrThreads) {
            const int q = idx / 9, r = idx % 9;
            V.rx[rb * 9 + idx] = (q & 1) ? x_odd(q)[r] : solve_from(q, r, 1);
        }
    } else if (PRE == 2) {
        // separators 4b+3 are the blocks solved here.  4b+1 come from the level-1 eliminations, x = Pg - PL x_{q-2} - PU x_{q+2},
        // then the even ones from level 0, x = Pg - PL x_{q-1} - PU x_{q+1}
        double* x1 = smem + (size_t)n1 * 252;       // [ceil(n0 / 4)][9]: the level-1 solutions (behind the blocks)
        auto x_odd = [&](int q) -> const double* {  // solution of an odd separator
            return (q & 3) == 3 ? smem + (size_t)(q >> 2) * 252 + 243 : x1 + (size_t)(q >> 2) * 9;
        };
        for (int idx = tid; idx < ((n0 + 3) / 4) * 9; idx += kCrThreads) {
            const int q = 4 * (idx / 9) + 1, r = idx % 9;
            if (q < n0) {
                const double* P = V.csol2 + (rb + q) * 171;
                double x = P[162 + r];
                if (q >= 3) {
                    const double* xm = smem + (size_t)((q - 2) >> 2) * 252 + 243;
#pragma unroll
                    for (int k = 0; k < 9; ++k) x -= P[r * 9 + k] * xm[k];
                }
                if (q + 2 < n0) {
                    const double* xp = smem + (size_t)((q + 2) >> 2) * 252 + 243;
#pragma unroll
                    for (int k = 0; k < 9; ++k) x -= P[81 + r * 9 + k] * xp[k];
                }
                x1[(size_t)(q >> 2) * 9 + r] = x;
            }
        }
        __syncthreads();
        VBA_KSTAMP(tson, tsi++);
        for (int idx = tid; idx < n0 * 9; idx += kCrThreads) {
            const int q = idx / 9, r = idx % 9;
            double x;
            if (q & 1) {
                x = x_odd(q)[r];
            } else {
                const double* P = V.csol2 + (rb + q) * 171;
                x = P[162 + r];
                if (q >= 1) {
                    const double* xm = x_odd(q - 1);
#pragma unroll
                    for (int k = 0; k < 9; ++k) x -= P[r * 9 + k] * xm[k];
                }
                if (q + 1 < n0) {
                    const double* xp = x_odd(q + 1);
#pragma unroll
                    for (int k = 0; k < 9; ++k) x -= P[81 + r * 9 + k] * xp[k];
                }
            }
            V.rx[rb * 9 + idx] = x;
        }
    } else if (!PRE) {
        for (int idx = tid; idx < n1 * 9; idx += kCrThreads) V.rx[rb * 9 + idx] = smem[(size_t)(idx / 9) * 252 + 243 + idx % 9];
    } else {
        // separators 2t+1 are the blocks solved here; 2t come from the level-0 eliminations:
        //   x_{2t} = Pg - PL x_{2t-1} - PU x_{2t+1}
        for (int idx = tid; idx < n0 * 9; idx += kCrThreads) {
            const int q = idx / 9, r = idx % 9;
            double x;
            if (q & 1) {
                x = smem[(size_t)(q >> 1) * 252 + 243 + r];
            } else {
                const double* P = V.csol2 + (rb + q) * 171;
                x = P[162 + r];
                if (q >= 1) {
                    const double* xm = smem + (size_t)((q - 1) >> 1) * 252 + 243;
#pragma unroll
                    for (int k = 0; k < 9; ++k) x -= P[r * 9 + k] * xm[k];
                }
                if (q + 1 < n0) {
                    const double* xp = smem + (size_t)((q + 1) >> 1) * 252 + 243;
#pragma unroll
                    for (int k = 0; k < 9; ++k) x -= P[81 + r * 9 + k] * xp[k];
                }
            }
            V.rx[rb * 9 + idx] = x;
        }
    }
    VBA_KSTAMP(tson, tsi++);
    (void)tsi; (void)tson;
    report_pivot<PIVOT>(bad, sc, lane, V.par);
}

template <bool PIVOT, int PRE>
__global__ __launch_bounds__(kCrThreads) void k_solve_reduced_cr(DevView V, int s) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    reduced_cr_body<PIVOT, PRE, kCrThreads>(V, s, blockIdx.x, smem);
}

#ifdef VBA_VARIANTS   // the solve as ONE grid of producer and waiting consumer blocks (VBA_OPT_FUSION bits 5, 6): measured slower, comparison builds
// ------------------------------------------------------------------------------------------------ resident solve
// The three launches of the latency-mode solve (chunk elimination -> cyclic-reduction levels 0 + 1 -> the remaining levels
// in one workgroup) as ONE grid whose consumer blocks are resident from the start and wait for their producers on flags
// (VBA_OPT_FUSION bit 5).  Block x of window y is
//   x <  P          : chunk x                        (produces flag x)
//   x <  P + G      : cyclic-reduction group x - P   (waits for chunks 4t .. 4t + 7, produces flag x)
//   x == P + G      : the one-workgroup tail         (TAIL; waits for all groups)
// The grid (at C3: 63 + 16 + 1 blocks) is far below what the 256 CUs hold at once and blocks are dispatched in index order, so
// every producer is running or done when a consumer starts to wait.  A flag holds the EPOCH of the launch that wrote it
// (a counter the host increments per launch, so nothing is ever reset) and is stored by the last wave of the block to get
// there, EVERY wave passing through resident_publish whatever path it took through its role (windows that skip the call,
// short windows, a failed pivot check) -- a consumer can therefore never wait for a block that has nothing to say.  The
// wait is bounded all the same: kResidentSpins polls (> 100 ms) and the window is flagged (fl bit 64 -> VBA_ESTATE).
// Same bodies, same operations, same bits as the three launches.
constexpr int kResidentSpins = 1 << 18;      // polls; one is a round trip to memory, ~1 us

__device__ __forceinline__ void resident_publish(unsigned* flag, unsigned epoch, unsigned* lds_count, int nwaves) {
    __threadfence();            // this wave's stores are visible device-wide before it is counted
    if ((threadIdx.x & 63) == 0) {
        const unsigned before = atomicAdd(lds_count, 1u);
        // (relaxed: the fences above have released every wave's stores; a release store would write the L2 back once more)
        if (before == (unsigned)nwaves - 1u) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// every wave waits by itself: lane l < count watches flags[first + l]; false when the bound was hit
__device__ __forceinline__ bool resident_wait(const unsigned* flags, int first, int count, unsigned epoch) {
    const int lane = threadIdx.x & 63;
    bool ok = true;
    for (int base = 0; base < count; base += 64) {
        const bool mine = base + lane < count;
        const unsigned* f = flags + first + (mine ? base + lane : 0);
        int spins = 0;
        for (;;) {
            const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool there = !mine || (int)(v - epoch) >= 0;
            if (__all(there)) break;
            if (++spins > kResidentSpins) { ok = false; break; }
        }
        if (!ok) break;
    }
    __threadfence();            // acquire: nothing read below is older than the flags
    return ok;
}

template <bool PIVOT, bool REG, bool TAIL>
__global__ __launch_bounds__(TAIL ? 512 : 256) void k_solve_resident(DevView V, int s, int P, int G, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ unsigned arrived;
    constexpr int kThreads = TAIL ? 512 : 256;
    const int w = blockIdx.y, x = blockIdx.x, tid = threadIdx.x;
    unsigned* flags = V.res_flags + (size_t)w * V.res_stride;
    if (tid == 0) arrived = 0u;
#ifdef VBA_RESIDENT_STAMPS
    // diagnostic build: 100 MHz wall clock at entry / after the wait / after the body / after the publish, wave 0 of every block
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(V.cR2) + ((size_t)w * V.res_stride + x) * 4;
#define VBA_RSTAMP(k) do { if (tid == 0) stamp[k] = wall_clock64(); } while (0)
#else
#define VBA_RSTAMP(k) do {} while (0)
#endif
    VBA_RSTAMP(0);
    __syncthreads();
    if (x < P) {
        VBA_RSTAMP(1);
        if (tid < 256) chunks_ts_fused_body<PIVOT, REG>(V, s, w, x, smem);
    } else if (x < P + G) {
        const int t = x - P;
        if (tid < 256) {
            const int first = 4 * t, last = 4 * t + 7 < P - 1 ? 4 * t + 7 : P - 1;
            if (!resident_wait(flags, first, last - first + 1, epoch) && (tid & 63) == 0) atomicOr(&V.sc[w].fl[V.par], 2u | 64u);
            VBA_RSTAMP(1);
            cr_level01_body<PIVOT>(V, s, w, t, smem);
        }
    } else if (TAIL) {
        if (!resident_wait(flags, P, G, epoch) && (tid & 63) == 0) atomicOr(&V.sc[w].fl[V.par], 2u | 64u);
        VBA_RSTAMP(1);
        reduced_cr_body<PIVOT, 2, kThreads>(V, s, w, smem);
    }
    VBA_RSTAMP(2);
    resident_publish(flags + x, epoch, &arrived, kThreads / 64);
    VBA_RSTAMP(3);
#undef VBA_RSTAMP
}

#endif  // VBA_VARIANTS

// Recovery of a partitioned chain: x_i = yhat_i - Vhat_i x_left - What_i x_right for interior blocks, separators
// copied from the reduced solution.
__device__ __forceinline__ void recover_block(int i, int n, int s, const double* csol, const double* xsep, double (&d9)[9]) {
    const int c = i / s;
    const int P = (n + s - 1) / s;
    const bool is_sep = (c < P - 1) && (i == (c + 1) * s - 1);
    if (is_sep) {
#pragma unroll
        for (int r = 0; r < 9; ++r) d9[r] = xsep[(size_t)c * 9 + r];
        return;
    }
    const double* so = csol + (size_t)i * 171;
    double xl[9], xr[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        xl[r] = c > 0 ? xsep[(size_t)(c - 1) * 9 + r] : 0.0;
        xr[r] = c < P - 1 ? xsep[(size_t)c * 9 + r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        double v = so[r];
#pragma unroll
        for (int k = 0; k < 9; ++k) v -= so[(1 + k) * 9 + r] * xl[k] + so[(10 + k) * 9 + r] * xr[k];
        d9[r] = v;
    }
}

// level 2 -> level 1: the solution of every level-1 separator
__global__ __launch_bounds__(64) void k_solve_recover2(DevView V, int s, int s2) {
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done) return;
    const int n1 = n_separators(V.n[w], s);
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= n1) return;
    const size_t rb = (size_t)w * V.p_max;
    double d9[9];
    recover_block(q, n1, s2, V.csol2 + rb * 171, V.rx2 + rb * 9, d9);
#pragma unroll
    for (int r = 0; r < 9; ++r) V.rx[(rb + q) * 9 + r] = d9[r];
}

// x_i for every block of the window (s == 0: dpose already holds the solution, block-diagonal phase), then the
// retraction (BA_filtering.py:56-60).
__global__ __launch_bounds__(256) void k_solve_recover(DevView V, int s) {
    // 16 lanes per pose, lane r < 9 forms row r of the step (for a fixed column of csol the nine lanes read nine
    // consecutive doubles); lane 0 of the group gathers the rows and retracts
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done) return;
    const int n = V.n[w];
    const int r = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const size_t sb = (size_t)w * V.n_max;
    const size_t rb = (size_t)w * V.p_max;
    bool bad = false;
    double v = 0.0;
    if (i < n && r < 9) {
        if (s == 0) {
            v = V.dpose[(sb + i) * 9 + r];
        } else {
            const int c = i / s;
            const int P = (n + s - 1) / s;
            const double* xsep = V.rx + rb * 9;
            if ((c < P - 1) && (i == (c + 1) * s - 1)) {
                v = xsep[(size_t)c * 9 + r];            // a separator: copied from the reduced solution
            } else {
                const double* so = V.csol + (sb + i) * 171;
                v = so[r];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const double xl = c > 0 ? xsep[(size_t)(c - 1) * 9 + k] : 0.0;
                    const double xr = c < P - 1 ? xsep[(size_t)c * 9 + k] : 0.0;
                    v -= so[(1 + k) * 9 + r] * xl + so[(10 + k) * 9 + r] * xr;
                }
            }
            V.dpose[(sb + i) * 9 + r] = v;
        }
        bad = !(fabs(v) <= 1.79e308);
    }
    double d9[9];
    const int base = (threadIdx.x & 63) & ~15;
#pragma unroll
    for (int q = 0; q < 9; ++q) d9[q] = __shfl(v, base + q, kWave);
    if (i < n && r == 0) {
        double o[10];
        retract(V.states + (sb + i) * 10, d9, o);
#pragma unroll
        for (int q = 0; q < 10; ++q) V.states_new[(sb + i) * 10 + q] = o[q];
        if (V.host_states) {        // (one-window handles: sb == 0; a pipelined call reads its result from host memory)
#pragma unroll
            for (int q = 0; q < 10; ++q) V.host_states[((size_t)V.par * V.n_max + i) * 10 + q] = o[q];
        }
    }
    const unsigned long long anybad = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && anybad) atomicOr(&sc.fl[V.par], 2u);
}

// Landmark-only phase (initialize): the dynamics factor is absent (BA_utils.py:463-466), the system is block
// DIAGONAL and every pose is an independent 9x9 solve; a wave takes PB consecutive poses.
template <bool PIVOT>
__global__ __launch_bounds__(64) void k_solve_blockdiag(DevView V, int PB) {
    __shared__ double blk[2][128];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done || !solver_mine<PIVOT>(V, sc)) return;
    const int n = V.n[w];
    const int i0 = blockIdx.x * PB;
    if (i0 >= n) return;
    const int lane = threadIdx.x;
    const size_t sb = (size_t)w * V.n_max;
    const double lam32 = (double)(float)sc.lam[V.par];
    if (blockIdx.x == 0 && lane == 0) {
        sc.lam32 = lam32;
        if (PIVOT) atomicAnd(&sc.fl[V.par], ~8u);
    }
    const int cnt = min(PB, n - i0);
    bool badp = false;
    double pre[2];
    auto fetch = [&](int i) {       // diagonal block (81) + right-hand side (9)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            pre[q] = e < 81 ? V.bands[(sb + i) * 243 + 81 + e] : (e < 90 ? V.rhs[(sb + i) * 9 + (e - 81)] : 0.0);
        }
    };
    fetch(i0);
    blk[0][lane] = pre[0];
    blk[0][lane + 64] = pre[1];
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
        const int buf = t & 1;
        if (t + 1 < cnt) fetch(i0 + t + 1);
        double base[9], a[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = 0.0;
            if (lane < 9) v = blk[buf][r * 9 + lane] + (r == lane ? lam32 : 0.0);
            else if (lane == 18) v = blk[buf][81 + r];
            base[r] = v;
            a[r] = 0.0;
        }
        forward_step<0, 1, PIVOT, false>(nullptr, base, a, lane, badp);
        if (lane == 18) {
#pragma unroll
            for (int r = 0; r < 9; ++r) V.dpose[(sb + i0 + t) * 9 + r] = a[r];
        }
        if (t + 1 < cnt) {
            blk[buf ^ 1][lane] = pre[0];
            blk[buf ^ 1][lane + 64] = pre[1];
        }
        __syncthreads();
    }
    report_pivot<PIVOT>(badp, sc, lane, V.par);
}

// ================================================================================================== accept test
// LM accept test (BA_filtering.py:51, 66-79) as its own launch, one block per window: vba_decide.h has the arithmetic.
// ranks == 0: this window's block partials; ranks > 0: the observation part is the rank-ordered sum of the gathered
// per-rank sums (sharded mode).  What a call hands to the next is written to the slots of the other parity; the states
// need no copy (the buffer the last trial wrote IS the next call's input).
__global__ __launch_bounds__(256) void k_decide(DevView V, const double* trial_all, int ranks) {
    __shared__ double red[5][4];
    const int w = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    const int t = threadIdx.x;
    const int par = V.par;
    if (sc.done || (V.pending_only && sc.pending != V.call)) return;
    const int n_trials = sc.n_trials;
    const double lam32 = sc.lam32;
    const DecideOut d = decide_eval(V, w, par, V.prm, n_trials, sc.init_residual, trial_all, ranks, red);
    unsigned* hist_next = hist0_of(V, w, par ^ 1);      // filled by k_trial<true>
    if (d.flags & 8u) {             // the un-pivoted solve failed its check: the host repeats it with pivoting
        if (V.emit) for (int b = t; b < kSelBins; b += 256) hist_next[b] = 0u;
        if (t == 0) {
            sc.miss = 0;
            V.host_head[w].flags = d.flags;
            V.host_head[w].done = 0;
        }
        return;
    }
    if (d.stop) {
        if (t < 81) {
            const double hv = V.lastD[(size_t)w * 81 + t] + ((t / 9 == t % 9) ? lam32 : 0.0);
            sc.last_hessian[t] = hv;
            if (V.host_states) V.host_head[w].last_hessian[t] = hv;     // (a pipelined call reads its result from host memory)
        }
    } else if (V.emit) {            // rejected: the next trial histograms its own keys
        for (int b = t; b < kSelBins; b += 256) hist_next[b] = 0u;
    }
    if (t == 0) {
        unsigned fl = d.flags;
        if (n_trials == 0) {
            sc.sum_abs_rpred = d.sum_pred;
            sc.init_residual = d.init_residual;
        }
        sc.trial_residual = d.residual;
        sc.n_trials = n_trials + 1;
        sc.pending = -1;
        sc.miss = 0;
        int call_idx = sc.call_idx;
        double lam_now = d.lam_next;
        if (d.stop) {
            sc.done = 1;
            if (V.call >= 0) sc.call_idx = call_idx = V.call + 1;
            if (!d.accept) fl |= 1u;
            if (!(d.residual == d.residual)) fl |= 2u;
            sc.fl[par] = fl;
            sc.lam[par ^ 1] = lam_now = d.lam_out;
            if (V.emit) sc.sum_in[par ^ 1] = d.sum_next;
        } else {
            sc.lam[par] = lam_now;
        }
        // read by the host after it has waited for the stream: no fence needed
        WinHead& hh = V.host_head[w];
        hh.lamda = lam_now;
        hh.trial_residual = d.residual;
        hh.n_trials = n_trials + 1;
        hh.flags = fl;
        hh.done = d.stop ? 1 : 0;
        hh.call_idx = call_idx;
    }
}

// latency mode with a partitioned chain whose chunk (blocks + staged inputs + elimination scratch) fits the LDS of a CU:
// the chunk kernel forms its blocks itself and k_assemble is not launched (vba_api.hip asks the same question)
// ... and the sequential walk of the batched mode forms them pose by pose (VBA_OPT_FUSION bit 2)
#ifdef VBA_VARIANTS
static bool walk_forms_blocks(const DevView& V) { return !V.lat && V.fuse_walk && !V.prm.initialize && V.chunk <= 0 && V.pack != 1; }
#else   // (one window per wavefront forming its own blocks -- k_solve_forming -- is a comparison variant)
static bool walk_forms_blocks(const DevView& V) { return !V.lat && V.fuse_walk && !V.prm.initialize && V.chunk <= 0 && V.pack == 2; }
#endif
bool solve_forms_blocks(const DevView& V) {
    // (latency-mode handles only.  Round 5 measured the forming chunk elimination on bandwidth-mode handles, chunks of 12: 260 / 324 /
    // 337 k it/s at 64 / 256 / 512 C3 windows against 270 / 347 / 367 with k_assemble_rows + k_solve_chunks_ts -- its 256-thread
    // blocks with the staged inputs in LDS leave fewer chunks resident than the assembly launch costs)
    return walk_forms_blocks(V) || (V.lat && V.fuse_blocks && !V.prm.initialize && V.chunk >= 2 && V.chunk <= kFusedChunkMax);
}

#ifdef VBA_VARIANTS
static std::atomic<unsigned> g_resident_epoch{0u};    // flags of k_solve_resident: one value per launch, process wide
#endif

template <bool PIVOT>
static void launch_solve_variant(const DevView& V, int initialize, hipStream_t s) {
    if (initialize) {       // block diagonal: independent poses
        const int PB = V.W >= 64 ? 8 : 2;
        hipLaunchKernelGGL(k_solve_blockdiag<PIVOT>, dim3((V.n_max + PB - 1) / PB, V.W), dim3(64), 0, s, V, PB);
        return;
    }
    if (V.chunk <= 0) {
        if (V.pack == 2) {      // four windows per wavefront (DPP row broadcasts)
            const dim3 g((V.W + kQuad - 1) / kQuad), b(64);
            if (!walk_forms_blocks(V)) hipLaunchKernelGGL((k_solve_quad<PIVOT, false, false>), g, b, 0, s, V);
            else if (V.reg) hipLaunchKernelGGL((k_solve_quad<PIVOT, true, true>), g, b, 0, s, V);
            else hipLaunchKernelGGL((k_solve_quad<PIVOT, true, false>), g, b, 0, s, V);
        }
#ifdef VBA_VARIANTS
        else if (V.pack) hipLaunchKernelGGL(k_solve_packed<PIVOT>, dim3((V.W + kPack - 1) / kPack), dim3(64), 0, s, V);   // equal pose counts: three windows per wavefront
        else if (walk_forms_blocks(V)) {     // (V.pack == 0: one window per wavefront)
            if (V.reg) hipLaunchKernelGGL((k_solve_forming<PIVOT, true>), dim3(V.W), dim3(64), 0, s, V);
            else hipLaunchKernelGGL((k_solve_forming<PIVOT, false>), dim3(V.W), dim3(64), 0, s, V);
        }
#endif
        else hipLaunchKernelGGL(k_solve<PIVOT>, dim3(V.W), dim3(64), 0, s, V);      // one window per wavefront
        return;
    }
    const int cs = V.chunk, cs2 = V.chunk2;
    const int P = (V.n_max + cs - 1) / cs;
    const size_t lds = (512 + (size_t)cs * 252 + 162) * sizeof(double);
#ifdef VBA_VARIANTS
    const int n0_all = P - 1;
    const bool resident = V.resident && solve_forms_blocks(V) && V.chunk_waves == 2 && cs >= 4 && cs2 < 0 && V.cr_levels == 2 &&
                          n0_all >= kCrSplitMin && n0_all <= 4 * kCrMax;
    if (resident) {
        // one grid: chunks, cyclic-reduction groups and (V.resident == 2) the one-workgroup tail, see k_solve_resident
        const bool reg = V.reg != 0, tail = V.resident == 2;
        const int G = (n0_all + 3) / 4;
        size_t lds_all = (size_t)twosided_fused_lds_doubles(cs, reg) * sizeof(double);
        if (lds_all < 7 * 252 * sizeof(double)) lds_all = 7 * 252 * sizeof(double);
        const size_t lds_tail = ((size_t)(n0_all / 4) * 252 + (size_t)((n0_all + 3) / 4) * 9) * sizeof(double);
        if (tail && lds_all < lds_tail) lds_all = lds_tail;
        const unsigned epoch = ++g_resident_epoch;
        const dim3 grid(P + G + (tail ? 1 : 0), V.W);
        if (tail) {
            if (reg) hipLaunchKernelGGL((k_solve_resident<PIVOT, true, true>), grid, dim3(512), lds_all, s, V, cs, P, G, epoch);
            else hipLaunchKernelGGL((k_solve_resident<PIVOT, false, true>), grid, dim3(512), lds_all, s, V, cs, P, G, epoch);
        } else {
            if (reg) hipLaunchKernelGGL((k_solve_resident<PIVOT, true, false>), grid, dim3(256), lds_all, s, V, cs, P, G, epoch);
            else hipLaunchKernelGGL((k_solve_resident<PIVOT, false, false>), grid, dim3(256), lds_all, s, V, cs, P, G, epoch);
            hipLaunchKernelGGL((k_solve_reduced_cr<PIVOT, 2>), dim3(V.W), dim3(kCrThreads),
                               ((size_t)(n0_all / 4) * 252 + (size_t)((n0_all + 3) / 4) * 9) * sizeof(double), s, V, cs);
        }
        const int n0_min = (V.n_min + cs - 1) / cs - 1;
        if (n0_min < kCrSplitMin)       // short windows of the handle: the one-workgroup variant as before
            hipLaunchKernelGGL((k_solve_reduced_cr<PIVOT, 0>), dim3(V.W), dim3(kCrThreads), (size_t)(kCrSplitMin - 1) * 252 * sizeof(double), s, V, cs);
        return;
    }
#endif
    if (solve_forms_blocks(V) && V.chunk_waves == 2 && cs >= 4) {
        const bool reg = V.reg != 0;
        const size_t ldsf = (size_t)twosided_fused_lds_doubles(cs, reg) * sizeof(double);
        if (reg) hipLaunchKernelGGL((k_solve_chunks_ts_fused<PIVOT, true>), dim3(P, V.W), dim3(256), ldsf, s, V, cs);
        else hipLaunchKernelGGL((k_solve_chunks_ts_fused<PIVOT, false>), dim3(P, V.W), dim3(256), ldsf, s, V, cs);
    } else if (solve_forms_blocks(V)) {
        const bool reg = V.reg != 0;
        const size_t ldsf = lds + ((size_t)(cs + 1) * 252 + (size_t)(cs + 2) * (kAsmBase + (reg ? kAsmPrior : 0))) * sizeof(double);
        if (reg) hipLaunchKernelGGL((k_solve_chunks_fused<PIVOT, true>), dim3(P, V.W), dim3(256), ldsf, s, V, cs);
        else hipLaunchKernelGGL((k_solve_chunks_fused<PIVOT, false>), dim3(P, V.W), dim3(256), ldsf, s, V, cs);
    } else if (V.chunk_waves == 2 && cs >= 4) {
        hipLaunchKernelGGL(k_solve_chunks_ts<PIVOT>, dim3(P, V.W), dim3(128), (size_t)twosided_lds_doubles(cs) * sizeof(double), s, V, cs);
    } else {
        hipLaunchKernelGGL(k_solve_chunks<PIVOT>, dim3(P, V.W), dim3(64), lds, s, V, cs);
    }
    if (cs2 > 0) {      // second level over the P-1 separators
        const int P2 = (P - 1 + cs2 - 1) / cs2;
        const size_t lds2 = (512 + (size_t)cs2 * 252 + 162) * sizeof(double);
        hipLaunchKernelGGL(k_solve_chunks2<PIVOT>, dim3(P2 > 0 ? P2 : 1, V.W), dim3(64), lds2, s, V, cs, cs2);
    }
    if (cs2 < 0) {      // one level, the reduced system by cyclic reduction (every window picks its variant by its own size)
        const int n0_max = P - 1, n0_min = (V.n_min + cs - 1) / cs - 1;
        if (n0_max >= kCrSplitMin) {    // first level(s) on their own CUs, the rest in one workgroup
#ifdef VBA_VARIANTS
            if (V.cr_levels == 3) {
                hipLaunchKernelGGL(k_cr_level012<PIVOT>, dim3((n0_max + 7) / 8, V.W), dim3(512), 0, s, V, cs);
                hipLaunchKernelGGL((k_solve_reduced_cr<PIVOT, 3>), dim3(V.W), dim3(kCrThreads),
                                   ((size_t)(n0_max / 8) * 252 + (size_t)((n0_max + 7) / 8) * 9 + (size_t)((n0_max + 3) / 4) * 9) * sizeof(double), s, V, cs);
            } else if (V.cr_levels == 2)
#endif
            {
                hipLaunchKernelGGL(k_cr_level01<PIVOT>, dim3((n0_max + 3) / 4, V.W), dim3(256), 0, s, V, cs);
                hipLaunchKernelGGL((k_solve_reduced_cr<PIVOT, 2>), dim3(V.W), dim3(kCrThreads),
                                   ((size_t)(n0_max / 4) * 252 + (size_t)((n0_max + 3) / 4) * 9) * sizeof(double), s, V, cs);
            }
#ifdef VBA_VARIANTS
            else {
                hipLaunchKernelGGL(k_cr_level0<PIVOT>, dim3((n0_max + 1) / 2, V.W), dim3(128), 0, s, V, cs);
                hipLaunchKernelGGL((k_solve_reduced_cr<PIVOT, 1>), dim3(V.W), dim3(kCrThreads), (size_t)(n0_max / 2) * 252 * sizeof(double), s, V, cs);
            }
#endif
        }
        if (n0_min < kCrSplitMin && n0_max > 0) {
            const int nb = n0_max < kCrSplitMin ? n0_max : kCrSplitMin - 1;
            hipLaunchKernelGGL((k_solve_reduced_cr<PIVOT, 0>), dim3(V.W), dim3(kCrThreads), (size_t)nb * 252 * sizeof(double), s, V, cs);
        }
        return;
    }
    hipLaunchKernelGGL(k_solve_reduced<PIVOT>, dim3(V.W), dim3(64), 0, s, V, cs, cs2);
}

// Dynamic-LDS limits of the solver kernels.  HIP function attributes are per DEVICE, so this runs in every vba_create
// (after hipSetDevice); a refused size is reported there instead of surfacing later as a failed launch.
hipError_t configure_solver_device() {
    const int cap = (int)((512 + 60 * 252 + 162) * sizeof(double));     // chunks above ~30 poses exceed the default 64 KiB
    const int cap_cr = kCrMax * 252 * (int)sizeof(double);
    const int cap_f = (int)((512 + kFusedChunkMax * 252 + 162 + (kFusedChunkMax + 1) * 252 + (kFusedChunkMax + 2) * (kAsmBase + kAsmPrior)) * sizeof(double));
#ifdef VBA_VARIANTS
    const int cap_ts = twosided_fused_lds_doubles(kFusedChunkMax, true) * 8;
    const int cap_res = cap_ts > cap_cr + 65 * 9 * 8 ? cap_ts : cap_cr + 65 * 9 * 8;
#endif
    const struct { const void* fn; int bytes; } set[] = {
        {reinterpret_cast<const void*>(k_solve_chunks_fused<false, false>), cap_f}, {reinterpret_cast<const void*>(k_solve_chunks_fused<true, false>), cap_f},
        {reinterpret_cast<const void*>(k_solve_chunks_fused<false, true>), cap_f}, {reinterpret_cast<const void*>(k_solve_chunks_fused<true, true>), cap_f},
        {reinterpret_cast<const void*>(k_solve_chunks_ts_fused<false, false>), twosided_fused_lds_doubles(kFusedChunkMax, true) * 8},
        {reinterpret_cast<const void*>(k_solve_chunks_ts_fused<true, false>), twosided_fused_lds_doubles(kFusedChunkMax, true) * 8},
        {reinterpret_cast<const void*>(k_solve_chunks_ts_fused<false, true>), twosided_fused_lds_doubles(kFusedChunkMax, true) * 8},
        {reinterpret_cast<const void*>(k_solve_chunks_ts_fused<true, true>), twosided_fused_lds_doubles(kFusedChunkMax, true) * 8},
        {reinterpret_cast<const void*>(k_solve_chunks<false>), cap}, {reinterpret_cast<const void*>(k_solve_chunks<true>), cap},
        {reinterpret_cast<const void*>(k_solve_chunks_ts<false>), twosided_lds_doubles(60) * 8}, {reinterpret_cast<const void*>(k_solve_chunks_ts<true>), twosided_lds_doubles(60) * 8},
        {reinterpret_cast<const void*>(k_solve_chunks2<false>), cap}, {reinterpret_cast<const void*>(k_solve_chunks2<true>), cap},
        {reinterpret_cast<const void*>(k_solve_reduced_cr<false, 0>), cap_cr}, {reinterpret_cast<const void*>(k_solve_reduced_cr<true, 0>), cap_cr},
#ifdef VBA_VARIANTS
        {reinterpret_cast<const void*>(k_solve_reduced_cr<false, 3>), cap_cr + 200 * 9 * 8}, {reinterpret_cast<const void*>(k_solve_reduced_cr<true, 3>), cap_cr + 200 * 9 * 8},
        {reinterpret_cast<const void*>(k_solve_reduced_cr<false, 1>), cap_cr}, {reinterpret_cast<const void*>(k_solve_reduced_cr<true, 1>), cap_cr},
        {reinterpret_cast<const void*>(k_solve_resident<false, false, false>), cap_res}, {reinterpret_cast<const void*>(k_solve_resident<true, false, false>), cap_res},
        {reinterpret_cast<const void*>(k_solve_resident<false, true, false>), cap_res}, {reinterpret_cast<const void*>(k_solve_resident<true, true, false>), cap_res},
        {reinterpret_cast<const void*>(k_solve_resident<false, false, true>), cap_res}, {reinterpret_cast<const void*>(k_solve_resident<true, false, true>), cap_res},
        {reinterpret_cast<const void*>(k_solve_resident<false, true, true>), cap_res}, {reinterpret_cast<const void*>(k_solve_resident<true, true, true>), cap_res},
#endif
        {reinterpret_cast<const void*>(k_solve_reduced_cr<false, 2>), cap_cr + 65 * 9 * 8}, {reinterpret_cast<const void*>(k_solve_reduced_cr<true, 2>), cap_cr + 65 * 9 * 8}};
    for (const auto& e : set) {
        const hipError_t rc = hipFuncSetAttribute(e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, e.bytes);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

void launch_solve(const DevView& V, int initialize, hipStream_t s) {
    if (V.pivot != 1) launch_solve_variant<false>(V, initialize, s);
    if (V.pivot != 0) launch_solve_variant<true>(V, initialize, s);
    // interiors / retraction: shared by both variants (k_solve and k_solve_packed retract themselves)
    if (initialize) hipLaunchKernelGGL(k_solve_recover, dim3((V.n_max + 15) / 16, V.W), dim3(256), 0, s, V, 0);
    else if (V.chunk > 0) {
        if (V.chunk2 > 0) hipLaunchKernelGGL(k_solve_recover2, dim3((V.p_max + 63) / 64, V.W), dim3(64), 0, s, V, V.chunk, V.chunk2);
        // latency mode: the trial kernel recovers the interiors and retracts (V.fused_trial == 2)
        if (V.fused_trial != 2) hipLaunchKernelGGL(k_solve_recover, dim3((V.n_max + 15) / 16, V.W), dim3(256), 0, s, V, V.chunk);
    }
}

void launch_decide(const DevView& V, const double* trial_all, int ranks, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(V.W), dim3(256), 0, s, V, trial_all, ranks);
}

}  // namespace vba
