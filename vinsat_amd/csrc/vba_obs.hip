// vba_obs.hip -- observation-indexed kernels of the BA iteration (gfx950).
//
//   k_obs_residual     A1: reprojection residuals at the input states, |r| keys, sum |r|
//   k_select_pass<P>   A3a: exact lower median of the 2m keys by most-significant-digit radix select:
//                      digit 0 (exponent) is histogrammed inside k_obs_residual, digits 1 and 2 read the keys
//                      once each, the second of them compacting the (few) keys that match the 32 known bits,
//                      and select_finish (prologue of k_obs_accumulate) finishes digits 3..5 on that short list
//   k_obs_accumulate<G> A2 + A3a + A3b: Jacobian, robust weight, per-pose 6x6 / 6 accumulation (G lanes per pose)
//   k_trial            A8: weighted trial residuals (observations) and dynamics residuals at the trial states
//   k_debug_project    recompute est / Jacobian at the step's input states for vba_debug_fetch
//
// All of these stream the observation arrays once, coalesced (SoA, 8 B per lane per array); the pose state
// is gathered through L1/L2 (observations are pose sorted, so a wave touches one or two poses).
#include "vba_device.h"
#include "vba_dyn_body.h"
#include "vba_launch.h"

namespace vba {

// Per-step state that must be clean before the first kernel touches it:
//   * radix histograms: zeroed by k_assemble of the PREVIOUS step (and by the allocation), because the first
//     kernel of a step already accumulates digit 0 into them;
//   * scalars (done, n_trials, flags, max weight, list length): reset by thread 0 of block 0 of k_obs_residual,
//     no later block or kernel of the step reads them before the next kernel boundary;
//   * states_prev (debug copy of the step's input): written by k_decide just before it commits the new states.
__device__ __forceinline__ void reset_step_scalars(WinScalars& sc) {
    sc.done = 0;
    sc.n_trials = 0;
    sc.flags = 0u;
    sc.wmax_bits = 0ull;
    sc.sum_abs_rpred = 0.0;
    sc.sel_cnt = 0u;
}

// ---------------------------------------------------------------------------------------------- A1
// HIST0: also histogram the top radix digit (the 10 exponent bits) of the keys this block produced.
template <bool HIST0>
__global__ __launch_bounds__(kObsBlock) void k_obs_residual(DevView V, double* abs_out /*null: V.absr*/) {
    __shared__ double red[kObsBlock / 64];
    __shared__ unsigned lh[HIST0 ? 1024 : 1];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int m = V.m[w];
    const size_t ob = (size_t)w * V.obs_stride;     // observation block of the window
    const size_t mb = (size_t)w * V.m_max;          // per-observation work arrays
    const int k = blockIdx.x * kObsBlock + threadIdx.x;
    if (HIST0) {
        for (int b = threadIdx.x; b < 1024; b += kObsBlock) lh[b] = 0u;
        __syncthreads();
    }
    double s = 0.0;
    if (k < m) {
        const int pose = V.opose[2 * ob + k];
        const size_t pb = (size_t)w * V.n_max + pose;
        PoseCam pc;
        pose_camera(V.states + pb * 10, V.intr + pb * 4, pc);
        double u, v, cam[3], d;
        project(pc, V.ox[ob + k], V.oy[ob + k], V.oz[ob + k], u, v, cam, d);
        const double ru = fabs(V.ou[ob + k] - u), rv = fabs(V.ov[ob + k] - v);
        double* ab = abs_out ? abs_out : V.absr + 2 * mb;
        reinterpret_cast<double2*>(ab)[k] = make_double2(ru, rv);
        s = ru + rv;
        if (HIST0) {
            atomicAdd(&lh[(unsigned)(f64_bits(ru) >> 53) & 1023u], 1u);
            atomicAdd(&lh[(unsigned)(f64_bits(rv) >> 53) & 1023u], 1u);
        }
    }
    const double t = block_sum<kObsBlock>(s, red);
    if (threadIdx.x == 0) V.part_init[(size_t)w * V.nblk_obs + blockIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) reset_step_scalars(V.sc[w]);
    if (HIST0) {
        unsigned* hist = V.hist + (size_t)w * kSelPasses * kSelBins;
        for (int b = threadIdx.x; b < 1024; b += kObsBlock) {
            const unsigned c = lh[b];
            if (c) atomicAdd(&hist[b], c);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            V.sc[w].sel_prefix[0] = 0ull;
            V.sc[w].sel_rank[0] = (2 * (long long)m - 1) / 2;
        }
    }
}

// ---------------------------------------------------------------------------------------------- A3a: select
// COMPACT: additionally append the keys that match the digits known so far to the short list V.ckeys.
// ITEMS keys per thread: 8 keeps a single window spread over many blocks (latency), 32 amortises the per-block
// prologue (histogram scan, LDS clear, flush) when many windows are batched.
template <int P, bool COMPACT, int ITEMS>
__global__ __launch_bounds__(256) void k_select_pass(DevView V) {
    __shared__ unsigned lh[kSelBins];
    __shared__ unsigned lds_u[260];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const double* keys = V.abs_all ? V.abs_all : V.absr + 2 * (size_t)w * V.m_max;
    const int64_t count = V.abs_all ? V.abs_all_count : 2 * (int64_t)V.m[w];
    // carried keys: k_obs_residual did not run, this is the first kernel of the call and owns the scalar reset
    if (P == 1 && V.carry && blockIdx.x == 0 && threadIdx.x == 0) reset_step_scalars(V.sc[w]);
    if ((int64_t)blockIdx.x * 256 * ITEMS >= count) return;
    unsigned* hist = V.hist + (size_t)w * kSelPasses * kSelBins;
    constexpr int nbins = 1 << sel_width(P);
    for (int b = threadIdx.x; b < nbins; b += 256) lh[b] = 0u;
    // few keys per thread (single window, latency matters): their loads are issued before the histogram of the
    // previous digit is resolved, not after
    constexpr bool PRELOAD = ITEMS <= 8;
    unsigned long long pk[PRELOAD ? ITEMS : 1];
    if (PRELOAD) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int64_t idx = ((int64_t)blockIdx.x * ITEMS + it) * 256 + threadIdx.x;
            pk[it] = idx < count ? f64_bits(keys[idx]) : 0ull;
        }
    }
    unsigned long long prefix = 0ull;
    // torch.median = lower median (BA_filtering.py:23); in sharded mode the gathered buffer may end in +inf padding
    long long rank = ((V.m_total ? 2 * V.m_total : count) - 1) / 2;
    if (P > 0) {
        constexpr int Q = P > 0 ? P - 1 : 0;
        select_resolve(hist + Q * kSelBins, 1 << sel_width(Q), sel_width(Q), V.sc[w].sel_prefix[Q], V.sc[w].sel_rank[Q],
                       prefix, rank, lds_u);
    } else {
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        V.sc[w].sel_prefix[P] = prefix;
        V.sc[w].sel_rank[P] = rank;
    }
#pragma unroll 8
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * ITEMS + it) * 256 + threadIdx.x;
        if (idx < count) {
            const unsigned long long key = PRELOAD ? pk[it] : f64_bits(keys[idx]);
            bool match = true;
            if (P > 0) match = (key >> sel_shift(P > 0 ? P - 1 : 0)) == prefix;
            if (match) atomicAdd(&lh[(unsigned)(key >> sel_shift(P)) & (nbins - 1)], 1u);
            if (COMPACT) {
                // wave-aggregated append: one atomic per wave instruction
                const unsigned long long mask = __ballot(match);
                if (mask) {
                    const int lane = threadIdx.x & 63;
                    const int leader = __ffsll((long long)mask) - 1;
                    unsigned base = 0;
                    if (lane == leader) base = atomicAdd(&V.sc[w].sel_cnt, (unsigned)__popcll(mask));
                    base = __shfl(base, leader, kWave);
                    if (match) {
                        const unsigned off = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                        // the list has room for 2 m_max keys; if more match (massive ties) select_finish sees
                        // sel_cnt > capacity and rescans the full key array instead
                        if ((int64_t)base + off < 2 * V.m_max) V.ckeys[2 * (size_t)w * V.m_max + base + off] = bits_f64(key);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256) {
        const unsigned c = lh[b];
        if (c) atomicAdd(&hist[P * kSelBins + b], c);
    }
}

// Finishes the select on the compacted list (keys whose top 32 bits are known to match): returns the lower median
// c_obs to every thread of the (256-thread) block.  It is the prologue of k_obs_accumulate -- every block of a window
// redoes it (a handful of keys: rank by counting) instead of one more single-block kernel on the critical path; long
// lists (massive ties) take digits 3, 4, 5 with a block-local histogram each, the full key array if the list
// overflowed.  The histograms it reads are cleared afterwards by k_assemble.
__device__ __forceinline__ double select_finish(const DevView& V, int w, unsigned* lh /*[kSelBins]*/, unsigned* lds_u /*[260]*/,
                                                unsigned long long* skeys /*[1024] + 1*/) {
    const WinScalars& sc = V.sc[w];
    const unsigned* hist = V.hist + (size_t)w * kSelPasses * kSelBins;
    // the list length, the wanted rank and the first 1024 list entries are loaded together (the entries
    // speculatively: the list is almost always that short)
    unsigned cnt = sc.sel_cnt;
    const long long want = sc.sel_rank[2];
    const double* ck = V.ckeys + 2 * (size_t)w * V.m_max;
    unsigned long long pre[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned q = threadIdx.x + 256u * j;
        pre[j] = (int64_t)q < 2 * V.m_max ? f64_bits(ck[q]) : 0ull;
    }
    if (cnt <= 1024u) {
        // every key of the list matches the 21 known bits and the wanted key is the one of rank sel_rank[2] among
        // them -- rank each key by counting (ties broken by position)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned q = threadIdx.x + 256u * j;
            if (q < cnt) skeys[q] = pre[j];
        }
        __syncthreads();
        for (unsigned q = threadIdx.x; q < cnt; q += 256) {
            const unsigned long long key = skeys[q];
            long long below = 0;
            for (unsigned j = 0; j < cnt; ++j) {
                const unsigned long long o = skeys[j];
                below += (o < key) || (o == key && j < q);
            }
            if (below == want) skeys[1024] = key;
        }
        __syncthreads();
        return bits_f64(skeys[1024]);
    }
    if ((int64_t)cnt > 2 * V.m_max) {       // list overflowed: fall back to the full key array
        ck = V.abs_all ? V.abs_all : V.absr + 2 * (size_t)w * V.m_max;
        cnt = (unsigned)(V.abs_all ? V.abs_all_count : 2 * (int64_t)V.m[w]);
    }
    unsigned long long prefix;
    long long rank;
    select_resolve(hist + 2 * kSelBins, 1 << sel_width(2), sel_width(2), sc.sel_prefix[2], sc.sel_rank[2], prefix, rank, lds_u);
#pragma unroll
    for (int P = 3; P < 6; ++P) {
        const int nbins = 1 << sel_width(P);
        for (int b = threadIdx.x; b < kSelBins; b += 256) lh[b] = 0u;
        __syncthreads();
        for (unsigned q = threadIdx.x; q < cnt; q += 256) {
            const unsigned long long key = f64_bits(ck[q]);
            if ((key >> sel_shift(P - 1)) == prefix) atomicAdd(&lh[(unsigned)(key >> sel_shift(P)) & (nbins - 1)], 1u);
        }
        __syncthreads();
        unsigned long long np;
        long long nr;
        select_resolve(lh, nbins, sel_width(P), prefix, rank, np, nr, lds_u);
        prefix = np;
        rank = nr;
    }
    return bits_f64(prefix);
}

// ---------------------------------------------------------------------------------------------- A2 + A3
// G lanes per pose (power of two): every lane strides over its share of the pose's observation segment and
// keeps the 21 + 6 unique entries of sum(w J^T J), sum(w J^T r) in registers; a log2(G)-step xor butterfly
// then gives the lanes of the group the totals.  The shape of the reduction is fixed, so results are bit
// reproducible (no float atomics).  The raw (un-normalised) weight is stored per observation for the trials.
#ifndef VBA_ACC_DEPTH
#define VBA_ACC_DEPTH 2
#endif
constexpr int kAccDepth = VBA_ACC_DEPTH;

// PAIR: a lane takes two consecutive observations per step with 16-byte loads, so that the G lanes of a pose read
// whole 128-byte lines (G = 8) instead of half lines whose other half is fetched again by the next step.
template <int G, bool PAIR>
__global__ __launch_bounds__(256) void k_obs_accumulate(DevView V) {
    __shared__ double wmx[4];
    __shared__ unsigned sel_lh[kSelBins];
    __shared__ unsigned sel_u[260];
    __shared__ unsigned long long sel_keys[1025];
    constexpr int PPB = 256 / G;            // poses per block
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const int nb_acc = (V.n_max * G + 255) / 256;
    if ((int)blockIdx.x >= nb_acc) {        // few windows: the dynamics factor rides in this grid (vba_dyn_body.h)
        dynamics_block(V, w, blockIdx.x - nb_acc);
        return;
    }
    const int n = V.n[w];
    if (blockIdx.x * PPB >= n) return;
    WinScalars& sc = V.sc[w];
    const StepParams& prm = V.prm;
    const int sub = threadIdx.x % G;
    const int i = blockIdx.x * PPB + threadIdx.x / G;
    const size_t pb = (size_t)w * V.n_max + (i < n ? i : 0);
    const size_t ob = (size_t)w * V.obs_stride;
    const size_t mb = (size_t)w * V.m_max;

    struct Obs { double x, y, z, u, v, c; };
    struct alignas(8) D2 { double a, b; };
    struct Obs2 { D2 x, y, z, u, v, c; };
    auto load = [&](int k) {
        Obs o;
        o.x = V.ox[ob + k]; o.y = V.oy[ob + k]; o.z = V.oz[ob + k];
        o.u = V.ou[ob + k]; o.v = V.ov[ob + k]; o.c = V.oconf[ob + k];
        return o;
    };
    auto load2 = [&](int k) {       // observations k, k + 1 (the second may belong to the next pose: masked below)
        Obs2 o;
        o.x = *reinterpret_cast<const D2*>(V.ox + ob + k); o.y = *reinterpret_cast<const D2*>(V.oy + ob + k);
        o.z = *reinterpret_cast<const D2*>(V.oz + ob + k); o.u = *reinterpret_cast<const D2*>(V.ou + ob + k);
        o.v = *reinterpret_cast<const D2*>(V.ov + ob + k); o.c = *reinterpret_cast<const D2*>(V.oconf + ob + k);
        return o;
    };

    // Phase 1: everything that does not need the median is started first (the pose's camera, its row range and the
    // first observations), so that those round trips overlap with the select finish below.
    // Software pipelined: the loads of the next observations are in flight while the current ones are processed (the
    // kernel sits at 2 waves per SIMD because of its accumulators either way; the registers between that and the next
    // occupancy step are spent on memory-level parallelism).
    PoseCam pc{};
    int beg = 0, end = 0;
    Obs ring[kAccDepth]{};
    Obs2 nxt{};
    if (i < n) {
        pose_camera(V.states + pb * 10, V.intr + pb * 4, pc);
        const int* ptr = V.pose_ptr + 2 * ob;
        beg = ptr[i];
        end = ptr[i + 1];
        if (PAIR) {
            if (beg + 2 * sub < end) nxt = load2(beg + 2 * sub);
        } else {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d)
                if (beg + sub + d * G < end) ring[d] = load(beg + sub + d * G);
        }
    }

    // Phase 2: the median (every block of the window finishes the select itself, see select_finish)
    RobustParams rp;
    rp.c = select_finish(V, w, sel_lh, sel_u, sel_keys);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc.c_obs = rp.c;      // for the record (vba_debug_fetch)
    rp.inv_c = 1.0 / rp.c;
    rp.inv_c2 = 1.0 / (rp.c * rp.c);
    rp.am2 = prm.am2;
    rp.inv_am2 = 1.0 / prm.am2;
    rp.expo = prm.expo;
    rp.alpha_is_2 = prm.alpha_is_2;
    rp.expo_is_mhalf = prm.expo == -0.5;

    // Phase 3: weights and accumulation
    double wmax_l = 0.0;
    double acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
    if (i < n) {
        auto process = [&](double ox_, double oy_, double oz_, double ou_, double ov_, double oc_, int k) {
            double u, v, cam[3], d, J[12];
            project(pc, ox_, oy_, oz_, u, v, cam, d);
            project_jacobian(pc, cam, d, J);
            const double ru = ou_ - u, rv = ov_ - v;
            const double wr = robust_weight_raw(rp, ru, rv);
            V.wraw[mb + k] = wr;
            wmax_l = fmax(wmax_l, wr);
            const double wc = wr * oc_;
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const double ja = wc * J[a], jb = wc * J[6 + a];
#pragma unroll
                for (int b = a; b < 6; ++b) { acc[q] += ja * J[b] + jb * J[6 + b]; ++q; }
                acc[21 + a] += ja * ru + jb * rv;
            }
        };
        if (PAIR) {
            int k = beg + 2 * sub;
            while (k < end) {
                const int kn = k + 2 * G;
                const Obs2 cur = nxt;
                if (kn < end) nxt = load2(kn);
                process(cur.x.a, cur.y.a, cur.z.a, cur.u.a, cur.v.a, cur.c.a, k);
                if (k + 1 < end) process(cur.x.b, cur.y.b, cur.z.b, cur.u.b, cur.v.b, cur.c.b, k + 1);
                k = kn;
            }
        } else {
            int k = beg + sub;
            while (k < end) {
                const int kn = k + G;
                const Obs cur = ring[0];
#pragma unroll
                for (int d = 0; d + 1 < kAccDepth; ++d) ring[d] = ring[d + 1];
                if (k + kAccDepth * G < end) ring[kAccDepth - 1] = load(k + kAccDepth * G);
                process(cur.x, cur.y, cur.z, cur.u, cur.v, cur.c, k);
                k = kn;
            }
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] += shfl_xor_f64(acc[q], off);
    }
    if (i < n) {
        double* H = V.Hraw + pb * 21;
        double* B = V.braw + pb * 6;
#pragma unroll
        for (int q = 0; q < 21; ++q) if (q % G == sub) H[q] = acc[q];
#pragma unroll
        for (int q = 0; q < 6; ++q) if ((21 + q) % G == sub) B[q] = acc[21 + q];
    }
    wmax_l = wave_max(wmax_l);
    if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = wmax_l;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mx = fmax(fmax(wmx[0], wmx[1]), fmax(wmx[2], wmx[3]));
        atomicMax(&sc.wmax_bits, f64_bits(mx));     // positive doubles order like their bit patterns
    }
}

// ---------------------------------------------------------------------------------------------- A8: trial residuals
// blocks [0, nblk_obs): sum |w (uv - est')| over the observations (BA_filtering.py:61, 66);
// blocks [nblk_obs, nblk_obs + nblk_dyn): sqrt(sigma) sum |r_pred'| over the pose edges (BA_filtering.py:65, 67).
// EMIT: the observation blocks also write the |r| keys, their exponent histogram (select digit 0, zeroed by this
// call's k_assemble) and the block sums of |r| at the trial states -- the input of the next call if this trial
// is accepted (k_decide clears the histogram again if it is not).
template <bool EMIT>
__global__ __launch_bounds__(kObsBlock) void k_trial(DevView V) {
    __shared__ double red[kObsBlock / 64];
    __shared__ unsigned lh[EMIT ? 1024 : 1];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const WinScalars& sc = V.sc[w];
    if (sc.done) return;
    const int n = V.n[w], m = V.m[w];
    const StepParams& prm = V.prm;
    double s = 0.0, s_raw = 0.0;
    const size_t sb = (size_t)w * V.n_max;
    const bool obs_block = (int)blockIdx.x < V.nblk_obs;
    if (EMIT && obs_block) {
        for (int b = threadIdx.x; b < 1024; b += kObsBlock) lh[b] = 0u;
        __syncthreads();
    }
    if (obs_block) {
        const int k = blockIdx.x * kObsBlock + threadIdx.x;
        if (k < m) {
            const size_t ob = (size_t)w * V.obs_stride, mb = (size_t)w * V.m_max;
            const int pose = V.opose[2 * ob + k];
            PoseCam pc;
            pose_camera(V.states_new + (sb + pose) * 10, V.intr + (sb + pose) * 4, pc);
            double u, v, cam[3], d;
            project(pc, V.ox[ob + k], V.oy[ob + k], V.oz[ob + k], u, v, cam, d);
            const double wk = (V.wraw[mb + k] / bits_f64(sc.wmax_bits)) * V.oconf[ob + k];
            const double du = V.ou[ob + k] - u, dv = V.ov[ob + k] - v;
            s = fabs(du * wk) + fabs(dv * wk);
            if (EMIT) {
                const double ru = fabs(du), rv = fabs(dv);
                reinterpret_cast<double2*>(V.absr + 2 * mb)[k] = make_double2(ru, rv);
                s_raw = ru + rv;
                atomicAdd(&lh[(unsigned)(f64_bits(ru) >> 53) & 1023u], 1u);
                atomicAdd(&lh[(unsigned)(f64_bits(rv) >> 53) & 1023u], 1u);
            }
        }
    } else {
        const int db = blockIdx.x - V.nblk_obs;
        const int i = db * kObsBlock + threadIdx.x;
        const bool reg = V.reg && !prm.initialize;
        if (!prm.initialize && i < n - 1) {
            const double* st = V.states_new + (sb + i) * 10;
            const double* sn = st + 10;
            double x[6] = {st[0], st[1], st[2], st[7], st[8], st[9]};
            const int steps = V.steps[sb + i];
            propagate_gap<false>(x, nullptr, steps, V.hop);
            s = fabs(x[0] - sn[0]) + fabs(x[1] - sn[1]) + fabs(x[2] - sn[2]) +
                fabs((x[3] - sn[7]) * kVelCoeff) + fabs((x[4] - sn[8]) * kVelCoeff) + fabs((x[5] - sn[9]) * kVelCoeff);
            double att = fabs(attitude_residual(st + 3, V.cumrot + (sb + i) * 4, sn + 3));
            // BA_reg evaluates the trial's dynamics residual with quat_coeff_prior = 1 where BA passes quat_coeff = 100
            // (BA_filtering.py:172, 174 vs :63, 65): reproduced as written
            if (reg) att *= 1.0 / kQuatCoeff;
            s += att;
            s *= prm.sqrt_sigma;
        }
        if (reg && i < n) {     // sum |r_prior| at the trial states (BA_filtering.py:175, 178), not scaled by sigma
            double r6[6];
            prior_residual(V.prior_H + (sb + i) * 36, V.prior_x + (sb + i) * 6, V.states_new + (sb + i) * 10, r6);
            s += fabs(r6[0]) + fabs(r6[1]) + fabs(r6[2]) + fabs(r6[3]) + fabs(r6[4]) + fabs(r6[5]);
        }
    }
    const double t = block_sum<kObsBlock>(s, red);
    if (threadIdx.x == 0) V.part_trial[(size_t)w * (V.nblk_obs + V.nblk_dyn) + blockIdx.x] = t;
    if (EMIT && obs_block) {
        const double t_raw = block_sum<kObsBlock>(s_raw, red);
        if (threadIdx.x == 0) V.part_next[(size_t)w * V.nblk_obs + blockIdx.x] = t_raw;
        unsigned* hist = V.hist + (size_t)w * kSelPasses * kSelBins;
        for (int b = threadIdx.x; b < 1024; b += kObsBlock) {
            const unsigned c = lh[b];
            if (c) atomicAdd(&hist[b], c);
        }
    }
}

// ---------------------------------------------------------------------------------------------- debug
__global__ __launch_bounds__(kObsBlock) void k_debug_project(DevView V, int w, double* est, double* J, double* wt) {
    const int m = V.m[w];
    const int k = blockIdx.x * kObsBlock + threadIdx.x;
    if (k >= m) return;
    const size_t ob = (size_t)w * V.obs_stride, mb = (size_t)w * V.m_max;
    const int pose = V.opose[2 * ob + k];
    const size_t pb = (size_t)w * V.n_max + pose;
    PoseCam pc;
    pose_camera(V.states_prev + pb * 10, V.intr + pb * 4, pc);
    double u, v, cam[3], d;
    project(pc, V.ox[ob + k], V.oy[ob + k], V.oz[ob + k], u, v, cam, d);
    est[2 * k] = u;
    est[2 * k + 1] = v;
    project_jacobian(pc, cam, d, J + 12 * (size_t)k);
    wt[k] = (V.wraw[mb + k] / bits_f64(V.sc[w].wmax_bits)) * V.oconf[ob + k];
}

// window 0's states and damping copied to every other window (vba_set_states with window == -1)
__global__ __launch_bounds__(256) void k_broadcast_states(DevView V, int n, double lamda) {
    const int w = blockIdx.y;
    const double* src = V.states;
    double* dst = V.states + (size_t)w * V.n_max * 10;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (w > 0 && k < n * 10) dst[k] = src[k];
    if (k == 0) V.sc[w].lamda = lamda;
}

// pose / row counts of a freshly uploaded window (m < 0: keep)
__global__ void k_set_counts(int* n_arr, int* m_arr, int w, int n, int m) {
    n_arr[w] = n;
    if (m >= 0) m_arr[w] = m;
}

__global__ void k_reset_calls(DevView V) {
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w < V.W) V.sc[w].call_idx = 0;
}

// exponent histogram left behind by a k_trial<true> whose states were replaced before anybody used it
__global__ __launch_bounds__(1024) void k_clear_hist0(DevView V) {
    V.hist[(size_t)blockIdx.x * kSelPasses * kSelBins + threadIdx.x] = 0u;
}

void launch_clear_hist0(const DevView& V, hipStream_t s) {
    hipLaunchKernelGGL(k_clear_hist0, dim3(V.W), dim3(1024), 0, s, V);
}

void launch_set_counts(const DevView& V, int w, int n, int m, hipStream_t s) {
    hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(1), 0, s, const_cast<int*>(V.n), const_cast<int*>(V.m), w, n, m);
}

void launch_reset_calls(const DevView& V, hipStream_t s) {
    hipLaunchKernelGGL(k_reset_calls, dim3((V.W + 63) / 64), dim3(64), 0, s, V);
}

void launch_broadcast_states(const DevView& V, int n, double lamda, hipStream_t s) {
    hipLaunchKernelGGL(k_broadcast_states, dim3((n * 10 + 255) / 256, V.W), dim3(256), 0, s, V, n, lamda);
}

// ---------------------------------------------------------------------------------------------- launchers

void launch_obs_residual(const DevView& V, double* abs_out, hipStream_t s) {
    // the exponent histogram is fused only when the keys of this launch are the whole key set (not sharded)
    if (abs_out) hipLaunchKernelGGL(k_obs_residual<false>, dim3(V.nblk_obs, V.W), dim3(kObsBlock), 0, s, V, abs_out);
    else hipLaunchKernelGGL(k_obs_residual<true>, dim3(V.nblk_obs, V.W), dim3(kObsBlock), 0, s, V, abs_out);
}

void launch_select(const DevView& V, hipStream_t s) {
    const int64_t count = V.abs_all ? V.abs_all_count : 2 * V.m_max;
    const dim3 b(256);
    if (V.W >= 16) {
        const int nb = (int)((count + 256 * 32 - 1) / (256 * 32));
        const dim3 g(nb > 0 ? nb : 1, V.W);
        hipLaunchKernelGGL((k_select_pass<1, false, 32>), g, b, 0, s, V);
        hipLaunchKernelGGL((k_select_pass<2, true, 32>), g, b, 0, s, V);
    } else {
        const int nb = (int)((count + 256 * kSelItems - 1) / (256 * kSelItems));
        const dim3 g(nb > 0 ? nb : 1, V.W);
        if (V.abs_all) hipLaunchKernelGGL((k_select_pass<0, false, kSelItems>), g, b, 0, s, V);   // sharded: digit 0 over the gathered keys
        hipLaunchKernelGGL((k_select_pass<1, false, kSelItems>), g, b, 0, s, V);
        hipLaunchKernelGGL((k_select_pass<2, true, kSelItems>), g, b, 0, s, V);
    }
}

void launch_obs_accumulate(const DevView& V, hipStream_t s) {
    const int G = V.acc_lanes;
    const int nb = (V.n_max * G + 255) / 256;
    // V.dyn_in_acc: the blocks of the dynamics factor are appended to the grid
    const dim3 g(nb + (V.dyn_in_acc ? (V.n_max * kDynLanes + 255) / 256 : 0), V.W), b(256);
#ifndef VBA_ACC_PAIR
#define VBA_ACC_PAIR 1
#endif
    constexpr bool kPair = VBA_ACC_PAIR != 0;
    switch (G) {
        case 4: hipLaunchKernelGGL((k_obs_accumulate<4, kPair>), g, b, 0, s, V); break;
        case 8: hipLaunchKernelGGL((k_obs_accumulate<8, kPair>), g, b, 0, s, V); break;
        case 16: hipLaunchKernelGGL((k_obs_accumulate<16, false>), g, b, 0, s, V); break;
        case 32: hipLaunchKernelGGL((k_obs_accumulate<32, false>), g, b, 0, s, V); break;
        default: hipLaunchKernelGGL((k_obs_accumulate<64, false>), g, b, 0, s, V); break;
    }
}

void launch_trial(const DevView& V, hipStream_t s) {
    const dim3 g(V.nblk_obs + V.nblk_dyn, V.W), b(kObsBlock);
    if (V.emit) hipLaunchKernelGGL(k_trial<true>, g, b, 0, s, V);
    else hipLaunchKernelGGL(k_trial<false>, g, b, 0, s, V);
}

void launch_debug_project(const DevView& V, int w, int m, double* est, double* J, double* wt, hipStream_t s) {
    hipLaunchKernelGGL(k_debug_project, dim3((m + kObsBlock - 1) / kObsBlock), dim3(kObsBlock), 0, s, V, w, est, J, wt);
}

}  // namespace vba
