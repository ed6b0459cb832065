// vba_obs.hip -- observation-indexed kernels of the BA iteration (gfx950).
//
//   k_obs_residual     A1: reprojection residuals at the input states, |r| keys, sum |r|
//   k_select_pass<P>   A3a: exact lower median of the 2m keys by most-significant-digit radix select:
//                      digit 0 (exponent) is histogrammed inside k_obs_residual, digits 1 and 2 read the keys
//                      once each, the second of them compacting the (few) keys that match the 32 known bits,
//                      and select_finish (prologue of k_obs_accumulate) finishes digits 3..5 on that short list
//   k_select_warm      A3a on carried keys: the trial that produced the keys binned them around the median of its own
//                      call (warm_bin, vba_device.h), so ONE pass compacts the bin of the wanted rank; in a chained
//                      schedule its prologue is the accept test of the call in front (vba_decide.h)
//                      (batched handles; latency mode keeps the keys in per-bin buckets instead and selects inside the
//                      accumulation: warm_front / front_resolve / select_finish_list below)
//   k_select_finish    many windows: the select finished once per window (one block each) instead of in every
//                      accumulation block
//   k_obs_accumulate<G> A2 + A3a + A3b: Jacobian, robust weight, per-pose 6x6 / 6 accumulation (G lanes per pose); latency
//                      mode: starts the call -- inline select on the bin buckets, accept test of the call in front at its end
//   k_trial            A8: step + retraction (latency mode), weighted trial residuals and dynamics residuals at the trial
//                      states, next call's keys: histogram + bin buckets
//   k_debug_project    recompute est / Jacobian at the step's input states for vba_debug_fetch
//
// All of these stream the observation arrays once, coalesced (SoA, 8 B per lane per array); the pose state
// is gathered through L1/L2 (observations are pose sorted, so a wave touches one or two poses).
#include "vba_decide.h"
#include <cstdlib>
#include "vba_device.h"
#include "vba_dyn_body.h"
#include "vba_launch.h"
#include "vba_step.h"

namespace vba {

// Per-call state that must be clean before the first kernel touches it:
//   * digit-0 histogram of the call's parity: zeroed by k_obs_accumulate / k_select_finish of the call that consumed it last
//     -- with the inline select by that call's k_trial, the accumulation's blocks are still reading it -- (and by the
//     allocation); digits 1, 2: zeroed by k_trial;
//   * scalars (done, n_trials, flags, max weight): reset by thread 0 of block 0 of the call's first kernel
//     (k_obs_residual, k_select_warm, k_select_pass<1> of a repeated select, or -- inline select -- k_obs_accumulate at its
//     end, except the max weight: per parity, cleared by the previous call's k_trial); no other block of that kernel reads them;
//   * the length of the compacted list: reset by the kernel in FRONT of the one that appends (k_obs_residual /
//     k_select_pass<1>, or the previous call's k_trial for k_select_warm).
// keep_wmax: the caller is a block of the accumulation itself (inline select): other blocks of the same kernel may
// already have entered their maximum, the slot was cleared by the previous call's trial kernel instead
__device__ __forceinline__ void begin_call_scalars(WinScalars& sc, int par, bool keep_wmax = false) {
    sc.done = 0;
    sc.n_trials = 0;
    sc.fl[par] = 0u;
    if (!keep_wmax) sc.wmax_bits[par] = 0ull;
    sc.sum_abs_rpred = 0.0;
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
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        begin_call_scalars(V.sc[w], V.par);
        V.sc[w].sel_cnt = 0u;
    }
    if (HIST0) {
        unsigned* hist = hist0_of(V, w, V.par);
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
    __shared__ double red[kObsBlock / 64];
    const int w = blockIdx.y;
    VBA_SKIP_CALL(V, w);
    const double* keys = V.abs_all ? V.abs_all : V.absr + 2 * (size_t)w * V.m_max;
    const int64_t count = V.abs_all ? V.abs_all_count : 2 * (int64_t)V.m[w];
    // carried keys (a select repeated with the exact digits): k_obs_residual did not run, this kernel owns the resets
    if (P == 1 && V.carry && blockIdx.x == 0 && threadIdx.x == 0) {
        begin_call_scalars(V.sc[w], V.par);
        V.sc[w].sel_cnt = 0u;
    }
    // sum |r_obs| at the input states for the accept test: fixed-order sum of k_obs_residual's block partials
    // (carried keys bring it along; sharded mode gets the sum over all ranks from k_shard_reduce)
    if (P == 1 && !V.carry && V.m_total == 0 && blockIdx.x == 0) {
        const double* pi = V.part_init + (size_t)w * V.nblk_obs;
        double s_init = 0.0;
        for (int b = threadIdx.x; b < V.nblk_obs; b += 256) s_init += pi[b];
        const double tot = block_sum<256>(s_init, red);
        if (threadIdx.x == 0) V.sc[w].sum_in[V.par] = tot;
    }
    if ((int64_t)blockIdx.x * 256 * ITEMS >= count) return;
    auto digit_hist = [&](int d) { return d == 0 ? hist0_of(V, w, V.par) : histd_of(V, w, d); };
    constexpr int nbins = 1 << sel_width(P);
    for (int b = threadIdx.x; b < nbins; b += 256) lh[b] = 0u;
    // few keys per thread (single window, latency matters): their loads are issued before the histogram of the
    // previous digit is resolved, not after
    // keys are read two at a time (16 bytes per lane: 8-byte accesses stream at little more than half that rate); the
    // number of keys is even (two per observation row)
    constexpr bool PRELOAD = ITEMS <= 8;
    constexpr int PAIRS = ITEMS / 2;
    const double2* keys2 = reinterpret_cast<const double2*>(keys);
    const int64_t npair = count / 2;
    double2 pk[PRELOAD ? PAIRS : 1];
    if (PRELOAD) {
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) {
            const int64_t idx = ((int64_t)blockIdx.x * PAIRS + it) * 256 + threadIdx.x;
            pk[it] = idx < npair ? keys2[idx] : make_double2(0.0, 0.0);
        }
    }
    unsigned long long prefix = 0ull;
    // torch.median = lower median (BA_filtering.py:23); in sharded mode the gathered buffer may end in +inf padding
    long long rank = ((V.m_total ? 2 * V.m_total : count) - 1) / 2;
    if (P > 0) {
        constexpr int Q = P > 0 ? P - 1 : 0;
        select_resolve(digit_hist(Q), 1 << sel_width(Q), sel_width(Q), V.sc[w].sel_prefix[Q], V.sc[w].sel_rank[Q],
                       prefix, rank, lds_u);
    } else {
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        V.sc[w].sel_prefix[P] = prefix;
        V.sc[w].sel_rank[P] = rank;
        if (COMPACT) V.sc[w].sel_mode = 0;
    }
    auto take = [&](unsigned long long key, bool have) {
        bool match = have;
        if (P > 0) match = have && (key >> sel_shift(P > 0 ? P - 1 : 0)) == prefix;
        if (match) atomicAdd(&lh[(unsigned)(key >> sel_shift(P)) & (nbins - 1)], 1u);
        if (COMPACT) {
            // wave-aggregated append: one atomic per wave instruction
            const unsigned long long mask = __ballot(match);
            if (mask) {
                const int lane = threadIdx.x & 63;
                const int leader = __ffsll((long long)mask) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&V.sc[w].sel_cnt, (unsigned)__popcll(mask));
                base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);      // (leader is uniform: one v_readlane, not a crossbar shuffle)
                if (match) {
                    const unsigned off = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                    // the list has room for 2 m_max keys; if more match (massive ties) select_finish sees
                    // sel_cnt > capacity and rescans the full key array instead
                    if ((int64_t)base + off < 2 * V.m_max) V.ckeys[2 * (size_t)w * V.m_max + base + off] = bits_f64(key);
                }
            }
        }
    };
#pragma unroll 8
    for (int it = 0; it < PAIRS; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * PAIRS + it) * 256 + threadIdx.x;
        const bool have = idx < npair;
        const double2 kk = PRELOAD ? pk[it] : (have ? keys2[idx] : make_double2(0.0, 0.0));
        take(f64_bits(kk.x), have);
        take(f64_bits(kk.y), have);
    }
    __syncthreads();
    unsigned* hist_out = digit_hist(P);
    for (int b = threadIdx.x; b < nbins; b += 256) {
        const unsigned c = lh[b];
        if (c) atomicAdd(&hist_out[b], c);
    }
}

// ---------------------------------------------------------------------------------------------- A3a: warm select
// One pass over carried keys.  Prologue (chained schedule, V.fold): the accept test of the call in front, evaluated by
// every block (vba_decide.h); a first trial that was cleanly accepted lets the window move on to this call at once.
// Then every block resolves the warm histogram the trial left behind (bin of the wanted rank), and the keys of that
// bin are compacted for select_finish.  A rank outside the binned range, or a bin longer than the list may be, is a
// miss: the window waits (sc.miss) until the host has repeated this call's select with the exact digits.
// The front of a call on carried keys, evaluated by EVERY block of the kernel that starts the call -- k_select_warm, or,
// with the bin buckets (inline select), k_obs_accumulate itself.  fold_here: the accept test of the call in front; a
// first trial that was not cleanly accepted leaves everything untouched (kWarmSkip).  Then the warm histogram the trial
// left behind is resolved to the bin of the wanted rank; block 0 records the outcome.  A rank outside the binned range,
// or a bin longer than `list_cap`, is a miss.
enum { kWarmSkip = 0, kWarmHit = 1, kWarmMiss = 2 };

// the accept test of the call in front (all threads); clean = its first trial was accepted with nothing to repair
__device__ __forceinline__ bool fold_decide(const DevView& V, int w, double (*red)[4], DecideOut& d) {
    d = decide_eval(V, w, V.par ^ 1, V.prev, 0, 0.0, nullptr, 0, red);
    return d.accept && !(d.flags & (2u | 8u | 32u));
}
// ... with its inputs loaded earlier (fold_load)
__device__ __forceinline__ DecideIn fold_load(const DevView& V, int w) { return decide_load(V, w, V.par ^ 1, V.prev, 0); }
__device__ __forceinline__ bool fold_decide_loaded(const DevView& V, int w, const DecideIn& in, double (*red)[4], DecideOut& d) {
    d = decide_finish(V, w, in, V.prev, 0, 0.0, nullptr, 0, red);
    return d.accept && !(d.flags & (2u | 8u | 32u));
}
// ... and what a clean one leaves behind (block 0 only): the window moves on to this call
__device__ __forceinline__ void fold_commit(const DevView& V, int w, const DecideOut& d) {
    WinScalars& sc = V.sc[w];
    const int t = threadIdx.x, par = V.par;
    const double lam32 = sc.lam32;      // of the decided call's solve
    if (t < 81) {
        const double hv = V.lastD[(size_t)w * 81 + t] + ((t / 9 == t % 9) ? lam32 : 0.0);
        sc.last_hessian[t] = hv;
        if (V.host_states) V.host_head[w].last_hessian[t] = hv;     // (a pipelined call reads its result from host memory)
    }
    if (t == 0) {
        sc.lam[par] = d.lam_out;
        sc.sum_in[par] = d.sum_next;
        sc.init_residual = d.init_residual;
        sc.trial_residual = d.residual;
        sc.call_idx = V.call;
        WinHead& hh = V.host_head[w];
        hh.lamda = d.lam_out;
        hh.trial_residual = d.residual;
        hh.n_trials = 1;
        hh.flags = d.flags;
        hh.done = 1;
        hh.call_idx = V.call;
    }
}
// the scalars of the call that begins (one thread): the selected bin, or the miss
__device__ __forceinline__ void front_commit(const DevView& V, int w, bool hit, unsigned bin, long long rank, unsigned in_bin, bool inline_select) {
    WinScalars& sc = V.sc[w];
    const int par = V.par;
    begin_call_scalars(sc, par, inline_select);
    if (hit) {
        sc.sel_mode = 1;
        sc.sel_rank[2] = rank;
        sc.warm_base = sc.warm_lo[par] + ((unsigned long long)(bin - 1u) << V.warm_shift);
        if (inline_select) sc.sel_cnt = in_bin;
    } else {
        sc.miss = 1;
        sc.fl[par] = 32u;
        V.host_head[w].flags = 32u;
        V.host_head[w].done = 0;
    }
}
// the warm histogram resolved to the bin of the wanted rank (all threads; hloc: select_load of hist0[par])
__device__ __forceinline__ bool front_resolve(const DevView& V, int w, const unsigned (&hloc)[8], long long list_cap, unsigned* lds_u,
                                              unsigned& bin, long long& rank, unsigned& in_bin) {
    const int64_t count = 2 * (int64_t)V.m[w];
    const unsigned long long lo = V.sc[w].warm_lo[V.par];
    unsigned long long prefix;
    select_resolve_loaded(hloc, kSelBins, 11, 0ull, (count - 1) / 2, prefix, rank, lds_u, &in_bin);
    bin = (unsigned)prefix;
    return lo != ~0ull && bin >= 1u && bin <= 2046u && (int64_t)in_bin <= list_cap && !V.warm_force_miss;
}

// in order: accept test, then this call's select (k_select_warm; the accumulation when something of the call in front
// could still read what it is about to overwrite)
__device__ __forceinline__ int warm_front(const DevView& V, int w, bool fold_here, bool inline_select, long long list_cap,
                                          double (*red)[4], unsigned* lds_u, unsigned& bin_out, long long& rank_out,
                                          unsigned& in_bin_out) {
    unsigned hloc[8];
    select_load(hist0_of(V, w, V.par), kSelBins, hloc);        // in flight while the accept test is evaluated
    if (fold_here) {
        DecideOut d;
        if (!fold_decide(V, w, red, d)) return kWarmSkip;      // not a clean first trial: the host finishes that call
        if (blockIdx.x == 0) fold_commit(V, w, d);
    }
    const bool hit = front_resolve(V, w, hloc, list_cap, lds_u, bin_out, rank_out, in_bin_out);
    if (blockIdx.x == 0 && threadIdx.x == 0) front_commit(V, w, hit, bin_out, rank_out, in_bin_out, inline_select);
    return hit ? kWarmHit : kWarmMiss;
}

template <int ITEMS>
__global__ __launch_bounds__(256) void k_select_warm(DevView V) {
    __shared__ unsigned lds_u[260];
    __shared__ double red[5][4];
    const int w = blockIdx.y;
    WinScalars& sc = V.sc[w];
    const int t = threadIdx.x;
    const bool fold_here = V.call >= 0 && V.fold && sc.pending == V.call - 1 && sc.call_idx == V.call - 1;
    if (!fold_here) VBA_SKIP_CALL(V, w);
    const double* keys = V.absr + 2 * (size_t)w * V.m_max;
    const int64_t count = 2 * (int64_t)V.m[w];
    if ((int64_t)blockIdx.x * 256 * ITEMS >= count) return;         // (never block 0)
    // the keys of a short block are loaded before anything is decided (latency)
    constexpr bool PRELOAD = ITEMS <= 8;
    constexpr int PAIRS = ITEMS / 2;        // two keys (16 bytes) per load
    const double2* keys2 = reinterpret_cast<const double2*>(keys);
    const int64_t npair = count / 2;
    // (both forms request their keys before the histogram is resolved: the resolve is a dependent round trip plus a scan)
    double2 pk[PAIRS];
#pragma unroll
    for (int it = 0; it < PAIRS; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * PAIRS + it) * 256 + t;
        pk[it] = idx < npair ? keys2[idx] : make_double2(0.0, 0.0);
    }
    const unsigned long long lo = sc.warm_lo[V.par];
    unsigned bin, in_bin;
    long long rank;
    if (warm_front(V, w, fold_here, false, 2 * V.m_max, red, lds_u, bin, rank, in_bin) != kWarmHit) return;
    if constexpr (!PRELOAD) {
        // Many windows per launch, coarse warm bins (1/8 binade: a few per cent of the keys match).  A returning atomic per
        // wave instruction would be a chain of ITEMS dependent round trips; instead the block counts its matches first,
        // reserves its share of the list with ONE atomic and then writes.  The keys stay in registers in between.
        double2 (&kk)[PAIRS] = pk;
        unsigned long long mbits = 0ull;        // bit 2 it: kk[it].x matches, bit 2 it + 1: kk[it].y
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) {
            const int64_t idx = ((int64_t)blockIdx.x * PAIRS + it) * 256 + t;
            const bool have = idx < npair;
            if (have && warm_bin(f64_bits(kk[it].x), lo, V.warm_shift) == bin) mbits |= 1ull << (2 * it);
            if (have && warm_bin(f64_bits(kk[it].y), lo, V.warm_shift) == bin) mbits |= 2ull << (2 * it);
        }
        const unsigned mine = (unsigned)__popcll(mbits);
        const unsigned inc = wave_inclusive_scan_u32(mine);
        __syncthreads();            // lds_u was read by the resolve above
        if ((t & 63) == 63) lds_u[t >> 6] = inc;
        __syncthreads();
        if (t == 0) {
            const unsigned total = lds_u[0] + lds_u[1] + lds_u[2] + lds_u[3];
            lds_u[4] = total ? atomicAdd(&sc.sel_cnt, total) : 0u;
        }
        __syncthreads();
        unsigned at = lds_u[4] + inc - mine;
        for (int q = 0; q < (t >> 6); ++q) at += lds_u[q];
        double* list = V.ckeys + 2 * (size_t)w * V.m_max;
#pragma unroll
        for (int it = 0; it < PAIRS; ++it) {
            if (mbits & (1ull << (2 * it))) list[at++] = kk[it].x;
            if (mbits & (2ull << (2 * it))) list[at++] = kk[it].y;
        }
        return;
    }
    auto take = [&](unsigned long long key, bool have) {
        const bool match = have && warm_bin(key, lo, V.warm_shift) == bin;
        const unsigned long long mask = __ballot(match);
        if (mask) {     // wave-aggregated append: one atomic per wave instruction
            const int lane = t & 63;
            const int leader = __ffsll((long long)mask) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&sc.sel_cnt, (unsigned)__popcll(mask));
            base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);      // (leader is uniform: one v_readlane, not a crossbar shuffle)
            if (match) V.ckeys[2 * (size_t)w * V.m_max + base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = bits_f64(key);
        }
    };
    if constexpr (PRELOAD) {
#pragma unroll 8
        for (int it = 0; it < PAIRS; ++it) {
            const int64_t idx = ((int64_t)blockIdx.x * PAIRS + it) * 256 + t;
            const bool have = idx < npair;
            take(f64_bits(pk[it].x), have);
            take(f64_bits(pk[it].y), have);
        }
    }
}

// Finishes the select on the compacted list (keys whose top 32 bits are known to match): returns the lower median
// c_obs to every thread of the (256-thread) block.  It is the prologue of k_obs_accumulate -- every block of a window
// redoes it (a handful of keys: rank by counting) instead of one more single-block kernel on the critical path; long
// lists (massive ties) take digits 3, 4, 5 with a block-local histogram each, the full key array if the list
// overflowed.  A list made by k_select_warm (one warm bin) is ranked by counting while short, by radix digits of the offset
// inside the bin otherwise.
// ck: the list (capacity `cap` entries, cnt of them valid -- cnt > cap: it overflowed), want: the rank wanted among them,
// mode 0: keys sharing the 32-bit prefix of an exact select, 1: the keys of one warm bin starting at warm_base.
// speculate: load the first 1024 entries before cnt is known to the caller's satisfaction (one round trip instead of two).
__device__ __forceinline__ double select_finish_list(const DevView& V, int w, const double* ck, int64_t cap, unsigned cnt, long long want,
                                                     int mode, unsigned long long warm_base, bool speculate, unsigned* lh /*[kSelBins]*/,
                                                     unsigned* lds_u /*[260]*/, unsigned long long* skeys /*[1024] + 1*/) {
    const WinScalars& sc = V.sc[w];
    // latency mode loads the first 1024 entries speculatively together with the length (one round trip instead of two);
    // with many windows per launch every block of every window would drag 8 KB through the caches for a handful of keys
    unsigned long long pre[4];
    if (V.sel_nslots > 0 && ck == V.sel_slots) {
        // sharded mode: the list is the concatenation of the ranks' buckets of one warm bin, slot r = [count_r, keys ...];
        // entry q of the list lives in the slot whose running count covers it (cnt <= 1024 is guaranteed by the front)
        unsigned lo_q = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[j] = 0ull;
        for (int r = 0; r < V.sel_nslots; ++r) {
            const double* slot = ck + (size_t)r * V.sel_slot_stride;
            const unsigned c_r = (unsigned)slot[0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned q = threadIdx.x + 256u * j;
                if (q >= lo_q && q < lo_q + c_r && q < cnt) pre[j] = f64_bits(slot[1 + (q - lo_q)]);
            }
            lo_q += c_r;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned q = threadIdx.x + 256u * j;
            pre[j] = ((int64_t)q < cap && (speculate || q < cnt)) ? f64_bits(ck[q]) : 0ull;
        }
    }
    if (mode == 1 && cnt <= 1024u && V.warm_shift >= 8) {
        // One warm bin: every key is warm_base + rel, rel < 2^warm_shift.  Ranking ~130 keys by counting is a serial loop of
        // ~130 LDS reads per thread on the critical path of every call; instead the top 8 bits of rel split the list over 256
        // sub-bins (one LDS atomic per key, one sub-bin per thread for the scan), and only the handful of keys in the sub-bin
        // of the wanted rank is ranked by counting.  Exact either way: the same key comes out.
        const int t = threadIdx.x;
        const int sh = V.warm_shift - 8;
        lh[t] = 0u;
        if (t == 0) { lds_u[16] = 0u; lds_u[17] = 0u; lds_u[18] = 0u; }
        __syncthreads();
        unsigned sb4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned q = t + 256u * j;
            sb4[j] = (unsigned)((pre[j] - warm_base) >> sh) & 255u;
            if (q < cnt) atomicAdd(&lh[sb4[j]], 1u);
        }
        __syncthreads();
        const unsigned c = lh[t];
        const unsigned inc = wave_inclusive_scan_u32(c);
        if ((t & 63) == 63) lds_u[t >> 6] = inc;
        __syncthreads();
        unsigned base = 0;
        for (int q = 0; q < (t >> 6); ++q) base += lds_u[q];
        const long long excl = (long long)base + inc - c;
        if (want >= excl && want < excl + (long long)c) { lds_u[16] = (unsigned)t; lds_u[17] = (unsigned)(want - excl); }
        __syncthreads();
        const unsigned tb = lds_u[16], r = lds_u[17];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned q = t + 256u * j;
            if (q < cnt && sb4[j] == tb) {
                const unsigned slot = atomicAdd(&lds_u[18], 1u);
                skeys[slot] = pre[j];       // (the order inside the list does not matter: equal keys are the same value)
            }
        }
        __syncthreads();
        const unsigned k = lds_u[18];
        for (unsigned q = t; q < k; q += 256) {
            const unsigned long long key = skeys[q];
            unsigned below = 0;
            for (unsigned j = 0; j < k; ++j) {
                const unsigned long long o = skeys[j];
                below += (o < key) || (o == key && j < q);
            }
            if (below == r) skeys[1024] = key;
        }
        __syncthreads();
        return bits_f64(skeys[1024]);
    }
    if (cnt <= (mode ? (unsigned)kWarmCount : 1024u)) {
        // the wanted key is the one of rank `want` among the list -- rank each key by counting (ties broken by position)
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
    if (mode == 1) {
        // a long warm bin: every key is warm_base + rel with rel < 2^warm_shift; radix digits of rel from the top, 11 bits at
        // a time, with a block-local histogram each
        unsigned long long prefix = 0ull;
        long long rank = want;
        int remaining = V.warm_shift;
        if (remaining > 11) {
            // first digit, then -- the usual case: thousands of keys spread over 2048 sub-bins -- the handful of keys of the
            // wanted sub-bin is gathered in LDS and ranked by counting: two passes over the list instead of one per digit
            remaining -= 11;
            for (int b = threadIdx.x; b < kSelBins; b += 256) lh[b] = 0u;
            if (threadIdx.x == 0) lds_u[18] = 0u;
            __syncthreads();
            for (unsigned q = threadIdx.x; q < cnt; q += 256) {
                const unsigned long long rel = f64_bits(ck[q]) - warm_base;
                atomicAdd(&lh[(unsigned)(rel >> remaining) & 2047u], 1u);
            }
            __syncthreads();
            unsigned sub_cnt;
            {
                unsigned loc[8];
                select_load(lh, kSelBins, loc);
                select_resolve_loaded(loc, kSelBins, 11, 0ull, rank, prefix, rank, lds_u, &sub_cnt);
            }
            if (sub_cnt <= 1024u) {
                for (unsigned q = threadIdx.x; q < cnt; q += 256) {
                    const unsigned long long key = f64_bits(ck[q]);
                    if (((key - warm_base) >> remaining) == prefix) skeys[atomicAdd(&lds_u[18], 1u)] = key;
                }
                __syncthreads();
                for (unsigned q = threadIdx.x; q < sub_cnt; q += 256) {
                    const unsigned long long key = skeys[q];
                    long long below = 0;
                    for (unsigned j = 0; j < sub_cnt; ++j) {
                        const unsigned long long o = skeys[j];
                        below += (o < key) || (o == key && j < q);
                    }
                    if (below == rank) skeys[1024] = key;
                }
                __syncthreads();
                return bits_f64(skeys[1024]);
            }
        }
        while (remaining > 0) {
            const int width = remaining < 11 ? remaining : 11;
            remaining -= width;
            const int nbins = 1 << width;
            for (int b = threadIdx.x; b < kSelBins; b += 256) lh[b] = 0u;
            __syncthreads();
            for (unsigned q = threadIdx.x; q < cnt; q += 256) {
                const unsigned long long rel = f64_bits(ck[q]) - warm_base;
                if ((rel >> (remaining + width)) == prefix) atomicAdd(&lh[(unsigned)(rel >> remaining) & (nbins - 1)], 1u);
            }
            __syncthreads();
            unsigned long long np;
            long long nr;
            select_resolve(lh, nbins, width, prefix, rank, np, nr, lds_u);
            prefix = np;
            rank = nr;
        }
        return bits_f64(warm_base + prefix);
    }
    if ((int64_t)cnt > 2 * V.m_max) {       // list overflowed: fall back to the full key array
        ck = V.abs_all ? V.abs_all : V.absr + 2 * (size_t)w * V.m_max;
        cnt = (unsigned)(V.abs_all ? V.abs_all_count : 2 * (int64_t)V.m[w]);
    }
    unsigned long long prefix;
    long long rank;
    select_resolve(histd_of(V, w, 2), 1 << sel_width(2), sel_width(2), sc.sel_prefix[2], sc.sel_rank[2], prefix, rank, lds_u);
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

// The list the select kernels of this call left (k_select_pass<2> / k_select_warm): its length, the wanted rank and the
// first entries are loaded together.
__device__ __forceinline__ double select_finish(const DevView& V, int w, unsigned* lh /*[kSelBins]*/, unsigned* lds_u /*[260]*/,
                                                unsigned long long* skeys /*[1024] + 1*/) {
    const WinScalars& sc = V.sc[w];
    if (V.sel_nslots > 0)       // sharded mode, carried keys: the ranks' buckets of the median's bin as gathered
        return select_finish_list(V, w, V.sel_slots, 1024, sc.sel_cnt, sc.sel_rank[2], 1, sc.warm_base, false, lh, lds_u, skeys);
    return select_finish_list(V, w, V.ckeys + 2 * (size_t)w * V.m_max, 2 * V.m_max, sc.sel_cnt, sc.sel_rank[2], sc.sel_mode, sc.warm_base,
                              V.lat != 0, lh, lds_u, skeys);
}

// Many windows per launch: the select is finished ONCE per window by a launch of its own (one block per window, ~20 us for
// 4096 windows) instead of by every accumulation block in its prologue -- there the two dependent round trips and the
// barriers of the finish were a third of a block's life at two blocks per CU, with nothing to overlap them.
__global__ __launch_bounds__(256) void k_select_finish(DevView V) {
    __shared__ unsigned sel_lh[kSelBins];
    __shared__ unsigned sel_u[260];
    __shared__ unsigned long long sel_keys[1025];
    const int w = blockIdx.x;
    VBA_SKIP_CALL(V, w);
    const double c = select_finish(V, w, sel_lh, sel_u, sel_keys);
    if (threadIdx.x == 0) V.sc[w].c_obs = c;
    unsigned* h0 = hist0_of(V, w, V.par);       // (see k_obs_accumulate: clean for the call after next)
    for (int b = threadIdx.x; b < kSelBins; b += 256) h0[b] = 0u;
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
// BATCH: the variant of handles with many windows -- the median is in sc.c_obs already (k_select_finish), nothing rides in
// the grid and nothing is selected inline, so none of that code (nor its registers: the rider alone needs ~195) is compiled in.
// Diagnostic builds (-DVBA_RESIDENT_STAMPS; tools/trial_stamps.py): 100 MHz wall-clock stamps of thread 0 of observation
// block 100 along k_trial, fetched with vba_debug_fetch(h, 0, 102, ...).
#ifdef VBA_RESIDENT_STAMPS
__device__ unsigned long long g_ostamps[64];
#define VBA_OSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x == 100 && blockIdx.y == 0) g_ostamps[slot] = wall_clock64(); } while (0)
#define VBA_ASTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x == 60 && blockIdx.y == 0) g_ostamps[16 + (slot)] = wall_clock64(); } while (0)
void fetch_ostamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ostamps), sizeof(g_ostamps)); }
#else
#define VBA_OSTAMP(slot) do {} while (0)
#define VBA_ASTAMP(slot) do {} while (0)
#endif
// The steps of the recursive-halving reduction of k_obs_accumulate (see there), unrolled over a compile-time mask so that the
// first two exchanges (24 of the 31 values that travel) are quad permutations instead of LDS-crossbar shuffles.
template <int G, int CNT, int MASK>
__device__ __forceinline__ void halving_steps(double (&acc)[32], int sub, int& own) {
    if constexpr (MASK < G && CNT > 1) {
        const bool up = (sub & MASK) != 0;
        constexpr int half = CNT >> 1;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const double lo = acc[j], hi = acc[half + j];
            acc[j] = (up ? hi : lo) + shfl_xor_f64_c<MASK>(up ? lo : hi);
        }
        if (up) own += half;
        halving_steps<G, (CNT >> 1), (MASK << 1)>(acc, sub, own);
    }
}

template <int G, bool PAIR, bool BATCH>
__global__ __launch_bounds__(256) void k_obs_accumulate(DevView V) {
    __shared__ double wmx[4];
    __shared__ unsigned sel_lh[BATCH ? 1 : kSelBins];
    __shared__ unsigned sel_u[BATCH ? 1 : 260];
    __shared__ unsigned long long sel_keys[BATCH ? 1 : 1025];
    __shared__ double dec_red[BATCH ? 1 : 5][4];
    constexpr int PPB = 256 / G;            // poses per block
    const int w = blockIdx.y;
    WinScalars& sc = V.sc[w];
    if (BATCH) { V.sel_inline = 0; V.dyn_in_acc = 0; V.median_ready = 1; }
    // The row range of this thread's pose, requested FIRST of all (its address needs the block and thread index only; the
    // index is clamped into the window's n_max + 1 entries): the range is a dependent round trip in front of the first
    // observation loads, and this way it runs beside the call snapshot's instead of behind it.
    int early_beg = 0, early_end = 0;
    if (!BATCH) {
        const int ie = min((int)(blockIdx.x * PPB + threadIdx.x / G), V.n_max - 1);
        const int* ptr0 = V.pose_ptr + 2 * (size_t)w * V.obs_stride;
        early_beg = ptr0[ie];
        early_end = ptr0[ie + 1];
    }
    // Inline select (V.sel_inline: latency mode, carried keys in bin buckets): this kernel STARTS the call -- no select
    // kernel in front of it.  In a chained schedule its blocks evaluate the accept test of the call in front themselves
    // (warm_front) and go on only if that first trial was cleanly accepted.
    // One relaxed atomic read of the two words, once per block.  The extra block of THIS grid (below) commits the call in front
    // and writes call_idx = V.call while other blocks may not have started yet -- an intra-grid race that is benign because
    // both outcomes let a block proceed: a block that still sees (pending, call_idx) = (call - 1, call - 1) takes fold_here,
    // one that already sees call_idx = V.call passes the ordinary "window is at this call" test; no other value can be seen
    // (the commit block is the only writer during this kernel, and it writes only after a clean accept).
    const int seen_call = __atomic_load_n(&sc.call_idx, __ATOMIC_RELAXED);
    const int seen_pending = __atomic_load_n(&sc.pending, __ATOMIC_RELAXED);
    const bool fold_here = V.sel_inline && V.call >= 0 && V.fold && seen_pending == V.call - 1 && seen_call == V.call - 1;
    if (!fold_here && !((V.call < 0 || seen_call == V.call) && (V.redo == 2 || (sc.miss != 0) == (V.redo != 0)))) return;     // VBA_SKIP_CALL on the snapshot
    // The accept test only GATES: nothing this kernel computes depends on it, and a trial that turns out not to be clean just
    // leaves no trace -- what this kernel writes on the way (weights, per-pose sums, the pose-chain factor of its rider
    // blocks) lives per call parity, the later trials of the call in front still find theirs.
    constexpr bool ordered = false;     // (kept: the in-order form, accept test first, is warm_front as k_select_warm uses it)
    const int nb_acc = (V.n_max * G + 255) / 256;
    // ... and it is evaluated by ONE extra block of the grid (the last one), which also leaves what the start of this call
    // leaves in the scalars: off the critical path of the blocks that accumulate.  Those need no gate at all: their
    // maximum goes into a slot that the trial kernel of the call in front clears whenever it runs again.
    if (V.sel_inline && !ordered && blockIdx.x == gridDim.x - 1) {
        unsigned hl[8];
        select_load(hist0_of(V, w, V.par), kSelBins, hl);
        DecideIn fin = {};
        if (fold_here) fin = fold_load(V, w);
        unsigned bin, in_bin;
        long long rank;
        const bool hit = front_resolve(V, w, hl, V.bucket_cap, sel_u, bin, rank, in_bin);
        double c = 0.0;
        if (hit) {
            const double* bucket = V.wbucket + (((size_t)w * 2 + V.par) * kSelBins + bin) * (size_t)V.bucket_cap;
            c = select_finish_list(V, w, bucket, V.bucket_cap, in_bin, rank, 1, sc.warm_lo[V.par] + ((unsigned long long)(bin - 1u) << V.warm_shift),
                                   false, sel_lh, sel_u, sel_keys);
        }
        DecideOut d;
        if (fold_here && !fold_decide_loaded(V, w, fin, dec_red, d)) return;    // not clean: no trace (the window stalls at the call in front)
        if (fold_here) fold_commit(V, w, d);
        if (threadIdx.x == 0) {
            front_commit(V, w, hit, bin, rank, in_bin, true);
            if (hit) sc.c_obs = c;
        }
        return;
    }
    if (!BATCH && (int)blockIdx.x >= nb_acc) {        // few windows: the dynamics factor rides in this grid (vba_dyn_body.h)
        // (a function of the input states only: neither a missed select nor, by default, the accept test concerns it --
        // what it writes is read by this call's own assembly, which runs only if the window has moved on)
        if (fold_here && ordered) {
            DecideOut d;
            if (!fold_decide(V, w, dec_red, d)) return;
        }
        dynamics_block(V, w, blockIdx.x - nb_acc);
        return;
    }
    const int n = V.n[w];
    if (blockIdx.x * PPB >= n) return;
    VBA_ASTAMP(0);
    const StepParams& prm = V.prm;
    const int sub = threadIdx.x % G;
    const size_t ob = (size_t)w * V.obs_stride;
    const size_t mb = (size_t)w * V.m_max;
    double wmax_l = 0.0;
    // BATCH: a block walks several groups of PPB poses (grid = a quarter of the groups) and requests the row range of its
    // NEXT group while it works on the current one -- the range is a dependent round trip in front of the first
    // observation loads, and at two waves per SIMD nobody covers it.  Otherwise: one group per block, one pass.
    const int gstride = BATCH ? (int)gridDim.x : 0;
    int pf_beg = 0, pf_end = 0;
    if (BATCH) {
        const int i0 = blockIdx.x * PPB + threadIdx.x / G;
        if (i0 < n) {
            const int* ptr0 = V.pose_ptr + 2 * ob;
            pf_beg = ptr0[i0];
            pf_end = ptr0[i0 + 1];
        }
    }
    for (int grp = blockIdx.x; grp * PPB < n; grp += gstride) {
    const int i = grp * PPB + threadIdx.x / G;
    const size_t pb = (size_t)w * V.n_max + (i < n ? i : 0);

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
    // (inline select: the histogram is requested first of all -- it depends on nothing, the row range below is a dependent
    // round trip)
    unsigned hloc[8] = {};
    if (V.sel_inline && !ordered) select_load(hist0_of(V, w, V.par), kSelBins, hloc);
    PoseCam pc{};
    int beg = 0, end = 0;
    Obs ring[kAccDepth]{};
    Obs2 nxt{}, nxt2{};
#ifndef VBA_ACC_PAIR_DEPTH
#define VBA_ACC_PAIR_DEPTH 1
#endif
    constexpr bool kPairDepth2 = BATCH && PAIR && VBA_ACC_PAIR_DEPTH == 2;     // two pairs in flight per lane (24 more VGPRs)
    if (i < n) {
        pose_camera(V.states + pb * 10, V.intr + pb * 4, pc);
        if (BATCH) {
            beg = pf_beg;
            end = pf_end;
        } else {
            beg = early_beg;
            end = early_end;
        }
        if (PAIR) {
            if (beg + 2 * sub < end) nxt = load2(beg + 2 * sub);
            if (kPairDepth2 && beg + 2 * sub + 2 * G < end) nxt2 = load2(beg + 2 * sub + 2 * G);
        } else {
#pragma unroll
            for (int d = 0; d < kAccDepth; ++d)
                if (beg + sub + d * G < end) ring[d] = load(beg + sub + d * G);
        }
    }
    if (BATCH) {        // the row range of this thread's pose in the block's next group
        const int in = i + gstride * PPB;
        pf_beg = pf_end = 0;
        if (in < n) {
            const int* ptr = V.pose_ptr + 2 * ob;
            pf_beg = ptr[in];
            pf_end = ptr[in + 1];
        }
    }

    // Phase 2: the median (every block of the window finishes the select itself, see select_finish)
    VBA_ASTAMP(1);
    RobustParams rp;
    if (BATCH) {
        rp.c = sc.c_obs;        // k_select_finish
    } else if (V.sel_inline) {
        // the trial kernel of the call in front dropped every key into the bucket of its warm bin: resolve the histogram,
        // rank the wanted bin's bucket.  Every block does this redundantly (a few hundred keys), nothing is compacted.
        unsigned bin, in_bin;
        long long rank;
        if (ordered) {
            if (warm_front(V, w, fold_here, true, V.bucket_cap, dec_red, sel_u, bin, rank, in_bin) != kWarmHit) return;
        } else {
            if (!front_resolve(V, w, hloc, V.bucket_cap, sel_u, bin, rank, in_bin)) return;    // a miss (the last block records it)
        }
        const unsigned long long lo = sc.warm_lo[V.par];
        const double* bucket = V.wbucket + (((size_t)w * 2 + V.par) * kSelBins + bin) * (size_t)V.bucket_cap;
        rp.c = select_finish_list(V, w, bucket, V.bucket_cap, in_bin, rank, 1, lo + ((unsigned long long)(bin - 1u) << V.warm_shift), false,
                                  sel_lh, sel_u, sel_keys);
        // (the histogram is still being read by the other blocks: the trial kernel of this call clears it)
        if (ordered && blockIdx.x == 0 && threadIdx.x == 0) sc.c_obs = rp.c;
    } else if (V.median_ready) {
        rp.c = sc.c_obs;        // k_select_finish
    } else {
        rp.c = select_finish(V, w, sel_lh, sel_u, sel_keys);
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) sc.c_obs = rp.c;      // the trial kernel centres the next call's warm bins on it
            // the select of this call is over (its last reader of the digit-0 histogram was the kernel in front): clean for
            // the call after next, which shares the parity
            unsigned* h0 = hist0_of(V, w, V.par);
            for (int b = threadIdx.x; b < kSelBins; b += 256) h0[b] = 0u;
        }
    }
    VBA_ASTAMP(2);
    rp.inv_c = 1.0 / rp.c;
    rp.inv_c2 = 1.0 / (rp.c * rp.c);
    rp.am2 = prm.am2;
    rp.inv_am2 = 1.0 / prm.am2;
    rp.expo = prm.expo;
    rp.alpha_is_2 = prm.alpha_is_2;
    rp.expo_is_mhalf = prm.expo == -0.5;

    // Phase 3: weights and accumulation.
    // J_k = [ -A R^T | 2 A hat(p_c) ] with A = d uv / d p_c (four non-zeros) and R the pose's rotation, the same for every row
    // of the pose.  CAM (groups of <= 16 lanes: a lane sees a dozen rows or more): the lane sums in the CAMERA frame --
    // G_k = [ -A | 2 A hat(p_c) ], whose translation part is the sparse A itself -- and rotates its sums once at the end
    // (J^T J = T G^T G T^T, T = diag(R, I)): ~130 VALU instructions per row instead of ~215.  Few rows per lane (latency
    // mode, 32 / 64 lanes per pose): the rotation per lane would cost what it saves, J is formed per row.
    constexpr bool CAM = G <= 16;
    double acc[32];         // 21 + 6 sums, padded to a power of two for the halving reduction
#pragma unroll
    for (int q = 0; q < 32; ++q) acc[q] = 0.0;
    // CAM sums: M = sum wc A^T A (00, 02, 11, 12, 22; 01 = 0), T[j][c] = sum wc A[:,j] . Gr[:,c], Crr = sum wc Gr^T Gr (upper),
    // st = sum wc A^T r, gr = sum wc Gr^T r
    double cM[5] = {0, 0, 0, 0, 0}, cT[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, cS[3] = {0, 0, 0};
    if (i < n) {
        auto process = [&](double ox_, double oy_, double oz_, double ou_, double ov_, double oc_, int k) {
            double u, v, cam[3], d;
            project(pc, ox_, oy_, oz_, u, v, cam, d);
            const double ru = ou_ - u, rv = ov_ - v;
            const double wr = robust_weight_raw(rp, ru, rv);
            V.wraw[mb + k] = wr;
            wmax_l = fmax(wmax_l, wr);
            const double wc = wr * oc_;
            if (CAM) {
                const double live = cam[2] > kZMin ? 1.0 : 0.0;
                const double a00 = pc.fx * d, a11 = pc.fy * d;
                const double dl = d * live;
                const double a02 = -(a00 * (cam[0] * dl)), a12 = -(a11 * (cam[1] * dl));
                const double x = cam[0], y = cam[1], z = cam[2];
                // Gr = 2 A hat(p_c): rows (g0..g2) and (h0..h2)
                const double b00 = 2.0 * a00, b02 = 2.0 * a02, b11 = 2.0 * a11, b12 = 2.0 * a12;
                const double g0 = -(b02 * y), g1 = fma(b02, x, -(b00 * z)), g2 = b00 * y;
                const double h0 = fma(b11, z, -(b12 * y)), h1 = b12 * x, h2 = -(b11 * x);
                const double w00 = wc * a00, w02 = wc * a02, w11 = wc * a11, w12 = wc * a12;
                cM[0] = fma(w00, a00, cM[0]); cM[1] = fma(w00, a02, cM[1]);
                cM[2] = fma(w11, a11, cM[2]); cM[3] = fma(w11, a12, cM[3]);
                cM[4] = fma(w02, a02, fma(w12, a12, cM[4]));
                cT[0] = fma(w00, g0, cT[0]); cT[1] = fma(w00, g1, cT[1]); cT[2] = fma(w00, g2, cT[2]);
                cT[3] = fma(w11, h0, cT[3]); cT[4] = fma(w11, h1, cT[4]); cT[5] = fma(w11, h2, cT[5]);
                cT[6] = fma(w02, g0, fma(w12, h0, cT[6])); cT[7] = fma(w02, g1, fma(w12, h1, cT[7]));
                cT[8] = fma(w02, g2, fma(w12, h2, cT[8]));
                const double wg0 = wc * g0, wg1 = wc * g1, wg2 = wc * g2, wh0 = wc * h0, wh1 = wc * h1, wh2 = wc * h2;
                // rotation-rotation block straight into its place in the packed 6x6 (rows 3..5)
                acc[15] = fma(wg0, g0, fma(wh0, h0, acc[15])); acc[16] = fma(wg0, g1, fma(wh0, h1, acc[16]));
                acc[17] = fma(wg0, g2, fma(wh0, h2, acc[17])); acc[18] = fma(wg1, g1, fma(wh1, h1, acc[18]));
                acc[19] = fma(wg1, g2, fma(wh1, h2, acc[19])); acc[20] = fma(wg2, g2, fma(wh2, h2, acc[20]));
                cS[0] = fma(w00, ru, cS[0]); cS[1] = fma(w11, rv, cS[1]); cS[2] = fma(w02, ru, fma(w12, rv, cS[2]));
                acc[24] = fma(wg0, ru, fma(wh0, rv, acc[24])); acc[25] = fma(wg1, ru, fma(wh1, rv, acc[25]));
                acc[26] = fma(wg2, ru, fma(wh2, rv, acc[26]));
            } else {
                double J[12];
                project_jacobian(pc, cam, d, J);
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double ja = wc * J[a], jb = wc * J[6 + a];
#pragma unroll
                    for (int b = a; b < 6; ++b) { acc[q] = fma(ja, J[b], fma(jb, J[6 + b], acc[q])); ++q; }
                    acc[21 + a] = fma(ja, ru, fma(jb, rv, acc[21 + a]));
                }
            }
        };
        if (PAIR) {
            int k = beg + 2 * sub;
            while (k < end) {
                const int kn = k + 2 * G;
                const Obs2 cur = nxt;
                if (kPairDepth2) {
                    nxt = nxt2;
                    if (kn + 2 * G < end) nxt2 = load2(kn + 2 * G);
                } else if (kn < end) nxt = load2(kn);
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
    VBA_ASTAMP(3);
    if (CAM) {
        // the lane's camera-frame sums into the world frame (sums are linear, so before the reduction):
        //   Htt = R M R^T, Htr = -R T, bt = -R st   (Jt = -A R^T; R[c][j] = pc.R[3 c + j])
        const double* R = pc.R;
        const double M[3][3] = {{cM[0], 0.0, cM[1]}, {0.0, cM[2], cM[3]}, {cM[1], cM[3], cM[4]}};
        double Y[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) Y[a][kk] = fma(R[3 * a], M[0][kk], fma(R[3 * a + 1], M[1][kk], R[3 * a + 2] * M[2][kk]));
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = a; b < 3; ++b)
                acc[sym6(a, b)] = fma(Y[a][0], R[3 * b], fma(Y[a][1], R[3 * b + 1], Y[a][2] * R[3 * b + 2]));
#pragma unroll
            for (int c = 0; c < 3; ++c)
                acc[sym6(a, 3 + c)] = -fma(R[3 * a], cT[c], fma(R[3 * a + 1], cT[3 + c], R[3 * a + 2] * cT[6 + c]));
            acc[21 + a] = -fma(R[3 * a], cS[0], fma(R[3 * a + 1], cS[1], R[3 * a + 2] * cS[2]));
        }
    }
    // Reduction over the G lanes of the pose by recursive halving: in step s (xor mask 2^s) a lane keeps the half of its
    // values that bit s of its lane index selects and receives the partner's partial sums of that half -- 16 + 8 + 4 + 2 + 1
    // shuffles for the (padded) 32 values instead of 27 per butterfly step; afterwards every lane owns the totals of
    // 32 / min(G, 32) consecutive values.  The shape is fixed by G, so the sums are bit reproducible.
    VBA_ASTAMP(4);
    int own = 0;
    halving_steps<G, 32, 1>(acc, sub, own);
    if (G == 64) acc[0] += shfl_xor_f64_c<32>(acc[0]);
    if (i < n && sub < 32) {
        double* H = V.Hraw + pb * 21;
        double* B = V.braw + pb * 6;
        constexpr int kOwn = 32 / (G < 32 ? G : 32);
#pragma unroll
        for (int j = 0; j < kOwn; ++j) {
            const int q = own + j;
            if (q < 21) H[q] = acc[j];
            else if (q < 27) B[q - 21] = acc[j];
        }
    }
    VBA_ASTAMP(5);
    if (!BATCH) break;
    }       // groups of this block
    wmax_l = wave_max(wmax_l);
    if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = wmax_l;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mx = fmax(fmax(wmx[0], wmx[1]), fmax(wmx[2], wmx[3]));
        atomicMax(V.wmax_ext ? V.wmax_ext : &sc.wmax_bits[V.par], f64_bits(mx));     // positive doubles order like their bit patterns
    }
    VBA_ASTAMP(6);
}

// ---------------------------------------------------------------------------------------------- A8: trial residuals
// blocks [0, nblk_obs): sum |w (uv - est')| over the observations (BA_filtering.py:61, 66);
// blocks [nblk_obs, nblk_obs + nblk_dyn): sqrt(sigma) sum |r_pred'| over the pose edges (BA_filtering.py:65, 67).
// EMIT: the observation blocks also write the |r| keys, their histogram (digit-0 slot of the NEXT call's parity) and the
// block sums of |r| at the trial states -- the input of the next call if this trial is accepted (k_decide clears the
// histogram again if it is not).  EMIT 2: warm histogram (bins around this call's median: the next call selects in one
// pass); EMIT 1: the 10 exponent bits = digit 0 of the exact select (many windows per launch: the ~1 global atomic per
// key that a 2048-bin histogram costs is dearer there than the second pass over the keys it saves).
// FUSED (latency mode, VBA_OPT_FUSION bit 0): 0 = the trial states are in memory, pose-chain blocks of 256 edges;
// 1 / 2 = the trial states do not exist yet and are formed here (vba_step.h: 1 landmark-only 6x6 solve, 2 recovery of the
// partitioned solve), 16 lanes per pose: an observation block for the poses its rows belong to (a handful), a pose-chain
// block for 16 poses = 15 edges, which also writes states_new / dpose for everybody after this kernel; 3 = the geometry
// of 1 / 2 with the trial states read from memory (a call of such a handle that cannot fuse: pivoted landmark-only solve).
constexpr int kEdgesPerBlock16 = 15;

// PART (many windows per launch, FUSED 0): 0 = one grid does both kinds of block; 1 = the observation blocks only, 2 = the
// pose-chain blocks only, as two launches -- the orbit propagation of the chain blocks costs the streaming blocks half
// their occupancy when both are one kernel (96 registers against 40).
// TILES (plain geometry, one grid, latency mode): an observation block takes TILES consecutive tiles of 256 rows.  The
// block's keys share ONE pass of bin reservations -- a window of 10^6 keys is ~2000 tiles, every one of which hits the few
// hundred central bins with a returning atomic of its own, and the same-address atomics queue up (block 100 of C5 waited
// 6 .. 8 of its 15 us for its bases, C3: 0.4 .. 1.8) -- and the grid fits the chip in one round.  Block sums stay per TILE
// (the slots and the bits of TILES = 1), the histogram is integers, a bucket is a set: the results do not depend on TILES.
template <int EMIT, int FUSED, int PART = 0, int TILES = 1>
__global__ __launch_bounds__(kObsBlock) void k_trial(DevView V) {
    static_assert(PART == 0 || FUSED == 0, "split launches exist for the plain geometry only");
    static_assert(TILES == 1 || (PART == 0 && FUSED == 0), "tiled observation blocks exist for the plain one-grid geometry only");
    constexpr bool FORM = FUSED == 1 || FUSED == 2;
    __shared__ double red[kObsBlock / 64];
    __shared__ double redt[TILES > 1 ? 2 * TILES * (kObsBlock / 64) : 1];
    __shared__ unsigned lh[EMIT == 2 ? kSelBins : (EMIT == 1 ? 1024 : 1)];
    __shared__ double snew[FORM ? (kObsBlock + 1) * 10 : 1];
    __shared__ int lpose[FORM ? kObsBlock : 1];
    __shared__ unsigned wlead[FORM ? 4 : 1];
    const int w = blockIdx.y;
    // The pose of this thread's row (and of the row in front of it), requested FIRST of all: the address needs the block and
    // thread index only (clamped into the window's rows), and everything an observation block does hangs on it -- this way
    // the round trip runs beside the one of the call counter instead of behind it.
    int early_pose = 0, early_prev = 0;
    constexpr bool kEarlyPose = FUSED == 1 || FUSED == 2;      // (the streaming blocks of the batched mode keep their load where it was)
    if (kEarlyPose && PART != 2) {
        const int64_t ke = min((int64_t)blockIdx.x * kObsBlock + threadIdx.x, V.m_max - 1);
        const int* op = V.opose + 2 * (size_t)w * V.obs_stride;
        early_pose = op[ke];
        early_prev = op[ke > 0 ? ke - 1 : 0];
    }
    VBA_SKIP_CALL(V, w);
    WinScalars& sc = V.sc[w];
    if (sc.done) return;
    VBA_OSTAMP(0);
    const int n = V.n[w], m = V.m[w];
    const StepParams& prm = V.prm;
    const int par = V.par;
    const int tid = threadIdx.x;
    double s = 0.0, s_raw = 0.0;
    const size_t sb = (size_t)w * V.n_max;
    const int nfat = (V.nblk_obs + TILES - 1) / TILES;     // observation blocks of this grid
    const bool obs_block = PART == 1 || (PART == 0 && (int)blockIdx.x < nfat);
    // this block's place in part_trial (an observation block of several tiles: its first tile's)
    const int part_slot = PART == 2 ? V.nblk_obs + (int)blockIdx.x : (obs_block ? (int)blockIdx.x * TILES : V.nblk_obs + ((int)blockIdx.x - nfat));
    const double lam32 = (double)(float)sc.lam[par];      // torch.eye() is float32 (BA_filtering.py:54)
    // a window that has fallen back to the pivoted kernels (landmark-only phase) reads the trial states they wrote
    const bool fz = FUSED == 2 || (FUSED == 1 && !(sc.fl[par] & 16u));
    const double wmax = bits_f64(sc.wmax_bits[par]);
    const double inv_wmax = 1.0 / wmax;
    const int l16 = tid & 15, grp = tid >> 4, gbase = (tid & 63) & ~15;
    unsigned long long wlo = 0ull;
    constexpr int kEmitBins = EMIT == 2 ? kSelBins : 1024;     // warm bins, or the 10 exponent bits (digit 0 of the exact select)
    if (EMIT == 2) wlo = warm_range_start(f64_bits(sc.c_obs), V.warm_shift);
    if (EMIT && PART != 2 && obs_block) {
        for (int b = tid; b < kEmitBins; b += kObsBlock) lh[b] = 0u;
        __syncthreads();
    }
    if (PART != 2 && blockIdx.x == 0) {
        // digits 1, 2 of an exact select are dead since the accumulation; the list of the next warm select starts empty
        unsigned* h12 = histd_of(V, w, 1);
        for (int b = tid; b < 2 * kSelBins; b += kObsBlock) h12[b] = 0u;
        if (V.wbucket) {    // inline select: nobody clears these in front of the next accumulation
            unsigned* h0 = hist0_of(V, w, par);     // this call's histogram: its last readers were the accumulation's prologues
            for (int b = tid; b < kSelBins; b += kObsBlock) h0[b] = 0u;
            if (tid == 0) sc.wmax_bits[par ^ 1] = 0ull;
        }
        if (tid == 0) {
            sc.sel_cnt = 0u;
            sc.pending = V.call;
            if (EMIT == 2) sc.warm_lo[par ^ 1] = wlo;
            if (EMIT == 1) {        // digit 0 of the next call's exact select is the histogram this kernel leaves
                sc.sel_prefix[0] = 0ull;
                sc.sel_rank[0] = (2 * (long long)m - 1) / 2;
            }
            if (FUSED == 1 && fz) sc.lam32 = lam32;
        }
    }
    VBA_OSTAMP(1);
    unsigned bad = 0u;
    unsigned kbin[2 * TILES] = {}, kslot[2 * TILES] = {};      // EMIT 2: warm bin of this thread's keys and their place in the block's share
    double kkey[2 * TILES] = {};
    bool kvalid[TILES] = {};
    double s_tile[TILES] = {}, sraw_tile[TILES] = {};          // (TILES > 1: the sums of the tiles, reduced together below)
    if (PART != 2 && obs_block) {
    // (the loop over this block's tiles; its body keeps the indentation of the one tile it was)
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl) {
        const int k = (blockIdx.x * TILES + tl) * kObsBlock + tid;
        const size_t ob = (size_t)w * V.obs_stride, mb = (size_t)w * V.m_max;
        const bool have = k < m;
        const int pose = have ? (kEarlyPose ? early_pose : V.opose[2 * ob + k]) : -1;
        const double* stp = V.states_new + (sb + (have ? pose : 0)) * 10;
        if (FORM && fz) {
            // the poses of this block's rows (rows are pose sorted): the first row of every pose inside the block leads,
            // leaders are numbered in row order and 16 lanes form the trial state of each
            const int prev = (have && tid > 0) ? early_prev : -2;
            const bool lead = have && (tid == 0 || prev != pose);
            const unsigned long long lm = __ballot(lead);
            const int lane = tid & 63, wv = tid >> 6;
            if (lane == 0) wlead[wv] = (unsigned)__popcll(lm);
            __syncthreads();
            unsigned before = 0, nlead = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                before += q < wv ? wlead[q] : 0u;
                nlead += wlead[q];
            }
            const int slot = (int)(before + (unsigned)__popcll(lm & ((2ull << lane) - 1ull))) - 1;   // leaders up to and including me
            if (lead) lpose[slot] = pose;
            __syncthreads();
            VBA_OSTAMP(2);
            for (unsigned base = 0; base < nlead; base += 16) {
                const unsigned idx = base + (unsigned)grp;
                const bool live = idx < nlead;
                double o[10], d9[9];
                unsigned b2 = 0u;
                pose_trial_state_group<FORM ? FUSED : 1>(V, w, live ? lpose[idx] : 0, live, l16, gbase, inv_wmax, lam32, o, d9, b2);
                if (live && l16 == 0) {
#pragma unroll
                    for (int r = 0; r < 10; ++r) snew[(size_t)idx * 10 + r] = o[r];
                }
            }
            __syncthreads();
            VBA_OSTAMP(3);
            stp = snew + (size_t)(slot < 0 ? 0 : slot) * 10;
        }
        if (have) {
            PoseCam pc;
            pose_camera(stp, V.intr + (sb + pose) * 4, pc);
            double u, v, cam[3], d;
            project(pc, V.ox[ob + k], V.oy[ob + k], V.oz[ob + k], u, v, cam, d);
            const double wk = (V.wraw[mb + k] / wmax) * V.oconf[ob + k];
            const double du = V.ou[ob + k] - u, dv = V.ov[ob + k] - v;
            s = fabs(du * wk) + fabs(dv * wk);
            s_tile[tl] = s;
            if (EMIT) {
                const double ru = fabs(du), rv = fabs(dv);
                reinterpret_cast<double2*>(V.absr + 2 * mb)[k] = make_double2(ru, rv);
                s_raw = ru + rv;
                sraw_tile[tl] = s_raw;
                if (EMIT == 2) {
                    kbin[2 * tl] = warm_bin(f64_bits(ru), wlo, V.warm_shift);
                    kbin[2 * tl + 1] = warm_bin(f64_bits(rv), wlo, V.warm_shift);
                    if (V.wbucket) {        // the place inside the block's share of the bin: a returning atomic
                        kslot[2 * tl] = atomicAdd(&lh[kbin[2 * tl]], 1u);
                        kslot[2 * tl + 1] = atomicAdd(&lh[kbin[2 * tl + 1]], 1u);
                        kkey[2 * tl] = ru;
                        kkey[2 * tl + 1] = rv;
                        kvalid[tl] = true;
                    } else {
                        atomicAdd(&lh[kbin[2 * tl]], 1u);
                        atomicAdd(&lh[kbin[2 * tl + 1]], 1u);
                    }
                } else {
                    atomicAdd(&lh[(unsigned)(f64_bits(ru) >> 53) & 1023u], 1u);
                    atomicAdd(&lh[(unsigned)(f64_bits(rv) >> 53) & 1023u], 1u);
                }
            }
        }
    }   // tiles
    } else if (PART != 1) {
        const int db = part_slot - V.nblk_obs;
        const bool reg = V.reg && !prm.initialize;
        // which pose / edge this thread evaluates, and where its two states are
        int i;                      // pose; edge i -> i + 1
        bool edge_thread;           // this thread evaluates the edge i -> i + 1
        bool pose_thread;           // this thread accounts for pose i (prior residual; FORM: writes its trial state)
        const double* st;
        const double* sn;
        if (FUSED == 0) {
            i = db * kObsBlock + tid;
            edge_thread = pose_thread = true;
            st = V.states_new + (sb + i) * 10;
            sn = st + 10;
        } else {
            const int i0 = db * kEdgesPerBlock16;
            const int j = i0 + grp;                     // the pose of this 16-lane group
            i = j;
            edge_thread = l16 == 0 && grp < kEdgesPerBlock16;
            // the block's 16th pose is the next block's first -- unless there is no next block
            pose_thread = l16 == 0 && (grp < kEdgesPerBlock16 || j / kEdgesPerBlock16 >= V.nblk_dyn);
            if (FORM && fz) {
                const bool live = j < n;
                double o[10], d9[9];
                unsigned b2 = 0u;
                pose_trial_state_group<FORM ? FUSED : 1>(V, w, j, live, l16, gbase, inv_wmax, lam32, o, d9, b2);
                bad = b2;
                if (live && l16 == 0) {
#pragma unroll
                    for (int r = 0; r < 10; ++r) snew[(size_t)grp * 10 + r] = o[r];
                    if (pose_thread) {
#pragma unroll
                        for (int r = 0; r < 10; ++r) V.states_new[(sb + j) * 10 + r] = o[r];
#pragma unroll
                        for (int r = 0; r < 9; ++r) V.dpose[(sb + j) * 9 + r] = d9[r];
                        if (V.host_states) {        // (one-window handles: sb == 0)
#pragma unroll
                            for (int r = 0; r < 10; ++r) V.host_states[((size_t)par * V.n_max + j) * 10 + r] = o[r];
                        }
                    }
                }
                if (FUSED == 1 && live && j == n - 1) {     // last_hessian of a landmark-only call: H / w_max on the 6x6, zeros elsewhere
                    const double* H = V.Hraw + (sb + j) * 21;
                    for (int e = l16; e < 81; e += 16) {
                        const int a = e / 9, c = e % 9;
                        V.lastD[(size_t)w * 81 + e] = (a < 6 && c < 6) ? H[sym6(a, c)] * inv_wmax : 0.0;
                    }
                }
                __syncthreads();
                st = snew + (size_t)grp * 10;
                sn = st + 10;
            } else {
                st = V.states_new + (sb + j) * 10;
                sn = st + 10;
            }
        }
        if (edge_thread && !prm.initialize && i < n - 1) {
            double x[6] = {st[0], st[1], st[2], st[7], st[8], st[9]};
            const int steps = V.steps[sb + i];
            if (steps > 0 || V.hop) {       // (a long edge's orbit residual is k_long_trial's, in a slot of its own)
                propagate_gap<false>(x, nullptr, abs(steps), V.hop);
                s = fabs(x[0] - sn[0]) + fabs(x[1] - sn[1]) + fabs(x[2] - sn[2]) +
                    fabs((x[3] - sn[7]) * kVelCoeff) + fabs((x[4] - sn[8]) * kVelCoeff) + fabs((x[5] - sn[9]) * kVelCoeff);
            }
            double att = fabs(attitude_residual(st + 3, V.cumrot + (sb + i) * 4, sn + 3));
            // BA_reg evaluates the trial's dynamics residual with quat_coeff_prior = 1 where BA passes quat_coeff = 100
            // (BA_filtering.py:172, 174 vs :63, 65): reproduced as written
            if (reg) att *= 1.0 / kQuatCoeff;
            s += att;
            s *= prm.sqrt_sigma;
        }
        if (reg && pose_thread && i < n) {     // sum |r_prior| at the trial states (BA_filtering.py:175, 178), not scaled by sigma
            double r6[6];
            prior_residual(V.prior_H + (sb + i) * 36, V.prior_x + (sb + i) * 6, st, r6);
            s += fabs(r6[0]) + fabs(r6[1]) + fabs(r6[2]) + fabs(r6[3]) + fabs(r6[4]) + fabs(r6[5]);
        }
    }
    // bin buckets: the block reserves its share of every bin it touched with one returning atomic per bin -- requested
    // here, in flight while the block sums below are formed
    VBA_OSTAMP(4);
    constexpr int kBinsPerThread = kSelBins / kObsBlock;
    static_assert(kSelBins % kObsBlock == 0, "bins per thread");
    unsigned bb[kBinsPerThread] = {};
    const bool bucketing = EMIT == 2 && PART == 0 && obs_block && V.wbucket;
    if (bucketing) {
        __syncthreads();        // the block's counts are complete
        unsigned* hist = hist0_of(V, w, par ^ 1);
#pragma unroll
        for (int q = 0; q < kBinsPerThread; ++q) {
            const unsigned c = lh[tid + q * kObsBlock];
            bb[q] = c ? atomicAdd(&hist[tid + q * kObsBlock], c) : 0u;
        }
    }
    VBA_OSTAMP(5);
    if (TILES > 1 && obs_block) {
        // the 2 TILES sums of the tiles in ONE round of barriers; per sum the order of block_sum (waves 0 .. 3 onto 0.0)
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            const double a = wave_sum(s_tile[tl]), r2 = wave_sum(sraw_tile[tl]);
            if ((tid & 63) == 0) {
                redt[(2 * tl) * (kObsBlock / 64) + (tid >> 6)] = a;
                redt[(2 * tl + 1) * (kObsBlock / 64) + (tid >> 6)] = r2;
            }
        }
        __syncthreads();
        if (tid < 2 * TILES && part_slot + tid / 2 < V.nblk_obs) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < kObsBlock / 64; ++i) t += redt[tid * (kObsBlock / 64) + i];
            if (tid & 1) { if (EMIT) V.part_next[(size_t)w * V.nblk_obs + part_slot + tid / 2] = t; }
            else V.part_trial[(size_t)w * V.trial_stride + part_slot + tid / 2] = t;
        }
    } else {
        const double t = block_sum<kObsBlock>(s, red);
        if (tid == 0) V.part_trial[(size_t)w * V.trial_stride + part_slot] = t;
    }
    VBA_OSTAMP(6);
    if (FORM && !obs_block) {
        const unsigned long long bp = __ballot(bad & 1u), bn = __ballot(bad & 2u);
        if ((tid & 63) == 0 && (bp || bn)) atomicOr(&sc.fl[par], (bp ? (8u | 16u) : 0u) | (bn ? 2u : 0u));
    }
    if (EMIT && PART != 2 && obs_block) {
        if (TILES == 1) {
            const double t_raw = block_sum<kObsBlock>(s_raw, red);
            if (tid == 0) V.part_next[(size_t)w * V.nblk_obs + part_slot] = t_raw;
        }
        unsigned* hist = hist0_of(V, w, par ^ 1);
        if (bucketing) {
            // ... and each key goes to its place: the next call finds the keys of the wanted bin together, no pass over all keys
            // (k_select_warm) is needed
#pragma unroll
            for (int q = 0; q < kBinsPerThread; ++q) lh[tid + q * kObsBlock] = bb[q];
            __syncthreads();
            VBA_OSTAMP(7);
            double* pool = V.wbucket + ((size_t)w * 2 + (par ^ 1)) * kSelBins * (size_t)V.bucket_cap;
#pragma unroll
            for (int q = 0; q < 2 * TILES; ++q) {
                if (kvalid[q / 2]) {
                    const unsigned slot = lh[kbin[q]] + kslot[q];
                    if (kbin[q] >= 1u && kbin[q] <= 2046u && slot < (unsigned)V.bucket_cap) pool[(size_t)kbin[q] * V.bucket_cap + slot] = kkey[q];
                }
            }
            VBA_OSTAMP(8);
        } else {
            for (int b = tid; b < kEmitBins; b += kObsBlock) {
                const unsigned c = lh[b];
                if (c) atomicAdd(&hist[b], c);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- sharded mode: front of a carried call
// Observation-sharded window, carried-keys protocol (vba_sh_run_schedule).  The trial kernel of every rank has left the keys of
// ITS rows in bin buckets, their warm histogram and its block sums in the rank's exchange buffer; `gathered` holds those of
// all R ranks (one all-gather of ~10 kB per rank instead of 16 B per observation).  One block, the same arithmetic on the same
// data on every rank, so every rank decides alike:
//   fold     the accept test of the call in front on the gathered block sums (observation part: every rank's; pose-chain
//            part: rank 0's -- all ranks computed the same); a first trial that is not cleanly accepted leaves everything
//            untouched and the window stalls there for the host's LM loop;
//   resolve  the R histograms are added up (integers), the bin of the global lower median is found, and this rank's bucket
//            of that bin goes into `bucket_out` = [count, keys ...] for the second (and last key-sized) exchange; a rank whose
//            bucket overflowed, a bin longer than 1024 keys over all ranks or a rank outside the binned range is a MISS: the
//            call takes the exact select over all keys instead (rank-consistent: the decision uses gathered data only).
// Layout of a rank's slot of `gathered` (doubles): [hist: 1024 (2048 u32) | part_next: nblk_obs | part_trial: nblk_obs + nblk_dyn].
__global__ __launch_bounds__(256) void k_sh_front(DevView V, const double* gathered, int ranks, int slot_len, double* bucket_out, int do_fold,
                                                  int do_resolve) {
    __shared__ unsigned lds_u[260];
    __shared__ double red[5][4];
    __shared__ unsigned over;
    const int w = 0, t = threadIdx.x;
    WinScalars& sc = V.sc[w];
    const int off_next = 1024, off_trial = 1024 + V.nblk_obs;
    // Everything this kernel reads is requested FIRST (the addresses depend on nothing it learns later): the scalars of the
    // window, the block sums and histograms of all ranks, this rank's own histogram, the inputs of the accept test -- one
    // round trip to memory instead of a chain of five (one block: nobody hides a latency here).
    const int seen_call = sc.call_idx, seen_pending = sc.pending, seen_miss = sc.miss;
    const int pc = V.par ^ 1;
    const double lam_in = sc.lam[pc], so_in = sc.sum_in[pc];
    const unsigned fl_in = sc.fl[pc];
    const unsigned long long lo = sc.warm_lo[V.par];
    double s_next = 0.0, s_trial = 0.0, s_pred = 0.0;
    for (int q = 0; q < ranks; ++q) {
        const double* slot = gathered + (size_t)q * slot_len;
        for (int b = t; b < V.nblk_obs; b += 256) { s_next += slot[off_next + b]; s_trial += slot[off_trial + b]; }
    }
    for (int b = t; b < V.nblk_dyn + (V.prev.initialize ? 0 : V.nblk_long); b += 256) s_trial += gathered[off_trial + V.nblk_obs + b];
    if (do_fold && !V.prev.initialize) {
        const double* pp = V.part_pred + ((size_t)w * 2 + pc) * V.pred_stride;
        for (int b = t; b < V.nblk_pred + V.nblk_long; b += 256) s_pred += pp[b];
    }
    unsigned hl[8], mine[8];
    if (do_resolve) {
        const uint4* own4 = reinterpret_cast<const uint4*>(hist0_of(V, w, V.par)) + 2 * t;
        const uint4 o0 = own4[0], o1 = own4[1];
        mine[0] = o0.x; mine[1] = o0.y; mine[2] = o0.z; mine[3] = o0.w; mine[4] = o1.x; mine[5] = o1.y; mine[6] = o1.z; mine[7] = o1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) hl[j] = 0u;
        for (int q = 0; q < ranks; ++q) {
            const uint4* g4 = reinterpret_cast<const uint4*>(gathered + (size_t)q * slot_len) + 2 * t;
            const uint4 g0 = g4[0], g1 = g4[1];
            hl[0] += g0.x; hl[1] += g0.y; hl[2] += g0.z; hl[3] += g0.w; hl[4] += g1.x; hl[5] += g1.y; hl[6] += g1.z; hl[7] += g1.w;
        }
    }
    if (do_fold) {
        if (!(V.call >= 0 && seen_pending == V.call - 1 && seen_call == V.call - 1)) return;
    } else {
        if (!((V.call < 0 || seen_call == V.call) && (V.redo == 2 || (seen_miss != 0) == (V.redo != 0)))) return;     // VBA_SKIP_CALL on the snapshot
    }
    if (do_fold) {
        DecideIn in;
        in.s_pred = s_pred;
        in.s_prior = 0.0;
        in.s_trial = s_trial;
        in.s_next = s_next;
        in.lam_in = lam_in;
        in.so = so_in;
        in.flags = fl_in;
        const DecideOut d = decide_finish(V, w, in, V.prev, 0, 0.0, nullptr, 0, red);
        if (!(d.accept && !(d.flags & (2u | 8u | 32u)))) return;        // not clean: no trace
        fold_commit(V, w, d);
        if (!do_resolve && t == 0) {                    // the last call of a schedule: decided, nothing begins
            sc.pending = -1;
            sc.n_trials = 1;                            // (vba_get_states reports the decided call's)
            sc.done = 1;
        }
    } else if (do_resolve) {
        // (the call in front was decided by k_decide, which knows this rank's part only)
        const double v = wave_sum(s_next);
        __syncthreads();
        if ((t & 63) == 0) red[0][t >> 6] = v;
        __syncthreads();
        if (t == 0) sc.sum_in[V.par] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    }
    if (!do_resolve) return;
    if (t == 0) over = 0u;
    __syncthreads();
    const int64_t count = 2 * V.m_total;
    unsigned long long prefix;
    long long rank;
    unsigned in_bin;
    select_resolve_loaded(hl, kSelBins, 11, 0ull, (count - 1) / 2, prefix, rank, lds_u, &in_bin);
    const unsigned bin = (unsigned)prefix;
    // every rank's bucket of that bin must be complete
    if ((unsigned)(t * 8) <= bin && bin < (unsigned)(t * 8 + 8)) {
        for (int q = 0; q < ranks; ++q)
            if (reinterpret_cast<const unsigned*>(gathered + (size_t)q * slot_len)[bin] > (unsigned)V.bucket_cap) over = 1u;
    }
    __syncthreads();
    const bool hit = lo != ~0ull && bin >= 1u && bin <= 2046u && in_bin <= 1024u && !over && !V.warm_force_miss;
    if (t == 0) {
        front_commit(V, w, hit, bin, rank, in_bin, false);
        if (hit) sc.sel_cnt = in_bin;
        if (V.wmax_ext) *V.wmax_ext = 0ull;
    }
    if (!hit) return;
    // this rank's bucket of the bin: [count, keys ...]
    unsigned my_cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if ((unsigned)(t * 8 + j) == bin) my_cnt = mine[j];
    __syncthreads();
    if ((unsigned)(t * 8) <= bin && bin < (unsigned)(t * 8 + 8)) lds_u[20] = my_cnt;
    __syncthreads();
    const unsigned cnt = lds_u[20];
    const double* bucket = V.wbucket + (((size_t)w * 2 + V.par) * kSelBins + bin) * (size_t)V.bucket_cap;
    if (t == 0) bucket_out[0] = (double)cnt;
    for (unsigned q = t; q < cnt; q += 256) bucket_out[1 + q] = bucket[q];
}

// a call that missed its warm select is repeated with the exact select over all keys: the window takes part again
__global__ void k_sh_clear_miss(DevView V) {
    V.sc[0].miss = 0;
    V.sc[0].fl[V.par] = 0u;
    V.host_head[0].flags = 0u;
}
void launch_sh_clear_miss(const DevView& V, hipStream_t s) { hipLaunchKernelGGL(k_sh_clear_miss, dim3(1), dim3(1), 0, s, V); }

void launch_sh_front(const DevView& V, const double* gathered, int ranks, int slot_len, double* bucket_out, int do_fold, int do_resolve, hipStream_t s) {
    hipLaunchKernelGGL(k_sh_front, dim3(1), dim3(256), 0, s, V, gathered, ranks, slot_len, bucket_out, do_fold, do_resolve);
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
    wt[k] = (V.wraw[mb + k] / bits_f64(V.sc[w].wmax_bits[V.par])) * V.oconf[ob + k];
}

// window 0's states and damping copied to every other window (vba_set_states with window == -1)
__global__ __launch_bounds__(256) void k_broadcast_states(DevView V, int n, double lamda) {
    const int w = blockIdx.y;
    const double* src = V.states;
    double* dst = V.states + (size_t)w * V.n_max * 10;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (w > 0 && k < n * 10) dst[k] = src[k];
    if (k == 0) V.sc[w].lam[V.par] = lamda;
}

// pose / row counts of a freshly uploaded window (m < 0: keep)
__global__ void k_set_counts(int* n_arr, int* m_arr, int w, int n, int m) {
    n_arr[w] = n;
    if (m >= 0) m_arr[w] = m;
}

__global__ void k_reset_calls(DevView V) {
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w < V.W) {
        V.sc[w].call_idx = 0;
        V.sc[w].pending = -1;
        V.sc[w].miss = 0;       // (a speculated call that was dropped may have recorded a missed warm select: nobody will repeat it)
    }
}

// Digit-0 histogram of parity V.par (a warm histogram left by a trial whose states were then replaced, or one that a
// repeated select is about to rebuild by exponent) and, which == 1, digits 1 and 2 as well (an exact select that no
// trial followed).  which == 2: only the windows whose warm select missed.
__global__ __launch_bounds__(1024) void k_clear_hist(DevView V, int which) {
    const int w = blockIdx.x;
    if (which == 2 && !V.sc[w].miss) return;
    unsigned* h0 = hist0_of(V, w, V.par);
    for (int b = threadIdx.x; b < kSelBins; b += 1024) h0[b] = 0u;
    if (which == 1) {
        unsigned* h12 = histd_of(V, w, 1);
        for (int b = threadIdx.x; b < 2 * kSelBins; b += 1024) h12[b] = 0u;
    }
}

void launch_clear_hist(const DevView& V, int which, hipStream_t s) {
    hipLaunchKernelGGL(k_clear_hist, dim3(V.W), dim3(1024), 0, s, V, which);
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

// exact select (digits 1 and 2 over the keys; digit 0 comes from k_obs_residual, or -- with_digit0 -- from a pass of its
// own: sharded mode's gathered keys, a select repeated after a warm miss)
void launch_select(const DevView& V, bool with_digit0, hipStream_t s) {
    const int64_t count = V.abs_all ? V.abs_all_count : 2 * V.m_max;
    const dim3 b(256);
    if (!V.lat) {
        const int nb = (int)((count + 256 * 32 - 1) / (256 * 32));
        const dim3 g(nb > 0 ? nb : 1, V.W);
        if (with_digit0) hipLaunchKernelGGL((k_select_pass<0, false, 32>), g, b, 0, s, V);
        hipLaunchKernelGGL((k_select_pass<1, false, 32>), g, b, 0, s, V);
        hipLaunchKernelGGL((k_select_pass<2, true, 32>), g, b, 0, s, V);
    } else {
        const int nb = (int)((count + 256 * kSelItems - 1) / (256 * kSelItems));
        const dim3 g(nb > 0 ? nb : 1, V.W);
        if (with_digit0) hipLaunchKernelGGL((k_select_pass<0, false, kSelItems>), g, b, 0, s, V);
        hipLaunchKernelGGL((k_select_pass<1, false, kSelItems>), g, b, 0, s, V);
        hipLaunchKernelGGL((k_select_pass<2, true, kSelItems>), g, b, 0, s, V);
    }
}

// warm select on carried keys: one pass (plus, V.fold, the accept test of the call in front)
void launch_select_warm(const DevView& V, hipStream_t s) {
    const int64_t count = 2 * V.m_max;
    if (!V.lat) {
#ifndef VBA_SELW_ITEMS
#define VBA_SELW_ITEMS 32
#endif
        const int nb = (int)((count + 256 * VBA_SELW_ITEMS - 1) / (256 * VBA_SELW_ITEMS));
        hipLaunchKernelGGL((k_select_warm<VBA_SELW_ITEMS>), dim3(nb > 0 ? nb : 1, V.W), dim3(256), 0, s, V);
    } else {
        const int nb = (int)((count + 256 * kSelItems - 1) / (256 * kSelItems));
        hipLaunchKernelGGL((k_select_warm<kSelItems>), dim3(nb > 0 ? nb : 1, V.W), dim3(256), 0, s, V);
    }
}

void launch_select_finish(const DevView& V, hipStream_t s) {
    hipLaunchKernelGGL(k_select_finish, dim3(V.W), dim3(256), 0, s, V);
}

void launch_obs_accumulate(const DevView& V, hipStream_t s) {
    const int G = V.acc_lanes;
    const int nb = (V.n_max * G + 255) / 256;
    // V.dyn_in_acc: the blocks of the dynamics factor are appended to the grid
    // V.sel_inline: one more block, which evaluates the folded accept test and records the start of the call
    const dim3 g(nb + (V.dyn_in_acc ? (V.n_max * kDynLanes + 255) / 256 : 0) + (V.sel_inline ? 1 : 0), V.W), b(256);
#ifndef VBA_ACC_PAIR
#define VBA_ACC_PAIR 1
#endif
    constexpr bool kPair = VBA_ACC_PAIR != 0;
    if (V.median_ready && !V.dyn_in_acc && !V.sel_inline && (G == 8 || G == 16)) {      // many windows per launch
#ifndef VBA_ACC_GROUPS
#define VBA_ACC_GROUPS 4
#endif
        // a block walks up to VBA_ACC_GROUPS groups of poses (its grid stride) -- fewer when the windows of the handle would
        // otherwise leave compute units without a block (the chip holds 512 of these blocks at once)
        static const int groups_env = std::getenv("VBA_X_ACCGROUPS") ? std::atoi(std::getenv("VBA_X_ACCGROUPS")) : 0;
        int groups = VBA_ACC_GROUPS;
        while (groups > 1 && (int64_t)V.W * ((nb + groups - 1) / groups) < 2048) groups >>= 1;
        if (groups_env > 0) groups = groups_env;
        const dim3 gb((nb + groups - 1) / groups, V.W);
        if (G == 8) hipLaunchKernelGGL((k_obs_accumulate<8, kPair, true>), gb, b, 0, s, V);
        else hipLaunchKernelGGL((k_obs_accumulate<16, false, true>), gb, b, 0, s, V);
        return;
    }
    switch (G) {
        case 4: hipLaunchKernelGGL((k_obs_accumulate<4, kPair, false>), g, b, 0, s, V); break;
        case 8: hipLaunchKernelGGL((k_obs_accumulate<8, kPair, false>), g, b, 0, s, V); break;
        case 16: hipLaunchKernelGGL((k_obs_accumulate<16, false, false>), g, b, 0, s, V); break;
        case 32: hipLaunchKernelGGL((k_obs_accumulate<32, false, false>), g, b, 0, s, V); break;
        default: hipLaunchKernelGGL((k_obs_accumulate<64, false, false>), g, b, 0, s, V); break;
    }
    // the long edges of the dynamics factor that rode in this grid (behind the folded accept test: the window has moved on to
    // this call, or the kernel leaves it alone as every later kernel of the call does)
    if (V.dyn_in_acc) launch_long_factor(V, s);
}

template <int EMIT>
static void launch_trial_emit(const DevView& V, hipStream_t s) {
    const dim3 g(V.nblk_obs + V.nblk_dyn, V.W), b(kObsBlock);
    const int f = V.fused_trial;        // 0..3, see k_trial; V.nblk_dyn is the pose-chain block count of that geometry
    if (f == 0 && !V.lat && !V.wbucket) {        // many windows: the two kinds of block as two launches
        hipLaunchKernelGGL((k_trial<EMIT, 0, 2>), dim3(V.nblk_dyn, V.W), b, 0, s, V);
        hipLaunchKernelGGL((k_trial<EMIT, 0, 1>), dim3(V.nblk_obs, V.W), b, 0, s, V);
        return;
    }
    if (f == 1) hipLaunchKernelGGL((k_trial<EMIT, 1>), g, b, 0, s, V);
    else if (f == 2) hipLaunchKernelGGL((k_trial<EMIT, 2>), g, b, 0, s, V);
    else if (f == 3) hipLaunchKernelGGL((k_trial<EMIT, 3>), g, b, 0, s, V);
    else if (EMIT == 2 && V.trial_tiles == 8) hipLaunchKernelGGL((k_trial<EMIT, 0, 0, EMIT == 2 ? 8 : 1>), dim3((V.nblk_obs + 7) / 8 + V.nblk_dyn, V.W), b, 0, s, V);
    else if (EMIT == 2 && V.trial_tiles == 4) hipLaunchKernelGGL((k_trial<EMIT, 0, 0, EMIT == 2 ? 4 : 1>), dim3((V.nblk_obs + 3) / 4 + V.nblk_dyn, V.W), b, 0, s, V);
    else if (EMIT == 2 && V.trial_tiles == 2) hipLaunchKernelGGL((k_trial<EMIT, 0, 0, EMIT == 2 ? 2 : 1>), dim3((V.nblk_obs + 1) / 2 + V.nblk_dyn, V.W), b, 0, s, V);
    else hipLaunchKernelGGL((k_trial<EMIT, 0>), g, b, 0, s, V);
}

void launch_trial(const DevView& V, hipStream_t s) {
    if (V.emit == 2) launch_trial_emit<2>(V, s);
    else if (V.emit == 1) launch_trial_emit<1>(V, s);
    else launch_trial_emit<0>(V, s);
    launch_long_trial(V, s);        // the orbit residual of the long edges at the trial states (vba_long.hip)
}

void launch_debug_project(const DevView& V, int w, int m, double* est, double* J, double* wt, hipStream_t s) {
    hipLaunchKernelGGL(k_debug_project, dim3((m + kObsBlock - 1) / kObsBlock), dim3(kObsBlock), 0, s, V, w, est, J, wt);
}

}  // namespace vba
