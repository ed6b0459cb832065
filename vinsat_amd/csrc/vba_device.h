// vba_device.h -- device memory layout of a BA context and small wave/block primitives (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vba_math.h"

namespace vba {

constexpr int kWave = 64;
constexpr int kObsBlock = 256;          // threads per block of the per-observation kernels
constexpr int kSelItems = 8;            // keys per thread per radix-select pass
constexpr int kSelPasses = 6;           // 63 key bits = 10 + 11 + 11 + 11 + 11 + 9
constexpr int kSelBins = 2048;
constexpr int kDynLanes = 8;            // lanes per pose in the dynamics kernel (6 tangents + attitude + spare)
// Long gaps of the pose chain (vba_long.hip).  An edge of more than kLongGap one-second RK4 steps is propagated PARALLEL IN TIME
// by a workgroup of its own (at most kLongCap such edges per window; further ones take the ordinary serial walk).  The kernels
// that walk the chain leave such an edge alone; the block sums of its residuals go into slots of their own behind the ordinary ones.
constexpr int kLongGap = 64;
constexpr int kLongCap = 64;
// chain states per window and call parity: 512 (24 kB: four 1000 s edges, or more shorter ones), 4096 (196 kB: ~37 such edges) for
// handles of up to kLongPoolFewWindows windows; a long gap the pool has no room for takes the ordinary serial walk
constexpr int kLongPool = 512;
constexpr int kLongPoolFew = 4096;
constexpr int kLongPoolFewWindows = 256;
constexpr int kHistStride = (kSelPasses + 1) * kSelBins;    // per window: digit 0 twice (call parity), digits 1..5
constexpr int kWarmCount = 192;         // warm select: up to this many keys are ranked by counting, longer lists by radix digits

// Warm select.  An accepted trial leaves the next call's keys behind (carried keys); their median is close to the
// median c of the call that produced them, so that trial histograms the keys into 2046 bins of 2^shift bit patterns
// starting at c / 2^k (bit patterns of non-negative doubles order like the values) instead of by exponent: the bin of
// the wanted rank then holds a short list and ONE pass over the keys (compaction of that bin) replaces two.  Bin 0 =
// below the range, bin 2047 = above: a rank that falls there (or no usable c) is a miss and the call repeats its
// select with the exact exponent / mantissa digits.
__host__ __device__ __forceinline__ unsigned warm_bin(unsigned long long key, unsigned long long lo, int shift) {
    if (key < lo) return 0u;
    const unsigned long long d = (key - lo) >> shift;
    return d < 2046ull ? 1u + (unsigned)d : 2047u;
}
// lower end of the binned range for median bits c: 2046 bins of 2^shift cover 2046 / 2^(52 - shift) binades --
// [c / 16, c * 16) for shift 44, [c / 4, c * 4) for shift 43
__host__ __device__ __forceinline__ unsigned long long warm_range_start(unsigned long long c_bits, int shift) {
    const unsigned long long binades = 2046ull >> (52 - shift);         // 7 (shift 44), 3 (shift 43)
    const unsigned long long down = (binades + 1) / 2;                   // 4, 2
    const unsigned long long e = c_bits >> 52;
    if (e <= down || e >= 0x7feull) return ~0ull;                        // zero / denormal / inf / nan median: no warm range (every key "below")
    return c_bits - (down << 52);
}

__host__ __device__ constexpr int sel_shift(int p) { return p == 0 ? 53 : p == 1 ? 42 : p == 2 ? 31 : p == 3 ? 20 : p == 4 ? 9 : 0; }
__host__ __device__ constexpr int sel_width(int p) { return p == 0 ? 10 : p == 5 ? 9 : 11; }

// Per-call constants (BA_filtering.py:22, 26); rewritten by the host before every step.
struct StepParams {
    double alpha, am2, expo;
    double sigma, sqrt_sigma;
    int alpha_is_2;
    int initialize;
    int iter;
    int pad;
};

// Per-window scalars living in device memory.
// Call parity.  What one call hands to the next lives twice, indexed by the parity p of the call that READS it: the
// states (DevView::states = buffer p, ::states_new = buffer p ^ 1: a call's last trial IS the next call's input, no
// copy), the damping lam[p], sum |r_obs| at the input states sum_in[p], the flags fl[p], the digit-0 histogram of the
// carried keys hist0[p].  A kernel of call c reads slot p and the end of the call (k_decide, or the first kernel of
// call c + 1 when the accept test is folded into it) writes slot p ^ 1 -- so the blocks of that first kernel, which all
// re-evaluate the accept test redundantly, never read a word that one of them writes.
struct WinScalars {
    double lam[2];                  // LM damping of the call of that parity (BA_filtering.py:52-79)
    double sum_in[2];               // sum |r_obs| at that call's input states (carried keys; sharded mode: local part)
    unsigned fl[2];                 // flags of that call
    double lam32;                   // float32-rounded damping of the last trial (BA_filtering.py:54)
    double c_obs;                   // lower median of |r_obs|
    unsigned long long wmax_bits[2];    // max raw weight of the call of that parity, as ordered bits (reset by the kernel in front of
                                        // the accumulation, or -- inline select -- by the previous call's trial kernel)
    double init_residual;
    double trial_residual;
    double sum_abs_rpred;           // sqrt(sigma) * sum |r_pred|
    int done;
    int n_trials;
    unsigned sel_cnt;               // keys appended to the compacted select list
    int sel_mode;                   // how the list was made: 0 = exact digits (prefix of 32 bits known), 1 = one warm bin
    int call_idx;                   // calls of the current vba_run_schedule completed by this window
    int pending;                    // call whose first trial has been evaluated (k_trial) and not yet decided; -1: none
    int miss;                       // the warm select of the current call missed: the call is repeated with the exact select
    int pad2;
    unsigned long long warm_lo[2];  // lower end of the warm bins of hist0[p]
    unsigned long long warm_base;   // bit pattern at which the selected warm bin starts (sel_mode 1)
    unsigned long long sel_prefix[kSelPasses + 1];
    long long sel_rank[kSelPasses + 1];
    double last_hessian[81];
};

// What the host needs to know after a trial; written by k_decide straight into mapped pinned host memory so that
// the host only has to wait for the stream (no device-to-host copy in the loop).
struct WinHead {
    double lamda;
    double trial_residual;
    int done;
    int n_trials;
    unsigned flags;
    int call_idx;
    double last_hessian[81];        // of the call that has just been decided (BA_filtering.py:97), so that a pipelined call needs no copy
};

// Everything a kernel needs; passed by value.  Arrays of W windows use the *_max strides.
struct DevView {
    int W, n_max;
    int n_min;                      // smallest pose count over the windows of the handle (host maintained)
    int call;                       // >= 0: this launch belongs to call `call` of a schedule, windows elsewhere in it skip
    int par;                        // parity of this call (see WinScalars)
    int fold;                       // this call's first kernel first evaluates the accept test of call - 1 (chained schedule)
    int pending_only;               // k_decide: only windows whose trial of this call is evaluated and not yet decided (sc.pending)
    int redo;                       // 1: this launch repeats the call for the windows whose warm select missed (sc.miss), others skip
    int lat;                        // latency mode (few windows): fused kernels, see vba_api.hip
    int fuse_blocks;                // latency mode: the chunk elimination forms the blocks of its chunk itself (VBA_OPT_FUSION bit 1)
    int resident;                   // latency mode: the solve is one grid of producer and waiting consumer blocks (k_solve_resident;
                                    // VBA_OPT_FUSION bit 5: chunks + cyclic-reduction groups, bit 6: + the one-workgroup tail)
    int res_stride;                 // flags per window
    unsigned* res_flags;            // [W][res_stride] epoch of the launch that last completed the block
    double* wbucket;                // [W][2 (parity)][kSelBins][bucket_cap] carried keys by warm bin (latency mode; null: none)
    int bucket_cap;
    int median_ready;               // many windows: k_select_finish has left the median in sc.c_obs
    int sel_inline;                 // this call's accumulation starts the call: inline warm select on the buckets (+ folded accept test)
    int cr_levels;                  // cyclic-reduction levels that run as their own multi-CU kernel in front of the one-workgroup kernel (1 or 2)
    int chunk_waves;                // partitioned solve: waves per chunk (2: eliminated from both ends, VBA_OPT_CHUNK_WAVES)
    int asm_rows;                   // full-phase assembly in uniform passes (vba_asm_fast.h; VBA_OPT_FUSION bit 3)
    int fuse_walk;                  // batched mode: the sequential walk forms the blocks itself (VBA_OPT_FUSION bit 2)
    int warm_shift;                 // log2 of the bit-pattern width of a warm bin
    int warm_force_miss;            // test knob: every warm select reports a miss (exercises the repeat with the exact digits)
    int fused_trial;                // k_trial forms the step itself (0 no, 1 landmark-only 6x6 solve, 2 recovery of the partitioned solve)
    int64_t m_max;
    int64_t obs_stride;             // doubles between the observation blocks of consecutive windows (see ox)
    int nblk_obs;                   // ceil(m_max / kObsBlock)
    const int* n;                   // [W]
    const int* m;                   // [W]
    StepParams prm;                 // per-call constants, travel with the kernel arguments
    StepParams prev;                // those of the call in front (fold: its accept test is evaluated with them)
    WinHead* host_head;             // [W] mapped pinned host memory: k_decide publishes the outcome here
    double* host_states;            // [2 (call parity)][n_max][10] mapped pinned host memory or null (one-window handles): the trial kernel
                                    // that forms the trial states also writes them here -- the result of the call if the trial is accepted
    WinScalars* sc;                 // [W]
    // observations, pose sorted, SoA.  One contiguous block per window -- [ox | oy | oz | ou | ov | oconf] of m_pad doubles
    // each, then opose (m_pad ints) and the CSR pose_ptr (n_max + 1 ints) -- so that a window is uploaded with ONE copy;
    // the pointers are those of window 0, window w adds w * obs_stride doubles (2 * obs_stride ints)
    const double *ox, *oy, *oz, *ou, *ov, *oconf;
    const int* opose;
    const int* pose_ptr;
    // per pose [W][n_max][...]
    double* states;                 // [10] current estimate (input of the step)
    double* states_new;             // [10] last trial
    double* states_prev;            // [10] copy of the step's input (debug)
    const double* intr;             // [4]
    const double* cumrot;           // [4]
    const int* steps;               // RK4 steps to the next pose (last = 1)
    // long gaps (vba_long.hip): the poses i of this window whose edge i -> i + 1 spans more than kLongGap steps, in pose order
    const int* long_idx;            // [W][kLongCap]
    const int* n_long;              // [W]
    int nblk_long;                  // largest n_long over the windows of the handle = extra blocks / slots per window; 0 with the hop integrator
    // carried chunk states of the long edges: the trial kernel's propagation of an edge passes through the states the next call's
    // factor needs (an accepted trial is evaluated at exactly the states the next call starts from -- the carried-keys argument,
    // applied to dynamics): [W][2 (parity of the call that READS)][kLongPool][6]; an edge's slot = three states of header (x_hat,
    // the start state the chain belongs to, which the reader checks, one spare), its G sub-chunk start states, and room for the
    // eight partial products of its transition matrix, at long_off[w][k] (-1: no room, the factor finds its own)
    double* long_pool;
    const int* long_off;            // [W][kLongCap]
    int long_pool_cap;              // states per window and parity (kLongPool, or kLongPoolFew for handles of few windows)
    // BA_reg (BA_filtering.py:100-210): per-pose prior, active when reg != 0 and the call is not landmark-only
    const double* prior_H;          // [36] hessian_state_t
    const double* prior_x;          // [6]  prior position, velocity
    int reg;
    // work
    double* absr;                   // [W][2 m_max]  |r| components
    double* wraw;                   // [W][m_max]    raw robust weight
    double* ckeys;                  // [W][2 m_max]  keys surviving the first two select digits (usually a handful)
    int acc_lanes;                  // lanes per pose in k_obs_accumulate (4..64)
    double* part_init;              // [W][nblk_obs] block sums of |r_obs|
    double* part_trial;             // [W][trial_stride]: nblk_obs observation blocks, then nblk_dyn pose-chain blocks, then nblk_long long edges
    double* part_next;              // [W][nblk_obs] block sums of |r_obs| at the trial states (carried keys)
    double* part_pred;              // [W][2 (call parity)][pred_stride] block sums of |r_pred| at the input states (dynamics factor, 32 poses per
                                    // block: nblk_pred of them, then nblk_long slots of the long edges)
    double* part_prior;             // [W][2][pred_stride] block sums of |r_prior| at the input states (BA_reg; the long slots stay unused)
    int nblk_pred;
    int pred_stride;                // nblk_pred + kLongCap
    double* lastD;                  // [W][81] undamped diagonal block of the last pose (BA_filtering.py:97: last_hessian)
    // Carried keys: the trial residual of an accepted trial is evaluated at exactly the states the next call starts
    // from, so k_trial<true> (emit) also leaves that call's |r| keys, their exponent histogram and sum |r| behind and
    // the next call (carry) starts at the select without re-reading the observations.
    int emit, carry;                // emit: 0 no, 1 keys + exponent histogram, 2 keys + warm histogram; carry: 0 no, else the kind that was emitted
    int dyn_in_acc;                 // full-phase call with few windows: k_obs_accumulate's grid also runs the dynamics factor
    unsigned* hist;                 // [W][kHistStride]: digit 0 for parity 0, digit 0 for parity 1, digits 1..5
    double* Hraw;                   // [21]
    double* braw;                   // [6]
    double* xhat;                   // [6]
    double* Phi;                    // [36]
    double* rorb;                   // [6]
    double* fatt;                   // [1]
    double* qgrad;                  // [3]
    double* Hd;                     // [9]
    double* Hu;                     // [9]
    double* Hl;                     // [9]
    double* bands;                  // [3][81]
    double* rhs;                    // [9]
    double* Xs;                     // [81]  D'^-1 U of the forward sweep
    double* zs;                     // [9]
    double* dpose;                  // [9]
    // partitioned solve: chunk size (0 = one wave walks the whole chain), separators per window <= p_max
    int chunk, p_max;
    int chunk2;                     // > 0: the reduced system over the separators is partitioned again with this chunk size
    int pack;                       // sequential driver: 1 = three windows per wavefront (needs equal pose counts)
    int hop;                        // orbit integrator: 0 = 1 s RK4 steps (reference CPU branch), 1 = <=100 s hops (predict_gpu)
    int pivot;                      // which solver variants are launched: 0 unpivoted (checked) only, 1 pivoted only for every window, 2 both (per-window sticky choice)
    double* csol;                   // [W][n_max][19][9]  chunk solutions for the 19 right-hand sides
    double *cL, *cR;                // [W][p_max][19][9]  L_j / U_j times the neighbouring chunk solutions
    double *rXs, *rzs, *rx;         // reduced system over the separators, [W][p_max][81 | 9 | 9]
    double *csol2, *cL2, *cR2, *rx2;  // second level: chunk solutions over the level-1 separators, [W][p_max][...]
    int nblk_dyn;                   // pose-chain blocks of k_trial: ceil(n_max / 256), or ceil((n_max - 1) / 15) in the 16-lanes-per-pose geometry
    int trial_stride;               // doubles between the windows of part_trial (room for either geometry)
    // sharded mode: number of observation rows over all ranks (0 = not sharded) and external key buffer
    int64_t m_total;
    const double* abs_all;          // gathered |r| of all ranks or nullptr
    int64_t abs_all_count;
    // sharded mode, carried-keys protocol (vba_sh_run_schedule): what a rank exchanges lives where the kernels write it anyway
    unsigned* hist0_ext[2];         // digit-0 / warm histogram of that call parity inside the exchange buffer (null: the handle's own slot)
    const double* sel_slots;        // the gathered buckets of the median's bin, one slot per rank: [count, keys ...] (null: the list V.ckeys)
    int sel_nslots, sel_slot_stride;
    int trial_tiles;                // tiles of 256 rows per observation block of the plain latency-mode trial kernel (1, 2, 4; see k_trial)
    unsigned long long* wmax_ext;   // where the accumulation enters its maximum raw weight (null: WinScalars::wmax_bits of the call's parity)
};

// Speculatively chained calls (vba_run_schedule): the kernels of call c are enqueued before call c-1 is known to
// have finished with its first LM trial; a window that needs more trials (or a pivoted repeat) simply does not
// advance its call counter, and every later kernel leaves it alone until the host has finished that call.
// V.redo: 0 = ordinary launch (windows whose warm select missed wait), 1 = the repeat of such a call (only they run),
// 2 = a trial round both kinds take part in.
#define VBA_WINDOW_RUNS(V, w) (((V).call < 0 || (V).sc[(w)].call_idx == (V).call) && ((V).redo == 2 || ((V).sc[(w)].miss != 0) == ((V).redo != 0)))
#define VBA_SKIP_CALL(V, w) do { if (!VBA_WINDOW_RUNS(V, w)) return; } while (0)

__device__ __forceinline__ unsigned* hist0_of(const DevView& V, int w, int par) {
    unsigned* own = V.hist + (size_t)w * kHistStride + (size_t)par * kSelBins;
    unsigned* ext = V.hist0_ext[par];
    return ext ? ext : own;
}
__device__ __forceinline__ unsigned* histd_of(const DevView& V, int w, int digit /*1..5*/) { return V.hist + (size_t)w * kHistStride + (size_t)(digit + 1) * kSelBins; }

// ------------------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, kWave); }
// The same exchange for a mask known at compile time: partners one or two lanes apart sit in the same quad, and a quad
// permutation is a DPP move per 32 bits (~8 cycles) where the general shuffle is a round trip through the LDS crossbar
// (ds_bpermute).  Same partner, same value: same bits.
template <int MASK>
__device__ __forceinline__ double shfl_xor_f64_c(double v) {
#ifndef VBA_SHFL_BPERMUTE
    if constexpr (MASK == 1 || MASK == 2) {
        constexpr int ctrl = MASK == 1 ? 0xB1 : 0x4E;       // quad_perm [1,0,3,2] / [2,3,0,1]
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, ctrl, 0xf, 0xf, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), ctrl, 0xf, 0xf, false);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    if constexpr (MASK == 4 || MASK == 8) {
        // partners four / eight lanes apart sit in the same row of 16: a row shift to the left for the lanes whose bit is
        // clear, to the right for the others -- two DPP moves into one register, each writing its own banks of four lanes
        constexpr int shl = 0x100 + MASK, shr = 0x110 + MASK;
        constexpr int lo_banks = MASK == 4 ? 0x5 : 0x3, hi_banks = MASK == 4 ? 0xA : 0xC;
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const int x0 = (int)(unsigned)b, x1 = (int)(unsigned)(b >> 32);
        int r0 = __builtin_amdgcn_update_dpp(0, x0, shl, 0xf, lo_banks, false);
        r0 = __builtin_amdgcn_update_dpp(r0, x0, shr, 0xf, hi_banks, false);
        int r1 = __builtin_amdgcn_update_dpp(0, x1, shl, 0xf, lo_banks, false);
        r1 = __builtin_amdgcn_update_dpp(r1, x1, shr, 0xf, hi_banks, false);
        return __longlong_as_double((long long)(((unsigned long long)(unsigned)r1 << 32) | (unsigned)r0));
    }
    if constexpr (MASK == 16 || MASK == 32) {
        // partners in another row / the other half of the wavefront: gfx950's lane-swap instructions.  With both operands
        // the same value, v_permlane32_swap leaves (low half, low half) in the first result and (high half, high half) in
        // the second -- a lane's partner value is in the second if it sits in the low half, in the first otherwise;
        // v_permlane16_swap does the same with rows 0 / 2 against rows 1 / 3
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned x0 = (unsigned)b, x1 = (unsigned)(b >> 32);
        const bool low = (__lane_id() & MASK) == 0;
        unsigned r0, r1;
        if constexpr (MASK == 32) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(x0, x0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(x1, x1, false, false);
            r0 = low ? s0[1] : s0[0];
            r1 = low ? s1[1] : s1[0];
        } else {
            const auto s0 = __builtin_amdgcn_permlane16_swap(x0, x0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(x1, x1, false, false);
            r0 = low ? s0[1] : s0[0];
            r1 = low ? s1[1] : s1[0];
        }
        return __longlong_as_double((long long)(((unsigned long long)r1 << 32) | r0));
    }
#endif
    return __shfl_xor(v, MASK, kWave);
}

// Inclusive prefix sum over the wavefront (integers: any order is exact).  Four row shifts inside each row of 16 lanes, then
// the last lane of a row broadcast into the next row (row_bcast15, rows 1 and 3) and lane 31 into the upper half
// (row_bcast31): six DPP additions where the __shfl_up form is six round trips through the LDS crossbar.
__device__ __forceinline__ unsigned wave_inclusive_scan_u32(unsigned v) {
#ifndef VBA_SHFL_BPERMUTE
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1 (a lane without a source keeps the 0)
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast31 into rows 2 and 3
    return v;
#else
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(v, o, kWave);
        if (lane >= o) v += t;
    }
    return v;
#endif
}

// (six dependent exchanges: as LDS-crossbar shuffles ~0.4 us per call of either; same partners in the same order through
// shfl_xor_f64_c, so the same bits)
__device__ __forceinline__ double wave_sum(double v) {
    v += shfl_xor_f64_c<32>(v); v += shfl_xor_f64_c<16>(v); v += shfl_xor_f64_c<8>(v);
    v += shfl_xor_f64_c<4>(v); v += shfl_xor_f64_c<2>(v); v += shfl_xor_f64_c<1>(v);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, shfl_xor_f64_c<32>(v)); v = fmax(v, shfl_xor_f64_c<16>(v)); v = fmax(v, shfl_xor_f64_c<8>(v));
    v = fmax(v, shfl_xor_f64_c<4>(v)); v = fmax(v, shfl_xor_f64_c<2>(v)); v = fmax(v, shfl_xor_f64_c<1>(v));
    return v;
}

// fixed-order block sum (result valid in thread 0); BLOCK a multiple of 64, <= 1024
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v, double* lds /*[BLOCK/64]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) lds[wv] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < BLOCK / 64; ++i) t += lds[i];
    }
    __syncthreads();
    return t;
}

// (bcast_row16 below needs these first)
__device__ __forceinline__ unsigned long long f64_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ double bits_f64(unsigned long long b) { return __longlong_as_double((long long)b); }

// Lane K of this lane's ROW of 16 lanes, to every lane of the row: one DPP move per 32 bits (row_newbcast, gfx90a and later;
// control 0x150 + K) -- no LDS crossbar round trip (ds_bpermute) and no scalar register (v_readlane serves one row only).
template <int K>
__device__ __forceinline__ double bcast_row16(double v) {       // lane K of this lane's row of 16
#ifdef VBA_DPP32
    const unsigned long long b = f64_bits(v);
    // (mov_dpp: no `old` operand -- every lane is written, and update_dpp(0, ...) costs a v_mov of the zero per use)
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, 0x150 + K, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), 0x150 + K, 0xf, 0xf, false);
    return bits_f64(((unsigned long long)hi << 32) | lo);
#else
    // ONE v_mov_b64_dpp (row_newbcast is the one control the 64-bit DPP moves of gfx90a and later take)
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, false);
#endif
}

// Rank-1 update of a Gauss-Jordan pivot with the broadcast folded INTO the multiply-add: for the eight rows r != K
//   b_r -= a_r[lane K of the row] * bk ;  a_r -= a_r[lane K of the row] * ak      (fused, the operations of fma(-f, x, y))
// as v_fmac_f64_dpp ... row_newbcast:K -- no separate broadcast of the multiplier column (16 instead of 8 + 16 instructions
// per pivot with 64-bit moves, 16 + 16 with 32-bit ones).  The multiplier of BOTH updates is a_r as the pivot lane holds it
// BEFORE the pivot: the b update comes first, and the a update reads its own destination through the DPP -- an instruction
// reads all its sources before it writes (the idiom of every DPP reduction).  One asm statement per pivot so that the order
// inside is fixed; the s_nop covers the two wait states a DPP read needs behind a VALU write of the same register, should
// the compiler have placed one right in front (it knows the hazard for its own instructions, not for these).
template <int K>
__device__ __forceinline__ void dpp_rank1_rows16(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& a7,
                                                 double& b0, double& b1, double& b2, double& b3, double& b4, double& b5, double& b6, double& b7,
                                                 double ak, double bk) {
#define VBA_R1(A, B) "v_fmac_f64_dpp " B ", -" A ", %17 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t" \
                     "v_fmac_f64_dpp " A ", -" A ", %16 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" VBA_R1("%0", "%8") VBA_R1("%1", "%9") VBA_R1("%2", "%10") VBA_R1("%3", "%11") VBA_R1("%4", "%12") VBA_R1("%5", "%13")
        VBA_R1("%6", "%14") VBA_R1("%7", "%15")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4),
          "+v"(b5), "+v"(b6), "+v"(b7)
        : "v"(ak), "v"(bk), "n"(K));
#undef VBA_R1
}
// ... the same for ONE column per lane: a_r -= a_r[lane K of the row] * ak for the eight rows r != K
template <int K>
__device__ __forceinline__ void dpp_rank1_rows8(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& a7, double ak) {
#define VBA_R1(A) "v_fmac_f64_dpp " A ", -" A ", %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t" VBA_R1("%0") VBA_R1("%1") VBA_R1("%2") VBA_R1("%3") VBA_R1("%4") VBA_R1("%5") VBA_R1("%6") VBA_R1("%7")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
        : "v"(ak), "n"(K));
#undef VBA_R1
}
// the eight rows of a 9-vector that are not row K, as references
template <int K, int I>
__device__ __forceinline__ double& row_not(double (&v)[9]) { return v[I < K ? I : I + 1]; }
template <int K>
__device__ __forceinline__ void dpp_rank1_9(double (&A)[9]) {
    dpp_rank1_rows8<K>(row_not<K, 0>(A), row_not<K, 1>(A), row_not<K, 2>(A), row_not<K, 3>(A), row_not<K, 4>(A), row_not<K, 5>(A), row_not<K, 6>(A),
                       row_not<K, 7>(A), A[K]);
}
template <int K>
__device__ __forceinline__ void dpp_rank1_9(double (&A)[9], double (&B)[9]) {
    dpp_rank1_rows16<K>(row_not<K, 0>(A), row_not<K, 1>(A), row_not<K, 2>(A), row_not<K, 3>(A), row_not<K, 4>(A), row_not<K, 5>(A), row_not<K, 6>(A),
                        row_not<K, 7>(A), row_not<K, 0>(B), row_not<K, 1>(B), row_not<K, 2>(B), row_not<K, 3>(B), row_not<K, 4>(B), row_not<K, 5>(B),
                        row_not<K, 6>(B), row_not<K, 7>(B), A[K], B[K]);
}
template <int K>
__device__ __forceinline__ int bcast_row16_i32(int v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, false); }


// Resolve one radix-select digit: given the histogram of digit p among keys matching the prefix and the
// rank wanted inside that set, every thread of the block gets (new prefix, new rank).  256 threads.
// this thread's bins of a histogram of nbins (t * per + j, j < per = nbins / 256): loaded apart from the resolve so that a
// caller can have them in flight while it does something else
__device__ __forceinline__ void select_load(const unsigned* hist, int nbins, unsigned (&loc)[8]) {
    const int t = threadIdx.x;
    const int per = nbins / 256 > 0 ? nbins / 256 : 1;      // bins per thread (8, 4 or 2)
#pragma unroll
    for (int j = 0; j < 8; ++j) loc[j] = (j < per && t * per + j < nbins) ? hist[t * per + j] : 0u;
}

// count_out (optional): the count of the bin that was found
__device__ __forceinline__ void select_resolve_loaded(const unsigned (&loc)[8], int nbins, int width, unsigned long long prefix,
                                                      long long rank, unsigned long long& prefix_out, long long& rank_out,
                                                      unsigned* lds_u /*[260]*/, unsigned* count_out = nullptr) {
    const int t = threadIdx.x;
    const int per = nbins / 256 > 0 ? nbins / 256 : 1;
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += loc[j];
    // inclusive scan over 256 threads: wave scan + 4 wave totals
    const unsigned inc = wave_inclusive_scan_u32(s);
    if ((t & 63) == 63) lds_u[t >> 6] = inc;
    if (t == 0) { lds_u[8] = 0; lds_u[9] = 0; lds_u[10] = 0; }
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += lds_u[w];
    const long long excl = (long long)base + inc - s;
    if (rank >= excl && rank < excl + (long long)s) {
        long long cum = excl;
        int bin = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < per) {
                if (rank >= cum && rank < cum + (long long)loc[j]) bin = t * per + j, lds_u[8] = (unsigned)(rank - cum), lds_u[9] = (unsigned)bin, lds_u[10] = loc[j];
                cum += loc[j];
            }
        }
    }
    __syncthreads();
    rank_out = (long long)lds_u[8];
    prefix_out = (prefix << width) | (unsigned long long)lds_u[9];
    if (count_out) *count_out = lds_u[10];
    __syncthreads();
}

__device__ __forceinline__ void select_resolve(const unsigned* hist, int nbins, int width, unsigned long long prefix,
                                               long long rank, unsigned long long& prefix_out, long long& rank_out,
                                               unsigned* lds_u /*[260]*/) {
    unsigned loc[8];
    select_load(hist, nbins, loc);
    select_resolve_loaded(loc, nbins, width, prefix, rank, prefix_out, rank_out, lds_u);
}

}  // namespace vba
