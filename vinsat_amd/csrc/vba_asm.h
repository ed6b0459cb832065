// vba_asm.h -- staging of the per-pose inputs of the block-tridiagonal assembly (BA_filtering.py:40-48) in LDS and
// the row descriptor built on them.  Shared by k_assemble (its own launch: batched windows, sharded mode, debug) and
// by the chunk elimination of the latency mode, which forms the blocks of its chunk itself.
#pragma once

#include "vba_device.h"

namespace vba {

constexpr int kAsmBase = 21 + 6 + 36 + 6 + 3 + 27;    // Hraw, braw, Phi, rorb, qgrad, Hd|Hu|Hl
constexpr int kAsmPrior = 36 + 6;                     // BA_reg: prior H, prior r

// input q (0 .. kAsmBase [+ kAsmPrior]) of pose pb = w * n_max + i, as the assembly stages it
template <bool REG>
__device__ __forceinline__ double asm_input(const DevView& V, size_t pb, int q, bool dyn) {
    if (q < 21) return V.Hraw[pb * 21 + q];
    if (q < 27) return V.braw[pb * 6 + (q - 21)];
    if (!dyn) return 0.0;
    if (q < 63) return V.Phi[pb * 36 + (q - 27)];
    if (q < 69) return V.rorb[pb * 6 + (q - 63)];
    if (q < 72) return V.qgrad[pb * 3 + (q - 69)];
    if (q < 81) return V.Hd[pb * 9 + (q - 72)];
    if (q < 90) return V.Hu[pb * 9 + (q - 81)];
    if (q < 99) return V.Hl[pb * 9 + (q - 90)];
    if (REG) {
        if (q < 135) return V.prior_H[pb * 36 + (q - 99)];
        // one component of r = H [p_prior - p ; v_prior - v]
        const double* Hr = V.prior_H + pb * 36 + (q - 135) * 6;
        const double* xp = V.prior_x + pb * 6;
        const double* st = V.states + pb * 10;
        // (explicit fma chain: every path that stages this value must round it alike, see vba_math.h)
        double r = vba_mul(Hr[0], xp[0] - st[0]);
        r = fma(Hr[1], xp[1] - st[1], r);
        r = fma(Hr[2], xp[2] - st[2], r);
        r = fma(Hr[3], xp[3] - st[7], r);
        r = fma(Hr[4], xp[4] - st[8], r);
        return fma(Hr[5], xp[5] - st[9], r);
    }
    return 0.0;
}

// The same value with the ADDRESS selected and one unconditional load (no per-element branch ladder: a thread that stages
// several inputs has all its loads in flight together).  REG's prior residual (q >= 135) is computed: that alone branches.
template <bool REG>
__device__ __forceinline__ double asm_input_nobranch(const DevView& V, size_t pb, int q) {
    const double* p = V.Hraw + pb * 21 + q;
    p = q >= 21 ? V.braw + pb * 6 + (q - 21) : p;
    p = q >= 27 ? V.Phi + pb * 36 + (q - 27) : p;
    p = q >= 63 ? V.rorb + pb * 6 + (q - 63) : p;
    p = q >= 69 ? V.qgrad + pb * 3 + (q - 69) : p;
    p = q >= 72 ? V.Hd + pb * 9 + (q - 72) : p;
    p = q >= 81 ? V.Hu + pb * 9 + (q - 81) : p;
    p = q >= 90 ? V.Hl + pb * 9 + (q - 90) : p;
    if (REG) {
        p = q >= 99 ? V.prior_H + pb * 36 + (q >= 135 ? 0 : q - 99) : p;
        if (q >= 135) return asm_input<REG>(V, pb, q, true);
    }
    return *p;
}

// slots [0, slots) of `in` receive the inputs of poses first .. first + slots - 1 (zeros outside [0, n)); every thread
// of the block takes part (stride = block size); the caller synchronises
template <bool REG>
__device__ __forceinline__ void asm_stage(const DevView& V, int w, int n, bool dyn, int first, int slots, double* in, int tid,
                                          int nthreads) {
    constexpr int kAsmIn = kAsmBase + (REG ? kAsmPrior : 0);
    const size_t sb = (size_t)w * V.n_max;
    for (int e = tid; e < slots * kAsmIn; e += nthreads) {
        const int slot = e / kAsmIn, q = e % kAsmIn;
        const int i = first + slot;
        in[e] = (i >= 0 && i < n) ? asm_input<REG>(V, sb + i, q, dyn) : 0.0;
    }
}

// The same with the slot count known at compile time: all loads of a thread are issued before its first store (the loop
// above waits for every load in turn -- one round trip per element and thread; the assembly kernels were bound by that).
template <bool REG, int SLOTS, int NTHREADS>
__device__ __forceinline__ void asm_stage_all(const DevView& V, int w, int n, bool dyn, int first, double* in, int tid) {
    constexpr int kAsmIn = kAsmBase + (REG ? kAsmPrior : 0);
    constexpr int kTotal = SLOTS * kAsmIn;
    constexpr int kIter = (kTotal + NTHREADS - 1) / NTHREADS;
    const size_t sb = (size_t)w * V.n_max;
    double v[kIter];
#pragma unroll
    for (int k = 0; k < kIter; ++k) {
        const int e = tid + k * NTHREADS;
        const int slot = e / kAsmIn, q = e % kAsmIn;
        const int i = first + slot;
        v[k] = (e < kTotal && i >= 0 && i < n) ? asm_input<REG>(V, sb + i, q, dyn) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kIter; ++k) {
        const int e = tid + k * NTHREADS;
        if (e < kTotal) in[e] = v[k];
    }
}

// row i of the system from the staged inputs of pose i (`me`) and pose i - 1 (`pv`)
template <bool REG>
__device__ __forceinline__ AsmRow asm_row(const double* me, const double* pv, int i, int n, bool dyn, double sigma, double inv_wmax) {
    AsmRow R;
    R.Hraw = me;
    R.braw = me + 21;
    R.inv_wmax = inv_wmax;
    R.sigma = dyn ? sigma : 0.0;
    R.Phi_i = (dyn && i < n - 1) ? me + 27 : nullptr;
    R.Phi_im1 = (dyn && i > 0) ? pv + 27 : nullptr;
    R.rorb_i = (dyn && i < n - 1) ? me + 63 : nullptr;
    R.rorb_im1 = (dyn && i > 0) ? pv + 63 : nullptr;
    R.qgrad = me + 69;
    R.Hd = me + 72;
    R.Hu = me + 81;
    R.Hl = me + 90;
    R.prior_H = REG ? me + 99 : nullptr;
    R.prior_r = me + 135;
    return R;
}

}  // namespace vba
