// vba_asm_fast.h -- the 252 entries of one pose row of the block-tridiagonal system (BA_filtering.py:40-48) formed by ONE
// wave in seven passes of a single kind each: sub-diagonal (2 passes of 64 entries), diagonal (2), super-diagonal (2),
// right-hand side (1).
//
// band_entry / rhs_entry (vba_math.h) are written per entry: which band, which row and column decide the code path, and a
// wave whose lanes hold different kinds of entries executes every path (~450 instructions per pose and wave in
// k_assemble).  Here a pass is uniform: what differs between its lanes (rotation row or not, diagonal or not, which Phi
// column) is data -- offsets and masks decoded once per kernel (AsmLanes) -- and the arithmetic per entry is the SAME
// expression, operation for operation, as in band_entry / rhs_entry, so the system is the same to the bit (tests: uniform
// vs per-entry assembly, every band entry).
// Measured (MI355X): one window, 0.8 us off the average call (the per-entry form has every thread form its entry of four
// poses in turn, ~600 serial instructions); 4096 windows, k_assemble_rows 1.77 ms against k_assemble's 1.85 ms -- there the
// assembly is not bound by its instruction count.  Default (VBA_OPT_FUSION bit 3); the block-forming walk built on the
// same rows (bit 2) stays slower than assembly + walk and is opt-in.  Both bit-exact and under test.
//
// Slot layout of the staged inputs of a pose (vba_asm.h): Hraw 0, braw 21, Phi 27, rorb 63, qgrad 69, Hd 72, Hu 81, Hl 90,
// prior H 99, prior r 135.
#pragma once

#include "vba_asm.h"
#include "vba_device.h"

namespace vba {

// What differs between the lanes of a pass, packed into one word per pass (seven registers per lane for the whole row;
// decoded with bit-field extracts where it is used -- the walk that forms its blocks while it eliminates has no registers
// to spare for descriptors).
//   simple (sub / super-diagonal):  off 8 | kind 2 (0 zero, 1 orbit factor, 2 attitude term) | fsel 2 | vel 1
//   diag:   offH 5 | pa 3 (7 = rotation) | pb 3 | hd 4 (Hd entry, rot) | ra 3 | rb 3 | hh 1 | diagF 1 | rot 1 | pr 1 | vel 1
//   rhs:    a 4 | pa 3 | ra 3 (7 = rotation) | live 1 | vel 1
// fsel: F_val of the entry's row / column: 0 -> -1, 1 -> -kVelCoeff, 2 -> 1 (attitude term), 3 -> 0.   vel: D of the Phi row is kVelCoeff.
struct AsmLanes {
    unsigned L[2], U[2], D[2], g;
};

__device__ __forceinline__ int asm_pcol(int c) { return c < 3 ? c : c - 3; }      // column of Phi that state slot c (not a rotation) maps to
__device__ __forceinline__ bool asm_rot(int c) { return c >= 3 && c < 6; }
__device__ __forceinline__ unsigned asm_fsel(int c) { return c < 3 ? 0u : 1u; }    // F_val(c) = -1 | -kVelCoeff
__device__ __forceinline__ double asm_fval(unsigned fsel) { return fsel == 0u ? -1.0 : (fsel == 1u ? -kVelCoeff : (fsel == 2u ? 1.0 : 0.0)); }

__device__ __forceinline__ unsigned asm_pack_simple(int off, int kind, unsigned fsel, bool vel) {
    return (unsigned)off | ((unsigned)kind << 8) | (fsel << 10) | ((vel ? 1u : 0u) << 12);
}

__device__ __forceinline__ AsmLanes asm_lanes(int lane) {
    AsmLanes g;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = lane + 64 * q;
        const int a = e < 81 ? e / 9 : 0, b = e < 81 ? e % 9 : 0;
        const bool in = e < 81, rotA = asm_rot(a), rotB = asm_rot(b);
        const int ra = F_row(a), rb = F_row(b);
        // sub-diagonal (band_entry which == 0): (F_val(a) sigma) E(Phi_{i-1}, F_row(a), b)  |  sigma Hl
        unsigned l = asm_pack_simple(0, 0, 3u, false);
        if (in && ra >= 0 && !rotB) l = asm_pack_simple(27 + 6 * ra + asm_pcol(b), 1, asm_fsel(a), ra >= 3);
        else if (in && rotA && rotB) l = asm_pack_simple(90 + 3 * (a - 3) + (b - 3), 2, 2u, false);
        g.L[q] = l;
        // super-diagonal (which == 2): (E(Phi_i, F_row(b), a) sigma) F_val(b)  |  sigma Hu
        unsigned u = asm_pack_simple(0, 0, 3u, false);
        if (in && rb >= 0 && !rotA) u = asm_pack_simple(27 + 6 * rb + asm_pcol(a), 1, asm_fsel(b), rb >= 3);
        else if (in && rotA && rotB) u = asm_pack_simple(81 + 3 * (a - 3) + (b - 3), 2, 2u, false);
        g.U[q] = u;
        const bool hh = in && a < 6 && b < 6, rot = in && rotA && rotB;
        unsigned d = hh ? (unsigned)sym6(a, b) : 0u;
        d |= (unsigned)(rotA ? 7 : asm_pcol(a)) << 5;
        d |= (unsigned)(rotB ? 7 : asm_pcol(b)) << 8;
        d |= (unsigned)(rot ? 3 * (a - 3) + (b - 3) : 0) << 11;
        d |= (unsigned)(ra >= 0 ? ra : 0) << 15;
        d |= (unsigned)(rb >= 0 ? rb : 0) << 18;
        d |= (hh ? 1u : 0u) << 21;
        d |= ((in && a == b && ra >= 0) ? 1u : 0u) << 22;
        d |= (rot ? 1u : 0u) << 23;
        d |= ((in && ra >= 0 && rb >= 0) ? 1u : 0u) << 24;
        d |= asm_fsel(a) << 25;
        g.D[q] = d;
    }
    const int a = lane < 9 ? lane : 0;
    unsigned r = (unsigned)a;
    r |= (unsigned)(asm_rot(a) ? 7 : asm_pcol(a)) << 4;
    r |= (unsigned)(F_row(a) >= 0 ? F_row(a) : 7) << 7;
    r |= (lane < 9 ? 1u : 0u) << 10;
    r |= asm_fsel(a) << 11;
    g.g = r;
    return g;
}

// me / pv: staged inputs of pose i and of pose i - 1; has_next: i < n - 1 (Phi_i exists), has_prev: i > 0.
// store(e, v) receives entry e in [0, 252) (sub | diag | super row major, then the right-hand side); every lane calls it
// for the entries of its passes only.
template <bool REG, class Store>
__device__ __forceinline__ void asm_form_row(const AsmLanes& g, const double* me, const double* pv, bool has_next, bool has_prev,
                                             double sigma, double inv_wmax, int lane, Store&& store) {
    const bool dyn = sigma != 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const bool in = lane + 64 * q < 81;
        {   // sub-diagonal: band_entry which == 0
            const unsigned d = g.L[q];
            const int off = d & 255u, kind = (d >> 8) & 3u;
            const double f = asm_fval((d >> 10) & 3u), B = ((d >> 12) & 1u) ? kVelCoeff : 1.0;
            const double x = (kind == 1 ? pv : me)[off];
            const double t1 = vba_mul(vba_mul(f, sigma), vba_mul(B, x));     // (F_val sigma) * (D Phi)
            const double t2 = vba_mul(sigma, x);                             // sigma * Hl
            const double v = (dyn && has_prev && kind != 0) ? (kind == 1 ? t1 : t2) : 0.0;
            if (in) store(lane + 64 * q, v);
        }
        {   // super-diagonal: which == 2
            const unsigned d = g.U[q];
            const int off = d & 255u, kind = (d >> 8) & 3u;
            const double f = asm_fval((d >> 10) & 3u), B = ((d >> 12) & 1u) ? kVelCoeff : 1.0;
            const double x = me[off];
            const double t1 = vba_mul(vba_mul(vba_mul(B, x), sigma), f);     // (E sigma) * F_val
            const double t2 = vba_mul(sigma, x);                             // sigma * Hu
            const double v = (dyn && has_next && kind != 0) ? (kind == 1 ? t1 : t2) : 0.0;
            if (in) store(162 + lane + 64 * q, v);
        }
        {   // diagonal: which == 1
            const unsigned d = g.D[q];
            const int offH = d & 31u, pa = (d >> 5) & 7u, pb = (d >> 8) & 7u, hd = (d >> 11) & 15u, ra = (d >> 15) & 7u, rb = (d >> 18) & 7u;
            const bool hh = (d >> 21) & 1u, diagF = (d >> 22) & 1u, rot = (d >> 23) & 1u, pr = (d >> 24) & 1u;
            const bool rotA = pa == 7, rotB = pb == 7;
            const double fv = ((d >> 25) & 1u) ? -kVelCoeff : -1.0;
            double v = 0.0;
            {
                const double h = vba_mul(me[offH], inv_wmax);
                v = hh ? h : v;
            }
            if (dyn) {
                if (has_next) {
                    double s = 0.0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const double Dr = r < 3 ? 1.0 : kVelCoeff;
                        const double xa = me[27 + 6 * r + (rotA ? 0 : pa)], xb = me[27 + 6 * r + (rotB ? 0 : pb)];
                        const double Ea = rotA ? 0.0 : vba_mul(Dr, xa), Eb = rotB ? 0.0 : vba_mul(Dr, xb);
                        s = fma(vba_mul(Ea, sigma), Eb, s);
                    }
                    v = vba_add(v, s);
                }
                {
                    const double t = fma(vba_mul(fv, sigma), fv, v);
                    v = (has_prev && diagF) ? t : v;
                }
                {
                    const double t = fma(sigma, me[72 + hd], v);
                    v = rot ? t : v;
                }
            }
            if (REG) {
                const double* H = me + 99;
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) s = fma(H[k * 6 + ra], H[k * 6 + rb], s);
                const double t = vba_add(v, s);
                v = pr ? t : v;
            }
            if (in) store(81 + lane + 64 * q, v);
        }
    }
    {   // right-hand side (rhs_entry)
        const unsigned d = g.g;
        const int a = d & 15u, pa = (d >> 4) & 7u, ra = (d >> 7) & 7u;
        const bool live = (d >> 10) & 1u, rotA = pa == 7;
        const double fv = ((d >> 11) & 1u) ? -kVelCoeff : -1.0;
        double v = 0.0;
        {
            const double h = vba_mul(me[21 + (a < 6 ? a : 0)], inv_wmax);
            v = a < 6 ? h : v;
        }
        if (dyn) {
            if (has_next) {
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const double Dr = r < 3 ? 1.0 : kVelCoeff;
                    const double xa = me[27 + 6 * r + (rotA ? 0 : pa)];
                    const double Ea = rotA ? 0.0 : vba_mul(Dr, xa);
                    s = fma(vba_mul(Ea, sigma), me[63 + r], s);
                }
                v = vba_add(v, -s);
            }
            {
                const double t = fma(vba_mul(fv, sigma), -pv[63 + (rotA ? 0 : ra)], v);
                v = (has_prev && !rotA) ? t : v;
            }
            {
                const double t = fma(sigma, -me[69 + (rotA ? a - 3 : 0)], v);
                v = rotA ? t : v;
            }
        }
        if (REG) {
            const double* H = me + 99;
            const double* r6 = me + 135;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s = fma(H[k * 6 + (rotA ? 0 : ra)], r6[k], s);
            const double t = vba_add(v, s);
            v = !rotA ? t : v;
        }
        if (live) store(243 + lane, v);
    }
}

// ---------------------------------------------------------------------------------------------- column-wise formation
// The same pose row formed by COLUMN: lane c of a row of 16 lanes gets column c of the diagonal block D_j (c < 9; lane 9: the
// right-hand side), of the super-diagonal block U_j and of the sub-diagonal block L_j in registers -- the layout in which the
// four-windows-per-wave walk (k_solve_quad) and the chunk elimination that forms its own blocks hold them.  Entry by
// entry the operations of band_entry / rhs_entry (vba_math.h) in their order, so the same system to the bit; what differs
// between the lanes (rotation column or not, which Phi column, right-hand side) is data.  ~250 instructions for the 16
// lanes of a row, i.e. for FOUR pose rows per wave, where the uniform passes above cost ~400 per pose row.
// Every read of the staged inputs is UNCONDITIONAL: the address is selected, one load is made, the value is selected
// (`cond ? lds[i] : 0` compiles to a masked load in a basic block of its own with a full wait behind it; a few dozen of
// those per block were most of the walk's time at one wave per SIMD).
struct AsmColLane {
    bool is_col, is_rhs, rotc, nonrot;
    int pcl;            // column of Phi / row of F that this lane's state slot maps to (0 for a rotation slot)
    int crl;            // rotation slot: its index 0..2
    double fvc, Dcl;    // F_val of the lane's column; D of Phi row pcl
};

__device__ __forceinline__ AsmColLane asm_col_lane(int c) {
    AsmColLane L;
    L.is_col = c < 9;
    L.is_rhs = c == 9;
    L.rotc = c >= 3 && c < 6;
    L.nonrot = L.is_col && !L.rotc;
    L.pcl = c < 3 ? c : (L.nonrot ? c - 3 : 0);
    L.crl = L.rotc ? c - 3 : 0;
    L.fvc = c < 3 ? -1.0 : -kVelCoeff;
    L.Dcl = L.pcl < 3 ? 1.0 : kVelCoeff;
    return L;
}

// me / pv: staged inputs of pose j and of pose j - 1 (vba_asm.h layout); live: the pose exists (else zeros); A: column of
// [D_j | rhs_j] WITHOUT damping; B: column of U_j; Lc: column of L_j (zeros in lanes without a column).
template <bool REG>
__device__ __forceinline__ void asm_form_columns(const AsmColLane& L, int c, const double* me, const double* pv, bool live, bool has_next,
                                                 bool has_prev, double sigma, double iw, double (&A)[9], double (&B)[9], double (&Lc)[9]) {
    const bool is_col = L.is_col, is_rhs = L.is_rhs, rotc = L.rotc, nonrot = L.nonrot;
    const int pcl = L.pcl, crl = L.crl;
    const double fs[2] = {vba_mul(-1.0, sigma), vba_mul(-kVelCoeff, sigma)};
    // the lane's second factor of the J_f^T Sigma J_f sums: its column of E_j = D Phi_j, or r_orb (right-hand side)
    double X[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double Dr = r < 3 ? 1.0 : kVelCoeff;
        const double xv = me[is_rhs ? 63 + r : 27 + 6 * r + pcl];
        const double e = vba_mul(Dr, xv);
        X[r] = nonrot ? e : (is_rhs ? xv : 0.0);
    }
    double Xp[6];
    if (REG) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Xp[k] = me[is_rhs ? 135 + k : 99 + k * 6 + pcl];
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const bool rota = a >= 3 && a < 6;
        const int pa = a < 3 ? a : a - 3;           // (non-rotation a)
        double v = 0.0;
        if (a < 6) {
            const int idx = is_rhs ? 21 + a : sym6(a, c < 6 ? c : 0);
            const double h = vba_mul(me[idx], iw);
            v = (is_rhs || c < 6) ? h : 0.0;
        }
        {
            double sdyn = 0.0;
            if (!rota) {
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const double Dr = r < 3 ? 1.0 : kVelCoeff;
                    sdyn = fma(vba_mul(vba_mul(Dr, me[27 + 6 * r + pa]), sigma), X[r], sdyn);
                }
            }
            const double t = vba_add(v, is_rhs ? -sdyn : sdyn);
            v = has_next ? t : v;
        }
        if (!rota) {
            const double fsa = fs[a < 3 ? 0 : 1], fva = a < 3 ? -1.0 : -kVelCoeff;
            const double zr = pv[63 + pa];
            const double z = is_rhs ? -zr : fva;
            const double t = fma(fsa, z, v);
            v = (has_prev && (is_rhs || c == a)) ? t : v;
        } else {
            const double yv = me[is_rhs ? 69 + (a - 3) : 72 + 3 * (a - 3) + crl];
            const double t = fma(sigma, is_rhs ? -yv : yv, v);
            v = (is_rhs || rotc) ? t : v;
        }
        if (REG && !rota) {
            double sp = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) sp = fma(me[99 + k * 6 + pa], Xp[k], sp);
            const double t = vba_add(v, sp);
            v = (is_rhs || nonrot) ? t : v;
        }
        A[a] = (live && c < 10) ? v : 0.0;
        // super-diagonal column and sub-diagonal column
        double u, l;
        if (!rota) {
            const double fsa = fs[a < 3 ? 0 : 1];
            const double eu = vba_mul(L.Dcl, me[27 + 6 * pcl + pa]);                // E_entry(Phi_j, F_row(c), a)
            u = vba_mul(vba_mul(eu, sigma), L.fvc);
            const double Dra = pa < 3 ? 1.0 : kVelCoeff;
            const double el = vba_mul(Dra, pv[27 + 6 * pa + pcl]);                  // E_entry(Phi_{j-1}, F_row(a), c)
            l = vba_mul(fsa, el);
            u = nonrot ? u : 0.0;
            l = nonrot ? l : 0.0;
        } else {
            u = vba_mul(sigma, me[81 + 3 * (a - 3) + crl]);
            l = vba_mul(sigma, me[90 + 3 * (a - 3) + crl]);
            u = rotc ? u : 0.0;
            l = rotc ? l : 0.0;
        }
        B[a] = (has_next && is_col) ? u : 0.0;
        Lc[a] = (has_prev && is_col) ? l : 0.0;
    }
}

}  // namespace vba
