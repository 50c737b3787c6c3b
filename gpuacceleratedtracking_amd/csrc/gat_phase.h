// gat_phase.h -- the phase arithmetic every kernel of libgat shares (device code only).
//
// Bit-exact parity between the three correlator kernels (gat_kernels.hip, gat_mfma.hip, gat_mfma_bf16.hip), the
// stand-alone replica / signal generators and the CPU oracle rests on these few functions, so they exist ONCE.
// Files that include this header must be built with -ffp-contract=off (the double-precision code phase of the
// reference, src/algorithms.jl:179-182, is a multiply followed by an add, not an fma).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gat {

// exp(j*2*pi*theta) for theta in cycles (double).  Octant reduction in double (exact), float
// Taylor polynomials on |a| <= pi/4 (|err| < 3e-8), quadrant fix-up.  Carrier of src/algorithms.jl:172-174.
__device__ __forceinline__ void sincos_cycles(double theta, float &c, float &s)
{
    const double q = __builtin_rint(theta * 4.0);
    const double r = __builtin_fma(q, -0.25, theta); // exact: |r| <= 0.125 cycles
    const float a = (float)r * 6.283185307179586f;
    const float a2 = a * a;
    // (the leading term as multiply + add: an fma with TWO constants keeps one of them in a register, and the compiler
    // holds that register across every loop of a kernel that calls this function per segment)
    float sp = a2 * 2.7557319e-6f + -1.9841270e-4f;
    sp = __builtin_fmaf(a2, sp, 8.3333333e-3f);
    sp = __builtin_fmaf(a2, sp, -1.6666667e-1f);
    sp = __builtin_fmaf(a2 * a, sp, a);
    float cp = a2 * 2.4801587e-5f + -1.3888889e-3f;
    cp = __builtin_fmaf(a2, cp, 4.1666667e-2f);
    cp = __builtin_fmaf(a2, cp, -0.5f);
    cp = __builtin_fmaf(a2, cp, 1.0f);
    const int qi = (int)(long long)q & 3;
    const float cs = (qi & 1) ? sp : cp;
    const float sn = (qi & 1) ? cp : sp;
    c = (qi == 1 || qi == 2) ? -cs : cs;
    s = (qi >= 2) ? -sn : sn;
}

// floor(p) mod Lc with floored (Julia) semantics; valid for |ip| < 2^30 and |ip| / Lc < 2^21
// (checked on the host and again per workgroup by the correlator kernels; other callers clamp).
__device__ __forceinline__ int floormod_fast(int ip, int Lc, float inv_lc)
{
    const float q = __builtin_floorf((float)ip * inv_lc);
    int r = ip - (int)q * Lc;
    r += (r < 0) ? Lc : 0;
    r -= (r >= Lc) ? Lc : 0;
    return r;
}

// Parameters floormod_fast / chip_index cannot evaluate exactly: `reach` = the largest |sample + shift| of the call.  The
// correlator kernels, the stand-alone replica and the materialising debug kernel all use this ONE predicate (they write
// NaN instead of indexing outside the chip table); the host entry points with host-resident parameters reject the same set.
__device__ __forceinline__ bool code_span_bad(double ratio, double tau, double reach, int Lc)
{
    const double span = __builtin_fabs(tau) + __builtin_fabs(ratio) * reach + 1.0;
    return !(span < 1073741824.0) || !(span < 2097152.0 * (double)Lc) || !(ratio >= 0.0);
}

// The reference's code phase of sample x = n + shift, src/algorithms.jl:179: one double multiply and one double
// add, NOT fused (bit-identical to the CPU oracle).
__device__ __forceinline__ double code_phase(double ratio, double tau, int x)
{
    return __dadd_rn(__dmul_rn(ratio, (double)x), tau);
}

// chip index of sample x: floor(code phase) mod Lc, src/algorithms.jl:179-182.
__device__ __forceinline__ int chip_index(double ratio, double tau, int x, int Lc, float inv_lc)
{
    return floormod_fast((int)__builtin_floor(code_phase(ratio, tau, x)), Lc, inv_lc);
}

// ---------------------------------------------------------------------------------------------------------------
// Exact chip-index WALK over consecutive samples.
//
// floor(code_phase(x)) is needed for every sample; evaluating the double expression per sample costs ~10 vector
// instructions of which 5 are double-precision (2-4x issue cost).  The walk evaluates it ONCE (the anchor, exact)
// and advances a 32.32 fixed-point copy of the phase by the 32.32 chip rate for the following samples.  The fixed-
// point value differs from the reference's rounded double by at most
//     2 * 2^-52 * span   (two double roundings at the anchor and at the target, |values| <= span)
//   + (j + 1) * 2^-32    (truncation of the anchor's fraction and of the rate, j steps)
// so its integer part equals the reference's floor whenever its fraction is at least `margin` away from both 0 and
// 1.  A step that lands inside the margin is NOT trusted: the caller re-evaluates that sample exactly (in practice
// ~1e-8 of all samples at GNSS magnitudes).  Result: bit-identical chip edges at ~1/4 of the instruction cost.
// ---------------------------------------------------------------------------------------------------------------
struct ChipWalkConst {   // per (block, channel): uniform over the workgroup
    unsigned long long rate; // floor(ratio * 2^32): chips per sample in 32.32 fixed point (truncated)
    unsigned margin;         // in 2^-32 chips; a fraction within `margin` of 0 or 1 is ambiguous
    int exact_only;          // 1: the walk cannot be trusted at these magnitudes -> evaluate every sample exactly
};

// max_run: most samples an anchor is walked away from (margin); max_advance: most samples ONE step advances
// (fewer than Lc chips per step: one wrap of the table index at most).  span: the kernels' range bound
// |tau| + ratio * (N + max|shift|) + 1 (every |code phase| and |ratio * x| of the block is below it).
__device__ __forceinline__ ChipWalkConst chip_walk_setup(double ratio, double span, int max_run, int max_advance, int Lc)
{
    ChipWalkConst w;
    // two double roundings on each side, each <= 2^-53 * 2 span: 2 * 2^-52 * span * 2^32 = span * 2^-19; doubled for slack
    const double m = span * 0x1p-18 + (double)(max_run + 2);
    w.exact_only = !(ratio >= 0.0) || !(ratio * (double)(max_advance + 1) + 2.0 < (double)Lc) || !(m < 1.0e9) ||
                   !(ratio < 1.0e6);
    w.rate = w.exact_only ? 0ull : (unsigned long long)(ratio * 4294967296.0); // truncation: rate <= ratio * 2^32
    w.margin = w.exact_only ? 0u : (unsigned)m + 1u;
    return w;
}

struct ChipWalk {
    unsigned long long q; // 32.32 fixed point: floor(code phase) in the high word (two's complement), fraction low
    int base;             // floor(code phase) - table index at the current position: index = hi(q) - base
};

// exact anchor at sample x; returns its chip index
__device__ __forceinline__ int chip_walk_anchor(ChipWalk &w, double ratio, double tau, int x, int Lc, float inv_lc)
{
    const double p = code_phase(ratio, tau, x);
    const double fl = __builtin_floor(p);
    const int ip = (int)fl;
    const unsigned frac = (unsigned)((p - fl) * 4294967296.0); // p - floor(p) is exact; truncation
    w.q = ((unsigned long long)(unsigned)ip << 32) | frac;
    const int idx = floormod_fast(ip, Lc, inv_lc);
    w.base = ip - idx;
    return idx;
}

// advance by `rate` (one sample: c.rate; several: a multiple of it); returns the predicted chip index and sets
// `ambiguous` when the prediction is not proven to equal the reference's floor
__device__ __forceinline__ int chip_walk_next(ChipWalk &w, unsigned long long rate, unsigned margin, int Lc, bool &ambiguous)
{
    w.q += rate;
    const unsigned frac = (unsigned)w.q;
    // safe  <=>  margin <= frac <= 2^32 - 1 - margin  <=>  (frac - margin) <= (2^32 - 1 - 2 margin)   (unsigned)
    ambiguous = (frac - margin) > (0xffffffffu - 2u * margin);
    const unsigned idx = (unsigned)((int)(w.q >> 32) - w.base);
    const unsigned wrapped = min(idx, idx - (unsigned)Lc); // idx < 2 Lc: one conditional subtraction of Lc
    w.base += (int)(idx - wrapped);                        // keep the table index relative to the current lap
    return (int)wrapped;
}

} // namespace gat
