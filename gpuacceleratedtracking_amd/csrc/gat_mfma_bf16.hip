// gat_mfma_bf16.hip -- downconvert + correlate on the bf16 matrix pipe with FP32-equivalent
// accuracy: both GEMM operands are split into three bf16 terms (hi + mid + lo carries all 24
// mantissa bits of an f32) and the cross products are laid along the MFMA's reduction dimension.
//
//   R[m,(k,l)] = sum_n x[n,m] * W[n,(k,l)],   W = conj(carrier_k[n]) * c_k[n + shift_l]
//
// as the real GEMM  C[32 x 32] += X^T[32 x 16] * W[16 x 32]  on v_mfma_f32_32x32x16_bf16:
//   rows    i = 2*m + {x_re, x_im}            (16 antennas per row tile, RT row tiles per wave)
//   columns j = 2*(k*L + l) + {w_re = chip*cos, w_im = -chip*sin}, packed flat over all channels
//   the 16 reduction slots = 2 samples (one per 32-lane half) x 8 cross products:
//        slot        0    1    2    3    4    5    6    7
//        x term      hi   mid  hi   mid  hi   mid  lo   lo       = dwords {a, a, a, b}
//        w term      hi   hi   mid  mid  lo   lo   hi   mid      = dwords {hh, mm, ll, hm}
//   i.e. every product of the 3 x 3 expansion except lo*lo (2^-32 relative).  bf16 x bf16 is exact
//   in f32 and the pipe accumulates in f32, so the result carries f32 accuracy at 8/16 of the bf16
//   MFMA rate -- 8x the FP32 MFMA / vector rate (gat_mfma.hip measured the f32 MFMA at the vector rate,
//   sharing its issue with the VALU; this pipe runs beside the VALU).
//
// x is split once per sample by the producer waves into LDS as {a = hi|mid<<16, b = lo|lo<<16}
// (8 B per value; the consumer's X fragment is {a, a, a, b}: no arithmetic).  The carrier is split once
// per (sample, channel) into the four W dwords for w_re and w_im (16 B each); the chip only flips
// signs: the consumer XORs the fragment with a 0x80008000 / 0 mask from the LDS code replica -- the
// three taps and the 16*RT antennas of a tile all reuse one carrier split.
// Code replica: exactly the reference's FP64 expression, unfused (src/algorithms.jl:179), as in the
// other kernels; per step only the T new entries are evaluated: the chip-sign words of a channel slot live in a ring in
// LDS that the step's window moves along (round 5, s_rep below; instances of one row tile: two buffers and a copy of
// the tap-span overlap).
// Sample formats: planar / interleaved ComplexF32, int16 and int8 pairs (template FMT).  int8 samples are exact in
// ONE bf16 term: 3 products {x*hi, x*mid, x*lo} = 4 slots per sample, a slice covers 4 samples (half the MFMAs) and
// both fragments come ready-made from LDS (X = {x|x, x|0}, W = {hi|mid, lo|0} per sample; see X1 below).  int16 samples are
// exact in TWO terms: 5 products per sample laid as a stream over consecutive MFMAs (round 5, FragSet<RT, kMbTwo> below).
// What bounds the kernel (profiles/r05/mfma_cycle_stamps_*.txt, mfma_bf16_ablation_*.txt): float samples --
// the consumers (matrix pipe + fragment fetches + operand preparation) at the power cap's clock; int16 / int8 -- the producer
// waves' chains and the consumers about level, coupled through LDS and the issue ports.
// One accumulation chain covers at most kMaxChain samples (the planner splits longer blocks over
// workgroups, finalize_kernel adds the partials in a fixed order), so the f32 rounding of a running
// sum of millions of samples stays far inside 1e-5.
#include "gat_internal.h"
#include "gat_phase.h"

#include <utility>

namespace gat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

// (workgroup size mb_threads, consumer waves, tile size, slot and chain limits, LDS bytes: gat_internal.h -- the host's
// planner and this kernel size everything with the same functions)
constexpr int consumer_waves(int RT, int NCT) { return mb_consumer_waves(RT, NCT); }
constexpr int kMaxChain = kMbMaxChain;
constexpr int kReanchor = 16;   // steps between FP64 re-anchors of the producers' carried phasor / code index
constexpr int kHeader = kMbHeader;

// hi/mid/lo bf16 terms of a float by TRUNCATION: hi = top 16 bits of v, mid = top 16 bits of
// r = v - hi, lo = top 16 bits of r2 = r - mid.  Every residual is exact in f32 and each term takes 8
// of the 24 mantissa bits, so v == hi + mid + lo exactly (all three carry v's sign).
struct Split3 {
    unsigned v, r, r2; // bit patterns whose top halves are the hi / mid / lo terms
};
__device__ __forceinline__ Split3 split3(float v)
{
    Split3 s;
    s.v = __float_as_uint(v);
    const float r = v - __uint_as_float(s.v & 0xffff0000u);
    s.r = __float_as_uint(r);
    s.r2 = __float_as_uint(r - __uint_as_float(s.r & 0xffff0000u));
    return s;
}
// The sample prefetch is written as inline assembly with its own s_waitcnt: hipcc's counter
// insertion over-waits across the two-phase loop (it asked for the NEWEST loads too, i.e. no
// prefetch distance).  The compiler does not see these loads; every use of the destination goes
// through wait_loads(), which ties the registers to the wait -- in straight-line code only: a branch
// around the wait makes the register allocator COPY the registers before it, i.e. before the data is there.
__device__ __forceinline__ void gload_nt(f32x4_ &dst, const void *p)
{
#ifdef GAT_MB_NT_LOADS // A/B build: non-temporal sample loads, as up to round 4
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
#endif
}
template <int XI> // XI loads per register set; the two other sets' 2 * XI loads are newer and may stay in flight
__device__ __forceinline__ void wait_loads(f32x4_ (&xv)[XI])
{
    static_assert(XI >= 1 && XI <= 4, "prefetch group size");
    if constexpr (XI == 1)
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(xv[0]) : : "memory");
    else if constexpr (XI == 2)
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(xv[0]), "+v"(xv[1]) : : "memory");
    else if constexpr (XI == 3)
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]) : : "memory");
    else
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]) : : "memory");
}
// __builtin_amdgcn_perm(a, b, sel): result byte i = byte sel[i] of {b: 0-3, a: 4-7}
#define GAT_PERM(a, b, sel) __builtin_amdgcn_perm((a), (b), (sel))
#define GAT_P2(hi_, lo_) GAT_PERM(hi_, lo_, 0x07060302u) // two bf16: {low half: top half of lo_, high half: top half of hi_}

// ---- consumer fragment fetches -------------------------------------------------------------------
// All LDS reads of the MFMA loop are inline assembly, issued D = 2..4 k-slices ahead of their use into a
// ring of D + 1 register sets, with one explicit counted s_waitcnt per slice (the slice in between may
// stay in flight): an LDS round trip under this kernel's load is longer than the 32 * RT cycles of a
// slice.  lgkmcnt holds at most 15 operations, hence one ds_read_b64 per X fragment (2 + RT reads per
// slice, 2 slices outstanding).  The X fragment {a, a, a, b} is the fetched {a, b} plus ONE
// v_pk_mov_b32 for the {a, a} half (plain C++ gets three v_mov per MFMA, on the unit that limits
// this kernel).
// X1 = true: single-term samples (int8 input is exact in ONE bf16 term): 4 k-slots per sample
// {x*hi, x*mid, x*lo, 0}, a slice covers 4 samples (two adjacent ones per 32-lane half) -- half the MFMAs, and
// the fragments come ready-made from LDS: X = {x|x, x|0} per sample, W = {hi|mid, lo|0} per sample, one 16-byte read
// fetches a sample pair; the chip signs of the two samples come as two mask dwords.
// XM = kMbTwo (int16 samples, two exact bf16 terms; gat_internal.h MbMode): the k-slots of a 32-lane half are a STREAM -- 5 products
// per sample {a h, a m, a l, b h, b m} -- cut into slices of 8: a slice's X and W fragments are 16 consecutive bytes of the
// slot-ordered rows the producers write (no assembly here), 8 samples per half every 5 slices.  Slice j (mod 5) of a period that
// starts at sample sb covers the samples {sb, sb+1}, {sb+1, sb+2, sb+3}, {sb+3, sb+4}, {sb+4, sb+5, sb+6}, {sb+6, sb+7}: it
// fetches their chip-sign words (two with one ds_read2_b32, a third with a ds_read_b32 for j = 1, 3) and lays them onto the
// fragment's four dwords (mfma_apply).
template <int RT, int XM>
struct FragSet {
    u32x4 w;
    unsigned m;
    u32x2 xa[RT];
#if defined(GAT_ABLATE) && (GAT_ABLATE & 64)
    u32x2 wl;
#endif
};
template <int RT>
struct FragSet<RT, kMbOne> {
    u32x4 w;
    u32x2 m;
    u32x4 xa[RT];
#if defined(GAT_ABLATE) && (GAT_ABLATE & 64)
    u32x2 wl;
#endif
};
template <int RT>
struct FragSet<RT, kMbTwo> {
    u32x4 w;
    u32x2 m;
    unsigned m3;
    u32x4 xa[RT];
};
// LDS reads of slice J (they count in lgkmcnt)
template <int RT, int XM>
constexpr int frag_reads(int J) { return XM == kMbTwo ? 2 + RT + ((J % 5 == 1 || J % 5 == 3) ? 1 : 0) : 2 + RT; }
// ... of the n slices issued after slice J (slice numbers wrap at NM: the cross-step pipeline runs into the next step)
template <int RT, int XM>
constexpr int frag_newer(int J, int n, int NM)
{
    int c = 0;
    for (int i = 1; i <= n; ++i) c += frag_reads<RT, XM>((J + i) % NM);
    return c;
}
template <int J, int RT, int XM>
__device__ __forceinline__ void frag_issue(FragSet<RT, XM> &s, unsigned w_addr, unsigned r_addr, const unsigned (&x_addr)[RT])
{
#if defined(GAT_ABLATE) && (GAT_ABLATE & 256) // diagnostic: no fragment fetches (one LDS read keeps the counted waits meaningful)
    {
        unsigned one;
        asm volatile("ds_read_b32 %0, %1" : "=v"(one) : "v"(r_addr));
        if constexpr (XM == kMbThree) s.m = one; else s.m[0] = one;
        return;
    }
#endif
#if defined(GAT_ABLATE) && (GAT_ABLATE & 64) // diagnostic: half the W fragment bytes (results wrong)
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(s.wl) : "v"(w_addr), "n"(J * 16));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(s.w) : "v"(w_addr), "n"(J * 16));
#endif
    if constexpr (XM == kMbTwo) {
        constexpr int j = J % 5, sb = 8 * (J / 5);
        constexpr int o0 = sb + (j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 3 : j == 3 ? 4 : 6); // first sample of the slice
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(s.m) : "v"(r_addr), "n"(o0), "n"(o0 + 1));
        if constexpr (j == 1 || j == 3) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s.m3) : "v"(r_addr), "n"((o0 + 2) * 4));
#pragma unroll
        for (int t = 0; t < RT; ++t)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(s.xa[t]) : "v"(x_addr[t]), "n"(J * 16));
    } else if constexpr (XM == kMbOne) {
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(s.m) : "v"(r_addr), "n"(2 * J), "n"(2 * J + 1));
#pragma unroll
        for (int t = 0; t < RT; ++t)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(s.xa[t]) : "v"(x_addr[t]), "n"(J * 16));
    } else {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s.m) : "v"(r_addr), "n"(J * 4));
#pragma unroll
        for (int t = 0; t < RT; ++t)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(s.xa[t]) : "v"(x_addr[t]), "n"(J * 8));
    }
}
template <int RT, int XM, int NEWER> // NEWER: reads issued after this set's that may stay in flight
__device__ __forceinline__ void frag_wait(FragSet<RT, XM> &s)
{
    static_assert(RT == 1 || RT == 2 || RT == 4, "row tiles");
    static_assert(NEWER <= 15, "lgkmcnt is a 4-bit counter");
    if constexpr (XM == kMbTwo) {
        if constexpr (RT == 1)
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(s.w), "+v"(s.m), "+v"(s.m3), "+v"(s.xa[0]) : "n"(NEWER));
        else if constexpr (RT == 2)
            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(s.w), "+v"(s.m), "+v"(s.m3), "+v"(s.xa[0]), "+v"(s.xa[1]) : "n"(NEWER));
        else
            asm volatile("s_waitcnt lgkmcnt(%7)"
                         : "+v"(s.w), "+v"(s.m), "+v"(s.m3), "+v"(s.xa[0]), "+v"(s.xa[1]), "+v"(s.xa[2]), "+v"(s.xa[3])
                         : "n"(NEWER));
    } else if constexpr (RT == 1)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(s.w), "+v"(s.m), "+v"(s.xa[0]) : "n"(NEWER));
    else if constexpr (RT == 2)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(s.w), "+v"(s.m), "+v"(s.xa[0]), "+v"(s.xa[1]) : "n"(NEWER));
    else
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(s.w), "+v"(s.m), "+v"(s.xa[0]), "+v"(s.xa[1]), "+v"(s.xa[2]), "+v"(s.xa[3])
                     : "n"(NEWER));
}
// fetch distance in slices: as deep as the 4-bit lgkmcnt allows (2 + RT reads per slice, 3 + RT at most on the two-term path), at most 4
template <int RT, int XM = kMbThree>
constexpr int frag_depth()
{
    constexpr int per = XM == kMbTwo ? 3 + RT : 2 + RT;
#ifndef GAT_MB_X2_DEPTH2
    // (four row tiles on the two-term and one-term paths: 64 accumulator registers + 23 per fragment set -- three sets do not fit the 128 of a
    // 16-wave workgroup, and what the allocator then spills are fragments that LDS reads may still have in flight: two sets)
    if ((XM == kMbTwo || XM == kMbOne) && RT == 4) return 1; // (one term: 22 registers per set, the same arithmetic)
#endif
    return (15 / per) < 4 ? (15 / per) : 4;
}

template <int RT, int XM>
using FragRing = FragSet<RT, XM>[frag_depth<RT, XM>() + 1]; // fragment sets in rotation

template <int J, int RT, int XM>
__device__ __forceinline__ void mfma_apply(f32x16 (&acc)[RT == 1 ? 2 : RT], FragSet<RT, XM> &cur);

template <int J, int NM, int RT, int XM>
__device__ __forceinline__ void mfma_slice(f32x16 (&acc)[RT == 1 ? 2 : RT], FragRing<RT, XM> &fs, unsigned w_addr,
                                           unsigned r_addr, const unsigned (&x_addr)[RT])
{
    constexpr int D = frag_depth<RT, XM>();
    FragSet<RT, XM> &cur = fs[J % (D + 1)];
    // slices J+1 .. J+D-1 (those that exist) were issued after this one and may stay in flight
    constexpr int newer = (NM - 1 - J) < (D - 1) ? (NM - 1 - J) : (D - 1);
    frag_wait<RT, XM, frag_newer<RT, XM>(J, newer, NM)>(cur);
    if constexpr (J + D < NM) frag_issue<J + D, RT, XM>(fs[(J + D) % (D + 1)], w_addr, r_addr, x_addr);
    mfma_apply<J, RT, XM>(acc, cur);
}
// the arithmetic of one k-slice: chip signs onto W, the {a, a} halves of the X fragments, RT MFMAs
template <int J, int RT, int XM>
__device__ __forceinline__ void mfma_apply(f32x16 (&acc)[RT == 1 ? 2 : RT], FragSet<RT, XM> &cur)
{
    constexpr bool X1 = XM == kMbOne;
#if defined(GAT_ABLATE) && (GAT_ABLATE & 128) // diagnostic (results wrong): the arithmetic of two k-slices out of three -- what a
    if constexpr (XM == kMbThree && J % 3 == 2) return;   // 2-term split of the SAMPLE operand (5 products, 3 samples per MFMA) would issue
#endif
#if defined(GAT_ABLATE) && (GAT_ABLATE & 512) // diagnostic: no vector preparation of the operands (3-term path)
    if constexpr (XM == kMbThree) {
        const bf16x8 bw0 = __builtin_bit_cast(bf16x8, cur.w);
#pragma unroll
        for (int t = 0; t < RT; ++t)
            acc[RT == 1 ? (J & 1) : t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, u32x4{cur.xa[t][0], cur.xa[t][1], cur.xa[t][0], cur.xa[t][1]}), bw0, acc[RT == 1 ? (J & 1) : t], 0, 0, 0);
        return;
    }
#endif
    u32x4 w = cur.w;
#if defined(GAT_ABLATE) && (GAT_ABLATE & 64)
    if constexpr (XM == kMbThree) w = u32x4{cur.wl[0], cur.wl[1], cur.wl[0], cur.wl[1]};
#endif
    if constexpr (XM == kMbTwo) {
        // chip signs of the slice's two or three samples onto the dwords (= slot pairs) of the W fragment; LH(a, b) = low half of
        // a | high half of b, where a sample ends inside a dword
        const unsigned A = cur.m[0], B = cur.m[1], C = cur.m3;
        constexpr int j = J % 5;
        auto LH = [](unsigned a_, unsigned b_) { return GAT_PERM(b_, a_, 0x07060100u); };
        if constexpr (j == 0) { w[0] ^= A; w[1] ^= A; w[2] ^= LH(A, B); w[3] ^= B; }          // samples 0 0 0 0 0 1 1 1
        else if constexpr (j == 1) { w[0] ^= A; w[1] ^= B; w[2] ^= B; w[3] ^= LH(B, C); }      // 1 1 2 2 2 2 2 3
        else if constexpr (j == 2) { w[0] ^= A; w[1] ^= A; w[2] ^= B; w[3] ^= B; }             // 3 3 3 3 4 4 4 4
        else if constexpr (j == 3) { w[0] ^= LH(A, B); w[1] ^= B; w[2] ^= B; w[3] ^= C; }      // 4 5 5 5 5 5 6 6
        else { w[0] ^= A; w[1] ^= LH(A, B); w[2] ^= B; w[3] ^= B; }                            // 6 6 6 7 7 7 7 7
        const bf16x8 bw = __builtin_bit_cast(bf16x8, w);
#pragma unroll
        for (int t = 0; t < RT; ++t)
            acc[RT == 1 ? (J & 1) : t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.xa[t]), bw,
                                                                             acc[RT == 1 ? (J & 1) : t], 0, 0, 0);
    } else if constexpr (X1) { // chip signs of the lane's two samples
        const unsigned m0 = cur.m[0], m1 = cur.m[1];
        w[0] ^= m0;
        w[1] ^= m0;
        w[2] ^= m1;
        w[3] ^= m1;
        const bf16x8 bw = __builtin_bit_cast(bf16x8, w);
#pragma unroll
        for (int t = 0; t < RT; ++t)
            acc[RT == 1 ? (J & 1) : t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.xa[t]), bw,
                                                                             acc[RT == 1 ? (J & 1) : t], 0, 0, 0);
    } else {
        const unsigned mk = cur.m; // chip sign of this column's tap: 0x80008000 or 0
        w[0] ^= mk;
        w[1] ^= mk;
        w[2] ^= mk;
        w[3] ^= mk;
        const bf16x8 bw = __builtin_bit_cast(bf16x8, w);
        // {a, a} halves of the RT fragments.  The trailing s_nop is the VALU-write -> MFMA-read hazard the
        // compiler would handle itself if it could see inside the asm (it pads its own v_xor the same way).
        u32x2 aa[RT];
        if constexpr (RT == 1)
            asm("v_pk_mov_b32 %0, %1, %1 op_sel:[0,0]\n\ts_nop 1" : "=&v"(aa[0]) : "v"(cur.xa[0]));
        else if constexpr (RT == 2)
            asm("v_pk_mov_b32 %0, %2, %2 op_sel:[0,0]\n\tv_pk_mov_b32 %1, %3, %3 op_sel:[0,0]\n\ts_nop 1"
                : "=&v"(aa[0]), "=&v"(aa[1])
                : "v"(cur.xa[0]), "v"(cur.xa[1]));
        else
            asm("v_pk_mov_b32 %0, %4, %4 op_sel:[0,0]\n\tv_pk_mov_b32 %1, %5, %5 op_sel:[0,0]\n\t"
                "v_pk_mov_b32 %2, %6, %6 op_sel:[0,0]\n\tv_pk_mov_b32 %3, %7, %7 op_sel:[0,0]\n\ts_nop 1"
                : "=&v"(aa[0]), "=&v"(aa[1]), "=&v"(aa[2]), "=&v"(aa[3])
                : "v"(cur.xa[0]), "v"(cur.xa[1]), "v"(cur.xa[2]), "v"(cur.xa[3]));
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const unsigned a0 = aa[t][0], a1 = aa[t][1], a2 = cur.xa[t][0], b3 = cur.xa[t][1];
            const u32x4 af = u32x4{a0, a1, a2, b3};
            // (one row tile: two accumulators take the slices in turn -- a dependent chain of this MFMA issues
            // every 52 cycles, independent ones every 32: scripts/probes/mfma_bf16_probe.hip)
            acc[RT == 1 ? (J & 1) : t] =
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), bw, acc[RT == 1 ? (J & 1) : t], 0, 0, 0);
        }
    }
}
template <int NM, int RT, int XM, int... J>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[RT == 1 ? 2 : RT], unsigned w_addr, unsigned r_addr, const unsigned (&x_addr)[RT],
                                          std::integer_sequence<int, J...>)
{
    constexpr int D = frag_depth<RT, XM>();
    FragSet<RT, XM> fs[D + 1];
    frag_issue<0, RT, XM>(fs[0], w_addr, r_addr, x_addr);
    if constexpr (NM > 1 && D > 1) frag_issue<1, RT, XM>(fs[1], w_addr, r_addr, x_addr);
    if constexpr (NM > 2 && D > 2) frag_issue<2, RT, XM>(fs[2], w_addr, r_addr, x_addr);
    if constexpr (NM > 3 && D > 3) frag_issue<3, RT, XM>(fs[3], w_addr, r_addr, x_addr);
    (mfma_slice<J, NM, RT, XM>(acc, fs, w_addr, r_addr, x_addr), ...);
}

// ---- k-slices pipelined ACROSS the step barrier (round 3) ---------------------------------------------------------
// mfma_step above starts every step with an empty pipeline: after the barrier the first fragments have to come in before the
// first MFMA can issue -- one LDS round trip under load per 32-sample step, ~12 % of the consumers' time at configs[4].
// Here the fragment ring lives across steps: the last D slices of a step are fetched BEFORE the barrier (all of the step's
// reads are home by then: the producers may overwrite its buffer) and their MFMAs run AFTER it, while the first D slices of
// the next step -- from the other buffer, complete at the barrier -- are already on their way.  R = ring position of the
// step's slice 0 (the ring has D + 1 sets and a step NM slices: R advances by NM mod (D + 1) per step).  There are always
// exactly D - 1 newer slices in flight at a wait (after the last step too: its "next step" reads fetch stale data nobody uses).
template <int J, int NM, int RT, int XM, int R>
__device__ __forceinline__ void xstep_slice(f32x16 (&acc)[RT == 1 ? 2 : RT], FragRing<RT, XM> &fs, unsigned w_addr,
                                            unsigned r_addr, const unsigned (&x_addr)[RT], int dw, int dr, int dx)
{
    constexpr int D = frag_depth<RT, XM>(), RING = D + 1;
    FragSet<RT, XM> &cur = fs[(R + J) % RING];
    frag_wait<RT, XM, frag_newer<RT, XM>(J, D - 1, NM)>(cur);
    if constexpr (J + D < NM) {
        frag_issue<J + D, RT, XM>(fs[(R + J + D) % RING], w_addr, r_addr, x_addr);
    } else {
        if constexpr (J + D == NM) { // the first fetch that belongs to the next step: this step's reads are home, then the barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(defined(GAT_ABLATE) && (GAT_ABLATE & 32)) // diagnostic: no step barrier (both roles)
            __syncthreads();
#endif
        }
        unsigned x_next[RT]; // the other buffer: a wave-uniform distance away
#pragma unroll
        for (int t = 0; t < RT; ++t) x_next[t] = x_addr[t] + (unsigned)dx;
        frag_issue<J + D - NM, RT, XM>(fs[(R + J + D) % RING], w_addr + (unsigned)dw, r_addr + (unsigned)dr, x_next);
    }
    mfma_apply<J, RT, XM>(acc, cur);
}
template <int NM, int RT, int XM, int R, int... J>
__device__ __forceinline__ void xstep(f32x16 (&acc)[RT == 1 ? 2 : RT], FragRing<RT, XM> &fs, unsigned w_addr,
                                      unsigned r_addr, const unsigned (&x_addr)[RT], int dw, int dr, int dx, std::integer_sequence<int, J...>)
{
    (xstep_slice<J, NM, RT, XM, R>(acc, fs, w_addr, r_addr, x_addr, dw, dr, dx), ...);
}

struct ChanInfoB { // per channel slot of the workgroup, in LDS
    double ratio, tau, step, phi;
    float wr, wi;   // e^{j 2 pi OS step}: carrier rotation between the two samples of a producer item (OS samples apart)
    float wTr, wTi; // e^{j 2 pi T step}: one-step carrier rotation
    int prn, valid, bad;
    int inc_ok;     // fewer than Lc chips per step: the code index can be advanced by floor differences
};
static_assert(sizeof(ChanInfoB) == 64, "ChanInfoB layout");

constexpr int tile_samples(int RT, int NCT) { return mb_tile_samples(RT, NCT); }

} // namespace

// RT: 16-antenna row tiles per workgroup (1, 2, 4); NCT: 32-column channel tiles per workgroup (1, 2, 4).
// Workgroup = 16 (or 12) waves, four (three) per SIMD: waves 0-3 consumers (MFMA + fragment fetches), waves 4-15 (4-11)
// producers (HBM loads two steps ahead, bf16 splits, carriers, code replica) -- their long dependent
// chains (FP64 code phase, sincos, split) hide behind each other and behind the matrix pipe.
// Double-buffered LDS, one s_barrier per step of T samples.
template <int RT, int NCT, int FMT, int XM>
__global__ void __launch_bounds__(mb_threads(RT, NCT)) mfma_bf16_kernel(const MfArgs a)
{
    static_assert((XM == kMbOne) == (FMT == GAT_LAYOUT_INTERLEAVED_I8), "one term: int8 samples, and only they");
    static_assert(XM != kMbTwo || FMT == GAT_LAYOUT_INTERLEAVED_I16, "two terms: int16 samples");
    constexpr int kMbThreads = mb_threads(RT, NCT);
    constexpr int T = tile_samples(RT, NCT);
    constexpr int NCW = consumer_waves(RT, NCT);
    constexpr int WPT = NCW / NCT; // consumer waves per channel tile (they split the step's samples)
    constexpr int SW = T / WPT;    // samples per consumer wave and step
    constexpr bool X1 = XM == kMbOne; // samples exact in one bf16 term: 4 samples per MFMA
    constexpr bool X2 = XM == kMbTwo; // ... in two: 16 samples per 5 MFMAs (the slot stream, FragSet above)
    constexpr int SPS = X1 ? 4 : 2; // samples per k-slice
    constexpr int NM = X2 ? 5 * (SW / 2) / 8 : SW / SPS; // MFMA k-slices per consumer wave and step (two sample streams)
    static_assert(!X2 || (SW / 2) % 8 == 0, "two-term stream: 8 samples per half and period");
    // A producer item = one channel slot x TWO samples OS apart.  3-term path: samples q and q + T/2, so that the 64 lanes
    // of a wave (consecutive q) store CONSECUTIVE 16-byte W entries and consecutive replica words (round 2 owned the
    // adjacent samples 2q, 2q + 1: every W store 32 bytes from its neighbour's -- every other bank group, 2-way conflicts,
    // SQ_LDS_BANK_CONFLICT 24.6 % of SQ_LDS_IDX_ACTIVE at configs[4]; 18.3 % now, profiles/r03).  1-term path (int8): the
    // consumers read adjacent sample pairs as one 16-byte entry, the item keeps 2q, 2q + 1.  2-term path: an item is FOUR
    // adjacent samples 4q .. 4q + 3 (40 bytes of a slot-ordered row, five 8-byte stores), rotated one sample at a time.
    constexpr int OS = XM == kMbThree ? T / 2 : 1;
#ifdef GAT_MB_X2_ITEMS4 // A/B build: four samples per item of the two-term path (40 bytes, five 8-byte stores)
    constexpr int IS = X2 ? 4 : 2;    // samples per producer item
#else
    constexpr int IS = 2;             // samples per producer item (2-term path: 20 bytes of a slot-ordered row, five 4-byte stores)
#endif
    constexpr int RB = mb_two_row_bytes(T); // 2-term path: bytes per X / W row (10 per sample, slot order)
    // LDS row strides in entries.  3-term: X 8 B / W 16 B per sample, odd (32 planes x one sample = 32 distinct
    // bank pairs).  1-term: X 8 B / W 8 B per sample, read in 16-byte pairs: even, rows 4 banks apart.
    constexpr int XS = X1 ? T + 2 : T + 1;
    constexpr int WS = X1 ? T + 2 : T + 1;
    constexpr int WE = X1 ? 1 : 2; // u32x2 units per W entry
    // one producer "group" = one 16-byte load: GS consecutive complex samples of one plane (planar f32) or of
    // one antenna (interleaved ComplexF32 / int16 / int8 pairs)
    constexpr bool PLANAR = FMT == GAT_LAYOUT_PLANAR;
    constexpr int GS = PLANAR ? 4 : FMT == GAT_LAYOUT_INTERLEAVED ? 2 : FMT == GAT_LAYOUT_INTERLEAVED_I16 ? 4 : 8;
    constexpr int BYTES = FMT == GAT_LAYOUT_INTERLEAVED_I16 ? 4 : FMT == GAT_LAYOUT_INTERLEAVED_I8 ? 2 : 8; // per complex sample
    constexpr int QPR = T / GS;    // groups per row
    constexpr int PT = kMbThreads - 64 * NCW; // producer threads (768 or 512)
    constexpr int NG = (PLANAR ? RT * 32 : RT * 16) * QPR; // groups per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Columns are packed flat: column c = 2*L*k + 2*l + comp, a workgroup owns columns [32*NCT*cg, 32*NCT*(cg+1))
    // -- a channel may straddle two tiles (or two workgroups: both generate its carrier), no column is
    // wasted on alignment (64 channels x 3 taps = 12 tiles, not 13).  Channel slots of the workgroup:
    // k_first .. k_first + nslots - 1 (nslots = the host's upper bound, LDS is laid out for it).
    const int L = a.L;
    const int nslots = a.nslots;
    const int wrows = 2 * nslots + 1; // + one row of zeros for dead columns
    const int RS = a.rep_stride;
    ChanInfoB *s_chan = reinterpret_cast<ChanInfoB *>(smem);
    u32x2 *s_x = reinterpret_cast<u32x2 *>(smem + kHeader);                 // [2][RT*32][XS]   (2-term: [2][RT*32] rows of RB bytes)
    u32x2 *s_w = X2 ? s_x + 2 * RT * 32 * (RB / 8) : s_x + 2 * RT * 32 * XS; // [2][wrows][WS] entries of WE u32x2 (2-term: [2][wrows] rows of RB bytes)
    // Chip-sign words: ONE ring of RR entries per channel slot (round 5; up to round 4: two buffers and a copy of the span
    // overlapping the next step -- 98 entries x 23 slots per step at configs[4], 7 % of the kernel).  Entry e of the step
    // whose window starts at ring position s (s advances by T per step, modulo RR) sits at (s + e) mod RR; positions below
    // the window length WIN = span + T are ALSO kept at RR + position, so a window [s, s + WIN) is contiguous wherever it
    // starts.  RR >= WIN + T: the entries written for the next step never touch the window being read.
    // Instances of ONE row tile keep the two buffers and the copy (mb_rep_ring_rows): with the ring their consumer loops come out
    // of the compiler with 40-350 scratch accesses (5 fragment sets in rotation), and no default path runs them on float data.
    constexpr bool RINGREP = mb_rep_ring_rows(RT);
    unsigned *s_rep = reinterpret_cast<unsigned *>(X2 ? s_w + 2 * wrows * (RB / 8) : s_w + 2 * wrows * WS * WE); // ring: [nslots][RS], RS >= RR + WIN; else [2][nslots][RS]
    unsigned *s_code = s_rep + (((RINGREP ? 1 : 2) * nslots * RS + 3) & ~3); // [nslots][code_bits_stride] sign-bit tables
    const int RR = a.rep_ring, WIN = a.rep_span + tile_samples(RT, NCT);
    // a window's start one step on: (s + T) mod RR without a select -- a select on a wave-uniform value may be compiled into a
    // BRANCH, and a control-flow merge inside the step loops makes the register allocator copy registers that inline-assembly
    // loads still have in flight (the consumers' fragments, the producers' sample sets): wrong results in single instances
    auto ring_next = [RR](int s_) {
        const int n = s_ + tile_samples(RT, NCT), m = (n - RR) >> 31; // m = -1 while n < RR
        return (n & m) | ((n - RR) & ~m);
    };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= NCW;
    const int ptid = tid - 64 * NCW;

    // workgroup -> (tile, channel group); blocks id and id+8 share an XCD, so the channel groups of
    // one sample tile run back to back on one L2 (speed only)
    const unsigned xcd = blockIdx.x & 7u, jq = blockIdx.x >> 3;
    const int cg = (int)(jq % (unsigned)a.chan_groups);
    unsigned tile = (jq / (unsigned)a.chan_groups) * 8u + xcd;
    if (tile >= (unsigned)a.num_tiles) return;
    const int split = tile % a.splits;
    tile /= a.splits;
    const int at = tile % a.ant_tiles;
    const int b = tile / a.ant_tiles;
    const int N = (int)a.N; // a multiple of 4 (planner)
    const int Lc = a.Lc;
    const float inv_lc = 1.0f / (float)Lc;
    const int span = a.rep_span;

    const int k_first = (32 * NCT * cg) / (2 * L);
    if (tid < nslots) {
        const int k = k_first + tid;
        ChanInfoB ci{};
        ci.valid = k < a.K;
        if (ci.valid) {
            const gat_channel_params P = a.params[(size_t)b * a.K + k];
            ci.ratio = P.code_freq_hz / a.fs;
            ci.step = P.carrier_freq_hz / a.fs;
            ci.tau = P.code_phase_chips;
            ci.phi = P.carrier_phase_cycles;
            const double sp = __builtin_fabs(ci.tau) + __builtin_fabs(ci.ratio) * (double)(N + a.max_abs_shift) + 1.0;
            ci.bad = P.prn < 0 || P.prn >= a.num_prns || !(sp < 1073741824.0) || !(sp < 2097152.0 * (double)Lc) ||
                     !(ci.ratio >= 0.0) || !(ci.ratio * 32.0 < (double)Lc) || !(ci.step == ci.step) || !(ci.phi == ci.phi);
            ci.prn = (P.prn < 0 || P.prn >= a.num_prns) ? 0 : P.prn;
            if (ci.bad) { ci.ratio = 0.0; ci.tau = 0.0; ci.step = 0.0; ci.phi = 0.0; }
            const double st_O = ci.step * (double)OS;
            sincos_cycles(st_O - __builtin_rint(st_O), ci.wr, ci.wi);
            const double st_T = ci.step * (double)T;
            sincos_cycles(st_T - __builtin_rint(st_T), ci.wTr, ci.wTi);
            ci.inc_ok = ci.ratio * (double)(T + 2) < (double)Lc;
        }
        s_chan[tid] = ci;
    }
    // the zero rows of both carrier buffers (read by dead columns, never written again)
    if constexpr (X2) {
        for (int e = tid; e < 2 * (RB / 8); e += kMbThreads)
            s_w[((e / (RB / 8)) * wrows + 2 * nslots) * (RB / 8) + e % (RB / 8)] = u32x2{0u, 0u};
    } else {
        for (int e = tid; e < 2 * WS * WE; e += kMbThreads)
            s_w[((e / (WS * WE)) * wrows * WS + 2 * nslots * WS) * WE + e % (WS * WE)] = u32x2{0u, 0u};
    }
    __syncthreads();
    { // sign-bit tables of this workgroup's channels (the producers must not touch global memory for chips:
      // vector-memory returns are in order, a table gather would wait for the sample prefetch in flight)
        const int vec_per_row = a.code_bits_stride / 4;
        for (int e = tid; e < nslots * vec_per_row; e += kMbThreads) {
            const int slot = e / vec_per_row, v = e - slot * vec_per_row;
            const ChanInfoB c = s_chan[slot];
            if (c.valid)
                reinterpret_cast<uint4 *>(s_code)[slot * vec_per_row + v] =
                    reinterpret_cast<const uint4 *>(a.code_bits + (size_t)c.prn * a.code_bits_stride)[v];
        }
        __syncthreads();
    }

    const size_t base = (size_t)b * a.block_stride + (size_t)(at * 16 * RT) * a.ant_stride; // in (complex) samples
    const int s_begin = split * a.steps_per_split;
    const int s_end = min(s_begin + a.steps_per_split, a.total_steps);

    // ---- producers ---------------------------------------------------------------------------------
    // Sample loads run three steps ahead of the split/store (three register sets in rotation).  Every load is an
    // unconditional 16-byte load (out-of-range groups read 16 zero bytes the context keeps for this):
    // straight-line code, no selects, and the compiler's s_waitcnt counts only the loads that matter.
    // (the im plane is addressed as re + im_delta: a lane-dependent choice between the two kernel-argument
    // pointers would be compiled into a per-lane LOAD of the pointer and a wait for it)
    const long long im_delta = PLANAR ? reinterpret_cast<const char *>(a.im) - reinterpret_cast<const char *>(a.re) : 0ll;
    const char *re_base = reinterpret_cast<const char *>(a.re) + (PLANAR ? 4ll : (long long)BYTES) * (long long)base;
    // Groups per thread are weighted by role so that all producer waves finish together: the waves that
    // own a carrier / replica item (the first item_waves ones) take few sample groups, the others up to 4.
    const int n_items = nslots * (T / IS);
    const int item_waves = (n_items + 63) >> 6;
    const int niw = item_waves * 64, tn = PT - niw;            // item / non-item producer threads
    const int xi_other = tn > 0 ? min(4, (NG + tn - 1) / tn) : 0;
    const int rem = NG - xi_other * tn;                          // groups left for the item waves
    const int xi_item = rem > 0 ? (rem + niw - 1) / niw : 0;     // <= 4 (planner)
    const bool item_wave = (wave - NCW) < item_waves;
    const int my_xi = __builtin_amdgcn_readfirstlane(item_wave ? xi_item : xi_other); // wave-uniform, in an SGPR
    const int g_first = item_wave ? xi_other * tn + ptid : ptid - niw;
    const int g_stride = item_wave ? niw : tn;
    // 2-term path: a ds_write_b64 is served 16 consecutive lanes at a time = the 8 sample groups of TWO rows; the banks one row's
    // 40-byte items cover and those of the row 4 planes (2 antennas) on are complementary, those of the next antenna overlap
    // (RB = 84 dwords: 4 planes = 16 banks on): consecutive group rows are antennas 0 2 1 3 of every four
    auto x_row = [&](int row) { return X2 ? ((row & ~3) | ((row & 1) << 1) | ((row >> 1) & 1)) : row; };
    auto load_x = [&](auto &xv, int st) {
        constexpr int XI = sizeof(xv) / sizeof(xv[0]);
        const int nb = st * T;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int id = g_first + it * g_stride;
            const int row = x_row(id / QPR), q = id % QPR; // planar: plane 2*m_local + comp; interleaved: antenna m_local
            const int n = nb + GS * q;
            const bool ok = id < NG && st < s_end && n < N;
            const long long off = PLANAR ? ((row & 1) ? im_delta : 0ll) + 4ll * ((long long)(row >> 1) * a.ant_stride + n)
                                         : (long long)BYTES * ((long long)row * a.ant_stride + n);
#if defined(GAT_ABLATE) && (GAT_ABLATE & 16) // diagnostic: no sample loads at all
            (void)ok; (void)off; xv[it] = f32x4_{1.f, 1.f, 1.f, 1.f};
#else
            gload_nt(xv[it], ok ? re_base + off : reinterpret_cast<const char *>(a.zeros));
#endif
        }
    };
    auto store_x = [&](auto &xv, int st, int buf) {
        constexpr int XI = sizeof(xv) / sizeof(xv[0]);
#if !(defined(GAT_ABLATE) && (GAT_ABLATE & 16))
        wait_loads<XI>(xv);
#endif
#if defined(GAT_ABLATE) && (GAT_ABLATE & 4) // diagnostic: samples are split / stored only for the first two steps
        if (st >= s_begin + 2) return;
#endif
        const int nb = st * T;
        u32x2 *xb = X2 ? s_x + buf * RT * 32 * (RB / 8) : s_x + buf * RT * 32 * XS;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int id = g_first + it * g_stride;
            if (id >= NG) continue;
            const int row = x_row(id / QPR), q = id % QPR;
            if constexpr (X2) {
                // int16 v = a + b exactly: a = v rounded to bf16's 8 significant bits (to nearest: the magnitude's bit pattern
                // + half an ulp, truncated), b = v - a -- at most 7 significant bits, |b| <= 2^-8 |a|: both exact in bf16 (the top
                // halves of their float patterns), and the product the slot stream leaves out, b * lo(w), is below 2^-24 of
                // a * w WHATEVER the sample's magnitude (a fixed cut at bit 8 would leave small samples -- an 8-bit ADC in an
                // int16 container -- with the two-term carrier only: 2^-17).  Slot order of a sample: a a a b b.
                unsigned ar[4], br[4], ai[4], bi[4];
                auto two_terms = [](int v, unsigned &a_, unsigned &b_) {
                    const float f = (float)v;
                    a_ = (__float_as_uint(f) + 0x8000u) & 0xffff0000u;
                    b_ = __float_as_uint(f - __uint_as_float(a_));
                };
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int w = (int)__float_as_uint(xv[it][u]); // {re: low half, im: high half}
                    two_terms((int)(short)(w & 0xffff), ar[u], br[u]);
                    two_terms(w >> 16, ai[u], bi[u]);
                }
                auto put4 = [&](int plane, const unsigned (&A)[4], const unsigned (&B)[4]) {
                    u32x2 *d = reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned char *>(xb) + plane * RB + 40 * q);
                    d[0] = u32x2{GAT_P2(A[0], A[0]), GAT_P2(B[0], A[0])};
                    d[1] = u32x2{GAT_P2(A[1], B[0]), GAT_P2(A[1], A[1])};
                    d[2] = u32x2{GAT_P2(B[1], B[1]), GAT_P2(A[2], A[2])};
                    d[3] = u32x2{GAT_P2(B[2], A[2]), GAT_P2(A[3], B[2])};
                    d[4] = u32x2{GAT_P2(A[3], A[3]), GAT_P2(B[3], B[3])};
                };
                put4(2 * row, ar, br);
                put4(2 * row + 1, ai, bi);
                continue;
            }
            // {a = hi | mid << 16, b = lo | lo << 16} of one value into the LDS row of its plane
            auto put = [&](int plane, int rel, float v) {
                if constexpr (X1) { // an int8 value is exact in bf16: {x|x, x|0}
                    const unsigned fb = __float_as_uint(v);
                    xb[plane * XS + rel] = u32x2{GAT_PERM(fb, fb, 0x03020302u), fb >> 16};
                } else {
#if defined(GAT_ABLATE) && (GAT_ABLATE & 128) // diagnostic: the producers' share of a 2-term sample split (no lo term)
                    const unsigned fb = __float_as_uint(v);
                    const float r = v - __uint_as_float(fb & 0xffff0000u);
                    xb[plane * XS + rel] = u32x2{GAT_PERM(__float_as_uint(r), fb, 0x07060302u), 0u};
#else
                    const Split3 sp = split3(v);
                    xb[plane * XS + rel] = u32x2{GAT_PERM(sp.r, sp.v, 0x07060302u), GAT_PERM(sp.r2, sp.r2, 0x03020302u)};
#endif
                }
            };
            if constexpr (PLANAR) {
#pragma unroll
                for (int u = 0; u < 4; ++u) put(row, 4 * q + u, xv[it][u]);
            } else if constexpr (FMT == GAT_LAYOUT_INTERLEAVED) { // (re0, im0, re1, im1)
#pragma unroll
                for (int u = 0; u < 4; ++u) put(2 * row + (u & 1), 2 * q + (u >> 1), xv[it][u]);
            } else if constexpr (FMT == GAT_LAYOUT_INTERLEAVED_I16) { // dword u = sample u: {re: low half, im: high half}
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float f = xv[it][u];
                    const int w = (int)__float_as_uint(f);
                    put(2 * row, 4 * q + u, (float)(short)(w & 0xffff));
                    put(2 * row + 1, 4 * q + u, (float)(w >> 16));
                }
            } else { // int8: dword u = samples 2u, 2u+1: bytes (re0, im0, re1, im1)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float f = xv[it][u];
                    const int w = (int)__float_as_uint(f);
                    put(2 * row, 8 * q + 2 * u, (float)(signed char)(w & 0xff));
                    put(2 * row + 1, 8 * q + 2 * u, (float)(signed char)((w >> 8) & 0xff));
                    put(2 * row, 8 * q + 2 * u + 1, (float)(signed char)((w >> 16) & 0xff));
                    put(2 * row + 1, 8 * q + 2 * u + 1, (float)(w >> 24));
                }
            }
        }
    };
    // replica sign masks for entries e0, e0 + 1 of one slot (entry e <-> code sample nb + shifts[0] + e).
    // Every entry is the reference's FP64 expression, unfused (src/algorithms.jl:179).  Table index: the
    // full floored modulo only when `anchor`; otherwise it follows from the exact floor differences
    // (chips advance monotonically): ip - ip_prev chips on from the index of T samples ago
    // (c.inc_ok: fewer than Lc chips per step), and the pair's second entry (dist samples on) at most two wraps on.
    auto gen_rep2 = [&](const ChanInfoB &c, const unsigned *tab, unsigned *row, int win, int nb, int e0, int dist, int e_end, bool anchor,
                        int &ip_state, int &t_state) {
        const int x0 = nb + a.shifts[0] + e0;
        const double p0 = code_phase(c.ratio, c.tau, x0);
        const double p1 = code_phase(c.ratio, c.tau, x0 + dist);
        const int ip0 = (int)__builtin_floor(p0);
        const int ip1 = (int)__builtin_floor(p1);
        int t0;
        if (anchor) {
            t0 = floormod_fast(ip0, Lc, inv_lc);
        } else {
            t0 = t_state + (ip0 - ip_state);
            t0 -= (t0 >= Lc) ? Lc : 0;
        }
        ip_state = ip0;
        t_state = t0;
        int t1 = t0 + (ip1 - ip0);
        t1 -= (t1 >= Lc) ? Lc : 0;
        if (dist > 1) t1 -= (t1 >= Lc) ? Lc : 0; // up to T/2 = 64 samples on: fewer than 2 Lc chips (ratio * 32 < Lc)
        // both table words first, then both masks: two LDS round trips in flight together, not one after the other
        const unsigned w0 = tab[t0 >> 5], w1 = tab[t1 >> 5];
        const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe(w0, t0 & 31, 1) & 0x80008000u; // bit set: chip -1
        const unsigned m1 = (unsigned)__builtin_amdgcn_sbfe(w1, t1 & 31, 1) & 0x80008000u;
        auto put = [&](int e, unsigned m) { // window entry e -> ring position (and its copy behind the ring's end)
            if constexpr (!RINGREP) {
                row[e] = m;
                return;
            }
            int q = win + e;
            q -= q >= RR ? RR : 0;
            row[q] = m;
            if (q < WIN) row[RR + q] = m;
        };
        put(e0, m0);
        if (e0 + dist < e_end) put(e0 + dist, m1);
    };
    // one item = (slot, 2 consecutive samples): carrier fragments + the 2 new replica entries.  A producer
    // thread owns the same item in every step (at most one: the planner keeps nslots * T / 2 <= PT) and
    // carries its phasor and code index from step to step: one complex rotation by e^{j 2 pi T step}
    // instead of an FP64 range reduction + sincos; re-anchored in FP64 every kReanchor steps.
    float car_r = 0.f, car_i = 0.f;
    int rep_ip = 0, rep_t = 0, rep_ip2 = 0, rep_t2 = 0;
    // (2-term path: the items of 16 consecutive lanes are two slots 2 apart -- rows 4 apart, complementary banks, as for X)
    const int item_i = ptid / (T / IS), item_q = ptid % (T / IS);
    const int item_slot = X2 ? ((item_i & ~3) | ((item_i & 1) << 1) | ((item_i >> 1) & 1)) : item_i;
    const int item_s0 = X2 ? IS * item_q : X1 ? 2 * item_q : item_q; // the item's first sample (step-relative); its second one is OS further
    const bool have_item = producer && item_slot < nslots;
    int p_win = 0; // ring position of the window of the step being produced (wave-uniform)
    auto produce = [&](int st, int buf, bool first) { // everything of step st except the samples
        const int nb = st * T;
        const int win = p_win;
        if constexpr (RINGREP) p_win = ring_next(p_win);
        u32x2 *wb = X2 ? s_w + buf * wrows * (RB / 8) : s_w + buf * wrows * WS * WE;
        unsigned *rb = RINGREP ? s_rep : s_rep + buf * nslots * RS;
        const ChanInfoB c = s_chan[have_item ? item_slot : 0]; // fetched first: in flight behind what follows
        if (first) { // entries [0, span): ceil(span / 2) pairs per slot
            const int gps = (span + 1) >> 1;
            for (int id = ptid; id < nslots * gps; id += PT) {
                const int slot = id / gps, g = id - slot * gps;
                const ChanInfoB cs = s_chan[slot];
                if (!cs.valid) continue;
                int ip_unused, t_unused;
                gen_rep2(cs, s_code + slot * a.code_bits_stride, rb + slot * RS, win, nb, 2 * g, 1, span, true, ip_unused, t_unused);
            }
        } else if constexpr (!RINGREP) { // the overlap with the previous step is already known: from the other buffer
            const unsigned *rprev = s_rep + (buf ^ 1) * nslots * RS;
            const int pw = wave - NCW;
            for (int slot = pw; slot < nslots; slot += PT / 64)
                for (int e = lane; e < span; e += 64) rb[slot * RS + e] = rprev[slot * RS + e + T];
        } // (ring: the overlap with the previous step is in place already)
        if (!have_item) return;
        if (!c.valid) return;
        const bool anchor = first || ((st - s_begin) % kReanchor) == 0; // wave-uniform
#if defined(GAT_ABLATE) && (GAT_ABLATE & 1) // diagnostic builds (ablate_mfma_bf16.sh of an earlier round: git history; results wrong on purpose): no replica
        if (first)
#endif
        {
            gen_rep2(c, s_code + item_slot * a.code_bits_stride, rb + item_slot * RS, win, nb, span + item_s0, OS, span + T,
                     anchor || !c.inc_ok, rep_ip, rep_t);
            if constexpr (X2 && IS == 4) // the item's samples 2 and 3
                gen_rep2(c, s_code + item_slot * a.code_bits_stride, rb + item_slot * RS, win, nb, span + item_s0 + 2, 1, span + T,
                         anchor || !c.inc_ok, rep_ip2, rep_t2);
        }
#if defined(GAT_ABLATE) && (GAT_ABLATE & 2) // diagnostic: carrier fragments only in the first step
        if (!first) return;
#endif
        float cr, ci;
        if (anchor) {
            const double th = __builtin_fma((double)(nb + item_s0), c.step, c.phi);
            sincos_cycles(th - __builtin_rint(th), cr, ci);
        } else { // T samples on from the previous step's first sample
            cr = __builtin_fmaf(car_r, c.wTr, -(car_i * c.wTi));
            ci = __builtin_fmaf(car_r, c.wTi, car_i * c.wTr);
        }
        car_r = cr;
        car_i = ci;
        if constexpr (X2 && IS == 2) { // slot order of a sample: h m l h m  (against a a a b b); two samples = five dwords
            const Split3 c0 = split3(cr), s0 = split3(-ci);
            const float tr = __builtin_fmaf(cr, c.wr, -(ci * c.wi));
            ci = __builtin_fmaf(cr, c.wi, ci * c.wr);
            cr = tr;
            const Split3 c1 = split3(cr), s1 = split3(-ci);
            unsigned *dre = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(wb) + (2 * item_slot) * RB + 20 * item_q);
            unsigned *dim = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(dre) + RB);
            dre[0] = GAT_P2(c0.r, c0.v); dre[1] = GAT_P2(c0.v, c0.r2); dre[2] = GAT_P2(c1.v, c0.r); dre[3] = GAT_P2(c1.r2, c1.r); dre[4] = GAT_P2(c1.r, c1.v);
            dim[0] = GAT_P2(s0.r, s0.v); dim[1] = GAT_P2(s0.v, s0.r2); dim[2] = GAT_P2(s1.v, s0.r); dim[3] = GAT_P2(s1.r2, s1.r); dim[4] = GAT_P2(s1.r, s1.v);
            return;
        } else if constexpr (X2) {
            unsigned ch[4], cm[4], cl[4], sh[4], sm[4], sl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const Split3 sc = split3(cr), ss = split3(-ci);
                ch[u] = sc.v; cm[u] = sc.r; cl[u] = sc.r2;
                sh[u] = ss.v; sm[u] = ss.r; sl[u] = ss.r2;
                if (u < 3) {
                    const float tr = __builtin_fmaf(cr, c.wr, -(ci * c.wi));
                    ci = __builtin_fmaf(cr, c.wi, ci * c.wr);
                    cr = tr;
                }
            }
            auto put4 = [&](int wrow, const unsigned (&H)[4], const unsigned (&M)[4], const unsigned (&Lo)[4]) {
                u32x2 *d = reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned char *>(wb) + wrow * RB + 40 * item_q);
                d[0] = u32x2{GAT_P2(M[0], H[0]), GAT_P2(H[0], Lo[0])};
                d[1] = u32x2{GAT_P2(H[1], M[0]), GAT_P2(Lo[1], M[1])};
                d[2] = u32x2{GAT_P2(M[1], H[1]), GAT_P2(M[2], H[2])};
                d[3] = u32x2{GAT_P2(H[2], Lo[2]), GAT_P2(H[3], M[2])};
                d[4] = u32x2{GAT_P2(Lo[3], M[3]), GAT_P2(M[3], H[3])};
            };
            put4(2 * item_slot, ch, cm, cl);
            put4(2 * item_slot + 1, sh, sm, sl);
            return;
        }
        u32x2 *w_re = wb + ((2 * item_slot) * WS + item_s0) * WE;
        u32x2 *w_im = w_re + WS * WE;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const Split3 sc = split3(cr), ss = split3(-ci); // w_re = chip * cos, w_im = -chip * sin (conjugate)
            if constexpr (X1) { // {hi|mid, lo|0}
                w_re[u] = u32x2{GAT_PERM(sc.r, sc.v, 0x07060302u), sc.r2 >> 16}; // OS == 1
                w_im[u] = u32x2{GAT_PERM(ss.r, ss.v, 0x07060302u), ss.r2 >> 16};
            } else { // {hh, mm, ll, hm}
                const unsigned c_hh = GAT_PERM(sc.v, sc.v, 0x03020302u), c_mm = GAT_PERM(sc.r, sc.r, 0x03020302u),
                               c_ll = GAT_PERM(sc.r2, sc.r2, 0x03020302u), c_hm = GAT_PERM(sc.r, sc.v, 0x07060302u);
                const unsigned s_hh = GAT_PERM(ss.v, ss.v, 0x03020302u), s_mm = GAT_PERM(ss.r, ss.r, 0x03020302u),
                               s_ll = GAT_PERM(ss.r2, ss.r2, 0x03020302u), s_hm = GAT_PERM(ss.r, ss.v, 0x07060302u);
                *reinterpret_cast<u32x4 *>(w_re + 2 * OS * u) = u32x4{c_hh, c_mm, c_ll, c_hm};
                *reinterpret_cast<u32x4 *>(w_im + 2 * OS * u) = u32x4{s_hh, s_mm, s_ll, s_hm};
            }
            if (u == 0) {
                const float tr = __builtin_fmaf(cr, c.wr, -(ci * c.wi));
                ci = __builtin_fmaf(cr, c.wi, ci * c.wr);
                cr = tr;
            }
        }
    };

    // ---- consumer state: this lane's column of W ------------------------------------------------
    const int cw = producer ? 0 : wave;
    const int ctl = cw / WPT; // channel tile of this consumer wave within the workgroup
    const int sub = cw % WPT; // sample sub-range of the step
    const int r = lane & 31, h = lane >> 5;
    const int col = 32 * (NCT * cg + ctl) + r;           // flat column of this lane
    const int k_col = col / (2 * L), rem_col = col - 2 * L * k_col;
    const int l = rem_col >> 1, comp = rem_col & 1;      // (2L is even: comp == r & 1, partner column = lane ^ 1)
    const bool in_range = k_col < a.K && k_col - k_first < nslots;
    const int slot_c = in_range ? k_col - k_first : 0;
    const ChanInfoB my = s_chan[slot_c];
    const bool live_col = in_range && my.valid;
    const int col0 = sub * SW + h * (SW / 2); // first sample (step-relative) of this lane's stream
    // (2-term: w_off / x_off in BYTES; else in entries)
    const int w_off = X2 ? (live_col ? 2 * slot_c + comp : 2 * nslots) * RB + 10 * col0 : (live_col ? 2 * slot_c + comp : 2 * nslots) * WS + col0;
    const int r_off = slot_c * RS + (a.shifts[l] - a.shifts[0]) + col0;
    const int x_off = X2 ? r * RB + 10 * col0 : r * XS + col0;
    // bytes of one buffer's X rows / W rows
    constexpr int XBUF = X2 ? RT * 32 * RB : 8 * RT * 32 * XS, XTILE = X2 ? 32 * RB : 8 * 32 * XS, XU = X2 ? 1 : 8;
    const int WBUF = X2 ? wrows * RB : 8 * WE * wrows * WS, WU = X2 ? 1 : 8 * WE;

    // the consumers' few vector instructions per slice must not queue behind the producers' streams
    if (!producer) __builtin_amdgcn_s_setprio(3);

    constexpr int NA = RT == 1 ? 2 : RT; // accumulators (one row tile: two, summed at the end)
    f32x16 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // LDS byte addresses of this lane's streams (the low 32 bits of a flat LDS address are the LDS offset)
    const unsigned lds_x = (unsigned)(uintptr_t)s_x, lds_w = (unsigned)(uintptr_t)s_w, lds_r = (unsigned)(uintptr_t)s_rep;
    int c_win = 0; // ring position of the window of the step being consumed (wave-uniform)
    auto consume = [&](int buf) {
#if defined(GAT_ABLATE) && (GAT_ABLATE & 8) // diagnostic: no MFMA work
        return;
#endif
        unsigned x_addr[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) x_addr[t] = lds_x + (unsigned)(buf * XBUF + t * XTILE + XU * x_off);
        const unsigned w_addr = lds_w + (unsigned)(buf * WBUF + WU * w_off);
        const unsigned r_addr = lds_r + 4u * (unsigned)((RINGREP ? c_win : buf * nslots * RS) + r_off);
        if constexpr (RINGREP) c_win = ring_next(c_win);
        mfma_step<NM, RT, XM>(acc, w_addr, r_addr, x_addr, std::make_integer_sequence<int, NM>{});
    };

    // ---- pipeline ----------------------------------------------------------------------------------
#ifdef GAT_MFMA_STAMPS // per wave: [0] work, [1] barrier wait, [2] producers: carriers + replica, [3] producers: split/store
    unsigned long long t_work = 0, t_wait = 0, t_gen = 0, t_st = 0, t0_ = 0, t1_ = 0, t2_ = 0;
#define GAT_STAMP(v) v = __builtin_amdgcn_s_memtime()
#define GAT_ACC(dst, a_, b_) dst += (b_) - (a_)
#else
#define GAT_STAMP(v)
#define GAT_ACC(dst, a_, b_)
#endif
    // one instance of the producer loop per prefetch-group size: the counted waits need straight-line code
    auto producer_loop = [&](auto xi_tag) {
        constexpr int XI = decltype(xi_tag)::value;
        // three register sets: the loads of steps p+1 and p+2 stay in flight while step p is split and stored
        // (config 5: with two sets the consumers + bare loads took 1.32 ms against 1.06 ms without any load)
        f32x4_ xv0[XI ? XI : 1], xv1[XI ? XI : 1], xv2[XI ? XI : 1];
        if (s_begin < s_end) {
            if constexpr (XI > 0) {
                load_x(xv0, s_begin);
                load_x(xv1, s_begin + 1);
                load_x(xv2, s_begin + 2);
            }
            produce(s_begin, 0, true);
            if constexpr (XI > 0) {
                store_x(xv0, s_begin, 0);
                load_x(xv0, s_begin + 3);
            }
        }
        __syncthreads();
        // phase j of an iteration: the consumers work on step st + j (buffer j & 1) while step st + j + 1 is produced
        // into the other buffer from register set (j + 1) % 3 (the iteration advances by 6 = lcm(2 buffers, 3 sets))
        auto phase = [&](int step, int buf, auto &xv) {
            GAT_STAMP(t0_);
            if (step < s_end) {
                produce(step, buf, false);
                GAT_STAMP(t2_);
                if constexpr (XI > 0) {
                    store_x(xv, step, buf);
                    load_x(xv, step + 3);
                }
                GAT_STAMP(t1_);
                GAT_ACC(t_gen, t0_, t2_);
                GAT_ACC(t_st, t2_, t1_);
            }
            GAT_STAMP(t1_);
#if !(defined(GAT_ABLATE) && (GAT_ABLATE & 32))
            __syncthreads();
#endif
            GAT_STAMP(t2_);
            GAT_ACC(t_work, t0_, t1_);
            GAT_ACC(t_wait, t1_, t2_);
        };
        for (int st = s_begin; st < s_end; st += 6) {
            phase(st + 1, 1, xv1);
            if (st + 1 >= s_end) break;
            phase(st + 2, 0, xv2);
            if (st + 2 >= s_end) break;
            phase(st + 3, 1, xv0);
            if (st + 3 >= s_end) break;
            phase(st + 4, 0, xv1);
            if (st + 4 >= s_end) break;
            phase(st + 5, 1, xv2);
            if (st + 5 >= s_end) break;
            phase(st + 6, 0, xv0);
        }
    };
    // Role-specific loops (same barrier count): a shared loop would keep the producers' prefetch
    // registers alive in the consumers and the accumulators alive in the producers.
    if (__builtin_amdgcn_readfirstlane(wave) >= NCW) {
        switch (my_xi) {
        case 0: producer_loop(std::integral_constant<int, 0>{}); break;
        case 1: producer_loop(std::integral_constant<int, 1>{}); break;
        case 2: producer_loop(std::integral_constant<int, 2>{}); break;
        case 3: producer_loop(std::integral_constant<int, 3>{}); break;
        default: producer_loop(std::integral_constant<int, 4>{}); break;
        }
    } else {
#if defined(GAT_MB_STEP_PIPELINE_OFF) || defined(GAT_MFMA_STAMPS) || (defined(GAT_ABLATE) && (GAT_ABLATE & 8))
        __syncthreads();
        for (int st = s_begin; st < s_end; st += 2) {
            GAT_STAMP(t0_);
            consume(0);
            GAT_STAMP(t1_);
            __syncthreads();
            GAT_STAMP(t2_);
            GAT_ACC(t_work, t0_, t1_);
            GAT_ACC(t_wait, t1_, t2_);
            if (st + 1 >= s_end) break;
            GAT_STAMP(t0_);
            consume(1);
            GAT_STAMP(t1_);
            __syncthreads();
            GAT_STAMP(t2_);
            GAT_ACC(t_work, t0_, t1_);
            GAT_ACC(t_wait, t1_, t2_);
        }
#else
        // k-slices pipelined across the step barrier (xstep above): one barrier per step as before, inside the slice stream.
        // The ring position of a step's first slice advances by NM mod RING per step, so RING consecutive steps are
        // unrolled (each with its position as a constant: no run-time choice between code copies -- a merge of copies
        // would make the register allocator move fragment registers that are still being loaded); what is left over at the
        // end (fewer than RING steps) runs unpipelined.
        constexpr int D = frag_depth<RT, XM>(), RING = D + 1;
        static_assert(NM >= D, "a step holds at least as many slices as the fetch distance");
        const int nsteps = s_end - s_begin, groups = nsteps > 0 ? nsteps / RING : 0;
        // this lane's stream addresses in the CURRENT buffer only; the other buffer is a wave-uniform distance away (its
        // addresses are formed when the look-ahead fetches are issued: six persistent registers less)
        unsigned xa_c[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) xa_c[t] = lds_x + (unsigned)(t * XTILE + XU * x_off);
        unsigned wa_c = lds_w + (unsigned)(WU * w_off);
        unsigned ra_c = lds_r + 4u * (unsigned)r_off;
        int dx = XBUF, dw = __builtin_amdgcn_readfirstlane(WBUF), // bytes from the current buffer to the other one (sign flips per step)
            dr_flip = __builtin_amdgcn_readfirstlane(4 * (nslots * RS)); // (chip signs in two buffers: one-row-tile instances)
        __syncthreads(); // the first step's buffer is complete
        if (groups > 0) {
            FragSet<RT, XM> fs[RING];
            frag_issue<0, RT, XM>(fs[0], wa_c, ra_c, xa_c);
            if constexpr (D > 1) frag_issue<1, RT, XM>(fs[1], wa_c, ra_c, xa_c);
            if constexpr (D > 2) frag_issue<2, RT, XM>(fs[2], wa_c, ra_c, xa_c);
            if constexpr (D > 3) frag_issue<3, RT, XM>(fs[3], wa_c, ra_c, xa_c);
            auto one = [&](auto r_c) {
                constexpr int R = decltype(r_c)::value;
                // the chip-sign window moves T entries on along its ring, or back to the ring's start
                int dr = dr_flip;
                if constexpr (RINGREP) {
                    const int c_next = ring_next(c_win);
                    dr = __builtin_amdgcn_readfirstlane(4 * (c_next - c_win));
                    c_win = c_next;
                } else {
                    dr_flip = -dr_flip;
                }
                xstep<NM, RT, XM, R>(acc, fs, wa_c, ra_c, xa_c, dw, dr, dx, std::make_integer_sequence<int, NM>{});
                wa_c += (unsigned)dw; // the buffers change roles
                ra_c += (unsigned)dr;
#pragma unroll
                for (int t = 0; t < RT; ++t) xa_c[t] += (unsigned)dx;
                dw = -dw;
                dx = -dx;
            };
            for (int g = 0; g < groups; ++g) {
                one(std::integral_constant<int, 0>{});
                if constexpr (RING > 1) one(std::integral_constant<int, (1 * NM) % RING>{});
                if constexpr (RING > 2) one(std::integral_constant<int, (2 * NM) % RING>{});
                if constexpr (RING > 3) one(std::integral_constant<int, (3 * NM) % RING>{});
                if constexpr (RING > 4) one(std::integral_constant<int, (4 * NM) % RING>{});
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the last step's look-ahead reads (unused)
        }
        for (int st = s_begin + groups * RING; st < s_end; ++st) { // the remainder, unpipelined
            consume((st - s_begin) & 1);
            __syncthreads();
        }
#endif
    }
#ifdef GAT_MFMA_STAMPS
    if (lane == 0 && a.dbg) { // [workgroup][wave][4]
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 16 + wave) * 4;
        d[0] = t_work; d[1] = t_wait; d[2] = t_gen; d[3] = t_st;
    }
#endif

    if constexpr (RT == 1) acc[0] += acc[1];

    // ---- epilogue ------------------------------------------------------------------------------------
    if constexpr (WPT > 1) { // sum the consumer waves that shared a channel tile (the x staging area is free now)
        float *s_red = reinterpret_cast<float *>(s_x); // [4 waves][RT*16][64] floats = RT * 16 KB <= 2*RT*32*XS*8 B
        if (!producer && sub != 0) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) s_red[(((ctl * (WPT - 1) + sub - 1) * RT + t) * 16 + i) * 64 + lane] = acc[t][i];
        }
        __syncthreads();
        if (!producer && sub == 0) { // fixed order: own samples first, then sub-ranges 1 .. WPT-1 (deterministic)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
#pragma unroll
                for (int qq = 1; qq < WPT; ++qq)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += s_red[(((ctl * (WPT - 1) + qq - 1) * RT + t) * 16 + i) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0); // one row tile at a time: 16 loads in flight, not 64 * WPT
            }
        }
    }
    if (producer || sub != 0) return;
    // C[row][col]: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).  Rows 2m / 2m+1 are
    // registers i / i+1 of one lane; columns w_re / w_im are lanes c / c^1.
    const int k = k_col;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const float v0 = acc[t][i], v1 = acc[t][i + 1]; // x_re * w_comp, x_im * w_comp
            const float o1 = __shfl_xor(v1, 1, 64);         // partner column's x_im product
            float val = comp ? (v0 + o1) : (v0 - o1); // comp 0: R_re = xr*wr - xi*wi ; comp 1: R_im = xr*wi + xi*wr
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int m = (at * RT + t) * 16 + (row >> 1);
            if (live_col) {
                if (my.bad) val = __builtin_nanf("");
                const int lo = a.tap_index[l];
                const size_t bk = (size_t)b * a.K + k;
                const size_t o = (bk * a.L + lo) * a.M + m;
                float *dst = comp ? a.out_im : a.out_re;
                if (a.flags & GAT_FLAG_ATOMIC) {
                    atomicAdd(dst + o, val);
                } else if (a.splits == 1) {
                    dst[o] = val;
                } else {
                    const size_t elems = (size_t)a.L * a.M * 2;
                    a.partial[(bk * a.splits + split) * elems + ((size_t)lo * a.M + m) * 2 + comp] = val;
                }
            }
        }
    }
}

int mfma_bf16_tile_samples(int rt, int nct) { return tile_samples(rt, nct); }
int mfma_bf16_max_chain() { return kMaxChain; }
int mfma_bf16_max_slots() { return kMbMaxSlots; }
int mfma_bf16_threads(int rt, int nct) { return mb_threads(rt, nct); }
int mfma_bf16_producer_threads(int rt, int nct) { return mb_threads(rt, nct) - 64 * consumer_waves(rt, nct); }

int mfma_bf16_slots(int nct, int L, int K) { return mb_slots(nct, L, K); }

size_t mfma_bf16_lds_bytes(int rt, int nct, int fmt, int nslots, int rep_stride, int code_bits_stride, int mode)
{
    static_assert(sizeof(ChanInfoB) * kMbMaxSlots <= kHeader, "channel table must fit the header");
    return mb_lds_bytes(rt, nct, fmt, nslots, rep_stride, code_bits_stride, mode);
}
int mfma_bf16_mode(int rt, int nct, int fmt, bool force_three) { return mb_mode(rt, nct, fmt, force_three); }

namespace {
// the operand split of the launch: the host's a.mb_mode, which for int16 may be the three-term path on request
template <int RT, int NCT, int F>
hipError_t launch_one(const MfArgs &a, dim3 g, dim3 blk, unsigned lds_bytes, hipStream_t s)
{
    constexpr int natural = F == GAT_LAYOUT_INTERLEAVED_I8 ? kMbOne : kMbThree;
    if constexpr (mb_mode(RT, NCT, F) == kMbTwo) {
        if (a.mb_mode == kMbTwo) {
            hipLaunchKernelGGL((mfma_bf16_kernel<RT, NCT, F, kMbTwo>), g, blk, lds_bytes, s, a);
            return hipSuccess;
        }
    }
    if (a.mb_mode != natural) return hipErrorInvalidValue;
    hipLaunchKernelGGL((mfma_bf16_kernel<RT, NCT, F, natural>), g, blk, lds_bytes, s, a);
    return hipSuccess;
}
} // namespace

hipError_t launch_mfma_bf16(const MfArgs &a, int rt, int nct, int fmt, unsigned grid, unsigned lds_bytes, hipStream_t s)
{
    const dim3 g(grid), blk(mb_threads(rt, nct));
    hipError_t e = hipSuccess;
#define GAT_MB1(RT_, NCT_, F_) \
    case (F_ * 8 + RT_) * 8 + NCT_: e = launch_one<RT_, NCT_, F_>(a, g, blk, lds_bytes, s); break;
#define GAT_MB(F_)                                                                                                    \
    GAT_MB1(1, 1, F_) GAT_MB1(1, 2, F_) GAT_MB1(1, 4, F_) GAT_MB1(2, 1, F_) GAT_MB1(2, 2, F_) GAT_MB1(2, 4, F_) GAT_MB1(4, 2, F_) \
        GAT_MB1(4, 4, F_)
    switch ((fmt * 8 + rt) * 8 + nct) {
        GAT_MB(GAT_LAYOUT_PLANAR)
        GAT_MB(GAT_LAYOUT_INTERLEAVED)
        GAT_MB(GAT_LAYOUT_INTERLEAVED_I16)
        GAT_MB(GAT_LAYOUT_INTERLEAVED_I8)
    default: return hipErrorInvalidValue;
    }
#undef GAT_MB
#undef GAT_MB1
    return e != hipSuccess ? e : hipGetLastError();
}

} // namespace gat
