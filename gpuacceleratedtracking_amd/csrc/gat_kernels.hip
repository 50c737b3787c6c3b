// gat_kernels.hip -- gfx950 (MI355X, wave64) kernels of the downconvert + correlate path.
//
// What is computed (reference: downconvert_and_correlate_kernel_1330!, src/algorithms.jl:170-187;
// replica convention of kernel 5431, src/algorithms.jl:752-758; equation paper/paper.tex:48-52):
//
//   R[m,l,k,b] = sum_n x[n,m,b] * conj(exp(j2pi(n*f/fs + phi))) * c_k[floor(fc/fs*(n+shift_l)+tau) mod Lc]
//
// How (CDNA4-first, not the reference's shared-memory tree per sample):
//   * one workgroup (4 waves) streams a contiguous run of one integration block; every lane
//     owns VEC consecutive samples per step and loads them as 16-byte vectors per antenna
//     plane (planar) or 2 x 16 B (interleaved ComplexF32) -- 1 KiB per wave-instruction;
//   * the +-1 chip table of the workgroup's PRN lives in LDS as int8 (1 KB for C/A, 10 KB L5);
//   * carrier: one double-precision phase anchor per lane and step, reduced to an octant in
//     double, float polynomial sincos, then VEC-1 complex rotations -- no per-(antenna,tap)
//     redundant FP64 sincos as in the reference (src/algorithms.jl:172);
//   * code phase: the reference's exact double-precision expression, unfused (this file is
//     built with -ffp-contract=off) so chip edges fall on the same sample as on the CPU;
//   * MT x L complex accumulators stay in registers for the whole run; ONE reduction per
//     workgroup: a butterfly that halves the value count at each of the 6 wave64 shuffle
//     steps (2*MT*L -> 1 value per lane), then 4 waves through LDS;
//   * the result is written once (deterministic).  When a block is split over several
//     workgroups (small batch), per-split partials are summed by finalize_kernel in fixed
//     order; GAT_FLAG_ATOMIC uses float atomics instead (reference alg. 4/5).
#include "gat_internal.h"

namespace gat {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-byte streaming load.  The signal is read exactly once, so the loads are non-temporal
// (global_load_dwordx4 ... nt): measured +7 % on a pure read of this access pattern
// (scripts/bw_probe.hip: 6.13 -> 6.58 TB/s on MI355X).
__device__ __forceinline__ f32x4 load_stream16(const float *p)
{
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
}

// exp(j*2*pi*theta) for theta in cycles (double).  Octant reduction in double (exact), float
// Taylor polynomials on |a| <= pi/4 (|err| < 3e-8), quadrant fix-up.
__device__ __forceinline__ void sincos_cycles(double theta, float &c, float &s)
{
    const double q = __builtin_rint(theta * 4.0);
    const double r = __builtin_fma(q, -0.25, theta); // exact: |r| <= 0.125 cycles
    const float a = (float)r * 6.283185307179586f;
    const float a2 = a * a;
    float sp = __builtin_fmaf(a2, 2.7557319e-6f, -1.9841270e-4f);
    sp = __builtin_fmaf(a2, sp, 8.3333333e-3f);
    sp = __builtin_fmaf(a2, sp, -1.6666667e-1f);
    sp = __builtin_fmaf(a2 * a, sp, a);
    float cp = __builtin_fmaf(a2, 2.4801587e-5f, -1.3888889e-3f);
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
// (checked on the host and again per workgroup in dc_kernel; other callers clamp).
__device__ __forceinline__ int floormod_fast(int ip, int Lc, float inv_lc)
{
    const float q = __builtin_floorf((float)ip * inv_lc);
    int r = ip - (int)q * Lc;
    r += (r < 0) ? Lc : 0;
    r -= (r >= Lc) ? Lc : 0;
    return r;
}

// chip index of sample x = n + shift: the reference's expression, src/algorithms.jl:179-182.
// One double multiply and one double add, NOT fused (bit-identical to the CPU oracle).
__device__ __forceinline__ int chip_index(double ratio, double tau, int x, int Lc, float inv_lc)
{
    const double p = __dadd_rn(__dmul_rn(ratio, (double)x), tau);
    const int ip = (int)__builtin_floor(p);
    return floormod_fast(ip, Lc, inv_lc);
}

// Butterfly reduce-scatter over one wave64: NV per-lane values -> after 6 steps each lane
// holds the full wave sum of ONE value; 25 shuffles for NV = 24 instead of 144.  At a step with
// offset OFF, values 2i / 2i+1 are paired: the lane whose OFF bit is clear keeps 2i and sends
// 2i+1, its partner does the opposite; an odd leftover is all-reduced.  Which value a lane ends
// up with is a function of its lane id only (butterfly_index) -- no index array travels with the
// values (it would double the register footprint of the epilogue, the kernel's pressure peak).
template <int NV, int OFF>
struct Butterfly {
    static __device__ __forceinline__ void run(float *v, int lane)
    {
        constexpr int H = NV / 2;
        const bool up = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            // load both operands unconditionally: a ternary on the array elements themselves is
            // turned into a dynamically indexed (scratch) access by the compiler
            const float lo = v[2 * i], hi = v[2 * i + 1];
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            v[i] = keep + __shfl_xor(send, OFF, 64);
        }
        if constexpr (NV & 1) v[H] = v[NV - 1] + __shfl_xor(v[NV - 1], OFF, 64);
        Butterfly<(NV + 1) / 2, OFF / 2>::run(v, lane);
    }
    // original index of the value that ends in slot `slot` after this and all later steps
    static __device__ __forceinline__ int index(int lane)
    {
        const int j = Butterfly<(NV + 1) / 2, OFF / 2>::index(lane); // slot before the later steps
        constexpr int H = NV / 2;
        if ((NV & 1) && j == H) return NV - 1;
        return 2 * j + ((lane & OFF) ? 1 : 0);
    }
};
template <int NV>
struct Butterfly<NV, 0> {
    static __device__ __forceinline__ void run(float *, int) {}
    static __device__ __forceinline__ int index(int) { return 0; } // the survivor sits in slot 0
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// fused downconvert + correlate
// ------------------------------------------------------------------------------------------
// The chips of one step are generated ONCE per workgroup into an LDS replica segment
// [CHUNK + span of the taps] (what gen_code_replica! materialises in global memory in the
// reference, src/algorithms.jl:752-758) and every tap reads it at its own offset, so a step costs
// (CHUNK + span)/256 = ~4.1 FP64 code-phase evaluations per lane instead of samples*taps = 12.
// The segment is stored as 4 interleaved planes (element i at plane i&3, slot i>>2) so that
// the lanes of a wave, which own samples 4*lane + j, read consecutive dwords (no bank conflict).
//
// FMT: sample format of the signal (GAT_LAYOUT_*): planar f32, interleaved ComplexF32,
// interleaved int16 pairs, interleaved int8 pairs.  VEC = 4: one 16-byte non-temporal load per
// lane, plane and group (4 / 2 / 4 / 8 complex samples); VEC = 1: scalar loads (unaligned input).
typedef int i32x4 __attribute__((ext_vector_type(4)));
// sample loads: non-temporal when every byte is used once (one channel per signal: + 7 % at configs[1]), plain when
// the K channel workgroups of a tile are to find it in L2 (+ 4-9 % with 8-12 channels)
#define GAT_NT_LOAD(p) (KEEP ? *(p) : __builtin_nontemporal_load(p))

template <int FMT>
struct SampleIO {
    // number of 16-byte vectors per antenna and group
    static constexpr int NV = (FMT == GAT_LAYOUT_PLANAR) ? 2 : 1;
    // bytes of one complex sample
    static constexpr int BYTES = (FMT == GAT_LAYOUT_INTERLEAVED_I16) ? 4 : (FMT == GAT_LAYOUT_INTERLEAVED_I8) ? 2 : 8;

    // 16-byte loads of the group starting at complex-sample index e
    template <bool KEEP>
    static __device__ __forceinline__ void load16(i32x4 (&raw)[NV], const void *re, const void *im, size_t e)
    {
        if constexpr (FMT == GAT_LAYOUT_PLANAR) {
            raw[0] = GAT_NT_LOAD(reinterpret_cast<const i32x4 *>(static_cast<const float *>(re) + e));
            raw[1] = GAT_NT_LOAD(reinterpret_cast<const i32x4 *>(static_cast<const float *>(im) + e));
        } else {
            raw[0] = GAT_NT_LOAD(
                reinterpret_cast<const i32x4 *>(static_cast<const unsigned char *>(re) + e * BYTES));
        }
    }
    // sample j of a loaded group
    static __device__ __forceinline__ void get(const i32x4 (&raw)[NV], int j, float &xr, float &xi)
    {
        // NOTE: copy the vector element into a scalar BEFORE the bit cast: __builtin_bit_cast applied
        // directly to an ext-vector element lvalue reads element 0 whatever the index (hipcc 7.2).
        if constexpr (FMT == GAT_LAYOUT_PLANAR) {
            const int wr_ = raw[0][j], wi_ = raw[1][j];
            xr = __int_as_float(wr_);
            xi = __int_as_float(wi_);
        } else if constexpr (FMT == GAT_LAYOUT_INTERLEAVED) {
            const int wr_ = raw[0][2 * j], wi_ = raw[0][2 * j + 1];
            xr = __int_as_float(wr_);
            xi = __int_as_float(wi_);
        } else if constexpr (FMT == GAT_LAYOUT_INTERLEAVED_I16) {
            const int w = raw[0][j]; // {re: low half, im: high half}, little endian
            xr = (float)(short)(w & 0xffff);
            xi = (float)(w >> 16);
        } else {
            const int w = raw[0][j >> 1] >> ((j & 1) * 16); // two complex int8 samples per dword
            xr = (float)(signed char)(w & 0xff);
            xi = (float)(signed char)((w >> 8) & 0xff);
        }
    }
    // one sample with scalar loads
    static __device__ __forceinline__ void load1(const void *re, const void *im, size_t e, float &xr, float &xi)
    {
        if constexpr (FMT == GAT_LAYOUT_PLANAR) {
            xr = static_cast<const float *>(re)[e];
            xi = static_cast<const float *>(im)[e];
        } else if constexpr (FMT == GAT_LAYOUT_INTERLEAVED) {
            xr = static_cast<const float *>(re)[2 * e];
            xi = static_cast<const float *>(re)[2 * e + 1];
        } else if constexpr (FMT == GAT_LAYOUT_INTERLEAVED_I16) {
            xr = (float)static_cast<const short *>(re)[2 * e];
            xi = (float)static_cast<const short *>(re)[2 * e + 1];
        } else {
            xr = (float)static_cast<const signed char *>(re)[2 * e];
            xi = (float)static_cast<const signed char *>(re)[2 * e + 1];
        }
    }
};

// No minimum-waves bound on purpose: <4,3,4,planar> needs 136 VGPRs (3 waves/SIMD) and any tighter
// bound spills to scratch (measured: 4 waves/SIMD no gain at configs[1], 5 and 6 are 1.3-1.8x slower).
template <int MT, int L, int VEC, int FMT, bool KEEP>
__global__ void __launch_bounds__(kThreads) dc_kernel(const DcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int8_t *s_code = reinterpret_cast<int8_t *>(smem);
    float *s_part = reinterpret_cast<float *>(smem + ((a.Lc + 15) & ~15)); // [4][64]
    float *s_rep = s_part + 4 * 64;                                        // [2][4][rep_ps]
    using IO = SampleIO<FMT>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // Workgroup -> (tile, channel).  A tile = (block b, antenna tile, split): the bytes K channel
    // workgroups share.  Blocks id and id+8 land on the same XCD (round-robin dispatch), so the K
    // workgroups of one tile get ids tile%8 + 8*(k + K*(tile/8)): same XCD, dispatched back to
    // back -> the tile comes from HBM once and from that XCD's L2 for the other K-1 channels.
    // (Speed only: nothing depends on the placement.)
    const unsigned xcd = blockIdx.x & 7u, jq = blockIdx.x >> 3;
    const int k = (int)(jq % (unsigned)a.K);
    unsigned tile = (jq / (unsigned)a.K) * 8u + xcd;
    if (tile >= (unsigned)a.num_tiles) return; // padding of the last group of 8 (whole workgroup exits)
    const int split = tile % a.splits;
    tile /= a.splits;
    const int at = tile % a.ant_tiles;
    const int b = tile / a.ant_tiles;

    const gat_channel_params P = a.params[(size_t)b * a.K + k];
    const int Lc = a.Lc;
    const int N = (int)a.N;
    const double ratio = P.code_freq_hz / a.fs;    // src/algorithms.jl:179 (Float64 division)
    const double step = P.carrier_freq_hz / a.fs;  // cycles per sample
    const double tau = P.code_phase_chips;
    const double phi = P.carrier_phase_cycles;
    const float inv_lc = 1.0f / (float)Lc;

    // Parameters this kernel cannot evaluate exactly poison the output with NaN (fail loudly):
    // prn outside the table, or a code-phase span beyond the int32 / float-reciprocal modulo range
    // (the host entry point rejects these up front; device-resident parameters are checked here).
    const double span = __builtin_fabs(tau) + __builtin_fabs(ratio) * (double)(N + a.max_abs_shift) + 1.0;
    const bool bad = P.prn < 0 || P.prn >= a.num_prns || !(span < 1073741824.0) ||
                     !(span < 2097152.0 * (double)Lc) || !(ratio >= 0.0) || !(step == step) || !(phi == phi);
    const int prn = (P.prn < 0 || P.prn >= a.num_prns) ? 0 : P.prn;

    { // stage this PRN's chip table: rows are padded to 16 bytes on the device -> 16-byte copies
        const i32x4 *g = reinterpret_cast<const i32x4 *>(a.codes + (size_t)prn * a.code_row_stride);
        i32x4 *d = reinterpret_cast<i32x4 *>(s_code);
        for (int i = tid; i < (Lc + 15) / 16; i += kThreads) d[i] = g[i];
    }
    __syncthreads();

    float wr, wi; // one-sample rotation exp(+j*2*pi*step)
    sincos_cycles(step - __builtin_rint(step), wr, wi);

    f32x2 acc[MT][L]; // (re, im) pairs: one v_pk_fma_f32 per antenna, tap and sample
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int l = 0; l < L; ++l) acc[m][l] = f32x2{0.f, 0.f};

    const size_t base = (size_t)b * a.block_stride + (size_t)k * a.chan_stride +
                        (size_t)(at * MT) * a.ant_stride;
    // A lane owns G groups of S consecutive samples per step; one group = one 16-byte load per
    // plane, so every wave-instruction covers 1 KiB of contiguous memory in every format.
    constexpr int S = dc_group_samples(VEC, FMT);
    constexpr int G = dc_groups(VEC, FMT);
    constexpr int GSTRIDE = kThreads * S;
    constexpr int CHUNK = GSTRIDE * G;
    static_assert(CHUNK == dc_chunk(VEC, FMT), "host and device disagree on the chunk size");
    const int c_begin = split * a.chunks_per_split;
    const int c_end = bad ? c_begin : min(c_begin + a.chunks_per_split, a.total_chunks);
    const int shift0 = a.shifts[0];
    const int rep_ps = a.rep_plane_stride;
    const int rep_cnt = CHUNK + a.rep_span; // entries of one replica segment

    // chips of the sample at chunk-relative position rel, for the L taps
    auto get_chips = [&](float (&chip)[L], int rel, const float *rep) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int i = rel + (a.shifts[l] - shift0);
            chip[l] = rep[(i & 3) * rep_ps + (i >> 2)];
        }
    };
    // one sample of one antenna: conj(carrier) wipe-off (src/algorithms.jl:175-176), L taps.
    // (re, im) += chip * (dr, di) written on 2-vectors: ONE v_pk_fma_f32 with the chip broadcast by
    // op_sel_hi.  (With separate re / im accumulator arrays the vectoriser also gets to v_pk_fma_f32 but
    // pairs its operands with v_mov first -- more moves than FMAs; with several channels per signal
    // byte this kernel is bound by vector issue.)
    auto accumulate = [&](int m, float xr, float xi, float cr, float ci, const float (&chip)[L]) {
        const f32x2 dw = {__builtin_fmaf(xr, cr, xi * ci), __builtin_fmaf(xi, cr, -(xr * ci))};
#pragma unroll
        for (int l = 0; l < L; ++l) acc[m][l] = __builtin_elementwise_fma(f32x2{chip[l], chip[l]}, dw, acc[m][l]);
    };
    auto load_group = [&](i32x4 (&raw)[MT][IO::NV], int n) {
#pragma unroll
        for (int m = 0; m < MT; ++m) IO::template load16<KEEP>(raw[m], a.re, a.im, base + (size_t)m * a.ant_stride + n);
    };
    // S consecutive samples starting at n: one FP64 carrier anchor, then S-1 rotations
    auto process_group = [&](const i32x4 (&raw)[MT][IO::NV], int n, int rel, const float *rep) {
        float cr, ci;
        const double th0 = __builtin_fma((double)n, step, phi);
        sincos_cycles(th0 - __builtin_rint(th0), cr, ci);
#pragma unroll
        for (int j = 0; j < S; ++j) {
            float chip[L];
            get_chips(chip, rel + j, rep);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float xr, xi;
                IO::get(raw[m], j, xr, xi);
                accumulate(m, xr, xi, cr, ci, chip);
            }
            if (j + 1 < S) {
                const float t = __builtin_fmaf(cr, wr, -(ci * wi));
                ci = __builtin_fmaf(cr, wi, ci * wr);
                cr = t;
            }
        }
    };
    // samples [n_lo, n_hi) one at a time with scalar loads (ragged block end, unaligned input)
    auto scalar_run = [&](int n_lo, int n_hi, int rel, const float *rep) {
        for (int n = n_lo; n < n_hi; ++n, ++rel) {
            const double th = __builtin_fma((double)n, step, phi);
            float cr, ci, chip[L];
            sincos_cycles(th - __builtin_rint(th), cr, ci);
            get_chips(chip, rel, rep);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float xr, xi;
                IO::load1(a.re, a.im, base + (size_t)m * a.ant_stride + n, xr, xi);
                accumulate(m, xr, xi, cr, ci, chip);
            }
        }
    };
    // this step's replica segment: entry i <-> sample c*CHUNK + shift0 + i (src/algorithms.jl:753-757)
    auto fill_replica = [&](float *rep, int c) {
        const int x0 = c * CHUNK + shift0;
        // not unrolled on purpose: the FP64 temporaries of 4-5 unrolled iterations cost ~30 VGPRs,
        // i.e. one wave per SIMD of occupancy, and this loop runs in the shadow of the sample loads
#pragma unroll 1
        for (int i = tid; i < rep_cnt; i += kThreads)
            rep[(i & 3) * rep_ps + (i >> 2)] = (float)s_code[chip_index(ratio, tau, x0 + i, Lc, inv_lc)];
    };

    for (int c = c_begin; c < c_end; ++c) {
        const int rel0 = tid * S;
        const int cb = c * CHUNK + rel0;
        float *rep = s_rep + ((c - c_begin) & 1) * 4 * rep_ps; // double-buffered: one barrier per step
        if (VEC == 4 && c * CHUNK + CHUNK <= N) { // whole chunk inside the block: issue every load first
            i32x4 raw[G][MT][IO::NV];
#pragma unroll
            for (int g = 0; g < G; ++g) load_group(raw[g], cb + g * GSTRIDE);
            fill_replica(rep, c); // generated while the loads are in flight
            __syncthreads();
#pragma unroll
            for (int g = 0; g < G; ++g) process_group(raw[g], cb + g * GSTRIDE, rel0 + g * GSTRIDE, rep);
        } else {
            fill_replica(rep, c);
            __syncthreads();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int n = cb + g * GSTRIDE;
                if (VEC == 4 && n + S <= N) {
                    i32x4 raw[MT][IO::NV];
                    load_group(raw, n);
                    process_group(raw, n, rel0 + g * GSTRIDE, rep);
                } else if (n < N) {
                    scalar_run(n, min(n + S, N), rel0 + g * GSTRIDE, rep);
                }
            }
        }
    }

    // ---- workgroup reduction: 2*MT*L values, id = (l*MT + m)*2 + {0: re, 1: im} ----------
    constexpr int NV = 2 * MT * L;
    static_assert(NV <= 64, "one value per lane after the butterfly");
    float v[NV];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            v[(l * MT + m) * 2 + 0] = acc[m][l][0];
            v[(l * MT + m) * 2 + 1] = acc[m][l][1];
        }
    Butterfly<NV, 32>::run(v, lane);
    s_part[wave * 64 + Butterfly<NV, 32>::index(lane)] = v[0]; // lanes sharing an index hold bit-identical sums
    __syncthreads();

    if (tid < NV) {
        float tot = (s_part[tid] + s_part[64 + tid]) + (s_part[128 + tid] + s_part[192 + tid]);
        if (bad) tot = __builtin_nanf("");
        const int comp = tid & 1;
        const int ml = tid >> 1;
        const int m = at * MT + (ml % MT);
        const int l = a.tap_index[ml / MT]; // position of this tap in the caller's shift list
        const size_t bk = (size_t)b * a.K + k;
        const size_t o = (bk * a.Ltot + l) * a.M + m;
        if (a.flags & GAT_FLAG_ATOMIC) {
            atomicAdd((comp ? a.out_im : a.out_re) + o, tot);
        } else if (a.splits == 1) {
            (comp ? a.out_im : a.out_re)[o] = tot;
        } else {
            const size_t elems = (size_t)a.Ltot * a.M * 2;
            a.partial[(bk * a.splits + split) * elems + ((size_t)l * a.M + m) * 2 + comp] = tot;
        }
    }
}

// second stage: sum of the per-split partials in a FIXED order (deterministic): one wave64 per
// output element, lane i adds splits i, i+64, ... sequentially, then a fixed butterfly.
// partial [groups][splits][elems], elems = cols*2 with re/im interleaved innermost.
__global__ void __launch_bounds__(kThreads)
finalize_kernel(const float *__restrict__ partial, float *__restrict__ out_re,
                float *__restrict__ out_im, int splits, int elems, long long groups)
{
    const long long wave_id = (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (wave_id >= groups * elems) return;
    const int lane = threadIdx.x & 63;
    const long long g = wave_id / elems;
    const int e = (int)(wave_id - g * elems);
    const float *p = partial + (size_t)g * splits * elems + e;
    float s = 0.f;
    for (int i = lane; i < splits; i += 64) s += p[(size_t)i * elems];
    s = wave_sum(s);
    if (lane == 0) {
        float *o = (e & 1) ? out_im : out_re;
        o[(size_t)g * (elems / 2) + (e >> 1)] = s;
    }
}

// ------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------

// gen_code_replica_kernel! (src/algorithms.jl:13-32), grid-stride as _strided_ (:34-54).
// F32COORD = true emulates the reference's texture-memory variants
// (gen_code_replica_texture_mem_kernel!, src/algorithms.jl:121-140): the code phase is divided by
// the code length in Float64, rounded to a Float32 NORMALISED coordinate, wrapped and scaled back
// in Float32 with nearest-texel (floor) addressing -- the arithmetic that gives the texture path
// its code-phase error (paper/paper.tex:318-331).  It is an emulation of float32 normalised-
// coordinate addressing, not of a particular texture unit.
template <bool F32COORD>
__global__ void __launch_bounds__(kThreads)
code_replica_kernel(float *__restrict__ rep, long long count, const int8_t *__restrict__ code,
                    int Lc, double fc, double fs, double tau, long long first_shift)
{
    const double ratio = fc / fs;
    const float inv_lc = 1.0f / (float)Lc;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count;
         i += (long long)gridDim.x * kThreads) {
        int idx;
        if constexpr (F32COORD) {
            const double p = __dadd_rn(__dmul_rn(ratio, (double)(i + first_shift)), tau);
            const float u = (float)__ddiv_rn(p, (double)Lc);  // normalised coordinate, Float32
            const float w = u - __builtin_floorf(u);          // ADDRESS_MODE_WRAP
            idx = (int)__builtin_floorf(w * (float)Lc);       // NearestNeighbour == floor(u * N)
            idx = idx >= Lc ? Lc - 1 : (idx < 0 ? 0 : idx);
        } else {
            idx = chip_index(ratio, tau, (int)(i + first_shift), Lc, inv_lc);
        }
        rep[i] = (float)code[idx];
    }
}

// gen_signal! (src/gen_signal.jl:64-70, :86-90): Float64 code phase, carrier phase evaluated in
// Float64 then rounded to Float32 BEFORE cos/sin (src/gen_signal.jl:88), identical antennas.
// `amplitude` scales the sum (1 = the reference); integer formats store rint(value) saturated.
__global__ void __launch_bounds__(kThreads)
gen_signal_kernel(void *__restrict__ re_v, void *__restrict__ im_v, int format, long long N,
                  int M, long long ant_stride, long long block_stride, int K,
                  const gat_channel_params *__restrict__ params, const int8_t *__restrict__ codes,
                  int code_row_stride, int Lc, int num_prns, double fs, float amplitude)
{
    const int b = blockIdx.y;
    const float inv_lc = 1.0f / (float)Lc;
    for (long long n = (long long)blockIdx.x * kThreads + threadIdx.x; n < N;
         n += (long long)gridDim.x * kThreads) {
        float sr = 0.f, si = 0.f;
        for (int k = 0; k < K; ++k) {
            const gat_channel_params P = params[(size_t)b * K + k];
            const int prn = (P.prn < 0 || P.prn >= num_prns) ? 0 : P.prn;
            const double ratio = P.code_freq_hz / fs;
            const float chip = (float)codes[(size_t)prn * code_row_stride +
                                            chip_index(ratio, P.code_phase_chips, (int)n, Lc, inv_lc)];
            // 2pi * n * f / fs + phase, left to right as the reference broadcasts it
            const double ph64 = __dadd_rn(__ddiv_rn(__dmul_rn(__dmul_rn(6.283185307179586, (double)n), P.carrier_freq_hz), fs),
                                          P.carrier_phase_cycles /* radians here, see gat.h */);
            const float ph = (float)ph64;
            sr = __builtin_fmaf(cosf(ph), chip, sr);
            si = __builtin_fmaf(sinf(ph), chip, si);
        }
        sr *= amplitude;
        si *= amplitude;
        for (int m = 0; m < M; ++m) {
            const size_t e = (size_t)b * block_stride + (size_t)m * ant_stride + n;
            if (format == GAT_LAYOUT_PLANAR) {
                static_cast<float *>(re_v)[e] = sr;
                static_cast<float *>(im_v)[e] = si;
            } else if (format == GAT_LAYOUT_INTERLEAVED) {
                static_cast<float *>(re_v)[2 * e] = sr;
                static_cast<float *>(re_v)[2 * e + 1] = si;
            } else if (format == GAT_LAYOUT_INTERLEAVED_I16) {
                static_cast<short *>(re_v)[2 * e] = (short)fminf(fmaxf(rintf(sr), -32768.f), 32767.f);
                static_cast<short *>(re_v)[2 * e + 1] = (short)fminf(fmaxf(rintf(si), -32768.f), 32767.f);
            } else {
                static_cast<signed char *>(re_v)[2 * e] = (signed char)fminf(fmaxf(rintf(sr), -128.f), 127.f);
                static_cast<signed char *>(re_v)[2 * e + 1] = (signed char)fminf(fmaxf(rintf(si), -128.f), 127.f);
            }
        }
    }
}

// reduce_cplx_multi first pass (src/reduction.jl:331-403): per (chunk, column) partial sums.
// partial layout [chunks][cols*2] so that finalize_kernel produces the column sums.
__global__ void __launch_bounds__(kThreads)
reduce_stage1_kernel(const float *__restrict__ in_re, const float *__restrict__ in_im, long long n,
                     int cols, int chunks, float *__restrict__ partial)
{
    __shared__ float s[2][4];
    const int col = blockIdx.y;
    const int chunk = blockIdx.x;
    const long long per = (n + chunks - 1) / chunks;
    const long long lo = (long long)chunk * per;
    const long long hi = min(lo + per, n);
    const float *pr = in_re + (size_t)col * n;
    const float *pi = in_im + (size_t)col * n;
    float ar = 0.f, ai = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += kThreads) {
        ar += pr[i];
        ai += pi[i];
    }
    ar = wave_sum(ar);
    ai = wave_sum(ai);
    if ((threadIdx.x & 63) == 0) {
        s[0][threadIdx.x >> 6] = ar;
        s[1][threadIdx.x >> 6] = ai;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float t = (s[threadIdx.x][0] + s[threadIdx.x][1]) + (s[threadIdx.x][2] + s[threadIdx.x][3]);
        partial[(size_t)chunk * cols * 2 + (size_t)col * 2 + threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// closed tracking-loop step: discriminators + loop filters, one thread per channel (FP64; a few
// hundred flops per channel and block -- launch-bound, kept on the device to avoid a host round trip)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
tracking_update_kernel(const float *__restrict__ acc_re, const float *__restrict__ acc_im, int K, int M,
                       const gat_loop_config cfg, gat_loop_state *__restrict__ state,
                       const gat_channel_params *cur, gat_channel_params *next)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= K) return;
    const int L = cfg.num_taps;
    auto tap = [&](int l, double &re, double &im) {
        re = im = 0.0;
        for (int m = 0; m < M; ++m) {
            const size_t o = ((size_t)k * L + l) * M + m;
            re += (double)acc_re[o];
            im += (double)acc_im[o];
        }
    };
    double pr, pi, er, ei, lr, li;
    tap(cfg.prompt_index, pr, pi);
    tap(cfg.early_index, er, ei);
    tap(cfg.late_index, lr, li);
    gat_loop_state st = state[k];
    const gat_channel_params c = cur[k];
    const double T = cfg.block_seconds;

    // discriminators
    const double pll_err = (pr == 0.0 && pi == 0.0) ? 0.0 : atan(pi / pr) * 0.15915494309189535; // cycles
    const double e = sqrt(er * er + ei * ei), l = sqrt(lr * lr + li * li);
    // triangle autocorrelation: L - E = 2*eps, L + E = 2 - d  =>  eps = (2-d)/2 * (L-E)/(L+E) chips
    const double dll_err = (e + l > 0.0) ? 0.5 * (2.0 - cfg.early_late_spacing_chips) * (l - e) / (e + l) : 0.0;

    // 3rd-order bilinear PLL filter (Kaplan table 5.6: w0 = Bn/0.7845, a3 = 1.1, b3 = 2.4)
    const double w0p = cfg.pll_bandwidth_hz / 0.7845;
    const double in1 = w0p * w0p * w0p * pll_err;
    const double out1 = st.pll_acc1 + 0.5 * T * in1; // bilinear integrator 1
    st.pll_acc1 += T * in1;
    const double in2 = out1 + 1.1 * w0p * w0p * pll_err;
    const double out2 = st.pll_acc2 + 0.5 * T * in2; // bilinear integrator 2
    st.pll_acc2 += T * in2;
    const double carrier_rate = out2 + 2.4 * w0p * pll_err; // Hz correction
    // 2nd-order bilinear DLL filter (w0 = Bn/0.53, a2 = 1.414)
    const double w0d = cfg.dll_bandwidth_hz / 0.53;
    const double ind = w0d * w0d * dll_err;
    const double outd = st.dll_acc + 0.5 * T * ind;
    st.dll_acc += T * ind;
    const double code_rate = outd + 1.414 * w0d * dll_err; // chips/s correction

    const double carrier_doppler = st.init_carrier_doppler_hz + carrier_rate;
    const double code_doppler = code_rate + carrier_doppler * cfg.code_freq_nominal_hz / cfg.carrier_center_hz;

    // propagate the replica NCOs over the block that was just correlated, then retune
    gat_channel_params n = c;
    double phi = c.carrier_phase_cycles + c.carrier_freq_hz * T;
    phi -= floor(phi);
    double tau = c.code_phase_chips + c.code_freq_hz * T;
    tau -= floor(tau / (double)cfg.code_length) * (double)cfg.code_length;
    n.carrier_phase_cycles = phi;
    n.code_phase_chips = tau;
    n.carrier_freq_hz = cfg.if_hz + carrier_doppler;
    n.code_freq_hz = cfg.code_freq_nominal_hz + code_doppler;
    next[k] = n;

    st.carrier_doppler_hz = carrier_doppler; // filter output = correction on the initial estimate
    st.code_doppler_hz = code_doppler;
    st.last_pll_error_cycles = pll_err;
    st.last_dll_error_chips = dll_err;
    st.prompt_power = pr * pr + pi * pi;
    state[k] = st;
}

hipError_t launch_tracking_update(const float *acc_re, const float *acc_im, int K, int M, const gat_loop_config &cfg,
                                  gat_loop_state *state, const gat_channel_params *cur, gat_channel_params *next,
                                  hipStream_t s)
{
    hipLaunchKernelGGL(tracking_update_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, s, acc_re, acc_im, K, M,
                       cfg, state, cur, next);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
template <int MT, int L>
static hipError_t launch_dc_ml(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    const dim3 grid(cfg.grid), block(kThreads);
#define GAT_LAUNCH(V, F, K_) hipLaunchKernelGGL((dc_kernel<MT, L, V, F, K_>), grid, block, cfg.lds_bytes, s, a)
#define GAT_LAUNCH_V(F) do { if (cfg.vec == 4) GAT_LAUNCH(4, F, false); else GAT_LAUNCH(1, F, false); } while (0)
    // the L2-keeping variant exists for the 16-byte-load float formats only (instance count)
#define GAT_LAUNCH_VK(F) do { if (cfg.vec == 4 && cfg.keep_l2) GAT_LAUNCH(4, F, true); else GAT_LAUNCH_V(F); } while (0)
    switch (cfg.format) {
    case GAT_LAYOUT_PLANAR: GAT_LAUNCH_VK(GAT_LAYOUT_PLANAR); break;
    case GAT_LAYOUT_INTERLEAVED: GAT_LAUNCH_VK(GAT_LAYOUT_INTERLEAVED); break;
    case GAT_LAYOUT_INTERLEAVED_I16: GAT_LAUNCH_V(GAT_LAYOUT_INTERLEAVED_I16); break;
    case GAT_LAYOUT_INTERLEAVED_I8: GAT_LAUNCH_V(GAT_LAYOUT_INTERLEAVED_I8); break;
    default: return hipErrorInvalidValue;
    }
#undef GAT_LAUNCH_VK
#undef GAT_LAUNCH_V
#undef GAT_LAUNCH
    return hipGetLastError();
}

template <int MT>
static hipError_t launch_dc_m(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    switch (cfg.taps) {
    case 1: return launch_dc_ml<MT, 1>(a, cfg, s);
    case 2: return launch_dc_ml<MT, 2>(a, cfg, s);
    case 3: return launch_dc_ml<MT, 3>(a, cfg, s);
    case 4: return launch_dc_ml<MT, 4>(a, cfg, s);
    case 5: return launch_dc_ml<MT, 5>(a, cfg, s);
    case 6: return launch_dc_ml<MT, 6>(a, cfg, s);
    case 7: return launch_dc_ml<MT, 7>(a, cfg, s);
    case 8: return launch_dc_ml<MT, 8>(a, cfg, s);
    default: return hipErrorInvalidValue;
    }
}

bool dc_supported(int ant_tile, int taps)
{
    return ant_tile >= 1 && ant_tile <= kMaxAntTile && taps >= 1 && taps <= kMaxTapsPerLaunch;
}

hipError_t launch_dc(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    switch (cfg.ant_tile) {
    case 1: return launch_dc_m<1>(a, cfg, s);
    case 2: return launch_dc_m<2>(a, cfg, s);
    case 3: return launch_dc_m<3>(a, cfg, s);
    case 4: return launch_dc_m<4>(a, cfg, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize(const float *partial, float *out_re, float *out_im, int splits, int elems,
                           long long groups, hipStream_t s)
{
    const long long waves = groups * elems;
    const unsigned grid = (unsigned)((waves + kThreads / 64 - 1) / (kThreads / 64));
    hipLaunchKernelGGL(finalize_kernel, dim3(grid), dim3(kThreads), 0, s, partial, out_re, out_im,
                       splits, elems, groups);
    return hipGetLastError();
}

hipError_t launch_gen_code_replica(float *rep, long long count, const int8_t *code_row, int Lc,
                                   double fc, double fs, double tau, long long first_shift,
                                   bool f32_coordinates, hipStream_t s)
{
    long long blocks = (count + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    if (f32_coordinates)
        hipLaunchKernelGGL(code_replica_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, s, rep, count,
                           code_row, Lc, fc, fs, tau, first_shift);
    else
        hipLaunchKernelGGL(code_replica_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, s, rep, count,
                           code_row, Lc, fc, fs, tau, first_shift);
    return hipGetLastError();
}

hipError_t launch_gen_signal(void *re, void *im, int format, long long N, int M,
                             long long ant_stride, long long block_stride, int B, int K,
                             const gat_channel_params *params, const int8_t *codes, int code_row_stride,
                             int Lc, int num_prns, double fs, float amplitude, hipStream_t s)
{
    long long bx = (N + kThreads - 1) / kThreads;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(gen_signal_kernel, dim3((unsigned)bx, (unsigned)B), dim3(kThreads), 0, s, re,
                       im, format, N, M, ant_stride, block_stride, K, params, codes, code_row_stride, Lc,
                       num_prns, fs, amplitude);
    return hipGetLastError();
}

hipError_t launch_reduce_stage1(const float *in_re, const float *in_im, long long n, int cols,
                                int chunks, float *partial, hipStream_t s)
{
    hipLaunchKernelGGL(reduce_stage1_kernel, dim3((unsigned)chunks, (unsigned)cols), dim3(kThreads), 0,
                       s, in_re, in_im, n, cols, chunks, partial);
    return hipGetLastError();
}

} // namespace gat
