// gat_dc.h -- the fused downconvert + correlate kernel of libgat (vector path), gfx950 / wave64.
//
// What is computed (reference: downconvert_and_correlate_kernel_1330!, src/algorithms.jl:170-187;
// replica convention of kernel 5431, src/algorithms.jl:752-758; equation paper/paper.tex:48-52):
//
//   R[m,l,k,b] = sum_n x[n,m,b] * conj(exp(j2pi(n*f/fs + phi))) * c_k[floor(fc/fs*(n+shift_l)+tau) mod Lc]
//
// How (CDNA4-first, not the reference's shared-memory tree per sample):
//   * workgroup = 4 wave64 (or ONE, for short blocks in a long stream: NW below) on one (integration block, antenna
//     group, sample split) and KT channels.  Wave w owns
//     antenna tile (w % AW) -- MT <= 4 antennas -- and sample sub-chunk (w / AW): with AW = 4 the four waves walk the
//     SAME samples on 16 antennas and share one code replica;
//   * every lane owns S consecutive samples per group and loads them as ONE 16-byte buffer load per antenna plane
//     (1 KiB per wave-instruction in every sample format; one descriptor per plane, a 32-bit lane offset shared by all
//     loads of a step, a scalar offset per antenna; cache policy = template parameter: non-temporal when the tile is
//     read once, plain when several channel groups share it through L2);
//   * the samples of step c+1 are fetched while step c is consumed, antenna by antenna, into the registers that
//     antenna's samples have just left: loads are in flight all the time without a second register set; with KT > 1
//     the samples stay in registers while the workgroup loops over its channels (AW = 4 only: measured, a channel
//     loop over a single antenna tile never beat separate channel workgroups sharing the tile through L2);
//   * code replica: produced for a SEGMENT of up to 8 steps at once into LDS ([steps*CHUNK + tap span] chips per
//     channel), by an EXACT walk (gat_phase.h): one double-precision anchor per producer thread, then 32.32 fixed-point
//     steps, branch-free; a batch holding a step that is not proven equal to the reference's floor is re-evaluated with
//     the reference's expression -- bit-identical chip edges to the CPU oracle at ~1/2 of the vector instructions and
//     none of the double-precision ones of evaluating floor(ratio*(n+shift)+tau) per sample.  Two barriers per segment,
//     none in the step loop.  Stored linearly (entry i <-> sample + first tap + i): the chips of a lane's S consecutive
//     samples for one tap are ONE vector read (ds_read2_b64: 16 bytes at 8-byte alignment; consecutive lanes read
//     consecutive 16 bytes).  A tap at an odd distance from the first reads a second copy stored one entry further
//     (measured on gfx950: a ds_read_b128 at a 4-byte-aligned address works but takes 11x the cycles);
//   * carrier: one phasor per group, anchored in double precision at every segment start, carried from step to step
//     by one complex rotation, S-1 rotations inside the group; no per-(antenna, tap) double-precision sincos as in
//     the reference (src/algorithms.jl:172);
//   * KT x MT x L complex accumulators stay in registers for the whole block (plain scalar FMAs: the packed form
//     issues at the same rate and costs operand-pairing moves); ONE reduction per block: a butterfly that halves the
//     value count at each of the 6 wave64 shuffle steps, then the waves that share an antenna tile through LDS;
//   * the result is written once (deterministic).  When a block is split over several workgroups (small batch),
//     per-split partials are summed by finalize_kernel in fixed order; GAT_FLAG_ATOMIC uses float atomics instead
//     (reference alg. 4/5); short blocks: one workgroup walks several consecutive blocks (chip table staged once,
//     sample prefetch continues across the block boundary).
#pragma once

#include <type_traits>

#include "gat_internal.h"
#include "gat_phase.h"

namespace gat {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

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

// wave-uniform values computed on the vector ALU go back to scalar registers
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned uni(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ double uni(double x)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = uni((unsigned)b), hi = uni((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned long long uni(unsigned long long b)
{
    const unsigned lo = uni((unsigned)b), hi = uni((unsigned)(b >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// FMT: sample format of the signal (GAT_LAYOUT_*): planar f32, interleaved ComplexF32,
// interleaved int16 pairs, interleaved int8 pairs.  VEC = 4: one 16-byte load per
// lane, plane and group (4 / 2 / 4 / 8 complex samples); VEC = 1: scalar loads (unaligned input).
// Sample loads are non-temporal when every byte is used once (one workgroup per signal tile: + 7 % at configs[1]),
// plain when several channel groups are to find the tile in L2 (+ 4-9 % with 8-12 channels).
template <int FMT>
struct SampleIO {
    // number of 16-byte vectors per antenna and group
    static constexpr int NV = (FMT == GAT_LAYOUT_PLANAR) ? 2 : 1;
    // bytes of one complex sample
    static constexpr int BYTES = (FMT == GAT_LAYOUT_INTERLEAVED_I16) ? 4 : (FMT == GAT_LAYOUT_INTERLEAVED_I8) ? 2 : 8;

    template <bool KEEP>
    static __device__ __forceinline__ i32x4 ld(const i32x4 *p)
    {
        if constexpr (KEEP) return *p;
        else return __builtin_nontemporal_load(p);
    }
    // 16-byte loads of the group starting at complex-sample index e
    template <bool KEEP>
    static __device__ __forceinline__ void load16(i32x4 (&raw)[NV], const void *re, const void *im, size_t e)
    {
        if constexpr (FMT == GAT_LAYOUT_PLANAR) {
            raw[0] = ld<KEEP>(reinterpret_cast<const i32x4 *>(static_cast<const float *>(re) + e));
            raw[1] = ld<KEEP>(reinterpret_cast<const i32x4 *>(static_cast<const float *>(im) + e));
        } else {
            raw[0] = ld<KEEP>(reinterpret_cast<const i32x4 *>(static_cast<const unsigned char *>(re) + e * BYTES));
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
            // two complex int8 samples per dword; one sign-extending byte convert per component (v_cvt_f32_i32 with an
            // SDWA byte select).  The compiler finds bytes 0, 1 and 3 by itself but shifts the dword first for byte 2.
            const int w = raw[0][j >> 1];
            if (j & 1) {
                asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(xr) : "v"(w));
                xi = (float)(w >> 24);
            } else {
                xr = (float)(signed char)(w & 0xff);
                xi = (float)(signed char)((w >> 8) & 0xff);
            }
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


// The compiler sinks every kernel-argument load to its first use, so a kernel with several hundred bytes of arguments
// pays one scalar-cache miss per 64-byte line, one after the other, each a full trip to the runtime's argument buffer
// (set-up, replica, steps, tail: in a single-block call these serial trips were a third of the kernel's time,
// r03_latency_cuts.sh of an earlier round: git history; HIP_FORCE_DEV_KERNARG=1 changes nothing).  Touching every line at the kernel's first
// instruction overlaps them into one.
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch()
{
    typedef const unsigned __attribute__((address_space(4))) *kptr;
    kptr k = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) s |= __builtin_nontemporal_load(k + i * 16);
    asm volatile("" ::"s"(s));
}

// Completion flag (latency regime): the last workgroup of a launch to get here stores the call's sequence number into
// pinned host memory, where gat_sync spins on it -- a kernel's end reaches the host ~5 us sooner that way than through
// hipStreamSynchronize (scripts/probes/sync_probe.hip: 7.0 vs 11.7 us for an empty kernel).  Called by every thread of every
// workgroup that did work, after its result stores.
__device__ __forceinline__ void completion_flag(unsigned *done_counter, unsigned *host_flag, unsigned seq, unsigned total_wgs)
{
    if (!done_counter) return; // launch-uniform
    __syncthreads(); // every thread of this workgroup has issued its result stores
    if (total_wgs == 1u) { // nobody else to wait for: the release store alone (0.6 us less than with the counter, sync_probe2.hip)
        if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (threadIdx.x == 0) {
        // this XCD's L2 holds the workgroup's results: written back before it counts as arrived (the host may hand the
        // buffers to a copy engine or another stream as soon as it sees the flag)
        // (acquire-release on the counter: the last arriver's system-scope store below is then ordered after EVERY
        // workgroup's result stores, not only its own -- each arrival releases, the last one acquires them all)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned arrived = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == total_wgs - 1u) {
            __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Diagnostic builds (-DGAT_DC_LAT_CUT=n, r03_latency_cuts.sh of an earlier round: git history): the kernel ends at cut point n -- 1 entry, 2 block
// set-up + chip tables, 3 first replica segment + carrier anchors, 4 step loop -- so that the single-block latency can be
// attributed to its phases (5: + reduction up to its barrier, 6: everything but the result stores).  Results are wrong by
// construction; never part of the product build.
#ifdef GAT_DC_LAT_CUT
#define GAT_DC_LAT_CUT_AT(n)                                                                  \
    do {                                                                                      \
        if (GAT_DC_LAT_CUT == (n)) {                                                          \
            completion_flag(a.done_counter, a.host_flag, a.flag_seq, a.total_wgs);            \
            return;                                                                           \
        }                                                                                     \
    } while (0)
#else
#define GAT_DC_LAT_CUT_AT(n) do { } while (0)
#endif

template <int MT, int L, int VEC, int FMT, int AW, int KT, bool KEEP, int NW, int D>
__global__ void __launch_bounds__(64 * NW, dc_min_waves(MT, L, KT, D, FMT, AW)) dc_kernel(const DcArgs a)
{
#define GAT_DC_BODY_RESIDENT 0
#include "gat_dc_body.inc"
#undef GAT_DC_BODY_RESIDENT
    completion_flag(a.done_counter, a.host_flag, a.flag_seq, a.total_wgs);
}

// (which instances exist: dc_instance, gat_internal.h)

template <int FMT, int MT, int L, int VEC, int AW, int KT, int NW>
static hipError_t launch_dc_nw(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    if constexpr (dc_instance(MT, L, VEC, AW, KT, NW)) {
        // two sample sets: streaming regime of float samples only (non-temporal loads)
        if constexpr (dc_instance(MT, L, VEC, AW, KT, NW, 2) && (FMT == GAT_LAYOUT_PLANAR || FMT == GAT_LAYOUT_INTERLEAVED)) {
            if (cfg.depth == 2) {
                if (a.keep_l2) return hipErrorInvalidValue;
                hipLaunchKernelGGL((dc_kernel<MT, L, VEC, FMT, AW, KT, false, NW, 2>), dim3(cfg.grid), dim3(64 * NW), cfg.lds_bytes, s, a);
                return hipGetLastError();
            }
        }
        if (cfg.depth != 1) return hipErrorInvalidValue;
        if (VEC == 4 && a.keep_l2)
            hipLaunchKernelGGL((dc_kernel<MT, L, VEC, FMT, AW, KT, VEC == 4, NW, 1>), dim3(cfg.grid), dim3(64 * NW), cfg.lds_bytes, s, a);
        else
            hipLaunchKernelGGL((dc_kernel<MT, L, VEC, FMT, AW, KT, false, NW, 1>), dim3(cfg.grid), dim3(64 * NW), cfg.lds_bytes, s, a);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

template <int FMT, int MT, int L, int VEC, int AW, int KT>
static hipError_t launch_dc_one(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    if (cfg.nw == 1) return launch_dc_nw<FMT, MT, L, VEC, AW, KT, 1>(a, cfg, s);
    return launch_dc_nw<FMT, MT, L, VEC, AW, KT, 4>(a, cfg, s);
}

template <int FMT, int MT, int L>
static hipError_t launch_dc_ml(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    if (cfg.vec != 4) return launch_dc_one<FMT, MT, L, 1, 1, 1>(a, cfg, s);
    switch (cfg.aw * 8 + cfg.kt) {
    case 1 * 8 + 1: return launch_dc_one<FMT, MT, L, 4, 1, 1>(a, cfg, s);
    case 1 * 8 + 2: return launch_dc_one<FMT, MT, L, 4, 1, 2>(a, cfg, s);
    case 1 * 8 + 4: return launch_dc_one<FMT, MT, L, 4, 1, 4>(a, cfg, s);
    case 2 * 8 + 1: return launch_dc_one<FMT, MT, L, 4, 2, 1>(a, cfg, s);
    case 2 * 8 + 2: return launch_dc_one<FMT, MT, L, 4, 2, 2>(a, cfg, s);
    case 4 * 8 + 1: return launch_dc_one<FMT, MT, L, 4, 4, 1>(a, cfg, s);
    case 4 * 8 + 2: return launch_dc_one<FMT, MT, L, 4, 4, 2>(a, cfg, s);
    case 4 * 8 + 4: return launch_dc_one<FMT, MT, L, 4, 4, 4>(a, cfg, s);
    default: return hipErrorInvalidValue;
    }
}

template <int FMT, int MT>
static hipError_t launch_dc_m(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    switch (cfg.taps) {
    case 1: return launch_dc_ml<FMT, MT, 1>(a, cfg, s);
    case 2: return launch_dc_ml<FMT, MT, 2>(a, cfg, s);
    case 3: return launch_dc_ml<FMT, MT, 3>(a, cfg, s);
    case 4: return launch_dc_ml<FMT, MT, 4>(a, cfg, s);
    case 5: return launch_dc_ml<FMT, MT, 5>(a, cfg, s);
    case 6: return launch_dc_ml<FMT, MT, 6>(a, cfg, s);
    case 7: return launch_dc_ml<FMT, MT, 7>(a, cfg, s);
    case 8: return launch_dc_ml<FMT, MT, 8>(a, cfg, s);
    default: return hipErrorInvalidValue;
    }
}

template <int FMT>
hipError_t launch_dc_fmt(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    switch (cfg.ant_tile) {
    case 1: return launch_dc_m<FMT, 1>(a, cfg, s);
    case 2: return launch_dc_m<FMT, 2>(a, cfg, s);
    case 3: return launch_dc_m<FMT, 3>(a, cfg, s);
    case 4: return launch_dc_m<FMT, 4>(a, cfg, s);
    default: return hipErrorInvalidValue;
    }
}

} // namespace gat
