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
// scripts/history/r03/r03_latency_cuts.sh; HIP_FORCE_DEV_KERNARG=1 changes nothing).  Touching every line at the kernel's first
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

// Diagnostic builds (-DGAT_DC_LAT_CUT=n, scripts/history/r03/r03_latency_cuts.sh): the kernel ends at cut point n -- 1 entry, 2 block
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
__global__ void __launch_bounds__(64 * NW, dc_min_waves(MT, L, KT, D, FMT)) dc_kernel(const DcArgs a)
{
    // NW = 4 waves per workgroup, or 1: short blocks in a long stream (a few steps per block) spend their time in the
    // per-block set-up, which all four waves of a workgroup repeat, and at its three barriers; a one-wave workgroup
    // does the set-up once per block and never waits for another wave.
    constexpr int T = 64 * NW;          // threads per workgroup
    using IO = SampleIO<FMT>;
    // A lane owns G groups of S consecutive samples per step; one group = one 16-byte load per plane.
    constexpr int S = dc_group_samples(VEC, FMT);
    constexpr int G = dc_groups(VEC, FMT);
    constexpr int SUBS = NW / AW;       // sample sub-chunks per workgroup (waves that share an antenna tile)
    constexpr int VT = 64 * SUBS;       // lanes that share an antenna tile
    constexpr int GSTRIDE = VT * S;
    constexpr int CHUNK = GSTRIDE * G;
    static_assert(CHUNK == dc_chunk(VEC, FMT, AW, NW), "host and device disagree on the chunk size");
    static_assert(NW == 4 || (NW == 1 && AW == 1 && KT == 1), "one-wave workgroups: one antenna tile, one channel");
    constexpr int RPC = T / KT;         // replica producer threads per channel (>= 64: a wave serves one channel)
    constexpr int NV = 2 * MT * L;      // values of one channel's reduction, id = (l*MT + m)*2 + {0: re, 1: im}
    constexpr int EB = (FMT == GAT_LAYOUT_PLANAR) ? 4 : IO::BYTES; // bytes per sample in one plane
    static_assert(NV <= 64, "one value per lane after the butterfly");
    static_assert(AW == 1 || AW == 2 || AW == 4, "antenna-tile waves");
    static_assert(KT == 1 || KT == 2 || KT == 4, "channels per workgroup");

    kernarg_prefetch<sizeof(DcArgs)>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    struct ChanConst { double ratio, tau, step, phi; }; // per channel slot: read at segment starts and on ragged ends
    ChanConst *s_const = reinterpret_cast<ChanConst *>(smem);                          // [KT]
    float *s_part = reinterpret_cast<float *>(smem + KT * sizeof(ChanConst));          // [KT][NW][64]
    float *s_ucar = s_part + KT * NW * 64;                                             // [KT][kUcarSteps][G * S][re, im]
    float *s_rep = s_ucar + KT * kUcarFloats;                                          // [KT][RCH]
    const int SEG = a.seg_steps;        // steps whose replica is produced at once (<= dc_segment_steps(CHUNK, KT, MT))
    // floats per channel: the replica (+ its shifted copy); one-wave workgroups: sized by the host for this launch
    const int RCH = a.rep_chan_floats;
    int8_t *s_code = reinterpret_cast<int8_t *>(s_rep + KT * RCH);                     // [KT][code_row_stride]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uni(tid >> 6);
    const int at_w = wave % AW;  // antenna tile of this wave inside the workgroup's antenna group
    const int sub = wave / AW;   // sample sub-chunk of this wave
    const int vt = sub * 64 + lane;
    const int rel0 = vt * S;

    // Workgroup -> (tile, channel group).  A tile = (block group, antenna group, split): the bytes the KG channel
    // groups share.  Blocks id and id+8 land on the same XCD (round-robin dispatch), so the KG workgroups of one
    // tile get ids tile%8 + 8*(kg + KG*(tile/8)): same XCD, dispatched back to back -> the tile comes from HBM
    // once and from that XCD's L2 for the other KG-1 channel groups.  (Speed only: nothing depends on placement.)
    const unsigned xcd = blockIdx.x & 7u, jq = blockIdx.x >> 3;
    const int kg = (int)(jq % (unsigned)a.KG);
    unsigned tile = (jq / (unsigned)a.KG) * 8u + xcd;
    if (tile >= (unsigned)a.num_tiles) return; // padding of the last group of 8 (whole workgroup exits)
    GAT_DC_LAT_CUT_AT(1);
    const int split = tile % a.splits;
    tile /= a.splits;
    const int ag = tile % a.ant_groups;
    const int bg = tile / a.ant_groups;

    const int Lc = a.Lc;
    const int N = (int)a.N;
    const float inv_lc = 1.0f / (float)Lc;
    const int shift0 = a.shifts[0];

    // replica producer role of this thread: channel slot gk (wave-uniform), entries gr, gr + RPC, gr + 2 RPC, ... of
    // every segment (consecutive lanes store consecutive floats); entry i <-> sample (segment start) + shift0 + i
    const int gk = KT == 1 ? 0 : uni(tid / RPC);
    const int gr_t = tid % RPC;

    const int c_begin = split * a.chunks_per_split;
    const int c_end = min(c_begin + a.chunks_per_split, a.total_chunks);
    // VEC == 4 (16-byte aligned block starts): EVERY chunk takes the vector path -- the lanes of the last chunk that lie
    // beyond the block's last whole group read zeros (buffer range check, no memory traffic); the N % S samples behind
    // that group are added by dc_tail_kernel, a second launch that exists for such block lengths only.
    // VEC == 1 (unaligned input): everything takes the per-sample path.
    const int c_full = VEC == 4 ? c_end : c_begin;
    // D register sets of samples: set d holds the steps d, d + D, ... of a block; while a step is consumed the following
    // D - 1 are in flight, and a set is refilled with step + D as soon as it is done (D = 2: the host picks it for the
    // streaming regime of the configs[1] family only, dc_depth_max).  The step loop runs whole groups of D steps in a
    // fixed order -- the compiler's count of outstanding loads at every wait is then exact (with a conditional last step it
    // assumes the worst order and waits for the newest loads too, which takes the depth away again); a step past the end
    // of the block reads zeros (range check) against a replica that exists there as well (the host uses D = 2 only when a
    // workgroup owns whole blocks: no other workgroup reads those samples).
    static_assert(D == 1 || (D == 2 && VEC == 4), "prefetch depth");
    const int c_stop = VEC == 4 ? c_begin + (c_end - c_begin + D - 1) / D * D : c_begin;
    int staged_prn[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) staged_prn[kk] = -1;
    i32x4 raw[D][G][MT][IO::NV]; // the samples of the step being consumed / in flight for the following ones
    bool preloaded = false;   // the previous block's last step has already fetched this block's first chunk

    for (int bb = 0; bb < a.blocks_per_wg; ++bb) {
        const int b = bg * a.blocks_per_wg + bb;
        if (b >= a.B) break;
        const bool next_block = bb + 1 < a.blocks_per_wg && b + 1 < a.B && c_begin < c_full; // splits == 1 here

        // Sample loads are raw buffer loads through ONE descriptor per plane that spans the wave's MT antennas of this
        // block (built once per step from wave-uniform values); the antenna is the instruction's scalar offset, the lane
        // offset is shared by every load of a step.  Lanes beyond the block's end get the offset 2^31: beyond every
        // record (the host keeps a tile's span below 2^31), they read zeros without touching memory -- the ragged
        // last chunk needs no special path (measured on gfx950: the range check is per dword and covers voffset +
        // soffset).  (A descriptor per (antenna, plane) of one block's length needs no lane mask but ~6 scalar
        // instructions per load: a fifth of all instructions of the four-antenna five-tap step.)  The cache policy is a template parameter: a
        // wave-uniform `if (keep) plain else non-temporal` pair of ordinary loads is merged by the compiler into plain
        // loads (the hint is only metadata; that cost 7 % at configs[1]), and the same branch around buffer loads
        // breaks the step into many basic blocks (+ 50 registers).
        const size_t base = (size_t)b * a.block_stride + (size_t)kg * a.chan_stride /* != 0 only with KT == 1 */ +
                            (size_t)((ag * AW + at_w) * MT) * a.ant_stride;
        // Line alignment: a wave-instruction fetches 1 KiB of contiguous memory per plane; when the block does not start on a
        // 128-byte line (N = 50 000 floats: every other block starts 64 bytes into one) it straddles nine lines instead of
        // eight and the line it shares with its neighbour in time is fetched twice (configs[3] shard: traffic 1.05 x the
        // algorithmic bytes, 2 % of the time: profiles/r04/r04a_block_alignment_probe.txt).  The workgroup therefore walks
        // the block from a VIRTUAL start `head_b` bytes ahead of it, on the line's boundary: lane offsets, replica entries
        // and the carrier table are all in virtual samples, lanes in front of the real start read zeros (range check).
        // head_b is a multiple of 16 (whole groups); launch-uniform switch (the host turns it on where a stride or the base
        // is not a multiple of 128 bytes, and then gives every block one chunk of slack and one block per workgroup).
        unsigned head_b = 0;
        if constexpr (VEC == 4) {
            if (a.align_head) {
                const size_t base0 = (size_t)b * a.block_stride + (size_t)kg * a.chan_stride + (size_t)(ag * AW * MT) * a.ant_stride;
                head_b = uni((unsigned)(reinterpret_cast<size_t>(a.re) + base0 * EB) & 127u); // the tile's first antenna
            }
        }
        const int head_s = (int)(head_b / (unsigned)EB); // virtual sample v <-> sample v - head_s of the block
        const char *const p_re = static_cast<const char *>(a.re) + base * EB - head_b;
        const char *const p_im = static_cast<const char *>(FMT == GAT_LAYOUT_PLANAR ? a.im : a.re) + base * EB - head_b;
        const size_t ant_bytes = (size_t)a.ant_stride * EB;
        const size_t blk_bytes = (size_t)a.block_stride * EB;
        // whole 16-byte groups only: of a block length that is no multiple of the group size S the vector path covers the
        // first n_vec = N - N % S samples (lanes beyond read zeros through the range check); the N % S < 8 samples behind
        // them are added by dc_tail_kernel (gat_kernels.hip), launched behind this kernel for such lengths only.  Inside
        // this kernel a per-sample path costs every instance registers whether it runs or not -- measured twice: behind
        // the step loop its scalar state stays live across the loop (+ 4 registers everywhere, the one-wave three-tap
        // instance 6 -> 5 waves per SIMD), ahead of the loop the accumulators become live on two paths (+ 6-18).
        const int blk_len = (VEC == 4 ? a.n_vec : N) * EB; // bytes of one antenna's block that this kernel covers
        const unsigned v_end = (unsigned)blk_len + head_b; // virtual byte offset of the block's end
        const unsigned tile_len = (unsigned)((MT - 1) * ant_bytes) + v_end; // the descriptors' num_records (host: < 2^31)
        constexpr unsigned kNoRecord = 0x80000000u;
        auto plane_rsrc = [&](const char *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p), 0, (int)tile_len, 0x00020000); };
        // (one unsigned compare covers both ends: offsets in front of the real start wrap to huge values)
        auto lane_offset = [&](unsigned off) { return off - head_b < (unsigned)blk_len ? off : kNoRecord; };

        // 16-byte loads of antenna m's group at byte offset `off` of the block that starts at (bre, bim) (KEEP: plain
        // loads that stay in L2 for the other channel groups, otherwise non-temporal: aux bit 1)
        auto load_ant = [&](i32x4 (&raw)[IO::NV], int m, __amdgpu_buffer_rsrc_t rr, __amdgpu_buffer_rsrc_t ri, unsigned off) {
#if defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 1)
            off &= 0x3ff0u; // every load hits the same 16 KB (cache-resident): the arithmetic without the HBM stream
#endif
            constexpr int aux = KEEP ? 0 : 2;
            const int soff = (int)((unsigned)m * (unsigned)ant_bytes); // wave-uniform
            raw[0] = (i32x4)__builtin_amdgcn_raw_buffer_load_b128(rr, off, soff, aux);
            if constexpr (IO::NV == 2) raw[1] = (i32x4)__builtin_amdgcn_raw_buffer_load_b128(ri, off, soff, aux);
        };
        // ---- the block's first samples: requested before anything else (they depend on the launch geometry only), so
        // that their trip from HBM overlaps the parameter fetch, the chip-table staging and the first replica segment --
        // in a single-block call that trip is a fifth of the kernel's time (scripts/history/r03/r03_latency_cuts.sh)
        if (c_begin < c_full && !preloaded) {
            const __amdgpu_buffer_rsrc_t rr = plane_rsrc(p_re), ri = plane_rsrc(p_im);
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const unsigned off = lane_offset((unsigned)((c_begin + d) * CHUNK + g * GSTRIDE + rel0) * EB);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        load_ant(raw[d][g][m], m, rr, ri, off);
                        // the order of the step loop's refills: the scheduler would group the preload by plane, and the
                        // compiler's wait counts at the loop head are exact only if both orders agree
                        if constexpr (D > 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }

        // ---- per-channel constants of this block --------------------------------------------------------------
        // The double-precision ones (code rate, code phase, carrier step, carrier phase) are needed at segment starts
        // and on ragged ends only: they live in LDS.  In registers (wave-uniform -> scalar): the one-sample and
        // one-step carrier rotations.  (The previous block's last reads of s_const precede its reduction barrier.)
        // Carrier = U x Q: U[step][sample of a lane's groups] = exp(j 2 pi f/fs (step * CHUNK + g * GSTRIDE + j)) is the same
        // for every lane and wave -- one table per segment in LDS (below) --, Q = exp(j 2 pi (f/fs * rel0 + phi)) is this
        // lane's constant for the whole block: the step loop wipes off with U (no per-lane phasor arithmetic at all) and
        // the block's accumulators are rotated by conj(Q) once, ahead of the reduction.
        unsigned valid_mask = 0, bad_mask = 0;
        bool restage = false;
        int prn_k[KT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const int k = kg * KT + kk;
            const bool valid = k < a.K;
            const size_t pi_ = (size_t)b * a.K + (valid ? k : a.K - 1); // wave-uniform
            gat_channel_params P;
            if (a.params) {
                P = a.params[pi_];
            } else { // records inside the kernel arguments: scalar loads from the constant address space (written as
                     // `a.inl[pi_]` the compiler merges both sources into one FLAT vector load)
                typedef const unsigned long long __attribute__((address_space(4))) *kwords;
                kwords w = (kwords)__builtin_amdgcn_kernarg_segment_ptr() + (offsetof(DcArgs, inl) / 8 + pi_ * 5);
                static_assert(sizeof(gat_channel_params) == 40 && offsetof(DcArgs, inl) % 8 == 0, "record = five 8-byte words");
                P.prn = (int)(unsigned)w[0];
                P.reserved = 0;
                P.code_freq_hz = __longlong_as_double((long long)w[1]);
                P.carrier_freq_hz = __longlong_as_double((long long)w[2]);
                P.code_phase_chips = __longlong_as_double((long long)w[3]);
                P.carrier_phase_cycles = __longlong_as_double((long long)w[4]);
            }
            double ratio = P.code_freq_hz / a.fs;         // src/algorithms.jl:179 (Float64 division)
            double tau = P.code_phase_chips;
            double step = P.carrier_freq_hz / a.fs;       // cycles per sample
            double phi = P.carrier_phase_cycles;
            // Parameters this kernel cannot evaluate exactly poison the output with NaN (fail loudly):
            // prn outside the table, or a code-phase span beyond the int32 / float-reciprocal modulo range
            // (the host entry point rejects these up front; device-resident parameters are checked here).
            const bool bad = P.prn < 0 || P.prn >= a.num_prns || code_span_bad(ratio, tau, (double)(N + a.max_abs_shift), Lc) ||
                             !(step == step) || !(phi == phi) ||
                             !(__builtin_fabs(step) < 1.0e15) || !(__builtin_fabs(phi) < 1.0e15);
            if (bad) ratio = 0.0, tau = 0.0, step = 0.0, phi = 0.0; // tame values; the output is poisoned below
            if (valid) valid_mask |= 1u << kk;
            if (valid && bad) bad_mask |= 1u << kk;
            prn_k[kk] = (P.prn < 0 || P.prn >= a.num_prns) ? 0 : P.prn;
            restage |= valid && prn_k[kk] != staged_prn[kk];
            if (tid == 0) s_const[kk] = ChanConst{ratio, tau, step, phi};
        }
        valid_mask = uni(valid_mask);
        bad_mask = uni(bad_mask);

        if (uni((int)restage)) { // (re)stage the chip tables: rows are padded to 16 bytes on the device -> 16-byte copies
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                const i32x4 *g = reinterpret_cast<const i32x4 *>(a.codes + (size_t)prn_k[kk] * a.code_row_stride);
                i32x4 *d = reinterpret_cast<i32x4 *>(s_code + (size_t)kk * a.code_row_stride);
                for (int i = tid; i < (Lc + 15) / 16; i += T) d[i] = g[i];
                staged_prn[kk] = prn_k[kk];
            }
        }
        __syncthreads(); // s_const and the tables are in place
        GAT_DC_LAT_CUT_AT(2);

        // ---- replica producer: walk constants of this thread's channel ----------------------------------------
        // one producer step advances RPC samples (the thread's next entry)
        // (four wave-uniform words live across the step loop; everything else the producers need is derived from them at
        // each segment start, behind an opaque copy -- the compiler otherwise hoists the derived constants out of the
        // segment loop and holds them, spilled into vector-register lanes, across the step loop)
        unsigned w_r1lo, w_r1hi, w_margin, w_flags; // 32.32 chips per sample; margin; bit 0: exact only
        const bool g_valid = (valid_mask >> gk) & 1u;
        {
            const ChanConst cc = s_const[gk];
            const double span = __builtin_fabs(cc.tau) + cc.ratio * (double)(N + a.max_abs_shift) + 1.0;
            // a producer walks RPC samples per step, at most one segment (+ overshoot) away from its anchor
            const ChipWalkConst wc = chip_walk_setup(cc.ratio, span, SEG * CHUNK + a.rep_span + 5 * RPC, RPC, Lc);
            const unsigned long long r1 = uni(wc.rate);
            w_r1lo = (unsigned)r1;
            w_r1hi = (unsigned)(r1 >> 32);
            w_margin = uni(wc.margin);
            w_flags = uni(wc.exact_only) != 0 ? 1u : 0u;
        }

        f32x2 acc[KT][MT][L]; // (re, im)
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int l = 0; l < L; ++l) acc[kk][m][l] = f32x2{0.f, 0.f};

        // chips of the sample at segment-relative position rel, for the L taps (scalar path)
        auto get_chips = [&](float (&chip)[L], int rel, const float *rep) {
#pragma unroll
            for (int l = 0; l < L; ++l) chip[l] = rep[rel + (a.shifts[l] - shift0)];
        };
        // chips of the S samples of one group (first sample at segment-relative position rel, a multiple of S) for the
        // L taps: one 8-byte-aligned vector read per tap and 4 samples -- tap_off[l] is the tap's distance from the first
        // when that is even, else (distance - 1) into the copy stored one entry further (host: gat_api.cpp).
        // SB = samples handled at once: the whole group, or half of the eight-sample groups of int8 pairs (see the step)
        constexpr int SB = dc_sub_batch(S, MT, L, KT);
        constexpr int NH = S / SB;
        auto get_chips_sub = [&](float (&chip)[SB][L], int rel, const float *rep) {
#if defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 2)
            for (int j = 0; j < SB; ++j) for (int l = 0; l < L; ++l) chip[j][l] = __int_as_float(0x3f800000 + ((rel + j + l) & 1));
            return;
#endif
            typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
            typedef float f32x2a8 __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float *p = rep + rel + a.tap_off[l];
                if constexpr (SB == 4) {
                    const f32x4a8 v = *reinterpret_cast<const f32x4a8 *>(p);
#pragma unroll
                    for (int j = 0; j < 4; ++j) chip[j][l] = v[j];
                } else if constexpr (SB == 2) {
                    const f32x2a8 v = *reinterpret_cast<const f32x2a8 *>(p);
                    chip[0][l] = v[0];
                    chip[1][l] = v[1];
                } else {
#pragma unroll
                    for (int j = 0; j < SB; ++j) chip[j][l] = rep[rel + j + (a.shifts[l] - shift0)];
                }
            }
        };
        auto get_chips_group = [&](float (&chip)[S][L], int rel, const float *rep) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float c[SB][L];
                get_chips_sub(c, rel + h * SB, rep);
#pragma unroll
                for (int j = 0; j < SB; ++j)
#pragma unroll
                    for (int l = 0; l < L; ++l) chip[h * SB + j][l] = c[j][l];
            }
        };
        // one sample of one antenna: conj(carrier) wipe-off (src/algorithms.jl:175-176), L taps.  Plain scalar FMAs:
        // v_pk_fma_f32 issues at the rate of two v_fma_f32 on gfx950 but needs its operands in aligned register
        // pairs -- the packed form cost ~20 % extra v_mov in this loop (the dc translation units are built with
        // -fno-slp-vectorize so that the compiler does not re-pack them).
        auto accumulate = [&](f32x2 (&ac)[L], float xr, float xi, float cr, float ci, const float (&chip)[L]) {
            const float dr = __builtin_fmaf(xr, cr, xi * ci), di = __builtin_fmaf(xi, cr, -(xr * ci));
#pragma unroll
            for (int l = 0; l < L; ++l) {
                ac[l][0] = __builtin_fmaf(chip[l], dr, ac[l][0]);
                ac[l][1] = __builtin_fmaf(chip[l], di, ac[l][1]);
            }
        };
        // the S samples of one antenna's group (one channel per workgroup): all wipe-off products first, then the tap
        // multiply-adds -- every instruction then has its operands ready when it issues (written sample by sample the
        // compiler forms the products right in front of their first use; configs[2] 1.36 -> 1.30 ms).  Not for the
        // channel-looping instances: six more live registers there (configs[3] shard 0.70 -> 0.725 ms).
        // (`between` runs after the wipe-off products are formed and ahead of the tap multiply-adds: the last antenna of a
        // pass re-reads the phasor registers for the next pass there)
        auto accumulate_sub = [&](f32x2 (&ac)[L], const i32x4 (&rw)[IO::NV], int j0, const float (&pr)[SB], const float (&pi)[SB],
                                  const float (&chip)[SB][L], auto between) {
            float xr[SB], xi[SB], tr[SB], ti[SB], dr[SB], di[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                IO::get(rw, j0 + j, xr[j], xi[j]);
                tr[j] = xi[j] * pi[j];
                ti[j] = -(xr[j] * pi[j]);
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                dr[j] = __builtin_fmaf(xr[j], pr[j], tr[j]);
                di[j] = __builtin_fmaf(xi[j], pr[j], ti[j]);
            }
            between();
#pragma unroll
            for (int j = 0; j < SB; ++j)
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    ac[l][0] = __builtin_fmaf(chip[j][l], dr[j], ac[l][0]);
                    ac[l][1] = __builtin_fmaf(chip[j][l], di[j], ac[l][1]);
                }
        };
        // SB phasors of the step at segment position st: entries [e0, e0 + SB) of the step's row of the table U (every lane
        // reads the same address: broadcast reads, 16 bytes = two phasors each)
        auto table_phasors = [&](auto cnt_c, float *pr, float *pi, int kk, int st, int e0) {
            constexpr int CNT = decltype(cnt_c)::value;
            const float *u = s_ucar + (kk * kUcarSteps + st) * (2 * G * S) + 2 * e0;
            if constexpr (CNT % 2 == 0) {
#pragma unroll
                for (int j = 0; j < CNT; j += 2) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(u + 2 * j);
                    pr[j] = v[0]; pi[j] = v[1]; pr[j + 1] = v[2]; pi[j + 1] = v[3];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CNT; ++j) pr[j] = u[2 * j], pi[j] = u[2 * j + 1];
            }
        };
        // samples [n_lo, n_hi) one at a time with scalar loads (ragged block end, unaligned input)
        auto scalar_run = [&](int kk, int n_lo, int n_hi, int rel, const float *rep) {
            for (int n = n_lo; n < n_hi; ++n, ++rel) {
                float cr, ci, chip[L];
                const double th = __builtin_fma((double)n, s_const[kk].step, s_const[kk].phi);
                sincos_cycles(th - __builtin_rint(th), cr, ci);
                get_chips(chip, rel, rep);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float xr, xi;
                    IO::load1(a.re, a.im, base + (size_t)m * a.ant_stride + n, xr, xi);
                    accumulate(acc[kk][m], xr, xi, cr, ci, chip);
                }
            }
        };
        // The replica of one SEGMENT (SEG steps): entry i of a channel <-> sample c0*CHUNK + shift0 + i
        // (src/algorithms.jl:753-757), i < seg_cnt = steps*CHUNK + tap span.  Producer thread gr of a channel owns
        // entries gr, gr + RPC, ...: ONE exact double-precision anchor, then the 32.32 walk (gat_phase.h) RPC samples
        // at a time in batches of 4, branch-free; a batch with an unproven entry is redone with the reference's
        // expression.  Every thread takes the same number of steps (the overshoot lands in the copy's spare room).
        // (TWO: taps at odd distances -> the copy shifted by one entry is stored as well; the store addresses of a batch are
        // one base register + immediate offsets, and only the last, partial batch of a run checks its bound)
        auto fill_impl = [&](auto two_c, int c0, int seg_cnt) {
            constexpr bool TWO = decltype(two_c)::value;
            // (opaque copy of the thread's producer index: addresses and sample indices derived from it are loop-invariant,
            // and hoisted out of the segment loop they sat in registers across the step loop -- three in every instance,
            // a wave per SIMD in 24 of the 141 planar instances)
            int gr = gr_t;
            asm volatile("" : "+v"(gr));
            float *rep = s_rep + gk * RCH + gr;
            float *rep1 = rep + a.rep_copy_stride - 1;            // copy[i] = entry i + 1
            const int8_t *tab = s_code + (size_t)gk * a.code_row_stride;
            const int run = (seg_cnt + RPC - 1) / RPC;            // entries per producer thread (wave-uniform)
            unsigned r1lo = w_r1lo, r1hi = w_r1hi, flags = w_flags;
            asm volatile("" : "+v"(r1lo), "+v"(r1hi), "+v"(flags));
            r1lo = uni(r1lo), r1hi = uni(r1hi), flags = uni(flags);
            const unsigned long long rp = ((unsigned long long)r1hi << 32 | r1lo) * (unsigned long long)RPC; // one producer step
            const unsigned w_rate_lo = (unsigned)rp, w_rate_hi = (unsigned)(rp >> 32);
            const bool w_exact = (flags & 1u) != 0;
            const int x0 = c0 * CHUNK - head_s + shift0 + gr;
            const double ratio = s_const[gk].ratio, tau = s_const[gk].tau;
            // exact anchor (src/algorithms.jl:179-182)
            const double p0 = code_phase(ratio, tau, x0);
            const double fl0 = __builtin_floor(p0);
            unsigned frac = (unsigned)((p0 - fl0) * 4294967296.0); // p - floor(p) is exact; truncation
            unsigned idx = (unsigned)floormod_fast((int)fl0, Lc, inv_lc);
            {
                const float v = (float)tab[idx];
                rep[0] = v;
                if (TWO && gr > 0) rep1[0] = v;
            }
            // one batch of NB walked entries j0 .. j0 + NB - 1 (stored at w[0], w[RPC], ...)
            auto batch = [&](auto partial_c, auto nb_c, int j0) {
                constexpr bool PARTIAL = decltype(partial_c)::value;
                constexpr int NB = decltype(nb_c)::value;
                float *const w = rep + j0 * RPC;
                unsigned id[NB];
                bool amb = w_exact;
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const unsigned f2 = frac + w_rate_lo;
                    idx += w_rate_hi + (f2 < frac ? 1u : 0u); // carry of the fraction = one more chip
                    frac = f2;
                    idx = min(idx, idx - (unsigned)Lc);       // fewer than Lc chips per step: one wrap at most
                    // proven <=> margin <= frac <= 2^32 - 1 - margin <=> (frac - margin) + 2 margin does not carry
                    amb |= (frac - w_margin) > (0xffffffffu - 2u * w_margin);
                    id[u] = idx;
                }
                if (__builtin_expect(amb, 0)) { // some entry of the batch is not proven (or nothing is): evaluate exactly
#pragma unroll 1
                    for (int u = 0; u < NB; ++u) id[u] = (unsigned)chip_index(ratio, tau, x0 + RPC * (j0 + u), Lc, inv_lc);
                }
                // all table reads first: the table is int8 (a character type may alias anything), so a read written
                // after a replica store would have to wait for it -- one serial LDS round trip per entry
                int8_t t[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) t[u] = tab[id[u]];
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (!PARTIAL || j0 + u < run) { // wave-uniform bound
                        const float v = (float)t[u];
                        w[u * RPC] = v;
                        if constexpr (TWO) w[a.rep_copy_stride - 1 + u * RPC] = v;
                    }
            };
            int j0 = 1;
            // (eight entries per LDS round trip, NB = 8, measured: configs[2] 1.186 -> 1.195 ms, the rest unchanged --
            // profiles/r03/r03x_ab_fill_batch8_not_kept.txt)
            for (; j0 + 4 <= run; j0 += 4) batch(std::false_type{}, std::integral_constant<int, 4>{}, j0);
            if (j0 < run) batch(std::true_type{}, std::integral_constant<int, 4>{}, j0);
        };
        auto fill_segment = [&](int c0, int seg_cnt) {
#if defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 4)
            return;
#endif
            if (!g_valid) return;
            if (a.rep_copy_stride != 0) fill_impl(std::true_type{}, c0, seg_cnt); // wave-uniform
            else fill_impl(std::false_type{}, c0, seg_cnt);
        };

        preloaded = next_block;
        const int c_last = VEC == 4 ? c_stop : c_end;
        for (int c0 = c_begin; c0 < c_last; c0 += SEG) {
            const int c1 = min(c0 + SEG, c_last);
#if !(defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 8))
            if (c0 > c_begin) __syncthreads(); // everybody has finished reading the previous segment's replica
#endif
            fill_segment(c0, (c1 - c0) * CHUNK + a.rep_span);
            // the segment's carrier table U: one exact evaluation (src/algorithms.jl:172: double-precision phase, reduced,
            // float sincos) per (channel, step, sample of a lane's groups) -- KT * steps * G * S <= 256 entries, one thread each
            static_assert(KT * kUcarSteps * G * S <= T, "one table entry per thread");
            {
                // (the thread id goes through an opaque copy: everything derived from it is otherwise hoisted out of the
                // segment loop and kept in registers across the step loop -- twelve of them, a wave per SIMD in some instances)
                int t_ = tid;
                asm volatile("" : "+v"(t_));
                const int e = t_ % (G * S), st = (t_ / (G * S)) % kUcarSteps, kk = t_ / ((G * S) * kUcarSteps);
                if (kk < KT && st < c1 - c0) {
                const int n = (c0 + st) * CHUNK - head_s + (e / S) * GSTRIDE + (e % S); // sample of lane 0 (rel0 = 0)
                const double th = (double)n * s_const[kk].step;
                float cr, ci;
                sincos_cycles(th - __builtin_rint(th), cr, ci);
                *reinterpret_cast<f32x2 *>(s_ucar + ((kk * kUcarSteps + st) * (G * S) + e) * 2) = f32x2{cr, ci};
                }
            }
#if !(defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 8))
            __syncthreads();
#endif
            GAT_DC_LAT_CUT_AT(3);

            // ---- whole chunks.  The samples of step c+1 are loaded while step c is consumed: antenna by antenna, into
            // the registers that antenna's samples of step c have just left (loads in flight all the time, no second
            // register set).  The prefetch is unconditional -- a conditional one makes the compiler copy the whole
            // register array around the branch; after the last whole chunk every lane re-loads the block's first bytes.
            const int cf = VEC == 4 ? c1 : c_begin; // c1 <= c_stop: a multiple of D steps from c0 (SEG is one: host)
            // one channel per workgroup: the phasors of a pass live in these registers from the end of the previous pass
            float pr[SB], pi[SB];
            if constexpr (KT == 1 && VEC == 4 && MT >= 2) table_phasors(std::integral_constant<int, SB>{}, pr, pi, 0, 0, 0);
            // one step: the samples of chunk c sit in register set DI, which is refilled with chunk c + D
            auto step = [&](auto di, int c) {
                constexpr int DI = decltype(di)::value;
                const int srel = (c - c0) * CHUNK; // position of the step inside the segment
                // refill: chunk c + D of this block; past its end chunk DI of the next block this workgroup walks (the last
                // D steps of a block cover every set once); else an offset beyond the block: zeros, no memory traffic
                const bool more = c + D < c_stop;
                const bool hop = !more && next_block; // wave-uniform
                const char *const n_re = hop ? p_re + blk_bytes : p_re, *const n_im = hop ? p_im + blk_bytes : p_im;
                const unsigned next_off = more ? (unsigned)((c + D) * CHUNK + rel0) * EB
                                          : (hop ? (unsigned)((c_begin + DI) * CHUNK + rel0) * EB : v_end);
                const unsigned next_g = more || hop ? (unsigned)(GSTRIDE * EB) : 0u;
                const __amdgpu_buffer_rsrc_t n_rr = plane_rsrc(n_re), n_ri = plane_rsrc(n_im);
                unsigned n_off[G];
#pragma unroll
                for (int g = 0; g < G; ++g) n_off[g] = lane_offset(next_off + g * next_g);
                if constexpr (KT == 1) {
                    // the samples of the next step are fetched antenna by antenna, into the registers that antenna's samples
                    // of this step have just left
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int rel = srel + rel0 + g * GSTRIDE;
                        // SB samples at a time (NH = 2 passes over the eight-sample groups of int8 pairs: chips, phasors and
                        // wipe-off products of four samples live at once instead of eight -- 153 -> fewer registers, a
                        // fourth wave per SIMD); an antenna's registers are refilled when its last samples are consumed
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            float chip[SB][L];
                            // the chips' LDS reads go out first: left alone the scheduler issues them behind the first
                            // antenna's wipe-off (and its wait for the samples), right in front of their first use
                            // (configs[2] 1.178 -> 1.169 ms, the other shapes unchanged: profiles/r03/r03x_ab_chips_first.txt)
                            get_chips_sub(chip, rel + h * SB, s_rep);
                            __builtin_amdgcn_sched_barrier(0);
                            // the pass's phasors were read at the end of the previous pass; the NEXT pass's are read while
                            // the last antenna's tap multiply-adds run (same registers: no wait for LDS at a pass's start)
                            // (one-antenna tiles read them at the pass's start: there is no other antenna's arithmetic to
                            // read behind, and the early read costs the one-wave instances four registers = a wave per SIMD)
                            constexpr bool PF = MT >= 2;
                            if constexpr (!PF) table_phasors(std::integral_constant<int, SB>{}, pr, pi, 0, c - c0, g * S + h * SB);
                            const bool last_pass = h + 1 == NH && g + 1 == G;
                            const int st_n = (c - c0) + (last_pass ? 1 : 0);
                            const int e_n = last_pass ? 0 : (h + 1 == NH ? (g + 1) * S : g * S + (h + 1) * SB);
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                if (PF && m + 1 == MT)
                                    accumulate_sub(acc[0][m], raw[DI][g][m], h * SB, pr, pi, chip,
                                                   [&]() { table_phasors(std::integral_constant<int, SB>{}, pr, pi, 0, st_n, e_n); });
                                else
                                    accumulate_sub(acc[0][m], raw[DI][g][m], h * SB, pr, pi, chip, []() {});
                                if (h + 1 == NH) {
#if !(defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 16))
                                    load_ant(raw[DI][g][m], m, n_rr, n_ri, n_off[g]);
#endif
                                }
                                // antenna by antenna: left alone the scheduler wipes off all antennas first (their
                                // products and the refilled sample registers are then live together: + 30 registers)
#if !(defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 32))
                                __builtin_amdgcn_sched_barrier(0);
#endif
                            }
                        }
                    }
                } else {
                    // Several channels on register-resident samples: the chips and phasors of ALL channels of the step are
                    // fetched first, then ANTENNA by antenna -- all channels of antenna m, then antenna m's registers are
                    // refilled for the next step, so that its load has a whole step of arithmetic to land.  (Channel by
                    // channel the refills could only start in the last channel's pass: the 16-antenna shard of
                    // configs[3] spent 3/4 of every step with no load in flight.)  A slot without a channel (K not a
                    // multiple of KT, last channel group only) runs on the last channel's parameters and whatever its
                    // replica slot holds; nothing of it is written (a branch per channel and antenna cost more).
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int rel = srel + rel0 + g * GSTRIDE;
                        float pr[KT][S], pi[KT][S], chip[KT][S][L];
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk) {
                            table_phasors(std::integral_constant<int, S>{}, pr[kk], pi[kk], kk, c - c0, g * S);
                            get_chips_group(chip[kk], rel, s_rep + kk * RCH);
                        }
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
#pragma unroll
                            for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
                                for (int j = 0; j < S; ++j) {
                                    float xr, xi;
                                    IO::get(raw[DI][g][m], j, xr, xi);
                                    accumulate(acc[kk][m], xr, xi, pr[kk][j], pi[kk][j], chip[kk][j]);
                                }
                            }
#if !(defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 16))
                            load_ant(raw[DI][g][m], m, n_rr, n_ri, n_off[g]);
#endif
#if !(defined(GAT_DC_ABLATE) && (GAT_DC_ABLATE & 32))
                            __builtin_amdgcn_sched_barrier(0);
#endif
                        }
                    }
                }
            };
            for (int c = c0; c < cf; c += D) {
                step(std::integral_constant<int, 0>{}, c);
                if constexpr (D > 1) step(std::integral_constant<int, 1>{}, c + 1);
            }
            // ---- unaligned input (VEC == 1): one sample at a time with scalar loads -----------------------------------
            if constexpr (VEC != 4) {
#pragma unroll 1
                for (int c = max(c0, cf); c < c1; ++c) {
                    const int n = c * CHUNK + rel0; // S == 1, G == 1
                    if (n < N) {
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk) {
                            if (KT > 1 && !((valid_mask >> kk) & 1u)) continue;
                            scalar_run(kk, n, n + 1, (c - c0) * CHUNK + rel0, s_rep + kk * RCH);
                        }
                    }
                }
            }
        }

        GAT_DC_LAT_CUT_AT(4);
        // ---- block reduction: per channel 2*MT*L values per wave -> butterfly -> waves sharing an antenna tile ----
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            float v[NV];
            float q_r = 1.f, q_i = 0.f;
            if constexpr (VEC == 4) { // Q of this lane (computed here, not at the block's start: nothing of it lives across the step loop)
                const double thq = __builtin_fma((double)rel0, s_const[kk].step, s_const[kk].phi); // src/algorithms.jl:172 at the lane's first sample
                sincos_cycles(thq - __builtin_rint(thq), q_r, q_i);
            }
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float ar = acc[kk][m][l][0], ai = acc[kk][m][l][1];
                    if constexpr (VEC == 4) { // this lane's share of the carrier: acc * conj(Q)
                        v[(l * MT + m) * 2 + 0] = __builtin_fmaf(ar, q_r, ai * q_i);
                        v[(l * MT + m) * 2 + 1] = __builtin_fmaf(ai, q_r, -(ar * q_i));
                    } else { // scalar-load path: every sample was wiped off with its own exact phasor
                        v[(l * MT + m) * 2 + 0] = ar;
                        v[(l * MT + m) * 2 + 1] = ai;
                    }
                }
            Butterfly<NV, 32>::run(v, lane);
            // lanes sharing an index hold bit-identical sums
            s_part[(kk * NW + wave) * 64 + Butterfly<NV, 32>::index(lane)] = v[0];
        }
        __syncthreads();
        GAT_DC_LAT_CUT_AT(5);

#if defined(GAT_DC_LAT_CUT) && GAT_DC_LAT_CUT == 6
        if (false)
#endif
        for (int o = tid; o < KT * AW * NV; o += T) {
            const int kk = o / (AW * NV);
            const int r = o - kk * (AW * NV);
            const int at = r / NV, vi = r - (r / NV) * NV;
            const int k = kg * KT + kk;
            if (k >= a.K) continue;
            const float *p = s_part + (kk * NW + at) * 64 + vi; // wave = sub * AW + at
            float tot;
            if constexpr (SUBS == 4) tot = (p[0] + p[64]) + (p[128] + p[192]);
            else if constexpr (SUBS == 2) tot = p[0] + p[AW * 64];
            else tot = p[0];
            if ((bad_mask >> kk) & 1u) tot = __builtin_nanf("");
            const int comp = vi & 1;
            const int ml = vi >> 1;
            const int m = (ag * AW + at) * MT + (ml % MT);
            // position of this tap in the caller's shift list (a select chain over scalar registers: indexed by a lane
            // value the argument array would be read with a VECTOR load -- through the vector memory path, a serial uncached trip at the very end of the kernel)
            int l = a.tap_index[0];
#pragma unroll
            for (int q = 1; q < L; ++q) l = ml / MT == q ? a.tap_index[q] : l;
            const size_t bk = (size_t)b * a.K + k;
            const size_t oidx = (bk * a.Ltot + l) * a.M + m;
            if (a.flags & GAT_FLAG_ATOMIC) {
                atomicAdd((comp ? a.out_im : a.out_re) + oidx, tot);
            } else if (a.splits == 1) {
                (comp ? a.out_im : a.out_re)[oidx] = tot;
            } else {
                const size_t elems = (size_t)a.Ltot * a.M * 2;
                a.partial[(bk * a.splits + split) * elems + ((size_t)l * a.M + m) * 2 + comp] = tot;
            }
        }
        // the next block's s_const / table writes come after this barrier-separated reduction: the threads that still
        // read s_part above do not touch s_const, s_rep or s_code
    }
    completion_flag(a.done_counter, a.host_flag, a.flag_seq, a.total_wgs);
}

// ------------------------------------------------------------------------------------------------------------
// instance table: which (MT, L, VEC, AW, KT) combinations exist, per sample format (one translation unit each)
// ------------------------------------------------------------------------------------------------------------
// Which instances exist.  Register accumulators 2 * MT * L * KT <= 64; antenna-parallel waves (AW > 1) need full
// 4-antenna tiles; unaligned input (VEC == 1, scalar loads) is served one antenna per workgroup.
constexpr bool dc_instance(int mt, int l, int vec, int aw, int kt, int nw = 4, int depth = 1)
{
    if (depth != 1 && !(vec == 4 && depth == dc_depth_max(mt, l, aw, kt, nw))) return false;
    // one-wave workgroups: short blocks of one- and two-antenna tiles
    if (nw != 4 && !(nw == 1 && vec == 4 && aw == 1 && kt == 1 && mt <= 2)) return false;
#ifdef GAT_DC_DEV // development builds: only the instances the BASELINE shapes use (compiles in seconds)
    if (!((mt == 1 || mt == 4) && (l == 3 || l == 5) && vec == 4)) return false;
#endif
    if (mt < 1 || mt > kMaxAntTile || l < 1 || l > kMaxTapsPerLaunch) return false;
    if (vec != 4) return vec == 1 && mt == 1 && aw == 1 && kt == 1;
    if (aw != 1 && (mt != 4 || aw != 4)) return false;
    // several channels per workgroup only with antenna-parallel waves: measured on MI355X, a channel loop over one
    // antenna tile never beat separate channel workgroups sharing the tile through L2 (M = 1, 4; K = 8, 12)
    if (kt != 1 && aw != 4) return false;
    return (kt == 1 || kt == 2 || kt == 4) && mt * l * kt <= 48;
}

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
