// gat_internal.h -- shared between the C-ABI layer (gat_api.cpp) and the kernels (gat_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gat.h"

// Diagnostic code (ablations, latency cuts, cycle stamps: kernels that compute WRONG results on purpose) and the
// environment knobs of gat_create exist in development builds only; gat_version() names every flag of a build.
#if (defined(GAT_DC_ABLATE) || defined(GAT_DC_LAT_CUT) || defined(GAT_ABLATE) || defined(GAT_MFMA_STAMPS) || defined(GAT_MB_CW8) || defined(GAT_MB_NT_LOADS) || defined(GAT_MB_NO_RING) || defined(GAT_MB_X2_ITEMS4) || defined(GAT_MB_X2_DEPTH2) || \
     defined(GAT_RES_STAMPS) || defined(GAT_RES_FENCE)) &&                                                                                                 \
    !defined(GAT_DEV)
#error "diagnostic builds (-DGAT_DC_ABLATE, -DGAT_DC_LAT_CUT, -DGAT_ABLATE, -DGAT_MFMA_STAMPS, -DGAT_MB_CW8, -DGAT_RES_STAMPS, -DGAT_RES_FENCE) need -DGAT_DEV"
#endif

namespace gat {

constexpr int kThreads = 256;       // 4 wave64 per workgroup
constexpr int kMaxTapsPerLaunch = 8; // taps handled by one launch (register accumulators)
constexpr int kInlineParams = 4;     // channel parameter records of a host call that travel inside the kernel arguments
constexpr int kMaxAntTile = 4;       // antennas handled by one workgroup
constexpr int kFinalizeFewSplits = 32; // second stage: up to this many splits are summed by one thread per output element
constexpr int kMaxReplicaSpan = 512;  // tap span the LDS replica segment of a launch is sized for by default
constexpr int kMaxLaunchSpan = 2048;  // largest tap span one launch serves (the replica's LDS grows with the span beyond
                                      // kMaxReplicaSpan; wider tap lists are cut into several launches)

// Sample ownership of one lane per step in dc_kernel: G groups of S consecutive samples, one
// 16-byte load per plane and group (vec == 4) or scalar loads (vec == 1).  S by format:
// planar f32 4, interleaved ComplexF32 2, interleaved int16 4, interleaved int8 8.
#ifndef GAT_PLANAR_GROUPS
#define GAT_PLANAR_GROUPS 1
#endif
constexpr int dc_group_samples(int vec, int fmt)
{
    return vec != 4 ? 1 : fmt == GAT_LAYOUT_PLANAR ? 4 : fmt == GAT_LAYOUT_INTERLEAVED ? 2 : fmt == GAT_LAYOUT_INTERLEAVED_I16 ? 4 : 8;
}
constexpr int dc_groups(int vec, int fmt)
{
    return vec != 4 ? 1 : fmt == GAT_LAYOUT_PLANAR ? GAT_PLANAR_GROUPS : fmt == GAT_LAYOUT_INTERLEAVED ? 2 : 1;
}
// samples one workgroup covers per step: the nw / aw waves that share an antenna tile, 64 lanes each (nw = 4 waves per
// workgroup, or 1: one-wave workgroups for short blocks, see gat_dc.h)
constexpr int dc_chunk(int vec, int fmt, int aw = 1, int nw = 4) { return (64 * nw / aw) * dc_group_samples(vec, fmt) * dc_groups(vec, fmt); }

// Arguments of the fused correlator kernel (passed by value in the kernarg segment).
struct DcArgs {
    const void *re;                   // planar: float re plane; interleaved formats: base pointer
    const void *im;                   // planar: float im plane; otherwise unused
    const gat_channel_params *params; // dev, [B*K], channel fastest; null: the records are in `inl` (B*K <= kInlineParams)
    const int8_t *codes;              // dev, [P][code_row_stride] (rows padded to 16 bytes)
    const uint32_t *code_bits;        // dev, [P][table_stride / 4]: sign bits of +-1 chips (bit i & 31 of dword i >> 5), or null:
                                      // the workgroups stage the int8 rows.  Long codes (GPS L5: 10 KB per PRN as int8) are staged as
                                      // bits -- 1.3 KB -- so that the LDS holds long replica segments and two channels' tables
    float *out_re;                    // dev, [B][K][Ltot][M]
    float *out_im;
    float *partial;                   // dev, [B*K][splits][Ltot*M*2] (splits > 1 only)
    unsigned *done_counter;           // dev, arrival counter of the launch's workgroups, or null: no completion flag
    unsigned *host_flag;              // pinned host memory (device address): the last workgroup stores flag_seq there
    unsigned flag_seq, total_wgs;     // total_wgs: workgroups that do work (num_tiles * KG)
    long long N, ant_stride, block_stride, chan_stride;
    double fs;
    int M, K, B, Lc, num_prns, code_row_stride;
    int table_stride;      // bytes of ONE channel's chip table in LDS and between rows of what is staged: code_row_stride (int8
                           // chips) or 4 * dwords per sign-bit row (code_bits != null); a multiple of 16
    int KG;                // channel groups: ceil(K / KT)
    int splits, chunks_per_split, total_chunks;
    int ant_groups;        // M / (MT * AW): antenna groups, one workgroup each
    int blocks_per_wg;     // consecutive integration blocks one workgroup loops over (splits == 1 only)
    int num_tiles;         // ceil(B / blocks_per_wg) * ant_groups * splits
    int Ltot;              // taps of the whole call (output indexing)
    int max_abs_shift;     // max |shift| over ALL taps of the call (range check)
    int rep_span;          // shifts[last] - shifts[0] of THIS launch's taps (replica halo)
    int seg_steps;         // steps per segment (replica produced at once), <= dc_segment_steps()
    int rep_copy_stride;   // floats between the replica and its copy shifted by one entry (taps at odd offsets), 0: one copy
    int rep_chan_floats;   // floats of LDS per channel replica (one-wave workgroups: sized for this launch's taps; four-wave
                           // ones: for the segment the host chose, <= dc_rep_chan_floats())
    int keep_l2;           // 1: plain loads (several channel groups share the tile through L2), 0: non-temporal
    int n_vec;             // samples of a block the vector path covers: N - N % (samples per 16-byte load); N for scalar loads
    int align_head;        // 1: workgroups walk a block from the 128-byte line its first sample lies in (gat_dc.h)
    int fill_quads;        // 1: replica producers write four consecutive entries at a time where the code rate allows (gat_dc_body.inc)
    unsigned flags;
    int shifts[kMaxTapsPerLaunch];    // ascending
    int tap_off[kMaxTapsPerLaunch];   // float offset of tap l's chips from the lane's group base: even (8-byte aligned reads)
    int tap_index[kMaxTapsPerLaunch]; // position of each tap in the caller's list
    // host-parameter calls of up to kInlineParams records (the single-block call of a receiver loop, the reference's
    // kernel_algorithm with scalar arguments): no 40-byte upload in front of the launch (3.3 us of a 15 us call)
    gat_channel_params inl[kInlineParams];
};

// Arguments of dc_tail_kernel (gat_kernels.hip): the N % S samples at the end of every block that the vector path of
// dc_kernel leaves out when the block length is no multiple of the samples S one 16-byte load holds.
struct DcTailArgs {
    const void *re, *im;
    const gat_channel_params *params; // dev [B*K], or null: the records are in `inl`
    const int8_t *codes;
    float *out_re, *out_im;           // [B][K][L][M]: the tail's sums are ADDED to what dc_kernel (+ second stage) wrote
    unsigned *done_counter, *host_flag;
    unsigned flag_seq;
    long long N, ant_stride, block_stride, chan_stride;
    double fs;
    int M, K, B, L, Lc, num_prns, code_row_stride, format, n_vec, max_abs_shift;
    int shifts[GAT_MAX_TAPS];         // in the caller's order
    gat_channel_params inl[kInlineParams];
};
hipError_t launch_dc_tail(const DcTailArgs &a, hipStream_t s);

struct DcLaunch {
    int ant_tile; // MT: antennas per wave
    int aw;       // antenna tiles (waves) per workgroup: 1, 2, 4
    int kt;       // channels per workgroup: 1, 2, 4
    int nw;       // waves per workgroup: 4, or 1 (short blocks: one wave per block, no workgroup barrier to wait at)
    int depth;    // register sets of samples per wave (steps in flight): 1, or 2 (dc_depth_max)
    int taps;     // L of this launch
    int vec;      // 4 or 1
    int format;   // GAT_LAYOUT_*
    unsigned grid;
    unsigned lds_bytes;
};
// ---- resident correlator (gat_resident.h): calls without a launch ----------------------------------------------------
// One bounded-lifetime kernel stays on the device and serves single-block calls that the host rings in through a
// doorbell in pinned host memory; every workgroup posts its sums to pinned host memory as result lines.
// Doorbell: one 64-byte line per channel (K <= kResMaxChannels lines, written by the host in descending order, line 0 last):
//   dword 0 seq | 1 reserved | 2..11 gat_channel_params | 12..13 block offset in samples (int64) | 14 check | 15 seq
// check = XOR of dwords 0..13: a poll that catches a line half-written fails the check and is repeated.
// Result lines: workgroup `slot` (tile * KG + channel group, the body's decode of blockIdx) owns ceil(2 MT L / 14) lines of
//   14 values | check = XOR of the 14 values and seq | seq
// its value o is the sum for (tap l, antenna m of its tile, re / im) = (o / 2 / MT, o / 2 % MT, o % 2).  Written with plain
// stores and no fence: the host takes a call's results when every line carries the call's number and passes its check.
constexpr int kBellDwords = 16;
constexpr int kResMaxChannels = 16;    // channels (doorbell lines) of a resident correlator: four 256-byte loads of one wave
constexpr unsigned kBellQuit = 0xffffffffu; // never a call's sequence number
constexpr int kResLinePayload = 14;
struct ResidentArgs {
    const unsigned *host_bell; // the doorbell the HOST writes: [bell_copies][kResMaxChannels][16] -- device memory that the host reaches
                               // through the PCIe BAR (bell_copies = 8: one per blockIdx % 8), or pinned host memory (1 copy)
    int bell_copies;
    unsigned *host_lines;      // pinned host: [workgroups][lines per workgroup][16]
    unsigned *host_state;      // pinned host: [0] why the kernel ended (0: it runs), [1] calls the master served
    unsigned *dev_quit;        // device: set by the master when it leaves (every workgroup polls the host: forward == 0)
    unsigned *dev_bell;        // device: [8][K lines] copies of the doorbell written by the master (forward != 0)
    int forward;               // 1: only the master polls the host and forwards; 0: every workgroup polls the host
    unsigned start_seq;        // sequence number of the last call served before this launch
    unsigned max_calls;        // the kernel ends after this many calls ...
    long long idle_ticks;      // ... or this long without one (100 MHz wall clock) ...
    long long life_ticks;      // ... or this long after its start, whatever happens
};
enum ResidentExit : unsigned { kResidentRuns = 0, kResidentQuit = 1, kResidentIdle = 2, kResidentLife = 3, kResidentCalls = 4 };
bool dc_has_resident_instance(int ant_tile, int taps, int format);
hipError_t launch_dc_resident(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s);
// workgroups of the instance that one compute unit holds at once (occupancy API with the launch's LDS)
hipError_t dc_resident_blocks_per_cu(const DcLaunch &cfg, int *blocks_per_cu);
template <int FMT> hipError_t launch_dc_resident_fmt(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s, int *blocks_per_cu);

// carrier table of one segment: [steps <= kUcarSteps][samples of a lane's groups, G * S <= 8][re, im] floats per channel
constexpr int kUcarSteps = 8;
constexpr int kUcarFloats = kUcarSteps * 8 * 2;
#ifndef GAT_DC_SEG_ENTRIES
#define GAT_DC_SEG_ENTRIES 8192
#endif
// Most steps whose code replica one workgroup produces at once (a "segment").  One channel per workgroup: ~8192 entries
// (39 KB of LDS) when a wave carries 3-4 antennas -- those instances are limited to 3 workgroups per CU by registers
// anyway, and longer segments mean fewer barriers (configs[1]: 0.846 vs 0.822 of HBM) --, ~4096 entries (25 KB) for 1-2
// antennas per wave, where LDS is what limits the workgroups per CU; ~8192 entries over the channels of a channel-
// looping workgroup.  The host may ask for fewer (short blocks).
constexpr int dc_segment_steps(int chunk, int kt, int mt)
{
    const int s = (kt == 1 ? (mt >= 3 ? GAT_DC_SEG_ENTRIES : 4096) : 8192 / kt) / chunk;
    return s < 2 ? 2 : (s > kUcarSteps ? kUcarSteps : s);
}
// Floats of LDS per channel for one segment's replica: entry i <-> sample (segment start) + shifts[0] + i, linear, so
// that the chips of a lane's S consecutive samples are ONE 8-byte-aligned vector read per tap (ds_read2_b64).  Taps at
// an odd distance from the first read a second copy shifted by one entry (rep_copy_stride floats further); the host
// then halves the segment so that both fit here.  Room: segment samples + kMaxReplicaSpan taps + one entry per
// producer thread of overshoot; at least one step with two copies.
constexpr int dc_rep_copy_floats(int steps, int chunk, int span, int threads = kThreads) { return (steps * chunk + span + threads + 2 + 1) & ~1; }
constexpr int dc_rep_chan_floats_steps(int steps, int chunk, int span = kMaxReplicaSpan)
{
    const int one = dc_rep_copy_floats(steps, chunk, span);
    const int two = 2 * dc_rep_copy_floats(1, chunk, span);
    return ((one > two ? one : two) + 7) & ~7;
}
constexpr int dc_rep_chan_floats(int chunk, int kt, int mt) { return dc_rep_chan_floats_steps(dc_segment_steps(chunk, kt, mt), chunk); }
// dynamic LDS of one four-wave dc_kernel workgroup: per-channel constants, reduction scratch, carrier table, one segment's
// replica (chan_floats per channel: the host may size it for fewer steps than dc_segment_steps), chip tables
constexpr size_t dc_lds_bytes_floats(int kt, int code_row_stride, int chan_floats)
{
    return (size_t)kt * 32 + (size_t)kt * 4 * 64 * sizeof(float) + (size_t)kt * kUcarFloats * sizeof(float) +
           (size_t)kt * chan_floats * sizeof(float) + (size_t)kt * code_row_stride;
}
constexpr size_t dc_lds_bytes(int kt, int mt, int code_row_stride, int chunk)
{
    return dc_lds_bytes_floats(kt, code_row_stride, dc_rep_chan_floats(chunk, kt, mt));
}
// Samples of a group handled at once in the one-channel step (chips, phasors and wipe-off products of SB samples live
// together): the whole group, half of the eight-sample groups of int8 pairs (a fourth wave per SIMD, round 3), and two of
// four for the tiles with 25-40 accumulator registers (four antennas x five taps: 145 -> 128 registers, round 4).
constexpr int dc_sub_batch(int s, int mt, int l, int kt, int aw = 1)
{
    if (aw == 2 && mt == 2 && kt == 2 && s == 4) return 2; // the two-channel 2 x 2 tile: two-sample passes (chips of both channels live)
    return s > 4 ? 4 : (s == 4 && kt == 1 && 2 * mt * l > 24 && 2 * mt * l <= 40 ? 2 : s);
}
// Occupancy hints (__launch_bounds__ of dc_kernel; the host sizes the LDS segment for the same number of workgroups per
// CU).  scripts/kernel_resources.sh prints what every instance needs; a bound tighter than that spills into the step loop
// (1.2-3x slower, rounds 1-2).  Round 4 took the per-lane phasor state and three hoisted fill addresses out of every
// instance: see the rule below; the channel-looping instances with more than 40 accumulator registers -> two waves
// (<= 256 registers, no AGPR copies), without spills for float and int16 samples.  The int8 forms (eight samples per
// group) keep round 3's bounds: 13-92 registers over 256 in their channel-looping instances (AGPR copies, one wave per
// SIMD; by default such shapes run on the split-bf16 matrix kernel).
constexpr int dc_min_waves(int mt, int l, int kt, int d, int fmt, int aw = 1)
{
    // the two-channel 2 x 2 tile: four waves per SIMD up to five taps (127 registers at five; three waves measured no faster than
    // the one-channel tile, profiles/r05/ab_kt2.txt), three beyond
    // (ComplexF32 pairs: two two-sample groups per lane and step cost ~20 registers more -- four waves up to three taps)
    if (mt == 2 && aw == 2)
        return fmt == GAT_LAYOUT_INTERLEAVED_I8 ? 2 : fmt == GAT_LAYOUT_INTERLEAVED ? (l <= 3 ? 4 : l <= 6 ? 3 : 2) : (l <= 5 ? 4 : 3);
    const int accs = 2 * mt * l * kt;
    const bool i8 = fmt == GAT_LAYOUT_INTERLEAVED_I8;
    if (accs > 40) return i8 ? 1 : (accs <= 48 && kt == 2 ? 3 : 2); // (four antennas x three taps x two channels: 167 registers)
    if (d != 1 || i8 || kt != 1) return 3;
    // one channel, one sample set: four waves where the step fits 128 registers -- up to 24 accumulator registers in every
    // format, up to 40 where the step runs in two-sample passes (dc_sub_batch: four-sample groups, i.e. planar float and
    // int16 samples; the four-antenna five-tap planar instance spills two registers outside its step loop for it)
    const int s = dc_group_samples(4, fmt);
    return accs <= 24 || dc_sub_batch(s, mt, l, kt) < s ? 4 : 3;
}
// does an instance of dc_kernel exist for this combination (gat_dc.h: dc_instance)
bool dc_has_instance(int ant_tile, int taps, int vec, int aw, int kt, int nw = 4, int depth = 1);
// one-wave workgroups: steps per segment and LDS bytes (replica sized for the launch's own tap span)
constexpr int kOneWaveSegSteps = 4;
// Deepest sample prefetch an instance family is built with.  Two register sets (steps c+1 and c+2 in flight) exist for
// the four-antenna, <= 3-tap, one-channel tile only -- the configs[1] family, streaming every byte once at the HBM rate:
// measured + 1.3 % there on fast and slow boxes alike; every other family measured no gain or a loss.
constexpr int dc_depth_max(int mt, int l, int aw, int kt, int nw)
{
    // (round 4: and the one-antenna one-wave tile of short blocks in a long stream -- 6-7 waves per SIMD with ONE step of 2 KB
    // in flight each are 13.5 KB per SIMD, 0.78 of the HBM rate by Little's law at 2.2 us: what that shape measured, 0.74)
    return (nw == 4 && aw == 1 && kt == 1 && mt == 4 && l <= 3) || (nw == 1 && aw == 1 && kt == 1 && mt == 1 && l <= 3) ? 2 : 1;
}
constexpr size_t dc_lds_bytes_one_wave(int rep_chan_floats, int code_row_stride)
{
    return 32 + 64 * sizeof(float) + kUcarFloats * sizeof(float) + (size_t)rep_chan_floats * sizeof(float) + (size_t)code_row_stride;
}

// Which instances carry the replica fill by quads / from sign-bit chip tables (the host asks for them nowhere else: every
// form costs an instance scalar registers whether it runs or not)
constexpr bool dc_fill_quads(int aw, int kt, int nw) { return nw == 4 && aw == 2 && kt == 2; }
constexpr bool dc_bit_tables(int nw) { return nw == 4; }

// Which instances exist.  Register accumulators 2 * MT * L * KT <= 64; antenna-parallel waves (AW > 1) need full
// 4-antenna tiles; unaligned input (VEC == 1, scalar loads) is served one antenna per workgroup.
constexpr bool dc_instance(int mt, int l, int vec, int aw, int kt, int nw = 4, int depth = 1)
{
    if (depth != 1 && !(vec == 4 && depth == dc_depth_max(mt, l, aw, kt, nw))) return false;
    // one-wave workgroups: short blocks of one- and two-antenna tiles
    if (nw != 4 && !(nw == 1 && vec == 4 && aw == 1 && kt == 1 && mt <= 2)) return false;
#ifdef GAT_DC_DEV // development builds: only the instances the BASELINE shapes use (compiles in seconds)
    if (!((mt == 1 || mt == 4 || mt == 2) && (l == 3 || l == 5) && vec == 4)) return false;
#endif
    if (mt < 1 || mt > kMaxAntTile || l < 1 || l > kMaxTapsPerLaunch) return false;
    if (vec != 4) return vec == 1 && mt == 1 && aw == 1 && kt == 1;
    // the two-channel 2 x 2 tile: two waves of two antennas each on the same samples, two sample sub-chunks, two channels per
    // workgroup (the one-channel form of it measured 4-12 % slower than one wave of four antennas at 4, 5 and 6 waves per SIMD:
    // profiles/r05/ab_aw2.txt -- it does not exist)
    if (mt == 2 && aw == 2) return kt == 2 && nw == 4 && depth == 1;
    if (aw != 1 && (mt != 4 || aw != 4)) return false;
    // several channels per workgroup only with antenna-parallel waves: measured on MI355X, a channel loop over one
    // antenna tile never beat separate channel workgroups sharing the tile through L2 (M = 1, 4; K = 8, 12)
    if (kt != 1 && aw != 4) return false;
    return (kt == 1 || kt == 2 || kt == 4) && mt * l * kt <= 48;
}

// Arguments of the matrix-core kernel (gat_mfma.hip): 16-antenna tiles, planar f32 input.
constexpr int kMfmaMaxTaps = 16;     // 2 * CT * L <= 32 columns with CT >= 1
constexpr int kMfmaMaxSpan = 768;    // replica halo served per 256-sample step
struct MfArgs {
    const void *re; // planar f32: re plane; interleaved formats (split-bf16 kernel only): base pointer
    const void *im; // planar f32: im plane; otherwise null
    const gat_channel_params *params;
    const int8_t *codes;
    const uint32_t *code_bits; // split-bf16 kernel: sign-bit tables [P][code_bits_stride]
    const void *zeros;         // split-bf16 kernel: 16 zero bytes (target of out-of-range sample loads)
    float *out_re;
    float *out_im;
    float *partial;
    long long N, ant_stride, block_stride;
    double fs;
    int M, K, B, L, Lc, num_prns, code_row_stride;
    int CT;              // f32 kernel: channels per 32-column tile = 16 / L
    int nslots;          // split-bf16 kernel: channel slots per workgroup (columns packed flat)
    int chan_groups;     // ceil(ceil(K / CT) / NCT)
    int ant_tiles;       // M / 16 (f32 kernel) or M / (16 * rt) (split-bf16 kernel)
    int splits, steps_per_split, total_steps, num_tiles;
    int max_abs_shift, rep_span, rep_stride;
    int code_bits_stride; // dwords per sign-bit row (multiple of 4)
    int codes_in_lds;    // 1: the workgroup's chip tables are staged in LDS (they fit)
    int rep_ring;        // split-bf16 kernel: entries of a channel slot's chip-sign ring (a multiple of the tile, >= rep_span + 2 tiles; rep_stride >= rep_ring + rep_span + tile)
    int mb_mode;         // split-bf16 kernel: MbMode of this launch (the host's choice: mb_mode(), or kMbThree on request for int16)
    unsigned long long *dbg; // diagnostic builds only (GAT_MFMA_STAMPS): per-wave cycle sums
    unsigned flags;
    int shifts[kMfmaMaxTaps];    // ascending
    int tap_index[kMfmaMaxTaps]; // position in the caller's list
};
// Geometry of the matrix-core kernels that the host's planner and the kernels share (gat_mfma.hip, gat_mfma_bf16.hip).
// split-bf16 kernel: workgroup size -- 4 consumer waves + 12 producer waves (4 per SIMD, 128 VGPRs) where the instance fits
// that register budget, otherwise + 8 producer waves (3 per SIMD, 168 VGPRs)
constexpr int mb_threads(int RT, int NCT) { return ((RT == 2 && NCT == 1) || (RT == 4 && NCT == 2)) ? 768 : 1024; }
// consumer waves: one per SIMD, or (experiment GAT_MB_CW8) two per SIMD for the 4 x 4 instance
#ifdef GAT_MB_CW8
constexpr int mb_consumer_waves(int RT, int NCT) { return (RT == 4 && NCT == 4) ? 8 : 4; }
#else
constexpr int mb_consumer_waves(int, int) { return 4; }
#endif
constexpr int kMbMaxChain = 8192;  // samples per accumulation chain (f32 rounding of the running sum)
constexpr int kMbMaxSlots = 24;    // channel slots per workgroup (header size)
constexpr int kMbHeader = 1536;    // ChanInfoB[<= 20] (64 B each) + slack, 16-byte aligned
constexpr int mb_tile_samples(int RT, int NCT)
{
    const int t = 32 * (4 / NCT), cap = 128 / RT;
    return t < cap ? t : cap;
}
constexpr int mb_slots(int nct, int L, int K)
{ // upper bound of the channels a workgroup's 32 * nct flat columns touch
    const int s = (32 * nct + 2 * L - 1) / (2 * L) + 1;
    return s < K ? s : K;
}
// How the split-bf16 kernel lays the cross products along the MFMA's 16 reduction slots (per 32-lane half: 8 slots):
//   kMbThree  float / int16 samples: x = hi + mid + lo, w = hi + mid + lo, 8 products per sample  -> 1 sample per half and MFMA
//   kMbOne    int8 samples: exact in ONE bf16 term, 3 products + a zero slot per sample          -> 2 samples per half and MFMA
//   kMbTwo    int16 samples, round 5: exact in TWO bf16 terms (a = x rounded to 8 significant bits, b = x - a: at most 7 bits,
//             |b| <= 2^-8 |a|), 5 products per sample {a h, a m, a l, b h, b m} (b * lo(w) is 2^-24 of the product: below f32), laid as a
//             STREAM of slots across consecutive MFMAs: 8 samples per half every 5 MFMAs -- 1.6 samples per half and MFMA.
//             Needs the half's sample count per step (T / consumer waves per tile / 2) to be a multiple of 8.
enum MbMode : int { kMbThree = 0, kMbOne = 1, kMbTwo = 2 };
constexpr int mb_mode(int rt, int nct, int fmt, bool force_three = false)
{
    if (fmt == GAT_LAYOUT_INTERLEAVED_I8) return kMbOne;
    const int wpt = mb_consumer_waves(rt, nct) / nct, hs = mb_tile_samples(rt, nct) / (wpt > 0 ? wpt : 1) / 2;
    return fmt == GAT_LAYOUT_INTERLEAVED_I16 && !force_three && wpt > 0 && hs >= 8 && hs % 8 == 0 ? kMbTwo : kMbThree;
}
// two-term path: bytes of one LDS row of slot-ordered bf16 terms (10 bytes per sample), an odd multiple of 16 (rows of one
// 16-lane group of a 16-byte read then start on distinct bank quads)
constexpr int mb_two_row_bytes(int T) { return ((10 * T + 15) / 16 * 16) | 16; }
// chip-sign ring of the split-bf16 kernel: ring length and the least row length for a tap span (gat_mfma_bf16.hip, s_rep)
// (instances of two or four row tiles; those of one keep two buffers of span + T entries and copy the overlap)
#ifdef GAT_MB_NO_RING // A/B build: two buffers and the overlap copy in every instance, as up to round 4
constexpr bool mb_rep_ring_rows(int) { return false; }
#else
constexpr bool mb_rep_ring_rows(int rt) { return rt >= 2; }
#endif
constexpr int mb_rep_ring(int rt, int T, int span) { return mb_rep_ring_rows(rt) ? (span + 2 * T + T - 1) / T * T : 0; }
constexpr int mb_rep_row(int rt, int T, int span) { return mb_rep_ring_rows(rt) ? mb_rep_ring(rt, T, span) + span + T : span + T; }
constexpr size_t mb_lds_bytes(int rt, int nct, int fmt, int nslots, int rep_stride, int code_bits_stride, int mode = -1)
{
    if (mode < 0) mode = mb_mode(rt, nct, fmt);
    const bool x1 = mode == kMbOne;
    const int T = mb_tile_samples(rt, nct);
    const int xs = x1 ? T + 2 : T + 1, wbytes = x1 ? 8 : 16;
    const size_t xw = mode == kMbTwo ? (size_t)2 * rt * 32 * mb_two_row_bytes(T) + (size_t)2 * (2 * nslots + 1) * mb_two_row_bytes(T)
                                     : (size_t)2 * rt * 32 * xs * 8 + (size_t)2 * (2 * nslots + 1) * xs * wbytes;
    return (size_t)kMbHeader + xw + (size_t)(((mb_rep_ring_rows(rt) ? 1 : 2) * nslots * rep_stride + 3) & ~3) * 4 + (size_t)nslots * code_bits_stride * 4;
}
// f32-MFMA kernel: 256-sample tiles, 32 planes per row tile
constexpr int kMfTile = 256;
constexpr int kMfXStride = kMfTile + 1; // floats per plane row in LDS: odd (bank spread)
constexpr size_t mf_lds_bytes(int nct, int ct, int rep_stride, int code_row_stride, int codes_in_lds)
{
    return (size_t)1024 + (size_t)2 * 32 * kMfXStride * sizeof(float) + (size_t)2 * nct * ct * rep_stride * sizeof(float) +
           (codes_in_lds ? (size_t)nct * ct * code_row_stride + 16 : 0);
}
hipError_t launch_mfma(const MfArgs &a, int nct, unsigned grid, unsigned lds_bytes, hipStream_t s);
size_t mfma_lds_bytes(int nct, int ct, int rep_stride, int code_row_stride, int codes_in_lds);
// split-bf16 matrix-core kernel (gat_mfma_bf16.hip): rt = 16-antenna row tiles per workgroup (1, 2, 4)
hipError_t launch_mfma_bf16(const MfArgs &a, int rt, int nct, int fmt, unsigned grid, unsigned lds_bytes, hipStream_t s);
size_t mfma_bf16_lds_bytes(int rt, int nct, int fmt, int nslots, int rep_stride, int code_bits_stride, int mode);
int mfma_bf16_mode(int rt, int nct, int fmt, bool force_three);
int mfma_bf16_slots(int nct, int L, int K); // channel slots per workgroup (flat column packing)
int mfma_bf16_max_slots();
int mfma_bf16_tile_samples(int rt, int nct);
int mfma_bf16_max_chain(); // samples one accumulation chain may cover
int mfma_bf16_threads(int rt, int nct); // workgroup size of the instance
int mfma_bf16_producer_threads(int rt, int nct);

// Launchers implemented in gat_kernels.hip.  All return hipError_t of the launch.
hipError_t launch_dc(const DcArgs &a, const DcLaunch &cfg, hipStream_t s);
// one translation unit per sample format (gat_dc_f*.hip), so that the instances compile in parallel
template <int FMT> hipError_t launch_dc_fmt(const DcArgs &a, const DcLaunch &cfg, hipStream_t s);
// done_counter != null: the launch ends by storing flag_seq into host_flag (completion flag, see gat_dc.h)
hipError_t launch_finalize(const float *partial, float *out_re, float *out_im, int splits, int elems,
                           long long groups, hipStream_t s, unsigned *done_counter = nullptr, unsigned *host_flag = nullptr,
                           unsigned flag_seq = 0);
hipError_t launch_gen_code_replica(float *rep, long long count, const int8_t *code_row, int Lc,
                                   double fc, double fs, double tau, long long first_shift,
                                   bool f32_coordinates, hipStream_t s);
// the texture-addressing study (gat.h gat_gen_code_replica_texaddr): fixed-point normalised coordinate / texel address
hipError_t launch_gen_code_replica_texaddr(float *rep, long long count, const int8_t *code_row, int Lc, double fc, double fs,
                                           double tau, long long first_shift, int coord_frac_bits, int texel_frac_bits,
                                           hipStream_t s);
// a kernel that only reads `bytes` (16-byte groups) of device memory (gat.h gat_debug_read_stream); sink: 4 bytes nobody reads
hipError_t launch_read_stream(const void *dev, size_t bytes, int variant, int num_cus, float *sink, hipStream_t s);
hipError_t launch_gen_code_replica_multi(float *rep, long long count, long long row_stride, int K,
                                         const gat_channel_params *params, const int8_t *codes, int code_row_stride,
                                         int Lc, int num_prns, double fs, long long first_shift, hipStream_t s);
hipError_t launch_accumulate_debug(const float *sig_re, const float *sig_im, long long N, int M, long long ant_stride,
                                   const gat_channel_params &P, const int8_t *code_row, int Lc, double fs, int L,
                                   const int *shifts_dev, float *car_re, float *car_im, float *dw_re, float *dw_im,
                                   float *acc_re, float *acc_im, hipStream_t s);
hipError_t launch_gen_signal(void *re, void *im, int format, long long N, int M,
                             long long ant_stride, long long block_stride, int B, int K,
                             const gat_channel_params *params, const int8_t *codes, int code_row_stride,
                             int Lc, int num_prns, double fs, float amplitude, const float *steering_cycles, float noise_sigma,
                             unsigned long long seed, hipStream_t s);
hipError_t launch_reduce_stage1(const float *in_re, const float *in_im, long long n, int cols,
                                int chunks, float *partial, hipStream_t s);
hipError_t launch_tracking_update(const float *acc_re, const float *acc_im, int K, int M, const gat_loop_config &cfg,
                                  gat_loop_state *state, const gat_channel_params *cur, gat_channel_params *next,
                                  hipStream_t s);


} // namespace gat
