// gat_resident.h -- the resident correlator: single-block calls without a kernel launch (gfx950 / wave64).
//
// The reference times ONE 1 ms block per call (src/benchmarks.jl:120-146); on this platform a launch plus the wait for its
// end costs 7 us before the kernel has done anything (scripts/probes/sync_probe.hip), a doorbell in pinned host memory that
// a kernel already on the device polls costs 2-3 us there and back (scripts/probes/doorbell_probe.hip).  dc_resident_kernel
// is the fused correlator of gat_dc.h (the same body, gat_dc_body.inc) inside a loop, and every workgroup of it is on its
// own:
//
//   poll the doorbell (in DEVICE memory where the host can write there through the PCIe BAR: a ring is then a posted write and
//   every poll a local read -- 2.10 us there and back, median, against 2.50 with the doorbell in pinned host memory and
//   a third of the spread, scripts/probes/doorbell_bar_probe.hip --; else in pinned host memory, polled by every
//   workgroup or, with more than ~20 of them, by the master, which forwards it through device memory)
//   -> the call's channel records and block offset arrive WITH the ring (one 64-byte line per channel)
//   -> correlate this workgroup's share (antenna tile, channel, sample split) with system-scope sample loads (the signal may
//   have been rewritten by a copy engine or another kernel since the last call; no cache invalidate) -> post its sums to the
//   host as result lines
//   (64 bytes: 14 values | check | the call's number; plain stores, no fence, nothing to wait for) -> poll again.
//
// The host walks the workgroups' lines in slot order while it waits: a workgroup whose lines carry the call's number and pass
// their check is added to the outputs (the second stage of the ordinary call, splits in rising order: deterministic).  No workgroup exchanges
// anything with another one: no arrival counter, no release / acquire pair, no partial sums in device memory -- each of
// those was a trip through the memory system on the critical path (0.5-1.2 us apiece, profiles/r04/resident/).
// Chip tables stay staged in LDS from call to call; the kernel's arguments are read once.
//
// Lifetime: the kernel ends BY ITSELF, whatever the host does -- after `max_calls` calls, after `idle_ticks` without a
// ring, after `life_ticks` in total, or when the host rings kBellQuit.  Workgroup 0 (the master) decides: it sets a word
// in device memory that the others read with every poll (or, when it forwards the rings, puts kBellQuit into the
// forwarded doorbells), and tells the host why it left.  The others leave when that word
// is set, or when 1.5 x life_ticks have passed (the master can no longer be there).  Every wait in here is a poll with a
// deadline on the constant 100 MHz clock: the grid drains even if its workgroups never run together.  A ring that arrives
// while the master is leaving may be served by some workgroups and not by others: the host sees the kernel gone and the
// call unanswered, starts the kernel again and the call is served whole (same values: the results are deterministic).
//
// Which shapes: one block per call, K <= 16 channels, one tap launch (<= 8 taps within 2048 samples), 16-byte aligned block
// starts with N a multiple of the load group (what runs as ONE vector launch otherwise), all four sample formats; antennas
// in tiles of MT <= 4, one tile and one channel per workgroup (AW = KT = 1).
#pragma once

#include "gat_dc.h"

namespace gat {

constexpr bool dc_resident_instance(int mt, int l) { return dc_instance(mt, l, 4, 1, 1, 4, 1); }

struct ResidentEnv {
    const unsigned *line; // LDS: the call's doorbell lines
    float *stage;         // LDS: the workgroup's sums, in the order of the body's output loop (value o = 2 * (l * MT + m) + comp)
    __device__ __forceinline__ gat_channel_params params(int k) const
    {
        const unsigned *w = line + k * kBellDwords;
        auto dbl = [&](int i) { return __longlong_as_double((long long)(((unsigned long long)uni(w[i + 1]) << 32) | uni(w[i]))); };
        gat_channel_params P;
        P.prn = (int)uni(w[2]);
        P.reserved = 0;
        P.code_freq_hz = dbl(4);
        P.carrier_freq_hz = dbl(6);
        P.code_phase_chips = dbl(8);
        P.carrier_phase_cycles = dbl(10);
        return P;
    }
    __device__ __forceinline__ size_t block_offset() const
    {
        return (size_t)(((unsigned long long)uni(line[13]) << 32) | uni(line[12]));
    }
};

// (no occupancy bound: a CU holds one resident workgroup, all registers are its own -- bounded like dc_kernel the arguments
// that stay live across the loop spilled into the step)
template <int MT, int L, int FMT>
__global__ void __launch_bounds__(256) dc_resident_kernel(const DcArgs a, const ResidentArgs r)
{
    constexpr int VEC = 4, AW = 1, KT = 1, NW = 4, D = 1;
    constexpr bool KEEP = false;
    constexpr int NVAL = 2 * MT * L;                                          // sums one workgroup posts per call
    constexpr int LW = (NVAL + kResLinePayload - 1) / kResLinePayload;        // its result lines
    __shared__ __attribute__((aligned(16))) unsigned s_bell[kResMaxChannels * kBellDwords];
    __shared__ unsigned s_ctl[1]; // the call's sequence number or kBellQuit
    __shared__ float s_out[64];

    kernarg_prefetch<sizeof(DcArgs) + sizeof(ResidentArgs)>();
    // the working workgroup's ordinal (the body's decode of blockIdx: tile, channel group); padding of the grid leaves
    const unsigned tile0 = ((blockIdx.x >> 3) / (unsigned)a.KG) * 8u + (blockIdx.x & 7u);
    if (tile0 >= (unsigned)a.num_tiles) return;
    const unsigned slot = tile0 * (unsigned)a.KG + (blockIdx.x >> 3) % (unsigned)a.KG; // (B = 1: tile = antenna group * splits + split)
    const bool master = blockIdx.x == 0;
    const int K = a.K;
    int staged_prn[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) staged_prn[kk] = -1;
    unsigned last = r.start_seq, calls = 0, why = kResidentRuns;
    const long long t_start = wall_clock64();
    long long t_last = t_start;
    const ResidentEnv env{s_bell, s_out};

#ifdef GAT_RES_STAMPS
    long long t_seen_ = 0;
#endif
    for (;;) {
        // ---- wait for a ring: wave 0 reads all K lines with ONE 16-byte load per lane (lane i <-> dwords 4i .. 4i+3: a line is
        // four lanes; sixteen channels are one load -- K > 4 used to cost a second trip, ~1 us).  Doorbell in device memory (the
        // host writes it through the BAR, eight copies): every workgroup polls its copy, r.forward = 0.  Doorbell in pinned host
        // memory -- few workgroups: every one polls it itself (nothing between the ring and any workgroup); many: reads of one host line queue up
        // behind each other (~0.15 us apiece: 33 pollers took 10 us to see a ring), so only the master polls the host and
        // copies what it sees -- rings and its decision to leave -- into eight doorbells in device memory (one per
        // blockIdx % 8, on different memory channels) that the others poll.  Every line proves itself (number | check | number):
        // no order among the lines of a ring is needed, on either hop.
        if (threadIdx.x < 64) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int ln = (int)threadIdx.x;
            const bool from_host = master || r.forward == 0;
            const unsigned *src = (from_host ? r.host_bell + (blockIdx.x & (unsigned)(r.bell_copies - 1)) * (kResMaxChannels * kBellDwords)
                                             : r.dev_bell + (blockIdx.x & 7u) * (kResMaxChannels * kBellDwords)) + 4 * ln;
            const bool mine = ln < 4 * K;
            u32x4 v = {0u, 0u, 0u, 0u};
            unsigned seq = last;
            // (one load of the doorbell in flight: several loads of one uncached line queue up behind each other and the oldest
            // sees the ring.  Direct polling: the master's word in device memory is read with every poll, both loads in flight
            // together -- one asm statement, or the compiler waits for the first before it issues the second)
            const bool ask_leave = !(master || r.forward != 0);
            unsigned leave_w = 0u;
            auto load16 = [&]() {
                u32x4 w;
                if (ask_leave)
                    asm volatile("global_load_dword %1, %3, off sc1\n\tglobal_load_dwordx4 %0, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(w), "=&v"(leave_w) : "v"(src), "v"(r.dev_quit) : "memory");
                else if (from_host) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(src) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(src) : "memory");
                return w;
            };
            // every line whole and of ring `want`: first and last dword = want, XOR of dwords 0 .. 13 = dword 14
            auto whole = [&](u32x4 w, unsigned want) {
                unsigned x = w.x ^ w.y ^ ((ln & 3) == 3 ? 0u : (w.z ^ w.w));
                x ^= __shfl_xor(x, 1, 64); x ^= __shfl_xor(x, 2, 64);
                const unsigned head = __shfl(w.x, ln & ~3, 64), chk = __shfl(w.z, ln | 3, 64), tail = __shfl(w.w, ln | 3, 64);
                const bool ok = !mine || (x == chk && head == want && tail == want);
                return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
            };
            for (;;) {
                if (mine) v = load16();
                const unsigned leave = (unsigned)__builtin_amdgcn_readlane((int)leave_w, 0); // (lane 0 always loads)
                seq = (unsigned)__builtin_amdgcn_readlane((int)v.x, 0);
                if (seq == kBellQuit) { why = kResidentQuit; break; }
                // (a whole pending ring is served BEFORE the master's leave word is honoured: a workgroup that starts late --
                // more workgroups than the device holds at once -- would otherwise see "leave" first and exit without serving,
                // and the host would restart the same schedule until its deadline)
                if (seq != last) {
                    if (whole(v, seq)) break;
                    seq = last; // a line caught half-written (or the host has not reached line 0's siblings yet): read again
                }
                if (leave != 0u) { why = kResidentQuit; seq = kBellQuit; break; }
                const long long now = wall_clock64();
                if (master) {
                    unsigned reason = kResidentRuns;
                    if (calls >= r.max_calls) reason = kResidentCalls;
                    else if (now - t_last > r.idle_ticks) reason = kResidentIdle;
                    else if (now - t_start > r.life_ticks) reason = kResidentLife;
                    if (reason != kResidentRuns) { why = reason; seq = kBellQuit; break; }
                } else if (now - t_start > r.life_ticks + (r.life_ticks >> 1)) { // the master is gone and has not said so
                    why = kResidentLife;
                    seq = kBellQuit;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (master) {
                if (r.forward != 0) { // the ring, or the decision to leave, to the eight device doorbells
                    u32x4 fwd = v;
                    if (seq == kBellQuit && ln == 0) fwd.x = kBellQuit;
                    if (mine)
                        for (int c8 = 0; c8 < 8; ++c8) {
                            unsigned *dst = r.dev_bell + c8 * (kResMaxChannels * kBellDwords) + 4 * ln;
                            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(fwd) : "memory");
                        }
                } else if (seq == kBellQuit && ln == 0) {
                    __hip_atomic_store(r.dev_quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            *reinterpret_cast<u32x4 *>(&s_bell[4 * ln]) = v;
            if (ln == 0) s_ctl[0] = seq;
#ifdef GAT_RES_STAMPS
            t_seen_ = wall_clock64();
#endif
        }
        __syncthreads();
        const unsigned seq = uni(s_ctl[0]);
        if (seq == kBellQuit) break;
        last = seq;
        // The signal of this call was written by somebody else (copy engine, another kernel, the host) since the last call:
        // nothing of it may come from this XCD's caches.  The body's sample loads are system-scope loads for that -- a cache
        // invalidate here (a system-scope acquire fence) costs every workgroup ~0.3 us and the invalidates of an XCD's workgroups
        // queue up behind each other: 1.8 us from the ring to the first loads with 20 workgroups, 8.3 us with 240
        // (profiles/r04/resident/v10_*; -DGAT_RES_FENCE builds that form).
#ifdef GAT_RES_FENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#endif

#ifdef GAT_RES_STAMPS // development builds: where a call's time goes (100 MHz clock), read back by gat_resident_close
        long long st_[10] = {};
        st_[0] = wall_clock64();
        const long long cyc0_ = clock64();
#undef GAT_DC_LAT_CUT_AT
#define GAT_DC_LAT_CUT_AT(n) do { st_[n] = wall_clock64(); } while (0)
#endif
        { // the correlator itself: the text dc_kernel is made of
#define GAT_DC_BODY_RESIDENT 1
#include "gat_dc_body.inc"
#undef GAT_DC_BODY_RESIDENT
        }

        // ---- this workgroup's NVAL sums (staged in LDS by the body) go to the host as LW result lines
        __syncthreads();
        for (int t = (int)threadIdx.x; t < LW * 16; t += 256) {
            const int i = t & 15, pidx = (t >> 4) * kResLinePayload + i;
            unsigned v = (i < kResLinePayload && pidx < NVAL) ? __float_as_uint(s_out[pidx]) : 0u;
            unsigned x = i < kResLinePayload ? v : 0u;
            x ^= __shfl_xor(x, 8, 64); x ^= __shfl_xor(x, 4, 64); x ^= __shfl_xor(x, 2, 64); x ^= __shfl_xor(x, 1, 64);
            if (i == 14) v = x ^ seq;
            if (i == 15) v = seq;
            __hip_atomic_store(r.host_lines + (size_t)slot * (LW * 16) + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#ifdef GAT_RES_STAMPS
        st_[6] = wall_clock64();
        if (master && threadIdx.x == 0)
            for (int i = 0; i < 7; ++i) r.host_state[4 + i] = (unsigned)(st_[i] - st_[0]);
        if (master && threadIdx.x == 0) {
            r.host_state[11] = (unsigned)(st_[0] - t_seen_);
            r.host_state[12] = (unsigned)(st_[8] - st_[0]);
            r.host_state[13] = (unsigned)(st_[7] - st_[0]);
            r.host_state[14] = (unsigned)(clock64() - cyc0_); // shader-clock cycles of the same span as st_[6] - st_[0]
        }
#endif
        ++calls;
        t_last = wall_clock64();
    }
    if (master && threadIdx.x == 0) {
        r.host_state[1] = calls;
        __hip_atomic_store(r.host_state, why, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// blocks_per_cu != null: nothing is launched; *blocks_per_cu = workgroups of the instance one compute unit holds at once
// (every workgroup of a resident kernel has to be ON the device for a call to complete: gat_resident_open sizes by it)
template <int FMT, int MT>
static hipError_t launch_dc_resident_m(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s, int *blocks_per_cu)
{
    auto go = [&](auto l_c) -> hipError_t {
        constexpr int L = decltype(l_c)::value;
        if constexpr (dc_resident_instance(MT, L)) {
            if (blocks_per_cu)
                return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, dc_resident_kernel<MT, L, FMT>, 256, cfg.lds_bytes);
            hipLaunchKernelGGL((dc_resident_kernel<MT, L, FMT>), dim3(cfg.grid), dim3(256), cfg.lds_bytes, s, a, r);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    };
    switch (cfg.taps) {
    case 1: return go(std::integral_constant<int, 1>{});
    case 2: return go(std::integral_constant<int, 2>{});
    case 3: return go(std::integral_constant<int, 3>{});
    case 4: return go(std::integral_constant<int, 4>{});
    case 5: return go(std::integral_constant<int, 5>{});
    case 6: return go(std::integral_constant<int, 6>{});
    case 7: return go(std::integral_constant<int, 7>{});
    case 8: return go(std::integral_constant<int, 8>{});
    default: return hipErrorInvalidValue;
    }
}

template <int FMT>
hipError_t launch_dc_resident_fmt(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s, int *blocks_per_cu)
{
    if (cfg.vec != 4 || cfg.aw != 1 || cfg.kt != 1 || cfg.nw != 4 || cfg.depth != 1) return hipErrorInvalidValue;
    switch (cfg.ant_tile) {
    case 1: return launch_dc_resident_m<FMT, 1>(a, cfg, r, s, blocks_per_cu);
    case 2: return launch_dc_resident_m<FMT, 2>(a, cfg, r, s, blocks_per_cu);
    case 3: return launch_dc_resident_m<FMT, 3>(a, cfg, r, s, blocks_per_cu);
    case 4: return launch_dc_resident_m<FMT, 4>(a, cfg, r, s, blocks_per_cu);
    default: return hipErrorInvalidValue;
    }
}

} // namespace gat
