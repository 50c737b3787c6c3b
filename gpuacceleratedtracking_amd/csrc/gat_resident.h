// gat_resident.h -- the resident correlator: single-block calls without a kernel launch (gfx950 / wave64).
//
// The reference times ONE 1 ms block per call (src/benchmarks.jl:120-146); on this platform a launch plus the wait for its
// end costs 7 us before the kernel has done anything (scripts/probes/sync_probe.hip), a doorbell in pinned host memory that
// a kernel already on the device polls costs 2-3 us there and back (scripts/probes/doorbell_probe.hip).  dc_resident_kernel
// is the fused correlator of gat_dc.h (the same body, gat_dc_body.inc) inside a loop:
//
//   poll the doorbell -> the call's channel records and block offset arrive WITH the ring (one 64-byte line per channel)
//   -> system-scope acquire (the signal may have been rewritten by a copy engine or another kernel since the last call)
//   -> correlate -> results into pinned host memory -> release store of the call's sequence number -> poll again.
//
// Chip tables stay staged in LDS from call to call; the kernel's arguments are read once.
//
// Lifetime: the kernel ends BY ITSELF, whatever the host does -- after `max_calls` calls, after `idle_ticks` without a
// ring, after `life_ticks` in total, or when the host rings kBellQuit.  Only workgroup 0 (the master) polls host memory and
// only it decides to leave; with several workgroups it copies every ring (and its decision to leave) into a doorbell in
// device memory that the others poll, and it leaves only when the call it forwarded last has been finished by all of them
// -- a forwarded ring is never left half served.  The other workgroups leave when told to, or when 1.5 x life_ticks have
// passed (the master can no longer be there).  No workgroup ever waits for another one to make progress: every wait in
// here is a poll with a deadline on the constant 100 MHz clock, so the grid drains even if workgroups never run together.
//
// Which shapes: one block per call, K <= 4 channels, one tap launch, 16-byte aligned block starts with N a multiple of the
// load group (what runs as ONE vector launch otherwise); antennas in tiles of MT <= 4, one tile per workgroup (AW = KT = 1).
#pragma once

#include "gat_dc.h"

namespace gat {

constexpr bool dc_resident_instance(int mt, int l) { return dc_instance(mt, l, 4, 1, 1, 4, 1); }

struct ResidentEnv {
    const unsigned *line; // LDS: the call's doorbell lines
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
    __shared__ unsigned s_bell[kInlineParams * kBellDwords];
    __shared__ unsigned s_ctl[2]; // [0] the call's sequence number or kBellQuit, [1] this workgroup arrived last

    kernarg_prefetch<sizeof(DcArgs) + sizeof(ResidentArgs)>();
    { // padding of the grid (same decode as the body's)
        const unsigned tile0 = ((blockIdx.x >> 3) / (unsigned)a.KG) * 8u + (blockIdx.x & 7u);
        if (tile0 >= (unsigned)a.num_tiles) return;
    }
    const bool master = blockIdx.x == 0;
    const bool alone = a.total_wgs == 1u;
    const int K = a.K;
    int staged_prn[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) staged_prn[kk] = -1;
    unsigned last = r.start_seq, calls = 0, why = kResidentRuns;
    const long long t_start = wall_clock64();
    long long t_last = t_start;
    const ResidentEnv env{s_bell};

    for (;;) {
        // ---- wait for a ring: wave 0 reads all K lines with ONE load (lane i <-> dword i)
        if (threadIdx.x < 64) {
            const int ln = (int)threadIdx.x;
            const unsigned *src = (master ? r.host_bell : r.dev_bell) + ln;
            const bool mine = ln < K * kBellDwords;
            unsigned v = 0, seq = last;
            for (;;) {
                if (mine) v = master ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                     : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                seq = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
                if (seq != last) {
                    if (seq == kBellQuit) { why = kResidentQuit; break; }
                    // every line whole and of this ring: first and last dword = seq, XOR of dwords 0..13 = dword 14
                    unsigned x = (ln & 15) < 14 ? v : 0u;
                    x ^= __shfl_xor(x, 8, 64); x ^= __shfl_xor(x, 4, 64); x ^= __shfl_xor(x, 2, 64); x ^= __shfl_xor(x, 1, 64);
                    const unsigned chk = __shfl(v, (ln & ~15) + 14, 64), head = __shfl(v, ln & ~15, 64), tail = __shfl(v, (ln & ~15) + 15, 64);
                    const bool ok = !mine || (x == chk && head == seq && tail == seq);
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    seq = last; // a line caught half-written (or the host has not reached line 0's siblings yet): read again
                }
                const long long now = wall_clock64();
                if (master) {
                    // the master leaves only between calls: the one it forwarded last is finished by every workgroup
                    unsigned reason = kResidentRuns;
                    if (calls >= r.max_calls) reason = kResidentCalls;
                    else if (now - t_last > r.idle_ticks) reason = kResidentIdle;
                    else if (now - t_start > r.life_ticks) reason = kResidentLife;
                    if (reason != kResidentRuns) {
                        const bool finished = alone || __hip_atomic_load(r.dev_done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == last;
                        if (finished || now - t_start > r.life_ticks + (r.life_ticks >> 2)) { why = reason; seq = kBellQuit; break; }
                    }
                } else if (now - t_start > r.life_ticks + (r.life_ticks >> 1)) { // the master is gone: nobody will ring again
                    why = kResidentLife;
                    seq = kBellQuit;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (master && !alone) { // pass the ring (or the decision to leave) on
                const unsigned fwd = (seq == kBellQuit && ln == 0) ? kBellQuit : v;
                if (mine) __hip_atomic_store(r.dev_bell + ln, fwd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_bell[ln] = v;
            if (ln == 0) s_ctl[0] = seq;
        }
        __syncthreads();
        const unsigned seq = uni(s_ctl[0]);
        if (seq == kBellQuit) break;
        last = seq;
        // the signal of this call was written by somebody else (copy engine, another kernel, the host) before the ring:
        // nothing of it may come from this XCD's caches
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");

        { // the correlator itself: the text dc_kernel is made of
#define GAT_DC_BODY_RESIDENT 1
#include "gat_dc_body.inc"
#undef GAT_DC_BODY_RESIDENT
        }

        // ---- completion: the results (splits == 1: written to host memory by the body; else partial sums in device memory)
        __syncthreads();
        bool last_wg = true;
        if (!alone) {
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); // this workgroup's stores: out of its XCD, on their way to the host
                const unsigned arrived = __hip_atomic_fetch_add(r.done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                s_ctl[1] = arrived == a.total_wgs - 1u ? 1u : 0u;
            }
            __syncthreads();
            last_wg = uni(s_ctl[1]) != 0u;
        }
        if (last_wg) {
            if (a.splits > 1) {
                // second stage: the order of finalize_few_kernel (four interleaved chains, (0+1)+(2+3)): deterministic
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const int elems = a.Ltot * a.M * 2, total = K * elems, splits = a.splits;
                for (int o = (int)threadIdx.x; o < total; o += 256) {
                    const int g = o / elems, e = o - g * elems;
                    const float *p = a.partial + (size_t)g * splits * elems + e;
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    int i = 0;
                    for (; i + 4 <= splits; i += 4) {
                        s0 += p[(size_t)(i + 0) * elems];
                        s1 += p[(size_t)(i + 1) * elems];
                        s2 += p[(size_t)(i + 2) * elems];
                        s3 += p[(size_t)(i + 3) * elems];
                    }
                    if (i < splits) s0 += p[(size_t)i * elems];
                    if (i + 1 < splits) s1 += p[(size_t)(i + 1) * elems];
                    if (i + 2 < splits) s2 += p[(size_t)(i + 2) * elems];
                    ((e & 1) ? r.host_out_im : r.host_out_re)[(size_t)g * (elems / 2) + (e >> 1)] = (s0 + s1) + (s2 + s3);
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                if (!alone) {
                    __hip_atomic_store(r.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(r.dev_done_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(r.host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        ++calls;
        t_last = wall_clock64();
    }
    if (master && threadIdx.x == 0) {
        r.host_state[1] = calls;
        __hip_atomic_store(r.host_state, why, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int FMT, int MT>
static hipError_t launch_dc_resident_m(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s)
{
    auto go = [&](auto l_c) -> hipError_t {
        constexpr int L = decltype(l_c)::value;
        if constexpr (dc_resident_instance(MT, L)) {
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
hipError_t launch_dc_resident_fmt(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s)
{
    if (cfg.vec != 4 || cfg.aw != 1 || cfg.kt != 1 || cfg.nw != 4 || cfg.depth != 1) return hipErrorInvalidValue;
    switch (cfg.ant_tile) {
    case 1: return launch_dc_resident_m<FMT, 1>(a, cfg, r, s);
    case 2: return launch_dc_resident_m<FMT, 2>(a, cfg, r, s);
    case 3: return launch_dc_resident_m<FMT, 3>(a, cfg, r, s);
    case 4: return launch_dc_resident_m<FMT, 4>(a, cfg, r, s);
    default: return hipErrorInvalidValue;
    }
}

} // namespace gat
