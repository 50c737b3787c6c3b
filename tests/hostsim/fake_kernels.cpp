// fake_kernels.cpp -- stand-ins for libgat's kernel launchers (the functions gat_kernels.hip, gat_dc_f*.hip, gat_mfma*.hip and
// gat_resident_f*.hip define): nothing is computed.  Instead every launch of the vector kernel is checked against what the
// kernel ASSUMES about its arguments (LDS carve-up, replica room, grid decode, descriptor spans, tap tables): the planner's
// contract, over thousands of random shapes, under ASan / UBSan on the CPU.  A launch of the resident kernel starts a host
// thread that plays the device's side of the doorbell protocol (gat_resident.h), so that gat_resident_*'s host side --
// ring, wait, second stage, restart after the kernel has left -- runs for real.  Test infrastructure only.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#include "gat_internal.h"
#include "hostsim.h"

namespace gat {

using hostsim::counters;

#define REQUIRE(cond, ...)                                                    \
    do {                                                                      \
        if (!(cond)) {                                                        \
            ++counters.violations;                                            \
            std::fprintf(stderr, "planner invariant broken: %s -- ", #cond);  \
            std::fprintf(stderr, __VA_ARGS__);                                \
            std::fprintf(stderr, "\n");                                       \
        }                                                                     \
    } while (0)

bool dc_has_instance(int ant_tile, int taps, int vec, int aw, int kt, int nw, int depth) { return dc_instance(ant_tile, taps, vec, aw, kt, nw, depth); }
bool dc_has_resident_instance(int ant_tile, int taps, int format)
{
    return format >= GAT_LAYOUT_PLANAR && format <= GAT_LAYOUT_INTERLEAVED_I8 && dc_instance(ant_tile, taps, 4, 1, 1, 4, 1);
}

static void check_dc(const DcArgs &a, const DcLaunch &cfg, bool resident)
{
    const int fmt = cfg.format, vec = cfg.vec, nw = cfg.nw, kt = cfg.kt, aw = cfg.aw, MT = cfg.ant_tile, L = cfg.taps;
    const long long chunk = dc_chunk(vec, fmt, aw, nw);
    const int spv = dc_group_samples(vec, fmt);
    const long long plane_bytes = fmt == GAT_LAYOUT_PLANAR ? 4 : fmt == GAT_LAYOUT_INTERLEAVED ? 8 : fmt == GAT_LAYOUT_INTERLEAVED_I16 ? 4 : 2;
    const char *tag = resident ? "resident" : "launch";
    REQUIRE(dc_instance(MT, L, vec, aw, kt, nw, cfg.depth), "%s: no instance <%d,%d,%d,aw %d,kt %d,nw %d,d %d>", tag, MT, L, vec, aw, kt, nw, cfg.depth);
    REQUIRE(a.M % (MT * aw) == 0 && a.ant_groups == a.M / (MT * aw), "%s: antenna tiling M %d MT %d aw %d groups %d", tag, a.M, MT, aw, a.ant_groups);
    REQUIRE(a.KG == (a.K + kt - 1) / kt, "%s: channel groups %d for K %d kt %d", tag, a.KG, a.K, kt);
    REQUIRE(a.splits >= 1 && a.chunks_per_split >= 1 && (long long)a.splits * a.chunks_per_split >= a.total_chunks &&
                (long long)(a.splits - 1) * a.chunks_per_split < a.total_chunks,
            "%s: splits %d x %d chunks for %d", tag, a.splits, a.chunks_per_split, a.total_chunks);
    const long long head = a.align_head ? 112 / plane_bytes : 0;
    REQUIRE((long long)a.total_chunks * chunk >= a.n_vec + (vec == 4 ? head : 0) && (long long)(a.total_chunks - 1) * chunk < a.n_vec + head + chunk,
            "%s: %d chunks of %lld for n_vec %d (+%lld)", tag, a.total_chunks, chunk, a.n_vec, head);
    REQUIRE(a.blocks_per_wg >= 1 && (a.blocks_per_wg == 1 || a.splits == 1), "%s: blocks per workgroup %d with %d splits", tag, a.blocks_per_wg, a.splits);
    const long long BG = (a.B + a.blocks_per_wg - 1) / a.blocks_per_wg;
    REQUIRE(a.num_tiles == BG * a.ant_groups * a.splits && a.total_wgs == (unsigned)(a.num_tiles * a.KG), "%s: tiles %d wgs %u", tag, a.num_tiles, a.total_wgs);
    REQUIRE(cfg.grid % 8 == 0 && cfg.grid >= a.total_wgs && cfg.grid == (unsigned)(((a.num_tiles + 7) / 8) * 8 * a.KG), "%s: grid %u for %d tiles x %d", tag, cfg.grid, a.num_tiles, a.KG);
    REQUIRE(vec == 4 ? (a.n_vec % spv == 0 && a.n_vec <= a.N && a.N - a.n_vec < spv) : a.n_vec == a.N, "%s: n_vec %d of N %lld", tag, a.n_vec, a.N);
    if (vec == 4) {
        REQUIRE(((long long)(MT - 1) * a.ant_stride + a.N) * plane_bytes < (1ll << 31), "%s: tile span beyond a descriptor", tag);
        REQUIRE((reinterpret_cast<uintptr_t>(a.re) & 15u) == 0, "%s: vector loads from an unaligned plane", tag);
        REQUIRE(a.M == 1 || a.ant_stride % spv == 0, "%s: antenna stride %lld", tag, a.ant_stride);
    } else {
        REQUIRE(MT == 1 && aw == 1 && kt == 1, "%s: scalar loads with a tile", tag);
    }
    // taps
    REQUIRE(L >= 1 && L <= kMaxTapsPerLaunch, "%s: %d taps", tag, L);
    for (int l = 1; l < L; ++l) REQUIRE(a.shifts[l] >= a.shifts[l - 1], "%s: taps not ascending", tag);
    REQUIRE(a.rep_span == a.shifts[L - 1] - a.shifts[0] && a.rep_span <= kMaxLaunchSpan, "%s: tap span %d", tag, a.rep_span);
    for (int l = 0; l < L; ++l) {
        const int d = a.shifts[l] - a.shifts[0];
        REQUIRE(a.tap_index[l] >= 0 && a.tap_index[l] < a.Ltot, "%s: tap index %d of %d", tag, a.tap_index[l], a.Ltot);
        REQUIRE(a.tap_off[l] % 2 == 0 && a.tap_off[l] == ((d & 1) ? a.rep_copy_stride + d - 1 : d), "%s: tap offset %d for distance %d", tag, a.tap_off[l], d);
        REQUIRE(!(d & 1) || a.rep_copy_stride > 0, "%s: odd tap distance without the shifted copy", tag);
    }
    // LDS: what the kernel carves must fit what the launch gives it; the replica must hold a segment + taps + the producers' overshoot
    const int threads = 64 * nw, rpc = threads / kt;
    REQUIRE(a.seg_steps >= 1 && a.seg_steps <= kUcarSteps && a.seg_steps % cfg.depth == 0, "%s: %d steps per segment (depth %d)", tag, a.seg_steps, cfg.depth);
    const size_t carve = nw == 1 ? dc_lds_bytes_one_wave(a.rep_chan_floats, a.table_stride) : dc_lds_bytes_floats(kt, a.table_stride, a.rep_chan_floats);
    REQUIRE(carve <= cfg.lds_bytes && cfg.lds_bytes <= 160 * 1024, "%s: LDS carve %zu of %u", tag, carve, cfg.lds_bytes);
    const long long seg_entries = (long long)a.seg_steps * chunk + a.rep_span;
    const long long filled = (seg_entries + rpc - 1) / rpc * rpc; // every producer takes the same number of steps
    if (a.rep_copy_stride) {
        REQUIRE(a.rep_copy_stride >= filled + 1 - 1 && (long long)a.rep_copy_stride - 1 + filled <= a.rep_chan_floats,
                "%s: replica + copy: stride %d, %lld entries, room %d", tag, a.rep_copy_stride, filled, a.rep_chan_floats);
    } else {
        REQUIRE(filled <= a.rep_chan_floats, "%s: replica: %lld entries, room %d", tag, filled, a.rep_chan_floats);
    }
    REQUIRE((long long)a.N + a.max_abs_shift < (1ll << 30), "%s: sample range", tag);
    REQUIRE(a.code_row_stride % 16 == 0 && a.code_row_stride >= a.Lc && a.codes != nullptr, "%s: chip table", tag);
    REQUIRE(a.table_stride % 16 == 0 && (a.code_bits ? a.table_stride * 8 >= a.Lc : a.table_stride == a.code_row_stride), "%s: staged table of %d bytes (%s) for %d chips",
            tag, a.table_stride, a.code_bits ? "sign bits" : "int8", a.Lc);
    if (!resident) {
        REQUIRE(a.params != nullptr || (long long)a.B * a.K <= kInlineParams, "launch: %d x %d records without a buffer", a.B, a.K);
        REQUIRE(a.out_re != nullptr && a.out_im != nullptr, "launch: no outputs");
        REQUIRE(a.splits == 1 || (a.flags & GAT_FLAG_ATOMIC) || a.partial != nullptr, "launch: split without a partial buffer");
        REQUIRE((a.done_counter == nullptr) == (a.host_flag == nullptr) && (a.done_counter == nullptr || a.flag_seq != 0), "launch: completion flag half set");
        REQUIRE(cfg.depth == 1 || (a.splits == 1 && a.KG == 1 && !a.keep_l2), "launch: two sample sets outside the streaming regime");
    } else {
        REQUIRE(a.B == 1 && kt == 1 && aw == 1 && nw == 4 && cfg.depth == 1 && vec == 4 && a.n_vec == a.N && a.K <= kResMaxChannels, "resident: geometry");
    }
}

hipError_t launch_dc(const DcArgs &a, const DcLaunch &cfg, hipStream_t)
{
    ++counters.dc_launches;
    check_dc(a, cfg, false);
    return hipSuccess;
}
hipError_t launch_finalize(const float *partial, float *out_re, float *out_im, int splits, int elems, long long groups, hipStream_t, unsigned *done, unsigned *flag, unsigned seq)
{
    ++counters.finalize_launches;
    REQUIRE(partial && out_re && out_im && splits >= 1 && elems >= 2 && elems % 2 == 0 && groups >= 1, "finalize: %d splits x %d elems x %lld groups", splits, elems, groups);
    REQUIRE((done == nullptr) == (seq == 0) || flag != nullptr, "finalize: completion flag half set");
    return hipSuccess;
}
hipError_t launch_dc_tail(const DcTailArgs &a, hipStream_t)
{
    ++counters.tail_launches;
    REQUIRE(a.n_vec < a.N && a.N - a.n_vec < 8 && a.out_re && a.out_im, "tail: n_vec %d of %lld", a.n_vec, a.N);
    return hipSuccess;
}
// matrix-core kernels: the planner's helpers are the kernels' own (gat_internal.h); every launch is checked against what
// the kernels assume
size_t mfma_lds_bytes(int nct, int ct, int rep_stride, int code_row_stride, int codes_in_lds) { return mf_lds_bytes(nct, ct, rep_stride, code_row_stride, codes_in_lds); }
size_t mfma_bf16_lds_bytes(int rt, int nct, int fmt, int nslots, int rep_stride, int code_bits_stride, int mode) { return mb_lds_bytes(rt, nct, fmt, nslots, rep_stride, code_bits_stride, mode); }
int mfma_bf16_mode(int rt, int nct, int fmt, bool force_three) { return mb_mode(rt, nct, fmt, force_three); }
int mfma_bf16_slots(int nct, int L, int K) { return mb_slots(nct, L, K); }
int mfma_bf16_max_slots() { return kMbMaxSlots; }
int mfma_bf16_tile_samples(int rt, int nct) { return mb_tile_samples(rt, nct); }
int mfma_bf16_max_chain() { return kMbMaxChain; }
int mfma_bf16_threads(int rt, int nct) { return mb_threads(rt, nct); }
int mfma_bf16_producer_threads(int rt, int nct) { return mb_threads(rt, nct) - 64 * mb_consumer_waves(rt, nct); }

static void check_mf_common(const MfArgs &a, unsigned grid, int T, const char *tag)
{
    REQUIRE(a.re && a.params && a.codes && a.out_re && a.out_im, "%s: null pointer", tag);
    REQUIRE(a.total_steps == (a.N + T - 1) / T && a.steps_per_split >= 1 && (long long)a.splits * a.steps_per_split >= a.total_steps &&
                (long long)(a.splits - 1) * a.steps_per_split < a.total_steps,
            "%s: %d splits x %d steps for %d", tag, a.splits, a.steps_per_split, a.total_steps);
    REQUIRE(a.num_tiles == a.B * a.ant_tiles * a.splits && grid == (unsigned)(((a.num_tiles + 7) / 8) * 8 * a.chan_groups), "%s: tiles %d grid %u", tag, a.num_tiles, grid);
    REQUIRE(a.splits == 1 || (a.flags & GAT_FLAG_ATOMIC) || a.partial != nullptr, "%s: split without a partial buffer", tag);
    REQUIRE(a.L >= 1 && a.L <= kMfmaMaxTaps && a.rep_span == a.shifts[a.L - 1] - a.shifts[0] && a.rep_span <= kMfmaMaxSpan, "%s: taps %d span %d", tag, a.L, a.rep_span);
    for (int l = 1; l < a.L; ++l) REQUIRE(a.shifts[l] >= a.shifts[l - 1], "%s: taps not ascending", tag);
    for (int l = 0; l < a.L; ++l) REQUIRE(a.tap_index[l] >= 0 && a.tap_index[l] < a.L, "%s: tap index", tag);
    REQUIRE(a.rep_stride >= T + a.rep_span && (a.rep_stride & 1), "%s: replica row %d for a tile of %d + %d", tag, a.rep_stride, T, a.rep_span);
    REQUIRE((reinterpret_cast<uintptr_t>(a.re) & 15u) == 0 && a.ant_stride % 2 == 0, "%s: alignment", tag);
}

hipError_t launch_mfma(const MfArgs &a, int nct, unsigned grid, unsigned lds_bytes, hipStream_t)
{
    ++counters.mfma_launches;
    REQUIRE((nct == 1 || nct == 2 || nct == 4) && a.CT >= 1 && a.CT == 16 / a.L && nct * a.CT <= 20, "f32 mfma: nct %d CT %d L %d", nct, a.CT, a.L);
    REQUIRE(a.M % 16 == 0 && a.ant_tiles == a.M / 16 && a.im != nullptr, "f32 mfma: antennas %d tiles %d", a.M, a.ant_tiles);
    REQUIRE(a.chan_groups == ((a.K + a.CT - 1) / a.CT + nct - 1) / nct, "f32 mfma: channel groups %d", a.chan_groups);
    REQUIRE(lds_bytes == mf_lds_bytes(nct, a.CT, a.rep_stride, a.code_row_stride, a.codes_in_lds) && lds_bytes <= 160 * 1024, "f32 mfma: LDS %u", lds_bytes);
    check_mf_common(a, grid, kMfTile, "f32 mfma");
    return hipSuccess;
}
hipError_t launch_mfma_bf16(const MfArgs &a, int rt, int nct, int fmt, unsigned grid, unsigned lds_bytes, hipStream_t)
{
    ++counters.mfma_launches;
    const bool inst = (rt == 1 || rt == 2 || rt == 4) && (nct == 1 || nct == 2 || nct == 4) && !(rt == 4 && nct == 1);
    REQUIRE(inst && fmt >= 0 && fmt <= 3, "bf16 mfma: instance rt %d nct %d fmt %d", rt, nct, fmt);
    const int T = mb_tile_samples(rt, nct), spv = dc_group_samples(4, fmt);
    REQUIRE(a.M % (16 * rt) == 0 && a.ant_tiles == a.M / (16 * rt), "bf16 mfma: antennas %d rt %d tiles %d", a.M, rt, a.ant_tiles);
    REQUIRE(a.N % spv == 0, "bf16 mfma: N %lld is no multiple of the load group", a.N);
    const int tiles = (2 * a.L * a.K + 31) / 32;
    REQUIRE(a.chan_groups == (tiles + nct - 1) / nct, "bf16 mfma: %d column tiles in %d groups of %d", tiles, a.chan_groups, nct);
    REQUIRE(a.nslots == mb_slots(nct, a.L, a.K) && a.nslots <= kMbMaxSlots && a.nslots * T / 2 <= mb_threads(rt, nct) - 64 * mb_consumer_waves(rt, nct),
            "bf16 mfma: %d slots x %d samples for %d producers", a.nslots, T, mb_threads(rt, nct) - 64 * mb_consumer_waves(rt, nct));
    REQUIRE((long long)a.steps_per_split * T <= kMbMaxChain || a.steps_per_split == 1, "bf16 mfma: chain of %d x %d samples", a.steps_per_split, T);
    REQUIRE(a.code_bits && a.zeros && a.code_bits_stride % 4 == 0 && a.code_bits_stride * 32 >= a.Lc, "bf16 mfma: sign-bit tables");
    REQUIRE(a.mb_mode == mb_mode(rt, nct, fmt) || (fmt == GAT_LAYOUT_INTERLEAVED_I16 && a.mb_mode == kMbThree), "bf16 mfma: operand split %d for layout %d", a.mb_mode, fmt);
    if (mb_rep_ring_rows(rt))
        REQUIRE(a.rep_ring % T == 0 && a.rep_ring >= a.rep_span + 2 * T && a.rep_stride >= a.rep_ring + a.rep_span + T, "bf16 mfma: chip-sign ring %d in rows of %d for span %d, tile %d", a.rep_ring, a.rep_stride, a.rep_span, T);
    else
        REQUIRE(a.rep_ring == 0 && a.rep_stride >= a.rep_span + T, "bf16 mfma: chip-sign rows of %d for span %d, tile %d", a.rep_stride, a.rep_span, T);
    REQUIRE(a.mb_mode != kMbTwo || a.nslots * T / 4 <= mb_threads(rt, nct) - 64 * mb_consumer_waves(rt, nct), "bf16 mfma: two-term items");
    REQUIRE(lds_bytes == mb_lds_bytes(rt, nct, fmt, a.nslots, a.rep_stride, a.code_bits_stride, a.mb_mode) && lds_bytes <= 160 * 1024, "bf16 mfma: LDS %u", lds_bytes);
    check_mf_common(a, grid, T, "bf16 mfma");
    return hipSuccess;
}

hipError_t launch_gen_code_replica(float *, long long, const int8_t *, int, double, double, double, long long, bool, hipStream_t) { ++counters.other_launches; return hipSuccess; }
hipError_t launch_gen_code_replica_texaddr(float *, long long, const int8_t *, int, double, double, double, long long, int coord_bits, int texel_bits, hipStream_t)
{
    ++counters.other_launches;
    REQUIRE(coord_bits >= 0 && coord_bits <= 32 && texel_bits >= -1 && texel_bits <= 24, "texture addressing study: %d / %d bits", coord_bits, texel_bits);
    return hipSuccess;
}
hipError_t launch_read_stream(const void *dev, size_t bytes, int variant, int num_cus, float *sink, hipStream_t)
{
    ++counters.other_launches;
    REQUIRE(dev && sink && bytes >= 16 && bytes % 16 == 0 && variant >= 0 && variant < 16 && num_cus > 0, "read stream: %zu bytes, variant %d", bytes, variant);
    return hipSuccess;
}
hipError_t launch_gen_code_replica_multi(float *, long long, long long, int, const gat_channel_params *, const int8_t *, int, int, int, double, long long, hipStream_t) { ++counters.other_launches; return hipSuccess; }
hipError_t launch_accumulate_debug(const float *, const float *, long long, int, long long, const gat_channel_params &, const int8_t *, int, double, int, const int *, float *, float *, float *, float *, float *, float *, hipStream_t) { ++counters.other_launches; return hipSuccess; }
hipError_t launch_gen_signal(void *, void *, int, long long, int, long long, long long, int, int, const gat_channel_params *, const int8_t *, int, int, int, double, float, const float *, float, unsigned long long, hipStream_t) { ++counters.other_launches; return hipSuccess; }
hipError_t launch_reduce_stage1(const float *, const float *, long long, int, int, float *, hipStream_t) { ++counters.other_launches; return hipSuccess; }
hipError_t launch_tracking_update(const float *, const float *, int, int, const gat_loop_config &, gat_loop_state *, const gat_channel_params *cur, gat_channel_params *next, hipStream_t)
{
    ++counters.other_launches;
    REQUIRE(cur != nullptr && next != nullptr && cur != next, "tracking update: parameter ping-pong");
    return hipSuccess;
}

// ---- the resident kernel, played by a host thread -------------------------------------------------------------------------
} // namespace gat
namespace hostsim {
float resident_value(unsigned slot, int o) { return (float)(slot * 64u + (unsigned)o) + 0.5f; }
} // namespace hostsim
namespace gat {

static void resident_device(DcArgs a, DcLaunch cfg, ResidentArgs r)
{
    using clk = std::chrono::steady_clock;
    const int nval = 2 * cfg.ant_tile * cfg.taps, lw = (nval + kResLinePayload - 1) / kResLinePayload;
    auto ld = [](const unsigned *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); };
    unsigned last = r.start_seq, calls = 0, why = kResidentRuns;
    const auto t_start = clk::now();
    auto t_last = t_start;
    auto ticks = [](clk::duration d) { return std::chrono::duration_cast<std::chrono::nanoseconds>(d).count() / 10; }; // 100 MHz
    unsigned poll = 0;
    for (;;) {
        const unsigned *bell = r.host_bell + (size_t)(poll++ % (unsigned)r.bell_copies) * kResMaxChannels * kBellDwords; // (a copy per workgroup: all must ring)
        const unsigned seq = ld(bell);
        if (seq == kBellQuit) { why = kResidentQuit; break; }
        if (seq != last) {
            bool ok = true;
            for (int k = 0; k < a.K && ok; ++k) {
                const unsigned *ln = bell + k * kBellDwords;
                unsigned x = 0, w[kBellDwords];
                for (int i = 0; i < kBellDwords; ++i) w[i] = ld(ln + i);
                for (int i = 0; i < 14; ++i) x ^= w[i];
                ok = w[0] == seq && w[15] == seq && w[14] == x;
                if (ok) { // the record the host rang: what the kernel would read
                    gat_channel_params p;
                    std::memcpy(&p, &w[2], sizeof p);
                    long long off;
                    std::memcpy(&off, &w[12], sizeof off);
                    if (!(p.prn >= 0 && p.prn < a.num_prns && off >= 0 && p.reserved == 0)) {
                        ++counters.violations;
                        std::fprintf(stderr, "resident: bad record in a validated ring (prn %d, offset %lld)\n", p.prn, off);
                    }
                }
            }
            if (ok) {
                // every working workgroup posts its lines (here: one after the other, last line of the last slot last)
                for (unsigned slot = 0; slot < a.total_wgs; ++slot)
                    for (int j = 0; j < lw; ++j) {
                        unsigned *ln = r.host_lines + ((size_t)slot * lw + j) * 16, x = seq;
                        for (int i = 0; i < kResLinePayload; ++i) {
                            const int o = j * kResLinePayload + i;
                            const float v = o < nval ? hostsim::resident_value(slot, o) : 0.f;
                            unsigned u;
                            std::memcpy(&u, &v, 4);
                            x ^= u;
                            __atomic_store_n(ln + i, u, __ATOMIC_RELAXED);
                        }
                        __atomic_store_n(ln + 14, x, __ATOMIC_RELAXED);
                        __atomic_store_n(ln + 15, seq, __ATOMIC_RELEASE);
                    }
                last = seq;
                ++calls;
                ++counters.resident_calls;
                t_last = clk::now();
                continue;
            }
        }
        const auto now = clk::now();
        if (calls >= r.max_calls) { why = kResidentCalls; break; }
        if (ticks(now - t_last) > r.idle_ticks) { why = kResidentIdle; break; }
        if (ticks(now - t_start) > r.life_ticks) { why = kResidentLife; break; }
    }
    r.host_state[1] = calls;
    __atomic_store_n(r.host_state, why, __ATOMIC_RELEASE);
}

// what one "compute unit" of the simulated device holds of a resident instance: two workgroups by the occupancy API's word
// (the library takes one off every answer above one)
hipError_t dc_resident_blocks_per_cu(const DcLaunch &cfg, int *blocks_per_cu)
{
    REQUIRE(blocks_per_cu != nullptr && dc_has_resident_instance(cfg.ant_tile, cfg.taps, cfg.format), "resident occupancy query: instance");
    *blocks_per_cu = 2;
    return hipSuccess;
}

hipError_t launch_dc_resident(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s)
{
    ++counters.resident_starts;
    check_dc(a, cfg, true);
    REQUIRE((r.bell_copies == 1 || r.bell_copies == 8) && (r.bell_copies == 1 || r.forward == 0), "resident: %d doorbell copies, forward %d", r.bell_copies, r.forward);
    REQUIRE(r.host_bell && r.host_lines && r.host_state && r.dev_quit && r.dev_bell && r.idle_ticks > 0 && r.life_ticks > 0 && r.max_calls > 0, "resident: arguments");
    REQUIRE(s != nullptr, "resident: launched on the default stream");
    hostsim::attach_worker(s, std::thread(resident_device, a, cfg, r));
    return hipSuccess;
}

} // namespace gat
