// gat_api.cpp -- the C ABI declared in include/gat.h: context, validation, launch planning.
//
// Launch planning for the fused correlator (DESIGN.md "Kernels"):
//   ant_tile MT = largest of {4,3,2,1} dividing M          (register accumulators 2*MT*L <= 64)
//   vec      = 4 when every plane base/stride is 16-byte aligned, else 1
//   aw, kt   = antenna tiles (waves) and channels per workgroup: 16 antennas share one replica, up to 4 channels loop
//              over register-resident samples
//   nw       = waves per workgroup: 4, or 1 for short blocks of 1-2 antenna tiles in a long stream
//   splits   = workgroups per (block, channel, antenna tile): 1 once B*K*M/MT already fills the
//              chip (>= 8 workgroups per CU), otherwise the block's samples are split and a
//              finalize launch sums the per-split partials in fixed order.
//   matrix-core kernels where they measured faster (auto rule below).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <dlfcn.h>
#include <new>
#include <string>
#include <vector>

#include "gat_ctx.h"

using namespace gat;

namespace {

// Tracing ranges around the library's launch sequences (the reference wraps every launch of kernel_algorithm in
// NVTX.@range, src/algorithms.jl:953, :973 ... :1526): roctxRangePush / Pop from librocprofiler-sdk-roctx, resolved at
// the first use so that the library has no link-time dependency on the profiler SDK; no-ops when it is not installed.
// Ranges show up in rocprofv3 --marker-trace.  GAT_ROCTX=0 disables the lookup.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = std::getenv("GAT_ROCTX");
        if (e && e[0] == '0') return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
struct TraceRange { // RAII: one range per launch sequence of an entry point
    explicit TraceRange(const char *name)
    {
        static const Roctx r; // resolved once, thread-safe
        rx = &r;
        if (rx->push) (void)rx->push(name);
    }
    ~TraceRange()
    {
        if (rx->pop) (void)rx->pop();
    }
    const Roctx *rx;
};

constexpr size_t kMaxLoopGraphs = 4;

// Every instantiated graph bakes in device pointers of the library's own buffers (split partials, code tables) and the
// launch geometry: drop them all whenever one of those can change (scratch reallocation, gat_set_codes, gat_set_stream,
// kernel-selection knobs).
void drop_loop_graphs(gat_ctx *c)
{
    // gat_tracking_run(GAT_FLAG_GRAPH) is asynchronous: a recorded graph may still be executing on the stream
    if (!c->loop_graphs.empty()) (void)hipStreamSynchronize(c->stream);
    for (auto &g : c->loop_graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    c->loop_graphs.clear();
}


int32_t ensure_partial(gat_ctx *c, size_t bytes)
{
    if (bytes <= c->partial_bytes) return GAT_OK;
    drop_loop_graphs(c); // recorded launches point at the old buffer
    if (c->d_partial) {
        GAT_HIP(c, hipStreamSynchronize(c->stream)); // previous launches may still read it
        park_residents(c);
        GAT_HIP(c, hipFree(c->d_partial));
        c->d_partial = nullptr;
        c->partial_bytes = 0;
    }
    GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_partial), bytes));
    c->partial_bytes = bytes;
    return GAT_OK;
}

// the parameter records of a host call reach the device: into the context's buffer, on the context's stream
int32_t upload_params(gat_ctx *c, const gat_channel_params *params_host, size_t n)
{
    if (n > c->params_cap) {
        if (c->d_params) {
            GAT_HIP(c, hipStreamSynchronize(c->stream));
            park_residents(c);
            GAT_HIP(c, hipFree(c->d_params));
            c->d_params = nullptr;
            c->params_cap = 0;
        }
        GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_params), n * sizeof(gat_channel_params)));
        c->params_cap = n;
    }
    GAT_HIP(c, hipMemcpyAsync(c->d_params, params_host, n * sizeof(gat_channel_params), hipMemcpyHostToDevice, c->stream));
    return GAT_OK;
}


} // namespace

// (declared in gat_ctx.h: the resident correlator's host side asks it for a launch plan)
int32_t gat::correlate_impl(gat_ctx *c, const gat_signal_desc *sig, const gat_channel_params *params_dev,
                            int32_t B, int32_t K, int32_t L, const int32_t *shifts, double fs,
                            float *out_re, float *out_im, uint32_t flags, const gat_channel_params *params_inline,
                            DcPlan *plan_out)
{
    c->wait_seq = 0;
    const TraceRange trace("gat_downconvert_and_correlate");
    if (!sig || (!params_dev && !params_inline) || !shifts || !out_re || !out_im) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    const int fmt = sig->layout;
    if (fmt < GAT_LAYOUT_PLANAR || fmt > GAT_LAYOUT_INTERLEAVED_I8) return fail(c, GAT_ERR_ARG, "unknown signal layout");
    const bool planar = fmt == GAT_LAYOUT_PLANAR;
    if (!sig->re || (planar && !sig->im) || (!planar && sig->im))
        return fail(c, GAT_ERR_ARG, "signal pointers do not match the layout");
    if (B < 1 || K < 1 || sig->num_ants < 1 || sig->num_samples < 1)
        return fail(c, GAT_ERR_ARG, "sizes must be positive");
    if (L < 1 || L > GAT_MAX_TAPS) return fail(c, GAT_ERR_RANGE, "num_taps outside 1..GAT_MAX_TAPS");
    if (!(fs > 0.0) || !std::isfinite(fs)) return fail(c, GAT_ERR_ARG, "sampling frequency must be positive");
    if (flags & ~GAT_FLAG_ATOMIC) return fail(c, GAT_ERR_ARG, "unknown flag bits");
    long long max_shift = 0;
    for (int l = 0; l < L; ++l) max_shift = std::max<long long>(max_shift, std::llabs((long long)shifts[l]));
    if (sig->num_samples + max_shift >= (1ll << 30))
        return fail(c, GAT_ERR_RANGE, "num_samples + |shift| must stay below 2^30");
    if (sig->ant_stride < 0 || sig->block_stride < 0 || sig->chan_stride < 0)
        return fail(c, GAT_ERR_ARG, "negative stride");

    const int M = sig->num_ants;
    int MT = 1;
    for (int mt = c->max_ant_tile; mt >= 1; --mt)
        if (M % mt == 0) {
            MT = mt;
            break;
        }
    // 16-byte vector loads need every group start 16-byte aligned: plane bases and all strides
    // multiples of the samples one 16-byte load holds (4 / 2 / 4 / 8 by format)
    const int spv = dc_group_samples(4, fmt);
    const long long plane_bytes = fmt == GAT_LAYOUT_PLANAR ? 4 : fmt == GAT_LAYOUT_INTERLEAVED ? 8 : fmt == GAT_LAYOUT_INTERLEAVED_I16 ? 4 : 2;
    int vec = 1;
    // ... in a block that a 32-bit descriptor length can describe.  The block LENGTH may be anything (the reference
    // bounds each thread by num_samples, src/algorithms.jl:170): lanes beyond the last whole group read zeros through
    // the buffer range check and the N % spv samples behind it are taken one per lane after the step loop.
    // (a stride that is never applied -- one antenna, one block -- does not matter)
    if (aligned16(sig->re) && (!planar || aligned16(sig->im)) && (sig->num_ants == 1 || sig->ant_stride % spv == 0) &&
        (B == 1 || sig->block_stride % spv == 0) && sig->chan_stride % spv == 0 &&
        sig->num_samples * plane_bytes < (1ll << 31))
        vec = 4;

    const long long N = sig->num_samples;
    if (vec != 4) MT = 1; // unaligned input (scalar loads) is served one antenna per wave
    // the vector kernel reaches a wave's MT antennas through ONE descriptor per plane (antenna = scalar offset): the
    // tile's span of bytes must stay below 2^31 (a lane offset of 2^31 then means "beyond every record")
    while (MT > 1 && ((long long)(MT - 1) * sig->ant_stride + N) * plane_bytes >= (1ll << 31)) {
        int next = 1;
        for (int mt = MT - 1; mt >= 1; --mt)
            if (M % mt == 0) { next = mt; break; }
        MT = next;
    }

    // ---- matrix-core paths: antenna-rich shapes whose (channel, tap) columns fill a useful part of
    // a 32-column tile run on the matrix cores -- the split-bf16 kernel (gat_mfma_bf16.hip) by default,
    // the f32-MFMA kernel (gat_mfma.hip) on request; everything else takes the vector kernel below.
    {
        int order[GAT_MAX_TAPS];
        for (int l = 0; l < L; ++l) order[l] = l;
        std::stable_sort(order, order + L, [&](int x, int y) { return shifts[x] < shifts[y]; });
        const long long span = (long long)shifts[order[L - 1]] - shifts[order[0]];
        const int CT = L <= kMfmaMaxTaps ? 16 / L : 0;
        const bool shape_any = !plan_out && c->mc_mode != 0 && vec == 4 && N % spv == 0 /* whole load groups */ && M % 16 == 0 && sig->chan_stride == 0 && CT >= 1 &&
                              span <= kMfmaMaxSpan && 2 * std::min(K, CT) * L >= 12 /* >= 3/8 of the columns */;
        const bool shape_ok = shape_any && planar; // the f32-MFMA kernel reads planar f32 only
        const int nct_total = shape_ok ? (K + CT - 1) / CT : 1;
        int nct = nct_total >= 4 ? 4 : (nct_total >= 2 ? 2 : 1);
        while (nct > 1 && nct * CT > 20) nct >>= 1; // both kernels keep at most 20 channel slots per workgroup
        // kernel choice: 3 / 1 (auto) -> split-bf16 when its tile fits in LDS, 2 -> f32 MFMA.  GAT_MC_AUTO picks a matrix
        // kernel only where it measured faster than the vector kernel (scripts/history/r02/r02_planner_scan.sh, profiles/r02/
        // r02h_planner_scan.txt; N = 50 000, 3 taps): with float samples the round-2 vector kernel (16 antennas per
        // workgroup, channel loop over register-resident samples, lean step loop) wins or ties up to 24 channels at 64
        // antennas, 32 at 32 and 16 at 128, so the split-bf16 kernel takes M >= 32 with K >= 32 and M * K >= 2048
        // (64 x 32: 0.53 vs 0.58 ms, 32 x 64: 0.84 vs 0.99, 64 x 64: 0.75 vs 1.03); from int8 pairs (single-term
        // path, half the MFMAs) it wins from 24 (channel, tap, re/im) columns on at every M.  The f32-MFMA kernel is
        // never chosen by itself any more: the round-2 vector kernel is faster everywhere (configs[4]: 2.45 vs 4.19 ms,
        // 64 antennas x 32 channels: 0.58 vs 0.92 ms); it runs on request (GAT_MC_F32).
        const bool int8_in = fmt == GAT_LAYOUT_INTERLEAVED_I8;
        const bool auto_bf16 = 2ll * L * K >= 24 && (int8_in || (M >= 32 && K >= 32 && (long long)M * K >= 2048));
        const bool auto_f32 = false;
        const bool want_bf16 = c->mc_mode == 3 || (c->mc_mode == 1 && auto_bf16);
        const bool want_f32 = c->mc_mode == 2 || (c->mc_mode == 1 && auto_f32);
        int kind = 0, rt = 1, rep_stride_m = 0;
        int nslots_b = 0, tiles_b = 0, nct_b = 1;
        if (shape_any && want_bf16 && c->d_code_bits && c->d_zeros && N % spv == 0 && spv <= 8) {
            // split-bf16 kernel: columns packed flat (2 L per channel), 32 per tile
            tiles_b = (2 * L * K + 31) / 32;
            const int rt_max = (M / 16) % 4 == 0 ? 4 : ((M / 16) % 2 == 0 ? 2 : 1);
            for (int n = tiles_b >= 4 ? 4 : (tiles_b >= 2 ? 2 : 1); n >= 1 && !kind; n >>= 1) {
                rt = (rt_max == 4 && n == 1) ? 2 : rt_max; // <4,1> does not fit the VGPR budget of a 12-wave workgroup
                const int T = mfma_bf16_tile_samples(rt, n);
                const int rs = ((T + (int)span + 31) / 32) * 32 + 1; // odd: channel rows land on different banks
                const int ns = mfma_bf16_slots(n, L, K);
                // at most one (slot, sample pair) item per producer thread
                if (ns * T / 2 > mfma_bf16_producer_threads(rt, n) || ns > mfma_bf16_max_slots()) continue;
                if (mfma_bf16_lds_bytes(rt, n, fmt, ns, rs, c->code_bits_stride) <= 160 * 1024) {
                    kind = 2;
                    nct_b = n;
                    nslots_b = ns;
                    rep_stride_m = rs;
                }
            }
        }
        if (shape_ok && !kind && want_f32) {
            rep_stride_m = ((256 + (int)span + 31) / 32) * 32 + 1;
            if (mfma_lds_bytes(nct, CT, rep_stride_m, c->code_row_stride, 0) <= 160 * 1024) kind = 1;
        }
        if (kind) {
            if (kind == 2) nct = nct_b;
            const int T = kind == 2 ? mfma_bf16_tile_samples(rt, nct) : 256;
            if (!params_dev) {
                const int32_t rc = upload_params(c, params_inline, (size_t)B * K);
                if (rc != GAT_OK) return rc;
                params_dev = c->d_params;
            }
            MfArgs m{};
            m.re = sig->re;
            m.im = sig->im;
            m.params = params_dev;
            m.codes = c->d_codes;
            m.out_re = out_re;
            m.out_im = out_im;
            m.N = N;
            m.ant_stride = sig->ant_stride;
            m.block_stride = sig->block_stride;
            m.fs = fs;
            m.M = M; m.K = K; m.B = B; m.L = L; m.Lc = c->Lc; m.num_prns = c->P; m.code_row_stride = c->code_row_stride;
            m.CT = CT;
            m.chan_groups = kind == 2 ? (tiles_b + nct - 1) / nct : (nct_total + nct - 1) / nct;
            m.nslots = nslots_b;
            m.ant_tiles = kind == 2 ? M / (16 * rt) : M / 16;
            m.total_steps = (int)((N + T - 1) / T);
            const long long groups_m = (long long)B * m.ant_tiles * m.chan_groups;
            // the split-bf16 kernel runs one 8-wave workgroup per CU (its LDS tile): 2 rounds fill the chip
            const long long want = (kind == 2 ? 2ll : 4ll) * c->num_cus;
            long long sp = std::max<long long>(1, (want + groups_m - 1) / groups_m);
            sp = std::min<long long>(sp, m.total_steps);
            m.steps_per_split = (int)((m.total_steps + sp - 1) / sp);
            if (kind == 2) m.steps_per_split = std::min(m.steps_per_split, std::max(1, mfma_bf16_max_chain() / T));
            m.splits = (m.total_steps + m.steps_per_split - 1) / m.steps_per_split;
            if (kind == 2 && m.splits > 1) {
                // one workgroup per CU at a time: prefer a split count whose workgroups fill whole rounds of the chip
                // (735 workgroups on 256 CUs idle 13 % of the third round; 768 do not)
                int best = m.splits;
                double best_eff = 0.0;
                for (int sp2 = m.splits; sp2 <= std::min<long long>(m.total_steps, (long long)m.splits + m.splits / 4 + 8); ++sp2) {
                    const int sps = (m.total_steps + sp2 - 1) / sp2;
                    const int real = (m.total_steps + sps - 1) / sps; // split count that step size really gives
                    const long long wgs = groups_m * real;
                    const long long rounds = (wgs + c->num_cus - 1) / c->num_cus;
                    const double eff = (double)wgs / (double)(rounds * c->num_cus);
                    if (eff > best_eff + 1e-9) {
                        best_eff = eff;
                        best = real;
                    }
                }
                m.steps_per_split = (m.total_steps + best - 1) / best;
                m.splits = (m.total_steps + m.steps_per_split - 1) / m.steps_per_split;
            }
            m.num_tiles = B * m.ant_tiles * m.splits;
            m.max_abs_shift = (int)max_shift;
            m.rep_span = (int)span;
            m.rep_stride = rep_stride_m;
            m.flags = flags;
            for (int l = 0; l < kMfmaMaxTaps; ++l) {
                m.shifts[l] = shifts[order[std::min(l, L - 1)]];
                m.tap_index[l] = order[std::min(l, L - 1)];
            }
            const long long grid_m = ((long long)(m.num_tiles + 7) / 8) * 8 * m.chan_groups;
            if (grid_m >= (1ll << 31)) return fail(c, GAT_ERR_RANGE, "grid too large");
            const bool atomic_m = (flags & GAT_FLAG_ATOMIC) != 0;
            const size_t out_elems_m = (size_t)B * K * L * M;
            if (atomic_m) {
                GAT_HIP(c, hipMemsetAsync(out_re, 0, out_elems_m * sizeof(float), c->stream));
                GAT_HIP(c, hipMemsetAsync(out_im, 0, out_elems_m * sizeof(float), c->stream));
            } else if (m.splits > 1) {
                const int32_t rc = ensure_partial(c, (size_t)B * K * m.splits * L * M * 2 * sizeof(float));
                if (rc != GAT_OK) return rc;
            }
            m.partial = c->d_partial;
#ifdef GAT_MFMA_STAMPS
            {
                static unsigned long long *dbg = nullptr;
                if (!dbg) hipMalloc(reinterpret_cast<void **>(&dbg), 8u << 20);
                m.dbg = dbg;
                c->dbg_ptr = dbg;
            }
#endif
            unsigned lds;
            if (kind == 2) {
                m.codes_in_lds = 1; // sign-bit tables, always staged
                m.code_bits = c->d_code_bits;
                m.zeros = c->d_zeros;
                m.code_bits_stride = c->code_bits_stride;
                lds = (unsigned)mfma_bf16_lds_bytes(rt, nct, fmt, m.nslots, m.rep_stride, c->code_bits_stride);
                GAT_HIP(c, launch_mfma_bf16(m, rt, nct, fmt, (unsigned)grid_m, lds, c->stream));
            } else {
                m.codes_in_lds = mfma_lds_bytes(nct, CT, m.rep_stride, c->code_row_stride, 1) <= 160 * 1024;
                lds = (unsigned)mfma_lds_bytes(nct, CT, m.rep_stride, c->code_row_stride, m.codes_in_lds);
                GAT_HIP(c, launch_mfma(m, nct, (unsigned)grid_m, lds, c->stream));
            }
            const bool fin_m = !atomic_m && m.splits > 1;
            if (fin_m)
                GAT_HIP(c, launch_finalize(c->d_partial, out_re, out_im, m.splits, L * M * 2, (long long)B * K, c->stream));
            c->last.workgroups = (int32_t)grid_m;
            c->last.threads = kind == 2 ? mfma_bf16_threads(rt, nct) : 2 * kThreads;
            c->last.splits = m.splits;
            c->last.ant_tile = kind == 2 ? 16 * rt : 16;
            c->last.vec = 4;
            c->last.lds_bytes = (int32_t)lds;
            c->last.finalize_launched = fin_m ? 1 : 0;
            c->last.matrix_core = kind;
            c->last.channels_per_wg = kind == 2 ? m.nslots : nct * CT;
            c->last.blocks_per_wg = 1;
            c->last.prefetch_depth = 0;
            return GAT_OK;
        }
    }
    c->last.matrix_core = 0;

    // ---- vector kernel (gat_dc.h): launch geometry ---------------------------------------------------------
    // aw: antenna tiles (waves) per workgroup -- 16 antennas on 4 waves walk the same samples, so carrier and replica
    //     are produced once per workgroup; kt: channels a workgroup loops over with the samples held in registers.
    const int AT = M / MT;
    int aw = 1, kt = 1;
    if (vec == 4 && MT == 4) aw = AT % 4 == 0 ? 4 : (AT % 2 == 0 ? 2 : 1);
    aw = std::min(aw, c->max_aw);
    if (vec == 4 && aw == 4 && sig->chan_stride == 0 && K > 1) kt = K >= 3 ? 4 : 2;
    kt = std::min(kt, c->max_kt);
    if (plan_out) aw = 1, kt = 1;
    // tap launches: sorted taps cut into groups of <= kMaxTapsPerLaunch whose span fits the LDS replica segment
    int order[GAT_MAX_TAPS];
    for (int l = 0; l < L; ++l) order[l] = l;
    std::stable_sort(order, order + L, [&](int x, int y) { return shifts[x] < shifts[y]; });
    int max_taps = 1; // taps of the widest launch (register accumulators 2 * MT * taps * kt)
    for (int t0 = 0; t0 < L;) {
        int t1 = t0 + 1;
        while (t1 < L && t1 - t0 < kMaxTapsPerLaunch && (long long)shifts[order[t1]] - shifts[order[t0]] <= kMaxLaunchSpan) ++t1;
        max_taps = std::max(max_taps, t1 - t0);
        t0 = t1;
    }
    while (kt > 1 && !dc_has_instance(MT, max_taps, vec, aw, kt)) kt >>= 1;
    while (aw > 1 && !dc_has_instance(MT, max_taps, vec, aw, kt)) aw >>= 1;
    // LDS: two workgroups per CU at least (80 KB each); a chip table that does not even fit alone is an error
    auto lds_of = [&](int kt_, int aw_) { return dc_lds_bytes(kt_, MT, c->code_row_stride, dc_chunk(vec, fmt, aw_)); };
    while (kt > 1 && lds_of(kt, aw) > 80 * 1024) kt >>= 1;
    if (lds_of(kt, aw) > 160 * 1024)
        return fail(c, GAT_ERR_RANGE, "code table too long for the LDS-resident chip table of the vector kernel");
    if (!dc_has_instance(MT, max_taps, vec, aw, kt)) return fail(c, GAT_ERR_UNSUPPORTED, "no kernel instance for this shape");
    const int AG = AT / aw;
    const int KG = (K + kt - 1) / kt;

    const long long groups = (long long)B * KG * AG;
    // One-wave workgroups: short blocks (a few steps of a four-wave workgroup) of one- or two-antenna tiles in a stream
    // long enough to fill the chip with single waves.  Per block the set-up (parameters, rotations, walk constants) is
    // then done by one wave instead of four, and no wave waits at a workgroup barrier.
    int nw = 4;
    if (c->one_wave && !plan_out && vec == 4 && aw == 1 && kt == 1 && MT <= 2 && c->code_row_stride <= 2048 &&
        (N + dc_chunk(vec, fmt, 1) - 1) / dc_chunk(vec, fmt, 1) <= 8 &&
        groups >= (c->one_wave_min >= 0 ? c->one_wave_min : 32ll * c->num_cus) &&
        c->max_aw >= 4 /* the (1, 1, 1) tiling of the A/B tests keeps the four-wave geometry */ &&
        dc_has_instance(MT, max_taps, vec, 1, 1, 1))
        nw = 1;
    const long long chunk = dc_chunk(vec, fmt, aw, nw);
    // Workgroups per CU the split aims for: 8 -- except for the channel-looping instances (KT >= 2: 170-250 registers,
    // two workgroups resident per CU), where a finer split only adds partial sums, a second launch and workgroup starts
    // (configs[3] shard, 512 tiles: 2 / 4 / 8 per CU = 0.667 / 0.675 / 0.687 ms, profiles/r03/r03a_c4_split.txt).
    const int per_cu = c->wgs_per_cu > 0 ? c->wgs_per_cu : (kt >= 2 ? 2 : 8);
    const long long target = plan_out ? plan_out->max_wgs : (long long)per_cu * c->num_cus * (nw == 1 ? 4 : 1);
    long long chunks = 0, splits = 1, cps = 1, bpw = 1;
    auto plan = [&](long long slack) { // slack: virtual samples in front of a block (line alignment, below)
        chunks = (N + slack + chunk - 1) / chunk;
        splits = std::max<long long>(1, (target + groups - 1) / groups);
        splits = std::min(splits, chunks);
        // tiny blocks (latency regime): a second launch costs more than a few serial steps
        // (a resident correlator has no second launch: its workgroups post their sums to the host, which adds them)
        if (chunks <= 4 && !(flags & GAT_FLAG_ATOMIC) && !plan_out) splits = 1;
        cps = (chunks + splits - 1) / splits;
        splits = (chunks + cps - 1) / cps;
    };
    plan(0);
    // short blocks in a long stream: one workgroup loops over several consecutive blocks (chip table, channel set-up
    // and the workgroup launch are paid once) while the chip stays filled 16 workgroups deep per CU
    if (splits == 1 && c->max_bpw > 1) {
        const long long by_fill = std::max<long long>(1, groups / (16ll * c->num_cus * (nw == 1 ? 4 : 1)));
        const long long by_len = std::max<long long>(1, (nw == 1 ? 64 : 16) / chunks);
        bpw = std::min<long long>(std::min(by_fill, by_len), c->max_bpw);
        if (c->force_bpw > 0) bpw = std::min<long long>(c->force_bpw, B); // A/B runs: option dc_bpw_force
    }
    // Line alignment (gat_dc.h): where a block of some antenna may start off a 128-byte line -- the base pointer or a stride
    // that is applied is no multiple of 128 bytes (N = 50 000 floats: every other block) -- workgroups walk each block from
    // the line its first sample lies in: up to 112 bytes of virtual samples in front of the block, hence the slack in the
    // chunk count.  Four-wave workgroups that own one block each (several short blocks per workgroup keep their walk across
    // block boundaries instead); option dc_align = 0 turns it off for A/B runs.
    // (a resident correlator is told the block's offset with every call: it always walks from the line)
    const bool align_head = c->align_head && vec == 4 && nw == 4 && bpw == 1 &&
                            (plan_out || (reinterpret_cast<uintptr_t>(sig->re) & 127u) != 0 || (B > 1 && (sig->block_stride * plane_bytes) % 128 != 0) ||
                             (M > MT && (sig->ant_stride * plane_bytes * MT) % 128 != 0) || (sig->chan_stride * plane_bytes) % 128 != 0);
    if (align_head) plan(112 / plane_bytes);
    const long long BG = (B + bpw - 1) / bpw;
    const long long tiles = BG * AG * splits;
    const long long grid_wgs = ((tiles + 7) / 8) * 8 * KG;
    if (grid_wgs >= (1ll << 31)) return fail(c, GAT_ERR_RANGE, "grid too large");

    const bool atomic = (flags & GAT_FLAG_ATOMIC) != 0;
    const size_t out_elems = (size_t)B * K * L * M;
    if (atomic) {
        GAT_HIP(c, hipMemsetAsync(out_re, 0, out_elems * sizeof(float), c->stream));
        GAT_HIP(c, hipMemsetAsync(out_im, 0, out_elems * sizeof(float), c->stream));
    } else if (splits > 1 && !plan_out) {
        const int32_t rc = ensure_partial(c, (size_t)B * K * splits * L * M * 2 * sizeof(float));
        if (rc != GAT_OK) return rc;
    }

    DcArgs a{};
    a.re = sig->re;
    a.im = sig->im;
    a.params = params_dev;
    if (!params_dev && !plan_out) std::memcpy(a.inl, params_inline, (size_t)B * K * sizeof(gat_channel_params));
    a.codes = c->d_codes;
    a.out_re = out_re;
    a.out_im = out_im;
    a.partial = c->d_partial;
    a.total_wgs = (unsigned)(tiles * KG);
    // completion flag: small launches outside a stream capture (a replayed graph would store a stale number)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(c->stream, &cap);
    // (library-owned streams only: nobody else can have enqueued newer work on them behind the library's back)
    const bool flagged = !plan_out && c->own_stream && c->d_flag && c->flag_max_wgs > 0 && tiles * KG <= c->flag_max_wgs && cap == hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    auto next_seq = [&]() { // sequence numbers of flagged launches: never 0 (0 = "nothing to wait for")
        if (++c->flag_seq == 0) ++c->flag_seq;
        return c->flag_seq;
    };
    a.N = N;
    a.ant_stride = sig->ant_stride;
    a.block_stride = sig->block_stride;
    a.chan_stride = sig->chan_stride;
    a.fs = fs;
    a.M = M;
    a.K = K;
    a.B = B;
    a.Lc = c->Lc;
    a.num_prns = c->P;
    a.code_row_stride = c->code_row_stride;
    a.KG = KG;
    a.splits = (int)splits;
    a.chunks_per_split = (int)cps;
    a.total_chunks = (int)chunks;
    a.ant_groups = AG;
    a.blocks_per_wg = (int)bpw;
    a.num_tiles = (int)tiles;
    a.Ltot = L;
    a.flags = flags;
    a.keep_l2 = c->keep_l2 >= 0 ? c->keep_l2 : (KG > 1 && sig->chan_stride == 0);
    a.n_vec = (int)(vec == 4 ? N - N % spv : N);
    a.align_head = align_head ? 1 : 0;
    // a block length that is no multiple of the load group: the N % spv samples behind the last whole group are added
    // by dc_tail_kernel, one more (tiny) launch behind the vector kernel and its second stage
    const bool tail = vec == 4 && N % spv != 0;
    a.max_abs_shift = (int)max_shift;

    DcLaunch cfg{};
    cfg.ant_tile = MT;
    cfg.aw = aw;
    cfg.kt = kt;
    cfg.nw = nw;
    // Two register sets of samples (steps c+1 and c+2 in flight): the streaming regime of the four-antenna <= 3-tap tile
    // only -- every byte read once (one channel group), a workgroup owns whole blocks (no split), >= 2 steps per block.
    // (float samples: with int16 / int8 pairs the conversions make the step vector-bound and the third wave per SIMD that
    // the second set costs is worth more: 0.206 -> 0.209 ms, 0.169 -> 0.170 ms)
    const bool deep_ok = !plan_out && c->max_depth >= 2 && vec == 4 && splits == 1 && KG == 1 && c->keep_l2 != 1 && sig->chan_stride == 0 && chunks >= 2 &&
                         (fmt == GAT_LAYOUT_PLANAR || fmt == GAT_LAYOUT_INTERLEAVED);
    cfg.vec = vec;
    cfg.format = fmt;
    cfg.grid = (unsigned)grid_wgs;
    const int seg_max = nw == 1 ? c->one_wave_seg : dc_segment_steps((int)chunk, kt, MT);
    cfg.lds_bytes = (unsigned)dc_lds_bytes(kt, MT, c->code_row_stride, (int)chunk);

    // Taps in any order: tap_index maps each tap of a launch back to its position in the caller's list
    // (a single-tap launch always fits: span 0).
    for (int t0 = 0; t0 < L;) {
        int t1 = t0 + 1;
        while (t1 < L && t1 - t0 < kMaxTapsPerLaunch &&
               (long long)shifts[order[t1]] - shifts[order[t0]] <= kMaxLaunchSpan)
            ++t1;
        cfg.taps = t1 - t0;
        for (int l = 0; l < kMaxTapsPerLaunch; ++l) {
            a.shifts[l] = shifts[order[t0 + std::min(l, cfg.taps - 1)]];
            a.tap_index[l] = order[t0 + std::min(l, cfg.taps - 1)];
        }
        a.rep_span = a.shifts[cfg.taps - 1] - a.shifts[0];
        // Replica layout in LDS (gat_dc.h): linear, one 8-byte-aligned vector read per tap and 4 samples.  Taps at an
        // even distance from the first read the replica itself; any tap at an odd distance needs the copy stored one
        // entry further, and the segment shrinks so that both fit the channel's share of LDS.
        bool odd = false;
        for (int l = 0; l < cfg.taps; ++l) odd |= ((a.shifts[l] - a.shifts[0]) & 1) != 0;
        int seg = seg_max;
        if (nw == 1) { // the replica's LDS is sized for this launch: segment + tap span + one entry per producer lane
            cfg.depth = 1;
            a.seg_steps = (int)std::min<long long>(seg, cps);
            if (deep_ok && seg >= 2 && dc_has_instance(MT, cfg.taps, vec, aw, kt, nw, 2)) {
                cfg.depth = 2; // whole groups of two steps per segment; the kernel pads the block's last group
                a.seg_steps = (int)std::min<long long>(seg - seg % 2, (cps + 1) / 2 * 2);
            }
            const int one = dc_rep_copy_floats(a.seg_steps, (int)chunk, a.rep_span, 64);
            a.rep_copy_stride = odd ? one : 0;
            a.rep_chan_floats = ((odd ? 2 : 1) * one + 7) & ~7;
            cfg.lds_bytes = (unsigned)dc_lds_bytes_one_wave(a.rep_chan_floats, c->code_row_stride);
        } else {
            // An instance that holds four waves per SIMD (dc_min_waves) needs four workgroups per CU to get them: with
            // 10 KB chip tables (GPS L5) the full eight-step segment makes a workgroup 47 KB -- three per CU.  Such launches
            // take a segment short enough for 40 KB (configs[2]: six steps; 1.151 -> 1.106 ms together with the two-sample
            // passes that bring the five-tap instance to 128 registers, profiles/r04/r04g_c2_four_waves.txt).
            // (a tap span beyond the default sizing -- seven taps half a chip apart at 262 MHz span 768 samples -- gets the
            // room it needs in the same launch instead of a second launch: 22.8 -> 17 us for that call)
            const int span_sz = std::max(kMaxReplicaSpan, a.rep_span);
            const int want_waves = dc_min_waves(MT, cfg.taps, kt, 1, fmt);
            if (want_waves >= 4)
                while (seg > 2 && dc_lds_bytes_floats(kt, c->code_row_stride, dc_rep_chan_floats_steps(seg, (int)chunk, span_sz)) > (size_t)(160 / want_waves) * 1024) --seg;
            if (span_sz > kMaxReplicaSpan)
                while (seg > 1 && dc_lds_bytes_floats(kt, c->code_row_stride, dc_rep_chan_floats_steps(seg, (int)chunk, span_sz)) > 64 * 1024) --seg;
            const int chan_floats = dc_rep_chan_floats_steps(seg, (int)chunk, span_sz);
            if (dc_lds_bytes_floats(kt, c->code_row_stride, chan_floats) > 160 * 1024)
                return fail(c, GAT_ERR_RANGE, "tap span and code table do not fit the LDS of one workgroup");
            a.rep_chan_floats = chan_floats;
            cfg.lds_bytes = (unsigned)dc_lds_bytes_floats(kt, c->code_row_stride, chan_floats);
            if (odd)
                while (seg > 1 && 2 * dc_rep_copy_floats(seg, (int)chunk, a.rep_span) > chan_floats) --seg;
            cfg.depth = 1;
            if (deep_ok && seg >= 2 && dc_has_instance(MT, cfg.taps, vec, aw, kt, nw, 2)) {
                cfg.depth = 2;
                seg -= seg % 2; // whole groups of two steps per segment; the kernel pads the block's last group
            }
            a.seg_steps = (int)std::min<long long>(seg, (cps + cfg.depth - 1) / cfg.depth * cfg.depth);
            a.rep_copy_stride = odd ? dc_rep_copy_floats(a.seg_steps, (int)chunk, a.rep_span) : 0;
        }
        for (int l = 0; l < kMaxTapsPerLaunch; ++l) {
            const int d = a.shifts[l] - a.shifts[0];
            a.tap_off[l] = (d & 1) ? a.rep_copy_stride + d - 1 : d;
        }
        // completion flag: carried by the call's last launch -- the last tap group's kernel, or the second stage behind it
        const bool later_follows = (!atomic && splits > 1) || tail;
        if (flagged && !later_follows && t1 >= L) {
            a.done_counter = c->d_done;
            a.host_flag = c->d_flag;
            a.flag_seq = next_seq();
        }
        if (plan_out) {
            if (t1 < L) return fail(c, GAT_ERR_UNSUPPORTED, "resident correlator: the taps need more than one launch");
            if (vec != 4 || tail) return fail(c, GAT_ERR_UNSUPPORTED, "resident correlator: block starts must be 16-byte aligned and num_samples a multiple of the load group");
            if (cfg.depth != 1 || nw != 4) return fail(c, GAT_ERR_UNSUPPORTED, "resident correlator: no instance for this geometry");
            plan_out->a = a;
            plan_out->a.keep_l2 = 0; // the resident instances read the signal with non-temporal loads
            plan_out->cfg = cfg;
            return GAT_OK;
        }
        GAT_HIP(c, launch_dc(a, cfg, c->stream));
        t0 = t1;
    }
    const bool fin = !atomic && splits > 1;
    if (fin) {
        const bool carry = flagged && !tail;
        GAT_HIP(c, launch_finalize(c->d_partial, out_re, out_im, (int)splits, L * M * 2, (long long)B * K, c->stream,
                                   carry ? c->d_done : nullptr, c->d_flag, carry ? next_seq() : 0u));
    }
    if (tail) {
        DcTailArgs t{};
        t.re = sig->re;
        t.im = sig->im;
        t.params = params_dev;
        if (!params_dev) std::memcpy(t.inl, params_inline, (size_t)B * K * sizeof(gat_channel_params));
        t.codes = c->d_codes;
        t.out_re = out_re;
        t.out_im = out_im;
        if (flagged) {
            t.done_counter = c->d_done;
            t.host_flag = c->d_flag;
            t.flag_seq = next_seq();
        }
        t.N = N; t.ant_stride = sig->ant_stride; t.block_stride = sig->block_stride; t.chan_stride = sig->chan_stride;
        t.fs = fs;
        t.M = M; t.K = K; t.B = B; t.L = L; t.Lc = c->Lc; t.num_prns = c->P; t.code_row_stride = c->code_row_stride;
        t.format = fmt; t.n_vec = a.n_vec; t.max_abs_shift = (int)max_shift;
        for (int l = 0; l < L; ++l) t.shifts[l] = shifts[l];
        GAT_HIP(c, launch_dc_tail(t, c->stream));
    }
    if (flagged) c->wait_seq = c->flag_seq;

    c->last.workgroups = (int32_t)cfg.grid;
    c->last.threads = 64 * nw;
    c->last.splits = (int32_t)splits;
    c->last.ant_tile = MT * aw;
    c->last.vec = vec;
    c->last.lds_bytes = (int32_t)cfg.lds_bytes;
    c->last.finalize_launched = fin ? 1 : 0;
    c->last.channels_per_wg = kt;
    c->last.blocks_per_wg = (int32_t)bpw;
    c->last.prefetch_depth = cfg.depth;
    return GAT_OK;
}



namespace {

int32_t tracking_run_enqueue(gat_ctx *c, const gat_signal_desc *sig, int32_t num_blocks, int32_t K, int32_t L,
                             const int32_t *shifts, double fs, const gat_loop_config *cfg, gat_loop_state *state,
                             gat_channel_params *params_a, gat_channel_params *params_b, float *acc_re, float *acc_im,
                             int64_t acc_block_stride, uint32_t flags, int32_t *current_is_b)
{
    const size_t sample_bytes = sig->layout == GAT_LAYOUT_PLANAR ? 4 : sig->layout == GAT_LAYOUT_INTERLEAVED ? 8
                              : sig->layout == GAT_LAYOUT_INTERLEAVED_I16 ? 4 : 2;
    gat_channel_params *cur = params_a, *nxt = params_b;
    for (int32_t b = 0; b < num_blocks; ++b) { // everything is enqueued on the ctx stream, nothing synchronises
        gat_signal_desc d = *sig;
        const size_t off = (size_t)b * (size_t)sig->block_stride * sample_bytes;
        d.re = static_cast<const char *>(sig->re) + off;
        if (sig->im) d.im = static_cast<const char *>(sig->im) + off;
        float *o_re = acc_re + (size_t)b * (size_t)acc_block_stride, *o_im = acc_im + (size_t)b * (size_t)acc_block_stride;
        int32_t rc = correlate_impl(c, &d, cur, 1, K, L, shifts, fs, o_re, o_im, flags);
        if (rc != GAT_OK) return rc;
        rc = gat_tracking_update(c, o_re, o_im, K, sig->num_ants, cfg, state, cur, nxt);
        if (rc != GAT_OK) return rc;
        std::swap(cur, nxt);
    }
    if (current_is_b) *current_is_b = cur == params_b ? 1 : 0;
    return GAT_OK;
}

template <typename T>
void key_put(std::vector<unsigned char> &k, const T &v)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&v);
    k.insert(k.end(), p, p + sizeof(T));
}

// GAT_FLAG_GRAPH: replay the launch sequence of `enqueue` as one instantiated hipGraph when the same call repeats.
// Every argument that shapes the sequence is part of the key (make_key), together with the library-owned pointers and
// sizes the recorded launches bake in; a call with a known key replays its graph.  The first call with a key runs
// eagerly (this also sizes the library's scratch buffers, which must not be reallocated inside a capture -- a
// reallocation drops every recorded graph), then the same sequence is recorded for the following calls under the key of
// the buffers the capture really uses.  Up to kMaxLoopGraphs graphs are kept (least recently used goes).
template <class MakeKey, class Enqueue>
int32_t graph_replay_or_record(gat_ctx *c, MakeKey make_key, Enqueue enqueue)
{
    {
        const std::vector<unsigned char> key = make_key();
        for (auto &g : c->loop_graphs)
            if (g.exec && g.key == key) {
                c->wait_seq = 0;
                g.last_use = ++c->loop_graph_clock;
                GAT_HIP(c, hipGraphLaunch(g.exec, c->stream));
                return GAT_OK;
            }
    }
    int32_t rc = enqueue();
    if (rc != GAT_OK) return rc;
    const std::vector<unsigned char> key = make_key();
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return GAT_OK; // no graph (e.g. the legacy default stream): stay eager
    }
    rc = enqueue();
    const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc == GAT_OK && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        if (c->loop_graphs.size() >= kMaxLoopGraphs) { // evict the least recently used
            size_t lru = 0;
            for (size_t i = 1; i < c->loop_graphs.size(); ++i)
                if (c->loop_graphs[i].last_use < c->loop_graphs[lru].last_use) lru = i;
            if (c->loop_graphs[lru].exec) {
                (void)hipStreamSynchronize(c->stream); // its last replay may still be running
                (void)hipGraphExecDestroy(c->loop_graphs[lru].exec);
            }
            c->loop_graphs.erase(c->loop_graphs.begin() + (long)lru);
        }
        gat_ctx::LoopGraph g;
        g.key = key;
        g.exec = exec;
        g.last_use = ++c->loop_graph_clock;
        c->loop_graphs.push_back(std::move(g));
    }
    if (graph) (void)hipGraphDestroy(graph);
    return GAT_OK;
}

// the part of a graph key every recorded launch sequence shares: kernel-selection knobs and library-owned buffers
void key_put_ctx(std::vector<unsigned char> &key, const gat_ctx *c)
{
    key_put(key, c->mc_mode); key_put(key, c->max_aw); key_put(key, c->max_kt); key_put(key, c->max_bpw); key_put(key, c->force_bpw);
    key_put(key, c->wgs_per_cu); key_put(key, c->max_depth); key_put(key, c->one_wave); key_put(key, c->keep_l2); key_put(key, c->align_head);
    key_put(key, c->d_codes); key_put(key, c->d_code_bits); key_put(key, c->Lc); key_put(key, c->P);
    key_put(key, c->d_partial); key_put(key, c->partial_bytes);
}

} // namespace

namespace {

// gat_set_option: launch-geometry options (tests force a code path on a small case with them, A/B measurements compare
// geometries).  None changes a result beyond summation order.
struct OptionDesc {
    const char *name;
    long long lo, hi;
};
constexpr OptionDesc kOptions[] = {
    {"sync_flag_wgs", 0, 1 << 20},   // largest launch (workgroups) that carries the completion flag; 0: never
    {"max_ant_tile", 1, kMaxAntTile}, // antennas per wave
    {"dc_aw", 1, 4},                 // antenna tiles (waves) per workgroup, cap
    {"dc_kt", 1, 4},                 // channels per workgroup, cap
    {"dc_bpw", 1, 1 << 20},          // consecutive blocks per workgroup, cap
    {"dc_bpw_force", 0, 1 << 20},    // blocks per workgroup whatever the planner's rule says (0: planner)
    {"dc_wgs_per_cu", 0, 1024},      // workgroups per CU the split planner aims for (0: by instance)
    {"dc_one_wave", 0, 1},           // one-wave workgroups allowed
    {"dc_one_wave_min", -1, 1ll << 40}, // fewest (block, channel, tile) groups for them (-1: 32 per CU)
    {"dc_ow_seg", 1, kUcarSteps},    // steps per replica segment of a one-wave workgroup
    {"dc_depth", 1, 2},              // cap of the sample prefetch depth (register sets per wave)
    {"dc_keep_l2", -1, 1},           // sample loads: -1 by rule (plain when channel groups share a tile through L2), 0 non-temporal, 1 plain
    {"dc_align", 0, 1},              // blocks walked from the 128-byte line their first sample lies in (1) or from the sample itself (0)
};

int32_t set_option(gat_ctx *c, const char *name, long long v)
{
    const OptionDesc *o = nullptr;
    for (const auto &d : kOptions)
        if (std::strcmp(d.name, name) == 0) o = &d;
    if (!o) return fail(c, GAT_ERR_ARG, "unknown option");
    if (v < o->lo || v > o->hi) return fail(c, GAT_ERR_RANGE, "option value out of range");
    const std::string n = name;
    if (n == "sync_flag_wgs") c->flag_max_wgs = (int)v;
    else if (n == "max_ant_tile") c->max_ant_tile = (int)v;
    else if (n == "dc_aw") c->max_aw = (int)v;
    else if (n == "dc_kt") c->max_kt = (int)v;
    else if (n == "dc_bpw") c->max_bpw = (int)v;
    else if (n == "dc_bpw_force") c->force_bpw = (int)v;
    else if (n == "dc_wgs_per_cu") c->wgs_per_cu = (int)v;
    else if (n == "dc_one_wave") c->one_wave = (int)v;
    else if (n == "dc_one_wave_min") c->one_wave_min = v;
    else if (n == "dc_ow_seg") c->one_wave_seg = (int)v;
    else if (n == "dc_depth") c->max_depth = (int)v;
    else if (n == "dc_keep_l2") c->keep_l2 = (int)v;
    else if (n == "dc_align") c->align_head = (int)v;
    return GAT_OK;
}

} // namespace

extern "C" {

GAT_API int32_t gat_create(int32_t device, void *hip_stream, gat_ctx **out_ctx)
{
    if (!out_ctx) return GAT_ERR_ARG;
    *out_ctx = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return -(int32_t)e;
    if (device < 0 || device >= ndev) return GAT_ERR_RANGE;
    gat_ctx *c = new (std::nothrow) gat_ctx();
    if (!c) return GAT_ERR_NOMEM;
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        delete c;
        return -(int32_t)e;
    }
    if (hip_stream != GAT_OWN_STREAM) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream); // NULL == the default stream
    } else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            delete c;
            return -(int32_t)e;
        }
        c->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if (hipMalloc(&c->d_zeros, 64) == hipSuccess) (void)hipMemset(c->d_zeros, 0, 64);
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_flag), 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
        *c->h_flag = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_flag), c->h_flag, 0) != hipSuccess) c->d_flag = nullptr;
        if (c->d_flag && hipMalloc(reinterpret_cast<void **>(&c->d_done), 64) == hipSuccess) (void)hipMemset(c->d_done, 0, 64);
        else c->d_flag = nullptr;
    }
    (void)hipGetLastError();
#ifdef GAT_DEV
    // development builds only (-DGAT_DEV; gat_version() names the flag): the launch-geometry options of gat_set_option
    // taken from the environment, so that scripts can A/B them without code changes.  The product library reads no
    // environment variable that selects kernels or geometry.
    for (const auto &o : kOptions) {
        std::string env = "GAT_";
        for (const char *p = o.name; *p; ++p) env += (char)std::toupper((unsigned char)*p);
        if (const char *v = std::getenv(env.c_str())) (void)set_option(c, o.name, std::atoll(v));
    }
    if (const char *v = std::getenv("GAT_NO_MFMA")) c->mc_mode = v[0] == '1' ? 0 : 1;
    if (const char *v = std::getenv("GAT_MC_MODE")) c->mc_mode = (v[0] >= '0' && v[0] <= '3') ? v[0] - '0' : 1;
#endif
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        delete c;
        return -(int32_t)e;
    }
    *out_ctx = c;
    return GAT_OK;
}

GAT_API int32_t gat_destroy(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    (void)hipSetDevice(c->device);
    park_residents(c);
    for (gat_resident *r : c->residents) resident_free(r); // handles of correlators that were not closed die with the context
    c->residents.clear();
    (void)hipStreamSynchronize(c->stream);
    if (c->d_codes) (void)hipFree(c->d_codes);
    if (c->d_code_bits) (void)hipFree(c->d_code_bits);
    if (c->d_zeros) (void)hipFree(c->d_zeros);
    drop_loop_graphs(c);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_done) (void)hipFree(c->d_done);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->d_params) (void)hipFree(c->d_params);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GAT_OK;
}

GAT_API int32_t gat_set_stream(gat_ctx *c, void *hip_stream)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    drop_loop_graphs(c);
    if (c->own_stream) {
        GAT_HIP(c, hipStreamDestroy(c->stream));
        c->own_stream = false;
    }
    if (hip_stream != GAT_OWN_STREAM) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        GAT_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return GAT_OK;
}

GAT_API int32_t gat_sync(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    if (c->wait_seq && c->h_flag) {
        // the newest work on the stream is a flagged correlator launch: its last workgroup stores wait_seq into pinned
        // host memory after its results are out (system-scope release).  Launches of one stream finish in order, so
        // the flag reaches wait_seq exactly when everything enqueued is done.  Bounded spin, then the ordinary wait.
        const unsigned want = c->wait_seq;
        c->wait_seq = 0;
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == want) return GAT_OK;
            if ((spins & 255u) == 255u) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 200000) break; // 200 us
            }
        }
    }
    GAT_HIP(c, hipSetDevice(c->device));
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    return GAT_OK;
}

GAT_API const char *gat_last_error(const gat_ctx *c) { return c ? c->err.c_str() : "null context"; }

GAT_API int32_t gat_device_info(gat_ctx *c, char *name_buf, size_t name_len, int32_t *runtime_version,
                                int32_t *num_cus)
{
    if (!c) return GAT_ERR_ARG;
    hipDeviceProp_t prop;
    GAT_HIP(c, hipGetDeviceProperties(&prop, c->device));
    if (name_buf && name_len) {
        std::snprintf(name_buf, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (runtime_version) {
        int v = 0;
        GAT_HIP(c, hipRuntimeGetVersion(&v));
        *runtime_version = v;
    }
    if (num_cus) *num_cus = prop.multiProcessorCount;
    return GAT_OK;
}

GAT_API int32_t gat_set_codes(gat_ctx *c, const int8_t *codes_host, int32_t code_length, int32_t num_prns)
{
    if (!c || !codes_host) return fail(c, GAT_ERR_ARG, "null argument");
    if (code_length < 1 || num_prns < 1) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    // the vector kernel keeps a workgroup's chip table in LDS next to one replica segment (~40 KB): 160 KB - that
    if (code_length > 120000) return fail(c, GAT_ERR_RANGE, "code table does not fit in LDS (max 120000 chips)");
    GAT_HIP(c, hipSetDevice(c->device));
    park_residents(c); // their kernels hold the old tables (and hipFree waits for the whole device)
    for (gat_resident *r : c->residents) r->stale = true;
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    drop_loop_graphs(c); // recorded launches point at the old tables
    if (c->d_codes) {
        GAT_HIP(c, hipFree(c->d_codes));
        c->d_codes = nullptr;
    }
    if (c->d_code_bits) {
        GAT_HIP(c, hipFree(c->d_code_bits));
        c->d_code_bits = nullptr;
    }
    const int stride = (code_length + 15) & ~15; // 16-byte rows: dc_kernel stages them with 16-byte copies
    const size_t bytes = (size_t)stride * num_prns;
    GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_codes), bytes));
    GAT_HIP(c, hipMemset(c->d_codes, 0, bytes));
    GAT_HIP(c, hipMemcpy2D(c->d_codes, (size_t)stride, codes_host, (size_t)code_length, (size_t)code_length,
                           (size_t)num_prns, hipMemcpyHostToDevice));
    c->code_row_stride = stride;
    c->Lc = code_length;
    c->P = num_prns;
    // sign-bit tables for the split-bf16 matrix kernel (a chip only flips signs there): 1/8 of the LDS
    bool pm1 = true;
    for (size_t i = 0; i < (size_t)code_length * num_prns && pm1; ++i) pm1 = codes_host[i] == 1 || codes_host[i] == -1;
    if (pm1) {
        const int bstride = (((code_length + 31) / 32) + 3) & ~3;
        std::vector<uint32_t> bits((size_t)bstride * num_prns, 0u);
        for (int p = 0; p < num_prns; ++p)
            for (int i = 0; i < code_length; ++i)
                if (codes_host[(size_t)p * code_length + i] < 0) bits[(size_t)p * bstride + (i >> 5)] |= 1u << (i & 31);
        GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_code_bits), bits.size() * sizeof(uint32_t)));
        GAT_HIP(c, hipMemcpy(c->d_code_bits, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->code_bits_stride = bstride;
    }
    return GAT_OK;
}

GAT_API int32_t gat_downconvert_and_correlate_dev(gat_ctx *c, const gat_signal_desc *sig,
                                                  const gat_channel_params *params_dev, int32_t B,
                                                  int32_t K, int32_t L, const int32_t *shifts, double fs,
                                                  float *out_re, float *out_im, uint32_t flags)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    if (!(flags & GAT_FLAG_GRAPH)) return correlate_impl(c, sig, params_dev, B, K, L, shifts, fs, out_re, out_im, flags);
    // a receiver that calls the operator block after block on the same buffers: the one to three launches of a call
    // (tap groups, second stage) replayed as one instantiated graph
    if (!sig || !shifts || L < 1 || L > GAT_MAX_TAPS) return fail(c, GAT_ERR_ARG, "bad argument");
    const uint32_t kflags = flags & ~GAT_FLAG_GRAPH;
    auto make_key = [&]() {
        std::vector<unsigned char> key;
        key_put(key, (int)2 /* sequence: one correlate call */);
        key_put(key, sig->re); key_put(key, sig->im); key_put(key, sig->layout); key_put(key, sig->num_ants);
        key_put(key, sig->num_samples); key_put(key, sig->ant_stride); key_put(key, sig->block_stride);
        key_put(key, sig->chan_stride); key_put(key, params_dev); key_put(key, B); key_put(key, K); key_put(key, L);
        key_put(key, fs); key_put(key, out_re); key_put(key, out_im); key_put(key, kflags);
        for (int l = 0; l < L; ++l) key_put(key, shifts[l]);
        key_put_ctx(key, c);
        return key;
    };
    return graph_replay_or_record(c, make_key, [&]() { return correlate_impl(c, sig, params_dev, B, K, L, shifts, fs, out_re, out_im, kflags); });
}

GAT_API int32_t gat_downconvert_and_correlate(gat_ctx *c, const gat_signal_desc *sig,
                                              const gat_channel_params *params_host, int32_t B, int32_t K,
                                              int32_t L, const int32_t *shifts, double fs, float *out_re,
                                              float *out_im, uint32_t flags)
{
    if (!c) return GAT_ERR_ARG;
    if (!params_host || !sig || !shifts) return fail(c, GAT_ERR_ARG, "null argument");
    if (B < 1 || K < 1) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    GAT_HIP(c, hipSetDevice(c->device));
    // host-side validation the device-params variant cannot do
    long long max_shift = 0;
    for (int l = 0; l < L && l < GAT_MAX_TAPS; ++l)
        max_shift = std::max<long long>(max_shift, std::llabs((long long)shifts[l]));
    const size_t n = (size_t)B * K;
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (!(fs > 0.0) || !std::isfinite(fs)) return fail(c, GAT_ERR_ARG, "sampling frequency must be positive");
    {
        const int32_t rcv = validate_params(c, params_host, n, (double)(sig->num_samples + max_shift), fs);
        if (rcv != GAT_OK) return rcv;
    }
    if (n <= (size_t)kInlineParams) // no upload: the records ride in the kernel arguments
        return correlate_impl(c, sig, nullptr, B, K, L, shifts, fs, out_re, out_im, flags, params_host);
    const int32_t rc = upload_params(c, params_host, n);
    if (rc != GAT_OK) return rc;
    return correlate_impl(c, sig, c->d_params, B, K, L, shifts, fs, out_re, out_im, flags);
}

static int32_t gen_code_replica_impl(gat_ctx *c, float *rep, int64_t count, int32_t prn, double fc, double fs,
                                     double tau, int64_t first_shift, bool f32_coordinates)
{
    if (!c || !rep) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (count < 1) return fail(c, GAT_ERR_ARG, "count must be positive");
    if (prn < 0 || prn >= c->P) return fail(c, GAT_ERR_RANGE, "prn outside the code table");
    if (!(fs > 0.0) || !std::isfinite(fc) || !std::isfinite(tau)) return fail(c, GAT_ERR_ARG, "bad frequency / phase");
    if (count + std::llabs((long long)first_shift) >= (1ll << 30)) return fail(c, GAT_ERR_RANGE, "replica too long");
    if (!f32_coordinates && !code_span_ok(fc / fs, tau, (double)count + (double)std::llabs((long long)first_shift), c->Lc))
        return fail(c, GAT_ERR_RANGE, "code phase span too large");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_gen_code_replica");
    GAT_HIP(c, launch_gen_code_replica(rep, count, c->d_codes + (size_t)prn * c->code_row_stride, c->Lc, fc, fs, tau,
                                       first_shift, f32_coordinates, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_gen_code_replica(gat_ctx *c, float *rep, int64_t count, int32_t prn, double fc,
                                     double fs, double tau, int64_t first_shift)
{
    return gen_code_replica_impl(c, rep, count, prn, fc, fs, tau, first_shift, false);
}

GAT_API int32_t gat_gen_code_replica_f32coord(gat_ctx *c, float *rep, int64_t count, int32_t prn, double fc,
                                              double fs, double tau, int64_t first_shift)
{
    return gen_code_replica_impl(c, rep, count, prn, fc, fs, tau, first_shift, true);
}

GAT_API int32_t gat_gen_code_replica_multi(gat_ctx *c, float *rep, int64_t count, int64_t row_stride, int32_t K,
                                           const gat_channel_params *params_dev, double fs, int64_t first_shift)
{
    if (!c || !rep || !params_dev) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (count < 1 || K < 1 || K > 65535 || row_stride < count) return fail(c, GAT_ERR_ARG, "bad sizes");
    if (!(fs > 0.0) || count + std::llabs((long long)first_shift) >= (1ll << 30)) return fail(c, GAT_ERR_RANGE, "replica too long");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_gen_code_replica_multi");
    GAT_HIP(c, launch_gen_code_replica_multi(rep, count, row_stride, K, params_dev, c->d_codes, c->code_row_stride, c->Lc,
                                             c->P, fs, first_shift, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_downconvert_and_accumulate(gat_ctx *c, const gat_signal_desc *sig, const gat_channel_params *p,
                                               int32_t L, const int32_t *shifts, double fs, float *car_re, float *car_im,
                                               float *dw_re, float *dw_im, float *acc_re, float *acc_im)
{
    if (!c || !sig || !p || !shifts) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (sig->layout != GAT_LAYOUT_PLANAR || !sig->re || !sig->im) return fail(c, GAT_ERR_UNSUPPORTED, "planar float signal only");
    if (L < 1 || L > GAT_MAX_TAPS || sig->num_ants < 1 || sig->num_samples < 1 || sig->num_samples >= (1ll << 30))
        return fail(c, GAT_ERR_RANGE, "size out of range");
    if (p->prn < 0 || p->prn >= c->P) return fail(c, GAT_ERR_RANGE, "prn outside the code table");
    if (!(fs > 0.0) || !std::isfinite(p->code_freq_hz) || !(p->code_freq_hz >= 0.0) || !std::isfinite(p->carrier_freq_hz) ||
        !std::isfinite(p->code_phase_chips) || !std::isfinite(p->carrier_phase_cycles))
        return fail(c, GAT_ERR_ARG, "bad frequency / phase");
    long long max_shift = 0;
    for (int l = 0; l < L; ++l) max_shift = std::max<long long>(max_shift, std::llabs((long long)shifts[l]));
    if (sig->num_samples + max_shift >= (1ll << 30)) return fail(c, GAT_ERR_RANGE, "num_samples + |shift| must stay below 2^30");
    if (!code_span_ok(p->code_freq_hz / fs, p->code_phase_chips, (double)(sig->num_samples + max_shift), c->Lc))
        return fail(c, GAT_ERR_RANGE, "code phase span too large");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_downconvert_and_accumulate");
    // the tap list goes through the library's parameter scratch (device memory the kernel can read)
    const size_t need = ((size_t)L * sizeof(int32_t) + sizeof(gat_channel_params) - 1) / sizeof(gat_channel_params);
    if (need > c->params_cap) {
        if (c->d_params) {
            GAT_HIP(c, hipStreamSynchronize(c->stream));
            park_residents(c);
            GAT_HIP(c, hipFree(c->d_params));
            c->d_params = nullptr;
            c->params_cap = 0;
        }
        GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_params), need * sizeof(gat_channel_params)));
        c->params_cap = need;
    }
    GAT_HIP(c, hipMemcpyAsync(c->d_params, shifts, (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    GAT_HIP(c, launch_accumulate_debug(static_cast<const float *>(sig->re), static_cast<const float *>(sig->im),
                                       sig->num_samples, sig->num_ants, sig->ant_stride, *p,
                                       c->d_codes + (size_t)p->prn * c->code_row_stride, c->Lc, fs, L,
                                       reinterpret_cast<const int *>(c->d_params), car_re, car_im, dw_re, dw_im, acc_re, acc_im,
                                       c->stream));
    return GAT_OK;
}

static int32_t gen_signal_impl(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M, int64_t ant_stride,
                               int64_t block_stride, int32_t B, int32_t K, const gat_channel_params *params_dev, double fs,
                               double amplitude, const float *steering_cycles_dev, double noise_sigma, uint64_t seed);

GAT_API int32_t gat_gen_signal(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M,
                               int64_t ant_stride, int64_t block_stride, int32_t B, int32_t K,
                               const gat_channel_params *params_dev, double fs, double amplitude)
{
    return gen_signal_impl(c, re, im, layout, N, M, ant_stride, block_stride, B, K, params_dev, fs, amplitude, nullptr, 0.0, 0);
}

GAT_API int32_t gat_gen_signal_noisy(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M,
                                     int64_t ant_stride, int64_t block_stride, int32_t B, int32_t K,
                                     const gat_channel_params *params_dev, double fs, double amplitude,
                                     const float *steering_cycles_dev, double noise_sigma, uint64_t seed)
{
    if (c && (!(noise_sigma >= 0.0) || !std::isfinite(noise_sigma))) return fail(c, GAT_ERR_ARG, "noise sigma must be finite and >= 0");
    return gen_signal_impl(c, re, im, layout, N, M, ant_stride, block_stride, B, K, params_dev, fs, amplitude,
                           steering_cycles_dev, noise_sigma, seed);
}

static int32_t gen_signal_impl(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M, int64_t ant_stride,
                               int64_t block_stride, int32_t B, int32_t K, const gat_channel_params *params_dev, double fs,
                               double amplitude, const float *steering_cycles_dev, double noise_sigma, uint64_t seed)
{
    if (!c || !re || !params_dev) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (layout < GAT_LAYOUT_PLANAR || layout > GAT_LAYOUT_INTERLEAVED_I8) return fail(c, GAT_ERR_ARG, "unknown layout");
    if ((layout == GAT_LAYOUT_PLANAR) != (im != nullptr)) return fail(c, GAT_ERR_ARG, "signal pointers do not match the layout");
    if (N < 1 || N >= (1ll << 30) || M < 1 || B < 1 || B > 65535 || K < 1) return fail(c, GAT_ERR_RANGE, "size out of range");
    if (!(fs > 0.0) || !std::isfinite(amplitude)) return fail(c, GAT_ERR_ARG, "bad sampling frequency / amplitude");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_gen_signal");
    GAT_HIP(c, launch_gen_signal(re, im, layout, N, M, ant_stride, block_stride, B, K, params_dev, c->d_codes,
                                 c->code_row_stride, c->Lc, c->P, fs, (float)amplitude, steering_cycles_dev, (float)noise_sigma,
                                 (unsigned long long)seed, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_reduce_cplx_multi(gat_ctx *c, const float *in_re, const float *in_im, int64_t n,
                                      int32_t cols, float *out_re, float *out_im)
{
    if (!c || !in_re || !in_im || !out_re || !out_im) return fail(c, GAT_ERR_ARG, "null argument");
    if (n < 1 || cols < 1 || cols > 65535) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_reduce_cplx_multi");
    long long chunks = (n + 4 * kThreads - 1) / (4 * kThreads);
    const long long want = std::max<long long>(1, (4ll * c->num_cus + cols - 1) / cols);
    chunks = std::max<long long>(1, std::min(chunks, want));
    const int32_t rc = ensure_partial(c, (size_t)chunks * cols * 2 * sizeof(float));
    if (rc != GAT_OK) return rc;
    GAT_HIP(c, launch_reduce_stage1(in_re, in_im, n, cols, (int)chunks, c->d_partial, c->stream));
    GAT_HIP(c, launch_finalize(c->d_partial, out_re, out_im, (int)chunks, cols * 2, 1, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_tracking_update(gat_ctx *c, const float *acc_re, const float *acc_im, int32_t K, int32_t M,
                                    const gat_loop_config *cfg, gat_loop_state *state,
                                    const gat_channel_params *cur, gat_channel_params *next)
{
    if (!c || !acc_re || !acc_im || !cfg || !state || !cur || !next) return fail(c, GAT_ERR_ARG, "null argument");
    if (K < 1 || M < 1) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    const int L = cfg->num_taps;
    if (L < 1 || L > GAT_MAX_TAPS || cfg->early_index < 0 || cfg->early_index >= L || cfg->prompt_index < 0 ||
        cfg->prompt_index >= L || cfg->late_index < 0 || cfg->late_index >= L)
        return fail(c, GAT_ERR_RANGE, "tap indices outside the tap list");
    if (!(cfg->block_seconds > 0.0) || !(cfg->pll_bandwidth_hz >= 0.0) || !(cfg->dll_bandwidth_hz >= 0.0) ||
        !(cfg->code_freq_nominal_hz > 0.0) || !(cfg->carrier_center_hz > 0.0) || cfg->code_length < 1 ||
        !(cfg->early_late_spacing_chips > 0.0 && cfg->early_late_spacing_chips < 2.0))
        return fail(c, GAT_ERR_ARG, "bad loop configuration");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_tracking_update");
    GAT_HIP(c, launch_tracking_update(acc_re, acc_im, K, M, *cfg, state, cur, next, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_tracking_run(gat_ctx *c, const gat_signal_desc *sig, int32_t num_blocks, int32_t K, int32_t L,
                                 const int32_t *shifts, double fs, const gat_loop_config *cfg, gat_loop_state *state,
                                 gat_channel_params *params_a, gat_channel_params *params_b, float *acc_re,
                                 float *acc_im, int64_t acc_block_stride, uint32_t flags, int32_t *current_is_b)
{
    if (!c || !sig || !shifts || !cfg || !state || !params_a || !params_b || !acc_re || !acc_im)
        return fail(c, GAT_ERR_ARG, "null argument");
    if (num_blocks < 1 || acc_block_stride < 0) return fail(c, GAT_ERR_ARG, "bad block count / stride");
    if (cfg->num_taps != L || L < 1 || L > GAT_MAX_TAPS) return fail(c, GAT_ERR_ARG, "loop configuration and tap list disagree");
    if (flags & ~(GAT_FLAG_ATOMIC | GAT_FLAG_GRAPH)) return fail(c, GAT_ERR_ARG, "unknown flag bits");
    GAT_HIP(c, hipSetDevice(c->device));
    const TraceRange trace("gat_tracking_run");
    const uint32_t kflags = flags & ~GAT_FLAG_GRAPH;
    if (!(flags & GAT_FLAG_GRAPH))
        return tracking_run_enqueue(c, sig, num_blocks, K, L, shifts, fs, cfg, state, params_a, params_b, acc_re, acc_im,
                                    acc_block_stride, kflags, current_is_b);

    // hipGraph path: the 2-3 launches per block are too short to hide their launch gaps.  Every argument that
    // shapes the launch sequence is part of the key -- field by field (struct padding of a C caller is not
    // initialised) --, together with the library-owned pointers and sizes the recorded launches bake in; a call with
    // a known key replays its instantiated graph.  Up to kMaxLoopGraphs graphs are kept (least recently used goes).
    auto make_key = [&]() {
        std::vector<unsigned char> key;
        key_put(key, sig->re); key_put(key, sig->im); key_put(key, sig->layout); key_put(key, sig->num_ants);
        key_put(key, sig->num_samples); key_put(key, sig->ant_stride); key_put(key, sig->block_stride);
        key_put(key, sig->chan_stride); key_put(key, num_blocks); key_put(key, K); key_put(key, L); key_put(key, fs);
        key_put(key, cfg->block_seconds); key_put(key, cfg->pll_bandwidth_hz); key_put(key, cfg->dll_bandwidth_hz);
        key_put(key, cfg->code_freq_nominal_hz); key_put(key, cfg->carrier_center_hz); key_put(key, cfg->if_hz);
        key_put(key, cfg->early_late_spacing_chips); key_put(key, cfg->code_length); key_put(key, cfg->num_taps);
        key_put(key, cfg->early_index); key_put(key, cfg->prompt_index); key_put(key, cfg->late_index);
        key_put(key, state); key_put(key, params_a); key_put(key, params_b); key_put(key, acc_re); key_put(key, acc_im);
        key_put(key, acc_block_stride); key_put(key, kflags); key_put(key, (int)1 /* sequence: tracking run */);
        key_put_ctx(key, c);
        for (int l = 0; l < L; ++l) key_put(key, shifts[l]);
        return key;
    };
    if (current_is_b) *current_is_b = (num_blocks & 1) ? 1 : 0; // the buffers swap once per block
    return graph_replay_or_record(c, make_key, [&]() {
        return tracking_run_enqueue(c, sig, num_blocks, K, L, shifts, fs, cfg, state, params_a, params_b, acc_re, acc_im,
                                    acc_block_stride, kflags, nullptr);
    });
}

GAT_API int32_t gat_malloc(gat_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || bytes == 0) return fail(c, GAT_ERR_ARG, "bad argument");
    GAT_HIP(c, hipSetDevice(c->device));
    GAT_HIP(c, hipMalloc(out, bytes));
    return GAT_OK;
}

GAT_API int32_t gat_free(gat_ctx *c, void *p)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    park_residents(c); // hipFree waits for every kernel on the device: a resident one would hold it until its idle limit
    GAT_HIP(c, hipFree(p));
    return GAT_OK;
}

GAT_API int32_t gat_memcpy_h2d(gat_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(c, GAT_ERR_ARG, "null argument");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_memcpy_d2h(gat_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(c, GAT_ERR_ARG, "null argument");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_memset(gat_ctx *c, void *dst, int32_t value, size_t bytes)
{
    if (!c || !dst) return fail(c, GAT_ERR_ARG, "null argument");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_timer_start(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipEventRecord(c->ev0, c->stream));
    c->timer_running = true;
    return GAT_OK;
}

GAT_API int32_t gat_timer_stop(gat_ctx *c, float *ms)
{
    if (!c || !ms) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->timer_running) return fail(c, GAT_ERR_STATE, "timer not started");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipEventRecord(c->ev1, c->stream));
    GAT_HIP(c, hipEventSynchronize(c->ev1));
    GAT_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    c->timer_running = false;
    return GAT_OK;
}

#ifdef GAT_MFMA_STAMPS
extern "C" GAT_API int32_t gat_debug_read(gat_ctx *c, unsigned long long *host, size_t count)
{
    if (!c || !c->dbg_ptr) return GAT_ERR_STATE;
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    GAT_HIP(c, hipMemcpy(host, c->dbg_ptr, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return GAT_OK;
}
#endif

GAT_API int32_t gat_set_matrix_core(gat_ctx *c, int32_t enable)
{
    if (!c) return GAT_ERR_ARG;
    if (enable < GAT_MC_VECTOR || enable > GAT_MC_BF16_SPLIT) return fail(c, GAT_ERR_ARG, "unknown matrix-core mode");
    c->mc_mode = enable;
    drop_loop_graphs(c);
    return GAT_OK;
}

GAT_API int32_t gat_set_vector_tiling(gat_ctx *c, int32_t max_antenna_tiles, int32_t max_channels, int32_t max_blocks)
{
    if (!c) return GAT_ERR_ARG;
    if (max_antenna_tiles < 0 || max_channels < 0 || max_blocks < 0) return fail(c, GAT_ERR_ARG, "negative cap");
    if (max_antenna_tiles) c->max_aw = max_antenna_tiles >= 4 ? 4 : (max_antenna_tiles >= 2 ? 2 : 1);
    if (max_channels) c->max_kt = max_channels >= 4 ? 4 : (max_channels >= 2 ? 2 : 1);
    if (max_blocks) c->max_bpw = max_blocks;
    drop_loop_graphs(c);
    return GAT_OK;
}

GAT_API int32_t gat_set_option(gat_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return fail(c, GAT_ERR_ARG, "null argument");
    const int32_t rc = set_option(c, name, (long long)value);
    if (rc == GAT_OK) drop_loop_graphs(c); // recorded launch sequences bake the geometry in
    return rc;
}

GAT_API int32_t gat_last_launch_info(const gat_ctx *c, gat_launch_info *out, size_t struct_size)
{
    if (!c || !out || struct_size == 0) return GAT_ERR_ARG;
    // the struct grows at its end: a caller built against an older header gets the fields it knows
    std::memcpy(out, &c->last, std::min(struct_size, sizeof(gat_launch_info)));
    return GAT_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Device groups: satellite channels sharded over several devices from ONE host thread.
// Every member is an ordinary context with its own stream; nothing below synchronises unless it says so, so the
// per-device launches of one gat_group_correlate overlap.  No collective: the signal is replicated with peer copies
// (xGMI between the GPUs of one node), outputs are disjoint per channel.
// ---------------------------------------------------------------------------------------------------------------------
struct gat_group {
    std::vector<gat_ctx *> ctx;
    std::vector<gat_channel_params> staging; // host copy of one shard's parameters ([K_r x B]); reused per rank
    std::string err;
};

namespace {
int32_t gfail(gat_group *g, int32_t code, const char *msg)
{
    if (g) g->err = msg;
    return code;
}
void shard_bounds(int total, int n, int r, int *lo, int *cnt)
{
    const int base = total / n, extra = total % n;
    *lo = r * base + std::min(r, extra);
    *cnt = base + (r < extra ? 1 : 0);
}
} // namespace

GAT_API int32_t gat_device_count(int32_t *count)
{
    if (!count) return GAT_ERR_ARG;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(int32_t)e;
    *count = n;
    return GAT_OK;
}

namespace {
// one peer copy on dst's stream behind `filled` (an event recorded on the source stream); *copied (optional) receives an
// event recorded behind the copy on dst's stream
hipError_t peer_copy_after(gat_ctx *dst, void *dst_dev, gat_ctx *src, const void *src_dev, size_t bytes, hipEvent_t filled,
                           hipEvent_t *copied)
{
    hipError_t e = hipSetDevice(dst->device);
    if (e == hipSuccess) e = hipStreamWaitEvent(dst->stream, filled, 0);
    if (e == hipSuccess) {
        if (dst->device == src->device)
            e = hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, dst->stream);
        else
            e = hipMemcpyPeerAsync(dst_dev, dst->device, src_dev, src->device, bytes, dst->stream);
    }
    if (e == hipSuccess && copied && dst->stream != src->stream) {
        e = hipEventCreateWithFlags(copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(*copied, dst->stream);
    }
    return e;
}

// dsts[i] <- src for every i, all copies in flight together: ONE "source is complete" event on the source stream, a
// copy on every destination's stream behind it, then the source stream waits for all of them -- whatever it is given
// next (the following block's ingest overwriting src) runs after the copies have read the buffer.  (Waiting per copy
// would chain them: the second destination's "complete" event would sit behind the wait for the first copy.)
int32_t peer_fanout(gat_ctx *err_ctx, gat_ctx *src, const void *src_dev, size_t n, gat_ctx *const *dsts, void *const *dst_devs,
                    size_t bytes)
{
    hipEvent_t filled = nullptr;
    std::vector<hipEvent_t> copied(n, nullptr);
    GAT_HIP(err_ctx, hipSetDevice(src->device));
    GAT_HIP(err_ctx, hipEventCreateWithFlags(&filled, hipEventDisableTiming));
    hipError_t e = hipEventRecord(filled, src->stream);
    src->wait_seq = 0;
    for (size_t i = 0; i < n && e == hipSuccess; ++i) {
        dsts[i]->wait_seq = 0;
        e = peer_copy_after(dsts[i], dst_devs[i], src, src_dev, bytes, filled, &copied[i]);
    }
    if (e == hipSuccess) e = hipSetDevice(src->device);
    for (size_t i = 0; i < n; ++i) {
        if (!copied[i]) continue;
        if (e == hipSuccess) e = hipStreamWaitEvent(src->stream, copied[i], 0);
        (void)hipEventDestroy(copied[i]); // released once the recorded work has completed
    }
    (void)hipEventDestroy(filled);
    if (e != hipSuccess) return hipfail(err_ctx, e, "peer copy");
    return GAT_OK;
}
} // namespace

GAT_API int32_t gat_memcpy_peer(gat_ctx *dst_ctx, void *dst_dev, gat_ctx *src_ctx, const void *src_dev, size_t bytes)
{
    if (!dst_ctx || !src_ctx || !dst_dev || !src_dev) return fail(dst_ctx, GAT_ERR_ARG, "null argument");
    if (bytes == 0) return GAT_OK;
    // Order, both ways: everything enqueued so far on the SOURCE stream (the upload / generator that fills src) completes
    // before the copy, which runs on the DESTINATION stream (its correlator launches follow in stream order); and whatever
    // the source stream is given AFTER this call (the next block's ingest overwriting src) waits for the copy to have read
    // it -- a streaming receiver refills its ingest buffer every millisecond without a group-wide sync in between.
    return peer_fanout(dst_ctx, src_ctx, src_dev, 1, &dst_ctx, &dst_dev, bytes);
}

GAT_API int32_t gat_group_create(int32_t num_members, const int32_t *devices, gat_group **out)
{
    if (!out || num_members < 1 || num_members > 64) return GAT_ERR_ARG;
    *out = nullptr;
    gat_group *g = new (std::nothrow) gat_group();
    if (!g) return GAT_ERR_NOMEM;
    for (int r = 0; r < num_members; ++r) {
        gat_ctx *c = nullptr;
        const int32_t rc = gat_create(devices ? devices[r] : r, GAT_OWN_STREAM, &c);
        if (rc != GAT_OK) {
            for (gat_ctx *x : g->ctx) (void)gat_destroy(x);
            delete g;
            return rc;
        }
        g->ctx.push_back(c);
    }
    // direct peer access between distinct member devices where the platform offers it (xGMI); without it
    // hipMemcpyPeerAsync still works, staged by the runtime
    for (gat_ctx *a : g->ctx)
        for (gat_ctx *b : g->ctx) {
            if (a->device == b->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a->device, b->device) == hipSuccess && can) {
                (void)hipSetDevice(a->device);
                const hipError_t e = hipDeviceEnablePeerAccess(b->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    (void)hipGetLastError();
    *out = g;
    return GAT_OK;
}

GAT_API int32_t gat_group_destroy(gat_group *g)
{
    if (!g) return GAT_ERR_ARG;
    for (gat_ctx *c : g->ctx) (void)gat_destroy(c);
    delete g;
    return GAT_OK;
}

GAT_API int32_t gat_group_size(const gat_group *g, int32_t *num_members)
{
    if (!g || !num_members) return GAT_ERR_ARG;
    *num_members = (int32_t)g->ctx.size();
    return GAT_OK;
}

GAT_API int32_t gat_group_ctx(gat_group *g, int32_t rank, gat_ctx **ctx)
{
    if (!g || !ctx || rank < 0 || rank >= (int32_t)g->ctx.size()) return gfail(g, GAT_ERR_ARG, "rank outside the group");
    *ctx = g->ctx[(size_t)rank];
    return GAT_OK;
}

GAT_API const char *gat_group_last_error(const gat_group *g)
{
    if (!g) return "null group";
    if (!g->err.empty()) return g->err.c_str();
    for (const gat_ctx *c : g->ctx)
        if (!c->err.empty()) return c->err.c_str();
    return "";
}

GAT_API int32_t gat_group_shard(const gat_group *g, int32_t num_channels, int32_t rank, int32_t *first, int32_t *count)
{
    if (!g || !first || !count || num_channels < 0 || rank < 0 || rank >= (int32_t)g->ctx.size()) return GAT_ERR_ARG;
    int lo, cnt;
    shard_bounds(num_channels, (int)g->ctx.size(), rank, &lo, &cnt);
    *first = lo;
    *count = cnt;
    return GAT_OK;
}

GAT_API int32_t gat_group_set_codes(gat_group *g, const int8_t *codes_host, int32_t code_length, int32_t num_prns)
{
    if (!g) return GAT_ERR_ARG;
    for (gat_ctx *c : g->ctx) {
        const int32_t rc = gat_set_codes(c, codes_host, code_length, num_prns);
        if (rc != GAT_OK) return rc;
    }
    return GAT_OK;
}

GAT_API int32_t gat_group_replicate(gat_group *g, int32_t src_rank, void *const *bufs_dev, size_t bytes)
{
    if (!g || !bufs_dev || src_rank < 0 || src_rank >= (int32_t)g->ctx.size()) return gfail(g, GAT_ERR_ARG, "bad argument");
    std::vector<gat_ctx *> dsts;
    std::vector<void *> ptrs;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        if (!bufs_dev[r]) return gfail(g, GAT_ERR_ARG, "null buffer");
        if ((int32_t)r == src_rank || bufs_dev[r] == bufs_dev[src_rank]) continue;
        dsts.push_back(g->ctx[r]);
        ptrs.push_back(bufs_dev[r]);
    }
    if (dsts.empty() || bytes == 0) return GAT_OK;
    gat_ctx *src = g->ctx[(size_t)src_rank];
    // all peers at once (one link per peer on an xGMI node), the source stream ordered behind all of them
    return peer_fanout(src, src, bufs_dev[src_rank], dsts.size(), dsts.data(), ptrs.data(), bytes);
}

GAT_API int32_t gat_group_correlate(gat_group *g, const gat_signal_desc *signals, const gat_channel_params *params_host,
                                    int32_t B, int32_t K, int32_t L, const int32_t *shifts, double fs,
                                    float *const *out_re_dev, float *const *out_im_dev, uint32_t flags)
{
    if (!g || !signals || !params_host || !shifts || !out_re_dev || !out_im_dev) return gfail(g, GAT_ERR_ARG, "null argument");
    if (B < 1 || K < 1) return gfail(g, GAT_ERR_ARG, "sizes must be positive");
    const int n = (int)g->ctx.size();
    for (int r = 0; r < n; ++r) {
        int lo, cnt;
        shard_bounds(K, n, r, &lo, &cnt);
        if (cnt == 0) continue; // fewer channels than members: this one idles
        if (!out_re_dev[r] || !out_im_dev[r]) return gfail(g, GAT_ERR_ARG, "null output buffer");
        // this member's channels of every block, channel fastest: [cnt x B]
        g->staging.resize((size_t)cnt * B);
        for (int b = 0; b < B; ++b)
            std::memcpy(&g->staging[(size_t)b * cnt], &params_host[(size_t)b * K + lo], (size_t)cnt * sizeof(gat_channel_params));
        gat_ctx *c = g->ctx[(size_t)r];
        // the parameter upload is asynchronous from pageable host memory: the runtime copies it out before returning,
        // so the staging vector may be reused for the next member
        const int32_t rc = gat_downconvert_and_correlate(c, &signals[r], g->staging.data(), B, cnt, L, shifts, fs,
                                                         out_re_dev[r], out_im_dev[r], flags);
        if (rc != GAT_OK) return rc;
    }
    return GAT_OK;
}

GAT_API int32_t gat_group_gather(gat_group *g, float *const *out_re_dev, float *const *out_im_dev, int32_t B, int32_t K,
                                 int32_t L, int32_t M, float *host_re, float *host_im)
{
    if (!g || !out_re_dev || !out_im_dev || !host_re || !host_im) return gfail(g, GAT_ERR_ARG, "null argument");
    if (B < 1 || K < 1 || L < 1 || M < 1) return gfail(g, GAT_ERR_ARG, "sizes must be positive");
    const int n = (int)g->ctx.size();
    const size_t lm = (size_t)L * M;
    std::vector<float> tmp;
    for (int r = 0; r < n; ++r) {
        int lo, cnt;
        shard_bounds(K, n, r, &lo, &cnt);
        if (cnt == 0) continue;
        tmp.resize((size_t)B * cnt * lm);
        for (int comp = 0; comp < 2; ++comp) {
            const float *src = comp ? out_im_dev[r] : out_re_dev[r];
            float *dst = comp ? host_im : host_re;
            const int32_t rc = gat_memcpy_d2h(g->ctx[(size_t)r], tmp.data(), src, tmp.size() * sizeof(float)); // synchronises
            if (rc != GAT_OK) return rc;
            for (int b = 0; b < B; ++b)
                std::memcpy(dst + ((size_t)b * K + lo) * lm, tmp.data() + (size_t)b * cnt * lm, (size_t)cnt * lm * sizeof(float));
        }
    }
    return GAT_OK;
}

GAT_API int32_t gat_group_sync(gat_group *g)
{
    if (!g) return GAT_ERR_ARG;
    for (gat_ctx *c : g->ctx) {
        const int32_t rc = gat_sync(c);
        if (rc != GAT_OK) return rc;
    }
    return GAT_OK;
}

} // extern "C"
