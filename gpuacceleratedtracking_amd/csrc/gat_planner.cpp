// gat_planner.cpp -- launch planning of the correlator call (gat::correlate_impl): validation of the signal description,
// kernel selection (vector / matrix-core), tiling, splits, LDS sizing, the launches themselves.
//
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

    // split-bf16 kernel: a consumer lane reads the chip-sign word of its column's (channel slot, tap) -- dword slot * rs + shift
    // + const -- 32 lanes (columns of one tile) per LDS access, 32 banks: choose the row stride rs (odd, in [base, base + 32))
    // under which the fewest of a tile's distinct words share a bank (rows 1 apart at taps -49 / 0 / +49: slot s + 2's early
    // tap sits on slot s's late tap's bank, every read a 2-way conflict)
    auto pick_rep_stride = [&](int base, int tiles, const int *order_) {
        int best_rs = base + 1;
        long long best_cost = -1;
        for (int rs = base + 1; rs < base + 32; rs += 2) {
            long long cost = 0;
            for (int t = 0; t < tiles; ++t) {
                int words[32], nw = 0; // distinct words of the tile
                for (int r = 0; r < 32; r += 2) { // columns (k, l, re / im): the pair reads one word
                    const int col = 32 * t + r, k = col / (2 * L), l = (col - 2 * L * k) >> 1;
                    if (k >= K) break;
                    words[nw++] = k * rs + (shifts[order_[l]] - shifts[order_[0]]);
                }
                for (int i = 0; i < nw; ++i)
                    for (int j = 0; j < i; ++j) cost += ((words[i] - words[j]) & 31) == 0 && words[i] != words[j];
            }
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best_rs = rs;
            }
        }
        return best_rs;
    };

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
        // split-bf16 kernel: columns packed flat (2 L per channel), 32 per tile; rt_max 16-antenna row tiles per workgroup.
        // Column tiles per workgroup: a workgroup of nct tiles costs the same whether its last tiles are live or dead, so
        // the launch costs (groups x nct) tile SLOTS -- at two tiles per workgroup a slot costs w2 x what it costs at four
        // (more sample splitting per column, the 12-wave instance at 64 antennas; profiles/r05/mfma_nct_scan_*.txt:
        // 1.13-1.41 by layout and row tiles).  Three tiles: 4 slots either way -> one workgroup of four (0.55 vs 0.74 ms
        // at 64 antennas x 16 channels of int16); five or six: 6 x w2 slots vs 8 -> mostly two per workgroup (0.62 vs
        // 0.75 ms at 32 x 32 int16, 0.86 vs 0.93 float); nine or more: four.
        const bool int8_in = fmt == GAT_LAYOUT_INTERLEAVED_I8;
        const bool two_term = fmt == GAT_LAYOUT_INTERLEAVED_I16 && c->mc_i16_terms != 3;
        const int tiles_b = (2 * L * K + 31) / 32;
        const int rt_max = (M / 16) % 4 == 0 ? 4 : ((M / 16) % 2 == 0 ? 2 : 1);
        int n_first = tiles_b >= 4 ? 4 : (tiles_b >= 2 ? 2 : 1);
        const double w2 = int8_in ? (rt_max == 4 ? 1.41 : 1.22) : two_term ? (rt_max == 4 ? 1.30 : 1.13) : (rt_max == 4 ? 1.26 : 1.235);
        if (tiles_b >= 3) n_first = ((tiles_b + 1) / 2 * 2) * w2 < (tiles_b + 3) / 4 * 4 ? 2 : 4;
        // kernel choice: 3 / 1 (auto) -> split-bf16 when its tile fits in LDS, 2 -> f32 MFMA.  GAT_MC_AUTO picks a matrix
        // kernel only where it measured faster than the vector kernel (scripts/r05_i16_planner_scan.sh, profiles/r05/
        // mfma_planner_scan_*.txt: N = 50 000, 3 taps, 16-64 antennas x 4-32 channels; round 2's scan: profiles/r02/).  What
        // decides is the share of the launch's tile slots that carry live columns, u = 2 L K / (32 x slots x slot cost): the
        // vector kernel's time grows with K, the matrix kernel's with the slots.  Float samples (three bf16 terms): the
        // matrix kernel wins from u = 0.70 on at four row tiles (64 x 16: 0.76 vs 0.90 ms, 64 x 32: 1.48 vs 1.69; 64 x 24
        // at u = 0.60: 1.48 vs 1.30) and only with nearly full slots at two (32 x 32, u = 0.81: 0.88 vs 0.85; 32 x 64 wins);
        // int16 samples (two exact terms, 5/8 of the MFMAs): from u = 0.5 on (32 x 8: 0.22 vs 0.27 ms, 64 x 12: 0.55 vs
        // 0.74, 64 x 32: 1.12 vs 1.81; a single tile -- 4 channels -- loses); one row tile (M = 16, 48) never wins; tap
        // counts beyond three were not scanned again: round 2's rule (M >= 32, K >= 32, M K >= 2048) stays for them.
        // From int8 pairs (single-term path) it wins from 24 (channel, tap, re/im) columns on at every M.  The f32-MFMA
        // kernel is never chosen by itself any more: the round-2 vector kernel is faster everywhere (configs[4]: 2.45 vs
        // 4.19 ms, 64 antennas x 32 channels: 0.58 vs 0.92 ms); it runs on request (GAT_MC_F32).
        const int slots_first = (tiles_b + n_first - 1) / n_first * n_first;
        const double slot_cost = n_first == 4 ? 1.0 : n_first == 2 ? w2 : 1.8;
        const double util = 2.0 * L * K / (32.0 * slots_first * slot_cost);
        const double util_min = rt_max == 1 ? 2.0 : two_term ? (rt_max == 4 ? 0.50 : 0.53) : (rt_max == 4 ? 0.70 : 0.90);
        const bool round2_rule = M >= 32 && K >= 32 && (long long)M * K >= 2048;
        const bool auto_bf16 = 2ll * L * K >= 24 && (int8_in || (L <= 3 ? util >= util_min : round2_rule));
        const bool auto_f32 = false;
        const bool want_bf16 = c->mc_mode == 3 || (c->mc_mode == 1 && auto_bf16);
        const bool want_f32 = c->mc_mode == 2 || (c->mc_mode == 1 && auto_f32);
        int kind = 0, rt = 1, rep_stride_m = 0;
        int nslots_b = 0, nct_b = 1;
        if (shape_any && want_bf16 && c->d_code_bits && c->d_zeros && N % spv == 0 && spv <= 8) {
            if (c->mc_nct > 0) n_first = c->mc_nct; // A/B runs
            for (int n = n_first; n >= 1 && !kind; n >>= 1) {
                rt = (rt_max == 4 && n == 1) ? 2 : rt_max; // <4,1> does not fit the VGPR budget of a 12-wave workgroup
                const int T = mfma_bf16_tile_samples(rt, n);
                // chip-sign rows: a ring + the copy of its first window (gat_mfma_bf16.hip, s_rep); odd: channel rows land on different banks
                int rs = ((mb_rep_row(rt, T, (int)span) + 31) / 32) * 32 + 1;
                const int ns = mfma_bf16_slots(n, L, K);
                // at most one (slot, sample pair) item per producer thread
                if (ns * T / 2 > mfma_bf16_producer_threads(rt, n) || ns > mfma_bf16_max_slots()) continue;
                const int mode_n = mfma_bf16_mode(rt, n, fmt, c->mc_i16_terms == 3);
                // ... and the taps of neighbouring channels too, where the longer rows still fit
                const int rs_tuned = pick_rep_stride(rs - 1, tiles_b, order);
                if (mfma_bf16_lds_bytes(rt, n, fmt, ns, rs_tuned, c->code_bits_stride, mode_n) <= 160 * 1024) rs = rs_tuned;
                if (mfma_bf16_lds_bytes(rt, n, fmt, ns, rs, c->code_bits_stride, mode_n) <= 160 * 1024) {
                    if (c->mc_mode == 1) {
                        // GAT_MC_AUTO: the kernel's set-up (sign tables and the first replica into LDS, three sample loads
                        // ahead) is paid per workgroup, and a launch is cut into ~2 workgroups per CU: with few steps per
                        // workgroup the vector kernel is faster whatever the shape (ONE 50 MHz block of 64 antennas x 32
                        // int16 channels: 10 steps each, 0.056 vs 0.050 ms; round 4's rule sent it here).  Crossovers:
                        // 10-14 steps with int16 pairs, 25-40 with float samples (profiles/r05/mfma_blocks_scan_*.txt).
                        const long long steps_total = (N + T - 1) / T;
                        const long long groups = (long long)B * (M / (16 * rt)) * ((tiles_b + n - 1) / n);
                        const long long cut = std::min<long long>(steps_total, std::max<long long>(1, (2ll * c->num_cus + groups - 1) / groups));
                        const long long steps_each = (steps_total + cut - 1) / cut;
                        if (steps_each < (two_term || int8_in ? 16 : 32)) break; // -> the vector kernel
                    }
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
            m.rep_ring = kind == 2 ? mb_rep_ring(rt, T, (int)span) : 0;
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
                // int16 samples: two exact bf16 terms per value (5 products per sample), or the float path's three on request
                m.mb_mode = mfma_bf16_mode(rt, nct, fmt, c->mc_i16_terms == 3);
                lds = (unsigned)mfma_bf16_lds_bytes(rt, nct, fmt, m.nslots, m.rep_stride, c->code_bits_stride, m.mb_mode);
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
            c->last.bf16_terms = kind == 2 ? (m.mb_mode == kMbOne ? 1 : m.mb_mode == kMbTwo ? 2 : 3) : 0;
            return GAT_OK;
        }
    }
    c->last.matrix_core = 0;

    // ---- vector kernel (gat_dc.h): launch geometry ---------------------------------------------------------
    // aw: antenna tiles (waves) per workgroup -- 16 antennas on 4 waves walk the same samples, so carrier and replica
    //     are produced once per workgroup; kt: channels a workgroup loops over with the samples held in registers.
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
    // The two-channel 2 x 2 tile (round 5): a four-antenna tile as TWO waves of two antennas on the same samples, each wave
    // looping over TWO channels on its register-resident samples -- half the sample loads per channel of the one-wave-of-four
    // tile at the same registers per wave.  Several channels on one signal were bound by the CU's load path (L2 -> L1 -> registers:
    // 12 channel workgroups re-read every tile), not by HBM or the vector pipe: configs[2] 1.14 -> 1.02 ms, four antennas x
    // eight channels at 20 MHz 0.305 -> 0.253 ms, eight antennas x four channels 0.368 -> 0.291 ms
    // (profiles/r05/ab_aw2_rule.txt).  Not for int8 pairs (eight-sample groups: the two-channel step spills), not for tiles of
    // 16 antennas (AW = 4 with up to four channels per workgroup), not in the latency regime (too few workgroups to matter;
    // + 6 % there), not for one signal per channel.  Option dc_aw2: -1 this rule, 0 never, 1 wherever an instance exists.
    const long long pairs_wgs = (long long)B * ((K + 1) / 2) * (M / 4);
    // (ComplexF32 pairs beyond five taps: the instance drops to two waves per SIMD)
    const bool aw2_rule = fmt != GAT_LAYOUT_INTERLEAVED_I8 && !(fmt == GAT_LAYOUT_INTERLEAVED && max_taps > 5) && K >= 2 && sig->chan_stride == 0 &&
                          pairs_wgs >= 2ll * c->num_cus;
    const bool aw4_tile = (M / MT) % 4 == 0 && c->max_aw >= 4; // sixteen antennas on four waves (with up to four channels per workgroup)
    const bool aw2 = (c->aw2 == 1 || (c->aw2 < 0 && aw2_rule)) && !plan_out && vec == 4 && MT == 4 && !aw4_tile && c->max_kt >= 2 &&
                     c->max_aw >= 2 && K >= 2 && sig->chan_stride == 0 && dc_has_instance(2, max_taps, 4, 2, 2);
    if (aw2) MT = 2;
    const int AT = M / MT;
    int aw = 1, kt = 1;
    if (vec == 4 && MT == 4) aw = AT % 4 == 0 ? 4 : (AT % 2 == 0 ? 2 : 1);
    if (aw2) aw = 2;
    aw = std::min(aw, c->max_aw);
    if (vec == 4 && aw == 4 && sig->chan_stride == 0 && K > 1) kt = K >= 3 ? 4 : 2;
    if (aw2 && sig->chan_stride == 0 && K > 1) kt = 2; // the 2 x 2 tile: two channels on the wave's two antennas (half the loads per channel)
    kt = std::min(kt, c->max_kt);
    if (plan_out) aw = 1, kt = 1;
    while (kt > 1 && !dc_has_instance(MT, max_taps, vec, aw, kt)) kt >>= 1;
    while (aw > 1 && !dc_has_instance(MT, max_taps, vec, aw, kt)) aw >>= 1;
    // LDS: two workgroups per CU at least (80 KB each); a chip table that does not even fit alone is an error
    // chip tables in LDS: int8 rows, or -- long codes (GPS L5: 10 KB per PRN) whose chips are all +-1 -- sign-bit rows (1.3 KB):
    // room for long replica segments and for a second channel's table (option dc_bits: 0 never, 1 long codes, 2 always)
    bool bit_tables = c->d_code_bits && c->code_bits_stride > 0 && (c->bit_tables == 2 || (c->bit_tables == 1 && c->code_row_stride > 2048));
    int tab_bytes = bit_tables ? c->code_bits_stride * 4 : c->code_row_stride;
    auto lds_of = [&](int kt_, int aw_) { return dc_lds_bytes(kt_, MT, tab_bytes, dc_chunk(vec, fmt, aw_)); };
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
    if (!dc_bit_tables(nw)) bit_tables = false, tab_bytes = c->code_row_stride; // (one-wave workgroups read int8 rows)
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
    a.code_bits = bit_tables ? c->d_code_bits : nullptr;
    a.table_stride = tab_bytes;
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
    // replica fill by quads: where two channels share a wave's samples (the 2 x 2 tile) the fill's instructions are on the critical
    // resource; the one-channel tiles measured no gain (option dc_quads: -1 by rule, 0 never, 1 wherever the code rate allows)
    a.fill_quads = dc_fill_quads(aw, kt, nw) && (c->quads >= 0 ? c->quads : 1) ? 1 : 0;
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
    int seg_max = nw == 1 ? c->one_wave_seg : dc_segment_steps((int)chunk, kt, MT);
    if (nw == 4 && c->seg_cap > 0) seg_max = std::max(cfg.depth == 2 ? 2 : 1, std::min(seg_max, c->seg_cap));
    cfg.lds_bytes = (unsigned)dc_lds_bytes(kt, MT, tab_bytes, (int)chunk);

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
            cfg.lds_bytes = (unsigned)dc_lds_bytes_one_wave(a.rep_chan_floats, tab_bytes);
        } else {
            // An instance that holds four waves per SIMD (dc_min_waves) needs four workgroups per CU to get them: with
            // 10 KB chip tables (GPS L5) the full eight-step segment makes a workgroup 47 KB -- three per CU.  Such launches
            // take a segment short enough for 40 KB (configs[2]: six steps; 1.151 -> 1.106 ms together with the two-sample
            // passes that bring the five-tap instance to 128 registers, profiles/r04/r04g_c2_four_waves.txt).
            // (a tap span beyond the default sizing -- seven taps half a chip apart at 262 MHz span 768 samples -- gets the
            // room it needs in the same launch instead of a second launch: 22.8 -> 17 us for that call)
            const int span_sz = std::max(kMaxReplicaSpan, a.rep_span);
            const int want_waves = dc_min_waves(MT, cfg.taps, kt, 1, fmt, aw);
            if (want_waves >= 4 || (aw == 2 && want_waves >= 3))
                while (seg > 2 && dc_lds_bytes_floats(kt, tab_bytes, dc_rep_chan_floats_steps(seg, (int)chunk, span_sz)) > (size_t)(160 / want_waves) * 1024) --seg;
            if (span_sz > kMaxReplicaSpan)
                while (seg > 1 && dc_lds_bytes_floats(kt, tab_bytes, dc_rep_chan_floats_steps(seg, (int)chunk, span_sz)) > 64 * 1024) --seg;
            const int chan_floats = dc_rep_chan_floats_steps(seg, (int)chunk, span_sz);
            if (dc_lds_bytes_floats(kt, tab_bytes, chan_floats) > 160 * 1024)
                return fail(c, GAT_ERR_RANGE, "tap span and code table do not fit the LDS of one workgroup");
            a.rep_chan_floats = chan_floats;
            cfg.lds_bytes = (unsigned)dc_lds_bytes_floats(kt, tab_bytes, chan_floats);
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
    c->last.bf16_terms = 0;
    return GAT_OK;
}



