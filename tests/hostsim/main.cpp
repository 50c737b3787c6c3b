// main.cpp -- drives libgat's C ABI (the real gat_api.cpp + gat_codes.cpp + gat_version.cpp) on the host-only stand-ins of
// this directory, under AddressSanitizer + UndefinedBehaviorSanitizer:
//   1. thousands of random correlate calls (formats, alignments, ragged lengths, strides, tap lists, flags, options) -- every
//      launch the planner emits is checked against the kernel's contract by fake_kernels.cpp; error paths must return
//      GAT_ERR_* codes, never crash;
//   2. the closed loop (eager and graph replay with its LRU), device groups (shard, replicate, correlate, gather), the
//      stand-alone operators, timers, scratch reallocation;
//   3. the resident correlator's host side against a host thread that plays the device: rings at random distances around
//      the kernel's idle limit and call budget, park, code-table change, free, close, destroy with correlators still open --
//      every call must return exactly what the emulated workgroups posted, summed by the host's second stage.
// Exit code 0: no sanitizer report, no broken invariant, no wrong result.   usage: hostsim [calls] [seed]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "gat.h"
#include "gat_internal.h"
#include "hostsim.h"

static int failures = 0;
#define EXPECT(cond, ...)                                   \
    do {                                                    \
        if (!(cond)) {                                      \
            ++failures;                                     \
            std::fprintf(stderr, "FAILED: %s -- ", #cond);  \
            std::fprintf(stderr, __VA_ARGS__);              \
            std::fprintf(stderr, "\n");                     \
        }                                                   \
    } while (0)

static std::mt19937_64 rng;
static long long uni(long long lo, long long hi) { return std::uniform_int_distribution<long long>(lo, hi)(rng); }
static double unif(double lo, double hi) { return std::uniform_real_distribution<double>(lo, hi)(rng); }
template <class T> static T pick(std::initializer_list<T> l) { return *(l.begin() + uni(0, (long long)l.size() - 1)); }

static const int kBytes[4] = {4, 8, 4, 2};   // bytes of one sample in one plane, by layout
static const int kSpv[4] = {4, 2, 4, 8};     // samples of one 16-byte load

int main(int argc, char **argv)
{
    const int calls = argc > 1 ? std::atoi(argv[1]) : 4000;
    rng.seed(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 20260401ull);
    gat_ctx *ctx = nullptr;
    EXPECT(gat_create(0, GAT_OWN_STREAM, &ctx) == GAT_OK, "gat_create");
    std::printf("%s\n", gat_version());

    // correlate before the code table: a state error, not a crash
    {
        gat_signal_desc sig = {(void *)0x100000, (void *)0x200000, GAT_LAYOUT_PLANAR, 1, 1000, 1000, 1000, 0};
        gat_channel_params p = {0, 0, 1.023e6, 0.0, 0.0, 0.0};
        int32_t sh[3] = {-1, 0, 1};
        float o[3];
        EXPECT(gat_downconvert_and_correlate(ctx, &sig, &p, 1, 1, 3, sh, 1e6, o, o, 0) == GAT_ERR_STATE, "correlate without codes");
    }
    int32_t lc = 0;
    double fc = 0.0;
    EXPECT(gat_gen_codes("GPSL1", 0, nullptr, &lc, &fc) == GAT_OK && lc == 1023, "code sizes");
    std::vector<int8_t> codes((size_t)lc * 32);
    EXPECT(gat_gen_codes("GPSL1", 32, codes.data(), &lc, &fc) == GAT_OK, "gen codes");
    EXPECT(gat_set_codes(ctx, codes.data(), lc, 32) == GAT_OK, "set codes");
    int32_t lc5 = 0;
    double fc5 = 0.0;
    gat_gen_codes("GPSL5", 0, nullptr, &lc5, &fc5);
    std::vector<int8_t> codes5((size_t)lc5 * 8);
    EXPECT(gat_gen_codes("GPSL5", 8, codes5.data(), &lc5, &fc5) == GAT_OK, "gen L5 codes");

    // ---- 1. random correlate calls ---------------------------------------------------------------------------------------
    const char *opts[] = {"sync_flag_wgs", "max_ant_tile", "dc_aw", "dc_kt", "dc_bpw", "dc_bpw_force", "dc_wgs_per_cu", "dc_one_wave",
                          "dc_one_wave_min", "dc_ow_seg", "dc_depth", "dc_keep_l2", "dc_align", "dc_aw2", "dc_seg", "dc_bits", "dc_quads", "mc_i16_terms"};
    const long long opt_lo[] = {0, 1, 1, 1, 1, 0, 0, 0, -1, 1, 1, -1, 0, 0, 0, 0, -1, 2}, opt_hi[] = {2048, 4, 4, 4, 64, 8, 16, 1, 64, 8, 2, 1, 1, 1, 8, 2, 1, 3};
    EXPECT(gat_set_option(ctx, "no_such_option", 1) == GAT_ERR_ARG && gat_set_option(ctx, "dc_depth", 7) == GAT_ERR_RANGE, "option errors");
    EXPECT(gat_set_matrix_core(ctx, 7) != GAT_OK, "kernel selection: bad mode");
    long ok_calls = 0, rejected = 0;
    bool l5 = false;
    for (int it = 0; it < calls; ++it) {
        if (it % 97 == 0) { // a dual-frequency receiver alternates tables on one context
            l5 = !l5;
            EXPECT((l5 ? gat_set_codes(ctx, codes5.data(), lc5, 8) : gat_set_codes(ctx, codes.data(), lc, 32)) == GAT_OK, "rebind codes");
        }
        if (it % 13 == 0) EXPECT(gat_set_matrix_core(ctx, (int32_t)uni(0, 3)) == GAT_OK, "kernel selection");
        if (it % 11 == 0) {
            const int o = (int)uni(0, (long long)(sizeof(opts) / sizeof(opts[0])) - 1);
            EXPECT(gat_set_option(ctx, opts[o], uni(opt_lo[o], opt_hi[o])) == GAT_OK, "option %s", opts[o]);
        }
        const int fmt = (int)uni(0, 3), M = (int)pick<long long>({1, 1, 2, 3, 4, 4, 5, 8, 12, 16, 16, 20, 32, 48, 64, 128});
        const int K = (int)pick<long long>({1, 1, 1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 64});
        const int B = M * K >= 256 ? (int)pick<long long>({1, 1, 2, 3}) : (int)pick<long long>({1, 1, 1, 2, 3, 7, 16, 64, 500});
        const int L = (int)pick<long long>({1, 2, 3, 3, 5, 7, 8, 11, 17, 32});
        long long N = pick<long long>({uni(1, 64), uni(64, 5000), 2048, 2500, 4096, 20000, 50000, uni(5000, 300000), 262144});
        if (uni(0, 3) == 0) N -= N % kSpv[fmt];
        if (N < 1) N = kSpv[fmt];
        const long long pad = pick<long long>({0, 0, 0, kSpv[fmt], 1, 3, 32});
        long long bstride = N + pad, astride = bstride * B + pick<long long>({0, 0, kSpv[fmt] * 4, 1});
        const long long cstride = (K > 1 && uni(0, 5) == 0) ? astride * M : 0;
        const uintptr_t mis = pick<long long>({0, 0, 0, 0, 4, 8, 2});
        gat_signal_desc sig = {(void *)(uintptr_t)(0x10000000 + mis), fmt == 0 ? (void *)(uintptr_t)(0x50000000 + mis) : nullptr, fmt, M, N, astride, bstride, cstride};
        std::vector<int32_t> sh(L);
        const int spread = (int)pick<long long>({1, 2, 10, 128, 400, 700, 1500, 5000});
        for (int l = 0; l < L; ++l) sh[l] = (int32_t)uni(-spread, spread);
        if (uni(0, 1)) std::sort(sh.begin(), sh.end());
        const double fs = N / 1e-3;
        const int P = l5 ? 8 : 32, Lc = l5 ? lc5 : lc;
        std::vector<gat_channel_params> prm((size_t)B * K);
        for (auto &p : prm) p = {(int32_t)uni(0, P - 1), 0, (l5 ? fc5 : fc) * (1 + unif(-1e-5, 1e-5)), unif(-5e3, 5e3), unif(0, Lc), unif(0, 1)};
        const int bad = (int)uni(0, 40); // now and then something the validation must catch
        if (bad == 0) prm[0].prn = P + 3;
        if (bad == 1) prm.back().code_phase_chips = NAN;
        if (bad == 2) prm[0].code_freq_hz = -1.0;
        const uint32_t flags = uni(0, 6) == 0 ? GAT_FLAG_ATOMIC : 0u;
        const size_t outs = (size_t)B * K * L * M;
        void *o_re = nullptr, *o_im = nullptr, *prm_dev = nullptr;
        gat_malloc(ctx, outs * sizeof(float), &o_re);
        gat_malloc(ctx, outs * sizeof(float), &o_im);
        int32_t rc;
        if (uni(0, 1)) {
            rc = gat_downconvert_and_correlate(ctx, &sig, prm.data(), B, K, L, sh.data(), fs, (float *)o_re, (float *)o_im, flags);
            EXPECT(rc == GAT_OK || rc == GAT_ERR_RANGE || rc == GAT_ERR_ARG || rc == GAT_ERR_UNSUPPORTED, "host-parameter call: %d (%s)", rc, gat_last_error(ctx));
            EXPECT(bad > 2 || rc != GAT_OK, "a bad record passed the validation (case %d)", bad);
        } else {
            gat_malloc(ctx, prm.size() * sizeof(gat_channel_params), &prm_dev);
            gat_memcpy_h2d(ctx, prm_dev, prm.data(), prm.size() * sizeof(gat_channel_params));
            rc = gat_downconvert_and_correlate_dev(ctx, &sig, (gat_channel_params *)prm_dev, B, K, L, sh.data(), fs, (float *)o_re, (float *)o_im,
                                                   flags | (uni(0, 3) == 0 && !flags ? GAT_FLAG_GRAPH : 0u));
            EXPECT(rc == GAT_OK || rc == GAT_ERR_RANGE || rc == GAT_ERR_ARG || rc == GAT_ERR_UNSUPPORTED, "device-parameter call: %d (%s)", rc, gat_last_error(ctx));
        }
        if (rc == GAT_OK) {
            ++ok_calls;
            gat_launch_info li;
            EXPECT(gat_last_launch_info(ctx, &li, sizeof li) == GAT_OK && li.workgroups > 0 && (li.vec == 4 || li.vec == 1), "launch info");
            const bool aligned = mis % 16 == 0 && (M == 1 || astride % kSpv[fmt] == 0) && (B == 1 || bstride % kSpv[fmt] == 0) && cstride % kSpv[fmt] == 0;
            EXPECT((li.vec == 4) == (aligned && N * kBytes[fmt] < (1ll << 31)), "vector path: vec %d for aligned %d (fmt %d N %lld M %d B %d)", li.vec, (int)aligned, fmt, N, M, B);
        } else {
            ++rejected;
        }
        EXPECT(gat_sync(ctx) == GAT_OK, "sync");
        gat_free(ctx, o_re);
        gat_free(ctx, o_im);
        if (prm_dev) gat_free(ctx, prm_dev);
    }
    std::printf("correlate sweep: %ld calls planned and launched, %ld rejected by validation; %ld vector launches, %ld matrix-core launches, %ld second stages, %ld tails, %ld graphs (%ld replays)\n",
                ok_calls, rejected, hostsim::counters.dc_launches.load(), hostsim::counters.mfma_launches.load(), hostsim::counters.finalize_launches.load(), hostsim::counters.tail_launches.load(),
                hostsim::counters.graphs.load(), hostsim::counters.graph_launches.load());
    EXPECT(ok_calls > calls / 2 && hostsim::counters.tail_launches > 0 && hostsim::counters.finalize_launches > 0 && (calls < 500 || hostsim::counters.mfma_launches > 0),
           "the sweep covers second stages, tails and the matrix-core kernels");
    EXPECT(gat_set_matrix_core(ctx, 1) == GAT_OK, "kernel selection back to auto");
    for (int o = 0; o < 18; ++o) gat_set_option(ctx, opts[o], o == 17 ? 2 : o == 0 ? 1024 : o == 1 ? 4 : o == 2 ? 4 : o == 3 ? 4 : o == 4 ? 16 : o == 5 ? 0 : o == 6 ? 0 : o == 7 ? 1 : o == 8 ? -1 : o == 9 ? 4 : o == 10 ? 2 : o == 11 ? -1 : o == 12 ? 1 : o == 15 ? 1 : o == 16 ? -1 : 0);
    EXPECT(gat_set_codes(ctx, codes.data(), lc, 32) == GAT_OK, "rebind L1");

    // ---- 2. closed loop, stand-alone operators, groups ---------------------------------------------------------------------
    {
        const int K = 6, M = 4, L = 3, N = 20000, NB = 9;
        int32_t sh[3];
        EXPECT(gat_sample_shifts(L, N / 1e-3, fc, 0.5, sh) == GAT_OK && sh[0] < 0 && sh[1] == 0 && sh[2] > 0, "sample shifts");
        gat_signal_desc sig = {(void *)0x10000000, (void *)0x50000000, GAT_LAYOUT_PLANAR, M, N, (long long)N * NB, N, 0};
        gat_loop_config cfg = {1e-3, 18.0, 1.0, fc, 1575.42e6, 0.0, 1.0, lc, L, 0, 1, 2};
        void *state, *pa, *pb, *are, *aim;
        gat_malloc(ctx, sizeof(gat_loop_state) * K, &state);
        gat_memset(ctx, state, 0, sizeof(gat_loop_state) * K);
        gat_malloc(ctx, sizeof(gat_channel_params) * K, &pa);
        gat_malloc(ctx, sizeof(gat_channel_params) * K, &pb);
        std::vector<gat_channel_params> p0(K, gat_channel_params{1, 0, fc, 1000.0, 10.0, 0.0});
        gat_memcpy_h2d(ctx, pa, p0.data(), sizeof(gat_channel_params) * K);
        gat_malloc(ctx, sizeof(float) * NB * K * L * M, &are);
        gat_malloc(ctx, sizeof(float) * NB * K * L * M, &aim);
        int32_t is_b = 0;
        for (int rep = 0; rep < 7; ++rep) { // the same arguments again: graph replay; different block counts: the LRU
            const int nb = rep < 4 ? NB : NB - (rep - 3);
            EXPECT(gat_tracking_run(ctx, &sig, nb, K, L, sh, N / 1e-3, &cfg, (gat_loop_state *)state, (gat_channel_params *)pa, (gat_channel_params *)pb,
                                    (float *)are, (float *)aim, (long long)K * L * M, GAT_FLAG_GRAPH, &is_b) == GAT_OK, "tracking run %d: %s", rep, gat_last_error(ctx));
        }
        EXPECT(hostsim::counters.graph_launches >= 3, "the repeated tracking run replays its graph (%ld replays)", hostsim::counters.graph_launches.load());
        { // the loop's update on the host (needs no device): K channels of the accumulators left in `are` / `aim`
            std::vector<gat_loop_state> hs(K);
            std::vector<gat_channel_params> hc(K, gat_channel_params{1, 0, fc, 1000.0, 10.0, 0.0}), hn(K);
            std::vector<float> hre((size_t)K * L * M, 100.f), him((size_t)K * L * M, -3.f);
            EXPECT(gat_tracking_update_host(hre.data(), him.data(), K, M, &cfg, hs.data(), hc.data(), hn.data()) == GAT_OK && hn[0].code_freq_hz > 0.0, "host loop update");
            EXPECT(gat_tracking_update_host(hre.data(), him.data(), K, M, &cfg, hs.data(), hc.data(), hc.data()) == GAT_OK, "host loop update in place");
            gat_loop_config badc = cfg;
            badc.prompt_index = L;
            EXPECT(gat_tracking_update_host(hre.data(), him.data(), K, M, &badc, hs.data(), hc.data(), hn.data()) == GAT_ERR_RANGE, "host loop update: tap index");
        }
        EXPECT(gat_tracking_update(ctx, (float *)are, (float *)aim, K, M, &cfg, (gat_loop_state *)state, (gat_channel_params *)pa, (gat_channel_params *)pb) == GAT_OK, "tracking update");
        void *rep;
        gat_malloc(ctx, sizeof(float) * (N + 2) * 2, &rep);
        EXPECT(gat_gen_code_replica(ctx, (float *)rep, N + 2, 3, fc, N / 1e-3, 5.5, -1) == GAT_OK, "replica");
        EXPECT(gat_gen_code_replica(ctx, (float *)rep, N + 2, 99, fc, N / 1e-3, 5.5, -1) == GAT_ERR_RANGE, "replica: prn");
        EXPECT(gat_gen_code_replica_multi(ctx, (float *)rep, N + 2, N + 2, 2, (gat_channel_params *)pa, N / 1e-3, -1) == GAT_OK, "replica, two rows");
        EXPECT(gat_reduce_cplx_multi(ctx, (float *)are, (float *)aim, 1000, 6, (float *)are, (float *)aim) == GAT_OK, "reduction");
        float ms = -1.f;
        EXPECT(gat_timer_start(ctx) == GAT_OK && gat_timer_stop(ctx, &ms) == GAT_OK && ms >= 0.f, "timer");
        {
            const int laps = (int)uni(0, 40);
            float iv[64];
            int32_t got = -1;
            for (int i = 0; i < laps; ++i) EXPECT(gat_timer_lap(ctx) == GAT_OK, "lap");
            const int32_t cap = (int32_t)uni(0, 64);
            EXPECT(gat_timer_laps(ctx, iv, cap, &got) == GAT_OK && got == std::min(std::max(laps - 1, 0), (int)cap), "laps: %d intervals of %d laps (capacity %d)", got, laps, cap);
            for (int i = 0; i < got; ++i) EXPECT(iv[i] >= 0.f, "lap interval");
            EXPECT(gat_timer_laps(ctx, iv, 64, &got) == GAT_OK && got == 0, "laps are forgotten once read");
            float each[3] = {-1.f, -1.f, -1.f};
            EXPECT(gat_debug_read_stream(ctx, rep, sizeof(float) * (N + 2) * 2 / 16 * 16, (int32_t)uni(0, 15), 3, each) == GAT_OK && each[2] >= 0.f, "read stream");
            EXPECT(gat_debug_read_stream(ctx, rep, 24, 0, 1, each) == GAT_ERR_ARG && gat_debug_read_stream(ctx, rep, 64, 16, 1, each) == GAT_ERR_RANGE, "read stream: refusals");
            EXPECT(gat_gen_code_replica_texaddr(ctx, (float *)rep, N + 2, 3, fc, N / 1e-3, 5.5, -1, (int32_t)uni(0, 32), (int32_t)uni(-1, 24)) == GAT_OK, "texture addressing study");
            EXPECT(gat_gen_code_replica_texaddr(ctx, (float *)rep, N + 2, 3, fc, N / 1e-3, 5.5, -1, 33, 8) == GAT_ERR_RANGE, "texture addressing study: bits");
        }
        char name[64];
        int32_t ver = 0, cus = 0;
        EXPECT(gat_device_info(ctx, name, sizeof name, &ver, &cus) == GAT_OK && cus == 256, "device info");
        for (void *p : {state, pa, pb, are, aim, rep}) gat_free(ctx, p);
    }
    {
        gat_group *grp = nullptr;
        const int32_t devs[3] = {0, 1, 0};
        EXPECT(gat_group_create(3, devs, &grp) == GAT_OK, "group");
        EXPECT(gat_group_set_codes(grp, codes.data(), lc, 32) == GAT_OK, "group codes");
        const int K = 7, M = 4, L = 3, N = 4000, B = 2;
        int32_t sh[3] = {-2, 0, 2}, first, count, total = 0;
        for (int r = 0; r < 3; ++r) {
            EXPECT(gat_group_shard(grp, K, r, &first, &count) == GAT_OK && first == total, "shard %d", r);
            total += count;
        }
        EXPECT(total == K && gat_group_shard(grp, K, 3, &first, &count) != GAT_OK, "shards cover the channels");
        std::vector<void *> bufs(3), ore(3), oim(3);
        std::vector<gat_signal_desc> sigs(3);
        for (int r = 0; r < 3; ++r) {
            gat_ctx *m = nullptr;
            EXPECT(gat_group_ctx(grp, r, &m) == GAT_OK, "member");
            gat_malloc(m, sizeof(float) * N * B * M * 2, &bufs[r]);
            gat_malloc(m, sizeof(float) * B * 3 * L * M, &ore[r]);
            gat_malloc(m, sizeof(float) * B * 3 * L * M, &oim[r]);
            sigs[r] = {bufs[r], (float *)bufs[r] + (size_t)N * B * M, GAT_LAYOUT_PLANAR, M, N, (long long)N * B, N, 0};
        }
        EXPECT(gat_group_replicate(grp, 0, bufs.data(), sizeof(float) * N * B * M * 2) == GAT_OK, "replicate");
        std::vector<gat_channel_params> prm((size_t)B * K, gat_channel_params{2, 0, fc, 500.0, 1.0, 0.0});
        EXPECT(gat_group_correlate(grp, sigs.data(), prm.data(), B, K, L, sh, N / 1e-3, (float *const *)ore.data(), (float *const *)oim.data(), 0) == GAT_OK,
               "group correlate: %s", gat_group_last_error(grp));
        std::vector<float> hre((size_t)B * K * L * M), him(hre.size());
        EXPECT(gat_group_gather(grp, (float *const *)ore.data(), (float *const *)oim.data(), B, K, L, M, hre.data(), him.data()) == GAT_OK, "gather");
        EXPECT(gat_group_sync(grp) == GAT_OK, "group sync");
        for (int r = 0; r < 3; ++r) {
            gat_ctx *m = nullptr;
            gat_group_ctx(grp, r, &m);
            gat_free(m, bufs[r]); gat_free(m, ore[r]); gat_free(m, oim[r]);
        }
        EXPECT(gat_group_destroy(grp) == GAT_OK, "group destroy");
    }

    // ---- 3. the resident correlator against an emulated device -------------------------------------------------------------
    long res_calls = 0, res_opened = 0, res_refused = 0, res_capacity_refused = 0;
    long open_cus = 0; // workgroups of the correlators left open on ctx (the simulated device holds one resident workgroup per unit)
    gat_ctx *probe_ctx = nullptr; // a context without open correlators: tells how many workgroups a refused geometry has
    EXPECT(gat_create(0, GAT_OWN_STREAM, &probe_ctx) == GAT_OK && gat_set_codes(probe_ctx, codes.data(), lc, 32) == GAT_OK, "probe context");
    for (int it = 0; it < 60; ++it) {
        const int fmt = (int)uni(0, 3), M = (int)pick<long long>({1, 2, 3, 4, 8, 16}), K = (int)pick<long long>({1, 1, 2, 3, 4, 5, 9, 12, 16, 17}), L = (int)pick<long long>({1, 3, 5, 7, 8, 9});
        long long N = pick<long long>({2048, 2500, 4096, 16384, 20000, 65536, 262144, uni(100, 100000)});
        if (uni(0, 4) != 0) N -= N % kSpv[fmt];
        if (N < kSpv[fmt]) N = kSpv[fmt];
        std::vector<int32_t> sh(L);
        const int spread = (int)pick<long long>({1, 8, 300, 1000, 1500});
        for (int l = 0; l < L; ++l) sh[l] = (int32_t)uni(-spread, spread);
        const uintptr_t mis = pick<long long>({0, 0, 0, 0, 8});
        gat_signal_desc sig = {(void *)(uintptr_t)(0x10000000 + mis), fmt == 0 ? (void *)(uintptr_t)(0x50000000 + mis) : nullptr, fmt, M, N, N * 4, N, 0};
        gat_resident_config cfg = {sizeof cfg, (uint32_t)pick<long long>({150, 400, 100000}), (uint32_t)pick<long long>({5, 50, 2000}), (uint32_t)pick<long long>({0, 5, 17}),
                                   (uint32_t)pick<long long>({0, 1, 8, 200}), (uint32_t)pick<long long>({0, 1, 64}), (uint32_t)pick<long long>({0, 1, 2})};
        gat_resident *res = nullptr;
        const gat_resident_config *const cfgp = uni(0, 4) ? &cfg : nullptr;
        const int32_t rc = gat_resident_open(ctx, &sig, K, L, sh.data(), N / 1e-3, cfgp, &res);
        std::vector<int32_t> sorted(sh);
        std::sort(sorted.begin(), sorted.end());
        const bool servable = K <= 16 && L <= 8 && sorted.back() - sorted.front() <= 2048 && N % kSpv[fmt] == 0 && mis == 0;
        if (rc == GAT_ERR_UNSUPPORTED && servable) {
            // refused for want of room: the same geometry opens on a context without open correlators, and its workgroups
            // together with those left open here are more than the device's 256 units hold
            gat_resident *probe = nullptr;
            gat_resident_info pi{};
            const int32_t prc = gat_resident_open(probe_ctx, &sig, K, L, sh.data(), N / 1e-3, cfgp, &probe);
            EXPECT(prc == GAT_OK && gat_resident_info_get(probe, &pi, sizeof pi) == GAT_OK, "capacity refusal: the geometry itself is servable (%d, %s)", prc, gat_last_error(probe_ctx));
            EXPECT(open_cus + pi.workgroups > 256, "refused %d workgroups beside %ld left open: room for 256", pi.workgroups, open_cus);
            if (probe) EXPECT(gat_resident_close(probe) == GAT_OK, "close probe");
            ++res_capacity_refused;
            continue;
        }
        EXPECT((rc == GAT_OK) == servable, "resident open: rc %d for fmt %d M %d K %d L %d N %lld span %d mis %d (%s)", rc, fmt, M, K, L, N, sorted.back() - sorted.front(),
               (int)mis, gat_last_error(ctx));
        if (rc != GAT_OK) {
            EXPECT(rc == GAT_ERR_UNSUPPORTED && res == nullptr, "resident open: refusal code %d", rc);
            ++res_refused;
            continue;
        }
        ++res_opened;
        gat_resident_info info;
        EXPECT(gat_resident_info_get(res, &info, sizeof info) == GAT_OK && info.workgroups >= 1 && info.launches == 1, "resident info");
        // what the emulated workgroups post, summed by the rule of the kernel's grid decode
        const int MT = M % 4 == 0 ? 4 : M % 3 == 0 ? 3 : M % 2 == 0 ? 2 : 1, AG = M / MT, SP = info.splits, KG = K;
        std::vector<int32_t> order(L);
        for (int l = 0; l < L; ++l) order[l] = l;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sh[x] < sh[y]; });
        std::vector<float> want_re((size_t)K * L * M, 0.f), want_im(want_re.size(), 0.f);
        for (int sp = 0; sp < SP; ++sp)
            for (int ag = 0; ag < AG; ++ag)
                for (int kg = 0; kg < KG; ++kg) {
                    const unsigned slot = (unsigned)((ag * SP + sp) * KG + kg);
                    for (int o = 0; o < 2 * MT * L; ++o) {
                        const int ml = o >> 1, m = ag * MT + ml % MT, l = order[ml / MT];
                        ((o & 1) ? want_im : want_re)[((size_t)kg * L + l) * M + m] += hostsim::resident_value(slot, o);
                    }
                }
        EXPECT(info.workgroups == SP * AG * KG, "resident geometry: %d workgroups, %d splits x %d tiles x %d channels", info.workgroups, SP, AG, KG);
        std::vector<gat_channel_params> prm(K, gat_channel_params{4, 0, fc, 1234.0, 17.25, 0.125});
        std::vector<float> r_re(want_re.size()), r_im(want_re.size());
        const int n_calls = (int)pick<long long>({3, 40, 150});
        for (int cidx = 0; cidx < n_calls; ++cidx) {
            if (uni(0, 2) == 0) std::this_thread::sleep_for(std::chrono::microseconds(uni(0, 600)));
            const long long off = uni(0, 3) * N;
            if (uni(0, 30) == 0) { // calls the validation must refuse leave the correlator usable
                prm[0].prn = 77;
                EXPECT(gat_resident_correlate(res, prm.data(), off, r_re.data(), r_im.data()) == GAT_ERR_RANGE, "resident: bad prn");
                prm[0].prn = 4;
                EXPECT(gat_resident_correlate(res, prm.data(), 1, r_re.data(), r_im.data()) != GAT_OK || kSpv[fmt] == 1, "resident: misaligned offset");
            }
            std::fill(r_re.begin(), r_re.end(), -1.f);
            const int32_t rcc = gat_resident_correlate(res, prm.data(), off, r_re.data(), r_im.data());
            EXPECT(rcc == GAT_OK, "resident call %d: %d (%s)", cidx, rcc, gat_last_error(ctx));
            EXPECT(r_re == want_re && r_im == want_im, "resident call %d: results (first %g, want %g)", cidx, r_re[0], want_re[0]);
            ++res_calls;
            if (uni(0, 60) == 0) EXPECT(gat_resident_park(res) == GAT_OK, "park");
            if (uni(0, 90) == 0) { // scratch traffic on the context while a resident kernel is there
                void *p = nullptr;
                gat_malloc(ctx, 4096, &p);
                gat_free(ctx, p);
            }
        }
        int loop_calls = 0;
        if (it % 3 == 0) { // the host-closed loop from native code: {resident call, host update} per block
            gat_loop_config lc_cfg = {};
            lc_cfg.block_seconds = 1e-3; lc_cfg.pll_bandwidth_hz = 18.0; lc_cfg.dll_bandwidth_hz = 1.0; lc_cfg.code_freq_nominal_hz = 1.023e6;
            lc_cfg.carrier_center_hz = 1575.42e6; lc_cfg.early_late_spacing_chips = 1.0; lc_cfg.code_length = lc; lc_cfg.num_taps = L;
            lc_cfg.early_index = 0; lc_cfg.prompt_index = L / 2; lc_cfg.late_index = L - 1;
            std::vector<gat_loop_state> st((size_t)K);
            std::vector<gat_channel_params> lp(prm);
            const int nb = uni(1, 5);
            std::vector<float> a_re(want_re.size() * (size_t)nb, -1.f), a_im(a_re.size(), -1.f);
            const int32_t rl = gat_resident_tracking_run(res, nb, 0, N, &lc_cfg, st.data(), lp.data(), a_re.data(), a_im.data(), (int64_t)want_re.size());
            EXPECT(rl == GAT_OK, "resident tracking run: %d (%s)", rl, gat_last_error(ctx));
            for (int b = 0; b < nb && rl == GAT_OK; ++b)
                EXPECT(std::equal(want_re.begin(), want_re.end(), a_re.begin() + (size_t)b * want_re.size()) &&
                       std::equal(want_im.begin(), want_im.end(), a_im.begin() + (size_t)b * want_im.size()), "resident tracking run: block %d's accumulators", b);
            if (rl == GAT_OK) loop_calls = nb;
            lc_cfg.num_taps = L + 1;
            EXPECT(gat_resident_tracking_run(res, 1, 0, N, &lc_cfg, st.data(), lp.data(), a_re.data(), a_im.data(), 0) == GAT_ERR_ARG, "resident tracking run: tap count");
            res_calls += loop_calls;
        }
        EXPECT(gat_resident_info_get(res, &info, sizeof info) == GAT_OK && info.calls == (uint64_t)(n_calls + loop_calls), "resident: %llu calls counted", (unsigned long long)info.calls);
        if (it % 9 == 4) { // a new code table invalidates it
            EXPECT(gat_set_codes(ctx, codes.data(), lc, 32) == GAT_OK, "rebind");
            EXPECT(gat_resident_correlate(res, prm.data(), 0, r_re.data(), r_im.data()) == GAT_ERR_STATE, "stale correlator");
        }
        EXPECT(open_cus + info.workgroups <= 256, "resident open accepted %d workgroups beside %ld left open", info.workgroups, open_cus);
        if (it % 7 != 3) EXPECT(gat_resident_close(res) == GAT_OK, "close"); // (the others die with the context)
        else open_cus += info.workgroups;
    }
    { // room on the device: every workgroup of every open correlator has to be resident at once
        const long long N = 262144;
        const int32_t sh3[3] = {-10, 0, 10};
        gat_signal_desc sig = {(void *)(uintptr_t)0x10000000, (void *)(uintptr_t)0x50000000, 0, 4, N, N * 4, N, 0};
        gat_resident_config cfg = {sizeof cfg, 100000, 2000, 0, 200, 0, 0};
        gat_resident *a = nullptr, *b = nullptr;
        gat_resident_info ia{};
        EXPECT(gat_resident_open(probe_ctx, &sig, 1, 3, sh3, N / 1e-3, &cfg, &a) == GAT_OK && gat_resident_info_get(a, &ia, sizeof ia) == GAT_OK && ia.workgroups > 128,
               "a correlator of > 128 workgroups (%d)", ia.workgroups);
        EXPECT(gat_resident_open(probe_ctx, &sig, 1, 3, sh3, N / 1e-3, &cfg, &b) == GAT_ERR_UNSUPPORTED && b == nullptr, "a second one beside it must be refused");
        EXPECT(gat_resident_park_all(probe_ctx) == GAT_OK && gat_resident_info_get(a, &ia, sizeof ia) == GAT_OK && ia.running == 0, "park_all");
        EXPECT(gat_resident_close(a) == GAT_OK && gat_resident_open(probe_ctx, &sig, 1, 3, sh3, N / 1e-3, &cfg, &b) == GAT_OK, "room again once the first is closed");
        EXPECT(gat_destroy(probe_ctx) == GAT_OK, "destroy probe context (one correlator still open)");
    }
    std::printf("resident correlator: %ld opened, %ld refused as unsupported (+ %ld for want of room), %ld calls answered; the emulated kernel was started %ld times and served %ld rings\n",
                res_opened, res_refused, res_capacity_refused, res_calls, hostsim::counters.resident_starts.load(), hostsim::counters.resident_calls.load());
    EXPECT(res_opened > 10 && res_refused > 3 && hostsim::counters.resident_starts > res_opened, "the resident sweep restarted kernels");
    EXPECT(gat_destroy(ctx) == GAT_OK, "destroy");
    EXPECT(hostsim::counters.violations == 0, "%ld planner invariants broken", hostsim::counters.violations.load());
    std::printf("%s: %d failures, %ld broken invariants, device allocations %ld / frees %ld\n", failures || hostsim::counters.violations ? "FAILED" : "ok", failures,
                hostsim::counters.violations.load(), hostsim::counters.mallocs.load(), hostsim::counters.frees.load());
    return failures || hostsim::counters.violations ? 1 : 0;
}
