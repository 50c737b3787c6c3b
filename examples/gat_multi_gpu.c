/* gat_multi_gpu.c -- satellite channels sharded over every GPU of the node from ONE host thread, plain C.
 *
 * What a multi-antenna receiver on an 8-GPU node does (BASELINE configs[3]: 16 antennas, 32 PRNs, 4 per GPU; SURVEY
 * section 8-e): the antenna signal arrives on ONE device (here: synthesised there with gat_gen_signal), is replicated
 * to the peers with peer copies (gat_group_replicate -> hipMemcpyPeerAsync, xGMI), every device correlates its
 * contiguous slice of the PRNs on its own stream (gat_group_correlate, no collective: outputs are disjoint), and the
 * host concatenates the results (gat_group_gather).  The reference is single-device (src/benchmarks.jl:24).
 *
 * Check: the gathered result must equal, BIT FOR BIT, the result of running the same shards one after the other on
 * device 0 alone from device 0's original signal (same launch geometry per shard -> same summation order), and agree
 * to 1e-6 with all channels in one single-device launch.
 *
 * The last line of output is one JSON object (what bench.py's N > 1 line quotes as "group_check"): devices, members,
 * bit_identical, peer_copy_GBps (the signal planes replicated from member 0 to every peer, all peers at once) ...
 *
 *   build/gat_multi_gpu [members [channels_per_member [blocks]]]
 * members defaults to the device count; more members than devices wrap around (members = 2 on a one-GPU box puts two
 * contexts with their own streams on device 0 and exercises every code path, peer copy included).
 *
 * build:  gcc -O2 -Iinclude examples/gat_multi_gpu.c -o build/gat_multi_gpu -Lgpuacceleratedtracking_amd -lgat \
 *             -Wl,-rpath,'$ORIGIN/../gpuacceleratedtracking_amd' -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gat.h"

#define CHECKG(call)                                                                              \
    do {                                                                                          \
        int32_t rc_ = (call);                                                                     \
        if (rc_ != GAT_OK) {                                                                      \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, grp ? gat_group_last_error(grp) : ""); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

enum { MAXG = 64 };

int main(int argc, char **argv)
{
    gat_group *grp = NULL;
    int32_t ndev = 0;
    CHECKG(gat_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    const int G = argc > 1 ? atoi(argv[1]) : ndev;
    const int KPG = argc > 2 ? atoi(argv[2]) : 4;      /* PRNs per member: BASELINE configs[3] */
    const int B = argc > 3 ? atoi(argv[3]) : 8;         /* 1 ms blocks per launch */
    if (G < 1 || G > MAXG || KPG < 1 || B < 1 || G * KPG > 32) { fprintf(stderr, "bad arguments\n"); return 1; }
    enum { N = 50000, M = 16, L = 3 };                  /* 16 antennas, 1 ms @ 50 MHz, E/P/L */
    const int K = G * KPG;
    const double fs = N / 1e-3;
    int32_t devices[MAXG];
    for (int r = 0; r < G; ++r) devices[r] = r % ndev;
    CHECKG(gat_group_create(G, devices, &grp));

    int32_t lc = 0;
    double fc = 0.0;
    CHECKG(gat_gen_codes("GPSL1", 0, NULL, &lc, &fc));
    int8_t *codes = malloc((size_t)lc * 32);
    CHECKG(gat_gen_codes("GPSL1", 32, codes, &lc, &fc));
    CHECKG(gat_group_set_codes(grp, codes, lc, 32));
    int32_t shifts[L];
    CHECKG(gat_sample_shifts(L, fs, fc, 0.5, shifts));

    /* per-(block, channel) parameters of all K channels, channel fastest: a constellation with its own Doppler and code
     * phase per PRN, phases advancing block by block as a tracking loop would hand them over */
    gat_channel_params *prm = malloc(sizeof(*prm) * (size_t)B * K);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k) {
            gat_channel_params *p = &prm[(size_t)b * K + k];
            const double dop = -4000.0 + 250.0 * k, cf = fc * (1.0 + dop / 1575.42e6), f = 1500.0 + dop;
            p->prn = k;
            p->reserved = 0;
            p->code_freq_hz = cf;
            p->carrier_freq_hz = f;
            p->code_phase_chips = fmod(37.25 * k + cf * 1e-3 * b, (double)lc);
            p->carrier_phase_cycles = fmod(0.11 * k + f * 1e-3 * b, 1.0);
        }

    /* buffers on every member's device */
    const size_t plane = sizeof(float) * (size_t)N * B * M;
    void *sig[2][MAXG], *o_re[MAXG], *o_im[MAXG];
    gat_signal_desc desc[MAXG];
    gat_ctx *ctx[MAXG];
    for (int r = 0; r < G; ++r) {
        int32_t lo, cnt;
        CHECKG(gat_group_ctx(grp, r, &ctx[r]));
        CHECKG(gat_group_shard(grp, K, r, &lo, &cnt));
        CHECKG(gat_malloc(ctx[r], plane, &sig[0][r]));
        CHECKG(gat_malloc(ctx[r], plane, &sig[1][r]));
        CHECKG(gat_malloc(ctx[r], sizeof(float) * (size_t)B * cnt * L * M, &o_re[r]));
        CHECKG(gat_malloc(ctx[r], sizeof(float) * (size_t)B * cnt * L * M, &o_im[r]));
        const gat_signal_desc d = {sig[0][r], sig[1][r], GAT_LAYOUT_PLANAR, M, N, (int64_t)N * B, N, 0};
        desc[r] = d;
    }

    /* ingest on member 0: the sum of all K satellites' signals (gen_signal.jl:86-90 per satellite), then peer copies */
    {
        void *prm_dev;
        gat_channel_params *gp = malloc(sizeof(*gp) * (size_t)B * K);
        memcpy(gp, prm, sizeof(*gp) * (size_t)B * K);
        for (size_t i = 0; i < (size_t)B * K; ++i) gp[i].carrier_phase_cycles *= 6.283185307179586; /* radians for gen_signal */
        CHECKG(gat_malloc(ctx[0], sizeof(*gp) * (size_t)B * K, &prm_dev));
        CHECKG(gat_memcpy_h2d(ctx[0], prm_dev, gp, sizeof(*gp) * (size_t)B * K));
        CHECKG(gat_gen_signal(ctx[0], sig[0][0], sig[1][0], GAT_LAYOUT_PLANAR, N, M, (int64_t)N * B, N, B, K, prm_dev, fs, 1.0));
        CHECKG(gat_group_replicate(grp, 0, sig[0], plane)); /* ordered after the generator on member 0's stream */
        CHECKG(gat_group_replicate(grp, 0, sig[1], plane));
        CHECKG(gat_group_sync(grp));
        gat_free(ctx[0], prm_dev);
        free(gp);
    }
    /* the replication, timed: both planes to all G - 1 peers at once (hipMemcpyPeerAsync between distinct devices: xGMI
     * links of the node, one per peer; members that share a device copy inside it) */
    double peer_gbps = 0.0, peer_ms = 0.0;
    int distinct = 0;
    for (int r = 1; r < G; ++r) distinct += devices[r] != devices[0];
    if (G > 1) {
        const int reps = 5;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int rep = 0; rep < reps; ++rep) {
            CHECKG(gat_group_replicate(grp, 0, sig[0], plane));
            CHECKG(gat_group_replicate(grp, 0, sig[1], plane));
            CHECKG(gat_group_sync(grp));
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        peer_ms = dt / reps * 1e3;
        peer_gbps = 2.0 * (double)plane * (G - 1) * reps / dt / 1e9;
    }

    /* the sharded call + gather */
    const size_t out_n = (size_t)B * K * L * M;
    float *g_re = malloc(sizeof(float) * out_n), *g_im = malloc(sizeof(float) * out_n);
    float *s_re = malloc(sizeof(float) * out_n), *s_im = malloc(sizeof(float) * out_n);
    float *a_re = malloc(sizeof(float) * out_n), *a_im = malloc(sizeof(float) * out_n);
    for (int rep = 0; rep < 3; ++rep)
        CHECKG(gat_group_correlate(grp, desc, prm, B, K, L, shifts, fs, (float *const *)o_re, (float *const *)o_im, 0));
    CHECKG(gat_group_gather(grp, (float *const *)o_re, (float *const *)o_im, B, K, L, M, g_re, g_im));

    /* single-device references on member 0 from ITS signal: shard by shard (bit-identical expected), and all at once */
    {
        void *r_re, *r_im;
        gat_channel_params *sp = malloc(sizeof(*sp) * (size_t)B * K);
        float *t = malloc(sizeof(float) * out_n);
        CHECKG(gat_malloc(ctx[0], sizeof(float) * out_n, &r_re));
        CHECKG(gat_malloc(ctx[0], sizeof(float) * out_n, &r_im));
        for (int r = 0; r < G; ++r) {
            int32_t lo, cnt;
            CHECKG(gat_group_shard(grp, K, r, &lo, &cnt));
            if (!cnt) continue;
            for (int b = 0; b < B; ++b) memcpy(&sp[(size_t)b * cnt], &prm[(size_t)b * K + lo], sizeof(*sp) * (size_t)cnt);
            CHECKG(gat_downconvert_and_correlate(ctx[0], &desc[0], sp, B, cnt, L, shifts, fs, r_re, r_im, 0));
            for (int comp = 0; comp < 2; ++comp) {
                CHECKG(gat_memcpy_d2h(ctx[0], t, comp ? r_im : r_re, sizeof(float) * (size_t)B * cnt * L * M));
                float *dst = comp ? s_im : s_re;
                for (int b = 0; b < B; ++b)
                    memcpy(dst + ((size_t)b * K + lo) * L * M, t + (size_t)b * cnt * L * M, sizeof(float) * (size_t)cnt * L * M);
            }
        }
        CHECKG(gat_downconvert_and_correlate(ctx[0], &desc[0], prm, B, K, L, shifts, fs, r_re, r_im, 0));
        CHECKG(gat_memcpy_d2h(ctx[0], a_re, r_re, sizeof(float) * out_n));
        CHECKG(gat_memcpy_d2h(ctx[0], a_im, r_im, sizeof(float) * out_n));
        gat_free(ctx[0], r_re); gat_free(ctx[0], r_im);
        free(sp); free(t);
    }
    const int bit_exact = memcmp(g_re, s_re, sizeof(float) * out_n) == 0 && memcmp(g_im, s_im, sizeof(float) * out_n) == 0;
    double max_rel = 0.0, peak = 0.0, prompt_min = 1e300;
    for (size_t i = 0; i < out_n; ++i) peak = fmax(peak, hypot(a_re[i], a_im[i]));
    for (size_t i = 0; i < out_n; ++i) max_rel = fmax(max_rel, hypot(g_re[i] - a_re[i], g_im[i] - a_im[i]) / peak);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k)
            for (int m = 0; m < M; ++m) {
                const size_t i = (((size_t)b * K + k) * L + 1) * M + m; /* prompt tap */
                prompt_min = fmin(prompt_min, hypot(g_re[i], g_im[i]));
            }

    /* the sharded step, timed: all members launch from this one thread, then one wait */
    double ms = 0.0;
    {
        const int reps = 20;
        float t_ms = 0.f;
        CHECKG(gat_group_sync(grp));
        CHECKG(gat_timer_start(ctx[0]));
        for (int rep = 0; rep < reps; ++rep)
            CHECKG(gat_group_correlate(grp, desc, prm, B, K, L, shifts, fs, (float *const *)o_re, (float *const *)o_im, 0));
        CHECKG(gat_group_sync(grp));
        CHECKG(gat_timer_stop(ctx[0], &t_ms)); /* member 0's stream; the sync above covered the others */
        ms = t_ms / reps;
    }
    printf("members %d on %d device(s), %d PRNs (%d per member), %d antennas, %d blocks of 1 ms @ %.0f MHz: "
           "gather == shard-by-shard on device 0: %s; vs one launch of all channels: max rel diff %.2e; "
           "min |prompt| %.0f of %d; %.3f ms per sharded call (member 0's stream)\n",
           G, ndev, K, KPG, M, B, fs / 1e6, bit_exact ? "BIT-IDENTICAL" : "MISMATCH", max_rel, prompt_min, N, ms);

    const int ok = bit_exact && max_rel <= 1e-6 && prompt_min > 0.5 * N / K; /* every PRN found its signal on every antenna */
    printf("{\"devices\": %d, \"members\": %d, \"peers_on_other_devices\": %d, \"bit_identical\": %s, "
           "\"max_rel_diff_vs_one_launch\": %.3e, \"peer_copy_GBps\": %.2f, \"peer_copy_ms\": %.4f, "
           "\"peer_copy_bytes_per_peer\": %zu, \"ms_per_sharded_call\": %.4f, \"channels\": %d, \"antennas\": %d, "
           "\"blocks\": %d, \"ok\": %s, \"libgat\": \"%s\"}\n",
           ndev, G, distinct, bit_exact ? "true" : "false", max_rel, peer_gbps, peer_ms, 2 * plane, ms, K, M, B,
           ok ? "true" : "false", gat_version());
    for (int r = 0; r < G; ++r) {
        gat_free(ctx[r], sig[0][r]); gat_free(ctx[r], sig[1][r]); gat_free(ctx[r], o_re[r]); gat_free(ctx[r], o_im[r]);
    }
    gat_group_destroy(grp);
    free(codes); free(prm); free(g_re); free(g_im); free(s_re); free(s_im); free(a_re); free(a_im);
    return ok ? 0 : 2;
}
