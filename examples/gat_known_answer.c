/* gat_known_answer.c -- the C ABI used from plain C (no Python, no torch): reproduces the
 * reference's known answer (test/algorithms.jl:85): GPS L1 C/A PRN 1, N = 2500 samples of 1 ms,
 * f = 1500 Hz, taps (-1, 0, +1)  ->  [1476, 2500, 1476] on every antenna.
 *
 * build:  gcc -O2 -Iinclude examples/gat_known_answer.c -o build/gat_known_answer \
 *             -Lgpuacceleratedtracking_amd -lgat -Wl,-rpath,'$ORIGIN/../gpuacceleratedtracking_amd' -lm
 * This is the call sequence of julia/GATHip.jl in C: output buffers allocated ONCE (GATHip.reserve_outputs!), then per
 * call nothing but gat_downconvert_and_correlate (GATHip.correlate_async!) + read-back (GATHip.fetch_result!) -- the
 * operator is called several times below on the same buffers, as the reference's @benchmark loop would. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gat.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int32_t rc_ = (call);                                                                \
        if (rc_ != GAT_OK) {                                                                 \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? gat_last_error(ctx) : ""); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

int main(void)
{
    gat_ctx *ctx = NULL;
    enum { N = 2500, M = 4, L = 3 };
    const double fs = N / 1e-3, f = 1500.0;

    CHECK(gat_create(0, GAT_OWN_STREAM, &ctx));
    int32_t lc = 0;
    double fc = 0.0;
    CHECK(gat_gen_codes("GPSL1", 0, NULL, &lc, &fc));
    int8_t *codes = malloc((size_t)lc * 32);
    CHECK(gat_gen_codes("GPSL1", 32, codes, &lc, &fc));
    CHECK(gat_set_codes(ctx, codes, lc, 32));

    int32_t shifts[L];
    CHECK(gat_sample_shifts(L, fs, fc, 0.5, shifts));

    /* device buffers */
    void *re, *im, *prm_dev, *out_re, *out_im;
    CHECK(gat_malloc(ctx, sizeof(float) * N * M, &re));
    CHECK(gat_malloc(ctx, sizeof(float) * N * M, &im));
    CHECK(gat_malloc(ctx, sizeof(gat_channel_params), &prm_dev));
    CHECK(gat_malloc(ctx, sizeof(float) * M * L, &out_re));
    CHECK(gat_malloc(ctx, sizeof(float) * M * L, &out_im));

    gat_channel_params p = {0, 0, fc, f, 0.0, 0.0};
    CHECK(gat_memcpy_h2d(ctx, prm_dev, &p, sizeof p));
    /* synthetic signal on the device (gen_signal, src/gen_signal.jl:86-90) */
    CHECK(gat_gen_signal(ctx, re, im, GAT_LAYOUT_PLANAR, N, M, N, N, 1, 1, prm_dev, fs, 1.0));

    gat_signal_desc sig = {re, im, GAT_LAYOUT_PLANAR, M, N, N, N, 0};
    float h_re[M * L], h_im[M * L];
    for (int rep = 0; rep < 5; ++rep) { /* the timed body of the harness: launch + sync, no allocation */
        CHECK(gat_downconvert_and_correlate(ctx, &sig, &p, 1, 1, L, shifts, fs, out_re, out_im, 0));
        CHECK(gat_sync(ctx));
    }
    CHECK(gat_memcpy_d2h(ctx, h_re, out_re, sizeof h_re));
    CHECK(gat_memcpy_d2h(ctx, h_im, out_im, sizeof h_im));

    const float want[L] = {1476.f, 2500.f, 1476.f};
    int bad = 0;
    for (int l = 0; l < L; ++l)
        for (int m = 0; m < M; ++m) {
            const float r = h_re[m + l * M], i = h_im[m + l * M];
            if (fabsf(r - want[l]) > 1e-5f * N || fabsf(i) > 1e-5f * N) ++bad;
        }
    printf("taps %d %d %d -> antenna 0: [%.2f%+.2fj, %.2f%+.2fj, %.2f%+.2fj]  %s\n", shifts[0], shifts[1], shifts[2],
           h_re[0], h_im[0], h_re[M], h_im[M], h_re[2 * M], h_im[2 * M], bad ? "MISMATCH" : "OK (known answer 1476 2500 1476)");

    /* ---- the stand-alone stage operators the reference's tests launch directly, in the order julia/GATHip.jl binds
     * them: downconvert_and_accumulate! (test/algorithms.jl:1438-1514: prompt products == 1, column sums == the known
     * answer), reduce_cplx_multi! (test/reduction.jl:13-52: the two-pass sum of those very products), and
     * gen_code_replica_nsat! (test/algorithms.jl:1199: one replica row per satellite == the single-satellite operator) */
    int bad2 = 0;
    {
        void *acc_re, *acc_im, *sum_re, *sum_im;
        CHECK(gat_malloc(ctx, sizeof(float) * N * M * L, &acc_re));
        CHECK(gat_malloc(ctx, sizeof(float) * N * M * L, &acc_im));
        CHECK(gat_malloc(ctx, sizeof(float) * M * L, &sum_re));
        CHECK(gat_malloc(ctx, sizeof(float) * M * L, &sum_im));
        CHECK(gat_downconvert_and_accumulate(ctx, &sig, &p, L, shifts, fs, NULL, NULL, NULL, NULL, acc_re, acc_im));
        float *prod = malloc(sizeof(float) * N * M * L);
        CHECK(gat_memcpy_d2h(ctx, prod, acc_re, sizeof(float) * N * M * L));
        for (int n = 0; n < N; ++n) /* prompt tap (l = 1), antenna 0: every product is 1 (test/algorithms.jl:1514) */
            if (fabsf(prod[((size_t)1 * M + 0) * N + n] - 1.f) > 1e-5f) ++bad2;
        /* columns = (tap, antenna) pairs, n = N rows each: exactly the layout the products were written in */
        CHECK(gat_reduce_cplx_multi(ctx, acc_re, acc_im, N, M * L, sum_re, sum_im));
        float s_re[M * L];
        CHECK(gat_memcpy_d2h(ctx, s_re, sum_re, sizeof s_re));
        for (int l = 0; l < L; ++l)
            for (int m = 0; m < M; ++m)
                if (fabsf(s_re[m + l * M] - want[l]) > 1e-5f * N) ++bad2;
        /* replicas of two satellites in one launch against the single-satellite operator */
        void *rep2, *rep1, *prm2_dev;
        const int cnt = N + shifts[L - 1] - shifts[0];
        const gat_channel_params p2[2] = {{0, 0, fc, f, 0.0, 0.0}, {6, 0, fc * 1.000002, f, 511.75, 0.0}};
        CHECK(gat_malloc(ctx, sizeof(float) * cnt * 2, &rep2));
        CHECK(gat_malloc(ctx, sizeof(float) * cnt, &rep1));
        CHECK(gat_malloc(ctx, sizeof p2, &prm2_dev));
        CHECK(gat_memcpy_h2d(ctx, prm2_dev, p2, sizeof p2));
        CHECK(gat_gen_code_replica_multi(ctx, rep2, cnt, cnt, 2, prm2_dev, fs, shifts[0]));
        float *r2 = malloc(sizeof(float) * cnt * 2), *r1 = malloc(sizeof(float) * cnt);
        CHECK(gat_memcpy_d2h(ctx, r2, rep2, sizeof(float) * cnt * 2));
        for (int k = 0; k < 2; ++k) {
            CHECK(gat_gen_code_replica(ctx, rep1, cnt, p2[k].prn, p2[k].code_freq_hz, fs, p2[k].code_phase_chips, shifts[0]));
            CHECK(gat_memcpy_d2h(ctx, r1, rep1, sizeof(float) * cnt));
            for (int i = 0; i < cnt; ++i)
                if (r1[i] != r2[(size_t)k * cnt + i]) ++bad2;
        }
        printf("stage operators: accumulate products, two-pass reduction [%.0f %.0f %.0f], two-satellite replica: %s\n", s_re[0],
               s_re[M], s_re[2 * M], bad2 ? "MISMATCH" : "OK");
        free(prod); free(r2); free(r1);
        gat_free(ctx, acc_re); gat_free(ctx, acc_im); gat_free(ctx, sum_re); gat_free(ctx, sum_im);
        gat_free(ctx, rep2); gat_free(ctx, rep1); gat_free(ctx, prm2_dev);
    }
    bad += bad2;

    /* ---- the same call rung into a resident correlator: one kernel stays on the device, a call is a doorbell ring through
     * pinned host memory and the outputs come back on the host (what a receiver loop calling block after block uses) */
    {
        gat_resident *rs = NULL;
        float r_re[M * L], r_im[M * L];
        int bad3 = 0;
        CHECK(gat_sync(ctx)); /* the signal is in device memory before the first ring */
        CHECK(gat_resident_open(ctx, &sig, 1, L, shifts, fs, NULL /* default lifetime limits */, &rs));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(gat_resident_correlate(rs, &p, 0 /* block offset in samples */, r_re, r_im));
            for (int l = 0; l < L; ++l)
                for (int m = 0; m < M; ++m)
                    if (fabsf(r_re[m + l * M] - want[l]) > 1e-5f * N || fabsf(r_im[m + l * M]) > 1e-5f * N) ++bad3;
        }
        gat_resident_info ri;
        CHECK(gat_resident_info_get(rs, &ri, sizeof ri));
        CHECK(gat_resident_close(rs));
        printf("resident correlator: 3 calls, %d workgroup(s), %llu kernel start(s) -> antenna 0: [%.2f, %.2f, %.2f]  %s\n", ri.workgroups,
               (unsigned long long)ri.launches, r_re[0], r_re[M], r_re[2 * M], bad3 ? "MISMATCH" : "OK");
        bad += bad3;
    }
    gat_free(ctx, re); gat_free(ctx, im); gat_free(ctx, prm_dev); gat_free(ctx, out_re); gat_free(ctx, out_im);
    gat_destroy(ctx);
    free(codes);
    return bad ? 2 : 0;
}
