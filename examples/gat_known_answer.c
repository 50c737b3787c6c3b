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
    gat_free(ctx, re); gat_free(ctx, im); gat_free(ctx, prm_dev); gat_free(ctx, out_re); gat_free(ctx, out_im);
    gat_destroy(ctx);
    free(codes);
    return bad ? 2 : 0;
}
