/* gat_latency.c -- single-block latency from native code over the reference's sweep grid.
 *
 * The reference times ONE 1 ms block per call: @benchmark CUDA.@sync kernel_algorithm(...) (src/benchmarks.jl:120-146),
 * grids scripts/run_benchmarks_gpsl1.jl:5-18 (GPS L1: N = 2^11 .. 2^18, M in {1, 4}, L in {3, 7}) and
 * scripts/run_benchmarks_gpsl5.jl:5-18 (GPS L5: N = 2^15 .. 2^18, M in {1, 4}, L = 3); prn 1, 1500 Hz, phases 0,
 * half-chip spacing; BenchmarkTools "Minimum".  This is what a Julia harness calling the shim would see,
 * without a Python / ctypes layer in between:
 *   host   : gat_downconvert_and_correlate (host parameters: validation + 40-byte upload + launch) + gat_sync
 *   dev    : gat_downconvert_and_correlate_dev (parameters already on the device) + gat_sync
 *   graph  : the same call with GAT_FLAG_GRAPH (the launch sequence replayed as one instantiated hipGraph) + gat_sync
 *   enqueue: the call alone, many in a row, one sync at the end (what the device needs per call when the host does not wait)
 *   call   : host time inside the `dev` call itself (planning + the launches), median -- the rest of `dev` is waiting
 *   resident: gat_resident_correlate -- the call rung into a kernel that stays on the device (no launch, no stream wait),
 *             outputs copied to the host included; "wgs" = workgroups of that kernel
 * Output: one line per grid point, minimum / median in microseconds.   build/gat_latency [reps [resident max_workgroups [host_pollers [doorbell]]]]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gat.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int32_t rc_ = (call);                                                                \
        if (rc_ != GAT_OK) {                                                                 \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? gat_last_error(ctx) : ""); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp_d(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

int main(int argc, char **argv)
{
    gat_ctx *ctx = NULL;
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    const unsigned res_wgs = argc > 2 ? (unsigned)atoi(argv[2]) : 0u; /* resident correlator: max_workgroups (0: library default) */
    const unsigned res_pollers = argc > 3 ? (unsigned)atoi(argv[3]) : 0u; /* ... host_pollers (0: library default) */
    const unsigned res_bell = argc > 4 ? (unsigned)atoi(argv[4]) : 0u;    /* ... doorbell: 0 library's choice, 1 pinned host memory, 2 device memory */
    CHECK(gat_create(0, GAT_OWN_STREAM, &ctx));
    double *t = malloc(sizeof(double) * (size_t)reps), *tc = malloc(sizeof(double) * (size_t)reps);
    const int Ms[2] = {1, 4}, Ls[2] = {3, 7};
    /* the two committed grids of the reference */
    const struct { const char *system; int e0, e1, nl; } grids[2] = {{"GPSL1", 11, 18, 2}, {"GPSL5", 15, 18, 1}};
    printf("# one 1 ms block per call, prn 1, 1500 Hz (src/benchmarks.jl:96-99); %d calls per point; microseconds; %s\n", reps, gat_version());
    printf("# %-5s %8s %2s %2s | %-15s | %-15s | %-15s | %s | %s | %s\n", "GNSS", "N", "M", "L", "host min/med", "dev min/med", "graph min/med", "enqueue-only per call", "call med", "resident min/med (wgs)");
    for (int gi = 0; gi < 2; ++gi) {
    int32_t lc = 0;
    double fc = 0.0;
    CHECK(gat_gen_codes(grids[gi].system, 0, NULL, &lc, &fc));
    int8_t *codes = malloc((size_t)lc * 32);
    CHECK(gat_gen_codes(grids[gi].system, 32, codes, &lc, &fc));
    CHECK(gat_set_codes(ctx, codes, lc, 32));
    for (int e = grids[gi].e0; e <= grids[gi].e1; ++e)
        for (int mi = 0; mi < 2; ++mi)
            for (int li = 0; li < grids[gi].nl; ++li) {
                const int N = 1 << e, M = Ms[mi], L = Ls[li];
                const double fs = N / 1e-3;
                int32_t shifts[7];
                CHECK(gat_sample_shifts(L, fs, fc, 0.5, shifts));
                void *re, *im, *prm_dev, *o_re, *o_im;
                CHECK(gat_malloc(ctx, sizeof(float) * (size_t)N * M, &re));
                CHECK(gat_malloc(ctx, sizeof(float) * (size_t)N * M, &im));
                CHECK(gat_malloc(ctx, sizeof(gat_channel_params), &prm_dev));
                CHECK(gat_malloc(ctx, sizeof(float) * M * L, &o_re));
                CHECK(gat_malloc(ctx, sizeof(float) * M * L, &o_im));
                const gat_channel_params p = {0, 0, fc, 1500.0, 0.0, 0.0};
                float h[4 * 7] = {0};
                CHECK(gat_memcpy_h2d(ctx, o_re, h, sizeof(float) * M * L)); /* the prompt printed below is this run's */
                CHECK(gat_memcpy_h2d(ctx, prm_dev, &p, sizeof p));
                CHECK(gat_gen_signal(ctx, re, im, GAT_LAYOUT_PLANAR, N, M, N, N, 1, 1, prm_dev, fs, 1.0));
                const gat_signal_desc sig = {re, im, GAT_LAYOUT_PLANAR, M, N, N, N, 0};
                double res[3][2], call_med = 0.0;
                for (int mode = 0; mode < 3; ++mode) {
                    for (int r = -50; r < reps; ++r) { /* 50 untimed calls first */
                        const double t0 = now_us();
                        if (mode == 0) CHECK(gat_downconvert_and_correlate(ctx, &sig, &p, 1, 1, L, shifts, fs, o_re, o_im, 0));
                        else CHECK(gat_downconvert_and_correlate_dev(ctx, &sig, prm_dev, 1, 1, L, shifts, fs, o_re, o_im, mode == 2 ? GAT_FLAG_GRAPH : 0));
                        const double t1 = now_us();
                        CHECK(gat_sync(ctx));
                        if (r >= 0) t[r] = now_us() - t0, tc[r] = t1 - t0;
                    }
                    qsort(t, (size_t)reps, sizeof(double), cmp_d);
                    if (mode == 1) {
                        qsort(tc, (size_t)reps, sizeof(double), cmp_d);
                        call_med = tc[reps / 2];
                    }
                    res[mode][0] = t[0];
                    res[mode][1] = t[reps / 2];
                }
                CHECK(gat_sync(ctx));
                const double t0 = now_us();
                for (int r = 0; r < reps; ++r) CHECK(gat_downconvert_and_correlate_dev(ctx, &sig, prm_dev, 1, 1, L, shifts, fs, o_re, o_im, 0));
                CHECK(gat_sync(ctx));
                const double per = (now_us() - t0) / reps;
                CHECK(gat_memcpy_d2h(ctx, h, o_re, sizeof(float) * M * L));
                /* the same call through a resident correlator */
                double rmin = 0.0, rmed = 0.0, rprompt = 0.0;
                int rwgs = 0;
                {
                    gat_resident *rs = NULL;
                    const gat_resident_config rcfg = {sizeof(gat_resident_config), 200000, 60000, 0, res_wgs, res_pollers, res_bell};
                    float r_re[4 * 7], r_im[4 * 7];
                    CHECK(gat_sync(ctx));
                    const int32_t orc = gat_resident_open(ctx, &sig, 1, L, shifts, fs, &rcfg, &rs);
                    if (orc == GAT_OK) {
                        for (int r = -50; r < reps; ++r) {
                            const double t0 = now_us();
                            CHECK(gat_resident_correlate(rs, &p, 0, r_re, r_im));
                            if (r >= 0) t[r] = now_us() - t0;
                        }
                        gat_resident_info ri;
                        CHECK(gat_resident_info_get(rs, &ri, sizeof ri));
                        rwgs = ri.workgroups;
                        rprompt = r_re[(L / 2) * M];
                        CHECK(gat_resident_close(rs));
                        qsort(t, (size_t)reps, sizeof(double), cmp_d);
                        rmin = t[0];
                        rmed = t[reps / 2];
                    } else if (orc != GAT_ERR_UNSUPPORTED) { /* (taps wider than one launch's replica: the ordinary call only) */
                        CHECK(orc);
                    }
                }
                printf("  %-5s %8d %2d %2d | %6.2f / %6.2f | %6.2f / %6.2f | %6.2f / %6.2f | %6.2f | %5.2f | %6.2f / %6.2f (%2d)   (prompt %.0f / %.0f)\n", grids[gi].system, N, M, L, res[0][0], res[0][1],
                       res[1][0], res[1][1], res[2][0], res[2][1], per, call_med, rmin, rmed, rwgs, h[(L / 2) * M], rprompt);
                fflush(stdout);
                gat_free(ctx, re); gat_free(ctx, im); gat_free(ctx, prm_dev); gat_free(ctx, o_re); gat_free(ctx, o_im);
            }
    free(codes);
    }
    free(t);
    free(tc);
    gat_destroy(ctx);
    return 0;
}
