/* sanitize_main.c -- the host-only code of this project under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build;
 * never on the GPU).  Test infrastructure like everything under oracle/: linked against gat_oracle.c compiled WITH the
 * sanitizers and against libgat's host-only translation unit gat_codes.cpp (PRN generators, tap-shift helper -- the
 * part of the product library that needs no HIP runtime).  Exercises the cases that stress index arithmetic:
 *   - the oracle's 4-pass port (fixed-point code walk with its exact-redo margin, table extended past the code length,
 *     64-bit carrier NCO) on ragged sizes, negative tap shifts at n = 0, ratio = 1/16 with code phases within an ulp of
 *     chip edges (every batch lands inside the margin), tiny negative carrier phases (the cast the advisor flagged),
 *     GPS L5 lengths, wide tap spans;
 *   - every oracle result against the FP64 direct evaluation (1e-5) and the replica bit for bit;
 *   - gat_gen_codes for both systems, all PRNs, size queries and error paths; gat_sample_shifts incl. ties and clamps.
 * Exit code 0 = no sanitizer report and all comparisons hold.  `make -C oracle sanitize` builds and runs it. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gat.h"

/* oracle/gat_oracle.c (compiled into this program) */
int gat_oracle_code_gpsl1(int prn, int8_t *out);
int gat_oracle_code_gpsl5(int prn, int8_t *out);
int gat_oracle_sample_shifts(int L, double fs, double fc, double spacing_chips, int32_t *shifts);
int gat_oracle_gen_signal(const int8_t *codes, int Lc, int prn0, double fc, double fs, double f, double tau, double phi_rad,
                          int64_t N, int M, int64_t ld, float *re, float *im);
int gat_oracle_gen_code_replica(const int8_t *codes, int Lc, int prn0, double fc, double fs, double tau, int64_t first_shift,
                                int64_t count, float *rep);
int gat_oracle_correlate_f64(const float *re, const float *im, int64_t ld, int64_t N, int M, const int8_t *codes, int Lc,
                             int prn0, double fc, double fs, double f, double tau, double phi_cycles, int L,
                             const int32_t *shifts, double *out_re, double *out_im);
size_t gat_oracle_dc_f32_scratch_floats(int64_t N, int M, int Lc, int64_t nshift);
int gat_oracle_dc_f32_4pass(const float *re, const float *im, int64_t ld, int64_t N, int M, const int8_t *codes, int Lc,
                            int prn0, double fc, double fs, double f, double tau, double phi_cycles, int L,
                            const int32_t *shifts, float *scratch, float *out_re, float *out_im);
int gat_oracle_reduce_cplx_multi(const float *in_re, const float *in_im, int64_t n, int ML, double *out_re, double *out_im);

static int failures = 0;
#define EXPECT(cond, ...)                         \
    do {                                          \
        if (!(cond)) {                            \
            ++failures;                           \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);         \
            fprintf(stderr, "\n");                \
        }                                         \
    } while (0)

/* one correlator case through the f32 4-pass port and the f64 direct evaluation; exact-size heap buffers, so that any
 * out-of-bounds access is a sanitizer report */
static void dc_case(const char *name, const int8_t *codes, int Lc, int prn0, double fc, double fs, double f, double tau,
                    double phi, int64_t N, int M, int L, const int32_t *shifts, double tol)
{
    const int64_t nshift = (int64_t)shifts[L - 1] - shifts[0];
    float *re = malloc(sizeof(float) * (size_t)N * M), *im = malloc(sizeof(float) * (size_t)N * M);
    gat_oracle_gen_signal(codes, Lc, prn0, fc, fs, f, tau, phi * 6.283185307179586, N, M, N, re, im);
    float *scratch = malloc(sizeof(float) * gat_oracle_dc_f32_scratch_floats(N, M, Lc, nshift));
    float *o_re = malloc(sizeof(float) * (size_t)M * L), *o_im = malloc(sizeof(float) * (size_t)M * L);
    double *r_re = malloc(sizeof(double) * (size_t)M * L), *r_im = malloc(sizeof(double) * (size_t)M * L);
    EXPECT(gat_oracle_dc_f32_4pass(re, im, N, N, M, codes, Lc, prn0, fc, fs, f, tau, phi, L, shifts, scratch, o_re, o_im) == 0, "%s: rc", name);
    gat_oracle_correlate_f64(re, im, N, N, M, codes, Lc, prn0, fc, fs, f, tau, phi, L, shifts, r_re, r_im);
    double peak = 0.0, err = 0.0;
    for (int i = 0; i < M * L; ++i) peak = fmax(peak, hypot(r_re[i], r_im[i]));
    for (int i = 0; i < M * L; ++i) err = fmax(err, hypot(o_re[i] - r_re[i], o_im[i] - r_im[i]) / peak);
    EXPECT(err <= tol, "%s: f32 port vs f64 oracle %.3e", name, err);
    /* the port's replica (first thing in its scratch) is bit-identical to the stand-alone replica */
    float *rep = malloc(sizeof(float) * (size_t)(N + nshift));
    gat_oracle_gen_code_replica(codes, Lc, prn0, fc, fs, tau, shifts[0], N + nshift, rep);
    EXPECT(memcmp(rep, scratch, sizeof(float) * (size_t)(N + nshift)) == 0, "%s: replica differs", name);
    free(rep); free(r_re); free(r_im); free(o_re); free(o_im); free(scratch); free(re); free(im);
}

int main(void)
{
    /* ---- libgat host-only part: code generators and tap shifts ---- */
    int32_t lc = 0;
    double fc = 0.0;
    EXPECT(gat_gen_codes("GPSL1", 0, NULL, &lc, &fc) == GAT_OK && lc == 1023 && fc == 1.023e6, "L1 size query");
    EXPECT(gat_gen_codes("GPSL5", 0, NULL, &lc, &fc) == GAT_OK && lc == 10230 && fc == 10.23e6, "L5 size query");
    EXPECT(gat_gen_codes("GALILEO", 1, NULL, &lc, &fc) != GAT_OK, "unknown system must fail");
    EXPECT(gat_gen_codes(NULL, 1, NULL, &lc, &fc) != GAT_OK, "null system must fail");
    int8_t *l1 = malloc(1023 * 37), *l5 = malloc(10230 * 37), *ref = malloc(10230);
    EXPECT(gat_gen_codes("GPSL1", 37, l1, &lc, &fc) == GAT_OK, "L1 codes");
    EXPECT(gat_gen_codes("GPSL5", 37, l5, &lc, &fc) == GAT_OK, "L5 codes");
    EXPECT(gat_gen_codes("GPSL1", 38, l1, &lc, &fc) != GAT_OK, "38 PRNs must fail");
    for (int p = 1; p <= 37; ++p) { /* libgat's formulation vs the oracle's, chip for chip */
        gat_oracle_code_gpsl1(p, ref);
        EXPECT(memcmp(ref, l1 + (size_t)(p - 1) * 1023, 1023) == 0, "L1 PRN %d", p);
        gat_oracle_code_gpsl5(p, ref);
        EXPECT(memcmp(ref, l5 + (size_t)(p - 1) * 10230, 10230) == 0, "L5 PRN %d", p);
    }
    {
        int32_t a[GAT_MAX_TAPS], b[GAT_MAX_TAPS];
        const double fss[] = {2.5e6, 4e6, 20e6, 50e6, 1.023e6, 3.069e6 /* tie: 1.5 */, 5.115e6 /* tie: 2.5 */, 1e3 /* clamps to 1 */};
        for (size_t i = 0; i < sizeof(fss) / sizeof(*fss); ++i)
            for (int L = 1; L <= GAT_MAX_TAPS; ++L) {
                EXPECT(gat_sample_shifts(L, fss[i], 1.023e6, 0.5, a) == GAT_OK, "shifts rc");
                gat_oracle_sample_shifts(L, fss[i], 1.023e6, 0.5, b);
                EXPECT(memcmp(a, b, sizeof(int32_t) * (size_t)L) == 0, "shifts fs=%g L=%d", fss[i], L);
            }
        EXPECT(gat_sample_shifts(0, 1e6, 1e6, 0.5, a) != GAT_OK && gat_sample_shifts(3, -1.0, 1e6, 0.5, a) != GAT_OK &&
               gat_sample_shifts(3, 1e6, 1e6, 0.5, NULL) != GAT_OK, "shift errors");
    }

    /* ---- the oracle's 4-pass port on index-stressing cases ---- */
    const int32_t t3[3] = {-1, 0, 1}, t3w[3] = {-24, 0, 24}, t5[5] = {-4, -2, 0, 2, 4}, t7[7] = {-30, -20, -10, 0, 10, 20, 30},
                  tneg[3] = {-400, -3, 0}, tpos[2] = {0, 511};
    dc_case("known answer N=2500", l1, 1023, 0, 1.023e6, 2.5e6, 1500.0, 0.0, 0.0, 2500, 1, 3, t3, 1e-5);
    dc_case("ragged N=1", l1, 1023, 3, 1.023e6, 2.5e6, 1500.0, 12.5, 0.3, 1, 2, 3, t3, 1e-5);
    dc_case("ragged N=1023", l1, 1023, 5, 1.023e6, 4e6, -4321.0, 1022.999, 0.9, 1023, 3, 5, t5, 1e-5);
    dc_case("N=1025 batch+1", l1, 1023, 7, 1.023e6, 4e6, 987.0, 511.5, -0.25, 1025, 1, 3, t3, 1e-5);
    dc_case("N=4099 wide taps", l1, 1023, 31, 1.023e6, 50e6, 2.5e6, 1000.25, 0.123, 4099, 2, 3, t3w, 1e-5);
    dc_case("7 taps", l1, 1023, 36, 1.023e6, 20e6, 5000.0, 3.75, 0.5, 20000, 4, 7, t7, 1e-5);
    dc_case("negative-only taps", l1, 1023, 1, 1.023e6, 20e6, 100.0, 0.0, 0.0, 3000, 1, 3, tneg, 1e-5);
    dc_case("one-sided span 511", l1, 1023, 2, 1.023e6, 20e6, 100.0, 1.0, 0.0, 2047, 1, 2, tpos, 1e-5);
    dc_case("L5", l5, 10230, 11, 10.23e6, 50e6, -3000.0, 10229.5, 0.77, 50000, 2, 5, t5, 1e-5);
    dc_case("L5 ragged", l5, 10230, 36, 10.23e6, 32.768e6, 1500.0, 5115.123, 0.0, 32771, 1, 3, t3, 1e-5);
    /* ratio = 1/16 exactly: with tau on or within an ulp of a chip edge every fixed-point fraction sits inside the
     * walk's margin, so each batch takes the exact-redo path; chips must still be bit-identical */
    {
        const double taus[] = {0.0, 1.0, 1022.0, nextafter(1.0, 0.0), nextafter(1.0, 2.0), nextafter(512.0, 0.0), 0.0625,
                               nextafter(0.0625, 1.0), -0.0, nextafter(0.0, -1.0), -1.0, -1023.0, nextafter(-1.0, 0.0)};
        for (size_t i = 0; i < sizeof(taus) / sizeof(*taus); ++i) {
            char name[64];
            snprintf(name, sizeof name, "ratio 1/16 tau[%zu]", i);
            dc_case(name, l1, 1023, 8, 1.023e6, 16.0 * 1.023e6, 250.0, taus[i], 0.0, 5000, 1, 3, t3, 1e-5);
        }
    }
    /* carrier NCO: step / phase that round to a whole cycle (x - floor(x) == 1.0), negative, huge */
    {
        const double fs = 4e6;
        const double fcar[] = {-1e-20 * fs, 1e-20 * fs, -fs, fs, 0.0, -1.25e6, 3.999999e6, 1e9};
        const double phis[] = {-1e-20, 1e-20, -1.0, 1.0, 0.0, -0.75, 123456.789, -98765.4321};
        for (size_t i = 0; i < sizeof(fcar) / sizeof(*fcar); ++i) {
            char name[64];
            snprintf(name, sizeof name, "carrier NCO edge %zu", i);
            dc_case(name, l1, 1023, 9, 1.023e6, fs, fcar[i], 100.5, phis[i], 4000, 1, 3, t3, 2e-5);
        }
    }
    /* column sums */
    {
        enum { n = 2500, ML = 12 };
        float *a = malloc(sizeof(float) * n * ML), *b = malloc(sizeof(float) * n * ML);
        for (int i = 0; i < n * ML; ++i) a[i] = 1.0f, b[i] = 0.0f;
        double sr[ML], si[ML];
        gat_oracle_reduce_cplx_multi(a, b, n, ML, sr, si);
        for (int j = 0; j < ML; ++j) EXPECT(sr[j] == n && si[j] == 0.0, "reduce column %d", j);
        free(a); free(b);
    }
    free(l1); free(l5); free(ref);
    if (failures) {
        fprintf(stderr, "%d comparison(s) failed\n", failures);
        return 1;
    }
    printf("sanitize: ok (ASan + UBSan build, no report; 40+ correlator cases, 74 code rows, 256 tap lists)\n");
    return 0;
}
