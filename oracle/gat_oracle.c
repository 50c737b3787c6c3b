/*
 * gat_oracle.c -- CPU restatement of the reference's downconvert + correlate path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gpuacceleratedtracking_amd/ (the product) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * Parity status: PINNED for the operator's mathematical result by the reference's own
 * known-answer literals (test/algorithms.jl:85, :191, :300, :1374, :1513 -> [1476 2500 1476];
 * test/reduction.jl:51-52 -> [N N N]), see tests/test_oracle_golden.py; the C/A code tables by the
 * IS-GPS-200 first-10-chip octals (PRN 1-32); the GPS L5 I5 code tables, every chip of PRN 1-37, by
 * IS-GPS-705 data alone: the initial XB code states of all 37 PRNs (consistent with the XB advances
 * through an independent register, 37 x 13 bits) and the XA short-cycle decode state 1111111111101
 * (reached by an independent register with the ICD's XA polynomial exactly at chip 8190) -- an
 * independent XA xor XB built from those reproduces the whole table (tests/golden/golden.json
 * "l5i_xb_initial_state", "l5_xa_decode_state").  ([1024 2048 1024] for the N = 2048 shape is this
 * build's own derivation, not a reference literal: the reference asserts 1476 there,
 * test/algorithms.jl:1310, a defect.)
 * UNPINNED for: the Tracking.jl CPU call itself (Julia, un-vendored fork, Manifest.toml:1392-1398
 * -- cannot run here) and tap spacing for L > 3 (no fixture exists in the reference).
 *
 * All paths below are relative to /root/reference.  Indices are 0-based here; the reference
 * is 1-based (sample_idx - 1 at src/algorithms.jl:172, :179).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the double-precision code phase
 * must not be fused, src/algorithms.jl:179 evaluates ratio*(n+shift) then + phase).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define GAT_ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * PRN code tables.  The reference takes them from GNSSSignals.jl v0.15.4 (un-vendored,
 * Manifest.toml:440-444; call sites src/benchmarks.jl:43-48, src/gen_signal.jl:64-65), so they
 * are regenerated here from the public interface specs.
 * GPS L1 C/A: IS-GPS-200, G1 = 1+x^3+x^10, G2 = 1+x^2+x^3+x^6+x^8+x^9+x^10, phase selector taps.
 * Chip mapping: logic 0 -> +1, logic 1 -> -1.
 * ---------------------------------------------------------------------------------------- */
static const unsigned char ca_taps[37][2] = {
    {2, 6},  {3, 7},  {4, 8},  {5, 9},  {1, 9},  {2, 10}, {1, 8},  {2, 9},  {3, 10}, {2, 3},
    {3, 4},  {5, 6},  {6, 7},  {7, 8},  {8, 9},  {9, 10}, {1, 4},  {2, 5},  {3, 6},  {4, 7},
    {5, 8},  {6, 9},  {1, 3},  {4, 6},  {5, 7},  {6, 8},  {7, 9},  {8, 10}, {1, 6},  {2, 7},
    {3, 8},  {4, 9},  {5, 10}, {4, 10}, {1, 7},  {2, 8},  {4, 10}};

GAT_ORACLE_API int gat_oracle_code_gpsl1(int prn, int8_t *out /* [1023] */)
{
    if (prn < 1 || prn > 37) return 1;
    int g1[11], g2[11]; /* stages 1..10 */
    for (int i = 1; i <= 10; ++i) g1[i] = g2[i] = 1;
    const int t1 = ca_taps[prn - 1][0], t2 = ca_taps[prn - 1][1];
    for (int c = 0; c < 1023; ++c) {
        const int bit = g1[10] ^ g2[t1] ^ g2[t2];
        out[c] = (int8_t)(1 - 2 * bit);
        const int f1 = g1[3] ^ g1[10];
        const int f2 = g2[2] ^ g2[3] ^ g2[6] ^ g2[8] ^ g2[9] ^ g2[10];
        for (int i = 10; i > 1; --i) { g1[i] = g1[i - 1]; g2[i] = g2[i - 1]; }
        g1[1] = f1;
        g2[1] = f2;
    }
    return 0;
}

/* GPS L5 I5: IS-GPS-705.  XA = 1+x^9+x^10+x^12+x^13 short-cycled to 8190 chips,
 * XB = 1+x+x^3+x^4+x^6+x^7+x^8+x^12+x^13 (period 8191), XB advanced per PRN (Table 3-Ia, I5
 * column).  10230 chips.  Pinned for PRN 1-37 by the ICD's initial XB code states and the XA decode
 * state (see the header). */
static const unsigned short l5i_advance[37] = {
    266,  365,  804,  1138, 1509, 1559, 1756, 2084, 2170, 2303, 2527, 2687, 2930,
    3471, 3940, 4132, 4332, 4924, 5343, 5443, 5641, 5816, 5898, 5918, 5955, 6243,
    6345, 6477, 6518, 6875, 7168, 7187, 7329, 7577, 7720, 7777, 8057};

static int lfsr13_step(unsigned *state, unsigned tapmask)
{
    /* state bit (i-1) = stage i; output = stage 13; feedback = xor of tapped stages */
    const int out = (*state >> 12) & 1u;
    const unsigned fb = (unsigned)__builtin_parity(*state & tapmask);
    *state = ((*state << 1) | fb) & 0x1fffu;
    return out;
}

GAT_ORACLE_API int gat_oracle_code_gpsl5(int prn, int8_t *out /* [10230] */)
{
    if (prn < 1 || prn > 37) return 1;
    const unsigned xa_taps = (1u << 8) | (1u << 9) | (1u << 11) | (1u << 12);
    const unsigned xb_taps = (1u << 0) | (1u << 2) | (1u << 3) | (1u << 5) | (1u << 6) | (1u << 7) |
                             (1u << 11) | (1u << 12);
    unsigned xa = 0x1fffu, xb = 0x1fffu;
    for (int i = 0; i < l5i_advance[prn - 1]; ++i) (void)lfsr13_step(&xb, xb_taps);
    int xa_count = 0;
    for (int c = 0; c < 10230; ++c) {
        const int a = lfsr13_step(&xa, xa_taps);
        const int b = lfsr13_step(&xb, xb_taps);
        out[c] = (int8_t)(1 - 2 * (a ^ b));
        if (++xa_count == 8190) { xa = 0x1fffu; xa_count = 0; }
        /* XB is reset at the 1 ms epoch only, i.e. at c == 10229 -> next call starts fresh */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Correlator tap shifts: get_correlator_sample_shifts(system, correlator, fs, 0.5)
 * (call site src/benchmarks.jl:105-107; implementation lives in the un-vendored Tracking.jl
 * fork).  s = max(1, round(0.5 * fs / fc)), taps (l - L/2) * s (integer division; symmetric
 * for odd L).  L = 3 is pinned by the 1476 known answer (taps -1,0,+1 at fs = 2.5 MHz); L != 3
 * is this build's definition.
 * ---------------------------------------------------------------------------------------- */
GAT_ORACLE_API int gat_oracle_sample_shifts(int L, double fs, double fc, double spacing_chips,
                                            int32_t *shifts)
{
    if (L < 1) return 1;
    long s = lrint(spacing_chips * fs / fc); /* Julia round(Int, x): ties to even */
    if (s < 1) s = 1;
    for (int l = 0; l < L; ++l) shifts[l] = (int32_t)((l - L / 2) * s); /* odd L: symmetric */
    return 0;
}

/* floored modulo, Julia mod() (src/algorithms.jl:182) */
static inline int64_t floormod64(int64_t a, int64_t m)
{
    int64_t r = a % m;
    return r < 0 ? r + m : r;
}

/* code index for sample n (0-based) with tap shift, src/algorithms.jl:179-182:
 *   code_phase = code_frequency / sampling_frequency * ((sample_idx - 1) + shift) + start_code_phase
 *   idx        = mod(floor(Int32, code_phase), code_length)
 * One double division, one double multiply, one double add (file built with -ffp-contract=off). */
static inline int code_index(double ratio, int64_t n_plus_shift, double tau, int Lc)
{
    const double p = ratio * (double)n_plus_shift + tau;
    return (int)floormod64((int64_t)floor(p), Lc);
}

/* ------------------------------------------------------------------------------------------
 * gen_signal restatement, src/gen_signal.jl:64-70 (1-D) / :86-90 (matrix, identical antennas):
 *   code_phases    = fc / fs .* (0:N-1) .+ tau                         (Float64)
 *   carrier_phases = Float32( 2pi * (0:N-1) * f / fs .+ phi_rad )      (Float64 then cast, :88)
 *   re = cos.(carrier_phases) .* code ; im = sin.(carrier_phases) .* code      (Float32)
 * Output planar re/im, column-major [N x M] with leading dimension ld (src/gen_signal.jl:179).
 * ---------------------------------------------------------------------------------------- */
GAT_ORACLE_API int gat_oracle_gen_signal(const int8_t *codes, int Lc, int prn0, double fc,
                                         double fs, double f, double tau, double phi_rad,
                                         int64_t N, int M, int64_t ld, float *re, float *im)
{
    const double ratio = fc / fs;
    const int8_t *c = codes + (size_t)prn0 * (size_t)Lc;
    for (int64_t n = 0; n < N; ++n) {
        const int idx = code_index(ratio, n, tau, Lc);
        const double ph64 = 2.0 * M_PI * (double)n * f / fs + phi_rad;
        const float ph = (float)ph64;
        const float cr = cosf(ph) * (float)c[idx];
        const float ci = sinf(ph) * (float)c[idx];
        for (int m = 0; m < M; ++m) {
            re[n + (size_t)m * (size_t)ld] = cr;
            im[n + (size_t)m * (size_t)ld] = ci;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Stand-alone code replica, gen_code_replica_kernel! (src/algorithms.jl:13-32) with the 5431
 * convention for the earliest tap (src/algorithms.jl:752-758, SURVEY defect D1):
 *   rep[i] = c[ mod(floor(ratio * (i + first_shift) + tau), Lc) ],  i = 0 .. count-1
 * with count = N + (last_shift - first_shift).  correlate reads rep[n + (shift_l - first_shift)].
 * ---------------------------------------------------------------------------------------- */
GAT_ORACLE_API int gat_oracle_gen_code_replica(const int8_t *codes, int Lc, int prn0, double fc,
                                               double fs, double tau, int64_t first_shift,
                                               int64_t count, float *rep)
{
    const double ratio = fc / fs;
    const int8_t *c = codes + (size_t)prn0 * (size_t)Lc;
    for (int64_t i = 0; i < count; ++i) rep[i] = (float)c[code_index(ratio, i + first_shift, tau, Lc)];
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Correlator, FP64 direct evaluation of downconvert_and_correlate_kernel_1330!
 * (src/algorithms.jl:170-187; equation paper/paper.tex:48-52):
 *   carrier = sincos(2pi * ((n * f) / fs + phi_cycles))            :172
 *   dw_re = x_re * c_re + x_im * c_im ; dw_im = x_im * c_re - x_re * c_im     :175-176
 *   R[m,l] += code[idx(n + shift_l)] * dw                          :179-186
 * Signal planar float32, element (n, m) at n + m*ld.  Output double, [M x L] antenna fastest
 * (accum[antenna_idx, corr_idx], src/algorithms.jl:628).  This is THE parity oracle.
 * ---------------------------------------------------------------------------------------- */
GAT_ORACLE_API int gat_oracle_correlate_f64(const float *re, const float *im, int64_t ld, int64_t N,
                                            int M, const int8_t *codes, int Lc, int prn0, double fc,
                                            double fs, double f, double tau, double phi_cycles,
                                            int L, const int32_t *shifts, double *out_re,
                                            double *out_im)
{
    const double ratio = fc / fs;
    const int8_t *c = codes + (size_t)prn0 * (size_t)Lc;
    for (int i = 0; i < M * L; ++i) out_re[i] = out_im[i] = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        const double th = 2.0 * M_PI * ((double)n * f / fs + phi_cycles);
        const double cr = cos(th), ci = sin(th);
        for (int l = 0; l < L; ++l) {
            const double chip = (double)c[code_index(ratio, n + shifts[l], tau, Lc)];
            for (int m = 0; m < M; ++m) {
                const double xr = re[n + (size_t)m * (size_t)ld], xi = im[n + (size_t)m * (size_t)ld];
                out_re[m + l * M] += chip * (xr * cr + xi * ci);
                out_im[m + l * M] += chip * (xi * cr - xr * ci);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline: restatement of the 4-pass structure of Tracking.downconvert_and_correlate!
 * (call site src/benchmarks.jl:63-79; the four operations are listed at paper/paper.tex:209;
 * the correlate inner loop is quoted at paper/paper.tex:286-292):
 *   pass 1 gen_code_replica!  -> code[N + num_of_shifts]   (float32)
 *   pass 2 carrier replica    -> carrier_re/im[N]          (float32)
 *   pass 3 downconvert!       -> dw_re/im[N x M]           (float32)
 *   pass 4 correlate          -> a[m,l] += dw[i,m] * code[i + shift_l - shift_0]
 * Single thread, float32 accumulation, EVERY pass auto-vectorised (the reference uses one Julia
 * thread with LoopVectorization @avx; gcc -O3 -fopt-info-vec reports the loops below, the report
 * of the build is kept under profiles/).  Passes 1-2 run on integer NCOs, as Tracking.jl 0.14's CPU
 * path does (recollection, the fork's source is not available):
 *   pass 1: 32.32 fixed-point code phase, re-anchored with the reference's double expression every
 *           GAT_ORACLE_BATCH samples, table lookup as a vector gather from a table extended past
 *           the code length (no modulo in the loop).  A batch that holds a sample whose fixed-point
 *           fraction is too close to a chip edge to PROVE the same floor as the double expression
 *           is redone with that expression: the replica is bit-identical to gen_code_replica above.
 *   pass 2: 64-bit carrier NCO (one cycle = 2^64), float32 polynomial sincos on the quadrant-
 *           reduced top 32 bits (|err| < 2e-7).
 * scratch must hold (N+nshift) + 2N + 2NM + 2Lc + GAT_ORACLE_BATCH floats (gat_oracle_dc_f32_scratch_floats); N < 2^30.  This is the
 * "port" timed by bench.py's cpu_baseline leg; it is itself checked against correlate_f64.
 * ---------------------------------------------------------------------------------------- */
#define GAT_ORACLE_BATCH 1024

static inline double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* pass 1 */
static void pass_code_replica(const int8_t *c, int Lc, double ratio, double tau, int64_t first, int64_t cnt,
                              float *restrict code, float *restrict ctab /* [2 Lc] */,
                              int32_t *restrict idxbuf /* [BATCH] */)
{
    for (int i = 0; i < Lc; ++i) ctab[i] = ctab[i + Lc] = (float)c[i];
    /* the walk is provable only while a batch advances less than one code period and the margin
     * stays far below one chip (GNSS magnitudes always do); otherwise every sample is exact */
    const double span = fabs(tau) + fabs(ratio) * ((double)cnt + fabs((double)first)) + 1.0;
    const double m = span * 0x1p-18 + (double)(GAT_ORACLE_BATCH + 2); /* 2^-32 chips, as gat_phase.h */
    const int walk_ok = ratio >= 0.0 && ratio * (double)(GAT_ORACLE_BATCH + 1) + 2.0 < (double)Lc && m < 1.0e9 &&
                        span < 2147483648.0;
    const uint64_t rate = walk_ok ? (uint64_t)(ratio * 4294967296.0) : 0u; /* truncated */
    const uint32_t margin = walk_ok ? (uint32_t)m + 1u : 0u;
    for (int64_t i0 = 0; i0 < cnt; i0 += GAT_ORACLE_BATCH) {
        const int len = (int)((cnt - i0) < GAT_ORACLE_BATCH ? (cnt - i0) : GAT_ORACLE_BATCH);
        int exact = !walk_ok;
        if (walk_ok) {
            /* anchor: the reference's expression (src/algorithms.jl:179-182), exact */
            const double p0 = ratio * (double)(i0 + first) + tau;
            const double fl0 = floor(p0);
            const uint64_t q0 = (uint64_t)((p0 - fl0) * 4294967296.0); /* p - floor(p) is exact */
            const uint32_t idx0 = (uint32_t)floormod64((int64_t)fl0, Lc);
            uint32_t amb = 0;
            for (int i = 0; i < len; ++i) { /* 64-bit lanes */
                const uint64_t q = q0 + (uint64_t)i * rate;
                const uint32_t fr = (uint32_t)q;
                idxbuf[i] = (int32_t)(idx0 + (uint32_t)(q >> 32)); /* < 2 Lc */
                amb |= (uint32_t)((fr - margin) > (0xffffffffu - 2u * margin));
            }
            exact = amb != 0;
            if (!exact)
                for (int i = 0; i < len; ++i) code[i0 + i] = ctab[idxbuf[i]]; /* gather */
        }
        if (exact)
            for (int i = 0; i < len; ++i) code[i0 + i] = (float)c[code_index(ratio, i0 + i + first, tau, Lc)];
    }
}

/* pass 2 */
static void pass_carrier_replica(double f, double fs, double phi_cycles, int64_t N, float *restrict car_re,
                                 float *restrict car_im, uint32_t *restrict ubuf /* [BATCH] */)
{
    const double step = f / fs; /* cycles per sample */
    double sfrac = step - floor(step), pfrac = phi_cycles - floor(phi_cycles);
    /* x - floor(x) rounds to exactly 1.0 for a tiny negative x (-1e-20): ldexp(1.0, 64) does not fit a uint64_t and
     * the cast would be undefined -- the phase is then a whole cycle, i.e. 0 */
    if (sfrac >= 1.0) sfrac = 0.0;
    if (pfrac >= 1.0) pfrac = 0.0;
    const uint64_t step64 = (uint64_t)ldexp(sfrac, 64), phi64 = (uint64_t)ldexp(pfrac, 64);
    for (int64_t n0 = 0; n0 < N; n0 += GAT_ORACLE_BATCH) {
        const int len = (int)((N - n0) < GAT_ORACLE_BATCH ? (N - n0) : GAT_ORACLE_BATCH);
        const uint64_t base = phi64 + (uint64_t)n0 * step64; /* wraps with the cycle */
        for (int i = 0; i < len; ++i) ubuf[i] = (uint32_t)((base + (uint64_t)i * step64) >> 32); /* 64-bit lanes */
        for (int i = 0; i < len; ++i) { /* 32-bit lanes */
            const uint32_t u = ubuf[i];                  /* phase, 2^32 = one cycle */
            const uint32_t qd = (u + 0x20000000u) >> 30; /* nearest quadrant, 0..4 */
            const int32_t r = (int32_t)(u - (qd << 30)); /* |r| <= 2^29: +-1/8 cycle */
            const float a = (float)r * 1.4629180792671596e-9f; /* 2 pi / 2^32 */
            const float a2 = a * a;
            const float sp = a * (1.0f + a2 * (-1.6666667e-1f + a2 * (8.3333333e-3f + a2 * (-1.9841270e-4f + a2 * 2.7557319e-6f))));
            const float cp = 1.0f + a2 * (-0.5f + a2 * (4.1666667e-2f + a2 * (-1.3888889e-3f + a2 * 2.4801587e-5f)));
            const uint32_t qi = qd & 3u;
            const float cs = (qi & 1u) ? sp : cp; /* |cos| source */
            const float sn = (qi & 1u) ? cp : sp; /* |sin| source */
            car_re[n0 + i] = (qi == 1u || qi == 2u) ? -cs : cs;
            car_im[n0 + i] = (qi >= 2u) ? -sn : sn;
        }
    }
}

/* pass 4 for one antenna: the reference's loop nest (paper/paper.tex:286-292) -- samples outer, taps
 * inner, so the downconverted sample is loaded once for all taps.  16 independent float partial sums
 * per tap (what a SIMD loop with vector accumulators does), flushed into a double every 1024 samples:
 * a single running float sum of ~N near-constant terms drifts by > 1e-4 relative at N = 20000. */
#define GAT_ORACLE_CORRELATE(LL)                                                                     \
    static void correlate_taps_##LL(const float *restrict dr, const float *restrict di,             \
                                    const float *restrict code, const int32_t *off, int64_t N,      \
                                    double *restrict tr, double *restrict ti)                        \
    {                                                                                                \
        for (int l = 0; l < LL; ++l) tr[l] = ti[l] = 0.0;                                            \
        for (int64_t n0 = 0; n0 < N; n0 += 1024) {                                                   \
            const int len = (int)((N - n0) < 1024 ? (N - n0) : 1024);                                \
            float pr[LL][16], pi[LL][16];                                                            \
            for (int l = 0; l < LL; ++l)                                                             \
                for (int j = 0; j < 16; ++j) pr[l][j] = pi[l][j] = 0.f;                              \
            int n = 0;                                                                               \
            for (; n + 16 <= len; n += 16)                                                           \
                for (int l = 0; l < LL; ++l) {                                                       \
                    const float *cl = code + n0 + n + off[l];                                        \
                    for (int j = 0; j < 16; ++j) {                                                   \
                        pr[l][j] += dr[n0 + n + j] * cl[j];                                          \
                        pi[l][j] += di[n0 + n + j] * cl[j];                                          \
                    }                                                                                \
                }                                                                                    \
            for (; n < len; ++n)                                                                     \
                for (int l = 0; l < LL; ++l) {                                                       \
                    pr[l][0] += dr[n0 + n] * code[n0 + n + off[l]];                                  \
                    pi[l][0] += di[n0 + n] * code[n0 + n + off[l]];                                  \
                }                                                                                    \
            for (int l = 0; l < LL; ++l) {                                                           \
                float sr = 0.f, si = 0.f;                                                            \
                for (int j = 0; j < 16; ++j) { sr += pr[l][j]; si += pi[l][j]; }                     \
                tr[l] += sr;                                                                         \
                ti[l] += si;                                                                         \
            }                                                                                        \
        }                                                                                            \
    }
GAT_ORACLE_CORRELATE(1)
GAT_ORACLE_CORRELATE(2)
GAT_ORACLE_CORRELATE(3)
GAT_ORACLE_CORRELATE(4)
GAT_ORACLE_CORRELATE(5)
GAT_ORACLE_CORRELATE(6)
GAT_ORACLE_CORRELATE(7)
GAT_ORACLE_CORRELATE(8)

static void correlate_taps(int L, const float *dr, const float *di, const float *code, const int32_t *off,
                           int64_t N, double *tr, double *ti)
{
    switch (L) {
    case 1: correlate_taps_1(dr, di, code, off, N, tr, ti); break;
    case 2: correlate_taps_2(dr, di, code, off, N, tr, ti); break;
    case 3: correlate_taps_3(dr, di, code, off, N, tr, ti); break;
    case 4: correlate_taps_4(dr, di, code, off, N, tr, ti); break;
    case 5: correlate_taps_5(dr, di, code, off, N, tr, ti); break;
    case 6: correlate_taps_6(dr, di, code, off, N, tr, ti); break;
    case 7: correlate_taps_7(dr, di, code, off, N, tr, ti); break;
    case 8: correlate_taps_8(dr, di, code, off, N, tr, ti); break;
    default: /* wider tap lists: eight at a time */
        for (int l0 = 0; l0 < L; l0 += 8) {
            const int ll = L - l0 < 8 ? L - l0 : 8;
            correlate_taps(ll, dr, di, code, off + l0, N, tr + l0, ti + l0);
        }
    }
}

#define GAT_ORACLE_MAX_TAPS 64

static int dc_f32_4pass_impl(const float *re, const float *im, int64_t ld, int64_t N, int M, const int8_t *codes,
                             int Lc, int prn0, double fc, double fs, double f, double tau, double phi_cycles,
                             int L, const int32_t *shifts, float *scratch, float *out_re, float *out_im,
                             double *pass_seconds /* [4] accumulated, or NULL */)
{
    if (L < 1 || L > GAT_ORACLE_MAX_TAPS) return 1;
    const int64_t nshift = (int64_t)shifts[L - 1] - shifts[0];
    float *code = scratch;
    float *car_re = code + (N + nshift);
    float *car_im = car_re + N;
    float *dw_re = car_im + N;
    float *dw_im = dw_re + (size_t)N * M;
    float *ctab = dw_im + (size_t)N * M;                         /* 2 Lc floats */
    uint32_t *ibuf = (uint32_t *)(ctab + 2 * (size_t)Lc);        /* GAT_ORACLE_BATCH words */
    const double ratio = fc / fs;
    const int8_t *c = codes + (size_t)prn0 * (size_t)Lc;
    double t0 = pass_seconds ? now_s() : 0.0, t1;

    pass_code_replica(c, Lc, ratio, tau, shifts[0], N + nshift, code, ctab, (int32_t *)ibuf);
    if (pass_seconds) { t1 = now_s(); pass_seconds[0] += t1 - t0; t0 = t1; }
    pass_carrier_replica(f, fs, phi_cycles, N, car_re, car_im, ibuf);
    if (pass_seconds) { t1 = now_s(); pass_seconds[1] += t1 - t0; t0 = t1; }
    /* pass 3: downconvert (conjugate carrier) */
    for (int m = 0; m < M; ++m) {
        const float *restrict xr = re + (size_t)m * (size_t)ld, *restrict xi = im + (size_t)m * (size_t)ld;
        float *restrict dr = dw_re + (size_t)m * N, *restrict di = dw_im + (size_t)m * N;
        const float *restrict cr = car_re, *restrict ci = car_im;
        for (int64_t n = 0; n < N; ++n) {
            dr[n] = xr[n] * cr[n] + xi[n] * ci[n];
            di[n] = xi[n] * cr[n] - xr[n] * ci[n];
        }
    }
    if (pass_seconds) { t1 = now_s(); pass_seconds[2] += t1 - t0; t0 = t1; }
    /* pass 4: correlate */
    int32_t off[GAT_ORACLE_MAX_TAPS];
    double tr[GAT_ORACLE_MAX_TAPS], ti[GAT_ORACLE_MAX_TAPS];
    for (int l = 0; l < L; ++l) off[l] = shifts[l] - shifts[0];
    for (int m = 0; m < M; ++m) {
        correlate_taps(L, dw_re + (size_t)m * N, dw_im + (size_t)m * N, code, off, N, tr, ti);
        for (int l = 0; l < L; ++l) {
            out_re[m + l * M] = (float)tr[l];
            out_im[m + l * M] = (float)ti[l];
        }
    }
    if (pass_seconds) { t1 = now_s(); pass_seconds[3] += t1 - t0; }
    return 0;
}

GAT_ORACLE_API size_t gat_oracle_dc_f32_scratch_floats(int64_t N, int M, int Lc, int64_t nshift)
{
    return (size_t)(N + nshift) + 2 * (size_t)N + 2 * (size_t)N * M + 2 * (size_t)Lc + GAT_ORACLE_BATCH;
}

GAT_ORACLE_API int gat_oracle_dc_f32_4pass(const float *re, const float *im, int64_t ld, int64_t N,
                                           int M, const int8_t *codes, int Lc, int prn0, double fc,
                                           double fs, double f, double tau, double phi_cycles,
                                           int L, const int32_t *shifts, float *scratch,
                                           float *out_re, float *out_im)
{
    return dc_f32_4pass_impl(re, im, ld, N, M, codes, Lc, prn0, fc, fs, f, tau, phi_cycles, L, shifts, scratch, out_re,
                             out_im, NULL);
}

/* Batched CPU baseline over B consecutive blocks and K channels; params arrays are [K x B]
 * (channel fastest).  OpenMP over (block, channel) when built with -fopenmp; threads = 1
 * reproduces the reference's single-thread configuration. */
typedef struct {
    int32_t prn0;
    int32_t pad_;
    double code_freq_hz, carrier_freq_hz, code_phase_chips, carrier_phase_cycles;
} gat_oracle_params;

GAT_ORACLE_API int gat_oracle_dc_f32_batched(const float *re, const float *im, int64_t ant_stride,
                                             int64_t blk_stride, int64_t N, int M, int B, int K,
                                             const int8_t *codes, int Lc,
                                             const gat_oracle_params *prm, double fs, int L,
                                             const int32_t *shifts, int threads, float *out_re,
                                             float *out_im)
{
    const int64_t nshift = (int64_t)shifts[L - 1] - shifts[0];
    const size_t scratch_n = gat_oracle_dc_f32_scratch_floats(N, M, Lc, nshift);
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
#endif
    {
        float *scratch = (float *)malloc(scratch_n * sizeof(float));
#ifdef _OPENMP
#pragma omp for schedule(static) collapse(2)
#endif
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k) {
                const gat_oracle_params *p = &prm[k + (size_t)b * K];
                const size_t o = ((size_t)k + (size_t)b * K) * (size_t)(M * L);
                gat_oracle_dc_f32_4pass(re + (size_t)b * blk_stride, im + (size_t)b * blk_stride,
                                        ant_stride, N, M, codes, Lc, p->prn0, p->code_freq_hz, fs,
                                        p->carrier_freq_hz, p->code_phase_chips,
                                        p->carrier_phase_cycles, L, shifts, scratch, out_re + o,
                                        out_im + o);
            }
        free(scratch);
    }
    (void)threads;
    return rc;
}

/* One thread, B x K calls as above, with the wall time of each pass accumulated into pass_seconds[4]
 * (code replica, carrier replica, downconvert, correlate): what bench.py reports as per_pass_us. */
GAT_ORACLE_API int gat_oracle_dc_f32_profile(const float *re, const float *im, int64_t ant_stride,
                                             int64_t blk_stride, int64_t N, int M, int B, int K,
                                             const int8_t *codes, int Lc, const gat_oracle_params *prm,
                                             double fs, int L, const int32_t *shifts, float *out_re,
                                             float *out_im, double *pass_seconds)
{
    const int64_t nshift = (int64_t)shifts[L - 1] - shifts[0];
    float *scratch = (float *)malloc(gat_oracle_dc_f32_scratch_floats(N, M, Lc, nshift) * sizeof(float));
    if (!scratch) return 2;
    for (int i = 0; i < 4; ++i) pass_seconds[i] = 0.0;
    int rc = 0;
    for (int b = 0; b < B && !rc; ++b)
        for (int k = 0; k < K && !rc; ++k) {
            const gat_oracle_params *p = &prm[k + (size_t)b * K];
            const size_t o = ((size_t)k + (size_t)b * K) * (size_t)(M * L);
            rc = dc_f32_4pass_impl(re + (size_t)b * blk_stride, im + (size_t)b * blk_stride, ant_stride, N, M, codes,
                                   Lc, p->prn0, p->code_freq_hz, fs, p->carrier_freq_hz, p->code_phase_chips,
                                   p->carrier_phase_cycles, L, shifts, scratch, out_re + o, out_im + o, pass_seconds);
        }
    free(scratch);
    return rc;
}

/* The reference's CPU benchmark (src/benchmarks.jl:63-79: @benchmark Tracking.downconvert_and_correlate!(...), one
 * block, one thread, buffers allocated once outside the timed call) timed from C: `reps` calls on the same buffers,
 * the wall time of each in times_ns (BenchmarkTools' samples).  A ctypes call per sample would add ~10 us of Python
 * to blocks that take 5 us. */
GAT_ORACLE_API int gat_oracle_dc_f32_time(const float *re, const float *im, int64_t ld, int64_t N, int M,
                                          const int8_t *codes, int Lc, int prn0, double fc, double fs, double f,
                                          double tau, double phi_cycles, int L, const int32_t *shifts, int reps,
                                          double *times_ns, float *out_re, float *out_im)
{
    const int64_t nshift = (int64_t)shifts[L - 1] - shifts[0];
    float *scratch = (float *)malloc(gat_oracle_dc_f32_scratch_floats(N, M, Lc, nshift) * sizeof(float));
    if (!scratch) return 2;
    int rc = 0;
    for (int r = 0; r < reps && !rc; ++r) {
        const double t0 = now_s();
        rc = dc_f32_4pass_impl(re, im, ld, N, M, codes, Lc, prn0, fc, fs, f, tau, phi_cycles, L, shifts, scratch, out_re,
                               out_im, NULL);
        times_ns[r] = (now_s() - t0) * 1e9;
    }
    free(scratch);
    return rc;
}

/* Same batching for the FP64 oracle (used by parity tests at moderate sizes). */
GAT_ORACLE_API int gat_oracle_correlate_f64_batched(const float *re, const float *im,
                                                    int64_t ant_stride, int64_t blk_stride,
                                                    int64_t chan_stride, int64_t N, int M, int B,
                                                    int K, const int8_t *codes, int Lc,
                                                    const gat_oracle_params *prm, double fs, int L,
                                                    const int32_t *shifts, double *out_re,
                                                    double *out_im)
{
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k) {
            const gat_oracle_params *p = &prm[k + (size_t)b * K];
            const size_t o = ((size_t)k + (size_t)b * K) * (size_t)(M * L);
            const size_t so = (size_t)b * blk_stride + (size_t)k * chan_stride;
            gat_oracle_correlate_f64(re + so, im + so, ant_stride, N, M, codes, Lc, p->prn0,
                                     p->code_freq_hz, fs, p->carrier_freq_hz, p->code_phase_chips,
                                     p->carrier_phase_cycles, L, shifts, out_re + o, out_im + o);
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * reduce_cplx_multi_* restatement (src/reduction.jl:93-160, :331-403, :548-625): sum over the
 * first dimension of a complex [n x M x L] planar array.  The reference's two-pass tree is an
 * implementation detail; the result is the column sum.  Accumulated in double.
 * ---------------------------------------------------------------------------------------- */
GAT_ORACLE_API int gat_oracle_reduce_cplx_multi(const float *in_re, const float *in_im, int64_t n,
                                                int ML, double *out_re, double *out_im)
{
    for (int j = 0; j < ML; ++j) {
        double sr = 0.0, si = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            sr += in_re[i + (size_t)j * (size_t)n];
            si += in_im[i + (size_t)j * (size_t)n];
        }
        out_re[j] = sr;
        out_im[j] = si;
    }
    return 0;
}

GAT_ORACLE_API int gat_oracle_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}
