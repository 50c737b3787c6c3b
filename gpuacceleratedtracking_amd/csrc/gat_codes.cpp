// gat_codes.cpp -- host-side PRN code generators behind gat_gen_codes (include/gat.h).
//
// Stand-in for GNSSSignals.GPSL1() / GPSL5() `system.codes` (reference: src/benchmarks.jl:93,
// src/GPUAcceleratedTracking.jl:39-42; GNSSSignals.jl itself is an un-vendored dependency).
// Written from the public interface specifications, in a different formulation from the test
// oracle (which uses the G2 tap-selector form) so that the two cross-check:
//   GPS L1 C/A (IS-GPS-200): chip_i = G1_i xor G2_{i - delay(prn)}, 10-bit Fibonacci LFSRs.
//   GPS L5 I5  (IS-GPS-705): chip_i = XA_i xor XB_{i + advance(prn)}, 13-bit LFSRs, XA
//                            short-cycled after 8190 chips, both restarted every 10230 chips.
// Chip mapping logic 0 -> +1, logic 1 -> -1.  Output column-major [code_length x num_prns].
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

#include "gat.h"
#include "gat_loop.h"

namespace {

// Generic Fibonacci LFSR over `bits` stages, all-ones start, output = last stage.
// `poly` lists the feedback stages (1-based).
std::vector<uint8_t> lfsr_sequence(int bits, std::initializer_list<int> poly, int length)
{
    uint32_t fbmask = 0;
    for (int t : poly) fbmask |= 1u << (t - 1);
    const uint32_t all = (1u << bits) - 1u;
    uint32_t reg = all;
    std::vector<uint8_t> seq((size_t)length);
    for (int i = 0; i < length; ++i) {
        seq[(size_t)i] = (uint8_t)((reg >> (bits - 1)) & 1u);
        const uint32_t fb = (uint32_t)__builtin_popcount(reg & fbmask) & 1u;
        reg = ((reg << 1) | fb) & all;
    }
    return seq;
}

const int kCaDelay[37] = {5,   6,   7,   8,   17,  18,  139, 140, 141, 251, 252, 254, 255,
                          256, 257, 258, 469, 470, 471, 472, 473, 474, 509, 512, 513, 514,
                          515, 516, 859, 860, 861, 862, 863, 950, 947, 948, 950};

const int kL5iAdvance[37] = {266,  365,  804,  1138, 1509, 1559, 1756, 2084, 2170, 2303,
                             2527, 2687, 2930, 3471, 3940, 4132, 4332, 4924, 5343, 5443,
                             5641, 5816, 5898, 5918, 5955, 6243, 6345, 6477, 6518, 6875,
                             7168, 7187, 7329, 7577, 7720, 7777, 8057};

void gen_gpsl1(int prn, int8_t *out)
{
    static const std::vector<uint8_t> g1 = lfsr_sequence(10, {3, 10}, 1023);
    static const std::vector<uint8_t> g2 = lfsr_sequence(10, {2, 3, 6, 8, 9, 10}, 1023);
    const int d = kCaDelay[prn - 1];
    for (int i = 0; i < 1023; ++i) {
        const int bit = g1[(size_t)i] ^ g2[(size_t)((i - d + 1023) % 1023)];
        out[i] = (int8_t)(bit ? -1 : 1);
    }
}

void gen_gpsl5(int prn, int8_t *out)
{
    static const std::vector<uint8_t> xa = lfsr_sequence(13, {9, 10, 12, 13}, 8190);
    static const std::vector<uint8_t> xb = lfsr_sequence(13, {1, 3, 4, 6, 7, 8, 12, 13}, 8191);
    const int adv = kL5iAdvance[prn - 1];
    for (int i = 0; i < 10230; ++i) {
        const int a = xa[(size_t)(i % 8190)];
        const int b = xb[(size_t)((i + adv) % 8191)];
        out[i] = (int8_t)((a ^ b) ? -1 : 1);
    }
}

} // namespace

extern "C" GAT_API int32_t gat_gen_codes(const char *system, int32_t num_prns, int8_t *out,
                                         int32_t *code_length, double *code_freq_hz)
{
    if (!system) return GAT_ERR_ARG;
    int lc = 0;
    double fc = 0.0;
    void (*gen)(int, int8_t *) = nullptr;
    if (std::strcmp(system, "GPSL1") == 0) {
        lc = 1023;
        fc = 1.023e6;
        gen = gen_gpsl1;
    } else if (std::strcmp(system, "GPSL5") == 0) {
        lc = 10230;
        fc = 10.23e6;
        gen = gen_gpsl5;
    } else {
        return GAT_ERR_UNSUPPORTED;
    }
    if (code_length) *code_length = lc;
    if (code_freq_hz) *code_freq_hz = fc;
    if (!out) return GAT_OK;
    if (num_prns < 1 || num_prns > 37) return GAT_ERR_RANGE;
    for (int p = 1; p <= num_prns; ++p) gen(p, out + (size_t)(p - 1) * (size_t)lc);
    return GAT_OK;
}

// get_correlator_sample_shifts(system, correlator, fs, spacing) (src/benchmarks.jl:105-107; the implementation lives in the
// un-vendored Tracking.jl fork): s = max(1, round(spacing * fs / fc)), taps (l - L / 2) * s.  Host-only like the generators
// above, so that this translation unit can be built and run under the CPU sanitizers (tests/test_host_sanitizers.py).
extern "C" GAT_API int32_t gat_sample_shifts(int32_t L, double fs, double fc, double spacing, int32_t *shifts)
{
    if (!shifts || L < 1 || L > GAT_MAX_TAPS || !(fs > 0.0) || !(fc > 0.0) || !(spacing == spacing)) return GAT_ERR_ARG;
    const double x = std::nearbyint(spacing * fs / fc); // round-half-even like Julia round(Int, x)
    long long s = x >= 1.0 ? (x < 1.0e9 ? (long long)x : 1000000000ll) : 1;
    for (int l = 0; l < L; ++l) shifts[l] = (int32_t)((l - L / 2) * s);
    return GAT_OK;
}

// The closed loop's update on the HOST: what gat_tracking_update does on the device (the same text, gat_loop.h), for a
// receiver that has its correlator outputs on the host -- from a resident correlator -- and closes its loops there, as
// Tracking.jl does.  next_host may alias cur_host.
extern "C" GAT_API int32_t gat_tracking_update_host(const float *acc_re_host, const float *acc_im_host, int32_t num_channels, int32_t num_ants,
                                                    const gat_loop_config *config, gat_loop_state *state_host, const gat_channel_params *cur_host,
                                                    gat_channel_params *next_host)
{
    if (!acc_re_host || !acc_im_host || !config || !state_host || !cur_host || !next_host) return GAT_ERR_ARG;
    if (num_channels < 1 || num_ants < 1) return GAT_ERR_ARG;
    const int L = config->num_taps;
    if (L < 1 || L > GAT_MAX_TAPS || config->early_index < 0 || config->early_index >= L || config->prompt_index < 0 || config->prompt_index >= L ||
        config->late_index < 0 || config->late_index >= L || config->code_length < 1 || !(config->block_seconds > 0.0))
        return GAT_ERR_RANGE;
    for (int k = 0; k < num_channels; ++k) {
        gat_channel_params n;
        gat::loop_update_channel(acc_re_host, acc_im_host, k, num_ants, *config, state_host[k], cur_host[k], n);
        next_host[k] = n;
    }
    return GAT_OK;
}
