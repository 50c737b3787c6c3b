// gat_loop.h -- the closed tracking loop's per-channel update: Costas-arctan PLL discriminator + 3rd-order bilinear filter,
// normalised early-minus-late DLL + 2nd-order bilinear filter, carrier aiding, NCO propagation -- written once for the device
// kernel (tracking_update_kernel, gat_kernels.hip) and for the host entry point gat_tracking_update_host (gat_codes.cpp: a
// receiver that takes its correlator outputs on the host, e.g. from a resident correlator, closes its loops there as
// Tracking.jl does).  Double precision throughout; compiled without contraction in both builds.
#pragma once

#include <math.h>

#include "gat.h"

#if defined(__HIPCC__)
#define GAT_HD __host__ __device__
#else
#define GAT_HD
#endif

namespace gat {

// acc_re / acc_im: [K][L][M] of ONE block; k: the channel; st / cur: its loop state and this block's parameters; returns the
// next block's parameters in `next`
GAT_HD inline void loop_update_channel(const float *acc_re, const float *acc_im, int k, int M, const gat_loop_config &cfg, gat_loop_state &st,
                                       const gat_channel_params &c, gat_channel_params &next)
{
    const int L = cfg.num_taps;
    auto tap = [&](int l, double &re, double &im) {
        re = im = 0.0;
        for (int m = 0; m < M; ++m) {
            const size_t o = ((size_t)k * L + l) * M + m;
            re += (double)acc_re[o];
            im += (double)acc_im[o];
        }
    };
    double pr, pi, er, ei, lr, li;
    tap(cfg.prompt_index, pr, pi);
    tap(cfg.early_index, er, ei);
    tap(cfg.late_index, lr, li);
    const double T = cfg.block_seconds;

    // discriminators
    const double pll_err = (pr == 0.0 && pi == 0.0) ? 0.0 : atan(pi / pr) * 0.15915494309189535; // cycles
    const double e = sqrt(er * er + ei * ei), l = sqrt(lr * lr + li * li);
    // triangle autocorrelation: L - E = 2*eps, L + E = 2 - d  =>  eps = (2-d)/2 * (L-E)/(L+E) chips
    const double dll_err = (e + l > 0.0) ? 0.5 * (2.0 - cfg.early_late_spacing_chips) * (l - e) / (e + l) : 0.0;

    // 3rd-order bilinear PLL filter (Kaplan table 5.6: w0 = Bn/0.7845, a3 = 1.1, b3 = 2.4)
    const double w0p = cfg.pll_bandwidth_hz / 0.7845;
    const double in1 = w0p * w0p * w0p * pll_err;
    const double out1 = st.pll_acc1 + 0.5 * T * in1; // bilinear integrator 1
    st.pll_acc1 += T * in1;
    const double in2 = out1 + 1.1 * w0p * w0p * pll_err;
    const double out2 = st.pll_acc2 + 0.5 * T * in2; // bilinear integrator 2
    st.pll_acc2 += T * in2;
    const double carrier_rate = out2 + 2.4 * w0p * pll_err; // Hz correction
    // 2nd-order bilinear DLL filter (w0 = Bn/0.53, a2 = 1.414)
    const double w0d = cfg.dll_bandwidth_hz / 0.53;
    const double ind = w0d * w0d * dll_err;
    const double outd = st.dll_acc + 0.5 * T * ind;
    st.dll_acc += T * ind;
    const double code_rate = outd + 1.414 * w0d * dll_err; // chips/s correction

    const double carrier_doppler = st.init_carrier_doppler_hz + carrier_rate;
    const double code_doppler = code_rate + carrier_doppler * cfg.code_freq_nominal_hz / cfg.carrier_center_hz;

    // propagate the replica NCOs over the block that was just correlated, then retune
    gat_channel_params n = c;
    double phi = c.carrier_phase_cycles + c.carrier_freq_hz * T;
    phi -= floor(phi);
    double tau = c.code_phase_chips + c.code_freq_hz * T;
    tau -= floor(tau / (double)cfg.code_length) * (double)cfg.code_length;
    n.carrier_phase_cycles = phi;
    n.code_phase_chips = tau;
    n.carrier_freq_hz = cfg.if_hz + carrier_doppler;
    n.code_freq_hz = cfg.code_freq_nominal_hz + code_doppler;
    next = n;

    st.carrier_doppler_hz = carrier_doppler; // filter output = correction on the initial estimate
    st.code_doppler_hz = code_doppler;
    st.last_pll_error_cycles = pll_err;
    st.last_dll_error_chips = dll_err;
    st.prompt_power = pr * pr + pi * pi;
}

} // namespace gat
