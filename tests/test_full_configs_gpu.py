"""GPU tests at BASELINE.json's FULL sizes.  The FP64 oracle is too slow for whole workloads, so
each config is checked (a) against the oracle on a slice it finishes in seconds and (b) through
size-independent properties of the operator on the whole output: block/channel independence
(batched result == the same block or channel run alone, bit for bit -- the sum order per
(block, channel, antenna tile) does not depend on the batch), exact linearity under power-of-two
scaling, antenna-permutation equivariance, and run-to-run determinism."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def _oracle_params(prm):
    return oracle.make_params(prm["prn"], prm["code_freq_hz"], prm["carrier_freq_hz"], prm["code_phase_chips"],
                              prm["carrier_phase_cycles"])


def _check_slice(g, op, sig, prm, blocks, chans, N, fs):
    """oracle on (blocks x chans) of the stream."""
    got = op.result()
    for b in blocks:
        re = sig[0][:, b * N:(b + 1) * N].cpu().numpy()
        im = sig[1][:, b * N:(b + 1) * N].cpu().numpy()
        p = np.ascontiguousarray(_oracle_params(prm)[b:b + 1][:, chans])
        ref = oracle.correlate_f64(re, im, op.system.codes, p, fs, op.shifts, N=N)[0]
        for i, k in enumerate(chans):
            scale = np.abs(ref[i]).max()
            assert np.abs(got[b, k] - ref[i]).max() / scale <= RTOL, (b, k)


CONFIGS = {
    # name: (gnss, N, M, L, K, B, block_ms)
    "C1": ("GPSL1", 4000, 1, 3, 1, 64, 1.0),
    "C2": ("GPSL1", 20000, 4, 3, 1, 4096, 1.0),     # the bench workload, full batch
    "C3": ("GPSL5", 50000, 4, 5, 12, 8, 1.0),
    "C4": ("GPSL1", 50000, 16, 3, 4, 8, 1.0),       # per-GPU shard of config 4 (4 of 32 PRNs)
    "C4x32": ("GPSL1", 50000, 16, 3, 32, 8, 1.0),   # the whole constellation on ONE GPU (bench.py constellation_config3 at N = 1)
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_full_size(g, name):
    import torch
    gnss, N, M, L, K, B, _ = CONFIGS[name]
    op, desc, sig, prm = g.build_stream(gnss, N, M, L, K, B)
    fs = N / 1e-3
    op.launch(desc)
    full = op.result().copy()
    geometry = op.ctx.last_launch_info()
    assert np.isfinite(full.view(np.float32)).all()
    # (a) oracle on the first and last block, two channels
    chans = sorted({0, K - 1})
    _check_slice(g, op, sig, prm, sorted({0, B - 1}), chans, N, fs)
    # (b1) determinism
    op.launch(desc)
    assert np.array_equal(op.result().view(np.float32), full.view(np.float32))
    # (b2) block independence: blocks [b0, b0+2) run alone (their samples get split over many
    # workgroups -> other, still deterministic, sum order) agree to FP32 rounding
    b0 = B // 2
    sub = g.StreamCorrelator(op.system, N, M, 2, K, op.shifts, fs)
    sub.set_params(prm[b0:b0 + 2])
    sub(sig[0][:, b0 * N:(b0 + 2) * N], sig[1][:, b0 * N:(b0 + 2) * N])
    assert np.abs(sub.result() - full[b0:b0 + 2]).max() <= 1e-6 * np.abs(full).max()
    # (b3) exact linearity: x -> 4x scales every accumulator by exactly 4
    re4, im4 = sig[0] * 4.0, sig[1] * 4.0
    op(re4, im4)
    assert np.array_equal(op.result().view(np.float32), (4.0 * full).view(np.float32))
    del re4, im4
    # (b4) antenna permutation equivariance (within an antenna tile the arithmetic is per antenna)
    if M > 1:
        perm = torch.arange(M - 1, -1, -1, device=sig[0].device)
        op(sig[0][perm].contiguous(), sig[1][perm].contiguous())
        assert np.array_equal(np.ascontiguousarray(op.result()[..., ::-1]).view(np.float32), full.view(np.float32))
    # (b5) channel independence: channel K-1 alone == its slot in the batch
    if K > 1:
        one = g.StreamCorrelator(op.system, N, M, B, 1, op.shifts, fs)
        one.set_params(prm[:, K - 1:K])
        one(sig[0], sig[1])
        same_geometry = one.ctx.last_launch_info()["splits"] == geometry["splits"]
        if same_geometry:
            assert np.array_equal(np.ascontiguousarray(one.result()[:, 0]).view(np.float32),
                                  np.ascontiguousarray(full[:, K - 1]).view(np.float32))
        else:  # different split count -> different (still deterministic) sum order
            assert np.abs(one.result()[:, 0] - full[:, K - 1]).max() <= 1e-6 * np.abs(full).max()


def test_config5_wideband_crpa(g):
    """BASELINE config 5: 64 antennas, 64 channels, 20 ms coherent @ 100 MHz (N = 2 000 000, 1 GB of
    signal).  Oracle on one channel x 8 antennas; channel independence + linearity on the rest."""
    import torch
    N, M, L, K, B = 2_000_000, 64, 3, 64, 1
    fs = 100e6
    system = g.GPSL1()
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, L), fs, 0.5)
    rng = np.random.default_rng(5)
    prn = np.arange(K) % 32
    f = 2.5e6 + rng.uniform(-5e3, 5e3, size=K)
    tau = rng.uniform(0, 1023, size=K)
    phi = rng.uniform(0, 1, size=K)
    prm = g.make_params(prn, 1.023e6, f, tau, phi, shape=(1, K))
    # signal: 4 of the channels are present (sum), identical on all antennas, then per-antenna gain
    psig = prm[:, :4].copy()
    psig["carrier_phase_cycles"] = 2 * np.pi * psig["carrier_phase_cycles"]
    re, im = g.gen_signal_stream(system, psig, fs, N, M)
    gain = torch.linspace(0.5, 1.5, M, device=re.device)[:, None]
    re *= gain
    im *= gain
    op = g.StreamCorrelator(system, N, M, B, K, shifts, fs)
    op.set_params(prm)
    op(re, im)
    full = op.result().copy()
    assert np.isfinite(full.view(np.float32)).all()
    # oracle: channel 1, antennas 0..7 (8 x 2e6 samples x 3 taps in FP64)
    sub = slice(0, 8)
    o = oracle.correlate_f64(re[sub].cpu().numpy(), im[sub].cpu().numpy(), system.codes,
                             oracle.make_params(prm["prn"], prm["code_freq_hz"], prm["carrier_freq_hz"],
                                                prm["code_phase_chips"], prm["carrier_phase_cycles"])[:, 1:2],
                             fs, shifts, N=N)[0, 0]
    assert np.abs(full[0, 1][:, sub] - o).max() / np.abs(o).max() <= RTOL
    # present channels correlate (|prompt| ~ N * gain), absent ones do not
    assert abs(full[0, 1, 1, 32]) > 0.5 * N * float(gain[32]) and abs(full[0, 40, 1, 32]) < 0.02 * N
    # linearity (exact) and determinism
    op(re * 0.5, im * 0.5)
    assert np.array_equal(op.result().view(np.float32), (0.5 * full).view(np.float32))
    # channel independence: channels 8..15 alone == their slots in the 64-channel batch (to rounding)
    one = g.StreamCorrelator(system, N, M, B, 8, shifts, fs)
    one.set_params(prm[:, 8:16])
    one(re, im)
    assert np.abs(one.result()[0] - full[0, 8:16]).max() <= 1e-6 * np.abs(full).max()
