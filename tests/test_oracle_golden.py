"""CPU tests: pin the oracle.

1. against the reference's own known-answer literals (test/algorithms.jl:85 ...; test/reduction.jl:51);
2. against IS-GPS-200's first-10-chip octals and IS-GPS-705's I5 initial XB code states (typed by hand in
   scripts/make_golden.py);
3. the C restatement against an independent numpy restatement;
4. the FP32 4-pass CPU baseline against the FP64 oracle (north-star tolerance 1e-5);
5. against the committed fixtures (regression)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from tests.helpers import RTOL, check_close, make_case, oracle_result

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


@pytest.mark.parametrize("ka", GOLD["known_answers"], ids=lambda k: k["id"])
def test_reference_known_answers(ka):
    """GPS L1 PRN 1, f = 1500 Hz, phases 0, taps from get_correlator_sample_shifts(.., 0.5)."""
    lc, fc, _ = oracle.SYSTEMS[ka["system"]]
    codes = oracle.codes(ka["system"], 32)
    N, M = ka["N"], ka["M"]
    fs = N / 1e-3
    sh = oracle.sample_shifts(3, fs, fc)
    assert sh.tolist() == [-1, 0, 1]
    re, im = oracle.gen_signal(codes, ka["prn"] - 1, fc, fs, ka["f"], 0.0, 0.0, N, M)
    prm = oracle.make_params(ka["prn"] - 1, fc, ka["f"], 0.0, 0.0, shape=(1, 1))
    R = oracle.correlate_f64(re, im, codes, prm, fs, sh)[0, 0]  # [L, M]
    R32 = oracle.dc_f32(re, im, codes, prm, fs, sh)[0, 0]
    for m in range(M):
        # the reference asserts `≈` (rtol = sqrt(eps(Float32)) = 3.4e-4); we hold 1e-5
        assert np.allclose(R[:, m], ka["expect"], rtol=RTOL, atol=0)
        assert np.allclose(R32[:, m], ka["expect"], rtol=RTOL, atol=RTOL * N)
    assert np.abs(R.imag).max() < 1e-3  # noise-free, phase-aligned: imaginary part ~ 0


def test_ca_first_ten_chips_octal():
    codes = oracle.codes("GPSL1", 32)
    for p, want in enumerate(GOLD["ca_first10_octal"]):
        bits = (1 - codes[p, :10].astype(int)) // 2
        v = 0
        for b in bits:
            v = (v << 1) | int(b)
        assert oct(v)[2:] == want, f"PRN {p + 1}"


def _xb_register_states(count):
    """States of the L5 XB register (IS-GPS-705: 1 + x + x^3 + x^4 + x^6 + x^7 + x^8 + x^12 + x^13, all ones at the
    start, feedback into stage 1, output = stage 13) -- a third implementation, independent of both generators."""
    taps = (1, 3, 4, 6, 7, 8, 12, 13)
    reg = [1] * 13
    out = []
    for _ in range(count):
        out.append("".join(map(str, reg)))
        fb = 0
        for t in taps:
            fb ^= reg[t - 1]
        reg = [fb] + reg[:-1]
    return out


def test_l5_icd_columns_are_consistent():
    """IS-GPS-705 Table 3-Ia gives, per PRN, the XB code advance AND the initial XB code state: clocking the register
    `advance` times from all ones must give the state.  37 x 13 bits agree -> neither column was mistyped."""
    assert len(GOLD["l5i_xb_advance"]) == len(GOLD["l5i_xb_initial_state"]) == 37
    states = _xb_register_states(max(GOLD["l5i_xb_advance"]) + 1)
    for prn, (adv, want) in enumerate(zip(GOLD["l5i_xb_advance"], GOLD["l5i_xb_initial_state"]), 1):
        assert states[adv] == want, f"PRN {prn}"


def test_l5_first_chips_from_icd_initial_states():
    """The external pin of the GPS L5 I5 code CONTENT (the reference holds no L5 vector): the first 13 XB output chips
    of a PRN are its ICD initial state read from stage 13 down to stage 1; XA starts at all ones, so the first 13 code
    chips are their complement (logic 0 -> +1, logic 1 -> -1).  PRN 1-37: every row of the table."""
    codes = oracle.codes("GPSL5", 37)
    for prn, state in enumerate(GOLD["l5i_xb_initial_state"], 1):
        xb = [int(c) for c in reversed(state)]          # output order: stage 13 first
        want = [1 - 2 * (1 ^ b) for b in xb]            # XA = 1 for the first 13 chips
        assert codes[prn - 1, :13].tolist() == want, f"PRN {prn}"


def _xa_register_outputs(count):
    """Output (stage 13) and state sequence of the L5 XA register (IS-GPS-705: 1 + x^9 + x^10 + x^12 + x^13, all ones at
    the start), WITHOUT the short cycle -- independent of both generators."""
    taps = (9, 10, 12, 13)
    reg = [1] * 13
    outs, states = [], []
    for _ in range(count):
        states.append("".join(map(str, reg)))
        outs.append(reg[12])
        fb = 0
        for t in taps:
            fb ^= reg[t - 1]
        reg = [fb] + reg[:-1]
    return outs, states


def test_l5_xa_short_cycle_decode_state():
    """The XA half: the ICD names the state whose decode resets the XA coder (short cycle to 8190 chips).  An independent
    register with the ICD's polynomial holds exactly that state while it outputs chip number 8190 (one clock before its
    natural all-ones): polynomial and cycle length agree with the ICD."""
    outs, states = _xa_register_outputs(8192)
    period = GOLD["l5_xa_period"]
    assert states[period - 1] == GOLD["l5_xa_decode_state"]
    assert states[8191] == "1" * 13 and states.index(GOLD["l5_xa_decode_state"]) == period - 1  # natural period 8191; unique


def test_l5_full_code_content_from_icd_data():
    """Every chip of every GPS L5 I5 code of the oracle = XA (independent register, reset to all ones after the ICD's
    decode state, i.e. every 8190 chips) xor XB (independent register started from the ICD's initial state of that PRN,
    free-running over the 10230 chips of 1 ms).  With the two tests above nothing of the L5 table is unpinned."""
    xa1, _ = _xa_register_outputs(GOLD["l5_xa_period"])
    xa = np.array((xa1 + xa1)[:10230])
    codes = oracle.codes("GPSL5", 37)
    taps = (1, 3, 4, 6, 7, 8, 12, 13)
    for prn, state in enumerate(GOLD["l5i_xb_initial_state"], 1):
        reg = [int(c) for c in state]
        xb = np.empty(10230, dtype=np.int64)
        for i in range(10230):
            xb[i] = reg[12]
            fb = 0
            for t in taps:
                fb ^= reg[t - 1]
            reg = [fb] + reg[:-1]
        assert np.array_equal(codes[prn - 1], 1 - 2 * (xa ^ xb)), f"PRN {prn}"


def test_code_tables_properties_and_digest():
    for system, (lc, _, _) in oracle.SYSTEMS.items():
        c = oracle.codes(system, 32)
        assert c.shape == (32, lc) and set(np.unique(c)) == {-1, 1}
        assert hashlib.sha256(c.tobytes()).hexdigest() == GOLD["code_sha256"][system]
    assert hashlib.sha256(oracle.codes("GPSL5", 37).tobytes()).hexdigest() == GOLD["code_sha256"]["GPSL5_37"]
    ca = oracle.codes("GPSL1", 32).astype(np.int64)
    # Gold-code properties: balance -1 (one more logic-1), 3-valued autocorrelation {-1, -65, 63}
    assert (ca.sum(axis=1) == -1).all()
    ac = np.array([np.dot(ca[0], np.roll(ca[0], s)) for s in range(1, 1023)])
    assert set(np.unique(ac)) <= {-1, -65, 63}
    for p in (1, 2, 7, 19, 32):
        assert np.array_equal(oracle.np_code_gpsl1(p), ca[p - 1])


def test_reduction_all_ones():
    """test/reduction.jl:51-52, :126-127 ...: ones + 0im of (N, M, L) sums to [N N N]."""
    for n, ml in ((2048, 3), (2500, 12), (32768, 16)):
        out = oracle.reduce_cplx_multi(np.ones((ml, n), np.float32), np.zeros((ml, n), np.float32))
        assert (out.real == n).all() and (out.imag == 0).all()


@pytest.mark.parametrize("seed", range(6))
def test_c_oracle_matches_numpy_restatement(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(50, 3000))
    M = int(rng.integers(1, 4))
    L = int(rng.choice([1, 3, 5]))
    case = make_case(100 + seed, N=N, M=M, L=L, K=1, B=1, if_hz=float(rng.choice([0.0, 3.1e5])))
    ref = oracle_result(case)[0, 0]
    p = case["prm"][0, 0]
    alt = oracle.np_correlate(case["re"], case["im"], case["codes"][p["prn0"]], p["code_freq_hz"], case["fs"],
                              p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"], case["shifts"])
    assert np.abs(ref - alt).max() <= 1e-9 * np.abs(ref).max()


def test_negative_shift_wraps_to_code_end():
    """Early tap at n = 0 has a negative code phase: floored mod wraps to the code end
    (Julia mod, src/algorithms.jl:182)."""
    codes = oracle.codes("GPSL1", 1)
    rep = oracle.gen_code_replica(codes, 0, 1.023e6, 2.5e6, 0.0, -3, 8)
    want = [codes[0, (int(np.floor(0.4092 * i)) % 1023)] for i in range(-3, 5)]
    assert rep.tolist() == [float(w) for w in want]
    assert rep[0] == codes[0, 1021] and rep[2] == codes[0, 1022] and rep[3] == codes[0, 0]


def test_replica_fixtures():
    for r in GOLD["replicas"]:
        lc, fc, _ = oracle.SYSTEMS[r["system"]]
        got = oracle.gen_code_replica(oracle.codes(r["system"], 32), r["prn0"], fc, r["fs"], r["tau"], r["first_shift"], 40)
        assert got.astype(int).tolist() == r["rep"]


@pytest.mark.parametrize("g", GOLD["cases"], ids=lambda g: "seed%d" % g["config"]["seed"])
def test_golden_cases(g):
    case = make_case(**g["config"])
    assert case["shifts"].tolist() == g["shifts"]
    assert np.isclose(case["re"].astype(np.float64).sum(), g["signal_checksum"][0], rtol=0, atol=1e-3)
    ref = np.array(g["out_re"]) + 1j * np.array(g["out_im"])
    got = oracle_result(case)
    assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
    # the FP32 4-pass CPU baseline (what bench.py times) holds the north-star tolerance too
    if case["K"] >= 1:
        f32 = oracle.dc_f32(case["re"], case["im"], case["codes"], case["prm"], case["fs"], case["shifts"], N=case["N"])
        check_close(f32, ref, what="cpu f32 4-pass")


def test_sample_shifts():
    assert oracle.sample_shifts(3, 2.5e6, 1.023e6).tolist() == [-1, 0, 1]       # pinned by 1476
    assert oracle.sample_shifts(3, 20e6, 1.023e6).tolist() == [-10, 0, 10]      # BASELINE C2
    assert oracle.sample_shifts(5, 50e6, 10.23e6).tolist() == [-4, -2, 0, 2, 4]  # BASELINE C3
    assert oracle.sample_shifts(3, 50e6, 1.023e6).tolist() == [-24, 0, 24]      # BASELINE C4
    assert oracle.sample_shifts(3, 1.0e6, 1.023e6).tolist() == [-1, 0, 1]       # max(1, .)
    assert oracle.sample_shifts(7, 20e6, 1.023e6).tolist() == [-30, -20, -10, 0, 10, 20, 30]
