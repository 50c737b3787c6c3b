#!/usr/bin/env python3
"""Generate tests/golden/*.json with the CPU oracle in THIS container (G1-G6).

The reference is Julia and cannot run here, so the fixtures are produced by the oracle's FP64
restatement; their pins to the reference are the literals of its own tests (G1-G4) and the
IS-GPS-200 first-10-chip octals (G5), which are typed in below by hand, NOT computed.
Inputs are stored as seeds + parameters (tests/helpers.make_case rebuilds them), outputs as FP64.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests.helpers import make_case, oracle_result  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# IS-GPS-200 Table 3-Ia, "First 10 chips octal C/A", PRN 1..32 (typed from the ICD)
CA_OCTAL = ["1440", "1620", "1710", "1744", "1133", "1455", "1131", "1454", "1626", "1504", "1642", "1750", "1764",
            "1772", "1775", "1776", "1156", "1467", "1633", "1715", "1746", "1763", "1063", "1706", "1743", "1761",
            "1770", "1774", "1127", "1453", "1625", "1712"]

# IS-GPS-705 Table 3-Ia, I5 columns, PRN 1..37 (typed from the ICD; stage 1 is the leftmost digit): the per-PRN XB
# code advance in chips and the "Initial XB Code State".  The two columns are redundant -- clocking the XB register
# (1 + x + x^3 + x^4 + x^6 + x^7 + x^8 + x^12 + x^13, all ones) `advance` times must give the state -- which
# tests/test_oracle_golden.py checks with its own 13-stage register (37 x 13 bits) before using the states to pin
# both generators.  PRN 1-16 since round 2, PRN 17-37 since round 3: every row of the L5 table now rests on the ICD.
L5I_XB_ADVANCE = [266, 365, 804, 1138, 1509, 1559, 1756, 2084, 2170, 2303, 2527, 2687, 2930, 3471, 3940, 4132, 4332, 4924,
                  5343, 5443, 5641, 5816, 5898, 5918, 5955, 6243, 6345, 6477, 6518, 6875, 7168, 7187, 7329, 7577, 7720, 7777,
                  8057]
L5I_XB_INITIAL_STATE = ["0101011100100", "1100000110101", "0100000001000", "1011000100110", "1110111010111",
                        "0110011111010", "1010010011111", "1011110100100", "1111100101011", "0111111011110",
                        "0000100111010", "1110011111001", "0001110011100", "0100000100111", "0110101011010",
                        "0001111001001", "0100110001111", "1111000011110", "1100100011111", "0110101101101",
                        "0010000001000", "1110111101111", "1000011111110", "1100010110100", "1101001101101",
                        "1010110010110", "0101011011110", "0111101010110", "0101111100001", "1000010110111",
                        "0001010011110", "0000010111001", "1101010000001", "1101111111001", "1111011011100",
                        "1001011001000", "0011010010000"]
# IS-GPS-705 section 3.2.1.1 / Figure 3-3: the XA coder (1 + x^9 + x^10 + x^12 + x^13, all ones) is short-cycled to
# 8190 chips by resetting it when this state is decoded -- the state that outputs the 8190th chip
L5_XA_DECODE_STATE = "1111111111101"
L5_XA_PERIOD = 8190

# reference literals: test/algorithms.jl:85 (and :191, :300, :1374, :1513).  G3 is NOT a reference literal: the
# reference's N = 2048 test asserts 1476 there (test/algorithms.jl:1310, a known defect); [1024 2048 1024] is this
# build's own derivation for that shape (fs = 2.048 MHz: exactly two samples per chip)
KNOWN = [
    {"id": "G1", "system": "GPSL1", "prn": 1, "N": 2500, "M": 1, "f": 1500.0, "expect": [1476, 2500, 1476], "ref": "test/algorithms.jl:85"},
    {"id": "G2", "system": "GPSL1", "prn": 1, "N": 2500, "M": 4, "f": 1500.0, "expect": [1476, 2500, 1476], "ref": "test/algorithms.jl:191"},
    {"id": "G3", "system": "GPSL1", "prn": 1, "N": 2048, "M": 4, "f": 1500.0, "expect": [1024, 2048, 1024], "ref": "builder-derived (shape of test/algorithms.jl:1161; the reference asserts 1476 there, a defect)"},
]

# G6: seeded randomised cases (tau != 0, phi != 0, Doppler +-5 kHz, IF, ragged N, K > 1)
CASES = [
    dict(seed=11, system="GPSL1", N=2500, M=1, L=3, K=1, B=1, if_hz=0.0),
    dict(seed=12, system="GPSL1", N=4000, M=1, L=3, K=1, B=2, if_hz=0.0),
    dict(seed=13, system="GPSL1", N=20000, M=4, L=3, K=1, B=1, if_hz=0.0),
    dict(seed=14, system="GPSL1", N=20000, M=4, L=3, K=2, B=2, if_hz=4.3e6),
    dict(seed=15, system="GPSL1", N=1021, M=3, L=7, K=2, B=1, if_hz=1.0e5),
    dict(seed=16, system="GPSL5", N=50000, M=4, L=5, K=3, B=1, if_hz=0.0),
    dict(seed=17, system="GPSL1", N=5000, M=16, L=3, K=4, B=1, if_hz=0.0),
    dict(seed=18, system="GPSL1", N=777, M=2, L=9, K=1, B=3, if_hz=2.5e5),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []
    for c in CASES:
        case = make_case(**c)
        ref = oracle_result(case)
        cases.append({"config": c, "shifts": case["shifts"].tolist(),
                      "params": {k: case["prm"][k].tolist() for k in case["prm"].dtype.names if k != "pad_"},
                      "signal_checksum": [float(case["re"].astype(np.float64).sum()), float(case["im"].astype(np.float64).sum())],
                      "out_re": ref.real.tolist(), "out_im": ref.imag.tolist()})
    # code-table digests (regression pins for both generators)
    import hashlib
    digests = {s: hashlib.sha256(oracle.codes(s, 32).tobytes()).hexdigest() for s in ("GPSL1", "GPSL5")}
    digests["GPSL5_37"] = hashlib.sha256(oracle.codes("GPSL5", 37).tobytes()).hexdigest()
    # replica vectors: first 40 entries for a few settings
    reps = []
    for (system, fs, tau, prn0) in (("GPSL1", 2.5e6, 0.0, 0), ("GPSL1", 20e6, 511.75, 6), ("GPSL5", 50e6, 10229.5, 2)):
        lc, fc, _ = oracle.SYSTEMS[system]
        sh = oracle.sample_shifts(3, fs, fc)
        r = oracle.gen_code_replica(oracle.codes(system, 32), prn0, fc, fs, tau, int(sh[0]), 40)
        reps.append({"system": system, "fs": fs, "tau": tau, "prn0": prn0, "first_shift": int(sh[0]), "rep": r.astype(int).tolist()})
    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump({"generator": "scripts/make_golden.py (oracle/gat_oracle.c FP64 restatement)",
                   "ca_first10_octal": CA_OCTAL, "l5i_xb_advance": L5I_XB_ADVANCE,
                   "l5i_xb_initial_state": L5I_XB_INITIAL_STATE, "l5_xa_decode_state": L5_XA_DECODE_STATE,
                   "l5_xa_period": L5_XA_PERIOD, "known_answers": KNOWN, "cases": cases, "code_sha256": digests,
                   "replicas": reps}, f)
    print("wrote", os.path.join(OUT, "golden.json"), os.path.getsize(os.path.join(OUT, "golden.json")), "bytes")


if __name__ == "__main__":
    main()
