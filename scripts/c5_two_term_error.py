#!/usr/bin/env python3
"""configs[4], review item: would a 2-term bf16 split of the SAMPLE operand (hi + mid = 16 of 24 mantissa bits; products
{hh, mh, hm, mm, lh} -> 3 samples per 16-slot MFMA instead of 2) hold the 1e-5 tolerance?  Pure arithmetic, evaluated on the
CPU in float64: R = sum_n x_n w_n against R2 = sum_n (hi + mid)(x_n) w_n with the W operand exact, for the accumulation
lengths of configs[4] (N = 2e6).  Truncation (what split3 in gat_mfma_bf16.hip does) and round-to-nearest for the mid term."""
import numpy as np


def bf16_trunc(v):
    return (v.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def bf16_rne(v):
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def two_term(x, rounding):
    hi = bf16_trunc(x)
    r = (x - hi).astype(np.float32)
    mid = bf16_trunc(r) if rounding == "trunc" else bf16_rne(r)
    return hi.astype(np.float64) + mid.astype(np.float64)


def case(name, x, w):
    x = x.astype(np.float32)
    ref = np.sum(x.astype(np.float64) * w)
    for rounding in ("trunc", "rne"):
        got = np.sum(two_term(x, rounding) * w)
        print(f"{name:58s} {rounding:5s}  rel err {abs(got - ref) / abs(ref):.2e}")


def main():
    rng = np.random.default_rng(4)
    N = 2_000_000
    n = np.arange(N)
    car = np.exp(-2j * np.pi * (n * 0.01234567 + 0.3))
    chips = rng.choice([-1.0, 1.0], size=N)
    sig = (chips * np.conj(car))  # the matching signal: R = N
    print("# 2-term sample split vs exact, float64 accumulation, N = 2e6 (one block of configs[4]); tolerance of the path: 1e-5,")
    print("# bar for keeping the variant (review): worst case < 3e-6")
    case("unit signal + AWGN sigma 1 (re part)", (sig.real + rng.standard_normal(N)), (chips * car).real)
    case("unit signal, no noise (gen_signal.jl)", sig.real, (chips * car).real)
    case("AWGN only, sigma 1 (non-matching PRN: judged on the norm)", rng.standard_normal(N), (chips * car).real)
    # adversarial: constant carrier (f = 0), constant samples whose mantissa bits below the 16th are all ones
    x_bad = np.float32(1.0) + np.float32(2.0 ** -16) * np.float32(1 - 2.0 ** -7)
    case("constant sample 1 + 2^-16 (1 - 2^-7), carrier f = 0", np.full(N, x_bad), np.ones(N))
    x_bad2 = np.float32(1.0 + 2.0 ** -9 + 2.0 ** -17 + 2.0 ** -18 + 2.0 ** -19)
    case("constant sample 1 + 2^-9 + 2^-17 + 2^-18 + 2^-19, f = 0", np.full(N, x_bad2), np.ones(N))
    # int16 front end: 16 significant bits at most -> two terms are exact
    xi = rng.integers(-32768, 32767, size=N).astype(np.float32)
    case("int16 samples (exact in two terms)", xi, (chips * car).real)


if __name__ == "__main__":
    main()
