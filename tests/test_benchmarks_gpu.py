"""GPU tests of the harness mirrors: run_kernel_benchmark / run_reduction_benchmark / run_replica_benchmark
(src/benchmarks.jl:963-979, :1137-1148; src/replica_benchmarks.jl:137-147) return the reference's result keys
and correct values on the reference's own scenarios."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


@pytest.mark.parametrize("alg", ["pure", "cplx", "cplx_multi"])
def test_run_reduction_benchmark(g, alg):
    r = g.run_reduction_benchmark({"num_samples": 2500, "num_ants": 4, "num_correlators": 3, "algorithm": alg},
                                  seconds=0.02)
    # (the [N N N] known answer of test/reduction.jl:51-52 is asserted inside for every algorithm)
    for key in ("Minimum", "Mean", "Median", "Std", "num_samples", "num_ants", "num_correlators", "algorithm"):
        assert key in r
    assert 0 < r["Minimum"] <= r["Median"] and r["samples"] >= 10


@pytest.mark.parametrize("alg", ["gmem", "textmem"])
def test_run_replica_benchmark(g, alg):
    r = g.run_replica_benchmark({"num_samples": 2048, "algorithm": alg}, seconds=0.02)
    for key in ("Minimum", "Mean", "Median", "Std", "num_samples", "algorithm"):
        assert key in r
    assert 0 < r["Minimum"] <= r["Median"]


@pytest.mark.parametrize("alg", ["hip_fused", "hip_resident"])
def test_run_kernel_benchmark_known_answer(g, alg):
    r = g.run_kernel_benchmark({"processor": "GPU", "GNSS": "GPSL1", "num_samples": 2500, "num_ants": 4,
                                "num_correlators": 3, "algorithm": alg}, seconds=0.02)
    assert np.allclose(r["accumulators"][:, 0], [1476, 2500, 1476], rtol=1e-5)  # test/algorithms.jl:85
    for key in ("Minimum", "Median", "Mean", "σ", "Maximum", "RawTimes", "os", "CPU_model", "GPU_model", "algorithm"):
        assert key in r
