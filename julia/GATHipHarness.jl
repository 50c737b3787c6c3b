# GATHipHarness.jl -- the two methods that plug libgat into the reference's own harness.
#
# `include` this file from src/GPUAcceleratedTracking.jl AFTER `include("GATHip.jl"); using .GATHip` (it is written
# against the parent module's names: KernelAlgorithm, gen_signal, NumAnts, ..., exactly as src/benchmarks.jl and
# src/algorithms.jl use them).  With the ALGODICT entry "hip_fused" => 9000, `run_kernel_benchmark(d)`
# (src/benchmarks.jl:963-979) dispatches here unchanged, and scripts/run_benchmarks_gpsl1.jl needs no edit beyond
# naming the algorithm.  NOT EXECUTED IN THIS PROJECT (no `julia` binary); the same sequence runs in C
# (examples/gat_known_answer.c) and in Python (gpuacceleratedtracking_amd/benchmarks.py::_run_kernel_benchmark).

# _run_kernel_benchmark(gnss, ::Val{true}, num_samples, num_ants, num_correlators, ::KernelAlgorithm{N})
# (src/benchmarks.jl:83-174): fixed scenario prn 1, 1500 Hz, phases 0, 0.5-chip spacing; device-resident input;
# sync-inclusive wall time per call.  What is timed: argument marshalling + ONE fused launch + stream sync -- no
# allocation, no copy (GATHip.correlate_async! writes into the context's cached buffers).
function _run_kernel_benchmark(
    gnss,
    enable_gpu::Val{true},
    num_samples,
    num_ants,
    num_correlators,
    algorithm::KernelAlgorithm{9000}
)
    cpu_system = gnss(use_gpu = Val(false))
    system = GATHip.HipSystem(cpu_system)
    code_frequency = get_code_frequency(cpu_system)
    start_code_phase = 0.0f0
    carrier_phase = 0.0f0
    carrier_frequency = 1500Hz
    prn = 1

    signal_cpu, sampling_frequency = gen_signal(cpu_system, prn, carrier_frequency, num_samples,
        num_ants = NumAnts(num_ants), start_code_phase = start_code_phase, start_carrier_phase = carrier_phase)
    signal = GATHip.HipSignal(system.ctx, Matrix{Float32}(reshape(signal_cpu.re, num_samples, num_ants)),
                              Matrix{Float32}(reshape(signal_cpu.im, num_samples, num_ants)))

    correlator = EarlyPromptLateCorrelator(NumAnts(num_ants), NumAccumulators(num_correlators))
    correlator_sample_shifts = get_correlator_sample_shifts(cpu_system, correlator, sampling_frequency, 0.5)

    desc = GATHip.signal_desc(signal, 1, num_samples)
    prm = [GATHip.ChannelParams(prn - 1, 0, ustrip(Hz, code_frequency), ustrip(Hz, carrier_frequency),
                                Float64(start_code_phase), Float64(carrier_phase))]
    shifts = Int32[correlator_sample_shifts...]
    fs = Float64(ustrip(Hz, sampling_frequency))
    GATHip.reserve_outputs!(system.ctx, num_ants * num_correlators)   # outside the timed region
    result = @benchmark begin
        GATHip.correlate_async!($(system.ctx), $desc, $prm, $shifts, $fs, $num_ants)
        GATHip.sync($(system.ctx))                      # CUDA.@sync equivalent (src/benchmarks.jl:120)
    end
    GATHip.free!(system.ctx, signal)
    return result
end

# kernel_algorithm(..., ::KernelAlgorithm{9000}) with the 26-argument form of the reference's single-launch
# algorithms (src/algorithms.jl:1485-1512, KernelAlgorithm{4431}): launch-shape arguments and scratch buffers are
# accepted and ignored; `signal_re` / `signal_im` are device pointers of a planar [num_samples x NANT] signal
# (GATHip.HipSignal planes), `accum_re` / `accum_im` device pointers that receive the [NANT x NCOR] result;
# `codes` is the GATHip.HipSystem whose table is resident on the device.
function kernel_algorithm(
    threads_per_block,
    blocks_per_grid,
    shmem_size,
    code_replica,
    codes::GATHip.HipSystem,
    code_frequency,
    sampling_frequency,
    start_code_phase,
    prn,
    num_samples,
    num_of_shifts,
    code_length,
    accum_re::Ptr{Cfloat},
    accum_im::Ptr{Cfloat},
    carrier_replica_re,
    carrier_replica_im,
    downconverted_signal_re,
    downconverted_signal_im,
    signal_re::Ptr,
    signal_im::Ptr,
    correlator_sample_shifts::SVector{NCOR, Int64},
    carrier_frequency,
    carrier_phase,
    num_ants::NumAnts{NANT},
    num_corrs,
    algorithm::KernelAlgorithm{9000}
) where {NANT, NCOR}
    # no allocation here: parameters, tap list and signal descriptor live in the context and are written in place
    desc = GATHip.SignalDesc(signal_re, signal_im, GATHip.GAT_LAYOUT_PLANAR, NANT, num_samples, num_samples, num_samples, 0) # isbits
    GATHip.correlate_single!(codes.ctx, desc, prn - 1, Float64(ustrip(Hz, code_frequency)),
                             Float64(ustrip(Hz, carrier_frequency)), Float64(start_code_phase), Float64(carrier_phase),
                             correlator_sample_shifts, Float64(ustrip(Hz, sampling_frequency)), accum_re, accum_im)
    return nothing
end
