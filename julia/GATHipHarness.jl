# GATHipHarness.jl -- the three methods that plug libgat into the reference's own harness.
#
# `include` this file from src/GPUAcceleratedTracking.jl AFTER `include("GATHip.jl"); using .GATHip` (it is written
# against the parent module's names: KernelAlgorithm, gen_signal, NumAnts, ..., exactly as src/benchmarks.jl and
# src/algorithms.jl use them).  With the ALGODICT entry "hip_fused" => 9000, `run_kernel_benchmark(d)`
# (src/benchmarks.jl:963-979) dispatches here unchanged -- its three calls are `_run_kernel_benchmark` (method below),
# `add_results!` (generic: works on any BenchmarkTools.Trial) and `add_metadata!` (method below: the reference's own
# calls CUDA.version() and name(CUDA.CuDevice(0)), src/benchmarks.jl:20-24, which throw on a host without an NVIDIA
# device) -- and scripts/run_benchmarks_gpsl1.jl needs no edit beyond naming the algorithm.  NOT EXECUTED IN THIS PROJECT (no `julia` binary); the same sequence runs in C
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

# The same scenario through a resident correlator (ALGODICT entry "hip_resident" => 9002): one kernel stays on the device,
# the timed call is a doorbell ring + the wait for the result lines + the second stage on the host -- no launch, no stream
# wait, outputs ON THE HOST when it returns (GATHip.Resident; include/gat.h gat_resident_*).
function _run_kernel_benchmark(
    gnss,
    enable_gpu::Val{true},
    num_samples,
    num_ants,
    num_correlators,
    algorithm::KernelAlgorithm{9002}
)
    cpu_system = gnss(use_gpu = Val(false))
    system = GATHip.HipSystem(cpu_system)
    code_frequency = get_code_frequency(cpu_system)
    carrier_frequency = 1500Hz
    prn = 1
    signal_cpu, sampling_frequency = gen_signal(cpu_system, prn, carrier_frequency, num_samples,
        num_ants = NumAnts(num_ants), start_code_phase = 0.0f0, start_carrier_phase = 0.0f0)
    signal = GATHip.HipSignal(system.ctx, Matrix{Float32}(reshape(signal_cpu.re, num_samples, num_ants)),
                              Matrix{Float32}(reshape(signal_cpu.im, num_samples, num_ants)))
    correlator = EarlyPromptLateCorrelator(NumAnts(num_ants), NumAccumulators(num_correlators))
    correlator_sample_shifts = get_correlator_sample_shifts(cpu_system, correlator, sampling_frequency, 0.5)
    desc = GATHip.signal_desc(signal, 1, num_samples)
    GATHip.sync(system.ctx)                              # the signal is on the device before the first ring
    resident = GATHip.Resident(system.ctx, desc, 1, Int32[correlator_sample_shifts...], Float64(ustrip(Hz, sampling_frequency));
                               idle_us = 200000, life_ms = 60000)
    resident.prm[1] = GATHip.ChannelParams(prn - 1, 0, ustrip(Hz, code_frequency), ustrip(Hz, carrier_frequency), 0.0, 0.0)
    result = @benchmark GATHip.correlate!($resident)
    close(resident)
    GATHip.free!(system.ctx, signal)
    return result
end
add_metadata!(benchmark_results_w_params, processor, algorithm::KernelAlgorithm{9002}) = begin
    add_metadata!(benchmark_results_w_params, processor, KernelAlgorithm(9000))
    processor == "GPU" ? benchmark_results_w_params["algorithm"] = ALGODICTINV[9002] : nothing
end

# add_metadata!(benchmark_results_w_params, processor, ::KernelAlgorithm{ALGN}) (src/benchmarks.jl:11-32) for id 9000:
# more specific than the reference's `where ALGN` method, so `run_kernel_benchmark` (src/benchmarks.jl:977) lands here.
# Same keys as the reference writes ("os", "CPU_model", "GPU_model", "CUDA", "algorithm" -- `collect_results` builds its
# DataFrame columns from them) plus "HIP" (runtime version) and "libgat" (library version, git commit and build flags of
# the kernels that were timed; the script's @tagsave adds the harness's own commit, scripts/run_benchmarks_gpsl1.jl:24-27).
function add_metadata!(benchmark_results_w_params, processor, algorithm::KernelAlgorithm{9000})
    os_name = lowercase(string(Sys.KERNEL))                  # "linux" on every host libgat runs on (the "os" column)
    cpu_name = Sys.cpu_info()[1].model
    gpu_name, hip_version, _ = GATHip.device_info(0)          # gat_device_info: "AMD Instinct MI355X (gfx950...)", "7.2.x"
    benchmark_results_w_params["os"] = os_name
    benchmark_results_w_params["CPU_model"] = cpu_name
    benchmark_results_w_params["GPU_model"] = gpu_name
    benchmark_results_w_params["CUDA"] = "n/a"                # column kept for collect_results / plot scripts
    benchmark_results_w_params["HIP"] = hip_version
    benchmark_results_w_params["libgat"] = GATHip.version()
    processor == "GPU" ? benchmark_results_w_params["algorithm"] = ALGODICTINV[9000] : nothing
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
