# GATHip.jl -- reference-side binding of libgat (include/gat.h) for GPUAcceleratedTracking.jl.
#
# NOT EXECUTED ANYWHERE IN THIS PROJECT: neither this container nor the GPU box has a `julia`
# binary.  It is the thin `ccall` shim a maintainer of the reference would add (INTEGRATION.md);
# every entry point it binds is exercised through the same C ABI by the Python host layer and
# the tests.  Keep it mechanical: one ccall per exported symbol, no logic of its own.
#
# Usage inside the reference (src/GPUAcceleratedTracking.jl):
#     include("GATHip.jl"); using .GATHip
#     ALGODICT["hip_fused"] = 9000; ALGODICTINV[9000] = "hip_fused"
# and in scripts/run_benchmarks_gpsl1.jl:  "processor" => ["GPU"], "algorithm" => ["hip_fused"].
module GATHip

using StaticArrays
import Tracking
import Tracking: NumAnts, NumAccumulators, EarlyPromptLateCorrelator
import GNSSSignals: get_code_frequency, get_code_length
import Unitful: Hz, ustrip

const libgat = get(ENV, "LIBGAT", "libgat.so")

const GAT_OK = Int32(0)
const GAT_FLAG_ATOMIC = UInt32(1)
const GAT_LAYOUT_PLANAR = Int32(0)
const GAT_LAYOUT_INTERLEAVED = Int32(1)
const GAT_LAYOUT_INTERLEAVED_I16 = Int32(2)
const GAT_LAYOUT_INTERLEAVED_I8 = Int32(3)
# kernel selection (gat_set_matrix_core): vector kernel / automatic / f32 MFMA / split-bf16 MFMA
const GAT_MC_VECTOR, GAT_MC_AUTO, GAT_MC_F32, GAT_MC_BF16_SPLIT = Int32(0), Int32(1), Int32(2), Int32(3)

# struct gat_channel_params (40 bytes)
struct ChannelParams
    prn::Int32            # 0-based
    reserved::Int32
    code_freq_hz::Float64
    carrier_freq_hz::Float64
    code_phase_chips::Float64
    carrier_phase_cycles::Float64
end

# struct gat_signal_desc (56 bytes)
struct SignalDesc
    re::Ptr{Cfloat}
    im::Ptr{Cfloat}
    layout::Int32
    num_ants::Int32
    num_samples::Int64
    ant_stride::Int64
    block_stride::Int64
    chan_stride::Int64
end

struct GatError <: Exception
    status::Int32
    msg::String
end

mutable struct Context
    handle::Ptr{Cvoid}
    function Context(device::Integer = 0, stream::Ptr{Cvoid} = C_NULL)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gat_create, libgat), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, h)
        rc == GAT_OK || throw(GatError(rc, "gat_create"))
        ctx = new(h[])
        finalizer(c -> ccall((:gat_destroy, libgat), Int32, (Ptr{Cvoid},), c.handle), ctx)
        ctx
    end
end

function check(ctx::Context, rc::Int32)
    rc == GAT_OK && return nothing
    msg = unsafe_string(ccall((:gat_last_error, libgat), Cstring, (Ptr{Cvoid},), ctx.handle))
    throw(GatError(rc, msg))
end

sync(ctx::Context) = check(ctx, ccall((:gat_sync, libgat), Int32, (Ptr{Cvoid},), ctx.handle))
set_matrix_core(ctx::Context, mode::Integer) =
    check(ctx, ccall((:gat_set_matrix_core, libgat), Int32, (Ptr{Cvoid}, Int32), ctx.handle, Int32(mode)))

# ---- device memory (for hosts without AMDGPU.jl; with AMDGPU.jl pass ROCArray pointers instead)
function dmalloc(ctx::Context, bytes::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:gat_malloc, libgat), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, bytes, p))
    p[]
end
dfree(ctx::Context, p) = check(ctx, ccall((:gat_free, libgat), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.handle, p))
h2d(ctx::Context, dst, src::Array) = check(ctx, ccall((:gat_memcpy_h2d, libgat), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx.handle, dst, src, sizeof(src)))
d2h(ctx::Context, dst::Array, src) = check(ctx, ccall((:gat_memcpy_d2h, libgat), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx.handle, dst, src, sizeof(dst)))

# ---- system wrapper: `system.codes` stays whatever GNSSSignals provides; we upload it once.
struct HipSystem{S}
    system::S              # GPSL1() / GPSL5() from GNSSSignals (CPU codes)
    ctx::Context
end
function HipSystem(system; device = 0)
    ctx = Context(device)
    codes = Int8.(system.codes)                     # [code_length x num_prns], column-major
    check(ctx, ccall((:gat_set_codes, libgat), Int32, (Ptr{Cvoid}, Ptr{Int8}, Int32, Int32),
                     ctx.handle, codes, size(codes, 1), size(codes, 2)))
    HipSystem(system, ctx)
end
get_code_frequency(s::HipSystem) = get_code_frequency(s.system)
get_code_length(s::HipSystem) = get_code_length(s.system)

# ---- device-resident planar signal (StructArray{ComplexF32} layout: re / im planes, [N x M])
struct HipSignal
    re::Ptr{Cfloat}
    im::Ptr{Cfloat}
    num_samples::Int
    num_ants::Int
end
function HipSignal(ctx::Context, re::Matrix{Float32}, im::Matrix{Float32})
    dre = convert(Ptr{Cfloat}, dmalloc(ctx, sizeof(re))); h2d(ctx, dre, re)
    dim = convert(Ptr{Cfloat}, dmalloc(ctx, sizeof(im))); h2d(ctx, dim, im)
    HipSignal(dre, dim, size(re, 1), size(re, 2))
end

# ---- the operator: same argument list as Tracking.downconvert_and_correlate!
#      (call site src/benchmarks.jl:63-79).  Scratch arguments are ignored.
function Tracking.downconvert_and_correlate!(
    system::HipSystem, signal::HipSignal, correlator::EarlyPromptLateCorrelator,
    code_replica, code_phase, carrier_replica, carrier_phase, downconverted_signal,
    code_frequency, correlator_sample_shifts::SVector{L,<:Integer}, carrier_frequency,
    sampling_frequency, signal_start_sample, num_samples, prn
) where {L}
    ctx = system.ctx
    M = signal.num_ants
    off = (signal_start_sample - 1) * sizeof(Cfloat)
    desc = Ref(SignalDesc(signal.re + off, signal.im + off, GAT_LAYOUT_PLANAR, M, num_samples,
                          signal.num_samples, num_samples, 0))
    prm = Ref(ChannelParams(prn - 1, 0, ustrip(Hz, code_frequency), ustrip(Hz, carrier_frequency),
                            Float64(code_phase), Float64(carrier_phase)))
    shifts = Int32.(collect(correlator_sample_shifts))
    out_re = convert(Ptr{Cfloat}, dmalloc(ctx, 4 * M * L)); out_im = convert(Ptr{Cfloat}, dmalloc(ctx, 4 * M * L))
    rc = ccall((:gat_downconvert_and_correlate, libgat), Int32,
               (Ptr{Cvoid}, Ref{SignalDesc}, Ref{ChannelParams}, Int32, Int32, Int32, Ptr{Int32}, Float64,
                Ptr{Cfloat}, Ptr{Cfloat}, UInt32),
               ctx.handle, desc, prm, 1, 1, L, shifts, ustrip(Hz, sampling_frequency), out_re, out_im, 0)
    check(ctx, rc)
    hre = Matrix{Float32}(undef, M, L); him = Matrix{Float32}(undef, M, L)
    d2h(ctx, hre, out_re); d2h(ctx, him, out_im); dfree(ctx, out_re); dfree(ctx, out_im)
    accumulators = SVector{L}([SVector{M}(complex.(hre[:, l], him[:, l])) for l in 1:L])
    return EarlyPromptLateCorrelator(accumulators)       # functional update, as Tracking.jl does
end

# ---- Tracking.gen_code_replica! (scripts/code_replica_experiment.jl:70)
function gen_code_replica!(ctx::Context, code_replica_dev::Ptr{Cfloat}, code_frequency, sampling_frequency,
                           start_code_phase, start_sample, num_samples, shifts, prn)
    count = num_samples + shifts[end] - shifts[1]
    check(ctx, ccall((:gat_gen_code_replica, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cfloat}, Int64, Int32, Float64, Float64, Float64, Int64),
                     ctx.handle, code_replica_dev + (start_sample - 1) * sizeof(Cfloat), count, prn - 1,
                     ustrip(Hz, code_frequency), ustrip(Hz, sampling_frequency), Float64(start_code_phase), shifts[1]))
end

# ---- harness hook: the method the reference's run_kernel_benchmark dispatches to
#      (src/benchmarks.jl:963-979 -> _run_kernel_benchmark(gnss, Val(true), N, M, L, KernelAlgorithm{9000}()))
#      Written against the parent module's names; include this file from GPUAcceleratedTracking.jl.
#
# function _run_kernel_benchmark(gnss, enable_gpu::Val{true}, num_samples, num_ants, num_correlators,
#                                algorithm::KernelAlgorithm{9000})
#     system = HipSystem(gnss(use_gpu = Val(false)))
#     signal_cpu, fs = gen_signal(system.system, 1, 1500Hz, num_samples, num_ants = NumAnts(num_ants))
#     signal = HipSignal(system.ctx, Array(signal_cpu.re), Array(signal_cpu.im))
#     correlator = EarlyPromptLateCorrelator(NumAnts(num_ants), NumAccumulators(num_correlators))
#     shifts = get_correlator_sample_shifts(system.system, correlator, fs, 0.5)
#     @benchmark begin
#         Tracking.downconvert_and_correlate!($system, $signal, $correlator, nothing, 0.0, nothing, 0.0, nothing,
#             $(get_code_frequency(system)), $shifts, 1500Hz, $fs, 1, $num_samples, 1)
#         GATHip.sync($(system.ctx))                      # CUDA.@sync equivalent
#     end
# end

end # module
