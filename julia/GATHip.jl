# GATHip.jl -- reference-side binding of libgat (include/gat.h) for GPUAcceleratedTracking.jl.
#
# NOT EXECUTED ANYWHERE IN THIS PROJECT: neither this container nor the GPU box has a `julia`
# binary.  It is the thin `ccall` shim a maintainer of the reference would add (INTEGRATION.md);
# every entry point it binds is exercised through the same C ABI by the Python host layer, by
# examples/gat_known_answer.c (the same call sequence as `downconvert_and_correlate!` below, in C) and
# by the tests.  Keep it mechanical: one ccall per exported symbol, no logic of its own.
#
# Usage inside the reference (src/GPUAcceleratedTracking.jl), no other edits:
#     include("GATHip.jl"); using .GATHip
#     include("GATHipHarness.jl")         # methods of _run_kernel_benchmark / kernel_algorithm for id 9000
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
    re::Ptr{Cvoid}
    im::Ptr{Cvoid}
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

# One libgat context plus the buffers the operator re-uses on every call: device outputs and their host copies are
# allocated once (grown on demand), so that a timed `downconvert_and_correlate!` does no allocation at all.
mutable struct Context
    handle::Ptr{Cvoid}
    out_re::Ptr{Cfloat}      # device, out_cap floats each
    out_im::Ptr{Cfloat}
    out_cap::Int
    host_re::Vector{Float32}
    host_im::Vector{Float32}
    function Context(device::Integer = 0, stream::Ptr{Cvoid} = C_NULL)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gat_create, libgat), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, h)
        rc == GAT_OK || throw(GatError(rc, "gat_create"))
        ctx = new(h[], C_NULL, C_NULL, 0, Float32[], Float32[])
        finalizer(ctx) do c
            c.out_re == C_NULL || ccall((:gat_free, libgat), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), c.handle, c.out_re)
            c.out_im == C_NULL || ccall((:gat_free, libgat), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), c.handle, c.out_im)
            ccall((:gat_destroy, libgat), Int32, (Ptr{Cvoid},), c.handle)
        end
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
d2h(ctx::Context, dst::Array, src, bytes::Integer = sizeof(dst)) = check(ctx, ccall((:gat_memcpy_d2h, libgat), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx.handle, dst, src, bytes))

# make room for `n` output floats per plane (device + host); a no-op once the buffers are large enough
function reserve_outputs!(ctx::Context, n::Integer)
    n <= ctx.out_cap && return nothing
    ctx.out_re == C_NULL || dfree(ctx, ctx.out_re)
    ctx.out_im == C_NULL || dfree(ctx, ctx.out_im)
    ctx.out_re = convert(Ptr{Cfloat}, dmalloc(ctx, 4n))
    ctx.out_im = convert(Ptr{Cfloat}, dmalloc(ctx, 4n))
    ctx.out_cap = n
    resize!(ctx.host_re, n); resize!(ctx.host_im, n)
    nothing
end

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

# ---- device-resident signal [N x M] in any of libgat's four sample layouts
struct HipSignal
    re::Ptr{Cvoid}          # planar: Float32 real plane; interleaved layouts: base pointer
    im::Ptr{Cvoid}          # planar: Float32 imaginary plane; interleaved layouts: C_NULL
    layout::Int32
    sample_bytes::Int       # bytes of one sample in `re` (planar: 4; ComplexF32: 8; int16 pairs: 4; int8 pairs: 2)
    num_samples::Int
    num_ants::Int
end
# planar re / im planes: the reference's StructArray{ComplexF32} (src/gen_signal.jl:179)
function HipSignal(ctx::Context, re::Matrix{Float32}, im::Matrix{Float32})
    dre = dmalloc(ctx, sizeof(re)); h2d(ctx, dre, re)
    dim = dmalloc(ctx, sizeof(im)); h2d(ctx, dim, im)
    HipSignal(dre, dim, GAT_LAYOUT_PLANAR, 4, size(re, 1), size(re, 2))
end
# interleaved ComplexF32 [N x M]
function HipSignal(ctx::Context, x::Matrix{ComplexF32})
    d = dmalloc(ctx, sizeof(x)); h2d(ctx, d, x)
    HipSignal(d, C_NULL, GAT_LAYOUT_INTERLEAVED, 8, size(x, 1), size(x, 2))
end
# interleaved {Int16 re, Int16 im} pairs as a front-end delivers them ("sc16"): Complex{Int16} [N x M]
function HipSignal(ctx::Context, x::Matrix{Complex{Int16}})
    d = dmalloc(ctx, sizeof(x)); h2d(ctx, d, x)
    HipSignal(d, C_NULL, GAT_LAYOUT_INTERLEAVED_I16, 4, size(x, 1), size(x, 2))
end
function HipSignal(ctx::Context, x::Matrix{Complex{Int8}})
    d = dmalloc(ctx, sizeof(x)); h2d(ctx, d, x)
    HipSignal(d, C_NULL, GAT_LAYOUT_INTERLEAVED_I8, 2, size(x, 1), size(x, 2))
end
function free!(ctx::Context, s::HipSignal)
    dfree(ctx, s.re); s.im == C_NULL || dfree(ctx, s.im)
end

function signal_desc(signal::HipSignal, start_sample::Integer, num_samples::Integer)
    off = (start_sample - 1) * signal.sample_bytes
    SignalDesc(signal.re + off, signal.im == C_NULL ? C_NULL : signal.im + off, signal.layout, signal.num_ants,
               num_samples, signal.num_samples, num_samples, 0)
end

# ---- the hot call, nothing but the ccall: K channels of one block into the context's cached output buffers
#      [M x L x K] (asynchronous on the context's stream)
function correlate_async!(ctx::Context, desc::SignalDesc, prm::Vector{ChannelParams}, shifts::Vector{Int32},
                          sampling_frequency_hz::Float64, M::Integer, flags::UInt32 = UInt32(0))
    K, L = length(prm), length(shifts)
    reserve_outputs!(ctx, M * L * K)
    check(ctx, ccall((:gat_downconvert_and_correlate, libgat), Int32,
                     (Ptr{Cvoid}, Ref{SignalDesc}, Ptr{ChannelParams}, Int32, Int32, Int32, Ptr{Int32}, Float64,
                      Ptr{Cfloat}, Ptr{Cfloat}, UInt32),
                     ctx.handle, Ref(desc), prm, 1, K, L, shifts, sampling_frequency_hz, ctx.out_re, ctx.out_im, flags))
end
# blocking read-back of the last result into the cached host buffers: (re, im) views of length M*L*K
function fetch_result!(ctx::Context, n::Integer)
    d2h(ctx, ctx.host_re, ctx.out_re, 4n); d2h(ctx, ctx.host_im, ctx.out_im, 4n)   # gat_memcpy_d2h synchronises
    view(ctx.host_re, 1:n), view(ctx.host_im, 1:n)
end

# ---- the operator: same argument list as Tracking.downconvert_and_correlate!
#      (call site src/benchmarks.jl:63-79).  Scratch arguments are ignored.  No allocation on the device or of
#      result arrays: outputs live in the context (first call sizes them).
function Tracking.downconvert_and_correlate!(
    system::HipSystem, signal::HipSignal, correlator::EarlyPromptLateCorrelator,
    code_replica, code_phase, carrier_replica, carrier_phase, downconverted_signal,
    code_frequency, correlator_sample_shifts::SVector{L,<:Integer}, carrier_frequency,
    sampling_frequency, signal_start_sample, num_samples, prn
) where {L}
    ctx = system.ctx
    M = signal.num_ants
    prm = [ChannelParams(prn - 1, 0, ustrip(Hz, code_frequency), ustrip(Hz, carrier_frequency),
                         Float64(code_phase), Float64(carrier_phase))]
    correlate_async!(ctx, signal_desc(signal, signal_start_sample, num_samples), prm,
                     Int32[correlator_sample_shifts...], Float64(ustrip(Hz, sampling_frequency)), M)
    re, im = fetch_result!(ctx, M * L)
    accumulators = SVector{L}(ntuple(l -> SVector{M}(ntuple(m -> complex(re[(l - 1) * M + m], im[(l - 1) * M + m]), M)), L))
    return EarlyPromptLateCorrelator(accumulators)       # functional update, as Tracking.jl does
end

# ---- K satellite channels of one block in ONE launch (the reference's experimental _3d_4431! kernel,
#      src/algorithms.jl:637, correlates several satellites per launch too).  Returns ComplexF32 [M x L x K].
function downconvert_and_correlate_channels!(
    system::HipSystem, signal::HipSignal, code_phases, carrier_phases, code_frequencies,
    correlator_sample_shifts::SVector{L,<:Integer}, carrier_frequencies, sampling_frequency,
    signal_start_sample, num_samples, prns
) where {L}
    ctx = system.ctx
    M, K = signal.num_ants, length(prns)
    prm = [ChannelParams(prns[k] - 1, 0, ustrip(Hz, code_frequencies[k]), ustrip(Hz, carrier_frequencies[k]),
                         Float64(code_phases[k]), Float64(carrier_phases[k])) for k in 1:K]
    correlate_async!(ctx, signal_desc(signal, signal_start_sample, num_samples), prm,
                     Int32[correlator_sample_shifts...], Float64(ustrip(Hz, sampling_frequency)), M)
    re, im = fetch_result!(ctx, M * L * K)
    reshape(complex.(re, im), M, L, K)
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

end # module
