# GATHip.jl -- reference-side binding of libgat (include/gat.h) for GPUAcceleratedTracking.jl.
#
# NOT EXECUTED ANYWHERE IN THIS PROJECT: neither this container nor the GPU box has a `julia`
# binary.  It is the thin `ccall` shim a maintainer of the reference would add (INTEGRATION.md);
# every entry point it binds is exercised through the same C ABI by the Python host layer, by
# examples/gat_known_answer.c (the same call sequence as `downconvert_and_correlate!` below, in C) and
# by the tests.  Keep it mechanical: one ccall per exported symbol, no logic of its own.  Since it cannot run, it is
# linted instead: tests/test_julia_shim_lint.py parses every `ccall` below and checks symbol, argument count and
# argument type classes against include/gat.h, the struct mirrors field by field, and that every export is bound.
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
const GAT_FLAG_GRAPH = UInt32(2)
const GAT_OWN_STREAM = Ptr{Cvoid}(typemax(UInt))   # (void *)-1: the library creates and owns a non-blocking stream
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

# struct gat_loop_config (80 bytes: 7 doubles, 5 int32, 4 bytes of tail padding)
struct LoopConfig
    block_seconds::Float64
    pll_bandwidth_hz::Float64
    dll_bandwidth_hz::Float64
    code_freq_nominal_hz::Float64
    carrier_center_hz::Float64
    if_hz::Float64
    early_late_spacing_chips::Float64
    code_length::Int32
    num_taps::Int32
    early_index::Int32
    prompt_index::Int32
    late_index::Int32
end

# struct gat_loop_state (72 bytes), one per channel, device-resident
struct LoopState
    init_carrier_doppler_hz::Float64
    carrier_doppler_hz::Float64
    code_doppler_hz::Float64
    pll_acc1::Float64
    pll_acc2::Float64
    dll_acc::Float64
    last_pll_error_cycles::Float64
    last_dll_error_chips::Float64
    prompt_power::Float64
end

# struct gat_launch_info (48 bytes): geometry of the last correlate call
struct LaunchInfo
    workgroups::Int32
    threads::Int32
    splits::Int32
    ant_tile::Int32
    vec::Int32
    lds_bytes::Int32
    finalize_launched::Int32
    matrix_core::Int32
    channels_per_wg::Int32
    blocks_per_wg::Int32
    prefetch_depth::Int32
    bf16_terms::Int32
end

# struct gat_resident_config (28 bytes) / gat_resident_info (32 bytes): the resident correlator (single-block calls
# without a kernel launch, include/gat.h)
struct ResidentConfig
    struct_size::UInt32
    idle_us::UInt32
    life_ms::UInt32
    max_calls::UInt32
    max_workgroups::UInt32
    host_pollers::UInt32
    doorbell::UInt32
end
struct ResidentInfo
    workgroups::Int32
    splits::Int32
    running::Int32
    last_exit::Int32
    launches::UInt64
    calls::UInt64
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
    # argument buffers of the single-channel call, written in place: the harness method times nothing but the ccall
    prm1::Vector{ChannelParams}
    shifts::Vector{Int32}
    desc::Base.RefValue{SignalDesc}
    # a library-owned stream by default: single-block calls then end with the completion flag gat_sync spins on
    # (include/gat.h; ~4 us less per call + sync than hipStreamSynchronize on this platform)
    function Context(device::Integer = 0, stream::Ptr{Cvoid} = GAT_OWN_STREAM)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gat_create, libgat), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, h)
        rc == GAT_OK || throw(GatError(rc, "gat_create"))
        ctx = new(h[], C_NULL, C_NULL, 0, Float32[], Float32[], [ChannelParams(0, 0, 0.0, 0.0, 0.0, 0.0)], Int32[],
                  Ref(SignalDesc(C_NULL, C_NULL, GAT_LAYOUT_PLANAR, 0, 0, 0, 0, 0)))
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

set_stream(ctx::Context, stream::Ptr{Cvoid}) =
    check(ctx, ccall((:gat_set_stream, libgat), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.handle, stream))
set_vector_tiling(ctx::Context, max_antenna_tiles::Integer, max_channels::Integer, max_blocks::Integer) =
    check(ctx, ccall((:gat_set_vector_tiling, libgat), Int32, (Ptr{Cvoid}, Int32, Int32, Int32), ctx.handle,
                     Int32(max_antenna_tiles), Int32(max_channels), Int32(max_blocks)))
# launch-geometry option by name ("dc_one_wave_min", "dc_depth", "sync_flag_wgs", ...: include/gat.h gat_set_option)
set_option(ctx::Context, name::AbstractString, value::Integer) =
    check(ctx, ccall((:gat_set_option, libgat), Int32, (Ptr{Cvoid}, Cstring, Int64), ctx.handle, name, Int64(value)))
function last_launch_info(ctx::Context)
    info = Ref(LaunchInfo(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
    check(ctx, ccall((:gat_last_launch_info, libgat), Int32, (Ptr{Cvoid}, Ref{LaunchInfo}, Csize_t), ctx.handle, info,
                     sizeof(LaunchInfo)))
    info[]
end

# library build identity: "libgat <version> (gfx950) git:<sha> flags:<-D list | none>"
version() = unsafe_string(ccall((:gat_version, libgat), Cstring, ()))

# ---- what add_metadata! needs (src/benchmarks.jl:11-32: GPU_model = name(CUDA.CuDevice(0)), CUDA = CUDA.version()):
#      (device name, HIP runtime version as "major.minor.patch", compute units)
function device_info(ctx::Context)
    name = Vector{UInt8}(undef, 256)
    ver, cus = Ref{Int32}(0), Ref{Int32}(0)
    check(ctx, ccall((:gat_device_info, libgat), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Csize_t, Ref{Int32}, Ref{Int32}),
                     ctx.handle, name, length(name), ver, cus))
    v = Int(ver[])                                   # HIP_VERSION = major * 10^7 + minor * 10^5 + patch
    unsafe_string(pointer(name)), string(v ÷ 10_000_000, ".", (v ÷ 100_000) % 100, ".", v % 100_000), Int(cus[])
end
function device_info(device::Integer = 0)
    ctx = Context(device, Ptr{Cvoid}(C_NULL))        # default stream: nothing is launched
    info = device_info(ctx)
    finalize(ctx)
    info
end

# ---- get_correlator_sample_shifts(system, correlator, fs, preferred_code_shift) (src/benchmarks.jl:105-107) as the
#      library computes it: s = max(1, round(spacing * fs / fc)), shifts[l] = (l - L ÷ 2) * s
function sample_shifts(num_taps::Integer, sampling_frequency_hz::Float64, code_frequency_hz::Float64, spacing_chips::Float64 = 0.5)
    shifts = Vector{Int32}(undef, num_taps)
    rc = ccall((:gat_sample_shifts, libgat), Int32, (Int32, Float64, Float64, Float64, Ptr{Int32}),
               Int32(num_taps), sampling_frequency_hz, code_frequency_hz, spacing_chips, shifts)
    rc == GAT_OK || throw(GatError(rc, "gat_sample_shifts"))
    shifts
end

# ---- the library's own PRN generators ("GPSL1" / "GPSL5"; IS-GPS-200 / -705) for hosts without GNSSSignals.jl:
#      (codes Int8 [code_length x num_prns], code frequency in Hz)
function gen_codes(system::AbstractString, num_prns::Integer)
    lc, fc = Ref{Int32}(0), Ref{Float64}(0.0)
    rc = ccall((:gat_gen_codes, libgat), Int32, (Cstring, Int32, Ptr{Int8}, Ref{Int32}, Ref{Float64}),
               system, Int32(num_prns), Ptr{Int8}(C_NULL), lc, fc)               # size query
    rc == GAT_OK || throw(GatError(rc, "gat_gen_codes"))
    codes = Matrix{Int8}(undef, lc[], num_prns)
    rc = ccall((:gat_gen_codes, libgat), Int32, (Cstring, Int32, Ptr{Int8}, Ref{Int32}, Ref{Float64}),
               system, Int32(num_prns), codes, lc, fc)
    rc == GAT_OK || throw(GatError(rc, "gat_gen_codes"))
    codes, fc[]
end

# ---- hipEvent pair on the context's stream (CUDA.@elapsed of test/algorithms.jl:1242): device time of what is enqueued between
timer_start(ctx::Context) = check(ctx, ccall((:gat_timer_start, libgat), Int32, (Ptr{Cvoid},), ctx.handle))
function timer_stop(ctx::Context)                   # synchronises; milliseconds
    ms = Ref{Cfloat}(0)
    check(ctx, ccall((:gat_timer_stop, libgat), Int32, (Ptr{Cvoid}, Ref{Cfloat}), ctx.handle, ms))
    Float64(ms[])
end

# per-call statistics (BenchmarkTools keeps every sample's time, src/benchmarks.jl:1-9): one lap before the first timed launch and
# one after every launch; `timer_laps` waits for the newest and returns the intervals in milliseconds
timer_lap(ctx::Context) = check(ctx, ccall((:gat_timer_lap, libgat), Int32, (Ptr{Cvoid},), ctx.handle))
function timer_laps(ctx::Context, capacity::Integer = 65536)
    ms = Vector{Cfloat}(undef, capacity)
    n = Ref{Int32}(0)
    check(ctx, ccall((:gat_timer_laps, libgat), Int32, (Ptr{Cvoid}, Ptr{Cfloat}, Int32, Ref{Int32}), ctx.handle, ms, Int32(capacity), n))
    Float64.(ms[1:n[]])
end
# a kernel that only reads `bytes` of device memory, `launches` times: milliseconds per launch (the in-run read ceiling)
function debug_read_stream(ctx::Context, dev::Ptr{Cvoid}, bytes::Integer, variant::Integer = 0, launches::Integer = 8)
    ms = Vector{Cfloat}(undef, launches)
    check(ctx, ccall((:gat_debug_read_stream, libgat), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Int32, Int32, Ptr{Cfloat}),
                     ctx.handle, dev, Csize_t(bytes), Int32(variant), Int32(launches), ms))
    Float64.(ms)
end

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

dmemset(ctx::Context, dst, value::Integer, bytes::Integer) = check(ctx, ccall((:gat_memset, libgat), Int32,
    (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Csize_t), ctx.handle, dst, Int32(value), bytes))

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

# ---- the same launch with the [K x B] parameter records already on the device (a tracking loop that produces them
#      there; flags may carry GAT_FLAG_GRAPH: the call's launches replayed as one hipGraph when it repeats)
function correlate_dev_async!(ctx::Context, desc::SignalDesc, params_dev::Ptr{Cvoid}, B::Integer, K::Integer,
                              shifts::Vector{Int32}, sampling_frequency_hz::Float64, out_re::Ptr{Cfloat}, out_im::Ptr{Cfloat},
                              flags::UInt32 = UInt32(0))
    check(ctx, ccall((:gat_downconvert_and_correlate_dev, libgat), Int32,
                     (Ptr{Cvoid}, Ref{SignalDesc}, Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Float64,
                      Ptr{Cfloat}, Ptr{Cfloat}, UInt32),
                     ctx.handle, Ref(desc), params_dev, Int32(B), Int32(K), Int32(length(shifts)), shifts,
                     sampling_frequency_hz, out_re, out_im, flags))
end

# ---- closed tracking loop around the correlator (what Tracking.track does around downconvert_and_correlate!; the
#      reference only borrows TrackingState buffers, src/benchmarks.jl:54-61).  All arrays device-resident:
#      acc [M x L x K] of one block, state LoopState[K], cur / next ChannelParams[K] (may alias)
function tracking_update!(ctx::Context, acc_re::Ptr{Cfloat}, acc_im::Ptr{Cfloat}, K::Integer, M::Integer, cfg::LoopConfig,
                          state_dev::Ptr{Cvoid}, cur_dev::Ptr{Cvoid}, next_dev::Ptr{Cvoid})
    check(ctx, ccall((:gat_tracking_update, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Int32, Int32, Ref{LoopConfig}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                     ctx.handle, acc_re, acc_im, Int32(K), Int32(M), Ref(cfg), state_dev, cur_dev, next_dev))
end
# num_blocks x {correlate, update} enqueued from native code, parameters ping-ponging between params_a (current at
# entry) and params_b; returns true when params_b holds the parameters of the block after the last one
function tracking_run!(ctx::Context, desc::SignalDesc, num_blocks::Integer, K::Integer, shifts::Vector{Int32},
                       sampling_frequency_hz::Float64, cfg::LoopConfig, state_dev::Ptr{Cvoid}, params_a_dev::Ptr{Cvoid},
                       params_b_dev::Ptr{Cvoid}, acc_re::Ptr{Cfloat}, acc_im::Ptr{Cfloat}, acc_block_stride::Integer = 0,
                       flags::UInt32 = UInt32(0))
    cur_is_b = Ref{Int32}(0)
    check(ctx, ccall((:gat_tracking_run, libgat), Int32,
                     (Ptr{Cvoid}, Ref{SignalDesc}, Int32, Int32, Int32, Ptr{Int32}, Float64, Ref{LoopConfig}, Ptr{Cvoid}, Ptr{Cvoid},
                      Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, UInt32, Ref{Int32}),
                     ctx.handle, Ref(desc), Int32(num_blocks), Int32(K), Int32(length(shifts)), shifts, sampling_frequency_hz,
                     Ref(cfg), state_dev, params_a_dev, params_b_dev, acc_re, acc_im, Int64(acc_block_stride), flags, cur_is_b))
    cur_is_b[] != 0
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
    reserve_outputs!(ctx, M * L)                         # a no-op after the first call
    correlate_single!(ctx, signal_desc(signal, signal_start_sample, num_samples), prn - 1,
                      Float64(ustrip(Hz, code_frequency)), Float64(ustrip(Hz, carrier_frequency)), Float64(code_phase),
                      Float64(carrier_phase), correlator_sample_shifts, Float64(ustrip(Hz, sampling_frequency)),
                      ctx.out_re, ctx.out_im)
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

# ---- the same call with every argument buffer cached in the context (no allocation at all per call): what
#      kernel_algorithm(..., ::KernelAlgorithm{9000}) in GATHipHarness.jl runs
function correlate_single!(ctx::Context, desc::SignalDesc, prn0::Integer,
                           code_freq_hz::Float64, carrier_freq_hz::Float64, code_phase::Float64, carrier_phase::Float64,
                           shifts, sampling_frequency_hz::Float64, out_re::Ptr{Cfloat}, out_im::Ptr{Cfloat})
    L = length(shifts)
    length(ctx.shifts) == L || resize!(ctx.shifts, L)          # first call / another tap count only
    @inbounds for l in 1:L
        ctx.shifts[l] = shifts[l]
    end
    @inbounds ctx.prm1[1] = ChannelParams(prn0, 0, code_freq_hz, carrier_freq_hz, code_phase, carrier_phase)
    ctx.desc[] = desc
    check(ctx, ccall((:gat_downconvert_and_correlate, libgat), Int32,
                     (Ptr{Cvoid}, Ref{SignalDesc}, Ptr{ChannelParams}, Int32, Int32, Int32, Ptr{Int32}, Float64,
                      Ptr{Cfloat}, Ptr{Cfloat}, UInt32),
                     ctx.handle, ctx.desc, ctx.prm1, 1, 1, L, ctx.shifts, sampling_frequency_hz, out_re, out_im, UInt32(0)))
end

# ---- reduce_cplx_multi_3/4/5 two-pass column sums (src/reduction.jl:93, :331, :548; launch sequence
#      src/algorithms.jl:914-922; test/reduction.jl:13-52): planar complex [n x cols] on the device -> [cols]
function reduce_cplx_multi!(ctx::Context, out_re::Ptr{Cfloat}, out_im::Ptr{Cfloat}, in_re::Ptr{Cfloat}, in_im::Ptr{Cfloat},
                            n::Integer, cols::Integer)
    check(ctx, ccall((:gat_reduce_cplx_multi, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Int32, Ptr{Cfloat}, Ptr{Cfloat}),
                     ctx.handle, in_re, in_im, n, cols, out_re, out_im))
end

# ---- gen_code_replica_texture_mem_strided_nsat_kernel! (src/algorithms.jl:78-98; test/algorithms.jl:1199): the
#      replicas of K satellite channels in one launch, row k of `replica_dev` (row_stride floats apart) = channel k.
#      params_dev: device array of K ChannelParams (prn, code_freq_hz, code_phase_chips used)
function gen_code_replica_nsat!(ctx::Context, replica_dev::Ptr{Cfloat}, count::Integer, row_stride::Integer, K::Integer,
                                params_dev::Ptr{Cvoid}, sampling_frequency_hz::Float64, first_shift::Integer)
    check(ctx, ccall((:gat_gen_code_replica_multi, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cfloat}, Int64, Int64, Int32, Ptr{Cvoid}, Float64, Int64),
                     ctx.handle, replica_dev, count, row_stride, K, params_dev, sampling_frequency_hz, first_shift))
end

# ---- downconvert_and_accumulate_strided_kernel! (src/algorithms.jl:828-866; test/algorithms.jl:1438-1514): the
#      materialising middle stage of the reference's algorithm 2 as a debug export -- carrier [N], downconverted
#      signal [N x M], per-sample products [N x M x L]; any output pointer may be C_NULL
function downconvert_and_accumulate!(ctx::Context, signal::HipSignal, prm::ChannelParams, shifts::Vector{Int32},
                                     sampling_frequency_hz::Float64; carrier_re = C_NULL, carrier_im = C_NULL, dw_re = C_NULL,
                                     dw_im = C_NULL, accum_re = C_NULL, accum_im = C_NULL)
    desc = signal_desc(signal, 1, signal.num_samples)
    check(ctx, ccall((:gat_downconvert_and_accumulate, libgat), Int32,
                     (Ptr{Cvoid}, Ref{SignalDesc}, Ref{ChannelParams}, Int32, Ptr{Int32}, Float64, Ptr{Cfloat}, Ptr{Cfloat},
                      Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}),
                     ctx.handle, Ref(desc), Ref(prm), length(shifts), shifts, sampling_frequency_hz, carrier_re, carrier_im,
                     dw_re, dw_im, accum_re, accum_im))
end

# ---- synthetic input on the device: gen_signal! (src/gen_signal.jl:53-175) and its noisy, steered form (AWGN sigma per
#      component, per-antenna steering phases in cycles; the reference's generator is noise-free, paper/paper.tex:116)
function gen_signal!(ctx::Context, signal::HipSignal, params_dev::Ptr{Cvoid}, K::Integer, sampling_frequency_hz::Float64;
                     amplitude = 1.0, steering_cycles_dev::Ptr{Cfloat} = Ptr{Cfloat}(C_NULL), noise_sigma = 0.0, seed = UInt64(0))
    check(ctx, ccall((:gat_gen_signal_noisy, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Int32, Int64, Int64, Int32, Int32, Ptr{Cvoid}, Float64, Float64,
                      Ptr{Cfloat}, Float64, UInt64),
                     ctx.handle, signal.re, signal.im, signal.layout, signal.num_samples, signal.num_ants, signal.num_samples,
                     signal.num_samples, 1, K, params_dev, sampling_frequency_hz, Float64(amplitude), steering_cycles_dev,
                     Float64(noise_sigma), UInt64(seed)))
end

# the reference's noise-free generator itself (src/gen_signal.jl:53-175), B blocks of K satellites summed
function gen_signal_plain!(ctx::Context, signal::HipSignal, params_dev::Ptr{Cvoid}, B::Integer, K::Integer,
                           sampling_frequency_hz::Float64, block_samples::Integer = signal.num_samples; amplitude = 1.0)
    check(ctx, ccall((:gat_gen_signal, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Int32, Int64, Int64, Int32, Int32, Ptr{Cvoid}, Float64, Float64),
                     ctx.handle, signal.re, signal.im, signal.layout, Int64(block_samples), Int32(signal.num_ants),
                     Int64(signal.num_samples), Int64(block_samples), Int32(B), Int32(K), params_dev, sampling_frequency_hz,
                     Float64(amplitude)))
end

# ---- several GPUs from one Julia task: one context per device, channels sharded contiguously, no collective
#      (include/gat.h gat_group_*; the C form of this sequence is examples/gat_multi_gpu.c)
mutable struct DeviceGroup
    handle::Ptr{Cvoid}
    n::Int
    function DeviceGroup(devices::Vector{Int32})
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gat_group_create, libgat), Int32, (Int32, Ptr{Int32}, Ref{Ptr{Cvoid}}), length(devices), devices, h)
        rc == GAT_OK || throw(GatError(rc, "gat_group_create"))
        g = new(h[], length(devices))
        finalizer(x -> ccall((:gat_group_destroy, libgat), Int32, (Ptr{Cvoid},), x.handle), g)
        g
    end
end
function device_count()
    n = Ref{Int32}(0)
    ccall((:gat_device_count, libgat), Int32, (Ref{Int32},), n) == GAT_OK || throw(GatError(1, "gat_device_count"))
    Int(n[])
end
DeviceGroup() = DeviceGroup(Int32.(0:device_count() - 1))
function gcheck(g::DeviceGroup, rc::Int32)
    rc == GAT_OK && return nothing
    throw(GatError(rc, unsafe_string(ccall((:gat_group_last_error, libgat), Cstring, (Ptr{Cvoid},), g.handle))))
end
set_codes!(g::DeviceGroup, codes::Matrix{Int8}) = gcheck(g, ccall((:gat_group_set_codes, libgat), Int32,
    (Ptr{Cvoid}, Ptr{Int8}, Int32, Int32), g.handle, codes, size(codes, 1), size(codes, 2)))
function shard(g::DeviceGroup, K::Integer, rank::Integer)       # rank 0-based -> (first channel 0-based, count)
    lo, cnt = Ref{Int32}(0), Ref{Int32}(0)
    gcheck(g, ccall((:gat_group_shard, libgat), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Int32}, Ref{Int32}), g.handle, K, rank, lo, cnt))
    Int(lo[]), Int(cnt[])
end
# bufs[r + 1] (r != src_rank) <- bufs[src_rank + 1]: the ingest device's signal to its peers (hipMemcpyPeerAsync)
replicate!(g::DeviceGroup, src_rank::Integer, bufs::Vector{Ptr{Cvoid}}, bytes::Integer) = gcheck(g, ccall(
    (:gat_group_replicate, libgat), Int32, (Ptr{Cvoid}, Int32, Ptr{Ptr{Cvoid}}, Csize_t), g.handle, src_rank, bufs, bytes))
# member r correlates its channel slice of prm ([K x B], channel fastest) on descs[r + 1] into its own device outputs
correlate!(g::DeviceGroup, descs::Vector{SignalDesc}, prm::Matrix{ChannelParams}, shifts::Vector{Int32}, fs_hz::Float64,
           out_re::Vector{Ptr{Cfloat}}, out_im::Vector{Ptr{Cfloat}}, flags::UInt32 = UInt32(0)) = gcheck(g, ccall(
    (:gat_group_correlate, libgat), Int32,
    (Ptr{Cvoid}, Ptr{SignalDesc}, Ptr{ChannelParams}, Int32, Int32, Int32, Ptr{Int32}, Float64, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, UInt32),
    g.handle, descs, prm, size(prm, 2), size(prm, 1), length(shifts), shifts, fs_hz, out_re, out_im, flags))
# the members' outputs concatenated along the channel axis: ComplexF32 [M x L x K x B] (synchronises)
function gather(g::DeviceGroup, out_re::Vector{Ptr{Cfloat}}, out_im::Vector{Ptr{Cfloat}}, B, K, L, M)
    re, im = Array{Float32}(undef, M, L, K, B), Array{Float32}(undef, M, L, K, B)
    gcheck(g, ccall((:gat_group_gather, libgat), Int32,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Int32, Int32, Int32, Int32, Ptr{Cfloat}, Ptr{Cfloat}),
        g.handle, out_re, out_im, B, K, L, M, re, im))
    complex.(re, im)
end
sync(g::DeviceGroup) = gcheck(g, ccall((:gat_group_sync, libgat), Int32, (Ptr{Cvoid},), g.handle))
function group_size(g::DeviceGroup)
    n = Ref{Int32}(0)
    gcheck(g, ccall((:gat_group_size, libgat), Int32, (Ptr{Cvoid}, Ref{Int32}), g.handle, n))
    Int(n[])
end
# member `rank`'s context handle (borrowed: destroyed with the group) for the single-device entry points
function member_handle(g::DeviceGroup, rank::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    gcheck(g, ccall((:gat_group_ctx, libgat), Int32, (Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), g.handle, Int32(rank), h))
    h[]
end
# dst (on member dst_rank's device) <- src (on member src_rank's device), asynchronous on the destination's stream,
# ordered behind the source's stream and ahead of the source's later work
function memcpy_peer!(g::DeviceGroup, dst_rank::Integer, dst::Ptr{Cvoid}, src_rank::Integer, src::Ptr{Cvoid}, bytes::Integer)
    gcheck(g, ccall((:gat_memcpy_peer, libgat), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                    member_handle(g, dst_rank), dst, member_handle(g, src_rank), src, bytes))
end

# the closed loop's update on the host (same arithmetic as gat_tracking_update on the device): `acc_re` / `acc_im` are HOST arrays
# [M x L x K] of one block -- e.g. a `Resident`'s `re` / `im` --, `state` / `cur` / `next` host vectors of length K
tracking_update_host!(acc_re::Array{Float32}, acc_im::Array{Float32}, num_channels::Integer, num_ants::Integer, cfg::LoopConfig,
                      state::Vector{LoopState}, cur::Vector{ChannelParams}, next::Vector{ChannelParams}) =
    ccall((:gat_tracking_update_host, libgat), Int32,
          (Ptr{Cfloat}, Ptr{Cfloat}, Int32, Int32, Ref{LoopConfig}, Ptr{LoopState}, Ptr{ChannelParams}, Ptr{ChannelParams}),
          acc_re, acc_im, Int32(num_channels), Int32(num_ants), Ref(cfg), state, cur, next) == GAT_OK || throw(GatError(Int32(1), "gat_tracking_update_host"))

# ---- resident correlator: what `@benchmark CUDA.@sync kernel_algorithm(...)` (src/benchmarks.jl:120-146) times, without
#      the launch -- one kernel stays on the device for a fixed call geometry (signal buffer, channels, taps), a call rings
#      it through pinned host memory and returns the outputs on the host, ComplexF32 [M x L x K].  The kernel ends by itself
#      (idle / lifetime / call budget, ResidentConfig; zeros = library defaults) and is started again by the next call.
mutable struct Resident
    handle::Ptr{Cvoid}
    ctx::Context
    re::Array{Float32,3}
    im::Array{Float32,3}
    prm::Vector{ChannelParams}
end
function Resident(ctx::Context, desc::SignalDesc, num_channels::Integer, shifts::Vector{Int32}, sampling_frequency_hz::Float64;
                  idle_us = 0, life_ms = 0, max_calls = 0, max_workgroups = 0, host_pollers = 0, doorbell = 0)
    cfg = Ref(ResidentConfig(sizeof(ResidentConfig), idle_us, life_ms, max_calls, max_workgroups, host_pollers, doorbell))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:gat_resident_open, libgat), Int32,
                     (Ptr{Cvoid}, Ref{SignalDesc}, Int32, Int32, Ptr{Int32}, Float64, Ref{ResidentConfig}, Ref{Ptr{Cvoid}}),
                     ctx.handle, Ref(desc), Int32(num_channels), Int32(length(shifts)), shifts, sampling_frequency_hz, cfg, h))
    r = Resident(h[], ctx, Array{Float32}(undef, desc.num_ants, length(shifts), num_channels),
                 Array{Float32}(undef, desc.num_ants, length(shifts), num_channels),
                 [ChannelParams(0, 0, 0.0, 0.0, 0.0, 0.0) for _ in 1:num_channels])
    finalizer(x -> x.handle == C_NULL || ccall((:gat_resident_close, libgat), Int32, (Ptr{Cvoid},), x.handle), r)
    r
end
# one call: r.prm holds the channels' records (written in place by the caller), block_offset in samples from the buffer's
# start; the outputs are in r.re / r.im when it returns
function correlate!(r::Resident, block_offset::Integer = 0)
    check(r.ctx, ccall((:gat_resident_correlate, libgat), Int32, (Ptr{Cvoid}, Ptr{ChannelParams}, Int64, Ptr{Cfloat}, Ptr{Cfloat}),
                       r.handle, r.prm, Int64(block_offset), r.re, r.im))
    r
end
function info(r::Resident)
    i = Ref(ResidentInfo(0, 0, 0, 0, 0, 0))
    check(r.ctx, ccall((:gat_resident_info_get, libgat), Int32, (Ptr{Cvoid}, Ref{ResidentInfo}, Csize_t), r.handle, i, sizeof(ResidentInfo)))
    i[]
end
park!(r::Resident) = check(r.ctx, ccall((:gat_resident_park, libgat), Int32, (Ptr{Cvoid},), r.handle))
# every resident correlator of the context: before anything that waits for the whole device (AMDGPU.synchronize(), hipFree)
park_residents!(ctx::Context) = check(ctx, ccall((:gat_resident_park_all, libgat), Int32, (Ptr{Cvoid},), ctx.handle))
# the receiver loop with the host in it, from native code: num_blocks blocks (block b at first_block_offset + b * block_stride
# samples of the buffer) through {resident call, gat_tracking_update_host}; r.prm: in the first block's records, out the next
# ones; acc_re / acc_im: host arrays of at least max(1, acc_block_stride > 0 ? num_blocks : 1) * M * L * K floats
function tracking_run!(r::Resident, num_blocks::Integer, first_block_offset::Integer, block_stride::Integer, cfg::LoopConfig,
                       state::Vector{LoopState}, acc_re::Array{Float32}, acc_im::Array{Float32}, acc_block_stride::Integer = 0)
    check(r.ctx, ccall((:gat_resident_tracking_run, libgat), Int32,
                       (Ptr{Cvoid}, Int32, Int64, Int64, Ref{LoopConfig}, Ptr{LoopState}, Ptr{ChannelParams}, Ptr{Cfloat}, Ptr{Cfloat}, Int64),
                       r.handle, Int32(num_blocks), Int64(first_block_offset), Int64(block_stride), Ref(cfg), state, r.prm, acc_re, acc_im,
                       Int64(acc_block_stride)))
    r
end
function Base.close(r::Resident)
    r.handle == C_NULL && return nothing
    rc = ccall((:gat_resident_close, libgat), Int32, (Ptr{Cvoid},), r.handle)
    r.handle = C_NULL
    check(r.ctx, rc)
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


# ---- gen_code_replica_texture_mem_kernel! (src/algorithms.jl:121-140): the replica addressed through a Float32 normalised
#      coordinate -- emulation for the code-phase-error study (scripts/code_replica_experiment.jl:81-82); study use only
function gen_code_replica_f32coord!(ctx::Context, code_replica_dev::Ptr{Cfloat}, count::Integer, prn::Integer,
                                    code_frequency_hz::Float64, sampling_frequency_hz::Float64, start_code_phase::Float64,
                                    first_shift::Integer)
    check(ctx, ccall((:gat_gen_code_replica_f32coord, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cfloat}, Int64, Int32, Float64, Float64, Float64, Int64),
                     ctx.handle, code_replica_dev, Int64(count), Int32(prn - 1), code_frequency_hz, sampling_frequency_hz,
                     start_code_phase, Int64(first_shift)))
end

# the same study with the texture unit's fixed-point addressing modelled (include/gat.h gat_gen_code_replica_texaddr):
# coord_frac_bits > 0 truncates the wrapped normalised coordinate, texel_frac_bits >= 0 rounds the texel address
function gen_code_replica_texaddr!(ctx::Context, code_replica_dev::Ptr{Cfloat}, count::Integer, prn::Integer,
                                   code_frequency_hz::Float64, sampling_frequency_hz::Float64, start_code_phase::Float64,
                                   first_shift::Integer, coord_frac_bits::Integer, texel_frac_bits::Integer)
    check(ctx, ccall((:gat_gen_code_replica_texaddr, libgat), Int32,
                     (Ptr{Cvoid}, Ptr{Cfloat}, Int64, Int32, Float64, Float64, Float64, Int64, Int32, Int32),
                     ctx.handle, code_replica_dev, Int64(count), Int32(prn - 1), code_frequency_hz, sampling_frequency_hz,
                     start_code_phase, Int64(first_shift), Int32(coord_frac_bits), Int32(texel_frac_bits)))
end

end # module
