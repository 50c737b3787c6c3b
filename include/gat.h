/*
 * gat.h -- C ABI of libgat, the MI355X (gfx950) GNSS tracking correlator.
 *
 * Drop-in boundary for ONE path of coezmaden/GPUAcceleratedTracking: downconvert + correlate
 * (carrier wipe-off, PRN code replica, multi-tap multiply over antennas, reduction).  The
 * reference has no FFI layer -- its boundary is Julia multiple dispatch -- so every entry point
 * below names the reference interface it stands in for (paths relative to the reference tree).
 * INTEGRATION.md shows the Julia `ccall` shim that binds them.
 *
 * Conventions
 *   - plain C, no exceptions cross the ABI; every function returns int32 status:
 *       0 = GAT_OK, > 0 = argument/state error (GAT_ERR_*), < 0 = -(hipError_t).
 *     gat_last_error(ctx) returns a human-readable message for the last failure on that ctx.
 *   - "dev" pointers are device (HBM) addresses valid on the ctx's device; "host" pointers are
 *     ordinary host memory, read/written only during the call.
 *   - all work is enqueued on the ctx's HIP stream; calls are asynchronous unless stated.
 *     A ctx is not thread-safe; distinct ctxs are independent.
 *   - arrays are column-major as in the reference: signal [N x M] (sample fastest,
 *     src/gen_signal.jl:179), codes [Lc x P] (src/algorithms.jl:185), correlator outputs
 *     [M x L x K x B] (antenna fastest, src/algorithms.jl:628), planar re / im
 *     (StructArray{ComplexF32}).
 *   - indices are 0-based here (sample n = 0 is the reference's sample_idx = 1; prn is the
 *     0-based column of the code table, i.e. reference prn - 1).
 */
#ifndef GAT_H_
#define GAT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(GAT_BUILD)
#define GAT_API __attribute__((visibility("default")))
#else
#define GAT_API
#endif

/* ---- status codes ----------------------------------------------------------------------- */
#define GAT_OK 0
#define GAT_ERR_ARG 1         /* null pointer, non-positive size, bad enum                    */
#define GAT_ERR_RANGE 2       /* prn / phase / size outside the supported range               */
#define GAT_ERR_STATE 3       /* e.g. correlate before gat_set_codes                          */
#define GAT_ERR_UNSUPPORTED 4 /* valid request this build has no kernel for                   */
#define GAT_ERR_NOMEM 5

/* ---- flags for gat_downconvert_and_correlate ------------------------------------------- */
#define GAT_FLAG_ATOMIC 1u /* single-pass float atomics (reference alg. 4/5,                */
                           /* src/algorithms.jl:625-632); default = deterministic two-stage */

#define GAT_FLAG_GRAPH 2u  /* gat_tracking_run and gat_downconvert_and_correlate_dev: replay the call's launch sequence */
                           /* as an instantiated hipGraph when it repeats with the same arguments and buffers (needs a  */
                           /* non-default stream; else stays eager).  Not for the host-parameter entry point.          */

/* ---- signal layouts ---------------------------------------------------------------------- */
#define GAT_LAYOUT_PLANAR 0          /* float32 re[] and im[] planes (reference StructArray)   */
#define GAT_LAYOUT_INTERLEAVED 1     /* ComplexF32 {re,im} pairs; .im must be NULL             */
#define GAT_LAYOUT_INTERLEAVED_I16 2 /* {int16 re, int16 im} pairs (front-end "sc16"), 4 B/sample */
#define GAT_LAYOUT_INTERLEAVED_I8 3  /* {int8 re, int8 im} pairs (front-end "sc8"), 2 B/sample  */

#define GAT_MAX_TAPS 32 /* correlator taps per call (L), any order; reference uses 3 and 7 */

typedef struct gat_ctx gat_ctx;

/* One satellite channel in one integration block.  Replaces the scalar arguments
 * (prn, code_frequency, carrier_frequency, start_code_phase, carrier_phase) of
 * Tracking.downconvert_and_correlate! (call site src/benchmarks.jl:63-79) and of
 * downconvert_and_correlate_kernel_1330! (src/algorithms.jl:142-159). */
typedef struct gat_channel_params {
    int32_t prn;                 /* 0-based column of the code table                         */
    int32_t reserved;            /* must be 0                                                */
    double code_freq_hz;         /* code_frequency (chips/s), incl. code Doppler             */
    double carrier_freq_hz;      /* carrier_frequency = IF + Doppler (Hz)                    */
    double code_phase_chips;     /* start_code_phase at sample 0                             */
    double carrier_phase_cycles; /* carrier_phase at sample 0, in CYCLES (algorithms.jl:172) */
} gat_channel_params;

/* Device-resident antenna signal; replaces signal.re / signal.im (CuArray{Float32,2} or ,3)
 * handed to kernel_algorithm (src/algorithms.jl:887-888) -- element (n, m, b, k) lives at
 *   n + m*ant_stride + b*block_stride + k*chan_stride          (in samples). */
/* Fast path requirements (checked per call; a signal that misses them is still correlated, by the scalar-load kernel --
 * one antenna per wave, a double-precision sincos per sample, typically 5-10x slower; gat_launch_info.vec tells which ran):
 * plane base pointers 16-byte aligned, and ant_stride, block_stride and chan_stride multiples of the samples one 16-byte
 * load holds -- 4 (planar float), 2 (ComplexF32 pairs), 4 (int16 pairs), 8 (int8 pairs) --, i.e. every block of every
 * antenna STARTS on a 16-byte boundary.  num_samples itself may be anything (N = 2046 at fs = 2 x 1.023 MHz, N = 2500 as
 * int8 pairs ...): a receiver with such a block length pads block_stride to the next multiple, nothing else. */
typedef struct gat_signal_desc {
    const void *re;       /* dev; planar: float real plane. interleaved formats: base pointer */
    const void *im;       /* dev; planar: float imaginary plane. interleaved formats: NULL    */
    int32_t layout;       /* GAT_LAYOUT_*                                                     */
    int32_t num_ants;     /* M                                                                */
    int64_t num_samples;  /* N per integration block                                          */
    int64_t ant_stride;   /* samples between antennas (>= extent of one antenna's stream)     */
    int64_t block_stride; /* samples between consecutive integration blocks (normally N)     */
    int64_t chan_stride;  /* 0: every channel reads the same signal; != 0: one signal per     */
                          /* channel as in _3d_4431! (src/algorithms.jl:668)                 */
} gat_signal_desc;

/* ---- context ------------------------------------------------------------------------------
 * Replaces the implicit CUDA.jl task-local device + default stream (CUDA.@sync at
 * src/benchmarks.jl:120).  `hip_stream` is a hipStream_t the caller keeps alive (e.g. PyTorch's
 * current stream); NULL is the HIP default (null) stream -- which is also what PyTorch's default
 * stream is; GAT_OWN_STREAM asks the library to create (and own) a non-blocking stream. */
#define GAT_OWN_STREAM ((void *)(intptr_t)-1)
GAT_API int32_t gat_create(int32_t device, void *hip_stream, gat_ctx **out_ctx);
GAT_API int32_t gat_destroy(gat_ctx *ctx);
GAT_API int32_t gat_set_stream(gat_ctx *ctx, void *hip_stream);
GAT_API int32_t gat_sync(gat_ctx *ctx); /* CUDA.@sync equivalent: wait for the ctx stream  */
GAT_API const char *gat_last_error(const gat_ctx *ctx);
/* "libgat <version> (gfx950) git:<short commit of the kernel sources>[+dirty] flags:<-D flags beyond the product recipe | none>":
 * which library a result came from (the reference tags saved results with its commit: @tagsave,
 * scripts/run_benchmarks_gpsl1.jl:24-27).  A product build says "flags:none"; development builds (-DGAT_DEV: A/B knobs
 * from the environment, diagnostic kernels) name theirs. */
GAT_API const char *gat_version(void);

/* Device properties used for metadata (add_metadata!, src/benchmarks.jl:11-32: GPU_model, CUDA
 * version).  name_buf receives the device name; *runtime_version the HIP runtime version. */
GAT_API int32_t gat_device_info(gat_ctx *ctx, char *name_buf, size_t name_len,
                                int32_t *runtime_version, int32_t *num_cus);

/* ---- code tables --------------------------------------------------------------------------
 * Replaces `system.codes` (GNSSSignals.jl; src/benchmarks.jl:93, src/algorithms.jl:185):
 * int8 +-1 chips, column-major [code_length x num_prns], copied to the device once. */
GAT_API int32_t gat_set_codes(gat_ctx *ctx, const int8_t *codes_host, int32_t code_length,
                              int32_t num_prns);

/* Host-side PRN generators standing in for GNSSSignals.GPSL1()/GPSL5() (src/
 * GPUAcceleratedTracking.jl:39-42).  system = "GPSL1" (1023 chips, 1.023 Mcps, PRN 1..37) or
 * "GPSL5" (I5, 10230 chips, 10.23 Mcps, PRN 1..37).  out_codes may be NULL to query sizes. */
GAT_API int32_t gat_gen_codes(const char *system, int32_t num_prns, int8_t *out_codes_host,
                              int32_t *code_length, double *code_freq_hz);

/* get_correlator_sample_shifts(system, correlator, fs, preferred_code_shift)
 * (src/benchmarks.jl:105-107): s = max(1, round(spacing*fs/fc)); shifts[l] = (l - L/2)*s. */
GAT_API int32_t gat_sample_shifts(int32_t num_taps, double sampling_freq_hz, double code_freq_hz,
                                  double spacing_chips, int32_t *shifts_host);

/* ---- the hot path -------------------------------------------------------------------------
 * Tracking.downconvert_and_correlate!(system, signal, correlator, code_replica, code_phase,
 *   carrier_replica, carrier_phase, downconverted_signal, code_frequency,
 *   correlator_sample_shifts, carrier_frequency, sampling_frequency, start_sample,
 *   num_samples, prn)                                           (src/benchmarks.jl:63-79)
 * == kernel_algorithm(..., ::KernelAlgorithm{1330|1331|1431|2xxx|3431|4431|5431})
 *                                                               (src/algorithms.jl:869-1545)
 * == downconvert_and_correlate_kernel_3d_4431! for num_channels > 1 (src/algorithms.jl:637).
 *
 *   R[m,l,k,b] = sum_{n<N} x[n,m,b] * conj(exp(j2pi(n*f/fs + phi))) * c[floor(fc/fs*(n+shift_l)+tau) mod Lc]
 *
 * One fused launch (plus a tiny finalize launch when one block's samples are split over
 * several workgroups).  Outputs are OVERWRITTEN (the reference's `+=` into stale buffers,
 * src/algorithms.jl:207, is not reproduced).  params: [num_channels x num_blocks], channel
 * fastest.  out_re/out_im: dev float [M x L x K x B].  The replica / carrier / downconverted
 * scratch arguments of the reference call do not exist: nothing is materialised.
 * params_host is validated (GAT_ERR_RANGE / GAT_ERR_ARG instead of NaN results) and consumed before the call returns: up
 * to 4 records (num_blocks * num_channels <= 4: the single-block call of a receiver loop, the reference's kernel_algorithm
 * with scalar arguments) travel inside the kernel arguments -- no upload in front of the launch --, more are copied to a
 * context-owned device buffer on the context's stream. */
GAT_API int32_t gat_downconvert_and_correlate(gat_ctx *ctx, const gat_signal_desc *signal,
                                              const gat_channel_params *params_host,
                                              int32_t num_blocks, int32_t num_channels,
                                              int32_t num_taps, const int32_t *shifts_host,
                                              double sampling_freq_hz, float *out_re_dev,
                                              float *out_im_dev, uint32_t flags);

/* Same, with the per-(block, channel) parameters already resident on the device (a tracking
 * loop that produces them on-GPU; also the form bench.py times). */
GAT_API int32_t gat_downconvert_and_correlate_dev(gat_ctx *ctx, const gat_signal_desc *signal,
                                                  const gat_channel_params *params_dev,
                                                  int32_t num_blocks, int32_t num_channels,
                                                  int32_t num_taps, const int32_t *shifts_host,
                                                  double sampling_freq_hz, float *out_re_dev,
                                                  float *out_im_dev, uint32_t flags);

/* ---- stand-alone operators on the same path ---------------------------------------------- */

/* Tracking.gen_code_replica!(code_replica, system, code_frequency, sampling_frequency,
 *   start_code_phase, start_sample, num_samples, correlator_sample_shifts, prn)
 *   (scripts/code_replica_experiment.jl:70) == gen_code_replica_kernel! (src/algorithms.jl:13)
 * rep[i] = c[floor(fc/fs*(i + first_shift) + tau) mod Lc], i = 0..count-1, as float32 +-1. */
GAT_API int32_t gat_gen_code_replica(gat_ctx *ctx, float *replica_dev, int64_t count, int32_t prn,
                                     double code_freq_hz, double sampling_freq_hz,
                                     double code_phase_chips, int64_t first_shift);

/* gen_code_replica_texture_mem_kernel! (src/algorithms.jl:121-140): the same replica addressed
 * through a Float32 NORMALISED coordinate (phase / code_length rounded to float32, wrap, nearest
 * texel) -- an emulation of the arithmetic behind the texture path's code-phase error
 * (paper/paper.tex:318-331; scripts/code_replica_experiment.jl).  Study use only: the
 * correlator never uses it. */
GAT_API int32_t gat_gen_code_replica_f32coord(gat_ctx *ctx, float *replica_dev, int64_t count,
                                              int32_t prn, double code_freq_hz,
                                              double sampling_freq_hz, double code_phase_chips,
                                              int64_t first_shift);

/* The same study with the texture unit's FIXED-POINT addressing modelled (paper/paper.tex:322-331 reports min 0 / mean
 * 0.03 / median 0.02 / max 3.17 % relative code-phase error; Float32 rounding of the coordinate alone gives 1/16 of the
 * mean and 1/25 of the maximum).  From the Float32 normalised coordinate u = float32(phase / code_length), wrapped to
 * w = u - floor(u):
 *   coord_frac_bits > 0 : w is TRUNCATED to that many fractional bits (the unit's fixed-point normalised coordinate);
 *   x = w * code_length, exact (no Float32 rounding of the product);
 *   texel_frac_bits >= 0: x is ROUNDED to nearest at that many fractional bits (a fixed-point texel address; 8 = the
 *                         sub-texel precision CUDA documents for its filter weights); -1: left as it is;
 *   chip = floor(x).
 * (0, -1) is Float32 rounding of the coordinate alone with an exact product.  Study use only
 * (scripts/code_replica_experiment.py runs the reference's sweep over these modes). */
GAT_API int32_t gat_gen_code_replica_texaddr(gat_ctx *ctx, float *replica_dev, int64_t count, int32_t prn,
                                             double code_freq_hz, double sampling_freq_hz, double code_phase_chips,
                                             int64_t first_shift, int32_t coord_frac_bits, int32_t texel_frac_bits);

/* gen_code_replica_texture_mem_strided_nsat_kernel! (src/algorithms.jl:78-98): the replicas of num_channels
 * satellite channels in ONE call -- row k (row_stride floats apart) = channel k with its own prn, code_freq_hz and
 * code_phase_chips (params_dev[k]; the carrier fields are ignored); rep[k][i] = c_k[floor(fc_k/fs*(i + first_shift)
 * + tau_k) mod Lc].  Exact index arithmetic (prn is a table column here, not the reference's normalised texture
 * coordinate, SURVEY defect D5).  A prn outside the table poisons its row with NaN. */
GAT_API int32_t gat_gen_code_replica_multi(gat_ctx *ctx, float *replica_dev, int64_t count, int64_t row_stride,
                                           int32_t num_channels, const gat_channel_params *params_dev,
                                           double sampling_freq_hz, int64_t first_shift);

/* downconvert_and_accumulate_strided_kernel! (src/algorithms.jl:828-866), the materialising middle stage of the
 * reference's algorithm 2, as a DEBUG export: for one integration block (planar float signal) and one channel it
 * writes what the fused correlator never materialises -- carrier replica [N], downconverted signal [N x M]
 * (sample fastest) and the per-sample products [N x M x L] -- so that the reference's test of that stage
 * (test/algorithms.jl:1438-1514: prompt products == 1, column sums == [1476 2500 1476]) has a counterpart.
 * Any output pointer may be NULL.  Not a fast path: 8*N*(1 + M + M*L) bytes of stores for 8*N*M bytes of signal. */
GAT_API int32_t gat_downconvert_and_accumulate(gat_ctx *ctx, const gat_signal_desc *signal,
                                               const gat_channel_params *params_host, int32_t num_taps,
                                               const int32_t *shifts_host, double sampling_freq_hz,
                                               float *carrier_re_dev, float *carrier_im_dev, float *dw_re_dev,
                                               float *dw_im_dev, float *accum_re_dev, float *accum_im_dev);

/* gen_signal! (src/gen_signal.jl:53-175): noise-free synthetic IF signal, identical on every
 * antenna.  Writes, for every block b, x[n,m,b] = sum_k c_k[floor(fc_k/fs*n + tau_kb) mod Lc]
 *   * (cos, sin)(float32(2pi*n*f_kb/fs + phase_kb)).  NOTE: for THIS call the
 * carrier_phase_cycles field is interpreted in RADIANS, as start_carrier_phase is in the
 * reference (src/gen_signal.jl:88).  params_dev: [num_channels x num_blocks].  `amplitude`
 * scales the sum (1.0 = the reference); the integer layouts store rint(amplitude * x),
 * saturated -- e.g. amplitude 2000 for int16, 40 for int8. */
GAT_API int32_t gat_gen_signal(gat_ctx *ctx, void *re_dev, void *im_dev, int32_t layout,
                               int64_t num_samples, int32_t num_ants, int64_t ant_stride,
                               int64_t block_stride, int32_t num_blocks, int32_t num_channels,
                               const gat_channel_params *params_dev, double sampling_freq_hz,
                               double amplitude);

/* The same generator with what a receiver's input has and the reference's noise-free one lacks (paper/paper.tex:116;
 * SURVEY section 8-d "build additions"): a unit-modulus steering phase per antenna (steering_cycles_dev: M floats in cycles on
 * the device, or NULL: identical antennas) and complex white Gaussian noise of standard deviation noise_sigma per component
 * (in units of one satellite's amplitude; scaled by `amplitude` like the signal).  The noise of sample (block, antenna, n) is a
 * pure function of (seed, block, antenna, n) -- counter-based, independent of the launch geometry and of the layout. */
GAT_API int32_t gat_gen_signal_noisy(gat_ctx *ctx, void *re_dev, void *im_dev, int32_t layout,
                                     int64_t num_samples, int32_t num_ants, int64_t ant_stride,
                                     int64_t block_stride, int32_t num_blocks, int32_t num_channels,
                                     const gat_channel_params *params_dev, double sampling_freq_hz,
                                     double amplitude, const float *steering_cycles_dev, double noise_sigma,
                                     uint64_t seed);

/* reduce_cplx_multi_3/4/5 two-pass sum (src/reduction.jl:93, :331, :548; launch sequence
 * src/algorithms.jl:914-922): column sums of a planar complex [n x num_cols] array.
 * Deterministic; no single-block second pass limit (SURVEY defect D6). */
GAT_API int32_t gat_reduce_cplx_multi(gat_ctx *ctx, const float *in_re_dev, const float *in_im_dev,
                                      int64_t n, int32_t num_cols, float *out_re_dev,
                                      float *out_im_dev);

/* ---- closed tracking-loop step ------------------------------------
 * What Tracking.jl's `track` does around the correlator (the reference touches that layer only
 * to borrow buffers: TrackingState, src/benchmarks.jl:54-61): discriminators + loop filters turn
 * the accumulators of the block just correlated into the parameters of the next block, ON THE
 * DEVICE, so a receiver loop is {correlate, update} with no host round trip.  Textbook loop
 * (Kaplan & Hegarty ch. 5; the same structure Tracking.jl uses):
 *   prompt/early/late = sum over antennas of R[m, tap]                      (identical antennas)
 *   PLL: Costas arctan discriminator atan(Q/I)/2pi [cycles] -> 3rd-order bilinear loop filter
 *   DLL: (2-d)/2 * (|L|-|E|)/(|E|+|L|) [chips], d = E-L spacing in chips -> 2nd-order bilinear
 *        filter, carrier aided (code_doppler += carrier_doppler * code_freq / carrier_center)
 *   next block: carrier_phase += carrier_freq * T, code_phase += code_freq * T (mod code_length),
 *               carrier_freq = if_hz + doppler, code_freq = nominal + code_doppler
 * No reference implementation of this step is available here (Tracking.jl is un-vendored):
 * parity is against the oracle's restatement of the same equations only ("unpinned"). */
typedef struct gat_loop_config {
    double block_seconds;        /* T = N / fs                                                 */
    double pll_bandwidth_hz;     /* carrier loop noise bandwidth (e.g. 18 Hz)                  */
    double dll_bandwidth_hz;     /* code loop noise bandwidth (e.g. 1 Hz)                      */
    double code_freq_nominal_hz; /* e.g. 1.023e6                                               */
    double carrier_center_hz;    /* RF centre frequency for carrier aiding (e.g. 1575.42e6)    */
    double if_hz;                /* intermediate frequency of the sampled signal               */
    double early_late_spacing_chips; /* d: distance between the early and the late tap in chips */
    int32_t code_length;
    int32_t num_taps, early_index, prompt_index, late_index; /* positions in the tap list      */
} gat_loop_config;

typedef struct gat_loop_state { /* one per channel, device-resident; zero everything but      */
    double init_carrier_doppler_hz; /* ... this: the acquisition estimate the loop starts from */
    double carrier_doppler_hz;  /* out: current carrier Doppler estimate                       */
    double code_doppler_hz;     /* out: current code Doppler estimate                          */
    double pll_acc1, pll_acc2;  /* loop-filter integrators                                     */
    double dll_acc;
    double last_pll_error_cycles, last_dll_error_chips; /* diagnostics                         */
    double prompt_power;        /* |P|^2 of the last block                                     */
} gat_loop_state;

/* acc_re/acc_im: dev float [M x L x K] of ONE block (as written by the correlator);
 * cur/next: dev gat_channel_params[K] (may alias); state: dev gat_loop_state[K]. */
GAT_API int32_t gat_tracking_update(gat_ctx *ctx, const float *acc_re_dev, const float *acc_im_dev,
                                    int32_t num_channels, int32_t num_ants,
                                    const gat_loop_config *config_host, gat_loop_state *state_dev,
                                    const gat_channel_params *cur_dev, gat_channel_params *next_dev);

/* The same update on the HOST (the same arithmetic: csrc/gat_loop.h), for a receiver that takes its correlator outputs on
 * the host -- from a resident correlator, below -- and closes its loops there, as Tracking.jl does.  All pointers are host
 * memory; next_host may alias cur_host.  Needs no context and no device. */
GAT_API int32_t gat_tracking_update_host(const float *acc_re_host, const float *acc_im_host, int32_t num_channels,
                                         int32_t num_ants, const gat_loop_config *config_host, gat_loop_state *state_host,
                                         const gat_channel_params *cur_host, gat_channel_params *next_host);

/* num_blocks consecutive integration blocks of a device-resident signal (block b starts b * sig->block_stride
 * samples in) through {correlate, tracking update} with the parameters ping-ponging between params_a (current at
 * entry) and params_b: 2 * num_blocks launches enqueued from native code, no host round trip and no
 * synchronisation (a per-block call from a scripting host is launch-bound: ~16 us per 1 ms block).  Block b's
 * accumulators go to acc_re/acc_im + b * acc_block_stride floats ([M x L x K] each; stride 0 keeps only the last
 * block's).  *current_is_b tells which buffer holds the parameters for the block after the last one.
 * flags: GAT_FLAG_ATOMIC as for the correlator; GAT_FLAG_GRAPH: the first call with a given argument set runs
 * eagerly and records the sequence, later calls with the same arguments replay it with one hipGraphLaunch. */
GAT_API int32_t gat_tracking_run(gat_ctx *ctx, const gat_signal_desc *sig, int32_t num_blocks, int32_t num_channels,
                                 int32_t num_taps, const int32_t *shifts_host, double sampling_freq_hz,
                                 const gat_loop_config *config_host, gat_loop_state *state_dev,
                                 gat_channel_params *params_a_dev, gat_channel_params *params_b_dev,
                                 float *acc_re_dev, float *acc_im_dev, int64_t acc_block_stride, uint32_t flags,
                                 int32_t *current_is_b);

/* ---- memory + timing helpers (for hosts without their own HIP array type) ---------------- */
GAT_API int32_t gat_malloc(gat_ctx *ctx, size_t bytes, void **out_dev);
GAT_API int32_t gat_free(gat_ctx *ctx, void *dev);
GAT_API int32_t gat_memcpy_h2d(gat_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
GAT_API int32_t gat_memcpy_d2h(gat_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
GAT_API int32_t gat_memset(gat_ctx *ctx, void *dst_dev, int32_t value, size_t bytes);

/* hipEvent pair on the ctx stream (the reference times with BenchmarkTools around CUDA.@sync,
 * src/benchmarks.jl:120; CUDA.@elapsed in test/algorithms.jl:1242).  stop() synchronises. */
GAT_API int32_t gat_timer_start(gat_ctx *ctx);
GAT_API int32_t gat_timer_stop(gat_ctx *ctx, float *elapsed_ms);
/* Per-call statistics as the reference's harness keeps them (BenchmarkTools stores every sample's time: Minimum / Median /
 * Mean / sigma / Maximum + RawTimes, src/benchmarks.jl:1-9): gat_timer_lap records ONE event on the ctx stream -- call it
 * before the first timed launch and after every launch --; gat_timer_laps waits for the newest lap and returns the
 * intervals between consecutive laps in milliseconds (laps - 1 of them, at most `capacity`), then forgets the laps.  The
 * events come from a pool the context owns (it grows to the largest number of laps ever outstanding). */
GAT_API int32_t gat_timer_lap(gat_ctx *ctx);
GAT_API int32_t gat_timer_laps(gat_ctx *ctx, float *intervals_ms, int32_t capacity, int32_t *num_intervals);

/* Measurement aid ("also report vs the measured read ceiling"): a kernel that ONLY reads -- every lane
 * one 16-byte load per step over `bytes` of device memory (a multiple of 16), summed into a value nobody stores -- launched
 * `launches` times on the ctx stream, each launch timed by its own event pair (ms_each_host[launches]).  `variant` picks
 * the reader: bits 0-1 workgroups per CU (0: 8, 1: 16, 2: 32, 3: 64), bit 2 plain instead of non-temporal loads, bit 3
 * four instead of eight loads in flight per lane.  What the best variant reaches over the correlator's own stream is the
 * memory system's ceiling for a read-once kernel on this device, in this process, at this moment (bench.py:
 * roofline.read_ceiling_GBps; tests: the box-tolerant performance guard).  Not part of the reference's surface. */
GAT_API int32_t gat_debug_read_stream(gat_ctx *ctx, const void *dev, size_t bytes, int32_t variant, int32_t launches,
                                      float *ms_each_host);

/* Kernel selection.  By default (GAT_MC_AUTO) the library runs the split-bf16 matrix-core kernel (gat_mfma_bf16.hip:
 * both operands as hi+mid+lo bf16 terms, f32-equivalent accuracy; every sample format) where it measured faster than
 * the vector kernel -- M % 16 == 0 and at least 24 (channel, tap, re/im) columns, then by the share of the kernel's
 * 32-column tile slots that carry live columns: float samples (three bf16 terms per value) from 0.70 on at M % 64 == 0
 * (64 antennas x 16 or 32 channels) and 0.90 at M % 32 == 0; int16 pairs (two exact terms, 5/8 of the matrix work: round 5)
 * from 0.5 on at M % 32 == 0 (32 antennas x 8 channels); int8 pairs always; and only launches long enough to give each of
 * the kernel's ~2 workgroups per CU 16 steps (int16, int8) or 32 (float) of 32-128 samples: ONE 1 ms block stays on the
 * vector kernel whatever its shape --, everything else on the vector kernel.
 * GAT_MC_VECTOR forces the vector kernel (A/B measurements, bit-comparisons), GAT_MC_F32 the
 * f32-MFMA kernel (gat_mfma.hip), GAT_MC_BF16_SPLIT the split-bf16 kernel only (shapes neither
 * matrix kernel takes fall through to the vector kernel in every mode). */
#define GAT_MC_VECTOR 0
#define GAT_MC_AUTO 1
#define GAT_MC_F32 2
#define GAT_MC_BF16_SPLIT 3
GAT_API int32_t gat_set_matrix_core(gat_ctx *ctx, int32_t enable);

/* Caps on the vector kernel's workgroup tiling (0 = leave a cap unchanged).  By default a workgroup covers up to 4
 * antenna tiles (16 antennas: carrier and replica are produced once per workgroup), loops over up to 4 channels with
 * the samples held in registers, and over several consecutive short blocks.  (1, 1, 1) gives one antenna tile, one
 * channel and one block per workgroup -- the round-1 geometry; used by A/B measurements and by the parity tests,
 * which run every tiling against the oracle. */
GAT_API int32_t gat_set_vector_tiling(gat_ctx *ctx, int32_t max_antenna_tiles, int32_t max_channels,
                                      int32_t max_blocks);

/* Launch-geometry options by name, for tests (force a code path on a small case) and A/B measurements; none changes a
 * result beyond summation order.  The library reads NO environment variable that selects kernels or geometry (development
 * builds, -DGAT_DEV, map GAT_<NAME> onto these).  Names: "sync_flag_wgs" (largest launch in workgroups that carries the
 * completion flag; 0: never), "max_ant_tile", "dc_aw", "dc_kt", "dc_bpw" (the caps of gat_set_vector_tiling),
 * "dc_bpw_force", "dc_wgs_per_cu", "dc_one_wave", "dc_one_wave_min", "dc_ow_seg", "dc_depth", "dc_keep_l2", "dc_align",
 * "dc_aw2", "dc_quads", "dc_bits", "dc_seg" (round 5: the two-channel tile and its replica fill), "mc_i16_terms" (split-bf16
 * kernel, int16 samples: 2 = the exact two-term split, the default; 3 = the float path's three terms: bit-comparisons),
 * "mc_nct" (32-column tiles per workgroup of the split-bf16 kernel to try first; 0 = by rule).
 * GAT_ERR_ARG: unknown name; GAT_ERR_RANGE: value outside the option's range. */
GAT_API int32_t gat_set_option(gat_ctx *ctx, const char *name, int64_t value);

/* Launch geometry chosen for the last correlate call (diagnostics / DESIGN.md tables). */
typedef struct gat_launch_info {
    int32_t workgroups, threads, splits, ant_tile, vec, lds_bytes, finalize_launched;
    int32_t matrix_core; /* 0: the vector kernel ran, 1: the f32-MFMA kernel, 2: the split-bf16 MFMA kernel */
    int32_t channels_per_wg; /* channels one workgroup loops over (signal held in registers / LDS meanwhile)  */
    int32_t blocks_per_wg;   /* consecutive integration blocks one workgroup loops over                       */
    int32_t prefetch_depth;  /* vector kernel: register sets of samples per wave (steps in flight); else 0      */
    int32_t bf16_terms;      /* split-bf16 kernel: bf16 terms per sample value (1: int8, 2: int16, 3: float); else 0 */
} gat_launch_info;
/* struct_size = sizeof(gat_launch_info) of the CALLER's header: the struct grows at its end between versions and the
 * library copies no more than the caller has room for. */
GAT_API int32_t gat_last_launch_info(const gat_ctx *ctx, gat_launch_info *out, size_t struct_size);

/* ---- resident correlator: single-block calls without a kernel launch -------------------------------------------
 * The reference's benchmark is ONE 1 ms block per call, synchronised (`@benchmark CUDA.@sync kernel_algorithm(...)`,
 * src/benchmarks.jl:120-146), and its receiver loop consumes every block's correlator outputs on the host
 * (Tracking.jl's discriminators).  Through gat_downconvert_and_correlate + gat_sync such a call costs a kernel launch and
 * the wait for its end: 7 us on this platform before the kernel has done anything.  A resident correlator keeps ONE
 * kernel on the device for a fixed call geometry; a call rings a doorbell (the channel records and the block's position
 * travel with the ring), every workgroup of the kernel correlates its share and posts its sums to pinned host memory
 * stamped with the call's number, the host adds them in a fixed order: no launch, no stream wait, outputs already on
 * the host (2 MHz .. 8 MHz blocks: 5-6 us instead of 10.5-13.5; a 20 MHz block of four antennas and twelve channels:
 * 9 us instead of 16 -- DESIGN.md 4.2b).
 *
 * Lifetime is bounded on the DEVICE side, whatever the host does: the kernel ends by itself after `idle_us` without a
 * call, after `life_ms` in total, or after `max_calls` calls; the next call starts it again (that call then costs a
 * launch).  It occupies one workgroup slot per workgroup it uses (info.workgroups: about max_workgroups at most) and polls
 * its doorbell while it waits (in host memory: from up to `host_pollers` of them).  Other work of the process runs next to it on other streams; calls that
 * synchronise the whole device (hipDeviceSynchronize, hipFree) wait until it has ended: gat_free, gat_set_codes and
 * gat_destroy therefore ask every resident correlator of the context to leave first (gat_set_codes: for good -- the
 * correlator answers GAT_ERR_STATE afterwards and has to be opened again).
 *
 * Geometry (fixed at open): `signal` describes ONE block (num_samples, num_ants, ant_stride, layout; block_stride and
 * chan_stride as for the correlator) at the START of a device buffer; every call names its block by an offset in
 * samples from there (a ring buffer of blocks, or always 0).  Supported: what gat_downconvert_and_correlate would run as
 * ONE launch of the vector kernel -- any sample layout, block starts 16-byte aligned and num_samples a multiple of the
 * samples one 16-byte load holds (4 / 2 / 4 / 8 by layout), num_channels <= 16, at most 8 taps within a span of 2048
 * samples -- else GAT_ERR_UNSUPPORTED (use the ordinary call).  The caller makes sure the block's samples are in
 * device memory before the call (e.g. gat_sync after the copy that brought them); the kernel reads them past its caches.
 * Results: host arrays [M x L x K], the ordinary call's layout for one block; same values as the ordinary call up to the
 * summation order of the per-workgroup partial sums. */
typedef struct gat_resident gat_resident;
typedef struct gat_resident_config {
    uint32_t struct_size;    /* sizeof(gat_resident_config) of the caller's header                               */
    uint32_t idle_us;        /* the kernel ends after this long without a call          (0: default, 5 000 us)   */
    uint32_t life_ms;        /* ... and after this long whatever happens                (0: default, 2 000 ms)   */
    uint32_t max_calls;      /* ... and after this many calls                           (0: no limit)            */
    uint32_t max_workgroups; /* workgroups one block's samples may be split over   (0: default, 128 -- 64 with the    */
                             /* doorbell in host memory; at most the device's compute units: all of them have to be  */
                             /* on the device at once)                                                           */
    uint32_t host_pollers;   /* doorbell in host memory: up to this many workgroups poll it themselves; with more, one     */
                             /* does and forwards the ring through device memory        (0: default, 20)         */
    uint32_t doorbell;       /* where the doorbell lives: 0 device memory written through the PCIe BAR where the device   */
                             /* has a large BAR, else pinned host memory; 1 pinned host memory; 2 device memory          */
} gat_resident_config;
typedef struct gat_resident_info {
    int32_t workgroups;  /* workgroups of the resident kernel (sample splits x antenna tiles x channels)          */
    int32_t splits;      /* sample splits of one block                                                            */
    int32_t running;     /* 1: a kernel has been started and has not been seen to end                             */
    int32_t last_exit;   /* why the last kernel that ended did: 0 none yet, 1 asked to, 2 idle, 3 lifetime, 4 calls */
    uint64_t launches;   /* kernels started so far (1 + the restarts)                                             */
    uint64_t calls;      /* calls served                                                                          */
} gat_resident_info;
GAT_API int32_t gat_resident_open(gat_ctx *ctx, const gat_signal_desc *signal, int32_t num_channels, int32_t num_taps,
                                  const int32_t *shifts_host, double sampling_freq_hz, const gat_resident_config *config,
                                  gat_resident **out_resident);
/* One call = gat_downconvert_and_correlate(..., num_blocks = 1, ...) + gat_sync + the copy of the outputs to the host.
 * params_host: num_channels records, validated as by the host entry point.  block_offset_samples: >= 0, a multiple of
 * the samples one 16-byte load holds.  Blocks until the results are in out_re_host / out_im_host. */
GAT_API int32_t gat_resident_correlate(gat_resident *resident, const gat_channel_params *params_host,
                                       int64_t block_offset_samples, float *out_re_host, float *out_im_host);
/* The receiver loop with the host in it, from native code: num_blocks consecutive integration blocks (block b starts
 * first_block_offset + b * block_stride_samples samples into the correlator's buffer; the caller has put all of them on
 * the device) through {gat_resident_correlate, gat_tracking_update_host} -- Tracking.jl's structure: correlate on the
 * accelerator, discriminators and loop filters on the CPU -- without a trip through the scripting host per block.
 * params_host[K]: in the first block's parameters, out those for the block after the last; state_host[K] as for
 * gat_tracking_update_host.  Block b's accumulators go to acc_re/acc_im_host + b * acc_block_stride floats ([M x L x K]
 * each; stride 0 keeps only the last block's).  An error ends the run at the block it occurred in: params_host and
 * state_host then hold the loop's state in front of that block. */
GAT_API int32_t gat_resident_tracking_run(gat_resident *resident, int32_t num_blocks, int64_t first_block_offset,
                                          int64_t block_stride_samples, const gat_loop_config *config_host,
                                          gat_loop_state *state_host, gat_channel_params *params_host,
                                          float *acc_re_host, float *acc_im_host, int64_t acc_block_stride);
GAT_API int32_t gat_resident_info_get(const gat_resident *resident, gat_resident_info *out, size_t struct_size);
/* asks the kernel to leave and waits until it has (bounded by the kernel's own limits); the next call starts it again */
GAT_API int32_t gat_resident_park(gat_resident *resident);
/* the same for every resident correlator of the context: call it before anything that waits for the WHOLE device
 * (hipDeviceSynchronize, torch.cuda.synchronize(), hipFree of a foreign allocation) -- such a call otherwise sits out the
 * resident kernels' idle limit (5 ms by default).  gat_free, gat_set_codes and gat_destroy do it themselves. */
GAT_API int32_t gat_resident_park_all(gat_ctx *ctx);
/* (a correlator that is still open when its context is destroyed is closed by gat_destroy: its handle dies with the context) */
GAT_API int32_t gat_resident_close(gat_resident *resident);

/* ---- several devices from one host thread --------------------------------------------------
 * The reference is single-device (its only device call is CUDA.CuDevice(0) for the GPU's name, src/benchmarks.jl:24;
 * "parallelization over multiple channels can be easily extended", paper/paper.tex:114).  Satellite channels are
 * independent given the antenna signal, so they shard with NO collective: every device holds the full signal
 * (replicated by peer copies -- xGMI between the GPUs of one node; 6.4 MB per ms at 16 antennas x 50 MHz is 4 % of one
 * link) and correlates a contiguous slice of the channels; outputs are disjoint.  A group is a set of ordinary
 * contexts (one per member, each with its own stream: the members' launches of one call overlap); a member's context
 * is available for every single-device function above (allocation, gat_gen_signal, timers ...).  The same device
 * may appear several times (two members on device 0 rehearse the whole path on a one-GPU box). */
typedef struct gat_group gat_group;
GAT_API int32_t gat_device_count(int32_t *count);
/* devices: num_members device ordinals, or NULL for 0 .. num_members-1 */
GAT_API int32_t gat_group_create(int32_t num_members, const int32_t *devices, gat_group **out_group);
GAT_API int32_t gat_group_destroy(gat_group *group);
GAT_API int32_t gat_group_size(const gat_group *group, int32_t *num_members);
GAT_API int32_t gat_group_ctx(gat_group *group, int32_t rank, gat_ctx **ctx); /* borrowed: destroyed with the group */
GAT_API const char *gat_group_last_error(const gat_group *group);
/* the contiguous channel slice of member `rank`: channels [first, first + count) of num_channels (count may be 0) */
GAT_API int32_t gat_group_shard(const gat_group *group, int32_t num_channels, int32_t rank, int32_t *first,
                                int32_t *count);
GAT_API int32_t gat_group_set_codes(gat_group *group, const int8_t *codes_host, int32_t code_length,
                                    int32_t num_prns);
/* dst (on dst_ctx's device) <- src (on src_ctx's device), asynchronous: the copy runs on dst_ctx's stream after
 * everything enqueued so far on src_ctx's stream (hipMemcpyPeerAsync; a plain device copy when both are one device),
 * and work enqueued on src_ctx's stream AFTER this call waits for the copy: the source buffer may be refilled on its
 * own stream right away (a receiver replicating every millisecond needs no group-wide sync between blocks). */
GAT_API int32_t gat_memcpy_peer(gat_ctx *dst_ctx, void *dst_dev, gat_ctx *src_ctx, const void *src_dev, size_t bytes);
/* bufs_dev[r] (r != src_rank) <- bufs_dev[src_rank], `bytes` each: the ingest device's signal to its peers */
GAT_API int32_t gat_group_replicate(gat_group *group, int32_t src_rank, void *const *bufs_dev, size_t bytes);
/* gat_downconvert_and_correlate over the group: params_host [num_channels x num_blocks] (channel fastest) for ALL
 * channels; member r correlates its slice on signals[r] (its own copy of the signal) into out_*_dev[r], a device
 * buffer [M x L x count_r x B] on ITS device.  Asynchronous on every member's stream. */
GAT_API int32_t gat_group_correlate(gat_group *group, const gat_signal_desc *signals,
                                    const gat_channel_params *params_host, int32_t num_blocks, int32_t num_channels,
                                    int32_t num_taps, const int32_t *shifts_host, double sampling_freq_hz,
                                    float *const *out_re_dev, float *const *out_im_dev, uint32_t flags);
/* the members' outputs concatenated along the channel axis into host arrays [M x L x K x B]; synchronises */
GAT_API int32_t gat_group_gather(gat_group *group, float *const *out_re_dev, float *const *out_im_dev,
                                 int32_t num_blocks, int32_t num_channels, int32_t num_taps, int32_t num_ants,
                                 float *out_re_host, float *out_im_host);
GAT_API int32_t gat_group_sync(gat_group *group);

#ifdef __cplusplus
}
#endif
#endif /* GAT_H_ */
