// gat_kernels.hip -- gfx950 (MI355X, wave64) kernels around the fused correlator: second-stage sum, stand-alone
// code replica / signal generators, column-sum reduction, tracking-loop update, and the launch dispatch.
// The fused downconvert + correlate kernel itself is the template in gat_dc.h, instantiated per sample format in
// gat_dc_f0.hip ... gat_dc_f3.hip (four translation units that compile in parallel).
#include "gat_dc.h"
#include "gat_loop.h"

namespace gat {

// second stage: sum of the per-split partials in a FIXED order (deterministic).  Many splits (the matrix-core kernels
// leave 256 per (block, channel) at configs[4]: 25 MB of partials): a workgroup = 32 consecutive output elements x 8
// slices; thread (element, slice) adds splits slice, slice + 8, ... with four loads in flight -- consecutive threads read
// consecutive floats of one split row --, then the 8 slices are added in slice order through LDS.  (Round 2 spent one
// wave64 per output element, lane i adding splits i, i + 64, ...: every load instruction fetched 64 floats 1.5 KB apart,
// 33 us at configs[4]; now 9.)
// partial [groups][splits][elems], elems = cols*2 with re/im interleaved innermost.
constexpr int kFinElems = 32, kFinSlices = kThreads / kFinElems;
__global__ void __launch_bounds__(kThreads)
finalize_kernel(const float *__restrict__ partial, float *__restrict__ out_re,
                float *__restrict__ out_im, int splits, int elems, long long total, unsigned *done_counter,
                unsigned *host_flag, unsigned flag_seq)
{
    __shared__ float s_sum[kFinSlices][kFinElems];
    const int el = threadIdx.x % kFinElems, slice = threadIdx.x / kFinElems;
    const long long o = (long long)blockIdx.x * kFinElems + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (o < total) {
        const long long g = o / elems;
        const int e = (int)(o - g * elems);
        const float *p = partial + (size_t)g * splits * elems + e;
        int i = slice;
        for (; i + 3 * kFinSlices < splits; i += 4 * kFinSlices) {
            const float v0 = p[(size_t)i * elems], v1 = p[(size_t)(i + kFinSlices) * elems];
            const float v2 = p[(size_t)(i + 2 * kFinSlices) * elems], v3 = p[(size_t)(i + 3 * kFinSlices) * elems];
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
        }
        if (i < splits) s0 += p[(size_t)i * elems];
        if (i + kFinSlices < splits) s1 += p[(size_t)(i + kFinSlices) * elems];
        if (i + 2 * kFinSlices < splits) s2 += p[(size_t)(i + 2 * kFinSlices) * elems];
    }
    s_sum[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && o < total) {
        float tot = s_sum[0][el];
#pragma unroll
        for (int j = 1; j < kFinSlices; ++j) tot += s_sum[j][el];
        const long long g = o / elems;
        const int e = (int)(o - g * elems);
        float *out = (e & 1) ? out_im : out_re;
        out[(size_t)g * (elems / 2) + (e >> 1)] = tot;
    }
    completion_flag(done_counter, host_flag, flag_seq, gridDim.x);
}

// Few splits (what the vector kernel leaves behind: 2-32 per group): one THREAD per output element, its splits summed
// in a fixed order (four interleaved chains, then ((0+1)+(2+3)): deterministic).  Consecutive threads read consecutive
// floats of one split row, so every load instruction of a wave is one 256-byte line (round 2's wave-per-element kernel
// kept 4 of its 64 lanes busy at the 16-antenna shard of BASELINE configs[3]: 196 608 waves for 3 MB of partials,
// 40 us of a 0.72 ms step).
__global__ void __launch_bounds__(kThreads)
finalize_few_kernel(const float *__restrict__ partial, float *__restrict__ out_re,
                    float *__restrict__ out_im, int splits, int elems, long long total, unsigned *done_counter,
                    unsigned *host_flag, unsigned flag_seq)
{
    const long long o = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (o < total) {
        const long long g = o / elems;
        const int e = (int)(o - g * elems);
        const float *p = partial + (size_t)g * splits * elems + e;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = 0;
        for (; i + 4 <= splits; i += 4) {
            s0 += p[(size_t)(i + 0) * elems];
            s1 += p[(size_t)(i + 1) * elems];
            s2 += p[(size_t)(i + 2) * elems];
            s3 += p[(size_t)(i + 3) * elems];
        }
        if (i < splits) s0 += p[(size_t)i * elems];
        if (i + 1 < splits) s1 += p[(size_t)(i + 1) * elems];
        if (i + 2 < splits) s2 += p[(size_t)(i + 2) * elems];
        float *out = (e & 1) ? out_im : out_re;
        out[(size_t)g * (elems / 2) + (e >> 1)] = (s0 + s1) + (s2 + s3);
    }
    completion_flag(done_counter, host_flag, flag_seq, gridDim.x);
}

// The ragged end of every block: dc_kernel's vector path covers whole 16-byte load groups (n_vec = N - N % S samples of
// each block, S = 4 / 2 / 4 / 8 by sample format); for a block length that is no multiple of S -- N = 2046 at
// fs = 2 x 1.023 MHz, the reference's N = 2500 fixture as int8 pairs -- this kernel adds the remaining N % S < 8 samples
// to the results, stream-ordered behind dc_kernel and its second stage (one writer per output element: deterministic).
// The reference bounds every thread by num_samples instead (src/algorithms.jl:170).  One thread per (block, channel,
// antenna): the reference's expressions evaluated directly (double-precision code phase, unfused; carrier phase in
// double, float sincos on the reduced argument), chips straight from the global table.  A few hundred threads of a
// few dozen instructions; launched for such block lengths only.
__global__ void __launch_bounds__(kThreads) dc_tail_kernel(const DcTailArgs a)
{
    const long long t = (long long)blockIdx.x * kThreads + threadIdx.x;
    const long long total = (long long)a.B * a.K * a.M;
    if (t < total) {
        const int m = (int)(t % a.M);
        const long long bk = t / a.M;
        const int k = (int)(bk % a.K);
        const long long b = bk / a.K;
        const gat_channel_params P = a.params ? a.params[bk] : a.inl[bk];
        const double ratio = P.code_freq_hz / a.fs, tau = P.code_phase_chips; // src/algorithms.jl:179
        const double step = P.carrier_freq_hz / a.fs, phi = P.carrier_phase_cycles;
        const int N = (int)a.N;
        // the same predicate as dc_kernel: such a channel's outputs are NaN already, nothing to add (and nothing indexed)
        const bool bad = P.prn < 0 || P.prn >= a.num_prns || code_span_bad(ratio, tau, (double)(N + a.max_abs_shift), a.Lc) ||
                         !(step == step) || !(phi == phi) || !(__builtin_fabs(step) < 1.0e15) || !(__builtin_fabs(phi) < 1.0e15);
        if (!bad) {
            const int r = N - a.n_vec; // 1 .. 7
            const size_t base = (size_t)b * a.block_stride + (size_t)k * a.chan_stride + (size_t)m * a.ant_stride + a.n_vec;
            float dr[8], di[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dr[j] = di[j] = 0.f;
                if (j < r) {
                    float xr, xi, cr, ci;
                    switch (a.format) {
                    case GAT_LAYOUT_PLANAR: SampleIO<GAT_LAYOUT_PLANAR>::load1(a.re, a.im, base + j, xr, xi); break;
                    case GAT_LAYOUT_INTERLEAVED: SampleIO<GAT_LAYOUT_INTERLEAVED>::load1(a.re, a.im, base + j, xr, xi); break;
                    case GAT_LAYOUT_INTERLEAVED_I16: SampleIO<GAT_LAYOUT_INTERLEAVED_I16>::load1(a.re, a.im, base + j, xr, xi); break;
                    default: SampleIO<GAT_LAYOUT_INTERLEAVED_I8>::load1(a.re, a.im, base + j, xr, xi); break;
                    }
                    const double th = __builtin_fma((double)(a.n_vec + j), step, phi); // src/algorithms.jl:172
                    sincos_cycles(th - __builtin_rint(th), cr, ci);
                    dr[j] = __builtin_fmaf(xr, cr, xi * ci); // conjugate wipe-off, src/algorithms.jl:175-176
                    di[j] = __builtin_fmaf(xi, cr, -(xr * ci));
                }
            }
            const int8_t *code = a.codes + (size_t)P.prn * a.code_row_stride;
            const float inv_lc = 1.0f / (float)a.Lc;
            for (int l = 0; l < a.L; ++l) {
                float sr = 0.f, si = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < r) {
                        const float c = (float)code[chip_index(ratio, tau, a.n_vec + j + a.shifts[l], a.Lc, inv_lc)];
                        sr = __builtin_fmaf(c, dr[j], sr);
                        si = __builtin_fmaf(c, di[j], si);
                    }
                const size_t o = ((size_t)bk * a.L + l) * a.M + m;
                a.out_re[o] += sr;
                a.out_im[o] += si;
            }
        }
    }
    completion_flag(a.done_counter, a.host_flag, a.flag_seq, gridDim.x);
}

// ------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------

// gen_code_replica_kernel! (src/algorithms.jl:13-32), grid-stride as _strided_ (:34-54).
// F32COORD = true emulates the reference's texture-memory variants
// (gen_code_replica_texture_mem_kernel!, src/algorithms.jl:121-140): the code phase is divided by
// the code length in Float64, rounded to a Float32 NORMALISED coordinate, wrapped and scaled back
// in Float32 with nearest-texel (floor) addressing -- the arithmetic that gives the texture path
// its code-phase error (paper/paper.tex:318-331).  It is an emulation of float32 normalised-
// coordinate addressing, not of a particular texture unit.
template <bool F32COORD>
__global__ void __launch_bounds__(kThreads)
code_replica_kernel(float *__restrict__ rep, long long count, const int8_t *__restrict__ code,
                    int Lc, double fc, double fs, double tau, long long first_shift)
{
    const double ratio = fc / fs;
    const float inv_lc = 1.0f / (float)Lc;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count;
         i += (long long)gridDim.x * kThreads) {
        int idx;
        if constexpr (F32COORD) {
            const double p = __dadd_rn(__dmul_rn(ratio, (double)(i + first_shift)), tau);
            const float u = (float)__ddiv_rn(p, (double)Lc);  // normalised coordinate, Float32
            const float w = u - __builtin_floorf(u);          // ADDRESS_MODE_WRAP
            idx = (int)__builtin_floorf(w * (float)Lc);       // NearestNeighbour == floor(u * N)
            idx = idx >= Lc ? Lc - 1 : (idx < 0 ? 0 : idx);
        } else {
            idx = chip_index(ratio, tau, (int)(i + first_shift), Lc, inv_lc);
        }
        rep[i] = (float)code[idx];
    }
}

// The texture path's addressing with the unit's FIXED-POINT steps modelled (gat.h gat_gen_code_replica_texaddr; the study of
// paper/paper.tex:318-331): u = float32(phase / Lc) as the reference hands it to the texture fetch (src/algorithms.jl:136),
// wrapped; the wrapped coordinate truncated to `coord_bits` fractional bits (0: kept), multiplied by Lc EXACTLY (24-bit x
// 14-bit fits a double), the texel address rounded to nearest at `texel_bits` fractional bits (-1: kept); chip = floor.
__global__ void __launch_bounds__(kThreads)
code_replica_texaddr_kernel(float *__restrict__ rep, long long count, const int8_t *__restrict__ code, int Lc, double fc,
                            double fs, double tau, long long first_shift, int coord_bits, int texel_bits)
{
    const double ratio = fc / fs;
    const double cscale = __builtin_ldexp(1.0, coord_bits), tscale = __builtin_ldexp(1.0, texel_bits < 0 ? 0 : texel_bits);
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count; i += (long long)gridDim.x * kThreads) {
        const double p = __dadd_rn(__dmul_rn(ratio, (double)(i + first_shift)), tau);
        const float u = (float)__ddiv_rn(p, (double)Lc);   // normalised coordinate, Float32
        double w = (double)(u - __builtin_floorf(u));     // ADDRESS_MODE_WRAP (exact in Float32)
        if (coord_bits > 0) w = __builtin_floor(w * cscale) / cscale; // fixed-point normalised coordinate (truncated)
        double x = w * (double)Lc;                         // exact
        if (texel_bits >= 0) x = __builtin_rint(x * tscale) / tscale; // fixed-point texel address (round to nearest)
        int idx = (int)__builtin_floor(x);
        idx = idx >= Lc ? idx - Lc : (idx < 0 ? 0 : idx);  // (an address rounded up to Lc wraps to chip 0)
        rep[i] = (float)code[idx];
    }
}

// A kernel that ONLY reads (gat.h gat_debug_read_stream): what the memory system gives a read-once kernel on this device.
// Workgroup w of the grid walks 16-byte groups w * 256 + t, + grid * 256, ... -- UNROLL loads in flight per lane; the sum goes
// nowhere unless it equals a value it cannot have (keeps the loads alive).
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(kThreads) read_stream_kernel(const f32x4 *__restrict__ in, size_t n16, float *sink)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += in[i];
    const float t = (acc.x + acc.y) + (acc.z + acc.w);
    if (t == 123.456f) sink[0] = t;
}

// gen_code_replica_texture_mem_strided_nsat_kernel! (src/algorithms.jl:78-98): one replica row per satellite channel
// (grid.y = channel), each with its own PRN, code rate and code phase; exact index arithmetic.
__global__ void __launch_bounds__(kThreads)
code_replica_multi_kernel(float *__restrict__ rep, long long count, long long row_stride, int K,
                          const gat_channel_params *__restrict__ params, const int8_t *__restrict__ codes,
                          int code_row_stride, int Lc, int num_prns, double fs, long long first_shift)
{
    const int k = blockIdx.y;
    if (k >= K) return;
    const gat_channel_params P = params[k];
    const double ratio = P.code_freq_hz / fs;
    const double reach = (double)count + (double)(first_shift < 0 ? -first_shift : first_shift);
    const bool bad = P.prn < 0 || P.prn >= num_prns || code_span_bad(ratio, P.code_phase_chips, reach, Lc);
    const int8_t *code = codes + (size_t)(P.prn < 0 || P.prn >= num_prns ? 0 : P.prn) * code_row_stride;
    const float inv_lc = 1.0f / (float)Lc;
    float *row = rep + (size_t)k * row_stride;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count; i += (long long)gridDim.x * kThreads)
        row[i] = bad ? __builtin_nanf("") : (float)code[chip_index(ratio, P.code_phase_chips, (int)(i + first_shift), Lc, inv_lc)];
}

// downconvert_and_accumulate_strided_kernel! (src/algorithms.jl:828-866): the MATERIALISING middle stage of the
// reference's algorithm 2 -- carrier replica [N], downconverted signal [N x M] and the per-sample products
// [N x M x L] written to global memory (the fused correlator materialises none of them; this kernel exists so that
// the reference's test of that stage, test/algorithms.jl:1438-1514, has a counterpart: prompt products == 1, column
// sums == the correlator result).  Same arithmetic as the fused kernel: double-precision carrier phase reduced to
// a quadrant, float sincos; conj(carrier) wipe-off; exact chip index.  Any output pointer may be null.
__global__ void __launch_bounds__(kThreads)
accumulate_debug_kernel(const float *__restrict__ sig_re, const float *__restrict__ sig_im, long long N, int M,
                        long long ant_stride, gat_channel_params P, const int8_t *__restrict__ code, int Lc, double fs,
                        int L, const int *__restrict__ shifts, float *__restrict__ car_re, float *__restrict__ car_im,
                        float *__restrict__ dw_re, float *__restrict__ dw_im, float *__restrict__ acc_re,
                        float *__restrict__ acc_im)
{
    const double ratio = P.code_freq_hz / fs, step = P.carrier_freq_hz / fs;
    const float inv_lc = 1.0f / (float)Lc;
    // the host validated the code-phase span (gat_downconvert_and_accumulate); a caller that bypasses it gets NaN
    // products instead of a chip index outside the table
    int max_shift = 0;
    for (int l = 0; l < L; ++l) max_shift = max(max_shift, abs(shifts[l]));
    const bool bad = code_span_bad(ratio, P.code_phase_chips, (double)N + (double)max_shift, Lc);
    for (long long n = (long long)blockIdx.x * kThreads + threadIdx.x; n < N; n += (long long)gridDim.x * kThreads) {
        float cr, ci;
        const double th = __builtin_fma((double)n, step, P.carrier_phase_cycles);
        sincos_cycles(th - __builtin_rint(th), cr, ci);
        if (car_re) car_re[n] = cr;
        if (car_im) car_im[n] = ci;
        for (int m = 0; m < M; ++m) {
            const float xr = sig_re[(size_t)m * ant_stride + n], xi = sig_im[(size_t)m * ant_stride + n];
            const float dr = __builtin_fmaf(xr, cr, xi * ci), di = __builtin_fmaf(xi, cr, -(xr * ci));
            if (dw_re) dw_re[(size_t)m * N + n] = dr;
            if (dw_im) dw_im[(size_t)m * N + n] = di;
            for (int l = 0; l < L; ++l) {
                const float chip = bad ? __builtin_nanf("")
                                       : (float)code[chip_index(ratio, P.code_phase_chips, (int)n + shifts[l], Lc, inv_lc)];
                if (acc_re) acc_re[((size_t)l * M + m) * N + n] = chip * dr;
                if (acc_im) acc_im[((size_t)l * M + m) * N + n] = chip * di;
            }
        }
    }
}

// gen_signal! (src/gen_signal.jl:64-70, :86-90): Float64 code phase, carrier phase evaluated in
// Float64 then rounded to Float32 BEFORE cos/sin (src/gen_signal.jl:88), identical antennas.
// `amplitude` scales the sum (1 = the reference); integer formats store rint(value) saturated.
// Counter-based noise for the synthetic generator (SURVEY section 8-d "build additions": AWGN + per-antenna steering; the
// reference's generator is noise-free, paper/paper.tex:116): splitmix64 of (seed, sample index) -> two uniforms ->
// Box-Muller.  A sample's noise depends on (seed, block, antenna, n) only, never on the launch geometry.
__device__ __forceinline__ void gauss_pair(unsigned long long seed, unsigned long long idx, float &g0, float &g1)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u0 = ((float)(unsigned)(z >> 40) + 0.5f) * (1.0f / 16777216.0f); // (0, 1): 24 bits
    const float u1 = ((float)(unsigned)((z >> 8) & 0xffffffu) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u0));
    float sn, cs;
    sincosf(6.2831853071795865f * u1, &sn, &cs);
    g0 = r * cs;
    g1 = r * sn;
}

__global__ void __launch_bounds__(kThreads)
gen_signal_kernel(void *__restrict__ re_v, void *__restrict__ im_v, int format, long long N,
                  int M, long long ant_stride, long long block_stride, int K,
                  const gat_channel_params *__restrict__ params, const int8_t *__restrict__ codes,
                  int code_row_stride, int Lc, int num_prns, double fs, float amplitude,
                  const float *__restrict__ steering_cycles, float noise_sigma, unsigned long long seed)
{
    const int b = blockIdx.y;
    const float inv_lc = 1.0f / (float)Lc;
    for (long long n = (long long)blockIdx.x * kThreads + threadIdx.x; n < N;
         n += (long long)gridDim.x * kThreads) {
        float sr = 0.f, si = 0.f;
        for (int k = 0; k < K; ++k) {
            const gat_channel_params P = params[(size_t)b * K + k];
            const int prn = (P.prn < 0 || P.prn >= num_prns) ? 0 : P.prn;
            const double ratio = P.code_freq_hz / fs;
            const float chip = (float)codes[(size_t)prn * code_row_stride +
                                            chip_index(ratio, P.code_phase_chips, (int)n, Lc, inv_lc)];
            // 2pi * n * f / fs + phase, left to right as the reference broadcasts it
            const double ph64 = __dadd_rn(__ddiv_rn(__dmul_rn(__dmul_rn(6.283185307179586, (double)n), P.carrier_freq_hz), fs),
                                          P.carrier_phase_cycles /* radians here, see gat.h */);
            const float ph = (float)ph64;
            sr = __builtin_fmaf(cosf(ph), chip, sr);
            si = __builtin_fmaf(sinf(ph), chip, si);
        }
        sr *= amplitude;
        si *= amplitude;
        const float sr0 = sr, si0 = si;
        for (int m = 0; m < M; ++m) {
            const size_t e = (size_t)b * block_stride + (size_t)m * ant_stride + n;
            if (steering_cycles) { // per-antenna unit-modulus steering phase
                float c_, s_;
                sincos_cycles((double)steering_cycles[m], c_, s_);
                sr = sr0 * c_ - si0 * s_;
                si = sr0 * s_ + si0 * c_;
            }
            if (noise_sigma > 0.f) { // complex white Gaussian noise, variance sigma^2 per component (scaled like the signal)
                float g0, g1;
                gauss_pair(seed, ((unsigned long long)b * (unsigned long long)M + (unsigned long long)m) * (unsigned long long)N + (unsigned long long)n, g0, g1);
                sr = (steering_cycles ? sr : sr0) + amplitude * noise_sigma * g0;
                si = (steering_cycles ? si : si0) + amplitude * noise_sigma * g1;
            }
            if (format == GAT_LAYOUT_PLANAR) {
                static_cast<float *>(re_v)[e] = sr;
                static_cast<float *>(im_v)[e] = si;
            } else if (format == GAT_LAYOUT_INTERLEAVED) {
                static_cast<float *>(re_v)[2 * e] = sr;
                static_cast<float *>(re_v)[2 * e + 1] = si;
            } else if (format == GAT_LAYOUT_INTERLEAVED_I16) {
                static_cast<short *>(re_v)[2 * e] = (short)fminf(fmaxf(rintf(sr), -32768.f), 32767.f);
                static_cast<short *>(re_v)[2 * e + 1] = (short)fminf(fmaxf(rintf(si), -32768.f), 32767.f);
            } else {
                static_cast<signed char *>(re_v)[2 * e] = (signed char)fminf(fmaxf(rintf(sr), -128.f), 127.f);
                static_cast<signed char *>(re_v)[2 * e + 1] = (signed char)fminf(fmaxf(rintf(si), -128.f), 127.f);
            }
        }
    }
}

// reduce_cplx_multi first pass (src/reduction.jl:331-403): per (chunk, column) partial sums.
// partial layout [chunks][cols*2] so that finalize_kernel produces the column sums.
__global__ void __launch_bounds__(kThreads)
reduce_stage1_kernel(const float *__restrict__ in_re, const float *__restrict__ in_im, long long n,
                     int cols, int chunks, float *__restrict__ partial)
{
    __shared__ float s[2][4];
    const int col = blockIdx.y;
    const int chunk = blockIdx.x;
    const long long per = (n + chunks - 1) / chunks;
    const long long lo = (long long)chunk * per;
    const long long hi = min(lo + per, n);
    const float *pr = in_re + (size_t)col * n;
    const float *pi = in_im + (size_t)col * n;
    float ar = 0.f, ai = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += kThreads) {
        ar += pr[i];
        ai += pi[i];
    }
    ar = wave_sum(ar);
    ai = wave_sum(ai);
    if ((threadIdx.x & 63) == 0) {
        s[0][threadIdx.x >> 6] = ar;
        s[1][threadIdx.x >> 6] = ai;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float t = (s[threadIdx.x][0] + s[threadIdx.x][1]) + (s[threadIdx.x][2] + s[threadIdx.x][3]);
        partial[(size_t)chunk * cols * 2 + (size_t)col * 2 + threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// closed tracking-loop step: discriminators + loop filters, one thread per channel (FP64; a few
// hundred flops per channel and block -- launch-bound, kept on the device to avoid a host round trip)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
tracking_update_kernel(const float *__restrict__ acc_re, const float *__restrict__ acc_im, int K, int M,
                       const gat_loop_config cfg, gat_loop_state *__restrict__ state,
                       const gat_channel_params *cur, gat_channel_params *next)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= K) return;
    gat_loop_state st = state[k];
    gat_channel_params n;
    loop_update_channel(acc_re, acc_im, k, M, cfg, st, cur[k], n); // gat_loop.h: the same text the host entry point runs
    next[k] = n;
    state[k] = st;
}

hipError_t launch_tracking_update(const float *acc_re, const float *acc_im, int K, int M, const gat_loop_config &cfg,
                                  gat_loop_state *state, const gat_channel_params *cur, gat_channel_params *next,
                                  hipStream_t s)
{
    hipLaunchKernelGGL(tracking_update_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, s, acc_re, acc_im, K, M,
                       cfg, state, cur, next);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
// the instances live in gat_dc_f*.hip: do not instantiate them here as well
extern template hipError_t launch_dc_fmt<GAT_LAYOUT_PLANAR>(const DcArgs &, const DcLaunch &, hipStream_t);
extern template hipError_t launch_dc_fmt<GAT_LAYOUT_INTERLEAVED>(const DcArgs &, const DcLaunch &, hipStream_t);
extern template hipError_t launch_dc_fmt<GAT_LAYOUT_INTERLEAVED_I16>(const DcArgs &, const DcLaunch &, hipStream_t);
extern template hipError_t launch_dc_fmt<GAT_LAYOUT_INTERLEAVED_I8>(const DcArgs &, const DcLaunch &, hipStream_t);

hipError_t launch_dc(const DcArgs &a, const DcLaunch &cfg, hipStream_t s)
{
    if (!dc_instance(cfg.ant_tile, cfg.taps, cfg.vec, cfg.aw, cfg.kt)) return hipErrorInvalidValue;
    switch (cfg.format) {
    case GAT_LAYOUT_PLANAR: return launch_dc_fmt<GAT_LAYOUT_PLANAR>(a, cfg, s);
    case GAT_LAYOUT_INTERLEAVED: return launch_dc_fmt<GAT_LAYOUT_INTERLEAVED>(a, cfg, s);
    case GAT_LAYOUT_INTERLEAVED_I16: return launch_dc_fmt<GAT_LAYOUT_INTERLEAVED_I16>(a, cfg, s);
    case GAT_LAYOUT_INTERLEAVED_I8: return launch_dc_fmt<GAT_LAYOUT_INTERLEAVED_I8>(a, cfg, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize(const float *partial, float *out_re, float *out_im, int splits, int elems,
                           long long groups, hipStream_t s, unsigned *done_counter, unsigned *host_flag, unsigned flag_seq)
{
    const long long waves = groups * elems;
    if (splits <= kFinalizeFewSplits) { // one thread per output element
        hipLaunchKernelGGL(finalize_few_kernel, dim3((unsigned)((waves + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           partial, out_re, out_im, splits, elems, waves, done_counter, host_flag, flag_seq);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((waves + kFinElems - 1) / kFinElems);
    hipLaunchKernelGGL(finalize_kernel, dim3(grid), dim3(kThreads), 0, s, partial, out_re, out_im,
                       splits, elems, waves, done_counter, host_flag, flag_seq);
    return hipGetLastError();
}

hipError_t launch_dc_tail(const DcTailArgs &a, hipStream_t s)
{
    const long long total = (long long)a.B * a.K * a.M;
    hipLaunchKernelGGL(dc_tail_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_gen_code_replica(float *rep, long long count, const int8_t *code_row, int Lc,
                                   double fc, double fs, double tau, long long first_shift,
                                   bool f32_coordinates, hipStream_t s)
{
    long long blocks = (count + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    if (f32_coordinates)
        hipLaunchKernelGGL(code_replica_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, s, rep, count,
                           code_row, Lc, fc, fs, tau, first_shift);
    else
        hipLaunchKernelGGL(code_replica_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, s, rep, count,
                           code_row, Lc, fc, fs, tau, first_shift);
    return hipGetLastError();
}

hipError_t launch_gen_code_replica_texaddr(float *rep, long long count, const int8_t *code_row, int Lc, double fc, double fs,
                                           double tau, long long first_shift, int coord_frac_bits, int texel_frac_bits,
                                           hipStream_t s)
{
    long long blocks = (count + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(code_replica_texaddr_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, rep, count, code_row, Lc, fc,
                       fs, tau, first_shift, coord_frac_bits, texel_frac_bits);
    return hipGetLastError();
}

hipError_t launch_read_stream(const void *dev, size_t bytes, int variant, int num_cus, float *sink, hipStream_t s)
{
    const unsigned grid = (unsigned)num_cus * (8u << (variant & 3));
    const f32x4 *in = static_cast<const f32x4 *>(dev);
    const size_t n16 = bytes / 16;
    switch ((variant >> 2) & 3) {
    case 0: hipLaunchKernelGGL((read_stream_kernel<8, true>), dim3(grid), dim3(kThreads), 0, s, in, n16, sink); break;
    case 1: hipLaunchKernelGGL((read_stream_kernel<8, false>), dim3(grid), dim3(kThreads), 0, s, in, n16, sink); break;
    case 2: hipLaunchKernelGGL((read_stream_kernel<4, true>), dim3(grid), dim3(kThreads), 0, s, in, n16, sink); break;
    default: hipLaunchKernelGGL((read_stream_kernel<4, false>), dim3(grid), dim3(kThreads), 0, s, in, n16, sink); break;
    }
    return hipGetLastError();
}

hipError_t launch_gen_code_replica_multi(float *rep, long long count, long long row_stride, int K,
                                         const gat_channel_params *params, const int8_t *codes, int code_row_stride,
                                         int Lc, int num_prns, double fs, long long first_shift, hipStream_t s)
{
    long long bx = (count + kThreads - 1) / kThreads;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(code_replica_multi_kernel, dim3((unsigned)bx, (unsigned)K), dim3(kThreads), 0, s, rep, count,
                       row_stride, K, params, codes, code_row_stride, Lc, num_prns, fs, first_shift);
    return hipGetLastError();
}

hipError_t launch_accumulate_debug(const float *sig_re, const float *sig_im, long long N, int M, long long ant_stride,
                                   const gat_channel_params &P, const int8_t *code_row, int Lc, double fs, int L,
                                   const int *shifts_dev, float *car_re, float *car_im, float *dw_re, float *dw_im,
                                   float *acc_re, float *acc_im, hipStream_t s)
{
    long long bx = (N + kThreads - 1) / kThreads;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(accumulate_debug_kernel, dim3((unsigned)bx), dim3(kThreads), 0, s, sig_re, sig_im, N, M, ant_stride, P,
                       code_row, Lc, fs, L, shifts_dev, car_re, car_im, dw_re, dw_im, acc_re, acc_im);
    return hipGetLastError();
}

hipError_t launch_gen_signal(void *re, void *im, int format, long long N, int M,
                             long long ant_stride, long long block_stride, int B, int K,
                             const gat_channel_params *params, const int8_t *codes, int code_row_stride,
                             int Lc, int num_prns, double fs, float amplitude, const float *steering_cycles, float noise_sigma,
                             unsigned long long seed, hipStream_t s)
{
    long long bx = (N + kThreads - 1) / kThreads;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(gen_signal_kernel, dim3((unsigned)bx, (unsigned)B), dim3(kThreads), 0, s, re,
                       im, format, N, M, ant_stride, block_stride, K, params, codes, code_row_stride, Lc,
                       num_prns, fs, amplitude, steering_cycles, noise_sigma, seed);
    return hipGetLastError();
}

hipError_t launch_reduce_stage1(const float *in_re, const float *in_im, long long n, int cols,
                                int chunks, float *partial, hipStream_t s)
{
    hipLaunchKernelGGL(reduce_stage1_kernel, dim3((unsigned)chunks, (unsigned)cols), dim3(kThreads), 0,
                       s, in_re, in_im, n, cols, chunks, partial);
    return hipGetLastError();
}

} // namespace gat
