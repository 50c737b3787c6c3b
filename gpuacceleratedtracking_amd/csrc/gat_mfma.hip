// gat_mfma.hip -- matrix-core formulation of downconvert + correlate for antenna-rich receivers
// (M a multiple of 16; BASELINE configs 4 and 5: 16 and 64 antennas).
//
//   R[m,(k,l)] = sum_n x[n,m] * W[n,(k,l)],   W = conj(carrier_k[n]) * c_k[n + shift_l]
//
// recast as the real GEMM  C[32 x 32] += A^T[32 x n] * B[n x 32]  per 16-antenna tile:
//   rows    i = 2*m + {0: x_re, 1: x_im}                       (16 antennas)
//   columns j = 2*(kc*L + l) + {0: w_re = chip*cos, 1: w_im = -chip*sin}   (CT = 16/L channels)
//   R_re = C[x_re,w_re] - C[x_im,w_im],  R_im = C[x_re,w_im] + C[x_im,w_re]
// on v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bit-for-bit an fmaf chain, so the numerics
// equal the vector kernel's).  The matrix instruction does ALL multiply-accumulates; the vector
// ALU only builds B (one carrier rotation + one chip fetch per lane and MFMA) -- in the vector
// kernel the same work costs 4 + 2L FMAs per antenna, channel and sample plus per-4-antenna-tile
// carrier / replica overhead, and that kernel is VALU-bound for these shapes.
// Measured limit (scripts/probes/mfma_probe.hip, MI355X): a dependent chain of this instruction issues
// every 64.0 cycles bare, 94 with one carrier rotation + product (5 VALU) per MFMA, 132 with two
// LDS fetches more -- the f32 MFMA runs at the FP32 vector rate and vector work does NOT hide
// behind it as it does behind the bf16 matrix pipe, so this kernel gains what it saves in
// instruction count (1.3-1.5x over the vector kernel at 16-64 antennas x 32-64 channels), not more.
//
// A workgroup works on one tile of T = 256 samples x 16 antennas staged in LDS (coalesced 16-byte
// loads, XOR-swizzled columns so that the per-MFMA A fetch -- 32 lanes, 32 planes, one sample --
// hits 32 banks), double-buffered, with producer and consumer waves (see mfma_kernel).  The 4
// consumer waves take NCT in {1,2,4} different channel tiles (CT channels each) against the
// SAME staged samples.  Within a wave the two 32-lane halves carry two sample streams (MFMA
// K = 2): the sum over samples does not care which sample sits in which K slot as long as A and
// B agree.
// Code replica segments [CT][T + span] are generated per step into LDS exactly as in dc_kernel
// (FP64 code phase, unfused).
#include "gat_internal.h"
#include "gat_phase.h"

namespace gat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTile = kMfTile;       // samples per step (gat_internal.h: the host sizes LDS with the same numbers)
constexpr int kXStride = kMfXStride; // floats per plane row in LDS: odd, so that for one sample the 32 planes
                                     // (what one MFMA A fetch reads) sit in 32 different banks

struct ChanInfo { // per channel slot of the workgroup, in LDS
    double ratio, tau, step, phi;
    int prn, valid, bad, pad;
};

} // namespace

// NCT: channel tiles per workgroup (1, 2 or 4).  Workgroup = 8 waves with fixed roles:
//   waves 0-3  CONSUMERS: MFMA only -- per MFMA two LDS operand fetches, one multiply, one carrier
//              rotation.  They take NCT channel tiles; with NCT < 4 they split the tile's samples WPT = 4 / NCT
//              ways and their accumulators are summed through LDS at the end.
//   waves 4-7  PRODUCERS: while the consumers work on step s they stage step s+1 into the other
//              LDS buffer: coalesced 16-byte loads of the 32 sample planes (XOR-swizzled rows) and
//              the code replica segments of all NCT*CT channels (FP64 code phase, unfused).
// A consumer wave and a producer wave share each SIMD: the matrix pipe and the vector ALU run
// concurrently, so replica generation / staging costs the MFMA stream (almost) nothing.  One
// s_barrier per step.
template <int NCT>
__global__ void __launch_bounds__(2 * kThreads) mfma_kernel(const MfArgs a)
{
    constexpr int WPT = 4 / NCT;
    constexpr int SW = kTile / WPT; // samples of the tile handled by one consumer wave
    constexpr int NM = SW / 2;      // MFMAs per consumer wave and step (two sample streams)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int CT = a.CT, L = a.L;
    const int nslots = NCT * CT;
    ChanInfo *s_chan = reinterpret_cast<ChanInfo *>(smem);               // [nslots] (<= 20 * 48 B), 1 KB
    float *s_x = reinterpret_cast<float *>(smem + 1024);                 // [2][32][kXStride]
    float *s_rep = s_x + 2 * 32 * kXStride;                              // [2][nslots][rep_stride]
    // [nslots][code_row_stride] int8 chip tables (if staged), base rounded up to 16 bytes
    int8_t *s_code = reinterpret_cast<int8_t *>(
        smem + ((1024 + (2 * 32 * kXStride + 2 * nslots * a.rep_stride) * sizeof(float) + 15) & ~size_t(15)));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;

    // workgroup -> (tile, channel group); same XCD trick as dc_kernel (speed only)
    const unsigned xcd = blockIdx.x & 7u, jq = blockIdx.x >> 3;
    const int cg = (int)(jq % (unsigned)a.chan_groups);
    unsigned tile = (jq / (unsigned)a.chan_groups) * 8u + xcd;
    if (tile >= (unsigned)a.num_tiles) return;
    const int split = tile % a.splits;
    tile /= a.splits;
    const int at = tile % a.ant_tiles;
    const int b = tile / a.ant_tiles;
    const int N = (int)a.N;
    const int Lc = a.Lc;
    const float inv_lc = 1.0f / (float)Lc;

    // per-channel constants of this workgroup's channel slots
    if (tid < nslots) {
        const int k = (cg * NCT + tid / CT) * CT + tid % CT;
        ChanInfo ci{};
        ci.valid = k < a.K;
        if (ci.valid) {
            const gat_channel_params P = a.params[(size_t)b * a.K + k];
            ci.ratio = P.code_freq_hz / a.fs;
            ci.step = P.carrier_freq_hz / a.fs;
            ci.tau = P.code_phase_chips;
            ci.phi = P.carrier_phase_cycles;
            const double span = __builtin_fabs(ci.tau) + __builtin_fabs(ci.ratio) * (double)(N + a.max_abs_shift) + 1.0;
            ci.bad = P.prn < 0 || P.prn >= a.num_prns || !(span < 1073741824.0) || !(span < 2097152.0 * (double)Lc) ||
                     !(ci.ratio >= 0.0) || !(ci.ratio * 32.0 < (double)Lc) || !(ci.step == ci.step) || !(ci.phi == ci.phi);
            ci.prn = (P.prn < 0 || P.prn >= a.num_prns) ? 0 : P.prn;
            if (ci.bad) { ci.ratio = 0.0; ci.tau = 0.0; ci.step = 0.0; ci.phi = 0.0; }
        }
        s_chan[tid] = ci;
    }
    __syncthreads();
    if (a.codes_in_lds) { // chip tables of this workgroup's channels: LDS gathers instead of L2 gathers
        const int vec_per_row = a.code_row_stride / 16;
        for (int e = tid; e < nslots * vec_per_row; e += 2 * kThreads) {
            const int slot = e / vec_per_row, v = e - slot * vec_per_row;
            const ChanInfo c = s_chan[slot];
            if (c.valid)
                reinterpret_cast<int4 *>(s_code)[slot * vec_per_row + v] =
                    reinterpret_cast<const int4 *>(a.codes + (size_t)c.prn * a.code_row_stride)[v];
        }
        __syncthreads();
    }

    const size_t base = (size_t)b * a.block_stride + (size_t)(at * 16) * a.ant_stride;
    const int s_begin = split * a.steps_per_split;
    const int s_end = min(s_begin + a.steps_per_split, a.total_steps);
    const int rep_cnt = kTile + a.rep_span;

    // ---- producer: stage step st into buffer buf -----------------------------------------------
    // The sample loads run TWO steps ahead of the consumers (registers xv hold step st's samples,
    // loaded while step st-1 was being produced): one step of look-ahead leaves a CU with 32 KB in
    // flight per ~10k cycles and the pipeline runs at the HBM latency instead of the MFMA rate.
    f32x4_ xv[8];
    const int pw = wave - 4; // producer wave 0..3 (negative for consumers: unused)
    auto load_x = [&](int st) {
        const int nb = st * kTile;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int plane = pw * 8 + p; // row = 2*m_local + comp
            const float *src = static_cast<const float *>((plane & 1) ? a.im : a.re) + base + (size_t)(plane >> 1) * a.ant_stride + nb + 4 * lane;
            if (nb + 4 * lane + 4 <= N) {
                xv[p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ *>(src));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[p][j] = (nb + 4 * lane + j < N) ? src[j] : 0.f;
            }
        }
    };
    auto produce = [&](int st, int buf) { // xv holds the samples of step st
        const int nb = st * kTile;
        float *xb = s_x + buf * 32 * kXStride;
        float *rb = s_rep + buf * nslots * a.rep_stride;
        // replica segments: producer wave pw takes slots pw, pw+4, ... (at most 5).  For one slot the
        // channel constants are wave-uniform and a lane generates E consecutive entries, so only the
        // first needs the full floored modulo -- the others follow from the exact FP64 floor
        // differences (chips advance monotonically, at most one wrap).  All slots of the wave are
        // worked on together (up to 20 independent FP64 chains in flight): this code is latency-
        // bound, not issue-bound.
        const int x0 = nb + a.shifts[0];
        const int E = (rep_cnt + 63) >> 6;
        const int i0 = lane * E;
        constexpr int MS = 5; // max slots per producer wave (20 slots / 4 waves)
        double ratio_s[MS], tau_s[MS];
        int ip0_s[MS], idx0_s[MS], valid_s[MS];
        const int8_t *tab_s[MS];
#pragma unroll
        for (int z = 0; z < MS; ++z) {
            const int slot = pw + 4 * z;
            const bool have = slot < nslots;
            const ChanInfo c = s_chan[have ? slot : 0];
            valid_s[z] = have ? (c.valid ? 1 : -1) : 0; // 1: generate, -1: zero-fill, 0: no such slot
            ratio_s[z] = c.ratio;
            tau_s[z] = c.tau;
            tab_s[z] = a.codes_in_lds ? (s_code + (have ? slot : 0) * a.code_row_stride)
                                      : (a.codes + (size_t)c.prn * a.code_row_stride);
            const double p0 = code_phase(c.ratio, c.tau, x0 + i0);
            const int ip0 = (int)__builtin_floor(p0);
            const int idx0 = floormod_fast(ip0, Lc, inv_lc);
            ip0_s[z] = ip0;
            idx0_s[z] = idx0;
        }
        for (int j0 = 0; j0 < E; j0 += 4) {
            float ch[MS][4];
#pragma unroll
            for (int z = 0; z < MS; ++z) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double pj = code_phase(ratio_s[z], tau_s[z], x0 + i0 + j0 + u);
                    int t = idx0_s[z] + ((int)__builtin_floor(pj) - ip0_s[z]);
                    t -= (t >= Lc) ? Lc : 0; // at most one wrap: E * ratio < Lc (channel marked bad otherwise)
                    ch[z][u] = valid_s[z] > 0 ? (float)tab_s[z][t] : 0.f; // tab_s: LDS or global (generic pointer)
                }
            }
#pragma unroll
            for (int z = 0; z < MS; ++z) {
                float *row = rb + (pw + 4 * z) * a.rep_stride;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (valid_s[z] != 0 && j0 + u < E && i0 + j0 + u < rep_cnt) row[i0 + j0 + u] = ch[z][u];
            }
        }
        // x registers -> LDS rows (odd row stride: four 4-byte stores per plane instead of one 16-byte
        // store, the price of conflict-free A fetches with plain immediate offsets in the consumers)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float *dst = xb + (pw * 8 + p) * kXStride + 4 * lane;
            dst[0] = xv[p][0];
            dst[1] = xv[p][1];
            dst[2] = xv[p][2];
            dst[3] = xv[p][3];
        }
        if (st + 1 < s_end) load_x(st + 1); // in flight across the barrier and the next step's replica fill
    };

    // ---- consumer state: this lane's column of B -------------------------------------------------
    const int cw = wave & 3;
    const int ctl = cw / WPT;   // channel tile of this consumer wave within the workgroup
    const int sub = cw % WPT;   // sample sub-range of the tile
    const int r = lane & 31, h = lane >> 5;
    const int kl = r >> 1, comp = r & 1;
    const int kc = kl / L, l = kl - kc * L;
    const int slot_c = ctl * CT + (kc < CT ? kc : 0);
    const ChanInfo my = s_chan[slot_c];
    const bool live_col = kc < CT && my.valid;
    const int rep_off = slot_c * a.rep_stride + (a.shifts[l < L ? l : 0] - a.shifts[0]);
    float wr, wi;
    sincos_cycles(my.step - __builtin_rint(my.step), wr, wi);
    const float gain = live_col ? 1.f : 0.f; // dead columns multiply by zero
    const int col0 = sub * SW + h * NM;      // first sample (tile-relative) of this lane's stream

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // Consumer step.  Groups of 8 MFMAs; while group g's MFMAs issue back to back (64 cycles each,
    // one dependent accumulator chain), the VALU prepares group g+1 in OTHER registers: LDS operand
    // fetches, the 8 carrier rotations, the 8 chip * carrier products.  Nothing the vector ALU
    // writes is an operand of an MFMA in flight (a VALU write to a live MFMA source stalls the
    // pipe: the first version of this loop ran at 1/3 of the MFMA rate for that reason).
    auto consume = [&](int st, int buf) {
        const int nb = st * kTile;
        const float *xrow = s_x + buf * 32 * kXStride + r * kXStride;
        const float *rrow = s_rep + buf * nslots * a.rep_stride + rep_off;
        // (p, q): p multiplies the chip.  comp 0: (cos, sin); comp 1: (-sin, cos) -- the same phasor
        // turned by 90 degrees, so both columns use the identical rotation recurrence.
        float p = 0.f, q = 0.f;
        auto anchor = [&](int cb) { // FP64 carrier anchor
            const double th = __builtin_fma((double)(nb + cb), my.step, my.phi);
            float cr, ci;
            sincos_cycles(th - __builtin_rint(th), cr, ci);
            p = (comp ? -ci : cr) * gain;
            q = (comp ? cr : ci) * gain;
        };
        auto fetch = [&](float (&av)[8], float (&cv)[8], int cb) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = xrow[cb + u];
                cv[u] = rrow[cb + u];
            }
        };
        float avA[8], cvA[8], bA[8], avB[8], cvB[8], bB[8];
        anchor(col0);
        fetch(avA, cvA, col0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bA[u] = cvA[u] * p;
            const float tp = __builtin_fmaf(p, wr, -(q * wi));
            q = __builtin_fmaf(p, wi, q * wr);
            p = tp;
        }
        // one pipeline stage: run the 8 MFMAs of (avC, bC) while preparing (avN, bN) for column cbn.
        // Per MFMA slot: the two LDS fetches of one element of the next group and one carrier
        // rotation; the scheduler is told to keep exactly that interleaving (hipcc otherwise sinks
        // the ds_reads next to their uses, an s_waitcnt lgkmcnt(0) in front of every MFMA).
        auto stage = [&](const float (&avC)[8], const float (&bC)[8], float (&avN)[8], float (&cvN)[8], float (&bN)[8],
                         int cbn, bool have_next) {
            float pn[8];
            if (have_next && ((cbn - col0) & 31) == 0) anchor(cbn); // re-anchor every 32 samples
            const int cbr = have_next ? cbn : col0;                  // keep the reads in range on the last stage
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(avC[u], bC[u], acc, 0, 0, 0);
                avN[u] = xrow[cbr + u];
                cvN[u] = rrow[cbr + u];
                pn[u] = p;
                const float tp = __builtin_fmaf(p, wr, -(q * wi));
                q = __builtin_fmaf(p, wi, q * wr);
                p = tp;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); // 2 DS reads
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); // 5 VALU (one rotation)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) bN[u] = cvN[u] * pn[u];
        };
        for (int t0 = 0; t0 < NM; t0 += 16) { // two stages per iteration: registers A <-> B
            stage(avA, bA, avB, cvB, bB, col0 + t0 + 8, true);
            stage(avB, bB, avA, cvA, bA, col0 + t0 + 16, t0 + 16 < NM);
        }
    };

    // ---- pipeline ----------------------------------------------------------------------------------
#ifdef GAT_MFMA_STAMPS
    unsigned long long t_work = 0, t_wait = 0, t0_ = 0, t1_ = 0;
#define GAT_STAMP(v) v = __builtin_amdgcn_s_memtime()
#else
#define GAT_STAMP(v)
#endif
    if (producer && s_begin < s_end) {
        load_x(s_begin);
        produce(s_begin, 0);
    }
    __syncthreads();
    for (int st = s_begin; st < s_end; ++st) {
        const int buf = (st - s_begin) & 1;
        GAT_STAMP(t0_);
        if (producer) {
            if (st + 1 < s_end) produce(st + 1, buf ^ 1);
        } else {
            consume(st, buf);
        }
        GAT_STAMP(t1_);
        __syncthreads();
#ifdef GAT_MFMA_STAMPS
        t_work += t1_ - t0_;
        t_wait += __builtin_amdgcn_s_memtime() - t1_;
#endif
    }
#ifdef GAT_MFMA_STAMPS
    if (lane == 0 && a.dbg) { // [workgroup][wave][2]
        a.dbg[((size_t)blockIdx.x * 8 + wave) * 2 + 0] = t_work;
        a.dbg[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = t_wait;
    }
#endif

    // ---- epilogue ----------------------------------------------------------------------------
    if constexpr (WPT > 1) { // sum the accumulators of the consumer waves that shared a channel tile
        float *s_red = s_x; // [4 waves][16][64] floats = 16 KB, fits in the x staging area
        if (!producer) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s_red[(cw * 16 + i) * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (!producer && sub == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float s = 0.f;
                for (int qq = 0; qq < WPT; ++qq) s += s_red[((ctl * WPT + qq) * 16 + i) * 64 + lane];
                acc[i] = s;
            }
        }
    }
    if (producer || sub != 0) return;
    // C[row][col]: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).  Rows 2m / 2m+1 are
    // registers i / i+1 of one lane; columns w_re / w_im are lanes c / c^1.
    const int k = (cg * NCT + ctl) * CT + kc;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const float v0 = acc[i], v1 = acc[i + 1];  // x_re * w_comp, x_im * w_comp
        const float o1 = __shfl_xor(v1, 1, 64);    // partner column's x_im product
        float val = comp ? (v0 + o1) : (v0 - o1);  // comp 0: R_re = xr*wr - xi*wi ; comp 1: R_im = xr*wi + xi*wr
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int m = at * 16 + (row >> 1);
        if (live_col) {
            if (my.bad) val = __builtin_nanf("");
            const int lo = a.tap_index[l];
            const size_t bk = (size_t)b * a.K + k;
            const size_t o = (bk * a.L + lo) * a.M + m;
            float *dst = comp ? a.out_im : a.out_re;
            if (a.flags & GAT_FLAG_ATOMIC) {
                atomicAdd(dst + o, val);
            } else if (a.splits == 1) {
                dst[o] = val;
            } else {
                const size_t elems = (size_t)a.L * a.M * 2;
                a.partial[(bk * a.splits + split) * elems + ((size_t)lo * a.M + m) * 2 + comp] = val;
            }
        }
    }
}

hipError_t launch_mfma(const MfArgs &a, int nct, unsigned grid, unsigned lds_bytes, hipStream_t s)
{
    const dim3 g(grid), blk(2 * kThreads);
    switch (nct) {
    case 1: hipLaunchKernelGGL(mfma_kernel<1>, g, blk, lds_bytes, s, a); break;
    case 2: hipLaunchKernelGGL(mfma_kernel<2>, g, blk, lds_bytes, s, a); break;
    case 4: hipLaunchKernelGGL(mfma_kernel<4>, g, blk, lds_bytes, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

size_t mfma_lds_bytes(int nct, int ct, int rep_stride, int code_row_stride, int codes_in_lds)
{
    static_assert(sizeof(ChanInfo) * 20 <= 1024, "channel table must fit its 1 KB slot");
    return mf_lds_bytes(nct, ct, rep_stride, code_row_stride, codes_in_lds);
}

} // namespace gat
