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
// equal the vector kernel's).  The matrix pipe does ALL multiply-accumulates; the vector ALU only
// builds B (one carrier rotation + one chip fetch per lane and MFMA) -- in the vector kernel the
// same work costs 4 + 2L FMAs per antenna, channel and sample plus per-4-antenna-tile carrier /
// replica overhead, and that kernel is VALU-bound for these shapes.
//
// Workgroup = 4 waves on one tile of T = 256 samples x 16 antennas staged in LDS (coalesced
// 16-byte loads, XOR-swizzled columns so that the per-MFMA A fetch -- 32 lanes, 32 planes, one
// sample -- hits 32 banks).  The 4 waves take NCT in {1,2,4} different channel tiles (CT channels
// each) against the SAME staged samples; with fewer channel tiles they split the tile's samples
// instead and their accumulators are summed through LDS at the end.  Within a wave the two
// 32-lane halves carry two sample streams (MFMA K = 2): the sum over samples does not care
// which sample sits in which K slot as long as A and B agree.
// Code replica segments [CT][T + span] are generated per step into LDS exactly as in dc_kernel
// (FP64 code phase, unfused).
#include "gat_internal.h"

namespace gat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTile = 256;          // samples per step
constexpr int kXStride = kTile + 4; // floats per plane row in LDS (16-byte aligned rows)

__device__ __forceinline__ void sincos_cycles_m(double theta, float &c, float &s)
{
    const double q = __builtin_rint(theta * 4.0);
    const double r = __builtin_fma(q, -0.25, theta);
    const float a = (float)r * 6.283185307179586f;
    const float a2 = a * a;
    float sp = __builtin_fmaf(a2, 2.7557319e-6f, -1.9841270e-4f);
    sp = __builtin_fmaf(a2, sp, 8.3333333e-3f);
    sp = __builtin_fmaf(a2, sp, -1.6666667e-1f);
    sp = __builtin_fmaf(a2 * a, sp, a);
    float cp = __builtin_fmaf(a2, 2.4801587e-5f, -1.3888889e-3f);
    cp = __builtin_fmaf(a2, cp, 4.1666667e-2f);
    cp = __builtin_fmaf(a2, cp, -0.5f);
    cp = __builtin_fmaf(a2, cp, 1.0f);
    const int qi = (int)(long long)q & 3;
    const float cs = (qi & 1) ? sp : cp;
    const float sn = (qi & 1) ? cp : sp;
    c = (qi == 1 || qi == 2) ? -cs : cs;
    s = (qi >= 2) ? -sn : sn;
}

__device__ __forceinline__ int chip_index_m(double ratio, double tau, int x, int Lc, float inv_lc)
{
    const double p = __dadd_rn(__dmul_rn(ratio, (double)x), tau); // src/algorithms.jl:179, unfused
    const int ip = (int)__builtin_floor(p);
    const float q = __builtin_floorf((float)ip * inv_lc);
    int r = ip - (int)q * Lc;
    r += (r < 0) ? Lc : 0;
    r -= (r >= Lc) ? Lc : 0;
    return r;
}

// column of sample s of plane r inside its LDS row: XOR the low two bits with (r >> 3) & 3, so
// that for a fixed sample the 32 planes fall into 32 different banks (row stride == 4 mod 32).
__device__ __forceinline__ int swz(int r, int s) { return s ^ ((r >> 3) & 3); }

struct ChanInfo { // per channel slot of the workgroup, in LDS
    double ratio, tau, step, phi;
    int prn, valid, bad, pad;
};

} // namespace

// NCT: channel tiles per workgroup (1, 2 or 4); the 4 waves are split WPT = 4 / NCT ways over the samples.
template <int NCT>
__global__ void __launch_bounds__(kThreads) mfma_kernel(const MfArgs a)
{
    constexpr int WPT = 4 / NCT;
    constexpr int SW = kTile / WPT; // samples of the tile handled by one wave
    constexpr int NM = SW / 2;      // MFMAs per wave and step (two sample streams)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ChanInfo *s_chan = reinterpret_cast<ChanInfo *>(smem);               // [NCT*CT] (<= 20 * 48 B), 1 KB
    float *s_x = reinterpret_cast<float *>(smem + 1024);                 // [32][kXStride]
    float *s_rep = s_x + 32 * kXStride;                                  // [NCT*CT][rep_stride]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CT = a.CT, L = a.L;

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

    // per-channel constants of this workgroup's NCT*CT channel slots
    if (tid < NCT * CT) {
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
                     !(ci.ratio >= 0.0) || !(ci.step == ci.step) || !(ci.phi == ci.phi);
            ci.prn = (P.prn < 0 || P.prn >= a.num_prns) ? 0 : P.prn;
            if (ci.bad) { ci.ratio = 0.0; ci.tau = 0.0; ci.step = 0.0; ci.phi = 0.0; }
        }
        s_chan[tid] = ci;
    }
    __syncthreads();

    // ---- this lane's column of B ------------------------------------------------------------
    const int ctl = wave / WPT;   // channel tile of this wave within the workgroup
    const int sub = wave % WPT;   // sample sub-range of the tile
    const int r = lane & 31, h = lane >> 5;
    const int kl = r >> 1, comp = r & 1;
    const int kc = kl / L, l = kl - kc * L;
    const bool live_col = kc < CT && s_chan[ctl * CT + (kc < CT ? kc : 0)].valid;
    const ChanInfo my = s_chan[ctl * CT + (kc < CT ? kc : 0)];
    const float *rep_row = s_rep + (ctl * CT + (kc < CT ? kc : 0)) * a.rep_stride + (a.shifts[l < L ? l : 0] - a.shifts[0]);
    float wr, wi;
    sincos_cycles_m(my.step - __builtin_rint(my.step), wr, wi);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const size_t base = (size_t)b * a.block_stride + (size_t)(at * 16) * a.ant_stride;
    const int s_begin = split * a.steps_per_split;
    const int s_end = min(s_begin + a.steps_per_split, a.total_steps);
    const int rep_cnt = kTile + a.rep_span;
    const int col0 = sub * SW + h * NM; // first sample (tile-relative) of this lane's stream

    for (int st = s_begin; st < s_end; ++st) {
        const int nb = st * kTile;
        // (1) stage x: wave w loads planes 8w .. 8w+7, 16 bytes (4 samples) per lane and plane
        f32x4_ xv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int plane = wave * 8 + p; // row = 2*m_local + comp
            const float *src = ((plane & 1) ? a.im : a.re) + base + (size_t)(plane >> 1) * a.ant_stride + nb + 4 * lane;
            if (nb + 4 * lane + 4 <= N) {
                xv[p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ *>(src));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[p][j] = (nb + 4 * lane + j < N) ? src[j] : 0.f;
            }
        }
        // (2) replica segments of this wave's channel tile (shared by the WPT waves of the tile)
        for (int e = sub * 64 + lane; e < CT * rep_cnt; e += WPT * 64) {
            const int kq = e / rep_cnt, i = e - kq * rep_cnt;
            const ChanInfo c = s_chan[ctl * CT + kq];
            float chip = 0.f;
            if (c.valid)
                chip = (float)a.codes[(size_t)c.prn * a.code_row_stride +
                                      chip_index_m(c.ratio, c.tau, nb + a.shifts[0] + i, Lc, inv_lc)];
            s_rep[(ctl * CT + kq) * a.rep_stride + i] = chip;
        }
        // (3) x registers -> LDS rows (column XOR swizzle = a permutation inside the 16-byte group)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int plane = wave * 8 + p;
            // sample j goes to position j ^ g, g = (plane >> 3) & 3 == wave (wave-uniform): two
            // conditional pair swaps instead of a dynamically indexed vector
            float v0 = xv[p][0], v1 = xv[p][1], v2 = xv[p][2], v3 = xv[p][3], t_;
            if (wave & 1) { t_ = v0; v0 = v1; v1 = t_; t_ = v2; v2 = v3; v3 = t_; }
            if (wave & 2) { t_ = v0; v0 = v2; v2 = t_; t_ = v1; v1 = v3; v3 = t_; }
            f32x4_ w;
            w[0] = v0; w[1] = v1; w[2] = v2; w[3] = v3;
            *reinterpret_cast<f32x4_ *>(s_x + plane * kXStride + 4 * lane) = w;
        }
        __syncthreads();

        // (4) MFMA loop over this lane's stream of NM samples
        float cr = 0.f, ci = 0.f;
#pragma unroll 4
        for (int t = 0; t < NM; ++t) {
            const int col = col0 + t;
            if ((t & 31) == 0) { // FP64 carrier anchor, then rotations
                const double th = __builtin_fma((double)(nb + col), my.step, my.phi);
                sincos_cycles_m(th - __builtin_rint(th), cr, ci);
            }
            const float av = s_x[r * kXStride + swz(r, col)];
            const float chip = rep_row[col];
            const float sel = comp ? -ci : cr; // w = chip * (cos - j sin)
            const float bv = live_col ? chip * sel : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            const float tr = __builtin_fmaf(cr, wr, -(ci * wi));
            ci = __builtin_fmaf(cr, wi, ci * wr);
            cr = tr;
        }
        __syncthreads(); // everyone done with s_x / s_rep before the next step overwrites them
    }

    // ---- epilogue ----------------------------------------------------------------------------
    if constexpr (WPT > 1) { // sum the accumulators of the waves that shared a channel tile
        float *s_red = s_x; // [4 waves][16][64] floats = 16 KB, fits in the x staging area
#pragma unroll
        for (int i = 0; i < 16; ++i) s_red[(wave * 16 + i) * 64 + lane] = acc[i];
        __syncthreads();
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float s = 0.f;
                for (int q = 0; q < WPT; ++q) s += s_red[((ctl * WPT + q) * 16 + i) * 64 + lane];
                acc[i] = s;
            }
        }
    }
    if (sub != 0) return;
    // C[row][col]: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).  Rows 2m / 2m+1 are
    // registers i / i+1 of one lane; columns w_re / w_im are lanes c / c^1.
    const int k = (cg * NCT + ctl) * CT + kc;
    const bool write = live_col;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const float v0 = acc[i], v1 = acc[i + 1];  // x_re * w_comp, x_im * w_comp
        const float o1 = __shfl_xor(v1, 1, 64);    // partner column's x_im product
        float val = comp ? (v0 + o1) : (v0 - o1);  // comp 0: R_re = xr*wr - xi*wi ; comp 1: R_im = xr*wi + xi*wr
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int m = at * 16 + (row >> 1);
        if (write) {
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
    const dim3 g(grid), blk(kThreads);
    switch (nct) {
    case 1: hipLaunchKernelGGL(mfma_kernel<1>, g, blk, lds_bytes, s, a); break;
    case 2: hipLaunchKernelGGL(mfma_kernel<2>, g, blk, lds_bytes, s, a); break;
    case 4: hipLaunchKernelGGL(mfma_kernel<4>, g, blk, lds_bytes, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

size_t mfma_lds_bytes(int nct, int ct, int rep_stride)
{
    static_assert(sizeof(ChanInfo) * 20 <= 1024, "channel table must fit its 1 KB slot");
    return (size_t)1024 + (size_t)32 * kXStride * sizeof(float) + (size_t)nct * ct * rep_stride * sizeof(float);
}

} // namespace gat
