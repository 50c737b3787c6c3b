// valu_issue_probe.hip -- what does one SIMD of gfx950 sustain in plain f32 vector FMAs, by waves per SIMD?
// (round 2: the fused correlator's multi-channel instances run at the rate of ONE v_fma_f32 per ~4 cycles and SIMD with
// 2 waves per SIMD; is that the machine or the kernel?)  Independent FMA chains in registers, no memory.
//   MODE 0: v_fma_f32, ILP independent chains, all operands in VGPRs
//   MODE 1: v_pk_fma_f32, ILP/2 packed chains
//   MODE 3: MODE 2 at the size of the 4-channel x 4-antenna x 4-sample step of the real kernel: 96 accumulators, 48 chips and
//           32 phasor values live, ~700 straight-line vector instructions per iteration (register count and code size)
//   MODE 2: the correlator's inner pattern: dr = xr*cr + xi*ci; di = xi*cr - xr*ci; acc[l] += chip[l]*{dr,di}  (L = 3, 4 antennas)
// Build: hipcc -O3 -fno-slp-vectorize -ffp-contract=off --offload-arch=gfx950 scripts/probes/valu_issue_probe.hip -o build/vip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int ILP>
__global__ void __launch_bounds__(256) probe(float *out, int iters, unsigned long long *clk)
{
    extern __shared__ float pad[];
    const float a = 1.0f + threadIdx.x * 1e-9f, b = 1e-9f * blockIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float s = 0.f;
    if constexpr (MODE == 0) {
        float v[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) v[i] = (float)i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < ILP; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < ILP; ++i) s += v[i];
    } else if constexpr (MODE == 1) {
        f2 v[ILP / 2];
        const f2 a2 = {a, a}, b2 = {b, b};
#pragma unroll
        for (int i = 0; i < ILP / 2; ++i) v[i] = f2{(float)i, (float)i + 0.5f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < ILP / 2; ++i) v[i] = __builtin_elementwise_fma(v[i], a2, b2);
        }
#pragma unroll
        for (int i = 0; i < ILP / 2; ++i) s += v[i][0] + v[i][1];
    } else if constexpr (MODE == 3) {
        float acc[4][4][3][2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int l = 0; l < 3; ++l) acc[k][m][l][0] = acc[k][m][l][1] = 0.f;
        float xr[4][4], xi[4][4], chip[4][4][3], pr[4][4], pi[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[m][j] = a + m + 0.25f * j, xi[m][j] = b - m + 0.5f * j;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pr[k][j] = a + k * j; pi[k][j] = b - k - j;
#pragma unroll
                for (int l = 0; l < 3; ++l) chip[k][j][l] = ((threadIdx.x >> ((k + j + l) & 7)) & 1) ? 1.f : -1.f;
            }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dr = __builtin_fmaf(xr[m][j], pr[k][j], xi[m][j] * pi[k][j]);
                        const float di = __builtin_fmaf(xi[m][j], pr[k][j], -(xr[m][j] * pi[k][j]));
#pragma unroll
                        for (int l = 0; l < 3; ++l) {
                            acc[k][m][l][0] = __builtin_fmaf(chip[k][j][l], dr, acc[k][m][l][0]);
                            acc[k][m][l][1] = __builtin_fmaf(chip[k][j][l], di, acc[k][m][l][1]);
                        }
                    }
            // keep operands changing (cheap: 16 instructions)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) pr[k][j] = __builtin_fmaf(pi[k][j], b, pr[k][j]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int l = 0; l < 3; ++l) s += acc[k][m][l][0] + acc[k][m][l][1];
    } else {
        float acc[4][3][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int l = 0; l < 3; ++l) acc[m][l][0] = acc[m][l][1] = 0.f;
        float xr[4], xi[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xr[m] = a + m, xi[m] = b - m;
        float cr = a, ci = b, c0 = (threadIdx.x & 1) ? 1.f : -1.f, c1 = (threadIdx.x & 2) ? 1.f : -1.f, c2 = (blockIdx.x & 1) ? 1.f : -1.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float dr = __builtin_fmaf(xr[m], cr, xi[m] * ci), di = __builtin_fmaf(xi[m], cr, -(xr[m] * ci));
                acc[m][0][0] = __builtin_fmaf(c0, dr, acc[m][0][0]); acc[m][0][1] = __builtin_fmaf(c0, di, acc[m][0][1]);
                acc[m][1][0] = __builtin_fmaf(c1, dr, acc[m][1][0]); acc[m][1][1] = __builtin_fmaf(c1, di, acc[m][1][1]);
                acc[m][2][0] = __builtin_fmaf(c2, dr, acc[m][2][0]); acc[m][2][1] = __builtin_fmaf(c2, di, acc[m][2][1]);
            }
            // rotate the phasor (keeps the loop from being hoisted): 4 more
            const float t = __builtin_fmaf(cr, a, -(ci * b));
            ci = __builtin_fmaf(cr, b, ci * a);
            cr = t;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int l = 0; l < 3; ++l) s += acc[m][l][0] + acc[m][l][1];
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (s == 123.456f) out[0] = s + pad[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int ILP>
static int run(const char *name, int occ, int iters, double vinst_per_iter)
{
    float *o; unsigned long long *clk, h[2];
    CK(hipMalloc(&o, 4)); CK(hipMalloc(&clk, 16));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int cus = 256;
    // occ workgroups of 4 waves per CU: occ waves per SIMD; the dynamic LDS size keeps more from being resident
    const int lds = (160 * 1024 / occ) & ~1023;
    CK(hipFuncSetAttribute((const void *)probe<MODE, ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((probe<MODE, ILP>), dim3(cus * occ), dim3(256), lds > 65536 ? lds : lds, 0, o, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, ILP>), dim3(cus * occ), dim3(256), lds, 0, o, iters, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    CK(hipGetLastError());
    float ms; hipEventElapsedTime(&ms, a, b);
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double cyc = (double)h[0], mhz = cyc / ((double)h[1] / 100.0); // wall_clock64: 100 MHz
    const double vinst = vinst_per_iter * iters;                          // per wave
    printf("%-28s waves/SIMD %d: %.3f ms, shader clock %.0f MHz, %.2f cycles per wave-instruction per SIMD (%.2f per wave), %.1f TFLOP/s\n",
           name, occ, ms, mhz, cyc / (vinst * occ), cyc / vinst, vinst * 64 * 2 * (MODE == 1 ? 2 : 1) * 4.0 * cus * occ / (ms * 1e-3) / 1e12);
    hipFree(o); hipFree(clk);
    return 0;
}

int main()
{
    const int iters = 20000;
    for (int occ : {1, 2, 3, 4, 8}) {
        if (run<0, 8>("v_fma_f32 x8 chains", occ, iters, 32)) return 1;
        if (run<0, 16>("v_fma_f32 x16 chains", occ, iters, 64)) return 1;
        if (run<1, 16>("v_pk_fma_f32 x8 chains", occ, iters, 32)) return 1;
        if (run<2, 0>("correlator pattern (4 ant, 3 taps)", occ, iters, 4 * 10 + 4)) return 1;
        if (occ <= 2 && run<3, 0>("correlator step, 4 ch x 4 ant x 4 samples", occ, iters / 10, 64 * 10 + 16)) return 1;
    }
    return 0;
}
