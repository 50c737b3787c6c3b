// fma_mix_probe.hip -- does v_fma_mix_f32 (f16 multiplicand taken from one half of a register, f32 multiplier and
// accumulator) issue at the rate of v_fma_f32 on gfx950?  (round 3: code chips kept as f16 pairs would halve the chip
// registers and the LDS replica of the fused correlator if the mixed instruction costs nothing extra.)
// Independent chains in registers, no memory.  Build: hipcc -O3 --offload-arch=gfx950 scripts/probes/fma_mix_probe.hip -o build/fmp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) probe(float *out, int iters, unsigned long long *clk)
{
    extern __shared__ float pad[];
    constexpr int ILP = 16;
    const float a = 1.0f + threadIdx.x * 1e-9f;
    unsigned chips = (threadIdx.x & 1) ? 0x3c00bc00u : 0xbc003c00u; // {+1, -1} / {-1, +1} as f16 pairs
    float v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = (float)i;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if constexpr (MODE == 0) v[i] = __builtin_fmaf(a, a, v[i]);
                else if (i & 1) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(chips), "v"(a));
                else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(chips), "v"(a));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    if (s == 123.456f) out[0] = s + pad[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
static int run(const char *name, int occ, int iters)
{
    float *o; unsigned long long *clk, h[2];
    CK(hipMalloc(&o, 4)); CK(hipMalloc(&clk, 16));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int cus = 256, lds = (160 * 1024 / occ) & ~1023;
    CK(hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((probe<MODE>), dim3(cus * occ), dim3(256), lds, 0, o, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE>), dim3(cus * occ), dim3(256), lds, 0, o, iters, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    CK(hipGetLastError());
    float ms; hipEventElapsedTime(&ms, a, b);
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double cyc = (double)h[0], mhz = cyc / ((double)h[1] / 100.0), vinst = 64.0 * iters;
    printf("%-16s waves/SIMD %d: %.3f ms, shader clock %.0f MHz, %.2f cycles per wave-instruction per SIMD (%.2f per wave)\n", name, occ, ms, mhz,
           cyc / (vinst * occ), cyc / vinst);
    hipFree(o); hipFree(clk);
    return 0;
}

int main()
{
    for (int occ : {1, 2, 3, 4}) {
        if (run<0>("v_fma_f32", occ, 20000)) return 1;
        if (run<1>("v_fma_mix_f32", occ, 20000)) return 1;
    }
    return 0;
}
