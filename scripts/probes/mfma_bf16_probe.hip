// Issue-rate probe for v_mfma_f32_32x32x16_bf16 on the whole chip: waves/SIMD x accumulators, wall clock and
// s_memtime.  hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_bf16_probe.hip -o build/mfma_bf16_probe && ./build/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void __launch_bounds__(1024) probe(float *out, unsigned long long *ticks, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 x, w;
    for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + threadIdx.x % 3); w[i] = (short)(0x3f80 + i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *ticks;
    hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&ticks, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int nacc : {1, 4}) {
            const int threads = 256 * wps, blocks = 256;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (nacc == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
                else hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            const double n_per_simd = (double)iters * nacc * wps;
            printf("waves/SIMD %d acc %d: %.3f ms, %.1f ns per MFMA per SIMD, %.2f PFLOP/s, memtime ticks per MFMA per SIMD %.1f\n", wps, nacc, ms,
                   ms * 1e6 / n_per_simd, 256.0 * 4 * n_per_simd * 32768 / (ms * 1e-3) / 1e15, (double)t / n_per_simd);
        }
    }
    return 0;
}
