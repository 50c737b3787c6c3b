// mfma_probe.hip -- what does a dependent chain of v_mfma_f32_32x32x2_f32 cost per instruction on
// gfx950 when other instructions sit between the MFMAs?  One wave per SIMD (256 threads/block, one block per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VARIANT>
__global__ void __launch_bounds__(256) probe(float *out, const float *in, int iters, unsigned long long *cyc)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    float p = 1.0f, q = 0.0f;
    const float wr = in[1], wi = in[2];
    const float *row = lds + (threadIdx.x & 63);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (VARIANT == 0) { // bare dependent chain, constant operands
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            } else if (VARIANT == 1) { // + 4 VALU rotation + 1 mul feeding the NEXT mfma's B (dependent operand)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b * p, acc, 0, 0, 0);
                const float tp = __builtin_fmaf(p, wr, -(q * wi));
                q = __builtin_fmaf(p, wi, q * wr);
                p = tp;
            } else if (VARIANT == 2) { // + two LDS reads per MFMA used 8 MFMAs later
                const float x = row[(it * 8 + u) & 1023], y = row[2048 + ((it * 8 + u) & 1023)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b * p, acc, 0, 0, 0);
                const float tp = __builtin_fmaf(p, wr, -(q * wi));
                q = __builtin_fmaf(p, wi, q * wr);
                p = tp + x * 1e-30f + y * 1e-30f;
            } else if (VARIANT == 3) { // operands come straight from LDS reads issued just before (latency exposed)
                const float x = row[(it * 8 + u) & 1023], y = row[2048 + ((it * 8 + u) & 1023)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + p + q;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *in, *out; unsigned long long *cyc;
    CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = 0.001f * (i % 97) + 0.5f;
    h[1] = 0.9999f; h[2] = 0.0141f;
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    const int iters = 2000;
    unsigned long long hc[256];
#define RUN(V, name) do { hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 0, 0, out, in, iters, cyc); CK(hipDeviceSynchronize()); \
    hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 0, 0, out, in, iters, cyc); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost)); double m = 0; for (int i = 0; i < 256; ++i) m += hc[i]; \
    printf("%-60s %.1f cycles per MFMA\n", name, m / 256 / (iters * 8.0)); } while (0)
    RUN(0, "bare dependent chain");
    RUN(1, "+ rotation (4 VALU) + product feeding the next MFMA");
    RUN(2, "+ 2 LDS reads per MFMA (consumed later)");
    RUN(3, "operands straight from LDS reads (latency exposed)");
    return 0;
}
