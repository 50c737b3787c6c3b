// mfma_i8_probe.hip -- operand layout and issue rate of the gfx950 int8 MFMAs (v_mfma_i32_16x16x64_i8,
// v_mfma_i32_32x32x32_i8) that the int8-ingest kernel (gat_i8mfma.hip) builds on.
//   layout hypothesis (same scheme as the bf16 forms): A is M x K, B is K x N, D is M x N;
//     16x16x64: lane l holds A[row = l % 16][k = 16 * (l / 16) .. + 15] (16 bytes), B[k = 16 * (l / 16) .. + 15][col = l % 16],
//               D[row = 4 * (l / 16) + v][col = l % 16], v = 0 .. 3
//     32x32x32: lane l holds A[row = l % 32][k = 16 * (l / 32) .. + 15], B[k = ...][col = l % 32],
//               D[row = 8 * (v / 4) + 4 * (l / 32) + v % 4][col = l % 32], v = 0 .. 15
// Build: hipcc -O2 --offload-arch=gfx950 scripts/probes/mfma_i8_probe.hip -o build/mfma_i8_probe && ./build/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const i32x4 *a, const i32x4 *b, i32x4 *d)
{
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}
__global__ void k32(const i32x4 *a, const i32x4 *b, i32x16 *d)
{
    i32x16 acc = {};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}
// issue rate: a chain of independent MFMAs per wave, WPS waves per SIMD
template <int SHAPE, int NACC>
__global__ void rate(const i32x4 *a, const i32x4 *b, int *out, int iters, long long *cycles)
{
    const i32x4 av = a[threadIdx.x & 63], bv = b[threadIdx.x & 63];
    const long long t0 = clock64();
    if constexpr (SHAPE == 16) {
        i32x4 acc[NACC] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc[u], 0, 0, 0);
        int s = 0;
        for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        i32x16 acc[NACC] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[u], 0, 0, 0);
        int s = 0;
        for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][15];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = clock64() - t0;
}

int main()
{
    // ---- layouts
    for (int shape : {16, 32}) {
        const int M = shape, N = shape, K = shape == 16 ? 64 : 32;
        std::vector<signed char> A(M * K), B(K * N);
        for (int i = 0; i < M; ++i)
            for (int k = 0; k < K; ++k) A[i * K + k] = (signed char)((i * 7 + k * 3) % 23 - 11);
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < N; ++j) B[k * N + j] = (signed char)((k * 5 + j * 11) % 19 - 9);
        std::vector<int> Dref(M * N, 0);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                int s = 0;
                for (int k = 0; k < K; ++k) s += (int)A[i * K + k] * (int)B[k * N + j];
                Dref[i * N + j] = s;
            }
        std::vector<signed char> fa(64 * 16), fb(64 * 16);
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 16; ++e) {
                const int r = l % shape, kb = l / shape;
                fa[l * 16 + e] = A[r * K + 16 * kb + e];
                fb[l * 16 + e] = B[(16 * kb + e) * N + r];
            }
        void *da, *db, *dd;
        hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 64 * 16 * 4);
        hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice);
        hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice);
        const int nv = shape == 16 ? 4 : 16;
        if (shape == 16) hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, (const i32x4 *)da, (const i32x4 *)db, (i32x4 *)dd);
        else hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, (const i32x4 *)da, (const i32x4 *)db, (i32x16 *)dd);
        std::vector<int> D(64 * nv);
        hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < nv; ++v) {
                const int col = l % shape;
                const int row = shape == 16 ? 4 * (l / 16) + v : 8 * (v / 4) + 4 * (l / 32) + v % 4;
                if (D[l * nv + v] != Dref[row * N + col]) ++bad;
            }
        printf("v_mfma_i32_%dx%dx%d_i8: layout hypothesis %s (%d of %d outputs differ)\n", M, N, K, bad ? "WRONG" : "holds", bad, 64 * nv);
        hipFree(da); hipFree(db); hipFree(dd);
    }
    // ---- issue rates
    void *da, *db, *dout, *dcyc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 256 * 4 * 1024 * 4); hipMalloc(&dcyc, 8);
    hipMemset(da, 1, 1024); hipMemset(db, 1, 1024);
    const int iters = 20000;
    auto run = [&](const char *name, auto kern, int nacc, int waves) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        kern<<<dim3(1024), dim3(64 * waves), 0, 0>>>((const i32x4 *)da, (const i32x4 *)db, (int *)dout, 100, (long long *)dcyc); // warm-up
        hipEventRecord(e0);
        kern<<<dim3(1024), dim3(64 * waves), 0, 0>>>((const i32x4 *)da, (const i32x4 *)db, (int *)dout, iters, (long long *)dcyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // 1024 workgroups of `waves` waves on 256 CUs x 4 SIMDs: waves per SIMD = 1024 * waves / 1024 = waves
        const double per_simd = (double)iters * nacc * waves;
        printf("%-28s %d accumulators, %d wave(s) per SIMD: %.2f ns per MFMA and SIMD (%.1f ms)\n", name, nacc, waves, ms * 1e6 / per_simd, ms);
    };
    run("16x16x64_i8", rate<16, 4>, 4, 1);
    run("16x16x64_i8", rate<16, 2>, 2, 1);
    run("16x16x64_i8", rate<16, 1>, 1, 1);
    run("16x16x64_i8", rate<16, 2>, 2, 4);
    run("32x32x32_i8", rate<32, 2>, 2, 1);
    run("32x32x32_i8", rate<32, 1>, 1, 1);
    run("32x32x32_i8", rate<32, 1>, 1, 4);
    return 0;
}
