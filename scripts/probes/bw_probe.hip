// bw_probe.hip -- HBM read-ceiling probe for MI355X: what can a pure streaming-read kernel reach
// on a 2.6 GB buffer?  (cdna_hip_programming.md rule 10: a ceiling needs a known-good reference
// measured on the same hardware.)  Build: hipcc -O3 --offload-arch=gfx950 scripts/probes/bw_probe.hip -o build/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) read_sum(const f4 *__restrict__ in, size_t n4, float *out)
{
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            v[u] = NT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { f4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}

// contiguous-per-workgroup variant: each workgroup streams its own contiguous 640 KB segment split in
// 8 sub-streams (the access pattern of dc_kernel at M = 4 planar)
template <bool NT>
__global__ void __launch_bounds__(256) read_planes(const f4 *__restrict__ in, size_t n4_per_plane, int N4, float *out)
{
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t b = blockIdx.x;
    for (int c = threadIdx.x; c < N4; c += 256) {
        f4 v[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const f4 *q = in + (size_t)p * n4_per_plane + b * N4 + c;
            v[p] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) { acc.x += v[p].x; acc.y += v[p].y; acc.z += v[p].z; acc.w += v[p].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}

template <typename F>
static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const int B = 4096, N = 20000, M = 4;
    const size_t plane = (size_t)B * N;           // floats per plane
    const size_t total = plane * 2 * M;           // floats
    float *buf, *out;
    CK(hipMalloc(&buf, total * sizeof(float)));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 0x3c, total * sizeof(float)));
    const double bytes = (double)total * 4;
    const size_t n4 = total / 4;
    printf("buffer %.1f MB\n", bytes / 1e6);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        float ms;
        ms = time_ms([&] { hipLaunchKernelGGL((read_sum<1, false>), dim3(grid), dim3(256), 0, 0, (const f4 *)buf, n4, out); }, 15);
        printf("read_sum u1      grid %5d: %.4f ms  %.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((read_sum<4, false>), dim3(grid), dim3(256), 0, 0, (const f4 *)buf, n4, out); }, 15);
        printf("read_sum u4      grid %5d: %.4f ms  %.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((read_sum<8, false>), dim3(grid), dim3(256), 0, 0, (const f4 *)buf, n4, out); }, 15);
        printf("read_sum u8      grid %5d: %.4f ms  %.1f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((read_sum<8, true>), dim3(grid), dim3(256), 0, 0, (const f4 *)buf, n4, out); }, 15);
        printf("read_sum u8 nt   grid %5d: %.4f ms  %.1f GB/s\n", grid, ms, bytes / ms / 1e6);
    }
    float ms = time_ms([&] { hipLaunchKernelGGL((read_planes<false>), dim3(B), dim3(256), 0, 0, (const f4 *)buf, plane / 4, N / 4, out); }, 15);
    printf("read_planes      grid %5d: %.4f ms  %.1f GB/s\n", B, ms, bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((read_planes<true>), dim3(B), dim3(256), 0, 0, (const f4 *)buf, plane / 4, N / 4, out); }, 15);
    printf("read_planes nt   grid %5d: %.4f ms  %.1f GB/s\n", B, ms, bytes / ms / 1e6);
    return 0;
}
