// stream_pattern_probe.hip -- pure-read probe of dc_kernel's two access patterns on MI355X (round 2):
//   (a) "4 antennas per workgroup":   a workgroup streams 8 planes, 4 KB contiguous per plane and step (4 waves side by side)
//   (b) "16 antennas per workgroup":  a workgroup streams 32 planes, 1 KB contiguous per plane and step (every wave its own 8 planes)
// same total bytes, same loads per lane (8 x 16 B in flight per step), non-temporal loads, 256-thread workgroups.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/stream_pattern_probe.hip -o /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// planes: P planes of plane_f4 float4 each; block b owns samples [b*N4, (b+1)*N4) of every plane
// WAVE_PLANES = true: wave w reads planes 8w..8w+7, all waves the same 64-float4 (1 KB) window per step (pattern b)
// WAVE_PLANES = false: all waves read the same 8 planes, wave w the w-th 1 KB of a 4 KB window (pattern a)
template <bool WAVE_PLANES>
__global__ void __launch_bounds__(256) read_pattern(const f4 *__restrict__ in, size_t plane_f4, int N4, int planes_per_wg, float *out)
{
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = gridDim.y;                 // plane groups (of planes_per_wg planes)
    const size_t b = blockIdx.x;
    const int pg = blockIdx.y;
    const int step_f4 = WAVE_PLANES ? 64 : 256;   // float4 per plane and step
    const int first_plane = pg * planes_per_wg + (WAVE_PLANES ? 8 * wave : 0);
    (void)groups;
    for (int c = 0; c + step_f4 <= N4; c += step_f4) {
        f4 v[8];
        const int off = c + (WAVE_PLANES ? lane : threadIdx.x);
#pragma unroll
        for (int p = 0; p < 8; ++p) v[p] = __builtin_nontemporal_load(in + (size_t)(first_plane + p) * plane_f4 + b * N4 + off);
#pragma unroll
        for (int p = 0; p < 8; ++p) { acc.x += v[p].x; acc.y += v[p].y; acc.z += v[p].z; acc.w += v[p].w; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}

template <typename F> static float time_ms(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    // configs[3] shard: 32 planes (16 antennas x re/im) x 512 blocks x 50 000 samples (12 500 float4) = 3.28 GB
    const int P = 32, B = 512, N4 = 12500;
    const size_t plane_f4 = (size_t)B * N4;
    f4 *d; float *o;
    CK(hipMalloc(&d, P * plane_f4 * sizeof(f4))); CK(hipMalloc(&o, 4));
    CK(hipMemset(d, 0, P * plane_f4 * sizeof(f4)));
    const double gb = (double)P * plane_f4 * 16 / 1e9;
    // (a) 8 planes per workgroup, 4 KB per plane and step: grid (B, 4)
    float ta = time_ms([&] { hipLaunchKernelGGL(read_pattern<false>, dim3(B, 4), dim3(256), 0, 0, d, plane_f4, N4, 8, o); }, 20);
    // (b) 32 planes per workgroup, 1 KB per plane and step, block split in 4 sample ranges to keep the grid equal: emulate with grid (B*4) of N4/4
    float tb = time_ms([&] { hipLaunchKernelGGL(read_pattern<true>, dim3(B, 1), dim3(256), 0, 0, d, plane_f4, N4, 32, o); }, 20);
    printf("pattern a (8 planes/WG, 4 KB per plane-step, grid %d): %.4f ms  %.0f GB/s\n", B * 4, ta, gb / ta * 1e3);
    printf("pattern b (32 planes/WG, 1 KB per plane-step, grid %d): %.4f ms  %.0f GB/s\n", B, tb, gb / tb * 1e3);
    // (b') the same with the block cut into 4 sample ranges (grid as large as a)
    f4 *d2 = d;
    float tb2 = time_ms([&] { hipLaunchKernelGGL(read_pattern<true>, dim3(B * 4, 1), dim3(256), 0, 0, d2, plane_f4, N4 / 4, 32, o); }, 20);
    printf("pattern b' (32 planes/WG, 1 KB per plane-step, grid %d x quarter blocks): %.4f ms  %.0f GB/s\n", B * 4, tb2, gb / tb2 * 1e3);
    return 0;
}
