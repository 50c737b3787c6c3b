// sync_probe2.hip -- round 3: the single-block correlate call + gat_sync is 11.5-12 us even when the vector kernel ends at its
// first instruction (scripts/history/r03/r03_latency_cuts.sh), while sync_probe.hip's empty kernel + flag is 7.0 us.  Which difference
// between the two launches costs the 4.5 us?  Every variant: one launch + host spin on a pinned flag, minimum / median,
// plus the host time inside the launch call itself and the per-launch time of 2000 launches with ONE wait at the end.
//   0  sync_probe's mode C: 1 workgroup x 64 threads, system fence + flag store
//   1  8 workgroups x 256 threads, 7 of them exit at once (the vector kernel's padded grid of one tile)
//   2  1 + 48 KB of dynamic LDS
//   3  2 + a 384-byte argument struct
//   4  3 + the library's completion protocol: barrier, agent-scope release fence, arrival counter, system-scope release store
//   5  3 + completion protocol WITHOUT the arrival counter (single-workgroup launches need none)
//   6  4 with hipExtLaunchKernelGGL-free plain hipModule-style launch through hipLaunchKernel (args array)
// Build: hipcc -O2 --offload-arch=gfx950 scripts/probes/sync_probe2.hip -o build/sync_probe2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big {
    float *p;
    unsigned *flag;
    unsigned *counter;
    unsigned v, total, mode, real_wgs;
    double pad[42];
};
static_assert(sizeof(Big) >= 368, "argument struct");

__global__ void tiny(float *p, volatile unsigned *flag, unsigned v)
{
    if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) {
        __threadfence_system();
        *flag = v;
    }
}

__global__ void __launch_bounds__(256) padded(float *p, volatile unsigned *flag, unsigned v, unsigned real_wgs)
{
    extern __shared__ float lds[];
    if (blockIdx.x >= real_wgs) return;
    if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
    if (threadIdx.x == 0) {
        __threadfence_system();
        *flag = v;
    }
}

__global__ void __launch_bounds__(256) bigargs(const Big a)
{
    extern __shared__ float lds[];
    if (blockIdx.x >= a.real_wgs) return;
    if (threadIdx.x == 0) a.p[blockIdx.x] += 1.f;
    if (a.mode == 3) {
        if (threadIdx.x == 0) {
            __threadfence_system();
            *(volatile unsigned *)a.flag = a.v;
        }
        return;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (a.mode == 4) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned arrived = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == a.total - 1u) {
                __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.flag, a.v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else {
            __hip_atomic_store(a.flag, a.v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    unsigned *counter; CK(hipMalloc(&counter, 64)); CK(hipMemset(counter, 0, 64));
    unsigned *flag; CK(hipHostMalloc((void **)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    unsigned *dflag; CK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
    CK(hipFuncSetAttribute((const void *)padded, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void *)bigargs, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int reps = 3000;
    std::vector<double> t(reps), tl(reps);
    unsigned seq = 0;
    auto launch = [&](int mode, unsigned v) {
        if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned *)dflag, v);
        else if (mode == 1) hipLaunchKernelGGL(padded, dim3(8), dim3(256), 0, s, d, (volatile unsigned *)dflag, v, 1u);
        else if (mode == 2) hipLaunchKernelGGL(padded, dim3(8), dim3(256), 48 * 1024, s, d, (volatile unsigned *)dflag, v, 1u);
        else {
            Big a{};
            a.p = d; a.flag = dflag; a.counter = counter; a.v = v; a.total = 1; a.mode = mode == 6 ? 4 : mode; a.real_wgs = 1;
            if (mode == 6) {
                void *args[] = {&a};
                (void)hipLaunchKernel((const void *)bigargs, dim3(8), dim3(256), args, 48 * 1024, s);
            } else {
                hipLaunchKernelGGL(bigargs, dim3(8), dim3(256), 48 * 1024, s, a);
            }
        }
    };
    for (int mode = 0; mode <= 6; ++mode) {
        *flag = 0;
        for (int r = -100; r < reps; ++r) {
            ++seq;
            const double t0 = now_us();
            launch(mode, seq);
            const double t1 = now_us();
            while (*(volatile unsigned *)flag != seq) {}
            if (r >= 0) t[r] = now_us() - t0, tl[r] = t1 - t0;
        }
        CK(hipStreamSynchronize(s));
        const double p0 = now_us();
        for (int r = 0; r < 2000; ++r) launch(mode, ++seq);
        CK(hipStreamSynchronize(s));
        const double per = (now_us() - p0) / 2000;
        std::sort(t.begin(), t.end());
        std::sort(tl.begin(), tl.end());
        printf("mode %d: launch + flag min %6.2f  median %6.2f us | launch call median %5.2f us | pipelined %5.2f us per launch\n", mode, t[0],
               t[reps / 2], tl[reps / 2], per);
        fflush(stdout);
    }
    return 0;
}
