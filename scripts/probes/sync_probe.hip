// sync_probe.hip -- what does "launch one tiny kernel and wait for it" cost from native code on this machine, by the
// way the host waits?  (round 3: a single-block correlate call + gat_sync is 15-17 us from C, of which the device needs 5;
// the reference's harness times exactly this, @benchmark CUDA.@sync ..., src/benchmarks.jl:120)
//   A  hipStreamSynchronize                      B  hipEventRecord + spin on hipEventQuery
//   C  the kernel's last thread writes a flag in pinned host memory, the host spins on it
//   D  hipStreamWriteValue32 behind the kernel into pinned host memory, the host spins on it
// each with the device's default scheduling flags and with hipDeviceScheduleSpin.
// Build: hipcc -O2 --offload-arch=gfx950 scripts/probes/sync_probe.hip -o build/sync_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void tiny(float *p, volatile unsigned *flag, unsigned v)
{
    if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) {
        __threadfence_system();
        *flag = v;
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run(const char *tag)
{
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    unsigned *flag; CK(hipHostMalloc((void **)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    unsigned *dflag; CK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int reps = 3000;
    std::vector<double> t(reps);
    for (int mode = 0; mode < 4; ++mode) {
        unsigned seq = 0;
        *flag = 0;
        int ok = 1;
        for (int r = -100; r < reps; ++r) {
            ++seq;
            const double t0 = now_us();
            if (mode == 0) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned *)nullptr, 0u);
                CK(hipStreamSynchronize(s));
            } else if (mode == 1) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned *)nullptr, 0u);
                CK(hipEventRecord(ev, s));
                while (hipEventQuery(ev) == hipErrorNotReady) {}
            } else if (mode == 2) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned *)dflag, seq);
                while (*(volatile unsigned *)flag != seq) {}
            } else {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (volatile unsigned *)nullptr, 0u);
                if (hipStreamWriteValue32(s, dflag, seq, 0) != hipSuccess) { ok = 0; (void)hipGetLastError(); break; }
                while (*(volatile unsigned *)flag != seq) {}
            }
            if (r >= 0) t[r] = now_us() - t0;
        }
        CK(hipStreamSynchronize(s));
        if (!ok) { printf("%-8s mode %c: not supported\n", tag, 'A' + mode); continue; }
        std::sort(t.begin(), t.end());
        printf("%-8s mode %c: min %6.2f  median %6.2f  p90 %6.2f us\n", tag, 'A' + mode, t[0], t[reps / 2], t[reps * 9 / 10]);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 's') {
        CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
        return run("spin");
    }
    return run("default");
}
