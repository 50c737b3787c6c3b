// doorbell_bar_probe.hip -- can the HOST write a doorbell that lives in DEVICE memory (fine-grained allocation reached through
// the PCIe BAR), and is a ring through it seen sooner than through pinned host memory?  The resident kernel of
// doorbell_probe.hip mode A (1 workgroup x 64, doorbell -> flag in pinned host memory), the doorbell once in pinned host
// memory and once in device memory.  The host access is tried in a CHILD process first (a fault there costs nothing).
// Build: hipcc -O2 --offload-arch=gfx950 scripts/probes/doorbell_bar_probe.hip -o build/doorbell_bar_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr unsigned QUIT = 0xFFFFFFFFu;

__global__ void __launch_bounds__(64) resident(const unsigned *bell, unsigned *flag, unsigned max_calls, long long idle_ticks, int bell_on_device)
{
    unsigned last = 0, calls = 0;
    long long t_last = wall_clock64();
    for (;;) {
        unsigned seq;
        if (bell_on_device == 2) // the scalar unit's load, past its cache (glc): is the trip shorter than the vector memory path's?
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seq) : "s"(bell) : "memory");
        else
            seq = __hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (seq == QUIT) break;
        if (seq != last) {
            last = seq;
            if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            t_last = wall_clock64();
            if (++calls >= max_calls) break;
            continue;
        }
        if (wall_clock64() - t_last > idle_ticks) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    // 1. is fine-grained device memory writable by the host here?  (tried in a child process)
    pid_t pid = fork();
    if (pid == 0) {
        unsigned *p = nullptr;
        if (hipExtMallocWithFlags((void **)&p, 4096, hipDeviceMallocFinegrained) != hipSuccess) _exit(3);
        *(volatile unsigned *)p = 12345u; // faults if the allocation is not mapped for the host
        unsigned back = 0;
        if (hipMemcpy(&back, p, 4, hipMemcpyDeviceToHost) != hipSuccess) _exit(4);
        _exit(back == 12345u ? 0 : 5);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    const bool ok = WIFEXITED(st) && WEXITSTATUS(st) == 0;
    printf("host store into fine-grained device memory: %s (child %s %d)\n", ok ? "works" : "NOT available", WIFEXITED(st) ? "exit" : "signal", WIFEXITED(st) ? WEXITSTATUS(st) : WTERMSIG(st));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *hbell, *flag, *dbell = nullptr;
    CK(hipHostMalloc((void **)&hbell, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    if (ok) CK(hipExtMallocWithFlags((void **)&dbell, 4096, hipDeviceMallocFinegrained));
    const int reps = 3000;
    for (int mode = 0; mode < (ok ? 3 : 1); ++mode) {
        volatile unsigned *bell = mode ? dbell : hbell;
        *bell = 0;
        *flag = 0;
        resident<<<1, 64, 0, s>>>((const unsigned *)bell, flag, reps + 100, 2000000, mode);
        std::vector<double> t(reps);
        bool lost = false;
        for (int r = -50; r < reps && !lost; ++r) {
            const unsigned seq = (unsigned)(r + 51);
            const double t0 = now_us();
            __atomic_store_n((unsigned *)bell, seq, __ATOMIC_RELEASE);
            if (mode) __builtin_ia32_sfence(); // the BAR is mapped write-combining: push the store out of the core's buffer
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
                if (now_us() - t0 > 2e5) { lost = true; break; }
            if (r >= 0) t[r] = now_us() - t0;
        }
        __atomic_store_n((unsigned *)bell, QUIT, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(s));
        std::sort(t.begin(), t.end());
        printf("doorbell in %-26s min %5.2f  median %5.2f  p99 %5.2f us%s\n", mode == 2 ? "DEVICE memory, s_load glc" : mode ? "DEVICE memory (BAR)" : "pinned host memory", t[0], t[reps / 2], t[reps * 99 / 100], lost ? "  (LOST A CALL)" : "");
    }
    return 0;
}
