// doorbell_probe.hip -- round 4: what a call costs when NO launch is in it.  One bounded-lifetime resident kernel polls a
// doorbell in pinned host memory; the host writes a sequence number, the kernel answers into a pinned flag, the host spins.
// The kernel ends by itself whatever the host does: after `max_calls` answers, after `idle_us` without a ring, after
// `life_us` in total, or when the host rings QUIT -- every wave of every workgroup reaches one of these exits
// (wall_clock64 is the 100 MHz constant clock).
//   A  1 workgroup x 64 threads, doorbell -> flag, nothing else (the floor of the mechanism)
//   B  1 workgroup x 256 threads: thread 0 polls, LDS broadcast + barrier, 64-byte parameter record read from the same host
//      line as the doorbell, every thread reads 4 floats of a 16 KB device buffer with system-scope visibility (a fresh
//      signal written by a copy), a block reduction, 24 floats of results + flag written to host memory
//   C  B with 8 workgroups that all poll; arrival counter in device memory, the last one answers
//   D  A, but the host waits 200 us between calls (a receiver's cadence is 1 ms: does an idle poller answer as fast?)
//   L  for comparison on the same box: an empty launch + pinned flag (sync_probe.hip mode C)
// Build: hipcc -O2 --offload-arch=gfx950 scripts/probes/doorbell_probe.hip -o build/doorbell_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr unsigned QUIT = 0xFFFFFFFFu;

struct Bell {              // one 64-byte line in pinned host memory
    unsigned seq;          // written last by the host
    unsigned pad;
    double prm[6];         // the call's parameters
    unsigned seq_tail;     // = seq (a torn read shows as seq != seq_tail)
    unsigned pad2;
};
static_assert(sizeof(Bell) == 64, "one line");

struct Args {
    const Bell *bell;         // pinned host
    unsigned *flag;           // pinned host
    float *result;            // pinned host, 24 floats
    const float *signal;      // device, 4096 floats
    unsigned *counter;        // device
    unsigned *exit_code;      // pinned host: why the kernel ended
    unsigned max_calls, mode;
    long long idle_ticks, life_ticks;
};

__device__ inline unsigned ld_sys(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ void __launch_bounds__(256) resident(const Args a)
{
    __shared__ unsigned s_seq;
    __shared__ double s_prm[6];
    __shared__ float s_red[4];
    unsigned last = 0, calls = 0, why = 0;
    const long long t_start = wall_clock64();
    long long t_last = t_start;
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned seq = last;
            for (;;) {
                seq = ld_sys(&a.bell->seq);
                if (seq != last) {
                    if (seq == QUIT) break;
                    if (a.mode != 0) { // parameters ride in the same line; a torn read is retried
                        const unsigned tail = ld_sys(&a.bell->seq_tail);
                        if (tail != seq) continue;
                        for (int i = 0; i < 6; ++i) s_prm[i] = __builtin_nontemporal_load(&a.bell->prm[i]);
                        if (ld_sys(&a.bell->seq) != seq) continue;
                    }
                    break;
                }
                const long long now = wall_clock64();
                if (now - t_last > a.idle_ticks) { seq = QUIT; why = 2; break; }
                if (now - t_start > a.life_ticks) { seq = QUIT; why = 3; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            s_seq = seq;
        }
        __syncthreads();
        const unsigned seq = s_seq;
        __syncthreads();
        if (seq == QUIT) { if (why == 0) why = 1; break; }
        last = seq;
        t_last = wall_clock64();
        if (a.mode == 0) {
            if (threadIdx.x == 0) __hip_atomic_store(a.flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            // the signal may have been rewritten by a copy engine since the last call: read it past this XCD's L2
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            const float4 v = reinterpret_cast<const float4 *>(a.signal)[threadIdx.x + 256 * (blockIdx.x & 3)];
            float s = (v.x + v.y + v.z + v.w) * (float)s_prm[0];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
            __syncthreads();
            bool answer = true;
            if (gridDim.x > 1) {
                if (threadIdx.x == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    const unsigned arrived = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    s_seq = arrived == gridDim.x - 1u;
                    if (arrived == gridDim.x - 1u) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                answer = s_seq != 0;
                __syncthreads();
            }
            if (answer) {
                if (threadIdx.x < 24) a.result[threadIdx.x] = s_red[threadIdx.x & 3] + (float)threadIdx.x;
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(a.flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (++calls >= a.max_calls) { why = 4; break; }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(a.exit_code, why, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void tiny(float *p, volatile unsigned *flag, unsigned v)
{
    if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
    if (threadIdx.x == 0) {
        __threadfence_system();
        *flag = v;
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    Bell *bell; unsigned *flag, *exit_code; float *result;
    CK(hipHostMalloc((void **)&bell, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&exit_code, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&result, 128, hipHostMallocDefault));
    float *signal; unsigned *counter;
    CK(hipMalloc((void **)&signal, 16384));
    CK(hipMalloc((void **)&counter, 4));
    CK(hipMemset(counter, 0, 4));
    std::vector<float> h(4096, 1.f);
    CK(hipMemcpy(signal, h.data(), 16384, hipMemcpyHostToDevice));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    printf("wall clock %d kHz\n", clk_khz);
    const long long per_us = clk_khz / 1000;
    const int reps = 3000;
    struct Mode { const char *name; unsigned mode, wgs, threads; double gap_us; };
    const Mode modes[] = {{"A  1 wg x 64, doorbell -> flag", 0, 1, 64, 0}, {"B  1 wg x 256, params + 16 KB signal + 24 results", 1, 1, 256, 0},
                          {"C  8 wgs x 256, all poll, last arrival answers", 1, 8, 256, 0}, {"D  A with 200 us between calls", 0, 1, 64, 200}};
    for (const Mode &m : modes) {
        memset(bell, 0, 64);
        *flag = 0;
        *exit_code = 99;
        Args a{bell, flag, result, signal, counter, exit_code, (unsigned)reps + 100u, m.mode, 20000 * per_us /* 20 ms idle */, 3000000 * per_us /* 3 s */};
        resident<<<dim3(m.wgs), dim3(m.threads), 0, s>>>(a);
        CK(hipGetLastError());
        std::vector<double> t(reps);
        bool lost = false;
        for (int r = -50; r < reps && !lost; ++r) {
            const unsigned seq = (unsigned)(r + 51);
            if (m.gap_us > 0) { const double w = now_us(); while (now_us() - w < m.gap_us) {} }
            const double t0 = now_us();
            bell->prm[0] = 1.0 + seq;
            __atomic_store_n(&bell->seq_tail, seq, __ATOMIC_RELEASE);
            __atomic_store_n(&bell->seq, seq, __ATOMIC_RELEASE);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
                if (now_us() - t0 > 2e5) { lost = true; break; } // the kernel went away (idle / lifetime exit): stop
            if (r >= 0) t[r] = now_us() - t0;
        }
        __atomic_store_n(&bell->seq, QUIT, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(s));
        std::sort(t.begin(), t.end());
        printf("%-52s min %6.2f  median %6.2f  p99 %6.2f us   exit code %u%s   result[0] %.0f\n", m.name, t[0], t[reps / 2], t[reps * 99 / 100], *exit_code,
               lost ? "  (LOST A CALL)" : "", result[0]);
        fflush(stdout);
    }
    // the kernel leaves by itself: ring nothing, wait for the idle exit
    {
        memset(bell, 0, 64);
        *exit_code = 99;
        Args a{bell, flag, result, signal, counter, exit_code, 10u, 0u, 5000 * per_us, 3000000 * per_us};
        const double t0 = now_us();
        resident<<<dim3(1), dim3(64), 0, s>>>(a);
        CK(hipStreamSynchronize(s));
        printf("idle exit: kernel with a 5 ms idle limit and no ring ended after %.1f ms, exit code %u (2 = idle)\n", (now_us() - t0) * 1e-3, *exit_code);
    }
    // while a resident kernel polls: does an ordinary launch on another stream still run at its usual latency?
    {
        float *p; CK(hipMalloc((void **)&p, 1024)); CK(hipMemset(p, 0, 1024));
        std::vector<double> t(reps);
        for (int pass = 0; pass < 2; ++pass) {
            memset(bell, 0, 64);
            if (pass == 1) {
                Args a{bell, flag, result, signal, counter, exit_code, 10u, 0u, 2000000 * per_us, 3000000 * per_us};
                resident<<<dim3(1), dim3(64), 0, s>>>(a);
            }
            unsigned *f2; CK(hipHostMalloc((void **)&f2, 64, hipHostMallocDefault)); *f2 = 0;
            for (int r = -50; r < reps; ++r) {
                const unsigned v = (unsigned)(r + 51);
                const double t0 = now_us();
                tiny<<<dim3(1), dim3(64), 0, s2>>>(p, f2, v);
                while (__atomic_load_n(f2, __ATOMIC_ACQUIRE) != v) {}
                if (r >= 0) t[r] = now_us() - t0;
            }
            std::sort(t.begin(), t.end());
            printf("L  empty launch + pinned flag%s: min %6.2f  median %6.2f us\n", pass ? " WHILE a resident kernel polls" : "", t[0], t[reps / 2]);
            if (pass == 1) { __atomic_store_n(&bell->seq, QUIT, __ATOMIC_RELEASE); CK(hipStreamSynchronize(s)); }
            CK(hipStreamSynchronize(s2));
        }
    }
    return 0;
}
