// lds_unaligned_probe.hip -- does ds_read_b128 work at 4-byte-aligned (not 16-byte-aligned) LDS addresses on gfx950, and at what
// cost?  (round 2: the correlator reads the chips of 4 consecutive samples for a tap offset o; entry = 4*lane + o.)
// Build: hipcc -O3 --offload-arch=gfx950 scripts/probes/lds_unaligned_probe.hip -o build/lup
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: ds_read_b128 at byte address 16*lane + 4*o   MODE 1: 2 x ds_read2_b32   MODE 2: 4 x ds_read_b32 (planar layout equivalent: conflict-free)
template <int MODE>
__global__ void __launch_bounds__(256) probe(float *out, int o, int iters, unsigned long long *clk, int check)
{
    __shared__ float lds[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) lds[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    unsigned addr = base + threadIdx.x * 16 + o * 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        f4 v;
        if constexpr (MODE == 0) {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        } else if constexpr (MODE == 1) {
            float a0, a1, a2, a3;
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(*(double *)&a0), "=v"(*(double *)&a2) : "v"(addr) : "memory");
            v = f4{a0, a1, a2, a3};
        } else {
            // planar: plane p at p*1024 floats; lane reads slot lane of each plane (what the kernel does today)
            unsigned pa = base + threadIdx.x * 4;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4096\n\tds_read_b32 %2, %4 offset:8192\n\tds_read_b32 %3, %4 offset:12288\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "v"(pa) : "memory");
        }
        acc += v;
        addr ^= (it & 1) ? 0u : 0u; // keep addr live
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (check) {
        // expected: sum over iters of lds[4*tid + o + j]
        f4 want = {(float)(4 * threadIdx.x + o), (float)(4 * threadIdx.x + o + 1), (float)(4 * threadIdx.x + o + 2), (float)(4 * threadIdx.x + o + 3)};
        int bad = 0;
        for (int j = 0; j < 4; ++j) bad |= acc[j] != want[j] * iters;
        if (MODE != 2 && bad) atomicAdd((int *)out, 1);
    }
    if (acc[0] == 123.456f) out[1] = acc[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE> static int run(const char *name, int o)
{
    float *outp; unsigned long long *clk, h;
    CK(hipMalloc(&outp, 8)); CK(hipMalloc(&clk, 8)); CK(hipMemset(outp, 0, 8));
    hipLaunchKernelGGL(probe<MODE>, dim3(256 * 4), dim3(256), 0, 0, outp, o, 4, clk, 1); // correctness (exact small sums)
    CK(hipDeviceSynchronize());
    int bad; CK(hipMemcpy(&bad, outp, 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(probe<MODE>, dim3(256 * 4), dim3(256), 0, 0, outp, o, 4096, clk, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    printf("%-22s offset %d dwords: %s, %.1f cycles per read+wait (4 workgroups of 4 waves per CU)\n", name, o, bad ? "WRONG VALUES" : "values ok", (double)h / 4096);
    return 0;
}

int main()
{
    for (int o = 0; o < 5; ++o) if (run<0>("ds_read_b128", o)) return 1;
    for (int o = 0; o < 3; ++o) if (run<1>("2 x ds_read2_b32", o)) return 1;
    if (run<2>("4 x ds_read_b32 planar", 0)) return 1;
    return 0;
}
