// Probe: which part of a raw buffer load's address takes part in the range check (num_records)?
// buffer_load_dwordx4 v, voffset, srsrc, soffset offen  -- on gfx950.  Build: hipcc --offload-arch=gfx950 -O2 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *p, float *o, int records, unsigned voff_base, unsigned soff)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, records, 0x00020000);
    const unsigned voff = voff_base + threadIdx.x * 16;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    for (int i = 0; i < 4; ++i) o[threadIdx.x * 4 + i] = __uint_as_float(v[i]);
}
int main()
{
    const int n = 4096;
    float *h = new float[n], *d, *o, ho[256];
    for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    struct { int records; unsigned voff, soff; const char *what; } cases[] = {
        {64 * 4, 0, 0, "records = 64 floats, 64 lanes x 16 B from voffset 0: lanes 0-15 in range"},
        {64 * 4, 0, 1024 * 4, "same, soffset = 1024 floats: is soffset range-checked? (in range if NOT)"},
        {64 * 4 + 8, 0, 0, "records = 66 floats: lane 16's quad is half in range"},
        {64 * 4, 240, 0, "voffset base 240 B: lane 0 quad at floats 60..63 in range, lane 1 out"},
    };
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, c.records, c.voff, c.soff);
        hipMemcpy(ho, o, 256 * 4, hipMemcpyDeviceToHost);
        printf("%s\n  lane0: %g %g %g %g | lane15: %g %g %g %g | lane16: %g %g %g %g | lane17: %g %g %g %g | lane63: %g\n", c.what,
               ho[0], ho[1], ho[2], ho[3], ho[60], ho[61], ho[62], ho[63], ho[64], ho[65], ho[66], ho[67], ho[68], ho[69], ho[70], ho[71], ho[252]);
    }
    return 0;
}
