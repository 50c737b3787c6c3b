// fake_hip.cpp -- a HOST-ONLY stand-in for the HIP runtime, for one purpose: running libgat's host code (gat_api.cpp: argument
// validation, launch planning, scratch management, graph cache, device groups, the resident correlator's host side) under
// AddressSanitizer / UndefinedBehaviorSanitizer on a machine without a GPU (tests/test_host_sanitizers.py).  "Device" memory
// is host memory (so every copy size the library computes is checked by ASan), streams execute nothing, stream capture
// hands out dummy graphs.  Test infrastructure only: never linked into libgat.so.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "hostsim.h"

struct ihipStream_t {
    std::thread worker; // the emulated resident kernel launched on this stream, if any
    bool capturing = false;
};
struct ihipEvent_t {
    std::chrono::steady_clock::time_point t;
};
struct ihipGraph {
    int dummy = 0;
};
struct hipGraphExec {
    int dummy = 0;
};

namespace hostsim {
Counters counters;
void attach_worker(hipStream_t s, std::thread &&t)
{
    if (s->worker.joinable()) s->worker.join(); // launches of one stream run in order
    s->worker = std::move(t);
}
} // namespace hostsim

extern "C" {

hipError_t hipGetDeviceCount(int *count) { *count = 2; return hipSuccess; }
hipError_t hipSetDevice(int id) { return id >= 0 && id < 2 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int)
{
    std::memset(prop, 0, sizeof *prop);
    std::strcpy(prop->name, "hostsim device");
    std::strcpy(prop->gcnArchName, "gfx950:hostsim");
    prop->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *pi, hipDeviceAttribute_t attr, int)
{
    *pi = attr == hipDeviceAttributeWallClockRate ? 100000 : attr == hipDeviceAttributeIsLargeBar ? 1 : 0;
    return hipSuccess;
}
hipError_t hipRuntimeGetVersion(int *v) { *v = 70200000; return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hostsim error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); ++hostsim::counters.mallocs; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void *p) { std::free(p); if (p) ++hostsim::counters.frees; return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return posix_memalign(p, 64, (n + 63) & ~size_t(63)) == 0 ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }

hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind)
{
    for (size_t r = 0; r < h; ++r) std::memcpy(static_cast<char *>(d) + r * dp, static_cast<const char *>(s) + r * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new ihipStream_t(); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s)
{
    if (s && s->worker.joinable()) s->worker.join();
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (s && s->worker.joinable()) s->worker.join();
    delete s;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st)
{
    *st = s && s->capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode)
{
    if (!s) return hipErrorStreamCaptureUnsupported; // the legacy default stream cannot be captured
    s->capturing = true;
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g)
{
    s->capturing = false;
    *g = new ihipGraph();
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t) { *e = new hipGraphExec(); ++hostsim::counters.graphs; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { ++hostsim::counters.graph_launches; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t *e) { *e = new ihipEvent_t(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new ihipEvent_t(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }

} // extern "C"
