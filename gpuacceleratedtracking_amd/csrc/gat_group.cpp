// gat_group.cpp -- device groups of the C ABI (include/gat.h gat_group_*, gat_memcpy_peer, gat_device_count).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <dlfcn.h>
#include <new>
#include <string>
#include <vector>

#include "gat_ctx.h"

using namespace gat;

extern "C" {

// ---------------------------------------------------------------------------------------------------------------------
// Device groups: satellite channels sharded over several devices from ONE host thread.
// Every member is an ordinary context with its own stream; nothing below synchronises unless it says so, so the
// per-device launches of one gat_group_correlate overlap.  No collective: the signal is replicated with peer copies
// (xGMI between the GPUs of one node), outputs are disjoint per channel.
// ---------------------------------------------------------------------------------------------------------------------
struct gat_group {
    std::vector<gat_ctx *> ctx;
    std::vector<gat_channel_params> staging; // host copy of one shard's parameters ([K_r x B]); reused per rank
    std::string err;
};

namespace {
int32_t gfail(gat_group *g, int32_t code, const char *msg)
{
    if (g) g->err = msg;
    return code;
}
void shard_bounds(int total, int n, int r, int *lo, int *cnt)
{
    const int base = total / n, extra = total % n;
    *lo = r * base + std::min(r, extra);
    *cnt = base + (r < extra ? 1 : 0);
}
} // namespace

GAT_API int32_t gat_device_count(int32_t *count)
{
    if (!count) return GAT_ERR_ARG;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(int32_t)e;
    *count = n;
    return GAT_OK;
}

namespace {
// one peer copy on dst's stream behind `filled` (an event recorded on the source stream); *copied (optional) receives an
// event recorded behind the copy on dst's stream
hipError_t peer_copy_after(gat_ctx *dst, void *dst_dev, gat_ctx *src, const void *src_dev, size_t bytes, hipEvent_t filled,
                           hipEvent_t *copied)
{
    hipError_t e = hipSetDevice(dst->device);
    if (e == hipSuccess) e = hipStreamWaitEvent(dst->stream, filled, 0);
    if (e == hipSuccess) {
        if (dst->device == src->device)
            e = hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, dst->stream);
        else
            e = hipMemcpyPeerAsync(dst_dev, dst->device, src_dev, src->device, bytes, dst->stream);
    }
    if (e == hipSuccess && copied && dst->stream != src->stream) {
        e = hipEventCreateWithFlags(copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(*copied, dst->stream);
    }
    return e;
}

// dsts[i] <- src for every i, all copies in flight together: ONE "source is complete" event on the source stream, a
// copy on every destination's stream behind it, then the source stream waits for all of them -- whatever it is given
// next (the following block's ingest overwriting src) runs after the copies have read the buffer.  (Waiting per copy
// would chain them: the second destination's "complete" event would sit behind the wait for the first copy.)
int32_t peer_fanout(gat_ctx *err_ctx, gat_ctx *src, const void *src_dev, size_t n, gat_ctx *const *dsts, void *const *dst_devs,
                    size_t bytes)
{
    hipEvent_t filled = nullptr;
    std::vector<hipEvent_t> copied(n, nullptr);
    GAT_HIP(err_ctx, hipSetDevice(src->device));
    GAT_HIP(err_ctx, hipEventCreateWithFlags(&filled, hipEventDisableTiming));
    hipError_t e = hipEventRecord(filled, src->stream);
    src->wait_seq = 0;
    for (size_t i = 0; i < n && e == hipSuccess; ++i) {
        dsts[i]->wait_seq = 0;
        e = peer_copy_after(dsts[i], dst_devs[i], src, src_dev, bytes, filled, &copied[i]);
    }
    if (e == hipSuccess) e = hipSetDevice(src->device);
    for (size_t i = 0; i < n; ++i) {
        if (!copied[i]) continue;
        if (e == hipSuccess) e = hipStreamWaitEvent(src->stream, copied[i], 0);
        (void)hipEventDestroy(copied[i]); // released once the recorded work has completed
    }
    (void)hipEventDestroy(filled);
    if (e != hipSuccess) return hipfail(err_ctx, e, "peer copy");
    return GAT_OK;
}
} // namespace

GAT_API int32_t gat_memcpy_peer(gat_ctx *dst_ctx, void *dst_dev, gat_ctx *src_ctx, const void *src_dev, size_t bytes)
{
    if (!dst_ctx || !src_ctx || !dst_dev || !src_dev) return fail(dst_ctx, GAT_ERR_ARG, "null argument");
    if (bytes == 0) return GAT_OK;
    // Order, both ways: everything enqueued so far on the SOURCE stream (the upload / generator that fills src) completes
    // before the copy, which runs on the DESTINATION stream (its correlator launches follow in stream order); and whatever
    // the source stream is given AFTER this call (the next block's ingest overwriting src) waits for the copy to have read
    // it -- a streaming receiver refills its ingest buffer every millisecond without a group-wide sync in between.
    return peer_fanout(dst_ctx, src_ctx, src_dev, 1, &dst_ctx, &dst_dev, bytes);
}

GAT_API int32_t gat_group_create(int32_t num_members, const int32_t *devices, gat_group **out)
{
    if (!out || num_members < 1 || num_members > 64) return GAT_ERR_ARG;
    *out = nullptr;
    gat_group *g = new (std::nothrow) gat_group();
    if (!g) return GAT_ERR_NOMEM;
    for (int r = 0; r < num_members; ++r) {
        gat_ctx *c = nullptr;
        const int32_t rc = gat_create(devices ? devices[r] : r, GAT_OWN_STREAM, &c);
        if (rc != GAT_OK) {
            for (gat_ctx *x : g->ctx) (void)gat_destroy(x);
            delete g;
            return rc;
        }
        g->ctx.push_back(c);
    }
    // direct peer access between distinct member devices where the platform offers it (xGMI); without it
    // hipMemcpyPeerAsync still works, staged by the runtime
    for (gat_ctx *a : g->ctx)
        for (gat_ctx *b : g->ctx) {
            if (a->device == b->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a->device, b->device) == hipSuccess && can) {
                (void)hipSetDevice(a->device);
                const hipError_t e = hipDeviceEnablePeerAccess(b->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    (void)hipGetLastError();
    *out = g;
    return GAT_OK;
}

GAT_API int32_t gat_group_destroy(gat_group *g)
{
    if (!g) return GAT_ERR_ARG;
    for (gat_ctx *c : g->ctx) (void)gat_destroy(c);
    delete g;
    return GAT_OK;
}

GAT_API int32_t gat_group_size(const gat_group *g, int32_t *num_members)
{
    if (!g || !num_members) return GAT_ERR_ARG;
    *num_members = (int32_t)g->ctx.size();
    return GAT_OK;
}

GAT_API int32_t gat_group_ctx(gat_group *g, int32_t rank, gat_ctx **ctx)
{
    if (!g || !ctx || rank < 0 || rank >= (int32_t)g->ctx.size()) return gfail(g, GAT_ERR_ARG, "rank outside the group");
    *ctx = g->ctx[(size_t)rank];
    return GAT_OK;
}

GAT_API const char *gat_group_last_error(const gat_group *g)
{
    if (!g) return "null group";
    if (!g->err.empty()) return g->err.c_str();
    for (const gat_ctx *c : g->ctx)
        if (!c->err.empty()) return c->err.c_str();
    return "";
}

GAT_API int32_t gat_group_shard(const gat_group *g, int32_t num_channels, int32_t rank, int32_t *first, int32_t *count)
{
    if (!g || !first || !count || num_channels < 0 || rank < 0 || rank >= (int32_t)g->ctx.size()) return GAT_ERR_ARG;
    int lo, cnt;
    shard_bounds(num_channels, (int)g->ctx.size(), rank, &lo, &cnt);
    *first = lo;
    *count = cnt;
    return GAT_OK;
}

GAT_API int32_t gat_group_set_codes(gat_group *g, const int8_t *codes_host, int32_t code_length, int32_t num_prns)
{
    if (!g) return GAT_ERR_ARG;
    for (gat_ctx *c : g->ctx) {
        const int32_t rc = gat_set_codes(c, codes_host, code_length, num_prns);
        if (rc != GAT_OK) return rc;
    }
    return GAT_OK;
}

GAT_API int32_t gat_group_replicate(gat_group *g, int32_t src_rank, void *const *bufs_dev, size_t bytes)
{
    if (!g || !bufs_dev || src_rank < 0 || src_rank >= (int32_t)g->ctx.size()) return gfail(g, GAT_ERR_ARG, "bad argument");
    std::vector<gat_ctx *> dsts;
    std::vector<void *> ptrs;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        if (!bufs_dev[r]) return gfail(g, GAT_ERR_ARG, "null buffer");
        if ((int32_t)r == src_rank || bufs_dev[r] == bufs_dev[src_rank]) continue;
        dsts.push_back(g->ctx[r]);
        ptrs.push_back(bufs_dev[r]);
    }
    if (dsts.empty() || bytes == 0) return GAT_OK;
    gat_ctx *src = g->ctx[(size_t)src_rank];
    // all peers at once (one link per peer on an xGMI node), the source stream ordered behind all of them
    return peer_fanout(src, src, bufs_dev[src_rank], dsts.size(), dsts.data(), ptrs.data(), bytes);
}

GAT_API int32_t gat_group_correlate(gat_group *g, const gat_signal_desc *signals, const gat_channel_params *params_host,
                                    int32_t B, int32_t K, int32_t L, const int32_t *shifts, double fs,
                                    float *const *out_re_dev, float *const *out_im_dev, uint32_t flags)
{
    if (!g || !signals || !params_host || !shifts || !out_re_dev || !out_im_dev) return gfail(g, GAT_ERR_ARG, "null argument");
    if (B < 1 || K < 1) return gfail(g, GAT_ERR_ARG, "sizes must be positive");
    const int n = (int)g->ctx.size();
    for (int r = 0; r < n; ++r) {
        int lo, cnt;
        shard_bounds(K, n, r, &lo, &cnt);
        if (cnt == 0) continue; // fewer channels than members: this one idles
        if (!out_re_dev[r] || !out_im_dev[r]) return gfail(g, GAT_ERR_ARG, "null output buffer");
        // this member's channels of every block, channel fastest: [cnt x B]
        g->staging.resize((size_t)cnt * B);
        for (int b = 0; b < B; ++b)
            std::memcpy(&g->staging[(size_t)b * cnt], &params_host[(size_t)b * K + lo], (size_t)cnt * sizeof(gat_channel_params));
        gat_ctx *c = g->ctx[(size_t)r];
        // the parameter upload is asynchronous from pageable host memory: the runtime copies it out before returning,
        // so the staging vector may be reused for the next member
        const int32_t rc = gat_downconvert_and_correlate(c, &signals[r], g->staging.data(), B, cnt, L, shifts, fs,
                                                         out_re_dev[r], out_im_dev[r], flags);
        if (rc != GAT_OK) return rc;
    }
    return GAT_OK;
}

GAT_API int32_t gat_group_gather(gat_group *g, float *const *out_re_dev, float *const *out_im_dev, int32_t B, int32_t K,
                                 int32_t L, int32_t M, float *host_re, float *host_im)
{
    if (!g || !out_re_dev || !out_im_dev || !host_re || !host_im) return gfail(g, GAT_ERR_ARG, "null argument");
    if (B < 1 || K < 1 || L < 1 || M < 1) return gfail(g, GAT_ERR_ARG, "sizes must be positive");
    const int n = (int)g->ctx.size();
    const size_t lm = (size_t)L * M;
    std::vector<float> tmp;
    for (int r = 0; r < n; ++r) {
        int lo, cnt;
        shard_bounds(K, n, r, &lo, &cnt);
        if (cnt == 0) continue;
        tmp.resize((size_t)B * cnt * lm);
        for (int comp = 0; comp < 2; ++comp) {
            const float *src = comp ? out_im_dev[r] : out_re_dev[r];
            float *dst = comp ? host_im : host_re;
            const int32_t rc = gat_memcpy_d2h(g->ctx[(size_t)r], tmp.data(), src, tmp.size() * sizeof(float)); // synchronises
            if (rc != GAT_OK) return rc;
            for (int b = 0; b < B; ++b)
                std::memcpy(dst + ((size_t)b * K + lo) * lm, tmp.data() + (size_t)b * cnt * lm, (size_t)cnt * lm * sizeof(float));
        }
    }
    return GAT_OK;
}

GAT_API int32_t gat_group_sync(gat_group *g)
{
    if (!g) return GAT_ERR_ARG;
    for (gat_ctx *c : g->ctx) {
        const int32_t rc = gat_sync(c);
        if (rc != GAT_OK) return rc;
    }
    return GAT_OK;
}

} // extern "C"
