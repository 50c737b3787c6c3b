// gat_api.cpp -- the C ABI declared in include/gat.h: context, validation, the stand-alone operators, timers, options, the
// closed tracking loop with its graph cache.  The launch planner of the correlator is gat_planner.cpp, the device groups
// gat_group.cpp, the resident correlator's host side gat_resident_api.cpp; all four share gat_ctx.h.
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

namespace gat {

// Tracing ranges around the library's launch sequences (the reference wraps every launch of kernel_algorithm in
// NVTX.@range, src/algorithms.jl:953, :973 ... :1526): roctxRangePush / Pop from librocprofiler-sdk-roctx, resolved at
// the first use so that the library has no link-time dependency on the profiler SDK; no-ops when it is not installed.
// Ranges show up in rocprofv3 --marker-trace.  GAT_ROCTX=0 disables the lookup.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = std::getenv("GAT_ROCTX");
        if (e && e[0] == '0') return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
TraceRange::TraceRange(const char *name)
{
    static const Roctx r; // resolved once, thread-safe
    rx = &r;
    if (r.push) (void)r.push(name);
}
TraceRange::~TraceRange()
{
    const Roctx *r = static_cast<const Roctx *>(rx);
    if (r->pop) (void)r->pop();
}

constexpr size_t kMaxLoopGraphs = 4;

// Every instantiated graph bakes in device pointers of the library's own buffers (split partials, code tables) and the
// launch geometry: drop them all whenever one of those can change (scratch reallocation, gat_set_codes, gat_set_stream,
// kernel-selection knobs).
void drop_loop_graphs(gat_ctx *c)
{
    // gat_tracking_run(GAT_FLAG_GRAPH) is asynchronous: a recorded graph may still be executing on the stream
    if (!c->loop_graphs.empty()) (void)hipStreamSynchronize(c->stream);
    for (auto &g : c->loop_graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    c->loop_graphs.clear();
}


int32_t ensure_partial(gat_ctx *c, size_t bytes)
{
    if (bytes <= c->partial_bytes) return GAT_OK;
    drop_loop_graphs(c); // recorded launches point at the old buffer
    if (c->d_partial) {
        GAT_HIP(c, hipStreamSynchronize(c->stream)); // previous launches may still read it
        park_residents(c);
        GAT_HIP(c, hipFree(c->d_partial));
        c->d_partial = nullptr;
        c->partial_bytes = 0;
    }
    GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_partial), bytes));
    c->partial_bytes = bytes;
    return GAT_OK;
}

// the parameter records of a host call reach the device: into the context's buffer, on the context's stream
int32_t upload_params(gat_ctx *c, const gat_channel_params *params_host, size_t n)
{
    if (n > c->params_cap) {
        if (c->d_params) {
            GAT_HIP(c, hipStreamSynchronize(c->stream));
            park_residents(c);
            GAT_HIP(c, hipFree(c->d_params));
            c->d_params = nullptr;
            c->params_cap = 0;
        }
        GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_params), n * sizeof(gat_channel_params)));
        c->params_cap = n;
    }
    GAT_HIP(c, hipMemcpyAsync(c->d_params, params_host, n * sizeof(gat_channel_params), hipMemcpyHostToDevice, c->stream));
    return GAT_OK;
}


} // namespace gat

namespace {

int32_t tracking_run_enqueue(gat_ctx *c, const gat_signal_desc *sig, int32_t num_blocks, int32_t K, int32_t L,
                             const int32_t *shifts, double fs, const gat_loop_config *cfg, gat_loop_state *state,
                             gat_channel_params *params_a, gat_channel_params *params_b, float *acc_re, float *acc_im,
                             int64_t acc_block_stride, uint32_t flags, int32_t *current_is_b)
{
    const size_t sample_bytes = sig->layout == GAT_LAYOUT_PLANAR ? 4 : sig->layout == GAT_LAYOUT_INTERLEAVED ? 8
                              : sig->layout == GAT_LAYOUT_INTERLEAVED_I16 ? 4 : 2;
    gat_channel_params *cur = params_a, *nxt = params_b;
    for (int32_t b = 0; b < num_blocks; ++b) { // everything is enqueued on the ctx stream, nothing synchronises
        gat_signal_desc d = *sig;
        const size_t off = (size_t)b * (size_t)sig->block_stride * sample_bytes;
        d.re = static_cast<const char *>(sig->re) + off;
        if (sig->im) d.im = static_cast<const char *>(sig->im) + off;
        float *o_re = acc_re + (size_t)b * (size_t)acc_block_stride, *o_im = acc_im + (size_t)b * (size_t)acc_block_stride;
        int32_t rc = correlate_impl(c, &d, cur, 1, K, L, shifts, fs, o_re, o_im, flags);
        if (rc != GAT_OK) return rc;
        rc = gat_tracking_update(c, o_re, o_im, K, sig->num_ants, cfg, state, cur, nxt);
        if (rc != GAT_OK) return rc;
        std::swap(cur, nxt);
    }
    if (current_is_b) *current_is_b = cur == params_b ? 1 : 0;
    return GAT_OK;
}

template <typename T>
void key_put(std::vector<unsigned char> &k, const T &v)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&v);
    k.insert(k.end(), p, p + sizeof(T));
}

// GAT_FLAG_GRAPH: replay the launch sequence of `enqueue` as one instantiated hipGraph when the same call repeats.
// Every argument that shapes the sequence is part of the key (make_key), together with the library-owned pointers and
// sizes the recorded launches bake in; a call with a known key replays its graph.  The first call with a key runs
// eagerly (this also sizes the library's scratch buffers, which must not be reallocated inside a capture -- a
// reallocation drops every recorded graph), then the same sequence is recorded for the following calls under the key of
// the buffers the capture really uses.  Up to kMaxLoopGraphs graphs are kept (least recently used goes).
template <class MakeKey, class Enqueue>
int32_t graph_replay_or_record(gat_ctx *c, MakeKey make_key, Enqueue enqueue)
{
    {
        const std::vector<unsigned char> key = make_key();
        for (auto &g : c->loop_graphs)
            if (g.exec && g.key == key) {
                c->wait_seq = 0;
                g.last_use = ++c->loop_graph_clock;
                GAT_HIP(c, hipGraphLaunch(g.exec, c->stream));
                return GAT_OK;
            }
    }
    int32_t rc = enqueue();
    if (rc != GAT_OK) return rc;
    const std::vector<unsigned char> key = make_key();
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return GAT_OK; // no graph (e.g. the legacy default stream): stay eager
    }
    rc = enqueue();
    const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc == GAT_OK && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        if (c->loop_graphs.size() >= kMaxLoopGraphs) { // evict the least recently used
            size_t lru = 0;
            for (size_t i = 1; i < c->loop_graphs.size(); ++i)
                if (c->loop_graphs[i].last_use < c->loop_graphs[lru].last_use) lru = i;
            if (c->loop_graphs[lru].exec) {
                (void)hipStreamSynchronize(c->stream); // its last replay may still be running
                (void)hipGraphExecDestroy(c->loop_graphs[lru].exec);
            }
            c->loop_graphs.erase(c->loop_graphs.begin() + (long)lru);
        }
        gat_ctx::LoopGraph g;
        g.key = key;
        g.exec = exec;
        g.last_use = ++c->loop_graph_clock;
        c->loop_graphs.push_back(std::move(g));
    }
    if (graph) (void)hipGraphDestroy(graph);
    return GAT_OK;
}

// the part of a graph key every recorded launch sequence shares: kernel-selection knobs and library-owned buffers
void key_put_ctx(std::vector<unsigned char> &key, const gat_ctx *c)
{
    key_put(key, c->mc_mode); key_put(key, c->mc_i16_terms); key_put(key, c->mc_nct); key_put(key, c->max_aw); key_put(key, c->max_kt); key_put(key, c->max_bpw); key_put(key, c->force_bpw);
    key_put(key, c->wgs_per_cu); key_put(key, c->max_depth); key_put(key, c->one_wave); key_put(key, c->keep_l2); key_put(key, c->align_head);
    key_put(key, c->aw2); key_put(key, c->quads); key_put(key, c->bit_tables); key_put(key, c->seg_cap);
    key_put(key, c->d_codes); key_put(key, c->d_code_bits); key_put(key, c->Lc); key_put(key, c->P);
    key_put(key, c->d_partial); key_put(key, c->partial_bytes);
}

} // namespace

namespace {

// gat_set_option: launch-geometry options (tests force a code path on a small case with them, A/B measurements compare
// geometries).  None changes a result beyond summation order.
struct OptionDesc {
    const char *name;
    long long lo, hi;
};
constexpr OptionDesc kOptions[] = {
    {"sync_flag_wgs", 0, 1 << 20},   // largest launch (workgroups) that carries the completion flag; 0: never
    {"max_ant_tile", 1, kMaxAntTile}, // antennas per wave
    {"dc_aw", 1, 4},                 // antenna tiles (waves) per workgroup, cap
    {"dc_kt", 1, 4},                 // channels per workgroup, cap
    {"dc_bpw", 1, 1 << 20},          // consecutive blocks per workgroup, cap
    {"dc_bpw_force", 0, 1 << 20},    // blocks per workgroup whatever the planner's rule says (0: planner)
    {"dc_wgs_per_cu", 0, 1024},      // workgroups per CU the split planner aims for (0: by instance)
    {"dc_one_wave", 0, 1},           // one-wave workgroups allowed
    {"dc_one_wave_min", -1, 1ll << 40}, // fewest (block, channel, tile) groups for them (-1: 32 per CU)
    {"dc_ow_seg", 1, kUcarSteps},    // steps per replica segment of a one-wave workgroup
    {"dc_depth", 1, 2},              // cap of the sample prefetch depth (register sets per wave)
    {"dc_keep_l2", -1, 1},           // sample loads: -1 by rule (plain when channel groups share a tile through L2), 0 non-temporal, 1 plain
    {"dc_quads", -1, 1},             // replica fill four entries at a time: -1 by rule (the two-channel 2 x 2 tile), 0 never, 1 wherever the code rate allows
    {"dc_bits", 0, 2},               // chip tables in LDS as sign bits: 0 never, 1 long codes (> 2 KB per PRN), 2 whenever every chip is +-1
    {"dc_aw2", -1, 1},               // the two-channel 2 x 2 tile (two waves of two antennas, two channels each): -1 by rule, 0 never, 1 wherever an instance exists
    {"dc_seg", 0, kUcarSteps},       // cap of the steps per replica segment of a four-wave workgroup (0: by instance)
    {"mc_nct", 0, 4},                // split-bf16 kernel: 32-column channel tiles per workgroup to try first (1, 2, 4; 0 = by rule)
    {"mc_i16_terms", 2, 3},          // split-bf16 kernel, int16 samples: 2 = the exact two-term split (5 products per sample), 3 = the float path's three terms (8)
    {"dc_align", 0, 1},              // blocks walked from the 128-byte line their first sample lies in (1) or from the sample itself (0)
};

int32_t set_option(gat_ctx *c, const char *name, long long v)
{
    const OptionDesc *o = nullptr;
    for (const auto &d : kOptions)
        if (std::strcmp(d.name, name) == 0) o = &d;
    if (!o) return fail(c, GAT_ERR_ARG, "unknown option");
    if (v < o->lo || v > o->hi) return fail(c, GAT_ERR_RANGE, "option value out of range");
    const std::string n = name;
    if (n == "sync_flag_wgs") c->flag_max_wgs = (int)v;
    else if (n == "max_ant_tile") c->max_ant_tile = (int)v;
    else if (n == "dc_aw") c->max_aw = (int)v;
    else if (n == "dc_kt") c->max_kt = (int)v;
    else if (n == "dc_bpw") c->max_bpw = (int)v;
    else if (n == "dc_bpw_force") c->force_bpw = (int)v;
    else if (n == "dc_wgs_per_cu") c->wgs_per_cu = (int)v;
    else if (n == "dc_one_wave") c->one_wave = (int)v;
    else if (n == "dc_one_wave_min") c->one_wave_min = v;
    else if (n == "dc_ow_seg") c->one_wave_seg = (int)v;
    else if (n == "dc_depth") c->max_depth = (int)v;
    else if (n == "dc_keep_l2") c->keep_l2 = (int)v;
    else if (n == "dc_quads") c->quads = (int)v;
    else if (n == "dc_bits") c->bit_tables = (int)v;
    else if (n == "dc_aw2") c->aw2 = (int)v;
    else if (n == "dc_seg") c->seg_cap = (int)v;
    else if (n == "dc_align") c->align_head = (int)v;
    else if (n == "mc_i16_terms") c->mc_i16_terms = (int)v;
    else if (n == "mc_nct") c->mc_nct = v == 3 ? 2 : (int)v;
    return GAT_OK;
}

} // namespace

extern "C" {

GAT_API int32_t gat_create(int32_t device, void *hip_stream, gat_ctx **out_ctx)
{
    if (!out_ctx) return GAT_ERR_ARG;
    *out_ctx = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return -(int32_t)e;
    if (device < 0 || device >= ndev) return GAT_ERR_RANGE;
    gat_ctx *c = new (std::nothrow) gat_ctx();
    if (!c) return GAT_ERR_NOMEM;
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        delete c;
        return -(int32_t)e;
    }
    if (hip_stream != GAT_OWN_STREAM) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream); // NULL == the default stream
    } else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
            delete c;
            return -(int32_t)e;
        }
        c->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if (hipMalloc(&c->d_zeros, 64) == hipSuccess) (void)hipMemset(c->d_zeros, 0, 64);
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_flag), 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
        *c->h_flag = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_flag), c->h_flag, 0) != hipSuccess) c->d_flag = nullptr;
        if (c->d_flag && hipMalloc(reinterpret_cast<void **>(&c->d_done), 64) == hipSuccess) (void)hipMemset(c->d_done, 0, 64);
        else c->d_flag = nullptr;
    }
    (void)hipGetLastError();
#ifdef GAT_DEV
    // development builds only (-DGAT_DEV; gat_version() names the flag): the launch-geometry options of gat_set_option
    // taken from the environment, so that scripts can A/B them without code changes.  The product library reads no
    // environment variable that selects kernels or geometry.
    for (const auto &o : kOptions) {
        std::string env = "GAT_";
        for (const char *p = o.name; *p; ++p) env += (char)std::toupper((unsigned char)*p);
        if (const char *v = std::getenv(env.c_str())) (void)set_option(c, o.name, std::atoll(v));
    }
    if (const char *v = std::getenv("GAT_NO_MFMA")) c->mc_mode = v[0] == '1' ? 0 : 1;
    if (const char *v = std::getenv("GAT_MC_MODE")) c->mc_mode = (v[0] >= '0' && v[0] <= '3') ? v[0] - '0' : 1;
#endif
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        delete c;
        return -(int32_t)e;
    }
    *out_ctx = c;
    return GAT_OK;
}

GAT_API int32_t gat_destroy(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    (void)hipSetDevice(c->device);
    park_residents(c);
    for (gat_resident *r : c->residents) resident_free(r); // handles of correlators that were not closed die with the context
    c->residents.clear();
    (void)hipStreamSynchronize(c->stream);
    if (c->d_codes) (void)hipFree(c->d_codes);
    if (c->d_code_bits) (void)hipFree(c->d_code_bits);
    if (c->d_zeros) (void)hipFree(c->d_zeros);
    drop_loop_graphs(c);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_done) (void)hipFree(c->d_done);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->d_params) (void)hipFree(c->d_params);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->lap_events) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GAT_OK;
}

GAT_API int32_t gat_set_stream(gat_ctx *c, void *hip_stream)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    drop_loop_graphs(c);
    if (c->own_stream) {
        GAT_HIP(c, hipStreamDestroy(c->stream));
        c->own_stream = false;
    }
    if (hip_stream != GAT_OWN_STREAM) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        GAT_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return GAT_OK;
}

GAT_API int32_t gat_sync(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    if (c->wait_seq && c->h_flag) {
        // the newest work on the stream is a flagged correlator launch: its last workgroup stores wait_seq into pinned
        // host memory after its results are out (system-scope release).  Launches of one stream finish in order, so
        // the flag reaches wait_seq exactly when everything enqueued is done.  Bounded spin, then the ordinary wait.
        const unsigned want = c->wait_seq;
        c->wait_seq = 0;
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == want) return GAT_OK;
            if ((spins & 255u) == 255u) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 200000) break; // 200 us
            }
        }
    }
    GAT_HIP(c, hipSetDevice(c->device));
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    return GAT_OK;
}

GAT_API const char *gat_last_error(const gat_ctx *c) { return c ? c->err.c_str() : "null context"; }

GAT_API int32_t gat_device_info(gat_ctx *c, char *name_buf, size_t name_len, int32_t *runtime_version,
                                int32_t *num_cus)
{
    if (!c) return GAT_ERR_ARG;
    hipDeviceProp_t prop;
    GAT_HIP(c, hipGetDeviceProperties(&prop, c->device));
    if (name_buf && name_len) {
        std::snprintf(name_buf, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (runtime_version) {
        int v = 0;
        GAT_HIP(c, hipRuntimeGetVersion(&v));
        *runtime_version = v;
    }
    if (num_cus) *num_cus = prop.multiProcessorCount;
    return GAT_OK;
}

GAT_API int32_t gat_set_codes(gat_ctx *c, const int8_t *codes_host, int32_t code_length, int32_t num_prns)
{
    if (!c || !codes_host) return fail(c, GAT_ERR_ARG, "null argument");
    if (code_length < 1 || num_prns < 1) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    // the vector kernel keeps a workgroup's chip table in LDS next to one replica segment (~40 KB): 160 KB - that
    if (code_length > 120000) return fail(c, GAT_ERR_RANGE, "code table does not fit in LDS (max 120000 chips)");
    GAT_HIP(c, hipSetDevice(c->device));
    park_residents(c); // their kernels hold the old tables (and hipFree waits for the whole device)
    for (gat_resident *r : c->residents) r->stale = true;
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    drop_loop_graphs(c); // recorded launches point at the old tables
    if (c->d_codes) {
        GAT_HIP(c, hipFree(c->d_codes));
        c->d_codes = nullptr;
    }
    if (c->d_code_bits) {
        GAT_HIP(c, hipFree(c->d_code_bits));
        c->d_code_bits = nullptr;
    }
    // 16-byte rows (dc_kernel stages them with 16-byte copies) with room for one more chip: every row carries its FIRST chip
    // again at index code_length, so that "this chip and the next" are two reads without a wrap (the replica fill by quads)
    const int stride = (code_length + 16) & ~15;
    const size_t bytes = (size_t)stride * num_prns;
    {
        std::vector<int8_t> rows(bytes, 0);
        for (int p = 0; p < num_prns; ++p) {
            std::memcpy(&rows[(size_t)p * stride], codes_host + (size_t)p * code_length, (size_t)code_length);
            rows[(size_t)p * stride + code_length] = codes_host[(size_t)p * code_length];
        }
        GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_codes), bytes));
        GAT_HIP(c, hipMemcpy(c->d_codes, rows.data(), bytes, hipMemcpyHostToDevice));
    }
    c->code_row_stride = stride;
    c->Lc = code_length;
    c->P = num_prns;
    // sign-bit tables for the split-bf16 matrix kernel (a chip only flips signs there): 1/8 of the LDS
    bool pm1 = true;
    for (size_t i = 0; i < (size_t)code_length * num_prns && pm1; ++i) pm1 = codes_host[i] == 1 || codes_host[i] == -1;
    if (pm1) {
        // (rows: chip code_length repeats chip 0 -- see above --, and one whole dword of slack behind the last chip, so that the
        // two dwords holding "this chip and the next" can always be read together)
        const int bstride = (((code_length + 1 + 32 + 31) / 32) + 3) & ~3;
        std::vector<uint32_t> bits((size_t)bstride * num_prns, 0u);
        for (int p = 0; p < num_prns; ++p)
            for (int i = 0; i < code_length; ++i)
                if (codes_host[(size_t)p * code_length + i] < 0) bits[(size_t)p * bstride + (i >> 5)] |= 1u << (i & 31);
        for (int p = 0; p < num_prns; ++p)
            if (codes_host[(size_t)p * code_length] < 0) bits[(size_t)p * bstride + (code_length >> 5)] |= 1u << (code_length & 31);
        GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_code_bits), bits.size() * sizeof(uint32_t)));
        GAT_HIP(c, hipMemcpy(c->d_code_bits, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->code_bits_stride = bstride;
    }
    return GAT_OK;
}

GAT_API int32_t gat_downconvert_and_correlate_dev(gat_ctx *c, const gat_signal_desc *sig,
                                                  const gat_channel_params *params_dev, int32_t B,
                                                  int32_t K, int32_t L, const int32_t *shifts, double fs,
                                                  float *out_re, float *out_im, uint32_t flags)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    if (!(flags & GAT_FLAG_GRAPH)) return correlate_impl(c, sig, params_dev, B, K, L, shifts, fs, out_re, out_im, flags);
    // a receiver that calls the operator block after block on the same buffers: the one to three launches of a call
    // (tap groups, second stage) replayed as one instantiated graph
    if (!sig || !shifts || L < 1 || L > GAT_MAX_TAPS) return fail(c, GAT_ERR_ARG, "bad argument");
    const uint32_t kflags = flags & ~GAT_FLAG_GRAPH;
    auto make_key = [&]() {
        std::vector<unsigned char> key;
        key_put(key, (int)2 /* sequence: one correlate call */);
        key_put(key, sig->re); key_put(key, sig->im); key_put(key, sig->layout); key_put(key, sig->num_ants);
        key_put(key, sig->num_samples); key_put(key, sig->ant_stride); key_put(key, sig->block_stride);
        key_put(key, sig->chan_stride); key_put(key, params_dev); key_put(key, B); key_put(key, K); key_put(key, L);
        key_put(key, fs); key_put(key, out_re); key_put(key, out_im); key_put(key, kflags);
        for (int l = 0; l < L; ++l) key_put(key, shifts[l]);
        key_put_ctx(key, c);
        return key;
    };
    return graph_replay_or_record(c, make_key, [&]() { return correlate_impl(c, sig, params_dev, B, K, L, shifts, fs, out_re, out_im, kflags); });
}

GAT_API int32_t gat_downconvert_and_correlate(gat_ctx *c, const gat_signal_desc *sig,
                                              const gat_channel_params *params_host, int32_t B, int32_t K,
                                              int32_t L, const int32_t *shifts, double fs, float *out_re,
                                              float *out_im, uint32_t flags)
{
    if (!c) return GAT_ERR_ARG;
    if (!params_host || !sig || !shifts) return fail(c, GAT_ERR_ARG, "null argument");
    if (B < 1 || K < 1) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    GAT_HIP(c, hipSetDevice(c->device));
    // host-side validation the device-params variant cannot do
    long long max_shift = 0;
    for (int l = 0; l < L && l < GAT_MAX_TAPS; ++l)
        max_shift = std::max<long long>(max_shift, std::llabs((long long)shifts[l]));
    const size_t n = (size_t)B * K;
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (!(fs > 0.0) || !std::isfinite(fs)) return fail(c, GAT_ERR_ARG, "sampling frequency must be positive");
    {
        const int32_t rcv = validate_params(c, params_host, n, (double)(sig->num_samples + max_shift), fs);
        if (rcv != GAT_OK) return rcv;
    }
    if (n <= (size_t)kInlineParams) // no upload: the records ride in the kernel arguments
        return correlate_impl(c, sig, nullptr, B, K, L, shifts, fs, out_re, out_im, flags, params_host);
    const int32_t rc = upload_params(c, params_host, n);
    if (rc != GAT_OK) return rc;
    return correlate_impl(c, sig, c->d_params, B, K, L, shifts, fs, out_re, out_im, flags);
}

static int32_t gen_code_replica_impl(gat_ctx *c, float *rep, int64_t count, int32_t prn, double fc, double fs,
                                     double tau, int64_t first_shift, bool f32_coordinates)
{
    if (!c || !rep) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (count < 1) return fail(c, GAT_ERR_ARG, "count must be positive");
    if (prn < 0 || prn >= c->P) return fail(c, GAT_ERR_RANGE, "prn outside the code table");
    if (!(fs > 0.0) || !std::isfinite(fc) || !std::isfinite(tau)) return fail(c, GAT_ERR_ARG, "bad frequency / phase");
    if (count + std::llabs((long long)first_shift) >= (1ll << 30)) return fail(c, GAT_ERR_RANGE, "replica too long");
    if (!f32_coordinates && !code_span_ok(fc / fs, tau, (double)count + (double)std::llabs((long long)first_shift), c->Lc))
        return fail(c, GAT_ERR_RANGE, "code phase span too large");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_gen_code_replica");
    GAT_HIP(c, launch_gen_code_replica(rep, count, c->d_codes + (size_t)prn * c->code_row_stride, c->Lc, fc, fs, tau,
                                       first_shift, f32_coordinates, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_gen_code_replica(gat_ctx *c, float *rep, int64_t count, int32_t prn, double fc,
                                     double fs, double tau, int64_t first_shift)
{
    return gen_code_replica_impl(c, rep, count, prn, fc, fs, tau, first_shift, false);
}

GAT_API int32_t gat_gen_code_replica_f32coord(gat_ctx *c, float *rep, int64_t count, int32_t prn, double fc,
                                              double fs, double tau, int64_t first_shift)
{
    return gen_code_replica_impl(c, rep, count, prn, fc, fs, tau, first_shift, true);
}

GAT_API int32_t gat_gen_code_replica_texaddr(gat_ctx *c, float *rep, int64_t count, int32_t prn, double fc, double fs, double tau,
                                             int64_t first_shift, int32_t coord_frac_bits, int32_t texel_frac_bits)
{
    if (!c || !rep) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (count < 1) return fail(c, GAT_ERR_ARG, "count must be positive");
    if (prn < 0 || prn >= c->P) return fail(c, GAT_ERR_RANGE, "prn outside the code table");
    if (!(fs > 0.0) || !std::isfinite(fc) || !std::isfinite(tau)) return fail(c, GAT_ERR_ARG, "bad frequency / phase");
    if (count + std::llabs((long long)first_shift) >= (1ll << 30)) return fail(c, GAT_ERR_RANGE, "replica too long");
    if (coord_frac_bits < 0 || coord_frac_bits > 32 || texel_frac_bits < -1 || texel_frac_bits > 24)
        return fail(c, GAT_ERR_RANGE, "coord_frac_bits 0 .. 32, texel_frac_bits -1 .. 24");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_gen_code_replica_texaddr");
    GAT_HIP(c, launch_gen_code_replica_texaddr(rep, count, c->d_codes + (size_t)prn * c->code_row_stride, c->Lc, fc, fs, tau,
                                               first_shift, coord_frac_bits, texel_frac_bits, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_gen_code_replica_multi(gat_ctx *c, float *rep, int64_t count, int64_t row_stride, int32_t K,
                                           const gat_channel_params *params_dev, double fs, int64_t first_shift)
{
    if (!c || !rep || !params_dev) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (count < 1 || K < 1 || K > 65535 || row_stride < count) return fail(c, GAT_ERR_ARG, "bad sizes");
    if (!(fs > 0.0) || count + std::llabs((long long)first_shift) >= (1ll << 30)) return fail(c, GAT_ERR_RANGE, "replica too long");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_gen_code_replica_multi");
    GAT_HIP(c, launch_gen_code_replica_multi(rep, count, row_stride, K, params_dev, c->d_codes, c->code_row_stride, c->Lc,
                                             c->P, fs, first_shift, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_downconvert_and_accumulate(gat_ctx *c, const gat_signal_desc *sig, const gat_channel_params *p,
                                               int32_t L, const int32_t *shifts, double fs, float *car_re, float *car_im,
                                               float *dw_re, float *dw_im, float *acc_re, float *acc_im)
{
    if (!c || !sig || !p || !shifts) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (sig->layout != GAT_LAYOUT_PLANAR || !sig->re || !sig->im) return fail(c, GAT_ERR_UNSUPPORTED, "planar float signal only");
    if (L < 1 || L > GAT_MAX_TAPS || sig->num_ants < 1 || sig->num_samples < 1 || sig->num_samples >= (1ll << 30))
        return fail(c, GAT_ERR_RANGE, "size out of range");
    if (p->prn < 0 || p->prn >= c->P) return fail(c, GAT_ERR_RANGE, "prn outside the code table");
    if (!(fs > 0.0) || !std::isfinite(p->code_freq_hz) || !(p->code_freq_hz >= 0.0) || !std::isfinite(p->carrier_freq_hz) ||
        !std::isfinite(p->code_phase_chips) || !std::isfinite(p->carrier_phase_cycles))
        return fail(c, GAT_ERR_ARG, "bad frequency / phase");
    long long max_shift = 0;
    for (int l = 0; l < L; ++l) max_shift = std::max<long long>(max_shift, std::llabs((long long)shifts[l]));
    if (sig->num_samples + max_shift >= (1ll << 30)) return fail(c, GAT_ERR_RANGE, "num_samples + |shift| must stay below 2^30");
    if (!code_span_ok(p->code_freq_hz / fs, p->code_phase_chips, (double)(sig->num_samples + max_shift), c->Lc))
        return fail(c, GAT_ERR_RANGE, "code phase span too large");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_downconvert_and_accumulate");
    // the tap list goes through the library's parameter scratch (device memory the kernel can read)
    const size_t need = ((size_t)L * sizeof(int32_t) + sizeof(gat_channel_params) - 1) / sizeof(gat_channel_params);
    if (need > c->params_cap) {
        if (c->d_params) {
            GAT_HIP(c, hipStreamSynchronize(c->stream));
            park_residents(c);
            GAT_HIP(c, hipFree(c->d_params));
            c->d_params = nullptr;
            c->params_cap = 0;
        }
        GAT_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_params), need * sizeof(gat_channel_params)));
        c->params_cap = need;
    }
    GAT_HIP(c, hipMemcpyAsync(c->d_params, shifts, (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    GAT_HIP(c, launch_accumulate_debug(static_cast<const float *>(sig->re), static_cast<const float *>(sig->im),
                                       sig->num_samples, sig->num_ants, sig->ant_stride, *p,
                                       c->d_codes + (size_t)p->prn * c->code_row_stride, c->Lc, fs, L,
                                       reinterpret_cast<const int *>(c->d_params), car_re, car_im, dw_re, dw_im, acc_re, acc_im,
                                       c->stream));
    return GAT_OK;
}

static int32_t gen_signal_impl(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M, int64_t ant_stride,
                               int64_t block_stride, int32_t B, int32_t K, const gat_channel_params *params_dev, double fs,
                               double amplitude, const float *steering_cycles_dev, double noise_sigma, uint64_t seed);

GAT_API int32_t gat_gen_signal(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M,
                               int64_t ant_stride, int64_t block_stride, int32_t B, int32_t K,
                               const gat_channel_params *params_dev, double fs, double amplitude)
{
    return gen_signal_impl(c, re, im, layout, N, M, ant_stride, block_stride, B, K, params_dev, fs, amplitude, nullptr, 0.0, 0);
}

GAT_API int32_t gat_gen_signal_noisy(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M,
                                     int64_t ant_stride, int64_t block_stride, int32_t B, int32_t K,
                                     const gat_channel_params *params_dev, double fs, double amplitude,
                                     const float *steering_cycles_dev, double noise_sigma, uint64_t seed)
{
    if (c && (!(noise_sigma >= 0.0) || !std::isfinite(noise_sigma))) return fail(c, GAT_ERR_ARG, "noise sigma must be finite and >= 0");
    return gen_signal_impl(c, re, im, layout, N, M, ant_stride, block_stride, B, K, params_dev, fs, amplitude,
                           steering_cycles_dev, noise_sigma, seed);
}

static int32_t gen_signal_impl(gat_ctx *c, void *re, void *im, int32_t layout, int64_t N, int32_t M, int64_t ant_stride,
                               int64_t block_stride, int32_t B, int32_t K, const gat_channel_params *params_dev, double fs,
                               double amplitude, const float *steering_cycles_dev, double noise_sigma, uint64_t seed)
{
    if (!c || !re || !params_dev) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->d_codes) return fail(c, GAT_ERR_STATE, "gat_set_codes has not been called");
    if (layout < GAT_LAYOUT_PLANAR || layout > GAT_LAYOUT_INTERLEAVED_I8) return fail(c, GAT_ERR_ARG, "unknown layout");
    if ((layout == GAT_LAYOUT_PLANAR) != (im != nullptr)) return fail(c, GAT_ERR_ARG, "signal pointers do not match the layout");
    if (N < 1 || N >= (1ll << 30) || M < 1 || B < 1 || B > 65535 || K < 1) return fail(c, GAT_ERR_RANGE, "size out of range");
    if (!(fs > 0.0) || !std::isfinite(amplitude)) return fail(c, GAT_ERR_ARG, "bad sampling frequency / amplitude");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_gen_signal");
    GAT_HIP(c, launch_gen_signal(re, im, layout, N, M, ant_stride, block_stride, B, K, params_dev, c->d_codes,
                                 c->code_row_stride, c->Lc, c->P, fs, (float)amplitude, steering_cycles_dev, (float)noise_sigma,
                                 (unsigned long long)seed, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_reduce_cplx_multi(gat_ctx *c, const float *in_re, const float *in_im, int64_t n,
                                      int32_t cols, float *out_re, float *out_im)
{
    if (!c || !in_re || !in_im || !out_re || !out_im) return fail(c, GAT_ERR_ARG, "null argument");
    if (n < 1 || cols < 1 || cols > 65535) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_reduce_cplx_multi");
    long long chunks = (n + 4 * kThreads - 1) / (4 * kThreads);
    const long long want = std::max<long long>(1, (4ll * c->num_cus + cols - 1) / cols);
    chunks = std::max<long long>(1, std::min(chunks, want));
    const int32_t rc = ensure_partial(c, (size_t)chunks * cols * 2 * sizeof(float));
    if (rc != GAT_OK) return rc;
    GAT_HIP(c, launch_reduce_stage1(in_re, in_im, n, cols, (int)chunks, c->d_partial, c->stream));
    GAT_HIP(c, launch_finalize(c->d_partial, out_re, out_im, (int)chunks, cols * 2, 1, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_tracking_update(gat_ctx *c, const float *acc_re, const float *acc_im, int32_t K, int32_t M,
                                    const gat_loop_config *cfg, gat_loop_state *state,
                                    const gat_channel_params *cur, gat_channel_params *next)
{
    if (!c || !acc_re || !acc_im || !cfg || !state || !cur || !next) return fail(c, GAT_ERR_ARG, "null argument");
    if (K < 1 || M < 1) return fail(c, GAT_ERR_ARG, "sizes must be positive");
    const int L = cfg->num_taps;
    if (L < 1 || L > GAT_MAX_TAPS || cfg->early_index < 0 || cfg->early_index >= L || cfg->prompt_index < 0 ||
        cfg->prompt_index >= L || cfg->late_index < 0 || cfg->late_index >= L)
        return fail(c, GAT_ERR_RANGE, "tap indices outside the tap list");
    if (!(cfg->block_seconds > 0.0) || !(cfg->pll_bandwidth_hz >= 0.0) || !(cfg->dll_bandwidth_hz >= 0.0) ||
        !(cfg->code_freq_nominal_hz > 0.0) || !(cfg->carrier_center_hz > 0.0) || cfg->code_length < 1 ||
        !(cfg->early_late_spacing_chips > 0.0 && cfg->early_late_spacing_chips < 2.0))
        return fail(c, GAT_ERR_ARG, "bad loop configuration");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    const TraceRange trace("gat_tracking_update");
    GAT_HIP(c, launch_tracking_update(acc_re, acc_im, K, M, *cfg, state, cur, next, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_tracking_run(gat_ctx *c, const gat_signal_desc *sig, int32_t num_blocks, int32_t K, int32_t L,
                                 const int32_t *shifts, double fs, const gat_loop_config *cfg, gat_loop_state *state,
                                 gat_channel_params *params_a, gat_channel_params *params_b, float *acc_re,
                                 float *acc_im, int64_t acc_block_stride, uint32_t flags, int32_t *current_is_b)
{
    if (!c || !sig || !shifts || !cfg || !state || !params_a || !params_b || !acc_re || !acc_im)
        return fail(c, GAT_ERR_ARG, "null argument");
    if (num_blocks < 1 || acc_block_stride < 0) return fail(c, GAT_ERR_ARG, "bad block count / stride");
    if (cfg->num_taps != L || L < 1 || L > GAT_MAX_TAPS) return fail(c, GAT_ERR_ARG, "loop configuration and tap list disagree");
    if (flags & ~(GAT_FLAG_ATOMIC | GAT_FLAG_GRAPH)) return fail(c, GAT_ERR_ARG, "unknown flag bits");
    GAT_HIP(c, hipSetDevice(c->device));
    const TraceRange trace("gat_tracking_run");
    const uint32_t kflags = flags & ~GAT_FLAG_GRAPH;
    if (!(flags & GAT_FLAG_GRAPH))
        return tracking_run_enqueue(c, sig, num_blocks, K, L, shifts, fs, cfg, state, params_a, params_b, acc_re, acc_im,
                                    acc_block_stride, kflags, current_is_b);

    // hipGraph path: the 2-3 launches per block are too short to hide their launch gaps.  Every argument that
    // shapes the launch sequence is part of the key -- field by field (struct padding of a C caller is not
    // initialised) --, together with the library-owned pointers and sizes the recorded launches bake in; a call with
    // a known key replays its instantiated graph.  Up to kMaxLoopGraphs graphs are kept (least recently used goes).
    auto make_key = [&]() {
        std::vector<unsigned char> key;
        key_put(key, sig->re); key_put(key, sig->im); key_put(key, sig->layout); key_put(key, sig->num_ants);
        key_put(key, sig->num_samples); key_put(key, sig->ant_stride); key_put(key, sig->block_stride);
        key_put(key, sig->chan_stride); key_put(key, num_blocks); key_put(key, K); key_put(key, L); key_put(key, fs);
        key_put(key, cfg->block_seconds); key_put(key, cfg->pll_bandwidth_hz); key_put(key, cfg->dll_bandwidth_hz);
        key_put(key, cfg->code_freq_nominal_hz); key_put(key, cfg->carrier_center_hz); key_put(key, cfg->if_hz);
        key_put(key, cfg->early_late_spacing_chips); key_put(key, cfg->code_length); key_put(key, cfg->num_taps);
        key_put(key, cfg->early_index); key_put(key, cfg->prompt_index); key_put(key, cfg->late_index);
        key_put(key, state); key_put(key, params_a); key_put(key, params_b); key_put(key, acc_re); key_put(key, acc_im);
        key_put(key, acc_block_stride); key_put(key, kflags); key_put(key, (int)1 /* sequence: tracking run */);
        key_put_ctx(key, c);
        for (int l = 0; l < L; ++l) key_put(key, shifts[l]);
        return key;
    };
    if (current_is_b) *current_is_b = (num_blocks & 1) ? 1 : 0; // the buffers swap once per block
    return graph_replay_or_record(c, make_key, [&]() {
        return tracking_run_enqueue(c, sig, num_blocks, K, L, shifts, fs, cfg, state, params_a, params_b, acc_re, acc_im,
                                    acc_block_stride, kflags, nullptr);
    });
}

GAT_API int32_t gat_malloc(gat_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || bytes == 0) return fail(c, GAT_ERR_ARG, "bad argument");
    GAT_HIP(c, hipSetDevice(c->device));
    GAT_HIP(c, hipMalloc(out, bytes));
    return GAT_OK;
}

GAT_API int32_t gat_free(gat_ctx *c, void *p)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    park_residents(c); // hipFree waits for every kernel on the device: a resident one would hold it until its idle limit
    GAT_HIP(c, hipFree(p));
    return GAT_OK;
}

GAT_API int32_t gat_memcpy_h2d(gat_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(c, GAT_ERR_ARG, "null argument");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_memcpy_d2h(gat_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(c, GAT_ERR_ARG, "null argument");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_memset(gat_ctx *c, void *dst, int32_t value, size_t bytes)
{
    if (!c || !dst) return fail(c, GAT_ERR_ARG, "null argument");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return GAT_OK;
}

GAT_API int32_t gat_timer_start(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipEventRecord(c->ev0, c->stream));
    c->timer_running = true;
    return GAT_OK;
}

GAT_API int32_t gat_timer_stop(gat_ctx *c, float *ms)
{
    if (!c || !ms) return fail(c, GAT_ERR_ARG, "null argument");
    if (!c->timer_running) return fail(c, GAT_ERR_STATE, "timer not started");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipEventRecord(c->ev1, c->stream));
    GAT_HIP(c, hipEventSynchronize(c->ev1));
    GAT_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    c->timer_running = false;
    return GAT_OK;
}

// per-call statistics (src/benchmarks.jl:1-9 keeps every sample's time): one event per lap from a pool of the context's
GAT_API int32_t gat_timer_lap(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    if (c->laps == c->lap_events.size()) {
        if (c->laps >= (size_t)1 << 20) return fail(c, GAT_ERR_RANGE, "too many laps outstanding: call gat_timer_laps");
        hipEvent_t e = nullptr;
        GAT_HIP(c, hipEventCreate(&e));
        c->lap_events.push_back(e);
    }
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    GAT_HIP(c, hipEventRecord(c->lap_events[c->laps], c->stream));
    ++c->laps;
    return GAT_OK;
}

GAT_API int32_t gat_timer_laps(gat_ctx *c, float *ms, int32_t capacity, int32_t *num)
{
    if (!c || !num || capacity < 0 || (capacity > 0 && !ms)) return fail(c, GAT_ERR_ARG, "null argument or negative capacity");
    *num = 0;
    const size_t laps = c->laps;
    c->laps = 0; // forgotten whatever happens below
    if (laps == 0) return GAT_OK;
    GAT_HIP(c, hipSetDevice(c->device));
    GAT_HIP(c, hipEventSynchronize(c->lap_events[laps - 1]));
    const size_t n = std::min<size_t>(laps - 1, (size_t)capacity);
    for (size_t i = 0; i < n; ++i) GAT_HIP(c, hipEventElapsedTime(&ms[i], c->lap_events[i], c->lap_events[i + 1]));
    *num = (int32_t)n;
    return GAT_OK;
}

// a kernel that only reads (the measured read ceiling beside the spec peak)
GAT_API int32_t gat_debug_read_stream(gat_ctx *c, const void *dev, size_t bytes, int32_t variant, int32_t launches, float *ms_each)
{
    if (!c || !dev || !ms_each) return fail(c, GAT_ERR_ARG, "null argument");
    if (bytes < 16 || bytes % 16 != 0 || !aligned16(dev)) return fail(c, GAT_ERR_ARG, "the range must be whole, aligned 16-byte groups");
    if (launches < 1 || launches > 4096 || variant < 0 || variant > 15) return fail(c, GAT_ERR_RANGE, "1 .. 4096 launches, variants 0 .. 15");
    GAT_HIP(c, hipSetDevice(c->device));
    c->wait_seq = 0; // newer work than a flagged launch: gat_sync waits on the stream
    int32_t rc = ensure_partial(c, 64);
    if (rc != GAT_OK) return rc;
    const TraceRange trace("gat_debug_read_stream");
    for (int32_t i = 0; i < launches; ++i) {
        GAT_HIP(c, hipEventRecord(c->ev0, c->stream));
        GAT_HIP(c, launch_read_stream(dev, bytes, variant, c->num_cus, c->d_partial, c->stream));
        GAT_HIP(c, hipEventRecord(c->ev1, c->stream));
        GAT_HIP(c, hipEventSynchronize(c->ev1));
        GAT_HIP(c, hipEventElapsedTime(&ms_each[i], c->ev0, c->ev1));
    }
    return GAT_OK;
}

#ifdef GAT_MFMA_STAMPS
extern "C" GAT_API int32_t gat_debug_read(gat_ctx *c, unsigned long long *host, size_t count)
{
    if (!c || !c->dbg_ptr) return GAT_ERR_STATE;
    GAT_HIP(c, hipStreamSynchronize(c->stream));
    GAT_HIP(c, hipMemcpy(host, c->dbg_ptr, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return GAT_OK;
}
#endif

GAT_API int32_t gat_set_matrix_core(gat_ctx *c, int32_t enable)
{
    if (!c) return GAT_ERR_ARG;
    if (enable < GAT_MC_VECTOR || enable > GAT_MC_BF16_SPLIT) return fail(c, GAT_ERR_ARG, "unknown matrix-core mode");
    c->mc_mode = enable;
    drop_loop_graphs(c);
    return GAT_OK;
}

GAT_API int32_t gat_set_vector_tiling(gat_ctx *c, int32_t max_antenna_tiles, int32_t max_channels, int32_t max_blocks)
{
    if (!c) return GAT_ERR_ARG;
    if (max_antenna_tiles < 0 || max_channels < 0 || max_blocks < 0) return fail(c, GAT_ERR_ARG, "negative cap");
    if (max_antenna_tiles) c->max_aw = max_antenna_tiles >= 4 ? 4 : (max_antenna_tiles >= 2 ? 2 : 1);
    if (max_channels) c->max_kt = max_channels >= 4 ? 4 : (max_channels >= 2 ? 2 : 1);
    if (max_blocks) c->max_bpw = max_blocks;
    drop_loop_graphs(c);
    return GAT_OK;
}

GAT_API int32_t gat_set_option(gat_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return fail(c, GAT_ERR_ARG, "null argument");
    const int32_t rc = set_option(c, name, (long long)value);
    if (rc == GAT_OK) drop_loop_graphs(c); // recorded launch sequences bake the geometry in
    return rc;
}

GAT_API int32_t gat_last_launch_info(const gat_ctx *c, gat_launch_info *out, size_t struct_size)
{
    if (!c || !out || struct_size == 0) return GAT_ERR_ARG;
    // the struct grows at its end: a caller built against an older header gets the fields it knows
    std::memcpy(out, &c->last, std::min(struct_size, sizeof(gat_launch_info)));
    return GAT_OK;
}

} // extern "C"
