// gat_resident_api.cpp -- the resident correlator's host side (include/gat.h gat_resident_*; kernel: gat_resident.h): geometry
// from the planner, the doorbell (device memory behind the PCIe BAR, or pinned host memory) and the result lines in pinned
// host memory, ring / wait + second stage, restart after the kernel has left, park before anything that waits for the whole
// device, the host-closed tracking loop.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <new>

#include "gat_ctx.h"

using namespace gat;

namespace gat {

// ---- resident correlator: host side --------------------------------------------------------------------------------
constexpr size_t kResBellBytes = kResMaxChannels * kBellDwords * sizeof(unsigned); // 1024
constexpr size_t kResDevBytes = 64 + 8 * kResBellBytes;                          // "leaving" word (own line) | eight forwarded doorbells
// up to this many workgroups poll the host's doorbell themselves (gat_resident.h; 17 workgroups: 6.0 / 6.4 us polling
// directly, 5.7 / 7.4 forwarded; 33: 10.5 / 11.7 directly, 6.7 / 7.4 forwarded -- profiles/r04/r04r_*)
constexpr int kResHostPollers = 20;

static double mono_us()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

// host mirror of the kernels' `bad` predicate for host-resident records: what passes here is not poisoned there
int32_t validate_params(gat_ctx *c, const gat_channel_params *params_host, size_t n, double reach, double fs)
{
    for (size_t i = 0; i < n; ++i) {
        const gat_channel_params &p = params_host[i];
        if (p.prn < 0 || p.prn >= c->P) return fail(c, GAT_ERR_RANGE, "prn outside the code table");
        if (!std::isfinite(p.code_freq_hz) || !std::isfinite(p.carrier_freq_hz) ||
            !std::isfinite(p.code_phase_chips) || !std::isfinite(p.carrier_phase_cycles))
            return fail(c, GAT_ERR_ARG, "non-finite channel parameter");
        if (p.code_freq_hz < 0.0) return fail(c, GAT_ERR_RANGE, "negative code frequency");
        if (std::fabs(p.carrier_freq_hz / fs) >= 1.0e15 || std::fabs(p.carrier_phase_cycles) >= 1.0e15)
            return fail(c, GAT_ERR_RANGE, "carrier frequency / phase out of range");
        if (!code_span_ok(p.code_freq_hz / fs, p.code_phase_chips, reach, c->Lc))
            return fail(c, GAT_ERR_RANGE, "code phase span too large");
    }
    return GAT_OK;
}

// the doorbell may live behind the PCIe BAR (write-combining): stores are pushed out of the core's buffers before anybody waits
static inline void bell_flush(const gat_resident *res)
{
    if (res->d_bell) __builtin_ia32_sfence();
}
// dword 0 of line 0 (the ring's number, or "quit") in every copy of the doorbell
static void bell_set_seq(gat_resident *res, unsigned v)
{
    for (int cp = 0; cp < res->bell_copies; ++cp) __atomic_store_n(&res->h_bell[(size_t)cp * kResMaxChannels * kBellDwords], v, __ATOMIC_RELEASE);
    bell_flush(res);
}

// start the kernel: it has served everything up to start_seq; a ring with a newer number is served at once
static int32_t resident_start(gat_resident *res, unsigned start_seq)
{
    gat_ctx *c = res->ctx;
    __atomic_store_n(&res->h_state[0], (unsigned)kResidentRuns, __ATOMIC_RELEASE);
    res->h_state[1] = 0;
    // device words: the master's "leaving" word = 0; the eight forwarded doorbells say "nothing newer than start_seq"
    std::memset(res->h_init, 0, kResDevBytes);
    for (int c8 = 0; c8 < 8; ++c8) res->h_init[16 + c8 * (kResMaxChannels * kBellDwords)] = start_seq;
    res->r.start_seq = start_seq;
    res->a.codes = c->d_codes;
    hipError_t e = hipMemcpyAsync(res->d_quit, res->h_init, kResDevBytes, hipMemcpyHostToDevice, res->stream);
    if (e == hipSuccess) e = launch_dc_resident(res->a, res->cfg, res->r, res->stream);
    if (e != hipSuccess) {
        // no kernel is on the device: the state must not read "runs" (the next call would skip the restart and spin until its
        // deadline), and nothing is `running`
        __atomic_store_n(&res->h_state[0], (unsigned)kResidentQuit, __ATOMIC_RELEASE);
        res->last_exit = kResidentQuit;
        res->running = false;
        return hipfail(c, e, "resident correlator: kernel start");
    }
    res->running = true;
    ++res->launches;
    return GAT_OK;
}

// ask the kernel to leave and wait until it has (its own limits bound the wait)
static int32_t resident_park(gat_resident *res)
{
    if (!res->running) return GAT_OK;
    gat_ctx *c = res->ctx;
    if (__atomic_load_n(&res->h_state[0], __ATOMIC_ACQUIRE) == kResidentRuns) bell_set_seq(res, kBellQuit);
    GAT_HIP(c, hipStreamSynchronize(res->stream));
    bell_set_seq(res, res->seq); // line 0 is the last call's again
    res->last_exit = __atomic_load_n(&res->h_state[0], __ATOMIC_ACQUIRE);
    res->running = false;
    return GAT_OK;
}

void park_residents(gat_ctx *c)
{
    for (gat_resident *r : c->residents) (void)resident_park(r);
}

void resident_free(gat_resident *res)
{
    if (res->d_quit) (void)hipFree(res->d_quit);
    if (res->d_bell) (void)hipFree(res->d_bell);
    if (res->h_block) (void)hipHostFree(res->h_block);
    if (res->stream) (void)hipStreamDestroy(res->stream);
    delete res;
}

} // namespace gat

extern "C" {

// ---------------------------------------------------------------------------------------------------------------------
// Resident correlator (include/gat.h; kernel: gat_resident.h)
// ---------------------------------------------------------------------------------------------------------------------
GAT_API int32_t gat_resident_open(gat_ctx *c, const gat_signal_desc *sig, int32_t K, int32_t L, const int32_t *shifts,
                                  double fs, const gat_resident_config *config, gat_resident **out)
{
    if (!c) return GAT_ERR_ARG;
    if (!out || !sig || !shifts) return fail(c, GAT_ERR_ARG, "null argument");
    *out = nullptr;
    if (K < 1 || K > kResMaxChannels) return fail(c, K < 1 ? GAT_ERR_ARG : GAT_ERR_UNSUPPORTED, "resident correlator: 1 .. 16 channels");
    gat_resident_config cf{};
    if (config) {
        if (config->struct_size < sizeof(uint32_t)) return fail(c, GAT_ERR_ARG, "gat_resident_config.struct_size not set");
        std::memcpy(&cf, config, std::min<size_t>(config->struct_size, sizeof cf));
    }
    // (every workgroup has to be ON the device for a call to complete: no more of them than compute units)
    if ((int)cf.max_workgroups > c->num_cus) return fail(c, GAT_ERR_RANGE, "max_workgroups above the device's compute units");
    GAT_HIP(c, hipSetDevice(c->device));

    gat_resident *res = new (std::nothrow) gat_resident();
    if (!res) return fail(c, GAT_ERR_NOMEM, "out of memory");
    res->ctx = c;
    auto bail = [&](int32_t rc) {
        resident_free(res);
        return rc;
    };
    // the doorbell: device memory the host writes through the BAR (a ring is a posted write, every poll a local read), or the
    // pinned block below
    int large_bar = 0;
    if (cf.doorbell > 2) return bail(fail(c, GAT_ERR_ARG, "gat_resident_config.doorbell: 0, 1 or 2"));
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) != hipSuccess) large_bar = 0;
    (void)hipGetLastError();
    const bool bell_on_device = cf.doorbell == 2 || (cf.doorbell == 0 && large_bar);
    if (bell_on_device && !large_bar) return bail(fail(c, GAT_ERR_UNSUPPORTED, "resident correlator: the host cannot write device memory here (no large BAR)"));
    // geometry: the planner's, restricted to the resident instances.  More workgroups shorten every workgroup's walk and
    // lengthen the host's (two or three result lines each): best around 128 (profiles/r04/resident/v10_*); behind a
    // forwarded host doorbell 64
    DcPlan plan;
    plan.max_wgs = cf.max_workgroups ? cf.max_workgroups : std::min(c->num_cus, bell_on_device ? 128 : 64);
    const gat_channel_params dummy[kResMaxChannels] = {};
    float *const nonnull = reinterpret_cast<float *>(uintptr_t(64));
    int32_t rc = correlate_impl(c, sig, nullptr, 1, K, L, shifts, fs, nonnull, nonnull, 0, dummy, &plan);
    if (rc != GAT_OK) return bail(rc);
    if (!dc_has_resident_instance(plan.cfg.ant_tile, plan.cfg.taps, plan.cfg.format))
        return bail(fail(c, GAT_ERR_UNSUPPORTED, "resident correlator: no kernel instance for this shape"));
    res->a = plan.a;
    res->cfg = plan.cfg;
    res->K = K;
    res->L = L;
    res->M = sig->num_ants;
    res->N = sig->num_samples;
    res->fs = fs;
    res->spv = dc_group_samples(4, sig->layout);
    for (int l = 0; l < L; ++l) res->max_shift = std::max<long long>(res->max_shift, std::llabs((long long)shifts[l]));
    res->idle_us = cf.idle_us ? cf.idle_us : 5000u;
    res->life_ms = cf.life_ms ? cf.life_ms : 2000u;
    res->max_calls = cf.max_calls ? cf.max_calls : 0xfffffff0u;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) res->ticks_per_us = std::max(1, khz / 1000);
    (void)hipGetLastError();

    // Every workgroup has to be ON the device for a call to complete (a surplus workgroup starts only when a resident one has
    // left, i.e. never while the kernel serves calls): this correlator's workgroups and those of the context's other open
    // ones, each counted in compute units at what one unit holds of its instance (occupancy API with the launch's LDS; the
    // API may answer one block too many near scalar-register limits -- MI355X_MICROARCH.md, Residency --, so one is taken off
    // every answer above one), may not exceed the device's.
    res->wgs = (int)plan.a.total_wgs;
    {
        int occ = 0;
        hipError_t oe = dc_resident_blocks_per_cu(plan.cfg, &occ);
        if (oe != hipSuccess) return bail(hipfail(c, oe, "hipOccupancyMaxActiveBlocksPerMultiprocessor"));
        if (occ < 1) return bail(fail(c, GAT_ERR_UNSUPPORTED, "resident correlator: the kernel instance does not fit a compute unit with this LDS size"));
        res->blocks_per_cu = occ > 1 ? occ - 1 : 1;
        long long cus = (res->wgs + res->blocks_per_cu - 1) / res->blocks_per_cu;
        for (const gat_resident *o : c->residents) cus += (o->wgs + o->blocks_per_cu - 1) / o->blocks_per_cu;
        if (cus > c->num_cus)
            return bail(fail(c, GAT_ERR_UNSUPPORTED, "resident correlator: more workgroups than the device holds at once (with the context's other open resident correlators): lower max_workgroups, the channel count or close one"));
    }
    // pinned host block: doorbell | state | result lines of every workgroup
    res->nval = 2 * plan.cfg.ant_tile * plan.cfg.taps;
    res->lines_per_wg = (res->nval + kResLinePayload - 1) / kResLinePayload;
    for (int o = 0; o < res->nval; ++o) { // where the host's second stage finds a workgroup's values and where they go
        res->val_src.push_back((o / kResLinePayload) * 16 + o % kResLinePayload);
        const int ml = o >> 1;
        if (!(o & 1)) res->val_dst.push_back(plan.a.tap_index[ml / plan.cfg.ant_tile] * sig->num_ants + ml % plan.cfg.ant_tile);
    }
    const size_t host_bytes = kResBellBytes + 64 + kResDevBytes + (size_t)res->wgs * res->lines_per_wg * 64;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&res->h_block), host_bytes, hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) return bail(hipfail(c, e, "hipHostMalloc"));
    std::memset(res->h_block, 0, host_bytes);
    unsigned char *d_host = nullptr;
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void **>(&d_host), res->h_block, 0)) != hipSuccess) return bail(hipfail(c, e, "hipHostGetDevicePointer"));
    res->h_bell = reinterpret_cast<unsigned *>(res->h_block);
    res->r.host_bell = reinterpret_cast<const unsigned *>(d_host);
    res->h_state = reinterpret_cast<unsigned *>(res->h_block + kResBellBytes);
    res->r.host_state = reinterpret_cast<unsigned *>(d_host + kResBellBytes);
    res->h_init = reinterpret_cast<unsigned *>(res->h_block + kResBellBytes + 64); // staging of the device words' start values
    res->h_lines = reinterpret_cast<unsigned *>(res->h_block + kResBellBytes + 64 + kResDevBytes);
    res->r.host_lines = reinterpret_cast<unsigned *>(d_host + kResBellBytes + 64 + kResDevBytes);
    if ((e = hipMalloc(reinterpret_cast<void **>(&res->d_quit), kResDevBytes)) != hipSuccess) return bail(hipfail(c, e, "hipMalloc"));
    res->r.dev_quit = res->d_quit;
    res->r.dev_bell = res->d_quit + 16;
    if (bell_on_device) {
        res->bell_copies = 8;
        if ((e = hipExtMallocWithFlags(reinterpret_cast<void **>(&res->d_bell), 8 * kResBellBytes, hipDeviceMallocFinegrained)) != hipSuccess)
            return bail(hipfail(c, e, "hipExtMallocWithFlags"));
        res->h_bell = res->d_bell; // the host's view of it IS the device address
        res->r.host_bell = res->d_bell;
        for (size_t i = 0; i < 8 * kResBellBytes / sizeof(unsigned); ++i) res->h_bell[i] = 0u;
    }
    res->r.bell_copies = res->bell_copies;
    res->r.forward = !res->d_bell && res->wgs > (cf.host_pollers ? (int)cf.host_pollers : kResHostPollers) ? 1 : 0;
    // the body posts its sums through LDS: it stores nothing to device or host memory itself
    res->a.partial = nullptr;
    res->a.out_re = nullptr;
    res->a.out_im = nullptr;
    res->a.done_counter = nullptr;
    res->a.host_flag = nullptr;
    res->r.max_calls = res->max_calls;
    res->r.idle_ticks = (long long)res->idle_us * res->ticks_per_us;
    res->r.life_ticks = (long long)res->life_ms * 1000ll * res->ticks_per_us;
    if ((e = hipStreamCreateWithFlags(&res->stream, hipStreamNonBlocking)) != hipSuccess) return bail(hipfail(c, e, "hipStreamCreateWithFlags"));
    res->seq = 1; // "the last call": nothing is pending when the kernel starts
    bell_set_seq(res, res->seq);
    rc = resident_start(res, res->seq);
    if (rc != GAT_OK) return bail(rc);
    c->residents.push_back(res);
    *out = res;
    return GAT_OK;
}

GAT_API int32_t gat_resident_correlate(gat_resident *res, const gat_channel_params *params_host, int64_t block_offset,
                                       float *out_re_host, float *out_im_host)
{
    if (!res) return GAT_ERR_ARG;
    gat_ctx *c = res->ctx;
    if (!params_host || !out_re_host || !out_im_host) return fail(c, GAT_ERR_ARG, "null argument");
    if (res->stale) return fail(c, GAT_ERR_STATE, "the code table changed: open the resident correlator again");
    if (block_offset < 0 || block_offset % res->spv != 0 || block_offset >= (1ll << 40))
        return fail(c, GAT_ERR_ARG, "block offset must be a non-negative multiple of the samples one 16-byte load holds");
    int32_t rc = validate_params(c, params_host, (size_t)res->K, (double)(res->N + res->max_shift), res->fs);
    if (rc != GAT_OK) return rc;

    // ring: one line per channel, line 0 last; inside a line the two sequence words last
    unsigned prev = res->seq, seq = prev + 1;
    if (seq == 0u || seq == kBellQuit) seq = 1;
    if (seq == prev) ++seq;
    for (int k = res->K - 1; k >= 0; --k) {
        unsigned w[kBellDwords] = {};
        w[0] = seq;
        std::memcpy(&w[2], &params_host[k], sizeof(gat_channel_params));
        w[3] = 0; // the record's reserved word
        std::memcpy(&w[12], &block_offset, sizeof(int64_t));
        unsigned x = 0;
        for (int i = 0; i < 14; ++i) x ^= w[i];
        w[14] = x;
        w[15] = seq;
        for (int cp = 0; cp < res->bell_copies; ++cp) { // (device doorbell: one copy per blockIdx % 8, whole 64-byte lines: write-combined)
            unsigned *line = res->h_bell + ((size_t)cp * kResMaxChannels + k) * kBellDwords;
            if (res->d_bell) {
                for (int i = 0; i < kBellDwords; ++i) line[i] = w[i];
            } else {
                for (int i = 1; i < 15; ++i) line[i] = w[i];
                __atomic_store_n(&line[15], seq, __ATOMIC_RELEASE);
                __atomic_store_n(&line[0], seq, __ATOMIC_RELEASE);
            }
        }
    }
    bell_flush(res);
    res->seq = seq;
    if (!res->running || __atomic_load_n(&res->h_state[0], __ATOMIC_ACQUIRE) != kResidentRuns) {
        if (res->running) res->last_exit = res->h_state[0];
        GAT_HIP(c, hipSetDevice(c->device));
        if ((rc = resident_start(res, prev)) != GAT_OK) return rc;
    }
    // Wait + second stage in one walk over the workgroups' result lines, in slot order: a workgroup whose lines all carry the
    // call's number and pass their check is added to the outputs and never looked at again (its lines stay as they are until
    // the next ring; a restarted kernel posts the same values) -- what is left to do when the last line lands is the last
    // workgroup.  Slot = (antenna group * splits + split) * channels + channel: every output receives its splits in rising
    // order whatever the order of arrival (deterministic, the order of the device's second stage).  The lines are misses
    // in the host's caches once the device has written them: fetched ahead.
    const int lw = res->lines_per_wg, pairs = res->nval / 2, nlines = res->wgs * lw;
    const int MT = res->cfg.ant_tile, KG = res->a.KG, SP = res->a.splits;
    const size_t n = (size_t)res->K * res->L * res->M;
    std::memset(out_re_host, 0, n * sizeof(float));
    std::memset(out_im_host, 0, n * sizeof(float));
    const int *src = res->val_src.data(), *dst = res->val_dst.data();
    int slot = 0, kg = 0, ag = 0, sp = 0; // the next workgroup to take and its place in the call
    constexpr int kAhead = 12;
    auto take_arrived = [&]() { // true once every workgroup has been taken
        const unsigned long long *ln = reinterpret_cast<const unsigned long long *>(res->h_lines);
        for (; slot < res->wgs; ++slot) {
            const unsigned long long *q = ln + (size_t)slot * lw * 8;
            for (int j = 0; j < lw; ++j, q += 8) {
                if (slot * lw + j + kAhead < nlines) __builtin_prefetch(q + kAhead * 8);
                unsigned long long x = 0, last = 0;
                for (int i = 0; i < 8; ++i) x ^= (last = __atomic_load_n(&q[i], __ATOMIC_RELAXED));
                // words 0..13 payload, 14 = seq ^ xor(payload), 15 = seq: the sixteen words of a whole line of this call xor to zero
                if ((unsigned)(last >> 32) != seq || (unsigned)(x >> 32) != (unsigned)x) return false;
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            const float *w = reinterpret_cast<const float *>(res->h_lines) + (size_t)slot * lw * 16;
            float *o_re = out_re_host + (size_t)kg * res->L * res->M + ag * MT, *o_im = out_im_host + (size_t)kg * res->L * res->M + ag * MT;
            for (int i = 0; i < pairs; ++i) {
                o_re[dst[i]] += w[src[2 * i]];
                o_im[dst[i]] += w[src[2 * i + 1]];
            }
            if (++kg == KG) {
                kg = 0;
                if (++sp == SP) { sp = 0; ++ag; }
            }
        }
        return true;
    };
    const double t0 = mono_us(), deadline = (double)res->life_ms * 1000.0 + 1.0e6;
#ifdef GAT_RES_STAMPS
    double t_first = 0.0;
#endif
    for (unsigned spins = 0;; ++spins) {
        if (take_arrived()) break;
#ifdef GAT_RES_STAMPS
        if (t_first == 0.0 && slot > 0) t_first = mono_us();
#endif
        if ((spins & 15u) != 15u) continue;
        if (__atomic_load_n(&res->h_state[0], __ATOMIC_ACQUIRE) != kResidentRuns) {
            // the kernel has left (idle, lifetime, call budget) -- with this call served or not
            if (take_arrived()) break;
            res->last_exit = res->h_state[0];
            GAT_HIP(c, hipSetDevice(c->device));
            if ((rc = resident_start(res, prev)) != GAT_OK) return rc;
        }
        if (mono_us() - t0 > deadline) {
            (void)resident_park(res);
            return fail(c, GAT_ERR_STATE, "resident correlator: no answer from the device");
        }
    }
#ifdef GAT_RES_STAMPS
    const double t_all = mono_us();
    if (t_first == 0.0) t_first = t_all;
    res->host_us[0] += t_first - t0;
    res->host_us[1] += t_all - t_first;
#endif
    ++res->calls;
    return GAT_OK;
}

GAT_API int32_t gat_resident_tracking_run(gat_resident *res, int32_t num_blocks, int64_t first_block_offset, int64_t block_stride,
                                          const gat_loop_config *cfg, gat_loop_state *state_host, gat_channel_params *params_host,
                                          float *acc_re_host, float *acc_im_host, int64_t acc_block_stride)
{
    if (!res) return GAT_ERR_ARG;
    gat_ctx *c = res->ctx;
    if (!cfg || !state_host || !params_host || !acc_re_host || !acc_im_host) return fail(c, GAT_ERR_ARG, "null argument");
    if (num_blocks < 0 || first_block_offset < 0 || block_stride < 0 || acc_block_stride < 0) return fail(c, GAT_ERR_ARG, "negative count, offset or stride");
    const int64_t n = (int64_t)res->K * res->L * res->M;
    if (acc_block_stride != 0 && acc_block_stride < n) return fail(c, GAT_ERR_ARG, "acc_block_stride below one block's M x L x K accumulators");
    if (cfg->num_taps != res->L) return fail(c, GAT_ERR_ARG, "loop configuration: num_taps differs from the correlator's");
    for (int32_t b = 0; b < num_blocks; ++b) {
        float *re = acc_re_host + (size_t)b * acc_block_stride, *im = acc_im_host + (size_t)b * acc_block_stride;
        int32_t rc = gat_resident_correlate(res, params_host, first_block_offset + (int64_t)b * block_stride, re, im);
        if (rc != GAT_OK) return rc;
        rc = gat_tracking_update_host(re, im, res->K, res->M, cfg, state_host, params_host, params_host);
        if (rc != GAT_OK) return fail(c, rc, "gat_tracking_update_host: loop configuration");
    }
    return GAT_OK;
}

GAT_API int32_t gat_resident_info_get(const gat_resident *res, gat_resident_info *out, size_t struct_size)
{
    if (!res || !out || struct_size == 0) return GAT_ERR_ARG;
    gat_resident_info i{};
    i.workgroups = (int32_t)res->a.total_wgs;
    i.splits = res->a.splits;
    const bool ended = res->running && __atomic_load_n(&res->h_state[0], __ATOMIC_ACQUIRE) != kResidentRuns;
    i.running = res->running && !ended ? 1 : 0;
    i.last_exit = (int32_t)(ended ? res->h_state[0] : res->last_exit);
    i.launches = res->launches;
    i.calls = res->calls;
    std::memcpy(out, &i, std::min(struct_size, sizeof i));
    return GAT_OK;
}

GAT_API int32_t gat_resident_park(gat_resident *res)
{
    if (!res) return GAT_ERR_ARG;
    (void)hipSetDevice(res->ctx->device);
    return resident_park(res);
}

GAT_API int32_t gat_resident_park_all(gat_ctx *c)
{
    if (!c) return GAT_ERR_ARG;
    GAT_HIP(c, hipSetDevice(c->device));
    int32_t rc = GAT_OK;
    // ring "quit" into every running kernel first, then wait for them one by one: the waits overlap
    for (gat_resident *r : c->residents)
        if (r->running && __atomic_load_n(&r->h_state[0], __ATOMIC_ACQUIRE) == kResidentRuns) bell_set_seq(r, kBellQuit);
    for (gat_resident *r : c->residents) {
        const int32_t one = resident_park(r);
        if (rc == GAT_OK) rc = one;
    }
    return rc;
}

GAT_API int32_t gat_resident_close(gat_resident *res)
{
    if (!res) return GAT_ERR_ARG;
    gat_ctx *c = res->ctx;
    (void)hipSetDevice(c->device);
    const int32_t rc = resident_park(res);
#ifdef GAT_RES_STAMPS
    std::fprintf(stderr, "resident stamps of the last call (10 ns ticks): ring seen -> barrier + acquire %u; then tile decode %u, first loads issued %u, parameters %u, setup (barrier) %u, first segment %u, steps %u, reduction %u, result lines %u (= %u counts of clock64)\n",
                 res->h_state[11], res->h_state[4 + 1], res->h_state[12], res->h_state[13], res->h_state[4 + 2], res->h_state[4 + 3], res->h_state[4 + 4], res->h_state[4 + 5], res->h_state[4 + 6], res->h_state[14]);
    if (res->calls)
        std::fprintf(stderr, "host side, mean over %llu calls (us): ring written -> first workgroup's result lines whole and added %.2f, -> all %d lines of %d workgroups %.2f\n",
                     (unsigned long long)res->calls, res->host_us[0] / (double)res->calls, res->wgs * res->lines_per_wg, res->wgs, res->host_us[1] / (double)res->calls);
#endif
    c->residents.erase(std::remove(c->residents.begin(), c->residents.end(), res), c->residents.end());
    resident_free(res);
    return rc;
}

} // extern "C"
