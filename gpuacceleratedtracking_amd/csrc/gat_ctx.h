// gat_ctx.h -- what the translation units of the C-ABI layer share (gat_api.cpp: contexts, operators, loop; gat_planner.cpp: the
// correlator call's launch planning; gat_group.cpp: device groups; gat_resident_api.cpp: the resident correlator's host side): the context, error helpers, the planner's entry point.
#pragma once

#include <cmath>
#include <string>
#include <vector>

#include "gat_internal.h"

struct gat_ctx;

// A resident correlator (gat_resident_open): one bounded-lifetime kernel serving single-block calls rung in through
// pinned host memory (gat_resident.h).  Owned by its context's list until gat_resident_close.
struct gat_resident {
    gat_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;  // its own non-blocking stream: the kernel runs next to the context's work
    gat::DcArgs a{};
    gat::DcLaunch cfg{};
    gat::ResidentArgs r{};
    unsigned char *h_block = nullptr; // pinned: doorbell lines | state | result lines
    unsigned *h_bell = nullptr, *h_state = nullptr, *h_lines = nullptr, *h_init = nullptr;
    unsigned *d_quit = nullptr;       // device: the master's "I am leaving" word | eight forwarded doorbells
    unsigned *d_bell = nullptr;       // device (fine-grained, host-writable through the BAR): the doorbell's copies, or null: it is in h_block
    int bell_copies = 1;
    int wgs = 0, lines_per_wg = 0, nval = 0; // working workgroups, result lines and values of each
    int blocks_per_cu = 1;                   // workgroups of its kernel instance one compute unit holds at once
    std::vector<int> val_src, val_dst; // value pair (re, im) i of a workgroup: word of its lines holding re | tap * M + antenna of the tile it adds to
    unsigned seq = 0;                 // sequence number of the last call
    bool running = false;             // a kernel was started and has not been seen to end
    bool stale = false;               // the code table changed: the correlator has to be opened again
    int K = 0, L = 0, M = 0, spv = 1;
    long long N = 0, max_shift = 0;
    double fs = 0.0;
    uint32_t idle_us = 0, life_ms = 0, max_calls = 0;
    long long ticks_per_us = 100;
    unsigned last_exit = 0;
    uint64_t launches = 0, calls = 0;
#ifdef GAT_RES_STAMPS
    double host_us[2] = {}; // sums over the calls: ring written -> first workgroup taken -> every workgroup taken
#endif
};

struct gat_ctx {
    std::vector<gat_resident *> residents; // open resident correlators (parked before device-wide waits)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int8_t *d_codes = nullptr;
    // gat_tracking_run with GAT_FLAG_GRAPH: the launch sequence of the last such call, instantiated (replayed when the
    // next call has the same arguments: a receiver cycling through one ring buffer)
    struct LoopGraph {
        std::vector<unsigned char> key;
        hipGraphExec_t exec = nullptr;
        unsigned long long last_use = 0;
    };
    std::vector<LoopGraph> loop_graphs; // small LRU (kMaxLoopGraphs): e.g. the a/b parameter order of odd block counts
    unsigned long long loop_graph_clock = 0;
    void *d_zeros = nullptr;         // 64 zero bytes (out-of-range sample loads of the split-bf16 kernel read these)
    uint32_t *d_code_bits = nullptr; // bit i of row p = (chip i of PRN p is -1); only when every chip is +-1
    int code_bits_stride = 0;        // dwords per row, a multiple of 4
    int Lc = 0, P = 0, code_row_stride = 0; // rows padded to a multiple of 16 bytes
    float *d_partial = nullptr;
    size_t partial_bytes = 0;
    // completion flag (latency regime): small launches of the vector kernel end by storing a sequence number into pinned
    // host memory; gat_sync spins on it instead of going through hipStreamSynchronize (~5 us sooner)
    unsigned *h_flag = nullptr;      // pinned, host address
    unsigned *d_flag = nullptr;      // the same word, device address
    unsigned *d_done = nullptr;      // device: arrival counter of a launch's workgroups
    unsigned flag_seq = 0;           // last sequence number handed to a launch
    unsigned wait_seq = 0;           // != 0: the newest work on the stream is a flagged launch with this number
    int flag_max_wgs = 1024;         // option sync_flag_wgs: largest launch that carries the flag (0: never)
    gat_channel_params *d_params = nullptr;
    size_t params_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timer_running = false;
    std::vector<hipEvent_t> lap_events; // gat_timer_lap: pool, grows to the most laps ever outstanding
    size_t laps = 0;                    // laps recorded since the last gat_timer_laps
    int num_cus = 256;
    unsigned long long *dbg_ptr = nullptr; // diagnostic builds only
    int max_ant_tile = gat::kMaxAntTile; // option max_ant_tile (gat_set_option)
    int max_aw = 4, max_kt = 4, max_bpw = 16; // options dc_aw / dc_kt / dc_bpw (gat_set_vector_tiling): caps of the vector kernel's geometry
    int force_bpw = 0;                        // option dc_bpw_force: blocks per workgroup whatever the planner's rule says (A/B runs)
    int wgs_per_cu = 0;                       // option dc_wgs_per_cu: workgroups per CU the split planner aims for (0: by instance)
    int one_wave = 1;                         // option dc_one_wave = 0: never use one-wave workgroups
    long long one_wave_min = -1;              // option dc_one_wave_min: fewest (block, channel, tile) groups for them (default 32 per CU)
    int one_wave_seg = gat::kOneWaveSegSteps;      // option dc_ow_seg: steps per replica segment of a one-wave workgroup
    int max_depth = 2;                        // option dc_depth: cap of the sample prefetch depth (register sets per wave)
    int keep_l2 = -1;                         // option dc_keep_l2: cache policy of the sample loads (-1: by rule)
    int quads = -1;                           // option dc_quads: replica fill by quads (-1 by rule, 0 never, 1 wherever possible)
    int bit_tables = 1;                       // option dc_bits: chip tables staged as sign bits (0 never, 1 long codes, 2 whenever chips are +-1)
    int aw2 = -1;                             // option dc_aw2: the two-channel 2 x 2 tile (-1 by rule, 0 never, 1 wherever possible)
    int seg_cap = 0;                          // option dc_seg: cap of the steps per replica segment (0: by instance)
    int align_head = 1;                       // option dc_align: line-aligned virtual block starts where blocks start off a line
    int mc_mode = 1; // GAT_MC_* kernel selection (gat_set_matrix_core)
    int mc_nct = 0;       // option mc_nct: column tiles per workgroup of the split-bf16 kernel to try first (0: by rule; A/B runs)
    int mc_i16_terms = 2; // option mc_i16_terms: bf16 terms per int16 sample on the split-bf16 kernel (2: exact two-term split; 3: the float path's split)
    std::string err;
    gat_launch_info last{};
};

namespace gat {

inline int32_t fail(gat_ctx *c, int32_t code, const char *msg)
{
    if (c) c->err = msg;
    return code;
}

inline int32_t hipfail(gat_ctx *c, hipError_t e, const char *where)
{
    if (c) {
        c->err = std::string(where) + ": " + hipGetErrorString(e);
    }
    return -(int32_t)e;
}

#define GAT_HIP(c, call)                                      \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return hipfail((c), e_, #call); \
    } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// host mirror of the kernels' code_span_bad (gat_phase.h): what passes here is not poisoned there
inline bool code_span_ok(double ratio, double tau, double reach, int Lc)
{
    const double span = std::fabs(tau) + std::fabs(ratio) * reach + 1.0;
    return span < 1073741824.0 && (Lc <= 0 || span < 2097152.0 * (double)Lc) && ratio >= 0.0;
}

// Tracing ranges around the library's launch sequences (the reference wraps every launch of kernel_algorithm in
// NVTX.@range, src/algorithms.jl:953 ...): roctxRangePush / Pop, resolved at the first use (gat_api.cpp).  RAII.
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
    const void *rx;
};
// scratch and graph housekeeping shared by the planner, the operators and the loop (gat_api.cpp)
void drop_loop_graphs(gat_ctx *c);
int32_t ensure_partial(gat_ctx *c, size_t bytes);
int32_t upload_params(gat_ctx *c, const gat_channel_params *params_host, size_t n);

// What the planner hands to gat_resident_open instead of launching: the arguments and geometry of the ONE vector launch
// that would serve the call (four-wave workgroups, one antenna tile and one channel each: the resident instances).
struct DcPlan {
    long long max_wgs = 64; // in: workgroups the block's samples may be split over (times antenna tiles and channels)
    gat::DcArgs a{};
    gat::DcLaunch cfg{};
};

// params_dev: [B*K] records on the device -- or null with params_inline: B*K <= kInlineParams validated HOST records that
// travel inside the vector kernel's arguments (uploaded after all if a matrix-core kernel takes the call)
// plan_out != null: nothing is launched; GAT_ERR_UNSUPPORTED unless the call is exactly one launch of the vector kernel
int32_t correlate_impl(gat_ctx *c, const gat_signal_desc *sig, const gat_channel_params *params_dev, int32_t B, int32_t K, int32_t L,
                       const int32_t *shifts, double fs, float *out_re, float *out_im, uint32_t flags,
                       const gat_channel_params *params_inline = nullptr, DcPlan *plan_out = nullptr);
// host mirror of the kernels' `bad` predicate for host-resident records: what passes here is not poisoned there
int32_t validate_params(gat_ctx *c, const gat_channel_params *params_host, size_t n, double reach, double fs);
// the context's resident correlators (gat_resident_api.cpp): asked to leave before anything that waits for the whole device
// (hipFree, a new code table, the context's end); freed with the context
void park_residents(gat_ctx *c);
void resident_free(gat_resident *res);

} // namespace gat
