// ct_api.cpp -- the C ABI of include/cloudtrace.h on top of the gfx950 kernels.
//
// Host-side restatement of what the reference's scene items publish to OptiX before a
// launch (Scene::init Scene.cpp:36-46, Sun::init Sun.cpp:13-18, VDBCloud::init
// VDBCloud.cpp:15-137, Mie.cpp:8206-8297, Camera::init Camera.cpp:22-66), then thin launch
// wrappers.  No exception leaves this file and nothing here falls back to a CPU path: if
// HIP or the device is missing every entry point reports CT_E_NODEVICE / CT_E_HIP.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <map>
#include <vector>

#include "../../include/cloudtrace.h"
#include "ct_internal.hpp"

using namespace ct;

// ---- tuning knobs ----------------------------------------------------------------------------------------------------------
// Every environment variable the library reads, copied ONCE per ct_create into the handle (DESIGN.md 4.4 lists them with
// what they do and what was measured).  Knobs choose schedules, scratch sizes, layouts and diagnostics; none changes a
// result (tests/test_gpu_parity.py: the knob test).  get() returns the variable's text as it was at ct_create, or NULL.
struct Knob {
    bool is_set = false;
    std::string text;
    const char *get() const { return is_set ? text.c_str() : nullptr; }
    explicit operator bool() const { return is_set; }
};
#define CT_KNOBS(X) X(BURST_IDLE) X(BURST_MARCH_MIN) X(BURST_SCATTER) X(CHUNK_INTERLEAVE) X(CHUNK_MORTON) X(CONTINUATION) X(DEBUG_INVARIANTS) X(DELTA_INTERIOR) X(DELTA_NEE) X(EXCHANGE) X(HAND_ON_JOBS) X(HINT_PERIOD) X(JOB_MAX) X(JOB_WORK) X(MARCH_BURST) X(MAX_AGE) X(NEE_CACHE) X(NO_ADVANCE) X(POINT_BLOCKS_PER_CU) X(POINT_ORDER) X(REGEN_MIN) X(RENDER_AHEAD) X(SCATTER_MIN) X(SCATTER_RATIO) X(SCRATCH_GIB) X(SCRATCH_MIB) X(SERPENTINE) X(SHARED_DEPTH) X(SPARSE) X(STATS) X(TAIL_BURST) X(TILE_ORDER) X(TIMELINE) X(TRACE) X(TUNE_SUBFRAMES) X(XCD_QUEUES) X(XCD_QUEUES_UNTUNED) X(XCD_REGIONS) X(BLOCKS_PER_CU)
struct CtTuning {
#define X(name) Knob name;
    CT_KNOBS(X)
#undef X
    static CtTuning from_env()
    {
        CtTuning t;
#define X(name)                                  \
    if (const char *e = getenv("CT_" #name)) {   \
        t.name.is_set = true;                    \
        t.name.text = e;                         \
    }
        CT_KNOBS(X)
#undef X
        return t;
    }
};

struct CtHandle_ {
    CtTuning tune;         // the environment's knobs as they were at ct_create
    CtScene scene{};       // as given (host pointers are NOT retained)
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = { nullptr, nullptr, nullptr };

    DevScene dev{};
    bool camera_set = false;

    // device memory
    uint8_t *d_density = nullptr, *d_inscatter = nullptr, *d_dist = nullptr, *d_dist_tmp = nullptr, *d_majorant = nullptr, *d_maj_cells = nullptr, *d_maj_codes = nullptr;
    uint8_t *d_dbricks = nullptr, *d_ibricks = nullptr, *d_mbricks = nullptr, *d_tbricks = nullptr;
    uint2 *d_mrows = nullptr;          // sparse march bricks: extent of every brick row (DevScene::m_rows)
    uint8_t *d_mcoarse = nullptr;      // ... and the clearance of the coarse cells outside the extents
    size_t mbricks_dense_bytes = 0, mbricks_bytes = 0;
    // CT_FLAG_VMM_BRICKS: d_mbricks is a reserved virtual range (not a hipMalloc), backed chunk by chunk
    struct VmmBricks {
        void *va = nullptr;
        size_t size = 0, chunk = 0;
        std::vector<hipMemGenericAllocationHandle_t> handles;   // one per chunk with memory of its own + the shared ones
        size_t real_chunks = 0, shared_chunks = 0, mapped_chunks = 0;
    } vmm;
    uint8_t *d_pyramid = nullptr;     // density mip pyramid, built on first use (ct_collect_descriptors)
    MipPyramid pyramid{};
    float *d_mie = nullptr, *d_chopped = nullptr, *d_cdf = nullptr;
    uint16_t *d_guide = nullptr;
    float4 *d_frame = nullptr, *d_mean = nullptr, *d_m2 = nullptr;
    uchar4 *d_screen = nullptr;
    // Batches enqueued with ct_render_accumulate_async form a pipeline on the handle's stream (M = max_age):
    //     R1  R2 .. R(1+M) A1  R(2+M) A2  ...  (flush:) Rf A(n-M+1) .. An
    // The estimator launch R(k) does not run its surviving paths to their end when its job list is empty: it suspends
    // them (BatchArgs::cont_out) and R(k+1) resumes them first, so no launch ends with a tail of waves that carry a few
    // long paths each.  A path may be suspended M times, so batch k is complete once R(k+M) has run, and A(k), its
    // accumulate kernel, follows that launch (or the flush launch Rf, which only resumes and runs everything to its end).
    // The per-sample scratch is a ring of n_regions = M + 1 regions ([S][stride] compact, or [S][H][W] for the simple
    // kernel): batch k writes region k mod n_regions, which A(k - n_regions) has long left.  M = 1 (two regions) is
    // round 2's scheme and what long batches use -- a launch of 20 ms outlasts the longest path (2000 bounces, ~10 ms);
    // short batches (the reference renders 10 subframes per display update, Camera.cpp:189) get more regions, so that a
    // launch never has to wait for a path that an earlier one handed to it.
    // A slot owns a region, its queue counters and its events; the suspended paths alternate between two buffers.
    static constexpr int kMaxRegions = 64;
    struct Slot {
        uint32_t *queue = nullptr;
        hipEvent_t ev_in = nullptr, ev_start = nullptr, ev_done = nullptr, ev_acc0 = nullptr, ev_acc1 = nullptr;
        bool pending = false;              // launched, kernel times not booked yet
        bool accumulated = false;          // its accumulate kernel has been enqueued (ev_acc0/1 valid)
        bool awaits_accumulate = false;    // in `waiting`: some of its subframes are still to be accumulated
        bool complete = false;             // no path of the batch is in flight any more (max_age launches have followed, or a flush)
        uint32_t acc_done = 0;             // subframes of the batch whose accumulate kernels are enqueued
        uint32_t first = 0, S = 0;
        uint32_t rank_base = 0, groups = 0;   // the chunk of pixel groups this launch renders (places in the job order)
        bool with_misses = false;          // its accumulate kernel also accounts for the pixels that miss the box (once per batch)
        bool last_chunk = true;            // ... and is the last of its batch: the running mean is a whole image again after it
    };
    Slot slots[kMaxRegions];
    int n_regions = 2;                     // regions of the scratch in use
    int next_slot = 0;
    std::vector<int> waiting;              // slots with subframes still to be accumulated, oldest first
    // Render-ahead (ct_set_render_ahead, CT_RENDER_AHEAD): enqueued calls of fewer subframes than `ahead` -- the reference's
    // display loop asks for 10 at a time, Camera.cpp:189 -- are served by estimator launches of `ahead` subframes, of which
    // every call accumulates its own share: R(k) a(k-M,0) a(k-M,1) .. R(k+1) a(k-M+1,0) ..  A launch of 10 subframes is
    // mostly beginning and end -- every lane resumes a path and suspends one -- which a launch of `ahead` amortises (DESIGN.md
    // 4.3 item 13).  `rendered` >= `subframes`: the subframes the estimator has been launched for / the caller has asked
    // for; the running mean follows the calls by M * ahead subframes until something waits (flush: exactly `subframes`).
    uint32_t ahead = 0;
    uint32_t rendered = 0;
    // Stop-when-converged (ct_set_stop_when_converged): behind the accumulate kernel of every `stop_cadence`-th subframe (from
    // `stop_min` on) the convergence test runs on the device, and once it holds the accumulate kernels leave the running
    // mean alone -- Camera::render's `if (!isConverged())` (Camera.cpp:179) without a host round trip per update.
    // d_freeze: converged_freeze_kernel's state words; freeze_host: their first four, copied back after every test (pinned).
    uint32_t stop_cadence = 0, stop_min = 100;
    uint32_t *d_freeze = nullptr, *freeze_host = nullptr;
    uint32_t *cont[2] = { nullptr, nullptr }; // suspended paths: launch k writes cont[k & 1], launch k+1 reads it
    uint64_t launch_no = 0;                // estimator launches enqueued so far
    bool cont_live = false;                // the last launch may have suspended paths: the next one resumes them
    float4 *d_frames_all = nullptr;        // the whole per-sample scratch
    size_t frames_total = 0;               // float4 allocated
    size_t slot_capacity = 0;              // float4 per region
    uint64_t scratch_cap_bytes = 0;        // 0 = CT_SCRATCH_GIB / the default; else what an out-of-memory allocation left us with
    uint32_t layout_S = 0;                 // batch size the regions were laid out for
    uint32_t *left[2] = { nullptr, nullptr }; // job remainders handed on the same way (BatchArgs::left_out)
    size_t left_capacity = 0;              // entries per buffer: one per wave
    uint32_t *d_cont_count = nullptr;      // [0,1] entries in cont[i], [2] resume cursor, [3,4] entries in left[i], [5] its cursor
    unsigned long long *d_cont_total = nullptr; // paths handed from one launch to the next so far (ct_debug_suspended)
    size_t cont_capacity = 0;              // entries per buffer
    hipEvent_t ev_flush0 = nullptr, ev_flush1 = nullptr;
    bool continuation = true;              // CT_CONTINUATION=0: async batches run every path to its end
    int max_age_override = 0;              // CT_MAX_AGE=n: n + 1 regions whatever the batch size (0 = by batch duration)
    bool hand_on_jobs = true;              // CT_HAND_ON_JOBS=0: a wave finishes its own job before it suspends (A/B)
    bool serpentine = false;               // CT_SERPENTINE=1: short launches walk their job queues alternately forwards and backwards
    // work queue of the persistent kernel (rebuilt when the camera moves)
    float4 *d_primary = nullptr;      // cached primary rays, 2 float4 per pixel
    float4 *d_advance = nullptr;      // per pixel: pre-walked prefix of the primary march (MARCH estimator)
    uint32_t *d_pixels = nullptr;     // this shard's box-hitting pixels, padded to groups of 64
    uint8_t *d_hit = nullptr, *hit_host = nullptr;   // per pixel: the primary ray hits the box (device; pinned host copy)
    uint32_t *d_cost = nullptr;       // measured per group: [0,n) sum of path costs, [n,2n) deepest path
    uint32_t *d_touched[2] = { nullptr, nullptr };   // ct_debug_track_lines: one bit per line of the density / shadow arrays
    size_t touched_lines[2] = { 0, 0 };
    unsigned long long *d_timeline = nullptr;   // CT_TIMELINE=1: [start, end] of every wave of the last enqueued estimator launch (MARCH)
    uint2 *d_cost_plane = nullptr;    // ... as the cost-measuring launch leaves them, per sample (BatchArgs::cost)
    size_t cost_plane_capacity = 0;
    uint32_t *d_job_group = nullptr, *d_job_sub = nullptr; // job list of the current batch size
    uint32_t n_groups = 0, groups_capacity = 0;
    uint32_t n_jobs = 0, jobs_capacity = 0, jobs_S = 0;
    uint32_t *jobs_host_g = nullptr, *jobs_host_s = nullptr;   // the list as built on the host (pinned)
    size_t jobs_host_capacity = 0;
    bool jobs_brief = false;          // the list was laid out for short launches (short_batch)
    // The job list is built chunk by chunk: a chunk is a contiguous piece of `chunk_groups` groups of the cost-sorted
    // group order, and a launch renders all S subframes of ONE chunk into a scratch region of chunk_groups * 64 columns --
    // so the per-sample scratch does not have to hold the whole frame, while a launch still works through a few pixel groups at
    // a time for all their subframes (what keeps its paths close together in the volume; cutting a batch by SUBFRAMES
    // instead makes every launch sweep the whole image and costs 9-14 %).
    uint32_t chunk_groups = 0, n_chunks = 0;
    std::vector<std::array<uint32_t, kQueues + 2>> chunk_q_begin;   // per chunk: job ranges of the queues (absolute indices)
    uint32_t *d_group_rank = nullptr, *d_group_order = nullptr;     // place of a group in the job order / the group at a place
    uint32_t jobs_hint = 0;           // batch size the caller asked for last (job lists are built for it)
    // subframes per job at most (cheap groups), and the bounces (x cost unit) a job's lane is expected to run.
    // Re-swept on the final kernels (8 / 256 before): +4.3 % at 512^3, +5.6 % at 1024^3, +2.6 % at 256^3, +1.3 % DELTA
    // (16 / 48 until the end of round 2; 16 / 16 since: the whole-frame launch does not care, 3288 either way, a rank's
    // launch of an eighth of the tiles is 1.6 % shorter, 43.6 instead of 44.3 ms)
    uint32_t job_max = 16;
    float job_work = 16.f;
    uint32_t q_begin[kQueues + 2] = {}; // job ranges of the per-XCD queues + the shared one
    uint64_t own_pixels = 0, hit_pixels = 0;
    bool queue_dirty = true, order_tuned = false;
    bool no_advance = false;             // CT_NO_ADVANCE=1: samples start at the box face (A/B)
    bool queues_enabled = false;         // per-XCD regions (CT_XCD_QUEUES=1; default: one global list)
    float shared_depth = 1e30f;          // groups at least this deep (bounces) use the shared queue
    uint32_t regions = 128;              // image regions dealt to the per-XCD queues
    std::vector<uint32_t> group_order;   // groups, most expensive first (until tuned: by what the last pose measured for their tiles)
    std::vector<uint32_t> group_tile;    // the 8x8 tile of a group's first pixel
    std::vector<uint32_t> tile_deepest;  // per tile of the image: the deepest path the last measured pose produced there (0 = never measured)
    std::vector<uint32_t> job_order;     // the order the job list is built in: group_order, or its chunks interleaved (build_jobs)
    bool chunk_interleave = false;       // CT_CHUNK_INTERLEAVE=1: every chunk is every C-th group of group_order (A/B: worse, the neighbours are gone)
    bool chunk_morton = false;           // CT_CHUNK_MORTON=1: chunks are compact image regions (A/B)
    bool tile_hilbert = false;           // CT_TILE_ORDER=hilbert: pixel groups along a Hilbert curve instead of Morton order (A/B)
    std::vector<float> group_depth;      // measured mean path cost per group (0 until tuned), in the
                                         // units of BatchArgs::cost
    unsigned long long host_paths = 0, host_hits = 0; // paths / box hits of the persistent path
    uint32_t *d_queue = nullptr;
    unsigned long long *d_counters = nullptr; // kCounterCount + 1 (unconverged) + kStatCount
    float *d_colsum = nullptr, *d_avg = nullptr;
    uint32_t reinhard_generation = 0;   // launches on d_avg's barrier counter (launch_reinhard)

    // CT_DEBUG_INVARIANTS=1: the diagnostics build of the estimator counts samples dealt / paths resumed / results
    // written / paths suspended, the scratch is filled with NaNs before every launch, and every point at which
    // nothing is in flight checks: dealt + resumed == written + suspended, resumed == suspended, dealt == what the
    // host handed out, no sample without alpha 1 reached an accumulate kernel.  A violation fails the call.
    bool debug_invariants = false;
    uint64_t iv_expected_dealt = 0, iv_checks = 0, iv_violations = 0;

    // ct_point_radiance_launch: a collector calls it about a thousand times per scene setup, each call a launch of a few
    // milliseconds, so its device buffers stay (grown on demand; freeing one would wait for the whole device, i.e. for
    // the launches of the other scene setups in flight)
    struct PointBuffers {
        CtPointRadianceTask *tasks = nullptr;
        float4 *primary = nullptr, *frames = nullptr;
        uint32_t *pixels = nullptr, *jg = nullptr, *js = nullptr;
        size_t cap_tasks = 0, cap_primary = 0, cap_pixels = 0, cap_frames = 0, cap_jg = 0, cap_js = 0;
    } pt;
    bool point_order = true;             // CT_POINT_ORDER=0: jobs of 8 frames in task order over 8 queues, as until round 2 (A/B)

    size_t volume_bytes = 0;
    LaunchShape shape{ 1024, 256, false };
    // CT_EXCHANGE=1: the estimator kernels with a block-wide exchange of paths between waves (ct_exchange.hpp) render the
    // batches whose job order is tuned; the cost-measuring launch of a pose keeps the per-lane kernels
    int exchange = 0;                  // 0 per-lane kernels, 1 block-wide exchange, 2 exchange within a wave
    LaunchShape xshape{ 256, 1024, false };
    uint32_t subframes = 0;
    double render_ms = 0, accum_ms = 0;
    uint64_t launches = 0;
    std::string error;
};

static thread_local std::string g_create_error;

static int fail(CtHandle h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) {
        h->error = buf;
    } else {
        g_create_error = buf;
    }
    return code;
}

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            return fail((h), e_ == hipErrorOutOfMemory ? CT_E_NOMEM : CT_E_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                      \
        }                                                                                            \
    } while (0)

// CT_TRACE=1: host-side phases of a new pose (rebuild_queue, tune_order, build_jobs) with their durations on stderr.
struct TracePhase {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    bool on;
    TracePhase(CtHandle h, const char *n);
    ~TracePhase()
    {
        if (on) {
            fprintf(stderr, "[cloudtrace] %s %.2f ms\n", name,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    }
};

inline TracePhase::TracePhase(CtHandle h, const char *n) : name(n), t0(std::chrono::steady_clock::now()), on(h && h->tune.TRACE) {}

static int flush(CtHandle h);
static int check_invariants(CtHandle h);
static void discard_ahead(CtHandle h);

#define NEED_NOFLUSH(h)                                \
    do {                                               \
        if (!(h)) {                                    \
            return fail(nullptr, CT_E_INVAL, "null handle"); \
        }                                              \
        if (hipSetDevice((h)->device) != hipSuccess) { \
            return fail((h), CT_E_HIP, "hipSetDevice(%d) failed", (h)->device); \
        }                                              \
        (void)hipGetLastError(); /* a stale error of another library in this thread (RCCL leaves them) is not ours */ \
    } while (0)

// Every entry point except the *_async ones first waits for the batches in flight.
#define NEED(h)                                        \
    do {                                               \
        NEED_NOFLUSH(h);                               \
        const int rc_flush_ = flush(h);                \
        if (rc_flush_ != CT_OK) {                      \
            return rc_flush_;                          \
        }                                              \
    } while (0)

static void v3_normalize_twice(const float in[3], float out[3])
{
    float v[3] = { in[0], in[1], in[2] };
    for (int pass = 0; pass < 2; pass++) { // installers.cpp:74-78 then SceneDescription.h:16
        const float inv = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] *= inv;
        v[1] *= inv;
        v[2] *= inv;
    }
    out[0] = v[0];
    out[1] = v[1];
    out[2] = v[2];
}

// Mie.cpp:8206-8243: phase / mean(phase), float32 running sum in index order.
static void mie_phase_texture(const float *raw, uint32_t n, std::vector<float> &out)
{
    out.resize(n);
    float average = 0;
    for (uint32_t i = 0; i < n; i++) {
        average += raw[i];
    }
    average /= (float)n;
    for (uint32_t i = 0; i < n; i++) {
        out[i] = raw[i] / average;
    }
}

// Mie.cpp:8245-8282: running sum of phase / sum(phase).
static void mie_integral_texture(const float *raw, uint32_t n, std::vector<float> &out)
{
    out.resize(n);
    float sum = 0;
    for (uint32_t i = 0; i < n; i++) {
        sum += raw[i];
    }
    float integral = 0;
    for (uint32_t i = 0; i < n; i++) {
        integral += raw[i] / sum;
        out[i] = integral;
    }
}

// guide[b] = #{ i : cdf[i] < b / 4096 }, b = 0..4096 (+1 pad): brackets the texel search of
// sample_cos_theta for every 24-bit random whose top 12 bits are b.
static void build_guide(const std::vector<float> &cdf, std::vector<uint16_t> &guide)
{
    guide.assign(kGuideN + 2, 0);
    uint32_t t = 0;
    for (uint32_t b = 0; b <= (uint32_t)kGuideN; b++) {
        const float edge = (float)b / (float)kGuideN;
        while (t < cdf.size() && cdf[t] < edge) {
            t++;
        }
        guide[b] = (uint16_t)t;
    }
    guide[kGuideN + 1] = guide[kGuideN];
}

static uint32_t morton2(uint32_t x, uint32_t y)
{
    auto spread = [](uint32_t v) {
        v &= 0xffffu;
        v = (v | (v << 8)) & 0x00ff00ffu;
        v = (v | (v << 4)) & 0x0f0f0f0fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

// Position of tile (x, y) along a Hilbert curve over a 2^15 x 2^15 grid (the standard xy2d): like the Morton index a
// locality-preserving order of the tiles, but without its jumps -- consecutive tiles are always neighbours.
static uint32_t hilbert2(uint32_t x, uint32_t y)
{
    const uint32_t n = 1u << 15;
    uint32_t d = 0;
    for (uint32_t s = n / 2; s > 0; s /= 2) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) {
                x = n - 1u - x;
                y = n - 1u - y;
            }
            const uint32_t t = x;
            x = y;
            y = t;
        }
    }
    return d;
}

template <typename T>
static hipError_t dmalloc(T **p, size_t count)
{
    return hipMalloc((void **)p, count * sizeof(T));
}

static void release(CtHandle h)
{
    if (!h) {
        return;
    }
    hipSetDevice(h->device);
    if (h->stream) {
        hipStreamSynchronize(h->stream);
    }
#ifdef CT_EXPERIMENTS
    if (h->vmm.va) {
        // the march bricks live in a reserved virtual range: unmap, release the physical chunks, free the range
        hipMemUnmap(h->vmm.va, h->vmm.size);
        for (auto &hd : h->vmm.handles) {
            hipMemRelease(hd);
        }
        hipMemAddressFree(h->vmm.va, h->vmm.size);
        h->vmm.va = nullptr;
        h->d_mbricks = nullptr;
    }
#endif
    void *ptrs[] = { h->d_density, h->d_inscatter, h->d_dbricks, h->d_ibricks, h->d_mbricks, h->d_tbricks, h->d_touched[0], h->d_touched[1], h->d_mrows, h->d_mcoarse, h->d_pyramid, h->d_mie, h->d_chopped, h->d_cdf,
                     h->d_guide, h->d_dist, h->d_dist_tmp, h->d_majorant, h->d_maj_cells, h->d_maj_codes, h->d_frame, h->d_mean, h->d_m2, h->d_screen, h->d_frames_all, h->cont[0], h->cont[1], h->left[0], h->left[1], h->d_cont_count, h->d_cont_total, h->d_primary, h->d_advance, h->d_pixels, h->d_cost, h->d_group_rank, h->d_group_order, h->d_job_group, h->d_job_sub, h->d_queue,
                     h->d_counters, h->d_colsum, h->d_avg, h->d_freeze, h->d_hit, h->d_cost_plane, h->d_timeline, h->pt.tasks, h->pt.primary, h->pt.frames, h->pt.pixels, h->pt.jg, h->pt.js };
    for (void *p : ptrs) {
        if (p) {
            hipFree(p);
        }
    }
    for (auto &e : h->ev) {
        if (e) {
            hipEventDestroy(e);
        }
    }
    for (auto &sl : h->slots) {
        if (sl.queue) {
            hipFree(sl.queue);
        }
        for (hipEvent_t e : { sl.ev_in, sl.ev_start, sl.ev_done, sl.ev_acc0, sl.ev_acc1 }) {
            if (e) {
                hipEventDestroy(e);
            }
        }
    }
    for (hipEvent_t e : { h->ev_flush0, h->ev_flush1 }) {
        if (e) {
            hipEventDestroy(e);
        }
    }
    if (h->freeze_host) {
        hipHostFree(h->freeze_host);
    }
    if (h->hit_host) {
        hipHostFree(h->hit_host);
    }
    for (uint32_t *p : { h->jobs_host_g, h->jobs_host_s }) {
        if (p) {
            hipHostFree(p);
        }
    }
    if (h->own_stream) {
        hipStreamDestroy(h->own_stream);
    }
    delete h;
}

#ifdef CT_EXPERIMENTS
// EXPERIMENTS BUILD ONLY (measured and rejected in round 4: a gather over memory mapped in 2-MiB pieces runs at 0.55 of the speed of
// the same gather over one hipMalloc, whatever the layout -- DESIGN.md 4.3 item 7b, profiles/r04f, r04g).
// CT_FLAG_VMM_BRICKS / CT_SPARSE=2 (BASELINE.json configs[4], "sparse brick-compressed density"): the dense march-brick array
// h->d_mbricks is replaced by a reserved VIRTUAL range of the same size and layout in which chunks of the virtual-memory
// granularity (2 MiB) SHARE memory wherever they may.  A chunk with a non-zero texel byte gets a copy of its own.  A chunk
// without one is described by its meta bytes alone, and a clearance may be rounded down without changing a result (the exact
// free-space skip gets shorter, nothing else): the clearances are quantised to {0, 4, 8, 16, 32, 64, 127} texels, and all
// chunks whose quantised bytes are equal are mapped onto ONE piece of memory.  For that to happen chunk boundaries must fall
// on brick-row boundaries, so this layout pads a brick row to a power of two bricks (create_impl).  The estimator's kernel is
// the dense one, unchanged -- same address arithmetic, one more level of sharing in the page tables -- and its results are
// identical (tests/test_gpu_parity.py: test_vmm_backed_march_bricks..., the knob test with CT_SPARSE=2; tests/test_configs.py).
static int vmm_back_mbricks(CtHandle h)
{
    const size_t dense_bytes = h->mbricks_dense_bytes;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    size_t gran = 0;
    HIPCHK(h, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t chunk = std::max<size_t>(gran, (size_t)2 << 20);
    if (chunk % 128 != 0 || (chunk & (chunk - 1)) != 0) {
        return fail(h, CT_E_HIP, "virtual-memory granularity %zu is not a power of two", chunk);
    }
    const size_t n_chunks = (dense_bytes + chunk - 1) / chunk;
    const size_t va_size = n_chunks * chunk;
    uint4 *d_class = nullptr;
    HIPCHK(h, dmalloc(&d_class, n_chunks));
    std::vector<uint4> cls(n_chunks);
    uint8_t *dense = h->d_mbricks;
    void *va = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    auto undo = [&]() {
        if (va) {
            hipMemUnmap(va, va_size);
            for (auto &hd : handles) {
                hipMemRelease(hd);
            }
            hipMemAddressFree(va, va_size);
        }
        hipFree(d_class);
    };
#define VMMCHK(expr)                                                                                       \
    do {                                                                                                   \
        const hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                            \
            undo();                                                                                        \
            return fail(h, CT_E_HIP, "virtual-memory bricks: %s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                                  \
    } while (0)
    VMMCHK(launch_mbrick_chunk_class(dense, (int64_t)dense_bytes, (int64_t)chunk, d_class, h->stream));
    VMMCHK(hipMemcpyAsync(cls.data(), d_class, n_chunks * sizeof(uint4), hipMemcpyDeviceToHost, h->stream));
    VMMCHK(hipStreamSynchronize(h->stream));
    VMMCHK(hipMemAddressReserve(&va, va_size, chunk, nullptr, 0));
    // owner[c] = the chunk whose memory chunk c uses (itself: memory of its own)
    std::vector<size_t> owner(n_chunks);
    std::vector<hipMemGenericAllocationHandle_t> of_chunk(n_chunks);
    std::map<uint64_t, size_t> first_with;
    size_t own = 0, with_data = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        const bool whole = (c + 1) * chunk <= dense_bytes;   // (the last, partial chunk always gets memory of its own)
        owner[c] = c;
        if (whole && cls[c].x == 0u) {
            const uint64_t key = (uint64_t)cls[c].z | (uint64_t)cls[c].w << 32;
            const auto it = first_with.find(key);
            if (it != first_with.end()) {
                owner[c] = it->second;
            } else {
                first_with[key] = c;
            }
        } else {
            with_data += 1;
        }
        void *at = (uint8_t *)va + c * chunk;
        if (owner[c] == c) {
            hipMemGenericAllocationHandle_t hd;
            VMMCHK(hipMemCreate(&hd, chunk, &prop, 0));
            handles.push_back(hd);
            of_chunk[c] = hd;
            own += 1;
        } else {
            of_chunk[c] = of_chunk[owner[c]];
        }
        VMMCHK(hipMemMap(at, chunk, 0, of_chunk[c], 0));
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    VMMCHK(hipMemSetAccess(va, va_size, &acc, 1));
    for (size_t c = 0; c < n_chunks; c++) {
        if (owner[c] != c) {
            continue;
        }
        uint8_t *at = (uint8_t *)va + c * chunk;
        const size_t n = std::min(chunk, dense_bytes - c * chunk);
        VMMCHK(hipMemcpyAsync(at, dense + c * chunk, n, hipMemcpyDeviceToDevice, h->stream));
        if (n < chunk) {
            VMMCHK(hipMemsetAsync(at + n, 0, chunk - n, h->stream));
        } else if (cls[c].x == 0u) {
            VMMCHK(launch_mbrick_chunk_quantize(at, (int64_t)chunk, h->stream));   // others are mapped onto these bytes
        }
    }
    VMMCHK(hipStreamSynchronize(h->stream));
#undef VMMCHK
    hipFree(d_class);
    HIPCHK(h, hipFree(dense));
    h->d_mbricks = (uint8_t *)va;
    h->vmm.va = va;
    h->vmm.size = va_size;
    h->vmm.chunk = chunk;
    h->vmm.handles = std::move(handles);
    h->vmm.real_chunks = with_data;
    h->vmm.shared_chunks = own - with_data;
    h->vmm.mapped_chunks = n_chunks - own;
    h->mbricks_bytes = own * chunk;
    if (h->tune.STATS) {
        fprintf(stderr, "[cloudtrace] march bricks, dense addressing with sparse backing: %zu chunks of %zu KiB, %zu with texels, %zu distinct "
                        "empty ones, %zu mapped onto those: %.3f GB behind %.3f GB of addresses\n", n_chunks, chunk >> 10, with_data, own - with_data,
                n_chunks - own, (double)(own * chunk) / 1e9, (double)va_size / 1e9);
    }
    return CT_OK;
}
#endif

static int create_impl(const CtScene *s, CtHandle h)
{
    const uint32_t nx = s->dims[0], ny = s->dims[1], nz = s->dims[2];
    const size_t texels = (size_t)nx * ny * nz;
    h->scene = *s;
    h->device = s->device;
    h->volume_bytes = texels;
    h->tune = CtTuning::from_env();

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        return fail(h, CT_E_NODEVICE, "no HIP device visible (libcloudtrace has no CPU fallback)");
    }
    if (s->device < 0 || s->device >= ndev) {
        return fail(h, CT_E_INVAL, "device %d out of range (0..%d)", s->device, ndev - 1);
    }
    HIPCHK(h, hipSetDevice(s->device));
    (void)hipGetLastError(); // (see NEED_NOFLUSH)
    HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    for (auto &e : h->ev) {
        HIPCHK(h, hipEventCreate(&e));
    }
    h->shape = persistent_shape(s->device, s->estimator == CT_EST_DELTA, h->tune.BLOCKS_PER_CU ? atoi(h->tune.BLOCKS_PER_CU.get()) : 0);
    if (const char *e = h->tune.DEBUG_INVARIANTS.get()) {
        h->debug_invariants = atoi(e) != 0;
    }
    h->shape.stats = h->tune.STATS.get() != nullptr || h->debug_invariants;
    if (const char *e = h->tune.EXCHANGE.get()) {
        h->exchange = s->estimator == CT_EST_DELTA ? std::min(2, std::max(0, atoi(e))) : 0;
    }
#ifdef CT_EXPERIMENTS
    if (h->exchange) {
        h->xshape = h->exchange == 2 ? wave_exchange_shape(s->device) : exchange_shape(s->device);
        h->xshape.stats = h->shape.stats;
    }
#else
    if (h->exchange) {
        // the product's library holds the product's kernels only; the measured-and-rejected ones live in the experiments build
        return fail(h, CT_E_INVAL, "CT_EXCHANGE needs the experiments build (python -m deepestscatter_amd.build --variant exp; "
                                   "CT_LIBRARY=libcloudtrace_exp.so)");
    }
#endif

    // ---- uniforms: VDBCloud::setupVolumeVariables (VDBCloud.cpp:98-111), Sun::init (Sun.cpp:13-18)
    DevScene &d = h->dev;
    const float fx = (float)nx, fy = (float)ny, fz = (float)nz;
    const float maxs = fmaxf(fmaxf(fx, fy), fz);
    d.nx = (int32_t)nx;
    d.ny = (int32_t)ny;
    d.nz = (int32_t)nz;
    d.bx = fx / maxs;
    d.by = fy / maxs;
    d.bz = fz / maxs;
    d.hx = d.bx + 0.01f;
    d.hy = d.by + 0.01f;
    d.hz = d.bz + 0.01f;
    d.tsx = maxs / fx;
    d.tsy = maxs / fy;
    d.tsz = maxs / fz;
    d.sx = (maxs / fx) * fx; // textureScale * N
    d.sy = (maxs / fy) * fy;
    d.sz = (maxs / fz) * fz;
    d.density_multiplier = s->cloud_size_m / s->mean_free_path_m;
    d.sample_step = s->sample_step;
    float l[3] = { s->light_direction[0], s->light_direction[1], s->light_direction[2] };
    if (!(s->flags & CT_FLAG_LIGHT_NORMALIZED)) {
        v3_normalize_twice(s->light_direction, l);
    }
    d.nlx = -l[0];
    d.nly = -l[1];
    d.nlz = -l[2];
    d.lr = s->light_color[0] * s->light_intensity;
    d.lg = s->light_color[1] * s->light_intensity;
    d.lb = s->light_color[2] * s->light_intensity;
    {
        // cloud.cuh:148-151 evaluated once on the host, in float like the device code.
        const float sunAngularRadiusDeg = 0.53f / 2;
        const float sphereArea = 4 * kPi;
        const float sunArea = 2 * kPi * (1 - cosf(sunAngularRadiusDeg * kPi / 180.0f));
        d.sun_ratio = sunArea / sphereArea;
    }
    d.width = s->width;
    d.height = s->height;
    d.max_depth = s->max_depth;
    d.mode = s->mode;
    d.tiles_x = (s->width + kTile - 1) / kTile;
    d.tiles_y = (s->height + kTile - 1) / kTile;
    // (re-swept on the final kernel: 8 instead of 16 is worth +1.4 % at 512^3 and +5 % at 256^3; 16 stays best at 1024^3)
    d.regen_min = 8;
    // measured: running the scatter phase as soon as any lane needs it beats waiting for a fuller
    // phase (927 vs 877 Msamples/s); a waiting lane is latency added to a serial path
    d.scatter_num = 0;
    d.scatter_den = 1;
    d.scatter_min = 1;
    // measured (profiles/README.md, burst sweep): 1207 Msamples/s without bursts, 1453 with 8/32/32
    // (re-swept on the final kernel, 512 spp per launch: burst_scatter 48 -> 40 is worth +3 %, 2871 -> 2963)
    d.march_burst = 8;
    d.burst_scatter = 40;
    d.burst_idle = 32;
    if (s->estimator == CT_EST_DELTA) {
        // a tracking visit ends in a real collision 6 times out of 10: short bursts and an early refill
        // (sweep at 512^3/1024^2: 8/16 -> 2300 Msamples/s, 1/16 -> 2700, 1/4 -> 2870, 3/4 -> 3460)
        // (round 2, on the final kernel: 3/4 with scatter phases as soon as one lane waits 4200; the scatter phase held back
        // until 16 lanes wait 4241; that with bursts of 2: 4295 -- profiles/r02e/delta_scatter_min_sweep.log)
        d.march_burst = 2;
        d.burst_scatter = 48;
        d.regen_min = 4;
        d.scatter_min = 12;   // (16 until the tracking burst followed the scatter phase in one iteration; 8 / 12 / 16 / 24 / 32: 5284 / 5290 / 5273 / 5168 / 5104, profiles/r04h, r04i)
    }
    // measured (profiles/README.md): regional queues raise the L2 hit rate from 67 % to 77 % but not
    // the speed (the kernel is bound by the L1 gather rate and by instruction issue, not by L2
    // misses), and any imbalance between regions costs more than that: off unless asked for
    h->queues_enabled = false;
    // ... unless the volume is far larger than the caches: at 1024^3 (2.7 GB of march bricks) the regional
    // queues are worth +4.5 % (1824 -> 1906 Msamples/s at 2048^2); at 512^3 they cost 0.5-2 %
    if ((uint64_t)nx * ny * nz >= 768ull * 768ull * 768ull) {
        h->queues_enabled = true;
        d.burst_scatter = 48; // there every fetch is an L2 miss and longer bursts pay: 1914 vs 1855 Msamples/s at 1024^3
        if (s->estimator == CT_EST_MARCH) {
            d.regen_min = 16;
        }
    }
    if (const char *e = h->tune.JOB_MAX.get()) {
        h->job_max = (uint32_t)std::min(4096, std::max(1, atoi(e)));
    }
    if (s->estimator == CT_EST_DELTA) {
        h->job_work = 48.f;     // (its cost unit is a bounce; 16 measured 0.4 % slower there)
        // measured at 512^3 / 1024^2 (profiles/r04b, r04c): 0 -> 5020 Msamples/s, 1 -> 5125 (+2.1 %: 2.8 % fewer vector
        // instructions, the NEE miss off the bounce's critical path), 2 -> 5100 (+1.6 % although the launch moves 32 % fewer
        // bytes: the kernel is bound by instruction issue, not by line fills -- DESIGN.md 4.2 "Round 4")
        d.delta_nee = 1u;
        // ... and at 1024^3 / 2048^2, where nine fetches in ten miss L2 and the volume is four times the Infinity Cache, the
        // twin bricks win: 0 -> 3841, 1 -> 3914, 2 -> 4562 Msamples/s (+19 %; profiles/r04d)
        if ((uint64_t)nx * ny * nz >= 768ull * 768ull * 768ull) {
            d.delta_nee = 2u;
        }
        if (const char *e = h->tune.DELTA_NEE.get()) {   // render_delta_kernel<.., NEE>: where a collision's two lookups come from
            d.delta_nee = (uint32_t)std::min(2, std::max(0, atoi(e)));
        }
    }
    if (const char *e = h->tune.JOB_WORK.get()) {
        h->job_work = (float)std::max(1.0, atof(e));
    }
    if (const char *e = h->tune.NO_ADVANCE.get()) {
        h->no_advance = atoi(e) != 0;
    }
    if (const char *e = h->tune.XCD_QUEUES.get()) {
        h->queues_enabled = atoi(e) != 0;
    }
    d.burst_march_min = 1;
    if (const char *e = h->tune.BURST_MARCH_MIN.get()) {
        d.burst_march_min = (uint32_t)std::min(64, std::max(1, atoi(e)));
    }
    d.hint_period = 64;
    if (const char *e = h->tune.HINT_PERIOD.get()) {
        const int v = atoi(e);
        d.hint_period = (v > 0 && (v & (v - 1)) == 0) ? (uint32_t)v : 0u;
    }
    d.tail_burst = 8;
    if (const char *e = h->tune.TAIL_BURST.get()) {
        d.tail_burst = (uint32_t)std::min(1024, std::max(1, atoi(e)));
    }
    if (const char *e = h->tune.XCD_REGIONS.get()) {
        h->regions = (uint32_t)std::min(65536, std::max(1, atoi(e)));
    }
    if (const char *e = h->tune.SHARED_DEPTH.get()) {
        h->shared_depth = (float)atof(e);
    }
    if (const char *e = h->tune.MARCH_BURST.get()) {
        d.march_burst = (uint32_t)std::min(1024, std::max(1, atoi(e)));
    }
    if (const char *e = h->tune.BURST_SCATTER.get()) {
        d.burst_scatter = (uint32_t)std::min(65, std::max(1, atoi(e)));
    }
    if (const char *e = h->tune.BURST_IDLE.get()) {
        d.burst_idle = (uint32_t)std::min(65, std::max(1, atoi(e)));
    }
    if (const char *e = h->tune.SCATTER_MIN.get()) {
        d.scatter_min = (uint32_t)std::min(64, std::max(1, atoi(e)));
    }
    // tuning knobs for experiments (schedule only; results never change)
    if (const char *e = h->tune.REGEN_MIN.get()) {
        d.regen_min = (uint32_t)std::min(64, std::max(1, atoi(e)));
    }
    if (const char *e = h->tune.SCATTER_RATIO.get()) { // "num/den"
        unsigned a = 1, b = 1;
        if (sscanf(e, "%u/%u", &a, &b) == 2 && b > 0 && a < 1000 && b < 1000) {
            d.scatter_num = a;
            d.scatter_den = b;
        }
    }

    // apron: farthest texel a marching path can address: slack box + the steps it fetches
    // speculatively in one scheduler visit (kSpec in ct_kernels.hip, 8 allowed for here)
    const int apron = (int)ceilf((0.01f + 8.0f * s->sample_step) * fmaxf(fmaxf(d.sx, d.sy), d.sz) + 0.5f) + 1;
    if (apron > 160) {
        return fail(h, CT_E_INVAL, "sample_step %g too coarse for a %u^3 volume", (double)s->sample_step,
                    (unsigned)maxs);
    }
    // brick coordinates = (texel index + bias) / 4, bias a multiple of 4 covering the apron
    const int bbias = ((apron + 3) / 4) * 4;
    const int64_t bgx = ((int64_t)nx + 2 * bbias + 3) / 4 + 1, bgy = ((int64_t)ny + 2 * bbias + 3) / 4 + 1,
                  bgz = ((int64_t)nz + 2 * bbias + 3) / 4 + 1;
    if (bgx * bgy * bgz >= (1ll << 31) || bgx * bgy >= (1ll << 24)) {
        return fail(h, CT_E_INVAL, "volume too large for 32-bit brick indices");
    }
    d.brick_bias = bbias;
    d.nee_cache = (bgx * bgy * bgz < (1ll << 25)) ? 1u : 0u;
    if (const char *e = h->tune.NEE_CACHE.get()) {
        d.nee_cache = (atoi(e) != 0 && d.nee_cache) ? 1u : 0u;
    }
    if (const char *e = h->tune.POINT_ORDER.get()) {
        h->point_order = atoi(e) != 0;
    }
    d.brick_gx = (int32_t)bgx;
    d.brick_gxy = (int32_t)(bgx * bgy);
    d.brick_gy = (int32_t)bgy;
    d.brick_gz = (int32_t)bgz;
    const size_t brick_bytes = (size_t)(bgx * bgy * bgz) * 128;

    // ---- Mie textures (Mie.cpp:8206-8297) + guide table
    std::vector<float> mie_tex, chopped_tex, cdf_tex;
    std::vector<uint16_t> guide;
    mie_phase_texture(s->mie_host, s->mie_count, mie_tex);
    mie_phase_texture(s->chopped_mie_host, s->mie_count, chopped_tex);
    mie_integral_texture(s->chopped_mie_host, s->mie_count, cdf_tex);
    build_guide(cdf_tex, guide);
    HIPCHK(h, dmalloc(&h->d_mie, kMieN));
    HIPCHK(h, dmalloc(&h->d_chopped, kMieN));
    HIPCHK(h, dmalloc(&h->d_cdf, kMieN));
    HIPCHK(h, dmalloc(&h->d_guide, kGuideN + 2));
    HIPCHK(h, hipMemcpyAsync(h->d_mie, mie_tex.data(), kMieN * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_chopped, chopped_tex.data(), kMieN * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_cdf, cdf_tex.data(), kMieN * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_guide, guide.data(), (kGuideN + 2) * sizeof(uint16_t), hipMemcpyHostToDevice,
                             h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream)); // the vectors die at scope exit
    d.mie = h->d_mie;
    d.chopped = h->d_chopped;
    d.cdf = h->d_cdf;
    d.guide = h->d_guide;

    // ---- density -> corner cells; shadow volume (VDBCloud::InitInScatter) -> corner cells
    HIPCHK(h, dmalloc(&h->d_density, texels));
    HIPCHK(h, dmalloc(&h->d_inscatter, texels));
    HIPCHK(h, dmalloc(&h->d_dbricks, brick_bytes));
    HIPCHK(h, dmalloc(&h->d_ibricks, brick_bytes));
    HIPCHK(h, hipMemcpyAsync(h->d_density, s->density_host, texels, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_build_bricks(h->d_density, nx, ny, nz, bbias, (int)bgx, (int)bgy, (int)bgz, h->d_dbricks, h->stream));
    d.dbricks = h->d_dbricks;
    d.ibricks = h->d_ibricks;
    {
        // free-space distance field on the brick grid itself, embedded as the bricks' meta byte
        const size_t nb = (size_t)(bgx * bgy * bgz);
        HIPCHK(h, dmalloc(&h->d_dist, nb));
        HIPCHK(h, dmalloc(&h->d_dist_tmp, nb));
        HIPCHK(h, dmalloc(&h->d_majorant, nb));
        HIPCHK(h, launch_build_dist(h->d_density, nx, ny, nz, bbias, (int)bgx, (int)bgy, (int)bgz, h->d_dist,
                                    h->d_dist_tmp, h->d_majorant, h->stream));
        HIPCHK(h, launch_brick_meta(h->d_dist, h->d_majorant, nx, ny, nz, bbias, (int)bgx, (int)bgy, (int)bgz,
                                    h->d_dbricks, h->stream));
    }
    if (s->estimator == CT_EST_DELTA) {
        // Majorant cells of the DELTA grid (orc_majorant_grid / orc_build_majorants in the oracle, same rule).  A VIRTUAL grid of
        // cubic cells of C texels covers [-bias, n + bias); stored -- and copied to LDS by every block -- is only the box of cells
        // that can have a non-zero majorant: per axis the cells whose clamped read interval [clamp(lo - 1), clamp(lo + C + 1)] meets
        // the bounding interval of the non-zero texels.  C = the smallest value >= 4 whose box fits the kernel's LDS array: the
        // benchmark cloud at 512^3 gets 10-texel cells where a grid over the whole volume allowed 16 (round 4).
        const int32_t n[3] = { (int32_t)nx, (int32_t)ny, (int32_t)nz };
        int32_t lo[3] = { n[0], n[1], n[2] }, hi[3] = { -1, -1, -1 };
        {
            const uint8_t *t = s->density_host;
            for (int32_t z = 0; z < n[2]; z++) {
                for (int32_t y = 0; y < n[1]; y++) {
                    const uint8_t *row = t + ((size_t)z * n[1] + y) * n[0];
                    int32_t x0 = 0, x1 = n[0] - 1;
                    while (x0 <= x1 && row[x0] == 0) {
                        x0++;
                    }
                    if (x0 > x1) {
                        continue;
                    }
                    while (row[x1] == 0) {
                        x1--;
                    }
                    lo[0] = std::min(lo[0], x0);
                    hi[0] = std::max(hi[0], x1);
                    lo[1] = std::min(lo[1], y);
                    hi[1] = std::max(hi[1], y);
                    lo[2] = std::min(lo[2], z);
                    hi[2] = std::max(hi[2], z);
                }
            }
        }
        int32_t C = 4, origin[3] = { 0, 0, 0 }, stored[3] = { 1, 1, 1 }, virt[3] = { 1, 1, 1 };
        for (;; C++) {
            int64_t cells = 1;
            for (int a = 0; a < 3; a++) {
                const int32_t v = (n[a] + 2 * bbias + C - 1) / C;
                int32_t c0 = v, c1 = -1;
                for (int32_t c = 0; c < v; c++) {
                    const int32_t r0 = std::min(std::max(C * c - bbias - 1, 0), n[a] - 1), r1 = std::min(std::max(C * c - bbias + C + 1, 0), n[a] - 1);
                    if (r0 <= hi[a] && r1 >= lo[a]) {
                        c0 = std::min(c0, c);
                        c1 = c;
                    }
                }
                if (c1 < c0) {   // an empty volume: one stored cell, whose majorant will be 0
                    c0 = c1 = 0;
                }
                origin[a] = c0;
                stored[a] = c1 - c0 + 1;
                virt[a] = v;
                cells *= (int64_t)stored[a];
            }
            if (cells <= kMajCellsMax) {
                break;
            }
        }
        // x / C as (x * div) >> 20 for every texel index of the grid (dda_begin): checked, not argued
        const int32_t div = (int32_t)(((1u << 20) + (uint32_t)C - 1u) / (uint32_t)C);
        for (int32_t x = 0; x < (std::max({ virt[0], virt[1], virt[2] }) + 1) * C; x++) {
            if ((int32_t)(((uint32_t)x * (uint32_t)div) >> 20) != x / C || (uint64_t)x * (uint64_t)div >= (1ull << 32)) {
                return fail(h, CT_E_INVAL, "volume too large for the majorant grid's index arithmetic");
            }
        }
        const size_t cells = (size_t)stored[0] * stored[1] * stored[2];
        HIPCHK(h, dmalloc(&h->d_maj_cells, (cells + 3) & ~(size_t)3)); // the kernel copies whole words
        HIPCHK(h, hipMemsetAsync(h->d_maj_cells, 0, (cells + 3) & ~(size_t)3, h->stream));
        HIPCHK(h, dmalloc(&h->d_maj_codes, (cells + 3) & ~(size_t)3));
        HIPCHK(h, hipMemsetAsync(h->d_maj_codes, 0, (cells + 3) & ~(size_t)3, h->stream));
        HIPCHK(h, launch_majorant_cells(h->d_density, nx, ny, nz, bbias, C, origin, stored[0], stored[1], stored[2], h->d_maj_cells,
                                        h->d_maj_codes, h->stream));
        d.maj_cells = h->d_maj_cells;
        d.maj_codes = h->d_maj_codes;
        d.mc_cell = C;
        d.mc_div = div;
        d.mc_gx = stored[0];
        d.mc_gy = stored[1];
        d.mc_gz = stored[2];
        d.mc_x0 = origin[0];
        d.mc_y0 = origin[1];
        d.mc_z0 = origin[2];
        d.mc_vx = virt[0];
        d.mc_vy = virt[1];
        d.mc_vz = virt[2];
        // (a real collision has a non-zero texel in its footprint, i.e. its texel coordinate is within one texel of the non-zero
        // texels' bounding box: two texels of margin put it inside the box whatever the rounding; CT_DELTA_INTERIOR=0: always test)
        d.delta_interior = 1u;
        // (and a tentative collision lies in a stored cell -- to a hundredth of a texel: the error of the crossings' sums -- so
        // when the stored box, grown by one texel, is inside the brick grid its texel index needs no clamp)
        const int32_t bg[3] = { d.brick_gx, d.brick_gy, d.brick_gz };
        for (int a = 0; a < 3; a++) {
            if (lo[a] < 2 || hi[a] > n[a] - 3 || C * origin[a] < 1 || C * (origin[a] + stored[a]) > 4 * bg[a] - 1) {
                d.delta_interior = 0u;
            }
        }
        if (const char *e = h->tune.DELTA_INTERIOR.get()) {
            d.delta_interior = (atoi(e) != 0 && d.delta_interior) ? 1u : 0u;
        }
    }
    // which of the volume's six boundary layers are empty (launch_inscatter: a march that leaves through one of those is over)
    uint32_t zero_faces = 0;
    {
        const uint8_t *t = s->density_host;
        const auto layer_is_zero = [&](int axis, uint32_t at) {
            const uint32_t n[3] = { nx, ny, nz };
            const uint32_t u_axis = (axis + 1) % 3, v_axis = (axis + 2) % 3;
            for (uint32_t v = 0; v < n[v_axis]; v++) {
                for (uint32_t u = 0; u < n[u_axis]; u++) {
                    uint32_t c[3];
                    c[axis] = at;
                    c[u_axis] = u;
                    c[v_axis] = v;
                    if (t[((size_t)c[2] * ny + c[1]) * nx + c[0]] != 0) {
                        return false;
                    }
                }
            }
            return true;
        };
        const uint32_t last[3] = { nx - 1, ny - 1, nz - 1 };
        for (int axis = 0; axis < 3; axis++) {
            zero_faces |= (layer_is_zero(axis, 0) ? 1u : 0u) << (2 * axis);
            zero_faces |= (layer_is_zero(axis, last[axis]) ? 1u : 0u) << (2 * axis + 1);
        }
    }
    {
        // march bricks (3x4x4 texels, one meta byte per row; DevScene::mbricks).  The two scratch
        // volumes of the distance transform are borrowed from the shadow-volume brick array's
        // neighbourhood: plain temporaries, freed when the build has run.
        const int mbias = ((apron + 2) / 3) * 3;
        int64_t mgx = ((int64_t)nx + 2 * mbias + 2) / 3 + 1;
        bool vmm = (s->flags & CT_FLAG_VMM_BRICKS) != 0;
        bool pad_only = false;   // CT_SPARSE=3 (A/B): the padded rows of the virtual-memory layout in ordinary memory
        if (const char *env = h->tune.SPARSE.get()) {
            vmm = atoi(env) == 2 || atoi(env) == 4;
            pad_only = atoi(env) == 3;
        }
        vmm = vmm && s->estimator == CT_EST_MARCH;
#ifndef CT_EXPERIMENTS
        if (vmm || pad_only) {
            return fail(h, CT_E_INVAL, "march bricks behind virtual memory (CT_FLAG_VMM_BRICKS, CT_SPARSE=2..4) were measured and rejected "
                                       "(DESIGN.md 4.3 item 7b): they exist in the experiments build only (build --variant exp; CT_LIBRARY=libcloudtrace_exp.so)");
        }
#endif
        const bool no_pad = h->tune.SPARSE.get() && atoi(h->tune.SPARSE.get()) == 4;   // (A/B: the mapping without the padded rows)
        if ((vmm && !no_pad) || pad_only) {
            // dense addressing with sparse backing (vmm_back_mbricks): chunks of 2 MiB = 2^14 bricks must hold whole brick rows for
            // empty chunks to have equal bytes, so a row is padded to a power of two bricks (addresses cost nothing)
            int64_t p2 = 1;
            while (p2 < mgx) {
                p2 <<= 1;
            }
            mgx = p2;
        }
        if (mgx * bgy * bgz >= (1ll << 31) || mgx * bgy >= (1ll << 24) || (int64_t)nx + 2 * mbias >= (1 << 17)) {
            return fail(h, CT_E_INVAL, "volume too large for 32-bit brick indices");
        }
        uint8_t *tmp_a = nullptr, *tmp_b = nullptr;
        const size_t dense_bytes = (size_t)(mgx * bgy * bgz) * 128;
        h->mbricks_dense_bytes = h->mbricks_bytes = dense_bytes;
        HIPCHK(h, dmalloc(&h->d_mbricks, dense_bytes));
        HIPCHK(h, dmalloc(&tmp_a, texels));
        if (hipMalloc(&tmp_b, texels) != hipSuccess) {
            hipFree(tmp_a);
            return fail(h, CT_E_NOMEM, "out of device memory (distance transform scratch)");
        }
        hipError_t e = launch_build_mbricks(h->d_density, nx, ny, nz, mbias, bbias, (int)mgx, (int)bgy, (int)bgz,
                                            tmp_a, tmp_b, h->d_mbricks, h->stream);
        if (e == hipSuccess) {
            // the shadow volume (VDBCloud::InitInScatter): its march towards the sun walks the (dense) march bricks
            DevScene walk = d;
            walk.mbricks = h->d_mbricks;
            walk.m_bias_x = mbias;
            walk.m_gx = (int32_t)mgx;
            walk.m_gxy = (int32_t)(mgx * bgy);
            e = launch_inscatter(walk, h->d_inscatter, zero_faces, h->stream);
        }
        if (e == hipSuccess) {
            e = launch_build_bricks(h->d_inscatter, nx, ny, nz, bbias, (int)bgx, (int)bgy, (int)bgz, h->d_ibricks, h->stream);
        }
        if (e == hipSuccess && s->estimator == CT_EST_DELTA && d.delta_nee == 2u) {
            // twin bricks (DevScene::tbricks): density and shadow volume of 3^3 base texels in one line, over the texel range
            // of the apron bricks (which is that of the majorant cells: every position a flight can reach)
            const int tbias = ((bbias + 2) / 3) * 3;
            const int64_t tgx = ((int64_t)nx + bbias + tbias + 2) / 3 + 1, tgy = ((int64_t)ny + bbias + tbias + 2) / 3 + 1,
                          tgz = ((int64_t)nz + bbias + tbias + 2) / 3 + 1;
            if (tgx * tgy * tgz >= (1ll << 31) || tgx * tgy >= (1ll << 24) || (int64_t)std::max({ nx, ny, nz }) + 2 * tbias >= (1 << 15)) {
                hipFree(tmp_a);
                hipFree(tmp_b);
                return fail(h, CT_E_INVAL, "volume too large for 32-bit brick indices");
            }
            if (hipMalloc(&h->d_tbricks, (size_t)(tgx * tgy * tgz) * 128) != hipSuccess) {
                hipFree(tmp_a);
                hipFree(tmp_b);
                return fail(h, CT_E_NOMEM, "out of device memory (twin bricks)");
            }
            e = launch_build_twin_bricks(h->d_density, h->d_inscatter, nx, ny, nz, tbias, (int)tgx, (int)tgy, (int)tgz, h->d_tbricks, h->stream);
            d.tbricks = h->d_tbricks;
            d.t_bias = tbias;
            d.t_gx = (int32_t)tgx;
            d.t_gy = (int32_t)tgy;
            d.t_gz = (int32_t)tgz;
            // (delta_interior's second half for this layout: the stored cells, grown by a texel, inside the TWIN grid)
            const int32_t tg[3] = { d.t_gx, d.t_gy, d.t_gz }, o[3] = { d.mc_x0, d.mc_y0, d.mc_z0 }, st[3] = { d.mc_gx, d.mc_gy, d.mc_gz };
            for (int a = 0; a < 3; a++) {
                if (d.mc_cell * o[a] - bbias - 1 < -tbias || d.mc_cell * (o[a] + st[a]) - bbias > 3 * tg[a] - 1 - tbias) {
                    d.delta_interior = 0u;
                }
            }
        }
        const hipError_t e2 = hipStreamSynchronize(h->stream);
        // Sparse storage (BASELINE.json configs[4]: "1024^3 sparse brick-compressed density"): CT_FLAG_SPARSE_BRICKS, or
        // CT_SPARSE=0/1 in the environment.  Not the default at any size: measured at 1024^3 / 2048^2 the stored bricks
        // shrink from 3.41 GB to 0.36 GB, but the march runs 20 % slower (1670 vs 2083 Msamples/s,
        // profiles/r02c/ab_sparse_1024.log) -- the row-extent lookup is one more dependent load in a gather that is bound
        // by latency and by the L1's address rate, and only 5 % of the fetches land outside the stored extents, so there
        // are few line fills to save.  With 288 GB of HBM the dense array is the faster choice; this is the option for
        // when capacity matters.
        bool sparse = (s->flags & CT_FLAG_SPARSE_BRICKS) != 0;
        if (const char *env = h->tune.SPARSE.get()) {
            sparse = atoi(env) == 1;   // (2 = dense addressing with sparse backing, below)
        }
        int src = CT_OK;
        if (e == hipSuccess && e2 == hipSuccess && sparse && s->estimator == CT_EST_MARCH) {
            src = [&]() -> int {
                const size_t rows = (size_t)(bgy * bgz);
                uint32_t *d_x0 = nullptr, *d_x1 = nullptr;
                uint8_t *compact = nullptr;
                auto cleanup = [&]() {
                    for (void *q : { (void *)d_x0, (void *)d_x1 }) {
                        if (q) {
                            hipFree(q);
                        }
                    }
                };
                auto body = [&]() -> int {
                    HIPCHK(h, dmalloc(&d_x0, rows));
                    HIPCHK(h, dmalloc(&d_x1, rows));
                    HIPCHK(h, hipMemsetAsync(d_x0, 0xff, rows * sizeof(uint32_t), h->stream));
                    HIPCHK(h, hipMemsetAsync(d_x1, 0, rows * sizeof(uint32_t), h->stream));
                    HIPCHK(h, launch_mbrick_extent(h->d_mbricks, (int)mgx, (int)bgy, (int)bgz, d_x0, d_x1, h->stream));
                    std::vector<uint32_t> x0(rows), x1(rows);
                    HIPCHK(h, hipMemcpyAsync(x0.data(), d_x0, rows * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(h, hipMemcpyAsync(x1.data(), d_x1, rows * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    std::vector<uint2> ri(rows);
                    uint64_t lines = 0;
                    for (size_t r = 0; r < rows; r++) {
                        const uint32_t cnt = x1[r] > x0[r] ? x1[r] - x0[r] : 0u;
                        ri[r] = make_uint2((uint32_t)lines, cnt ? (x0[r] | (cnt << 16)) : 0u);
                        lines += cnt;
                    }
                    if (lines >= (1ull << 32)) {
                        return fail(h, CT_E_INVAL, "volume too large for 32-bit brick indices");
                    }
                    HIPCHK(h, dmalloc(&h->d_mrows, rows));
                    HIPCHK(h, hipMemcpyAsync(h->d_mrows, ri.data(), rows * sizeof(uint2), hipMemcpyHostToDevice, h->stream));
                    HIPCHK(h, dmalloc(&compact, std::max<size_t>((size_t)lines, 1) * 128));
                    HIPCHK(h, launch_mbrick_compact(h->d_mbricks, (int)mgx, (int)bgy, (int)bgz, h->d_mrows, compact, h->stream));
                    // clearance of the cells outside the extents: cubic cells of 8 texels over the brick grid's texel range
                    const int cshift = 3;
                    const int64_t cgx = (4 * bgx + 7) >> cshift, cgy = (4 * bgy + 7) >> cshift, cgz = (4 * bgz + 7) >> cshift;
                    HIPCHK(h, dmalloc(&h->d_mcoarse, (size_t)(cgx * cgy * cgz)));
                    HIPCHK(h, launch_coarse_clearance(tmp_b, nx, ny, nz, bbias, cshift, (int)cgx, (int)cgy, (int)cgz, h->d_mcoarse,
                                                      h->stream));
                    HIPCHK(h, hipStreamSynchronize(h->stream)); // `ri` dies at scope exit; the dense array is freed below
                    HIPCHK(h, hipFree(h->d_mbricks));
                    h->d_mbricks = compact;
                    compact = nullptr;
                    h->mbricks_bytes = (size_t)lines * 128;
                    d.m_rows = h->d_mrows;
                    d.m_coarse = h->d_mcoarse;
                    d.m_cshift = cshift;
                    d.m_cgx = (int32_t)cgx;
                    d.m_cgxy = (int32_t)(cgx * cgy);
                    return CT_OK;
                };
                const int rc = body();
                if (rc != CT_OK) {
                    hipStreamSynchronize(h->stream);
                    if (compact) {
                        hipFree(compact);
                    }
                }
                cleanup();
                return rc;
            }();
        }
        hipFree(tmp_a);
        hipFree(tmp_b);
        HIPCHK(h, e);
        HIPCHK(h, e2);
        if (src != CT_OK) {
            return src;
        }
#ifdef CT_EXPERIMENTS
        if (vmm && !sparse) {
            const int rc = vmm_back_mbricks(h);
            if (rc != CT_OK) {
                return rc;
            }
        }
#endif
        d.mbricks = h->d_mbricks;
        d.m_bias_x = mbias;
        d.m_gx = (int32_t)mgx;
        d.m_gxy = (int32_t)(mgx * bgy);
    }

    // ---- Camera::init buffers (Camera.cpp:45-48) + reset (:77-86)
    const size_t pixels = (size_t)s->width * s->height;
    HIPCHK(h, dmalloc(&h->d_frame, pixels));
    HIPCHK(h, dmalloc(&h->d_mean, pixels));
    HIPCHK(h, dmalloc(&h->d_m2, pixels));
    HIPCHK(h, dmalloc(&h->d_screen, pixels));
    HIPCHK(h, dmalloc(&h->d_colsum, s->width));
    HIPCHK(h, dmalloc(&h->d_avg, 2)); // average luminance + the fused tonemap kernel's grid barrier
    HIPCHK(h, dmalloc(&h->d_queue, kQueueWords));
    HIPCHK(h, dmalloc(&h->d_cont_count, 6));
    HIPCHK(h, hipMemsetAsync(h->d_cont_count, 0, 6 * sizeof(uint32_t), h->stream));
    // a lane suspends at most one path per launch, and a launch resumes up to 64 paths per wave: the same number
    h->cont_capacity = (size_t)h->shape.blocks * h->shape.threads;
    HIPCHK(h, hipEventCreate(&h->ev_flush0));
    HIPCHK(h, hipEventCreate(&h->ev_flush1));
    if (const char *e = h->tune.CONTINUATION.get()) {
        h->continuation = atoi(e) != 0;
    }
    if (h->exchange) {
        h->continuation = false;   // (the exchange kernels run every path to its end)
    }
    if (const char *e = h->tune.CHUNK_INTERLEAVE.get()) {
        h->chunk_interleave = atoi(e) != 0;
    }
    if (const char *e = h->tune.TILE_ORDER.get()) {
        h->tile_hilbert = strcmp(e, "hilbert") == 0;
    }
    if (const char *e = h->tune.CHUNK_MORTON.get()) {
        h->chunk_morton = atoi(e) != 0;
    }
    if (const char *e = h->tune.SERPENTINE.get()) {
        h->serpentine = atoi(e) != 0;
    }
    if (const char *e = h->tune.HAND_ON_JOBS.get()) {
        h->hand_on_jobs = atoi(e) != 0;
    }
    if (const char *e = h->tune.MAX_AGE.get()) {
        h->max_age_override = std::min(CtHandle_::kMaxRegions - 1, std::max(0, atoi(e)));
    }
    if (h->tune.TIMELINE.get()) {
        const size_t waves = (size_t)h->shape.blocks * h->shape.threads / 64u;
        HIPCHK(h, dmalloc(&h->d_timeline, 4 * waves));
        HIPCHK(h, hipMemsetAsync(h->d_timeline, 0, 4 * waves * sizeof(unsigned long long), h->stream));
    }
    if (const char *e = h->tune.RENDER_AHEAD.get()) {
        h->ahead = (uint32_t)std::min(65535, std::max(0, atoi(e)));
    }
    HIPCHK(h, dmalloc(&h->d_cont_total, 1));
    HIPCHK(h, hipMemsetAsync(h->d_cont_total, 0, sizeof(unsigned long long), h->stream));
    for (auto &c : h->cont) {
        HIPCHK(h, dmalloc(&c, h->cont_capacity * (s->estimator == CT_EST_DELTA ? kContWordsDelta : kContWords)));
    }
    h->left_capacity = h->cont_capacity / 64;   // a wave hands on at most the one job it is working on
    for (auto &c : h->left) {
        HIPCHK(h, dmalloc(&c, h->left_capacity * kLeftWords));
    }
    for (auto &sl : h->slots) {
        HIPCHK(h, dmalloc(&sl.queue, kQueueWords));
        for (hipEvent_t *e : { &sl.ev_in, &sl.ev_start, &sl.ev_done, &sl.ev_acc0, &sl.ev_acc1 }) {
            HIPCHK(h, hipEventCreate(e));
        }
    }
    HIPCHK(h, dmalloc(&h->d_counters, kCounterCount + 1 + kStatCount));
    HIPCHK(h, dmalloc(&h->d_freeze, 8));
    HIPCHK(h, hipMemsetAsync(h->d_freeze, 0, 8 * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipHostMalloc((void **)&h->freeze_host, 4 * sizeof(uint32_t), hipHostMallocDefault));
    memset(h->freeze_host, 0, 4 * sizeof(uint32_t));
    HIPCHK(h, hipMemsetAsync(h->d_frame, 0, pixels * sizeof(float4), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_mean, 0, pixels * sizeof(float4), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_m2, 0, pixels * sizeof(float4), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_screen, 0, pixels * sizeof(uchar4), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_counters, 0, (kCounterCount + 1 + kStatCount) * sizeof(unsigned long long), h->stream));

    HIPCHK(h, dmalloc(&h->d_primary, 2 * pixels));
    HIPCHK(h, dmalloc(&h->d_advance, (s->estimator == CT_EST_DELTA ? 4 : 1) * pixels));
    HIPCHK(h, hipStreamSynchronize(h->stream));

    // ---- default pose: Camera.cpp:37-39 through sutil::calculateCameraVariables
    const float eye[3] = { 2.5f, -0.4f, 0.f }, lookat[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 };
    float U[3], V[3], W[3];
    ct_calculate_camera_variables(eye, lookat, up, 30.0f, (float)s->width / (float)s->height, U, V, W);
    return ct_set_camera(h, eye, U, V, W);
}

extern "C" int ct_create(const CtScene *s, CtHandle *out)
{
    if (!out) {
        return fail(nullptr, CT_E_INVAL, "out is NULL");
    }
    *out = nullptr;
    if (!s) {
        return fail(nullptr, CT_E_INVAL, "scene is NULL");
    }
    if (s->abi_version != CT_ABI_VERSION) {
        return fail(nullptr, CT_E_INVAL, "abi_version %u, library is %u", s->abi_version, CT_ABI_VERSION);
    }
    if (!s->density_host || s->dims[0] < 2 || s->dims[1] < 2 || s->dims[2] < 2 || s->dims[0] > 2048 ||
        s->dims[1] > 2048 || s->dims[2] > 2048) {
        return fail(nullptr, CT_E_INVAL, "density volume missing or dims out of range [2,2048]");
    }
    if (s->mode < 0 || s->mode > 2) {
        return fail(nullptr, CT_E_INVAL, "Invalid Render Mode %d", s->mode); // CloudMaterial.cpp:62
    }
    if (s->estimator != CT_EST_MARCH && s->estimator != CT_EST_DELTA) {
        return fail(nullptr, CT_E_INVAL, "unknown estimator %d", s->estimator);
    }
    if (s->estimator == CT_EST_DELTA && (s->flags & CT_FLAG_SIMPLE_KERNEL)) {
        return fail(nullptr, CT_E_INVAL, "the one-thread-per-pixel cross-check kernel only implements MARCH");
    }
    if (!s->mie_host || !s->chopped_mie_host || s->mie_count != (uint32_t)kMieN) {
        return fail(nullptr, CT_E_INVAL, "Mie tables must be %d floats each", kMieN);
    }
    if (s->width == 0 || s->height == 0 || s->width > 12288 || s->height > 4096) {   // (width: the tonemap kernel keeps a row of column sums in LDS)
        return fail(nullptr, CT_E_INVAL, "frame %ux%u out of range (seed packing x*4096+y needs H <= 4096)", s->width,
                    s->height);
    }
    if (!(s->sample_step > 0.f) || !(s->sample_step <= 0.03125f) || !(s->cloud_size_m > 0.f) ||
        !(s->mean_free_path_m > 0.f) || s->max_depth < 2 || s->max_depth > 65535) {
        // (a suspended path carries its depth in 16 bits: BatchArgs::cont_out)
        return fail(nullptr, CT_E_INVAL, "sample_step/cloud_size_m/mean_free_path_m/max_depth out of range (max_depth 2..65535)");
    }
    if (!(s->cloud_size_m / s->mean_free_path_m * s->sample_step < 80.f)) {
        return fail(nullptr, CT_E_INVAL, "optical depth per step %g too large (must be < 80)",
                    (double)(s->cloud_size_m / s->mean_free_path_m * s->sample_step));
    }
    const float ll = s->light_direction[0] * s->light_direction[0] + s->light_direction[1] * s->light_direction[1] +
                     s->light_direction[2] * s->light_direction[2];
    if (!(ll > 0.f)) {
        return fail(nullptr, CT_E_INVAL, "light_direction is zero");
    }
    if (s->shard_count == 0 || s->shard_index >= s->shard_count) {
        return fail(nullptr, CT_E_INVAL, "shard %u of %u", s->shard_index, s->shard_count);
    }
    CtHandle h = new (std::nothrow) CtHandle_();
    if (!h) {
        return fail(nullptr, CT_E_NOMEM, "out of host memory");
    }
    const int rc = create_impl(s, h);
    if (rc != CT_OK) {
        g_create_error = h->error;
        release(h);
        return rc;
    }
    // host pointers are not retained
    h->scene.density_host = nullptr;
    h->scene.mie_host = nullptr;
    h->scene.chopped_mie_host = nullptr;
    *out = h;
    return CT_OK;
}

extern "C" int ct_destroy(CtHandle h)
{
    release(h);
    return CT_OK;
}

extern "C" const char *ct_last_error(CtHandle h)
{
    return h ? h->error.c_str() : g_create_error.c_str();
}

extern "C" int ct_set_stream(CtHandle h, void *hip_stream)
{
    NEED(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return CT_OK;
}

extern "C" int ct_set_camera(CtHandle h, const float eye[3], const float U[3], const float V[3], const float W[3])
{
    NEED(h);
    if (!eye || !U || !V || !W) {
        return fail(h, CT_E_INVAL, "camera vectors must not be NULL");
    }
    DevScene &d = h->dev;
    d.ex = eye[0]; d.ey = eye[1]; d.ez = eye[2];
    d.ux = U[0]; d.uy = U[1]; d.uz = U[2];
    d.vx = V[0]; d.vy = V[1]; d.vz = V[2];
    d.wx = W[0]; d.wy = W[1]; d.wz = W[2];
    h->camera_set = true;
    h->queue_dirty = true; // primary rays and the work queue depend on the pose
    discard_ahead(h);      // (and so does every sample rendered ahead of the calls)
    return CT_OK;
}

// Primary rays of the current pose + the list of this shard's pixels that hit the box, in
// tile-Morton order, cut into groups of 64 (one wave's worth).
static int rebuild_queue(CtHandle h)
{
    TracePhase trace(h, "rebuild_queue");
    const uint32_t W = h->scene.width, H = h->scene.height;
    const size_t pixels = (size_t)W * H;
    HIPCHK(h, launch_primary_rays(h->dev, h->d_primary, h->stream));
    if (h->scene.estimator == CT_EST_DELTA) {
        HIPCHK(h, launch_primary_advance_delta(h->dev, h->d_primary, h->d_advance, h->stream));
    } else {
        HIPCHK(h, launch_primary_advance(h->dev, h->d_primary, h->d_advance, h->stream));
    }
    if (!h->d_hit) {
        HIPCHK(h, dmalloc(&h->d_hit, pixels));
        HIPCHK(h, hipHostMalloc((void **)&h->hit_host, pixels, hipHostMallocDefault));
    }
    HIPCHK(h, launch_hit_flags(h->d_primary, h->d_hit, (uint32_t)pixels, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->hit_host, h->d_hit, pixels, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint8_t *hit = h->hit_host;
    const uint32_t tiles_x = h->dev.tiles_x, tiles_y = h->dev.tiles_y;
    std::vector<std::pair<uint32_t, uint32_t>> tiles;
    for (uint32_t ty = 0; ty < tiles_y; ty++) {
        for (uint32_t tx = 0; tx < tiles_x; tx++) {
            if (tile_owner(tx, ty, h->scene.shard_count) == h->scene.shard_index) {
                tiles.emplace_back(h->tile_hilbert ? hilbert2(tx, ty) : morton2(tx, ty), ty * tiles_x + tx);
            }
        }
    }
    std::sort(tiles.begin(), tiles.end());
    std::vector<uint32_t> list;
    list.reserve(pixels / std::max(1u, h->scene.shard_count) + 64);
    uint64_t own = 0;
    for (const auto &t : tiles) {
        const uint32_t ty = t.second / tiles_x, tx = t.second - ty * tiles_x;
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t x = tx * kTile + (l & 7u), y = ty * kTile + (l >> 3);
            if (x < W && y < H) {
                own++;
                const uint32_t p = y * W + x;
                if (hit[p]) {
                    list.push_back(p);
                }
            }
        }
    }
    h->own_pixels = own;
    h->hit_pixels = list.size();
    while (list.size() % 64 != 0) {
        list.push_back(0xffffffffu);
    }
    h->n_groups = (uint32_t)(list.size() / 64);
    if (h->n_groups > h->groups_capacity) {
        for (void *p : { (void *)h->d_pixels, (void *)h->d_cost }) {
            if (p) {
                HIPCHK(h, hipFree(p));
            }
        }
        h->d_pixels = h->d_cost = nullptr;
        for (void *p : { (void *)h->d_group_rank, (void *)h->d_group_order }) {
            if (p) {
                HIPCHK(h, hipFree(p));
            }
        }
        h->d_group_rank = h->d_group_order = nullptr;
        HIPCHK(h, dmalloc(&h->d_pixels, (size_t)h->n_groups * 64));
        HIPCHK(h, dmalloc(&h->d_cost, 2 * (size_t)h->n_groups));
        HIPCHK(h, dmalloc(&h->d_group_rank, (size_t)h->n_groups));
        HIPCHK(h, dmalloc(&h->d_group_order, (size_t)h->n_groups));
        h->groups_capacity = h->n_groups;
    }
    h->group_order.resize(h->n_groups);
    h->group_tile.resize(h->n_groups);
    for (uint32_t i = 0; i < h->n_groups; i++) {
        h->group_order[i] = i;
        const uint32_t p = list[(size_t)i * 64];
        h->group_tile[i] = (p / W / kTile) * tiles_x + (p % W) / kTile;
    }
    if (h->tile_deepest.size() == (size_t)tiles_x * tiles_y) {
        // The cost-measuring launch of this pose is waited for to its last path, so it should START with the groups whose paths
        // are long.  Where those were for the last pose that was measured is a good guess while the camera is dragged
        // (Camera::rotate, Camera.cpp:93-98): classes of the deepest path seen in a group's tile, deepest first, tile order within.
        auto log2_class = [](uint32_t c) {
            uint32_t k = 0;
            while (c) {
                k++;
                c >>= 1;
            }
            return k;
        };
        std::stable_sort(h->group_order.begin(), h->group_order.end(), [&](uint32_t a, uint32_t b) {
            return log2_class(h->tile_deepest[h->group_tile[a]]) > log2_class(h->tile_deepest[h->group_tile[b]]);
        });
    }
    h->group_depth.assign(h->n_groups, 0.f);
    if (h->n_groups) {
        HIPCHK(h, hipMemcpyAsync(h->d_pixels, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                 h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_cost, 0, 2 * (size_t)h->n_groups * sizeof(uint32_t), h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->queue_dirty = false;
    h->order_tuned = false;
    h->jobs_S = 0; // force a new job list
    return CT_OK;
}

// A launch of about 5 ms or less (10-40 subframes of a 1024^2 frame: the reference's display cadence, Camera.cpp:189).  Such a
// launch works on EVERY pixel group at once instead of a few neighbouring ones for a thousand subframes each, and its paths
// miss L2 half again as often at the same instruction count (profiles/r03f/pmc_prog.txt: hit rate 41 % instead of 62 %); what
// helps is keeping an XCD on a compact part of the image -- per-XCD queues over 16 regions -- with single-subframe jobs for
// every group deeper than four bounces and refills of 16 lanes at a time: 5.43 -> 4.74 ms per 10-subframe update.
static bool short_batch(CtHandle h, uint32_t S)
{
    // (MARCH only: the DELTA kernel, which makes a sixteenth of the fetches, loses 3 % with this layout -- 3.94 vs 3.82 ms)
    return h->scene.estimator != CT_EST_DELTA && (double)S * (double)std::max<uint64_t>(h->hit_pixels, 1) < 4.0e7;
}

// Job list for batches of S subframes.  A job is (group, subframe range); its length is chosen
// so that a job's expected serial work per lane stays bounded: groups whose paths are deep get
// one-subframe jobs, cheap groups up to 16 subframes per job (Handle::job_max, job_work).
static int build_jobs(CtHandle h, uint32_t S, uint32_t chunk_groups)
{
    chunk_groups = std::max(1u, std::min(chunk_groups, std::max(h->n_groups, 1u)));
    if (h->jobs_S >= S && h->jobs_brief == short_batch(h, S) && h->chunk_groups == chunk_groups) {
        return CT_OK; // a list for a larger batch serves a smaller one (the kernel clips the jobs)
    }
    TracePhase trace(h, "build_jobs");
    if (h->order_tuned) {
        S = std::max(S, h->jobs_hint);   // (the cost-measuring launch gets a list of its own: one-subframe jobs, its few subframes only)
    }
    // One queue per XCD: groups are in tile-Morton order, so a contiguous range of them is a
    // compact image region whose paths read a compact part of the volume.  The ranges are cut at
    // equal shares of the measured cost (path depth + the primary march), not of the pixel count.
    // cost units per bounce (BatchArgs::cost): MARCH counts fetches + 4 per bounce, DELTA bounces
    const float unit = (h->scene.estimator == CT_EST_DELTA) ? 1.f : 16.f;
    const bool brief = short_batch(h, S);
    // (per-XCD queues are cut at equal shares of the MEASURED cost: before anything is measured they would be equal shares of
    // pixels, and the XCD that got the cloud's body would finish long after the others)
    const bool queues = (h->queues_enabled || (brief && !h->tune.XCD_QUEUES.get())) && !(h->scene.flags & CT_FLAG_SIMPLE_KERNEL) &&
                        (h->order_tuned || h->tune.XCD_QUEUES_UNTUNED.get());
    const uint32_t nq = (queues && h->n_groups >= (uint32_t)kQueues) ? (uint32_t)kQueues : 1u;
    const uint32_t regions_wanted = (brief && !h->queues_enabled && !h->tune.XCD_REGIONS.get()) ? 16u : h->regions;
    const float job_work = (brief && !h->tune.JOB_WORK.get()) ? 4.f : h->job_work;
    // Groups whose paths are deep (mean cost >= shared_depth bounces) go to the shared queue.
    const float shared_cost = h->shared_depth * unit;
    std::vector<uint8_t> queue_of(h->n_groups, 0);
    {
        double total = 0;
        for (uint32_t g = 0; g < h->n_groups; g++) {
            if (nq > 1u && h->group_depth[g] >= shared_cost) {
                queue_of[g] = (uint8_t)kQueues;
            } else {
                total += (double)h->group_depth[g] + unit;
            }
        }
        double run = 0;
        for (uint32_t g = 0; g < h->n_groups; g++) {
            if (queue_of[g] == (uint8_t)kQueues) {
                continue;
            }
            const double w = (double)h->group_depth[g] + unit;
            // h->regions compact image regions of equal cost, dealt round-robin to the queues: every
            // XCD still works on 1/8 of the image (a handful of compact pieces), and errors of the
            // cost estimate, which are correlated in space, average out over its pieces
            const uint32_t regions = std::max(nq, regions_wanted);
            const uint32_t r = (uint32_t)std::min<double>(regions - 1, std::floor((run + 0.5 * w) / total * regions));
            queue_of[g] = (uint8_t)(r % nq);
            run += w;
        }
    }
    // (the list itself is written into pinned host memory that stays with the handle: a new pose builds it twice, and 860 k
    // jobs in two fresh vectors cost 9 ms of page faults and staged copies)
    auto job_length = [&](uint32_t g) {
        const float d = h->group_depth[g];
        if (!h->order_tuned) {
            // the cost-measuring launch of a pose knows no depths yet: ONE subframe per job, or the wave that gets a group of the
            // cloud's body runs its eight or ten subframes one after another in the same 64 lanes -- 21 ms for 10 subframes
            // of a 1024^2 frame that are 13 ms once the order is set
            return 1u;
        }
        uint32_t len = h->job_max;
        if (d > 0.f) {
            len = (uint32_t)std::min((float)h->job_max, std::max(1.f, job_work * unit / d));
        }
        return len;
    };
    size_t total_jobs = 0;
    for (uint32_t g = 0; g < h->n_groups; g++) {
        const uint32_t len = job_length(g);
        total_jobs += (S + len - 1) / len;
    }
    if (total_jobs > h->jobs_host_capacity) {
        for (uint32_t **pp : { &h->jobs_host_g, &h->jobs_host_s }) {
            if (*pp) {
                HIPCHK(h, hipHostFree(*pp));
                *pp = nullptr;
            }
        }
        h->jobs_host_capacity = total_jobs + total_jobs / 4 + 1024;
        HIPCHK(h, hipHostMalloc((void **)&h->jobs_host_g, h->jobs_host_capacity * sizeof(uint32_t), hipHostMallocDefault));
        HIPCHK(h, hipHostMalloc((void **)&h->jobs_host_s, h->jobs_host_capacity * sizeof(uint32_t), hipHostMallocDefault));
    }
    uint32_t *const jg = h->jobs_host_g, *const js = h->jobs_host_s;
    size_t nj = 0;
    std::vector<double> q_weight(kQueues + 1, 0.0);
    // chunk by chunk (a contiguous piece of the group order each), and within a chunk queue by queue
    const uint32_t n_chunks = h->n_groups ? (h->n_groups + chunk_groups - 1) / chunk_groups : 1u;
    if (n_chunks > 1 && h->chunk_interleave) {
        // Every chunk gets the same mix of expensive and cheap groups (2 % of the groups make 98 % of the cost): the job order
        // becomes places 0, C, 2C, ... of the cost-sorted order, then 1, C + 1, ..., so that a contiguous piece of it is every
        // C-th group -- launches of equal length instead of a few long ones and many that are over before the paths they
        // resumed have moved.
        std::vector<uint32_t> mixed;
        mixed.reserve(h->n_groups);
        for (uint32_t c = 0; c < n_chunks; c++) {
            for (uint32_t r = c; r < h->n_groups; r += n_chunks) {
                mixed.push_back(h->group_order[r]);
            }
        }
        // (pieces must be whole chunks: with n_groups not a multiple of n_chunks the first chunks are one group longer than
        // chunk_groups allows only if chunk_groups * n_chunks < n_groups, which the rounding above excludes)
        h->job_order = mixed;
    } else if (n_chunks > 1 && h->chunk_morton) {
        // (probe: chunks = compact image regions -- the groups in tile-Morton order, which is their index order)
        h->job_order.resize(h->n_groups);
        for (uint32_t g = 0; g < h->n_groups; g++) {
            h->job_order[g] = g;
        }
    } else {
        h->job_order = h->group_order;
    }
    h->chunk_q_begin.assign(n_chunks, std::array<uint32_t, kQueues + 2>{});
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint32_t r0 = c * chunk_groups, r1 = std::min(h->n_groups, r0 + chunk_groups);
        for (uint32_t x = 0; x <= (uint32_t)kQueues; x++) {
            h->chunk_q_begin[c][x] = (uint32_t)nj;
            if (x != 0u && nq == 1u && x != (uint32_t)kQueues) {
                continue;   // (one global list: everything is in queue 0)
            }
            for (uint32_t r = r0; r < r1; r++) {
                const uint32_t g = h->job_order[r];
                if (queue_of[g] != x) {
                    continue;
                }
                q_weight[x] += (double)h->group_depth[g] + unit;
                const uint32_t len = job_length(g);
                for (uint32_t s0 = 0; s0 < S; s0 += len) {
                    jg[nj] = g;
                    js[nj] = s0 | (std::min(len, S - s0) << 16);
                    nj += 1;
                }
            }
        }
        h->chunk_q_begin[c][kQueues + 1] = (uint32_t)nj;
    }
    for (int x = 0; x <= kQueues + 1; x++) {
        h->q_begin[x] = h->chunk_q_begin[0][x];
    }
    if (h->tune.STATS.get()) {
        fprintf(stderr, "[cloudtrace] job queues (S=%u, %u chunk(s) of %u groups):", S, n_chunks, chunk_groups);
        for (int x = 0; x <= kQueues; x++) {
            fprintf(stderr, " q%d weight %.0f;", x, q_weight[x]);
        }
        fprintf(stderr, "\n");
    }
    {
        // the job order itself, both ways: where a group's results go in its chunk's scratch, and whose a column is
        std::vector<uint32_t> rank(h->n_groups);
        for (uint32_t r = 0; r < h->n_groups; r++) {
            rank[h->job_order[r]] = r;
        }
        if (h->n_groups) {
            HIPCHK(h, hipMemcpyAsync(h->d_group_rank, rank.data(), rank.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(h->d_group_order, h->job_order.data(), h->job_order.size() * sizeof(uint32_t),
                                     hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));   // (`rank` dies at the end of this block)
        }
    }
    h->chunk_groups = chunk_groups;
    h->n_chunks = n_chunks;
    // (nq == 1: everything sits in queue 0 and the other XCDs' waves steal from it -- one global
    // cost-sorted list, the behaviour before the queues were split)
    h->n_jobs = (uint32_t)nj;
    if (h->n_jobs > h->jobs_capacity) {
        for (void *p : { (void *)h->d_job_group, (void *)h->d_job_sub }) {
            if (p) {
                HIPCHK(h, hipFree(p));
            }
        }
        h->d_job_group = h->d_job_sub = nullptr;
        h->jobs_capacity = h->n_jobs + h->n_jobs / 4;
        HIPCHK(h, dmalloc(&h->d_job_group, h->jobs_capacity));
        HIPCHK(h, dmalloc(&h->d_job_sub, h->jobs_capacity));
    }
    if (h->n_jobs) {
        HIPCHK(h, hipMemcpyAsync(h->d_job_group, jg, nj * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_job_sub, js, nj * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));   // (the pinned list is rewritten by the next build)
    }
    h->jobs_S = S;
    h->jobs_brief = brief;
    return CT_OK;
}

// Longest-processing-time-first: visit the groups whose paths were deepest first, so that the
// 2000-bounce paths start at the head of the launch instead of becoming its tail.  Groups of
// the same cost class (power of two) keep their Morton order for cache locality.  The order and
// the job lengths only change the schedule, never a result.
static int tune_order(CtHandle h, uint32_t measured_subframes)
{
    TracePhase trace(h, "tune_order");
    h->order_tuned = true;
    if (h->n_groups < 2 || measured_subframes == 0) {
        return CT_OK;
    }
    std::vector<uint32_t> cost(2 * (size_t)h->n_groups);
    HIPCHK(h, hipMemcpyAsync(cost.data(), h->d_cost, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    auto log2_class = [](uint32_t c) {
        uint32_t k = 0;
        while (c) {
            k++;
            c >>= 1;
        }
        return k;
    };
    // Order: by the deepest path a group has produced (a launch ends when its last path does, so
    // the groups that can produce long paths must not be the last ones running), then by mean cost.
    const uint32_t *deepest = cost.data() + h->n_groups;
    h->tile_deepest.assign((size_t)h->dev.tiles_x * h->dev.tiles_y, 0u);
    for (uint32_t g = 0; g < h->n_groups; g++) {
        h->group_order[g] = g;
        h->group_depth[g] = (float)cost[g] / (64.f * (float)measured_subframes);
        if (g < h->group_tile.size()) {
            uint32_t &t = h->tile_deepest[h->group_tile[g]];
            t = std::max(t, deepest[g]);   // (a guess for the next pose's cost-measuring launch: rebuild_queue)
        }
    }
    std::stable_sort(h->group_order.begin(), h->group_order.end(), [&](uint32_t a, uint32_t b) {
        const uint32_t ka = log2_class(deepest[a]), kb = log2_class(deepest[b]);
        if (ka != kb) {
            return ka > kb;
        }
        return log2_class(cost[a]) > log2_class(cost[b]);
    });
    if (h->tune.STATS.get()) {
        // share of the measured cost by class of the deepest path seen
        std::vector<double> share(34, 0.0);
        std::vector<uint32_t> count(34, 0);
        double total = 0;
        for (uint32_t g = 0; g < h->n_groups; g++) {
            share[log2_class(deepest[g])] += cost[g];
            count[log2_class(deepest[g])] += 1;
            total += cost[g];
        }
        fprintf(stderr, "[cloudtrace] cost share by deepest-path class (2^(k-1) <= depth < 2^k):");
        for (int k = 0; k < 34; k++) {
            if (count[k]) {
                fprintf(stderr, " k=%d groups %u share %.1f%%;", k, count[k], 100.0 * share[k] / std::max(total, 1.0));
            }
        }
        fprintf(stderr, "\n");
    }
    h->jobs_S = 0; // rebuild the job list with the new order
    return CT_OK;
}

// Entries of one subframe in the batch scratch: the columns of a chunk's pixel groups (compact) or the frame (simple).
static size_t frame_stride(CtHandle h)
{
    if (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) {
        return (size_t)h->scene.width * h->scene.height;
    }
    return (size_t)std::max(h->chunk_groups, 1u) * 64;
}

// How many regions a batch of S subframes should have: enough launches between a batch and its accumulate kernel that
// no path of it is still running -- a path needs up to ~10 ms (2000 bounces) of running time, a launch lasts about as long
// as its samples take at ~3 Gsamples/s -- but never more than the budget holds.
static int wanted_regions(CtHandle h, uint32_t S, uint64_t need, uint64_t budget_float4)
{
    int r;
    if (h->max_age_override > 0) {
        r = h->max_age_override + 1;
    } else {
        const double est_ms = std::max(0.05, (double)need / 3.0e6);
        r = 1 + (int)std::ceil(15.0 / est_ms);
    }
    (void)S;
    r = (int)std::min<uint64_t>((uint64_t)r, std::max<uint64_t>(budget_float4 / std::max<uint64_t>(need, 1), 2));
    return std::min(CtHandle_::kMaxRegions, std::max(2, r));
}

static uint64_t scratch_slot_bytes(CtHandle h)
{
    if (h && h->scratch_cap_bytes) {
        return h->scratch_cap_bytes;   // (the device could not give more: ensure_frames failed with less)
    }
    // CT_SCRATCH_GIB sets the size of one of the two regions a long batch uses.  Default 16: the 1024-spp job of a 1024^2
    // frame is then ONE launch, 14 GB, instead of two of 512 -- a launch costs a few ms besides its samples -- and 28 GB of
    // scratch are a tenth of this GPU's memory.  Allocated as needed; a device that cannot give that much gets less (below).
    uint64_t slot_bytes = 16ull << 30;
    if (const char *e = h->tune.SCRATCH_GIB.get()) {
        slot_bytes = (uint64_t)std::min(64, std::max(1, atoi(e))) << 30;
    }
    if (const char *e = h->tune.SCRATCH_MIB.get()) {   // (tests: chunks of a few pixel groups on small frames)
        slot_bytes = (uint64_t)std::min(65536, std::max(1, atoi(e))) << 20;
    }
    return slot_bytes;
}

// How many pixel groups a launch of S subframes renders at once: what one scratch region holds (all of them when it can).
static uint32_t groups_per_chunk(CtHandle h, uint32_t S)
{
    if (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) {
        return std::max(h->n_groups, 1u);
    }
    const uint64_t region = scratch_slot_bytes(h) / sizeof(float4);
    // (two regions at least, and result indices are 32 bits)
    const uint64_t cap = std::min<uint64_t>(region, 0xffffffffull / 2);
    const uint64_t g = cap / ((uint64_t)std::max(S, 1u) * 64);
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(g, std::max(h->n_groups, 1u)));
}

// At least `total` float4 of per-sample scratch (nothing may be in flight when it grows: the caller has flushed).
static int reserve_frames(CtHandle h, size_t total)
{
    if (total <= h->frames_total) {
        return CT_OK;
    }
    if (h->d_frames_all) {
        HIPCHK(h, hipFree(h->d_frames_all));
        h->d_frames_all = nullptr;
        h->frames_total = 0;
        h->slot_capacity = 0;
    }
    const hipError_t e = dmalloc(&h->d_frames_all, total);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        h->d_frames_all = nullptr;
        return fail(h, e == hipErrorOutOfMemory ? CT_E_NOMEM : CT_E_HIP, "per-sample scratch of %.1f GB: %s", (double)total * 16e-9,
                    hipGetErrorString(e));
    }
    // touch it now: the first launch that writes a fresh part of a large allocation has been seen to
    // take 30 ms longer (measured on the 3.5 GB scratch of a 256-subframe batch)
    HIPCHK(h, hipMemsetAsync(h->d_frames_all, 0, total * sizeof(float4), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->frames_total = total;
    return CT_OK;
}

// The per-sample scratch laid out for launches of S subframes over chunks of h->chunk_groups pixel groups: regions of
// S * stride entries each.  Nothing may be in flight when the layout changes (the caller has flushed).
static int ensure_frames(CtHandle h, uint32_t S, bool relayout)
{
    const size_t need = std::max<size_t>((size_t)S * frame_stride(h), 1);
    if (!relayout && need <= h->slot_capacity) {
        return CT_OK;   // (a waited-for batch, the cost-measuring launch: any region that is large enough will do)
    }
    discard_ahead(h);
    const uint64_t budget = 2 * scratch_slot_bytes(h) / sizeof(float4);
    // (a waited-for batch needs one region; the others are allocated when batches are first enqueued)
    const int regions = relayout ? wanted_regions(h, S, need, std::max<uint64_t>(budget, 2 * need)) : 1;
    size_t total = (size_t)regions * need;
    if (total > 0xffffffffull) {
        if ((relayout ? 2 : 1) * need > 0xffffffffull) {
            return fail(h, CT_E_INVAL, "batch of %u subframes is too large for 32-bit result indices", S);
        }
        total = (0xffffffffull / need) * need;
    }
    {
        const int rc = reserve_frames(h, total);
        if (rc != CT_OK) {
            return rc;
        }
    }
    h->slot_capacity = need;
    h->n_regions = (int)std::min<size_t>((size_t)regions, h->frames_total / need);
    h->layout_S = S;
    h->next_slot = 0;
    return CT_OK;
}

static float4 *slot_frames(CtHandle h, int slot)
{
    return h->d_frames_all + (size_t)slot * h->slot_capacity;
}

// Waits for one slot's launch (and its accumulate kernel, if enqueued) and books the kernel times.
static int collect(CtHandle h, CtHandle_::Slot &sl)
{
    if (!sl.pending) {
        return CT_OK;
    }
    sl.pending = false;
    HIPCHK(h, hipEventSynchronize(sl.accumulated ? sl.ev_acc1 : sl.ev_done));
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, sl.ev_start, sl.ev_done));
    h->render_ms += ms;
    if (h->tune.TRACE.get()) {
        uint32_t cnt[3] = { 0, 0, 0 };
        hipMemcpy(cnt, h->d_cont_count, sizeof cnt, hipMemcpyDeviceToHost);
        fprintf(stderr, "[cloudtrace] estimator launch %.2f ms (suspended paths per buffer now: %u %u, cursor %u)\n", ms,
                cnt[0], cnt[1], cnt[2]);
    }
    if (sl.accumulated) {
        HIPCHK(h, hipEventElapsedTime(&ms, sl.ev_acc0, sl.ev_acc1));
        h->accum_ms += ms;
    }
    h->launches += 1;
    return CT_OK;
}

// The convergence test on the running mean after `subframes` subframes, and its outcome copied to where the host can
// read it without waiting (ct_converged_at).
static int enqueue_convergence_test(CtHandle h, uint32_t subframes)
{
    HIPCHK(h, launch_converged_freeze(h->d_mean, h->d_m2, subframes, (uint64_t)h->scene.width * h->scene.height, 500u /* Camera.cpp:267 */,
                                      h->d_freeze, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->freeze_host, h->d_freeze, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    return CT_OK;
}

// The accumulate kernel(s) for subframes [sl.acc_done, sl.acc_done + n) of a slot's batch.  With stop-when-converged they
// are cut at the multiples of its cadence and each is followed by the test (a batch rendered in several chunks of pixel
// groups is a whole image only after its last chunk: it is tested there, if it ends on a multiple).
static int enqueue_accumulate(CtHandle h, CtHandle_::Slot &sl, const float4 *frames, bool dense, uint32_t n)
{
    const bool simple = (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) != 0;
    const bool chunked = !(simple || dense) && h->n_chunks > 1;
    const uint32_t cadence = h->stop_cadence;
    const uint32_t *frozen = cadence ? h->d_freeze : nullptr;
    HIPCHK(h, hipEventRecord(sl.ev_acc0, h->stream));
    while (n != 0u) {
        const uint32_t off = sl.acc_done, next = sl.first + off;   // (next: the subframe id of the first sample of this piece)
        uint32_t piece = n;
        if (cadence && !chunked) {
            piece = std::min(n, cadence - (next - 1u) % cadence);
        }
        if (simple || dense) {
            HIPCHK(h, launch_accumulate_batch(frames + (size_t)off * h->scene.width * h->scene.height, h->d_mean, h->d_m2, next, piece,
                                              h->scene.width, h->scene.height, h->scene.shard_index, h->scene.shard_count,
                                              h->d_counters + 8, frozen, h->stream));
        } else {
            HIPCHK(h, launch_accumulate_list(frames + (size_t)off * frame_stride(h), (uint32_t)frame_stride(h), h->d_pixels, sl.groups * 64u,
                                             h->n_chunks <= 1 ? nullptr : h->d_group_order, sl.rank_base,
                                             sl.with_misses, h->d_primary, h->d_mean, h->d_m2, next, piece, h->scene.width,
                                             h->scene.height, h->scene.shard_index, h->scene.shard_count, h->d_counters + 8, frozen,
                                             h->stream));
        }
        sl.acc_done += piece;
        n -= piece;
        const uint32_t upto = sl.first + sl.acc_done - 1u;
        const bool whole_image = !chunked || (sl.last_chunk && sl.acc_done == sl.S);
        if (cadence && whole_image && upto % cadence == 0u && upto >= h->stop_min) {
            const int rc = enqueue_convergence_test(h, upto);
            if (rc != CT_OK) {
                return rc;
            }
        }
    }
    HIPCHK(h, hipEventRecord(sl.ev_acc1, h->stream));   // (of several partial accumulates the last one is the one whose time is booked)
    sl.accumulated = true;
    return CT_OK;
}

// How far the running mean may follow: everything that is complete, or -- with render-ahead -- the calls minus the lag that
// lets every call find its share complete (see CtHandle_::ahead).
static uint32_t accumulate_target(CtHandle h)
{
    if (h->ahead == 0) {
        return 0xffffffffu;
    }
    const uint64_t lag = (uint64_t)std::max(h->n_regions - 1, 1) * h->ahead;
    return h->subframes > lag ? (uint32_t)(h->subframes - lag) : 0u;
}

// Enqueues the accumulate kernels of the complete batches, oldest first, up to subframe `target`.
static int advance_accumulate(CtHandle h, uint32_t target)
{
    while (!h->waiting.empty()) {
        const int w = h->waiting.front();
        CtHandle_::Slot &sl = h->slots[w];
        if (!sl.complete) {
            break;
        }
        const uint32_t next = sl.first + sl.acc_done, last = sl.first + sl.S - 1u;
        const uint32_t upto = std::min(last, target);
        if (upto >= next && sl.acc_done < sl.S) {
            const int rc = enqueue_accumulate(h, sl, slot_frames(h, w), false, upto - next + 1u);
            if (rc != CT_OK) {
                return rc;
            }
        }
        if (sl.acc_done < sl.S) {
            break;
        }
        sl.awaits_accumulate = false;
        h->waiting.erase(h->waiting.begin());
    }
    return CT_OK;
}

// Forgets the subframes that were rendered ahead of the calls (the pose, the layout of the scratch or the count of
// subframes changes).  Nothing may be in flight: the caller has flushed, so what is left in `waiting` is exactly that.
static void discard_ahead(CtHandle h)
{
    for (int w : h->waiting) {
        h->slots[w].awaits_accumulate = false;
    }
    h->waiting.clear();
    h->rendered = h->subframes;
}

// The estimator launch itself: the kernel that fits the handle and the batch.
static int launch_estimator(CtHandle h, const BatchArgs &ba)
{
    DevScene sc = h->dev;
    if (ba.S != 0 && short_batch(h, ba.S) && !h->tune.REGEN_MIN.get()) {
        sc.regen_min = 16;   // (see short_batch)
    }
    bool launched = false;
#ifdef CT_EXPERIMENTS
    if (h->exchange && !ba.cost && !ba.cont_in && !ba.cont_out && h->scene.estimator == CT_EST_DELTA) {
        if (h->exchange == 2) {
            HIPCHK(h, launch_render_delta_w(sc, ba, h->xshape, h->stream));
        } else {
            HIPCHK(h, launch_render_delta_x(sc, ba, h->xshape, h->stream));
        }
        launched = true;
    }
#endif
    if (launched) {
        // (an experiments-build kernel rendered the batch)
    } else if (h->scene.estimator == CT_EST_DELTA) {
        HIPCHK(h, launch_render_delta(sc, ba, h->shape, h->stream));
    } else {
        HIPCHK(h, launch_render_persistent(sc, ba, h->shape, h->stream));
    }
    return CT_OK;
}

// Enqueues one launch of the estimator over S subframes.  `frames` is the dense frame buffer
// (ct_render_subframe) or NULL for the slot's region of the scratch.  With `suspend` the launch hands its
// surviving paths to the next one and the batch's accumulate kernel follows the launch that is max_age = n_regions - 1
// batches younger; the paths the previous launch suspended are resumed in any case.  Does not wait.
static int submit_batch(CtHandle h, int slot, float4 *dense_frames, uint32_t first, uint32_t S, bool accumulate,
                        bool suspend, uint32_t chunk)
{
    CtHandle_::Slot &sl = h->slots[slot];
    const bool simple = (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) != 0;
    const bool dense = dense_frames != nullptr;
    const uint32_t max_age = (uint32_t)(h->n_regions - 1);
    const uint32_t rank_base = chunk * h->chunk_groups;
    const uint32_t chunk_n = h->n_groups > rank_base ? std::min(h->chunk_groups, h->n_groups - rank_base) : 0u;
    BatchArgs ba{};
    ba.frames = dense ? dense_frames : (simple ? slot_frames(h, slot) : h->d_frames_all);
    ba.frame_stride = (simple || dense) ? 0u : (uint32_t)frame_stride(h);
    ba.out_offset = (simple || dense) ? 0u : (uint32_t)((size_t)slot * h->slot_capacity);
    ba.primary = h->d_primary;
    ba.advance = h->no_advance ? nullptr : h->d_advance;
    ba.pixels = h->d_pixels;
    ba.group_rank = (simple || dense || h->n_chunks <= 1) ? nullptr : h->d_group_rank;   // (one chunk: column g * 64, as ever)
    ba.rank_base = rank_base;
    ba.job_group = h->d_job_group;
    ba.job_sub = h->d_job_sub;
    // the cost-measuring launch of a pose: every path leaves what it cost beside its result, summed per group below
    const bool measure = !h->order_tuned && !simple && !dense && chunk_n != 0u;
    if (measure) {
        const size_t entries = (size_t)S * frame_stride(h);
        if (entries > h->cost_plane_capacity) {
            if (h->d_cost_plane) {
                HIPCHK(h, hipFree(h->d_cost_plane));
                h->d_cost_plane = nullptr;
            }
            h->cost_plane_capacity = entries + entries / 4;
            HIPCHK(h, dmalloc(&h->d_cost_plane, h->cost_plane_capacity));
        }
        HIPCHK(h, hipMemsetAsync(h->d_cost_plane, 0, entries * sizeof(uint2), h->stream));
        ba.cost = h->d_cost_plane;
    }
    static const std::array<uint32_t, kQueues + 2> no_jobs{};
    const auto &qb = chunk < h->chunk_q_begin.size() ? h->chunk_q_begin[chunk] : no_jobs;   // (the simple kernel has no job list)
    ba.n_jobs = qb[kQueues + 1];            // (absolute index of the chunk's end: the job that raises the "list empty" flag)
    for (int x = 0; x <= kQueues + 1; x++) {
        ba.q_begin[x] = qb[x];
    }
    const uint32_t chunk_jobs = qb[kQueues + 1] - qb[0];
    ba.first_subframe = first;
    ba.S = S;
    ba.queue = sl.queue;
    ba.counters = h->d_counters;
    ba.stats = h->d_counters + kCounterCount + 1;
    ba.timeline = h->d_timeline;
    ba.touched_density = h->d_touched[0];
    ba.touched_shadow = h->d_touched[1];
    if (suspend && short_batch(h, S) && h->serpentine) {
        ba.reverse = (uint32_t)(h->launch_no & 1u);
    }
    const int buf_out = (int)(h->launch_no & 1u), buf_in = buf_out ^ 1;
    ZeroList zero;
    if (h->cont_live) {
        ba.cont_in = h->cont[buf_in];
        ba.cont_in_count = h->d_cont_count + buf_in;
        ba.cont_cursor = h->d_cont_count + 2;
        ba.left_in = h->left[buf_in];
        ba.left_in_count = h->d_cont_count + 3 + buf_in;
        ba.left_cursor = h->d_cont_count + 5;
        zero.add(h->d_cont_count + 2, 1);
        zero.add(h->d_cont_count + 5, 1);
    }
    if (suspend) {
        ba.cont_out = h->cont[buf_out];
        ba.cont_out_count = h->d_cont_count + buf_out;
        ba.cont_capacity = (uint32_t)h->cont_capacity;
        ba.max_age = std::max(1u, max_age);
        ba.cont_total = h->d_cont_total;
        if (h->hand_on_jobs) {
            ba.left_out = h->left[buf_out];
            ba.left_out_count = h->d_cont_count + 3 + buf_out;
            ba.left_capacity = (uint32_t)h->left_capacity;
        }
        zero.add(h->d_cont_count + buf_out, 1);
        zero.add(h->d_cont_count + 3 + buf_out, 1);
    }
    zero.add(sl.queue, kQueueWords);
    HIPCHK(h, launch_zero_words(zero, h->stream));
    if (h->debug_invariants && !simple && !dense && h->n_groups != 0) {
        // NaNs (with a NaN alpha) wherever this batch is going to write: a sample that is never written cannot
        // pass for the one an earlier batch left there
        HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)slot_frames(h, slot), 0x7fc0deadu, (size_t)S * frame_stride(h) * 4u,
                                    h->stream));
    }
    HIPCHK(h, hipEventRecord(sl.ev_start, h->stream));
    if (simple) {
        h->host_paths += h->own_pixels * S;
        h->host_hits += h->hit_pixels * S;
        for (uint32_t s = 0; s < S; s++) {
            BatchArgs one = ba;
            one.frames = ba.frames + (size_t)s * h->scene.width * h->scene.height;
            one.first_subframe = first + s;
            one.S = 1;
            HIPCHK(h, launch_render_simple(h->dev, one, h->scene.shard_index, h->scene.shard_count, h->stream));
        }
    } else {
        if (chunk_jobs != 0 || ba.cont_in) {
            const int rc = launch_estimator(h, ba);
            if (rc != CT_OK) {
                return rc;
            }
            if (measure) {
                HIPCHK(h, launch_cost_reduce(h->d_cost_plane, (uint32_t)frame_stride(h), S, chunk_n,
                                             h->n_chunks <= 1 ? nullptr : h->d_group_order, rank_base, h->d_cost,
                                             h->d_cost + h->n_groups, h->stream));
            }
        }
        // (every group is full but the last one of the pixel list, which padding completes; it sits somewhere in the job order)
        uint64_t chunk_hits = 0;
        for (uint32_t r = rank_base; r < rank_base + chunk_n; r++) {
            chunk_hits += (h->job_order[r] + 1u == h->n_groups) ? 64u - (uint64_t)((uint64_t)h->n_groups * 64u - h->hit_pixels) : 64u;
        }
        if (chunk == 0) {
            h->host_paths += h->own_pixels * S;
        }
        h->host_hits += chunk_hits * S;
        h->iv_expected_dealt += chunk_hits * S;
    }
    HIPCHK(h, hipEventRecord(sl.ev_done, h->stream));
    h->launch_no += 1;
    h->cont_live = suspend && !simple;
    sl.first = first;
    sl.S = S;
    sl.rank_base = rank_base;
    sl.groups = chunk_n;
    sl.with_misses = chunk == 0;
    sl.last_chunk = simple || dense || chunk + 1u >= std::max(h->n_chunks, 1u);
    sl.pending = true;
    sl.accumulated = false;
    sl.awaits_accumulate = false;
    sl.complete = false;
    sl.acc_done = 0;
    if (accumulate) {
        if (suspend) {
            sl.awaits_accumulate = true;
            h->waiting.push_back(slot);
            // every batch that max_age launches have followed is complete: its accumulate kernels may go behind this launch
            uint32_t younger = 0;
            for (size_t i = h->waiting.size(); i-- > 0;) {
                CtHandle_::Slot &w = h->slots[h->waiting[i]];
                if (!w.complete && ++younger > max_age) {
                    w.complete = true;
                }
            }
            return advance_accumulate(h, accumulate_target(h));
        } else {
            sl.complete = true;
            const int rc = enqueue_accumulate(h, sl, dense ? dense_frames : slot_frames(h, slot), dense, S);
            if (rc != CT_OK) {
                return rc;
            }
        }
    }
    return CT_OK;
}

// Nothing is in flight: the conservation identities of CtHandle_::debug_invariants must hold.
static int check_invariants(CtHandle h)
{
    unsigned long long iv[4] = { 0, 0, 0, 0 }, bad = 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(iv, h->d_counters + kCounterCount + 1 + 64, sizeof iv, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(&bad, h->d_counters + 8, sizeof bad, hipMemcpyDeviceToHost));
    h->iv_checks += 1;
    const unsigned long long dealt = iv[0], resumed = iv[1], written = iv[2], suspended = iv[3];
    const bool simple = (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) != 0;
    const bool ok = simple || (dealt + resumed == written + suspended && resumed == suspended && dealt == h->iv_expected_dealt &&
                               bad == 0);
    if (!ok) {
        h->iv_violations += 1;
        fprintf(stderr, "[cloudtrace] INVARIANT VIOLATED: dealt %llu (host expects %llu) resumed %llu written %llu suspended %llu, "
                        "samples without alpha 1 seen by accumulate: %llu\n",
                dealt, (unsigned long long)h->iv_expected_dealt, resumed, written, suspended, bad);
        return fail(h, CT_E_STATE, "invariant violated: dealt %llu (expected %llu) resumed %llu written %llu suspended %llu bad samples %llu",
                    dealt, (unsigned long long)h->iv_expected_dealt, resumed, written, suspended, bad);
    }
    return CT_OK;
}

// Every batch in flight has finished when this returns.  Batches that still wait for suspended paths get ONE launch that
// only resumes them (no jobs, no suspension: everything runs to its end), then their accumulate kernels in order.
static int flush(CtHandle h)
{
    if (h->cont_live) {
        BatchArgs ba{};
        ba.frames = h->d_frames_all;
        ba.frame_stride = (uint32_t)frame_stride(h);   // (unused: resumed paths and handed-on jobs carry their own places)
        ba.primary = h->d_primary;
        ba.advance = h->no_advance ? nullptr : h->d_advance;
        ba.pixels = h->d_pixels;
        ba.job_group = h->d_job_group;
        ba.job_sub = h->d_job_sub;
        ba.n_jobs = 0;                  // q_begin stays all zero: every queue is empty
        ba.first_subframe = 1;
        ba.S = 0;
        ba.queue = h->d_queue;
        ba.counters = h->d_counters;
        ba.stats = h->d_counters + kCounterCount + 1;
        const int buf_in = (int)((h->launch_no & 1u) ^ 1u);
        ba.cont_in = h->cont[buf_in];
        ba.cont_in_count = h->d_cont_count + buf_in;
        ba.cont_cursor = h->d_cont_count + 2;
        ba.left_in = h->left[buf_in];
        ba.left_in_count = h->d_cont_count + 3 + buf_in;
        ba.left_cursor = h->d_cont_count + 5;
        HIPCHK(h, hipMemsetAsync(h->d_cont_count + 2, 0, sizeof(uint32_t), h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_cont_count + 5, 0, sizeof(uint32_t), h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_queue, 0, kQueueWords * sizeof(uint32_t), h->stream));
        HIPCHK(h, hipEventRecord(h->ev_flush0, h->stream));
        const int rc = launch_estimator(h, ba);
        if (rc != CT_OK) {
            return rc;
        }
        HIPCHK(h, hipEventRecord(h->ev_flush1, h->stream));
        h->launch_no += 1;
        h->cont_live = false;
        HIPCHK(h, hipEventSynchronize(h->ev_flush1));
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev_flush0, h->ev_flush1));
        h->render_ms += ms; // the launch that only resumes belongs to the estimator's time
        if (h->tune.TRACE.get()) {
            fprintf(stderr, "[cloudtrace] resume-only launch %.2f ms\n", ms);
        }
    }
    for (int w : h->waiting) {
        h->slots[w].complete = true;
    }
    {
        // everything the caller has asked for; what was rendered ahead of the calls stays in its region for the next ones
        const int rc = advance_accumulate(h, h->ahead ? h->subframes : 0xffffffffu);
        if (rc != CT_OK) {
            return rc;
        }
    }
    for (int k = 0; k < CtHandle_::kMaxRegions; k++) {
        const int rc = collect(h, h->slots[k]);
        if (rc != CT_OK) {
            return rc;
        }
    }
    return h->debug_invariants ? check_invariants(h) : CT_OK;
}

// Job list etc. for batches of S subframes; everything in flight is waited for when it has to change.
static int prepare_batches(CtHandle h, uint32_t S)
{
    const bool simple = (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) != 0;
    if (simple) {
        h->chunk_groups = std::max(h->n_groups, 1u);
        h->n_chunks = 1;
        return CT_OK;
    }
    if (h->queue_dirty) {
        int rc = flush(h);
        discard_ahead(h);
        if (rc == CT_OK) {
            rc = rebuild_queue(h);
        }
        if (rc != CT_OK) {
            return rc;
        }
    }
    const uint32_t want = groups_per_chunk(h, S);
    if (h->jobs_S < S || h->jobs_brief != short_batch(h, S) || h->chunk_groups != want || h->chunk_q_begin.empty()) {
        const int rc = flush(h);   // the job list (and with it the layout of the scratch) changes: nothing may be in flight
        discard_ahead(h);
        return rc == CT_OK ? build_jobs(h, S, want) : rc;
    }
    return CT_OK;
}

// One batch of S subframes (every chunk of the pixel groups, + optional accumulate), waited for, every path run to its
// end.  `dense_frames` is the frame buffer of ct_render_subframe or NULL for the batch scratch.
static int run_batch(CtHandle h, float4 *dense_frames, uint32_t first, uint32_t S, bool accumulate)
{
    int rc = flush(h);
    discard_ahead(h);   // (slot 0 is about to be reused)
    if (rc == CT_OK) {
        rc = prepare_batches(h, S);
    }
    if (rc == CT_OK && !dense_frames) {
        rc = ensure_frames(h, S, false);
    }
    if (rc != CT_OK) {
        return rc;
    }
    for (uint32_t c = 0; c < std::max(h->n_chunks, 1u) && rc == CT_OK; c++) {
        rc = submit_batch(h, 0, dense_frames, first, S, accumulate, false, c);
        if (rc == CT_OK) {
            rc = collect(h, h->slots[0]);
        }
    }
    if (rc != CT_OK) {
        return rc;
    }
    const bool simple = (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) != 0;
    if (!simple && !h->order_tuned && !dense_frames) {   // (a dense frame -- ct_render_subframe -- measures nothing)
        return tune_order(h, S);
    }
    return CT_OK;
}

extern "C" int ct_render_subframe(CtHandle h, uint32_t subframe_id, float *frame_rgba_dev)
{
    NEED(h);
    if (!h->camera_set) {
        return fail(h, CT_E_STATE, "ct_set_camera has not been called");
    }
    const size_t bytes = (size_t)h->scene.width * h->scene.height * sizeof(float4);
    // own pixels start as a miss (0,0,0,1); pixels of other shards stay (0,0,0,0)
    HIPCHK(h, launch_fill_frame(h->d_frame, h->scene.width, h->scene.height, h->scene.shard_index,
                                h->scene.shard_count, h->stream));
    const int rc = run_batch(h, h->d_frame, subframe_id, 1, false);
    if (rc != CT_OK) {
        return rc;
    }
    if (frame_rgba_dev) {
        HIPCHK(h, hipMemcpyAsync(frame_rgba_dev, h->d_frame, bytes, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return CT_OK;
}

extern "C" int ct_accumulate(CtHandle h, uint32_t subframe_id, const float *frame_rgba_dev)
{
    NEED(h);
    if (subframe_id == 0) {
        return fail(h, CT_E_INVAL, "subframe ids are 1-based (Camera.cpp:191)");
    }
    const float4 *src = frame_rgba_dev ? (const float4 *)frame_rgba_dev : h->d_frame;
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    // (no alpha check here: the caller may accumulate frames of its own)
    HIPCHK(h, launch_accumulate_batch(src, h->d_mean, h->d_m2, subframe_id, 1, h->scene.width, h->scene.height,
                                      h->scene.shard_index, h->scene.shard_count, nullptr, h->stop_cadence ? h->d_freeze : nullptr,
                                      h->stream));
    if (h->stop_cadence && subframe_id % h->stop_cadence == 0u && subframe_id >= h->stop_min) {
        const int rc = enqueue_convergence_test(h, subframe_id);
        if (rc != CT_OK) {
            return rc;
        }
    }
    HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev[2]));
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[1], h->ev[2]));
    h->accum_ms += ms;
    h->subframes = subframe_id;
    discard_ahead(h);
    return CT_OK;
}

// Subframes of the launch that measures the job costs of a new pose.
// (16 since round 3 -- 32 before: a waited-for launch of 32 subframes lasted 48 ms, most of it the wait for its longest paths.
// The order 16 subframes yield is as good as that of 32 -- long launches 349 ms either way, 10-subframe display updates 4.49 vs
// 4.52 ms -- while 8 subframes cost the display cadence 2-3 %: 4.61 ms, 3.70 instead of 3.62 with render-ahead.)
constexpr uint32_t kTuneSubframes = 16;
// (Also tried in round 3: the cost-measuring launch ENQUEUED -- it ends when its job list is empty, its survivors go to the next
// launch booking what they have cost so far, times 1..4 -- 18 ms instead of 26-48.  The order made from paths cut short is worse:
// the launches that follow take 352-359 ms per 1024 subframes instead of 349 for as long as the pose lasts, and the first image of
// a pose needs its longest path either way.  Removed.)
static uint32_t tune_subframes(CtHandle h)
{
    const char *e = h->tune.TUNE_SUBFRAMES.get();   // (A/B: how short may the cost-measuring launch be?)
    return e ? (uint32_t)std::min(1024, std::max(1, atoi(e))) : kTuneSubframes;
}

static int render_accumulate_impl(CtHandle h, uint32_t first_subframe_id, uint32_t count, bool wait)
{
    if (!h->camera_set) {
        return fail(h, CT_E_STATE, "ct_set_camera has not been called");
    }
    if (first_subframe_id != h->subframes + 1) {
        return fail(h, CT_E_STATE, "first_subframe_id %u but %u subframes are accumulated", first_subframe_id,
                    h->subframes);
    }
    if (count == 0) {
        return wait ? flush(h) : CT_OK;
    }
    const bool simple = (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) != 0;
    if (h->rendered > h->subframes) {
        // some of these subframes were rendered ahead of the calls: their samples wait in the scratch
        const uint32_t covered = std::min(count, h->rendered - h->subframes);
        h->subframes += covered;
        first_subframe_id += covered;
        count -= covered;
        if (count == 0) {
            const int rc = advance_accumulate(h, accumulate_target(h));
            return rc != CT_OK ? rc : (wait ? flush(h) : CT_OK);
        }
    }
    // (from here on every rendered subframe has been asked for)
    const uint32_t asked_end = h->subframes + count;
    if (!wait && h->ahead > count && !simple && h->continuation && h->order_tuned && !h->queue_dirty &&
        groups_per_chunk(h, h->ahead) >= h->n_groups) {
        count = h->ahead;   // the launch renders ahead of the calls; this call accumulates its own share
    }
    if (h->queue_dirty && !simple) {
        int rc = flush(h);
        if (rc == CT_OK) {
            rc = rebuild_queue(h);
        }
        if (rc != CT_OK) {
            return rc;
        }
    }
    // A call is cut by SUBFRAMES only where it must be (the subframe offset of a job has 16 bits); what does not fit the
    // scratch is cut by PIXEL GROUPS: every launch renders all the subframes of one chunk of groups (prepare_batches).
    const uint64_t cap = 0xffffull;
    uint32_t done = 0;
    // job lists are built for the size of the call's batches, not for the remainder that follows the short
    // cost-measuring launch (the next call of the same size would rebuild them)
    h->jobs_hint = (uint32_t)((count + (count + cap - 1) / cap - 1) / ((count + cap - 1) / cap));
    if (!simple) {
        // the scratch is sized for the call's batches now, not grown when the first of them follows the short cost-measuring
        // launch (freeing and allocating tens of GB takes hundreds of milliseconds)
        for (;;) {
            const uint32_t gc = groups_per_chunk(h, h->jobs_hint);
            // (reserved for a pose in which EVERY pixel of this shard hits the box, if a region holds that: the pixel list grows
            // and shrinks as the camera moves, and every growth of the scratch would be a hipFree + hipMalloc of tens of GB --
            // 25 ms when the driver has the pages at hand, a second when it has not)
            const uint64_t all_groups = (h->own_pixels + 63u) / 64u;
            const uint64_t region_groups = std::min<uint64_t>(scratch_slot_bytes(h) / sizeof(float4), 0xffffffffull / 2) / ((uint64_t)std::max(h->jobs_hint, 1u) * 64u);
            const uint32_t gc_reserve = (gc >= h->n_groups && all_groups <= region_groups) ? (uint32_t)std::max<uint64_t>(all_groups, gc) : gc;
            const size_t need = (size_t)h->jobs_hint * gc_reserve * 64;
            const bool enqueue = !wait || gc < h->n_groups;
            const uint64_t budget = 2 * scratch_slot_bytes(h) / sizeof(float4);
            const size_t total = std::min<size_t>((size_t)(enqueue ? wanted_regions(h, h->jobs_hint, need, std::max<uint64_t>(budget, 2 * need)) : 1) * need,
                                                  (0xffffffffull / std::max<size_t>(need, 1)) * need);
            if (total <= h->frames_total) {
                break;
            }
            int rc = flush(h);
            if (rc == CT_OK) {
                rc = reserve_frames(h, total);
            }
            if (rc == CT_E_NOMEM && scratch_slot_bytes(h) > (64ull << 20)) {
                h->scratch_cap_bytes = scratch_slot_bytes(h) / 2;   // the device cannot give that much: smaller chunks
                continue;
            }
            if (rc != CT_OK) {
                return rc;
            }
            break;
        }
    }
    while (done < count) {
        const uint64_t left = count - done, parts = (left + cap - 1) / cap;
        uint32_t S = (uint32_t)((left + parts - 1) / parts);
        int rc;
        if (!simple && !h->order_tuned) {
            // the first launch of a pose measures the job costs (two atomics per path, jobs in image
            // order): it is kept short, waited for, then the order is set
            // (a call of a display update's size is the cost-measuring launch as a whole: 10 subframes waited for cost 27 ms, 8 of
            // them 26 and the other 2 a launch and a flush of their own)
            S = std::min(S, h->jobs_hint <= tune_subframes(h) + tune_subframes(h) / 2u ? std::max(h->jobs_hint, 1u) : tune_subframes(h));
            rc = run_batch(h, nullptr, first_subframe_id + done, S, true);
        } else {
            const bool trace = h->tune.TRACE.get() != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            rc = prepare_batches(h, S);
            // several chunks: they are enqueued like batches (paths pass from chunk to chunk) and a waited-for call waits at the end
            const bool enqueue = !wait || (h->n_chunks > 1 && h->continuation && !simple);
            // (laid out for the call's batch size, not for the remainder that follows the short cost-measuring launch: the next
            // call of the same size then finds its layout, and nothing has to be flushed for it)
            uint32_t layout = S;
            if (enqueue && h->n_chunks <= 1 && 2 * (uint64_t)std::max(S, h->jobs_hint) * frame_stride(h) <= 0xffffffffull) {
                layout = std::max(S, h->jobs_hint);   // (one chunk only: a chunk's size was chosen for S)
            }
            const size_t need = (size_t)layout * frame_stride(h);
            // (an enqueued batch may use the regions of a layout made for larger ones, as long as the ring is long enough for it)
            const bool ring_too_short = enqueue && (h->n_regions < 2 ||
                                                    (layout != h->layout_S && h->n_regions < wanted_regions(h, layout, need, 2 * scratch_slot_bytes(h) / sizeof(float4))));
            if (rc == CT_OK && (need > h->slot_capacity || ring_too_short)) {
                rc = flush(h); // the layout of the scratch changes: nothing may be in flight
                if (rc == CT_OK) {
                    rc = ensure_frames(h, layout, enqueue);
                }
                if (rc == CT_E_NOMEM && scratch_slot_bytes(h) > (64ull << 20)) {
                    h->scratch_cap_bytes = scratch_slot_bytes(h) / 2;   // the device cannot give that much: smaller chunks
                    continue;
                }
            }
            const auto t1 = std::chrono::steady_clock::now();
            if (!enqueue) {
                rc = rc == CT_OK ? flush(h) : rc;
                h->next_slot = 0; // a synchronous batch uses one region and runs every path to its end
            }
            for (uint32_t c = 0; c < std::max(h->n_chunks, 1u) && rc == CT_OK; c++) {
                const int slot = h->next_slot;
                if (rc == CT_OK && h->slots[slot].awaits_accumulate) {
                    // the region's previous batch: with render-ahead its last share is due now at the latest
                    h->subframes = std::min(h->rendered + S, asked_end);
                    rc = advance_accumulate(h, accumulate_target(h));
                    h->subframes = std::min(h->rendered, asked_end);
                    if (rc == CT_OK && h->slots[slot].awaits_accumulate) {
                        rc = fail(h, CT_E_STATE, "internal: scratch region %d is reused before its batch is accumulated", slot);
                    }
                }
                rc = rc == CT_OK ? collect(h, h->slots[slot]) : rc; // book the launch that used this region n_regions launches ago
                const auto t2 = std::chrono::steady_clock::now();
                if (rc == CT_OK) {
                    const bool suspend = enqueue && h->continuation && !simple;
                    rc = submit_batch(h, slot, nullptr, first_subframe_id + done, S, true, suspend, c);
                }
                if (trace) {
                    const auto t3 = std::chrono::steady_clock::now();
                    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                    fprintf(stderr, "[cloudtrace] batch first=%u S=%u chunk %u of %u, region %d of %d: prepare %.2f ms, slot %.2f ms, submit %.2f ms\n",
                            first_subframe_id + done, S, c, h->n_chunks, slot, h->n_regions, ms(t0, t1), ms(t1, t2), ms(t2, t3));
                }
                if (enqueue) {
                    h->next_slot = (h->next_slot + 1) % h->n_regions;
                } else if (rc == CT_OK) {
                    rc = collect(h, h->slots[slot]);
                }
            }
        }
        if (rc != CT_OK) {
            flush(h);
            return rc;
        }
        done += S;
        h->rendered += S;
        h->subframes = std::min(h->rendered, asked_end);
        if (h->ahead && rc == CT_OK) {
            rc = advance_accumulate(h, accumulate_target(h));   // (the launch was enqueued before the count moved)
            if (rc != CT_OK) {
                flush(h);
                return rc;
            }
        }
    }
    return wait ? flush(h) : CT_OK;
}

extern "C" int ct_render_accumulate(CtHandle h, uint32_t first_subframe_id, uint32_t count)
{
    NEED(h);
    return render_accumulate_impl(h, first_subframe_id, count, true);
}

extern "C" int ct_render_accumulate_async(CtHandle h, uint32_t first_subframe_id, uint32_t count)
{
    NEED_NOFLUSH(h);
    return render_accumulate_impl(h, first_subframe_id, count, false);
}

extern "C" int ct_set_render_ahead(CtHandle h, uint32_t subframes)
{
    NEED(h);
    if (subframes > 0xffffu) {
        return fail(h, CT_E_INVAL, "render-ahead of %u subframes (a launch holds at most 65535)", subframes);
    }
    discard_ahead(h);
    h->ahead = subframes;
    return CT_OK;
}

extern "C" int ct_set_stop_when_converged(CtHandle h, uint32_t cadence, uint32_t min_subframes)
{
    NEED(h);
    if (cadence != 0u && h->scene.shard_count > 1u) {
        return fail(h, CT_E_INVAL, "stop-when-converged is a whole-frame decision: this handle renders shard %u of %u "
                                   "(test the merged frame with ct_is_converged_buffers)", h->scene.shard_index, h->scene.shard_count);
    }
    h->stop_cadence = cadence;
    h->stop_min = min_subframes;
    HIPCHK(h, hipMemsetAsync(h->d_freeze, 0, 8 * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memset(h->freeze_host, 0, 4 * sizeof(uint32_t));
    return CT_OK;
}

extern "C" int ct_converged_at(CtHandle h, uint32_t *subframes_out, uint32_t *tested_at_out, uint64_t *unconverged_pixels_out)
{
    NEED_NOFLUSH(h);
    // (written by the copies that follow the tests in stream order; read without waiting)
    const volatile uint32_t *st = h->freeze_host;
    if (subframes_out) {
        *subframes_out = st[0];
    }
    if (tested_at_out) {
        *tested_at_out = st[1];
    }
    if (unconverged_pixels_out) {
        *unconverged_pixels_out = st[2];
    }
    return CT_OK;
}

extern "C" int ct_rendered_subframes(CtHandle h, uint32_t *count_out)
{
    NEED_NOFLUSH(h);
    if (count_out) {
        *count_out = std::max(h->rendered, h->subframes);
    }
    return CT_OK;
}

extern "C" int ct_synchronize(CtHandle h)
{
    NEED(h);
    const int rc = flush(h);
    if (rc != CT_OK) {
        return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CT_OK;
}

static_assert(sizeof(CtPointRadianceTask) == 40, "Gpu::PointRadianceTask is 40 bytes (PointRadianceTask.h:70-77)");

extern "C" int ct_point_radiance_launch(CtHandle h, CtPointRadianceTask *tasks_host, uint32_t count,
                                        uint32_t first_frame_id, uint32_t launches)
{
    NEED(h);
    if (!tasks_host || count == 0 || launches == 0 || launches > 0xffffu || count > (1u << 24)) {
        return fail(h, CT_E_INVAL, "ct_point_radiance_launch: need 1..2^24 tasks and 1..65535 launches");
    }
    if (h->scene.flags & CT_FLAG_SIMPLE_KERNEL) {
        return fail(h, CT_E_INVAL, "point radiance tasks need the persistent kernel");
    }
    const uint32_t n_pad = (count + 63u) & ~63u, n_groups = n_pad / 64u;
    if ((uint64_t)n_pad * launches > 0xffffffffull) {
        return fail(h, CT_E_INVAL, "too many task-launches for one call");
    }
    // ---- job list.  A job = (group of 64 tasks, range of frames).  A collector's call is small -- 20480 tasks x 100
    // frames are 32000 wave-jobs of 64 experiments for the chip's 6144 resident waves -- and most of its work is in the few
    // tasks deep inside the cloud, whose every experiment runs to the depth cap (2000 bounces): jobs are single frames
    // unless the call is large, in one queue, frame-major, so that the deep groups are spread evenly over the list and over
    // the waves.  (What bounds such a call is the serial latency of those paths, ~6 us per bounce: DESIGN.md section 8 f-1.)
    // Several collectors at once (cloudtrace collect --jobs K: one handle and one host thread per scene setup) each launch a
    // persistent grid, and a grid that fills every CU leaves the others waiting for its last, longest paths.  A call that
    // finds others in flight launches a share of the grid instead -- blocks per CU divided by the calls in flight, at least
    // one -- so that the kernels are resident side by side: 8.72 -> 7.86 ms per update with four setups in flight, 7.76 with six
    // (profiles/r03y).  Alone, a call keeps the whole chip.  (The schedule only: results do not depend on the grid.)
    struct InFlight {
        std::atomic<int> &n;
        int mine;
        explicit InFlight(std::atomic<int> &c) : n(c), mine(c.fetch_add(1) + 1) {}
        ~InFlight() { n.fetch_sub(1); }
    };
    static std::atomic<int> point_calls_in_flight[64];   // (per device: cloudtrace collect --gpus deals the setups to several)
    const InFlight in_flight(point_calls_in_flight[(uint32_t)h->device & 63u]);
    LaunchShape shape = h->shape;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) {
            const int per_cu = std::max(1, shape.blocks / cus);
            int share = std::max(1, per_cu / std::max(1, in_flight.mine));
            if (const char *e = h->tune.POINT_BLOCKS_PER_CU.get()) {   // (A/B)
                share = std::min(per_cu, std::max(1, atoi(e)));
            }
            shape.blocks = share * cus;
        }
    }
    const uint32_t waves = (uint32_t)shape.blocks * (uint32_t)shape.threads / 64u;
    const uint32_t chunk = h->point_order
                               ? (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, (uint64_t)n_groups * launches / (16ull * std::max(waves, 1u))))
                               : 8u;
    std::vector<uint32_t> jg, js;
    jg.reserve((size_t)n_groups * ((launches + chunk - 1) / chunk));
    js.reserve(jg.capacity());
    if (h->point_order) {
        for (uint32_t s0 = 0; s0 < launches; s0 += chunk) {
            for (uint32_t g = 0; g < n_groups; g++) {
                jg.push_back(g);
                js.push_back(s0 | (std::min(chunk, launches - s0) << 16));
            }
        }
    } else {
        for (uint32_t g = 0; g < n_groups; g++) {
            for (uint32_t s0 = 0; s0 < launches; s0 += chunk) {
                jg.push_back(g);
                js.push_back(s0 | (std::min(chunk, launches - s0) << 16));
            }
        }
    }
    CtHandle_::PointBuffers &pt = h->pt;
    auto run = [&]() -> int {
        // (re)allocation waits for the device, so it happens only when a call is larger than every call before it
        const auto grow = [&](auto **p, size_t &cap, size_t need) -> hipError_t {
            if (need <= cap) {
                return hipSuccess;
            }
            if (*p) {
                hipError_t e = hipFree(*p);
                *p = nullptr;
                cap = 0;
                if (e != hipSuccess) {
                    return e;
                }
            }
            hipError_t e = hipMalloc((void **)p, need * sizeof(**p));
            if (e == hipSuccess) {
                cap = need;
            }
            return e;
        };
        HIPCHK(h, grow(&pt.tasks, pt.cap_tasks, count));
        HIPCHK(h, grow(&pt.primary, pt.cap_primary, 2 * (size_t)n_pad));
        HIPCHK(h, grow(&pt.pixels, pt.cap_pixels, n_pad));
        HIPCHK(h, grow(&pt.frames, pt.cap_frames, (size_t)n_pad * launches));
        HIPCHK(h, grow(&pt.jg, pt.cap_jg, jg.size()));
        HIPCHK(h, grow(&pt.js, pt.cap_js, js.size()));
        HIPCHK(h, hipMemcpyAsync(pt.tasks, tasks_host, (size_t)count * sizeof(CtPointRadianceTask),
                                 hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(pt.jg, jg.data(), jg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(pt.js, js.data(), js.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_queue, 0, kQueueWords * sizeof(uint32_t), h->stream));
        HIPCHK(h, launch_point_rays(h->dev, pt.tasks, count, n_pad, pt.primary, pt.pixels, h->stream));
        BatchArgs ba{};
        ba.frames = pt.frames;
        ba.frame_stride = n_pad;
        ba.primary = pt.primary;
        ba.pixels = pt.pixels;
        ba.job_group = pt.jg;
        ba.job_sub = pt.js;
        ba.cost = nullptr;
        ba.n_jobs = (uint32_t)jg.size();
        if (h->point_order) {
            // one queue: the order above is the order the chip's waves take the jobs in
            ba.q_begin[0] = 0;
            for (int x = 1; x <= kQueues + 1; x++) {
                ba.q_begin[x] = ba.n_jobs;
            }
        } else {
            split_queues_evenly(ba);
        }
        ba.first_subframe = first_frame_id;
        ba.S = launches;
        ba.queue = h->d_queue;
        ba.counters = h->d_counters;
        ba.stats = h->d_counters + kCounterCount + 1;
        HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
        if (h->scene.estimator == CT_EST_DELTA) {
            HIPCHK(h, launch_render_delta(h->dev, ba, shape, h->stream));
        } else {
            HIPCHK(h, launch_render_persistent(h->dev, ba, shape, h->stream));
        }
        HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
        HIPCHK(h, launch_point_accumulate(pt.frames, n_pad, pt.tasks, count, launches, h->stream));
        HIPCHK(h, hipMemcpyAsync(tasks_host, pt.tasks, (size_t)count * sizeof(CtPointRadianceTask),
                                 hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
        h->render_ms += ms;
        h->launches += 1;
        h->host_paths += (unsigned long long)count * launches;
        h->iv_expected_dealt += (unsigned long long)count * launches;
        if (pt.cap_frames > ((size_t)64 << 20)) {
            // an unusually large call (> 1 GiB of per-experiment results): do not keep that much for the handle's life
            HIPCHK(h, hipFree(pt.frames));
            pt.frames = nullptr;
            pt.cap_frames = 0;
        }
        return CT_OK;
    };
    const int rc = run();
    if (rc != CT_OK) {
        hipStreamSynchronize(h->stream);
    }
    return rc;
}

extern "C" int ct_generate_scatter_samples(CtHandle h, uint32_t count, uint32_t batch_seed,
                                           float *positions_host_out, float *directions_host_out)
{
    NEED(h);
    if (!positions_host_out || !directions_host_out || count == 0 || count > (1u << 20)) {
        return fail(h, CT_E_INVAL, "ct_generate_scatter_samples: need 1..2^20 samples and two output arrays");
    }
    float *d_pos = nullptr, *d_dir = nullptr;
    auto run = [&]() -> int {
        HIPCHK(h, dmalloc(&d_pos, 3 * (size_t)count));
        HIPCHK(h, dmalloc(&d_dir, 3 * (size_t)count));
        HIPCHK(h, launch_scatter_samples(h->dev, count, batch_seed, d_pos, d_dir, h->stream));
        HIPCHK(h, hipMemcpyAsync(positions_host_out, d_pos, 3 * (size_t)count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(directions_host_out, d_dir, 3 * (size_t)count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return CT_OK;
    };
    const int rc = run();
    if (rc != CT_OK) {
        hipStreamSynchronize(h->stream);
    }
    for (void *p : { (void *)d_pos, (void *)d_dir }) {
        if (p) {
            hipFree(p);
        }
    }
    return rc;
}

// Resources::generateMipmaps (Resources.cpp:169-209) on the device: levels = floor(log2(maxDim)) + 1.
static int ensure_pyramid(CtHandle h)
{
    if (h->d_pyramid) {
        return CT_OK;
    }
    const uint32_t nx = h->scene.dims[0], ny = h->scene.dims[1], nz = h->scene.dims[2];
    uint32_t m = std::max(nx, std::max(ny, nz)), levels = 1;
    while (m /= 2) {
        levels++;
    }
    if (levels > (uint32_t)kMaxMipLevels) {
        return fail(h, CT_E_INVAL, "volume too large for the mip pyramid");
    }
    MipPyramid mp{};
    mp.levels = levels;
    size_t total = 0;
    for (uint32_t l = 0; l < levels; l++) {
        mp.nx[l] = (int32_t)std::max(1u, nx >> l);
        mp.ny[l] = (int32_t)std::max(1u, ny >> l);
        mp.nz[l] = (int32_t)std::max(1u, nz >> l);
        mp.offset[l] = (uint32_t)total;
        total += (size_t)mp.nx[l] * mp.ny[l] * mp.nz[l];
    }
    if (total >= (1ull << 32)) {
        return fail(h, CT_E_INVAL, "volume too large for the mip pyramid");
    }
    HIPCHK(h, dmalloc(&h->d_pyramid, total));
    HIPCHK(h, hipMemcpyAsync(h->d_pyramid, h->d_density, (size_t)nx * ny * nz, hipMemcpyDeviceToDevice, h->stream));
    for (uint32_t l = 1; l < levels; l++) {
        HIPCHK(h, launch_mip_level(h->d_pyramid + mp.offset[l - 1], mp.nx[l - 1], mp.ny[l - 1], mp.nz[l - 1],
                                   h->d_pyramid + mp.offset[l], mp.nx[l], mp.ny[l], mp.nz[l], h->stream));
    }
    mp.base = h->d_pyramid;
    h->pyramid = mp;
    return CT_OK;
}

extern "C" int ct_collect_descriptors(CtHandle h, const float *positions_host, const float *directions_host,
                                      uint32_t count, uint8_t *descriptors_host_out)
{
    NEED(h);
    if (!positions_host || !directions_host || !descriptors_host_out || count == 0 || count > (1u << 20)) {
        return fail(h, CT_E_INVAL, "ct_collect_descriptors: need 1..2^20 samples, two input arrays and an output array");
    }
    const int prc = ensure_pyramid(h);
    if (prc != CT_OK) {
        return prc;
    }
    // VDBCloud::getVoxelSizeInMeters / getVoxelSizeInTermsOfFreePath (VDBCloud.cpp:35-46), DisneyDescriptor.cuh:83
    const float maxs = (float)std::max(h->scene.dims[0], std::max(h->scene.dims[1], h->scene.dims[2]));
    const float voxel_m = h->scene.cloud_size_m / maxs;
    const float voxel_fp = voxel_m / h->scene.mean_free_path_m;
    const float level0 = -ct_log2f(voxel_fp) - 1;
    float *d_pos = nullptr, *d_dir = nullptr;
    uint8_t *d_out = nullptr;
    auto run = [&]() -> int {
        HIPCHK(h, dmalloc(&d_pos, 3 * (size_t)count));
        HIPCHK(h, dmalloc(&d_dir, 3 * (size_t)count));
        HIPCHK(h, dmalloc(&d_out, (size_t)count * CT_DESCRIPTOR_BYTES));
        HIPCHK(h, hipMemcpyAsync(d_pos, positions_host, 3 * (size_t)count * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_dir, directions_host, 3 * (size_t)count * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, launch_descriptors(h->dev, h->pyramid, d_pos, d_dir, count, level0, voxel_m, h->scene.cloud_size_m,
                                     d_out, h->stream));
        HIPCHK(h, hipMemcpyAsync(descriptors_host_out, d_out, (size_t)count * CT_DESCRIPTOR_BYTES, hipMemcpyDeviceToHost,
                                 h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return CT_OK;
    };
    const int rc = run();
    if (rc != CT_OK) {
        hipStreamSynchronize(h->stream);
    }
    for (void *p : { (void *)d_pos, (void *)d_dir, (void *)d_out }) {
        if (p) {
            hipFree(p);
        }
    }
    return rc;
}

extern "C" int ct_reset(CtHandle h)
{
    NEED(h);
    const size_t pixels = (size_t)h->scene.width * h->scene.height;
    HIPCHK(h, hipMemsetAsync(h->d_frame, 0, pixels * sizeof(float4), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_mean, 0, pixels * sizeof(float4), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_m2, 0, pixels * sizeof(float4), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_counters, 0, (kCounterCount + 1 + kStatCount) * sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_freeze, 0, 8 * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memset(h->freeze_host, 0, 4 * sizeof(uint32_t));
    h->subframes = 0;
    discard_ahead(h);
    h->render_ms = h->accum_ms = 0;
    h->launches = 0;
    h->host_paths = h->host_hits = 0;
    h->iv_expected_dealt = 0;
    return CT_OK;
}

static int tonemap_impl(CtHandle h, const float4 *mean, float exposure, uint8_t *rgba_host, float *avg_luminance_out)
{
    HIPCHK(h, launch_reinhard(mean, h->scene.width, h->scene.height, exposure, h->d_colsum, h->d_avg,
                              h->d_screen, &h->reinhard_generation, h->device, h->stream));
    if (rgba_host) {
        HIPCHK(h, hipMemcpyAsync(rgba_host, h->d_screen, (size_t)h->scene.width * h->scene.height * sizeof(uchar4),
                                 hipMemcpyDeviceToHost, h->stream));
    }
    if (avg_luminance_out) {
        HIPCHK(h, hipMemcpyAsync(avg_luminance_out, h->d_avg, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CT_OK;
}

extern "C" int ct_tonemap(CtHandle h, float exposure, uint8_t *rgba_host, float *avg_luminance_out)
{
    NEED(h);
    return tonemap_impl(h, h->d_mean, exposure, rgba_host, avg_luminance_out);
}

extern "C" int ct_tonemap_async(CtHandle h, float exposure)
{
    NEED_NOFLUSH(h);
    // (behind whatever is enqueued: the running mean of the batches whose accumulate kernels precede it on the stream)
    HIPCHK(h, launch_reinhard(h->d_mean, h->scene.width, h->scene.height, exposure, h->d_colsum, h->d_avg, h->d_screen, &h->reinhard_generation,
                              h->device, h->stream));
    return CT_OK;
}

extern "C" int ct_tonemap_buffer(CtHandle h, const float *mean_rgba_dev, float exposure, uint8_t *rgba_host,
                                 float *avg_luminance_out)
{
    NEED(h);
    if (!mean_rgba_dev) {
        return fail(h, CT_E_INVAL, "mean_rgba_dev is NULL");
    }
    return tonemap_impl(h, (const float4 *)mean_rgba_dev, exposure, rgba_host, avg_luminance_out);
}

static int converged_impl(CtHandle h, const float4 *mean, const float4 *m2, uint32_t subframes, int32_t *converged_out,
                          uint64_t *unconverged_pixels_out)
{
    if (!converged_out) {
        return fail(h, CT_E_INVAL, "converged_out is NULL");
    }
    const uint64_t pixels = (uint64_t)h->scene.width * h->scene.height;
    if (subframes < 100) { // Camera.cpp:234-237
        *converged_out = 0;
        if (unconverged_pixels_out) {
            *unconverged_pixels_out = pixels;
        }
        return CT_OK;
    }
    unsigned long long *d_cnt = h->d_counters + kCounterCount;
    HIPCHK(h, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), h->stream));
    HIPCHK(h, launch_converged(mean, m2, subframes, pixels, d_cnt, h->stream));
    unsigned long long bad = 0;
    HIPCHK(h, hipMemcpyAsync(&bad, d_cnt, sizeof bad, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *converged_out = bad < 500 ? 1 : 0; // Camera.cpp:267
    if (unconverged_pixels_out) {
        *unconverged_pixels_out = bad;
    }
    return CT_OK;
}

extern "C" int ct_is_converged(CtHandle h, int32_t *converged_out, uint64_t *unconverged_pixels_out)
{
    NEED(h);
    // (a running mean that stop-when-converged froze is the image after that many subframes, whatever was asked for since)
    const uint32_t frozen_at = h->stop_cadence ? h->freeze_host[0] : 0u;
    return converged_impl(h, h->d_mean, h->d_m2, frozen_at ? frozen_at : h->subframes, converged_out, unconverged_pixels_out);
}

extern "C" int ct_is_converged_buffers(CtHandle h, const float *mean_rgba_dev, const float *m2_rgba_dev, uint32_t subframes,
                                       int32_t *converged_out, uint64_t *unconverged_pixels_out)
{
    NEED(h);
    if (!mean_rgba_dev || !m2_rgba_dev) {
        return fail(h, CT_E_INVAL, "mean_rgba_dev / m2_rgba_dev is NULL");
    }
    return converged_impl(h, (const float4 *)mean_rgba_dev, (const float4 *)m2_rgba_dev, subframes, converged_out,
                          unconverged_pixels_out);
}

static int buffer_info(CtHandle h, int32_t which, void **ptr, size_t *bytes)
{
    const size_t pixels = (size_t)h->scene.width * h->scene.height;
    switch (which) {
    case CT_BUF_MEAN: *ptr = h->d_mean; *bytes = pixels * sizeof(float4); return CT_OK;
    case CT_BUF_M2: *ptr = h->d_m2; *bytes = pixels * sizeof(float4); return CT_OK;
    case CT_BUF_FRAME: *ptr = h->d_frame; *bytes = pixels * sizeof(float4); return CT_OK;
    case CT_BUF_SCREEN: *ptr = h->d_screen; *bytes = pixels * sizeof(uchar4); return CT_OK;
    case CT_BUF_INSCATTER: *ptr = h->d_inscatter; *bytes = h->volume_bytes; return CT_OK;
    case CT_BUF_DENSITY: *ptr = h->d_density; *bytes = h->volume_bytes; return CT_OK;
    default: return fail(h, CT_E_INVAL, "unknown buffer %d", which);
    }
}

extern "C" int ct_buffer_bytes(CtHandle h, int32_t which, size_t *bytes_out)
{
    NEED(h);
    void *p;
    size_t b = 0;
    const int rc = buffer_info(h, which, &p, &b);
    if (rc == CT_OK && bytes_out) {
        *bytes_out = b;
    }
    return rc;
}

extern "C" int ct_download(CtHandle h, int32_t which, void *dst_host, size_t dst_bytes)
{
    NEED(h);
    void *p;
    size_t b = 0;
    const int rc = buffer_info(h, which, &p, &b);
    if (rc != CT_OK) {
        return rc;
    }
    if (!dst_host || dst_bytes != b) {
        return fail(h, CT_E_INVAL, "ct_download: need %zu bytes, got %zu", b, dst_bytes);
    }
    HIPCHK(h, hipMemcpyAsync(dst_host, p, b, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CT_OK;
}

// The image changes under the convergence rule's feet (ct_upload, ct_set_subframes: a checkpoint is loaded): whatever
// ct_set_stop_when_converged had frozen was the OLD image -- left in place, the flag would make every later accumulate
// kernel return early and the new image's samples would be dropped without a word.  As in ct_reset.
static int clear_freeze(CtHandle h)
{
    HIPCHK(h, hipMemsetAsync(h->d_freeze, 0, 8 * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memset(h->freeze_host, 0, 4 * sizeof(uint32_t));
    return CT_OK;
}

extern "C" int ct_upload(CtHandle h, int32_t which, const void *src_host, size_t src_bytes)
{
    NEED(h);
    if (which != CT_BUF_MEAN && which != CT_BUF_M2) {
        return fail(h, CT_E_INVAL, "ct_upload: only the running mean and M2 can be set (buffer %d)", which);
    }
    void *p;
    size_t b = 0;
    const int rc = buffer_info(h, which, &p, &b);
    if (rc != CT_OK) {
        return rc;
    }
    if (!src_host || src_bytes != b) {
        return fail(h, CT_E_INVAL, "ct_upload: need %zu bytes, got %zu", b, src_bytes);
    }
    discard_ahead(h);   // (what was rendered ahead belongs to the image that is being replaced)
    HIPCHK(h, hipMemcpyAsync(p, src_host, b, hipMemcpyHostToDevice, h->stream));
    return clear_freeze(h);
}

extern "C" int ct_copy_to_device(CtHandle h, int32_t which, void *dst_dev, size_t dst_bytes)
{
    NEED(h);
    void *p;
    size_t b = 0;
    const int rc = buffer_info(h, which, &p, &b);
    if (rc != CT_OK) {
        return rc;
    }
    if (!dst_dev || dst_bytes != b) {
        return fail(h, CT_E_INVAL, "ct_copy_to_device: need %zu bytes, got %zu", b, dst_bytes);
    }
    HIPCHK(h, hipMemcpyAsync(dst_dev, p, b, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CT_OK;
}

extern "C" int ct_copy_to_device_async(CtHandle h, int32_t which, void *dst_dev, size_t dst_bytes)
{
    NEED_NOFLUSH(h);
    void *p;
    size_t b = 0;
    const int rc = buffer_info(h, which, &p, &b);
    if (rc != CT_OK) {
        return rc;
    }
    if (!dst_dev || dst_bytes != b) {
        return fail(h, CT_E_INVAL, "ct_copy_to_device_async: need %zu bytes, got %zu", b, dst_bytes);
    }
    HIPCHK(h, hipMemcpyAsync(dst_dev, p, b, hipMemcpyDeviceToDevice, h->stream));
    return CT_OK;
}

extern "C" int ct_device_ptr(CtHandle h, int32_t which, void **ptr_out)
{
    NEED(h);
    size_t b = 0;
    if (!ptr_out) {
        return fail(h, CT_E_INVAL, "ptr_out is NULL");
    }
    return buffer_info(h, which, ptr_out, &b);
}

extern "C" int ct_subframes(CtHandle h, uint32_t *count_out)
{
    NEED_NOFLUSH(h); // host-side count of the subframes submitted so far
    if (count_out) {
        *count_out = h->subframes;
    }
    return CT_OK;
}

extern "C" int ct_set_subframes(CtHandle h, uint32_t count)
{
    NEED(h);
    h->subframes = count;
    discard_ahead(h);
    return clear_freeze(h);
}

extern "C" int ct_counters(CtHandle h, CtCounters *out)
{
    NEED(h);
    if (!out) {
        return fail(h, CT_E_INVAL, "out is NULL");
    }
    unsigned long long c[kCounterCount];
    HIPCHK(h, hipMemcpyAsync(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out->paths = c[0] + h->host_paths;
    out->box_hits = c[1] + h->host_hits;
    out->density_lookups = c[2];
    out->inscatter_lookups = c[3];
    out->scatter_events = c[4];
    out->depth_capped = c[5];
    return CT_OK;
}

extern "C" int ct_fetch_counters(CtHandle h, CtFetchCounters *out)
{
    NEED(h);
    if (!out) {
        return fail(h, CT_E_INVAL, "out is NULL");
    }
    unsigned long long c[kCounterCount];
    HIPCHK(h, hipMemcpyAsync(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out->density_fetches = c[6];
    out->inscatter_fetches = c[7];
    return CT_OK;
}

extern "C" int ct_debug_invariants(CtHandle h, uint64_t out[8])
{
    NEED(h);
    if (!out) {
        return fail(h, CT_E_INVAL, "out is NULL");
    }
    unsigned long long iv[4] = { 0, 0, 0, 0 }, bad = 0;
    HIPCHK(h, hipMemcpyAsync(iv, h->d_counters + kCounterCount + 1 + 64, sizeof iv, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&bad, h->d_counters + 8, sizeof bad, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out[0] = h->debug_invariants ? 1 : 0;
    out[1] = h->iv_checks;
    out[2] = h->iv_violations;
    out[3] = bad;
    out[4] = iv[0];
    out[5] = iv[1];
    out[6] = iv[2];
    out[7] = iv[3];
    return CT_OK;
}

extern "C" int ct_kernel_time(CtHandle h, double *render_ms_out, double *accumulate_ms_out, uint64_t *launches_out)
{
    NEED(h);
    if (render_ms_out) {
        *render_ms_out = h->render_ms;
    }
    if (accumulate_ms_out) {
        *accumulate_ms_out = h->accum_ms;
    }
    if (launches_out) {
        *launches_out = h->launches;
    }
    return CT_OK;
}

extern "C" int ct_debug_stats(CtHandle h, uint64_t out[64])
{
    NEED(h);
    if (!out) {
        return fail(h, CT_E_INVAL, "out is NULL");
    }
    unsigned long long c[kStatCount];
    HIPCHK(h, hipMemcpyAsync(c, h->d_counters + kCounterCount + 1, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 64; i++) { // (the conservation tallies behind them: ct_debug_invariants)
        out[i] = c[i];
    }
    return CT_OK;
}

extern "C" int ct_debug_stats_ex(CtHandle h, uint64_t *out, uint32_t count)
{
    NEED(h);
    if (!out || count > (uint32_t)kStatCount) {
        return fail(h, CT_E_INVAL, "out is NULL or count > %d", kStatCount);
    }
    unsigned long long c[kStatCount];
    HIPCHK(h, hipMemcpyAsync(c, h->d_counters + kCounterCount + 1, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (uint32_t i = 0; i < count; i++) {
        out[i] = c[i];
    }
    return CT_OK;
}

// The estimator's working set: which 128-B lines of the density array it reads (march bricks for MARCH; apron or twin bricks
// for DELTA) and of the shadow volume's apron bricks a launch fetches from.  Needs the diagnostics kernels (CT_STATS=1 or
// CT_DEBUG_INVARIANTS=1 at ct_create).  enable != 0 allocates and clears the two bitmaps; 0 frees them.
extern "C" int ct_debug_track_lines(CtHandle h, int32_t enable)
{
    NEED(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 2; k++) {
        if (h->d_touched[k]) {
            HIPCHK(h, hipFree(h->d_touched[k]));
            h->d_touched[k] = nullptr;
        }
    }
    if (!enable) {
        return CT_OK;
    }
    if (!h->shape.stats) {
        return fail(h, CT_E_STATE, "ct_debug_track_lines needs the diagnostics kernels (CT_STATS=1 at ct_create)");
    }
    if (h->dev.m_rows) {
        return fail(h, CT_E_STATE, "ct_debug_track_lines: not implemented for sparse march bricks");
    }
    const size_t apron_lines = (size_t)h->dev.brick_gxy * (size_t)h->dev.brick_gz;
    size_t density_lines = apron_lines;
    if (h->scene.estimator == CT_EST_MARCH) {
        density_lines = h->mbricks_dense_bytes / 128;
    } else if (h->dev.delta_nee == 2u && h->dev.tbricks) {
        density_lines = (size_t)h->dev.t_gx * h->dev.t_gy * h->dev.t_gz;
    }
    h->touched_lines[0] = density_lines;
    h->touched_lines[1] = apron_lines;
    for (int k = 0; k < 2; k++) {
        const size_t words = (h->touched_lines[k] + 31) / 32;
        HIPCHK(h, dmalloc(&h->d_touched[k], words));
        HIPCHK(h, hipMemsetAsync(h->d_touched[k], 0, words * sizeof(uint32_t), h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CT_OK;
}

// out[0], out[1] = distinct lines touched in the density / shadow arrays since ct_debug_track_lines(h, 1) (or since the last
// call with clear != 0), out[2], out[3] = the arrays' sizes in lines.  Waits for the batches in flight.
extern "C" int ct_debug_touched_lines(CtHandle h, uint64_t out[4], int32_t clear)
{
    NEED(h);
    if (!out || !h->d_touched[0]) {
        return fail(h, CT_E_STATE, "ct_debug_touched_lines: call ct_debug_track_lines(h, 1) first");
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 2; k++) {
        const size_t words = (h->touched_lines[k] + 31) / 32;
        std::vector<uint32_t> bits(words);
        HIPCHK(h, hipMemcpy(bits.data(), h->d_touched[k], words * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t n = 0;
        for (uint32_t w : bits) {
            n += (uint64_t)__builtin_popcount(w);
        }
        out[k] = n;
        out[2 + k] = h->touched_lines[k];
        if (clear) {
            HIPCHK(h, hipMemset(h->d_touched[k], 0, words * sizeof(uint32_t)));
        }
    }
    return CT_OK;
}

extern "C" int ct_debug_memory(CtHandle h, uint64_t out[8])
{
    NEED(h);
    if (!out) {
        return fail(h, CT_E_INVAL, "out is NULL");
    }
    const size_t brick_bytes = (size_t)h->dev.brick_gxy * (size_t)h->dev.brick_gz * 128;
    out[0] = h->volume_bytes;
    out[1] = brick_bytes;                 // density apron bricks
    out[2] = brick_bytes;                 // shadow-volume apron bricks
    out[3] = h->mbricks_dense_bytes;
    out[4] = h->mbricks_bytes;
    out[5] = h->dev.m_rows ? 1 : (h->vmm.va ? 2 : 0);   // 1 row extents, 2 dense addressing with sparse backing (out[4] = the memory behind it)
    out[6] = h->dev.m_rows ? (size_t)h->dev.brick_gy * h->dev.brick_gz * sizeof(uint2) : 0;
    out[7] = h->dev.m_rows ? (size_t)h->dev.m_cgxy * (size_t)((4 * h->dev.brick_gz + 7) >> h->dev.m_cshift) : 0;
    return CT_OK;
}

extern "C" int ct_debug_delta_grid(CtHandle h, uint32_t out[8])
{
    NEED(h);
    if (!out) {
        return fail(h, CT_E_INVAL, "out is NULL");
    }
    const DevScene &d = h->dev;
    const bool delta = d.maj_cells != nullptr;
    out[0] = delta ? (uint32_t)d.mc_cell : 0u;
    out[1] = delta ? (uint32_t)d.mc_gx : 0u;
    out[2] = delta ? (uint32_t)d.mc_gy : 0u;
    out[3] = delta ? (uint32_t)d.mc_gz : 0u;
    out[4] = delta ? (uint32_t)d.mc_x0 : 0u;
    out[5] = delta ? (uint32_t)d.mc_y0 : 0u;
    out[6] = delta ? (uint32_t)d.mc_z0 : 0u;
    out[7] = delta ? (d.delta_nee | (d.delta_interior ? 0x100u : 0u)) : 0u;
    return CT_OK;
}

extern "C" int ct_debug_timeline(CtHandle h, uint64_t *out, uint32_t waves)
{
    NEED(h);
    const size_t have = (size_t)h->shape.blocks * h->shape.threads / 64u;
    if (!h->d_timeline || !out || waves > have) {
        return fail(h, CT_E_INVAL, "ct_debug_timeline: CT_TIMELINE=1 at ct_create, at most %zu waves", have);
    }
    HIPCHK(h, hipMemcpy(out, h->d_timeline, 4 * (size_t)waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CT_OK;
}

extern "C" int ct_debug_suspended(CtHandle h, uint64_t *paths_out)
{
    NEED(h);
    if (!paths_out) {
        return fail(h, CT_E_INVAL, "paths_out is NULL");
    }
    unsigned long long n = 0;
    HIPCHK(h, hipMemcpy(&n, h->d_cont_total, sizeof n, hipMemcpyDeviceToHost));
    *paths_out = n;
    return CT_OK;
}

extern "C" int ct_debug_fetch_probe(int32_t device, uint32_t log2_lines, uint32_t repeats, uint64_t *sum_out)
{
    if (log2_lines < 10 || log2_lines > 28 || hipSetDevice(device) != hipSuccess) {
        return fail(nullptr, CT_E_INVAL, "ct_debug_fetch_probe: bad device or size");
    }
    uint8_t *buf = nullptr;
    unsigned long long *sum = nullptr;
    const size_t bytes = (size_t)128 << log2_lines;
    if (hipMalloc((void **)&buf, bytes) != hipSuccess || hipMalloc((void **)&sum, 8) != hipSuccess) {
        if (buf) {
            hipFree(buf);
        }
        return fail(nullptr, CT_E_NOMEM, "ct_debug_fetch_probe: out of device memory");
    }
    hipMemset(buf, 0, bytes);
    hipMemset(sum, 0, 8);
    hipError_t e = hipSuccess;
    for (uint32_t r = 0; r < repeats && e == hipSuccess; r++) {
        e = launch_fetch_probe(buf, log2_lines, (r & 1u) ? 72u : 25u, sum, nullptr); // odd repeats touch both 64-B halves
    }
    if (e == hipSuccess) {
        e = hipDeviceSynchronize();
    }
    unsigned long long v = 0;
    hipMemcpy(&v, sum, 8, hipMemcpyDeviceToHost);
    hipFree(buf);
    hipFree(sum);
    if (sum_out) {
        *sum_out = v;
    }
    return e == hipSuccess ? CT_OK : fail(nullptr, CT_E_HIP, "fetch probe failed: %s", hipGetErrorString(e));
}

// The same access shape over a WORKING SET: 2^log2_threads lanes each read one pseudo-random line out of `ws_lines` lines, so
// a set smaller than a cache level is re-read from that level (2^25 lanes over 2^20 lines = 128 MiB: every line 32 times per
// launch, far apart in time) -- the line-fill ceiling of the level the set fits in (L2 4 MiB per XCD, Infinity Cache 256 MiB,
// HBM beyond).  `repeats` launches; time two calls with different repeats and take the difference.
extern "C" int ct_debug_fetch_probe_ws(int32_t device, uint32_t log2_threads, uint64_t ws_lines, uint32_t repeats, uint64_t *sum_out)
{
    if (log2_threads < 10 || log2_threads > 28 || ws_lines < 1024 || ws_lines > (1ull << 28) || hipSetDevice(device) != hipSuccess) {
        return fail(nullptr, CT_E_INVAL, "ct_debug_fetch_probe_ws: bad device or size");
    }
    uint8_t *buf = nullptr;
    unsigned long long *sum = nullptr;
    const size_t bytes = (size_t)128 * ws_lines;
    if (hipMalloc((void **)&buf, bytes) != hipSuccess || hipMalloc((void **)&sum, 8) != hipSuccess) {
        if (buf) {
            hipFree(buf);
        }
        return fail(nullptr, CT_E_NOMEM, "ct_debug_fetch_probe_ws: out of device memory");
    }
    hipMemset(buf, 0, bytes);
    hipMemset(sum, 0, 8);
    hipError_t e = hipSuccess;
    for (uint32_t r = 0; r < repeats && e == hipSuccess; r++) {
        e = launch_fetch_probe_ws(buf, log2_threads, (uint32_t)ws_lines, r, sum, nullptr);
    }
    if (e == hipSuccess) {
        e = hipDeviceSynchronize();
    }
    unsigned long long v = 0;
    hipMemcpy(&v, sum, 8, hipMemcpyDeviceToHost);
    hipFree(buf);
    hipFree(sum);
    if (sum_out) {
        *sum_out = v;
    }
    return e == hipSuccess ? CT_OK : fail(nullptr, CT_E_HIP, "fetch probe failed: %s", hipGetErrorString(e));
}

extern "C" int ct_debug_math_selftest(CtHandle h, int32_t which, uint64_t out[3])
{
    NEED(h);
    if (!out || (which != 0 && which != 1)) {
        return fail(h, CT_E_INVAL, "ct_debug_math_selftest: which = 0 (reciprocal) or 1 (square root)");
    }
    // every float with 2^-60 <= x < 2^61 (positive: both functions are odd / undefined for negative arguments in the same way
    // as the IEEE operations, and the kernels only pass positive values); the reciprocal also for the negative range
    unsigned long long *d = nullptr;
    HIPCHK(h, hipMalloc((void **)&d, 3 * sizeof(unsigned long long)));
    const unsigned long long init[3] = { 0, 0, 0xffffffffull };
    hipError_t e = hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, h->stream);
    const uint32_t lo = (127u - 60u) << 23, hi = ((127u + 61u) << 23) - 1u;
    if (e == hipSuccess) {
        e = launch_math_selftest(which, lo, hi, d, h->stream);
    }
    if (e == hipSuccess && which == 0) {
        e = launch_math_selftest(which, lo | 0x80000000u, hi | 0x80000000u, d, h->stream);
    }
    unsigned long long r[3] = { 0, 0, 0 };
    if (e == hipSuccess) {
        e = hipMemcpyAsync(r, d, sizeof r, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) {
        e = hipStreamSynchronize(h->stream);
    }
    hipFree(d);
    HIPCHK(h, e);
    out[0] = r[0];
    out[1] = r[1];
    out[2] = r[2];
    return CT_OK;
}

extern "C" int ct_debug_cdf_inversion(CtHandle h, uint32_t first_u24, uint32_t count, uint32_t *k_host_out)
{
    NEED(h);
    if (!k_host_out || count == 0 || (uint64_t)first_u24 + count > (1ull << 24)) {
        return fail(h, CT_E_INVAL, "range must lie inside [0, 2^24)");
    }
    uint32_t *d_k = nullptr;
    HIPCHK(h, dmalloc(&d_k, count));
    hipError_t e = launch_cdf_selftest(h->d_cdf, h->d_guide, first_u24, count, d_k, h->stream);
    if (e == hipSuccess) {
        e = hipMemcpyAsync(k_host_out, d_k, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) {
        e = hipStreamSynchronize(h->stream);
    }
    hipFree(d_k);
    HIPCHK(h, e);
    return CT_OK;
}

extern "C" uint32_t ct_tile_owner(uint32_t tile_x, uint32_t tile_y, uint32_t shard_count)
{
    return tile_owner(tile_x, tile_y, shard_count);
}
