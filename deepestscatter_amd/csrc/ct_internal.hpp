// ct_internal.hpp -- launcher declarations shared by ct_kernels.hip and ct_api.cpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ct_device.hpp"

namespace ct {

constexpr int kTile = 8;            // pixel tile edge: 8x8 = one wave of primary rays
constexpr int kCounterCount = 9;    // paths, box_hits, density, inscatter, scatter, capped (the algorithm's, = the oracle's);
                                    // then what the kernels ISSUED: density fetches, shadow-volume fetches (ct_fetch_counters),
                                    // and the samples the accumulate kernels found without alpha == 1 (ct_debug_invariants)
constexpr int kStatCount = 72;      // scheduler diagnostics (ct_debug_stats): [0,64) as before, [64,68) path conservation
                                    // (samples dealt, paths resumed, results written, paths suspended; STATS kernels only)
constexpr int kContWords = 16;      // words of a suspended path (render_persistent_kernel)
constexpr int kContWordsDelta = 32; // the same for render_delta_kernel (its DDA state rides along)
constexpr int kLeftWords = 8;        // words of a job handed to the next launch (BatchArgs::left_out)
constexpr int kQueueFlag = 32;       // word of the queue array (its own 128-B line) that says "job list empty"
constexpr int kQueueDone = 16;       // ... and the word that counts the queues whose last job has been taken
constexpr int kQueueWords = 64;      // size of a queue array
constexpr int kQueues = 8;          // one job queue per XCD (MI355X: 8 XCDs, each with its own L2)

// One progressive batch: subframes first .. first+S-1 of the handle's own tiles.
struct BatchArgs {
    // per-sample results (frameResultBuffer x S).  Compact form: frames[s * frame_stride + e] for
    // entry e of the pixel list (only this shard's box-hitting pixels exist); dense form
    // (frame_stride == 0): frames[pixel], one full W x H frame, used by ct_render_subframe.
    float4 *frames;
    uint32_t frame_stride;
    const float4 *primary;     // 2 float4 per pixel: cached primary ray (primary_rays_kernel)
    const float4 *advance;     // per pixel: state after the pre-walked prefix of the primary flight: 1 float4
                               // (MARCH, primary_advance_kernel) or 4 (DELTA, primary_advance_delta_kernel);
                               // NULL = start at the entry point
    const uint32_t *pixels;    // this shard's box-hitting pixels, 64 per group, 0xffffffff padded
    // A launch may cover only a CHUNK of the pixel groups (all S subframes of each): the scratch then has a column of 64
    // entries per group of the chunk, group g's at (group_rank[g] - rank_base) * 64 -- group_rank = the group's place in the
    // cost-sorted order the job list is built in, of which a chunk is a contiguous piece.  NULL: column g * 64 (every group).
    const uint32_t *group_rank;
    uint32_t rank_base;
    // job list: job j renders subframes [begin, begin+count) of pixel group job_group[j];
    // job_sub[j] = begin | count << 16.  Sorted most expensive group first, and expensive groups
    // are cut into short jobs, so the long paths start early and spread over all waves while
    // every wave still stays on one 64-pixel group at a time (cache locality).
    const uint32_t *job_group;
    const uint32_t *job_sub;
    uint2 *cost;               // the cost-measuring launch of a pose (NULL once the job order is tuned): what every path has
                               // cost (MARCH: fetches + 4 per bounce; DELTA: bounces) and how deep it went, stored where its
                               // result goes -- cost[out_idx - out_offset] -- and summed per pixel group by
                               // launch_cost_reduce.  (Until round 3 two atomics per path on per-group words: 17.6 M
                               // device-scope atomics were 5.7 of the 21 ms of a pose's first 10 subframes.)
    uint32_t n_jobs;
    // Queue x < kQueues holds jobs [q_begin[x], q_begin[x+1]): the pixel groups of one compact image
    // region, so that the waves of one XCD share that region's bricks in their L2.  Queue kQueues
    // is shared: the jobs of the deepest groups, which every wave of the chip must start on at
    // once (a launch cannot be shorter than its deepest path).  A wave drains the shared queue,
    // then the queue of the XCD it runs on, then the following ones (work stealing).
    uint32_t q_begin[kQueues + 2];
    uint32_t reverse;          // 1 = every queue is walked from its end (short launches alternate: the sweep over the image
                               // turns round where the previous launch stopped, whose paths this one resumes)
    uint32_t first_subframe;   // 1-based subframeId of slice 0
    uint32_t S;
    uint32_t *queue;           // kQueueWords words, zero before launch: kQueues + 1 work counters, the flag
    // Path continuation (MARCH estimator, batches enqueued with ct_render_accumulate_async): when the job
    // queue is empty a wave does not run its surviving paths to their end -- a launch would end with a
    // tail of waves that carry a few long paths each, 19 ms of 107 at 256 subframes -- it writes their
    // state (kContWords words each) to cont_out and exits; the next launch resumes them from cont_in
    // before it takes jobs.  A resumed path is never suspended again and writes its result where it
    // always would: out_offset selects the half of the scratch that belongs to its batch.
    const uint32_t *cont_in;        // NULL: nothing to resume
    const uint32_t *cont_in_count;  // entries in cont_in (device memory: written by the previous launch)
    uint32_t *cont_cursor;          // zero before launch
    uint32_t *cont_out;             // NULL: run every path to its end
    uint32_t *cont_out_count;       // zero before launch
    uint32_t cont_capacity;         // entries cont_out can hold
    uint32_t max_age;               // a path that has been suspended this often is run to its end (>= 1 when cont_out is set):
                                    // the host accumulates a batch once max_age further launches have run
    unsigned long long *cont_total; // running total of suspended paths (diagnostic), or NULL
    // Jobs handed on like paths: a wave that learns that the job list is empty while it still has samples of its own job
    // to start (its lanes are busy with long paths) does not hold the launch up until lanes come free -- it writes the
    // rest of the job to left_out (kLeftWords words: group, first subframe of the job, next and end sample, the batch's
    // scratch offset and first subframe id, the age its samples start with) and the next launch's waves take those
    // before the job list.  Same samples, same seeds, same result indices: only the launch that runs them changes.
    const uint32_t *left_in;        // NULL: none
    const uint32_t *left_in_count;
    uint32_t *left_cursor;          // zero before launch
    uint32_t *left_out;             // NULL: every wave finishes its job
    uint32_t *left_out_count;       // zero before launch
    uint32_t left_capacity;
    uint32_t out_offset;            // added to a compact result index (frame_stride != 0)
    unsigned long long *counters; // kCounterCount
    unsigned long long *stats;    // kStatCount
    unsigned long long *timeline; // diagnostics (CT_TIMELINE=1), else NULL: per wave [start, end] on the 100 MHz wall clock
    // diagnostics (ct_debug_track_lines; STATS kernels only), else NULL: one bit per 128-B line of the density array the
    // estimator reads (march bricks / apron bricks / twin bricks) and of the shadow volume's apron bricks, set when a launch
    // fetches from the line -- the kernel's working set, to be held against the 256 MiB of the Infinity Cache
    uint32_t *touched_density, *touched_shadow;
};

struct LaunchShape {
    int blocks;
    int threads;
    bool stats;   // launch the diagnostics build of the kernel (CT_STATS / CT_DEBUG_INVARIANTS at ct_create)
    uint32_t pool_slots = 0;   // exchange kernels (ct_exchange.hpp): slots of the block's path pool in LDS
    uint32_t scatter_waves = 0; // ... and how many of a block's 16 waves only run scatter batches
};

// Evenly split job list (no locality information): point tasks, first launches.
inline void split_queues_evenly(BatchArgs &ba)
{
    for (int x = 0; x <= kQueues; x++) {
        ba.q_begin[x] = (uint32_t)(((uint64_t)ba.n_jobs * (uint32_t)x) / kQueues);
    }
    ba.q_begin[kQueues + 1] = ba.n_jobs; // empty shared queue
}

// Tile -> shard map (also exported as ct_tile_owner).
#ifndef CT_SHARD_SHIFT
#define CT_SHARD_SHIFT 0   // experiment: interleave blocks of 2^k x 2^k tiles instead of single tiles
#endif
__host__ __device__ inline uint32_t tile_owner(uint32_t tx, uint32_t ty, uint32_t shard_count)
{
    return shard_count <= 1 ? 0u : ((tx >> CT_SHARD_SHIFT) + 3u * (ty >> CT_SHARD_SHIFT)) % shard_count;
}

hipError_t launch_build_bricks(const uint8_t *texels, int nx, int ny, int nz, int bias, int gx, int gy, int gz,
                               uint8_t *bricks, hipStream_t stream);
hipError_t launch_build_twin_bricks(const uint8_t *density, const uint8_t *shadow, int nx, int ny, int nz, int bias, int gx, int gy, int gz,
                                    uint8_t *bricks, hipStream_t stream);
hipError_t launch_build_dist(const uint8_t *texels, int nx, int ny, int nz, int bias, int gx, int gy, int gz,
                             uint8_t *dist, uint8_t *scratch, uint8_t *majorant, hipStream_t stream);
hipError_t launch_majorant_cells(const uint8_t *texels, int nx, int ny, int nz, int bias, int cell, const int origin[3], int gx, int gy, int gz,
                                 uint8_t *out, uint8_t *out_codes, hipStream_t stream);
hipError_t launch_brick_meta(const uint8_t *dist, const uint8_t *majorant, int nx, int ny, int nz, int bias, int gx,
                             int gy, int gz, uint8_t *bricks, hipStream_t stream);
hipError_t launch_build_mbricks(const uint8_t *texels, int nx, int ny, int nz, int bias_x, int bias, int gx, int gy,
                                int gz, uint8_t *tmp_a, uint8_t *tmp_b, uint8_t *bricks, hipStream_t stream);
#ifdef CT_EXPERIMENTS
hipError_t launch_mbrick_chunk_class(const uint8_t *bricks, int64_t total_bytes, int64_t chunk_bytes, uint4 *out, hipStream_t stream);
hipError_t launch_mbrick_chunk_quantize(uint8_t *chunk, int64_t chunk_bytes, hipStream_t stream);
#endif
hipError_t launch_mbrick_extent(const uint8_t *bricks, int gx, int gy, int gz, uint32_t *row_x0, uint32_t *row_x1, hipStream_t stream);
hipError_t launch_mbrick_compact(const uint8_t *dense, int gx, int gy, int gz, const uint2 *rows, uint8_t *compact, hipStream_t stream);
hipError_t launch_coarse_clearance(const uint8_t *dist, int nx, int ny, int nz, int bias, int cshift, int cgx, int cgy, int cgz,
                                   uint8_t *out, hipStream_t stream);
hipError_t launch_render_delta(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream);
hipError_t launch_inscatter(const DevScene &sc, uint8_t *out, uint32_t zero_faces, hipStream_t stream);
hipError_t launch_primary_rays(const DevScene &sc, float4 *primary, hipStream_t stream);
hipError_t launch_cost_reduce(const uint2 *cost_plane, uint32_t frame_stride, uint32_t S, uint32_t n_groups_in_chunk,
                              const uint32_t *group_order, uint32_t rank_base, uint32_t *cost_sum, uint32_t *cost_deepest,
                              hipStream_t stream);
hipError_t launch_hit_flags(const float4 *primary, uint8_t *flags, uint32_t pixels, hipStream_t stream);
hipError_t launch_primary_advance(const DevScene &sc, const float4 *primary, float4 *advance, hipStream_t stream);
hipError_t launch_primary_advance_delta(const DevScene &sc, const float4 *primary, float4 *advance, hipStream_t stream);
// Zeroes up to six short word arrays in ONE dispatch (the counters a launch of the estimator starts from: five memsets
// of a few words each were five dispatches of 5 us in front of every 10-subframe launch).
struct ZeroList {
    uint32_t *ptr[6];
    uint32_t words[6];
    int n = 0;
    void add(uint32_t *p, uint32_t w) { ptr[n] = p; words[n] = w; n += 1; }
};
hipError_t launch_zero_words(const ZeroList &z, hipStream_t stream);
hipError_t launch_fill_frame(float4 *frame, uint32_t width, uint32_t height, uint32_t shard_index,
                             uint32_t shard_count, hipStream_t stream);
hipError_t launch_render_persistent(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream);
hipError_t launch_render_simple(const DevScene &sc, const BatchArgs &ba, uint32_t shard_index,
                                uint32_t shard_count, hipStream_t stream);
hipError_t launch_accumulate_batch(const float4 *frames, float4 *mean, float4 *m2, uint32_t first_subframe,
                                   uint32_t S, uint32_t width, uint32_t height, uint32_t shard_index,
                                   uint32_t shard_count, unsigned long long *bad_samples, const uint32_t *frozen,
                                   hipStream_t stream);
hipError_t launch_accumulate_list(const float4 *frames, uint32_t frame_stride, const uint32_t *pixels,
                                  uint32_t n_entries, const uint32_t *group_order, uint32_t rank_base, bool with_misses,
                                  const float4 *primary, float4 *mean, float4 *m2,
                                  uint32_t first_subframe, uint32_t S, uint32_t width, uint32_t height,
                                  uint32_t shard_index, uint32_t shard_count, unsigned long long *bad_samples,
                                  const uint32_t *frozen, hipStream_t stream);
hipError_t launch_reinhard(const float4 *mean, uint32_t width, uint32_t height, float exposure,
                           float *column_sums, float *avg, uchar4 *screen, uint32_t *generation, int device, hipStream_t stream);
hipError_t launch_converged(const float4 *mean, const float4 *m2, uint32_t subframe_id, uint64_t pixels,
                            unsigned long long *unconverged, hipStream_t stream);
hipError_t launch_converged_freeze(const float4 *mean, const float4 *m2, uint32_t subframe_id, uint64_t pixels,
                                   uint32_t limit, uint32_t *state, hipStream_t stream);
hipError_t launch_cdf_selftest(const float *cdf, const uint16_t *guide, uint32_t first_u24, uint32_t count,
                               uint32_t *k_out, hipStream_t stream);
hipError_t launch_math_selftest(int which, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out, hipStream_t stream);
hipError_t launch_fetch_probe(const uint8_t *buf, uint32_t log2_lines, uint32_t second_offset,
                              unsigned long long *sum, hipStream_t stream);
hipError_t launch_fetch_probe_ws(const uint8_t *buf, uint32_t log2_threads, uint32_t ws_lines, uint32_t salt, unsigned long long *sum,
                                 hipStream_t stream);
hipError_t launch_point_rays(const DevScene &sc, const void *tasks, uint32_t n, uint32_t n_pad, float4 *primary,
                             uint32_t *pixels, hipStream_t stream);
hipError_t launch_point_accumulate(const float4 *frames, uint32_t stride, void *tasks, uint32_t n, uint32_t launches,
                                   hipStream_t stream);
hipError_t launch_scatter_samples(const DevScene &sc, uint32_t count, uint32_t batch_seed, float *positions,
                                  float *directions, hipStream_t stream);
// Density pyramid (Resources::generateMipmaps) and the descriptor gather.
constexpr int kMaxMipLevels = 16;
struct MipPyramid {
    const uint8_t *base;               // all levels, level l at base + offset[l], [Z][Y][X]
    uint32_t levels;
    uint32_t offset[kMaxMipLevels];
    int32_t nx[kMaxMipLevels], ny[kMaxMipLevels], nz[kMaxMipLevels];
};
hipError_t launch_mip_level(const uint8_t *prev, int px, int py, int pz, uint8_t *cur, int cx, int cy, int cz,
                            hipStream_t stream);
hipError_t launch_descriptors(const DevScene &sc, const MipPyramid &mp, const float *positions, const float *directions,
                              uint32_t count, float level0, float voxel_m, float cloud_size_m, uint8_t *out,
                              hipStream_t stream);
LaunchShape persistent_shape(int device, bool delta, int blocks_per_cu = 0);   // blocks_per_cu: 0 = as many as fit
#ifdef CT_EXPERIMENTS
// The estimators with a block-wide exchange of paths between waves (ct_exchange.hpp): one 1024-thread block per CU.
LaunchShape exchange_shape(int device);
hipError_t launch_render_delta_x(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream);
LaunchShape wave_exchange_shape(int device);   // the exchange within a wave (render_delta_w_kernel): pool_slots = slots per wave
hipError_t launch_render_delta_w(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream);
#endif

} // namespace ct
