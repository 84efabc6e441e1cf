/*
 * cloudtrace.h -- C ABI of libcloudtrace.so, the MI355X (gfx950) replacement for
 * the Monte-Carlo cloud radiance estimator of marsermd/DeepestScatter.
 *
 * The reference hides its estimator behind the `ARenderer` plug-in interface
 *     class ARenderer { getCamera(); init(); render(optix::Buffer frameResultBuffer); }
 *     (src/Scene/Cameras/ARenderer.h:6-16, implementation PathTracingRenderer.cpp:14-31)
 * whose real "signature" is the set of OptiX context variables the device
 * programs read.  That interface is OptiX-typed and cannot
 * be kept literally; this header keeps its three verbs and makes the implicit
 * inputs explicit.  Every entry point cites the reference code it replaces.
 * "src/" below = DeepestScatter_DataGen/DeepestScatter_DataGen/src/.
 *
 * Conventions
 *   - plain C: pointers, sizes, PODs.  No C++/torch/HIP types in signatures.
 *   - every function returns 0 (CT_OK) or a negative CtStatus; nothing throws,
 *     nothing calls exit().  ct_last_error() gives the message of the last
 *     failure on that handle (or of the last failed ct_create when h == NULL).
 *   - a handle is NOT thread-safe (the reference is single-threaded:
 *     GuiExecutionLoop.cpp:53-60); distinct handles are independent.
 *   - image layout: row-major W x H, 4 floats (or 4 bytes) per pixel, row 0 is the
 *     BOTTOM of the picture (cameraCommon.cuh:22-25).
 *   - pointers named *_dev are device pointers valid on the handle's GPU,
 *     pointers named *_host are host pointers; `ct_download` copies device->host.
 *   - all device work is issued on one HIP stream owned by the handle (or the one
 *     given with ct_set_stream) and the *_async-free entry points return after
 *     that stream is idle.
 */
#ifndef CLOUDTRACE_H
#define CLOUDTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CT_ABI_VERSION 1

#if defined(__GNUC__)
#define CT_API __attribute__((visibility("default")))
#else
#define CT_API
#endif

typedef enum CtStatus {
    CT_OK = 0,
    CT_E_INVAL = -1,   /* bad argument (std::invalid_argument in CloudMaterial.cpp:62) */
    CT_E_HIP = -2,     /* a HIP call or kernel failed (optix::Exception path, main.cpp:65-69) */
    CT_E_NOMEM = -3,   /* host or device allocation failed */
    CT_E_STATE = -4,   /* call order violated (e.g. render before ct_set_camera) */
    CT_E_NODEVICE = -5, /* no usable gfx950 device / extension missing */
    CT_E_RCCL = -6     /* librccl missing or a collective failed (ct_group_*) */
} CtStatus;

/* Cloud::Rendering::Mode, src/Scene/SceneDescription.h:39-44 -> program chosen in
 * CloudMaterial.cpp:51-64 */
typedef enum CtMode {
    CT_MODE_SUN_AND_SKY_ALL_SCATTER = 0, /* totalRadiance,              cloudRadianceMaterials.cu:9-66  */
    CT_MODE_SUN_MULTIPLE_SCATTER = 1,    /* multipleScatterSunRadiance, cloudRadianceMaterials.cu:72-115 */
    CT_MODE_SUN_SINGLE_SCATTER = 2       /* singleScatterSunRadiance,   cloudRadianceMaterials.cu:120-148 */
} CtMode;

/* Free-flight sampler. MARCH is the reference's estimator (cloud.cuh:77-114);
 * DELTA is Woodcock tracking over a grid of majorant cells (BASELINE.json north_star). */
typedef enum CtEstimator {
    CT_EST_MARCH = 0,
    CT_EST_DELTA = 1
} CtEstimator;

/* Which buffer ct_download / ct_device_ptr refers to. */
typedef enum CtBuffer {
    CT_BUF_MEAN = 0,       /* progressiveBuffer  float4 W*H   (Camera.cpp:46)        */
    CT_BUF_M2 = 1,         /* varianceBuffer     float4 W*H   (Camera.cpp:47)        */
    CT_BUF_FRAME = 2,      /* frameResultBuffer  float4 W*H   (Camera.cpp:45)        */
    CT_BUF_SCREEN = 3,     /* screenBuffer       uchar4 W*H   (Camera.cpp:48)        */
    CT_BUF_INSCATTER = 4,  /* inScatterBuffer    uint8  X*Y*Z (VDBCloud.cpp:70)      */
    CT_BUF_DENSITY = 5     /* density            uint8  X*Y*Z (Resources.cpp:127-141)*/
} CtBuffer;

/*
 * Everything the reference's device programs read from OptiX variable scopes,
 * as one POD (SURVEY.md section 8b "implicit inputs").
 */
typedef struct CtScene {
    uint32_t abi_version;     /* must be CT_ABI_VERSION */

    /* --- density texture: Resources::loadVolumeBuffer, src/Util/Resources.cpp:68-155 --- */
    uint32_t dims[3];         /* X,Y,Z texels, INCLUDING the 1-texel zero border (:97-101) */
    const uint8_t *density_host; /* X*Y*Z bytes, x fastest, z slowest (:127-141); copied at create */

    /* --- Cloud::Model, src/Scene/SceneDescription.h:60-83 --- */
    float cloud_size_m;       /* 7000 in main.cpp:63 */
    float mean_free_path_m;   /* 10, SceneDescription.h:80; densityMultiplier = size/mfp (VDBCloud.cpp:109) */

    /* --- Cloud::Rendering, SceneDescription.h:35-57; installers.cpp:86 --- */
    float sample_step;        /* 1/512 */
    int32_t mode;             /* CtMode */
    int32_t estimator;        /* CtEstimator */
    uint32_t max_depth;       /* MAX_DEPTH = 2000, cloudRadianceMaterials.cu:4 */

    /* --- DirectionalLight, SceneDescription.h:13-26; Sun.cpp:13-18; installers.cpp:74-101 --- */
    float light_direction[3]; /* direction the light TRAVELS; normalised twice like the reference */
    float light_color[3];     /* (1,1,1) */
    float light_intensity;    /* 1e6 */

    /* --- Camera::Settings, src/Scene/Cameras/Camera.h:20-28 --- */
    uint32_t width, height;

    /* --- Mie tables: Scene::init binds them (Scene.cpp:38-40); data of Mie.cpp:8-8203 --- */
    const float *mie_host;         /* raw `mie` table,        mie_count floats */
    const float *chopped_mie_host; /* raw `choppedMie` table, mie_count floats */
    uint32_t mie_count;            /* 4096 */

    /* --- placement (new: the reference is single-GPU) --- */
    int32_t device;           /* HIP device ordinal this handle lives on */
    uint32_t shard_index;     /* this handle renders the 8x8-pixel tiles (tx,ty) with        */
    uint32_t shard_count;     /*   ct_tile_owner(tx,ty,shard_count) == shard_index; 1 = all  */

    uint32_t flags;           /* CT_FLAG_* */
} CtScene;

#define CT_FLAG_NONE 0u
#define CT_FLAG_SIMPLE_KERNEL 1u /* one thread per pixel, nested loops (A/B + cross-check only) */
#define CT_FLAG_LIGHT_NORMALIZED 2u /* light_direction already went through the reference's two
                                       normalisations (a host that mirrors installSceneSetup +
                                       DirectionalLight); use it as is */

#define CT_FLAG_SPARSE_BRICKS 4u    /* store the MARCH estimator's density bricks sparsely (only the bricks between the first and
                                       the last non-empty one of every brick row; 9.5x smaller at 1024^3, 20 % slower: for
                                       volumes whose dense bricks would not fit; identical results) */
#define CT_FLAG_VMM_BRICKS 8u       /* EXPERIMENTS BUILD ONLY (libcloudtrace_exp.so; the product's library answers CT_E_INVAL): measured and
                                       rejected in round 4 -- memory mapped in 2-MiB pieces is gathered at 0.55 of the hipMalloc rate.
                                       The same bricks with dense ADDRESSING and sparse BACKING: the dense array's virtual range
                                       is reserved (hipMemAddressReserve) and only the 2-MiB chunks that hold a non-zero texel,
                                       or lie within a few texels of one, get memory of their own; every other chunk is mapped
                                       onto one of a few shared chunks (all-zero texels, a clearance rounded down to 4, 8, 16,
                                       32, 64 or 127 texels).  The kernel and its address arithmetic are the dense ones;
                                       identical results (a smaller clearance only makes the exact free-space skip shorter) */

/* Deterministic work counters of everything rendered since create/ct_reset
 * (the algorithmic-bytes figure is built from these). */
typedef struct CtCounters {
    uint64_t paths;            /* primary rays traced (pixels x subframes in this shard) */
    uint64_t box_hits;         /* primary rays that hit the cloud's box */
    uint64_t density_lookups;  /* trilinear fetches of the density texture (8 B each) */
    uint64_t inscatter_lookups;/* trilinear fetches of the shadow volume   (8 B each) */
    uint64_t scatter_events;   /* accepted collisions = NEE evaluations */
    uint64_t depth_capped;     /* paths stopped by max_depth */
} CtCounters;

/* What the kernels actually ISSUED for the lookups above since create/ct_reset (implementation figures, not
 * the algorithm's: the MARCH estimator replays free-space steps without a fetch, walks the part of a primary
 * flight that is the same for every sample of a pixel once per pose, and reuses a lane's last shadow-volume
 * footprint; every such step is still a density/inscatter lookup of the algorithm).  bench.py builds the
 * issued-bytes roofline figure from these: 8 useful bytes per fetch. */
typedef struct CtFetchCounters {
    uint64_t density_fetches;   /* trilinear footprints of the density actually loaded by the estimator kernel */
    uint64_t inscatter_fetches; /* the same for the shadow volume */
} CtFetchCounters;

typedef struct CtHandle_ *CtHandle;

/* ---- lifetime --------------------------------------------------------------------- */

/* Scene::init for the path-tracing configuration (Scene.cpp:36-46): uploads density,
 * prepares the three Mie textures (Mie.cpp:8206-8297), publishes the Sun/VDBCloud
 * variables (Sun.cpp:13-18, VDBCloud.cpp:88-117), runs the inScatter precompute
 * (VDBCloud.cpp:57-86 -> inScatter.cu:40-66), allocates frame/progressive/variance/
 * screen buffers and clears them (Camera.cpp:45-48,77-86).  Sets the default camera
 * pose of Camera.cpp:37-39. */
CT_API int ct_create(const CtScene *scene, CtHandle *out);

/* Context destruction between tasks, GuiExecutionLoop.cpp:93-97. NULL is a no-op. */
CT_API int ct_destroy(CtHandle h);

/* Message of the last failure (UTF-8, owned by the library, valid until the next call
 * on the same handle).  h == NULL: last failure of ct_create in this thread. */
CT_API const char *ct_last_error(CtHandle h);

/* Use `hip_stream` (a hipStream_t passed as void*) for all later work on this handle
 * instead of the handle's own stream.  NULL restores the handle's stream. */
CT_API int ct_set_stream(CtHandle h, void *hip_stream);

/* ---- ARenderer verbs ----------------------------------------------------------------- */

/* ARenderer::getCamera()["eye"|"U"|"V"|"W"]->setFloat(...)   Camera.cpp:126-133 */
CT_API int ct_set_camera(CtHandle h, const float eye[3], const float U[3], const float V[3], const float W[3]);

/* ARenderer::render(frameResultBuffer) after context["subframeId"]=id  (Camera.cpp:191-195,
 * PathTracingRenderer.cpp:21-31): one path per pixel, result float4(r,g,b,1) per pixel into
 * the handle's frame buffer; if frame_rgba_dev != NULL the frame is also copied there
 * (device pointer, W*H*4 floats, caller-owned).  Pixels of other shards are written as 0
 * with alpha 0.  Seeds are tea<4>(x*4096+y, subframe_id)  (documented deviation from
 * random.cuh:38, which mixes in clock()). */
CT_API int ct_render_subframe(CtHandle h, uint32_t subframe_id, float *frame_rgba_dev);

/* updateFrameResult, progressive.cu:17-27, launched by Camera.cpp:197-199: Welford update
 * of mean/M2 with n = subframe_id from the handle's frame buffer (or from frame_rgba_dev
 * when not NULL). */
CT_API int ct_accumulate(CtHandle h, uint32_t subframe_id, const float *frame_rgba_dev);

/* The hot path of Camera::render's loop (Camera.cpp:189-200) fused: for id = first ..
 * first+count-1 { render(id); updateFrameResult(id) } with identical results, in one launch,
 * without materialising the frame buffer.  Requires first == (subframes accumulated so far)+1. */
CT_API int ct_render_accumulate(CtHandle h, uint32_t first_subframe_id, uint32_t count);

/* The same, pipelined: the call enqueues the batch on the handle's stream and returns.  With the MARCH
 * estimator a launch does not run its surviving paths to their end once its job list is empty (that tail of
 * waves carrying a few long paths each is 19 ms of a 107 ms launch at 256 subframes): it suspends them and the
 * next batch's launch resumes them first, so the accumulate kernel of batch k runs behind the launch of batch
 * k+1, and ct_synchronize() (or any entry point that waits) finishes the last batch with a launch that only
 * resumes.  Paths, arithmetic and the subframe order of the accumulation are unchanged: results are identical
 * to ct_render_accumulate's.  Every other entry point waits for the batches in flight first, except
 * ct_copy_to_device_async and ct_subframes; a copy enqueued after batch k sees the running mean up to batch
 * k-1.  (The reference is synchronous: Camera::render maps its buffers right after context->launch,
 * Camera.cpp:189-240; SURVEY section 8b asks for an _async variant.) */
CT_API int ct_render_accumulate_async(CtHandle h, uint32_t first_subframe_id, uint32_t count);
CT_API int ct_synchronize(CtHandle h);

/* Render-ahead for the reference's display cadence (Camera::render: 10 subframes, then tonemap and display,
 * Camera.cpp:189-214).  With subframes > 0, a call of ct_render_accumulate_async for FEWER subframes than that is served by
 * an estimator launch of `subframes` subframes -- a launch of 10 subframes is mostly beginning and end: every lane resumes a
 * path and suspends one, the lanes fill and drain; per subframe a launch of 80 keeps the chip busy for 21 % fewer cycles
 *  -- and every call accumulates ITS OWN share of it, in order: the images the calls produce are the
 * reference's images for those subframe counts, bit for bit, each a fixed number of calls later (the running mean follows
 * the calls by `subframes` subframes per launch a path may span; ct_synchronize and every entry point that waits bring it
 * to exactly the subframes asked for).  Samples rendered ahead and not asked for yet stay in the scratch for the next
 * calls; ct_set_camera, ct_reset, ct_set_subframes and ct_accumulate drop them, and CtCounters count them when they are
 * rendered.  0 (the default) turns it off; also CT_RENDER_AHEAD in the environment at ct_create.
 * ct_rendered_subframes: how far the estimator has been launched (>= ct_subframes). */
CT_API int ct_set_render_ahead(CtHandle h, uint32_t subframes);
CT_API int ct_rendered_subframes(CtHandle h, uint32_t *count_out);

/* Camera::reset, Camera.cpp:77-86 (clearScreen, progressive.cu:29-34): zero frame, mean, M2,
 * subframe count.  Counters are zeroed too. */
CT_API int ct_reset(CtHandle h);

/* Reinhard tonemap of the running mean: firstPass/secondPass/applyReinhard,
 * reinhard.cu:26-84, launched by Camera.cpp:202-210; default exposure 0.4 (Camera.h:90).
 * Result goes to the handle's screen buffer and, if rgba_host != NULL, is copied to the host
 * (W*H*4 bytes). avg_luminance_out (optional) receives reinhard.cu:53 averageLuminance. */
CT_API int ct_tonemap(CtHandle h, float exposure, uint8_t *rgba_host, float *avg_luminance_out);

/* The same enqueued behind the batches of ct_render_accumulate_async, without waiting: the display update of
 * Camera::render (Camera.cpp:202-210, every 10 subframes) at the reference's cadence without a host round trip per
 * update.  It tonemaps the running mean of the batches accumulated so far -- with path continuation a batch is
 * accumulated a few launches after it was enqueued (ct_synchronize brings everything up to date) -- into the handle's
 * CT_BUF_SCREEN; read it with ct_download / ct_copy_to_device after ct_synchronize, or display it from the device. */
CT_API int ct_tonemap_async(CtHandle h, float exposure);

/* Camera::isConverged, Camera.cpp:232-268, evaluated on the device.  *converged_out = 1 when
 * fewer than 500 pixels are outside the 95 % interval; *unconverged_pixels_out optional. */
CT_API int ct_is_converged(CtHandle h, int32_t *converged_out, uint64_t *unconverged_pixels_out);

/* Camera::render's `if (!isConverged())` (Camera.cpp:179) without a host round trip per update.  With cadence > 0 the
 * library enqueues the test of Camera::isConverged behind the accumulate kernel of every cadence-th subframe from
 * min_subframes on (the reference: 10 and 100, Camera.cpp:189,234) -- the accumulate kernels are cut at those counts, whatever
 * the sizes of the calls, render-ahead included -- and takes the reference's decision on the device: at the first such count
 * with fewer than 500 pixels outside the interval the running mean and M2 are FROZEN; every later accumulate kernel leaves
 * them alone.  So a host that enqueues update after update (ct_render_accumulate_async + ct_tonemap_async) and looks at
 * ct_converged_at now and then -- it never waits -- ends with exactly the image, and the subframe count, at which the
 * reference's loop stops; what it enqueued beyond that point is rendered and dropped.  ct_converged_at: *subframes_out = the
 * count the image was frozen at (0 = still running), *tested_at_out / *unconverged_pixels_out = the last test that has
 * finished (any of them may be NULL); up to date after ct_synchronize.  ct_reset and ct_set_stop_when_converged clear the
 * state; ct_is_converged after a freeze tests the frozen image with its own count.  A batch rendered in several chunks of
 * pixel groups (it did not fit the scratch) is tested at its end only.  Whole frames only: CT_E_INVAL on a shard of a
 * multi-GPU job (test the merged frame with ct_is_converged_buffers).  cadence 0 (the default) turns it off. */
CT_API int ct_set_stop_when_converged(CtHandle h, uint32_t cadence, uint32_t min_subframes);
CT_API int ct_converged_at(CtHandle h, uint32_t *subframes_out, uint32_t *tested_at_out, uint64_t *unconverged_pixels_out);

/* The same two on caller-owned device buffers of W*H float4 each: the frame a multi-GPU reduce merged on rank 0
 * (every shard's handle holds its own tiles and zeros elsewhere; the SUM of the shards' CT_BUF_MEAN / CT_BUF_M2 is
 * the whole image).  reinhard.cu:44-55 sums the luminance of the WHOLE frame in a fixed order
 * and Camera.cpp:232-268 counts over the WHOLE frame, so both run on the merged buffers, on one rank, with the
 * single-GPU arithmetic -- not as per-shard partial sums, which would change the order of the float additions.
 * `subframes` = samples per pixel the buffers hold.  The screen bytes go to the handle's CT_BUF_SCREEN as usual. */
CT_API int ct_tonemap_buffer(CtHandle h, const float *mean_rgba_dev, float exposure, uint8_t *rgba_host,
                             float *avg_luminance_out);
CT_API int ct_is_converged_buffers(CtHandle h, const float *mean_rgba_dev, const float *m2_rgba_dev, uint32_t subframes,
                                   int32_t *converged_out, uint64_t *unconverged_pixels_out);

/* ---- radiance samples: the estimator over (point, direction) tasks -------------------------- */

/* Gpu::PointRadianceTask, src/CUDA/PointRadianceTask.h:12-78 (40 bytes, same field order). */
typedef struct CtPointRadianceTask {
    int32_t id;
    uint32_t experimentCount;
    float radiance;          /* running mean of prd.result.x */
    float runningVariance;   /* running M2 */
    float position[3];       /* ray origin, world coordinates (box centred at 0) */
    float direction[3];      /* ray direction (need not be normalised) */
} CtPointRadianceTask;

/* `launches` consecutive launches of estimateEmission (src/CUDA/pointEmissionCamera.cu:20-40) over
 * tasks[0..count): for frame = first_frame_id .. first_frame_id+launches-1, thread i traces one path
 * from (position, direction) with the handle's render mode (the reference uses SunMultipleScatter,
 * Tasks.cpp:135), seed tea<4>(i*4096, frame), and folds prd.result.x into task i with
 * PointRadianceTask::addExperimentResult (:40-51), in frame order.  tasks_host is updated in place.
 * This is the device part of RadianceCollector::update (RadianceCollector.cpp:88-96); replication,
 * merging, convergence and rescheduling stay with the caller. */
CT_API int ct_point_radiance_launch(CtHandle h, CtPointRadianceTask *tasks_host, uint32_t count,
                                    uint32_t first_frame_id, uint32_t launches);

/* generatePoints (src/CUDA/pointGeneratorCamera.cu:20-42) + firstScatterPosition
 * (cloudFirstScatterMaterial.cu:8-29), launched over `count` threads by ScatterSampleCollector::collect
 * (ScatterSampleCollector.cpp:36-62): thread i draws a direction uniformly on the sphere and a point on
 * the reference's disc of radius sqrt(3)/2, shoots a ray from 2 units away and keeps the first scatter
 * position (world coordinates, box centred at 0) and the view direction; it retries until a ray
 * scatters inside the cloud (at most 4096 times, then the sample stays NaN like after `clear`).
 * Seeds: tea<4>(i, batch_seed) for the generator and tea<4>(i*4096, batch_seed + attempt) for the
 * flight (the reference mixes clock() into both).  Outputs are host arrays of 3*count floats. */
CT_API int ct_generate_scatter_samples(CtHandle h, uint32_t count, uint32_t batch_seed,
                                       float *positions_host_out, float *directions_host_out);

/* setupHierarchicalDescriptor<DisneyDescriptor, uint8_t> (src/CUDA/DisneyDescriptor.cuh:71-112) as launched
 * by DisneyDescriptorCollector::collect (src/Scene/DisneyDescriptorCollector.cpp:57-63; program `collect`,
 * src/CUDA/disneyDescriptorCollector.cu:21-28): for every (position, view direction) sample, 10 layers of
 * 9 x 5 x 5 trilinear + mip-linear samples of the density pyramid (Resources::generateMipmaps,
 * Resources.cpp:169-209) in the frame eZ = -light, eX = normalize(eZ x view), eY = eX x eZ; layer l spans
 * [-1,1]^2 x [-1,3] * 2^l free paths at LOD level0 + l, faded to zero outside the box, stored as uint8
 * (f * 255, truncating).  positions are the ScatterSample records' `point` (world coordinates, box centred
 * at 0); both inputs are host arrays of 3*count floats; descriptors_host_out receives
 * count * CT_DESCRIPTOR_BYTES bytes, layer-major, then z, y, x -- the `grid` field of
 * Persistance::DisneyDescriptor (DeepestScatter_Train/Protocols/DisneyDescriptor.proto:7-10). */
#define CT_DESCRIPTOR_LAYERS 10
#define CT_DESCRIPTOR_LAYER_SIZE 225
#define CT_DESCRIPTOR_BYTES (CT_DESCRIPTOR_LAYERS * CT_DESCRIPTOR_LAYER_SIZE)
CT_API int ct_collect_descriptors(CtHandle h, const float *positions_host, const float *directions_host,
                                  uint32_t count, uint8_t *descriptors_host_out);

/* ---- data access -------------------------------------------------------------------- */

/* BufferBind<T>(buffer) map/copy, src/Util/BufferBind.h:11-74 (e.g. Camera.cpp:161,239-240).
 * Copies the whole buffer to dst_host; dst_bytes must equal ct_buffer_bytes(). */
CT_API int ct_download(CtHandle h, int32_t which /*CtBuffer*/, void *dst_host, size_t dst_bytes);
CT_API int ct_buffer_bytes(CtHandle h, int32_t which /*CtBuffer*/, size_t *bytes_out);

/* The other direction, for CT_BUF_MEAN and CT_BUF_M2 only: with ct_download and ct_set_subframes this is checkpoint / resume of
 * a progressive render -- (mean, M2, subframe count) IS its whole state, because a sample's seed is (pixel, subframe id): a
 * handle that is given the three continues exactly where the saved one was (the reference's own EXR dumps
 * every 40 subframes, Camera.cpp:211-214, cannot be resumed from -- no variance, no count).  src_bytes must equal
 * ct_buffer_bytes(); other buffers are CT_E_INVAL.  With ct_set_stop_when_converged the count that belongs to the buffers
 * is the FROZEN count (ct_converged_at) once the image has frozen, not ct_subframes, which goes on counting what the host
 * submits; and both ct_upload and ct_set_subframes release a frozen handle (the image they describe is a new one). */
CT_API int ct_upload(CtHandle h, int32_t which /*CtBuffer*/, const void *src_host, size_t src_bytes);

/* Raw device pointer of CT_BUF_MEAN / CT_BUF_M2 / CT_BUF_FRAME / CT_BUF_SCREEN, so a caller can
 * hand the accumulated radiance buffer to a collective (RCCL) without a copy. */
CT_API int ct_device_ptr(CtHandle h, int32_t which /*CtBuffer*/, void **ptr_out);

/* Device-to-device copy of a whole buffer into caller-owned device memory (e.g. a tensor that a
 * collective will reduce), on the handle's stream; returns after the copy completed. */
CT_API int ct_copy_to_device(CtHandle h, int32_t which /*CtBuffer*/, void *dst_dev, size_t dst_bytes);
/* Enqueued on the handle's stream, not waited for (see ct_render_accumulate_async). */
CT_API int ct_copy_to_device_async(CtHandle h, int32_t which, void *dst_dev, size_t dst_bytes);

/* Number of subframes submitted so far (Camera::subframeId, Camera.h:76); equals the samples in the buffers unless the image
 * has frozen (ct_set_stop_when_converged: ct_converged_at then names the count the buffers hold). */
CT_API int ct_subframes(CtHandle h, uint32_t *count_out);

/* After an external reduction wrote merged data into CT_BUF_MEAN/CT_BUF_M2 (multi-GPU frame
 * reduce), tell the handle how many subframes those buffers now represent. */
CT_API int ct_set_subframes(CtHandle h, uint32_t count);

CT_API int ct_counters(CtHandle h, CtCounters *out);

CT_API int ct_fetch_counters(CtHandle h, CtFetchCounters *out);

/* Milliseconds the GPU spent in the estimator kernel and in the accumulate kernel since
 * create/reset, measured with HIP events on the handle's stream, and the number of estimator
 * launches.  Any pointer may be NULL. */
CT_API int ct_kernel_time(CtHandle h, double *render_ms_out, double *accumulate_ms_out, uint64_t *launches_out);

/* Scheduler statistics of the estimator kernel since create/reset (implementation diagnostics, not
 * part of the algorithm): out[0..15] = regen phases, regen lanes, march phases, march lanes,
 * scatter phases, scatter lanes, fetched march steps, fetched steps whose 8 texels were all 0,
 * skipped (replayed) march steps, skip-loop trips (wave level), rest reserved. */
CT_API int ct_debug_stats(CtHandle h, uint64_t out[64]);
/* All `count` <= 72 diagnostic words (64..67: path conservation, 68..69: brick-line reuse of the march fetches). */
CT_API int ct_debug_stats_ex(CtHandle h, uint64_t *out, uint32_t count);
/* Diagnostics (a library built with -DCT_DIAG_TIMELINE, CT_TIMELINE=1 in the environment at ct_create, MARCH estimator): [start, end, the time it learnt that no job is left, what it
 * held then: live lanes | lanes too old to hand on << 8 | job unfinished << 16] of every wave of the last ENQUEUED estimator
 * launch, times on the device's 100 MHz wall clock, 4 * waves words (the launch that only resumes does not write it). */
CT_API int ct_debug_timeline(CtHandle h, uint64_t *out, uint32_t waves);

/* Paths that ct_render_accumulate_async launches have handed to their successors so far (diagnostic). */
CT_API int ct_debug_suspended(CtHandle h, uint64_t *paths_out);

/* Path conservation and sample integrity (diagnostic).  out[0] = 1 when the handle was created with
 * CT_DEBUG_INVARIANTS=1 in the environment: then the diagnostics build of the estimator runs, the per-sample scratch
 * is filled with NaNs before every launch, and every entry point that waits for the batches in flight checks
 *     samples dealt + paths resumed == results written + paths suspended,   paths resumed == paths suspended,
 *     samples dealt == what the host handed out,   no sample without alpha == 1 reached an accumulate kernel
 * and fails with CT_E_STATE otherwise.  out[1] = checks made, out[2] = violations, out[3] = samples without
 * alpha 1 seen by the accumulate kernels (always counted), out[4..7] = dealt, resumed, written, suspended
 * (0 unless out[0]). */
CT_API int ct_debug_invariants(CtHandle h, uint64_t out[8]);

/* Device memory of the volume representations (bytes): out[0] raw density texture, out[1] density apron bricks,
 * out[2] shadow-volume apron bricks, out[3] march bricks if stored densely, out[4] march bricks as stored,
 * out[5] = 1 when they are stored sparsely (row extents; CT_FLAG_SPARSE_BRICKS or CT_SPARSE=1), out[6] bytes of the
 * row-extent table, out[7] bytes of the coarse clearance grid.  out[5] = 2: dense addressing with sparse backing
 * (CT_FLAG_VMM_BRICKS or CT_SPARSE=2); out[3] is then the size of the address range and out[4] the memory behind it. */
CT_API int ct_debug_memory(CtHandle h, uint64_t out[8]);

/* The DELTA estimator's majorant grid and kernel variant (zeros for a MARCH handle): out[0] cell edge in texels, out[1..3] stored
 * cells per axis, out[4..6] the stored box's first cell in the virtual grid, out[7] = NEE variant (0, 1, 2) | 0x100 when every
 * non-zero texel lies two texels or more inside the volume's faces and the stored cells a texel or more inside the brick grid
 * (then a real collision is inside the box and a tentative one inside the grid by construction, and the kernel instantiated
 * without that test and that clamp runs; CT_DELTA_INTERIOR=0 keeps both). */
CT_API int ct_debug_delta_grid(CtHandle h, uint32_t out[8]);

/* PMC calibration probe (no handle): allocates 2^log2_lines 128-byte lines on `device`, and has one
 * thread per line issue the estimator's access pattern (two unaligned 8-byte loads at byte 13 and
 * byte 38 of a pseudo-randomly chosen, never repeated line).  Under rocprofv3 --pmc FETCH_SIZE this
 * tells how many bytes the counter reports per touched line.  Odd-numbered repeats move the second
 * load to byte 85 so that both 64-byte halves of the line are touched.  Returns a checksum. */
CT_API int ct_debug_fetch_probe(int32_t device, uint32_t log2_lines, uint32_t repeats, uint64_t *sum_out);

/* The same access shape over a WORKING SET of `ws_lines` 128-byte lines: each of 2^log2_threads lanes reads one
 * pseudo-random line of the set, so a set smaller than a cache level is re-read from that level -- the random-line
 * fill ceiling of L2 (4 MiB per XCD), of the Infinity Cache (256 MiB) and of HBM, by size.  `repeats` launches;
 * time two calls with different repeat counts and divide the difference. */
CT_API int ct_debug_fetch_probe_ws(int32_t device, uint32_t log2_threads, uint64_t ws_lines, uint32_t repeats, uint64_t *sum_out);

/* The estimator's working set (diagnostics kernels: CT_STATS=1 or CT_DEBUG_INVARIANTS=1 at ct_create).
 * ct_debug_track_lines(h, 1) allocates and clears one bit per 128-byte line of the density array the estimator reads
 * (MARCH: march bricks; DELTA: apron or twin bricks) and of the shadow volume's apron bricks; every fetch of a later
 * launch sets its line's bit; (h, 0) frees them.  ct_debug_touched_lines waits for the batches in flight and returns
 * out[0], out[1] = distinct lines touched (density, shadow), out[2], out[3] = lines in the two arrays; clear != 0
 * resets the bits. */
CT_API int ct_debug_track_lines(CtHandle h, int32_t enable);
CT_API int ct_debug_touched_lines(CtHandle h, uint64_t out[4], int32_t clear);

/* Self-test hook: the kernels' short correctly-rounded reciprocal (which = 0) and square root (which = 1) against the IEEE
 * operations for EVERY float of their range, 2^-60 <= |x| < 2^61 (reciprocal: both signs), on the device:
 * out[0] = floats tested, out[1] = mismatches (must be 0), out[2] = smallest mismatching bit pattern or 0xffffffff. */
CT_API int ct_debug_math_selftest(CtHandle h, int32_t which, uint64_t out[3]);

/* Self-test hook: k(val) of the CDF inversion (cloud.cuh:162-180) for `count` consecutive 24-bit
 * random integers starting at first_u24, evaluated by the device code; cos(theta) = (2k+1)/65536-1. */
CT_API int ct_debug_cdf_inversion(CtHandle h, uint32_t first_u24, uint32_t count, uint32_t *k_host_out);

/* ---- multi-GPU below the C ABI: one process, the GPUs of one node (SURVEY section 8b "multi-GPU is internal",
 * section 8e) ------------------------------------------------------------------------------------------------------
 * The reference is single-GPU; BASELINE.json shards the frame by pixel tile over the GPUs of a
 * node and reduces the accumulated radiance buffer with RCCL.  A CtGroup is that for a C or C++ host: one CtHandle
 * per entry of `devices` (handle i renders the tiles of shard i of `count`, see ct_tile_owner) and an RCCL
 * communicator over them (ncclCommInitAll; librccl is loaded on first use, CT_E_RCCL when it cannot be).
 * ct_group_render_accumulate enqueues the batch on every device and waits for all of them; ct_group_merge SUM-reduces
 * every shard's [mean | M2] (2 x W*H float4; foreign pixels are exactly 0, so the sum is the merged frame, bit for bit
 * the single-GPU image) onto devices[0] with one ncclReduce per device inside ncclGroupStart/End; ct_group_download,
 * ct_group_tonemap and ct_group_is_converged work on the merged frame (and merge first when needed).
 * scene->device, shard_index and shard_count are ignored.  A device may appear more than once in `devices` -- a
 * rehearsal of an N-GPU job on fewer GPUs; RCCL refuses two ranks on one device, so such a group merges with peer
 * copies and an add kernel instead of a collective.  The Python host does the same job with one process per GPU over
 * torch.distributed (deepestscatter_amd/distributed.py); both produce identical frames. */
typedef struct CtGroup_ *CtGroup;
CT_API int ct_group_create(const CtScene *scene, const int32_t *devices, uint32_t count, CtGroup *out);
CT_API int ct_group_destroy(CtGroup g);
CT_API const char *ct_group_last_error(CtGroup g);      /* g == NULL: last failure of ct_group_create in this thread */
CT_API int ct_group_size(CtGroup g, uint32_t *count_out);
CT_API int ct_group_handle(CtGroup g, uint32_t index, CtHandle *out);   /* shard `index` (owned by the group) */
CT_API int ct_group_set_camera(CtGroup g, const float eye[3], const float U[3], const float V[3], const float W[3]);
CT_API int ct_group_render_accumulate(CtGroup g, uint32_t first_subframe_id, uint32_t count);
CT_API int ct_group_reset(CtGroup g);
CT_API int ct_group_merge(CtGroup g);
CT_API int ct_group_download(CtGroup g, int32_t which /*CT_BUF_MEAN | CT_BUF_M2*/, void *dst_host, size_t dst_bytes);
CT_API int ct_group_tonemap(CtGroup g, float exposure, uint8_t *rgba_host, float *avg_luminance_out);
CT_API int ct_group_is_converged(CtGroup g, int32_t *converged_out, uint64_t *unconverged_pixels_out);
CT_API int ct_group_counters(CtGroup g, CtCounters *out);              /* sums over the shards */

/* ---- host-side helpers of the same path (pure CPU, no handle, no GPU) ------------------- */

/* sutil::calculateCameraVariables(..., fov_is_vertical=false), src/Util/sutil.cpp:501-524, as
 * called by Camera::updatePosition (Camera.cpp:100-134): hfov in degrees. */
CT_API int ct_calculate_camera_variables(const float eye[3], const float lookat[3], const float up[3],
                                  float hfov_deg, float aspect_ratio,
                                  float U_out[3], float V_out[3], float W_out[3]);

/* The quantiser of Resources::loadVolumeBuffer, Resources.cpp:92-141: payload float grid
 * (nx*ny*nz, x fastest) -> uint8 texture of (nx+2)(ny+2)(nz+2) with a 1-texel zero border,
 * value = (uint8)(v / max * 255) computed in double, truncating. */
CT_API int ct_quantize_volume(const float *grid_host, const uint32_t payload_dims[3], uint8_t *texture_host_out);

/* Resources::loadVolumeBuffer for a .vdb file, Resources.cpp:82-143, without OpenVDB (deepestscatter_amd/host/
 * VdbReader.h reads the file format): the FIRST grid of the file, which must be a FloatGrid; maxDensity = the largest
 * ACTIVE value; the active bounding box expanded by one voxel on every side; dense fill by getValue (inactive voxels
 * and tiles included), x fastest; uint8 = (uint8)(value / maxDensity * 255).  dims_out = the texture's size in
 * texels (what CtScene::dims wants); texture_host_out may be NULL to query dims_out / *bytes_out.  On failure the
 * message (unsupported compression codec, not a FloatGrid, ...) goes to error_out. */
CT_API int ct_load_vdb(const char *path, uint32_t dims_out[3], uint8_t *texture_host_out, size_t capacity, size_t *bytes_out,
                       char *error_out, size_t error_capacity);

/* Resources::generateMipmaps, Resources.cpp:169-209: level count = floor(log2(maxdim))+1; each
 * level is the 2x2x2 integer mean (uint16 sum / 8, zero outside).  level_offsets_out[l] = byte
 * offset of level l in `pyramid_host_out`; returns CT_E_INVAL if capacity is too small.  Pass
 * pyramid_host_out == NULL to query *levels_out and *bytes_out. */
CT_API int ct_generate_mipmaps(const uint8_t *level0_host, const uint32_t dims[3],
                        uint8_t *pyramid_host_out, size_t capacity,
                        uint32_t *levels_out, size_t *bytes_out, size_t *level_offsets_out /*[32]*/);

/* The pixel-tile -> shard map used for multi-GPU sharding: the frame is cut
 * into 8x8-pixel tiles; tile (tx,ty) belongs to shard (tx + 3*ty) mod shard_count, a skewed
 * interleave that gives every GPU the same mix of cloud and background. */
CT_API uint32_t ct_tile_owner(uint32_t tile_x, uint32_t tile_y, uint32_t shard_count);

/* Synthetic cloud of SURVEY section 8(d): 5-octave gradient-noise fBm x ellipsoidal falloff,
 * thresholded, quantised by ct_quantize_volume's rule into an n^3 texture (payload n-2). */
CT_API int ct_make_procedural_cloud(uint32_t n, uint32_t seed, uint8_t *texture_host_out);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDTRACE_H */
