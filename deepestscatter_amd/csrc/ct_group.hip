// ct_group.hip -- multi-GPU below the C ABI: one process drives the GPUs of one node.
//
// A CtGroup owns one CtHandle per device -- handle i renders the 8x8-pixel tiles of shard i of N (ct_tile_owner) --
// and an RCCL communicator over those devices (ncclCommInitAll).  ct_group_merge stages every shard's running mean and
// M2 (2 x W*H float4, zeros outside the shard's own tiles) and SUM-reduces them to the first device with ONE ncclReduce
// per device inside ncclGroupStart/End: tiles are disjoint, so the sum is the exact merged frame.  Whole-frame
// quantities (Reinhard's average luminance, Camera::isConverged's count) then run on the merged buffers on the first
// device with the single-GPU arithmetic (ct_tonemap_buffer / ct_is_converged_buffers).
//
// The Python host does the same thing with one process per GPU over torch.distributed (deepestscatter_amd/distributed.py:
// that is what bench.py times); this file is what a C or C++ caller -- the reference's own host is one -- uses instead:
// deepestscatter_amd/host/main.cpp --gpus N.
//
// librccl is loaded with dlopen when the first group is created, so libcloudtrace.so itself does not depend on it (and
// a process that already holds a copy, e.g. PyTorch's, shares it).  A missing library is CT_E_RCCL, never a fallback.
// RCCL refuses two ranks on one device; a device list that repeats a device (a rehearsal on a one-GPU box) therefore
// merges with peer copies and an add kernel on the first device instead -- the same sums, no communicator.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <set>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cloudtrace.h"

namespace {

// the part of rccl.h this file needs (the header's own declarations would make the symbols link-time dependencies)
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclSuccess = 0, kNcclFloat = 7, kNcclSum = 0;

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;

    std::mutex mu;     // two threads may create their first group at once

    bool load()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (lib) {
            return true;
        }
        void *l = nullptr;
        std::string why;
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            l = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (l) {
                break;
            }
            const char *e = dlerror();   // (returns the message ONCE and clears it)
            why = e ? e : "?";
        }
        if (!l) {
            error = "cannot load librccl: " + why;
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(l, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Reduce = (decltype(Reduce))sym("ncclReduce");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Reduce || !GroupStart || !GroupEnd || !GetErrorString) {
            error = "librccl lacks an expected symbol";
            return false;
        }
        lib = l;
        return true;
    }
};

Rccl g_rccl;

__global__ void add_into_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = dst[i], b = src[i];
        dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

} // namespace

struct CtGroup_ {
    std::vector<CtHandle> handles;
    std::vector<int> devices;
    std::vector<float *> staging;       // per shard, on its device: [mean | M2], 2 * W*H float4
    std::vector<hipStream_t> streams;   // per shard: the stream of the merge
    std::vector<ncclComm_t> comms;      // empty in the rehearsal mode (a device appears twice)
    float *scratch = nullptr;           // rehearsal mode: landing buffer on the first device
    uint32_t width = 0, height = 0, subframes = 0;
    bool merged = false;
    std::string error;
};

static thread_local std::string g_group_create_error;

static int gfail(CtGroup g, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) {
        g->error = buf;
    } else {
        g_group_create_error = buf;
    }
    return code;
}

#define GHIP(g, expr)                                                                                           \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            return gfail((g), e_ == hipErrorOutOfMemory ? CT_E_NOMEM : CT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                                       \
    } while (0)

#define GNCCL(g, expr)                                                                        \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != kNcclSuccess) {                                                             \
            return gfail((g), CT_E_RCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
        }                                                                                     \
    } while (0)

// A shard's failure, with its message.
static int shard_fail(CtGroup g, uint32_t i, int rc)
{
    return gfail(g, rc, "shard %u (device %d): %s", i, g->devices[i], ct_last_error(g->handles[i]));
}

extern "C" int ct_group_destroy(CtGroup g)
{
    if (!g) {
        return CT_OK;
    }
    for (ncclComm_t c : g->comms) {
        if (c) {
            g_rccl.CommDestroy(c);
        }
    }
    for (size_t i = 0; i < g->handles.size(); i++) {
        if (i < g->devices.size()) {
            hipSetDevice(g->devices[i]);
        }
        if (i < g->streams.size() && g->streams[i]) {
            hipStreamSynchronize(g->streams[i]);
            hipStreamDestroy(g->streams[i]);
        }
        if (i < g->staging.size() && g->staging[i]) {
            hipFree(g->staging[i]);
        }
        ct_destroy(g->handles[i]);
    }
    if (g->scratch) {
        hipSetDevice(g->devices[0]);
        hipFree(g->scratch);
    }
    delete g;
    return CT_OK;
}

extern "C" int ct_group_create(const CtScene *scene, const int32_t *devices, uint32_t count, CtGroup *out)
{
    if (!out) {
        return gfail(nullptr, CT_E_INVAL, "out is NULL");
    }
    *out = nullptr;
    if (!scene || !devices || count == 0 || count > 64) {
        return gfail(nullptr, CT_E_INVAL, "ct_group_create: need a scene and 1..64 devices");
    }
    CtGroup g = new (std::nothrow) CtGroup_();
    if (!g) {
        return gfail(nullptr, CT_E_NOMEM, "out of host memory");
    }
    auto bail = [&](int rc) {
        g_group_create_error = g->error;
        ct_group_destroy(g);
        return rc;
    };
    g->width = scene->width;
    g->height = scene->height;
    g->devices.assign(devices, devices + count);
    const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == count;
    const size_t floats = (size_t)2 * scene->width * scene->height * 4;
    for (uint32_t i = 0; i < count; i++) {
        CtScene s = *scene;
        s.device = devices[i];
        s.shard_index = i;
        s.shard_count = count;
        CtHandle h = nullptr;
        const int rc = ct_create(&s, &h);
        if (rc != CT_OK) {
            g->error = std::string("shard ") + std::to_string(i) + ": " + ct_last_error(nullptr);
            return bail(rc);
        }
        g->handles.push_back(h);
        float *st = nullptr;
        hipStream_t stream = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc((void **)&st, floats * sizeof(float)) != hipSuccess ||
            hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
            g->staging.push_back(st);
            g->streams.push_back(stream);
            g->error = "out of device memory (merge staging buffer)";
            return bail(CT_E_NOMEM);
        }
        g->staging.push_back(st);
        g->streams.push_back(stream);
    }
    if (distinct) {
        if (!g_rccl.load()) {
            g->error = g_rccl.error;
            return bail(CT_E_RCCL);
        }
        g->comms.assign(count, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), (int)count, g->devices.data());
        if (r != kNcclSuccess) {
            g->comms.clear();
            g->error = std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(r);
            return bail(CT_E_RCCL);
        }
    } else {
        // rehearsal: several shards on one device (RCCL refuses that); merge with copies and adds on the first device
        if (hipSetDevice(devices[0]) != hipSuccess || hipMalloc((void **)&g->scratch, floats * sizeof(float)) != hipSuccess) {
            g->error = "out of device memory (merge scratch)";
            return bail(CT_E_NOMEM);
        }
    }
    *out = g;
    return CT_OK;
}

extern "C" const char *ct_group_last_error(CtGroup g)
{
    return g ? g->error.c_str() : g_group_create_error.c_str();
}

extern "C" int ct_group_size(CtGroup g, uint32_t *count_out)
{
    if (!g || !count_out) {
        return gfail(g, CT_E_INVAL, "null argument");
    }
    *count_out = (uint32_t)g->handles.size();
    return CT_OK;
}

extern "C" int ct_group_handle(CtGroup g, uint32_t index, CtHandle *out)
{
    if (!g || !out || index >= g->handles.size()) {
        return gfail(g, CT_E_INVAL, "ct_group_handle: index out of range");
    }
    *out = g->handles[index];
    return CT_OK;
}

extern "C" int ct_group_set_camera(CtGroup g, const float eye[3], const float U[3], const float V[3], const float W[3])
{
    if (!g) {
        return gfail(nullptr, CT_E_INVAL, "null group");
    }
    for (uint32_t i = 0; i < g->handles.size(); i++) {
        const int rc = ct_set_camera(g->handles[i], eye, U, V, W);
        if (rc != CT_OK) {
            return shard_fail(g, i, rc);
        }
    }
    g->merged = false;
    return CT_OK;
}

// Every shard enqueues its batch (the devices work at the same time), then every shard is waited for.
extern "C" int ct_group_render_accumulate(CtGroup g, uint32_t first_subframe_id, uint32_t count)
{
    if (!g) {
        return gfail(nullptr, CT_E_INVAL, "null group");
    }
    for (uint32_t i = 0; i < g->handles.size(); i++) {
        const int rc = ct_render_accumulate_async(g->handles[i], first_subframe_id, count);
        if (rc != CT_OK) {
            return shard_fail(g, i, rc);
        }
    }
    for (uint32_t i = 0; i < g->handles.size(); i++) {
        const int rc = ct_synchronize(g->handles[i]);
        if (rc != CT_OK) {
            return shard_fail(g, i, rc);
        }
    }
    g->subframes = first_subframe_id + count - 1;
    g->merged = false;
    return CT_OK;
}

extern "C" int ct_group_reset(CtGroup g)
{
    if (!g) {
        return gfail(nullptr, CT_E_INVAL, "null group");
    }
    for (uint32_t i = 0; i < g->handles.size(); i++) {
        const int rc = ct_reset(g->handles[i]);
        if (rc != CT_OK) {
            return shard_fail(g, i, rc);
        }
    }
    g->subframes = 0;
    g->merged = false;
    return CT_OK;
}

// The frame reduce: [mean | M2] of every shard, summed onto the first device.
extern "C" int ct_group_merge(CtGroup g)
{
    if (!g) {
        return gfail(nullptr, CT_E_INVAL, "null group");
    }
    const size_t n = (size_t)g->width * g->height;       // float4 per buffer
    const size_t bytes = n * sizeof(float4);
    for (uint32_t i = 0; i < g->handles.size(); i++) {
        int rc = ct_copy_to_device(g->handles[i], CT_BUF_MEAN, g->staging[i], bytes);
        if (rc == CT_OK) {
            rc = ct_copy_to_device(g->handles[i], CT_BUF_M2, (float4 *)g->staging[i] + n, bytes);
        }
        if (rc != CT_OK) {
            return shard_fail(g, i, rc);
        }
    }
    if (!g->comms.empty()) {
        GNCCL(g, g_rccl.GroupStart());
        for (uint32_t i = 0; i < g->handles.size(); i++) {
            GHIP(g, hipSetDevice(g->devices[i]));
            GNCCL(g, g_rccl.Reduce(g->staging[i], g->staging[i], 2 * n * 4, kNcclFloat, kNcclSum, 0, g->comms[i], g->streams[i]));
        }
        GNCCL(g, g_rccl.GroupEnd());
        for (uint32_t i = 0; i < g->handles.size(); i++) {
            GHIP(g, hipSetDevice(g->devices[i]));
            GHIP(g, hipStreamSynchronize(g->streams[i]));
        }
    } else {
        GHIP(g, hipSetDevice(g->devices[0]));
        (void)hipGetLastError();
        for (uint32_t i = 1; i < g->handles.size(); i++) {
            GHIP(g, hipMemcpyPeerAsync(g->scratch, g->devices[0], g->staging[i], g->devices[i], 2 * bytes, g->streams[0]));
            hipLaunchKernelGGL(add_into_kernel, dim3(1024), dim3(256), 0, g->streams[0], (float4 *)g->staging[0],
                               (const float4 *)g->scratch, 2 * n);
            GHIP(g, hipGetLastError());
        }
        GHIP(g, hipStreamSynchronize(g->streams[0]));
    }
    g->merged = true;
    return CT_OK;
}

static int need_merge(CtGroup g)
{
    if (!g) {
        return gfail(nullptr, CT_E_INVAL, "null group");
    }
    return g->merged ? CT_OK : ct_group_merge(g);
}

extern "C" int ct_group_download(CtGroup g, int32_t which, void *dst_host, size_t dst_bytes)
{
    const int rc = need_merge(g);
    if (rc != CT_OK) {
        return rc;
    }
    const size_t n = (size_t)g->width * g->height, bytes = n * sizeof(float4);
    if ((which != CT_BUF_MEAN && which != CT_BUF_M2) || !dst_host || dst_bytes != bytes) {
        return gfail(g, CT_E_INVAL, "ct_group_download: CT_BUF_MEAN or CT_BUF_M2, %zu bytes", bytes);
    }
    GHIP(g, hipSetDevice(g->devices[0]));
    GHIP(g, hipMemcpy(dst_host, (float4 *)g->staging[0] + (which == CT_BUF_M2 ? n : 0), bytes, hipMemcpyDeviceToHost));
    return CT_OK;
}

extern "C" int ct_group_tonemap(CtGroup g, float exposure, uint8_t *rgba_host, float *avg_luminance_out)
{
    int rc = need_merge(g);
    if (rc != CT_OK) {
        return rc;
    }
    rc = ct_tonemap_buffer(g->handles[0], g->staging[0], exposure, rgba_host, avg_luminance_out);
    return rc == CT_OK ? CT_OK : shard_fail(g, 0, rc);
}

extern "C" int ct_group_is_converged(CtGroup g, int32_t *converged_out, uint64_t *unconverged_pixels_out)
{
    int rc = need_merge(g);
    if (rc != CT_OK) {
        return rc;
    }
    const size_t n = (size_t)g->width * g->height;
    rc = ct_is_converged_buffers(g->handles[0], g->staging[0], (float *)((float4 *)g->staging[0] + n), g->subframes, converged_out,
                                 unconverged_pixels_out);
    return rc == CT_OK ? CT_OK : shard_fail(g, 0, rc);
}

extern "C" int ct_group_counters(CtGroup g, CtCounters *out)
{
    if (!g || !out) {
        return gfail(g, CT_E_INVAL, "null argument");
    }
    std::memset(out, 0, sizeof *out);
    for (uint32_t i = 0; i < g->handles.size(); i++) {
        CtCounters c;
        const int rc = ct_counters(g->handles[i], &c);
        if (rc != CT_OK) {
            return shard_fail(g, i, rc);
        }
        out->paths += c.paths;
        out->box_hits += c.box_hits;
        out->density_lookups += c.density_lookups;
        out->inscatter_lookups += c.inscatter_lookups;
        out->scatter_events += c.scatter_events;
        out->depth_capped += c.depth_capped;
    }
    return CT_OK;
}
