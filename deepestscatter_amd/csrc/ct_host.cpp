// ct_host.cpp -- CPU-side pieces of the path that the reference also runs on the host:
// camera frame (sutil.cpp:501-524), volume quantiser + mip pyramid (Resources.cpp:92-209),
// and the synthetic cloud generator used as the benchmark input.
// Pure C++; no GPU, no HIP calls.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/cloudtrace.h"
#include "../host/VdbReader.h"

namespace {

struct V3 {
    float x, y, z;
};
inline V3 cross(V3 a, V3 b) { return V3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 scale(V3 a, float s) { return V3{ a.x * s, a.y * s, a.z * s }; }
inline V3 normalize(V3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }

constexpr float kPiF = 3.14159265358979323846f;

template <typename F>
void parallel_for(uint32_t n, F &&body)
{
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned workers = std::max(1u, std::min(hw ? hw : 1u, std::min(n, 64u)));
    if (workers == 1) {
        for (uint32_t i = 0; i < n; i++) {
            body(i);
        }
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < workers; w++) {
        pool.emplace_back([&, w] {
            for (uint32_t i = w; i < n; i += workers) {
                body(i);
            }
        });
    }
    for (auto &t : pool) {
        t.join();
    }
}

// ---- gradient noise: integer hash -> one of 12 edge directions; quintic fade ---------------
inline uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
    uint32_t h = seed ^ (x * 0x8da6b343u) ^ (y * 0xd8163841u) ^ (z * 0xcb1ab31fu);
    h ^= h >> 15;
    h *= 0x2c1b3c6du;
    h ^= h >> 12;
    h *= 0x297a2d39u;
    h ^= h >> 15;
    return h;
}

inline float grad(uint32_t h, float x, float y, float z)
{
    switch (h % 12u) {
    case 0: return x + y;
    case 1: return -x + y;
    case 2: return x - y;
    case 3: return -x - y;
    case 4: return x + z;
    case 5: return -x + z;
    case 6: return x - z;
    case 7: return -x - z;
    case 8: return y + z;
    case 9: return -y + z;
    case 10: return y - z;
    default: return -y - z;
    }
}

inline float fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
inline float mix(float a, float b, float t) { return a + (b - a) * t; }

float perlin(float x, float y, float z, uint32_t seed)
{
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const uint32_t ix = (uint32_t)(int32_t)fx, iy = (uint32_t)(int32_t)fy, iz = (uint32_t)(int32_t)fz;
    const float rx = x - fx, ry = y - fy, rz = z - fz;
    const float u = fade(rx), v = fade(ry), w = fade(rz);
    float c[8];
    for (uint32_t d = 0; d < 8; d++) {
        const uint32_t dx = d & 1u, dy = (d >> 1) & 1u, dz = d >> 2;
        c[d] = grad(hash3(ix + dx, iy + dy, iz + dz, seed), rx - (float)dx, ry - (float)dy, rz - (float)dz);
    }
    const float x00 = mix(c[0], c[1], u), x10 = mix(c[2], c[3], u);
    const float x01 = mix(c[4], c[5], u), x11 = mix(c[6], c[7], u);
    return mix(mix(x00, x10, v), mix(x01, x11, v), w);
}

} // namespace

extern "C" int ct_calculate_camera_variables(const float eye[3], const float lookat[3], const float up[3],
                                             float hfov_deg, float aspect_ratio, float U_out[3], float V_out[3],
                                             float W_out[3])
{
    if (!eye || !lookat || !up || !U_out || !V_out || !W_out || !(aspect_ratio > 0.f)) {
        return CT_E_INVAL;
    }
    // sutil.cpp:505-523 with fov_is_vertical == false (Camera.cpp:109)
    const V3 W{ lookat[0] - eye[0], lookat[1] - eye[1], lookat[2] - eye[2] }; // not normalised: focal length
    const float wlen = sqrtf(dot(W, W));
    V3 U = normalize(cross(W, V3{ up[0], up[1], up[2] }));
    V3 V = normalize(cross(U, W));
    const float ulen = wlen * tanf(0.5f * hfov_deg * kPiF / 180.0f);
    U = scale(U, ulen);
    const float vlen = ulen / aspect_ratio;
    V = scale(V, vlen);
    U_out[0] = U.x; U_out[1] = U.y; U_out[2] = U.z;
    V_out[0] = V.x; V_out[1] = V.y; V_out[2] = V.z;
    W_out[0] = W.x; W_out[1] = W.y; W_out[2] = W.z;
    return CT_OK;
}

extern "C" int ct_quantize_volume(const float *grid, const uint32_t pd[3], uint8_t *tex)
{
    if (!grid || !pd || !tex || pd[0] == 0 || pd[1] == 0 || pd[2] == 0) {
        return CT_E_INVAL;
    }
    const size_t n = (size_t)pd[0] * pd[1] * pd[2];
    double max_density = grid[0]; // openvdb::tools::extrema(...).max(), Resources.cpp:92-95
    for (size_t i = 1; i < n; i++) {
        max_density = std::max(max_density, (double)grid[i]);
    }
    const uint32_t tx = pd[0] + 2, ty = pd[1] + 2, tz = pd[2] + 2; // expandBy(1), :97-101
    memset(tex, 0, (size_t)tx * ty * tz);
    parallel_for(pd[2], [&](uint32_t z) {
        for (uint32_t y = 0; y < pd[1]; y++) {
            const float *src = grid + ((size_t)z * pd[1] + y) * pd[0];
            uint8_t *dst = tex + ((size_t)(z + 1) * ty + (y + 1)) * tx + 1;
            for (uint32_t x = 0; x < pd[0]; x++) {
                dst[x] = (uint8_t)(src[x] / max_density * 255); // narrow_cast<uint8_t>, :137
            }
        }
    });
    return CT_OK;
}

extern "C" int ct_generate_mipmaps(const uint8_t *level0, const uint32_t dims[3], uint8_t *out, size_t capacity,
                                   uint32_t *levels_out, size_t *bytes_out, size_t *level_offsets_out)
{
    if (!level0 || !dims || dims[0] == 0 || dims[1] == 0 || dims[2] == 0) {
        return CT_E_INVAL;
    }
    // level count, Resources.cpp:103-117
    uint32_t m = std::max({ dims[0], dims[1], dims[2] });
    uint32_t levels = 1;
    while (m /= 2) {
        levels++;
    }
    auto ld = [&](uint32_t d, uint32_t l) { return std::max(d >> l, 1u); };
    size_t total = 0;
    for (uint32_t l = 0; l < levels; l++) {
        if (level_offsets_out && l < 32) {
            level_offsets_out[l] = total;
        }
        total += (size_t)ld(dims[0], l) * ld(dims[1], l) * ld(dims[2], l);
    }
    if (levels_out) {
        *levels_out = levels;
    }
    if (bytes_out) {
        *bytes_out = total;
    }
    if (!out) {
        return CT_OK;
    }
    if (capacity < total) {
        return CT_E_INVAL;
    }
    memcpy(out, level0, (size_t)dims[0] * dims[1] * dims[2]);
    const uint8_t *prev = out;
    size_t off = (size_t)dims[0] * dims[1] * dims[2];
    uint32_t px = dims[0], py = dims[1], pz = dims[2];
    for (uint32_t l = 1; l < levels; l++) { // generateMipmaps, :169-209
        const uint32_t cx = ld(dims[0], l), cy = ld(dims[1], l), cz = ld(dims[2], l);
        uint8_t *cur = out + off;
        parallel_for(cz, [&](uint32_t z) {
            for (uint32_t y = 0; y < cy; y++) {
                for (uint32_t x = 0; x < cx; x++) {
                    uint16_t acc = 0;
                    for (uint32_t d = 0; d < 8; d++) {
                        const uint32_t sx = 2 * x + (d & 1u), sy = 2 * y + ((d >> 1) & 1u), sz = 2 * z + (d >> 2);
                        if (sx < px && sy < py && sz < pz) { // TextureView3D::get -> 0 out of range
                            acc = (uint16_t)(acc + prev[((size_t)sz * py + sy) * px + sx]);
                        }
                    }
                    cur[((size_t)z * cy + y) * cx + x] = (uint8_t)(acc / 8);
                }
            }
        });
        prev = cur;
        off += (size_t)cx * cy * cz;
        px = cx; py = cy; pz = cz;
    }
    return CT_OK;
}

extern "C" int ct_make_procedural_cloud(uint32_t n, uint32_t seed, uint8_t *tex)
{
    if (n < 4 || n > 4096 || !tex) {
        return CT_E_INVAL;
    }
    const uint32_t p = n - 2; // payload per axis; the quantiser adds the zero border back
    float *grid = new (std::nothrow) float[(size_t)p * p * p];
    if (!grid) {
        return CT_E_NOMEM;
    }
    const float inv = 1.0f / (float)p;
    const float semi[3] = { 0.45f, 0.30f, 0.40f };
    parallel_for(p, [&](uint32_t k) {
        for (uint32_t j = 0; j < p; j++) {
            for (uint32_t i = 0; i < p; i++) {
                const float x = ((float)i + 0.5f) * inv, y = ((float)j + 0.5f) * inv, z = ((float)k + 0.5f) * inv;
                float amp = 1.0f, freq = 4.0f, sum = 0.0f, norm = 0.0f;
                for (uint32_t o = 0; o < 5; o++) {
                    sum += amp * perlin(x * freq, y * freq, z * freq, seed + o * 0x9e3779b9u);
                    norm += amp;
                    amp *= 0.5f;
                    freq *= 2.0f;
                }
                const float nval = 0.5f + 0.5f * (sum / norm) * 1.6f; // ~[0,1]
                const float ex = (x - 0.5f) / semi[0], ey = (y - 0.5f) / semi[1], ez = (z - 0.5f) / semi[2];
                const float r2 = ex * ex + ey * ey + ez * ez;
                const float f = r2 < 1.0f ? (1.0f - r2) * (1.0f - r2) * (3.0f - 2.0f * (1.0f - r2)) : 0.0f;
                grid[((size_t)k * p + j) * p + i] = std::max(0.0f, nval * f - 0.12f);
            }
        }
    });
    const uint32_t pd[3] = { p, p, p };
    const int rc = ct_quantize_volume(grid, pd, tex);
    delete[] grid;
    return rc;
}


// Resources::loadVolumeBuffer for a .vdb file (Resources.cpp:82-143) without OpenVDB: host/VdbReader.h.
extern "C" int ct_load_vdb(const char *path, uint32_t dims_out[3], uint8_t *texture_host_out, size_t capacity, size_t *bytes_out,
                           char *error_out, size_t error_capacity)
{
    auto report = [&](const char *msg) {
        if (error_out && error_capacity) {
            std::strncpy(error_out, msg, error_capacity - 1);
            error_out[error_capacity - 1] = 0;
        }
    };
    if (!path || !dims_out) {
        report("path / dims_out is NULL");
        return CT_E_INVAL;
    }
    try {
        std::vector<uint8_t> tex;
        std::array<uint32_t, 3> dims{};
        DeepestScatter::vdb::loadVolumeTexture(path, tex, dims);
        dims_out[0] = dims[0];
        dims_out[1] = dims[1];
        dims_out[2] = dims[2];
        if (bytes_out) {
            *bytes_out = tex.size();
        }
        if (texture_host_out) {
            if (capacity < tex.size()) {
                report("capacity too small");
                return CT_E_INVAL;
            }
            std::memcpy(texture_host_out, tex.data(), tex.size());
        }
        return CT_OK;
    } catch (const std::bad_alloc &) {
        report("out of host memory");
        return CT_E_NOMEM;
    } catch (const std::exception &e) {
        report(e.what());
        return CT_E_INVAL;
    }
}
