// ct_kernels.hip -- gfx950 kernels of the cloud path tracer.
//
//   render_persistent_kernel   the estimator (cloudRadianceMaterials.cu:9-148 + cloud.cuh:77-188 +
//                              pathTracingCamera.cu:12-21 + cloudBBox.cu:7-37) as a persistent,
//                              wave-scheduled state machine: path regeneration from a job queue, march
//                              bursts / scatter phases, exact free-space skipping, suspension of the
//                              surviving paths at the end of a launch and their resumption by the next
//   render_delta_kernel        the same scheduler around Woodcock tracking over a grid of majorant cells that
//                              every block keeps in LDS (DELTA); majorant_cells_kernel builds the grid
//   render_simple_kernel       the MARCH estimator, one thread per pixel, nested loops (A/B and cross-check)
//   primary_rays_kernel, primary_advance_kernel, primary_advance_delta_kernel
//                              per pose: primary ray per pixel and the part of its flight that is the same
//                              for every sample of the pixel
//   accumulate_batch_kernel, accumulate_list_kernel
//                              updateFrameResult (progressive.cu:17-27) over S subframes in order
//   inscatter_kernel           inScatter.cu:40-66
//   build_bricks_kernel, brick_free_kernel, brick_dist_relax_kernel, brick_meta_kernel
//                              uint8 volume -> 128-byte apron bricks of 4^3 texels + meta bytes
//   blocked_mask_kernel, cheb_pass_kernel, build_mbricks_kernel
//                              texel-granular clearance (capped Chebyshev distance transform) and the
//                              3x4x4 march bricks with a meta byte per row
//   point_rays_kernel, point_accumulate_kernel, scatter_samples_kernel
//                              the dataset collectors' device parts (pointEmissionCamera.cu,
//                              PointRadianceTask.h, pointGeneratorCamera.cu, cloudFirstScatterMaterial.cu)
//   mip_level_kernel, descriptor_kernel
//                              Resources::generateMipmaps + setupHierarchicalDescriptor (DisneyDescriptor.cuh)
//   reinhard_fused_kernel      reinhard.cu:26-84 (firstPass + secondPass + applyReinhard in one launch)
//   converged_kernel           Camera::isConverged (Camera.cpp:232-268)
//   cdf_selftest_kernel, fetch_probe_kernel
//                              diagnostics (exhaustive CDF inversion test, FETCH_SIZE calibration)
//
// Compile with -ffp-contract=off: results must be bit-identical to oracle/ct_oracle.c.
#include <algorithm>
#include <cstdlib>

#include "ct_internal.hpp"

namespace ct {

// =============================================================================================
// corner cells
// =============================================================================================
// One thread per byte of the brick array: brick b = (bz*gy + by)*gx + bx holds the texels
// [4b - bias, 4b - bias + 4]^3 with clamp-to-edge at byte lz*25 + ly*5 + lx; bytes 125..127 pad.
__global__ void build_bricks_kernel(const uint8_t *__restrict__ t, int nx, int ny, int nz, int bias,
                                    uint8_t *__restrict__ bricks, int gx, int gy, int gz)
{
    const int64_t total = (int64_t)gx * gy * gz * 128;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i >> 7;
        const int o = (int)(i & 127);
        uint8_t v = 0;
        if (o < 125) {
            const int lx = o % 5, ly = (o / 5) % 5, lz = o / 25;
            const int x = (int)(b % gx) * 4 + lx - bias;
            const int y = (int)((b / gx) % gy) * 4 + ly - bias;
            const int z = (int)(b / ((int64_t)gx * gy)) * 4 + lz - bias;
            const int xc = min(max(x, 0), nx - 1), yc = min(max(y, 0), ny - 1), zc = min(max(z, 0), nz - 1);
            v = t[((size_t)zc * ny + yc) * nx + xc];
        }
        bricks[i] = v;
    }
}

// Twin bricks (DevScene::tbricks): one thread per byte; brick b holds the texels [3b - bias, 3b - bias + 3]^3 of the density
// (bytes 0..63) and of the shadow volume (bytes 64..127), clamp-to-edge applied, at byte lz*16 + ly*4 + lx of each half.
__global__ void build_twin_bricks_kernel(const uint8_t *__restrict__ density, const uint8_t *__restrict__ shadow, int nx, int ny, int nz,
                                         int bias, uint8_t *__restrict__ bricks, int gx, int gy, int gz)
{
    const int64_t total = (int64_t)gx * gy * gz * 128;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i >> 7;
        const int o = (int)(i & 63);
        const int lx = o & 3, ly = (o >> 2) & 3, lz = o >> 4;
        const int x = (int)(b % gx) * 3 + lx - bias;
        const int y = (int)((b / gx) % gy) * 3 + ly - bias;
        const int z = (int)(b / ((int64_t)gx * gy)) * 3 + lz - bias;
        const int xc = min(max(x, 0), nx - 1), yc = min(max(y, 0), ny - 1), zc = min(max(z, 0), nz - 1);
        const uint8_t *t = (i & 64) ? shadow : density;
        bricks[i] = t[((size_t)zc * ny + yc) * nx + xc];
    }
}

hipError_t launch_build_twin_bricks(const uint8_t *density, const uint8_t *shadow, int nx, int ny, int nz, int bias, int gx, int gy, int gz,
                                    uint8_t *bricks, hipStream_t stream)
{
    const int64_t total = (int64_t)gx * gy * gz * 128;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 65536 ? (total + threads - 1) / threads : 65536);
    hipLaunchKernelGGL(build_twin_bricks_kernel, dim3(blocks), dim3(threads), 0, stream, density, shadow, nx, ny, nz, bias, bricks, gx, gy, gz);
    return hipGetLastError();
}

hipError_t launch_build_bricks(const uint8_t *texels, int nx, int ny, int nz, int bias, int gx, int gy, int gz,
                               uint8_t *bricks, hipStream_t stream)
{
    const int64_t total = (int64_t)gx * gy * gz * 128;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 65536 ? (total + threads - 1) / threads : 65536);
    hipLaunchKernelGGL(build_bricks_kernel, dim3(blocks), dim3(threads), 0, stream, texels, nx, ny, nz, bias, bricks,
                       gx, gy, gz);
    return hipGetLastError();
}

// =============================================================================================
// march bricks (DevScene::mbricks): texel-granular clearance + 3x4x4 bricks with a meta byte per row
// =============================================================================================
constexpr int kClearMax = 127;

// blocked[b] = 1 when a march step may not be skipped at base texel b: the footprint based there
// has a non-zero texel, or b is not "interior" (1 <= b <= N-3 on every axis).
__global__ void blocked_mask_kernel(const uint8_t *__restrict__ t, int nx, int ny, int nz, uint8_t *__restrict__ blocked)
{
    const int64_t total = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((int64_t)nx * ny));
        const bool interior = x >= 1 && y >= 1 && z >= 1 && x <= nx - 3 && y <= ny - 3 && z <= nz - 3;
        bool b = !interior;
        if (!b) {
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                m |= t[((size_t)(z + (k >> 2)) * ny + (y + ((k >> 1) & 1))) * nx + (x + (k & 1))];
            }
            b = m != 0;
        }
        blocked[i] = b ? 1 : 0;
    }
}

// One separable pass of the capped Chebyshev distance transform along AXIS (0 x, 1 y, 2 z):
//   pass 0: out = min |x - x'| over blocked x' of the row;
//   pass 1, 2: out = min over k of max(k, in[.. -+ k ..]).
// Everything outside the grid counts as blocked.  The scan walks outwards and stops at the first k
// that cannot improve the result, so its cost is proportional to the distance found.
template <int AXIS>
__global__ void cheb_pass_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int nx, int ny, int nz)
{
    const int64_t total = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((int64_t)nx * ny));
        const int c = AXIS == 0 ? x : (AXIS == 1 ? y : z);
        const int n = AXIS == 0 ? nx : (AXIS == 1 ? ny : nz);
        const int64_t stride = AXIS == 0 ? 1 : (AXIS == 1 ? (int64_t)nx : (int64_t)nx * ny);
        int best = kClearMax + 1;
        for (int k = 0; k < best; k++) {
            int v = AXIS == 0 ? (kClearMax + 1) : 0; // value at an out-of-grid neighbour: blocked
            bool hit = false;
            if (c - k >= 0) {
                const int a = in[i - k * stride];
                if (AXIS == 0) {
                    hit = hit || a != 0;
                } else {
                    v = a;
                }
            } else {
                hit = true;
                v = 0;
            }
            int w = AXIS == 0 ? (kClearMax + 1) : 0;
            if (c + k < n) {
                const int a = in[i + k * stride];
                if (AXIS == 0) {
                    hit = hit || a != 0;
                } else {
                    w = a;
                }
            } else {
                hit = true;
                w = 0;
            }
            if (AXIS == 0) {
                if (hit) {
                    best = k;
                }
            } else {
                best = min(best, max(k, min(v, w)));
            }
        }
        out[i] = (uint8_t)min(best, kClearMax + 1);
    }
}

// One thread per byte of the march-brick array.  dist = Chebyshev distance (texels, capped) of a
// base texel to the nearest blocked one; a row's clearance is min over its three bases of dist-1.
__global__ void build_mbricks_kernel(const uint8_t *__restrict__ t, const uint8_t *__restrict__ dist, int nx, int ny,
                                     int nz, int bias_x, int bias, uint8_t *__restrict__ bricks, int gx, int gy, int gz)
{
    const int64_t total = (int64_t)gx * gy * gz * 128;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i >> 7;
        const int o = (int)(i & 127);
        uint8_t v = 0;
        if (o < 125) {
            const int lx = o % 5, ly = (o / 5) % 5, lz = o / 25;
            const int x0 = (int)(b % gx) * 3 - bias_x;
            const int y = (int)((b / gx) % gy) * 4 + ly - bias;
            const int z = (int)(b / ((int64_t)gx * gy)) * 4 + lz - bias;
            if (lx < 4) {
                const int x = x0 + lx;
                const int xc = min(max(x, 0), nx - 1), yc = min(max(y, 0), ny - 1), zc = min(max(z, 0), nz - 1);
                v = t[((size_t)zc * ny + yc) * nx + xc];
            } else if (ly < 4 && lz < 4) {
                // meta byte of the base row (ly, lz): bases x0 .. x0+2
                int c = kClearMax;
                bool interior = true;
                for (int k = 0; k < 3; k++) {
                    const int x = x0 + k;
                    const bool in_grid = x >= 0 && y >= 0 && z >= 0 && x < nx && y < ny && z < nz;
                    const int d = in_grid ? (int)dist[((size_t)z * ny + y) * nx + x] : 0;
                    c = min(c, d - 1);
                    interior = interior && x >= 1 && y >= 1 && z >= 1 && x <= nx - 3 && y <= ny - 3 && z <= nz - 3;
                }
                v = (uint8_t)(max(c, 0) | (interior ? 0x80 : 0));
            }
        }
        bricks[i] = v;
    }
}

// (leaves the distance volume in tmp_b: launch_coarse_clearance reads it)
hipError_t launch_build_mbricks(const uint8_t *texels, int nx, int ny, int nz, int bias_x, int bias, int gx, int gy,
                                int gz, uint8_t *tmp_a, uint8_t *tmp_b, uint8_t *bricks, hipStream_t stream)
{
    const int64_t texels_n = (int64_t)nx * ny * nz;
    const int threads = 256;
    const int tb = (int)std::min<int64_t>((texels_n + threads - 1) / threads, 65536 * 4);
    hipLaunchKernelGGL(blocked_mask_kernel, dim3(tb), dim3(threads), 0, stream, texels, nx, ny, nz, tmp_a);
    hipLaunchKernelGGL(cheb_pass_kernel<0>, dim3(tb), dim3(threads), 0, stream, tmp_a, tmp_b, nx, ny, nz);
    hipLaunchKernelGGL(cheb_pass_kernel<1>, dim3(tb), dim3(threads), 0, stream, tmp_b, tmp_a, nx, ny, nz);
    hipLaunchKernelGGL(cheb_pass_kernel<2>, dim3(tb), dim3(threads), 0, stream, tmp_a, tmp_b, nx, ny, nz);
    const int64_t total = (int64_t)gx * gy * gz * 128;
    const int bb = (int)std::min<int64_t>((total + threads - 1) / threads, 65536 * 4);
    hipLaunchKernelGGL(build_mbricks_kernel, dim3(bb), dim3(threads), 0, stream, texels, tmp_b, nx, ny, nz, bias_x, bias,
                       bricks, gx, gy, gz);
    return hipGetLastError();
}

#ifdef CT_EXPERIMENTS
// ---- march bricks with dense ADDRESSING and sparse BACKING (CT_FLAG_VMM_BRICKS; ct_api.cpp maps the chunks) ---------------
// Texel bytes: lx < 4; meta bytes: lx == 4 of the base rows ly, lz < 4 (build_mbricks_kernel).  A chunk without a non-zero
// texel byte is described by its meta bytes alone, and its clearances may be rounded DOWN without changing a result (a
// smaller clearance only shortens the exact free-space skip): quantised to {0, 4, 8, 16, 32, 64, 127} texels, two such chunks
// with the same bytes can share one piece of memory.
CT_DEV uint32_t quantize_meta(uint32_t v)
{
    const uint32_t c = v & 0x7fu;
    const uint32_t q = c >= 127u ? 127u : (c >= 64u ? 64u : (c >= 32u ? 32u : (c >= 16u ? 16u : (c >= 8u ? 8u : (c >= 4u ? 4u : 0u)))));
    return (v & 0x80u) | q;
}

// One block per chunk: out[chunk] = (any texel byte non-zero, 0, hash of the quantised meta bytes: low, high word).
__global__ __launch_bounds__(256) void mbrick_chunk_class_kernel(const uint8_t *__restrict__ bricks, int64_t total_bytes, int64_t chunk_bytes,
                                                                 uint4 *__restrict__ out)
{
    __shared__ uint32_t s_data;
    __shared__ unsigned long long s_hash;
    if (threadIdx.x == 0) {
        s_data = 0;
        s_hash = 0;
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * chunk_bytes, end = min(base + chunk_bytes, total_bytes);
    uint32_t data = 0;
    unsigned long long hsh = 0;   // order-independent: a sum of per-byte mixes of (position in the chunk, quantised value)
    for (int64_t i = base + (int64_t)threadIdx.x * 4; i < end; i += 256 * 4) {
        const uint32_t w = *(const uint32_t *)(bricks + i);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t v = (w >> (8 * k)) & 0xffu;
            const int o = (int)((i + k) & 127);
            if (o < 125) {
                const int lx = o % 5, ly = (o / 5) % 5, lz = o / 25;
                if (lx < 4) {
                    data |= v;
                } else if (ly < 4 && lz < 4) {
                    unsigned long long x = ((unsigned long long)(i + k - base) << 8) | quantize_meta(v);
                    x *= 0x9e3779b97f4a7c15ull;
                    x ^= x >> 29;
                    x *= 0xbf58476d1ce4e5b9ull;
                    x ^= x >> 32;
                    hsh += x;
                }
            }
        }
    }
    if (data) {
        atomicOr(&s_data, 1u);
    }
    atomicAdd(&s_hash, hsh);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = make_uint4(s_data, 0u, (uint32_t)s_hash, (uint32_t)(s_hash >> 32));
    }
}

// Rounds the clearances of a chunk down in place (the chunk that others are mapped onto holds the quantised bytes).
__global__ void mbrick_chunk_quantize_kernel(uint8_t *__restrict__ chunk, int64_t chunk_bytes)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunk_bytes; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i & 127);
        const int lx = o % 5, ly = (o / 5) % 5, lz = o / 25;
        if (o < 125 && lx == 4 && ly < 4 && lz < 4) {
            chunk[i] = (uint8_t)quantize_meta(chunk[i]);
        }
    }
}

hipError_t launch_mbrick_chunk_class(const uint8_t *bricks, int64_t total_bytes, int64_t chunk_bytes, uint4 *out, hipStream_t stream)
{
    const int64_t chunks = (total_bytes + chunk_bytes - 1) / chunk_bytes;
    hipLaunchKernelGGL(mbrick_chunk_class_kernel, dim3((unsigned)chunks), dim3(256), 0, stream, bricks, total_bytes, chunk_bytes, out);
    return hipGetLastError();
}

hipError_t launch_mbrick_chunk_quantize(uint8_t *chunk, int64_t chunk_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL(mbrick_chunk_quantize_kernel, dim3(1024), dim3(256), 0, stream, chunk, chunk_bytes);
    return hipGetLastError();
}
#endif

// ---- sparse march bricks (DevScene::m_rows, m_coarse) ------------------------------------------------------
// Extent of every brick row: the bricks from the first to the last one with a non-zero texel byte (meta bytes, at
// lx == 4, do not count).  row_x0 starts at 0xffffffff, row_x1 at 0.
__global__ void mbrick_extent_kernel(const uint8_t *__restrict__ bricks, int gx, int gy, int gz, uint32_t *__restrict__ row_x0,
                                     uint32_t *__restrict__ row_x1)
{
    const int64_t total = (int64_t)gx * gy * gz;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < total; b += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t *p = bricks + ((size_t)b << 7);
        uint32_t any = 0;
        for (int o = 0; o < 125; o++) {
            any |= (o % 5 != 4) ? (uint32_t)p[o] : 0u;
        }
        if (any) {
            const uint32_t bx = (uint32_t)(b % gx), row = (uint32_t)(b / gx);
            atomicMin(&row_x0[row], bx);
            atomicMax(&row_x1[row], bx + 1u);
        }
    }
}

// One thread per 16 bytes of the dense array: bricks inside their row's extent move to their place in the compact array.
__global__ void mbrick_compact_kernel(const uint4 *__restrict__ dense, int gx, int64_t bricks_total, const uint2 *__restrict__ rows,
                                      uint4 *__restrict__ compact)
{
    const int64_t total = bricks_total * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i >> 3;
        const uint32_t bx = (uint32_t)(b % gx), row = (uint32_t)(b / gx);
        const uint2 ri = rows[row];
        const uint32_t rel = bx - (ri.y & 0xffffu);
        if (rel < (ri.y >> 16)) {
            compact[((size_t)(ri.x + rel) << 3) | (size_t)(i & 7)] = dense[i];
        }
    }
}

// One thread per coarse cell: the minimum clearance (dist - 1, as in build_mbricks_kernel) and the AND of the
// "interior" flags of the cell's base texels; base texels outside the volume are blocked and not interior.
__global__ void coarse_clearance_kernel(const uint8_t *__restrict__ dist, int nx, int ny, int nz, int bias, int cshift, int cgx,
                                        int cgy, int cgz, uint8_t *__restrict__ out)
{
    const int64_t total = (int64_t)cgx * cgy * cgz;
    const int C = 1 << cshift;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cx = (int)(i % cgx), cy = (int)((i / cgx) % cgy), cz = (int)(i / ((int64_t)cgx * cgy));
        const int x0 = cx * C - bias, y0 = cy * C - bias, z0 = cz * C - bias;
        int c = kClearMax;
        bool interior = true;
        for (int z = z0; z < z0 + C; z++) {
            for (int y = y0; y < y0 + C; y++) {
                for (int x = x0; x < x0 + C; x++) {
                    const bool in_grid = x >= 0 && y >= 0 && z >= 0 && x < nx && y < ny && z < nz;
                    const int d = in_grid ? (int)dist[((size_t)z * ny + y) * nx + x] : 0;
                    c = min(c, d - 1);
                    interior = interior && x >= 1 && y >= 1 && z >= 1 && x <= nx - 3 && y <= ny - 3 && z <= nz - 3;
                }
            }
        }
        out[i] = (uint8_t)(max(c, 0) | (interior ? 0x80 : 0));
    }
}

hipError_t launch_mbrick_extent(const uint8_t *bricks, int gx, int gy, int gz, uint32_t *row_x0, uint32_t *row_x1, hipStream_t stream)
{
    const int64_t total = (int64_t)gx * gy * gz;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 65536 * 4);
    hipLaunchKernelGGL(mbrick_extent_kernel, dim3(blocks), dim3(256), 0, stream, bricks, gx, gy, gz, row_x0, row_x1);
    return hipGetLastError();
}

hipError_t launch_mbrick_compact(const uint8_t *dense, int gx, int gy, int gz, const uint2 *rows, uint8_t *compact, hipStream_t stream)
{
    const int64_t total = (int64_t)gx * gy * gz;
    const int blocks = (int)std::min<int64_t>((total * 8 + 255) / 256, 65536 * 8);
    hipLaunchKernelGGL(mbrick_compact_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4 *)dense, gx, total, rows, (uint4 *)compact);
    return hipGetLastError();
}

hipError_t launch_coarse_clearance(const uint8_t *dist, int nx, int ny, int nz, int bias, int cshift, int cgx, int cgy, int cgz,
                                   uint8_t *out, hipStream_t stream)
{
    const int64_t total = (int64_t)cgx * cgy * cgz;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 65536 * 4);
    hipLaunchKernelGGL(coarse_clearance_kernel, dim3(blocks), dim3(256), 0, stream, dist, nx, ny, nz, bias, cshift, cgx, cgy, cgz, out);
    return hipGetLastError();
}

// =============================================================================================
// free-space distance field over kBrick^3-texel bricks (see DevScene::dist)
// =============================================================================================
constexpr int kDistMax = 64;

// Majorant of every brick (see DevScene) and the seed of the distance transform: dist = kDistMax
// for free bricks (majorant 0 and texel range inside [1, N-3] on every axis, which implies isInBox
// for every position whose base texel is in the brick), 0 otherwise.
__global__ void brick_free_kernel(const uint8_t *__restrict__ t, int nx, int ny, int nz, int bias, int gx, int gy,
                                  int gz, uint8_t *__restrict__ dist, uint8_t *__restrict__ majorant)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= gx * gy * gz) {
        return;
    }
    const int bx = b % gx, by = (b / gx) % gy, bz = b / (gx * gy);
    const int x0 = bx * kBrick - bias, y0 = by * kBrick - bias, z0 = bz * kBrick - bias;
    const bool interior = x0 >= 1 && y0 >= 1 && z0 >= 1 && x0 + kBrick - 1 <= nx - 3 && y0 + kBrick - 1 <= ny - 3 &&
                          z0 + kBrick - 1 <= nz - 3;
    int m = 0;
    for (int z = z0 - 1; z <= z0 + kBrick + 1; z++) {
        const int zc = min(max(z, 0), nz - 1);
        for (int y = y0 - 1; y <= y0 + kBrick + 1; y++) {
            const uint8_t *row = t + ((size_t)zc * ny + min(max(y, 0), ny - 1)) * nx;
            for (int x = x0 - 1; x <= x0 + kBrick + 1; x++) {
                m = max(m, (int)row[min(max(x, 0), nx - 1)]);
            }
        }
    }
    majorant[b] = (uint8_t)m;
    dist[b] = (interior && m == 0) ? (uint8_t)kDistMax : (uint8_t)0;
}

// Writes the meta byte (see DevScene) into byte 125 of every density brick.
__global__ void brick_meta_kernel(const uint8_t *__restrict__ dist, const uint8_t *__restrict__ majorant, int nx,
                                  int ny, int nz, int bias, int gx, int gy, int gz, uint8_t *__restrict__ bricks)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= gx * gy * gz) {
        return;
    }
    const int bx = b % gx, by = (b / gx) % gy, bz = b / (gx * gy);
    const int x0 = bx * kBrick - bias, y0 = by * kBrick - bias, z0 = bz * kBrick - bias;
    const bool interior = x0 >= 1 && y0 >= 1 && z0 >= 1 && x0 + kBrick - 1 <= nx - 3 && y0 + kBrick - 1 <= ny - 3 &&
                          z0 + kBrick - 1 <= nz - 3;
    bricks[((size_t)b << 7) + 125] = (uint8_t)(min((int)dist[b], 127) | (interior ? 0x80 : 0));
    bricks[((size_t)b << 7) + 126] = majorant[b];
}

// One relaxation of the Chebyshev distance transform: d = min(d, 1 + min over the 26 neighbours).
__global__ void brick_dist_relax_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int gx, int gy,
                                        int gz)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= gx * gy * gz) {
        return;
    }
    const int bx = b % gx, by = (b / gx) % gy, bz = b / (gx * gy);
    int m = 255;
    for (int dz = -1; dz <= 1; dz++) {
        for (int dy = -1; dy <= 1; dy++) {
            for (int dx = -1; dx <= 1; dx++) {
                const int x = bx + dx, y = by + dy, z = bz + dz;
                const int v = (x < 0 || y < 0 || z < 0 || x >= gx || y >= gy || z >= gz)
                                  ? 0
                                  : (int)in[((size_t)z * gy + y) * gx + x];
                m = min(m, v);
            }
        }
    }
    out[b] = (uint8_t)min((int)in[b], m + 1);
}

hipError_t launch_build_dist(const uint8_t *texels, int nx, int ny, int nz, int bias, int gx, int gy, int gz,
                             uint8_t *dist, uint8_t *scratch, uint8_t *majorant, hipStream_t stream)
{
    const int total = gx * gy * gz;
    const int threads = 256, blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(brick_free_kernel, dim3(blocks), dim3(threads), 0, stream, texels, nx, ny, nz, bias, gx, gy,
                       gz, dist, majorant);
    // kDistMax relaxations make every value exact up to the cap; ping-pong, ending in `dist`
    uint8_t *a = dist, *b = scratch;
    for (int i = 0; i < kDistMax; i++) {
        hipLaunchKernelGGL(brick_dist_relax_kernel, dim3(blocks), dim3(threads), 0, stream, a, b, gx, gy, gz);
        uint8_t *tmp = a;
        a = b;
        b = tmp;
    }
    if (a != dist) {
        hipError_t e = hipMemcpyAsync(dist, a, (size_t)total, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) {
            return e;
        }
    }
    return hipGetLastError();
}

hipError_t launch_brick_meta(const uint8_t *dist, const uint8_t *majorant, int nx, int ny, int nz, int bias, int gx,
                             int gy, int gz, uint8_t *bricks, hipStream_t stream)
{
    const int total = gx * gy * gz;
    hipLaunchKernelGGL(brick_meta_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, dist, majorant, nx, ny, nz,
                       bias, gx, gy, gz, bricks);
    return hipGetLastError();
}

// =============================================================================================
// the estimator
// =============================================================================================
// LDS holds what the CDF inversion reads ~5 times per scatter; the two phase tables are read
// once per scatter and stay in global memory (L1/L2 resident, 16 KiB each).
struct MieLds {
    float cdf[kMieN];
    uint16_t guide[kGuideN + 2];
};

// The MARCH kernel also keeps the chopped phase table there (NEE reads two neighbouring entries of it at
// every bounce but the first): 40 KiB per block, which 512-thread blocks make affordable (3 per CU, the
// same 6 waves per SIMD) -- one divergent gather per scatter less on an L1 that is busy 77 % of all cycles.
struct MieLdsFull {
    float cdf[kMieN];
    float chopped[kMieN];
    uint16_t guide[kGuideN + 2];
};

CT_DEV void load_tables(const DevScene &sc, MieLdsFull &lds)
{
    for (int i = threadIdx.x; i < kMieN; i += blockDim.x) {
        lds.cdf[i] = sc.cdf[i];
        lds.chopped[i] = sc.chopped[i];
    }
    for (int i = threadIdx.x; i < kGuideN + 2; i += blockDim.x) {
        lds.guide[i] = sc.guide[i];
    }
    __syncthreads();
}

CT_DEV void load_tables(const DevScene &sc, MieLds &lds)
{
    for (int i = threadIdx.x; i < kMieN; i += blockDim.x) {
        lds.cdf[i] = sc.cdf[i];
    }
    for (int i = threadIdx.x; i < kGuideN + 2; i += blockDim.x) {
        lds.guide[i] = sc.guide[i];
    }
    __syncthreads();
}

// Primary ray of pixel (x,y): trace<>, cameraCommon.cuh:18-30.
CT_DEV f3 primary_direction(const DevScene &sc, uint32_t x, uint32_t y)
{
    const float dx = (float)x / (float)sc.width * 2.f - 1.f;
    const float dy = (float)y / (float)sc.height * 2.f - 1.f;
    const f3 U = mk3(sc.ux, sc.uy, sc.uz), V = mk3(sc.vx, sc.vy, sc.vz), W = mk3(sc.wx, sc.wy, sc.wz);
    return normalize3(add3(add3(scale3(U, dx), scale3(V, dy)), W));
}

// The primary ray of a pixel does not depend on the subframe (no jitter: cameraCommon.cuh:22),
// so pinholeCamera + intersect + the closest-hit prologue (cloudRadianceMaterials.cu:11-17) are
// evaluated once per camera pose: primary[2p] = (entry point in box coordinates, hit ? 1 : 0),
// primary[2p+1] = (normalize(ray.direction), bits of the seed base x*4096+y).
__global__ __launch_bounds__(256) void primary_rays_kernel(DevScene sc, float4 *__restrict__ primary)
{
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= sc.width || y >= sc.height) {
        return;
    }
    const f3 eye = mk3(sc.ex, sc.ey, sc.ez);
    const f3 d = primary_direction(sc, x, y);
    float t_hit = 0;
    const bool hit = intersect_box(sc, eye, d, t_hit);
    f3 pos = add3(eye, scale3(d, t_hit));                       // :11
    pos = add3(pos, scale3(mk3(sc.bx, sc.by, sc.bz), 0.5f));    // :12
    const f3 dir = normalize3(d);                               // :17
    const size_t p = (size_t)y * sc.width + x;
    primary[2 * p] = make_float4(pos.x, pos.y, pos.z, hit ? 1.f : 0.f);
    primary[2 * p + 1] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(x * 4096u + y)); // :21
}

hipError_t launch_primary_rays(const DevScene &sc, float4 *primary, hipStream_t stream)
{
    const dim3 grid((sc.width + 31) / 32, (sc.height + 7) / 8), block(256);
    hipLaunchKernelGGL(primary_rays_kernel, grid, block, 0, stream, sc, primary);
    return hipGetLastError();
}

// One byte per pixel: does the primary ray hit the box?  (What the host needs of the 32 B per pixel of `primary` to build
// the pixel list of a new pose; fetching the rays themselves was 32 MB over PCIe into pageable memory, 10 of the 13 ms.)
__global__ __launch_bounds__(256) void hit_flags_kernel(const float4 *__restrict__ primary, uint8_t *__restrict__ flags, uint32_t pixels)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < pixels) {
        flags[p] = primary[2 * (size_t)p].w != 0.f ? 1u : 0u;
    }
}

// The cost plane of a pose's first launch -> per pixel group: sum of the path costs, deepest path.  One block per group of the
// launch's chunk (its 64 columns x S rows); group_order == NULL: column c * 64 is group c.
__global__ __launch_bounds__(64) void cost_reduce_kernel(const uint2 *__restrict__ plane, uint32_t frame_stride, uint32_t S,
                                                         const uint32_t *__restrict__ group_order, uint32_t rank_base,
                                                         uint32_t *__restrict__ cost_sum, uint32_t *__restrict__ cost_deepest)
{
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    uint32_t sum = 0, deepest = 0;
    for (uint32_t s = 0; s < S; s++) {
        const uint2 v = plane[(size_t)s * frame_stride + c * 64u + lane];
        sum += v.x;
        deepest = max(deepest, v.y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        deepest = max(deepest, (uint32_t)__shfl_xor((int)deepest, off));
    }
    if (lane == 0) {
        const uint32_t g = group_order ? group_order[rank_base + c] : c;
        cost_sum[g] += sum;
        cost_deepest[g] = max(cost_deepest[g], deepest);
    }
}

hipError_t launch_cost_reduce(const uint2 *cost_plane, uint32_t frame_stride, uint32_t S, uint32_t n_groups_in_chunk,
                              const uint32_t *group_order, uint32_t rank_base, uint32_t *cost_sum, uint32_t *cost_deepest,
                              hipStream_t stream)
{
    if (n_groups_in_chunk != 0u && S != 0u) {
        hipLaunchKernelGGL(cost_reduce_kernel, dim3(n_groups_in_chunk), dim3(64), 0, stream, cost_plane, frame_stride, S, group_order,
                           rank_base, cost_sum, cost_deepest);
    }
    return hipGetLastError();
}

hipError_t launch_hit_flags(const float4 *primary, uint8_t *flags, uint32_t pixels, hipStream_t stream)
{
    hipLaunchKernelGGL(hit_flags_kernel, dim3((pixels + 255u) / 256u), dim3(256), 0, stream, primary, flags, pixels);
    return hipGetLastError();
}

// estimateEmission's ray setup (pointEmissionCamera.cu:22-31): the same closest-hit prologue for
// an arbitrary origin/direction; launchID is 1-D, so the seed base is i*4096 + 0.
struct PointTask {
    int32_t id;
    uint32_t experimentCount;
    float radiance, runningVariance;
    float px, py, pz, dx, dy, dz;
};

__global__ __launch_bounds__(256) void point_rays_kernel(DevScene sc, const PointTask *__restrict__ tasks, uint32_t n,
                                                         uint32_t n_pad, float4 *__restrict__ primary,
                                                         uint32_t *__restrict__ pixels)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) {
        return;
    }
    if (i >= n) {
        pixels[i] = 0xffffffffu;
        return;
    }
    pixels[i] = i;
    const PointTask t = tasks[i];
    const f3 o = mk3(t.px, t.py, t.pz), d = mk3(t.dx, t.dy, t.dz);
    float t_hit = 0;
    const bool hit = intersect_box(sc, o, d, t_hit);
    f3 pos = add3(o, scale3(d, t_hit));
    pos = add3(pos, scale3(mk3(sc.bx, sc.by, sc.bz), 0.5f));
    const f3 dir = normalize3(d);
    primary[2 * (size_t)i] = make_float4(pos.x, pos.y, pos.z, hit ? 1.f : 0.f);
    primary[2 * (size_t)i + 1] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(i * 4096u));
}

// PointRadianceTask::addExperimentResult (PointRadianceTask.h:40-51) for `launches` results in order.
__global__ __launch_bounds__(256) void point_accumulate_kernel(const float4 *__restrict__ frames, uint32_t stride,
                                                               PointTask *__restrict__ tasks, uint32_t n,
                                                               uint32_t launches)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    PointTask t = tasks[i];
    for (uint32_t f = 0; f < launches; f++) {
        const float new_radiance = frames[(size_t)f * stride + i].x;
        t.experimentCount++;
        const float N = (float)t.experimentCount;
        const float new_weight = (float)(1.0 / (double)N);
        const float previous_mean = t.radiance;
        const float new_mean = t.radiance + (new_radiance - previous_mean) * new_weight;
        t.radiance = new_mean;
        t.runningVariance += (new_radiance - previous_mean) * (new_radiance - new_mean);
    }
    tasks[i] = t;
}

hipError_t launch_point_rays(const DevScene &sc, const void *tasks, uint32_t n, uint32_t n_pad, float4 *primary,
                             uint32_t *pixels, hipStream_t stream)
{
    hipLaunchKernelGGL(point_rays_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, stream, sc,
                       (const PointTask *)tasks, n, n_pad, primary, pixels);
    return hipGetLastError();
}

hipError_t launch_point_accumulate(const float4 *frames, uint32_t stride, void *tasks, uint32_t n, uint32_t launches,
                                   hipStream_t stream)
{
    hipLaunchKernelGGL(point_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, frames, stride,
                       (PointTask *)tasks, n, launches);
    return hipGetLastError();
}

// frameResultBuffer before a single-subframe render: (0,0,0,1) for this shard's pixels (misses
// keep that value: empty miss program, progressive.cu:44-46), (0,0,0,0) for foreign pixels.
__global__ void fill_frame_kernel(float4 *__restrict__ frame, uint32_t width, uint32_t height, uint32_t shard_index,
                                  uint32_t shard_count)
{
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= width || y >= height) {
        return;
    }
    const bool own = tile_owner(x / kTile, y / kTile, shard_count) == shard_index;
    frame[(size_t)y * width + x] = make_float4(0.f, 0.f, 0.f, own ? 1.f : 0.f);
}

__global__ __launch_bounds__(64) void zero_words_kernel(ZeroList z)
{
    for (int k = 0; k < z.n; k++) {
        for (uint32_t i = threadIdx.x; i < z.words[k]; i += 64u) {
            z.ptr[k][i] = 0u;
        }
    }
}

hipError_t launch_zero_words(const ZeroList &z, hipStream_t stream)
{
    if (z.n > 0) {
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, stream, z);
    }
    return hipGetLastError();
}

hipError_t launch_fill_frame(float4 *frame, uint32_t width, uint32_t height, uint32_t shard_index,
                             uint32_t shard_count, hipStream_t stream)
{
    const dim3 grid((width + 31) / 32, (height + 7) / 8), block(256);
    hipLaunchKernelGGL(fill_frame_kernel, grid, block, 0, stream, frame, width, height, shard_index, shard_count);
    return hipGetLastError();
}

// NEE: getInScattering, cloud.cuh:146-158.
CT_DEV f3 in_scattering(const DevScene &sc, f3 pos, f3 dir, bool chopped)
{
    const float cos_light = dot3(mk3(sc.nlx, sc.nly, sc.nlz), dir);
    const float u = (cos_light + 1) / 2;
    const float *table = chopped ? sc.chopped : sc.mie;
    const float phase = tex1(table, u);
    const float ins = tex3_apron(sc, sc.ibricks, pos);
    f3 l = scale3(mk3(sc.lr, sc.lg, sc.lb), ins);
    l = scale3(l, phase);
    return scale3(l, sc.sun_ratio);
}

// The same NEE in two halves, so that its three loads (two phase-table entries, the shadow
// volume's footprint) are in flight while the direction sampling runs: identical arithmetic.
struct NeeLoads {
    float a, b, w;     // phase table entries and filter weight
    uint2 cell;        // shadow-volume footprint
};

CT_DEV NeeLoads in_scattering_issue(const DevScene &sc, f3 pos, f3 dir, bool chopped)
{
    NeeLoads n;
    const float cos_light = dot3(mk3(sc.nlx, sc.nly, sc.nlz), dir);
    const float u = (cos_light + 1) / 2;
    const float *table = chopped ? sc.chopped : sc.mie;
    const float x = fmaf(u, (float)kMieN, -0.5f);
    const int32_t i = (int32_t)floorf(x);
    // entries clamp(i) and clamp(i+1) with one 8-byte gather: they are neighbours unless i is -1 or 4095
    float2 pair;
    __builtin_memcpy(&pair, table + min(max(i, 0), kMieN - 2), sizeof pair);
    n.a = (i > kMieN - 2) ? pair.y : pair.x;
    n.b = (i < 0) ? pair.x : pair.y;
    n.w = fract_(x);
    uint32_t meta_unused;
    n.cell = fetch_cell(sc, sc.ibricks, pos, meta_unused);
    return n;
}

// Only the phase-table half of in_scattering_issue (the caller has the shadow-volume footprint already).
CT_DEV NeeLoads in_scattering_issue_phase(const DevScene &sc, f3 dir, bool chopped)
{
    NeeLoads n;
    const float cos_light = dot3(mk3(sc.nlx, sc.nly, sc.nlz), dir);
    const float u = (cos_light + 1) / 2;
    const float *table = chopped ? sc.chopped : sc.mie;
    const float x = fmaf(u, (float)kMieN, -0.5f);
    const int32_t i = (int32_t)floorf(x);
    float2 pair;
    __builtin_memcpy(&pair, table + min(max(i, 0), kMieN - 2), sizeof pair);
    n.a = (i > kMieN - 2) ? pair.y : pair.x;
    n.b = (i < 0) ? pair.x : pair.y;
    n.w = fract_(x);
    n.cell = make_uint2(0u, 0u);
    return n;
}

// The same with the chopped table in LDS and the lane's one-entry footprint cache (MARCH kernel; in the DELTA kernel,
// which is bound by instruction issue at its register limit, the cache cost 2 %); the un-chopped table (first bounce)
// stays global.
CT_DEV NeeLoads in_scattering_issue_lds(const DevScene &sc, const float *lds_chopped, f3 pos, f3 dir, bool chopped,
                                        uint32_t &nee_key, uint2 &nee_cell, bool &reused)
{
    NeeLoads n;
    const float cos_light = dot3(mk3(sc.nlx, sc.nly, sc.nlz), dir);
    const float u = (cos_light + 1) / 2;
    const float x = fmaf(u, (float)kMieN, -0.5f);
    const int32_t i = (int32_t)floorf(x);
    const int32_t j = min(max(i, 0), kMieN - 2);
    float2 pair;
    if (chopped) {
        pair = make_float2(lds_chopped[j], lds_chopped[j + 1]);
    } else {
        __builtin_memcpy(&pair, sc.mie + j, sizeof pair);
    }
    n.a = (i > kMieN - 2) ? pair.y : pair.x;
    n.b = (i < 0) ? pair.x : pair.y;
    n.w = fract_(x);
    n.cell = fetch_cell_cached(sc, sc.ibricks, pos, nee_key, nee_cell, reused);
    return n;
}

CT_DEV f3 in_scattering_finish(const DevScene &sc, const NeeLoads &n, f3 pos)
{
    const float phase = fmaf(n.w, n.b - n.a, n.a);
    const float ins = filter_at(sc, n.cell, pos);
    f3 l = scale3(mk3(sc.lr, sc.lg, sc.lb), ins);
    l = scale3(l, phase);
    return scale3(l, sc.sun_ratio);
}

enum : int { ST_IDLE = 0, ST_MARCH = 1, ST_BOUNCE = 2 };

// A scatter phase is followed by the march burst in the SAME scheduler iteration -- the lanes it has just redirected march at
// once -- instead of a pass through the loop's top in between: the scheduler's own instructions (idle / marching / bouncing
// ballots, the job and drain tests, the queue-empty hint) run once per pair of phases.  Round 4, 512^3 / 1024^2: 3297 -> 3349
// Msamples/s (+1.6 %; DELTA, where the scheduler is a larger share: 5120 -> 5273, +3.0 %; profiles/r04i).  Schedule only.
// -DCT_MARCH_FUSE=0 / -DCT_DELTA_FUSE=0 (build --variant nofuse): separate iterations, as until round 3.
#ifndef CT_MARCH_FLAT_ZERO
#define CT_MARCH_FLAT_ZERO 1   // (1: an all-zero footprint goes through the density evaluation like the others)
#endif
#ifndef CT_MARCH_FLAT_EXIT
#define CT_MARCH_FLAT_EXIT 1
#endif
#ifndef CT_MARCH_FUSE
#define CT_MARCH_FUSE 1
#endif
constexpr bool MARCH_FUSE = CT_MARCH_FUSE != 0;

// 1 / max per-axis advance of one march step, in texels (approximate reciprocal is fine: it only
// sizes a conservative skip, see the free-space skip in the march phase).
CT_DEV float inv_max_advance(const DevScene &sc, f3 stepv)
{
    const float m = fmaxf(fmaxf(fabsf(stepv.x) * sc.sx, fabsf(stepv.y) * sc.sy), fabsf(stepv.z) * sc.sz);
    return __builtin_amdgcn_rcpf(fmaxf(m, 1e-20f));
}


// Free-space skip of the march (see render_persistent_kernel): how many steps the clearance `c`
// (texels) of the row the last fetch was based in allows, and their replay.  The adds are the work;
// a wave-level branch per add is not: blocks of 8, then singles (measured: 1815 -> 1891 Msamples/s;
// 16/4/1 is no better).
CT_DEV int skip_steps(uint32_t c, float inv_maxd)
{
    return (int)(((float)c - 0.03125f) * inv_maxd);
}

CT_DEV void replay_steps(f3 &pos, f3 stepv, int n)
{
    int i = 0;
    for (; i + 8 <= n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            pos = add3(pos, stepv);
        }
    }
    for (; i < n; i++) {
        pos = add3(pos, stepv);
    }
}

// =============================================================================================
// shadow volume  (inScatter.cu:40-66): sample, THEN step; <= 1/sampleStep steps; early out
// when T*255 < 1; uchar(T*255) truncating.  One thread per texel.
//
// Nine samples in ten of a march towards the sun are taken in empty space, where the footprint is eight zeros, the
// density 0 and exp(-0) = 1 exactly: the transmittance -- and with it the early-out test -- stays as it was and only the
// position moves on.  So the walk is the estimator's (primary_advance_kernel): inside the box it reads the march bricks,
// whose row clearances say how many of the next samples are such no-ops, and replays their position adds (the same float
// adds, so the next real sample is taken at the same bits); once the position has left the volume through a face whose
// texel layer is all zero and keeps moving away from it, every remaining footprint is that layer's (clamp to edge) and
// the walk ends; only a volume that is not empty at the face it is left through takes the remaining samples one by one
// from the clamped sampler.  167 -> 24 ms at 512^3, 21 -> 4.7 ms at 256^3 (tools/gpu_create_profile.sh, profiles/r02r).
// zero_faces: bit 0/1 = the texel layers x = 0 / x = nx-1 are all zero, bits 2/3 the same for y, 4/5 for z.
// =============================================================================================
__global__ __launch_bounds__(256) void inscatter_kernel(DevScene sc, uint8_t *__restrict__ out, uint32_t zero_faces)
{
    const int64_t total = (int64_t)sc.nx * sc.ny * sc.nz;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) {
        return;
    }
    const int x = (int)(i % sc.nx), y = (int)((i / sc.nx) % sc.ny), z = (int)(i / ((int64_t)sc.nx * sc.ny));
    const float max_size = (float)max(max(sc.nx, sc.ny), sc.nz);
    // textureScale = sx / nx etc. was folded on the host; min(textureScale) is passed in sun_ratio's
    // neighbour: recompute it exactly as the host did (float division, same operands).
    const float tsx = max_size / (float)sc.nx, tsy = max_size / (float)sc.ny, tsz = max_size / (float)sc.nz;
    const float min_scale = fminf(fminf(tsx, tsy), tsz);
    const float inv_min_scale = 1.0f / min_scale;
    f3 p = mk3((float)x / max_size, (float)y / max_size, (float)z / max_size);
    p = scale3(p, inv_min_scale);
    // stepToLight = (-normalize(lightDirection)) * sampleStep; (nlx,nly,nlz) = -lightDirection and
    // normalize(-v) = -normalize(v) exactly (sign symmetry of every operation involved).
    const f3 step_to_light = scale3(normalize3(mk3(sc.nlx, sc.nly, sc.nlz)), sc.sample_step);
    const int step_count = (int)(1 / sc.sample_step);
    float transmittance = 1;
    int s = 0;
    bool done = false;
    if (sc.mbricks != nullptr && sc.m_rows == nullptr) {
        const float inv_maxd = inv_max_advance(sc, step_to_light);
        while (s < step_count && in_box(sc, p)) {
            uint32_t meta;
            const uint2 cell = fetch_cell_m<false>(sc, p, meta);
            if ((cell.x | cell.y) != 0u) {
                const float density = filter_at(sc, cell, p) * sc.density_multiplier;
                const float extinction = density * sc.sample_step;
                transmittance *= ct_expf(-extinction);
                if (transmittance * 255.f < 1.f) {
                    done = true;
                    break;
                }
            }
            p = add3(p, step_to_light);
            s += 1;
            const uint32_t clear = meta & 0x7fu;
            if (clear != 0u) {
                const int n = min(skip_steps(clear, inv_maxd), step_count - s);
                replay_steps(p, step_to_light, n);
                s += n;
            }
        }
    }
    for (; !done && s < step_count; s++) {
        // texel coordinates; an axis that is past its first / last texel centre and moving on reads that layer only
        const float tx = fmaf(p.x, sc.sx, -0.5f), ty = fmaf(p.y, sc.sy, -0.5f), tz = fmaf(p.z, sc.sz, -0.5f);
        const bool gone = ((zero_faces & 1u) && tx < 0.0f && step_to_light.x <= 0.0f) ||
                          ((zero_faces & 2u) && tx >= (float)(sc.nx - 1) && step_to_light.x >= 0.0f) ||
                          ((zero_faces & 4u) && ty < 0.0f && step_to_light.y <= 0.0f) ||
                          ((zero_faces & 8u) && ty >= (float)(sc.ny - 1) && step_to_light.y >= 0.0f) ||
                          ((zero_faces & 16u) && tz < 0.0f && step_to_light.z <= 0.0f) ||
                          ((zero_faces & 32u) && tz >= (float)(sc.nz - 1) && step_to_light.z >= 0.0f);
        if (gone) {
            break;
        }
        const float density = tex3_clamped(sc, sc.dbricks, p) * sc.density_multiplier;
        const float extinction = density * sc.sample_step;
        transmittance *= ct_expf(-extinction);
        p = add3(p, step_to_light);
        if (transmittance * 255.f < 1.f) {
            break;
        }
    }
    out[i] = (uint8_t)(transmittance * 255.f);
}

hipError_t launch_inscatter(const DevScene &sc, uint8_t *out, uint32_t zero_faces, hipStream_t stream)
{
    const int64_t total = (int64_t)sc.nx * sc.ny * sc.nz;
    const int threads = 256;
    const int64_t blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(inscatter_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, sc, out, zero_faces);
    return hipGetLastError();
}

// The XCD this wave runs on (XCC_ID, hardware register 20, bits 3:0).
CT_DEV uint32_t xcd_id()
{
    return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (uint32_t)(kQueues - 1);
}

// A job the previous launch handed on (BatchArgs::left_in).  Wave-uniform.
struct JobState {
    uint32_t g, next, end;       // pixel group; samples [next, end) still to start: sample q is lane q & 63 of the job's subframe q >> 6
    uint32_t base;               // scratch index of the job's first subframe, lane 0 (absolute: region, row and the group's column)
    uint32_t first;              // subframe id of the job's first subframe
    uint32_t age;                // the age its samples start with
};   // (six SGPRs that live through the whole scheduler loop; the scratch's row stride is the same for every job in flight)

// Where group g's 64 results of a subframe go within a subframe's row of the scratch (BatchArgs::group_rank).
CT_DEV uint32_t group_column(const BatchArgs &ba, uint32_t g)
{
    return (ba.group_rank ? __builtin_amdgcn_readfirstlane(ba.group_rank[g]) - ba.rank_base : g) * 64u;
}

CT_DEV bool take_leftover(const BatchArgs &ba, uint32_t lane, bool &left_done, JobState &job)
{
    if (!ba.left_in || left_done) {
        return false;
    }
    uint32_t i = 0;
    if (lane == 0) {
        i = atomicAdd(ba.left_cursor, 1u);
    }
    i = __builtin_amdgcn_readfirstlane(i);
    if (i >= __builtin_amdgcn_readfirstlane(*ba.left_in_count)) {
        left_done = true;
        return false;
    }
    const uint32_t *r = ba.left_in + (size_t)i * kLeftWords;
    job.g = __builtin_amdgcn_readfirstlane(r[0]);
    job.next = __builtin_amdgcn_readfirstlane(r[1]);
    job.end = __builtin_amdgcn_readfirstlane(r[2]);
    job.base = __builtin_amdgcn_readfirstlane(r[3]);
    job.first = __builtin_amdgcn_readfirstlane(r[4]);
    job.age = __builtin_amdgcn_readfirstlane(r[5]);
    return true;
}

// The rest of this wave's job goes to the next launch (see BatchArgs::left_out).  Returns false if there is no room.
CT_DEV bool hand_on_job(const BatchArgs &ba, uint32_t lane, const JobState &job)
{
    uint32_t i = 0;
    if (lane == 0) {
        i = atomicAdd(ba.left_out_count, 1u);
        if (i >= ba.left_capacity) {
            atomicSub(ba.left_out_count, 1u);
            i = 0xffffffffu;
        } else {
            uint4 *r = (uint4 *)(ba.left_out + (size_t)i * kLeftWords);
            r[0] = make_uint4(job.g, job.next, job.end, job.base);
            r[1] = make_uint4(job.first, job.age + 1u, 0u, 0u);
        }
    }
    return __builtin_amdgcn_readfirstlane(i) != 0xffffffffu;
}

// Next job for this wave: from the queue it is working on (first the shared one), else from its
// XCD's, else from the following ones.  Wave-uniform.  Returns false when every queue is empty.
CT_DEV bool take_job(const BatchArgs &ba, uint32_t lane, uint32_t &q_cur, uint32_t &q_tried, uint32_t &job)
{
    while (q_tried < (uint32_t)kQueues) {
        const uint32_t begin = ba.q_begin[q_cur], end = ba.q_begin[q_cur + 1];
        if (begin != end) {
            uint32_t j = 0;
            if (lane == 0) {
                j = atomicAdd(&ba.queue[q_cur], 1u);
            }
            j = __builtin_amdgcn_readfirstlane(j);
            if (j < end - begin) {
                job = ba.reverse ? end - 1u - j : begin + j;
                if (j + 1u == end - begin && lane == 0) {
                    // The last job of THIS queue.  The list is empty when that has happened to every queue that had jobs; whoever
                    // finds it so raises the flag that the other waves look at now and then (see the suspend logic).  (Until round
                    // 3 the flag went up with the job of the highest index, which is "the list is empty" for one queue only: with
                    // per-XCD queues the waves did not look at it, and a busy wave learnt that nothing was left only when 16 of
                    // its lanes had run out of work -- a launch of 10 subframes drained for 0.9 of its 4.3 ms.)
                    uint32_t with_jobs = 0;
                    for (uint32_t x = 0; x <= (uint32_t)kQueues; x++) {
                        with_jobs += (ba.q_begin[x] != ba.q_begin[x + 1u]) ? 1u : 0u;
                    }
                    if (atomicAdd(&ba.queue[kQueueDone], 1u) + 1u == with_jobs) {
                        __atomic_store_n(ba.queue + kQueueFlag, 1u, __ATOMIC_RELAXED);
                    }
                }
                return true;
            }
        }
        if (q_cur == (uint32_t)kQueues) {
            q_cur = xcd_id();
        } else {
            q_tried += 1;
            q_cur = (q_cur + 1u) & (uint32_t)(kQueues - 1);
        }
    }
    return false;
}

CT_DEV uint32_t lane_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// No jitter also means that every sample of a pixel marches the same way until its primary ray
// meets the first non-zero footprint: T stays 1 and xi < 1 cannot collide.  That prefix -- mostly the
// long free-space hops from the box face to the cloud -- is walked once per pose here, with exactly
// the march phase's operations, and a sample starts at the state recorded BEFORE the decisive
// iteration (the one whose fetch is non-zero or leaves the box), which it then executes itself:
// advance[p] = (position, bits: steps done so far | clearance << 24).
template <bool SPARSE>
__global__ __launch_bounds__(256) void primary_advance_kernel(DevScene sc, const float4 *__restrict__ primary,
                                                              float4 *__restrict__ advance)
{
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= sc.width || y >= sc.height) {
        return;
    }
    const size_t p = (size_t)y * sc.width + x;
    const float4 p0 = primary[2 * p], p1 = primary[2 * p + 1];
    f3 pos = mk3(p0.x, p0.y, p0.z);
    const f3 dir = mk3(p1.x, p1.y, p1.z);
    f3 start = pos;
    uint32_t start_bits = 0;
    if (p0.w != 0.f && in_box(sc, pos)) {
        const f3 stepv = scale3(dir, sc.sample_step);
        const float inv_maxd = inv_max_advance(sc, stepv);
        uint32_t dfree = 0, steps = 0;
        for (int it = 0; it < 65536 && steps < 0x00f00000u; it++) {
            start = pos;
            start_bits = steps | (dfree << 24);
            if (dfree != 0u) {
                const int n = skip_steps(dfree, inv_maxd);
                replay_steps(pos, stepv, n);
                steps += (uint32_t)n;
            }
            pos = add3(pos, stepv);
            steps += 1u;
            uint32_t meta;
            const uint2 cell = fetch_cell_m<SPARSE>(sc, pos, meta);
            dfree = meta & 0x7fu;
            if ((cell.x | cell.y) != 0u || ((meta & 0x80u) == 0u && !in_box(sc, pos))) {
                break;
            }
        }
    }
    advance[p] = make_float4(start.x, start.y, start.z, __uint_as_float(start_bits));
}

hipError_t launch_primary_advance(const DevScene &sc, const float4 *primary, float4 *advance, hipStream_t stream)
{
    const dim3 grid((sc.width + 31) / 32, (sc.height + 7) / 8), block(256);
    if (sc.m_rows) {
        hipLaunchKernelGGL(primary_advance_kernel<true>, grid, block, 0, stream, sc, primary, advance);
    } else {
        hipLaunchKernelGGL(primary_advance_kernel<false>, grid, block, 0, stream, sc, primary, advance);
    }
    return hipGetLastError();
}

// Persistent wave-scheduled estimator.  Every lane owns one path at a time; a wave repeatedly
// picks the phase (regenerate / march one step / scatter) that the most lanes are waiting
// for, so lanes marching through empty space do not hold back lanes that collide every step
// and finished paths are replaced immediately from a global sample queue.  Paths are
// independent and seeded by (pixel, subframe) only, so the schedule cannot change a result.
//
// The queue hands out jobs = (pixel group, subframe range); a group is 64 consecutive entries of
// the list of this shard's box-hitting pixels (tile-Morton order).  See BatchArgs for the order.
// (6 waves per SIMD = 3 blocks of 512 threads per CU: the register allocator must stay within 80 VGPRs.  It
// uses 75 today; an edit of take_job once moved it to 85 and cost a third of the occupancy, hence the bound.)
// COST = false: a launch that does not record the paths' costs (BatchArgs::cost is null: every launch but the one that measures
// them) runs the kernel compiled without the per-path work counter.  Instantiated for the dense, non-diagnostic kernel.
template <int MODE, bool STATS, bool SPARSE, bool COST = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6))) void render_persistent_kernel(DevScene sc, BatchArgs ba)
{
    __shared__ MieLdsFull lds;
    load_tables(sc, lds);

    const uint32_t lane = threadIdx.x & 63u;
#ifdef CT_DIAG_TIMELINE   // (a build of its own, -DCT_DIAG_TIMELINE: in the product's kernel the three hooks cost 0.9 % of the headline)
    if (ba.timeline && lane == 0u) {   // (diagnostics: when does this wave start?)
        ba.timeline[4u * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))] = wall_clock64();
    }
#endif
    f3 pos = mk3(0, 0, 0), dir = mk3(0, 0, 1), stepv = mk3(0, 0, 0), rad = mk3(0, 0, 0);
    uint32_t seed = 0, depth = 0, out_idx = 0;
    uint32_t work = 0;      // scheduler visits this path has cost so far (feeds the job order)
    float xi = 0, T = 1;
    uint32_t dfree = 0;     // free-space distance of the brick `pos` is in (0 = unknown / none)
    float inv_maxd = 0;     // 1 / (largest per-axis texel advance of one step)
    int state = ST_IDLE;

    // wave-uniform bookkeeping (lives in SGPRs): the current job and the samples left in it
    JobState job{ 0, 0, 0, 0, 0, 0 };   // the current job and the samples left in it
    bool left_done = false;
    uint32_t q_cur = (uint32_t)kQueues, q_tried = 0;
    bool drained = false;
    uint32_t c_dl = 0, c_il = 0, c_cap = 0; // per-lane tallies
    // What this wave ISSUED, as opposed to what the algorithm counts (c_dl includes replayed free-space steps and the
    // pre-walked prefix, c_il the shadow-volume footprints a lane reused): march fetches, shadow-volume fetches.
    // Wave-uniform sums of lane counts, so they live in SGPRs and cost the lanes nothing (ct_fetch_counters).
    uint32_t w_fetch = 0, w_nee = 0;
    // path conservation (STATS kernels only, CT_DEBUG_INVARIANTS): samples dealt, results written
    uint32_t iv_dealt = 0, iv_written = 0, iv_resumed = 0, iv_suspended = 0;
    uint32_t nee_key = 0xffffffffu;         // fetch_cell_cached: this lane's last shadow-volume footprint
    uint2 nee_cell = make_uint2(0u, 0u);
    // scheduler diagnostics (STATS builds only), see ct_debug_stats
    uint32_t st_regen = 0, st_regen_l = 0, st_march = 0, st_march_l = 0, st_scat = 0, st_scat_l = 0;
    uint32_t st_fetch = 0, st_zero = 0, st_skip = 0, st_zero_d0 = 0, st_zero_d1 = 0, st_skip_iters = 0, st_stolen = 0, st_iters = 0, st_first = 0;
    uint32_t st_hit2 = 0, st_hit3 = 0, st_h2 = 0xffffffffu, st_h3 = 0xffffffffu;   // (diagnostics: a deeper footprint cache)
    uint32_t st_same_line = 0, st_dup_line = 0, st_prev_line = 0xffffffffu;   // brick-line reuse of the march fetches (STATS)

    // ---------------- resume the paths the previous launch suspended ----------------
    // How often this lane's path has been suspended so far (0 = started by this launch).  A path may be handed on while its
    // age is below ba.max_age: the host accumulates a batch only after max_age further launches, so a path that old must
    // be run to its end here (max_age = 1: a resumed path is never suspended again, as until round 3).
    uint32_t age = 0;
    if (ba.cont_in) {
        const uint32_t total = __builtin_amdgcn_readfirstlane(*ba.cont_in_count);
        uint32_t base = 0;
        if (lane == 0) {
            base = atomicAdd(ba.cont_cursor, 64u);
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + lane < total) {
            const uint4 *e = (const uint4 *)(ba.cont_in + (size_t)(base + lane) * kContWords);
            const uint4 w0 = e[0], w1 = e[1], w2 = e[2], w3 = e[3];
            pos = mk3(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z));
            dir = mk3(__uint_as_float(w0.w), __uint_as_float(w1.x), __uint_as_float(w1.y));
            rad = mk3(__uint_as_float(w1.z), __uint_as_float(w1.w), __uint_as_float(w2.x));
            seed = w2.y;
            out_idx = w2.z;
            xi = __uint_as_float(w2.w);
            T = __uint_as_float(w3.x);
            inv_maxd = __uint_as_float(w3.y);
            depth = w3.z & 0xffffu;
            dfree = (w3.z >> 16) & 0xffu;
            state = (int)(w3.z >> 24);
            stepv = scale3(dir, sc.sample_step);
            age = w3.w;
            if (STATS) {
                iv_resumed += 1;
            }
        }
    }
    const bool may_suspend = ba.cont_out != nullptr;
    // One global list (queue 0 holds everything): the waves look at the "list empty" flag now and then and hand their paths
    // on as soon as it is up.  With per-XCD queues -- short launches -- they do NOT: a wave then learns that nothing is left
    // when it runs out of work itself, and finishes what it holds meanwhile.  Round 3 tried the flag there too (it is raised
    // when the last job of EVERY queue has been taken, take_job): the waves hand on 10 % more paths, more of them grow too old
    // to be handed on again, and a 10-subframe launch takes 4.39 ms with a ring of 13 regions and 5.04 with the usual 6,
    // against 4.34 without (tools/launch_timeline.py, profiles/r03y): the 0.9 ms in which such a launch drains are not idle
    // time -- the waves that are left run the faster for being fewer.
    const bool single_queue = ba.q_begin[1] == ba.n_jobs;

    const unsigned long long t_start = STATS ? wall_clock64() : 0ull; // 100 MHz
    unsigned long long t_drained = 0;
#ifdef CT_DIAG_TIMELINE
    bool tl_marked = false;
#endif
    uint32_t visit = 0;
    for (;;) {
        visit += 1;
        if (may_suspend && single_queue && !drained && sc.hint_period != 0u && (visit & (sc.hint_period - 1u)) == 0u) {
            // is the job list empty?  (then this wave finishes its own job and suspends, below.)  The wave
            // that takes the last job raises a flag in a cache line of its own: reading the job counter
            // itself, the target of every wave's atomics, cost 10-30 % of the launch
            const uint32_t empty = __builtin_amdgcn_readfirstlane(__atomic_load_n(ba.queue + kQueueFlag, __ATOMIC_RELAXED));
            if (empty != 0u) {
                // (samples of its own job that this wave has not started go to the next launch with its paths)
                if (job.next != job.end && ba.left_out && job.age < ba.max_age && hand_on_job(ba, lane, job)) {
                    job.next = job.end;
                }
                if (job.next == job.end) {
                    drained = true;
                }
            }
        }
#ifdef CT_DIAG_TIMELINE
        if (ba.timeline && drained && !tl_marked) {   // (diagnostics: when did this wave learn that no job is left, and what did it hold then?)
            tl_marked = true;
            const uint32_t live = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state != ST_IDLE));
            const uint32_t old = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state != ST_IDLE && age >= ba.max_age));
            if (lane == 0u) {
                unsigned long long *tl = ba.timeline + 4u * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
                tl[2] = wall_clock64();
                tl[3] = live | (old << 8) | ((job.next != job.end ? 1u : 0u) << 16);
            }
        }
#endif
        if (STATS) {
            st_iters += 1;
            if (drained && t_drained == 0) {
                t_drained = wall_clock64();
            }
        }
        // ---------------- regenerate ----------------
        const uint64_t idle = __builtin_amdgcn_ballot_w64(state == ST_IDLE);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
        if ((n_idle >= sc.regen_min || n_idle == 64u) && !(drained && job.next == job.end)) {
            if (job.next == job.end && !take_leftover(ba, lane, left_done, job)) {
                uint32_t j = 0;
                if (!take_job(ba, lane, q_cur, q_tried, j)) {
                    drained = true;
                } else {
                    if (STATS) {
                        st_stolen += (q_tried != 0u) ? 1u : 0u;
                    }
                    const uint32_t sub = __builtin_amdgcn_readfirstlane(ba.job_sub[j]);
                    const uint32_t s0 = sub & 0xffffu;
                    job.g = __builtin_amdgcn_readfirstlane(ba.job_group[j]);
                    job.next = 0;
                    // the list may have been built for a larger batch: clip the job to this launch's subframes
                    job.end = (s0 < ba.S ? min(sub >> 16, ba.S - s0) : 0u) * 64u;
                    job.base = ba.out_offset + s0 * ba.frame_stride + group_column(ba, job.g);
                    job.first = ba.first_subframe + s0;
                    job.age = 0;
                }
            }
            if (job.next != job.end) {
                const uint32_t avail = job.end - job.next;
                if (STATS) {
                    st_regen += 1;
                    st_regen_l += min(n_idle, avail);
                }
                const uint32_t rank = lane_rank(idle);
                const bool take = (state == ST_IDLE) && rank < avail;
                const uint32_t q = job.next + rank;
                job.next += min(n_idle, avail);
                if (take) {
                    const uint32_t s = q >> 6, l = q & 63u;   // (subframe within the job, lane)
                    const uint32_t g = job.g;
                    const uint32_t pixel = ba.pixels[g * 64u + l];
                    if (pixel != 0xffffffffu) {
                        if (STATS) {
                            iv_dealt += 1;
                        }
                        const float4 p0 = ba.primary[2 * (size_t)pixel];
                        const float4 p1 = ba.primary[2 * (size_t)pixel + 1];
                        out_idx = ba.frame_stride ? job.base + s * ba.frame_stride + l : pixel;
                        pos = mk3(p0.x, p0.y, p0.z);
                        const bool hit = p0.w != 0.f; // image jobs list hitting pixels only; point tasks may miss
                        dir = mk3(p1.x, p1.y, p1.z);
                        seed = tea4(__float_as_uint(p1.w), job.first + s); // :21
                        rad = mk3(0, 0, 0);
                        depth = 0;
                        work = 0;
                        age = job.age;
                        if (MODE == 1) {
                            dir = new_direction(lds.cdf, lds.guide, seed, dir);  // :86
                        }
                        // first loop test + depth bump (:28-34); mode 2 has no loop (:134)
                        bool go = hit && in_box(sc, pos);
                        if (MODE != 2 && go) {
                            depth = 1;
                            if (depth == sc.max_depth) {
                                c_cap += 1;
                                go = false;
                            }
                        }
                        if (go) {
                            xi = u24_to_float(lcg24(seed));
                            T = 1;
                            stepv = scale3(dir, sc.sample_step);
                            inv_maxd = inv_max_advance(sc, stepv);
                            dfree = 0;
                            if (MODE != 1 && ba.advance) {
                                // the pixel's pre-walked prefix (primary_advance_kernel)
                                const float4 a = ba.advance[pixel];
                                const uint32_t bits = __float_as_uint(a.w);
                                pos = mk3(a.x, a.y, a.z);
                                dfree = bits >> 24;
                                c_dl += bits & 0x00ffffffu;
                            }
                            state = ST_MARCH;
                        } else {
                            ba.frames[out_idx] = make_float4(0.f, 0.f, 0.f, 1.f);
                            if (STATS) {
                                iv_written += 1;
                            }
                        }
                    }
                }
            }
        }

        const uint64_t marching = __builtin_amdgcn_ballot_w64(state == ST_MARCH);
        const uint64_t bouncing = __builtin_amdgcn_ballot_w64(state == ST_BOUNCE);
        const uint32_t nm = (uint32_t)__builtin_popcountll(marching);
        const uint32_t nb = (uint32_t)__builtin_popcountll(bouncing);
        if ((marching | bouncing) == 0 && drained && job.next == job.end) {
            break;
        }
        // (everything idle but samples left: both phases below are no-ops and the loop regenerates)

        bool finished = false;
        const bool do_scatter = nb * sc.scatter_den > nm * sc.scatter_num && (nb >= sc.scatter_min || nm == 0);
        if (do_scatter) {
            // ---------------- scatter: NEE + new direction (cloudRadianceMaterials.cu:53-61) ----------------
            if (STATS) {
                st_scat += 1;
                st_scat_l += nb;
            }
            bool nee_fetched = false;
            if (state == ST_BOUNCE) {
                // The collision's back-off (cloud.cuh:99) was left for this phase: in the march phase
                // one lane in fifteen collides per step, so its log and two divisions would run for
                // nearly every wave at 7 % lane occupancy.  `inv_maxd` carries the density sampled at
                // the collision, `dfree` the brick's "interior" flag (both are reset below).
                {
                    const float lg = logf_above_one(div_(xi, T));
                    // (the density of a collision is of moderate magnitude: T fell at this step, so exp(-density * step) < 1,
                    // i.e. density * step >= 2^-25; and density <= densityMultiplier < 80 / step)
                    const float inv = rcp_moderate(inv_maxd);
                    pos = sub3(pos, scale3(scale3(dir, lg), inv)); // scatterPos, :99
                }
                // isInBox(scatterPos), cloudRadianceMaterials.cu:49-52
                if (dfree != 0u || in_box(sc, pos)) {
                    const bool chopped = (MODE == 1) ? true : (MODE == 0 ? (depth != 1) : false);
                    bool nee_reused;
                    const uint32_t st_key_before = nee_key;
                    const NeeLoads nee = in_scattering_issue_lds(sc, lds.chopped, pos, dir, chopped, nee_key, nee_cell, nee_reused);
                    if (STATS && !nee_reused) {
                        // (would a second / third entry -- the footprints before the one just replaced -- have held this one?)
                        st_hit2 += (nee_key == st_h2) ? 1u : 0u;
                        st_hit3 += (nee_key != st_h2 && nee_key == st_h3) ? 1u : 0u;
                        st_h3 = st_h2;
                        st_h2 = st_key_before;
                    }
                    nee_fetched = !nee_reused;
                    if (STATS) {
                        st_first += nee_reused ? 1u : 0u;
                        if (ba.touched_shadow && !nee_reused) {
                            const uint32_t line = (uint32_t)(apron_offset_in_grid(sc, pos) >> 7);
                            atomicOr(&ba.touched_shadow[line >> 5], 1u << (line & 31u));
                        }
                    }
                    c_il += 1;
                    if (COST) {
                        work += 4u;
                    }
                    bool go = (MODE != 2);
                    if (go) {
                        dir = new_direction(lds.cdf, lds.guide, seed, dir);
                        depth++;
                        if (depth == sc.max_depth) {
                            c_cap += 1;
                            go = false;
                        }
                    }
                    rad = add3(rad, in_scattering_finish(sc, nee, pos));
                    // (setting the next flight up for every lane and selecting the state -- no region for the rare path at its
                    // depth cap -- measured: MARCH -0.7 %, the DELTA kernel's counterpart -1.6 %, profiles/r04aa)
                    if (go) {
                        xi = u24_to_float(lcg24(seed));
                        T = 1;
                        stepv = scale3(dir, sc.sample_step);
                        inv_maxd = inv_max_advance(sc, stepv);
                        dfree = 0;
                        state = ST_MARCH;
                    } else {
                        finished = true;
                    }
                } else {
                    finished = true;
                }
            }
            w_nee += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(nee_fetched));
        }
        // (MARCH_FUSE: the march burst follows a scatter phase in the same scheduler iteration, as in render_delta_kernel)
        bool track = !do_scatter;
        uint32_t nm_track = nm;
        if (do_scatter && MARCH_FUSE) {
            nm_track = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_MARCH));
            track = nm_track != 0u;
        }
        if (track) {
            // ---------------- march (getNextScatteringEvent, cloud.cuh:87-105) ----------------
            if (STATS) {
                st_march += 1;
                st_march_l += nm_track;
            }
            w_fetch += nm_track; // every marching lane issues one footprint fetch per burst iteration
            // A burst of up to sc.march_burst steps per scheduler visit: the scheduler's own
            // instructions are paid once per burst, and the lanes that collide meanwhile wait for a
            // fuller scatter phase.  The burst ends early when enough lanes wait for the scatter
            // phase or for new samples, or nobody marches any more.
            // once the queue is empty nothing is left to amortise: what remains of the launch is the
            // latency of its longest paths, so a collided lane no longer waits for a burst to end
            uint32_t burst = drained ? sc.tail_burst : sc.march_burst;
            for (;;) {
            if (state == ST_MARCH) {
                // Free-space skip: every brick within Chebyshev distance dfree-1 of the one `pos` is
                // in is free, so the next n steps can neither collide (all 8 texels are 0, T *= 1)
                // nor leave the box; only the position updates have to be replayed (same float adds
                // in the same order -> bit-identical path).  They still count as density lookups:
                // the counter is the algorithm's lookup count, not the loads this kernel issued.
                if (dfree != 0u) {
                    const int n = skip_steps(dfree, inv_maxd);
                    replay_steps(pos, stepv, n);
                    c_dl += (uint32_t)n;
                    if (STATS) {
                        st_skip += (uint32_t)n;
                        // wave-level iterations of the replay loop = the largest n of the wave
                        int wmax = n;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) {
                            wmax = max(wmax, __shfl_xor(wmax, off));
                        }
                        st_skip_iters += (lane_rank(__builtin_amdgcn_ballot_w64(true)) == 0u) ? (uint32_t)wmax : 0u;
                    }
                }
                pos = add3(pos, stepv);
                uint32_t meta;
                const uint2 cell = fetch_cell_m<SPARSE>(sc, pos, meta);
                dfree = meta & 0x7fu;
                c_dl += 1;
                if (COST) {
                    work += 1u;
                }
                if (STATS) {
                    st_fetch += 1;
                    // (round-3 question: would a brick cache in LDS find anything?  The 128-B line of this footprint against
                    // the lane's previous one, and against the lines the wave's other lanes fetch in this same instruction.)
                    {
                        const float fx = fmaf(pos.x, sc.sx, -0.5f), fy = fmaf(pos.y, sc.sy, -0.5f), fz = fmaf(pos.z, sc.sz, -0.5f);
                        const uint32_t lx = (uint32_t)(floor_to_int(fx) + sc.m_bias_x), ly = (uint32_t)(floor_to_int(fy) + sc.brick_bias),
                                       lz = (uint32_t)(floor_to_int(fz) + sc.brick_bias);
                        const uint32_t line = __umul24(lz >> 2, (uint32_t)sc.m_gxy) + __umul24(ly >> 2, (uint32_t)sc.m_gx) + (__umul24(lx, 43691u) >> 17);
                        st_same_line += (line == st_prev_line) ? 1u : 0u;
                        st_prev_line = line;
                        if (ba.touched_density && !SPARSE) {   // ct_debug_track_lines: the distinct 128-B lines a launch reads
                            atomicOr(&ba.touched_density[line >> 5], 1u << (line & 31u));
                        }
                        const uint64_t act = __builtin_amdgcn_ballot_w64(true);
                        bool dup = false;
                        for (uint32_t i = 0; i < 64u; i++) {
                            const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)line, (int)i);
                            dup = dup || (((act >> i) & 1ull) != 0ull && i < lane && other == line);
                        }
                        st_dup_line += dup ? 1u : 0u;
                    }
                    st_zero += ((cell.x | cell.y) == 0u) ? 1u : 0u;
                    st_zero_d0 += ((cell.x | cell.y) == 0u && dfree == 0u) ? 1u : 0u;
                    st_zero_d1 += ((cell.x | cell.y) == 0u && dfree == 1u) ? 1u : 0u;
                }
                bool collided = false;
                if (CT_MARCH_FLAT_ZERO || (cell.x | cell.y) != 0u) {
                    // all-zero footprints give density 0, exp(-0) = 1, T unchanged: nothing to evaluate
                    const float density = filter_at(sc, cell, pos) * sc.density_multiplier;
                    const float extinction = density * sc.sample_step;
                    T *= expf_inrange(-extinction);
                    if (xi > T) {
                        // collision: scatterPos and its box test are evaluated by the scatter phase
                        collided = true;
                        inv_maxd = density;
                    }
                }
                // isInBox(pos), the loop condition of cloud.cuh:87.  In an "interior" brick it is
                // known to hold (see DevScene), so the six comparisons are skipped.
#if CT_MARCH_FLAT_EXIT
                state = collided ? ST_BOUNCE : state;
                dfree = collided ? (meta & 0x80u) : dfree;
                if (!collided & ((meta & 0x80u) == 0u) & !in_box_flat(sc, pos)) {
#else
                if (collided) {
                    state = ST_BOUNCE;
                    dfree = meta & 0x80u;
                } else if ((meta & 0x80u) == 0u && !in_box(sc, pos)) {
#endif
                    ba.frames[out_idx] = make_float4(rad.x, rad.y, rad.z, 1.f);
                    if (STATS) {
                        iv_written += 1;
                    }
                    if (COST && ba.cost) {
                        ba.cost[out_idx - ba.out_offset] = make_uint2(work, depth);
                    }
                    state = ST_IDLE;
                }
            }
            if (--burst == 0u) {
                break;
            }
            const uint32_t m_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_MARCH));
            const uint32_t b_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_BOUNCE));
            // (these conditions as bitwise tests on four thresholds fixed before the burst -- one scalar branch per step instead of five --
            // measured: no difference, here or in render_delta_kernel, profiles/r04aa: scalar branches are not what costs)
            if (drained) {
                // nothing left to regenerate, so idle lanes are no reason to leave the burst: what matters
                // now is the latency of the surviving paths -- march until half of them wait for the scatter
                if (m_now == 0u || 2u * b_now >= m_now + b_now) {
                    break;
                }
            } else if (m_now < sc.burst_march_min || b_now >= sc.burst_scatter || 64u - m_now - b_now >= sc.burst_idle) {
                break;
            }
            w_fetch += m_now;
            if (STATS) {
                st_march += 1;
                st_march_l += m_now;
            }
            }
        }
        if (finished) {
            ba.frames[out_idx] = make_float4(rad.x, rad.y, rad.z, 1.f);
            if (STATS) {
                iv_written += 1;
            }
            if (COST && ba.cost) {
                ba.cost[out_idx - ba.out_offset] = make_uint2(work, depth);
            }
            state = ST_IDLE;
        }
        // ---------------- suspend: nothing left to take, hand the surviving paths to the next launch ----------------
        // (only once this wave's own job is used up: a lane then suspends at most one path per launch, which
        // is what the next launch can resume -- 64 paths per wave -- and what cont_out is sized for)
        if (may_suspend && drained && job.next == job.end) {
            const bool mine = state != ST_IDLE && age < ba.max_age;
            const uint64_t live = __builtin_amdgcn_ballot_w64(mine);
            if (live != 0ull) {
                const uint32_t n = (uint32_t)__builtin_popcountll(live);
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(live)) {
                    base = atomicAdd(ba.cont_out_count, n);
                    if (base + n > ba.cont_capacity) {
                        atomicSub(ba.cont_out_count, n); // no room (cannot happen with the host's sizing): keep running
                        base = 0xffffffffu;
                    } else if (ba.cont_total) {
                        atomicAdd(ba.cont_total, (unsigned long long)n);
                    }
                }
                base = __builtin_amdgcn_readlane(base, __builtin_ctzll(live));
                if (mine && base != 0xffffffffu) {
                    uint4 *e = (uint4 *)(ba.cont_out + (size_t)(base + lane_rank(live)) * kContWords);
                    e[0] = make_uint4(__float_as_uint(pos.x), __float_as_uint(pos.y), __float_as_uint(pos.z), __float_as_uint(dir.x));
                    e[1] = make_uint4(__float_as_uint(dir.y), __float_as_uint(dir.z), __float_as_uint(rad.x), __float_as_uint(rad.y));
                    e[2] = make_uint4(__float_as_uint(rad.z), seed, out_idx, __float_as_uint(xi));
                    e[3] = make_uint4(__float_as_uint(T), __float_as_uint(inv_maxd), depth | (dfree << 16) | ((uint32_t)state << 24), age + 1u);
                    state = ST_IDLE;
                    if (STATS) {
                        iv_suspended += 1;
                    }
                }
            }
        }
    }

#ifdef CT_DIAG_TIMELINE
    if (ba.timeline && lane == 0u) {   // (... and when does it end?)
        ba.timeline[4u * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + 1u] = wall_clock64();
    }
#endif
    // flush counters: per-lane tallies -> one atomic per counter per wave
    uint32_t vals[3] = { c_dl, c_il, c_cap };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint32_t v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_xor(v, off);
        }
        vals[i] = v;
    }
    if (STATS) {
        uint32_t sv[15] = { st_fetch, st_zero, st_skip, st_zero_d0, st_zero_d1, st_skip_iters, st_first,
                            iv_dealt, iv_resumed, iv_written, iv_suspended, st_same_line, st_dup_line, st_hit2, st_hit3 };
#pragma unroll
        for (int i = 0; i < 15; i++) {
            uint32_t v = sv[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                v += __shfl_xor(v, off);
            }
            sv[i] = v;
        }
        if (lane == 0) {
            atomicAdd(&ba.stats[0], (unsigned long long)st_regen);
            atomicAdd(&ba.stats[1], (unsigned long long)st_regen_l);
            atomicAdd(&ba.stats[2], (unsigned long long)st_march);
            atomicAdd(&ba.stats[3], (unsigned long long)st_march_l);
            atomicAdd(&ba.stats[4], (unsigned long long)st_scat);
            atomicAdd(&ba.stats[5], (unsigned long long)st_scat_l);
            atomicAdd(&ba.stats[6], (unsigned long long)sv[0]);
            atomicAdd(&ba.stats[7], (unsigned long long)sv[1]);
            atomicAdd(&ba.stats[8], (unsigned long long)sv[2]);
            atomicAdd(&ba.stats[9], (unsigned long long)sv[3]);
            atomicAdd(&ba.stats[10], (unsigned long long)sv[4]);
            atomicAdd(&ba.stats[11], (unsigned long long)sv[5]);
            atomicAdd(&ba.stats[12], (unsigned long long)sv[6]);
            atomicAdd(&ba.stats[13], 1ull);
            atomicAdd(&ba.stats[14], (unsigned long long)st_stolen);
            atomicMax(&ba.stats[15], (unsigned long long)st_iters);
            const unsigned long long t_end = wall_clock64();
#ifdef CT_STATS_FINE   // (analysis builds: 0.25 ms / 0.05 ms bins for launches of a few milliseconds)
            atomicAdd(&ba.stats[16 + min((unsigned long long)23, (t_end - t_start) / 25000ull)], 1ull);
            atomicAdd(&ba.stats[40 + min((unsigned long long)23, (t_end - (t_drained ? t_drained : t_end)) / 5000ull)], 1ull);
#else
            atomicAdd(&ba.stats[16 + min((unsigned long long)23, (t_end - t_start) / 500000ull)], 1ull);
            atomicAdd(&ba.stats[40 + min((unsigned long long)23, (t_end - (t_drained ? t_drained : t_end)) / 50000ull)], 1ull);
#endif
            atomicAdd(&ba.stats[64], (unsigned long long)sv[7]);
            atomicAdd(&ba.stats[65], (unsigned long long)sv[8]);
            atomicAdd(&ba.stats[66], (unsigned long long)sv[9]);
            atomicAdd(&ba.stats[67], (unsigned long long)sv[10]);
            atomicAdd(&ba.stats[68], (unsigned long long)sv[11]);
            atomicAdd(&ba.stats[69], (unsigned long long)sv[12]);
            atomicAdd(&ba.stats[70], (unsigned long long)sv[13]);
            atomicAdd(&ba.stats[71], (unsigned long long)sv[14]);
        }
    }
    if (lane == 0) {
        atomicAdd(&ba.counters[2], (unsigned long long)vals[0]);
        atomicAdd(&ba.counters[3], (unsigned long long)vals[1]);
        atomicAdd(&ba.counters[4], (unsigned long long)vals[1]); // scatter events == NEE lookups
        atomicAdd(&ba.counters[5], (unsigned long long)vals[2]);
        atomicAdd(&ba.counters[6], (unsigned long long)w_fetch);
        atomicAdd(&ba.counters[7], (unsigned long long)w_nee);
    }
}

// =============================================================================================
// Density pyramid + hierarchical descriptor
// =============================================================================================
// Resources::generateMipmaps, Resources.cpp:193-203: uint16 sum of the (up to) 8 children, zero
// outside the parent level, / 8.
__global__ void mip_level_kernel(const uint8_t *__restrict__ prev, int px, int py, int pz, uint8_t *__restrict__ cur,
                                 int cx, int cy, int cz)
{
    const int64_t total = (int64_t)cx * cy * cz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % cx), y = (int)((i / cx) % cy), z = (int)(i / ((int64_t)cx * cy));
        uint32_t acc = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const int sx = 2 * x + (d & 1), sy = 2 * y + ((d >> 1) & 1), sz = 2 * z + (d >> 2);
            if (sx < px && sy < py && sz < pz) {
                acc += prev[((size_t)sz * py + sy) * px + sx];
            }
        }
        cur[i] = (uint8_t)(acc / 8u);
    }
}

hipError_t launch_mip_level(const uint8_t *prev, int px, int py, int pz, uint8_t *cur, int cx, int cy, int cz,
                            hipStream_t stream)
{
    const int64_t total = (int64_t)cx * cy * cz;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(mip_level_kernel, dim3(blocks), dim3(256), 0, stream, prev, px, py, pz, cur, cx, cy, cz);
    return hipGetLastError();
}

// One level of rtTex3DLod: trilinear, clamp-to-edge, x = fma(pos, textureScale * dim, -0.5).
CT_DEV float tex3_level(const DevScene &sc, const MipPyramid &mp, uint32_t l, f3 p)
{
    const int32_t nx = mp.nx[l], ny = mp.ny[l], nz = mp.nz[l];
    const float x = fmaf(p.x, sc.tsx * (float)nx, -0.5f), y = fmaf(p.y, sc.tsy * (float)ny, -0.5f),
                z = fmaf(p.z, sc.tsz * (float)nz, -0.5f);
    const float flx = floorf(x), fly = floorf(y), flz = floorf(z);
    const float wx = fract_(x), wy = fract_(y), wz = fract_(z);
    const int32_t ix = (int32_t)flx, iy = (int32_t)fly, iz = (int32_t)flz;
    const int32_t x0 = min(max(ix, 0), nx - 1), x1 = min(max(ix + 1, 0), nx - 1);
    const int32_t y0 = min(max(iy, 0), ny - 1), y1 = min(max(iy + 1, 0), ny - 1);
    const int32_t z0 = min(max(iz, 0), nz - 1), z1 = min(max(iz + 1, 0), nz - 1);
    const uint8_t *b = mp.base + mp.offset[l];
    const size_t sy_ = (size_t)nx, sz_ = (size_t)nx * (size_t)ny;
    uint2 c;
    c.x = (uint32_t)b[z0 * sz_ + y0 * sy_ + x0] | ((uint32_t)b[z0 * sz_ + y0 * sy_ + x1] << 8) |
          ((uint32_t)b[z0 * sz_ + y1 * sy_ + x0] << 16) | ((uint32_t)b[z0 * sz_ + y1 * sy_ + x1] << 24);
    c.y = (uint32_t)b[z1 * sz_ + y0 * sy_ + x0] | ((uint32_t)b[z1 * sz_ + y0 * sy_ + x1] << 8) |
          ((uint32_t)b[z1 * sz_ + y1 * sy_ + x0] << 16) | ((uint32_t)b[z1 * sz_ + y1 * sy_ + x1] << 24);
    return filter_cell(c, wx, wy, wz);
}

// setupHierarchicalDescriptor, DisneyDescriptor.cuh:71-112.  One block per (sample, layer), one thread
// per grid point: the 225 points of a layer touch neighbouring texels of one or two pyramid levels.
__global__ __launch_bounds__(256) void descriptor_kernel(DevScene sc, MipPyramid mp, const float *__restrict__ positions,
                                                         const float *__restrict__ directions, uint32_t count,
                                                         float level0, float voxel_m, float cloud_size_m,
                                                         uint8_t *__restrict__ out)
{
    const uint32_t layer = blockIdx.x % 10u, i = blockIdx.x / 10u;
    const uint32_t sample = threadIdx.x;
    if (i >= count || sample >= 225u) {
        return;
    }
    const f3 world = mk3(positions[3 * (size_t)i], positions[3 * (size_t)i + 1], positions[3 * (size_t)i + 2]);
    const f3 view = mk3(directions[3 * (size_t)i], directions[3 * (size_t)i + 1], directions[3 * (size_t)i + 2]);
    const f3 ez = normalize3(mk3(sc.nlx, sc.nly, sc.nlz)); // normalize(-lightDirection), :76
    const f3 ex = normalize3(cross3(ez, view));
    const f3 ey = cross3(ex, ez);
    const f3 origin = add3(world, scale3(mk3(sc.bx, sc.by, sc.bz), 0.5f));
    float scale = 0.5f / sc.density_multiplier;
    float lod = level0;
    for (uint32_t l = 0; l < layer; l++) { // the same float operations as the reference's layer loop
        scale *= 2;
        lod += 1;
    }
    const float mip_voxel = ct_powf(2.0f, lod) * voxel_m / cloud_size_m;
    const int x = (int)(sample % 5u) - 2, y = (int)((sample / 5u) % 5u) - 2, z = (int)(sample / 25u) - 2;
    const f3 dir = add3(add3(scale3(ex, (float)x), scale3(ey, (float)y)), scale3(ez, (float)z));
    const f3 pos = add3(origin, scale3(dir, scale));
    // rtTex3DLod, mip-linear
    float lc = fminf(fmaxf(0.0f, lod), (float)(mp.levels - 1u));
    const float fl = floorf(lc);
    const uint32_t l0 = (uint32_t)fl;
    const float w = lc - fl;
    float density = tex3_level(sc, mp, l0, pos);
    if (w > 0.0f) {
        const float s1 = tex3_level(sc, mp, min(l0 + 1u, mp.levels - 1u), pos);
        density = fmaf(w, s1 - density, density);
    }
    // distanceToBox, :47-55
    const f3 half = scale3(mk3(sc.bx, sc.by, sc.bz), 0.5f);
    f3 dist = sub3(pos, half);
    dist = mk3(fabsf(dist.x), fabsf(dist.y), fabsf(dist.z));
    const float hv = mip_voxel * 0.5f;
    dist = sub3(dist, mk3(fmaxf(half.x - hv, 0.f), fmaxf(half.y - hv, 0.f), fmaxf(half.z - hv, 0.f)));
    dist = mk3(fmaxf(dist.x, 0.f), fmaxf(dist.y, 0.f), fmaxf(dist.z, 0.f));
    const float distance = sqrtf(dot3(dist, dist));
    const float t = fminf(fmaxf(distance / mip_voxel, 0.0f), 1.0f);
    density = density + t * (0.0f - density);
    out[((size_t)i * 10u + layer) * 225u + sample] = (uint8_t)(density * 255.0f);
}

hipError_t launch_descriptors(const DevScene &sc, const MipPyramid &mp, const float *positions, const float *directions,
                              uint32_t count, float level0, float voxel_m, float cloud_size_m, uint8_t *out,
                              hipStream_t stream)
{
    hipLaunchKernelGGL(descriptor_kernel, dim3(count * 10u), dim3(256), 0, stream, sc, mp, positions, directions, count,
                       level0, voxel_m, cloud_size_m, out);
    return hipGetLastError();
}

// =============================================================================================
// DELTA estimator: Woodcock tracking over a grid of majorant cells (BASELINE.json north_star; not the
// reference's sampler -- the oracle twin is delta_flight() in oracle/ct_oracle.c, whose header
// states the algorithm).  Same persistent wave scheduler.  The majorant grid (DevScene::maj_cells, cells
// of 2^mc_shift texels, at most kMajCellsMax of them) is copied to LDS by every block, so the DDA that
// carries a flight from cell to cell reads no memory; the "march" phase becomes a tracking visit:
//     current cell M == 0 -> cross into the next one; M > 0 -> draw dt = -log(1-u)/sigma_bar, and if it
//     leaves the cell, cross (the exponential is memoryless); otherwise advance, load the footprint at
//     p = fma(dir, t, origin) -- the visit's one global load -- and accept the collision when
//     u' * sigma_bar < sigma(p).  (Several crossings per visit were tried: the lanes that keep crossing
//     hold up the wave; one step per visit and bursts of 3 visits measured best.)
// =============================================================================================
#ifndef CT_DELTA_THREADS
#define CT_DELTA_THREADS 768   // (A/B builds: -DCT_DELTA_THREADS=1024 -DCT_DELTA_WAVES=8 = two 1024-thread blocks per CU, 8 waves per SIMD, 64 VGPRs)
#endif
#ifndef CT_DELTA_WAVES
#define CT_DELTA_WAVES 6
#endif
#ifndef CT_DELTA_FUSE
#define CT_DELTA_FUSE 1        // (see CT_MARCH_FUSE)
#endif
constexpr bool DELTA_FUSE = CT_DELTA_FUSE != 0;
#ifndef CT_DELTA_CHECK_EVERY
#define CT_DELTA_CHECK_EVERY 2   // (5303 -> 5324 Msamples/s against 1, profiles/r04o; -DCT_DELTA_CHECK_EVERY=1: after every visit, as until round 4)
#endif
constexpr uint32_t DELTA_CHECK_EVERY = CT_DELTA_CHECK_EVERY;
#ifndef CT_DELTA_FLAT_EMPTY
#define CT_DELTA_FLAT_EMPTY 1
#endif
#ifndef CT_DELTA_END_MERGE
#define CT_DELTA_END_MERGE 1     // (a flight that crosses out of the stored box ends in the visit of that crossing; 0: in its next visit, as until round 4)
#endif
constexpr bool DELTA_END_MERGE = CT_DELTA_END_MERGE != 0;
constexpr int kDeltaThreads = CT_DELTA_THREADS;

struct Dda {
    f3 org;            // origin of the flight (box coordinates), positions are fma(dir, t, org)
    float t;
    f3 tmax, tdelta;   // ray parameter at the next cell boundary per axis / between boundaries
    int32_t bx, by, bz;
};

// Is the flight's cell one of the STORED cells (the box around the cloud)?  A flight that leaves the box is over: a straight
// line does not come back into a box, and outside it every cell is empty (no random number is drawn there).
CT_DEV bool cell_in_grid(const DevScene &sc, const Dda &d)
{
    // (one test, not three nested ones: the three comparisons are combined on the lane masks)
    return ((uint32_t)(d.bx - sc.mc_x0) < (uint32_t)sc.mc_gx) & ((uint32_t)(d.by - sc.mc_y0) < (uint32_t)sc.mc_gy) &
           ((uint32_t)(d.bz - sc.mc_z0) < (uint32_t)sc.mc_gz);
}

CT_DEV bool cell_in_virtual_grid(const DevScene &sc, const Dda &d)
{
    return ((uint32_t)d.bx < (uint32_t)sc.mc_vx) & ((uint32_t)d.by < (uint32_t)sc.mc_vy) & ((uint32_t)d.bz < (uint32_t)sc.mc_vz);
}

CT_DEV uint32_t cell_index(const DevScene &sc, const Dda &d)
{
    return __umul24((uint32_t)(d.bz - sc.mc_z0), (uint32_t)(sc.mc_gx * sc.mc_gy)) + __umul24((uint32_t)(d.by - sc.mc_y0), (uint32_t)sc.mc_gx) +
           (uint32_t)(d.bx - sc.mc_x0);
}

// The same two for a flight whose cell coordinates are kept RELATIVE to the stored box's first cell (render_delta_kernel with
// DELTA_END_MERGE: no subtraction per visit).
CT_DEV bool cell_in_grid_rel(const DevScene &sc, const Dda &d)
{
    return ((uint32_t)d.bx < (uint32_t)sc.mc_gx) & ((uint32_t)d.by < (uint32_t)sc.mc_gy) & ((uint32_t)d.bz < (uint32_t)sc.mc_gz);
}
CT_DEV uint32_t cell_index_rel(const DevScene &sc, const Dda &d)
{
    return __umul24((uint32_t)d.bz, (uint32_t)(sc.mc_gx * sc.mc_gy)) + __umul24((uint32_t)d.by, (uint32_t)sc.mc_gx) + (uint32_t)d.bx;
}
CT_DEV bool dda_enter_box(const DevScene &sc, Dda &d)   // absolute -> relative; is the flight's cell a stored one?
{
    const bool inside = cell_in_grid(sc, d);
    d.bx -= sc.mc_x0;
    d.by -= sc.mc_y0;
    d.bz -= sc.mc_z0;
    return inside;
}

CT_DEV void dda_cross(Dda &d, f3 dir);

// DDA set-up of a flight from `pos` along `dir`.
CT_DEV void dda_begin(const DevScene &sc, Dda &d, f3 pos, f3 dir)
{
    d.org = pos;
    d.t = 0.0f;
    const float tpx = fmaf(pos.x, sc.sx, -0.5f), tpy = fmaf(pos.y, sc.sy, -0.5f), tpz = fmaf(pos.z, sc.sz, -0.5f);
    const float vx = dir.x * sc.sx, vy = dir.y * sc.sy, vz = dir.z * sc.sz;
    // texel index / cell size by a multiplication (exact for every index of the grid: checked when the grid is made)
    d.bx = (int32_t)(__umul24((uint32_t)(floor_to_int(tpx) + sc.brick_bias), (uint32_t)sc.mc_div) >> 20);
    d.by = (int32_t)(__umul24((uint32_t)(floor_to_int(tpy) + sc.brick_bias), (uint32_t)sc.mc_div) >> 20);
    d.bz = (int32_t)(__umul24((uint32_t)(floor_to_int(tpz) + sc.brick_bias), (uint32_t)sc.mc_div) >> 20);
    const float inf = __uint_as_float(0x7f800000u);
    const float edge = (float)sc.mc_cell;
    // One division per axis and NO branch around it: the lanes of a wave disagree about the sign of V, so an
    // if / else-if with a division in each arm made every wave execute both (six IEEE divisions per set-up).  The values
    // are the oracle's: inv = 1 / V; V > 0: tmax = (upper bound - tp) * inv, tdelta = edge * inv; V < 0: tmax = (lower
    // bound - tp) * inv, tdelta = edge * -inv = edge * |inv|; V == 0: both infinite.
    // The reciprocals: correctly rounded either way.  hipcc's IEEE sequence (eleven instructions, most of them scaling for
    // subnormal and huge arguments) is needed only when a component of V is tiny -- a direction that is parallel to an axis
    // to within 2^-60 -- so the wave asks once whether ANY of its lanes has such a component and otherwise takes
    // rcp_moderate's three instructions per axis (every float of its range checked on the device).
    // An exact zero counts as tiny here: axis-parallel rays are rare outside tests.
    const float v_min = fminf(fminf(fabsf(vx), fabsf(vy)), fabsf(vz)), v_max = fmaxf(fmaxf(fabsf(vx), fabsf(vy)), fabsf(vz));
    const bool moderate = v_min >= 0x1p-60f && v_max <= 0x1p60f;
    auto axis = [&](int32_t B, float TP, float V, float inv, float &TMAX, float &TDELTA) {
        const int32_t bound = (int32_t)__umul24((uint32_t)(B + (V > 0.0f ? 1 : 0)), (uint32_t)sc.mc_cell) - sc.brick_bias;
        const bool moving = V != 0.0f;
        TMAX = moving ? ((float)bound - TP) * inv : inf;
        TDELTA = moving ? edge * fabsf(inv) : inf;
    };
    // (every lane of the wave moderate: no component is zero, so the test for a resting axis -- a divergent region per axis in
    // the compiled code -- is left out; the values are the same.  5525 -> 5603 Msamples/s, profiles/r04aa)
    auto axis_moving = [&](int32_t B, float TP, float V, float inv, float &TMAX, float &TDELTA) {
        const int32_t bound = (int32_t)__umul24((uint32_t)(B + (V > 0.0f ? 1 : 0)), (uint32_t)sc.mc_cell) - sc.brick_bias;
        TMAX = ((float)bound - TP) * inv;
        TDELTA = edge * fabsf(inv);
    };
    if (__builtin_amdgcn_ballot_w64(!moderate) == 0ull) {
        axis_moving(d.bx, tpx, vx, rcp_moderate(vx), d.tmax.x, d.tdelta.x);
        axis_moving(d.by, tpy, vy, rcp_moderate(vy), d.tmax.y, d.tdelta.y);
        axis_moving(d.bz, tpz, vz, rcp_moderate(vz), d.tmax.z, d.tdelta.z);
    } else {
        axis(d.bx, tpx, vx, rcp_(vx), d.tmax.x, d.tdelta.x);
        axis(d.by, tpy, vy, rcp_(vy), d.tmax.y, d.tdelta.y);
        axis(d.bz, tpz, vz, rcp_(vz), d.tmax.z, d.tdelta.z);
    }
    // A flight that starts OUTSIDE the stored box (a primary ray that enters the volume beside the cloud; a scatter position that
    // rounds into the neighbour of a cell on the box's face) crosses the empty virtual cells up to it, with the crossings' own
    // arithmetic -- the oracle's flight steps through the same cells one by one -- or leaves the virtual grid without meeting it.
    // (A scatter position lies in the cloud, a pre-walked primary ray does not come here: a wave rarely holds such a lane, and asks
    // once -- the loop's own header, two box tests per lane, is not run otherwise: +0.3 %, profiles/r04aa.)
    if (__builtin_amdgcn_ballot_w64(!cell_in_grid(sc, d)) == 0ull) {
        return;
    }
    while (cell_in_virtual_grid(sc, d) && !cell_in_grid(sc, d)) {
        dda_cross(d, dir);
    }
}

// One cell crossing: t = exit parameter, step along the axis with the smallest tmax
// (ties: x before y before z), exactly like the oracle.
CT_DEV void dda_cross(Dda &d, f3 dir)
{
    const float t_exit = fminf(fminf(d.tmax.x, d.tmax.y), d.tmax.z);
    d.t = t_exit;
    // (three exclusive divergent regions; as selects -- three lane masks alive at once in a kernel that already spills scalar
    // registers -- the DELTA kernel lost 5 %: 5628 -> 5325 Msamples/s, profiles/r04aa)
    if (d.tmax.x <= d.tmax.y && d.tmax.x <= d.tmax.z) {
        d.bx += (dir.x > 0.0f) ? 1 : -1;
        d.tmax.x += d.tdelta.x;
    } else if (d.tmax.y <= d.tmax.z) {
        d.by += (dir.y > 0.0f) ? 1 : -1;
        d.tmax.y += d.tdelta.y;
    } else {
        d.bz += (dir.z > 0.0f) ? 1 : -1;
        d.tmax.z += d.tdelta.z;
    }
}

// The DELTA twin of primary_advance_kernel: the cell DDA of a primary ray draws no random number before
// it reaches the first cell with a non-zero majorant (or leaves the grid), so that walk is the same
// for every sample of the pixel.  It is done once per pose with the tracking visit's own operations;
// advance[4p .. 4p+3] = the Dda state a sample starts its first visit with.
__global__ __launch_bounds__(256) void primary_advance_delta_kernel(DevScene sc, const float4 *__restrict__ primary,
                                                                    float4 *__restrict__ advance)
{
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= sc.width || y >= sc.height) {
        return;
    }
    const size_t p = (size_t)y * sc.width + x;
    const float4 p0 = primary[2 * p], p1 = primary[2 * p + 1];
    const f3 pos = mk3(p0.x, p0.y, p0.z), dir = mk3(p1.x, p1.y, p1.z);
    Dda d{};
    if (p0.w != 0.f && in_box(sc, pos)) {
        dda_begin(sc, d, pos, dir);
        while (cell_in_grid(sc, d) && sc.maj_cells[cell_index(sc, d)] == 0u) {
            dda_cross(d, dir); // (every crossing moves one cell: the walk ends at the grid's faces at the latest)
        }
    }
    advance[4 * p] = make_float4(d.org.x, d.org.y, d.org.z, d.t);
    advance[4 * p + 1] = make_float4(d.tmax.x, d.tmax.y, d.tmax.z, __int_as_float(d.bx));
    advance[4 * p + 2] = make_float4(d.tdelta.x, d.tdelta.y, d.tdelta.z, __int_as_float(d.by));
    advance[4 * p + 3] = make_float4(__int_as_float(d.bz), 0.f, 0.f, 0.f);   // (cell coordinates of the virtual grid; the walk may have left the stored box)
}

// Majorant of every STORED cell of the DELTA grid: the max of the texels [lo-1, lo+C+1]^3 (clamped), and the 2-bit code
// of their min (q = min(3, 4*min/max)); one block per cell, cell (cx, cy, cz) = virtual cell origin + (cx, cy, cz)
// (orc_build_majorants in the oracle).
__global__ __launch_bounds__(256) void majorant_cells_kernel(const uint8_t *__restrict__ texels, int nx, int ny, int nz, int bias,
                                                             int C, int ox, int oy, int oz, int gx, int gy, uint8_t *__restrict__ out,
                                                             uint8_t *__restrict__ out_codes)
{
    const int cx = blockIdx.x, cy = blockIdx.y, cz = blockIdx.z;
    const int w = C + 3;
    const int x0 = C * (cx + ox) - bias - 1, y0 = C * (cy + oy) - bias - 1, z0 = C * (cz + oz) - bias - 1;
    uint32_t m = 0, lo = 255;
    for (int i = threadIdx.x; i < w * w * w; i += 256) {
        const int lx = i % w, ly = (i / w) % w, lz = i / (w * w);
        const int x = min(max(x0 + lx, 0), nx - 1), y = min(max(y0 + ly, 0), ny - 1), z = min(max(z0 + lz, 0), nz - 1);
        const uint32_t v = texels[((size_t)z * ny + y) * nx + x];
        m = max(m, v);
        lo = min(lo, v);
    }
    __shared__ uint32_t red_max, red_min;
    if (threadIdx.x == 0) {
        red_max = 0;
        red_min = 255;
    }
    __syncthreads();
    atomicMax(&red_max, m);
    atomicMin(&red_min, lo);
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t cell = ((size_t)cz * gy + cy) * gx + cx;
        out[cell] = (uint8_t)red_max;
        out_codes[cell] = (uint8_t)(red_max ? min(3u, 4u * red_min / red_max) : 0u);
    }
}

hipError_t launch_majorant_cells(const uint8_t *texels, int nx, int ny, int nz, int bias, int cell, const int origin[3], int gx, int gy, int gz,
                                 uint8_t *out, uint8_t *out_codes, hipStream_t stream)
{
    hipLaunchKernelGGL(majorant_cells_kernel, dim3(gx, gy, gz), dim3(256), 0, stream, texels, nx, ny, nz, bias, cell, origin[0], origin[1],
                       origin[2], gx, gy, out, out_codes);
    return hipGetLastError();
}

hipError_t launch_primary_advance_delta(const DevScene &sc, const float4 *primary, float4 *advance, hipStream_t stream)
{
    const dim3 grid((sc.width + 31) / 32, (sc.height + 7) / 8), block(256);
    hipLaunchKernelGGL(primary_advance_delta_kernel, grid, block, 0, stream, sc, primary, advance);
    return hipGetLastError();
}

// NEE selects where a collision's two lookups come from (same values, same radiance -- only the loads differ):
//   0  density from the 4^3 apron bricks when the collision is drawn, the shadow volume from its own apron bricks in the
//      scatter phase (rounds 1-3);
//   1  the same two arrays, but the shadow-volume footprint is REQUESTED in the tracking visit, as soon as the collision is
//      known to be real, and consumed in the scatter phase: its miss runs beside the rest of the visit instead of in front
//      of the bounce;
//   2  twin bricks (DevScene::tbricks): both footprints in one 128-byte line, the shadow half requested like in 1 -- an L1/L2
//      hit on the line the density lookup has just brought, or that line's one fill when the lower bound made the lookup
//      unnecessary.
// INTERIOR: DevScene::delta_interior as a compile-time fact (the box test of a real collision and its three scene constants are
// not in the kernel at all; as a run-time test it cost 3.4 %: the kernel spills scalar registers).  Instantiated for NEE = 1 and 2.
template <int MODE, bool STATS, int NEE, bool INTERIOR = false>
__global__ __launch_bounds__(kDeltaThreads) __attribute__((amdgpu_waves_per_eu(CT_DELTA_WAVES))) void render_delta_kernel(DevScene sc, BatchArgs ba)
{
    // 24 KiB of Mie tables + 40 KiB of majorants + 10 KiB of lower-bound codes + 2 KiB per block of 768 threads:
    // two blocks per CU, 6 waves per SIMD
    __shared__ MieLds lds;
    __shared__ uint32_t maj_words[kMajCellsMax / 4];
    {
        const uint32_t words = ((uint32_t)(sc.mc_gx * sc.mc_gy * sc.mc_gz) + 3u) >> 2; // (the array is padded to whole words)
        const uint32_t *src = (const uint32_t *)sc.maj_cells;
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
            maj_words[i] = src[i];
        }
    }
    const uint8_t *lds_maj = (const uint8_t *)maj_words;
    // the cells' lower-bound codes, 2 bits each (the array in memory holds one per byte, padded to whole words)
    __shared__ uint8_t lds_codes[kMajCellsMax / 4];
    {
        const uint32_t words = ((uint32_t)(sc.mc_gx * sc.mc_gy * sc.mc_gz) + 3u) >> 2;
        const uint32_t *src = (const uint32_t *)sc.maj_codes;
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
            const uint32_t w = src[i];
            lds_codes[i] = (uint8_t)((w & 3u) | ((w >> 6) & 0xcu) | ((w >> 12) & 0x30u) | ((w >> 18) & 0xc0u));
        }
    }
    // sigma_bar and 1/sigma_bar of every majorant value (delta_flight in the oracle computes the same two floats)
    __shared__ float2 sigma_table[256];
    if (threadIdx.x < 256u) {
        const float sb = ((float)threadIdx.x * (1.0f / 255.0f)) * sc.density_multiplier;
        sigma_table[threadIdx.x] = make_float2(sb, 1.0f / sb);
    }
    load_tables(sc, lds);

    const uint32_t lane = threadIdx.x & 63u;
    f3 pos = mk3(0, 0, 0), dir = mk3(0, 0, 1), rad = mk3(0, 0, 0);
    Dda dda{};
    uint32_t seed = 0, depth = 0, out_idx = 0;
    int state = ST_IDLE;
    RawCell nee_raw{};   // NEE != 0: the shadow-volume footprint of the collision this lane waits to scatter at (state == ST_BOUNCE)

    JobState job{ 0, 0, 0, 0, 0, 0 };   // the current job and the samples left in it
    bool left_done = false;
    uint32_t q_cur = (uint32_t)kQueues, q_tried = 0;
    bool drained = false;
    uint32_t c_dl = 0, c_il = 0, c_cap = 0;
    uint32_t st_regen = 0, st_regen_l = 0, st_march = 0, st_march_l = 0, st_scat = 0, st_scat_l = 0;
    uint32_t st_fetch = 0, st_zero = 0, st_skip = 0, st_empty = 0;   // (st_empty: crossings of cells whose majorant is zero)
    uint32_t iv_dealt = 0, iv_written = 0, iv_resumed = 0, iv_suspended = 0; // path conservation (STATS kernels)
    uint32_t st_hist_t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, st_hist_s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // visits by lanes taking part (bins of 8)
    uint32_t st_drained_t = 0;                                                                          // tracking visits after the wave found the job list empty

    // ---------------- path continuation, as in render_persistent_kernel (kContWordsDelta words per path) ----------------
    uint32_t age = 0;   // times this lane's path has been suspended (see render_persistent_kernel)
    if (ba.cont_in) {
        const uint32_t total = __builtin_amdgcn_readfirstlane(*ba.cont_in_count);
        uint32_t base = 0;
        if (lane == 0) {
            base = atomicAdd(ba.cont_cursor, 64u);
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + lane < total) {
            const uint4 *e = (const uint4 *)(ba.cont_in + (size_t)(base + lane) * kContWordsDelta);
            const uint4 w0 = e[0], w1 = e[1], w2 = e[2], w3 = e[3], w4 = e[4], w5 = e[5], w6 = e[6];
            pos = mk3(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z));
            dir = mk3(__uint_as_float(w0.w), __uint_as_float(w1.x), __uint_as_float(w1.y));
            rad = mk3(__uint_as_float(w1.z), __uint_as_float(w1.w), __uint_as_float(w2.x));
            seed = w2.y;
            out_idx = w2.z;
            depth = w2.w & 0xffffu;
            state = (int)(w2.w >> 24);
            dda.org = mk3(__uint_as_float(w3.x), __uint_as_float(w3.y), __uint_as_float(w3.z));
            dda.t = __uint_as_float(w3.w);
            dda.tmax = mk3(__uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z));
            dda.bx = (int32_t)w4.w;
            dda.tdelta = mk3(__uint_as_float(w5.x), __uint_as_float(w5.y), __uint_as_float(w5.z));
            dda.by = (int32_t)w5.w;
            dda.bz = (int32_t)w6.x;
            age = w6.y;
            if (STATS) {
                iv_resumed += 1;
            }
            if (NEE != 0 && state == ST_BOUNCE) {
                // suspended between its collision and its bounce: the footprint the tracking visit had requested went with the launch
                nee_raw = (NEE == 1) ? load_raw_apron(sc.ibricks + apron_offset_in_grid(sc, pos))
                                     : load_raw_twin(sc.tbricks + twin_offset_in_grid(sc, pos) + 64);
            }
        }
    }
    const bool may_suspend = ba.cont_out != nullptr;
    // One global list (queue 0 holds everything): the waves look at the "list empty" flag now and then and hand their paths
    // on as soon as it is up.  With per-XCD queues -- short launches -- they do NOT: a wave then learns that nothing is left
    // when it runs out of work itself, and finishes what it holds meanwhile.  Round 3 tried the flag there too (it is raised
    // when the last job of EVERY queue has been taken, take_job): the waves hand on 10 % more paths, more of them grow too old
    // to be handed on again, and a 10-subframe launch takes 4.39 ms with a ring of 13 regions and 5.04 with the usual 6,
    // against 4.34 without (tools/launch_timeline.py, profiles/r03y): the 0.9 ms in which such a launch drains are not idle
    // time -- the waves that are left run the faster for being fewer.
    const bool single_queue = ba.q_begin[1] == ba.n_jobs;
    // one global list: only then does "the last job has been taken" (the flag) mean that every queue is empty
    uint32_t visit = 0;

    for (;;) {
        visit += 1;
        if (may_suspend && single_queue && !drained && sc.hint_period != 0u && (visit & (sc.hint_period - 1u)) == 0u) {
            const uint32_t empty = __builtin_amdgcn_readfirstlane(__atomic_load_n(ba.queue + kQueueFlag, __ATOMIC_RELAXED));
            if (empty != 0u) {
                // (samples of its own job that this wave has not started go to the next launch with its paths)
                if (job.next != job.end && ba.left_out && job.age < ba.max_age && hand_on_job(ba, lane, job)) {
                    job.next = job.end;
                }
                if (job.next == job.end) {
                    drained = true;
                }
            }
        }
        // ---------------- regenerate ----------------
        const uint64_t idle = __builtin_amdgcn_ballot_w64(state == ST_IDLE);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
        if ((n_idle >= sc.regen_min || n_idle == 64u) && !(drained && job.next == job.end)) {
            if (job.next == job.end && !take_leftover(ba, lane, left_done, job)) {
                uint32_t j = 0;
                if (!take_job(ba, lane, q_cur, q_tried, j)) {
                    drained = true;
                } else {
                    const uint32_t sub = __builtin_amdgcn_readfirstlane(ba.job_sub[j]);
                    const uint32_t s0 = sub & 0xffffu;
                    job.g = __builtin_amdgcn_readfirstlane(ba.job_group[j]);
                    job.next = 0;
                    // the list may have been built for a larger batch: clip the job to this launch's subframes
                    job.end = (s0 < ba.S ? min(sub >> 16, ba.S - s0) : 0u) * 64u;
                    job.base = ba.out_offset + s0 * ba.frame_stride + group_column(ba, job.g);
                    job.first = ba.first_subframe + s0;
                    job.age = 0;
                }
            }
            if (job.next != job.end) {
                const uint32_t avail = job.end - job.next;
                if (STATS) {
                    st_regen += 1;
                    st_regen_l += min(n_idle, avail);
                }
                const uint32_t rank = lane_rank(idle);
                const bool take = (state == ST_IDLE) && rank < avail;
                const uint32_t q = job.next + rank;
                job.next += min(n_idle, avail);
                if (take) {
                    const uint32_t s = q >> 6, l = q & 63u;   // (subframe within the job, lane)
                    const uint32_t g = job.g;
                    const uint32_t pixel = ba.pixels[g * 64u + l];
                    if (pixel != 0xffffffffu) {
                        if (STATS) {
                            iv_dealt += 1;
                        }
                        const float4 p0 = ba.primary[2 * (size_t)pixel];
                        const float4 p1 = ba.primary[2 * (size_t)pixel + 1];
                        out_idx = ba.frame_stride ? job.base + s * ba.frame_stride + l : pixel;
                        pos = mk3(p0.x, p0.y, p0.z);
                        const bool hit = p0.w != 0.f;
                        dir = mk3(p1.x, p1.y, p1.z);
                        seed = tea4(__float_as_uint(p1.w), job.first + s);
                        rad = mk3(0, 0, 0);
                        depth = 0;
                        age = job.age;
                        if (MODE == 1) {
                            dir = new_direction(lds.cdf, lds.guide, seed, dir);
                        }
                        bool go = hit && in_box(sc, pos);
                        if (MODE != 2 && go) {
                            depth = 1;
                            if (depth == sc.max_depth) {
                                c_cap += 1;
                                go = false;
                            }
                        }
                        if (go) {
                            if (MODE != 1 && ba.advance) {
                                // the pixel's pre-walked DDA prefix (primary_advance_delta_kernel)
                                const float4 a0 = ba.advance[4 * (size_t)pixel], a1 = ba.advance[4 * (size_t)pixel + 1];
                                const float4 a2 = ba.advance[4 * (size_t)pixel + 2], a3 = ba.advance[4 * (size_t)pixel + 3];
                                dda.org = mk3(a0.x, a0.y, a0.z);
                                dda.t = a0.w;
                                dda.tmax = mk3(a1.x, a1.y, a1.z);
                                dda.bx = __float_as_int(a1.w);
                                dda.tdelta = mk3(a2.x, a2.y, a2.z);
                                dda.by = __float_as_int(a2.w);
                                dda.bz = __float_as_int(a3.x);
                            } else {
                                dda_begin(sc, dda, pos, dir);
                            }
                            if (DELTA_END_MERGE && !dda_enter_box(sc, dda)) {
                                // the flight never meets the stored box: what its first visit would have written
                                ba.frames[out_idx] = make_float4(rad.x, rad.y, rad.z, 1.f);
                                if (STATS) {
                                    iv_written += 1;
                                }
                                if (!INTERIOR && ba.cost) {   // (the interior kernel serves the launches that record no path costs)
                                    ba.cost[out_idx - ba.out_offset] = make_uint2(depth, depth);
                                }
                            } else {
                                state = ST_MARCH;
                            }
                        } else {
#ifdef CT_DEBUG_BOUNDS
                            if (age == 0u && ba.frame_stride && out_idx - ba.out_offset >= ba.S * ba.frame_stride) {   // (a resumed path writes to an earlier batch's region)
                                printf("CT_DEBUG_BOUNDS frames[%u] outside the batch (offset %u, S %u, stride %u)\n", out_idx, ba.out_offset, ba.S, ba.frame_stride);
                                out_idx = ba.out_offset;
                            }
#endif
                            ba.frames[out_idx] = make_float4(0.f, 0.f, 0.f, 1.f);
                            if (STATS) {
                                iv_written += 1;
                            }
                        }
                    }
                }
            }
        }

        const uint64_t marching = __builtin_amdgcn_ballot_w64(state == ST_MARCH);
        const uint64_t bouncing = __builtin_amdgcn_ballot_w64(state == ST_BOUNCE);
        const uint32_t nm = (uint32_t)__builtin_popcountll(marching);
        const uint32_t nb = (uint32_t)__builtin_popcountll(bouncing);
        if ((marching | bouncing) == 0 && drained && job.next == job.end) {
            break;
        }
        // (everything idle but samples left: both phases below are no-ops and the loop regenerates)

        bool finished = false;
        const bool do_scatter = nb != 0 && (nb >= sc.scatter_min || nm == 0);
        if (do_scatter) {
            // ---------------- scatter ----------------
            if (STATS) {
                st_scat += 1;
                st_scat_l += nb;
                st_hist_s[min((nb - 1u) >> 3, 7u)] += 1;
            }
            if (state == ST_BOUNCE) {
                const bool chopped = (MODE == 1) ? true : (MODE == 0 ? (depth != 1) : false);
                NeeLoads nee;
                if (NEE == 0) {
                    nee = in_scattering_issue(sc, pos, dir, chopped);
                } else {
                    nee = in_scattering_issue_phase(sc, dir, chopped);
                }
                c_il += 1;
                bool go = (MODE != 2);
                if (go) {
                    dir = new_direction(lds.cdf, lds.guide, seed, dir);
                    depth++;
                    if (depth == sc.max_depth) {
                        c_cap += 1;
                        go = false;
                    }
                }
                if (NEE != 0) {
                    nee.cell = (NEE == 1) ? combine_apron(nee_raw) : combine_twin(nee_raw);
                }
                rad = add3(rad, in_scattering_finish(sc, nee, pos));
                if (go) {
                    dda_begin(sc, dda, pos, dir);
                    if (DELTA_END_MERGE && !dda_enter_box(sc, dda)) {
                        finished = true;   // (the new flight never meets the stored box)
                    } else {
                        state = ST_MARCH;
                    }
                } else {
                    finished = true;
                }
            }
        }
        // A scatter phase is followed by the tracking burst in the SAME scheduler iteration -- the lanes it has just redirected
        // march at once -- instead of a pass through the loop's top in between (round 4: the scheduler's own instructions are a
        // seventh of the kernel's, and this removes a third of its iterations: 5015 -> 5205 Msamples/s, profiles/r04h; DELTA_FUSE
        // = 0 restores the separate iterations).  Schedule only: a path's arithmetic does not know when it runs.
        bool track = !do_scatter;
        uint32_t nm_track = nm;
        if (do_scatter && DELTA_FUSE) {
            nm_track = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_MARCH));
            track = nm_track != 0u;
        }
        if (track) {
            // ---------------- tracking visits: a burst, like the march bursts of render_persistent_kernel ----------------
            if (STATS) {
                st_march += 1;
                st_march_l += nm_track;
                if (nm_track) {
                    st_hist_t[min((nm_track - 1u) >> 3, 7u)] += 1;
                }
                st_drained_t += drained ? 1u : 0u;
            }
            uint32_t burst = drained ? sc.tail_burst : sc.march_burst;
            for (;;) {
            if (state == ST_MARCH) {
                // one step of the flight: cross into the next cell, or draw a tentative collision in this one
                bool ended = false, collide = false;
                float sigma_bar = 0.0f, sigma_low = 0.0f;
                // (DELTA_END_MERGE: a marching lane's cell is always a stored one -- the crossing that leaves the box ends the flight
                // in its own visit, and a flight that begins outside never starts marching -- so the visit begins with the look-up)
                if (!DELTA_END_MERGE && !cell_in_grid(sc, dda)) {
                    ended = true; // left the grid without a collision
                } else {
                    const uint32_t ci = DELTA_END_MERGE ? cell_index_rel(sc, dda) : cell_index(sc, dda);
                    const uint32_t M = lds_maj[ci];
#if CT_DELTA_FLAT_EMPTY
                    {
                        // (an empty cell -- one visit in twelve -- goes through the same instructions with the table's entry 0 and keeps
                        // its seed and its t: no divergent region around the sampling)
                        const uint32_t q = ((uint32_t)lds_codes[ci >> 2] >> ((ci & 3u) * 2u)) & 3u;
                        sigma_low = sigma_table[(q * M) >> 2].x;
                        const float2 sb = sigma_table[M];
                        sigma_bar = sb.x;
                        uint32_t drawn = seed;
                        const float u = u24_to_float(lcg24(drawn));
                        const float dt = -logf_above_one(1.0f - u) * sb.y;
                        const float t_exit = fminf(fminf(dda.tmax.x, dda.tmax.y), dda.tmax.z);
                        const float t_next = dda.t + dt;
                        collide = (M != 0u) & (t_next < t_exit);
                        seed = (M != 0u) ? drawn : seed;
                        dda.t = collide ? t_next : dda.t;
                    }
#else
                    if (M != 0u) {
                        // (requesting this byte WITH the majorant, before it is known to be non-zero, measured: no gain, profiles/r04aa)
                        const uint32_t q = ((uint32_t)lds_codes[ci >> 2] >> ((ci & 3u) * 2u)) & 3u;
                        sigma_low = sigma_table[(q * M) >> 2].x;
                        const float2 sb = sigma_table[M];
                        sigma_bar = sb.x;
                        const float u = u24_to_float(lcg24(seed));
                        const float dt = -logf_above_one(1.0f - u) * sb.y;
                        const float t_exit = fminf(fminf(dda.tmax.x, dda.tmax.y), dda.tmax.z);
                        const float t_next = dda.t + dt;
                        collide = t_next < t_exit;
                        dda.t = collide ? t_next : dda.t;
                    }
#endif
                    if (!collide) {
                        dda_cross(dda, dir);
                        if (DELTA_END_MERGE) {
                            ended = !cell_in_grid_rel(sc, dda);
                        }
                        if (STATS) {
                            st_skip += 1;
                            st_empty += (M == 0u) ? 1u : 0u;
                        }
                    }
                }
                if (collide) {
                    const f3 p = mk3(fmaf(dir.x, dda.t, dda.org.x), fmaf(dir.y, dda.t, dda.org.y), fmaf(dir.z, dda.t, dda.org.z));
                    const float z = u24_to_float(lcg24(seed));
                    // sigma(p) >= sigma_low throughout the cell: below it the collision is real without a lookup
                    bool real = z * sigma_bar < sigma_low;
                    // (NEE != 0: where the collision's footprints are -- the same offset in the density's and the shadow volume's
                    // apron bricks, or the density half of the twin line; inside the box the clamp to the grid changes nothing)
                    size_t off = 0;
                    if (NEE == 1) {
                        off = INTERIOR ? apron_offset_unclamped(sc, p) : apron_offset_in_grid(sc, p);
                    } else if (NEE == 2) {
                        off = INTERIOR ? twin_offset_unclamped(sc, p) : twin_offset_in_grid(sc, p);
                    }
                    if (!real) {
                        uint2 cell;
                        if (NEE == 0) {
                            uint32_t meta_unused;
                            cell = fetch_cell_in_grid(sc, sc.dbricks, p, meta_unused);
                        } else if (NEE == 1) {
                            cell = combine_apron(load_raw_apron(sc.dbricks + off));
                        } else {
                            cell = combine_twin(load_raw_twin(sc.tbricks + off));
                        }
                        c_dl += 1;
                        if (STATS) {
                            st_fetch += 1;
                            st_zero += ((cell.x | cell.y) == 0u) ? 1u : 0u;
                            if (ba.touched_density) {   // ct_debug_track_lines
                                const uint32_t line = (uint32_t)((NEE == 2 ? twin_offset_in_grid(sc, p) : apron_offset_in_grid(sc, p)) >> 7);
                                atomicOr(&ba.touched_density[line >> 5], 1u << (line & 31u));
                            }
                        }
                        real = z * sigma_bar < filter_at(sc, cell, p) * sc.density_multiplier;
                    }
                    // (one divergent region instead of two nested ones: the box test runs under the collision's mask -- the same
                    // wave instructions as under the real collisions' -- and the position is a select)
                    const bool bounce = INTERIOR ? real : (real & in_box_flat(sc, p));
                    ended = ended | (real & !bounce);
                    pos = mk3(real ? p.x : pos.x, real ? p.y : pos.y, real ? p.z : pos.z);
                    if (bounce) {
                        state = ST_BOUNCE;
                        if (STATS && ba.touched_shadow) {   // (twin bricks: the shadow half lies in the density's line)
                            const uint32_t line = (uint32_t)((NEE == 2 ? twin_offset_in_grid(sc, p) : apron_offset_in_grid(sc, p)) >> 7);
                            atomicOr(&(NEE == 2 ? ba.touched_density : ba.touched_shadow)[line >> 5], 1u << (line & 31u));
                        }
                        if (NEE == 1) {
                            nee_raw = load_raw_apron(sc.ibricks + off);
                        } else if (NEE == 2) {
                            nee_raw = load_raw_twin(sc.tbricks + off + 64);
                        }
                    }
                }
                if (ended) {
#ifdef CT_DEBUG_BOUNDS
                    if (age == 0u && ba.frame_stride && out_idx - ba.out_offset >= ba.S * ba.frame_stride) {   // (a resumed path writes to an earlier batch's region)
                        printf("CT_DEBUG_BOUNDS frames[%u] outside the batch (offset %u, S %u, stride %u)\n", out_idx, ba.out_offset, ba.S, ba.frame_stride);
                        out_idx = ba.out_offset;
                    }
#endif
                    ba.frames[out_idx] = make_float4(rad.x, rad.y, rad.z, 1.f);
                    if (STATS) {
                        iv_written += 1;
                    }
                    if (!INTERIOR && ba.cost) {   // (the interior kernel serves the launches that record no path costs)
                        ba.cost[out_idx - ba.out_offset] = make_uint2(depth, depth);
                    }
                    state = ST_IDLE;
                }
            }
            if (--burst == 0u) {
                break;
            }
            // (CT_DELTA_CHECK_EVERY = 2: the burst's end conditions are looked at after every second visit only -- with bursts of
            // two, never: a visit without marching lanes is a no-op, and after one visit of a full wave 26 lanes wait for the
            // scatter phase, not the 48 that end a burst)
            if (DELTA_CHECK_EVERY > 1 && !STATS && (burst % DELTA_CHECK_EVERY) != 0u) {
                continue;
            }
            const uint32_t m_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_MARCH));
            const uint32_t b_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == ST_BOUNCE));
            if ((DELTA_CHECK_EVERY == 1 || (burst % DELTA_CHECK_EVERY) == 0u) &&
                (m_now == 0u || b_now >= sc.burst_scatter || 64u - m_now - b_now >= sc.burst_idle)) {
                break;
            }
            if (STATS) {
                st_march += 1;
                st_march_l += m_now;
                if (m_now) {
                    st_hist_t[min((m_now - 1u) >> 3, 7u)] += 1;
                }
                st_drained_t += drained ? 1u : 0u;
            }
            }
        }
        if (finished) {
#ifdef CT_DEBUG_BOUNDS
            if (age == 0u && ba.frame_stride && out_idx - ba.out_offset >= ba.S * ba.frame_stride) {   // (a resumed path writes to an earlier batch's region)
                printf("CT_DEBUG_BOUNDS frames[%u] outside the batch (offset %u, S %u, stride %u)\n", out_idx, ba.out_offset, ba.S, ba.frame_stride);
                out_idx = ba.out_offset;
            }
#endif
            ba.frames[out_idx] = make_float4(rad.x, rad.y, rad.z, 1.f);
            if (STATS) {
                iv_written += 1;
            }
            if (!INTERIOR && ba.cost) {   // (the interior kernel serves the launches that record no path costs)
                ba.cost[out_idx - ba.out_offset] = make_uint2(depth, depth);
            }
            state = ST_IDLE;
        }
        // ---------------- suspend (see render_persistent_kernel) ----------------
        if (may_suspend && drained && job.next == job.end) {
            const bool mine = state != ST_IDLE && age < ba.max_age;
            const uint64_t live = __builtin_amdgcn_ballot_w64(mine);
            if (live != 0ull) {
                const uint32_t n = (uint32_t)__builtin_popcountll(live);
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(live)) {
                    base = atomicAdd(ba.cont_out_count, n);
                    if (base + n > ba.cont_capacity) {
                        atomicSub(ba.cont_out_count, n);
                        base = 0xffffffffu;
                    } else if (ba.cont_total) {
                        atomicAdd(ba.cont_total, (unsigned long long)n);
                    }
                }
                base = __builtin_amdgcn_readlane(base, __builtin_ctzll(live));
                if (mine && base != 0xffffffffu) {
                    uint4 *e = (uint4 *)(ba.cont_out + (size_t)(base + lane_rank(live)) * kContWordsDelta);
                    e[0] = make_uint4(__float_as_uint(pos.x), __float_as_uint(pos.y), __float_as_uint(pos.z), __float_as_uint(dir.x));
                    e[1] = make_uint4(__float_as_uint(dir.y), __float_as_uint(dir.z), __float_as_uint(rad.x), __float_as_uint(rad.y));
                    e[2] = make_uint4(__float_as_uint(rad.z), seed, out_idx, depth | ((uint32_t)state << 24));
                    e[3] = make_uint4(__float_as_uint(dda.org.x), __float_as_uint(dda.org.y), __float_as_uint(dda.org.z), __float_as_uint(dda.t));
                    e[4] = make_uint4(__float_as_uint(dda.tmax.x), __float_as_uint(dda.tmax.y), __float_as_uint(dda.tmax.z), (uint32_t)dda.bx);
                    e[5] = make_uint4(__float_as_uint(dda.tdelta.x), __float_as_uint(dda.tdelta.y), __float_as_uint(dda.tdelta.z), (uint32_t)dda.by);
                    e[6] = make_uint4((uint32_t)dda.bz, age + 1u, 0u, 0u);
                    state = ST_IDLE;
                    if (STATS) {
                        iv_suspended += 1;
                    }
                }
            }
        }
    }

    uint32_t vals[3] = { c_dl, c_il, c_cap };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint32_t v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_xor(v, off);
        }
        vals[i] = v;
    }
    if (STATS) {
        uint32_t sv[8] = { st_fetch, st_zero, st_skip, iv_dealt, iv_resumed, iv_written, iv_suspended, st_empty };
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t v = sv[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                v += __shfl_xor(v, off);
            }
            sv[i] = v;
        }
        if (lane == 0) {
            atomicAdd(&ba.stats[0], (unsigned long long)st_regen);
            atomicAdd(&ba.stats[1], (unsigned long long)st_regen_l);
            atomicAdd(&ba.stats[2], (unsigned long long)st_march);
            atomicAdd(&ba.stats[3], (unsigned long long)st_march_l);
            atomicAdd(&ba.stats[4], (unsigned long long)st_scat);
            atomicAdd(&ba.stats[5], (unsigned long long)st_scat_l);
            atomicAdd(&ba.stats[6], (unsigned long long)sv[0]);
            atomicAdd(&ba.stats[7], (unsigned long long)sv[1]);
            atomicAdd(&ba.stats[8], (unsigned long long)sv[2]);
            for (int i = 0; i < 8; i++) {
                atomicAdd(&ba.stats[16 + i], (unsigned long long)st_hist_t[i]);
                atomicAdd(&ba.stats[24 + i], (unsigned long long)st_hist_s[i]);
            }
            atomicAdd(&ba.stats[9], (unsigned long long)sv[7]);
            atomicAdd(&ba.stats[32], (unsigned long long)st_drained_t);
            atomicAdd(&ba.stats[64], (unsigned long long)sv[3]);
            atomicAdd(&ba.stats[65], (unsigned long long)sv[4]);
            atomicAdd(&ba.stats[66], (unsigned long long)sv[5]);
            atomicAdd(&ba.stats[67], (unsigned long long)sv[6]);
        }
    }
    if (lane == 0) {
        atomicAdd(&ba.counters[2], (unsigned long long)vals[0]);
        atomicAdd(&ba.counters[3], (unsigned long long)vals[1]);
        atomicAdd(&ba.counters[4], (unsigned long long)vals[1]);
        atomicAdd(&ba.counters[5], (unsigned long long)vals[2]);
        // this kernel issues one fetch per counted lookup: cells below their lower bound are neither counted nor fetched
        atomicAdd(&ba.counters[6], (unsigned long long)vals[0]);
        atomicAdd(&ba.counters[7], (unsigned long long)vals[1]);
    }
}

template <bool STATS, int NEE>
static void launch_render_delta_mode(const DevScene &sc, const BatchArgs &ba, dim3 grid, dim3 block, hipStream_t stream)
{
    if (NEE != 0 && !STATS && sc.delta_interior != 0u && ba.cost == nullptr) {   // (the launch that measures path costs: the general kernel)
        constexpr int N = NEE != 0 ? NEE : 1;   // (NEE = 0, the round-3 layout kept for A/Bs, has no interior kernel)
        switch (sc.mode) {
        case 0: hipLaunchKernelGGL((render_delta_kernel<0, false, N, true>), grid, block, 0, stream, sc, ba); break;
        case 1: hipLaunchKernelGGL((render_delta_kernel<1, false, N, true>), grid, block, 0, stream, sc, ba); break;
        default: hipLaunchKernelGGL((render_delta_kernel<2, false, N, true>), grid, block, 0, stream, sc, ba); break;
        }
        return;
    }
    switch (sc.mode) {
    case 0: hipLaunchKernelGGL((render_delta_kernel<0, STATS, NEE>), grid, block, 0, stream, sc, ba); break;
    case 1: hipLaunchKernelGGL((render_delta_kernel<1, STATS, NEE>), grid, block, 0, stream, sc, ba); break;
    default: hipLaunchKernelGGL((render_delta_kernel<2, STATS, NEE>), grid, block, 0, stream, sc, ba); break;
    }
}

static void launch_render_delta_nee(const DevScene &sc, const BatchArgs &ba, dim3 grid, dim3 block, hipStream_t stream, bool stats)
{
    // (DevScene::delta_nee: 2 needs the twin bricks, which ct_create builds only when it chose that layout)
    const int nee = sc.delta_nee == 2u && sc.tbricks ? 2 : (sc.delta_nee == 1u ? 1 : 0);
    if (stats) {
        switch (nee) {
        case 0: launch_render_delta_mode<true, 0>(sc, ba, grid, block, stream); break;
        case 1: launch_render_delta_mode<true, 1>(sc, ba, grid, block, stream); break;
        default: launch_render_delta_mode<true, 2>(sc, ba, grid, block, stream); break;
        }
    } else {
        switch (nee) {
        case 0: launch_render_delta_mode<false, 0>(sc, ba, grid, block, stream); break;
        case 1: launch_render_delta_mode<false, 1>(sc, ba, grid, block, stream); break;
        default: launch_render_delta_mode<false, 2>(sc, ba, grid, block, stream); break;
        }
    }
}

hipError_t launch_render_delta(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream)
{
    const dim3 grid(shape.blocks), block(shape.threads);
    launch_render_delta_nee(sc, ba, grid, block, stream, shape.stats);
    return hipGetLastError();
}

// The measured-and-rejected experiments (two DELTA kernels that regroup paths by phase: DESIGN.md 4.2 "Round 3") are not part
// of the product's library: -DCT_EXPERIMENTS (python -m deepestscatter_amd.build --variant exp -> libcloudtrace_exp.so).
#ifdef CT_EXPERIMENTS
#include "ct_exchange.hpp"
#endif

LaunchShape persistent_shape(int device, bool delta, int blocks_per_cu)
{
    hipDeviceProp_t prop;
    LaunchShape s{ 1024, delta ? kDeltaThreads : 512, false };
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        // as many blocks per CU as the kernel's registers and LDS admit: MARCH 512 threads / 40 KiB,
        // DELTA 768 threads / 64 KiB
        int per_cu = 0;
        const hipError_t e = delta ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_delta_kernel<0, false, 0>, kDeltaThreads, 0)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_persistent_kernel<0, false, false>, 512, 0);
        if (e != hipSuccess || per_cu < 1) {
            per_cu = delta ? 2 : 3;
        }
        s.blocks = prop.multiProcessorCount * std::min(per_cu, 8);
        if (blocks_per_cu >= 1 && blocks_per_cu <= 8) {   // (CT_BLOCKS_PER_CU: tuning knob for experiments)
            s.blocks = prop.multiProcessorCount * blocks_per_cu;
        }
    }
    return s;
}

template <bool STATS, bool SPARSE>
static void launch_render_persistent_mode(const DevScene &sc, const BatchArgs &ba, dim3 grid, dim3 block, hipStream_t stream)
{
    if (!STATS && !SPARSE && ba.cost == nullptr) {
        switch (sc.mode) {
        case 0: hipLaunchKernelGGL((render_persistent_kernel<0, false, false, false>), grid, block, 0, stream, sc, ba); break;
        case 1: hipLaunchKernelGGL((render_persistent_kernel<1, false, false, false>), grid, block, 0, stream, sc, ba); break;
        default: hipLaunchKernelGGL((render_persistent_kernel<2, false, false, false>), grid, block, 0, stream, sc, ba); break;
        }
        return;
    }
    switch (sc.mode) {
    case 0: hipLaunchKernelGGL((render_persistent_kernel<0, STATS, SPARSE>), grid, block, 0, stream, sc, ba); break;
    case 1: hipLaunchKernelGGL((render_persistent_kernel<1, STATS, SPARSE>), grid, block, 0, stream, sc, ba); break;
    default: hipLaunchKernelGGL((render_persistent_kernel<2, STATS, SPARSE>), grid, block, 0, stream, sc, ba); break;
    }
}

hipError_t launch_render_persistent(const DevScene &sc, const BatchArgs &ba, LaunchShape shape, hipStream_t stream)
{
    const dim3 grid(shape.blocks), block(shape.threads);
    const bool sparse = sc.m_rows != nullptr;
    if (shape.stats) { // diagnostics build of the same kernel
        if (sparse) {
            launch_render_persistent_mode<true, true>(sc, ba, grid, block, stream);
        } else {
            launch_render_persistent_mode<true, false>(sc, ba, grid, block, stream);
        }
    } else if (sparse) {
        launch_render_persistent_mode<false, true>(sc, ba, grid, block, stream);
    } else {
        launch_render_persistent_mode<false, false>(sc, ba, grid, block, stream);
    }
    return hipGetLastError();
}

// One thread per pixel of one subframe, the reference's control flow verbatim (nested loops).
// Kept as an independent second implementation on the GPU (tests cross-check it against the
// persistent kernel and the oracle) and as the divergence baseline in bench_kernels.
__global__ __launch_bounds__(256) void render_simple_kernel(DevScene sc, BatchArgs ba, uint32_t shard_index,
                                                            uint32_t shard_count)
{
    __shared__ MieLds lds;
    load_tables(sc, lds);
    const uint32_t tile = blockIdx.x / 4u; // 4 waves per block, one 8x8 tile per wave
    const uint32_t wave = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const uint32_t tile_id = blockIdx.x * 4u + wave;
    (void)tile;
    if (tile_id >= sc.tiles_x * sc.tiles_y) {
        return;
    }
    const uint32_t ty = tile_id / sc.tiles_x, tx = tile_id - ty * sc.tiles_x;
    if (tile_owner(tx, ty, shard_count) != shard_index) {
        return;
    }
    const uint32_t x = tx * kTile + (l & 7u), y = ty * kTile + (l >> 3);
    if (x >= sc.width || y >= sc.height) {
        return;
    }
    const uint32_t out_idx = y * sc.width + x;
    uint32_t c_hits = 0, c_dl = 0, c_il = 0, c_cap = 0;
    f3 rad = mk3(0, 0, 0);
    const f3 eye = mk3(sc.ex, sc.ey, sc.ez);
    const f3 d = primary_direction(sc, x, y);
    float t_hit;
    if (intersect_box(sc, eye, d, t_hit)) {
        c_hits = 1;
        f3 pos = add3(eye, scale3(d, t_hit));
        pos = add3(pos, scale3(mk3(sc.bx, sc.by, sc.bz), 0.5f));
        f3 dir = normalize3(d);
        uint32_t seed = tea4(x * 4096u + y, ba.first_subframe);
        if (sc.mode == 1) {
            dir = new_direction(lds.cdf, lds.guide, seed, dir);
        }
        uint32_t depth = 0;
        while (in_box(sc, pos)) {
            if (sc.mode != 2) {
                depth++;
                if (depth == sc.max_depth) {
                    c_cap = 1;
                    break;
                }
            }
            const float xi = u24_to_float(lcg24(seed));
            const f3 stepv = scale3(dir, sc.sample_step);
            float T = 1;
            bool scattered = false;
            while (in_box(sc, pos)) {
                pos = add3(pos, stepv);
                const float density = tex3_apron(sc, sc.dbricks, pos) * sc.density_multiplier;
                c_dl++;
                const float extinction = density * sc.sample_step;
                T *= ct_expf(-extinction);
                if (xi > T) {
                    scattered = true;
                    const float lg = ct_logf(xi / T);
                    const float inv = 1.0f / density;
                    pos = sub3(pos, scale3(scale3(dir, lg), inv));
                    break;
                }
            }
            if (!scattered || !in_box(sc, pos)) {
                break;
            }
            const bool chopped = (sc.mode == 1) ? true : (sc.mode == 0 ? (depth != 1) : false);
            rad = add3(rad, in_scattering(sc, pos, dir, chopped));
            c_il++;
            if (sc.mode == 2) {
                break;
            }
            dir = new_direction(lds.cdf, lds.guide, seed, dir);
        }
    }
    ba.frames[out_idx] = make_float4(rad.x, rad.y, rad.z, 1.f);
    atomicAdd(&ba.counters[0], 1ull);
    atomicAdd(&ba.counters[1], (unsigned long long)c_hits);
    atomicAdd(&ba.counters[2], (unsigned long long)c_dl);
    atomicAdd(&ba.counters[3], (unsigned long long)c_il);
    atomicAdd(&ba.counters[4], (unsigned long long)c_il);
    atomicAdd(&ba.counters[5], (unsigned long long)c_cap);
    atomicAdd(&ba.counters[6], (unsigned long long)c_dl); // nothing is skipped or reused here: issued == counted
    atomicAdd(&ba.counters[7], (unsigned long long)c_il);
}

hipError_t launch_render_simple(const DevScene &sc, const BatchArgs &ba, uint32_t shard_index,
                                uint32_t shard_count, hipStream_t stream)
{
    const uint32_t tiles = sc.tiles_x * sc.tiles_y;
    hipLaunchKernelGGL(render_simple_kernel, dim3((tiles + 3) / 4), dim3(256), 0, stream, sc, ba, shard_index,
                       shard_count);
    return hipGetLastError();
}

// =============================================================================================
// scatter-sample generator: generatePoints + firstScatterPosition (dataset tooling, not hot)
// =============================================================================================
// getNextScatteringEvent (cloud.cuh:77-114) as a plain loop; returns true when the flight collided
// and leaves the scatter position (or the exit position) in `pos`.
CT_DEV bool plain_flight(const DevScene &sc, float xi, f3 &pos, f3 dir)
{
    const f3 stepv = scale3(dir, sc.sample_step);
    float T = 1;
    while (in_box(sc, pos)) {
        pos = add3(pos, stepv);
        const float density = tex3_apron(sc, sc.dbricks, pos) * sc.density_multiplier;
        const float extinction = density * sc.sample_step;
        T *= ct_expf(-extinction);
        if (xi > T) {
            const float lg = ct_logf(xi / T);
            const float inv = 1.0f / density;
            pos = sub3(pos, scale3(scale3(dir, lg), inv));
            return true;
        }
    }
    return false;
}

__global__ __launch_bounds__(64) void scatter_samples_kernel(DevScene sc, uint32_t count, uint32_t batch_seed,
                                                            float *__restrict__ positions,
                                                            float *__restrict__ directions)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) {
        return;
    }
    uint32_t seed = tea4(i, batch_seed);
    const float nanv = __uint_as_float(0x7fc00000u);
    f3 out_p = mk3(nanv, nanv, nanv), out_d = mk3(nanv, nanv, nanv); // `clear`, pointGeneratorCamera.cu:44-48
    for (uint32_t attempt = 0; attempt < 4096u; attempt++) {
        // uniformOnSphere, random.cuh:133-148
        const float u = u24_to_float(lcg24(seed));
        const float v = u24_to_float(lcg24(seed));
        const float phi = u * kPi * 2;
        const float cos_theta = 2 * v - 1;
        const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
        float sn, cs;
        ct_sincosf(phi, &sn, &cs);
        const f3 normal = mk3(cs * sin_theta, sn * sin_theta, cos_theta);
        // uniformOnDisc(seed, normal), random.cuh:161-172: (x, 0, y) through Onb(normal)
        const float theta = u24_to_float(lcg24(seed)) * kPi * 2;
        const float sqrt_r = sqrtf(u24_to_float(lcg24(seed)));
        float st, ct;
        ct_sincosf(theta, &st, &ct);
        const float dx = sqrt_r * ct, dy = sqrt_r * st;
        f3 b;
        if (fabsf(normal.x) > fabsf(normal.z)) {
            b = mk3(-normal.y, normal.x, 0.0f);
        } else {
            b = mk3(0.0f, -normal.z, normal.y);
        }
        b = normalize3(b);
        const f3 tg = cross3(b, normal);
        const f3 disc = add3(add3(scale3(tg, dx), scale3(b, 0.0f)), scale3(normal, dy));
        const float disc_radius = sqrtf(3.0f) / 2;
        const f3 position = scale3(disc, disc_radius);
        const f3 origin = add3(position, scale3(normal, 2.0f));
        const f3 rdir = mk3(-normal.x, -normal.y, -normal.z);
        float t_hit;
        if (!intersect_box(sc, origin, rdir, t_hit)) {
            continue; // the reference's miss program: nothing recorded, try again
        }
        // firstScatterPosition, cloudFirstScatterMaterial.cu:8-29
        f3 pos = add3(origin, scale3(rdir, t_hit));
        pos = add3(pos, scale3(mk3(sc.bx, sc.by, sc.bz), 0.5f));
        const f3 direction = normalize3(rdir);
        uint32_t seed2 = tea4(i * 4096u, batch_seed + attempt);
        const float xi = u24_to_float(lcg24(seed2));
        const bool scattered = plain_flight(sc, xi, pos, direction);
        if (scattered && in_box(sc, pos)) {
            out_p = sub3(pos, scale3(mk3(sc.bx, sc.by, sc.bz), 0.5f));
            out_d = rdir;
            break;
        }
    }
    positions[3 * (size_t)i + 0] = out_p.x; positions[3 * (size_t)i + 1] = out_p.y; positions[3 * (size_t)i + 2] = out_p.z;
    directions[3 * (size_t)i + 0] = out_d.x; directions[3 * (size_t)i + 1] = out_d.y; directions[3 * (size_t)i + 2] = out_d.z;
}

hipError_t launch_scatter_samples(const DevScene &sc, uint32_t count, uint32_t batch_seed, float *positions,
                                  float *directions, hipStream_t stream)
{
    hipLaunchKernelGGL(scatter_samples_kernel, dim3((count + 63) / 64), dim3(64), 0, stream, sc, count, batch_seed,
                       positions, directions);
    return hipGetLastError();
}

// =============================================================================================
// progressive accumulation (progressive.cu:17-27) of S consecutive subframes, in order.
// Only this shard's pixels are touched; everything else stays exactly 0 so that a sum over
// shards (RCCL) reproduces the single-GPU image bit for bit.
// =============================================================================================
CT_DEV void welford(float4 &mu, float4 &var, float4 nr, uint32_t subframe_id)
{
    const float w = 1.0f / (float)subframe_id;
    float4 nm;
    nm.x = mu.x + (nr.x - mu.x) * w;
    nm.y = mu.y + (nr.y - mu.y) * w;
    nm.z = mu.z + (nr.z - mu.z) * w;
    nm.w = mu.w + (nr.w - mu.w) * w;
    var.x = var.x + (nr.x - mu.x) * (nr.x - nm.x);
    var.y = var.y + (nr.y - mu.y) * (nr.y - nm.y);
    var.z = var.z + (nr.z - mu.z) * (nr.z - nm.z);
    var.w = var.w + (nr.w - mu.w) * (nr.w - nm.w);
    mu = nm;
}

// Dense form: frames[s][y][x] holds every pixel of this shard.
__global__ __launch_bounds__(256) void accumulate_batch_kernel(const float4 *__restrict__ frames,
                                                               float4 *__restrict__ mean, float4 *__restrict__ m2,
                                                               uint32_t first_subframe, uint32_t S, uint32_t width,
                                                               uint32_t height, uint32_t shard_index,
                                                               uint32_t shard_count,
                                                               unsigned long long *__restrict__ bad_samples,
                                                               const uint32_t *__restrict__ frozen)
{
    if (frozen && *frozen != 0u) {
        return;   // the image has converged (converged_freeze_kernel): the running mean stays as it is
    }
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= width || y >= height) {
        return;
    }
    if (tile_owner(x / kTile, y / kTile, shard_count) != shard_index) {
        return;
    }
    const size_t pix = (size_t)y * width + x;
    const size_t plane = (size_t)width * height;
    float4 mu = mean[pix], var = m2[pix];
    uint32_t bad = 0;
    for (uint32_t s = 0; s < S; s++) {
        const float4 v = frames[s * plane + pix];
        bad += (v.w != 1.0f) ? 1u : 0u; // every sample of this path tracer has alpha 1 (cameraCommon.cuh:29)
        welford(mu, var, v, first_subframe + s);
    }
    mean[pix] = mu;
    m2[pix] = var;
    if (bad != 0u && bad_samples) {
        atomicAdd(bad_samples, (unsigned long long)bad);
    }
}

hipError_t launch_accumulate_batch(const float4 *frames, float4 *mean, float4 *m2, uint32_t first_subframe,
                                   uint32_t S, uint32_t width, uint32_t height, uint32_t shard_index,
                                   uint32_t shard_count, unsigned long long *bad_samples, const uint32_t *frozen,
                                   hipStream_t stream)
{
    const dim3 grid((width + 31) / 32, (height + 7) / 8), block(256);
    hipLaunchKernelGGL(accumulate_batch_kernel, grid, block, 0, stream, frames, mean, m2, first_subframe, S, width,
                       height, shard_index, shard_count, bad_samples, frozen);
    return hipGetLastError();
}

// Compact form: one thread per entry of the pixel list (this shard's box-hitting pixels) in the first `list_blocks`
// blocks; behind them (with_misses) one thread per pixel of the frame for this shard's pixels whose primary ray misses the
// box: those are never rendered, their sample is (0,0,0,1) every subframe (empty miss program, progressive.cu:44-46).  One
// dispatch for both: a display update of 10 subframes pays every dispatch it makes.
__global__ __launch_bounds__(256) void accumulate_list_kernel(const float4 *__restrict__ frames, uint32_t frame_stride,
                                                              const uint32_t *__restrict__ pixels, uint32_t n_entries,
                                                              const uint32_t *__restrict__ group_order, uint32_t rank_base,
                                                              float4 *__restrict__ mean, float4 *__restrict__ m2,
                                                              uint32_t first_subframe, uint32_t S,
                                                              unsigned long long *__restrict__ bad_samples,
                                                              uint32_t list_blocks, const float4 *__restrict__ primary,
                                                              uint32_t width, uint32_t height, uint32_t shard_index,
                                                              uint32_t shard_count, const uint32_t *__restrict__ frozen)
{
    if (frozen && *frozen != 0u) {
        return;   // the image has converged (converged_freeze_kernel): the running mean stays as it is
    }
    if (blockIdx.x >= list_blocks) {
        const uint32_t m = blockIdx.x - list_blocks, gx = (width + 31u) / 32u;
        const uint32_t x = (m % gx) * 32u + (threadIdx.x & 31u);
        const uint32_t y = (m / gx) * 8u + (threadIdx.x >> 5);
        if (x >= width || y >= height) {
            return;
        }
        if (tile_owner(x / kTile, y / kTile, shard_count) != shard_index) {
            return;
        }
        const size_t pix = (size_t)y * width + x;
        if (primary[2 * pix].w != 0.f) {
            return;
        }
        float4 mu = mean[pix], var = m2[pix];
        for (uint32_t s = 0; s < S; s++) {
            welford(mu, var, make_float4(0.f, 0.f, 0.f, 1.f), first_subframe + s);
        }
        mean[pix] = mu;
        m2[pix] = var;
        return;
    }
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) {
        return;
    }
    // (entry e of a chunk's scratch row belongs to the group at place rank_base + e / 64 of the job order: BatchArgs::group_rank)
    const uint32_t pix = group_order ? pixels[group_order[rank_base + (e >> 6)] * 64u + (e & 63u)] : pixels[e];
    if (pix == 0xffffffffu) {
        return;
    }
    float4 mu = mean[pix], var = m2[pix];
    // eight loads in flight per thread: a shard of an 8-GPU job has few pixels and many subframes, and one load
    // per trip round the recurrence left it waiting for memory (3.6 ms for 131 k pixels x 4096 subframes, 2.5 now; sixteen are no better)
    // A sample the estimator never wrote (or wrote twice into a neighbour's place) has no alpha of exactly 1: the
    // count goes to ct_debug_invariants, and with CT_DEBUG_INVARIANTS=1 the scratch is filled with NaNs before every
    // launch so that a lost sample cannot pass as the previous batch's.
    uint32_t s = 0, bad = 0;
    for (; s + 8 <= S; s += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            v[k] = frames[(size_t)(s + k) * frame_stride + e];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            bad += (v[k].w != 1.0f) ? 1u : 0u;
            welford(mu, var, v[k], first_subframe + s + k);
        }
    }
    for (; s < S; s++) {
        const float4 v = frames[(size_t)s * frame_stride + e];
        bad += (v.w != 1.0f) ? 1u : 0u;
        welford(mu, var, v, first_subframe + s);
    }
    mean[pix] = mu;
    m2[pix] = var;
    if (bad != 0u && bad_samples) {
        atomicAdd(bad_samples, (unsigned long long)bad);
    }
}

hipError_t launch_accumulate_list(const float4 *frames, uint32_t frame_stride, const uint32_t *pixels,
                                  uint32_t n_entries, const uint32_t *group_order, uint32_t rank_base, bool with_misses,
                                  const float4 *primary, float4 *mean, float4 *m2,
                                  uint32_t first_subframe, uint32_t S, uint32_t width, uint32_t height,
                                  uint32_t shard_index, uint32_t shard_count, unsigned long long *bad_samples,
                                  const uint32_t *frozen, hipStream_t stream)
{
    const uint32_t list_blocks = (n_entries + 255u) / 256u;
    // (the misses once per batch of subframes, whatever number of chunks the pixel groups are rendered in)
    const uint32_t miss_blocks = with_misses ? ((width + 31u) / 32u) * ((height + 7u) / 8u) : 0u;
    if (list_blocks + miss_blocks != 0u) {
        hipLaunchKernelGGL(accumulate_list_kernel, dim3(list_blocks + miss_blocks), dim3(256), 0, stream, frames,
                           frame_stride, pixels, n_entries, group_order, rank_base, mean, m2, first_subframe, S, bad_samples,
                           list_blocks, primary, width, height, shard_index, shard_count, frozen);
    }
    return hipGetLastError();
}

// =============================================================================================
// Reinhard tonemap (reinhard.cu:20-84).  The summation ORDER is part of the result, so the
// column sums run y = 0..H-1 in one thread per column and the final sum runs over the columns
// in order in one lane, exactly like firstPass / secondPass.
// =============================================================================================
CT_DEV float luminance4(float4 c)
{
    return c.x * 0.265068f + c.y * 0.67023428f + c.z * 0.06409157f + c.w * 0.0f;
}

// One launch for firstPass + secondPass + applyReinhard (the reference launches three, Camera.cpp:202-210; BASELINE.json
// asks for a fused epilogue).  Phase 1: the column sums, every column from y = 0 upwards.  Grid barrier (the grid has at most
// one block per CU, so every block is resident or will be without anyone's help).  Phase 2: EVERY block adds the W
// column sums itself, in column order, in one lane -- the same float additions as secondPass's single thread, so no
// second barrier and no broadcast are needed.  Phase 3: applyReinhard over the pixels, grid-stride.
// The barrier counts up for ever: launch n waits for n * gridDim.x arrivals (`barrier_target`), so no launch has to zero it.
constexpr uint32_t kReinhardThreads = 1024, kReinhardCols = 4;

__global__ __launch_bounds__(kReinhardThreads) void reinhard_fused_kernel(const float4 *__restrict__ mean, uint32_t width, uint32_t height,
                                                                          float exposure, float *column_sums, float *__restrict__ avg_out,
                                                                          uint32_t *barrier, uint32_t barrier_target,
                                                                          uchar4 *__restrict__ screen)
{
    extern __shared__ float cols[]; // width floats
    __shared__ float avg_s;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsize = gridDim.x * blockDim.x;
    const float DELTA = 0.00001f;
    // firstPass, reinhard.cu:29-40: per column, sum += luminance + DELTA for y = 0 .. height-1 -- IN THAT ORDER (float adds do
    // not commute with regrouping, and the tonemapped bytes must equal the reference's arithmetic).  A thread per column that
    // loads as it adds is a chain of `height` dependent memory round trips (1.9 ms at 1024^2).  So the loads are taken out of
    // the chain, and spread over the whole chip: a block takes FOUR columns (64 B of every row), its 1024 threads fetch 1024
    // rows of them at once and leave luminance + DELTA in LDS, and one lane per column -- in four different waves -- adds the
    // values in order; only the adds are serial (~3 us per 1024 rows).  (Round 3 until here: 32 columns per block, 64 rows at a
    // time -- 32 busy blocks and 16 dependent round trips at 1024^2, 0.12 ms of a 0.23 ms kernel that a 10-subframe display
    // update pays every time.)
    {
        __shared__ float lum[kReinhardCols][kReinhardThreads];
        const uint32_t groups = (width + kReinhardCols - 1u) / kReinhardCols;
        const uint32_t adder = threadIdx.x >> 6;   // lane 0 of waves 0..3 adds column `adder` of the group
        const bool adds = (threadIdx.x & 63u) == 0u && adder < kReinhardCols;
        for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
            const uint32_t x0 = g * kReinhardCols;
            float sum = 0;
            for (uint32_t y0 = 0; y0 < height; y0 += kReinhardThreads) {
                const uint32_t y = y0 + threadIdx.x;
                float v[kReinhardCols];
#pragma unroll
                for (uint32_t c = 0; c < kReinhardCols; c++) {
                    v[c] = (x0 + c < width && y < height) ? luminance4(mean[(size_t)y * width + x0 + c]) + DELTA : 0.f;
                }
                __syncthreads();   // (the previous rows have been added up)
#pragma unroll
                for (uint32_t c = 0; c < kReinhardCols; c++) {
                    lum[c][threadIdx.x] = v[c];
                }
                __syncthreads();
                if (adds) {
                    const uint32_t rows = min(kReinhardThreads, height - y0);
                    if (rows == kReinhardThreads) {
#pragma unroll 16
                        for (uint32_t r = 0; r < kReinhardThreads; r++) {
                            sum += lum[adder][r];
                        }
                    } else {
                        for (uint32_t r = 0; r < rows; r++) {
                            sum += lum[adder][r];
                        }
                    }
                }
            }
            if (adds && x0 + adder < width) {
                __hip_atomic_store(column_sums + x0 + adder, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(barrier, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // (wrap-safe: the host starts the count again long before 2^31)
        while ((int32_t)(__hip_atomic_load(barrier, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - barrier_target) < 0) {
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) {
        cols[i] = __hip_atomic_load(column_sums + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (from L2, not this CU's L1)
    }
    __syncthreads();
    if (threadIdx.x == 0) { // secondPass, reinhard.cu:44-55
        float result = 0;
        for (uint32_t i = 0; i < width; i++) {
            result += cols[i];
        }
        avg_s = result / (float)(width * height);
        if (blockIdx.x == 0) {
            avg_out[0] = avg_s;
        }
    }
    __syncthreads();
    const float avg = avg_s;
    const uint32_t pixels = width * height;
    for (uint32_t i = gtid; i < pixels; i += gsize) { // applyReinhard, reinhard.cu:59-84
        const float4 color = mean[i];
        const float lw = luminance4(color);
        float ld = lw * exposure / avg;
        ld = ld / (1.f + ld);
        const float sc = ld / lw;
        const float inv_gamma = 1.f / 2.2f;
        // optix clamp(f,0,1) = fmaxf(0, fminf(f,1)): the NaN of black pixels becomes 1
        float r = fmaxf(0.f, fminf(color.x * sc, 1.f));
        float g = fmaxf(0.f, fminf(color.y * sc, 1.f));
        float b = fmaxf(0.f, fminf(color.z * sc, 1.f));
        r = ct_powf(r, inv_gamma) * 255;
        g = ct_powf(g, inv_gamma) * 255;
        b = ct_powf(b, inv_gamma) * 255;
        screen[i] = make_uchar4((unsigned char)r, (unsigned char)g, (unsigned char)b, 255);
    }
}

// `avg` points at two words: the average luminance and the grid barrier's counter.  `generation` (host, one per `avg`)
// counts the launches that have ARRIVED at that counter; 0 = the counter is to be zeroed first.  `device` is the handle's
// device (not the calling thread's current one): the barrier needs every block resident, one per CU at most, and every
// launch on a counter must have the same block count.
hipError_t launch_reinhard(const float4 *mean, uint32_t width, uint32_t height, float exposure, float *column_sums,
                           float *avg, uchar4 *screen, uint32_t *generation, int device, hipStream_t stream)
{
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) {
        cus = 64;
    }
    const uint32_t pixels = width * height;
    const uint32_t blocks = std::min<uint32_t>((uint32_t)cus, (pixels + kReinhardThreads - 1u) / kReinhardThreads);
    if ((size_t)width * sizeof(float) > 48u * 1024u) {
        return hipErrorInvalidValue;   // (the column sums of a row do not fit the kernel's LDS: frames up to 12288 pixels wide)
    }
    uint32_t local_generation = 0;
    uint32_t &gen = generation ? *generation : local_generation;
    if (gen == 0 || (uint64_t)(gen + 1u) * blocks >= 0x40000000ull) {
        const hipError_t e = hipMemsetAsync(avg + 1, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) {
            return e;
        }
        gen = 0;
    }
    hipLaunchKernelGGL(reinhard_fused_kernel, dim3(blocks), dim3(kReinhardThreads), width * sizeof(float), stream, mean, width, height,
                       exposure, column_sums, avg, (uint32_t *)(avg + 1), (gen + 1u) * blocks, screen);
    const hipError_t e = hipGetLastError();
    // A launch that did not happen never arrives at the counter: counting it would make every later tonemap wait in its grid
    // barrier for arrivals that cannot come.  On failure the counter is re-zeroed before the next launch.
    gen = (e == hipSuccess) ? gen + 1u : 0u;
    return e;
}

// =============================================================================================
// Camera::isConverged (Camera.cpp:232-268), channel x only; counts pixels outside the interval.
// =============================================================================================
__global__ __launch_bounds__(1024) void converged_kernel(const float4 *__restrict__ mean, const float4 *__restrict__ m2,
                                                         uint32_t subframe_id, uint64_t pixels, unsigned long long *unconverged)
{
    // (one block per CU and one atomic per block: a wave's worth of pixels per atomic was 16 k atomics on one address, 0.5 ms)
    const float N = (float)subframe_id;
    int bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (uint64_t)gridDim.x * blockDim.x) {
        const float sigma = sqrtf(m2[i].x / N);
        const float abs_ci = 1.96f * sigma / sqrtf(N);
        const float rel_ci = abs_ci / (mean[i].x + 1.1920929e-07f);
        bad += !(rel_ci < 0.02f || abs_ci < 1e-2f) ? 1 : 0;
    }
    __shared__ int block_bad;
    if (threadIdx.x == 0) {
        block_bad = 0;
    }
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_xor(bad, off);
    }
    if ((threadIdx.x & 63u) == 0u && bad != 0) {
        atomicAdd(&block_bad, bad);
    }
    __syncthreads();
    if (threadIdx.x == 0 && block_bad != 0) {
        atomicAdd(unconverged, (unsigned long long)block_bad);
    }
}

static unsigned converged_grid(uint64_t pixels)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        cus = 64;
    }
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus, (pixels + 1023) / 1024));
}

hipError_t launch_converged(const float4 *mean, const float4 *m2, uint32_t subframe_id, uint64_t pixels,
                            unsigned long long *unconverged, hipStream_t stream)
{
    hipLaunchKernelGGL(converged_kernel, dim3(converged_grid(pixels)), dim3(1024), 0, stream, mean, m2, subframe_id, pixels, unconverged);
    return hipGetLastError();
}

// The same test enqueued behind the accumulate kernel of every `cadence`-th subframe (ct_set_stop_when_converged), with the
// reference's decision taken on the device: Camera::render tests isConverged() before every update of 10 subframes and stops
// at the first count N >= 100 with fewer than 500 pixels outside the interval (Camera.cpp:179, 232-268).  The last block to
// finish writes state[0] = N when that holds, and from then on the accumulate kernels leave the running mean alone: the
// buffers keep the reference's final image although the host, which does not wait, has enqueued more.
// state: [0] frozen at N (0 = running)  [1] last count tested  [2] pixels outside the interval then  [4,5] one 64-bit word:
// blocks finished << 32 | pixels outside the interval so far.  ONE atomic per block and a grid of one block per CU: with a
// block per 256 pixels the 8192 atomics on one line took 0.24 ms (35 M/s), twenty times the reading of the two buffers.
__global__ __launch_bounds__(1024) void converged_freeze_kernel(const float4 *__restrict__ mean, const float4 *__restrict__ m2,
                                                                uint32_t subframe_id, uint64_t pixels, uint32_t limit,
                                                                uint32_t *state)
{
    if (__hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        return;   // (set by an earlier kernel of this stream: every block reads the same)
    }
    const float N = (float)subframe_id;
    int bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (uint64_t)gridDim.x * blockDim.x) {
        const float sigma = sqrtf(m2[i].x / N);
        const float abs_ci = 1.96f * sigma / sqrtf(N);
        const float rel_ci = abs_ci / (mean[i].x + 1.1920929e-07f);
        bad += !(rel_ci < 0.02f || abs_ci < 1e-2f) ? 1 : 0;
    }
    __shared__ int block_bad;
    if (threadIdx.x == 0) {
        block_bad = 0;
    }
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_xor(bad, off);
    }
    if ((threadIdx.x & 63u) == 0u && bad != 0) {
        atomicAdd(&block_bad, bad);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *word = (unsigned long long *)(state + 4);
        const unsigned long long old = __hip_atomic_fetch_add(word, (1ull << 32) | (unsigned long long)(uint32_t)block_bad, __ATOMIC_ACQ_REL,
                                                              __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(old >> 32) == gridDim.x - 1u) {
            const uint32_t total = (uint32_t)old + (uint32_t)block_bad;
            __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            state[1] = subframe_id;
            state[2] = total;
            if (total < limit) {
                __hip_atomic_store(state, subframe_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

hipError_t launch_converged_freeze(const float4 *mean, const float4 *m2, uint32_t subframe_id, uint64_t pixels,
                                   uint32_t limit, uint32_t *state, hipStream_t stream)
{
    hipLaunchKernelGGL(converged_freeze_kernel, dim3(converged_grid(pixels)), dim3(1024), 0, stream, mean, m2, subframe_id, pixels, limit,
                       state);
    return hipGetLastError();
}

// =============================================================================================
// self-test hook: k(val) of the guide-table CDF inversion for a range of 24-bit randoms
// =============================================================================================
__global__ void cdf_selftest_kernel(const float *__restrict__ cdf, const uint16_t *__restrict__ guide,
                                    uint32_t first_u24, uint32_t count, uint32_t *__restrict__ k_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) {
        return;
    }
    const float c = sample_cos_theta(cdf, guide, first_u24 + i);
    // cos = (2k+1)/65536 - 1  ->  k
    k_out[i] = ((uint32_t)((c + 1.0f) * 65536.0f) - 1u) >> 1;
}

hipError_t launch_cdf_selftest(const float *cdf, const uint16_t *guide, uint32_t first_u24, uint32_t count,
                               uint32_t *k_out, hipStream_t stream)
{
    hipLaunchKernelGGL(cdf_selftest_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, cdf, guide, first_u24,
                       count, k_out);
    return hipGetLastError();
}

// =============================================================================================
// self-test hook: rcp_moderate / sqrt_moderate against the IEEE operations for EVERY float of their range
// =============================================================================================
// out[0] = inputs tested, out[1] = mismatches, out[2] = smallest mismatching bit pattern (0xffffffff: none)
__global__ void math_selftest_kernel(int which, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out)
{
    unsigned long long tested = 0, bad = 0;
    uint32_t first = 0xffffffffu;
    const uint64_t total = (uint64_t)hi_bits - lo_bits + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = lo_bits + (uint32_t)i;
        const float x = __uint_as_float(bits);
        float got, want;
        if (which == 0) {
            got = rcp_moderate(x);
            want = 1.0f / x;
        } else {
            got = sqrt_moderate(x);
            want = sqrtf(x);
        }
        tested += 1;
        if (__float_as_uint(got) != __float_as_uint(want)) {
            bad += 1;
            first = min(first, bits);
        }
    }
    atomicAdd(&out[0], tested);
    if (bad) {
        atomicAdd(&out[1], bad);
        atomicMin(&out[2], (unsigned long long)first);
    }
}

hipError_t launch_math_selftest(int which, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(math_selftest_kernel, dim3(4096), dim3(256), 0, stream, which, lo_bits, hi_bits, out);
    return hipGetLastError();
}

// =============================================================================================
// PMC calibration probe: the estimator's memory access shape on a buffer far larger than the caches
// =============================================================================================
__global__ void fetch_probe_kernel(const uint8_t *__restrict__ buf, uint32_t line_mask, uint32_t magic,
                                   uint32_t second_offset, unsigned long long *sum)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t line = (i * 2654435761u + 12345u) & line_mask; // odd multiplier: a bijection on 2^k lines
    const uint8_t *p = buf + ((size_t)line << 7) + 13;
    uint2 a, c;
    __builtin_memcpy(&a, p, 8);
    __builtin_memcpy(&c, p + second_offset, 8);
    const uint32_t v = a.x ^ a.y ^ c.x ^ c.y;
    if (v == magic) { // a run-time value the zeroed buffer never produces; keeps the loads alive
        atomicAdd(sum, 1ull);
    }
}

// Working-set variant: lane i reads line hash(i, salt) mod ws_lines (multiply-shift range reduction: uniform over any count).
__global__ void fetch_probe_ws_kernel(const uint8_t *__restrict__ buf, uint32_t ws_lines, uint32_t salt, uint32_t magic,
                                      unsigned long long *sum)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t hsh = (i ^ (salt * 0x9e3779b9u)) * 2654435761u;
    hsh ^= hsh >> 15;
    hsh *= 2246822519u;
    hsh ^= hsh >> 13;
    const uint32_t line = (uint32_t)(((uint64_t)hsh * ws_lines) >> 32);
    const uint8_t *p = buf + ((size_t)line << 7) + 13;
    uint2 a, c;
    __builtin_memcpy(&a, p, 8);
    __builtin_memcpy(&c, p + 25, 8);
    const uint32_t v = a.x ^ a.y ^ c.x ^ c.y;
    if (v == magic) {
        atomicAdd(sum, 1ull);
    }
}

hipError_t launch_fetch_probe_ws(const uint8_t *buf, uint32_t log2_threads, uint32_t ws_lines, uint32_t salt, unsigned long long *sum,
                                 hipStream_t stream)
{
    const uint32_t n = 1u << log2_threads;
    hipLaunchKernelGGL(fetch_probe_ws_kernel, dim3(n / 256), dim3(256), 0, stream, buf, ws_lines, salt, 0xdeadbeefu, sum);
    return hipGetLastError();
}

hipError_t launch_fetch_probe(const uint8_t *buf, uint32_t log2_lines, uint32_t second_offset,
                              unsigned long long *sum, hipStream_t stream)
{
    const uint32_t n = 1u << log2_lines;
    hipLaunchKernelGGL(fetch_probe_kernel, dim3(n / 256), dim3(256), 0, stream, buf, n - 1u, 0xdeadbeefu, second_offset, sum);
    return hipGetLastError();
}

} // namespace ct
