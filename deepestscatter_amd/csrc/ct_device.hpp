// ct_device.hpp -- device-side building blocks of the gfx950 cloud path tracer.
//
// Everything here is written for CDNA4 (wave64, no texture filtering hardware): the
// reference's texture fetches (cloud.cuh:58-68) become two 8-byte loads from one 128-byte
// "apron brick" followed by an explicit float trilinear filter; the
// Mie tables live in LDS; the CDF inversion (cloud.cuh:160-180) is a guide-table search
// that returns exactly what the reference's 16-step bisection returns.
//
// Arithmetic contract: include/ct_fmath.h + -ffp-contract=off.  Operation order follows
// the reference line by line where the reference's own code fixes it; see DESIGN.md.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ct_fmath.h"

namespace ct {

#define CT_DEV __device__ __forceinline__

struct f3 {
    float x, y, z;
};

CT_DEV f3 mk3(float x, float y, float z) { return f3{ x, y, z }; }
CT_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
CT_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
CT_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
CT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CT_DEV f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// IEEE reciprocal, division and square root of the bounce's arithmetic (correctly rounded: part of the numeric contract).
// (Round 2 measured the upper bound of cheaper sequences with a wrong-results build that used the one-instruction
// approximations: DESIGN.md 4.3 item 8.)
CT_DEV float rcp_(float x) { return 1.0f / x; }
CT_DEV float div_(float a, float b) { return a / b; }
CT_DEV float sqrt_(float x) { return sqrtf(x); }
// The same correctly rounded results in fewer instructions for arguments of moderate magnitude, 2^-60 <= x <= 2^60
// (what the bounce computes: a density, the squared length of a direction-sized vector): the hardware's 1-ulp estimate
// plus one Newton step (two for the square root) carried out with fused multiply-adds, whose residual is exact.  hipcc's IEEE sequences spend
// most of their ten instructions on scaling for subnormal and huge arguments, which cannot occur here.  That these
// return the bits of 1.0f / x and sqrtf(x) for EVERY float in the range is not argued but checked, exhaustively, on the
// device: ct_debug_math_selftest / test_fast_rcp_and_sqrt_are_correctly_rounded_for_every_float_in_range.
CT_DEV float rcp_moderate(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = fmaf(-x, r0, 1.0f);
    return fmaf(e, r0, r0);
}
CT_DEV float sqrt_moderate(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float h = __builtin_amdgcn_rcpf(s0 + s0);    // ~ 1 / (2 sqrt(x))
    const float s1 = fmaf(fmaf(-s0, s0, x), h, s0);    // (the residual is exact)
    // one step is wrong for 60 floats -- mantissa all ones, every other exponent -- so a second one
    return fmaf(fmaf(-s1, s1, x), h, s1);
}

// optix::normalize = v * (1.0f / sqrtf(dot(v,v)))  (OptiX SDK optixu_math_namespace.h)
CT_DEV f3 normalize3(f3 a)
{
    const float inv = rcp_(sqrt_(dot3(a, a)));
    return scale3(a, inv);
}
// The same for a vector whose squared length is of moderate magnitude (see rcp_moderate): the two normalisations of
// getNewDirection, whose arguments are built from unit vectors (|b|^2 >= 1/2, |r| ~ 1).
CT_DEV f3 normalize3_unitish(f3 a)
{
    const float inv = rcp_moderate(sqrt_moderate(dot3(a, a)));
    return scale3(a, inv);
}

constexpr float kPi = 3.14159265358979323846f;
constexpr int kMieN = 4096;           // entries per Mie table (Mie.cpp:8-8203)
constexpr int kGuideN = 4096;         // buckets of the CDF guide table
constexpr float kFracMax = 0x1.fffffep-1f;
constexpr int kBrick = 4;             // texels per edge of a free-space brick
constexpr int kBrickShift = 2;
constexpr int kMajCellsMax = 43008;   // cells of the DELTA majorant grid (one byte each, LDS-resident): 42 KiB -- with the codes and the
                                      // tables 80400 of the 81920 bytes a block may use with two blocks per CU; 40960 until round 4,
                                      // which left a 256^3 volume (35^3 = 42875 cells of 8 texels) with 16-texel cells

// All uniforms of the path: the OptiX variable scopes of SURVEY section 8b, flattened.
struct DevScene {
    // Volumes live in HBM as "apron bricks": one 128-byte cache line per 4x4x4 texels holding the
    // 5x5x5 texels [4b, 4b+4]^3 (clamp-to-edge applied) at byte lz*25 + ly*5 + lx.  The 2x2x2
    // footprint of any trilinear lookup whose base texel lies in the brick is inside that one
    // line and is read with two unaligned 8-byte loads (bytes o.. and o+25..).  Footprint is 2x the
    // raw volume (a pre-gathered 8-byte corner per texel would be 8x and falls out of L2 and the
    // Infinity Cache).  An apron of bricks around the texture makes clamp addressing implicit for
    // every position the tracer can reach.
    const uint8_t *dbricks;  // density
    const uint8_t *ibricks;  // inScatter (shadow volume)
    int32_t brick_bias;      // added to a texel index to make it non-negative (multiple of 4)
    int32_t brick_gx;        // bricks per row
    int32_t brick_gxy;       // bricks per slice
    int32_t brick_gy, brick_gz;
    // Bytes 125 and 126 of every DENSITY brick are its meta bytes.  Byte 126 = majorant M: the
    // max of the texels [lo-1, lo+5]^3, i.e. of everything a trilinear footprint based in the brick
    // (or on its boundary) can read.  Byte 125: bits 0-6 = free-space distance D (Chebyshev
    // distance, in bricks, capped, to the nearest brick that is not "free" = M == 0 and interior),
    // bit 7 = "interior" (every position whose base texel lies in the brick passes isInBox, and so
    // does a scatter position backed off from it).  They ride in the footprint's cache line.
    // The MARCH estimator reads the density through a second brick array, `mbricks`: bricks of
    // 3x4x4 texels whose 25 rows are [t0 t1 t2 t3 M] (4 texels = 3 + apron, then a meta byte), at
    // byte lz*25 + ly*5 + lx like the others.  The first 8-byte load of a footprint based at
    // (lx,ly,lz) starts at t_lx of row (ly,lz) and therefore always contains that row's M (at byte
    // 4-lx): the march step needs two loads per lane, not three, and the L1 gather rate is what
    // bounds this kernel.  M: bits 0-6 = c, the clearance in TEXELS of the row's three base texels
    // (every footprint based within Chebyshev distance c of any of them is all zero and every such
    // position passes isInBox; texel-granular, so rows inside partly filled bricks get one too),
    // bit 7 = "interior" for the row's bases (1 <= base <= N-3 on every axis: isInBox holds for
    // every position based there and for a scatter position backed off from it).
    const uint8_t *mbricks;
    // Sparse march bricks (CT_FLAG_SPARSE_BRICKS / CT_SPARSE=1): of every brick row (by, bz) only the bricks
    // between its first and its last one that holds a non-zero texel are stored, one row after the other.
    // m_rows[bz * brick_gy + by] = (line index of the row's first stored brick, x0 | count << 16).  A footprint based
    // in a brick outside its row's extent is all zero by construction; its clearance and "interior" flag come
    // from m_coarse, one byte (same format as a row's meta byte) per cubic cell of 2^m_cshift texels: the minimum over
    // the cell's base texels.  Both tables are small (0.5 MB + 2.5 MB at 1024^3 against 2.9 GB of dense bricks) and
    // stay in L2, so a landing in open space costs no line fill.  m_rows == NULL: dense array, direct addressing.
    const uint2 *m_rows;
    const uint8_t *m_coarse;
    int32_t m_cshift, m_cgx, m_cgxy;
    // The DELTA estimator's majorants: one byte per stored cubic cell of mc_cell texels, x-fastest.  mc_cell is the smallest
    // value >= 4 for which the stored box has at most kMajCellsMax cells, so that every block keeps it in LDS and a flight
    // crosses cells without touching memory (oracle/ct_oracle.c, orc_majorant_grid: same grid, same majorants).
    // Twin bricks (DELTA estimator): one 128-byte line per 3x3x3 base texels.  Bytes 0..63 = the 4x4x4 DENSITY texels
    // [3b, 3b+3]^3 (clamp-to-edge applied) at byte lz*16 + ly*4 + lx, bytes 64..127 = the SHADOW volume's texels at the same
    // places.  A Woodcock collision is a point: the density lookup that decides it (cloud.cuh:58-62) and the NEE lookup of the
    // shadow volume that follows when it is real (cloud.cuh:146-158) have the same footprint, so with both halves in ONE line the
    // second lookup never leaves the line the first one brought -- one line fill per scatter event instead of two.  A footprint
    // based at (lx,ly,lz) in [0,2]^3 is read with two unaligned 8-byte loads at o = lz*16 + ly*4 + lx (bytes 0,1 and 4,5 = rows
    // y, y+1) and at o + 14 (bytes 2,3 and 6,7 = the same rows one slice up): both stay inside their 64-byte half.
    const uint8_t *tbricks;
    int32_t t_bias;             // added to a texel index to make it non-negative (multiple of 3)
    int32_t t_gx, t_gy, t_gz;   // bricks per axis
    uint32_t delta_nee;         // render_delta_kernel<.., NEE>: 0, 1 or 2 (2 only with tbricks)
    uint32_t delta_interior;    // 1: every non-zero texel lies at least two texels inside the volume's faces, so a REAL collision of the
                                // DELTA flight -- a position with a non-zero footprint -- is inside the box and isInBox is not evaluated;
                                // AND every stored majorant cell lies a texel or more inside the brick grid, so the texel index of a
                                // tentative collision needs no clamp (render_delta_kernel<.., INTERIOR = true>)
    const uint8_t *maj_cells;
    const uint8_t *maj_codes;   // per cell q = min(3, 4*min/max) of its texels: texel value (q*M) >> 2 bounds the cell from below
    // (round 4) A VIRTUAL grid of mc_vx x mc_vy x mc_vz cubic cells of mc_cell texels (any size >= 4) covers [-brick_bias, n + brick_bias);
    // only the box of cells around the non-zero texels is stored: mc_gx x mc_gy x mc_gz cells from virtual cell (mc_x0, mc_y0, mc_z0).
    // A virtual cell outside the box has majorant 0.  mc_div: x / mc_cell == (x * mc_div) >> 20 for every texel index of the grid.
    int32_t mc_cell, mc_div, mc_gx, mc_gy, mc_gz, mc_x0, mc_y0, mc_z0, mc_vx, mc_vy, mc_vz;
    int32_t m_bias_x;        // x bias of the 3-texel brick columns (multiple of 3)
    int32_t m_gx, m_gxy;     // bricks per row / per slice of mbricks (y and z use brick_gy/gz, brick_bias)
    int32_t nx, ny, nz;    // texels
    float sx, sy, sz;      // box coordinate -> texel coordinate (textureScale * N)
    float tsx, tsy, tsz;   // textureScale = maxDim / dims (VDBCloud.cpp:105)
    float bx, by, bz;      // bboxSize          (VDBCloud.cpp:104)
    float hx, hy, hz;      // bboxSize + 0.01f  (cloud.cuh:43)
    float density_multiplier;  // VDBCloud.cpp:109
    float sample_step;         // CloudMaterial.cpp:14
    float nlx, nly, nlz;   // -lightDirection   (cloud.cuh:153)
    float lr, lg, lb;      // lightColor * lightIntensity
    float sun_ratio;       // sunToSphereAreaRatio, cloud.cuh:148-151
    const float *mie;      // un-chopped phase texture (global; used once per path)
    const float *chopped;  // chopped phase texture   (copied to LDS)
    const float *cdf;      // chopped CDF texture     (copied to LDS)
    const uint16_t *guide; // kGuideN+2 entries       (copied to LDS)
    float ex, ey, ez, ux, uy, uz, vx, vy, vz, wx, wy, wz; // eye, U, V, W
    uint32_t width, height;
    uint32_t max_depth;
    int32_t mode;
    uint32_t tiles_x, tiles_y;
    uint32_t regen_min;    // idle lanes that trigger a regeneration phase
    uint32_t march_burst;  // march steps per scheduler visit (a burst ends early when ...
    uint32_t burst_scatter; // ... this many lanes wait for the scatter phase, or ...
    uint32_t burst_idle;   // ... this many lanes are idle)
    uint32_t tail_burst;   // march steps per visit once the job queue is empty
    uint32_t nee_cache;    // 1 = fetch_cell_cached may reuse a lane's last shadow-volume footprint (fewer than 2^25 bricks)
    uint32_t hint_period;  // scheduler visits between two looks at the job counter (power of two, 0 = never)
    uint32_t burst_march_min; // a burst also ends when fewer lanes than this still march (>= 1)
    uint32_t scatter_num, scatter_den; // run the scatter phase when nb * den > nm * num ...
    uint32_t scatter_min;  // ... and at least this many lanes wait for it (or nobody marches)
};

// ---- RNG: random.cuh:34-70 (tea<4> with v1 = subframeId) -------------------
CT_DEV uint32_t tea4(uint32_t v0, uint32_t v1)
{
    uint32_t s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

CT_DEV uint32_t lcg24(uint32_t &prev)
{
    prev = 1664525u * prev + 1013904223u;
    return prev & 0x00FFFFFFu;
}

CT_DEV float u24_to_float(uint32_t u) { return (float)u / (float)0x01000000; }

// ---- isInBox, cloud.cuh:40-44 ------------------------------------------------------------
CT_DEV bool in_box(const DevScene &sc, f3 p)
{
    return p.x >= -0.01f && p.y >= -0.01f && p.z >= -0.01f && p.x <= sc.hx && p.y <= sc.hy && p.z <= sc.hz;
}
// The same six comparisons combined on the lane masks (no short circuit: no nested divergent regions).
CT_DEV bool in_box_flat(const DevScene &sc, f3 p)
{
    return (p.x >= -0.01f) & (p.y >= -0.01f) & (p.z >= -0.01f) & (p.x <= sc.hx) & (p.y <= sc.hy) & (p.z <= sc.hz);
}

// ---- 3-D texture unit ---------------------------------------------------------------------
// floor / frac of a texel coordinate; v_fract_f32 returns min(x - floor(x), 0x1.fffffep-1)
// which is exactly the oracle's fracf().
CT_DEV float fract_(float x) { return __builtin_amdgcn_fractf(x); }

// Trilinear filter of one corner cell: x, then y, then z; lerp(a,b,t)=fma(t,b-a,a); /255 last.
CT_DEV float filter_cell(uint2 c, float wx, float wy, float wz)
{
    const float t000 = (float)(c.x & 0xffu), t100 = (float)((c.x >> 8) & 0xffu);
    const float t010 = (float)((c.x >> 16) & 0xffu), t110 = (float)(c.x >> 24);
    const float t001 = (float)(c.y & 0xffu), t101 = (float)((c.y >> 8) & 0xffu);
    const float t011 = (float)((c.y >> 16) & 0xffu), t111 = (float)(c.y >> 24);
    const float c00 = fmaf(wx, t100 - t000, t000), c10 = fmaf(wx, t110 - t010, t010);
    const float c01 = fmaf(wx, t101 - t001, t001), c11 = fmaf(wx, t111 - t011, t011);
    const float c0 = fmaf(wy, c10 - c00, c00), c1 = fmaf(wy, c11 - c01, c01);
    return fmaf(wz, c1 - c0, c0) * (1.0f / 255.0f);
}

// The 8 texels of the trilinear footprint based at texel (ix,iy,iz), packed as
// .x = t000 | t100<<8 | t010<<16 | t110<<24, .y = the same for z+1.
CT_DEV uint2 load_footprint(const DevScene &sc, const uint8_t *bricks, int32_t ix, int32_t iy, int32_t iz)
{
    const uint32_t x = (uint32_t)(ix + sc.brick_bias), y = (uint32_t)(iy + sc.brick_bias), z = (uint32_t)(iz + sc.brick_bias);
    const uint32_t brick = __umul24(z >> 2, (uint32_t)sc.brick_gxy) + __umul24(y >> 2, (uint32_t)sc.brick_gx) + (x >> 2);
    const uint32_t local = __umul24(z & 3u, 25u) + __umul24(y & 3u, 5u) + (x & 3u);
#ifdef CT_DEBUG_BOUNDS
    if ((x >> 2) >= (uint32_t)sc.brick_gx || (y >> 2) >= (uint32_t)sc.brick_gy || (z >> 2) >= (uint32_t)sc.brick_gz) {
        printf("CT_DEBUG_BOUNDS footprint texel (%d,%d,%d) outside the brick grid\n", ix, iy, iz);
        return make_uint2(0u, 0u);
    }
#endif
    const uint8_t *p = bricks + (((size_t)brick << 7) | local);
    uint2 a, c;
    __builtin_memcpy(&a, p, 8);       // bytes o+0, o+1 (y) and o+5, o+6 (y+1)
    __builtin_memcpy(&c, p + 25, 8);  // the same one z-slice up
    uint2 r;
    r.x = __builtin_amdgcn_perm(a.y, a.x, 0x06050100u);
    r.y = __builtin_amdgcn_perm(c.y, c.x, 0x06050100u);
    return r;
}

// The same plus the brick's meta byte (one more load from the same cache line).
CT_DEV uint2 load_footprint_meta(const DevScene &sc, const uint8_t *bricks, int32_t ix, int32_t iy, int32_t iz,
                                 uint32_t &meta)
{
    const uint32_t x = (uint32_t)(ix + sc.brick_bias), y = (uint32_t)(iy + sc.brick_bias), z = (uint32_t)(iz + sc.brick_bias);
    const uint32_t brick = __umul24(z >> 2, (uint32_t)sc.brick_gxy) + __umul24(y >> 2, (uint32_t)sc.brick_gx) + (x >> 2);
    const uint32_t local = __umul24(z & 3u, 25u) + __umul24(y & 3u, 5u) + (x & 3u);
#ifdef CT_DEBUG_BOUNDS
    if ((x >> 2) >= (uint32_t)sc.brick_gx || (y >> 2) >= (uint32_t)sc.brick_gy || (z >> 2) >= (uint32_t)sc.brick_gz) {
        printf("CT_DEBUG_BOUNDS footprint texel (%d,%d,%d) outside the brick grid\n", ix, iy, iz);
        meta = 0u;
        return make_uint2(0u, 0u);
    }
#endif
    const uint8_t *base = bricks + ((size_t)brick << 7);
    const uint8_t *p = base + local;
    uint2 a, c;
    __builtin_memcpy(&a, p, 8);
    __builtin_memcpy(&c, p + 25, 8);
    meta = base[125];
    uint2 r;
    r.x = __builtin_amdgcn_perm(a.y, a.x, 0x06050100u);
    r.y = __builtin_amdgcn_perm(c.y, c.x, 0x06050100u);
    return r;
}

// Footprint + row meta byte from the 3x4x4 march bricks (see DevScene::mbricks): two loads.
template <bool SPARSE>
CT_DEV uint2 load_footprint_m(const DevScene &sc, int32_t ix, int32_t iy, int32_t iz, uint32_t &meta)
{
    const uint32_t x = (uint32_t)(ix + sc.m_bias_x), y = (uint32_t)(iy + sc.brick_bias), z = (uint32_t)(iz + sc.brick_bias);
    const uint32_t bx = __umul24(x, 43691u) >> 17; // x / 3, exact for x < 2^17
    const uint32_t lx = x - __umul24(bx, 3u);
    const uint32_t brick = __umul24(z >> 2, (uint32_t)sc.m_gxy) + __umul24(y >> 2, (uint32_t)sc.m_gx) + bx;
    const uint32_t local = __umul24(z & 3u, 25u) + __umul24(y & 3u, 5u) + lx;
#ifdef CT_DEBUG_BOUNDS
    if (bx >= (uint32_t)sc.m_gx || (y >> 2) >= (uint32_t)sc.brick_gy || (z >> 2) >= (uint32_t)sc.brick_gz) {
        printf("CT_DEBUG_BOUNDS march-brick texel (%d,%d,%d) outside the grid\n", ix, iy, iz);
        meta = 0u;
        return make_uint2(0u, 0u);
    }
#endif
    const uint8_t *p = sc.mbricks + (((size_t)brick << 7) | local);
    if (SPARSE) {
        const uint32_t row = __umul24(z >> 2, (uint32_t)sc.brick_gy) + (y >> 2);
        const uint2 ri = sc.m_rows[row];
        const uint32_t rel = bx - (ri.y & 0xffffu);
        if (rel >= (ri.y >> 16)) {
            // outside the row's stored extent: zero footprint; clearance and interior flag of the coarse cell
            const uint32_t cx = (uint32_t)(ix + sc.brick_bias) >> sc.m_cshift;
            meta = sc.m_coarse[__umul24(z >> sc.m_cshift, (uint32_t)sc.m_cgxy) + __umul24(y >> sc.m_cshift, (uint32_t)sc.m_cgx) + cx];
            return make_uint2(0u, 0u);
        }
        p = sc.mbricks + (((size_t)(ri.x + rel) << 7) | local);
    }
    uint2 a, c;
    __builtin_memcpy(&a, p, 8);       // t_lx, t_lx+1 of row ly at bytes 0,1; of row ly+1 at 5,6; M at 4-lx
    __builtin_memcpy(&c, p + 25, 8);  // the same one z-slice up
    meta = __builtin_amdgcn_perm(a.y, a.x, 0x0c0c0c04u - lx);
    uint2 r;
    r.x = __builtin_amdgcn_perm(a.y, a.x, 0x06050100u);
    r.y = __builtin_amdgcn_perm(c.y, c.x, 0x06050100u);
    return r;
}

// (int)floorf(x) in one instruction (v_cvt_flr_i32_f32); identical for every in-range x.
CT_DEV int32_t floor_to_int(float x)
{
    int32_t r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// tex3D for positions the tracer can reach (inside the slack box +- one step): no clamp, the
// apron supplies clamp-to-edge.
CT_DEV float tex3_apron(const DevScene &sc, const uint8_t *bricks, f3 p)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    const int32_t ix = floor_to_int(x), iy = floor_to_int(y), iz = floor_to_int(z);
    const uint2 c = load_footprint(sc, bricks, ix, iy, iz);
    return filter_cell(c, fract_(x), fract_(y), fract_(z));
}

// The march splits the texture fetch in three so that several loads can be in flight and all-zero
// cells can skip the filter: the cell load, the free-space distance of the brick a position is in,
// and the filter at a position.
CT_DEV uint2 fetch_cell(const DevScene &sc, const uint8_t *bricks, f3 p, uint32_t &meta)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    return load_footprint_meta(sc, bricks, floor_to_int(x), floor_to_int(y), floor_to_int(z), meta);
}

// For the DELTA flights, which run to the edge of the brick grid: a position within an ulp of the grid's
// outer faces may round into a brick that does not exist, so the texel index is clamped to the grid.  The
// value is unchanged: out there the volume's clamp-to-edge makes both texels of the clamped axis equal, and
// the filter of two equal texels is that texel whatever the weight.
CT_DEV uint2 fetch_cell_in_grid(const DevScene &sc, const uint8_t *bricks, f3 p, uint32_t &meta)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    const int32_t lo = -sc.brick_bias;
    const int32_t ix = min(max(floor_to_int(x), lo), 4 * sc.brick_gx - 1 + lo);
    const int32_t iy = min(max(floor_to_int(y), lo), 4 * sc.brick_gy - 1 + lo);
    const int32_t iz = min(max(floor_to_int(z), lo), 4 * sc.brick_gz - 1 + lo);
    return load_footprint_meta(sc, bricks, ix, iy, iz, meta);
}

// ---- the DELTA kernel's two fetch layouts (render_delta_kernel<.., NEE>) ----------------------------------------------
// Byte offset of the footprint of `p` in the apron-brick arrays (density and shadow volume share it), texel index clamped
// to the grid like fetch_cell_in_grid.
CT_DEV size_t apron_offset_in_grid(const DevScene &sc, f3 p)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    const int32_t lo = -sc.brick_bias;
    const uint32_t ux = (uint32_t)(min(max(floor_to_int(x), lo), 4 * sc.brick_gx - 1 + lo) + sc.brick_bias);
    const uint32_t uy = (uint32_t)(min(max(floor_to_int(y), lo), 4 * sc.brick_gy - 1 + lo) + sc.brick_bias);
    const uint32_t uz = (uint32_t)(min(max(floor_to_int(z), lo), 4 * sc.brick_gz - 1 + lo) + sc.brick_bias);
    const uint32_t brick = __umul24(uz >> 2, (uint32_t)sc.brick_gxy) + __umul24(uy >> 2, (uint32_t)sc.brick_gx) + (ux >> 2);
    const uint32_t local = __umul24(uz & 3u, 25u) + __umul24(uy & 3u, 5u) + (ux & 3u);
    return ((size_t)brick << 7) | local;
}
// The same without the clamp, for scenes whose stored majorant cells all lie a texel or more inside the brick grid
// (DevScene::delta_interior): there the clamp never changes an index.
CT_DEV size_t apron_offset_unclamped(const DevScene &sc, f3 p)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    const uint32_t ux = (uint32_t)(floor_to_int(x) + sc.brick_bias);
    const uint32_t uy = (uint32_t)(floor_to_int(y) + sc.brick_bias);
    const uint32_t uz = (uint32_t)(floor_to_int(z) + sc.brick_bias);
    const uint32_t brick = __umul24(uz >> 2, (uint32_t)sc.brick_gxy) + __umul24(uy >> 2, (uint32_t)sc.brick_gx) + (ux >> 2);
    const uint32_t local = __umul24(uz & 3u, 25u) + __umul24(uy & 3u, 5u) + (ux & 3u);
#ifdef CT_DEBUG_BOUNDS
    if ((ux >> 2) >= (uint32_t)sc.brick_gx || (uy >> 2) >= (uint32_t)sc.brick_gy || (uz >> 2) >= (uint32_t)sc.brick_gz) {
        printf("CT_DEBUG_BOUNDS unclamped footprint texel (%u,%u,%u) outside the brick grid\n", ux, uy, uz);
        return 0;
    }
#endif
    return ((size_t)brick << 7) | local;
}
// The two 8-byte loads of a footprint, not yet combined (a load that is consumed later leaves the wave free meanwhile).
struct RawCell {
    uint2 a, c;
};
CT_DEV RawCell load_raw_apron(const uint8_t *q)
{
    RawCell r;
    __builtin_memcpy(&r.a, q, 8);
    __builtin_memcpy(&r.c, q + 25, 8);
    return r;
}
CT_DEV uint2 combine_apron(const RawCell &r)
{
    return make_uint2(__builtin_amdgcn_perm(r.a.y, r.a.x, 0x06050100u), __builtin_amdgcn_perm(r.c.y, r.c.x, 0x06050100u));
}
// Twin bricks (DevScene::tbricks): offset of the DENSITY footprint of `p`; the shadow volume's is 64 bytes further.
CT_DEV size_t twin_offset_in_grid(const DevScene &sc, f3 p)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    const int32_t lo = -sc.t_bias;
    const uint32_t ux = (uint32_t)(min(max(floor_to_int(x), lo), 3 * sc.t_gx - 1 + lo) + sc.t_bias);
    const uint32_t uy = (uint32_t)(min(max(floor_to_int(y), lo), 3 * sc.t_gy - 1 + lo) + sc.t_bias);
    const uint32_t uz = (uint32_t)(min(max(floor_to_int(z), lo), 3 * sc.t_gz - 1 + lo) + sc.t_bias);
    const uint32_t bx = __umul24(ux, 43691u) >> 17, by = __umul24(uy, 43691u) >> 17, bz = __umul24(uz, 43691u) >> 17; // u / 3, exact for u < 2^15
    const uint32_t lx = ux - __umul24(bx, 3u), ly = uy - __umul24(by, 3u), lz = uz - __umul24(bz, 3u);
    const uint32_t brick = __umul24(__umul24(bz, (uint32_t)sc.t_gy) + by, (uint32_t)sc.t_gx) + bx;
    const uint32_t local = (lz << 4) + (ly << 2) + lx;
#ifdef CT_DEBUG_BOUNDS
    if (bx >= (uint32_t)sc.t_gx || by >= (uint32_t)sc.t_gy || bz >= (uint32_t)sc.t_gz) {
        printf("CT_DEBUG_BOUNDS twin-brick texel (%u,%u,%u) outside the grid\n", ux, uy, uz);
        return 0;
    }
#endif
    return ((size_t)brick << 7) | local;
}
// (without the clamp: DevScene::delta_interior, like apron_offset_unclamped)
CT_DEV size_t twin_offset_unclamped(const DevScene &sc, f3 p)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    const uint32_t ux = (uint32_t)(floor_to_int(x) + sc.t_bias);
    const uint32_t uy = (uint32_t)(floor_to_int(y) + sc.t_bias);
    const uint32_t uz = (uint32_t)(floor_to_int(z) + sc.t_bias);
    const uint32_t bx = __umul24(ux, 43691u) >> 17, by = __umul24(uy, 43691u) >> 17, bz = __umul24(uz, 43691u) >> 17; // u / 3, exact for u < 2^15
    const uint32_t lx = ux - __umul24(bx, 3u), ly = uy - __umul24(by, 3u), lz = uz - __umul24(bz, 3u);
    const uint32_t brick = __umul24(__umul24(bz, (uint32_t)sc.t_gy) + by, (uint32_t)sc.t_gx) + bx;
    const uint32_t local = (lz << 4) + (ly << 2) + lx;
#ifdef CT_DEBUG_BOUNDS
    if (bx >= (uint32_t)sc.t_gx || by >= (uint32_t)sc.t_gy || bz >= (uint32_t)sc.t_gz) {
        printf("CT_DEBUG_BOUNDS unclamped twin-brick texel (%u,%u,%u) outside the grid\n", ux, uy, uz);
        return 0;
    }
#endif
    return ((size_t)brick << 7) | local;
}
CT_DEV RawCell load_raw_twin(const uint8_t *q)
{
    RawCell r;
    __builtin_memcpy(&r.a, q, 8);
    __builtin_memcpy(&r.c, q + 14, 8);
    return r;
}
CT_DEV uint2 combine_twin(const RawCell &r)
{
    return make_uint2(__builtin_amdgcn_perm(r.a.y, r.a.x, 0x05040100u), __builtin_amdgcn_perm(r.c.y, r.c.x, 0x07060302u));
}

// A lane's own one-entry cache in front of fetch_cell: consecutive scatter events of a path are about a mean
// free path apart, which at the reference's settings is about a texel, so the shadow-volume footprint of the next
// event is often the one just read.  `key` = the footprint's byte offset in the brick array (0xffffffff = empty);
// the volumes are immutable, so an entry never goes stale and may outlive the path that loaded it.
CT_DEV uint2 fetch_cell_cached(const DevScene &sc, const uint8_t *bricks, f3 p, uint32_t &key, uint2 &cached, bool &reused)
{
    const float fx = fmaf(p.x, sc.sx, -0.5f), fy = fmaf(p.y, sc.sy, -0.5f), fz = fmaf(p.z, sc.sz, -0.5f);
    const uint32_t x = (uint32_t)(floor_to_int(fx) + sc.brick_bias), y = (uint32_t)(floor_to_int(fy) + sc.brick_bias),
                   z = (uint32_t)(floor_to_int(fz) + sc.brick_bias);
    const uint32_t brick = __umul24(z >> 2, (uint32_t)sc.brick_gxy) + __umul24(y >> 2, (uint32_t)sc.brick_gx) + (x >> 2);
    const uint32_t local = __umul24(z & 3u, 25u) + __umul24(y & 3u, 5u) + (x & 3u);
    const uint32_t off = (brick << 7) | local;   // unique while there are fewer than 2^25 bricks: DevScene::nee_cache
    reused = sc.nee_cache != 0u && off == key;
    if (!reused) {
        const uint8_t *q = bricks + (((size_t)brick << 7) | local);
        uint2 a, c;
        __builtin_memcpy(&a, q, 8);
        __builtin_memcpy(&c, q + 25, 8);
        cached.x = __builtin_amdgcn_perm(a.y, a.x, 0x06050100u);
        cached.y = __builtin_amdgcn_perm(c.y, c.x, 0x06050100u);
        key = off;
    }
    return cached;
}

template <bool SPARSE>
CT_DEV uint2 fetch_cell_m(const DevScene &sc, f3 p, uint32_t &meta)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    return load_footprint_m<SPARSE>(sc, floor_to_int(x), floor_to_int(y), floor_to_int(z), meta);
}

CT_DEV float filter_at(const DevScene &sc, uint2 cell, f3 p)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    return filter_cell(cell, fract_(x), fract_(y), fract_(z));
}

// tex3D with explicit clamp-to-edge, for the shadow-volume precompute which marches up to
// one box length away from the box (inScatter.cu:55-59 has no box test).
CT_DEV float tex3_clamped(const DevScene &sc, const uint8_t *bricks, f3 p)
{
    const float x = fmaf(p.x, sc.sx, -0.5f), y = fmaf(p.y, sc.sy, -0.5f), z = fmaf(p.z, sc.sz, -0.5f);
    const float flx = floorf(x), fly = floorf(y), flz = floorf(z);
    // clamp in float first (far-away positions would overflow the int conversion)
    const int32_t ix = (int32_t)fminf(fmaxf(flx, -1.0f), (float)(sc.nx - 1));
    const int32_t iy = (int32_t)fminf(fmaxf(fly, -1.0f), (float)(sc.ny - 1));
    const int32_t iz = (int32_t)fminf(fmaxf(flz, -1.0f), (float)(sc.nz - 1));
    const uint2 c = load_footprint(sc, bricks, ix, iy, iz);
    return filter_cell(c, fract_(x), fract_(y), fract_(z));
}

// ---- 1-D texture unit on a float table (Mie.cpp:8229-8240): linear, clamp, normalised -----
template <typename Ptr>
CT_DEV float tex1(Ptr t, float u)
{
    const float x = fmaf(u, (float)kMieN, -0.5f);
    const float fl = floorf(x);
    const int32_t i = (int32_t)fl;
    const int32_t i0 = min(max(i, 0), kMieN - 1);
    const int32_t i1 = min(max(i + 1, 0), kMieN - 1);
    const float a = t[i0], b = t[i1];
    return fmaf(fract_(x), b - a, a);
}

// ---- CDF inversion ------------------------------------------------------------------------
// The reference bisects 16 times on tex1D(choppedMieIntegral, m) (cloud.cuh:162-180) and
// returns cosTheta = l + r - 1 with l = k/65536, r = (k+1)/65536, where
//     k = #{ j in 1..65535 : tex1D(cdf, j/65536) < val }
// because the filtered CDF is monotone.  tex1D at j/65536 sits in texel i = (j-8)>>4 with
// weight ((j-8)&15)/16, so with t = #{ i : cdf[i] < val } (found from a 4096-bucket guide
// table + a short binary search) only the 15 interior points of texel t-1 remain:
//     k = min(16(t-1) + 8 + s, 65535),   s = #{ s' in 1..15 : fma(s'/16, cdf[t]-cdf[t-1], cdf[t-1]) < val }.
// tests/test_cdf_inversion.py checks all 2^24 values of val against the literal bisection.
template <typename CdfPtr, typename GuidePtr>
CT_DEV float sample_cos_theta(CdfPtr cdf, GuidePtr guide, uint32_t u24)
{
    const float val = u24_to_float(u24);
    const uint32_t bucket = u24 >> 12;
    uint32_t lo = guide[bucket], hi = guide[bucket + 1];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] < val) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    const uint32_t t = lo; // smallest t with cdf[t] >= val, or 4096
    uint32_t k = 0;
    if (t != 0) {
        const float a = cdf[t - 1];
        const float b = cdf[min(t, (uint32_t)kMieN - 1)];
        const float d = b - a;
        uint32_t s = 0; // largest s' in 0..15 with f(s') < val (f(0) = a < val)
#pragma unroll
        for (uint32_t bit = 8; bit != 0; bit >>= 1) {
            const uint32_t cand = s | bit;
            const float f = fmaf((float)cand * 0.0625f, d, a);
            s = (f < val) ? cand : s;
        }
        k = min(16u * (t - 1) + 8u + s, 65535u);
    }
    return (float)(2u * k + 1u) * (1.0f / 65536.0f) - 1.0f;
}

// ---- getNewDirection, cloud.cuh:160-188 + uniformOnSphereCircle random.cuh:122-131 + Onb ----
template <typename CdfPtr, typename GuidePtr>
CT_DEV f3 new_direction(CdfPtr cdf, GuidePtr guide, uint32_t &seed, f3 prev)
{
    const float cos_theta = sample_cos_theta(cdf, guide, lcg24(seed));
    const float phi = u24_to_float(lcg24(seed)) * kPi * 2;
    const float sin_theta = sqrt_moderate(1 - cos_theta * cos_theta); // cos = (2k+1)/65536 - 1: 1 - cos^2 is in [2^-15, 1]
    float sn, cs;
    ct_sincosf(phi, &sn, &cs);
    const float px = sin_theta * cs, py = sin_theta * sn, pz = cos_theta;
    // optix::Onb(prev)
    f3 b;
    if (fabsf(prev.x) > fabsf(prev.z)) {
        b = mk3(-prev.y, prev.x, 0.0f);
    } else {
        b = mk3(0.0f, -prev.z, prev.y);
    }
    b = normalize3_unitish(b);
    const f3 tg = cross3(b, prev);
    // inverse_transform: p.x*tangent + p.y*binormal + p.z*normal
    const f3 r = add3(add3(scale3(tg, px), scale3(b, py)), scale3(prev, pz));
    return normalize3_unitish(r);
}

// ---- intersect, cloudBBox.cu:7-37 -----------------------------------------------------------
CT_DEV bool intersect_box(const DevScene &sc, f3 o, f3 d, float &t_hit)
{
    const float nxh = -sc.bx / 2, nyh = -sc.by / 2, nzh = -sc.bz / 2;
    const float pxh = sc.bx / 2, pyh = sc.by / 2, pzh = sc.bz / 2;
    const float t0x = (nxh - o.x) / d.x, t0y = (nyh - o.y) / d.y, t0z = (nzh - o.z) / d.z;
    const float t1x = (pxh - o.x) / d.x, t1y = (pyh - o.y) / d.y, t1z = (pzh - o.z) / d.z;
    const float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    const float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    if (tmin <= tmax) {
        if (tmin > 0.0f && tmin < 1e27f) {
            t_hit = tmin;
            return true;
        }
        t_hit = 0.000001f; // minimalRayDistance, CloudMaterial.cpp:23
        return true;
    }
    return false;
}

// ---- log of a positive NORMAL float (the march's xi/T > 1, the delta tracker's 1-u >= 2^-24):
// ct_logf's special cases (<= 0, inf, subnormal) can never fire; same instruction sequence otherwise.
CT_DEV float logf_above_one(float x)
{
    const uint32_t u = ct_float_to_bits(x);
    int32_t e = (int32_t)(u >> 23) - 126;
    float m = ct_bits_to_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    const float fe = (float)e;
    float y = (p * m) * z;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(fe, 0.693359375f, m + y);
}

// ---- exp for the march: arguments are -sigma*step in (-87, 0], so ct_expf's range checks
// can never fire; same instruction sequence otherwise (bit-identical for in-range x). ---------
CT_DEV float expf_inrange(float x)
{
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, r * r, r) + 1.0f;
    return y * ct_bits_to_float((uint32_t)((int32_t)n + 127) << 23);
}

} // namespace ct
