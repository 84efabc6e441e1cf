/*
 * ct_oracle.c -- CPU ORACLE: a plain-C restatement of the reference's cloud radiance
 * estimator (marsermd/DeepestScatter, DataGen).  TEST INFRASTRUCTURE ONLY.
 *
 *   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *     The product (libcloudtrace.so, deepestscatter_amd/) never includes, links or calls it.
 *   * PARITY UNPINNED: the reference ships no tests, golden vectors or assets (SURVEY.md
 *     section 4) and cannot be built here (OptiX 5.1 / CUDA 9.2 device programs; SURVEY.md
 *     section 8c), so this restatement is checked only against closed-form answers
 *     (tests/test_oracle_*.py) and against goldens it produced itself (tests/golden/).
 *   * Each function cites the reference lines it follows.  "src/" =
 *     DeepestScatter_DataGen/DeepestScatter_DataGen/src/.
 *
 * Third-party arithmetic the reference leans on and that is not in its tree is restated
 * from the published definitions:
 *   - NVIDIA OptiX SDK 5.1.0 optixu_math_namespace.h: normalize (v * (1/sqrtf(dot))), dot,
 *     cross, float3/float (multiply by reciprocal), clamp(f,a,b)=fmaxf(a,fminf(f,b)), Onb.
 *   - CUDA texture unit: unnormalised coordinate x = u*N - 0.5, i = floor(x), weight
 *     frac(x), clamp-to-edge on i and i+1, uchar -> float /255 (cudaReadModeNormalizedFloat).
 *     Our spec uses exact float weights (hardware uses 8-bit fixed point) and evaluates
 *     lerp(a,b,t) = fmaf(t, b-a, a), x first, then y, then z.
 *   - CUDA libm (expf/log/sin/cos/powf under --use_fast_math): replaced by the
 *     deterministic functions of include/ct_fmath.h (the numeric contract shared with the
 *     HIP kernels so that a path takes identical branches on both sides).
 *
 * Documented deviation: the reference seeds with tea<4>(pixel) where v1 = clock()
 * (random.cuh:38) -- irreproducible.  Here v1 = subframeId (NVIDIA's original two-argument
 * tea), as SURVEY.md section 8(a9) prescribes.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ct_fmath.h"

/* -DORC_LIBM: the same restatement with the C library's expf / logf / sincosf / powf in place of ct_fmath.h's polynomials
 * (libct_oracle_libm.so).  The kernels and the bit-exact oracle share ct_fmath.h, so an error in it would pass every
 * parity test; this build shares nothing of it and must agree with them statistically (a path whose xi > T or isInBox flips
 * on the last ulp goes elsewhere, so not bit for bit): tests/test_oracle_basics.py::test_oracle_with_libm_math_agrees. */
#ifdef ORC_LIBM
#define ct_expf(x) expf(x)
#define ct_logf(x) logf(x)
#define ct_powf(x, y) powf((x), (y))
#define ct_sincosf(x, s, c) sincosf((x), (s), (c))
#endif

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* optix::normalize: v * (1.0f / sqrtf(dot(v, v))) */
static inline v3 v3_normalize(v3 a)
{
    const float inv = 1.0f / sqrtf(v3_dot(a, a));
    return v3_scale(a, inv);
}
/* optix float3 / float: multiply by the reciprocal */
static inline v3 v3_div(v3 a, float s)
{
    const float inv = 1.0f / s;
    return v3_scale(a, inv);
}

#define ORC_PI_F 3.14159265358979323846f

/* ------------------------------------------------------------------------------------------
 * RNG: src/CUDA/random.cuh:34-70.  tea<4>, 24-bit LCG, rnd in [0,1).
 * ------------------------------------------------------------------------------------------ */
ORC_API uint32_t orc_tea4(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

ORC_API uint32_t orc_lcg(uint32_t *prev)
{
    *prev = 1664525u * (*prev) + 1013904223u;
    return *prev & 0x00FFFFFFu;
}

ORC_API float orc_rnd(uint32_t *prev)
{
    return (float)orc_lcg(prev) / (float)0x01000000;
}

/* ------------------------------------------------------------------------------------------
 * Mie textures: src/Mie.cpp:8206-8243 (phase / mean), :8245-8282 (running-sum CDF).
 * float32 running sums in index order.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_mie_phase_texture(const float *raw, uint32_t cnt, float *out)
{
    float average = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        average += raw[i];
    }
    average /= (float)cnt;
    for (uint32_t i = 0; i < cnt; i++) {
        out[i] = raw[i] / average;
    }
}

ORC_API void orc_mie_integral_texture(const float *raw, uint32_t cnt, float *out)
{
    float sum = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        sum += raw[i];
    }
    float integral = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        integral += raw[i] / sum;
        out[i] = integral;
    }
}

/* ------------------------------------------------------------------------------------------
 * Texture units (CUDA semantics restated, see header).
 * ------------------------------------------------------------------------------------------ */
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }
/* Filter weight: frac(x) = x - floor(x), kept strictly below 1 (a tiny negative x would
 * otherwise round to 1.0f); this is also what gfx950's v_fract_f32 returns. */
static inline float fracf(float x, float fl) { return fminf(x - fl, 0x1.fffffep-1f); }
/* The weight a filtered fetch uses.  Default (this repository's specification, shared with the kernels): the exact float
 * fraction.  -DORC_TEX_FIXED8 (libct_oracle_fixed8.so): what the reference's hardware does instead -- the CUDA C Programming
 * Guide's appendix "Texture Fetching", linear filtering: "alpha, beta and gamma are stored in 9-bit fixed point format with
 * 8 bits of fractional value (so 1.0 is exactly represented)", i.e. the fraction rounded to a multiple of 1/256.  The guide
 * does not state the rounding; round-to-nearest is used here (truncation moves every weight by up to 1/256 instead of
 * 1/512, the same order).  The texture unit's internal arithmetic is not published either: the nested lerps below stay.
 * This build exists to MEASURE how far the repository's exact-weight specification is from a sampler like the reference's
 * (samplers: VDBCloud.cpp:123-135, Mie.cpp:8229-8240); it is not a parity target -- paths branch differently, so it agrees
 * with the default build statistically only (tests/test_oracle_basics.py, tests/test_parity_gaps.py). */
#ifdef ORC_TEX_FIXED8
static inline float filter_weight(float x, float fl) { return rintf(fracf(x, fl) * 256.0f) * (1.0f / 256.0f); }
#else
static inline float filter_weight(float x, float fl) { return fracf(x, fl); }
#endif

/* tex1D on a float buffer: linear, clamp, normalised coords (Mie.cpp:8229-8240). */
ORC_API float orc_tex1d(const float *t, uint32_t n, float u)
{
    const float x = fmaf(u, (float)n, -0.5f);
    const float fl = floorf(x);
    const float w = filter_weight(x, fl);
    const int32_t i = (int32_t)fl;
    const float a = t[clampi(i, 0, (int32_t)n - 1)];
    const float b = t[clampi(i + 1, 0, (int32_t)n - 1)];
    return lerpf(a, b, w);
}

typedef struct {
    const uint8_t *texels;
    int32_t nx, ny, nz;
    float sx, sy, sz; /* box coordinate -> texel coordinate: textureScale * N, in float */
} Tex3;

/* tex3D / rtTex3D on a uchar buffer with cudaReadModeNormalizedFloat, linear, clamp,
 * normalised coords (VDBCloud.cpp:119-137; cloud.cuh:58-68).  `p` is the box-space position;
 * the texel coordinate is p * (textureScale * N) - 0.5 with the product folded on the host. */
static inline float tex3_fetch(const Tex3 *t, v3 p)
{
    const float x = fmaf(p.x, t->sx, -0.5f);
    const float y = fmaf(p.y, t->sy, -0.5f);
    const float z = fmaf(p.z, t->sz, -0.5f);
    const float flx = floorf(x), fly = floorf(y), flz = floorf(z);
    const float wx = filter_weight(x, flx), wy = filter_weight(y, fly), wz = filter_weight(z, flz);
    const int32_t ix = (int32_t)flx, iy = (int32_t)fly, iz = (int32_t)flz;
    const int32_t x0 = clampi(ix, 0, t->nx - 1), x1 = clampi(ix + 1, 0, t->nx - 1);
    const int32_t y0 = clampi(iy, 0, t->ny - 1), y1 = clampi(iy + 1, 0, t->ny - 1);
    const int32_t z0 = clampi(iz, 0, t->nz - 1), z1 = clampi(iz + 1, 0, t->nz - 1);
    const size_t sy_ = (size_t)t->nx, sz_ = (size_t)t->nx * (size_t)t->ny;
    const uint8_t *b = t->texels;
    const float t000 = (float)b[z0 * sz_ + y0 * sy_ + x0], t100 = (float)b[z0 * sz_ + y0 * sy_ + x1];
    const float t010 = (float)b[z0 * sz_ + y1 * sy_ + x0], t110 = (float)b[z0 * sz_ + y1 * sy_ + x1];
    const float t001 = (float)b[z1 * sz_ + y0 * sy_ + x0], t101 = (float)b[z1 * sz_ + y0 * sy_ + x1];
    const float t011 = (float)b[z1 * sz_ + y1 * sy_ + x0], t111 = (float)b[z1 * sz_ + y1 * sy_ + x1];
    const float c00 = lerpf(t000, t100, wx), c10 = lerpf(t010, t110, wx);
    const float c01 = lerpf(t001, t101, wx), c11 = lerpf(t011, t111, wx);
    const float c0 = lerpf(c00, c10, wy), c1 = lerpf(c01, c11, wy);
    return lerpf(c0, c1, wz) * (1.0f / 255.0f);
}

/* ------------------------------------------------------------------------------------------
 * Scene block: the OptiX variable scopes of the path, flattened.
 * ------------------------------------------------------------------------------------------ */
typedef struct OrcScene {
    uint32_t dims[3];
    const uint8_t *density;    /* X*Y*Z, x fastest (Resources.cpp:127-141) */
    const uint8_t *inscatter;  /* X*Y*Z, result of orc_inscatter */
    float cloud_size_m;        /* main.cpp:63 */
    float mean_free_path_m;    /* SceneDescription.h:80 */
    float sample_step;         /* installers.cpp:86 */
    int32_t mode;              /* 0 totalRadiance, 1 multipleScatterSunRadiance, 2 singleScatterSunRadiance */
    uint32_t max_depth;        /* cloudRadianceMaterials.cu:4 */
    float light_direction[3];  /* as given by the user; normalised twice below */
    float light_color[3];
    float light_intensity;
    uint32_t width, height;
    float eye[3], U[3], V[3], W[3];
    const float *mie_tex;          /* orc_mie_phase_texture(mie) */
    const float *chopped_mie_tex;  /* orc_mie_phase_texture(choppedMie) */
    const float *chopped_cdf_tex;  /* orc_mie_integral_texture(choppedMie) */
    uint32_t mie_count;
    /* free-flight sampler: 0 = MARCH (the reference's, cloud.cuh:77-114), 1 = DELTA (Woodcock tracking
     * over per-brick majorants; NOT in the reference -- BASELINE.json north_star / SURVEY section 7.6) */
    int32_t estimator;
    const uint8_t *majorant;       /* orc_build_majorants(), needed for DELTA */
    int32_t maj_bias, maj_gx, maj_gy, maj_gz, maj_cell; /* stored cells per axis; cell edge in texels (orc_majorant_grid) */
    const uint8_t *maj_codes;      /* orc_build_majorants(): per cell q = min(3, 4*min/max), see delta_flight */
    /* Lazy shadow volume (NULL = `inscatter` is complete).  Otherwise one byte per texel, 0 = `inscatter` does not hold
     * that texel yet: sampleInScatter evaluates inScatter (inScatter.cu:40-66) for the missing ones of its footprint
     * and stores them -- the oracle's OWN shadow values for exactly the texels its paths touch, which is what lets a
     * 512^3 / 1024^3 window test check the product's whole shadow-volume kernel where it matters without a CPU pass
     * over 1e8-1e9 texels.  `inscatter` must then be writable. */
    uint8_t *inscatter_valid;
    /* DELTA grid, continued: the stored cells are the sub-range [maj_origin, maj_origin + (maj_gx, maj_gy, maj_gz)) of a virtual
     * grid of maj_virtual cells over the texel range [-bias, n + bias); a virtual cell outside the stored range has majorant 0 */
    int32_t maj_origin[3], maj_virtual[3];
} OrcScene;

typedef struct OrcCounters {
    uint64_t paths, box_hits, density_lookups, inscatter_lookups, scatter_events, depth_capped;
} OrcCounters;

/* Derived uniforms: VDBCloud::setupVolumeVariables (VDBCloud.cpp:98-111), Sun::init
 * (Sun.cpp:13-18), DirectionalLight ctor (SceneDescription.h:15-16) after
 * installSceneSetup's normalize (installers.cpp:74-78). */
typedef struct {
    v3 bbox;           /* bboxSize = dims / maxDim */
    v3 tscale;         /* textureScale = maxDim / dims */
    float density_multiplier;
    float sample_step;
    v3 light_dir;      /* twice-normalised travel direction */
    v3 light_rgb;      /* lightColor * lightIntensity */
    float sun_ratio;   /* sunToSphereAreaRatio, cloud.cuh:148-151 */
    Tex3 density, inscatter;
    const float *mie, *chopped, *cdf;
    uint32_t mie_n;
    uint32_t max_depth;
    int32_t mode;
    int32_t estimator;
    const uint8_t *maj, *maj_codes;
    int32_t maj_bias, maj_gx, maj_gy, maj_gz, maj_cell, maj_x0, maj_y0, maj_z0, maj_vx, maj_vy, maj_vz;
    uint8_t *ins_valid;      /* lazy shadow volume: see OrcScene::inscatter_valid */
    float sigma_global;      /* estimator 2: majorant of the whole volume */
} Ctx;

static void ctx_init(Ctx *c, const OrcScene *s)
{
    const float fx = (float)s->dims[0], fy = (float)s->dims[1], fz = (float)s->dims[2];
    const float maxs = fmaxf(fmaxf(fx, fy), fz);
    c->bbox = v3_make(fx / maxs, fy / maxs, fz / maxs);
    c->tscale = v3_make(maxs / fx, maxs / fy, maxs / fz);
    c->density_multiplier = s->cloud_size_m / s->mean_free_path_m;
    c->sample_step = s->sample_step;
    v3 l = v3_make(s->light_direction[0], s->light_direction[1], s->light_direction[2]);
    l = v3_normalize(l);
    l = v3_normalize(l);
    c->light_dir = l;
    c->light_rgb = v3_scale(v3_make(s->light_color[0], s->light_color[1], s->light_color[2]), s->light_intensity);
    {
        /* cloud.cuh:148-151, evaluated in float like the device code */
        const float sunAngularRadiusDeg = 0.53f / 2;
        const float sphereArea = 4 * ORC_PI_F;
        const float sunArea = 2 * ORC_PI_F * (1 - cosf(sunAngularRadiusDeg * ORC_PI_F / 180.0f));
        c->sun_ratio = sunArea / sphereArea;
    }
    c->density.texels = s->density;
    c->density.nx = (int32_t)s->dims[0];
    c->density.ny = (int32_t)s->dims[1];
    c->density.nz = (int32_t)s->dims[2];
    c->density.sx = c->tscale.x * fx;
    c->density.sy = c->tscale.y * fy;
    c->density.sz = c->tscale.z * fz;
    c->inscatter = c->density;
    c->inscatter.texels = s->inscatter;
    c->mie = s->mie_tex;
    c->chopped = s->chopped_mie_tex;
    c->cdf = s->chopped_cdf_tex;
    c->mie_n = s->mie_count;
    c->max_depth = s->max_depth;
    c->mode = s->mode;
    c->estimator = s->estimator;
    c->maj = s->majorant;
    c->maj_bias = s->maj_bias;
    c->maj_gx = s->maj_gx;
    c->maj_gy = s->maj_gy;
    c->maj_gz = s->maj_gz;
    c->maj_cell = s->maj_cell;
    c->maj_x0 = s->maj_origin[0];
    c->maj_y0 = s->maj_origin[1];
    c->maj_z0 = s->maj_origin[2];
    c->maj_vx = s->maj_virtual[0];
    c->maj_vy = s->maj_virtual[1];
    c->maj_vz = s->maj_virtual[2];
    c->maj_codes = s->maj_codes;
    c->ins_valid = s->inscatter_valid;
    c->sigma_global = 0.0f;
    if (s->estimator == 2) {
        uint8_t m = 0;
        const size_t n = (size_t)s->dims[0] * s->dims[1] * s->dims[2];
        for (size_t i = 0; i < n; i++) {
            m = s->density[i] > m ? s->density[i] : m;
        }
        c->sigma_global = ((float)m / 255.0f) * c->density_multiplier;
    }
}

/* Exposes the derived uniforms so tests can compare them with the product's. */
ORC_API void orc_derived_uniforms(const OrcScene *s, float out[16])
{
    Ctx c;
    ctx_init(&c, s);
    out[0] = c.bbox.x; out[1] = c.bbox.y; out[2] = c.bbox.z;
    out[3] = c.tscale.x; out[4] = c.tscale.y; out[5] = c.tscale.z;
    out[6] = c.density_multiplier;
    out[7] = c.light_dir.x; out[8] = c.light_dir.y; out[9] = c.light_dir.z;
    out[10] = c.light_rgb.x; out[11] = c.light_rgb.y; out[12] = c.light_rgb.z;
    out[13] = c.sun_ratio;
    out[14] = c.density.sx; out[15] = c.sample_step;
}

/* isInBox, cloud.cuh:40-44 (0.01 slack on every side). */
static inline int in_box(const Ctx *c, v3 p)
{
    return p.x >= -0.01f && p.y >= -0.01f && p.z >= -0.01f &&
           p.x <= c->bbox.x + 0.01f && p.y <= c->bbox.y + 0.01f && p.z <= c->bbox.z + 0.01f;
}

/* sampleCloud cloud.cuh:58-62 / sampleInScatter :64-68 */
static inline float sample_cloud(const Ctx *c, v3 pos, OrcCounters *k)
{
    k->density_lookups++;
    return tex3_fetch(&c->density, pos);
}
static uint8_t inscatter_texel(const Ctx *c, int32_t x, int32_t y, int32_t z);

/* Lazy shadow volume: make sure the 8 texels of the footprint at `pos` are there (same addressing as tex3_fetch).
 * Racing threads store the same byte; the flag is published after the value. */
static void inscatter_ensure(const Ctx *c, v3 pos)
{
    const Tex3 *t = &c->inscatter;
    const int32_t ix = (int32_t)floorf(fmaf(pos.x, t->sx, -0.5f)), iy = (int32_t)floorf(fmaf(pos.y, t->sy, -0.5f)),
                  iz = (int32_t)floorf(fmaf(pos.z, t->sz, -0.5f));
    for (int dz = 0; dz < 2; dz++) {
        for (int dy = 0; dy < 2; dy++) {
            for (int dx = 0; dx < 2; dx++) {
                const int32_t x = clampi(ix + dx, 0, t->nx - 1), y = clampi(iy + dy, 0, t->ny - 1), z = clampi(iz + dz, 0, t->nz - 1);
                const size_t at = ((size_t)z * t->ny + y) * t->nx + x;
                if (!__atomic_load_n(&c->ins_valid[at], __ATOMIC_ACQUIRE)) {
                    const uint8_t v = inscatter_texel(c, x, y, z);
                    __atomic_store_n((uint8_t *)&t->texels[at], v, __ATOMIC_RELAXED);
                    __atomic_store_n(&c->ins_valid[at], 1, __ATOMIC_RELEASE);
                }
            }
        }
    }
}

static inline float sample_inscatter(const Ctx *c, v3 pos, OrcCounters *k)
{
    k->inscatter_lookups++;
    if (c->ins_valid) {
        inscatter_ensure(c, pos);
    }
    return tex3_fetch(&c->inscatter, pos);
}

typedef struct { int scattered; v3 pos; float transmittance; } Event;

/* getNextScatteringEvent, cloud.cuh:77-114 with stopAtScatterPos = true (shouldSampleSky is
 * false, cloudRadianceMaterials.cu:25,36): step, THEN sample; collide when xi > T (strict). */
static Event next_scattering_event(const Ctx *c, float optical_distance, v3 pos, v3 direction, OrcCounters *k)
{
    const v3 step_along_ray = v3_scale(direction, c->sample_step);
    float transmittance = 1;
    int scattered = 0;
    v3 scatter_pos = v3_make(0, 0, 0);
    while (in_box(c, pos)) {
        pos = v3_add(pos, step_along_ray);
        const float density = sample_cloud(c, pos, k) * c->density_multiplier;
        const float extinction = density * c->sample_step;
        const float current_transmit = ct_expf(-extinction);
        transmittance *= current_transmit;
        if (optical_distance > transmittance) {
            scattered = 1;
            /* pos - direction * log(xi / T) / density  (cloud.cuh:99) */
            const float lg = ct_logf(optical_distance / transmittance);
            scatter_pos = v3_sub(pos, v3_div(v3_scale(direction, lg), density));
            break;
        }
    }
    if (!scattered && !in_box(c, pos)) {
        scatter_pos = pos;
    }
    Event e = { scattered, scatter_pos, transmittance };
    return e;
}

/* ------------------------------------------------------------------------------------------
 * DELTA estimator (not in the reference): Woodcock tracking of the SAME medium -- sigma(x) =
 * densityMultiplier * trilinear(texture)/255, the field the reference's march samples once per
 * step -- over a grid of cubic cells of C = 2^shift texels with majorants.
 *
 *   The grid covers the texel range [-bias, n + bias) per axis (bias = the brick grid's apron, a
 *   multiple of 4).  shift is the smallest value >= 2 for which the grid has at most
 *   ORC_MAJ_CELLS_MAX cells: the product keeps the whole grid in the LDS of a compute unit, so a
 *   flight crosses cells without touching memory (16-texel cells for a 512^3 volume).
 *   cell c (per axis) covers base texels [C*c - bias, C*c - bias + C - 1]; its majorant is the max
 *   of the texels [lo-1, lo+C+1]^3 (clamped): every texel a trilinear footprint based in the cell
 *   can read, plus one texel of slack on each side for positions that sit on a boundary.
 *   sigma_bar = (float)M * (1/255.f) * densityMultiplier.
 *
 *   flight from pos along dir:  tp = pos*scale - 0.5 (texel coordinates), v = dir*scale;
 *   3-D DDA over the cells in the ray parameter t (box units): tmax_a = (bound_a - tp_a)*(1/v_a),
 *   tdelta_a = C*|1/v_a|.  In a cell with M > 0:  dt = -log(1 - rnd)*(1/sigma_bar); if t + dt reaches
 *   the cell's exit the flight moves on to the next cell (the exponential is memoryless);
 *   otherwise t += dt, p = fma(dir, t, pos), sigma = sample(p) [one density lookup], and the
 *   collision is real when rnd * sigma_bar < sigma.  A cell also carries a 2-bit code q = min(3, 4*min/max) of
 *   its texels: sigma >= sigma_low = sigma of the texel value (q*M) >> 2 throughout the cell, so when
 *   rnd * sigma_bar < sigma_low the collision is known to be real and the lookup is not made (half of all real
 *   collisions on the benchmark cloud).  Leaving the grid ends the flight without a
 *   collision.  Unbiased for the trilinear medium; the reference's march is an O(step)-biased
 *   estimator of the same free-flight distribution.
 * ------------------------------------------------------------------------------------------ */
#define ORC_MAJ_CELLS_MAX 43008   /* (42 KiB: what the product's LDS holds beside its tables; a 256^3 volume gets 8-texel cells: 35^3) */

/* The grid (round 4).  A VIRTUAL grid of cubic cells of C texels covers the texel range [-bias, n + bias) per axis, cell c
 * the base texels [C*c - bias, C*c - bias + C - 1]; of it only the cells that can have a non-zero majorant are STORED: per axis
 * the range of cells whose clamped read interval [clamp(lo - 1), clamp(lo + C + 1)] meets the bounding interval of the non-zero
 * texels (a product of three ranges: a box of cells around the cloud).  C is the smallest value >= 4 for which the stored box
 * has at most ORC_MAJ_CELLS_MAX cells -- 12-texel cells for the benchmark cloud at 512^3, whose box is 0.9 x 0.6 x 0.8 of the
 * volume, where the whole volume allowed 16.  A virtual cell outside the stored box has majorant 0 by construction.
 * out = { bias, C, origin x, y, z (virtual cell of the first stored one), stored cells x, y, z, virtual cells x, y, z } */
ORC_API void orc_majorant_grid(const uint8_t *texels, const uint32_t dims[3], float sample_step, int32_t out[11])
{
    const float m = (float)(dims[0] > dims[1] ? (dims[0] > dims[2] ? dims[0] : dims[2]) : (dims[1] > dims[2] ? dims[1] : dims[2]));
    const int32_t apron = (int32_t)ceilf((0.01f + 8.0f * sample_step) * m + 0.5f) + 1;
    const int32_t bias = ((apron + 3) / 4) * 4;
    const int32_t n[3] = { (int32_t)dims[0], (int32_t)dims[1], (int32_t)dims[2] };
    /* bounding interval of the non-zero texels per axis (lo > hi: none) */
    int32_t lo[3] = { n[0], n[1], n[2] }, hi[3] = { -1, -1, -1 };
    for (int32_t z = 0; z < n[2]; z++) {
        for (int32_t y = 0; y < n[1]; y++) {
            const uint8_t *row = texels + ((size_t)z * n[1] + y) * n[0];
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
            lo[0] = x0 < lo[0] ? x0 : lo[0];
            hi[0] = x1 > hi[0] ? x1 : hi[0];
            lo[1] = y < lo[1] ? y : lo[1];
            hi[1] = y > hi[1] ? y : hi[1];
            lo[2] = z < lo[2] ? z : lo[2];
            hi[2] = z > hi[2] ? z : hi[2];
        }
    }
    out[0] = bias;
    for (int32_t C = 4;; C++) {
        int64_t cells = 1;
        for (int a = 0; a < 3; a++) {
            const int32_t v = (n[a] + 2 * bias + C - 1) / C;
            int32_t c0 = v, c1 = -1;
            for (int32_t c = 0; c < v; c++) {
                const int32_t r0 = clampi(C * c - bias - 1, 0, n[a] - 1), r1 = clampi(C * c - bias + C + 1, 0, n[a] - 1);
                if (r0 <= hi[a] && r1 >= lo[a]) {
                    c0 = c < c0 ? c : c0;
                    c1 = c;
                }
            }
            if (c1 < c0) { /* an empty volume: one stored cell (its majorant will be 0) */
                c0 = c1 = 0;
            }
            out[2 + a] = c0;
            out[5 + a] = c1 - c0 + 1;
            out[8 + a] = v;
            cells *= (int64_t)(c1 - c0 + 1);
        }
        if (cells <= ORC_MAJ_CELLS_MAX) {
            out[1] = C;
            return;
        }
    }
}

/* Majorants and lower-bound codes of the stored cells: cell (cx, cy, cz) of the stored box is virtual cell origin + (cx, cy, cz). */
ORC_API void orc_build_majorants(const uint8_t *texels, const uint32_t dims[3], int32_t bias, int32_t C, const int32_t origin[3],
                                 int32_t gx, int32_t gy, int32_t gz, uint8_t *out, uint8_t *out_codes)
{
    const int32_t nx = (int32_t)dims[0], ny = (int32_t)dims[1], nz = (int32_t)dims[2];
#pragma omp parallel for schedule(static)
    for (int32_t cz = 0; cz < gz; cz++) {
        for (int32_t cy = 0; cy < gy; cy++) {
            for (int32_t cx = 0; cx < gx; cx++) {
                const int32_t x0 = C * (cx + origin[0]) - bias, y0 = C * (cy + origin[1]) - bias, z0 = C * (cz + origin[2]) - bias;
                uint8_t m = 0, lo = 255;
                for (int32_t z = z0 - 1; z <= z0 + C + 1; z++) {
                    const int32_t zc = clampi(z, 0, nz - 1);
                    for (int32_t y = y0 - 1; y <= y0 + C + 1; y++) {
                        const int32_t yc = clampi(y, 0, ny - 1);
                        for (int32_t x = x0 - 1; x <= x0 + C + 1; x++) {
                            const uint8_t v = texels[((size_t)zc * ny + yc) * nx + clampi(x, 0, nx - 1)];
                            m = v > m ? v : m;
                            lo = v < lo ? v : lo;
                        }
                    }
                }
                out[((size_t)cz * gy + cy) * gx + cx] = m;
                const uint32_t q = m ? (4u * lo) / m : 0u;
                out_codes[((size_t)cz * gy + cy) * gx + cx] = (uint8_t)(q > 3u ? 3u : q);
            }
        }
    }
}

static Event delta_flight(const Ctx *c, uint32_t *seed, v3 pos, v3 dir, OrcCounters *k)
{
    Event e = { 0, pos, 1.0f };
    const float tp[3] = { fmaf(pos.x, c->density.sx, -0.5f), fmaf(pos.y, c->density.sy, -0.5f),
                          fmaf(pos.z, c->density.sz, -0.5f) };
    const float v[3] = { dir.x * c->density.sx, dir.y * c->density.sy, dir.z * c->density.sz };
    int32_t b[3], step[3];
    float tmax[3], tdelta[3];
    const float edge = (float)c->maj_cell;
    for (int a = 0; a < 3; a++) {
        const int32_t cell = (int32_t)floorf(tp[a]) + c->maj_bias;
        b[a] = cell / c->maj_cell;   /* (cell >= 0: the bias covers every position a flight can start from) */
        if (v[a] > 0.0f) {
            const float inv = 1.0f / v[a]; /* one division per axis; the products below are what the kernel computes */
            step[a] = 1;
            tmax[a] = ((float)((b[a] + 1) * c->maj_cell - c->maj_bias) - tp[a]) * inv;
            tdelta[a] = edge * inv;
        } else if (v[a] < 0.0f) {
            const float inv = 1.0f / v[a];
            step[a] = -1;
            tmax[a] = ((float)(b[a] * c->maj_cell - c->maj_bias) - tp[a]) * inv;
            tdelta[a] = edge * -inv;
        } else {
            step[a] = 0;
            tmax[a] = INFINITY;
            tdelta[a] = INFINITY;
        }
    }
    float t = 0.0f;
    for (;;) {
        if (b[0] < 0 || b[1] < 0 || b[2] < 0 || b[0] >= c->maj_vx || b[1] >= c->maj_vy || b[2] >= c->maj_vz) {
            e.pos = v3_make(fmaf(dir.x, t, pos.x), fmaf(dir.y, t, pos.y), fmaf(dir.z, t, pos.z));
            return e; /* left the (virtual) grid: no collision */
        }
        const float t_exit = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
        /* a cell outside the stored box has majorant 0 (the kernel ends a flight that LEAVES the box -- a straight line does not
         * come back into a box, and nothing out there draws a random number -- and walks one that starts outside up to it) */
        const int32_t sx = b[0] - c->maj_x0, sy = b[1] - c->maj_y0, sz = b[2] - c->maj_z0;
        const int stored = sx >= 0 && sy >= 0 && sz >= 0 && sx < c->maj_gx && sy < c->maj_gy && sz < c->maj_gz;
        const size_t cell = stored ? ((size_t)sz * c->maj_gy + sy) * c->maj_gx + sx : 0;
        const uint8_t M = stored ? c->maj[cell] : 0;
        if (M != 0) {
            const float sigma_bar = ((float)M * (1.0f / 255.0f)) * c->density_multiplier;
            const float mean_free = 1.0f / sigma_bar; /* (the kernel keeps both in a 256-entry LDS table) */
            /* lower bound of sigma in the cell: texel value (q*M) >> 2 <= the cell's smallest texel */
            const float sigma_low = ((float)(((uint32_t)c->maj_codes[cell] * M) >> 2) * (1.0f / 255.0f)) * c->density_multiplier;
            for (;;) {
                const float u = orc_rnd(seed);
                const float dt = -ct_logf(1.0f - u) * mean_free;
                if (t + dt >= t_exit) {
                    break;
                }
                t = t + dt;
                const v3 p = v3_make(fmaf(dir.x, t, pos.x), fmaf(dir.y, t, pos.y), fmaf(dir.z, t, pos.z));
                const float z = orc_rnd(seed);
                /* sigma(p) >= sigma_low everywhere in the cell (every rounding on the way is monotone), so below
                 * sigma_low the collision is real whatever the lookup would say: it is not made (nor counted) */
                if (z * sigma_bar < sigma_low || z * sigma_bar < sample_cloud(c, p, k) * c->density_multiplier) {
                    e.scattered = 1;
                    e.pos = p;
                    return e;
                }
            }
        }
        t = t_exit;
        const int a = (tmax[0] <= tmax[1]) ? ((tmax[0] <= tmax[2]) ? 0 : 2) : ((tmax[1] <= tmax[2]) ? 1 : 2);
        b[a] += step[a];
        tmax[a] += tdelta[a];
    }
}

/* estimator 2 -- an INDEPENDENT check of the DELTA kernel, not its twin: textbook Woodcock tracking of the same medium
 * with ONE majorant for the whole volume -- no cell grid, no DDA, no lower-bound codes, libm's logf instead of
 * ct_fmath.h's, a plain division.  It shares nothing with delta_flight (or with the kernel) but the sampler, and consumes
 * the random stream differently, so it agrees with them statistically only: tests compare means within the combined
 * confidence interval.  The flight ends when it leaves the slack box: a straight line cannot come back, a collision out
 * there ends the path (cloudRadianceMaterials.cu:49-52), and so does leaving the grid in delta_flight. */
static Event global_majorant_flight(const Ctx *c, uint32_t *seed, v3 pos, v3 dir, OrcCounters *k)
{
    Event e = { 0, pos, 1.0f };
    if (!(c->sigma_global > 0.0f)) {
        return e;
    }
    double t = 0.0;
    for (;;) {
        const float u = orc_rnd(seed);
        t += (double)(-logf(1.0f - u)) / (double)c->sigma_global;
        const v3 p = v3_make((float)(pos.x + dir.x * t), (float)(pos.y + dir.y * t), (float)(pos.z + dir.z * t));
        if (!in_box(c, p)) {
            e.pos = p;
            return e;
        }
        const float z = orc_rnd(seed);
        if (z * c->sigma_global < sample_cloud(c, p, k) * c->density_multiplier) {
            e.scattered = 1;
            e.pos = p;
            return e;
        }
    }
}

/* One free flight with the scene's estimator; MARCH draws its random number first (cloud.cuh:120). */
static Event free_flight(const Ctx *c, uint32_t *seed, v3 pos, v3 dir, OrcCounters *k)
{
    if (c->estimator == 2) {
        if (!in_box(c, pos)) {
            const Event e = { 0, pos, 1.0f };
            return e;
        }
        return global_majorant_flight(c, seed, pos, dir, k);
    }
    if (c->estimator == 1) {
        /* getNextScatteringEvent only marches while isInBox(pos) (cloud.cuh:87): a flight that STARTS outside the
         * slack box does nothing.  The bounce loops of totalRadiance / multipleScatterSunRadiance test that before every
         * flight anyway (cloudRadianceMaterials.cu:28,91); singleScatterSunRadiance (:134) relies on the march's own
         * loop condition, so the DELTA twin needs the test here -- the box "hit" of a camera that looks AWAY from a box
         * right behind it starts at eye + 1e-6 * dir (cloudBBox.cu:26-33), outside the box.  (Round 2: this test was
         * missing, the kernel had it; found by the soak as a counter-only difference, seed 3003 case 634.) */
        if (!in_box(c, pos)) {
            const Event e = { 0, pos, 1.0f };
            return e;
        }
        return delta_flight(c, seed, pos, dir, k);
    }
    const float xi = orc_rnd(seed);
    return next_scattering_event(c, xi, pos, dir, k);
}

/* getInScattering, cloud.cuh:146-158 (NEE through the pre-integrated shadow volume). */
static v3 in_scattering(const Ctx *c, v3 scatter_pos, v3 direction, int chopped, OrcCounters *k)
{
    const float cos_light = v3_dot(v3_neg(c->light_dir), direction);
    const float phase = orc_tex1d(chopped ? c->chopped : c->mie, c->mie_n, (cos_light + 1) / 2);
    const float ins = sample_inscatter(c, scatter_pos, k);
    v3 l = v3_scale(c->light_rgb, ins);
    l = v3_scale(l, phase);
    l = v3_scale(l, c->sun_ratio);
    k->scatter_events++;
    return l;
}

/* optix::Onb(n).inverse_transform(p) -- OptiX SDK 5.1 optixu_math_namespace.h (restated). */
static v3 onb_inverse_transform(v3 n, v3 p)
{
    v3 b;
    if (fabsf(n.x) > fabsf(n.z)) {
        b = v3_make(-n.y, n.x, 0);
    } else {
        b = v3_make(0, -n.z, n.y);
    }
    b = v3_normalize(b);
    const v3 t = v3_cross(b, n);
    return v3_add(v3_add(v3_scale(t, p.x), v3_scale(b, p.y)), v3_scale(n, p.z));
}

/* getNewDirection, cloud.cuh:160-188: 16-step bisection on the chopped-Mie CDF texture, then
 * uniformOnSphereCircle (random.cuh:122-131), rotate into the previous direction's frame. */
static v3 new_direction(const Ctx *c, uint32_t *seed, v3 previous)
{
    float l = 0.f, r = 1.f, m;
    const float val = orc_rnd(seed);
    for (int i = 0; i < 16; i++) {
        m = (l + r) / 2.f;
        if (val > orc_tex1d(c->cdf, c->mie_n, m)) {
            l = m;
        } else {
            r = m;
        }
    }
    const float cos_theta = (l + r) - 1;
    const float phi = orc_rnd(seed) * ORC_PI_F * 2;
    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    float sn, cs;
    ct_sincosf(phi, &sn, &cs);
    const v3 local = v3_make(sin_theta * cs, sin_theta * sn, cos_theta);
    return v3_normalize(onb_inverse_transform(previous, local));
}

/* intersect, cloudBBox.cu:7-37 with minimalRayDistance = 1e-6 (CloudMaterial.cpp:23), ray
 * interval (sceneEPS = 0 [never set], RT_DEFAULT_MAX = 1e27).  Returns 1 and *t_hit on a hit. */
static int intersect_box(const Ctx *c, v3 origin, v3 dir, float *t_hit)
{
    const v3 boxmin = v3_make(-c->bbox.x / 2, -c->bbox.y / 2, -c->bbox.z / 2);
    const v3 boxmax = v3_make(c->bbox.x / 2, c->bbox.y / 2, c->bbox.z / 2);
    const v3 t0 = v3_make((boxmin.x - origin.x) / dir.x, (boxmin.y - origin.y) / dir.y, (boxmin.z - origin.z) / dir.z);
    const v3 t1 = v3_make((boxmax.x - origin.x) / dir.x, (boxmax.y - origin.y) / dir.y, (boxmax.z - origin.z) / dir.z);
    const float tmin = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
    const float tmax = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
    const float ray_tmin = 0.0f, ray_tmax = 1e27f;
    if (tmin <= tmax) {
        if (tmin > ray_tmin && tmin < ray_tmax) {
            *t_hit = tmin;
            return 1;
        }
        const float minimal = 0.000001f;
        if (minimal > ray_tmin && minimal < ray_tmax) {
            *t_hit = minimal;
            return 1;
        }
    }
    return 0;
}

/* Closest-hit programs of cloudRadianceMaterials.cu for one primary ray. */
static v3 radiance_of_ray(const Ctx *c, v3 origin, v3 ray_dir, uint32_t seed, OrcCounters *k)
{
    v3 radiance = v3_make(0, 0, 0);
    float t_hit;
    k->paths++;
    if (!intersect_box(c, origin, ray_dir, &t_hit)) {
        return radiance; /* progressive.cu:44-46: empty miss program */
    }
    k->box_hits++;
    v3 pos = v3_add(origin, v3_scale(ray_dir, t_hit));   /* :11 */
    pos = v3_add(pos, v3_scale(c->bbox, 0.5f));            /* :12 */
    v3 direction = v3_normalize(ray_dir);                  /* :17 */

    if (c->mode == 2) {
        /* singleScatterSunRadiance, :120-148 */
        const Event e = free_flight(c, &seed, pos, direction, k);
        if (e.scattered && in_box(c, e.pos)) {
            radiance = v3_add(radiance, in_scattering(c, e.pos, direction, 0, k));
        }
        return radiance;
    }
    if (c->mode == 1) {
        direction = new_direction(c, &seed, direction);    /* :86 */
    }
    uint32_t depth = 0;
    while (in_box(c, pos)) {
        depth++;
        if (depth == c->max_depth) {
            k->depth_capped++;
            break;
        }
        const Event e = free_flight(c, &seed, pos, direction, k);
        if (!e.scattered || !in_box(c, e.pos)) {
            break;
        }
        /* totalRadiance :56 un-chopped Mie at depth 1; multipleScatter :105 always chopped */
        const int chopped = (c->mode == 1) ? 1 : (depth != 1);
        radiance = v3_add(radiance, in_scattering(c, e.pos, direction, chopped, k));
        pos = e.pos;
        direction = new_direction(c, &seed, direction);
    }
    return radiance;
}

/* pinholeCamera pathTracingCamera.cu:12-21 + trace<> cameraCommon.cuh:18-30 for the pixel
 * window [x0,x1) x [y0,y1); frame_rgba is the full W x H float4 image (row 0 = bottom).
 * Seed: tea<4>(x*4096 + y, subframe_id)  (cloudRadianceMaterials.cu:21 + deviation above). */
ORC_API void orc_render_subframe(const OrcScene *s, uint32_t subframe_id,
                                 uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                                 float *frame_rgba, OrcCounters *counters, int32_t threads)
{
    Ctx c;
    ctx_init(&c, s);
    const v3 eye = v3_make(s->eye[0], s->eye[1], s->eye[2]);
    const v3 U = v3_make(s->U[0], s->U[1], s->U[2]);
    const v3 V = v3_make(s->V[0], s->V[1], s->V[2]);
    const v3 W = v3_make(s->W[0], s->W[1], s->W[2]);
    uint64_t paths = 0, hits = 0, dl = 0, il = 0, se = 0, dc = 0;
    const int64_t w = (int64_t)x1 - (int64_t)x0, h = (int64_t)y1 - (int64_t)y0;
    const int64_t total = (w > 0 && h > 0) ? w * h : 0;
#ifdef _OPENMP
    if (threads <= 0) {
        threads = omp_get_max_threads();
    }
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : paths, hits, dl, il, se, dc)
    for (int64_t idx = 0; idx < total; idx++) {
        const uint32_t px = x0 + (uint32_t)(idx % w);
        const uint32_t py = y0 + (uint32_t)(idx / w);
        OrcCounters k = { 0, 0, 0, 0, 0, 0 };
        /* d = pixel / size * 2 - 1  (cameraCommon.cuh:22) */
        const float dx = (float)px / (float)s->width * 2.f - 1.f;
        const float dy = (float)py / (float)s->height * 2.f - 1.f;
        const v3 direction = v3_normalize(v3_add(v3_add(v3_scale(U, dx), v3_scale(V, dy)), W));
        const uint32_t seed = orc_tea4(px * 4096u + py, subframe_id);
        const v3 rad = radiance_of_ray(&c, eye, direction, seed, &k);
        float *out = frame_rgba + 4 * ((size_t)py * s->width + px);
        out[0] = rad.x; out[1] = rad.y; out[2] = rad.z; out[3] = 1.0f;
        paths += k.paths; hits += k.box_hits; dl += k.density_lookups;
        il += k.inscatter_lookups; se += k.scatter_events; dc += k.depth_capped;
    }
    if (counters) {
        counters->paths += paths; counters->box_hits += hits; counters->density_lookups += dl;
        counters->inscatter_lookups += il; counters->scatter_events += se; counters->depth_capped += dc;
    }
}

/* The same estimator started from an arbitrary (origin, direction) -- estimateEmission,
 * pointEmissionCamera.cu:20-40: launchID is 1-D so the seed is tea<4>(id*4096 + 0, frame). */
ORC_API void orc_point_radiance(const OrcScene *s, uint32_t launch_id, uint32_t subframe_id,
                                const float origin[3], const float direction[3], float rgb_out[3],
                                OrcCounters *counters)
{
    Ctx c;
    ctx_init(&c, s);
    OrcCounters k = { 0, 0, 0, 0, 0, 0 };
    const uint32_t seed = orc_tea4(launch_id * 4096u, subframe_id);
    const v3 rad = radiance_of_ray(&c, v3_make(origin[0], origin[1], origin[2]),
                                   v3_make(direction[0], direction[1], direction[2]), seed, &k);
    rgb_out[0] = rad.x; rgb_out[1] = rad.y; rgb_out[2] = rad.z;
    if (counters) {
        counters->paths += k.paths; counters->box_hits += k.box_hits;
        counters->density_lookups += k.density_lookups; counters->inscatter_lookups += k.inscatter_lookups;
        counters->scatter_events += k.scatter_events; counters->depth_capped += k.depth_capped;
    }
}

/* generatePoints (pointGeneratorCamera.cu:20-42) with uniformOnSphere / uniformOnDisc
 * (random.cuh:133-172) and firstScatterPosition (cloudFirstScatterMaterial.cu:8-29).  Seeds:
 * tea<4>(i, batch_seed) and tea<4>(i*4096, batch_seed + attempt) (the reference mixes clock() in). */
ORC_API void orc_generate_scatter_samples(const OrcScene *s, uint32_t count, uint32_t batch_seed,
                                          float *positions, float *directions)
{
    Ctx c;
    ctx_init(&c, s);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t ii = 0; ii < (int64_t)count; ii++) {
        const uint32_t i = (uint32_t)ii;
        OrcCounters k = { 0, 0, 0, 0, 0, 0 };
        uint32_t seed = orc_tea4(i, batch_seed);
        v3 out_p = v3_make(NAN, NAN, NAN), out_d = v3_make(NAN, NAN, NAN);
        for (uint32_t attempt = 0; attempt < 4096u; attempt++) {
            const float u = orc_rnd(&seed);
            const float v = orc_rnd(&seed);
            const float phi = u * ORC_PI_F * 2;
            const float cos_theta = 2 * v - 1;
            const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
            float sn, cs;
            ct_sincosf(phi, &sn, &cs);
            const v3 normal = v3_make(cs * sin_theta, sn * sin_theta, cos_theta);
            const float theta = orc_rnd(&seed) * ORC_PI_F * 2;
            const float sqrt_r = sqrtf(orc_rnd(&seed));
            float st, ct;
            ct_sincosf(theta, &st, &ct);
            const v3 disc = onb_inverse_transform(normal, v3_make(sqrt_r * ct, 0.0f, sqrt_r * st));
            const float disc_radius = sqrtf(3.0f) / 2;
            const v3 position = v3_scale(disc, disc_radius);
            const v3 origin = v3_add(position, v3_scale(normal, 2.0f));
            const v3 rdir = v3_neg(normal);
            float t_hit;
            if (!intersect_box(&c, origin, rdir, &t_hit)) {
                continue;
            }
            v3 pos = v3_add(origin, v3_scale(rdir, t_hit));
            pos = v3_add(pos, v3_scale(c.bbox, 0.5f));
            const v3 direction = v3_normalize(rdir);
            uint32_t seed2 = orc_tea4(i * 4096u, batch_seed + attempt);
            const Event e = next_scattering_event(&c, orc_rnd(&seed2), pos, direction, &k);
            if (in_box(&c, e.pos) && e.scattered) {
                out_p = v3_sub(e.pos, v3_scale(c.bbox, 0.5f));
                out_d = rdir;
                break;
            }
        }
        positions[3 * (size_t)i + 0] = out_p.x; positions[3 * (size_t)i + 1] = out_p.y; positions[3 * (size_t)i + 2] = out_p.z;
        directions[3 * (size_t)i + 0] = out_d.x; directions[3 * (size_t)i + 1] = out_d.y; directions[3 * (size_t)i + 2] = out_d.z;
    }
}

/* ------------------------------------------------------------------------------------------
 * Hierarchical descriptor: setupHierarchicalDescriptor<DisneyDescriptor, uint8_t>
 * (src/CUDA/DisneyDescriptor.cuh:71-112) launched by DisneyDescriptorCollector::collect
 * (src/Scene/DisneyDescriptorCollector.cpp:57-63, program src/CUDA/disneyDescriptorCollector.cu:21-28).
 * 10 layers x (9 x 5 x 5) samples of the density PYRAMID (Resources.cpp:169-209) in the frame
 * (eX, eY, eZ = -light), layer support doubling, LOD rising by one per layer, faded to zero
 * outside the box, stored as uint8 (TFromFloat<uint8_t>: f * 255, truncating).
 *
 * rtTex3DLod with RT_FILTER_LINEAR for min/mag/mip (VDBCloud.cpp:128-132), restated like the
 * level-0 sampler of this file: lod clamped to [0, levels-1]; l0 = floor(lod), w = lod - l0;
 * each level trilinear with x = fma(pos, textureScale * dim_l, -0.5), clamp-to-edge;
 * result = fma(w, s1 - s0, s0) (s0 alone when w == 0).
 * ------------------------------------------------------------------------------------------ */
#define ORC_DESC_LAYERS 10
ORC_API uint32_t orc_mip_levels(const uint32_t dims[3]);
ORC_API size_t orc_generate_mipmaps(const uint8_t *level0, const uint32_t dims[3], uint8_t *out);
static inline uint32_t mip_dim_fwd(uint32_t d, uint32_t level)
{
    const uint32_t v = d >> level;
    return v ? v : 1;
}
#define ORC_DESC_LAYER_SIZE 225

static float tex3_lod(const Ctx *c, const uint8_t *pyramid, const size_t *offsets, const uint32_t dims[3],
                      uint32_t levels, v3 pos, float lod)
{
    lod = fmaxf(0.0f, lod);
    lod = fminf(lod, (float)(levels - 1));
    const float fl = floorf(lod);
    const uint32_t l0 = (uint32_t)fl;
    const float w = lod - fl;
    float s[2] = { 0.f, 0.f };
    for (uint32_t k = 0; k < 2; k++) {
        if (k == 1 && !(w > 0.0f)) {
            break;
        }
        uint32_t l = l0 + k;
        if (l > levels - 1) {
            l = levels - 1;
        }
        Tex3 t;
        t.texels = pyramid + offsets[l];
        t.nx = (int32_t)mip_dim_fwd(dims[0], l);
        t.ny = (int32_t)mip_dim_fwd(dims[1], l);
        t.nz = (int32_t)mip_dim_fwd(dims[2], l);
        t.sx = c->tscale.x * (float)t.nx;
        t.sy = c->tscale.y * (float)t.ny;
        t.sz = c->tscale.z * (float)t.nz;
        s[k] = tex3_fetch(&t, pos);
    }
    if (!(w > 0.0f)) {
        return s[0];
    }
    return fmaf(w, s[1] - s[0], s[0]);
}

static float distance_to_box(const Ctx *c, v3 pos, float voxel_size)
{
    /* DisneyDescriptor.cuh:47-55 */
    const v3 half = v3_scale(c->bbox, 0.5f);
    v3 dist = v3_sub(pos, half);
    dist = v3_make(fabsf(dist.x), fabsf(dist.y), fabsf(dist.z));
    const float hv = voxel_size * 0.5f;
    const v3 corner = v3_make(fmaxf(half.x - hv, 0.f), fmaxf(half.y - hv, 0.f), fmaxf(half.z - hv, 0.f));
    dist = v3_sub(dist, corner);
    dist = v3_make(fmaxf(dist.x, 0.f), fmaxf(dist.y, 0.f), fmaxf(dist.z, 0.f));
    return sqrtf(v3_dot(dist, dist));
}

/* positions are "world" positions as the ScatterSample records hold them (box centred at 0);
 * out: count x 10 x 225 bytes, layer-major, then z (9), y (5), x (5). */
ORC_API void orc_collect_descriptors(const OrcScene *s, const float *positions, const float *directions,
                                     uint32_t count, uint8_t *out)
{
    Ctx c;
    ctx_init(&c, s);
    const uint32_t levels = orc_mip_levels(s->dims);
    size_t total = 0, offsets[32];
    for (uint32_t l = 0; l < levels; l++) {
        offsets[l] = total;
        total += (size_t)mip_dim_fwd(s->dims[0], l) * mip_dim_fwd(s->dims[1], l) * mip_dim_fwd(s->dims[2], l);
    }
    uint8_t *pyramid = (uint8_t *)malloc(total);
    orc_generate_mipmaps(s->density, s->dims, pyramid);
    const float fx = (float)s->dims[0], fy = (float)s->dims[1], fz = (float)s->dims[2];
    const float maxs = fmaxf(fmaxf(fx, fy), fz);
    const float voxel_m = s->cloud_size_m / maxs;                 /* VDBCloud.cpp:35-41 */
    const float voxel_fp = voxel_m / s->mean_free_path_m;         /* VDBCloud.cpp:43-46 */
    const float level0 = -ct_log2f(voxel_fp) - 1;                 /* DisneyDescriptor.cuh:83 */
    const v3 ez = v3_normalize(v3_neg(c.light_dir));
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t ii = 0; ii < (int64_t)count; ii++) {
        const v3 world = v3_make(positions[3 * ii], positions[3 * ii + 1], positions[3 * ii + 2]);
        const v3 view = v3_make(directions[3 * ii], directions[3 * ii + 1], directions[3 * ii + 2]);
        const v3 ex = v3_normalize(v3_cross(ez, view));
        const v3 ey = v3_cross(ex, ez);
        const v3 origin = v3_add(world, v3_scale(c.bbox, 0.5f));
        float scale = 0.5f / c.density_multiplier;
        float lod = level0;
        uint8_t *o = out + (size_t)ii * ORC_DESC_LAYERS * ORC_DESC_LAYER_SIZE;
        for (int layer = 0; layer < ORC_DESC_LAYERS; layer++) {
            const float mip_voxel = ct_powf(2.0f, lod) * voxel_m / s->cloud_size_m;
            int sample = 0;
            for (int z = -2; z <= 6; z++) {
                for (int y = -2; y <= 2; y++) {
                    for (int x = -2; x <= 2; x++) {
                        const v3 dir = v3_add(v3_add(v3_scale(ex, (float)x), v3_scale(ey, (float)y)), v3_scale(ez, (float)z));
                        const v3 pos = v3_add(origin, v3_scale(dir, scale));
                        float density = tex3_lod(&c, pyramid, offsets, s->dims, levels, pos, lod);
                        const float distance = distance_to_box(&c, pos, mip_voxel);
                        const float t = fminf(fmaxf(distance / mip_voxel, 0.0f), 1.0f); /* saturate */
                        density = density + t * (0.0f - density);                      /* optix lerp */
                        o[layer * ORC_DESC_LAYER_SIZE + sample] = (uint8_t)(density * 255.0f);
                        sample++;
                    }
                }
            }
            scale *= 2;
            lod += 1;
        }
    }
    free(pyramid);
}

/* Gpu::PointRadianceTask (src/CUDA/PointRadianceTask.h:12-78), 40 bytes. */
typedef struct OrcPointTask {
    int32_t id;
    uint32_t experimentCount;
    float radiance;
    float runningVariance;
    float position[3];
    float direction[3];
} OrcPointTask;

/* addExperimentResult, PointRadianceTask.h:40-51 (newWeight = 1.0 / N is a double division). */
static void point_task_add(OrcPointTask *t, float new_radiance)
{
    t->experimentCount++;
    const float N = (float)t->experimentCount;
    const float new_weight = (float)(1.0 / (double)N);
    const float previous_mean = t->radiance;
    const float new_mean = t->radiance + (new_radiance - previous_mean) * new_weight;
    t->radiance = new_mean;
    t->runningVariance += (new_radiance - previous_mean) * (new_radiance - new_mean);
}

/* `launches` launches of estimateEmission (pointEmissionCamera.cu:20-40) over tasks[0..count):
 * frame = first_frame .. first_frame+launches-1, launchID = task index. */
ORC_API void orc_point_radiance_launch(const OrcScene *s, OrcPointTask *tasks, uint32_t count,
                                       uint32_t first_frame, uint32_t launches, OrcCounters *counters, int32_t threads)
{
    Ctx c;
    ctx_init(&c, s);
    uint64_t paths = 0, hits = 0, dl = 0, il = 0, se = 0, dc = 0;
#ifdef _OPENMP
    if (threads <= 0) {
        threads = omp_get_max_threads();
    }
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads) reduction(+ : paths, hits, dl, il, se, dc)
    for (int64_t i = 0; i < (int64_t)count; i++) {
        OrcCounters k = { 0, 0, 0, 0, 0, 0 };
        OrcPointTask *t = &tasks[i];
        for (uint32_t f = 0; f < launches; f++) {
            const uint32_t seed = orc_tea4((uint32_t)i * 4096u, first_frame + f);
            const v3 rad = radiance_of_ray(&c, v3_make(t->position[0], t->position[1], t->position[2]),
                                           v3_make(t->direction[0], t->direction[1], t->direction[2]), seed, &k);
            point_task_add(t, rad.x);
        }
        paths += k.paths; hits += k.box_hits; dl += k.density_lookups;
        il += k.inscatter_lookups; se += k.scatter_events; dc += k.depth_capped;
    }
    if (counters) {
        counters->paths += paths; counters->box_hits += hits; counters->density_lookups += dl;
        counters->inscatter_lookups += il; counters->scatter_events += se; counters->depth_capped += dc;
    }
}

/* PointRadianceTask::operator+= (:56-68): merge of a replica; note that the reference adds the
 * two M2 values without the between-means term.  Returns -1 when the ids differ. */
ORC_API int32_t orc_point_task_merge(OrcPointTask *into, const OrcPointTask *other)
{
    if (other->id != into->id) {
        return -1;
    }
    const float new_weight = other->experimentCount * 1.0f / (into->experimentCount + other->experimentCount);
    into->radiance += (other->radiance - into->radiance) * new_weight;
    into->runningVariance += other->runningVariance;
    into->experimentCount += other->experimentCount;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Shadow volume: inScatter, src/CUDA/inScatter.cu:40-66 (host side VDBCloud.cpp:57-86).
 * Sample THEN step; marches 1/sampleStep steps toward -lightDirection; early out when
 * T*255 < 1; stores uchar(T*255) truncating.  `s->inscatter` is ignored.
 * ------------------------------------------------------------------------------------------ */
/* one texel of the shadow volume: inScatter.cu:40-66 for launch index (x, y, z) */
static uint8_t inscatter_texel(const Ctx *c, int32_t x, int32_t y, int32_t z)
{
    const int32_t nx = c->density.nx, ny = c->density.ny, nz = c->density.nz;
    const int32_t maxn = nx > ny ? (nx > nz ? nx : nz) : (ny > nz ? ny : nz);
    const float max_size = (float)maxn;
    const float min_scale = fminf(fminf(c->tscale.x, c->tscale.y), c->tscale.z);
    const v3 step_to_light = v3_scale(v3_neg(v3_normalize(c->light_dir)), c->sample_step);
    const int step_count = (int)(1 / c->sample_step);
    OrcCounters k = { 0, 0, 0, 0, 0, 0 };
    v3 p = v3_make((float)x / max_size, (float)y / max_size, (float)z / max_size);
    p = v3_div(p, min_scale);
    float transmittance = 1;
    for (int i = 0; i < step_count; i++) {
        const float density = sample_cloud(c, p, &k) * c->density_multiplier;
        const float extinction = density * c->sample_step;
        transmittance *= ct_expf(-extinction);
        p = v3_add(p, step_to_light);
        if (transmittance * 255.f < 1.f) {
            break;
        }
    }
    return (uint8_t)(transmittance * 255.f);
}

ORC_API void orc_inscatter(const OrcScene *s, uint8_t *out, int32_t threads)
{
    Ctx c;
    ctx_init(&c, s);
    const int32_t nx = c.density.nx, ny = c.density.ny, nz = c.density.nz;
#ifdef _OPENMP
    if (threads <= 0) {
        threads = omp_get_max_threads();
    }
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int32_t z = 0; z < nz; z++) {
        for (int32_t y = 0; y < ny; y++) {
            for (int32_t x = 0; x < nx; x++) {
                out[((size_t)z * ny + y) * nx + x] = inscatter_texel(&c, x, y, z);
            }
        }
    }
}

/* The same for a list of texels (xyz[3i..3i+2]): what the 256^3 / 512^3 / 1024^3 shadow-volume tests compare the
 * product's inscatter_kernel with, texel by texel, without a CPU pass over the whole volume. */
ORC_API void orc_inscatter_texels(const OrcScene *s, const uint32_t *xyz, uint64_t count, uint8_t *out, int32_t threads)
{
    Ctx c;
    ctx_init(&c, s);
#ifdef _OPENMP
    if (threads <= 0) {
        threads = omp_get_max_threads();
    }
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)count; i++) {
        out[i] = inscatter_texel(&c, (int32_t)xyz[3 * i], (int32_t)xyz[3 * i + 1], (int32_t)xyz[3 * i + 2]);
    }
}

/* ------------------------------------------------------------------------------------------
 * Camera frame: sutil::calculateCameraVariables, src/Util/sutil.cpp:501-524, called with
 * fov_is_vertical = false by Camera::updatePosition (Camera.cpp:100-134).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_camera_variables(const float eye[3], const float lookat[3], const float up[3],
                                  float fov, float aspect_ratio, float U[3], float V[3], float W[3])
{
    const v3 e = v3_make(eye[0], eye[1], eye[2]);
    const v3 w = v3_sub(v3_make(lookat[0], lookat[1], lookat[2]), e);
    const float wlen = sqrtf(v3_dot(w, w));
    v3 u = v3_normalize(v3_cross(w, v3_make(up[0], up[1], up[2])));
    v3 v = v3_normalize(v3_cross(u, w));
    const float ulen = wlen * tanf(0.5f * fov * ORC_PI_F / 180.0f);
    u = v3_scale(u, ulen);
    const float vlen = ulen / aspect_ratio;
    v = v3_scale(v, vlen);
    U[0] = u.x; U[1] = u.y; U[2] = u.z;
    V[0] = v.x; V[1] = v.y; V[2] = v.z;
    W[0] = w.x; W[1] = w.y; W[2] = w.z;
}

/* ------------------------------------------------------------------------------------------
 * Volume quantiser + mip pyramid: Resources::loadVolumeBuffer Resources.cpp:92-141 and
 * generateMipmaps :169-209.  `grid` is the dense payload (active bbox), x fastest.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_quantize_volume(const float *grid, const uint32_t pd[3], uint8_t *tex)
{
    const size_t n = (size_t)pd[0] * pd[1] * pd[2];
    double max_density = grid[0];
    for (size_t i = 1; i < n; i++) {
        if (grid[i] > max_density) {
            max_density = grid[i];
        }
    }
    const uint32_t tx = pd[0] + 2, ty = pd[1] + 2, tz = pd[2] + 2;
    memset(tex, 0, (size_t)tx * ty * tz);
    for (uint32_t z = 0; z < pd[2]; z++) {
        for (uint32_t y = 0; y < pd[1]; y++) {
            for (uint32_t x = 0; x < pd[0]; x++) {
                const float v = grid[((size_t)z * pd[1] + y) * pd[0] + x];
                tex[((size_t)(z + 1) * ty + (y + 1)) * tx + (x + 1)] = (uint8_t)(v / max_density * 255);
            }
        }
    }
}

static inline uint32_t mip_dim(uint32_t d, uint32_t level)
{
    const uint32_t v = d >> level;
    return v ? v : 1;
}

ORC_API uint32_t orc_mip_levels(const uint32_t dims[3])
{
    uint32_t m = dims[0] > dims[1] ? dims[0] : dims[1];
    m = m > dims[2] ? m : dims[2];
    uint32_t levels = 1;
    while (m /= 2) {
        levels++;
    }
    return levels;
}

/* out must hold sum over levels of the level sizes; returns total bytes written. */
ORC_API size_t orc_generate_mipmaps(const uint8_t *level0, const uint32_t dims[3], uint8_t *out)
{
    const uint32_t levels = orc_mip_levels(dims);
    size_t off = 0;
    uint32_t px = dims[0], py = dims[1], pz = dims[2];
    memcpy(out, level0, (size_t)px * py * pz);
    const uint8_t *prev = out;
    off += (size_t)px * py * pz;
    for (uint32_t level = 1; level < levels; level++) {
        const uint32_t cx = mip_dim(dims[0], level), cy = mip_dim(dims[1], level), cz = mip_dim(dims[2], level);
        uint8_t *cur = out + off;
        for (uint32_t z = 0; z < cz; z++) {
            for (uint32_t y = 0; y < cy; y++) {
                for (uint32_t x = 0; x < cx; x++) {
                    uint16_t acc = 0;
                    for (uint32_t d = 0; d < 8; d++) {
                        const uint32_t sx = x * 2 + (d & 1), sy = y * 2 + ((d >> 1) & 1), sz = z * 2 + (d >> 2);
                        if (sx < px && sy < py && sz < pz) { /* TextureView3D::get returns 0 out of range */
                            acc = (uint16_t)(acc + prev[((size_t)sz * py + sy) * px + sx]);
                        }
                    }
                    cur[((size_t)z * cy + y) * cx + x] = (uint8_t)(acc / 8);
                }
            }
        }
        prev = cur;
        off += (size_t)cx * cy * cz;
        px = cx; py = cy; pz = cz;
    }
    return off;
}

/* ------------------------------------------------------------------------------------------
 * Progressive accumulation: updateFrameResult, src/CUDA/progressive.cu:17-27 (n = 1-based
 * subframeId, all four channels).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_accumulate(const float *frame, float *mean, float *m2, uint32_t subframe_id, size_t pixels)
{
    const float new_weight = 1.0f / (float)subframe_id;
    for (size_t i = 0; i < pixels * 4; i++) {
        const float nr = frame[i];
        const float pm = mean[i];
        const float nm = pm + (nr - pm) * new_weight;
        mean[i] = nm;
        m2[i] = m2[i] + (nr - pm) * (nr - nm);
    }
}

/* ------------------------------------------------------------------------------------------
 * Tonemap: reinhard.cu:20-84 (firstPass column sums in y order, secondPass serial sum over
 * columns / totalPixels, applyReinhard).  clamp(f,0,1) = fmaxf(0, fminf(f,1)) maps the NaN of
 * black pixels (0 * (0/0)) to 1.
 * ------------------------------------------------------------------------------------------ */
static inline float luminance4(const float *c)
{
    return c[0] * 0.265068f + c[1] * 0.67023428f + c[2] * 0.06409157f + c[3] * 0.0f;
}

ORC_API float orc_reinhard(const float *mean, uint32_t width, uint32_t height, float exposure, uint8_t *screen)
{
    const float DELTA = 0.00001f;
    float *cols = (float *)malloc(sizeof(float) * width);
    for (uint32_t x = 0; x < width; x++) {
        float sum = 0;
        for (uint32_t y = 0; y < height; y++) {
            const float lum = luminance4(mean + 4 * ((size_t)y * width + x));
            sum += lum + DELTA;
        }
        cols[x] = sum;
    }
    float result = 0;
    for (uint32_t i = 0; i < width; i++) {
        result += cols[i];
    }
    free(cols);
    const uint32_t total_pixels = width * height;
    result = result / (float)total_pixels;
    const float avg = result;
    const float inv_gamma = 1.f / 2.2f;
    for (size_t i = 0; i < (size_t)width * height; i++) {
        const float *color = mean + 4 * i;
        const float lw = luminance4(color);
        float ld = lw * exposure / avg;
        ld = ld / (1.f + ld);
        const float sc = ld / lw;
        float rgb[3];
        for (int ch = 0; ch < 3; ch++) {
            float v = color[ch] * sc;
            v = fmaxf(0.f, fminf(v, 1.f));
            v = ct_powf(v, inv_gamma);
            rgb[ch] = v * 255;
        }
        screen[4 * i + 0] = (uint8_t)rgb[0];
        screen[4 * i + 1] = (uint8_t)rgb[1];
        screen[4 * i + 2] = (uint8_t)rgb[2];
        screen[4 * i + 3] = 255;
    }
    return avg;
}

/* ------------------------------------------------------------------------------------------
 * Convergence: Camera::isConverged, src/Scene/Cameras/Camera.cpp:232-268 (channel x only).
 * Returns 1 if converged; *unconverged gets the count of pixels outside the interval.
 * ------------------------------------------------------------------------------------------ */
ORC_API int32_t orc_is_converged(const float *mean, const float *m2, uint32_t subframe_id, size_t pixels,
                                 uint64_t *unconverged)
{
    if (subframe_id < 100) {
        if (unconverged) {
            *unconverged = pixels;
        }
        return 0;
    }
    size_t converged = 0;
    for (size_t id = 0; id < pixels; id++) {
        const float var = m2[4 * id];
        const float N = (float)subframe_id;
        const float sigma = sqrtf(var / N);
        const float abs_ci = 1.96f * sigma / sqrtf(N);
        const float rel_ci = abs_ci / (mean[4 * id] + 1.1920929e-07f /* FLT_EPSILON */);
        if (rel_ci < 0.02f || abs_ci < 1e-2f) {
            converged++;
        }
    }
    if (unconverged) {
        *unconverged = pixels - converged;
    }
    return (pixels - converged) < 500;
}

/* The literal 16-step bisection of getNewDirection (cloud.cuh:162-180) for `count` consecutive
 * 24-bit randoms: k_out[i] = l * 65536 after the loop, i.e. cos(theta) = (2k+1)/65536 - 1.
 * Used to check the product's guide-table inversion exhaustively. */
ORC_API void orc_cdf_bisect_k(const float *cdf, uint32_t n, uint32_t first_u24, uint32_t count, uint32_t *k_out,
                              int32_t threads)
{
#ifdef _OPENMP
    if (threads <= 0) {
        threads = omp_get_max_threads();
    }
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)count; i++) {
        const float val = (float)(first_u24 + (uint32_t)i) / (float)0x01000000;
        float l = 0.f, r = 1.f, m;
        for (int it = 0; it < 16; it++) {
            m = (l + r) / 2.f;
            if (val > orc_tex1d(cdf, n, m)) {
                l = m;
            } else {
                r = m;
            }
        }
        k_out[i] = (uint32_t)(l * 65536.0f);
    }
}

/* Thin wrappers so tests can drive the deterministic math and texture units directly. */
ORC_API float orc_expf(float x) { return ct_expf(x); }
ORC_API float orc_logf(float x) { return ct_logf(x); }
ORC_API float orc_powf(float x, float y) { return ct_powf(x, y); }
ORC_API void orc_sincosf(float x, float *s, float *c) { ct_sincosf(x, s, c); }
ORC_API float orc_tex3d(const uint8_t *texels, const uint32_t dims[3], const float pos_box[3])
{
    OrcScene s;
    memset(&s, 0, sizeof s);
    s.dims[0] = dims[0]; s.dims[1] = dims[1]; s.dims[2] = dims[2];
    s.density = texels; s.inscatter = texels;
    s.cloud_size_m = 1; s.mean_free_path_m = 1; s.sample_step = 1; s.light_direction[2] = 1;
    Ctx c;
    ctx_init(&c, &s);
    return tex3_fetch(&c.density, v3_make(pos_box[0], pos_box[1], pos_box[2]));
}
ORC_API int32_t orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
