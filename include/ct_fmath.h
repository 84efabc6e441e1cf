/*
 * ct_fmath.h -- the numeric contract of the cloud path tracer.
 *
 * Deterministic IEEE-754 binary32 elementary functions.  Every operation is a
 * correctly rounded +,-,*,/ or an explicit fmaf(), so the same source gives
 * bit-identical results when compiled by gcc for the host CPU and by hipcc for
 * gfx950 -- PROVIDED both are compiled with -ffp-contract=off (no implicit
 * contraction) and without fast-math.  This replaces the third-party
 * arithmetic the reference relies on (CUDA libm under --use_fast_math:
 * DeepestScatter_DataGen.vcxproj:320; call sites cloud.cuh:93,99,
 * random.cuh:124-128, reinhard.cu:74-76), which cannot be reproduced off an
 * NVIDIA GPU.  Kernels use these (not v_exp_f32 & friends) so that a path takes
 * the same branches on the GPU as in the CPU oracle at a fixed seed.
 *
 * Polynomials are the classic single-precision Cephes minimax sets; measured
 * error vs. double libm is <= 2 ulp over the ranges the tracer uses
 * (tests/test_fmath.py).
 *
 * Usable from C99, C++ and HIP device code.
 */
#ifndef CT_FMATH_H
#define CT_FMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define CT_FN __host__ __device__ static inline
#else
#define CT_FN static inline
#endif

CT_FN float ct_bits_to_float(uint32_t u)
{
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
}

CT_FN uint32_t ct_float_to_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
}

/* e^x.  Cody-Waite reduction by ln2 (two-term), degree-5 polynomial, exact 2^n scale.
 * Results below the normal range flush to 0 (the tracer never needs them). */
CT_FN float ct_expf(float x)
{
    if (!(x > -87.0f)) {
        return (x != x) ? x : 0.0f;
    }
    if (x > 88.0f) {
        return INFINITY;
    }
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
    const int32_t ni = (int32_t)n;
    return y * ct_bits_to_float((uint32_t)(ni + 127) << 23);
}

/* Natural logarithm, x > 0 (x == 0 -> -inf, x < 0 or NaN -> NaN). */
CT_FN float ct_logf(float x)
{
    if (!(x > 0.0f)) {
        return (x == 0.0f) ? -INFINITY : NAN;
    }
    if (x == INFINITY) {
        return x;
    }
    int32_t e = 0;
    uint32_t u = ct_float_to_bits(x);
    if (u < 0x00800000u) { /* subnormal: renormalise exactly */
        x = x * 16777216.0f;
        u = ct_float_to_bits(x);
        e = -24;
    }
    e += (int32_t)(u >> 23) - 126;
    float m = ct_bits_to_float((u & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
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

/* x^y for x >= 0 as exp(y*log x); x == 0 -> 0 (y > 0 assumed).
 * Only used by the display tonemap (reinhard.cu:74-76). */
/* log2(x) = ln(x) * (1/ln 2), one extra rounding (used for mip levels only) */
CT_FN float ct_log2f(float x)
{
    return ct_logf(x) * 1.44269504088896341f;
}

CT_FN float ct_powf(float x, float y)
{
    if (!(x > 0.0f)) {
        return (x == 0.0f) ? 0.0f : NAN;
    }
    return ct_expf(y * ct_logf(x));
}

/* sin and cos of x for |x| <= 8 (the tracer passes phi in [0, 2pi)).
 * Octant reduction by pi/4 in three Cody-Waite terms, Cephes sinf/cosf kernels. */
CT_FN void ct_sincosf(float x, float *s_out, float *c_out)
{
    const float ax = fabsf(x);
    int32_t j = (int32_t)(ax * 1.27323954473516f); /* floor(|x| / (pi/4)) */
    j += (j & 1);                                   /* to even octant boundary */
    const float y = (float)j;
    float r = fmaf(y, -0.78515625f, ax);
    r = fmaf(y, -2.4187564849853515625e-4f, r);
    r = fmaf(y, -3.77489497744594108e-8f, r);
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sn = fmaf(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cs = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int32_t q = (j >> 1) & 3; /* quadrant of the reduced angle */
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    if (q & 2) {
        s = -s;
    }
    if ((q + 1) & 2) {
        c = -c;
    }
    if (x < 0.0f) {
        s = -s;
    }
    *s_out = s;
    *c_out = c;
}

#endif /* CT_FMATH_H */
