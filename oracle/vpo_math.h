/* vpo_math.h -- TEST INFRASTRUCTURE (oracle side). Not part of the product.
 *
 * Deterministic single-precision elementary functions used by the CPU oracle.
 *
 * Why they exist: the reference integrator (src/volumeRender_kernel.cu) calls the CUDA
 * device libm (logf :2085, powf :602, sinf/cosf :596, acosf/atanf :884-891, expf :2186).
 * Those are 1-2 ulp approximations whose bits differ from glibc's and from AMD's ocml, so
 * no two platforms can agree bit-for-bit on them.  The oracle therefore pins each function
 * to one explicit sequence of IEEE-754 binary32 operations (+, -, *, /, sqrt, fma) that
 * produces identical bits on x86 and on gfx950.  The HIP product carries its own independent
 * copy of the same sequences (cuda-volpath_amd/csrc/vp_math.h); tests/test_math.py checks
 * both against float64 libm (accuracy) and against each other (bit equality).
 *
 * Polynomials: classic Cephes single-precision kernels (Moshier), public domain constants.
 * Build with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#ifndef VPO_MATH_H
#define VPO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t vpo_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    vpo_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* natural log; domain {0} U [2^-126, +inf).  vpo_logf(0) = -inf (reference relies on
 * -log(0) = +inf being harmless: sampler.h:24-28 can return exactly 0). */
static inline float vpo_logf(float x)
{
    if (x == 0.0f) return -INFINITY;
    uint32_t ix = vpo_f2u(x);
    int      e  = (int)(ix >> 23) - 127;
    float    m  = vpo_u2f((ix & 0x007fffffu) | 0x3f800000u); /* [1,2) */
    if (m > 1.41421356f) { m *= 0.5f; e += 1; }
    float r = m - 1.0f;
    float z = r * r;
    float p = 7.0376836292E-2f;
    p = fmaf(p, r, -1.1514610310E-1f);
    p = fmaf(p, r, 1.1676998740E-1f);
    p = fmaf(p, r, -1.2420140846E-1f);
    p = fmaf(p, r, 1.4249322787E-1f);
    p = fmaf(p, r, -1.6668057665E-1f);
    p = fmaf(p, r, 2.0000714765E-1f);
    p = fmaf(p, r, -2.4999993993E-1f);
    p = fmaf(p, r, 3.3333331174E-1f);
    float fe = (float)e;
    float y  = (r * z) * p;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float res = r + y;
    res = fmaf(fe, 0.693359375f, res);
    return res;
}

/* exp; used only on arguments <= 0 (transmittance, kernel.cu:2186).  Results below 2^-126
 * flush to 0. */
static inline float vpo_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    float fn = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r  = fmaf(fn, -0.693359375f, x);
    r        = fmaf(fn, 2.12194440e-4f, r);
    float z  = r * r;
    float p  = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float y = fmaf(p, z, r) + 1.0f;
    int   n = (int)fn;
    return y * vpo_u2f((uint32_t)(n + 127) << 23);
}

/* sin and cos of an angle in [0, 2*pi] (phi = 2*pi*u, kernel.cu:595-596). */
static inline void vpo_sincosf(float a, float* s, float* c)
{
    float fk = floorf(fmaf(a, 0.636619772367581343f, 0.5f)); /* nearest multiple of pi/2 */
    int   k  = (int)fk;
    float r  = fmaf(fk, -1.5703125f, a);
    r        = fmaf(fk, -4.837512969970703125e-4f, r);
    r        = fmaf(fk, -7.54978995489188216e-8f, r);
    float z  = r * r;
    float ps = -1.9515295891E-4f;
    ps = fmaf(ps, z, 8.3321608736E-3f);
    ps = fmaf(ps, z, -1.6666654611E-1f);
    float sn = fmaf(ps * z, r, r);
    float pc = 2.443315711809948E-005f;
    pc = fmaf(pc, z, -1.388731625493765E-003f);
    pc = fmaf(pc, z, 4.166664568298827E-002f);
    float cs = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
    switch (k & 3)
    {
        case 0: *s = sn;  *c = cs;  break;
        case 1: *s = cs;  *c = -sn; break;
        case 2: *s = -sn; *c = -cs; break;
        default: *s = -cs; *c = sn; break;
    }
}

/* acos on [-1,1]; arguments a hair outside (|y| of a normalised vector) are clamped. */
static inline float vpo_acosf(float x)
{
    float ax = fabsf(x);
    if (ax > 1.0f) ax = 1.0f;
    int   big = ax > 0.5f;
    float z, t;
    if (big) { z = 0.5f * (1.0f - ax); t = sqrtf(z); }
    else     { z = ax * ax;           t = ax; }
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    float as = fmaf(p * z, t, t); /* asin(t) */
    if (big)
    {
        float r = 2.0f * as;
        return x < 0.0f ? 3.14159265358979323846f - r : r;
    }
    return x < 0.0f ? 1.57079632679489661923f + as : 1.57079632679489661923f - as;
}

/* atan on the extended reals; NaN (0/0 for a direction on the +-y axis) maps to 0. */
static inline float vpo_atanf(float x)
{
    if (x != x) return 0.0f;
    float t = fabsf(x);
    float y;
    if (t > 2.414213562373095f) { y = 1.57079632679489661923f; t = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y = 0.785398163397448309616f; t = (t - 1.0f) / (t + 1.0f); }
    else y = 0.0f;
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032E-1f);
    p = fmaf(p, z, 1.99777106478E-1f);
    p = fmaf(p, z, -3.33329491539E-1f);
    y = y + fmaf(p * z, t, t);
    return x < 0.0f ? -y : y;
}

/* x^1.5 for x >= 0 (HG denominator, kernel.cu:602). */
static inline float vpo_pow15f(float x) { return x * sqrtf(x); }

#endif
