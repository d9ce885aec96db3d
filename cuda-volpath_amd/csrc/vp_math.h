// vp_math.h -- deterministic binary32 elementary functions for the gfx950 kernels.
//
// The integrator's libm calls (reference: logf kernel.cu:2085, powf :602, sinf/cosf :596,
// acosf/atanf :884-891, expf :2186) are pinned to explicit sequences of IEEE add/mul/div/sqrt/fma
// so that results do not depend on which vendor libm (CUDA, ocml, glibc) evaluates them; the
// parity tests compare this path bit for bit against an independent CPU statement of the same
// sequences.  Cephes single-precision kernels (Moshier).  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace vp
{
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }

// natural logarithm on {0} U [2^-126, inf); log(0) = -inf
__device__ __forceinline__ float logf_(float x)
{
#ifdef VP_EXP_FASTLOG
    return __builtin_amdgcn_logf(x) * 0.69314718056f;
#endif
    // mantissa folded into (sqrt(2)/2, sqrt(2)] without a select: adding 2^23 - 0x3504f4 to the bit pattern carries
    // into the exponent field exactly when the mantissa field exceeds that of fl(sqrt 2) = 0x3fb504f3
    unsigned ix = f2u(x);
    unsigned iy = ix + 0x004afb0cu;
    int      e  = (int)(iy >> 23) - 127;
    float    m  = u2f((iy & 0x007fffffu) + 0x3f3504f4u);
    float    r  = m - 1.0f;
    float z     = r * r;
    float p     = 7.0376836292E-2f;
    p           = fma_(p, r, -1.1514610310E-1f);
    p           = fma_(p, r, 1.1676998740E-1f);
    p           = fma_(p, r, -1.2420140846E-1f);
    p           = fma_(p, r, 1.4249322787E-1f);
    p           = fma_(p, r, -1.6668057665E-1f);
    p           = fma_(p, r, 2.0000714765E-1f);
    p           = fma_(p, r, -2.4999993993E-1f);
    p           = fma_(p, r, 3.3333331174E-1f);
    float fe    = (float)e;
    float y     = (r * z) * p;
    y           = fma_(fe, -2.12194440e-4f, y);
    y           = fma_(z, -0.5f, y);
    float res   = r + y;
    res         = fma_(fe, 0.693359375f, res);
    return x == 0.0f ? -__builtin_inff() : res;
}

// exponential; used on arguments <= 0; results below 2^-126 flush to 0
__device__ __forceinline__ float expf_(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return __builtin_inff();
    float fn = __builtin_floorf(fma_(x, 1.44269504088896341f, 0.5f));
    float r  = fma_(fn, -0.693359375f, x);
    r        = fma_(fn, 2.12194440e-4f, r);
    float z  = r * r;
    float p  = 1.9875691500E-4f;
    p        = fma_(p, r, 1.3981999507E-3f);
    p        = fma_(p, r, 8.3334519073E-3f);
    p        = fma_(p, r, 4.1665795894E-2f);
    p        = fma_(p, r, 1.6666665459E-1f);
    p        = fma_(p, r, 5.0000001201E-1f);
    float y  = fma_(p, z, r) + 1.0f;
    int   n  = (int)fn;
    return y * u2f((unsigned)(n + 127) << 23);
}

// sine and cosine of an angle in [0, 2*pi]
__device__ __forceinline__ void sincosf_(float a, float& s, float& c)
{
    float fk = __builtin_floorf(fma_(a, 0.636619772367581343f, 0.5f));
    int   k  = (int)fk;
    float r  = fma_(fk, -1.5703125f, a);
    r        = fma_(fk, -4.837512969970703125e-4f, r);
    r        = fma_(fk, -7.54978995489188216e-8f, r);
    float z  = r * r;
    float ps = -1.9515295891E-4f;
    ps       = fma_(ps, z, 8.3321608736E-3f);
    ps       = fma_(ps, z, -1.6666654611E-1f);
    float sn = fma_(ps * z, r, r);
    float pc = 2.443315711809948E-005f;
    pc       = fma_(pc, z, -1.388731625493765E-003f);
    pc       = fma_(pc, z, 4.166664568298827E-002f);
    float cs = fma_(pc * z, z, fma_(z, -0.5f, 1.0f));
    int   q  = k & 3;
    float a0 = (q & 1) ? cs : sn;  // |sin|
    float b0 = (q & 1) ? sn : cs;  // |cos|
    s        = (q & 2) ? -a0 : a0;
    c        = (q == 1 || q == 2) ? -b0 : b0;
}

__device__ __forceinline__ float acosf_(float x)
{
    float ax  = __builtin_fabsf(x);
    ax        = ax > 1.0f ? 1.0f : ax;
    bool  big = ax > 0.5f;
    float z   = big ? 0.5f * (1.0f - ax) : ax * ax;
    float t   = big ? __builtin_sqrtf(z) : ax;
    float p   = 4.2163199048E-2f;
    p         = fma_(p, z, 2.4181311049E-2f);
    p         = fma_(p, z, 4.5470025998E-2f);
    p         = fma_(p, z, 7.4953002686E-2f);
    p         = fma_(p, z, 1.6666752422E-1f);
    float as  = fma_(p * z, t, t);
    if (big)
    {
        float r = 2.0f * as;
        return x < 0.0f ? 3.14159265358979323846f - r : r;
    }
    return x < 0.0f ? 1.57079632679489661923f + as : 1.57079632679489661923f - as;
}

__device__ __forceinline__ float atanf_(float x)
{
    if (x != x) return 0.0f;
    float t = __builtin_fabsf(x);
    float y;
    if (t > 2.414213562373095f) { y = 1.57079632679489661923f; t = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y = 0.785398163397448309616f; t = (t - 1.0f) / (t + 1.0f); }
    else y = 0.0f;
    float z = t * t;
    float p = 8.05374449538e-2f;
    p       = fma_(p, z, -1.38776856032E-1f);
    p       = fma_(p, z, 1.99777106478E-1f);
    p       = fma_(p, z, -3.33329491539E-1f);
    y       = y + fma_(p * z, t, t);
    return x < 0.0f ? -y : y;
}

// the quotient of the collision weights (experiment hook: VP_EXP_FASTDIV replaces the IEEE divide by v_rcp_f32)
__device__ __forceinline__ float wdiv_(float a, float b)
{
#ifdef VP_EXP_FASTDIV
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}

__device__ __forceinline__ float pow15f_(float x) { return x * __builtin_sqrtf(x); }
}  // namespace vp
