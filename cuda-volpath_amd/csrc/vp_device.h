// vp_device.h -- device-side building blocks of the gfx950 radiance integrator.
//
// Arithmetic contract: binary32, no contraction, left-to-right; texture fetches of the reference
// (tex3D / tex2D, kernel.cu:173-178, :971) are explicit loads + ALU filtering; see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vp_math.h"

namespace vp
{
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }  // helper_math.h:997-1000
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b)
{
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ f3 normalize(f3 v) { return v * (1.0f / __builtin_sqrtf(dot(v, v))); }
__device__ __forceinline__ float max3(f3 v) { return fmaxf(fmaxf(v.x, v.y), v.z); }
__device__ __forceinline__ float min3(f3 v) { return fminf(fminf(v.x, v.y), v.z); }

// vecmath.h:9-16 evaluated in float, as the reference's constexpr does
#define VP_PI_F 3.1415926535897932384626422832795028841971f
constexpr float kPi     = VP_PI_F;
constexpr float kPi2    = VP_PI_F / 2.0f;
constexpr float k1Pi    = 1.0f / VP_PI_F;
constexpr float k1TwoPi = 1.0f / (VP_PI_F * 2.0f);
constexpr float kTwoPi  = VP_PI_F * 2.0f;
constexpr float k1TwoPiPi = 1.0f / VP_PI_F / (VP_PI_F * 2.0f);  // vecmath.h:16

// Param of the reference (src/param.h:4-12); 44 bytes
struct ParamDev
{
    unsigned width, height;
    float    density, brightness;
    float    albedo[3];
    float    g;
    float    sigma_t[3];
};

// Everything the integrator reads besides Param; uniform (kernel argument, scalar loads)
struct SceneDev
{
    const uint2*         cells_u8;   // quantized volume: the 2x2x2 texel neighbourhood of each voxel in 8 bytes
    const float*         cells_f32;  // float volume: the same neighbourhood as 8 floats
    const unsigned char* bounds_u8;  // (max,min) byte pairs per brick
    const float*         bounds_f32; // (max,min) float pairs per brick
    const float*         opacity;    // optical depth toward the sun, N^3 floats (or null): what precompute_opacity computes (vp_get_opacity)
    const float*         opacity_cells;  // ... the same values as per-voxel 2x2x2 neighbourhood cells of 8 floats (pack_cells_f32_k): what
                                     // the integrator reads -- one 32-byte cell instead of eight scattered floats (four cache lines)
    const float4*        env;        // lat-long environment, row 0 = zenith
    int   nx, ny, nz, linear;
    int   cell_bricks;   // layout of the packed cells: 0 = x fastest over the whole grid; 1 = 4x4x4 bricks of 64 cells (512 B of uchar
                         // cells: four cache lines per brick), bricks x fastest -- cell_index()
    int   brick_shift, bnx, bny, bnz;
    int   env_w, env_h;
    float bmin[3], bmax[3], linv[3];
    float sun_dir[3], sun_power[3], sun_orig[3];
    float sun_cos;   // 94 / sqrt(94^2 + 0.45^2), kernel.cu:1263
    float cam[12];   // row-major 3x4 camera-to-world, kernel.cu:626
    float cam_z;     // -1 / tan(54.43 * 0.00872664626), kernel.cu:1985
    // active environment sampling (!PASSIVE_ENVMAP): luminance CDFs (kernel.cu:1144-1210) and HDRpdfnormAlt
    const float* env_cdf_y;  // env_h
    const float* env_cdf_x;  // env_w * env_h
    float        env_pdfnorm_alt;
};

// ------------------------------------------------------------------------------ RNG
// sampler.h:3-11
__device__ __forceinline__ unsigned wang_hash(unsigned seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

// sampler.h-compatible stream (parity mode; quirk Q2)
struct RngSamplerH
{
    unsigned sx, sy;
    __device__ __forceinline__ void init(unsigned px, unsigned py, unsigned frame, unsigned, unsigned)
    {
        sx = wang_hash((px << 16) | py);
        sy = wang_hash(frame);
        word();
    }
    __device__ __forceinline__ unsigned word()
    {
        unsigned result = sx * 0x9e3779bbu;
        sy ^= sx;
        sx = ((sx << 26) | (sx >> 6)) ^ sy ^ (sy << 9);
        sy = (sx << 13) | (sx >> 19);
        return result;
    }
    __device__ __forceinline__ float next() { return u2f(0x3f800000u | (word() >> 9)) - 1.0f; }
    // the reference's stream is sequential: the pair tags of the counter-based generator mean nothing here
    __device__ __forceinline__ float next_a() { return next(); }
    __device__ __forceinline__ float next_b() { return next(); }
    // discard n draws (values that cannot influence the path: the restart crawl in front of the volume)
    __device__ __forceinline__ void skip(unsigned n) { for (unsigned i = 0; i < n; i++) word(); }
    // the reference's stream is sequential: a shadow ray draws from it like everything else
    static constexpr bool kShadowSubstream = false;
    __device__ __forceinline__ unsigned enter_shadow(unsigned) { return 0u; }
    __device__ __forceinline__ void leave_shadow(unsigned) {}
    __device__ __forceinline__ void set_pair(unsigned) {}
    // where the stream stands, in two words (approach_k hands a path over to render_k)
    __device__ __forceinline__ void save(unsigned& a, unsigned& b) const { a = sx; b = sy; }
    __device__ __forceinline__ void load(unsigned a, unsigned b) { sx = a; sy = b; }
};

// Philox2x32-R (Salmon et al., SC'11; R = 10 or 7), numbered in PAIRS of draws: next_a() computes
// philox2x32_10(counter = (pair index, x<<16|y), key = (frame ^ k0) + k1) and returns word 0, next_b() returns
// word 1 of the same block.  The integrator draws (free flight, collision test) and the two phase-function
// variates as such pairs, so the ten multiply rounds sit at one place per tracking step, for the whole wave,
// with no buffer bookkeeping; a pair whose second word is not needed just drops it.
template <int ROUNDS>
struct RngPhiloxR
{
    unsigned pix, key, pair;  // next pair index to generate
    unsigned w1;              // second word of the current pair
    __device__ __forceinline__ void init(unsigned px, unsigned py, unsigned frame, unsigned key0, unsigned key1)
    {
        pix = (px << 16) | py; key = (frame ^ key0) + key1; pair = 0; w1 = 0;
    }
    __device__ __forceinline__ float next_a()
    {
        unsigned c0 = pair, c1 = pix, k = key;
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
        {
            unsigned long long p = (unsigned long long)0xD256D193u * c0;
            unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p >> 32), k, c1, 0x96);  // hi ^ k ^ c1 in one instruction (gfx950 V_BITOP3_B32)
            c1 = (unsigned)p;
            c0 = n0;
            k += 0x9E3779B9u;
        }
        w1 = c1;
        pair++;
        return u2f(0x3f800000u | (c0 >> 9)) - 1.0f;
    }
    __device__ __forceinline__ float next_b() { return u2f(0x3f800000u | (w1 >> 9)) - 1.0f; }
    // discard n pairs: the counter moves, nothing is computed
    __device__ __forceinline__ void skip(unsigned n) { pair += n; }
    // A shadow ray draws from a sub-stream of its own: pair indices 0x80000000 + (id << 20) + 0, 1, 2, ... with
    // id = 2 * (scatter depth) + (0 sun ray, 1 environment ray of the one-sample MIS); the path's own stream goes on
    // afterwards where it stood (oracle: rng_enter_shadow).  What the path draws after a light estimate then does not depend
    // on the number of steps the estimate took, so a shadow ray may stop as soon as nothing can change its result any more.
    static constexpr bool kShadowSubstream = true;
    // (the caller keeps the position of the path's own stream while the shadow ray draws: cold state, render_k)
    __device__ __forceinline__ unsigned enter_shadow(unsigned id) { const unsigned saved = pair; pair = 0x80000000u + (id << 20); return saved; }
    __device__ __forceinline__ void leave_shadow(unsigned saved) { pair = saved; }
    // take the stream up at pair index n (approach_k has consumed the pairs before it)
    __device__ __forceinline__ void set_pair(unsigned n) { pair = n; }
    __device__ __forceinline__ void save(unsigned& a, unsigned& b) const { a = pair; b = 0u; }
    __device__ __forceinline__ void load(unsigned a, unsigned) { pair = a; }
};
typedef RngPhiloxR<10> RngPhilox;   // VP_RNG_PHILOX
typedef RngPhiloxR<7>  RngPhilox7;  // VP_RNG_PHILOX7: Random123's smallest Crush-resistant round count

// ------------------------------------------------------------------ texture fetches
#define VP_U8_TRI_SCALE 2.3374372e-10f  // fl(1/(255*2^24)): full scale -> exactly 1.0f
#define VP_U8_SCALE 0.003921569f        // fl(1/255)

__device__ __forceinline__ f3 to_local(const SceneDev& S, f3 pos)
{
    return f3{(pos.x - S.bmin[0]) * S.linv[0], (pos.y - S.bmin[1]) * S.linv[1], (pos.z - S.bmin[2]) * S.linv[2]};
}

// texel-centre split of one axis: cell index (clamped) and the 8-bit filter weight as a float w/256
// (an exact multiple of 2^-8 in [0,1])
__device__ __forceinline__ void axis_linear(float pn, int n, int& i, float& fw)
{
    float xb = fma_(pn, (float)n, -0.5f);  // the unit's own scaling: one rounding
    // below the first texel centre both taps clamp to texel 0: the same as sitting exactly on it (i = 0, w = 0)
    xb       = __builtin_fmaxf(xb, 0.0f);
    float fl = __builtin_floorf(xb);
    float fr = xb - fl;
    i        = (int)fl;
    fw       = __builtin_floorf(fma_(fr, 256.0f, 0.5f)) * (1.0f / 256.0f);  // round-to-nearest of fr*256, /256
    // the packed cell of voxel i already holds the clamped (i, i+1) pair
    i = i > n - 1 ? n - 1 : i;
}
// The same split for float texels, where the short cut above is not exact: below the first texel centre the two taps are the SAME
// texel but the weight is still the fraction of the coordinate, and a*(1-w) + a*w is not a in binary32.  `low` tells the filter
// to use the first tap twice (the packed cell of voxel 0 holds texels 0 and 1).
__device__ __forceinline__ void axis_linear_f32(float pn, int n, int& i, float& fw, bool& low)
{
    float xb = fma_(pn, (float)n, -0.5f);
    float fl = __builtin_floorf(xb);
    float fr = xb - fl;
    i        = (int)fl;
    fw       = __builtin_floorf(fma_(fr, 256.0f, 0.5f)) * (1.0f / 256.0f);
    low      = i < 0;
    i        = i < 0 ? 0 : i;
    i        = i > n - 1 ? n - 1 : i;
}
__device__ __forceinline__ int axis_point(float pn, int n)
{
    int i = (int)__builtin_floorf(pn * (float)n);
    i     = i < 0 ? 0 : i;
    i     = i > n - 1 ? n - 1 : i;
    return i;
}

__device__ __forceinline__ float lerpf(float a, float b, float w) { return a * (1.0f - w) + b * w; }

// Filter the 8 bytes of a packed cell.  Defined (oracle: sample_volume) as the EXACT integer sum
//   v = sum_t t * wx' * wy' * wz'   (weights in 0..256, v <= 255 * 2^24), result = float(v) * fl(1/(255*2^24)).
// Evaluated here in binary32 without any rounding before the last step: with weights w/256 every
// x- and y-stage value is an integer multiple of 2^-16 below 2^8 (<= 24 significant bits, exact), and
// the z-stage is ONE fma, so its result is the correctly rounded v / 2^24 = float(v) / 2^24.
// the same filter on eight texels given as floats (opacity_lds_k reads them from its LDS tile): tXYZ = texel (i+X, j+Y, k+Z)
__device__ __forceinline__ float filter_texels_u8(float t000, float t100, float t010, float t110, float t001, float t101, float t011, float t111,
                                                  float fx, float fy, float fz)
{
    float x00  = fma_(t100 - t000, fx, t000);
    float x10  = fma_(t110 - t010, fx, t010);
    float x01  = fma_(t101 - t001, fx, t001);
    float x11  = fma_(t111 - t011, fx, t011);
    float y0   = fma_(x10 - x00, fy, x00);
    float y1   = fma_(x11 - x01, fy, x01);
    float v    = fma_(y1 - y0, fz, y0);
    return v * (VP_U8_TRI_SCALE * 16777216.0f);
}
__device__ __forceinline__ float filter_cell_u8(uint2 c, float fx, float fy, float fz)
{
    // (float)(byte k of a dword) selects v_cvt_f32_ubyte<k>
    float t000 = (float)(c.x & 0xffu), t100 = (float)((c.x >> 8) & 0xffu);
    float t010 = (float)((c.x >> 16) & 0xffu), t110 = (float)(c.x >> 24);
    float t001 = (float)(c.y & 0xffu), t101 = (float)((c.y >> 8) & 0xffu);
    float t011 = (float)((c.y >> 16) & 0xffu), t111 = (float)(c.y >> 24);
    float x00  = fma_(t100 - t000, fx, t000);
    float x10  = fma_(t110 - t010, fx, t010);
    float x01  = fma_(t101 - t001, fx, t001);
    float x11  = fma_(t111 - t011, fx, t011);
    float y0   = fma_(x10 - x00, fy, x00);
    float y1   = fma_(x11 - x01, fy, x01);
    float v    = fma_(y1 - y0, fz, y0);  // = float(v_int) * 2^-24, one rounding
    return v * (VP_U8_TRI_SCALE * 16777216.0f);
}

// where the packed cell of voxel (i, j, k) lies.  The 4^3-brick order keeps the cells a ray touches over a few steps in a few
// cache lines whatever its direction (x-fastest rows serve rays along x only): for grids whose cells do not fit the caches.
__device__ __forceinline__ size_t cell_index(const SceneDev& S, int i, int j, int k)
{
    if (S.cell_bricks)
    {
        const unsigned nbx = ((unsigned)S.nx + 3u) >> 2, nby = ((unsigned)S.ny + 3u) >> 2;
        const unsigned b   = ((unsigned)i >> 2) + nbx * (((unsigned)j >> 2) + nby * ((unsigned)k >> 2));
        return ((size_t)b << 6) | (size_t)((((unsigned)k & 3u) << 4) | (((unsigned)j & 3u) << 2) | ((unsigned)i & 3u));
    }
    // dims <= 4096 (checked by init_cuda): 24-bit operands, 32-bit result
    return (size_t)((unsigned)i + __umul24((unsigned)S.nx, (unsigned)j + __umul24((unsigned)S.ny, (unsigned)k)));
}

// normalised density in [0,1] at a world position: tex3D<float>(density_tex) of kernel.cu:692
template <bool QUANT>
__device__ __forceinline__ float sample_density01(const SceneDev& S, f3 pos)
{
    f3    p = to_local(S, pos);
    int   i, j, k;
    float fx, fy, fz;
    bool  lx = false, ly = false, lz = false;
    if (S.linear)
    {
        if (QUANT)
        {
            axis_linear(p.x, S.nx, i, fx);
            axis_linear(p.y, S.ny, j, fy);
            axis_linear(p.z, S.nz, k, fz);
        }
        else
        {
            axis_linear_f32(p.x, S.nx, i, fx, lx);
            axis_linear_f32(p.y, S.ny, j, fy, ly);
            axis_linear_f32(p.z, S.nz, k, fz, lz);
        }
    }
    else
    {
        i = axis_point(p.x, S.nx);
        j = axis_point(p.y, S.ny);
        k = axis_point(p.z, S.nz);
        fx = fy = fz = 0.0f;
    }
    size_t idx = cell_index(S, i, j, k);
    if (QUANT)
    {
        uint2 c = S.cells_u8[idx];
        return filter_cell_u8(c, fx, fy, fz);
    }
    else
    {
        const float4* q  = reinterpret_cast<const float4*>(S.cells_f32) + idx * 2;
        float4        lo = q[0], hi = q[1];
        // below the first texel centre of an axis both taps are texel 0 (see axis_linear_f32)
        if (lx) { lo.y = lo.x; lo.w = lo.z; hi.y = hi.x; hi.w = hi.z; }
        if (ly) { lo.z = lo.x; lo.w = lo.y; hi.z = hi.x; hi.w = hi.y; }
        if (lz) hi = lo;
        float x00 = lerpf(lo.x, lo.y, fx);
        float x10 = lerpf(lo.z, lo.w, fx);
        float x01 = lerpf(hi.x, hi.y, fx);
        float x11 = lerpf(hi.z, hi.w, fx);
        float y0  = lerpf(x00, x10, fy);
        float y1  = lerpf(x01, x11, fy);
        return lerpf(y0, y1, fz);
    }
}

// trilinear fetch from a plain float volume (the opacity table, kernel.cu:541-542, :2187)
__device__ __forceinline__ float sample_float_volume(const SceneDev& S, const float* vol, f3 pos)
{
    f3 p = to_local(S, pos);
    int   n[3]  = {S.nx, S.ny, S.nz};
    float pn[3] = {p.x, p.y, p.z};
    int   a[3], b[3];
    float w[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++)
    {
        float xb = fma_(pn[ax], (float)n[ax], -0.5f);
        float fl = __builtin_floorf(xb);
        float fr = xb - fl;
        int   i  = (int)fl;
        w[ax]    = (float)(int)fma_(fr, 256.0f, 0.5f) * (1.0f / 256.0f);
        int i0 = i < 0 ? 0 : i;     i0 = i0 > n[ax] - 1 ? n[ax] - 1 : i0;
        int i1 = i + 1 < 0 ? 0 : i + 1; i1 = i1 > n[ax] - 1 ? n[ax] - 1 : i1;
        a[ax] = i0; b[ax] = i1;
    }
    size_t sx = 1, sy = (size_t)S.nx, sz = (size_t)S.nx * S.ny;
    float x00 = lerpf(vol[a[0] * sx + a[1] * sy + a[2] * sz], vol[b[0] * sx + a[1] * sy + a[2] * sz], w[0]);
    float x10 = lerpf(vol[a[0] * sx + b[1] * sy + a[2] * sz], vol[b[0] * sx + b[1] * sy + a[2] * sz], w[0]);
    float x01 = lerpf(vol[a[0] * sx + a[1] * sy + b[2] * sz], vol[b[0] * sx + a[1] * sy + b[2] * sz], w[0]);
    float x11 = lerpf(vol[a[0] * sx + b[1] * sy + b[2] * sz], vol[b[0] * sx + b[1] * sy + b[2] * sz], w[0]);
    float y0  = lerpf(x00, x10, w[1]);
    float y1  = lerpf(x01, x11, w[1]);
    return lerpf(y0, y1, w[2]);
}

// The same fetch from the neighbourhood-packed copy of the table (SceneDev::opacity_cells): the cell of voxel (i, j, k) holds the
// clamped taps (i..i+1, j..j+1, k..k+1), so the eight loads above become two 16-byte loads of one line.  Same taps, same weights, same
// order of the lerps: the same bits.  Below the first texel centre of an axis both taps are texel 0 (a[ax] = b[ax] = 0 above): the
// cell of voxel 0 holds texels 0 and 1, so its first tap is used twice, as in sample_density01's float path.
__device__ __forceinline__ float sample_float_cells(const SceneDev& S, const float* cells, f3 pos)
{
    f3    p = to_local(S, pos);
    int   i, j, k;
    float fx, fy, fz;
    bool  lx, ly, lz;
    axis_linear_f32(p.x, S.nx, i, fx, lx);
    axis_linear_f32(p.y, S.ny, j, fy, ly);
    axis_linear_f32(p.z, S.nz, k, fz, lz);
    const size_t  idx = (size_t)((unsigned)i + __umul24((unsigned)S.nx, (unsigned)j + __umul24((unsigned)S.ny, (unsigned)k)));
    const float4* q   = reinterpret_cast<const float4*>(cells) + idx * 2;
    float4        lo = q[0], hi = q[1];
    if (lx) { lo.y = lo.x; lo.w = lo.z; hi.y = hi.x; hi.w = hi.z; }
    if (ly) { lo.z = lo.x; lo.w = lo.y; hi.z = hi.x; hi.w = hi.y; }
    if (lz) hi = lo;
    float x00 = lerpf(lo.x, lo.y, fx);
    float x10 = lerpf(lo.z, lo.w, fx);
    float x01 = lerpf(hi.x, hi.y, fx);
    float x11 = lerpf(hi.z, hi.w, fx);
    float y0  = lerpf(x00, x10, fy);
    float y1  = lerpf(x01, x11, fy);
    return lerpf(y0, y1, fz);
}

// point-sampled (max,min) bound of the brick containing pos: vol_bound_minmax kernel.cu:1610-1624
template <bool QUANT>
__device__ __forceinline__ void sample_bound(const SceneDev& S, f3 pos, float& bmax, float& bmin)
{
    f3  p = to_local(S, pos);
    int i = axis_point(p.x, S.nx) >> S.brick_shift;
    int j = axis_point(p.y, S.ny) >> S.brick_shift;
    int k = axis_point(p.z, S.nz) >> S.brick_shift;
    size_t o = (size_t)((unsigned)i + (unsigned)S.bnx * ((unsigned)j + (unsigned)S.bny * (unsigned)k));
    if (QUANT)
    {
        unsigned short v = reinterpret_cast<const unsigned short*>(S.bounds_u8)[o];
        bmax = (float)(v & 0xffu) * VP_U8_SCALE;
        bmin = (float)(v >> 8) * VP_U8_SCALE;
    }
    else
    {
        float2 v = reinterpret_cast<const float2*>(S.bounds_f32)[o];
        bmax = v.x;
        bmin = v.y;
    }
}

// eval_envmap kernel.cu:956-973 with dir_to_uv :882-895; point sampled, clamped
__device__ __forceinline__ f3 eval_envmap(const SceneDev& S, f3 dir)
{
    float phi   = acosf_(dir.y);
    float theta = atanf_(dir.z / dir.x) + kPi2;
    if (dir.x < 0.0f) theta += kPi;
    float  u = theta * k1TwoPi;
    float  v = phi * k1Pi;
    int    i = axis_point(u, S.env_w);
    int    j = axis_point(v, S.env_h);
    float4 t = S.env[(size_t)i + (size_t)S.env_w * (size_t)j];
    return f3{t.x, t.y, t.z};
}

// luminance kernel.cu:945-953: the literals are double, so is the arithmetic
__device__ __forceinline__ float luminance(f3 c)
{
    return (float)((double)c.x * 0.2126 + (double)c.y * 0.7152 + (double)c.z * 0.0722);
}
// sample_y / sample_x kernel.cu:904-943: lower-bound binary search on a point-sampled CDF
__device__ __forceinline__ int cdf_search(const float* cdf, int n, float r)
{
    int begin = 0, end = n - 1;
    while (end > begin)
    {
        int   mid = begin + (end - begin) / 2;
        float c   = cdf[mid];
        if (c >= r) end = mid;
        else begin = mid + 1;
    }
    return begin;
}
// sample_envmap kernel.cu:979-1006 (MULT_PDF 0, PRE_WARP 1): returns the pdf, (u,v) become texel-centre coordinates
__device__ __forceinline__ float sample_envmap(const SceneDev& S, float& u, float& v, f3& c)
{
    int iy = cdf_search(S.env_cdf_y, S.env_h, v);
    int ix = cdf_search(S.env_cdf_x + (size_t)iy * (size_t)S.env_w, S.env_w, u);
    u = ((float)ix + 0.5f) / (float)S.env_w;
    v = ((float)iy + 0.5f) / (float)S.env_h;
    int    i = axis_point(u, S.env_w);
    int    j = axis_point(v, S.env_h);
    float4 t = S.env[(size_t)i + (size_t)S.env_w * (size_t)j];
    c        = f3{t.x, t.y, t.z};
    return luminance(c) * S.env_pdfnorm_alt;
}
// uv_to_dir kernel.cu:897-902
__device__ __forceinline__ f3 uv_to_dir(float u, float v)
{
    float theta = u * kTwoPi;
    float phi   = v * kPi;
    float st, ct, sp, cp;
    sincosf_(theta, st, ct);
    sincosf_(phi, sp, cp);
    return f3{sp * st, cp, sp * -ct};
}

// intersectBox kernel.cu:654-680 (quirk Q13)
__device__ __forceinline__ bool intersect_box(f3 o, f3 d, const SceneDev& S, float& tnear, float& tfar)
{
#ifndef VP_EXP_BOX_PARALLEL
    // Axis by axis -- the same expressions as below, the three divisions and their slab products one after another instead of side
    // by side (round 5).  This is where the integrator's register pressure peaks (the shadow ray's box test in the collision block);
    // in sequence it needs 72 registers instead of 79 (global majorant: seven waves again, +2.3 % on C2) and 79 instead of 87 (chromatic
    // local majorants: six waves instead of five).  profiles/experiments/r05_box_sequence.txt
    float tmn, tmx;
    {
        const float ir = 1.0f / d.x, tb = ir * (S.bmin[0] - o.x), tt = ir * (S.bmax[0] - o.x);
        tmn = fminf(tt, tb); tmx = fmaxf(tt, tb);
    }
    __builtin_amdgcn_sched_barrier(0);
    float tmn_y, tmx_y;
    {
        const float ir = 1.0f / d.y, tb = ir * (S.bmin[1] - o.y), tt = ir * (S.bmax[1] - o.y);
        tmn_y = fminf(tt, tb); tmx_y = fmaxf(tt, tb);
    }
    __builtin_amdgcn_sched_barrier(0);
    float tmn_z, tmx_z;
    {
        const float ir = 1.0f / d.z, tb = ir * (S.bmin[2] - o.z), tt = ir * (S.bmax[2] - o.z);
        tmn_z = fminf(tt, tb); tmx_z = fmaxf(tt, tb);
    }
    const float lt = max3(f3{tmn, tmn_y, tmn_z}), st_ = min3(f3{tmx, tmx_y, tmx_z});
    tnear = lt; tfar = st_;
    return st_ > lt && st_ >= 1e-3f;
#endif
    f3 invR = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    f3 tbot = invR * (f3{S.bmin[0], S.bmin[1], S.bmin[2]} - o);
    f3 ttop = invR * (f3{S.bmax[0], S.bmax[1], S.bmax[2]} - o);
    f3 tmin = f3{fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z)};
    f3 tmax = f3{fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z)};
    float largest_tmin  = max3(tmin);
    float smallest_tmax = min3(tmax);
    tnear = largest_tmin;
    tfar  = smallest_tmax;
    return smallest_tmax > largest_tmin && smallest_tmax >= 1e-3f;
}

// the same slab test with the reciprocal direction supplied (1/d is what intersectBox computes first; a ray
// keeps its direction over many restart segments, so the three IEEE divides are hoisted out)
__device__ __forceinline__ bool intersect_box_inv(f3 o, f3 invR, const SceneDev& S, float& tnear, float& tfar)
{
    f3 tbot = invR * (f3{S.bmin[0], S.bmin[1], S.bmin[2]} - o);
    f3 ttop = invR * (f3{S.bmax[0], S.bmax[1], S.bmax[2]} - o);
    f3 tmin = f3{fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z)};
    f3 tmax = f3{fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z)};
    float largest_tmin  = max3(tmin);
    float smallest_tmax = min3(tmax);
    tnear = largest_tmin;
    tfar  = smallest_tmax;
    return smallest_tmax > largest_tmin && smallest_tmax >= 1e-3f;
}

// camera ray of a pixel, kernel.cu:1977-1987 (quirk Q3: no jitter -- the same ray in every frame)
__device__ __forceinline__ void camera_ray(const SceneDev& S, unsigned width, unsigned height, unsigned px, unsigned py, f3& ro, f3& rd)
{
    float u = ((float)px * 2.0f - (float)width) / (float)width;
    float v = ((float)py * 2.0f - (float)height) / (float)width;
    ro      = f3{S.cam[3], S.cam[7], S.cam[11]};
    f3 dv   = f3{u, v, S.cam_z};
    rd      = normalize(f3{dot(dv, f3{S.cam[0], S.cam[1], S.cam[2]}), dot(dv, f3{S.cam[4], S.cam[5], S.cam[6]}), dot(dv, f3{S.cam[8], S.cam[9], S.cam[10]})});
}

// Frame kernel.cu:557-573 (fabs(n.x) > 0.1 is a DOUBLE compare: equivalent to >= 0.1f in float)
struct Frame
{
    f3 n, t, b;
    __device__ __forceinline__ explicit Frame(f3 normal)
    {
        n    = normal;
        f3 a = (__builtin_fabsf(n.x) >= 0.1f) ? f3{0.0f, 1.0f, 0.0f} : f3{1.0f, 0.0f, 0.0f};
        t    = normalize(cross(a, n));
        b    = cross(n, t);
    }
    __device__ __forceinline__ f3 to_world(f3 c) const { return (t * c.x + b * c.y) + n * c.z; }
};

// HGPhaseFunction::sample kernel.cu:580-598 (quirk Q1)
__device__ __forceinline__ f3 hg_sample_local(float g, float rnd0, float rnd1)
{
    float cos_theta;
    if (__builtin_fabsf(g) > 1e-6f)
    {
        float s   = 2.0f * rnd0 - 1.0f;
        float f   = (1.0f - g * g) / (1.0f + g * s);
        cos_theta = (0.5f / g) * (1.0f + g * g - f * f);
        cos_theta = fmaxf(0.0f, fminf(1.0f, cos_theta));
    }
    else
        cos_theta = 2.0f * rnd0 - 1.0f;
    float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
    float phi       = (2.0f * kPi) * rnd1;
    float sp, cp;
    sincosf_(phi, sp, cp);
    return f3{cp * sin_theta, sp * sin_theta, cos_theta};
}
// HGPhaseFunction::evaluate kernel.cu:600-603
__device__ __forceinline__ float hg_eval(float g, float cos_theta)
{
    return (1.0f - g * g) / ((4.0f * kPi) * pow15f_(1.0f + g * g - (2.0f * g) * cos_theta));
}

// hyperion trick kernel.cu:2039 / :1358
__device__ __forceinline__ float hyperion_s(int n_minus)
{
    return fmaxf(0.0f, fminf(1.0f, (float)n_minus * 0.066666666666666666667f));
}
}  // namespace vp
