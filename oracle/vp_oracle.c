/* vp_oracle.c -- TEST INFRASTRUCTURE.  Plain-C CPU restatement of the per-pixel Monte-Carlo
 * radiance integrator of RNG65536/CUDA-volpath (see vp_oracle.h for pin status and rules).
 *
 * Every function cites the reference lines it follows; "kernel.cu" = src/volumeRender_kernel.cu,
 * "host.cpp" = src/volumeRender.cpp.  Compile-time configuration restated: SPECTRAL_TRACKING=1,
 * MULTI_CHANNEL=0, SUN_LIGHT=1, PASSIVE_ENVMAP=1, PRECOMPUTE_OPACITY=1, USE_MODEL_TRANSFORM=0,
 * max_depth=800 (kernel.cu:15-34).
 *
 * Arithmetic contract (shared, by independent implementation, with the HIP product):
 *   - binary32 throughout, no contraction (-ffp-contract=off), evaluation left to right;
 *   - elementary functions from vpo_math.h; rsqrtf(x) := 1/sqrtf(x); powf(x,1.5) := x*sqrtf(x);
 *   - texture fetches restated in software (CUDA prog. guide "Texture Fetching"): texel-centre
 *     convention, clamp addressing, 8-bit fixed-point filter weights (round to nearest);
 *     uchar texels filter in exact integer arithmetic and normalise by one multiply.
 */
#include "vp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "vpo_math.h"

typedef struct { float x, y, z; } f3;

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 muls(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 divs(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); } /* helper_math.h:997-1000 */
/* helper_math.h:1248-1251, :1420-1423, :1291-1294, :1309-1313 (host branch :62-65) */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
static inline f3 normalize3(f3 v) { float inv = 1.0f / sqrtf(dot3(v, v)); return muls(v, inv); }
static inline float max_of3(f3 v) { return fmaxf(fmaxf(v.x, v.y), v.z); } /* kernel.cu:67 */
static inline float min_of3(f3 v) { return fminf(fminf(v.x, v.y), v.z); } /* kernel.cu:71 */

/* vecmath.h:9-16 -- the float constants as the reference's constexpr arithmetic produces them */
#define VP_PI 3.1415926535897932384626422832795028841971f
static const float VP_TWO_PI   = VP_PI * 2.0f;
static const float VP_PI_2     = VP_PI / 2.0f;
static const float VP_1_PI     = 1.0f / VP_PI;
static const float VP_1_TWOPI  = 1.0f / (VP_PI * 2.0f);
static const float VP_1_TWO_PI_PI = 1.0f / VP_PI / (VP_PI * 2.0f); /* vecmath.h:16 */

/* ------------------------------------------------------------------ RNG -- */
/* sampler.h:3-11 */
uint32_t vpo_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

/* Random123 philox2x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): the
 * counter-based generator north_star asks for in place of sampler.h.  One call = two 32-bit words. */
void vpo_philox2x32_r(int rounds, const uint32_t ctr[2], uint32_t key, uint32_t out[2])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], k = key;
    for (int r = 0; r < rounds; r++)
    {
        uint64_t p  = (uint64_t)0xD256D193u * c0;
        uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        c0 = n0;
        k += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}
void vpo_philox2x32_10(const uint32_t ctr[2], uint32_t key, uint32_t out[2]) { vpo_philox2x32_r(10, ctr, key, out); }
/* philox2x32-7: Random123 documents 7 as the smallest round count of this generator that passes BigCrush ("Crush-resistant");
 * VPO_RNG_PHILOX7 uses it with the same counter / key layout */
void vpo_philox2x32_7(const uint32_t ctr[2], uint32_t key, uint32_t out[2]) { vpo_philox2x32_r(7, ctr, key, out); }

typedef struct
{
    int      mode;
    uint32_t sx, sy;         /* sampler.h state */
    uint32_t ctr[2], key;    /* philox: ctr = (draw pair index, x<<16|y), key = (frame ^ seed0) + seed1 */
    uint32_t buf[2];
    uint32_t n; /* philox: pairs so far */
    uint32_t n_saved; /* philox: position of the path's own stream while a shadow ray draws from its sub-stream */
    uint32_t n_base;  /* ... and where that sub-stream began */
    uint64_t* draws;
} rng_t;

/* sampler.h:13-23 (quirk Q2: seed_y is the rotation of the NEW seed_x) */
static inline uint32_t samplerh_next(uint32_t* sx, uint32_t* sy)
{
    uint32_t result = *sx * 0x9e3779bbu;
    *sy ^= *sx;
    *sx = ((*sx << 26) | (*sx >> (32 - 26))) ^ *sy ^ (*sy << 9);
    *sy = (*sx << 13) | (*sx >> (32 - 13));
    return result;
}

/* sampler.h:35-43 */
static void rng_init(rng_t* r, const vpo_scene* S, uint32_t px, uint32_t py, uint32_t frame, uint64_t* draws)
{
    r->mode  = S->rng_mode;
    r->n     = 0;
    r->draws = draws;
    if (r->mode == VPO_RNG_SAMPLERH)
    {
        uint32_t s0 = (px << 16) | py;
        uint32_t s1 = frame;
        r->sx = vpo_hash(s0);
        r->sy = vpo_hash(s1);
        samplerh_next(&r->sx, &r->sy);
    }
    else
    {
        r->ctr[0] = 0; r->ctr[1] = (px << 16) | py;
        r->key = (frame ^ S->seed[0]) + S->seed[1];
    }
}

/* sampler.h:25-29: float in [0,1) from the top 23 bits.
 *
 * The integrator's draws come in natural pairs -- (free-flight distance, collision test) and the two
 * Henyey-Greenstein variates -- so the counter-based mode numbers PAIRS: the first draw of a pair (`second` = 0)
 * computes philox2x32_10(counter = (pair index, x<<16|y), key) and returns word 0, the second returns word 1 of
 * the same block.  A first draw that follows a first draw (a tracking step that left its segment before the
 * collision test, or the lone control-distance draw of kernel.cu:2052) simply drops the unused word.
 * sampler.h mode is the reference's sequential stream and ignores the tag. */
static inline float rng_draw(rng_t* r, int second)
{
    uint32_t w;
    if (r->mode == VPO_RNG_SAMPLERH)
        w = samplerh_next(&r->sx, &r->sy);
    else if (!second)
    {
        r->ctr[0] = r->n++;
        vpo_philox2x32_r(r->mode == VPO_RNG_PHILOX7 ? 7 : 10, r->ctr, r->key, r->buf);
        w = r->buf[0];
    }
    else
        w = r->buf[1];
    if (r->draws) (*r->draws)++;
    return vpo_u2f(0x3f800000u | (w >> 9)) - 1.0f;
}
#define rng_next_a(r) rng_draw((r), 0)
#define rng_next_b(r) rng_draw((r), 1)

/* Counter-based streams only: a shadow ray (Tr_spectral / Tr, kernel.cu:754-808 / :712-751) draws from a SUB-STREAM of its
 * own -- pair indices 0x80000000 + (id << 20) + 0, 1, 2, ... with id = 2 * (scatter depth) + (0 sun ray, 1 the environment
 * ray of the one-sample MIS) -- and the path's own stream goes on afterwards where it stood before the shadow ray.  The
 * draws the path makes after a light estimate then do not depend on how many steps that estimate took, so an
 * implementation may stop a shadow ray as soon as nothing can change its result any more (the HIP path does: a ray that
 * has reached cells which are empty all the way out).  sampler.h mode is the reference's sequential
 * stream: nothing changes there. */
static inline void rng_enter_shadow(rng_t* r, uint32_t id)
{
    if (r->mode == VPO_RNG_SAMPLERH) return;
    r->n_saved = r->n;
    r->n       = 0x80000000u + (id << 20);
    r->n_base  = r->n;
}
/* A sub-stream holds 2^20 pairs: a shadow ray that draws more would run into the sub-stream of the next scatter depth.  It would
 * take a majorant above 3e5 per unit length (the default medium: 800) -- counted here, where every shadow ray is walked to its end,
 * so that the tests can say it never happened (vpo_debug_shadow_overflow). */
static uint64_t g_shadow_overflow = 0;
uint64_t vpo_debug_shadow_overflow(void) { return __atomic_load_n(&g_shadow_overflow, __ATOMIC_RELAXED); }
static inline void rng_leave_shadow(rng_t* r)
{
    if (r->mode == VPO_RNG_SAMPLERH) return;
    if (r->n - r->n_base > (1u << 20)) __atomic_fetch_add(&g_shadow_overflow, 1, __ATOMIC_RELAXED);
    r->n = r->n_saved;
}

void vpo_rng_stream(int mode, uint32_t x, uint32_t y, uint32_t frame, uint32_t k0, uint32_t k1, int n, float* out)
{
    vpo_scene S;
    memset(&S, 0, sizeof S);
    S.rng_mode = mode; S.seed[0] = k0; S.seed[1] = k1;
    rng_t r;
    rng_init(&r, &S, x, y, frame, NULL);
    for (int i = 0; i < n; i++) out[i] = rng_draw(&r, i & 1);
}

/* ------------------------------------------------------- Julia voxeliser -- */
/* kernel.cu:84-140: quaternion Julia q <- q^2 + c, c=(-0.2,0.8,0,0), radius 1.4, maxIter 30,
 * density = (iter > maxIter*0.9).  SURVEY S2: voxelised at texel centres over [-1,1]^3. */
static float julia_density(f3 pos)
{
    const float radius = 1.4f;
    const int   maxIter = 30;
    float qx = pos.x * radius, qy = pos.y * radius, qz = pos.z * radius, qw = 0.0f;
    int   iter = 0;
    float d;
    do
    {
        /* quatSq :90-98 */
        float r0 = qx * qx - (qy * qy + qz * qz + qw * qw);
        float s  = qx * 2.0f; /* q.x * 2 (int 2 promoted) */
        float ry = qy * s, rz = qz * s, rw = qw * s;
        qx = r0 + -0.2f; qy = ry + 0.8f; qz = rz + 0.0f; qw = rw + 0.0f;
        d  = qx * qx + qy * qy + qz * qz + qw * qw; /* dot(float4) helper_math.h:1252-1255 */
    } while (d < 10.0f && iter++ < maxIter);
    return (float)((double)iter > (double)maxIter * 0.9); /* :114, int vs double compare */
}

void vpo_julia_voxelize(int n, uint8_t* grid)
{
    float fn = (float)n;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < n; k++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++)
            {
                f3 p = mk3(((float)i + 0.5f) / fn * 2.0f - 1.0f, ((float)j + 0.5f) / fn * 2.0f - 1.0f,
                           ((float)k + 0.5f) / fn * 2.0f - 1.0f);
                grid[(size_t)i + (size_t)n * ((size_t)j + (size_t)n * (size_t)k)] =
                    (uint8_t)(255.0f * julia_density(p));
            }
}

/* The FLAGGED SYNTHETIC cloud of the 512^3 workloads (include/volpath.h vp_cloud_voxelize; SURVEY.md section 8(d) asks for
 * such a stand-in where the WDAS data does not exist): restated operation by operation from its definition -- five octaves of
 * value noise on hashed lattices, thresholded, soft spherical edge -- in binary32 without contraction. */
static inline float cloud_lattice(int ix, int iy, int iz, uint32_t seed)
{
    uint32_t h = vpo_hash(((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ ((uint32_t)iz * 83492791u) ^ seed);
    return (float)(h & 0xffffffu) * (1.0f / 16777216.0f);
}
static float cloud_noise(float x, float y, float z, uint32_t seed)
{
    float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    int   ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float tx = x - fx, ty = y - fy, tz = z - fz;
    tx = (tx * tx) * (3.0f - 2.0f * tx);
    ty = (ty * ty) * (3.0f - 2.0f * ty);
    tz = (tz * tz) * (3.0f - 2.0f * tz);
    float c000 = cloud_lattice(ix, iy, iz, seed), c100 = cloud_lattice(ix + 1, iy, iz, seed);
    float c010 = cloud_lattice(ix, iy + 1, iz, seed), c110 = cloud_lattice(ix + 1, iy + 1, iz, seed);
    float c001 = cloud_lattice(ix, iy, iz + 1, seed), c101 = cloud_lattice(ix + 1, iy, iz + 1, seed);
    float c011 = cloud_lattice(ix, iy + 1, iz + 1, seed), c111 = cloud_lattice(ix + 1, iy + 1, iz + 1, seed);
    float x00 = c000 + (c100 - c000) * tx, x10 = c010 + (c110 - c010) * tx;
    float x01 = c001 + (c101 - c001) * tx, x11 = c011 + (c111 - c011) * tx;
    float y0 = x00 + (x10 - x00) * ty, y1 = x01 + (x11 - x01) * ty;
    return y0 + (y1 - y0) * tz;
}
void vpo_cloud_voxelize(int n, uint32_t seed, float* grid)
{
    float fn = (float)n;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < n; k++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++)
            {
                float px = ((float)i + 0.5f) / fn * 2.0f - 1.0f, py = ((float)j + 0.5f) / fn * 2.0f - 1.0f,
                      pz = ((float)k + 0.5f) / fn * 2.0f - 1.0f;
                float sum = 0.0f, amp = 0.5f, freq = 2.0f;
                for (int o = 0; o < 5; o++)
                {
                    sum  = sum + amp * cloud_noise(px * freq + 17.0f, py * freq + 17.0f, pz * freq + 17.0f,
                                                   seed + (uint32_t)o * 0x9E3779B9u);
                    amp  = amp * 0.5f;
                    freq = freq * 2.0f;
                }
                float v = (sum * (1.0f / 0.96875f) - 0.44f) * 3.0f;
                v = fminf(fmaxf(v, 0.0f), 1.0f);
                float r    = sqrtf(px * px + py * py + pz * pz);
                float edge = fminf(fmaxf((1.2f - r) * (1.0f / 0.4f), 0.0f), 1.0f);
                grid[(size_t)i + (size_t)n * ((size_t)j + (size_t)n * (size_t)k)] = v * edge;
            }
}

/* ------------------------------------------------------ local bounds (H1) -- */
/* host.cpp:1098-1101: diffusion_iters = ceil(search_radius / (2.0f / width)) */
int vpo_bound_radius(int nx, float search_radius)
{
    float cell_size = 2.0f / (float)nx;
    return (int)ceilf(search_radius / cell_size);
}

/* host.cpp:1088-1267: three separable sliding-window sweeps; the window at output o is
 * [o-r, o+r] clipped to the grid (derivation in DESIGN.md).  brick>1 (build's coarser table,
 * SURVEY S4): the voxel windows of one brick are merged. */
#define DEFINE_BOUNDS(NAME, T)                                                                         \
    void NAME(const T* grid, int nx, int ny, int nz, int radius, int brick, T* out)                    \
    {                                                                                                  \
        size_t n   = (size_t)nx * ny * nz;                                                             \
        T*     mxA = (T*)malloc(n * sizeof(T));                                                        \
        T*     mnA = (T*)malloc(n * sizeof(T));                                                        \
        T*     mxB = (T*)malloc(n * sizeof(T));                                                        \
        T*     mnB = (T*)malloc(n * sizeof(T));                                                        \
        memcpy(mxA, grid, n * sizeof(T));                                                              \
        memcpy(mnA, grid, n * sizeof(T));                                                              \
        int    dims[3]   = {nx, ny, nz};                                                               \
        size_t stride[3] = {1, (size_t)nx, (size_t)nx * ny};                                           \
        for (int axis = 0; axis < 3; axis++)                                                           \
        {                                                                                              \
            int    na = dims[axis];                                                                    \
            size_t sa = stride[axis];                                                                  \
            int    a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;                                           \
            _Pragma("omp parallel for schedule(static)") for (int v = 0; v < dims[a2]; v++)            \
                for (int u = 0; u < dims[a1]; u++)                                                     \
                {                                                                                      \
                    size_t base = (size_t)u * stride[a1] + (size_t)v * stride[a2];                     \
                    for (int o = 0; o < na; o++)                                                       \
                    {                                                                                  \
                        int lo = o - radius < 0 ? 0 : o - radius;                                      \
                        int hi = o + radius > na - 1 ? na - 1 : o + radius;                            \
                        T   mx = mxA[base + lo * sa], mn = mnA[base + lo * sa];                        \
                        for (int q = lo + 1; q <= hi; q++)                                             \
                        {                                                                              \
                            T a = mxA[base + q * sa], b = mnA[base + q * sa];                          \
                            if (a > mx) mx = a;                                                        \
                            if (b < mn) mn = b;                                                        \
                        }                                                                              \
                        mxB[base + o * sa] = mx;                                                       \
                        mnB[base + o * sa] = mn;                                                       \
                    }                                                                                  \
                }                                                                                      \
            T* t = mxA; mxA = mxB; mxB = t;                                                            \
            t = mnA; mnA = mnB; mnB = t;                                                               \
        }                                                                                              \
        int bnx = (nx + brick - 1) / brick, bny = (ny + brick - 1) / brick, bnz = (nz + brick - 1) / brick; \
        for (int bk = 0; bk < bnz; bk++)                                                               \
            for (int bj = 0; bj < bny; bj++)                                                           \
                for (int bi = 0; bi < bnx; bi++)                                                       \
                {                                                                                      \
                    T   mx = 0, mn = 0;                                                                \
                    int first = 1;                                                                     \
                    for (int k = bk * brick; k < (bk + 1) * brick && k < nz; k++)                      \
                        for (int j = bj * brick; j < (bj + 1) * brick && j < ny; j++)                  \
                            for (int i = bi * brick; i < (bi + 1) * brick && i < nx; i++)              \
                            {                                                                          \
                                size_t idx = (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k);    \
                                if (first || mxA[idx] > mx) mx = mxA[idx];                             \
                                if (first || mnA[idx] < mn) mn = mnA[idx];                             \
                                first = 0;                                                             \
                            }                                                                          \
                    size_t o   = ((size_t)bi + (size_t)bnx * ((size_t)bj + (size_t)bny * bk)) * 2;     \
                    out[o]     = mx; /* Data2.x = max, .y = min (host.cpp:1141-1144) */                \
                    out[o + 1] = mn;                                                                   \
                }                                                                                      \
        free(mxA); free(mnA); free(mxB); free(mnB);                                                    \
    }

DEFINE_BOUNDS(vpo_bounds_u8, uint8_t)
DEFINE_BOUNDS(vpo_bounds_f32, float)

/* ------------------------------------------------------- texture fetches -- */
/* texel-centre coordinate split with 8-bit weight: returns clamped i0,i1 and w in 0..256 */
static inline void tex_axis_linear(float pn, int n, int* i0, int* i1, int* w)
{
    float xb = fmaf(pn, (float)n, -0.5f); /* the unit's own scaling: one rounding */
    float fl = floorf(xb);
    float fr = xb - fl;
    int   i  = (int)fl;
    *w       = (int)fmaf(fr, 256.0f, 0.5f); /* fr >= 0: truncation == round-to-nearest of fr*256 */
    int a = i, b = i + 1;
    if (a < 0) a = 0;
    if (a > n - 1) a = n - 1;
    if (b < 0) b = 0;
    if (b > n - 1) b = n - 1;
    *i0 = a;
    *i1 = b;
}
static inline int tex_axis_point(float pn, int n)
{
    int i = (int)floorf(pn * (float)n);
    if (i < 0) i = 0;
    if (i > n - 1) i = n - 1;
    return i;
}
/* normalised texture coordinates of a world position: CudaTexture::sample_w, kernel.cu:173-178 */
static inline f3 to_local(const vpo_scene* S, f3 pos)
{
    f3 mn = mk3(S->box_min[0], S->box_min[1], S->box_min[2]);
    f3 mx = mk3(S->box_max[0], S->box_max[1], S->box_max[2]);
    f3 d  = sub3(mx, mn);
    f3 linv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); /* kernel.cu:313 */
    return mul3(sub3(pos, mn), linv);
}

#define VP_U8_TRI_SCALE 2.3374372e-10f /* fl(1 / (255 * 2^24)); maps the full-scale sum to exactly 1.0f */
#define VP_U8_SCALE 0.003921569f      /* fl(1 / 255) */

static inline float lerpf(float a, float b, float w) { return a * (1.0f - w) + b * w; }

/* tex3D<float> on the density (uchar normalised, or float), linear or point: kernel.cu:682-695,
 * host.cpp:39,1344 */
static float sample_volume(const vpo_scene* S, const uint8_t* g8, const float* gf, int linear, f3 pos)
{
    f3  p  = to_local(S, pos);
    int nx = S->nx, ny = S->ny, nz = S->nz;
    int i0, i1, j0, j1, k0, k1, wx, wy, wz;
    if (linear)
    {
        tex_axis_linear(p.x, nx, &i0, &i1, &wx);
        tex_axis_linear(p.y, ny, &j0, &j1, &wy);
        tex_axis_linear(p.z, nz, &k0, &k1, &wz);
    }
    else
    {
        i0 = i1 = tex_axis_point(p.x, nx);
        j0 = j1 = tex_axis_point(p.y, ny);
        k0 = k1 = tex_axis_point(p.z, nz);
        wx = wy = wz = 0;
    }
#define IDX(i, j, k) ((size_t)(i) + (size_t)nx * ((size_t)(j) + (size_t)ny * (size_t)(k)))
    if (g8)
    {
        uint32_t t000 = g8[IDX(i0, j0, k0)], t100 = g8[IDX(i1, j0, k0)];
        uint32_t t010 = g8[IDX(i0, j1, k0)], t110 = g8[IDX(i1, j1, k0)];
        uint32_t t001 = g8[IDX(i0, j0, k1)], t101 = g8[IDX(i1, j0, k1)];
        uint32_t t011 = g8[IDX(i0, j1, k1)], t111 = g8[IDX(i1, j1, k1)];
        uint32_t ux = (uint32_t)wx, uy = (uint32_t)wy, uz = (uint32_t)wz;
        uint32_t x00 = t000 * (256u - ux) + t100 * ux;
        uint32_t x10 = t010 * (256u - ux) + t110 * ux;
        uint32_t x01 = t001 * (256u - ux) + t101 * ux;
        uint32_t x11 = t011 * (256u - ux) + t111 * ux;
        uint32_t y0  = x00 * (256u - uy) + x10 * uy;
        uint32_t y1  = x01 * (256u - uy) + x11 * uy;
        uint32_t v   = y0 * (256u - uz) + y1 * uz; /* <= 255 * 2^24 < 2^32 */
        return (float)v * VP_U8_TRI_SCALE;
    }
    else
    {
        float fx = (float)wx * (1.0f / 256.0f), fy = (float)wy * (1.0f / 256.0f), fz = (float)wz * (1.0f / 256.0f);
        float x00 = lerpf(gf[IDX(i0, j0, k0)], gf[IDX(i1, j0, k0)], fx);
        float x10 = lerpf(gf[IDX(i0, j1, k0)], gf[IDX(i1, j1, k0)], fx);
        float x01 = lerpf(gf[IDX(i0, j0, k1)], gf[IDX(i1, j0, k1)], fx);
        float x11 = lerpf(gf[IDX(i0, j1, k1)], gf[IDX(i1, j1, k1)], fx);
        float y0  = lerpf(x00, x10, fy);
        float y1  = lerpf(x01, x11, fy);
        return lerpf(y0, y1, fz);
    }
#undef IDX
}

float vpo_sample_density(const vpo_scene* S, const float pos[3])
{
    return sample_volume(S, S->grid_u8, S->grid_f32, S->linear, mk3(pos[0], pos[1], pos[2]));
}
float vpo_sample_opacity(const vpo_scene* S, const float pos[3])
{
    return sample_volume(S, NULL, S->opacity, 1, mk3(pos[0], pos[1], pos[2])); /* kernel.cu:541-542 linear */
}

/* tex3D<float2> point-sampled (max,min): vol_bound_minmax kernel.cu:1610-1624, texture :392-395 */
static inline void sample_bound(const vpo_scene* S, f3 pos, float* bmax, float* bmin)
{
    f3  p = to_local(S, pos);
    int i = tex_axis_point(p.x, S->nx) / S->brick;
    int j = tex_axis_point(p.y, S->ny) / S->brick;
    int k = tex_axis_point(p.z, S->nz) / S->brick;
    size_t o = ((size_t)i + (size_t)S->bnx * ((size_t)j + (size_t)S->bny * (size_t)k)) * 2;
    if (S->bounds_u8)
    {
        *bmax = (float)S->bounds_u8[o] * VP_U8_SCALE;
        *bmin = (float)S->bounds_u8[o + 1] * VP_U8_SCALE;
    }
    else
    {
        *bmax = S->bounds_f32[o];
        *bmin = S->bounds_f32[o + 1];
    }
}
void vpo_sample_bound(const vpo_scene* S, const float pos[3], float out[2])
{
    sample_bound(S, mk3(pos[0], pos[1], pos[2]), &out[0], &out[1]);
}

/* vol_sigma_t kernel.cu:682-695 (USE_OPENVDB branch: the only compilable one) */
static inline float vol_sigma_t(const vpo_scene* S, f3 pos, float density, vpo_counters* C)
{
    C->density_lookups++;
    float t = sample_volume(S, S->grid_u8, S->grid_f32, S->linear, pos);
    t *= density;
    return t;
}

/* ---------------------------------------------------------- environment -- */
/* dir_to_theta / dir_to_uv kernel.cu:882-895; eval_envmap :956-973; tex2D point, normalised,
 * clamp (:1099-1101) */
static f3 eval_envmap(const vpo_scene* S, f3 dir, vpo_counters* C)
{
    C->env_lookups++;
    float phi   = vpo_acosf(dir.y);
    float theta = vpo_atanf(dir.z / dir.x) + VP_PI_2;
    if (dir.x < 0.0f) theta += VP_PI;
    float u = theta * VP_1_TWOPI;
    float v = phi * VP_1_PI;
    int   i = tex_axis_point(u, S->env_w);
    int   j = tex_axis_point(v, S->env_h);
    const float* t = S->env + 4 * ((size_t)i + (size_t)S->env_w * (size_t)j);
    return mk3(t[0], t[1], t[2]);
}
void vpo_eval_envmap(const vpo_scene* S, const float dir[3], float rgb[3])
{
    vpo_counters C;
    memset(&C, 0, sizeof C);
    f3 c = eval_envmap(S, mk3(dir[0], dir[1], dir[2]), &C);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

static inline float vpo_sinf(float a) { float s_, c_; vpo_sincosf(a, &s_, &c_); return s_; }
static inline float vpo_cosf(float a) { float s_, c_; vpo_sincosf(a, &s_, &c_); return c_; }

/* ---- active environment sampling (compiled out in the shipped reference: PASSIVE_ENVMAP 1, kernel.cu:21) ---- */
/* luminance kernel.cu:945-953: double arithmetic (the literals are double), result narrowed to float */
static inline float luminance3(float r, float g, float b)
{
    return (float)((double)r * 0.2126 + (double)g * 0.7152 + (double)b * 0.0722);
}

/* init_envmap kernel.cu:1144-1168 + build_cdf_1d/2d :1036-1070, configuration MULT_PDF 0, PRE_WARP 1 (:855-856):
 * row CDFs of luminance*sin(phi), CDF of the row sums, and HDRpdfnormAlt.  pdfX/pdfY are only read under MULT_PDF
 * and are not produced.  All sums are sequential float sums in texel order, as in the reference. */
void vpo_build_env_tables(const float* env, int width, int height, float* cdf_y, float* cdf_x, float* pdfnorm_alt)
{
    size_t total = (size_t)width * (size_t)height;
    float* lum   = (float*)malloc(total * sizeof(float));
    float* rows  = (float*)malloc((size_t)height * sizeof(float));
    for (int y = 0; y < height; y++)
    {
        float phi = VP_PI * ((float)y + 0.5f) / (float)height; /* :1158 */
        float sp  = vpo_sinf(phi);
        for (int x = 0; x < width; x++)
        {
            const float* t = env + 4 * ((size_t)x + (size_t)width * (size_t)y);
            lum[x + (size_t)y * width] = luminance3(t[0], t[1], t[2]) * sp;
        }
    }
    float lumsum = 0.0f;
    for (size_t i = 0; i < total; i++) lumsum += lum[i];
    *pdfnorm_alt = (float)width * (float)height * VP_1_TWO_PI_PI / lumsum; /* :1166 */
    for (int y = 0; y < height; y++)
    {
        const float* f   = lum + (size_t)y * width;
        float*       cdf = cdf_x + (size_t)y * width;
        float        sum = 0.0f;
        for (int i = 0; i < width; i++) sum += f[i];
        float norm = 1.0f / sum;
        float I    = 0.0f;
        for (int i = 0; i < width; i++) { I += f[i] * norm; cdf[i] = I; }
        cdf[width - 1] = 1.0f;
        rows[y]        = sum;
    }
    {
        float sum = 0.0f;
        for (int i = 0; i < height; i++) sum += rows[i];
        float norm = 1.0f / sum;
        float I    = 0.0f;
        for (int i = 0; i < height; i++) { I += rows[i] * norm; cdf_y[i] = I; }
        cdf_y[height - 1] = 1.0f;
    }
    free(lum); free(rows);
}

/* sample_y / sample_x kernel.cu:904-943: lower-bound binary search on a point-sampled CDF texture */
static int cdf_search(const float* cdf, int n, float r)
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

/* sample_envmap kernel.cu:979-1006: returns the pdf; (u,v) become the texel-centre coordinates */
static float sample_envmap(const vpo_scene* S, float* u, float* v, f3* c, vpo_counters* C)
{
    int iy = cdf_search(S->env_cdf_y, S->env_h, *v);
    int ix = cdf_search(S->env_cdf_x + (size_t)iy * S->env_w, S->env_w, *u);
    *u = ((float)ix + 0.5f) / (float)S->env_w;
    *v = ((float)iy + 0.5f) / (float)S->env_h;
    C->env_lookups++;
    int          i = tex_axis_point(*u, S->env_w);
    int          j = tex_axis_point(*v, S->env_h);
    const float* t = S->env + 4 * ((size_t)i + (size_t)S->env_w * (size_t)j);
    *c = mk3(t[0], t[1], t[2]);
    return luminance3(c->x, c->y, c->z) * S->env_pdfnorm_alt; /* consistent with the sine warp, :998 */
}

/* uv_to_dir kernel.cu:897-902 */
static f3 uv_to_dir(float u, float v)
{
    float theta = u * VP_TWO_PI;
    float phi   = v * VP_PI;
    return mk3(vpo_sinf(phi) * vpo_sinf(theta), vpo_cosf(phi), vpo_sinf(phi) * -vpo_cosf(theta));
}

/* background kernel.cu:1258-1267 (quirk Q11) */
static f3 background(const vpo_scene* S, f3 dir, int depth, vpo_counters* C)
{
    f3 sun = mk3(S->sun_dir[0], S->sun_dir[1], S->sun_dir[2]);
    if (depth == 0 && (dot3(dir, sun) > 94.0f / sqrtf(94.0f * 94.0f + 0.45f * 0.45f)))
        return mk3(S->sun_power_original[0], S->sun_power_original[1], S->sun_power_original[2]);
    return eval_envmap(S, dir, C);
}

/* set_sun kernel.cu:1269-1283: r = 0.45/94.0f in double -> float; p *= M_PI*(r*r) in float */
void vpo_set_sun(vpo_scene* S, const float dir[3], const float power[3])
{
    float r = (float)(0.45 / (double)94.0f);
    float f = VP_PI * (r * r);
    for (int i = 0; i < 3; i++)
    {
        S->sun_dir[i]            = dir[i];
        S->sun_power_original[i] = power[i];
        S->sun_power[i]          = power[i] * f;
    }
}

/* ----------------------------------------------------- phase function (A8) -- */
typedef struct { f3 n, t, b; } frame_t;

/* Frame kernel.cu:557-573.  fabs(n.x) > 0.1 compares against the DOUBLE 0.1; the smallest float
 * above it is 0.1f, hence >=. */
static frame_t make_frame(f3 normal)
{
    frame_t f;
    f.n  = normal;
    f3 a = (fabsf(normal.x) >= 0.1f) ? mk3(0, 1, 0) : mk3(1, 0, 0);
    f.t  = normalize3(cross3(a, f.n));
    f.b  = cross3(f.n, f.t);
    return f;
}
static inline f3 frame_to_world(const frame_t* f, f3 c)
{
    /* t * c.x + b * c.y + n * c.z */
    return add3(add3(muls(f->t, c.x), muls(f->b, c.y)), muls(f->n, c.z));
}

/* WHAT-IF switches for the radiometric pin (tests/test_oracle_cpu.py::test_julia_interior_prefers_the_reference_as_read): the
 * reference's own screenshot is compared with this restatement AND with variants in which one of its quirks is read differently.
 * 0 = the restatement (everything else in tests/ and bench.py runs with 0).  bit 0: no "Hyperion" reduction (quirk Q9: s = 0);
 * bit 1: Henyey-Greenstein sampling with cos(theta) clamped to [-1, 1] instead of [0, 1] (quirk Q1). */
static int g_what_if = 0;
void vpo_debug_set_what_if(int mask) { g_what_if = mask; }
/* HGPhaseFunction::sample kernel.cu:580-598 (quirk Q1: cos_theta clamped to [0,1]) */
static f3 hg_sample_local(float g, float rnd0, float rnd1)
{
    float cos_theta;
    if (fabsf(g) > 1e-6f)
    {
        float s   = 2.0f * rnd0 - 1.0f;
        float f   = (1.0f - g * g) / (1.0f + g * s);
        cos_theta = (0.5f / g) * (1.0f + g * g - f * f);
        cos_theta = fmaxf((g_what_if & 2) ? -1.0f : 0.0f, fminf(1.0f, cos_theta));
    }
    else
        cos_theta = 2.0f * rnd0 - 1.0f;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi       = (2.0f * VP_PI) * rnd1;
    float sp, cp;
    vpo_sincosf(phi, &sp, &cp);
    return mk3(cp * sin_theta, sp * sin_theta, cos_theta);
}
/* HGPhaseFunction::evaluate kernel.cu:600-603 */
float vpo_hg_eval(float g, float cos_theta)
{
    return (1.0f - g * g) / ((4.0f * VP_PI) * vpo_pow15f(1.0f + g * g - (2.0f * g) * cos_theta));
}
void vpo_hg_sample(float g, const float n[3], float u0, float u1, float out[3])
{
    frame_t f = make_frame(mk3(n[0], n[1], n[2]));
    f3      d = normalize3(frame_to_world(&f, hg_sample_local(g, u0, u1))); /* kernel.cu:2301 */
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

/* ------------------------------------------------------ box intersections -- */
/* intersectBox kernel.cu:654-680 (quirk Q13) */
static int intersect_box(f3 o, f3 d, f3 bmin, f3 bmax, float* tnear, float* tfar)
{
    f3 invR = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    f3 tbot = mul3(invR, sub3(bmin, o));
    f3 ttop = mul3(invR, sub3(bmax, o));
    f3 tmin = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    f3 tmax = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    float largest_tmin  = max_of3(tmin);
    float smallest_tmax = min_of3(tmax);
    *tnear = largest_tmin;
    *tfar  = smallest_tmax;
    return smallest_tmax > largest_tmin && smallest_tmax >= 1e-3f;
}
int vpo_intersect_box(const float o[3], const float d[3], const float bmin[3], const float bmax[3], float* tn,
                      float* tf)
{
    return intersect_box(mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), mk3(bmin[0], bmin[1], bmin[2]),
                         mk3(bmax[0], bmax[1], bmax[2]), tn, tf);
}

#define VP_SEARCH_RADIUS 0.05f /* kernel.cu:151 */

/* intersectSuperVolume kernel.cu:1626-1661 (quirks Q6, Q10) */
static int intersect_super_volume(const vpo_scene* S, f3 o, f3 d, f3 bmin, f3 bmax, float* tnear, float* tfar,
                                  float* dmin, float* dmax, vpo_counters* C)
{
    float tn, tf;
    int   hit = intersect_box(o, d, bmin, bmax, &tn, &tf);
    *tnear    = fmaxf(tn, 0.0f);
    *tfar     = fminf(tf, VP_SEARCH_RADIUS);
    float bx, by;
    C->bound_lookups++;
    sample_bound(S, add3(o, muls(d, *tnear)), &bx, &by);
    *dmin = by;
    *dmax = fmaxf(0.0001f, bx);
    return hit;
}

/* ------------------------------------------------ shadow transmittance (A7) -- */
/* Tr_spectral kernel.cu:754-808: shared free-flight sample, per-channel termination flags */
static f3 tr_spectral(const vpo_scene* S, f3 bmin, f3 bmax, f3 start, f3 end, float inv_sigma, float density,
                      f3 sigma_t_spectral, rng_t* rng, uint32_t shadow_id, vpo_counters* C)
{
    f3    o = start;
    f3    d = normalize3(sub3(end, start));
    float t_near, t_far;
    int   shade_vol = intersect_box(o, d, bmin, bmax, &t_near, &t_far);
    if (!shade_vol) return mk3(1.0f, 1.0f, 1.0f);
    if (t_near < 0.0f) t_near = 0.0f;
    float max_t = fminf(t_far, length3(sub3(start, end)));
    float dist  = t_near;
    int   xterm = 0, yterm = 0, zterm = 0;
    rng_enter_shadow(rng, shadow_id);
    for (;;)
    {
        dist += -vpo_logf(rng_next_a(rng)) * inv_sigma;
        if (dist >= max_t || (xterm && yterm && zterm)) break;
        f3    pos = add3(o, muls(d, dist));
        float e   = rng_next_b(rng);
        float den = vol_sigma_t(S, pos, density, C);
        if (!xterm && e < sigma_t_spectral.x * den * inv_sigma) xterm = 1;
        if (!yterm && e < sigma_t_spectral.y * den * inv_sigma) yterm = 1;
        if (!zterm && e < sigma_t_spectral.z * den * inv_sigma) zterm = 1;
    }
    rng_leave_shadow(rng);
    return mk3((float)(1 - xterm), (float)(1 - yterm), (float)(1 - zterm));
}

/* test hook: how often the zero-pdf `continue` of the MIS block was taken (it needs a draw of exactly 0) */
static uint64_t g_mis_zero_pdf = 0;
uint64_t vpo_debug_mis_zero_pdf(void) { return __atomic_load_n(&g_mis_zero_pdf, __ATOMIC_RELAXED); }

/* ------------------------------------ one-sample MIS of the environment -- */
/* kernel.cu:2220-2297 (A1), :1494-1560 (A2), :1855-1932 (A3): identical in the three kernels.  With probability
 * 1/2 a direction is drawn from the phase function, else from the environment's luminance CDF; balance heuristic.
 * Returns 1 for the reference's `continue` on a zero-pdf environment sample: the path then goes on WITHOUT the
 * scattered-direction update, i.e. from the old ray origin in the old direction (with the scatter already counted). */
static int mis_envmap(const vpo_scene* S, rng_t* rng, vpo_counters* C, const frame_t* frame, float g, f3 pos,
                      f3 throughput, float inv_sigma, float density_prime, f3 sigma_t_spectral, f3 boxMin,
                      f3 boxMax, f3* radiance, uint32_t depth)
{
    const float P_phase  = 0.5f;
    const float P_envmap = 1.0f - P_phase;
    if (rng_next_a(rng) < P_phase)
    {
        float u = rng_next_a(rng);
        float v = rng_next_b(rng);
        f3    brdf_dir = frame_to_world(frame, hg_sample_local(g, u, v));
        f3    envc     = eval_envmap(S, brdf_dir, C);
        float pdf_brdf = vpo_hg_eval(g, dot3(frame->n, brdf_dir));
        float pdf_env_virtual = luminance3(envc.x, envc.y, envc.z) * S->env_pdfnorm_alt; /* pdf_envmap :1009-1034 */
        float a_ = pdf_brdf * P_phase, b_ = pdf_env_virtual * P_envmap;
        float weight = a_ / (a_ + b_) / P_phase;
        f3 a = tr_spectral(S, boxMin, boxMax, pos, muls(brdf_dir, 1e10f), inv_sigma, density_prime, sigma_t_spectral,
                           rng, 2u * depth + 1u, C);
        *radiance = add3(*radiance, mul3(envc, mul3(muls(throughput, weight), a)));
    }
    else
    {
        float u = rng_next_a(rng);
        float v = rng_next_b(rng);
        f3    envc;
        float pdf_env = sample_envmap(S, &u, &v, &envc, C);
        if (pdf_env <= 0.0f)
        {
            __atomic_fetch_add(&g_mis_zero_pdf, 1, __ATOMIC_RELAXED);
            return 1;
        }
        f3    envmap_dir       = uv_to_dir(u, v);
        float pdf_brdf_virtual = vpo_hg_eval(g, dot3(frame->n, envmap_dir));
        float a_ = pdf_env * P_envmap, b_ = pdf_brdf_virtual * P_phase;
        float weight = a_ / (a_ + b_) / P_envmap;
        f3 a = tr_spectral(S, boxMin, boxMax, pos, muls(envmap_dir, 1e10f), inv_sigma, density_prime, sigma_t_spectral,
                           rng, 2u * depth + 1u, C);
        float ph = vpo_hg_eval(g, dot3(frame->n, envmap_dir));
        f3    t  = muls(divs(muls(throughput, ph), pdf_env), weight);
        *radiance = add3(*radiance, mul3(envc, mul3(t, a)));
    }
    return 0;
}

/* ----------------------------------------------------------- camera (Q3) -- */
/* kernel.cu:1977-1987 / :1304-1314 */
static void camera_ray(const vpo_scene* S, const vpo_param* P, uint32_t x, uint32_t y, f3* o, f3* d)
{
    float u = ((float)x * 2.0f - (float)P->width) / (float)P->width;
    float v = ((float)y * 2.0f - (float)P->height) / (float)P->width;
    float fovx = 54.43f;
    float cz   = (float)(-1.0f / tan((double)fovx * 0.00872664626));
    const float* m = S->inv_view;
    *o = mk3(m[3], m[7], m[11]); /* mul(M, (0,0,0,1)) :641-649 */
    f3 dv = mk3(u, v, cz);
    f3 r  = mk3(dot3(dv, mk3(m[0], m[1], m[2])), dot3(dv, mk3(m[4], m[5], m[6])), dot3(dv, mk3(m[8], m[9], m[10])));
    *d = normalize3(r);
}

/* H4: lookAt(pos, pos+fwd*4, up) -> inverse -> transpose -> first 12 floats (host.cpp:108-115,617-623);
 * for the default camera this is the explicit matrix of SURVEY 8(a) H4. */
void vpo_default_camera(float m[12])
{
    const float v[12] = {0.0f, 0.207912f, 0.978148f, 3.922986f, 0.0f, 0.978148f, -0.207912f, -0.782739f,
                         -1.0f, 0.0f, 0.0f, 0.03f};
    memcpy(m, v, sizeof v);
}

/* hyperion trick kernel.cu:2039-2041 / :1358-1359 (quirk Q9) */
static inline float hyperion_s(int n_minus)
{
    if (g_what_if & 1) return 0.0f;
    return fmaxf(0.0f, fminf(1.0f, (float)n_minus * 0.066666666666666666667f));
}

/* ---------------------------------------- A1: __d_render_bounded_decomp -- */
/* kernel.cu:1958-2318 */
static void sample_decomp(const vpo_scene* S, const vpo_param* P, uint32_t x, uint32_t y, int spp, float out[4],
                          vpo_counters* C)
{
    const float density    = P->density;
    const float brightness = P->brightness;
    f3 boxMin = mk3(S->box_min[0], S->box_min[1], S->box_min[2]);
    f3 boxMax = mk3(S->box_max[0], S->box_max[1], S->box_max[2]);
    f3 sun_dir   = mk3(S->sun_dir[0], S->sun_dir[1], S->sun_dir[2]);
    f3 sun_power = mk3(S->sun_power[0], S->sun_power[1], S->sun_power[2]);

    rng_t rng;
    rng_init(&rng, S, x, y, (uint32_t)spp, &C->rng_draws);

    f3 cr_o, cr_d;
    camera_ray(S, P, x, y, &cr_o, &cr_d);

    f3 radiance   = mk3(0, 0, 0);
    f3 throughput = mk3(1, 1, 1);

    f3    sigma_t_spectral = mk3(P->sigma_t[0], P->sigma_t[1], P->sigma_t[2]);
    f3    sigma_s_spectral = mul3(sigma_t_spectral, mk3(P->albedo[0], P->albedo[1], P->albedo[2]));
    float max_sigma_t      = max_of3(sigma_t_spectral);
    float min_sigma_t      = min_of3(sigma_t_spectral);

    float sigma_c_prime = 0, distc = 0, sigma_r_prime = 0, inv_sigma = 0, inv_sigma_t = 0;
    f3    sigma_c_spectral = mk3(0, 0, 0);

    int num_scatters = 0;
    while (num_scatters < 800)
    {
        float t_near, t_far, d_min, d_max;
        int   hit = intersect_super_volume(S, cr_o, cr_d, boxMin, boxMax, &t_near, &t_far, &d_min, &d_max, C);
        int   use_decomposition = d_min > 0.0f;
        if (!hit)
        {
            /* PASSIVE_ENVMAP: every escaping path sees the environment; otherwise only unscattered ones (:2026-2030) */
            if (!S->env_mis || 0 == num_scatters)
                radiance = add3(radiance, mul3(background(S, cr_d, num_scatters, C), throughput));
            break;
        }
        f3    pos  = add3(cr_o, muls(cr_d, t_near));
        float dist = t_near;

        float s = hyperion_s(num_scatters - 5);
        float g = (1.0f - s) * P->g;
        float reduction_factor = (1.0f - s) + s * (1.0f - P->g);
        float density_prime = reduction_factor * density;
        float sigma_t_prime = max_sigma_t * density_prime * d_max;

        if (use_decomposition)
        {
            sigma_c_prime    = min_sigma_t * density_prime * d_min;
            distc            = dist - vpo_logf(rng_next_a(&rng)) / fmaxf(sigma_c_prime, 1e-20f);
            sigma_r_prime    = fmaxf(sigma_t_prime - sigma_c_prime, 1e-20f);
            sigma_c_spectral = mk3(sigma_c_prime, sigma_c_prime, sigma_c_prime);
        }
        else
        {
            distc            = 1e20f;
            sigma_c_spectral = mk3(0, 0, 0);
        }
        const float phase_g = g;

        inv_sigma_t = 1.0f / sigma_t_prime;
        if (use_decomposition) inv_sigma = 1.0f / sigma_r_prime;
        else inv_sigma = inv_sigma_t;

        int through;
        for (;;)
        {
            dist += -vpo_logf(rng_next_a(&rng)) * inv_sigma;
            if (dist >= distc || dist >= t_far)
            {
                pos = add3(cr_o, muls(cr_d, distc));
                break;
            }
            else
                pos = add3(cr_o, muls(cr_d, dist));

            float den = vol_sigma_t(S, pos, density_prime, C);
            f3 sigma_t_den    = sub3(muls(sigma_t_spectral, den), sigma_c_spectral);
            f3 sigma_s_den    = sub3(muls(sigma_s_spectral, den), sigma_c_spectral);
            /* what-if 8 (test-only): the control component carries the medium's albedo -- sigma_s - albedo * sigma_c -- instead of being
             * carved whole from the scattering coefficient: what quirk Q7 is NOT (the same thing when the albedo is 1) */
            if (g_what_if & 8)
                sigma_s_den = sub3(muls(sigma_s_spectral, den), mul3(sigma_c_spectral, mk3(P->albedo[0], P->albedo[1], P->albedo[2])));
            f3 sigma_null_den = sub3(mk3(sigma_t_prime, sigma_t_prime, sigma_t_prime), sigma_t_den);

            float Ps = fabsf(sigma_t_den.x * throughput.x) + fabsf(sigma_t_den.y * throughput.y) +
                       fabsf(sigma_t_den.z * throughput.z);
            float Pn = fabsf(sigma_null_den.x * throughput.x) + fabsf(sigma_null_den.y * throughput.y) +
                       fabsf(sigma_null_den.z * throughput.z);
            float c = (Ps + Pn);
            float e = rng_next_b(&rng) * c;
            /* what-if 16 (test-only): the collision weights with the majorant the flight was SAMPLED with (the residual one while a
             * control component is in use) instead of the total one: what quirk Q8 is NOT (the same thing wherever d_min = 0) */
            const float inv_w = (g_what_if & 16) ? inv_sigma : inv_sigma_t;
            if (e < Ps)
            {
                throughput = mul3(throughput, muls(sigma_s_den, inv_w * c / (Ps)));
                break;
            }
            else
                throughput = mul3(throughput, muls(sigma_null_den, inv_w * c / Pn));
        }

        through = fminf(distc, dist) >= t_far;
        num_scatters += (!through);
        if (!through && use_decomposition) C->control_segments++;   /* collisions (real or control) in a segment with d_min > 0 */
        if (through)
        {
            cr_o = add3(cr_o, muls(cr_d, t_far));
            continue;
        }
        C->scatters++;

        frame_t frame = make_frame(cr_d);
        {
            float s2 = hyperion_s(num_scatters - 5);
            float reduction2 = (1.0f - s2) + s2 * (1.0f - P->g);
            float density_prime2 = reduction2 * density;
            float sigma_t_prime2 = max_sigma_t * density_prime2 * d_max;
            /* what-if 4 (test-only, tests/test_oracle_cpu.py): the shadow ray on the GLOBAL majorant (volume maximum 1) instead of the
             * local segment's d_max -- what quirk Q4 is NOT */
            if (g_what_if & 4) sigma_t_prime2 = max_sigma_t * density_prime2 * 1.0f;
            float inv_sigma2     = 1.0f / sigma_t_prime2;
            float ph = vpo_hg_eval(phase_g, dot3(frame.n, sun_dir));
            f3    a;
            if (spp > 10 && num_scatters > 20)
            {
                /* quirk Q5, kernel.cu:2183-2189 */
                C->opacity_lookups++;
                float op = sample_volume(S, NULL, S->opacity, 1, pos);
                f3 tau = muls(muls(mk3(-sigma_t_spectral.x, -sigma_t_spectral.y, -sigma_t_spectral.z), density_prime2), op);
                a = mk3(vpo_expf(tau.x), vpo_expf(tau.y), vpo_expf(tau.z));
            }
            else
                a = tr_spectral(S, boxMin, boxMax, pos, muls(sun_dir, 1e10f), inv_sigma2, density_prime2,
                                sigma_t_spectral, &rng, 2u * (uint32_t)num_scatters, C);
            /* sun_light_power * (throughput * phase * a) */
            radiance = add3(radiance, mul3(sun_power, mul3(muls(throughput, ph), a)));
            if (S->env_mis && mis_envmap(S, &rng, C, &frame, phase_g, pos, throughput, inv_sigma2, density_prime2,
                                         sigma_t_spectral, boxMin, boxMax, &radiance, (uint32_t)num_scatters))
                continue; /* :2266 */
        }

        float r0 = rng_next_a(&rng);
        float r1 = rng_next_b(&rng);
        f3 new_dir = normalize3(frame_to_world(&frame, hg_sample_local(phase_g, r0, r1)));
        cr_o = pos;
        cr_d = new_dir;
    }

    radiance = muls(radiance, brightness);
    out[0] = fmaxf(radiance.x, 0.0f);
    out[1] = fmaxf(radiance.y, 0.0f);
    out[2] = fmaxf(radiance.z, 0.0f);
    out[3] = (float)num_scatters;
}

/* ---------------------------------------------- A3: __d_render_bounded -- */
/* kernel.cu:1667-1952: the local-majorant tracker without the control component.  Dead in the reference
 * (its launch is commented out at :2368) but part of the same TU.  Against A1 it differs in four places:
 * the loop is bounded by the segment index i < max_depth (:1716), a transmitted segment counts as an
 * iteration (`continue`, :1812), the heat channel is i * 0.001 (:1942), and there is no opacity-volume branch. */
static void sample_bounded(const vpo_scene* S, const vpo_param* P, uint32_t x, uint32_t y, int spp, float out[4],
                           vpo_counters* C)
{
    const float density    = P->density;
    const float brightness = P->brightness;
    f3 boxMin = mk3(S->box_min[0], S->box_min[1], S->box_min[2]);
    f3 boxMax = mk3(S->box_max[0], S->box_max[1], S->box_max[2]);
    f3 sun_dir   = mk3(S->sun_dir[0], S->sun_dir[1], S->sun_dir[2]);
    f3 sun_power = mk3(S->sun_power[0], S->sun_power[1], S->sun_power[2]);

    rng_t rng;
    rng_init(&rng, S, x, y, (uint32_t)spp, &C->rng_draws);
    f3 cr_o, cr_d;
    camera_ray(S, P, x, y, &cr_o, &cr_d);

    f3 radiance   = mk3(0, 0, 0);
    f3 throughput = mk3(1, 1, 1);
    f3    sigma_t_spectral = mk3(P->sigma_t[0], P->sigma_t[1], P->sigma_t[2]);
    f3    sigma_s_spectral = mul3(sigma_t_spectral, mk3(P->albedo[0], P->albedo[1], P->albedo[2]));
    float max_sigma_t      = max_of3(sigma_t_spectral);

    int num_scatters = 0;
    int i;
    for (i = 0; i < 800; i++) /* max_depth :34 */
    {
        float t_near, t_far, d_min, d_max;
        int   hit = intersect_super_volume(S, cr_o, cr_d, boxMin, boxMax, &t_near, &t_far, &d_min, &d_max, C);
        (void)d_min;
        if (!hit)
        {
            if (!S->env_mis || 0 == num_scatters) /* :1724-1731 */
                radiance = add3(radiance, mul3(background(S, cr_d, num_scatters, C), throughput));
            break;
        }
        f3    pos;
        float dist = t_near;

        /* :1737-1751 */
        float s = hyperion_s(num_scatters - 5);
        float g = (1.0f - s) * P->g;
        float reduction_factor = (1.0f - s) + s * (1.0f - P->g);
        float density_prime = reduction_factor * density;
        float sigma_t_prime = max_sigma_t * density_prime * d_max;
        float inv_sigma     = 1.0f / sigma_t_prime;

        int through = 0;
        for (;;)
        {
            dist += -vpo_logf(rng_next_a(&rng)) * inv_sigma; /* :1757 */
            pos = add3(cr_o, muls(cr_d, dist));
            if (dist >= t_far)
            {
                through = 1;
                break;
            }
            /* :1766-1795 */
            float den            = vol_sigma_t(S, pos, density_prime, C);
            f3    sigma_t_den    = muls(sigma_t_spectral, den);
            f3    sigma_s_den    = muls(sigma_s_spectral, den);
            f3    sigma_null_den = sub3(mk3(sigma_t_prime, sigma_t_prime, sigma_t_prime), sigma_t_den);
            float Ps = fabsf(sigma_t_den.x * throughput.x) + fabsf(sigma_t_den.y * throughput.y) +
                       fabsf(sigma_t_den.z * throughput.z);
            float Pn = fabsf(sigma_null_den.x * throughput.x) + fabsf(sigma_null_den.y * throughput.y) +
                       fabsf(sigma_null_den.z * throughput.z);
            float c = (Ps + Pn);
            float e = rng_next_b(&rng) * c;
            if (e < Ps)
            {
                throughput = mul3(throughput, muls(sigma_s_den, inv_sigma * c / (Ps)));
                ++num_scatters;
                break;
            }
            else
                throughput = mul3(throughput, muls(sigma_null_den, inv_sigma * c / Pn));
        }
        if (through)
        {
            cr_o = add3(cr_o, muls(cr_d, t_far)); /* :1809-1813 */
            continue;
        }
        C->scatters++;

        frame_t frame = make_frame(cr_d);
        {
            /* :1822-1851, num_scatters already incremented */
            float s2 = hyperion_s(num_scatters - 5);
            float reduction2 = (1.0f - s2) + s2 * (1.0f - P->g);
            float density_prime2 = reduction2 * density;
            float sigma_t_prime2 = max_sigma_t * density_prime2 * d_max;
            float inv_sigma2     = 1.0f / sigma_t_prime2;
            float ph = vpo_hg_eval(g, dot3(frame.n, sun_dir));
            f3    a  = tr_spectral(S, boxMin, boxMax, pos, muls(sun_dir, 1e10f), inv_sigma2, density_prime2,
                                   sigma_t_spectral, &rng, 2u * (uint32_t)num_scatters, C);
            radiance = add3(radiance, mul3(sun_power, mul3(muls(throughput, ph), a)));
            if (S->env_mis && mis_envmap(S, &rng, C, &frame, g, pos, throughput, inv_sigma2, density_prime2,
                                         sigma_t_spectral, boxMin, boxMax, &radiance, (uint32_t)num_scatters))
                continue; /* :1900; the loop index still advances */
        }
        float r0 = rng_next_a(&rng);
        float r1 = rng_next_b(&rng);
        f3 new_dir = normalize3(frame_to_world(&frame, hg_sample_local(g, r0, r1))); /* :1935-1937 */
        cr_o = pos;
        cr_d = new_dir;
    }

    radiance = muls(radiance, brightness);
    out[0] = fmaxf(radiance.x, 0.0f);
    out[1] = fmaxf(radiance.y, 0.0f);
    out[2] = fmaxf(radiance.z, 0.0f);
    out[3] = (float)((double)i * 0.001); /* :1942 */
}

/* ------------------------------------------------------- A2: __d_render -- */
/* kernel.cu:1285-1591: global majorant (volume max assumed 1), no restarts */
static void sample_global(const vpo_scene* S, const vpo_param* P, uint32_t x, uint32_t y, int spp, float out[4],
                          vpo_counters* C)
{
    const float density    = P->density;
    const float brightness = P->brightness;
    f3 boxMin = mk3(S->box_min[0], S->box_min[1], S->box_min[2]);
    f3 boxMax = mk3(S->box_max[0], S->box_max[1], S->box_max[2]);
    f3 sun_dir   = mk3(S->sun_dir[0], S->sun_dir[1], S->sun_dir[2]);
    f3 sun_power = mk3(S->sun_power[0], S->sun_power[1], S->sun_power[2]);

    rng_t rng;
    rng_init(&rng, S, x, y, (uint32_t)spp, &C->rng_draws);
    f3 cr_o, cr_d;
    camera_ray(S, P, x, y, &cr_o, &cr_d);

    f3 radiance   = mk3(0, 0, 0);
    f3 throughput = mk3(1, 1, 1);
    f3    sigma_t_spectral = mk3(P->sigma_t[0], P->sigma_t[1], P->sigma_t[2]);
    f3    sigma_s_spectral = mul3(sigma_t_spectral, mk3(P->albedo[0], P->albedo[1], P->albedo[2]));
    float max_sigma_t      = max_of3(sigma_t_spectral);

    int i;
    for (i = 0; i < 800; i++)
    {
        float t_near, t_far;
        int   hit = intersect_box(cr_o, cr_d, boxMin, boxMax, &t_near, &t_far);
        if (!hit)
        {
            if (!S->env_mis || 0 == i) /* :1340-1344 */
                radiance = add3(radiance, mul3(background(S, cr_d, i, C), throughput));
            break;
        }
        if (t_near < 0.0f) t_near = 0.0f;
        f3    pos  = add3(cr_o, muls(cr_d, t_near));
        float dist = t_near;

        float s = hyperion_s(i - 5);
        float g = (1.0f - s) * P->g;
        float density_prime = (1.0f - s) * density + s * density * (1.0f - P->g);
        float sigma_t_prime = max_sigma_t * density_prime;
        float inv_sigma     = 1.0f / sigma_t_prime;

        int through = 0;
        for (;;)
        {
            dist += -vpo_logf(rng_next_a(&rng)) * inv_sigma;
            pos = add3(cr_o, muls(cr_d, dist));
            if (dist >= t_far) { through = 1; break; }

            float den = vol_sigma_t(S, pos, density_prime, C);
            f3 sigma_t_den    = muls(sigma_t_spectral, den);
            f3 sigma_s_den    = muls(sigma_s_spectral, den);
            f3 sigma_null_den = sub3(mk3(sigma_t_prime, sigma_t_prime, sigma_t_prime), sigma_t_den);
            float Pa = 0.0f;
            float Ps = fabsf(sigma_t_den.x * throughput.x) + fabsf(sigma_t_den.y * throughput.y) +
                       fabsf(sigma_t_den.z * throughput.z);
            float Pn = fabsf(sigma_null_den.x * throughput.x) + fabsf(sigma_null_den.y * throughput.y) +
                       fabsf(sigma_null_den.z * throughput.z);
            float c = (Pa + Ps + Pn);
            float e = rng_next_b(&rng) * c;
            if (e < Pa + Ps)
            {
                throughput = mul3(throughput, muls(sigma_s_den, inv_sigma * c / (Pa + Ps)));
                break;
            }
            else
                throughput = mul3(throughput, muls(sigma_null_den, inv_sigma * c / Pn));
        }
        if (through)
        {
            if (!S->env_mis || 0 == i) /* :1446-1450 */
                radiance = add3(radiance, mul3(background(S, cr_d, i, C), throughput));
            break;
        }
        C->scatters++;

        frame_t frame = make_frame(cr_d);
        {
            float s2 = hyperion_s(i - 4);
            float density_prime2 = (1.0f - s2) * density + s2 * density * (1.0f - P->g);
            float sigma_t_prime2 = max_sigma_t * density_prime2;
            float inv_sigma2     = 1.0f / sigma_t_prime2;
            f3 a = tr_spectral(S, boxMin, boxMax, pos, muls(sun_dir, 1e10f), inv_sigma2, density_prime2,
                               sigma_t_spectral, &rng, 2u * (uint32_t)i, C);
            float ph = vpo_hg_eval(g, dot3(frame.n, sun_dir));
            radiance = add3(radiance, mul3(sun_power, mul3(muls(throughput, ph), a)));
            if (S->env_mis && mis_envmap(S, &rng, C, &frame, g, pos, throughput, inv_sigma2, density_prime2,
                                         sigma_t_spectral, boxMin, boxMax, &radiance, (uint32_t)i))
                continue; /* :1539; the depth index still advances */
        }
        float r0 = rng_next_a(&rng);
        float r1 = rng_next_b(&rng);
        f3 new_dir = normalize3(frame_to_world(&frame, hg_sample_local(g, r0, r1)));
        cr_o = pos;
        cr_d = new_dir;
    }
    radiance = muls(radiance, brightness);
    out[0] = fmaxf(radiance.x, 0.0f);
    out[1] = fmaxf(radiance.y, 0.0f);
    out[2] = fmaxf(radiance.z, 0.0f);
    out[3] = (float)((double)i * 0.001); /* heat = i * 0.001, kernel.cu:1582 */
}

/* -------------------------- scalar tracking: SPECTRAL_TRACKING 0 / MULTI_CHANNEL 1 -- */
/* Tr kernel.cu:712-751: the scalar shadow ray; stops AT its first collision (no further draw) */
static float tr_scalar(const vpo_scene* S, f3 bmin, f3 bmax, f3 start, f3 end, float inv_sigma, float density,
                       rng_t* rng, uint32_t shadow_id, vpo_counters* C)
{
    f3    o = start;
    f3    d = normalize3(sub3(end, start));
    float t_near, t_far;
    if (!intersect_box(o, d, bmin, bmax, &t_near, &t_far)) return 1.0f;
    if (t_near < 0.0f) t_near = 0.0f;
    float max_t = fminf(t_far, length3(sub3(start, end)));
    float dist  = t_near;
    rng_enter_shadow(rng, shadow_id);
    for (;;)
    {
        dist += -vpo_logf(rng_next_a(rng)) * inv_sigma;
        if (dist >= max_t) break;
        f3 pos = add3(o, muls(d, dist));
        if (rng_next_b(rng) < vol_sigma_t(S, pos, density, C) * inv_sigma) break;
    }
    rng_leave_shadow(rng);
    return (float)(dist >= max_t);
}

/* The three kernels compiled with SPECTRAL_TRACKING 0 (track_mode 1) or MULTI_CHANNEL 1 (track_mode 2), both compiled
 * out in the shipped reference (kernel.cu:15-34).  One extinction coefficient -- density, or density * sigma_t[channel]
 * with the channel drawn per sample (:1993-1994) -- classic delta tracking against the Hyperion-reduced coefficient WITHOUT
 * the local bound (the bound is still fetched by intersectSuperVolume), throughput *= albedo per collision, scalar
 * shadow rays.  A1 :2063/:2101-2143/:2157-2215, A3 :1745/:1797-1817, A2 :1363/:1435-1455; PASSIVE_ENVMAP 1 only. */
static void sample_scalar(const vpo_scene* S, const vpo_param* P, uint32_t x, uint32_t y, int spp, float out[4],
                          vpo_counters* C)
{
    const int   est        = S->estimator;
    const float density    = P->density;
    const float brightness = P->brightness;
    f3 boxMin = mk3(S->box_min[0], S->box_min[1], S->box_min[2]);
    f3 boxMax = mk3(S->box_max[0], S->box_max[1], S->box_max[2]);
    f3 sun_dir   = mk3(S->sun_dir[0], S->sun_dir[1], S->sun_dir[2]);
    f3 sun_power = mk3(S->sun_power[0], S->sun_power[1], S->sun_power[2]);
    f3 albedo    = mk3(P->albedo[0], P->albedo[1], P->albedo[2]);

    rng_t rng;
    rng_init(&rng, S, x, y, (uint32_t)spp, &C->rng_draws);
    f3 cr_o, cr_d;
    camera_ray(S, P, x, y, &cr_o, &cr_d);
    f3 radiance   = mk3(0, 0, 0);
    f3 throughput = mk3(1, 1, 1);

    int   channel = 0;
    float sigma_t = density;
    if (S->track_mode == 2)
    {
        channel = (int)fminf((1.0f - rng_next_a(&rng)) * 3.0f, 2.9999998f); /* :1993 */
        sigma_t = density * P->sigma_t[channel];
    }

    int num_scatters = 0; /* A1, A3 */
    int i = 0;            /* loop index of A2, A3 */
    for (;;)
    {
        if (est == VPO_EST_DECOMP ? !(num_scatters < 800) : !(i < 800)) break;
        const int depth = est == VPO_EST_GLOBAL ? i : num_scatters; /* what hyperion and background see */
        float t_near, t_far, d_min, d_max;
        int   hit;
        if (est == VPO_EST_GLOBAL)
        {
            hit = intersect_box(cr_o, cr_d, boxMin, boxMax, &t_near, &t_far);
            if (hit && t_near < 0.0f) t_near = 0.0f;
        }
        else
            hit = intersect_super_volume(S, cr_o, cr_d, boxMin, boxMax, &t_near, &t_far, &d_min, &d_max, C);
        if (!hit)
        {
            radiance = add3(radiance, mul3(background(S, cr_d, depth, C), throughput));
            break;
        }
        f3    pos;
        float dist = t_near;
        float s = hyperion_s(depth - 5);
        float g = (1.0f - s) * P->g;
        float sigma_t_prime = est == VPO_EST_GLOBAL ? (1.0f - s) * sigma_t + s * sigma_t * (1.0f - P->g)
                                                    : ((1.0f - s) + s * (1.0f - P->g)) * sigma_t;
        float inv_sigma = 1.0f / sigma_t_prime;
        int   through = 0;
        for (;;)
        {
            dist += -vpo_logf(rng_next_a(&rng)) * inv_sigma;
            if (dist >= t_far) { through = 1; break; }
            pos = add3(cr_o, muls(cr_d, dist));
            if (rng_next_b(&rng) < vol_sigma_t(S, pos, sigma_t_prime, C) * inv_sigma)
            {
                if (est != VPO_EST_GLOBAL) ++num_scatters;
                break;
            }
        }
        if (through)
        {
            if (est == VPO_EST_GLOBAL)
            {
                radiance = add3(radiance, mul3(background(S, cr_d, i, C), throughput)); /* :1446-1452 */
                break;
            }
            cr_o = add3(cr_o, muls(cr_d, t_far));
            if (est == VPO_EST_BOUNDED) i++;
            continue;
        }
        C->scatters++;
        throughput = mul3(throughput, albedo);

        frame_t frame = make_frame(cr_d);
        {
            float s2 = hyperion_s(est == VPO_EST_GLOBAL ? i - 4 : num_scatters - 5);
            float sigma_t_prime2 = est == VPO_EST_GLOBAL ? (1.0f - s2) * sigma_t + s2 * sigma_t * (1.0f - P->g)
                                                         : ((1.0f - s2) + s2 * (1.0f - P->g)) * sigma_t;
            float inv_sigma2 = 1.0f / sigma_t_prime2;
            float ph = vpo_hg_eval(g, dot3(frame.n, sun_dir));
            float a;
            if (est == VPO_EST_DECOMP && spp > 10 && num_scatters > 20)
            {
                C->opacity_lookups++;
                a = vpo_expf(-sigma_t_prime2 * sample_volume(S, NULL, S->opacity, 1, pos)); /* :2190 */
            }
            else
                a = tr_scalar(S, boxMin, boxMax, pos, muls(sun_dir, 1e10f), inv_sigma2, sigma_t_prime2, &rng,
                              2u * (uint32_t)(est == VPO_EST_GLOBAL ? i : num_scatters), C);
            radiance = add3(radiance, mul3(sun_power, muls(muls(throughput, ph), a)));
        }
        float r0 = rng_next_a(&rng);
        float r1 = rng_next_b(&rng);
        f3 new_dir = normalize3(frame_to_world(&frame, hg_sample_local(g, r0, r1)));
        cr_o = pos;
        cr_d = new_dir;
        if (est != VPO_EST_DECOMP) i++;
    }
    radiance = muls(radiance, brightness);
    float heat = est == VPO_EST_DECOMP ? (float)num_scatters : (float)((double)i * 0.001);
    if (S->track_mode == 2)
    {
        float r[3] = {radiance.x, radiance.y, radiance.z};
        out[0] = out[1] = out[2] = 0.0f;
        out[channel] = fmaxf(r[channel], 0.0f) * 3.0f; /* :2311-2313 */
    }
    else
    {
        out[0] = fmaxf(radiance.x, 0.0f);
        out[1] = fmaxf(radiance.y, 0.0f);
        out[2] = fmaxf(radiance.z, 0.0f);
    }
    out[3] = heat;
}

void vpo_render_sample(const vpo_scene* S, const vpo_param* P, int x, int y, int frame, float out[4],
                       vpo_counters* C)
{
    vpo_counters local;
    memset(&local, 0, sizeof local);
    if (S->track_mode) sample_scalar(S, P, (uint32_t)x, (uint32_t)y, frame, out, &local);
    else if (S->estimator == VPO_EST_DECOMP) sample_decomp(S, P, (uint32_t)x, (uint32_t)y, frame, out, &local);
    else if (S->estimator == VPO_EST_BOUNDED) sample_bounded(S, P, (uint32_t)x, (uint32_t)y, frame, out, &local);
    else sample_global(S, P, (uint32_t)x, (uint32_t)y, frame, out, &local);
    local.samples = 1;
    if (C)
    {
        C->samples += local.samples; C->density_lookups += local.density_lookups;
        C->bound_lookups += local.bound_lookups; C->opacity_lookups += local.opacity_lookups;
        C->env_lookups += local.env_lookups; C->scatters += local.scatters; C->rng_draws += local.rng_draws;
        C->control_segments += local.control_segments;
    }
}

/* one frame: d_output[x + y*W] += (rgb, heat)  (kernel.cu:2315, host.cpp:631,640) */
void vpo_render_frame(const vpo_scene* S, const vpo_param* P, int frame, float* accum, int y0, int y1, int threads,
                      vpo_counters* C)
{
    vpo_counters tot;
    memset(&tot, 0, sizeof tot);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
    int W = (int)P->width;
#pragma omp parallel num_threads(threads)
    {
        vpo_counters loc;
        memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 1)
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < W; x++)
            {
                float o[4];
                vpo_render_sample(S, P, x, y, frame, o, &loc);
                float* a = accum + 4 * ((size_t)x + (size_t)y * W);
                a[0] += o[0]; a[1] += o[1]; a[2] += o[2]; a[3] += o[3];
            }
#pragma omp critical
        {
            tot.samples += loc.samples; tot.density_lookups += loc.density_lookups;
            tot.bound_lookups += loc.bound_lookups; tot.opacity_lookups += loc.opacity_lookups;
            tot.env_lookups += loc.env_lookups; tot.scatters += loc.scatters; tot.rng_draws += loc.rng_draws;
            tot.control_segments += loc.control_segments;
        }
    }
    if (C)
    {
        C->samples += tot.samples; C->density_lookups += tot.density_lookups;
        C->bound_lookups += tot.bound_lookups; C->opacity_lookups += tot.opacity_lookups;
        C->env_lookups += tot.env_lookups; C->scatters += tot.scatters; C->rng_draws += tot.rng_draws;
        C->control_segments += tot.control_segments;
    }
}

/* -------------------------------------------- A10: _precompute_opacity -- */
/* intersect_box kernel.cu:453-481 (tnear clamped at 0 inside) and _precompute_opacity :483-524 */
/* one voxel of the table: the march from the voxel centre (i, j, k) toward the light, kernel.cu:497-523 */
float vpo_opacity_voxel(const vpo_scene* S, const float light_dir[3], int i, int j, int k)
{
    f3  bmin = mk3(S->box_min[0], S->box_min[1], S->box_min[2]);
    f3  bmax = mk3(S->box_max[0], S->box_max[1], S->box_max[2]);
    f3  ext  = sub3(bmax, bmin);
    f3  d    = mk3(light_dir[0], light_dir[1], light_dir[2]);
    const float dt = 0.001f;
    /* normalized_coord :164-167, to_world :171 */
    f3 s0 = mk3(((float)i + 0.5f) / (float)S->nx, ((float)j + 0.5f) / (float)S->ny, ((float)k + 0.5f) / (float)S->nz);
    f3 start = add3(mul3(s0, ext), bmin);
    float tn, tf;
    int   hit = intersect_box(start, d, bmin, bmax, &tn, &tf);
    if (tn <= 0.0f) tn = 0.0f;
    float opacity = 0.0f;
    if (hit)
    {
        for (float t = tn; t < tf; t += dt)
        {
            f3 pos = add3(start, muls(d, t));
            opacity += sample_volume(S, S->grid_u8, S->grid_f32, S->linear, pos);
        }
        opacity *= dt;
    }
    return opacity;
}
void vpo_precompute_opacity(const vpo_scene* S, const float light_dir[3], float* out, int threads)
{
    int nx = S->nx, ny = S->ny, nz = S->nz;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++)
                out[(size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * (size_t)k)] = vpo_opacity_voxel(S, light_dir, i, j, k);
}
/* a list of voxels (full-size tables are an N^4 march: tests sample them), ijk = n x (i, j, k) */
void vpo_opacity_voxels(const vpo_scene* S, const float light_dir[3], const int* ijk, int n, float* out, int threads)
{
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
    for (int q = 0; q < n; q++) out[q] = vpo_opacity_voxel(S, light_dir, ijk[3 * q], ijk[3 * q + 1], ijk[3 * q + 2]);
}

/* ---------------------------------------------------- A11: scale / gamma -- */
/* __scale kernel.cu:2333-2341 */
void vpo_scale(float* dst, const float* src, int n, float s)
{
    for (int i = 0; i < 4 * n; i++) dst[i] = src[i] * s;
}
/* __gamma_correct kernel.cu:2348-2362; the host wrapper passes 1/gamma (:2361).  powf restated as
 * exp(log(x)*y) on the deterministic kernels (x<=0 -> 0). */
static inline float vpo_powf_pos(float x, float y)
{
    if (x <= 0.0f) return 0.0f;
    return vpo_expf(vpo_logf(x) * y);
}
void vpo_gamma_correct(float* dst, const float* src, int n, float s, float gamma)
{
    float ig = 1.0f / gamma;
    for (int i = 0; i < n; i++)
    {
        dst[4 * i + 0] = vpo_powf_pos(src[4 * i + 0] * s, ig);
        dst[4 * i + 1] = vpo_powf_pos(src[4 * i + 1] * s, ig);
        dst[4 * i + 2] = vpo_powf_pos(src[4 * i + 2] * s, ig);
        dst[4 * i + 3] = 1.0f;
    }
}

/* H2: Mat host.cpp:44-57 */
void vpo_mat(vpo_param* P, float X, float Y, float Z, float R, float G, float B)
{
    P->sigma_t[0] = X + R; P->sigma_t[1] = Y + G; P->sigma_t[2] = Z + B;
    P->albedo[0] = X / P->sigma_t[0]; P->albedo[1] = Y / P->sigma_t[1]; P->albedo[2] = Z / P->sigma_t[2];
    float f = fmaxf(fmaxf(P->sigma_t[0], P->sigma_t[1]), P->sigma_t[2]);
    P->sigma_t[0] /= f; P->sigma_t[1] /= f; P->sigma_t[2] /= f;
}

void vpo_math_array(int which, const float* in, float* out, int n)
{
    for (int i = 0; i < n; i++)
    {
        float s, c;
        switch (which)
        {
            case 0: out[i] = vpo_logf(in[i]); break;
            case 1: out[i] = vpo_expf(in[i]); break;
            case 2: vpo_sincosf(in[i], &s, &c); out[i] = s; break;
            case 3: vpo_sincosf(in[i], &s, &c); out[i] = c; break;
            case 4: out[i] = vpo_acosf(in[i]); break;
            case 5: out[i] = vpo_atanf(in[i]); break;
            default: out[i] = vpo_pow15f(in[i]); break;
        }
    }
}
