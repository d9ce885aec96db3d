/* vp_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement of the hot path of RNG65536/CUDA-volpath.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline -- never as something the product calls.
 *
 * PARITY PIN STATUS (see DESIGN.md "Oracle"):
 *   - the reference's integrator lives in one CUDA translation unit (src/volumeRender_kernel.cu)
 *     that needs cuda_runtime.h and texture hardware: it is UNBUILDABLE in this image without
 *     writing stand-in headers, so it was not built.  The reference has no tests and no golden
 *     vectors of its own.
 *   - pinned: RNG (sampler.h) by the known-answer values recorded from the reference in
 *     SURVEY.md section 4; the sun/sky inputs by oracle/_ref (the reference's own Hosek sources
 *     compiled where they lie); the Julia voxeliser by the occupancy figure and the integrator
 *     by the per-sample work counters of SURVEY.md section 6 (statistical pins).
 *   - pinned since round 3 by an output the reference itself holds: the GEOMETRY chain -- FractalJuliaSet and its voxelisation,
 *     the volume box, the camera matrix, field of view and pixel-to-ray map, intersectBox -- by the silhouette of the reference's
 *     own screenshot of the Julia scene (/root/reference/2.jpg -> tests/golden/ref_julia_silhouette.npz, IoU 0.97 at the
 *     reference's default camera distance; tests/test_oracle_cpu.py).
 *   - pinned since round 4 by the same output, statistically: the RADIOMETRY of the integrator as a whole -- sun power x phase
 *     function x albedo x transmittance x multiple scattering x display transform -- by the INTERIOR of that screenshot: per 16x16
 *     block the luminance of this oracle's render of the fitted pose under the fitted sun (two parameters; exposure not fitted)
 *     against the screenshot's, Pearson >= 0.98 (0.996 on the GPU at full resolution), absolute level within 13 %; the reference's
 *     default medium and no other (g = 0, density 80, albedo 0.8 miss by 4-8x the residual).  tests/golden/ref_julia_interior.npz,
 *     tests/test_oracle_cpu.py::test_julia_interior_matches_the_references_own_screenshot.  And the one quirk with a first-order
 *     effect on that image, Q9 (the "Hyperion" reduction): the same fit WITHOUT it (vpo_debug_set_what_if) misses the screenshot by
 *     more than twice the residual (test_julia_interior_prefers_the_reference_as_read).
 *   - everything else -- the quirks Q4-Q8, the tex3D rule, bounds, opacity as such: "parity unpinned" -- a line-by-line
 *     restatement citing file:line.
 */
#ifndef VP_ORACLE_H
#define VP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* param.h:4-12 -- 44-byte POD handed to the kernels by value. */
typedef struct
{
    uint32_t width, height;
    float    density, brightness;
    float    albedo[3];
    float    g;
    float    sigma_t[3];
} vpo_param;

enum { VPO_RNG_SAMPLERH = 0, VPO_RNG_PHILOX = 1 /* philox2x32-10 */, VPO_RNG_PHILOX7 = 2 /* philox2x32-7 */ };
enum { VPO_EST_GLOBAL = 0 /* __d_render, kernel.cu:1285 */, VPO_EST_DECOMP = 1 /* __d_render_bounded_decomp, :1958 */,
       VPO_EST_BOUNDED = 2 /* __d_render_bounded, :1667 */ };

typedef struct
{
    /* density volume, x fastest (load_vdb.cpp:47-50) */
    int            nx, ny, nz;
    const uint8_t* grid_u8;  /* quantized volume, or NULL */
    const float*   grid_f32; /* float volume, or NULL */
    float          box_min[3], box_max[3];
    int            linear; /* set_texture_filter_mode */
    /* local bounds: (max,min) byte or float pairs per brick of `brick`^3 voxels; brick==1 is the reference's dense table */
    int            brick;
    int            bnx, bny, bnz;
    const uint8_t* bounds_u8;
    const float*   bounds_f32;
    /* precomputed optical depth toward the sun (kernel.cu:483-553) or NULL */
    const float* opacity;
    /* environment */
    const float* env; /* float4 rows, row 0 = zenith */
    int          env_w, env_h;
    float        sun_dir[3];
    float        sun_power[3];          /* directional: original * pi*(0.45/94)^2, kernel.cu:1275-1277 */
    float        sun_power_original[3]; /* disc radiance, kernel.cu:1271 */
    float        inv_view[12];          /* row-major 3x4 camera-to-world, kernel.cu:626,631-649 */
    /* estimator / rng */
    int      estimator;
    int      rng_mode;
    uint32_t seed[2]; /* Philox: key = (frame ^ seed[0]) + seed[1] */
    /* active environment sampling + one-sample MIS (!PASSIVE_ENVMAP, kernel.cu:2220-2297); 0 = the shipped passive mode */
    int          env_mis;
    const float* env_cdf_y; /* env_h */
    const float* env_cdf_x; /* env_w * env_h */
    float        env_pdfnorm_alt;
    /* 0 = SPECTRAL_TRACKING 1 (shipped); 1 = SPECTRAL_TRACKING 0; 2 = MULTI_CHANNEL 1 (kernel.cu:15-34); 1 and 2 with env_mis 0 only */
    int          track_mode;
} vpo_scene;

typedef struct
{
    uint64_t samples;
    uint64_t density_lookups;
    uint64_t bound_lookups;
    uint64_t opacity_lookups;
    uint64_t env_lookups;
    uint64_t scatters;
    uint64_t rng_draws;
    uint64_t control_segments;  /* A1: collisions found in a segment whose bound minimum is positive (the control component of quirk Q7 in use) */
} vpo_counters;

/* one sample per pixel of frame `frame`, accum[pix] += (rgb, heat).  rows [y0,y1).  threads<=0: all. */
void vpo_render_frame(const vpo_scene* S, const vpo_param* P, int frame, float* accum, int y0, int y1,
                      int threads, vpo_counters* C);
/* one (x,y,frame) sample, returned instead of accumulated */
void vpo_render_sample(const vpo_scene* S, const vpo_param* P, int x, int y, int frame, float out[4],
                       vpo_counters* C);

/* building blocks (each cites the reference lines it follows in vp_oracle.c) */
uint32_t vpo_hash(uint32_t seed);
void     vpo_rng_stream(int mode, uint32_t x, uint32_t y, uint32_t frame, uint32_t k0, uint32_t k1, int n,
                        float* out);
void     vpo_philox2x32_10(const uint32_t ctr[2], uint32_t key, uint32_t out[2]);
void     vpo_philox2x32_7(const uint32_t ctr[2], uint32_t key, uint32_t out[2]);
void     vpo_julia_voxelize(int n, uint8_t* grid);
void     vpo_cloud_voxelize(int n, uint32_t seed, float* grid); /* the flagged synthetic cloud of the 512^3 workloads */
int      vpo_bound_radius(int nx, float search_radius);
void     vpo_bounds_u8(const uint8_t* grid, int nx, int ny, int nz, int radius, int brick, uint8_t* out);
void     vpo_bounds_f32(const float* grid, int nx, int ny, int nz, int radius, int brick, float* out);
void     vpo_precompute_opacity(const vpo_scene* S, const float light_dir[3], float* out, int threads);
/* single voxels of that table (kernel.cu:497-523): what tests compare full-size GPU tables with */
float    vpo_opacity_voxel(const vpo_scene* S, const float light_dir[3], int i, int j, int k);
void     vpo_opacity_voxels(const vpo_scene* S, const float light_dir[3], const int* ijk, int n, float* out, int threads);
float    vpo_sample_density(const vpo_scene* S, const float pos[3]);
void     vpo_sample_bound(const vpo_scene* S, const float pos[3], float out_max_min[2]);
float    vpo_sample_opacity(const vpo_scene* S, const float pos[3]);
void     vpo_build_env_tables(const float* env, int width, int height, float* cdf_y, float* cdf_x, float* pdfnorm_alt);
void     vpo_debug_set_what_if(int mask); /* test hook: variants of the restatement for the radiometric pin (0 = the restatement; vp_oracle.c) */
uint64_t vpo_debug_shadow_overflow(void); /* test hook: shadow rays that drew more than the 2^20 pairs of their sub-stream (must stay 0) */
uint64_t vpo_debug_mis_zero_pdf(void); /* test hook: zero-pdf `continue`s taken so far (kernel.cu:2266) */
void     vpo_eval_envmap(const vpo_scene* S, const float dir[3], float rgb[3]);
void     vpo_hg_sample(float g, const float n[3], float u0, float u1, float out[3]);
float    vpo_hg_eval(float g, float cos_theta);
int      vpo_intersect_box(const float o[3], const float d[3], const float bmin[3], const float bmax[3],
                           float* tnear, float* tfar);
void     vpo_scale(float* dst, const float* src, int n, float s);
void     vpo_gamma_correct(float* dst, const float* src, int n, float s, float gamma);
void     vpo_mat(vpo_param* P, float X, float Y, float Z, float R, float G, float B);
void     vpo_default_camera(float inv_view[12]);
void     vpo_set_sun(vpo_scene* S, const float dir[3], const float power[3]);
/* math kernels on arrays: which: 0 log 1 exp 2 sin(a) 3 cos(a) 4 acos 5 atan 6 pow1.5 */
void vpo_math_array(int which, const float* in, float* out, int n);

#ifdef __cplusplus
}
#endif
#endif
