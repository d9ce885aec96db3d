/* volpath.h -- C ABI of libvolpath_hip.so: the MI355X-native replacement for the device
 * translation unit of RNG65536/CUDA-volpath (src/volumeRender_kernel.cu).
 *
 * Part 1 re-exports, under their original names, the 14 extern "C" entry points the reference
 * host binds (src/volumeRender.cpp:117-128 and :347-356); a maintainer swaps the CUDA TU for this
 * library and relinks.  Part 2 is additive: batched rendering, estimator / RNG
 * selection, pixel-tile sharding for multi-GPU, counters, and raw device-memory helpers so that a
 * C / ctypes caller needs no other GPU runtime binding.
 *
 * Only plain C types cross the boundary.  The CUDA vector types of the reference map onto the
 * layout-identical PODs below (float3 = 3 floats, float4 = 4 floats 16-byte aligned,
 * dim3 = 3 x uint32, cudaExtent = 3 x size_t).
 *
 * Error behaviour: Part 1 functions return void and, exactly like the reference
 * (checkCudaErrors -> exit(EXIT_FAILURE), src/cuda/helper_cuda.h:566-579; null volume -> exit(1),
 * kernel.cu:360-364), print a diagnostic and exit the process on failure.  Part 2 functions
 * return 0 on success or a negative VP_E* code and leave a message in vp_last_error().
 * There is NO CPU fallback anywhere: without a usable gfx950 device every entry point fails.
 */
#ifndef VOLPATH_H
#define VOLPATH_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } vp_float3;
typedef struct { float x, y, z, w; } vp_float4;
typedef struct { uint32_t x, y, z; } vp_dim3;
typedef struct { size_t width, height, depth; } vp_extent;

/* src/param.h:4-12 -- 44 bytes, passed by value to the reference kernels */
#ifndef VOLPATH_PARAM_DEFINED
#define VOLPATH_PARAM_DEFINED
typedef struct Param
{
    unsigned int width, height;
    float        density, brightness;
    vp_float3    albedo;
    float        g;
    vp_float3    sigma_t;
} Param;
#endif

/* ------------------------------------------------------------------------------------------
 * Part 1: the reference's kernel-TU interface
 * ------------------------------------------------------------------------------------------ */

/* kernel.cu:354-420 (declared host.cpp:349-353).  Uploads the density volume (uchar if
 * `quantized`, else float; x fastest), builds the local (max,min) bound table (replaces the
 * call-back into host.cpp:1269-1280) and publishes the descriptors.  boxmin/boxmax may be NULL
 * (box = +-(1, Ny/Nx, Nz/Nx)).  Caller keeps ownership of h_volume.  NULL volume -> exit(1). */
void init_cuda(void* h_volume, vp_extent volumeSize, bool quantized, const vp_float3* boxmin,
               const vp_float3* boxmax);
/* kernel.cu:422-439 (host.cpp:354): point / trilinear density sampling */
void set_texture_filter_mode(bool bLinearFilter);
/* kernel.cu:441-451 (host.cpp:355) */
void free_cuda_buffers(void);
/* kernel.cu:526-553 (host.cpp:128): optical depth toward light_dir[3], dt = 0.001 */
void precompute_opacity(const float* light_dir);
/* kernel.cu:1072-1229 (host.cpp:125): row-major float4 lat-long map, row 0 = zenith; copied */
void init_envmap(const vp_float4* HDRmap, int width, int height);
/* kernel.cu:1231-1250 (host.cpp:126) */
void free_envmap(void);
/* kernel.cu:1269-1283 (host.cpp:127): dir[3], disc radiance power[3] */
void set_sun(float* sun_dir, float* sun_power);
/* kernel.cu:2320-2328 (host.cpp:119-120): row-major 3x4, sizeofMatrix = 48 */
void copy_inv_view_matrix(float* invViewMatrix, size_t sizeofMatrix);
void copy_inv_model_matrix(float* invModelMatrix, size_t sizeofMatrix);
/* kernel.cu:2330-2331 (host.cpp:121-122): no-ops in the reference, no-ops here */
void init_rng(vp_dim3 gridSize, vp_dim3 blockSize, int width, int height);
void free_rng(void);
/* kernel.cu:2364-2370 (host.cpp:117-118): adds ONE sample per pixel of frame `spp` into the
 * caller-owned device buffer d_output[width*height]; asynchronous on the library stream.
 * gridSize/blockSize are accepted for signature compatibility and ignored (the launch shape is
 * the library's business).  In C++ the last parameter is `const Param&`, same ABI. */
#ifdef __cplusplus
void render_kernel(vp_dim3 gridSize, vp_dim3 blockSize, vp_float4* d_output, int spp, const Param& p);
#else
void render_kernel(vp_dim3 gridSize, vp_dim3 blockSize, vp_float4* d_output, int spp, const Param* p);
#endif
/* kernel.cu:2333-2346 / :2348-2362 (host.cpp:123-124): device pointers, in place allowed */
void scale(vp_float4* dst, vp_float4* src, int size, float scale);
void gamma_correct(vp_float4* dst, vp_float4* src, int size, float scale, float gamma);

/* ------------------------------------------------------------------------------------------
 * Part 2: additive interface
 * ------------------------------------------------------------------------------------------ */
enum
{
    VP_OK          = 0,
    VP_E_NODEVICE  = -1, /* no gfx950 device / HIP runtime error */
    VP_E_STATE     = -2, /* call order (e.g. render before init_cuda / init_envmap) */
    VP_E_ARG       = -3,
    VP_E_NOOPACITY = -4, /* frame > 10 with the decomposition estimator needs precompute_opacity */
    VP_E_NOMEM     = -5  /* device memory exhausted (hipErrorOutOfMemory) */
};

enum { VP_EST_GLOBAL = 0, /* __d_render, kernel.cu:1285-1591: global majorant (BASELINE config 2) */
       VP_EST_DECOMP = 1, /* __d_render_bounded_decomp, kernel.cu:1958-2318: the reference's live kernel */
       VP_EST_BOUNDED = 2 /* __d_render_bounded, kernel.cu:1667-1952: local majorant, no control component,
                             800 tracked segments at most, heat = segments * 0.001, never reads the opacity volume */ };
enum { VP_RNG_SAMPLERH = 0, /* src/sampler.h bit-compatible streams: THE PARITY MODE -- the reference's generator, seeding and order of
                               draws (sampler.h:3-46; Tr_spectral draws from the path's own sequential stream), i.e. what a run of the
                               reference computes sample for sample up to the arithmetic contract of DESIGN.md section 2 */
       VP_RNG_PHILOX   = 1, /* Philox2x32-10, counter = (draw/2, x<<16|y), key = (frame ^ key0) + key1.  SAME ESTIMATOR, BUILD-DEFINED
                               STREAM (north_star: a counter-based generator replaces sampler.h): same free flights, collision tests and
                               transmittance flags, but shadow rays draw from sub-streams -- counter word 0 = 0x80000000 +
                               ((2 * depth + ray) << 20) + step, 2^20 pairs each: a shadow ray of more steps (a majorant above 3e5 per
                               unit length; the default medium has 800) would run into the next sub-stream -- and the phase function is
                               sampled before the shadow ray.  Defined by oracle/vp_oracle.c; tied to the sampler.h images statistically
                               (tests/test_parity_gpu.py::test_full_size_estimators_and_builds_converge_to_one_image) */
       VP_RNG_PHILOX7  = 2  /* Philox2x32-7 (the fewest rounds Random123 documents as Crush-resistant), same counter / key / sub-streams;
                               built for the shipped configuration (spectral tracking, passive environment) */ };

const char* vp_last_error(void);
const char* vp_version(void);
int  vp_device_count(void);
int  vp_set_device(int device);       /* device of the CURRENT context; before its first GPU call; default 0 */

/* Contexts.  The reference keeps its scene in file-scope statics and __constant__ symbols: one scene, one device per process
 * (kernel.cu:148-151, :621-629; cudaSetDevice(0) src/denoiser.cpp:94-97).  Here all of that state lives in a context.  Every
 * entry point of this header -- Part 1 included -- acts on the calling thread's current context; a thread that never set one
 * uses the process-wide default context, so the reference host binds the 14 Part-1 symbols unchanged.  One context per GPU
 * gives a single-process multi-GPU host (host/main.cpp --gpus N: N contexts, pixel tiles dealt by vp_set_shard, one RCCL
 * reduce).  A context is not thread-safe; different contexts may be driven from different threads. */
typedef struct vp_ctx vp_ctx;
vp_ctx* vp_ctx_create(int device);         /* NULL on failure (vp_last_error of the current context says why) */
int     vp_ctx_destroy(vp_ctx* ctx);       /* frees its device memory; the current context becomes the default one if it was ctx */
int     vp_ctx_set_current(vp_ctx* ctx);   /* NULL = the default context */
vp_ctx* vp_ctx_get_current(void);          /* NULL while the default context is current */
int     vp_ctx_device(void);               /* device index of the current context */
/* dst[i] += src[i] for n float4 on the current context's stream (device pointers of ITS device): sums per-shard accumulators
 * where no collective is available (several contexts on one GPU) */
int     vp_accumulate(vp_float4* dst, const vp_float4* src, size_t n);
int  vp_set_stream(void* hip_stream); /* hipStream_t to launch on; NULL = library-owned stream */
void* vp_get_stream(void);            /* the hipStream_t the current context launches on (for a collective queued behind a render) */
int  vp_synchronize(void);

int vp_set_estimator(int est);                         /* default VP_EST_DECOMP */
int vp_set_rng(int mode, uint32_t key0, uint32_t key1); /* default VP_RNG_SAMPLERH */
/* Environment lighting.  VP_ENV_PASSIVE is the reference's shipped build (PASSIVE_ENVMAP 1, kernel.cu:21): escaping paths
 * look the environment up.  VP_ENV_MIS is its compiled-out alternative: luminance CDFs built in init_envmap
 * (kernel.cu:1144-1210) and one-sample MIS between phase-function and environment sampling after each collision
 * (kernel.cu:2220-2297, MULT_PDF 0, PRE_WARP 1); only unscattered paths then see the environment directly. */
/* Collision sampling.  VP_TRACK_SPECTRAL is the reference's shipped build (SPECTRAL_TRACKING 1, kernel.cu:15-34): one path
 * for the three channels with history-aware collision probabilities.  The other two are its compiled-out alternatives:
 * VP_TRACK_SCALAR = SPECTRAL_TRACKING 0 (one extinction coefficient = density, throughput *= albedo per collision, scalar
 * shadow rays), VP_TRACK_MULTI_CHANNEL = MULTI_CHANNEL 1 (the same with coefficient density * sigma_t[channel], the channel
 * drawn per sample and written times three, kernel.cu:1993-1994, :2311-2313).  Both ignore the local bound (kernel.cu:2063)
 * and exist with VP_ENV_PASSIVE only. */
enum { VP_TRACK_SPECTRAL = 0, VP_TRACK_SCALAR = 1, VP_TRACK_MULTI_CHANNEL = 2 };
int vp_set_tracking(int mode);                         /* default VP_TRACK_SPECTRAL */
/* render_kernel renders up to max_frames consecutive frames per launch when the host asks for frame f right after f-1 with
 * unchanged state, stages them, and serves the following calls from the staged frames (bit-identical to one launch per
 * frame).  The first frame of a run is rendered alone, then batches of 32, 64, ... frames, each with its
 * successor queued behind it; beyond 64 a batch is at most half of what the run has accumulated.  A setter or a camera move
 * stops the batches in flight within a fraction of a millisecond.  Default 256; 0 or 1 = one launch per call.  Env: VP_LOOKAHEAD. */
int vp_set_lookahead(int max_frames);
enum { VP_ENV_PASSIVE = 0, VP_ENV_MIS = 1 };
int vp_set_envmap_sampling(int mode);                  /* default VP_ENV_PASSIVE */
/* test hook: the tables of the current environment: cdf_y[h], cdf_x[w*h] (row CDFs), HDRpdfnormAlt; any may be NULL */
int vp_get_env_tables(float* cdf_y, float* cdf_x, float* pdfnorm_alt);
/* brick edge (power of two, 1 = the reference's per-voxel table) used by the NEXT init_cuda */
int vp_set_bound_brick(int brick);
/* Pixel-tile sharding: this context renders the 8x8 pixel tiles (tx, ty) with vp_tile_owner(tx, ty, world) == rank: within
 * a tile row every world-th tile, the rows shifted against each other by a hash of the row index, so that neither columns
 * nor rows nor diagonals of the image belong to one rank whatever tiles_x % world is. */
int vp_set_shard(int rank, int world);
int vp_tile_owner(unsigned tx, unsigned ty, int world);

/* Adds frames [first_frame, first_frame + n_frames) into d_output[width*height] (device).
 * Per pixel the samples are added in frame order, so the result equals n_frames successive
 * render_kernel calls bit for bit.  Asynchronous. */
int vp_render_frames(vp_float4* d_output, int first_frame, int n_frames, const Param* p);

typedef struct
{
    uint64_t samples;
    uint64_t density_lookups; /* trilinear density evaluations of the estimator */
    uint64_t density_loads;   /* of those, the ones that issued a global load */
    uint64_t bound_lookups;
    uint64_t opacity_lookups;
    uint64_t env_lookups;
    uint64_t scatters;
    uint64_t rng_draws;
} vp_counters;
/* counters are collected only while enabled (a separately compiled kernel variant) */
int vp_enable_counters(int on);
int vp_read_counters(vp_counters* out, int reset); /* synchronises */

/* kernel time of the render launches since the last reset, measured with HIP events on the
 * launch stream; synchronises.  At most 64 launches are kept pending: older ones are folded into the running sum when
 * their events have completed, so a host that never asks does not accumulate events. */
int vp_render_time_ms(double* total_ms, int* launches, int reset);

/* The same per pixel class: ms[0] the general kernel (pixels whose camera ray can meet the medium), ms[1] the
 * light kernel (the whole chord is certified empty), ms[2] the fill of the pixels whose ray misses the box; HIP events around each
 * kernel on the stream it runs on (the first two run side by side, so the times overlap).  pixels[] = the pixels of each class
 * in the current lists of this context.  Either pointer may be NULL.  Synchronises. */
int vp_render_class_time_ms(double ms[3], unsigned pixels[3], int reset);

/* How the last render launch of this context took its general pixels' camera rays to the medium: 0 = the integrator walked them
 * itself (one-frame launches, counting launches, scalar / MIS builds, VP_NO_APPROACH), 1 = approach_k / approach_local_k walked the
 * certified-empty stretch ahead of it, 2 = as 1 with the walked throughput looked up by the number of steps (a global-majorant
 * medium whose null collision in empty space is not neutral).  Never changes a result. */
int vp_last_approach_mode(void);
/* 1 if that walk (decomposition estimator, launches of 64 frames and more) read the restart segments of each pixel's camera ray from
 * the per-view table (approach_segments_k) instead of setting them up per sample; VP_NO_APPROACH_TABLE=1 switches the table off. */
int vp_last_approach_table(void);
/* 1 if the last render call of this context wrote its light class (pixels whose camera ray meets empty cells only) as per-pixel
 * constants (miss_fill_k: a null collision in empty space leaves a throughput of 1 as it is in this medium), 0 if it integrated it. */
int vp_last_light_const(void);
/* How the last render launch of the decomposition estimator read its brick table: 0 from global memory, 1 as 16-bit (max,min) pairs
 * staged through LDS, 2 as 2-bit codes into a four-entry palette staged through LDS beside the cold per-path state (tables with at
 * most four distinct pairs -- binary volumes --, achromatic media, timed launches of the counter-based streams).  Performance
 * only: the three forms render the same bits (VP_NO_LDS_BOUNDS / VP_NO_LDS_COMPACT select them). */
int vp_last_lds_form(void);
/* test hook: look-ahead batches this context has launched so far (render_kernel's staged frames), and how many of them were told to
 * stop while they were still running (a setter, a camera move); either pointer may be NULL */
int vp_lookahead_stats(unsigned* launched, unsigned* cancelled_in_flight);
/* Builds everything a render of this Param would build first -- the per-pixel tables of the current camera, the pixel lists
 * of the shard, the sun table -- and waits for it.  A host that moves the camera may call it to take that work out of its
 * first frame; bench.py times it (per_camera_setup_ms).  Not needed for correctness: render_kernel does the same on demand. */
int vp_prepare(const Param* p);
/* Sizes the per-launch sample staging for a coming vp_render_frames(…, n_frames, p) job of this context now (the reference's host
 * allocates its buffers at start-up too): the first launch of the job then finds its buffer instead of allocating up to 16 GiB
 * inside the caller's timed region.  Optional; never changes a result; does nothing for one-frame calls. */
int vp_reserve_frames(const Param* p, int n_frames);
/* test hook: the pixel lists of this context for p (after vp_prepare): dst[0 .. counts[0]) the general pixels, then counts[1]
 * light ones, then counts[2] whose camera ray misses the box, each y << 16 | x in tile order; dst may be NULL to ask for the counts */
int vp_get_pixel_lists(const Param* p, uint32_t* dst, size_t count, unsigned counts[3]);

/* the derived tables, for tests: bound table dims/brick and a device->host copy */
int vp_get_bound_table(void* dst, size_t bytes, int* bnx, int* bny, int* bnz, int* brick, int* radius);
int vp_get_opacity(float* dst, size_t count);
/* the per-pixel table of the current estimator / camera / volume for a width x height image, 8 floats per pixel:
 * [0..2] where the restart crawl in front of the volume ends (local-majorant estimators; the camera origin otherwise),
 * [3] its segment and draw counts (bits: segments | draws << 16), [4] the distance from there up to which the camera ray is
 * certified to meet only empty cells, [5] the pixel class (0 general; 1 the whole chord is certified empty: light kernel; 2 the
 * camera ray misses the box: one constant per pixel), [6..7] unused.  Test hook for the certificates. */
int vp_get_pixel_table(const Param* p, float* dst, size_t count);
/* Counter-based streams (VP_RNG_PHILOX / VP_RNG_PHILOX7): a shadow ray draws from a sub-stream of its own, so the path's later
 * draws do not depend on the number of steps it takes, and a sun shadow ray ends once it has only empty cells in front of it.
 * dst[cell] (x fastest, count >= nx*ny*nz) = that distance from anywhere in the cell, in units of *step (world units), or
 * 0xffff = unknown (the ray is walked to its end).  Test hook for the certificate; VP_NO_SUN_CLIP=1 switches the table off. */
int vp_get_sun_clip_table(unsigned short* dst, size_t count, float* step);
/* Exit flights (any stream): a path in empty space that can meet empty cells only on its way out of the box, and whose null
 * collisions leave its throughput bit for bit as it is, ends with the environment whatever it draws, and is ended at once instead of
 * walking there.  The certificate for the cells: dst[A * nx*ny*nz + cell] (x fastest within a plane, count >= 3*nx*ny*nz), A = the
 * dominant axis of a direction in cell units (d * N / box extent), bit (e_A > 0) | (e_B > 0) << 1 | (e_C > 0) << 2 with (B, C) the
 * other two axes in increasing order: every cell a ray from anywhere in `cell` with a direction of that class can meet is empty and
 * has empty neighbours.  Test hook for the certificate; VP_NO_EXIT=1 switches the table off. */
int vp_get_exit_table(unsigned char* dst, size_t count);
/* 0: off (every path walks to the box exit); 1: the global-majorant estimator only, where that walk is 800 null collisions per
 * unit length; 2: the decomposition estimator as well (uchar bound tables with at most four distinct maxima), where the walk is one
 * free flight per restart segment.  Until this call is made (and without VP_EXIT_LOCAL): mode 2 on the counter-based streams
 * (+1...4 % on the decomposition workloads), mode 1 on sampler.h.  Performance only: the same bits in every mode. */
int vp_set_exit_flights(int mode);
/* dst[n] = throughput of an unscattered path of the global-majorant estimator after n null collisions in empty space, n < count
 * (spectral tracking: the weight of such a collision is 1 only up to rounding; the light kernel looks the product up by n).
 * Test hook: the sequence is three float32 operations per step and can be restated anywhere. */
int vp_get_null_collision_table(const Param* p, float* dst, size_t count);

/* building blocks exposed for parity tests (device execution, host arrays) */
int vp_test_math(int which, const float* in, float* out, int n);
int vp_test_rng(int mode, uint32_t x, uint32_t y, uint32_t frame, uint32_t k0, uint32_t k1, int n, float* out);
int vp_test_sample_density(const float* pos_xyz, float* out, int n);
/* component hooks for known-answer tests against float64 closed forms (no oracle involved):
 * HGPhaseFunction::sample through Frame (kernel.cu:557-598, the phase-function block :2301-2303) and ::evaluate (:600-603);
 * intersectBox (kernel.cu:654-680) against the current volume box; eval_envmap (kernel.cu:956-973, dir_to_uv :882-895) */
int vp_test_hg(const float* g, const float* r0, const float* r1, const float* normal_xyz, const float* cos_query, float* dir_xyz,
               float* eval, int n);
int vp_test_intersect_box(const float* origin_xyz, const float* dir_xyz, int* hit, float* tnear, float* tfar, int n);
int vp_test_eval_envmap(const float* dir_xyz, float* rgb, int n);

/* The procedural Julia-set volume of the reference (FractalJuliaSet, kernel.cu:84-140) voxelised at
 * texel centres over [-1,1]^3 to an n^3 uchar grid (0 / 255), x fastest; written to HOST memory so it
 * can be handed to init_cuda like any other volume. */
int vp_julia_voxelize(int n, unsigned char* host_out);
/* A FLAGGED SYNTHETIC stand-in for the WDAS cloud of BASELINE configs 4/5 (neither the data set nor OpenVDB exists in the build
 * image): five octaves of hashed-lattice value noise, thresholded, with a soft spherical edge, voxelised at texel centres over
 * [-1,1]^3 to n^3 float densities in [0,1] (not binary), x fastest, written to HOST memory -- to be dumped with dump_dense_volume
 * and read back with loadBinaryFile like a converted .vdb.  Deterministic in (n, seed); the oracle restates it bit for bit. */
int vp_cloud_voxelize(int n, uint32_t seed, float* host_out);

/* raw device memory helpers */
void* vp_malloc(size_t bytes);
int   vp_free(void* dptr);
int   vp_memset(void* dptr, int value, size_t bytes);
int   vp_upload(void* dptr, const void* src, size_t bytes);
int   vp_download(void* dst, const void* dptr, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* VOLPATH_H */
