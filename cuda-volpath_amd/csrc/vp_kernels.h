// vp_kernels.h -- launch descriptors and host-callable launchers of vp_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "vp_device.h"

#define VP_BLOCK 256
#ifndef VP_BLOCK_LDS
#define VP_BLOCK_LDS 512              // workgroup size of the LDS-bound-table variant
#endif
#ifndef VP_MIN_WAVES
#define VP_MIN_WAVES 1                // launch-bounds hint: waves per SIMD the register budget must allow
#endif
#define VP_LDS_BYTES_PER_CU (160 * 1024)   // gfx950 (MI355X_MICROARCH.md); the static_assert at render_k's cold state ties the occupancy budgets to it
#define VP_LDS_BOUND_ENTRIES 32768     // (max,min) byte pairs staged in LDS: 64 KiB
#ifndef VP_CHUNK
#define VP_CHUNK 128  // samples a wave takes from a queue per atomic (256 until round 5: at the bench's launch size 128 is +1 % on the
                      // Julia workloads -- finer hand-out at the end of a band --, 64 the same, 512 / 1024 -2 / -4 %: r05_knob_sweeps.txt)
#endif
// One sample queue per XCD: each hands out a contiguous band of the image (all frames of it), so the rays an XCD's
// L2 serves stay in one slab of the volume; a wave starts on the queue of its own XCD and moves on when it runs dry.
#define VP_NQUEUES 8
#define VP_QUEUE_STRIDE 16  // words between queue heads (one cache line each)
// the inner tracking loop of a wave runs until this many lanes are parked on an event, or until
// a parked lane has waited this many steps
// (28 since round 5 -- 24 before: C2 +1.3 % at the bench's launch size with a 0.15 % noise floor, the achromatic local kernels +0...0.3 %;
// the chromatic local kernels take 32: vp_render.cpp, profiles/experiments/r05_knob_sweeps.txt)
#ifndef VP_WAIT_LANES
#define VP_WAIT_LANES 28
#endif
#ifndef VP_WAIT_ITERS
#define VP_WAIT_ITERS 16
#endif
// lanes that must be waiting for a restart-segment set-up before it runs in the middle of a pass (it always runs at the first
// step of a pass)
#ifndef VP_END_LANES
#define VP_END_LANES 4
#endif
#ifndef VP_SETUP_LANES
#define VP_SETUP_LANES 8
#endif
// tracking steps per pass of the inner loop: the wave-level bookkeeping (ballots, wait policy) is paid once per pass
#ifndef VP_LIGHT_MIN_WAVES
#define VP_LIGHT_MIN_WAVES 8   // the light kernels fit 64 vector registers: two of their waves beside four of a 96-register kernel
#endif
#ifndef VP_PROFILE_BLOCKS
#define VP_PROFILE_BLOCKS 0   // 1: the counting kernels also carry cycle stamps, loop statistics and block tallies (scripts/block_profile.py)
#endif
#ifndef VP_GLOBAL_MIN_WAVES
#define VP_GLOBAL_MIN_WAVES 7  // global-majorant kernels: waves per SIMD their register budget is held to.  Seven = 72 registers, no spill, since
                               // intersect_box runs axis by axis (vp_device.h; round 5): C2 3010-3030 -> 3090.  History: six since the cold
                               // per-path state lives in LDS (round 4: 79 registers; it came out at 72 = seven by itself until the event section
                               // read its uniforms from LDS: 79 again, and held to 72 THEN it spilled six: 2743 vs 2999); with the cold state in
                               // registers six cost three spilled registers and lost to five (1190 vs 1341 Msamples/s on C2, round 3)
#endif
#ifndef VP_LIGHT_LOCAL_MIN_WAVES
#define VP_LIGHT_LOCAL_MIN_WAVES 7   // the local-majorant light kernels need 66-68 registers with the Philox2x32-10 and sampler.h streams:
                                     // eight waves cost three or four spilled registers, seven (72) none
#endif
#ifndef VP_LIGHT_STEPS_PER_PASS
#define VP_LIGHT_STEPS_PER_PASS 16
#endif
#ifndef VP_STEPS_PER_PASS
#define VP_STEPS_PER_PASS 4
#endif
// exit flights: the lane counter value from which a lane is tested
#define VP_EXIT_TRIP 64

namespace vp
{
// Pixel-tile deal (include/volpath.h vp_tile_owner): tile (tx, ty) belongs to rank (tx + tile_row_shift(ty)) % world, i.e.
// within a tile row every world-th tile, rows shifted against each other by a hash of the row index.  The host lists a rank's
// tiles (vp_tables.cpp pixel lists); the kernels read the list.
__host__ __device__ inline unsigned tile_row_shift(unsigned ty, unsigned world) { return ((ty * 0x9E3779B1u) >> 15) % world; }
// same values as the VP_EST_* / VP_RNG_* enums of include/volpath.h
constexpr int EST_GLOBAL = 0, EST_DECOMP = 1, EST_BOUNDED = 2;
constexpr int RNG_SAMPLERH = 0, RNG_PHILOX = 1, RNG_PHILOX7 = 2;

// one render launch: frames [frame0, frame0+nframes) x a list of pixels (one class of the pixels this rank owns)
struct LaunchDev
{
    ParamDev P;
    int      frame0, nframes;
    unsigned nslots;        // pixels of THIS launch (its slice of the pixel list) = samples per frame
    unsigned total_items;   // nframes * nslots
    const unsigned* pixels; // pixel of slot s: pixels[s] = y << 16 | x.  The rank's pixels (those of its 8x8 tiles, tile by tile) are
                            // listed class by class: general pixels, then "light" pixels whose camera ray meets empty cells only;
                            // a launch covers one class
    unsigned stage_stride;  // samples per frame in `stage`, all classes together
    unsigned slot_base;     // first sample slot of this launch's class within a frame of `stage`
    float4*  out;           // W*H accumulator (caller-owned)
    float4*  stage;         // [nframes][stage_stride] per-sample results, or null = accumulate directly
    const float4* crawl;    // per pixel two float4, or null: [0] = where the restart crawl in front of the volume ends and its segment /
                            // draw counts (crawl_table_k, local-majorant estimators), [1].x = certified-empty distance from there,
                            // [1].y = pixel class (0 general, 1 the whole chord is certified empty, 2 the ray misses the box)
    unsigned* queue;        // VP_NQUEUES sample-queue heads, VP_QUEUE_STRIDE words apart (zeroed before the launch)
    unsigned chunk_fshift;  // log2 of the frames a chunk spans (0: a chunk is VP_CHUNK pixels of one frame; 6: four pixels x 64 frames); nframes is a multiple
    unsigned q_start[VP_NQUEUES + 1];  // slot range [q_start[q], q_start[q+1]) of a frame that queue q hands out
    unsigned long long* counters;  // work counters, loop statistics and block tallies (vp_state.h kCounterWords) or null
    unsigned key0, key1;    // Philox key
    unsigned wait_lanes, wait_iters;  // inner-loop exit policy (VP_WAIT_LANES / VP_WAIT_ITERS)
    unsigned end_lanes;     // lanes that must ask for the path-end chain (environment, write, refill) before it runs in a visit (VP_END_LANES;
                            // it also runs every fourth visit, and whenever no lane of the wave is tracking)
    unsigned setup_lanes;   // lanes that must ask for a segment set-up before it runs mid-pass (VP_SETUP_LANES; 1 = at every step)
    // Counter-based streams: per cell of the volume, the distance (in steps of clip_ds, 0xffff = unknown) beyond which a ray from
    // anywhere in the cell toward the sun meets empty cells only (sun_clip_k); a sun shadow ray ends there.  Null = walk to the end.
    const unsigned short* sun_clip;
    float    clip_ds;
    unsigned count_clips;   // counting build: 0 = walk every shadow ray to its end (density_lookups = the estimator's count, as the oracle's),
                            // 1 = end them where the timed build does (VP_DEBUG_COUNT_CLIPS: block tallies of what the timed kernels execute)
    const float* thr_table; // light kernel of the global-majorant estimator: thr_table[n] = throughput after n null collisions in empty
    unsigned thr_n;         // space (thr_table_k: a function of n alone there), n < thr_n; beyond the table the recurrence is run
    // Counter-based streams: approach_k (global majorant) / approach_local_k (decomposition) has walked the camera ray of every sample
    // of this launch through its certified-empty stretch already and left (distance reached, draw pairs used) / (segment origin, pairs
    // used) in the sample's staging slot; the integrator takes a new sample up from there.  0 = start as the reference does.
    // The brick table of the decomposition estimator in its compact form (render_k<..., LDSB = 2>): where it holds at most four distinct
    // (max,min) byte pairs -- a binary volume: three -- 2-bit codes (sixteen per word, brick order, padded to 16 bytes) and the palette
    // (pair 0 | pair 1 << 16, pair 2 | pair 3 << 16); built with the volume (vp_context.cpp), null otherwise
    // (the fields themselves: at the END of this struct -- appended, so that the offsets the other kernels read do not move)
    uint2*   approach_aux;    // decomposition estimator: the stream's state per staging slot (the slot holds the segment origin and the distance
                              // reached in it): .x = pair index (counter-based) / sampler.h's two words
    unsigned approach;        // 1: the walk's null collisions leave the throughput at 1; 2 (global majorant): look it up by their number in thr_table
    unsigned approach_fshift; // log2 of the frames a wave of the approach kernels spans (6 where the launch has 64 frames or more)
    unsigned approach_steps;  // most free-flight steps (restart segments) the walk makes per sample (the integrator does what is left)
    // Exit flights (render_k).  A path that can meet certified-empty cells only on its way out of the box, and whose null collisions
    // there leave its throughput bit for bit as it is, ends with the environment whatever it draws: it is ended at once.
    // exit_oct: three byte planes (one per dominant axis of a direction in cell units), per cell eight bits, one per class of signs:
    // every cell of the quarter pyramid that opens from this cell in that class of directions has no non-empty cell in its 3x3x3
    // neighbourhood (exit_dir_slice_k); null = off.
    const unsigned char* exit_oct;
    int      exit_start;      // a lane counts its null collisions in empty space from here and is tested from VP_EXIT_TRIP on:
                              // VP_EXIT_TRIP - K (VP_EXIT_K); far below zero when the test is off
    unsigned exit_nbytes;     // local-majorant estimators: how many distinct bytes occur as maxima in the bound table (1..4, else off) ...
    unsigned exit_bytes;      // ... and the bytes, packed: every majorant a segment in empty cells can have
    // Look-ahead batches (render_kernel's staged frames) can be told to stop: the host writes the number of the newest batch of this
    // slot that is no longer wanted into *cancel (numbers only grow: nothing is ever re-armed); a batch whose number is not above it
    // hands out no more samples -- render_k asks at every chunk it takes, the approach kernels when they start.  Null = never cancelled.
    unsigned* cancel;
    unsigned  batch_id;
    // Samples that are per-pixel CONSTANTS of a launch (the box-missing pixels always; the light class where a null collision in empty
    // space is neutral: miss_fill_k) are staged ONCE, in the row of the launch's first frame: slots from const_from on.  reduce_stage_k
    // adds such a sample nframes times, in the order it would add nframes staged copies: the same bits, without 2 x 16 bytes of
    // traffic per sample for 88 % of BASELINE config 2's samples.  (Fields appended: the offsets render_k reads do not move.)
    unsigned      const_from;    // first slot (of the rank's pixel list) whose sample is a constant of the launch; >= nslots: none
    const float4* stage_const;   // the staging row that holds the constants (the first frame of the batch the frames come from)
    const unsigned* bound_codes;   // the compact brick table (above): 2-bit codes ...
    unsigned        bound_pal[2];  // ... and the palette
    // decomposition estimator, uchar bound table: per pixel slot of the general class the chain of restart segments of its camera
    // ray through certified-empty cells (approach_segments_k: segment_table_records() float4 per slot); null: approach_local_k sets
    // every segment up per sample
    const float4* seg_table;
};

// lds_form: how the decomposition estimator reads its brick table -- 0 global memory, 1 the 16-bit table through LDS (512-thread
// workgroups), 2 2-bit codes into a four-entry palette through LDS (LaunchDev::bound_codes; 256-thread workgroups, plain occupancy)
void launch_render(const SceneDev& S, const LaunchDev& L, int est, int rng, bool quant, bool count, int lds_form, bool mis, int trk,
                   int blocks, hipStream_t st);
// the light pixel class (spectral tracking): pixels whose camera ray meets empty cells only
void launch_render_light(const SceneDev& S, const LaunchDev& L, int est, int rng, bool quant, bool count, int blocks, hipStream_t st);
void launch_miss_fill(const SceneDev& S, const LaunchDev& L, bool local_estimator, hipStream_t st);
// the camera rays' free flights through certified-empty cells, one thread per sample of the launch (approach_k); rng: RNG_PHILOX*
void launch_approach(const SceneDev& S, const LaunchDev& L, int est, int rng, bool quant, hipStream_t st);
// the per-pixel segment table of the decomposition estimator's approach walk (approach_segments_k): float4 per slot, and its builder
unsigned segment_table_records(void);
void launch_segment_table(const SceneDev& S, unsigned width, unsigned height, const float4* crawl, const unsigned* pixels, unsigned nslots, float4* seg,
                          hipStream_t st);
// throughput of an unscattered global-majorant path after n null collisions with density +0, n = 0..count-1 (thr_table_k)
void launch_thr_table(const ParamDev& P, float* table, unsigned count, hipStream_t st);
// mask[8]: the bytes that occur as a maximum in a uchar bound table; flag[0] (preset to 1) is cleared unless a null collision in
// empty space leaves a throughput of 1 unchanged for every sigma_t' a light path can meet (light_identity_k)
void launch_bound_bytes(const unsigned char* bounds, size_t nbricks, unsigned* mask, hipStream_t st);
void launch_light_identity(const ParamDev& P, bool local, const unsigned* mask, unsigned* flag, hipStream_t st);
// the rank's pixel lists, class by class, on the GPU (pixlist_*_k): d_row_start[tiles_y + 1] = first owned tile of each tile row,
// d_block_counts[3 * pixel_list_blocks(ntiles)] scratch, d_totals[3] = pixels per class (general, light, box-missing)
inline unsigned pixel_list_blocks(unsigned ntiles) { return (unsigned)(((size_t)ntiles * 64 + 1023) / 1024); }
void launch_pixel_lists(unsigned width, unsigned height, unsigned rank, unsigned world, unsigned ntiles, const unsigned* d_row_start,
                        const float4* table, unsigned* d_block_counts, unsigned* d_totals, unsigned* d_out, hipStream_t st);
void launch_env_tables(const float4* env, int w, int h, float* lum, float* row_sum, float* cdf_x, float* cdf_y, float* pdfnorm_alt,
                       hipStream_t st);
void launch_crawl_table(const SceneDev& S, bool quant, unsigned width, unsigned height, bool control_draw, const unsigned char* danger, float4* table,
                        hipStream_t st);
// marked (may be null): a device counter that receives the number of cells marked (a non-empty cell in the 3x3x3 neighbourhood)
void launch_danger(const SceneDev& S, bool quant, unsigned char* out, unsigned long long* marked, hipStream_t st);
float sun_clip_step(const SceneDev& S);  // the table's distance unit: a quarter of the smallest cell edge
void launch_sun_clip(const SceneDev& S, const unsigned char* danger, float ds, unsigned short* out, hipStream_t st);
void launch_empty_table(const SceneDev& S, unsigned width, unsigned height, const unsigned char* danger, float4* table, hipStream_t st);
// the direction table of the exit flights from the danger volume: planes = 3 * nx*ny*nz bytes; one small kernel per slice and direction
void launch_exit_table(const unsigned char* danger, unsigned char* planes, int nx, int ny, int nz, hipStream_t st);
void launch_reduce(const LaunchDev& L, hipStream_t st);
// bricks: cells in 4x4x4 bricks (vp_device.h cell_index); the buffer then holds ceil(n/4)^3 * 64 cells
void launch_pack_u8(const unsigned char* vol, uint2* cells, int nx, int ny, int nz, bool bricks, hipStream_t st);
void launch_pack_f32(const float* vol, float* cells, int nx, int ny, int nz, bool bricks, hipStream_t st);
// lds: uchar volumes through opacity_lds_k (the density grid staged through LDS tile by tile), else opacity_k: the same bits
void launch_opacity(const SceneDev& S, bool quant, bool lds, const float dir[3], float* out, hipStream_t st);
void launch_build_bounds(const void* d_vol, bool quant, void* d_out, void* d_tmp_a, void* d_tmp_b, int nx, int ny, int nz, int radius, int brick,
                         hipStream_t st);
void launch_julia(unsigned char* grid, int n, hipStream_t st);
void launch_cloud(float* grid, int n, unsigned seed, hipStream_t st);
void launch_scale(float4* dst, const float4* src, int size, float s, hipStream_t st);
void launch_gamma(float4* dst, const float4* src, int size, float s, float inv_gamma, hipStream_t st);
void launch_accumulate(float4* dst, const float4* src, size_t n, hipStream_t st);
void launch_test_hg(const float* g, const float* r0, const float* r1, const float* nrm, const float* cosq, float* dir, float* ev, int n, hipStream_t st);
void launch_test_box(const SceneDev& S, const float* o, const float* d, int* hit, float* tn, float* tf, int n, hipStream_t st);
void launch_test_env(const SceneDev& S, const float* d, float* out, int n, hipStream_t st);
void launch_test_math(int which, const float* in, float* out, int n, hipStream_t st);
void launch_test_rng(int mode, unsigned x, unsigned y, unsigned f, unsigned k0, unsigned k1, int n, float* out, hipStream_t st);
void launch_test_density(const SceneDev& S, bool quant, const float* pos, float* out, int n, hipStream_t st);
}  // namespace vp
