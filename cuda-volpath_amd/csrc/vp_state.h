// vp_state.h -- what the host-side translation units of libvolpath_hip.so share: the per-context State, the error helpers and the
// internal functions of vp_context.cpp (device, volume, environment, contexts, Part 1 and the setters of include/volpath.h),
// vp_tables.cpp (the tables a launch reads: optical depth, per-pixel, sun, exit, pixel lists), vp_render.cpp (do_render: one
// staged launch; counters and timing) and vp_lookahead.cpp (render_kernel's frame look-ahead: frozen since round 4, see there).
// Round 5 split of the former vp_api.cpp (2 200 lines), no behaviour change.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/volpath.h"
#include "vp_kernels.h"

// internal linkage in spirit: hidden from the library's dynamic symbol table, which carries include/volpath.h and nothing else
namespace vph __attribute__((visibility("hidden")))
{
using namespace vp;

// host.cpp:1098-1101: diffusion_iters = ceil(search_radius / (2.0f / width))
inline int bound_radius(int nx, float search_radius)
{
    float cell_size = 2.0f / (float)nx;
    return (int)std::ceil(search_radius / cell_size);
}

struct State
{
    int         device      = 0;
    bool        dev_ready   = false;
    int         num_cu      = 0;
    hipStream_t own_stream  = nullptr;
    hipStream_t stream      = nullptr;
    SceneDev    S           = {};
    bool        quant       = true;
    bool        have_volume = false, have_env = false, have_sun = false, have_cam = false;
    void*       d_cells     = nullptr;
    void*       d_bounds    = nullptr;
    float*      d_opacity   = nullptr;
    float*      d_opacity_cells = nullptr;   // the same table as 8-float neighbourhood cells: what the integrator reads
    float4*     d_env       = nullptr;
    int         env_w = 0, env_h = 0;
    // frame look-ahead of render_kernel (see serve_frame): what the staged frames were rendered with
    unsigned long long epoch = 0;     // bumped whenever device CONTENT changes behind unchanged pointers
    int         la_max      = 256;    // most frames rendered ahead per launch; <= 1 switches the look-ahead off
    int         la_floor    = 64;     // batches up to this size are used from the start of a run; larger ones once the run is twice as long (VP_LOOKAHEAD_FLOOR)
    bool        la_habit    = false;  // the caller has been served a staged frame: it asks for consecutive frames
    bool        la_spec_unserved = false;   // a speculative batch is out and none of its frames has been asked for yet (the habit decays)
    bool        la_speculate = true;  // ... then the first batch of a run is queued beside the run's first frame (VP_LOOKAHEAD_NO_SPECULATION=1: behind it)
    int         la_div      = 2;      // ... as long as la_div times the batch (VP_LOOKAHEAD_DIV)
    int         la_run_first = 0;     // first frame of the current run of consecutive render_kernel calls
    int         la_ramp_from = 32;    // size of the first batch of a run (VP_LOOKAHEAD_RAMP_FROM)
    int         la_overlap_from = 2;  // a batch of at least this many frames has its successor queued behind it on the other slot (VP_LOOKAHEAD_OVERLAP_FROM)
    struct LaSlot  // one staged batch of frames, rendered on its own stream so that two batches overlap
    {
        float4*     buf = nullptr;
        size_t      bytes = 0;
        hipStream_t stream = nullptr;
        hipEvent_t  done = nullptr;   // recorded after the batch's render
        bool        valid = false;
        unsigned    const_from = 0;   // LaunchDev::const_from of the batch (slots whose samples are per-pixel constants, staged once in the batch's first row)
        unsigned    launched_seq = 0; // the number of the batch whose completion `done` stands for (the slot's batch_seq while no launch is under way)
        unsigned    cancel_seq = 0;   // the number of the batch of this slot that was last told to stop (la_cancel_running)
        bool        touched = false;  // a frame of this batch was handed to the caller while the batch was still running: its add-kernel
                                      // waits for the WHOLE batch, which must then run to its end (la_quiesce does not cancel it)
        int         first = 0, count = 0;
        std::vector<unsigned char> key;
    } la[2];
    // la_quiesce tells batches in flight to stop handing out samples: a stream of the HIGHEST priority -- such streams have hardware
    // queues of their own; on an ordinary stream the write shared a queue with the very batch it was to stop and arrived when that
    // batch had finished (3-22 ms later: profiles/experiments/r04_lookahead_cancel.txt)
    hipStream_t ctrl_stream = nullptr;
    unsigned*   d_cancel    = nullptr; // [3] per render target: the newest batch number of the slot that is cancelled (LaunchDev::cancel)
    unsigned    batch_seq[3] = {0, 0, 0};   // number of the last batch queued on each target
    bool        la_cancel   = true;    // VP_NO_LA_CANCEL=1: batches in flight always run to their end
    int         la_prev_n   = 0;      // batch size of the last miss
    int         la_last     = -2;     // frame index of the last render_kernel call
    std::vector<unsigned char> la_key;  // render state of the staged frames / of the last call
    // active environment sampling (!PASSIVE_ENVMAP): CDF tables, built on demand
    int         trk         = 0;      // VP_TRACK_*: spectral (shipped), scalar, multi-channel
    bool        env_mis     = false;
    bool        env_tables  = false;  // tables match the current envmap
    float*      d_env_cdf_x = nullptr;
    float*      d_env_cdf_y = nullptr;
    bool        linear      = false;  // kernel.cu:351: point filtering until set_texture_filter_mode(true)
    int         brick_next  = 1;
    int         brick       = 1;
    int         radius      = 0;
    int         est         = VP_EST_DECOMP;
    int         rng         = VP_RNG_SAMPLERH;
    unsigned    key0 = 0, key1 = 0;
    unsigned    rank = 0, world = 1;
    float4*     d_stage       = nullptr;
    size_t      stage_bytes   = 0;
    unsigned*   d_queue       = nullptr;
    unsigned long long* d_counters = nullptr;
    bool        count       = false;
    bool        use_opacity_cells = true; // the packed copy of the optical-depth table lives on the device (VP_NO_OPACITY_CELLS=1: in pinned host memory, the out-of-memory fall-back)
    bool        opacity_cells_on_host = false;
    void*       h_opacity_cells = nullptr;   // its host address (hipHostFree wants that one)
    bool        opacity_lds = true;       // precompute_opacity stages the density grid through LDS (opacity_lds_k; VP_NO_OPACITY_LDS=1: opacity_k)
    bool        wait_lanes_set = false;   // VP_WAIT_LANES given: no per-kernel default
    unsigned    wait_lanes  = VP_WAIT_LANES, wait_iters = VP_WAIT_ITERS, setup_lanes = VP_SETUP_LANES, end_lanes = VP_END_LANES, light_wait_iters = 0;  // 0 = by estimator
    unsigned    blocks_per_cu = 8;  // 256-thread workgroups per CU launched for a kernel that runs alone: as many as can be resident (seven of
                                    // the achromatic global-majorant kernel, six of the other plain ones, five of the chromatic local ones; a
                                    // workgroup too many starts when the queues are empty and ends at once)
    unsigned    chunk_fshift = 0;         // VP_CHUNK_FRAMES_LOG2: a chunk = (VP_CHUNK >> k) pixels x (1 << k) frames (general class), k <= log2(VP_CHUNK)
    bool        use_lds_bounds = true;
    bool        use_lds_compact = true;   // a brick table of at most four distinct pairs goes through LDS as 2-bit codes (VP_NO_LDS_COMPACT=1: as 16-bit pairs)
    bool        lds_compact_chromatic = true;    // ... for chromatic media too (VP_LDS_COMPACT_CHROMATIC=0: not; since their kernel fits six waves
                                                 // -- 79 registers, intersect_box axis by axis -- the codes beat the 16-bit table and its helper: c4s +4.4 %)
    bool        lds_pairs = false;               // timed launches of the counter-based streams whose table cannot go as codes read it from global memory
                                                 // (six waves) rather than as 16-bit pairs through LDS (four and a helper): c4f +1.6 %.  VP_LDS_PAIRS=1: the pairs
    unsigned*   d_bound_codes = nullptr;  // ... the codes (built with the volume), and the palette
    unsigned    bound_pal[2] = {0, 0};
    bool        bound_codes_ok = false;
    bool        lds_helper  = true;       // one plain workgroup per CU beside the LDS-table kernel (VP_NO_LDS_HELPER=1)
    int         cell_bricks = 0;          // packed cells in 4x4x4 bricks (VP_CELL_BRICKS=1; an A/B knob, see do_init_volume_)
    // where the restart crawl in front of the volume ends, per pixel (crawl_table_k); rebuilt when what it depends on changes
    bool        use_crawl_table = true;
    bool        use_empty_table = true;   // global-majorant estimator: certified-empty distances of the camera rays
    bool        use_light   = true;       // ... and the light kernel for pixels whose ray meets empty cells only
    bool        use_light_local = true;   // ... also for the local-majorant estimators
    int         debug_only_class = -1;    // VP_DEBUG_ONLY_CLASS = 0 / 1: launch only the general / only the light kernel (INCOMPLETE images;
                                          // for the block tallies of one kernel)
    // the pixels this context owns (those of its tiles), class by class: [general..., light...], each y << 16 | x
    unsigned*   d_tiles     = nullptr;
    size_t      tiles_cap   = 0;
    unsigned    n_general   = 0, n_light = 0, n_miss = 0;
    std::vector<unsigned char> tiles_key;
    unsigned*   d_tile_rows = nullptr;    // first owned tile of each tile row (pixlist kernels)
    unsigned*   d_tile_scratch = nullptr; // per-block class counts + the three totals
    std::vector<unsigned char> tiles_shape_key;
    // the light kernel runs beside the general one on a stream of its own (ALU-bound waves fill the issue slots the general
    // kernel's waves leave while they wait for cells): one auxiliary stream and two events per launch target
    bool        light_overlap = true;
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t  aux_ev[3][2]  = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    // resident 256-thread workgroups per CU while both kernels run (0 = the measured defaults of profiles/r02_light_overlap.txt:
    // 3 + 4 for the global-majorant estimator, 5 + 2 for the local-majorant ones)
    unsigned    general_blocks_per_cu = 0, light_blocks_per_cu = 0;
    unsigned char* d_danger = nullptr;    // per cell: a non-empty cell within its 3x3x3 neighbourhood (danger_k)
    float4*     d_crawl     = nullptr;
    size_t      crawl_bytes = 0;
    std::vector<unsigned char> crawl_key;
    // counter-based streams: where a sun shadow ray has only empty cells left (sun_clip_k), per cell; rebuilt when the volume, the
    // box or the sun direction changes
    // Tables that pay in EMPTY space -- the sun table (a shadow ray ends where only empty cells are left) and the decomposition
    // estimator's approach walk -- are switched off for volumes with little of it: where more than dense_fraction of the cells have a
    // non-empty cell in their 3x3x3 neighbourhood (danger_k counts them).  Measured (profiles/experiments/r05_dense_volumes.txt): the
    // Julia sets (7 % marked) gain 15-27 % from the sun table, the frame-filling cloud (54 %) LOSES 1 % to each of the two.  Results
    // never depend on it.  VP_DENSE_PERCENT overrides the threshold (101: never dense).
    float       marked_fraction = 0.0f, dense_fraction = 0.40f;
    bool        use_sun_clip = true;
    unsigned short* d_sunclip = nullptr;
    float       sunclip_ds  = 0.0f;
    std::vector<unsigned char> sunclip_key;
    // the samples of the light class are per-pixel constants when a null collision in empty space leaves a throughput of 1
    // exactly 1 (light_identity_k): decided per (medium, estimator, volume), then miss_fill_k writes them
    bool        use_light_const = true;
    uint2*      d_appr_aux[3] = {nullptr, nullptr, nullptr};   // per render target (caller's stream, two look-ahead slots): LaunchDev::approach_aux
    size_t      appr_aux_bytes[3] = {0, 0, 0};
    int         last_approach = 0;            // vp_last_approach_mode
    int         last_approach_table = 0;      // vp_last_approach_table
    int         last_light_const = 0;         // vp_last_light_const
    int         last_lds_form = 0;            // vp_last_lds_form
    unsigned    la_launched = 0, la_cancelled = 0;   // vp_lookahead_stats
    bool        use_const_rows = true;        // VP_NO_CONST_ROWS=1: per-pixel constants are staged for every frame, as before round 4's end
    unsigned    last_const_from = 0;          // LaunchDev::const_from of the last staged launch (a look-ahead slot keeps it for its add-kernels)
    bool        use_approach_table = true;    // ... with the restart segments of a pixel's camera ray tabulated per pixel (VP_NO_APPROACH_TABLE=1: set up per sample)
    float4*     d_seg = nullptr;              // the table (approach_segments_k), its size and what it was built for
    size_t      seg_bytes = 0;
    std::vector<unsigned char> seg_key;
    bool        use_approach_local = true;    // ... and approach_local_k ahead of the decomposition estimator (VP_NO_APPROACH_LOCAL=1: off)
    bool        use_approach = true;          // approach_k ahead of the global-majorant integrator (VP_NO_APPROACH=1: off)
    unsigned    approach_fshift_max = 6;      // a wave of the approach kernels = one pixel x 2^6 frames (VP_APPROACH_FRAMES_LOG2: 0 = 64 pixels of one frame)
    unsigned    approach_steps = 1u << 20;    // its step cap per sample (VP_APPROACH_STEPS)
    unsigned*   d_light_flag = nullptr;   // [0] the flag, [1..8] the bytes that occur as maxima in the bound table
    bool        bound_mask_valid = false;
    unsigned    h_bound_mask[8] = {};     // ... read back (ensure_bound_mask)
    // exit flights (render_k): per cell and class of directions, is every cell such a ray can meet empty?  Built with the volume
    bool        use_exit    = true;       // VP_NO_EXIT=1: every path walks to the box exit
    bool        exit_local  = false;      // VP_EXIT_LOCAL=1 / vp_set_exit_flights(2): also for the local-majorant estimators, whatever the stream
    bool        exit_local_auto = true;   // until either is used: for the local-majorant estimators on the counter-based streams (round 5: C3 +3.9 %,
                                          // c4s +1.6 %, c3ref +1 % with six-wave kernels; sampler.h +-0.5 %: profiles/experiments/r05_exit_local.txt)
    unsigned    exit_k      = 8;          // null collisions in empty space before a lane asks for the test (VP_EXIT_K)
    unsigned char* d_exit   = nullptr;
    float       light_key[7] = {};
    unsigned long long light_epoch = ~0ull;
    bool        light_const = false;
    float*      d_thr       = nullptr;    // throughput after n null collisions in empty space (light kernel, global majorant)
    float       thr_key[5]  = {};
    bool        thr_valid   = false;
    unsigned    thr_entries = 4096;       // paths with more null collisions run the recurrence on from the last entry (VP_THR_TABLE)
    // launch timing: a ring of the last kMaxPendingEvents launches; older pairs are folded into the running sum
    std::deque<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> event_pool;
    double      timed_ms    = 0.0;   // folded launches
    int         timed_n     = 0;     // launches counted (folded or pending or dropped)
    // the same per pixel class: event pairs around the general kernel, the light kernel and the box-missing fill of each launch
    struct ClassEv { int cls; hipEvent_t a, b; };
    std::deque<ClassEv> class_events;
    double      class_ms[3] = {0.0, 0.0, 0.0};
    // Per-sample staging per launch.  A launch ends with a tail in which only the deepest paths are still running (about
    // 14 ms at 800x600 whatever the launch size), so launches should be long: 128 frames per launch (1 GiB) lose 9 % to
    // tails, 1024 frames (8 GB) 1 %.  288 GB of HBM make that cheap; the cap is also held to a quarter of the free memory
    // at allocation time.  VP_STAGE_MB overrides.
    size_t      max_stage_bytes = (size_t)16 << 30;
    float       inv_model[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::string err;
};
// The default context serves every thread that never called vp_ctx_set_current: the reference host (one scene, one device,
// kernel.cu's file-scope statics) binds Part 1 and never sees a context.
// (cur() is a function of vp_context.cpp, not an inline over `extern thread_local`: a thread_local that is only DECLARED in a translation
// unit is reached through its init wrapper, whose weak, hidden, undefined init symbol resolves to the library's load address in a shared
// object -- the first call from another file jumped there.  Found by the GPU suite the moment the file was split.)
State& cur();
#define G vph::cur()
constexpr size_t kMaxPendingEvents = 64;

constexpr size_t kQueueWords = VP_NQUEUES * VP_QUEUE_STRIDE;  // queue heads of one launch
constexpr size_t kCounterWords = 76;  // 6 work counters, 6 loop statistics, 15 x (wave, lane) block tallies from word 16, 3 x 8 histogram buckets from word 48, the control-component tally at 72, danger_k's marked-cell count in the last word

int fail(int code, const char* fmt, ...);
[[noreturn]] void die(const char* what);
#define HIPCHK(expr)                                                                            \
    do                                                                                          \
    {                                                                                           \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
        {                                                                                       \
            if (e_ == hipErrorOutOfMemory) (void)hipGetLastError(); /* not sticky: later calls may succeed */ \
            return fail(e_ == hipErrorOutOfMemory ? VP_E_NOMEM : VP_E_NODEVICE, "%s -> %s", #expr, hipGetErrorString(e_)); \
        }                                                                                       \
    } while (0)

// ---- vp_context.cpp
int        ensure_device();
hipError_t create_internal_stream(hipStream_t* st);
int        free_volume();
int        do_init_volume(const void* h_volume, vp_extent ext, bool quantized, const vp_float3* bmin, const vp_float3* bmax);
int        do_envmap(const vp_float4* data, int w, int h);
int        build_env_tables();
hipEvent_t get_event();   // an event from the pool, or a new one; nullptr if the runtime cannot create one (the launch then goes untimed)
inline void put_event(hipEvent_t e) { if (e) G.event_pool.push_back(e); }
// ---- vp_tables.cpp
// the shard of this context (include/volpath.h vp_tile_owner): its 8x8 tiles and the pixels of the image they hold
struct Shard { unsigned tiles_x, tiles_y, owned; size_t per_frame; };  // owned = tiles, per_frame = pixels = samples per frame
Shard  shard_of(const Param* p);
size_t stage_frames_cap(size_t per_frame, size_t have_bytes);
int    do_opacity(const float* dir);
int    ensure_crawl_table(const Param* p, const float4** out);
int    ensure_sun_clip(const unsigned short** out, float* ds);
int    ensure_light_const(const Param* p, bool* out);
int    ensure_light_identity(const Param* p, bool* out);
int    ensure_bound_mask();
int    exit_flights(LaunchDev& L);
int    ensure_thr_table(const Param* p, const float** out);
int    ensure_pixel_lists(const Param* p, const float4* table, const Shard& sh);
int    ensure_segment_table(const Param* p, const float4* crawl, const float4** out);
// ---- vp_render.cpp
// where a launch stages its samples: the caller's stream and buffers, or a look-ahead slot's
struct Target { hipStream_t stream; float4** stage; size_t* stage_bytes; unsigned* queue; int index; };
int  do_render(vp_float4* d_out, int first, int nframes, const Param* p, bool stage_only = false, const Target* tgt = nullptr);
void trim_events();
// ---- vp_lookahead.cpp
void render_key(const Param* p, std::vector<unsigned char>& key);
bool la_cancel_running();
int  la_quiesce();
int  la_render_slot(int si, vp_float4* d_out, int first, int n, const Param* p, const std::vector<unsigned char>& key);
int  la_limit(int first, int n, size_t per_frame, size_t have_bytes);
int  serve_frame(vp_float4* d_out, int frame, const Param* p);
}  // namespace vph
