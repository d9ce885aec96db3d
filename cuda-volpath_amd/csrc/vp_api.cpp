// vp_api.cpp -- the C ABI of libvolpath_hip.so (include/volpath.h): scene state in HBM and the
// launches.  Part 1 mirrors the reference's kernel-TU entry points (kernel.cu:354-451, :526-553,
// :1072-1283, :2320-2370); Part 2 is the additive interface.
//
// HBM layout (DESIGN.md "Data layout"): the density volume is stored as one 8-byte (uchar) or
// 32-byte (float) cell per voxel holding that voxel's clamped 2x2x2 texel neighbourhood, so a
// trilinear fetch is ONE aligned load; bounds are (max,min) pairs per brick; opacity is a plain
// float N^3 array; the environment is float4 rows.  No textures, no CPU fallback.
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

namespace
{
using namespace vp;

// host.cpp:1098-1101: diffusion_iters = ceil(search_radius / (2.0f / width))
int bound_radius(int nx, float search_radius)
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
    bool        use_opacity_cells = true; // the integrator reads the optical-depth table from neighbourhood-packed cells (best effort: 8x the table)
    bool        opacity_lds = true;       // precompute_opacity stages the density grid through LDS (opacity_lds_k; VP_NO_OPACITY_LDS=1: opacity_k)
    bool        wait_lanes_set = false;   // VP_WAIT_LANES given: no per-kernel default
    unsigned    wait_lanes  = VP_WAIT_LANES, wait_iters = VP_WAIT_ITERS, setup_lanes = VP_SETUP_LANES, end_lanes = VP_END_LANES, light_wait_iters = 0;  // 0 = by estimator
    unsigned    blocks_per_cu = 8;  // 256-thread workgroups per CU launched for a kernel that runs alone: as many as can be resident (seven of
                                    // the achromatic global-majorant kernel, six of the other plain ones, five of the chromatic local ones; a
                                    // workgroup too many starts when the queues are empty and ends at once)
    unsigned    chunk_fshift = 0;         // VP_CHUNK_FRAMES_LOG2: a chunk = (256 >> k) pixels x (1 << k) frames (general class)
    bool        use_lds_bounds = true;
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
    int         last_light_const = 0;         // vp_last_light_const
    unsigned    la_launched = 0, la_cancelled = 0;   // vp_lookahead_stats
    bool        use_const_rows = true;        // VP_NO_CONST_ROWS=1: per-pixel constants are staged for every frame, as before round 4's end
    unsigned    last_const_from = 0;          // LaunchDev::const_from of the last staged launch (a look-ahead slot keeps it for its add-kernels)
    bool        use_approach_local = true;    // ... and approach_local_k ahead of the decomposition estimator (VP_NO_APPROACH_LOCAL=1: off)
    bool        use_approach = true;          // approach_k ahead of the global-majorant integrator (VP_NO_APPROACH=1: off)
    unsigned    approach_fshift_max = 6;      // a wave of the approach kernels = one pixel x 2^6 frames (VP_APPROACH_FRAMES_LOG2: 0 = 64 pixels of one frame)
    unsigned    approach_steps = 1u << 20;    // its step cap per sample (VP_APPROACH_STEPS)
    unsigned*   d_light_flag = nullptr;   // [0] the flag, [1..8] the bytes that occur as maxima in the bound table
    bool        bound_mask_valid = false;
    unsigned    h_bound_mask[8] = {};     // ... read back (ensure_bound_mask)
    // exit flights (render_k): per cell and class of directions, is every cell such a ray can meet empty?  Built with the volume
    bool        use_exit    = true;       // VP_NO_EXIT=1: every path walks to the box exit
    bool        exit_local  = false;      // VP_EXIT_LOCAL=1: also for the local-majorant estimators (measured: nothing to gain there)
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
State                g_default;
thread_local State*  t_current = nullptr;
inline State& cur() { return t_current ? *t_current : g_default; }
#define G cur()
constexpr size_t kMaxPendingEvents = 64;

constexpr size_t kQueueWords = VP_NQUEUES * VP_QUEUE_STRIDE;  // queue heads of one launch
constexpr size_t kCounterWords = 72;  // 6 work counters, 6 loop statistics, 15 x (wave, lane) block tallies from word 16, 3 x 8 histogram buckets from word 48

int fail(int code, const char* fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    G.err = buf;
    return code;
}
[[noreturn]] void die(const char* what)
{
    // the reference's failure mode: checkCudaErrors -> fprintf + exit(EXIT_FAILURE) (helper_cuda.h:566-579)
    fprintf(stderr, "volpath_hip: %s: %s\n", what, G.err.c_str());
    exit(EXIT_FAILURE);
}
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

int ensure_device()
{
    if (G.dev_ready)
    {
        // another context of this process may have left a different device current on this thread
        HIPCHK(hipSetDevice(G.device));
        return VP_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(VP_E_NODEVICE, "no HIP device visible");
    if (G.device >= n) return fail(VP_E_NODEVICE, "device %d out of range (%d visible)", G.device, n);
    HIPCHK(hipSetDevice(G.device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, G.device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VP_E_NODEVICE, "device %d is %s; this library carries gfx950 code only", G.device, prop.gcnArchName);
    G.num_cu = prop.multiProcessorCount;
    HIPCHK(hipStreamCreateWithFlags(&G.own_stream, hipStreamNonBlocking));
    if (!G.stream) G.stream = G.own_stream;
    HIPCHK(hipMalloc((void**)&G.d_queue, 6 * kQueueWords * sizeof(unsigned)));  // (caller's stream + two look-ahead slots) x two tile classes
    HIPCHK(hipMalloc((void**)&G.d_counters, kCounterWords * sizeof(unsigned long long)));
    HIPCHK(hipMemset(G.d_counters, 0, kCounterWords * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void**)&G.d_cancel, 3 * sizeof(unsigned)));
    HIPCHK(hipMemset(G.d_cancel, 0, 3 * sizeof(unsigned)));
    G.S.sun_cos = 94.0f / sqrtf(94.0f * 94.0f + 0.45f * 0.45f);                    // kernel.cu:1263
    G.S.cam_z   = (float)(-1.0f / tan((double)54.43f * 0.00872664626));             // kernel.cu:1981-1985
    // tuning knobs (performance only; results never depend on them).  Out-of-range or malformed values are ignored:
    // wait_lanes = 0 would end the tracking loop before its first step (a persistent kernel that never finishes),
    // blocks_per_cu = 0 is an empty grid, a negative VP_STAGE_MB a huge size_t.
    auto knob = [](const char* name, long lo, long hi, long& out) {
        const char* e = getenv(name);
        if (!e || !*e) return false;
        char* end = nullptr;
        long  v   = strtol(e, &end, 10);
        if (*end || v < lo || v > hi)
        {
            fprintf(stderr, "volpath_hip: ignoring %s=%s (allowed %ld..%ld)\n", name, e, lo, hi);
            return false;
        }
        out = v;
        return true;
    };
    long v;
    if (knob("VP_WAIT_LANES", 1, 64, v)) { G.wait_lanes = (unsigned)v; G.wait_lanes_set = true; }
    if (knob("VP_WAIT_ITERS", VP_STEPS_PER_PASS, 1 << 20, v)) G.wait_iters = (unsigned)v;
    if (knob("VP_SETUP_LANES", 1, 64, v)) G.setup_lanes = (unsigned)v;
    if (knob("VP_END_LANES", 1, 64, v)) G.end_lanes = (unsigned)v;
    if (knob("VP_LIGHT_WAIT_ITERS", VP_STEPS_PER_PASS, 1 << 20, v)) G.light_wait_iters = (unsigned)v;
    if (knob("VP_STAGE_MB", 1, 256 << 10, v)) G.max_stage_bytes = (size_t)v << 20;
    if (knob("VP_BLOCKS_PER_CU", 1, 8, v)) G.blocks_per_cu = (unsigned)v;
    if (knob("VP_CHUNK_FRAMES_LOG2", 0, 8, v)) G.chunk_fshift = (unsigned)v;
    if (knob("VP_NO_LDS_BOUNDS", 0, 1, v)) G.use_lds_bounds = v == 0;
    if (knob("VP_NO_LDS_HELPER", 0, 1, v)) G.lds_helper = v == 0;
    if (knob("VP_CELL_BRICKS", 0, 1, v)) G.cell_bricks = (int)v;
    if (knob("VP_NO_CRAWL_TABLE", 0, 1, v)) G.use_crawl_table = v == 0;
    if (knob("VP_NO_EMPTY_TABLE", 0, 1, v)) G.use_empty_table = v == 0;
    if (knob("VP_NO_SUN_CLIP", 0, 1, v)) G.use_sun_clip = v == 0;
    if (knob("VP_NO_OPACITY_LDS", 0, 1, v)) G.opacity_lds = v == 0;
    if (knob("VP_NO_OPACITY_CELLS", 0, 1, v)) G.use_opacity_cells = v == 0;
    if (knob("VP_NO_LIGHT_CONST", 0, 1, v)) G.use_light_const = v == 0;
    if (knob("VP_NO_CONST_ROWS", 0, 1, v)) G.use_const_rows = v == 0;
    if (knob("VP_NO_APPROACH", 0, 1, v)) G.use_approach = v == 0;
    if (knob("VP_NO_APPROACH_LOCAL", 0, 1, v)) G.use_approach_local = v == 0;
    if (knob("VP_APPROACH_FRAMES_LOG2", 0, 6, v)) G.approach_fshift_max = (unsigned)v;
    if (knob("VP_APPROACH_STEPS", 0, 1 << 30, v)) G.approach_steps = (unsigned)v;
    if (knob("VP_NO_LIGHT", 0, 1, v)) G.use_light = v == 0;
    if (knob("VP_NO_LIGHT_OVERLAP", 0, 1, v)) G.light_overlap = v == 0;
    if (knob("VP_NO_LIGHT_LOCAL", 0, 1, v)) G.use_light_local = v == 0;
    if (knob("VP_DEBUG_ONLY_CLASS", 0, 1, v)) G.debug_only_class = (int)v;
    if (knob("VP_GENERAL_BLOCKS_PER_CU", 1, 8, v)) G.general_blocks_per_cu = (unsigned)v;
    if (knob("VP_LIGHT_BLOCKS_PER_CU", 1, 8, v)) G.light_blocks_per_cu = (unsigned)v;
    if (knob("VP_THR_TABLE", 2, 1 << 20, v)) G.thr_entries = (unsigned)v;
    if (knob("VP_LOOKAHEAD", 0, 4096, v)) G.la_max = (int)v;
    if (knob("VP_LOOKAHEAD_OVERLAP_FROM", 1, 4096, v)) G.la_overlap_from = (int)v;
    if (knob("VP_LOOKAHEAD_RAMP_FROM", 2, 4096, v)) G.la_ramp_from = (int)v;
    if (knob("VP_LOOKAHEAD_FLOOR", 1, 4096, v)) G.la_floor = (int)v;
    if (knob("VP_LOOKAHEAD_DIV", 1, 16, v)) G.la_div = (int)v;
    if (knob("VP_LOOKAHEAD_NO_SPECULATION", 0, 1, v)) G.la_speculate = v == 0;
    if (knob("VP_NO_LA_CANCEL", 0, 1, v)) G.la_cancel = v == 0;
    if (knob("VP_NO_EXIT", 0, 1, v)) G.use_exit = v == 0;
    if (knob("VP_EXIT_LOCAL", 0, 1, v)) G.exit_local = v != 0;
    if (knob("VP_EXIT_K", 1, VP_EXIT_TRIP, v)) G.exit_k = (unsigned)v;
    G.dev_ready = true;
    return VP_OK;
}

int la_quiesce();
// The library's internal streams -- look-ahead slots, the light kernel's / helper workgroups' side streams -- have the LOWEST priority:
// streams of one priority share a pool of four hardware queues, and among a caller's ordinary streams an internal stream came to
// share a queue with the caller's (whose kernels wait for events of the other internal streams): what was meant to overlap ran in
// turn (host loop 2030 -> 1635 Msamples/s with one extra stream in the process).  What runs ahead or beside also SHOULD yield to
// what the caller asked for.
hipError_t create_internal_stream(hipStream_t* st)
{
    int lo = 0, hi = 0;   // (numerically greater = lower priority)
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, lo);
}

int free_volume()
{
    G.epoch++;  // staged look-ahead frames no longer describe this scene ...
    (void)la_quiesce();  // ... and batches in flight must not see device buffers change under them

    if (G.d_cells) HIPCHK(hipFree(G.d_cells));
    if (G.d_bounds) HIPCHK(hipFree(G.d_bounds));
    if (G.d_opacity) HIPCHK(hipFree(G.d_opacity));
    if (G.d_opacity_cells) HIPCHK(hipFree(G.d_opacity_cells));
    G.d_opacity_cells = nullptr; G.S.opacity_cells = nullptr;
    if (G.d_danger) HIPCHK(hipFree(G.d_danger));
    G.d_danger = nullptr;
    if (G.d_sunclip) HIPCHK(hipFree(G.d_sunclip));
    G.d_sunclip = nullptr; G.sunclip_key.clear();
    if (G.d_exit) HIPCHK(hipFree(G.d_exit));
    G.d_exit = nullptr;
    G.d_cells = G.d_bounds = nullptr;
    G.d_opacity   = nullptr;
    G.S.cells_u8  = nullptr;
    G.S.cells_f32 = nullptr;
    G.S.bounds_u8 = nullptr;
    G.S.bounds_f32 = nullptr;
    G.S.opacity    = nullptr;
    G.have_volume  = false;
    return VP_OK;
}

int do_init_volume_(const void* h_volume, vp_extent ext, bool quantized, const vp_float3* bmin, const vp_float3* bmax);
int do_init_volume(const void* h_volume, vp_extent ext, bool quantized, const vp_float3* bmin, const vp_float3* bmax)
{
    int rc = do_init_volume_(h_volume, ext, quantized, bmin, bmax);
    if (rc)
    {
        // a failed upload leaves no half-built scene behind (free_volume keeps the error text of the failure)
        std::string why = G.err;
        (void)free_volume();
        G.err = why;
    }
    return rc;
}
int do_init_volume_(const void* h_volume, vp_extent ext, bool quantized, const vp_float3* bmin, const vp_float3* bmax)
{
    G.epoch++;  // staged look-ahead frames no longer describe this scene ...
    (void)la_quiesce();  // ... and batches in flight must not see device buffers change under them

    int rc = ensure_device();
    if (rc) return rc;
    if (ext.width == 0 || ext.height == 0 || ext.depth == 0) return fail(VP_E_ARG, "empty volume extent");
    size_t n = ext.width * ext.height * ext.depth;
    if (n > ((size_t)1 << 32) - 1) return fail(VP_E_ARG, "volume of %zu voxels exceeds the 2^32 cell index", n);
    if (ext.width > 4096 || ext.height > 4096 || ext.depth > 4096) return fail(VP_E_ARG, "volume edge > 4096 voxels");
    HIPCHK(hipStreamSynchronize(G.stream));
    rc = free_volume();
    if (rc) return rc;
    const int nx = (int)ext.width, ny = (int)ext.height, nz = (int)ext.depth;
    SceneDev& S = G.S;
    S.nx = nx; S.ny = ny; S.nz = nz;
    if (bmin && bmax)
    {
        S.bmin[0] = bmin->x; S.bmin[1] = bmin->y; S.bmin[2] = bmin->z;
        S.bmax[0] = bmax->x; S.bmax[1] = bmax->y; S.bmax[2] = bmax->z;
    }
    else
    {
        // kernel.cu:373-378
        S.bmin[0] = -1.0f; S.bmin[1] = -(float)ny / (float)nx; S.bmin[2] = -(float)nz / (float)nx;
        S.bmax[0] = 1.0f;  S.bmax[1] = (float)ny / (float)nx;  S.bmax[2] = (float)nz / (float)nx;
    }
    for (int a = 0; a < 3; a++) S.linv[a] = 1.0f / (S.bmax[a] - S.bmin[a]);  // kernel.cu:313
    G.quant = quantized;
    // volume -> packed neighbourhood cells.  The three scratch buffers belong to a guard: every early return frees them.
    struct Scratch
    {
        void* p[3] = {nullptr, nullptr, nullptr};
        ~Scratch() { for (void* q : p) if (q) (void)hipFree(q); }
    } tmp;
    void*& d_raw = tmp.p[0];
    void*& d_ta  = tmp.p[1];
    void*& d_tb  = tmp.p[2];
    const size_t vbytes = n * (quantized ? 1 : 4);
    HIPCHK(hipMalloc(&d_raw, vbytes));
    HIPCHK(hipMemcpyAsync(d_raw, h_volume, vbytes, hipMemcpyHostToDevice, G.stream));
    // cell layout: x fastest (default), or 4x4x4 bricks of cells (VP_CELL_BRICKS=1).  Measured on every workload incl. the two
    // 512^3 ones whose cells (1.07 GB) are outside every cache (profiles/r03_cell_layout_ab.txt): fabric reads -3...-12 %, L2 hit
    // rate +1...+3 points, Msamples/s within 1 % (c3ref -2 %) -- the rays of a wave are too many and too incoherent for either order
    // to keep their lines in a 4 MiB L2.  The x-fastest order stays; the knob is kept for the A/B.
    const size_t ncells_bricks = (((size_t)nx + 3) / 4) * (((size_t)ny + 3) / 4) * (((size_t)nz + 3) / 4) * 64;
    const bool   bricks = G.cell_bricks > 0;
    const size_t ncells = bricks ? ncells_bricks : n;
    S.cell_bricks = bricks ? 1 : 0;
    HIPCHK(hipMalloc(&G.d_cells, ncells * (quantized ? 8 : 32)));
    if (bricks && ncells != n) HIPCHK(hipMemsetAsync(G.d_cells, 0, ncells * (quantized ? 8 : 32), G.stream));   // the padding of partial bricks
    if (quantized) launch_pack_u8((const unsigned char*)d_raw, (uint2*)G.d_cells, nx, ny, nz, bricks, G.stream);
    else launch_pack_f32((const float*)d_raw, (float*)G.d_cells, nx, ny, nz, bricks, G.stream);
    HIPCHK(hipGetLastError());
    // bound table: three separable max/min passes + brick reduction on the GPU (replaces host.cpp:1088-1267)
    G.brick  = G.brick_next;
    G.radius = bound_radius(nx, 0.05f /* search_radius kernel.cu:151 */) + (G.brick > 1 ? 1 : 0);
    int shift = 0;
    while ((1 << shift) < G.brick) shift++;
    S.brick_shift = shift;
    S.bnx = (nx + G.brick - 1) / G.brick; S.bny = (ny + G.brick - 1) / G.brick; S.bnz = (nz + G.brick - 1) / G.brick;
    const size_t nb    = (size_t)S.bnx * S.bny * S.bnz;
    const size_t psize = quantized ? 2 : 8;
    HIPCHK(hipMalloc(&d_ta, n * psize));
    HIPCHK(hipMalloc(&d_tb, n * psize));
    HIPCHK(hipMalloc(&G.d_bounds, nb * psize + 16));  // padded: the LDS stage copies whole 16-byte words
    HIPCHK(hipMemsetAsync(G.d_bounds, 0, nb * psize + 16, G.stream));
    launch_build_bounds(d_raw, quantized, G.d_bounds, d_ta, d_tb, nx, ny, nz, G.radius, G.brick, G.stream);
    HIPCHK(hipGetLastError());
    if (quantized)
    {
        S.bounds_u8 = (const unsigned char*)G.d_bounds;
        S.cells_u8  = (const uint2*)G.d_cells;
    }
    else
    {
        S.bounds_f32 = (const float*)G.d_bounds;
        S.cells_f32  = (const float*)G.d_cells;
    }
    // cells with a non-empty cell in their neighbourhood: input of the certified-empty table of the global-majorant estimator
    if (G.use_empty_table && hipMalloc((void**)&G.d_danger, n) == hipSuccess)
    {
        launch_danger(S, quantized, G.d_danger, G.stream);
        HIPCHK(hipGetLastError());
    }
    else { (void)hipGetLastError(); G.d_danger = nullptr; }  // no memory for it: the estimator fetches every cell, same bits
    // ... and of the direction table of the exit flights
    if (G.d_danger && G.use_exit && hipMalloc((void**)&G.d_exit, 3 * n) == hipSuccess)
    {
        launch_exit_table(G.d_danger, G.d_exit, nx, ny, nz, G.stream);
        HIPCHK(hipGetLastError());
    }
    else { (void)hipGetLastError(); G.d_exit = nullptr; }   // none: every path walks to the box exit, same bits
    G.bound_mask_valid = false;
    HIPCHK(hipStreamSynchronize(G.stream));  // caller may free h_volume on return (host.cpp:1343); scratch freed by the guard
    S.linear      = G.linear ? 1 : 0;
    G.have_volume = true;
    return VP_OK;
}

int do_opacity(const float* dir)
{
    G.epoch++;  // staged look-ahead frames no longer describe this scene ...
    (void)la_quiesce();  // ... and batches in flight must not see device buffers change under them

    if (!G.have_volume) return fail(VP_E_STATE, "precompute_opacity before init_cuda");
    size_t n = (size_t)G.S.nx * G.S.ny * G.S.nz;
    if (!G.d_opacity) HIPCHK(hipMalloc((void**)&G.d_opacity, n * sizeof(float)));
    SceneDev S = G.S;
    S.linear   = G.linear ? 1 : 0;
    launch_opacity(S, G.quant, G.opacity_lds, dir, G.d_opacity, G.stream);
    HIPCHK(hipGetLastError());
    // the integrator's copy: per voxel its clamped 2x2x2 neighbourhood, 32 bytes -- a lookup (frames > 10, more than 20 scatters:
    // 20 per sample on the frame-filling cloud) touches one cache line instead of four
    // (best effort, ADVICE r4: the copy is 8x the table -- 4.3 GB at 512^3, 34 GB at 1024^3.  Where it cannot be had the integrator reads
    // the plain table, eight loads instead of two, the same bits -- like every other table of this file that is an optimisation)
    const bool no_cells = !G.use_opacity_cells;   // (VP_NO_OPACITY_CELLS=1: the fall-back on purpose -- the knob test renders through it)
    if (no_cells && G.d_opacity_cells) { HIPCHK(hipFree(G.d_opacity_cells)); G.d_opacity_cells = nullptr; }
    if (!G.d_opacity_cells && !no_cells && hipMalloc((void**)&G.d_opacity_cells, n * 8 * sizeof(float)) != hipSuccess)
    {
        (void)hipGetLastError();
        G.d_opacity_cells = nullptr;
    }
    if (G.d_opacity_cells)
    {
        launch_pack_f32(G.d_opacity, G.d_opacity_cells, G.S.nx, G.S.ny, G.S.nz, false, G.stream);
        HIPCHK(hipGetLastError());
    }
    G.S.opacity = G.d_opacity;
    G.S.opacity_cells = G.d_opacity_cells;
    return VP_OK;
}

int build_env_tables();

int do_envmap(const vp_float4* data, int w, int h)
{
    G.epoch++;  // staged look-ahead frames no longer describe this scene ...
    (void)la_quiesce();  // ... and batches in flight must not see device buffers change under them

    int rc = ensure_device();
    if (rc) return rc;
    if (!data || w <= 0 || h <= 0) return fail(VP_E_ARG, "bad envmap");
    if (w != G.env_w || h != G.env_h)
    {
        HIPCHK(hipStreamSynchronize(G.stream));
        if (G.d_env) HIPCHK(hipFree(G.d_env));
        HIPCHK(hipMalloc((void**)&G.d_env, (size_t)w * h * sizeof(float4)));
        G.env_w = w; G.env_h = h;
    }
    HIPCHK(hipMemcpyAsync(G.d_env, data, (size_t)w * h * sizeof(float4), hipMemcpyHostToDevice, G.stream));
    HIPCHK(hipStreamSynchronize(G.stream));  // caller owns `data`
    G.S.env = G.d_env; G.S.env_w = w; G.S.env_h = h;
    G.have_env = true;
    G.env_tables = false;
    if (G.env_mis) return build_env_tables();
    return VP_OK;
}

// init_envmap kernel.cu:1144-1210: luminance CDFs and HDRpdfnormAlt for the current environment
int build_env_tables()
{
    G.epoch++;  // staged look-ahead frames no longer describe this scene ...
    (void)la_quiesce();  // ... and batches in flight must not see device buffers change under them

    if (G.env_tables || !G.have_env) return VP_OK;
    const int w = G.env_w, h = G.env_h;
    float *lum = nullptr, *rows = nullptr, *norm = nullptr;
    if (G.d_env_cdf_x) HIPCHK(hipFree(G.d_env_cdf_x));
    if (G.d_env_cdf_y) HIPCHK(hipFree(G.d_env_cdf_y));
    HIPCHK(hipMalloc((void**)&G.d_env_cdf_x, (size_t)w * h * sizeof(float)));
    HIPCHK(hipMalloc((void**)&G.d_env_cdf_y, (size_t)h * sizeof(float)));
    HIPCHK(hipMalloc((void**)&lum, (size_t)w * h * sizeof(float)));
    HIPCHK(hipMalloc((void**)&rows, (size_t)h * sizeof(float)));
    HIPCHK(hipMalloc((void**)&norm, sizeof(float)));
    launch_env_tables(G.d_env, w, h, lum, rows, G.d_env_cdf_x, G.d_env_cdf_y, norm, G.stream);
    HIPCHK(hipGetLastError());
    float hnorm = 0.0f;
    HIPCHK(hipMemcpyAsync(&hnorm, norm, sizeof(float), hipMemcpyDeviceToHost, G.stream));
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipFree(lum)); HIPCHK(hipFree(rows)); HIPCHK(hipFree(norm));
    G.S.env_cdf_x = G.d_env_cdf_x; G.S.env_cdf_y = G.d_env_cdf_y; G.S.env_pdfnorm_alt = hnorm;
    G.env_tables = true;
    return VP_OK;
}

// an event from the pool, or a new one; nullptr if the runtime cannot create one (the launch then goes untimed)
hipEvent_t get_event()
{
    if (!G.event_pool.empty()) { hipEvent_t e = G.event_pool.back(); G.event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}
void put_event(hipEvent_t e) { if (e) G.event_pool.push_back(e); }
// a pair of events around one kernel of a launch (per-class kernel time, vp_render_class_time_ms); best effort
struct ClassTimer
{
    int cls; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool ok = false;
    ClassTimer(int c, hipStream_t s) : cls(c), st(s)
    {
        a = get_event(); b = get_event();
        ok = a && b && hipEventRecord(a, st) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    ~ClassTimer() { put_event(a); put_event(b); }   // a timer that was never stopped (an early return) hands its events back
    void stop()
    {
        if (ok && hipEventRecord(b, st) == hipSuccess) { G.class_events.push_back({cls, a, b}); a = b = nullptr; }
        else (void)hipGetLastError();
        put_event(a); put_event(b); a = b = nullptr;
        while (G.class_events.size() > 3 * kMaxPendingEvents)
        {
            auto  ev = G.class_events.front();
            float ms = 0.0f;
            G.class_events.pop_front();
            if (hipEventQuery(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) G.class_ms[ev.cls] += ms;
            else (void)hipGetLastError();
            put_event(ev.a); put_event(ev.b);
        }
    }
};
// keep at most kMaxPendingEvents launch pairs: fold the oldest into the running sum (its elapsed time if the pair
// has completed; a launch this old that has not is counted without a time rather than waited for)
void trim_events()
{
    while (G.events.size() > kMaxPendingEvents)
    {
        auto  ev = G.events.front();
        float ms = 0.0f;
        G.events.pop_front();
        if (hipEventQuery(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) G.timed_ms += ms;
        else (void)hipGetLastError();
        put_event(ev.first); put_event(ev.second);
    }
}

// the shard of this context (include/volpath.h vp_tile_owner): its 8x8 tiles and the pixels of the image they hold
struct Shard { unsigned tiles_x, tiles_y, owned; size_t per_frame; };  // owned = tiles, per_frame = pixels = samples per frame
Shard shard_of(const Param* p)
{
    Shard s;
    s.tiles_x = (p->width + 7) / 8;
    s.tiles_y = (p->height + 7) / 8;
    s.owned   = 0;
    s.per_frame = 0;
    for (unsigned ty = 0; ty < s.tiles_y; ty++)
    {
        const unsigned rows = std::min(8u, p->height - ty * 8u);
        // tiles tx of this row with (tx + shift) % world == rank
        for (unsigned tx = (G.rank + G.world - tile_row_shift(ty, G.world)) % G.world; tx < s.tiles_x; tx += G.world)
        {
            s.owned++;
            s.per_frame += (size_t)rows * std::min(8u, p->width - tx * 8u);
        }
    }
    return s;
}
// frames of per_frame samples one staged launch may hold: the configured cap, a quarter of the memory that is free
// now (plus what the target buffer already holds), and the 32-bit sample queue
size_t stage_frames_cap(size_t per_frame, size_t have_bytes)
{
    size_t cap = G.max_stage_bytes;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap = std::min(cap, std::max(have_bytes, (free_b + have_bytes) / 4));
    else (void)hipGetLastError();
    size_t f = cap / (per_frame * sizeof(float4));
    f = std::min<size_t>(f, 0xfffffff0u / per_frame);
    return std::max<size_t>(f, 1);
}

// The per-pixel table of the restart crawl in front of the volume (vp_kernels.hip crawl_table_k) for the local-majorant
// estimators.  It depends on the camera, the box, the bound table and the image size only -- not on the frame -- and is
// rebuilt (one small kernel, synchronously: launches on other streams read it) when any of those changed.
int ensure_crawl_table(const Param* p, const float4** out)
{
    *out = nullptr;
    const bool global = G.est == VP_EST_GLOBAL;
    // the certificate of the global-majorant estimator is stated for trilinear fetches (cell = floor(p*N - 0.5))
    if (global ? !(G.use_empty_table && G.d_danger && G.linear) : !G.use_crawl_table) return VP_OK;
    struct K { SceneDev S; unsigned w, h; int control, quant, global; unsigned long long epoch; };
    std::vector<unsigned char> key(sizeof(K), 0);
    K* k = reinterpret_cast<K*>(key.data());
    memcpy(&k->S, &G.S, sizeof(SceneDev));
    k->S.linear = G.linear ? 1 : 0; k->S.env = nullptr; k->S.opacity = nullptr; k->S.opacity_cells = nullptr; k->S.env_cdf_x = k->S.env_cdf_y = nullptr;  // not read by the walk
    k->S.env_w = k->S.env_h = 0; k->S.env_pdfnorm_alt = 0.0f;
    memset(k->S.sun_dir, 0, sizeof k->S.sun_dir); memset(k->S.sun_power, 0, sizeof k->S.sun_power); memset(k->S.sun_orig, 0, sizeof k->S.sun_orig);
    k->w = p->width; k->h = p->height;
    k->control = (G.est == VP_EST_DECOMP && G.trk == VP_TRACK_SPECTRAL) ? 1 : 0;
    k->quant = G.quant; k->epoch = G.epoch; k->global = global ? 1 : 0;
    const size_t need = (size_t)p->width * p->height * 2 * sizeof(float4);
    if (key != G.crawl_key || !G.d_crawl)
    {
        if (la_quiesce()) return VP_E_NODEVICE;   // batches in flight read the old table
        HIPCHK(hipStreamSynchronize(G.stream));
        if (need > G.crawl_bytes)
        {
            if (G.d_crawl) HIPCHK(hipFree(G.d_crawl));
            G.d_crawl = nullptr; G.crawl_bytes = 0; G.crawl_key.clear();
            if (hipMalloc((void**)&G.d_crawl, need) != hipSuccess)
            {
                (void)hipGetLastError();
                G.d_crawl = nullptr;
                return VP_OK;   // no table: the paths walk the crawl themselves, same bits
            }
            G.crawl_bytes = need;
        }
        if (global)
        {
            SceneDev S = G.S;
            S.linear   = 1;
            launch_empty_table(S, p->width, p->height, G.d_danger, G.d_crawl, G.stream);
        }
        else
        {
            SceneDev S = G.S;
            S.linear   = G.linear ? 1 : 0;
            launch_crawl_table(S, G.quant, p->width, p->height, k->control != 0, G.use_empty_table ? G.d_danger : nullptr, G.d_crawl, G.stream);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(G.stream));
        G.crawl_key = key;
    }
    *out = G.d_crawl;
    return VP_OK;
}

// Where sun shadow rays end early (vp_kernels.hip sun_clip_k): for the counter-based streams, whose shadow rays draw from
// sub-streams of their own.  Depends on the volume, its box, the filter mode and the sun direction; one kernel over the
// non-empty cells (a millisecond at 256^3), synchronously like the other tables.
int ensure_sun_clip(const unsigned short** out, float* ds)
{
    *out = nullptr; *ds = 0.0f;
    if (!G.use_sun_clip || G.rng == VP_RNG_SAMPLERH || !G.d_danger || !G.linear) return VP_OK;
    struct K { int nx, ny, nz, quant; float bmin[3], bmax[3], sun[3]; unsigned long long epoch; };
    std::vector<unsigned char> key(sizeof(K), 0);
    K* k = reinterpret_cast<K*>(key.data());
    k->nx = G.S.nx; k->ny = G.S.ny; k->nz = G.S.nz; k->quant = G.quant; k->epoch = G.epoch;
    memcpy(k->bmin, G.S.bmin, sizeof k->bmin); memcpy(k->bmax, G.S.bmax, sizeof k->bmax); memcpy(k->sun, G.S.sun_dir, sizeof k->sun);
    if (key != G.sunclip_key || !G.d_sunclip)
    {
        if (la_quiesce()) return VP_E_NODEVICE;   // batches in flight read the old table
        HIPCHK(hipStreamSynchronize(G.stream));
        const size_t n = (size_t)G.S.nx * G.S.ny * G.S.nz;
        if (!G.d_sunclip && hipMalloc((void**)&G.d_sunclip, n * sizeof(unsigned short)) != hipSuccess)
        {
            (void)hipGetLastError();
            G.d_sunclip = nullptr;
            return VP_OK;   // no table: the shadow rays walk to their end, same bits
        }
        SceneDev S = G.S;
        S.linear   = 1;
        G.sunclip_ds = sun_clip_step(S);
        launch_sun_clip(S, G.d_danger, G.sunclip_ds, G.d_sunclip, G.stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(G.stream));
        G.sunclip_key = key;
    }
    *out = G.d_sunclip; *ds = G.sunclip_ds;
    return VP_OK;
}

// Are the samples of the light class independent of the draws for this medium (vp_kernels.hip light_identity_k)?  Global-majorant
// and decomposition estimators with spectral tracking; the bounded estimator's heat channel counts segments, scalar tracking has no
// light class, float bound tables are not enumerable.
int ensure_light_identity(const Param* p, bool* out);
int ensure_light_const(const Param* p, bool* out)
{
    *out = false;
    if (!G.use_light_const || G.count) return VP_OK;   // (the counting build walks the light paths: its counters are the estimator's)
    return ensure_light_identity(p, out);
}
// the bytes that occur as maxima in the (uchar) bound table: 256 bits on the device (d_light_flag + 1) and on the host
int ensure_bound_mask()
{
    if (G.bound_mask_valid) return VP_OK;
    if (!G.d_light_flag) HIPCHK(hipMalloc((void**)&G.d_light_flag, 9 * sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(G.d_light_flag + 1, 0, 8 * sizeof(unsigned), G.stream));
    launch_bound_bytes((const unsigned char*)G.d_bounds, (size_t)G.S.bnx * G.S.bny * G.S.bnz, G.d_light_flag + 1, G.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(G.h_bound_mask, G.d_light_flag + 1, sizeof G.h_bound_mask, hipMemcpyDeviceToHost, G.stream));
    HIPCHK(hipStreamSynchronize(G.stream));
    G.bound_mask_valid = true;
    return VP_OK;
}
// Exit flights (vp_kernels.hip render_k): which launches may end a path that can only leave the box.  Spectral tracking,
// passive environment, trilinear fetches (the emptiness certificate is stated for them), not the bounded estimator (it counts
// segments); local majorants: a uchar bound table with at most four distinct maxima (the Julia grids have two: 0 and 255) -- the
// test checks every majorant a segment through empty cells can have.
int exit_flights(LaunchDev& L)
{
    L.exit_oct = nullptr; L.exit_start = -(1 << 30); L.exit_nbytes = 0; L.exit_bytes = 0;
    if (!G.use_exit || !G.d_exit || !G.linear || G.trk || G.env_mis || G.est == VP_EST_BOUNDED) return VP_OK;
    if (G.est != VP_EST_GLOBAL)
    {
        // Off by default: with local majorants the way out through empty bricks is a restart segment and ONE free flight per 0.05 of
        // length, made by lanes that ride along with their wave's fetching lanes -- ending those paths early removes 11 % of the
        // lane-steps of the decomposition workloads and not one wave-iteration (C3 +2 %, c3ref 0, c4s -0.5 %, sampler.h -1...-2 %:
        // DESIGN.md section 5).  The global-majorant walk is 800 null collisions per unit length: there it is +27 %.
        if (!G.exit_local || !G.quant) return VP_OK;
        int rc = ensure_bound_mask();
        if (rc) return rc;
        unsigned nb = 0, packed = 0;
        for (unsigned b = 0; b < 256; b++)
            if (G.h_bound_mask[b >> 5] >> (b & 31u) & 1u)
            {
                if (nb < 4) packed |= b << (8u * nb);
                nb++;
            }
        if (nb == 0 || nb > 4) return VP_OK;
        L.exit_nbytes = nb; L.exit_bytes = packed;
    }
    L.exit_oct   = G.d_exit;
    L.exit_start = VP_EXIT_TRIP - (int)G.exit_k;
    return VP_OK;
}
// the check itself; also what approach_k rests on (a null collision in empty space leaves a throughput of 1 as it is)
int ensure_light_identity(const Param* p, bool* out)
{
    *out = false;
    if (G.trk != VP_TRACK_SPECTRAL || G.est == VP_EST_BOUNDED) return VP_OK;
    const bool local = G.est != VP_EST_GLOBAL;
    if (local && !G.quant) return VP_OK;
    const float key[7] = {p->sigma_t.x, p->sigma_t.y, p->sigma_t.z, p->density, p->g, (float)G.est, (float)G.brick};
    if (G.light_epoch != G.epoch || memcmp(key, G.light_key, sizeof key) != 0)
    {
        if (!G.d_light_flag) HIPCHK(hipMalloc((void**)&G.d_light_flag, 9 * sizeof(unsigned)));
        if (local)
        {
            int rcm = ensure_bound_mask();
            if (rcm) return rcm;
        }
        const unsigned one = 1u;
        unsigned flag = 0u;
        HIPCHK(hipMemcpyAsync(G.d_light_flag, &one, sizeof one, hipMemcpyHostToDevice, G.stream));
        ParamDev P;
        memcpy(&P, p, sizeof(Param));
        launch_light_identity(P, local, G.d_light_flag + 1, G.d_light_flag, G.stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&flag, G.d_light_flag, sizeof flag, hipMemcpyDeviceToHost, G.stream));
        HIPCHK(hipStreamSynchronize(G.stream));
        G.light_const = flag == 1u;
        memcpy(G.light_key, key, sizeof key);
        G.light_epoch = G.epoch;
    }
    *out = G.light_const;
    return VP_OK;
}

// The light kernel of the global-majorant estimator looks the throughput of a path up by its number of null collisions
// (vp_kernels.hip thr_table_k); the sequence depends on sigma_t, density and g only.
int ensure_thr_table(const Param* p, const float** out)
{
    *out = nullptr;
    const float key[5] = {p->sigma_t.x, p->sigma_t.y, p->sigma_t.z, p->density, p->g};
    if (!G.thr_valid || !G.d_thr || memcmp(key, G.thr_key, sizeof key) != 0)
    {
        if (la_quiesce()) return VP_E_NODEVICE;   // batches in flight read the old table
        HIPCHK(hipStreamSynchronize(G.stream));
        if (!G.d_thr) HIPCHK(hipMalloc((void**)&G.d_thr, G.thr_entries * sizeof(float)));
        ParamDev P;
        memcpy(&P, p, sizeof(Param));
        launch_thr_table(P, G.d_thr, G.thr_entries, G.stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(G.stream));
        memcpy(G.thr_key, key, sizeof key);
        G.thr_valid = true;
    }
    *out = G.d_thr;
    return VP_OK;
}

// The pixel lists of this context: the pixels of its tiles, tile by tile (row-major tiles, row-major pixels within a tile: the
// order keeps the rays of a wave in one pencil of the volume), general pixels first, then -- with spectral tracking and a pixel
// table -- the light class (camera rays that meet certified-empty cells over their whole chord) and the pixels whose camera ray
// misses the box.  Built on the GPU (pixlist_*_k: a stable three-way partition of the tile-ordered pixels by the class in the pixel
// table; the host only reads back the three counts).  Rebuilt when the image size, the shard or the table changes.
int ensure_pixel_lists(const Param* p, const float4* table, const Shard& sh)
{
    const bool light = G.use_light && table && G.trk == VP_TRACK_SPECTRAL && !(G.est != VP_EST_GLOBAL && !G.use_light_local);
    struct K { unsigned w, h, rank, world; int light; };
    std::vector<unsigned char> key(sizeof(K), 0);
    K* k = reinterpret_cast<K*>(key.data());
    k->w = p->width; k->h = p->height; k->rank = G.rank; k->world = G.world; k->light = light ? 1 : 0;
    std::vector<unsigned char> shape_key = key;   // what the tile enumeration depends on (not the camera)
    if (light) key.insert(key.end(), G.crawl_key.begin(), G.crawl_key.end());
    if (key == G.tiles_key && G.d_tiles) return VP_OK;
    if (la_quiesce()) return VP_E_NODEVICE;   // batches in flight read the old lists
    HIPCHK(hipStreamSynchronize(G.stream));
    if ((size_t)sh.owned * 64 > (size_t)0xffffffffu) return fail(VP_E_ARG, "image too large for the 32-bit pixel-list index");
    const unsigned nblocks = pixel_list_blocks(sh.owned);
    if (shape_key != G.tiles_shape_key || !G.d_tile_rows)
    {
        // first owned tile of each tile row: depends on the image size and the shard only
        std::vector<unsigned> rows(sh.tiles_y + 1, 0);
        for (unsigned ty = 0; ty < sh.tiles_y; ty++)
        {
            const unsigned first = (G.rank + G.world - tile_row_shift(ty, G.world)) % G.world;
            rows[ty + 1] = rows[ty] + (first < sh.tiles_x ? (sh.tiles_x - first + G.world - 1) / G.world : 0u);
        }
        if (rows[sh.tiles_y] != sh.owned) return fail(VP_E_STATE, "tile enumeration disagrees with the shard (%u vs %u tiles)", rows[sh.tiles_y], sh.owned);
        if (G.d_tile_rows) HIPCHK(hipFree(G.d_tile_rows));
        if (G.d_tile_scratch) HIPCHK(hipFree(G.d_tile_scratch));
        G.d_tile_rows = G.d_tile_scratch = nullptr; G.tiles_shape_key.clear();
        HIPCHK(hipMalloc((void**)&G.d_tile_rows, rows.size() * sizeof(unsigned)));
        HIPCHK(hipMalloc((void**)&G.d_tile_scratch, ((size_t)3 * nblocks + 4) * sizeof(unsigned)));
        HIPCHK(hipMemcpyAsync(G.d_tile_rows, rows.data(), rows.size() * sizeof(unsigned), hipMemcpyHostToDevice, G.stream));
        HIPCHK(hipStreamSynchronize(G.stream));   // `rows` goes out of scope
        G.tiles_shape_key = shape_key;
    }
    if (sh.per_frame > G.tiles_cap)
    {
        if (G.d_tiles) HIPCHK(hipFree(G.d_tiles));
        G.d_tiles = nullptr; G.tiles_cap = 0; G.tiles_key.clear();
        HIPCHK(hipMalloc((void**)&G.d_tiles, sh.per_frame * sizeof(unsigned)));
        G.tiles_cap = sh.per_frame;
    }
    unsigned* d_totals = G.d_tile_scratch + (size_t)3 * nblocks;
    launch_pixel_lists(p->width, p->height, G.rank, G.world, sh.owned, G.d_tile_rows, light ? table : nullptr, G.d_tile_scratch, d_totals, G.d_tiles, G.stream);
    HIPCHK(hipGetLastError());
    unsigned totals[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(totals, d_totals, sizeof totals, hipMemcpyDeviceToHost, G.stream));
    HIPCHK(hipStreamSynchronize(G.stream));
    if ((size_t)totals[0] + totals[1] + totals[2] != sh.per_frame)
        return fail(VP_E_STATE, "pixel lists hold %zu pixels, the shard has %zu", (size_t)totals[0] + totals[1] + totals[2], sh.per_frame);
    G.n_general = totals[0]; G.n_light = totals[1]; G.n_miss = totals[2];
    G.tiles_key = key;
    return VP_OK;
}

// where a render launch goes: the caller's stream with the shared staging buffer, or a look-ahead slot
struct Target { hipStream_t stream; float4** stage; size_t* stage_bytes; unsigned* queue; int index; };

int do_render(vp_float4* d_out, int first, int nframes, const Param* p, bool stage_only = false, const Target* tgt = nullptr)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!G.have_volume) return fail(VP_E_STATE, "render before init_cuda");
    if (!G.have_env) return fail(VP_E_STATE, "render before init_envmap");
    if (!G.have_sun) return fail(VP_E_STATE, "render before set_sun");
    if (!G.have_cam) return fail(VP_E_STATE, "render before copy_inv_view_matrix");
    if (!d_out || !p || nframes <= 0 || first < 0) return fail(VP_E_ARG, "bad render arguments");
    const Target main_tgt = {G.stream, &G.d_stage, &G.stage_bytes, G.d_queue, 0};
    const Target& T = tgt ? *tgt : main_tgt;
    if (p->width == 0 || p->height == 0 || p->width > 65535 || p->height > 65535)
        return fail(VP_E_ARG, "image %ux%u out of range (sampler.h packs x<<16|y)", p->width, p->height);
    if (G.trk && G.env_mis) return fail(VP_E_STATE, "scalar tracking builds exist with passive environment lighting only");
    if (G.trk && G.count) return fail(VP_E_STATE, "work counters are not built for the scalar tracking kernels");
    if (G.rng == VP_RNG_PHILOX7 && (G.trk || G.env_mis))
        return fail(VP_E_STATE, "VP_RNG_PHILOX7 is built for spectral tracking with passive environment lighting only");
    if (G.est == VP_EST_DECOMP && first + nframes - 1 > 10 && !G.S.opacity)
        return fail(VP_E_NOOPACITY, "frames beyond 10 need precompute_opacity (kernel.cu:2183, host.cpp:336-343)");
    LaunchDev L = {};
    static_assert(sizeof(ParamDev) == sizeof(Param) && sizeof(Param) == 44, "Param layout (param.h:4-12)");
    memcpy(&L.P, p, sizeof(Param));
    const Shard sh = shard_of(p);
    L.out = (float4*)d_out;
    L.counters = G.count ? G.d_counters : nullptr;
    L.key0 = G.key0; L.key1 = G.key1;
    L.wait_lanes = G.wait_lanes; L.wait_iters = G.wait_iters; L.setup_lanes = G.setup_lanes; L.end_lanes = G.end_lanes;
    {
        // The chromatic local-majorant kernels (BASELINE configs[3]/[4]'s shape) wait for 32 parked lanes instead of 24: their event pass
        // is the most expensive (three-channel collision block, the optical-depth lookup) and comes every four steps on a frame-filling
        // cloud; round 5's sweep with a 0.05 % noise floor: c4f +1.6 %, every other workload within its noise (profiles/experiments/
        // r05_knob_sweeps.txt).  Performance only: the knob test renders the same bits at 1...64.
        const bool ach = p->sigma_t.x == p->sigma_t.y && p->sigma_t.y == p->sigma_t.z && p->albedo.x == p->albedo.y && p->albedo.y == p->albedo.z;
        if (!G.wait_lanes_set && G.est != VP_EST_GLOBAL && !ach && G.trk == VP_TRACK_SPECTRAL) L.wait_lanes = 32;
    }
    if (sh.per_frame == 0) return VP_OK;
    rc = ensure_crawl_table(p, &L.crawl);
    if (rc) return rc;
    rc = ensure_pixel_lists(p, L.crawl, sh);
    if (rc) return rc;
    rc = ensure_sun_clip(&L.sun_clip, &L.clip_ds);
    if (rc) return rc;
    L.count_clips = getenv("VP_DEBUG_COUNT_CLIPS") ? 1u : 0u;
    rc = exit_flights(L);
    if (rc) return rc;
    // look-ahead batches carry their slot's cancel word and their number (la_render_slot, la_quiesce); the caller's own launches cannot be cancelled
    L.cancel = tgt ? G.d_cancel + T.index : nullptr; L.batch_id = G.batch_seq[T.index];
    bool light_const = false;
    if (G.n_light)
    {
        rc = ensure_light_const(p, &light_const);
        if (rc) return rc;
    }
    G.last_light_const = light_const ? 1 : 0;
    if (G.est == VP_EST_GLOBAL && G.n_light && !light_const)
    {
        rc = ensure_thr_table(p, &L.thr_table);
        if (rc) return rc;
        L.thr_n = G.thr_entries;
    }
    // the camera rays' free flights through certified-empty cells in kernels of their own, ahead of the integrator (approach_k: global
    // majorant; approach_local_k: decomposition estimator; spectral tracking, passive environment, staged launches)
    bool approach = false, approach_thr = false;
    if (G.use_approach && (G.est == VP_EST_GLOBAL || (G.est == VP_EST_DECOMP && G.use_approach_local)) && !G.trk && !G.env_mis && L.crawl && G.n_general &&
        (!G.count || getenv("VP_COUNT_APPROACH")))   // counting launches: the integrator makes every step itself unless asked (block tallies)
    {
        // global majorant: one majorant for the whole walk, checked here; decomposition: approach_local_k checks each segment's own
        bool identity = true;
        if (G.est == VP_EST_GLOBAL) rc = ensure_light_identity(p, &identity);
        if (rc) return rc;
        approach = true;
        if (!identity)
        {
            // the walk's null collisions change the throughput: render_k looks it up by their number (the light kernel's table)
            rc = ensure_thr_table(p, &L.thr_table);
            if (rc) return rc;
            L.thr_n  = G.thr_entries;
            approach_thr = true;
        }
    }
    const size_t per_frame = sh.per_frame;
    if (0xfffffff0u / per_frame < 1) return fail(VP_E_ARG, "image too large for the 32-bit sample queue");
    L.stage_stride = (unsigned)per_frame;
    size_t max_f = (nframes > 1 || stage_only) ? stage_frames_cap(per_frame, *T.stage_bytes) : 1;
    SceneDev S = G.S;
    S.linear   = G.linear ? 1 : 0;
    // The staging slot of the decomposition estimator's hand-over holds the segment origin and the distance reached in it: the stream's
    // state (a pair index, or sampler.h's two words) goes beside it.  Sized ONCE, before the launch loop (no synchronisation, no early
    // return between a launch's events).
    const bool appr_aux_needed = approach && G.est == VP_EST_DECOMP && (nframes > 1 || stage_only);
    if (appr_aux_needed)
    {
        const int    ti    = T.index;
        // (a look-ahead slot is sized for the largest batch at once: a slot that grew with every doubling of the ramp would
        // synchronise its stream -- and the batch running beside it -- at every step)
        const size_t fr4   = stage_only ? std::max<size_t>((size_t)nframes, (size_t)std::max(G.la_max, 1)) : (size_t)nframes;
        const size_t need4 = per_frame * std::min<size_t>(fr4, max_f) * sizeof(uint2);
        if (per_frame * std::min<size_t>((size_t)nframes, max_f) * sizeof(uint2) > G.appr_aux_bytes[ti])
        {
            HIPCHK(hipStreamSynchronize(T.stream));
            if (G.d_appr_aux[ti]) HIPCHK(hipFree(G.d_appr_aux[ti]));
            G.d_appr_aux[ti] = nullptr; G.appr_aux_bytes[ti] = 0;
            if (hipMalloc((void**)&G.d_appr_aux[ti], need4) != hipSuccess) { (void)hipGetLastError(); G.d_appr_aux[ti] = nullptr; }   // no walk ahead then
            else G.appr_aux_bytes[ti] = need4;
        }
    }
    for (int done = 0; done < nframes;)
    {
        int f = (int)std::min<size_t>((size_t)(nframes - done), max_f);
        if (stage_only && f != nframes) return fail(VP_E_ARG, "look-ahead batch does not fit the staging buffer");
        L.frame0 = first + done;
        L.nframes = f;
        if (f > 1 || stage_only)
        {
            size_t need = per_frame * (size_t)f * sizeof(float4);
            if (need > *T.stage_bytes)
            {
                // (a look-ahead slot is sized for the largest batch at once: growing with every doubling of the ramp would synchronise
                // its stream, and the batch running beside it, at every step)
                const size_t exact = need;
                if (stage_only) need = per_frame * std::min<size_t>(std::max<size_t>((size_t)f, (size_t)std::max(G.la_max, 1)), max_f) * sizeof(float4);
                HIPCHK(hipStreamSynchronize(T.stream));
                HIPCHK(hipStreamSynchronize(G.stream));  // add-kernels of earlier frames may still read the old buffer
                if (*T.stage) HIPCHK(hipFree(*T.stage));
                *T.stage = nullptr; *T.stage_bytes = 0;
                hipError_t me = hipMalloc((void**)T.stage, need);
                if (me != hipSuccess && need > exact) { (void)hipGetLastError(); need = exact; me = hipMalloc((void**)T.stage, need); }
                if (me != hipSuccess)
                {
                    // another allocator took the memory since it was measured: a smaller batch renders the same bits
                    (void)hipGetLastError();
                    *T.stage = nullptr;
                    if (stage_only) return fail(VP_E_NOMEM, "no memory for a look-ahead batch of %d frames", f);
                    if (f > 1) { max_f = (size_t)std::max(f / 2, 1); continue; }
                    return fail(VP_E_NOMEM, "no memory for one staged frame (%zu bytes)", need);
                }
                *T.stage_bytes = need;
            }
            L.stage = *T.stage;
        }
        else
            L.stage = nullptr;
        // per-pixel constants of the launch (the box-missing pixels; the light class where it is written by miss_fill_k) are staged once,
        // in the launch's first row: the slots behind the general (and an integrated light) class
        L.const_from = 0xffffffffu; L.stage_const = nullptr;
        if (L.stage && G.use_const_rows)
        {
            L.const_from  = (unsigned)(G.n_general + ((G.n_light && !light_const) ? G.n_light : 0u));
            L.stage_const = L.stage;
        }
        G.last_const_from = L.const_from;
        HIPCHK(hipMemsetAsync(T.queue, 0, 2 * kQueueWords * sizeof(unsigned), T.stream));
        // the brick table goes through LDS when it fits (decomposition estimator, byte table <= 64 KiB)
        const bool lds_bounds = G.use_lds_bounds && G.est == VP_EST_DECOMP && G.quant && !G.env_mis && !G.trk &&
                                (size_t)S.bnx * S.bny * S.bnz <= (size_t)VP_LDS_BOUND_ENTRIES;
        hipEvent_t e0 = get_event(), e1 = get_event();
        bool timed = e0 && e1 && hipEventRecord(e0, T.stream) == hipSuccess;
        hipError_t le = hipSuccess;
        // the fork point of the light kernel's auxiliary stream: BEFORE the general kernel is queued (the two run side by side),
        // after the queue heads are zeroed; an event of its own, created on first use
        // (the same fork serves the helper workgroups of the LDS-table kernel, below, when no light kernel needs the stream)
        // (not for look-ahead batches: two of them overlap -- the next one's approach walk and first workgroups run beside the current
        // one's body and tail -- only if the current one leaves registers free: four LDS-table waves per SIMD do, the helper's fifth does
        // not.  C3 host loop 1301 -> 1510 Msamples/s without it, profiles/r03_render_kernel_lookahead.txt)
        // COUPLING (two tuning decisions that depend on each other): approach_local_k needs 47 vector registers (kernel_resources.py);
        // beside four 97-102-register LDS-table waves AND the helper's fifth 96-register wave a SIMD has 27 left, beside the four
        // alone 124.  If approach_local_k's register count or the helper's occupancy changes, re-measure the `!tgt` below.
        const bool lds_helper = lds_bounds && G.lds_helper && G.n_general && !(G.n_light && !light_const) && !tgt;
        bool fork_recorded = false;
        if ((G.n_light && G.n_general && !light_const && G.light_overlap) || lds_helper)
        {
            const int ti = T.index;
            if (!G.aux_ev[ti][0] && hipEventCreateWithFlags(&G.aux_ev[ti][0], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); G.aux_ev[ti][0] = nullptr; }
            if (G.aux_ev[ti][0])
            {
                fork_recorded = hipEventRecord(G.aux_ev[ti][0], T.stream) == hipSuccess;
                if (!fork_recorded) (void)hipGetLastError();
            }
        }
        // one launch per pixel class: the general pixels, then the light ones (their own kernel, their own sample queues)
        for (int cls = 0; cls < 2 && le == hipSuccess; cls++)
        {
            const unsigned nt = cls ? G.n_light : G.n_general;
            if (!nt) continue;
            if (G.debug_only_class >= 0 && G.debug_only_class != cls) continue;  // VP_DEBUG_ONLY_CLASS: block tallies of one kernel
            L.pixels      = G.d_tiles + (cls ? G.n_general : 0);
            L.nslots      = nt;
            L.slot_base   = cls ? G.n_general : 0u;
            L.total_items = (unsigned)((size_t)nt * (size_t)f);
            L.queue       = T.queue + (cls ? kQueueWords : 0);
            // chunks of pixels x frames (general class): only when the frame count is a multiple of the frame block
            L.chunk_fshift = (!cls && G.chunk_fshift && f % (1 << G.chunk_fshift) == 0) ? G.chunk_fshift : 0u;
            // the pixels of the class split into VP_NQUEUES bands (whole 64-pixel groups, the last band takes the rest)
            for (unsigned q = 0; q <= VP_NQUEUES; q++) L.q_start[q] = q == VP_NQUEUES ? nt : (unsigned)((unsigned long long)(nt / 64u) * q / VP_NQUEUES) * 64u;
            const bool     ldsb = lds_bounds && !cls;
            const unsigned bsz  = ldsb ? VP_BLOCK_LDS : VP_BLOCK;
            unsigned waves  = (L.total_items + 63) / 64;
            unsigned blocks = (waves + (bsz / 64) - 1) / (bsz / 64);
            const bool     both = G.n_light && G.n_general && !light_const;
            unsigned       bpc  = G.blocks_per_cu;
            // what fits a SIMD's 512 vector registers side by side: global majorant 4 x 96 + 2 x 64,
            // local majorant 5 x 96 + ... the light kernel's blocks take what is left as general blocks retire
            // (local majorant, five 96-register general waves per SIMD: the light kernel's workgroups find room as general ones retire,
            // i.e. mostly at the end -- then as many of them as fit)
            if (both && cls) bpc = G.light_blocks_per_cu ? G.light_blocks_per_cu : (G.est == VP_EST_GLOBAL ? 2u : 6u);
            if (!both && cls) bpc = 8u;   // the light kernel alone: 64 registers
            if (both && !cls) bpc = G.general_blocks_per_cu ? G.general_blocks_per_cu : (G.est == VP_EST_GLOBAL ? 4u : 5u);
            // look-ahead batches overlap in pairs: the next batch's approach walk (23 / 47 registers) must find room beside the current
            // batch's integrator -- six of its 72-register workgroups leave 80 registers per SIMD lane, five 80-register ones 112
            if (tgt && !cls && !both) bpc = std::min(bpc, G.est == VP_EST_GLOBAL ? 6u : 5u);
            unsigned cap    = (unsigned)G.num_cu * (ldsb ? 2u : bpc);
            if (blocks > cap) blocks = cap;
            // the light kernel's paths are long and end rarely: its waves leave the tracking loop for the (refill / environment /
            // write) pass less often than the general kernel's do for their collisions
            L.wait_iters = cls ? (G.light_wait_iters ? G.light_wait_iters : (G.est == VP_EST_GLOBAL ? 128u : 64u)) : G.wait_iters;
            if (cls && light_const)
            {
                // the samples of the light class do not depend on the draws in this medium: one constant per pixel, written for
                // every frame (the environment along the camera ray, as for the box-missing pixels)
                ClassTimer ct(1, T.stream);
                launch_miss_fill(S, L, false, T.stream);
                le = hipGetLastError();
                ct.stop();
            }
            else if (cls)
            {
                // the light kernel: beside the general one on the target's auxiliary stream when both classes have work
                hipStream_t ls = T.stream;
                if (both && G.light_overlap)
                {
                    const int ti = T.index;
                    if (!G.aux_stream[ti] && create_internal_stream(&G.aux_stream[ti]) != hipSuccess) { (void)hipGetLastError(); G.aux_stream[ti] = nullptr; }
                    for (int q = 0; q < 2; q++)
                        if (!G.aux_ev[ti][q] && hipEventCreateWithFlags(&G.aux_ev[ti][q], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); G.aux_ev[ti][q] = nullptr; }
                    // the auxiliary stream starts at the fork point recorded above (queue heads zeroed, the previous launch's reduce,
                    // uploads: everything queued on the target stream before the general kernel); it is used only if both the record
                    // and the wait succeeded (waiting on an unrecorded event returns at once)
                    if (G.aux_stream[ti] && G.aux_ev[ti][0] && G.aux_ev[ti][1] && fork_recorded)
                    {
                        if (hipStreamWaitEvent(G.aux_stream[ti], G.aux_ev[ti][0], 0) == hipSuccess) ls = G.aux_stream[ti];
                        else (void)hipGetLastError();
                    }
                }
                ClassTimer ct(1, ls);
                launch_render_light(S, L, G.est, G.rng, G.quant, G.count, (int)blocks, ls);
                le = hipGetLastError();
                ct.stop();
                if (ls != T.stream && le == hipSuccess)
                {
                    // the target stream goes on (end-of-launch event, add-kernel) only when the light kernel is done too
                    if (hipEventRecord(G.aux_ev[T.index][1], ls) != hipSuccess || hipStreamWaitEvent(T.stream, G.aux_ev[T.index][1], 0) != hipSuccess)
                        le = hipGetLastError();
                }
            }
            else
            {
                ClassTimer ct(0, T.stream);
                L.approach = 0;
                G.last_approach = 0;
                const bool aux_ok = !appr_aux_needed || G.d_appr_aux[T.index] != nullptr;
                if (appr_aux_needed) L.approach_aux = G.d_appr_aux[T.index];
                if (approach && aux_ok && L.stage && f <= 65535)
                {
                    L.approach       = approach_thr ? 2u : 1u;
                    L.approach_steps = G.approach_steps;
                    L.approach_fshift = 0;
                    while (L.approach_fshift < G.approach_fshift_max && (2u << L.approach_fshift) <= (unsigned)f) L.approach_fshift++;
                    launch_approach(S, L, G.est, G.rng, G.quant, T.stream);
                    le = hipGetLastError();
                    G.last_approach = (int)L.approach;
                    // the helper workgroups of the LDS-table kernel (auxiliary stream, below) read the staging slots as well: their
                    // fork point moves behind the walk
                    if (lds_helper && fork_recorded && le == hipSuccess)
                    {
                        fork_recorded = hipEventRecord(G.aux_ev[T.index][0], T.stream) == hipSuccess;
                        if (!fork_recorded) (void)hipGetLastError();
                    }
                }
                if (le == hipSuccess)
                {
                    launch_render(S, L, G.est, G.rng, G.quant, G.count, lds_bounds, G.env_mis, G.trk, (int)blocks, T.stream);
                    le = hipGetLastError();
                }
                // The LDS-table kernel holds 2 x 64 KiB of a CU's LDS with 2 x 512 threads: four waves per SIMD, where the
                // registers would allow five.  The fifth comes from the SAME kernel without the LDS stage (the brick table read
                // from global memory), one 256-thread workgroup per CU beside it on the auxiliary stream, drawing from the same
                // sample queues: a sample is computed by whichever wave takes its chunk, with the same bits.
                if (lds_helper && ldsb && le == hipSuccess && fork_recorded && blocks >= cap)
                {
                    const int ti = T.index;
                    if (!G.aux_stream[ti] && create_internal_stream(&G.aux_stream[ti]) != hipSuccess) { (void)hipGetLastError(); G.aux_stream[ti] = nullptr; }
                    if (!G.aux_ev[ti][1] && hipEventCreateWithFlags(&G.aux_ev[ti][1], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); G.aux_ev[ti][1] = nullptr; }
                    if (G.aux_stream[ti] && G.aux_ev[ti][1] && hipStreamWaitEvent(G.aux_stream[ti], G.aux_ev[ti][0], 0) == hipSuccess)
                    {
                        launch_render(S, L, G.est, G.rng, G.quant, G.count, false, G.env_mis, G.trk, G.num_cu, G.aux_stream[ti]);
                        le = hipGetLastError();
                        if (le == hipSuccess && (hipEventRecord(G.aux_ev[ti][1], G.aux_stream[ti]) != hipSuccess || hipStreamWaitEvent(T.stream, G.aux_ev[ti][1], 0) != hipSuccess))
                            le = hipGetLastError();
                    }
                    else (void)hipGetLastError();
                }
                ct.stop();
            }
        }
        if (G.n_miss && le == hipSuccess)
        {
            // the pixels whose camera ray misses the box: one constant per pixel, written for every frame (miss_fill_k)
            L.pixels      = G.d_tiles + G.n_general + G.n_light;
            L.nslots      = G.n_miss;
            L.slot_base   = G.n_general + G.n_light;
            L.total_items = 0;
            ClassTimer ct(2, T.stream);
            launch_miss_fill(S, L, G.est != VP_EST_GLOBAL, T.stream);
            le = hipGetLastError();
            ct.stop();
        }
        timed = timed && le == hipSuccess && hipEventRecord(e1, T.stream) == hipSuccess;
        G.timed_n++;
        if (timed) { G.events.emplace_back(e0, e1); trim_events(); }
        else { put_event(e0); put_event(e1); }
        if (le != hipSuccess) return fail(VP_E_NODEVICE, "render launch -> %s", hipGetErrorString(le));
        // for the add-kernel: all tiles of the rank
        L.pixels = G.d_tiles; L.nslots = (unsigned)per_frame; L.slot_base = 0;
        if (L.stage && !stage_only)
        {
            launch_reduce(L, T.stream);
            HIPCHK(hipGetLastError());
        }
        done += f;
    }
    return VP_OK;
}

// everything a sample's value depends on besides (x, y, frame): compared bytewise between render_kernel calls
void render_key(const Param* p, std::vector<unsigned char>& key)
{
    struct K { SceneDev S; Param P; int est, rng, linear, quant, mis, trk; unsigned k0, k1, rank, world; unsigned long long epoch; };
    key.assign(sizeof(K), 0);
    K* k = reinterpret_cast<K*>(key.data());
    memcpy(&k->S, &G.S, sizeof(SceneDev));
    memcpy(&k->P, p, sizeof(Param));
    k->est = G.est; k->rng = G.rng; k->linear = G.linear; k->quant = G.quant; k->mis = G.env_mis; k->trk = G.trk;
    k->k0 = G.key0; k->k1 = G.key1; k->rank = G.rank; k->world = G.world; k->epoch = G.epoch;
}

// render_kernel with frame look-ahead.  The reference host calls render_kernel once per frame and synchronises
// (host.cpp:631-632); a one-frame launch is bound by its longest path (about 14 ms for 0.48 M samples, 12x off the
// batched rate).  A sample is a pure function of (x, y, frame, scene), so when the host asks for frame f right after
// f-1 with nothing changed, frames f..f+n-1 are rendered in ONE launch into a staging slot (n = 32, 64, ... la_max)
// and only frame f is added to the caller's accumulator; the next calls find their frame staged and
// just add it.  Two slots are kept in flight on two streams -- the successor of a batch (twice its size, up to la_max) is queued
// when the batch's first frame is asked for --, so the tail of one batch (its
// deepest paths) overlaps the body of the next.  Any state change drops the staged frames.  Bit-identical to one
// launch per frame.
// Batches in flight whose frames nobody will ask for any more (a setter, a camera move, new device contents, a frame jump) are told
// to stop -- unless a frame of the batch has already been handed out (its add-kernel sits on the caller's stream behind the
// batch's completion event and needs that frame whole: such a batch runs to its end).  The slot's cancel word gets the batch's
// number: its render_k takes no further chunk (it asks at every chunk, in every launch of a multi-launch batch), approach kernels
// that have not started yet return at once, and the waves of render_k<..., CANCEL> give up their paths at their next look at the
// word (every eighth event visit; nothing reads what a cancelled batch has staged): a camera move waits 0.3 ms instead of the rest
// of the batch or its deepest paths.  Numbers only grow, so nothing has to be re-armed and a cancel can neither be lost nor reach a
// later batch.  Written from a stream of the highest priority (State::ctrl_stream: a hardware queue of its own);
// results are discarded, so nothing depends on where the cut falls.  Does not wait.  Whether the slot's frames are still VALID
// for serving does not matter here: a miss invalidates the slots first and finds the batches running all the same.
bool la_cancel_running()
{
    bool any = false;
    bool cancel[2] = {false, false};
    for (int si = 0; si < 2 && G.la_cancel; si++)
        // (the batch `done` stands for -- launched_seq -- not the slot's newest number: this runs INSIDE the launch of a slot's next
        // batch too, when a table has to be rebuilt first, and the number of a batch about to start must not get into its cancel word)
        if (G.la[si].stream && G.la[si].done && !G.la[si].touched && G.la[si].launched_seq && G.la[si].cancel_seq != G.la[si].launched_seq &&
            hipEventQuery(G.la[si].done) == hipErrorNotReady)
            cancel[si] = any = true;
    (void)hipGetLastError();
    if (!any) return false;
    if (!G.ctrl_stream)
    {
        int lo = 0, hi = 0;   // (numerically lower = higher priority)
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        if (hipStreamCreateWithPriority(&G.ctrl_stream, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); G.ctrl_stream = nullptr; }
    }
    if (!G.ctrl_stream) return false;
    for (int si = 0; si < 2; si++)
        if (cancel[si])
        {
            G.la_cancelled++;
            G.la[si].cancel_seq = G.la[si].launched_seq;   // (told once)
            (void)hipMemsetD32Async((hipDeviceptr_t)(G.d_cancel + (si + 1)), (int)G.la[si].launched_seq, 1, G.ctrl_stream);
        }
    (void)hipGetLastError();
    return true;
}
int la_quiesce()
{
    // every caller is about to change what batches in flight read (tables, lists, the volume): stop them and wait
    const bool any = la_cancel_running();
    if (G.la_spec_unserved) { G.la_habit = false; G.la_spec_unserved = false; }   // speculated and nobody came: stop speculating until a real hit
    static const bool dbg = getenv("VP_DEBUG_QUIESCE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& s : G.la)
    {
        if (s.stream) HIPCHK(hipStreamSynchronize(s.stream));
        s.valid = false;
    }
    if (any && G.ctrl_stream) HIPCHK(hipStreamSynchronize(G.ctrl_stream));
    if (dbg && any)
        fprintf(stderr, "[vp] quiesce: batches stopped and drained after %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return VP_OK;
}
int la_render_slot(int si, vp_float4* d_out, int first, int n, const Param* p, const std::vector<unsigned char>& key)
{
    auto& s = G.la[si];
    if (!s.stream)
    {
        HIPCHK(create_internal_stream(&s.stream));   // (lowest priority: see there)
    }
    if (!s.done) HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    s.valid = false;
    // `touched` guards the batch that is being replaced as well: it may still be running on this stream (a miss invalidates a slot
    // without waiting for it) and shares the slot's sample queues with the new one, so a cancel aimed at the new batch would cut
    // the old one short -- while an add-kernel for one of its frames may still be pending.  The mark is dropped only once the old
    // batch has completed.
    if (!s.done || hipEventQuery(s.done) == hipSuccess) s.touched = false;
    else (void)hipGetLastError();
    // after everything queued on the caller's stream: uploads the scene depends on, and add-kernels still reading this slot
    hipEvent_t ev = get_event();
    if (!ev) return fail(VP_E_NODEVICE, "hipEventCreate failed");
    HIPCHK(hipEventRecord(ev, G.stream));
    HIPCHK(hipStreamWaitEvent(s.stream, ev, 0));
    put_event(ev);
    const Target t = {s.stream, &s.buf, &s.bytes, G.d_queue + 2 * kQueueWords * (si + 1), si + 1};
    G.la_launched++;
    G.batch_seq[si + 1]++;   // (numbers only grow: a cancel of an earlier batch of this slot can never reach this one)
    int rc = do_render(d_out, first, n, p, true, &t);
    if (rc) return rc;
    HIPCHK(hipEventRecord(s.done, s.stream));
    s.launched_seq = G.batch_seq[si + 1];
    s.const_from = G.last_const_from;
    s.valid = true; s.first = first; s.count = n; s.key = key;
    return VP_OK;
}
int la_limit(int first, int n, size_t per_frame, size_t have_bytes)
{
    if (G.est == VP_EST_DECOMP && !G.S.opacity) n = first <= 10 ? std::min(n, 11 - first) : 0;  // quirk Q5 needs the opacity volume
    if (per_frame && n > 0) n = (int)std::min<size_t>((size_t)n, stage_frames_cap(per_frame, have_bytes));
    return n;
}
int serve_frame(vp_float4* d_out, int frame, const Param* p)
{
    if (G.la_max <= 1 || G.count || !p) return do_render(d_out, frame, 1, p);
    int rc = ensure_device();
    if (rc) return rc;
    std::vector<unsigned char> key;
    render_key(p, key);
    const Shard sh = shard_of(p);
    const size_t per_frame = sh.per_frame;
    for (int si = 0; si < 2 && d_out && per_frame; si++)
    {
        auto& s = G.la[si];
        if (!(s.valid && s.key == key && frame >= s.first && frame < s.first + s.count)) continue;
        // entering a full-size batch: the other slot is free, start the batch after this one.  Queued BEFORE this
        // frame's wait on its own batch, so that the new batch depends only on work already on the caller's stream
        // (the add-kernels that read the other slot) and can fill the tail of the batch now finishing.
        auto& o = G.la[si ^ 1];
        const int next = s.first + s.count;
        // (during the ramp as well: the batch after a batch of n is one of 2n, queued when the first frame of this one is asked for)
        if (frame == s.first && s.count >= G.la_overlap_from && !(o.valid && o.key == key && o.first == next))
        {
            // twice this one, up to la_max -- but beyond la_floor frames never more than half of what the run has accumulated by then: a
            // batch is delivered whole (its first frame waits for its last), so a big one early in a run is a long wait for few frames,
            // and most of it is thrown away when the camera moves on after a few hundred frames
            int want = std::min(s.count * 2, G.la_max);
            while (want > G.la_floor && want > (next - G.la_run_first) / G.la_div) want >>= 1;
            int n = la_limit(next, want, per_frame, o.bytes);
            if (n > 1 && la_render_slot(si ^ 1, d_out, next, n, p, key)) G.la[si ^ 1].valid = false;  // best effort
            // (that launch may have rebuilt a table and quiesced -- stopping THIS batch, whose frames are then not to be
            // served: today every table key equals the running batch's, but nothing else enforces it)
            if (!s.valid || (s.cancel_seq == s.launched_seq && s.launched_seq)) break;
        }
        // hit: add the staged frame once its batch is rendered
        // (a batch that has completed needs no wait queued for it)
        if (hipEventQuery(s.done) != hipSuccess)
        {
            (void)hipGetLastError();
            s.touched = true;
            HIPCHK(hipStreamWaitEvent(G.stream, s.done, 0));
        }
        LaunchDev L = {};
        memcpy(&L.P, p, sizeof(Param));
        L.pixels = G.d_tiles; L.nslots = (unsigned)per_frame; L.stage_stride = (unsigned)per_frame;
        L.out = (float4*)d_out;
        L.stage = s.buf + (size_t)(frame - s.first) * per_frame;
        L.const_from = s.const_from; L.stage_const = s.buf;
        L.nframes = 1;
        launch_reduce(L, G.stream);
        HIPCHK(hipGetLastError());
        G.la_last = frame;
        G.la_habit = true;   // this caller asks for consecutive frames
        G.la_spec_unserved = false;
        return VP_OK;
    }
    // miss: how far ahead?  only when this call continues the previous one
    const bool same = key == G.la_key;
    // (the first frame of a run alone: it is what the caller waits for after a camera move, ~10 ms of its deepest paths; the call
    // after it starts the ramp at la_ramp_from frames -- a batch of up to ~32 frames lasts as long as one frame, its deepest path --
    // and every batch has its successor, twice its size, queued behind it)
    int n = (same && frame == G.la_last + 1) ? std::min(std::max(G.la_prev_n * 2, G.la_ramp_from), std::min(G.la_max, G.la_floor)) : 1;
    if (n == 1) G.la_run_first = frame;
    if (n > 1) n = std::max(la_limit(frame, n, per_frame, G.la[0].bytes), 1);
    G.la_key = key; G.la_last = frame; G.la_prev_n = n;
    // The habit decays: a speculative batch none of whose frames was asked for -- an interactive drag: every call is frame 0
    // of a new camera -- cost the first frame after the move 2 ms and returned nothing; no more of them until a staged frame is served again
    if (G.la_spec_unserved) { G.la_habit = false; G.la_spec_unserved = false; }
    (void)la_cancel_running();   // (what runs ahead for frames that will not be asked for: out of this frame's way)
    G.la[0].valid = G.la[1].valid = false;
    if (n <= 1 || !per_frame || !d_out)
    {
        // The first frame of a run (after a camera move, a setter, a frame jump) is rendered alone: it is what the caller waits for.  A
        // caller that has been asking for consecutive frames will ask for the next ones: the first batch of the ramp is queued on a slot
        // BEFORE this frame's launch (the slot's stream waits for what is on the caller's stream now), so the two run side by side --
        // both are bound by their deepest paths, not by the chip.
        if (G.la_habit && G.la_speculate && per_frame && d_out && n == 1)
        {
            const int m = la_limit(frame + 1, std::min(G.la_ramp_from, std::min(G.la_max, G.la_floor)), per_frame, G.la[0].bytes);
            if (m > 1)
            {
                if (la_render_slot(0, d_out, frame + 1, m, p, key)) G.la[0].valid = false;   // best effort
                else G.la_spec_unserved = true;
            }
        }
        return do_render(d_out, frame, 1, p);
    }
    // The look-ahead is an optimisation the caller never asked for: if the batch cannot be rendered (no memory for its
    // staging slot, a stream that cannot be created) this frame is rendered alone, exactly as without look-ahead, and
    // the batch size starts over.
    if (la_render_slot(0, d_out, frame, n, p, key))
    {
        G.la[0].valid = false; G.la_prev_n = 0;
        return do_render(d_out, frame, 1, p);
    }
    return serve_frame(d_out, frame, p);  // now a hit (which also starts the following batch once n is full size)
}
}  // namespace

// =============================================================================== Part 1
extern "C" {

void init_cuda(void* h_volume, vp_extent volumeSize, bool quantized, const vp_float3* boxmin, const vp_float3* boxmax)
{
    if (!h_volume)
    {
        fprintf(stderr, "cannot init without host volume\n");  // kernel.cu:360-364
        exit(1);
    }
    if (do_init_volume(h_volume, volumeSize, quantized, boxmin, boxmax)) die("init_cuda");
}

void set_texture_filter_mode(bool bLinearFilter) { G.linear = bLinearFilter; G.S.linear = bLinearFilter ? 1 : 0; }

void free_cuda_buffers(void)
{
    if (!G.dev_ready) return;
    (void)hipStreamSynchronize(G.stream);
    if (free_volume()) die("free_cuda_buffers");
}

void precompute_opacity(const float* light_dir)
{
    if (do_opacity(light_dir)) die("precompute_opacity");
}

void init_envmap(const vp_float4* HDRmap, int width, int height)
{
    if (do_envmap(HDRmap, width, height)) die("init_envmap");
}

void free_envmap(void)
{
    if (!G.have_env) return;
    G.epoch++;
    (void)la_quiesce();
    (void)hipStreamSynchronize(G.stream);
    (void)hipFree(G.d_env);
    (void)hipFree(G.d_env_cdf_x);
    (void)hipFree(G.d_env_cdf_y);
    G.d_env = nullptr; G.S.env = nullptr; G.env_w = G.env_h = 0; G.have_env = false;
    G.d_env_cdf_x = G.d_env_cdf_y = nullptr; G.S.env_cdf_x = G.S.env_cdf_y = nullptr; G.env_tables = false;
}

void set_sun(float* sun_dir, float* sun_power)
{
    // kernel.cu:1269-1283: disc radiance kept for the depth-0 sun test, directional power = p * pi * (0.45/94)^2
    float r = (float)(0.45 / (double)94.0f);
    float f = 3.1415926535897932384626422832795028841971f * (r * r);
    for (int i = 0; i < 3; i++)
    {
        G.S.sun_dir[i]   = sun_dir[i];
        G.S.sun_orig[i]  = sun_power[i];
        G.S.sun_power[i] = sun_power[i] * f;
    }
    G.have_sun = true;
}

void copy_inv_view_matrix(float* invViewMatrix, size_t sizeofMatrix)
{
    memcpy(G.S.cam, invViewMatrix, sizeofMatrix < sizeof G.S.cam ? sizeofMatrix : sizeof G.S.cam);
    G.have_cam = true;
}
void copy_inv_model_matrix(float* invModelMatrix, size_t sizeofMatrix)
{
    // kept for interface parity; unused while USE_MODEL_TRANSFORM=0 (kernel.cu:32)
    memcpy(G.inv_model, invModelMatrix, sizeofMatrix < sizeof G.inv_model ? sizeofMatrix : sizeof G.inv_model);
}

void init_rng(vp_dim3, vp_dim3, int, int) {}  // kernel.cu:2330
void free_rng(void) {}                         // kernel.cu:2331

void render_kernel(vp_dim3, vp_dim3, vp_float4* d_output, int spp, const Param& p)
{
    if (serve_frame(d_output, spp, &p)) die("render_kernel");
}

void scale(vp_float4* dst, vp_float4* src, int size, float s)
{
    if (ensure_device()) die("scale");
    launch_scale((float4*)dst, (const float4*)src, size, s, G.stream);
}
void gamma_correct(vp_float4* dst, vp_float4* src, int size, float s, float gamma)
{
    if (ensure_device()) die("gamma_correct");
    launch_gamma((float4*)dst, (const float4*)src, size, s, 1.0f / gamma, G.stream);  // kernel.cu:2361
}

// =============================================================================== Part 2
const char* vp_last_error(void) { return G.err.c_str(); }
const char* vp_version(void) { return "volpath_hip 0.3 (gfx950)"; }
int vp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int vp_set_device(int device)
{
    if (G.dev_ready && device != G.device)
        return fail(VP_E_STATE, "this context is bound to device %d; use vp_ctx_create(%d) for another GPU", G.device, device);
    G.device = device;
    return ensure_device();
}

// ---- contexts
struct vp_ctx { State st; };
vp_ctx* vp_ctx_create(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
    {
        fail(VP_E_NODEVICE, "vp_ctx_create: device %d not visible (%d devices)", device, n);
        return nullptr;
    }
    vp_ctx* c = new vp_ctx();
    c->st.device = device;
    State* prev = t_current;
    t_current   = &c->st;
    int rc      = ensure_device();
    std::string why = c->st.err;
    t_current   = prev;
    if (rc)
    {
        delete c;
        fail(rc, "vp_ctx_create(%d): %s", device, why.c_str());
        return nullptr;
    }
    return c;
}
int vp_ctx_set_current(vp_ctx* ctx)
{
    t_current = ctx ? &ctx->st : nullptr;
    if (G.dev_ready) HIPCHK(hipSetDevice(G.device));
    return VP_OK;
}
vp_ctx* vp_ctx_get_current(void) { return t_current ? reinterpret_cast<vp_ctx*>(t_current) : nullptr; }
int vp_ctx_device(void) { return G.device; }
int vp_ctx_destroy(vp_ctx* ctx)
{
    if (!ctx) return VP_OK;
    State* prev = t_current;
    t_current   = &ctx->st;
    State& D    = ctx->st;
    int rc = VP_OK;
    if (D.dev_ready)
    {
        (void)hipSetDevice(D.device);
        (void)hipStreamSynchronize(D.stream);
        (void)la_quiesce();
        rc = free_volume();
        free_envmap();
        for (auto& sl : D.la)
        {
            if (sl.buf) (void)hipFree(sl.buf);
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
        }
        for (auto& ev : D.events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        for (auto& ev : D.class_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
        for (auto e : D.event_pool) (void)hipEventDestroy(e);
        if (D.d_stage) (void)hipFree(D.d_stage);
        if (D.d_crawl) (void)hipFree(D.d_crawl);
        if (D.d_thr) (void)hipFree(D.d_thr);
        if (D.d_sunclip) (void)hipFree(D.d_sunclip);
        if (D.d_light_flag) (void)hipFree(D.d_light_flag);
        if (D.d_tiles) (void)hipFree(D.d_tiles);
        if (D.d_tile_rows) (void)hipFree(D.d_tile_rows);
        if (D.d_tile_scratch) (void)hipFree(D.d_tile_scratch);
        for (int i = 0; i < 3; i++)
        {
            if (D.aux_stream[i]) { (void)hipStreamSynchronize(D.aux_stream[i]); (void)hipStreamDestroy(D.aux_stream[i]); }
            if (D.d_appr_aux[i]) (void)hipFree(D.d_appr_aux[i]);
            for (int q = 0; q < 2; q++) if (D.aux_ev[i][q]) (void)hipEventDestroy(D.aux_ev[i][q]);
        }
        if (D.d_queue) (void)hipFree(D.d_queue);
        if (D.d_counters) (void)hipFree(D.d_counters);
        if (D.ctrl_stream) (void)hipStreamDestroy(D.ctrl_stream);
        if (D.d_cancel) (void)hipFree(D.d_cancel);
        if (D.own_stream) (void)hipStreamDestroy(D.own_stream);
    }
    t_current = (prev == &ctx->st) ? nullptr : prev;
    delete ctx;
    if (G.dev_ready) (void)hipSetDevice(G.device);
    return rc;
}
int vp_accumulate(vp_float4* dst, const vp_float4* src, size_t n)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!dst || !src) return fail(VP_E_ARG, "vp_accumulate: null pointer");
    if (n) launch_accumulate((float4*)dst, (const float4*)src, n, G.stream);
    HIPCHK(hipGetLastError());
    return VP_OK;
}
int vp_tile_owner(unsigned tx, unsigned ty, int world)
{
    if (world < 1) return -1;
    return (int)((tx + vp::tile_row_shift(ty, (unsigned)world)) % (unsigned)world);
}
int vp_set_stream(void* s)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(G.stream));
    if (la_quiesce()) return VP_E_NODEVICE;
    G.stream = s ? (hipStream_t)s : G.own_stream;
    return VP_OK;
}
void* vp_get_stream(void)
{
    if (ensure_device()) return nullptr;
    return (void*)G.stream;
}
int vp_synchronize(void)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(G.stream));
    return VP_OK;
}
int vp_set_estimator(int est)
{
    if (est != VP_EST_GLOBAL && est != VP_EST_DECOMP && est != VP_EST_BOUNDED) return fail(VP_E_ARG, "unknown estimator %d", est);
    G.est = est;
    return VP_OK;
}
int vp_set_rng(int mode, uint32_t k0, uint32_t k1)
{
    if (mode != VP_RNG_SAMPLERH && mode != VP_RNG_PHILOX && mode != VP_RNG_PHILOX7) return fail(VP_E_ARG, "unknown rng %d", mode);
    G.rng = mode; G.key0 = k0; G.key1 = k1;
    return VP_OK;
}
int vp_set_exit_flights(int mode)
{
    if (mode < 0 || mode > 2) return fail(VP_E_ARG, "exit flights: 0 off, 1 global-majorant estimator (default), 2 every estimator that has them");
    int rc = ensure_device();   // (first: it parses VP_NO_EXIT / VP_EXIT_LOCAL, which a later call must not override -- ADVICE r4)
    if (rc) return rc;
    G.use_exit = mode != 0; G.exit_local = mode == 2;
    if (G.use_exit && G.have_volume && !G.d_exit && G.d_danger)
    {
        // switched on after a volume was initialised without the table: build it now (the header promises the three modes unconditionally)
        if (la_quiesce()) return VP_E_NODEVICE;
        const size_t n = (size_t)G.S.nx * G.S.ny * G.S.nz;
        if (hipMalloc((void**)&G.d_exit, 3 * n) == hipSuccess)
        {
            launch_exit_table(G.d_danger, G.d_exit, G.S.nx, G.S.ny, G.S.nz, G.stream);
            HIPCHK(hipGetLastError());
        }
        else { (void)hipGetLastError(); G.d_exit = nullptr; }   // none: every path walks to the box exit, same bits
    }
    return VP_OK;
}
int vp_set_tracking(int mode)
{
    if (mode != VP_TRACK_SPECTRAL && mode != VP_TRACK_SCALAR && mode != VP_TRACK_MULTI_CHANNEL)
        return fail(VP_E_ARG, "unknown tracking mode %d", mode);
    G.trk = mode;
    return VP_OK;
}
int vp_set_lookahead(int max_frames)
{
    if (max_frames < 0 || max_frames > 4096) return fail(VP_E_ARG, "look-ahead of %d frames out of range [0,4096]", max_frames);
    if (la_quiesce()) return VP_E_NODEVICE;
    G.la_max = max_frames; G.la_prev_n = 0;
    return VP_OK;
}
int vp_set_envmap_sampling(int mode)
{
    if (mode != VP_ENV_PASSIVE && mode != VP_ENV_MIS) return fail(VP_E_ARG, "unknown environment sampling mode %d", mode);
    int rc = ensure_device();
    if (rc) return rc;
    G.env_mis = mode == VP_ENV_MIS;
    if (G.env_mis) return build_env_tables();
    return VP_OK;
}
int vp_get_env_tables(float* cdf_y, float* cdf_x, float* pdfnorm_alt)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!G.env_tables) return fail(VP_E_STATE, "no environment tables: vp_set_envmap_sampling(VP_ENV_MIS) and init_envmap first");
    HIPCHK(hipStreamSynchronize(G.stream));
    if (cdf_y) HIPCHK(hipMemcpy(cdf_y, G.d_env_cdf_y, (size_t)G.env_h * sizeof(float), hipMemcpyDeviceToHost));
    if (cdf_x) HIPCHK(hipMemcpy(cdf_x, G.d_env_cdf_x, (size_t)G.env_w * G.env_h * sizeof(float), hipMemcpyDeviceToHost));
    if (pdfnorm_alt) *pdfnorm_alt = G.S.env_pdfnorm_alt;
    return VP_OK;
}
int vp_set_bound_brick(int brick)
{
    if (brick < 1 || brick > 64 || (brick & (brick - 1))) return fail(VP_E_ARG, "brick edge must be a power of two in [1,64]");
    G.brick_next = brick;
    return VP_OK;
}
int vp_set_shard(int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) return fail(VP_E_ARG, "bad shard %d/%d", rank, world);
    G.rank = (unsigned)rank; G.world = (unsigned)world;
    return VP_OK;
}
int vp_render_frames(vp_float4* d_output, int first_frame, int n_frames, const Param* p)
{
    return do_render(d_output, first_frame, n_frames, p);
}
int vp_enable_counters(int on) { G.count = on != 0; return VP_OK; }
int vp_read_counters(vp_counters* out, int reset)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(G.stream));
    unsigned long long h[kCounterWords];
    HIPCHK(hipMemcpy(h, G.d_counters, sizeof h, hipMemcpyDeviceToHost));
    if (out)
    {
        memset(out, 0, sizeof *out);
        out->samples = h[0]; out->density_lookups = h[1]; out->density_loads = h[12]; out->bound_lookups = h[2];
        out->opacity_lookups = h[3]; out->env_lookups = h[4]; out->scatters = h[5];
        if (getenv("VP_DEBUG_COUNTERS") && h[6]) fprintf(stderr, "[vp] wave-iterations %llu, active lane-steps %llu (%.1f per iteration), slow-path visits %llu (every %.1f iterations), shadow lane-steps %llu; wave cycles: slow path %llu, fast loop %llu (%.1f%% slow, %.0f cycles per visit, %.0f per step)\n", h[6], h[7], h[6] ? (double)h[7] / h[6] : 0.0, h[8], h[8] ? (double)h[6] / h[8] : 0.0, h[9], h[10], h[11], 100.0 * h[10] / (double)(h[10] + h[11] + 1), h[8] ? (double)h[10] / h[8] : 0.0, h[6] ? (double)h[11] / h[6] : 0.0);
    }
        if (getenv("VP_DEBUG_COUNTERS"))
        {
            static const char* names[15] = {"setup", "half-step", "lookup+collision", "segment/ray end", "scatter", "nee", "phase", "background", "write", "refill", "global set-up", "fetch", "zero fetch (path)", "zero fetch (shadow)", "exit test"};
            fprintf(stderr, "[vp] exit flights: %llu tests, %llu paths ended; %llu null collisions in empty space on flights that WALKED out of the box (global majorant)\n", h[13], h[15], h[14]);
            fprintf(stderr, "[vp] block: wave executions, lanes per execution (of 64)\n");
            for (int b = 0; b < 15; b++)
                if (h[16 + 2 * b]) fprintf(stderr, "[vp]   %-18s %14llu  %5.1f\n", names[b], h[16 + 2 * b], (double)h[17 + 2 * b] / (double)h[16 + 2 * b]);
            static const char* hn[3] = {"scatter", "segment/ray end", "setup"};
            for (int q = 0; q < 3; q++)
            {
                unsigned long long tot = 0;
                for (int k = 0; k < 8; k++) tot += h[48 + 8 * q + k];
                if (!tot) continue;
                fprintf(stderr, "[vp]   executions of %-16s by lanes 1-8 .. 57-64 (%%):", hn[q]);
                for (int k = 0; k < 8; k++) fprintf(stderr, " %5.1f", 100.0 * (double)h[48 + 8 * q + k] / (double)tot);
                fprintf(stderr, "\n");
            }
        }
    if (reset) HIPCHK(hipMemset(G.d_counters, 0, sizeof h));
    return VP_OK;
}
int vp_render_time_ms(double* total_ms, int* launches, int reset)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(G.stream));
    for (auto& sl : G.la)  // look-ahead batches still in flight are launches too
        if (sl.stream) HIPCHK(hipStreamSynchronize(sl.stream));
    double tot = G.timed_ms;
    for (auto& ev : G.events)
    {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = G.timed_n;
    if (reset)
    {
        for (auto& ev : G.events) { put_event(ev.first); put_event(ev.second); }
        G.events.clear();
        G.timed_ms = 0.0; G.timed_n = 0;
    }
    return VP_OK;
}
int vp_last_approach_mode(void) { return G.last_approach; }
int vp_last_light_const(void) { return G.last_light_const; }
int vp_lookahead_stats(unsigned* launched, unsigned* cancelled_in_flight)
{
    if (launched) *launched = G.la_launched;
    if (cancelled_in_flight) *cancelled_in_flight = G.la_cancelled;
    return VP_OK;
}
int vp_render_class_time_ms(double ms[3], unsigned pixels[3], int reset)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(G.stream));
    for (auto& sl : G.la)
        if (sl.stream) HIPCHK(hipStreamSynchronize(sl.stream));
    for (int i = 0; i < 3; i++)
        if (G.aux_stream[i]) HIPCHK(hipStreamSynchronize(G.aux_stream[i]));
    double tot[3] = {G.class_ms[0], G.class_ms[1], G.class_ms[2]};
    for (auto& ev : G.class_events)
    {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, ev.a, ev.b));
        tot[ev.cls] += t;
    }
    if (ms) for (int i = 0; i < 3; i++) ms[i] = tot[i];
    if (pixels) { pixels[0] = G.n_general; pixels[1] = G.n_light; pixels[2] = G.n_miss; }
    if (reset)
    {
        for (auto& ev : G.class_events) { put_event(ev.a); put_event(ev.b); }
        G.class_events.clear();
        G.class_ms[0] = G.class_ms[1] = G.class_ms[2] = 0.0;
    }
    return VP_OK;
}
int vp_get_pixel_lists(const Param* p, uint32_t* dst, size_t count, unsigned counts[3])
{
    int rc = vp_prepare(p);
    if (rc) return rc;
    const size_t n = (size_t)G.n_general + G.n_light + G.n_miss;
    if (counts) { counts[0] = G.n_general; counts[1] = G.n_light; counts[2] = G.n_miss; }
    if (dst)
    {
        if (count < n) return fail(VP_E_ARG, "pixel lists hold %zu entries", n);
        if (n) HIPCHK(hipMemcpy(dst, G.d_tiles, n * sizeof(unsigned), hipMemcpyDeviceToHost));
    }
    return VP_OK;
}
int vp_prepare(const Param* p)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p) return fail(VP_E_ARG, "vp_prepare: null Param");
    if (!G.have_volume || !G.have_cam) return fail(VP_E_STATE, "vp_prepare needs a volume and a camera");
    if (p->width == 0 || p->height == 0 || p->width > 65535 || p->height > 65535) return fail(VP_E_ARG, "image %ux%u out of range", p->width, p->height);
    const Shard sh = shard_of(p);
    if (sh.per_frame == 0)
    {
        // a shard without a tile (more ranks than tiles): empty lists
        G.n_general = G.n_light = G.n_miss = 0; G.tiles_key.clear();
        return VP_OK;
    }
    const float4* table = nullptr;
    rc = ensure_crawl_table(p, &table);
    if (rc) return rc;
    rc = ensure_pixel_lists(p, table, sh);
    if (rc) return rc;
    if (G.have_sun)
    {
        const unsigned short* sc = nullptr; float ds = 0.0f;
        rc = ensure_sun_clip(&sc, &ds);
        if (rc) return rc;
    }
    if (G.est == VP_EST_GLOBAL && G.n_light)
    {
        const float* thr = nullptr;
        rc = ensure_thr_table(p, &thr);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(G.stream));
    return VP_OK;
}
int vp_reserve_frames(const Param* p, int nframes)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || nframes <= 0) return fail(VP_E_ARG, "vp_reserve_frames: bad arguments");
    if (p->width == 0 || p->height == 0 || p->width > 65535 || p->height > 65535) return fail(VP_E_ARG, "image %ux%u out of range", p->width, p->height);
    const Shard sh = shard_of(p);
    if (sh.per_frame == 0 || nframes == 1) return VP_OK;
    // what do_render would allocate for the first launch of such a job (a one-frame call accumulates directly and stages nothing)
    const size_t f    = std::min<size_t>((size_t)nframes, stage_frames_cap(sh.per_frame, G.stage_bytes));
    const size_t need = sh.per_frame * f * sizeof(float4);
    // (decomposition estimator: the stream's state beside each staging slot of the approach kernel's hand-over, do_render)
    const size_t need4 = (G.est == VP_EST_DECOMP && G.use_approach && G.use_approach_local) ? sh.per_frame * f * sizeof(uint2) : 0;
    if (need4 > G.appr_aux_bytes[0])
    {
        HIPCHK(hipStreamSynchronize(G.stream));
        if (G.d_appr_aux[0]) HIPCHK(hipFree(G.d_appr_aux[0]));
        G.d_appr_aux[0] = nullptr; G.appr_aux_bytes[0] = 0;
        if (hipMalloc((void**)&G.d_appr_aux[0], need4) != hipSuccess) { (void)hipGetLastError(); G.d_appr_aux[0] = nullptr; }
        else G.appr_aux_bytes[0] = need4;
    }
    if (need <= G.stage_bytes) return VP_OK;
    if (la_quiesce()) return VP_E_NODEVICE;
    HIPCHK(hipStreamSynchronize(G.stream));
    if (G.d_stage) HIPCHK(hipFree(G.d_stage));
    G.d_stage = nullptr; G.stage_bytes = 0;
    if (hipMalloc((void**)&G.d_stage, need) != hipSuccess)
    {
        (void)hipGetLastError();
        G.d_stage = nullptr;
        return VP_OK;   // the render call will stage smaller batches: same bits
    }
    G.stage_bytes = need;
    return VP_OK;
}
int vp_get_bound_table(void* dst, size_t bytes, int* bnx, int* bny, int* bnz, int* brick, int* radius)
{
    if (!G.have_volume) return fail(VP_E_STATE, "no volume");
    size_t need = (size_t)G.S.bnx * G.S.bny * G.S.bnz * (G.quant ? 2 : 8);
    if (bnx) *bnx = G.S.bnx;
    if (bny) *bny = G.S.bny;
    if (bnz) *bnz = G.S.bnz;
    if (brick) *brick = G.brick;
    if (radius) *radius = G.radius;
    if (dst)
    {
        if (bytes < need) return fail(VP_E_ARG, "bound table needs %zu bytes", need);
        HIPCHK(hipMemcpy(dst, G.d_bounds, need, hipMemcpyDeviceToHost));
    }
    return VP_OK;
}
int vp_get_pixel_table(const Param* p, float* dst, size_t count)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !dst) return fail(VP_E_ARG, "vp_get_pixel_table: null argument");
    if (!G.have_volume || !G.have_cam) return fail(VP_E_STATE, "vp_get_pixel_table needs a volume and a camera");
    const size_t need = (size_t)p->width * p->height * 8;
    if (count < need) return fail(VP_E_ARG, "pixel table needs %zu floats", need);
    const float4* t = nullptr;
    rc = ensure_crawl_table(p, &t);
    if (rc) return rc;
    if (!t) return fail(VP_E_STATE, "no pixel table in this configuration (point filtering, or the tables are switched off)");
    HIPCHK(hipMemcpy(dst, t, need * sizeof(float), hipMemcpyDeviceToHost));
    return VP_OK;
}
int vp_get_exit_table(unsigned char* dst, size_t count)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!G.have_volume) return fail(VP_E_STATE, "vp_get_exit_table needs a volume");
    const size_t n = 3 * (size_t)G.S.nx * G.S.ny * G.S.nz;
    if (!dst || count < n) return fail(VP_E_ARG, "exit table needs %zu bytes", n);
    if (!G.d_exit) return fail(VP_E_STATE, "no exit table in this configuration (switched off, or no memory)");
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(dst, G.d_exit, n, hipMemcpyDeviceToHost));
    return VP_OK;
}
int vp_get_sun_clip_table(unsigned short* dst, size_t count, float* step)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!G.have_volume || !G.have_sun) return fail(VP_E_STATE, "vp_get_sun_clip_table needs a volume and a sun");
    const size_t n = (size_t)G.S.nx * G.S.ny * G.S.nz;
    if (!dst || count < n) return fail(VP_E_ARG, "sun clip table needs %zu entries", n);
    const unsigned short* t = nullptr;
    float ds = 0.0f;
    rc = ensure_sun_clip(&t, &ds);
    if (rc) return rc;
    if (!t) return fail(VP_E_STATE, "no sun clip table in this configuration (sampler.h streams, point filtering, or switched off)");
    HIPCHK(hipMemcpy(dst, t, n * sizeof(unsigned short), hipMemcpyDeviceToHost));
    if (step) *step = ds;
    return VP_OK;
}
int vp_get_null_collision_table(const Param* p, float* dst, size_t count)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !dst || !count || count > (1u << 24)) return fail(VP_E_ARG, "vp_get_null_collision_table: bad argument");
    struct Tmp { float* p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } tmp;
    HIPCHK(hipMalloc((void**)&tmp.p, count * sizeof(float)));
    ParamDev P;
    memcpy(&P, p, sizeof(Param));
    launch_thr_table(P, tmp.p, (unsigned)count, G.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(dst, tmp.p, count * sizeof(float), hipMemcpyDeviceToHost));
    return VP_OK;
}
int vp_get_opacity(float* dst, size_t count)
{
    if (!G.d_opacity) return fail(VP_E_STATE, "no opacity table");
    size_t n = (size_t)G.S.nx * G.S.ny * G.S.nz;
    if (count < n) return fail(VP_E_ARG, "opacity needs %zu floats", n);
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(dst, G.d_opacity, n * sizeof(float), hipMemcpyDeviceToHost));
    return VP_OK;
}

int vp_test_math(int which, const float* in, float* out, int n)
{
    int rc = ensure_device();
    if (rc) return rc;
    float *di = nullptr, *d_o = nullptr;
    HIPCHK(hipMalloc((void**)&di, (size_t)n * 4));
    HIPCHK(hipMalloc((void**)&d_o, (size_t)n * 4));
    HIPCHK(hipMemcpy(di, in, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_test_math(which, di, d_o, n, G.stream);
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(out, d_o, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(di));
    HIPCHK(hipFree(d_o));
    return VP_OK;
}
int vp_test_rng(int mode, uint32_t x, uint32_t y, uint32_t frame, uint32_t k0, uint32_t k1, int n, float* out)
{
    int rc = ensure_device();
    if (rc) return rc;
    float* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)n * 4));
    launch_test_rng(mode, x, y, frame, k0, k1, n, d, G.stream);
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(out, d, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    return VP_OK;
}
int vp_test_sample_density(const float* pos_xyz, float* out, int n)
{
    if (!G.have_volume) return fail(VP_E_STATE, "no volume");
    float *dp = nullptr, *dq = nullptr;
    HIPCHK(hipMalloc((void**)&dp, (size_t)n * 12));
    HIPCHK(hipMalloc((void**)&dq, (size_t)n * 4));
    HIPCHK(hipMemcpy(dp, pos_xyz, (size_t)n * 12, hipMemcpyHostToDevice));
    SceneDev S = G.S;
    S.linear   = G.linear ? 1 : 0;
    launch_test_density(S, G.quant, dp, dq, n, G.stream);
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(out, dq, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(dp));
    HIPCHK(hipFree(dq));
    return VP_OK;
}

// device buffers of one test call: freed on every return path
struct DevArrays
{
    std::vector<void*> p;
    ~DevArrays() { for (void* q : p) (void)hipFree(q); }
    void* get(size_t bytes)
    {
        void* q = nullptr;
        if (hipMalloc(&q, bytes ? bytes : 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        p.push_back(q);
        return q;
    }
};
int vp_test_hg(const float* g, const float* r0, const float* r1, const float* normal_xyz, const float* cos_query, float* dir_xyz, float* eval, int n)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (n <= 0) return VP_OK;
    DevArrays D;
    const size_t b = (size_t)n * 4;
    float *dg = (float*)D.get(b), *d0 = (float*)D.get(b), *d1 = (float*)D.get(b), *dn = (float*)D.get(3 * b), *dc = (float*)D.get(b);
    float *dd = (float*)D.get(3 * b), *de = (float*)D.get(b);
    if (!dg || !d0 || !d1 || !dn || !dc || !dd || !de) return fail(VP_E_NOMEM, "vp_test_hg: no device memory");
    HIPCHK(hipMemcpy(dg, g, b, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d0, r0, b, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d1, r1, b, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dn, normal_xyz, 3 * b, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc, cos_query, b, hipMemcpyHostToDevice));
    launch_test_hg(dg, d0, d1, dn, dc, dd, de, n, G.stream);
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(dir_xyz, dd, 3 * b, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(eval, de, b, hipMemcpyDeviceToHost));
    return VP_OK;
}
int vp_test_intersect_box(const float* origin_xyz, const float* dir_xyz, int* hit, float* tnear, float* tfar, int n)
{
    if (!G.have_volume) return fail(VP_E_STATE, "no volume (the box comes from init_cuda)");
    int rc = ensure_device();
    if (rc) return rc;
    if (n <= 0) return VP_OK;
    DevArrays D;
    const size_t b = (size_t)n * 4;
    float *dor = (float*)D.get(3 * b), *ddi = (float*)D.get(3 * b), *dtn = (float*)D.get(b), *dtf = (float*)D.get(b);
    int*   dh  = (int*)D.get(b);
    if (!dor || !ddi || !dtn || !dtf || !dh) return fail(VP_E_NOMEM, "vp_test_intersect_box: no device memory");
    HIPCHK(hipMemcpy(dor, origin_xyz, 3 * b, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ddi, dir_xyz, 3 * b, hipMemcpyHostToDevice));
    launch_test_box(G.S, dor, ddi, dh, dtn, dtf, n, G.stream);
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(hit, dh, b, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tnear, dtn, b, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tfar, dtf, b, hipMemcpyDeviceToHost));
    return VP_OK;
}
int vp_test_eval_envmap(const float* dir_xyz, float* rgb, int n)
{
    if (!G.have_env) return fail(VP_E_STATE, "no environment map");
    int rc = ensure_device();
    if (rc) return rc;
    if (n <= 0) return VP_OK;
    DevArrays D;
    const size_t b = (size_t)n * 12;
    float *dd = (float*)D.get(b), *dq = (float*)D.get(b);
    if (!dd || !dq) return fail(VP_E_NOMEM, "vp_test_eval_envmap: no device memory");
    HIPCHK(hipMemcpy(dd, dir_xyz, b, hipMemcpyHostToDevice));
    launch_test_env(G.S, dd, dq, n, G.stream);
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(rgb, dq, b, hipMemcpyDeviceToHost));
    return VP_OK;
}

int vp_julia_voxelize(int n, unsigned char* host_out)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (n < 1 || n > 1024 || !host_out) return fail(VP_E_ARG, "bad julia grid size %d", n);
    size_t total = (size_t)n * n * n;
    unsigned char* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, total));
    launch_julia(d, n, G.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(host_out, d, total, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    return VP_OK;
}

int vp_cloud_voxelize(int n, uint32_t seed, float* host_out)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (n < 1 || n > 1024 || !host_out) return fail(VP_E_ARG, "bad cloud grid size %d", n);
    size_t total = (size_t)n * n * n;
    float* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, total * sizeof(float)));
    launch_cloud(d, n, seed, G.stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(G.stream);
    if (e == hipSuccess) e = hipMemcpy(host_out, d, total * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(VP_E_NODEVICE, "vp_cloud_voxelize -> %s", hipGetErrorString(e));
    return VP_OK;
}

void* vp_malloc(size_t bytes)
{
    if (ensure_device()) return nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
    {
        (void)hipGetLastError();
        fail(e == hipErrorOutOfMemory ? VP_E_NOMEM : VP_E_NODEVICE, "hipMalloc(%zu) -> %s (%s)", bytes, hipGetErrorString(e),
             e == hipErrorOutOfMemory ? "VP_E_NOMEM" : "VP_E_NODEVICE");
        return nullptr;
    }
    return p;
}
int vp_free(void* p) { HIPCHK(hipFree(p)); return VP_OK; }
int vp_memset(void* p, int v, size_t bytes)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(p, v, bytes, G.stream));
    return VP_OK;
}
int vp_upload(void* d, const void* s, size_t bytes)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, G.stream));
    HIPCHK(hipStreamSynchronize(G.stream));
    return VP_OK;
}
int vp_download(void* d, const void* s, size_t bytes)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(G.stream));
    HIPCHK(hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost));
    return VP_OK;
}
}  // extern "C"
