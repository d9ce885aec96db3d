// vp_context.cpp -- the C ABI of libvolpath_hip.so (include/volpath.h), part one of four: device, volume, environment and contexts;
// Part 1 of the header (the reference's kernel-TU entry points, kernel.cu:354-451, :526-553, :1072-1283, :2320-2370) and the
// setters, test hooks and memory helpers of Part 2.  The tables a launch reads are vp_tables.cpp, the launch vp_render.cpp,
// render_kernel's frame look-ahead vp_lookahead.cpp; vp_state.h holds what they share.
//
// HBM layout (DESIGN.md "Data layout"): the density volume is stored as one 8-byte (uchar) or
// 32-byte (float) cell per voxel holding that voxel's clamped 2x2x2 texel neighbourhood, so a
// trilinear fetch is ONE aligned load; bounds are (max,min) pairs per brick; opacity is a plain
// float N^3 array; the environment is float4 rows.  No textures, no CPU fallback.
#include "vp_state.h"

namespace vph __attribute__((visibility("hidden")))
{
// The default context (vp_state.h)
State                      g_default;
static thread_local State* t_current = nullptr;
State& cur() { return t_current ? *t_current : g_default; }

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
    {
        // (a chunk is (VP_CHUNK >> k) pixels x 2^k frames: k beyond log2(VP_CHUNK) would leave it no pixel -- and the kernel a division by zero)
        long kmax = 0;
        while ((VP_CHUNK >> (kmax + 1)) >= 1) kmax++;
        if (knob("VP_CHUNK_FRAMES_LOG2", 0, kmax, v)) G.chunk_fshift = (unsigned)v;
    }
    if (knob("VP_NO_LDS_BOUNDS", 0, 1, v)) G.use_lds_bounds = v == 0;
    if (knob("VP_NO_LDS_HELPER", 0, 1, v)) G.lds_helper = v == 0;
    if (knob("VP_NO_LDS_COMPACT", 0, 1, v)) G.use_lds_compact = v == 0;
    if (knob("VP_LDS_COMPACT_CHROMATIC", 0, 1, v)) G.lds_compact_chromatic = v != 0;
    if (knob("VP_LDS_PAIRS", 0, 1, v)) G.lds_pairs = v != 0;
    if (knob("VP_CELL_BRICKS", 0, 1, v)) G.cell_bricks = (int)v;
    if (knob("VP_NO_CRAWL_TABLE", 0, 1, v)) G.use_crawl_table = v == 0;
    if (knob("VP_NO_EMPTY_TABLE", 0, 1, v)) G.use_empty_table = v == 0;
    if (knob("VP_NO_SUN_CLIP", 0, 1, v)) G.use_sun_clip = v == 0;
    if (knob("VP_DENSE_PERCENT", 0, 101, v)) G.dense_fraction = (float)v / 100.0f;
    if (knob("VP_NO_OPACITY_LDS", 0, 1, v)) G.opacity_lds = v == 0;
    if (knob("VP_NO_OPACITY_CELLS", 0, 1, v)) G.use_opacity_cells = v == 0;
    if (knob("VP_NO_LIGHT_CONST", 0, 1, v)) G.use_light_const = v == 0;
    if (knob("VP_NO_CONST_ROWS", 0, 1, v)) G.use_const_rows = v == 0;
    if (knob("VP_NO_APPROACH", 0, 1, v)) G.use_approach = v == 0;
    if (knob("VP_NO_APPROACH_LOCAL", 0, 1, v)) G.use_approach_local = v == 0;
    if (knob("VP_NO_APPROACH_TABLE", 0, 1, v)) G.use_approach_table = v == 0;
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
    if (knob("VP_EXIT_LOCAL", 0, 1, v)) { G.exit_local = v != 0; G.exit_local_auto = false; }
    if (knob("VP_EXIT_K", 1, VP_EXIT_TRIP, v)) G.exit_k = (unsigned)v;
    G.dev_ready = true;
    return VP_OK;
}


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
    if (G.d_opacity_cells) { if (G.opacity_cells_on_host) HIPCHK(hipHostFree(G.h_opacity_cells)); else HIPCHK(hipFree(G.d_opacity_cells)); }
    G.d_opacity_cells = nullptr; G.h_opacity_cells = nullptr; G.S.opacity_cells = nullptr; G.opacity_cells_on_host = false;
    if (G.d_bound_codes) HIPCHK(hipFree(G.d_bound_codes));
    G.d_bound_codes = nullptr; G.bound_codes_ok = false;
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

static int do_init_volume_(const void* h_volume, vp_extent ext, bool quantized, const vp_float3* bmin, const vp_float3* bmax);
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
    // The compact form of the brick table (render_k<..., LDSB = 2>, vp_kernels.h LaunchDev::bound_codes): where the table fits the LDS
    // stage and holds at most FOUR distinct (max,min) byte pairs -- a binary volume such as the Julia sets has three -- 2-bit codes
    // into a palette: 8 KiB at 32768 bricks instead of 64, which fits beside a plain workgroup's cold per-path state.  Built on the host
    // from one 64 KiB read-back (set-up time, with the volume); best effort: without it the 16-bit table goes through LDS as before.
    G.bound_codes_ok = false;
    if (quantized && G.use_lds_compact && nb <= (size_t)VP_LDS_BOUND_ENTRIES)
    {
        std::vector<unsigned short> tab(nb);
        HIPCHK(hipMemcpyAsync(tab.data(), G.d_bounds, nb * 2, hipMemcpyDeviceToHost, G.stream));
        HIPCHK(hipStreamSynchronize(G.stream));
        unsigned short pal[4] = {0, 0, 0, 0};
        int            npal   = 0;
        bool           fits   = true;
        std::vector<unsigned> codes((nb + 15) / 16 + 4, 0u);   // + padding: the LDS stage copies whole 16-byte words
        for (size_t b = 0; b < nb && fits; b++)
        {
            int c = 0;
            while (c < npal && pal[c] != tab[b]) c++;
            if (c == npal)
            {
                if (npal == 4) { fits = false; break; }
                pal[npal++] = tab[b];
            }
            codes[b >> 4] |= (unsigned)c << ((b & 15u) << 1);
        }
        if (fits)
        {
            const size_t bytes = (((nb + 15) / 16) * 4 + 15) / 16 * 16;
            if (hipMalloc((void**)&G.d_bound_codes, bytes) == hipSuccess)
            {
                HIPCHK(hipMemcpyAsync(G.d_bound_codes, codes.data(), bytes, hipMemcpyHostToDevice, G.stream));
                HIPCHK(hipStreamSynchronize(G.stream));   // (codes is a local)
                G.bound_pal[0] = (unsigned)pal[0] | (unsigned)pal[1] << 16;
                G.bound_pal[1] = (unsigned)pal[2] | (unsigned)pal[3] << 16;
                G.bound_codes_ok = true;
            }
            else { (void)hipGetLastError(); G.d_bound_codes = nullptr; }
        }
    }
    // cells with a non-empty cell in their neighbourhood: input of the certified-empty table of the global-majorant estimator
    if (G.use_empty_table && hipMalloc((void**)&G.d_danger, n) == hipSuccess)
    {
        // (the counter: the last word of the work-counter block is nobody's)
        unsigned long long* d_marked = G.d_counters + (kCounterWords - 1);
        HIPCHK(hipMemsetAsync(d_marked, 0, sizeof(unsigned long long), G.stream));
        launch_danger(S, quantized, G.d_danger, d_marked, G.stream);
        HIPCHK(hipGetLastError());
        unsigned long long marked = 0;
        HIPCHK(hipMemcpyAsync(&marked, d_marked, sizeof marked, hipMemcpyDeviceToHost, G.stream));
        HIPCHK(hipStreamSynchronize(G.stream));
        HIPCHK(hipMemsetAsync(d_marked, 0, sizeof(unsigned long long), G.stream));
        G.marked_fraction = (float)((double)marked / (double)n);
    }
    else { (void)hipGetLastError(); G.d_danger = nullptr; G.marked_fraction = 0.0f; }  // no memory for it: the estimator fetches every cell, same bits
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
}  // namespace vph

using namespace vph;

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
        if (D.d_seg) (void)hipFree(D.d_seg);
        if (D.d_bound_codes) (void)hipFree(D.d_bound_codes);
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
    G.use_exit = mode != 0; G.exit_local = mode == 2; G.exit_local_auto = false;
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

