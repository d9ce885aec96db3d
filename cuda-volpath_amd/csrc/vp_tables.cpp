// vp_tables.cpp -- the tables a launch reads besides the volume: optical depth (precompute_opacity), per-pixel tables, sun table, exit table, pixel lists; and their read-back entry points
#include "vp_state.h"

namespace vph __attribute__((visibility("hidden")))
{
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
    // the copy is 8x the table -- 4.3 GB at 512^3, 34 GB at 1024^3.  Where the device cannot hold it, it goes to PINNED HOST
    // memory and the integrator reads it across the bus: the same kernel, the same bits, slow -- but precompute_opacity succeeds where
    // it did before the copy existed, and the frames from 11 on can be rendered.  (A branch to the plain table inside render_k was built
    // first: same bits, and 1.0-1.5 % off EVERY chromatic launch whatever its form -- per-lane, wave-uniform, out of line:
    // profiles/experiments/r05_opacity_fallback.txt.  A fall-back must not tax the path that does not need it.)
    // VP_NO_OPACITY_CELLS=1 takes the host copy on purpose: the knob test renders through it.
    const size_t cbytes = n * 8 * sizeof(float);
    if (G.d_opacity_cells && (G.opacity_cells_on_host != !G.use_opacity_cells))
    {
        if (G.opacity_cells_on_host) HIPCHK(hipHostFree(G.h_opacity_cells)); else HIPCHK(hipFree(G.d_opacity_cells));
        G.d_opacity_cells = nullptr; G.h_opacity_cells = nullptr;
    }
    if (!G.d_opacity_cells)
    {
        G.opacity_cells_on_host = !G.use_opacity_cells;
        if (G.use_opacity_cells && hipMalloc((void**)&G.d_opacity_cells, cbytes) != hipSuccess)
        {
            (void)hipGetLastError();
            G.d_opacity_cells = nullptr;
            G.opacity_cells_on_host = true;
        }
        if (!G.d_opacity_cells)
        {
            void* h = nullptr;
            if (hipHostMalloc(&h, cbytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess)
            {
                (void)hipGetLastError();
                return fail(VP_E_NOMEM, "no memory for the integrator's copy of the optical-depth table (%zu bytes), on the device or pinned on the host", cbytes);
            }
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(h); return fail(VP_E_NOMEM, "pinned host memory is not visible to the device"); }
            G.d_opacity_cells = (float*)d; G.h_opacity_cells = h;
            if (G.use_opacity_cells)   // (not when the knob asked for it)
                fprintf(stderr, "volpath_hip: no device memory for the integrator's copy of the optical-depth table (%zu MB): it lies in pinned host "
                                "memory, frames from 11 on are read across the bus (same results, slow)\n", cbytes >> 20);
        }
    }
    launch_pack_f32(G.d_opacity, G.d_opacity_cells, G.S.nx, G.S.ny, G.S.nz, false, G.stream);
    HIPCHK(hipGetLastError());
    G.S.opacity = G.d_opacity;
    G.S.opacity_cells = G.d_opacity_cells;
    return VP_OK;
}


// the shard of this context (include/volpath.h vp_tile_owner): its 8x8 tiles and the pixels of the image they hold
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
    if (G.marked_fraction > G.dense_fraction) return VP_OK;   // a dense volume: the table costs a lookup per collision and ends few rays early (vp_state.h)
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
        // With local majorants the way out through empty bricks is a restart segment and ONE free flight per 0.05 of length, made by
        // lanes that ride along with their wave's fetching lanes: in round 4 ending those paths early removed 11 % of the lane-steps
        // and not one wave-iteration (C3 +2 %, c3ref 0, c4s -0.5 %, sampler.h -1...-2 %) and stayed off.  Re-measured in round 5 with
        // the six-wave kernels and the compact LDS table, 1024 frames per launch, noise floor 0.1 %: C3 +3.9 %, c4s +1.6 %, c3ref
        // +1.0 ... 1.5 %, c4f 0; sampler.h +0.4 / -0.5 / +0.6 %.  On by default for the counter-based streams (exit_local_auto).
        const bool local_on = G.exit_local_auto ? G.rng != VP_RNG_SAMPLERH : G.exit_local;
        if (!local_on || !G.quant) return VP_OK;
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
// The chain of restart segments of every general pixel's camera ray (vp_kernels.hip approach_segments_k): what approach_local_tab_k
// reads instead of setting every segment up per sample.  Depends on what the crawl table depends on (camera, box, volume, bound
// table, image size) and on the pixel list; rebuilt with either.  Best effort: without it approach_local_k walks as before.
int ensure_segment_table(const Param* p, const float4* crawl, const float4** out)
{
    *out = nullptr;
    if (!G.use_approach_table || !crawl || !G.quant || G.est != VP_EST_DECOMP || !G.n_general || !G.d_tiles) return VP_OK;
    std::vector<unsigned char> key = G.crawl_key;
    key.insert(key.end(), G.tiles_key.begin(), G.tiles_key.end());
    const size_t need = (size_t)G.n_general * segment_table_records() * sizeof(float4);
    if (key != G.seg_key || !G.d_seg)
    {
        if (la_quiesce()) return VP_E_NODEVICE;   // batches in flight read the old table
        HIPCHK(hipStreamSynchronize(G.stream));
        if (need > G.seg_bytes)
        {
            if (G.d_seg) HIPCHK(hipFree(G.d_seg));
            G.d_seg = nullptr; G.seg_bytes = 0; G.seg_key.clear();
            if (hipMalloc((void**)&G.d_seg, need) != hipSuccess)
            {
                (void)hipGetLastError();
                G.d_seg = nullptr;
                return VP_OK;   // no table: every sample sets its segments up itself, same bits
            }
            G.seg_bytes = need;
        }
        SceneDev S = G.S;
        S.linear   = G.linear ? 1 : 0;
        launch_segment_table(S, p->width, p->height, crawl, G.d_tiles, G.n_general, G.d_seg, G.stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(G.stream));   // (launches on other streams read it)
        G.seg_key = key;
    }
    *out = G.d_seg;
    return VP_OK;
}
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

}  // namespace vph

using namespace vph;

extern "C" {
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

}  // extern "C"
