// vp_render.cpp -- do_render: one staged launch of the integrator (approach kernel, render_k per pixel class, per-pixel constants, the add-kernel); counters, timing, vp_prepare / vp_reserve_frames
#include "vp_state.h"

namespace vph __attribute__((visibility("hidden")))
{
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


// where a render launch goes: the caller's stream with the shared staging buffer, or a look-ahead slot

int do_render(vp_float4* d_out, int first, int nframes, const Param* p, bool stage_only, const Target* tgt)
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
        // (Rounds 4-5 gave the chromatic local-majorant kernels 32 parked lanes: their event pass was the most expensive.  Since the event
        // section reads its uniforms from LDS -- vp_kernels.hip kargs_lds_ -- a visit is cheap enough that the general default is the
        // better one there too: c4s +1.1 %, c4f +0.9 % at 28, profiles/experiments/r05_kargs_lds.txt.  Performance only.)
        // the sequential sampler.h stream (the parity mode): its shadow rays walk to their ends, a wave's lanes park later -- 24 lanes, the
        // default of rounds 1-4, stays 0.8 % ahead of 28 on the reference's live configuration (profiles/r05_raw/sweep_samplerh.txt)
        if (!G.wait_lanes_set && G.rng == VP_RNG_SAMPLERH && G.trk == VP_TRACK_SPECTRAL) L.wait_lanes = 24;
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
    const bool dense_volume = G.marked_fraction > G.dense_fraction;   // (vp_state.h: little empty space for the walk to cross)
    if (G.use_approach && (G.est == VP_EST_GLOBAL || (G.est == VP_EST_DECOMP && G.use_approach_local && !dense_volume)) && !G.trk && !G.env_mis && L.crawl && G.n_general &&
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
    // (decomposition estimator: the restart segments of every general pixel's camera ray, tabulated once per view)
    L.seg_table = nullptr;
    if (approach && G.est == VP_EST_DECOMP && nframes >= 64 && G.approach_fshift_max >= 6)   // (read by waves of one pixel x 64 frames)
    {
        rc = ensure_segment_table(p, L.crawl, &L.seg_table);
        if (rc) return rc;
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
        // ... as 2-bit codes where it has at most four distinct pairs (round 5): the plain kernel's registers and occupancy with the table
        // in LDS.  Timed launches of the counter-based streams; counting launches and look-ahead batches keep the 16-bit form.
        // Chromatic media too, since their kernel fits six waves per SIMD (79 registers: intersect_box axis by axis, vp_device.h): c4s
        // 1935 with the 16-bit table and its helper, 2001 from global memory, 2024 with the codes (profiles/experiments/r05_box_sequence.txt;
        // before that change 1.7 % behind the 16-bit table: r05_lds_compact_table.txt).  VP_LDS_COMPACT_CHROMATIC=0: not.
        // A table that cannot go as codes: for those launches from GLOBAL memory (six / five waves) rather than as 16-bit pairs through LDS
        // (four waves and a helper workgroup) -- c4f +1.6 %; C3 was level already (round 5).  VP_LDS_PAIRS=1: the pairs.
        const bool ach_lds = p->sigma_t.x == p->sigma_t.y && p->sigma_t.y == p->sigma_t.z && p->albedo.x == p->albedo.y && p->albedo.y == p->albedo.z;
        // (the sequential sampler.h stream has no codes kernel; its timed launches, too, are ahead without the pairs: c3 +2.2 %, c4s +4.3 %)
        const bool timed_l  = !G.count && !tgt;                       // a timed launch (not a counting one, not a look-ahead batch)
        const bool timed_cb = timed_l && G.rng != VP_RNG_SAMPLERH;    // ... of a counter-based stream
        const int lds_form = !lds_bounds ? 0 : (G.bound_codes_ok && G.d_bound_codes && timed_cb && (ach_lds || G.lds_compact_chromatic)) ? 2
                                             : (timed_l && !G.lds_pairs) ? 0 : 1;
        G.last_lds_form = lds_form;
        L.bound_codes = lds_form == 2 ? G.d_bound_codes : nullptr;
        L.bound_pal[0] = G.bound_pal[0]; L.bound_pal[1] = G.bound_pal[1];
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
        const bool lds_helper = lds_form == 1 && G.lds_helper && G.n_general && !(G.n_light && !light_const) && !tgt;
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
            const bool     ldsb = lds_form == 1 && !cls;   // (the 16-bit table: 512-thread workgroups, two per CU)
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
                G.last_approach_table = 0;
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
                    G.last_approach_table = (G.est == VP_EST_DECOMP && G.quant && L.seg_table && L.approach_fshift == 6u) ? 1 : 0;
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
                    launch_render(S, L, G.est, G.rng, G.quant, G.count, lds_form, G.env_mis, G.trk, (int)blocks, T.stream);
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
                        launch_render(S, L, G.est, G.rng, G.quant, G.count, 0, G.env_mis, G.trk, G.num_cu, G.aux_stream[ti]);
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

}  // namespace vph

using namespace vph;

extern "C" {
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
            if (h[72]) fprintf(stderr, "[vp]   %-18s %14llu  %5.1f\n", "control component", h[72], (double)h[73] / (double)h[72]);
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
int vp_last_approach_table(void) { return G.last_approach_table; }
int vp_last_light_const(void) { return G.last_light_const; }
int vp_last_lds_form(void) { return G.last_lds_form; }
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
    const size_t need4 = (G.est == VP_EST_DECOMP && G.use_approach && G.use_approach_local && !(G.marked_fraction > G.dense_fraction)) ? sh.per_frame * f * sizeof(uint2) : 0;
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
}  // extern "C"
