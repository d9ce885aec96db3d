// vp_lookahead.cpp -- render_kernel's frame look-ahead (the reference host's one-frame-per-call pattern).  FROZEN since round 4: the interactive shell it serves is out of scope; tests/test_fuzz_gpu.py's in-flight call sequences are its safety net
#include "vp_state.h"

namespace vph __attribute__((visibility("hidden")))
{
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
}  // namespace vph

