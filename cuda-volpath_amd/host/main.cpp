// main.cpp -- volpath_render: headless driver in place of the reference's GLUT shell (host.cpp:1284-1403).
// It performs main()'s set-up sequence against the SAME entry points (init_cuda, set_texture_filter_mode,
// copy_inv_model_matrix, copy_inv_view_matrix, init_envmap, set_sun, precompute_opacity, render_kernel,
// scale / gamma_correct) and writes what the 'c' key captures (host.cpp:585-610): a .ppm of the
// gamma-corrected image or a .hdr of the scaled accumulator.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "camera.h"
#include "image.h"
#include "param.h"  // before volpath.h: the host spells Param with float3 members (same 44 bytes)
#include "multigpu.h"
#include "sky.h"
#include "volume_io.h"
#include "volpath.h"

static void usage()
{
    printf("volpath_render [--julia N | --bin file.bin | --vdb file.vdb] [--size W H] [--spp N] [--preset 0..12]\n"
           "               [--density D] [--g G] [--estimator decomp|global|bounded] [--brick B] [--rng samplerh|philox|philox7]\n"
           "               [--tracking spectral|scalar|multichannel] [--env passive|mis]\n"
           "               [--sun X Y] [--batch F] [--out name(.ppm|.hdr)]\n"
           "               [--gpus N [--devices a,b,...]]   N contexts, pixel tiles dealt by vp_set_shard, one RCCL reduce;\n"
           "                                                a repeated device (e.g. --gpus 2 --devices 0,0) shares one GPU\n"
           "                                                (distinct devices: the RCCL path has run with one rank only so far -- UNVERIFIED\n"
           "                                                between GPUs; --rccl-selftest checks that RCCL loads and reduces here)\n"
           "               [--rccl-selftest [device]]       load RCCL, one-rank communicator, one reduce: the calls of the N > 1 path\n");
}

int main(int argc, char** argv)
{
    int         julia = 128, W = 400, H = 300, spp = 16, preset = 12, brick = 1, batch = 0;
    float       density = 800.0f, g = 0.877f, sunx = 0.5f, suny = 0.2f;
    int         philox = 0;   // 0 sampler.h, 1 Philox2x32-10, 2 Philox2x32-7
    int         est = VP_EST_DECOMP, tracking = VP_TRACK_SPECTRAL, env_mode = VP_ENV_PASSIVE;
    std::string bin, vdb, out = "output0.ppm", devlist;
    int         gpus = 1;
    for (int i = 1; i < argc; i++)
    {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { usage(); exit(2); } };
        if (a == "--julia") { need(1); julia = atoi(argv[++i]); }
        else if (a == "--bin") { need(1); bin = argv[++i]; }
        else if (a == "--vdb") { need(1); vdb = argv[++i]; }
        else if (a == "--size") { need(2); W = atoi(argv[++i]); H = atoi(argv[++i]); }
        else if (a == "--spp") { need(1); spp = atoi(argv[++i]); }
        else if (a == "--preset") { need(1); preset = atoi(argv[++i]); }
        else if (a == "--density") { need(1); density = (float)atof(argv[++i]); }
        else if (a == "--g") { need(1); g = (float)atof(argv[++i]); }
        else if (a == "--estimator")
        {
            need(1);
            const char* e = argv[++i];
            est = !strcmp(e, "global") ? VP_EST_GLOBAL : !strcmp(e, "bounded") ? VP_EST_BOUNDED : VP_EST_DECOMP;
        }
        else if (a == "--brick") { need(1); brick = atoi(argv[++i]); }
        else if (a == "--rng") { need(1); ++i; philox = !strcmp(argv[i], "philox") ? 1 : !strcmp(argv[i], "philox7") ? 2 : 0; }
        else if (a == "--tracking")
        {
            need(1);
            const char* t = argv[++i];
            tracking = !strcmp(t, "scalar") ? VP_TRACK_SCALAR : !strcmp(t, "multichannel") ? VP_TRACK_MULTI_CHANNEL : VP_TRACK_SPECTRAL;
        }
        else if (a == "--env") { need(1); env_mode = !strcmp(argv[++i], "mis") ? VP_ENV_MIS : VP_ENV_PASSIVE; }
        else if (a == "--sun") { need(2); sunx = (float)atof(argv[++i]); suny = (float)atof(argv[++i]); }
        else if (a == "--batch") { need(1); batch = atoi(argv[++i]); }
        else if (a == "--out") { need(1); out = argv[++i]; }
        else if (a == "--gpus") { need(1); gpus = atoi(argv[++i]); }
        else if (a == "--devices") { need(1); devlist = argv[++i]; }
        else if (a == "--rccl-selftest")
        {
            // the RCCL calls of the multi-GPU path on one device (multigpu.h)
            const int   dev = i + 1 < argc ? atoi(argv[i + 1]) : 0;
            std::string report;
            const bool  ok = volpath::rccl_selftest(dev, (size_t)1280 * 720 * 4, report);
            printf("%s\n", report.c_str());
            return ok ? 0 : 1;
        }
        else { usage(); return a == "--help" ? 0 : 2; }
    }

    Param P = default_param(W, H);  // host.cpp:1286-1292
    P.density = density;
    P.g       = g;
    if (!material_preset(P, preset)) { fprintf(stderr, "preset must be 0..12\n"); return 2; }

    // ---- the GPUs: one context per rank (a single rank runs in the default context, as the reference's host would)
    if (gpus < 1) { usage(); return 2; }
    std::vector<int> devices;
    for (size_t pos = 0; pos < devlist.size();)
    {
        size_t e = devlist.find(',', pos);
        if (e == std::string::npos) e = devlist.size();
        devices.push_back(atoi(devlist.substr(pos, e - pos).c_str()));
        pos = e + 1;
    }
    if (devices.empty()) for (int r = 0; r < gpus; r++) devices.push_back(r);
    if ((int)devices.size() != gpus) { fprintf(stderr, "--devices must list %d devices\n", gpus); return 2; }
    if (gpus > 1 && batch <= 0) batch = std::min(spp, 256);  // shards render in batches: one launch per rank keeps every GPU busy
    std::vector<vp_ctx*> ctx(gpus, nullptr);
    if (gpus > 1)
        for (int r = 0; r < gpus; r++)
        {
            ctx[r] = vp_ctx_create(devices[r]);
            if (!ctx[r]) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
        }
    else if (vp_set_device(devices[0])) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
    auto use = [&](int r) { if (gpus > 1) vp_ctx_set_current(ctx[r]); };

    // ---- volume (host.cpp:1330-1344): loaded once, uploaded to every rank (all read-only scene data is replicated)
    int   width = 0, height = 0, depth = 0;
    void* h_volume = nullptr;
    if (!bin.empty()) h_volume = loadBinaryFile(bin.c_str(), width, height, depth, true);
    else if (!vdb.empty()) h_volume = loadVdbFile(vdb.c_str(), width, height, depth, true);
    else
    {
        width = height = depth = julia;
        h_volume = malloc((size_t)julia * julia * julia);
        use(0);
        if (!h_volume || vp_julia_voxelize(julia, (unsigned char*)h_volume)) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
    }
    if (!h_volume) return 1;
    vp_float3 box_min = {-1.0f, -(float)height / (float)width, -(float)depth / (float)width};
    vp_float3 box_max = {1.0f, (float)height / (float)width, (float)depth / (float)width};
    float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    Camera cam;
    float  m[12];
    cam.inv_view_matrix(m);
    // sun / sky (host.cpp:1388-1390 -> update_sunsky)
    volpath::SunSky sky = volpath::bake_sunsky(sunx, suny);
    printf("sun power = %f, %f, %f\n", sky.sun_power.x, sky.sun_power.y, sky.sun_power.z);

    const int               npix = W * H;
    std::vector<vp_float4*> accum(gpus, nullptr);
    std::vector<void*>      streams(gpus, nullptr);
    for (int r = 0; r < gpus; r++)
    {
        use(r);
        vp_set_bound_brick(brick);
        init_cuda(h_volume, vp_extent{(size_t)width, (size_t)height, (size_t)depth}, true, &box_min, &box_max);
        set_texture_filter_mode(true);
        copy_inv_model_matrix(identity, sizeof(identity));  // host.cpp:1350-1353
        copy_inv_view_matrix(m, sizeof(m));                 // host.cpp:617-623
        init_envmap(reinterpret_cast<const vp_float4*>(sky.envmap.data()), sky.width, sky.height);
        set_sun(&sky.sun_dir.x, &sky.sun_power.x);
        vp_set_estimator(est);
        if (vp_set_tracking(tracking) || vp_set_envmap_sampling(env_mode) || vp_set_shard(r, gpus)) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
        vp_set_rng(philox == 2 ? VP_RNG_PHILOX7 : philox ? VP_RNG_PHILOX : VP_RNG_SAMPLERH, 0x9E3779B9u, 0x85EBCA6Bu);
        // frame buffer (CudaFrameBuffer host.cpp:358-389), full frame on every rank: zero outside its tiles
        accum[r] = (vp_float4*)vp_malloc((size_t)npix * sizeof(vp_float4));
        if (!accum[r]) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
        vp_memset(accum[r], 0, (size_t)npix * sizeof(vp_float4));
        streams[r] = vp_get_stream();
    }
    free(h_volume);
    volpath::NodeReducer reducer;
    std::string          rerr;
    if (!reducer.init(devices, rerr)) { fprintf(stderr, "%s\n", rerr.c_str()); return 1; }
    if (gpus > 1) printf("multi-GPU reducer: %s\n", reducer.describe().c_str());
    use(0);
    vp_float4* disp = (vp_float4*)vp_malloc((size_t)npix * sizeof(vp_float4));
    if (!disp) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
    for (int r = 0; r < gpus; r++) { use(r); vp_synchronize(); vp_render_time_ms(nullptr, nullptr, 1); }

    auto t0 = std::chrono::high_resolution_clock::now();
    vp_dim3 block = {8, 8, 1}, grid = {(unsigned)(W + 7) / 8, (unsigned)(H + 7) / 8, 1};
    bool    have_opacity = false;
    for (int s = 0; s < spp;)
    {
        if (!have_opacity && est == VP_EST_DECOMP && (s > 10 || (batch > 0 && s + batch > 11)))
        {
            for (int r = 0; r < gpus; r++) { use(r); precompute_opacity(&sky.sun_dir.x); }  // host.cpp:336-343
            have_opacity = true;
        }
        if (batch > 0)
        {
            int n = std::min(batch, spp - s);
            for (int r = 0; r < gpus; r++)  // asynchronous: every rank's launch is queued before any is waited for
            {
                use(r);
                if (vp_render_frames(accum[r], s, n, &P)) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
            }
            s += n;
        }
        else
        {
            use(0);
            render_kernel(grid, block, accum[0], s, P);  // host.cpp:631
            s += 1;
        }
    }
    // ---- the one collective of the job: HDR accumulators -> rank 0
    if (gpus > 1 && !reducer.reduce_to_root(ctx, accum, streams, (size_t)npix, rerr)) { fprintf(stderr, "%s\n", rerr.c_str()); return 1; }
    for (int r = gpus - 1; r >= 0; r--) { use(r); vp_synchronize(); }
    double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("%f M samples / s, %d x %d, %d spp, %f s\n", (double)W * H * spp / sec / 1e6, W, H, spp, sec);
    if (gpus > 1)
    {
        double tmax = 0, tsum = 0;
        printf("%d ranks (%s): kernel ms per rank", gpus, reducer.uses_rccl() ? "RCCL reduce" : "shared device, on-device sum");
        for (int r = 0; r < gpus; r++)
        {
            use(r);
            double ms = 0; int nl = 0;
            vp_render_time_ms(&ms, &nl, 1);
            printf(" %.2f", ms);
            tmax = std::max(tmax, ms); tsum += ms;
        }
        printf("; balance max/mean %.3f\n", tsum > 0 ? tmax / (tsum / gpus) : 1.0);
    }

    // ---- capture (host.cpp:585-610, :508-517) from rank 0
    use(0);
    bool  hdr = out.size() > 4 && out.substr(out.size() - 4) == ".hdr";
    Image image(W, H);
    if (hdr) scale(disp, accum[0], npix, 1.0f / spp);
    else gamma_correct(disp, accum[0], npix, 1.0f / spp, 2.2f);
    vp_download(image.buffer(), disp, (size_t)npix * sizeof(vp_float4));
    if (hdr) image.dump_hdr(out.c_str());
    else image.dump_ppm(out.c_str());
    printf("wrote %s\n", out.c_str());
    vp_free(disp);
    for (int r = 0; r < gpus; r++)
    {
        use(r);
        vp_free(accum[r]);
        free_cuda_buffers();
        free_envmap();
    }
    reducer.shutdown();   // communicators first: their collectives ran on the contexts' streams
    if (gpus > 1) { vp_ctx_set_current(nullptr); for (auto c : ctx) vp_ctx_destroy(c); }
    return 0;
}
