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
#include "param.h"
#include "sky.h"
#include "volume_io.h"
#include "volpath.h"

static void usage()
{
    printf("volpath_render [--julia N | --bin file.bin | --vdb file.vdb] [--size W H] [--spp N] [--preset 0..12]\n"
           "               [--density D] [--g G] [--estimator decomp|global|bounded] [--brick B] [--rng samplerh|philox]\n"
           "               [--tracking spectral|scalar|multichannel] [--env passive|mis]\n"
           "               [--sun X Y] [--batch F] [--out name(.ppm|.hdr)]\n");
}

int main(int argc, char** argv)
{
    int         julia = 128, W = 400, H = 300, spp = 16, preset = 12, brick = 1, batch = 0;
    float       density = 800.0f, g = 0.877f, sunx = 0.5f, suny = 0.2f;
    bool        philox = false;
    int         est = VP_EST_DECOMP, tracking = VP_TRACK_SPECTRAL, env_mode = VP_ENV_PASSIVE;
    std::string bin, vdb, out = "output0.ppm";
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
        else if (a == "--rng") { need(1); philox = !strcmp(argv[++i], "philox"); }
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
        else { usage(); return a == "--help" ? 0 : 2; }
    }

    Param P = default_param(W, H);  // host.cpp:1286-1292
    P.density = density;
    P.g       = g;
    if (!material_preset(P, preset)) { fprintf(stderr, "preset must be 0..12\n"); return 2; }

    // ---- volume (host.cpp:1330-1344)
    int   width = 0, height = 0, depth = 0;
    void* h_volume = nullptr;
    if (!bin.empty()) h_volume = loadBinaryFile(bin.c_str(), width, height, depth, true);
    else if (!vdb.empty()) h_volume = loadVdbFile(vdb.c_str(), width, height, depth, true);
    else
    {
        width = height = depth = julia;
        h_volume = malloc((size_t)julia * julia * julia);
        if (vp_julia_voxelize(julia, (unsigned char*)h_volume)) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
    }
    if (!h_volume) return 1;
    vp_float3 box_min = {-1.0f, -(float)height / (float)width, -(float)depth / (float)width};
    vp_float3 box_max = {1.0f, (float)height / (float)width, (float)depth / (float)width};
    vp_set_bound_brick(brick);
    init_cuda(h_volume, vp_extent{(size_t)width, (size_t)height, (size_t)depth}, true, &box_min, &box_max);
    free(h_volume);
    set_texture_filter_mode(true);

    float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    copy_inv_model_matrix(identity, sizeof(identity));  // host.cpp:1350-1353
    Camera cam;
    float  m[12];
    cam.inv_view_matrix(m);
    copy_inv_view_matrix(m, sizeof(m));                 // host.cpp:617-623

    // ---- sun / sky (host.cpp:1388-1390 -> update_sunsky)
    volpath::SunSky sky = volpath::bake_sunsky(sunx, suny);
    init_envmap(reinterpret_cast<const vp_float4*>(sky.envmap.data()), sky.width, sky.height);
    printf("sun power = %f, %f, %f\n", sky.sun_power.x, sky.sun_power.y, sky.sun_power.z);
    set_sun(&sky.sun_dir.x, &sky.sun_power.x);

    vp_set_estimator(est);
    if (vp_set_tracking(tracking) || vp_set_envmap_sampling(env_mode)) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
    vp_set_rng(philox ? VP_RNG_PHILOX : VP_RNG_SAMPLERH, 0x9E3779B9u, 0x85EBCA6Bu);

    // ---- frame buffer (CudaFrameBuffer host.cpp:358-389)
    const int  npix  = W * H;
    vp_float4* accum = (vp_float4*)vp_malloc((size_t)npix * sizeof(vp_float4));
    vp_float4* disp  = (vp_float4*)vp_malloc((size_t)npix * sizeof(vp_float4));
    if (!accum || !disp) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
    vp_memset(accum, 0, (size_t)npix * sizeof(vp_float4));

    auto t0 = std::chrono::high_resolution_clock::now();
    vp_dim3 block = {8, 8, 1}, grid = {(unsigned)(W + 7) / 8, (unsigned)(H + 7) / 8, 1};
    for (int s = 0; s < spp;)
    {
        if (s > 10 || (batch > 0 && s + batch > 11)) { static bool done = false; if (!done && est == VP_EST_DECOMP) { precompute_opacity(&sky.sun_dir.x); done = true; } }
        if (batch > 0)
        {
            int n = std::min(batch, spp - s);
            if (vp_render_frames(accum, s, n, &P)) { fprintf(stderr, "%s\n", vp_last_error()); return 1; }
            s += n;
        }
        else
        {
            render_kernel(grid, block, accum, s, P);  // host.cpp:631
            s += 1;
        }
    }
    vp_synchronize();
    double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("%f M samples / s, %d x %d, %d spp, %f s\n", (double)W * H * spp / sec / 1e6, W, H, spp, sec);

    // ---- capture (host.cpp:585-610, :508-517)
    bool  hdr = out.size() > 4 && out.substr(out.size() - 4) == ".hdr";
    Image image(W, H);
    if (hdr) scale(disp, accum, npix, 1.0f / spp);
    else gamma_correct(disp, accum, npix, 1.0f / spp, 2.2f);
    vp_download(image.buffer(), disp, (size_t)npix * sizeof(vp_float4));
    if (hdr) image.dump_hdr(out.c_str());
    else image.dump_ppm(out.c_str());
    printf("wrote %s\n", out.c_str());
    vp_free(accum);
    vp_free(disp);
    free_cuda_buffers();
    free_envmap();
    return 0;
}
