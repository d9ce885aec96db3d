// capi.cpp -- C entry points of libvolpath_host.so so that the host-side pieces (sky bake, image
// writers, volume ingest, camera, material presets) can be exercised from Python tests and bench.py.
#include <cstdlib>
#include <cstring>

#include "camera.h"
#include "image.h"
#include "param.h"
#include "sky.h"
#include "volume_io.h"

extern "C" {
// update_sunsky (host.cpp:276-333): env[w*h*4], sun_dir[3], sun_power[3]
int vph_bake_sunsky(float x, float y, int w, int h, float* env, float* sun_dir, float* sun_power)
{
    volpath::SunSky s = volpath::bake_sunsky(x, y, w, h);
    memcpy(env, s.envmap.data(), (size_t)w * h * sizeof(float4));
    sun_dir[0] = s.sun_dir.x; sun_dir[1] = s.sun_dir.y; sun_dir[2] = s.sun_dir.z;
    sun_power[0] = s.sun_power.x; sun_power[1] = s.sun_power.y; sun_power[2] = s.sun_power.z;
    return 0;
}
// raw Hosek model: out[0] = sky radiance, out[1] = solar radiance (alien-world state at `elevation`)
int vph_hosek(double elevation, double intensity, double kelvin, double turbidity, double albedo, double theta, double gamma,
              double lambda, double* out)
{
    volpath::HosekState st;
    if (!volpath::hosek_alienworld_init(st, elevation, intensity, kelvin, turbidity, albedo)) return -1;
    out[0] = volpath::hosek_radiance(st, theta, gamma, lambda);
    out[1] = volpath::hosek_solar_radiance(st, theta, gamma, lambda);
    return 0;
}
int vph_sky_color(float sun_theta, float sun_phi, const float* dir, int cel, float* rgb)
{
    volpath::Skydome s;
    s.setSunTheta(sun_theta);
    s.setSunPhi(sun_phi);
    float3 c = s.skyColor(make_float3(dir[0], dir[1], dir[2]), cel != 0);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return 0;
}
int vph_sun_color(float sun_theta, float sun_phi, float* rgb, float* dir)
{
    volpath::Skydome s;
    s.setSunTheta(sun_theta);
    s.setSunPhi(sun_phi);
    float3 c = s.sunColor(), d = s.getSunDir();
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
    return 0;
}
// image writers / tone mapping on a float4 buffer (mode: 0 none, 1 gamma, 2 reinhard)
int vph_write_image(const float* rgba, int w, int h, const char* path, int hdr, int tonemap, float gamma, float scale)
{
    Image img(w, h);
    memcpy(img.buffer(), rgba, (size_t)w * h * 16);
    if (scale != 1.0f) img.scale(scale);
    if (tonemap == 1) img.tonemap_gamma(gamma);
    if (tonemap == 2) img.tonemap_reinhard();
    if (hdr) img.dump_hdr(path);
    else img.dump_ppm(path);
    return 0;
}
int vph_image_ops(float* rgba, int w, int h, int op, float arg)
{
    Image img(w, h);
    memcpy(img.buffer(), rgba, (size_t)w * h * 16);
    switch (op)
    {
        case 0: img.scale(arg); break;
        case 1: img.flip_updown(); break;
        case 2: img.tonemap_gamma(arg); break;
        case 3: img.tonemap_reinhard(); break;
        default: return -1;
    }
    memcpy(rgba, img.buffer(), (size_t)w * h * 16);
    return 0;
}
// volume ingest; returned pointers are malloc'ed, release with vph_free
void* vph_load_binary(const char* path, int* w, int* h, int* d, int quantized) { return loadBinaryFile(path, *w, *h, *d, quantized != 0); }
void* vph_load_raw(const char* path, size_t size) { return loadRawFile(path, size); }
void* vph_load_vdb(const char* path, int* w, int* h, int* d, int quantized) { return loadVdbFile(path, *w, *h, *d, quantized != 0); }
int   vph_dump_dense(const char* path, const float* data, int nx, int ny, int nz) { return dump_dense_volume(path, data, nx, ny, nz) ? 0 : -1; }
void  vph_quantize(const float* src, size_t n, float max_value, unsigned char* dst)
{
    if (max_value > 0.0f) quantize_by_max(src, n, max_value, dst);
    else quantize_unit(src, n, dst);
}
void vph_free(void* p) { free(p); }
// camera and materials
void vph_camera_matrix(const float* pos, const float* fwd, const float* up, float focus, float* m12)
{
    Camera c;
    if (pos) c.position = make_float3(pos[0], pos[1], pos[2]);
    if (fwd) c.forward = make_float3(fwd[0], fwd[1], fwd[2]);
    if (up) c.up = make_float3(up[0], up[1], up[2]);
    if (focus > 0.0f) c.focus_dist = focus;
    c.inv_view_matrix(m12);
}
int vph_material_preset(int index, float* sigma_t, float* albedo)
{
    Param P = default_param();
    if (!material_preset(P, index)) return -1;
    sigma_t[0] = P.sigma_t.x; sigma_t[1] = P.sigma_t.y; sigma_t[2] = P.sigma_t.z;
    albedo[0] = P.albedo.x; albedo[1] = P.albedo.y; albedo[2] = P.albedo.z;
    return 0;
}
}
