// sky.h -- sun/sky environment of the integrator: the reference's SkyModel<Tungsten::Skydome>
// (src/sunsky/sunsky.h, sky_tungsten.{h,cpp}) over the Hosek-Wilkie spectral sky-dome model in its
// "alien world" initialisation (T = 5777 K, turbidity 2, intensity 100, ground albedo 0.2), and the
// host-side bake of host.cpp:276-333.
#ifndef VOLPATH_HOST_SKY_H
#define VOLPATH_HOST_SKY_H
#include <vector>

#include "vec.h"

namespace volpath
{
// Hosek-Wilkie model state for one solar elevation (restated from the published model; only the
// turbidity-2 coefficient rows are carried, see data/sky_tables.inc)
struct HosekState
{
    double config[11][9];
    double radiance[11];
    double turbidity, solar_radius, albedo, elevation;
    double corr_sky[11], corr_sun[11];
};
// returns false for a turbidity other than 2 (no coefficients carried)
bool   hosek_alienworld_init(HosekState& s, double solar_elevation, double solar_intensity, double kelvin,
                             double turbidity, double ground_albedo);
double hosek_radiance(const HosekState& s, double theta, double gamma, double wavelength);
double hosek_solar_radiance(const HosekState& s, double theta, double gamma, double wavelength);

// sky_tungsten.h:12-53 surface
class Skydome
{
public:
    Skydome();
    void   setSunTheta(float t) { _theta = t; _prepared = false; }
    void   setSunPhi(float p) { _phi = p; _prepared = false; }
    float3 sunDirection() const;
    float3 getSunDir() const { return sunDirection(); }
    float3 skyColor(const float3& direction, bool CEL = false);
    float3 sunColor();
    float  turbidity() const { return _turbidity; }
    float  intensity() const { return _intensity; }

private:
    void       prepare();
    float      _temperature, _gammaScale, _turbidity, _intensity;
    float      _theta = 0.0f, _phi = 0.0f;
    bool       _prepared = false;
    HosekState _state{};
    float3     _sun{};
};

struct SunSky
{
    int                 width = 0, height = 0;
    std::vector<float4> envmap;       // lat-long, row 0 = zenith (init_envmap input)
    float3              sun_dir{};    // set_sun inputs
    float3              sun_power{};
};
// update_sunsky(baked = true), host.cpp:276-333, for setup_sunsky(x, y)
SunSky bake_sunsky(float x, float y, int width = 1024, int height = 512);
}  // namespace volpath
#endif
