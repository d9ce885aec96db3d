// sky.cpp -- see sky.h.  Model evaluation restated from the papers' closed forms:
//   sky radiance   F(theta,gamma) = (1 + A e^{B/(cos theta + 0.01)}) (C + D e^{E gamma} + F cos^2 gamma
//                                    + G chi(H, gamma) + I sqrt(cos theta)),
//   chi(g, a) = (1 + cos^2 a) / (1 + g^2 - 2 g cos a)^{3/2}
//   coefficients: quintic Bezier in (elevation / (pi/2))^{1/3}, bilinear in albedo (and turbidity)
//   solar radiance: piecewise cubic in elevation (45 pieces, breaks at ((k/45)^3) pi/2) x limb darkening
// Reference call sites: sky_tungsten.cpp:433-502 (Skydome), hosek/ArHosekSkyModel.cpp:147-304, :402-561,
// :653-815 (model), host.cpp:276-333 (bake).  Arithmetic is double inside the model, float around it,
// as in the reference.
//
// The model evaluated here is the Hosek-Wilkie sky-dome / solar-radiance model; its evaluation routines (e.g. the solar
// polynomial below) necessarily follow the structure of the authors' reference implementation, which the upstream project
// vendors under the following licence (src/sunsky/hosek/ArHosekSkyModel.cpp:1-30):
//
//   Copyright (c) 2012 - 2013, Lukas Hosek and Alexander Wilkie.  All rights reserved.
//
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the
//   following conditions are met:
//     * Redistributions of source code must retain the above copyright notice, this list of conditions and the following
//       disclaimer.
//     * Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the following
//       disclaimer in the documentation and/or other materials provided with the distribution.
//     * None of the names of the contributors may be used to endorse or promote products derived from this software without
//       specific prior written permission.
//
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES,
//   INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE
//   DISCLAIMED.  IN NO EVENT SHALL THE COPYRIGHT HOLDERS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY,
//   OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA,
//   OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT
//   LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF
//   ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
#include "sky.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "data/sky_tables.inc"

namespace volpath
{
namespace
{
constexpr double kPiD = 3.141592653589793;
constexpr float  kPiF = 3.1415926535897932384626422832795028841971f;  // vecmath.h:9

// quintic Bernstein blend of 6 control values at t
inline double bezier5(const double* c, int stride, double t)
{
    double u = 1.0 - t;
    return std::pow(u, 5.0) * c[0] + 5.0 * std::pow(u, 4.0) * t * c[stride] + 10.0 * std::pow(u, 3.0) * std::pow(t, 2.0) * c[2 * stride] +
           10.0 * std::pow(u, 2.0) * std::pow(t, 3.0) * c[3 * stride] + 5.0 * u * std::pow(t, 4.0) * c[4 * stride] + std::pow(t, 5.0) * c[5 * stride];
}

inline double blackbody(double kelvin, double lambda)
{
    const double c1 = 3.74177 * 10E-17, c2 = 0.0143878;
    return (c1 / std::pow(lambda, 5.0)) * (1.0 / (std::exp(c2 / (lambda * kelvin)) - 1.0));
}

// solar spectrum the model was fitted with (Preetham et al., extended to the UV), per waveband
const double kOriginalSolar[11] = {7500.0, 12500.0, 21127.5, 26760.5, 30663.7, 27825.0, 25503.8, 25134.2, 23212.1, 21526.7, 19870.8};

inline double sky_internal(const double* cf, double theta, double gamma)
{
    const double cg   = std::cos(gamma), ct = std::cos(theta);
    const double expM = std::exp(cf[4] * gamma);
    const double rayM = cg * cg;
    const double mieM = (1.0 + cg * cg) / std::pow(1.0 + cf[8] * cf[8] - 2.0 * cf[8] * cg, 1.5);
    const double zen  = std::sqrt(ct);
    return (1.0 + cf[0] * std::exp(cf[1] / (ct + 0.01))) * (cf[2] + cf[3] * expM + cf[5] * rayM + cf[6] * mieM + cf[7] * zen);
}

inline double solar_piece(const HosekState& s, int wl, double elevation)
{
    const int pieces = 45, order = 4;
    int pos = (int)(std::pow(2.0 * elevation / kPiD, 1.0 / 3.0) * pieces);
    if (pos > 44) pos = 44;
    const double break_x = std::pow((double)pos / (double)pieces, 3.0) * (kPiD * 0.5);
    const double* c = kHosekSolar + wl * 180 + order * (pos + 1) - 1;
    double res = 0.0, x = elevation - break_x, xe = 1.0;
    for (int i = 0; i < order; ++i) { res += xe * *c--; xe *= x; }
    return res * s.corr_sun[wl];
}
}  // namespace

bool hosek_alienworld_init(HosekState& s, double solar_elevation, double solar_intensity, double kelvin, double turbidity,
                           double ground_albedo)
{
    if (turbidity != (double)VPH_SKY_TURBIDITY_ROW) return false;
    s.turbidity = turbidity; s.albedo = ground_albedo; s.elevation = solar_elevation;
    const double t = std::pow(solar_elevation / (kPiD / 2.0), 1.0 / 3.0);
    for (int wl = 0; wl < 11; ++wl)
    {
        const double* c0 = kHosekConfig + (wl * 2 + 0) * 54;
        const double* c1 = kHosekConfig + (wl * 2 + 1) * 54;
        for (int i = 0; i < 9; ++i)
            s.config[wl][i] = (1.0 - ground_albedo) * bezier5(c0 + i, 9, t) + ground_albedo * bezier5(c1 + i, 9, t);
        const double* r0 = kHosekRadiance + (wl * 2 + 0) * 6;
        const double* r1 = kHosekRadiance + (wl * 2 + 1) * 6;
        s.radiance[wl]   = (1.0 - ground_albedo) * bezier5(r0, 1, t) + ground_albedo * bezier5(r1, 1, t);
        const double owl = (320.0 + 40.0 * wl) * 10E-10;
        const double nsr = blackbody(kelvin, owl) * (3.19992 * 10E-11);
        s.corr_sun[wl]   = nsr / kOriginalSolar[wl];
    }
    double sum = 0.0;
    for (int i = 2; i < 11; i++) sum += s.corr_sun[i];
    const double ratio = sum / 9.0;
    const double terrestrial_radius = (0.51 * (kPiD / 180.0)) / 2.0;
    s.solar_radius = (std::sqrt(solar_intensity) * terrestrial_radius) / std::sqrt(ratio);
    for (int i = 0; i < 11; i++) s.corr_sky[i] = solar_intensity * s.corr_sun[i] / ratio;
    return true;
}

double hosek_radiance(const HosekState& s, double theta, double gamma, double wavelength)
{
    int low = (int)((wavelength - 320.0) / 40.0);
    if (low < 0 || low >= 11) return 0.0;
    double interp = std::fmod((wavelength - 320.0) / 40.0, 1.0);
    double v_low  = sky_internal(s.config[low], theta, gamma) * s.radiance[low] * s.corr_sky[low];
    if (interp < 1e-6) return v_low;
    double res = (1.0 - interp) * v_low;
    if (low + 1 < 11) res += interp * sky_internal(s.config[low + 1], theta, gamma) * s.radiance[low + 1] * s.corr_sky[low + 1];
    return res;
}

double hosek_solar_radiance(const HosekState& s, double theta, double gamma, double wavelength)
{
    const double elevation = (kPiD / 2.0) - theta;
    int    wl_low  = (int)((wavelength - 320.0) / 40.0);
    double wl_frac = std::fmod(wavelength, 40.0) / 40.0;
    if (wl_low == 10) { wl_low = 9; wl_frac = 1.0; }
    // turbidity 2 exactly: the blend weight of the neighbouring turbidity row is zero
    double direct = (1.0 - wl_frac) * solar_piece(s, wl_low, elevation) + wl_frac * solar_piece(s, wl_low + 1, elevation);
    double ld[6];
    for (int i = 0; i < 6; i++) ld[i] = (1.0 - wl_frac) * kHosekLimb[wl_low * 6 + i] + wl_frac * kHosekLimb[(wl_low + 1) * 6 + i];
    const double srs = std::sin(s.solar_radius);
    const double ar2 = 1 / (srs * srs);
    const double sg  = std::sin(gamma);
    double sc2 = 1.0 - ar2 * sg * sg;
    if (sc2 < 0.0) sc2 = 0.0;
    const double sc = std::sqrt(sc2);
    const double dark = ld[0] + ld[1] * sc + ld[2] * std::pow(sc, 2.0) + ld[3] * std::pow(sc, 3.0) + ld[4] * std::pow(sc, 4.0) +
                        ld[5] * std::pow(sc, 5.0);
    direct *= dark;
    return direct + hosek_radiance(s, theta, gamma, wavelength);
}

// ---------------------------------------------------------------------------------------- Skydome
Skydome::Skydome() : _temperature(5777.0f), _gammaScale(1.0f), _turbidity(2.0f), _intensity(100.0f) {}

float3 Skydome::sunDirection() const
{
    float st = sinf(_theta);
    return make_float3(sinf(_phi) * st, cosf(_theta), cosf(_phi) * st);  // sky_tungsten.h:29-33
}

void Skydome::prepare()
{
    if (_prepared) return;
    _sun = sunDirection();
    float elev = std::asin(clampf(_sun.y, -1.0f, 1.0f));
    // sky_tungsten.cpp:449-451.  Skydome's turbidity is the constant 2 of the reference (its constructor, no setter); should
    // that ever change, a sky silently baked from an uninitialised state would be worse than stopping here.
    if (!hosek_alienworld_init(_state, elev, _intensity, _temperature, _turbidity, 0.2f))
    {
        fprintf(stderr, "Skydome: no Hosek-Wilkie coefficients for turbidity %g (only the turbidity-2 rows are carried, data/sky_tables.inc)\n",
                (double)_turbidity);
        abort();
    }
    _prepared = true;
}

static inline float3 xyz_to_rgb(float3 c)
{
    return make_float3(3.240479f * c.x + -1.537150f * c.y + -0.498535f * c.z, -0.969256f * c.x + 1.875991f * c.y + 0.041556f * c.z,
                       0.055648f * c.x + -0.204043f * c.y + 1.057311f * c.z);
}

float3 Skydome::skyColor(const float3& direction, bool CEL)
{
    prepare();
    if (CEL && dot(direction, _sun) > 94.0f / sqrtf(94.0f * 94.0f + 0.45f * 0.45f)) return sunColor();
    float  theta = std::acos(direction.y);
    float  gamma = clampf(std::acos(clampf(dot(direction, _sun), -1.0f, 1.0f)) * _gammaScale, 0.0f, kPiF);
    float3 xyz   = make_float3(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < 7; ++i)  // NumSamplesValid, sky_tungsten.cpp:431
    {
        float  r = (float)hosek_radiance(_state, theta, gamma, kSkyLambda[i]);
        float3 w = make_float3(kSkyXyzWeight[i][0], kSkyXyzWeight[i][1], kSkyXyzWeight[i][2]);
        xyz      = xyz + w * r;
    }
    return xyz_to_rgb(xyz);
}

float3 Skydome::sunColor()
{
    float3 direction = sunDirection();
    prepare();
    float  theta = std::acos(direction.y);
    float  gamma = clampf(std::acos(clampf(dot(direction, _sun), -1.0f, 1.0f)) * _gammaScale, 0.0f, kPiF);
    float3 xyz   = make_float3(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < 7; ++i)
    {
        float  r = (float)hosek_solar_radiance(_state, theta, gamma, kSkyLambda[i]);
        float3 w = make_float3(kSkyXyzWeight[i][0], kSkyXyzWeight[i][1], kSkyXyzWeight[i][2]);
        xyz      = xyz + w * r;
    }
    return xyz_to_rgb(xyz) * _intensity;
}

// ------------------------------------------------------------------------------------------ bake
SunSky bake_sunsky(float x, float y, int width, int height)
{
    SunSky out;
    out.width = width; out.height = height;
    out.envmap.assign((size_t)width * height, make_float4(0, 0, 0, 0));
    y *= 0.5f;
    y = clampf(y, 0.0f, 0.49999f);
    Skydome s;
    s.setSunPhi(x * kPiF * 2);
    s.setSunTheta(y * kPiF);
    const float scale = 0.02f;  // sunsky_scale host.cpp:292
    float3 sun_dir   = s.getSunDir();
    float3 sun_power = s.sunColor() * scale;
    s.skyColor(make_float3(0, 1, 0));  // prepare the state once, outside the parallel loop
    const float disc = (float)((double)kPiF * (0.45 / 94.0f * 0.45 / 94.0f));  // host.cpp:319
#pragma omp parallel for schedule(static)
    for (int i = 0; i < width; i++)
    {
        Skydome local = s;
        for (int j = 0; j < height; j++)
        {
            if (j < height / 2)
            {
                float  phi   = float(i) / width * 2 * kPiF;
                float  theta = (float(j) / height) * kPiF;
                float3 d     = make_float3(sinf(theta) * sinf(phi), cosf(theta), sinf(theta) * -cosf(phi));
                float3 c     = local.skyColor(d, false);
                out.envmap[i + (size_t)j * width] = make_float4(c.x * scale, c.y * scale, c.z * scale, 1.0f * scale);
            }
            else
            {
                float3 ga = make_float3(0.01f, 0.01f, 0.01f);
                float3 r  = ((ga * sun_dir.y) * sun_power) * disc;
                out.envmap[i + (size_t)j * width] = make_float4(r.x, r.y, r.z, 1.0f);
            }
        }
    }
    out.sun_dir = sun_dir; out.sun_power = sun_power;
    return out;
}
}  // namespace volpath
