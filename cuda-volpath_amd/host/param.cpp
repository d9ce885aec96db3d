#include "param.h"

#include <algorithm>

void set_material(Param& P, float X, float Y, float Z, float R, float G, float B)
{
    P.sigma_t = make_float3(X + R, Y + G, Z + B);
    P.albedo  = make_float3(X / P.sigma_t.x, Y / P.sigma_t.y, Z / P.sigma_t.z);
    float f   = std::max(std::max(P.sigma_t.x, P.sigma_t.y), P.sigma_t.z);
    P.sigma_t.x /= f;
    P.sigma_t.y /= f;
    P.sigma_t.z /= f;
}

bool material_preset(Param& P, int index)
{
    // scattering / absorption coefficient table used by the reference (host.cpp:1296-1308)
    static const float T[13][6] = {
        {2.29f, 2.39f, 1.97f, 0.0030f, 0.0034f, 0.046f},  {0.15f, 0.21f, 0.38f, 0.015f, 0.077f, 0.19f},
        {0.19f, 0.25f, 0.32f, 0.018f, 0.088f, 0.20f},     {7.38f, 5.47f, 3.15f, 0.0002f, 0.0028f, 0.0163f},
        {0.18f, 0.07f, 0.03f, 0.061f, 0.97f, 1.45f},      {2.19f, 2.62f, 3.00f, 0.0021f, 0.0041f, 0.0071f},
        {0.68f, 0.70f, 0.55f, 0.0024f, 0.0090f, 0.12f},   {0.70f, 1.22f, 1.90f, 0.0014f, 0.0025f, 0.0142f},
        {0.74f, 0.88f, 1.01f, 0.032f, 0.17f, 0.48f},      {1.09f, 1.59f, 1.79f, 0.013f, 0.070f, 0.145f},
        {11.6f, 20.4f, 14.9f, 0.0f, 0.0f, 0.0f},          {2.55f, 3.21f, 3.77f, 0.0011f, 0.0024f, 0.014f},
        {1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f}};
    if (index < 0 || index >= 13) return false;
    set_material(P, T[index][0], T[index][1], T[index][2], T[index][3], T[index][4], T[index][5]);
    return true;
}

Param default_param(unsigned width, unsigned height)
{
    Param P;
    P.brightness = 1.0f;
    P.width      = width;
    P.height     = height;
    P.albedo     = make_float3(1.0f, 1.0f, 1.0f);
    P.g          = 0.877f;
    P.density    = 800.0f;
    P.sigma_t    = make_float3(1.0f, 1.0f, 1.0f);
    return P;
}
