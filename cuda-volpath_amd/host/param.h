// param.h -- the render parameter block handed to the integrator, layout-identical to the reference's
// src/param.h:4-12 (44 bytes: width, height, density, brightness, albedo rgb, g, sigma_t rgb).
// The C ABI header (include/volpath.h) carries the same struct; this file gives the C++ host the
// reference's spelling `Param` with float3 members.
#ifndef VOLPATH_HOST_PARAM_H
#define VOLPATH_HOST_PARAM_H

#define VOLPATH_PARAM_DEFINED
#include "vec.h"

struct Param
{
    unsigned int width = 0, height = 0;
    float        density = 0.0f, brightness = 0.0f;
    float3       albedo{};
    float        g = 0.0f;
    float3       sigma_t{};
};
static_assert(sizeof(Param) == 44, "Param must stay layout-compatible with the reference kernels");

// Mat(), host.cpp:44-57: sigma_s (X,Y,Z) and sigma_a (R,G,B) -> sigma_t normalised by its max, albedo
void set_material(Param& P, float X, float Y, float Z, float R, float G, float B);
// the 13 presets pushed by main(), host.cpp:1296-1308; index 0..12
bool material_preset(Param& P, int index);
// defaults of main(), host.cpp:1286-1292
Param default_param(unsigned width = 960, unsigned height = 512);

#endif
