// vp_bounds.h -- CPU builder of the local (max,min) bound table (see vp_bounds.cpp)
#pragma once
#include <cstdint>
namespace vp
{
int  bound_radius(int nx, float search_radius);
void build_bounds_u8(const uint8_t* grid, int nx, int ny, int nz, int radius, int brick, uint8_t* out);
void build_bounds_f32(const float* grid, int nx, int ny, int nz, int radius, int brick, float* out);
}  // namespace vp
