// vec.h -- the few vector PODs the host needs at the C ABI seam (layout of CUDA/HIP float3/float4)
#ifndef VOLPATH_HOST_VEC_H
#define VOLPATH_HOST_VEC_H
#include <cmath>
#include <cstddef>

struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };

inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float  dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float3 normalize(float3 v) { float inv = 1.0f / std::sqrt(dot(v, v)); return v * inv; }
inline float  clampf(float x, float a, float b) { return x < a ? a : (x > b ? b : x); }

#endif
