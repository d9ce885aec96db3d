// camera.h -- the reference's orbit camera reduced to what the integrator consumes: the row-major 3x4
// camera-to-world matrix that host.cpp:617-623 builds with glm::lookAt -> inverse -> transpose.
#ifndef VOLPATH_HOST_CAMERA_H
#define VOLPATH_HOST_CAMERA_H
#include "vec.h"

struct Camera
{
    // defaults of host.cpp:108-112
    float3 position   = {3.922986f, -0.782739f, 0.030000f};
    float3 forward    = {-0.978148f, 0.207912f, 0.000000f};
    float3 up         = {0.207912f, 0.978148f, -0.000000f};
    float  focus_dist = 4.0f;
    // m[12]: rows (right, up', -forward') with the eye position in the 4th column
    void inv_view_matrix(float m[12]) const;
};
#endif
