#include "camera.h"

// lookAt(eye, eye + fwd*focus, up) builds the orthonormal basis f = normalize(centre - eye),
// s = normalize(f x up), u = s x f; the inverse view matrix has columns (s, u, -f, eye); transposed and
// truncated to 3 rows it is  [s.x u.x -f.x eye.x; s.y u.y -f.y eye.y; s.z u.z -f.z eye.z].
void Camera::inv_view_matrix(float m[12]) const
{
    float3 centre = position + forward * focus_dist;
    float3 f      = normalize(centre - position);
    float3 s      = normalize(cross(f, up));
    float3 u      = cross(s, f);
    m[0] = s.x; m[1] = u.x; m[2]  = -f.x; m[3]  = position.x;
    m[4] = s.y; m[5] = u.y; m[6]  = -f.y; m[7]  = position.y;
    m[8] = s.z; m[9] = u.z; m[10] = -f.z; m[11] = position.z;
}
