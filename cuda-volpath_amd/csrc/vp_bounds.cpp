// vp_bounds.cpp -- local (max,min) density bounds for decomposition tracking.
//
// Replaces the host call-back of the reference (compute_volume_value_bound, host.cpp:1088-1280,
// called from init_cuda kernel.cu:389,406): per voxel, the max and min of the density over the
// (2r+1)^3 window centred on it, clipped to the grid, r = ceil(search_radius / (2/Nx)).
// brick > 1 is this build's coarser table (one entry per brick^3 voxels): the windows
// of all voxels of a brick merged.  Max/min filters are separable; each 1-D pass uses the
// van Herk / Gil-Werman block prefix-suffix scheme (3 compares per element, any r).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "vp_bounds.h"

namespace vp
{
int bound_radius(int nx, float search_radius)
{
    float cell_size = 2.0f / (float)nx;  // host.cpp:1098
    return (int)std::ceil(search_radius / cell_size);  // host.cpp:1101
}

namespace
{
// 1-D running max and min of window [o-r, o+r] over a strided line
template <typename T>
void filter_line(T* mx, T* mn, size_t stride, int n, int r, std::vector<T>& buf)
{
    if (r <= 0) return;
    const int w   = 2 * r + 1;
    const int pad = r;
    const int len = ((n + 2 * pad + w - 1) / w) * w;
    // layout of buf: in_max | g_max | h_max | in_min | g_min | h_min
    buf.resize((size_t)6 * len);
    T* im = buf.data();
    T* gm = im + len;
    T* hm = gm + len;
    T* in = hm + len;
    T* gn = in + len;
    T* hn = gn + len;
    const T lo = std::numeric_limits<T>::lowest(), hi = std::numeric_limits<T>::max();
    for (int i = 0; i < len; i++)
    {
        int s = i - pad;
        bool inside = s >= 0 && s < n;
        im[i] = inside ? mx[(size_t)s * stride] : lo;
        in[i] = inside ? mn[(size_t)s * stride] : hi;
    }
    for (int b = 0; b < len; b += w)
    {
        gm[b] = im[b]; gn[b] = in[b];
        for (int i = b + 1; i < b + w; i++) { gm[i] = std::max(gm[i - 1], im[i]); gn[i] = std::min(gn[i - 1], in[i]); }
        hm[b + w - 1] = im[b + w - 1]; hn[b + w - 1] = in[b + w - 1];
        for (int i = b + w - 2; i >= b; i--) { hm[i] = std::max(hm[i + 1], im[i]); hn[i] = std::min(hn[i + 1], in[i]); }
    }
    for (int o = 0; o < n; o++)
    {
        int a = o, b = o + 2 * r;  // padded indices of o-r and o+r
        mx[(size_t)o * stride] = std::max(hm[a], gm[b]);
        mn[(size_t)o * stride] = std::min(hn[a], gn[b]);
    }
}

template <typename T>
void build(const T* grid, int nx, int ny, int nz, int radius, int brick, T* out)
{
    const size_t n = (size_t)nx * ny * nz;
    std::vector<T> mx(grid, grid + n), mn(grid, grid + n);
    const size_t sx = 1, sy = (size_t)nx, sz = (size_t)nx * ny;
#pragma omp parallel
    {
        std::vector<T> buf;
#pragma omp for schedule(static)
        for (long long l = 0; l < (long long)ny * nz; l++)  // x lines
            filter_line(mx.data() + (size_t)l * nx, mn.data() + (size_t)l * nx, sx, nx, radius, buf);
#pragma omp for schedule(static)
        for (long long l = 0; l < (long long)nx * nz; l++)  // y lines
        {
            size_t i = (size_t)(l % nx), k = (size_t)(l / nx);
            filter_line(mx.data() + i + k * sz, mn.data() + i + k * sz, sy, ny, radius, buf);
        }
#pragma omp for schedule(static)
        for (long long l = 0; l < (long long)nx * ny; l++)  // z lines
            filter_line(mx.data() + (size_t)l, mn.data() + (size_t)l, sz, nz, radius, buf);
    }
    const int bnx = (nx + brick - 1) / brick, bny = (ny + brick - 1) / brick, bnz = (nz + brick - 1) / brick;
#pragma omp parallel for schedule(static)
    for (long long bl = 0; bl < (long long)bny * bnz; bl++)
    {
        int bj = (int)(bl % bny), bk = (int)(bl / bny);
        for (int bi = 0; bi < bnx; bi++)
        {
            T m1 = std::numeric_limits<T>::lowest(), m0 = std::numeric_limits<T>::max();
            for (int k = bk * brick; k < std::min((bk + 1) * brick, nz); k++)
                for (int j = bj * brick; j < std::min((bj + 1) * brick, ny); j++)
                    for (int i = bi * brick; i < std::min((bi + 1) * brick, nx); i++)
                    {
                        size_t idx = (size_t)i + sy * j + sz * k;
                        m1 = std::max(m1, mx[idx]);
                        m0 = std::min(m0, mn[idx]);
                    }
            size_t o = ((size_t)bi + (size_t)bnx * ((size_t)bj + (size_t)bny * bk)) * 2;
            out[o]     = m1;  // .x = max, .y = min (host.cpp:1141-1144)
            out[o + 1] = m0;
        }
    }
}
}  // namespace

void build_bounds_u8(const uint8_t* grid, int nx, int ny, int nz, int radius, int brick, uint8_t* out)
{
    build<uint8_t>(grid, nx, ny, nz, radius, brick, out);
}
void build_bounds_f32(const float* grid, int nx, int ny, int nz, int radius, int brick, float* out)
{
    build<float>(grid, nx, ny, nz, radius, brick, out);
}
}  // namespace vp
