// Calibration of rocprofv3's FETCH_SIZE for THIS project's access pattern: one 8-byte load per lane at a
// pseudo-random 8-byte-aligned address (the density-cell gather of render_k).  Known byte counts:
//   N loads, each from its own random cache line of a buffer far larger than L2 + Infinity Cache
//   => N x 32 B (sector), N x 64 B or N x 128 B (line) must show up, whichever unit the counter tallies.
// Usage: rocprofv3 --pmc FETCH_SIZE -d out -- ./gather_calib ; compare FETCH_SIZE*1024 with the printed counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// mode 0: random 8-byte gathers (one per lane, distinct lines with overwhelming probability)
// mode 1: the same number of loads, but 8 consecutive lanes share one 64-byte block (coalesced 8-byte loads)
// mode 2: wide streaming read, 16 B per lane (the guide's calibrated case: FETCH_SIZE reports half)
// mode 3: 16 lanes cover one random 128-byte line; mode 4: two lanes touch the two 64-byte halves of one random line
__global__ void gather_k(const uint2* buf, size_t ncell, unsigned long long* out, int mode, int reps)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (int r = 0; r < reps; r++)
    {
        size_t idx;
        if (mode == 0) idx = ((size_t)hash32(tid * 31u + r) * 8191u + hash32(tid ^ (r * 0x9e3779b9u))) % ncell;
        else if (mode == 1) idx = ((((size_t)hash32((tid >> 3) * 31u + r) * 8191u) % (ncell / 8)) * 8) + (tid & 7u);
        else if (mode == 3) idx = ((((size_t)hash32((tid >> 4) * 31u + r) * 8191u) % (ncell / 16)) * 16) + (tid & 15u);
        else if (mode == 4) idx = ((((size_t)hash32((tid >> 1) * 31u + r) * 8191u) % (ncell / 16)) * 16) + (tid & 1u) * 8u;
        else idx = ((size_t)r * gridDim.x * blockDim.x + tid) * 2 % (ncell - 1);
        if (mode == 2)
        {
            uint4 v = *reinterpret_cast<const uint4*>(buf + (idx & ~(size_t)1));
            acc += v.x + v.y + v.z + v.w;
        }
        else
        {
            uint2 v = buf[idx];
            acc += v.x + v.y;
        }
    }
    if (acc == 0x123456789abcull) out[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)4 << 30;  // 4 GiB >> 256 MiB Infinity Cache
    const size_t ncell = bytes / 8;
    uint2* buf; unsigned long long* out;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&out, 8));
    CHK(hipMemset(buf, 1, bytes)); CHK(hipMemset(out, 0, 8));
    CHK(hipDeviceSynchronize());
    const int blocks = 256 * 8, threads = 256, reps = 64;
    const double loads = (double)blocks * threads * reps;
    for (int mode = 0; mode < 5; mode++)
    {
        hipLaunchKernelGGL(gather_k, dim3(blocks), dim3(threads), 0, 0, buf, ncell, out, mode, reps);
        CHK(hipDeviceSynchronize());
        if (mode == 0) printf("mode 0 random 8-byte gathers : %.0f loads = %.3f MB useful, %.3f MB at 32 B, %.3f MB at 64 B, %.3f MB at 128 B per load\n", loads, loads * 8 / 1e6, loads * 32 / 1e6, loads * 64 / 1e6, loads * 128 / 1e6);
        if (mode == 1) printf("mode 1 8 lanes per 64 B block : %.0f loads = %.3f MB useful = %.3f MB in 64 B blocks, %.3f MB in 128 B lines\n", loads, loads * 8 / 1e6, loads * 8 / 1e6, loads * 16 / 1e6);
        if (mode == 3) printf("mode 3 16 lanes cover one random 128 B line: %.0f lines = %.3f MB\n", loads / 16, loads * 8 / 1e6);
        if (mode == 4) printf("mode 4 lane pairs touch both 64 B halves of one random 128 B line: %.0f lines, %.3f MB at 128 B per line\n", loads / 2, loads / 2 * 128 / 1e6);
        if (mode == 2) printf("mode 2 streaming 16 B per lane: %.0f loads = %.3f MB\n", loads, loads * 16 / 1e6);
    }
    return 0;
}
