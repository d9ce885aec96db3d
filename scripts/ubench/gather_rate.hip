// How many random 8-byte gathers per second does the memory system deliver when every gather is its own 128-byte line?
// The ceiling of c4f's cell fetches (512^3 cells x 8 B = 1 GiB, beyond L2 and the 256 MiB Infinity Cache): render_k makes ONE
// dependent load per lane and step, its waves cover each other.  Here: U independent loads per lane in flight, W waves per SIMD.
//   ./gather_rate            -> table of G lines/s and TB/s (at 128 B per line) over buffer size x loads in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <int U>
__global__ __launch_bounds__(256) void gather_k(const uint2* buf, unsigned mask, unsigned long long* out, int reps)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0, h = hash32(tid * 2654435761u + 12345u);
    for (int r = 0; r < reps; r++)
    {
        uint2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { h = hash32(h + acc * 0u + 0x9e3779b9u * (u + 1)); v[u] = buf[h & mask]; }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x + v[u].y;
        h ^= acc;   // the next round's addresses depend on this round's data: U loads in flight per lane, no more
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t maxb = (size_t)4 << 30;
    uint2* buf; unsigned long long* out;
    CHK(hipMalloc(&buf, maxb)); CHK(hipMalloc(&out, 8));
    CHK(hipMemset(buf, 0, maxb)); CHK(hipMemset(out, 0, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int threads = 256, reps = 256;
    printf("random 8-byte gathers, one 128-byte line each; rows: buffer size, columns: (workgroups per CU) x (loads in flight per lane)\n");
    const size_t sizes[] = {(size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30};
    for (size_t bytes : sizes)
    {
        const unsigned mask = (unsigned)(bytes / 8 - 1);
        printf("%5zu MiB:", bytes >> 20);
        for (int bpc : {4, 6, 8})
            for (int U : {1, 2, 4})
            {
                const int blocks = 256 * bpc;
                const double loads = (double)blocks * threads * reps * U;
                float best = 1e30f;
                for (int t = 0; t < 3; t++)
                {
                    CHK(hipEventRecord(e0));
                    if (U == 1) hipLaunchKernelGGL(gather_k<1>, dim3(blocks), dim3(threads), 0, 0, buf, mask, out, reps);
                    if (U == 2) hipLaunchKernelGGL(gather_k<2>, dim3(blocks), dim3(threads), 0, 0, buf, mask, out, reps);
                    if (U == 4) hipLaunchKernelGGL(gather_k<4>, dim3(blocks), dim3(threads), 0, 0, buf, mask, out, reps);
                    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("  %dx%d %5.1f G/s", bpc, U, loads / best / 1e6);
            }
        printf("\n");
    }
    return 0;
}
