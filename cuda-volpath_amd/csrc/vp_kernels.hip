// vp_kernels.hip -- gfx950 kernels of the volumetric radiance integrator.
//
// render_k: persistent wave64 workgroups pull (frame, pixel) samples from per-XCD queues with one
// atomic per 256 samples (ballot + mbcnt compaction), so a lane whose path ended is refilled while
// its neighbours keep tracking; every path carries its own RNG state and therefore computes the
// same bits wherever and whenever it runs.  A wave alternates between an inner loop of free-flight
// steps (four per pass; segment set-up included for the local-majorant estimators) and an event
// pass that serves the lanes parked on collisions, exits and refills.  With the counter-based streams a collision takes ONE
// visit (shadow rays draw from sub-streams: the phase function is sampled in the collision block, the light is added where the
// shadow ray ends) and a sun shadow ray ends where only empty cells are left (sun_clip_k).  What depends on a pixel's camera
// ray alone (the same ray in every frame) is tabulated once per pixel -- the restart crawl in front of the volume, the
// distance up to which the ray meets only empty cells, the pixel's class -- and pixels whose ray never meets a non-empty
// cell are per-pixel constants where a null collision in empty space leaves the throughput at exactly 1 (light_identity_k,
// miss_fill_k), else they run the LIGHT specialisation of the kernel beside the general one.
// approach_k / approach_local_k: in staged launches the remaining pixels' camera rays are walked through their certified-empty
// stretch -- draw, logarithm, add, compare -- by a thread per sample ahead of render_k, which takes each path up from its
// staging slot: work sorted by kind across the chip instead of lanes waiting beside lanes that fetch.
// Exit flights (round 4): a path in empty space that can meet empty cells only on its way out of the box (a per-cell table of 24
// direction classes, exit_dir_slice_k) and whose null collision multiplies its throughput by exactly 1.0f ends with the environment
// whatever it draws -- it is ended at once, tested for free-riding lanes whenever their wave is in the event pass anyway.
// Restates
//   __d_render_bounded_decomp  kernel.cu:1958-2318  (EST_DECOMP, the reference's live kernel)
//   __d_render                 kernel.cu:1285-1591  (EST_GLOBAL, BASELINE config 2)
//   __d_render_bounded         kernel.cu:1667-1952  (EST_BOUNDED, dead in the reference)
// under SPECTRAL_TRACKING=1, SUN_LIGHT=1, PRECOMPUTE_OPACITY=1 (kernel.cu:15-34), with PASSIVE_ENVMAP=1
// (MIS = false, the shipped build) or 0 (MIS = true).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "vp_device.h"
#include "vp_kernels.h"

namespace vp
{
// per-lane path states.  FAST states advance inside the inner tracking loop; the others are events
// a lane parks in until the wave runs its slow path.
enum : int
{
    ST_DONE    = 0,  // no path: wants a new sample from the queue
    ST_SETUP   = 1,  // needs a segment set-up (FAST for the decomposition estimator)
    ST_TRACK   = 2,  // FAST: free-flight steps of the primary ray
    ST_SHADOW  = 3,  // FAST: free-flight steps of the sun shadow ray (Tr_spectral)
    EV_SCATTER = 4,  // collision found: direct-lighting set-up
    EV_NEE     = 5,  // shadow transmittance known: add sun light, sample the phase function
    EV_BG      = 6,  // ray left the medium: environment lookup
    EV_WRITE   = 7,  // path finished: emit the sample
    EV_MIS     = 8,  // active environment sampling only: draw the one-sample-MIS direction, start its shadow ray
    EV_HG      = 9   // sample the phase function, continue the path
};

// Perturbation profiling (round 5; scripts/r05_pad_profile.sh, profiles/r05_pad_profile.md): no per-PC sampling is to be had on this
// pool, so the marginal cost of a code block is measured by ADDING work to it: -DVP_PAD_<BLOCK>=N puts N dependent v_fma_f32 (the
// cheapest vector instruction: 1.06 ns per wave-instruction per SIMD at saturation) on a scratch register into that block of
// render_k; the launch-time difference against the unpadded build, divided by the block's executions, is what one more
// instruction there costs -- the full issue slot if the vector pipe is the bound, less if the block runs in the shadow of waits.
// All zero in the shipped build: vp_pad<0> is empty.
#ifndef VP_PAD_STEP
#define VP_PAD_STEP 0
#endif
#ifndef VP_PAD_FETCH
#define VP_PAD_FETCH 0
#endif
#ifndef VP_PAD_EOF
#define VP_PAD_EOF 0
#endif
#ifndef VP_PAD_SETUP
#define VP_PAD_SETUP 0
#endif
#ifndef VP_PAD_COLL
#define VP_PAD_COLL 0
#endif
#ifndef VP_PAD_END
#define VP_PAD_END 0
#endif
template <int N>
__device__ __forceinline__ void vp_pad()
{
    if constexpr (N > 0)
    {
#ifdef VP_PAD_INDEPENDENT   // four independent chains: issue cost without the dependent-issue latency of one chain
        float t0 = 1.0f, t1 = 1.0f, t2 = 1.0f, t3 = 1.0f;
#pragma unroll
        for (int i = 0; i < N / 4; i++)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
#else
        float t = 1.0f;
#pragma unroll
        for (int i = 0; i < N; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(t));
#endif
    }
}

__device__ __forceinline__ unsigned lane_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Cold per-path state (round 4).  A path's radiance sum, the direction it resumes after a shadow ray, its sample
// coordinates, scatter count, phase-function parameters and the stream position saved across a shadow ray are touched in the event
// blocks and where a shadow ray ends -- never in the free-flight step.  In the plain kernels they live in LDS, one word per lane
// and field (14 words: 14 KB per workgroup), so that the kernel fits SIX waves per SIMD (80 registers) where it held five:
// +4 % on both estimators (profiles/experiments/r04_cold_state_in_lds.txt).  ColdVal / ColdF3 are that word or triple in LDS
// (COLD = true) or an ordinary variable (the LDS-table kernel, whose LDS is the table's, and the light kernels, which fit seven
// or eight waves as they are).
template <class T, bool COLD>
struct ColdVal;
template <class T>
struct ColdVal<T, true>
{
    static_assert(sizeof(T) == sizeof(float), "one LDS word per lane and field");
    float* p;   // the word is a float in LDS whatever T is: values go through __builtin_bit_cast, not through a punned pointer
    __device__ __forceinline__ explicit ColdVal(float* q) : p(q) {}
    __device__ __forceinline__ operator T() const { return __builtin_bit_cast(T, *p); }
    __device__ __forceinline__ ColdVal& operator=(T v) { *p = __builtin_bit_cast(float, v); return *this; }
};
template <class T>
struct ColdVal<T, false>
{
    T v = T();
    __device__ __forceinline__ explicit ColdVal(float*) {}
    __device__ __forceinline__ operator T() const { return v; }
    __device__ __forceinline__ ColdVal& operator=(T w) { v = w; return *this; }
};
template <bool COLD, int STRIDE>
struct ColdF3;
template <int STRIDE>
struct ColdF3<true, STRIDE>
{
    float* p;
    __device__ __forceinline__ explicit ColdF3(float* q) : p(q) {}
    __device__ __forceinline__ operator f3() const { return f3{p[0], p[STRIDE], p[2 * STRIDE]}; }
    __device__ __forceinline__ ColdF3& operator=(f3 v) { p[0] = v.x; p[STRIDE] = v.y; p[2 * STRIDE] = v.z; return *this; }
};
template <int STRIDE>
struct ColdF3<false, STRIDE>
{
    f3 v = {};
    __device__ __forceinline__ explicit ColdF3(float*) {}
    __device__ __forceinline__ operator f3() const { return v; }
    __device__ __forceinline__ ColdF3& operator=(f3 w) { v = w; return *this; }
};
// One null collision of the spectral tracker where the density is +0 (kernel.cu:2107-2134 with sigma_t_den = +0: Ps = +0, c = Pn,
// `real` false for any draw, sigma_null_den = sigma_t'), for a throughput with three equal channels t:
// Pn = (m + m) + m with m = |sigma_t' t|, t *= sigma_t' * ((inv_sigma_t * Pn) / Pn).  The factor is 1 up to rounding, not exactly.
__device__ __forceinline__ float null_collision_in_empty_space(float t, float sigma_t_prime, float inv_sigma_t)
{
    float mn = __builtin_fabsf(sigma_t_prime * t);
    float Pn = (mn + mn) + mn;
    return t * (sigma_t_prime * wdiv_(inv_sigma_t * Pn, Pn));
}
// Does a null collision where the density is +0 leave the throughput (t.x, t.y, t.z) bit for bit as it is, whatever is drawn?  The
// spectral tracker's expressions (kernel.cu:2107-2134; tracking_step below) with sigma_t_den = +0 -- sigma_c = 0: no control
// component where the cells are empty --: Ps = |0 t.x| + |0 t.y| + |0 t.z| must be +0 (then `real`, e * c < Ps, is false for any
// draw e), and the factor every channel is multiplied by, sigma_null * f = sigma_t' * ((inv_sigma_t * c) / Pn), must be exactly 1.
// The one-channel (ACH) form of the step computes the same sums ((m + m) + m) from one product.
__device__ __forceinline__ bool null_collision_is_identity(f3 t, float sigma_t_prime, float inv_sigma_t)
{
    const float Ps = __builtin_fabsf(0.0f * t.x) + __builtin_fabsf(0.0f * t.y) + __builtin_fabsf(0.0f * t.z);
    const float Pn = __builtin_fabsf(sigma_t_prime * t.x) + __builtin_fabsf(sigma_t_prime * t.y) + __builtin_fabsf(sigma_t_prime * t.z);
    const float c  = Ps + Pn;
    return Ps == 0.0f && sigma_t_prime * wdiv_(inv_sigma_t * c, Pn) == 1.0f;
}
// table[n] = throughput of an unscattered path of the global-majorant estimator after n null collisions in empty space: it
// starts at (1,1,1) and every sample has the same sigma_t' (segment set-up of __d_render with no scatter behind it,
// kernel.cu:1355-1366), so the sequence is the same for every sample of a launch.  One thread, `count` dependent steps.
__global__ void thr_table_k(ParamDev P, float* table, unsigned count)
{
    if (threadIdx.x || blockIdx.x) return;
    const float max_sig       = max3(f3{P.sigma_t[0], P.sigma_t[1], P.sigma_t[2]});
    const float s             = hyperion_s(0 - 5);
    const float cur_density   = (1.0f - s) * P.density + s * P.density * (1.0f - P.g);
    const float sigma_t_prime = max_sig * cur_density;
    const float inv_sigma_t   = 1.0f / sigma_t_prime;
    float t = 1.0f;
    for (unsigned n = 0; n < count; n++)
    {
        table[n] = t;
        t = null_collision_in_empty_space(t, sigma_t_prime, inv_sigma_t);
    }
}

// ---- when the samples of the light class do not depend on the draws.
// A light path's throughput starts at (1,1,1) and changes only through null collisions in empty space, each a multiplication by
// sigma_t' * ((1/sigma_t' * Pn) / Pn) with Pn = 3 |sigma_t' t|.  That factor is exactly 1 at t = 1 for most sigma_t' (800, the
// default, among them) -- and then the throughput stays exactly 1 whatever the number of collisions, i.e. whatever is drawn: the
// sample is the environment seen along the camera ray, a per-pixel constant like the samples of a ray that misses the box
// (quirk Q3), and miss_fill_k writes it.  This kernel decides that with the device's own arithmetic: global majorant: the one
// sigma_t' of an unscattered path (thr_table_k); decomposition estimator: sigma_t' = max_sig * density * max(0.0001, b / 255) for
// every byte b that occurs as a maximum in the bound table (bound_bytes_k's mask).  flag[0] = 1 if the factor is 1 for all of them.
__global__ void bound_bytes_k(const unsigned short* bounds, size_t n, unsigned* mask)
{
    __shared__ unsigned m[8];
    if (threadIdx.x < 8) m[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const unsigned b = bounds[i] & 0xffu;   // the maximum (PairU8)
        if (!(m[b >> 5] >> (b & 31u) & 1u)) atomicOr(&m[b >> 5], 1u << (b & 31u));
    }
    __syncthreads();
    if (threadIdx.x < 8 && m[threadIdx.x]) atomicOr(&mask[threadIdx.x], m[threadIdx.x]);
}
__global__ void light_identity_k(ParamDev P, int local, const unsigned* mask, unsigned* flag)
{
    const unsigned b = threadIdx.x;   // 256 threads
    if (local && !(mask[b >> 5] >> (b & 31u) & 1u)) return;
    if (!local && b) return;
    const float max_sig = max3(f3{P.sigma_t[0], P.sigma_t[1], P.sigma_t[2]});
    const float s       = hyperion_s(0 - 5);
    float sigma_t_prime;
    if (local)
    {
        const float reduction   = (1.0f - s) + s * (1.0f - P.g);           // segment_medium(), no scatter behind the path
        const float cur_density = reduction * P.density;
        const float d_max       = fmaxf(0.0001f, (float)b * VP_U8_SCALE);   // segment_setup()
        sigma_t_prime           = max_sig * cur_density * d_max;
    }
    else
    {
        const float cur_density = (1.0f - s) * P.density + s * P.density * (1.0f - P.g);   // thr_table_k
        sigma_t_prime           = max_sig * cur_density;
    }
    const float inv_sigma_t = 1.0f / sigma_t_prime;
    if (!(null_collision_in_empty_space(1.0f, sigma_t_prime, inv_sigma_t) == 1.0f)) atomicAnd(flag, 0u);
}

// LDSB: the (max,min) brick table of the decomposition estimator is staged through LDS (BASELINE config 3:
// 256^3 / 8^3 bricks = 32768 byte pairs = 64 KiB).  Those workgroups are 512 threads so that two of them
// (2 x 64 KiB of the CU's 160 KiB) keep 16 waves per CU resident.
// ACH: achromatic medium (sigma_t and albedo equal in all three channels): the three throughput channels stay
// bitwise identical (same operations on the same values), so one is carried and the collision sums are
// formed from one product, (m + m) + m, exactly as the three-channel expression evaluates.
// MIS: active environment sampling with one-sample MIS after the sun estimate (the reference's !PASSIVE_ENVMAP
// build, kernel.cu:2220-2297); the shipped configuration is passive (MIS = false).
// TRK: 0 = spectral tracking (SPECTRAL_TRACKING 1, the shipped build); 1 = scalar tracking (SPECTRAL_TRACKING 0);
// 2 = MULTI_CHANNEL 1: scalar tracking of one colour channel drawn per sample (kernel.cu:15-34, :1993-1994, :2311-2313).
// LIGHT: the kernel of the "light" pixel class (spectral tracking): every camera ray of those pixels either misses the box or
// meets certified-empty cells over its whole chord (empty_table_k / crawl_table_k), so a path is: restart segments (local-majorant
// estimators) and free-flight steps whose null collisions have den = +0, then the environment.  No fetch code, no collision,
// shadow or phase states.
#ifndef VP_LOCAL_MIN_WAVES
#define VP_LOCAL_MIN_WAVES 6   // plain achromatic local-majorant kernels, cold state in LDS: 80 registers, no spill.  The chromatic ones would
                               // spill four or five at six waves and run as the LDS-table kernel's helper workgroups, where a fifth wave is
                               // all a SIMD has room for: they keep five
#endif
// Waves per SIMD the register budget of a render_k instance is held to (the second argument of its __launch_bounds__), one case per
// line instead of the nested conditional it used to be; the reasons are in the comment at the kernel:
constexpr int render_min_waves(int est, bool count, int ldsb, bool ach, bool mis, int trk, bool light, bool cancel)
{
    if (VP_MIN_WAVES > 1) return VP_MIN_WAVES;                                            // a build-wide override
    if (light) return count ? 5 : (est != EST_GLOBAL ? VP_LIGHT_LOCAL_MIN_WAVES : VP_LIGHT_MIN_WAVES);
    if (mis || ldsb == 1) return 1;                                                        // the MIS build; the 16-bit LDS table (512-thread workgroups)
    if (count) return 4;                                                                   // counting variants: untimed, no spills
    if (est == EST_GLOBAL && trk == 0) return VP_GLOBAL_MIN_WAVES;
    if (trk) return 4;                                                                     // scalar tracking builds
    return (ach && !cancel) ? VP_LOCAL_MIN_WAVES : 5;                                      // local majorants: achromatic six, chromatic / look-ahead five
}
template <int EST, class RNG, bool QUANT, bool COUNT, int LDSB, bool ACH, bool MIS, int TRK, bool LIGHT = false, bool CANCEL = false>
// Occupancy (round 4: the cold per-path state in LDS, ColdVal above; profiles/r04_kernel_resources.txt).  The achromatic
// global-majorant kernel needs 72 registers: SEVEN waves per SIMD (C2 2541 -> 2781 Msamples/s); the chromatic one and the plain
// achromatic local-majorant kernels 80: six (c3ref 2398 -> 2513); the LDS-table kernel keeps its state in registers (its LDS is the
// table's): 98, four waves and the helper workgroup's fifth.  Before, with everything in registers: 91-96, five waves (six cost
// three spilled registers and lost).
// (CANCEL instances of the local-majorant kernels: look-ahead batches are launched with five workgroups per CU -- vp_render.cpp -- so five
// waves are what their registers are budgeted for: no spill.)
__global__ __launch_bounds__(LDSB == 1 ? VP_BLOCK_LDS : VP_BLOCK, render_min_waves(EST, COUNT, LDSB, ACH, MIS, TRK, LIGHT, CANCEL))
void render_k(SceneDev S, LaunchDev L)
{
    __shared__ unsigned short lds_bounds[LDSB == 1 ? VP_LDS_BOUND_ENTRIES : 1];
    // LDSB == 2 (round 5): the table as 2-bit CODES into a palette of at most four distinct (max,min) pairs -- a binary volume has three:
    // (0,0), (255,0), (255,255) -- 8 KiB instead of 64: small enough to sit BESIDE the cold per-path state of a 256-thread workgroup, so
    // this kernel keeps the plain kernel's registers and occupancy (six / five waves per SIMD) where the 16-bit table's costs a third
    // of both (98-102 registers, four waves and a helper workgroup)
    __shared__ unsigned lds_codes[LDSB == 2 ? VP_LDS_BOUND_ENTRIES / 16 : 1];
    if (LDSB == 2)
    {
        const uint4* src = reinterpret_cast<const uint4*>(L.bound_codes);
        uint4*       dst = reinterpret_cast<uint4*>(lds_codes);
        const int    n16 = (S.bnx * S.bny * S.bnz + 63) / 64;   // 64 codes per 16 bytes; the device table is padded
        for (int w = threadIdx.x; w < n16; w += VP_BLOCK) dst[w] = src[w];
        __syncthreads();
    }
    if (LDSB == 1)
    {
        // coalesced 16-byte loads of the table, 16-byte LDS stores
        const uint4* src = reinterpret_cast<const uint4*>(S.bounds_u8);
        uint4*       dst = reinterpret_cast<uint4*>(lds_bounds);
        const int    n16 = (S.bnx * S.bny * S.bnz * 2 + 15) / 16;  // the device table is padded to 16 bytes
        for (int w = threadIdx.x; w < n16; w += VP_BLOCK_LDS) dst[w] = src[w];
        __syncthreads();
    }
    // The kernel arguments once more, in LDS (round 5).  The event section and light_done() read their uniforms -- camera, sun,
    // environment, queue and image descriptors: ~70 scalars the tracking loop never touches -- afresh at every visit instead of holding
    // them in SGPRs across the loop (round 3).  They did so with flat loads of the argument segment: L1/L2 round trips of their own, and
    // every s_waitcnt vmcnt(0) behind one also waits for the wave's density fetches.  An LDS read waits for itself only.
    // VP_EXP_FLAT_KARGS: the flat loads, for the A/B (profiles/experiments/r05_kargs_lds.txt).
    constexpr unsigned KARG_L_   = ((sizeof(SceneDev) + alignof(LaunchDev) - 1) / alignof(LaunchDev)) * alignof(LaunchDev);
    constexpr unsigned KARG_WDS_ = (KARG_L_ + sizeof(LaunchDev) + 3) / 4;
    __shared__ __attribute__((aligned(16))) unsigned kargs_lds_[KARG_WDS_];
    {
        const unsigned* src = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
        for (unsigned w = threadIdx.x; w < KARG_WDS_; w += (LDSB == 1 ? VP_BLOCK_LDS : VP_BLOCK)) kargs_lds_[w] = src[w];
        __syncthreads();
    }
    constexpr bool LOCAL = EST != EST_GLOBAL;  // the two local-majorant estimators share the segment logic
    // Counter-based streams, passive environment: ONE event visit per collision.  The shadow ray draws from a sub-stream of its own,
    // so the two phase-function variates -- the path's next draws -- are the same whether they are taken before or after it: the
    // new direction is sampled in the collision block (and, global majorant, the box is intersected for it), the shadow ray is
    // tracked, and where it ends the tracking step itself adds the light and goes on with the new segment.  The sequential
    // sampler.h stream (and the MIS build) keep the reference's order: collision, shadow ray, light, phase function.
    constexpr bool EARLY = RNG::kShadowSubstream && !MIS && !LIGHT;
    // Global majorant, counter-based streams: a new sample's camera ray may have been walked through its certified-empty stretch by
    // approach_k already (L.approach): the path is taken up where that walk stopped -- same draws, same sums, made elsewhere.
    constexpr bool APPR = EST == EST_GLOBAL && TRK == 0 && !LIGHT && !MIS;   // (any stream: the hand-over carries its state)
    // decomposition estimator: the same for the restart segments that end before the certified-empty distance (approach_local_k)
    constexpr bool APPR_L = EST == EST_DECOMP && TRK == 0 && !LIGHT && !MIS;
    // Exit flights.  A path in empty space that can do nothing but leave the box -- every cell its ray can still meet is certified
    // empty (L.exit_oct: the quarter pyramid of cells that opens from its cell along the direction's dominant axis) and a null collision
    // there leaves its throughput bit for bit as it is (null_collision_is_identity) -- ends with the environment along its direction
    // whatever it draws on the way (no draw is used after a path's end; the heat channel counts scatters): it goes to EV_BG at once
    // instead of walking there at 800 null collisions per unit length (38 % of the lane-steps of BASELINE config 2's general class).
    // A lane counts its null collisions in empty space in `terms` (free while no shadow ray is tracked); whenever its wave is in the
    // event pass anyway, lanes that have counted K are tested: one byte load, no loop, no parking.  Not for the bounded estimator (its
    // heat channel counts the segments of that walk), the scalar builds and MIS.
    constexpr bool EXITC = TRK == 0 && !MIS && !LIGHT && EST != EST_BOUNDED && (QUANT || EST == EST_GLOBAL);
    const ParamDev& P = L.P;
    const f3    sig_t     = f3{P.sigma_t[0], P.sigma_t[1], P.sigma_t[2]};
    const f3    sig_s     = sig_t * f3{P.albedo[0], P.albedo[1], P.albedo[2]};
    const float max_sig   = max3(sig_t);
    const float min_sig   = min3(sig_t);
    const float density   = P.density;

    // ---- per-lane path state
    int      st = ST_DONE;
    bool     exhausted = false;
    // cold state: in LDS for the plain kernels (ColdVal / ColdF3 above)
    constexpr bool COLD = LDSB != 1 && !LIGHT;
    constexpr int  CS_  = COLD ? VP_BLOCK : 1;
    // (the occupancy these kernels are budgeted for holds only while that many workgroups' cold state fits the CU's LDS --
    // a workgroup is one wave per SIMD, so waves per SIMD = workgroups per CU; gfx950: 160 KiB)
    static_assert(!COLD || (VP_GLOBAL_MIN_WAVES > VP_LOCAL_MIN_WAVES ? VP_GLOBAL_MIN_WAVES : VP_LOCAL_MIN_WAVES) * (14 * VP_BLOCK * 4 + KARG_WDS_ * 4 + (LDSB == 2 ? VP_LDS_BOUND_ENTRIES / 4 : 0)) <= VP_LDS_BYTES_PER_CU,
                  "cold per-path state: more workgroups per CU than the LDS holds -- lower VP_*_MIN_WAVES for this ARCH");
    __shared__ float cold_[COLD ? 14 : 1][CS_];
    float* const cold_p = &cold_[0][COLD ? threadIdx.x : 0];
    ColdF3<COLD, CS_>       rad(cold_p), pd(cold_p + 3 * CS_);   // radiance sum; primary direction, kept while the shadow ray is tracked
    ColdVal<float, COLD>    ph(cold_p + 6 * CS_), phase_g(cold_p + 7 * CS_);
    ColdVal<unsigned, COLD> item(cold_p + 8 * CS_);              // where this lane's sample goes in the staging buffer
    ColdVal<unsigned, COLD> px(cold_p + 9 * CS_), py(cold_p + 10 * CS_);
    ColdVal<int, COLD>      frame(cold_p + 11 * CS_);
    ColdVal<int, COLD>      nsc(cold_p + 12 * CS_);              // num_scatters (DECOMP, BOUNDED) / depth i (GLOBAL)
    ColdVal<unsigned, COLD> rng_saved(cold_p + 13 * CS_);        // the path's own stream position while a shadow ray draws from its sub-stream
    if (COLD) { rad = f3{0.0f, 0.0f, 0.0f}; pd = f3{0.0f, 0.0f, 0.0f}; ph = 0.0f; phase_g = 0.0f; item = 0u; px = 0u; py = 0u; frame = 0; nsc = 0; rng_saved = 0u; }
    RNG      rng;
    f3       ro = {}, rd = {};   // the ray being tracked (primary, or the shadow ray while ST_SHADOW)
    f3       inv_rd = {};        // 1 / rd of the primary ray (decomposition set-up)
    f3       thr = {};
    int      seg = 0;            // BOUNDED only: loop index i, one per tracked segment (kernel.cu:1716)
    float    dist = 0, t_end = 0;  // position on the tracked ray; where the current free flight ends
    float    t_far = 0, distc = 0, inv_sigma = 0, inv_sigma_t = 0, sigma_t_prime = 0, sigma_c = 0;
    float    cur_density = 0, d_max = 0;
    f3       nee_a = {};
    int      terms = 0;
    // MIS only: colour and weight of the light estimate in flight, rad += nee_c * (nee_t * transmittance);
    // 0 = sun, 1 = environment; the shadow majorant of this collision; the ray origin before the collision
    f3       nee_c = {}, nee_t = {}, seg_o = {};
    int      nee_stage = 0;
    float    sh_inv_sigma = 0, sh_density = 0;
    // scalar tracking only: the sample's extinction coefficient (density, or density * sigma_t[chan]) and its channel
    float    sig_base = density;
    int      chan = 0;
    // the unscattered camera ray is certified to run through empty cells (all eight texels of every fetch zero) up to this
    // distance from the current ray / segment origin (empty_table_k, crawl_table_k); 0 once the path has scattered
    float    t_empty = 0.0f;
    // COUNT build only: where the timed kernel ends the current sun shadow ray (it walks on here, so that density_lookups stays
    // the estimator's count, and stops counting loads)
    float    t_clip = 1e30f;
    bool     ex_clear = false;   // COUNT build only: the timed kernel has ended this path (exit flight); it walks on here, counting no loads
    unsigned long long c_load = 0, c_xtest = 0, c_xout = 0, c_xok = 0;
    unsigned zrun = 0;   // COUNT build only: null collisions in empty space of the current flight (c_xout: those of flights that left the box)

    unsigned long long c_den = 0, c_bnd = 0, c_opa = 0, c_env = 0, c_sca = 0, c_smp = 0;
    unsigned long long d_iter = 0, d_act = 0, d_outer = 0, d_shadow = 0;  // debug (lane 0 counts wave events)
    // PROF: the profiling form of the counting build (-DVP_PROFILE_BLOCKS=1: `make dev DEVNAME=prof DEVFLAGS=-DVP_PROFILE_BLOCKS=1`, scripts/
    // block_profile.py): cycle stamps of the two phases, loop statistics and the block tallies below -- 30 64-bit counters per lane that
    // the shipped counting build (the work counters of bench.py's lookups_per_sample) does not carry, nor their spills
    constexpr bool PROF = COUNT && VP_PROFILE_BLOCKS;
    unsigned long long t_slow = 0, t_fast = 0, t_mark = PROF ? __builtin_amdgcn_s_memtime() : 0ull;  // shader cycles
    // PROF build: how often each code block runs (wave executions) and for how many lanes -- where the lane slots go
    enum { B_SETUP, B_HALF, B_LOOK, B_EXIT, B_SCATTER, B_NEE, B_HG, B_BG, B_WRITE, B_REFILL, B_GSETUP, B_FETCH, B_ZERO, B_ZERO_SH, B_EXITT, B_NBLK };
    unsigned long long bw[B_NBLK] = {}, bl[B_NBLK] = {};
    // ... and, for the three blocks a regrouping of work would have to fill (collision, end of flight, restart set-up), the HISTOGRAM of
    // lanes per execution in eight buckets of eight lanes (round 5: profiles/experiments/r05_wavefront_break_even.md)
    unsigned long long hist[3][8] = {};
    unsigned long long ctrl_w = 0, ctrl_l = 0;
    auto tally = [&](int b, bool on) __attribute__((always_inline)) {
        if (PROF)
        {
            unsigned long long m = __ballot(on);
            if (m)
            {
                bw[b] += 1; bl[b] += (unsigned)__popcll(m);
                const int h = b == B_SCATTER ? 0 : b == B_EXIT ? 1 : b == B_SETUP ? 2 : -1;
                if (h >= 0) hist[h][((unsigned)__popcll(m) - 1u) >> 3] += 1;
            }
        }
    };

    const unsigned lane = threadIdx.x & 63u;
    unsigned chunk_s = 0, chunk_n = 0;    // next sample of the current chunk, samples in it (wave-uniform)
    unsigned chunk_q = 0, chunk_f0 = 0;   // its first pixel slot and first frame (wave-uniform)
    bool     queue_empty = false;
    // the queue this wave draws from: its XCD's first (HW_REG_XCC_ID, bits 3:0), then the others in turn
    unsigned q_cur   = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & (VP_NQUEUES - 1);
    unsigned q_tried = 0;
    unsigned end_skipped = 0;   // event visits since the path-end chain last ran (wave-uniform)
    unsigned visits = 0;        // CANCEL: event visits of this wave (wave-uniform)

    for (;;)
    {
        if (PROF && lane == 0) d_outer++;
        // =========================================================== slow path: events
        // The event section reads its uniforms (camera, sun, environment, queue and image descriptors: ~70 scalars that the
        // tracking loop never touches) from the kernel-argument segment afresh in every visit instead of holding them in SGPRs
        // across the tracking loop, whose own scalars then fit without spilling into vector lanes.  The empty asm keeps the
        // compiler from hoisting those loads back out of the loop; it also makes the pointer divergent for it, so the reads are
        // per-lane loads of the fields a visit's branches need, not scalar loads -- which, tried, load the whole structs and spill
        // 35-49 SGPRs: -20...-45 % (profiles/experiments/r03_scalar_kernarg_reload.txt).  Rounds 3-4: flat loads of the argument
        // segment; round 5: reads of its copy in LDS (kargs_lds_, above).
        {
#ifdef VP_EXP_FLAT_KARGS
        const char* kargs_ = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kargs_));
#else
        unsigned ko_ = 0;
        asm volatile("" : "+v"(ko_));   // (an opaque offset: the reads stay LDS reads and stay inside the visit)
        const char* kargs_ = reinterpret_cast<const char*>(kargs_lds_) + ko_;
#endif
        const SceneDev&  S = *reinterpret_cast<const SceneDev*>(kargs_);
        const LaunchDev& L = *reinterpret_cast<const LaunchDev*>(kargs_ + KARG_L_);
        const ParamDev&  P = L.P;
        const f3 sun_dir   = f3{S.sun_dir[0], S.sun_dir[1], S.sun_dir[2]};
        const f3 sun_power = f3{S.sun_power[0], S.sun_power[1], S.sun_power[2]};
        // Tr_spectral set-up kernel.cu:763-780: shadow ray from the collision point ro toward `end`
        // stage: 0 = the sun ray, 1 = the environment ray of the one-sample MIS
        auto start_shadow = [&](f3 end, float inv_s, float den, unsigned stage) __attribute__((always_inline)) {
            f3    sd = normalize(end - ro);
            float tn, tf;
            bool  hitv = intersect_box(ro, sd, S, tn, tf);
            if (!hitv)
            {
                nee_a = f3{1.0f, 1.0f, 1.0f};
                st    = EV_NEE;
            }
            else
            {
                if (tn < 0.0f) tn = 0.0f;
                f3 se       = ro - end;
                t_end       = fminf(tf, __builtin_sqrtf(dot(se, se)));
                dist        = tn;
                terms       = 0;
                rd          = sd;
                inv_sigma   = inv_s;
                cur_density = den;
                st          = ST_SHADOW;
                rng_saved = rng.enter_shadow(2u * (unsigned)(int)nsc + stage);
                if (COUNT) t_clip = 1e30f;
                if (RNG::kShadowSubstream && stage == 0u && L.sun_clip)
                {
                    // Beyond sun_clip[cell of ro] * clip_ds every fetch of this ray filters eight zero texels (sun_clip_k): no
                    // channel can terminate there any more, whatever is drawn, so the ray's result is known when it gets
                    // there.  Its draws come from a sub-stream of their own: nothing else depends on how many it makes.
                    f3    pl = to_local(S, ro);
                    int   ci, cj, ck;
                    float w_;
                    axis_linear(pl.x, S.nx, ci, w_);
                    axis_linear(pl.y, S.ny, cj, w_);
                    axis_linear(pl.z, S.nz, ck, w_);
                    const unsigned n = L.sun_clip[(size_t)((unsigned)ci + __umul24((unsigned)S.nx, (unsigned)cj + __umul24((unsigned)S.ny, (unsigned)ck)))];
                    if (n != 0xffffu)
                    {
                        const float tc = (float)n * L.clip_ds;
                        if (COUNT && !L.count_clips) t_clip = tc;
                        else t_end = fminf(t_end, tc);
                    }
                }
            }
        };
        // the path goes on with a new segment; loop bounds kernel.cu:34 with :2015 / :1332 / :1716
        // Local-majorant estimators: the Hyperion-reduced phase function and density of a segment (kernel.cu:2038-2046)
        // depend on the scatter count only, which restarts do not change: set where the count changes, not per restart.
        auto segment_medium = [&]() __attribute__((always_inline)) {
            if (LOCAL)
            {
                float s         = hyperion_s(nsc - 5);
                phase_g         = (1.0f - s) * P.g;
                float reduction = (1.0f - s) + s * (1.0f - P.g);
                cur_density     = TRK ? reduction * sig_base : reduction * density;  // scalar build: the coefficient itself (:2063)
            }
        };
        auto next_segment = [&]() __attribute__((always_inline)) {
            st = ST_SETUP;
            if (LOCAL) dist = -1.0f;
            if (EXITC) terms = L.exit_start;
            if (EST == EST_GLOBAL) nsc = nsc + 1;
            if (EST == EST_BOUNDED) seg++;
            if ((EST == EST_BOUNDED ? seg : nsc) >= 800) st = EV_WRITE;
            segment_medium();
        };
        // EARLY: the light estimate is known without a shadow ray (optical-depth table, or the ray misses the box)
        auto finish_light = [&]() __attribute__((always_inline)) {
            rad = rad + sun_power * (((ACH ? f3{thr.x, thr.x, thr.x} : thr) * ph) * nee_a);
            rd  = pd;
            next_segment();
        };
        // ---- collision: direct lighting set-up (kernel.cu:2161-2217 / :1458-1491)
        tally(B_SCATTER, st == EV_SCATTER);
        if (!LIGHT && st == EV_SCATTER)
        {
            vp_pad<VP_PAD_COLL>();
            if (COUNT) c_sca++;
            if (PROF) zrun = 0;
            t_empty = 0.0f;  // the certificate is for the unscattered camera ray only
            if (LOCAL) nsc = nsc + 1;  // num_scatters += !through, kernel.cu:2146
            // "to match passive result": post-increment count (DECOMP :2168) / i-4 (GLOBAL :1465)
            float s2 = hyperion_s((LOCAL) ? (nsc - 5) : (nsc - 4));
            float dp2, stp2;
            if (TRK)
            {
                thr  = thr * f3{P.albedo[0], P.albedo[1], P.albedo[2]};  // kernel.cu:2157-2159
                stp2 = LOCAL ? ((1.0f - s2) + s2 * (1.0f - P.g)) * sig_base : (1.0f - s2) * sig_base + s2 * sig_base * (1.0f - P.g);
                dp2  = stp2;  // Tr(..., inv_sigma, sigma_t_prime, rng): the coefficient is the "density" of the shadow ray
            }
            else if (LOCAL)
            {
                float reduction2 = (1.0f - s2) + s2 * (1.0f - P.g);
                dp2              = reduction2 * density;
                stp2             = max_sig * dp2 * d_max;  // quirk Q4: the local majorant for the whole shadow ray
            }
            else
            {
                dp2  = (1.0f - s2) * density + s2 * density * (1.0f - P.g);
                stp2 = max_sig * dp2;
            }
            ph = hg_eval(phase_g, dot(rd, sun_dir));
            pd = rd;
            if (EARLY)
            {
                // the direction the path takes up when the light estimate is in (kernel.cu:2301-2303)
                Frame fr(rd);
                float r0 = rng.next_a();
                float r1 = rng.next_b();
                pd       = normalize(fr.to_world(hg_sample_local(phase_g, r0, r1)));
                if (LOCAL) { const f3 pdv = pd; inv_rd = f3{1.0f / pdv.x, 1.0f / pdv.y, 1.0f / pdv.z}; }
            }
            if (MIS)
            {
                sh_inv_sigma = 1.0f / stp2;
                sh_density   = dp2;
                nee_stage    = 0;
                nee_c        = sun_power;
                nee_t        = (ACH ? f3{thr.x, thr.x, thr.x} : thr) * ph;
            }
            // here ro already holds the collision point (set by the tracking step)
            if (EST == EST_DECOMP && frame > 10 && nsc > 20)
            {
                // precomputed optical depth kernel.cu:2183-2189 (quirk Q5)
                // (the packed copy always exists: where the device could not hold it -- 8x the table -- it lies in pinned host memory)
                float op = sample_float_cells(S, S.opacity_cells, ro);   // = sample_float_volume(S, S.opacity, ro), from one line of the packed copy
                if (COUNT) c_opa++;
                if (TRK)
                {
                    float a = expf_(-stp2 * op);  // kernel.cu:2190
                    nee_a   = f3{a, a, a};
                }
                else if (ACH)
                {
                    float a = expf_(((-sig_t.x) * dp2) * op);  // the three channels are the same expression
                    nee_a   = f3{a, a, a};
                }
                else
                {
                    f3 tau = (f3{-sig_t.x, -sig_t.y, -sig_t.z} * dp2) * op;
                    nee_a  = f3{expf_(tau.x), expf_(tau.y), expf_(tau.z)};
                }
                st     = EV_NEE;
                if (EARLY) finish_light();
            }
            else
            {
                start_shadow(sun_dir * 1e10f, 1.0f / stp2, dp2, 0u);
                if (EARLY && st == EV_NEE) finish_light();   // the ray misses the box
                if (EARLY && EST == EST_GLOBAL && st == ST_SHADOW)
                {
                    // the set-up of the next segment (kernel.cu:1332-1345) for the new direction, while the lane is here anyway:
                    // where it enters (kept in t_empty, which is 0 for a scattered path and not read by shadow steps) and leaves
                    // the box (t_far; negative = it does not: the path ends with the environment)
                    float tn2, tf2;
                    bool  hit2 = intersect_box(ro, pd, S, tn2, tf2);
                    t_empty    = tn2 < 0.0f ? 0.0f : tn2;
                    t_far      = hit2 ? tf2 : -1.0f;
                }
            }
        }
#pragma unroll
        for (int pass = 0; pass < (MIS ? 2 : 1); pass++)
        {
            // ---- a light estimate is complete (kernel.cu:2188-2189,:2209-2210 and :2254,:2290)
            tally(B_NEE, st == EV_NEE);
            if (!LIGHT && !EARLY && st == EV_NEE)
            {
                if (MIS)
                {
                    rad = rad + nee_c * (nee_t * nee_a);
                    st  = nee_stage == 0 ? EV_MIS : EV_HG;
                }
                else
                {
                    rad = rad + sun_power * (((ACH ? f3{thr.x, thr.x, thr.x} : thr) * ph) * nee_a);
                    st  = EV_HG;
                }
            }
            // ---- one-sample MIS of the environment (kernel.cu:2220-2297; same block at :1494-1560 and :1855-1932)
            if (MIS && st == EV_MIS)
            {
                const float P_phase = 0.5f, P_envmap = 1.0f - P_phase;
                const f3    thr3    = ACH ? f3{thr.x, thr.x, thr.x} : thr;
                Frame       fr(pd);
                nee_stage = 1;
                if (rng.next_a() < P_phase)
                {
                    float u        = rng.next_a();
                    float v        = rng.next_b();
                    f3    brdf_dir = fr.to_world(hg_sample_local(phase_g, u, v));
                    f3    envc     = eval_envmap(S, brdf_dir);
                    if (COUNT) c_env++;
                    float pdf_brdf        = hg_eval(phase_g, dot(fr.n, brdf_dir));
                    float pdf_env_virtual = luminance(envc) * S.env_pdfnorm_alt;  // pdf_envmap :1009-1034
                    float wa = pdf_brdf * P_phase, wb = pdf_env_virtual * P_envmap;
                    float weight = wa / (wa + wb) / P_phase;
                    nee_c = envc;
                    nee_t = thr3 * weight;
                    start_shadow(brdf_dir * 1e10f, sh_inv_sigma, sh_density, 1u);
                }
                else
                {
                    float u = rng.next_a();
                    float v = rng.next_b();
                    f3    envc;
                    float pdf_env = sample_envmap(S, u, v, envc);
                    if (COUNT) c_env++;
                    if (pdf_env <= 0.0f)
                    {
                        // the reference `continue`s here (:2266): no scattered direction, the OLD ray goes on
                        ro = seg_o;
                        rd = pd;
                        next_segment();
                    }
                    else
                    {
                        f3    envmap_dir       = uv_to_dir(u, v);
                        float pdf_brdf_virtual = hg_eval(phase_g, dot(fr.n, envmap_dir));
                        float wa = pdf_env * P_envmap, wb = pdf_brdf_virtual * P_phase;
                        float weight = wa / (wa + wb) / P_envmap;
                        nee_c = envc;
                        nee_t = ((thr3 * pdf_brdf_virtual) / pdf_env) * weight;
                        start_shadow(envmap_dir * 1e10f, sh_inv_sigma, sh_density, 1u);
                    }
                }
            }
        }
        // ---- phase-function sampling (kernel.cu:2301-2303)
        tally(B_HG, st == EV_HG);
        if (!LIGHT && !EARLY && st == EV_HG)
        {
            Frame fr(pd);
            float r0 = rng.next_a();
            float r1 = rng.next_b();
            rd       = normalize(fr.to_world(hg_sample_local(phase_g, r0, r1)));
            if (LOCAL) inv_rd = f3{1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
            next_segment();
        }
        // ---- exit flights: can this path do anything but leave the box?  Lanes in flight whose counter has tripped ride along.
        if (EXITC)
        {
            const bool cand = (st == ST_TRACK || (LOCAL && st == ST_SETUP)) && terms >= VP_EXIT_TRIP;
            tally(B_EXITT, cand);
            if (cand)
            {
                if (COUNT) c_xtest++;
                // (1) the cells: every fetch the ray can still make lies in the quarter pyramid of cells that opens from its cell
                // along the dominant axis of its direction, toward the sides the other two components point to; the table says
                // whether every cell of that pyramid is empty with empty neighbours (exit_dir_slice_k)
                const f3 pl = to_local(S, st == ST_TRACK ? ro + rd * dist : ro);
                int   ci, cj, ck;
                float w_;
                axis_linear(pl.x, S.nx, ci, w_);
                axis_linear(pl.y, S.ny, cj, w_);
                axis_linear(pl.z, S.nz, ck, w_);
                // the direction in cell units, its dominant axis A and the signs (A, then the other two in increasing order)
                const float ex = rd.x * (S.linv[0] * (float)S.nx), ey = rd.y * (S.linv[1] * (float)S.ny), ez = rd.z * (S.linv[2] * (float)S.nz);
                const float ax = __builtin_fabsf(ex), ay = __builtin_fabsf(ey), az = __builtin_fabsf(ez);
                const unsigned A   = (ax >= ay && ax >= az) ? 0u : (ay >= az ? 1u : 2u);
                const unsigned cls = A == 0u ? ((ex > 0.0f ? 1u : 0u) | (ey > 0.0f ? 2u : 0u) | (ez > 0.0f ? 4u : 0u))
                                   : A == 1u ? ((ey > 0.0f ? 1u : 0u) | (ex > 0.0f ? 2u : 0u) | (ez > 0.0f ? 4u : 0u))
                                             : ((ez > 0.0f ? 1u : 0u) | (ex > 0.0f ? 2u : 0u) | (ey > 0.0f ? 4u : 0u));
                const size_t   ncell = (size_t)S.nx * (size_t)S.ny * (size_t)S.nz;
                const unsigned bits  = L.exit_oct[(size_t)A * ncell + (size_t)((unsigned)ci + __umul24((unsigned)S.nx, (unsigned)cj + __umul24((unsigned)S.ny, (unsigned)ck)))];
                bool clear = (bits >> cls) & 1u;
                // (2) the throughput: every majorant a null collision can meet on the way must leave it as it is: the segment's own
                // (global majorant); that of every byte that occurs as a maximum in the bound table (local majorants: a segment
                // through empty cells may still lie in a brick with a positive maximum)
                bool unit = true;
                if (clear)
                {
                    const f3 t3 = ACH ? f3{thr.x, thr.x, thr.x} : thr;
                    if (LOCAL)
                    {
                        for (unsigned q = 0; q < L.exit_nbytes; q++)
                        {
                            const float dm  = fmaxf(0.0001f, (float)((L.exit_bytes >> (8u * q)) & 0xffu) * VP_U8_SCALE);   // segment_setup()
                            const float stp = max_sig * cur_density * dm;
                            unit = unit && null_collision_is_identity(t3, stp, 1.0f / stp);
                        }
                    }
                    else
                        unit = null_collision_is_identity(t3, sigma_t_prime, inv_sigma_t);
                }
                if (COUNT && clear && unit) c_xok++;
                if (clear && unit && !(COUNT && !L.count_clips)) st = EV_BG;
                else
                {
                    // not (yet): the next test comes after another K null collisions in empty space -- much later where the throughput
                    // was the obstacle (it moves by an ulp per null collision until it meets a fixed point of the factor).  The counting
                    // build walks on (its density_lookups are the estimator's) and stops counting loads.
                    if (COUNT && clear && unit) ex_clear = true;
                    terms = (clear && unit) ? -(1 << 30) : (unit ? L.exit_start : L.exit_start - 4 * VP_EXIT_TRIP);
                }
            }
        }
        // Path ends (environment, write, refill, and the global-majorant set-up of a fresh sample) come one or two lanes at a time:
        // the ~300 instructions of this chain are not run in every visit for them.  They wait -- an idle lane or two -- until
        // end_lanes lanes ask, or four visits have passed, or nothing else is left to do in this wave.
        {
            const unsigned long long wantm = __ballot(st == EV_BG || st == EV_WRITE || (st == ST_DONE && !exhausted) || (EST == EST_GLOBAL && st == ST_SETUP));
            const bool fast_any = __ballot(st == ST_TRACK || st == ST_SHADOW || (LOCAL && st == ST_SETUP)) != 0ull;
            end_skipped++;
            if (wantm == 0ull || (!LIGHT && fast_any && (unsigned)__popcll(wantm) < L.end_lanes && end_skipped < 4u)) goto ends_done;
            end_skipped = 0u;
        }
#pragma unroll 1
        for (int rep = 0; rep < 4; rep++)
        {
            vp_pad<VP_PAD_END>();
            bool fresh = false;   // APPR: this lane took a new sample in this round, `dist` holds where approach_k left its camera ray
            // order: a path that ends here is written, its lane refilled and the new segment set up in ONE round
            // ---- ray left the medium: background() kernel.cu:1258-1267 (quirk Q11)
            tally(B_BG, st == EV_BG);
            if (st == EV_BG)
            {
                // with active environment sampling only unscattered paths see it directly (kernel.cu:2026-2030, :1340-1344)
                if (!MIS || nsc == 0)
                {
                    f3 bg;
                    if (nsc == 0 && dot(rd, sun_dir) > S.sun_cos) bg = f3{S.sun_orig[0], S.sun_orig[1], S.sun_orig[2]};
                    else { bg = eval_envmap(S, rd); if (COUNT) c_env++; }
                    if (LIGHT && !LOCAL)
                    {
                        // throughput after `seg` null collisions in empty space (see tracking_step)
                        const unsigned n = (unsigned)seg, last = L.thr_n - 1u;
                        float t = L.thr_table[n < last ? n : last];
                        for (unsigned k = last; k < n; k++) t = null_collision_in_empty_space(t, sigma_t_prime, inv_sigma_t);
                        thr = f3{t, t, t};
                    }
                    rad = rad + bg * (ACH ? f3{thr.x, thr.x, thr.x} : thr);
                }
                st = EV_WRITE;
            }
            // ---- path end: emit the sample (kernel.cu:2306-2316 / :1579-1589)
            tally(B_WRITE, st == EV_WRITE);
            if (st == EV_WRITE)
            {
                f3     r    = rad * P.brightness;
                // heat: num_scatters (:2307) or loop index * 0.001 in double (:1581, :1942)
                float  heat = (EST == EST_DECOMP) ? (float)nsc : (float)((double)(EST == EST_BOUNDED ? seg : nsc) * 0.001);
                float4 v    = make_float4(fmaxf(r.x, 0.0f), fmaxf(r.y, 0.0f), fmaxf(r.z, 0.0f), heat);
                if (TRK == 2)  // kernel.cu:2311-2313: the drawn channel only, times three
                    v = make_float4(chan == 0 ? v.x * 3.0f : 0.0f, chan == 1 ? v.y * 3.0f : 0.0f, chan == 2 ? v.z * 3.0f : 0.0f, heat);
                if (L.stage) L.stage[item] = v;
                else
                {
                    size_t idx = (size_t)px + (size_t)py * P.width;
                    float4 a   = L.out[idx];
                    L.out[idx] = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
                }
                st = ST_DONE;
            }
            // ---- refill finished lanes from the queue.  The wave owns a chunk of samples (chunk_s .. chunk_n)
            // of consecutive samples (one atomic per VP_CHUNK samples); idle lanes are compacted
            // with ballot + mbcnt and take the next samples of the chunk.
            {
                bool               need = (st == ST_DONE) && !exhausted;
                unsigned long long m    = __ballot(need);
                tally(B_REFILL, need);
                if (m)
                {
                    while (chunk_s >= chunk_n && !queue_empty)
                    {
                        // queue q_cur: chunk c is frame (c % nframes) of chunk position (c / nframes) of its band -- the waves
                        // running at the same time work on the same few tiles in different frames, i.e. on rays through the
                        // same pencil of the volume
                        // A chunk is (VP_CHUNK >> chunk_fshift) consecutive pixels of the band x (1 << chunk_fshift) consecutive frames
                        // (the host sets chunk_fshift = 0 unless the frame count is a multiple): sample s of it is pixel s >> shift,
                        // frame s & mask -- with shift 6 a wave starts on ONE pixel in 64 frames: the same camera ray in every lane.
                        const unsigned q0 = L.q_start[q_cur], len = L.q_start[q_cur + 1] - q0;
                        static_assert((VP_CHUNK & (VP_CHUNK - 1)) == 0 && VP_CHUNK >= 64, "VP_CHUNK: a power of two, at least a wave");
                        const unsigned sh = L.chunk_fshift, ppc = (unsigned)VP_CHUNK >> sh, fblocks = (unsigned)L.nframes >> sh;   // (the host keeps sh <= log2(VP_CHUNK): ppc >= 1)
                        const unsigned cpf = (len + ppc - 1u) / ppc;
                        unsigned c = 0xffffffffu;
                        if (len)
                        {
                            // (a cancelled look-ahead batch hands out nothing more: an atomic load at agent scope, so that the host's write from
                            // another stream is seen whichever XCD this wave runs on)
                            if (lane == 0 && !(L.cancel && __hip_atomic_load(L.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= L.batch_id)) c = atomicAdd(L.queue + q_cur * VP_QUEUE_STRIDE, 1u);
                            c = __builtin_amdgcn_readfirstlane(c);
                        }
                        if (c < cpf * fblocks)
                        {
                            const unsigned pos = c / fblocks, fb = c - pos * fblocks, off = pos * ppc;
                            chunk_q  = q0 + off;
                            chunk_f0 = fb << sh;
                            chunk_n  = (len - off < ppc ? len - off : ppc) << sh;
                            chunk_s  = 0;
                        }
                        else
                        {
                            // this band is handed out: help with the next one
                            q_cur = (q_cur + 1) & (VP_NQUEUES - 1);
                            if (++q_tried >= VP_NQUEUES) queue_empty = true;
                        }
                    }
                    unsigned cnt   = (unsigned)__popcll(m);
                    unsigned avail = chunk_n - chunk_s;
                    unsigned take  = cnt < avail ? cnt : avail;
                    unsigned rank  = lane_rank(m);
                    if (need)
                    {
                        if (rank >= take) { if (queue_empty) exhausted = true; /* else: next chunk, next round */ }
                        else
                        {
                            const unsigned sn  = chunk_s + rank, sh = L.chunk_fshift;
                            const unsigned rem = chunk_q + (sn >> sh);              // sample slot of this class within the frame
                            const unsigned fl  = chunk_f0 + (sn & ((1u << sh) - 1u));
                            item = fl * L.stage_stride + L.slot_base + rem;
                            unsigned pix = L.pixels[rem];
                            px    = pix & 0xffffu;
                            py    = pix >> 16;
                            frame = L.frame0 + (int)fl;
                            if (px < P.width && py < P.height)
                            {
                                // camera ray, kernel.cu:1977-1987 (quirk Q3)
                                rng.init(px, py, (unsigned)frame, L.key0, L.key1);
                                if (TRK == 2)
                                {
                                    chan     = (int)fminf((1.0f - rng.next_a()) * 3.0f, 2.9999998f);  // kernel.cu:1993
                                    sig_base = density * (chan == 0 ? P.sigma_t[0] : chan == 1 ? P.sigma_t[1] : P.sigma_t[2]);
                                }
                                camera_ray(S, P.width, P.height, px, py, ro, rd);
                                if (LOCAL) inv_rd = f3{1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
                                thr = f3{1.0f, 1.0f, 1.0f};
                                rad = f3{0.0f, 0.0f, 0.0f};
                                nsc = 0;
                                seg = 0;
                                if (EXITC) terms = L.exit_start;
                                if (COUNT) ex_clear = false;
                                if (!LIGHT) t_empty = L.crawl ? L.crawl[2 * ((size_t)px + (size_t)py * P.width) + 1].x : 0.0f;
                                if (LOCAL) dist = -1.0f;   // a segment starts where the ray enters it (segment_setup), unless approach_local_k got further
                                if (APPR && L.approach)
                                {
                                    const float4 a = L.stage[item];   // approach_k: distance reached, where the stream stands, steps made
                                    dist  = a.x;
                                    rng.load(f2u(a.y), f2u(a.z));
                                    fresh = true;
                                    if (L.approach == 2u)
                                    {
                                        // a medium whose null collision in empty space is not exactly neutral: the throughput after the
                                        // walk's n of them is the n-th iterate of one function of one float (thr_table_k, as the light
                                        // kernel of this estimator looks it up; beyond the table the recurrence is run)
                                        const unsigned n = f2u(a.w), last = L.thr_n - 1u;
                                        float t = L.thr_table[n < last ? n : last];
                                        if (n > last)
                                        {
                                            const float s0  = hyperion_s(0 - 5);
                                            const float stp = max_sig * ((1.0f - s0) * density + s0 * density * (1.0f - P.g));
                                            const float inv = 1.0f / stp;
                                            for (unsigned k = last; k < n; k++) t = null_collision_in_empty_space(t, stp, inv);
                                        }
                                        thr = f3{t, t, t};
                                    }
                                }
                                if (LOCAL && L.crawl)
                                {
                                    // the restart crawl in front of the volume, done once per pixel by crawl_table_k: the path starts
                                    // where that crawl ends, with its draws skipped and its segments counted
                                    float4   c = L.crawl[2 * ((size_t)px + (size_t)py * P.width)];
                                    unsigned k = f2u(c.w);
                                    ro = f3{c.x, c.y, c.z};
                                    if (!(APPR_L && L.approach)) rng.skip(k >> 16);   // (else the hand-over below carries the stream's state)
                                    if (EST == EST_BOUNDED) seg = (int)(k & 0xffffu);
                                    if (COUNT) c_bnd += k & 0xffffu;
                                    if (APPR_L && L.approach)
                                    {
                                        // approach_local_k walked on from there: origin of the first segment it did not finish, pairs used
                                        // so far.  The certificate is measured from the segment origin: less the distance walked (the
                                        // projection on the ray, a margin of 1e-4 against its rounding: a shorter certificate renders the
                                        // same bits, it only fetches a zero it could have skipped)
                                        const float4 a  = L.stage[item];
                                        const uint2  ax = L.approach_aux[item];
                                        const f3     ra = f3{a.x, a.y, a.z};
                                        t_empty = t_empty - dot(ra - ro, rd) - 1e-4f;
                                        ro      = ra;
                                        // how far into the segment at ra the walk got (-1: not at all): the set-up below keeps it
                                        dist    = a.w;
                                        // where the stream stands: the pair index, or sampler.h's two words
                                        rng.load(ax.x, ax.y);
                                    }
                                }
                                st  = ST_SETUP;
                                segment_medium();
                                if (COUNT) c_smp++;
                            }
                            // pixels of a partial edge tile outside the image: nothing to do, stay DONE
                        }
                    }
                    chunk_s += take;
                }
            }
            // ---- global-majorant segment set-up (__d_render kernel.cu:1332-1370); rare, so it lives here
            if (EST == EST_GLOBAL) tally(B_GSETUP, st == ST_SETUP);
            if (EST == EST_GLOBAL && st == ST_SETUP)
            {
                float t_near, tf;
                bool  hit = intersect_box(ro, rd, S, t_near, tf);
                if (!hit) st = EV_BG;
                else
                {
                    if (t_near < 0.0f) t_near = 0.0f;
                    t_far         = tf;
                    t_end         = tf;
                    dist          = (APPR && fresh) ? dist : t_near;
                    float s       = hyperion_s(nsc - 5);
                    phase_g       = (1.0f - s) * P.g;
                    if (TRK)
                    {
                        sigma_t_prime = (1.0f - s) * sig_base + s * sig_base * (1.0f - P.g);  // kernel.cu:1363
                        cur_density   = sigma_t_prime;  // vol_sigma_t(pos, sigma_t_prime), kernel.cu:1436
                    }
                    else
                    {
                        cur_density   = (1.0f - s) * density + s * density * (1.0f - P.g);
                        sigma_t_prime = max_sig * cur_density;
                    }
                    inv_sigma     = 1.0f / sigma_t_prime;
                    inv_sigma_t   = inv_sigma;
                    st            = ST_TRACK;
                }
            }
            // another round only for lanes this loop can serve: idle ones, and new samples that missed the volume
            if (__ballot((st == ST_DONE && !exhausted) || st == EV_BG || st == EV_WRITE) == 0ull) break;
        }
ends_done:
        // CANCEL: the instance look-ahead batches run (render_kernel's staged frames, LaunchDev::cancel).  A batch the host has dropped --
        // a camera move, a setter -- is of no use to anybody: every wave asks at every eighth event visit (a visit comes every ~15 us)
        // and gives up its paths at once instead of tracing them to their ends, which is what the move would otherwise wait for (the
        // deepest paths in flight: ~10 ms).  Nothing reads what such a batch has staged.  An instance of its own because the test,
        // small as it is, reshuffles the registers of the batched global-majorant kernel to the tune of -4...-5.5 %
        // (profiles/experiments/r04_lookahead_cancel.txt).
        if (CANCEL && L.cancel && (visits++ & 7u) == 0u)
        {
            unsigned w = 0;
            if (lane == 0) w = __hip_atomic_load(L.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a load at agent scope, not a read-modify-write: ADVICE r4)
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)w) >= L.batch_id) { st = ST_DONE; exhausted = true; queue_empty = true; }
        }
        if (__ballot(st != ST_DONE || !exhausted) == 0ull) break;  // queue drained and every lane idle
        }
        if (PROF) { unsigned long long t = __builtin_amdgcn_s_memtime(); if (lane == 0) t_slow += t - t_mark; t_mark = t; }

        // =========================================================== fast path: tracking
        // one segment set-up (local-majorant estimators) and one tracking step, as lambdas: the loop below runs
        // them twice per pass so that the wave-level bookkeeping (ballots, wait policy) is paid once per two steps
        auto segment_setup = [&]() __attribute__((always_inline)) {
            if (LOCAL) tally(B_SETUP, st == ST_SETUP);
            if (LOCAL && st == ST_SETUP)
            {
                vp_pad<VP_PAD_SETUP>();
                // intersectSuperVolume kernel.cu:1626-1661 (quirks Q6, Q10): the bound is fetched before the hit test
                float t_near, tf;
                bool  hit = intersect_box_inv(ro, inv_rd, S, t_near, tf);
                t_near    = fmaxf(t_near, 0.0f);
                t_far     = fminf(tf, 0.05f);
                float bx, by;
                if (LDSB)
                {
                    f3  pl = to_local(S, ro + rd * t_near);
                    int bi = axis_point(pl.x, S.nx) >> S.brick_shift;
                    int bj = axis_point(pl.y, S.ny) >> S.brick_shift;
                    int bk = axis_point(pl.z, S.nz) >> S.brick_shift;
                    const unsigned bidx = (unsigned)bi + __umul24((unsigned)S.bnx, (unsigned)bj + __umul24((unsigned)S.bny, (unsigned)bk));
                    unsigned v;
                    if (LDSB == 2)
                    {
                        // sixteen 2-bit codes per word; the palette's four byte pairs come with the launch
                        const unsigned code = (lds_codes[bidx >> 4] >> ((bidx & 15u) << 1)) & 3u;
                        const unsigned pal  = (code & 2u) ? L.bound_pal[1] : L.bound_pal[0];
                        v = (code & 1u) ? pal >> 16 : pal & 0xffffu;
                    }
                    else
                        v = lds_bounds[bidx];
                    bx = (float)(v & 0xffu) * VP_U8_SCALE;
                    by = (float)(v >> 8) * VP_U8_SCALE;
                }
                else
                    sample_bound<QUANT>(S, ro + rd * t_near, bx, by);
                if (COUNT) c_bnd++;
                float d_min = by;
                d_max       = fmaxf(0.0001f, bx);
                if (!hit) st = EV_BG;
                else
                {
                    // (where the free flight of this segment starts: where the ray enters it -- or, in the one segment approach_local_k
                    // handed over half-walked, where that walk got; every other set-up finds dist = -1)
                    dist          = fmaxf(dist, t_near);
                    // phase_g and cur_density of this scatter count: segment_medium().  Scalar build: no local bound (:2063 / :1745)
                    sigma_t_prime = TRK ? cur_density : max_sig * cur_density * d_max;
                    inv_sigma_t   = 1.0f / sigma_t_prime;
                    if (PROF)
                    {
                        // (how often the control component's 65 instructions -- a draw, a logarithm, two divisions -- run, and for how many lanes)
                        const unsigned long long cm = __ballot(TRK == 0 && EST == EST_DECOMP && d_min > 0.0f);
                        if (cm) { ctrl_w += 1; ctrl_l += (unsigned)__popcll(cm); }
                    }
                    if (TRK == 0 && EST == EST_DECOMP && d_min > 0.0f)
                    {
                        // analog decomposition tracking kernel.cu:2048-2054 (quirk Q7)
                        sigma_c       = min_sig * cur_density * d_min;
                        distc         = dist - logf_(rng.next_a()) / fmaxf(sigma_c, 1e-20f);
                        float sigma_r = fmaxf(sigma_t_prime - sigma_c, 1e-20f);
                        inv_sigma     = 1.0f / sigma_r;
                    }
                    else
                    {
                        distc     = 1e20f;
                        sigma_c   = 0.0f;
                        inv_sigma = inv_sigma_t;
                    }
                    t_end = fminf(distc, t_far);  // dist >= distc || dist >= t_far  (kernel.cu:2086)
                    st    = ST_TRACK;
                }
            }
        };
        // EARLY: a shadow ray has ended (nee_a known): add the light and go on with the segment prepared in the collision block
        auto light_done = [&]() __attribute__((always_inline)) {
#ifdef VP_EXP_FLAT_KARGS
            const char* ka_ = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka_));   // the sun's power is read here, not held in scalar registers across the loop
#else
            unsigned ko2_ = 0;
            asm volatile("" : "+v"(ko2_));
            const char* ka_ = reinterpret_cast<const char*>(kargs_lds_) + ko2_;
#endif
            const SceneDev& S2 = *reinterpret_cast<const SceneDev*>(ka_);
            const f3 sunp = f3{S2.sun_power[0], S2.sun_power[1], S2.sun_power[2]};
            rad = rad + sunp * (((ACH ? f3{thr.x, thr.x, thr.x} : thr) * ph) * nee_a);
            rd  = pd;
            if (EXITC) terms = reinterpret_cast<const LaunchDev*>(ka_ + KARG_L_)->exit_start;
            if (EST == EST_GLOBAL)
            {
                nsc = nsc + 1;
                if (nsc >= 800) st = EV_WRITE;
                else if (t_far < 0.0f) { st = EV_BG; t_empty = 0.0f; }
                else
                {
                    // __d_render's segment set-up kernel.cu:1355-1370 for the depth index just reached
                    t_end         = t_far;
                    dist          = t_empty;
                    t_empty       = 0.0f;
                    float s       = hyperion_s(nsc - 5);
                    phase_g       = (1.0f - s) * P.g;
                    if (TRK)
                    {
                        sigma_t_prime = (1.0f - s) * sig_base + s * sig_base * (1.0f - P.g);
                        cur_density   = sigma_t_prime;
                    }
                    else
                    {
                        cur_density   = (1.0f - s) * density + s * density * (1.0f - P.g);
                        sigma_t_prime = max_sig * cur_density;
                    }
                    inv_sigma     = 1.0f / sigma_t_prime;
                    inv_sigma_t   = inv_sigma;
                    st            = ST_TRACK;
                }
            }
            else
            {
                st   = ST_SETUP;
                dist = -1.0f;
                if (EST == EST_BOUNDED) seg++;
                if ((EST == EST_BOUNDED ? seg : nsc) >= 800) st = EV_WRITE;
                float s         = hyperion_s(nsc - 5);
                phase_g         = (1.0f - s) * P.g;
                float reduction = (1.0f - s) + s * (1.0f - P.g);
                cur_density     = TRK ? reduction * sig_base : reduction * density;
            }
        };
        auto tracking_step = [&]() __attribute__((always_inline)) {
            tally(B_HALF, st == ST_TRACK || st == ST_SHADOW);
            if (LIGHT && !LOCAL)
            {
                // Every fetch of this path would filter eight zero texels (certified: the whole chord): the general expressions
                // below with den = +0 reduce exactly to sigma_t_den = +0, Ps = +0, c = Pn, `real` false for any draw and
                // sigma_null_den = sigma_t_prime -- no position, fetch or filter.  (A throughput that is not
                // finite stays NaN either way and the sample is written as 0.)
                if (st == ST_TRACK)
                {
                    dist += -logf_(rng.next_a()) * inv_sigma;  // kernel.cu:1419
                    if (dist >= t_end) st = EV_BG;              // transmitted through the box kernel.cu:1444-1452
                    else
                    {
                        // the collision test's variate is not needed (`real` is false whatever it is), but a sequential stream
                        // (sampler.h) must still move past it; for the counter-based streams this is nothing
                        (void)rng.next_b();
                        if (COUNT) c_den++;
                        // The throughput update of this null collision, thr *= sigma_t' * ((inv_sigma_t * Pn) / Pn) with
                        // Pn = |sigma_t' thr.x| + |sigma_t' thr.y| + |sigma_t' thr.z|, draws nothing and starts from (1,1,1) with the
                        // same sigma_t' in every sample: after n null collisions the throughput is the n-th iterate of one
                        // function of one float (three equal channels), tabulated by thr_table_k.  Count here, look up at the exit.
                        seg++;
                    }
                }
                return;
            }
            if (st == ST_TRACK || st == ST_SHADOW)
            {
                const bool shadow = st == ST_SHADOW;
                vp_pad<VP_PAD_STEP>();
                dist += -logf_(rng.next_a()) * inv_sigma;  // kernel.cu:2085 / :784
                tally(B_EXIT, dist >= t_end || (shadow && terms == 7));
                tally(B_LOOK, !(dist >= t_end || (shadow && terms == 7)));
                if (dist >= t_end || (shadow && terms == 7))
                {
                    vp_pad<VP_PAD_EOF>();
                    if (shadow)
                    {
                        // Tr_spectral returns 1 - terminated flags (kernel.cu:807)
                        nee_a = f3{(float)(1 - (terms & 1)), (float)(1 - ((terms >> 1) & 1)), (float)(1 - ((terms >> 2) & 1))};
                        st    = EV_NEE;
                        rng.leave_shadow(rng_saved);
                        if (EARLY) light_done();
                    }
                    else if (LOCAL)
                    {
                        bool through = fminf(distc, dist) >= t_far;  // kernel.cu:2145
                        if (through)
                        {
                            ro = ro + rd * t_far;  // tracking restart kernel.cu:2151-2155 / :1809-1813
                            t_empty -= t_far;      // the certified-empty distance is measured from the segment origin
                            st   = ST_SETUP;
                            dist = -1.0f;
                            if (EXITC) terms += d_max <= 0.0001f ? VP_EXIT_TRIP : 0;   // exit flights: a segment through a brick with maximum zero
                            if (EST == EST_BOUNDED && ++seg >= 800) st = EV_WRITE;  // `continue` still counts, :1716
                        }
                        else
                        {
                            if (MIS) seg_o = ro;
                            ro = ro + rd * distc;  // control collision kernel.cu:2088
                            st = LIGHT ? EV_WRITE : EV_SCATTER;  // (LIGHT: excluded by the pixel class, which checks the brick minima)
                        }
                    }
                    else
                    {
                        st = EV_BG;  // transmitted through the box kernel.cu:1444-1452
                        if (PROF) { c_xout += zrun; zrun = 0; }
                    }
                }
                else
                {
                    vp_pad<VP_PAD_FETCH>();
                    f3    p   = ro + rd * dist;
                    float den;
                    tally(B_FETCH, !LIGHT && (shadow || !(dist < t_empty)));
                    if (LIGHT) den = 0.0f;  // light class of a local-majorant estimator: every fetch is certified to return +0
                    else if (EST == EST_GLOBAL)
                    {
                        // Before t_empty every texel this fetch would filter is zero (empty_table_k): the product is +0 without
                        // position, address, load or filter.  Whole waves of background rays take this branch together.
                        den = 0.0f;
                        if (shadow || !(dist < t_empty))
                        {
                            den = sample_density01<QUANT>(S, p) * cur_density;  // vol_sigma_t kernel.cu:682-695
                            if (COUNT && !(shadow && dist >= t_clip) && !(!shadow && ex_clear)) c_load++;
                        }
                    }
                    else
                    {
                        // local-majorant estimators: the same certificate, measured from the current segment origin
                        den = 0.0f;
                        if (shadow || !(dist < t_empty))
                        {
                            den = sample_density01<QUANT>(S, p) * cur_density;  // vol_sigma_t kernel.cu:682-695
                            if (COUNT && !(shadow && dist >= t_clip) && !(!shadow && ex_clear)) c_load++;
                        }
                    }
                    float e   = rng.next_b();
                    if (COUNT) c_den++;
                    tally(B_ZERO, !LIGHT && !shadow && !(dist < t_empty) && den == 0.0f);
                    tally(B_ZERO_SH, !LIGHT && shadow && den == 0.0f && !(COUNT && dist >= t_clip));
                    if (TRK)
                    {
                        // scalar delta tracking: kernel.cu:2137-2142 / :745-748 (Tr stops AT its collision, no further draw)
                        if (e < den * inv_sigma)
                        {
                            if (shadow) { nee_a = f3{0.0f, 0.0f, 0.0f}; st = EV_NEE; rng.leave_shadow(rng_saved); if (EARLY) light_done(); }
                            else { ro = p; st = EV_SCATTER; }
                        }
                    }
                    else if (shadow)
                    {
                        // kernel.cu:791-805
                        if (ACH) terms = (e < sig_t.x * den * inv_sigma) ? 7 : terms;
                        else
                        {
                            int t = terms;
                            t |= (e < sig_t.x * den * inv_sigma) ? 1 : 0;
                            t |= (e < sig_t.y * den * inv_sigma) ? 2 : 0;
                            t |= (e < sig_t.z * den * inv_sigma) ? 4 : 0;
                            terms = t;
                        }
                    }
                    else if (ACH)
                    {
                        // history-aware collision probabilities kernel.cu:2107-2134 (quirk Q8), one channel carried
                        float a_t = sig_t.x * den, a_s = sig_s.x * den;
                        if (LOCAL) { a_t = a_t - sigma_c; a_s = a_s - sigma_c; }
                        float a_n  = sigma_t_prime - a_t;
                        float mt   = __builtin_fabsf(a_t * thr.x), mn = __builtin_fabsf(a_n * thr.x);
                        float Ps   = (mt + mt) + mt;
                        float Pn   = (mn + mn) + mn;
                        float c    = Ps + Pn;
                        bool  real = e * c < Ps;
                        float f    = wdiv_(inv_sigma_t * c, real ? Ps : Pn);
                        thr.x      = thr.x * ((real ? a_s : a_n) * f);
                        if (real)
                        {
                            if (MIS) seg_o = ro;
                            ro = p;
                            st = LIGHT ? EV_WRITE : EV_SCATTER;  // (LIGHT: den = +0 makes `real` false)
                        }
                        else if (EXITC && den == 0.0f)
                        {
                            terms++;   // exit flights: a null collision in empty space
                            if (PROF) zrun++;
                        }
                    }
                    else
                    {
                        // history-aware collision probabilities kernel.cu:2107-2134 (quirk Q8)
                        f3 sigma_t_den = sig_t * den;
                        f3 sigma_s_den = sig_s * den;
                        if (LOCAL)
                        {
                            f3 sc       = f3{sigma_c, sigma_c, sigma_c};
                            sigma_t_den = sigma_t_den - sc;
                            sigma_s_den = sigma_s_den - sc;
                        }
                        f3    sigma_null_den = f3{sigma_t_prime, sigma_t_prime, sigma_t_prime} - sigma_t_den;
                        float Ps = __builtin_fabsf(sigma_t_den.x * thr.x) + __builtin_fabsf(sigma_t_den.y * thr.y) +
                                   __builtin_fabsf(sigma_t_den.z * thr.z);
                        float Pn = __builtin_fabsf(sigma_null_den.x * thr.x) + __builtin_fabsf(sigma_null_den.y * thr.y) +
                                   __builtin_fabsf(sigma_null_den.z * thr.z);
                        float c    = Ps + Pn;
                        bool  real = e * c < Ps;
                        float f    = wdiv_(inv_sigma_t * c, real ? Ps : Pn);
                        f3    sel  = real ? sigma_s_den : sigma_null_den;
                        thr        = thr * (sel * f);
                        if (real)
                        {
                            if (MIS) seg_o = ro;
                            ro = p;
                            st = LIGHT ? EV_WRITE : EV_SCATTER;  // (LIGHT: den = +0 makes `real` false)
                        }
                        else if (EXITC && den == 0.0f)
                        {
                            terms++;   // exit flights: a null collision in empty space
                            if (PROF) zrun++;
                        }
                    }
                }
            }
        };
        // the light kernel of the global-majorant estimator has the shortest step (a draw, a logarithm, an add and a compare):
        // more of them per round of wave-level bookkeeping
        constexpr int STEPS = (LIGHT && !LOCAL) ? VP_LIGHT_STEPS_PER_PASS : VP_STEPS_PER_PASS;
#pragma unroll 1
        for (int iter = 0;; iter += STEPS)
        {
            bool active = (st == ST_TRACK) || (st == ST_SHADOW) || (LOCAL && st == ST_SETUP);
            unsigned long long am = __ballot(active);
            unsigned long long wm = __ballot(!active && !(st == ST_DONE && exhausted));
            unsigned nwait = (unsigned)__popcll(wm);
            if (am == 0ull || nwait >= L.wait_lanes || (nwait > 0u && iter >= (int)L.wait_iters)) break;
            if (PROF)
            {
                const unsigned long long sm = __ballot(st == ST_SHADOW);   // (a ballot under `lane == 0` would see lane 0 only)
                if (lane == 0) { d_iter++; d_act += (unsigned)__popcll(am); d_shadow += (unsigned)__popcll(sm); }
            }
            if (!active) continue;
            segment_setup();
            tracking_step();
#pragma unroll
            for (int u = 1; u < STEPS; u++)
            {
                if (PROF)
                {
                    unsigned long long am2 = __ballot((st == ST_TRACK) || (st == ST_SHADOW) || (LOCAL && st == ST_SETUP));
                    const unsigned long long sm2 = __ballot(st == ST_SHADOW);
                    if (lane == 0) { d_iter++; d_act += (unsigned)__popcll(am2); d_shadow += (unsigned)__popcll(sm2); }
                }
                // a restart segment is set up at once while many lanes ask for one (the crawl toward and through empty bricks,
                // quirk Q6); a few stragglers -- a dense region ends a 0.05 segment every ~40 steps per lane -- wait for the
                // first step of the next pass, so that the ~100 instructions of the set-up are not run for one or two lanes
                if (LOCAL && (unsigned)__popcll(__ballot(st == ST_SETUP)) >= L.setup_lanes) segment_setup();
                tracking_step();
            }
        }
        if (PROF) { unsigned long long t = __builtin_amdgcn_s_memtime(); if (lane == 0) t_fast += t - t_mark; t_mark = t; }
    }

    if (COUNT)
    {
        // wave reduction, one atomic per counter per wave
        unsigned long long vals[12] = {c_smp, c_den, c_bnd, c_opa, c_env, c_sca, d_iter, d_act, d_outer, d_shadow, t_slow, t_fast};
#pragma unroll
        for (int q = 0; q < (PROF ? 12 : 6); q++)
        {
            unsigned long long v = vals[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) atomicAdd(&L.counters[q], v);
        }
        {
            unsigned long long v = c_load;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) atomicAdd(&L.counters[12], v);
        }
        {
            // exit flights: tests, hops through the distance field (one byte loaded each), paths ended
            unsigned long long xv[3] = {c_xtest, c_xout, c_xok};
#pragma unroll
            for (int q = 0; q < 3; q++)
            {
                unsigned long long v = xv[q];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) atomicAdd(&L.counters[13 + q], v);
            }
        }
        // block tallies are wave-uniform: lane 0 adds them
        if (PROF && lane == 0)
        {
#pragma unroll
            for (int b = 0; b < B_NBLK; b++) { atomicAdd(&L.counters[16 + 2 * b], bw[b]); atomicAdd(&L.counters[17 + 2 * b], bl[b]); }
#pragma unroll
            for (int h = 0; h < 3; h++)
#pragma unroll
                for (int q = 0; q < 8; q++) atomicAdd(&L.counters[48 + 8 * h + q], hist[h][q]);
            atomicAdd(&L.counters[72], ctrl_w); atomicAdd(&L.counters[73], ctrl_l);
        }
    }
}

// The restart crawl in front of the volume (quirk Q6), once per pixel.  The reference measures a segment's end
// t_far = min(t_exit, 0.05) from the ray origin even while the origin is outside the box (kernel.cu:1653-1654), so a camera
// 2.9 units away walks ~58 segments of 0.05 toward the volume, each one a bound fetch at the box entry point, one or two
// draws and `origin += d * 0.05` (kernel.cu:2151-2155).  While t_near >= t_far such a segment cannot collide: the free flight
// starts at dist = t_near and only grows (dist += -log(u) / sigma >= t_near >= t_far, likewise the control distance), so
// `through` (kernel.cu:2145) holds whatever the draws are.  The camera ray is the same in every frame (quirk Q3), hence so is
// this walk: the table holds, per pixel, the origin the walk ends at (the identical binary32 additions), the number of
// segments walked (16 bits: the bounded kernel counts them, kernel.cu:1716) and the number of draws they consume (one per
// segment, one more where the entry brick has a positive minimum and the decomposition estimator draws its control distance,
// kernel.cu:2048-2054).  A path then starts at the first segment that can interact.  Bit-identical by construction.
__device__ float certified_empty_distance(const SceneDev& S, f3 ro, f3 rd, const unsigned char* danger, float& cls);
template <bool QUANT>
__device__ bool chord_has_positive_minimum(const SceneDev& S, f3 ro, f3 rd);
template <bool QUANT>
__global__ __launch_bounds__(256) void crawl_table_k(SceneDev S, unsigned width, unsigned height, int control_draw, const unsigned char* danger, float4* table)
{
    unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height) return;
    unsigned py = idx / width, px = idx - py * width;
    f3 ro, rd;
    camera_ray(S, width, height, px, py, ro, rd);
    f3 inv_rd = f3{1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    // certified-empty distance of the whole camera ray (see danger_k / certified_empty_distance below), then measured from
    // where the walk ends: the origins of the restart segments differ from o + d * (walked distance) by the rounding of a
    // few dozen additions (1e-5), against a safety margin of three quarters of a cell
    float cls = 0.0f;
    float t_left = S.linear ? certified_empty_distance(S, ro, rd, danger, cls) : 0.0f;
    if (cls == 1.0f && chord_has_positive_minimum<QUANT>(S, ro, rd)) cls = 0.0f;
    unsigned segs = 0, draws = 0;
    for (; segs < 700u; segs++)   // far below the bounded kernel's 800-segment cap, and both counts stay within 16 bits
    {
        float t_near, tf;
        bool  hit = intersect_box_inv(ro, inv_rd, S, t_near, tf);
        t_near    = fmaxf(t_near, 0.0f);
        float t_far = fminf(tf, 0.05f);
        if (!hit || !(t_near >= t_far)) break;   // a NaN anywhere ends the walk: the path itself takes over
        float bx, by;
        sample_bound<QUANT>(S, ro + rd * t_near, bx, by);
        draws += (control_draw && by > 0.0f) ? 2u : 1u;
        ro = ro + rd * t_far;
        t_left -= t_far;
    }
    table[2 * idx]     = make_float4(ro.x, ro.y, ro.z, u2f(segs | (draws << 16)));
    table[2 * idx + 1] = make_float4(t_left > 0.0f ? t_left : 0.0f, cls, 0.0f, 0.0f);
}

// ---- certified-empty distances of the camera rays (global-majorant estimator).
// A trilinear fetch at p filters the 2x2x2 texels of cell c(p) = floor(p * N - 0.5) (axis_linear): it returns exactly +0
// when those eight texels are zero, i.e. when the packed cell is all-zero bits.  danger_k marks every cell that has a
// non-empty cell in its 3x3x3 neighbourhood.  empty_table_k marches each pixel's camera ray (the same ray in every frame,
// quirk Q3) through the box in steps of a quarter cell and records where it first meets a marked cell, less two steps.
// Any point of the ray before that distance lies within a quarter cell of a sample whose whole neighbourhood is empty, hence
// in an empty cell itself -- with three quarters of a cell to spare against the 1e-6 differences between the positions the
// march and the integrator compute.  The integrator's free-flight steps before that distance skip the fetch and use the +0
// it would have produced (render_k, tracking_step): same bits, ~65 fewer instructions and no memory access per step.
template <bool QUANT>
__global__ __launch_bounds__(256) void danger_k(SceneDev S, unsigned char* out, unsigned long long* marked)
{
    size_t n   = (size_t)S.nx * S.ny * S.nz;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int i = (int)(idx % S.nx), j = (int)((idx / S.nx) % S.ny), k = (int)(idx / ((size_t)S.nx * S.ny));
    bool any = false;
    for (int dk = -1; dk <= 1; dk++)
        for (int dj = -1; dj <= 1; dj++)
            for (int di = -1; di <= 1; di++)
            {
                int a = min(max(i + di, 0), S.nx - 1), b = min(max(j + dj, 0), S.ny - 1), c = min(max(k + dk, 0), S.nz - 1);
                size_t o = cell_index(S, a, b, c);
                if (QUANT) { uint2 v = S.cells_u8[o]; any = any || (v.x | v.y) != 0u; }
                else
                {
                    const float4* q = reinterpret_cast<const float4*>(S.cells_f32) + o * 2;
                    float4 lo = q[0], hi = q[1];
                    any = any || lo.x != 0.0f || lo.y != 0.0f || lo.z != 0.0f || lo.w != 0.0f || hi.x != 0.0f || hi.y != 0.0f || hi.z != 0.0f || hi.w != 0.0f;
                }
            }
    // bit 0: a non-empty cell in the 3x3x3 neighbourhood; bit 1: this cell itself is non-empty
    bool self;
    {
        size_t o = cell_index(S, i, j, k);
        if (QUANT) { uint2 v = S.cells_u8[o]; self = (v.x | v.y) != 0u; }
        else
        {
            const float4* q = reinterpret_cast<const float4*>(S.cells_f32) + o * 2;
            float4 lo = q[0], hi = q[1];
            self = lo.x != 0.0f || lo.y != 0.0f || lo.z != 0.0f || lo.w != 0.0f || hi.x != 0.0f || hi.y != 0.0f || hi.z != 0.0f || hi.w != 0.0f;
        }
    }
    out[idx] = (unsigned char)((any ? 1 : 0) | (self ? 2 : 0));
    // how much of the grid is marked (round 5): the host switches the tables that pay in EMPTY space off for volumes that have little
    if (marked)
    {
        const unsigned long long m = __ballot(any);
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m | (1ull << 63)) && m) atomicAdd(marked, (unsigned long long)__popcll(m));
    }
}
// ---- the direction table of the exit flights (render_k).  Three byte planes, one per DOMINANT axis A of a direction in cell
// units (|e_A| >= |e_B|, |e_C| with e = d * N / extent; (B, C) = the other two axes in increasing order); in plane A bit
// (e_A > 0) | (e_B > 0) << 1 | (e_C > 0) << 2 of cell c says: every cell a ray from ANY point of cell c with a direction of that class
// can meet is unmarked, i.e. has no non-empty cell in its 3x3x3 neighbourhood (danger_k bit 0).  Which cells those are: while such
// a ray advances by delta along A it advances by at most delta along B and C, and it starts less than one cell from the low corner
// of c: when its A index has advanced by m (delta < m + 1) its B and C indices have advanced by 0..m+1 -- a quarter pyramid of
// cells.  That set is the slab m = 0 (2 x 2 cells) and the sets of the four cells (A+1, B+{0,1}, C+{0,1}): a recurrence from slice to
// slice along A, one small kernel per slice (exit_dir_slice_k), from the far end of the grid for the classes that look up the axis
// and from the near end for the others.  Indices beyond the grid clamp, as the integrator's do.  An octant box (the first form
// of this table, profiles/experiments/r04_exit_octants.txt) holds three such pyramids and the rest of the octant besides: it sends a
// path on only once the WHOLE octant is clear, ~110 steps later on average on BASELINE config 2.
__global__ __launch_bounds__(256) void exit_dir_slice_k(const unsigned char* danger, unsigned char* plane, int nx, int ny, int nz, int axis, int a, int up)
{
    // one thread per cell (b, c) of slice `a` of `axis`; the four classes (sB, sC) of direction sA = up
    const int n[3]  = {nx, ny, nz};
    const int B     = axis == 0 ? 1 : 0, C = axis == 2 ? 1 : 2;
    const int nb    = n[B], nc = n[C], na = n[axis];
    const int t     = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * nc) return;
    const int b = t % nb, c = t / nb;
    const size_t st[3] = {1, (size_t)nx, (size_t)nx * ny};
    auto at = [&](int ia, int ib, int ic) -> size_t {
        ib = min(max(ib, 0), nb - 1); ic = min(max(ic, 0), nc - 1);
        return (size_t)ia * st[axis] + (size_t)ib * st[B] + (size_t)ic * st[C];
    };
    const int  an   = up ? a + 1 : a - 1;           // the slice the recurrence reads
    const bool more = an >= 0 && an < na;
    unsigned   bits = 0;
    for (int q = 0; q < 4; q++)
    {
        const int sb = (q & 1) ? 1 : -1, sc = (q & 2) ? 1 : -1;
        const unsigned bit = (up ? 1u : 0u) | ((q & 1) ? 2u : 0u) | ((q & 2) ? 4u : 0u);
        bool ok = !((danger[at(a, b, c)] | danger[at(a, b + sb, c)] | danger[at(a, b, c + sc)] | danger[at(a, b + sb, c + sc)]) & 1);
        if (ok && more)
            ok = ((plane[at(an, b, c)] & plane[at(an, b + sb, c)] & plane[at(an, b, c + sc)] & plane[at(an, b + sb, c + sc)]) >> bit) & 1u;
        bits |= ok ? (1u << bit) : 0u;
    }
    unsigned char* out = plane + at(a, b, c);
    const unsigned mine = up ? 0xaau : 0x55u;       // the classes of this direction: bit 0 = sA
    *out = (unsigned char)((*out & ~mine) | bits);
}
// ---- where a sun shadow ray has nothing left to meet (counter-based streams; render_k start_shadow).
// Per NON-EMPTY cell c (a collision needs a positive density, i.e. a non-empty cell; the others are marked 0xffff = unknown): the
// ray from the cell's centre toward the sun is marched in steps of ds = a quarter of the smallest cell edge until it is a whole
// cell outside the box, and the last sample whose cell has a non-empty cell in its 3x3x3 neighbourhood (danger_k bit 0) is
// recorded, plus two steps.  Claim: for ANY start point p in cell c, every point p + t * d with t >= out[c] * ds lies in an empty
// cell (all eight texels zero: a trilinear fetch there returns exactly +0).  Proof: p differs from the centre by at most half a
// cell per axis (three quarters in the first cell of an axis, which also takes the half texel below the first texel centre; the
// representative point is the middle of the cell's range in continuous cell coordinates), p + t * d from the marched ray's point
// at the same t by the same vector, and that point from the nearest sample by at most ds / 2 = an eighth of a cell: less than one
// cell per axis in all, so the cell index differs by at most one per axis from that of a sample beyond the recorded one, whose
// whole 3x3x3 neighbourhood is empty.  The shadow ray's own direction normalize(sun * 1e10 - p) differs from the marched
// one by 1e-7 (the origin's share of 1e10), i.e. by less than 1e-6 of the box over the whole chord.
__global__ __launch_bounds__(256) void sun_clip_k(SceneDev S, const unsigned char* danger, float ds, unsigned short* out)
{
    const size_t n   = (size_t)S.nx * S.ny * S.nz;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (!(danger[idx] & 2)) { out[idx] = 0xffffu; return; }
    const int i = (int)(idx % S.nx), j = (int)((idx / S.nx) % S.ny), k = (int)(idx / ((size_t)S.nx * S.ny));
    // the cell's range in continuous cell coordinates xb = p * N - 0.5 (cell index = clamp(floor(xb))) and its middle
    auto mid = [](int c, int nc) -> float {
        const float lo = c == 0 ? -0.5f : (float)c, hi = c == nc - 1 ? (float)nc - 0.5f : (float)c + 1.0f;
        return 0.5f * (lo + hi);
    };
    const f3 bmin = f3{S.bmin[0], S.bmin[1], S.bmin[2]};
    const f3 ext  = f3{S.bmax[0], S.bmax[1], S.bmax[2]} - bmin;
    const f3 pl   = f3{(mid(i, S.nx) + 0.5f) / (float)S.nx, (mid(j, S.ny) + 0.5f) / (float)S.ny, (mid(k, S.nz) + 0.5f) / (float)S.nz};
    const f3 o    = pl * ext + bmin;
    const f3 sun  = f3{S.sun_dir[0], S.sun_dir[1], S.sun_dir[2]};
    const f3 d    = normalize(sun * 1e10f - o);
    int      last = -1;
    unsigned m    = 0;
    for (; m < 65000u; m++)
    {
        const f3    q  = to_local(S, o + d * ((float)m * ds));
        const float xa = fma_(q.x, (float)S.nx, -0.5f), xb = fma_(q.y, (float)S.ny, -0.5f), xc = fma_(q.z, (float)S.nz, -0.5f);
        if (!(xa >= -1.5f && xa <= (float)S.nx + 0.5f && xb >= -1.5f && xb <= (float)S.ny + 0.5f && xc >= -1.5f && xc <= (float)S.nz + 0.5f)) break;
        int   a, b, c;
        float w_;
        axis_linear(q.x, S.nx, a, w_);
        axis_linear(q.y, S.ny, b, w_);
        axis_linear(q.z, S.nz, c, w_);
        if (danger[(size_t)a + (size_t)S.nx * ((size_t)b + (size_t)S.ny * c)] & 1) last = (int)m;
    }
    // a march that did not leave the box (never on a sane scene, or a NaN direction): no certificate rather than a wrong one
    out[idx] = (m >= 65000u || !(d.x == d.x && d.y == d.y && d.z == d.z)) ? 0xffffu : (unsigned short)min(last + 2, 0xfffe);
}
// distance from the origin up to which the ray (o, d) runs through certified-empty cells; 0 = no certificate.
// cls: 0 general, 1 the certificate covers the whole chord (the path can never collide), 2 the ray misses the box (the
// integrator's own test, intersectBox kernel.cu:654-680, says so: the path is the environment lookup alone)
template <bool QUANT>
__device__ bool chord_has_positive_minimum(const SceneDev& S, f3 ro, f3 rd)
{
    // a restart segment whose brick has a positive minimum draws a control distance (decomposition estimator) and may end in a
    // control collision; a light pixel must have none on its chord.  (A ray through empty cells cannot meet one -- the voxel
    // under a segment start is a texel of the cell there -- this march makes it a checked property, in steps of a quarter cell.)
    float t_near, tf;
    if (!intersect_box(ro, rd, S, t_near, tf)) return false;
    float t0   = fmaxf(t_near, 0.0f);
    float cell = fminf(fminf((S.bmax[0] - S.bmin[0]) / (float)S.nx, (S.bmax[1] - S.bmin[1]) / (float)S.ny), (S.bmax[2] - S.bmin[2]) / (float)S.nz);
    float ds   = 0.25f * cell;
    for (unsigned n = 0; n < 200000u; n++)
    {
        float tt = t0 + (float)n * ds;
        if (tt > tf + ds) return false;
        float bx, by;
        sample_bound<QUANT>(S, ro + rd * tt, bx, by);
        if (by > 0.0f) return true;
    }
    return true;
}
__device__ float certified_empty_distance(const SceneDev& S, f3 ro, f3 rd, const unsigned char* danger, float& cls)
{
    float t_near, tf;
    bool  hit = intersect_box(ro, rd, S, t_near, tf);
    cls = 0.0f;
    if (!hit && danger) cls = 2.0f;
    if (!danger || !hit || !(tf == tf) || !(t_near == t_near)) return 0.0f;
    float t0 = fmaxf(t_near, 0.0f);
    // a quarter of the smallest cell edge, in world units (the direction is a unit vector)
    float cell = fminf(fminf((S.bmax[0] - S.bmin[0]) / (float)S.nx, (S.bmax[1] - S.bmin[1]) / (float)S.ny), (S.bmax[2] - S.bmin[2]) / (float)S.nz);
    float ds   = 0.25f * cell;
    float t_empty = 1e30f;  // the whole chord is empty unless the march finds otherwise
    for (unsigned n = 0; n < 200000u; n++)
    {
        float tt = t0 + (float)n * ds;
        if (tt > tf + ds) break;
        f3    p = to_local(S, ro + rd * tt);
        int   i, j, k;
        float w;
        axis_linear(p.x, S.nx, i, w);
        axis_linear(p.y, S.ny, j, w);
        axis_linear(p.z, S.nz, k, w);
        if (danger[(size_t)i + (size_t)S.nx * ((size_t)j + (size_t)S.ny * k)] & 1)
        {
            // close to the medium: look at the cells themselves.  Every point of the ray within a quarter cell of this sample
            // lies in a cell whose index is floor(c - 0.27) or floor(c + 0.27) per axis (c = continuous cell coordinate of the
            // sample; 0.02 of a cell covers the rounding of positions, which is below 1e-4 of a cell): at most 2 x 2 x 2 cells.
            bool hit_cell = false;
            const float cx = fmaxf(fma_(p.x, (float)S.nx, -0.5f), 0.0f), cy = fmaxf(fma_(p.y, (float)S.ny, -0.5f), 0.0f),
                        cz = fmaxf(fma_(p.z, (float)S.nz, -0.5f), 0.0f);
            const int i0 = max((int)__builtin_floorf(cx - 0.27f), 0), i1 = min((int)__builtin_floorf(cx + 0.27f), S.nx - 1);
            const int j0 = max((int)__builtin_floorf(cy - 0.27f), 0), j1 = min((int)__builtin_floorf(cy + 0.27f), S.ny - 1);
            const int k0 = max((int)__builtin_floorf(cz - 0.27f), 0), k1 = min((int)__builtin_floorf(cz + 0.27f), S.nz - 1);
            for (int c = k0; c <= k1; c++)
                for (int b = j0; b <= j1; b++)
                    for (int a = i0; a <= i1; a++) hit_cell = hit_cell || (danger[(size_t)a + (size_t)S.nx * ((size_t)b + (size_t)S.ny * c)] & 2);
            if (hit_cell)
            {
                t_empty = fmaxf(tt - 2.0f * ds, 0.0f);
                break;
            }
        }
        if (n == 199999u) t_empty = 0.0f;  // never on a sane scene: no certificate rather than a wrong one
    }
    if (t_empty >= 1e29f) cls = 1.0f;
    return t_empty > t0 ? t_empty : 0.0f;
}
// per pixel two float4: [0] unused here (the local-majorant estimators keep the end of the restart crawl there), [1].x = the
// certified-empty distance of the camera ray
__global__ __launch_bounds__(256) void empty_table_k(SceneDev S, unsigned width, unsigned height, const unsigned char* danger, float4* table)
{
    unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height) return;
    unsigned py = idx / width, px = idx - py * width;
    f3 ro, rd;
    camera_ray(S, width, height, px, py, ro, rd);
    float cls;
    float te = certified_empty_distance(S, ro, rd, danger, cls);
    table[2 * idx]     = make_float4(ro.x, ro.y, ro.z, 0.0f);
    table[2 * idx + 1] = make_float4(te, cls, 0.0f, 0.0f);
}

// per pixel, add the staged samples in frame order:  acc = (((acc + s0) + s1) + ...)
__global__ __launch_bounds__(256) void reduce_stage_k(LaunchDev L)
{
    // L.pixels / L.nslots: ALL pixels of the rank (every class); L.stage: the first of L.nframes staged frames
    unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= L.nslots) return;
    unsigned pix = L.pixels[slot];
    size_t   idx = (size_t)(pix & 0xffffu) + (size_t)(pix >> 16) * L.P.width;
    float4   a   = L.out[idx];
    if (slot >= L.const_from)
    {
        // a per-pixel constant of the launch, staged once: added frame by frame all the same (binary32: n additions, not one product)
        const float4 v = L.stage_const[slot];
#pragma unroll 4
        for (int f = 0; f < L.nframes; f++) a = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
    }
    else
        for (int f = 0; f < L.nframes; f++)
        {
            float4 v = L.stage[(size_t)f * L.stage_stride + slot];
            a        = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
        }
    L.out[idx] = a;
}

// ---- the pixel lists of a rank, built on the GPU (vp_tables.cpp ensure_pixel_lists): a stable partition of the rank's pixels --
// those of its 8x8 tiles, tile by tile (row-major tiles, row-major pixels within a tile) -- by pixel class (0 general, 1 the whole
// chord is certified empty, 2 the camera ray misses the box; pixel table [1].y).  Padded slot s = 64 * (n-th owned tile) + 8 * row +
// column; slots outside the image (partial edge tiles) are dropped.  Three small kernels: per-block class counts, an exclusive scan
// of the block counts (one workgroup), and the scatter with wave-ballot ranks, so that an orbiting camera pays no host loop.
struct PixListDev
{
    unsigned width, height, rank, world, tiles_x, tiles_y;
    unsigned ntiles;            // owned tiles
    const unsigned* row_start;  // [tiles_y + 1]: index of the first owned tile of each tile row (camera-independent, from the host)
    const float4*   table;      // pixel table, or null = every pixel is general
};
#define VP_PIXLIST_BLOCK 1024
__device__ __forceinline__ unsigned pixlist_slot(const PixListDev& D, unsigned s, unsigned& pix)
{
    // class of padded slot s (3 = not a pixel of the image) and its pixel y << 16 | x
    const unsigned t = s >> 6;
    pix = 0;
    if (t >= D.ntiles) return 3u;
    // the tile row: the last ty with row_start[ty] <= t
    unsigned lo = 0, hi = D.tiles_y;
    while (hi - lo > 1u)
    {
        const unsigned mid = (lo + hi) >> 1;
        if (D.row_start[mid] <= t) lo = mid; else hi = mid;
    }
    const unsigned ty = lo;
    const unsigned first = (D.rank + D.world - tile_row_shift(ty, D.world)) % D.world;
    const unsigned tx = first + (t - D.row_start[ty]) * D.world;
    const unsigned x = tx * 8u + (s & 7u), y = ty * 8u + ((s >> 3) & 7u);
    if (x >= D.width || y >= D.height) return 3u;
    pix = y << 16 | x;
    if (!D.table) return 0u;
    const unsigned c = (unsigned)D.table[2 * ((size_t)x + (size_t)y * D.width) + 1].y;
    return c > 2u ? 0u : c;
}
__global__ __launch_bounds__(VP_PIXLIST_BLOCK) void pixlist_count_k(PixListDev D, unsigned* block_counts)
{
    __shared__ unsigned cnt[3];
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned pix;
    const unsigned c = pixlist_slot(D, blockIdx.x * VP_PIXLIST_BLOCK + threadIdx.x, pix);
    for (unsigned k = 0; k < 3; k++)
    {
        const unsigned long long m = __ballot(c == k);
        if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&cnt[k], (unsigned)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < 3) block_counts[3 * blockIdx.x + threadIdx.x] = cnt[threadIdx.x];
}
// exclusive scan of the per-block counts, class by class, in place; totals[3] = pixels per class.  One workgroup.
__global__ __launch_bounds__(VP_PIXLIST_BLOCK) void pixlist_scan_k(unsigned* block_counts, unsigned nblocks, unsigned* totals)
{
    __shared__ unsigned part[VP_PIXLIST_BLOCK];
    __shared__ unsigned carry;
    for (unsigned k = 0; k < 3; k++)
    {
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (unsigned base = 0; base < nblocks; base += VP_PIXLIST_BLOCK)
        {
            const unsigned i = base + threadIdx.x;
            const unsigned v = i < nblocks ? block_counts[3 * i + k] : 0u;
            part[threadIdx.x] = v;
            __syncthreads();
            // Hillis-Steele inclusive scan of the chunk
            for (unsigned off = 1; off < VP_PIXLIST_BLOCK; off <<= 1)
            {
                const unsigned a = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
                __syncthreads();
                part[threadIdx.x] += a;
                __syncthreads();
            }
            if (i < nblocks) block_counts[3 * i + k] = carry + part[threadIdx.x] - v;
            __syncthreads();
            if (threadIdx.x == 0) carry += part[VP_PIXLIST_BLOCK - 1];
            __syncthreads();
        }
        if (threadIdx.x == 0) totals[k] = carry;
        __syncthreads();
    }
}
__global__ __launch_bounds__(VP_PIXLIST_BLOCK) void pixlist_write_k(PixListDev D, const unsigned* block_offsets, const unsigned* totals, unsigned* out)
{
    __shared__ unsigned wave_cnt[VP_PIXLIST_BLOCK / 64][3];
    unsigned pix;
    const unsigned c    = pixlist_slot(D, blockIdx.x * VP_PIXLIST_BLOCK + threadIdx.x, pix);
    const unsigned wave = threadIdx.x >> 6;
    unsigned rank_in_wave = 0;
    for (unsigned k = 0; k < 3; k++)
    {
        const unsigned long long m = __ballot(c == k);
        if (c == k) rank_in_wave = lane_rank(m);
        if ((threadIdx.x & 63u) == 0) wave_cnt[wave][k] = (unsigned)__popcll(m);
    }
    __syncthreads();
    if (c > 2u) return;
    unsigned before = 0;
    for (unsigned w = 0; w < wave; w++) before += wave_cnt[w][c];
    const unsigned class_base = c == 0 ? 0u : c == 1 ? totals[0] : totals[0] + totals[1];
    out[class_base + block_offsets[3 * blockIdx.x + c] + before + rank_in_wave] = pix;
}

// The samples of a pixel whose camera ray misses the box are the same in every frame: set-up finds no hit (intersectBox,
// kernel.cu:1336-1345 / :2020-2031), background() is evaluated for the camera direction with throughput 1 (quirk Q3: no jitter)
// and the sample is written -- no draw is consumed.  One thread per such pixel evaluates that once, with the integrator's own
// expressions (EV_BG / EV_WRITE blocks of render_k), and writes it for every frame of the launch.
__global__ __launch_bounds__(256) void miss_fill_k(SceneDev S, LaunchDev L, int local_estimator)
{
    unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= L.nslots) return;
    const ParamDev& P = L.P;
    unsigned pix = L.pixels[slot], px = pix & 0xffffu, py = pix >> 16;
    f3 ro, rd;
    camera_ray(S, P.width, P.height, px, py, ro, rd);
    const f3 sun_dir = f3{S.sun_dir[0], S.sun_dir[1], S.sun_dir[2]};
    f3   rad  = f3{0.0f, 0.0f, 0.0f};
    const f3 thr = f3{1.0f, 1.0f, 1.0f};
    bool env_lookup = false;
    f3   bg;
    if (dot(rd, sun_dir) > S.sun_cos) bg = f3{S.sun_orig[0], S.sun_orig[1], S.sun_orig[2]};
    else { bg = eval_envmap(S, rd); env_lookup = true; }
    rad = rad + bg * thr;
    f3     r = rad * P.brightness;
    float4 v = make_float4(fmaxf(r.x, 0.0f), fmaxf(r.y, 0.0f), fmaxf(r.z, 0.0f), 0.0f);  // heat: 0 scatters / segments
    if (L.stage)
    {
        // (staged once where the add-kernel knows the slot for a constant: LaunchDev::const_from)
        const int rows = (L.slot_base + slot >= L.const_from) ? 1 : L.nframes;
        for (int f = 0; f < rows; f++) L.stage[(size_t)f * L.stage_stride + L.slot_base + slot] = v;
    }
    else
    {
        size_t idx = (size_t)px + (size_t)py * P.width;
        float4 a   = L.out[idx];
        L.out[idx] = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
    }
    if (L.counters)
    {
        atomicAdd(&L.counters[0], (unsigned long long)L.nframes);                         // samples
        if (env_lookup) atomicAdd(&L.counters[4], (unsigned long long)L.nframes);         // environment lookups
        if (local_estimator) atomicAdd(&L.counters[2], (unsigned long long)L.nframes);    // the bound fetched before the hit test
    }
}

// ---- the camera rays' way to the medium, ahead of the integrator (global-majorant estimator, counter-based streams).
// A general pixel's camera ray is the same in every frame (quirk Q3) and certified to run through empty cells up to t_empty
// (empty_table_k): until its free flight passes that distance a path does nothing but draw a pair, take the logarithm of its first
// word, add and compare -- the collision there is a null collision with density +0 that leaves a throughput of exactly 1 unchanged
// (light_identity_k has checked that for this medium; the host asks for this kernel only then).  About 500 such steps per sample at
// the default camera, 44 % of the lane-steps of the integrator's tracking loop on BASELINE config 2 -- where the lanes that make
// them sit beside lanes that fetch and collide, and wait while those do.  Here a thread per sample makes them and nothing else:
// slot s of the launch's pixel list in frame f, 64 neighbouring pixels per wave (their walks have about the same length).  It
// stops BEFORE the first flight that would pass t_empty or leave the box and leaves (distance reached, pairs used) in the sample's
// staging slot; render_k takes the sample up from there and makes that flight itself (its own test `dist < t_empty` is still in
// place: any prefix of the walk is a valid hand-over, so the step cap below costs nothing but the steps left over).
// a look-ahead batch that was cancelled before its walk began (LaunchDev::cancel): one atomic read per wave; render_k, which then
// hands out no sample, never reads the slots this walk would have written
__device__ __forceinline__ bool approach_cancelled(const LaunchDev& L)
{
    if (!L.cancel) return false;
    unsigned w = 0;
    if ((threadIdx.x & 63u) == 0u) w = __hip_atomic_load(L.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)w) >= L.batch_id;
}
template <class RNG>
__global__ __launch_bounds__(256) void approach_k(SceneDev S, LaunchDev L)
{
    if (approach_cancelled(L)) return;
    // a wave = ONE pixel in 64 consecutive frames where the launch has that many (fewer frames: 2^k frames x 64 / 2^k pixels): the
    // lanes share the ray, its certificate and the bricks it crosses, and differ only in what they draw -- their walks have
    // the same length up to the noise of a sum of exponentials
    const unsigned sh   = L.approach_fshift;
    const unsigned slot = blockIdx.x * (256u >> sh) + (threadIdx.x >> sh), fl = (blockIdx.y << sh) + (threadIdx.x & ((1u << sh) - 1u));
    if (slot >= L.nslots || fl >= (unsigned)L.nframes) return;
    const ParamDev& P = L.P;
    const unsigned pix = L.pixels[slot], px = pix & 0xffffu, py = pix >> 16;
    if (px >= P.width || py >= P.height) return;   // a partial edge tile: render_k skips the slot as well
    f3 ro, rd;
    camera_ray(S, P.width, P.height, px, py, ro, rd);
    float    t_near, tf, dist = 0.0f;
    unsigned pairs = 0, sa = 0, sb = 0;   // steps made; the stream's state before the step in hand
    RNG      rng;
    if (intersect_box(ro, rd, S, t_near, tf))
    {
        // the set-up of render_k's first segment (kernel.cu:1332-1370, depth index 0)
        if (t_near < 0.0f) t_near = 0.0f;
        dist                      = t_near;
        const float t_end         = tf;
        const float t_empty       = L.crawl[2 * ((size_t)px + (size_t)py * P.width) + 1].x;
        const float s             = hyperion_s(0 - 5);
        const float cur_density   = (1.0f - s) * P.density + s * P.density * (1.0f - P.g);
        const float sigma_t_prime = max3(f3{P.sigma_t[0], P.sigma_t[1], P.sigma_t[2]}) * cur_density;
        const float inv_sigma     = 1.0f / sigma_t_prime;
        rng.init(px, py, (unsigned)(L.frame0 + (int)fl), L.key0, L.key1);
        rng.save(sa, sb);
        for (; pairs < L.approach_steps; pairs++)
        {
            const float d2 = dist + -logf_(rng.next_a()) * inv_sigma;   // kernel.cu:1419
            if (!(d2 < t_empty) || d2 >= t_end) break;                   // the integrator's step: a fetch, or the way out
            dist = d2;
            (void)rng.next_b();   // the collision test's variate: `real` is false whatever it is; a sequential stream moves past it
            rng.save(sa, sb);
        }
    }
    L.stage[(size_t)fl * L.stage_stride + L.slot_base + slot] = make_float4(dist, u2f(sa), u2f(sb), u2f(pairs));
    if (L.counters && pairs) atomicAdd(&L.counters[1], (unsigned long long)pairs);   // density lookups the estimator makes on these steps
}

// The same for the decomposition estimator (uchar bound table): behind the crawl in front of the box (crawl_table_k) the camera ray
// walks restart segments of 0.05 through bricks whose cells it is certified not to meet non-empty (t_empty): a bound fetch, and free
// flights with the brick's majorant whose null collisions change nothing (checked per segment), until the flight leaves the segment.  A thread per sample walks every segment that ENDS before the certified distance -- no fetch can
// fall into it -- and then INTO the first one that does not, up to the flight that would need a fetch (round 4): (origin of that
// segment, distance reached in it) in the sample's staging slot, the stream's state beside it.  A segment whose brick has a positive
// minimum (the control distance is the integrator's business) or a non-neutral majorant is handed over at its origin.
template <class RNG, bool QUANT>
__global__ __launch_bounds__(256) void approach_local_k(SceneDev S, LaunchDev L)
{
    if (approach_cancelled(L)) return;
    const unsigned sh   = L.approach_fshift;   // (a wave = one pixel in 2^sh frames, as in approach_k)
    const unsigned slot = blockIdx.x * (256u >> sh) + (threadIdx.x >> sh), fl = (blockIdx.y << sh) + (threadIdx.x & ((1u << sh) - 1u));
    if (slot >= L.nslots || fl >= (unsigned)L.nframes) return;
    const ParamDev& P = L.P;
    const unsigned pix = L.pixels[slot], px = pix & 0xffffu, py = pix >> 16;
    if (px >= P.width || py >= P.height) return;
    f3 ro, rd;
    camera_ray(S, P.width, P.height, px, py, ro, rd);
    const f3     inv_rd = f3{1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    const size_t idx    = (size_t)px + (size_t)py * P.width;
    const float4 c      = L.crawl[2 * idx];
    float        t_empty = L.crawl[2 * idx + 1].x;
    ro                  = f3{c.x, c.y, c.z};
    // segment_medium() of render_k for an unscattered path
    const float s           = hyperion_s(0 - 5);
    const float reduction   = (1.0f - s) + s * (1.0f - P.g);
    const float cur_density = reduction * P.density;
    const float max_sig     = max3(f3{P.sigma_t[0], P.sigma_t[1], P.sigma_t[2]});
    RNG rng;
    rng.init(px, py, (unsigned)(L.frame0 + (int)fl), L.key0, L.key1);
    rng.skip(f2u(c.w) >> 16);   // the crawl's draws
    unsigned sa, sb;            // where the stream stands: at the origin of the segment in hand, or before the flight it stopped at
    rng.save(sa, sb);
    float    d_reached = -1.0f; // how far into the segment at `ro` the walk got (free-flight distance from its origin); -1: not at all
    unsigned long long n_steps = 0, n_segs = 0;
    for (unsigned n = 0; n < L.approach_steps; n++)
    {
        // segment_setup() of render_k (intersectSuperVolume kernel.cu:1626-1661)
        float t_near, tf;
        const bool hit = intersect_box_inv(ro, inv_rd, S, t_near, tf);
        t_near         = fmaxf(t_near, 0.0f);
        const float t_far = fminf(tf, 0.05f);
        float bx, by;
        sample_bound<QUANT>(S, ro + rd * t_near, bx, by);
        if (!hit || by > 0.0f) break;
        const float d_max         = fmaxf(0.0001f, bx);
        const float sigma_t_prime = max_sig * cur_density * d_max;
        const float inv_sigma     = 1.0f / sigma_t_prime;
        // a null collision in empty space must leave the throughput of 1 as it is for THIS majorant (most do; the constant light class
        // asks it of every majorant in the table, light_identity_k): where it does not, the integrator goes on from here
        if (!(null_collision_in_empty_space(1.0f, sigma_t_prime, inv_sigma) == 1.0f)) break;
        float    dist = t_near;
        unsigned steps = 0;
        bool     through = false;
        unsigned ta = sa, tb = sb;   // the stream before the flight in hand
        for (;;)
        {
            const float d2 = dist + -logf_(rng.next_a()) * inv_sigma;   // kernel.cu:2085
            if (d2 >= t_far) { through = true; break; }                   // t_end = min(1e20, t_far): `through`, kernel.cu:2145
            if (!(d2 < t_empty) || steps > 60000u) break;                 // a fetch: render_k's
            dist = d2;
            (void)rng.next_b();   // the collision test's variate (`real` is false whatever it is): a sequential stream moves past it
            rng.save(ta, tb);
            steps++;
        }
        n_steps += steps;
        if (!through)
        {
            // the flight in hand needs a fetch: render_k takes the path up INSIDE this segment -- its own set-up of the segment at `ro`
            // (the same bound, majorant and t_far), then the flight from `dist` with the stream as it stood before that flight
            // (round 4; before, the whole segment was handed back and walked again, ~38 steps per path of the decomposition workloads)
            d_reached = dist; sa = ta; sb = tb;
            break;
        }
        rng.save(sa, sb);
        n_segs++;
        ro      = ro + rd * t_far;   // tracking restart kernel.cu:2151-2155
        t_empty -= t_far;
    }
    const size_t item = (size_t)fl * L.stage_stride + L.slot_base + slot;
    L.stage[item] = make_float4(ro.x, ro.y, ro.z, d_reached);
    L.approach_aux[item] = make_uint2(sa, sb);   // where the stream stands: the pair index, or sampler.h's two words
    if (L.counters)
    {
        if (n_steps) atomicAdd(&L.counters[1], n_steps);   // density lookups and bound lookups the estimator makes on this stretch
        if (n_segs) atomicAdd(&L.counters[2], n_segs);
    }
}

// ---- The approach walk of the decomposition estimator, with what depends on the PIXEL alone tabulated per pixel (round 5).
// approach_local_k above sets every restart segment up per SAMPLE: box intersection, bound fetch, majorant, reciprocal, the neutrality
// test -- ~75 of its ~130 vector instructions per segment -- although the camera ray, hence the chain of segment origins
// ro_(n+1) = ro_n + rd * t_far_n and everything the set-up computes from them, is the same in every frame (quirk Q3); a wave of that
// kernel is ONE pixel in 64 frames computing the same 75 instructions in 64 lanes.  approach_segments_k walks the chain once per pixel
// of the general class (the identical binary32 operations, in the same order) and writes per segment (t_near, t_far, the brick's
// bytes, a stop flag; the segment's origin, the certified-empty distance left at its start); approach_local_tab_k reads the records --
// one 32-byte broadcast load per segment -- looks the majorant's reciprocal and its neutrality up by the byte (a 256-entry table in LDS,
// computed per launch with the integrator's own expressions: they depend on Param), and makes what is left: the draws, the
// logarithms, the sums and the compares.  Same hand-over, same bits (uchar bound tables; float tables keep approach_local_k).
#define VP_SEG_CAP 96   // records per pixel: a box diagonal of 4.7 at 0.05 per segment; a longer chain is handed over where the table ends
// layout per pixel slot, 2 * VP_SEG_CAP float4: [n] = (t_near, t_far, bound byte | stop << 8, t_empty at the segment's start),
// [VP_SEG_CAP + n] = the segment's origin (read once per sample, at the hand-over)
__global__ __launch_bounds__(256) void approach_segments_k(SceneDev S, unsigned width, unsigned height, const float4* crawl, const unsigned* pixels,
                                                           unsigned nslots, float4* seg)
{
    const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    float4* out = seg + (size_t)slot * (2 * VP_SEG_CAP);
    const unsigned pix = pixels[slot], px = pix & 0xffffu, py = pix >> 16;
    if (px >= width || py >= height) { out[0] = make_float4(0.0f, 0.0f, u2f(0x100u), 0.0f); out[VP_SEG_CAP] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); return; }
    f3 ro, rd;
    camera_ray(S, width, height, px, py, ro, rd);
    const f3     inv_rd = f3{1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    const size_t idx    = (size_t)px + (size_t)py * width;
    const float4 c      = crawl[2 * idx];
    float        t_empty = crawl[2 * idx + 1].x;
    ro                  = f3{c.x, c.y, c.z};
    for (unsigned n = 0; n < VP_SEG_CAP; n++)
    {
        // segment_setup() of render_k / approach_local_k (intersectSuperVolume kernel.cu:1626-1661): the same expressions
        float t_near, tf;
        const bool hit = intersect_box_inv(ro, inv_rd, S, t_near, tf);
        t_near         = fmaxf(t_near, 0.0f);
        const float t_far = fminf(tf, 0.05f);
        const f3  pl = to_local(S, ro + rd * t_near);
        const int bi = axis_point(pl.x, S.nx) >> S.brick_shift, bj = axis_point(pl.y, S.ny) >> S.brick_shift, bk = axis_point(pl.z, S.nz) >> S.brick_shift;
        const unsigned v = reinterpret_cast<const unsigned short*>(S.bounds_u8)[(size_t)((unsigned)bi + (unsigned)S.bnx * ((unsigned)bj + (unsigned)S.bny * (unsigned)bk))];
        // stop: the ray has left the box, the brick has a positive minimum (the control distance is the integrator's business), or the
        // table ends here.  (NOT where the certificate is used up: a flight that leaves its segment needs no fetch, and in empty
        // bricks nearly every flight does; the walk goes on as approach_local_k's does, flight by flight.)
        const bool stop = !hit || (v >> 8) != 0u || n == VP_SEG_CAP - 1u;
        out[n]              = make_float4(t_near, t_far, u2f((v & 0xffu) | (stop ? 0x100u : 0u)), t_empty);
        out[VP_SEG_CAP + n] = make_float4(ro.x, ro.y, ro.z, 0.0f);
        if (stop) break;
        ro      = ro + rd * t_far;   // tracking restart kernel.cu:2151-2155
        t_empty -= t_far;
    }
}
// A wave = one pixel in 64 frames (launched only with approach_fshift 6): it copies its pixel's chain into LDS with two coalesced loads
// and walks it from there -- what a sample waits for per segment is an LDS read, not a dependent read of global memory (the first form,
// records read from global memory segment by segment, took 17.0 of approach_local_k's 18.3 ms on C3 although it executes half the
// instructions: profiles/experiments/r05_segment_table.txt).
template <class RNG>
__global__ __launch_bounds__(256) void approach_local_tab_k(SceneDev S, LaunchDev L)
{
    // per launch: for every byte a brick maximum can be, the reciprocal of the segment's majorant and whether a null collision in
    // empty space leaves a throughput of 1 as it is under it -- approach_local_k's expressions, once per workgroup instead of per segment
    __shared__ float4 chain[4][VP_SEG_CAP];
    __shared__ float inv_tab[256];
    __shared__ unsigned char ok_tab[256];
    const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned slot = blockIdx.x * 4u + wv, fl = (blockIdx.y << 6) + lane;
    {
        const ParamDev& P = L.P;
        const float s           = hyperion_s(0 - 5);
        const float reduction   = (1.0f - s) + s * (1.0f - P.g);
        const float cur_density = reduction * P.density;
        const float max_sig     = max3(f3{P.sigma_t[0], P.sigma_t[1], P.sigma_t[2]});
        const unsigned b = threadIdx.x;   // 256 threads
        const float d_max         = fmaxf(0.0001f, (float)b * VP_U8_SCALE);
        const float sigma_t_prime = max_sig * cur_density * d_max;
        const float inv_sigma     = 1.0f / sigma_t_prime;
        inv_tab[b] = inv_sigma;
        ok_tab[b]  = null_collision_in_empty_space(1.0f, sigma_t_prime, inv_sigma) == 1.0f ? 1 : 0;
        if (slot < L.nslots)
        {
            // (records behind the chain's stop record were never written: read, never used)
            const float4* rec = L.seg_table + (size_t)slot * (2 * VP_SEG_CAP);
            chain[wv][lane] = rec[lane];
            if (lane < VP_SEG_CAP - 64u) chain[wv][64u + lane] = rec[64u + lane];
        }
    }
    __syncthreads();
    if (approach_cancelled(L)) return;
    if (slot >= L.nslots || fl >= (unsigned)L.nframes) return;
    const ParamDev& P = L.P;
    const unsigned pix = L.pixels[slot], px = pix & 0xffffu, py = pix >> 16;
    if (px >= P.width || py >= P.height) return;
    const size_t idx = (size_t)px + (size_t)py * P.width;
    RNG rng;
    rng.init(px, py, (unsigned)(L.frame0 + (int)fl), L.key0, L.key1);
    rng.skip(f2u(L.crawl[2 * idx].w) >> 16);   // the crawl's draws
    unsigned sa, sb;
    rng.save(sa, sb);
    float    d_reached = -1.0f;
    unsigned n = 0;
    unsigned long long n_steps = 0, n_segs = 0;
    for (;; n++)
    {
        const float4 A = chain[wv][n];
        const unsigned bits = f2u(A.z);
        if ((bits & 0x100u) || n >= L.approach_steps || !ok_tab[bits & 0xffu]) break;   // handed over at this segment's origin
        const float inv_sigma = inv_tab[bits & 0xffu], t_far = A.y, t_empty = A.w;
        float    dist = A.x;
        unsigned steps = 0;
        bool     through = false;
        unsigned ta = sa, tb = sb;   // the stream before the flight in hand
        for (;;)
        {
            const float d2 = dist + -logf_(rng.next_a()) * inv_sigma;   // kernel.cu:2085
            if (d2 >= t_far) { through = true; break; }
            if (!(d2 < t_empty) || steps > 60000u) break;                 // a fetch: render_k's
            dist = d2;
            (void)rng.next_b();
            rng.save(ta, tb);
            steps++;
        }
        n_steps += steps;
        if (!through) { d_reached = dist; sa = ta; sb = tb; break; }
        rng.save(sa, sb);
        n_segs++;
    }
    const float4 O = L.seg_table[(size_t)slot * (2 * VP_SEG_CAP) + VP_SEG_CAP + n];
    const size_t item = (size_t)fl * L.stage_stride + L.slot_base + slot;
    L.stage[item] = make_float4(O.x, O.y, O.z, d_reached);
    L.approach_aux[item] = make_uint2(sa, sb);
    if (L.counters)
    {
        if (n_steps) atomicAdd(&L.counters[1], n_steps);
        if (n_segs) atomicAdd(&L.counters[2], n_segs);
    }
}

// expand a dense volume into per-voxel 2x2x2 neighbourhood cells (clamped at the border)
__device__ __forceinline__ size_t pack_index(int nx, int ny, int i, int j, int k, int bricks)
{
    if (!bricks) return (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k);
    const size_t nbx = ((size_t)nx + 3) >> 2, nby = ((size_t)ny + 3) >> 2;
    return ((((size_t)i >> 2) + nbx * (((size_t)j >> 2) + nby * ((size_t)k >> 2))) << 6) | (size_t)(((k & 3) << 4) | ((j & 3) << 2) | (i & 3));
}
__global__ void pack_cells_u8_k(const unsigned char* vol, uint2* cells, int nx, int ny, int nz, int bricks)
{
    size_t n   = (size_t)nx * ny * nz;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int i = (int)(idx % nx), j = (int)((idx / nx) % ny), k = (int)(idx / ((size_t)nx * ny));
    int i1 = i + 1 < nx ? i + 1 : nx - 1, j1 = j + 1 < ny ? j + 1 : ny - 1, k1 = k + 1 < nz ? k + 1 : nz - 1;
    auto at = [&](int a, int b, int c) -> unsigned { return vol[(size_t)a + (size_t)nx * ((size_t)b + (size_t)ny * c)]; };
    uint2 c;
    c.x = at(i, j, k) | (at(i1, j, k) << 8) | (at(i, j1, k) << 16) | (at(i1, j1, k) << 24);
    c.y = at(i, j, k1) | (at(i1, j, k1) << 8) | (at(i, j1, k1) << 16) | (at(i1, j1, k1) << 24);
    cells[pack_index(nx, ny, i, j, k, bricks)] = c;
}
__global__ void pack_cells_f32_k(const float* vol, float* cells, int nx, int ny, int nz, int bricks)
{
    size_t n   = (size_t)nx * ny * nz;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int i = (int)(idx % nx), j = (int)((idx / nx) % ny), k = (int)(idx / ((size_t)nx * ny));
    int i1 = i + 1 < nx ? i + 1 : nx - 1, j1 = j + 1 < ny ? j + 1 : ny - 1, k1 = k + 1 < nz ? k + 1 : nz - 1;
    auto at = [&](int a, int b, int c) -> float { return vol[(size_t)a + (size_t)nx * ((size_t)b + (size_t)ny * c)]; };
    float4* q = reinterpret_cast<float4*>(cells) + pack_index(nx, ny, i, j, k, bricks) * 2;
    q[0] = make_float4(at(i, j, k), at(i1, j, k), at(i, j1, k), at(i1, j1, k));
    q[1] = make_float4(at(i, j, k1), at(i1, j, k1), at(i, j1, k1), at(i1, j1, k1));
}

// ---- local (max,min) bound table (compute_volume_value_bound_, host.cpp:1088-1267) on the GPU.
// Per voxel, max and min of the density over the (2r+1)^3 window clipped to the grid; max/min filters are
// separable, so three streaming passes (x, y, z) of a 1-D window each; then one entry per brick.
// Pairs are packed (max | min << 8) for uchar volumes and float2 for float volumes.
struct PairU8
{
    typedef unsigned char  T;
    typedef unsigned short P;
    static __device__ __forceinline__ P make(T mx, T mn) { return (P)((unsigned)mx | ((unsigned)mn << 8)); }
    static __device__ __forceinline__ T mx(P p) { return (T)(p & 0xffu); }
    static __device__ __forceinline__ T mn(P p) { return (T)(p >> 8); }
};
struct PairF32
{
    typedef float  T;
    typedef float2 P;
    static __device__ __forceinline__ P make(T mx, T mn) { return make_float2(mx, mn); }
    static __device__ __forceinline__ T mx(P p) { return p.x; }
    static __device__ __forceinline__ T mn(P p) { return p.y; }
};
template <class PR>
__global__ __launch_bounds__(256) void bounds_init_k(const typename PR::T* vol, typename PR::P* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = PR::make(vol[i], vol[i]);
}
// one 1-D pass along `axis` (stride in elements, extent na); threads run along x so every read is coalesced
template <class PR>
__global__ __launch_bounds__(256) void bounds_pass_k(const typename PR::P* in, typename PR::P* out, int nx, int ny, int nz, int axis, int r)
{
    size_t n   = (size_t)nx * ny * nz;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int    i = (int)(idx % nx), j = (int)((idx / nx) % ny), k = (int)(idx / ((size_t)nx * ny));
    int    o = axis == 0 ? i : (axis == 1 ? j : k), na = axis == 0 ? nx : (axis == 1 ? ny : nz);
    size_t stride = axis == 0 ? 1 : (axis == 1 ? (size_t)nx : (size_t)nx * ny);
    int    lo = o - r < 0 ? 0 : o - r, hi = o + r > na - 1 ? na - 1 : o + r;
    const typename PR::P* base = in + idx - (size_t)o * stride;
    typename PR::P v  = base[(size_t)lo * stride];
    typename PR::T mx = PR::mx(v), mn = PR::mn(v);
    for (int q = lo + 1; q <= hi; q++)
    {
        v = base[(size_t)q * stride];
        typename PR::T a = PR::mx(v), b = PR::mn(v);
        mx = a > mx ? a : mx;
        mn = b < mn ? b : mn;
    }
    out[idx] = PR::make(mx, mn);
}
template <class PR>
__global__ __launch_bounds__(256) void bounds_brick_k(const typename PR::P* in, typename PR::P* out, int nx, int ny, int nz, int brick, int bnx,
                                                       int bny, int bnz)
{
    size_t nb  = (size_t)bnx * bny * bnz;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nb) return;
    int bi = (int)(idx % bnx), bj = (int)((idx / bnx) % bny), bk = (int)(idx / ((size_t)bnx * bny));
    int i1 = min((bi + 1) * brick, nx), j1 = min((bj + 1) * brick, ny), k1 = min((bk + 1) * brick, nz);
    typename PR::P v  = in[(size_t)(bi * brick) + (size_t)nx * ((size_t)(bj * brick) + (size_t)ny * (bk * brick))];
    typename PR::T mx = PR::mx(v), mn = PR::mn(v);
    for (int k = bk * brick; k < k1; k++)
        for (int j = bj * brick; j < j1; j++)
            for (int i = bi * brick; i < i1; i++)
            {
                v = in[(size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k)];
                typename PR::T a = PR::mx(v), b = PR::mn(v);
                mx = a > mx ? a : mx;
                mn = b < mn ? b : mn;
            }
    out[idx] = PR::make(mx, mn);  // .x = max, .y = min (host.cpp:1141-1144)
}

// _precompute_opacity kernel.cu:483-524 with intersect_box :453-481; one thread per voxel
template <bool QUANT>
__global__ __launch_bounds__(256) void opacity_k(SceneDev S, f3 light_dir, float* out)
{
    size_t n   = (size_t)S.nx * S.ny * S.nz;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int i = (int)(idx % S.nx), j = (int)((idx / S.nx) % S.ny), k = (int)(idx / ((size_t)S.nx * S.ny));
    const float dt = 0.001f;
    f3 s0    = f3{((float)i + 0.5f) / (float)S.nx, ((float)j + 0.5f) / (float)S.ny, ((float)k + 0.5f) / (float)S.nz};
    f3 bmin  = f3{S.bmin[0], S.bmin[1], S.bmin[2]};
    f3 ext   = f3{S.bmax[0], S.bmax[1], S.bmax[2]} - bmin;
    f3 start = s0 * ext + bmin;
    float tn, tf;
    bool  hit = intersect_box(start, light_dir, S, tn, tf);
    if (tn <= 0.0f) tn = 0.0f;
    float opacity = 0.0f;
    if (hit)
    {
        for (float t = tn; t < tf; t += dt) opacity += sample_density01<QUANT>(S, start + light_dir * t);
        opacity *= dt;
    }
    out[idx] = opacity;
}

// The same march with the density grid STAGED THROUGH LDS (north_star: "the density grid ... staged through LDS with coalesced HBM
// loads"; VERDICT rows N1).  This is the one kernel of the path whose rays are coherent: every voxel of the grid marches along the
// same direction with the same step, so an 8x8x8 block of voxels is a rigid body of 512 sample points that translates through the
// grid, and the texels it filters over the next few steps are one small box.  A workgroup owns such a block; per chunk of steps it
// stages the 16x16x16 texels around the block's positions (raw bytes, clamped at the grid's faces exactly as the packed cells are:
// 4 KiB, ONE coalesced 8-byte load per thread instead of 512 x chunk 8-byte gathers) and then every thread filters its eight taps
// from LDS with the arithmetic of filter_cell_u8 -- same taps, same weights, same order: the same bits as opacity_k, which stays the
// definition (and the fall-back for float volumes).  The tile's place is computed from block-uniform values; a sample whose cell is
// not inside it (never in practice: one cell of margin on every side) takes the packed cell from global memory as opacity_k does, so
// the result does not depend on where the tile lies.
#define VP_OPA_B 8
#define VP_OPA_T 16
__global__ __launch_bounds__(VP_OPA_B * VP_OPA_B * VP_OPA_B) void opacity_lds_k(SceneDev S, f3 light_dir, float* out, int chunk)
{
    __shared__ __attribute__((aligned(4))) unsigned char tile[VP_OPA_T * VP_OPA_T * VP_OPA_T];
    const unsigned nbx = ((unsigned)S.nx + VP_OPA_B - 1u) / VP_OPA_B, nby = ((unsigned)S.ny + VP_OPA_B - 1u) / VP_OPA_B;
    const unsigned bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
    const int tx = threadIdx.x & 7, ty = (threadIdx.x >> 3) & 7, tz = threadIdx.x >> 6;
    const int i = (int)bx * VP_OPA_B + tx, j = (int)by * VP_OPA_B + ty, k = (int)bz * VP_OPA_B + tz;
    const bool inside = i < S.nx && j < S.ny && k < S.nz;
    const float dt = 0.001f;
    const f3 bmin  = f3{S.bmin[0], S.bmin[1], S.bmin[2]};
    const f3 ext   = f3{S.bmax[0], S.bmax[1], S.bmax[2]} - bmin;
    // (the expressions of opacity_k)
    const f3 s0    = f3{((float)i + 0.5f) / (float)S.nx, ((float)j + 0.5f) / (float)S.ny, ((float)k + 0.5f) / (float)S.nz};
    const f3 start = s0 * ext + bmin;
    float tn, tf;
    const bool hit = intersect_box(start, light_dir, S, tn, tf);
    if (tn <= 0.0f) tn = 0.0f;
    float t = tn, opacity = 0.0f;
    bool  active = inside && hit;
    // where the block is: its lowest voxel's sample point, and the lockstep march parameter (the same additions in every thread)
    const f3 c0 = f3{((float)((int)bx * VP_OPA_B) + 0.5f) / (float)S.nx, ((float)((int)by * VP_OPA_B) + 0.5f) / (float)S.ny,
                     ((float)((int)bz * VP_OPA_B) + 0.5f) / (float)S.nz} * ext + bmin;
    float tb = 0.0f;
    const float nn[3] = {(float)S.nx, (float)S.ny, (float)S.nz};
    const float dd[3] = {light_dir.x, light_dir.y, light_dir.z};
    const float cc[3] = {c0.x, c0.y, c0.z};
    for (;;)
    {
        if (!__syncthreads_or(active && t < tf)) break;   // (also: everybody is done with the previous tile)
        int lo[3];
#pragma unroll
        for (int a = 0; a < 3; a++)
        {
            // texel coordinate of the block's lowest sample point at the first and at the last step of this chunk, the lower of the
            // two, less one cell of margin against the roundings of the per-thread positions
            const float t1 = tb + dt * (float)(chunk - 1);
            const float u0 = fma_((cc[a] + dd[a] * tb - S.bmin[a]) * S.linv[a], nn[a], -0.5f);
            const float u1 = fma_((cc[a] + dd[a] * t1 - S.bmin[a]) * S.linv[a], nn[a], -0.5f);
            float u = __builtin_floorf(fminf(u0, u1)) - 1.0f;
            u = fminf(fmaxf(u, -16.0f), nn[a]);   // (a direction with a NaN or an infinity: any tile will do, every sample falls back)
            lo[a] = (int)u;
        }
        {
            // ONE 8-byte load per thread fills the tile: the packed cell of voxel c holds the texels (c, c + 1) of every axis, clamped at the
            // upper faces as the tile wants them, so the 8^3 cells at the even tile coordinates are its 16^3 texels.  Below the lower
            // faces the tile wants texel 0 twice where the cell of voxel 0 holds texels 0 and 1: tap 0 is used for both there.
            const int ex = 2 * ((int)threadIdx.x & 7), ey = 2 * (((int)threadIdx.x >> 3) & 7), ez = 2 * ((int)threadIdx.x >> 6);
            const int gx = lo[0] + ex, gy = lo[1] + ey, gz = lo[2] + ez;
            const int qx = gx < 0 ? 0 : (gx > S.nx - 1 ? S.nx - 1 : gx), qy = gy < 0 ? 0 : (gy > S.ny - 1 ? S.ny - 1 : gy),
                      qz = gz < 0 ? 0 : (gz > S.nz - 1 ? S.nz - 1 : gz);
            uint2 c = S.cells_u8[cell_index(S, qx, qy, qz)];
            // bytes of c: (x, y, z) taps 000 100 010 110 | 001 101 011 111
            if (gx < 0) { c.x = (c.x & 0x00ff00ffu) * 0x101u; c.y = (c.y & 0x00ff00ffu) * 0x101u; }                      // x tap 1 := x tap 0
            if (gy < 0) { c.x = (c.x & 0x0000ffffu) * 0x10001u; c.y = (c.y & 0x0000ffffu) * 0x10001u; }                  // y tap 1 := y tap 0
            if (gz < 0) c.y = c.x;                                                                                      // z tap 1 := z tap 0
            unsigned short* t16 = reinterpret_cast<unsigned short*>(tile);
            const int       w   = (ex + VP_OPA_T * (ey + VP_OPA_T * ez)) >> 1;   // 16-bit word index of tile texel (ex, ey, ez)
            t16[w]                                          = (unsigned short)(c.x & 0xffffu);
            t16[w + VP_OPA_T / 2]                           = (unsigned short)(c.x >> 16);
            t16[w + VP_OPA_T * VP_OPA_T / 2]                = (unsigned short)(c.y & 0xffffu);
            t16[w + VP_OPA_T * VP_OPA_T / 2 + VP_OPA_T / 2] = (unsigned short)(c.y >> 16);
        }
        __syncthreads();
        for (int s = 0; s < chunk; s++)
        {
            if (active && t < tf)
            {
                const f3 p = to_local(S, start + light_dir * t);
                int   ci, cj, ck;
                float fx, fy, fz;
                if (S.linear)
                {
                    axis_linear(p.x, S.nx, ci, fx);
                    axis_linear(p.y, S.ny, cj, fy);
                    axis_linear(p.z, S.nz, ck, fz);
                }
                else
                {
                    ci = axis_point(p.x, S.nx); cj = axis_point(p.y, S.ny); ck = axis_point(p.z, S.nz);
                    fx = fy = fz = 0.0f;
                }
                const int rx = ci - lo[0], ry = cj - lo[1], rz = ck - lo[2];
                float v;
                if ((unsigned)rx < VP_OPA_T - 1u && (unsigned)ry < VP_OPA_T - 1u && (unsigned)rz < VP_OPA_T - 1u)
                {
                    const unsigned char* q = tile + (rx + VP_OPA_T * (ry + VP_OPA_T * rz));
                    v = filter_texels_u8((float)q[0], (float)q[1], (float)q[VP_OPA_T], (float)q[VP_OPA_T + 1], (float)q[VP_OPA_T * VP_OPA_T],
                                         (float)q[VP_OPA_T * VP_OPA_T + 1], (float)q[VP_OPA_T * VP_OPA_T + VP_OPA_T], (float)q[VP_OPA_T * VP_OPA_T + VP_OPA_T + 1],
                                         fx, fy, fz);
                }
                else
                    v = filter_cell_u8(S.cells_u8[cell_index(S, ci, cj, ck)], fx, fy, fz);
                opacity += v;
                t += dt;
            }
            tb += dt;
        }
    }
    if (inside) out[(size_t)i + (size_t)S.nx * ((size_t)j + (size_t)S.ny * (size_t)k)] = hit ? opacity * dt : 0.0f;
}

// __scale kernel.cu:2333-2341
__global__ void scale_k(float4* dst, const float4* src, int size, float s)
{
    int idx = threadIdx.x + blockIdx.x * blockDim.x;
    if (idx >= size) return;
    float4 v = src[idx];
    dst[idx] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
}
__device__ __forceinline__ float pow_pos(float x, float y) { return x <= 0.0f ? 0.0f : expf_(logf_(x) * y); }
// __gamma_correct kernel.cu:2348-2357 (inv_gamma = 1/gamma computed by the host wrapper, :2361)
__global__ void gamma_k(float4* dst, const float4* src, int size, float s, float inv_gamma)
{
    int idx = threadIdx.x + blockIdx.x * blockDim.x;
    if (idx >= size) return;
    float4 v = src[idx];
    dst[idx] = make_float4(pow_pos(v.x * s, inv_gamma), pow_pos(v.y * s, inv_gamma), pow_pos(v.z * s, inv_gamma), 1.0f);
}

// FractalJuliaSet kernel.cu:84-140 voxelised at texel centres over [-1,1]^3:
// q <- q^2 + c, c = (-0.2, 0.8, 0, 0), radius 1.4, maxIter 30, escape dot(q,q) >= 10, density = iter > 27
__global__ void julia_k(unsigned char* grid, int n)
{
    size_t total = (size_t)n * n * n;
    size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int   i = (int)(idx % n), j = (int)((idx / n) % n), k = (int)(idx / ((size_t)n * n));
    float fn = (float)n;
    float px = ((float)i + 0.5f) / fn * 2.0f - 1.0f, py = ((float)j + 0.5f) / fn * 2.0f - 1.0f,
          pz = ((float)k + 0.5f) / fn * 2.0f - 1.0f;
    float qx = px * 1.4f, qy = py * 1.4f, qz = pz * 1.4f, qw = 0.0f;
    int   iter = 0;
    float dd;
    do
    {
        float r0 = qx * qx - (qy * qy + qz * qz + qw * qw);
        float s  = qx * 2.0f;
        float ry = qy * s, rz = qz * s, rw = qw * s;
        qx = r0 + -0.2f; qy = ry + 0.8f; qz = rz + 0.0f; qw = rw + 0.0f;
        dd = qx * qx + qy * qy + qz * qz + qw * qw;
    } while (dd < 10.0f && iter++ < 30);
    grid[idx] = iter > 27 ? 255 : 0;
}

// A FLAGGED SYNTHETIC cloud for the 512^3 workloads ("fBm-thresholded 512^3 cloud ... parameters
// logged -- never silently substituted"; the WDAS data set and OpenVDB do not exist in this image): five octaves of value noise
// on hashed lattices (wang_hash of the lattice point and the seed), thresholded, with a soft spherical edge; float densities
// in [0,1], NOT binary, so that the local minima / maxima of the bound table differ and the control component is active.
// Every operation is a binary32 add / multiply / divide / sqrt / floor in a fixed order: the oracle's restatement
// (vpo_cloud_voxelize) computes the same bits.
__device__ __forceinline__ float cloud_lattice(int ix, int iy, int iz, unsigned seed)
{
    const unsigned h = wang_hash(((unsigned)ix * 73856093u) ^ ((unsigned)iy * 19349663u) ^ ((unsigned)iz * 83492791u) ^ seed);
    return (float)(h & 0xffffffu) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float cloud_noise(float x, float y, float z, unsigned seed)
{
    const float fx = __builtin_floorf(x), fy = __builtin_floorf(y), fz = __builtin_floorf(z);
    const int   ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float tx = x - fx, ty = y - fy, tz = z - fz;
    tx = (tx * tx) * (3.0f - 2.0f * tx);
    ty = (ty * ty) * (3.0f - 2.0f * ty);
    tz = (tz * tz) * (3.0f - 2.0f * tz);
    const float c000 = cloud_lattice(ix, iy, iz, seed), c100 = cloud_lattice(ix + 1, iy, iz, seed);
    const float c010 = cloud_lattice(ix, iy + 1, iz, seed), c110 = cloud_lattice(ix + 1, iy + 1, iz, seed);
    const float c001 = cloud_lattice(ix, iy, iz + 1, seed), c101 = cloud_lattice(ix + 1, iy, iz + 1, seed);
    const float c011 = cloud_lattice(ix, iy + 1, iz + 1, seed), c111 = cloud_lattice(ix + 1, iy + 1, iz + 1, seed);
    const float x00 = c000 + (c100 - c000) * tx, x10 = c010 + (c110 - c010) * tx;
    const float x01 = c001 + (c101 - c001) * tx, x11 = c011 + (c111 - c011) * tx;
    const float y0 = x00 + (x10 - x00) * ty, y1 = x01 + (x11 - x01) * ty;
    return y0 + (y1 - y0) * tz;
}
__global__ void cloud_k(float* grid, int n, unsigned seed)
{
    size_t total = (size_t)n * n * n;
    size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int   i = (int)(idx % n), j = (int)((idx / n) % n), k = (int)(idx / ((size_t)n * n));
    float fn = (float)n;
    float px = ((float)i + 0.5f) / fn * 2.0f - 1.0f, py = ((float)j + 0.5f) / fn * 2.0f - 1.0f, pz = ((float)k + 0.5f) / fn * 2.0f - 1.0f;
    float sum = 0.0f, amp = 0.5f, freq = 2.0f;
    for (int o = 0; o < 5; o++)
    {
        sum = sum + amp * cloud_noise(px * freq + 17.0f, py * freq + 17.0f, pz * freq + 17.0f, seed + (unsigned)o * 0x9E3779B9u);
        amp = amp * 0.5f;
        freq = freq * 2.0f;
    }
    float v = (sum * (1.0f / 0.96875f) - 0.44f) * 3.0f;   // octave amplitudes sum to 0.96875
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    float r = __builtin_sqrtf(px * px + py * py + pz * pz);
    float edge = fminf(fmaxf((1.2f - r) * (1.0f / 0.4f), 0.0f), 1.0f);
    grid[idx] = v * edge;
}

// dst += src (float4): sums per-shard accumulators of several contexts on one device
__global__ void accumulate_k(float4* dst, const float4* src, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = dst[i], b = src[i];
    dst[i]   = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// ---- test kernels
// the phase-function block of the integrator (kernel.cu:2301-2303 with :557-598) and HGPhaseFunction::evaluate (:600-603)
__global__ void test_hg_k(const float* g, const float* r0, const float* r1, const float* nrm, const float* cosq, float* dir, float* ev, int n)
{
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    Frame fr(f3{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]});
    f3    d = normalize(fr.to_world(hg_sample_local(g[i], r0[i], r1[i])));
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
    ev[i] = hg_eval(g[i], cosq[i]);
}
// intersectBox kernel.cu:654-680 against the current volume box
__global__ void test_box_k(SceneDev S, const float* o, const float* d, int* hit, float* tn, float* tf, int n)
{
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    float a, b;
    hit[i] = intersect_box(f3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, f3{d[3 * i], d[3 * i + 1], d[3 * i + 2]}, S, a, b) ? 1 : 0;
    tn[i] = a; tf[i] = b;
}
// eval_envmap kernel.cu:956-973
__global__ void test_env_k(SceneDev S, const float* d, float* out, int n)
{
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    f3 c = eval_envmap(S, f3{d[3 * i], d[3 * i + 1], d[3 * i + 2]});
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}
__global__ void test_math_k(int which, const float* in, float* out, int n)
{
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    float x = in[i], s, c, r;
    switch (which)
    {
        case 0: r = logf_(x); break;
        case 1: r = expf_(x); break;
        case 2: sincosf_(x, s, c); r = s; break;
        case 3: sincosf_(x, s, c); r = c; break;
        case 4: r = acosf_(x); break;
        case 5: r = atanf_(x); break;
        default: r = pow15f_(x); break;
    }
    out[i] = r;
}
template <class RNG>
__global__ void test_rng_k(unsigned x, unsigned y, unsigned frame, unsigned k0, unsigned k1, int n, float* out)
{
    if (threadIdx.x || blockIdx.x) return;
    RNG r;
    r.init(x, y, frame, k0, k1);
    for (int i = 0; i < n; i++) out[i] = (i & 1) ? r.next_b() : r.next_a();
}
template <bool QUANT>
__global__ void test_density_k(SceneDev S, const float* pos, float* out, int n)
{
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    out[i] = sample_density01<QUANT>(S, f3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]});
}

// ------------------------------------------------------------------ host-side launchers
template <int EST, class RNG, int LDSB, bool ACH, bool MIS>
static void launch_render5(const SceneDev& S, const LaunchDev& L, bool quant, bool count, int blocks, hipStream_t st)
{
    const dim3 blk(LDSB == 1 ? VP_BLOCK_LDS : VP_BLOCK);
    if constexpr (LDSB == 2)
    {
        // the compact LDS table: the timed instance only (the host sends counting launches and look-ahead batches elsewhere)
        hipLaunchKernelGGL((render_k<EST, RNG, true, false, 2, ACH, false, 0>), dim3(blocks), blk, 0, st, S, L);
        return;
    }
    else
    {
#ifdef VP_DEV_BUILD
    quant = true;
    if (MIS) return;
    if constexpr (!MIS)
    {
        if (count) hipLaunchKernelGGL((render_k<EST, RNG, true, true, LDSB, ACH, false, 0>), dim3(blocks), blk, 0, st, S, L);
        else if (L.cancel) hipLaunchKernelGGL((render_k<EST, RNG, true, false, LDSB, ACH, false, 0, false, true>), dim3(blocks), blk, 0, st, S, L);
        else hipLaunchKernelGGL((render_k<EST, RNG, true, false, LDSB, ACH, false, 0>), dim3(blocks), blk, 0, st, S, L);
    }
#else
    // (look-ahead batches of the shipped configuration -- passive environment -- run the instance that can be stopped at once: CANCEL)
    constexpr bool CAN = !MIS;
    if (quant)
    {
        if (count) hipLaunchKernelGGL((render_k<EST, RNG, true, true, LDSB, ACH, MIS, 0>), dim3(blocks), blk, 0, st, S, L);
        else if (CAN && L.cancel) hipLaunchKernelGGL((render_k<EST, RNG, true, false, LDSB, ACH, MIS, 0, false, CAN>), dim3(blocks), blk, 0, st, S, L);
        else hipLaunchKernelGGL((render_k<EST, RNG, true, false, LDSB, ACH, MIS, 0>), dim3(blocks), blk, 0, st, S, L);
    }
    else
    {
        if (count) hipLaunchKernelGGL((render_k<EST, RNG, false, true, false, ACH, MIS, 0>), dim3(blocks), dim3(VP_BLOCK), 0, st, S, L);
        else if (CAN && L.cancel) hipLaunchKernelGGL((render_k<EST, RNG, false, false, false, ACH, MIS, 0, false, CAN>), dim3(blocks), dim3(VP_BLOCK), 0, st, S, L);
        else hipLaunchKernelGGL((render_k<EST, RNG, false, false, false, ACH, MIS, 0>), dim3(blocks), dim3(VP_BLOCK), 0, st, S, L);
    }
#endif
    }
}
// scalar tracking builds (the reference's compiled-out SPECTRAL_TRACKING 0 / MULTI_CHANNEL 1): three-channel throughput, no
// LDS / MIS / counting specialisations
template <int EST, class RNG>
static void launch_render_scalar(const SceneDev& S, const LaunchDev& L, bool quant, int trk, int blocks, hipStream_t st)
{
    const dim3 blk(VP_BLOCK);
    if (quant)
    {
        if (trk == 1) hipLaunchKernelGGL((render_k<EST, RNG, true, false, false, false, false, 1>), dim3(blocks), blk, 0, st, S, L);
        else hipLaunchKernelGGL((render_k<EST, RNG, true, false, false, false, false, 2>), dim3(blocks), blk, 0, st, S, L);
    }
    else
    {
        if (trk == 1) hipLaunchKernelGGL((render_k<EST, RNG, false, false, false, false, false, 1>), dim3(blocks), blk, 0, st, S, L);
        else hipLaunchKernelGGL((render_k<EST, RNG, false, false, false, false, false, 2>), dim3(blocks), blk, 0, st, S, L);
    }
}
template <int EST, class RNG, int LDSB>
static void launch_render3(const SceneDev& S, const LaunchDev& L, bool quant, bool count, bool ach, bool mis, int blocks, hipStream_t st)
{
    if (mis)
    {
        // active environment sampling: the rarely used build, kept off the LDS specialisation
        if (ach) launch_render5<EST, RNG, 0, true, true>(S, L, quant, count, blocks, st);
        else launch_render5<EST, RNG, 0, false, true>(S, L, quant, count, blocks, st);
    }
    else if (ach) launch_render5<EST, RNG, LDSB, true, false>(S, L, quant, count, blocks, st);
    else launch_render5<EST, RNG, LDSB, false, false>(S, L, quant, count, blocks, st);
}

// VP_RNG_PHILOX7: the shipped configuration only (spectral tracking, passive environment)
template <int EST, int LDSB>
static void launch_render_p7(const SceneDev& S, const LaunchDev& L, bool quant, bool count, bool ach, int blocks, hipStream_t st)
{
    if (ach) launch_render5<EST, RngPhilox7, LDSB, true, false>(S, L, quant, count, blocks, st);
    else launch_render5<EST, RngPhilox7, LDSB, false, false>(S, L, quant, count, blocks, st);
}

template <int EST, class RNGT>
static void launch_light2(const SceneDev& S, const LaunchDev& L, bool quant, bool count, bool ach, int blocks, hipStream_t st)
{
    // A light path never collides with matter: its throughput starts at (1,1,1) and every null collision in empty space multiplies
    // the three channels by the same factor (sigma_t' - 0 in each), so they stay bitwise equal whatever the medium -- the
    // one-channel (ACH) instance computes exactly what the three-channel one would.
    const dim3 g(blocks), b(VP_BLOCK);
    // QUANT only selects how the bound table of the local-majorant estimators is read; the light kernels fetch no cells
    constexpr bool LOC = EST != EST_GLOBAL;
#define VP_LL(Q, C, A) hipLaunchKernelGGL((render_k<EST, RNGT, Q, C, false, A, false, 0, true>), g, b, 0, st, S, L)
    (void)ach;
    if (LOC && !quant) { if (count) VP_LL(false, true, true); else VP_LL(false, false, true); }
    else { if (count) VP_LL(true, true, true); else VP_LL(true, false, true); }
#undef VP_LL
}
void launch_render_light(const SceneDev& S, const LaunchDev& L, int est, int rng, bool quant, bool count, int blocks, hipStream_t st)
{
    const ParamDev& P = L.P;
    const bool ach = P.sigma_t[0] == P.sigma_t[1] && P.sigma_t[1] == P.sigma_t[2] && P.albedo[0] == P.albedo[1] && P.albedo[1] == P.albedo[2];
#ifdef VP_DEV_BUILD
    if (rng == RNG_SAMPLERH || est == EST_BOUNDED) { fprintf(stderr, "volpath_hip DEV build: this kernel variant is not compiled\n"); abort(); }
#define VP_LE(RNGT)                                                                             \
    do                                                                                          \
    {                                                                                           \
        if (est == EST_DECOMP) launch_light2<EST_DECOMP, RNGT>(S, L, true, count, ach, blocks, st);  \
        else launch_light2<EST_GLOBAL, RNGT>(S, L, true, count, ach, blocks, st);                    \
    } while (0)
    if (rng == RNG_PHILOX) VP_LE(RngPhilox);
    else VP_LE(RngPhilox7);
#else
#define VP_LE(RNGT)                                                                             \
    do                                                                                          \
    {                                                                                           \
        if (est == EST_DECOMP) launch_light2<EST_DECOMP, RNGT>(S, L, quant, count, ach, blocks, st);        \
        else if (est == EST_BOUNDED) launch_light2<EST_BOUNDED, RNGT>(S, L, quant, count, ach, blocks, st); \
        else launch_light2<EST_GLOBAL, RNGT>(S, L, quant, count, ach, blocks, st);                          \
    } while (0)
    if (rng == RNG_PHILOX) VP_LE(RngPhilox);
    else if (rng == RNG_PHILOX7) VP_LE(RngPhilox7);
    else VP_LE(RngSamplerH);
#endif
#undef VP_LE
}
void launch_bound_bytes(const unsigned char* bounds, size_t nbricks, unsigned* mask, hipStream_t st)
{
    unsigned blocks = (unsigned)std::fmin((double)((nbricks + 255) / 256), 1024.0);
    hipLaunchKernelGGL(bound_bytes_k, dim3(blocks ? blocks : 1), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(bounds), nbricks, mask);
}
void launch_light_identity(const ParamDev& P, bool local, const unsigned* mask, unsigned* flag, hipStream_t st)
{
    hipLaunchKernelGGL(light_identity_k, dim3(1), dim3(256), 0, st, P, local ? 1 : 0, mask, flag);
}
void launch_thr_table(const ParamDev& P, float* table, unsigned count, hipStream_t st)
{
    hipLaunchKernelGGL(thr_table_k, dim3(1), dim3(64), 0, st, P, table, count);
}
void launch_miss_fill(const SceneDev& S, const LaunchDev& L, bool local_estimator, hipStream_t st)
{
    hipLaunchKernelGGL(miss_fill_k, dim3((L.nslots + 255) / 256), dim3(256), 0, st, S, L, local_estimator ? 1 : 0);
}
unsigned segment_table_records(void) { return 2u * VP_SEG_CAP; }
void launch_segment_table(const SceneDev& S, unsigned width, unsigned height, const float4* crawl, const unsigned* pixels, unsigned nslots, float4* seg,
                          hipStream_t st)
{
    hipLaunchKernelGGL(approach_segments_k, dim3((nslots + 255u) / 256u), dim3(256), 0, st, S, width, height, crawl, pixels, nslots, seg);
}
void launch_approach(const SceneDev& S, const LaunchDev& L, int est, int rng, bool quant, hipStream_t st)
{
    const unsigned sh = L.approach_fshift, spb = 256u >> sh;   // pixel slots per workgroup
    const dim3 grid((L.nslots + spb - 1u) / spb, ((unsigned)L.nframes + (1u << sh) - 1u) >> sh);
    if (est == EST_GLOBAL)
    {
        if (rng == RNG_PHILOX7) hipLaunchKernelGGL(approach_k<RngPhilox7>, grid, dim3(256), 0, st, S, L);
        else if (rng == RNG_PHILOX) hipLaunchKernelGGL(approach_k<RngPhilox>, grid, dim3(256), 0, st, S, L);
        else hipLaunchKernelGGL(approach_k<RngSamplerH>, grid, dim3(256), 0, st, S, L);
    }
    else if (quant && L.seg_table && sh == 6u)
    {
        static_assert(VP_SEG_CAP > 64 && VP_SEG_CAP <= 128, "approach_local_tab_k copies a chain with two loads per lane");
        // (uchar bound table, the per-pixel segment table built: the set-up of every restart segment comes from it)
        if (rng == RNG_PHILOX7) hipLaunchKernelGGL(approach_local_tab_k<RngPhilox7>, grid, dim3(256), 0, st, S, L);
        else if (rng == RNG_PHILOX) hipLaunchKernelGGL(approach_local_tab_k<RngPhilox>, grid, dim3(256), 0, st, S, L);
        else hipLaunchKernelGGL(approach_local_tab_k<RngSamplerH>, grid, dim3(256), 0, st, S, L);
    }
    else if (quant)
    {
        if (rng == RNG_PHILOX7) hipLaunchKernelGGL((approach_local_k<RngPhilox7, true>), grid, dim3(256), 0, st, S, L);
        else if (rng == RNG_PHILOX) hipLaunchKernelGGL((approach_local_k<RngPhilox, true>), grid, dim3(256), 0, st, S, L);
        else hipLaunchKernelGGL((approach_local_k<RngSamplerH, true>), grid, dim3(256), 0, st, S, L);
    }
    else
    {
        if (rng == RNG_PHILOX7) hipLaunchKernelGGL((approach_local_k<RngPhilox7, false>), grid, dim3(256), 0, st, S, L);
        else if (rng == RNG_PHILOX) hipLaunchKernelGGL((approach_local_k<RngPhilox, false>), grid, dim3(256), 0, st, S, L);
        else hipLaunchKernelGGL((approach_local_k<RngSamplerH, false>), grid, dim3(256), 0, st, S, L);
    }
}
void launch_pixel_lists(unsigned width, unsigned height, unsigned rank, unsigned world, unsigned ntiles, const unsigned* d_row_start,
                        const float4* table, unsigned* d_block_counts, unsigned* d_totals, unsigned* d_out, hipStream_t st)
{
    PixListDev D;
    D.width = width; D.height = height; D.rank = rank; D.world = world;
    D.tiles_x = (width + 7) / 8; D.tiles_y = (height + 7) / 8; D.ntiles = ntiles; D.row_start = d_row_start; D.table = table;
    const unsigned nblocks = pixel_list_blocks(ntiles);
    if (!nblocks) return;
    hipLaunchKernelGGL(pixlist_count_k, dim3(nblocks), dim3(VP_PIXLIST_BLOCK), 0, st, D, d_block_counts);
    hipLaunchKernelGGL(pixlist_scan_k, dim3(1), dim3(VP_PIXLIST_BLOCK), 0, st, d_block_counts, nblocks, d_totals);
    hipLaunchKernelGGL(pixlist_write_k, dim3(nblocks), dim3(VP_PIXLIST_BLOCK), 0, st, D, (const unsigned*)d_block_counts, (const unsigned*)d_totals, d_out);
}

void launch_render(const SceneDev& S, const LaunchDev& L, int est, int rng, bool quant, bool count, int lds_form, bool mis, int trk,
                   int blocks, hipStream_t st)
{
    const bool lds_bounds = lds_form != 0;
    // lds_form 2: the brick table as 2-bit codes beside the cold state (decomposition, uchar volume, counter-based streams, timed
    // launches only: the host checks all of it, vp_render.cpp)
    if (lds_form == 2 && est == EST_DECOMP && quant && !count && !mis && !trk && !L.cancel && (rng == RNG_PHILOX7 || rng == RNG_PHILOX))
    {
        const ParamDev& Pc = L.P;
        const bool achc = Pc.sigma_t[0] == Pc.sigma_t[1] && Pc.sigma_t[1] == Pc.sigma_t[2] && Pc.albedo[0] == Pc.albedo[1] && Pc.albedo[1] == Pc.albedo[2];
        if (rng == RNG_PHILOX7) launch_render_p7<EST_DECOMP, 2>(S, L, true, false, achc, blocks, st);
        else launch_render3<EST_DECOMP, RngPhilox, 2>(S, L, true, false, achc, false, blocks, st);
        return;
    }
#ifdef VP_DEV_BUILD
    // development build (make DEV=1 -> libvolpath_hip_dev.so): only the kernels of the bench workloads are compiled
    // (Philox streams, uchar volume, spectral tracking, passive environment; global-majorant and decomposition estimators)
    {
        const ParamDev& Pd = L.P;
        const bool achd = Pd.sigma_t[0] == Pd.sigma_t[1] && Pd.sigma_t[1] == Pd.sigma_t[2] && Pd.albedo[0] == Pd.albedo[1] && Pd.albedo[1] == Pd.albedo[2];
        if (trk || mis || !quant || (rng != RNG_PHILOX && rng != RNG_PHILOX7) || est == EST_BOUNDED)
        {
            fprintf(stderr, "volpath_hip DEV build: this kernel variant is not compiled\n");
            abort();
        }
        if (rng == RNG_PHILOX7)
        {
            if (est == EST_DECOMP && lds_bounds) launch_render_p7<EST_DECOMP, 1>(S, L, true, count, achd, blocks, st);
            else if (est == EST_DECOMP) launch_render_p7<EST_DECOMP, 0>(S, L, true, count, achd, blocks, st);
            else launch_render_p7<EST_GLOBAL, 0>(S, L, true, count, achd, blocks, st);
            return;
        }
        if (est == EST_DECOMP)
        {
            if (lds_bounds) launch_render3<EST_DECOMP, RngPhilox, 1>(S, L, true, count, achd, false, blocks, st);
            else launch_render3<EST_DECOMP, RngPhilox, 0>(S, L, true, count, achd, false, blocks, st);
        }
        else launch_render3<EST_GLOBAL, RngPhilox, 0>(S, L, true, count, achd, false, blocks, st);
        return;
    }
#else
    if (trk)
    {
        const bool ph = rng == RNG_PHILOX;
        if (est == EST_DECOMP) { if (ph) launch_render_scalar<EST_DECOMP, RngPhilox>(S, L, quant, trk, blocks, st); else launch_render_scalar<EST_DECOMP, RngSamplerH>(S, L, quant, trk, blocks, st); }
        else if (est == EST_BOUNDED) { if (ph) launch_render_scalar<EST_BOUNDED, RngPhilox>(S, L, quant, trk, blocks, st); else launch_render_scalar<EST_BOUNDED, RngSamplerH>(S, L, quant, trk, blocks, st); }
        else { if (ph) launch_render_scalar<EST_GLOBAL, RngPhilox>(S, L, quant, trk, blocks, st); else launch_render_scalar<EST_GLOBAL, RngSamplerH>(S, L, quant, trk, blocks, st); }
        return;
    }
    // achromatic medium: identical extinction and albedo in the three channels (e.g. preset #13, host.cpp:1308)
    const ParamDev& P = L.P;
    const bool ach = P.sigma_t[0] == P.sigma_t[1] && P.sigma_t[1] == P.sigma_t[2] && P.albedo[0] == P.albedo[1] &&
                     P.albedo[1] == P.albedo[2];
    if (rng == RNG_PHILOX7)
    {
        // (mis and trk were rejected by the API for this generator)
        if (est == EST_DECOMP && lds_bounds && quant) launch_render_p7<EST_DECOMP, 1>(S, L, quant, count, ach, blocks, st);
        else if (est == EST_DECOMP) launch_render_p7<EST_DECOMP, 0>(S, L, quant, count, ach, blocks, st);
        else if (est == EST_BOUNDED) launch_render_p7<EST_BOUNDED, 0>(S, L, quant, count, ach, blocks, st);
        else launch_render_p7<EST_GLOBAL, 0>(S, L, quant, count, ach, blocks, st);
        return;
    }
    if (est == EST_DECOMP)
    {
        if (lds_bounds && quant && !mis)
        {
            if (rng == RNG_PHILOX) launch_render3<EST_DECOMP, RngPhilox, 1>(S, L, quant, count, ach, mis, blocks, st);
            else launch_render3<EST_DECOMP, RngSamplerH, 1>(S, L, quant, count, ach, mis, blocks, st);
        }
        else
        {
            if (rng == RNG_PHILOX) launch_render3<EST_DECOMP, RngPhilox, 0>(S, L, quant, count, ach, mis, blocks, st);
            else launch_render3<EST_DECOMP, RngSamplerH, 0>(S, L, quant, count, ach, mis, blocks, st);
        }
    }
    else if (est == EST_BOUNDED)
    {
        // the dead reference variant: no LDS specialisation, it is there for completeness
        if (rng == RNG_PHILOX) launch_render3<EST_BOUNDED, RngPhilox, 0>(S, L, quant, count, ach, mis, blocks, st);
        else launch_render3<EST_BOUNDED, RngSamplerH, 0>(S, L, quant, count, ach, mis, blocks, st);
    }
    else
    {
        if (rng == RNG_PHILOX) launch_render3<EST_GLOBAL, RngPhilox, 0>(S, L, quant, count, ach, mis, blocks, st);
        else launch_render3<EST_GLOBAL, RngSamplerH, 0>(S, L, quant, count, ach, mis, blocks, st);
    }
#endif
}

// ------------------------------------------------------------------ environment CDF tables (init_envmap, kernel.cu:1144-1210)
// luminance * sin(phi) per texel (PRE_WARP, :1153-1161)
__global__ void env_lum_k(const float4* env, float* lum, int w, int h)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    int    y   = i / w;
    float  phi = kPi * ((float)y + 0.5f) / (float)h;
    float  sp, cp;
    sincosf_(phi, sp, cp);
    float4 t = env[i];
    lum[i]   = luminance(f3{t.x, t.y, t.z}) * sp;
}
// build_cdf_1d (:1036-1055) of one row per thread: sequential float sums, as the reference's host loop
__global__ void env_row_cdf_k(const float* lum, float* cdf_x, float* row_sum, int w, int h)
{
    int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h) return;
    const float* f   = lum + (size_t)y * w;
    float*       cdf = cdf_x + (size_t)y * w;
    float        sum = 0.0f;
    for (int i = 0; i < w; i++) sum += f[i];
    float norm = 1.0f / sum;
    float I    = 0.0f;
    for (int i = 0; i < w; i++) { I += f[i] * norm; cdf[i] = I; }
    cdf[w - 1] = 1.0f;
    row_sum[y] = sum;
}
// the CDF of the row sums and HDRpdfnormAlt (:1163-1166); one thread, the sums are sequential by definition
__global__ void env_col_cdf_k(const float* lum, const float* row_sum, float* cdf_y, float* pdfnorm_alt, int w, int h)
{
    if (blockIdx.x || threadIdx.x) return;
    float lumsum = 0.0f;
    for (size_t i = 0, n = (size_t)w * h; i < n; i++) lumsum += lum[i];
    *pdfnorm_alt = (float)w * (float)h * k1TwoPiPi / lumsum;
    float sum = 0.0f;
    for (int i = 0; i < h; i++) sum += row_sum[i];
    float norm = 1.0f / sum;
    float I    = 0.0f;
    for (int i = 0; i < h; i++) { I += row_sum[i] * norm; cdf_y[i] = I; }
    cdf_y[h - 1] = 1.0f;
}
void launch_env_tables(const float4* env, int w, int h, float* lum, float* row_sum, float* cdf_x, float* cdf_y, float* pdfnorm_alt,
                       hipStream_t st)
{
    int n = w * h;
    hipLaunchKernelGGL(env_lum_k, dim3((n + 255) / 256), dim3(256), 0, st, env, lum, w, h);
    hipLaunchKernelGGL(env_row_cdf_k, dim3((h + 63) / 64), dim3(64), 0, st, lum, cdf_x, row_sum, w, h);
    hipLaunchKernelGGL(env_col_cdf_k, dim3(1), dim3(64), 0, st, lum, row_sum, cdf_y, pdfnorm_alt, w, h);
}
void launch_crawl_table(const SceneDev& S, bool quant, unsigned width, unsigned height, bool control_draw, const unsigned char* danger, float4* table,
                        hipStream_t st)
{
    unsigned n = width * height;
    if (quant) hipLaunchKernelGGL(crawl_table_k<true>, dim3((n + 255) / 256), dim3(256), 0, st, S, width, height, control_draw ? 1 : 0, danger, table);
    else hipLaunchKernelGGL(crawl_table_k<false>, dim3((n + 255) / 256), dim3(256), 0, st, S, width, height, control_draw ? 1 : 0, danger, table);
}
void launch_danger(const SceneDev& S, bool quant, unsigned char* out, unsigned long long* marked, hipStream_t st)
{
    size_t n = (size_t)S.nx * S.ny * S.nz;
    dim3   g((unsigned)((n + 255) / 256));
    if (quant) hipLaunchKernelGGL(danger_k<true>, g, dim3(256), 0, st, S, out, marked);
    else hipLaunchKernelGGL(danger_k<false>, g, dim3(256), 0, st, S, out, marked);
}
void launch_exit_table(const unsigned char* danger, unsigned char* planes, int nx, int ny, int nz, hipStream_t st)
{
    const size_t n = (size_t)nx * ny * nz;
    const int    dims[3] = {nx, ny, nz};
    for (int axis = 0; axis < 3; axis++)
    {
        unsigned char* plane = planes + (size_t)axis * n;
        const int      na = dims[axis], cells = (int)(n / (size_t)na);
        const dim3     g((unsigned)((cells + 255) / 256));
        for (int a = na - 1; a >= 0; a--) hipLaunchKernelGGL(exit_dir_slice_k, g, dim3(256), 0, st, danger, plane, nx, ny, nz, axis, a, 1);
        for (int a = 0; a < na; a++) hipLaunchKernelGGL(exit_dir_slice_k, g, dim3(256), 0, st, danger, plane, nx, ny, nz, axis, a, 0);
    }
}
float sun_clip_step(const SceneDev& S)
{
    const float cx = (S.bmax[0] - S.bmin[0]) / (float)S.nx, cy = (S.bmax[1] - S.bmin[1]) / (float)S.ny, cz = (S.bmax[2] - S.bmin[2]) / (float)S.nz;
    return 0.25f * std::fmin(std::fmin(cx, cy), cz);
}
void launch_sun_clip(const SceneDev& S, const unsigned char* danger, float ds, unsigned short* out, hipStream_t st)
{
    size_t n = (size_t)S.nx * S.ny * S.nz;
    hipLaunchKernelGGL(sun_clip_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, danger, ds, out);
}
void launch_empty_table(const SceneDev& S, unsigned width, unsigned height, const unsigned char* danger, float4* table, hipStream_t st)
{
    unsigned n = width * height;
    hipLaunchKernelGGL(empty_table_k, dim3((n + 255) / 256), dim3(256), 0, st, S, width, height, danger, table);
}
void launch_reduce(const LaunchDev& L, hipStream_t st)
{
    unsigned per_frame = L.nslots;
    hipLaunchKernelGGL(reduce_stage_k, dim3((per_frame + 255) / 256), dim3(256), 0, st, L);
}
void launch_pack_u8(const unsigned char* vol, uint2* cells, int nx, int ny, int nz, bool bricks, hipStream_t st)
{
    size_t n = (size_t)nx * ny * nz;
    hipLaunchKernelGGL(pack_cells_u8_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, vol, cells, nx, ny, nz, bricks ? 1 : 0);
}
void launch_pack_f32(const float* vol, float* cells, int nx, int ny, int nz, bool bricks, hipStream_t st)
{
    size_t n = (size_t)nx * ny * nz;
    hipLaunchKernelGGL(pack_cells_f32_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, vol, cells, nx, ny, nz, bricks ? 1 : 0);
}
void launch_opacity(const SceneDev& S, bool quant, bool lds, const float dir[3], float* out, hipStream_t st)
{
    size_t n = (size_t)S.nx * S.ny * S.nz;
    f3     d = f3{dir[0], dir[1], dir[2]};
    dim3   g((unsigned)((n + 255) / 256));
    if (quant && lds)
    {
        // steps per staged tile: the block may move five cells in a chunk (tile 16 = block 8 + the second tap + margins + 5)
        float per_step = 0.0f;
        const int nn[3] = {S.nx, S.ny, S.nz};
        for (int a = 0; a < 3; a++) per_step = fmaxf(per_step, fabsf(dir[a]) * 0.001f * S.linv[a] * (float)nn[a]);
        int chunk = per_step > 0.0f && per_step == per_step ? (int)fminf(5.0f / per_step, 64.0f) : 1;
        if (chunk < 1) chunk = 1;
        const unsigned nb = (unsigned)((S.nx + VP_OPA_B - 1) / VP_OPA_B) * (unsigned)((S.ny + VP_OPA_B - 1) / VP_OPA_B) * (unsigned)((S.nz + VP_OPA_B - 1) / VP_OPA_B);
        hipLaunchKernelGGL(opacity_lds_k, dim3(nb), dim3(VP_OPA_B * VP_OPA_B * VP_OPA_B), 0, st, S, d, out, chunk);
    }
    else if (quant) hipLaunchKernelGGL(opacity_k<true>, g, dim3(256), 0, st, S, d, out);
    else hipLaunchKernelGGL(opacity_k<false>, g, dim3(256), 0, st, S, d, out);
}
template <class PR>
static void build_bounds_t(const void* d_vol, void* d_out, void* d_tmp_a, void* d_tmp_b, int nx, int ny, int nz, int radius, int brick, hipStream_t st)
{
    typedef typename PR::P P;
    size_t   n = (size_t)nx * ny * nz;
    unsigned g = (unsigned)((n + 255) / 256);
    P *a = (P*)d_tmp_a, *b = (P*)d_tmp_b;
    hipLaunchKernelGGL(bounds_init_k<PR>, dim3(g), dim3(256), 0, st, (const typename PR::T*)d_vol, a, n);
    for (int axis = 0; axis < 3; axis++)
    {
        hipLaunchKernelGGL(bounds_pass_k<PR>, dim3(g), dim3(256), 0, st, (const P*)a, b, nx, ny, nz, axis, radius);
        P* t = a; a = b; b = t;
    }
    int bnx = (nx + brick - 1) / brick, bny = (ny + brick - 1) / brick, bnz = (nz + brick - 1) / brick;
    size_t nb = (size_t)bnx * bny * bnz;
    hipLaunchKernelGGL(bounds_brick_k<PR>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const P*)a, (P*)d_out, nx, ny, nz, brick, bnx, bny, bnz);
}
// d_tmp_a / d_tmp_b: two scratch buffers of nx*ny*nz pairs (2 B uchar, 8 B float)
void launch_build_bounds(const void* d_vol, bool quant, void* d_out, void* d_tmp_a, void* d_tmp_b, int nx, int ny, int nz, int radius, int brick,
                         hipStream_t st)
{
    if (quant) build_bounds_t<PairU8>(d_vol, d_out, d_tmp_a, d_tmp_b, nx, ny, nz, radius, brick, st);
    else build_bounds_t<PairF32>(d_vol, d_out, d_tmp_a, d_tmp_b, nx, ny, nz, radius, brick, st);
}
void launch_julia(unsigned char* grid, int n, hipStream_t st)
{
    size_t total = (size_t)n * n * n;
    hipLaunchKernelGGL(julia_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, grid, n);
}
void launch_cloud(float* grid, int n, unsigned seed, hipStream_t st)
{
    size_t total = (size_t)n * n * n;
    hipLaunchKernelGGL(cloud_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, grid, n, seed);
}
void launch_scale(float4* dst, const float4* src, int size, float s, hipStream_t st)
{
    hipLaunchKernelGGL(scale_k, dim3((size + 255) / 256), dim3(256), 0, st, dst, src, size, s);
}
void launch_gamma(float4* dst, const float4* src, int size, float s, float inv_gamma, hipStream_t st)
{
    hipLaunchKernelGGL(gamma_k, dim3((size + 255) / 256), dim3(256), 0, st, dst, src, size, s, inv_gamma);
}
void launch_accumulate(float4* dst, const float4* src, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(accumulate_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
}
void launch_test_hg(const float* g, const float* r0, const float* r1, const float* nrm, const float* cosq, float* dir, float* ev, int n, hipStream_t st)
{
    hipLaunchKernelGGL(test_hg_k, dim3((n + 255) / 256), dim3(256), 0, st, g, r0, r1, nrm, cosq, dir, ev, n);
}
void launch_test_box(const SceneDev& S, const float* o, const float* d, int* hit, float* tn, float* tf, int n, hipStream_t st)
{
    hipLaunchKernelGGL(test_box_k, dim3((n + 255) / 256), dim3(256), 0, st, S, o, d, hit, tn, tf, n);
}
void launch_test_env(const SceneDev& S, const float* d, float* out, int n, hipStream_t st)
{
    hipLaunchKernelGGL(test_env_k, dim3((n + 255) / 256), dim3(256), 0, st, S, d, out, n);
}
void launch_test_math(int which, const float* in, float* out, int n, hipStream_t st)
{
    hipLaunchKernelGGL(test_math_k, dim3((n + 255) / 256), dim3(256), 0, st, which, in, out, n);
}
void launch_test_rng(int mode, unsigned x, unsigned y, unsigned f, unsigned k0, unsigned k1, int n, float* out, hipStream_t st)
{
    if (mode == RNG_PHILOX) hipLaunchKernelGGL(test_rng_k<RngPhilox>, dim3(1), dim3(64), 0, st, x, y, f, k0, k1, n, out);
    else if (mode == RNG_PHILOX7) hipLaunchKernelGGL(test_rng_k<RngPhilox7>, dim3(1), dim3(64), 0, st, x, y, f, k0, k1, n, out);
    else hipLaunchKernelGGL(test_rng_k<RngSamplerH>, dim3(1), dim3(64), 0, st, x, y, f, k0, k1, n, out);
}
void launch_test_density(const SceneDev& S, bool quant, const float* pos, float* out, int n, hipStream_t st)
{
    if (quant) hipLaunchKernelGGL(test_density_k<true>, dim3((n + 255) / 256), dim3(256), 0, st, S, pos, out, n);
    else hipLaunchKernelGGL(test_density_k<false>, dim3((n + 255) / 256), dim3(256), 0, st, S, pos, out, n);
}
}  // namespace vp
