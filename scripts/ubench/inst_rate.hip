// Instruction issue-rate microbenchmark for gfx950: N independent chains of one instruction per wave,
// 8 waves per SIMD, timed with s_memtime.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void rate_k(unsigned long long* out, float seed, int iters)
{
    float    a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    unsigned u0 = threadIdx.x * 2654435761u + 1u, u1 = u0 + 17u, u2 = u0 + 31u, u3 = u0 + 51u, u4 = u0 + 71u, u5 = u0 + 91u, u6 = u0 + 111u, u7 = u0 + 131u;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    if (OP == 49) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
    if (OP == 36) asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
    {
#define R8F(stmt) { float& x = a0; stmt } { float& x = a1; stmt } { float& x = a2; stmt } { float& x = a3; stmt } { float& x = a4; stmt } { float& x = a5; stmt } { float& x = a6; stmt } { float& x = a7; stmt }
#define R8U(stmt) { unsigned& x = u0; stmt } { unsigned& x = u1; stmt } { unsigned& x = u2; stmt } { unsigned& x = u3; stmt } { unsigned& x = u4; stmt } { unsigned& x = u5; stmt } { unsigned& x = u6; stmt } { unsigned& x = u7; stmt }
        if (OP == 0) { R8F(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));) }
        if (OP == 1) { R8U(asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(x));) }
        if (OP == 2) { R8U(asm volatile("v_mul_u32_u24 %0, %0, %0" : "+v"(x));) }
        if (OP == 3) { R8F(asm volatile("v_floor_f32 %0, %0" : "+v"(x));) }
        if (OP == 4) { R8F(asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));) }
        if (OP == 5) { R8F(asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));) }
        if (OP == 6) { R8F(asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x));) }
        if (OP == 7) { R8F(asm volatile("v_rcp_f32 %0, %0" : "+v"(x));) }
        if (OP == 8) { R8F(asm volatile("v_div_fixup_f32 %0, %0, %0, %0" : "+v"(x));) }
        if (OP == 9) { R8U(asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x));) }
        if (OP == 10) { R8U(asm volatile("v_xor_b32 %0, %0, %0" : "+v"(x));) }
        if (OP == 11) { R8F(asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(x));) }
        if (OP == 12) { R8F(asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 13) { R8F(asm volatile("v_max_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 14) { R8U(asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(x));) }
        if (OP == 15) { R8U(asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x));) }
        if (OP == 16) { R8U(asm volatile("v_max_i32 %0, %0, %0" : "+v"(x));) }
        if (OP == 17) { R8F(asm volatile("v_cmp_lt_f32 vcc, %0, %0" :: "v"(x) : "vcc");) }
        if (OP == 18) { R8F(asm volatile("v_fract_f32 %0, %0" : "+v"(x));) }
        if (OP == 19) { R8F(asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));) }
        if (OP == 20) { R8F(asm volatile("v_log_f32 %0, %0" : "+v"(x));) }
        if (OP == 21) { R8F(asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 22) { R8U(asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));) }
        if (OP == 23) { R8U(asm volatile("v_sub_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(x));) }
        if (OP == 24) { R8F(asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(x));) }
        if (OP == 25) { R8F(asm volatile("v_ldexp_f32 %0, %0, 3" : "+v"(x));) }
        if (OP == 26) { R8U(asm volatile("v_mul_hi_u32 %0, %0, %0" : "+v"(x));) }
        if (OP == 27) { R8F(asm volatile("v_div_scale_f32 %0, vcc, %0, %0, %0" : "+v"(x) :: "vcc");) }
        if (OP == 28) { R8F(asm volatile("v_div_fmas_f32 %0, %0, %0, %0" : "+v"(x) :: "vcc");) }
        if (OP == 29) { R8F(asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 30) { R8F(asm volatile("v_exp_f32 %0, %0" : "+v"(x));) }
        if (OP == 31) { R8F(asm volatile("v_rndne_f32 %0, %0" : "+v"(x));) }
#define R8D(stmt) { double& x = d0; stmt } { double& x = d1; stmt } { double& x = d2; stmt } { double& x = d3; stmt } { double& x = d4; stmt } { double& x = d5; stmt } { double& x = d6; stmt } { double& x = d7; stmt }
        if (OP == 32) { R8D(asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 33) { R8D(asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 34) { R8D(asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(x));) }
        if (OP == 35) { R8D(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x) : "v"(u0) : "vcc");) }
        if (OP == 36) { R8F(asm volatile("v_cndmask_b32_e64 %0, %0, %0, s[10:11]" : "+v"(x) :: "s10", "s11");) }
        if (OP == 37) { R8D(asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(x));) }
        if (OP == 38) { R8F(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %0" :: "v"(x) : "s10", "s11");) }
        if (OP == 39) { R8U(asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));) }
        if (OP == 40) { R8U(asm volatile("v_and_b32 %0, %0, %0" : "+v"(x));) }
        if (OP == 41) { R8F(asm volatile("v_fmaak_f32 %0, %0, %0, 0x3e2aaaab" : "+v"(x));) }
        if (OP == 42) { R8F(asm volatile("v_cvt_f32_ubyte2 %0, %0" : "+v"(x));) }
        if (OP == 43) { R8U(asm volatile("v_bitop3_b32 %0, %0, %0, %0 bitop3:0x96" : "+v"(x));) }
        if (OP == 44) { R8U(asm volatile("v_xad_u32 %0, %0, %0, %0" : "+v"(x));) }
        if (OP == 45) { R8F(asm volatile("v_mov_b32 %0, %0" : "+v"(x));) }
        if (OP == 46) { R8U(asm volatile("v_lshl_or_b32 %0, %0, 3, %0" : "+v"(x));) }
        if (OP == 47) { R8D(asm volatile("v_mul_f64 %0, %0, %0" : "+v"(x));) }
        if (OP == 48) { R8F(asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(x));) }
        if (OP == 49) { R8F(asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(x));) }
        if (OP == 50) { R8F(asm volatile("v_cmp_lt_f32 vcc, %0, %0\n v_cndmask_b32 %0, %0, %0, vcc" : "+v"(x) :: "vcc");) }
        if (OP == 51) { R8F(asm volatile("v_cndmask_b32_e64 %0, %0, %0, vcc" : "+v"(x));) }
        if (OP == 52) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a0)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a1)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a2));
                        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a3)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a4)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a5));
                        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a6)); asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a7)); }
        if (OP == 53) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a0)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a1)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a2));
                        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a3)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a4)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a5));
                        asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a6), "v"(a5) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a7)); }
        if (OP == 54) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a0)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a1)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a2));
                        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a3)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a4)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a5));
                        asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" :: "v"(a6), "v"(a5) : "s10", "s11"); asm volatile("s_nop 1\n v_cndmask_b32_e64 %0, %0, %0, s[10:11]" : "+v"(a7)); }
        if (OP == 55) { R8U(asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) :: "vcc");) }
        if (OP == 56) { R8U(asm volatile("v_bfi_b32 %0, %0, %0, %0" : "+v"(x));) }
        if (OP == 57) { R8U(asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x));) }
        if (OP == 58) { R8F(asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0" : "+v"(x));) }
        if (OP == 59) { R8F(asm volatile("v_mul_f32_e64 %0, %0, %0" : "+v"(x));) }
        if (OP == 60) { R8F(asm volatile("v_add_f32_e64 %0, %0, %0" : "+v"(x));) }
        if (OP == 61) { R8F(asm volatile("v_sub_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 62) { R8F(asm volatile("v_min_f32 %0, %0, %0" : "+v"(x));) }
        if (OP == 63) { R8F(asm volatile("s_nop 0" ::: );) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    a0 += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if (s == 123.456f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int OP>
int run(const char* name, unsigned long long* d, int waves_per_simd)
{
    const int iters = 10000;
    // one CU-filling launch: blocks of 256 threads (4 waves = 1 per SIMD) x waves_per_simd per CU x 256 CUs
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_k<OP>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, 1.5f, iters);  // warm
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_k<OP>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, 1.5f, iters);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CHK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    // memtime ticks at 100 MHz on gfx9 (constant clock); convert with the shader clock estimate below
    double ticks = (double)h[0];
    // wall: the whole launch issues iters*8 instructions per wave, waves_per_simd waves on each SIMD
    printf("%-22s waves/SIMD %d: %8.0f ticks for %d x 8 inst/wave -> %.3f ticks per wave-inst per SIMD | launch %.1f us -> %.3f ns per wave-inst per SIMD\n", name, waves_per_simd, ticks, iters,
           ticks / (iters * 8.0 * waves_per_simd), ms * 1e3, ms * 1e6 / (iters * 8.0 * waves_per_simd));
    return 0;
}
int main()
{
    unsigned long long* d;
    CHK(hipMalloc(&d, 16));
    for (int w : {1, 5, 8})
    {
#define RUN(op, nm) run<op>(nm, d, w)
        RUN(0, "v_fma_f32"); RUN(12, "v_mul_f32"); RUN(21, "v_add_f32"); RUN(29, "v_fmac_f32"); RUN(13, "v_max_f32"); RUN(1, "v_mul_lo_u32"); RUN(26, "v_mul_hi_u32");
        RUN(2, "v_mul_u32_u24"); RUN(14, "v_mad_u32_u24"); RUN(3, "v_floor_f32"); RUN(18, "v_fract_f32"); RUN(31, "v_rndne_f32"); RUN(4, "v_cvt_i32_f32");
        RUN(5, "v_cvt_f32_i32"); RUN(6, "v_cvt_f32_ubyte0"); RUN(7, "v_rcp_f32"); RUN(19, "v_sqrt_f32"); RUN(20, "v_log_f32"); RUN(30, "v_exp_f32");
        RUN(27, "v_div_scale_f32"); RUN(28, "v_div_fmas_f32"); RUN(8, "v_div_fixup_f32"); RUN(9, "v_alignbit_b32"); RUN(10, "v_xor_b32"); RUN(22, "v_lshlrev_b32");
        RUN(11, "v_cndmask_b32"); RUN(15, "v_bfe_u32"); RUN(16, "v_max_i32"); RUN(17, "v_cmp_lt_f32"); RUN(23, "v_sub_u32_sdwa"); RUN(24, "v_frexp_mant_f32");
        RUN(25, "v_ldexp_f32");
        RUN(32, "v_pk_add_f32"); RUN(33, "v_pk_mul_f32"); RUN(34, "v_pk_fma_f32"); RUN(35, "v_mad_u64_u32"); RUN(36, "v_cndmask_e64_sgpr"); RUN(37, "v_lshl_add_u64");
        RUN(38, "v_cmp_lt_f32_e64"); RUN(39, "v_add_u32"); RUN(40, "v_and_b32"); RUN(41, "v_fmaak_f32"); RUN(42, "v_cvt_f32_ubyte2"); RUN(43, "v_bitop3_b32");
        RUN(44, "v_xad_u32"); RUN(45, "v_mov_b32"); RUN(46, "v_lshl_or_b32"); RUN(47, "v_mul_f64"); RUN(48, "v_med3_f32"); RUN(49, "v_cndmask_vcc_set");
        RUN(50, "cmp+cndmask_vcc(x2)"); RUN(51, "v_cndmask_e64_vcc"); RUN(52, "7fma+cndmask_vcc"); RUN(53, "6fma+cmp+cnd_vcc"); RUN(54, "6fma+cmp+cnd_sgpr"); RUN(55, "v_addc_co_u32");
        RUN(56, "v_bfi_b32"); RUN(57, "v_ashrrev_i32"); RUN(58, "v_mul_f32_lit"); RUN(59, "v_mul_f32_e64"); RUN(60, "v_add_f32_e64"); RUN(61, "v_sub_f32"); RUN(62, "v_min_f32"); RUN(63, "s_nop");
    }
    return 0;
}
