// Probe (round 4): SIMD occupancy of the conversion / packing / mixed-precision VALU encodings the matrix-pipe kernels
// spend their vector time on (inline asm, 8 independent chains per wave), at 1, 2, 3 and 4 waves per SIMD, and of
// v_mfma_f32_16x16x32_f16 alone and beside VALU work.   hipcc --offload-arch=gfx950 -O3 -o valu_enc2.bin valu_enc2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ void k(float* out, int iters) {
    float a[8]; float c = 1.0001f, d = 0.5f;
    unsigned int sel = 0x07060302u;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    h8 A = {1, 2, 3, 4, 5, 6, 7, 8}, B = {1, 1, 1, 1, 1, 1, 1, 1};
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_MIXLOH(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_CVTPKF16(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_CVTPKRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_CVTPKBF16(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_AND(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(d));
#define OP_ANDK(i) asm volatile("v_and_b32_e32 %0, 0xffff0000, %0" : "+v"(a[i]));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(d), "v"(sel));
#define OP_SUB(i) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(d));
#define OP_CVTF32F16(i) asm volatile("v_cvt_f32_f16_e32 %0, %0" : "+v"(a[i]));
#define OP_PKFMAF16(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_MAX3(i) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(d));
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(d));
#define OP_XORK(i) asm volatile("v_xor_b32_e32 %0, 0x70, %0" : "+v"(a[i]));
#define OP_CVTPKF16_SDWA(i) asm volatile("v_cvt_f16_f32_e32 %0, %0" : "+v"(a[i]));
#define OP_LOG(i) asm volatile("v_log_f32_e32 %0, %0" : "+v"(a[i]));
#define OP_MFMA(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[i & 3], 0, 0, 0);
#define OP_MFMA_FMA2(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[i & 3], 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d), "v"(a[(i + 1) & 7]));
#define OP_MFMA_MIX2(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[i & 3], 0, 0, 0); asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixhi_f16 %0, %2, %1, 0" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_MFMA_FMA4(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[i & 3], 0, 0, 0); asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d), "v"(a[(i + 1) & 7]));
#define OP_MFMA_MIX4(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[i & 3], 0, 0, 0); asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixlo_f16 %3, %2, %1, 0\n\tv_fma_mixhi_f16 %0, %2, %1, 0\n\tv_fma_mixhi_f16 %3, %2, %1, 0" : "+v"(a[i]) : "v"(c), "v"(d), "v"(a[(i + 1) & 7]));
            if (MODE == 0) { REP8(OP_FMA) } else if (MODE == 1) { REP8(OP_MIXLO) } else if (MODE == 2) { REP8(OP_MIXHI) } else if (MODE == 3) { REP8(OP_MIXLOH) }
            else if (MODE == 4) { REP8(OP_CVTPKF16) } else if (MODE == 5) { REP8(OP_CVTPKRTZ) } else if (MODE == 6) { REP8(OP_CVTPKBF16) } else if (MODE == 7) { REP8(OP_AND) }
            else if (MODE == 8) { REP8(OP_ANDK) } else if (MODE == 9) { REP8(OP_PERM) } else if (MODE == 10) { REP8(OP_SUB) } else if (MODE == 11) { REP8(OP_CVTF32F16) }
            else if (MODE == 12) { REP8(OP_PKFMAF16) } else if (MODE == 13) { REP8(OP_MAX3) } else if (MODE == 14) { REP8(OP_ANDOR) } else if (MODE == 15) { REP8(OP_BFI) }
            else if (MODE == 16) { REP8(OP_LSHLOR) } else if (MODE == 17) { REP8(OP_ALIGNBIT) } else if (MODE == 18) { REP8(OP_XORK) } else if (MODE == 19) { REP8(OP_CVTPKF16_SDWA) }
            else if (MODE == 20) { REP8(OP_LOG) } else if (MODE == 21) { REP8(OP_MFMA) } else if (MODE == 22) { REP8(OP_MFMA_FMA2) } else if (MODE == 23) { REP8(OP_MFMA_MIX2) }
            else if (MODE == 24) { REP8(OP_MFMA_FMA4) } else { REP8(OP_MFMA_MIX4) }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int per) {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 64 * 4 * wps;   // one block per CU, wps waves per SIMD
        k<MODE><<<256, threads>>>(d, 10);
        hipEventRecord(e0);
        k<MODE><<<256, threads>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d w/SIMD %6.2f ns", wps, ms * 1e6 / ((double)iters * 64 * wps));
    }
    printf("   per group of %d instr\n", per);
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 1); run<1>("v_fma_mixlo_f16 (f32 srcs)", 1); run<2>("v_fma_mixhi_f16", 1); run<3>("v_fma_mixlo_f16 (f16 src)", 1);
    run<4>("v_cvt_pk_f16_f32", 1); run<5>("v_cvt_pkrtz_f16_f32", 1); run<6>("v_cvt_pk_bf16_f32", 1); run<7>("v_and_b32", 1);
    run<8>("v_and_b32 literal", 1); run<9>("v_perm_b32", 1); run<10>("v_sub_f32", 1); run<11>("v_cvt_f32_f16", 1);
    run<12>("v_pk_fma_f16", 1); run<13>("v_max3_f32 |.|", 1); run<14>("v_and_or_b32", 1); run<15>("v_bfi_b32", 1);
    run<16>("v_lshl_or_b32", 1); run<17>("v_alignbit_b32", 1); run<18>("v_xor_b32 literal", 1); run<19>("v_cvt_f16_f32", 1);
    run<20>("v_log_f32", 1); run<21>("mfma_16x16x32_f16", 1); run<22>("mfma + 2 v_fma_f32", 3); run<23>("mfma + 2 v_fma_mix", 3);
    run<24>("mfma + 4 v_fma_f32", 5); run<25>("mfma + 4 v_fma_mix", 5);
    return 0;
}
