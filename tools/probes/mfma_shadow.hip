// Probe (round 4): how much vector work runs in the shadow of an MFMA on one SIMD -- v_mfma_f32_16x16x32_f16 (16 cycles of
// matrix pipe) against v_mfma_f32_32x32x16_f16 (32), each followed by N independent v_fma_f32 -- plus the lane-swap
// instructions.   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow.bin mfma_shadow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int SHAPE, int NV, int KIND>
__global__ void k(float* out, int iters) {
    float a[8]; float c = 1.0001f, d = 0.5f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    h8 A = {1, 2, 3, 4, 5, 6, 7, 8}, B = {1, 1, 1, 1, 1, 1, 1, 1};
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16v big[2];
    for (int i = 0; i < 16; ++i) { big[0][i] = 0; big[1][i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SHAPE == 16) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[u & 3], 0, 0, 0);
            if (SHAPE == 32) big[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, big[u & 1], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(u + v) & 7]) : "v"(c), "v"(d));
                if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[(u + v) & 7]) : "v"(c));
                if (KIND == 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[(u + v) & 7]), "+v"(a[(u + v + 4) & 7]));
                if (KIND == 3) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[(u + v) & 7]), "+v"(a[(u + v + 4) & 7]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += big[0][i] + big[1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int NV, int KIND> void run(const char* name) {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-34s", name);
    for (int wps = 1; wps <= 3; ++wps) {
        const int threads = 64 * 4 * wps;
        k<SHAPE, NV, KIND><<<256, threads>>>(d, 10);
        hipEventRecord(e0);
        k<SHAPE, NV, KIND><<<256, threads>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d w/SIMD %6.2f ns", wps, ms * 1e6 / ((double)iters * 8 * wps));
    }
    printf("  per group\n");
    hipFree(d);
}
int main() {
    run<0, 4, 0>("4 v_fma_f32"); run<0, 4, 1>("4 v_cvt_pk_f16_f32"); run<0, 1, 2>("1 v_permlane16_swap"); run<0, 1, 3>("1 v_permlane32_swap");
    run<16, 0, 0>("mfma16x16x32"); run<16, 2, 0>("mfma16x16x32 + 2 fma"); run<16, 4, 0>("mfma16x16x32 + 4 fma"); run<16, 8, 0>("mfma16x16x32 + 8 fma");
    run<32, 0, 0>("mfma32x32x16"); run<32, 4, 0>("mfma32x32x16 + 4 fma"); run<32, 8, 0>("mfma32x32x16 + 8 fma"); run<32, 12, 0>("mfma32x32x16 + 12 fma");
    run<32, 16, 0>("mfma32x32x16 + 16 fma"); run<32, 4, 1>("mfma32x32x16 + 4 cvt_pk"); run<32, 8, 1>("mfma32x32x16 + 8 cvt_pk"); run<16, 4, 1>("mfma16x16x32 + 4 cvt_pk");
    return 0;
}
