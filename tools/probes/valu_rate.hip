// Probe: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 on gfx950 at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f, d = 0.5f;
    const v2 pc = {c, c}, pd = {d, d};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = fmaf(a0, c, d); a1 = fmaf(a1, c, d); a2 = fmaf(a2, c, d); a3 = fmaf(a3, c, d);
                a4 = fmaf(a4, c, d); a5 = fmaf(a5, c, d); a6 = fmaf(a6, c, d); a7 = fmaf(a7, c, d);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0 = __builtin_elementwise_fma(p0, pc, pd); p1 = __builtin_elementwise_fma(p1, pc, pd);
                p2 = __builtin_elementwise_fma(p2, pc, pd); p3 = __builtin_elementwise_fma(p3, pc, pd);
                p4 = __builtin_elementwise_fma(p4, pc, pd); p5 = __builtin_elementwise_fma(p5, pc, pd);
                p6 = __builtin_elementwise_fma(p6, pc, pd); p7 = __builtin_elementwise_fma(p7, pc, pd);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0 += pd; p1 += pd; p2 += pd; p3 += pd; p4 += pd; p5 += pd; p6 += pd; p7 += pd;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y +
                                          p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}
template <int MODE> void run(const char* name, int blocks_per_cu) {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4);
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64 * blocks_per_cu;  // each wave: 64 instr/iter; waves/SIMD = blocks_per_cu
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, blocks_per_cu, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_add_f32", w); }
    return 0;
}
