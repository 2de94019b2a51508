// Probe: dependent-chain behaviour of scalar vs packed fp32 adds at several chain counts (ILP) and
// waves per SIMD.  CH independent chains per wave; each op depends on the previous op of its chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int MODE, int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[CH]; v2 p[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { a[i] = threadIdx.x + i; p[i] = v2{a[i], a[i] + 1}; }
    const float d = 0.5f; const v2 pd = {d, -d};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (MODE == 0) a[i] = a[i] + d;                                   // v_add_f32
                else if (MODE == 1) p[i] = p[i] + pd;                             // v_pk_add_f32
                else p[i] = __builtin_shufflevector(p[i], -p[i], 1, 2) + pd;       // v_pk_add with swap+neg
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int CH> void run(const char* name, int wps) {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4);
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, CH><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    k<MODE, CH><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s chains=%d waves/SIMD=%d : %.2f ns per wave-instr per SIMD\n", name, CH, wps, ms * 1e6 / ((double)iters * 64 * wps));
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0, 1>("add", w); run<1, 1>("pk_add", w); run<2, 1>("pk_add_sw", w);
        run<0, 2>("add", w); run<1, 2>("pk_add", w); run<2, 2>("pk_add_sw", w);
        run<0, 4>("add", w); run<1, 4>("pk_add", w); run<2, 4>("pk_add_sw", w);
    }
    return 0;
}
