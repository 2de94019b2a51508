// Probe: v_mfma_f32_16x16x4_f32 on gfx950 -- (1) operand / result register layout, checked against a
// host product; (2) issue rate alone and beside fp32 VALU work in the same wave and in sibling waves.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_probe.bin mfma_f32_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* A, const float* B, float* D) {
    // A: [16][4] row major, B: [4][16] row major, D: [16][16]
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16];   // lane holds A[i = l % 16][k = l / 16]
    const float b = B[(l / 16) * 16 + l % 16];  // lane holds B[k = l / 16][j = l % 16]
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + l % 16] = c[v];  // D[i = 4 (l/16) + v][j = l % 16]
}

// MODE 0: NV VALU fmas per MFMA in the same wave.  MODE 1: even waves MFMA only, odd waves VALU only.
template <int NV, int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a = 1.0f + l * 1e-3f, b = 0.5f - l * 1e-3f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = l + i;
    const float m = 1.0001f, d = 0.25f;
    const bool do_mfma = MODE == 0 || (w & 1) == 0, do_valu = MODE == 0 || (w & 1) == 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (do_mfma) {
                if (u == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                if (u == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                if (u == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                if (u == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
            if (do_valu) {
#pragma unroll
                for (int i = 0; i < (MODE == 0 ? NV : 12); ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 12]) : "v"(m), "v"(d));
            }
        }
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV, int MODE>
static void run_rate(const char* name, int blocks_per_cu) {
    float* d;
    hipMalloc(&d, 256 * 4 * 512 * 4);
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<NV, MODE><<<blocks, 512>>>(d, 10);
    hipEventRecord(e0);
    rate_kernel<NV, MODE><<<blocks, 512>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = 2 * blocks_per_cu; each wave issues 4 * iters MFMAs (MODE 0)
    const double wps = 2.0 * blocks_per_cu;
    const double mf = MODE == 0 ? 4.0 * iters * wps : 4.0 * iters * wps / 2;
    printf("%-44s waves/SIMD %.0f: %7.3f ms  -> %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz)\n", name, wps, ms,
           ms * 1e6 / mf, ms * 1e6 / mf * 2.4);
    hipFree(d);
}

int main() {
    std::vector<float> A(64), B(64), D(256), R(256, 0.f);
    for (int i = 0; i < 64; ++i) { A[i] = (float)(rand() % 17 - 8); B[i] = (float)(rand() % 13 - 6); }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("layout check (A[i=l%%16][k=l/16], B[k=l/16][j=l%%16], D[i=4(l/16)+v][j=l%%16]): %s (%d mismatches)\n",
           bad ? "WRONG" : "OK", bad);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run_rate<0, 0>("mfma only", bpc);
        run_rate<2, 0>("mfma + 2 v_fma per mfma (same wave)", bpc);
        run_rate<4, 0>("mfma + 4 v_fma per mfma (same wave)", bpc);
        run_rate<6, 0>("mfma + 6 v_fma per mfma (same wave)", bpc);
        run_rate<8, 0>("mfma + 8 v_fma per mfma (same wave)", bpc);
        run_rate<12, 0>("mfma + 12 v_fma per mfma (same wave)", bpc);
        run_rate<0, 1>("even waves mfma, odd waves 12 v_fma per slot", bpc);
    }
    return bad ? 1 : 0;
}
