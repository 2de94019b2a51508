// Probe: per-instruction SIMD occupancy of exact VALU encodings on gfx950 (inline asm, 8 independent
// chains per wave, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8]; v2 p[8]; float c = 1.0001f, d = 0.5f; v2 pc = {c, c};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; p[i] = v2{a[i], a[i] + 1}; }
    unsigned long long msk = 0x5555555555555555ull;
    asm volatile("s_mov_b64 vcc, %0" :: "s"(msk) : "vcc");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define OP_ADD32(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(d));
#define OP_ADD64(i) asm volatile("v_add_f32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(d));
#define OP_MUL32(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define OP_FMAC32(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_FMA64(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
#define OP_FMAAK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[i]) : "v"(c));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
#define OP_MOV(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(d));
#define OP_CND(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(d) : "vcc");
#define OP_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(d), "s"(msk));
#define OP_CMP(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a[i]), "v"(d) : "vcc");
#define OP_XOR(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(d));
#define OP_LSHL(i) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
#define OP_MAD(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(d));
#define OP_LOG(i) asm volatile("v_log_f32_e32 %0, %0" : "+v"(a[i]));
#define OP_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 0) { REP8(OP_ADD32) } else if (MODE == 1) { REP8(OP_ADD64) } else if (MODE == 2) { REP8(OP_MUL32) }
            else if (MODE == 3) { REP8(OP_FMAC32) } else if (MODE == 4) { REP8(OP_FMA64) } else if (MODE == 5) { REP8(OP_FMAAK) }
            else if (MODE == 6) { REP8(OP_PKADD) } else if (MODE == 7) { REP8(OP_PKMUL) } else if (MODE == 8) { REP8(OP_PKFMA) }
            else if (MODE == 9) { REP8(OP_MOV) } else if (MODE == 10) { REP8(OP_CND) } else if (MODE == 11) { REP8(OP_DPP) } else if (MODE == 12) { REP8(OP_CND64) }
            else if (MODE == 13) { REP8(OP_CMP) } else if (MODE == 14) { REP8(OP_XOR) } else if (MODE == 15) { REP8(OP_LSHL) } else if (MODE == 16) { REP8(OP_MAD) } else { REP8(OP_LOG) }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4);
    const int iters = 4000, wps = 4, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s %.2f ns per wave-instr per SIMD\n", name, ms * 1e6 / ((double)iters * 64 * wps));
    hipFree(d);
}
int main() {
    run<0>("v_add_f32_e32"); run<1>("v_add_f32_e64"); run<2>("v_mul_f32_e32"); run<3>("v_fmac_f32_e32"); run<4>("v_fma_f32(e64)");
    run<5>("v_fmaak_f32"); run<6>("v_pk_add_f32"); run<7>("v_pk_mul_f32"); run<8>("v_pk_fma_f32"); run<9>("v_mov_b32");
    run<10>("v_cndmask_e32 vcc"); run<11>("v_add_f32_dpp"); run<12>("v_cndmask_e64 sgpr"); run<13>("v_cmp_lt_f32 vcc"); run<14>("v_xor_b32"); run<15>("v_lshlrev_b32"); run<16>("v_mad_u32_u24"); run<17>("v_log_f32");
    return 0;
}
