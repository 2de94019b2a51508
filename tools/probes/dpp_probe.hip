// Probe: which whole-wave DPP controls does gfx950 implement?  (assembler accepts all of them)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __global__ void k(int* out) {
    int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, 0xF, 0xF, false);
}
template <int CTRL> void run(const char* name) {
    int* d; int h[64];
    hipMalloc(&d, 256);
    k<CTRL><<<1, 64>>>(d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-14s:", name);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}
int main() {
    run<0x138>("wave_shr:1");
    run<0x130>("wave_shl:1");
    run<0x13C>("wave_ror:1");
    run<0x111>("row_shr:1");
    run<0x142>("row_bcast:15");
    run<0x143>("row_bcast:31");
    run<0x141>("row_half_mirror");
    return 0;
}
