// Probe (round 4): where do the waves of a 512-block x 8-wave launch with two workgroups per CU land?  Every wave records
// HW_ID (SIMD, CU, SH, SE) and XCC_ID; the host prints, per CU, which (block, wave) pairs sit on which SIMD and a summary
// of the block pairs that share a CU.   hipcc --offload-arch=gfx950 -O3 -o hwid_map.bin hwid_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(512, 4) void k(unsigned* out, int spin) {
    extern __shared__ float lds[];
    const int wid = threadIdx.x >> 6;
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;             // keep every workgroup resident for a while
    lds[threadIdx.x] = a;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[2 * (blockIdx.x * 8 + wid)] = hw;
        out[2 * (blockIdx.x * 8 + wid) + 1] = (xcc & 15) | (lds[(threadIdx.x + 1) & 511] == -1.f ? 16 : 0);
    }
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* d;
    hipMalloc(&d, blocks * 8 * 2 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    k<<<blocks, 512, 72 * 1024>>>(d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu_blocks;   // CU key -> blocks
    int simd_hist[8][4] = {};
    for (int b = 0; b < blocks; ++b) {
        unsigned key = 0;
        for (int w = 0; w < 8; ++w) {
            const unsigned hw = h[2 * (b * 8 + w)], xcc = h[2 * (b * 8 + w) + 1] & 15;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            simd_hist[w][simd]++;
        }
        cu_blocks[key].push_back(b);
    }
    printf("blocks %d, distinct CUs %zu\n", blocks, cu_blocks.size());
    printf("wave -> SIMD histogram (rows = wave index in the workgroup, columns = SIMD 0..3)\n");
    for (int w = 0; w < 8; ++w) printf("  wave %d: %5d %5d %5d %5d\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
    std::map<int, int> delta_hist;
    int shown = 0;
    for (auto& kv : cu_blocks) {
        if (kv.second.size() == 2) delta_hist[kv.second[1] - kv.second[0]]++;
        else delta_hist[-(int)kv.second.size()]++;
        if (shown < 12) {
            printf("CU xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (int b : kv.second) {
                printf("  block %4d simds", b);
                for (int w = 0; w < 8; ++w) printf(" %u", (h[2 * (b * 8 + w)] >> 4) & 3);
            }
            printf("\n");
            ++shown;
        }
    }
    printf("block-index distance of the two workgroups of a CU (negative key = CUs holding that many workgroups):\n");
    for (auto& kv : delta_hist) printf("  %d: %d CUs\n", kv.first, kv.second);
    // wave 0 of both workgroups on the same SIMD?
    int same = 0, pairs = 0;
    for (auto& kv : cu_blocks)
        if (kv.second.size() == 2) {
            ++pairs;
            same += ((h[2 * (kv.second[0] * 8)] >> 4) & 3) == ((h[2 * (kv.second[1] * 8)] >> 4) & 3);
        }
    printf("CUs with two workgroups: %d, wave 0 of both on the same SIMD: %d\n", pairs, same);
    return 0;
}
