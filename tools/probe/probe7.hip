// probe7: where does the CU put the four waves of a 256-thread workgroup when two such workgroups share a CU (the Q28 chain
// kernel's shape: 57.6 KB of LDS each)?  Records HW_ID of every wave; prints, per CU, the (SIMD, wave slot) of waves 0..3 of the
// workgroups that were resident together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256, 2) void where(uint32_t *out, int spin) {
    extern __shared__ uint32_t lds[];
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID
    lds[threadIdx.x] = hw;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = 0; i < spin; i++) acc += lds[(threadIdx.x + i) & 255] * 3u + i;   // stay resident for a while
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc | (acc & 0x80000000u); }
}

int main() {
    const int n_wg = 1024;
    uint32_t *d; CK(hipMalloc(&d, n_wg * 4 * 2 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&where), hipFuncAttributeMaxDynamicSharedMemorySize, 58000));
    hipLaunchKernelGGL(where, dim3(n_wg), dim3(256), 58000, 0, d, 20000);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(n_wg * 8);
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    int same_simd_order = 0, distinct = 0, parity_all_equal = 0;
    std::map<uint32_t, std::vector<int>> by_cu;
    for (int w = 0; w < n_wg; w++) {
        uint32_t simd[4], slot[4];
        for (int k = 0; k < 4; k++) { uint32_t hw = h[(w * 4 + k) * 2]; simd[k] = (hw >> 4) & 3; slot[k] = hw & 15; }
        bool ord = true, dis = true, par = true;
        for (int k = 0; k < 4; k++) { if (simd[k] != (uint32_t)k) ord = false; for (int j = 0; j < k; j++) if (simd[j] == simd[k]) dis = false; if ((slot[k] & 1) != (slot[0] & 1)) par = false; }
        same_simd_order += ord; distinct += dis; parity_all_equal += par;
        const uint32_t hw0 = h[w * 8], xcc = h[w * 8 + 1] & 15;
        by_cu[(xcc << 16) | (hw0 & 0xff00)].push_back(w);
        if (w < 12 || (w >= 512 && w < 520)) {
            printf("wg %4d xcc %u cu %2u se %u: ", w, xcc, (hw0 >> 8) & 15, (hw0 >> 13) & 7);
            for (int k = 0; k < 4; k++) printf(" w%d simd %u slot %u |", k, simd[k], slot[k]);
            printf("\n");
        }
    }
    printf("%d workgroups: wave w on SIMD w in %d, four distinct SIMDs in %d, all four slot parities equal in %d; %zu distinct (xcc, se, sh, cu)\n",
           n_wg, same_simd_order, distinct, parity_all_equal, by_cu.size());
    int shown = 0;
    for (auto &kv : by_cu) { if (shown++ >= 6) break; printf("cu key %06x: wgs", kv.first); for (int w : kv.second) printf(" %d(slot %u)", w, h[w * 8] & 15); printf("\n"); }
    return 0;
}
