// probe10: lane semantics of the gfx950 cross-lane moves a register <-> lane transpose is built from (tools/experiments, round 5):
// v_permlane32_swap / v_permlane16_swap (which halves / rows trade places), ds_swizzle xor 4, DPP row_ror:8.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o) {
    const unsigned l = threadIdx.x;
    unsigned a = l, b = 100 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1]; o[128 + l] = q[0]; o[192 + l] = q[1];
    o[256 + l] = __builtin_amdgcn_ds_swizzle(a, 0x101F);
    o[320 + l] = __builtin_amdgcn_update_dpp(0u, a, 0x128, 0xf, 0xf, false);
}
int main() {
    unsigned *d, h[384];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *n[6] = {"permlane32_swap[0] (a)", "permlane32_swap[1] (b)", "permlane16_swap[0] (a)", "permlane16_swap[1] (b)", "ds_swizzle xor 4", "dpp row_ror:8"};
    for (int t = 0; t < 6; t++) { printf("%-24s", n[t]); for (int l = 0; l < 64; l += (t < 4 ? 8 : 1)) printf(" %u", h[t * 64 + l]); printf("\n"); }
    return 0;
}
