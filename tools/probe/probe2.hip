// probe2: VALU dependent-issue behaviour on gfx950 — cycles per instruction for a single wave as a function of
// ILP (independent mul+add chains) and waves per SIMD.  Uses s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ILP>
__global__ void chain(float* out, unsigned long long* cyc, float a, float b, int iters) {
    float x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x * 1e-3f + k;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int k = 0; k < ILP; k++) { x[k] = x[k] * a; x[k] = x[k] + b; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int ILP>
void run(float* dout, unsigned long long* dc, int waves_per_simd) {
    // one workgroup per CU-slot: blockDim = 256*waves_per_simd (4 SIMDs), grid = 256 CUs
    const int iters = 2000;
    int threads = 256 * waves_per_simd;
    int grid = 256;
    if (threads > 1024) { grid *= threads / 1024; threads = 1024; }
    hipLaunchKernelGGL(chain<ILP>, dim3(grid), dim3(threads), 0, 0, dout, dc, 0.999f, 1e-3f, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(chain<ILP>, dim3(grid), dim3(threads), 0, 0, dout, dc, 0.999f, 1e-3f, iters);
    CK(hipDeviceSynchronize());
    unsigned long long c; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    double insts = (double)iters * 16 * ILP * 2;
    printf("ILP %d, %d waves/SIMD: %.2f cycles per VALU instr per wave  (SIMD issue interval %.2f cycles)\n", ILP, waves_per_simd, c / insts, c / insts / waves_per_simd);
}

int main() {
    float* dout; unsigned long long* dc;
    CK(hipMalloc(&dout, 1 << 24)); CK(hipMalloc(&dc, 64));
    for (int w : {1, 2, 4, 8}) {
        run<1>(dout, dc, w); run<2>(dout, dc, w); run<3>(dout, dc, w); run<4>(dout, dc, w); run<8>(dout, dc, w);
    }
    return 0;
}
