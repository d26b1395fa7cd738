// probe3: AGGREGATE VALU issue rate per SIMD vs waves/SIMD and per-wave ILP (dependent mul+add chains), by wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ILP>
__global__ void chain(float* out, float a, float b, int iters) {
    float x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x * 1e-3f + k;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
#pragma unroll
            for (int k = 0; k < ILP; k++) { x[k] = x[k] * a; x[k] = x[k] + b; }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
void run(float* dout, int wps) {
    const int iters = 4000 / ILP;
    // blocks of 256 threads = 1 wave per SIMD; `wps` blocks per CU -> wps waves per SIMD (256 CUs)
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain<ILP>, grid, block, 0, 0, dout, 0.999f, 1e-3f, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain<ILP>, grid, block, 0, 0, dout, 0.999f, 1e-3f, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double inst_per_simd = (double)iters * 32 * ILP * 2 * wps;
    double ns_per_inst = ms * 1e6 / inst_per_simd;
    printf("waves/SIMD %d ILP %d: %.3f ms, %.3f ns per VALU instr per SIMD (= %.2f cycles @2.4GHz), %.1f T lane-ops/s\n", wps, ILP, ms, ns_per_inst,
           ns_per_inst * 2.4, 64.0 * 1024 / ns_per_inst / 1e3);
}

int main() {
    float* dout; CK(hipMalloc(&dout, 1 << 24));
    for (int w : {1, 2, 3, 4, 6, 8}) { run<1>(dout, w); run<2>(dout, w); run<4>(dout, w); }
    return 0;
}
