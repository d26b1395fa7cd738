// probe4: aggregate PACKED f32 VALU rate per SIMD (v_pk_mul_f32 / v_pk_add_f32 with an SGPR-pair coefficient) vs waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int ILP>
__global__ void chain(float* out, float a, float b, int iters) {
    f2 x[ILP];
    const f2 av = {a, a}, bv = {b, b};
#pragma unroll
    for (int k = 0; k < ILP; k++) { x[k].x = threadIdx.x * 1e-3f + k; x[k].y = x[k].x + 0.25f; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
#pragma unroll
            for (int k = 0; k < ILP; k++) { x[k] = x[k] * av; x[k] = x[k] + bv; }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += x[k].x + x[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
void run(float* dout, int wps) {
    const int iters = 4000 / ILP;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain<ILP>, grid, block, 0, 0, dout, 0.999f, 1e-3f, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain<ILP>, grid, block, 0, 0, dout, 0.999f, 1e-3f, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double inst_per_simd = (double)iters * 32 * ILP * 2 * wps;
    double ns = ms * 1e6 / inst_per_simd;
    printf("PACKED waves/SIMD %d ILP %d: %.3f ns per v_pk instr per SIMD (= %.2f cycles @2.4GHz), %.1f T element-ops/s\n", wps, ILP, ns, ns * 2.4, 128.0 * 1024 / ns / 1e3);
}

int main() {
    float* dout; CK(hipMalloc(&dout, 1 << 24));
    for (int w : {1, 2, 3, 4, 8}) { run<1>(dout, w); run<2>(dout, w); run<4>(dout, w); }
    return 0;
}
