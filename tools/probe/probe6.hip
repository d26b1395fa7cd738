// probe6 (probe5 parametrised: -DBL_INC=... -DBL_FN=... -DBL_COUNTS={...}): how fast do the generated packed band loops (dspi_bandloops_pk.inc) actually issue?
// One wave (or several) per SIMD runs band16pk_any on register data; reports cycles per v_pk instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef BL_INC
#define BL_INC "../../dspi_amd/csrc/dspi_bandloops_pk.inc"
#endif
#include BL_INC

__global__ void bands(float* out, uint32_t kind, v2f c01, v2f c23, v2f c45, int reps) {
    v2f x[16];
    for (int i = 0; i < 16; i++) { x[i].x = threadIdx.x * 1e-3f + i; x[i].y = x[i].x * 0.5f; }
    v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
    for (int r = 0; r < reps; r++) {
        uint32_t k = kind; v2f a = c01, b = c23, d = c45;
        asm volatile("" : "+s"(k), "+s"(a), "+s"(b), "+s"(d));
        BL_FN(x, s1, s2, k, a, b, d);
    }
    float s = s1.x + s2.y;
    for (int i = 0; i < 16; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* dout; CK(hipMalloc(&dout, 1 << 24));
    const int per_sample[6] = BL_COUNTS;
    const char* names[6] = {"", "biquad", "svf-lp", "svf-hp", "svf-pk", "svf-shelf"};
    const int reps = 2000;
    for (int wps : {1, 2, 3})
        for (uint32_t kind = 1; kind <= 5; kind++) {
            dim3 grid(256 * wps), block(256);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(bands, grid, block, 0, 0, dout, kind, v2f{0.01f, 0.005f}, v2f{0.0025f, 0.00125f}, v2f{0.003f, 0.007f}, reps);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(bands, grid, block, 0, 0, dout, kind, v2f{0.01f, 0.005f}, v2f{0.0025f, 0.00125f}, v2f{0.003f, 0.007f}, reps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double inst = (double)reps * 16 * per_sample[kind] * wps;
            double ns = ms * 1e6 / inst;
            printf("waves/SIMD %d %-9s: %.2f cycles per v_pk instr per SIMD @2.4GHz (%.0f cycles per band visit per wave)\n", wps, names[kind], ns * 2.4,
                   ms * 1e6 * 2.4 / reps);
        }
    return 0;
}
