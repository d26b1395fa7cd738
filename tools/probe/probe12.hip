// probe12 — the arithmetic core of the output side DESIGN.md section 9.1 costed ("two waves per SIMD, one and a half outputs per wave, two band
// recurrences interleaved in one loop") against today's (three waves per SIMD, one output per wave), in isolation: the GENERATED band loops of
// the firmware contract (dspi_bandloops_pk_fma.inc) on register data, nothing else in the kernel.  Per SIMD and step the work is the same in
// every shape — THREE outputs' visits of one band (16 frames x 128 streams each):
//   A  3 waves/SIMD, each band16pkf_any                                  (today's output SIMDs)
//   B  2 waves/SIMD: one runs band16pk2f_any (two outputs interleaved), the other band16pkf_any (one output)     (section 9.1, fixed roles)
//   C  2 waves/SIMD, both band16pk2f_any, three visits per two steps each (one and a half outputs per wave on average)
//   D  1 wave/SIMD doing all three (pk2 + pk): what a lone wave makes of the interleave
// Reports ns per (SIMD, three band visits) and the same in cycles at the clock the run reports nothing about (compare the shapes with each other,
// not with a clock).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o probe12 probe12.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
#include "../../dspi_amd/csrc/dspi_bandloops_pk_fma.inc"

__device__ __forceinline__ void init(v2f (&x)[16], float seed) { for (int i = 0; i < 16; i++) { x[i].x = threadIdx.x * 1e-3f + i + seed; x[i].y = x[i].x * 0.5f; } }
__device__ __forceinline__ float fold(v2f (&x)[16]) { float s = 0.f; for (int i = 0; i < 16; i++) s += x[i].x + x[i].y; return s; }

// MODE 0: pk only (1 output per visit); 1: pk2 only (2 outputs per visit); 2: the wave's half of the block picks (upper half pk2, lower pk);
// 3: pk2 then pk in every repetition (3 outputs per visit, one wave).  One kernel per mode and block size: each holds only the sample arrays
// it uses and is compiled for the registers its waves-per-SIMD leave (a common kernel bounded for 768 threads spilled in the two-wave shapes).
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void bands(float *out, uint32_t ka, uint32_t kb, v2f c01, v2f c23, v2f c45, int reps) {
    v2f xa[16], xb[16], xc[16];
    init(xa, 0.f);
    if (MODE != 0) init(xb, 3.f);
    if (MODE == 3) init(xc, 7.f);
    v2f s1a = {0.f, 0.f}, s2a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2b = {0.f, 0.f}, s1c = {0.f, 0.f}, s2c = {0.f, 0.f};
    const bool upper = (threadIdx.x >> 6) >= (THREADS >> 7);
    for (int r = 0; r < reps; r++) {
        uint32_t k1 = ka, k2 = kb; v2f a = c01, b = c23, d = c45;
        asm volatile("" : "+s"(k1), "+s"(k2), "+s"(a), "+s"(b), "+s"(d));
        if (MODE == 0) band16pkf_any(xa, s1a, s2a, k1, a, b, d);
        else if (MODE == 1) band16pk2f_any(xa, xb, s1a, s2a, s1b, s2b, k1, k2, a, b, d, a, b, d);
        else if (MODE == 2) { if (upper) band16pk2f_any(xa, xb, s1a, s2a, s1b, s2b, k1, k2, a, b, d, a, b, d); else band16pkf_any(xa, s1a, s2a, k1, a, b, d); }
        else { band16pk2f_any(xa, xb, s1a, s2a, s1b, s2b, k1, k2, a, b, d, a, b, d); band16pkf_any(xc, s1c, s2c, k1, a, b, d); }
    }
    float s = s1a.x + s2a.y + fold(xa);
    if (MODE != 0) s += s1b.x + s2b.y + fold(xb);
    if (MODE == 3) s += s1c.x + s2c.y + fold(xc);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int THREADS>
static float run(float *dout, uint32_t ka, uint32_t kb, int reps) {
    dim3 grid(256), block(THREADS);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const v2f c01{0.01f, 0.005f}, c23{0.0025f, 0.00125f}, c45{0.003f, 0.007f};
    hipLaunchKernelGGL((bands<MODE, THREADS>), grid, block, 0, 0, dout, ka, kb, c01, c23, c45, reps);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((bands<MODE, THREADS>), grid, block, 0, 0, dout, ka, kb, c01, c23, c45, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float *dout; CK(hipMalloc(&dout, 1 << 24));
    const char *names[6] = {"", "biquad", "svf-lp", "svf-hp", "svf-pk", "svf-shelf"};
    const int reps = 6000;      // a multiple of 6: every shape does reps / (its visits per repetition) x ... = the same 3 x reps_unit visits per SIMD
    printf("| band forms (a / b) | A: 3 waves x pk | B: pk2 + pk | C: 2 waves x pk2 (x 3/4 reps) | D: 1 wave pk2+pk | B / A | C / A |\n|---|---|---|---|---|---|---|\n");
    const uint32_t combos[][2] = {{4, 4}, {1, 1}, {5, 5}, {2, 2}, {4, 1}, {4, 5}, {1, 5}};
    for (auto &c : combos) {
        // per SIMD: A = 3 waves x reps visits = 3 reps output-visits; B = (2 + 1) x reps = 3 reps; C = 2 waves x 2 outputs x (3/4 reps) = 3 reps; D = 3 x reps
        const float a = run<0, 768>(dout, c[0], c[0], reps);
        const float b = run<2, 512>(dout, c[0], c[1], reps);
        const float cc = run<1, 512>(dout, c[0], c[1], reps * 3 / 4);
        const float d = run<3, 256>(dout, c[0], c[1], reps);
        const double per = 1e6 / reps;      // ns per (SIMD, 3 output-visits of one band)
        printf("| %s / %s | %.1f ns | %.1f ns | %.1f ns | %.1f ns | %.3f | %.3f |\n", names[c[0]], names[c[1]], a * per, b * per, cc * per, d * per, b / a, cc / a);
    }
    return 0;
}
