// probe8 — how fast can ONE stream's 10-band cascade go?  (config 2 is bound by the recurrence: 96 000 frames x 20 bands in sequence
// per stream)  Two kernels on the same arithmetic (TDF2 biquads, un-fused mul/add), bit-compared:
//   A: one lane per chain (stream x channel), the ten bands one after the other per frame — the shape of today's kernels;
//   B: one lane per (chain, band): band b of frame n runs while band b+1 runs frame n-1; the sample moves one lane up per step
//      through a DPP row shift (ten lanes of a 16-lane row per chain, four chains per wave).
// Third argument 1: bands 0-6 are SVF peaking sections, 7-9 biquads (config 2 at 48 kHz: bands under Fs/7.5 take the SVF form);
// kernel B then computes both forms in every lane and selects (the forms sit in different lanes of one wave).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o probe8 probe8.hip ; run: ./probe8 [chains] [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NB = 10;

__device__ __forceinline__ float input_sample(uint32_t chain, uint32_t n) {      // cheap deterministic input, same in both kernels
    uint32_t h = (chain * 2654435761u) ^ (n * 40503u + 12345u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (float)(int32_t)(h & 0xffffu) * (1.0f / 65536.0f) - 0.5f;
}

// one band-sample; svf: peaking section (v3, v1, v2, two integrator updates, in + m1 v1) with c0..c3 = a1 a2 a3 m1, else TDF2 biquad
__device__ __forceinline__ float band_step(bool svf, float x, float c0, float c1, float c2, float c3, float c4, float &s1, float &s2) {
    if (svf) {
        const float v3 = x - s2;
        const float v1 = c0 * s1 + c1 * v3;
        const float v2 = s2 + c1 * s1 + c2 * v3;
        s1 = 2.0f * v1 - s1;
        s2 = 2.0f * v2 - s2;
        return x + c3 * v1;
    }
    const float y = c0 * x + s1;
    s1 = c1 * x - c3 * y + s2;
    s2 = c2 * x - c4 * y;
    return y;
}

__global__ void kernel_a(const float *coef, float *out, uint32_t chains, uint32_t frames, uint32_t svf_mask) {
    const uint32_t chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= chains) return;
    float b0[NB], b1[NB], b2[NB], a1[NB], a2[NB], s1[NB], s2[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { b0[b] = coef[b * 5]; b1[b] = coef[b * 5 + 1]; b2[b] = coef[b * 5 + 2]; a1[b] = coef[b * 5 + 3]; a2[b] = coef[b * 5 + 4]; s1[b] = s2[b] = 0.0f; }
    float acc = 0.0f;
    for (uint32_t n = 0; n < frames; ++n) {
        float x = input_sample(chain, n);
#pragma unroll
        for (int b = 0; b < NB; ++b) x = band_step((svf_mask >> b) & 1u, x, b0[b], b1[b], b2[b], a1[b], a2[b], s1[b], s2[b]);
        if ((n & 1023u) == 1023u) out[(size_t)chain * (frames >> 10) + (n >> 10)] = x; else acc += x * 0.0f;
    }
    if (acc != 0.0f) out[0] = acc;
}

__device__ __forceinline__ float row_shr1(float v) {      // lane i takes lane i-1's value inside its 16-lane row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}

__global__ void kernel_b(const float *coef, float *out, uint32_t chains, uint32_t frames, uint32_t svf_mask) {
    const uint32_t lane = threadIdx.x & 63u, band = lane & 15u, row = lane >> 4;
    const uint32_t chain = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4u + row;
    const bool live = band < NB && chain < chains;
    const int bb = band < NB ? (int)band : 0;
    const float b0 = coef[bb * 5], b1 = coef[bb * 5 + 1], b2 = coef[bb * 5 + 2], a1 = coef[bb * 5 + 3], a2 = coef[bb * 5 + 4];
    float s1 = 0.0f, s2 = 0.0f, y = 0.0f;
    const bool svf = (svf_mask >> bb) & 1u;
    // step t: band b works on frame t - b
    for (uint32_t t = 0; t < frames + NB - 1; ++t) {
        const float from_below = row_shr1(y);
        const float x = band == 0 ? input_sample(chain, t) : from_below;
        const uint32_t n = t - band;
        if (live && n < frames) {
            float t1 = s1, t2 = s2, u1 = s1, u2 = s2;
            const float ya = band_step(true, x, b0, b1, b2, a1, a2, t1, t2), yb = band_step(false, x, b0, b1, b2, a1, a2, u1, u2);
            y = svf ? ya : yb; s1 = svf ? t1 : u1; s2 = svf ? t2 : u2;
            if (band == NB - 1 && (n & 1023u) == 1023u) out[(size_t)chain * (frames >> 10) + (n >> 10)] = y;
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t chains = argc > 1 ? atoi(argv[1]) : 8192, frames = argc > 2 ? atoi(argv[2]) : 96000;
    const uint32_t svf_mask = (argc > 3 && atoi(argv[3])) ? 0x7fu : 0u;
    std::vector<float> coef(NB * 5);
    for (int b = 0; b < NB; ++b) { coef[b * 5] = 0.98f - 0.01f * b; coef[b * 5 + 1] = -1.7f + 0.02f * b; coef[b * 5 + 2] = 0.80f + 0.005f * b; coef[b * 5 + 3] = -1.72f + 0.02f * b; coef[b * 5 + 4] = 0.79f + 0.004f * b;
        if ((svf_mask >> b) & 1u) { coef[b * 5] = 0.96f - 0.01f * b; coef[b * 5 + 1] = 0.02f + 0.004f * b; coef[b * 5 + 2] = 0.0006f + 0.0002f * b; coef[b * 5 + 3] = 0.3f - 0.05f * b; coef[b * 5 + 4] = 0.0f; } }
    float *d_coef, *d_a, *d_b;
    const size_t n_out = (size_t)chains * (frames >> 10);
    CK(hipMalloc(&d_coef, coef.size() * 4)); CK(hipMalloc(&d_a, n_out * 4)); CK(hipMalloc(&d_b, n_out * 4));
    CK(hipMemcpy(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_a, 0, n_out * 4)); CK(hipMemset(d_b, 0, n_out * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_a = 0, ms_b = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kernel_a, dim3((chains + 63) / 64), dim3(64), 0, 0, d_coef, d_a, chains, frames, svf_mask); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_a, e0, e1));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kernel_b, dim3((chains + 3) / 4), dim3(64), 0, 0, d_coef, d_b, chains, frames, svf_mask); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_b, e0, e1));
    }
    std::vector<float> ha(n_out), hb(n_out);
    CK(hipMemcpy(ha.data(), d_a, n_out * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, n_out * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < n_out; ++i) if (memcmp(&ha[i], &hb[i], 4) != 0) ++diff;
    printf("chains %u frames %u bands %d (%s)\n", chains, frames, NB, svf_mask ? "7 SVF peaking + 3 biquads" : "biquads");
    printf("A (one lane per chain, bands in sequence): %.3f ms  = %.3g band-samples/s, %.1f ns per frame\n", ms_a, (double)chains * frames * NB / ms_a * 1e3, ms_a * 1e6 / frames);
    printf("B (one lane per band, DPP hand-over):      %.3f ms  = %.3g band-samples/s, %.1f ns per frame\n", ms_b, (double)chains * frames * NB / ms_b * 1e3, ms_b * 1e6 / frames);
    printf("outputs that differ: %zu of %zu (sample %g)\n", diff, n_out, ha[n_out / 2]);
    return diff != 0;
}
