// dspi_pdm.hip — DSPi's PDM sub output on the GPU: 256x oversampled 2nd-order sigma-delta modulator with noise-shaped
// dither, the consumer of the chain's Q28 sub channel (reference firmware/DSPi/pdm_generator.c).
//
// Per input sample (reference lines):  hard limiter :351-354, fade-in :356-360, target :363, then 8 chunks of
// { xorshift32 dither :62-68/:368, noise shaper :79-108/:369, 32 modulator steps MSB first :371-378 }, leaky
// integrators :396-397.  Integer arithmetic only, every add/sub/mul wrapping mod 2^32 like the Cortex-M code, so the
// words are bit-exact against the CPU restatement the tests use.  The DMA pacing / under-run recovery / fade-out of
// the firmware loop are transport, not sample arithmetic, and are not modelled.
//
// Mapping: one lane = one stream (the modulator is a 256-step serial recurrence per sample; the only parallelism is
// streams).  A workgroup is one tile row of the context (128 float / 64 Q28 streams): with the tiled layouts every
// access is one coalesced row ([tile][frame][R] in, [tile][frame][8][R] out).  VALU-issue bound: ~2 800 integer
// instructions per sample; 32 bytes out per 4 bytes in.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dspi_kernels.h"

namespace dspi {

namespace {

constexpr int32_t kClip = 29500;          // PDM_CLIP_THRESH (config.h:64)
constexpr uint32_t kDitherMask = 0x1FF;   // PDM_DITHER_MASK (config.h:68)
constexpr int kLeak = 16;                 // PDM_LEAKAGE_SHIFT (config.h:71)
constexpr int kFadeShift = 10;            // PDM_FADE_IN_SHIFT (config.h:74)

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

// 256 threads = four waves = one per SIMD of a CU, 256 / row tile rows per workgroup: with two-wave workgroups the
// dispatcher put both workgroups of a CU on the same two SIMDs and left the other two idle (measured: exactly 2x).
constexpr int kPdmThreads = 256;

template <bool TILED>
__global__ __launch_bounds__(kPdmThreads) void pdm_kernel(uint32_t *state, const int32_t *sub, uint32_t *words, uint32_t n_streams, uint32_t n_frames,
                                                           uint32_t row) {
    const uint32_t wg = blockIdx.x * (kPdmThreads / row) + threadIdx.x / row, col = threadIdx.x % row;
    const uint32_t stream = wg * row + col;
    if (stream >= n_streams) return;
    uint32_t *gs = state + (size_t)wg * kPdmStateWords * row + col;
    int32_t err = (int32_t)gs[0 * row], err2 = (int32_t)gs[1 * row];
    int32_t x1 = (int32_t)gs[2 * row], x2 = (int32_t)gs[3 * row], y1 = (int32_t)gs[4 * row], y2 = (int32_t)gs[5 * row], err_acc = (int32_t)gs[6 * row];
    uint32_t rng = gs[7 * row], fade = gs[8 * row];

    const int32_t *in = TILED ? sub + (size_t)wg * n_frames * row + col : sub + (size_t)stream * n_frames;
    uint32_t *out = TILED ? words + (size_t)wg * n_frames * 8 * row + col : words + (size_t)stream * n_frames * 8;
    const size_t in_step = TILED ? row : 1, out_step = TILED ? row : 1;

    int32_t next = n_frames ? in[0] : 0;
    for (uint32_t f = 0; f < n_frames; ++f) {
        const int32_t sample = next;
        if (f + 1 < n_frames) next = in[(size_t)(f + 1) * in_step];      // the next sample lands under 2 800 instructions
        int32_t pcm = sample >> 14;
        pcm = pcm > kClip ? kClip : pcm;
        pcm = pcm < -kClip ? -kClip : pcm;
        if (fade < (1u << kFadeShift)) { pcm = wmul(pcm, (int32_t)fade) >> kFadeShift; ++fade; }
        const int32_t target = pcm + 32768;
        uint32_t w[8];
#pragma unroll
        for (int chunk = 0; chunk < 8; ++chunk) {
            rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;                         // fast_rand
            const int32_t raw = (int32_t)(rng & kDitherMask) - (int32_t)(kDitherMask >> 1);
            err_acc = wadd(wmul(err_acc, 248) >> 8, (err2 >> 8) >> 6);                   // noise_shaped_dither
            const int32_t input = wsub(raw, err_acc);
            int32_t acc = wmul(15778, input);
            acc = wadd(acc, wmul(-31556, x1));
            acc = wadd(acc, wmul(15778, x2));
            acc = wadd(acc, wmul(31531, y1));
            acc = wsub(acc, wmul(15580, y2));
            const int32_t dither = acc >> 14;
            x2 = x1; x1 = input; y2 = y1; y1 = dither;
            // 32 modulator steps (pdm_generator.c:371-378), five instructions each instead of the literal ten:
            //   fb = bit ? 65535 : 0;  err += target - fb;  err2 += err - fb;  bit = (err2 + dither >= 0)
            // with n = (err2 + dither) >> 31 (-1 for a 0 bit), fbn = n & 65535 = 65535 - fb, and the shifted variables
            // e = err - 65535, u = err2 + dither (dither is constant inside a chunk):
            //   e += (target - 65535) + fbn;   u += e + fbn;   word = 2*word + n   (= the bits minus 2^32-1, fixed up below)
            // Same values mod 2^32 at every step, so the words and the carried state are identical.
            const uint32_t tm = (uint32_t)target - 65535u;
            uint32_t e = (uint32_t)err - 65535u, u = (uint32_t)err2 + (uint32_t)dither, wacc = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t n = (uint32_t)((int32_t)u >> 31);
                const uint32_t fbn = n & 65535u;
                wacc = (wacc << 1) + n;
                e = e + tm + fbn;
                u = u + e + fbn;
            }
            const uint32_t word = wacc - 1u;
            err = (int32_t)(e + 65535u);
            err2 = (int32_t)(u - (uint32_t)dither);
            w[chunk] = word;
        }
        err = wsub(err, err >> kLeak);
        err2 = wsub(err2, err2 >> kLeak);
        uint32_t *o = out + (size_t)f * 8 * out_step;
        if (TILED) {
#pragma unroll
            for (int chunk = 0; chunk < 8; ++chunk) o[(size_t)chunk * row] = w[chunk];
        } else {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u4 *>(o) = u4{w[0], w[1], w[2], w[3]};
            *reinterpret_cast<u4 *>(o + 4) = u4{w[4], w[5], w[6], w[7]};
        }
    }
    gs[0 * row] = (uint32_t)err; gs[1 * row] = (uint32_t)err2;
    gs[2 * row] = (uint32_t)x1; gs[3 * row] = (uint32_t)x2; gs[4 * row] = (uint32_t)y1; gs[5 * row] = (uint32_t)y2; gs[6 * row] = (uint32_t)err_acc;
    gs[7 * row] = rng; gs[8 * row] = fade;
}

// power-on (init != 0: rng = 123456789, pdm_generator.c:63) or the re-enable path (:241-252: everything but the RNG)
__global__ void pdm_reset_kernel(uint32_t *state, uint32_t n_streams, uint32_t row, int32_t only_stream, int init) {
    const uint32_t wg = blockIdx.x, col = threadIdx.x;
    const uint32_t stream = wg * row + col;
    if (col >= row || stream >= n_streams || (only_stream >= 0 && stream != (uint32_t)only_stream)) return;
    uint32_t *gs = state + (size_t)wg * kPdmStateWords * row + col;
    for (int i = 0; i < kPdmStateWords; ++i)
        if (i != 7) gs[(size_t)i * row] = 0;
    if (init) gs[7 * row] = 123456789u;
}

}  // namespace

hipError_t launch_pdm(bool tiled, uint32_t *state, const int32_t *sub, uint32_t *words, uint32_t n_streams, uint32_t n_frames, uint32_t row,
                      uint32_t n_wg, hipStream_t stream) {
    const uint32_t per = kPdmThreads / row, blocks = (n_wg + per - 1) / per;
    if (tiled) hipLaunchKernelGGL(pdm_kernel<true>, dim3(blocks), dim3(kPdmThreads), 0, stream, state, sub, words, n_streams, n_frames, row);
    else hipLaunchKernelGGL(pdm_kernel<false>, dim3(blocks), dim3(kPdmThreads), 0, stream, state, sub, words, n_streams, n_frames, row);
    return hipGetLastError();
}

hipError_t launch_pdm_reset(uint32_t *state, uint32_t n_streams, uint32_t row, uint32_t n_wg, int32_t only_stream, int init, hipStream_t stream) {
    hipLaunchKernelGGL(pdm_reset_kernel, dim3(n_wg), dim3(128), 0, stream, state, n_streams, row, only_stream, init);
    return hipGetLastError();
}

}  // namespace dspi
