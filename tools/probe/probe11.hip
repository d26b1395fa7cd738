// probe11 — the calibration kernel DESIGN.md section 9.4 asked for: BASELINE config 2's EXACT global access pattern on the latency layout
// (dspi_chain_skew.inc, first shape), with nothing else in it, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be read against KNOWN bytes.
//
// The pattern (4 096 streams x 2 000 packets x 48 frames, 256 workgroups x 8 waves, one workgroup = 16 streams = 8 stream pairs):
//   loads   systolic wave w (0-3), per batch of 192 frames, for its two pairs' two streams each: three 4-byte loads per lane,
//           lane = frame — pcm[stream * F + ws + 64 g + lane]: 256 contiguous bytes per instruction, streams 384 000 bytes apart,
//           the batch window starts 14 frames into the packet grid (kSkDepth: not line-aligned)                       [dspi_chain_skew.inc load_raw]
//   stores  output wave w (4-7), a batch behind, for pairs w - 4 and w, both streams: 8-byte {L, R} words per lane, lane = frame —
//           pairs[((stream * 4 + 0) * F + frame) * 2]: 512 contiguous bytes per instruction, pair 0 only (DSPI_OUT_ENABLED_ONLY: pairs 1-3 and
//           the sub stay unwritten), and per packet and stream eleven 2-byte peak words — peaks[(stream * n_blocks + k) * 11 + lane]    [sk_outputs]
// Known bytes: reads 4 B/frame/stream (+ the 14-frame lead once), writes 8 B/frame/stream + 22 B per packet and stream.
// PEAKS=0 drops the peak words.  Run: tools/probe/probe11_run.sh (rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE, separate passes).
// Build: hipcc --offload-arch=gfx950 -O3 -o probe11 probe11.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t S = 4096, B = 48, NB = 2000, F = B * NB, P = 4, C = 11, G = 4, LEAD = 14, BATCH = G * B;

__global__ __launch_bounds__(512) void c2_pattern(const uint32_t *__restrict__ pcm, uint32_t *__restrict__ pairs, uint16_t *__restrict__ peaks, uint32_t *sink, int with_peaks) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, wg = blockIdx.x;
    uint32_t acc = 0;
    if (wave < 4) {
        // systolic wave: batch "-1" = the 14-frame fill, then batches of G packets; its two pairs, both streams
        for (int p0 = -1; p0 < (int)NB; p0 = p0 < 0 ? 0 : p0 + (int)G) {
            const uint32_t ws = p0 < 0 ? 0u : (uint32_t)p0 * B + LEAD, len = p0 < 0 ? LEAD : BATCH;
            for (uint32_t h = 0; h < 2; ++h)
                for (uint32_t c = 0; c < 2; ++c) {
                    const uint32_t stream = wg * 16u + (2u * wave + h) * 2u + c;
#pragma unroll
                    for (uint32_t g = 0; g < 3; ++g) {
                        const uint32_t j = g * 64u + lane, f = ws + j;
                        if (j < len && f < F) acc ^= pcm[(size_t)stream * F + f];
                    }
                }
        }
        if (acc == 0x12345u) sink[0] = acc;
    } else {
        const uint32_t ow = wave - 4u;
        for (uint32_t p0 = 0; p0 < NB; p0 += G)
            for (uint32_t h = 0; h < 2; ++h)
                for (uint32_t c = 0; c < 2; ++c) {
                    const uint32_t stream = wg * 16u + (ow + 4u * h) * 2u + c;
#pragma unroll
                    for (uint32_t g = 0; g < 3; ++g) {
                        const uint32_t j = g * 64u + lane, f = p0 * B + j;
                        if (j < BATCH) *reinterpret_cast<u32x2 *>(pairs + (((size_t)stream * P + 0u) * F + f) * 2u) = u32x2{f, stream};
                    }
                    if (with_peaks)
                        for (uint32_t k = p0; k < p0 + G; ++k)
                            if (lane < C) peaks[((size_t)stream * NB + k) * C + lane] = (uint16_t)(k + lane);
                }
    }
}

int main() {
    const int with_peaks = getenv("PEAKS") ? atoi(getenv("PEAKS")) : 1;
    uint32_t *pcm, *pairs, *sink; uint16_t *peaks;
    const size_t in_b = (size_t)S * F * 4, pairs_b = (size_t)S * P * F * 8, peaks_b = (size_t)S * NB * C * 2;
    CK(hipMalloc(&pcm, in_b)); CK(hipMalloc(&pairs, pairs_b)); CK(hipMalloc(&peaks, peaks_b)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(pcm, 0, in_b));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(c2_pattern, dim3(S / 16), dim3(512), 0, 0, pcm, pairs, peaks, sink, with_peaks);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 3) {
            const double rd = (double)S * F * 4, wr = (double)S * F * 8 + (with_peaks ? (double)S * NB * C * 2 : 0.0);
            printf("c2_pattern peaks=%d: %.3f ms; known bytes: read %.0f (%.3f B/frame) written %.0f (%.3f B/frame) frames %.0f\n", with_peaks, ms, rd, rd / ((double)S * F),
                   wr, wr / ((double)S * F), (double)S * F);
        }
    }
    return 0;
}
