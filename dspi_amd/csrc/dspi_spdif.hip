// dspi_spdif.hip — output wire formats of the chain's int24 pair words on the GPU: IEC 60958 (S/PDIF)
// subframe encoding, and the I2S slots' left-justified words (at the end of the file).
//
// Reference: firmware/pico-extras/src/rp2_common/pico_audio_spdif_multi/ — what DSPi's S/PDIF outputs do to the words
// of dspi_out.pairs before the PIO shifts them out:
//   spdif_update_subframe           include/pico/audio_spdif/sample_encoding.h:27-47 (three byte look-ups, parity)
//   byte table                      audio_spdif.c:141-153 (every data bit b becomes the cell pair "1 b": 0x5555 | b<<(2j+1))
//   preambles / channel status      audio_spdif.c:76-94, :101-116, sample-rate byte :250-256, block restamp :385-405
// A frame's two subframes (4 words) are a pure function of (left word, right word, position in the 192-frame block,
// sample rate), so this is byte shuffling at memory speed: 8 bytes in, 16 bytes out per frame and pair.  The byte table
// is replaced by the bit spread it tabulates (three shift/or/and steps), so there is no LDS gather.
//
// Mapping: workgroup = (tile row of the context, pair), lane = stream; with the tiled layouts every access is one
// coalesced row: in [tile][output][frame][R], out [tile][pair][frame][4][R].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dspi_kernels.h"
#include "dspi_spdif_dev.h"

namespace dspi {

namespace {

constexpr uint32_t kFrameSlice = 32;

template <bool TILED>
__global__ __launch_bounds__(128) void spdif_kernel(const int32_t *pairs, uint32_t *out, uint32_t n_streams, uint32_t n_pairs, uint32_t n_frames,
                                                     uint32_t row, uint32_t block_pos, uint32_t status_lo, uint32_t status_hi, SpdifRates rates) {
    const uint32_t wg = blockIdx.x, pair = blockIdx.y, col = threadIdx.x;
    const uint32_t stream = wg * row + col;
    if (col >= row || stream >= n_streams) return;
    if (rates.stream_image) status_lo = spdif_status_lo(rates.img[rates.stream_image[rates.stream0 + stream]].fs_hz);
    const size_t F = n_frames;
    // tiled: outputs 2*pair and 2*pair+1 are separate [frame][R] planes; stream-major: interleaved [frame][2]
    const int32_t *inL = TILED ? pairs + ((size_t)wg * (2 * n_pairs) + 2 * pair) * F * row + col : pairs + ((size_t)stream * n_pairs + pair) * F * 2;
    const int32_t *inR = TILED ? inL + F * row : inL + 1;
    const size_t in_step = TILED ? row : 2;
    uint32_t *o = TILED ? out + ((size_t)wg * n_pairs + pair) * F * 4 * row + col : out + ((size_t)stream * n_pairs + pair) * F * 4;
    // frames are independent: blockIdx.z takes a slice of kFrameSlice frames so that thousands of waves stream at once
    const uint32_t f0 = blockIdx.z * kFrameSlice, f1 = min(n_frames, f0 + kFrameSlice);
    uint32_t pos = (block_pos + f0) % 192u;
    for (uint32_t f = f0; f < f1; ++f) {
        const uint32_t wl = (uint32_t)inL[(size_t)f * in_step], wr = (uint32_t)inR[(size_t)f * in_step];
        // channel status bit of this block position (40 bits, the rest zero): audio_spdif.c:91-94
        const uint32_t c_bit = pos < 32 ? (status_lo >> pos) & 1u : (pos < 40 ? (status_hi >> (pos - 32)) & 1u : 0u);
        uint32_t l0, h0, l1, h1;
        subframe(wl, pos == 0 ? 0x39u : 0xC9u, c_bit, l0, h0);      // PREAMBLE_Z at the block start, else PREAMBLE_X
        subframe(wr, 0x69u, c_bit, l1, h1);                          // PREAMBLE_Y
        if (TILED) {
            uint32_t *q = o + (size_t)f * 4 * row;
            q[0] = l0; q[row] = h0; q[2 * row] = l1; q[3 * row] = h1;
        } else {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u4 *>(o + (size_t)f * 4) = u4{l0, h0, l1, h1};
        }
        pos = (pos + 1 == 192u) ? 0u : pos + 1;
    }
}

// Stream-major buffers are contiguous in time per (stream, pair), and a frame's subframes depend on nothing but its own
// words and its block position: one lane per FRAME here, so a wave reads 512 and writes 1024 contiguous bytes.
__global__ __launch_bounds__(256) void spdif_kernel_frames(const int32_t *pairs, uint32_t *out, uint64_t total, uint32_t n_frames, uint32_t block_pos,
                                                          uint32_t status_lo, uint32_t status_hi, SpdifRates rates, uint32_t n_pairs) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;      // (stream * n_pairs + pair) * n_frames + frame
    if (idx >= total) return;
    if (rates.stream_image) status_lo = spdif_status_lo(rates.img[rates.stream_image[rates.stream0 + (uint32_t)(idx / ((uint64_t)n_pairs * n_frames))]].fs_hz);
    const uint32_t f = (uint32_t)(idx % n_frames);
    const uint32_t pos = (block_pos + f) % 192u;
    const uint32_t wl = (uint32_t)pairs[idx * 2], wr = (uint32_t)pairs[idx * 2 + 1];
    const uint32_t c_bit = pos < 32 ? (status_lo >> pos) & 1u : (pos < 40 ? (status_hi >> (pos - 32)) & 1u : 0u);
    uint32_t l0, h0, l1, h1;
    subframe(wl, pos == 0 ? 0x39u : 0xC9u, c_bit, l0, h0);
    subframe(wr, 0x69u, c_bit, l1, h1);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u4 *>(out + idx * 4) = u4{l0, h0, l1, h1};
}

// even frame counts: two frames per lane (16 bytes in, 32 bytes out)
__global__ __launch_bounds__(256) void spdif_kernel_frames2(const int32_t *pairs, uint32_t *out, uint64_t total2, uint32_t half_frames, uint32_t block_pos,
                                                           uint32_t status_lo, uint32_t status_hi, SpdifRates rates, uint32_t n_pairs) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;      // (stream * n_pairs + pair) * (n_frames / 2) + frame pair
    if (idx >= total2) return;
    if (rates.stream_image) status_lo = spdif_status_lo(rates.img[rates.stream_image[rates.stream0 + (uint32_t)(idx / ((uint64_t)n_pairs * half_frames))]].fs_hz);
    const uint32_t f = (uint32_t)(idx % half_frames) * 2u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 w = *reinterpret_cast<const u4 *>(pairs + idx * 4);
    u4 o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t pos = (block_pos + f + k) % 192u;
        const uint32_t c_bit = pos < 32 ? (status_lo >> pos) & 1u : (pos < 40 ? (status_hi >> (pos - 32)) & 1u : 0u);
        uint32_t l0, h0, l1, h1;
        subframe(k ? w.z : w.x, pos == 0 ? 0x39u : 0xC9u, c_bit, l0, h0);
        subframe(k ? w.w : w.y, 0x69u, c_bit, l1, h1);
        o[k] = u4{l0, h0, l1, h1};
    }
    *reinterpret_cast<u4 *>(out + idx * 8) = o[0];
    *reinterpret_cast<u4 *>(out + idx * 8 + 4) = o[1];
}

// ---- I2S slots: pico_audio_i2s_multi/audio_i2s_multi.c:217-226 — the same producer words left-justified (<< 8), L then R.
// 8 bytes in, 8 bytes out per frame: a pure copy-with-shift at memory speed.  Pairs outside `pair_mask` are not touched.
__global__ __launch_bounds__(256) void i2s_kernel_frames(const int32_t *pairs, uint32_t *out, uint64_t total, uint32_t n_frames, uint32_t n_pairs, uint32_t pair_mask) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;      // (stream * n_pairs + pair) * n_frames + frame
    if (idx >= total) return;
    const uint32_t pair = (uint32_t)((idx / n_frames) % n_pairs);
    if (!((pair_mask >> pair) & 1u)) return;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const u2 w = *reinterpret_cast<const u2 *>(pairs + idx * 2);
    *reinterpret_cast<u2 *>(out + idx * 2) = u2{w.x << 8, w.y << 8};
}
// even frame counts: two frames (16 bytes) per lane
__global__ __launch_bounds__(256) void i2s_kernel_frames2(const int32_t *pairs, uint32_t *out, uint64_t total2, uint32_t half_frames, uint32_t n_pairs, uint32_t pair_mask) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;      // (stream * n_pairs + pair) * (n_frames / 2) + frame pair
    if (idx >= total2) return;
    const uint32_t pair = (uint32_t)((idx / half_frames) % n_pairs);
    if (!((pair_mask >> pair) & 1u)) return;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 w = *reinterpret_cast<const u4 *>(pairs + idx * 4);
    *reinterpret_cast<u4 *>(out + idx * 4) = u4{w.x << 8, w.y << 8, w.z << 8, w.w << 8};
}
// tiled: [tile][output][frame][R] words, R a multiple of 4: one lane per 4 streams of one (output, frame)
__global__ __launch_bounds__(256) void i2s_kernel_tiled(const int32_t *pairs, uint32_t *out, uint64_t total4, uint64_t plane_words, uint32_t n_out, uint32_t pair_mask) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= total4) return;
    const uint32_t output = (uint32_t)((idx * 4 / plane_words) % n_out);
    if (!((pair_mask >> (output >> 1)) & 1u)) return;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 w = *reinterpret_cast<const u4 *>(pairs + idx * 4);
    *reinterpret_cast<u4 *>(out + idx * 4) = u4{w.x << 8, w.y << 8, w.z << 8, w.w << 8};
}

}  // namespace

hipError_t launch_i2s(bool tiled, const int32_t *pairs, uint32_t *out, uint32_t n_streams, uint32_t n_pairs, uint32_t n_frames, uint32_t row,
                      uint32_t n_wg, uint32_t pair_mask, hipStream_t stream) {
    if (tiled) {
        const uint64_t plane = (uint64_t)n_frames * row, total4 = (uint64_t)n_wg * 2 * n_pairs * plane / 4;
        hipLaunchKernelGGL(i2s_kernel_tiled, dim3((uint32_t)((total4 + 255) / 256)), dim3(256), 0, stream, pairs, out, total4, plane, 2 * n_pairs, pair_mask);
    } else {
        const uint64_t total = (uint64_t)n_streams * n_pairs * n_frames;
        if (n_frames % 2 == 0) hipLaunchKernelGGL(i2s_kernel_frames2, dim3((uint32_t)((total / 2 + 255) / 256)), dim3(256), 0, stream, pairs, out, total / 2, n_frames / 2, n_pairs, pair_mask);
        else hipLaunchKernelGGL(i2s_kernel_frames, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, pairs, out, total, n_frames, n_pairs, pair_mask);
    }
    return hipGetLastError();
}

hipError_t launch_spdif(bool tiled, const int32_t *pairs, uint32_t *out, uint32_t n_streams, uint32_t n_pairs, uint32_t n_frames, uint32_t row,
                        uint32_t n_wg, uint32_t block_pos, uint32_t fs, const SpdifRates &rates, hipStream_t stream) {
    // IEC 60958-3 consumer channel status, 5 bytes (audio_spdif.c:83-89, sample-rate byte :250-256): of `fs` for every stream, or
    // (rates.stream_image set) of each stream's own image
    const uint32_t lo = spdif_status_lo(fs), hi = kSpdifStatusHi;
    if (tiled) hipLaunchKernelGGL(spdif_kernel<true>, dim3(n_wg, n_pairs, (n_frames + kFrameSlice - 1) / kFrameSlice), dim3(128), 0, stream, pairs, out, n_streams, n_pairs, n_frames, row, block_pos, lo, hi, rates);
    else {
        const uint64_t total = (uint64_t)n_streams * n_pairs * n_frames;
        if (n_frames % 2 == 0) hipLaunchKernelGGL(spdif_kernel_frames2, dim3((uint32_t)((total / 2 + 255) / 256)), dim3(256), 0, stream, pairs, out, total / 2, n_frames / 2, block_pos, lo, hi, rates, n_pairs);
        else hipLaunchKernelGGL(spdif_kernel_frames, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, pairs, out, total, n_frames, block_pos, lo, hi, rates, n_pairs);
    }
    return hipGetLastError();
}

}  // namespace dspi
