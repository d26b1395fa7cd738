// dspi_spdif_dev.h — IEC 60958 subframe encoding of one sample word, device code shared by the stand-alone encoder (dspi_spdif.hip) and the
// output waves of the latency layout (dspi_chain_skew*.inc, DSPI_OUT_SPDIF).  Reference: pico_audio_spdif_multi, see dspi_spdif.hip.
#pragma once
#include <stdint.h>
namespace dspi {
// spdif_lookup[b] (audio_spdif.c:141-153): low 16 bits = 0x5555 | (bit j of b) << (2j+1), bit 16 = parity of b
static __device__ __forceinline__ uint32_t bmc_byte(uint32_t b) {
    uint32_t x = b & 0xffu;
    x = (x | (x << 4)) & 0x0f0fu;
    x = (x | (x << 2)) & 0x3333u;
    x = (x | (x << 1)) & 0x5555u;
    return 0x5555u | (x << 1) | ((uint32_t)(__builtin_popcount(b & 0xffu) & 1) << 16);
}
// spdif_update_subframe on a pre-filled subframe {preamble, 0x55000000 | c << 29} (sample_encoding.h:27-47)
static __device__ __forceinline__ void subframe(uint32_t sample, uint32_t preamble, uint32_t c_bit, uint32_t &l, uint32_t &h) {
    const uint32_t s0 = bmc_byte(sample), s1 = bmc_byte(sample >> 8), s2 = bmc_byte(sample >> 16);
    l = preamble | ((s0 & 0xffffu) << 8) | (s1 << 24);
    const uint32_t ph = 0x55u | (c_bit << 5);
    uint32_t p = (s0 >> 16) ^ (s1 >> 16) ^ (s2 >> 16);
    p ^= (((ph & 0x2au) * 0x2au) >> 6) & 1u;
    h = ((s1 & 0xffffu) >> 8) | ((s2 & 0xffffu) << 8) | ((ph & 0x7fu) << 24) | (p << 31);
}
// IEC 60958-3 consumer channel status, first word (audio_spdif.c:83-89, sample-rate byte :250-256); the fifth byte is kSpdifStatusHi
static __host__ __device__ __forceinline__ uint32_t spdif_status_lo(uint32_t fs) {
    const uint32_t rate = fs == 44100 ? 0x00u : fs == 48000 ? 0x02u : fs == 96000 ? 0x0Au : 0x01u;
    return 0x04u | (rate << 24);
}
constexpr uint32_t kSpdifStatusHi = 0x0Bu;
// a frame's two subframes (left: preamble Z at the block start, else X; right: Y) with the channel-status bit of its block position
// (40 bits, the rest zero: audio_spdif.c:91-94)
typedef uint32_t spdif_u4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ spdif_u4 spdif_frame(uint32_t wl, uint32_t wr, uint32_t pos, uint32_t status_lo, uint32_t status_hi) {
    const uint32_t c_bit = pos < 32 ? (status_lo >> pos) & 1u : (pos < 40 ? (status_hi >> (pos - 32)) & 1u : 0u);
    uint32_t l0, h0, l1, h1;
    subframe(wl, pos == 0 ? 0x39u : 0xC9u, c_bit, l0, h0);
    subframe(wr, 0x69u, c_bit, l1, h1);
    return spdif_u4{l0, h0, l1, h1};
}
}  // namespace dspi
