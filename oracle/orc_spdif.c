/*
 * orc_spdif.c — CPU restatement of the S/PDIF (IEC 60958) subframe encoding the firmware applies to the chain's int24
 * pair words before its PIO shifts them out.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as the rest of oracle/).
 *
 * Reference (firmware/pico-extras/src/rp2_common/pico_audio_spdif_multi/):
 *   NRZI/BMC byte table            audio_spdif.c:141-153
 *   spdif_update_subframe          include/pico/audio_spdif/sample_encoding.h:27-47
 *   preambles, channel status      audio_spdif.c:76-94, buffer pre-fill :101-116, sample-rate byte :250-256
 *   block position restamp         audio_spdif.c:385-405  (Z preamble and C bits follow the 192-frame block position)
 *   producer (S32 stereo)          sample_encoding.cpp:38-45
 * Built twice: liborc_spdif.so (everything restated) and _ref/libref_spdif.so (ORC_USE_REF=1: spdif_update_subframe
 * comes from the reference's own header, compiled in place; tests/test_oracle_spdif.py compares the two).
 * What a frame's two subframes look like on the wire is a pure function of (left word, right word, block position,
 * sample rate); the pool/DMA machinery around it is transport and is not modelled.
 */
#include <stdint.h>

#if ORC_USE_REF
#include "pico/audio_spdif/sample_encoding.h"     /* the reference's own inline spdif_update_subframe + spdif_subframe_t */
uint32_t spdif_lookup[256];
#else
typedef struct { uint32_t l, h; } spdif_subframe_t;
static uint32_t spdif_lookup[256];
static inline void spdif_update_subframe(spdif_subframe_t *sf, int32_t sample) {     /* sample_encoding.h:27-47 */
    uint32_t s0 = spdif_lookup[(uint8_t)sample];
    uint32_t s1 = spdif_lookup[(uint8_t)(sample >> 8u)];
    uint32_t s2 = spdif_lookup[(uint8_t)(sample >> 16u)];
    sf->l = (sf->l & 0xffu) | (((uint16_t)s0) << 8u) | (s1 << 24u);
    uint32_t ph = sf->h >> 24u;
    uint32_t h = (((uint16_t)s1) >> 8u) | (((uint16_t)s2) << 8u);
    uint32_t p = (s0 >> 16u) ^ (s1 >> 16u) ^ (s2 >> 16u);
    p = p ^ ((((ph & 0x2a) * 0x2a) >> 6u) & 1u);
    sf->h = h | ((ph & 0x7f) << 24u) | (p << 31u);
}
#endif

#define PREAMBLE_X 0xC9u     /* 0b11001001 (audio_spdif.c:76-78) */
#define PREAMBLE_Y 0x69u
#define PREAMBLE_Z 0x39u

static int lookup_ready = 0;
static void lookup_init(void) {                      /* audio_spdif.c:141-153 */
    if (lookup_ready) return;
    for (unsigned i = 0; i < 256; i++) {
        uint32_t v = 0x5555;
        unsigned p = 0;
        for (unsigned j = 0; j < 8; j++)
            if (i & (1u << j)) { p ^= 1; v |= (2u << (j * 2)); }
        spdif_lookup[i] = v | ((uint32_t)p << 16u);
    }
    lookup_ready = 1;
}

static unsigned channel_status_bit(unsigned idx, uint32_t fs) {      /* audio_spdif.c:83-94, :250-256 */
    uint8_t cs[5] = {0x04, 0x00, 0x00, 0x01, 0x0B};
    cs[3] = fs == 44100 ? 0x00 : fs == 48000 ? 0x02 : fs == 96000 ? 0x0A : 0x01;
    if (idx >= 40) return 0;
    return (cs[idx / 8] >> (idx % 8)) & 1u;
}

int orc_spdif_is_ref_build(void) { return ORC_USE_REF ? 1 : 0; }

/* pair: [n][2] int32 words (24-bit payload); out: [n][2 subframes][2 words l,h]; returns the next block position */
uint32_t orc_spdif_encode(const int32_t *pair, uint32_t n, uint32_t block_pos, uint32_t fs, uint32_t *out) {
    lookup_init();
    for (uint32_t i = 0; i < n; i++) {
        const unsigned c_bit = channel_status_bit(block_pos, fs);
        spdif_subframe_t sf;
        sf.l = (block_pos == 0) ? PREAMBLE_Z : PREAMBLE_X;          /* init_spdif_buffer + the restamp at DMA start */
        sf.h = 0x55000000u | ((uint32_t)c_bit << 29u);
        spdif_update_subframe(&sf, 0);                               /* as the pre-fill does (audio_spdif.c:109) */
        spdif_update_subframe(&sf, pair[i * 2]);
        out[i * 4 + 0] = sf.l; out[i * 4 + 1] = sf.h;
        sf.l = PREAMBLE_Y;
        sf.h = 0x55000000u | ((uint32_t)c_bit << 29u);
        spdif_update_subframe(&sf, 0);
        spdif_update_subframe(&sf, pair[i * 2 + 1]);
        out[i * 4 + 2] = sf.l; out[i * 4 + 3] = sf.h;
        if (++block_pos == 192u) block_pos = 0;
    }
    return block_pos;
}

/* ---- I2S slots (pico_audio_i2s_multi/audio_i2s_multi.c:178-243) -------------------------------------------------------
 * An output slot whose type is OUTPUT_TYPE_I2S (config.h:286-287, output_types[]) takes the same producer words and
 * left-justifies them: dst = src << 8 for L and R (:223-226), MSB first on the wire, low byte zero.  The consumer-buffer
 * slicing around it (:206-238) is transport: completed buffers back to back are the input order.
 * pair: [n][2] int32; out: [n][2] uint32.  Pinned by _ref/libref_i2s.so (ref_i2s.c), tests/test_oracle_spdif.py. */
void orc_i2s_frames(const int32_t *pair, uint32_t n, uint32_t *out) {
    for (uint32_t i = 0; i < n; i++) {
        out[i * 2] = (uint32_t)pair[i * 2] << 8;
        out[i * 2 + 1] = (uint32_t)pair[i * 2 + 1] << 8;
    }
}
