/*
 * orc_pdm.c — CPU restatement of DSPi's PDM sub output: the 256x oversampled 2nd-order sigma-delta modulator with
 * noise-shaped dither that consumes the chain's Q28 sub channel (reference firmware/DSPi/pdm_generator.c).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as the rest of oracle/): never linked into or imported by the product.
 *
 * PINNED: pdm_generator.c itself is compiled in place over a host stand-in for the pico-sdk in the firmware build of the oracle
 * (ref_fw_core1.c: pdm_processing_loop runs as a coroutine with an emulated DMA read pointer) and this restatement reproduces
 * its words bit for bit (tests/test_oracle_vs_fw.py::test_pdm_modulator_is_pdm_processing_loop); the reference ships no vectors,
 * analytical identities are in tests/test_oracle_pdm.py (DC bit density, silence pattern, fade-in length).
 *
 * What is restated (per input sample, in this order):
 *   hard limiter  pcm = sample >> 14, clamp to +-PDM_CLIP_THRESH            pdm_generator.c:351-354, config.h:64
 *   fade-in       pcm = pcm * fade_in_pos >> 10 for the first 1024 samples   :356-360, config.h:74-75
 *   target        pcm + 32768                                                :363
 *   8 chunks of   raw dither (xorshift32 & 0x1FF) - 255                      :62-68, :368, config.h:68
 *                 noise-shaped dither (leaky error feedback + 2nd-order HP)  :79-108, :369
 *                 32 modulator steps, MSB first                              :371-378
 *   leaky integrators err -= err >> 16                                       :396-397, config.h:71
 * What is not: the DMA ring pacing, under/overrun recovery and the fade-OUT on disable (:217-230, :256-347) — those are
 * properties of the hardware transport, not of the sample stream.
 */
#include <stdint.h>
#include <string.h>

#define PDM_CLIP_THRESH 29500
#define PDM_DITHER_MASK 0x1FF
#define PDM_LEAKAGE_SHIFT 16
#define PDM_FADE_IN_SHIFT 10
#define PDM_FADE_IN_SAMPLES (1u << PDM_FADE_IN_SHIFT)
#define NS_B0 15778
#define NS_B1 (-31556)
#define NS_B2 15778
#define NS_A1 31531
#define NS_A2 15580

typedef struct {
    int32_t err, err2;                 /* local_pdm_err, local_pdm_err2 */
    int32_t x1, x2, y1, y2, err_acc;   /* noise_shaper_t */
    uint32_t rng;                      /* rng_state (file scope upstream: survives re-enable) */
    uint32_t fade_in_pos;
} orc_pdm_state;

/* all arithmetic wraps mod 2^32 like the Cortex-M code (-fwrapv); >> on negatives is arithmetic */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

void orc_pdm_init(orc_pdm_state *s) {        /* power-on: statics of pdm_generator.c:63 and :205-221 */
    memset(s, 0, sizeof *s);
    s->rng = 123456789u;
}

void orc_pdm_restart(orc_pdm_state *s) {     /* the re-enable path, pdm_generator.c:241-252: everything but the RNG */
    uint32_t r = s->rng;
    memset(s, 0, sizeof *s);
    s->rng = r;
}

static inline uint32_t fast_rand(orc_pdm_state *s) {          /* :64-69 */
    uint32_t r = s->rng;
    r ^= r << 13; r ^= r >> 17; r ^= r << 5;
    s->rng = r;
    return r;
}

static inline int32_t noise_shaped_dither(orc_pdm_state *s, int32_t raw, int32_t quant_error) {   /* :85-103 */
    s->err_acc = wadd(wmul(s->err_acc, 248) >> 8, quant_error >> 6);
    int32_t input = wsub(raw, s->err_acc);
    int32_t acc = wmul(NS_B0, input);
    acc = wadd(acc, wmul(NS_B1, s->x1));
    acc = wadd(acc, wmul(NS_B2, s->x2));
    acc = wadd(acc, wmul(NS_A1, s->y1));
    acc = wsub(acc, wmul(NS_A2, s->y2));
    int32_t out = acc >> 14;
    s->x2 = s->x1; s->x1 = input;
    s->y2 = s->y1; s->y1 = out;
    return out;
}

/* n Q28 sub samples in, n*8 32-bit PDM words out (MSB = first bit on the wire, :373) */
void orc_pdm_run(orc_pdm_state *s, const int32_t *sub, uint32_t n, uint32_t *words) {
    for (uint32_t i = 0; i < n; i++) {
        int32_t pcm = sub[i] >> 14;
        if (pcm > PDM_CLIP_THRESH) pcm = PDM_CLIP_THRESH;
        if (pcm < -PDM_CLIP_THRESH) pcm = -PDM_CLIP_THRESH;
        if (s->fade_in_pos < PDM_FADE_IN_SAMPLES) {
            pcm = wmul(pcm, (int32_t)s->fade_in_pos) >> PDM_FADE_IN_SHIFT;
            s->fade_in_pos++;
        }
        const int32_t target = pcm + 32768;
        for (int chunk = 0; chunk < 8; chunk++) {
            int32_t raw = (int32_t)(fast_rand(s) & PDM_DITHER_MASK) - (PDM_DITHER_MASK >> 1);
            int32_t dither = noise_shaped_dither(s, raw, s->err2 >> 8);
            uint32_t w = 0;
            for (int k = 0; k < 32; k++) {
                const int hi = wadd(s->err2, dither) >= 0;
                const int32_t fb = hi ? 65535 : 0;
                if (hi) w |= 1u << (31 - k);
                s->err = wadd(s->err, wsub(target, fb));
                s->err2 = wadd(s->err2, wsub(s->err, fb));
            }
            words[(size_t)i * 8 + chunk] = w;
        }
        s->err = wsub(s->err, s->err >> PDM_LEAKAGE_SHIFT);
        s->err2 = wsub(s->err2, s->err2 >> PDM_LEAKAGE_SHIFT);
    }
}

int orc_pdm_state_words(void) { return (int)(sizeof(orc_pdm_state) / 4); }
