/*
 * orc_chain.c — CPU restatement of the DSPi packet orchestrator and its control surface.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/README.md).  Nothing here is linked into the product.
 *
 * The reference keeps the whole chain in one static function over ~40 file-scope globals
 * (firmware/DSPi/usb_audio.c:500-1317) inside a translation unit that needs the un-vendored
 * pico-sdk, so it cannot be compiled here; this file restates it over an explicit context
 * struct, one stream per context.  Two builds share this file:
 *
 *   standalone (ORC_USE_REF=0): leaf DSP = orc_leaf.c (our restatement)        -> liborc_*.so
 *   _ref       (ORC_USE_REF=1): leaf DSP = the reference's own dsp_pipeline.c, leveller.c,
 *              crossfeed.c, loudness.c, bulk_params.c compiled in place          -> _ref/libref_*.so
 *
 * Restated here in both builds (no compilable reference exists):
 *   - process_audio_packet           usb_audio.c:528-532, :560-967 (float), :968-1283 (Q28)
 *   - Core-1 twin (same arithmetic)  pdm_generator.c:443-516 / :566-639
 *   - Q28 block biquad               dsp_process_rp2040.S:225-394 (Thumb asm)
 *   - volume / mute scalars          usb_audio.c:244-269, :409-440, :446-498
 *   - vendor SET / GET               usb_audio.c:1641-2017, :2271-2688
 *   - deferred-apply dispatcher      main.c:132-171, :826-894, :926-976, :1126-1162
 *   - preset slot (de)serialisation  flash_storage.c:136-189, :282-306, :464-742, :794-849, :1144-1238
 *
 * Build flags: -O2 -fwrapv -ffp-contract=off ; callers run with MXCSR FTZ|DAZ (orc_enter()),
 * mirroring FPSCR.FZ on the RP2350 (main.c:593-600).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <xmmintrin.h>

#if ORC_USE_REF
#include "config.h"
#include "dsp_pipeline.h"
#include "loudness.h"
#include "crossfeed.h"
#include "leveller.h"
#include "bulk_params.h"
#include "usb_audio.h"
#define LEAF(n) n
#ifndef PRESET_MUTE_SAMPLES
#define PRESET_MUTE_SAMPLES 256
#endif
#if PICO_RP2350
typedef float orc_sample;
#else
typedef int32_t orc_sample;
#endif
extern int orc_math_mode;
#else
#include "orc_types.h"
#include "orc_leaf.h"
#define LEAF(n) orc_##n
#endif
#include "orc_common.h"
#include "orc_api.h"

int orc_x86_cast_semantics = 0;

/* Float contract of the code restated in this file (see orc_leaf.c for the rule): the four places where the firmware's
 * compiler fuses a multiply into an add outside the leaf functions — the loudness SVF (usb_audio.c:697-712), the matrix mix
 * (:766), the sub-alignment term of dsp_update_delay_samples (dsp_pipeline.c:227-228) and the Taylor dB->linear of the bulk
 * path (bulk_params.c:54).  Pinned by tests/test_oracle_vs_fw.py against usb_audio.c compiled with contraction on. */
extern int orc_fma_mode;
#if PICO_RP2350
#define OMAD(a, b, c) (orc_fma_mode ? fmaf((a), (b), (c)) : (a) * (b) + (c))
#else
#define OMAD(a, b, c) ((a) * (b) + (c))
#endif

#if PICO_RP2350
_Static_assert(sizeof(Biquad) == 68, "float Biquad layout");
#else
_Static_assert(sizeof(Biquad) == 32, "Q28 Biquad layout (asm stride, dsp_process_rp2040.S:14)");
#endif
_Static_assert(sizeof(WireBulkParams) == 2896, "wire blob size (bulk_params.h:205)");
_Static_assert(sizeof(EqParamPacket) == 16, "EqParamPacket");
_Static_assert(sizeof(MatrixRoutePacket) == 8, "MatrixRoutePacket");

/* ------------------------------------------------------------------------------------- */
/* Preset slot image (flash_storage.c:78-92, :136-189) — lives in a .c file upstream       */
/* ------------------------------------------------------------------------------------- */
#define SLOT_MAGIC 0x44535033u
#define SLOT_DATA_VERSION 12
typedef struct __attribute__((packed)) { uint8_t enabled, phase_invert, reserved[2]; float gain_db; } SlotCrosspoint;
typedef struct __attribute__((packed)) { uint8_t enabled, mute, reserved[2]; float gain_db, delay_ms; } SlotOutput;
typedef struct __attribute__((packed)) {
    uint32_t magic; uint16_t version; uint16_t slot_index; uint32_t crc32;
    EqParamPacket filter_recipes[NUM_CHANNELS][MAX_BANDS];
    float preamp_db; uint8_t bypass; uint8_t padding[3];
    float delays_ms[NUM_CHANNELS];
    float channel_gain_db[3]; uint8_t channel_mute[3]; uint8_t padding2;
    uint8_t loudness_enabled; uint8_t padding3[3]; float loudness_ref_spl, loudness_intensity_pct;
    uint8_t crossfeed_enabled, crossfeed_preset, crossfeed_itd_enabled, padding4;
    float crossfeed_custom_fc, crossfeed_custom_feed_db;
    SlotCrosspoint matrix_crosspoints[NUM_INPUT_CHANNELS][NUM_OUTPUT_CHANNELS];
    SlotOutput matrix_outputs[NUM_OUTPUT_CHANNELS];
    uint8_t output_pins[NUM_PIN_OUTPUTS]; uint8_t pin_padding[8 - NUM_PIN_OUTPUTS];
    char channel_names[NUM_CHANNELS][PRESET_NAME_LEN];
    uint8_t output_types[4]; uint8_t i2s_bck_pin, i2s_mck_pin, i2s_mck_enabled, i2s_mck_multiplier;
    uint8_t leveller_enabled, leveller_speed, leveller_lookahead, leveller_padding;
    float leveller_amount, leveller_max_gain_db, leveller_gate_threshold_db;
    float preamp_db_per_ch[NUM_INPUT_CHANNELS];
    float master_volume_db;
} OrcPresetSlot;
#if PICO_RP2350
_Static_assert(sizeof(OrcPresetSlot) == 2864, "PresetSlot size (float flavour)");
#else
_Static_assert(sizeof(OrcPresetSlot) == 1840, "PresetSlot size (Q28 flavour)");
#endif

/* ------------------------------------------------------------------------------------- */
/* Context = the firmware's file-scope globals, one copy per stream                       */
/* ------------------------------------------------------------------------------------- */
typedef struct { uint32_t freq; int16_t volume; int16_t vol_mul; bool mute; } OrcAudioState; /* usb_audio.h:15-20 */

struct orc_ctx {
    /* dsp_pipeline.c:19-34 */
    EqParamPacket filter_recipes[NUM_CHANNELS][MAX_BANDS];
    Biquad filters[NUM_CHANNELS][MAX_BANDS];
    float channel_delays_ms[NUM_CHANNELS];
    bool channel_bypassed[NUM_CHANNELS];
    int32_t channel_delay_samples[NUM_DELAY_CHANNELS];
    bool any_delay_active;
    uint32_t delay_write_idx;
    orc_sample delay_lines[NUM_DELAY_CHANNELS][MAX_DELAY_SAMPLES];
    /* usb_audio.c:47-214 */
    OrcAudioState audio_state;
    bool bypass_master_eq;
    float global_preamp_db[NUM_INPUT_CHANNELS];
    int32_t global_preamp_mul[NUM_INPUT_CHANNELS];
    float global_preamp_linear[NUM_INPUT_CHANNELS];
    float master_volume_db, master_volume_linear;
    int32_t master_volume_q15;
    float channel_gain_db[3]; int32_t channel_gain_mul[3]; float channel_gain_linear[3]; bool channel_mute[3];
    MatrixMixer matrix_mixer;
    bool loudness_enabled; float loudness_ref_spl, loudness_intensity_pct;
    LoudnessCoeffs loudness_table[LOUDNESS_VOL_STEPS][LOUDNESS_BIQUAD_COUNT];
    bool loudness_table_valid;          /* loudness_active_table != NULL */
    int loudness_row;                   /* current_loudness_coeffs: row index, -1 = NULL */
#if PICO_RP2350
    LoudnessSvfState loudness_state[2][LOUDNESS_BIQUAD_COUNT];
#else
    Biquad loudness_biquads[2][LOUDNESS_BIQUAD_COUNT];
#endif
    CrossfeedConfig crossfeed_config; bool crossfeed_bypassed; CrossfeedState crossfeed_state;
    LevellerConfig leveller_config; bool leveller_bypassed; LevellerCoeffs leveller_coeffs; LevellerState leveller_state;
    char channel_names[NUM_CHANNELS][PRESET_NAME_LEN];
    uint8_t output_pins[NUM_PIN_OUTPUTS];
    uint8_t output_types[NUM_SPDIF_INSTANCES];
    uint8_t i2s_bck_pin, i2s_mck_pin; bool i2s_mck_enabled; uint16_t i2s_mck_multiplier;
    Core1Mode core1_mode;
    /* deferred-apply flags */
    bool loudness_recompute_pending, crossfeed_update_pending, leveller_update_pending, leveller_reset_pending;
    /* preset mute (flash_storage.c:255-256, usb_audio.c:457) */
    bool preset_loading; uint32_t preset_mute_counter; float preset_mute_smooth_gain;
    /* the slice of the preset directory that steers master volume (flash_storage.c:113-131) */
    uint8_t dir_master_volume_mode, dir_include_pins; float dir_master_volume_db;
    SystemStatusPacket status;
};

static const uint8_t band_counts[NUM_CHANNELS] = {
#if PICO_RP2350
    10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10
#else
    10, 10, 10, 10, 10, 10, 10
#endif
}; /* dsp_pipeline.c:36-44 */

/* ------------------------------------------------------------------------------------- */
/* FTZ|DAZ bracket                                                                         */
/* ------------------------------------------------------------------------------------- */
static unsigned orc_enter(void) { unsigned c = _mm_getcsr(); _mm_setcsr(c | 0x8040u); return c; }
static void orc_leave(unsigned c) { _mm_setcsr(c); }

/* ------------------------------------------------------------------------------------- */
/* _ref build: shuttle the context through the reference's globals for the few reference   */
/* functions that only work on globals                                                     */
/* ------------------------------------------------------------------------------------- */
#if ORC_USE_REF
extern uint8_t output_pins[NUM_PIN_OUTPUTS];
extern uint8_t output_types[];
extern uint8_t i2s_bck_pin, i2s_mck_pin; extern bool i2s_mck_enabled; extern uint16_t i2s_mck_multiplier;
extern MatrixMixer matrix_mixer;
extern volatile LevellerConfig leveller_config;
extern volatile bool leveller_update_pending, leveller_reset_pending;
#define CP_OUT(g, f) memcpy((void *)(g), (f), sizeof(f))
#define CP_IN(f, g) memcpy((f), (const void *)(g), sizeof(f))
static void ref_push(const orc_ctx *c) {
    CP_OUT(filter_recipes, c->filter_recipes); CP_OUT(filters, c->filters);
    CP_OUT(channel_delays_ms, c->channel_delays_ms); CP_OUT(channel_bypassed, c->channel_bypassed);
    CP_OUT(channel_delay_samples, c->channel_delay_samples); any_delay_active = c->any_delay_active;
    bypass_master_eq = c->bypass_master_eq;
    CP_OUT(global_preamp_db, c->global_preamp_db); CP_OUT(global_preamp_mul, c->global_preamp_mul);
    CP_OUT(global_preamp_linear, c->global_preamp_linear);
    master_volume_db = c->master_volume_db; master_volume_linear = c->master_volume_linear; master_volume_q15 = c->master_volume_q15;
    CP_OUT(channel_gain_db, c->channel_gain_db); CP_OUT(channel_gain_mul, c->channel_gain_mul);
    CP_OUT(channel_gain_linear, c->channel_gain_linear); CP_OUT(channel_mute, c->channel_mute);
    matrix_mixer = c->matrix_mixer;
    loudness_enabled = c->loudness_enabled; loudness_ref_spl = c->loudness_ref_spl; loudness_intensity_pct = c->loudness_intensity_pct;
    loudness_recompute_pending = c->loudness_recompute_pending;
    memcpy((void *)&crossfeed_config, &c->crossfeed_config, sizeof(CrossfeedConfig)); crossfeed_update_pending = c->crossfeed_update_pending;
    memcpy((void *)&leveller_config, &c->leveller_config, sizeof(LevellerConfig));
    leveller_update_pending = c->leveller_update_pending; leveller_reset_pending = c->leveller_reset_pending;
    CP_OUT(channel_names, c->channel_names); CP_OUT(output_pins, c->output_pins);
    memcpy(output_types, c->output_types, NUM_SPDIF_INSTANCES);
    i2s_bck_pin = c->i2s_bck_pin; i2s_mck_pin = c->i2s_mck_pin; i2s_mck_enabled = c->i2s_mck_enabled; i2s_mck_multiplier = c->i2s_mck_multiplier;
}
static void ref_pull(orc_ctx *c) {
    CP_IN(c->filter_recipes, filter_recipes); CP_IN(c->filters, filters);
    CP_IN(c->channel_delays_ms, channel_delays_ms); CP_IN(c->channel_bypassed, channel_bypassed);
    CP_IN(c->channel_delay_samples, channel_delay_samples); c->any_delay_active = any_delay_active;
    c->bypass_master_eq = bypass_master_eq;
    CP_IN(c->global_preamp_db, global_preamp_db); CP_IN(c->global_preamp_mul, global_preamp_mul);
    CP_IN(c->global_preamp_linear, global_preamp_linear);
    c->master_volume_db = master_volume_db; c->master_volume_linear = master_volume_linear; c->master_volume_q15 = master_volume_q15;
    CP_IN(c->channel_gain_db, channel_gain_db); CP_IN(c->channel_gain_mul, channel_gain_mul);
    CP_IN(c->channel_gain_linear, channel_gain_linear); CP_IN(c->channel_mute, channel_mute);
    c->matrix_mixer = matrix_mixer;
    c->loudness_enabled = loudness_enabled; c->loudness_ref_spl = loudness_ref_spl; c->loudness_intensity_pct = loudness_intensity_pct;
    c->loudness_recompute_pending = loudness_recompute_pending;
    memcpy(&c->crossfeed_config, (const void *)&crossfeed_config, sizeof(CrossfeedConfig)); c->crossfeed_update_pending = crossfeed_update_pending;
    memcpy(&c->leveller_config, (const void *)&leveller_config, sizeof(LevellerConfig));
    c->leveller_update_pending = leveller_update_pending; c->leveller_reset_pending = leveller_reset_pending;
    CP_IN(c->channel_names, channel_names); CP_IN(c->output_pins, output_pins);
    memcpy(c->output_types, output_types, NUM_SPDIF_INSTANCES);
    c->i2s_bck_pin = i2s_bck_pin; c->i2s_mck_pin = i2s_mck_pin; c->i2s_mck_enabled = i2s_mck_enabled; c->i2s_mck_multiplier = i2s_mck_multiplier;
}
#endif

/* ------------------------------------------------------------------------------------- */
/* Small restated helpers                                                                  */
/* ------------------------------------------------------------------------------------- */
#if !PICO_RP2350
static inline int32_t qmul(int32_t a, int32_t b) { return LEAF(fast_mul_q28)(a, b); }
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wabs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; } /* abs(INT_MIN) stays INT_MIN */

/* dsp_process_rp2040.S:225-394 — TDF2 cascade, five inlined Q28 multiplies per sample, band-major. */
static void q28_biquad_block(Biquad *bands, int32_t *x, uint32_t count, uint8_t nbands) {
    for (int b = 0; b < nbands; b++) {
        Biquad *q = &bands[b];
        if (q->bypass) continue;                        /* ldrb [r0,#28] */
        int32_t s1 = q->s1, s2 = q->s2;
        for (uint32_t i = 0; i < count; i++) {
            int32_t in = x[i];
            int32_t y = wadd(qmul(q->b0, in), s1);       /* :272-285 */
            int32_t t1 = qmul(q->b1, in);                /* :287-298 */
            int32_t t3 = qmul(q->b2, in);                /* :300-312 */
            int32_t t2 = qmul(q->a1, y);                 /* :318-329 */
            s1 = wadd(wsub(t1, t2), s2);                 /* :331-335 */
            int32_t t4 = qmul(q->a2, y);                 /* :337-348 */
            s2 = wsub(t3, t4);                           /* :350-353 */
            x[i] = y;
        }
        q->s1 = s1; q->s2 = s2;
    }
}
#endif

#if !PICO_RP2350
/* test hook: the Q28 block biquad on caller-supplied bands (coef [nbands][5] = b0 b1 b2 a1 a2, state [nbands][2] in/out), so that it can be
 * compared with the reference's assembly executed instruction by instruction (tests/thumb.py, tests/test_oracle_thumb.py) */
void orc_debug_q28_biquad_block(const int32_t *coef, int32_t *state, const uint8_t *bypass, int32_t *x, uint32_t count, uint32_t nbands) {
    Biquad bands[MAX_BANDS];
    if (nbands > MAX_BANDS) nbands = MAX_BANDS;
    memset(bands, 0, sizeof bands);
    for (uint32_t b = 0; b < nbands; b++) {
        bands[b].b0 = coef[b * 5]; bands[b].b1 = coef[b * 5 + 1]; bands[b].b2 = coef[b * 5 + 2]; bands[b].a1 = coef[b * 5 + 3]; bands[b].a2 = coef[b * 5 + 4];
        bands[b].s1 = state[b * 2]; bands[b].s2 = state[b * 2 + 1]; bands[b].bypass = bypass[b] != 0;
    }
    q28_biquad_block(bands, x, count, (uint8_t)nbands);
    for (uint32_t b = 0; b < nbands; b++) { state[b * 2] = bands[b].s1; state[b * 2 + 1] = bands[b].s2; }
}
#endif

static void eq_block(orc_ctx *c, int ch, orc_sample *x, uint32_t n) {
#if PICO_RP2350
#if ORC_USE_REF
    dsp_process_channel_block(c->filters[ch], x, n, (uint8_t)ch);
#else
    orc_dsp_process_channel_block(c->filters[ch], x, n, band_counts[ch]);
#endif
#else
    q28_biquad_block(c->filters[ch], x, n, band_counts[ch]);
#endif
}

static float db_to_linear_powf(float db) { /* flash_storage.c:302-306 */
    if (db <= -120.0f) return 0.0f;
    if (db >= +80.0f) db = 80.0f;
    return powf(10.0f, db / 20.0f);
}

static void update_preamp(orc_ctx *c, uint8_t ch, float db) { /* usb_audio.c:244-250 */
    if (!isfinite(db)) return;
    c->global_preamp_db[ch] = db;
    float lin = powf(10.0f, db / 20.0f);
    c->global_preamp_mul[ch] = orc_f2i(lin * (float)(1 << 28));
    c->global_preamp_linear[ch] = lin;
}

static void set_master_volume_clamped(orc_ctx *c, float db) { /* shared tail of :255-269 and flash_storage.c:558-571 */
    if (db < MASTER_VOL_MUTE_DB) db = MASTER_VOL_MUTE_DB;
    if (db > MASTER_VOL_MAX_DB) db = MASTER_VOL_MAX_DB;
    c->master_volume_db = db;
    if (db <= MASTER_VOL_MUTE_DB) { c->master_volume_linear = 0.0f; c->master_volume_q15 = 0; }
    else {
        float lin = powf(10.0f, db / 20.0f);
        c->master_volume_linear = lin;
        c->master_volume_q15 = orc_f2i(lin * 32768.0f);
    }
}
static void update_master_volume(orc_ctx *c, float db) { if (!isfinite(db)) return; set_master_volume_clamped(c, db); }
static void apply_master_volume_db(orc_ctx *c, float db) { if (!isfinite(db)) db = MASTER_VOL_MAX_DB; set_master_volume_clamped(c, db); }

/* usb_audio.c:409-420 — UAC1 volume table; entry 60 (0 dB) is 0x8000 and is stored in an int16_t */
static const uint16_t db_to_vol[61] = {
    0x0000, 0x0025, 0x0029, 0x002e, 0x0034, 0x003a, 0x0041, 0x0049, 0x0052, 0x005c, 0x0068, 0x0074, 0x0082, 0x0092, 0x00a4, 0x00b8,
    0x00cf, 0x00e8, 0x0104, 0x0124, 0x0148, 0x0170, 0x019d, 0x01cf, 0x0207, 0x0247, 0x028e, 0x02de, 0x0337, 0x039c, 0x040c, 0x048b,
    0x0519, 0x05b8, 0x066a, 0x0733, 0x0814, 0x0910, 0x0a2b, 0x0b68, 0x0ccd, 0x0e5d, 0x101d, 0x1215, 0x1449, 0x16c3, 0x198a, 0x1ca8,
    0x2027, 0x2413, 0x287a, 0x2d6b, 0x32f5, 0x392d, 0x4027, 0x47fb, 0x50c3, 0x5a9e, 0x65ad, 0x7215, 0x8000};

static uint8_t volume_index(int16_t volume) { /* :430-433 */
    volume = (int16_t)(volume + 60 * 256);
    if (volume < 0) volume = 0;
    if (volume >= 61 * 256) volume = 61 * 256 - 1;
    return (uint8_t)(((uint16_t)volume) >> 8u);
}

static void set_volume_(orc_ctx *c, int16_t volume) { /* :428-440 */
    c->audio_state.volume = volume;
    uint8_t idx = volume_index(volume);
    c->audio_state.vol_mul = (int16_t)db_to_vol[idx];      /* 0x8000 -> -32768: the sign quirk */
    if (c->loudness_enabled && c->loudness_table_valid) c->loudness_row = idx;
}

static Core1Mode derive_core1_mode_(const orc_ctx *c) { /* :1620-1630 */
    if (c->matrix_mixer.outputs[NUM_OUTPUT_CHANNELS - 1].enabled) return CORE1_MODE_PDM;
    for (int o = CORE1_EQ_FIRST_OUTPUT; o <= CORE1_EQ_LAST_OUTPUT; o++)
        if (c->matrix_mixer.outputs[o].enabled) return CORE1_MODE_EQ_WORKER;
    return CORE1_MODE_IDLE;
}

static void update_delay_samples(orc_ctx *c, float fs) { /* dsp_pipeline.c:216-239 */
#if ORC_USE_REF
    ref_push(c); dsp_update_delay_samples(fs); ref_pull(c);
#else
    c->any_delay_active = false;
    for (int o = 0; o < NUM_DELAY_CHANNELS; o++) {
        float ms = c->channel_delays_ms[CH_OUT_1 + o];
        if (o == NUM_DELAY_CHANNELS - 1) {
            ms = OMAD((float)SUB_ALIGN_SAMPLES / fs, 1000.0f, ms);      /* ms += 128 / fs * 1000 */
        }
        int32_t s = orc_f2i(ms * fs / 1000.0f);
        if (s > MAX_DELAY_SAMPLES) s = MAX_DELAY_SAMPLES;
        if (s < 0) s = 0;
        c->channel_delay_samples[o] = s;
        if (s > 0) c->any_delay_active = true;
    }
#endif
}

static void recalc_channel_bypass(orc_ctx *c, int ch) { /* main.c:846-854 */
    bool all = true;
    for (int b = 0; b < band_counts[ch]; b++) if (!c->filters[ch][b].bypass) { all = false; break; }
    c->channel_bypassed[ch] = all;
}

static void recalculate_all_filters(orc_ctx *c, float fs) { /* dsp_pipeline.c:241-253 */
#if ORC_USE_REF
    ref_push(c); dsp_recalculate_all_filters(fs); ref_pull(c);
#else
    update_delay_samples(c, fs);
    for (int ch = 0; ch < NUM_CHANNELS; ch++) {
        for (int b = 0; b < band_counts[ch]; b++)
            orc_dsp_compute_coefficients(&c->filter_recipes[ch][b], &c->filters[ch][b], fs);
        recalc_channel_bypass(c, ch);
    }
#endif
}

static void init_default_filters(orc_ctx *c) { /* dsp_pipeline.c:177-214 */
#if ORC_USE_REF
    ref_push(c); dsp_init_default_filters(); ref_pull(c);
#else
    memset(c->filters, 0, sizeof(c->filters));
    memset(c->channel_delays_ms, 0, sizeof(c->channel_delays_ms));
    for (int ch = 0; ch < NUM_CHANNELS; ch++) {
        c->channel_bypassed[ch] = true;
        for (int b = 0; b < MAX_BANDS; b++) {
            Biquad *q = &c->filters[ch][b];
            q->bypass = true;
#if PICO_RP2350
            q->b0 = 1.0f; q->use_svf = false; q->svf_type = FILTER_FLAT; q->svic1eq = q->svic2eq = 0.0f;
#else
            q->b0 = 1 << FILTER_SHIFT;
#endif
            EqParamPacket *r = &c->filter_recipes[ch][b];
            r->type = FILTER_FLAT; r->freq = 1000.0f; r->Q = 0.707f; r->gain_db = 0.0f;
        }
    }
    for (int ch = CH_OUT_1; ch < CH_OUT_SUB; ch++) {   /* 80 Hz high-pass on every S/PDIF output */
        EqParamPacket hp; memset(&hp, 0, sizeof hp);
        hp.type = FILTER_HIGHPASS; hp.freq = 80.0f; hp.Q = 0.707f; hp.gain_db = 0.0f;
        c->filter_recipes[ch][0] = hp;
    }
    EqParamPacket lp; memset(&lp, 0, sizeof lp);     /* 80 Hz low-pass on the sub */
    lp.type = FILTER_LOWPASS; lp.freq = 80.0f; lp.Q = 0.707f; lp.gain_db = 0.0f;
    c->filter_recipes[CH_OUT_SUB][0] = lp;
#endif
}

static void default_channel_name(int ch, char *buf) { /* usb_audio.c:216-235 */
    static const char *names[] = {
#if PICO_RP2350
        "USB L", "USB R", "SPDIF 1 L", "SPDIF 1 R", "SPDIF 2 L", "SPDIF 2 R", "SPDIF 3 L", "SPDIF 3 R", "SPDIF 4 L", "SPDIF 4 R", "PDM"
#else
        "USB L", "USB R", "SPDIF 1 L", "SPDIF 1 R", "SPDIF 2 L", "SPDIF 2 R", "PDM"
#endif
    };
    memset(buf, 0, PRESET_NAME_LEN);
    if (ch >= 0 && ch < NUM_CHANNELS) strncpy(buf, names[ch], PRESET_NAME_LEN - 1);
}

static void loudness_recompute(orc_ctx *c, float fs) {
#if ORC_USE_REF
    loudness_recompute_table(c->loudness_ref_spl, c->loudness_intensity_pct, fs);
    memcpy(c->loudness_table, loudness_active_table, sizeof(c->loudness_table));
#else
    orc_loudness_build_table(c->loudness_table, c->loudness_ref_spl, c->loudness_intensity_pct, fs);
#endif
    c->loudness_table_valid = true;
}

/* ------------------------------------------------------------------------------------- */
/* Deferred-apply dispatcher (main loop body)                                              */
/* ------------------------------------------------------------------------------------- */
static void service(orc_ctx *c) { /* main.c:867-894 */
    float fs = (float)c->audio_state.freq;
    if (c->loudness_recompute_pending) {
        c->loudness_recompute_pending = false;
        loudness_recompute(c, fs);
        if (c->loudness_enabled && c->loudness_table_valid) set_volume_(c, c->audio_state.volume);
    }
    if (c->crossfeed_update_pending) {
        c->crossfeed_update_pending = false;
        LEAF(crossfeed_compute_coefficients)(&c->crossfeed_state, &c->crossfeed_config, fs);
        c->crossfeed_bypassed = !c->crossfeed_config.enabled;
    }
    if (c->leveller_update_pending) {
        c->leveller_update_pending = false;
        LEAF(leveller_compute_coefficients)(&c->leveller_coeffs, &c->leveller_config, fs);
        if (c->leveller_reset_pending) { c->leveller_reset_pending = false; LEAF(leveller_reset_state)(&c->leveller_state); }
        c->leveller_bypassed = !c->leveller_config.enabled;
    }
}

static void perform_rate_change(orc_ctx *c, uint32_t freq) { /* main.c:132-171 (DSP part) */
    recalculate_all_filters(c, (float)freq);
    c->loudness_recompute_pending = true;
    c->crossfeed_update_pending = true;
    c->leveller_update_pending = true;
}

static void prepare_pipeline_reset(orc_ctx *c, uint32_t mute_samples) { /* main.c:449-458 */
    c->preset_mute_counter = mute_samples;
    c->preset_loading = true;
}

static void transition_core1(orc_ctx *c) { c->core1_mode = derive_core1_mode_(c); }
static uint32_t flash_mute_hold_samples_(uint32_t freq) { /* flash_storage.c:262-266 */
    uint64_t samples = ((uint64_t)freq * 10u + 999u) / 1000u;
    if (samples < 512u) samples = 512u;
    return (uint32_t)samples;
}

/* ------------------------------------------------------------------------------------- */
/* Factory defaults / presets / bulk blobs                                                 */
/* ------------------------------------------------------------------------------------- */
static void apply_master_volume_from_mode(orc_ctx *c, const OrcPresetSlot *s) { /* flash_storage.c:580-589 */
    float db;
    if (c->dir_master_volume_mode == MASTER_VOLUME_MODE_WITH_PRESET && s && s->version >= 12) db = s->master_volume_db;
    else db = c->dir_master_volume_db;
    apply_master_volume_db(c, db);
}

static void leveller_config_defaults(LevellerConfig *l) {
    l->enabled = LEVELLER_DEFAULT_ENABLED; l->amount = LEVELLER_DEFAULT_AMOUNT; l->speed = LEVELLER_DEFAULT_SPEED;
    l->max_gain_db = LEVELLER_DEFAULT_MAX_GAIN_DB; l->lookahead = LEVELLER_DEFAULT_LOOKAHEAD;
    l->gate_threshold_db = LEVELLER_DEFAULT_GATE_DB;
}

static void default_pins(uint8_t *pins) {
    pins[0] = PICO_AUDIO_SPDIF_PIN; pins[1] = PICO_SPDIF_PIN_2;
#if PICO_RP2350
    pins[2] = PICO_SPDIF_PIN_3; pins[3] = PICO_SPDIF_PIN_4; pins[4] = PICO_PDM_PIN;
#else
    pins[2] = PICO_PDM_PIN;
#endif
}

static void apply_factory_defaults(orc_ctx *c) { /* flash_storage.c:1144-1238 */
    init_default_filters(c);
    for (int i = 0; i < NUM_INPUT_CHANNELS; i++) {
        c->global_preamp_db[i] = 0.0f; c->global_preamp_mul[i] = 1 << 28; c->global_preamp_linear[i] = 1.0f;
    }
    apply_master_volume_from_mode(c, NULL);
    c->bypass_master_eq = false;
    for (int i = 0; i < 3; i++) { c->channel_gain_db[i] = 0.0f; c->channel_gain_mul[i] = 32768; c->channel_mute[i] = false; }
    c->loudness_enabled = false; c->loudness_ref_spl = 83.0f; c->loudness_intensity_pct = 100.0f;
    c->loudness_recompute_pending = true;
    c->crossfeed_config.enabled = false; c->crossfeed_config.itd_enabled = true;
    c->crossfeed_config.preset = CROSSFEED_PRESET_DEFAULT;
    c->crossfeed_config.custom_fc = 700.0f; c->crossfeed_config.custom_feed_db = 4.5f;
    c->crossfeed_update_pending = true;
    memset(&c->matrix_mixer, 0, sizeof(c->matrix_mixer));
    c->matrix_mixer.crosspoints[0][0].enabled = 1; c->matrix_mixer.crosspoints[0][0].gain_linear = 1.0f;
    c->matrix_mixer.crosspoints[1][1].enabled = 1; c->matrix_mixer.crosspoints[1][1].gain_linear = 1.0f;
    for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
        c->matrix_mixer.outputs[o].enabled = (o < 2) ? 1 : 0;
        c->matrix_mixer.outputs[o].gain_linear = 1.0f;
    }
    default_pins(c->output_pins);
    for (int ch = 0; ch < NUM_CHANNELS; ch++) default_channel_name(ch, c->channel_names[ch]);
    memset(c->output_types, 0, sizeof(c->output_types));
    c->i2s_bck_pin = PICO_I2S_BCK_PIN; c->i2s_mck_pin = PICO_I2S_MCK_PIN; c->i2s_mck_enabled = false; c->i2s_mck_multiplier = 128;
    leveller_config_defaults(&c->leveller_config);
    c->leveller_update_pending = true; c->leveller_reset_pending = true;
}

static uint32_t crc32_(const uint8_t *d, size_t n) { /* flash_storage.c:282-291 */
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        crc ^= d[i];
        for (int j = 0; j < 8; j++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

static bool pin_valid(uint8_t pin) {
    bool v = (pin <= 29) && (pin != 12) && !(pin >= 23 && pin <= 25);
#if !PICO_RP2350
    if (pin > 28) v = false;
#endif
    return v;
}

static void apply_slot_to_live(orc_ctx *c, const OrcPresetSlot *s, bool include_pins) { /* flash_storage.c:597-742 */
    memcpy(c->filter_recipes, s->filter_recipes, sizeof(c->filter_recipes));
    for (int i = 0; i < NUM_INPUT_CHANNELS; i++) {
        float db = (s->version >= 12) ? s->preamp_db_per_ch[i] : s->preamp_db;
        c->global_preamp_db[i] = db;
        float lin = db_to_linear_powf(db);
        c->global_preamp_mul[i] = orc_f2i(lin * (float)(1 << 28));
        c->global_preamp_linear[i] = lin;
    }
    c->bypass_master_eq = (s->bypass != 0);
    memcpy(c->channel_delays_ms, s->delays_ms, sizeof(c->channel_delays_ms));
    for (int i = 0; i < 3; i++) {
        c->channel_gain_db[i] = s->channel_gain_db[i];
        float g = db_to_linear_powf(s->channel_gain_db[i]);
        c->channel_gain_mul[i] = orc_f2i(g * 32768.0f);
        c->channel_mute[i] = (s->channel_mute[i] != 0);
    }
    c->loudness_enabled = (s->loudness_enabled != 0);
    c->loudness_ref_spl = s->loudness_ref_spl; c->loudness_intensity_pct = s->loudness_intensity_pct;
    c->loudness_recompute_pending = true;
    c->crossfeed_config.enabled = (s->crossfeed_enabled != 0);
    c->crossfeed_config.preset = s->crossfeed_preset;
    c->crossfeed_config.itd_enabled = (s->crossfeed_itd_enabled != 0);
    c->crossfeed_config.custom_fc = s->crossfeed_custom_fc; c->crossfeed_config.custom_feed_db = s->crossfeed_custom_feed_db;
    c->crossfeed_update_pending = true;
    for (int in = 0; in < NUM_INPUT_CHANNELS; in++)
        for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
            MatrixCrosspoint *xp = &c->matrix_mixer.crosspoints[in][o];
            xp->enabled = s->matrix_crosspoints[in][o].enabled;
            xp->phase_invert = s->matrix_crosspoints[in][o].phase_invert;
            xp->gain_db = s->matrix_crosspoints[in][o].gain_db;
            xp->gain_linear = db_to_linear_powf(s->matrix_crosspoints[in][o].gain_db);
        }
    for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
        OutputChannel *oc = &c->matrix_mixer.outputs[o];
        oc->enabled = s->matrix_outputs[o].enabled; oc->mute = s->matrix_outputs[o].mute;
        oc->gain_db = s->matrix_outputs[o].gain_db;
        oc->gain_linear = db_to_linear_powf(s->matrix_outputs[o].gain_db);
        oc->delay_ms = s->matrix_outputs[o].delay_ms;
        c->channel_delays_ms[CH_OUT_1 + o] = s->matrix_outputs[o].delay_ms;   /* overrides the delays_ms entry */
    }
    if (include_pins) {
        uint8_t def[NUM_PIN_OUTPUTS]; default_pins(def);
        for (int i = 0; i < NUM_PIN_OUTPUTS; i++) c->output_pins[i] = pin_valid(s->output_pins[i]) ? s->output_pins[i] : def[i];
    }
    if (s->version >= 8) memcpy(c->channel_names, s->channel_names, sizeof(c->channel_names));
    else for (int ch = 0; ch < NUM_CHANNELS; ch++) default_channel_name(ch, c->channel_names[ch]);
    if (s->version >= 9) {
        memcpy(c->output_types, s->output_types, NUM_SPDIF_INSTANCES);
        c->i2s_bck_pin = s->i2s_bck_pin; c->i2s_mck_pin = s->i2s_mck_pin; c->i2s_mck_enabled = (s->i2s_mck_enabled != 0);
        if (s->version >= 11) c->i2s_mck_multiplier = (s->i2s_mck_multiplier == 1) ? 256 : 128;
        else c->i2s_mck_multiplier = (s->i2s_mck_multiplier == 0) ? 256 : s->i2s_mck_multiplier;
    } else {
        memset(c->output_types, 0, NUM_SPDIF_INSTANCES);
        c->i2s_bck_pin = PICO_I2S_BCK_PIN; c->i2s_mck_pin = PICO_I2S_MCK_PIN; c->i2s_mck_enabled = false; c->i2s_mck_multiplier = 128;
    }
    if (s->version >= 10) {
        c->leveller_config.enabled = (s->leveller_enabled != 0); c->leveller_config.speed = s->leveller_speed;
        c->leveller_config.lookahead = (s->leveller_lookahead != 0); c->leveller_config.amount = s->leveller_amount;
        c->leveller_config.max_gain_db = s->leveller_max_gain_db; c->leveller_config.gate_threshold_db = s->leveller_gate_threshold_db;
    } else leveller_config_defaults(&c->leveller_config);
    c->leveller_update_pending = true; c->leveller_reset_pending = true;
}

static void collect_live_state(const orc_ctx *c, OrcPresetSlot *s, uint8_t slot_index) { /* flash_storage.c:464-552 */
    memset(s, 0, sizeof(*s));
    s->magic = SLOT_MAGIC; s->version = SLOT_DATA_VERSION; s->slot_index = slot_index;
    memcpy(s->filter_recipes, c->filter_recipes, sizeof(s->filter_recipes));
    s->preamp_db = c->global_preamp_db[0];
    s->bypass = c->bypass_master_eq ? 1 : 0;
    memcpy(s->delays_ms, c->channel_delays_ms, sizeof(s->delays_ms));
    memcpy(s->channel_gain_db, c->channel_gain_db, sizeof(s->channel_gain_db));
    for (int i = 0; i < 3; i++) s->channel_mute[i] = c->channel_mute[i] ? 1 : 0;
    s->loudness_enabled = c->loudness_enabled ? 1 : 0;
    s->loudness_ref_spl = c->loudness_ref_spl; s->loudness_intensity_pct = c->loudness_intensity_pct;
    s->crossfeed_enabled = c->crossfeed_config.enabled ? 1 : 0; s->crossfeed_preset = c->crossfeed_config.preset;
    s->crossfeed_itd_enabled = c->crossfeed_config.itd_enabled ? 1 : 0;
    s->crossfeed_custom_fc = c->crossfeed_config.custom_fc; s->crossfeed_custom_feed_db = c->crossfeed_config.custom_feed_db;
    for (int in = 0; in < NUM_INPUT_CHANNELS; in++)
        for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
            s->matrix_crosspoints[in][o].enabled = c->matrix_mixer.crosspoints[in][o].enabled;
            s->matrix_crosspoints[in][o].phase_invert = c->matrix_mixer.crosspoints[in][o].phase_invert;
            s->matrix_crosspoints[in][o].gain_db = c->matrix_mixer.crosspoints[in][o].gain_db;
        }
    for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
        s->matrix_outputs[o].enabled = c->matrix_mixer.outputs[o].enabled; s->matrix_outputs[o].mute = c->matrix_mixer.outputs[o].mute;
        s->matrix_outputs[o].gain_db = c->matrix_mixer.outputs[o].gain_db; s->matrix_outputs[o].delay_ms = c->matrix_mixer.outputs[o].delay_ms;
    }
    memcpy(s->output_pins, c->output_pins, sizeof(s->output_pins));
    memcpy(s->channel_names, c->channel_names, sizeof(s->channel_names));
    memcpy(s->output_types, c->output_types, NUM_SPDIF_INSTANCES);
    s->i2s_bck_pin = c->i2s_bck_pin; s->i2s_mck_pin = c->i2s_mck_pin; s->i2s_mck_enabled = c->i2s_mck_enabled ? 1 : 0;
    s->i2s_mck_multiplier = (c->i2s_mck_multiplier == 256) ? 1 : 0;
    s->leveller_enabled = c->leveller_config.enabled ? 1 : 0; s->leveller_speed = c->leveller_config.speed;
    s->leveller_lookahead = c->leveller_config.lookahead ? 1 : 0; s->leveller_amount = c->leveller_config.amount;
    s->leveller_max_gain_db = c->leveller_config.max_gain_db; s->leveller_gate_threshold_db = c->leveller_config.gate_threshold_db;
    for (int i = 0; i < NUM_INPUT_CHANNELS; i++) s->preamp_db_per_ch[i] = c->global_preamp_db[i];
    s->master_volume_db = c->master_volume_db;
    s->crc32 = crc32_((const uint8_t *)&s->filter_recipes, sizeof(OrcPresetSlot) - offsetof(OrcPresetSlot, filter_recipes));
}

/* bulk_params.c:49-56 — the 4-term Taylor dB->linear of the bulk path */
#if !ORC_USE_REF
static float db_to_linear_taylor(float db) {
    if (db == 0.0f) return 1.0f;
    if (db < -60.0f) db = -60.0f;
    if (db > 20.0f) db = 20.0f;
    float x = db * 0.1151292546f;
    float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
    float lin = OMAD(x4, 0.0416667f, OMAD(x3, 0.1666667f, OMAD(x2, 0.5f, 1.0f + x)));
    return (lin < 0.0f) ? 0.0f : lin;
}

static void bulk_collect(const orc_ctx *c, WireBulkParams *o) { /* bulk_params.c:62-172 */
    memset(o, 0, sizeof(*o));
    o->header.format_version = WIRE_FORMAT_VERSION;
    o->header.platform_id = PICO_RP2350 ? WIRE_PLATFORM_RP2350 : WIRE_PLATFORM_RP2040;
    o->header.num_channels = NUM_CHANNELS; o->header.num_output_channels = NUM_OUTPUT_CHANNELS;
    o->header.num_input_channels = NUM_INPUT_CHANNELS; o->header.max_bands = MAX_BANDS;
    o->header.payload_length = sizeof(WireBulkParams);
    o->header.fw_version_major = FW_VERSION_MAJOR; o->header.fw_version_minor = FW_VERSION_MINOR;
    o->global.preamp_gain_db = c->global_preamp_db[0];
    o->global.bypass = c->bypass_master_eq ? 1 : 0; o->global.loudness_enabled = c->loudness_enabled ? 1 : 0;
    o->global.loudness_ref_spl = c->loudness_ref_spl; o->global.loudness_intensity_pct = c->loudness_intensity_pct;
    o->crossfeed.enabled = c->crossfeed_config.enabled ? 1 : 0; o->crossfeed.preset = c->crossfeed_config.preset;
    o->crossfeed.itd_enabled = c->crossfeed_config.itd_enabled ? 1 : 0;
    o->crossfeed.custom_fc = c->crossfeed_config.custom_fc; o->crossfeed.custom_feed_db = c->crossfeed_config.custom_feed_db;
    for (int i = 0; i < 3; i++) { o->legacy.gain_db[i] = c->channel_gain_db[i]; o->legacy.mute[i] = c->channel_mute[i] ? 1 : 0; }
    for (int i = 0; i < NUM_CHANNELS; i++) o->delays.delay_ms[i] = c->channel_delays_ms[i];
    for (int in = 0; in < NUM_INPUT_CHANNELS; in++)
        for (int k = 0; k < NUM_OUTPUT_CHANNELS; k++) {
            o->crosspoints[in][k].enabled = c->matrix_mixer.crosspoints[in][k].enabled;
            o->crosspoints[in][k].phase_invert = c->matrix_mixer.crosspoints[in][k].phase_invert;
            o->crosspoints[in][k].gain_db = c->matrix_mixer.crosspoints[in][k].gain_db;
        }
    for (int k = 0; k < NUM_OUTPUT_CHANNELS; k++) {
        o->outputs[k].enabled = c->matrix_mixer.outputs[k].enabled; o->outputs[k].mute = c->matrix_mixer.outputs[k].mute;
        o->outputs[k].gain_db = c->matrix_mixer.outputs[k].gain_db; o->outputs[k].delay_ms = c->matrix_mixer.outputs[k].delay_ms;
    }
    o->pins.num_pin_outputs = NUM_PIN_OUTPUTS;
    for (int i = 0; i < NUM_PIN_OUTPUTS; i++) o->pins.pins[i] = c->output_pins[i];
    for (int ch = 0; ch < NUM_CHANNELS; ch++)
        for (int b = 0; b < MAX_BANDS; b++) {
            o->eq[ch][b].type = c->filter_recipes[ch][b].type; o->eq[ch][b].freq = c->filter_recipes[ch][b].freq;
            o->eq[ch][b].q = c->filter_recipes[ch][b].Q; o->eq[ch][b].gain_db = c->filter_recipes[ch][b].gain_db;
        }
    for (int ch = 0; ch < NUM_CHANNELS; ch++) memcpy(o->channel_names.names[ch], c->channel_names[ch], PRESET_NAME_LEN);
    memcpy(o->i2s_config.output_types, c->output_types, NUM_SPDIF_INSTANCES);
    o->i2s_config.bck_pin = c->i2s_bck_pin; o->i2s_config.mck_pin = c->i2s_mck_pin;
    o->i2s_config.mck_enabled = c->i2s_mck_enabled ? 1 : 0; o->i2s_config.mck_multiplier = (c->i2s_mck_multiplier == 256) ? 1 : 0;
    o->leveller.enabled = c->leveller_config.enabled ? 1 : 0; o->leveller.speed = c->leveller_config.speed;
    o->leveller.lookahead = c->leveller_config.lookahead ? 1 : 0; o->leveller.amount = c->leveller_config.amount;
    o->leveller.max_gain_db = c->leveller_config.max_gain_db; o->leveller.gate_threshold_db = c->leveller_config.gate_threshold_db;
    for (int i = 0; i < NUM_INPUT_CHANNELS; i++) o->preamp.preamp_db[i] = c->global_preamp_db[i];
    o->master_volume.master_volume_db = c->master_volume_db;
}

static int bulk_apply(orc_ctx *c, const WireBulkParams *in, bool apply_pins) { /* bulk_params.c:178-377 */
    if (in->header.format_version < 2 || in->header.format_version > WIRE_FORMAT_VERSION) return -1;
    if (in->header.platform_id != (PICO_RP2350 ? WIRE_PLATFORM_RP2350 : WIRE_PLATFORM_RP2040)) return -2;
    if (in->header.num_channels != NUM_CHANNELS) return -3;
    if (in->header.num_output_channels != NUM_OUTPUT_CHANNELS) return -3;
    uint16_t v5 = sizeof(WireBulkParams) - sizeof(WirePreampConfig) - sizeof(WireMasterVolume);
    uint16_t v2 = v5 - sizeof(WireI2SConfig) - sizeof(WireLevellerConfig);
    if (in->header.payload_length < v2 || in->header.payload_length > sizeof(WireBulkParams)) return -4;

    {
        float db = in->global.preamp_gain_db, lin = db_to_linear_taylor(db);
        for (int i = 0; i < NUM_INPUT_CHANNELS; i++) {
            c->global_preamp_db[i] = db; c->global_preamp_mul[i] = orc_f2i(lin * (float)(1 << 28)); c->global_preamp_linear[i] = lin;
        }
    }
    c->bypass_master_eq = (in->global.bypass != 0);
    c->loudness_enabled = (in->global.loudness_enabled != 0);
    c->loudness_ref_spl = in->global.loudness_ref_spl; c->loudness_intensity_pct = in->global.loudness_intensity_pct;
    c->loudness_recompute_pending = true;
    c->crossfeed_config.enabled = (in->crossfeed.enabled != 0); c->crossfeed_config.preset = in->crossfeed.preset;
    c->crossfeed_config.itd_enabled = (in->crossfeed.itd_enabled != 0);
    c->crossfeed_config.custom_fc = in->crossfeed.custom_fc; c->crossfeed_config.custom_feed_db = in->crossfeed.custom_feed_db;
    c->crossfeed_update_pending = true;
    for (int i = 0; i < 3; i++) {
        c->channel_gain_db[i] = in->legacy.gain_db[i];
        float g = db_to_linear_taylor(in->legacy.gain_db[i]);
        c->channel_gain_mul[i] = orc_f2i(g * 32768.0f); c->channel_gain_linear[i] = g;
        c->channel_mute[i] = (in->legacy.mute[i] != 0);
    }
    for (int i = 0; i < NUM_CHANNELS; i++) c->channel_delays_ms[i] = in->delays.delay_ms[i];
    for (int inp = 0; inp < NUM_INPUT_CHANNELS; inp++)
        for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
            MatrixCrosspoint *xp = &c->matrix_mixer.crosspoints[inp][o];
            xp->enabled = in->crosspoints[inp][o].enabled; xp->phase_invert = in->crosspoints[inp][o].phase_invert;
            xp->gain_db = in->crosspoints[inp][o].gain_db; xp->gain_linear = db_to_linear_taylor(in->crosspoints[inp][o].gain_db);
        }
    for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
        OutputChannel *oc = &c->matrix_mixer.outputs[o];
        oc->enabled = in->outputs[o].enabled; oc->mute = in->outputs[o].mute;
        oc->gain_db = in->outputs[o].gain_db; oc->gain_linear = db_to_linear_taylor(in->outputs[o].gain_db);
        oc->delay_ms = in->outputs[o].delay_ms;
        c->channel_delays_ms[CH_OUT_1 + o] = in->outputs[o].delay_ms;
    }
    if (apply_pins) {
        uint8_t def[NUM_PIN_OUTPUTS]; default_pins(def);
        for (int i = 0; i < NUM_PIN_OUTPUTS; i++) c->output_pins[i] = pin_valid(in->pins.pins[i]) ? in->pins.pins[i] : def[i];
    }
    for (int ch = 0; ch < NUM_CHANNELS; ch++)
        for (int b = 0; b < MAX_BANDS; b++) {
            EqParamPacket *r = &c->filter_recipes[ch][b];
            r->channel = (uint8_t)ch; r->band = (uint8_t)b; r->type = in->eq[ch][b].type;
            r->freq = in->eq[ch][b].freq; r->Q = in->eq[ch][b].q; r->gain_db = in->eq[ch][b].gain_db;
        }
    for (int ch = 0; ch < NUM_CHANNELS; ch++) {
        memcpy(c->channel_names[ch], in->channel_names.names[ch], PRESET_NAME_LEN);
        c->channel_names[ch][PRESET_NAME_LEN - 1] = '\0';
    }
    if (in->header.format_version >= 3 && in->header.payload_length >= v5) {
        memcpy(c->output_types, in->i2s_config.output_types, NUM_SPDIF_INSTANCES);
        c->i2s_bck_pin = in->i2s_config.bck_pin; c->i2s_mck_pin = in->i2s_config.mck_pin;
        c->i2s_mck_enabled = (in->i2s_config.mck_enabled != 0);
        if (in->header.format_version >= 5) c->i2s_mck_multiplier = (in->i2s_config.mck_multiplier == 1) ? 256 : 128;
        else c->i2s_mck_multiplier = (in->i2s_config.mck_multiplier == 0) ? 256 : in->i2s_config.mck_multiplier;
    }
    if (in->header.format_version >= 4) {
        c->leveller_config.enabled = (in->leveller.enabled != 0); c->leveller_config.speed = in->leveller.speed;
        c->leveller_config.lookahead = (in->leveller.lookahead != 0); c->leveller_config.amount = in->leveller.amount;
        c->leveller_config.max_gain_db = in->leveller.max_gain_db; c->leveller_config.gate_threshold_db = in->leveller.gate_threshold_db;
    } else leveller_config_defaults(&c->leveller_config);
    c->leveller_update_pending = true; c->leveller_reset_pending = true;
    if (in->header.format_version >= 6) {
        for (int i = 0; i < NUM_INPUT_CHANNELS; i++) {
            float db = in->preamp.preamp_db[i], lin = db_to_linear_taylor(db);
            c->global_preamp_db[i] = db; c->global_preamp_mul[i] = orc_f2i(lin * (float)(1 << 28)); c->global_preamp_linear[i] = lin;
        }
        apply_master_volume_db(c, in->master_volume.master_volume_db);
    }
    return 0;
}
#endif /* !ORC_USE_REF */

/* ------------------------------------------------------------------------------------- */
/* The packet                                                                              */
/* ------------------------------------------------------------------------------------- */
static float update_preset_mute_envelope(orc_ctx *c, uint32_t n, uint32_t fs_hz) { /* usb_audio.c:459-498 */
    bool active = c->preset_loading;
    if (active) {
        if (c->preset_mute_counter > n) c->preset_mute_counter -= n;
        else { c->preset_mute_counter = 0; c->preset_loading = false; }
    }
    float target = active ? 0.0f : 1.0f;
    if (n == 0) { c->preset_mute_smooth_gain = target; return target; }
    uint64_t tr = ((uint64_t)fs_hz * 8u + 999u) / 1000u;
    if (tr < 1u) tr = 1u;
    float step = (float)n / (float)(uint32_t)tr;
    if (step > 1.0f) step = 1.0f;
    float g = c->preset_mute_smooth_gain;
    if (g < target) { g += step; if (g > target) g = target; }
    else if (g > target) { g -= step; if (g < target) g = target; }
    c->preset_mute_smooth_gain = g;
    return g;
}

#define MAXB 192
static __thread orc_sample buf_l[MAXB], buf_r[MAXB], buf_out[NUM_OUTPUT_CHANNELS][MAXB];

static void process_packet(orc_ctx *c, const uint8_t *data, uint32_t n, int bit_depth,
                           int32_t *pairs /*[pair][n][2]*/, int32_t *sub /*[n]*/) {
    const MatrixMixer *mm = &c->matrix_mixer;
    float preset_mute_gain = update_preset_mute_envelope(c, n, c->audio_state.freq);
    const bool is_bypassed = c->bypass_master_eq;
    const bool loud_on = c->loudness_enabled;
    const LoudnessCoeffs *loud = (c->loudness_row >= 0) ? c->loudness_table[c->loudness_row] : NULL;
    const bool sub_active = (c->core1_mode != CORE1_MODE_EQ_WORKER);  /* usb_audio.c:782 vs :873 */
    const int n_proc = sub_active ? NUM_OUTPUT_CHANNELS : (CORE1_EQ_LAST_OUTPUT + 1);

#if PICO_RP2350
    /* ---- scalars (:564-575) ---- */
    const float inv_32768 = 1.0f / 32768.0f;
    float vol_mul = c->audio_state.mute ? 0.0f : (float)c->audio_state.vol_mul * inv_32768;
    vol_mul *= preset_mute_gain;
    float vol_mul_master = vol_mul * c->master_volume_linear;
    float preamp_l = c->global_preamp_linear[0], preamp_r = c->global_preamp_linear[1];
    float peak_ml = 0, peak_mr = 0;
    const float pdm_scale = (float)(1 << 28);

    /* ---- PASS 1 (:591-686) ---- */
    if (bit_depth == 24) {
        const float inv = 1.0f / 8388608.0f;
        const float gl = inv * preamp_l, gr = inv * preamp_r;
        const uint8_t *p = data;
        for (uint32_t i = 0; i < n; i++, p += 6) {
            int32_t l = (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16);
            int32_t r = (int32_t)((uint32_t)p[3] | (uint32_t)p[4] << 8 | (uint32_t)p[5] << 16);
            l = (int32_t)((uint32_t)l << 8) >> 8; r = (int32_t)((uint32_t)r << 8) >> 8;   /* sbfx #0,#24 */
            buf_l[i] = (float)l * gl; buf_r[i] = (float)r * gr;
        }
    } else {
        const int16_t *in = (const int16_t *)data;
        float gl = inv_32768 * preamp_l, gr = inv_32768 * preamp_r;
        for (uint32_t i = 0; i < n; i++) { buf_l[i] = (float)in[i * 2] * gl; buf_r[i] = (float)in[i * 2 + 1] * gr; }
    }
    /* ---- loudness (:688-718) ---- */
    if (loud_on && loud) {
        for (uint32_t i = 0; i < n; i++) {
            float x[2] = {buf_l[i], buf_r[i]};
            for (int ch = 0; ch < 2; ch++)
                for (int j = 0; j < LOUDNESS_BIQUAD_COUNT; j++) {
                    const LoudnessCoeffs *lc = &loud[j];
                    if (lc->bypass) continue;
                    LoudnessSvfState *st = &c->loudness_state[ch][j];
                    float v3 = x[ch] - st->ic2eq;
                    float v1 = OMAD(lc->sva1, st->ic1eq, lc->sva2 * v3);
                    float v2 = OMAD(lc->sva3, v3, OMAD(lc->sva2, st->ic1eq, st->ic2eq));
                    st->ic1eq = OMAD(2.0f, v1, -st->ic1eq);
                    st->ic2eq = OMAD(2.0f, v2, -st->ic2eq);
                    x[ch] = OMAD(lc->svm2, v2, OMAD(lc->svm0, x[ch], lc->svm1 * v1));
                }
            buf_l[i] = x[0]; buf_r[i] = x[1];
        }
    }
#else
    /* ---- scalars (:975-985) ---- */
    int32_t vol_mul = c->audio_state.mute ? 0 : c->audio_state.vol_mul;
    int32_t mute_q15 = orc_f2i(preset_mute_gain * 32768.0f + 0.5f);
    if (mute_q15 < 0) mute_q15 = 0;
    if (mute_q15 > 32768) mute_q15 = 32768;
    vol_mul = fast_mul_q15(vol_mul, mute_q15);
    int32_t vol_mul_master = fast_mul_q15(vol_mul, c->master_volume_q15);
    int32_t preamp_l = c->global_preamp_mul[0], preamp_r = c->global_preamp_mul[1];
    int32_t peak_ml = 0, peak_mr = 0;

    /* ---- PASS 1 (:997-1015) ---- */
    if (bit_depth == 24) {
        const uint8_t *p = data;
        for (uint32_t i = 0; i < n; i++, p += 6) {
            int32_t l = (int32_t)((uint32_t)p[2] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[0] << 8) >> 2;
            int32_t r = (int32_t)((uint32_t)p[5] << 24 | (uint32_t)p[4] << 16 | (uint32_t)p[3] << 8) >> 2;
            buf_l[i] = qmul(l, preamp_l); buf_r[i] = qmul(r, preamp_r);
        }
    } else {
        const int16_t *in = (const int16_t *)data;
        for (uint32_t i = 0; i < n; i++) {
            int32_t l = (int32_t)((uint32_t)(int32_t)in[i * 2] << 14), r = (int32_t)((uint32_t)(int32_t)in[i * 2 + 1] << 14);
            buf_l[i] = qmul(l, preamp_l); buf_r[i] = qmul(r, preamp_r);
        }
    }
    /* ---- loudness (:1017-1047) ---- */
    if (loud_on && loud) {
        for (uint32_t i = 0; i < n; i++) {
            int32_t x[2] = {buf_l[i], buf_r[i]};
            for (int ch = 0; ch < 2; ch++)
                for (int j = 0; j < LOUDNESS_BIQUAD_COUNT; j++) {
                    const LoudnessCoeffs *lc = &loud[j];
                    if (lc->bypass) continue;
                    Biquad *bq = &c->loudness_biquads[ch][j];
                    int32_t y = wadd(qmul(lc->b0, x[ch]), bq->s1);
                    bq->s1 = wadd(wsub(qmul(lc->b1, x[ch]), qmul(lc->a1, y)), bq->s2);
                    bq->s2 = wsub(qmul(lc->b2, x[ch]), qmul(lc->a2, y));
                    x[ch] = y;
                }
            buf_l[i] = x[0]; buf_r[i] = x[1];
        }
    }
#endif

    /* ---- PASS 2: master EQ (:721-728 / :1050-1055) ---- */
    if (!is_bypassed) {
        if (!c->channel_bypassed[CH_MASTER_LEFT]) eq_block(c, CH_MASTER_LEFT, buf_l, n);
        if (!c->channel_bypassed[CH_MASTER_RIGHT]) eq_block(c, CH_MASTER_RIGHT, buf_r, n);
    }
    /* ---- PASS 2.5: leveller (:731-735 / :1058-1062) ---- */
    if (!c->leveller_bypassed)
        LEAF(leveller_process_block)(&c->leveller_state, &c->leveller_coeffs, &c->leveller_config, buf_l, buf_r, n);

    /* ---- PASS 3: master peaks (pre-crossfeed) + crossfeed (:741-749 / :1065-1073) ---- */
    for (uint32_t i = 0; i < n; i++) {
        orc_sample ml = buf_l[i], mr = buf_r[i];
#if PICO_RP2350
        float al = fabsf(ml); if (al > peak_ml) peak_ml = al;
        float ar = fabsf(mr); if (ar > peak_mr) peak_mr = ar;
#else
        if (wabs(ml) > peak_ml) peak_ml = wabs(ml);
        if (wabs(mr) > peak_mr) peak_mr = wabs(mr);
#endif
        if (!c->crossfeed_bypassed) {
            LEAF(crossfeed_process_stereo)(&c->crossfeed_state, &ml, &mr);
            buf_l[i] = ml; buf_r[i] = mr;
        }
    }

    /* ---- PASS 4: matrix mix (:753-779 / :1076-1100) ---- */
    for (int o = 0; o < NUM_OUTPUT_CHANNELS; o++) {
        orc_sample *dst = buf_out[o];
        if (!mm->outputs[o].enabled) { memset(dst, 0, n * sizeof(orc_sample)); continue; }
        const MatrixCrosspoint *xl = &mm->crosspoints[0][o], *xr = &mm->crosspoints[1][o];
#if PICO_RP2350
        float gl = 0.0f, gr = 0.0f;
        if (xl->enabled) gl = xl->phase_invert ? -xl->gain_linear : xl->gain_linear;
        if (xr->enabled) gr = xr->phase_invert ? -xr->gain_linear : xr->gain_linear;
        if (gl != 0.0f && gr != 0.0f) for (uint32_t i = 0; i < n; i++) dst[i] = OMAD(buf_l[i], gl, buf_r[i] * gr);
        else if (gl != 0.0f) for (uint32_t i = 0; i < n; i++) dst[i] = buf_l[i] * gl;
        else if (gr != 0.0f) for (uint32_t i = 0; i < n; i++) dst[i] = buf_r[i] * gr;
        else memset(dst, 0, n * sizeof(float));
#else
        int32_t gl = xl->enabled ? orc_f2i((xl->phase_invert ? -xl->gain_linear : xl->gain_linear) * 32768.0f) : 0;
        int32_t gr = xr->enabled ? orc_f2i((xr->phase_invert ? -xr->gain_linear : xr->gain_linear) * 32768.0f) : 0;
        if (gl != 0 && gr != 0) for (uint32_t i = 0; i < n; i++) dst[i] = wadd(fast_mul_q15(buf_l[i], gl), fast_mul_q15(buf_r[i], gr));
        else if (gl != 0) for (uint32_t i = 0; i < n; i++) dst[i] = fast_mul_q15(buf_l[i], gl);
        else if (gr != 0) for (uint32_t i = 0; i < n; i++) dst[i] = fast_mul_q15(buf_r[i], gr);
        else memset(dst, 0, n * sizeof(int32_t));
#endif
    }

    /* ---- PASS 5: per-output EQ + gain (:877-895 / :1196-1213 ; Core-1 twin) ---- */
    for (int o = 0; o < n_proc; o++) {
        if (!mm->outputs[o].enabled) continue;
        if (!mm->outputs[o].mute) {
            int ch = CH_OUT_1 + o;
#if PICO_RP2350
            if (!c->channel_bypassed[ch]) eq_block(c, ch, buf_out[o], n);
#else
            if (!is_bypassed && !c->channel_bypassed[ch]) eq_block(c, ch, buf_out[o], n);   /* Q28 gates on master bypass too */
#endif
        }
#if PICO_RP2350
        float gain = mm->outputs[o].mute ? 0.0f : mm->outputs[o].gain_linear * vol_mul_master;
        if (gain == 0.0f) memset(buf_out[o], 0, n * sizeof(float));
        else if (gain != 1.0f) for (uint32_t i = 0; i < n; i++) buf_out[o][i] *= gain;
#else
        int32_t gain = mm->outputs[o].mute ? 0 : orc_f2i(mm->outputs[o].gain_linear * (float)vol_mul_master);
        if (gain == 0) memset(buf_out[o], 0, n * sizeof(int32_t));
        else for (uint32_t i = 0; i < n; i++) buf_out[o][i] = fast_mul_q15(buf_out[o][i], gain);
#endif
    }

    /* ---- PASS 6: delay (:898-912 / :1216-1230) ---- */
    if (c->any_delay_active) {
        for (int o = 0; o < n_proc; o++) {
            int32_t dly = c->channel_delay_samples[o];
            if (dly <= 0) continue;
            orc_sample *dst = buf_out[o], *line = c->delay_lines[o];
            uint32_t w = c->delay_write_idx;
            for (uint32_t i = 0; i < n; i++) {
                line[w] = dst[i];
                dst[i] = line[(w - (uint32_t)dly) & MAX_DELAY_MASK];
                w = (w + 1) & MAX_DELAY_MASK;
            }
        }
        c->delay_write_idx = (c->delay_write_idx + n) & MAX_DELAY_MASK;
    }

    /* ---- PASS 7: output peaks, S/PDIF words, sub (:914-959 / :1232-1275) ---- */
    for (int o = 0; o < NUM_SPDIF_INSTANCES * 2; o++) {
#if PICO_RP2350
        float peak = 0;
        for (uint32_t i = 0; i < n; i++) { float a = fabsf(buf_out[o][i]); if (a > peak) peak = a; }
        c->status.peaks[CH_OUT_1 + o] = (uint16_t)(fminf(1.0f, peak) * 32767.0f);
        if (peak > CLIP_THRESH_F) c->status.clip_flags |= (uint16_t)(1u << (CH_OUT_1 + o));
#else
        int32_t peak = 0;
        for (uint32_t i = 0; i < n; i++) { int32_t a = wabs(buf_out[o][i]); if (a > peak) peak = a; }
        c->status.peaks[CH_OUT_1 + o] = (uint16_t)(peak >> 13);
        if (peak > CLIP_THRESH_Q28) c->status.clip_flags |= (uint16_t)(1u << (CH_OUT_1 + o));
#endif
    }
    for (int pair = 0; pair < NUM_SPDIF_INSTANCES; pair++) {
        int32_t *op = pairs + (size_t)pair * n * 2;
        int lc = pair * 2, rc = pair * 2 + 1;
        if (!mm->outputs[lc].enabled && !mm->outputs[rc].enabled) { memset(op, 0, n * 8); continue; }
        for (uint32_t i = 0; i < n; i++) {
#if PICO_RP2350
            float dl = fmaxf(-1.0f, fminf(1.0f, buf_out[lc][i]));
            float dr = fmaxf(-1.0f, fminf(1.0f, buf_out[rc][i]));
            op[i * 2] = (int32_t)(dl * 8388607.0f);
            op[i * 2 + 1] = (int32_t)(dr * 8388607.0f);
#else
            op[i * 2] = clip_s24(wadd(buf_out[lc][i], 1 << 5) >> 6);
            op[i * 2 + 1] = clip_s24(wadd(buf_out[rc][i], 1 << 5) >> 6);
#endif
        }
    }
    {
        const int so = NUM_OUTPUT_CHANNELS - 1;
        if (sub_active && mm->outputs[so].enabled) {
#if PICO_RP2350
            float peak = 0;
            for (uint32_t i = 0; i < n; i++) { float a = fabsf(buf_out[so][i]); if (a > peak) peak = a; }
            c->status.peaks[CH_OUT_SUB] = (uint16_t)(fminf(1.0f, peak) * 32767.0f);
            if (peak > CLIP_THRESH_F) c->status.clip_flags |= (uint16_t)(1u << CH_OUT_SUB);
            for (uint32_t i = 0; i < n; i++) sub[i] = orc_f2i(buf_out[so][i] * pdm_scale);
#else
            int32_t peak = 0;
            for (uint32_t i = 0; i < n; i++) { int32_t a = wabs(buf_out[so][i]); if (a > peak) peak = a; }
            c->status.peaks[CH_OUT_SUB] = (uint16_t)(peak >> 13);
            if (peak > CLIP_THRESH_Q28) c->status.clip_flags |= (uint16_t)(1u << CH_OUT_SUB);
            for (uint32_t i = 0; i < n; i++) sub[i] = buf_out[so][i];
#endif
        } else {
            c->status.peaks[CH_OUT_SUB] = 0;
            memset(sub, 0, n * sizeof(int32_t));       /* nothing is pushed to the PDM ring */
        }
    }
#if PICO_RP2350
    c->status.peaks[0] = (uint16_t)(fminf(1.0f, peak_ml) * 32767.0f);
    c->status.peaks[1] = (uint16_t)(fminf(1.0f, peak_mr) * 32767.0f);
    if (peak_ml > CLIP_THRESH_F) c->status.clip_flags |= 1u << CH_MASTER_LEFT;
    if (peak_mr > CLIP_THRESH_F) c->status.clip_flags |= 1u << CH_MASTER_RIGHT;
#else
    c->status.peaks[0] = (uint16_t)(peak_ml >> 13);
    c->status.peaks[1] = (uint16_t)(peak_mr >> 13);
    if (peak_ml > CLIP_THRESH_Q28) c->status.clip_flags |= 1u << CH_MASTER_LEFT;
    if (peak_mr > CLIP_THRESH_Q28) c->status.clip_flags |= 1u << CH_MASTER_RIGHT;
#endif
}

/* ===================================================================================== */
/* Public API (orc_api.h)                                                                  */
/* ===================================================================================== */
int orc_flavor(void) { return PICO_RP2350 ? 1 : 0; }
int orc_is_ref_build(void) { return ORC_USE_REF ? 1 : 0; }
int orc_num_channels(void) { return NUM_CHANNELS; }
int orc_num_outputs(void) { return NUM_OUTPUT_CHANNELS; }
int orc_num_pairs(void) { return NUM_SPDIF_INSTANCES; }
int orc_preset_slot_size(void) { return (int)sizeof(OrcPresetSlot); }
void orc_set_math_mode(int detmath) { orc_math_mode = detmath; }
void orc_set_x86_cast_semantics(int on) { orc_x86_cast_semantics = on; }
void orc_set_fma_mode(int on) { orc_fma_mode = (on != 0) && PICO_RP2350; }

static int flash_select_(orc_ctx *c, const void *dump, int booting, int *wrote);

/* The power-on sequence.  `dump` NULL: an erased flash (first boot: preset_boot_load writes the fresh directory, which arms the mute).
 * Otherwise the device boots from that 48 KB preset area: preset_boot_load's selection, applied the boot path's way — apply_slot_to_live /
 * apply_factory_defaults and nothing else (flash_storage.c:1047-1082: no flash write, no mute, the delay lines untouched).  PINNED by the
 * firmware build booted from the same dumps, from the first frame (tests/test_oracle_vs_fw.py::test_boot_from_flash_dumps). */
static orc_ctx *orc_boot_(const void *dump, int *sel) {
    unsigned csr = orc_enter();
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    /* power-on values of the globals (usb_audio.c:47, :148-211, :457) */
    c->audio_state.freq = 44100;
    c->master_volume_db = MASTER_VOL_DEFAULT_DB; c->master_volume_linear = 0.1f; c->master_volume_q15 = 3277;
    for (int i = 0; i < NUM_INPUT_CHANNELS; i++) { c->global_preamp_mul[i] = 268435456; c->global_preamp_linear[i] = 1.0f; }
    for (int i = 0; i < 3; i++) { c->channel_gain_mul[i] = 32768; c->channel_gain_linear[i] = 1.0f; }
    c->loudness_ref_spl = 83.0f; c->loudness_intensity_pct = 100.0f; c->loudness_row = -1;
    c->crossfeed_config.itd_enabled = true; c->crossfeed_config.custom_fc = 700.0f; c->crossfeed_config.custom_feed_db = 4.5f;
    c->crossfeed_bypassed = true;
    leveller_config_defaults(&c->leveller_config); c->leveller_bypassed = true;
    c->preset_mute_smooth_gain = 1.0f;
    default_pins(c->output_pins);
    c->i2s_bck_pin = PICO_I2S_BCK_PIN; c->i2s_mck_pin = PICO_I2S_MCK_PIN; c->i2s_mck_multiplier = 128;
    c->dir_master_volume_mode = MASTER_VOLUME_MODE_INDEPENDENT; c->dir_master_volume_db = MASTER_VOL_DEFAULT_DB; /* flash_storage.c:450-451 */
    /* usb_sound_card_init (:3251-3270, :3382-3385) */
    apply_factory_defaults(c);           /* matrix_init_defaults + dsp_init_default_filters are subsets of this */
    recalculate_all_filters(c, 48000.0f);
    set_volume_(c, 0);
    /* core0_init (main.c:645-696): preset_boot_load on blank flash = factory defaults, and the fresh directory is written
     * (flash_storage.c:1097-1100): flash_write_sector re-arms the preset mute for flash_mute_hold_samples() = max(10 ms, 512)
     * samples at the power-on 44.1 kHz (:262-266, :349-350) — found by running the firmware build (tests/test_oracle_vs_fw.py) */
    if (!dump) prepare_pipeline_reset(c, 512);
    else {
        /* a boot that WRITES the flash arms the same mute: no directory (legacy migration or a fresh directory, flash_storage.c:1084-1104)
         * or a v1 directory, which dir_load_cache persists as v2 (:391-414) */
        int wrote = 0, r = flash_select_(c, dump, 1, &wrote);
        if (sel) *sel = r;
        if (wrote) prepare_pipeline_reset(c, 512);
    }
    recalculate_all_filters(c, 48000.0f); update_delay_samples(c, 48000.0f);
    /* slots the preset saved as I2S go through process_type_switches before Core 1 starts (main.c:651-684): prepare_pipeline_reset(PRESET_MUTE_SAMPLES), :279 */
    for (int i = 0; i < NUM_SPDIF_INSTANCES; i++) if (c->output_types[i] != 0) { prepare_pipeline_reset(c, PRESET_MUTE_SAMPLES); break; }
    loudness_recompute(c, 48000.0f); c->loudness_recompute_pending = false;
    if (c->loudness_enabled) set_volume_(c, c->audio_state.volume);
    LEAF(leveller_compute_coefficients)(&c->leveller_coeffs, &c->leveller_config, 48000.0f);
    LEAF(leveller_reset_state)(&c->leveller_state);
    c->leveller_bypassed = !c->leveller_config.enabled;
    c->leveller_update_pending = false; c->leveller_reset_pending = false;
    transition_core1(c);
    /* first main-loop pass: the rate change queued by _audio_reconfigure() (:3385) at freq = 44100 */
    perform_rate_change(c, c->audio_state.freq);
    service(c);
    orc_leave(csr);
    return c;
}

orc_ctx *orc_new(void) { return orc_boot_(NULL, NULL); }
orc_ctx *orc_new_from_flash(const void *dump48k, uint32_t len, int *selection) {      /* selection: orc_load_flash_dump's codes */
    if (!dump48k || len < 12u * 4096u) return NULL;
    return orc_boot_(dump48k, selection);
}

void orc_free(orc_ctx *c) { free(c); }

int orc_set_sample_rate(orc_ctx *c, uint32_t hz) { /* usb_audio.c:1491-1498 + main.c:860-865 */
    if (hz != 44100 && hz != 48000 && hz != 96000) return -1;
    unsigned csr = orc_enter();
    if (c->audio_state.freq != hz) { c->audio_state.freq = hz; perform_rate_change(c, hz); service(c); }
    orc_leave(csr);
    return 0;
}
void orc_set_host_volume(orc_ctx *c, int16_t v) { unsigned csr = orc_enter(); set_volume_(c, v); orc_leave(csr); }
void orc_set_mute(orc_ctx *c, int mute) { c->audio_state.mute = mute != 0; }

void orc_factory_defaults(orc_ctx *c) { /* REQ_FACTORY_RESET -> preset_load of an empty slot: flash_storage.c:811-833 */
    unsigned csr = orc_enter();
    prepare_pipeline_reset(c, PRESET_MUTE_SAMPLES);
    apply_factory_defaults(c);
    float fs = (float)c->audio_state.freq;
    recalculate_all_filters(c, fs); update_delay_samples(c, fs);
    memset(c->delay_lines, 0, sizeof(c->delay_lines));
    transition_core1(c);
    service(c);
    orc_leave(csr);
}

int orc_load_bulk(orc_ctx *c, const void *blob, uint32_t len) { /* main.c:1126-1162 */
    if (len != sizeof(WireBulkParams)) return -4;        /* usb_audio.c:2250-2251: only the exact length starts a transfer */
    unsigned csr = orc_enter();
    prepare_pipeline_reset(c, PRESET_MUTE_SAMPLES);
#if ORC_USE_REF
    ref_push(c);
    int err = bulk_params_apply((const WireBulkParams *)blob, c->dir_include_pins != 0);
    ref_pull(c);
#else
    int err = bulk_apply(c, (const WireBulkParams *)blob, c->dir_include_pins != 0);
#endif
    if (err == 0) {
        float fs = (float)c->audio_state.freq;
        recalculate_all_filters(c, fs); update_delay_samples(c, fs);
        transition_core1(c);
    }
    service(c);
    orc_leave(csr);
    return err;
}

int orc_collect_bulk(orc_ctx *c, void *blob) {
#if ORC_USE_REF
    ref_push(c); bulk_params_collect((WireBulkParams *)blob);
#else
    bulk_collect(c, (WireBulkParams *)blob);
#endif
    return (int)sizeof(WireBulkParams);
}

int orc_load_preset_slot(orc_ctx *c, const void *image, uint32_t len, int expect_slot) { /* main.c:926-976, flash_storage.c:750-849 */
    if (len < sizeof(OrcPresetSlot)) return PRESET_ERR_CRC;
    unsigned csr = orc_enter();
    prepare_pipeline_reset(c, PRESET_MUTE_SAMPLES);
    OrcPresetSlot s; memcpy(&s, image, sizeof s);
    bool ok = (s.magic == SLOT_MAGIC) && (expect_slot < 0 || s.slot_index == (uint16_t)expect_slot) &&
              crc32_((const uint8_t *)&s.filter_recipes, sizeof(OrcPresetSlot) - offsetof(OrcPresetSlot, filter_recipes)) == s.crc32;
    int rc = PRESET_OK;
    if (!ok) { c->preset_loading = false; rc = PRESET_ERR_CRC; }
    else {
        uint8_t old_types[NUM_SPDIF_INSTANCES]; memcpy(old_types, c->output_types, NUM_SPDIF_INSTANCES);   /* main.c:934-935 */
        apply_slot_to_live(c, &s, c->dir_include_pins != 0);
        apply_master_volume_from_mode(c, &s);
        float fs = (float)c->audio_state.freq;
        recalculate_all_filters(c, fs); update_delay_samples(c, fs);
        memset(c->delay_lines, 0, sizeof(c->delay_lines));
        transition_core1(c);
        /* preset_load ends by writing the directory (last_active_slot, flash_storage.c:846-847); every flash_write_sector
         * re-arms the mute for flash_mute_hold_samples() = max(10 ms, 512 samples) (:262-266, :349-350) — the hold after a
         * preset load is that, not PRESET_MUTE_SAMPLES (found by running the firmware build, tests/test_oracle_vs_fw.py) */
        prepare_pipeline_reset(c, flash_mute_hold_samples_(c->audio_state.freq));
        /* a slot type that changed goes through process_type_switches: prepare_pipeline_reset once more (main.c:957-972, :279) */
        if (memcmp(old_types, c->output_types, NUM_SPDIF_INSTANCES) != 0) prepare_pipeline_reset(c, PRESET_MUTE_SAMPLES);
        service(c);
    }
    orc_leave(csr);
    return rc;
}

/* ---- flash dump: directory, startup-slot selection, legacy migration ----
 * flash_storage.c:95-131 (directory v1 / v2), :370-417 (dir_load_cache), :997-1045 (migrate_legacy), :1047-1105
 * (preset_boot_load).  Restated here; PINNED by the firmware build (ref_fw_flash.c compiles flash_storage.c in place over a RAM
 * flash: tests/test_oracle_vs_fw.py::test_boot_from_flash_dumps, ::test_legacy_sector_migration); the slot payloads
 * go through orc_load_preset_slot, i.e. the selection is the boot path's and the application is preset_load's. */
typedef struct __attribute__((packed)) {
    uint32_t magic; uint16_t version; uint16_t reserved; uint32_t crc32;
    uint8_t startup_mode, default_slot, last_active_slot, include_pins;
    uint16_t slot_occupied; uint8_t include_master_volume; uint8_t padding[1];
    char slot_names[PRESET_SLOTS][PRESET_NAME_LEN];
} OrcDirV1;
typedef struct __attribute__((packed)) {
    uint32_t magic; uint16_t version; uint16_t reserved; uint32_t crc32;
    uint8_t startup_mode, default_slot, last_active_slot, include_pins;
    uint16_t slot_occupied; uint8_t master_volume_mode; uint8_t padding[1];
    float master_volume_db;
    char slot_names[PRESET_SLOTS][PRESET_NAME_LEN];
} OrcDirV2;
#define ORC_DIR_MAGIC 0x44535032u
#define ORC_LEGACY_MAGIC 0x44535031u
#define ORC_SECTOR 4096u
#ifndef PRESET_STARTUP_LAST_ACTIVE
#define PRESET_STARTUP_LAST_ACTIVE 1   /* config.h:259 */
#endif

/* a validated slot into the live parameters, nothing else (validate_slot + apply_slot_to_live + apply_master_volume_from_mode) */
static int slot_to_live_(orc_ctx *c, const void *image, int expect_slot) {
    OrcPresetSlot s; memcpy(&s, image, sizeof s);
    bool ok = (s.magic == SLOT_MAGIC) && (expect_slot < 0 || s.slot_index == (uint16_t)expect_slot) &&
              crc32_((const uint8_t *)&s.filter_recipes, sizeof(OrcPresetSlot) - offsetof(OrcPresetSlot, filter_recipes)) == s.crc32;
    if (!ok) return PRESET_ERR_CRC;
    apply_slot_to_live(c, &s, c->dir_include_pins != 0);
    apply_master_volume_from_mode(c, &s);
    return PRESET_OK;
}

int orc_load_flash_dump(orc_ctx *c, const void *dump, uint32_t len) {
    if (len < 12u * ORC_SECTOR) return -4;
    return flash_select_(c, dump, 0, NULL);
}

/* booting: the application is the boot path's (apply_slot_to_live / apply_factory_defaults, flash_storage.c:1066-1076); else preset_load's */
static int flash_select_(orc_ctx *c, const void *dump, int booting, int *wrote) {
    const uint8_t *p = (const uint8_t *)dump;
    if (wrote) *wrote = 1;
    OrcDirV2 d2; memcpy(&d2, p, sizeof d2);
    int have_dir = 0;
    if (d2.magic == ORC_DIR_MAGIC) {
        if (d2.version == 2) {
            have_dir = crc32_((const uint8_t *)&d2.startup_mode, sizeof d2 - offsetof(OrcDirV2, startup_mode)) == d2.crc32;
        } else if (d2.version == 1) {
            OrcDirV1 d1; memcpy(&d1, p, sizeof d1);
            if (crc32_((const uint8_t *)&d1.startup_mode, sizeof d1 - offsetof(OrcDirV1, startup_mode)) == d1.crc32) {
                have_dir = 2;
                d2.startup_mode = d1.startup_mode; d2.default_slot = d1.default_slot; d2.last_active_slot = d1.last_active_slot;
                d2.include_pins = d1.include_pins; d2.slot_occupied = d1.slot_occupied;
                d2.master_volume_mode = d1.include_master_volume ? MASTER_VOLUME_MODE_WITH_PRESET : MASTER_VOLUME_MODE_INDEPENDENT;
                d2.master_volume_db = MASTER_VOL_DEFAULT_DB;
            }
        }
    }
    if (have_dir) {
        if (wrote) *wrote = have_dir == 2;      /* (a v1 directory is written back as v2) */
        uint8_t target = d2.startup_mode == PRESET_STARTUP_LAST_ACTIVE ? d2.last_active_slot : d2.default_slot;
        if (target >= PRESET_SLOTS) { target = d2.default_slot; if (target >= PRESET_SLOTS) target = 0; }
        c->dir_master_volume_mode = d2.master_volume_mode; c->dir_master_volume_db = d2.master_volume_db; c->dir_include_pins = d2.include_pins;
        if (d2.slot_occupied & (1u << target))
            if ((booting ? slot_to_live_(c, p + (1u + target) * ORC_SECTOR, target)
                         : orc_load_preset_slot(c, p + (1u + target) * ORC_SECTOR, sizeof(OrcPresetSlot), target)) == PRESET_OK) return target;
        if (booting) apply_factory_defaults(c); else orc_factory_defaults(c);
        return 16 + target;
    }
    const uint8_t *lg = p + 11u * ORC_SECTOR;
    const size_t legacy_bytes = offsetof(OrcPresetSlot, channel_names);       /* LegacyFlashStorage ends where the names begin */
    uint32_t magic, crc; uint16_t version;
    memcpy(&magic, lg, 4); memcpy(&version, lg + 4, 2); memcpy(&crc, lg + 8, 4);
    c->dir_master_volume_mode = MASTER_VOLUME_MODE_INDEPENDENT; c->dir_master_volume_db = MASTER_VOL_DEFAULT_DB;
    if (magic == ORC_LEGACY_MAGIC && crc32_(lg + 12, legacy_bytes - 12) == crc) {
        OrcPresetSlot s; memset(&s, 0, sizeof s);
        memcpy((uint8_t *)&s + 12, lg + 12, legacy_bytes - 12);
        s.magic = SLOT_MAGIC; s.version = version; s.slot_index = 0;
        s.crc32 = crc32_((const uint8_t *)&s.filter_recipes, sizeof(OrcPresetSlot) - offsetof(OrcPresetSlot, filter_recipes));
        c->dir_include_pins = 0;
        int rc = booting ? slot_to_live_(c, &s, 0) : orc_load_preset_slot(c, &s, sizeof s, 0);
        c->dir_include_pins = 1;
        if (rc == PRESET_OK) return 32;
    }
    c->dir_include_pins = 1;
    if (booting) apply_factory_defaults(c); else orc_factory_defaults(c);
    return 48;
}

int orc_save_preset_slot(orc_ctx *c, void *image, int slot_index) {
    collect_live_state(c, (OrcPresetSlot *)image, (uint8_t)slot_index);
    return (int)sizeof(OrcPresetSlot);
}

static float rd_f32(const uint8_t *p) { float f; memcpy(&f, p, 4); return f; }

int orc_vendor_set(orc_ctx *c, uint8_t req, uint16_t wValue, const void *payload, uint16_t len) { /* usb_audio.c:1632-2021 */
    unsigned csr = orc_enter();
    const uint8_t *b = (const uint8_t *)payload;
    uint8_t idx = wValue & 0xFF;
    int handled = 1;
    float fs = (float)c->audio_state.freq;
    switch (req) {
        case REQ_SET_EQ_PARAM:
            if (len >= sizeof(EqParamPacket)) {
                EqParamPacket p; memcpy(&p, b, sizeof p);
                if (p.channel < NUM_CHANNELS && p.band < band_counts[p.channel]) {   /* main.c:826-857 */
                    c->filter_recipes[p.channel][p.band] = p;
                    LEAF(dsp_compute_coefficients)(&p, &c->filters[p.channel][p.band], fs);  /* note: clamps the COPY */
                    recalc_channel_bypass(c, p.channel);
                }
            }
            break;
        case REQ_SET_PREAMP: if (len >= 4) for (int ch = 0; ch < NUM_INPUT_CHANNELS; ch++) update_preamp(c, (uint8_t)ch, rd_f32(b)); break;
        case REQ_SET_PREAMP_CH: if (idx < NUM_INPUT_CHANNELS && len >= 4) update_preamp(c, idx, rd_f32(b)); break;
        case REQ_SET_MASTER_VOLUME: if (len >= 4) update_master_volume(c, rd_f32(b)); break;
        case REQ_SET_DELAY:
            if (idx < NUM_CHANNELS && len >= 4) { float ms = rd_f32(b); if (ms < 0) ms = 0; c->channel_delays_ms[idx] = ms; update_delay_samples(c, fs); }
            break;
        case REQ_SET_BYPASS: if (len >= 1) c->bypass_master_eq = (b[0] != 0); break;
        case REQ_SET_CHANNEL_GAIN:
            if (idx < 3 && len >= 4) {
                float db = rd_f32(b); c->channel_gain_db[idx] = db;
                float lin = powf(10.0f, db / 20.0f);
                c->channel_gain_mul[idx] = orc_f2i(lin * 32768.0f); c->channel_gain_linear[idx] = lin;
            }
            break;
        case REQ_SET_CHANNEL_MUTE: if (idx < 3 && len >= 1) c->channel_mute[idx] = (b[0] != 0); break;
        case REQ_SET_LOUDNESS:
            if (len >= 1) {
                c->loudness_enabled = (b[0] != 0);
                if (c->loudness_enabled && c->loudness_table_valid) c->loudness_row = volume_index(c->audio_state.volume);
                else c->loudness_row = -1;
            }
            break;
        case REQ_SET_LOUDNESS_REF:
            if (len >= 4) { float v = rd_f32(b); if (v < 40.0f) v = 40.0f; if (v > 100.0f) v = 100.0f; c->loudness_ref_spl = v; c->loudness_recompute_pending = true; }
            break;
        case REQ_SET_LOUDNESS_INTENSITY:
            if (len >= 4) { float v = rd_f32(b); if (v < 0.0f) v = 0.0f; if (v > 200.0f) v = 200.0f; c->loudness_intensity_pct = v; c->loudness_recompute_pending = true; }
            break;
        case REQ_SET_CROSSFEED: if (len >= 1) { c->crossfeed_config.enabled = (b[0] != 0); c->crossfeed_update_pending = true; } break;
        case REQ_SET_CROSSFEED_PRESET:
            if (len >= 1 && b[0] <= CROSSFEED_PRESET_CUSTOM) { c->crossfeed_config.preset = b[0]; c->crossfeed_update_pending = true; }
            break;
        case REQ_SET_CROSSFEED_FREQ:
            if (len >= 4) {
                float v = rd_f32(b); if (v < CROSSFEED_FREQ_MIN) v = CROSSFEED_FREQ_MIN; if (v > CROSSFEED_FREQ_MAX) v = CROSSFEED_FREQ_MAX;
                c->crossfeed_config.custom_fc = v;
                if (c->crossfeed_config.preset == CROSSFEED_PRESET_CUSTOM) c->crossfeed_update_pending = true;
            }
            break;
        case REQ_SET_CROSSFEED_FEED:
            if (len >= 4) {
                float v = rd_f32(b); if (v < CROSSFEED_FEED_MIN) v = CROSSFEED_FEED_MIN; if (v > CROSSFEED_FEED_MAX) v = CROSSFEED_FEED_MAX;
                c->crossfeed_config.custom_feed_db = v;
                if (c->crossfeed_config.preset == CROSSFEED_PRESET_CUSTOM) c->crossfeed_update_pending = true;
            }
            break;
        case REQ_SET_CROSSFEED_ITD: if (len >= 1) { c->crossfeed_config.itd_enabled = (b[0] != 0); c->crossfeed_update_pending = true; } break;
        case REQ_SET_LEVELLER_ENABLE:
            if (len >= 1) { c->leveller_config.enabled = (b[0] != 0); c->leveller_update_pending = true; c->leveller_reset_pending = true; }
            break;
        case REQ_SET_LEVELLER_AMOUNT:
            if (len >= 4) { float v = rd_f32(b); if (v < LEVELLER_AMOUNT_MIN) v = LEVELLER_AMOUNT_MIN; if (v > LEVELLER_AMOUNT_MAX) v = LEVELLER_AMOUNT_MAX; c->leveller_config.amount = v; c->leveller_update_pending = true; }
            break;
        case REQ_SET_LEVELLER_SPEED: if (len >= 1 && b[0] < LEVELLER_SPEED_COUNT) { c->leveller_config.speed = b[0]; c->leveller_update_pending = true; } break;
        case REQ_SET_LEVELLER_MAX_GAIN:
            if (len >= 4) { float v = rd_f32(b); if (v < LEVELLER_MAX_GAIN_MIN) v = LEVELLER_MAX_GAIN_MIN; if (v > LEVELLER_MAX_GAIN_MAX) v = LEVELLER_MAX_GAIN_MAX; c->leveller_config.max_gain_db = v; c->leveller_update_pending = true; }
            break;
        case REQ_SET_LEVELLER_LOOKAHEAD:
            if (len >= 1) { c->leveller_config.lookahead = (b[0] != 0); c->leveller_update_pending = true; c->leveller_reset_pending = true; }
            break;
        case REQ_SET_LEVELLER_GATE:
            if (len >= 4) { float v = rd_f32(b); if (v < LEVELLER_GATE_MIN) v = LEVELLER_GATE_MIN; if (v > LEVELLER_GATE_MAX) v = LEVELLER_GATE_MAX; c->leveller_config.gate_threshold_db = v; c->leveller_update_pending = true; }
            break;
        case REQ_SET_MATRIX_ROUTE:
            if (len >= sizeof(MatrixRoutePacket)) {
                MatrixRoutePacket pk; memcpy(&pk, b, sizeof pk);
                if (pk.input < NUM_INPUT_CHANNELS && pk.output < NUM_OUTPUT_CHANNELS) {
                    MatrixCrosspoint *xp = &c->matrix_mixer.crosspoints[pk.input][pk.output];
                    xp->enabled = pk.enabled; xp->phase_invert = pk.phase_invert; xp->gain_db = pk.gain_db;
                    xp->gain_linear = powf(10.0f, pk.gain_db / 20.0f);
                }
            }
            break;
        case REQ_SET_OUTPUT_ENABLE:
            if (idx < NUM_OUTPUT_CHANNELS && len >= 1) {
                bool want = (b[0] != 0), skip = false;
                if (want) {          /* PDM vs Core-1 EQ outputs are mutually exclusive (:1891-1904) */
                    bool is_pdm = (idx == NUM_OUTPUT_CHANNELS - 1);
                    bool is_c1 = (idx >= CORE1_EQ_FIRST_OUTPUT && idx <= CORE1_EQ_LAST_OUTPUT);
                    if (is_pdm) { for (int i = CORE1_EQ_FIRST_OUTPUT; i <= CORE1_EQ_LAST_OUTPUT; i++) if (c->matrix_mixer.outputs[i].enabled) skip = true; }
                    else if (is_c1) { if (c->matrix_mixer.outputs[NUM_OUTPUT_CHANNELS - 1].enabled) skip = true; }
                }
                if (!skip) { c->matrix_mixer.outputs[idx].enabled = want ? 1 : 0; transition_core1(c); }
            }
            break;
        case REQ_SET_OUTPUT_GAIN:
            if (idx < NUM_OUTPUT_CHANNELS && len >= 4) { float db = rd_f32(b); c->matrix_mixer.outputs[idx].gain_db = db; c->matrix_mixer.outputs[idx].gain_linear = powf(10.0f, db / 20.0f); }
            break;
        case REQ_SET_OUTPUT_MUTE: if (idx < NUM_OUTPUT_CHANNELS && len >= 1) c->matrix_mixer.outputs[idx].mute = b[0]; break;
        case REQ_SET_OUTPUT_DELAY:
            if (idx < NUM_OUTPUT_CHANNELS && len >= 4) {
                float ms = rd_f32(b); if (ms < 0) ms = 0;
                c->matrix_mixer.outputs[idx].delay_ms = ms; c->channel_delays_ms[CH_OUT_1 + idx] = ms; update_delay_samples(c, fs);
            }
            break;
        case REQ_SET_MASTER_VOLUME_MODE:
            if (len >= 1) { uint8_t m = b[0]; if (m > MASTER_VOLUME_MODE_WITH_PRESET) m = MASTER_VOLUME_MODE_INDEPENDENT; c->dir_master_volume_mode = m; }
            break;
        case REQ_SET_CHANNEL_NAME:
            if (idx < NUM_CHANNELS && len > 0) {
                memset(c->channel_names[idx], 0, PRESET_NAME_LEN);
                size_t n = len < (PRESET_NAME_LEN - 1) ? len : (PRESET_NAME_LEN - 1);
                memcpy(c->channel_names[idx], b, n);
            }
            break;
        default: handled = 0; break;
    }
    service(c);
    orc_leave(csr);
    return handled ? 0 : -1;
}

static int put(void *buf, uint16_t cap, const void *src, int n) { if (cap < n) return -2; memcpy(buf, src, (size_t)n); return n; }
static int put_u32(void *buf, uint16_t cap, uint32_t v, int n) { return put(buf, cap, &v, n); }

int orc_vendor_get(orc_ctx *c, uint8_t req, uint16_t wValue, void *buf, uint16_t cap) { /* usb_audio.c:2271-2688 */
    uint8_t idx = (uint8_t)wValue, u8;
    switch (req) {
        case REQ_GET_PREAMP: return put(buf, cap, &c->global_preamp_db[0], 4);
        case REQ_GET_PREAMP_CH: return idx < NUM_INPUT_CHANNELS ? put(buf, cap, &c->global_preamp_db[idx], 4) : -1;
        case REQ_GET_MASTER_VOLUME: return put(buf, cap, &c->master_volume_db, 4);
        case REQ_GET_MASTER_VOLUME_MODE: return put(buf, cap, &c->dir_master_volume_mode, 1);
        case REQ_GET_SAVED_MASTER_VOLUME: return put(buf, cap, &c->dir_master_volume_db, 4);
        case REQ_SAVE_MASTER_VOLUME: c->dir_master_volume_db = c->master_volume_db; u8 = 0; return put(buf, cap, &u8, 1);
        case REQ_GET_DELAY: return idx < NUM_CHANNELS ? put(buf, cap, &c->channel_delays_ms[idx], 4) : -1;
        case REQ_GET_BYPASS: u8 = c->bypass_master_eq ? 1 : 0; return put(buf, cap, &u8, 1);
        case REQ_GET_CHANNEL_GAIN: return idx < 3 ? put(buf, cap, &c->channel_gain_db[idx], 4) : -1;
        case REQ_GET_CHANNEL_MUTE: if (idx >= 3) return -1; u8 = c->channel_mute[idx] ? 1 : 0; return put(buf, cap, &u8, 1);
        case REQ_GET_LOUDNESS: u8 = c->loudness_enabled ? 1 : 0; return put(buf, cap, &u8, 1);
        case REQ_GET_LOUDNESS_REF: return put(buf, cap, &c->loudness_ref_spl, 4);
        case REQ_GET_LOUDNESS_INTENSITY: return put(buf, cap, &c->loudness_intensity_pct, 4);
        case REQ_GET_CROSSFEED: u8 = c->crossfeed_config.enabled ? 1 : 0; return put(buf, cap, &u8, 1);
        case REQ_GET_CROSSFEED_PRESET: return put(buf, cap, &c->crossfeed_config.preset, 1);
        case REQ_GET_CROSSFEED_FREQ: return put(buf, cap, &c->crossfeed_config.custom_fc, 4);
        case REQ_GET_CROSSFEED_FEED: return put(buf, cap, &c->crossfeed_config.custom_feed_db, 4);
        case REQ_GET_CROSSFEED_ITD: u8 = c->crossfeed_config.itd_enabled ? 1 : 0; return put(buf, cap, &u8, 1);
        case REQ_GET_LEVELLER_ENABLE: u8 = c->leveller_config.enabled ? 1 : 0; return put(buf, cap, &u8, 1);
        case REQ_GET_LEVELLER_AMOUNT: return put(buf, cap, &c->leveller_config.amount, 4);
        case REQ_GET_LEVELLER_SPEED: return put(buf, cap, &c->leveller_config.speed, 1);
        case REQ_GET_LEVELLER_MAX_GAIN: return put(buf, cap, &c->leveller_config.max_gain_db, 4);
        case REQ_GET_LEVELLER_LOOKAHEAD: u8 = c->leveller_config.lookahead ? 1 : 0; return put(buf, cap, &u8, 1);
        case REQ_GET_LEVELLER_GATE: return put(buf, cap, &c->leveller_config.gate_threshold_db, 4);
        case REQ_GET_STATUS: {
            if (wValue == 9) {
                uint8_t r[NUM_CHANNELS * 2 + 4];
                for (int i = 0; i < NUM_CHANNELS; i++) { r[i * 2] = c->status.peaks[i] & 0xFF; r[i * 2 + 1] = c->status.peaks[i] >> 8; }
                r[NUM_CHANNELS * 2] = 0; r[NUM_CHANNELS * 2 + 1] = 0;      /* cpu0/cpu1 load: not meaningful here */
                r[NUM_CHANNELS * 2 + 2] = c->status.clip_flags & 0xFF; r[NUM_CHANNELS * 2 + 3] = c->status.clip_flags >> 8;
                return put(buf, cap, r, (int)sizeof r);
            }
            uint32_t resp = 0;
            if (wValue == 0) resp = (uint32_t)c->status.peaks[0] | ((uint32_t)c->status.peaks[1] << 16);
            else if (wValue == 1) resp = (uint32_t)c->status.peaks[2] | ((uint32_t)c->status.peaks[3] << 16);
            else if (wValue == 2) resp = (uint32_t)c->status.peaks[4];
            else if (wValue == 15) resp = c->audio_state.freq;
            return put_u32(buf, cap, resp, 4);
        }
        case REQ_GET_EQ_PARAM: {
            uint8_t ch = (wValue >> 8) & 0xFF, band = (wValue >> 4) & 0x0F, param = wValue & 0x0F;
            if (ch >= NUM_CHANNELS || band >= band_counts[ch]) return -1;
            const EqParamPacket *p = &c->filter_recipes[ch][band];
            uint32_t v = 0;
            if (param == 0) v = p->type; else if (param == 1) memcpy(&v, &p->freq, 4);
            else if (param == 2) memcpy(&v, &p->Q, 4); else if (param == 3) memcpy(&v, &p->gain_db, 4);
            return put_u32(buf, cap, v, 4);
        }
        case REQ_GET_MATRIX_ROUTE: {
            uint8_t in = (wValue >> 8) & 0xFF, out = wValue & 0xFF;
            if (in >= NUM_INPUT_CHANNELS || out >= NUM_OUTPUT_CHANNELS) return -1;
            const MatrixCrosspoint *xp = &c->matrix_mixer.crosspoints[in][out];
            MatrixRoutePacket pk; pk.input = in; pk.output = out; pk.enabled = xp->enabled; pk.phase_invert = xp->phase_invert; pk.gain_db = xp->gain_db;
            return put(buf, cap, &pk, (int)sizeof pk);
        }
        case REQ_GET_OUTPUT_ENABLE: return idx < NUM_OUTPUT_CHANNELS ? put(buf, cap, &c->matrix_mixer.outputs[idx].enabled, 1) : -1;
        case REQ_GET_OUTPUT_GAIN: return idx < NUM_OUTPUT_CHANNELS ? put(buf, cap, &c->matrix_mixer.outputs[idx].gain_db, 4) : -1;
        case REQ_GET_OUTPUT_MUTE: return idx < NUM_OUTPUT_CHANNELS ? put(buf, cap, &c->matrix_mixer.outputs[idx].mute, 1) : -1;
        case REQ_GET_OUTPUT_DELAY: return idx < NUM_OUTPUT_CHANNELS ? put(buf, cap, &c->matrix_mixer.outputs[idx].delay_ms, 4) : -1;
        case REQ_GET_CORE1_MODE: u8 = (uint8_t)c->core1_mode; return put(buf, cap, &u8, 1);
        case REQ_GET_CORE1_CONFLICT: {
            uint8_t conflict = 0;
            if (idx < NUM_OUTPUT_CHANNELS) {
                bool is_pdm = (idx == NUM_OUTPUT_CHANNELS - 1), is_c1 = (idx >= CORE1_EQ_FIRST_OUTPUT && idx <= CORE1_EQ_LAST_OUTPUT);
                if (is_pdm) { for (int i = CORE1_EQ_FIRST_OUTPUT; i <= CORE1_EQ_LAST_OUTPUT; i++) if (c->matrix_mixer.outputs[i].enabled) { conflict = 1; break; } }
                else if (is_c1) { if (c->matrix_mixer.outputs[NUM_OUTPUT_CHANNELS - 1].enabled) conflict = 1; }
            }
            return put(buf, cap, &conflict, 1);
        }
        case REQ_GET_PLATFORM: { uint8_t r[4] = {PICO_RP2350 ? PLATFORM_RP2350 : PLATFORM_RP2040, 0x01, 0x13, NUM_OUTPUT_CHANNELS}; return put(buf, cap, r, 4); }
        case REQ_CLEAR_CLIPS: { uint16_t f = c->status.clip_flags; c->status.clip_flags = 0; return put(buf, cap, &f, 2); }
        case REQ_GET_CHANNEL_NAME: return idx < NUM_CHANNELS ? put(buf, cap, c->channel_names[idx], PRESET_NAME_LEN) : -1;
        case REQ_GET_ALL_PARAMS: if (cap < sizeof(WireBulkParams)) return -2; return orc_collect_bulk(c, buf);
        case REQ_FACTORY_RESET: orc_factory_defaults(c); u8 = 0; return put(buf, cap, &u8, 1);
        case REQ_SET_OUTPUT_TYPE: { /* usb_audio.c:2984-3016 + the deferred switch, main.c:1110-1121 -> :230-424: what the audio
                                     * path sees of it is prepare_pipeline_reset (:279) and the new type */
            uint8_t slot = wValue & 0xFF, type = (wValue >> 8) & 0xFF;
            if (slot >= NUM_SPDIF_INSTANCES) u8 = PIN_CONFIG_INVALID_OUTPUT;
            else if (type > 1) u8 = PIN_CONFIG_INVALID_PIN;
            else {
                u8 = PIN_CONFIG_SUCCESS;
                if (type != c->output_types[slot]) { c->output_types[slot] = type; prepare_pipeline_reset(c, PRESET_MUTE_SAMPLES); }
            }
            return put(buf, cap, &u8, 1);
        }
        case REQ_GET_OUTPUT_TYPE: return idx < NUM_SPDIF_INSTANCES ? put(buf, cap, &c->output_types[idx], 1) : -1;
        default: return -1;
    }
}

void orc_get_status(orc_ctx *c, void *buf) { orc_vendor_get(c, REQ_GET_STATUS, 9, buf, NUM_CHANNELS * 2 + 4); }

void orc_process(orc_ctx *c, const void *pcm, int bit_depth, uint32_t n_blocks, uint32_t block_len,
                 int32_t *pairs /*[pair][n_blocks*block_len][2]*/, int32_t *sub /*[n_blocks*block_len]*/,
                 uint16_t *peaks /*[n_blocks][C] or NULL*/, uint16_t *clip_flags /*[1] or NULL*/) {
    unsigned csr = orc_enter();
    const uint32_t bpf = (bit_depth == 24) ? 6 : 4;
    const size_t total = (size_t)n_blocks * block_len;
    static __thread int32_t pair_blk[NUM_SPDIF_INSTANCES * MAXB * 2], sub_blk[MAXB];
    for (uint32_t k = 0; k < n_blocks; k++) {
        const uint8_t *d = (const uint8_t *)pcm + (size_t)k * block_len * bpf;
        process_packet(c, d, block_len, bit_depth, pair_blk, sub_blk);
        for (int p = 0; p < NUM_SPDIF_INSTANCES; p++)
            memcpy(pairs + ((size_t)p * total + (size_t)k * block_len) * 2, pair_blk + (size_t)p * block_len * 2, (size_t)block_len * 8);
        memcpy(sub + (size_t)k * block_len, sub_blk, (size_t)block_len * 4);
        if (peaks) memcpy(peaks + (size_t)k * NUM_CHANNELS, c->status.peaks, NUM_CHANNELS * 2);
    }
    if (clip_flags) *clip_flags = c->status.clip_flags;
    orc_leave(csr);
}

/* ---- per-band taps of one EQ channel (float flavour; SURVEY.md section 8d parity procedure) ----
 * x[n] through the ten bands of channel `ch` from zero state, band-major (dsp_pipeline.c:281-365): taps[0] = x,
 * taps[b+1] = output of band b alone over the whole buffer.  In the _ref builds the band runs in the reference's own
 * dsp_process_channel_block (all other bands of a scratch copy bypassed); the context is not touched. */
int orc_debug_eq_taps(orc_ctx *c, int ch, const float *x, uint32_t n, float *taps /*[11][n]*/) {
#if PICO_RP2350
    if (ch < 0 || ch >= NUM_CHANNELS) return -1;
    unsigned csr = orc_enter();
    Biquad saved[MAX_BANDS];
    memcpy(saved, c->filters[ch], sizeof saved);
    memcpy(taps, x, (size_t)n * sizeof(float));
    for (int b = 0; b < band_counts[ch]; b++) {
        for (int k = 0; k < MAX_BANDS; k++) {
            c->filters[ch][k] = saved[k];
            c->filters[ch][k].s1 = c->filters[ch][k].s2 = 0.0f; c->filters[ch][k].svic1eq = c->filters[ch][k].svic2eq = 0.0f;
            if (k != b) c->filters[ch][k].bypass = true;
        }
        float *out = taps + (size_t)(b + 1) * n;
        memcpy(out, taps + (size_t)b * n, (size_t)n * sizeof(float));
        for (uint32_t i = 0; i < n; i += 96) eq_block(c, ch, out + i, (n - i) < 96 ? (n - i) : 96);
    }
    memcpy(c->filters[ch], saved, sizeof saved);
    orc_leave(csr);
    return 0;
#else
    (void)c; (void)ch; (void)x; (void)n; (void)taps;
    return -1;
#endif
}

/* ---- state taps for tests ---- */
const void *orc_tap(orc_ctx *c, int what, int *bytes) {
    switch (what) {
        case 0: *bytes = (int)sizeof(c->filters); return c->filters;
        case 1: *bytes = (int)sizeof(c->loudness_table); return c->loudness_table;
        case 2: *bytes = (int)sizeof(c->crossfeed_state); return &c->crossfeed_state;
        case 3: *bytes = (int)sizeof(c->leveller_coeffs); return &c->leveller_coeffs;
        case 4: *bytes = (int)sizeof(c->matrix_mixer); return &c->matrix_mixer;
        case 5: *bytes = (int)sizeof(c->channel_delay_samples); return c->channel_delay_samples;
        case 6: *bytes = (int)sizeof(c->leveller_state); return &c->leveller_state;
        case 7: *bytes = (int)sizeof(c->filter_recipes); return c->filter_recipes;
        case 8: *bytes = (int)sizeof(c->delay_lines); return c->delay_lines;
        case 9: *bytes = (int)sizeof(c->channel_bypassed); return c->channel_bypassed;
#if PICO_RP2350
        case 10: *bytes = (int)sizeof(c->loudness_state); return c->loudness_state;
#else
        case 10: *bytes = (int)sizeof(c->loudness_biquads); return c->loudness_biquads;
#endif
        default: *bytes = 0; return NULL;
    }
}
int orc_scalar(orc_ctx *c, int what) {
    switch (what) {
        case 0: return c->loudness_row;
        case 1: return (int)c->core1_mode;
        case 2: return c->any_delay_active;
        case 3: return (int)c->delay_write_idx;
        case 4: return c->crossfeed_bypassed;
        case 5: return c->leveller_bypassed;
        case 6: return c->audio_state.vol_mul;
        case 7: return c->master_volume_q15;
        case 8: return c->global_preamp_mul[0];
        case 9: return c->global_preamp_mul[1];
        case 10: return (int)c->audio_state.freq;
        case 11: return c->preset_loading;
        default: return 0;
    }
}
float orc_scalar_f(orc_ctx *c, int what) {
    switch (what) {
        case 0: return c->master_volume_linear;
        case 1: return c->global_preamp_linear[0];
        case 2: return c->global_preamp_linear[1];
        case 3: return c->preset_mute_smooth_gain;
        case 4: return c->master_volume_db;
        default: return 0.0f;
    }
}
