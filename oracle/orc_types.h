/*
 * orc_types.h — data-structure definitions for the STANDALONE oracle build.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing under oracle/ is linked into
 * the product library.
 *
 * These restate the firmware's DSP-config / preset / wire structures so that the same
 * binary blobs drive the oracle and the HIP path.  Field names follow the reference so the
 * shared orchestrator (orc_chain.c) compiles unchanged against the reference's own headers
 * in the `_ref` build (ORC_USE_REF=1), where this file is NOT included.
 *
 * Reference definitions restated here (firmware/DSPi/…):
 *   config.h:294-329   channel / output / band counts per flavour
 *   config.h:383-415   MatrixCrosspoint, OutputChannel, MatrixMixer, MatrixRoutePacket
 *   config.h:417-453   Biquad (both flavours), FilterType, EqParamPacket
 *   config.h:547-567   clip_s24, fast_mul_q15
 *   loudness.h:10-23   LoudnessCoeffs, LoudnessSvfState
 *   crossfeed.h:6-59   presets, limits, CrossfeedConfig, CrossfeedState
 *   leveller.h:34-134  constants, LevellerConfig, LevellerCoeffs, LevellerState
 *   bulk_params.h:27-205  wire format sections
 * Layout is asserted in orc_chain.c (sizeof checks: Biquad 68/32, wire blob 2896, …).
 */
#ifndef ORC_TYPES_H
#define ORC_TYPES_H

#include <stdint.h>
#include <stdbool.h>
#include <limits.h>

#ifndef PICO_RP2350
#error "build with -DPICO_RP2350=1 (float flavour) or =0 (Q28 flavour)"
#endif

#define PACKED __attribute__((packed))

/* ---- sizes ---------------------------------------------------------------------- */
#define NUM_INPUT_CHANNELS 2
#define MAX_BANDS 12
#if PICO_RP2350
#define NUM_CHANNELS 11
#define NUM_OUTPUT_CHANNELS 9
#define NUM_SPDIF_INSTANCES 4
#define NUM_PIN_OUTPUTS 5
#define MAX_DELAY_SAMPLES 4096
#define CORE1_EQ_LAST_OUTPUT 7
#else
#define NUM_CHANNELS 7
#define NUM_OUTPUT_CHANNELS 5
#define NUM_SPDIF_INSTANCES 2
#define NUM_PIN_OUTPUTS 3
#define MAX_DELAY_SAMPLES 2048
#define CORE1_EQ_LAST_OUTPUT 3
#endif
#define CORE1_EQ_FIRST_OUTPUT 2
#define MAX_DELAY_MASK (MAX_DELAY_SAMPLES - 1)
#define NUM_DELAY_CHANNELS NUM_OUTPUT_CHANNELS
#define CH_MASTER_LEFT 0
#define CH_MASTER_RIGHT 1
#define CH_OUT_1 2
#define CH_OUT_SUB (NUM_CHANNELS - 1)
#define FILTER_SHIFT 28
#define SUB_ALIGN_SAMPLES 128            /* 384 - 2048/8, config.h:93-95 */
#define CLIP_THRESH_F 1.001f
#define CLIP_THRESH_Q28 ((1 << 28) + 268)
#define PRESET_NAME_LEN 32
#define PRESET_SLOTS 10
#define PRESET_MUTE_SAMPLES 256          /* flash_storage.h:114 */

#define PLATFORM_RP2040 0
#define PLATFORM_RP2350 1
#define FW_VERSION_MAJOR 1
#define FW_VERSION_MINOR 1

#define MASTER_VOL_MUTE_DB (-128.0f)
#define MASTER_VOL_MIN_DB (-127.0f)
#define MASTER_VOL_MAX_DB (0.0f)
#define MASTER_VOL_DEFAULT_DB (-20.0f)
#define MASTER_VOLUME_MODE_INDEPENDENT 0
#define MASTER_VOLUME_MODE_WITH_PRESET 1

#define PRESET_OK 0x00
#define PRESET_ERR_INVALID_SLOT 0x01
#define PRESET_ERR_SLOT_EMPTY 0x02
#define PRESET_ERR_CRC 0x03

/* default GPIO map (only round-tripped through blobs) */
#define PICO_AUDIO_SPDIF_PIN 6
#define PICO_SPDIF_PIN_2 7
#define PICO_SPDIF_PIN_3 8
#define PICO_SPDIF_PIN_4 9
#define PICO_PDM_PIN 10
#define PICO_I2S_BCK_PIN 14
#define PICO_I2S_MCK_PIN 13

typedef enum { CORE1_MODE_IDLE = 0, CORE1_MODE_PDM = 1, CORE1_MODE_EQ_WORKER = 2 } Core1Mode;

/* ---- matrix mixer ------------------------------------------------------------------ */
typedef struct PACKED { uint8_t enabled, phase_invert, reserved[2]; float gain_db, gain_linear; } MatrixCrosspoint;
typedef struct PACKED {
    uint8_t enabled, mute, reserved[2];
    float gain_db, gain_linear, delay_ms;
    int32_t delay_samples;
} OutputChannel;
typedef struct {
    MatrixCrosspoint crosspoints[NUM_INPUT_CHANNELS][NUM_OUTPUT_CHANNELS];
    OutputChannel outputs[NUM_OUTPUT_CHANNELS];
} MatrixMixer;
typedef struct PACKED { uint8_t input, output, enabled, phase_invert; float gain_db; } MatrixRoutePacket;

/* ---- EQ ---------------------------------------------------------------------------- */
#if PICO_RP2350
typedef struct {
    float b0, b1, b2, a1, a2;
    float s1, s2;
    float sva1, sva2, sva3;
    float svm0, svm1, svm2;
    float svic1eq, svic2eq;
    uint32_t svf_type;
    bool use_svf;
    bool bypass;
} Biquad;
#else
typedef struct { int32_t b0, b1, b2, a1, a2; int32_t s1, s2; bool bypass; } Biquad;
#endif

enum FilterType { FILTER_FLAT = 0, FILTER_PEAKING, FILTER_LOWSHELF, FILTER_HIGHSHELF, FILTER_LOWPASS, FILTER_HIGHPASS };

typedef struct PACKED { uint8_t channel, band, type, reserved; float freq, Q, gain_db; } EqParamPacket;

/* ---- loudness ---------------------------------------------------------------------- */
#define LOUDNESS_BIQUAD_COUNT 2
#define LOUDNESS_VOL_STEPS 61
#if PICO_RP2350
typedef struct { float sva1, sva2, sva3; float svm0, svm1, svm2; bool bypass; } LoudnessCoeffs;
typedef struct { float ic1eq, ic2eq; } LoudnessSvfState;
#else
typedef struct { int32_t b0, b1, b2, a1, a2; bool bypass; } LoudnessCoeffs;
#endif

/* ---- crossfeed --------------------------------------------------------------------- */
#define CROSSFEED_PRESET_DEFAULT 0
#define CROSSFEED_PRESET_CUSTOM 3
#define CROSSFEED_FREQ_MIN 500.0f
#define CROSSFEED_FREQ_MAX 2000.0f
#define CROSSFEED_FEED_MIN 0.0f
#define CROSSFEED_FEED_MAX 15.0f
#define CROSSFEED_ITD_SEC 0.000220f
typedef struct { bool enabled, itd_enabled; uint8_t preset; float custom_fc, custom_feed_db; } CrossfeedConfig;
#if PICO_RP2350
typedef float xf_word;
#else
typedef int32_t xf_word;
#endif
typedef struct { xf_word lp_a0, lp_b1, lp_state_L, lp_state_R, ap_a, ap_state_L, ap_state_R; } CrossfeedState;

/* ---- leveller ---------------------------------------------------------------------- */
#define LEVELLER_LOOKAHEAD_SAMPLES 480
#define LEVELLER_SPEED_SLOW 0
#define LEVELLER_SPEED_MEDIUM 1
#define LEVELLER_SPEED_COUNT 3
#define LEVELLER_AMOUNT_MIN 0.0f
#define LEVELLER_AMOUNT_MAX 100.0f
#define LEVELLER_MAX_GAIN_MIN 0.0f
#define LEVELLER_MAX_GAIN_MAX 35.0f
#define LEVELLER_GATE_MIN (-96.0f)
#define LEVELLER_GATE_MAX 0.0f
#define LEVELLER_THRESHOLD_DB (-20.0f)
#define LEVELLER_KNEE_WIDTH_DB 6.0f
#define LEVELLER_LIMITER_CEIL 0.70795f
#define LEVELLER_DEFAULT_ENABLED false
#define LEVELLER_DEFAULT_AMOUNT 50.0f
#define LEVELLER_DEFAULT_SPEED LEVELLER_SPEED_SLOW
#define LEVELLER_DEFAULT_MAX_GAIN_DB 15.0f
#define LEVELLER_DEFAULT_LOOKAHEAD true
#define LEVELLER_DEFAULT_GATE_DB (-96.0f)
typedef struct { bool enabled; float amount; uint8_t speed; float max_gain_db; bool lookahead; float gate_threshold_db; } LevellerConfig;
typedef struct {
    float alpha_rms, alpha_attack, alpha_release;
    float threshold_db, ratio, knee_width_db, makeup_db, gate_threshold_db, max_gain_db;
} LevellerCoeffs;
#if PICO_RP2350
typedef struct {
    float env_sq_l, env_sq_r;
    float gain_smooth_db, gain_linear, gain_prev_linear;
    float lookahead_buf[2][LEVELLER_LOOKAHEAD_SAMPLES];
    uint32_t la_write_idx;
} LevellerState;
#else
typedef struct {
    int32_t env_sq_l, env_sq_r;
    float gain_smooth_db;
    int32_t gain_q28, gain_prev_q28;
    int32_t lookahead_buf[2][LEVELLER_LOOKAHEAD_SAMPLES];
    uint32_t la_write_idx;
} LevellerState;
#endif

/* ---- status ------------------------------------------------------------------------ */
typedef struct { uint16_t peaks[NUM_CHANNELS]; uint8_t cpu0_load, cpu1_load; uint16_t clip_flags; } SystemStatusPacket;

/* ---- vendor request codes (config.h:111-251), DSP subset --------------------------- */
enum {
    REQ_SET_EQ_PARAM = 0x42, REQ_GET_EQ_PARAM, REQ_SET_PREAMP, REQ_GET_PREAMP, REQ_SET_BYPASS, REQ_GET_BYPASS,
    REQ_SET_DELAY, REQ_GET_DELAY,
    REQ_GET_STATUS = 0x50,
    REQ_FACTORY_RESET = 0x53,
    REQ_SET_CHANNEL_GAIN = 0x54, REQ_GET_CHANNEL_GAIN, REQ_SET_CHANNEL_MUTE, REQ_GET_CHANNEL_MUTE,
    REQ_SET_LOUDNESS, REQ_GET_LOUDNESS, REQ_SET_LOUDNESS_REF, REQ_GET_LOUDNESS_REF,
    REQ_SET_LOUDNESS_INTENSITY, REQ_GET_LOUDNESS_INTENSITY,
    REQ_SET_CROSSFEED, REQ_GET_CROSSFEED, REQ_SET_CROSSFEED_PRESET, REQ_GET_CROSSFEED_PRESET,
    REQ_SET_CROSSFEED_FREQ, REQ_GET_CROSSFEED_FREQ, REQ_SET_CROSSFEED_FEED, REQ_GET_CROSSFEED_FEED,
    REQ_SET_CROSSFEED_ITD, REQ_GET_CROSSFEED_ITD,
    REQ_SET_MATRIX_ROUTE = 0x70, REQ_GET_MATRIX_ROUTE, REQ_SET_OUTPUT_ENABLE, REQ_GET_OUTPUT_ENABLE,
    REQ_SET_OUTPUT_GAIN, REQ_GET_OUTPUT_GAIN, REQ_SET_OUTPUT_MUTE, REQ_GET_OUTPUT_MUTE,
    REQ_SET_OUTPUT_DELAY, REQ_GET_OUTPUT_DELAY, REQ_GET_CORE1_MODE, REQ_GET_CORE1_CONFLICT,
    REQ_GET_PLATFORM = 0x7F,
    REQ_CLEAR_CLIPS = 0x83,
    REQ_SET_CHANNEL_NAME = 0x9B, REQ_GET_CHANNEL_NAME = 0x9C,
    REQ_GET_ALL_PARAMS = 0xA0, REQ_SET_ALL_PARAMS = 0xA1,
    REQ_SET_LEVELLER_ENABLE = 0xB4, REQ_GET_LEVELLER_ENABLE, REQ_SET_LEVELLER_AMOUNT, REQ_GET_LEVELLER_AMOUNT,
    REQ_SET_LEVELLER_SPEED, REQ_GET_LEVELLER_SPEED, REQ_SET_LEVELLER_MAX_GAIN, REQ_GET_LEVELLER_MAX_GAIN,
    REQ_SET_LEVELLER_LOOKAHEAD, REQ_GET_LEVELLER_LOOKAHEAD, REQ_SET_LEVELLER_GATE, REQ_GET_LEVELLER_GATE,
    REQ_SET_OUTPUT_TYPE = 0xC0, REQ_GET_OUTPUT_TYPE = 0xC1,      /* config.h:192-193 */
    REQ_SET_PREAMP_CH = 0xD0, REQ_GET_PREAMP_CH, REQ_SET_MASTER_VOLUME, REQ_GET_MASTER_VOLUME,
    REQ_SET_MASTER_VOLUME_MODE, REQ_GET_MASTER_VOLUME_MODE, REQ_SAVE_MASTER_VOLUME, REQ_GET_SAVED_MASTER_VOLUME,
};

/* pin / output-type request status (config.h:278-287) */
enum { PIN_CONFIG_SUCCESS = 0, PIN_CONFIG_INVALID_PIN = 1, PIN_CONFIG_INVALID_OUTPUT = 3, OUTPUT_TYPE_SPDIF = 0, OUTPUT_TYPE_I2S = 1 };

/* ---- integer helpers (Q28 flavour) -------------------------------------------------- */
#if !PICO_RP2350
static inline int32_t clip_s24(int32_t x) { return x > 0x7FFFFF ? 0x7FFFFF : (x < -0x800000 ? -0x800000 : x); }
/* (sample*gain)>>15 from 16-bit partial products, every step wrapping mod 2^32 (config.h:556-567) */
static inline int32_t fast_mul_q15(int32_t sample, int32_t gain) {
    int32_t sh = sample >> 16, gh = gain >> 16;
    uint32_t sl = (uint16_t)sample, gl = (uint16_t)gain;
    uint32_t hh = (uint32_t)(sh * gh);
    uint32_t mid = (uint32_t)sh * gl + sl * (uint32_t)gh;
    uint32_t ll = sl * gl;
    return (int32_t)((hh << 17) + (mid << 1) + (ll >> 15));
}
#endif

/* ---- bulk wire format (bulk_params.h) ----------------------------------------------- */
#define WIRE_MAX_CHANNELS 11
#define WIRE_MAX_OUTPUT_CHANNELS 9
#define WIRE_MAX_INPUT_CHANNELS 2
#define WIRE_MAX_BANDS 12
#define WIRE_MAX_PIN_OUTPUTS 5
#define WIRE_NAME_LEN 32
#define WIRE_FORMAT_VERSION 6
#define WIRE_MAX_SPDIF_INSTANCES 4
#define WIRE_PLATFORM_RP2040 0
#define WIRE_PLATFORM_RP2350 1

typedef struct PACKED {
    uint8_t format_version, platform_id, num_channels, num_output_channels, num_input_channels, max_bands;
    uint16_t payload_length, fw_version_major, fw_version_minor;
    uint32_t reserved;
} WireHeader;
typedef struct PACKED { float preamp_gain_db; uint8_t bypass, loudness_enabled, reserved[2]; float loudness_ref_spl, loudness_intensity_pct; } WireGlobalParams;
typedef struct PACKED { uint8_t enabled, preset, itd_enabled, reserved; float custom_fc, custom_feed_db; uint32_t reserved2; } WireCrossfeedParams;
typedef struct PACKED { float gain_db[3]; uint8_t mute[3], reserved; } WireLegacyChannels;
typedef struct PACKED { float delay_ms[WIRE_MAX_CHANNELS]; } WireChannelDelays;
typedef struct PACKED { uint8_t enabled, phase_invert, reserved[2]; float gain_db; } WireCrosspoint;
typedef struct PACKED { uint8_t enabled, mute, reserved[2]; float gain_db, delay_ms; } WireOutputChannel;
typedef struct PACKED { uint8_t num_pin_outputs, pins[WIRE_MAX_PIN_OUTPUTS], reserved[2]; } WirePinConfig;
typedef struct PACKED { uint8_t type, reserved[3]; float freq, q, gain_db; } WireBandParams;
typedef struct PACKED { char names[WIRE_MAX_CHANNELS][WIRE_NAME_LEN]; } WireChannelNames;
typedef struct PACKED { uint8_t output_types[WIRE_MAX_SPDIF_INSTANCES], bck_pin, mck_pin, mck_enabled, mck_multiplier, reserved[8]; } WireI2SConfig;
typedef struct PACKED { uint8_t enabled, speed, lookahead, reserved; float amount, max_gain_db, gate_threshold_db; } WireLevellerConfig;
typedef struct PACKED { float preamp_db[WIRE_MAX_INPUT_CHANNELS]; uint8_t reserved[8]; } WirePreampConfig;
typedef struct PACKED { float master_volume_db; uint8_t reserved[12]; } WireMasterVolume;
typedef struct PACKED {
    WireHeader header;
    WireGlobalParams global;
    WireCrossfeedParams crossfeed;
    WireLegacyChannels legacy;
    WireChannelDelays delays;
    WireCrosspoint crosspoints[WIRE_MAX_INPUT_CHANNELS][WIRE_MAX_OUTPUT_CHANNELS];
    WireOutputChannel outputs[WIRE_MAX_OUTPUT_CHANNELS];
    WirePinConfig pins;
    WireBandParams eq[WIRE_MAX_CHANNELS][WIRE_MAX_BANDS];
    WireChannelNames channel_names;
    WireI2SConfig i2s_config;
    WireLevellerConfig leveller;
    WirePreampConfig preamp;
    WireMasterVolume master_volume;
} WireBulkParams;

#endif /* ORC_TYPES_H */
