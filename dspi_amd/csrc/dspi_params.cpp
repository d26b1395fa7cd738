// dspi_params.cpp — host-side parameter model + coefficient design.  See dspi_params.h.
//
// Every routine names the firmware code whose observable behaviour it reproduces.  Arithmetic
// is single precision with the reference's expression order (build: -ffp-contract=off) and runs
// under MXCSR FTZ|DAZ like the MCU's FPSCR.FZ (main.c:593-600), so that the images built here
// are bit-identical to the oracle's (tests/test_host_params.py).
#include "dspi_params.h"

#include <math.h>
#include <string.h>
#include <vector>
#include <xmmintrin.h>

namespace dspi {

namespace {

constexpr float kPi = 3.1415926535f;
constexpr int kFilterShift = 28;
constexpr float kMasterMuteDb = -128.0f, kMasterMaxDb = 0.0f, kMasterDefaultDb = -20.0f;
constexpr uint32_t kPresetMuteSamples = 256;   // flash_storage.h:114

struct FtzScope {   // FPSCR.FZ analogue for the duration of a parameter call
    unsigned saved;
    FtzScope() : saved(_mm_getcsr()) { _mm_setcsr(saved | 0x8040u); }
    ~FtzScope() { _mm_setcsr(saved); }
};

// ---- vendor request codes (config.h:111-251) ----
enum : uint8_t {
    REQ_SET_EQ_PARAM = 0x42, REQ_GET_EQ_PARAM = 0x43, REQ_SET_PREAMP = 0x44, REQ_GET_PREAMP = 0x45, REQ_SET_BYPASS = 0x46,
    REQ_GET_BYPASS = 0x47, REQ_SET_DELAY = 0x48, REQ_GET_DELAY = 0x49, REQ_GET_STATUS = 0x50, REQ_FACTORY_RESET = 0x53,
    REQ_SET_CHANNEL_GAIN = 0x54, REQ_GET_CHANNEL_GAIN = 0x55, REQ_SET_CHANNEL_MUTE = 0x56, REQ_GET_CHANNEL_MUTE = 0x57,
    REQ_SET_LOUDNESS = 0x58, REQ_GET_LOUDNESS = 0x59, REQ_SET_LOUDNESS_REF = 0x5A, REQ_GET_LOUDNESS_REF = 0x5B,
    REQ_SET_LOUDNESS_INTENSITY = 0x5C, REQ_GET_LOUDNESS_INTENSITY = 0x5D, REQ_SET_CROSSFEED = 0x5E, REQ_GET_CROSSFEED = 0x5F,
    REQ_SET_CROSSFEED_PRESET = 0x60, REQ_GET_CROSSFEED_PRESET = 0x61, REQ_SET_CROSSFEED_FREQ = 0x62, REQ_GET_CROSSFEED_FREQ = 0x63,
    REQ_SET_CROSSFEED_FEED = 0x64, REQ_GET_CROSSFEED_FEED = 0x65, REQ_SET_CROSSFEED_ITD = 0x66, REQ_GET_CROSSFEED_ITD = 0x67,
    REQ_SET_MATRIX_ROUTE = 0x70, REQ_GET_MATRIX_ROUTE = 0x71, REQ_SET_OUTPUT_ENABLE = 0x72, REQ_GET_OUTPUT_ENABLE = 0x73,
    REQ_SET_OUTPUT_GAIN = 0x74, REQ_GET_OUTPUT_GAIN = 0x75, REQ_SET_OUTPUT_MUTE = 0x76, REQ_GET_OUTPUT_MUTE = 0x77,
    REQ_SET_OUTPUT_DELAY = 0x78, REQ_GET_OUTPUT_DELAY = 0x79, REQ_GET_CORE1_MODE = 0x7A, REQ_GET_CORE1_CONFLICT = 0x7B,
    REQ_GET_PLATFORM = 0x7F, REQ_CLEAR_CLIPS = 0x83, REQ_SET_CHANNEL_NAME = 0x9B, REQ_GET_CHANNEL_NAME = 0x9C,
    REQ_GET_ALL_PARAMS = 0xA0, REQ_SET_ALL_PARAMS = 0xA1,
    REQ_SET_LEVELLER_ENABLE = 0xB4, REQ_GET_LEVELLER_ENABLE = 0xB5, REQ_SET_LEVELLER_AMOUNT = 0xB6, REQ_GET_LEVELLER_AMOUNT = 0xB7,
    REQ_SET_LEVELLER_SPEED = 0xB8, REQ_GET_LEVELLER_SPEED = 0xB9, REQ_SET_LEVELLER_MAX_GAIN = 0xBA, REQ_GET_LEVELLER_MAX_GAIN = 0xBB,
    REQ_SET_LEVELLER_LOOKAHEAD = 0xBC, REQ_GET_LEVELLER_LOOKAHEAD = 0xBD, REQ_SET_LEVELLER_GATE = 0xBE, REQ_GET_LEVELLER_GATE = 0xBF,
    REQ_SET_PREAMP_CH = 0xD0, REQ_GET_PREAMP_CH = 0xD1, REQ_SET_MASTER_VOLUME = 0xD2, REQ_GET_MASTER_VOLUME = 0xD3,
    REQ_SET_OUTPUT_TYPE = 0xC0, REQ_GET_OUTPUT_TYPE = 0xC1,
    REQ_SET_MASTER_VOLUME_MODE = 0xD4, REQ_GET_MASTER_VOLUME_MODE = 0xD5, REQ_SAVE_MASTER_VOLUME = 0xD6, REQ_GET_SAVED_MASTER_VOLUME = 0xD7,
};

// ---- wire structures (bulk_params.h:42-205) ----
#pragma pack(push, 1)
struct WHeader { uint8_t format_version, platform_id, num_channels, num_output_channels, num_input_channels, max_bands; uint16_t payload_length, fw_major, fw_minor; uint32_t reserved; };
struct WGlobal { float preamp_gain_db; uint8_t bypass, loudness_enabled, reserved[2]; float loudness_ref_spl, loudness_intensity_pct; };
struct WCrossfeed { uint8_t enabled, preset, itd_enabled, reserved; float custom_fc, custom_feed_db; uint32_t reserved2; };
struct WLegacy { float gain_db[3]; uint8_t mute[3], reserved; };
struct WXp { uint8_t enabled, phase_invert, reserved[2]; float gain_db; };
struct WOut { uint8_t enabled, mute, reserved[2]; float gain_db, delay_ms; };
struct WPins { uint8_t num_pin_outputs, pins[5], reserved[2]; };
struct WBand { uint8_t type, reserved[3]; float freq, q, gain_db; };
struct WI2S { uint8_t output_types[4], bck_pin, mck_pin, mck_enabled, mck_multiplier, reserved[8]; };
struct WLeveller { uint8_t enabled, speed, lookahead, reserved; float amount, max_gain_db, gate_threshold_db; };
struct WPreamp { float preamp_db[2]; uint8_t reserved[8]; };
struct WMaster { float master_volume_db; uint8_t reserved[12]; };
struct WireBulk {
    WHeader header; WGlobal global; WCrossfeed crossfeed; WLegacy legacy; float delays[11];
    WXp crosspoints[2][9]; WOut outputs[9]; WPins pins; WBand eq[11][12]; char names[11][32];
    WI2S i2s; WLeveller leveller; WPreamp preamp; WMaster master;
};
#pragma pack(pop)
static_assert(sizeof(WireBulk) == 2896, "WireBulkParams is 2896 bytes (bulk_params.h:205)");

constexpr uint32_t kSlotMagic = 0x44535033u;   // "DSP3", flash_storage.c:67
constexpr uint16_t kSlotVersion = 12;

// PresetSlot field offsets depend on the flavour's channel counts; walk the image with a cursor
// instead of declaring two structs (layout: flash_storage.c:136-189).
struct SlotCursor {
    const uint8_t *rd;
    uint8_t *wr;
    size_t off;
    explicit SlotCursor(const void *p) : rd((const uint8_t *)p), wr(nullptr), off(0) {}
    explicit SlotCursor(void *p, int) : rd((const uint8_t *)p), wr((uint8_t *)p), off(0) {}
    template <class T> T get() { T v; memcpy(&v, rd + off, sizeof(T)); off += sizeof(T); return v; }
    template <class T> void put(const T &v) { memcpy(wr + off, &v, sizeof(T)); off += sizeof(T); }
    void skip(size_t n) { off += n; }
};

uint32_t crc32_edb88320(const uint8_t *d, size_t n) {   // flash_storage.c:282-291
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        crc ^= d[i];
        for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

float db_to_linear_preset(float db) {   // flash_storage.c:302-306 (powf)
    if (db <= -120.0f) return 0.0f;
    if (db >= 80.0f) db = 80.0f;
    return powf(10.0f, db / 20.0f);
}

// The firmware's float contract (DSPI_FLOAT_CONTRACT_FMA): GNU C's default -ffp-contract=fast on the Cortex-M33 turns
// a*b + c into one fused operation; which pairs is a property of GCC's GIMPLE pass (read off -fdump-tree-optimized of the
// reference sources; DESIGN.md section 5 lists them with their source lines).  Every mad() in this file is one
// of those statements: with the contract off it is the reference's expression with separate roundings, with it on the
// fused form.  (-a)*b + c is c - a*b exactly, so FNMA/FMS need no form of their own.
inline float mad(bool fma, float a, float b, float c) { return fma ? fmaf(a, b, c) : a * b + c; }

float db_to_linear_bulk(float db, bool fma) {     // bulk_params.c:49-56 (4-term Taylor, clamped)
    if (db == 0.0f) return 1.0f;
    if (db < -60.0f) db = -60.0f;
    if (db > 20.0f) db = 20.0f;
    float x = db * 0.1151292546f;
    float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
    float lin = mad(fma, x4, 0.0416667f, mad(fma, x3, 0.1666667f, mad(fma, x2, 0.5f, 1.0f + x)));
    return (lin < 0.0f) ? 0.0f : lin;
}

bool gpio_ok(int flavor, uint8_t pin) {   // bulk_params.c:260-267 / flash_storage.c:677-684
    bool v = (pin <= 29) && (pin != 12) && !(pin >= 23 && pin <= 25);
    if (!flavor && pin > 28) v = false;
    return v;
}

void default_pins(int flavor, uint8_t *p) {
    if (flavor) { p[0] = 6; p[1] = 7; p[2] = 8; p[3] = 9; p[4] = 10; }
    else { p[0] = 6; p[1] = 7; p[2] = 10; }
}

void default_name(int flavor, int ch, char *buf) {   // usb_audio.c:216-235
    static const char *f32[] = {"USB L", "USB R", "SPDIF 1 L", "SPDIF 1 R", "SPDIF 2 L", "SPDIF 2 R", "SPDIF 3 L", "SPDIF 3 R", "SPDIF 4 L", "SPDIF 4 R", "PDM"};
    static const char *q28[] = {"USB L", "USB R", "SPDIF 1 L", "SPDIF 1 R", "SPDIF 2 L", "SPDIF 2 R", "PDM"};
    memset(buf, 0, 32);
    strncpy(buf, flavor ? f32[ch] : q28[ch], 31);
}

// UAC1 volume table, usb_audio.c:409-420.  Entry 60 (0 dB) is 0x8000, stored into an int16_t.
const uint16_t kDbToVol[61] = {
    0x0000, 0x0025, 0x0029, 0x002e, 0x0034, 0x003a, 0x0041, 0x0049, 0x0052, 0x005c, 0x0068, 0x0074, 0x0082, 0x0092, 0x00a4, 0x00b8,
    0x00cf, 0x00e8, 0x0104, 0x0124, 0x0148, 0x0170, 0x019d, 0x01cf, 0x0207, 0x0247, 0x028e, 0x02de, 0x0337, 0x039c, 0x040c, 0x048b,
    0x0519, 0x05b8, 0x066a, 0x0733, 0x0814, 0x0910, 0x0a2b, 0x0b68, 0x0ccd, 0x0e5d, 0x101d, 0x1215, 0x1449, 0x16c3, 0x198a, 0x1ca8,
    0x2027, 0x2413, 0x287a, 0x2d6b, 0x32f5, 0x392d, 0x4027, 0x47fb, 0x50c3, 0x5a9e, 0x65ad, 0x7215, 0x8000};

uint8_t volume_index(int16_t v) {   // usb_audio.c:430-433
    v = (int16_t)(v + 60 * 256);
    if (v < 0) v = 0;
    if (v >= 61 * 256) v = 61 * 256 - 1;
    return (uint8_t)(((uint16_t)v) >> 8);
}

float rd_f32(const uint8_t *p) { float f; memcpy(&f, p, 4); return f; }

}  // namespace

int32_t f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

// ============================================================================================
// construction / boot
// ============================================================================================
Params::Params(int fl, bool fma, bool first_boot_) {
    memset((void *)this, 0, sizeof(*this));
    first_boot = first_boot_;
    flavor = fl;
    fma_contract = fma && fl;            // the RP2040 has no FPU, hence nothing to contract
    StateMap m = make_state_map(fl);
    n_ch = m.n_ch; n_out = m.n_out; n_pairs = m.n_pairs; max_delay = m.max_delay;
    n_pins = fl ? 5 : 3;
    boot();
}

int Params::slot_size() const { return flavor ? 2864 : 1840; }

// Power-on sequence: global initialisers (usb_audio.c:47-214), usb_sound_card_init (:3251-3270, :3382-3385), core0_init
// (main.c:645-696) with preset_boot_load on `flash` (a 48 KB preset area; nullptr: an erased one, or — !first_boot — one that holds a
// directory and no selected preset), and the first main-loop pass, which runs the rate change that _audio_reconfigure() queued for
// audio_state.freq = 44100.  Returns preset_boot_load's selection (flash_select's codes; 48 without a flash image).
int Params::boot(const void *flash) {
    FtzScope ftz;
    freq = 44100;
    master_db = kMasterDefaultDb; master_linear = 0.1f; master_q15 = 3277;
    for (int i = 0; i < 2; i++) { preamp_mul[i] = 1 << 28; preamp_linear[i] = 1.0f; }
    for (int i = 0; i < 3; i++) { legacy_gain_mul[i] = 32768; legacy_gain_linear[i] = 1.0f; }
    loudness_ref_spl = 83.0f; loudness_intensity_pct = 100.0f; loud_row = -1;
    xfeed_cfg.itd_enabled = true; xfeed_cfg.custom_fc = 700.0f; xfeed_cfg.custom_feed_db = 4.5f;
    crossfeed_bypassed = true; leveller_bypassed = true;
    lev_cfg.enabled = false; lev_cfg.amount = 50.0f; lev_cfg.speed = 0; lev_cfg.max_gain_db = 15.0f; lev_cfg.lookahead = true; lev_cfg.gate_threshold_db = -96.0f;
    default_pins(flavor, pins);
    i2s_bck_pin = 14; i2s_mck_pin = 13; i2s_mck_multiplier = 128;
    dir_master_volume_mode = 0; dir_master_volume_db = kMasterDefaultDb;   // flash_storage.c:450-451
    apply_factory_defaults();
    recalc_all_filters(48000.0f);
    set_volume(0);
    // core0_init: preset_boot_load — the slot goes into the live parameters and nothing else happens (apply_slot_to_live,
    // flash_storage.c:1047-1082: no mute, the delay lines are not touched); then the filters and delays of the loaded (or default) preset
    int sel = 48;
    bool boot_wrote = false;      // a boot that WRITES the flash arms the mute like the first boot: no directory (legacy migration or a fresh
                                  // directory, flash_storage.c:1084-1104), or a v1 directory, which dir_load_cache persists as v2 (:391-414)
    if (flash) {
        sel = flash_select(flash, true);
        FlashDirectory fd;
        boot_wrote = !parse_flash_directory(flash, kFlashDumpBytes, fd) || fd.version == 1;
    }
    recalc_all_filters(48000.0f);
    update_delay_samples(48000.0f);
    // slots the preset saved as I2S are converted by process_type_switches before Core 1 starts (main.c:651-684), and that arms the
    // pipeline mute like every type switch (prepare_pipeline_reset(PRESET_MUTE_SAMPLES), :279)
    bool boot_i2s = false;
    for (int i = 0; i < n_pairs; i++) boot_i2s |= output_types[i] != 0;
    loudness_recompute(48000.0f); loud_pending = false;
    if (loudness_enabled && loud_table_valid) set_volume(volume);      // main.c:688-691
    leveller_design(48000.0f); lev_pending = false; lev_reset_pending = false;
    leveller_bypassed = !lev_cfg.enabled;
    transition_core1();
    rate_changed(freq);
    service();
    memset(&ops, 0, sizeof(ops));     // nothing has run yet: state starts zeroed by the context
    // First boot on an erased flash writes the fresh directory (preset_boot_load -> dir_flush, flash_storage.c:1097-1100),
    // and every flash_write_sector re-arms the preset mute for flash_mute_hold_samples() = max(10 ms, 512) samples at the
    // power-on rate of 44.1 kHz (:272-276, :347-348): a new device starts with 512 muted samples and the fade-in.  A device whose
    // flash already holds a directory writes nothing at boot and does not mute (DSPI_BOOT_POPULATED_FLASH, include/dspi.h).
    if (first_boot || boot_wrote) pipeline_mute(512);
    if (boot_i2s) pipeline_mute(kPresetMuteSamples);      // (after the directory write in program order: the last writer's count stands)
    dirty = true;
    return sel;
}

// ============================================================================================
// coefficient design
// ============================================================================================
// dsp_compute_coefficients (dsp_pipeline.c:61-175) incl. is_filter_flat (:6-17).  `r` is clamped
// in place exactly as upstream; state resets are recorded in ops.reset_band.
void Params::design_band(Recipe &r, int ch, int b, float fs) {
    BandCoeffs &q = bands[ch][b];
    bool flat = (r.type == FT_FLAT) || (r.freq <= 0.0f);
    if (!flat && (r.type == FT_PEAKING || r.type == FT_LOWSHELF || r.type == FT_HIGHSHELF)) flat = fabsf(r.gain_db) < 0.01f;
    if (flat || fs == 0) {
        q.bypass = true;
        if (flavor) {
            q.b0.f = 1.0f; q.b1.f = q.b2.f = q.a1.f = q.a2.f = 0.0f;
            q.sva1 = q.sva2 = q.sva3 = 0.0f; q.svm0 = q.svm1 = q.svm2 = 0.0f;
            // A band parked on the SVF path keeps its integrators upstream, but use_svf drops to false,
            // so they are cleared before they can be used again; the biquad pair it would fall back to
            // is zero.  With one state pair per band that is "clear now".
            if (q.use_svf) ops.reset_band[ch] |= 1u << b;
            q.use_svf = false;
        } else {
            q.b0.i = 1 << kFilterShift; q.b1.i = q.b2.i = q.a1.i = q.a2.i = 0;
        }
        return;
    }
    q.bypass = false;
    if (r.Q < 0.1f) r.Q = 0.1f;
    if (r.Q > 20.0f) r.Q = 20.0f;
    if (r.freq < 10.0f) r.freq = 10.0f;
    if (r.freq > fs * 0.45f) r.freq = fs * 0.45f;

    float A = powf(10.0f, r.gain_db / 40.0f);

    if (flavor) {
        bool was = q.use_svf;
        q.use_svf = (r.freq < (fs / 7.5f));
        if (was != q.use_svf) ops.reset_band[ch] |= 1u << b;     // :89-92
        if (q.use_svf) {
            float g = tanf(kPi * r.freq / fs);
            float k = 1.0f / r.Q;
            switch (r.type) {
                case FT_PEAKING: k = 1.0f / (r.Q * A); break;
                case FT_LOWSHELF: { float s = sqrtf(A); g = g / s; break; }
                case FT_HIGHSHELF: { float s = sqrtf(A); g = g * s; break; }
                default: break;
            }
            const bool F = fma_contract;
            float a1 = 1.0f / mad(F, g, g + k, 1.0f);
            float a2 = g * a1;
            float a3 = g * a2;
            float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
            switch (r.type) {
                case FT_LOWPASS: m0 = 0.0f; m1 = 0.0f; m2 = 1.0f; break;
                case FT_HIGHPASS: m0 = 1.0f; m1 = -k; m2 = -1.0f; break;
                case FT_PEAKING: m0 = 1.0f; m1 = k * mad(F, A, A, -1.0f); m2 = 0.0f; break;
                case FT_LOWSHELF: m0 = 1.0f; m1 = k * (A - 1.0f); m2 = mad(F, A, A, -1.0f); break;
                case FT_HIGHSHELF: m0 = A * A; m1 = k * (1.0f - A) * A; m2 = 1.0f - m0; break;      // A*A is shared: never fused
                default: break;
            }
            q.sva1 = a1; q.sva2 = a2; q.sva3 = a3; q.svm0 = m0; q.svm1 = m1; q.svm2 = m2;
            q.svf_type = r.type;
            q.b0.f = 1.0f; q.b1.f = q.b2.f = q.a1.f = q.a2.f = 0.0f;
            return;
        }
        q.sva1 = q.sva2 = q.sva3 = 0.0f; q.svm0 = q.svm1 = q.svm2 = 0.0f;
    }

    float omega = 2.0f * kPi * r.freq / fs;
    float sn = sinf(omega), cs = cosf(omega);
    float alpha = sn / (2.0f * r.Q);
    float a0 = 1.0f, a1 = 0.0f, a2 = 0.0f, b0 = 1.0f, b1 = 0.0f, b2 = 0.0f;
    switch (r.type) {
        case FT_LOWPASS: b0 = (1 - cs) / 2; b1 = 1 - cs; b2 = (1 - cs) / 2; a0 = 1 + alpha; a1 = -2 * cs; a2 = 1 - alpha; break;
        case FT_HIGHPASS: b0 = (1 + cs) / 2; b1 = -(1 + cs); b2 = (1 + cs) / 2; a0 = 1 + alpha; a1 = -2 * cs; a2 = 1 - alpha; break;
        case FT_PEAKING:
            b0 = mad(fma_contract, alpha, A, 1.0f); b1 = -2 * cs; b2 = mad(fma_contract, -alpha, A, 1.0f);
            a0 = 1 + alpha / A; a1 = -2 * cs; a2 = 1 - alpha / A; break;
        case FT_LOWSHELF: {      // (A+1) -/+ (A-1)*cs are the sums GCC keeps rounded; 2*sqrt(A)*alpha and (A+1)*cs are the fused products
            const bool F = fma_contract;
            const float Ap = A + 1, Am = A - 1, t = Am * cs, S2 = 2 * sqrtf(A), u = Ap - t, w = Ap + t;
            b0 = A * mad(F, S2, alpha, u); b1 = 2 * A * mad(F, -Ap, cs, Am);
            b2 = A * mad(F, -S2, alpha, u); a0 = mad(F, S2, alpha, w);
            a1 = -2 * mad(F, Ap, cs, Am); a2 = mad(F, -S2, alpha, w); break;
        }
        case FT_HIGHSHELF: {
            const bool F = fma_contract;
            const float Ap = A + 1, Am = A - 1, t = Am * cs, S2 = 2 * sqrtf(A), u = Ap + t, w = Ap - t;
            b0 = A * mad(F, S2, alpha, u); b1 = -2 * A * mad(F, Ap, cs, Am);
            b2 = A * mad(F, -S2, alpha, u); a0 = mad(F, S2, alpha, w);
            a1 = 2 * mad(F, -Ap, cs, Am); a2 = mad(F, -S2, alpha, w); break;
        }
        default: break;
    }
    if (flavor) {
        float inv = 1.0f / a0;
        q.b0.f = b0 * inv; q.b1.f = b1 * inv; q.b2.f = b2 * inv; q.a1.f = a1 * inv; q.a2.f = a2 * inv;
    } else {
        float scale = (float)(1LL << kFilterShift);
        q.b0.i = f2i_sat((b0 / a0) * scale); q.b1.i = f2i_sat((b1 / a0) * scale); q.b2.i = f2i_sat((b2 / a0) * scale);
        q.a1.i = f2i_sat((a1 / a0) * scale); q.a2.i = f2i_sat((a2 / a0) * scale);
    }
}

void Params::recalc_channel_bypass(int ch) {   // main.c:846-854
    bool all = true;
    for (int b = 0; b < kBands; b++) if (!bands[ch][b].bypass) { all = false; break; }
    channel_bypassed[ch] = all;
}

void Params::update_delay_samples(float fs) {   // dsp_pipeline.c:216-239
    any_delay_active = false;
    for (int o = 0; o < n_out; o++) {
        float ms = channel_delays_ms[2 + o];
        if (o == n_out - 1) {
            ms = mad(fma_contract, (float)128 / fs, 1000.0f, ms);    // ms += SUB_ALIGN_SAMPLES / fs * 1000, config.h:93-95
        }
        int32_t s = f2i_sat(ms * fs / 1000.0f);
        if (s > max_delay) s = max_delay;
        if (s < 0) s = 0;
        delay_samples[o] = s;
        if (s > 0) any_delay_active = true;
    }
    dirty = true;
}

void Params::recalc_all_filters(float fs) {   // dsp_pipeline.c:241-253
    update_delay_samples(fs);
    for (int ch = 0; ch < n_ch; ch++) {
        for (int b = 0; b < kBands; b++) design_band(recipes[ch][b], ch, b, fs);
        recalc_channel_bypass(ch);
    }
    dirty = true;
}

void Params::init_default_filters() {   // dsp_pipeline.c:177-214
    memset(bands, 0, sizeof(bands));
    ops.reset_all_eq = 1;                // memset(filters, 0, ...) wipes every state word too
    memset(channel_delays_ms, 0, sizeof(channel_delays_ms));
    for (int ch = 0; ch < n_ch; ch++) {
        channel_bypassed[ch] = true;
        for (int b = 0; b < kStoredBands; b++) {
            BandCoeffs &q = bands[ch][b];
            q.bypass = true;
            if (flavor) q.b0.f = 1.0f; else q.b0.i = 1 << kFilterShift;
            Recipe &r = recipes[ch][b];
            r.type = FT_FLAT; r.freq = 1000.0f; r.Q = 0.707f; r.gain_db = 0.0f;
        }
    }
    for (int ch = 2; ch < n_ch - 1; ch++) {
        Recipe hp{}; hp.type = FT_HIGHPASS; hp.freq = 80.0f; hp.Q = 0.707f;
        recipes[ch][0] = hp;
    }
    Recipe lp{}; lp.type = FT_LOWPASS; lp.freq = 80.0f; lp.Q = 0.707f;
    recipes[n_ch - 1][0] = lp;
}

// ---- loudness: ISO 226:2003 derived shelves (loudness.c:37-217) ----
namespace {
float iso226_spl(bool F, float Tf, float af, float Lu, float phon) {
    float B = 0.4f * powf(10.0f, (Tf + Lu) / 10.0f - 9.0f);
    float thr = powf(B, af);
    float Af = mad(F, 4.47e-3f, powf(10.0f, 0.025f * phon) - 1.15f, thr);
    if (Af < 1e-10f) Af = 1e-10f;
    return mad(F, 10.0f / af, log10f(Af), -Lu) + 94.0f;
}
float loud_comp_db(bool F, float Tf, float af, float Lu, float ref, float eff, float pct) {
    if (eff >= ref) return 0.0f;
    float sr = iso226_spl(F, Tf, af, Lu, ref);
    float se = iso226_spl(F, Tf, af, Lu, eff);
    float flat = eff - ref;
    float fc = se - sr;
    float comp = fc - flat;
    comp *= (pct / 100.0f);
    return comp;
}
void loud_shelf(int flavor, bool F, float freq, float Q, float gain_db, bool high, float fs, LoudCoeffs &o) {
    if (fabsf(gain_db) < 0.01f) {
        o.bypass = true;
        for (auto &w : o.c) w.u = 0;
        if (!flavor) o.c[0].i = 1 << kFilterShift;
        return;
    }
    o.bypass = false;
    float A = powf(10.0f, gain_db / 40.0f);
    if (flavor) {
        float g = tanf(kPi * freq / fs);
        float rA = sqrtf(A);
        if (high) g = g * rA; else g = g / rA;
        float k = 1.0f / Q;
        float a1 = 1.0f / mad(F, g, g + k, 1.0f);
        float a2 = g * a1;
        float a3 = g * a2;
        o.c[0].f = a1; o.c[1].f = a2; o.c[2].f = a3;
        if (high) { o.c[3].f = A * A; o.c[4].f = k * (1.0f - A) * A; o.c[5].f = 1.0f - o.c[3].f; }
        else { o.c[3].f = 1.0f; o.c[4].f = k * (A - 1.0f); o.c[5].f = mad(F, A, A, -1.0f); }
    } else {
        float omega = 2.0f * kPi * freq / fs;
        float sn = sinf(omega), cs = cosf(omega);
        float alpha = sn / (2.0f * Q);
        float rA = sqrtf(A);
        float a0, a1, a2, b0, b1, b2;
        if (high) {
            b0 = A * ((A + 1) + (A - 1) * cs + 2 * rA * alpha); b1 = -2 * A * ((A - 1) + (A + 1) * cs);
            b2 = A * ((A + 1) + (A - 1) * cs - 2 * rA * alpha); a0 = (A + 1) - (A - 1) * cs + 2 * rA * alpha;
            a1 = 2 * ((A - 1) - (A + 1) * cs); a2 = (A + 1) - (A - 1) * cs - 2 * rA * alpha;
        } else {
            b0 = A * ((A + 1) - (A - 1) * cs + 2 * rA * alpha); b1 = 2 * A * ((A - 1) - (A + 1) * cs);
            b2 = A * ((A + 1) - (A - 1) * cs - 2 * rA * alpha); a0 = (A + 1) + (A - 1) * cs + 2 * rA * alpha;
            a1 = -2 * ((A - 1) + (A + 1) * cs); a2 = (A + 1) + (A - 1) * cs - 2 * rA * alpha;
        }
        float scale = (float)(1LL << kFilterShift);
        o.c[0].i = f2i_sat((b0 / a0) * scale); o.c[1].i = f2i_sat((b1 / a0) * scale); o.c[2].i = f2i_sat((b2 / a0) * scale);
        o.c[3].i = f2i_sat((a1 / a0) * scale); o.c[4].i = f2i_sat((a2 / a0) * scale); o.c[5].u = 0;
    }
}
}  // namespace

void Params::loudness_recompute(float fs) {   // loudness_recompute_table, loudness.c:169-217
    if (fs < 1.0f) fs = 48000.0f;
    float ref = loudness_ref_spl;
    if (ref < 40.0f) ref = 40.0f;
    if (ref > 100.0f) ref = 100.0f;
    for (int v = 0; v < 61; v++) {
        float vol_db = (float)(v - 60);
        float eff = ref + vol_db;
        if (eff < 20.0f) eff = 20.0f;
        if (eff > ref) eff = ref;
        float lo = loud_comp_db(fma_contract, 44.0f, 0.432f, 80.4f, ref, eff, loudness_intensity_pct);
        float hi = loud_comp_db(fma_contract, 13.9f, 0.301f, 17.8f, ref, eff, loudness_intensity_pct);
        loud_shelf(flavor, fma_contract, 200.0f, 0.707f, lo, false, fs, loud_table[v][0]);
        loud_shelf(flavor, fma_contract, 6000.0f, 0.707f, hi, true, fs, loud_table[v][1]);
    }
    loud_table_valid = true;
    dirty = true;
}

void Params::crossfeed_design(float fs) {   // crossfeed_compute_coefficients, crossfeed.c:35-127
    static const float presets[3][2] = {{700.0f, 4.5f}, {700.0f, 6.0f}, {650.0f, 9.5f}};
    ops.reset_crossfeed = 1;                 // every path through the function clears the filter states
    dirty = true;
    if (!xfeed_cfg.enabled || fs < 1.0f) { xf_lp_a0.u = xf_lp_b1.u = xf_ap_a.u = 0; return; }
    float fc, feed;
    if (xfeed_cfg.preset < 3) { fc = presets[xfeed_cfg.preset][0]; feed = presets[xfeed_cfg.preset][1]; }
    else {
        fc = xfeed_cfg.custom_fc; feed = xfeed_cfg.custom_feed_db;
        if (fc < 500.0f) fc = 500.0f;
        if (fc > 2000.0f) fc = 2000.0f;
        if (feed < 0.0f) feed = 0.0f;
        if (feed > 15.0f) feed = 15.0f;
    }
    float ratio = powf(10.0f, feed / 20.0f);
    float G = 1.0f / (1.0f + ratio);
    float x = expf(-2.0f * kPi * fc / fs);
    float a0 = G * (1.0f - x), b1 = x, ap;
    if (xfeed_cfg.itd_enabled) {
        float lp_delay = x / ((1.0f - x) * fs);
        float rem = 0.000220f - lp_delay;
        if (rem > 0.0f) ap = mad(fma_contract, -rem, fs, 1.0f) / mad(fma_contract, rem, fs, 1.0f);     // D = rem*fs: (1 - D) / (1 + D)
        else ap = 1.0f;
    } else ap = 1.0f;
    if (flavor) { xf_lp_a0.f = a0; xf_lp_b1.f = b1; xf_ap_a.f = ap; }
    else {
        float scale = (float)(1LL << 28);
        xf_lp_a0.i = f2i_sat(a0 * scale); xf_lp_b1.i = f2i_sat(b1 * scale); xf_ap_a.i = f2i_sat(ap * scale);
    }
}

void Params::leveller_design(float fs) {   // leveller_compute_coefficients, leveller.c:37-89
    static const float presets[3][3] = {{0.100f, 2.000f, 0.400f}, {0.050f, 1.000f, 0.200f}, {0.020f, 0.500f, 0.100f}};
    if (fs < 1.0f) fs = 48000.0f;
    uint8_t spd = lev_cfg.speed;
    if (spd >= 3) spd = 1;
    auto alpha = [&](float t) { return (t <= 0.0f || fs <= 0.0f) ? 0.0f : expf(-logf(10.0f) / (fs * t)); };
    lv_alpha_rms = alpha(presets[spd][2]);
    lv_alpha_attack = alpha(presets[spd][0]);
    lv_alpha_release = alpha(presets[spd][1]);
    lv_threshold_db = -20.0f; lv_knee_db = 6.0f;
    float gate = lev_cfg.gate_threshold_db;
    if (gate < -96.0f) gate = -96.0f;
    if (gate > 0.0f) gate = 0.0f;
    lv_gate_db = gate;
    float amount = lev_cfg.amount;
    if (amount < 0.0f) amount = 0.0f;
    if (amount > 100.0f) amount = 100.0f;
    float norm = amount / 100.0f;
    lv_ratio = mad(fma_contract, norm, 19.0f, 1.0f);
    float mg = lev_cfg.max_gain_db;
    if (mg < 0.0f) mg = 0.0f;
    if (mg > 35.0f) mg = 35.0f;
    lv_max_gain_db = mg;
    lv_makeup_db = 0.0f;
    dirty = true;
}

// ============================================================================================
// deferred-apply dispatcher, volume, rate
// ============================================================================================
void Params::service() {   // main.c:867-894
    float fs = (float)freq;
    if (loud_pending) {
        loud_pending = false;
        loudness_recompute(fs);
        if (loudness_enabled && loud_table_valid) set_volume(volume);
    }
    if (xfeed_pending) {
        xfeed_pending = false;
        crossfeed_design(fs);
        crossfeed_bypassed = !xfeed_cfg.enabled;
    }
    if (lev_pending) {
        lev_pending = false;
        leveller_design(fs);
        if (lev_reset_pending) { lev_reset_pending = false; ops.reset_leveller = 1; }
        leveller_bypassed = !lev_cfg.enabled;
    }
    dirty = true;
}

void Params::set_volume(int16_t v) {   // audio_set_volume, usb_audio.c:428-440
    volume = v;
    uint8_t idx = volume_index(v);
    vol_mul = (int16_t)kDbToVol[idx];
    if (loudness_enabled && loud_table_valid) loud_row = idx;
    dirty = true;
}

void Params::set_mute(bool m) { mute = m; dirty = true; }   // usb_audio.c:1483-1485

void Params::rate_changed(uint32_t hz) {   // perform_rate_change, main.c:132-171 (DSP part)
    recalc_all_filters((float)hz);
    loud_pending = true; xfeed_pending = true; lev_pending = true;
}

int Params::set_rate(uint32_t hz) {   // usb_audio.c:1491-1498 -> main.c:860-865
    if (hz != 44100 && hz != 48000 && hz != 96000) return -10;
    FtzScope ftz;
    if (freq != hz) { freq = hz; rate_changed(hz); service(); }
    return 0;
}

void Params::update_preamp(int ch, float db) {   // usb_audio.c:244-250
    if (!isfinite(db)) return;
    preamp_db[ch] = db;
    float lin = powf(10.0f, db / 20.0f);
    preamp_mul[ch] = f2i_sat(lin * (float)(1 << 28));
    preamp_linear[ch] = lin;
    dirty = true;
}

void Params::set_master_clamped(float db) {   // usb_audio.c:257-268 == flash_storage.c:560-570 == bulk_params.c:363-374
    if (db < kMasterMuteDb) db = kMasterMuteDb;
    if (db > kMasterMaxDb) db = kMasterMaxDb;
    master_db = db;
    if (db <= kMasterMuteDb) { master_linear = 0.0f; master_q15 = 0; }
    else {
        float lin = powf(10.0f, db / 20.0f);
        master_linear = lin;
        master_q15 = f2i_sat(lin * 32768.0f);
    }
    dirty = true;
}

void Params::apply_master_from_mode(bool have_slot, uint16_t slot_version, float slot_db) {   // flash_storage.c:580-589
    float db = (dir_master_volume_mode == 1 && have_slot && slot_version >= 12) ? slot_db : dir_master_volume_db;
    if (!isfinite(db)) db = kMasterMaxDb;
    set_master_clamped(db);
}

void Params::pipeline_mute(uint32_t samples) {   // prepare_pipeline_reset, main.c:449-458
    ops.mute_start = 1;
    ops.mute_samples = samples;
    ops.mute_cancel = 0;            // preset_loading = true again: the last writer wins, as the firmware's flag does
}

void Params::transition_core1() {   // derive_core1_mode, usb_audio.c:1620-1630
    int last_c1 = flavor ? 7 : 3;
    if (outs[n_out - 1].enabled) core1_mode = 1;
    else {
        core1_mode = 0;
        for (int o = 2; o <= last_c1; o++) if (outs[o].enabled) { core1_mode = 2; break; }
    }
    dirty = true;
}

// ============================================================================================
// factory defaults / bulk / presets
// ============================================================================================
void Params::apply_factory_defaults() {   // flash_storage.c:1144-1238
    init_default_filters();
    for (int i = 0; i < 2; i++) { preamp_db[i] = 0.0f; preamp_mul[i] = 1 << 28; preamp_linear[i] = 1.0f; }
    apply_master_from_mode(false, 0, 0.0f);
    bypass_master_eq = false;
    for (int i = 0; i < 3; i++) { legacy_gain_db[i] = 0.0f; legacy_gain_mul[i] = 32768; legacy_mute[i] = false; }
    loudness_enabled = false; loudness_ref_spl = 83.0f; loudness_intensity_pct = 100.0f; loud_pending = true;
    xfeed_cfg.enabled = false; xfeed_cfg.itd_enabled = true; xfeed_cfg.preset = 0; xfeed_cfg.custom_fc = 700.0f; xfeed_cfg.custom_feed_db = 4.5f;
    xfeed_pending = true;
    memset(xp, 0, sizeof(xp)); memset(outs, 0, sizeof(outs));
    xp[0][0].enabled = 1; xp[0][0].gain_linear = 1.0f;
    xp[1][1].enabled = 1; xp[1][1].gain_linear = 1.0f;
    for (int o = 0; o < n_out; o++) { outs[o].enabled = (o < 2) ? 1 : 0; outs[o].gain_linear = 1.0f; }
    default_pins(flavor, pins);
    for (int ch = 0; ch < n_ch; ch++) default_name(flavor, ch, names[ch]);
    memset(output_types, 0, sizeof(output_types));
    i2s_bck_pin = 14; i2s_mck_pin = 13; i2s_mck_enabled = false; i2s_mck_multiplier = 128;
    lev_cfg.enabled = false; lev_cfg.amount = 50.0f; lev_cfg.speed = 0; lev_cfg.max_gain_db = 15.0f; lev_cfg.lookahead = true; lev_cfg.gate_threshold_db = -96.0f;
    lev_pending = true; lev_reset_pending = true;
    dirty = true;
}

void Params::factory_reset() {   // REQ_FACTORY_RESET -> preset_load() of an empty slot, flash_storage.c:811-833
    FtzScope ftz;
    pipeline_mute(kPresetMuteSamples);
    apply_factory_defaults();
    float fs = (float)freq;
    recalc_all_filters(fs); update_delay_samples(fs);
    ops.zero_delay_lines = 1;
    transition_core1();
    service();
}

int Params::load_bulk(const void *blob, size_t len) {   // main.c:1126-1162 + bulk_params_apply, bulk_params.c:178-377
    if (len != sizeof(WireBulk)) return -4;               // usb_audio.c:2250-2251
    FtzScope ftz;
    pipeline_mute(kPresetMuteSamples);
    WireBulk in;
    memcpy(&in, blob, sizeof(in));
    auto apply = [&]() -> int {
        if (in.header.format_version < 2 || in.header.format_version > 6) return -1;
        if (in.header.platform_id != (flavor ? 1 : 0)) return -2;
        if (in.header.num_channels != n_ch || in.header.num_output_channels != n_out) return -3;
        const uint16_t v5 = sizeof(WireBulk) - sizeof(WPreamp) - sizeof(WMaster);
        const uint16_t v2 = v5 - sizeof(WI2S) - sizeof(WLeveller);
        if (in.header.payload_length < v2 || in.header.payload_length > sizeof(WireBulk)) return -4;

        auto set_preamp_all = [&](int i, float db) {
            float lin = db_to_linear_bulk(db, fma_contract);
            preamp_db[i] = db; preamp_mul[i] = f2i_sat(lin * (float)(1 << 28)); preamp_linear[i] = lin;
        };
        for (int i = 0; i < 2; i++) set_preamp_all(i, in.global.preamp_gain_db);
        bypass_master_eq = in.global.bypass != 0;
        loudness_enabled = in.global.loudness_enabled != 0;
        loudness_ref_spl = in.global.loudness_ref_spl; loudness_intensity_pct = in.global.loudness_intensity_pct;
        loud_pending = true;
        xfeed_cfg.enabled = in.crossfeed.enabled != 0; xfeed_cfg.preset = in.crossfeed.preset;
        xfeed_cfg.itd_enabled = in.crossfeed.itd_enabled != 0;
        xfeed_cfg.custom_fc = in.crossfeed.custom_fc; xfeed_cfg.custom_feed_db = in.crossfeed.custom_feed_db;
        xfeed_pending = true;
        for (int i = 0; i < 3; i++) {
            legacy_gain_db[i] = in.legacy.gain_db[i];
            float g = db_to_linear_bulk(in.legacy.gain_db[i], fma_contract);
            legacy_gain_mul[i] = f2i_sat(g * 32768.0f); legacy_gain_linear[i] = g;
            legacy_mute[i] = in.legacy.mute[i] != 0;
        }
        for (int i = 0; i < n_ch; i++) channel_delays_ms[i] = in.delays[i];
        for (int inp = 0; inp < 2; inp++)
            for (int o = 0; o < n_out; o++) {
                Crosspoint &c = xp[inp][o];
                c.enabled = in.crosspoints[inp][o].enabled; c.phase_invert = in.crosspoints[inp][o].phase_invert;
                c.gain_db = in.crosspoints[inp][o].gain_db; c.gain_linear = db_to_linear_bulk(c.gain_db, fma_contract);
            }
        for (int o = 0; o < n_out; o++) {
            OutputCh &oc = outs[o];
            oc.enabled = in.outputs[o].enabled; oc.mute = in.outputs[o].mute;
            oc.gain_db = in.outputs[o].gain_db; oc.gain_linear = db_to_linear_bulk(oc.gain_db, fma_contract);
            oc.delay_ms = in.outputs[o].delay_ms;
            channel_delays_ms[2 + o] = in.outputs[o].delay_ms;      // overrides the delays[] entry (:262)
        }
        if (dir_include_pins) {
            uint8_t def[5]; default_pins(flavor, def);
            for (int i = 0; i < n_pins; i++) pins[i] = gpio_ok(flavor, in.pins.pins[i]) ? in.pins.pins[i] : def[i];
        }
        for (int ch = 0; ch < n_ch; ch++)
            for (int b = 0; b < kStoredBands; b++) {
                Recipe &r = recipes[ch][b];
                r.channel = (uint8_t)ch; r.band = (uint8_t)b; r.type = in.eq[ch][b].type;
                r.freq = in.eq[ch][b].freq; r.Q = in.eq[ch][b].q; r.gain_db = in.eq[ch][b].gain_db;
            }
        for (int ch = 0; ch < n_ch; ch++) { memcpy(names[ch], in.names[ch], 32); names[ch][31] = '\0'; }
        if (in.header.format_version >= 3 && in.header.payload_length >= v5) {
            memcpy(output_types, in.i2s.output_types, (size_t)n_pairs);
            i2s_bck_pin = in.i2s.bck_pin; i2s_mck_pin = in.i2s.mck_pin; i2s_mck_enabled = in.i2s.mck_enabled != 0;
            if (in.header.format_version >= 5) i2s_mck_multiplier = (in.i2s.mck_multiplier == 1) ? 256 : 128;
            else i2s_mck_multiplier = (in.i2s.mck_multiplier == 0) ? 256 : in.i2s.mck_multiplier;
        }
        if (in.header.format_version >= 4) {
            lev_cfg.enabled = in.leveller.enabled != 0; lev_cfg.speed = in.leveller.speed; lev_cfg.lookahead = in.leveller.lookahead != 0;
            lev_cfg.amount = in.leveller.amount; lev_cfg.max_gain_db = in.leveller.max_gain_db; lev_cfg.gate_threshold_db = in.leveller.gate_threshold_db;
        } else {
            lev_cfg.enabled = false; lev_cfg.amount = 50.0f; lev_cfg.speed = 0; lev_cfg.max_gain_db = 15.0f; lev_cfg.lookahead = true; lev_cfg.gate_threshold_db = -96.0f;
        }
        lev_pending = true; lev_reset_pending = true;
        if (in.header.format_version >= 6) {
            for (int i = 0; i < 2; i++) set_preamp_all(i, in.preamp.preamp_db[i]);
            float db = in.master.master_volume_db;
            if (!isfinite(db)) db = kMasterMaxDb;
            set_master_clamped(db);
        }
        return 0;
    };
    int err = apply();
    if (err == 0) {
        float fs = (float)freq;
        recalc_all_filters(fs); update_delay_samples(fs);
        transition_core1();
    }
    service();
    return err;
}

int Params::collect_bulk(void *blob, size_t cap) const {   // bulk_params_collect, bulk_params.c:62-172
    if (cap < sizeof(WireBulk)) return -15;
    WireBulk o;
    memset(&o, 0, sizeof(o));
    o.header.format_version = 6; o.header.platform_id = flavor ? 1 : 0;
    o.header.num_channels = (uint8_t)n_ch; o.header.num_output_channels = (uint8_t)n_out;
    o.header.num_input_channels = 2; o.header.max_bands = kStoredBands;
    o.header.payload_length = sizeof(WireBulk); o.header.fw_major = 1; o.header.fw_minor = 1;
    o.global.preamp_gain_db = preamp_db[0];
    o.global.bypass = bypass_master_eq; o.global.loudness_enabled = loudness_enabled;
    o.global.loudness_ref_spl = loudness_ref_spl; o.global.loudness_intensity_pct = loudness_intensity_pct;
    o.crossfeed.enabled = xfeed_cfg.enabled; o.crossfeed.preset = xfeed_cfg.preset; o.crossfeed.itd_enabled = xfeed_cfg.itd_enabled;
    o.crossfeed.custom_fc = xfeed_cfg.custom_fc; o.crossfeed.custom_feed_db = xfeed_cfg.custom_feed_db;
    for (int i = 0; i < 3; i++) { o.legacy.gain_db[i] = legacy_gain_db[i]; o.legacy.mute[i] = legacy_mute[i]; }
    for (int i = 0; i < n_ch; i++) o.delays[i] = channel_delays_ms[i];
    for (int inp = 0; inp < 2; inp++)
        for (int k = 0; k < n_out; k++) {
            o.crosspoints[inp][k].enabled = xp[inp][k].enabled; o.crosspoints[inp][k].phase_invert = xp[inp][k].phase_invert;
            o.crosspoints[inp][k].gain_db = xp[inp][k].gain_db;
        }
    for (int k = 0; k < n_out; k++) {
        o.outputs[k].enabled = outs[k].enabled; o.outputs[k].mute = outs[k].mute;
        o.outputs[k].gain_db = outs[k].gain_db; o.outputs[k].delay_ms = outs[k].delay_ms;
    }
    o.pins.num_pin_outputs = (uint8_t)n_pins;
    for (int i = 0; i < n_pins; i++) o.pins.pins[i] = pins[i];
    for (int ch = 0; ch < n_ch; ch++)
        for (int b = 0; b < kStoredBands; b++) {
            o.eq[ch][b].type = recipes[ch][b].type; o.eq[ch][b].freq = recipes[ch][b].freq;
            o.eq[ch][b].q = recipes[ch][b].Q; o.eq[ch][b].gain_db = recipes[ch][b].gain_db;
        }
    for (int ch = 0; ch < n_ch; ch++) memcpy(o.names[ch], names[ch], 32);
    memcpy(o.i2s.output_types, output_types, (size_t)n_pairs);
    o.i2s.bck_pin = i2s_bck_pin; o.i2s.mck_pin = i2s_mck_pin; o.i2s.mck_enabled = i2s_mck_enabled; o.i2s.mck_multiplier = (i2s_mck_multiplier == 256) ? 1 : 0;
    o.leveller.enabled = lev_cfg.enabled; o.leveller.speed = lev_cfg.speed; o.leveller.lookahead = lev_cfg.lookahead;
    o.leveller.amount = lev_cfg.amount; o.leveller.max_gain_db = lev_cfg.max_gain_db; o.leveller.gate_threshold_db = lev_cfg.gate_threshold_db;
    for (int i = 0; i < 2; i++) o.preamp.preamp_db[i] = preamp_db[i];
    o.master.master_volume_db = master_db;
    memcpy(blob, &o, sizeof(o));
    return (int)sizeof(o);
}

// preset_load path (main.c:926-976, flash_storage.c:750-759, :794-849) + apply_slot_to_live (:597-742)
// validate_slot + apply_slot_to_live + apply_master_volume_from_mode (flash_storage.c:750-759, :597-742): a slot image into the live
// parameters, nothing else — what preset_load (load_slot below) and the boot path (boot_select) share.  false: bad magic / index / CRC.
bool Params::slot_to_live(const void *image, int expect_slot) {
    SlotCursor c(image);
    uint32_t magic = c.get<uint32_t>();
    uint16_t version = c.get<uint16_t>();
    uint16_t slot_index = c.get<uint16_t>();
    uint32_t crc = c.get<uint32_t>();
    bool ok = magic == kSlotMagic && (expect_slot < 0 || slot_index == (uint16_t)expect_slot) &&
              crc32_edb88320((const uint8_t *)image + 12, (size_t)slot_size() - 12) == crc;
    if (!ok) return false;

    for (int ch = 0; ch < n_ch; ch++) for (int b = 0; b < kStoredBands; b++) recipes[ch][b] = c.get<Recipe>();
    float legacy_preamp = c.get<float>();
    bypass_master_eq = c.get<uint8_t>() != 0; c.skip(3);
    for (int ch = 0; ch < n_ch; ch++) channel_delays_ms[ch] = c.get<float>();
    for (int i = 0; i < 3; i++) {
        legacy_gain_db[i] = c.get<float>();
        float g = db_to_linear_preset(legacy_gain_db[i]);
        legacy_gain_mul[i] = f2i_sat(g * 32768.0f);
    }
    for (int i = 0; i < 3; i++) legacy_mute[i] = c.get<uint8_t>() != 0;
    c.skip(1);
    loudness_enabled = c.get<uint8_t>() != 0; c.skip(3);
    loudness_ref_spl = c.get<float>(); loudness_intensity_pct = c.get<float>(); loud_pending = true;
    xfeed_cfg.enabled = c.get<uint8_t>() != 0; xfeed_cfg.preset = c.get<uint8_t>(); xfeed_cfg.itd_enabled = c.get<uint8_t>() != 0; c.skip(1);
    xfeed_cfg.custom_fc = c.get<float>(); xfeed_cfg.custom_feed_db = c.get<float>(); xfeed_pending = true;
    for (int inp = 0; inp < 2; inp++)
        for (int o = 0; o < n_out; o++) {
            Crosspoint &x = xp[inp][o];
            x.enabled = c.get<uint8_t>(); x.phase_invert = c.get<uint8_t>(); c.skip(2);
            x.gain_db = c.get<float>(); x.gain_linear = db_to_linear_preset(x.gain_db);
        }
    for (int o = 0; o < n_out; o++) {
        OutputCh &oc = outs[o];
        oc.enabled = c.get<uint8_t>(); oc.mute = c.get<uint8_t>(); c.skip(2);
        oc.gain_db = c.get<float>(); oc.gain_linear = db_to_linear_preset(oc.gain_db);
        oc.delay_ms = c.get<float>();
        channel_delays_ms[2 + o] = oc.delay_ms;
    }
    {
        uint8_t sp[8];
        for (int i = 0; i < 8; i++) sp[i] = c.get<uint8_t>();
        if (dir_include_pins) {
            uint8_t def[5]; default_pins(flavor, def);
            for (int i = 0; i < n_pins; i++) pins[i] = gpio_ok(flavor, sp[i]) ? sp[i] : def[i];
        }
    }
    if (version >= 8) for (int ch = 0; ch < n_ch; ch++) { memcpy(names[ch], c.rd + c.off, 32); c.skip(32); }
    else { c.skip((size_t)n_ch * 32); for (int ch = 0; ch < n_ch; ch++) default_name(flavor, ch, names[ch]); }
    {
        uint8_t t[4]; for (auto &v : t) v = c.get<uint8_t>();
        uint8_t bck = c.get<uint8_t>(), mck = c.get<uint8_t>(), en = c.get<uint8_t>(), mult = c.get<uint8_t>();
        if (version >= 9) {
            memcpy(output_types, t, (size_t)n_pairs);
            i2s_bck_pin = bck; i2s_mck_pin = mck; i2s_mck_enabled = en != 0;
            if (version >= 11) i2s_mck_multiplier = (mult == 1) ? 256 : 128;
            else i2s_mck_multiplier = (mult == 0) ? 256 : mult;
        } else {
            memset(output_types, 0, (size_t)n_pairs);
            i2s_bck_pin = 14; i2s_mck_pin = 13; i2s_mck_enabled = false; i2s_mck_multiplier = 128;
        }
    }
    {
        uint8_t en = c.get<uint8_t>(), spd = c.get<uint8_t>(), la = c.get<uint8_t>(); c.skip(1);
        float amount = c.get<float>(), mg = c.get<float>(), gate = c.get<float>();
        if (version >= 10) {
            lev_cfg.enabled = en != 0; lev_cfg.speed = spd; lev_cfg.lookahead = la != 0;
            lev_cfg.amount = amount; lev_cfg.max_gain_db = mg; lev_cfg.gate_threshold_db = gate;
        } else {
            lev_cfg.enabled = false; lev_cfg.amount = 50.0f; lev_cfg.speed = 0; lev_cfg.max_gain_db = 15.0f; lev_cfg.lookahead = true; lev_cfg.gate_threshold_db = -96.0f;
        }
        lev_pending = true; lev_reset_pending = true;
    }
    float per_ch[2] = {c.get<float>(), c.get<float>()};
    float slot_master = c.get<float>();
    for (int i = 0; i < 2; i++) {
        float db = (version >= 12) ? per_ch[i] : legacy_preamp;
        preamp_db[i] = db;
        float lin = db_to_linear_preset(db);
        preamp_mul[i] = f2i_sat(lin * (float)(1 << 28)); preamp_linear[i] = lin;
    }
    apply_master_from_mode(true, version, slot_master);
    return true;
}

int Params::load_slot(const void *image, size_t len, int expect_slot) {     // preset_load (flash_storage.c:794-849) under main.c:926-976
    if (len < (size_t)slot_size()) return 3;    // PRESET_ERR_CRC
    FtzScope ftz;
    pipeline_mute(kPresetMuteSamples);
    uint8_t old_types[4]; memcpy(old_types, output_types, 4);
    if (!slot_to_live(image, expect_slot)) { ops.mute_start = 0; ops.mute_cancel = 1; return 3; }    // preset_loading = false (:806)

    float fs = (float)freq;
    recalc_all_filters(fs); update_delay_samples(fs);
    ops.zero_delay_lines = 1;              // flash_storage.c:832
    transition_core1();
    // preset_load ends by writing the directory (flash_storage.c:846-847) and every flash_write_sector re-arms the mute
    // for flash_mute_hold_samples() = max(10 ms, 512 samples) (:272-276, :347-348)
    {
        uint64_t hold = ((uint64_t)freq * 10u + 999u) / 1000u;
        pipeline_mute(hold < 512u ? 512u : (uint32_t)hold);
    }
    // a preset that changes an output slot's type goes through process_type_switches, which arms the mute once more
    // with PRESET_MUTE_SAMPLES (main.c:957-972, :279)
    if (memcmp(old_types, output_types, (size_t)n_pairs) != 0) pipeline_mute(kPresetMuteSamples);
    service();
    return 0;
}

// ------------------------------------------------------------------------------------------
// flash dump: directory sector, startup-slot selection, legacy migration (flash_storage.c:370-417, :997-1105)
// ------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kDirMagic = 0x44535032u, kLegacyMagic = 0x44535031u;     // "DSP2", "DSP1" (flash_storage.c:64-68)
constexpr size_t kSector = 4096;
template <class V> V rd_at(const uint8_t *p, size_t off) { V v; memcpy(&v, p + off, sizeof v); return v; }
}

bool parse_flash_directory(const void *dump, size_t len, FlashDirectory &d) {
    memset(&d, 0, sizeof d);
    if (!dump || len < kSector) return false;
    const uint8_t *p = static_cast<const uint8_t *>(dump);
    if (rd_at<uint32_t>(p, 0) != kDirMagic) return false;
    const uint16_t version = rd_at<uint16_t>(p, 4);
    const uint32_t crc = rd_at<uint32_t>(p, 8);
    // v2: header 12, startup 4, occupied 2, mode 1, pad 1, master dB 4, names 320 = 344; v1 has no master dB = 340
    const size_t total = version == 2 ? 344 : version == 1 ? 340 : 0;
    if (!total || crc32_edb88320(p + 12, total - 12) != crc) return false;
    d.valid = 1; d.version = version;
    d.startup_mode = p[12]; d.default_slot = p[13]; d.last_active_slot = p[14]; d.include_pins = p[15];
    d.slot_occupied = rd_at<uint16_t>(p, 16);
    if (version == 2) {
        d.master_volume_mode = p[18];
        d.master_volume_db = rd_at<float>(p, 20);
        memcpy(d.slot_names, p + 24, sizeof d.slot_names);
    } else {                      // v1 -> v2 in memory (flash_storage.c:391-411)
        d.master_volume_mode = p[18] ? 1 : 0;
        d.master_volume_db = kMasterDefaultDb;
        memcpy(d.slot_names, p + 20, sizeof d.slot_names);
    }
    return true;
}

// preset_boot_load's SELECTION on a 48 KB flash image (flash_storage.c:1047-1105).  Return: 0..9 slot loaded | 16+slot: that slot was
// selected but is empty or corrupt -> factory defaults | 32: no directory, legacy sector migrated into slot 0 and loaded | 48: nothing
// usable -> factory defaults.  `booting`: the APPLICATION is the boot path's too (apply_slot_to_live / apply_factory_defaults and nothing
// else, :1066-1076); otherwise it is preset_load's (:794-849: mute, delay lines zeroed) — a running device switching to that preset.
int Params::flash_select(const void *dump, bool booting) {
    const uint8_t *p = static_cast<const uint8_t *>(dump);
    auto apply = [&](const void *image, int slot) { return booting ? (slot_to_live(image, slot) ? 0 : 3) : load_slot(image, (size_t)slot_size(), slot); };
    auto defaults = [&]() { if (booting) apply_factory_defaults(); else factory_reset(); };
    FlashDirectory d;
    if (parse_flash_directory(dump, kFlashDumpBytes, d)) {
        uint8_t target = d.startup_mode == 1 ? d.last_active_slot : d.default_slot;      // PRESET_STARTUP_LAST_ACTIVE
        if (target >= 10) { target = d.default_slot; if (target >= 10) target = 0; }
        dir_master_volume_mode = d.master_volume_mode; dir_master_volume_db = d.master_volume_db; dir_include_pins = d.include_pins;
        if ((d.slot_occupied >> target) & 1u)
            if (apply(p + (1 + (size_t)target) * kSector, target) == 0) return target;
        defaults();
        return 16 + target;
    }
    // no directory: migrate_legacy (flash_storage.c:997-1045) — the legacy sector's data section has the slot's layout
    const uint8_t *lg = p + 11 * kSector;
    const size_t legacy_bytes = 12 + (size_t)n_ch * kStoredBands * 16 + 4 + 4 + (size_t)n_ch * 4 + 12 + 4 + 4 + 8 + 4 + 8 +
                                (size_t)2 * n_out * 8 + (size_t)n_out * 12 + 8;
    dir_master_volume_mode = 0; dir_master_volume_db = kMasterDefaultDb;
    if (rd_at<uint32_t>(lg, 0) == kLegacyMagic && crc32_edb88320(lg + 12, legacy_bytes - 12) == rd_at<uint32_t>(lg, 8)) {
        std::vector<uint8_t> slot((size_t)slot_size(), 0);
        const uint32_t magic = kSlotMagic; const uint16_t version = rd_at<uint16_t>(lg, 4), idx = 0;
        memcpy(&slot[0], &magic, 4); memcpy(&slot[4], &version, 2); memcpy(&slot[6], &idx, 2);
        memcpy(&slot[12], lg + 12, legacy_bytes - 12);
        const uint32_t crc = crc32_edb88320(&slot[12], (size_t)slot_size() - 12);
        memcpy(&slot[8], &crc, 4);
        dir_include_pins = 0;                       // "Legacy migration: don't override pins" (:1091)
        const int rc = apply(slot.data(), 0);
        dir_include_pins = 1;
        if (rc == 0) return 32;
    }
    dir_include_pins = 1;                           // dir_ensure (:440-457)
    defaults();
    return 48;
}

// `as_boot`: the stream is a device with a populated flash that has not played anything yet (dspi_capi.cpp decides) — the dump is the
// flash it BOOTS from: the power-on sequence runs again over it (no mute, no line zeroing: preset_boot_load, flash_storage.c:1047-1082),
// and what the host has set since power-on in the firmware's own way (sample rate, UAC1 volume and mute) is set again afterwards.
int Params::load_flash_dump(const void *dump, size_t len, bool as_boot) {
    if (!dump || len < kFlashDumpBytes) return -4;
    if (!as_boot) { FtzScope ftz; return flash_select(dump, false); }
    const uint32_t hz = freq; const int16_t v = volume; const bool m = mute;
    const int fl = flavor; const bool fma = fma_contract;
    memset((void *)this, 0, sizeof(*this));
    first_boot = false; flavor = fl; fma_contract = fma;
    StateMap sm = make_state_map(fl);
    n_ch = sm.n_ch; n_out = sm.n_out; n_pairs = sm.n_pairs; max_delay = sm.max_delay;
    n_pins = fl ? 5 : 3;
    const int sel = boot(dump);
    if (hz != freq) set_rate(hz);
    set_volume(v); set_mute(m);
    return sel;
}

int Params::save_slot(void *image, size_t cap, int slot_index) const {   // collect_live_state, flash_storage.c:464-552
    if (cap < (size_t)slot_size()) return -15;
    memset(image, 0, (size_t)slot_size());
    SlotCursor c(image, 0);
    c.put<uint32_t>(kSlotMagic); c.put<uint16_t>(kSlotVersion); c.put<uint16_t>((uint16_t)(uint8_t)slot_index); c.put<uint32_t>(0);
    for (int ch = 0; ch < n_ch; ch++) for (int b = 0; b < kStoredBands; b++) c.put<Recipe>(recipes[ch][b]);
    c.put<float>(preamp_db[0]); c.put<uint8_t>(bypass_master_eq ? 1 : 0); c.skip(3);
    for (int ch = 0; ch < n_ch; ch++) c.put<float>(channel_delays_ms[ch]);
    for (int i = 0; i < 3; i++) c.put<float>(legacy_gain_db[i]);
    for (int i = 0; i < 3; i++) c.put<uint8_t>(legacy_mute[i] ? 1 : 0);
    c.skip(1);
    c.put<uint8_t>(loudness_enabled ? 1 : 0); c.skip(3); c.put<float>(loudness_ref_spl); c.put<float>(loudness_intensity_pct);
    c.put<uint8_t>(xfeed_cfg.enabled ? 1 : 0); c.put<uint8_t>(xfeed_cfg.preset); c.put<uint8_t>(xfeed_cfg.itd_enabled ? 1 : 0); c.skip(1);
    c.put<float>(xfeed_cfg.custom_fc); c.put<float>(xfeed_cfg.custom_feed_db);
    for (int inp = 0; inp < 2; inp++)
        for (int o = 0; o < n_out; o++) { c.put<uint8_t>(xp[inp][o].enabled); c.put<uint8_t>(xp[inp][o].phase_invert); c.skip(2); c.put<float>(xp[inp][o].gain_db); }
    for (int o = 0; o < n_out; o++) { c.put<uint8_t>(outs[o].enabled); c.put<uint8_t>(outs[o].mute); c.skip(2); c.put<float>(outs[o].gain_db); c.put<float>(outs[o].delay_ms); }
    for (int i = 0; i < 8; i++) c.put<uint8_t>(i < n_pins ? pins[i] : 0);
    for (int ch = 0; ch < n_ch; ch++) { memcpy(c.wr + c.off, names[ch], 32); c.skip(32); }
    for (int i = 0; i < 4; i++) c.put<uint8_t>(i < n_pairs ? output_types[i] : 0);
    c.put<uint8_t>(i2s_bck_pin); c.put<uint8_t>(i2s_mck_pin); c.put<uint8_t>(i2s_mck_enabled ? 1 : 0); c.put<uint8_t>(i2s_mck_multiplier == 256 ? 1 : 0);
    c.put<uint8_t>(lev_cfg.enabled ? 1 : 0); c.put<uint8_t>(lev_cfg.speed); c.put<uint8_t>(lev_cfg.lookahead ? 1 : 0); c.skip(1);
    c.put<float>(lev_cfg.amount); c.put<float>(lev_cfg.max_gain_db); c.put<float>(lev_cfg.gate_threshold_db);
    c.put<float>(preamp_db[0]); c.put<float>(preamp_db[1]); c.put<float>(master_db);
    uint32_t crc = crc32_edb88320((const uint8_t *)image + 12, (size_t)slot_size() - 12);
    memcpy((uint8_t *)image + 8, &crc, 4);
    return slot_size();
}

// ============================================================================================
// vendor requests
// ============================================================================================
int Params::vendor_set(uint8_t req, uint16_t wValue, const void *payload, uint16_t len) {   // vendor_cmd_packet, usb_audio.c:1632-2021
    FtzScope ftz;
    const uint8_t *b = (const uint8_t *)payload;
    const uint8_t idx = (uint8_t)(wValue & 0xFF);
    const float fs = (float)freq;
    auto clampf = [](float v, float lo, float hi) { if (v < lo) v = lo; if (v > hi) v = hi; return v; };
    bool known = true;
    switch (req) {
        case REQ_SET_EQ_PARAM:
            if (len >= sizeof(Recipe)) {
                Recipe p; memcpy(&p, b, sizeof p);
                if (p.channel < n_ch && p.band < kBands) {       // main.c:826-857
                    recipes[p.channel][p.band] = p;
                    design_band(p, p.channel, p.band, fs);       // upstream designs from a COPY: the stored recipe is not clamped
                    recalc_channel_bypass(p.channel);
                }
            }
            break;
        case REQ_SET_PREAMP: if (len >= 4) for (int ch = 0; ch < 2; ch++) update_preamp(ch, rd_f32(b)); break;
        case REQ_SET_PREAMP_CH: if (idx < 2 && len >= 4) update_preamp(idx, rd_f32(b)); break;
        case REQ_SET_MASTER_VOLUME: if (len >= 4) { float db = rd_f32(b); if (isfinite(db)) set_master_clamped(db); } break;
        case REQ_SET_DELAY:
            if (idx < n_ch && len >= 4) { float ms = rd_f32(b); if (ms < 0) ms = 0; channel_delays_ms[idx] = ms; update_delay_samples(fs); }
            break;
        case REQ_SET_BYPASS: if (len >= 1) bypass_master_eq = b[0] != 0; break;
        case REQ_SET_CHANNEL_GAIN:
            if (idx < 3 && len >= 4) {
                float db = rd_f32(b); legacy_gain_db[idx] = db;
                float lin = powf(10.0f, db / 20.0f);
                legacy_gain_mul[idx] = f2i_sat(lin * 32768.0f); legacy_gain_linear[idx] = lin;
            }
            break;
        case REQ_SET_CHANNEL_MUTE: if (idx < 3 && len >= 1) legacy_mute[idx] = b[0] != 0; break;
        case REQ_SET_LOUDNESS:
            if (len >= 1) {
                loudness_enabled = b[0] != 0;
                if (loudness_enabled && loud_table_valid) loud_row = volume_index(volume); else loud_row = -1;
            }
            break;
        case REQ_SET_LOUDNESS_REF: if (len >= 4) { loudness_ref_spl = clampf(rd_f32(b), 40.0f, 100.0f); loud_pending = true; } break;
        case REQ_SET_LOUDNESS_INTENSITY: if (len >= 4) { loudness_intensity_pct = clampf(rd_f32(b), 0.0f, 200.0f); loud_pending = true; } break;
        case REQ_SET_CROSSFEED: if (len >= 1) { xfeed_cfg.enabled = b[0] != 0; xfeed_pending = true; } break;
        case REQ_SET_CROSSFEED_PRESET: if (len >= 1 && b[0] <= 3) { xfeed_cfg.preset = b[0]; xfeed_pending = true; } break;
        case REQ_SET_CROSSFEED_FREQ:
            if (len >= 4) { xfeed_cfg.custom_fc = clampf(rd_f32(b), 500.0f, 2000.0f); if (xfeed_cfg.preset == 3) xfeed_pending = true; }
            break;
        case REQ_SET_CROSSFEED_FEED:
            if (len >= 4) { xfeed_cfg.custom_feed_db = clampf(rd_f32(b), 0.0f, 15.0f); if (xfeed_cfg.preset == 3) xfeed_pending = true; }
            break;
        case REQ_SET_CROSSFEED_ITD: if (len >= 1) { xfeed_cfg.itd_enabled = b[0] != 0; xfeed_pending = true; } break;
        case REQ_SET_LEVELLER_ENABLE: if (len >= 1) { lev_cfg.enabled = b[0] != 0; lev_pending = true; lev_reset_pending = true; } break;
        case REQ_SET_LEVELLER_AMOUNT: if (len >= 4) { lev_cfg.amount = clampf(rd_f32(b), 0.0f, 100.0f); lev_pending = true; } break;
        case REQ_SET_LEVELLER_SPEED: if (len >= 1 && b[0] < 3) { lev_cfg.speed = b[0]; lev_pending = true; } break;
        case REQ_SET_LEVELLER_MAX_GAIN: if (len >= 4) { lev_cfg.max_gain_db = clampf(rd_f32(b), 0.0f, 35.0f); lev_pending = true; } break;
        case REQ_SET_LEVELLER_LOOKAHEAD: if (len >= 1) { lev_cfg.lookahead = b[0] != 0; lev_pending = true; lev_reset_pending = true; } break;
        case REQ_SET_LEVELLER_GATE: if (len >= 4) { lev_cfg.gate_threshold_db = clampf(rd_f32(b), -96.0f, 0.0f); lev_pending = true; } break;
        case REQ_SET_MATRIX_ROUTE:
            if (len >= 8) {
                uint8_t in = b[0], out = b[1];
                if (in < 2 && out < n_out) {
                    Crosspoint &x = xp[in][out];
                    x.enabled = b[2]; x.phase_invert = b[3]; x.gain_db = rd_f32(b + 4);
                    x.gain_linear = powf(10.0f, x.gain_db / 20.0f);
                }
            }
            break;
        case REQ_SET_OUTPUT_ENABLE:
            if (idx < n_out && len >= 1) {      // PDM and the Core-1 EQ outputs exclude each other (:1891-1904)
                bool want = b[0] != 0, skip = false;
                const int last_c1 = flavor ? 7 : 3;
                if (want) {
                    if (idx == n_out - 1) { for (int i = 2; i <= last_c1; i++) if (outs[i].enabled) skip = true; }
                    else if (idx >= 2 && idx <= last_c1) { if (outs[n_out - 1].enabled) skip = true; }
                }
                if (!skip) { outs[idx].enabled = want ? 1 : 0; transition_core1(); }
            }
            break;
        case REQ_SET_OUTPUT_GAIN:
            if (idx < n_out && len >= 4) { outs[idx].gain_db = rd_f32(b); outs[idx].gain_linear = powf(10.0f, outs[idx].gain_db / 20.0f); }
            break;
        case REQ_SET_OUTPUT_MUTE: if (idx < n_out && len >= 1) outs[idx].mute = b[0]; break;
        case REQ_SET_OUTPUT_DELAY:
            if (idx < n_out && len >= 4) {
                float ms = rd_f32(b); if (ms < 0) ms = 0;
                outs[idx].delay_ms = ms; channel_delays_ms[2 + idx] = ms; update_delay_samples(fs);
            }
            break;
        case REQ_SET_MASTER_VOLUME_MODE: if (len >= 1) dir_master_volume_mode = b[0] > 1 ? 0 : b[0]; break;
        case REQ_SET_CHANNEL_NAME:
            if (idx < n_ch && len > 0) { memset(names[idx], 0, 32); memcpy(names[idx], b, len < 31 ? len : 31); }
            break;
        case REQ_SET_ALL_PARAMS: { int rc = load_bulk(payload, len); return rc; }
        default: known = false; break;
    }
    service();
    return known ? 0 : -14;
}

int Params::vendor_get(uint8_t req, uint16_t wValue, void *buf, uint16_t cap, const uint16_t *peaks, uint16_t *clip_flags) {   // usb_audio.c:2271-2688
    const uint8_t idx = (uint8_t)wValue;
    auto put = [&](const void *src, int n) -> int { if (cap < n) return -15; memcpy(buf, src, (size_t)n); return n; };
    auto put_u8 = [&](uint8_t v) { return put(&v, 1); };
    auto put_u32 = [&](uint32_t v) { return put(&v, 4); };
    switch (req) {
        case REQ_GET_PREAMP: return put(&preamp_db[0], 4);
        case REQ_GET_PREAMP_CH: return idx < 2 ? put(&preamp_db[idx], 4) : -14;
        case REQ_GET_MASTER_VOLUME: return put(&master_db, 4);
        case REQ_GET_MASTER_VOLUME_MODE: return put_u8(dir_master_volume_mode);
        case REQ_GET_SAVED_MASTER_VOLUME: return put(&dir_master_volume_db, 4);
        case REQ_SAVE_MASTER_VOLUME: dir_master_volume_db = master_db; return put_u8(0);
        case REQ_GET_DELAY: return idx < n_ch ? put(&channel_delays_ms[idx], 4) : -14;
        case REQ_GET_BYPASS: return put_u8(bypass_master_eq ? 1 : 0);
        case REQ_GET_CHANNEL_GAIN: return idx < 3 ? put(&legacy_gain_db[idx], 4) : -14;
        case REQ_GET_CHANNEL_MUTE: return idx < 3 ? put_u8(legacy_mute[idx] ? 1 : 0) : -14;
        case REQ_GET_LOUDNESS: return put_u8(loudness_enabled ? 1 : 0);
        case REQ_GET_LOUDNESS_REF: return put(&loudness_ref_spl, 4);
        case REQ_GET_LOUDNESS_INTENSITY: return put(&loudness_intensity_pct, 4);
        case REQ_GET_CROSSFEED: return put_u8(xfeed_cfg.enabled ? 1 : 0);
        case REQ_GET_CROSSFEED_PRESET: return put_u8(xfeed_cfg.preset);
        case REQ_GET_CROSSFEED_FREQ: return put(&xfeed_cfg.custom_fc, 4);
        case REQ_GET_CROSSFEED_FEED: return put(&xfeed_cfg.custom_feed_db, 4);
        case REQ_GET_CROSSFEED_ITD: return put_u8(xfeed_cfg.itd_enabled ? 1 : 0);
        case REQ_GET_LEVELLER_ENABLE: return put_u8(lev_cfg.enabled ? 1 : 0);
        case REQ_GET_LEVELLER_AMOUNT: return put(&lev_cfg.amount, 4);
        case REQ_GET_LEVELLER_SPEED: return put_u8(lev_cfg.speed);
        case REQ_GET_LEVELLER_MAX_GAIN: return put(&lev_cfg.max_gain_db, 4);
        case REQ_GET_LEVELLER_LOOKAHEAD: return put_u8(lev_cfg.lookahead ? 1 : 0);
        case REQ_GET_LEVELLER_GATE: return put(&lev_cfg.gate_threshold_db, 4);
        case REQ_GET_STATUS: {
            if (wValue == 9) {
                uint8_t r[kMaxCh * 2 + 4];
                for (int i = 0; i < n_ch; i++) { r[i * 2] = peaks[i] & 0xFF; r[i * 2 + 1] = peaks[i] >> 8; }
                r[n_ch * 2] = 0; r[n_ch * 2 + 1] = 0;     // cpu0_load / cpu1_load: no MCU here
                r[n_ch * 2 + 2] = *clip_flags & 0xFF; r[n_ch * 2 + 3] = *clip_flags >> 8;
                return put(r, n_ch * 2 + 4);
            }
            uint32_t resp = 0;
            if (wValue == 0) resp = (uint32_t)peaks[0] | ((uint32_t)peaks[1] << 16);
            else if (wValue == 1) resp = (uint32_t)peaks[2] | ((uint32_t)peaks[3] << 16);
            else if (wValue == 2) resp = (uint32_t)peaks[4];
            else if (wValue == 15) resp = freq;
            return put_u32(resp);
        }
        case REQ_GET_EQ_PARAM: {
            uint8_t ch = (wValue >> 8) & 0xFF, band = (wValue >> 4) & 0x0F, param = wValue & 0x0F;
            if (ch >= n_ch || band >= kBands) return -14;
            const Recipe &p = recipes[ch][band];
            uint32_t v = 0;
            if (param == 0) v = p.type; else if (param == 1) memcpy(&v, &p.freq, 4);
            else if (param == 2) memcpy(&v, &p.Q, 4); else if (param == 3) memcpy(&v, &p.gain_db, 4);
            return put_u32(v);
        }
        case REQ_GET_MATRIX_ROUTE: {
            uint8_t in = (wValue >> 8) & 0xFF, out = wValue & 0xFF;
            if (in >= 2 || out >= n_out) return -14;
            uint8_t pk[8] = {in, out, xp[in][out].enabled, xp[in][out].phase_invert};
            memcpy(pk + 4, &xp[in][out].gain_db, 4);
            return put(pk, 8);
        }
        case REQ_GET_OUTPUT_ENABLE: return idx < n_out ? put_u8(outs[idx].enabled) : -14;
        case REQ_GET_OUTPUT_GAIN: return idx < n_out ? put(&outs[idx].gain_db, 4) : -14;
        case REQ_GET_OUTPUT_MUTE: return idx < n_out ? put_u8(outs[idx].mute) : -14;
        case REQ_GET_OUTPUT_DELAY: return idx < n_out ? put(&outs[idx].delay_ms, 4) : -14;
        case REQ_GET_CORE1_MODE: return put_u8((uint8_t)core1_mode);
        case REQ_GET_CORE1_CONFLICT: {
            uint8_t conflict = 0;
            const int last_c1 = flavor ? 7 : 3;
            if (idx < n_out) {
                if (idx == n_out - 1) { for (int i = 2; i <= last_c1; i++) if (outs[i].enabled) { conflict = 1; break; } }
                else if (idx >= 2 && idx <= last_c1) { if (outs[n_out - 1].enabled) conflict = 1; }
            }
            return put_u8(conflict);
        }
        case REQ_GET_PLATFORM: { uint8_t r[4] = {(uint8_t)(flavor ? 1 : 0), 0x01, 0x13, (uint8_t)n_out}; return put(r, 4); }
        case REQ_CLEAR_CLIPS: { uint16_t f = *clip_flags; *clip_flags = 0; ops.clear_clips = 1; return put(&f, 2); }
        case REQ_GET_CHANNEL_NAME: return idx < n_ch ? put(names[idx], 32) : -14;
        case REQ_GET_ALL_PARAMS: return collect_bulk(buf, cap);
        case REQ_FACTORY_RESET: factory_reset(); return put_u8(0);
        case REQ_SET_OUTPUT_TYPE: {     // usb_audio.c:2984-3016; the deferred switch (main.c:230-424) is taken at once: its DSP side
            const uint8_t slot = wValue & 0xFF, type = (wValue >> 8) & 0xFF;     // effect is the pipeline mute (main.c:279)
            if (slot >= n_pairs) return put_u8(0x03);      // PIN_CONFIG_INVALID_OUTPUT
            if (type > 1) return put_u8(0x01);             // PIN_CONFIG_INVALID_PIN
            if (type != output_types[slot]) { output_types[slot] = type; pipeline_mute(kPresetMuteSamples); }
            return put_u8(0x00);
        }
        case REQ_GET_OUTPUT_TYPE: return idx < n_pairs ? put_u8(output_types[idx]) : -14;
        default: return -14;
    }
}

// ============================================================================================
// flatten into the device image
// ============================================================================================
void Params::build_image(DevImage &img) const {
    memset(&img, 0, sizeof(img));
    for (int ch = 0; ch < n_ch; ch++)
        for (int b = 0; b < kBands; b++) {
            const BandCoeffs &q = bands[ch][b];
            DevBand &d = img.eq[ch][b];
            if (q.bypass) { d.kind = K_BYPASS; continue; }
            if (flavor && q.use_svf) {
                d.c[0].f = q.sva1; d.c[1].f = q.sva2; d.c[2].f = q.sva3;
                switch (q.svf_type) {      // per-type inner loops, dsp_pipeline.c:298-343
                    case FT_LOWPASS: d.kind = K_SVF_LP; break;
                    case FT_HIGHPASS: d.kind = K_SVF_HP; d.c[3].f = q.svm1; break;
                    case FT_PEAKING: d.kind = K_SVF_PK; d.c[3].f = q.svm1; break;
                    default: d.kind = K_SVF_SHELF; d.c[3].f = q.svm0; d.c[4].f = q.svm1; d.c[5].f = q.svm2; break;
                }
            } else {
                d.kind = K_BIQUAD;
                d.c[0] = q.b0; d.c[1] = q.b1; d.c[2] = q.b2; d.c[3] = q.a1; d.c[4] = q.a2;
            }
        }
    const bool loud_on = loudness_enabled && loud_row >= 0;       // "loud_on && loud_coeffs", usb_audio.c:689 / :1018
    for (int j = 0; j < 2; j++) {
        DevBand &d = img.loud[j];
        d.kind = K_BYPASS;
        if (!loud_on) continue;
        const LoudCoeffs &l = loud_table[loud_row][j];
        if (l.bypass) continue;
        d.kind = flavor ? K_SVF_SHELF : K_BIQUAD;
        for (int k = 0; k < 6; k++) d.c[k] = l.c[k];
    }
    uint32_t fl = 0;
    if (bypass_master_eq) fl |= IF_BYPASS_MASTER_EQ;
    if (!leveller_bypassed) fl |= IF_LEVELLER_ON;
    if (lev_cfg.lookahead) fl |= IF_LOOKAHEAD;
    if (!crossfeed_bypassed) fl |= IF_CROSSFEED_ON;
    if (core1_mode != 2) fl |= IF_SUB_ACTIVE;
    if (any_delay_active) fl |= IF_ANY_DELAY;
    if (fma_contract) fl |= IF_FMA;
    img.flags = fl;
    for (int ch = 0; ch < n_ch; ch++) if (channel_bypassed[ch]) img.ch_bypassed |= 1u << ch;
    img.fs_hz = freq;
    {
        uint64_t s = ((uint64_t)freq * 8u + 999u) / 1000u;   // PRESET_MUTE_TRANSITION_MS = 8, usb_audio.c:456-464
        if (s < 1) s = 1;
        img.mute_transition = (uint32_t)s;
    }
    for (int i = 0; i < 2; i++) { if (flavor) img.preamp[i].f = preamp_linear[i]; else img.preamp[i].i = preamp_mul[i]; }
    if (flavor) {
        img.vol.f = mute ? 0.0f : (float)vol_mul * (1.0f / 32768.0f);      // usb_audio.c:569
        img.master.f = master_linear;
    } else {
        img.vol.i = mute ? 0 : (int32_t)vol_mul;                           // :975
        img.master.i = master_q15;
    }
    for (int o = 0; o < n_out; o++) {
        if (outs[o].enabled) img.out_enabled |= 1u << o;
        if (outs[o].mute) img.out_mute |= 1u << o;
        img.out_gain_lin[o] = outs[o].gain_linear;
        img.delay_samples[o] = delay_samples[o];
        for (int inp = 0; inp < 2; inp++) {
            const Crosspoint &x = xp[inp][o];
            float g = x.enabled ? (x.phase_invert ? -x.gain_linear : x.gain_linear) : 0.0f;   // :763-764 / :1084-1085
            if (flavor) img.mix[inp][o].f = g;
            else img.mix[inp][o].i = x.enabled ? f2i_sat(g * 32768.0f) : 0;
        }
    }
    img.lv_alpha_rms = lv_alpha_rms; img.lv_alpha_attack = lv_alpha_attack; img.lv_alpha_release = lv_alpha_release;
    img.lv_threshold_db = lv_threshold_db; img.lv_ratio = lv_ratio; img.lv_knee_db = lv_knee_db; img.lv_makeup_db = lv_makeup_db;
    img.lv_gate_db = lv_gate_db; img.lv_max_gain_db = lv_max_gain_db;
    img.lv_alpha_rms_q28 = f2i_sat(lv_alpha_rms * (float)(1 << 28));
    img.xf_lp_a0 = xf_lp_a0; img.xf_lp_b1 = xf_lp_b1; img.xf_ap_a = xf_ap_a;
    for (int i = 0; i < n_pairs; i++) if (output_types[i] == 1) img.i2s_pairs |= 1u << i;
}

}  // namespace dspi
