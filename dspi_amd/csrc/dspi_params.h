// dspi_params.h — host-side mirror of the firmware's parameter model (one DSPi "device").
//
// Holds what the reference keeps in globals (firmware/DSPi/usb_audio.c:47-214,
// dsp_pipeline.c:19-34) and implements the operations the USB control plane performs on them:
// vendor SET/GET, bulk blob apply/collect, preset slot apply/collect, factory defaults, UAC1
// volume/mute/rate, plus the coefficient design that the firmware's main loop runs between
// packets (main.c:826-894).  Pure host code — no HIP.  The result is flattened into a
// DevImage for the kernels, and a StateOps record of the per-stream state the firmware would
// have reset as a side effect.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "dspi_image.h"

namespace dspi {

#pragma pack(push, 1)
struct Recipe {   // == EqParamPacket, config.h:445-453
    uint8_t channel, band, type, reserved;
    float freq, Q, gain_db;
};
#pragma pack(pop)
static_assert(sizeof(Recipe) == 16, "EqParamPacket is 16 bytes");

enum FilterType : uint8_t { FT_FLAT = 0, FT_PEAKING, FT_LOWSHELF, FT_HIGHSHELF, FT_LOWPASS, FT_HIGHPASS };

// the preset directory sector of a flash dump (flash_storage.c:95-131), version 1 or 2, after CRC validation
struct FlashDirectory {
    int valid;                 // 0: no / corrupt / unknown-version directory
    int version;
    uint8_t startup_mode, default_slot, last_active_slot, include_pins;
    uint16_t slot_occupied;
    uint8_t master_volume_mode;
    float master_volume_db;
    char slot_names[10][32];
};
constexpr size_t kFlashDumpBytes = 12 * 4096;      // directory + 10 slots + legacy sector (flash_storage.c:4-15)
bool parse_flash_directory(const void *dump, size_t len, FlashDirectory &out);

struct BandCoeffs {          // derived; the coefficient half of the reference's Biquad (config.h:417-438)
    Word b0, b1, b2, a1, a2; // float flavour: .f ; Q28 flavour: .i
    float sva1, sva2, sva3, svm0, svm1, svm2;
    uint32_t svf_type;
    bool use_svf;
    bool bypass;
};

struct Crosspoint { uint8_t enabled, phase_invert; float gain_db, gain_linear; };
struct OutputCh { uint8_t enabled, mute; float gain_db, gain_linear, delay_ms; };
struct LoudCoeffs { Word c[6]; bool bypass; };   // f32: sva1..3, svm0..2 ; q28: b0 b1 b2 a1 a2

class Params {
public:
    explicit Params(int flavor, bool fma_contract = false, bool first_boot = true);

    // ---- operations (each cites the reference entry point in dspi_params.cpp) ----
    int boot(const void *flash = nullptr);      // the power-on sequence (over a 48 KB preset area, or none)
    void factory_reset();
    int load_bulk(const void *blob, size_t len);
    int collect_bulk(void *blob, size_t cap) const;
    int load_slot(const void *image, size_t len, int expect_slot);
    bool slot_to_live(const void *image, int expect_slot);      // apply_slot_to_live + apply_master_volume_from_mode, nothing else
    int flash_select(const void *dump, bool booting);           // preset_boot_load's selection; the application by `booting`
    int save_slot(void *image, size_t cap, int slot_index) const;
    int load_flash_dump(const void *dump, size_t len, bool as_boot = false);      // preset_boot_load on a 48 KB flash image: a running device switching to the selected preset, or (as_boot) the device booting from it
    int vendor_set(uint8_t req, uint16_t wValue, const void *payload, uint16_t len);
    int vendor_get(uint8_t req, uint16_t wValue, void *buf, uint16_t cap, const uint16_t *peaks, uint16_t *clip_flags);
    void set_volume(int16_t v);
    void set_mute(bool m);
    int set_rate(uint32_t hz);

    void build_image(DevImage &img) const;
    int slot_size() const;

    // ---- firmware-visible parameter state ----
    int flavor;
    bool first_boot;                    // power-on of a device with an erased flash: the boot writes the preset directory, which arms the 512-sample mute
    bool fma_contract;                  // float flavour as the firmware is built: GCC's FMA contraction (dspi.h DSPI_FLOAT_CONTRACT_FMA)
    int n_ch, n_out, n_pairs, n_pins, max_delay;
    Recipe recipes[kMaxCh][kStoredBands];
    float channel_delays_ms[kMaxCh];
    bool bypass_master_eq;
    float preamp_db[2];
    int32_t preamp_mul[2];
    float preamp_linear[2];
    float master_db, master_linear;
    int32_t master_q15;
    float legacy_gain_db[3];
    int32_t legacy_gain_mul[3];
    float legacy_gain_linear[3];
    bool legacy_mute[3];
    Crosspoint xp[2][kMaxOut];
    OutputCh outs[kMaxOut];
    bool loudness_enabled;
    float loudness_ref_spl, loudness_intensity_pct;
    struct { bool enabled, itd_enabled; uint8_t preset; float custom_fc, custom_feed_db; } xfeed_cfg;
    struct { bool enabled; float amount; uint8_t speed; float max_gain_db; bool lookahead; float gate_threshold_db; } lev_cfg;
    char names[kMaxCh][32];
    uint8_t pins[5];
    uint8_t output_types[4];
    uint8_t i2s_bck_pin, i2s_mck_pin;
    bool i2s_mck_enabled;
    uint16_t i2s_mck_multiplier;
    uint32_t freq;
    int16_t volume, vol_mul;
    bool mute;
    uint8_t dir_master_volume_mode, dir_include_pins;
    float dir_master_volume_db;

    // ---- derived (what the firmware recomputes in its main loop) ----
    BandCoeffs bands[kMaxCh][kStoredBands];
    bool channel_bypassed[kMaxCh];
    int32_t delay_samples[kMaxOut];
    bool any_delay_active;
    LoudCoeffs loud_table[61][2];
    bool loud_table_valid;
    int loud_row;                       // -1: current_loudness_coeffs == NULL
    Word xf_lp_a0, xf_lp_b1, xf_ap_a;
    bool crossfeed_bypassed;
    float lv_alpha_rms, lv_alpha_attack, lv_alpha_release, lv_threshold_db, lv_ratio, lv_knee_db, lv_makeup_db, lv_gate_db, lv_max_gain_db;
    bool leveller_bypassed;
    int core1_mode;                     // 0 idle, 1 PDM, 2 EQ worker (config.h:344-348)

    // ---- side effects on per-stream state, drained by the context before the next packet ----
    StateOps ops;
    bool dirty;                         // image must be rebuilt/uploaded

private:
    bool loud_pending, xfeed_pending, lev_pending, lev_reset_pending;
    void design_band(Recipe &r, int ch, int b, float fs);
    void recalc_channel_bypass(int ch);
    void recalc_all_filters(float fs);
    void update_delay_samples(float fs);
    void init_default_filters();
    void loudness_recompute(float fs);
    void crossfeed_design(float fs);
    void leveller_design(float fs);
    void service();
    void rate_changed(uint32_t hz);
    void apply_factory_defaults();
    void apply_master_from_mode(bool have_slot, uint16_t slot_version, float slot_db);
    void set_master_clamped(float db);
    void update_preamp(int ch, float db);
    void pipeline_mute(uint32_t samples);
    void transition_core1();
    void select_loud_row();
};

int32_t f2i_sat(float f);   // (int32_t)float as the MCUs do it: saturating, NaN -> 0

}  // namespace dspi
