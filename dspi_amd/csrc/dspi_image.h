// dspi_image.h — device-side parameter image and per-stream state layout (host + device).
//
// A DevImage is the "compiled" form of one DSPi parameter set: what process_audio_packet
// (reference firmware/DSPi/usb_audio.c:560-1283) reads from its globals at the top of a packet,
// flattened so that a wavefront fetches it with scalar loads (all 64 lanes = 64 streams of one
// workgroup share the image; see DESIGN.md "Data layout").
#pragma once
#include <stdint.h>

namespace dspi {

constexpr int kMaxCh = 11;        // RP2350: 2 master + 8 S/PDIF + 1 PDM   (config.h:315-322)
constexpr int kMaxOut = 9;
constexpr int kBands = 10;        // channel_band_counts (dsp_pipeline.c:36-44); 12 are stored in blobs
constexpr int kStoredBands = 12;
constexpr int kLanes = 64;        // streams per workgroup = wavefront width
constexpr int kRingLen = 1024;    // leveller block/lookahead ring (>= 480 + 2*192)
constexpr int kLookahead = 480;   // LEVELLER_LOOKAHEAD_SAMPLES (leveller.h:34)

// How a band is executed.  The four SVF forms are the reference's per-type inner loops
// (dsp_pipeline.c:298-343); K_BIQUAD is the TDF2 loop (:347-362, Q28: dsp_process_rp2040.S:263-365).
enum BandKind : uint32_t { K_BYPASS = 0, K_BIQUAD = 1, K_SVF_LP = 2, K_SVF_HP = 3, K_SVF_PK = 4, K_SVF_SHELF = 5 };

union Word {
    float f;
    int32_t i;
    uint32_t u;
};

// 32 bytes: one s_load_dwordx8.
//   K_BIQUAD    c = b0 b1 b2 a1 a2        (float, or Q28 int32)
//   K_SVF_LP    c = a1 a2 a3
//   K_SVF_HP/PK c = a1 a2 a3 m1
//   K_SVF_SHELF c = a1 a2 a3 m0 m1 m2
struct alignas(32) DevBand {
    Word c[6];
    uint32_t kind;
    uint32_t pad;
};

enum ImageFlags : uint32_t {
    IF_BYPASS_MASTER_EQ = 1u << 0,   // bypass_master_eq (usb_audio.c:576)
    IF_LEVELLER_ON = 1u << 1,        // !leveller_bypassed
    IF_LOOKAHEAD = 1u << 2,          // leveller_config.lookahead
    IF_CROSSFEED_ON = 1u << 3,       // !crossfeed_bypassed
    IF_SUB_ACTIVE = 1u << 4,         // core1_mode != CORE1_MODE_EQ_WORKER (usb_audio.c:782 vs :873)
    IF_ANY_DELAY = 1u << 5,          // any_delay_active
    IF_FMA = 1u << 6,                // float contract of the firmware build: contracted multiply-adds (context-wide, dspi.h)
};

struct DevImage {
    DevBand eq[kMaxCh][kBands];
    DevBand loud[2];                 // loudness shelves of the selected volume row; K_BYPASS when off
    uint32_t flags;
    uint32_t ch_bypassed;            // bit ch: channel_bypassed[ch]
    uint32_t out_enabled;            // bit o : matrix_mixer.outputs[o].enabled
    uint32_t out_mute;               // bit o : matrix_mixer.outputs[o].mute
    uint32_t fs_hz;
    uint32_t mute_transition;        // preset_mute_transition_samples(fs) (usb_audio.c:459-464)
    Word preamp[2];                  // f32: global_preamp_linear ; q28: global_preamp_mul
    Word vol;                        // f32: mute ? 0 : vol_mul/32768 ; q28: mute ? 0 : vol_mul (int16 incl. sign quirk)
    Word master;                     // f32: master_volume_linear ; q28: master_volume_q15
    Word mix[2][kMaxOut];            // crosspoint gains with phase and enable folded in (f32 / Q15)
    float out_gain_lin[kMaxOut];
    int32_t delay_samples[kMaxOut];
    // leveller (leveller.h:80-96)
    float lv_alpha_rms, lv_alpha_attack, lv_alpha_release, lv_threshold_db, lv_ratio, lv_knee_db, lv_makeup_db, lv_gate_db, lv_max_gain_db;
    int32_t lv_alpha_rms_q28;        // (int32)(alpha_rms * 2^28)  (leveller.c:286)
    // crossfeed (crossfeed.h:45-59)
    Word xf_lp_a0, xf_lp_b1, xf_ap_a;
    uint32_t i2s_pairs;              // bit p: output slot p is an I2S slot (output_types[p] == 1, config.h:286-287); read with DSPI_OUT_I2S_SLOTS
    uint32_t pad_[2];
};

// ------------------------------------------------------------------------------------------
// Value tile of a row (packed float kernel, per-lane values): what DIFFERS between the presets of a row whose 128 streams
// share one structure (flags, band kinds, bypasses, enables, delays in samples, zero pattern of the crosspoints) but not
// their numbers.  Lane l holds streams 2l (a) and 2l+1 (b) of the row; every access below is one coalesced row per wave.
//   bands    float4 [kPvBandSlots][3][64 lanes] = {c[2j].a, c[2j].b, c[2j+1].a, c[2j+1].b}   slot = ch*kBands + band, then loud[0..1]
//   scalars  float2 [PV_COUNT][64 lanes]        = {value of a, value of b}
//   mask     4 words: which band slots differ between the row's streams at all
// ------------------------------------------------------------------------------------------
constexpr int kPvBandSlots = kMaxCh * kBands + 2;
enum PvScalar : int {
    PV_PREAMP0 = 0, PV_PREAMP1, PV_VOL, PV_MASTER,
    PV_MIX0,                              // + o
    PV_MIX1 = PV_MIX0 + kMaxOut,          // + o
    PV_OG = PV_MIX1 + kMaxOut,            // + o : out_gain_lin
    PV_LV = PV_OG + kMaxOut,              // alpha_rms attack release threshold ratio knee makeup gate max_gain
    PV_XF = PV_LV + 9,                    // lp_a0 lp_b1 ap_a
    PV_COUNT = PV_XF + 3,
};
constexpr int kPvBandFloats = kPvBandSlots * 3 * kLanes * 4;            // 86 016
constexpr int kPvMaskWord = kPvBandFloats + PV_COUNT * kLanes * 2;      // 4 words: bit (slot) set = the band's coefficients DIFFER between the
                                                                        // row's streams; a channel whose ten bits are clear runs on the image's scalars
constexpr int kPvTileFloats = kPvMaskWord + 64;                         // 91 712 floats = 366 848 bytes per row

// ------------------------------------------------------------------------------------------
// Per-stream state: [workgroup][slot][lane] 32-bit words.  Slots 0..lds_slots-1 are staged in
// LDS for the whole launch; the rest live in VGPRs of the wave that owns them.
// ------------------------------------------------------------------------------------------
struct StateMap {
    int n_ch, n_out, n_pairs, max_delay;
    int row;       // streams per workgroup = width of every [..][row] device array: 128 (float: 2 per lane) / 64 (Q28)
    int eq;        // (ch*kBands + band)*2 + {0,1}           : s1,s2  or ic1eq,ic2eq
    int loud;      // ((ch*2 + stage)*2 + {0,1}
    int lds_slots; // = loud + 8
    int xfeed;     // lp_L lp_R ap_L ap_R
    int lev;       // env_l env_r gain_smooth_db gain_cur gain_prev
    int ring_pos;  // write position in the leveller ring
    int widx;      // delay_write_idx
    int mute;      // preset_loading, preset_mute_counter, preset_mute_smooth_gain
    int peaks;     // last packet's peak meter per channel (uint16 in the low half)
    int clip;      // sticky clip bits, one slot per wave of the workgroup (OR them)
    int n_slots;
};

constexpr StateMap make_state_map(int flavor) {
    StateMap m{};
    m.n_ch = flavor ? 11 : 7;
    m.n_out = flavor ? 9 : 5;
    m.n_pairs = flavor ? 4 : 2;
    m.max_delay = flavor ? 4096 : 2048;
    m.row = flavor ? 128 : 64;
    m.eq = 0;
    m.loud = m.n_ch * kBands * 2;
    m.lds_slots = m.loud + 8;
    m.xfeed = m.lds_slots;
    m.lev = m.xfeed + 4;
    m.ring_pos = m.lev + 5;
    m.widx = m.ring_pos + 1;
    m.mute = m.widx + 1;
    m.peaks = m.mute + 3;
    m.clip = m.peaks + m.n_ch;
    m.n_slots = m.clip + 4;
    return m;
}

// pending state mutations of an image, applied to all of its streams before the next packet
struct StateOps {
    uint32_t reset_band[kMaxCh];   // bit b: zero the state pair of (ch, band b)
    uint32_t reset_all_eq;         // dsp_init_default_filters memset (dsp_pipeline.c:178)
    uint32_t reset_crossfeed;      // crossfeed_compute_coefficients clears state (crossfeed.c:122-126)
    uint32_t reset_leveller;       // leveller_reset_state (leveller.c:95-105)
    uint32_t zero_delay_lines;     // preset_load (flash_storage.c:832)
    uint32_t mute_start;           // prepare_pipeline_reset (main.c:449-458)
    uint32_t mute_samples;
    uint32_t mute_cancel;          // preset_load failure path: preset_loading = false (flash_storage.c:806)
    uint32_t clear_clips;          // REQ_CLEAR_CLIPS
};

struct WgItem {
    uint32_t wg;
    uint32_t image;  // chain launches: index of the workgroup's parameter image (one launch covers every image)
    uint64_t mask;   // lanes of this workgroup that take part in the launch
    uint64_t mask1;  // state_ops only (float flavour): second stream of each lane; chain launches ignore it
};

}  // namespace dspi
