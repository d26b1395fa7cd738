/*
 * ref_globals.c — definitions of the firmware globals that the reference's leaf sources
 * reference but usb_audio.c (not compilable here: needs pico-sdk) defines
 * (firmware/DSPi/usb_audio.c:47-214, :2038-2075).  `_ref` build only.
 * TEST INFRASTRUCTURE ONLY.  Values are irrelevant: orc_chain.c overwrites them from the
 * per-stream context before every call into a globals-based reference function.
 */
#include <math.h>
#include "config.h"
#include "usb_audio.h"
#include "leveller.h"
#include "../include/dspi_detmath.h"

volatile AudioState audio_state = {.freq = 44100};
volatile bool bypass_master_eq = false;
volatile SystemStatusPacket global_status = {0};
volatile float global_preamp_db[NUM_INPUT_CHANNELS];
volatile int32_t global_preamp_mul[NUM_INPUT_CHANNELS];
volatile float global_preamp_linear[NUM_INPUT_CHANNELS];
volatile float master_volume_db, master_volume_linear;
volatile int32_t master_volume_q15;
volatile float channel_gain_db[3];
volatile int32_t channel_gain_mul[3];
volatile float channel_gain_linear[3];
volatile bool channel_mute[3];
MatrixMixer matrix_mixer;
volatile bool loudness_enabled;
volatile float loudness_ref_spl, loudness_intensity_pct;
volatile bool loudness_recompute_pending;
volatile CrossfeedConfig crossfeed_config;
volatile bool crossfeed_update_pending;
volatile bool crossfeed_bypassed = true;
CrossfeedState crossfeed_state;
volatile LevellerConfig leveller_config;
volatile bool leveller_update_pending, leveller_reset_pending;
char channel_names[NUM_CHANNELS][PRESET_NAME_LEN];
uint8_t output_pins[NUM_PIN_OUTPUTS];
uint8_t output_types[NUM_SPDIF_INSTANCES];
uint8_t i2s_bck_pin, i2s_mck_pin;
bool i2s_mck_enabled;
uint16_t i2s_mck_multiplier = 128;

/* leveller.c is compiled with -include ref_math_hook.h so the per-block
 * gain step can run on either math library without touching the reference source. */
int orc_math_mode = 0;
int orc_fma_mode = 0;        /* contract of the restated orchestrator; the reference objects have theirs from the compiler */
#undef log10f
#undef powf
float orc_hook_log10f(float x) { return orc_math_mode ? dspi_det_log10f(x) : log10f(x); }
float orc_hook_powf(float a, float b) { return orc_math_mode ? dspi_det_powf(a, b) : powf(a, b); }
