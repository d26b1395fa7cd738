/*
 * orc_api.h — C API of the CPU oracle (both builds export exactly this).
 * TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never from dspi_amd/ or the C-ABI library.
 *
 * One context = one DSPi device = one stereo stream (the firmware's globals, per stream).
 * Parameter calls take effect immediately, i.e. at the next block boundary — the same
 * guarantee the firmware gives by deferring work to its main loop (main.c:826-894).
 */
#ifndef ORC_API_H
#define ORC_API_H
#include <stdint.h>

typedef struct orc_ctx orc_ctx;

int orc_flavor(void);            /* 1 = RP2350 float, 0 = RP2040 Q28 */
int orc_is_ref_build(void);      /* 1 when leaf DSP comes from /root/reference objects */
int orc_num_channels(void);      /* 11 / 7 */
int orc_num_outputs(void);       /* 9 / 5 */
int orc_num_pairs(void);         /* 4 / 2 */
int orc_preset_slot_size(void);
void orc_set_math_mode(int detmath);        /* leveller per-block log10f/powf: 0 glibc, 1 dspi_detmath.h */
void orc_set_x86_cast_semantics(int on);    /* see orc_common.h */
void orc_set_fma_mode(int on);              /* float contract of the restated code: 0 canonical, 1 firmware as built (orc_leaf.c);
                                             * the _ref / firmware builds take theirs from the compiler flags and only switch the
                                             * restated orchestrator */

orc_ctx *orc_new(void);          /* power-on state: factory defaults, 44.1 kHz, host volume 0 dB */
orc_ctx *orc_new_from_flash(const void *dump48k, uint32_t len, int *selection);   /* power-on of a device whose flash holds this preset area: no first-boot mute,
                                                                                    * the selected preset applied the boot path's way (flash_storage.c:1047-1082) */
void orc_free(orc_ctx *);

int orc_set_sample_rate(orc_ctx *, uint32_t hz);        /* 44100 / 48000 / 96000, else -1 */
void orc_set_host_volume(orc_ctx *, int16_t vol_1_256_db);
void orc_set_mute(orc_ctx *, int mute);
void orc_factory_defaults(orc_ctx *);
int orc_load_bulk(orc_ctx *, const void *blob, uint32_t len);     /* 0 or -1..-4 (bulk_params.c:181-203) */
int orc_collect_bulk(orc_ctx *, void *blob2896);
int orc_load_preset_slot(orc_ctx *, const void *image, uint32_t len, int expect_slot /* -1: any */);  /* PRESET_OK / PRESET_ERR_CRC */
int orc_save_preset_slot(orc_ctx *, void *image, int slot_index);
int orc_load_flash_dump(orc_ctx *, const void *dump48k, uint32_t len);   /* 0..9 | 16+n | 32 | 48 | -4, see orc_chain.c */
int orc_vendor_set(orc_ctx *, uint8_t bRequest, uint16_t wValue, const void *payload, uint16_t len);  /* 0, -1 unsupported */
int orc_vendor_get(orc_ctx *, uint8_t bRequest, uint16_t wValue, void *buf, uint16_t cap);            /* bytes, -1 unsupported/stall */
void orc_get_status(orc_ctx *, void *buf);  /* REQ_GET_STATUS wValue 9 layout: 26 B float flavour, 18 B Q28 */

/* pcm: interleaved LE stereo, 16-bit (4 B/frame) or packed 24-bit (6 B/frame), n_blocks*block_len frames.
 * pairs: [pair][frame][2] int32 (24-bit payload), sub: [frame] int32 Q28, peaks: [block][C] uint16 (optional). */
void orc_process(orc_ctx *, const void *pcm, int bit_depth, uint32_t n_blocks, uint32_t block_len,
                 int32_t *pairs, int32_t *sub, uint16_t *peaks, uint16_t *clip_flags);

/* per-band taps of one EQ channel, float flavour: taps [11][n] (orc_chain.c) */
int orc_debug_eq_taps(orc_ctx *, int channel, const float *x, uint32_t n, float *taps);
/* Q28 builds only: the block biquad (dsp_process_rp2040.S:225-394 restated) on caller-supplied bands */
void orc_debug_q28_biquad_block(const int32_t *coef, int32_t *state, const uint8_t *bypass, int32_t *x, uint32_t count, uint32_t nbands);

/* debugging taps used by tests */
const void *orc_tap(orc_ctx *, int what, int *bytes);
int orc_scalar(orc_ctx *, int what);
float orc_scalar_f(orc_ctx *, int what);
#endif
