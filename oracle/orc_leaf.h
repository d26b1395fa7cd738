/* orc_leaf.h — prototypes of the restated leaf functions (standalone oracle build only).
 * TEST INFRASTRUCTURE ONLY; see orc_leaf.c for the reference file:line of each function. */
#ifndef ORC_LEAF_H
#define ORC_LEAF_H
#include "orc_types.h"

extern int orc_math_mode;                 /* 0 libm, 1 detmath (leveller per-block step) */
extern int orc_fma_mode;                  /* 0 canonical (no contraction), 1 the firmware's float contract (orc_leaf.c) */
#if PICO_RP2350
typedef float orc_sample;
#else
typedef int32_t orc_sample;
int32_t orc_fast_mul_q28(int32_t a, int32_t b);
#endif

void orc_dsp_compute_coefficients(EqParamPacket *p, Biquad *bq, float sample_rate);
#if PICO_RP2350
void orc_dsp_process_channel_block(Biquad *bands, float *x, uint32_t n, uint8_t nbands);
#endif
void orc_leveller_compute_coefficients(LevellerCoeffs *out, const LevellerConfig *cfg, float fs);
void orc_leveller_reset_state(LevellerState *st);
void orc_leveller_process_block(LevellerState *st, const LevellerCoeffs *c, const LevellerConfig *cfg,
                                orc_sample *l, orc_sample *r, uint32_t count);
void orc_crossfeed_compute_coefficients(CrossfeedState *st, const CrossfeedConfig *cfg, float fs);
void orc_crossfeed_process_stereo(CrossfeedState *s, orc_sample *left, orc_sample *right);
void orc_loudness_build_table(LoudnessCoeffs table[LOUDNESS_VOL_STEPS][LOUDNESS_BIQUAD_COUNT],
                              float ref_spl, float pct, float fs);
#endif
