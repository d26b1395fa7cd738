/*
 * orc_leaf.c — CPU restatement of the DSPi leaf DSP functions (STANDALONE oracle build).
 *
 * TEST INFRASTRUCTURE ONLY.  Plain C restatement, one function per reference function, of
 *   firmware/DSPi/dsp_pipeline.c  (fast_mul_q28 :47-58, coefficient design :61-175,
 *                                  float block EQ :281-365)
 *   firmware/DSPi/leveller.c      (:37-105 design/reset, :124-139 gain computer,
 *                                  :148-262 float block, :275-389 Q28 block)
 *   firmware/DSPi/crossfeed.c     (:35-127 design, :132-180 per-frame process)
 *   firmware/DSPi/loudness.c      (:37-217 ISO-226 table)
 * In the `_ref` build these functions are NOT compiled; the orchestrator links the
 * reference's own objects instead, and tests/test_oracle_vs_ref.py checks the two builds
 * produce identical bits.  Compile with -O2 -fwrapv -ffp-contract=off.
 */
#include <math.h>
#include <string.h>
#include "orc_types.h"
#include "orc_leaf.h"
#include "orc_common.h"
#include "../include/dspi_detmath.h"

int orc_math_mode = 0; /* 0: host libm (glibc) ; 1: dspi_detmath.h */

/* Float contract.  0 = canonical: every multiply and add rounds on its own (-ffp-contract=off).
 * 1 = the firmware as built: GNU C's default -ffp-contract=fast on a target with fused multiply-add (Cortex-M33 vfma,
 * firmware/DSPi/CMakeLists.txt:6-11 sets only -O2/-O3), where GCC turns a*b + c into one fused operation.  Which pairs it
 * fuses is decided on GIMPLE (tree-ssa-math-opts.c, pass widening_mul) and was read off `-fdump-tree-optimized` of the
 * reference sources (oracle/Makefile FMA_FLAGS); every MAD() below is one of those `.FMA/.FMS/.FNMA` statements, written so
 * that mode 0 evaluates the reference's expression with separate roundings and mode 1 with GCC's fused ones.
 * tests/test_oracle_vs_ref.py and test_oracle_vs_fw.py hold this file to the reference compiled both ways, bit for bit.
 * The RP2040 has no FPU and no fused operation: the Q28 flavour ignores the switch. */
int orc_fma_mode = 0;
#if PICO_RP2350
#define MAD(a, b, c) (orc_fma_mode ? fmaf((a), (b), (c)) : (a) * (b) + (c))
#else
#define MAD(a, b, c) ((a) * (b) + (c))
#endif

static float lv_log10f(float x) { return orc_math_mode ? dspi_det_log10f(x) : log10f(x); }
static float lv_powf(float a, float b) { return orc_math_mode ? dspi_det_powf(a, b) : powf(a, b); }

#define PI_F 3.1415926535f

/* ===================================================================================== */
/* Q28 multiply                                                                          */
/* ===================================================================================== */
#if !PICO_RP2350
/* dsp_pipeline.c:47-58 — NOT a 64-bit product: al*bl is dropped and every step wraps. */
int32_t orc_fast_mul_q28(int32_t a, int32_t b) {
    int32_t ah = a >> 16, bh = b >> 16;
    uint32_t al = (uint32_t)a & 0xFFFFu, bl = (uint32_t)b & 0xFFFFu;
    uint32_t high = (uint32_t)(ah * bh);
    int32_t mid = (int32_t)((uint32_t)ah * bl + al * (uint32_t)bh);
    return (int32_t)((high << 4) + (uint32_t)(mid >> 12));
}
#endif

/* ===================================================================================== */
/* EQ coefficient design                                                                 */
/* ===================================================================================== */
static bool recipe_is_flat(const EqParamPacket *p) { /* dsp_pipeline.c:6-17 */
    if (p->type == FILTER_FLAT || p->freq <= 0.0f) return true;
    if (p->type == FILTER_PEAKING || p->type == FILTER_LOWSHELF || p->type == FILTER_HIGHSHELF)
        return fabsf(p->gain_db) < 0.01f;
    return false;
}

void orc_dsp_compute_coefficients(EqParamPacket *p, Biquad *bq, float fs) { /* :61-175 */
    if (recipe_is_flat(p) || fs == 0) {
        bq->bypass = true;
#if PICO_RP2350
        bq->b0 = 1.0f; bq->b1 = bq->b2 = bq->a1 = bq->a2 = 0.0f;
        bq->sva1 = bq->sva2 = bq->sva3 = 0.0f;
        bq->svm0 = bq->svm1 = bq->svm2 = 0.0f;
        bq->use_svf = false;
#else
        bq->b0 = 1 << FILTER_SHIFT; bq->b1 = bq->b2 = bq->a1 = bq->a2 = 0;
#endif
        return;
    }
    bq->bypass = false;
    /* the recipe itself is clamped in place (:78-81) */
    if (p->Q < 0.1f) p->Q = 0.1f;
    if (p->Q > 20.0f) p->Q = 20.0f;
    if (p->freq < 10.0f) p->freq = 10.0f;
    if (p->freq > fs * 0.45f) p->freq = fs * 0.45f;

    float A = powf(10.0f, p->gain_db / 40.0f);

#if PICO_RP2350
    bool was_svf = bq->use_svf;
    bq->use_svf = (p->freq < (fs / 7.5f));
    if (was_svf != bq->use_svf) { bq->s1 = bq->s2 = 0.0f; bq->svic1eq = bq->svic2eq = 0.0f; }

    if (bq->use_svf) {
        float g = tanf(PI_F * p->freq / fs);
        float k = 1.0f / p->Q;
        if (p->type == FILTER_PEAKING) k = 1.0f / (p->Q * A);
        else if (p->type == FILTER_LOWSHELF) { float r = sqrtf(A); g = g / r; }
        else if (p->type == FILTER_HIGHSHELF) { float r = sqrtf(A); g = g * r; }

        float c1 = 1.0f / MAD(g, g + k, 1.0f);                       /* 1 + g*(g+k) */
        float c2 = g * c1;
        float c3 = g * c2;
        float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
        switch (p->type) {
            case FILTER_LOWPASS:   m0 = 0.0f;  m1 = 0.0f;               m2 = 1.0f;          break;
            case FILTER_HIGHPASS:  m0 = 1.0f;  m1 = -k;                 m2 = -1.0f;         break;
            case FILTER_PEAKING:   m0 = 1.0f;  m1 = k * MAD(A, A, -1.0f); m2 = 0.0f;        break;
            case FILTER_LOWSHELF:  m0 = 1.0f;  m1 = k * (A - 1.0f);     m2 = MAD(A, A, -1.0f); break;
            case FILTER_HIGHSHELF: m0 = A * A; m1 = k * (1.0f - A) * A; m2 = 1.0f - m0;     break;   /* A*A is shared: no fusion */
            default: break;
        }
        bq->sva1 = c1; bq->sva2 = c2; bq->sva3 = c3;
        bq->svm0 = m0; bq->svm1 = m1; bq->svm2 = m2;
        bq->svf_type = p->type;
        bq->b0 = 1.0f; bq->b1 = bq->b2 = bq->a1 = bq->a2 = 0.0f;
        return;
    }
    bq->sva1 = bq->sva2 = bq->sva3 = 0.0f;
    bq->svm0 = bq->svm1 = bq->svm2 = 0.0f;
#endif

    /* RBJ cookbook, single precision throughout (:145-156) */
    float omega = 2.0f * PI_F * p->freq / fs;
    float sn = sinf(omega), cs = cosf(omega);
    float alpha = sn / (2.0f * p->Q);
    float a0 = 1.0f, a1 = 0.0f, a2 = 0.0f, b0 = 1.0f, b1 = 0.0f, b2 = 0.0f;
    switch (p->type) {
        case FILTER_LOWPASS:
            b0 = (1 - cs) / 2; b1 = 1 - cs; b2 = (1 - cs) / 2;
            a0 = 1 + alpha; a1 = -2 * cs; a2 = 1 - alpha; break;
        case FILTER_HIGHPASS:
            b0 = (1 + cs) / 2; b1 = -(1 + cs); b2 = (1 + cs) / 2;
            a0 = 1 + alpha; a1 = -2 * cs; a2 = 1 - alpha; break;
        case FILTER_PEAKING:
            b0 = MAD(alpha, A, 1.0f); b1 = -2 * cs; b2 = MAD(-alpha, A, 1.0f);           /* 1 + alpha*A, 1 - alpha*A */
            a0 = 1 + alpha / A; a1 = -2 * cs; a2 = 1 - alpha / A; break;
        case FILTER_LOWSHELF: {
            const float Ap = A + 1, Am = A - 1, t = Am * cs, S2 = 2 * sqrtf(A);
            const float u = Ap - t, w = Ap + t;                                          /* the terms GCC keeps rounded */
            b0 = A * MAD(S2, alpha, u);
            b1 = 2 * A * MAD(-Ap, cs, Am);                                               /* (A-1) - (A+1)*cs */
            b2 = A * MAD(-S2, alpha, u);
            a0 = MAD(S2, alpha, w);
            a1 = -2 * MAD(Ap, cs, Am);                                                   /* (A-1) + (A+1)*cs */
            a2 = MAD(-S2, alpha, w); break;
        }
        case FILTER_HIGHSHELF: {
            const float Ap = A + 1, Am = A - 1, t = Am * cs, S2 = 2 * sqrtf(A);
            const float u = Ap + t, w = Ap - t;
            b0 = A * MAD(S2, alpha, u);
            b1 = -2 * A * MAD(Ap, cs, Am);
            b2 = A * MAD(-S2, alpha, u);
            a0 = MAD(S2, alpha, w);
            a1 = 2 * MAD(-Ap, cs, Am);
            a2 = MAD(-S2, alpha, w); break;
        }
        default: break;
    }
#if PICO_RP2350
    float inv = 1.0f / a0;
    bq->b0 = b0 * inv; bq->b1 = b1 * inv; bq->b2 = b2 * inv; bq->a1 = a1 * inv; bq->a2 = a2 * inv;
#else
    float scale = (float)(1LL << FILTER_SHIFT);
    bq->b0 = orc_f2i((b0 / a0) * scale);
    bq->b1 = orc_f2i((b1 / a0) * scale);
    bq->b2 = orc_f2i((b2 / a0) * scale);
    bq->a1 = orc_f2i((a1 / a0) * scale);
    bq->a2 = orc_f2i((a2 / a0) * scale);
#endif
}

/* ===================================================================================== */
/* Float block EQ (SVF / TDF2 hybrid)                                                     */
/* ===================================================================================== */
#if PICO_RP2350
void orc_dsp_process_channel_block(Biquad *bands, float *x, uint32_t n, uint8_t nbands) { /* :281-365 */
    for (int b = 0; b < nbands; b++) {
        Biquad *q = &bands[b];
        if (q->bypass) continue;
        if (q->use_svf) {
            const float a1 = q->sva1, a2 = q->sva2, a3 = q->sva3;
            const float m0 = q->svm0, m1 = q->svm1, m2 = q->svm2;
            float e1 = q->svic1eq, e2 = q->svic2eq;
            const uint32_t ty = q->svf_type;
            for (uint32_t i = 0; i < n; i++) {
                float in = x[i];
                float v3 = in - e2;
                float v1 = MAD(a1, e1, a2 * v3);                       /* a1*ic1 fused, a2*v3 rounded */
                float v2 = MAD(a3, v3, MAD(a2, e1, e2));               /* (ic2 + a2*ic1) + a3*v3, both fused */
                e1 = MAD(2.0f, v1, -e1);
                e2 = MAD(2.0f, v2, -e2);
                /* per-type output forms; the association below is the reference's */
                if (ty == FILTER_LOWPASS) x[i] = v2;
                else if (ty == FILTER_HIGHPASS) x[i] = MAD(m1, v1, in) - v2;
                else if (ty == FILTER_PEAKING) x[i] = MAD(m1, v1, in);
                else x[i] = MAD(m2, v2, MAD(m0, in, m1 * v1));         /* (m0*in + m1*v1) + m2*v2: m1*v1 rounded */
            }
            q->svic1eq = e1; q->svic2eq = e2;
        } else {
            const float b0 = q->b0, b1 = q->b1, b2 = q->b2, a1 = q->a1, a2 = q->a2;
            float s1 = q->s1, s2 = q->s2;
            for (uint32_t i = 0; i < n; i++) {
                float in = x[i];
                float y = MAD(b0, in, s1);
                s1 = MAD(b1, in, -(a1 * y)) + s2;                      /* b1*in fused with the rounded a1*y; + s2 separate */
                s2 = MAD(b2, in, -(a2 * y));
                x[i] = y;
            }
            q->s1 = s1; q->s2 = s2;
        }
    }
}
#endif

/* ===================================================================================== */
/* Leveller                                                                               */
/* ===================================================================================== */
static float lv_alpha(float fs, float t) { /* leveller.c:37-40 */
    if (t <= 0.0f || fs <= 0.0f) return 0.0f;
    return expf(-logf(10.0f) / (fs * t));
}

void orc_leveller_compute_coefficients(LevellerCoeffs *out, const LevellerConfig *cfg, float fs) { /* :42-89 */
    static const float presets[LEVELLER_SPEED_COUNT][3] = {
        {0.100f, 2.000f, 0.400f}, {0.050f, 1.000f, 0.200f}, {0.020f, 0.500f, 0.100f}};
    if (fs < 1.0f) fs = 48000.0f;
    uint8_t spd = cfg->speed;
    if (spd >= LEVELLER_SPEED_COUNT) spd = LEVELLER_SPEED_MEDIUM;
    out->alpha_rms = lv_alpha(fs, presets[spd][2]);
    out->alpha_attack = lv_alpha(fs, presets[spd][0]);
    out->alpha_release = lv_alpha(fs, presets[spd][1]);
    out->threshold_db = LEVELLER_THRESHOLD_DB;
    out->knee_width_db = LEVELLER_KNEE_WIDTH_DB;
    float gate = cfg->gate_threshold_db;
    if (gate < LEVELLER_GATE_MIN) gate = LEVELLER_GATE_MIN;
    if (gate > LEVELLER_GATE_MAX) gate = LEVELLER_GATE_MAX;
    out->gate_threshold_db = gate;
    float amount = cfg->amount;
    if (amount < LEVELLER_AMOUNT_MIN) amount = LEVELLER_AMOUNT_MIN;
    if (amount > LEVELLER_AMOUNT_MAX) amount = LEVELLER_AMOUNT_MAX;
    float norm = amount / 100.0f;
    out->ratio = MAD(norm, 19.0f, 1.0f);
    float mg = cfg->max_gain_db;
    if (mg < LEVELLER_MAX_GAIN_MIN) mg = LEVELLER_MAX_GAIN_MIN;
    if (mg > LEVELLER_MAX_GAIN_MAX) mg = LEVELLER_MAX_GAIN_MAX;
    out->max_gain_db = mg;
    out->makeup_db = 0.0f;
}

void orc_leveller_reset_state(LevellerState *st) { /* :95-105 */
    memset(st, 0, sizeof(*st));
#if PICO_RP2350
    st->gain_linear = 1.0f; st->gain_prev_linear = 1.0f;
#else
    st->gain_q28 = 1 << FILTER_SHIFT; st->gain_prev_q28 = 1 << FILTER_SHIFT;
#endif
    st->gain_smooth_db = 0.0f;
}

static float lv_gain_computer(float x_db, float thr, float ratio, float knee) { /* :124-139 */
    float half = knee * 0.5f;
    if (x_db > (thr + half)) return 0.0f;
    if (x_db >= (thr - half)) {
        float d = thr + half - x_db;
        return (1.0f - 1.0f / ratio) * d * d / (2.0f * knee);
    }
    return (thr - x_db) * (1.0f - 1.0f / ratio);
}

/* per-block gain decision shared by both flavours (:174-206 / :304-332) */
static float lv_block_gain_db(LevellerState *st, const LevellerCoeffs *c, float rms_sq, uint32_t count) {
    float rms_db = 10.0f * lv_log10f(rms_sq + 1e-30f);
    float gc;
    if (rms_db < c->gate_threshold_db) gc = 0.0f;
    else {
        gc = lv_gain_computer(rms_db, c->threshold_db, c->ratio, c->knee_width_db);
        gc += c->makeup_db;
        if (gc > c->max_gain_db) gc = c->max_gain_db;
    }
    float a_s = (gc < st->gain_smooth_db) ? c->alpha_attack : c->alpha_release;
    float alpha = lv_powf(a_s, (float)count);
    st->gain_smooth_db = MAD(alpha, st->gain_smooth_db, (1.0f - alpha) * gc);
    return lv_powf(10.0f, st->gain_smooth_db / 20.0f);
}

#if PICO_RP2350
void orc_leveller_process_block(LevellerState *st, const LevellerCoeffs *c, const LevellerConfig *cfg,
                                float *l, float *r, uint32_t count) { /* :148-262 */
    if (count == 0) return;
    float el = st->env_sq_l, er = st->env_sq_r;
    const float a = c->alpha_rms, na = 1.0f - a;
    for (uint32_t i = 0; i < count; i++) {
        float sl = l[i], sr = r[i];
        el = MAD(a, el, na * (sl * sl));
        er = MAD(a, er, na * (sr * sr));
    }
    if (el < 1e-30f) el = 0.0f;
    if (er < 1e-30f) er = 0.0f;
    st->env_sq_l = el; st->env_sq_r = er;

    float g_new = lv_block_gain_db(st, c, el > er ? el : er, count);
    st->gain_prev_linear = st->gain_linear;
    st->gain_linear = g_new;

    float g_prev = st->gain_prev_linear, g_cur = st->gain_linear, gain, step;
    if (count == 1) { gain = g_cur; step = 0.0f; }
    else { step = (g_cur - g_prev) / (float)(count - 1); gain = g_prev; }

    const float ceil_ = LEVELLER_LIMITER_CEIL;
    bool la = cfg->lookahead;
    uint32_t idx = st->la_write_idx;
    for (uint32_t i = 0; i < count; i++) {
        float ol, or_;
        if (la) {
            ol = st->lookahead_buf[0][idx]; or_ = st->lookahead_buf[1][idx];
            st->lookahead_buf[0][idx] = l[i]; st->lookahead_buf[1][idx] = r[i];
            if (++idx >= LEVELLER_LOOKAHEAD_SAMPLES) idx = 0;
        } else { ol = l[i]; or_ = r[i]; }
        float peak = fabsf(ol), pr = fabsf(or_);
        if (pr > peak) peak = pr;
        float g = gain;
        if (peak > 0.0f && g > 1.0f) {
            float mg = ceil_ / peak;
            if (mg < g) g = (mg > 1.0f) ? mg : 1.0f;
        }
        l[i] = ol * g; r[i] = or_ * g;
        gain += step;
    }
    st->la_write_idx = idx;
}
#else
void orc_leveller_process_block(LevellerState *st, const LevellerCoeffs *c, const LevellerConfig *cfg,
                                int32_t *l, int32_t *r, uint32_t count) { /* :275-389 */
    if (count == 0) return;
    int32_t a_q = orc_f2i(c->alpha_rms * (float)(1 << FILTER_SHIFT));
    int32_t na_q = (1 << FILTER_SHIFT) - a_q;
    int32_t el = st->env_sq_l, er = st->env_sq_r;
    for (uint32_t i = 0; i < count; i++) {
        int32_t sl = l[i], sr = r[i];
        int32_t ql = orc_fast_mul_q28(sl, sl), qr = orc_fast_mul_q28(sr, sr);
        el = (int32_t)((uint32_t)orc_fast_mul_q28(a_q, el) + (uint32_t)orc_fast_mul_q28(na_q, ql));
        er = (int32_t)((uint32_t)orc_fast_mul_q28(a_q, er) + (uint32_t)orc_fast_mul_q28(na_q, qr));
    }
    st->env_sq_l = el; st->env_sq_r = er;

    const float inv = 1.0f / (float)(1 << FILTER_SHIFT);
    float elf = (float)el * inv, erf = (float)er * inv;
    float g_lin = lv_block_gain_db(st, c, elf > erf ? elf : erf, count);
    st->gain_prev_q28 = st->gain_q28;
    st->gain_q28 = orc_f2i(g_lin * (float)(1 << FILTER_SHIFT));   /* >= 18.06 dB saturates on ARM */

    int32_t gp = st->gain_prev_q28, gc = st->gain_q28;
    const int32_t unity = 1 << FILTER_SHIFT;
    const float ceil_ = LEVELLER_LIMITER_CEIL;
    bool la = cfg->lookahead;
    uint32_t idx = st->la_write_idx;
    for (uint32_t i = 0; i < count; i++) {
        int32_t gain;
        if (count == 1) gain = gc;
        else gain = gp + (int32_t)(((int64_t)(gc - gp) * i) / (int32_t)(count - 1));
        int32_t ol, or_;
        if (la) {
            ol = st->lookahead_buf[0][idx]; or_ = st->lookahead_buf[1][idx];
            st->lookahead_buf[0][idx] = l[i]; st->lookahead_buf[1][idx] = r[i];
            if (++idx >= LEVELLER_LOOKAHEAD_SAMPLES) idx = 0;
        } else { ol = l[i]; or_ = r[i]; }
        if (gain > unity) {
            float peak = fabsf((float)ol * inv), pr = fabsf((float)or_ * inv);
            if (pr > peak) peak = pr;
            if (peak > 0.0f) {
                float mgf = ceil_ / peak;
                int32_t mgq = orc_f2i(mgf * (float)unity);
                if (mgq < gain) gain = (mgq > unity) ? mgq : unity;
            }
        }
        l[i] = orc_fast_mul_q28(ol, gain);
        r[i] = orc_fast_mul_q28(or_, gain);
    }
    st->la_write_idx = idx;
}
#endif

/* ===================================================================================== */
/* Crossfeed                                                                              */
/* ===================================================================================== */
void orc_crossfeed_compute_coefficients(CrossfeedState *st, const CrossfeedConfig *cfg, float fs) { /* :35-127 */
    static const float presets[3][2] = {{700.0f, 4.5f}, {700.0f, 6.0f}, {650.0f, 9.5f}};
    if (!cfg->enabled || fs < 1.0f) { memset(st, 0, sizeof(*st)); return; }
    float fc, feed;
    if (cfg->preset < 3) { fc = presets[cfg->preset][0]; feed = presets[cfg->preset][1]; }
    else {
        fc = cfg->custom_fc; feed = cfg->custom_feed_db;
        if (fc < CROSSFEED_FREQ_MIN) fc = CROSSFEED_FREQ_MIN;
        if (fc > CROSSFEED_FREQ_MAX) fc = CROSSFEED_FREQ_MAX;
        if (feed < CROSSFEED_FEED_MIN) feed = CROSSFEED_FEED_MIN;
        if (feed > CROSSFEED_FEED_MAX) feed = CROSSFEED_FEED_MAX;
    }
    float ratio = powf(10.0f, feed / 20.0f);
    float G = 1.0f / (1.0f + ratio);
    float x = expf(-2.0f * PI_F * fc / fs);
    float a0 = G * (1.0f - x), b1 = x, ap;
    if (cfg->itd_enabled) {
        float lp_delay = x / ((1.0f - x) * fs);
        float rem = CROSSFEED_ITD_SEC - lp_delay;
        if (rem > 0.0f) ap = MAD(-rem, fs, 1.0f) / MAD(rem, fs, 1.0f);      /* D = rem*fs: (1 - D) / (1 + D), D never rounded when fused */
        else ap = 1.0f;
    } else ap = 1.0f;
#if PICO_RP2350
    st->lp_a0 = a0; st->lp_b1 = b1; st->ap_a = ap;
#else
    float scale = (float)(1LL << 28);
    st->lp_a0 = orc_f2i(a0 * scale); st->lp_b1 = orc_f2i(b1 * scale); st->ap_a = orc_f2i(ap * scale);
#endif
    st->lp_state_L = st->lp_state_R = 0; st->ap_state_L = st->ap_state_R = 0;
}

#if PICO_RP2350
void orc_crossfeed_process_stereo(CrossfeedState *s, float *left, float *right) { /* :132-156 */
    float il = *left, ir = *right;
    float lpl = MAD(s->lp_a0, il, s->lp_b1 * s->lp_state_L);
    float lpr = MAD(s->lp_a0, ir, s->lp_b1 * s->lp_state_R);
    s->lp_state_L = lpl; s->lp_state_R = lpr;
    float apl = MAD(s->ap_a, lpl, s->ap_state_L);
    s->ap_state_L = MAD(-s->ap_a, apl, lpl);
    float apr = MAD(s->ap_a, lpr, s->ap_state_R);
    s->ap_state_R = MAD(-s->ap_a, apr, lpr);
    *left = (il - lpl) + apr;
    *right = (ir - lpr) + apl;
}
#else
#define WADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define WSUB(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))
void orc_crossfeed_process_stereo(CrossfeedState *s, int32_t *left, int32_t *right) { /* :161-180 */
    int32_t il = *left, ir = *right;
    int32_t lpl = WADD(orc_fast_mul_q28(s->lp_a0, il), orc_fast_mul_q28(s->lp_b1, s->lp_state_L));
    int32_t lpr = WADD(orc_fast_mul_q28(s->lp_a0, ir), orc_fast_mul_q28(s->lp_b1, s->lp_state_R));
    s->lp_state_L = lpl; s->lp_state_R = lpr;
    int32_t apl = WADD(orc_fast_mul_q28(s->ap_a, lpl), s->ap_state_L);
    s->ap_state_L = WSUB(lpl, orc_fast_mul_q28(s->ap_a, apl));
    int32_t apr = WADD(orc_fast_mul_q28(s->ap_a, lpr), s->ap_state_R);
    s->ap_state_R = WSUB(lpr, orc_fast_mul_q28(s->ap_a, apr));
    *left = WADD(WSUB(il, lpl), apr);
    *right = WADD(WSUB(ir, lpr), apl);
}
#endif

/* ===================================================================================== */
/* Loudness table                                                                         */
/* ===================================================================================== */
static float iso226_spl(float Tf, float af, float Lu, float phon) { /* loudness.c:37-50 */
    float B = 0.4f * powf(10.0f, (Tf + Lu) / 10.0f - 9.0f);
    float thr = powf(B, af);
    float Af = MAD(4.47e-3f, powf(10.0f, 0.025f * phon) - 1.15f, thr);
    if (Af < 1e-10f) Af = 1e-10f;
    return MAD(10.0f / af, log10f(Af), -Lu) + 94.0f;
}

static float loud_comp_db(float Tf, float af, float Lu, float ref, float eff, float pct) { /* :54-78 */
    if (eff >= ref) return 0.0f;
    float s_ref = iso226_spl(Tf, af, Lu, ref);
    float s_eff = iso226_spl(Tf, af, Lu, eff);
    float flat = eff - ref;
    float fchg = s_eff - s_ref;
    float comp = fchg - flat;
    comp *= (pct / 100.0f);
    return comp;
}

static void loud_shelf(float freq, float Q, float gain_db, int high, float fs, LoudnessCoeffs *o) { /* :85-163 */
    if (fabsf(gain_db) < 0.01f) {
        o->bypass = true;
#if PICO_RP2350
        o->sva1 = o->sva2 = o->sva3 = 0.0f; o->svm0 = o->svm1 = o->svm2 = 0.0f;
#else
        o->b0 = 1 << FILTER_SHIFT; o->b1 = o->b2 = o->a1 = o->a2 = 0;
#endif
        return;
    }
    o->bypass = false;
    float A = powf(10.0f, gain_db / 40.0f);
#if PICO_RP2350
    float g = tanf(PI_F * freq / fs);
    float rA = sqrtf(A);
    if (high) g = g * rA; else g = g / rA;
    float k = 1.0f / Q;
    o->sva1 = 1.0f / MAD(g, g + k, 1.0f);
    o->sva2 = g * o->sva1;
    o->sva3 = g * o->sva2;
    if (high) { o->svm0 = A * A; o->svm1 = k * (1.0f - A) * A; o->svm2 = 1.0f - o->svm0; }      /* A*A shared: no fusion */
    else { o->svm0 = 1.0f; o->svm1 = k * (A - 1.0f); o->svm2 = MAD(A, A, -1.0f); }
#else
    float omega = 2.0f * PI_F * freq / fs;
    float sn = sinf(omega), cs = cosf(omega);
    float alpha = sn / (2.0f * Q);
    float rA = sqrtf(A);
    float a0, a1, a2, b0, b1, b2;
    if (high) {
        b0 = A * ((A + 1) + (A - 1) * cs + 2 * rA * alpha);
        b1 = -2 * A * ((A - 1) + (A + 1) * cs);
        b2 = A * ((A + 1) + (A - 1) * cs - 2 * rA * alpha);
        a0 = (A + 1) - (A - 1) * cs + 2 * rA * alpha;
        a1 = 2 * ((A - 1) - (A + 1) * cs);
        a2 = (A + 1) - (A - 1) * cs - 2 * rA * alpha;
    } else {
        b0 = A * ((A + 1) - (A - 1) * cs + 2 * rA * alpha);
        b1 = 2 * A * ((A - 1) - (A + 1) * cs);
        b2 = A * ((A + 1) - (A - 1) * cs - 2 * rA * alpha);
        a0 = (A + 1) + (A - 1) * cs + 2 * rA * alpha;
        a1 = -2 * ((A - 1) + (A + 1) * cs);
        a2 = (A + 1) + (A - 1) * cs - 2 * rA * alpha;
    }
    float scale = (float)(1LL << FILTER_SHIFT);
    o->b0 = orc_f2i((b0 / a0) * scale);
    o->b1 = orc_f2i((b1 / a0) * scale);
    o->b2 = orc_f2i((b2 / a0) * scale);
    o->a1 = orc_f2i((a1 / a0) * scale);
    o->a2 = orc_f2i((a2 / a0) * scale);
#endif
}

void orc_loudness_build_table(LoudnessCoeffs table[LOUDNESS_VOL_STEPS][LOUDNESS_BIQUAD_COUNT],
                              float ref_spl, float pct, float fs) { /* :169-217 */
    if (fs < 1.0f) fs = 48000.0f;
    if (ref_spl < 40.0f) ref_spl = 40.0f;
    if (ref_spl > 100.0f) ref_spl = 100.0f;
    for (int v = 0; v < LOUDNESS_VOL_STEPS; v++) {
        float vol_db = (float)(v - 60);
        float eff = ref_spl + vol_db;
        if (eff < 20.0f) eff = 20.0f;
        if (eff > ref_spl) eff = ref_spl;
        float lo = loud_comp_db(44.0f, 0.432f, 80.4f, ref_spl, eff, pct);   /* ISO 226 @ 50 Hz  */
        float hi = loud_comp_db(13.9f, 0.301f, 17.8f, ref_spl, eff, pct);   /* ISO 226 @ 10 kHz */
        loud_shelf(200.0f, 0.707f, lo, 0, fs, &table[v][0]);
        loud_shelf(6000.0f, 0.707f, hi, 1, fs, &table[v][1]);
    }
}
