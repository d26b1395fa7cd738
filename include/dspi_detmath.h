/*
 * dspi_detmath.h — deterministic log10f / powf for the leveller's per-block gain step.
 *
 * WHY THIS EXISTS.  The DSPi volume leveller calls log10f() once and powf() twice per
 * audio block *inside the audio path* (reference: firmware/DSPi/leveller.c:178,200,206 for
 * the RP2350 float flavour; :311,327,332 for the RP2040 Q28 flavour).  On the MCU these
 * resolve to newlib / pico-float (version unpinned, un-vendored submodule), on a Linux host
 * to glibc, on gfx950 to the ROCm device libm — three different last-bit behaviours.  To
 * make the CPU oracle and the HIP kernel agree bit-for-bit, both compile THIS header, which
 * uses nothing but IEEE-754 binary64 add/sub/mul/div and integer bit manipulation (all
 * correctly rounded on x86-64 SSE2 and on CDNA4), with contraction disabled on both sides
 * (-ffp-contract=off).  Results are the correctly rounded float in all but ~2^-29 of
 * cases; tests/test_detmath.py bounds the distance to glibc (<= 1 ulp).
 *
 * Header-only, C99 / C++ / HIP.  No libm calls.
 */
#ifndef DSPI_DETMATH_H
#define DSPI_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define DSPI_DM_FN static __host__ __device__ __forceinline__
#else
#define DSPI_DM_FN static inline
#endif

#ifdef __cplusplus
#define DSPI_DM_BITCAST(T, v) __builtin_bit_cast(T, v)
#else
/* C: union punning is defined behaviour in C99+ */
static inline uint64_t dspi_dm_d2u(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
static inline double   dspi_dm_u2d(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }
#endif

DSPI_DM_FN uint64_t dspi_dm_bits(double d) {
#ifdef __cplusplus
    return DSPI_DM_BITCAST(uint64_t, d);
#else
    return dspi_dm_d2u(d);
#endif
}
DSPI_DM_FN double dspi_dm_from_bits(uint64_t u) {
#ifdef __cplusplus
    return DSPI_DM_BITCAST(double, u);
#else
    return dspi_dm_u2d(u);
#endif
}

/* Natural log of a positive, finite, normal double (floats widened to double always are). */
DSPI_DM_FN double dspi_dm_log(double x) {
    uint64_t u = dspi_dm_bits(x);
    int e = (int)((u >> 52) & 0x7ffu) - 1023;
    double m = dspi_dm_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }                            /* (0.707,1.414] */
    double s = (m - 1.0) / (m + 1.0);   /* |s| <= 0.17158 */
    double z = s * s;                   /* <= 0.02944     */
    /* atanh series: log(m) = 2 s (1 + z/3 + z^2/5 + ... + z^11/23); tail < 2e-19 relative */
    double p = 0.043478260869565216;            /* 1/23 */
    p = p * z + 0.047619047619047616;           /* 1/21 */
    p = p * z + 0.052631578947368418;           /* 1/19 */
    p = p * z + 0.058823529411764705;           /* 1/17 */
    p = p * z + 0.066666666666666666;           /* 1/15 */
    p = p * z + 0.076923076923076927;           /* 1/13 */
    p = p * z + 0.090909090909090912;           /* 1/11 */
    p = p * z + 0.11111111111111110;            /* 1/9  */
    p = p * z + 0.14285714285714285;            /* 1/7  */
    p = p * z + 0.20000000000000001;            /* 1/5  */
    p = p * z + 0.33333333333333331;            /* 1/3  */
    p = p * z + 1.0;
    return (2.0 * s) * p + (double)e * 0.69314718055994529;
}

/* e^y for |y| < 690. */
DSPI_DM_FN double dspi_dm_exp(double y) {
    double kf = y * 1.4426950408889634;         /* y / ln 2 */
    int k = (int)(kf + (kf >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    /* two-part ln2 (Cody-Waite split; hi has 32 trailing zero bits so kd*hi is exact) */
    double r = (y - kd * 0.69314718036912382) - kd * 1.9082149292705877e-10;
    /* Taylor to r^14/14!, |r| <= 0.3466 -> tail < 1e-19 */
    double p = 1.1470745597729725e-11;          /* 1/14! */
    p = p * r + 1.6059043836821613e-10;         /* 1/13! */
    p = p * r + 2.08767569878681e-09;           /* 1/12! */
    p = p * r + 2.505210838544172e-08;          /* 1/11! */
    p = p * r + 2.7557319223985888e-07;         /* 1/10! */
    p = p * r + 2.7557319223985893e-06;         /* 1/9!  */
    p = p * r + 2.48015873015873e-05;           /* 1/8!  */
    p = p * r + 0.00019841269841269841;         /* 1/7!  */
    p = p * r + 0.0013888888888888889;          /* 1/6!  */
    p = p * r + 0.0083333333333333332;          /* 1/5!  */
    p = p * r + 0.041666666666666664;           /* 1/4!  */
    p = p * r + 0.16666666666666666;            /* 1/3!  */
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    if (k < -1000) k = -1000;
    if (k > 1000) k = 1000;
    return p * dspi_dm_from_bits((uint64_t)(k + 1023) << 52);
}

/* log10f replacement.  Domain: x > 0 (the leveller passes rms_sq + 1e-30f). */
DSPI_DM_FN float dspi_det_log10f(float x) {
    if (!(x > 0.0f)) return -300.0f;            /* out of contract; keep total */
    return (float)(dspi_dm_log((double)x) * 0.43429448190325182);
}

/* powf replacement.  Domain: a > 0 (alpha in (0,1) ^ block_len, and 10 ^ (dB/20)). */
DSPI_DM_FN float dspi_det_powf(float a, float b) {
    if (b == 0.0f) return 1.0f;
    if (!(a > 0.0f)) return 0.0f;               /* 0^b for b>0; negative bases out of contract */
    double y = (double)b * dspi_dm_log((double)a);
    if (y > 88.0) y = 88.0;                     /* keeps the result a finite float */
    if (y < -103.0) return 0.0f;
    return (float)dspi_dm_exp(y);
}

#endif /* DSPI_DETMATH_H */
