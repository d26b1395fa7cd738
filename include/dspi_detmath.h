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
 * (-ffp-contract=off).
 *
 * WHAT THEY RETURN (round 6): the CORRECTLY ROUNDED binary32 value (round to nearest, ties to
 * even) of log10(x) and of a^b — a definition anyone can reproduce with any arbitrary-precision
 * library, not "what this header happens to compute".  Two steps (Ziv's strategy):
 *   1. a binary64 evaluation r with a proven-and-measured relative error bound eps (2^-46 for
 *      log10; 2^-46 + |b ln a| 2^-47 for pow: tests/test_detmath.py measures <= 2^-50 against
 *      binary128 and asserts the margin); if r(1 - eps) and r(1 + eps) round to the same float,
 *      that float is the answer (all but ~1 in 10^6 calls);
 *   2. otherwise a double-double evaluation (Dekker / Knuth error-free transformations built from
 *      binary64 add / mul / div only, ~2^-95 relative) and a rounding of its two words that
 *      breaks an apparent tie by the sign of the low word — unless the value is within 2^-85 of
 *      the midpoint, which for these functions happens only when a^b IS the midpoint (e.g.
 *      4097^2 = 2^24 + 2^13 + 1): then ties-to-even, as IEEE-754 says.
 * tests/test_detmath.py: 0 mismatches against binary128 (libquadmath) over > 10^7 arguments per
 * function on the leveller's ranges, their edges, and arguments constructed to hit step 2.
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

/* ------------------------------------------------------------------------------------------------------------------
 * Step 2: double-double arithmetic (value = h + l, |l| <= ulp(h) / 2).  Error-free transformations: Knuth's two-sum, Dekker's
 * fast-two-sum and Veltkamp-split two-product (no fused multiply-add: the same operations on x86-64 and on gfx950).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct { double h, l; } dspi_dd;

DSPI_DM_FN dspi_dd dspi_dd_make(double h, double l) { dspi_dd r; r.h = h; r.l = l; return r; }
DSPI_DM_FN dspi_dd dspi_dd_fast2sum(double a, double b) { const double s = a + b; return dspi_dd_make(s, b - (s - a)); }      /* |a| >= |b| */
DSPI_DM_FN dspi_dd dspi_dd_2sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return dspi_dd_make(s, (a - (s - bb)) + (b - bb));
}
DSPI_DM_FN dspi_dd dspi_dd_2prod(double a, double b) {
    const double p = a * b;
    double t = 134217729.0 * a;                 /* 2^27 + 1 */
    const double ah = t - (t - a), al = a - ah;
    t = 134217729.0 * b;
    const double bh = t - (t - b), bl = b - bh;
    return dspi_dd_make(p, ((ah * bh - p) + ah * bl + al * bh) + al * bl);
}
DSPI_DM_FN dspi_dd dspi_dd_add(dspi_dd x, dspi_dd y) {
    dspi_dd s = dspi_dd_2sum(x.h, y.h);
    const dspi_dd t = dspi_dd_2sum(x.l, y.l);
    s = dspi_dd_fast2sum(s.h, s.l + t.h);
    return dspi_dd_fast2sum(s.h, s.l + t.l);
}
DSPI_DM_FN dspi_dd dspi_dd_neg(dspi_dd x) { return dspi_dd_make(-x.h, -x.l); }
DSPI_DM_FN dspi_dd dspi_dd_mul(dspi_dd x, dspi_dd y) {
    const dspi_dd p = dspi_dd_2prod(x.h, y.h);
    return dspi_dd_fast2sum(p.h, p.l + (x.h * y.l + x.l * y.h));
}
DSPI_DM_FN dspi_dd dspi_dd_mul_d(dspi_dd x, double d) {
    const dspi_dd p = dspi_dd_2prod(x.h, d);
    return dspi_dd_fast2sum(p.h, p.l + x.l * d);
}
DSPI_DM_FN dspi_dd dspi_dd_div(dspi_dd x, dspi_dd y) {       /* three quotient digits */
    const double q1 = x.h / y.h;
    dspi_dd r = dspi_dd_add(x, dspi_dd_neg(dspi_dd_mul_d(y, q1)));
    const double q2 = r.h / y.h;
    r = dspi_dd_add(r, dspi_dd_neg(dspi_dd_mul_d(y, q2)));
    const double q3 = r.h / y.h;
    return dspi_dd_add(dspi_dd_fast2sum(q1, q2), dspi_dd_make(q3, 0.0));
}

/* generated by tools/gen_detmath_consts.py: double-double constants, hi = nearest double, lo = nearest double of the remainder */
#define DSPI_DD_LN2_H 0.6931471805599453
#define DSPI_DD_LN2_L 2.3190468138462996e-17
#define DSPI_DD_LOG10E_H 0.4342944819032518
#define DSPI_DD_LOG10E_L 1.098319650216765e-17
#define DSPI_DD_LOG_TERMS 21
/* 1/(2k+1), k = 0 .. 20 */
#define DSPI_DD_ATANH_COEFFS { {1.0, 0.0}, {0.3333333333333333, 1.850371707708594e-17}, {0.2, -1.1102230246251566e-17}, {0.14285714285714285, 7.93016446160826e-18}, {0.1111111111111111, 6.1679056923619804e-18}, {0.09090909090909091, -2.523234146875356e-18}, {0.07692307692307693, -4.270088556250602e-18}, {0.06666666666666667, 9.251858538542971e-19}, {0.058823529411764705, 8.163404592832033e-19}, {0.05263157894736842, 2.921639538487254e-18}, {0.047619047619047616, 2.64338815386942e-18}, {0.043478260869565216, 1.206764157201257e-18}, {0.04, -8.326672684688674e-19}, {0.037037037037037035, 2.05596856412066e-18}, {0.034482758620689655, 4.785444071660157e-19}, {0.03225806451612903, 8.953411488912552e-19}, {0.030303030303030304, -8.410780489584519e-19}, {0.02857142857142857, 8.921435019309293e-19}, {0.02702702702702703, -1.50030138462859e-18}, {0.02564102564102564, 8.896017825522087e-19}, {0.024390243902439025, -8.46206573647223e-19} }
#define DSPI_DD_EXP_TERMS 23
#define DSPI_DD_INVFACT_COEFFS { {1.0, 0.0}, {1.0, 0.0}, {0.5, 0.0}, {0.16666666666666666, 9.25185853854297e-18}, {0.041666666666666664, 2.3129646346357427e-18}, {0.008333333333333333, 1.1564823173178714e-19}, {0.001388888888888889, -5.300543954373577e-20}, {0.0001984126984126984, 1.7209558293420705e-22}, {2.48015873015873e-05, 2.1511947866775882e-23}, {2.7557319223985893e-06, -1.858393274046472e-22}, {2.755731922398589e-07, 2.3767714622250297e-23}, {2.505210838544172e-08, -1.448814070935912e-24}, {2.08767569878681e-09, -1.20734505911326e-25}, {1.6059043836821613e-10, 1.2585294588752098e-26}, {1.1470745597729725e-11, 2.0655512752830745e-28}, {7.647163731819816e-13, 7.03872877733453e-30}, {4.779477332387385e-14, 4.399205485834081e-31}, {2.8114572543455206e-15, 1.6508842730861433e-31}, {1.5619206968586225e-16, 1.1910679660273754e-32}, {8.22063524662433e-18, 2.2141894119604265e-34}, {4.110317623312165e-19, 1.4412973378659527e-36}, {1.9572941063391263e-20, -1.3643503830087908e-36}, {8.896791392450574e-22, -7.911402614872376e-38} }

/* ln(x), x a positive normal double whose significand fits 26 bits (a widened float does): ~2^-100 relative */
DSPI_DM_FN dspi_dd dspi_dd_log(double x) {
    static const dspi_dd c[DSPI_DD_LOG_TERMS] = DSPI_DD_ATANH_COEFFS;
    const uint64_t u = dspi_dm_bits(x);
    int e = (int)((u >> 52) & 0x7ffu) - 1023;
    double m = dspi_dm_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    /* m - 1 and m + 1 are exact (<= 26 significant bits each) */
    const dspi_dd s = dspi_dd_div(dspi_dd_make(m - 1.0, 0.0), dspi_dd_make(m + 1.0, 0.0));
    const dspi_dd z = dspi_dd_mul(s, s);
    dspi_dd p = c[DSPI_DD_LOG_TERMS - 1];
    for (int k = DSPI_DD_LOG_TERMS - 2; k >= 0; --k) p = dspi_dd_add(dspi_dd_mul(p, z), c[k]);
    p = dspi_dd_mul(p, s);
    p = dspi_dd_make(2.0 * p.h, 2.0 * p.l);
    return dspi_dd_add(dspi_dd_mul_d(dspi_dd_make(DSPI_DD_LN2_H, DSPI_DD_LN2_L), (double)e), p);
}

/* e^y for a double-double |y| < 700, as double-double: ~2^-100 relative */
DSPI_DM_FN dspi_dd dspi_dd_exp(dspi_dd y) {
    static const dspi_dd c[DSPI_DD_EXP_TERMS] = DSPI_DD_INVFACT_COEFFS;
    const double kf = y.h * 1.4426950408889634;
    int k = (int)(kf + (kf >= 0.0 ? 0.5 : -0.5));
    const dspi_dd r = dspi_dd_add(y, dspi_dd_neg(dspi_dd_mul_d(dspi_dd_make(DSPI_DD_LN2_H, DSPI_DD_LN2_L), (double)k)));
    dspi_dd p = c[DSPI_DD_EXP_TERMS - 1];
    for (int n = DSPI_DD_EXP_TERMS - 2; n >= 0; --n) p = dspi_dd_add(dspi_dd_mul(p, r), c[n]);
    if (k < -1000) k = -1000;
    if (k > 1000) k = 1000;
    const double sc = dspi_dm_from_bits((uint64_t)(k + 1023) << 52);      /* a power of two: both words scale exactly */
    return dspi_dd_make(p.h * sc, p.l * sc);
}

/* Round h + l (|l| <= ulp(h) / 2) to binary32, nearest-even, as if h + l were the exact value — except that a value within 2^-85 (relative) of a
 * float midpoint is taken to BE the midpoint (see the header comment). */
DSPI_DM_FN float dspi_dd_to_float(dspi_dd v) {
    if (v.h < 0.0) return -dspi_dd_to_float(dspi_dd_neg(v));
    const float f1 = (float)v.h;                /* v.h's own rounding */
    const double d1 = (double)f1;
    if (d1 == v.h || v.l == 0.0) return f1;     /* v.h is a float (l cannot carry it to a midpoint: |l| <= 2^-53 h, a float's half-ulp is >= 2^-25 h)
                                                 * ... or h + l = h: the conversion above already is the answer */
    /* the float on the other side of v.h */
#ifdef __cplusplus
    const uint32_t b1 = DSPI_DM_BITCAST(uint32_t, f1);
#else
    union { float f; uint32_t u; } c1; c1.f = f1; const uint32_t b1 = c1.u;
#endif
    const uint32_t b2 = (v.h > d1) ? b1 + 1u : b1 - 1u;
#ifdef __cplusplus
    const float f2 = DSPI_DM_BITCAST(float, b2);
#else
    union { float f; uint32_t u; } c2; c2.u = b2; const float f2 = c2.f;
#endif
    const double d2 = (double)f2;
    if ((b2 & 0x7f800000u) == 0x7f800000u) return f1;      /* (overflow edge: the callers clamp before it) */
    const double mid = 0.5 * d1 + 0.5 * d2;      /* exact: adjacent floats */
    const double off = (v.h - mid) + v.l;        /* v.h - mid is exact (both within one float ulp, multiples of a double ulp) */
    const double tol = mid * 2.5849394142282115e-26;      /* 2^-85 */
    if (off > tol) return d1 > d2 ? f1 : f2;     /* above the midpoint: the larger neighbour */
    if (off < -tol) return d1 > d2 ? f2 : f1;
    return (float)mid;                           /* the midpoint itself: ties to even (the conversion's own rule) */
}

/* Do all of [r (1 - eps), r (1 + eps)] round to the same float, i.e. does the interval hold no midpoint of two adjacent floats?
 * Where the float is normal: a midpoint is a double whose low 29 significand bits are 1 0000...0, and eps |r| is at most eps 2^53 ulps of r,
 * so three integer operations on the low word decide (the interval cannot reach a midpoint of a neighbouring binade: the nearest lies 2^27
 * ulps beyond the binade's edge).  `ulps` = ceil(eps 2^53) + 2, at most 2^27.  Otherwise (float subnormal, zero, overflow) by conversion. */
DSPI_DM_FN int dspi_dm_unambiguous_u(double r, double eps, uint32_t ulps, float *out) {
    const uint64_t u = dspi_dm_bits(r);
    const uint32_t e = (uint32_t)(u >> 52) & 0x7ffu;
    if (e >= 1023u - 126u && e <= 1023u + 126u) {
        *out = (float)r;
        return (((uint32_t)u & 0x1fffffffu) - (0x10000000u - ulps)) > 2u * ulps;
    }
    const double d = (r < 0.0 ? -r : r) * eps;
    const float lo = (float)(r - d), hi = (float)(r + d);
    *out = lo;
    return lo == hi;
}
DSPI_DM_FN int dspi_dm_unambiguous(double r, double eps, float *out) {
    double w = eps * 9007199254740992.0;        /* 2^53 */
    if (w > 134217000.0) w = 134217000.0;
    return dspi_dm_unambiguous_u(r, eps, (uint32_t)w + 3u, out);
}

/* Step 2 as functions of their own (on the device not inlined: only the test kernel dspi_debug_detmath evaluates them there; the chain kernels
 * use the table forms at the end of this header). */
#if defined(__HIP_DEVICE_COMPILE__)
#define DSPI_DM_SLOW static __device__ __attribute__((noinline))
#elif defined(__HIPCC__)
#define DSPI_DM_SLOW static __host__ __device__ __attribute__((noinline))
#else
#define DSPI_DM_SLOW static
#endif
DSPI_DM_SLOW float dspi_det_log10f_slow(float x) {
    return dspi_dd_to_float(dspi_dd_mul(dspi_dd_log((double)x), dspi_dd_make(DSPI_DD_LOG10E_H, DSPI_DD_LOG10E_L)));
}
DSPI_DM_SLOW float dspi_det_powf_slow(float a, float b) {
    dspi_dd yy = dspi_dd_mul_d(dspi_dd_log((double)a), (double)b);
    if (yy.h > 88.0) yy = dspi_dd_make(88.0, 0.0);
    return dspi_dd_to_float(dspi_dd_exp(yy));
}

/* Step 1 alone: the candidate, and *amb |= 1 when it is not proven (the caller then owes the value to step 2, or to the exception tables). */
DSPI_DM_FN float dspi_det_log10f_try(float x, int *amb) {
    if (!(x > 0.0f)) return -300.0f;            /* out of contract; keep total */
    if (x == 1.0f) return 0.0f;
    float f;
    if (!dspi_dm_unambiguous_u(dspi_dm_log((double)x) * 0.43429448190325182, 1.4210854715202004e-14 /* 2^-46 */, 130u /* 2^7 + 2 */, &f)) *amb |= 1;
    return f;
}
DSPI_DM_FN float dspi_det_powf_try(float a, float b, int *amb) {
    if (b == 0.0f) return 1.0f;
    if (!(a > 0.0f)) return 0.0f;               /* 0^b for b>0; negative bases out of contract */
    if (a == 1.0f) return 1.0f;
    double y = (double)b * dspi_dm_log((double)a);
    if (y > 88.0) y = 88.0;                     /* keeps the result a finite float */
    if (y < -103.0) return 0.0f;
    float f;
    const double ay = y < 0.0 ? -y : y;
    if (!dspi_dm_unambiguous(dspi_dm_exp(y), 1.4210854715202004e-14 + ay * 7.105427357601002e-15 /* 2^-46 + |y| 2^-47 */, &f)) *amb |= 1;
    return f;
}

/* log10f replacement: the correctly rounded binary32 log10(x).  Domain: x > 0 (the leveller passes rms_sq + 1e-30f). */
DSPI_DM_FN float dspi_det_log10f(float x) {
    int amb = 0;
    const float f = dspi_det_log10f_try(x, &amb);
    return amb ? dspi_det_log10f_slow(x) : f;
}

/* powf replacement: the correctly rounded binary32 a^b.  Domain: a > 0 (alpha in (0,1) ^ block_len, and 10 ^ (dB/20)); results clamp at
 * e^88 (finite) and flush to 0 below e^-103, as before. */
DSPI_DM_FN float dspi_det_powf(float a, float b) {
    int amb = 0;
    const float f = dspi_det_powf_try(a, b, &amb);
    return amb ? dspi_det_powf_slow(a, b) : f;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The DEVICE's forms (round 6): step 1 + a table of exceptions instead of step 2.
 *
 * Why: step 2 in a kernel that uses every register of its occupancy class is not free even when it never runs — its scalar and vector
 * registers leak into the hot loops' allocation (measured on the headline kernel: +2.0 .. +2.6 % with step 2 inlined or called, 0 % with
 * step 1 alone; profiles/r06_detmath.md).  log10f has 2^31 arguments and 10^y 2^32: tools/gen_detmath_tables.c walks ALL of them, takes
 * those whose step-1 value is not proven (~10^3), evaluates them exactly, and keeps the few whose step-1 candidate is the wrong neighbour.
 * So for EVERY binary32 argument: proven -> the candidate; not proven -> the table's value if the argument is listed, else the candidate —
 * the correctly rounded result either way, the same float the two-step functions above return (tests/test_detmath.py re-walks both ranges).
 * a^b in general has no such table; the leveller's a^count has 18 bases (three speeds x attack / release x three rates, leveller.c:37-89)
 * and counts 1 .. 192: the generator walks those, +-8 ulps around every base; a context whose actual (alpha, count) falls outside is
 * refused by the host, which compares the two forms when it builds the parameter image (dspi_params.cpp).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct { uint32_t in, out; } dspi_dm_exc;
typedef struct { uint32_t a, b, out; } dspi_dm_exc2;
#ifndef DSPI_DM_NO_TABLES
#include "dspi_detmath_tables.h"
#endif
/* what dspi_dm_log(10.0) returns, to the bit (tests/test_detmath.py): 10^y below is then the same binary64 value as step 1 of powf(10, y) */
#define DSPI_DM_LOG_OF_10 2.3025850929940455

DSPI_DM_FN uint32_t dspi_dm_fbits(float f) {
#ifdef __cplusplus
    return DSPI_DM_BITCAST(uint32_t, f);
#else
    union { float f; uint32_t u; } c; c.f = f; return c.u;
#endif
}
DSPI_DM_FN float dspi_dm_ffrom(uint32_t u) {
#ifdef __cplusplus
    return DSPI_DM_BITCAST(float, u);
#else
    union { float f; uint32_t u; } c; c.u = u; return c.f;
#endif
}
/* step 1 of 10^y alone (the value powf(10, y)'s step 1 computes, without re-deriving log 10) */
DSPI_DM_FN float dspi_det_exp10f_try(float y, int *amb) {
    if (y == 0.0f) return 1.0f;
    double yd = (double)y * DSPI_DM_LOG_OF_10;
    if (yd > 88.0) yd = 88.0;
    if (yd < -103.0) return 0.0f;
    float f;
    const double ay = yd < 0.0 ? -yd : yd;
    if (!dspi_dm_unambiguous(dspi_dm_exp(yd), 1.4210854715202004e-14 + ay * 7.105427357601002e-15, &f)) *amb |= 1;
    return f;
}
#ifndef DSPI_DM_NO_TABLES
DSPI_DM_FN float dspi_det_log10f_tab(float x) {
    int amb = 0;
    float f = dspi_det_log10f_try(x, &amb);
    if (amb) { const uint32_t k = dspi_dm_fbits(x); DSPI_DM_LOG10_EXC_FIX(k, f); }
    return f;
}
DSPI_DM_FN float dspi_det_exp10f_tab(float y) {
    int amb = 0;
    float f = dspi_det_exp10f_try(y, &amb);
    if (amb) { const uint32_t k = dspi_dm_fbits(y); DSPI_DM_EXP10_EXC_FIX(k, f); }
    return f;
}
/* the leveller's alpha^count (see above): correct on the walked (base, count) set; the host checks the actual pair against dspi_det_powf */
DSPI_DM_FN float dspi_det_powf_tab(float a, float b) {
    int amb = 0;
    float f = dspi_det_powf_try(a, b, &amb);
    if (amb) { const uint32_t ka = dspi_dm_fbits(a), kb = dspi_dm_fbits(b); DSPI_DM_POW_EXC_FIX(ka, kb, f); }
    return f;
}
#endif

#endif /* DSPI_DETMATH_H */
