/* tools/gen_detmath_tables.c — the exception tables of include/dspi_detmath.h's device forms (include/dspi_detmath_tables.h).
 *
 * Walks EVERY binary32 argument of log10f (x > 0: 2^31 - 2^23 bit patterns) and of 10^y (all finite y: 2^32 - 2^24), evaluates step 1, and for
 * the arguments whose step-1 value is not proven evaluates the function exactly (the header's two-step form, cross-checked here against
 * binary128); arguments whose step-1 CANDIDATE differs from the exact value go into the table.  For a^count: the 18 bases of the leveller
 * (expf(-ln 10 / (fs t)), t from leveller.c:37-89's three speed presets, fs in 44.1 / 48 / 96 kHz), +-8 ulps around each, counts 1 .. 192.
 *     gcc -O2 -ffp-contract=off -DDSPI_DM_NO_TABLES -o /tmp/gen tools/gen_detmath_tables.c -lquadmath -lm -lpthread && /tmp/gen > include/dspi_detmath_tables.h
 * (~1 min on 8 cores).  tests/test_detmath.py::test_tables_regenerate runs it again and compares. */
#include <math.h>
#include <pthread.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/dspi_detmath.h"

#define NT 8
#define MAXE 4096
typedef struct { uint32_t lo, hi; int which; dspi_dm_exc e[MAXE]; int n; long amb; long bad_exact; } Job;

static void *walk(void *arg) {
    Job *j = (Job *)arg;
    for (uint64_t k = j->lo; k < j->hi; k++) {
        const float x = dspi_dm_ffrom((uint32_t)k);
        int amb = 0;
        float cand, exact;
        if (j->which == 0) { cand = dspi_det_log10f_try(x, &amb); if (!amb) continue; exact = dspi_det_log10f(x); }
        else { cand = dspi_det_exp10f_try(x, &amb); if (!amb) continue; exact = dspi_det_powf(10.0f, x); }
        j->amb++;
        const float q = j->which == 0 ? (float)log10q((__float128)x) : (float)powq((__float128)10.0, (__float128)x);
        if (dspi_dm_fbits(q) != dspi_dm_fbits(exact)) {
            /* (10^y: the header clamps at e^88 and flushes below e^-103; binary128 does neither — only inside those bounds is it the judge) */
            const double yd = (double)x * DSPI_DM_LOG_OF_10;
            if (j->which == 0 || (yd <= 88.0 && yd >= -103.0)) j->bad_exact++;
        }
        if (dspi_dm_fbits(cand) != dspi_dm_fbits(exact)) { if (j->n < MAXE) { j->e[j->n].in = (uint32_t)k; j->e[j->n].out = dspi_dm_fbits(exact); } j->n++; }
    }
    return NULL;
}

static int run(int which, uint32_t lo, uint32_t hi, dspi_dm_exc *out, int *n_out, long *amb_total) {
    static Job jobs[NT];
    pthread_t th[NT];
    const uint64_t span = (uint64_t)hi - lo;
    for (int t = 0; t < NT; t++) {
        jobs[t].lo = lo + (uint32_t)(span * t / NT); jobs[t].hi = lo + (uint32_t)(span * (t + 1) / NT); jobs[t].which = which; jobs[t].n = 0; jobs[t].amb = 0; jobs[t].bad_exact = 0;
        pthread_create(&th[t], NULL, walk, &jobs[t]);
    }
    int bad = 0;
    for (int t = 0; t < NT; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].n > MAXE) { fprintf(stderr, "table overflow\n"); exit(1); }
        memcpy(out + *n_out, jobs[t].e, sizeof(dspi_dm_exc) * jobs[t].n); *n_out += jobs[t].n; *amb_total += jobs[t].amb; bad += jobs[t].bad_exact != 0;
        if (jobs[t].bad_exact) fprintf(stderr, "which %d: %ld arguments where the two-step form differs from binary128\n", which, jobs[t].bad_exact);
    }
    return bad;
}

int main(void) {
    static dspi_dm_exc lg[NT * MAXE], ex[NT * MAXE];
    int n_lg = 0, n_ex = 0; long amb_lg = 0, amb_ex = 0;
    if (dspi_dm_log(10.0) != DSPI_DM_LOG_OF_10) { fprintf(stderr, "DSPI_DM_LOG_OF_10 is not dspi_dm_log(10.0) = %.17g\n", dspi_dm_log(10.0)); return 1; }
    int bad = run(0, 0x00000001u, 0x7f800000u, lg, &n_lg, &amb_lg);            /* every positive finite float, subnormals included */
    bad |= run(1, 0x00000001u, 0x7f800000u, ex, &n_ex, &amb_ex);                /* y > 0 */
    bad |= run(1, 0x80000001u, 0xff800000u, ex, &n_ex, &amb_ex);                /* y < 0 */
    if (bad) return 1;
    /* a^count */
    static dspi_dm_exc2 pw[1024]; int n_pw = 0; long amb_pw = 0, pairs = 0;
    const float presets[3][2] = {{0.100f, 2.000f}, {0.050f, 1.000f}, {0.020f, 0.500f}}, rates[3] = {44100.0f, 48000.0f, 96000.0f};
    for (int sp = 0; sp < 3; sp++) for (int w = 0; w < 2; w++) for (int r = 0; r < 3; r++) {
        const float a0 = expf(-logf(10.0f) / (rates[r] * presets[sp][w]));
        for (int d = -8; d <= 8; d++) {
            const float a = dspi_dm_ffrom(dspi_dm_fbits(a0) + (uint32_t)d);
            for (int c = 1; c <= 192; c++) {
                int amb = 0; const float b = (float)c, cand = dspi_det_powf_try(a, b, &amb);
                pairs++;
                if (!amb) continue;
                amb_pw++;
                const float exact = dspi_det_powf(a, b), q = (float)powq((__float128)a, (__float128)b);
                if (dspi_dm_fbits(q) != dspi_dm_fbits(exact)) { fprintf(stderr, "pow: two-step form differs from binary128 at %a ^ %d\n", a, c); return 1; }
                if (dspi_dm_fbits(cand) != dspi_dm_fbits(exact)) { pw[n_pw].a = dspi_dm_fbits(a); pw[n_pw].b = dspi_dm_fbits(b); pw[n_pw].out = dspi_dm_fbits(exact); n_pw++; }
            }
        }
    }
    printf("/* include/dspi_detmath_tables.h — GENERATED by tools/gen_detmath_tables.c (do not edit): arguments whose step-1 candidate is the wrong\n"
           " * neighbour of the exact value.  log10f: all %u positive floats walked, %ld not proven by step 1, %d listed; 10^y: all %u finite non-zero\n"
           " * floats walked, %ld not proven, %d listed; a^count: %ld (base, count) pairs walked, %ld not proven, %d listed. */\n",
           0x7f800000u - 1u, amb_lg, n_lg, 2u * (0x7f800000u - 1u), amb_ex, n_ex, pairs, amb_pw, n_pw);
    printf("#define DSPI_DM_LOG10_EXC_N %d\n#define DSPI_DM_LOG10_EXC {", n_lg);
    for (int i = 0; i < n_lg; i++) printf(" {0x%08xu, 0x%08xu},", lg[i].in, lg[i].out);
    printf(" {0u, 0u} }\n#define DSPI_DM_EXP10_EXC_N %d\n#define DSPI_DM_EXP10_EXC {", n_ex);
    for (int i = 0; i < n_ex; i++) printf(" {0x%08xu, 0x%08xu},", ex[i].in, ex[i].out);
    printf(" {0u, 0u} }\n#define DSPI_DM_POW_EXC_N %d\n#define DSPI_DM_POW_EXC {", n_pw);
    for (int i = 0; i < n_pw; i++) printf(" {0x%08xu, 0x%08xu, 0x%08xu},", pw[i].a, pw[i].b, pw[i].out);
    printf(" {0u, 0u, 0u} }\n");
    /* the same lists as compare chains (what the device forms use: immediates in a cold block, no table in memory, no loop registers) */
    printf("#define DSPI_DM_LOG10_EXC_FIX(k, f) do {");
    for (int i = 0; i < n_lg; i++) printf(" %sif ((k) == 0x%08xu) (f) = dspi_dm_ffrom(0x%08xu);", i ? "else " : "", lg[i].in, lg[i].out);
    printf(" } while (0)\n#define DSPI_DM_EXP10_EXC_FIX(k, f) do {");
    for (int i = 0; i < n_ex; i++) printf(" %sif ((k) == 0x%08xu) (f) = dspi_dm_ffrom(0x%08xu);", i ? "else " : "", ex[i].in, ex[i].out);
    printf(" } while (0)\n#define DSPI_DM_POW_EXC_FIX(ka, kb, f) do {");
    for (int i = 0; i < n_pw; i++) printf(" %sif ((ka) == 0x%08xu && (kb) == 0x%08xu) (f) = dspi_dm_ffrom(0x%08xu);", i ? "else " : "", pw[i].a, pw[i].b, pw[i].out);
    printf(" (void)(ka); (void)(kb); } while (0)\n");
    return 0;
}
