/* ref_math_hook.h — force-included (-include) when compiling the reference's leveller.c for the
 * `_ref` oracle: routes the two per-block libm calls in the audio path (leveller.c:178,200,206 /
 * :311,327,332) through hooks so tests can select glibc or dspi_detmath.h.  The reference
 * source file itself is compiled unmodified, in place.  TEST INFRASTRUCTURE ONLY. */
#include <math.h>
float orc_hook_log10f(float x);
float orc_hook_powf(float a, float b);
#define log10f orc_hook_log10f
#define powf orc_hook_powf
