/* orc_common.h — helpers shared by the standalone and _ref oracle builds.
 * TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H
#include <stdint.h>
#include <limits.h>

/* float -> int32 conversion semantics.
 *
 * The firmware relies on `(int32_t)some_float` in many places (Q28 coefficients
 * dsp_pipeline.c:169-173, preamp usb_audio.c:248, output gain :1129, leveller gain
 * leveller.c:334 and limiter :376, PDM scaling usb_audio.c:953).  On both target MCUs this
 * SATURATES (Cortex-M33 `vcvt.s32.f32`; RP2040 `__aeabi_f2iz` -> bootrom float2int_z, which
 * clamps) and maps NaN to 0.  An x86-64 build of the same C yields INT_MIN for every
 * out-of-range input (cvttss2si).  The oracle implements the firmware behaviour; setting
 * orc_x86_cast_semantics = 1 reproduces the x86 build so that the restatement can be
 * cross-checked bit-for-bit against `oracle/_ref` (reference sources compiled here) even on
 * inputs that overflow.  gfx950 `v_cvt_i32_f32` saturates like ARM (tools/probe, 0 mismatches).
 */
extern int orc_x86_cast_semantics;
static inline int32_t orc_f2i(float f) {
    if (orc_x86_cast_semantics)
        return (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? INT32_MIN : (int32_t)f;
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
#endif
