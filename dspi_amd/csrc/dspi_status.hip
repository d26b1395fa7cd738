// dspi_status.hip — status readback at scale: the sticky clip flags of every stream in one pass.
//
// Reference: global_status.clip_flags, a uint16 of sticky per-channel bits that REQ_GET_STATUS (wValue 9) reports in the last two
// bytes of its 26 / 18-byte block and REQ_CLEAR_CLIPS clears (firmware/DSPi/usb_audio.c:2427-2443, :2682; set at :945-951, :963-966,
// :1262-1268, :1279-1282).  The chain kernels keep the bits in four state slots per stream (the output waves of a workgroup share three
// of them and OR their bits in; dspi_capi.cpp fetch_status ORs the four when ONE stream is asked for).  dspi_out.clip_flags
// (include/dspi.h, DSPI_OUT_CLIP_FLAGS) is the same word for ALL streams: uint16 [stream], 8 bytes read and 2 written per stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dspi_kernels.h"
#include "../../include/dspi_detmath.h"

namespace dspi {

namespace {
// dspi_debug_detmath: the leveller's two libm replacements as the DEVICE computes them, over caller-supplied arguments (tests: device ==
// host build of the same header == binary128, including arguments constructed to take the double-double step)
__global__ __launch_bounds__(256) void detmath_kernel(int which, const float *a, const float *b, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (which) {      // 0 / 1: the two-step forms (the oracle's); 2 / 3 / 4: the device forms the chain kernels use
        case 0: r = dspi_det_log10f(a[i]); break;
        case 1: r = dspi_det_powf(a[i], b[i]); break;
        case 2: r = dspi_det_log10f_tab(a[i]); break;
        case 3: r = dspi_det_exp10f_tab(a[i]); break;
        default: r = dspi_det_powf_tab(a[i], b[i]); break;
    }
    out[i] = r;
}

__global__ __launch_bounds__(256) void clip_gather_kernel(const uint32_t *state, uint32_t n_streams, uint32_t row, uint32_t n_slots, uint32_t clip_slot, uint16_t *out) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_streams) return;
    const uint32_t wg = s / row, col = s % row;
    const uint32_t *p = state + ((size_t)wg * n_slots + clip_slot) * row + col;      // consecutive streams of a row: coalesced
    out[s] = (uint16_t)(p[0] | p[row] | p[2 * (size_t)row] | p[3 * (size_t)row]);
}
}  // namespace

hipError_t launch_clip_gather(const uint32_t *state, uint32_t n_streams, uint32_t row, uint32_t n_slots, uint32_t clip_slot, uint16_t *out, hipStream_t stream) {
    hipLaunchKernelGGL(clip_gather_kernel, dim3((n_streams + 255u) / 256u), dim3(256), 0, stream, state, n_streams, row, n_slots, clip_slot, out);
    return hipGetLastError();
}

hipError_t launch_detmath(int which, const float *a, const float *b, uint32_t n, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(detmath_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, which, a, b, n, out);
    return hipGetLastError();
}

}  // namespace dspi
