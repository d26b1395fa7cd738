// dspi_kernels.hip — the DSPi per-sample chain as fused, persistent gfx950 kernels.
//
// Reference path: process_audio_packet, firmware/DSPi/usb_audio.c:560-967 (RP2350 float) and
// :968-1283 (RP2040 Q28), with its leaf loops in dsp_pipeline.c:281-365, dsp_process_rp2040.S:225-394,
// leveller.c:148-389, crossfeed.c:132-180.
//
// Two kernels share the data layout:
//   * chain_kernel_pk (dspi_chain_pk.inc, included below): float flavour, TWO streams per lane on packed FP32,
//     128 streams x 12 waves per workgroup — the bench path.
//   * chain_kernel (this file): ONE stream per lane, 64 streams x 4 waves — the Q28 flavour (integer arithmetic
//     has no packed form) and float lanes whose two streams carry different parameter images.
//       wave 0    = "master": input convert + preamp, loudness, master L/R EQ, leveller, master peaks,
//                   crossfeed -> post-crossfeed L/R chunk into LDS
//       waves 1-3 = "outputs": matrix mix, per-output EQ, gain, delay line, peaks, int24 / Q28 words
// Common to both: the time loop runs inside the kernel (16-frame chunks, packet semantics kept per block); filter
// state stays in LDS for the whole launch, coefficients come from one DevImage through scalar loads (all lanes of a
// launch share the image), delay lines / leveller ring are [position][stream] rows in HBM so every access is one
// coalesced row.  No MFMA: every stage is a per-stream recurrence.  No contraction, FTZ on (build flags).
//
// The leveller is a two-pass-per-packet algorithm (envelope over the whole packet, then a gain ramp over the same
// packet).  Pass 1 of packet k and pass 2 of packet k-1 are interleaved chunk by chunk through a [2][1024][stream]
// ring in HBM that doubles as the 480-sample lookahead line, so no wave ever holds more than one chunk in registers.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dspi_detmath.h"
#include "dspi_image.h"
#include "dspi_kernels.h"
#include "dspi_spdif_dev.h"

namespace dspi {

namespace {

constexpr int T = kChunk;   // frames per chunk

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U4a { u32x4 v; };
__device__ __forceinline__ u32x4 ld4(const void *p) { return reinterpret_cast<const U4a *>(p)->v; }
__device__ __forceinline__ void st4(void *p, u32x4 v) { reinterpret_cast<U4a *>(p)->v = v; }

// The image is never written while a launch is in flight; reading it through the constant address
// space lets the compiler use scalar loads (s_load_dwordx8 per band) instead of per-lane VMEM loads.
typedef const __attribute__((address_space(4))) DevImage *ImgPtr;
typedef const __attribute__((address_space(4))) DevBand *BandPtr;
__device__ __forceinline__ ImgPtr to_const(const DevImage *p) { return (ImgPtr)(uintptr_t)p; }

// (int32_t)float the way both MCUs do it (Cortex-M33 vcvt.s32.f32, RP2040 bootrom float2int_z): truncating, SATURATING,
// NaN -> 0.  v_cvt_i32_f32 has exactly these semantics; going through asm keeps C's out-of-range UB out of the picture.
__device__ __forceinline__ int32_t f2i_sat(float f) {
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

#include "dspi_bandloops.inc"
#include "dspi_bandloops_fma.inc"

// a*b + c under the context's float contract (include/dspi.h DSPI_FLOAT_CONTRACT_FMA): FMA = one fused rounding, as GCC
// contracts the firmware; otherwise the two roundings of the source read literally (the file is built -ffp-contract=off)
template <bool FMA> __device__ __forceinline__ float mad1(float a, float b, float c) {
    if (FMA) return __builtin_fmaf(a, b, c);
    return a * b + c;
}

// ------------------------------------------------------------------------------------------
// One band over one chunk, specialised on the output form so that the sample loop is branch-free
// straight-line code (the reference specialises the same way: dsp_pipeline.c:298-343).
template <bool TAIL, uint32_t KIND, bool FMA = false>
__device__ __forceinline__ void band_loop_f32(float (&x)[T], int n, float &s1, float &s2, float c0, float c1, float c2, float c3, float c4, float c5) {
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (TAIL && i >= n) break;
        const float in = x[i];
        if (KIND == K_BIQUAD) {           // dsp_pipeline.c:347-362
            float y = mad1<FMA>(c0, in, s1);
            s1 = mad1<FMA>(c1, in, -(c3 * y)) + s2;
            s2 = mad1<FMA>(c2, in, -(c4 * y));
            x[i] = y;
        } else {                          // Cytomic SVF core (s1 = ic1eq, s2 = ic2eq)
            float v3 = in - s2;
            float v1 = mad1<FMA>(c0, s1, c1 * v3);
            float v2 = mad1<FMA>(c2, v3, mad1<FMA>(c1, s1, s2));
            s1 = mad1<FMA>(2.0f, v1, -s1);
            s2 = mad1<FMA>(2.0f, v2, -s2);
            if (KIND == K_SVF_LP) x[i] = v2;
            else if (KIND == K_SVF_HP) x[i] = mad1<FMA>(c3, v1, in) - v2;
            else if (KIND == K_SVF_PK) x[i] = mad1<FMA>(c3, v1, in);
            else x[i] = mad1<FMA>(c5, v2, mad1<FMA>(c3, in, c4 * v1));
        }
    }
}

// float EQ band runner: the reference's block loops with the state pair in LDS.  Software-pipelined by
// hand: the coefficients (scalar loads) and state pair (LDS) of band b+1 are requested before band b's
// sample loop starts, so their latency hides under ~200 VALU instructions instead of stalling every band.
template <bool TAIL, int NB, bool SHELF_ONLY = false>
__device__ __forceinline__ void run_bands_f32(float (&x)[T], int n, BandPtr bands, float *__restrict__ st) {
    uint32_t kind = bands[0].kind;
    float c0 = bands[0].c[0].f, c1 = bands[0].c[1].f, c2 = bands[0].c[2].f, c3 = bands[0].c[3].f, c4 = bands[0].c[4].f, c5 = bands[0].c[5].f;
    float s1 = st[0], s2 = st[kLanes];
    // Make band 0's operands "used" before the loop: otherwise the wait for them is placed inside the loop
    // body, after the prefetch of band b+1 has been issued, and (SMEM returning out of order) drains that too.
    asm volatile("" ::"s"(kind), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "v"(s1), "v"(s2));
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
        uint32_t nkind = K_BYPASS;
        float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f, n4 = 0.f, n5 = 0.f, ns1 = 0.f, ns2 = 0.f;
        if (b + 1 < NB) {
            BandPtr nb = bands + b + 1;
            nkind = nb->kind;
            n0 = nb->c[0].f; n1 = nb->c[1].f; n2 = nb->c[2].f; n3 = nb->c[3].f; n4 = nb->c[4].f; n5 = nb->c[5].f;
            ns1 = st[(b + 1) * 2 * kLanes];
            ns2 = st[(b + 1) * 2 * kLanes + kLanes];
        }
        if (kind != K_BYPASS) {
            if (SHELF_ONLY) band_loop_f32<TAIL, K_SVF_SHELF>(x, n, s1, s2, c0, c1, c2, c3, c4, c5);
            else switch (kind) {
                case K_BIQUAD: band_loop_f32<TAIL, K_BIQUAD>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_LP: band_loop_f32<TAIL, K_SVF_LP>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_HP: band_loop_f32<TAIL, K_SVF_HP>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_PK: band_loop_f32<TAIL, K_SVF_PK>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                default: band_loop_f32<TAIL, K_SVF_SHELF>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
            }
            st[b * 2 * kLanes] = s1;
            st[b * 2 * kLanes + kLanes] = s2;
        }
        kind = nkind; c0 = n0; c1 = n1; c2 = n2; c3 = n3; c4 = n4; c5 = n5; s1 = ns1; s2 = ns2;
    }
}

// Full-chunk runner on the hand-scheduled loops of dspi_bandloops.inc.  The kind dispatch happens inside the asm
// statement, so from the compiler's point of view every band is one in-place update of x[0..15]: no copies.
template <int NB, bool SHELF_ONLY>
__device__ __forceinline__ void run_bands16(float (&x)[T], BandPtr bands, float *__restrict__ st) {
    uint32_t kind = bands[0].kind;
    float c0 = bands[0].c[0].f, c1 = bands[0].c[1].f, c2 = bands[0].c[2].f, c3 = bands[0].c[3].f, c4 = bands[0].c[4].f, c5 = bands[0].c[5].f;
    float s1 = st[0], s2 = st[kLanes];
    asm volatile("" ::"s"(kind), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "v"(s1), "v"(s2));   // see run_bands_f32
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
        // prefetch band b+1 (the last iteration harmlessly re-reads band NB-1: no control flow, nothing to peel)
        const int bn = (b + 1 < NB) ? b + 1 : b;
        BandPtr nb = bands + bn;
        const uint32_t nkind = nb->kind;
        const float n0 = nb->c[0].f, n1 = nb->c[1].f, n2 = nb->c[2].f, n3 = nb->c[3].f, n4 = nb->c[4].f, n5 = nb->c[5].f;
        const float ns1 = st[bn * 2 * kLanes], ns2 = st[bn * 2 * kLanes + kLanes];
        if (SHELF_ONLY) band16_shelf(x, s1, s2, kind, c0, c1, c2, c3, c4, c5);
        else band16_any(x, s1, s2, kind, c0, c1, c2, c3, c4, c5);
        st[b * 2 * kLanes] = s1;             // unconditional: a bypassed band writes back what it read
        st[b * 2 * kLanes + kLanes] = s2;
        kind = nkind; c0 = n0; c1 = n1; c2 = n2; c3 = n3; c4 = n4; c5 = n5; s1 = ns1; s2 = ns2;
    }
}

template <bool TAIL, int NB, bool SHELF_ONLY = false>
__device__ __forceinline__ void run_bands(float (&x)[T], int n, BandPtr bands, float *__restrict__ st) {
    if (TAIL && n != T) run_bands_f32<true, NB, SHELF_ONLY>(x, n, bands, st);
    else run_bands16<NB, SHELF_ONLY>(x, bands, st);
}

// Per-lane parameters (the one-stream float kernel): every lane reads ITS stream's DevImage, so coefficients are vector
// loads into VGPRs and the band kind is a per-lane value — the switch below is ordinary SIMT divergence (lanes of one kind
// run together, kinds one after the other).  Same arithmetic, same order.
template <bool TAIL, int NB, bool SHELF_ONLY = false, bool FMA = false>
__device__ __forceinline__ void run_bands(float (&x)[T], int n, const DevBand *bands, float *__restrict__ st) {
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    // band b+1's 32-byte descriptor (two 16-byte vector loads per lane) and state pair are requested before band b runs
    u32x8 cur = *reinterpret_cast<const u32x8 *>(bands);
    float s1 = st[0], s2 = st[kLanes];
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
        const int bn = (b + 1 < NB) ? b + 1 : b;
        const u32x8 nxt = *reinterpret_cast<const u32x8 *>(bands + bn);
        const float ns1 = st[bn * 2 * kLanes], ns2 = st[bn * 2 * kLanes + kLanes];
        const uint32_t kind = cur[6];
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(kind);
        const float c0 = as_f(cur[0]), c1 = as_f(cur[1]), c2 = as_f(cur[2]), c3 = as_f(cur[3]), c4 = as_f(cur[4]), c5 = as_f(cur[5]);
        if ((!TAIL || n == T) && __all(kind == k0)) {
            // the usual case: all lanes of the wave use the same form at this band index (presets of one product family):
            // the hand-written loop, with per-lane coefficients in VGPRs
            if (FMA) {
                if (SHELF_ONLY) band16vf_shelf(x, s1, s2, k0, c0, c1, c2, c3, c4, c5);
                else band16vf_any(x, s1, s2, k0, c0, c1, c2, c3, c4, c5);
            } else if (SHELF_ONLY) band16v_shelf(x, s1, s2, k0, c0, c1, c2, c3, c4, c5);
            else band16v_any(x, s1, s2, k0, c0, c1, c2, c3, c4, c5);
        } else if (kind != K_BYPASS) {
            if (SHELF_ONLY) band_loop_f32<TAIL, K_SVF_SHELF, FMA>(x, n, s1, s2, c0, c1, c2, c3, c4, c5);
            else switch (kind) {
                case K_BIQUAD: band_loop_f32<TAIL, K_BIQUAD, FMA>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_LP: band_loop_f32<TAIL, K_SVF_LP, FMA>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_HP: band_loop_f32<TAIL, K_SVF_HP, FMA>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_PK: band_loop_f32<TAIL, K_SVF_PK, FMA>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
                default: band_loop_f32<TAIL, K_SVF_SHELF, FMA>(x, n, s1, s2, c0, c1, c2, c3, c4, c5); break;
            }
        }
        st[b * 2 * kLanes] = s1;
        st[b * 2 * kLanes + kLanes] = s2;
        cur = nxt; s1 = ns1; s2 = ns2;
    }
}

// soft-knee upward gain computer, leveller.c:124-139
__device__ __forceinline__ float gain_computer(float x_db, float thr, float ratio, float knee) {
    float half = knee * 0.5f;
    if (x_db > (thr + half)) return 0.0f;
    if (x_db >= (thr - half)) {
        float d = thr + half - x_db;
        return (1.0f - 1.0f / ratio) * d * d / (2.0f * knee);
    }
    return (thr - x_db) * (1.0f - 1.0f / ratio);
}

// per-packet gain decision, leveller.c:174-206 (float) == :304-332 (Q28); libm -> dspi_detmath.h: the correctly rounded log10f, a^count and
// 10^y in their DEVICE forms (step 1 + the exception tables of include/dspi_detmath_tables.h — the same floats as the oracle's two-step
// functions for every argument, without the double-double code, whose mere presence cost the headline kernel 2 %: profiles/r06_detmath.md)
template <bool FMA = false, class IMG>
__device__ __forceinline__ float leveller_block_gain(IMG img, float &gsm_db, float rms_sq, uint32_t count) {
    float rms_db = 10.0f * dspi_det_log10f_tab(rms_sq + 1e-30f);
    float gc;
    if (rms_db < img->lv_gate_db) gc = 0.0f;
    else {
        gc = gain_computer(rms_db, img->lv_threshold_db, img->lv_ratio, img->lv_knee_db);
        gc += img->lv_makeup_db;
        if (gc > img->lv_max_gain_db) gc = img->lv_max_gain_db;
    }
    float a_s = (gc < gsm_db) ? img->lv_alpha_attack : img->lv_alpha_release;
    float alpha = dspi_det_powf_tab(a_s, (float)count);          // (the host has checked this context's alphas and block length against the exact form)
    gsm_db = mad1<FMA>(alpha, gsm_db, (1.0f - alpha) * gc);      // leveller.c:200
    return dspi_det_exp10f_tab(gsm_db / 20.0f);                   // leveller.c:206: powf(10.0f, x)
}

// Workgroup barrier for the chunk hand-off.  The only data exchanged between waves inside the time loop
// lives in LDS, so the barrier must drain LDS traffic (lgkmcnt) but NOT the HBM stores/loads in flight:
// __syncthreads() would add `s_waitcnt vmcnt(0)` and serialise every step behind its own write-backs.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifdef DSPI_WAVE_TIMING
// development aid: per-wave busy cycles (outside the barrier) and total cycles, summed over workgroups
__device__ unsigned long long g_wave_timing[36 + 48];   // [36 + 4*rank + k]: phase timers of the packed output waves   // [2w] busy, [2w+1] total, [24+w] HW_ID of wave w of workgroup 0
#define WT_PHASE(r, k, t) do { if (lane == 0) atomicAdd(&g_wave_timing[36 + 4 * (r) + (k)], (unsigned long long)(t)); } while (0)
#define WT_NOW() __builtin_amdgcn_s_memtime()
#define WT_COUNT(k, lo) do { if ((lo) == 0) atomicAdd(&g_wave_timing[36 + (k)], 1ull); } while (0)      // slots of the unused phase timers of ranks 0-2
#define WT_DECL unsigned long long wt_busy = 0, wt_t0 = __builtin_amdgcn_s_memtime(), wt_start = wt_t0
#define WT_BEFORE_BARRIER wt_busy += __builtin_amdgcn_s_memtime() - wt_t0
#define WT_AFTER_BARRIER wt_t0 = __builtin_amdgcn_s_memtime()
#define WT_FINISH(w) do { if (lane == 0) { atomicAdd(&g_wave_timing[(w) * 2], wt_busy); atomicAdd(&g_wave_timing[(w) * 2 + 1], __builtin_amdgcn_s_memtime() - wt_start); \
    if (blockIdx.x == 0) { uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_wave_timing[24 + (w)] = hw; } } } while (0)
#else
#define WT_PHASE(r, k, t)
#define WT_NOW() 0ull
#define WT_COUNT(k, lo)
#define WT_DECL
#define WT_BEFORE_BARRIER
#define WT_AFTER_BARRIER
#define WT_FINISH(w)
#endif

struct Geo {   // loop geometry shared by the four waves
    uint32_t n_blocks, B, cpb, items, lag, steps;
};

// ==========================================================================================
// wave 0 — float flavour
// ==========================================================================================
struct MasterF32 {
    float lpL, lpR, apL, apR;                 // crossfeed state (crossfeed.h:45-52)
    float env_l, env_r, gsm_db, g_cur, g_prev;  // LevellerState scalars (leveller.h:104-113)
    uint32_t rp1, rp2;                        // ring bases of the pass-1 / pass-2 packet
    float p2_gain, p2_step;
    float pk_l, pk_r;
    uint32_t clip;
    u32x4 pre[T / 4];                         // next chunk's PCM words, fetched one step ahead
    uint32_t pre_valid;
};

// The float one-stream kernel runs with PER-LANE parameter images (img is a per-lane pointer): it serves exactly the lanes
// whose neighbours carry different presets.  Consequences: every `img->` read is a vector load, every parameter test is
// SIMT divergence, and the step schedule cannot depend on a lane's flags — so every lane's samples travel through the ring
// (one packet late), whether its leveller is on or not; a leveller that is off just copies them through.
template <bool TAIL, bool FMA>
__device__ __forceinline__ void master_step_f32(const KArgs &a, const DevImage *img, const StateMap &sm, const Geo &g,
                                                MasterF32 &m, float *__restrict__ lds_state, float *__restrict__ xch_base,
                                                uint32_t wg, uint32_t lane, uint32_t col, uint32_t stream, bool active,
                                                bool do_p1, uint32_t k1, uint32_t c1, bool do_item, uint32_t kq, uint32_t cq, uint32_t q) {
    constexpr uint32_t ROW = make_state_map(1).row;   // global arrays are [..][row] with this stream in column `col`
    const uint32_t flags = img->flags;
    const bool lev_on = flags & IF_LEVELLER_ON;
    float xl[T], xr[T];
    uint32_t *ring = a.ring + (size_t)wg * kRingLen * 2 * ROW + col;
    const int nq = TAIL ? (int)min((uint32_t)T, g.B - cq * T) : T;

    // Pass-2 operands come out of the ring (written >= one packet ago): issue those loads first so
    // they are in flight underneath the pass-1 arithmetic below.
    float ol[T], orr[T];
    if (do_item) {
        if (lev_on && cq == 0) {   // latch the ramp of packet kq before pass 1 below can decide the next packet's gain
            if (g.B == 1) { m.p2_gain = m.g_cur; m.p2_step = 0.0f; }
            else { m.p2_step = (m.g_cur - m.g_prev) / (float)(g.B - 1); m.p2_gain = m.g_prev; }
        }
        const uint32_t back = (lev_on && (flags & IF_LOOKAHEAD)) ? (uint32_t)kLookahead : 0u;
        const uint32_t base = (m.rp2 + cq * T - back) & (kRingLen - 1);
        if (__all(base + T <= (uint32_t)kRingLen)) {          // no wrap inside the chunk: one base + immediate offsets
            const uint32_t *rl = ring + (size_t)base * ROW;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= nq) break;
                uint32_t ul = 0, ur = 0;
                if (active) { ul = rl[i * ROW]; ur = rl[(kRingLen + i) * ROW]; }
                ol[i] = as_f(ul);
                orr[i] = as_f(ur);
            }
        } else {
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= nq) break;
                uint32_t pos = (base + i) & (kRingLen - 1);
                uint32_t ul = 0, ur = 0;
                if (active) { ul = ring[(size_t)pos * ROW]; ur = ring[(size_t)(kRingLen + pos) * ROW]; }
                ol[i] = as_f(ul);
                orr[i] = as_f(ur);
            }
        }
    }

    if (do_p1) {
        const int n = TAIL ? (int)min((uint32_t)T, g.B - c1 * T) : T;
        // ---- PASS 1: input conversion + preamp (usb_audio.c:591-686) ----
        const size_t frame0 = ((size_t)stream * g.n_blocks + k1) * g.B + (size_t)c1 * T;
        if (a.bit_depth == 24) {
            const float gl = (1.0f / 8388608.0f) * img->preamp[0].f, gr = (1.0f / 8388608.0f) * img->preamp[1].f;
            const uint16_t *p = reinterpret_cast<const uint16_t *>(static_cast<const uint8_t *>(a.pcm) + frame0 * 6);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= n) break;
                uint32_t w0 = 0, w1 = 0, w2 = 0;
                if (active) { w0 = p[i * 3]; w1 = p[i * 3 + 1]; w2 = p[i * 3 + 2]; }
                int32_t l = (int32_t)((w0 | (w1 << 16)) << 8) >> 8;          // bytes 0..2, sign-extended
                int32_t r = (int32_t)(((w1 >> 8) | (w2 << 8)) << 8) >> 8;    // bytes 3..5
                xl[i] = (float)l * gl;
                xr[i] = (float)r * gr;
            }
        } else {
            const float gl = (1.0f / 32768.0f) * img->preamp[0].f, gr = (1.0f / 32768.0f) * img->preamp[1].f;
            const uint32_t *p = static_cast<const uint32_t *>(a.pcm) + frame0;
            if (!TAIL) {
                if (!m.pre_valid) {
#pragma unroll
                    for (int v = 0; v < T / 4; ++v) { m.pre[v] = u32x4{0, 0, 0, 0}; if (active) m.pre[v] = ld4(p + v * 4); }
                }
#pragma unroll
                for (int v = 0; v < T / 4; ++v) {
                    const u32x4 w = m.pre[v];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xl[v * 4 + e] = (float)(int32_t)(int16_t)(w[e] & 0xffffu) * gl;
                        xr[v * 4 + e] = (float)((int32_t)w[e] >> 16) * gr;
                    }
                }
                // fetch the next chunk now; it lands while this one goes through loudness + EQ
                // (only a FULL next chunk: in a kernel for ragged packets the short chunk is read by the guarded variant, and
                // 16 frames from its start may run past the stream's buffer)
                const uint32_t nc = (c1 + 1 < g.cpb) ? c1 + 1 : 0u;
                const bool more = ((c1 + 1 < g.cpb) || (k1 + 1 < g.n_blocks)) && (nc + 1) * T <= g.B;
                m.pre_valid = more;
                if (more && active) {
#pragma unroll
                    for (int v = 0; v < T / 4; ++v) m.pre[v] = ld4(p + T + v * 4);   // chunks of one stream are contiguous across packets
                }
            } else {
                m.pre_valid = 0;
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (i >= n) break;
                    uint32_t w = active ? p[i] : 0u;
                    xl[i] = (float)(int32_t)(int16_t)(w & 0xffffu) * gl;
                    xr[i] = (float)((int32_t)w >> 16) * gr;
                }
            }
        }
        // ---- loudness shelves (usb_audio.c:688-718) ----
        run_bands<TAIL, 2, true, FMA>(xl, n, &img->loud[0], lds_state + (sm.loud + 0) * kLanes + lane);
        run_bands<TAIL, 2, true, FMA>(xr, n, &img->loud[0], lds_state + (sm.loud + 4) * kLanes + lane);
        // ---- PASS 2: master EQ (usb_audio.c:721-728) ----
        if (!(flags & IF_BYPASS_MASTER_EQ)) {
            if (!(img->ch_bypassed & 1u)) run_bands<TAIL, kBands, false, FMA>(xl, n, &img->eq[0][0], lds_state + (sm.eq + 0) * kLanes + lane);
            if (!(img->ch_bypassed & 2u)) run_bands<TAIL, kBands, false, FMA>(xr, n, &img->eq[1][0], lds_state + (sm.eq + kBands * 2) * kLanes + lane);
        }
        {
            // ---- leveller pass 1: RMS envelopes (leveller.c:155-172); every lane parks its samples in the ring ----
            const float ar = img->lv_alpha_rms, nar = 1.0f - ar;
            const uint32_t base = (m.rp1 + c1 * T) & (kRingLen - 1);
            const bool flat = __all(base + T <= (uint32_t)kRingLen);
            uint32_t *wl = ring + (size_t)base * ROW;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= n) break;
                if (lev_on) {
                    m.env_l = mad1<FMA>(ar, m.env_l, nar * (xl[i] * xl[i]));
                    m.env_r = mad1<FMA>(ar, m.env_r, nar * (xr[i] * xr[i]));
                }
                if (active) {
                    if (flat) { wl[i * ROW] = as_u(xl[i]); wl[(kRingLen + i) * ROW] = as_u(xr[i]); }
                    else {
                        uint32_t pos = (base + i) & (kRingLen - 1);
                        ring[(size_t)pos * ROW] = as_u(xl[i]);
                        ring[(size_t)(kRingLen + pos) * ROW] = as_u(xr[i]);
                    }
                }
            }
            if (c1 == g.cpb - 1) {   // end of packet: gain decision (leveller.c:168-206)
                if (lev_on) {
                    if (m.env_l < 1e-30f) m.env_l = 0.0f;
                    if (m.env_r < 1e-30f) m.env_r = 0.0f;
                    float rms_sq = (m.env_l > m.env_r) ? m.env_l : m.env_r;
                    float gn = leveller_block_gain<FMA>(img, m.gsm_db, rms_sq, g.B);
                    m.g_prev = m.g_cur;
                    m.g_cur = gn;
                }
                m.rp1 = (m.rp1 + g.B) & (kRingLen - 1);
            }
        }
    }

    if (!do_item) return;

    if (lev_on) {
        // ---- leveller pass 2: interpolated gain, lookahead, gain-cap limiter (leveller.c:208-261) ----
        const float ceil_ = 0.70795f;   // LEVELLER_LIMITER_CEIL
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (TAIL && i >= nq) break;
            float peak = fabsf(ol[i]), pr = fabsf(orr[i]);
            if (pr > peak) peak = pr;
            float gg = m.p2_gain;
            if (peak > 0.0f && gg > 1.0f) {
                float mg = ceil_ / peak;
                if (mg < gg) gg = (mg > 1.0f) ? mg : 1.0f;
            }
            xl[i] = ol[i] * gg;
            xr[i] = orr[i] * gg;
            m.p2_gain += m.p2_step;
        }
    } else {      // leveller bypassed (usb_audio.c:731-738): the samples pass through untouched
#pragma unroll
        for (int i = 0; i < T; ++i) { if (TAIL && i >= nq) break; xl[i] = ol[i]; xr[i] = orr[i]; }
    }
    if (cq == g.cpb - 1) m.rp2 = (m.rp2 + g.B) & (kRingLen - 1);

    // ---- PASS 3: master peaks (pre-crossfeed) + crossfeed (usb_audio.c:741-749, crossfeed.c:132-156) ----
    if (cq == 0) { m.pk_l = 0.0f; m.pk_r = 0.0f; }
    const bool xf = flags & IF_CROSSFEED_ON;
    const float a0 = img->xf_lp_a0.f, b1 = img->xf_lp_b1.f, apa = img->xf_ap_a.f;
    float *xch = xch_base + (size_t)(q & 1u) * (2 * T * kLanes) + lane;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (TAIL && i >= nq) break;
        float ml = xl[i], mr = xr[i];
        float al = fabsf(ml); if (al > m.pk_l) m.pk_l = al;
        float ar = fabsf(mr); if (ar > m.pk_r) m.pk_r = ar;
        if (xf) {
            float lpl = mad1<FMA>(a0, ml, b1 * m.lpL);
            float lpr = mad1<FMA>(a0, mr, b1 * m.lpR);
            m.lpL = lpl; m.lpR = lpr;
            float apl = mad1<FMA>(apa, lpl, m.apL);
            m.apL = mad1<FMA>(-apa, apl, lpl);
            float apr = mad1<FMA>(apa, lpr, m.apR);
            m.apR = mad1<FMA>(-apa, apr, lpr);
            float dl = (ml - lpl) + apr;
            float dr = (mr - lpr) + apl;
            ml = dl; mr = dr;
        }
        xch[i * kLanes] = ml;
        xch[(T + i) * kLanes] = mr;
    }
    if (cq == g.cpb - 1) {   // usb_audio.c:963-966
        uint32_t p0 = (uint32_t)(fminf(1.0f, m.pk_l) * 32767.0f), p1 = (uint32_t)(fminf(1.0f, m.pk_r) * 32767.0f);
        if (m.pk_l > 1.001f) m.clip |= 1u;
        if (m.pk_r > 1.001f) m.clip |= 2u;
        if (active) {
            uint32_t *gs = a.state + (size_t)wg * sm.n_slots * ROW + col;
            gs[(sm.peaks + 0) * ROW] = p0;
            gs[(sm.peaks + 1) * ROW] = p1;
            if (a.peaks) {
                uint16_t *pp = a.peaks + ((size_t)stream * g.n_blocks + kq) * sm.n_ch;
                pp[0] = (uint16_t)p0; pp[1] = (uint16_t)p1;
            }
        }
    }
}

// ==========================================================================================
// waves 1..3 — float flavour
// ==========================================================================================
struct OutF32 {
    uint32_t widx;                       // delay_write_idx at the start of the current packet
    uint32_t loading, counter;           // preset_loading, preset_mute_counter
    float smooth;                        // preset_mute_smooth_gain
    float vmm;                           // vol_mul_master of the current packet
    uint32_t clip;
};

template <bool TAIL, bool FMA>
__device__ __forceinline__ void output_item_f32(const KArgs &a, const DevImage *img, const StateMap &sm, const Geo &g,
                                                OutF32 &s, float *__restrict__ lds_state, float *__restrict__ lds_pk, const float *__restrict__ xch_base,
                                                uint32_t wg, uint32_t lane, uint32_t col, uint32_t stream, bool active,
                                                int o_first, int o_count, uint32_t kq, uint32_t cq, uint32_t q) {
    constexpr uint32_t ROW = make_state_map(1).row;
    const uint32_t flags = img->flags;
    const int n = TAIL ? (int)min((uint32_t)T, g.B - cq * T) : T;
    const int N = sm.n_out;
    if (cq == 0) {
        // ---- packet scalars: preset-mute envelope (usb_audio.c:466-498) and volumes (:564-571) ----
        bool act = s.loading != 0;
        if (act) {
            if (s.counter > g.B) s.counter -= g.B;
            else { s.counter = 0; s.loading = 0; }
        }
        float target = act ? 0.0f : 1.0f;
        float step = (float)g.B / (float)img->mute_transition;
        if (step > 1.0f) step = 1.0f;
        float gg = s.smooth;
        if (gg < target) { gg += step; if (gg > target) gg = target; }
        else if (gg > target) { gg -= step; if (gg < target) gg = target; }
        s.smooth = gg;
        float vol_mul = img->vol.f;
        vol_mul *= gg;
        s.vmm = vol_mul * img->master.f;
    }
    // post-crossfeed L/R chunk
    float L[T], R[T];
    const float *xch = xch_base + (size_t)(q & 1u) * (2 * T * kLanes) + lane;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (TAIL && i >= n) break;
        L[i] = xch[i * kLanes];
        R[i] = xch[(T + i) * kLanes];
    }
    const bool sub_active = flags & IF_SUB_ACTIVE;
    const size_t F = (size_t)g.n_blocks * g.B;
    const size_t frame0 = (size_t)kq * g.B + (size_t)cq * T;
    int32_t held[T];   // left words of a pair waiting for the right channel
#pragma unroll
    for (int i = 0; i < T; ++i) held[i] = 0;

#pragma unroll 1
    for (int j = 0; j < o_count; ++j) {
        const int o = o_first + j;
        const bool is_sub = (o == N - 1);
        const bool processed = !is_sub || sub_active;     // EQ-worker mode leaves the sub untouched (usb_audio.c:844-845)
        const bool enabled = (img->out_enabled >> o) & 1u;
        const bool muted = (img->out_mute >> o) & 1u;
        float x[T];
        // Delay line: the packet's read positions were written >= T samples ago unless the delay is
        // shorter than a chunk (or aliases to zero at dly == max_delay, config.h:83-88), so the reads can
        // be issued up front and land underneath the EQ arithmetic.
        const int32_t dly = img->delay_samples[o];
        const bool dl_on = processed && (flags & IF_ANY_DELAY) && dly > 0;
        const bool dl_alias = dl_on && dly >= sm.max_delay;          // reads back what it just wrote
        const bool dl_early = dl_on && !dl_alias && dly >= T;
        const uint32_t dmask = (uint32_t)sm.max_delay - 1u;
        uint32_t *line = a.dlines + ((size_t)wg * N + o) * (size_t)sm.max_delay * ROW + col;
        const uint32_t w0 = s.widx + cq * T;
        float dl[T];
        const uint32_t wb = w0 & dmask, rb = (w0 - (uint32_t)dly) & dmask;
        const bool w_flat = __all(wb + T <= (uint32_t)sm.max_delay), r_flat = __all(rb + T <= (uint32_t)sm.max_delay);
        if (dl_early) {
            const uint32_t *rl = line + (size_t)rb * ROW;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= n) break;
                uint32_t u = 0;
                if (active) u = r_flat ? rl[i * ROW] : line[(size_t)((rb + i) & dmask) * ROW];
                dl[i] = as_f(u);
            }
        }
        // ---- PASS 4: matrix mix (usb_audio.c:753-779) ----
        const float gl = img->mix[0][o].f, gr = img->mix[1][o].f;
        if (enabled && gl != 0.0f && gr != 0.0f) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; x[i] = mad1<FMA>(L[i], gl, R[i] * gr); }
        } else if (enabled && gl != 0.0f) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; x[i] = L[i] * gl; }
        } else if (enabled && gr != 0.0f) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; x[i] = R[i] * gr; }
        } else {
#pragma unroll
            for (int i = 0; i < T; ++i) x[i] = 0.0f;
        }
        if (processed) {
            // ---- PASS 5: per-output EQ + gain (usb_audio.c:877-895) ----
            if (enabled) {
                const int ch = 2 + o;
                if (!muted && !((img->ch_bypassed >> ch) & 1u))
                    run_bands<TAIL, kBands, false, FMA>(x, n, &img->eq[ch][0], lds_state + (sm.eq + ch * kBands * 2) * kLanes + lane);
                float gain = muted ? 0.0f : img->out_gain_lin[o] * s.vmm;
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (TAIL && i >= n) break;
                    float y = x[i] * gain;               // x*1.0f == x, so the reference's "gain != 1" test is not needed
                    x[i] = (gain == 0.0f) ? 0.0f : y;    // memset branch
                }
            }
            // ---- PASS 6: delay line (usb_audio.c:898-912); [position][lane] rows in HBM ----
            if (dl_early || dl_alias) {
                uint32_t *wl = line + (size_t)wb * ROW;
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (TAIL && i >= n) break;
                    if (active) { if (w_flat) wl[i * ROW] = as_u(x[i]); else line[(size_t)((wb + i) & dmask) * ROW] = as_u(x[i]); }
                    if (dl_early) x[i] = dl[i];
                }
            } else if (dl_on) {      // delay shorter than a chunk: the reference's per-sample order
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (TAIL && i >= n) break;
                    uint32_t w = (w0 + i) & dmask;
                    if (active) {
                        line[(size_t)w * ROW] = as_u(x[i]);
                        x[i] = as_f(line[(size_t)((w - (uint32_t)dly) & dmask) * ROW]);
                    }
                }
            }
        }
        // ---- PASS 7: peaks, output words (usb_audio.c:914-959) ----
        float *pkp = lds_pk + (2 + o) * kLanes + lane;
        float pk = (cq == 0) ? 0.0f : *pkp;
        const bool metered = !is_sub || (sub_active && enabled);
        if (metered) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; float av = fabsf(x[i]); if (av > pk) pk = av; }
        }
        *pkp = pk;
        if (cq == g.cpb - 1) {
            uint32_t p16 = metered ? (uint32_t)(fminf(1.0f, pk) * 32767.0f) : 0u;
            if (metered && pk > 1.001f) s.clip |= 1u << (2 + o);
            if (active) {
                a.state[((size_t)wg * sm.n_slots + sm.peaks + 2 + o) * ROW + col] = p16;
                if (a.peaks) a.peaks[((size_t)stream * g.n_blocks + kq) * sm.n_ch + 2 + o] = (uint16_t)p16;
            }
        }
        if (is_sub) {
            if (a.sub && active && a.tiled_out) {
                int32_t *dst = a.sub + ((size_t)wg * F + frame0) * ROW + col;
                const bool live = sub_active && enabled;
#pragma unroll
                for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[(size_t)i * ROW] = live ? f2i_sat(x[i] * 268435456.0f) : 0; }
            } else if (a.sub && active) {
                int32_t *dst = a.sub + (size_t)stream * F + frame0;
                const bool live = sub_active && enabled;
                if (!TAIL) {
#pragma unroll
                    for (int v = 0; v < T / 4; ++v) {
                        u32x4 w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = live ? (uint32_t)f2i_sat(x[v * 4 + e] * 268435456.0f) : 0u;
                        st4(dst + v * 4, w);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < T; ++i) { if (i >= n) break; dst[i] = live ? f2i_sat(x[i] * 268435456.0f) : 0; }
                }
            }
        } else {
            // a pair whose two outputs are both disabled is zero-filled whatever its delay lines still hold (usb_audio.c:930-933)
            const bool pair_dead = !(((img->out_enabled >> o) | (img->out_enabled >> (o ^ 1))) & 1u);
            int32_t wv[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= n) break;
                float d = fmaxf(-1.0f, fminf(1.0f, x[i]));
                wv[i] = pair_dead ? 0 : (int32_t)(d * 8388607.0f);
                if (a.i2s_slots && ((img->i2s_pairs >> (o >> 1)) & 1u)) wv[i] = (int32_t)((uint32_t)wv[i] << 8);      // I2S slot (audio_i2s_multi.c:217-226)
            }
            const int pair = o >> 1, side = o & 1;
            const bool partner_here = side ? (j >= 1) : (j + 1 < o_count && o + 1 < N - 1);
            if (a.pairs && active && a.tiled_out) {
                int32_t *dst = a.pairs + (((size_t)wg * (N - 1) + o) * F + frame0) * ROW + col;
#pragma unroll
                for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[(size_t)i * ROW] = wv[i]; }
            } else if (a.pairs && active) {
                int32_t *dst = a.pairs + (((size_t)(stream - a.pairs_stream0) * sm.n_pairs + pair) * F + frame0) * 2;
                if (side == 1 && partner_here) {
                    if (!TAIL) {
#pragma unroll
                        for (int v = 0; v < T / 2; ++v) {
                            u32x4 w = {(uint32_t)held[2 * v], (uint32_t)wv[2 * v], (uint32_t)held[2 * v + 1], (uint32_t)wv[2 * v + 1]};
                            st4(dst + v * 4, w);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < T; ++i) { if (i >= n) break; dst[i * 2] = held[i]; dst[i * 2 + 1] = wv[i]; }
                    }
                } else if (!(side == 0 && partner_here)) {
#pragma unroll
                    for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[i * 2 + side] = wv[i]; }
                }
            }
            if (side == 0) {
#pragma unroll
                for (int i = 0; i < T; ++i) held[i] = wv[i];
            }
        }
    }
    if (cq == g.cpb - 1 && (flags & IF_ANY_DELAY)) s.widx = (s.widx + g.B) & ((uint32_t)sm.max_delay - 1u);
}


// ==========================================================================================
// RP2040 Q28 flavour — integer arithmetic, every step wrapping mod 2^32 exactly as the Thumb code does
// ==========================================================================================
// fast_mul_q28 (dsp_pipeline.c:47-58, inlined 5x per sample in dsp_process_rp2040.S:63-77): NOT a 64-bit product.
// The 16x16 partial products map onto v_mul_i32_i24 / v_mad_i32_i24 (full rate), no 64-bit multiplies.
__device__ __forceinline__ int32_t qmul(int32_t a, int32_t b) {
    int32_t ah = a >> 16, bh = b >> 16;
    uint32_t al = (uint32_t)a & 0xFFFFu, bl = (uint32_t)b & 0xFFFFu;
    uint32_t high = (uint32_t)(ah * bh);
    int32_t mid = (int32_t)((uint32_t)ah * bl + al * (uint32_t)bh);
    return (int32_t)((high << 4) + (uint32_t)(mid >> 12));
}
// fast_mul_q15 (config.h:556-567)
__device__ __forceinline__ int32_t q15mul(int32_t sample, int32_t gain) {
    int32_t sh = sample >> 16, gh = gain >> 16;
    uint32_t sl = (uint32_t)sample & 0xFFFFu, gl = (uint32_t)gain & 0xFFFFu;
    uint32_t hh = (uint32_t)(sh * gh);
    uint32_t mid = (uint32_t)sh * gl + sl * (uint32_t)gh;
    uint32_t ll = sl * gl;
    return (int32_t)((hh << 17) + (mid << 1) + (ll >> 15));
}
__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wabs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }
// TDF2 cascade of dsp_process_rp2040.S:225-394 with the state pair in LDS; operands of band b+1 fetched during band b
template <bool TAIL, int NB>
__device__ __forceinline__ void run_bands_q28(int32_t (&x)[T], int n, BandPtr bands, int32_t *__restrict__ st) {
    uint32_t kind = bands[0].kind;
    int32_t b0 = bands[0].c[0].i, b1 = bands[0].c[1].i, b2 = bands[0].c[2].i, a1 = bands[0].c[3].i, a2 = bands[0].c[4].i;
    int32_t s1 = st[0], s2 = st[kLanes];
    asm volatile("" ::"s"(kind), "s"(b0), "s"(b1), "s"(b2), "s"(a1), "s"(a2), "v"(s1), "v"(s2));
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
        const int bn = (b + 1 < NB) ? b + 1 : b;
        BandPtr nb = bands + bn;
        const uint32_t nkind = nb->kind;
        const int32_t n0 = nb->c[0].i, n1 = nb->c[1].i, n2 = nb->c[2].i, n3 = nb->c[3].i, n4 = nb->c[4].i;
        const int32_t ns1 = st[bn * 2 * kLanes], ns2 = st[bn * 2 * kLanes + kLanes];
        if (kind != K_BYPASS) {
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= n) break;
                const int32_t in = x[i];
                const int32_t y = wadd(qmul(b0, in), s1);
                const int32_t t1 = qmul(b1, in), t3 = qmul(b2, in);
                const int32_t t2 = qmul(a1, y), t4 = qmul(a2, y);
                s1 = wadd(wsub(t1, t2), s2);
                s2 = wsub(t3, t4);
                x[i] = y;
            }
            st[b * 2 * kLanes] = s1;
            st[b * 2 * kLanes + kLanes] = s2;
        }
        kind = nkind; b0 = n0; b1 = n1; b2 = n2; a1 = n3; a2 = n4; s1 = ns1; s2 = ns2;
    }
}

// per-lane parameters (a row whose streams carry different presets): coefficients by vector loads, the bypass flag per lane
template <bool TAIL, int NB>
__device__ __forceinline__ void run_bands_q28(int32_t (&x)[T], int n, const DevBand *bands, int32_t *__restrict__ st) {
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
        if (bands[b].kind == K_BYPASS) continue;
        const int32_t b0 = bands[b].c[0].i, b1 = bands[b].c[1].i, b2 = bands[b].c[2].i, a1 = bands[b].c[3].i, a2 = bands[b].c[4].i;
        int32_t s1 = st[b * 2 * kLanes], s2 = st[b * 2 * kLanes + kLanes];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (TAIL && i >= n) break;
            const int32_t in = x[i];
            const int32_t y = wadd(qmul(b0, in), s1);
            const int32_t t1 = qmul(b1, in), t3 = qmul(b2, in);
            const int32_t t2 = qmul(a1, y), t4 = qmul(a2, y);
            s1 = wadd(wsub(t1, t2), s2);
            s2 = wsub(t3, t4);
            x[i] = y;
        }
        st[b * 2 * kLanes] = s1;
        st[b * 2 * kLanes + kLanes] = s2;
    }
}

struct MasterQ28 {
    int32_t lpL, lpR, apL, apR;
    int32_t env_l, env_r;
    float gsm_db;
    int32_t g_cur, g_prev;
    int32_t g_last;          // hand-off: the gain the previous packet's ramp ended on
    uint32_t rp1, rp2;
    // pass-2 gain ramp: gain_i = g_prev + trunc((g_cur - g_prev) * i / (B-1)) (leveller.c:352) kept exact without 64-bit
    // division: (g_cur-g_prev) = D*(B-1) + R  ->  gain_i = g_prev + D*i + trunc(R*i/(B-1)), the last term carried incrementally
    int32_t p2_base, p2_D, p2_R, p2_acc, p2_carry;
    int32_t pk_l, pk_r;
    uint32_t clip;
};

// The Q28 master side is split over two waves so that no wave carries 24 band visits per chunk while another carries 10:
//   role 0: pass 1 of the LEFT channel (conversion, loudness, master EQ, envelope -> ring) + the hand-off of both channels
//           (ring -> leveller pass 2, peaks, crossfeed -> LDS), `lag` steps later;
//   role 3: pass 1 of the RIGHT channel + the fifth output.
// Every sample travels through the ring (also with the leveller off: it is the only path from role 3 to role 0).  Role 3
// posts its envelope in LDS when a packet ends; role 0 takes the gain decision (leveller.c:304-334) at the start of the
// next step from its own end-of-packet envelope and the posted one, and queues it for the hand-off of that packet.
// Visibility of the right-channel rows: role 3 drains its stores in the middle of the NEXT step (before that step's ring
// stores), lag >= 2, and role 0 reads them with agent-scope loads (past the CU's vector L1).
// IMG = ImgPtr: the workgroup's image, scalar loads (PL false).  IMG = const DevImage *: per-lane images (PL true), for rows
// whose streams carry different presets.
constexpr int kQ28Mail = 4;      // queued gain decisions (packets between decision and hand-off: at most 2)

template <bool TAIL, int CH, class IMG>
__device__ __forceinline__ void master_p1_q28(const KArgs &a, IMG img, const StateMap &sm, const Geo &g, int32_t &env, uint32_t &rp1,
                                              int32_t *__restrict__ lds_state, uint32_t wg, uint32_t lane, uint32_t stream, uint32_t k1, uint32_t c1) {
    constexpr uint32_t ROW = make_state_map(0).row;
    const uint32_t flags = img->flags;
    const bool lev_on = flags & IF_LEVELLER_ON;
    const int32_t unity = 1 << 28;
    int32_t x[T];
    uint32_t *plane = a.ring + ((size_t)wg * 2 + CH) * kRingLen * ROW + lane;
    const int n = TAIL ? (int)min((uint32_t)T, g.B - c1 * T) : T;
    // ---- PASS 1: input conversion + preamp (usb_audio.c:997-1015) ----
    const size_t frame0 = ((size_t)stream * g.n_blocks + k1) * g.B + (size_t)c1 * T;
    const int32_t pre = img->preamp[CH].i;
    if (a.bit_depth == 24) {
        const uint16_t *p = reinterpret_cast<const uint16_t *>(static_cast<const uint8_t *>(a.pcm) + frame0 * 6);
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (TAIL && i >= n) break;
            int32_t v;
            if (CH == 0) { uint32_t w0 = p[i * 3], w1 = p[i * 3 + 1]; v = (int32_t)((w0 | (w1 << 16)) << 8) >> 2; }     // 24-bit left-justified, then >>2: net <<6
            else { uint32_t w1 = p[i * 3 + 1], w2 = p[i * 3 + 2]; v = (int32_t)(((w1 >> 8) | (w2 << 8)) << 8) >> 2; }
            x[i] = qmul(v, pre);
        }
    } else {
        const uint32_t *p = static_cast<const uint32_t *>(a.pcm) + frame0;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (TAIL && i >= n) break;
            const uint32_t w = p[i];
            const int32_t v = CH == 0 ? (int32_t)((uint32_t)(int32_t)(int16_t)(w & 0xffffu) << 14) : (int32_t)((uint32_t)((int32_t)w >> 16) << 14);
            x[i] = qmul(v, pre);
        }
    }
    // ---- loudness (usb_audio.c:1017-1047) and master EQ (:1049-1055) ----
    run_bands_q28<TAIL, 2>(x, n, &img->loud[0], lds_state + (sm.loud + 4 * CH) * kLanes + lane);
    if (!(flags & IF_BYPASS_MASTER_EQ) && !(img->ch_bypassed & (1u << CH)))
        run_bands_q28<TAIL, kBands>(x, n, &img->eq[CH][0], lds_state + (sm.eq + CH * kBands * 2) * kLanes + lane);
    // ---- leveller pass 1 (leveller.c:282-302): envelope, samples parked in the ring ----
    const int32_t aq = img->lv_alpha_rms_q28, naq = unity - aq;
    const uint32_t base = (rp1 + c1 * T) & (kRingLen - 1);
    const bool flat = __all(base + T <= (uint32_t)kRingLen);
    uint32_t *wl = plane + (size_t)base * ROW;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the previous step's rows are complete before this step's barrier
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (TAIL && i >= n) break;
        if (lev_on) {
            const int32_t sq = qmul(x[i], x[i]);
            env = wadd(qmul(aq, env), qmul(naq, sq));
        }
        if (flat) wl[i * ROW] = (uint32_t)x[i];
        else plane[(size_t)((base + i) & (kRingLen - 1)) * ROW] = (uint32_t)x[i];
    }
    if (c1 == g.cpb - 1) rp1 = (rp1 + g.B) & (kRingLen - 1);
}

// hand-off of chunk (kq, cq): both channels from the ring, pass 2, peaks, crossfeed -> LDS
template <bool TAIL, class IMG>
__device__ __forceinline__ void master_item_q28(const KArgs &a, IMG img, const StateMap &sm, const Geo &g, MasterQ28 &m, const int32_t *__restrict__ mail,
                                                int32_t *__restrict__ xch_base, uint32_t wg, uint32_t lane, uint32_t stream, uint32_t kq, uint32_t cq, uint32_t q) {
    constexpr uint32_t ROW = make_state_map(0).row;
    const uint32_t col = lane;
    const uint32_t flags = img->flags;
    const bool lev_on = flags & IF_LEVELLER_ON;
    int32_t xl[T], xr[T];
    const uint32_t *ring = a.ring + (size_t)wg * kRingLen * 2 * ROW + col;
    const int nq = TAIL ? (int)min((uint32_t)T, g.B - cq * T) : T;
    const int32_t unity = 1 << 28;
    if (lev_on && cq == 0) {
        const int32_t g_new = mail[(kq & (kQ28Mail - 1)) * kLanes + lane];       // decided when pass 1 of packet kq ended
        const int32_t d = wsub(g_new, m.g_last);
        if (g.B == 1) { m.p2_base = g_new; m.p2_D = 0; m.p2_R = 0; }
        else { const int32_t mm = (int32_t)g.B - 1; m.p2_base = m.g_last; m.p2_D = d / mm; m.p2_R = d % mm; }
        m.p2_acc = 0; m.p2_carry = 0;
        m.g_last = g_new;
    }
    {
        const uint32_t back = (lev_on && (flags & IF_LOOKAHEAD)) ? (uint32_t)kLookahead : 0u;
        const uint32_t base = (m.rp2 + cq * T - back) & (kRingLen - 1);
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (TAIL && i >= nq) break;
            const uint32_t pos = (base + i) & (kRingLen - 1);
            xl[i] = (int32_t)ring[(size_t)pos * ROW];
            xr[i] = (int32_t)__hip_atomic_load(ring + (size_t)(kRingLen + pos) * ROW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written by role 3
        }
    }
    if (lev_on) {
        // ---- leveller pass 2 (leveller.c:336-386) ----
        const float inv = 1.0f / (float)(1 << 28);
        const int32_t mm = (int32_t)g.B - 1;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (TAIL && i >= nq) break;
            int32_t gain = wadd(m.p2_base, m.p2_carry);
            if (gain > unity) {
                float peak = fabsf((float)xl[i] * inv), prk = fabsf((float)xr[i] * inv);
                if (prk > peak) peak = prk;
                if (peak > 0.0f) {
                    const float mgf = 0.70795f / peak;
                    const int32_t mgq = f2i_sat(mgf * (float)unity);
                    if (mgq < gain) gain = (mgq > unity) ? mgq : unity;
                }
            }
            xl[i] = qmul(xl[i], gain);
            xr[i] = qmul(xr[i], gain);
            // advance the ramp to sample i+1
            m.p2_base = wadd(m.p2_base, m.p2_D);
            m.p2_acc += m.p2_R;
            if (m.p2_acc >= mm && mm > 0) { m.p2_acc -= mm; m.p2_carry += 1; }
            else if (m.p2_acc <= -mm && mm > 0) { m.p2_acc += mm; m.p2_carry -= 1; }
        }
    }
    if (cq == g.cpb - 1) m.rp2 = (m.rp2 + g.B) & (kRingLen - 1);
    // ---- PASS 3: master peaks + crossfeed (usb_audio.c:1065-1073, crossfeed.c:161-180) ----
    if (cq == 0) { m.pk_l = 0; m.pk_r = 0; }
    const bool xf = flags & IF_CROSSFEED_ON;
    const int32_t a0 = img->xf_lp_a0.i, b1 = img->xf_lp_b1.i, apa = img->xf_ap_a.i;
    int32_t *xch = xch_base + (size_t)(q & 1u) * (2 * T * kLanes) + lane;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (TAIL && i >= nq) break;
        int32_t ml = xl[i], mr = xr[i];
        if (wabs(ml) > m.pk_l) m.pk_l = wabs(ml);
        if (wabs(mr) > m.pk_r) m.pk_r = wabs(mr);
        if (xf) {
            const int32_t lpl = wadd(qmul(a0, ml), qmul(b1, m.lpL));
            const int32_t lpr = wadd(qmul(a0, mr), qmul(b1, m.lpR));
            m.lpL = lpl; m.lpR = lpr;
            const int32_t apl = wadd(qmul(apa, lpl), m.apL);
            m.apL = wsub(lpl, qmul(apa, apl));
            const int32_t apr = wadd(qmul(apa, lpr), m.apR);
            m.apR = wsub(lpr, qmul(apa, apr));
            const int32_t dl = wadd(wsub(ml, lpl), apr), dr = wadd(wsub(mr, lpr), apl);
            ml = dl; mr = dr;
        }
        xch[i * kLanes] = ml;
        xch[(T + i) * kLanes] = mr;
    }
    if (cq == g.cpb - 1) {   // usb_audio.c:1279-1282
        const uint32_t p0 = (uint32_t)(uint16_t)(m.pk_l >> 13), p1 = (uint32_t)(uint16_t)(m.pk_r >> 13);
        if (m.pk_l > (1 << 28) + 268) m.clip |= 1u;
        if (m.pk_r > (1 << 28) + 268) m.clip |= 2u;
        uint32_t *gs = a.state + (size_t)wg * sm.n_slots * ROW + col;
        gs[(sm.peaks + 0) * ROW] = p0;
        gs[(sm.peaks + 1) * ROW] = p1;
        if (a.peaks) { uint16_t *pp = a.peaks + ((size_t)stream * g.n_blocks + kq) * sm.n_ch; pp[0] = (uint16_t)p0; pp[1] = (uint16_t)p1; }
    }
}

struct OutQ28 {
    uint32_t widx, loading, counter;
    float smooth;
    int32_t vmm;      // vol_mul_master, Q15
    uint32_t clip;
};

template <bool TAIL, class IMG>
__device__ __forceinline__ void output_item_q28(const KArgs &a, IMG img, const StateMap &sm, const Geo &g, OutQ28 &s,
                                                int32_t *__restrict__ lds_state, int32_t *__restrict__ lds_pk, const int32_t *__restrict__ xch_base,
                                                uint32_t wg, uint32_t lane, uint32_t stream, int o_first, int o_count, uint32_t kq, uint32_t cq, uint32_t q) {
    constexpr uint32_t ROW = make_state_map(0).row;
    const uint32_t col = lane;
    const uint32_t flags = img->flags;
    const int n = TAIL ? (int)min((uint32_t)T, g.B - cq * T) : T;
    const int N = sm.n_out;
    if (cq == 0) {
        // preset-mute envelope (usb_audio.c:466-498) and Q15 volumes (:975-980)
        bool act = s.loading != 0;
        if (act) { if (s.counter > g.B) s.counter -= g.B; else { s.counter = 0; s.loading = 0; } }
        float target = act ? 0.0f : 1.0f;
        float step = (float)g.B / (float)img->mute_transition;
        if (step > 1.0f) step = 1.0f;
        float gg = s.smooth;
        if (gg < target) { gg += step; if (gg > target) gg = target; }
        else if (gg > target) { gg -= step; if (gg < target) gg = target; }
        s.smooth = gg;
        int32_t mq = f2i_sat(gg * 32768.0f + 0.5f);
        if (mq < 0) mq = 0;
        if (mq > 32768) mq = 32768;
        const int32_t vol = q15mul(img->vol.i, mq);
        s.vmm = q15mul(vol, img->master.i);
    }
    const int32_t *xch = xch_base + (size_t)(q & 1u) * (2 * T * kLanes) + lane;
    const bool sub_active = flags & IF_SUB_ACTIVE;
    const size_t F = (size_t)g.n_blocks * g.B;
    const size_t frame0 = (size_t)kq * g.B + (size_t)cq * T;
    int32_t held[T];
#pragma unroll
    for (int i = 0; i < T; ++i) held[i] = 0;

#pragma unroll 1
    for (int j = 0; j < o_count; ++j) {
        const int o = o_first + j;
        const bool is_sub = (o == N - 1);
        const bool processed = !is_sub || sub_active;
        const bool enabled = (img->out_enabled >> o) & 1u;
        const bool muted = (img->out_mute >> o) & 1u;
        int32_t x[T];
        const int32_t dly = img->delay_samples[o];
        const bool dl_on = processed && (flags & IF_ANY_DELAY) && dly > 0;
        const bool dl_alias = dl_on && dly >= sm.max_delay;
        const bool dl_early = dl_on && !dl_alias && dly >= T;
        const uint32_t dmask = (uint32_t)sm.max_delay - 1u;
        uint32_t *line = a.dlines + ((size_t)wg * N + o) * (size_t)sm.max_delay * ROW + col;
        const uint32_t w0 = s.widx + cq * T;
        const uint32_t wb = w0 & dmask, rb = (w0 - (uint32_t)dly) & dmask;
        const bool w_flat = __all(wb + T <= (uint32_t)sm.max_delay), r_flat = __all(rb + T <= (uint32_t)sm.max_delay);
        int32_t dl[T];
        if (dl_early) {
            const uint32_t *rl = line + (size_t)rb * ROW;
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dl[i] = (int32_t)(r_flat ? rl[i * ROW] : line[(size_t)((rb + i) & dmask) * ROW]); }
        }
        // ---- PASS 4: matrix mix, Q15 crosspoints (usb_audio.c:1076-1100) ----
        const int32_t gl = img->mix[0][o].i, gr = img->mix[1][o].i;
        if (enabled && gl != 0 && gr != 0) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; x[i] = wadd(q15mul(xch[i * kLanes], gl), q15mul(xch[(T + i) * kLanes], gr)); }
        } else if (enabled && gl != 0) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; x[i] = q15mul(xch[i * kLanes], gl); }
        } else if (enabled && gr != 0) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; x[i] = q15mul(xch[(T + i) * kLanes], gr); }
        } else {
#pragma unroll
            for (int i = 0; i < T; ++i) x[i] = 0;
        }
        if (processed) {
            // ---- PASS 5: per-output EQ (gated by the master bypass on this flavour, usb_audio.c:1200) + Q15 gain ----
            if (enabled) {
                const int ch = 2 + o;
                if (!muted && !(flags & IF_BYPASS_MASTER_EQ) && !((img->ch_bypassed >> ch) & 1u))
                    run_bands_q28<TAIL, kBands>(x, n, &img->eq[ch][0], lds_state + (sm.eq + ch * kBands * 2) * kLanes + lane);
                const int32_t gain = muted ? 0 : f2i_sat(img->out_gain_lin[o] * (float)s.vmm);
#pragma unroll
                for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; const int32_t y = q15mul(x[i], gain); x[i] = (gain == 0) ? 0 : y; }
            }
            // ---- PASS 6: delay line (usb_audio.c:1216-1230) ----
            if (dl_early || dl_alias) {
                uint32_t *wl = line + (size_t)wb * ROW;
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (TAIL && i >= n) break;
                    if (w_flat) wl[i * ROW] = (uint32_t)x[i]; else line[(size_t)((wb + i) & dmask) * ROW] = (uint32_t)x[i];
                    if (dl_early) x[i] = dl[i];
                }
            } else if (dl_on) {
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (TAIL && i >= n) break;
                    const uint32_t w = (wb + i) & dmask;
                    line[(size_t)w * ROW] = (uint32_t)x[i];
                    x[i] = (int32_t)line[(size_t)((w - (uint32_t)dly) & dmask) * ROW];
                }
            }
        }
        // ---- PASS 7: peaks + output words (usb_audio.c:1232-1275) ----
        int32_t *pkp = lds_pk + (2 + o) * kLanes + lane;
        int32_t pk = (cq == 0) ? 0 : *pkp;
        const bool metered = !is_sub || (sub_active && enabled);
        if (metered) {
#pragma unroll
            for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; const int32_t av = wabs(x[i]); if (av > pk) pk = av; }
        }
        *pkp = pk;
        if (cq == g.cpb - 1) {
            const uint32_t p16 = metered ? (uint32_t)(uint16_t)(pk >> 13) : 0u;
            if (metered && pk > (1 << 28) + 268) s.clip |= 1u << (2 + o);
            a.state[((size_t)wg * sm.n_slots + sm.peaks + 2 + o) * ROW + col] = p16;
            if (a.peaks) a.peaks[((size_t)stream * g.n_blocks + kq) * sm.n_ch + 2 + o] = (uint16_t)p16;
        }
        if (is_sub) {
            if (a.sub && a.tiled_out) {
                int32_t *dst = a.sub + ((size_t)wg * F + frame0) * ROW + col;
                const bool live = sub_active && enabled;
#pragma unroll
                for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[(size_t)i * ROW] = live ? x[i] : 0; }
            } else if (a.sub) {
                int32_t *dst = a.sub + (size_t)stream * F + frame0;
                const bool live = sub_active && enabled;
#pragma unroll
                for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[i] = live ? x[i] : 0; }
            }
        } else {
            // a pair whose two outputs are both disabled is zero-filled whatever its delay lines still hold (usb_audio.c:1247-1250)
            const bool pair_dead = !(((img->out_enabled >> o) | (img->out_enabled >> (o ^ 1))) & 1u);
            int32_t wv[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (TAIL && i >= n) break;
                int32_t v = wadd(x[i], 1 << 5) >> 6;                                      // (x + 32) >> 6
                v = v > 0x7FFFFF ? 0x7FFFFF : (v < -0x800000 ? -0x800000 : v);            // clip_s24
                wv[i] = pair_dead ? 0 : v;
                if (a.i2s_slots && ((img->i2s_pairs >> (o >> 1)) & 1u)) wv[i] = (int32_t)((uint32_t)wv[i] << 8);      // I2S slot (audio_i2s_multi.c:217-226)
            }
            const int pair = o >> 1, side = o & 1;
            const bool partner_here = side ? (j >= 1) : (j + 1 < o_count && o + 1 < N - 1);
            if (a.pairs && a.tiled_out) {
                int32_t *dst = a.pairs + (((size_t)wg * (N - 1) + o) * F + frame0) * ROW + col;
#pragma unroll
                for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[(size_t)i * ROW] = wv[i]; }
            } else if (a.pairs) {
                int32_t *dst = a.pairs + (((size_t)(stream - a.pairs_stream0) * sm.n_pairs + pair) * F + frame0) * 2;
                if (side == 1 && partner_here) {
#pragma unroll
                    for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[i * 2] = held[i]; dst[i * 2 + 1] = wv[i]; }
                } else if (!(side == 0 && partner_here)) {
#pragma unroll
                    for (int i = 0; i < T; ++i) { if (TAIL && i >= n) break; dst[i * 2 + side] = wv[i]; }
                }
            }
            if (side == 0) {
#pragma unroll
                for (int i = 0; i < T; ++i) held[i] = wv[i];
            }
        }
    }
    if (cq == g.cpb - 1 && (flags & IF_ANY_DELAY)) s.widx = (s.widx + g.B) & ((uint32_t)sm.max_delay - 1u);
}

#include "dspi_chain_q28_lat.inc"
#include "dspi_chain_pk.inc"
#include "dspi_chain_skew.inc"
#include "dspi_chain_skew_lev.inc"

// ==========================================================================================
// the one-stream-per-lane kernel (Q28 flavour; float flavour: lanes whose two streams differ in image)
// ==========================================================================================
// PL (Q28 only; the float instantiation is always per-lane): per-lane parameter images for rows with several presets
// NW (Q28 only): waves per workgroup.  4: the roles below, two workgroups per CU — the layout for launches that fill the chip.
// 7: one output per wave and the right channel's pass 1 on a wave of its own (16 / 12 / 10 x 5 band visits per chunk instead of
// 16 / 22 / 22 / 24), one workgroup per CU — for launches of at most one workgroup per CU (BASELINE config 5: 16 384 streams = 256
// workgroups), where four waves leave every SIMD with ONE wave and nothing to hide its dependent-issue latency behind.
template <int FLAVOR, bool TAIL, bool PL = false, bool FMA = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void chain_kernel(KArgs a) {
    constexpr StateMap sm = make_state_map(FLAVOR);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    float *lds_state = reinterpret_cast<float *>(lds);                      // [lds_slots][64]
    float *lds_pk = lds_state + sm.lds_slots * kLanes;                      // [n_ch][64]
    float *xch = lds_pk + sm.n_ch * kLanes;                                 // [2][2][T][64]

    const WgItem item = a.items[blockIdx.x];
    const uint32_t wg = item.wg;
    const uint32_t lane = threadIdx.x & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the float flavour keeps two streams per lane column pair (packed kernel); this scalar kernel then serves one
    // component (item.image: 0 = first, 1 = second stream of the lane; per-lane parameter mode has no use for an image
    // index) of the lanes whose two streams carry different parameter images — both components in one launch
    constexpr uint32_t ROW = sm.row;
    const uint32_t col = FLAVOR ? lane * 2 + (item.image & 1u) : lane;
    const uint32_t stream = wg * ROW + col;
    // Lanes that are not part of this launch (streams of another parameter image, or padding past n_streams; the
    // host never sets those mask bits) are switched off ONCE by narrowing EXEC for the whole kernel: every vector
    // instruction below — loads, stores, LDS traffic — is then masked for free, instead of wrapping each memory
    // access in its own saveexec/branch pair.  All four waves of a workgroup share the mask (never zero), so every
    // wave still reaches every barrier.
    asm volatile("s_mov_b64 exec, %0" ::"s"(item.mask) : "memory");
    constexpr bool active = true;
    ImgPtr img = to_const(a.img + item.image);                                   // Q28: the workgroup's image (scalar loads)
    const DevImage *img_l = a.img + ((FLAVOR || PL) ? a.stream_image[stream] : 0u);   // float / Q28 PL: this lane's own image (vector loads)

    uint32_t *gs = a.state + (size_t)wg * sm.n_slots * ROW + col;
    for (int s = wave; s < sm.lds_slots; s += NW) lds[s * kLanes + lane] = gs[(size_t)s * ROW];
    __syncthreads();

    Geo g;
    g.n_blocks = a.n_blocks; g.B = a.block_len;
    g.cpb = (g.B + T - 1) / T;
    g.items = g.n_blocks * g.cpb;
    // float (per-lane images): the schedule cannot depend on one lane's flags, every lane goes through the ring
    // Q28: every sample goes through the ring too (the two master waves meet there), never sooner than two steps
    g.lag = FLAVOR ? g.cpb : (g.cpb > 2u ? g.cpb : 2u);
    g.steps = g.items + g.lag + 1;

    int32_t *q_mail = reinterpret_cast<int32_t *>(xch) + 2 * 2 * T * kLanes;      // [kQ28Mail][64] queued gain decisions (role 0)
    int32_t *q_envr = q_mail + kQ28Mail * kLanes;                                 // [2][64] right envelope at packet end (role 3 -> role 0)
    // Q28 roles: 0 = left pass 1 + hand-off (the lightest, ~16 band-visit equivalents per chunk), 1 / 2 = two outputs each (22),
    // 3 = right pass 1 + the fifth output (24).  Two workgroups share a CU; the four waves of a workgroup always land on four
    // different SIMDs (in no fixed order) and all of them in the same wave slot, slot 0 for the first workgroup of the CU and
    // slot 1 for the second (tools/probe/probe7).  So the role is taken from the SIMD id, in reverse order in the odd slot: every
    // SIMD then carries roles r and 3 - r (40 / 44 / 44 / 40) instead of the same role twice (32 / 44 / 44 / 48).  Any
    // assignment is correct; should the placement ever not give four different roles, the wave index is used.
    int role = wave;
    if (FLAVOR == 0 && NW == 7) {
        // seven waves: three SIMDs carry two, one carries one (when the placement is the usual round robin) — the lone wave takes
        // role 0 (16), the others the right channel's pass 1 (role 6, 12) and the five outputs (roles 1..5, 10 each) in wave order
        uint32_t *role_tab = reinterpret_cast<uint32_t *>(q_envr + 2 * kLanes);     // [7]
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (5 << 11));
        if (lane == (uint32_t)__builtin_ctzll(item.mask)) role_tab[wave] = (hw >> 4) & 3u;
        __syncthreads();
        uint32_t tab[7];
#pragma unroll
        for (int w = 0; w < 7; ++w) tab[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)role_tab[w]) & 3u;
        int lone = 0;
#pragma unroll
        for (int w = 6; w >= 0; --w) {
            int same = 0;
#pragma unroll
            for (int v = 0; v < 7; ++v) same += (tab[v] == tab[w]);
            if (same == 1) lone = w;
        }
        int rank = 0;
#pragma unroll
        for (int w = 0; w < 7; ++w) if (w < wave && w != lone) ++rank;
        role = __builtin_amdgcn_readfirstlane(wave == lone ? 0 : (rank == 0 ? 6 : rank));
    } else if (FLAVOR == 0) {
        uint32_t *role_tab = reinterpret_cast<uint32_t *>(q_envr + 2 * kLanes);     // [4]
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (5 << 11));      // hwreg(HW_REG_HW_ID, 0, 6): wave slot [3:0], SIMD [5:4]
        const uint32_t simd = (hw >> 4) & 3u;
        const uint32_t mine = (hw & 1u) ? 3u - simd : simd;
        if (lane == (uint32_t)__builtin_ctzll(item.mask)) role_tab[wave] = mine;
        __syncthreads();
        const uint32_t seen = (1u << role_tab[0]) | (1u << role_tab[1]) | (1u << role_tab[2]) | (1u << role_tab[3]);
        role = __builtin_amdgcn_readfirstlane(seen == 15u ? (int)mine : wave);
    }
    if (role == 0 && FLAVOR == 0) {
        // ---- role 0: pass 1 of the left channel + hand-off of both ----
        int32_t *qstate = reinterpret_cast<int32_t *>(lds);
        int32_t *qxch = reinterpret_cast<int32_t *>(xch);
        MasterQ28 m;
        m.lpL = (int32_t)gs[(sm.xfeed + 0) * ROW]; m.lpR = (int32_t)gs[(sm.xfeed + 1) * ROW];
        m.apL = (int32_t)gs[(sm.xfeed + 2) * ROW]; m.apR = (int32_t)gs[(sm.xfeed + 3) * ROW];
        m.env_l = (int32_t)gs[(sm.lev + 0) * ROW]; m.env_r = 0;
        m.gsm_db = as_f(gs[(sm.lev + 2) * ROW]); m.g_cur = (int32_t)gs[(sm.lev + 3) * ROW]; m.g_prev = (int32_t)gs[(sm.lev + 4) * ROW];
        m.g_last = m.g_cur;
        m.rp1 = m.rp2 = gs[sm.ring_pos * ROW] & (kRingLen - 1);
        m.p2_base = 1 << 28; m.p2_D = m.p2_R = m.p2_acc = m.p2_carry = 0; m.pk_l = m.pk_r = 0;
        m.clip = gs[(sm.clip + 0) * ROW];
        uint32_t k1 = 0, c1 = 0, kq = 0, cq = 0;
        int32_t env_end = 0;
        bool decide = false;
        uint32_t kd = 0;
        WT_DECL;
        for (uint32_t st = 0; st < g.steps; ++st) {
            if (decide) {      // gain decision of packet kd (leveller.c:304-334): pass 1 of both channels ended in the previous step
                const bool lev_on = (PL ? img_l->flags : img->flags) & IF_LEVELLER_ON;
                if (lev_on) {
                    const float inv = 1.0f / (float)(1 << 28);
                    const float el = (float)env_end * inv, er = (float)q_envr[(kd & 1u) * kLanes + lane] * inv;
                    const float gl = PL ? leveller_block_gain(img_l, m.gsm_db, el > er ? el : er, g.B) : leveller_block_gain(img, m.gsm_db, el > er ? el : er, g.B);
                    m.g_prev = m.g_cur;
                    m.g_cur = f2i_sat(gl * (float)(1 << 28));
                }
                q_mail[(kd & (kQ28Mail - 1)) * kLanes + lane] = m.g_cur;
                decide = false;
            }
            const bool do_p1 = st < g.items;
            const bool do_item = st >= g.lag && st < g.items + g.lag;
            const uint32_t q = st - g.lag;
            if (do_p1) {
                // kernels for ragged packets: full chunks still take the exit-free instantiation (the sample arrays are local
                // to these functions, so the two variants meet in scalars only)
                const bool ragged = TAIL && (c1 + 1) * T > g.B;
                if (ragged) {
                    if (PL) master_p1_q28<true, 0>(a, img_l, sm, g, m.env_l, m.rp1, qstate, wg, lane, stream, k1, c1);
                    else master_p1_q28<true, 0>(a, img, sm, g, m.env_l, m.rp1, qstate, wg, lane, stream, k1, c1);
                } else {
                    if (PL) master_p1_q28<false, 0>(a, img_l, sm, g, m.env_l, m.rp1, qstate, wg, lane, stream, k1, c1);
                    else master_p1_q28<false, 0>(a, img, sm, g, m.env_l, m.rp1, qstate, wg, lane, stream, k1, c1);
                }
                if (++c1 == g.cpb) { env_end = m.env_l; decide = true; kd = k1; c1 = 0; ++k1; }
            }
            if (do_item) {
                if (TAIL && (cq + 1) * T > g.B) {
                    if (PL) master_item_q28<true>(a, img_l, sm, g, m, q_mail, qxch, wg, lane, stream, kq, cq, q);
                    else master_item_q28<true>(a, img, sm, g, m, q_mail, qxch, wg, lane, stream, kq, cq, q);
                } else {
                    if (PL) master_item_q28<false>(a, img_l, sm, g, m, q_mail, qxch, wg, lane, stream, kq, cq, q);
                    else master_item_q28<false>(a, img, sm, g, m, q_mail, qxch, wg, lane, stream, kq, cq, q);
                }
                if (++cq == g.cpb) { cq = 0; ++kq; }
            }
            WT_BEFORE_BARRIER;
            lds_barrier();
            WT_AFTER_BARRIER;
        }
        WT_FINISH(0);
        gs[(sm.xfeed + 0) * ROW] = (uint32_t)m.lpL; gs[(sm.xfeed + 1) * ROW] = (uint32_t)m.lpR;
        gs[(sm.xfeed + 2) * ROW] = (uint32_t)m.apL; gs[(sm.xfeed + 3) * ROW] = (uint32_t)m.apR;
        gs[(sm.lev + 0) * ROW] = (uint32_t)m.env_l;
        gs[(sm.lev + 2) * ROW] = as_u(m.gsm_db); gs[(sm.lev + 3) * ROW] = (uint32_t)m.g_cur; gs[(sm.lev + 4) * ROW] = (uint32_t)m.g_prev;
        gs[sm.ring_pos * ROW] = m.rp1;
        gs[(sm.clip + 0) * ROW] = m.clip;
    } else if (FLAVOR == 0) {
        int32_t *qstate = reinterpret_cast<int32_t *>(lds);
        int32_t *qpk = reinterpret_cast<int32_t *>(lds_pk);
        const int32_t *qxch = reinterpret_cast<const int32_t *>(xch);
        // 5 outputs over waves 1..3: pair 0, pair 1, sub; wave 3 also runs pass 1 of the right channel (role 3 above)
        const int o_first = NW == 7 ? role - 1 : (role - 1) * 2;
        const int o_count = NW == 7 ? (role <= sm.n_out ? 1 : 0) : ((role <= sm.n_pairs) ? 2 : 1);
        const bool right = NW == 7 ? (role == 6) : (role == 3);
        OutQ28 s;
        s.widx = gs[sm.widx * ROW];
        s.loading = gs[(sm.mute + 0) * ROW]; s.counter = gs[(sm.mute + 1) * ROW]; s.smooth = as_f(gs[(sm.mute + 2) * ROW]);
        s.vmm = 0;
        s.clip = NW == 7 ? 0u : gs[(sm.clip + role) * ROW];      // seven waves share the four sticky words: new bits are ORed in at the end
        int32_t env_r = right ? (int32_t)gs[(sm.lev + 1) * ROW] : 0;
        uint32_t rp1 = gs[sm.ring_pos * ROW] & (kRingLen - 1);
        uint32_t kq = 0, cq = 0, k1 = 0, c1 = 0;
        WT_DECL;
        for (uint32_t st = 0; st < g.steps; ++st) {
            if (right && st < g.items) {
                if (TAIL && (c1 + 1) * T > g.B) {
                    if (PL) master_p1_q28<true, 1>(a, img_l, sm, g, env_r, rp1, qstate, wg, lane, stream, k1, c1);
                    else master_p1_q28<true, 1>(a, img, sm, g, env_r, rp1, qstate, wg, lane, stream, k1, c1);
                } else {
                    if (PL) master_p1_q28<false, 1>(a, img_l, sm, g, env_r, rp1, qstate, wg, lane, stream, k1, c1);
                    else master_p1_q28<false, 1>(a, img, sm, g, env_r, rp1, qstate, wg, lane, stream, k1, c1);
                }
                if (++c1 == g.cpb) { q_envr[(k1 & 1u) * kLanes + lane] = env_r; c1 = 0; ++k1; }
            } else if (right) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last chunks' ring rows (master_p1_q28 drains one step late)
            }
            if (o_count && st >= g.lag + 1) {
                const uint32_t q = st - g.lag - 1;
                if (TAIL && (cq + 1) * T > g.B) {
                    if (PL) output_item_q28<true>(a, img_l, sm, g, s, qstate, qpk, qxch, wg, lane, stream, o_first, o_count, kq, cq, q);
                    else output_item_q28<true>(a, img, sm, g, s, qstate, qpk, qxch, wg, lane, stream, o_first, o_count, kq, cq, q);
                } else {
                    if (PL) output_item_q28<false>(a, img_l, sm, g, s, qstate, qpk, qxch, wg, lane, stream, o_first, o_count, kq, cq, q);
                    else output_item_q28<false>(a, img, sm, g, s, qstate, qpk, qxch, wg, lane, stream, o_first, o_count, kq, cq, q);
                }
                if (++cq == g.cpb) { cq = 0; ++kq; }
            }
            WT_BEFORE_BARRIER;
            lds_barrier();
            WT_AFTER_BARRIER;
        }
        WT_FINISH(role);
        if (role == 1) {
            gs[sm.widx * ROW] = s.widx;
            gs[(sm.mute + 0) * ROW] = s.loading; gs[(sm.mute + 1) * ROW] = s.counter; gs[(sm.mute + 2) * ROW] = as_u(s.smooth);
        }
        if (right) gs[(sm.lev + 1) * ROW] = (uint32_t)env_r;
        if (NW == 7) { if (s.clip) atomicOr(gs + (size_t)(sm.clip + 1 + (role % 3)) * ROW, s.clip); }
        else gs[(sm.clip + role) * ROW] = s.clip;
    } else if (wave == 0) {
        MasterF32 m;
        m.lpL = as_f(gs[(sm.xfeed + 0) * ROW]); m.lpR = as_f(gs[(sm.xfeed + 1) * ROW]);
        m.apL = as_f(gs[(sm.xfeed + 2) * ROW]); m.apR = as_f(gs[(sm.xfeed + 3) * ROW]);
        m.env_l = as_f(gs[(sm.lev + 0) * ROW]); m.env_r = as_f(gs[(sm.lev + 1) * ROW]);
        m.gsm_db = as_f(gs[(sm.lev + 2) * ROW]); m.g_cur = as_f(gs[(sm.lev + 3) * ROW]); m.g_prev = as_f(gs[(sm.lev + 4) * ROW]);
        m.rp1 = m.rp2 = gs[sm.ring_pos * ROW] & (kRingLen - 1);
        m.p2_gain = 1.0f; m.p2_step = 0.0f; m.pk_l = m.pk_r = 0.0f;
        m.pre_valid = 0;
#pragma unroll
        for (int v = 0; v < T / 4; ++v) m.pre[v] = u32x4{0, 0, 0, 0};
        m.clip = gs[(sm.clip + 0) * ROW];
        uint32_t k1 = 0, c1 = 0, kq = 0, cq = 0;
        WT_DECL;
        for (uint32_t st = 0; st < g.steps; ++st) {
            const bool do_p1 = st < g.items;
            const bool do_item = st >= g.lag && st < g.items + g.lag;
            const uint32_t q = st - g.lag;
            if (do_p1 || do_item) {
                // (ragged packets: full chunks take the exit-free instantiation; lag is a whole packet, so both halves of a
                // step sit at the same chunk of their packets)
                if (TAIL && ((do_p1 ? c1 : cq) + 1) * T > g.B)
                    master_step_f32<true, FMA>(a, img_l, sm, g, m, lds_state, xch, wg, lane, col, stream, active, do_p1, k1, c1, do_item, kq, cq, q);
                else
                    master_step_f32<false, FMA>(a, img_l, sm, g, m, lds_state, xch, wg, lane, col, stream, active, do_p1, k1, c1, do_item, kq, cq, q);
            }
            if (do_p1) { if (++c1 == g.cpb) { c1 = 0; ++k1; } }
            if (do_item) { if (++cq == g.cpb) { cq = 0; ++kq; } }
            WT_BEFORE_BARRIER;
            lds_barrier();
            WT_AFTER_BARRIER;
        }
        WT_FINISH(0);
        if (active) {
            gs[(sm.xfeed + 0) * ROW] = as_u(m.lpL); gs[(sm.xfeed + 1) * ROW] = as_u(m.lpR);
            gs[(sm.xfeed + 2) * ROW] = as_u(m.apL); gs[(sm.xfeed + 3) * ROW] = as_u(m.apR);
            gs[(sm.lev + 0) * ROW] = as_u(m.env_l); gs[(sm.lev + 1) * ROW] = as_u(m.env_r);
            gs[(sm.lev + 2) * ROW] = as_u(m.gsm_db); gs[(sm.lev + 3) * ROW] = as_u(m.g_cur); gs[(sm.lev + 4) * ROW] = as_u(m.g_prev);
            gs[sm.ring_pos * ROW] = m.rp1;
            gs[(sm.clip + 0) * ROW] = m.clip;
        }
    } else {
        // outputs split 3/3/3 (float) or 2/2/1 (Q28) over waves 1..3
        const int per = (sm.n_out + 2) / 3;
        const int o_first = (wave - 1) * per;
        int o_count = sm.n_out - o_first;
        if (o_count > per) o_count = per;
        if (o_count < 0) o_count = 0;
        OutF32 s;
        s.widx = gs[sm.widx * ROW];
        s.loading = gs[(sm.mute + 0) * ROW]; s.counter = gs[(sm.mute + 1) * ROW]; s.smooth = as_f(gs[(sm.mute + 2) * ROW]);
        s.vmm = 0.0f;
        s.clip = gs[(sm.clip + wave) * ROW];
        uint32_t kq = 0, cq = 0;
        WT_DECL;
        for (uint32_t st = 0; st < g.steps; ++st) {
            if (st >= g.lag + 1) {
                const uint32_t q = st - g.lag - 1;
                if (TAIL && (cq + 1) * T > g.B)
                    output_item_f32<true, FMA>(a, img_l, sm, g, s, lds_state, lds_pk, xch, wg, lane, col, stream, active, o_first, o_count, kq, cq, q);
                else
                    output_item_f32<false, FMA>(a, img_l, sm, g, s, lds_state, lds_pk, xch, wg, lane, col, stream, active, o_first, o_count, kq, cq, q);
                if (++cq == g.cpb) { cq = 0; ++kq; }
            }
            WT_BEFORE_BARRIER;
            lds_barrier();
            WT_AFTER_BARRIER;
        }
        WT_FINISH(wave);
        if (active) {
            if (wave == 1) {
                gs[sm.widx * ROW] = s.widx;
                gs[(sm.mute + 0) * ROW] = s.loading; gs[(sm.mute + 1) * ROW] = s.counter; gs[(sm.mute + 2) * ROW] = as_u(s.smooth);
            }
            gs[(sm.clip + wave) * ROW] = s.clip;
        }
    }
    __syncthreads();
    if (active)
        for (int s = wave; s < sm.lds_slots; s += NW) gs[(size_t)s * ROW] = lds[s * kLanes + lane];
}

// ------------------------------------------------------------------------------------------
// state maintenance: the per-stream side effects of parameter changes (StateOps)
// ------------------------------------------------------------------------------------------
template <int FLAVOR>
__global__ void state_ops_kernel(const WgItem *items, StateOps ops, uint32_t *state, uint32_t *dlines, uint32_t *ring, uint32_t n_streams) {
    constexpr StateMap sm = make_state_map(FLAVOR);
    constexpr uint32_t ROW = sm.row;
    const WgItem item = items[blockIdx.x];
    // one thread column per stream of the workgroup (64 or 128), `parts` row-slices for the bulk zeroing
    const uint32_t col = threadIdx.x % ROW, part = threadIdx.x / ROW, parts = blockDim.x / ROW;
    const uint32_t lane = FLAVOR ? col >> 1 : col;
    const uint64_t m = (FLAVOR && (col & 1u)) ? item.mask1 : item.mask;
    const uint32_t stream = item.wg * ROW + col;
    if (!(((m >> lane) & 1ull) && stream < n_streams)) return;
    uint32_t *gs = state + (size_t)item.wg * sm.n_slots * ROW + col;
    if (part == 0) {
        for (int ch = 0; ch < sm.n_ch; ++ch)
            for (int b = 0; b < kBands; ++b)
                if (ops.reset_all_eq || ((ops.reset_band[ch] >> b) & 1u)) {
                    gs[(size_t)(sm.eq + (ch * kBands + b) * 2) * ROW] = 0;
                    gs[(size_t)(sm.eq + (ch * kBands + b) * 2 + 1) * ROW] = 0;
                }
        if (ops.reset_crossfeed) for (int i = 0; i < 4; ++i) gs[(size_t)(sm.xfeed + i) * ROW] = 0;
        if (ops.reset_leveller) {   // leveller_reset_state: zero everything, unity gains
            const uint32_t unity = FLAVOR ? 0x3f800000u : (1u << 28);
            gs[(size_t)(sm.lev + 0) * ROW] = 0; gs[(size_t)(sm.lev + 1) * ROW] = 0; gs[(size_t)(sm.lev + 2) * ROW] = 0;
            gs[(size_t)(sm.lev + 3) * ROW] = unity; gs[(size_t)(sm.lev + 4) * ROW] = unity;
            gs[(size_t)sm.ring_pos * ROW] = 0;
        }
        if (ops.mute_start) { gs[(size_t)(sm.mute + 0) * ROW] = 1; gs[(size_t)(sm.mute + 1) * ROW] = ops.mute_samples; }
        if (ops.mute_cancel) gs[(size_t)(sm.mute + 0) * ROW] = 0;
        if (ops.clear_clips) for (int i = 0; i < 4; ++i) gs[(size_t)(sm.clip + i) * ROW] = 0;
    }
    if (ops.reset_leveller) {
        uint32_t *r = ring + (size_t)item.wg * kRingLen * 2 * ROW + col;
        for (uint32_t p = part; p < (uint32_t)kRingLen * 2; p += parts) r[(size_t)p * ROW] = 0;
    }
    if (ops.zero_delay_lines) {
        uint32_t *d = dlines + (size_t)item.wg * sm.n_out * (size_t)sm.max_delay * ROW + col;
        const uint32_t total = (uint32_t)sm.n_out * (uint32_t)sm.max_delay;
        for (uint32_t p = part; p < total; p += parts) d[(size_t)p * ROW] = 0;
    }
}

// power-on state of a stream: everything zero except unity leveller gains and mute envelope = 1.0
template <int FLAVOR>
__global__ void state_init_kernel(uint32_t *state, uint32_t n_wg) {
    constexpr StateMap sm = make_state_map(FLAVOR);
    constexpr uint32_t ROW = sm.row;
    const uint32_t wg = blockIdx.x, col = threadIdx.x;
    if (wg >= n_wg || col >= ROW) return;
    uint32_t *gs = state + (size_t)wg * sm.n_slots * ROW + col;
    const uint32_t unity = FLAVOR ? 0x3f800000u : (1u << 28);
    gs[(size_t)(sm.lev + 3) * ROW] = unity;
    gs[(size_t)(sm.lev + 4) * ROW] = unity;
    gs[(size_t)(sm.mute + 2) * ROW] = 0x3f800000u;   // preset_mute_smooth_gain = 1.0f (usb_audio.c:457)
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
// Parallel build (Makefile): this file is compiled once per DSPI_PART.  Part 0 holds everything except the packed float kernels;
// parts 1..6 hold one (FMA, PV, PVB) family of chain_kernel_pk each (16 kernels: TAIL x LEV x PCM24 x TILED).  Every part is its own
// code object inside the shared library.  Without DSPI_PART the file is one translation unit (the timing build).
#if !defined(DSPI_PART) || DSPI_PART == 0
#define DSPI_PART_MAIN 1
#endif

#ifdef DSPI_PART_MAIN
size_t chain_lds_bytes(int flavor, int packed) {
    const StateMap sm = make_state_map(flavor);
    // Q28 one-stream kernel: + queued gain decisions, the posted right-channel envelope and the role table (kQ28Mail + 2 + 1 rows)
    return (size_t)(sm.lds_slots + sm.n_ch + 2 * 2 * T + (packed ? kMailbox + 4 : (flavor ? 0 : 7))) * (packed ? ROWP : (uint32_t)kLanes) * sizeof(uint32_t) + (packed ? 16 : 0);      // packed: + role table (16 B) and the posted left / right envelopes (2 x 2 rows)
}

#endif

constexpr int kMaxDevices = 65;      // slot 64: any device index beyond (attribute set on every launch)

#ifdef DSPI_PART_MAIN
// Q28 launches of at most one workgroup per CU take the seven-wave layout (chain_kernel: NW)
template <bool PL>
static hipError_t launch_chain_q28_7(const KArgs &args, uint32_t n_items, hipStream_t stream) {
    const size_t lds = chain_lds_bytes(0, 0);
    static bool attr_set[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = kMaxDevices - 1;
    if (!attr_set[dev] || dev == kMaxDevices - 1) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_kernel<0, false, PL, false, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_kernel<0, true, PL, false, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    if (args.block_len % T) hipLaunchKernelGGL((chain_kernel<0, true, PL, false, 7>), dim3(n_items), dim3(64 * 7), lds, stream, args);
    else hipLaunchKernelGGL((chain_kernel<0, false, PL, false, 7>), dim3(n_items), dim3(64 * 7), lds, stream, args);
    return hipGetLastError();
}
// DSPI_Q28_WAVES=4 / 7 forces one wave layout of chain_kernel (tests, development) and keeps the latency layout out; any other value is ignored
static int q28_forced_waves() {
    if (const char *e = getenv("DSPI_Q28_WAVES")) { const int w = atoi(e); if (w == 7 || w == 4) return w; }
    return 0;
}
static uint32_t q28_seven_wave_limit() {      // work items up to which the seven-wave layout is used: one per CU
    if (const int w = q28_forced_waves()) return w == 7 ? 0xffffffffu : 0u;
    static int cus[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices - 1) return 256u;
    if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus[dev] = 256;
    return (uint32_t)cus[dev];
}

// Q28 contexts small enough to leave the chip underfilled take the latency layout (dspi_chain_q28_lat.inc: one stream per workgroup, one
// lane per (channel, stage)): up to eight of its two-wave workgroups per CU — 2 048 streams on 256 CUs, where it is still ahead for one-packet
// and for 50-packet calls alike; at 4 096 chain_kernel wins (tools/bench_q28_layouts.py, profiles/r05_q28_layouts.jsonl).
// DSPI_Q28_LAYOUT=lat|chain forces one (tests, development).
static uint32_t q28_latency_limit() {
    if (const char *e = getenv("DSPI_Q28_LAYOUT")) { if (!strcmp(e, "lat")) return 0xffffffffu; if (!strcmp(e, "chain")) return 0u; }
    return 8u * q28_seven_wave_limit();
}
template <int FLAVOR, bool PL, bool FMA = false>
static hipError_t launch_chain_t(const KArgs &args, uint32_t n_items, hipStream_t stream) {
    if constexpr (FLAVOR == 0) {
        if (args.n_streams <= q28_latency_limit() && q28_forced_waves() == 0) {
            hipLaunchKernelGGL((chain_kernel_q28_lat<PL>), dim3(n_items * 64u), dim3(128), 0, stream, args);
            return hipGetLastError();
        }
    }
    if constexpr (FLAVOR == 0) { if (n_items <= q28_seven_wave_limit()) return launch_chain_q28_7<PL>(args, n_items, stream); }
    const size_t lds = chain_lds_bytes(FLAVOR, 0);
    // the dynamic-LDS limit is a per-device function attribute: remember it per device (contexts on several GPUs may
    // share one process; a context is single-threaded)
    static bool attr_set[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = kMaxDevices - 1;
    if (!attr_set[dev] || dev == kMaxDevices - 1) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_kernel<FLAVOR, false, PL, FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_kernel<FLAVOR, true, PL, FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    if (args.block_len % T) hipLaunchKernelGGL((chain_kernel<FLAVOR, true, PL, FMA>), dim3(n_items), dim3(256), lds, stream, args);
    else hipLaunchKernelGGL((chain_kernel<FLAVOR, false, PL, FMA>), dim3(n_items), dim3(256), lds, stream, args);
    return hipGetLastError();
}

#endif  // DSPI_PART_MAIN

template <bool TAIL, bool LEV, bool PCM24, bool TILED, bool FMA, bool PV, bool PVB>
static hipError_t launch_chain_pk_t(const KArgs &args, uint32_t n_items, hipStream_t stream) {
    const size_t lds = chain_lds_bytes(1, 1);
    static bool attr_set[kMaxDevices] = {};      // per device, see launch_chain_t
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = kMaxDevices - 1;
    if (!attr_set[dev] || dev == kMaxDevices - 1) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_kernel_pk<TAIL, LEV, PCM24, TILED, FMA, PV, PVB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((chain_kernel_pk<TAIL, LEV, PCM24, TILED, FMA, PV, PVB>), dim3(n_items), dim3(64 * kPkWaves), lds, stream, args);
    return hipGetLastError();
}

template <bool TAIL, bool LEV, bool FMA, bool PV, bool PVB>
static hipError_t launch_chain_pk_2(const KArgs &args, uint32_t n_items, hipStream_t stream) {
    const bool p24 = args.bit_depth == 24, tl = args.tiled_out != 0;
    if (p24) return tl ? launch_chain_pk_t<TAIL, LEV, true, true, FMA, PV, PVB>(args, n_items, stream) : launch_chain_pk_t<TAIL, LEV, true, false, FMA, PV, PVB>(args, n_items, stream);
    return tl ? launch_chain_pk_t<TAIL, LEV, false, true, FMA, PV, PVB>(args, n_items, stream) : launch_chain_pk_t<TAIL, LEV, false, false, FMA, PV, PVB>(args, n_items, stream);
}

template <bool FMA, bool PV, bool PVB>
static hipError_t launch_chain_pk(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream) {
    const bool tail = (args.block_len % T) != 0;
    if (tail) return leveller_on ? launch_chain_pk_2<true, true, FMA, PV, PVB>(args, n_items, stream) : launch_chain_pk_2<true, false, FMA, PV, PVB>(args, n_items, stream);
    return leveller_on ? launch_chain_pk_2<false, true, FMA, PV, PVB>(args, n_items, stream) : launch_chain_pk_2<false, false, FMA, PV, PVB>(args, n_items, stream);
}


// one exported launcher per kernel family: family = (FMA ? 3 : 0) + (PVB ? 2 : PV ? 1 : 0)
hipError_t launch_chain_pk_f0(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream);
hipError_t launch_chain_pk_f1(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream);
hipError_t launch_chain_pk_f2(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream);
hipError_t launch_chain_pk_f3(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream);
hipError_t launch_chain_pk_f4(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream);
hipError_t launch_chain_pk_f5(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream);
#define DSPI_PK_FAMILY(N, FMA, PV, PVB) \
    hipError_t launch_chain_pk_f##N(const KArgs &args, bool leveller_on, uint32_t n_items, hipStream_t stream) { return launch_chain_pk<FMA, PV, PVB>(args, leveller_on, n_items, stream); }
#if !defined(DSPI_PART) || DSPI_PART == 1
DSPI_PK_FAMILY(0, false, false, false)
#endif
#if !defined(DSPI_PART) || DSPI_PART == 2
DSPI_PK_FAMILY(1, false, true, false)
#endif
#if !defined(DSPI_PART) || DSPI_PART == 3
DSPI_PK_FAMILY(2, false, true, true)
#endif
#if !defined(DSPI_PART) || DSPI_PART == 4
DSPI_PK_FAMILY(3, true, false, false)
#endif
#if !defined(DSPI_PART) || DSPI_PART == 5
DSPI_PK_FAMILY(4, true, true, false)
#endif
#if !defined(DSPI_PART) || DSPI_PART == 6
DSPI_PK_FAMILY(5, true, true, true)
#endif

// the latency layout of the float chain (dspi_chain_skew.inc): part 7
hipError_t launch_chain_skew(const KArgs &args, uint32_t n_items, int shape, hipStream_t stream);
hipError_t launch_chain_skew_pp(const KArgs &args, uint32_t n_items, int shape, hipStream_t stream);
#if !defined(DSPI_PART) || DSPI_PART == 7 || DSPI_PART == 8
// one launcher for the instances of a shape: float contract x input word size, with or without the S/PDIF encoder in the output waves
template <class F>
static hipError_t sk_launch(const KArgs &args, dim3 grid, dim3 block, size_t lds, hipStream_t stream, bool (&attr_set)[kMaxDevices], F kernels) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = kMaxDevices - 1;
    if (!attr_set[dev] || dev == kMaxDevices - 1) {      // per device, see launch_chain_t
        for (int i = 0; i < 8; i++) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernels[i]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        attr_set[dev] = true;
    }
    const int idx = (args.fma ? 4 : 0) | (args.bit_depth == 24 ? 2 : 0) | (args.spdif ? 1 : 0);
    hipLaunchKernelGGL(kernels[idx], grid, block, lds, stream, args);
    return hipGetLastError();
}
typedef void (*SkKernel)(KArgs);
template <bool EQO, bool PP>
static hipError_t launch_chain_skew_t(const KArgs &args, uint32_t n_items, hipStream_t stream) {
    static bool attr_set[kMaxDevices] = {};
    static const SkKernel k[8] = {chain_kernel_skew<false, false, EQO, false, PP>, chain_kernel_skew<false, false, EQO, true, PP>, chain_kernel_skew<false, true, EQO, false, PP>, chain_kernel_skew<false, true, EQO, true, PP>,
                                  chain_kernel_skew<true, false, EQO, false, PP>, chain_kernel_skew<true, false, EQO, true, PP>, chain_kernel_skew<true, true, EQO, false, PP>, chain_kernel_skew<true, true, EQO, true, PP>};
    return sk_launch(args, dim3(n_items), dim3(64 * kSkWaves), sizeof(SkShared<EQO>), stream, attr_set, k);
}
template <bool PP>
static hipError_t launch_chain_skew_lev(const KArgs &args, uint32_t n_items, hipStream_t stream) {
    static bool attr_set[kMaxDevices] = {};
    static const SkKernel k[8] = {chain_kernel_skew_lev<false, false, false, PP>, chain_kernel_skew_lev<false, false, true, PP>, chain_kernel_skew_lev<false, true, false, PP>, chain_kernel_skew_lev<false, true, true, PP>,
                                  chain_kernel_skew_lev<true, false, false, PP>, chain_kernel_skew_lev<true, false, true, PP>, chain_kernel_skew_lev<true, true, false, PP>, chain_kernel_skew_lev<true, true, true, PP>};
    return sk_launch(args, dim3(n_items), dim3(64 * kSlWaves), sizeof(SlShared), stream, attr_set, k);
}
#endif
#if !defined(DSPI_PART) || DSPI_PART == 7
hipError_t launch_chain_skew(const KArgs &args, uint32_t n_items, int shape, hipStream_t stream) {      // shape 1 / 2 / 3: dspi_capi.cpp skew_class
    if (shape == 3) return launch_chain_skew_lev<false>(args, n_items, stream);
    return shape == 2 ? launch_chain_skew_t<true, false>(args, n_items, stream) : launch_chain_skew_t<false, false>(args, n_items, stream);
}
#endif
#if !defined(DSPI_PART) || DSPI_PART == 8
// ... with paired presets (every stream slot of a workgroup its own image of one structure, dspi_chain_skew.inc SkNum): part 8
hipError_t launch_chain_skew_pp(const KArgs &args, uint32_t n_items, int shape, hipStream_t stream) {
    if (shape == 3) return launch_chain_skew_lev<true>(args, n_items, stream);
    return shape == 2 ? launch_chain_skew_t<true, true>(args, n_items, stream) : launch_chain_skew_t<false, true>(args, n_items, stream);
}
#endif

#ifdef DSPI_PART_MAIN
// ---- value tiles of the per-lane-value rows (dspi_image.h): one thread per (row, word, column) ----
__global__ void pv_clear_kernel(const uint32_t *rows, float *vals, uint32_t all_differ) {
    uint32_t *mask = reinterpret_cast<uint32_t *>(vals + (size_t)rows[blockIdx.x] * kPvTileFloats + kPvMaskWord);
    if (threadIdx.x < 4) mask[threadIdx.x] = all_differ ? 0xffffffffu : 0u;
}
__global__ __launch_bounds__(256) void pv_build_kernel(const DevImage *img, const uint32_t *stream_image, const uint32_t *rows, float *vals, uint32_t n_streams) {
    constexpr int kWords = kPvBandSlots * 6 + PV_COUNT;
    const uint32_t wg = rows[blockIdx.y];
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;             // word * 128 + column: a wave writes columns of one word
    const uint32_t w = idx >> 7, col = idx & 127u;
    if (w >= (uint32_t)kWords) return;
    const uint32_t stream = wg * 128u + col;
    float *tile = vals + (size_t)wg * kPvTileFloats;
    float v = 0.0f;
    const DevImage *im = stream < n_streams ? img + stream_image[stream] : nullptr;
    size_t dst;
    if (w < (uint32_t)kPvBandSlots * 6u) {
        const uint32_t slot = w / 6u, k = w % 6u;
        if (im) {
            auto coef = [&](const DevImage *m) { return (slot < (uint32_t)(kMaxCh * kBands)) ? m->eq[slot / kBands][slot % kBands].c[k].u : m->loud[slot - kMaxCh * kBands].c[k].u; };
            const uint32_t bits = coef(im);
            v = __builtin_bit_cast(float, bits);
            // a band whose coefficient words are the same in every stream of the row can run on the image's scalars (value-tile mask)
            if (bits != coef(img + stream_image[wg * 128u])) atomicOr(reinterpret_cast<uint32_t *>(tile + kPvMaskWord) + (slot >> 5), 1u << (slot & 31u));
        }
        dst = ((size_t)(slot * 3u + (k >> 1)) * kLanes + (col >> 1)) * 4u + (k & 1u) * 2u + (col & 1u);
    } else {
        const int s = (int)w - kPvBandSlots * 6;
        if (im) {
            if (s == PV_PREAMP0) v = im->preamp[0].f;
            else if (s == PV_PREAMP1) v = im->preamp[1].f;
            else if (s == PV_VOL) v = im->vol.f;
            else if (s == PV_MASTER) v = im->master.f;
            else if (s < PV_MIX1) v = im->mix[0][s - PV_MIX0].f;
            else if (s < PV_OG) v = im->mix[1][s - PV_MIX1].f;
            else if (s < PV_LV) v = im->out_gain_lin[s - PV_OG];
            else if (s < PV_XF) { const float *lv = &im->lv_alpha_rms; v = lv[s - PV_LV]; }      // nine consecutive floats (dspi_image.h)
            else { const Word *xf = &im->xf_lp_a0; v = xf[s - PV_XF].f; }
        }
        dst = (size_t)kPvBandFloats + ((size_t)s * kLanes + (col >> 1)) * 2u + (col & 1u);
    }
    tile[dst] = v;
}

hipError_t launch_pv_build(const DevImage *img, const uint32_t *stream_image, const uint32_t *rows, uint32_t n_rows, float *vals, uint32_t n_streams, bool all_differ,
                           hipStream_t stream) {
    constexpr int kWords = kPvBandSlots * 6 + PV_COUNT;
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(pv_clear_kernel, dim3(n_rows), dim3(64), 0, stream, rows, vals, all_differ ? 1u : 0u);
    hipLaunchKernelGGL(pv_build_kernel, dim3((kWords * 128 + 255) / 256, n_rows), dim3(256), 0, stream, img, stream_image, rows, vals, n_streams);
    return hipGetLastError();
}

hipError_t launch_chain(int flavor, int packed, bool leveller_on, const KArgs &args, uint32_t n_items, hipStream_t stream) {
    // packed: 0 = one stream per lane, workgroup-uniform image (Q28) | 1 = packed float kernel | 2 = one stream per lane,
    // per-lane images (float always; Q28 rows with several presets)
    if (!flavor) return packed == 2 ? launch_chain_t<0, true>(args, n_items, stream) : launch_chain_t<0, false>(args, n_items, stream);
    // float: the context's contract (DSPI_FLOAT_CONTRACT_FMA) picks the kernel family
    if (packed == 5 || packed == 6) return launch_chain_skew(args, n_items, leveller_on ? 3 : (packed == 6 ? 2 : 1), stream);
    if (packed == 7 || packed == 8) return launch_chain_skew_pp(args, n_items, leveller_on ? 3 : (packed == 8 ? 2 : 1), stream);
    if (packed != 1 && packed != 3 && packed != 4) return args.fma ? launch_chain_t<1, false, true>(args, n_items, stream) : launch_chain_t<1, false, false>(args, n_items, stream);
    if (packed == 3) return args.fma ? launch_chain_pk_f5(args, leveller_on, n_items, stream) : launch_chain_pk_f2(args, leveller_on, n_items, stream);
    if (packed == 4) return args.fma ? launch_chain_pk_f4(args, leveller_on, n_items, stream) : launch_chain_pk_f1(args, leveller_on, n_items, stream);
    return args.fma ? launch_chain_pk_f3(args, leveller_on, n_items, stream) : launch_chain_pk_f0(args, leveller_on, n_items, stream);
}

// ---- debug: per-band taps of one float EQ channel (include/dspi.h dspi_debug_eq_taps) ----
// One thread, block-major like the reference loop (dsp_pipeline.c:281-365): band b over the whole buffer from zero state with
// the production sample loop (band_loop_f32) in the context's contract; `other` gets, for every sample, what the OTHER
// contract computes from the same input and the same state (a one-step comparison: the per-stage rounding difference).
template <bool FMA>
__global__ void eq_taps_kernel(const DevImage *img, int ch, const float *x, uint32_t n, float *taps, float *other) {
    for (uint32_t i = 0; i < n; ++i) taps[i] = x[i];
    for (int b = 0; b < kBands; ++b) {
        const DevBand &bd = img->eq[ch][b];
        const float c0 = bd.c[0].f, c1 = bd.c[1].f, c2 = bd.c[2].f, c3 = bd.c[3].f, c4 = bd.c[4].f, c5 = bd.c[5].f;
        float s1 = 0.0f, s2 = 0.0f;
        const float *in = taps + (size_t)b * n;
        float *out = taps + (size_t)(b + 1) * n, *alt = other + (size_t)b * n;
        for (uint32_t i = 0; i < n; ++i) {
            float xa[T], xb[T];
            xa[0] = xb[0] = in[i];
            float a1 = s1, a2 = s2;
            switch (bd.kind) {
                case K_BYPASS: break;
                case K_BIQUAD: band_loop_f32<true, K_BIQUAD, !FMA>(xb, 1, a1, a2, c0, c1, c2, c3, c4, c5); band_loop_f32<true, K_BIQUAD, FMA>(xa, 1, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_LP: band_loop_f32<true, K_SVF_LP, !FMA>(xb, 1, a1, a2, c0, c1, c2, c3, c4, c5); band_loop_f32<true, K_SVF_LP, FMA>(xa, 1, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_HP: band_loop_f32<true, K_SVF_HP, !FMA>(xb, 1, a1, a2, c0, c1, c2, c3, c4, c5); band_loop_f32<true, K_SVF_HP, FMA>(xa, 1, s1, s2, c0, c1, c2, c3, c4, c5); break;
                case K_SVF_PK: band_loop_f32<true, K_SVF_PK, !FMA>(xb, 1, a1, a2, c0, c1, c2, c3, c4, c5); band_loop_f32<true, K_SVF_PK, FMA>(xa, 1, s1, s2, c0, c1, c2, c3, c4, c5); break;
                default: band_loop_f32<true, K_SVF_SHELF, !FMA>(xb, 1, a1, a2, c0, c1, c2, c3, c4, c5); band_loop_f32<true, K_SVF_SHELF, FMA>(xa, 1, s1, s2, c0, c1, c2, c3, c4, c5); break;
            }
            out[i] = xa[0];
            alt[i] = xb[0];
        }
    }
}

hipError_t launch_eq_taps(bool fma, const DevImage *img, int ch, const float *x, uint32_t n, float *taps, float *other, hipStream_t stream) {
    if (fma) hipLaunchKernelGGL(eq_taps_kernel<true>, dim3(1), dim3(1), 0, stream, img, ch, x, n, taps, other);
    else hipLaunchKernelGGL(eq_taps_kernel<false>, dim3(1), dim3(1), 0, stream, img, ch, x, n, taps, other);
    return hipGetLastError();
}

hipError_t launch_state_ops(int flavor, const WgItem *items, uint32_t n_items, const StateOps &ops, uint32_t *state, uint32_t *dlines,
                            uint32_t *ring, uint32_t n_streams, hipStream_t stream) {
    if (flavor) hipLaunchKernelGGL(state_ops_kernel<1>, dim3(n_items), dim3(1024), 0, stream, items, ops, state, dlines, ring, n_streams);
    else hipLaunchKernelGGL(state_ops_kernel<0>, dim3(n_items), dim3(1024), 0, stream, items, ops, state, dlines, ring, n_streams);
    return hipGetLastError();
}

hipError_t launch_state_init(int flavor, uint32_t *state, uint32_t n_wg, hipStream_t stream) {
    if (flavor) hipLaunchKernelGGL(state_init_kernel<1>, dim3(n_wg), dim3(128), 0, stream, state, n_wg);
    else hipLaunchKernelGGL(state_init_kernel<0>, dim3(n_wg), dim3(64), 0, stream, state, n_wg);
    return hipGetLastError();
}

#ifdef DSPI_WAVE_TIMING
extern "C" int dspi_debug_wave_timing(unsigned long long *out84, int reset) {
    if (hipMemcpyFromSymbol(out84, HIP_SYMBOL(g_wave_timing), sizeof(unsigned long long) * 84) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[84] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_timing), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#endif  // DSPI_PART_MAIN

}  // namespace dspi
