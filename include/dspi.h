/*
 * dspi.h — C-ABI of libdspi_mi355x: the DSPi per-sample DSP chain on AMD Instinct MI355X.
 *
 * Drop-in boundary.  The reference firmware exposes no plugin/FFI interface: its chain is the
 * static function process_audio_packet(const uint8_t *data, uint16_t data_len)
 * (firmware/DSPi/usb_audio.c:500) reading ~40 file-scope globals that the USB control plane
 * writes.  What is frozen here is therefore the firmware's DATA interface:
 *
 *   audio in        interleaved little-endian stereo PCM, 16-bit or packed 24-bit, in packets
 *                   of block_len frames                 usb_audio.c:528-530, :591-686, :997-1015
 *   audio out       per S/PDIF pair int32 L,R words carrying a 24-bit sample; PDM sub as int32 Q28
 *                                                        usb_audio.c:926-955, :1244-1271
 *   whole state     WireBulkParams (2896 B, v2..v6)      bulk_params.h:42-205, bulk_params.c:178-377
 *                   PresetSlot v<=12 (+CRC-32)           flash_storage.c:136-189, :597-742
 *   parameters      EP0 vendor requests bRequest/wValue/payload
 *                                                        config.h:111-251, usb_audio.c:1632-2021 (SET), :2241-2688 (GET)
 *   UAC1 controls   volume (1/256 dB), mute, sample rate usb_audio.c:428-440, :1477-1499
 *   status          REQ_GET_STATUS wValue 9 byte layout  usb_audio.c:2427-2443
 *
 * One context drives `n_streams` independent DSPi devices ("streams") on one GPU.  Every
 * stream owns its filter state, delay lines and meters; parameters live in images shared by
 * any number of streams (all streams start on one image = factory defaults).
 *
 * Numerics: DSPI_FLAVOR_RP2040_Q28 is bit-exact integer arithmetic; DSPI_FLAVOR_RP2350_F32
 * is IEEE binary32 with flush-to-zero, either with no contraction or (DSPI_FLOAT_CONTRACT_FMA) with
 * exactly the fused multiply-adds GCC gives the firmware.  All match oracle/ bit-for-bit (tests/).
 * ONE boundary is defined by this repository rather than by the reference: the leveller calls
 * log10f once and powf twice per packet (leveller.c:178, :200, :206), the firmware links them from
 * an unpinned newlib / pico-float, and no libm agrees with another in the last bit.  The definition
 * here is one anyone can reproduce: the IEEE-754 CORRECTLY ROUNDED binary32 value (round to nearest,
 * ties to even) of log10(x) and of a^b — for every binary32 argument of log10f and of 10^y, and for
 * a^count on the firmware's eighteen smoothing coefficients (leveller.c:37-89) with count 1..192,
 * which are all the calls the leveller makes (results above e^88 clamp there, below e^-103 flush to 0).
 * include/dspi_detmath.h computes it (binary64 with a proven bound, double-double where the bound does
 * not decide; the kernels: the same floats through exception tables from an exhaustive walk,
 * include/dspi_detmath_tables.h); tests compare it with binary128 (libquadmath): 0 mismatches over
 * 10^7 arguments per function and over every argument the first step does not prove.  The product
 * and the oracle both use it, and the reference build used for pinning is compiled with
 * oracle/ref_math_hook.h (-include), which routes leveller.c's log10f / powf to the same header — the
 * reference build is MODIFIED at exactly this boundary and nowhere else.  glibc's own log10f / powf
 * differ from the correctly rounded value in ~2 % of calls, by at most 2 ulp; one golden vector
 * crosses the boundary: tests/golden/f32_full_96k_libm.npz (the reference with glibc's own libm,
 * BASELINE config 3's preset) is reproduced word for word by the oracle and by the GPU.
 * The float-to-int casts saturate as on both MCUs (vcvt.s32.f32 / the RP2040 bootrom's
 * float2int_z); an x86 build of the same C gives INT_MIN instead — for Q28 the one place this
 * shows in normal operation is the limiter quotient (leveller.c:376), where "bit-exact" rests on
 * that restated rule, not on executed reference code.
 *
 * Threading: a context is single-threaded.  Parameter calls take effect at the next
 * dspi_process() (= at a packet boundary, as in the firmware's main loop, main.c:826-894).
 * All functions return 0 / a non-negative count on success or a negative DSPI_E_* code; no
 * exceptions cross the boundary.
 */
#ifndef DSPI_H
#define DSPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSPI_ABI_VERSION 8   /* 2: DSPI_OUT_TILED, dspi_tile_streams, dspi_pdm_*, dspi_spdif_encode; 3: DSPI_FLOAT_CONTRACT_FMA,
                              * dspi_debug_eq_taps; 4: dspi_i2s_encode, vendor requests 0xC0 / 0xC1,
                              * dspi_debug_launch_plan, dspi_debug_image_count; 5: DSPI_OUT_ENABLED_ONLY, DSPI_OUT_I2S_SLOTS, DSPI_BOOT_POPULATED_FLASH, dspi_debug_launch_plan counts[5];
                              * 6: DSPI_OUT_SPDIF, dspi_spdif_block_pos; 7: dspi_out.clip_flags behind DSPI_OUT_CLIP_FLAGS, DSPI_OUT_SPDIF on every
                              * context, (additions only: a v6 caller's three-member dspi_out is never read past `peaks`);
                              * 8: dspi_debug_direct_stats, dspi_debug_detmath, the direct path polls a completion word for the call's own audio time (DSPI_DIRECT_SPIN_US, DSPI_DIRECT_POLL) */

/* flavours: values equal the firmware's platform ids (config.h:269-270) */
#define DSPI_FLAVOR_RP2040_Q28 0   /* 7 channels, 5 outputs, int32 Q28, 2048-sample delay lines */
#define DSPI_FLAVOR_RP2350_F32 1   /* 11 channels, 9 outputs, float, 4096-sample delay lines   */
/* OR into the flavour of dspi_create: the float flavour AS THE FIRMWARE IS BUILT.  firmware/DSPi/CMakeLists.txt:6-11 sets only
 * -O2/-O3, so GNU C's default -ffp-contract=fast applies and, the Cortex-M33 having vfma, GCC fuses a*b + c into one
 * rounding wherever its contraction pass pairs them: biquad / SVF recurrences (dsp_pipeline.c:298-362), loudness shelves
 * (usb_audio.c:697-712), matrix mix (:766), leveller envelope and smoother (leveller.c:165-166, :200), crossfeed
 * (crossfeed.c:137-148) and the coefficient design functions.  Without the flag every multiply and add rounds on its own
 * (the source read literally, -ffp-contract=off).  Both contracts are pinned bit-for-bit to the reference compiled the
 * corresponding way (oracle/Makefile, tests/test_oracle_vs_fw.py); the fused one needs a third fewer vector instructions.
 * PRECISELY what this contract is: "GCC 11, x86-64, -ffp-contract=fast -mfma with the vectorisers off: the GIMPLE-level contraction of
 * the same source" — not "the Cortex-M33 binary".  That arm-none-eabi-gcc fuses the same pairs is read off GCC's target-independent
 * tree pass (the .FMA / .FMS / .FNMA calls of -fdump-tree-optimized, listed in DESIGN.md section 5), not executed: the image has no ARM
 * toolchain.  What north_star asks of the two contracts — within 1 ULP per stage of each other — is measured (profiles/r02_ulp_per_stage.md). */
#define DSPI_FLOAT_CONTRACT_FMA 0x100
#define DSPI_FLAVOR_RP2350_F32_FMA (DSPI_FLAVOR_RP2350_F32 | DSPI_FLOAT_CONTRACT_FMA)
/* OR into the flavour of dspi_create: what the power-on models.  By default every stream is a device that boots for the FIRST time on an
 * erased flash: preset_boot_load writes the fresh preset directory (flash_storage.c:1097-1100), and every flash sector write arms the
 * preset mute for flash_mute_hold_samples() = max(10 ms, 512) samples (:272-276, :347-348) — so a new context starts with 512 muted
 * samples and the fade-in (5 to 12 ms of silence / ramp at the start of the first packets).  With DSPI_BOOT_POPULATED_FLASH the streams
 * are devices whose flash already holds a directory: nothing is written at boot, audio starts unmuted; load the preset such a device
 * would have booted with dspi_load_flash_dump BEFORE the first dspi_process: on such a context that call IS the boot — the power-on
 * sequence runs again over the dump (preset_boot_load -> apply_slot_to_live, flash_storage.c:1047-1082: the selected slot goes into
 * the live parameters, no mute is armed, no line is zeroed; a boot that itself writes the flash — no directory, a legacy sector to
 * migrate, a v1 directory to convert — arms the 512-sample mute like a first boot; slots saved as I2S arm the type-switch mute,
 * main.c:651-684), and the sample rate, UAC1 volume and mute set since dspi_create are set again after it.  Exact against the firmware
 * build booted from the same flash FROM FRAME 0 (tests/test_gpu_parity.py::test_flash_dump_boots_device_context,
 * tests/test_oracle_vs_fw.py::test_boot_from_flash_dumps).  Once a context has processed audio — and on every context without the
 * flag — dspi_load_flash_dump is a running device switching to the preset the dump selects (preset_load, flash_storage.c:794-849:
 * max(10 ms, 512)-sample mute, delay lines zeroed), exactly like dspi_load_preset_slot. */
#define DSPI_BOOT_POPULATED_FLASH 0x200

#define DSPI_ALL_STREAMS (-1)
#define DSPI_DEVICE_NONE (-1)      /* host-only context: parameter surface works, dspi_process fails */
#define DSPI_MAX_BLOCK_LEN 192     /* usb_audio.c:273-276, :588 */

/* error codes */
#define DSPI_OK 0
#define DSPI_E_INVAL (-10)         /* bad argument */
#define DSPI_E_NODEVICE (-11)      /* HIP runtime / GPU unavailable or host-only context */
#define DSPI_E_NOMEM (-12)
#define DSPI_E_HIP (-13)           /* a HIP call failed; see dspi_last_error() */
#define DSPI_E_UNSUPPORTED (-14)   /* vendor request outside the DSP subset (USB stall in the firmware) */
#define DSPI_E_SHORT (-15)         /* output buffer too small */
/* dspi_load_bulk returns the firmware's own codes: -1 version, -2 platform, -3 channel counts, -4 length */
/* dspi_load_preset_slot returns PRESET_OK (0) or PRESET_ERR_CRC (3)  (config.h:262-266) */

/* dspi_process flags.  Bits not defined here are refused (DSPI_E_INVAL), never ignored: a later ABI may give a bit a meaning that
 * reads further members of dspi_out (as DSPI_OUT_CLIP_FLAGS did in ABI 7). */
#define DSPI_MEM_DEVICE 0x1u       /* pcm_in and every pointer in dspi_out are device pointers (zero-copy) */
#define DSPI_OUT_TILED 0x2u        /* pairs / sub use the device-native tiled layout described at dspi_out */
#define DSPI_OUT_I2S_SLOTS 0x8u    /* pairs whose slot is an I2S slot in the stream's own parameters (output_types[], REQ_SET_OUTPUT_TYPE 0xC0) are written as
                                    * the words the I2S driver shifts out — the S/PDIF producer word left-justified, word << 8
                                    * (pico_audio_i2s_multi/audio_i2s_multi.c:217-226) — instead of going through dspi_i2s_encode afterwards: the same
                                    * words, without the second pass over 64 bytes per frame.  S/PDIF-typed pairs are unaffected. */
#define DSPI_OUT_SPDIF 0x10u       /* `pairs` takes what the S/PDIF driver shifts out instead of the producer words: per frame and pair the two IEC 60958
                                    * subframes of spdif_update_subframe (pico_audio_spdif_multi, sample_encoding.h:27-47; dspi_spdif_encode below) —
                                    *   pairs  uint32 [stream][pair][F][4]   {left lo, left hi, right lo, right hi}: TWICE the bytes of the word layout —
                                    * the words dspi_process + dspi_spdif_encode give, without the second pass (8 bytes in, 16 out per frame and pair).
                                    * The position in the 192-frame channel-status block runs on from call to call (dspi_spdif_block_pos); the
                                    * sample-rate byte of the channel status is each stream's own rate (audio_spdif.c:250-256).  On EVERY context:
                                    * where the float chain's latency layout serves the launch (small contexts) its output waves encode the
                                    * subframes themselves; on every other launch the library runs the chain into a scratch buffer of pair words,
                                    * row chunk by row chunk, and the subframe encoder from there (on the packed kernel the fused encoder would
                                    * cost the chain 60 % more instructions).  Not with DSPI_OUT_TILED or DSPI_OUT_I2S_SLOTS. */
#define DSPI_OUT_ENABLED_ONLY 0x4u /* the caller does not read the sample words of SILENT outputs — an S/PDIF pair whose two outputs are
                                    * disabled (the firmware zero-fills it, usb_audio.c:930-933), the sub while it is disabled or Core 1
                                    * runs the EQ worker — so the library may leave those parts of pairs / sub unwritten instead of storing
                                    * zeros (36 of the 40 bytes per frame for a preset with one live pair).  Peaks, status and every live
                                    * output are unaffected.  Honoured by the float chain's latency layout; the other kernels write the zeros
                                    * (host buffers: the silent parts come back as zeros either way).  With DSPI_OUT_SPDIF: where the library
                                    * serves the call in two passes (any lane off the latency layout) the silent pairs carry the subframes of
                                    * SILENCE, exactly as without this flag; where the latency layout's output waves encode the subframes
                                    * themselves the silent pairs are left unwritten (host buffers: zero words, which are not valid subframes —
                                    * do not shift them out). */
#define DSPI_OUT_CLIP_FLAGS 0x20u  /* the dspi_out passed has the ABI-7 member clip_flags (below) and the library may write through it */

typedef struct dspi_ctx dspi_ctx;

/* Output buffers, all optional (NULL = not produced).  Layouts, F = n_blocks*block_len frames:
 *   pairs      int32 [stream][pair][F][2]   pair p = outputs 2p,2p+1; 24-bit sample per word
 *   sub        int32 [stream][F]            PDM sub channel, Q28 (zeros when the firmware would push nothing)
 *   peaks      uint16 [stream][block][C]    per-packet peak meters (global_status.peaks, config.h:455-460)
 *   clip_flags uint16 [stream]              sticky clip bits (with DSPI_OUT_CLIP_FLAGS only, see the member)
 * where C = 11 / 7 channels and pair count = 4 / 2.
 *
 * With DSPI_OUT_TILED the sample words are stream-minor instead ("tiles" of R = dspi_tile_streams() consecutive
 * streams, 128 float / 64 Q28; tile = stream / R, column = stream % R):
 *   pairs      int32 [tile][output][F][R]   output o = 0 .. 2*pairs-1 (pair o/2, side o%2), same 24-bit words
 *   sub        int32 [tile][F][R]
 * Buffers cover whole tiles (ceil(n_streams / R) of them); columns past n_streams are never written.
 * This is how the kernel produces the samples (one 512-byte row per frame per output, like its delay lines): the
 * layout for a GPU-resident consumer.  The stream-major layout above is the per-device view (one S/PDIF pair buffer
 * per stream) and costs scattered 4-byte writes. */
typedef struct dspi_out {
    int32_t *pairs;
    int32_t *sub;
    uint16_t *peaks;
    uint16_t *clip_flags;  /* ABI 7, read ONLY when flags carry DSPI_OUT_CLIP_FLAGS: uint16 [stream], every stream's sticky clip bits after the
                            * call's last packet — global_status.clip_flags as REQ_GET_STATUS reports them (usb_audio.c:2427-2443: bit c =
                            * channel c's peak exceeded full scale since the last REQ_CLEAR_CLIPS) — one pass instead of a dspi_get_status
                            * per stream.  Not cleared by reading. */
} dspi_out;

/* ---- lifecycle ---------------------------------------------------------------------- */
int dspi_create(dspi_ctx **out, int flavor, uint32_t n_streams, int hip_device);
void dspi_destroy(dspi_ctx *ctx);
const char *dspi_last_error(const dspi_ctx *ctx);
int dspi_abi_version(void);
int dspi_num_channels(const dspi_ctx *ctx);   /* 11 / 7 */
int dspi_num_outputs(const dspi_ctx *ctx);    /* 9 / 5  */
int dspi_num_pairs(const dspi_ctx *ctx);      /* 4 / 2  */
uint32_t dspi_num_streams(const dspi_ctx *ctx);
uint32_t dspi_tile_streams(const dspi_ctx *ctx);   /* R of the tiled layouts: 128 / 64 */

/* ---- whole-state blobs ---------------------------------------------------------------- */
/* REQ_FACTORY_RESET / loading an empty preset slot: flash_storage.c:1144-1238, :811-833 */
int dspi_factory_defaults(dspi_ctx *ctx, int32_t stream);
/* REQ_SET_ALL_PARAMS: main.c:1126-1162 + bulk_params.c:178-377 */
int dspi_load_bulk(dspi_ctx *ctx, int32_t stream, const void *blob, size_t len);
/* REQ_GET_ALL_PARAMS: bulk_params.c:62-172.  Returns 2896. */
int dspi_collect_bulk(dspi_ctx *ctx, int32_t stream, void *blob, size_t cap);
/* REQ_PRESET_LOAD of an occupied slot: main.c:926-976, flash_storage.c:750-849.
 * `expect_slot` = slot number the image must carry (validate_slot), or -1 to accept any. */
int dspi_load_preset_slot(dspi_ctx *ctx, int32_t stream, const void *image, size_t len, int expect_slot);
/* collect_live_state: flash_storage.c:464-552.  Returns the image size (2864 / 1840). */
int dspi_save_preset_slot(dspi_ctx *ctx, int32_t stream, void *image, size_t cap, int slot_index);

/* ---- flash dump --------------------------------------------------------------- */
/* A raw image of the firmware's 48 KB preset area (flash_storage.c:4-26): sector 0 = preset directory (v1 or v2),
 * sectors 1-10 = preset slots 0-9, sector 11 = the legacy single-preset sector. */
#define DSPI_FLASH_DUMP_BYTES (12 * 4096)
typedef struct dspi_flash_dir {
    int32_t valid;                 /* 0: no directory, bad CRC or unknown version */
    int32_t version;               /* 1 or 2 as stored; v1 fields are mapped to v2 as the firmware's migration does (:391-411) */
    uint8_t startup_mode;          /* 0 PRESET_STARTUP_SPECIFIED, 1 _LAST_ACTIVE (config.h:258-259) */
    uint8_t default_slot, last_active_slot, include_pins;
    uint16_t slot_occupied;        /* bit n: slot n holds a preset */
    uint8_t master_volume_mode, pad_;
    float master_volume_db;
    char slot_names[10][32];
} dspi_flash_dir;
/* dir_load_cache: flash_storage.c:370-417 (no context needed) */
int dspi_flash_read_directory(const void *dump, size_t len, dspi_flash_dir *out);
/* Picks the startup preset the way preset_boot_load does (flash_storage.c:1047-1105: startup mode, occupancy, slot
 * validation, legacy-sector migration :997-1045).  On a DSPI_BOOT_POPULATED_FLASH context that has not processed audio yet
 * the call is the device's BOOT from that flash (see the flag: no mute, no line zeroing, exact from frame 0); on every
 * other context the preset is applied the way REQ_PRESET_LOAD does on a running device.  Returns 0..9 = that slot was loaded; 16+n = slot n was selected but is empty or corrupt, factory defaults
 * applied; 32 = no directory, legacy sector migrated and loaded; 48 = nothing usable, factory defaults; negative DSPI_E_*. */
int dspi_load_flash_dump(dspi_ctx *ctx, int32_t stream, const void *dump, size_t len);

/* ---- incremental parameters ------------------------------------------------------------ */
/* vendor_cmd_packet (usb_audio.c:1632-2021): payloads shorter than the request needs are ignored, as upstream */
int dspi_vendor_set(dspi_ctx *ctx, int32_t stream, uint8_t bRequest, uint16_t wValue, const void *payload, uint16_t len);
/* vendor_setup_request_handler IN direction (usb_audio.c:2271-2688).  Returns the byte count. */
int dspi_vendor_get(dspi_ctx *ctx, int32_t stream, uint8_t bRequest, uint16_t wValue, void *buf, uint16_t cap);
int dspi_set_host_volume(dspi_ctx *ctx, int32_t stream, int16_t volume_1_256_db);   /* audio_set_volume */
int dspi_set_mute(dspi_ctx *ctx, int32_t stream, int mute);
int dspi_set_sample_rate(dspi_ctx *ctx, int32_t stream, uint32_t hz);               /* 44100 / 48000 / 96000 */

/* ---- audio ------------------------------------------------------------------------------ */
/* Runs n_blocks packets of block_len frames for every stream.
 * pcm_in: [stream][n_blocks*block_len] frames of interleaved LE stereo; bit_depth 16 (4 B/frame)
 * or 24 (6 B/frame, packed).  With DSPI_MEM_DEVICE the call is asynchronous on the context's
 * HIP stream (use dspi_sync); without it buffers are host memory and the call returns when
 * the outputs are in place. */
int dspi_process(dspi_ctx *ctx, const void *pcm_in, int bit_depth, uint32_t n_blocks, uint32_t block_len,
                 const dspi_out *out, uint32_t flags);
int dspi_sync(dspi_ctx *ctx);

/* ---- PDM sub output ---------------------------------------------------- */
/* The consumer of dspi_out.sub: the firmware's 256x oversampled 2nd-order sigma-delta modulator with noise-shaped
 * dither (pdm_generator.c:62-108, :351-397; the loop Core 1 runs in CORE1_MODE_PDM).  One stream = one modulator;
 * its state (integrators, noise shaper, dither RNG, 1024-sample fade-in) lives in the context and carries across calls.
 *   sub    int32 [stream][n_frames]        Q28, exactly what dspi_process wrote to dspi_out.sub
 *   words  uint32 [stream][n_frames][8]    8 x 32 PDM bits per sample, MSB = first bit on the wire (the words the
 *                                          firmware queues for its PIO/DMA)
 * With DSPI_OUT_TILED: sub = [tile][n_frames][R], words = [tile][n_frames][8][R].  DSPI_MEM_DEVICE as in dspi_process.
 * Not modelled: DMA ring pacing, under-run recovery and the fade-out on disable (transport, not sample arithmetic). */
int dspi_pdm_modulate(dspi_ctx *ctx, const int32_t *sub, uint32_t n_frames, uint32_t *words, uint32_t flags);
/* the re-enable path (pdm_generator.c:241-252): integrators, noise shaper and fade-in restart, the dither RNG runs on */
int dspi_pdm_restart(dspi_ctx *ctx, int32_t stream);

/* ---- S/PDIF subframes ---------------------------------------------------- */
/* What the firmware's S/PDIF outputs do to the words of dspi_out.pairs before the PIO shifts them out
 * (pico_audio_spdif_multi: spdif_update_subframe, sample_encoding.h:27-47; preambles Z/X/Y, consumer channel status with
 * the sample-rate byte of stream 0's current rate, 192-frame block position: audio_spdif.c:76-116, :250-256, :385-405).
 *   pairs      int32 [stream][pair][n_frames][2]       exactly what dspi_process wrote
 *   subframes  uint32 [stream][pair][n_frames][4]      {l, h} of the left subframe, {l, h} of the right one: the 16 bytes
 *                                                      the firmware's DMA buffer holds per stereo frame
 * With DSPI_OUT_TILED: pairs = [tile][output][n_frames][R], subframes = [tile][pair][n_frames][4][R].
 * block_pos = position of the first frame in the 192-frame channel-status block (0..191).  Returns the position that
 * follows the last frame (>= 0; feed it to the next call) or a negative DSPI_E_*. */
int dspi_spdif_encode(dspi_ctx *ctx, const int32_t *pairs, uint32_t n_frames, uint32_t block_pos, uint32_t *subframes, uint32_t flags);
/* DSPI_OUT_SPDIF: the position in the 192-frame block of the next dspi_process call's first frame (0 after dspi_create, then advanced by
 * every call made with the flag).  set in 0..191 sets it first; set < 0 only reads.  Returns the position or a negative DSPI_E_*. */
int dspi_spdif_block_pos(dspi_ctx *ctx, int32_t set);
/* ---- I2S slots ----------------------------------------------------------- */
/* An output slot switched to I2S (REQ_SET_OUTPUT_TYPE 0xC0 / output_types[] of a preset, config.h:286-287) takes the same
 * words as an S/PDIF slot and left-justifies them into 32-bit I2S slots, L then R, MSB first on the wire
 * (pico_audio_i2s_multi/audio_i2s_multi.c:217-226: dst = src << 8).
 *   pairs  int32  [stream][pair][n_frames][2]   exactly what dspi_process wrote        (DSPI_OUT_TILED: [tile][output][n_frames][R])
 *   words  uint32 [stream][pair][n_frames][2]   same shape; only the pairs in pair_mask (bit p = pair p) are written
 * pair_mask = DSPI_I2S_PAIRS_BY_TYPE (0): the pairs whose current output type is I2S in stream 0's parameters (the type of a
 * slot is a property of the device, like the sample rate).  Returns the mask that was encoded (>= 0) or a negative DSPI_E_*. */
#define DSPI_I2S_PAIRS_BY_TYPE 0u
int dspi_i2s_encode(dspi_ctx *ctx, const int32_t *pairs, uint32_t n_frames, uint32_t pair_mask, uint32_t *words, uint32_t flags);
/* the HIP stream (hipStream_t) the context launches on, for event timing by the caller */
void *dspi_hip_stream(dspi_ctx *ctx);

/* ---- status ------------------------------------------------------------------------------ */
/* REQ_GET_STATUS wValue 9: peaks[C] LE u16, cpu0, cpu1 (always 0 here), clip_flags LE u16 = 26 / 18 bytes */
int dspi_get_status(dspi_ctx *ctx, int32_t stream, void *buf, size_t cap);
/* REQ_CLEAR_CLIPS: returns the flags that were set (DSPI_ALL_STREAMS: clears every stream, returns stream 0's flags) */
int dspi_clear_clips(dspi_ctx *ctx, int32_t stream);

/* ---- introspection for tests (host-side derived parameter image of a stream) ------------ */
/* Copies the packed device parameter image; returns its size.  Layout is internal (csrc/dspi_image.h). */
int dspi_debug_image(dspi_ctx *ctx, int32_t stream, void *buf, size_t cap);
/* Which kernel the context's rows currently go to (after the last dspi_process): work items per launch list, counts[5] =
 * {Q28 shared image, packed float shared image, one-stream kernel with per-lane parameter images, packed float with per-lane values
 * incl. band coefficients, packed float with per-lane values and shared band coefficients}.  Tests use it to prove that a scenario
 * ran on the path it was written for.  With n_counts >= 6, counts[5] = items of the float chain's latency layout (any of its three
 * shapes: launches small enough to leave the chip underfilled); with n_counts >= 7, counts[6] = those of them that serve several
 * presets of one structure at once (a workgroup's stream slots each read their own image).  Returns the number of counts written
 * (5, 6 or 7) or a negative DSPI_E_*. */
int dspi_debug_launch_plan(dspi_ctx *ctx, uint32_t *counts, size_t n_counts);
/* include/dspi_detmath.h evaluated on the DEVICE, host buffers, n <= 2^24: which = 0: out[i] = log10f(a[i]) and 1: powf(a[i], b[i]) in the two-step
 * forms; 2: log10f, 3: 10^a[i], 4: a[i]^b[i] in the forms the chain kernels use (step 1 + exception tables).  Tests compare all of them bit for
 * bit with the host build of the same header and with binary128. */
int dspi_debug_detmath(dspi_ctx *ctx, int which, const float *a, const float *b, uint32_t n, float *out);
/* Small calls on host buffers (one packet per call, usb_audio.c:1326-1332) do not sleep on their stream: the stream writes the call's sequence
 * number into a word of pinned host memory behind the launches (hipStreamWriteValue32) and the host polls that word — a load per poll, no call
 * into the runtime while waiting, 3-4 us less per call than polling hipStreamQuery — for the audio time the call carries (frames at 44.1 kHz; at
 * least 300 us, at most 50 ms; DSPI_DIRECT_SPIN_US, read at dspi_create, overrides), then the blocking wait.  DSPI_DIRECT_POLL=query polls the
 * stream instead (also the fallback where the write is not available).  What neither way of waiting removes: about one call in 10^5 during
 * which the calling thread itself does not run for 0.5 - 10 ms (the wait phase is long, yet the loop's own clock check never fired): the
 * host's scheduler, for a caller to address with a real-time priority or an isolated core.
 * out[5] = {such calls so far, calls that reached the blocking wait, longest enqueue phase in ns (entry -> launches issued),
 * longest wait phase in ns (these three over the calls after the context's first eight), the last call's polling budget in ns}.  tools/bench_realtime.py reports them next to the latency percentiles.
 * Returns 5 or a negative DSPI_E_*. */
int dspi_debug_direct_stats(dspi_ctx *ctx, uint64_t *out, size_t n);
/* Number of distinct parameter objects the context holds (streams share one until a per-stream call separates them; streams that
 * received the same whole state again through broadcast calls are folded back, here or at the next dspi_process).  Works on
 * host-only contexts.  Returns the count or a negative DSPI_E_*. */
int dspi_debug_image_count(dspi_ctx *ctx);

/* Per-band taps of one EQ channel (float flavour; the parity procedure of SURVEY.md section 8d).  x[n] is run through the ten bands
 * of `channel` (0-1 master, 2.. outputs) of `stream`'s current parameters from zero state, band-major like the firmware's block loop
 * (dsp_pipeline.c:281-365), with the production sample loop in the context's float contract:
 *   taps   float [11][n]   taps[0] = x, taps[b+1] = output of band b
 *   other  float [10][n]   for every sample of band b, what the OTHER contract (canonical <-> DSPI_FLOAT_CONTRACT_FMA) computes from the
 *                          same input and the same filter state: the per-stage rounding difference between the two
 * Host buffers; n <= 2^20.  The chain's own state is not touched. */
int dspi_debug_eq_taps(dspi_ctx *ctx, int32_t stream, int channel, const float *x, uint32_t n, float *taps, float *other);

#ifdef __cplusplus
}
#endif
#endif /* DSPI_H */
