/*
 * ref_fw.c — "firmware build" of the oracle: the reference's usb_audio.c compiled IN PLACE.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/README.md).  Built only where /root/reference exists, into
 * oracle/_ref/libref_fw_{f32,q28}[_fma].so (oracle/Makefile `ref`).
 *
 * What this pins: the packet orchestrator itself (`process_audio_packet`, usb_audio.c:500-1317), the vendor
 * SET/GET handlers (usb_audio.c:1632-2021, :2241-3147), audio_set_volume / update_preamp / update_master_volume,
 * flash_storage.c (ref_fw_flash.c) and pdm_generator.c's Core-1 twin and sigma-delta loop (ref_fw_core1.c) —
 * everything orc_chain.c restates by hand — by running the reference's own translation units behind the
 * API of orc_api.h, so tests drive both builds with the same calls and compare every word.
 *
 * How: usb_audio.c is #included below (its statics are needed: process_audio_packet,
 * vendor_setup_request_handler, _audio_reconfigure, preset_mute_smooth_gain, ...).  The un-vendored pico-sdk
 * is replaced by ref_stub_sdk/ (declarations) and ref_fw_stubs.c (RAM flash, no-op hardware); pico-extras
 * headers are the reference's own.  One library instance = one device (the firmware keeps its state in
 * file-scope globals): tests load a private copy of the .so per stream.
 *
 * main.c (core0_init, the main loop with its deferred-apply dispatcher, rate changes, pipeline resets, output type switches) is
 * compiled in place too and runs as a coroutine, one loop iteration per fw_main_step() (ref_fw_main.c).
 * Restated here: the ARM inline asm of the 24-bit float unpack (usb_audio.c:613-646, :657-675, two statements) and, for the Q28
 * flavour, the Thumb block biquad dsp_process_rp2040.S:225-394 (ref_fw_stubs.c; pinned by executing the assembly, tests/thumb.py).
 */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>
#include <assert.h>
#include <limits.h>
#include <xmmintrin.h>

void fw_core1_eq_once(void);
#define ORC_WFE() fw_core1_eq_once()      /* process_audio_packet waits for Core 1 with __wfe() (usb_audio.c:866-868) */
#include "pico_stub_all.h"
#define DCP_INLINE_H                       /* dcp_inline.h: unused static inlines made of RP2350 coprocessor asm */

/* ---- the two ARM asm statements of the 24-bit float input path (usb_audio.c:613-646, :657-675) ----
 * `__asm__ volatile ( ... )` is rewritten by argument count into the C the asm comments describe:
 * three little-endian words i0,i1,i2 hold l1 r1 l2 r2 as packed 24-bit samples. */
static inline int32_t fw_sx24(uint32_t v) { return (int32_t)(v << 8) >> 8; }
#define ORC_NARGS_(a, b, c, d, e, f, g, h, N, ...) N
#define ORC_NARGS(...) ORC_NARGS_(__VA_ARGS__, 8, 7, 6, 5, 4, 3, 2, 1, 0)
#define ORC_CAT_(a, b) a##b
#define ORC_CAT(a, b) ORC_CAT_(a, b)
#define ORC_ASM_7() do { (void)temp; \
        l1 = (float)fw_sx24((uint32_t)i0 & 0xFFFFFFu); \
        r1 = (float)fw_sx24((((uint32_t)i1 & 0xFFFFu) << 8) | ((uint32_t)i0 >> 24)); \
        l2 = (float)fw_sx24((((uint32_t)i2 & 0xFFu) << 16) | ((uint32_t)i1 >> 16)); \
        r2 = (float)((int32_t)i2 >> 8); } while (0)
#define ORC_ASM_4() do { (void)temp; \
        l1 = (float)fw_sx24((uint32_t)i0 & 0xFFFFFFu); \
        r1 = (float)fw_sx24((((uint32_t)i1 & 0xFFFFu) << 8) | ((uint32_t)i0 >> 24)); } while (0)
#define volatile(...) ORC_CAT(ORC_ASM_, ORC_NARGS(__VA_ARGS__))()
#define __asm__

#include "usb_audio.c"

#undef volatile
#undef __asm__

#include "../include/dspi_detmath.h"
#include "orc_api.h"

/* ------------------------------------------------------------------------------------- */
/* other translation units of this library                                                */
/* ------------------------------------------------------------------------------------- */
int fw_core1_take_sub(int32_t *dst, int max);             /* ref_fw_core1.c: drain pdm_push_sample()'s ring */
int fw_flash_slot_size(void);                             /* ref_fw_flash.c */
void fw_flash_forget_dir(void);
void fw_flash_mark_occupied(int slot);
void fw_flash_collect_slot(void *image, int slot_index);
extern uint8_t orc_flash_image[];
extern uint32_t orc_flash_bytes;
extern uint32_t orc_flash_preset_base;

/* ------------------------------------------------------------------------------------- */
/* pico-extras entry points the orchestrator touches: audio buffers and control transfers  */
/* ------------------------------------------------------------------------------------- */
static int32_t fw_pair_words[NUM_SPDIF_INSTANCES][192 * 2];
static mem_buffer_t fw_mem[NUM_SPDIF_INSTANCES];
static audio_buffer_t fw_abuf[NUM_SPDIF_INSTANCES];
static audio_buffer_pool_t fw_pools[NUM_SPDIF_INSTANCES];
static int fw_pool_count;

audio_buffer_pool_t *audio_new_producer_pool(audio_buffer_format_t *format, int buffer_count, int buffer_sample_count) {
    (void)format; (void)buffer_count; (void)buffer_sample_count;
    int i = fw_pool_count++ % NUM_SPDIF_INSTANCES;
    fw_mem[i].bytes = (uint8_t *)fw_pair_words[i]; fw_mem[i].size = sizeof(fw_pair_words[i]);
    fw_abuf[i].buffer = &fw_mem[i]; fw_abuf[i].max_sample_count = 192;
    return &fw_pools[i];
}
audio_buffer_t *take_audio_buffer(audio_buffer_pool_t *ac, bool block) { (void)block; return &fw_abuf[ac - fw_pools]; }
void give_audio_buffer(audio_buffer_pool_t *ac, audio_buffer_t *buffer) { (void)ac; (void)buffer; }

struct usb_endpoint usb_control_in, usb_control_out;
static uint8_t fw_out_bytes[64], fw_in_bytes[4096];
static struct usb_buffer fw_out_buf = {.data = fw_out_bytes, .data_max = sizeof fw_out_bytes};
static struct usb_buffer fw_in_buf = {.data = fw_in_bytes, .data_max = 64};
static const struct usb_transfer_type *fw_pending_out_type;
static int fw_in_len = -1;                                  /* bytes of the last GET response, -1 = none */
static const uint8_t *fw_set_payload; static uint32_t fw_set_len;

struct usb_buffer *usb_current_out_packet_buffer(struct usb_endpoint *ep) { (void)ep; return &fw_out_buf; }
struct usb_buffer *usb_current_in_packet_buffer(struct usb_endpoint *ep) { (void)ep; return &fw_in_buf; }
void usb_start_control_out_transfer(const struct usb_transfer_type *type) { fw_pending_out_type = type; }
void usb_start_single_buffer_control_in_transfer(void) { fw_in_len = fw_in_buf.data_len; }
void usb_start_tiny_control_in_transfer(uint32_t data, uint len) { memcpy(fw_in_bytes, &data, 4); fw_in_len = (int)len; }
void usb_start_empty_control_in_transfer_null_completion(void) {}
void usb_start_empty_transfer(struct usb_endpoint *endpoint, struct usb_transfer *transfer, usb_transfer_completed_func on_complete) {
    if (on_complete) on_complete(endpoint, transfer);
}
static usb_transfer_completed_func fw_stream_done;
void usb_stream_setup_transfer(struct usb_stream_transfer *transfer, const struct usb_stream_transfer_funcs *funcs,
                               uint8_t *chunk_buffer, uint32_t chunk_size, uint32_t transfer_length,
                               usb_transfer_completed_func on_complete) {
    memset(transfer, 0, sizeof *transfer);
    transfer->funcs = funcs; transfer->chunk_buffer = chunk_buffer; transfer->chunk_size = chunk_size;
    transfer->transfer_length = transfer_length; fw_stream_done = on_complete;
}
void usb_start_transfer(struct usb_endpoint *ep, struct usb_transfer *transfer) {
    struct usb_stream_transfer *st = (struct usb_stream_transfer *)transfer;     /* .core is the first member */
    if (ep == &usb_control_out) {                      /* host -> device data stage of REQ_SET_ALL_PARAMS */
        uint32_t n = fw_set_len < st->transfer_length ? fw_set_len : st->transfer_length;
        memcpy(st->chunk_buffer, fw_set_payload, n);
    } else {                                           /* device -> host: REQ_GET_ALL_PARAMS */
        memcpy(fw_in_bytes, st->chunk_buffer, st->transfer_length);
        fw_in_len = (int)st->transfer_length;
    }
    if (fw_stream_done) fw_stream_done(ep, transfer);
}
void usb_grow_transfer(struct usb_transfer *transfer, uint packet_count) { (void)transfer; (void)packet_count; }
bool usb_stream_noop_on_chunk(uint32_t chunk_len, struct usb_stream_transfer *transfer) { (void)chunk_len; (void)transfer; return false; }
void usb_stream_noop_on_packet_complete(struct usb_stream_transfer *transfer) { (void)transfer; }
void usb_packet_done(struct usb_endpoint *ep) { (void)ep; }
void usb_set_default_transfer(struct usb_endpoint *ep, struct usb_transfer *transfer) { (void)ep; (void)transfer; }
struct usb_interface *usb_interface_init(struct usb_interface *interface, const struct usb_interface_descriptor *desc,
                                         struct usb_endpoint *const *endpoints, uint endpoint_count, bool double_buffered) {
    (void)desc; (void)endpoints; (void)endpoint_count; (void)double_buffered; return interface;
}
static struct usb_device fw_usb_device;
struct usb_device *usb_device_init(const struct usb_device_descriptor *desc, const struct usb_configuration_descriptor *config_desc,
                                   struct usb_interface *const *interfaces, uint interface_count,
                                   const char *(*get_descriptor_string)(uint index)) {
    (void)desc; (void)config_desc; (void)interfaces; (void)interface_count; (void)get_descriptor_string; return &fw_usb_device;
}
void usb_device_start(void) {}

/* ------------------------------------------------------------------------------------- */
/* main.c: compiled in place and run as a coroutine (ref_fw_main.c)                        */
/* ------------------------------------------------------------------------------------- */
extern int fw_last_bulk_err, fw_last_preset_status;
void fw_main_start(void);      /* power-on: main() up to the top of its loop (core0_init, boot preset) */
void fw_main_step(void);       /* one iteration of the reference's main loop */
static unsigned fw_enter(void) { unsigned c = _mm_getcsr(); _mm_setcsr(c | 0x8040u); return c; }   /* FPSCR.FZ, main.c:593-600 */
static void fw_leave(unsigned c) { _mm_setcsr(c); }
/* the loop runs until nothing is pending (a serviced flag can raise another, e.g. factory reset -> loudness) */
static void fw_service_all(void) { for (int i = 0; i < 4; i++) fw_main_step(); }

/* ------------------------------------------------------------------------------------- */
/* orc_api.h over the firmware's globals                                                   */
/* ------------------------------------------------------------------------------------- */
extern int orc_math_mode;
int orc_flavor(void) { return PICO_RP2350 ? 1 : 0; }
int orc_is_ref_build(void) { return 2; }
int orc_num_channels(void) { return NUM_CHANNELS; }
int orc_num_outputs(void) { return NUM_OUTPUT_CHANNELS; }
int orc_num_pairs(void) { return NUM_SPDIF_INSTANCES; }
int orc_preset_slot_size(void) { return fw_flash_slot_size(); }
void orc_set_math_mode(int detmath) { orc_math_mode = detmath; }
void orc_set_x86_cast_semantics(int on) { (void)on; }      /* compiled casts: always the x86 behaviour */

static int fw_booted;
static void fw_boot(void) { fw_main_start(); fw_service_all(); }      /* main.c:722-736, then a few passes of the loop */

orc_ctx *orc_new(void) {                    /* power-on with an erased flash; one device per library instance */
    unsigned csr = fw_enter();
    if (!fw_booted) { memset(orc_flash_image, 0xFF, orc_flash_bytes); fw_boot(); fw_booted = 1; }
    fw_leave(csr);
    return (orc_ctx *)&fw_booted;
}
void orc_free(orc_ctx *c) { (void)c; }

/* power-on with a given 48 KB preset area (12 sectors: directory, 10 slots, legacy); the library must be fresh */
int orc_boot_from_flash(const void *dump48k, uint32_t len) {
    if (fw_booted || len != 12u * FLASH_SECTOR_SIZE) return -1;
    unsigned csr = fw_enter();
    memset(orc_flash_image, 0xFF, orc_flash_bytes);
    memcpy(orc_flash_image + orc_flash_preset_base, dump48k, len);
    fw_boot(); fw_booted = 1;
    fw_leave(csr);
    return 0;
}
void orc_read_flash(void *dump48k) { memcpy(dump48k, orc_flash_image + orc_flash_preset_base, 12u * FLASH_SECTOR_SIZE); }

int orc_set_sample_rate(orc_ctx *c, uint32_t hz) { /* audio_cmd_packet ENDPOINT_FREQ_CONTROL, usb_audio.c:1491-1498 */
    (void)c;
    if (hz != 44100 && hz != 48000 && hz != 96000) return -1;
    unsigned csr = fw_enter();
    if (audio_state.freq != hz) { audio_state.freq = hz; _audio_reconfigure(); }
    fw_service_all();
    fw_leave(csr);
    return 0;
}
void orc_set_host_volume(orc_ctx *c, int16_t v) { (void)c; unsigned csr = fw_enter(); audio_set_volume(v); fw_leave(csr); }
void orc_set_mute(orc_ctx *c, int mute) { (void)c; audio_state.mute = mute != 0; }

static int fw_request(uint8_t bmRequestType, uint8_t req, uint16_t wValue, uint16_t wLength) {
    struct usb_setup_packet setup __attribute__((aligned(4)));
    memset(&setup, 0, sizeof setup);
    setup.bmRequestType = bmRequestType; setup.bRequest = req; setup.wValue = wValue; setup.wIndex = 2; setup.wLength = wLength;
    fw_pending_out_type = NULL; fw_in_len = -1;
    return vendor_setup_request_handler(&vendor_interface, &setup) ? 1 : 0;
}

int orc_vendor_set(orc_ctx *c, uint8_t req, uint16_t wValue, const void *payload, uint16_t len) {
    (void)c;
    unsigned csr = fw_enter();
    fw_set_payload = (const uint8_t *)payload; fw_set_len = len;
    int ok = fw_request(0x41, req, wValue, len);               /* vendor | interface | host-to-device */
    if (ok && fw_pending_out_type) {                           /* data stage -> vendor_cmd_packet */
        uint16_t n = len < sizeof fw_out_bytes ? len : (uint16_t)sizeof fw_out_bytes;
        memcpy(fw_out_bytes, payload, n); fw_out_buf.data_len = n;
        fw_pending_out_type->on_packet(&usb_control_out);
    }
    fw_service_all();
    fw_leave(csr);
    return ok ? 0 : -1;
}

int orc_vendor_get(orc_ctx *c, uint8_t req, uint16_t wValue, void *buf, uint16_t cap) {
    (void)c;
    unsigned csr = fw_enter();
    fw_in_buf.data_len = 0;
    int ok = fw_request(0xC1, req, wValue, cap);
    int n = -1;
    if (ok && fw_in_len >= 0) { n = fw_in_len; if (n > cap) n = -2; else memcpy(buf, fw_in_bytes, (size_t)n); }
    fw_service_all();
    fw_leave(csr);
    return n;
}

void orc_factory_defaults(orc_ctx *c) { uint8_t r; orc_vendor_get(c, REQ_FACTORY_RESET, 0, &r, 1); }

int orc_load_bulk(orc_ctx *c, const void *blob, uint32_t len) {
    if (len != sizeof(WireBulkParams)) return -4;              /* usb_audio.c:2250-2251: any other length never starts the transfer */
    fw_last_bulk_err = 0;
    if (orc_vendor_set(c, REQ_SET_ALL_PARAMS, 0, blob, (uint16_t)len) != 0) return -4;
    return fw_last_bulk_err;
}
int orc_collect_bulk(orc_ctx *c, void *blob) { return orc_vendor_get(c, REQ_GET_ALL_PARAMS, 0, blob, sizeof(WireBulkParams)); }

/* a slot image is written to its flash sector, marked occupied and loaded through REQ_PRESET_LOAD */
int orc_load_preset_slot(orc_ctx *c, const void *image, uint32_t len, int expect_slot) {
    if ((int)len < fw_flash_slot_size()) return PRESET_ERR_CRC;
    uint16_t idx; memcpy(&idx, (const uint8_t *)image + 6, 2);
    int slot = expect_slot >= 0 ? expect_slot : (idx < PRESET_SLOTS ? idx : 0);
    memset(orc_flash_image + orc_flash_preset_base + (1u + (uint32_t)slot) * FLASH_SECTOR_SIZE, 0xFF, FLASH_SECTOR_SIZE);
    memcpy(orc_flash_image + orc_flash_preset_base + (1u + (uint32_t)slot) * FLASH_SECTOR_SIZE, image, (size_t)fw_flash_slot_size());
    fw_flash_mark_occupied(slot);
    uint8_t r = 0xFF;
    fw_last_preset_status = 0xFF;
    orc_vendor_get(c, REQ_PRESET_LOAD, (uint16_t)slot, &r, 1);     /* IN request, wValue = slot (usb_audio.c:2711-2730) */
    return fw_last_preset_status;
}
int orc_save_preset_slot(orc_ctx *c, void *image, int slot_index) { (void)c; fw_flash_collect_slot(image, slot_index); return fw_flash_slot_size(); }
int orc_load_flash_dump(orc_ctx *c, const void *dump48k, uint32_t len) { (void)c; (void)dump48k; (void)len; return -100; }   /* use orc_boot_from_flash on a fresh instance */

void orc_get_status(orc_ctx *c, void *buf) { orc_vendor_get(c, REQ_GET_STATUS, 9, buf, NUM_CHANNELS * 2 + 4); }

void orc_process(orc_ctx *c, const void *pcm, int bit_depth, uint32_t n_blocks, uint32_t block_len,
                 int32_t *pairs, int32_t *sub, uint16_t *peaks, uint16_t *clip_flags) {
    (void)c;
    unsigned csr = fw_enter();
    const uint32_t bpf = (bit_depth == 24) ? 6 : 4;
    const size_t total = (size_t)n_blocks * block_len;
    usb_input_bit_depth = (uint8_t)bit_depth;                  /* as_set_alternate, usb_audio.c:1581-1585 */
    /* the 24-bit float path reads whole 32-bit words (usb_audio.c:598-609): give it a padded copy */
    static uint8_t pkt[USB_RING_MAX_PKT + 16];
    for (uint32_t k = 0; k < n_blocks; k++) {
        memset(pkt, 0, sizeof pkt);
        memcpy(pkt, (const uint8_t *)pcm + (size_t)k * block_len * bpf, (size_t)block_len * bpf);
        memset(fw_pair_words, 0, sizeof fw_pair_words);
        process_audio_packet(pkt, (uint16_t)(block_len * bpf));
        for (int p = 0; p < NUM_SPDIF_INSTANCES; p++)
            memcpy(pairs + ((size_t)p * total + (size_t)k * block_len) * 2, fw_pair_words[p], (size_t)block_len * 8);
        int32_t *sd = sub + (size_t)k * block_len;
        int got = fw_core1_take_sub(sd, (int)block_len);
        for (uint32_t i = (uint32_t)got; i < block_len; i++) sd[i] = 0;
        if (peaks) memcpy(peaks + (size_t)k * NUM_CHANNELS, (const void *)global_status.peaks, NUM_CHANNELS * 2);
        fw_service_all();
    }
    if (clip_flags) *clip_flags = global_status.clip_flags;
    fw_leave(csr);
}

/* ---- state taps, same numbering as orc_chain.c ---- */
const void *orc_tap(orc_ctx *c, int what, int *bytes) {
    (void)c;
    switch (what) {
        case 0: *bytes = (int)sizeof(filters); return filters;
        case 1: *bytes = (int)(sizeof(LoudnessCoeffs) * LOUDNESS_VOL_STEPS * LOUDNESS_BIQUAD_COUNT); return loudness_active_table;
        case 2: *bytes = (int)sizeof(crossfeed_state); return &crossfeed_state;
        case 3: *bytes = (int)sizeof(leveller_coeffs); return &leveller_coeffs;
        case 4: *bytes = (int)sizeof(matrix_mixer); return &matrix_mixer;
        case 5: *bytes = (int)sizeof(channel_delay_samples); return channel_delay_samples;
        case 6: *bytes = (int)sizeof(leveller_state); return &leveller_state;
        case 7: *bytes = (int)sizeof(filter_recipes); return filter_recipes;
        case 8: *bytes = (int)sizeof(delay_lines); return delay_lines;
        default: *bytes = 0; return NULL;
    }
}
int orc_scalar(orc_ctx *c, int what) {
    (void)c;
    switch (what) {
        case 0: {   /* row of current_loudness_coeffs, whichever of the two table buffers it points into (loudness.c:5-10) */
            extern LoudnessCoeffs loudness_tables[2][LOUDNESS_VOL_STEPS][LOUDNESS_BIQUAD_COUNT];
            if (!current_loudness_coeffs) return -1;
            return (int)(((current_loudness_coeffs - &loudness_tables[0][0][0]) / LOUDNESS_BIQUAD_COUNT) % LOUDNESS_VOL_STEPS);
        }
        case 12:    /* 1 = loudness is on while the coefficient pointer is outside the active table (would be a stale table) */
            return loudness_enabled && current_loudness_coeffs && loudness_active_table &&
                   (current_loudness_coeffs < &loudness_active_table[0][0] || current_loudness_coeffs >= &loudness_active_table[LOUDNESS_VOL_STEPS][0]);
        case 1: return (int)core1_mode;
        case 2: return any_delay_active;
        case 3: return (int)delay_write_idx;
        case 4: return crossfeed_bypassed;
        case 5: return leveller_bypassed;
        case 6: return audio_state.vol_mul;
        case 7: return master_volume_q15;
        case 8: return global_preamp_mul[0];
        case 9: return global_preamp_mul[1];
        case 10: return (int)audio_state.freq;
        case 11: return preset_loading;
        default: return 0;
    }
}
float orc_scalar_f(orc_ctx *c, int what) {
    (void)c;
    switch (what) {
        case 0: return master_volume_linear;
        case 1: return global_preamp_linear[0];
        case 2: return global_preamp_linear[1];
        case 3: return preset_mute_smooth_gain;
        case 4: return master_volume_db;
        default: return 0.0f;
    }
}
