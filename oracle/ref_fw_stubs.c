/*
 * ref_fw_stubs.c — definitions behind ref_stub_sdk/pico_stub_all.h for oracle/_ref/libref_fw_*.so.
 * TEST INFRASTRUCTURE ONLY (see ref_fw.c).  Nothing here is reference code: hardware entry points become no-ops,
 * time stands still, flash is a RAM array, and the few globals main.c / usb_feedback_controller.c would define
 * exist so the reference's translation units link.  The one piece of arithmetic is the Q28 block biquad, which
 * upstream is Thumb assembly (dsp_process_rp2040.S:225-394) and is restated over the reference's own
 * fast_mul_q28 (dsp_pipeline.c:47-58), exactly as orc_chain.c does.
 */
#include <math.h>
#include <string.h>
#include "pico_stub_all.h"
#include "config.h"
#include "usb_audio.h"
#include "usb_feedback_controller.h"
#include "pico/audio_spdif.h"
#include "pico/audio_i2s_multi.h"
#include "pico/usb_device.h"
#include "../include/dspi_detmath.h"

/* ---- flash: 64 KB image, the 48 KB preset area is its tail (flash_storage.c:50-58) ---- */
uint8_t orc_flash_image[PICO_FLASH_SIZE_BYTES];
uint32_t orc_flash_bytes = PICO_FLASH_SIZE_BYTES;
uint32_t orc_flash_preset_base = PICO_FLASH_SIZE_BYTES - 12u * FLASH_SECTOR_SIZE;
void flash_range_erase(uint32_t offs, size_t count) { memset(orc_flash_image + offs, 0xFF, count); }
void flash_range_program(uint32_t offs, const uint8_t *data, size_t count) { memcpy(orc_flash_image + offs, data, count); }
void dspi_flash_range_erase(uint32_t offs, size_t count) { flash_range_erase(offs, count); }          /* flash_clkdiv.c wrappers */
void dspi_flash_range_program(uint32_t offs, const uint8_t *data, size_t count) { flash_range_program(offs, data, count); }
void dspi_flash_apply_clkdiv(void) {}

/* ---- libm hook of leveller.c (compiled with -include ref_math_hook.h, as the _ref build) ---- */
int orc_math_mode = 0;
float orc_hook_log10f(float x) { return orc_math_mode ? dspi_det_log10f(x) : log10f(x); }
float orc_hook_powf(float a, float b) { return orc_math_mode ? dspi_det_powf(a, b) : powf(a, b); }

/* ---- globals of main.c / usb_feedback_controller.c / usb_descriptors.c that the compiled files name ---- */
/* (main.c's own globals come from main.c now: ref_fw_main.c) */
void fb_ctrl_init(usb_feedback_ctrl_t *c) { (void)c; }
void fb_ctrl_reset(usb_feedback_ctrl_t *c, uint32_t v) { (void)c; (void)v; }
void fb_ctrl_stream_stop(usb_feedback_ctrl_t *c) { (void)c; }
void fb_ctrl_sof_update(usb_feedback_ctrl_t *c, uint32_t words, uint32_t shift, uint8_t fill) { (void)c; (void)words; (void)shift; (void)fill; }
uint32_t fb_ctrl_get_10_14(const usb_feedback_ctrl_t *c) { (void)c; return 0; }
volatile uint32_t usb_error_count, usb_crc_error_count, usb_bitstuff_error_count, usb_rx_overflow_count, usb_rx_timeout_count, usb_data_seq_error_count;
pio_hw_t orc_pio_hw[3];
dma_hw_t orc_dma_hw;

/* ---- hardware entry points: no-ops ---- */
uint get_core_num(void) { return 0; }
uint32_t save_and_disable_interrupts(void) { return 0; }
void restore_interrupts(uint32_t s) { (void)s; }
uint32_t spin_lock_blocking(spin_lock_t *l) { (void)l; return 0; }
void spin_unlock(spin_lock_t *l, uint32_t s) { (void)l; (void)s; }
uint32_t time_us_32(void) { return 0; }
static uint64_t orc_clock_us;
uint64_t time_us_64(void) { return orc_clock_us += 100; }      /* a clock that moves: the settle wait before flash writes polls it (main.c:560-567) */
void busy_wait_ms(uint32_t ms) { (void)ms; }
uint32_t clock_get_hz(uint clk) { (void)clk; return 307200000u; }
enum vreg_voltage vreg_get_voltage(void) { return VREG_VOLTAGE_1_15; }
void adc_init(void) {}
void adc_select_input(uint i) { (void)i; }
uint16_t adc_read(void) { return 0; }
void adc_set_temp_sensor_enabled(bool e) { (void)e; }
void gpio_set_dir(uint g, bool o) { (void)g; (void)o; }
void gpio_set_function(uint g, enum gpio_function f) { (void)g; (void)f; }
void irq_set_priority(uint n, uint8_t p) { (void)n; (void)p; }
void reset_usb_boot(uint32_t a, uint32_t b) { (void)a; (void)b; }
void multicore_lockout_victim_init(void) {}
bool multicore_lockout_victim_is_initialized(uint core) { (void)core; return false; }
void multicore_lockout_start_blocking(void) {}
void multicore_lockout_end_blocking(void) {}
uint pio_add_program(PIO pio, const pio_program_t *p) { (void)pio; (void)p; return 0; }
pio_sm_config pio_get_default_sm_config(void) { pio_sm_config c = {0, 0, 0, 0}; return c; }
uint pio_get_dreq(PIO pio, uint sm, bool tx) { (void)pio; (void)sm; (void)tx; return 0; }
void pio_gpio_init(PIO pio, uint pin) { (void)pio; (void)pin; }
void pio_sm_init(PIO pio, uint sm, uint pc, const pio_sm_config *c) { (void)pio; (void)sm; (void)pc; (void)c; }
void pio_sm_set_clkdiv(PIO pio, uint sm, float div) { (void)pio; (void)sm; (void)div; }
void pio_sm_set_consecutive_pindirs(PIO pio, uint sm, uint base, uint count, bool out) { (void)pio; (void)sm; (void)base; (void)count; (void)out; }
void pio_sm_set_enabled(PIO pio, uint sm, bool e) { (void)pio; (void)sm; (void)e; }
void sm_config_set_fifo_join(pio_sm_config *c, enum pio_fifo_join j) { (void)c; (void)j; }
void sm_config_set_out_pins(pio_sm_config *c, uint b, uint n) { (void)c; (void)b; (void)n; }
void sm_config_set_out_shift(pio_sm_config *c, bool r, bool a, uint t) { (void)c; (void)r; (void)a; (void)t; }
void sm_config_set_wrap(pio_sm_config *c, uint t, uint w) { (void)c; (void)t; (void)w; }
int dma_claim_unused_channel(bool required) { (void)required; return 0; }
dma_channel_config dma_channel_get_default_config(uint ch) { (void)ch; dma_channel_config c = {0}; return c; }
void channel_config_set_transfer_data_size(dma_channel_config *c, enum dma_channel_transfer_size s) { (void)c; (void)s; }
void channel_config_set_read_increment(dma_channel_config *c, bool i) { (void)c; (void)i; }
void channel_config_set_write_increment(dma_channel_config *c, bool i) { (void)c; (void)i; }
void channel_config_set_dreq(dma_channel_config *c, uint d) { (void)c; (void)d; }
void channel_config_set_ring(dma_channel_config *c, bool w, uint bits) { (void)c; (void)w; (void)bits; }
void dma_channel_configure(uint ch, const dma_channel_config *c, volatile void *w, const volatile void *r, uint n, bool t) { (void)ch; (void)c; (void)w; (void)r; (void)n; (void)t; }
void dma_channel_abort(uint ch) { (void)ch; }

/* ---- pico-extras output drivers (S/PDIF, I2S): setup calls only ---- */
const audio_format_t *audio_spdif_setup(audio_spdif_instance_t *inst, const audio_format_t *f, const audio_spdif_config_t *c) { (void)inst; (void)c; return f; }
bool audio_spdif_connect_extra(audio_spdif_instance_t *inst, audio_buffer_pool_t *p, bool b, uint n, audio_connection_t *c) { (void)inst; (void)p; (void)b; (void)n; (void)c; return true; }
void audio_spdif_enable_sync(audio_spdif_instance_t *instances[], uint count) { (void)instances; (void)count; }
void audio_spdif_set_enabled(audio_spdif_instance_t *inst, bool e) { (void)inst; (void)e; }
void audio_spdif_change_pin(audio_spdif_instance_t *inst, uint pin) { (void)inst; (void)pin; }
uint32_t audio_spdif_get_dma_starvations(void) { return 0; }
uint32_t audio_spdif_get_dma_starvations_instance(uint i) { (void)i; return 0; }
void audio_spdif_reset_dma_starvations(void) {}
void audio_spdif_set_starvation_monitoring(bool e) { (void)e; }
void audio_i2s_change_data_pin(audio_i2s_instance_t *inst, uint pin) { (void)inst; (void)pin; }
void audio_i2s_mck_change_pin(uint pin) { (void)pin; }
void audio_i2s_mck_set_enabled(bool e) { (void)e; }
void audio_i2s_mck_setup(PIO pio, uint sm, uint pin) { (void)pio; (void)sm; (void)pin; }
void audio_i2s_mck_update_frequency(uint32_t f, uint32_t m) { (void)f; (void)m; }
void audio_i2s_set_enabled(audio_i2s_instance_t *inst, bool e) { (void)inst; (void)e; }
const audio_format_t *audio_i2s_setup(audio_i2s_instance_t *inst, const audio_format_t *f, const audio_i2s_config_t *c) { (void)inst; (void)c; return f; }
bool audio_i2s_connect_extra(audio_i2s_instance_t *inst, audio_buffer_pool_t *p, bool b, uint n, audio_connection_t *c) { (void)inst; (void)p; (void)b; (void)n; (void)c; return true; }
void audio_i2s_teardown(audio_i2s_instance_t *inst) { (void)inst; }
void audio_i2s_enable_sync(audio_i2s_instance_t *instances[], uint count) { (void)instances; (void)count; }
void audio_i2s_update_all_frequencies(uint32_t f) { (void)f; }
void audio_complete_connection(audio_connection_t *c, audio_buffer_pool_t *a, audio_buffer_pool_t *b) { (void)c; (void)a; (void)b; }
void queue_free_audio_buffer(audio_buffer_pool_t *p, audio_buffer_t *b) { (void)p; (void)b; }
audio_buffer_t *get_full_audio_buffer(audio_buffer_pool_t *p, bool block) { (void)p; (void)block; return NULL; }      /* nothing is ever queued here */

/* ---- what main.c's bring-up and pipeline resets touch (main.c:588-720, :230-528) ---- */
static dma_channel_hw_t orc_dma_ch;
dma_channel_hw_t *dma_channel_hw_addr(uint ch) { (void)ch; return &orc_dma_ch; }
void dma_irqn_set_channel_enabled(uint irq, uint ch, bool e) { (void)irq; (void)ch; (void)e; }
void dma_irqn_acknowledge_channel(uint irq, uint ch) { (void)irq; (void)ch; }
void irq_set_enabled(uint n, bool e) { (void)n; (void)e; }
bool irq_is_enabled(uint n) { (void)n; return true; }
int NVIC_GetPriority(int irq) { (void)irq; return 0; }
void gpio_init(uint g) { (void)g; }
void gpio_put(uint g, bool v) { (void)g; (void)v; }
void gpio_xor_mask(uint32_t m) { (void)m; }
bool set_sys_clock_hz(uint32_t hz, bool required) { (void)hz; (void)required; return true; }
void set_sys_clock_pll(uint32_t vco, uint pd1, uint pd2) { (void)vco; (void)pd1; (void)pd2; }
void vreg_set_voltage(enum vreg_voltage v) { (void)v; }
void pio_sm_claim(PIO pio, uint sm) { (void)pio; (void)sm; }
void pio_sm_unclaim(PIO pio, uint sm) { (void)pio; (void)sm; }
void pico_get_unique_board_id_string(char *id, uint len) { if (len) { memset(id, '0', len - 1); id[len - 1] = 0; } }
void multicore_launch_core1(void (*entry)(void)) { (void)entry; }      /* Core 1 is driven by the harness (ref_fw_core1.c) */

#if !PICO_RP2350
/* dsp_process_rp2040.S:225-394 — TDF2 cascade, five inlined Q28 multiplies per sample, band-major (same restatement
 * as orc_chain.c:q28_biquad_block, over the reference's compiled fast_mul_q28). */
#include "dsp_pipeline.h"
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
void dsp_process_channel_block(Biquad *__restrict bands, int32_t *__restrict x, uint32_t count, uint8_t channel) {
    uint8_t nbands = channel_band_counts[channel];
    for (int b = 0; b < nbands; b++) {
        Biquad *q = &bands[b];
        if (q->bypass) continue;
        int32_t s1 = q->s1, s2 = q->s2;
        for (uint32_t i = 0; i < count; i++) {
            int32_t in = x[i];
            int32_t y = wadd(fast_mul_q28(q->b0, in), s1);
            int32_t t1 = fast_mul_q28(q->b1, in);
            int32_t t3 = fast_mul_q28(q->b2, in);
            int32_t t2 = fast_mul_q28(q->a1, y);
            s1 = wadd(wsub(t1, t2), s2);
            int32_t t4 = fast_mul_q28(q->a2, y);
            s2 = wsub(t3, t4);
            x[i] = y;
        }
        q->s1 = s1; q->s2 = s2;
    }
}
int32_t dsp_process_channel(Biquad *__restrict bands, int32_t input_32, uint8_t channel) {
    int32_t x = input_32;
    dsp_process_channel_block(bands, &x, 1, channel);
    return x;
}
#endif
