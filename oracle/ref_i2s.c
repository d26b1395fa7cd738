/*
 * ref_i2s.c — the reference's I2S producer-give (pico_audio_i2s_multi/audio_i2s_multi.c:198-243, static) compiled IN PLACE
 * over the stub pico-sdk, with the pools it talks to replaced by a capture list.  TEST INFRASTRUCTURE ONLY: pins the
 * restated orc_i2s_frames() (orc_spdif.c) and, through it, the product's dspi_i2s_encode.
 *
 * Output: oracle/_ref/libref_i2s.so (Makefile target `ref`).  The three pioasm-generated headers the file includes are
 * host stand-ins (ref_stub_sdk/audio_*.pio.h): the PIO programs are hardware, the conversion under test is C.
 */
#include <stdlib.h>
#include <string.h>
#include "pico_stub_all.h"
#ifndef __isr
#define __isr
#endif
#ifndef PICO_SHARED_IRQ_HANDLER_DEFAULT_ORDER_PRIORITY
#define PICO_SHARED_IRQ_HANDLER_DEFAULT_ORDER_PRIORITY 0x80
#endif
uint pio_get_index(PIO pio);
void pio_clkdiv_restart_sm_mask(PIO pio, uint32_t mask);
void dma_irqn_set_channel_enabled(uint irq, uint ch, bool e);
bool dma_irqn_get_channel_status(uint irq, uint ch);
void dma_irqn_acknowledge_channel(uint irq, uint ch);
void __mem_fence_release(void);

#include "audio_i2s_multi.c"

/* ---- names the file refers to (main.c / usb_audio.c globals, hardware): inert ---- */
int overruns;
volatile bool preset_loading;
volatile uint32_t pio_samples_dma;
pio_hw_t orc_pio_hw[3];
uint get_core_num(void) { return 0; }
uint32_t save_and_disable_interrupts(void) { return 0; }
void restore_interrupts(uint32_t s) { (void)s; }
uint32_t clock_get_hz(uint clk) { (void)clk; return 307200000u; }
void __mem_fence_release(void) {}
void gpio_set_dir(uint g, bool o) { (void)g; (void)o; }
void gpio_set_function(uint g, enum gpio_function f) { (void)g; (void)f; }
void irq_set_enabled(uint n, bool e) { (void)n; (void)e; }
void irq_add_shared_handler(uint n, irq_handler_t h, uint8_t p) { (void)n; (void)h; (void)p; }
uint pio_get_index(PIO pio) { (void)pio; return 0; }
uint pio_add_program(PIO pio, const pio_program_t *p) { (void)pio; (void)p; return 0; }
uint pio_get_dreq(PIO pio, uint sm, bool tx) { (void)pio; (void)sm; (void)tx; return 0; }
uint pio_encode_jmp(uint addr) { return addr; }
void pio_gpio_init(PIO pio, uint pin) { (void)pio; (void)pin; }
void pio_sm_claim(PIO pio, uint sm) { (void)pio; (void)sm; }
void pio_sm_unclaim(PIO pio, uint sm) { (void)pio; (void)sm; }
void pio_sm_restart(PIO pio, uint sm) { (void)pio; (void)sm; }
void pio_sm_exec(PIO pio, uint sm, uint i) { (void)pio; (void)sm; (void)i; }
void pio_sm_clear_fifos(PIO pio, uint sm) { (void)pio; (void)sm; }
void pio_sm_set_enabled(PIO pio, uint sm, bool e) { (void)pio; (void)sm; (void)e; }
void pio_sm_set_clkdiv_int_frac(PIO pio, uint sm, uint16_t a, uint8_t b) { (void)pio; (void)sm; (void)a; (void)b; }
void pio_enable_sm_mask_in_sync(PIO pio, uint32_t m) { (void)pio; (void)m; }
void pio_clkdiv_restart_sm_mask(PIO pio, uint32_t m) { (void)pio; (void)m; }
void dma_irqn_set_channel_enabled(uint irq, uint ch, bool e) { (void)irq; (void)ch; (void)e; }
bool dma_irqn_get_channel_status(uint irq, uint ch) { (void)irq; (void)ch; return false; }
void dma_irqn_acknowledge_channel(uint irq, uint ch) { (void)irq; (void)ch; }
void dma_channel_abort(uint ch) { (void)ch; }
void dma_channel_claim(uint ch) { (void)ch; }
void dma_channel_unclaim(uint ch) { (void)ch; }
dma_channel_config dma_channel_get_default_config(uint ch) { (void)ch; dma_channel_config c = {0}; return c; }
void channel_config_set_dreq(dma_channel_config *c, uint d) { (void)c; (void)d; }
void dma_channel_configure(uint ch, const dma_channel_config *c, volatile void *w, const volatile void *r, uint n, bool t) { (void)ch; (void)c; (void)w; (void)r; (void)n; (void)t; }
void dma_channel_transfer_from_buffer_now(uint ch, const volatile void *r, uint32_t n) { (void)ch; (void)r; (void)n; }

/* ---- pico_audio pools (common/pico_audio/audio.cpp): a free list of ONE staging buffer and a capture list ---- */
static audio_buffer_t cap_buf; static mem_buffer_t cap_mem;
static uint32_t *cap_out; static size_t cap_words, cap_cap; static uint32_t cap_full_buffers;
audio_buffer_t *get_free_audio_buffer(audio_buffer_pool_t *p, bool block) { (void)p; (void)block; return &cap_buf; }
void queue_full_audio_buffer(audio_buffer_pool_t *p, audio_buffer_t *b) {          /* what the DMA would send next */
    (void)p;
    size_t n = (size_t)b->sample_count * 2;
    if (cap_words + n <= cap_cap) memcpy(cap_out + cap_words, b->buffer->bytes, n * 4);
    cap_words += n; cap_full_buffers++;
}
void queue_free_audio_buffer(audio_buffer_pool_t *p, audio_buffer_t *b) { (void)p; (void)b; }
void give_audio_buffer(audio_buffer_pool_t *p, audio_buffer_t *b) { (void)p; (void)b; }
audio_buffer_t *take_audio_buffer(audio_buffer_pool_t *p, bool block) { (void)p; (void)block; return NULL; }
audio_buffer_t *producer_pool_take_buffer_default(audio_connection_t *c, bool block) { (void)c; (void)block; return NULL; }
audio_buffer_t *consumer_pool_take_buffer_default(audio_connection_t *c, bool block) { (void)c; (void)block; return NULL; }
void consumer_pool_give_buffer_default(audio_connection_t *c, audio_buffer_t *b) { (void)c; (void)b; }
audio_buffer_pool_t *audio_new_consumer_pool(audio_buffer_format_t *f, int n, int samples) { (void)f; (void)n; (void)samples; return NULL; }
void audio_free_buffer_pool(audio_buffer_pool_t *p) { (void)p; }
void audio_complete_connection(audio_connection_t *c, audio_buffer_pool_t *a, audio_buffer_pool_t *b) { (void)c; (void)a; (void)b; }

/* pairs: int32 [n_frames][2] as process_audio_packet leaves them in a producer buffer, handed over in packets of `packet`
 * frames; consumer buffers hold `consumer_len` frames.  out receives every COMPLETED consumer buffer back to back; the
 * return value is the number of frames in them (the remainder stays in the staging buffer, as in the firmware). */
uint32_t orc_i2s_ref_give(const int32_t *pairs, uint32_t n_frames, uint32_t packet, uint32_t consumer_len, uint32_t *out) {
    static audio_buffer_pool_t producer, consumer;
    struct producer_pool_blocking_give_connection conn;
    memset(&conn, 0, sizeof conn);
    conn.core.producer_pool = &producer; conn.core.consumer_pool = &consumer;
    cap_mem.bytes = (uint8_t *)malloc((size_t)consumer_len * 8); cap_mem.size = consumer_len * 8;
    cap_buf.buffer = &cap_mem; cap_buf.max_sample_count = consumer_len; cap_buf.sample_count = 0;
    cap_out = out; cap_words = 0; cap_cap = (size_t)n_frames * 2; cap_full_buffers = 0;
    mem_buffer_t pm; audio_buffer_t pb; memset(&pb, 0, sizeof pb);
    pb.buffer = &pm;
    for (uint32_t f = 0; f < n_frames; f += packet) {
        uint32_t n = n_frames - f < packet ? n_frames - f : packet;
        pm.bytes = (uint8_t *)(pairs + (size_t)f * 2); pm.size = n * 8;
        pb.sample_count = n; pb.max_sample_count = n;
        i2s_wrap_producer_give(&conn.core, &pb);
    }
    free(cap_mem.bytes);
    return (uint32_t)(cap_words / 2);
}
