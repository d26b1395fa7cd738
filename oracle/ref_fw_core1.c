/*
 * ref_fw_core1.c — the reference's pdm_generator.c compiled IN PLACE; Core 1 runs as a coroutine.
 * TEST INFRASTRUCTURE ONLY; part of oracle/_ref/libref_fw_*.so (see ref_fw.c).
 *
 * pdm_generator.c holds the two things Core 1 does: the EQ worker (eq_worker_loop, pdm_generator.c:428-544 float,
 * :551-667 Q28 — the twin of PASS 5-7 for outputs 2..N-2) and the PDM sigma-delta modulator
 * (pdm_processing_loop, :208-397, with its dither :62-108).  Both are `while` loops that sleep in __wfe() and
 * keep their state in locals, so they are run on their own stack (ucontext) and every place the reference
 * waits — __wfe(), or polling the DMA read pointer with an empty sample ring — switches back to the caller:
 *
 *   fw_core1_eq_once()      process_audio_packet's wait for Core 1 (usb_audio.c:866-868) lands here: one pass of
 *                           eq_worker_loop over the posted Core1EqWork, then back.
 *   orc_pdm_ref_run()       feeds sub-channel samples through pdm_push_sample() and resumes pdm_processing_loop
 *                           until it has consumed them; returns the 8 words per sample it wrote to its DMA ring.
 *
 * The DMA engine is emulated by keeping the read pointer a constant 400 words behind the modulator's write
 * pointer (TARGET_LEAD = 256 < 400 < half the ring): no underrun, no overrun, never a "wait for the DMA" branch.
 */
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>
#include <ucontext.h>

static void fw_c1_wait(void);
static void *fw_dma_touch(void);
#define ORC_WFE() fw_c1_wait()
#include "pico_stub_all.h"
#undef dma_hw
#define dma_hw ((dma_hw_t *)fw_dma_touch())
#define volatile(...) ((void)0)            /* the two FPSCR asm statements of pdm_core1_entry (:697-699), never run here */
#define __asm__

#include "pdm_generator.c"

#undef volatile
#undef __asm__

/* ---- coroutine plumbing ---- */
static ucontext_t fw_main_ctx, fw_c1_ctx;
static uint8_t *fw_c1_stack;
static int fw_c1_alive, fw_c1_kind;               /* kind: 1 = EQ worker, 2 = PDM loop */
static Core1Mode fw_saved_mode; static int fw_mode_parked;

static void fw_yield(void) { swapcontext(&fw_c1_ctx, &fw_main_ctx); }
static void fw_c1_wait(void) {
    if (fw_c1_kind == 1) {
        /* eq_worker_loop sleeps between packets: leave it by the door the reference provides — it returns when
         * core1_mode is no longer EQ_WORKER (pdm_generator.c:432-436) — and restore the mode afterwards */
        if (!fw_mode_parked) { fw_saved_mode = core1_mode; fw_mode_parked = 1; core1_mode = CORE1_MODE_IDLE; }
        return;
    }
    fw_yield();
}
static void *fw_dma_touch(void) {
    if (fw_c1_kind == 2 && pdm_dma_chan >= 0) {
        uint32_t base = (uint32_t)(uintptr_t)pdm_dma_buffer;
        uint32_t rd = (pdm_stats_write_idx - 400u) & (PDM_DMA_BUFFER_SIZE - 1);
        orc_dma_hw.ch[pdm_dma_chan].read_addr = base + 4u * rd;
        if (pdm_head == pdm_tail) fw_yield();                          /* nothing to modulate: hand control back */
    }
    return &orc_dma_hw;
}

static void fw_pdm_entry(void) { pdm_processing_loop(); fw_c1_alive = 0; swapcontext(&fw_c1_ctx, &fw_main_ctx); }

static void fw_spawn(void (*fn)(void)) {
    if (!fw_c1_stack) fw_c1_stack = (uint8_t *)malloc(256 * 1024);
    getcontext(&fw_c1_ctx);
    fw_c1_ctx.uc_stack.ss_sp = fw_c1_stack; fw_c1_ctx.uc_stack.ss_size = 256 * 1024; fw_c1_ctx.uc_link = &fw_main_ctx;
    makecontext(&fw_c1_ctx, fn, 0);
}

void fw_core1_eq_once(void) {                      /* the EQ worker returns by itself: a plain call, no second stack */
    if (core1_mode != CORE1_MODE_EQ_WORKER || !core1_eq_work.work_ready) return;
    fw_c1_kind = 1; fw_mode_parked = 0;
    eq_worker_loop();
    if (fw_mode_parked) { core1_mode = fw_saved_mode; fw_mode_parked = 0; }
    fw_c1_kind = 0;
}

/* sub-channel samples posted by process_audio_packet through pdm_push_sample() (usb_audio.c:953-957, :1271-1273) */
int fw_core1_take_sub(int32_t *dst, int max) {
    int n = 0;
    while (pdm_tail != pdm_head && n < max) { dst[n++] = pdm_ring[pdm_tail].sample; pdm_tail++; }
    pdm_tail = pdm_head;
    return n;
}

/* ---- the PDM modulator as a stream: n Q28 samples in, 8 x 32 PDM bits per sample out ---- */
static uint32_t fw_pdm_rd;                         /* next DMA-ring word not yet handed out */
static void fw_pdm_start(int reseed);
void orc_pdm_ref_restart(void) { fw_pdm_start(1); }            /* power-on */
void orc_pdm_ref_restart_keep_rng(void) { fw_pdm_start(0); }   /* re-enable: every local is reset, the file-scope PRNG is not (:63, :241-252) */
static void fw_pdm_start(int reseed) {             /* a fresh entry into pdm_processing_loop = hardware restart (:248-281) */
    if (pdm_dma_chan < 0) pdm_dma_chan = 0;
    pdm_tail = pdm_head = 0;
    if (reseed) rng_state = 123456789;             /* the file-scope PRNG seed (:63) */
    core1_mode = CORE1_MODE_PDM; pdm_enabled = true;
    pdm_stats_write_idx = 0;
    fw_c1_kind = 2; fw_c1_alive = 1;
    fw_spawn(fw_pdm_entry);
    swapcontext(&fw_main_ctx, &fw_c1_ctx);        /* runs into the restart branch up to its first look at the DMA */
    /* the restart branch places the write pointer TARGET_LEAD (256, :221) words ahead of the read pointer (:265-267) */
    fw_pdm_rd = (pdm_stats_write_idx - 400u + 256u) & (PDM_DMA_BUFFER_SIZE - 1);
}
int orc_pdm_ref_dither_seed(uint32_t seed) { rng_state = seed; return 0; }
void orc_pdm_ref_run(const int32_t *sub, uint32_t n, uint32_t *words /*[n][8]*/) {
    uint32_t done = 0;
    while (done < n) {
        uint32_t chunk = n - done; if (chunk > 128) chunk = 128;
        for (uint32_t i = 0; i < chunk; i++) pdm_push_sample(sub[done + i], false);
        swapcontext(&fw_main_ctx, &fw_c1_ctx);
        for (uint32_t i = 0; i < chunk * 8; i++) {
            words[(size_t)(done * 8 + i)] = pdm_dma_buffer[fw_pdm_rd];
            fw_pdm_rd = (fw_pdm_rd + 1) & (PDM_DMA_BUFFER_SIZE - 1);
        }
        done += chunk;
    }
}
