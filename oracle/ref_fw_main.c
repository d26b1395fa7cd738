/*
 * ref_fw_main.c — the firmware's main() (firmware/DSPi/main.c: core0_init and the main loop with its deferred-apply dispatcher,
 * :588-1171) compiled IN PLACE and run as a coroutine (a hand-rolled stack switch, x86-64).  TEST INFRASTRUCTURE ONLY (oracle/_ref/libref_fw_*.so).
 *
 * main() initialises and then loops forever; the loop starts every iteration with watchdog_update().  That call is the yield
 * point here: fw_main_start() runs main() on its own stack up to the first watchdog_update() (power-on: core0_init, boot preset),
 * fw_main_step() resumes it for exactly one iteration of the reference's loop — ring drain, flash requests, EQ / rate / loudness /
 * crossfeed / leveller recomputation, preset load / save / delete, factory reset, output type switches, bulk parameters — all of it
 * the reference's own code.  Only hardware entry points are stand-ins (ref_fw_stubs.c).
 *
 * What the requests return to their caller is not stored by the firmware (its main loop drops the status of bulk_params_apply and
 * preset_load); the two calls are wrapped so that the oracle API can report them (fw_last_bulk_err, fw_last_preset_status).
 */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include "pico_stub_all.h"
#include "config.h"
#include "dsp_pipeline.h"
#include "flash_clkdiv.h"
#include "flash_storage.h"
#include "pico/audio_i2s_multi.h"
#include "pdm_generator.h"
#include "usb_audio.h"
#include "loudness.h"
#include "crossfeed.h"
#include "leveller.h"
#include "bulk_params.h"
#include "pico/audio_spdif.h"
#include "usb_feedback_controller.h"
#include "pico/audio.h"
#include "hardware/structs/bus_ctrl.h"

int fw_last_bulk_err, fw_last_preset_status;
orc_bus_ctrl_hw_t orc_bus_ctrl_hw;

/* the switch between the harness and main(): callee-saved registers and the stack pointer (System V x86-64); MXCSR is left alone on
 * purpose — the harness sets FTZ|DAZ around every call and main() runs under the same mode.  (swapcontext would do, but its two
 * sigprocmask system calls per switch cost the CPU baseline 16 % of a packet.) */
static void *fw_sp_main, *fw_sp_caller;
__attribute__((naked, noinline)) static void fw_switch(void **save_sp, void *load_sp) {
    __asm__ volatile("pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
                     "movq %rsp, (%rdi)\n movq %rsi, %rsp\n"
                     "popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n ret\n");
}
static int fw_main_running;
void watchdog_update(void) { if (fw_main_running) fw_switch(&fw_sp_main, fw_sp_caller); }     /* top of every main-loop iteration */
void watchdog_enable(uint32_t ms, bool pause) { (void)ms; (void)pause; }

#define bulk_params_apply(p, pins) (fw_last_bulk_err = (bulk_params_apply)((p), (pins)))
#define preset_load(slot) (fw_last_preset_status = (preset_load)(slot))
#define preset_save(slot) (fw_last_preset_status = (preset_save)(slot))
#define __asm__                                   /* the two FPSCR accesses (main.c:597-599): the harness sets FTZ|DAZ around every call */
#define volatile(...) ((void)0)
#define main fw_main_entry
#define printf(...) ((void)0)                      /* the firmware logs to its UART */
#include "main.c"
#undef main
#undef printf
#undef volatile
#undef __asm__

static char fw_main_stack[1 << 20] __attribute__((aligned(16)));
static void fw_main_tramp(void) { fw_main_entry(); abort(); }      /* main() never returns */
/* power-on: main() up to the top of its loop */
void fw_main_start(void) {
    void **sp = (void **)(fw_main_stack + sizeof fw_main_stack);
    *--sp = NULL;                                  /* where fw_main_tramp would return to: keeps its frame 16-byte aligned */
    *--sp = (void *)fw_main_tramp;                 /* popped by the switch's ret */
    for (int i = 0; i < 6; i++) *--sp = NULL;      /* rbp rbx r12 r13 r14 r15 */
    fw_sp_main = sp;
    fw_main_running = 1;
    fw_switch(&fw_sp_caller, fw_sp_main);
}
/* one iteration of the firmware's main loop */
void fw_main_step(void) { fw_switch(&fw_sp_caller, fw_sp_main); }
