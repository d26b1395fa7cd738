/* Host stand-in for the pioasm-generated header of audio_mck.pio (see audio_i2s_clkout.pio.h). */
#pragma once
#include "pico_stub_all.h"
static const uint16_t audio_mck_program_instructions[1] = {0};
static const struct pio_program audio_mck_program = {audio_mck_program_instructions, 1, -1, 0};
static inline void audio_mck_program_init(PIO pio, uint sm, uint offset, uint pin) { (void)pio; (void)sm; (void)offset; (void)pin; }
