/* Host stand-in for the pioasm-generated header of audio_i2s_dataout.pio (see audio_i2s_clkout.pio.h). */
#pragma once
#include "pico_stub_all.h"
#define audio_i2s_dataout_offset_entry_point 0u
static const uint16_t audio_i2s_dataout_program_instructions[1] = {0};
static const struct pio_program audio_i2s_dataout_program = {audio_i2s_dataout_program_instructions, 1, -1, 0};
static inline void audio_i2s_dataout_program_init(PIO pio, uint sm, uint offset, uint data_pin) { (void)pio; (void)sm; (void)offset; (void)data_pin; }
