/* Host stand-in for the pioasm-generated header of audio_i2s_clkout.pio (oracle/ref_i2s.c; test infrastructure only).
 * The PIO program itself is hardware: only the names audio_i2s_multi.c refers to exist here. */
#pragma once
#include "pico_stub_all.h"
#define audio_i2s_clkout_offset_entry_point 0u
static const uint16_t audio_i2s_clkout_program_instructions[1] = {0};
static const struct pio_program audio_i2s_clkout_program = {audio_i2s_clkout_program_instructions, 1, -1, 0};
static inline void audio_i2s_clkout_program_init(PIO pio, uint sm, uint offset, uint data_pin, uint clock_pin_base) {
    (void)pio; (void)sm; (void)offset; (void)data_pin; (void)clock_pin_base;
}
