/* oracle/ref_stub/pico/audio.h — NOT a copy of pico-extras' header: a five-line stand-in that lets the reference's
 * pico/audio_spdif/sample_encoding.h (which only needs these names) compile on the host, in place, for oracle/_ref. */
#ifndef ORC_STUB_PICO_AUDIO_H
#define ORC_STUB_PICO_AUDIO_H
#include <stdint.h>
typedef unsigned int uint;
typedef struct audio_connection audio_connection_t;
typedef struct audio_buffer audio_buffer_t;
#define __mul_instruction(a, b) ((a) * (b))      /* pico/platform: a plain 32-bit multiply */
#endif
