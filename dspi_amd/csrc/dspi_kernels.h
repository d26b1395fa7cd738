// dspi_kernels.h — launch interface between the context (dspi_capi.cpp) and dspi_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>
#include "dspi_image.h"

namespace dspi {

constexpr int kChunk = 16;   // frames per in-kernel chunk (divides the 48- and 96-frame packets)

struct KArgs {
    const DevImage *img;     // the context's image array; a workgroup uses img[item.image] (scalar loads: its lanes share it)
    const WgItem *items;     // workgroups of this launch: row, image, lane mask
    uint32_t *state;         // [n_wg][n_slots][64]
    uint32_t *dlines;        // [n_wg][n_out][max_delay][64]
    uint32_t *ring;          // [n_wg][kRingLen][2][64]
    const void *pcm;         // [stream][n_blocks*block_len] frames, 4 or 6 bytes each
    int32_t *pairs;          // [stream][pair][frames][2] or null
    int32_t *sub;            // [stream][frames] or null
    uint16_t *peaks;         // [stream][block][C] or null
    uint32_t n_streams, n_blocks, block_len, bit_depth;
    const uint32_t *stream_image;   // [n_streams] image index of every stream (float one-stream kernel: per-lane parameters)
    uint32_t tiled_out;      // DSPI_OUT_TILED: pairs = [tile][output][frames][row], sub = [tile][frames][row] (row = StateMap::row)
    uint32_t *xwords;        // packed float kernel, stream-major output: mini lines [n_wg][3][kMaxOut][kChunk][128] words — the rows of outputs that do not reach their pair's waves through the delay line (dspi_chain_pk.inc output_item_pk)
    const float *vals;       // packed float kernel, per-lane values: value tiles [n_wg][kPvTileFloats] (dspi_image.h) or null
    uint32_t pairs_stream0;  // stream-major `pairs` starts at this stream (0: the caller's whole buffer; the two-pass S/PDIF path hands the kernels a scratch buffer that
                             // holds a chunk of rows)
    uint32_t skip_silent;    // DSPI_OUT_ENABLED_ONLY: sample words of silent outputs (a disabled S/PDIF pair, the sub while it is off) need not be stored
    uint32_t i2s_slots;      // DSPI_OUT_I2S_SLOTS: pairs whose slot is an I2S slot (DevImage::i2s_pairs) carry left-justified I2S words (word << 8)
    uint32_t spdif;          // DSPI_OUT_SPDIF (latency layout only): `pairs` takes IEC 60958 subframes, uint32 [stream][pair][frame][4]
    uint32_t spdif_pos;      // position of the launch's first frame in the 192-frame block (the channel status follows each image's own fs_hz)
    uint32_t fma;            // float flavour: the context's contract is DSPI_FLOAT_CONTRACT_FMA (selects the kernel family at launch)
};

size_t chain_lds_bytes(int flavor, int packed);
// packed: 1 = packed float kernel (items list lanes whose two streams share the item's image); 0 = one stream per lane
// with the item's image for the whole workgroup (Q28); 2 = one stream per lane, every lane its own image
// (args.stream_image; float: stream WgItem::image (0 / 1) of each listed lane; Q28: rows with several presets)
// packed 3 = packed float kernel with per-lane VALUES: one item per row whose streams share a structure, WgItem::image = any
// image of the row (read for the structure only), numbers from args.vals; packed 4 = the same for rows whose presets have identical
// FILTERS (band coefficients from the image's scalars, everything else from args.vals)
// packed 5 / 6 = the latency layout of the float chain (dspi_chain_skew.inc, dspi_chain_skew_lev.inc): items as for packed 1, launches small
// enough to leave the chip underfilled (dspi_capi.cpp decides).  5 with leveller_on false: images without an active output EQ; 6: with
// one; 5 with leveller_on true: images with the leveller on (the third shape)
// leveller_on: IF_LEVELLER_ON of every image in args.items (the host groups them; the packed kernel is specialised on it)
hipError_t launch_chain(int flavor, int packed, bool leveller_on, const KArgs &args, uint32_t n_items, hipStream_t stream);
hipError_t launch_state_ops(int flavor, const WgItem *items, uint32_t n_items, const StateOps &ops, uint32_t *state, uint32_t *dlines,
                            uint32_t *ring, uint32_t n_streams, hipStream_t stream);
// (re)build the value tiles of the listed rows from the images of their streams
// all_differ: development switch, every band is treated as different between the row's streams (the worst case, for timing)
hipError_t launch_pv_build(const DevImage *img, const uint32_t *stream_image, const uint32_t *rows, uint32_t n_rows, float *vals, uint32_t n_streams, bool all_differ,
                           hipStream_t stream);
hipError_t launch_state_init(int flavor, uint32_t *state, uint32_t n_wg, hipStream_t stream);
// debug: taps [kBands+1][n] after every band of EQ channel `ch` of *img (float flavour), other [kBands][n] = the other
// contract's one-step result from the same input and state
hipError_t launch_eq_taps(bool fma, const DevImage *img, int ch, const float *x, uint32_t n, float *taps, float *other, hipStream_t stream);

// ---- PDM sub output (dspi_pdm.hip): per-stream state [n_wg][kPdmStateWords][row]: err err2 x1 x2 y1 y2 err_acc rng fade_in_pos
constexpr int kPdmStateWords = 9;
hipError_t launch_pdm(bool tiled, uint32_t *state, const int32_t *sub, uint32_t *words, uint32_t n_streams, uint32_t n_frames, uint32_t row,
                      uint32_t n_wg, hipStream_t stream);
hipError_t launch_pdm_reset(uint32_t *state, uint32_t n_streams, uint32_t row, uint32_t n_wg, int32_t only_stream, int init, hipStream_t stream);

// ---- S/PDIF subframe encoder (dspi_spdif.hip)
// The sample-rate byte of the channel status (audio_spdif.c:250-256) is a property of the DEVICE: streams of one context may run at
// different rates.  stream_image == nullptr: every stream at `fs`; else stream s reads img[stream_image[stream0 + s]].fs_hz.
struct SpdifRates { const DevImage *img; const uint32_t *stream_image; uint32_t stream0; };
hipError_t launch_spdif(bool tiled, const int32_t *pairs, uint32_t *out, uint32_t n_streams, uint32_t n_pairs, uint32_t n_frames, uint32_t row,
                        uint32_t n_wg, uint32_t block_pos, uint32_t fs, const SpdifRates &rates, hipStream_t stream);
// I2S slots (audio_i2s_multi.c:217-226): words << 8 for the pairs in pair_mask; same layouts as the pair words themselves
hipError_t launch_i2s(bool tiled, const int32_t *pairs, uint32_t *out, uint32_t n_streams, uint32_t n_pairs, uint32_t n_frames, uint32_t row,
                      uint32_t n_wg, uint32_t pair_mask, hipStream_t stream);

// ---- status at scale (dspi_status.hip): out[s] = OR of stream s's four sticky clip slots (state slots clip_slot .. clip_slot + 3)
hipError_t launch_detmath(int which, const float *a, const float *b, uint32_t n, float *out, hipStream_t stream);      // dspi_status.hip
hipError_t launch_clip_gather(const uint32_t *state, uint32_t n_streams, uint32_t row, uint32_t n_slots, uint32_t clip_slot, uint16_t *out, hipStream_t stream);

}  // namespace dspi
