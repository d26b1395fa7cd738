#!/usr/bin/env python3
"""Generates tests/golden/*.npz from oracle/_ref — the reference's own leaf C sources (dsp_pipeline.c, leveller.c,
crossfeed.c, loudness.c, bulk_params.c) compiled in place under the restated orchestrator.  Run in the build container
(needs /root/reference):   python tests/golden/make_golden.py [case ...]
The vectors pin the standalone restatement (tests/test_oracle_golden.py) on machines without the reference.
Each file stores the exact inputs (parameter blob + PCM seed description) and, per output array, its CRC-32, length,
and the first/last 96 frames.
"""
import os, sys, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orclib import Oracle
from dspi_amd import wire as W, workloads as WL

CASES = [
    # name, flavor, fs, block_len, blocks, blob builder, volume (1/256 dB), bit depth, detmath, first_stream
    ("f32_config2_svf_biquad", 1, 48000, 48, 40, lambda: WL.config2_blob(False), -10 * 256, 16, False, 0),
    ("f32_config2_all_biquad", 1, 48000, 48, 40, lambda: WL.config2_blob(True), 0, 16, False, 1),
    ("f32_full_96k_libm", 1, 96000, 96, 40, lambda: WL.full_chain_blob(1), -20 * 256, 16, False, 3),
    ("f32_full_96k_detmath", 1, 96000, 96, 40, lambda: WL.full_chain_blob(1), -20 * 256, 16, True, 3),
    ("f32_full_441_24bit_detmath", 1, 44100, 45, 30, lambda: WL.full_chain_blob(1), -6 * 256, 24, True, 16),
    ("f32_full_96k_burst_detmath", 1, 96000, 96, 40, lambda: WL.full_chain_blob(1), -20 * 256, 16, True, 16),
    ("q28_config1_vol-10", 0, 48000, 48, 40, WL.config1_blob, -10 * 256, 16, False, 0),
    ("q28_config1_vol0_signquirk", 0, 48000, 48, 40, WL.config1_blob, 0, 16, False, 0),
    ("q28_full_48k_detmath", 0, 48000, 48, 40, lambda: WL.full_chain_blob(0), -20 * 256, 16, True, 5),
    ("q28_full_48k_square_detmath", 0, 48000, 48, 40, lambda: WL.full_chain_blob(0), -3 * 256, 16, True, 19),
    # The Q28 limiter (leveller.c:366-383) ENGAGED with its float->int cast in range, so that the x86 build of the reference is a valid
    # checker for it (q28_full_48k_detmath above is the documented divergence: quiet noise samples overflow that cast).  A square wave
    # keeps every sample at one magnitude: 0.7 s at -26 dBFS (boosted 9 %, the limiter evaluates ceil/peak = 7.07 < 8 on every sample
    # and leaves the gain alone), then -6 dBFS (gain still above 1 while it decays: the limiter caps it to unity on every sample).
    # Master EQ, loudness and preamp are flat in this preset — ringing would put near-zero samples next to a gain above 1.
    ("q28_limiter_in_range", 0, 48000, 48, 760, "limiter_blob", -20 * 256, 16, True, "limiter_pcm"),
    # the firmware's float contract: the same leaf sources compiled with GCC's contraction (oracle/_ref/libref_f32_fma.so)
    ("f32fma_full_96k_detmath", W.F32_FMA, 96000, 96, 40, lambda: WL.full_chain_blob(1), -20 * 256, 16, True, 3),
    ("f32fma_full_441_24bit_detmath", W.F32_FMA, 44100, 45, 30, lambda: WL.full_chain_blob(1), -6 * 256, 24, True, 16),
    ("f32fma_config2_svf_biquad", W.F32_FMA, 48000, 48, 40, lambda: WL.config2_blob(False), -10 * 256, 16, False, 0),
]


def limiter_blob():
    b = WL.full_chain_blob(0).copy()
    b["eq"]["type"][0:2] = 0
    b["global_"]["loudness_enabled"] = 0
    b["preamp"]["preamp_db"][:] = 0.0
    b["global_"]["preamp_gain_db"] = 0.0
    return b


def limiter_pcm(n):
    quiet, loud = int(32767 * 10 ** (-26 / 20)), int(32767 * 10 ** (-6 / 20))
    amp = np.where(np.arange(n) < 700 * 48, quiet, loud)
    sq = (np.where((np.arange(n) // 24) % 2 == 0, 1, -1) * amp).astype(np.int16)
    return np.stack([sq, -sq], axis=-1)


def summarise(a: np.ndarray):
    flat = np.ascontiguousarray(a)
    return np.uint32(zlib.crc32(flat.tobytes()) & 0xFFFFFFFF)


def main():
    only = sys.argv[1:]
    for name, flavor, fs, B, blocks, mk, vol, depth, detmath, first in CASES:
        if only and name not in only: continue
        blob = globals()[mk]() if isinstance(mk, str) else mk()
        # x86_casts=False: the firmware's saturating float->int conversions (see oracle/orc_common.h); the _ref objects
        # themselves are x86 builds, so cases are chosen where no conversion overflows (checked below against x86 mode).
        o = Oracle(flavor, ref=True, detmath=detmath)
        assert o.set_rate(fs) == 0
        o.set_volume(vol)
        assert o.load_bulk(blob) == 0
        pcm = globals()[first](B * blocks) if isinstance(first, str) else WL.synth_pcm16(1, B * blocks, fs, first_stream=first)[0]
        data = pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm[None])[0]
        pairs, sub, peaks, clip = o.process(data, blocks, B, depth)
        status = np.frombuffer(o.status(), dtype=np.uint8)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            flavor=int(flavor), fma=int(bool(getattr(flavor, "fma", False))), fs=fs, block_len=B, blocks=blocks, volume=vol, bit_depth=depth, detmath=int(detmath), first_stream=(-1 if isinstance(first, str) else first),
            blob=np.frombuffer(blob.tobytes(), dtype=np.uint8), pcm=data,
            pairs_crc=summarise(pairs), sub_crc=summarise(sub), peaks_crc=summarise(peaks), clip=np.uint16(clip), status=status,
            pairs_head=pairs[:, :96], pairs_tail=pairs[:, -96:], sub_head=sub[:96], sub_tail=sub[-96:], peaks=peaks)
        print(f"{name}: pairs crc {summarise(pairs):08x} sub crc {summarise(sub):08x} clip {clip:#x} nonzero {np.count_nonzero(pairs)}/{pairs.size}")


if __name__ == "__main__":
    main()
