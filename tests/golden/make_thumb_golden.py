#!/usr/bin/env python3
"""Generates tests/golden/q28_thumb_biquad.npz: the RP2040 block biquad (firmware/DSPi/dsp_process_rp2040.S:225-394) EXECUTED — the
reference's assembly text interpreted instruction by instruction (tests/thumb.py) — on seeded coefficient sets, states and samples.
Run in the build container (needs /root/reference):   python tests/golden/make_thumb_golden.py
The vectors pin the oracle's restatement (oracle/orc_chain.c:q28_biquad_block) on machines without the reference."""
import os, sys, struct
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from thumb import Thumb

ASM = "/root/reference/firmware/DSPi/dsp_process_rp2040.S"
CBC, BQ, SAMP = 0x1000, 0x2000, 0x4000


def cases(rng, n_cases):
    """(coef [nb][5], state [nb][2], bypass [nb], x [count]) — ordinary EQ-like coefficients, full-range garbage (every step must wrap
    like the Thumb code), extreme samples, bypassed bands, every packet length class."""
    out = []
    for k in range(n_cases):
        nb = int(rng.integers(1, 11))
        count = int(rng.choice([1, 2, 16, 44, 45, 48, 96, 97]))
        if k % 3 == 0:      # garbage: any int32
            coef = rng.integers(-(1 << 31), 1 << 31, size=(nb, 5), dtype=np.int64).astype(np.int32)
            state = rng.integers(-(1 << 31), 1 << 31, size=(nb, 2), dtype=np.int64).astype(np.int32)
            x = rng.integers(-(1 << 31), 1 << 31, size=count, dtype=np.int64).astype(np.int32)
        else:               # filter-like: |b|, |a| < 2 in Q28, samples within +-2.0
            coef = (rng.uniform(-2, 2, size=(nb, 5)) * (1 << 28)).astype(np.int64).astype(np.int32)
            coef[:, 0] = ((1.0 + rng.uniform(-0.5, 0.5, size=nb)) * (1 << 28)).astype(np.int32)
            state = (rng.uniform(-1, 1, size=(nb, 2)) * (1 << 28)).astype(np.int32)
            x = (rng.uniform(-2, 2, size=count) * (1 << 28)).astype(np.int64).astype(np.int32)
            if k % 5 == 1: x[: min(4, count)] = np.array([0x7FFFFFFF, -0x80000000, 0xFFFF, -0x10000], dtype=np.int64).astype(np.int32)[: min(4, count)]
        bypass = (rng.random(nb) < 0.2).astype(np.uint8)
        out.append((coef, state, bypass, x))
    return out


def run_thumb(t, coef, state, bypass, x, channel=3):
    nb = coef.shape[0]
    for i in range(7): t.mem[CBC + i] = nb if i == channel else 0
    for b in range(nb):
        t.mem[BQ + 32 * b:BQ + 32 * b + 32] = struct.pack("<7iB3x", *[int(v) for v in coef[b]], int(state[b, 0]), int(state[b, 1]), int(bypass[b]))
    t.mem[SAMP:SAMP + 4 * len(x)] = np.ascontiguousarray(x, dtype="<i4").tobytes()
    t.call("dsp_process_channel_block", [BQ, SAMP, len(x), channel])
    y = np.frombuffer(bytes(t.mem[SAMP:SAMP + 4 * len(x)]), dtype="<i4").copy()
    st = np.array([struct.unpack_from("<2i", t.mem, BQ + 32 * b + 20) for b in range(nb)], dtype=np.int32)
    return y, st


def main():
    t = Thumb(open(ASM).read(), symbols={"channel_band_counts": CBC})
    rng = np.random.default_rng(2040)
    cs = cases(rng, 60)
    arrays = {}
    for k, (coef, state, bypass, x) in enumerate(cs):
        y, st = run_thumb(t, coef, state, bypass, x)
        arrays[f"coef{k}"], arrays[f"state{k}"], arrays[f"bypass{k}"], arrays[f"x{k}"], arrays[f"y{k}"], arrays[f"state_out{k}"] = coef, state, bypass, x, y, st
    np.savez_compressed(os.path.join(HERE, "q28_thumb_biquad.npz"), n=len(cs), **arrays)
    print("wrote q28_thumb_biquad.npz:", len(cs), "cases")


if __name__ == "__main__":
    main()
