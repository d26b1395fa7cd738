"""The restated oracle (oracle/orc_chain.c) against the FIRMWARE BUILD of the reference (oracle/ref_fw.c): usb_audio.c's own
`process_audio_packet`, vendor SET/GET handlers and volume code, flash_storage.c, pdm_generator.c's Core-1 EQ worker and
main.c's main loop (deferred-apply dispatcher, preset and type-switch handling), compiled in place from /root/reference over a stub pico-sdk.  Same calls into both, every output word / peak / status byte /
parameter byte compared.  This is what pins the hand restatement of the orchestrator, the control surface and the preset
code to the reference itself.  Runs where oracle/_ref/libref_fw_*.so exists (needs /root/reference to build); skipped
elsewhere — the golden fixtures carry the pin to other machines."""
import struct

import numpy as np
import pytest

import orclib
from orclib import Oracle
from dspi_amd import wire as W, workloads as WL

pytestmark = pytest.mark.skipif(not (orclib.ref_available(1, "fw") and orclib.ref_available(0, "fw") and orclib.ref_available(1, "fw", True)),
                                reason="oracle/_ref/libref_fw_* not built (needs /root/reference)")

# (flavour, float contract): canonical float, float as the firmware is built (FMA contraction, oracle/orc_leaf.c), Q28
FL = [(1, False), (1, True), (0, False)]

CASES = [(1, 48000, 48, 16, False), (1, 96000, 96, 16, False), (1, 44100, 45, 24, False), (1, 44100, 44, 16, False), (1, 96000, 97, 24, False),
         (1, 48000, 48, 24, True), (1, 96000, 96, 16, True), (1, 44100, 45, 16, True), (1, 44100, 44, 24, True),
         (0, 48000, 48, 16, False), (0, 96000, 96, 24, False), (0, 44100, 45, 16, False), (0, 44100, 44, 24, False)]


def pair(flavor, detmath=True, fma=False):
    return Oracle(flavor, ref=False, detmath=detmath, x86_casts=True, fma=fma), Oracle(flavor, ref="fw", detmath=detmath, fma=fma)


def same_audio(a, b, data, blocks, B, depth, what=""):
    ra, rb = a.process(data, blocks, B, depth), b.process(data, blocks, B, depth)
    for name, x, y in zip(("pairs", "sub", "peaks"), ra[:3], rb[:3]):
        assert np.array_equal(x, y), f"{what}: {name} differ at {np.argwhere(x != y)[:4].tolist()}"
    assert ra[3] == rb[3], f"{what}: clip flags"
    assert a.status() == b.status(), f"{what}: status block"


def same_state(a, b, what=""):
    assert a.collect_bulk() == b.collect_bulk(), f"{what}: bulk blob"
    for t in range(9):
        assert a.tap(t) == b.tap(t), f"{what}: state tap {t}"
    loud_on = a.vendor_get(W.REQ["GET_LOUDNESS"], 0) == b"\x01"
    for k in range(12):
        if k == 0 and not loud_on: continue      # the row pointer is only read while loudness is on (usb_audio.c:579-580, :693)
        assert a.scalar(k) == b.scalar(k), f"{what}: scalar {k}"
    assert b.scalar(12) == 0, f"{what}: firmware would filter with a stale loudness table"
    for k in range(5):
        assert a.scalar_f(k) == b.scalar_f(k) or (np.isnan(a.scalar_f(k)) and np.isnan(b.scalar_f(k))), f"{what}: float scalar {k}"


def signal(B, blocks, fs, first, depth):
    pcm = WL.synth_pcm16(1, B * blocks, fs, first_stream=first)[0]
    return pcm if depth == 16 else WL.pcm16_to_pcm24_bytes(pcm[None])[0]


@pytest.mark.parametrize("flavor,fs,B,depth,fma", CASES)
def test_full_chain_is_process_audio_packet(flavor, fs, B, depth, fma):
    a, b = pair(flavor, fma=fma)
    same_state(a, b, "power-on")
    blob = WL.full_chain_blob(flavor)
    for o in (a, b):
        assert o.set_rate(fs) == 0
        o.set_volume(-12 * 256)
        assert o.load_bulk(blob) == 0
    same_state(a, b, "after blob")
    for first in (2, 15, 16, 18, 19):       # noise, sweep, bursts, silence tail, full-scale square
        same_audio(a, b, signal(B, 25, fs, first, depth), 25, B, depth, f"stream class {first}")
    same_state(a, b, "after audio")


@pytest.mark.parametrize("flavor", (1, 0))
def test_glibc_math_and_sign_quirk_volume(flavor):
    """leveller per-packet log10f/powf through glibc in both builds; host volume 0 dB (vol_mul = -32768, usb_audio.c:410-434)."""
    a, b = pair(flavor, detmath=False)
    for o in (a, b):
        assert o.set_rate(48000) == 0; o.set_volume(0); assert o.load_bulk(WL.full_chain_blob(flavor)) == 0
    same_audio(a, b, signal(48, 40, 48000, 16, 16), 40, 48, 16, "0 dB host volume")
    for o in (a, b): o.set_mute(True)
    same_audio(a, b, signal(48, 4, 48000, 3, 16), 4, 48, 16, "muted")


@pytest.mark.parametrize("flavor,fma", FL)
def test_core1_eq_worker_twin(flavor, fma):
    """Sub output off, outputs 2.. on: the reference hands outputs 2..N-2 to Core 1 (usb_audio.c:782-872,
    pdm_generator.c:428-667 eq_worker_loop); the firmware build runs that loop, the oracle its single restatement."""
    a, b = pair(flavor, fma=fma)
    blob = WL.full_chain_blob(flavor)
    N = a.N
    blob["outputs"][N - 1]["enabled"] = 0
    for o in (a, b):
        assert o.set_rate(96000) == 0; o.set_volume(-6 * 256); assert o.load_bulk(blob) == 0
    assert a.scalar(1) == b.scalar(1) == 2          # CORE1_MODE_EQ_WORKER
    for first in (2, 16):
        same_audio(a, b, signal(96, 30, 96000, first, 16), 30, 96, 16, "EQ worker")
    same_state(a, b, "EQ worker")
    # enabling the sub while outputs 2.. are on is refused (usb_audio.c:1891-1904); disabling them flips the mode
    for o in (a, b):
        o.vendor_set(W.REQ["SET_OUTPUT_ENABLE"], N - 1, b"\x01")
    same_state(a, b, "refused sub enable")
    for o in (a, b):
        for out in range(2, N - 1):
            o.vendor_set(W.REQ["SET_OUTPUT_ENABLE"], out, b"\x00")
        o.vendor_set(W.REQ["SET_OUTPUT_ENABLE"], N - 1, b"\x01")
    same_state(a, b, "sub enabled")
    same_audio(a, b, signal(96, 10, 96000, 5, 16), 10, 96, 16, "PDM mode")


@pytest.mark.parametrize("flavor,fma", FL)
@pytest.mark.parametrize("seed", range(8))
def test_random_presets(flavor, fma, seed):
    from test_gpu_fuzz import random_blob, RATES
    rng = np.random.default_rng(31000 + 100 * flavor + seed)
    fs, Bs = RATES[seed % 3]
    B = int(rng.choice(Bs)); depth = 16 if rng.random() < 0.5 else 24
    a, b = pair(flavor, fma=fma)
    blob = random_blob(rng, flavor, fs)
    for o in (a, b):
        assert o.set_rate(fs) == 0; o.set_volume(int(rng.choice([0, -5 * 256, -20 * 256, 3 * 256]))) if False else None
    vol = int(rng.choice([0, -5 * 256, -20 * 256, -40 * 256, 3 * 256]))
    for o in (a, b):
        o.set_volume(vol); assert o.load_bulk(blob) == 0
    same_state(a, b, "random blob")
    same_audio(a, b, signal(B, 30, fs, int(rng.integers(0, 40)), depth), 30, B, depth, "random blob")


@pytest.mark.parametrize("flavor,fma", FL)
@pytest.mark.parametrize("seed", range(8))
def test_random_request_sequences(flavor, fma, seed):
    """Random vendor SETs (in and out of range), UAC1 volume / mute, blobs, preset loads, factory resets and a rate change,
    each followed by audio: the reference's vendor_cmd_packet + main-loop semantics vs orc_vendor_set."""
    from test_gpu_fuzz import random_blob, random_request, RATES
    rng = np.random.default_rng(52000 + 100 * flavor + seed)
    fs, Bs = RATES[seed % 3]
    a, b = pair(flavor, fma=fma)
    for o in (a, b):
        assert o.set_rate(fs) == 0; o.set_volume(-12 * 256); assert o.load_bulk(WL.full_chain_blob(flavor)) == 0
    B = int(rng.choice(Bs)); per = 5
    for k in range(14):
        for _ in range(int(rng.integers(1, 5))):
            req, wv, payload = random_request(rng, flavor, fs)
            if rng.random() < 0.1: payload = payload[:max(0, len(payload) - 1)]      # short payloads are ignored silently
            if len(payload):
                a.vendor_set(req, wv, payload); b.vendor_set(req, wv, payload)
        if rng.random() < 0.3:
            v = int(rng.choice([0, -256 * 30, -256 * 3, 256 * 2, -256 * 70])); a.set_volume(v); b.set_volume(v)
        big = rng.random()
        if big < 0.08:
            mu = bool(rng.integers(0, 2)); a.set_mute(mu); b.set_mute(mu)
        elif big < 0.16:
            blob = random_blob(rng, flavor, fs)
            assert a.load_bulk(blob) == 0 and b.load_bulk(blob) == 0
        elif big < 0.24:
            ref = Oracle(flavor, x86_casts=True, fma=fma); ref.set_rate(fs); ref.load_bulk(random_blob(rng, flavor, fs)); image = ref.save_slot(3); ref.close()   # the cast switch is per library, not per context
            assert a.load_slot(image) == 0 and b.load_slot(image) == 0
        elif big < 0.28:
            a.factory_defaults(); b.factory_defaults()
        if k == 7:
            fs, Bs = RATES[(seed + 1) % 3]; B = int(rng.choice(Bs))
            assert a.set_rate(fs) == 0 and b.set_rate(fs) == 0
        same_state(a, b, f"step {k}")
        same_audio(a, b, signal(B, per, fs, int(rng.integers(0, 40)), 16), per, B, 16, f"step {k}")


@pytest.mark.parametrize("flavor,fma", FL)
def test_vendor_get_surface(flavor, fma):
    """Every GET of the DSP subset answers with the same bytes (usb_audio.c:2271-2688)."""
    from test_gpu_fuzz import random_blob
    a, b = pair(flavor, fma=fma)
    blob = random_blob(np.random.default_rng(9), flavor, 48000)
    for o in (a, b):
        assert o.set_rate(48000) == 0 and o.load_bulk(blob) == 0
    same_audio(a, b, signal(48, 10, 48000, 19, 16), 10, 48, 16)
    R = W.REQ
    for name, values in (("GET_PREAMP", [0]), ("GET_PREAMP_CH", [0, 1, 2]), ("GET_MASTER_VOLUME", [0]), ("GET_DELAY", range(a.C + 1)),
                         ("GET_BYPASS", [0]), ("GET_CHANNEL_GAIN", range(4)), ("GET_CHANNEL_MUTE", range(4)), ("GET_LOUDNESS", [0]),
                         ("GET_LOUDNESS_REF", [0]), ("GET_LOUDNESS_INTENSITY", [0]), ("GET_CROSSFEED", [0]), ("GET_CROSSFEED_PRESET", [0]),
                         ("GET_CROSSFEED_FREQ", [0]), ("GET_CROSSFEED_FEED", [0]), ("GET_CROSSFEED_ITD", [0]), ("GET_LEVELLER_ENABLE", [0]),
                         ("GET_LEVELLER_AMOUNT", [0]), ("GET_LEVELLER_SPEED", [0]), ("GET_LEVELLER_MAX_GAIN", [0]), ("GET_LEVELLER_LOOKAHEAD", [0]),
                         ("GET_LEVELLER_GATE", [0]), ("GET_STATUS", [0, 1, 2, 9, 15]), ("GET_OUTPUT_ENABLE", range(a.N + 1)),
                         ("GET_OUTPUT_GAIN", range(a.N + 1)), ("GET_OUTPUT_MUTE", range(a.N + 1)), ("GET_OUTPUT_DELAY", range(a.N + 1)),
                         ("GET_CORE1_MODE", [0]), ("GET_CORE1_CONFLICT", range(a.N + 1)), ("GET_PLATFORM", [0]), ("GET_CHANNEL_NAME", range(a.C + 1)),
                         ("GET_MASTER_VOLUME_MODE", [0]), ("GET_SAVED_MASTER_VOLUME", [0])):
        if name not in R: continue
        for v in values:
            ga, gb = a.vendor_get(R[name], v), b.vendor_get(R[name], v)
            if name == "GET_PLATFORM" and ga is not None: ga, gb = ga[:1] + ga[3:], gb[:1] + gb[3:]     # firmware version bytes
            assert ga == gb, f"{name} wValue {v}: {ga} vs {gb}"
    for ch in range(a.C):
        for band in (0, 3, 9, 10):
            for param in range(4):
                wv = (ch << 8) | (band << 4) | param
                assert a.vendor_get(R["GET_EQ_PARAM"], wv) == b.vendor_get(R["GET_EQ_PARAM"], wv)
    for i in range(2):
        for o_ in range(a.N):
            assert a.vendor_get(R["GET_MATRIX_ROUTE"], (i << 8) | o_) == b.vendor_get(R["GET_MATRIX_ROUTE"], (i << 8) | o_)
    assert a.vendor_get(R["CLEAR_CLIPS"], 0) == b.vendor_get(R["CLEAR_CLIPS"], 0)
    assert a.status() == b.status()


@pytest.mark.parametrize("flavor,fma", FL)
def test_preset_slot_images(flavor, fma):
    """collect_live_state / apply_slot_to_live / CRC (flash_storage.c:464-742): slot images byte for byte, and a slot saved by
    either side loads identically into both; a corrupt image is rejected by both and cancels the preset mute."""
    from test_gpu_fuzz import random_blob
    rng = np.random.default_rng(77 + flavor)
    a, b = pair(flavor, fma=fma)
    for trial in range(4):
        blob = random_blob(rng, flavor, 48000)
        mv = struct.pack("<f", float(rng.uniform(-40, 0)))
        for o in (a, b):
            assert o.set_rate(48000) == 0 and o.load_bulk(blob) == 0
            o.vendor_set(W.REQ["SET_MASTER_VOLUME"], 0, mv)
            o.vendor_set(W.REQ["SET_CHANNEL_NAME"], trial, b"Left Woofer")
        ia, ib = a.save_slot(trial), b.save_slot(trial)
        assert ia == ib, f"slot image differs at {[i for i, (x, y) in enumerate(zip(ia, ib)) if x != y][:8]}"
        for o in (a, b): o.factory_defaults()
        same_state(a, b, "factory")
        assert a.load_slot(ia, trial) == 0 and b.load_slot(ia, trial) == 0
        same_state(a, b, f"slot {trial} loaded")
        same_audio(a, b, signal(48, 12, 48000, trial, 16), 12, 48, 16, "after preset load (mute envelope)")
    bad = bytearray(ia); bad[200] ^= 0x10
    assert a.load_slot(bytes(bad), 3) == b.load_slot(bytes(bad), 3) != 0
    same_state(a, b, "after corrupt slot")
    same_audio(a, b, signal(48, 8, 48000, 7, 16), 8, 48, 16, "after corrupt slot")
    # older slot versions take the version-gated paths of apply_slot_to_live (:602, :688, :702-713, :725)
    for ver in (11, 10, 9, 8, 7, 6, 4, 2):
        img = bytearray(ia); struct.pack_into("<H", img, 4, ver)
        struct.pack_into("<I", img, 8, W.slot_crc(bytes(img)))
        assert a.load_slot(bytes(img), 3) == b.load_slot(bytes(img), 3) == 0
        same_state(a, b, f"slot version {ver}")
        same_audio(a, b, signal(48, 14, 48000, ver, 16), 14, 48, 16, f"slot version {ver}")


# ---------------------------------------------------------------------------------------------------------------------
# flash_storage.c: boot from a 48 KB preset area (directory + 10 slots + legacy sector)
# ---------------------------------------------------------------------------------------------------------------------
def _slots(flavor):
    out = {}
    for n, (pre, master) in {0: (-3.0, -10.0), 4: (2.5, -30.0), 9: (-9.0, -5.0)}.items():
        o = Oracle(flavor, x86_casts=True)
        o.load_bulk(WL.full_chain_blob(flavor))
        o.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", pre))
        o.vendor_set(W.REQ["SET_MASTER_VOLUME"], 0, struct.pack("<f", master))
        out[n] = o.save_slot(n)
    return out


def _boot_both(flavor, dump):
    """The firmware boots from the dump (preset_boot_load, core0_init); the oracle and the product open the same dump with
    load_flash_dump.  Parameter state must agree; the mute state differs by design (a boot is not a preset switch)."""
    from dspi_amd.host import Dspi
    fw = Oracle(flavor, ref="fw", flash=dump)
    o = Oracle(flavor, x86_casts=True)
    d = Dspi(flavor, 2, device=None)
    rc_o, rc_d = o.load_flash_dump(dump), d.load_flash_dump(dump)
    assert rc_o == rc_d
    for x in (fw, o, d):
        assert x.set_rate(48000) == 0
    assert fw.collect_bulk() == o.collect_bulk() == d.collect_bulk(), "parameter blob after boot"
    for t in (0, 1, 2, 3, 4, 5, 7):
        assert fw.tap(t) == o.tap(t), f"state tap {t} after boot"
    for k in (0, 1, 2, 4, 5, 6, 7, 8, 9):
        assert fw.scalar(k) == o.scalar(k), f"scalar {k} after boot"
    assert fw.save_slot(2) == o.save_slot(2) == d.save_slot(2)
    for req in ("GET_MASTER_VOLUME", "GET_MASTER_VOLUME_MODE", "GET_SAVED_MASTER_VOLUME"):
        assert fw.vendor_get(W.REQ[req], 0) == o.vendor_get(W.REQ[req], 0) == d.vendor_get(W.REQ[req], 0), req
    # audio once the mute has run out on both sides and its trace has left the longest delay line (80 ms + 128 samples)
    pcm = signal(48, 120, 48000, 4, 16)
    fw.process(pcm, 120, 48); o.process(pcm, 120, 48)
    same_audio(o, fw, signal(48, 20, 48000, 9, 16), 20, 48, 16, "after boot")
    # The BOOT itself (preset_boot_load -> apply_slot_to_live, flash_storage.c:1047-1082: no mute unless the boot writes the flash, no line
    # zeroing): the restatement booted from the dump (orc_new_from_flash) and a host-only product context of devices with a populated flash
    # (DSPI_BOOT_POPULATED_FLASH + dspi_load_flash_dump before any audio) against a second firmware instance booted from it — parameters,
    # state taps, and every word FROM THE FIRST FRAME.
    fw2 = Oracle(flavor, ref="fw", flash=dump)
    ob = Oracle(flavor, x86_casts=True, flash=dump)
    db = Dspi(flavor, 2, device=None, populated_flash=True)
    assert ob.boot_selection == rc_o and db.load_flash_dump(dump) == rc_o
    for x in (fw2, ob, db):
        assert x.set_rate(48000) == 0
        x.set_volume(-9 * 256)
    assert fw2.collect_bulk() == ob.collect_bulk() == db.collect_bulk(), "parameter blob after the boot path"
    assert fw2.save_slot(2) == ob.save_slot(2) == db.save_slot(2)
    for t in (0, 1, 2, 3, 4, 5, 7):
        assert fw2.tap(t) == ob.tap(t), f"state tap {t} after the boot path"
    for k in (0, 1, 2, 4, 5, 6, 7, 8, 9):
        assert fw2.scalar(k) == ob.scalar(k), f"scalar {k} after the boot path"
    same_audio(ob, fw2, signal(48, 40, 48000, 11, 16), 40, 48, 16, "from the first frame after booting from the dump")
    fw2.close(); ob.close(); db.close()
    return rc_o, fw, o


@pytest.mark.parametrize("flavor", (1, 0))
def test_boot_from_flash_dumps(flavor):
    slots = _slots(flavor)
    occ = sum(1 << n for n in slots)
    D, F = W.flash_dump, W.flash_directory
    assert _boot_both(flavor, D(F(default_slot=4, last_active_slot=9, slot_occupied=occ, master_volume_db=-17.0), slots))[0] == 4
    assert _boot_both(flavor, D(F(startup_mode=1, default_slot=4, last_active_slot=9, slot_occupied=occ, master_volume_mode=1), slots))[0] == 9
    assert _boot_both(flavor, D(F(startup_mode=1, default_slot=4, last_active_slot=77, slot_occupied=occ), slots))[0] == 4
    assert _boot_both(flavor, D(F(startup_mode=0, default_slot=200, slot_occupied=occ), slots))[0] == 0
    assert _boot_both(flavor, D(F(default_slot=2, slot_occupied=occ), slots))[0] == 16 + 2             # not occupied -> factory defaults
    bad = dict(slots); b = bytearray(bad[4]); b[100] ^= 0x40; bad[4] = bytes(b)
    assert _boot_both(flavor, D(F(default_slot=4, slot_occupied=occ), bad))[0] == 16 + 4                # CRC mismatch
    swapped = dict(slots); swapped[4] = slots[9]
    assert _boot_both(flavor, D(F(default_slot=4, slot_occupied=occ), swapped))[0] == 16 + 4            # slot_index mismatch
    assert _boot_both(flavor, D(None, slots))[0] == 48                                                  # erased directory, no legacy
    v1 = F(version=1, default_slot=9, slot_occupied=occ, master_volume_mode=1, names={9: "Night"})
    rc, fw, _ = _boot_both(flavor, D(v1, slots))
    assert rc == 9
    # dir_load_cache persisted the migrated v2 directory (flash_storage.c:370-417): version 2, same names, mode carried over
    sector0 = fw.read_flash()[:4096]
    assert struct.unpack_from("<IH", sector0) == (0x44535032, 2)
    broken = bytearray(F(default_slot=4, slot_occupied=occ)); broken[30] ^= 1
    assert _boot_both(flavor, D(bytes(broken), slots))[0] == 48
    future = bytearray(F(default_slot=4, slot_occupied=occ)); future[4] = 3
    assert _boot_both(flavor, D(bytes(future), slots))[0] == 48


@pytest.mark.parametrize("flavor", (1, 0))
def test_legacy_sector_migration(flavor):
    """migrate_legacy (flash_storage.c:997-1045): a pre-preset "DSP1" sector becomes slot 0 and is loaded without its pins."""
    slots = _slots(flavor)
    legacy = W.legacy_sector_from_slot(slots[4], flavor)
    rc, fw, o = _boot_both(flavor, W.flash_dump(None, {}, legacy=legacy))
    assert rc == 32
    dump = fw.read_flash()
    assert struct.unpack_from("<I", dump, 4096)[0] == 0x44535033            # slot 0 now holds the migrated preset
    assert struct.unpack_from("<I", dump, 0)[0] == 0x44535032               # and a directory exists


@pytest.mark.parametrize("flavor", (1, 0))
def test_preset_save_load_delete_through_vendor_requests(flavor):
    """REQ_PRESET_SAVE / LOAD / DELETE / SET_NAME / SET_STARTUP through the reference's handlers and flash code; the saved
    sector must be the image the oracle's collect_live_state gives, and loading it back must reproduce the state."""
    from test_gpu_fuzz import random_blob
    rng = np.random.default_rng(5 + flavor)
    fw = Oracle(flavor, ref="fw"); o = Oracle(flavor, x86_casts=True)
    blob = random_blob(rng, flavor, 48000)
    for x in (fw, o):
        assert x.set_rate(48000) == 0 and x.load_bulk(blob) == 0
    assert fw.vendor_get(W.REQ["PRESET_SAVE"], 6, 1) == b"\x00"
    dump = fw.read_flash()
    sector = dump[(1 + 6) * 4096:(1 + 6) * 4096 + len(o.save_slot(6))]
    assert sector == o.save_slot(6), "sector written by preset_save vs collect_live_state restated"
    assert struct.unpack_from("<H", dump, 16)[0] & (1 << 6)                 # slot_occupied in the directory
    for x in (fw, o): x.factory_defaults()
    same_state(o, fw, "factory")
    assert fw.vendor_get(W.REQ["PRESET_LOAD"], 6, 1) == b"\x00" and o.load_slot(sector, 6) == 0
    same_state(o, fw, "reloaded")
    same_audio(o, fw, signal(48, 30, 48000, 3, 16), 30, 48, 16, "reloaded")
    assert fw.vendor_get(W.REQ["PRESET_GET_ACTIVE"], 0, 1) == b"\x06"


@pytest.mark.parametrize("flavor", [1, 0])
def test_output_type_switches(flavor):
    """REQ_SET_OUTPUT_TYPE / GET (usb_audio.c:2984-3026) and a preset whose slot types differ from the live ones: the handler is
    the reference's and so is the deferred switch (main.c:230-424, run by the firmware build's own main loop); its audio-path
    effect is the mute (main.c:279; a preset load that changes a type re-arms it with PRESET_MUTE_SAMPLES after the flash hold, main.c:957-972).
    The saved sector carries the types (flash_storage.c:522-529)."""
    R = W.REQ
    fw = Oracle(flavor, ref="fw"); o = Oracle(flavor, x86_casts=True)
    P = 4 if flavor else 2
    for x in (fw, o):
        assert x.set_rate(48000) == 0 and x.load_bulk(WL.full_chain_blob(flavor)) == 0
    same_audio(o, fw, signal(48, 20, 48000, 3, 16), 20, 48, 16, "before")
    for wv in (0x0101, 0x0101, 0x0200, (1 << 8) | P, 0x0001 if False else 0x0100):      # switch, no-op, bad type, bad slot, slot 0
        assert fw.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1) == o.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1), hex(wv)
        for slot in range(P + 1):
            assert fw.vendor_get(R["GET_OUTPUT_TYPE"], slot, 1) == o.vendor_get(R["GET_OUTPUT_TYPE"], slot, 1)
        same_state(o, fw, hex(wv))
        same_audio(o, fw, signal(48, 3, 48000, 5, 16), 3, 48, 16, hex(wv))       # 144 frames: inside the 256-sample mute
        same_audio(o, fw, signal(48, 12, 48000, 6, 16), 12, 48, 16, hex(wv))
    assert o.vendor_get(R["GET_OUTPUT_TYPE"], 1, 1) == b"\x01" and o.vendor_get(R["GET_OUTPUT_TYPE"], 0, 1) == b"\x01"
    # save with slots 0 and 1 = I2S, go back to S/PDIF, load: the types come back and the mute is the type switch's
    assert fw.vendor_get(R["PRESET_SAVE"], 2, 1) == b"\x00"
    sector = fw.read_flash()[(1 + 2) * 4096:(1 + 2) * 4096 + len(o.save_slot(2))]
    assert sector == o.save_slot(2)
    for wv in (0x0000, 0x0001):
        assert fw.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1) == o.vendor_get(R["SET_OUTPUT_TYPE"], wv, 1) == b"\x00"
    same_audio(o, fw, signal(48, 12, 48000, 7, 16), 12, 48, 16, "back to S/PDIF")
    assert fw.vendor_get(R["PRESET_LOAD"], 2, 1) == b"\x00" and o.load_slot(sector, 2) == 0
    assert o.vendor_get(R["GET_OUTPUT_TYPE"], 1, 1) == fw.vendor_get(R["GET_OUTPUT_TYPE"], 1, 1) == b"\x01"
    same_state(o, fw, "types from the preset")
    same_audio(o, fw, signal(48, 16, 48000, 8, 16), 16, 48, 16, "types from the preset")
    # the same preset again: no type changes, the flash hold stands
    assert fw.vendor_get(R["PRESET_LOAD"], 2, 1) == b"\x00" and o.load_slot(sector, 2) == 0
    same_state(o, fw, "same types")
    same_audio(o, fw, signal(48, 16, 48000, 9, 16), 16, 48, 16, "same types")


# ---------------------------------------------------------------------------------------------------------------------
# pdm_generator.c: the sigma-delta modulator (pdm_processing_loop, run as a coroutine inside the firmware build)
# ---------------------------------------------------------------------------------------------------------------------
def test_pdm_modulator_is_pdm_processing_loop():
    """oracle/orc_pdm.c against the reference loop (pdm_generator.c:208-397 + dither :62-108): every 32-bit PDM word, from the
    hardware restart (fade-in) through noise, a limiter-driving level, silence, and a second restart (PRNG state carries on)."""
    import ctypes as C
    from orclib import PdmOracle
    fw = orclib.load(1, "fw")
    rng = np.random.default_rng(11)
    sig = np.concatenate([
        (rng.uniform(-0.9, 0.9, 3000) * (1 << 28)).astype(np.int32),                 # noise through the 1024-sample fade-in
        np.full(700, int(1.95 * (1 << 28)), dtype=np.int32),                         # above PDM_CLIP_THRESH
        np.zeros(500, dtype=np.int32),
        (np.sin(np.arange(4000) * 2 * np.pi * 60 / 48000) * 0.7 * (1 << 28)).astype(np.int32),
        np.array([-(1 << 31), (1 << 31) - 1, -1, 1, 0], dtype=np.int32)])
    o = PdmOracle()
    fw.orc_pdm_ref_restart()
    for part in (sig, sig[::-1].copy()):
        want = o.run(part)
        got = np.zeros((part.size, 8), dtype=np.uint32)
        fw.orc_pdm_ref_run(part.ctypes.data_as(C.c_void_p), C.c_uint32(part.size), got.ctypes.data_as(C.c_void_p))
        assert np.array_equal(want, got), f"first differing sample {np.argwhere((want != got).any(axis=1))[:3].ravel().tolist()}"
        o.restart(); fw.orc_pdm_ref_restart_keep_rng()
