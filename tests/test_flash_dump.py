"""raw 48 KB flash dumps (directory sector + 10 preset slots + legacy sector).  The product's host model
(dspi_load_flash_dump, dspi_flash_read_directory) against the oracle's restatement of preset_boot_load's slot selection,
dir_load_cache and migrate_legacy (flash_storage.c:370-417, :997-1105) — both restated (flash_storage.c needs pico-sdk),
so this pins them to each other and to hand-built images; no GPU needed (host-only contexts)."""
import ctypes as C
import struct

import numpy as np
import pytest

from orclib import Oracle
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi, lib


def make_slots(flavor):
    """Three distinguishable presets saved through the oracle's collect_live_state."""
    out = {}
    for n, (pre, master) in {0: (-3.0, -10.0), 4: (2.5, -30.0), 9: (-9.0, -5.0)}.items():
        o = Oracle(flavor)
        o.load_bulk(WL.full_chain_blob(flavor))
        o.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", pre))
        o.vendor_set(W.REQ["SET_MASTER_VOLUME"], 0, struct.pack("<f", master))
        out[n] = o.save_slot(n)
    return out


def both(flavor, dump):
    d = Dspi(flavor, 3, device=None); o = Oracle(flavor)
    for x in (d, o): x.set_rate(48000)
    rd, ro = d.load_flash_dump(dump), o.load_flash_dump(dump)
    assert rd == ro, (rd, ro)
    assert d.collect_bulk() == o.collect_bulk()
    assert d.save_slot(1) == o.save_slot(1)
    return rd, d, o


@pytest.mark.parametrize("flavor", (1, 0))
def test_startup_slot_selection_and_master_volume_mode(flavor):
    slots = make_slots(flavor)
    occ = sum(1 << n for n in slots)
    # specified slot 4, master volume comes from the directory (independent mode)
    rc, d, _ = both(flavor, W.flash_dump(W.flash_directory(default_slot=4, last_active_slot=9, slot_occupied=occ, master_volume_db=-17.0), slots))
    assert rc == 4
    assert struct.unpack("<f", d.vendor_get(W.REQ["GET_MASTER_VOLUME"], 0))[0] == pytest.approx(-17.0)
    # last-active mode -> slot 9; master volume saved with the preset (mode 1)
    rc, d, _ = both(flavor, W.flash_dump(W.flash_directory(startup_mode=1, default_slot=4, last_active_slot=9, slot_occupied=occ, master_volume_mode=1), slots))
    assert rc == 9
    assert struct.unpack("<f", d.vendor_get(W.REQ["GET_MASTER_VOLUME"], 0))[0] == pytest.approx(-5.0)
    # out-of-range last-active falls back to the default slot; out-of-range default to slot 0
    assert both(flavor, W.flash_dump(W.flash_directory(startup_mode=1, default_slot=4, last_active_slot=77, slot_occupied=occ), slots))[0] == 4
    assert both(flavor, W.flash_dump(W.flash_directory(startup_mode=0, default_slot=200, slot_occupied=occ), slots))[0] == 0


@pytest.mark.parametrize("flavor", (1, 0))
def test_empty_and_corrupt_slots_fall_back_to_factory_defaults(flavor):
    slots = make_slots(flavor)
    occ = sum(1 << n for n in slots)
    assert both(flavor, W.flash_dump(W.flash_directory(default_slot=2, slot_occupied=occ), slots))[0] == 16 + 2      # not occupied
    bad = dict(slots); b = bytearray(bad[4]); b[100] ^= 0x40; bad[4] = bytes(b)
    assert both(flavor, W.flash_dump(W.flash_directory(default_slot=4, slot_occupied=occ), bad))[0] == 16 + 4      # CRC mismatch
    swapped = dict(slots); swapped[4] = slots[9]                                                                    # slot_index mismatch (validate_slot)
    assert both(flavor, W.flash_dump(W.flash_directory(default_slot=4, slot_occupied=occ), swapped))[0] == 16 + 4
    rc, d, _ = both(flavor, W.flash_dump(None, slots))                                                               # erased directory, no legacy
    assert rc == 48


@pytest.mark.parametrize("flavor", (1, 0))
def test_directory_v1_migration_and_bad_directories(flavor):
    slots = make_slots(flavor)
    occ = sum(1 << n for n in slots)
    v1 = W.flash_directory(version=1, default_slot=9, slot_occupied=occ, master_volume_mode=1, names={9: "Night"})
    rc, d, _ = both(flavor, W.flash_dump(v1, slots))
    assert rc == 9
    info = read_dir(W.flash_dump(v1, slots))
    assert info.valid == 1 and info.version == 1 and info.master_volume_mode == 1 and info.master_volume_db == pytest.approx(-20.0)
    assert bytes(info.slot_names[9]).rstrip(b"\0") == b"Night" and info.slot_occupied == occ
    broken = bytearray(W.flash_directory(default_slot=4, slot_occupied=occ)); broken[30] ^= 1
    assert both(flavor, W.flash_dump(bytes(broken), slots))[0] == 48 and read_dir(W.flash_dump(bytes(broken), slots)).valid == 0
    future = bytearray(W.flash_directory(default_slot=4, slot_occupied=occ)); future[4] = 3
    assert both(flavor, W.flash_dump(bytes(future), slots))[0] == 48


@pytest.mark.parametrize("flavor", (1, 0))
def test_legacy_sector_migration(flavor):
    slots = make_slots(flavor)
    legacy = W.legacy_sector_from_slot(slots[4], flavor, version=7)
    rc, d, o = both(flavor, W.flash_dump(None, {}, legacy))
    assert rc == 32
    assert struct.unpack("<f", d.vendor_get(W.REQ["GET_PREAMP"], 0))[0] == pytest.approx(2.5)        # the legacy scalar preamp (version < 12)
    bad = bytearray(legacy); bad[40] ^= 2
    assert both(flavor, W.flash_dump(None, {}, bytes(bad)))[0] == 48
    # a directory, even an empty one, wins over the legacy sector
    assert both(flavor, W.flash_dump(W.flash_directory(), {}, legacy))[0] == 16 + 0


class FlashDir(C.Structure):
    _fields_ = [("valid", C.c_int32), ("version", C.c_int32), ("startup_mode", C.c_uint8), ("default_slot", C.c_uint8),
                ("last_active_slot", C.c_uint8), ("include_pins", C.c_uint8), ("slot_occupied", C.c_uint16),
                ("master_volume_mode", C.c_uint8), ("pad_", C.c_uint8), ("master_volume_db", C.c_float), ("slot_names", (C.c_char * 32) * 10)]


def read_dir(dump):
    info = FlashDir()
    assert lib().dspi_flash_read_directory(dump, len(dump), C.byref(info)) == 0
    return info


def test_short_dump_is_rejected():
    d = Dspi(1, 1, device=None)
    assert d.load_flash_dump(b"\xff" * 1000) == -15
