"""Binary layouts of the DSPi parameter blobs, as numpy structured dtypes.

These mirror the firmware's own packed little-endian structures so the *same* blobs a DSPi
Console session would send drive this library:

* ``WIRE_BULK`` — ``WireBulkParams`` (2896 bytes, format versions 2..6):
  reference ``firmware/DSPi/bulk_params.h:42-205``.
* ``preset_slot_dtype(flavor)`` — ``PresetSlot`` v12 (2864 B float flavour / 1840 B Q28):
  reference ``firmware/DSPi/flash_storage.c:136-189``; CRC-32 (poly 0xEDB88320) over
  everything after the 12-byte header, ``flash_storage.c:282-291, 548-551``.
* vendor request codes ``REQ_*``: ``firmware/DSPi/config.h:111-251``.

Pure data-format code: no DSP arithmetic lives here.
"""
from __future__ import annotations

import zlib
import numpy as np

FLAVOR_Q28 = 0   # RP2040, PLATFORM_RP2040 (config.h:269)
FLAVOR_F32 = 1   # RP2350, PLATFORM_RP2350 (config.h:270)

WIRE_MAX_CHANNELS = 11
WIRE_MAX_OUTPUTS = 9
MAX_BANDS = 12
NAME_LEN = 32
WIRE_FORMAT_VERSION = 6

FILTER_FLAT, FILTER_PEAKING, FILTER_LOWSHELF, FILTER_HIGHSHELF, FILTER_LOWPASS, FILTER_HIGHPASS = range(6)


def dims(flavor: int):
    """(num_channels, num_outputs, num_spdif_pairs, num_pin_outputs, max_delay_samples)"""
    return (11, 9, 4, 5, 4096) if flavor == FLAVOR_F32 else (7, 5, 2, 3, 2048)


_hdr = np.dtype([("format_version", "u1"), ("platform_id", "u1"), ("num_channels", "u1"),
                 ("num_output_channels", "u1"), ("num_input_channels", "u1"), ("max_bands", "u1"),
                 ("payload_length", "<u2"), ("fw_version_major", "<u2"), ("fw_version_minor", "<u2"),
                 ("reserved", "<u4")])
_glob = np.dtype([("preamp_gain_db", "<f4"), ("bypass", "u1"), ("loudness_enabled", "u1"), ("reserved", "u1", 2),
                  ("loudness_ref_spl", "<f4"), ("loudness_intensity_pct", "<f4")])
_xf = np.dtype([("enabled", "u1"), ("preset", "u1"), ("itd_enabled", "u1"), ("reserved", "u1"),
                ("custom_fc", "<f4"), ("custom_feed_db", "<f4"), ("reserved2", "<u4")])
_legacy = np.dtype([("gain_db", "<f4", 3), ("mute", "u1", 3), ("reserved", "u1")])
_xp = np.dtype([("enabled", "u1"), ("phase_invert", "u1"), ("reserved", "u1", 2), ("gain_db", "<f4")])
_out = np.dtype([("enabled", "u1"), ("mute", "u1"), ("reserved", "u1", 2), ("gain_db", "<f4"), ("delay_ms", "<f4")])
_pins = np.dtype([("num_pin_outputs", "u1"), ("pins", "u1", 5), ("reserved", "u1", 2)])
_band = np.dtype([("type", "u1"), ("reserved", "u1", 3), ("freq", "<f4"), ("q", "<f4"), ("gain_db", "<f4")])
_i2s = np.dtype([("output_types", "u1", 4), ("bck_pin", "u1"), ("mck_pin", "u1"), ("mck_enabled", "u1"),
                 ("mck_multiplier", "u1"), ("reserved", "u1", 8)])
_lev = np.dtype([("enabled", "u1"), ("speed", "u1"), ("lookahead", "u1"), ("reserved", "u1"),
                 ("amount", "<f4"), ("max_gain_db", "<f4"), ("gate_threshold_db", "<f4")])
_pre = np.dtype([("preamp_db", "<f4", 2), ("reserved", "u1", 8)])
_mv = np.dtype([("master_volume_db", "<f4"), ("reserved", "u1", 12)])

WIRE_BULK = np.dtype([
    ("header", _hdr), ("global_", _glob), ("crossfeed", _xf), ("legacy", _legacy),
    ("delays", "<f4", WIRE_MAX_CHANNELS),
    ("crosspoints", _xp, (2, WIRE_MAX_OUTPUTS)), ("outputs", _out, WIRE_MAX_OUTPUTS), ("pins", _pins),
    ("eq", _band, (WIRE_MAX_CHANNELS, MAX_BANDS)), ("channel_names", "S32", WIRE_MAX_CHANNELS),
    ("i2s_config", _i2s), ("leveller", _lev), ("preamp", _pre), ("master_volume", _mv)])
assert WIRE_BULK.itemsize == 2896
WIRE_BULK_SIZE = 2896

_eqp = np.dtype([("channel", "u1"), ("band", "u1"), ("type", "u1"), ("reserved", "u1"),
                 ("freq", "<f4"), ("Q", "<f4"), ("gain_db", "<f4")])
EQ_PARAM_PACKET = _eqp
MATRIX_ROUTE_PACKET = np.dtype([("input", "u1"), ("output", "u1"), ("enabled", "u1"), ("phase_invert", "u1"), ("gain_db", "<f4")])

SLOT_MAGIC = 0x44535033
SLOT_DATA_VERSION = 12


def preset_slot_dtype(flavor: int) -> np.dtype:
    C, N, _, P, _ = dims(flavor)
    d = np.dtype([
        ("magic", "<u4"), ("version", "<u2"), ("slot_index", "<u2"), ("crc32", "<u4"),
        ("filter_recipes", _eqp, (C, MAX_BANDS)),
        ("preamp_db", "<f4"), ("bypass", "u1"), ("padding", "u1", 3),
        ("delays_ms", "<f4", C),
        ("channel_gain_db", "<f4", 3), ("channel_mute", "u1", 3), ("padding2", "u1"),
        ("loudness_enabled", "u1"), ("padding3", "u1", 3), ("loudness_ref_spl", "<f4"), ("loudness_intensity_pct", "<f4"),
        ("crossfeed_enabled", "u1"), ("crossfeed_preset", "u1"), ("crossfeed_itd_enabled", "u1"), ("padding4", "u1"),
        ("crossfeed_custom_fc", "<f4"), ("crossfeed_custom_feed_db", "<f4"),
        ("matrix_crosspoints", _xp, (2, N)), ("matrix_outputs", _out, N),
        ("output_pins", "u1", P), ("pin_padding", "u1", 8 - P),
        ("channel_names", "S32", C),
        ("output_types", "u1", 4), ("i2s_bck_pin", "u1"), ("i2s_mck_pin", "u1"), ("i2s_mck_enabled", "u1"), ("i2s_mck_multiplier", "u1"),
        ("leveller_enabled", "u1"), ("leveller_speed", "u1"), ("leveller_lookahead", "u1"), ("leveller_padding", "u1"),
        ("leveller_amount", "<f4"), ("leveller_max_gain_db", "<f4"), ("leveller_gate_threshold_db", "<f4"),
        ("preamp_db_per_ch", "<f4", 2), ("master_volume_db", "<f4")])
    assert d.itemsize == (2864 if flavor == FLAVOR_F32 else 1840)
    return d


def slot_crc(image: bytes) -> int:
    """CRC-32 over the data section (bytes after the 12-byte header); equals zlib.crc32."""
    return zlib.crc32(image[12:]) & 0xFFFFFFFF


def seal_slot(slot: np.ndarray) -> bytes:
    """Fill in the CRC and return the slot image bytes."""
    raw = bytearray(slot.tobytes())
    crc = slot_crc(bytes(raw))
    raw[8:12] = int(crc).to_bytes(4, "little")
    return bytes(raw)


# --- vendor request codes (config.h:111-251), DSP subset --------------------------------------
REQ = dict(
    SET_EQ_PARAM=0x42, GET_EQ_PARAM=0x43, SET_PREAMP=0x44, GET_PREAMP=0x45, SET_BYPASS=0x46, GET_BYPASS=0x47,
    SET_DELAY=0x48, GET_DELAY=0x49, GET_STATUS=0x50, FACTORY_RESET=0x53,
    SET_CHANNEL_GAIN=0x54, GET_CHANNEL_GAIN=0x55, SET_CHANNEL_MUTE=0x56, GET_CHANNEL_MUTE=0x57,
    SET_LOUDNESS=0x58, GET_LOUDNESS=0x59, SET_LOUDNESS_REF=0x5A, GET_LOUDNESS_REF=0x5B,
    SET_LOUDNESS_INTENSITY=0x5C, GET_LOUDNESS_INTENSITY=0x5D,
    SET_CROSSFEED=0x5E, GET_CROSSFEED=0x5F, SET_CROSSFEED_PRESET=0x60, GET_CROSSFEED_PRESET=0x61,
    SET_CROSSFEED_FREQ=0x62, GET_CROSSFEED_FREQ=0x63, SET_CROSSFEED_FEED=0x64, GET_CROSSFEED_FEED=0x65,
    SET_CROSSFEED_ITD=0x66, GET_CROSSFEED_ITD=0x67,
    SET_MATRIX_ROUTE=0x70, GET_MATRIX_ROUTE=0x71, SET_OUTPUT_ENABLE=0x72, GET_OUTPUT_ENABLE=0x73,
    SET_OUTPUT_GAIN=0x74, GET_OUTPUT_GAIN=0x75, SET_OUTPUT_MUTE=0x76, GET_OUTPUT_MUTE=0x77,
    SET_OUTPUT_DELAY=0x78, GET_OUTPUT_DELAY=0x79, GET_CORE1_MODE=0x7A, GET_CORE1_CONFLICT=0x7B,
    GET_PLATFORM=0x7F, CLEAR_CLIPS=0x83, SET_CHANNEL_NAME=0x9B, GET_CHANNEL_NAME=0x9C,
    GET_ALL_PARAMS=0xA0, SET_ALL_PARAMS=0xA1,
    SET_LEVELLER_ENABLE=0xB4, GET_LEVELLER_ENABLE=0xB5, SET_LEVELLER_AMOUNT=0xB6, GET_LEVELLER_AMOUNT=0xB7,
    SET_LEVELLER_SPEED=0xB8, GET_LEVELLER_SPEED=0xB9, SET_LEVELLER_MAX_GAIN=0xBA, GET_LEVELLER_MAX_GAIN=0xBB,
    SET_LEVELLER_LOOKAHEAD=0xBC, GET_LEVELLER_LOOKAHEAD=0xBD, SET_LEVELLER_GATE=0xBE, GET_LEVELLER_GATE=0xBF,
    SET_OUTPUT_TYPE=0xC0, GET_OUTPUT_TYPE=0xC1,
    SET_PREAMP_CH=0xD0, GET_PREAMP_CH=0xD1, SET_MASTER_VOLUME=0xD2, GET_MASTER_VOLUME=0xD3,
    PRESET_SAVE=0x90, PRESET_LOAD=0x91, PRESET_DELETE=0x92, PRESET_GET_NAME=0x93, PRESET_SET_NAME=0x94, PRESET_GET_DIR=0x95,
    PRESET_SET_STARTUP=0x96, PRESET_GET_STARTUP=0x97, PRESET_SET_INCLUDE_PINS=0x98, PRESET_GET_INCLUDE_PINS=0x99, PRESET_GET_ACTIVE=0x9A,
    SET_MASTER_VOLUME_MODE=0xD4, GET_MASTER_VOLUME_MODE=0xD5, SAVE_MASTER_VOLUME=0xD6, GET_SAVED_MASTER_VOLUME=0xD7,
)


class FlavorFma(int):
    """The float flavour with the firmware build's FMA contraction (include/dspi.h DSPI_FLOAT_CONTRACT_FMA): behaves as the
    int 1 everywhere (blob builders, channel counts), and host.Dspi / tests' Oracle read the contract off `.fma`."""
    fma = True

    def __repr__(self):
        return "1fma"
    __str__ = __repr__


F32_FMA = FlavorFma(1)


def new_bulk(flavor: int) -> np.ndarray:
    """A zeroed V6 blob with a valid header for `flavor` (all bands FLAT-typed at 1 kHz/0.707 like
    dsp_init_default_filters, outputs disabled, unity gains, leveller/crossfeed/loudness off)."""
    C, N, S, P, _ = dims(flavor)
    b = np.zeros((), dtype=WIRE_BULK)
    h = b["header"]
    h["format_version"] = WIRE_FORMAT_VERSION
    h["platform_id"] = flavor
    h["num_channels"] = C
    h["num_output_channels"] = N
    h["num_input_channels"] = 2
    h["max_bands"] = MAX_BANDS
    h["payload_length"] = WIRE_BULK_SIZE
    h["fw_version_major"] = 1
    h["fw_version_minor"] = 1
    b["global_"]["loudness_ref_spl"] = 83.0
    b["global_"]["loudness_intensity_pct"] = 100.0
    b["crossfeed"]["itd_enabled"] = 1
    b["crossfeed"]["custom_fc"] = 700.0
    b["crossfeed"]["custom_feed_db"] = 4.5
    b["eq"]["freq"][:C] = 1000.0
    b["eq"]["q"][:C] = 0.707
    b["pins"]["num_pin_outputs"] = P
    b["pins"]["pins"][:P] = [6, 7, 8, 9, 10][:P] if flavor == FLAVOR_F32 else [6, 7, 10]
    lv = b["leveller"]
    lv["speed"] = 0
    lv["lookahead"] = 1
    lv["amount"] = 50.0
    lv["max_gain_db"] = 15.0
    lv["gate_threshold_db"] = -96.0
    b["i2s_config"]["bck_pin"] = 14
    b["i2s_config"]["mck_pin"] = 13
    return b


def set_band(b: np.ndarray, ch: int, band: int, ftype: int, freq: float, q: float, gain_db: float) -> None:
    e = b["eq"][ch, band]
    e["type"], e["freq"], e["q"], e["gain_db"] = ftype, freq, q, gain_db


# --- flash dump (flash_storage.c:4-26, :95-131): 12 sectors of 4 KB — directory, 10 preset slots, legacy sector -------
FLASH_SECTOR = 4096
FLASH_DUMP_BYTES = 12 * FLASH_SECTOR
DIR_MAGIC, LEGACY_MAGIC = 0x44535032, 0x44535031


def flash_directory(version=2, startup_mode=0, default_slot=0, last_active_slot=0, include_pins=1, slot_occupied=0,
                    master_volume_mode=0, master_volume_db=-20.0, names=None) -> bytes:
    """One directory sector image (PresetDirectory v2, or PresetDirectory_v1 where master_volume_mode is the old
    include_master_volume flag), CRC filled in."""
    import struct
    names = names or {}
    body = struct.pack("<BBBBHBB", startup_mode, default_slot, last_active_slot, include_pins, slot_occupied, master_volume_mode, 0)
    if version == 2:
        body += struct.pack("<f", master_volume_db)
    for n in range(10):
        body += names.get(n, "").encode()[:31].ljust(32, b"\0")
    hdr = struct.pack("<IHHI", DIR_MAGIC, version, 0, zlib.crc32(body) & 0xFFFFFFFF)
    return (hdr + body).ljust(FLASH_SECTOR, b"\xff")


def flash_dump(directory: bytes = None, slots=None, legacy: bytes = None) -> bytes:
    """Assemble a 48 KB dump: erased flash (0xFF) except the given directory sector, slot images {n: bytes}, legacy sector."""
    img = bytearray(b"\xff" * FLASH_DUMP_BYTES)
    if directory: img[0:len(directory)] = directory
    for n, s in (slots or {}).items():
        img[(1 + n) * FLASH_SECTOR:(1 + n) * FLASH_SECTOR + len(s)] = s
    if legacy: img[11 * FLASH_SECTOR:11 * FLASH_SECTOR + len(legacy)] = legacy
    return bytes(img)


def legacy_sector_from_slot(slot_image: bytes, flavor: int, version: int = 7) -> bytes:
    """A LegacyFlashStorage image (flash_storage.c:203-232) carrying the data section of a slot image up to the pins."""
    import struct
    n_ch, n_out = (11, 9) if flavor else (7, 5)
    legacy_bytes = 12 + n_ch * 12 * 16 + 4 + 4 + n_ch * 4 + 12 + 4 + 4 + 8 + 4 + 8 + 2 * n_out * 8 + n_out * 12 + 8
    data = slot_image[12:legacy_bytes]
    return struct.pack("<IHHI", LEGACY_MAGIC, version, 0, zlib.crc32(data) & 0xFFFFFFFF) + data
