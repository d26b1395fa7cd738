"""ctypes binding of the CPU oracle (oracle/orc_api.h).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORC_DIR = ROOT / "oracle"

_libs: dict = {}


def _build():
    need = [ORC_DIR / "liborc_f32.so", ORC_DIR / "liborc_q28.so", ORC_DIR / "liborc_pdm.so", ORC_DIR / "liborc_spdif.so"]
    srcs = [ORC_DIR / n for n in ("orc_chain.c", "orc_leaf.c", "orc_types.h", "orc_leaf.h", "orc_common.h", "orc_api.h", "orc_pdm.c", "orc_spdif.c")]
    srcs.append(ROOT / "include" / "dspi_detmath.h")
    newest = max(s.stat().st_mtime for s in srcs)
    if all(p.exists() and p.stat().st_mtime >= newest for p in need):
        return
    subprocess.run(["make", "-C", str(ORC_DIR), "-s"], check=True)


def ref_available(flavor: int, kind="ref", fma: bool = False) -> bool:
    """kind "ref": reference leaf sources under our orchestrator; "fw": the firmware build (usb_audio.c, flash_storage.c,
    pdm_generator.c compiled in place, oracle/ref_fw.c).  fma: built with the firmware's float contract (float flavour only)."""
    return _ref_path(flavor, kind, fma).exists()


def _ref_path(flavor: int, kind, fma: bool) -> Path:
    name = ("f32" if flavor else "q28") + ("_fma" if fma else "")
    return ORC_DIR / "_ref" / (f"libref_fw_{name}.so" if kind == "fw" else f"libref_{name}.so")


_fw_tmp = None
_fw_count = 0


def _private_copy(path: Path) -> str:
    """The firmware build keeps one device in file-scope globals: every Oracle gets its own copy of the library."""
    global _fw_tmp, _fw_count
    import shutil, tempfile
    if _fw_tmp is None:
        _fw_tmp = tempfile.TemporaryDirectory(prefix="dspi_fw_")
    _fw_count += 1
    dst = Path(_fw_tmp.name) / f"{path.stem}_{_fw_count}.so"
    shutil.copyfile(path, dst)
    return str(dst)


def load(flavor: int, ref=False, fma: bool = False) -> C.CDLL:
    key = (flavor, ref, fma)
    if key in _libs and ref != "fw":
        return _libs[key]
    name = "f32" if flavor else "q28"
    if ref:
        path = _ref_path(flavor, ref, fma)
        if not path.exists() and Path("/root/reference/firmware/DSPi").is_dir():
            subprocess.run(["make", "-C", str(ORC_DIR), "-s", "ref"], check=True)
        if ref == "fw":
            path = Path(_private_copy(path))
    else:
        _build()
        path = ORC_DIR / f"liborc_{name}.so"
    # RTLD_LOCAL + distinct files: the builds export identical symbol names
    lib = C.CDLL(str(path), mode=os.RTLD_LOCAL)
    lib.orc_new.restype = C.c_void_p
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_set_sample_rate.argtypes = [C.c_void_p, C.c_uint32]
    lib.orc_set_host_volume.argtypes = [C.c_void_p, C.c_int16]
    lib.orc_set_mute.argtypes = [C.c_void_p, C.c_int]
    lib.orc_factory_defaults.argtypes = [C.c_void_p]
    lib.orc_load_bulk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.orc_collect_bulk.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_load_preset_slot.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    lib.orc_save_preset_slot.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    if hasattr(lib, "orc_load_flash_dump"): lib.orc_load_flash_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    if hasattr(lib, "orc_new_from_flash"):
        lib.orc_new_from_flash.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
        lib.orc_new_from_flash.restype = C.c_void_p
    lib.orc_vendor_set.argtypes = [C.c_void_p, C.c_uint8, C.c_uint16, C.c_void_p, C.c_uint16]
    lib.orc_vendor_get.argtypes = [C.c_void_p, C.c_uint8, C.c_uint16, C.c_void_p, C.c_uint16]
    lib.orc_get_status.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_tap.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.orc_tap.restype = C.c_void_p
    lib.orc_scalar.argtypes = [C.c_void_p, C.c_int]
    lib.orc_scalar_f.argtypes = [C.c_void_p, C.c_int]
    lib.orc_scalar_f.restype = C.c_float
    assert lib.orc_flavor() == flavor and lib.orc_is_ref_build() == (2 if ref == "fw" else int(bool(ref)))
    if hasattr(lib, "orc_set_fma_mode"): lib.orc_set_fma_mode.argtypes = [C.c_int]
    if ref == "fw":
        lib.orc_boot_from_flash.argtypes = [C.c_void_p, C.c_uint32]
        lib.orc_read_flash.argtypes = [C.c_void_p]
        lib.orc_pdm_ref_run.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.orc_pdm_ref_restart.restype = None; lib.orc_pdm_ref_restart_keep_rng.restype = None
    else:
        _libs[key] = lib
    return lib


class Oracle:
    """One DSPi device (= one stereo stream)."""

    def __init__(self, flavor: int, ref=False, detmath: bool = True, x86_casts: bool = False, fma: bool = False, flash: bytes = None):
        """ref: False = standalone restatement, True = reference leaf sources (_ref), "fw" = firmware build (ref_fw.c).
        fma: the firmware's float contract (contraction on).  The standalone build switches at run time
        (orc_set_fma_mode, explicit fmaf in the pattern GCC produces); the reference builds are separate libraries.
        flash: boot from this 48 KB preset area instead of an erased flash (the firmware build runs the reference's own core0_init /
        preset_boot_load over it; the restatement its orc_new_from_flash: `boot_selection` holds preset_boot_load's choice)."""
        fma = bool(fma or getattr(flavor, "fma", False))      # wire.F32_FMA: the int 1 carrying the contract
        flavor = int(flavor)
        self.lib = load(flavor, ref, fma if ref else False)
        self.flavor = flavor
        self.fma = fma
        self.detmath, self.x86_casts, self.ref = detmath, x86_casts, ref
        self._sync()
        self.C = self.lib.orc_num_channels()
        self.N = self.lib.orc_num_outputs()
        self.P = self.lib.orc_num_pairs()
        self.boot_selection = None
        if flash is not None and ref != "fw":
            sel = C.c_int(-1)
            self.h = self.lib.orc_new_from_flash(flash, len(flash), C.byref(sel))
            assert self.h, "orc_new_from_flash"
            self.boot_selection = sel.value
            return
        if flash is not None:
            assert self.lib.orc_boot_from_flash(flash, len(flash)) == 0
        self.h = self.lib.orc_new()

    def _sync(self):
        """The math / cast / contract switches are globals of the library, and Oracles of one build share the library:
        every call re-asserts this Oracle's settings first."""
        self.lib.orc_set_math_mode(1 if self.detmath else 0)
        self.lib.orc_set_x86_cast_semantics(1 if self.x86_casts else 0)
        if hasattr(self.lib, "orc_set_fma_mode") and self.ref != "fw":
            self.lib.orc_set_fma_mode(1 if self.fma else 0)

    def close(self):
        if self.h:
            self.lib.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_math(self, detmath: bool):
        self.detmath = detmath

    def set_rate(self, hz: int) -> int:
        self._sync()
        return self.lib.orc_set_sample_rate(self.h, hz)

    def set_volume(self, v: int):
        self._sync()
        self.lib.orc_set_host_volume(self.h, v)

    def set_mute(self, m: bool):
        self._sync()
        self.lib.orc_set_mute(self.h, int(m))

    def factory_defaults(self):
        self._sync()
        self.lib.orc_factory_defaults(self.h)

    def load_bulk(self, blob) -> int:
        self._sync()
        raw = blob.tobytes() if hasattr(blob, "tobytes") else bytes(blob)
        return self.lib.orc_load_bulk(self.h, raw, len(raw))

    def collect_bulk(self) -> bytes:
        self._sync()
        buf = C.create_string_buffer(2896)
        self.lib.orc_collect_bulk(self.h, buf)
        return buf.raw

    def load_slot(self, image: bytes, expect_slot: int = -1) -> int:
        self._sync()
        return self.lib.orc_load_preset_slot(self.h, image, len(image), expect_slot)

    def read_flash(self) -> bytes:
        buf = C.create_string_buffer(12 * 4096)
        self.lib.orc_read_flash(buf)
        return buf.raw

    def load_flash_dump(self, dump: bytes) -> int:
        self._sync()
        return self.lib.orc_load_flash_dump(self.h, dump, len(dump))

    def save_slot(self, slot_index: int = 0) -> bytes:
        self._sync()
        n = self.lib.orc_preset_slot_size()
        buf = C.create_string_buffer(n)
        self.lib.orc_save_preset_slot(self.h, buf, slot_index)
        return buf.raw

    def vendor_set(self, req: int, wvalue: int, payload: bytes) -> int:
        self._sync()
        return self.lib.orc_vendor_set(self.h, req, wvalue, payload, len(payload))

    def vendor_get(self, req: int, wvalue: int, cap: int = 64):
        self._sync()
        buf = C.create_string_buffer(max(cap, 1))
        n = self.lib.orc_vendor_get(self.h, req, wvalue, buf, cap)
        return None if n < 0 else buf.raw[:n]

    def status(self) -> bytes:
        self._sync()
        buf = C.create_string_buffer(self.C * 2 + 4)
        self.lib.orc_get_status(self.h, buf)
        return buf.raw

    def eq_taps(self, x: np.ndarray, channel: int) -> np.ndarray:
        """orc_debug_eq_taps: float [11][n], x through the ten bands of `channel` from zero state (float flavour, not the fw build)."""
        self._sync()
        x = np.ascontiguousarray(x, dtype=np.float32)
        taps = np.zeros((11, x.size), dtype=np.float32)
        self.lib.orc_debug_eq_taps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        assert self.lib.orc_debug_eq_taps(self.h, channel, x.ctypes.data, x.size, taps.ctypes.data) == 0
        return taps

    def tap(self, what: int) -> bytes:
        n = C.c_int(0)
        p = self.lib.orc_tap(self.h, what, C.byref(n))
        return C.string_at(p, n.value)

    def scalar(self, what: int) -> int:
        return self.lib.orc_scalar(self.h, what)

    def scalar_f(self, what: int) -> float:
        return self.lib.orc_scalar_f(self.h, what)

    def process(self, pcm: np.ndarray, n_blocks: int, block_len: int, bit_depth: int = 16, want_peaks: bool = True):
        """pcm: int16 [frames][2] or uint8 [frames*6].  Returns (pairs [P][frames][2], sub [frames], peaks [blocks][C], clip)."""
        self._sync()
        frames = n_blocks * block_len
        pcm = np.ascontiguousarray(pcm)
        assert pcm.nbytes == frames * (6 if bit_depth == 24 else 4), (pcm.nbytes, frames)
        pairs = np.zeros((self.P, frames, 2), dtype=np.int32)
        sub = np.zeros(frames, dtype=np.int32)
        peaks = np.zeros((n_blocks, self.C), dtype=np.uint16)
        clip = np.zeros(1, dtype=np.uint16)
        self.lib.orc_process(self.h, pcm.ctypes.data, bit_depth, n_blocks, block_len, pairs.ctypes.data, sub.ctypes.data,
                             peaks.ctypes.data if want_peaks else None, clip.ctypes.data)
        return pairs, sub, peaks, int(clip[0])


class PdmOracle:
    """One PDM sigma-delta modulator (oracle/orc_pdm.c): the consumer of the chain's Q28 sub output."""

    def __init__(self):
        _build()
        self.L = C.CDLL(str(ORC_DIR / "liborc_pdm.so"), mode=os.RTLD_LOCAL)
        self.L.orc_pdm_state_words.restype = C.c_int
        self.st = (C.c_uint32 * self.L.orc_pdm_state_words())()
        self.L.orc_pdm_init(self.st)

    def restart(self):
        self.L.orc_pdm_restart(self.st)

    def run(self, sub: np.ndarray) -> np.ndarray:
        sub = np.ascontiguousarray(sub, dtype=np.int32)
        words = np.zeros((sub.size, 8), dtype=np.uint32)
        self.L.orc_pdm_run(self.st, sub.ctypes.data_as(C.c_void_p), C.c_uint32(sub.size), words.ctypes.data_as(C.c_void_p))
        return words


def spdif_ref_available() -> bool:
    return (ORC_DIR / "_ref" / "libref_spdif.so").exists()


def spdif_encode(pair_words: np.ndarray, block_pos: int, fs: int, ref: bool = False):
    """oracle/orc_spdif.c: int32 [frames][2] -> uint32 [frames][4] ({l,h} left, {l,h} right), next block position."""
    _build()
    L = C.CDLL(str(ORC_DIR / ("_ref/libref_spdif.so" if ref else "liborc_spdif.so")), mode=os.RTLD_LOCAL)
    L.orc_spdif_encode.restype = C.c_uint32
    x = np.ascontiguousarray(pair_words, dtype=np.int32)
    out = np.zeros((x.shape[0], 4), dtype=np.uint32)
    nxt = L.orc_spdif_encode(x.ctypes.data_as(C.c_void_p), C.c_uint32(x.shape[0]), C.c_uint32(block_pos), C.c_uint32(fs), out.ctypes.data_as(C.c_void_p))
    return out, int(nxt)


def i2s_ref_available() -> bool:
    return (ORC_DIR / "_ref" / "libref_i2s.so").exists()


def i2s_frames(pair_words: np.ndarray) -> np.ndarray:
    """oracle/orc_spdif.c:orc_i2s_frames — int32 [frames][2] -> uint32 [frames][2], the words an I2S slot shifts out."""
    _build()
    L = C.CDLL(str(ORC_DIR / "liborc_spdif.so"), mode=os.RTLD_LOCAL)
    x = np.ascontiguousarray(pair_words, dtype=np.int32)
    out = np.zeros((x.shape[0], 2), dtype=np.uint32)
    L.orc_i2s_frames(x.ctypes.data_as(C.c_void_p), C.c_uint32(x.shape[0]), out.ctypes.data_as(C.c_void_p))
    return out


def i2s_ref_give(pair_words: np.ndarray, packet: int, consumer_len: int) -> np.ndarray:
    """the reference's i2s_wrap_producer_give (oracle/ref_i2s.c) fed in packets: completed consumer buffers back to back"""
    L = C.CDLL(str(ORC_DIR / "_ref" / "libref_i2s.so"), mode=os.RTLD_LOCAL)
    L.orc_i2s_ref_give.restype = C.c_uint32
    x = np.ascontiguousarray(pair_words, dtype=np.int32)
    out = np.zeros((x.shape[0], 2), dtype=np.uint32)
    n = L.orc_i2s_ref_give(x.ctypes.data_as(C.c_void_p), C.c_uint32(x.shape[0]), C.c_uint32(packet), C.c_uint32(consumer_len), out.ctypes.data_as(C.c_void_p))
    return out[:n]
