"""ctypes binding of the C-ABI (include/dspi.h) — the Python mirror of the reference's data interface.

There is no fallback: if ``libdspi_mi355x.so`` is missing this module raises, and a context without a
GPU (``device=None``) can only exercise the parameter surface — ``process`` fails with DSPI_E_NODEVICE.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import os

LIB_PATH = Path(os.environ.get("DSPI_LIB") or (Path(__file__).resolve().parent / "csrc" / "libdspi_mi355x.so"))

ALL = -1
MEM_DEVICE = 0x1
OUT_TILED = 0x2
OUT_ENABLED_ONLY = 0x4
OUT_I2S_SLOTS = 0x8
OUT_SPDIF = 0x10
OUT_CLIP_FLAGS = 0x20
E_NODEVICE = -11
E_UNSUPPORTED = -14


class DspiError(RuntimeError):
    def __init__(self, code: int, msg: str = ""):
        super().__init__(f"dspi error {code}: {msg}")
        self.code = code


class _Out(C.Structure):
    _fields_ = [("pairs", C.c_void_p), ("sub", C.c_void_p), ("peaks", C.c_void_p), ("clip_flags", C.c_void_p)]      # clip_flags: ABI 7, read with OUT_CLIP_FLAGS only


_lib = None


def source_fingerprint(pkg_root: Path | None = None) -> str:
    """sha256 (first 16 hex digits) over the library's sources (dspi_amd/csrc/*.hip *.inc *.h *.cpp, include/*.h), in name order.
    tools/prof_summary.py stores it with every counter profile and bench.py compares it with the tree it runs from, so that a
    roofline.traffic figure taken from an older build says so (traffic_stale)."""
    import hashlib
    root = Path(pkg_root) if pkg_root else Path(__file__).resolve().parent
    files = sorted([p for pat in ("*.hip", "*.inc", "*.h", "*.cpp") for p in (root / "csrc").glob(pat)] + list((root.parent / "include").glob("*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode()); h.update(b"\0"); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)")
    L = C.CDLL(str(LIB_PATH))
    vp, i32, u32, u16, u8 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint16, C.c_uint8
    L.dspi_create.argtypes = [C.POINTER(vp), C.c_int, u32, C.c_int]
    L.dspi_destroy.argtypes = [vp]
    L.dspi_last_error.argtypes = [vp]
    L.dspi_last_error.restype = C.c_char_p
    for name in ("dspi_num_channels", "dspi_num_outputs", "dspi_num_pairs"):
        getattr(L, name).argtypes = [vp]
    L.dspi_num_streams.argtypes = [vp]
    L.dspi_num_streams.restype = u32
    L.dspi_tile_streams.argtypes = [vp]
    L.dspi_tile_streams.restype = u32
    L.dspi_factory_defaults.argtypes = [vp, i32]
    L.dspi_load_bulk.argtypes = [vp, i32, vp, C.c_size_t]
    L.dspi_collect_bulk.argtypes = [vp, i32, vp, C.c_size_t]
    L.dspi_load_preset_slot.argtypes = [vp, i32, vp, C.c_size_t, C.c_int]
    L.dspi_load_flash_dump.argtypes = [vp, i32, vp, C.c_size_t]
    L.dspi_flash_read_directory.argtypes = [vp, C.c_size_t, vp]
    L.dspi_save_preset_slot.argtypes = [vp, i32, vp, C.c_size_t, C.c_int]
    L.dspi_vendor_set.argtypes = [vp, i32, u8, u16, vp, u16]
    L.dspi_vendor_get.argtypes = [vp, i32, u8, u16, vp, u16]
    L.dspi_set_host_volume.argtypes = [vp, i32, C.c_int16]
    L.dspi_set_mute.argtypes = [vp, i32, C.c_int]
    L.dspi_set_sample_rate.argtypes = [vp, i32, u32]
    L.dspi_process.argtypes = [vp, vp, C.c_int, u32, u32, C.POINTER(_Out), u32]
    L.dspi_pdm_modulate.argtypes = [vp, vp, u32, vp, u32]
    L.dspi_pdm_restart.argtypes = [vp, C.c_int32]
    L.dspi_spdif_encode.argtypes = [vp, vp, u32, u32, vp, u32]
    L.dspi_i2s_encode.argtypes = [vp, vp, u32, u32, vp, u32]
    L.dspi_sync.argtypes = [vp]
    L.dspi_hip_stream.argtypes = [vp]
    L.dspi_hip_stream.restype = vp
    L.dspi_get_status.argtypes = [vp, i32, vp, C.c_size_t]
    L.dspi_clear_clips.argtypes = [vp, i32]
    L.dspi_debug_image.argtypes = [vp, i32, vp, C.c_size_t]
    L.dspi_debug_launch_plan.argtypes = [vp, vp, C.c_size_t]
    if hasattr(L, "dspi_debug_direct_stats"): L.dspi_debug_direct_stats.argtypes = [vp, vp, C.c_size_t]      # (ABI 8; bench.py's same-box A/B loads the previous round's library through this module)
    _lib = L
    return L


class Dspi:
    """`n_streams` DSPi devices on one GPU (device=None: host-only, parameter surface only)."""

    def __init__(self, flavor: int, n_streams: int, device: int | None = 0, fma: bool = False, populated_flash: bool = False):
        """fma: the float flavour with the firmware build's FMA contraction (DSPI_FLOAT_CONTRACT_FMA, include/dspi.h).
        populated_flash: DSPI_BOOT_POPULATED_FLASH — devices that do not boot for the first time (no first-boot preset mute)."""
        self.L = lib()
        self.h = C.c_void_p()
        fma = bool(fma or getattr(flavor, "fma", False))      # tests pass wire.F32_FMA: the int 1 carrying the contract
        flavor = int(flavor)
        self.fma = fma
        rc = self.L.dspi_create(C.byref(self.h), flavor | (0x100 if fma else 0) | (0x200 if populated_flash else 0), n_streams, -1 if device is None else device)
        if rc != 0:
            raise DspiError(rc, "dspi_create")
        self.flavor, self.n_streams = flavor, n_streams
        self.C = self.L.dspi_num_channels(self.h)
        self.N = self.L.dspi_num_outputs(self.h)
        self.P = self.L.dspi_num_pairs(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.dspi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int, what: str) -> int:
        if rc < 0:
            raise DspiError(rc, f"{what}: {self.L.dspi_last_error(self.h).decode()}")
        return rc

    # ---- parameters ----
    def factory_defaults(self, stream: int = ALL):
        return self._ck(self.L.dspi_factory_defaults(self.h, stream), "factory_defaults")

    def load_bulk(self, blob, stream: int = ALL) -> int:
        raw = blob.tobytes() if hasattr(blob, "tobytes") else bytes(blob)
        return self.L.dspi_load_bulk(self.h, stream, raw, len(raw))      # firmware codes 0,-1..-4 are returned as-is

    def collect_bulk(self, stream: int = 0) -> bytes:
        buf = C.create_string_buffer(2896)
        self._ck(self.L.dspi_collect_bulk(self.h, stream, buf, 2896), "collect_bulk")
        return buf.raw

    def load_slot(self, image: bytes, expect_slot: int = -1, stream: int = ALL) -> int:
        return self.L.dspi_load_preset_slot(self.h, stream, image, len(image), expect_slot)

    def load_flash_dump(self, dump: bytes, stream: int = ALL) -> int:
        """dspi_load_flash_dump: 0..9 slot loaded | 16+n selected slot empty/corrupt | 32 legacy migrated | 48 factory."""
        return self.L.dspi_load_flash_dump(self.h, stream, dump, len(dump))

    def save_slot(self, slot_index: int = 0, stream: int = 0) -> bytes:
        buf = C.create_string_buffer(4096)
        n = self._ck(self.L.dspi_save_preset_slot(self.h, stream, buf, 4096, slot_index), "save_slot")
        return buf.raw[:n]

    def vendor_set(self, req: int, wvalue: int, payload: bytes, stream: int = ALL) -> int:
        return self.L.dspi_vendor_set(self.h, stream, req, wvalue, payload, len(payload))

    def vendor_get(self, req: int, wvalue: int, cap: int = 64, stream: int = 0):
        buf = C.create_string_buffer(max(cap, 1))
        n = self.L.dspi_vendor_get(self.h, stream, req, wvalue, buf, cap)
        return None if n < 0 else buf.raw[:n]

    def set_volume(self, v: int, stream: int = ALL):
        return self._ck(self.L.dspi_set_host_volume(self.h, stream, v), "set_host_volume")

    def set_mute(self, m: bool, stream: int = ALL):
        return self._ck(self.L.dspi_set_mute(self.h, stream, int(m)), "set_mute")

    def set_rate(self, hz: int, stream: int = ALL) -> int:
        return self.L.dspi_set_sample_rate(self.h, stream, hz)

    def status(self, stream: int = 0) -> bytes:
        n = self.C * 2 + 4
        buf = C.create_string_buffer(n)
        self._ck(self.L.dspi_get_status(self.h, stream, buf, n), "get_status")
        return buf.raw

    def clear_clips(self, stream: int = ALL) -> int:
        return self._ck(self.L.dspi_clear_clips(self.h, stream), "clear_clips")

    def debug_image(self, stream: int = 0) -> bytes:
        buf = C.create_string_buffer(8192)
        n = self._ck(self.L.dspi_debug_image(self.h, stream, buf, 8192), "debug_image")
        return buf.raw[:n]

    def spdif_block_pos(self, set: int = -1) -> int:
        """dspi_spdif_block_pos: position in the 192-frame channel-status block of the next DSPI_OUT_SPDIF call's first frame."""
        return self._ck(self.L.dspi_spdif_block_pos(self.h, set), "spdif_block_pos")

    def launch_plan(self) -> dict:
        """dspi_debug_launch_plan: work items per kernel path after the last process call."""
        c = (C.c_uint32 * 7)()
        self._ck(min(self.L.dspi_debug_launch_plan(self.h, c, 7), 0), "debug_launch_plan")
        return dict(zip(("q28_shared", "packed_shared", "one_stream_per_lane_images", "packed_per_lane_values_and_bands", "packed_per_lane_values", "latency_layout",
                         "latency_layout_paired"), list(c)))

    def direct_stats(self) -> dict:
        """dspi_debug_direct_stats: the one-packet-per-call path's polling record (include/dspi.h)."""
        c = (C.c_uint64 * 5)()
        self._ck(min(self.L.dspi_debug_direct_stats(self.h, c, 5), 0), "debug_direct_stats")
        return dict(calls=int(c[0]), blocking_waits=int(c[1]), max_enqueue_us=c[2] / 1e3, max_wait_us=c[3] / 1e3, spin_budget_us=c[4] / 1e3)

    def image_count(self) -> int:
        """dspi_debug_image_count: distinct parameter objects held (equal ones are folded after broadcast calls)."""
        return self._ck(self.L.dspi_debug_image_count(self.h), "debug_image_count")

    def eq_taps(self, x: np.ndarray, channel: int, stream: int = 0):
        """dspi_debug_eq_taps: (taps [11][n], other [10][n]) of one EQ channel on the GPU, see include/dspi.h."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        taps = np.zeros((11, x.size), dtype=np.float32); other = np.zeros((10, x.size), dtype=np.float32)
        self.L.dspi_debug_eq_taps.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        self._ck(self.L.dspi_debug_eq_taps(self.h, stream, channel, x.ctypes.data, x.size, taps.ctypes.data, other.ctypes.data), "debug_eq_taps")
        return taps, other

    # ---- audio ----
    def tile_streams(self) -> int:
        """R of the tiled layouts (dspi.h DSPI_OUT_TILED): 128 float / 64 Q28."""
        return int(self.L.dspi_tile_streams(self.h))

    def process_host(self, pcm: np.ndarray, n_blocks: int, block_len: int, bit_depth: int = 16,
                     want_pairs=True, want_sub=True, want_peaks=True, tiled=False, out=None, enabled_only=False, i2s_slots=False, spdif=False, clip=False):
        """Host-memory convenience path (tests): pcm = int16 [streams][frames][2] or uint8 [streams][frames*6].
        Returns (pairs [S][P][F][2], sub [S][F], peaks [S][blocks][C]); with tiled=True the sample words come back in
        the DSPI_OUT_TILED layout: pairs [tiles][outputs][F][R], sub [tiles][F][R] (see untile()); with spdif=True (DSPI_OUT_SPDIF)
        pairs are the IEC 60958 subframes, uint32 [S][P][F][4].  clip=True (DSPI_OUT_CLIP_FLAGS): every stream's sticky clip flags after the
        call are left in self.last_clip, uint16 [S]."""
        S, F = self.n_streams, n_blocks * block_len
        pcm = np.ascontiguousarray(pcm)
        assert pcm.nbytes == S * F * (6 if bit_depth == 24 else 4), (pcm.shape, S, F)
        if out is not None:      # the arrays of an earlier call, written in place (a host that reuses its buffers: no fresh pages to fault in)
            pairs, sub, peaks = out
            want_pairs, want_sub, want_peaks = pairs is not None, sub is not None, peaks is not None
        elif tiled:
            R = self.tile_streams(); nt = (S + R - 1) // R
            pairs = np.zeros((nt, self.P * 2, F, R), dtype=np.int32) if want_pairs else None
            sub = np.zeros((nt, F, R), dtype=np.int32) if want_sub else None
        else:
            pairs = (np.zeros((S, self.P, F, 4), dtype=np.uint32) if spdif else np.zeros((S, self.P, F, 2), dtype=np.int32)) if want_pairs else None
            sub = np.zeros((S, F), dtype=np.int32) if want_sub else None
        peaks = np.zeros((S, n_blocks, self.C), dtype=np.uint16) if want_peaks else None
        self.last_clip = np.zeros(S, dtype=np.uint16) if clip else None
        out = _Out(pairs.ctypes.data if want_pairs else None, sub.ctypes.data if want_sub else None, peaks.ctypes.data if want_peaks else None, self.last_clip.ctypes.data if clip else None)
        self._ck(self.L.dspi_process(self.h, pcm.ctypes.data, bit_depth, n_blocks, block_len, C.byref(out), (OUT_TILED if tiled else 0) | (OUT_ENABLED_ONLY if enabled_only else 0) | (OUT_I2S_SLOTS if i2s_slots else 0) | (OUT_SPDIF if spdif else 0) | (OUT_CLIP_FLAGS if clip else 0)), "process")
        return pairs, sub, peaks

    def untile(self, pairs_t: np.ndarray, sub_t: np.ndarray):
        """Tiled -> stream-major view of process_host(tiled=True) results (test helper)."""
        S = self.n_streams
        pairs = sub = None
        if pairs_t is not None:
            nt, O, F, R = pairs_t.shape
            pairs = pairs_t.transpose(0, 3, 1, 2).reshape(nt * R, O // 2, 2, F).transpose(0, 1, 3, 2)[:S].copy()
        if sub_t is not None:
            nt, F, R = sub_t.shape
            sub = sub_t.transpose(0, 2, 1).reshape(nt * R, F)[:S].copy()
        return pairs, sub

    def process_device(self, pcm_ptr: int, n_blocks: int, block_len: int, bit_depth: int = 16,
                       pairs_ptr: int = 0, sub_ptr: int = 0, peaks_ptr: int = 0, tiled: bool = False, enabled_only: bool = False, i2s_slots: bool = False, spdif: bool = False,
                       clip_ptr: int = 0):
        """Zero-copy path: raw device pointers (e.g. torch.Tensor.data_ptr()); asynchronous, see sync().  clip_ptr: uint16 [S] (DSPI_OUT_CLIP_FLAGS)."""
        out = _Out(pairs_ptr or None, sub_ptr or None, peaks_ptr or None, clip_ptr or None)
        self._ck(self.L.dspi_process(self.h, pcm_ptr, bit_depth, n_blocks, block_len, C.byref(out), MEM_DEVICE | (OUT_TILED if tiled else 0) | (OUT_ENABLED_ONLY if enabled_only else 0) | (OUT_I2S_SLOTS if i2s_slots else 0) | (OUT_SPDIF if spdif else 0) | (OUT_CLIP_FLAGS if clip_ptr else 0)), "process")

    def pdm_host(self, sub: np.ndarray, tiled: bool = False) -> np.ndarray:
        """PDM sub output (dspi_pdm_modulate) on host arrays: sub int32 [streams][frames] -> uint32 [streams][frames][8];
        tiled: [tiles][frames][R] -> [tiles][frames][8][R]."""
        sub = np.ascontiguousarray(sub, dtype=np.int32)
        words = np.zeros(sub.shape[:2] + (8,) + sub.shape[2:], dtype=np.uint32)
        n_frames = sub.shape[1]
        self._ck(self.L.dspi_pdm_modulate(self.h, sub.ctypes.data, n_frames, words.ctypes.data, OUT_TILED if tiled else 0), "pdm_modulate")
        return words

    def pdm_device(self, sub_ptr: int, n_frames: int, words_ptr: int, tiled: bool = False):
        self._ck(self.L.dspi_pdm_modulate(self.h, sub_ptr, n_frames, words_ptr, MEM_DEVICE | (OUT_TILED if tiled else 0)), "pdm_modulate")

    def spdif_host(self, pairs: np.ndarray, block_pos: int = 0, tiled: bool = False):
        """S/PDIF subframes (dspi_spdif_encode) on host arrays: pairs int32 [streams][P][frames][2] -> uint32
        [streams][P][frames][4]; tiled: [tiles][2P][frames][R] -> [tiles][P][frames][4][R].  Returns (subframes, next_pos)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32)
        if tiled:
            nt, O, F, R = pairs.shape
            out = np.zeros((nt, O // 2, F, 4, R), dtype=np.uint32)
        else:
            S, P, F, _ = pairs.shape
            out = np.zeros((S, P, F, 4), dtype=np.uint32)
        nxt = self.L.dspi_spdif_encode(self.h, pairs.ctypes.data, F, block_pos, out.ctypes.data, OUT_TILED if tiled else 0)
        self._ck(min(nxt, 0), "spdif_encode")
        return out, nxt

    def i2s_host(self, pairs: np.ndarray, pair_mask: int = 0, out: np.ndarray | None = None, tiled: bool = False):
        """I2S slot words (dspi_i2s_encode) on host arrays shaped like the pair words; pair_mask 0 = the slots whose type is I2S.
        Returns (words, mask encoded); pairs outside the mask keep what `out` held (zeros when not given)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32)
        F = pairs.shape[2]
        words = np.zeros(pairs.shape, dtype=np.uint32) if out is None else np.ascontiguousarray(out, dtype=np.uint32)
        m = self.L.dspi_i2s_encode(self.h, pairs.ctypes.data, F, pair_mask, words.ctypes.data, OUT_TILED if tiled else 0)
        self._ck(min(m, 0), "i2s_encode")
        return words, m

    def i2s_device(self, pairs_ptr: int, n_frames: int, pair_mask: int, out_ptr: int, tiled: bool = False) -> int:
        m = self.L.dspi_i2s_encode(self.h, pairs_ptr, n_frames, pair_mask, out_ptr, MEM_DEVICE | (OUT_TILED if tiled else 0))
        self._ck(min(m, 0), "i2s_encode")
        return m

    def spdif_device(self, pairs_ptr: int, n_frames: int, block_pos: int, out_ptr: int, tiled: bool = False) -> int:
        nxt = self.L.dspi_spdif_encode(self.h, pairs_ptr, n_frames, block_pos, out_ptr, MEM_DEVICE | (OUT_TILED if tiled else 0))
        self._ck(min(nxt, 0), "spdif_encode")
        return nxt

    def pdm_restart(self, stream: int = ALL):
        self._ck(self.L.dspi_pdm_restart(self.h, stream), "pdm_restart")

    def sync(self):
        self._ck(self.L.dspi_sync(self.h), "sync")

    def hip_stream(self) -> int:
        return self.L.dspi_hip_stream(self.h) or 0
