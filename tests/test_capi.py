"""The C-ABI library loads and exports every entry point include/dspi.h declares (no GPU needed)."""
import ctypes
import os
import re

from dspi_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(product_lib):
    hdr = open(os.path.join(ROOT, "include", "dspi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(dspi_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 24
    for n in names:
        assert hasattr(product_lib, n), f"{n} declared in dspi.h but not exported"
    product_lib.dspi_abi_version.restype = ctypes.c_int
    assert product_lib.dspi_abi_version() == 8


def test_argument_validation(product_lib):
    h = ctypes.c_void_p()
    assert product_lib.dspi_create(ctypes.byref(h), 7, 4, -1) == -10       # bad flavour
    assert product_lib.dspi_create(ctypes.byref(h), 1, 0, -1) == -10       # no streams
    d = host.Dspi(0, 5, device=None)
    assert (d.C, d.N, d.P) == (7, 5, 2)
    assert d.set_rate(32000) == -10 and d.set_rate(96000) == 0
    assert d.L.dspi_vendor_set(d.h, 9, 0x44, 0, b"\0\0\0\0", 4) == -10     # stream out of range
    assert len(d.status()) == 18
    d2 = host.Dspi(1, 5, device=None)
    assert len(d2.status()) == 26 and d2.clear_clips() == 0
    # dspi_process: undefined flag bits are refused (a later ABI may let a flag read further dspi_out members), before anything
    # else is looked at; the defined ones pass the check (a host-only context then has no device: DSPI_E_NODEVICE)
    out = host._Out(None, None, None, None)
    buf = ctypes.create_string_buffer(5 * 48 * 4)
    for bad in (0x40, 0x80, 0x100, 0x80000000, 0x3F | 0x400):
        assert d2.L.dspi_process(d2.h, buf, 16, 1, 48, ctypes.byref(out), bad) == -10, hex(bad)
    for ok in (0, 0x1, 0x3F):
        assert d2.L.dspi_process(d2.h, buf, 16, 1, 48, ctypes.byref(out), ok) == -11, hex(ok)


def test_no_oracle_in_product():
    """The product must not import, link or call anything under oracle/ (the oracle is test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dspi_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in txt and "orclib" not in txt and "liborc" not in txt, os.path.join(dirpath, f)
    out = os.popen(f"ldd {host.LIB_PATH}").read()
    assert "liborc" not in out and "libref" not in out
