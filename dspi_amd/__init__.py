"""dspi_amd — MI355X-native implementation of the DSPi per-sample DSP chain.

The product is the C-ABI shared library ``dspi_amd/csrc/libdspi_mi355x.so`` (include/dspi.h);
this package is the thin Python mirror used by tests and bench.py:

* ``dspi_amd.wire``      — DSPi blob layouts (bulk params, preset slots, vendor codes)
* ``dspi_amd.workloads`` — BASELINE.json configurations and synthetic PCM
* ``dspi_amd.host``      — ctypes binding of the C-ABI (fails loudly if the HIP library is missing)
"""
__all__ = ["wire", "workloads"]
