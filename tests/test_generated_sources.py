"""The generated band loops in dspi_amd/csrc are exactly what tools/gen_bandloops.py produces (no hand edits, no drift)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bandloops_are_regenerable(tmp_path):
    env = dict(os.environ, BL_OUT=str(tmp_path))
    for k in ("BL_LAT", "BL_LAT2", "BL_NTSETS", "BL_NTSETS2", "BL_NTSETS_V", "BL_VERBOSE"):
        env.pop(k, None)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_bandloops.py")], check=True, env=env, capture_output=True)
    for name in ("dspi_bandloops.inc", "dspi_bandloops_pk.inc", "dspi_bandloops_fma.inc", "dspi_bandloops_pk_fma.inc", "dspi_bandloops_pkv.inc", "dspi_bandloops_pkv_fma.inc"):
        fresh = (tmp_path / name).read_bytes()
        committed = open(os.path.join(ROOT, "dspi_amd", "csrc", name), "rb").read()
        assert fresh == committed, f"{name} differs from the generator's output: run tools/gen_bandloops.py"
