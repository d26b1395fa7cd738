"""The mutation fuzz of the untrusted-input parsers (tools/host_fuzz_asan.py: blobs, slot images, flash dumps, vendor requests on host-only
contexts) as a short CPU test: 400 contexts, ~5 000 parser calls, no sanitizer here (profiles/r06_host_sanitizers.md has the long runs under
ASan + UBSan).  Passes if nothing crashes and every context still collects a blob that loads again."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_survive_mutated_input():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_fuzz_asan.py"), "400", "11"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "no crash" in r.stdout
