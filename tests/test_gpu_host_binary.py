"""The C host (dspi_amd/csrc/dspi_host.c -> dspi_host, plain C over include/dspi.h) run as a process on the GPU box: preset slot file
in, PCM file in, pair words out, against the oracle fed the same bytes.  Needs an MI355X."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import has_gpu
from orclib import Oracle
from dspi_amd import wire as W, workloads as WL

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no GPU")]

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dspi_amd", "csrc", "dspi_host")


@pytest.mark.parametrize("flavor,load", [(1, "slot"), (1, "bulk"), (W.F32_FMA, "slot"), (0, "slot")])
def test_dspi_host_against_oracle(tmp_path, flavor, load):
    fs, B, blocks, calls, vol_db = 48000, 48, 20, 3, -20
    ref = Oracle(flavor); assert ref.load_bulk(WL.full_chain_blob(flavor)) == 0
    image = ref.save_slot(4) if load == "slot" else ref.collect_bulk()
    pcm = WL.synth_pcm16(1, B * blocks, fs, first_stream=3)[0]
    (tmp_path / "preset.bin").write_bytes(image)
    (tmp_path / "pcm.raw").write_bytes(np.ascontiguousarray(pcm).tobytes())
    r = subprocess.run([HOST, "-f", ("f32fma" if getattr(flavor, "fma", False) else "f32") if flavor else "q28", "-s", "70", "-r", str(fs), "-b", str(B), "-n", str(blocks), "-c", str(calls),
                        "-P" if load == "slot" else "-B", str(tmp_path / "preset.bin"), "-i", str(tmp_path / "pcm.raw"), "-o", str(tmp_path / "pairs.raw"), "-v", str(vol_db)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert ("preset_load -> 0" if load == "slot" else "bulk_params_apply -> 0") in r.stdout
    o = Oracle(flavor, detmath=True)
    assert o.set_rate(fs) == 0
    o.set_volume(vol_db * 256)
    assert (o.load_slot(image) if load == "slot" else o.load_bulk(image)) == 0
    for _ in range(calls):
        pairs, sub, peaks, _ = o.process(pcm, blocks, B)
    got = np.frombuffer((tmp_path / "pairs.raw").read_bytes(), dtype=np.int32).reshape(pairs.shape)      # stream 0, last call
    assert np.array_equal(got, pairs)
    m = re.search(r"stream 0 status \((\d+) bytes\): peaks((?: \d+)+) clip 0x([0-9a-f]+)", r.stdout)
    st = o.status()
    assert m and int(m.group(1)) == len(st)
    assert [int(v) for v in m.group(2).split()] == np.frombuffer(st[:o.C * 2], dtype="<u2").tolist()
    assert int(m.group(3), 16) == int.from_bytes(st[-2:], "little")
