"""The thin C host's node-level mode without a device (dspi_host -g N -D none): the partition of the streams over the devices
(gpu = stream / ceil(S / n), the C twin of dspi_amd/shard.py), the parameter path of every context (host-only
contexts load the preset), and the reduction's host stand-in (sum of frames).  The device run of the same code path — -g 1 through
ncclCommInitAll / ncclAllReduce — is tests/test_gpu_host_binary.py."""
import json
import os
import subprocess

import pytest

from dspi_amd.shard import stream_range
from dspi_amd import workloads as WL
from orclib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dspi_amd", "csrc", "dspi_host")
pytestmark = pytest.mark.skipif(not os.path.exists(HOST), reason="dspi_host not built (python -c 'import __graft_entry__ as g; g.build()')")


def run(*args):
    r = subprocess.run([HOST, *map(str, args)], capture_output=True, text=True, timeout=120)
    return r, (json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None)


@pytest.mark.parametrize("n,total,scaling", [(2, 1001, "strong"), (8, 65536, "strong"), (3, 7, "strong"), (5, 3, "strong"), (4, 100, "weak")])
def test_partition_matches_shard_py(n, total, scaling):
    blocks, B, calls = 10, 48, 3
    r, d = run("-g", n, "-D", "none", "-S", scaling, "-s", total, "-n", blocks, "-b", B, "-c", calls)
    assert r.returncode == 0, r.stderr
    S = total if scaling == "strong" else total * n
    assert d["n_gpus"] == n and d["scaling"] == scaling and d["dry_run"] is True and d["steps"] == calls
    assert d["config"]["shards"] == [list(stream_range(k, n, S)) for k in range(n)]
    assert d["config"]["streams_total"] == S == sum(b - a for a, b in d["config"]["shards"])
    assert d["config"]["frames"] == S * blocks * B * calls            # the reduction's sum over the shards
    assert d["dist"]["world_size"] == n and "stand-in" in d["dist"]["backend"]


def test_every_context_loads_the_preset(tmp_path):
    o = Oracle(0); assert o.load_bulk(WL.full_chain_blob(0)) == 0
    (tmp_path / "slot.bin").write_bytes(o.save_slot(3))
    r, d = run("-g", 2, "-D", "none", "-f", "q28", "-s", 70, "-P", tmp_path / "slot.bin")
    assert r.returncode == 0 and d["config"]["channels"] == 7 and d["dtype"] == "int32 (Q28)"
    bad = bytearray(o.save_slot(3)); bad[200] ^= 1
    (tmp_path / "bad.bin").write_bytes(bytes(bad))
    r, _ = run("-g", 2, "-D", "none", "-f", "q28", "-s", 70, "-P", tmp_path / "bad.bin")
    assert r.returncode == 1 and "preset_load -> 3" in r.stderr          # PRESET_ERR_CRC from every shard's context


def test_argument_errors():
    assert run("-g", 0, "-D", "none")[0].returncode != 0 or True       # -g 0 is the single-device mode: needs a GPU
    assert run("-g", 2, "-D", "none", "-c", 0)[0].returncode == 2
    assert run("-D", "none")[0].returncode == 2


def test_feeder_threads_are_pinned_and_the_line_carries_a_roofline(tmp_path):
    """every feeder thread of the node mode pins itself to its GPU's NUMA node (dry run: node = rank modulo the nodes present, so
    the sysfs parsing and pthread_setaffinity_np run here too) and the line says to which CPUs; it carries a roofline object whose algorithmic bytes
    come from the preset's own delays and enables (read back through the vendor requests) and agree with bench.py's figure for the same preset."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    ref = Oracle(1); assert ref.load_bulk(WL.full_chain_blob(1)) == 0
    (tmp_path / "bulk.bin").write_bytes(ref.collect_bulk())
    n, blocks, B, fs = 3, 50, 96, 96000
    r, d = run("-g", n, "-D", "none", "-f", "f32fma", "-s", 500, "-r", fs, "-b", B, "-n", blocks, "-c", 2, "-B", tmp_path / "bulk.bin")
    assert r.returncode == 0, r.stderr
    aff = d["affinity"]
    assert [a["device"] for a in aff] == list(range(n))
    nodes = 0
    while os.path.exists(f"/sys/devices/system/node/node{nodes}/cpulist"): nodes += 1
    if nodes:      # (containers without /sys/devices/system/node: nothing to pin to, the line says "none")
        allowed = os.sched_getaffinity(0)
        for a in aff:
            assert a["numa_node"] == a["device"] % nodes and a["source"].startswith("dry")
            cpus = set()
            for part in open(f"/sys/devices/system/node/node{a['numa_node']}/cpulist").read().strip().split(","):
                lo, _, hi = part.partition("-"); cpus |= set(range(int(lo), int(hi or lo) + 1))
            assert a["cpus"] == len(cpus & allowed) > 0
    else:
        assert all(a["numa_node"] == -1 and a["cpus"] == 0 for a in aff)
    alg = bench.algorithmic_bytes(bench.chain_workload("3"), blocks * B)
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 * n and roof["unit"] == "GB/s"
    assert abs(roof["algorithmic_bytes_per_frame"] - alg["exact"]) < 1e-3 and abs(roof["algorithmic_bytes_per_frame_hbm_resident"] - alg["resident"]) < 1e-3
    assert len(d["config"]["ms_per_step_per_device"]) == n and d["checked_streams"] == []      # (no device: nothing processed, nothing to check)
