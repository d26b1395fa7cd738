"""Two ranks on the GPU box (gloo rendezvous; the box has one MI355X, so the ranks share it): each rank runs the PRODUCT on its
shard of streams (dspi_amd.shard.stream_range), the per-stream checksums are all-reduced, and the result must equal the oracle's
for every stream; bench.py's own `--gpus 2` self-spawn is run the same way and must print one well-formed line for n_gpus = 2.
Needs an MI355X."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from dspi_amd import shard, wire as W, workloads as WL  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no GPU")]

TOTAL, FS, B, BLOCKS = 203, 48000, 48, 10


def crc(pairs, sub, peaks) -> int:
    return zlib.crc32(pairs.tobytes() + sub.tobytes() + peaks.tobytes()) & 0x7FFFFFFF


def worker(rank: int, world: int, port: int, fma: bool, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dspi_amd.host import Dspi
    first, last = shard.stream_range(rank, world, TOTAL)
    d = Dspi(1, last - first, device=rank % torch.cuda.device_count(), fma=fma)
    d.set_rate(FS); d.set_volume(-20 * 256)
    assert d.load_bulk(WL.full_chain_blob(1)) == 0
    pcm = WL.synth_pcm16(last - first, B * BLOCKS, FS, first_stream=first)
    pairs, sub, peaks = d.process_host(pcm, BLOCKS, B)
    crcs = torch.zeros(TOTAL, dtype=torch.int64)
    for s in range(first, last):
        crcs[s] = crc(pairs[s - first], sub[s - first], peaks[s - first])
    dist.all_reduce(crcs, op=dist.ReduceOp.SUM)          # disjoint shards: the sum is the concatenation
    frames, elapsed, fps = shard.reduce_throughput(dist, float((last - first) * B * BLOCKS), 1.0 + rank)
    if rank == 0:
        q.put((crcs.tolist(), frames, elapsed))
    dist.barrier()
    d.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("fma", [True, False])
def test_two_ranks_product_shards_match_oracle(fma):
    from orclib import Oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200) + (50 if fma else 0)
    procs = [ctx.Process(target=worker, args=(r, 2, port, fma, q)) for r in range(2)]
    [p.start() for p in procs]
    crcs, frames, elapsed = q.get(timeout=600)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert frames == TOTAL * B * BLOCKS and elapsed == 2.0
    for s in range(TOTAL):
        o = Oracle(1, detmath=True, fma=fma)
        o.set_rate(FS); o.set_volume(-20 * 256)
        assert o.load_bulk(WL.full_chain_blob(1)) == 0
        pcm = WL.synth_pcm16(1, B * BLOCKS, FS, first_stream=s)[0]
        rp, rs, rk, _ = o.process(pcm, BLOCKS, B)
        assert crcs[s] == crc(rp, rs, rk), s


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_self_spawn_two_ranks(scaling):
    """bench.py --gpus 2 outside a launcher spawns its own ranks; with one GPU on the box they rendezvous over gloo and share it."""
    env = dict(os.environ, DSPI_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"): env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "2048", "--scaling", scaling,
                        "--no-cpu-baseline", "--no-variants"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == scaling and rec["value"] > 0 and rec["steps"] == 2
    per_rank = 2048 if scaling == "weak" else 1024
    assert rec["config"]["streams_per_gpu"] == per_rank, rec["config"]
    # the N > 1 line is complete — every rank's own step time and its own 8-stream oracle check, gathered to rank 0
    pr = rec["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["ms_per_step"] > 0 and x["kernel_ms"] > 0 and x["parity_checked"] == 8 for x in pr)
    assert rec["parity_checked"] == 16 and rec["parity_ranks"] == 2 and len(set(rec["parity_streams"])) == 16
    first1 = per_rank      # rank 1's first global stream (weak: rank * streams; strong: stream_range)
    assert all(s < first1 for s in pr[0]["parity_streams"]) and all(s >= first1 for s in pr[1]["parity_streams"])
    assert abs(rec["ms_per_step"] - max(x["ms_per_step"] for x in pr)) < 0.05 * rec["ms_per_step"]


def test_bench_two_ranks_carry_the_cpu_baseline():
    """... and the reference's C path on the host cores sits next to the N > 1 figure (north_star), timed by rank 0 after the GPU work."""
    env = dict(os.environ, DSPI_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"): env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "1024", "--no-variants", "--no-parity"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    cb = rec["cpu_baseline"]
    assert rec["n_gpus"] == 2 and cb["value"] > 0 and cb["kind"] in ("reference", "port") and cb["cores"] >= 1


def test_bench_rccl_branch_world_size_one():
    """The RCCL branch of bench.py on real hardware: DSPI_BENCH_FORCE_DIST=1 makes a one-rank run create the nccl (= RCCL) process
    group bound to the context's device, pass the barriers around the timed region and all-reduce the elapsed time on a device tensor.
    (N > 1 over xGMI is the driver's 8-GPU run; this proves communicator creation and the collective where one GPU exists.)"""
    env = dict(os.environ, DSPI_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSPI_BENCH_BACKEND", "MASTER_PORT"): env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--streams", "2048",
                        "--no-cpu-baseline", "--no-variants"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["dist"]["world_size"] == 1 and rec["dist"]["backend"].startswith("rccl"), rec.get("dist")
