"""Multi-process path on CPU: world_size 2, gloo.  Streams are sharded by dspi_amd.shard.stream_range; each rank runs its
shard (here through the oracle, since there is no GPU), and the only collectives are the SUM/MAX of the throughput record
plus, for this test, an all_gather of per-stream checksums that must equal a single-process run."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from dspi_amd import shard, workloads as WL  # noqa: E402

TOTAL, FS, B, BLOCKS = 7, 48000, 48, 12


def stream_crc(s: int) -> int:
    from orclib import Oracle
    o = Oracle(1, detmath=True)
    o.set_rate(FS); o.set_volume(-20 * 256)
    assert o.load_bulk(WL.full_chain_blob(1)) == 0
    pcm = WL.synth_pcm16(1, B * BLOCKS, FS, first_stream=s)[0]
    pairs, sub, peaks, _ = o.process(pcm, BLOCKS, B)
    return zlib.crc32(pairs.tobytes() + sub.tobytes() + peaks.tobytes()) & 0x7FFFFFFF


def worker(rank: int, world: int, port: int, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = shard.stream_range(rank, world, TOTAL)
    crcs = torch.zeros(TOTAL, dtype=torch.int64)
    for s in range(first, last):
        crcs[s] = stream_crc(s)
    dist.all_reduce(crcs, op=dist.ReduceOp.SUM)          # disjoint shards: the sum is the concatenation
    frames, elapsed, fps = shard.reduce_throughput(dist, float((last - first) * B * BLOCKS), 1.0 + rank)
    if rank == 0:
        q.put((crcs.tolist(), frames, elapsed, fps))
    dist.barrier()
    dist.destroy_process_group()


def test_stream_range_partitions():
    for total in (1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = shard.stream_range(r, world, total)
                seen += list(range(a, b)) if total < 1000 else [(a, b)]
            if total < 1000:
                assert seen == list(range(total))
            else:
                assert seen[0][0] == 0 and seen[-1][1] == total and all(seen[i][1] == seen[i + 1][0] for i in range(world - 1))


def test_two_rank_gloo_sharding_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    crcs, frames, elapsed, fps = q.get(timeout=240)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert crcs == [stream_crc(s) for s in range(TOTAL)]
    assert frames == TOTAL * B * BLOCKS and elapsed == 2.0 and fps == frames / 2.0
