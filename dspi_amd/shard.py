"""Stream sharding across GPUs: streams are fully independent, so the multi-GPU path is a
partition plus one tiny reduction for the throughput report — no data-path collective.

Used by bench.py (RCCL, one process per GPU) and tests/test_dist_cpu.py (gloo, world_size 2)."""
from __future__ import annotations


def stream_range(rank: int, world: int, total_streams: int) -> tuple[int, int]:
    """Contiguous ranges, gpu = stream_id // ceil(S / n_gpu).  Returns [first, last)."""
    per = -(-total_streams // world)
    first = min(rank * per, total_streams)
    return first, min(first + per, total_streams)


def reduce_throughput(dist, frames_local: float, elapsed_local: float, device=None):
    """sum(frames), max(elapsed) over ranks -> (total_frames, elapsed, frames_per_s).  `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return frames_local, elapsed_local, frames_local / elapsed_local
    import torch
    f = torch.tensor([frames_local], dtype=torch.float64, device=device)
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item()), float(t.item()), float(f.item()) / float(t.item())


def gather_ranks(dist, values, device=None):
    """Every rank's short list of numbers on every rank: [[rank 0's values], [rank 1's], ...] (one all_gather of len(values) doubles per
    rank, after the timed region — the per-rank times and parity verdicts of bench.py's N > 1 line)."""
    vals = [float(v) for v in values]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [vals]
    import torch
    mine = torch.tensor(vals, dtype=torch.float64, device=device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(x) for x in t.tolist()] for t in out]
