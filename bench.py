#!/usr/bin/env python3
"""bench.py — DSPi chain throughput on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[2], "config 3"): 65 536 independent stereo streams per GPU, 96 kHz,
96-frame packets, the full RP2350 11-channel chain (preamp, loudness, 10-band master PEQ, leveller
with lookahead, BS2B crossfeed, 2x9 matrix, 9 x 10-band output PEQ, gains, 9 delay lines, int24 /
Q28 conversion).  One "step" = one dspi_process() call = `blocks_per_step` packets for every stream,
inputs and outputs resident in HBM (DSPI_MEM_DEVICE).  Multi-GPU: one process per GPU, streams
sharded per rank, no data-path collective (weak scaling: 65 536 streams per GPU); RCCL only for the
barrier and the max-over-ranks time.

metric  : audio samples/s = frames/s x 11 channels (BASELINE.json "metric")
roofline: algorithmic HBM bytes (40 B compulsory I/O + 8 B per delayed output =
          104 B/frame for this configuration) / measured kernel time, against 8 TB/s.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_FRAME = 104          # config 3: 4 in + 36 out + 8 * 8 delayed outputs
CHANNELS = 11


def hip_runtime():
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64.so not found")


class HipEvents:
    """HIP events recorded on the context's own stream (torch.cuda.Event only sees torch's stream)."""

    def __init__(self, stream_handle: int):
        self.hip = hip_runtime()
        self.stream = ctypes.c_void_p(stream_handle)
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def new(self):
        e = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def record(self, e):
        assert self.hip.hipEventRecord(e, self.stream) == 0

    def elapsed_ms(self, a, b) -> float:
        assert self.hip.hipEventSynchronize(b) == 0
        ms = ctypes.c_float()
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return float(ms.value)


def cpu_baseline(fs: int, block_len: int, budget_s: float = 12.0):
    """The reference C path timed on this box's host cores (rank 0, N=1 only): oracle/_ref (the
    reference's own leaf sources under our restated orchestrator) when its prebuilt .so is present,
    else the standalone restatement.  Bounded sample: every host thread runs one independent stream
    of the same config-3 workload for ~budget_s seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from dspi_amd import workloads as WL

    use_ref = orclib.ref_available(1)
    cores = os.cpu_count() or 1
    blob = WL.full_chain_blob(1)
    blocks = 250                                    # 0.25 s of audio per call
    pcm = WL.synth_pcm16(1, blocks * block_len, fs, mix=False)[0]
    oracles = []
    for _ in range(cores):
        o = orclib.Oracle(1, ref=use_ref, detmath=True)
        o.set_rate(fs); o.set_volume(-20 * 256)
        assert o.load_bulk(blob) == 0
        oracles.append(o)
    # (i) one stream on one core: median of 5 runs of ~0.5 s
    rates = []
    for _ in range(5):
        t0 = time.perf_counter(); n1 = 0
        while time.perf_counter() - t0 < 0.5:
            oracles[0].process(pcm, blocks, block_len, want_peaks=False); n1 += blocks * block_len
        rates.append(n1 / (time.perf_counter() - t0))
    single = sorted(rates)[2]
    # (ii) one independent stream per hardware thread
    counts = [0] * cores
    stop = time.perf_counter() + budget_s

    def work(i):
        while time.perf_counter() < stop:
            oracles[i].process(pcm, blocks, block_len, want_peaks=False)   # ctypes releases the GIL
            counts[i] += blocks * block_len

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    frames = sum(counts)
    return {
        "value": frames * CHANNELS / dt, "unit": "samples/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": f"{cores} threads x 1 stream each, config-3 chain, {frames // cores} frames/thread in {dt:.1f} s "
                  f"({'reference leaf C (oracle/_ref) under the restated orchestrator' if use_ref else 'oracle restatement'}, gcc -O3, FTZ|DAZ)",
        "frames_per_s": frames / dt,
        "single_core_frames_per_s": single, "single_core_realtime_x": single / fs,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU")
    ap.add_argument("--blocks-per-step", type=int, default=50, help="packets per dspi_process call (50 ms of audio per stream at 96 kHz)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out-layout", choices=["tiled", "stream"], default="tiled",
                    help="sample-word layout in HBM: the kernel's native tiles (DSPI_OUT_TILED) or stream-major S/PDIF pair buffers")
    args = ap.parse_args()

    import torch
    from dspi_amd import workloads as WL
    from dspi_amd.host import Dspi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernel is the only audio path")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)          # before the process group: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") over xGMI; DSPI_BENCH_BACKEND=gloo exists only to smoke-test the multi-rank control flow on a
        # box with fewer GPUs than ranks (ranks then share devices)
        backend = os.environ.get("DSPI_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    FS, B = 96000, 96
    S, NB = args.streams, args.blocks_per_step
    frames = NB * B
    ctx = Dspi(1, S, device=local_rank)
    ctx.set_rate(FS)
    ctx.set_volume(-20 * 256)
    assert ctx.load_bulk(WL.full_chain_blob(1)) == 0

    # synthetic PCM, -6 dBFS white noise, generated on the device (data: "synthetic")
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    pcm = torch.randint(-16384, 16385, (S, frames, 2), dtype=torch.int16, device=dev, generator=g)
    tiled = args.out_layout == "tiled"
    R = ctx.tile_streams()
    tiles = (S + R - 1) // R
    if tiled:      # [tile][output][frame][R] / [tile][frame][R]  (include/dspi.h, DSPI_OUT_TILED)
        pairs = torch.empty((tiles, 8, frames, R), dtype=torch.int32, device=dev)
        sub = torch.empty((tiles, frames, R), dtype=torch.int32, device=dev)
    else:          # [stream][pair][frame][2] / [stream][frame]
        pairs = torch.empty((S, 4, frames, 2), dtype=torch.int32, device=dev)
        sub = torch.empty((S, frames), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, NB, CHANNELS), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()

    def step():
        ctx.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr(), tiled=tiled)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    ev = HipEvents(ctx.hip_stream())
    e0, e1 = ev.new(), ev.new()

    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.record(e0)
    for _ in range(args.steps):
        step()
    ev.record(e1)
    ctx.sync()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t1 = time.perf_counter()
    kernel_ms = ev.elapsed_ms(e0, e1) / args.steps     # one chain-kernel launch per step
    elapsed = t1 - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)       # the only collective: 8 bytes over RCCL/xGMI
        elapsed = float(t.item())

    total_frames = float(world) * S * frames * args.steps
    frames_per_s = total_frames / elapsed
    if rank == 0:
        achieved_gbs = S * frames * BYTES_PER_FRAME / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, tools/prof.sh + tools/prof_summary.py);
        # counters cannot be collected inside the timed run, so this is the committed figure of the latest profile
        traffic, traffic_src, valu_insts = None, None, None
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
            try:
                t = json.load(open(tpath))
                if t.get("out_layout", "stream") == args.out_layout and t.get("hbm_bytes_per_launch"):
                    scale = S * frames / float(t.get("frames_per_launch", 157286400))      # profile and run may differ in packets per launch
                    traffic, traffic_src = t["hbm_bytes_per_launch"] * scale, os.path.basename(tpath)
                    valu_insts = t["valu_insts_per_launch"] * scale if t.get("valu_insts_per_launch") else None
            except Exception:
                pass
        out = {
            "metric": "audio samples/s (whole node), 96 kHz 11-ch 10-band PEQ; % HBM roofline",
            "value": frames_per_s * CHANNELS, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: full RP2350 chain (preamp+loudness+master PEQ+leveller/lookahead+crossfeed+2x9 matrix+9x10-band PEQ+gain+9 delay lines), "
                                   "96 kHz, 96-frame packets, int16 in, 4 S/PDIF pairs + PDM sub out",
                       "out_layout": "tiled [tile][output][frame][128] (DSPI_OUT_TILED)" if tiled else "stream-major [stream][pair][frame][2]",
                       "streams_per_gpu": S, "blocks_per_step": NB, "frames_per_step_per_stream": frames,
                       "frames_per_s": frames_per_s, "realtime_streams": frames_per_s / FS, "parallelism": f"streams sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "chain_kernel_pk<false, true, false, %s>" % ("true" if tiled else "false"), "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_frame": BYTES_PER_FRAME, "frames_per_launch": S * frames,
                         # SURVEY.md §8d asks for both sides of the ridge.  VALU: wave-instructions per launch (SQ_INSTS_VALU of the
                         # committed profile, same workload) / kernel time / (1024 SIMDs x 2.4 GHz / 4 cycles per wave-instruction)
                         "valu_fraction": (valu_insts / (kernel_ms * 1e-3) / (1024 * 2.4e9 / 4.0)) if valu_insts else None,
                         "hbm_fraction_measured_traffic": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "binds": "valu"},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(FS, B)
            except Exception as e:  # the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    ctx.close()
    if dist: dist.destroy_process_group()


if __name__ == "__main__":
    main()
