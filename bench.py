#!/usr/bin/env python3
"""bench.py — DSPi chain throughput on MI355X (driver contract: one JSON line on rank 0).

Default workload (BASELINE.json configs[2], "config 3"): 65 536 independent stereo streams per GPU, 96 kHz, 96-frame
packets, the full RP2350 11-channel chain (preamp, loudness, 10-band master PEQ, leveller with lookahead, BS2B crossfeed,
2x9 matrix, 9 x 10-band output PEQ, gains, 9 delay lines, int24 / Q28 conversion).  One "step" = one dspi_process() call
= `blocks_per_step` packets for every stream, inputs and outputs resident in HBM (DSPI_MEM_DEVICE).

    python bench.py                         1 GPU, config 3
    python bench.py --gpus N                N ranks on one node: spawned here (torchrun) when not already under a launcher
    python bench.py --scaling strong        65 536 streams in total, split over the ranks (dspi_amd/shard.py)
    python bench.py --config {2,2b,5,perstream,perstream_eq,pdm,spdif,i2s}     the other BASELINE configs / SURVEY section 8f consumers, same JSON shape

Multi-GPU: one process per GPU, streams sharded per rank, no data-path collective; RCCL only for the barrier and the
max-over-ranks time.  DSPI_BENCH_BACKEND=gloo lets the ranks share GPUs (control-flow smoke test on a 1-GPU box).

What a line says (config 3):
  value       frames/s x 11 channels, whole job, on the PRIMARY variant: float contract `--contract` (default fma = the firmware
              as built: GCC's contracted multiply-adds, include/dspi.h DSPI_FLOAT_CONTRACT_FMA), word layout `--out-layout`,
              input = SURVEY section 8d's synthetic mix (70 % noise / 10 % sweep / 10 % bursts / 5 % silence / 5 % square)
  also        the same step measured on the other variants (other contract, other layout, white-noise input), N=1 only
  roofline    algorithmic HBM bytes / measured kernel time against 8 TB/s.  `frac` is the STRICT figure: the bytes a launch of T frames must
              move when the lines keep the reference's history exactly — compulsory I/O + 4 B per delayed output for the launch's first
              min(dly, T) frames (line reads) + 4 B for its last min(line length, T) frames (line writes): 77 B/frame for config 3 at 50
              packets per launch.  Beside it `frac_hbm_resident` (SURVEY section 8d's 104 B rule: every delayed sample written and
              read) and `frac_launch_span` (8d's lower figure, 8 * min(dly, T) / T: 59.4 B).  `traffic_ratio` = counter bytes / the
              strict bytes.  The chain is on the VALU side of the ridge: `valu_fraction_at_sclk` is the issue fraction at the clock the
              socket's power limit allows; `binds` says "power" when that clock is below 0.9 x the part's maximum.
  configs     (default run, N=1) short driver-timed runs of BASELINE configs 2, 2b and 5, each with its own roofline and its own
              post-run oracle check; `value` stays config 3
  cpu_baseline  the reference C path on this box's host cores (a reported baseline, not the target)
  realtime_call  (default configuration, N=1) the call as the firmware makes it: ONE packet per dspi_process(), host buffers, one stream
              of the same preset, through the C host (dspi_host -rt): p50 / p99 us per call, every word of every call checked against
              the oracle.  Reported beside the batched figure; never part of `value`.  `realtime_call_q28`: the same for the RP2040 Q28
              flavour (one 48-frame packet at 48 kHz, BASELINE config 5's preset).
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 4.0      # 1024 SIMDs x 2.4 GHz / 4 cycles per wave-instruction


# ---------------------------------------------------------------------------------------------------------------------
# HIP events on the context's own stream (torch.cuda.Event only sees torch's stream)
# ---------------------------------------------------------------------------------------------------------------------
def hip_runtime():
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64.so not found")


class HipEvents:
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


class PowerSampler:
    """Socket power and shader clock of one GPU while the timed region runs: a thread polling librocm_smi64 (ctypes, ~100 Hz).
    The numbers go into the line as roofline.power_w / sclk_mhz (mean over the timed region) so that the claim "the kernel runs
    into the board's power cap" is in the driver-run record and not only in a builder-side log.  Fails soft: no library, no
    sensor -> the fields are null."""

    class _Freq(ctypes.Structure):
        _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * 33)]

    def __init__(self, device_index: int):
        self.ok, self.samples, self._stop, self.cap_w, self.max_mhz = False, [], threading.Event(), None, None
        self.dev = ctypes.c_uint32(device_index)
        try:
            self.smi = ctypes.CDLL("librocm_smi64.so")
        except OSError:
            try:
                self.smi = ctypes.CDLL("/opt/rocm/lib/librocm_smi64.so")
            except OSError:
                return
        try:
            if self.smi.rsmi_init(ctypes.c_uint64(0)) != 0: return
            cap = ctypes.c_uint64(0)
            if self.smi.rsmi_dev_power_cap_get(self.dev, ctypes.c_uint32(0), ctypes.byref(cap)) == 0: self.cap_w = cap.value / 1e6
            self.ok = self._read() is not None
        except Exception:
            self.ok = False

    def _read(self):
        pw = ctypes.c_uint64(0)
        if self.smi.rsmi_dev_current_socket_power_get(self.dev, ctypes.byref(pw)) != 0:
            if self.smi.rsmi_dev_power_ave_get(self.dev, ctypes.c_uint32(0), ctypes.byref(pw)) != 0: return None
        f = PowerSampler._Freq()
        mhz = None
        if self.smi.rsmi_dev_gpu_clk_freq_get(self.dev, ctypes.c_int(0), ctypes.byref(f)) == 0 and f.current < 33:
            mhz = f.frequency[f.current] / 1e6
            if 0 < f.num_supported <= 33: self.max_mhz = max(f.frequency[i] for i in range(f.num_supported)) / 1e6      # the top of the part's clock table
        return (time.perf_counter(), pw.value / 1e6, mhz)

    def start(self):
        if not self.ok: return
        def loop():
            while not self._stop.is_set():
                r = self._read()
                if r: self.samples.append(r)
                time.sleep(0.008)
        self.th = threading.Thread(target=loop, daemon=True)
        self.th.start()

    def stop(self):
        if not self.ok: return
        self._stop.set(); self.th.join()

    def window(self, t0, t1):
        """mean / max over the samples taken in [t0, t1] (perf_counter times) plus the first one after it.  The caller has kept the device
        under the same load for >= 0.6 s before t0 (timed_steps: untimed pre-load steps), so the whole window is the settled state."""
        w = [x for x in self.samples if t0 <= x[0] <= t1 + 0.02]
        if not w: return None
        pw = [x[1] for x in w]; ck = [x[2] for x in w if x[2]]
        return {"power_w": sum(pw) / len(pw), "power_w_max": max(pw), "sclk_mhz": (sum(ck) / len(ck)) if ck else None, "power_cap_w": self.cap_w, "sclk_max_mhz": self.max_mhz, "power_samples": len(w)}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(flavor: int, fs: int, block_len: int, blob, channels: int, fma: bool, vol: int, what: str, budget_s: float = 25.0):
    """The reference C path timed on this box's host cores.  Default: oracle/_ref's leaf build (the reference's leaf sources compiled
    unmodified, under the restated orchestrator); DSPI_CPU_BASELINE=fw: the firmware build (the reference's own process_audio_packet +
    leaf sources compiled in place over SDK stand-ins, oracle/ref_fw.c; one private library copy per thread because the firmware keeps its
    state in globals); without oracle/_ref the restatement ("port").  Protocol: CLOCK_MONOTONIC around >= 5 s of work, median of 5 — all host threads,
    one independent stream each — and 5 x 1 s for the single-core figure; gcc -O3 -march=x86-64-v3, FTZ|DAZ."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from dspi_amd import workloads as WL

    # Default: the strict reference build (the leaf sources alone, compiled unmodified, no stand-in for anything) under the restated packet loop.
    # DSPI_CPU_BASELINE=fw times the firmware build instead — the reference's own process_audio_packet, but its usb_audio.c compiles over
    # stand-ins for the un-vendored pico-sdk's headers, which this task's rules do not accept as a reference build
    want = os.environ.get("DSPI_CPU_BASELINE", "leaf")
    stand_ins = None
    if want == "fw" and orclib.ref_available(flavor, "fw", fma):
        ref, kind, how = "fw", "reference", "the reference's process_audio_packet + leaf sources compiled in place (oracle/_ref/libref_fw_*)"
        stand_ins = "usb_audio.c includes pico-sdk headers (un-vendored submodule): compiled over oracle/ref_stub_sdk (types, attribute macros, hardware entry points; no DSP code); the default (DSPI_CPU_BASELINE=leaf) times the leaf sources alone"
    elif want in ("fw", "leaf") and orclib.ref_available(flavor, "ref", fma): ref, kind, how = True, "reference", "reference leaf C (oracle/_ref) under the restated orchestrator"
    else: ref, kind, how = False, "port", "oracle restatement"
    cores = os.cpu_count() or 1
    blocks = max(1, int(0.25 * fs / block_len))                     # 0.25 s of audio per call
    pcm = WL.synth_pcm16(1, blocks * block_len, fs, mix=False)[0]
    oracles = []
    for _ in range(cores):
        o = orclib.Oracle(flavor, ref=ref, detmath=True, fma=fma)
        o.set_rate(fs); o.set_volume(vol)
        assert o.load_bulk(blob) == 0
        oracles.append(o)

    def run_for(seconds, idx):
        counts = [0] * len(idx)
        stop = time.perf_counter() + seconds

        def work(j, i):
            while time.perf_counter() < stop:
                oracles[i].process(pcm, blocks, block_len, want_peaks=False)   # ctypes releases the GIL
                counts[j] += blocks * block_len
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(j, i)) for j, i in enumerate(idx)]
        [t.start() for t in th]
        [t.join() for t in th]
        return sum(counts) / (time.perf_counter() - t0)

    single = sorted(run_for(1.0, [0]) for _ in range(5))[2]
    per = budget_s / 5.0
    rates = sorted(run_for(per, list(range(cores))) for _ in range(5))
    fps = rates[2]
    return {
        "value": fps * channels, "unit": "samples/s", "cores": cores, "kind": kind,
        "sample": f"{what}: {cores} threads x 1 stream each, median of 5 runs of {per:.0f} s ({how}; gcc -O3 -march=x86-64-v3, "
                  f"{'-ffp-contract=fast (FMA), ' if fma else '-ffp-contract=off, '}FTZ|DAZ)",
        "frames_per_s": fps, "frames_per_s_min_max": [rates[0], rates[-1]],
        "single_core_frames_per_s": single, "single_core_realtime_x": single / fs,
        "stand_ins": stand_ins,
    }


# ---------------------------------------------------------------------------------------------------------------------
# synthetic input on the device
# ---------------------------------------------------------------------------------------------------------------------
def synth_device(torch, dev, S, frames, fs, seed, mix, first_stream=0):
    """int16 [S][frames][2].  mix: stream classes by (global stream index % 20): 0-13 white noise +-16384 (-6 dBFS), 14-15 log
    sine sweep 20 Hz -> 20 kHz over the buffer at -12 dBFS with L/R 90 degrees apart, 16-17 speech-like bursts (first half of the
    buffer at -30 dBFS, second at -6 dBFS: leveller attack / release and gate), 18 digital silence (denormal / FTZ decay after
    the warm-up), 19 full-scale square (clip flags, s24 saturation).  The buffer is one step and is fed again every step."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pcm = torch.randint(-16384, 16385, (S, frames, 2), dtype=torch.int16, device=dev, generator=g)
    if not mix:
        return pcm
    cls = (torch.arange(first_stream, first_stream + S, device=dev) % 20)
    t = torch.arange(frames, device=dev, dtype=torch.float64) / fs
    dur = frames / fs
    k = np.log(20000.0 / 20.0) / dur
    phase = 2 * np.pi * 20.0 * (torch.exp(k * t) - 1.0) / k
    amp = 32767.0 * 10 ** (-12 / 20.0)
    sweep = torch.stack([torch.sin(phase), torch.cos(phase)], dim=-1).mul(amp).to(torch.int16)
    pcm[(cls == 14) | (cls == 15)] = sweep
    half = frames // 2
    quiet = (cls == 16) | (cls == 17)
    pcm[quiet, :half] = (pcm[quiet, :half].to(torch.float32) * (10 ** (-30 / 20.0) / 0.5)).to(torch.int16)
    pcm[cls == 18] = 0
    sq = torch.where((torch.arange(frames, device=dev) // 24) % 2 == 0, 32767, -32768).to(torch.int16)
    pcm[cls == 19] = torch.stack([sq, (-sq.to(torch.int32) - 1).to(torch.int16)], dim=-1)
    return pcm


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
def chain_workload(name):
    from dspi_amd import workloads as WL
    if name == "3":
        return dict(flavor=1, fs=96000, B=96, streams=65536, blocks=50, channels=11, vol=-20 * 256, blob=WL.full_chain_blob(1),
                    text="BASELINE config 3: full RP2350 chain (preamp+loudness+master PEQ+leveller/lookahead+crossfeed+2x9 matrix+9x10-band PEQ+gain+"
                         "9 delay lines), 96 kHz, 96-frame packets, int16 in, 4 S/PDIF pairs + PDM sub out")
    if name in ("2", "2b"):
        return dict(flavor=1, fs=48000, B=48, streams=4096, blocks=2000, channels=2, vol=-10 * 256, blob=WL.config2_blob(name == "2b"), enabled_only=True,
                    text="BASELINE config 2: 4 096 streams, 48 kHz, 48-frame packets, 2 000 packets per launch, master L/R 10-band PEQ only (%s), "
                         "outputs 0-1 pass-through; DSPI_OUT_ENABLED_ONLY: the three disabled pairs and the sub are not zero-filled" % ("all-biquad variant, bands 6.5-20 kHz" if name == "2b" else "SVF below 6.4 kHz + biquad above"))
    if name == "5":
        return dict(flavor=0, fs=48000, B=48, streams=16384, blocks=50, channels=7, vol=-20 * 256, blob=WL.full_chain_blob(0),
                    text="BASELINE config 5: RP2040 Q28 fixed-point 7-channel chain (5 outputs, delays <= 40 ms), 16 384 streams, 48 kHz, 48-frame packets")
    if name == "perstream":
        return dict(flavor=1, fs=96000, B=96, streams=65536, blocks=50, channels=11, vol=-20 * 256, blob=WL.full_chain_blob(1), perstream=True,
                    text="SURVEY 8f-1: config 3 with 65 536 DISTINCT presets of one structure and identical filters (preamp per stream; one parameter image "
                         "per stream; packed kernel, gains / volumes / leveller numbers per lane from value tiles, band coefficients shared)")
    if name == "perstream_eq":
        return dict(flavor=1, fs=96000, B=96, streams=65536, blocks=50, channels=11, vol=-20 * 256, blob=WL.full_chain_blob(1), perstream="eq",
                    text="SURVEY 8f-1: config 3 with 65 536 DISTINCT presets of one structure whose FILTERS differ too (a master band's gain per stream: "
                         "every band coefficient is then read per lane from the value tiles, 180 B/frame of extra traffic)")
    raise SystemExit(f"unknown config {name}")


def algorithmic_bytes(w, frames_per_launch):
    """Per-frame HBM bytes the path must move, three ways:
      exact    — the STRICT figure (roofline.frac): compulsory I/O + per delayed output 4 B for the launch's first min(dly, T) frames (the
                 samples the previous launch left in the line) + 4 B for its last min(L, T) frames (the line keeps the reference's whole
                 history of L = 4 096 / 2 048 samples: a later, longer delay reads it, usb_audio.c:1944-1951 clears nothing);
      resident — SURVEY 8d's rule with HBM-resident lines: every delayed sample written and read, 8 B per delayed output and frame;
      span     — SURVEY 8d's launch-span figure: 8 * min(dly, T) / T."""
    from dspi_amd import wire as W
    flavor, fs, blob = w["flavor"], w["fs"], w["blob"]
    C, N, _, P, _ = W.dims(flavor)
    if w["channels"] == 2:                       # config 2: 4 B in + 2 x int32 out (the one live pair); no delay is active
        return dict(exact=12.0, resident=12.0, span=12.0)
    io = 4 + 4 * (N - 1) + 4                     # int16 stereo in + (N-1) int32 S/PDIF words + the Q28 sub word: 40 B float / 24 B Q28
    max_d = 4096 if flavor else 2048
    T = float(frames_per_launch)
    full = span = exact = float(io)
    for o in range(N):
        ms = float(blob["outputs"][o]["delay_ms"]) + (128.0 / fs * 1000.0 if o == N - 1 else 0.0)
        d = min(max(int(ms * fs / 1000.0), 0), max_d)          # a delay clamped to the line length aliases to 0 samples, but the firmware
        if d > 0 and blob["outputs"][o]["enabled"]:            # still writes and re-reads the line for it (dly > 0, usb_audio.c:899-911)
            full += 8.0
            span += 8.0 * min(1.0, d / T)
            exact += 4.0 * min(d, T) / T + 4.0 * min(max_d, T) / T
    return dict(exact=exact, resident=full, span=span)


def latest_profile(kernel_key, contract, layout):
    """HBM bytes and VALU wave-instructions per frame from the committed PMC profile of this kernel variant (tools/prof.sh +
    tools/prof_summary.py; counters cannot be collected inside the timed run)."""
    best = None
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
        try:
            t = json.load(open(tpath))
        except Exception:
            continue
        if t.get("kernel_key", "chain3") != kernel_key or t.get("contract", "canonical") != contract or t.get("out_layout", "stream") != layout:
            continue
        if t.get("hbm_bytes_per_launch") and t.get("frames_per_launch"):
            best = dict(source=os.path.basename(tpath), src_sha16=t.get("src_sha16"), hbm_bytes_per_frame=t["hbm_bytes_per_launch"] / float(t["frames_per_launch"]),
                        valu_insts_per_frame=(t["valu_insts_per_launch"] / float(t["frames_per_launch"])) if t.get("valu_insts_per_launch") else None)
    return best


def latest_power_model():
    """profiles/power_model_r*.json (tools/ablate_power.py at the round's HEAD): joules per frame of the chain's arithmetic alone and joules per
    HBM byte of a streaming copy, both measured on the builder's box through the same sampler as roofline.power_w."""
    best = None
    for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "power_model_r*.json"))):
        try:
            t = json.load(open(pth))
            if t.get("arith_j_per_frame") and t.get("hbm_j_per_byte"): best = dict(t, source=os.path.basename(pth))
        except Exception:
            continue
    return best


def side_run(extra, env=None, timeout=600):
    """bench.py run again as a child (same box, minutes apart) for one more variant of config 3: returns the child's line (dict) or an error record."""
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-variants", "--no-realtime", "--no-configs", "--no-side-runs"] + extra
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSPI_BENCH_FORCE_DIST"): e.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            if "PARITY FAILURE" in (r.stdout + r.stderr): raise SystemExit("bench.py: PARITY FAILURE in a side run: " + " ".join(extra))
            return {"error": (r.stderr or r.stdout)[-300:]}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}


RT_CALLS = 30000      # one-packet calls per flavour in the default line (a 1-in-3 000 outlier needs more than 3 000 calls to be seen twice)


def rt_record(what, r):
    """The one-packet-per-call record of the line: percentiles AND the tail (calls longer than the packet they carry, log2 histogram, the worst
    calls' indices, the library's polling record)."""
    keys = ("calls", "p50_us", "p99_us", "p99_9_us", "p99_99_us", "max_us", "packet_us", "over_packet_time", "n_over_packet", "hist_log2_us", "worst_calls", "direct_path", "first_calls",
            "first_calls_max_us", "parity")
    rec = {"what": what}
    rec.update({k: r.get(k) for k in keys})
    return rec


# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="3", choices=["3", "2", "2b", "5", "perstream", "perstream_eq", "pdm", "spdif", "i2s"])
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (weak) or in total (strong); 0 = the config's own")
    ap.add_argument("--blocks-per-step", type=int, default=0, help="packets per dspi_process call; 0 = the config's own")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--contract", choices=["fma", "canonical"], default="fma", help="float contract of the primary figure (include/dspi.h)")
    ap.add_argument("--out-layout", choices=["tiled", "stream"], default="stream",
                    help="sample-word layout in HBM: the kernel's native tiles (DSPI_OUT_TILED) or the firmware's stream-major S/PDIF pair buffers")
    ap.add_argument("--input", choices=["mix", "noise"], default="mix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="measure only the primary variant")
    ap.add_argument("--no-realtime", action="store_true", help="skip the one-packet-per-call measurement (dspi_host -rt, one stream) of the default configuration")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-run oracle check of the timed context (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="default run: skip the short runs of BASELINE configs 2, 2b and 5 that follow config 3")
    ap.add_argument("--no-side-runs", action="store_true", help="default run: skip the child runs (200 packets per launch on tiled words; the previous round's library on this box)")
    args = ap.parse_args()

    # ---- N ranks without an external launcher: become the launcher ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    from dspi_amd import wire as W
    from dspi_amd.host import Dspi
    from dspi_amd.shard import stream_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("DSPI_BENCH_BACKEND") or "nccl"      # (an empty value means the default, not a backend called "")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernel is the only audio path")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks need {world} GPUs, {torch.cuda.device_count()} visible (DSPI_BENCH_BACKEND=gloo shares devices for a control-flow smoke test)")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)          # before the process group: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local_rank)
    dist = None
    # DSPI_BENCH_FORCE_DIST=1: take the process-group path at world size 1 too — communicator creation, the barriers around the timed
    # region and the device-tensor all_reduce(MAX) then run over RCCL on a one-GPU box (tests/test_gpu_dist.py)
    if world > 1 or os.environ.get("DSPI_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1: os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":      # RCCL over xGMI
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.config in ("pdm", "spdif", "i2s"):
        out = bench_consumer(args, torch, dev, rank, world, dist, backend)
    else:
        out = bench_chain(args, torch, dev, rank, world, dist, backend, Dspi, W, stream_range)
    # The other BASELINE configurations, driver-timed in the same line: short runs (a few launches after the pre-load; each context is checked
    # against the oracle afterwards like the primary one).  `value` stays config 3.
    if args.config == "3" and world == 1 and out is not None and not args.no_configs and not args.streams and not args.blocks_per_step:
        out["configs"] = {}
        for name, steps, warm in (("2", 5, 2), ("2b", 5, 2), ("5", 20, 3)):
            a2 = argparse.Namespace(**vars(args))
            a2.config, a2.steps, a2.warmup, a2.no_variants, a2.no_realtime, a2.no_cpu_baseline = name, steps, warm, True, True, True
            a2.contract, a2.out_layout, a2.input = "fma", "stream", "mix"
            try:
                r = bench_chain(a2, torch, dev, rank, world, None, backend, Dspi, W, stream_range)
                out["configs"][name] = {"workload": r["config"]["workload"], "dtype": r["dtype"], "steps": steps, "warmup": warm, "preload_steps": r["config"].get("preload_steps", 0),
                                        "streams": r["config"]["streams_per_gpu"], "blocks_per_step": r["config"]["blocks_per_step"], "channels": r["config"]["channels"],
                                        "ms_per_step": r["ms_per_step"], "frames_per_s": r["config"]["frames_per_s"], "value": r["value"], "unit": r["unit"],
                                        "realtime_streams": r["config"]["realtime_streams"], "enabled_only": r["config"].get("enabled_only"),
                                        "roofline": r["roofline"], "parity_checked": r.get("parity_checked", 0), "parity_streams": r.get("parity_streams"),
                                        "parity_launches_replayed": r.get("parity_launches_replayed")}
            except SystemExit as e:      # a parity failure of a side configuration fails the run like the primary one's
                raise
            except Exception as e:
                out["configs"][name] = {"error": str(e)[:300]}
    if dist and out is not None:      # (ranks other than 0 return None)
        out["dist"] = {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "world_size": world,
                       "collective": "all_reduce(SUM) of the frames + all_reduce(MAX) of the elapsed time, 8 bytes each, then one all_gather of 12 doubles per rank (its times, its parity verdict and checked streams), all after the timed region (dspi_amd/shard.py)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()      # (rank 0 has timed the CPU baseline meanwhile: every rank leaves the group together)
        dist.destroy_process_group()


def timed_steps(args, torch, dist, backend, dev, ctx, step, min_load_s=0.0):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
    min_load_s: when the W warm-up steps are shorter than this, further UNTIMED steps follow them ("preload_steps") until the device has
    been under this load for that long — socket power and shader clock need ~0.5 s to settle, and the sampler's window is the timed region."""
    tw = time.perf_counter()
    for _ in range(args.warmup):
        step()
    ctx.sync()
    tw = time.perf_counter() - tw
    extra = 0
    if min_load_s > 0.0 and tw < min_load_s:
        per = tw / max(1, args.warmup) if args.warmup else 0.01
        extra = min(400, int((min_load_s - tw) / max(per, 1e-4)) + 1)
        for _ in range(extra):
            step()
        ctx.sync()
    timed_steps.preload_steps = extra
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
    elapsed = t1 - t0
    timed_steps.window = (t0, t1)
    kernel_ms = ev.elapsed_ms(e0, e1) / args.steps
    return elapsed, kernel_ms      # (this rank's; the reduction over ranks is dspi_amd/shard.py:reduce_throughput, the job's only collective)


def bench_chain(args, torch, dev, rank, world, dist, backend, Dspi, W, stream_range):
    w = chain_workload(args.config)
    flavor, FS, B, CH = w["flavor"], w["fs"], w["B"], w["channels"]
    NB = args.blocks_per_step or w["blocks"]
    total = args.streams or w["streams"]
    if args.scaling == "strong":
        first, last = stream_range(rank, world, total)           # contiguous ranges, SURVEY section 8e
    else:
        first, last = rank * total, (rank + 1) * total
    S = last - first
    frames = NB * B
    _, N, P, _, _ = W.dims(flavor)          # channels, outputs, S/PDIF pairs
    pcm_cache = {}

    def per_stream_requests(s):
        """SURVEY 8f-1 workloads: the vendor requests that make stream s's preset its own (the same list for the context and for the checker)."""
        import struct
        reqs = []
        if w.get("perstream"):
            reqs.append((W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.001 * s)))
            if w["perstream"] == "eq":
                ch = int(os.environ.get("DSPI_BENCH_EQ_CH", "0"))      # which channel's first band differs per stream (0-1 master, 2.. outputs)
                p = w["blob"]["eq"][ch][1]
                reqs.append((W.REQ["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, 1, int(p["type"]), 0, float(p["freq"]), float(p["q"]), 1.0 + 0.0001 * s)))
        return reqs

    def parity_check(ctx, fma, pcm, pairs, sub, peaks, tiled, launches, k=32):
        """After the timed region, outside it: K sampled streams of the TIMED context — the words, sub words and peaks its last launch left in
        the output buffers — against the CPU oracle replaying every launch the context has run (warm-up + pre-load + timed steps, the same input
        buffer each time, state carried from launch to launch).  The sample: the context's first and last streams, one stream of every input
        class (noise, sweep, bursts, silence, square) in every third of the context, and streams spread over the workgroup rows
        (different rows, different lanes, both halves of a lane).  The checker never touches the product path; a mismatch fails the run."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        from concurrent.futures import ThreadPoolExecutor
        cand = [0, 1, 14, (S // 3) // 20 * 20 + 16, (S // 2) // 20 * 20 + 18, (2 * S // 3) // 20 * 20 + 19, S - 2, S - 1]
        cand += [(S // 3) // 20 * 20 + c for c in (15, 17, 18, 19)] + [(2 * S // 3) // 20 * 20 + c for c in (3, 14, 16, 18)]
        R = ctx.tile_streams()
        rows = max(1, (S + R - 1) // R)
        j = 0
        while len(set(min(max(c, 0), S - 1) for c in cand)) < min(k, S) and j < 4 * k:      # one stream in each of k rows spread over the context, lane varying
            cand.append(((j * rows) // k) * R + (37 * j + 5) % R)
            j += 1
        sample = sorted({min(max(c, 0), S - 1) for c in cand})[:k]
        t0 = time.perf_counter()

        def one(s):
            o = orclib.Oracle(flavor, detmath=True, fma=fma)
            o.set_rate(FS); o.set_volume(w["vol"])
            assert o.load_bulk(w["blob"]) == 0
            for req, wv, pl in per_stream_requests(first + s): o.vendor_set(req, wv, pl)
            x = pcm[s].cpu().numpy()
            for _ in range(launches):
                rp, rs, rk, _ = o.process(x, NB, B, 16)      # (ctypes releases the GIL: the streams replay in parallel)
            return rp, rs, rk, o.status()
        with ThreadPoolExecutor(max_workers=min(len(sample), os.cpu_count() or 1)) as ex:
            refs = list(ex.map(one, sample))
        for s, (rp, rs, rk, rstat) in zip(sample, refs):
            if tiled:
                gp = pairs[s // R, :, :, s % R].cpu().numpy().reshape(P, 2, frames).transpose(0, 2, 1)
                gs = sub[s // R, :, s % R].cpu().numpy()
            else:
                gp, gs = pairs[s].cpu().numpy(), sub[s].cpu().numpy()
            gk = peaks[s].cpu().numpy().view(np.uint16)
            live = [pr for pr in range(P) if not w.get("enabled_only") or int(np.abs(rp[pr]).max()) > 0]      # (DSPI_OUT_ENABLED_ONLY: silent pairs stay unwritten)
            if not all(np.array_equal(rp[pr], gp[pr]) for pr in live) or not np.array_equal(rk, gk) or \
               not (np.array_equal(rs, gs) or (w.get("enabled_only") and int(np.abs(rs).max()) == 0)):
                raise SystemExit(f"bench.py: PARITY FAILURE — stream {first + s} of the timed configuration differs from the oracle after {launches} launches")
            if rstat != ctx.status(s):
                raise SystemExit(f"bench.py: PARITY FAILURE — status bytes of stream {first + s} differ from the oracle")
        return {"parity_checked": len(sample), "parity_streams": [first + s for s in sample], "parity_launches_replayed": launches,
                "parity_what": "every pair word, sub word, peak of the timed context's last launch + the status bytes, bit-exact vs the CPU oracle", "parity_s": time.perf_counter() - t0}

    def measure(contract, layout, inp, check=False, enabled_only=None):
        fma = (contract == "fma") and flavor == 1
        ctx = Dspi(flavor, S, device=dev.index, fma=fma)
        ctx.set_rate(FS); ctx.set_volume(w["vol"])
        assert ctx.load_bulk(w["blob"]) == 0
        if w.get("perstream"):
            for s in range(S):
                for req, wv, pl in per_stream_requests(first + s): ctx.vendor_set(req, wv, pl, stream=s)
        if inp not in pcm_cache:
            pcm_cache.clear()
            pcm_cache[inp] = synth_device(torch, dev, S, frames, FS, 1234 + rank, inp == "mix", first)
        pcm = pcm_cache[inp]
        tiled = layout == "tiled"
        R = ctx.tile_streams(); tiles = (S + R - 1) // R
        if tiled:      # [tile][output][frame][R] / [tile][frame][R]  (include/dspi.h, DSPI_OUT_TILED)
            pairs = torch.empty((tiles, 2 * P, frames, R), dtype=torch.int32, device=dev)
            sub = torch.empty((tiles, frames, R), dtype=torch.int32, device=dev)
        else:          # [stream][pair][frame][2] / [stream][frame]  (usb_audio.c:934-940)
            pairs = torch.empty((S, P, frames, 2), dtype=torch.int32, device=dev)
            sub = torch.empty((S, frames), dtype=torch.int32, device=dev)
        peaks = torch.empty((S, NB, 2 + N), dtype=torch.int16, device=dev)
        no_peaks = os.environ.get("DSPI_BENCH_NO_PEAKS") == "1"      # development: the per-packet peak array not requested (traffic accounting)
        torch.cuda.synchronize()
        smi = PowerSampler(dev.index) if check else None
        if smi: smi.start()
        elapsed, kernel_ms = timed_steps(args, torch, dist, backend, dev, ctx,
                                         lambda: ctx.process_device(pcm.data_ptr(), NB, B, 16, pairs.data_ptr(), sub.data_ptr(), 0 if no_peaks else peaks.data_ptr(), tiled=tiled,
                                                                    enabled_only=bool(w.get("enabled_only")) if enabled_only is None else enabled_only),
                                         min_load_s=0.6 if (smi and smi.ok) else 0.0)
        preload = timed_steps.preload_steps
        if smi: smi.stop()
        plan = ctx.launch_plan()
        # whole job: frames of all ranks / the slowest rank's time — sum and max over ranks, 2 x 8 bytes over RCCL
        from dspi_amd.shard import reduce_throughput, gather_ranks
        elapsed_local = elapsed
        _, elapsed, fps = reduce_throughput(dist, float(S) * frames * args.steps, elapsed, device=dev if backend == "nccl" else "cpu")
        m = dict(contract=contract if flavor == 1 else "integer", out_layout=layout, input=inp, frames_per_s=fps, value=fps * CH,
                 ms_per_step=elapsed / args.steps * 1e3, kernel_ms=kernel_ms, latency_layout=plan.get("latency_layout", 0) > 0)
        m["enabled_only"] = bool(w.get("enabled_only")) if enabled_only is None else enabled_only      # True: silent pairs and the sub are not zero-filled (fewer bytes than the firmware's own stores)
        if smi: m["power"] = smi.window(*timed_steps.window) if smi.ok else None
        m["preload_steps"] = preload
        # N = 1: 32 streams of the timed context against the oracle.  N > 1: EVERY rank checks 8 streams of its own shard (its own context, its own
        # input) and the verdicts travel to rank 0 with the per-rank times in one all_gather — a rank that computes wrong words fails the job
        # wherever it sits, and the line lists every rank's checked streams and step time.
        if check and not args.no_parity and (rank == 0 or world > 1):
            try:
                m["parity"] = parity_check(ctx, fma, pcm, pairs, sub, peaks, tiled, args.warmup + preload + args.steps, k=32 if world == 1 else 8)
            except SystemExit as e:
                if world == 1: raise
                m["parity"] = {"parity_checked": 0, "parity_streams": [], "parity_error": str(e)}
        if check and world > 1:
            par = m.get("parity") or {}
            ids = (list(par.get("parity_streams", [])) + [-1] * 8)[:8]
            rows = gather_ranks(dist, [elapsed_local / args.steps * 1e3, kernel_ms, float(par.get("parity_checked", 0)), 1.0 if par.get("parity_error") else 0.0] + ids,
                                device=dev if backend == "nccl" else "cpu")
            m["per_rank"] = [{"rank": r, "ms_per_step": v[0], "kernel_ms": v[1], "parity_checked": int(v[2]), "parity_streams": [int(x) for x in v[4:] if x >= 0]} for r, v in enumerate(rows)]
            if any(v[3] for v in rows):
                raise SystemExit("bench.py: PARITY FAILURE on rank(s) %s" % [r for r, v in enumerate(rows) if v[3]])
            if not args.no_parity:
                m["parity"] = dict(par, parity_checked=sum(int(v[2]) for v in rows), parity_streams=[int(x) for v in rows for x in v[4:] if x >= 0],
                                   parity_ranks=world)
        ctx.close()
        del pairs, sub, peaks
        torch.cuda.empty_cache()
        return m

    primary = measure(args.contract, args.out_layout, args.input, check=True)
    also = []
    if world == 1 and not args.no_variants:
        other_contract = [("canonical" if args.contract == "fma" else "fma")] if flavor == 1 else []
        other_layout = "stream" if args.out_layout == "tiled" else "tiled"
        for c, l, i in [(oc, args.out_layout, args.input) for oc in other_contract] + [(args.contract, other_layout, args.input)] + \
                       [(oc, other_layout, args.input) for oc in other_contract] + [(args.contract, args.out_layout, "noise" if args.input == "mix" else "mix")]:
            also.append(measure(c, l, i))
        if w.get("enabled_only"):      # the same workload with the firmware's own zero-fill of the silent pairs and the sub (usb_audio.c:930-933): what earlier rounds' BENCH files measured
            also.append(measure(args.contract, args.out_layout, args.input, enabled_only=False))
    if rank != 0:
        return None

    from dspi_amd.host import source_fingerprint
    SRC_SHA16 = source_fingerprint()
    alg = algorithmic_bytes(w, frames)
    kernel_key = {"3": "chain3", "2": "chain2", "2b": "chain2b", "5": "chain5", "perstream": "perstream", "perstream_eq": "perstream_eq"}[args.config]      # (2b: no counter profile of its own — its traffic fields stay null rather than borrow config 2's)

    def roof(m):
        per_launch_frames = S * frames
        gbs = lambda b: per_launch_frames * b / (m["kernel_ms"] * 1e-3) / 1e9
        ach = gbs(alg["exact"])
        prof = latest_profile(kernel_key, m["contract"], m["out_layout"])
        traffic = prof["hbm_bytes_per_frame"] * per_launch_frames if prof else None
        r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "frac_what": "STRICT: bytes a launch of this span must move with the lines' whole history kept (I/O + 4 B x first min(dly, T) frames read + 4 B x last min(line, T) frames written, per delayed output)",
             "algorithmic_bytes_per_frame": alg["exact"],
             "algorithmic_bytes_per_frame_hbm_resident": alg["resident"], "frac_hbm_resident": gbs(alg["resident"]) / HBM_PEAK_GBS,
             "algorithmic_bytes_per_frame_launch_span": alg["span"], "frac_launch_span": gbs(alg["span"]) / HBM_PEAK_GBS,
             "kernel_ms": m["kernel_ms"], "frames_per_launch": per_launch_frames,
             "traffic": traffic, "traffic_source": prof["source"] if prof else None,
             "traffic_what": "2 x FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 PMC profile of this kernel variant (counters cannot run inside the timed region; "
                             "taken on the builder's box, not this one), per launch" if prof else None,
             # that profile names the sources it was taken from — a different tree means the figure may describe an older kernel
             "traffic_stale": (prof.get("src_sha16") != SRC_SHA16) if prof else None,
             "traffic_ratio": (traffic / (alg["exact"] * per_launch_frames)) if prof else None,
             "hbm_fraction_measured_traffic": (gbs(prof["hbm_bytes_per_frame"]) / HBM_PEAK_GBS) if prof else None,
             "valu_fraction": (prof["valu_insts_per_frame"] * per_launch_frames / (m["kernel_ms"] * 1e-3) / VALU_PEAK_WAVE_INSTS) if prof and prof["valu_insts_per_frame"] else None,
             "binds": None}
        # evidence for what binds: socket power and shader clock sampled over the timed region (the device under this load for >= 0.6 s before
        # it), the instruction issue fraction at THAT clock (a SIMD issues one wave-instruction per 4 cycles), the bytes moved against 8 TB/s.
        # The label follows the clock, not a power threshold: below 0.9 x the top of the part's clock table the socket's sustained limit is
        # what sets the pace (time follows energy per frame: profiles/r04_power_model.md).
        pw = m.get("power")
        if pw:
            r.update(power_w=pw["power_w"], power_w_max=pw["power_w_max"], power_cap_w=pw["power_cap_w"], sclk_mhz=pw["sclk_mhz"], sclk_max_mhz=pw["sclk_max_mhz"], power_samples=pw["power_samples"])
            if pw["sclk_mhz"] and prof and prof["valu_insts_per_frame"]:
                r["valu_fraction_at_sclk"] = prof["valu_insts_per_frame"] * per_launch_frames / (m["kernel_ms"] * 1e-3) / (1024 * pw["sclk_mhz"] * 1e6 / 4.0)
        else:
            r.update(power_w=None, sclk_mhz=None, sclk_max_mhz=None)
        # The energy ceiling: every full-size run of this chain sits at the socket's power limit, so the roofline of this path
        # on this part is joules.  energy_j = mean socket power over the timed region x the kernel's time; energy_floor_j = what the launch cannot
        # avoid spending by the builder's model: the chain's arithmetic alone (the "arithmetic only" ablation: no delays, leveller off, no word
        # buffers, J per frame as measured, static power included for its own duration) + the STRICT bytes at the measured joules per HBM byte.
        pm = latest_power_model() if kernel_key in ("chain3", "perstream", "perstream_eq") else None
        if pw and pw.get("power_w"):
            r["energy_j"] = pw["power_w"] * m["kernel_ms"] * 1e-3
            if pm:
                r["energy_floor_j"] = per_launch_frames * (pm["arith_j_per_frame"] + alg["exact"] * pm["hbm_j_per_byte"])
                r["frac_energy"] = r["energy_floor_j"] / r["energy_j"]
                r["energy_model"] = {"source": pm["source"], "arith_j_per_frame": pm["arith_j_per_frame"], "hbm_j_per_byte": pm["hbm_j_per_byte"], "idle_w": pm.get("idle_w"),
                                     "stale": pm.get("src_sha16") != SRC_SHA16,
                                     "what": "floor = frames x (arithmetic-only J/frame + strict bytes/frame x J/byte); tools/ablate_power.py on the builder's box"}
            else:
                r["energy_floor_j"] = r["frac_energy"] = None
        else:
            r["energy_j"] = r["energy_floor_j"] = r["frac_energy"] = None
        issue = r.get("valu_fraction_at_sclk") or r["valu_fraction"]
        mem = r["hbm_fraction_measured_traffic"]
        if pw and pw["sclk_mhz"] and pw["sclk_max_mhz"] and pw["sclk_mhz"] < 0.9 * pw["sclk_max_mhz"]:
            r["binds"] = "power (shader clock %.0f MHz < 0.9 x %.0f MHz under this load)" % (pw["sclk_mhz"], pw["sclk_max_mhz"])
        elif issue and (not mem or issue >= mem): r["binds"] = "valu issue"
        elif mem: r["binds"] = "memory (bytes moved at the L2-fabric boundary)"
        else: r["binds"] = "unknown (no counter profile of this variant, no clock reading)"
        return r

    if flavor == 1:
        kname = "chain_kernel_pk<false, true, false, %s, %s, %s, %s>" % ("true" if args.out_layout == "tiled" else "false", "true" if args.contract == "fma" else "false",
                                                                         "true" if w.get("perstream") else "false", "true" if w.get("perstream") == "eq" else "false")
        if CH == 2: kname = kname.replace("<false, true", "<false, false")
        if primary.get("latency_layout"):      # small launches: the skewed cascade (dspi_chain_skew.inc: no output EQ / output rows; dspi_chain_skew_lev.inc: leveller on)
            fma_s = "true" if args.contract == "fma" else "false"
            lev_on = bool(w["blob"]["leveller"]["enabled"])
            out_eq = CH != 2
            kname = ("chain_kernel_skew_lev<%s, false, false>" % fma_s) if lev_on else ("chain_kernel_skew<%s, false, %s, false>" % (fma_s, "true" if out_eq else "false"))
    else:
        # wave layout by launch size (dspi_kernels.hip chain_kernel NW): seven waves up to one 64-stream workgroup per CU, four beyond
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        forced = os.environ.get("DSPI_Q28_WAVES")
        nw = int(forced) if forced in ("4", "7") else (7 if (S + 63) // 64 <= cus else 4)
        kname = "chain_kernel<0, false, false, false, %d>" % nw
        lay = os.environ.get("DSPI_Q28_LAYOUT")      # contexts of up to eight streams per CU take the Q28 latency layout (dspi_kernels.hip q28_latency_limit)
        if lay == "lat" or (lay != "chain" and forced not in ("4", "7") and S <= 8 * cus): kname = "chain_kernel_q28_lat<false>"
    roofline = roof(primary)
    roofline["kernel"] = kname
    for m in also:
        r = roof(m)
        m["roofline_frac"], m["roofline_frac_hbm_resident"], m["valu_fraction"] = r["frac"], r["frac_hbm_resident"], r["valu_fraction"]
    out = {
        "metric": "audio samples/s (whole node), 96 kHz 11-ch 10-band PEQ; % HBM roofline",
        "value": primary["value"], "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": primary["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32" if flavor else "int32 (Q28)", "data": "synthetic",
        "config": {"workload": w["text"], "float_contract": ("fma: the firmware as built (GCC -ffp-contract=fast on Cortex-M33), DSPI_FLOAT_CONTRACT_FMA"
                                                             if args.contract == "fma" else "canonical: no contraction") if flavor else "n/a (integer)",
                   "out_layout": "tiled [tile][output][frame][row] (DSPI_OUT_TILED)" if args.out_layout == "tiled" else "stream-major [stream][pair][frame][2] (usb_audio.c:934-940)",
                   "input": "SURVEY 8d synthetic mix (70% noise, 10% sweep, 10% bursts, 5% silence, 5% square)" if args.input == "mix" else "white noise -6 dBFS",
                   "streams_per_gpu": S, "streams_total": total if args.scaling == "strong" else total * world, "channels": CH,
                   "blocks_per_step": NB, "frames_per_step_per_stream": frames, "preload_steps": primary.get("preload_steps", 0),
                   "frames_per_s": primary["frames_per_s"], "realtime_streams": primary["frames_per_s"] / FS, "parallelism": f"streams sharded x{world}"},
        "roofline": roofline,
    }
    if primary.get("parity"): out.update(primary["parity"])
    else: out["parity_checked"] = 0
    if primary.get("enabled_only"): out["config"]["enabled_only"] = "DSPI_OUT_ENABLED_ONLY: silent pairs and the sub are left unwritten (the firmware zero-fills them, usb_audio.c:930-933)"
    # two more entries of `also`, each a child run of this file on the same box: 200 packets per launch on tiled words (the history store and the
    # launch's head amortise: the launch-span end of the kernel), and the PREVIOUS round's library with the driver's own arguments — the headline's
    # box-to-box spread (+-5 %) is larger than a round's gain, so only a same-box pair says what the round did (tools/ab_bench.sh, in the line)
    if world == 1 and args.config == "3" and not args.no_side_runs and not args.no_variants and not args.streams and not args.blocks_per_step:
        r = side_run(["--out-layout", "tiled", "--blocks-per-step", "200", "--steps", "10", "--warmup", "3", "--contract", args.contract])
        also.append({"contract": args.contract, "out_layout": "tiled", "input": "mix", "blocks_per_step": 200, "child_run": True, "error": r.get("error"),
                     "ms_per_step": r.get("ms_per_step"), "ms_per_50_packets": (r["ms_per_step"] / 4.0) if r.get("ms_per_step") else None, "value": r.get("value"),
                     "kernel_ms": (r.get("roofline") or {}).get("kernel_ms"), "roofline_frac": (r.get("roofline") or {}).get("frac"),
                     "roofline_frac_hbm_resident": (r.get("roofline") or {}).get("frac_hbm_resident"), "parity_checked": r.get("parity_checked")})
        prev = sorted(glob.glob(os.path.join(ROOT, "dspi_amd", "csrc", "libr0[0-9].so")))
        if prev:
            ab = []
            for tag, envv in (("this", None), ("prev", {"DSPI_LIB": prev[-1]}), ("this", None), ("prev", {"DSPI_LIB": prev[-1]})):      # A B A B, minutes apart on one box
                r = side_run(["--steps", str(args.steps), "--warmup", str(args.warmup), "--contract", args.contract, "--out-layout", args.out_layout, "--no-parity"], env=envv)
                ab.append({"library": os.path.basename(prev[-1]) if tag == "prev" else "libdspi_mi355x.so", "ms_per_step": r.get("ms_per_step"),
                           "kernel_ms": (r.get("roofline") or {}).get("kernel_ms"), "error": r.get("error")})
            ok = [x for x in ab if x["kernel_ms"]]
            mean = lambda lib: (lambda v: sum(v) / len(v) if v else None)([x["kernel_ms"] for x in ok if x["library"] == lib])
            a_ms, b_ms = mean("libdspi_mi355x.so"), mean(os.path.basename(prev[-1]))
            out["same_box_ab"] = {"what": "this round's library and the previous round's (rebuilt from its last commit by __graft_entry__.build()), alternating child runs of bench.py on THIS box",
                                  "runs": ab, "kernel_ms_this": a_ms, "kernel_ms_prev": b_ms, "ratio_this_over_prev": (a_ms / b_ms) if a_ms and b_ms else None}
        else:
            out["same_box_ab"] = {"what": "previous round's library not present (dspi_amd/csrc/libr0N.so: built by __graft_entry__.build() from git history)", "runs": []}
    if also:
        out["also"] = also
    # The drop-in call as the firmware makes it (usb_audio.c:1326-1332: ONE packet per call, host buffers), beside the batched figure above:
    # dspi_host -rt (plain C over include/dspi.h) on one stream of this preset, every word of every call checked against the oracle
    # (tools/bench_realtime.py; profiles/<round>_realtime.json holds the full table).  Reported, never part of `value`.
    if world == 1 and flavor == 1 and args.config == "3" and not args.no_realtime:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_realtime
            from dspi_amd.host import Dspi
            d1 = Dspi(W.F32_FMA if args.contract == "fma" else 1, 1, device=dev.index)
            d1.set_rate(FS); assert d1.load_bulk(w["blob"]) == 0
            bulk = d1.collect_bulk(); d1.close()
            r = bench_realtime.run("f32fma" if args.contract == "fma" else "f32", W.F32_FMA if args.contract == "fma" else 1, 1, FS, B, RT_CALLS, 1000, check=True, bulk=bulk)
            out["realtime_call"] = rt_record("one packet per dspi_process(), host buffers, one stream of this preset (dspi_host -rt)", r)
        except (OSError, subprocess.SubprocessError) as e:  # a missing dspi_host binary must not cost the line; a parity failure (SystemExit / AssertionError) fails the run
            out["realtime_call"] = {"what": "one packet per dspi_process()", "p50_us": None, "error": str(e)[:200]}
        # ... and the RP2040 Q28 flavour's: one 48-frame packet at 48 kHz, one stream of BASELINE config 5's preset (the integer chain's latency
        # layout, dspi_chain_q28_lat.inc), every word of every call checked against the oracle
        try:
            from dspi_amd import workloads as WL
            r = bench_realtime.run("q28", 0, 1, 48000, 48, RT_CALLS, 1000, check=True)
            out["realtime_call_q28"] = rt_record("one packet per dspi_process(), host buffers, one stream of BASELINE config 5's preset (dspi_host -rt -f q28)", r)
        except (OSError, subprocess.SubprocessError) as e:
            out["realtime_call_q28"] = {"what": "one packet per dspi_process(), Q28", "p50_us": None, "error": str(e)[:200]}
    if primary.get("per_rank"): out["per_rank"] = primary["per_rank"]
    # (N > 1 too: north_star wants the reference's C path on the host cores NEXT TO the 2 / 4 / 8-GPU figures; rank 0 times it after the GPU
    #  work of every rank has ended — the other ranks idle at the process group's teardown meanwhile)
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(flavor, FS, B, w["blob"], CH, args.contract == "fma" and flavor == 1, w["vol"], f"config {args.config}")
        except Exception as e:  # the GPU number stands on its own
            out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    return out


def bench_consumer(args, torch, dev, rank, world, dist, backend):
    """the PDM sigma-delta modulator and the S/PDIF subframe encoder on the bench shape."""
    from dspi_amd.host import Dspi
    S = args.streams or 65536
    F = 2400
    ctx = Dspi(1, S, device=dev.index)
    ctx.set_rate(96000)
    R = ctx.tile_streams(); nt = (S + R - 1) // R
    tiled = args.out_layout == "tiled"
    if args.config == "pdm":
        sub = torch.randint(-(1 << 27), 1 << 27, (nt, F, R) if tiled else (S, F), dtype=torch.int32, device=dev)
        words = torch.empty((nt * R if tiled else S) * F * 8, dtype=torch.int32, device=dev)
        step = lambda: ctx.pdm_device(sub.data_ptr(), F, words.data_ptr(), tiled=tiled)
        per_unit_bytes, unit_name, units = 36.0, "sub samples", S * F
        text = "SURVEY 8f-2: PDM sigma-delta modulator (256x oversampled, 2nd order, noise-shaped dither), 65 536 streams x 2 400 Q28 samples per call"
        kernel, bound = "pdm_kernel", "valu (integer)"
    elif args.config == "i2s":
        n_in = (nt * R if tiled else S) * 4 * F * 2
        pairs = torch.randint(-(1 << 23), 1 << 23, (n_in,), dtype=torch.int32, device=dev)
        outb = torch.empty(n_in, dtype=torch.int32, device=dev)
        step = lambda: ctx.i2s_device(pairs.data_ptr(), F, 0xF, outb.data_ptr(), tiled=tiled)
        per_unit_bytes, unit_name, units = 64.0, "stream-frames (4 pairs)", S * F
        text = "SURVEY 8f-3: I2S slot words (left-justified 24-bit in 32-bit slots), 65 536 streams x 4 pairs x 2 400 frames per call"
        kernel, bound = "i2s_kernel", "hbm"
    else:
        n_in = (nt * R if tiled else S) * 4 * F * 2
        pairs = torch.randint(-(1 << 23), 1 << 23, (n_in,), dtype=torch.int32, device=dev)
        outb = torch.empty(n_in * 2, dtype=torch.int32, device=dev)
        step = lambda: ctx.spdif_device(pairs.data_ptr(), F, 0, outb.data_ptr(), tiled=tiled)
        per_unit_bytes, unit_name, units = 96.0, "stream-frames (4 pairs)", S * F
        text = "SURVEY 8f-3: IEC-60958 subframe encoder, 65 536 streams x 4 pairs x 2 400 frames per call"
        kernel, bound = "spdif_kernel", "hbm"
    elapsed, kernel_ms = timed_steps(args, torch, dist, backend, dev, ctx, step)
    ctx.close()
    if rank != 0:
        return None
    ups = float(world) * units * args.steps / elapsed
    ach = units * per_unit_bytes / (kernel_ms * 1e-3) / 1e9
    return {"metric": f"{unit_name}/s", "value": ups, "unit": f"{unit_name}/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": text, "out_layout": args.out_layout, "streams_per_gpu": S},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel, "kernel_ms": kernel_ms, "algorithmic_bytes_per_unit": per_unit_bytes, "binds": bound}}


if __name__ == "__main__":
    main()
