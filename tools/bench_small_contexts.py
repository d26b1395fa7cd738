#!/usr/bin/env python3
"""Small contexts: BASELINE config 3's preset (LEVELLER=0: with the leveller off, the class of the latency layout's second shape; default:
the third shape, dspi_chain_skew_lev.inc) at stream counts from 2 up, 96 kHz, 96-frame packets, 200 packets per launch, device buffers —
the latency layout (the library's own choice, or DSPI_F32_LAYOUT=skew beyond its size rule) against the packed kernel
(DSPI_F32_LAYOUT=packed).  PERSTREAM=1: every stream its own preset.  One JSON line per (streams, layout)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from dspi_amd import wire as W, workloads as WL
    from dspi_amd.host import Dspi
    S = int(sys.argv[2]); B, blocks, fs = 96, 200, 96000
    FR = B * blocks
    dev = torch.device("cuda", 0)
    blob = WL.full_chain_blob(1)
    if os.environ.get("LEVELLER", "1") == "0": blob["leveller"]["enabled"] = 0
    d = Dspi(W.F32_FMA, S, device=0); d.set_rate(fs); d.set_volume(-20 * 256); assert d.load_bulk(blob) == 0
    if os.environ.get("PERSTREAM"):      # every stream its own preset: a preamp and one master band of its own (the per-lane-filter class)
        import struct
        p = blob["eq"][0][1]
        for s_ in range(S):
            d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.001 * s_), stream=s_)
            d.vendor_set(W.REQ["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", 0, 1, int(p["type"]), 0, float(p["freq"]), float(p["q"]), 1.0 + 0.0001 * s_), stream=s_)
    pcm = torch.randint(-16384, 16385, (S, FR, 2), dtype=torch.int16, device=dev)
    pairs = torch.empty((S, 4, FR, 2), dtype=torch.int32, device=dev); sub = torch.empty((S, FR), dtype=torch.int32, device=dev)
    peaks = torch.empty((S, blocks, 11), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr()); d.sync()
    t0 = time.perf_counter()
    for _ in range(3): d.process_device(pcm.data_ptr(), blocks, B, 16, pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr())
    d.sync(); dt = (time.perf_counter() - t0) / 3
    plan = d.launch_plan()
    print(json.dumps({"streams": S, "layout": "latency" if plan["latency_layout"] else "packed", "forced": os.environ.get("DSPI_F32_LAYOUT", ""), "per_stream_presets": bool(os.environ.get("PERSTREAM")),
                      "workgroups": plan["latency_layout"], "workgroups_with_paired_presets": plan["latency_layout_paired"], "ms_per_launch": dt * 1e3,
                      "frames_per_s": S * FR / dt, "realtime_x_per_stream": FR / fs / dt}))
    sys.exit(0)
SIZES = [int(x) for x in os.environ.get("SIZES", "1,2,16,128,512,1024,2048,4096,8192").split(",")]
# PERSTREAM=1 adds a run with DSPI_SKEW_PAIRED=0: one workgroup per preset, the form before the slots of a workgroup read their own images
for S in SIZES:
    for lay in ("", "skew", "packed") + (("unpaired",) if os.environ.get("PERSTREAM") else ()):
        env = dict(os.environ)
        if lay == "unpaired": env["DSPI_F32_LAYOUT"] = "skew"; env["DSPI_SKEW_PAIRED"] = "0"
        elif lay: env["DSPI_F32_LAYOUT"] = lay
        out = subprocess.run([sys.executable, __file__, "child", str(S)], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        print(out, flush=True)
