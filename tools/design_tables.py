#!/usr/bin/env python3
"""tools/design_tables.py [round tag] — rebuild the measurement tables of DESIGN.md section 6.0 from the committed profiles, so that the
prose never carries numbers the profiles do not: rows between `<!-- table:NAME -->` and `<!-- /table -->` are replaced in place.

    headline   profiles/<tag>[a-p]_summary.md (rocprofv3 trace + PMC passes, tools/prof.sh) and profiles/bench_<tag>_*.json
    realtime   profiles/<tag>_realtime.json (tools/bench_realtime.py)
"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(ROOT, "profiles")


def summary(letter):
    path = os.path.join(P, f"{TAG}{letter}_summary.md")
    if not os.path.exists(path): return None
    s = open(path).read()
    def grab(rx, cast=float):
        m = re.search(rx, s)
        return cast(m.group(1)) if m else None
    return {"us": grab(r"- at (\d+) us/launch"), "moved": grab(r"GB/launch = ([\d.]+) B/frame \(algorithmic"),
            "algo": grab(r"\(algorithmic: ([\d.]+) B/frame"), "issue": grab(r"cycles\) = ([\d.]+)"),
            "algo_tbs": grab(r"moved, ([\d.]+) TB/s algorithmic"), "insts": grab(r"streams per lane\): ([\d.]+)")}


def bench(name):
    path = os.path.join(P, f"bench_{TAG}_{name}.json")
    if not os.path.exists(path): return None
    for line in open(path):
        if line.startswith("{"): return json.loads(line)
    return None


def ms(x): return "—" if x is None else f"{x / 1000:.2f}"


def headline():
    rows = [("3, FMA, stream-major (**default**)", "b", "`chain_kernel_pk` (the delay line as the hand-over)"),
            ("3, FMA, tiled", "a", "`chain_kernel_pk` (words written ahead)"),
            ("3, canonical, tiled", "c", ""), ("3, canonical, stream-major", "d", ""),
            ("**2** (`--config 2`, `DSPI_OUT_ENABLED_ONLY`)", "l", "`chain_kernel_skew` (section 4.3)"),
            ("2b (`--config 2b`: every band a biquad)", "r", ""),
            ("2 forced onto the packed kernel", "m", "`chain_kernel_pk`"),
            ("3's preset on 512 streams", "p", "`chain_kernel_skew_lev`"),
            ("5, Q28, 16 384 streams", "e", "`chain_kernel<0,…,7>`"), ("5 at 65 536 streams", "k", "`chain_kernel<0,…,4>`"),
            ("5's preset on 1 024 streams", "q", "`chain_kernel_q28_lat` (section 4.4)"),
            ("65 536 presets, identical filters, stream-major", "f", "`chain_kernel_pk<…,PV>`"),
            ("65 536 presets, one master band each, tiled", "i", "`chain_kernel_pk<…,PV,PVB>`"),
            ("65 536 presets, one master band each, stream-major", "o", ""),
            ("65 536 presets, EVERY band differs, tiled", "n", ""),
            ("I2S slot words", "j", "`i2s_kernel`"), ("PDM modulator", "g", "`pdm_kernel`"), ("S/PDIF subframes", "h", "`spdif_kernel`")]
    out = ["| config | kernel | ms / launch (rocprofv3) | algorithmic B/frame (HBM-resident rule, SURVEY 8d) -> frac of 8 TB/s | moved B/frame (2 x FETCH + WRITE) | VALU issue | VALU wave-instructions per stream-frame x 64 |",
           "|---|---|---|---|---|---|---|"]
    for label, letter, kern in rows:
        s = summary(letter)
        if not s: continue
        frac = f"{s['algo']:.0f} -> {s['algo_tbs'] / 8.0:.3f}" if s["algo"] and s["algo_tbs"] is not None else "—"
        out.append(f"| {label} | {kern} | {ms(s['us'])} | {frac} | {s['moved']} | {s['issue']} | {s['insts']} |")
    d = bench("default")
    extra = []
    if d:
        r = d["roofline"]
        extra.append(f"- `bench.py` (no flags, `profiles/bench_{TAG}_default.json`): **{d['ms_per_step']:.2f} ms** per launch = {d['config']['frames_per_s']:.3e} frames/s = "
                     f"**{d['value']:.3e} samples/s**; `roofline.frac` **{r['frac']:.3f}** (STRICT: {r['algorithmic_bytes_per_frame']:.1f} B/frame, {r['achieved']:.0f} GB/s), "
                     f"`frac_hbm_resident` {r['frac_hbm_resident']:.3f} (104 B), `frac_launch_span` {r['frac_launch_span']:.3f} (59.4 B); {r['power_w']:.0f} W of "
                     f"{r['power_cap_w']:.0f} W at {r['sclk_mhz']:.0f} of {r.get('sclk_max_mhz') or 0:.0f} MHz -> binds: {r['binds']}; {d.get('parity_checked', 0)} streams of the timed context checked against the oracle.")
        for a in d.get("also", []):
            extra.append(f"- also: {a['contract']}, {a['out_layout']}, {a['input']}: {a['ms_per_step']:.2f} ms, frac {a['roofline_frac']:.3f} (HBM-resident rule: {a.get('roofline_frac_hbm_resident', 0):.3f})")
        for name, c in (d.get("configs") or {}).items():
            if "error" in c: extra.append(f"- config {name} (same line): {c['error']}"); continue
            rr = c["roofline"]
            extra.append(f"- config {name} in the same line ({c['streams']} streams x {c['blocks_per_step']} packets, {c['steps']} timed launches after {c['warmup']} + {c['preload_steps']}): "
                         f"**{c['ms_per_step']:.2f} ms** per launch, {c['frames_per_s']:.3e} frames/s, {c['realtime_streams']:.0f} real-time streams, frac {rr['frac']:.3f}, "
                         f"VALU issue at the clock {rr.get('valu_fraction_at_sclk') or 0:.2f}, {rr.get('sclk_mhz') or 0:.0f} MHz, binds: {rr['binds']}, `{rr['kernel']}`, {c['parity_checked']} streams checked")
        for key in ("realtime_call", "realtime_call_q28"):
            rt = d.get(key)
            if rt and rt.get("p50_us"): extra.append(f"- `{key}`: p50 {rt['p50_us']:.1f} us, p99 {rt['p99_us']:.1f} us, max {rt['max_us']:.0f} us over {rt['calls']} calls; {rt.get('parity')}")
        cb = d.get("cpu_baseline")
        if cb and cb.get("value"): extra.append(f"- `cpu_baseline`: {cb['value']:.3e} samples/s on {cb['cores']} host threads ({cb['kind']}), single core {cb.get('single_core_realtime_x', 0):.1f} x real time")
    for name, what in (("steps200", "200 timed launches"), ("blocks200_tiled", "200 packets per launch, tiled words")):
        b = bench(name)
        if b: extra.append(f"- {what} (`bench_{TAG}_{name}.json`): {b['ms_per_step']:.2f} ms per launch of {b['config']['blocks_per_step']} packets, frac {b['roofline']['frac']:.3f}")
    return "\n".join(out) + "\n\n" + "\n".join(extra)


def realtime():
    path = os.path.join(P, f"{TAG}_realtime.json")
    if not os.path.exists(path): return "(not taken)"
    rows = json.load(open(path))["runs"]
    out = ["| context | calls | p50 us | p99 us | max us | bit-exact |", "|---|---|---|---|---|---|"]
    for r in rows:
        out.append(f"| {r['flavor']}, {r.get('preset', 'config3')}, {r['streams']} stream(s), {r['block_len']} frames at {r['fs']} Hz | {r['calls']} | {r['p50_us']:.1f} | {r['p99_us']:.1f} | "
                   f"{r['max_us']:.0f} | {r['parity']} |")
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for name, fn in (("headline", headline), ("realtime", realtime)):
        rx = re.compile(rf"(<!-- table:{name} -->\n).*?(\n<!-- /table -->)", re.S)
        if not rx.search(s):
            print(f"DESIGN.md has no table:{name} markers", file=sys.stderr); continue
        s = rx.sub(lambda m: m.group(1) + fn() + m.group(2), s)
    open(path, "w").write(s)


if __name__ == "__main__":
    main()
