#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/prof.sh into profiles/<tag>_summary.md.

usage: python tools/prof_summary.py gpurun_out/prof_<tag> profiles/<name>.md ["note"]
"""
import glob, json, os, sqlite3, sys

src, dst = sys.argv[1], sys.argv[2]
PACKETS = int(os.environ.get("PACKETS_PER_LAUNCH", 50))      # bench.py --blocks-per-step default
STREAMS = int(os.environ.get("STREAMS", 65536)); BLOCK = int(os.environ.get("BLOCK_LEN", 96))
KERNEL_LIKE = os.environ.get("KERNEL_LIKE", "%chain_kernel%")   # which kernel of the run the counters are summed for
ALGO_BYTES = float(os.environ.get("ALGO_BYTES", 104))
note = sys.argv[3] if len(sys.argv) > 3 else ""
lines = [f"# rocprofv3 summary: {os.path.basename(src)}", "", note, "",
         f"Commands profiled: `python bench.py --steps 20 --warmup 3 --no-cpu-baseline` (kernel trace) and `--steps 3 --warmup 1` (each PMC pass); {os.environ.get('BENCH_ARGS', 'config 3')}: {STREAMS} streams x {PACKETS} packets x {BLOCK} frames per launch = {STREAMS * PACKETS * BLOCK:,} frames/launch)".replace(",", " "), ""]
tr = os.path.join(src, "trace", "trace_results.db")
kernel_avg_us = None
if os.path.exists(tr):
    db = sqlite3.connect(tr)
    lines += ["## --kernel-trace --stats (top kernels)", "", "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name if len(name) < 100 else name[:60] + "..." + name[-30:]
        lines.append(f"| `{short}` | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f} |")
        if KERNEL_LIKE.strip("%") in name:
            kernel_avg_us = avg
    q = "select name, vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x from kernels where name like '" + KERNEL_LIKE + "' limit 1"
    try:
        for r in db.execute(q):
            lines += ["", f"chain kernel resources: VGPR {r[1]} as rocprofv3 reports it (= half the allocation: hipcc's resource remark, tools/regs.sh, says {2 * r[1]}), SGPR {r[2]}, LDS {r[3]} B/workgroup, scratch {r[4]} B/lane, workgroup {r[5]}, grid {r[6]}"]
    except Exception as e:
        lines.append(f"(resource query failed: {e})")
counters = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_results.db"))):
    db = sqlite3.connect(d)
    try:
        rows = db.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '" + KERNEL_LIKE + "' group by counter_name").fetchall()
    except Exception as e:
        rows = []
    for n, v, c in rows:
        counters[n] = (v, c, os.path.basename(os.path.dirname(d)))
if counters:
    lines += ["", "## PMC counters (per chain_kernel launch, averaged over dispatches; each pass is a separate run)", "", "| counter | value / launch | pass |", "|---|---|---|"]
    for n in sorted(counters):
        v, c, p = counters[n]
        lines.append(f"| {n} | {v:.6g} | {p} |")
    frames = float(STREAMS) * PACKETS * BLOCK
    d = {k: v[0] for k, v in counters.items()}
    lines += ["", "## Derived", ""]
    if "SQ_INSTS_VALU" in d:
        lines.append(f"- VALU wave-instructions per stream-frame (x64 lanes; the packed kernel carries two streams per lane): {d['SQ_INSTS_VALU'] * 64 / frames:.1f}")
    if "SQ_INSTS_VALU" in d and "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        lines.append(f"- kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs): {cyc:.4g}; VALU issue utilisation = INSTS_VALU x 4.1 cycles (tools/probe/probe3: one wave-instruction per ~4.1 cycles per SIMD) / (1024 SIMDs x cycles) = {d['SQ_INSTS_VALU'] * 4.1 / (1024 * cyc):.3f}")
    if "SQ_WAVE_CYCLES" in d:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in d:
                lines.append(f"- {k} / SQ_WAVE_CYCLES = {d[k] / d['SQ_WAVE_CYCLES']:.3f}")
    hb = {}
    if "FETCH_SIZE" in d:
        hb["fetch_bytes_raw"] = d["FETCH_SIZE"] * 1024
        lines.append(f"- FETCH_SIZE = {d['FETCH_SIZE'] * 1024 / 1e9:.3f} GB/launch raw (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream; x2 = {d['FETCH_SIZE'] * 2048 / 1e9:.3f} GB)")
    if "WRITE_SIZE" in d:
        hb["write_bytes"] = d["WRITE_SIZE"] * 1024
        lines.append(f"- WRITE_SIZE = {d['WRITE_SIZE'] * 1024 / 1e9:.3f} GB/launch")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        tot = d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024
        lines.append(f"- HBM traffic (2 x FETCH + WRITE) = {tot / 1e9:.3f} GB/launch = {tot / frames:.1f} B/frame (algorithmic: {ALGO_BYTES:g} B/frame = {ALGO_BYTES * frames / 1e9:.3f} GB/launch)")
        if kernel_avg_us:
            lines.append(f"- at {kernel_avg_us:.0f} us/launch: {tot / kernel_avg_us / 1e6:.3f} TB/s moved, {ALGO_BYTES * frames / kernel_avg_us / 1e6:.3f} TB/s algorithmic")
        try:      # which sources the profiled library was built from (bench.py flags a profile of another tree as stale)
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from dspi_amd.host import source_fingerprint, LIB_PATH
            import hashlib
            hb["src_sha16"] = source_fingerprint()
            hb["lib_sha16"] = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16]
        except Exception as e:
            hb["src_sha16"] = None; hb["fingerprint_error"] = str(e)
        hb["hbm_bytes_per_launch"] = tot
        hb["frames_per_launch"] = frames
        hb["out_layout"] = os.environ.get("OUT_LAYOUT", "tiled")
        hb["contract"] = os.environ.get("CONTRACT", "fma")            # bench.py's keys for picking the profile of a variant
        hb["kernel_key"] = os.environ.get("KERNEL_KEY", "chain3")
        if kernel_avg_us: hb["kernel_avg_us"] = kernel_avg_us
        if "SQ_INSTS_VALU" in d: hb["valu_insts_per_launch"] = d["SQ_INSTS_VALU"]
        json.dump(hb, open(os.path.join(os.path.dirname(dst), "traffic_" + os.path.basename(dst).split("_")[0] + ".json"), "w"))
os.makedirs(os.path.dirname(dst), exist_ok=True)
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
