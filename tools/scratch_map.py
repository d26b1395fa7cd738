#!/usr/bin/env python3
"""Where a packed chain kernel's scratch (spill) instructions sit: per time loop of each wave role, how many scratch loads / stores one
iteration (= one 16-frame step) contains.  A spill outside the time loops costs nothing; one inside costs a 256- or 512-byte row of
memory traffic per wave and step (per-lane filters).
    python tools/scratch_map.py <DSPI_PART> '<template args as hipcc prints them, e.g. false, true, false, true, true, true, true>'
Compiles dspi_amd/csrc/dspi_kernels.hip for that part to assembly (/tmp/scratch_map_p<part>.s; reused when newer than the sources)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
part, targs = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "all")      # "all": one line per instance of the part
src = os.path.join(ROOT, "dspi_amd", "csrc")
out = f"/tmp/scratch_map_p{part}.s"
newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".hip", ".inc", ".h")))
if not os.path.exists(out) or os.path.getmtime(out) < newest:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fgpu-flush-denormals-to-zero",
                           "-fno-slp-vectorize", "-fPIC", f"-DDSPI_PART={part}", "--cuda-device-only", "-S", os.path.join(src, "dspi_kernels.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
T = open(out).read().split("\n")
if targs == "all":
    import itertools
    names = sorted({l.split(":")[0] for l in T if l.startswith("_Z") and "chain_kernel_pkI" in l and ":" in l})
    for nm in names:
        bits = re.search(r"chain_kernel_pkI((?:Lb[01]E)+)", nm).group(1)
        ta = ", ".join("true" if b == "1" else "false" for b in re.findall(r"Lb([01])E", bits))
        r = subprocess.run([sys.executable, __file__, part, ta], capture_output=True, text=True).stdout.strip().split("\n")
        tot = sum(int(m.group(1)) + int(m.group(2)) for l in r[1:] for m in [re.search(r": (\d+) scratch loads \+ (\d+) stores", l)] if m)
        print(f"<{ta}>  in-loop scratch instructions per step, all roles: {tot}   " + "; ".join(re.sub(r"time loop lines \S+ \(", "(", l.strip()).split(" per step")[0] for l in r[1:] if not " 0 scratch loads + 0 stores" in l))
    sys.exit(0)
mangled = "chain_kernel_pkI" + "".join("Lb%dE" % (a.strip() == "true") for a in targs.split(","))
starts = [i for i, l in enumerate(T) if l.startswith("_Z") and mangled in l.split(":")[0] and ":" in l]
if not starts: sys.exit("no kernel " + mangled)
a = starts[0]
b = next(i for i in range(a, len(T)) if T[i].startswith("\t.size") and mangled in T[i])
L = T[a:b]
lab = {m.group(1): i for i, l in enumerate(L) for m in [re.match(r"^(\.L\w+):", l)] if m}
loops = sorted({(lab[t], i) for i, l in enumerate(L) for m in [re.search(r"s_c?branch\w* (\.L\w+)", l)] if m for t in [m.group(1)] if t in lab and lab[t] < i})
outer = [p for p in loops if p[1] - p[0] > 200 and not any(q[0] <= p[0] and p[1] <= q[1] and q != p for q in loops)]
def role(x, y):
    body = "\n".join(L[x:y])
    if ".Lsplit" in body: return "intake (left / both channels)"
    if "buffer_load_dwordx4" in body and "v_pk_fma" not in body: return "copy wave"
    if body.count("v_cvt_i32_f32") >= 16: return "output waves"
    if "v_rcp_f32" in body or "v_div_scale" in body: return "hand-off (pass 2, crossfeed)"
    return "third wave (right channel / copy)"
print(f"{mangled}: {len(L)} lines, {sum('scratch_' in l for l in L)} scratch instructions in all")
for x, y in outer:
    ld = [i for i in range(x, y) if "scratch_load" in L[i]]; st = [i for i in range(x, y) if "scratch_store" in L[i]]
    inner = [(p, q) for p, q in loops if x < p and q < y]
    deep = [i for i in ld + st if any(p <= i <= q for p, q in inner)]
    print(f"  time loop lines {x}-{y} ({role(x, y)}): {len(ld)} scratch loads + {len(st)} stores per step, {len(deep)} of them inside band loops")
