#!/bin/bash
# tools/regs.sh [part] [filter] — VGPR / scratch / LDS of the kernels of one DSPI_PART of dspi_kernels.hip (default 4: the packed float
# kernels of the firmware contract, shared preset), from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
PART=${1:-4}; FILTER=${2:-.}
cd "$(dirname "$0")/../dspi_amd/csrc"
[ -n "$SKIP_COMPILE" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fgpu-flush-denormals-to-zero -fno-slp-vectorize -fPIC -DDSPI_PART=$PART $EXTRA \
  -c dspi_kernels.hip -o /tmp/regs_p$PART.o -Rpass-analysis=kernel-resource-usage 2> /tmp/regs_p$PART.txt
python3 - "$PART" "$FILTER" <<'PY'
import re, subprocess, sys
txt = open(f"/tmp/regs_p{sys.argv[1]}.txt").read()
blocks = txt.split("Function Name: ")[1:]
names = [b.split("\n")[0].split()[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
seen = set()
for b, n in zip(blocks, dem):
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, -1])[1]
    m = re.search(r"(\w+<.*>)\(", n)
    key = m.group(1) if m else n
    if key in seen or not re.search(sys.argv[2], key): continue
    seen.add(key)
    scr = g(r"ScratchSize \[bytes/lane\]")
    print("%-70s VGPR %3s AGPR %3s scratch %4s SGPR %3s" % (key[-70:], g("VGPRs"), g("AGPRs"), scr, g("SGPRs")))
PY
