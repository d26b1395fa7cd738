"""A minimal interpreter for the Thumb-1 (Cortex-M0+) subset that firmware/DSPi/dsp_process_rp2040.S uses — TEST INFRASTRUCTURE ONLY.

The RP2040 flavour's block biquad is hand-written assembly; there is no ARM toolchain or emulator in the build container, so the
oracle's restatement of it (oracle/orc_chain.c:q28_biquad_block) could only be checked by reading.  This module executes the
reference's assembly TEXT, read in place from the reference tree, instruction by instruction on a byte-addressed memory model with
32-bit wrapping registers, so the restatement is pinned to the instruction sequence the firmware actually runs
(tests/test_oracle_thumb.py; the vectors it produced are committed as tests/golden/q28_thumb_biquad.npz).

Supported: push / pop (register lists, lr / pc), mov (any registers), ldr rX, =symbol | [rN, #imm] | [sp, #imm], ldrb [rN, #imm] |
[rN, rM], str, cmp #imm, beq / bne, asrs / lsls #imm, uxth, muls (two-operand), adds / subs (three-operand, two-operand, immediate),
add sp, #imm; `#define NAME value` constants; labels.  Flags: Z only (all the conditional branches in the file are eq / ne).
"""
import re

MASK = 0xFFFFFFFF
REGS = {f"r{i}": i for i in range(13)}
REGS.update(sp=13, lr=14, pc=15)
RETURN = 0xFFFFFFFE                      # the lr value that ends a call


def _s32(v):
    v &= MASK
    return v - (1 << 32) if v & 0x80000000 else v


class Thumb:
    def __init__(self, text, symbols=None, mem_size=1 << 20):
        self.defines, self.labels, self.prog = {}, {}, []
        self.symbols = dict(symbols or {})
        self.mem = bytearray(mem_size)
        for raw in text.splitlines():
            line = raw.split("//")[0].split("@")[0].strip()
            if not line: continue
            m = re.match(r"#define\s+(\w+)\s+(\S+)", line)
            if m:
                self.defines[m.group(1)] = int(m.group(2), 0); continue
            if line.startswith("."):
                m = re.match(r"(\.\w+):$", line)
                if m: self.labels[m.group(1)] = len(self.prog)
                continue                                   # other directives
            m = re.match(r"(\w+):$", line)
            if m:
                self.labels[m.group(1)] = len(self.prog); continue
            op, _, rest = line.partition(" ")
            self.prog.append((op.strip(), rest.strip(), raw.strip()))

    # ---- memory ----
    def rd32(self, a): return int.from_bytes(self.mem[a:a + 4], "little")
    def wr32(self, a, v): self.mem[a:a + 4] = (v & MASK).to_bytes(4, "little")

    def _imm(self, tok):
        tok = tok.strip().lstrip("#")
        return self.defines[tok] if tok in self.defines else int(tok, 0)

    def _addr(self, expr, r):
        inner = expr.strip()[1:-1]
        parts = [p.strip() for p in inner.split(",")]
        base = r[REGS[parts[0]]]
        if len(parts) == 1: return base & MASK
        return (base + (r[REGS[parts[1]]] if parts[1] in REGS else self._imm(parts[1]))) & MASK

    @staticmethod
    def _reglist(s):
        out = []
        for part in s.strip()[1:-1].split(","):
            part = part.strip()
            if "-" in part:
                a, b = part.split("-"); out += list(range(REGS[a.strip()], REGS[b.strip()] + 1))
            else: out.append(REGS[part])
        return sorted(out)

    def call(self, label, args, max_steps=50_000_000):
        r = [0] * 16
        for i, v in enumerate(args): r[i] = v & MASK
        r[13] = len(self.mem) - 64
        r[14] = RETURN
        pc = self.labels[label]
        z = False
        steps = 0
        while True:
            steps += 1
            if steps > max_steps: raise RuntimeError("runaway")
            op, rest, raw = self.prog[pc]
            pc += 1
            a = [x.strip() for x in re.split(r",\s*(?![^\[]*\])(?![^{]*})", rest)] if rest else []
            if op == "push":
                for reg in reversed(self._reglist(rest)):
                    r[13] = (r[13] - 4) & MASK; self.wr32(r[13], r[reg])
            elif op == "pop":
                ret = False
                for reg in self._reglist(rest):
                    v = self.rd32(r[13]); r[13] = (r[13] + 4) & MASK
                    if reg == 15:
                        if v != RETURN: raise RuntimeError("pop pc to an address that is not the caller")
                        ret = True
                    else: r[reg] = v
                if ret: return r
            elif op == "mov":
                r[REGS[a[0]]] = r[REGS[a[1]]]
            elif op == "ldr":
                if a[1].startswith("="): r[REGS[a[0]]] = self.symbols[a[1][1:]] & MASK
                else: r[REGS[a[0]]] = self.rd32(self._addr(a[1], r))
            elif op == "ldrb":
                r[REGS[a[0]]] = self.mem[self._addr(a[1], r)]
            elif op == "str":
                self.wr32(self._addr(a[1], r), r[REGS[a[0]]])
            elif op == "cmp":
                z = ((r[REGS[a[0]]] - self._imm(a[1])) & MASK) == 0
            elif op in ("beq", "bne"):
                if z == (op == "beq"): pc = self.labels[a[0]]
            elif op == "asrs":
                v = (_s32(r[REGS[a[1]]]) >> self._imm(a[2])) & MASK; r[REGS[a[0]]] = v; z = v == 0
            elif op == "lsls":
                v = (r[REGS[a[1]]] << self._imm(a[2])) & MASK; r[REGS[a[0]]] = v; z = v == 0
            elif op == "uxth":
                r[REGS[a[0]]] = r[REGS[a[1]]] & 0xFFFF
            elif op == "muls":
                v = (r[REGS[a[0]]] * r[REGS[a[1]]]) & MASK; r[REGS[a[0]]] = v; z = v == 0       # low 32 bits: sign-agnostic
            elif op in ("adds", "subs", "add"):
                sign = -1 if op == "subs" else 1
                if len(a) == 3: x, y = r[REGS[a[1]]], (r[REGS[a[2]]] if a[2] in REGS else self._imm(a[2]))
                else: x, y = r[REGS[a[0]]], (r[REGS[a[1]]] if a[1] in REGS else self._imm(a[1]))
                v = (x + sign * y) & MASK; r[REGS[a[0]]] = v
                if op != "add": z = v == 0
            else:
                raise NotImplementedError(raw)
