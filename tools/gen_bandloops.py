#!/usr/bin/env python3
"""Generates dspi_amd/csrc/dspi_bandloops.inc: the hand-scheduled gfx950 EQ band loops (inline asm).

Each function runs ONE band over a 16-sample chunk, in place on tied VGPRs ("+v"), with the coefficients as
SGPR operands.  One multiply / add / subtract per operation of the reference loops
(firmware/DSPi/dsp_pipeline.c:298-362), in the reference's association order, so the results are bit-identical
to the C code compiled without contraction.

Two families:
  band16_*     one stream per lane  (v_mul_f32 / v_add_f32 / v_sub_f32)          — Q28-free scalar float kernel
  band16pk_*   two streams per lane (v_pk_mul_f32 / v_pk_add_f32, VOP3P)         — the packed float kernel
               coefficients arrive as the three aligned SGPR pairs {c0,c1} {c2,c3} {c4,c5} straight from the
               s_load of a DevBand; op_sel picks the half and broadcasts it to both streams.  a - b is emitted as
               v_pk_add_f32 a, -b (neg_lo/neg_hi), which is the same IEEE operation.

Why asm: hipcc's phi/copy handling around the five-way kind switch cost ~48 v_mov per band visit, and a SIMD
issues one VALU instruction per ~4.1 cycles whatever the occupancy (tools/probe/probe3), so every copy is a lost
slot.  The packed family is additionally list-scheduled: a dependent v_pk op can only issue ~9 cycles after its
producer (tools/probe/probe4), so independent work of the following samples is interleaved into the recurrences.
"""
import os

T = 16
FMA = False     # which float contract the lists below describe; main() emits both (see ops())
VCOEF = False   # packed family with PER-LANE coefficients: six VGPR pairs c0..c5 (one value per stream of the lane), no op_sel

# ---------------------------------------------------------------------------------------------------------
# operation lists: (op, dst, a, b) with op in mul/add/sub/mov; operands: 'x' (sample, in place), 's1','s2',
# 'c0'..'c5', temps 't0'..'t3'
# ---------------------------------------------------------------------------------------------------------
def ops(kind):
    """Canonical contract: one instruction per reference operation.  FMA contract (the firmware as built, GCC's
    -ffp-contract=fast on the Cortex-M33): the fused multiply-adds of GCC's GIMPLE for dsp_pipeline.c:298-362, read off
    -fdump-tree-optimized (oracle/Makefile FMA_FLAGS; oracle/orc_leaf.c orc_dsp_process_channel_block is the same list):
        biquad  y = fma(b0,in,s1); s1 = fma(b1,in,-(a1*y)) + s2; s2 = fma(b2,in,-(a2*y))
        SVF     v3 = in - ic2; v1 = fma(a1,ic1,a2*v3); v2 = fma(a3,v3,fma(a2,ic1,ic2)); ic = fma(2,v,-ic)
                LP v2 | HP fma(m1,v1,in) - v2 | PK fma(m1,v1,in) | shelf fma(m2,v2,fma(m0,in,m1*v1))
    ops are (op, dst, a, b[, c]); 'fma' is dst = a*b + c, a leading '-' on c negates it."""
    if FMA:
        return ops_fma(kind)
    if kind == 'BQ':   # TDF2 biquad
        return [('mul', 't0', 'c0', 'x'),      # b0*in
                ('mul', 't1', 'c1', 'x'),      # b1*in
                ('mul', 't2', 'c2', 'x'),      # b2*in
                ('add', 'x', 't0', 's1'),      # y = b0*in + s1
                ('mul', 't0', 'c3', 'x'),      # a1*y
                ('mul', 't3', 'c4', 'x'),      # a2*y
                ('sub', 't1', 't1', 't0'),     # b1*in - a1*y
                ('add', 's1', 't1', 's2'),     # ... + s2
                ('sub', 's2', 't2', 't3')]     # b2*in - a2*y
    v2 = 'x' if kind == 'LP' else 't2'         # low-pass: the output IS v2, form it in place (no copy)
    core = [('sub', 't0', 'x', 's2'),          # v3 = in - ic2eq
            ('mul', 't1', 'c0', 's1'),         # a1*ic1eq
            ('mul', 't2', 'c1', 't0'),         # a2*v3
            ('add', 't1', 't1', 't2'),         # v1
            ('mul', 't2', 'c1', 's1'),         # a2*ic1eq
            ('add', 't2', 's2', 't2'),         # ic2eq + a2*ic1eq
            ('mul', 't0', 'c2', 't0'),         # a3*v3
            ('add', v2, 't2', 't0'),           # v2
            ('fma2', 's1', 't1', 's1'),        # ic1eq = 2 v1 - ic1eq: ONE fma(2, v1, -ic1eq).  2*v1 is exact, so the
            ('fma2', 's2', v2, 's2')]          # ic2eq = 2 v2 - ic2eq  single rounding equals the reference's mul+sub bit for bit
    if kind == 'LP':
        return core
    if kind == 'HP':
        return core + [('mul', 't0', 'c3', 't1'), ('add', 'x', 'x', 't0'), ('sub', 'x', 'x', 't2')]
    if kind == 'PK':
        return core + [('mul', 't0', 'c3', 't1'), ('add', 'x', 'x', 't0')]
    if kind == 'SH':
        return core + [('mul', 't0', 'c3', 'x'), ('mul', 't1', 'c4', 't1'), ('add', 't0', 't0', 't1'),
                       ('mul', 't2', 'c5', 't2'), ('add', 'x', 't0', 't2')]
    raise ValueError(kind)


def ops_fma(kind):
    if kind == 'BQ':   # the input must outlive y (b1*in and b2*in are fused with products of y): y lives in t3, one copy at the end
        return [('fma', 't3', 'c0', 'x', 's1'),     # y = b0*in + s1
                ('mul', 't0', 'c3', 't3'),          # a1*y
                ('fma', 't1', 'c1', 'x', '-t0'),    # b1*in - a1*y
                ('add', 's1', 't1', 's2'),          # ... + s2
                ('mul', 't2', 'c4', 't3'),          # a2*y
                ('fma', 's2', 'c2', 'x', '-t2'),    # b2*in - a2*y
                ('mov', 'x', 't3', None)]
    v2 = 'x' if kind == 'LP' else 't2'
    core = [('sub', 't0', 'x', 's2'),               # v3 = in - ic2eq
            ('mul', 't2', 'c1', 't0'),              # a2*v3
            ('fma', 't1', 'c0', 's1', 't2'),        # v1 = a1*ic1eq + a2*v3
            ('fma', 't2', 'c1', 's1', 's2'),        # a2*ic1eq + ic2eq
            ('fma', v2, 'c2', 't0', 't2'),          # v2 = a3*v3 + (...)
            ('fma2', 's1', 't1', 's1'),             # ic1eq = 2 v1 - ic1eq
            ('fma2', 's2', v2, 's2')]               # ic2eq = 2 v2 - ic2eq
    if kind == 'LP':
        return core
    if kind == 'HP':
        return core + [('fma', 'x', 'c3', 't1', 'x'), ('sub', 'x', 'x', 't2')]
    if kind == 'PK':
        return core + [('fma', 'x', 'c3', 't1', 'x')]
    if kind == 'SH':
        return core + [('mul', 't0', 'c4', 't1'), ('fma', 'x', 'c3', 'x', 't0'), ('fma', 'x', 'c5', 't2', 'x')]
    raise ValueError(kind)


def split(o):
    """(op, d, a, b, c, negc) of an ops() entry"""
    op, d, a, b = o[:4]
    c = o[4] if len(o) > 4 else None
    neg = bool(c) and c[0] == '-'
    return op, d, a, b, (c[1:] if neg else c), neg


# ---------------------------------------------------------------------------------------------------------
# scalar family: straight program order (a dependent non-packed op issues back to back)
# ---------------------------------------------------------------------------------------------------------
def scalar_line(o, i):
    op, d, a, b, c, neg = split(o)

    def r(n):
        return '%%[x%d]' % i if n == 'x' else '%%[%s]' % n
    if op == 'fma2':
        return "v_fma_f32 %s, 2.0, %s, -%s" % (r(d), r(a), r(b))
    if op == 'fma':
        return "v_fma_f32 %s, %s, %s, %s%s" % (r(d), r(a), r(b), '-' if neg else '', r(c))
    if op == 'mov':
        return "v_mov_b32 %s, %s" % (r(d), r(a))
    return "v_%s_f32 %s, %s, %s" % (op, r(d), r(a), r(b))


def block_scalar(kind):
    return [scalar_line(o, i) for i in range(T) for o in ops(kind)]


# ---------------------------------------------------------------------------------------------------------
# packed family: list scheduling over the whole 16-sample block
# ---------------------------------------------------------------------------------------------------------
NTSETS = int(os.environ.get('BL_NTSETS', 3))        # rotating temp sets; sample i uses set i % NTSETS
LAT_SLOTS = int(os.environ.get('BL_LAT', 3))        # a dependent packed op wants >= this many issue slots after its producer


def pk_operand(n, i):
    if n == 'x':
        return '%%[x%d]' % i, ''
    if n in ('s1', 's2'):
        return '%%[%s]' % n, ''
    if n[0] == 't':
        return '%%[t%d_%s]' % (i % NTSETS, n[1]), ''
    c = int(n[1])
    if VCOEF:
        return '%%[c%d]' % c, ''
    return '%%[c%d%d]' % (c & ~1, c | 1), ('lo' if c % 2 == 0 else 'hi')


def pk_text(op, dn, an, asel, bn, cn, neg):
    """One VOP3P instruction.  A coefficient is always the first source: op_sel picks its half of the SGPR pair and
    broadcasts it to both streams (lo: op_sel 0 / op_sel_hi 0, hi: op_sel 1 / op_sel_hi 1)."""
    if op == 'fma2':
        return "v_pk_fma_f32 %s, %%[two], %s, %s neg_lo:[0,0,1] neg_hi:[0,0,1]" % (dn, an, bn)
    if op == 'mov':
        return "v_pk_mov_b32 %s, %s, %s op_sel:[0,1]" % (dn, an, an)
    if op == 'fma':
        mods = ''
        if asel == 'lo':
            mods += ' op_sel_hi:[0,1,1]'
        elif asel == 'hi':
            mods += ' op_sel:[1,0,0]'
        if neg:
            mods += ' neg_lo:[0,0,1] neg_hi:[0,0,1]'
        return "v_pk_fma_f32 %s, %s, %s, %s%s" % (dn, an, bn, cn, mods)
    mods = ''
    if asel == 'lo':
        mods += ' op_sel_hi:[0,1]'
    elif asel == 'hi':
        mods += ' op_sel:[1,0]'
    if op == 'sub':
        mods += ' neg_lo:[0,1] neg_hi:[0,1]'
    return "v_pk_%s_f32 %s, %s, %s%s" % ('mul' if op == 'mul' else 'add', dn, an, bn, mods)


def pk_line(o, i, operand=None):
    operand = operand or (lambda n: pk_operand(n, i))
    op, d, a, b, c, neg = split(o)
    dn, _ = operand(d)
    an, asel = operand(a)
    bn, bsel = operand(b) if b else (None, '')
    cn, csel = operand(c) if c else (None, '')
    assert not bsel and not csel, "coefficients are always the first source"
    return pk_text(op, dn, an, asel, bn, cn, neg)


def reads(o, phys):
    op, d, a, b, c, neg = split(o)
    return [phys(n) for n in (a, b, c) if n]


def schedule(inst, lat_slots):
    """List scheduling of `inst` (dicts with text / w / r): a dependent packed op wants >= lat_slots issue slots after
    its producer; ties go to the longest remaining critical path; bounded look-ahead keeps the schedule local."""
    n = len(inst)
    preds = [set() for _ in range(n)]      # (j, is_raw)
    last_w = {}
    readers = {}
    for k, it in enumerate(inst):
        for rr in it['r']:
            if rr in last_w:
                preds[k].add((last_w[rr], True))
        if it['w'] in last_w:
            preds[k].add((last_w[it['w']], False))
        for j in readers.get(it['w'], []):
            if j != k:
                preds[k].add((j, False))
        last_w[it['w']] = k
        readers[it['w']] = []
        for rr in it['r']:
            readers.setdefault(rr, []).append(k)
    succs = [[] for _ in range(n)]
    for k in range(n):
        for (j, raw) in preds[k]:
            succs[j].append((k, raw))
    # critical path (in slots) to the end of the block
    cp = [0] * n
    for k in range(n - 1, -1, -1):
        cp[k] = 1 + max([cp[s] + (lat_slots - 1 if raw else 0) for (s, raw) in succs[k]] or [0])
    done_slot = {}
    order = []
    slot = 0
    remaining = set(range(n))
    while remaining:
        best = None
        for k in sorted(remaining):
            if any(j not in done_slot for (j, _) in preds[k]):
                continue
            est = max([done_slot[j] + (lat_slots if raw else 1) for (j, raw) in preds[k]] or [0])
            key = (max(est, slot), -cp[k], k)
            if best is None or key < best[0]:
                best = (key, k)
            if k - min(remaining) > 64:      # bounded look-ahead keeps the schedule local
                break
        k = best[1]
        slot = max(best[0][0], slot)
        done_slot[k] = slot
        slot += 1
        order.append(k)
        remaining.discard(k)
    return [inst[k]['text'] for k in order], slot


def block_pk(kind):
    # instances with concrete register names for hazard analysis
    inst = []
    for i in range(T):
        for o in ops(kind):
            def phys(n):
                if n == 'x':
                    return 'x%d' % i
                if n[0] == 't':
                    return 't%d_%s' % (i % NTSETS, n[1])
                return n
            inst.append(dict(text=pk_line(o, i), w=phys(o[1]), r=reads(o, phys)))
    return schedule(inst, LAT_SLOTS)[0]


# ---------------------------------------------------------------------------------------------------------
# dual packed family: the SAME band kind on two independent channels (master EQ left / right), interleaved.
# One channel alone is latency-bound (a biquad's s1 -> y -> a1*y -> ... -> s1 loop is 4 dependent ops = ~64 cycles
# per sample against 36 cycles of issue); two interleaved chains keep a wave that has a SIMD to itself issuing.
# ---------------------------------------------------------------------------------------------------------
NTSETS2 = int(os.environ.get('BL_NTSETS2', 1))
LAT_SLOTS2 = int(os.environ.get('BL_LAT2', 4))


def pk_operand2(n, i, ch):
    if n == 'x':
        return '%%[x%s%d]' % (ch, i), ''
    if n in ('s1', 's2'):
        return '%%[%s%s]' % (n, ch), ''
    if n[0] == 't':
        return '%%[t%s%d_%s]' % (ch, i % NTSETS2, n[1]), ''
    c = int(n[1])
    return '%%[%s%d%d]' % (ch, c & ~1, c | 1), ('lo' if c % 2 == 0 else 'hi')


def pk_line2(o, i, ch):
    return pk_line(o, i, lambda n: pk_operand2(n, i, ch))


def block_pk2(kind):
    inst = []
    for i in range(T):
        for ch in 'ab':
            for o in ops(kind):
                def phys(n):
                    if n == 'x':
                        return 'x%s%d' % (ch, i)
                    if n[0] == 't':
                        return 't%s%d_%s' % (ch, i % NTSETS2, n[1])
                    if n in ('s1', 's2'):
                        return n + ch
                    return ch + n
                inst.append(dict(text=pk_line2(o, i, ch), w=phys(o[1]), r=reads(o, phys)))
    lines, slots = schedule(inst, LAT_SLOTS2)
    if os.environ.get('BL_VERBOSE'):
        print("dual %s: %d instructions in %d slots" % (kind, len(inst), slots))
    return lines


def block_pk1of2(kind, ch):
    """One channel of the pair on its own (the two channels' kinds differ): the single-channel schedule on this channel's
    operands, with both channels' temporaries as its rotating sets."""
    other = 'b' if ch == 'a' else 'a'
    sets = [(c, k) for k in range(NTSETS2) for c in (ch, other)]

    def tname(i, j):
        c, k = sets[i % len(sets)]
        return 't%s%d_%s' % (c, k, j)
    inst = []
    for i in range(T):
        for o in ops(kind):
            def phys(n):
                if n == 'x':
                    return 'x%s%d' % (ch, i)
                if n[0] == 't':
                    return tname(i, n[1])
                if n in ('s1', 's2'):
                    return n + ch
                return ch + n

            def opnd(n):
                if n[0] == 't':
                    return '%%[%s]' % tname(i, n[1]), ''
                return pk_operand2(n, i, ch)
            inst.append(dict(text=pk_line(o, i, opnd), w=phys(o[1]), r=reads(o, phys)))
    return schedule(inst, LAT_SLOTS)[0]


def emit_dual(name, kinds, out, with_n=False):
    """Two channels in ONE asm statement (a single in-place update of both sample arrays for the register allocator):
    equal kinds run the interleaved loops, different kinds (or one channel bypassed) run channel a, then channel b.
    with_n: a ragged chunk (n < 16) runs channel a, then channel b, in program order with early exits."""
    lines = []
    if with_n:
        lines += ["s_cmp_lg_u32 %[n], 16", "s_cbranch_scc1 .Ltail_%="]
    lines += ["s_cmp_lg_u32 %[ka], %[kb]", "s_cbranch_scc1 .Lsplit_%="]
    lines += dispatch(kinds, block_pk2, 'ka', 'd', '.Lend_%=')
    lines += [".Lsplit_%=:"] + dispatch(kinds, lambda kind: block_pk1of2(kind, 'a'), 'ka', 'a', '.Lsplitb_%=')
    lines += [".Lsplitb_%=:"] + dispatch(kinds, lambda kind: block_pk1of2(kind, 'b'), 'kb', 'b', '.Lend_%=')
    if with_n:
        def tail_ch(ch, nxt):
            def blk(kind):
                ls = []
                for i in range(T):
                    ls += ["s_cmp_le_u32 %%[n], %d" % i, "s_cbranch_scc1 %s" % nxt]
                    ls += [pk_line2(o, i, ch) for o in ops(kind)]
                return ls
            return blk
        lines += [".Ltail_%=:"] + dispatch(kinds, tail_ch('a', '.Ltailb_%='), 'ka', 'ta', '.Ltailb_%=')
        lines += [".Ltailb_%=:"] + dispatch(kinds, tail_ch('b', '.Lend_%='), 'kb', 'tb', '.Lend_%=')
    lines += [".Lend_%=:"]
    body = '\n'.join('        "%s\\n\\t"' % l for l in lines)
    xs = ', '.join('[x%s%d] "+v"(x%s[%d])' % (ch, i, ch, i) for ch in 'ab' for i in range(T))
    tn = ['t%s%d_%d' % (ch, s_, j) for ch in 'ab' for s_ in range(NTSETS2) for j in range(4)]
    ts = ', '.join('[%s] "=&v"(%s)' % (t, t) for t in tn)
    out.append("__device__ __forceinline__ void %s(v2f (&xa)[16], v2f (&xb)[16], v2f &s1a, v2f &s2a, v2f &s1b, v2f &s2b, uint32_t ka, uint32_t kb,\n"
               "        v2f a01, v2f a23, v2f a45, v2f b01, v2f b23, v2f b45%s) {" % (name, ", uint32_t n" if with_n else ""))
    out.append("    v2f %s;" % ', '.join(tn))
    out.append("    const v2f two = {2.0f, 2.0f};")
    out.append("    asm volatile(")
    out.append(body)
    out.append("        : %s, [s1a] \"+v\"(s1a), [s2a] \"+v\"(s2a), [s1b] \"+v\"(s1b), [s2b] \"+v\"(s2b), %s" % (xs, ts))
    out.append("        : [ka] \"s\"(ka), [kb] \"s\"(kb), [a01] \"s\"(a01), [a23] \"s\"(a23), [a45] \"s\"(a45), [b01] \"s\"(b01), [b23] \"s\"(b23), [b45] \"s\"(b45), [two] \"s\"(two)%s" % (', [n] \"s\"(n)' if with_n else ''))
    out.append("        : \"scc\");")
    out.append("}")
    out.append("")


# ---------------------------------------------------------------------------------------------------------
def emit(name, kinds, out, packed, vcoef=False):
    """One asm statement that dispatches on the (wave-uniform) band kind with scalar branches and runs the
    matching 16-sample loop.  Keeping the dispatch inside the asm means the compiler sees a single in-place
    update of x[0..15] — no phi copies on any path."""
    block = block_pk if packed else block_scalar
    lines = ["s_cmp_eq_u32 %[k], 0", "s_cbranch_scc1 .Lend_%="]
    for kname, kval in kinds[:-1]:
        lines += ["s_cmp_eq_u32 %%[k], %d" % kval, "s_cbranch_scc1 .L%s_%%=" % kname]
    last = kinds[-1][0]                                  # last kind is the fall-through
    lines += block(last)
    lines += ["s_branch .Lend_%="]
    for kname, kval in kinds[:-1]:
        lines += [".L%s_%%=:" % kname] + block(kname) + ["s_branch .Lend_%="]
    lines += [".Lend_%=:"]
    body = '\n'.join('        "%s\\n\\t"' % l for l in lines)
    xs = ', '.join('[x%d] "+v"(x[%d])' % (i, i) for i in range(T))
    if packed:
        tn = ['t%d_%d' % (s, j) for s in range(NTSETS) for j in range(4)]
        ts = ', '.join('[%s] "=&v"(%s)' % (t, t) for t in tn)
        cs = '[c01] "s"(c01), [c23] "s"(c23), [c45] "s"(c45), [two] "s"(two)'
        out.append("__device__ __forceinline__ void %s(v2f (&x)[16], v2f &s1, v2f &s2, uint32_t kind, v2f c01, v2f c23, v2f c45) {" % name)
        out.append("    v2f %s;" % ', '.join(tn))
        out.append("    const v2f two = {2.0f, 2.0f};")
    else:
        ts = ', '.join('[t%d] "=&v"(t%d)' % (i, i) for i in range(4))
        cs = ', '.join('[c%d] "%s"(c%d)' % (i, 'v' if vcoef else 's', i) for i in range(6))
        out.append("__device__ __forceinline__ void %s(float (&x)[16], float &s1, float &s2, uint32_t kind, float c0, float c1, float c2, float c3, float c4, float c5) {" % name)
        out.append("    float t0, t1, t2, t3;")
    out.append("    asm volatile(")
    out.append(body)
    out.append("        : %s, [s1] \"+v\"(s1), [s2] \"+v\"(s2), %s" % (xs, ts))
    out.append("        : [k] \"s\"(kind), %s" % cs)
    out.append("        : \"scc\");")
    out.append("}")
    out.append("")


def dispatch(kinds, blockfn, kreg, tag, nxt):
    """Scalar dispatch on the band kind held in %[kreg]: kind 0 -> nxt, every block ends with a branch to nxt."""
    lines = ["s_cmp_eq_u32 %%[%s], 0" % kreg, "s_cbranch_scc1 %s" % nxt]
    for kname, kval in kinds[:-1]:
        lines += ["s_cmp_eq_u32 %%[%s], %d" % (kreg, kval), "s_cbranch_scc1 .L%s%s_%%=" % (tag, kname)]
    lines += blockfn(kinds[-1][0]) + ["s_branch %s" % nxt]
    for kname, kval in kinds[:-1]:
        lines += [".L%s%s_%%=:" % (tag, kname)] + blockfn(kname) + ["s_branch %s" % nxt]
    return lines


def tail_block(kind, line_fn):
    """Ragged chunk: samples in program order with a scalar early exit in front of each one (n is wave-uniform: the
    last chunk of a 44/45-frame packet)."""
    lines = []
    for i in range(T):
        lines += ["s_cmp_le_u32 %%[n], %d" % i, "s_cbranch_scc1 .Lend_%="]
        lines += [line_fn(o, i) for o in ops(kind)]
    return lines


def emit_n(name, kinds, out):
    """Packed, full OR ragged chunk in ONE asm statement (kernels for 44.1 kHz packets): n == 16 takes the scheduled loop,
    n < 16 the program-order loop with early exits.  Two alternative statements would make the register allocator
    reconcile two placements of x[] at every band."""
    lines = ["s_cmp_lg_u32 %[n], 16", "s_cbranch_scc1 .Ltail_%="]
    lines += dispatch(kinds, block_pk, 'k', 'f', '.Lend_%=')
    lines += [".Ltail_%=:"]
    lines += dispatch(kinds, lambda kind: tail_block(kind, pk_line), 'k', 't', '.Lend_%=')
    lines += [".Lend_%=:"]
    body = '\n'.join('        "%s\\n\\t"' % l for l in lines)
    xs = ', '.join('[x%d] "+v"(x[%d])' % (i, i) for i in range(T))
    tn = ['t%d_%d' % (s_, j) for s_ in range(NTSETS) for j in range(4)]
    ts = ', '.join('[%s] "=&v"(%s)' % (t, t) for t in tn)
    out.append("__device__ __forceinline__ void %s(v2f (&x)[16], v2f &s1, v2f &s2, uint32_t kind, v2f c01, v2f c23, v2f c45, uint32_t n) {" % name)
    out.append("    v2f %s;" % ', '.join(tn))
    out.append("    const v2f two = {2.0f, 2.0f};")
    out.append("    asm volatile(")
    out.append(body)
    out.append("        : %s, [s1] \"+v\"(s1), [s2] \"+v\"(s2), %s" % (xs, ts))
    out.append("        : [k] \"s\"(kind), [c01] \"s\"(c01), [c23] \"s\"(c23), [c45] \"s\"(c45), [two] \"s\"(two), [n] \"s\"(n)")
    out.append("        : \"scc\");")
    out.append("}")
    out.append("")


def emit_v(name, kinds, out, with_n):
    """Packed, per-lane coefficients (rows whose streams carry different presets of one structure): the band kind is still
    wave-uniform (scalar dispatch), the six coefficients are VGPR pairs {stream a, stream b} loaded from the row's value tile."""
    lines = []
    if with_n:
        lines += ["s_cmp_lg_u32 %[n], 16", "s_cbranch_scc1 .Ltail_%="]
    lines += dispatch(kinds, block_pk, 'k', 'f', '.Lend_%=')
    if with_n:
        lines += [".Ltail_%=:"]
        lines += dispatch(kinds, lambda kind: tail_block(kind, pk_line), 'k', 't', '.Lend_%=')
    lines += [".Lend_%=:"]
    body = '\n'.join('        "%s\\n\\t"' % l for l in lines)
    xs = ', '.join('[x%d] "+v"(x[%d])' % (i, i) for i in range(T))
    tn = ['t%d_%d' % (s_, j) for s_ in range(NTSETS) for j in range(4)]
    ts = ', '.join('[%s] "=&v"(%s)' % (t, t) for t in tn)
    out.append("__device__ __forceinline__ void %s(v2f (&x)[16], v2f &s1, v2f &s2, uint32_t kind, v2f c0, v2f c1, v2f c2, v2f c3, v2f c4, v2f c5%s) {"
               % (name, ", uint32_t n" if with_n else ""))
    out.append("    v2f %s;" % ', '.join(tn))
    out.append("    const v2f two = {2.0f, 2.0f};")
    out.append("    asm volatile(")
    out.append(body)
    out.append("        : %s, [s1] \"+v\"(s1), [s2] \"+v\"(s2), %s" % (xs, ts))
    out.append("        : [k] \"s\"(kind), %s, [two] \"s\"(two)%s" % (', '.join('[c%d] \"v\"(c%d)' % (i, i) for i in range(6)), ', [n] \"s\"(n)' if with_n else ''))
    out.append("        : \"scc\");")
    out.append("}")
    out.append("")


HEADER = ["// %s — GENERATED by tools/gen_bandloops.py; do not edit by hand.",
          "// Hand-scheduled gfx950 band loops: 16 samples of one EQ band, in place on tied VGPRs (\"+v\"), coefficients in SGPRs.",
          "// One multiply/add/subtract per reference operation, in the reference's association order (dsp_pipeline.c:298-362).",
          "// kind: BandKind (dspi_image.h) — 0 bypass, 1 biquad, 2 SVF low-pass, 3 SVF high-pass, 4 SVF peaking, 5 SVF shelf."]


def main():
    global FMA, VCOEF, NTSETS
    allk = [('BQ', 1), ('LP', 2), ('HP', 3), ('PK', 4), ('SH', 5)]
    here = os.environ.get('BL_OUT') or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dspi_amd", "csrc")
    for fma in (False, True):
        FMA = fma
        f = "f" if fma else ""          # the FMA-contract families carry an f: band16pkf_any, band16pk2f_any, band16f_any, band16vf_any ...
        tag = "_fma" if fma else ""
        for fname, packed, note in (("dspi_bandloops%s.inc" % tag, False, "// one stream per lane (v_mul_f32 / v_add_f32 / v_sub_f32%s), program order" % (" / v_fma_f32" if fma else "")),
                                    ("dspi_bandloops_pk%s.inc" % tag, True, "// two streams per lane (v_pk_mul_f32 / v_pk_add_f32%s), list-scheduled; needs v2f" % (" / v_pk_fma_f32" if fma else ""))):
            out = [HEADER[0] % fname] + HEADER[1:] + [note]
            if fma:
                out += ["// FLOAT CONTRACT OF THE FIRMWARE BUILD (DSPI_FLOAT_CONTRACT_FMA): the fused multiply-adds GCC forms for these loops, see ops() in the generator."]
            out += [""]
            pre = ("band16pk" if packed else "band16") + f
            emit(pre + "_any", allk, out, packed)
            if not packed:
                emit(pre + "_shelf", [('SH', 5)], out, packed)      # loudness stages are shelves (or bypassed)
            if not packed:      # per-lane parameter kernel: coefficients are per-lane values (VGPRs), the kind is wave-uniform here
                emit("band16v%s_any" % f, allk, out, False, vcoef=True)
                emit("band16v%s_shelf" % f, [('SH', 5)], out, False, vcoef=True)
            if packed:
                emit_n(pre + "_any_n", allk, out)                         # kernels for ragged packets: full or short chunk
                emit_dual("band16pk2%s_any" % f, allk, out)              # master EQ, left + right interleaved (same kind in both)
                emit_dual("band16pk2%s_shelf" % f, [('SH', 5)], out)     # loudness, left + right
                emit_dual("band16pk2%s_any_n" % f, allk, out, with_n=True)
                emit_dual("band16pk2%s_shelf_n" % f, [('SH', 5)], out, with_n=True)
            path = os.path.join(here, fname)
            open(path, "w").write('\n'.join(out))
            print("wrote", os.path.normpath(path))
        # packed, per-lane coefficients: two rotating temp sets (that kernel runs at the memory system's pace and has no registers
        # to spare: its coefficients are twelve VGPRs per band in flight)
        VCOEF = True
        ntsets_pk, NTSETS = NTSETS, int(os.environ.get('BL_NTSETS_V', 2))
        fname = "dspi_bandloops_pkv%s.inc" % tag
        out = [HEADER[0] % fname] + HEADER[1:] + ["// two streams per lane, PER-LANE coefficients (six VGPR pairs), wave-uniform kind; list-scheduled; needs v2f"]
        if fma:
            out += ["// FLOAT CONTRACT OF THE FIRMWARE BUILD (DSPI_FLOAT_CONTRACT_FMA), see ops() in the generator."]
        out += [""]
        emit_v("band16pkv%s_any" % f, allk, out, False)
        emit_v("band16pkv%s_any_n" % f, allk, out, True)
        VCOEF = False
        NTSETS = ntsets_pk
        path = os.path.join(here, fname)
        open(path, "w").write('\n'.join(out))
        print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
