"""Ablation timing of the chain kernel on the bench workload (development aid): which stage costs what."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi

S = int(os.environ.get('S', 65536)); NB = 25; B = 96; FS = 96000
dev = torch.device('cuda', 0)
pcm = torch.randint(-16384, 16385, (S, NB * B, 2), dtype=torch.int16, device=dev)
TILED = not os.environ.get('STREAM_MAJOR')
pairs = torch.empty((S * 8 * NB * B,), dtype=torch.int32, device=dev)      # either layout: same size (S % 128 == 0)
sub = torch.empty((S * NB * B,), dtype=torch.int32, device=dev)
peaks = torch.empty((S, NB, 11), dtype=torch.int16, device=dev)


def run(label, blob, outputs=True, steps=4):
    if os.environ.get('ONLY') and os.environ['ONLY'] not in label: return
    d = Dspi(1, S, device=0, fma=bool(os.environ.get('FMA')))
    d.set_rate(FS); d.set_volume(-20 * 256)
    assert d.load_bulk(blob) == 0
    if os.environ.get('PERSTREAM'):      # every stream its own preset of one structure: the packed kernel with per-lane values
        import struct
        for s_ in range(S):
            d.vendor_set(W.REQ["SET_PREAMP"], 0, struct.pack("<f", -6.0 - 0.001 * s_), stream=s_)
            if os.environ.get('PERSTREAM') == 'eq':      # a band's gain per stream as well: per-lane filters (DSPI_DEBUG=1: every band)
                ch = int(os.environ.get('EQ_CH', 0)); p = blob['eq'][ch][1]
                d.vendor_set(W.REQ["SET_EQ_PARAM"], 0, struct.pack("<BBBBfff", ch, 1, int(p['type']), 0, float(p['freq']), float(p['q']), 1.0 + 0.0001 * s_), stream=s_)
    args = (pairs.data_ptr(), sub.data_ptr(), peaks.data_ptr()) if outputs else (0, 0, 0)
    d.process_device(pcm.data_ptr(), NB, B, 16, *args, tiled=TILED); d.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        d.process_device(pcm.data_ptr(), NB, B, 16, *args, tiled=TILED)
    d.sync()
    dt = (time.perf_counter() - t0) / steps
    extra = ''
    if os.environ.get('DSPI_LIB', '').endswith('timing.so'):
        import ctypes
        buf = (ctypes.c_ulonglong * 84)()
        d.L.dspi_debug_wave_timing(buf, 1)
        nwg = (S + 127) // 128
        per = [buf[i] / nwg / (steps + 1) for i in range(24)]     # cycles per workgroup per launch
        extra = '\n      busy/total Mcyc per WG: ' + ' '.join(f'w{w}:{per[2*w]/1e6:.2f}/{per[2*w+1]/1e6:.2f}' for w in range(12)) + '  (r0 intake r1 hand-off r2 idle r3.. outputs 0 3 6 | 1 4 7 | 2 5 8)  simd(wg0): ' + ' '.join(str((buf[24+w] >> 4) & 3) for w in range(12))
    if os.environ.get('DSPI_LIB', '').endswith('timing.so'):
        extra += '\n      output phases Mcyc (eq / wait dl / emit / dl store): ' + ' '.join('o%d:' % o + '/'.join(f'{buf[36 + 4 * (3 + o) + k] / nwg / (steps + 1) / 1e6:.2f}' for k in range(4)) for o in range(9))
    if os.environ.get('DSPI_LIB', '').endswith('timing.so'):
        extra += '\n      delay / ring row accesses per launch (fast / per-stream path): loads %d / %d, stores %d / %d, other-wave loads %d / %d' % tuple(buf[36 + k] // (steps + 1) for k in (0, 1, 2, 3, 4, 5))
    print(f'{label:44s} {dt * 1e3:8.2f} ms/step  {S * NB * B / dt / 1e9:7.2f} Gframes/s{extra}', flush=True)
    d.close()


full = WL.full_chain_blob(1)
run('full chain', full)
if os.environ.get('QUICK'): sys.exit(0)
run('full chain, no output buffers', full, outputs=False)
b = full.copy(); b['outputs']['delay_ms'] = 0.0
run('no user delays (sub align only)', b)
b = full.copy(); b['leveller']['enabled'] = 0
run('leveller off', b)
b = full.copy(); b['leveller']['enabled'] = 0; b['outputs']['delay_ms'] = 0.0
run('leveller off + no delays', b)
run('leveller off + no delays + no outputs', b, outputs=False)
b2 = b.copy(); b2['eq']['type'][2:] = 0
run('  ... + output EQ flat', b2, outputs=False)
b3 = b2.copy(); b3['eq']['type'][:] = 0; b3['global_']['loudness_enabled'] = 0; b3['crossfeed']['enabled'] = 0
run('  ... + everything flat (I/O skeleton only)', b3, outputs=False)
run('everything flat, with outputs', b3)
b4 = full.copy(); b4['outputs']['enabled'][2:] = 0
run('full chain, only outputs 0-1 enabled', b4)
b = full.copy(); b['eq']['type'][0:2] = 0
run('full chain, master EQ flat', b)
b = full.copy(); b['global_']['loudness_enabled'] = 0
run('full chain, loudness off', b)
b = full.copy(); b['eq']['type'][0:2] = 0; b['global_']['loudness_enabled'] = 0
run('full chain, master EQ flat + loudness off', b)
b = full.copy(); b['eq']['type'][2:] = 0
run('full chain, output EQ flat', b)
b = full.copy(); b['eq']['type'][2:, 5:] = 0
run('full chain, output EQ bands 5-9 flat', b)
b = full.copy(); b['crossfeed']['enabled'] = 0
run('full chain, crossfeed off', b)
