"""Quick GPU parity + throughput check used during development (the real tests live in tests/)."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from orclib import Oracle
from dspi_amd import wire as W, workloads as WL
from dspi_amd.host import Dspi


def parity(flavor, fs, B, n_blocks, S, blob, vol=-20 * 256, bit_depth=16, label=''):
    d = Dspi(flavor, S, device=0)
    d.set_rate(fs); d.set_volume(vol)
    assert d.load_bulk(blob) == 0
    pcm = WL.synth_pcm16(S, B * n_blocks, fs)
    data = pcm if bit_depth == 16 else WL.pcm16_to_pcm24_bytes(pcm)
    # two calls to exercise state carry across launches
    h = n_blocks // 2
    outs = []
    for part in range(2):
        sl = data[:, part * h * B:(part + 1) * h * B] if bit_depth == 16 else data[:, part * h * B * 6:(part + 1) * h * B * 6]
        outs.append(d.process_host(np.ascontiguousarray(sl), h, B, bit_depth))
    pairs = np.concatenate([o[0] for o in outs], axis=2)
    sub = np.concatenate([o[1] for o in outs], axis=1)
    peaks = np.concatenate([o[2] for o in outs], axis=1)
    bad = 0
    for s in range(S):
        o = Oracle(flavor, detmath=True)
        o.set_rate(fs); o.set_volume(vol); assert o.load_bulk(blob) == 0
        rp, rs, rk, rclip = o.process(data[s], 2 * h, B, bit_depth)
        ok = np.array_equal(rp, pairs[s]) and np.array_equal(rs, sub[s]) and np.array_equal(rk, peaks[s])
        st_ok = o.status() == d.status(s)
        if not (ok and st_ok):
            bad += 1
            if bad <= 3:
                dp = np.argwhere(rp != pairs[s]); ds = np.argwhere(rs != sub[s]); dk = np.argwhere(rk != peaks[s])
                print(f'  stream {s}: pairs diff {len(dp)} first {dp[:1].tolist()} sub diff {len(ds)} first {ds[:1].tolist()} peaks diff {len(dk)} first {dk[:1].tolist()} status {st_ok}')
                if len(dp):
                    i = tuple(dp[0]); print('    ref', rp[i], 'gpu', pairs[s][i])
    print(f'{label}: flavor {flavor} fs {fs} B {B} S {S}: {"OK" if bad == 0 else f"MISMATCH in {bad} streams"}')
    d.close()
    return bad == 0


if __name__ == '__main__':
    ok = True
    ok &= parity(1, 48000, 48, 20, 8, WL.config2_blob(), label='config2 svf+biquad')
    ok &= parity(1, 48000, 48, 20, 8, WL.config2_blob(True), label='config2 all-biquad')
    ok &= parity(1, 96000, 96, 12, 70, WL.full_chain_blob(1), label='config3 full chain')
    ok &= parity(1, 44100, 45, 12, 5, WL.full_chain_blob(1), label='full chain 44.1k B=45 (tail path)')
    ok &= parity(1, 48000, 48, 12, 5, WL.full_chain_blob(1), bit_depth=24, label='full chain 24-bit')
    ok &= parity(1, 96000, 96, 12, 3, WL.full_chain_blob(1), vol=0, label="full chain vol 0 dB (sign quirk)")
    ok &= parity(0, 48000, 48, 20, 8, WL.config1_blob(), vol=-10 * 256, label="Q28 config1")
    ok &= parity(0, 48000, 48, 12, 70, WL.full_chain_blob(0), label="Q28 full chain (config 5)")
    ok &= parity(0, 44100, 45, 12, 5, WL.full_chain_blob(0), bit_depth=24, label="Q28 full chain 44.1k 24-bit tail")
    ok &= parity(0, 48000, 48, 12, 3, WL.full_chain_blob(0), vol=0, label="Q28 vol 0 dB")
    print('ALL OK' if ok else 'FAILURES')
