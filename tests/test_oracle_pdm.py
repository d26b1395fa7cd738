"""PDM sigma-delta oracle (oracle/orc_pdm.c): analytical identities.  The reference file (pdm_generator.c) cannot be
compiled here and ships no vectors — parity for this consumer is UNPINNED; these checks tie the restatement to what
the modulator must do by construction."""
import numpy as np

from orclib import PdmOracle

Q28 = 1 << 28


def density(words):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8)).mean()


def test_dc_bit_density_tracks_the_target():
    """2nd-order sigma-delta with feedback 65535: long-run ones density = (pcm + 32768) / 65535 (pdm_generator.c:363-377)."""
    for level in (-0.8, -0.25, 0.0, 0.1, 0.5, 0.89):
        o = PdmOracle()
        w = o.run(np.full(6000, int(level * Q28), dtype=np.int32))
        pcm = int(level * Q28) >> 14
        assert abs(density(w[2000:]) - (pcm + 32768) / 65535) < 2e-4, level


def test_hard_limiter_and_fade_in():
    o = PdmOracle()
    # Q28 full scale (1.0) is only 16384 after the >> 14; the limiter at PDM_CLIP_THRESH = 29500 (config.h:64) sits at 1.8
    w = o.run(np.full(4000, int(0.999 * Q28), dtype=np.int32))
    assert abs(density(w[2000:]) - ((int(0.999 * Q28) >> 14) + 32768) / 65535) < 2e-4
    o = PdmOracle()
    w = o.run(np.full(4000, int(1.95 * Q28), dtype=np.int32))
    assert abs(density(w[2000:]) - (29500 + 32768) / 65535) < 2e-4
    o = PdmOracle()
    w = o.run(np.full(4000, int(-3.0 * Q28), dtype=np.int32))
    assert abs(density(w[2000:]) - (-29500 + 32768) / 65535) < 2e-4
    # fade-in: the first sample is multiplied by 0/1024, sample 512 by 512/1024 (pdm_generator.c:356-360)
    o = PdmOracle()
    w = o.run(np.full(1100, int(0.5 * Q28), dtype=np.int32))
    d = [density(w[a:a + 64]) for a in (0, 480, 1030)]
    assert abs(d[0] - 0.5) < 0.02 and 0.5 < d[1] < d[2] and abs(d[2] - (8192 + 32768) / 65535) < 5e-3


def test_restart_keeps_the_dither_rng_running():
    a, b = PdmOracle(), PdmOracle()
    x = (np.sin(np.arange(3000) * 0.01) * 0.3 * Q28).astype(np.int32)
    wa = a.run(x); a.restart(); wa2 = a.run(x)
    wb = b.run(x); wb2 = b.run(x)
    assert np.array_equal(wa, wb) and not np.array_equal(wa2, wb2)        # state was cleared ...
    assert not np.array_equal(wa2, wa)                                    # ... but not the RNG (pdm_generator.c:241-252)


def test_silence_idles_around_half_density_with_dither():
    o = PdmOracle()
    w = o.run(np.zeros(3000, dtype=np.int32))
    assert abs(density(w[1000:]) - 32768 / 65535) < 2e-4
    assert len(np.unique(w[1000:])) > 50          # dithered: not a fixed idle pattern
