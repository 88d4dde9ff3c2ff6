"""CPU: FLAC writer/reader round trip (bit-exact PCM) and the drop-in
audio_from_file / audio_to_flac helpers (util_audio.py:962-968)."""
import os
import sys

import numpy as np
import pytest

from amt_saga import flac

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('n,ch,bps', [(10000, 1, 24), (4096, 1, 16), (4097, 2, 24), (1, 1, 24), (12345, 1, 8)])
def test_round_trip_bit_exact(tmp_path, n, ch, bps):
    rng = np.random.default_rng(n)
    pcm = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=(n, ch))
    path = str(tmp_path / 'x.flac')
    flac.encode(pcm, path, sr=44100, bps=bps)
    got, sr, b2 = flac.decode(path)
    assert sr == 44100 and b2 == bps and np.array_equal(got, pcm)


def test_float_helpers_and_dropin(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'amt-saga_amd', 'dropin'))
    import util_audio as ua
    t = np.arange(30000) / 44100.0
    y = (0.6 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    y[100] = 1.5                                         # clips like soundfile PCM_24
    path = str(tmp_path / 'tone.flac')
    ua.audio_to_flac(y, path)
    back, sr = ua.audio_from_file(path)
    assert sr == 44100 and back.dtype == np.float32 and back.shape == y.shape
    m = np.arange(len(y)) != 100
    assert np.abs(back[m] - y[m]).max() <= 0.5 / (1 << 23) + 1e-9
    assert abs(back[100] - (1 - 2.0 ** -23)) < 1e-7


def test_decoder_matches_committed_fixture(golden_dir):
    """The fixtures in tests/golden were decoded from the reference's FLAC files with this
    same decoder; re-encoding them (verbatim) and decoding again must reproduce the PCM."""
    import recorded as rec
    import tempfile
    g = rec.triple('piano')['guess']
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'g.flac')
        flac.encode(g[:20000], p, bps=24)
        got, sr, bps = flac.decode(p)
        assert np.array_equal(got[:, 0], g[:20000])


def test_integrity_checks(tmp_path):
    """The reader proves byte work bit-exact: a flipped payload bit fails the frame CRC-16, a flipped header
    bit the CRC-8, and samples that decode but differ from the encoded audio fail the STREAMINFO MD5."""
    rng = np.random.default_rng(1)
    pcm = rng.integers(-(1 << 23), 1 << 23, size=(9000, 1))
    path = str(tmp_path / 'x.flac')
    flac.encode(pcm, path, bps=24)
    raw = bytearray(open(path, 'rb').read())
    assert raw[8 + 18:8 + 34] == flac.pcm_md5(pcm, 24) != bytes(16)
    flac.decode(path)
    bad = bytearray(raw); bad[200] ^= 0x10                     # inside the first frame's samples
    open(path, 'wb').write(bad)
    with pytest.raises(ValueError, match='CRC-16'):
        flac.decode(path)
    got, _, _ = flac.decode(path, verify=False)
    assert not np.array_equal(got, pcm)
    first = 4 + 4 + 34                                         # fLaC + block header + STREAMINFO
    bad = bytearray(raw); bad[first + 3] ^= 0x02               # channel/sample-size byte of the frame header
    open(path, 'wb').write(bad)
    with pytest.raises(ValueError, match='CRC-8'):
        flac.decode(path)
    bad = bytearray(raw); bad[8 + 20] ^= 0xFF                  # the stored MD5 itself
    open(path, 'wb').write(bad)
    with pytest.raises(ValueError, match='MD5'):
        flac.decode(path)
