"""SoundFont 2 reader (amt_saga/sf2.py) on a file written by tests/sf2_fixture.py: zone flattening rules (global zones,
preset-level additions, range intersection, bank filter), the float64 playback definition (oracle/sf2.py) on known
answers, and malformed files.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd'), os.path.dirname(os.path.abspath(__file__))]
import sf2_fixture                                   # noqa: E402
from amt_saga import sf2                             # noqa: E402
from oracle import sf2 as osf2                       # noqa: E402


@pytest.fixture(scope='module')
def font():
    return sf2.SoundFont(sf2_fixture.build())


def test_reader_recovers_the_written_font(font):
    assert [h['name'] for h in font.headers] == ['sine441', 'pluck', 'square']
    assert font.headers[0]['rate'] == 22050 and font.headers[1]['correction'] == 5 and font.headers[2]['pitch'] == 71
    assert sorted(font.programs) == [0, 24, 40] and font.names[24] == 'Pluck'      # bank 128 is not a melodic program
    p0 = font.programs[0]
    assert [z['key'] for z in p0] == [(0, 72), (73, 127)] and all(z['loop'] for z in p0)
    # global zone: release 0.5 s and 3 dB for the zone without its own attenuation, 6 dB where the zone sets it
    assert abs(p0[0]['release'] - 0.5) < 1e-3 and abs(p0[1]['release'] - 0.5) < 1e-3
    assert abs(p0[0]['gain'] - 10 ** (-3 / 20)) < 1e-6 and abs(p0[1]['gain'] - 10 ** (-6 / 20)) < 1e-6
    assert p0[0]['tune'] == 0 and p0[1]['tune'] == 10 and p0[0]['root'] == 69
    assert p0[0]['loop_start'] == 2000 and p0[0]['loop_end'] == 17000
    # preset-level generators add: 2 dB, -5 cents (+5 cents pitch correction of the sample); velocity split; coarse tune
    p24 = font.programs[24]
    assert [z['vel'] for z in p24] == [(0, 100), (101, 127)] and not p24[0]['loop']
    assert abs(p24[0]['gain'] - 10 ** (-2 / 20)) < 1e-6 and p24[0]['tune'] == 0 and p24[1]['tune'] == -1200
    # preset global zone (+12 semitones) reaches both layers; the preset's key range cuts the instrument's
    p40 = font.programs[40]
    assert [z['key'] for z in p40] == [(40, 72), (73, 80), (0, 127)]
    assert [z['tune'] for z in p40] == [1200, 1210, 1200] and p40[2]['root'] == 72
    assert abs(p40[2]['attack'] - 0.02) < 1e-4 and abs(p40[2]['decay'] - 0.3) < 1e-3 and p40[2]['sustain_db'] == 12.0
    z, first = font.tables()
    assert z.shape == (7, sf2.ZONE_FIELDS) and first[0] == 0 and first[1] == 2 and first[24] == 2 and first[25] == 4
    assert first[41] == 7 and first[-1] == 7


def test_playback_definition_known_answers(font):
    sr = 44100
    # the looped 441 Hz sine (root 69) played at key 81 sounds 882 Hz for the whole note, long past the sample's end
    w = osf2.render_window([(0, 81, 100, 0.0, 1.5)], 2 * sr, font.samples, font.programs, sr)
    seg = w[sr:sr + 16384] * np.hanning(16384)
    f = np.fft.rfftfreq(16384, 1 / sr)[np.argmax(np.abs(np.fft.rfft(seg)))]
    assert abs(f - 882.0 * 2 ** (10 / 1200)) < 2.0                   # keys 73.. carry fineTune +10
    # amplitude law of render() (util_audio.py:778-781): a single note peaks at ((v - 12) / 128)^4
    assert abs(np.abs(w).max() - (88 / 128) ** 4) < 1e-6
    # release: 100 dB over 0.5 s -> 0.1 s after note off the level is 20 dB down
    k = int(1.6 * sr)
    a_on = np.abs(w[int(1.3 * sr):int(1.4 * sr)]).max()
    a_rel = np.abs(w[k - 30:k + 30]).max()
    assert abs(20 * np.log10(a_rel / a_on) + 20.0) < 0.7
    # the unlooped pluck ends with its sample: silence after 0.8 s of sample time (key 57 = root, 44.1 kHz)
    w = osf2.render_window([(24, 57, 90, 0.0, 1.2)], 2 * sr, font.samples, font.programs, sr)
    assert np.abs(w[int(0.81 * sr):]).max() == 0 and np.abs(w[:int(0.5 * sr)]).max() > 0
    # velocity split: above 100 the pluck sounds an octave lower (coarseTune -12) and lasts twice as long
    w = osf2.render_window([(24, 57, 120, 0.0, 2.0)], 3 * sr, font.samples, font.programs, sr)
    assert np.abs(w[int(1.5 * sr):int(1.58 * sr)]).max() > 0 and np.abs(w[int(1.62 * sr):]).max() == 0
    # a key outside every zone of the program is silent; an unknown program too
    assert np.all(osf2.render_window([(40, 20, 90, 0.0, 0.2), (7, 60, 90, 0.0, 0.2)], sr // 2,
                                     font.samples, {40: [font.programs[40][0]]}, sr) == 0)


def test_malformed_files_are_refused():
    good = bytearray(sf2_fixture.build())
    with pytest.raises(ValueError):
        sf2.SoundFont(b'RIFF\x04\x00\x00\x00WAVE')
    with pytest.raises(ValueError):
        sf2.SoundFont(bytes(good[:len(good) // 2]))                   # truncated
    bad = bytearray(good)
    i = bad.find(b'shdr')
    bad[i + 8 + 20:i + 8 + 24] = (10 ** 9).to_bytes(4, 'little')    # first sample starts far outside the pool
    with pytest.raises(ValueError):
        sf2.SoundFont(bytes(bad))
    bad = bytearray(good)
    i = bad.find(b'igen')
    bad[i + 4:i + 8] = (int.from_bytes(bad[i + 4:i + 8], 'little') - 2).to_bytes(4, 'little')   # not whole records
    with pytest.raises(ValueError):
        sf2.SoundFont(bytes(bad))
