"""Writer of a small SoundFont 2 file for the tests (no soundfont exists in this image or in the reference tree:
main.py:25-29 points at a user path).  Written from the SoundFont 2.04 specification's record layouts, independently of
the product's reader (amt_saga/sf2.py): RIFF sfbk = LIST INFO + LIST sdta (smpl) + LIST pdta (phdr pbag pmod pgen inst
ibag imod igen shdr), every list closed by its terminal record.

Content (what the tests assert the reader recovers):
  samples   0 'sine441'  22050 Hz, one second of a 441 Hz sine (50 frames per period), loop = 300 whole periods, root 69
            1 'pluck'    44100 Hz, decaying 220 Hz saw-ish tone, no loop, root 57, pitch correction +5 cents
            2 'square'   32000 Hz, 500 Hz band-limited square, looped, root 71 (overridden to 72 in its zone)
  instruments 0 'SineInst'  global zone (release 0.5 s, attenuation 3 dB); zone keys 0-72 -> sine441 looped;
                            zone keys 73-127 -> sine441 looped, fineTune +10, attenuation 6 dB
              1 'PluckInst' one zone, all keys, velocities 0-100 -> pluck; one zone velocities 101-127 -> pluck, coarse -12
              2 'SquareInst' one zone -> square looped, root key 72, attack 20 ms, decay 0.3 s, sustain 12 dB, release 0.2 s
  presets   program 0  'Sine'   -> SineInst
            program 24 'Pluck'  -> PluckInst, preset-level attenuation +2 dB, fineTune -5
            program 40 'Layer'  global zone (coarseTune +12); SineInst limited to keys 40-80 + SquareInst (layered)
            bank 128 program 0 'Drums' -> PluckInst (ignored by the reader: bank != 0)
"""
import struct

import numpy as np

GEN = dict(startloop=2, endloop=3, delay=33, attack=34, hold=35, decay=36, sustain=37, release=38, instrument=41,
           keyRange=43, velRange=44, attenuation=48, coarseTune=51, fineTune=52, sampleID=53, sampleModes=54, rootKey=58)


def tc(seconds):
    return int(round(1200 * np.log2(seconds)))


def _chunk(cid, payload):
    return cid + struct.pack('<I', len(payload)) + payload + (b'\0' if len(payload) & 1 else b'')


def _gen(name, amount):
    if name in ('keyRange', 'velRange'):
        return struct.pack('<HBB', GEN[name], amount[0], amount[1])
    return struct.pack('<Hh', GEN[name], amount) if name not in ('instrument', 'sampleID', 'sampleModes') else \
        struct.pack('<HH', GEN[name], amount)


def build():
    sr0, sr1, sr2 = 22050, 44100, 32000
    n0 = np.arange(sr0)
    s0 = 0.8 * np.sin(2 * np.pi * 441.0 * n0 / sr0)
    n1 = np.arange(int(0.8 * sr1))
    s1 = sum(np.sin(2 * np.pi * 220.0 * h * n1 / sr1) / h for h in range(1, 9)) * np.exp(-n1 / (0.2 * sr1)) * 0.4
    n2 = np.arange(6400)
    s2 = sum(np.sin(2 * np.pi * 500.0 * h * n2 / sr2) / h for h in (1, 3, 5, 7)) * 0.5
    pool, hdrs, off = [], [], 0
    for name, s, rate, pitch, corr, loop in (('sine441', s0, sr0, 69, 0, (2000, 2000 + 50 * 300)),
                                             ('pluck', s1, sr1, 57, 5, (0, 0)),
                                             ('square', s2, sr2, 71, 0, (640, 640 + 64 * 50))):
        q = np.clip(np.round(s * 32767), -32768, 32767).astype('<i2')
        pool.append(q)
        pool.append(np.zeros(46, '<i2'))                       # the 46 zero frames the format asks for after a sample
        hdrs.append(struct.pack('<20sIIIIIBbHH', name.encode(), off, off + len(q), off + loop[0], off + loop[1], rate, pitch,
                                corr, 0, 1))
        off += len(q) + 46
    hdrs.append(struct.pack('<20sIIIIIBbHH', b'EOS', 0, 0, 0, 0, 0, 0, 0, 0, 0))
    smpl = np.concatenate(pool).tobytes()

    # instruments: list of zones, each a list of (generator, amount); the terminal generator comes last
    insts = [
        ('SineInst', [[('release', tc(0.5)), ('attenuation', 30)],
                      [('keyRange', (0, 72)), ('sampleModes', 1), ('sampleID', 0)],
                      [('keyRange', (73, 127)), ('fineTune', 10), ('attenuation', 60), ('sampleModes', 1), ('sampleID', 0)]]),
        ('PluckInst', [[('velRange', (0, 100)), ('sampleID', 1)],
                       [('velRange', (101, 127)), ('coarseTune', -12), ('sampleID', 1)]]),
        ('SquareInst', [[('rootKey', 72), ('attack', tc(0.02)), ('decay', tc(0.3)), ('sustain', 120), ('release', tc(0.2)),
                         ('sampleModes', 1), ('sampleID', 2)]]),
    ]
    presets = [
        ('Sine', 0, 0, [[('instrument', 0)]]),
        ('Pluck', 24, 0, [[('attenuation', 20), ('fineTune', -5), ('instrument', 1)]]),
        ('Layer', 40, 0, [[('coarseTune', 12)], [('keyRange', (40, 80)), ('instrument', 0)], [('instrument', 2)]]),
        ('Drums', 0, 128, [[('instrument', 1)]]),
    ]

    def pack(groups, hdr_fmt, hdr_of):
        hdr, bag, gen = b'', b'', b''
        nbag = ngen = 0
        for g in groups:
            hdr += hdr_of(g, nbag)
            for zone in g[-1]:
                bag += struct.pack('<HH', ngen, 0)
                for name, amount in zone:
                    gen += _gen(name, amount)
                    ngen += 1
                nbag += 1
        bag += struct.pack('<HH', ngen, 0)
        gen += struct.pack('<HH', 0, 0)
        return hdr, bag, gen, nbag
    ih, ibag, igen, nib = pack(insts, None, lambda g, nb: struct.pack('<20sH', g[0].encode(), nb))
    ih += struct.pack('<20sH', b'EOI', nib)
    ph, pbag, pgen, npb = pack(presets, None, lambda g, nb: struct.pack('<20sHHHIII', g[0].encode(), g[1], g[2], nb, 0, 0, 0))
    ph += struct.pack('<20sHHHIII', b'EOP', 0, 0, npb, 0, 0, 0)
    mod_end = struct.pack('<HHhHH', 0, 0, 0, 0, 0)
    pdta = b'pdta' + _chunk(b'phdr', ph) + _chunk(b'pbag', pbag) + _chunk(b'pmod', mod_end) + _chunk(b'pgen', pgen) + \
        _chunk(b'inst', ih) + _chunk(b'ibag', ibag) + _chunk(b'imod', mod_end) + _chunk(b'igen', igen) + \
        _chunk(b'shdr', b''.join(hdrs))
    info = b'INFO' + _chunk(b'ifil', struct.pack('<HH', 2, 4)) + _chunk(b'isng', b'EMU8000\0') + _chunk(b'INAM', b'amt test font\0')
    body = b'sfbk' + _chunk(b'LIST', info) + _chunk(b'LIST', b'sdta' + _chunk(b'smpl', smpl)) + _chunk(b'LIST', pdta)
    return b'RIFF' + struct.pack('<I', len(body)) + body
