"""SoundFont 2 sample playback for the guess synthesis ("optionally SF2 sample playback if a
soundfont is supplied by the user").

The reference renders every guessed note with fluidsynth through a General MIDI soundfont
(/root/reference/util_audio.py:758-786, :819-936; main.py:25-29 ``-soundfont_path``).  Neither fluidsynth nor a
soundfont exists in this image, so this module is a from-scratch reader of the SoundFont 2.04 container (RIFF
``sfbk``: sample pool ``smpl`` + the nine ``pdta`` sub-chunks) and a statement of what is played:

    for every (preset zone x instrument zone) of the note's program (bank 0) whose key and velocity ranges hold the note:
        cents  = (pitch - root) * scaleTuning + 100 * coarseTune + fineTune + pitchCorrection
        pos(n) = n * 2**(cents / 1200) * sampleRate / sr        sample frames after n output samples; looped zones
                 (sampleModes 1 / 3) wrap [startloop, endloop) once pos reaches endloop; others end at `end`
        value  = linear interpolation of the 16-bit sample at pos
        gain   = 10**(-initialAttenuation / 200)               centibels
        env    = delay -> linear attack -> hold -> decay at 100 dB per decayVolEnv down to sustainVolEnv -> at note
                 off release at 100 dB per releaseVolEnv (all volume-envelope generators, timecents)
    note = (velocity / 128)**4 * sum of zones, cut 1 s after note off (:876); windows are scaled as render() does
    (:778-781), exactly like the additive synthesiser (amt_saga/synth.py).

Generators combine as the specification says: an instrument zone takes its own value, else the instrument's global
zone's, else the default; preset-level values ADD to that; key / velocity ranges intersect.  Modulators, the filter,
the modulation envelope / LFOs, chorus, reverb and stereo panning are not modelled (mono mix): this is sample
playback, not a fluidsynth clone -- parity with fluidsynth is unpinned, as for the additive timbres.

The renderer is the HIP kernel amt_sf2_synth_windows (csrc/amt_synth.hip); its float64 checker is oracle/sf2.py.
"""
import struct

import numpy as np
import torch

# generator numbers (SoundFont 2.04, section 8.1.2)
G_START, G_END, G_STARTLOOP, G_ENDLOOP, G_START_COARSE = 0, 1, 2, 3, 4
G_END_COARSE = 12
G_DELAY, G_ATTACK, G_HOLD, G_DECAY, G_SUSTAIN, G_RELEASE = 33, 34, 35, 36, 37, 38
G_INSTRUMENT, G_KEYRANGE, G_VELRANGE, G_STARTLOOP_COARSE = 41, 43, 44, 45
G_ATTENUATION, G_ENDLOOP_COARSE, G_COARSE_TUNE, G_FINE_TUNE, G_SAMPLE_ID, G_SAMPLE_MODES = 48, 50, 51, 52, 53, 54
G_SCALE_TUNING, G_ROOT_KEY = 56, 58
DEFAULTS = {G_DELAY: -12000, G_ATTACK: -12000, G_HOLD: -12000, G_DECAY: -12000, G_RELEASE: -12000,
            G_SUSTAIN: 0, G_ATTENUATION: 0, G_COARSE_TUNE: 0, G_FINE_TUNE: 0, G_SAMPLE_MODES: 0,
            G_SCALE_TUNING: 100, G_ROOT_KEY: -1}
# generators a preset zone may add to the instrument's value
PRESET_ADDS = (G_DELAY, G_ATTACK, G_HOLD, G_DECAY, G_SUSTAIN, G_RELEASE, G_ATTENUATION, G_COARSE_TUNE, G_FINE_TUNE,
               G_SCALE_TUNING)
ZONE_FIELDS = 20            # floats per row of the device zone table (layout: zone_row)


def _chunks(buf, lo, hi):
    """(id, payload offset, size) of the RIFF chunks in buf[lo:hi]."""
    out = []
    while lo + 8 <= hi:
        cid = bytes(buf[lo:lo + 4])
        size = struct.unpack_from('<I', buf, lo + 4)[0]
        if lo + 8 + size > hi:
            raise ValueError('SF2: chunk %r runs past its parent' % cid)
        out.append((cid, lo + 8, size))
        lo += 8 + size + (size & 1)
    return out


def _zones(bags, gens, b0, b1, terminal):
    """Generator dicts of bags b0..b1-1: (global zone dict or {}, [zone dicts that end in `terminal`])."""
    glob, zones = {}, []
    for b in range(b0, b1):
        g0, g1 = bags[b][0], bags[b + 1][0]
        z = {}
        for oper, raw in gens[g0:g1]:
            if oper in (G_KEYRANGE, G_VELRANGE):
                z[oper] = (raw & 0xFF, (raw >> 8) & 0xFF)
            elif oper in (G_INSTRUMENT, G_SAMPLE_ID, G_SAMPLE_MODES):
                z[oper] = raw
            else:
                z[oper] = raw - 0x10000 if raw >= 0x8000 else raw        # signed 16-bit amounts
        if terminal in z:
            zones.append(z)
        elif b == b0 and not zones:
            glob = z                                                      # a first zone without a terminal: the global zone
    return glob, zones


class SoundFont:
    """Parsed SoundFont: ``samples`` (float32 in [-1, 1)), ``headers`` (shdr records) and ``programs``: bank-0 MIDI
    program -> list of flattened playback zones (dicts, see zone_row)."""

    def __init__(self, path_or_bytes):
        buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
        if len(buf) < 12 or buf[:4] != b'RIFF' or buf[8:12] != b'sfbk':
            raise ValueError('not a SoundFont 2 file (RIFF sfbk header missing)')
        riff_end = 8 + struct.unpack_from('<I', buf, 4)[0]
        if riff_end > len(buf):
            raise ValueError('SF2: RIFF size exceeds the file')
        smpl, pd = None, {}
        for cid, off, size in _chunks(buf, 12, riff_end):
            if cid != b'LIST':
                continue
            kind = bytes(buf[off:off + 4])
            for sid, soff, ssize in _chunks(buf, off + 4, off + size):
                if kind == b'sdta' and sid == b'smpl':
                    smpl = np.frombuffer(buf, dtype='<i2', count=ssize // 2, offset=soff)
                elif kind == b'pdta':
                    pd[sid] = (soff, ssize)
        need = (b'phdr', b'pbag', b'pgen', b'inst', b'ibag', b'igen', b'shdr')
        if smpl is None or any(k not in pd for k in need):
            raise ValueError('SF2: sample pool or a pdta sub-chunk is missing')

        def recs(key, fmt):
            off, size = pd[key]
            n = struct.calcsize(fmt)
            if size % n:
                raise ValueError('SF2: %r is not a whole number of records' % key)
            return [struct.unpack_from(fmt, buf, off + i * n) for i in range(size // n)]
        phdr = recs(b'phdr', '<20sHHHIII')
        pbag = recs(b'pbag', '<HH')
        pgen = recs(b'pgen', '<HH')
        inst = recs(b'inst', '<20sH')
        ibag = recs(b'ibag', '<HH')
        igen = recs(b'igen', '<HH')
        shdr = recs(b'shdr', '<20sIIIIIBbHH')
        self.samples = (smpl.astype(np.float32) / 32768.0)
        self.headers = [dict(name=h[0].split(b'\0')[0].decode('latin-1'), start=h[1], end=h[2], startloop=h[3], endloop=h[4],
                             rate=h[5], pitch=h[6], correction=h[7]) for h in shdr[:-1]]          # last record = EOS
        self.names = {}
        self.programs = {}
        for i in range(len(phdr) - 1):                                    # last record = EOP
            name, prog, bank, bag0 = phdr[i][0], phdr[i][1], phdr[i][2], phdr[i][3]
            if bank != 0:
                continue
            pglob, pzones = _zones(pbag, pgen, bag0, phdr[i + 1][3], G_INSTRUMENT)
            flat = []
            for pz in pzones:
                k = pz[G_INSTRUMENT]
                if k >= len(inst) - 1:
                    raise ValueError('SF2: preset %d points at instrument %d' % (prog, k))
                iglob, izones = _zones(ibag, igen, inst[k][1], inst[k + 1][1], G_SAMPLE_ID)
                for iz in izones:
                    z = self._flatten(pglob, pz, iglob, iz)
                    if z is not None:
                        flat.append(z)
            self.programs[prog] = flat
            self.names[prog] = name.split(b'\0')[0].decode('latin-1')

    def _flatten(self, pglob, pz, iglob, iz):
        def ival(g):
            return iz.get(g, iglob.get(g, DEFAULTS.get(g, 0)))

        def pval(g):
            return pz.get(g, pglob.get(g, 0))
        ik, iv = iz.get(G_KEYRANGE, iglob.get(G_KEYRANGE, (0, 127))), iz.get(G_VELRANGE, iglob.get(G_VELRANGE, (0, 127)))
        pk, pv = pz.get(G_KEYRANGE, pglob.get(G_KEYRANGE, (0, 127))), pz.get(G_VELRANGE, pglob.get(G_VELRANGE, (0, 127)))
        klo, khi = max(ik[0], pk[0]), min(ik[1], pk[1])
        vlo, vhi = max(iv[0], pv[0]), min(iv[1], pv[1])
        if klo > khi or vlo > vhi:
            return None
        sid = iz[G_SAMPLE_ID]
        if sid >= len(self.headers):
            raise ValueError('SF2: zone points at sample %d' % sid)
        h = self.headers[sid]
        v = {g: ival(g) + (pval(g) if g in PRESET_ADDS else 0) for g in DEFAULTS}
        start = h['start'] + ival(G_START) + 32768 * ival(G_START_COARSE)
        end = h['end'] + ival(G_END) + 32768 * ival(G_END_COARSE)
        ls = h['startloop'] + ival(G_STARTLOOP) + 32768 * ival(G_STARTLOOP_COARSE)
        le = h['endloop'] + ival(G_ENDLOOP) + 32768 * ival(G_ENDLOOP_COARSE)
        if not (0 <= start < end <= len(self.samples)):
            raise ValueError('SF2: sample %d lies outside the pool' % sid)
        mode = v[G_SAMPLE_MODES] & 3
        loop = mode in (1, 3) and start <= ls < le <= end
        root = v[G_ROOT_KEY] if v[G_ROOT_KEY] >= 0 else h['pitch']
        if root > 127:
            root = 60                                                     # "unpitched" (255): conventional middle C
        tc = lambda g: float(2.0 ** (max(-12000, min(v[g], 8000)) / 1200.0))
        return dict(key=(klo, khi), vel=(vlo, vhi), start=start, end=end, loop_start=ls, loop_end=le, loop=bool(loop),
                    rate=float(h['rate']), root=int(root), scale=float(v[G_SCALE_TUNING]),
                    tune=float(100 * v[G_COARSE_TUNE] + v[G_FINE_TUNE] + h['correction']),
                    gain=float(10.0 ** (-max(0, min(v[G_ATTENUATION], 1440)) / 200.0)),
                    delay=tc(G_DELAY), attack=tc(G_ATTACK), hold=tc(G_HOLD), decay=tc(G_DECAY), release=tc(G_RELEASE),
                    sustain_db=float(max(0, min(v[G_SUSTAIN], 1440)) / 10.0), sample=sid)

    @staticmethod
    def zone_row(z):
        """The float32 [ZONE_FIELDS] row the kernel reads."""
        return [z['key'][0], z['key'][1], z['vel'][0], z['vel'][1], z['start'], z['end'], z['loop_start'], z['loop_end'],
                1.0 if z['loop'] else 0.0, z['rate'], z['root'], z['scale'], z['tune'], z['gain'], z['delay'], z['attack'],
                z['hold'], z['decay'], z['sustain_db'], z['release']]

    def tables(self, n_prog=128):
        """(zones float32 [nz, ZONE_FIELDS], first int32 [n_prog + 1]): program p plays zones first[p] .. first[p+1]-1.
        Sample offsets ride in float32: pools beyond 2**24 frames (16.7 M, 33 MB of 16-bit audio) are refused."""
        if len(self.samples) >= 1 << 24:
            raise ValueError('SF2: sample pool too large for the float32 zone table (>= 2**24 frames)')
        rows, first = [], [0]
        for p in range(n_prog):
            rows += [self.zone_row(z) for z in self.programs.get(p, [])]
            first.append(len(rows))
        z = np.array(rows, dtype=np.float32).reshape(-1, ZONE_FIELDS)
        return z, np.array(first, dtype=np.int32)

    def device(self, n_prog=128):
        """Device-resident (samples, zones, first) for render_windows_device; cached."""
        if getattr(self, '_dev', None) is None or self._dev[3] != n_prog:
            from .device import to_dev
            z, first = self.tables(n_prog)
            if z.shape[0] == 0:
                raise ValueError('SF2: no bank-0 preset with playable zones')
            self._dev = (to_dev(self.samples), to_dev(z), to_dev(first, torch.int32), n_prog)
        return self._dev[:3]


def render_windows_device(notes, L, soundfont, sr=44100, out=None):
    """Sample-playback counterpart of synth.render_windows_device: notes rows {MIDI program, pitch, velocity, onset_s,
    dur_s} (list of note lists or a device float32 [B, M, 5] tensor) -> device float32 [B, L]."""
    from . import _lib, synth
    from .device import empty, ptr, stream_ptr, to_dev
    lib = _lib.load()
    nt = notes if isinstance(notes, torch.Tensor) else to_dev(synth.notes_tensor(notes))
    if nt.dim() == 2:
        nt = nt[:, None, :]
    assert nt.dtype == torch.float32 and nt.shape[-1] == 5 and nt.is_contiguous()
    samples, zones, first = soundfont.device()
    B, M = nt.shape[0], nt.shape[1]
    wave = out if out is not None else empty((B, int(L)))
    peak = empty((B,))
    _lib.check(lib.amt_sf2_synth_windows(ptr(nt), M, B, int(L), float(sr), ptr(samples), int(samples.shape[0]),
                                         ptr(zones), int(zones.shape[0]), ptr(first), int(first.shape[0]) - 1,
                                         ptr(wave), wave.stride(0), ptr(peak), stream_ptr()))
    return wave
