"""FLAC reader / writer from the public FLAC format specification (SURVEY 8f,
next-row 3): stands where the reference uses librosa.load / soundfile.write
(util_audio.audio_from_file :962-964, audio_to_flac :966-968 -- 44.1 kHz mono
PCM-24).  Host-side file I/O, outside the hot path; pure Python + numpy.

Reader:
Supports: mono/independent channels, CONSTANT / VERBATIM / FIXED / LPC
subframes, Rice and Rice2 residual coding with escapes, wasted bits, all
block-size / sample-size header codes.  No stereo decorrelation is needed for
the fixtures but left/side, right/side and mid/side are handled for completeness.
Returns integer PCM; load_float scales by 2**-(bps-1) like libsndfile does.

Writer: VERBATIM subframes (valid FLAC, no compression), fixed block size 4096,
PCM-24 by default like audio_to_flac; STREAMINFO MD5 of the unencoded samples is written.

Integrity: the reader verifies every frame header's CRC-8, every frame's CRC-16 and the
STREAMINFO MD5 signature of the decoded samples (when the file carries one) and raises
ValueError on a mismatch -- byte work is proven bit-exact, not assumed.
"""
import hashlib

import numpy as np


class _Bits:
    def __init__(self, data, pos_bytes):
        self.data = data
        self.pos = pos_bytes * 8
        bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
        self.ones = np.flatnonzero(bits)

    def read(self, n):
        if n == 0:
            return 0
        p = self.pos
        b0 = p >> 3
        b1 = (p + n + 7) >> 3
        v = int.from_bytes(self.data[b0:b1], 'big')
        v >>= (b1 * 8 - (p + n))
        self.pos = p + n
        return v & ((1 << n) - 1)

    def read_signed(self, n):
        v = self.read(n)
        if n and v >> (n - 1):
            v -= 1 << n
        return v

    def unary(self):
        i = np.searchsorted(self.ones, self.pos)
        q = int(self.ones[i]) - self.pos
        self.pos += q + 1
        return q

    def align(self):
        self.pos = (self.pos + 7) & ~7


def _residual(br, blocksize, order, out):
    method = br.read(2)
    pbits = 4 if method == 0 else 5
    esc = (1 << pbits) - 1
    porder = br.read(4)
    nparts = 1 << porder
    idx = order
    for p in range(nparts):
        n = (blocksize >> porder) - (order if p == 0 else 0)
        k = br.read(pbits)
        if k == esc:
            nb = br.read(5)
            for _ in range(n):
                out[idx] = br.read_signed(nb)
                idx += 1
        else:
            for _ in range(n):
                q = br.unary()
                v = (q << k) | br.read(k)
                out[idx] = (v >> 1) ^ -(v & 1)
                idx += 1


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(br, blocksize, bps):
    if br.read(1):
        raise ValueError('subframe padding bit set')
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
    out = [0] * blocksize
    if typ == 0:
        out = [br.read_signed(bps)] * blocksize
    elif typ == 1:
        out = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= typ <= 12:
        order = typ - 8
        for i in range(order):
            out[i] = br.read_signed(bps)
        _residual(br, blocksize, order, out)
        co = _FIXED[order]
        for i in range(order, blocksize):
            out[i] += sum(c * out[i - j - 1] for j, c in enumerate(co))
    elif typ >= 32:
        order = (typ & 31) + 1
        for i in range(order):
            out[i] = br.read_signed(bps)
        prec = br.read(4) + 1
        shift = br.read_signed(5)
        co = [br.read_signed(prec) for _ in range(order)]
        _residual(br, blocksize, order, out)
        for i in range(order, blocksize):
            s = 0
            for j in range(order):
                s += co[j] * out[i - j - 1]
            out[i] += s >> shift
    else:
        raise ValueError('reserved subframe type %d' % typ)
    if wasted:
        out = [v << wasted for v in out]
    return out


def pcm_md5(pcm, bps):
    """MD5 of the unencoded audio as STREAMINFO defines it: samples interleaved by channel,
    little-endian, each sign-extended to a whole number of bytes."""
    nbytes = (bps + 7) // 8
    a = np.ascontiguousarray(np.asarray(pcm, dtype=np.int64).reshape(-1))
    raw = a.astype('<i8').view(np.uint8).reshape(-1, 8)[:, :nbytes]
    return hashlib.md5(np.ascontiguousarray(raw).tobytes()).digest()


def decode(path, verify=True):
    """Returns (pcm int64 [n_samples, channels], sample_rate, bits_per_sample).
    verify: check CRC-8 / CRC-16 of every frame and the STREAMINFO MD5 (if non-zero)."""
    data = open(path, 'rb').read()
    if data[:4] != b'fLaC':
        raise ValueError('not a FLAC file')
    pos = 4
    sr = ch = bps = total = None
    md5 = bytes(16)
    while True:
        hdr = data[pos]
        last, btype = hdr >> 7, hdr & 0x7F
        blen = int.from_bytes(data[pos + 1:pos + 4], 'big')
        body = data[pos + 4:pos + 4 + blen]
        if btype == 0:
            v = int.from_bytes(body[10:18], 'big')
            sr = v >> 44
            ch = ((v >> 41) & 7) + 1
            bps = ((v >> 36) & 31) + 1
            total = v & ((1 << 36) - 1)
            md5 = bytes(body[18:34])
        pos += 4 + blen
        if last:
            break
    br = _Bits(data, pos)
    chans = [[] for _ in range(ch)]
    got = 0
    while got < total:
        frame_start = br.pos >> 3
        sync = br.read(14)
        if sync != 0x3FFE:
            raise ValueError('lost sync at bit %d' % br.pos)
        br.read(1)
        br.read(1)                                   # blocking strategy
        bs_code = br.read(4)
        sr_code = br.read(4)
        ca = br.read(4)
        ss_code = br.read(3)
        br.read(1)
        first = br.read(8)                           # UTF-8 coded number
        n_extra = 0
        while first & (0x80 >> n_extra):
            n_extra += 1
        for _ in range(max(0, n_extra - 1)):
            br.read(8)
        if bs_code == 1:
            bs = 192
        elif 2 <= bs_code <= 5:
            bs = 576 << (bs_code - 2)
        elif bs_code == 6:
            bs = br.read(8) + 1
        elif bs_code == 7:
            bs = br.read(16) + 1
        else:
            bs = 256 << (bs_code - 8)
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        hdr_end = br.pos >> 3
        crc8 = br.read(8)                            # CRC-8 of the frame header
        if verify and crc8 != _crc8(data[frame_start:hdr_end]):
            raise ValueError('FLAC frame header CRC-8 mismatch at byte %d' % frame_start)
        fbps = {0: bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24}[ss_code]
        if ca < 8:
            subs = [_subframe(br, bs, fbps) for _ in range(ca + 1)]
        else:
            side_first = ca == 9
            b0 = fbps + (1 if side_first else 0)
            b1 = fbps + (0 if side_first else 1)
            a = np.array(_subframe(br, bs, b0), dtype=np.int64)
            b = np.array(_subframe(br, bs, b1), dtype=np.int64)
            if ca == 8:
                subs = [a, a - b]
            elif ca == 9:
                subs = [a + b, b]
            else:
                mid = (a << 1) | (b & 1)
                subs = [(mid + b) >> 1, (mid - b) >> 1]
        br.align()
        body_end = br.pos >> 3
        crc16 = br.read(16)                          # CRC-16 of the whole frame
        if verify and crc16 != _crc16(data[frame_start:body_end]):
            raise ValueError('FLAC frame CRC-16 mismatch at byte %d' % frame_start)
        for c in range(ch):
            chans[c].extend(subs[c])
        got += bs
    pcm = np.array(chans, dtype=np.int64).T[:total]
    if verify and md5 != bytes(16) and pcm_md5(pcm, bps) != md5:
        raise ValueError('FLAC STREAMINFO MD5 mismatch: decoded samples differ from the encoded audio')
    return pcm, sr, bps


def load_float(path):
    """libsndfile-style float read: pcm / 2**(bps-1); mono -> 1-D float64."""
    pcm, sr, bps = decode(path)
    y = pcm.astype(np.float64) / float(1 << (bps - 1))
    return (y[:, 0] if y.shape[1] == 1 else y), sr


# ------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------
def _crc8(data):
    crc = 0
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = ((crc << 1) ^ 0x07) & 0xFF if crc & 0x80 else (crc << 1) & 0xFF
    return crc


_CRC16_TAB = None


def _crc16(data):
    global _CRC16_TAB
    if _CRC16_TAB is None:
        tab = []
        for i in range(256):
            c = i << 8
            for _ in range(8):
                c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
            tab.append(c)
        _CRC16_TAB = tab
    crc = 0
    for b in data:
        crc = ((crc << 8) & 0xFFFF) ^ _CRC16_TAB[((crc >> 8) ^ b) & 0xFF]
    return crc


def _utf8_num(v):
    if v < 0x80:
        return bytes([v])
    out = []
    n = 0
    while v >= (0x40 >> n) and n < 5:
        out.append(0x80 | (v & 0x3F))
        v >>= 6
        n += 1
    lead = (0xFF << (6 - n)) & 0xFF
    out.append(lead | v)
    return bytes(reversed(out))


def encode(pcm, path, sr=44100, bps=24, blocksize=4096):
    """pcm: int array [n] (mono) or [n, channels], already quantised to `bps` bits."""
    pcm = np.asarray(pcm, dtype=np.int64)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    n, ch = pcm.shape
    ss_code = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps]
    frames = []
    sizes = []
    for fi, s0 in enumerate(range(0, n, blocksize)):
        blk = pcm[s0:s0 + blocksize]
        bs = blk.shape[0]
        hdr = bytearray()
        hdr += bytes([0xFF, 0xF8])                       # sync, fixed block size stream
        hdr += bytes([(7 << 4) | 0])                     # block size in 16 bits below; sample rate from STREAMINFO
        hdr += bytes([((ch - 1) << 4) | (ss_code << 1)])
        hdr += _utf8_num(fi)
        hdr += (bs - 1).to_bytes(2, 'big')
        hdr += bytes([_crc8(hdr)])
        # subframes: VERBATIM (type 000001), samples as bps-bit two's complement, MSB first
        acc, nbits = 0, 0
        body = bytearray()
        mask = (1 << bps) - 1
        for c in range(ch):
            acc = (acc << 8) | 0x02; nbits += 8          # 0 | 000001 | 0
            vals = blk[:, c] & mask
            for v in vals.tolist():
                acc = (acc << bps) | v
                nbits += bps
                while nbits >= 8:
                    nbits -= 8
                    body.append((acc >> nbits) & 0xFF)
                acc &= (1 << nbits) - 1
        if nbits:
            body.append((acc << (8 - nbits)) & 0xFF)
        fr = bytes(hdr) + bytes(body)
        fr += _crc16(fr).to_bytes(2, 'big')
        frames.append(fr)
        sizes.append(len(fr))
    si = bytearray()
    si += blocksize.to_bytes(2, 'big') * 2
    si += (min(sizes) if sizes else 0).to_bytes(3, 'big') + (max(sizes) if sizes else 0).to_bytes(3, 'big')
    v = (sr << 44) | ((ch - 1) << 41) | ((bps - 1) << 36) | n
    si += v.to_bytes(8, 'big') + pcm_md5(pcm, bps)
    with open(path, 'wb') as f:
        f.write(b'fLaC' + bytes([0x80]) + len(si).to_bytes(3, 'big') + bytes(si))
        for fr in frames:
            f.write(fr)


def save_float(waveform, path, sr=44100, bps=24):
    """soundfile.write(..., subtype='PCM_24') semantics: scale by 2**(bps-1), round, clip."""
    y = np.asarray(waveform, dtype=np.float64)
    q = np.clip(np.rint(y * float(1 << (bps - 1))), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)
    encode(q, path, sr=sr, bps=bps)
