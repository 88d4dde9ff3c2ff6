"""Minimal HDF5 reader (pure Python + numpy) for Keras ``save_weights`` files.

The reference checkpoints its heads with ``model.save_weights('checkpoint_<name>_<batch>.h5')``
(RDCNN.py:490-494) and resumes with ``load_weights`` (RDCNN.py:778-782, training.py:104-138); h5py
/ libhdf5 are not part of this stack, so the subset of the HDF5 file format those files use is
read here from the public format specification (HDF5 File Format Specification v1.1 / 2.0):

  * superblock versions 0 and 1 (what h5py writes by default, ``libver='earliest'``);
  * version-1 object headers with continuation blocks;
  * old-style groups: symbol-table message -> v1 B-tree (node type 0) -> symbol-table nodes (SNOD)
    -> names in a local heap;
  * datasets with contiguous or compact layout (Keras weights are never chunked or filtered);
    chunked layout without filters is read too (v1 B-tree, node type 1);
  * datatypes: IEEE floats, fixed-point integers, fixed-length strings (numpy 'S' attributes such as
    ``layer_names`` / ``weight_names``), variable-length strings through the global heap;
  * attributes (message versions 1-3) stored in the object header.

Anything else (superblock 2/3 "latest" files, dense link/attribute storage, filters) raises
``ValueError`` with the name of the unsupported feature.  Host-side file I/O, outside the hot path.
"""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'


class _Reader:
    def __init__(self, data):
        self.d = data
        self.O = 8          # size of offsets
        self.L = 8          # size of lengths

    def u(self, pos, n):
        return int.from_bytes(self.d[pos:pos + n], 'little')

    def off(self, pos):
        return self.u(pos, self.O)

    def length(self, pos):
        return self.u(pos, self.L)

    def undefined(self, addr):
        return addr == (1 << (8 * self.O)) - 1


class _Datatype:
    def __init__(self, r, pos):
        cv = r.d[pos]
        self.cls, self.version = cv & 0x0F, cv >> 4
        self.bits = r.d[pos + 1:pos + 4]
        self.size = r.u(pos + 4, 4)
        self.vlen_string = False
        p = pos + 8
        if self.cls == 0:                                    # fixed point
            signed = bool(self.bits[0] & 0x08)
            order = '>' if self.bits[0] & 1 else '<'
            self.dtype = np.dtype('%s%s%d' % (order, 'i' if signed else 'u', self.size))
        elif self.cls == 1:                                  # floating point
            order = '>' if self.bits[0] & 1 else '<'
            if self.size not in (2, 4, 8):
                raise ValueError('HDF5: unsupported float size %d' % self.size)
            self.dtype = np.dtype('%sf%d' % (order, self.size))
        elif self.cls == 3:                                  # fixed-length string
            self.dtype = np.dtype('S%d' % self.size)
        elif self.cls == 9:                                  # variable length
            if (self.bits[0] & 0x0F) != 1:
                raise ValueError('HDF5: variable-length sequences are not supported (only strings)')
            self.vlen_string = True
            self.dtype = None
        else:
            raise ValueError('HDF5: unsupported datatype class %d' % self.cls)
        del p


def _dataspace(r, pos):
    ver, rank, flags = r.d[pos], r.d[pos + 1], r.d[pos + 2]
    if ver == 1:
        p = pos + 8
    elif ver == 2:
        if r.d[pos + 3] == 2:                                # null dataspace
            return None
        p = pos + 4
    else:
        raise ValueError('HDF5: unsupported dataspace version %d' % ver)
    return tuple(r.length(p + i * r.L) for i in range(rank))


class File:
    """``File(path)`` -- read-only.  ``f.keys('/grp')``, ``f.attrs('/grp')`` (dict),
    ``f.dataset('/grp/name')`` (numpy array), ``f.visit()`` (all dataset paths)."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            data = fh.read()
        self.r = r = _Reader(data)
        base = None
        for start in (0, 512, 1024, 2048):
            if data[start:start + 8] == SIGNATURE:
                base = start
                break
        if base is None:
            raise ValueError('not an HDF5 file')
        ver = data[base + 8]
        if ver in (0, 1):
            r.O, r.L = data[base + 13], data[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = r.off(p)
            p += 4 * r.O                                     # base, free-space, end-of-file, driver info
            # root group symbol table entry
            self.root = r.off(p + r.O)
        elif ver in (2, 3):
            raise ValueError('HDF5: superblock version %d ("latest" format) is not supported; re-save the '
                             'weights with the default h5py settings' % ver)
        else:
            raise ValueError('HDF5: unknown superblock version %d' % ver)
        self._objs = {}

    # ---- object headers ------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, pos, size)] of a version-1 object header at `addr` (absolute positions of message data)."""
        r = self.r
        addr += self.base
        if r.d[addr:addr + 4] == b'OHDR':
            raise ValueError('HDF5: version-2 object headers ("latest" format) are not supported')
        if r.d[addr] != 1:
            raise ValueError('HDF5: bad object header version %d' % r.d[addr])
        nmsg = r.u(addr + 2, 2)
        size = r.u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = r.u(p, 2), r.u(p + 2, 2)
                body = p + 8
                if mtype == 0x0010:                          # continuation
                    blocks.append((r.off(body) + self.base, r.length(body + r.O)))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    def _group_entries(self, addr):
        """{name: object header address} of an old-style group."""
        r = self.r
        for mtype, p, _ in self._messages(addr):
            if mtype == 0x0011:
                btree, heap = r.off(p), r.off(p + r.O)
                break
            if mtype in (0x0002, 0x0006):
                raise ValueError('HDF5: new-style groups (link messages) are not supported')
        else:
            raise ValueError('HDF5: object is not a group')
        h = heap + self.base
        if r.d[h:h + 4] != b'HEAP':
            raise ValueError('HDF5: bad local heap signature')
        heap_data = r.off(h + 8 + 2 * r.L) + self.base
        out = {}

        def name_at(o):
            e = r.d.index(b'\x00', heap_data + o)
            return r.d[heap_data + o:e].decode('utf8')

        def walk(node):
            n = node + self.base
            if r.d[n:n + 4] == b'SNOD':
                cnt = r.u(n + 6, 2)
                p = n + 8
                for _ in range(cnt):
                    out[name_at(r.off(p))] = r.off(p + r.O)
                    p += 2 * r.O + 24
                return
            if r.d[n:n + 4] != b'TREE' or r.d[n + 4] != 0:
                raise ValueError('HDF5: bad group B-tree node')
            used = r.u(n + 6, 2)
            p = n + 8 + 2 * r.O                              # skip siblings
            p += r.L                                         # key 0
            for _ in range(used):
                walk(r.off(p))
                p += r.O + r.L
        walk(btree)
        return out

    def _resolve(self, path):
        addr = self.root
        for part in [q for q in path.split('/') if q]:
            ent = self._group_entries(addr)
            if part not in ent:
                raise KeyError(path)
            addr = ent[part]
        return addr

    def is_group(self, path):
        return any(m[0] == 0x0011 for m in self._messages(self._resolve(path)))

    def keys(self, path='/'):
        return sorted(self._group_entries(self._resolve(path)))

    # ---- data ---------------------------------------------------------------------------
    def _vlen_string(self, p):
        r = self.r
        n, heap, index = r.u(p, 4), r.off(p + 4), r.u(p + 4 + r.O, 4)
        h = heap + self.base
        if r.d[h:h + 4] != b'GCOL':
            raise ValueError('HDF5: bad global heap collection')
        size = r.length(h + 8)
        q = h + 8 + r.L
        while q < h + size:
            idx, osize = r.u(q, 2), r.length(q + 8)
            if idx == index:
                return r.d[q + 8 + r.L:q + 8 + r.L + n]
            if idx == 0:
                break
            q += 8 + r.L + ((osize + 7) & ~7)
        raise ValueError('HDF5: global heap object %d not found' % index)

    def _decode(self, dt, shape, pos):
        r = self.r
        count = int(np.prod(shape)) if shape is not None else 0
        if shape is None:
            return np.zeros((0,), np.float64)
        if dt.vlen_string:
            step = 4 + r.O + 4
            vals = [self._vlen_string(pos + i * step) for i in range(count)]
            a = np.array(vals, dtype=object).reshape(shape)
            return a[()] if shape == () else a
        a = np.frombuffer(r.d, dtype=dt.dtype, count=count, offset=pos).reshape(shape)
        return a[()] if shape == () else np.array(a)

    def attrs(self, path='/'):
        r = self.r
        out = {}
        for mtype, p, _ in self._messages(self._resolve(path)):
            if mtype == 0x0015:
                raise ValueError('HDF5: dense attribute storage is not supported')
            if mtype != 0x000C:
                continue
            ver = r.d[p]
            nsz, tsz, ssz = r.u(p + 2, 2), r.u(p + 4, 2), r.u(p + 6, 2)
            q = p + 8 + (1 if ver == 3 else 0)
            pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
            name = r.d[q:q + nsz].split(b'\x00')[0].decode('utf8')
            q += pad(nsz)
            dt = _Datatype(r, q)
            q += pad(tsz)
            shape = _dataspace(r, q)
            q += pad(ssz)
            out[name] = self._decode(dt, shape, q)
        return out

    def dataset(self, path):
        r = self.r
        dt = shape = layout = None
        for mtype, p, _ in self._messages(self._resolve(path)):
            if mtype == 0x0001:
                shape = _dataspace(r, p)
            elif mtype == 0x0003:
                dt = _Datatype(r, p)
            elif mtype == 0x0008:
                layout = p
            elif mtype == 0x000B:
                if r.u(p + 1, 1):
                    raise ValueError('HDF5: filtered (compressed) datasets are not supported')
        if dt is None or layout is None:
            raise ValueError('HDF5: %s is not a dataset' % path)
        ver = r.d[layout]
        if ver == 3:
            cls = r.d[layout + 1]
            if cls == 0:                                     # compact
                return self._decode(dt, shape, layout + 4)
            if cls == 1:                                     # contiguous
                addr = r.off(layout + 2)
                if r.undefined(addr):
                    return np.zeros(shape, dt.dtype)
                return self._decode(dt, shape, addr + self.base)
            if cls == 2:
                return self._chunked(dt, shape, layout)
            raise ValueError('HDF5: unsupported data layout class %d' % cls)
        if ver in (1, 2):
            rank, cls = r.d[layout + 1], r.d[layout + 2]
            if cls == 1:
                return self._decode(dt, shape, r.off(layout + 8) + self.base)
            if cls == 0:
                return self._decode(dt, shape, layout + 8 + 4 * rank + 4)
        raise ValueError('HDF5: unsupported data layout version %d' % ver)

    def _chunked(self, dt, shape, layout):
        r = self.r
        rank1 = r.d[layout + 2]                              # dataset rank + 1
        btree = r.off(layout + 3)
        cdims = [r.u(layout + 3 + r.O + 4 * i, 4) for i in range(rank1)]
        out = np.zeros(shape, dt.dtype)
        if r.undefined(btree):
            return out
        csh = tuple(cdims[:-1])

        def walk(node):
            n = node + self.base
            if r.d[n:n + 4] != b'TREE' or r.d[n + 4] != 1:
                raise ValueError('HDF5: bad chunk B-tree node')
            level, used = r.d[n + 5], r.u(n + 6, 2)
            p = n + 8 + 2 * r.O
            ksz = 8 + 8 * rank1
            for _ in range(used):
                csize, fmask = r.u(p, 4), r.u(p + 4, 4)
                offs = [r.u(p + 8 + 8 * i, 8) for i in range(rank1 - 1)]
                child = r.off(p + ksz)
                if level:
                    walk(child)
                else:
                    if fmask:
                        raise ValueError('HDF5: filtered chunks are not supported')
                    blk = np.frombuffer(r.d, dtype=dt.dtype, count=int(np.prod(csh)), offset=child + self.base).reshape(csh)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, csh, shape))
                    out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
                    del csize
                p += ksz + r.O
        walk(btree)
        return out

    def visit(self, path='/'):
        """All dataset paths below `path`, depth first."""
        out = []
        for k in self.keys(path):
            q = (path.rstrip('/') + '/' + k)
            if self.is_group(q):
                out += self.visit(q)
            else:
                out.append(q)
        return out


# =====================================================================================================
# Writer: the subset above, enough for Keras ``load_weights`` / h5py / libhdf5 to open the result
# =====================================================================================================
UNDEF = 0xFFFFFFFFFFFFFFFF
_LEAF_K, _INTERNAL_K = 64, 16            # symbol-table node holds 2 * 64 names, one B-tree node 2 * 16 of those


def _pad8(b):
    return b + bytes((-len(b)) % 8)


def _dt_float(size):
    if size == 4:
        props = struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
        return struct.pack('<BBBBI', 0x11, 0x20, 31, 0, 4) + props
    props = struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
    return struct.pack('<BBBBI', 0x11, 0x20, 63, 0, 8) + props


def _dt_string(n):
    return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, n)        # fixed length, null padded, ASCII


def _w_dataspace(shape):
    if shape is None:                                         # scalar
        return struct.pack('<BBB5x', 1, 0, 0)
    return struct.pack('<BBB5x', 1, len(shape), 0) + b''.join(struct.pack('<Q', int(d)) for d in shape)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _attribute(name, value):
    """Attribute message (version 1) for bytes (scalar fixed string), a list of bytes (1-D fixed strings)
    or an empty list (Keras writes np.asarray([]) -> float64, shape (0,))."""
    nm = name.encode('utf8') + b'\x00'
    if isinstance(value, bytes):
        dt, sp, data = _dt_string(max(len(value), 1)), _w_dataspace(None), value or b'\x00'
    elif len(value) == 0:
        dt, sp, data = _dt_float(8), _w_dataspace((0,)), b''
    else:
        w = max(len(v) for v in value)
        dt, sp = _dt_string(w), _w_dataspace((len(value),))
        data = b''.join(v.ljust(w, b'\x00') for v in value)
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + data
    if len(body) > 64000:
        raise ValueError('HDF5: attribute %s exceeds the object-header message limit' % name)
    return _message(0x000C, body)


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BBHII4x', 1, 0, len(messages), 1, len(body)) + body


class Writer:
    """Builds an HDF5 file in memory: ``w = Writer(); w.group('/a', attrs={...}); w.dataset('/a/a/k:0', array);
    w.save(path)``.  Parent groups are created on demand.  float32 / float64 datasets, contiguous layout."""

    def __init__(self):
        self.groups = {'/': {}}                    # path -> attrs (insertion order irrelevant: names are sorted on disk)
        self.datasets = {}                         # path -> ndarray

    def group(self, path, attrs=None):
        path = '/' + path.strip('/')
        parts = [p for p in path.split('/') if p]
        for i in range(len(parts)):
            self.groups.setdefault('/' + '/'.join(parts[:i + 1]), {})
        self.groups[path if parts else '/'].update(attrs or {})

    def attrs(self, path, attrs):
        self.group(path, attrs)

    def dataset(self, path, array):
        path = '/' + path.strip('/')
        parent = path.rsplit('/', 1)[0] or '/'
        self.group(parent)
        a = np.ascontiguousarray(array)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float32)
        self.datasets[path] = a

    def save(self, path):
        out = bytearray(96)                        # superblock, patched at the end
        addr = {}

        def alloc(b):
            a = len(out)
            out.extend(_pad8(b))
            return a

        # datasets first (raw data, then headers)
        for p, a in self.datasets.items():
            raw = alloc(a.astype('<f%d' % a.dtype.itemsize).tobytes()) if a.size else UNDEF
            msgs = [_message(0x0001, _w_dataspace(a.shape)), _message(0x0003, _dt_float(a.dtype.itemsize), 1),
                    _message(0x0005, bytes([2, 2, 2, 1, 0, 0, 0, 0])),
                    _message(0x0008, struct.pack('<BBQQ', 3, 1, raw, a.size * a.dtype.itemsize))]
            addr[p] = alloc(_object_header(msgs))
        # groups bottom-up (children's addresses are needed by the parent's symbol table)
        for g in sorted(self.groups, key=lambda s: -s.count('/') if s != '/' else 1):
            prefix = '' if g == '/' else g
            kids = sorted({p[len(prefix) + 1:] for p in list(self.groups) + list(self.datasets)
                           if p != g and p.startswith(prefix + '/') and '/' not in p[len(prefix) + 1:]},
                          key=lambda s: s.encode('utf8'))
            if len(kids) > 2 * _LEAF_K * 2 * _INTERNAL_K:
                raise ValueError('HDF5: too many links in one group')
            heap = bytearray(8)                    # offset 0: the empty name
            offs = {}
            for k in kids:
                offs[k] = len(heap)
                heap += _pad8(k.encode('utf8') + b'\x00')
            free_off = len(heap)
            heap += struct.pack('<QQ', 1, 32) + bytes(16)          # one free block closes the heap
            heap_data = alloc(bytes(heap))
            heap_addr = alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), free_off, heap_data))
            snods, keys = [], [0]
            for i in range(0, len(kids), 2 * _LEAF_K):
                chunk = kids[i:i + 2 * _LEAF_K]
                body = b'SNOD' + struct.pack('<BBH', 1, 0, len(chunk))
                for k in chunk:
                    body += struct.pack('<QQII16x', offs[k], addr[(prefix + '/' + k)], 0, 0)
                body += bytes(40 * (2 * _LEAF_K - len(chunk)))
                snods.append(alloc(body))
                keys.append(offs[chunk[-1]])
            node = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(snods), UNDEF, UNDEF)
            for i in range(2 * _INTERNAL_K):
                node += struct.pack('<Q', keys[i] if i < len(keys) else 0)
                node += struct.pack('<Q', snods[i] if i < len(snods) else 0)
            node += struct.pack('<Q', keys[len(snods)] if len(snods) < len(keys) and len(snods) == 2 * _INTERNAL_K else 0)
            btree = alloc(node)
            msgs = [_message(0x0011, struct.pack('<QQ', btree, heap_addr))]
            msgs += [_attribute(k, v) for k, v in self.groups[g].items()]
            addr[g] = alloc(_object_header(msgs))
            if g == '/':
                root = (addr[g], btree, heap_addr)
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack('<HHI', _LEAF_K, _INTERNAL_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, len(out), UNDEF)
        sb += struct.pack('<QQII', 0, root[0], 1, 0) + struct.pack('<QQ', root[1], root[2])
        out[:96] = sb
        with open(path, 'wb') as fh:
            fh.write(bytes(out))
