"""Minimal FAST5 (HDF5) writer -- synthetic inputs for the tests and the bench on boxes without
h5py, and the way to turn a read bundle back into FAST5 files.

Writes the classic on-disk format (superblock 0, version-1 object headers, symbol-table groups
with v1 B-trees and local heaps, version-1 attribute messages, contiguous or chunked datasets
with a v1 chunk B-tree): what h5py / the HDF5 library write by default (`libver earliest`) and
what every HDF5 release reads.  tests/test_fast5_native.py opens these files with the real
HDF5 library (h5py under the image's python3.9) to make sure they ARE HDF5, and with
csrc/pxg_h5.cpp, the reader this build ships.

    with Fast5Writer(path) as f5:                       # multi-read layout
        f5.add_read(read_id, raw_int16, calib_row, start_time=..., channel_number=...,
                    run_id=..., sample_id=..., basecall=dict_or_None, compression='gzip')
    write_single_read(path, ...)                        # single-read layout

compression: None (contiguous int16), 'gzip' (chunked, deflate), 'vbz' (chunked, ONT filter
32020 version 1: zstd over 16-bit streamvbyte of zig-zag deltas; needs libzstd.so.1).
Format facts from the HDF5 File Format Specification (public); no reference code involved.
"""
import ctypes as C
import struct
import zlib

import numpy as np

__all__ = ['Fast5Writer', 'write_single_read', 'vbz_encode', 'H5Writer']

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K, CHUNK_K = 4, 16, 32


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


# ---- datatype / dataspace / attribute messages -------------------------------------------
def _dt_int(size, signed):
    return struct.pack('<BBBBIHH', 0x10, 0x08 if signed else 0, 0, 0, size, 0, 8 * size)


def _dt_float(size):
    if size == 8:
        return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 0x3F, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 0x1F, 0, 4, 0, 32, 23, 8, 0, 23, 127)


def _dt_string(size):
    return struct.pack('<BBBBI', 0x13, 0, 0, 0, size)          # null-terminated ASCII


def _dt_bytes(size):
    return struct.pack('<BBBBI', 0x13, 1, 0, 0, size)          # fixed width, null-padded (numpy 'S<n>')


def _dt_of(dtype):
    """Datatype message of a numpy dtype: integers, floats, 'S<n>' and flat compounds of those
    (version-1 compound members: name, byte offset, no dimensions, member type)."""
    dtype = np.dtype(dtype)
    if dtype.names:
        body = b''
        for name in dtype.names:
            sub, offset = dtype.fields[name][:2]
            if sub.names or sub.shape:
                raise TypeError('nested / array members are not written')
            body += _pad8(name.encode() + b'\0') + struct.pack('<IB3xII16x', offset, 0, 0, 0) + _dt_of(sub)
        return struct.pack('<BBBBI', 0x16, len(dtype.names) & 0xFF, len(dtype.names) >> 8, 0, dtype.itemsize) + body
    if dtype.kind == 'S':
        return _dt_bytes(dtype.itemsize)
    if dtype.kind in 'iu':
        return _dt_int(dtype.itemsize, dtype.kind == 'i')
    if dtype.kind == 'f' and dtype.itemsize in (4, 8):
        return _dt_float(dtype.itemsize)
    raise TypeError('dtype {} is not written'.format(dtype))


def _little(a):
    """The array's bytes with every number little-endian (a no-op on the hosts this runs on)."""
    return np.ascontiguousarray(a).astype(a.dtype.newbyteorder('<'), copy=False).tobytes()


def _space(dims):
    if dims is None:
        return struct.pack('<BBB5x', 1, 0, 0)                  # scalar
    return struct.pack('<BBB5x', 1, len(dims), 0) + b''.join(struct.pack('<Q', d) for d in dims)


def _typed(value):
    """(datatype message, raw bytes) of one attribute value."""
    if isinstance(value, (bytes, str)):
        b = value.encode() if isinstance(value, str) else value
        return _dt_string(len(b) + 1), b + b'\0'
    v = np.asarray(value)
    if v.dtype.kind == 'f':
        return _dt_float(v.dtype.itemsize), v.astype(v.dtype.newbyteorder('<')).tobytes()
    if v.dtype.kind in 'iu':
        return _dt_int(v.dtype.itemsize, v.dtype.kind == 'i'), v.astype(v.dtype.newbyteorder('<')).tobytes()
    raise TypeError('attribute type {}'.format(v.dtype))


def _message(mtype, body):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), 0) + body


def _attribute(name, value):
    dt, data = _typed(value)
    nm = name.encode() + b'\0'
    sp = _space(None)
    return _message(0x000C, struct.pack('<BxHHH', 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) +
                    _pad8(sp) + data)


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BxHII4x', 1, len(messages), 1, len(body)) + body


# ---- VBZ (filter 32020, version 1, 16-bit samples) ---------------------------------------------
_zstd = None


def _libzstd():
    global _zstd
    if _zstd is None:
        lib = C.CDLL('libzstd.so.1')
        lib.ZSTD_compress.restype = C.c_size_t
        lib.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        lib.ZSTD_compressBound.restype = C.c_size_t
        lib.ZSTD_compressBound.argtypes = [C.c_size_t]
        lib.ZSTD_isError.restype = C.c_uint
        lib.ZSTD_isError.argtypes = [C.c_size_t]
        _zstd = lib
    return _zstd


def vbz_encode(samples, level=1):
    """int16 samples -> the bytes of one VBZ (version 1) chunk: deltas from the previous sample
    (the first from 0), zig-zag, one control BIT per sample (set: two bytes) in ceil(n/8) key
    bytes followed by the data bytes, then one zstd frame over all of it."""
    x = np.ascontiguousarray(samples, dtype='<i2').astype(np.uint16)
    d = np.diff(x, prepend=np.uint16(0)).astype(np.uint16).view(np.int16)
    zz = ((d.astype(np.int32) << 1) ^ (d.astype(np.int32) >> 15)).astype(np.uint16)
    two = zz > 0xFF
    keys = np.packbits(two, bitorder='little')
    width = np.where(two, 2, 1)
    pos = np.concatenate([[0], np.cumsum(width)])
    data = np.zeros(int(pos[-1]), dtype=np.uint8)
    data[pos[:-1]] = zz & 0xFF
    data[pos[:-1][two] + 1] = zz[two] >> 8
    svb = keys.tobytes() + data.tobytes()
    lib = _libzstd()
    cap = lib.ZSTD_compressBound(len(svb))
    out = C.create_string_buffer(cap)
    got = lib.ZSTD_compress(out, cap, svb, len(svb), level)
    if lib.ZSTD_isError(got):
        raise RuntimeError('ZSTD_compress failed')
    return out.raw[:got]


class _File:
    """Append-only image of the file; every structure is placed at an 8-byte boundary."""

    def __init__(self):
        self.buf = bytearray(96)                               # superblock goes here at the end

    def put(self, data):
        at = len(self.buf)
        self.buf += data
        self.buf += b'\0' * (-len(self.buf) % 8)
        return at


class _Group:
    def __init__(self):
        self.children = {}          # name -> object header address
        self.attrs = []

    def write(self, f):
        """Local heap + B-tree + object header; returns (header address, btree, heap)."""
        names = sorted(self.children, key=lambda s: s.encode())
        heap = bytearray(b'\0' * 8)                            # offset 0: the empty string
        off = {}
        for n in names:
            off[n] = len(heap)
            heap += n.encode() + b'\0'
            heap += b'\0' * (-len(heap) % 8)
        heap += b'\0' * 16                                     # a free block, as the library leaves one
        free_at = len(heap) - 16
        struct.pack_into('<QQ', heap, free_at, 1, 16)          # next free (1 = none), size
        data_at = f.put(bytes(heap))
        heap_at = f.put(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), free_at, data_at))
        # leaves
        level = []
        for a in range(0, max(len(names), 1), 2 * LEAF_K):
            part = names[a:a + 2 * LEAF_K]
            node = b'SNOD' + struct.pack('<BxH', 1, len(part))
            for n in part:
                node += struct.pack('<QQII16x', off[n], self.children[n], 0, 0)
            node += b'\0' * (8 + 2 * LEAF_K * 40 - len(node))
            level.append((f.put(node), off[part[-1]] if part else 0))
        depth = 0
        while True:
            nxt = []
            for a in range(0, len(level), 2 * INTERNAL_K):
                part = level[a:a + 2 * INTERNAL_K]
                node = b'TREE' + struct.pack('<BBHQQ', 0, depth, len(part), UNDEF, UNDEF) + struct.pack('<Q', 0)
                for addr, last in part:
                    node += struct.pack('<QQ', addr, last)
                node += b'\0' * (24 + 2 * INTERNAL_K * 8 + (2 * INTERNAL_K + 1) * 8 - len(node))
                nxt.append((f.put(node), part[-1][1]))
            level, depth = nxt, depth + 1
            if len(level) == 1:
                break
        btree = level[0][0]
        hdr = f.put(_object_header([_message(0x0011, struct.pack('<QQ', btree, heap_at))] +
                                   [_attribute(k, v) for k, v in self.attrs]))
        return hdr, btree, heap_at


def _dataset(f, data, compression=None, chunk=None, attrs=()):
    """Object header of a 1-D (or scalar string) dataset; returns its address."""
    if isinstance(data, (bytes, str)):
        b = (data.encode() if isinstance(data, str) else data) + b'\0'
        msgs = [_message(0x0001, _space(None)), _message(0x0003, _dt_string(len(b))),
                _message(0x0005, struct.pack('<BBBB', 2, 2, 0, 0)),
                _message(0x0008, struct.pack('<BBQQ', 3, 1, f.put(b), len(b)))]
        return f.put(_object_header(msgs + [_attribute(k, v) for k, v in attrs]))
    a = np.ascontiguousarray(data)
    n, esz = len(a), a.dtype.itemsize
    if compression is not None and (a.dtype.names or a.dtype.kind == 'S'):
        raise TypeError('compressed datasets are written for numbers')
    msgs = [_message(0x0001, _space([n])), _message(0x0003, _dt_of(a.dtype))]
    raw = _little(a)
    if compression is None:
        msgs += [_message(0x0005, struct.pack('<BBBB', 2, 2, 0, 0)),
                 _message(0x0008, struct.pack('<BBQQ', 3, 1, f.put(raw) if n else UNDEF, len(raw)))]
    else:
        clen = int(chunk or max(n, 1))
        n_chunks = max(-(-n // clen), 1)
        if n_chunks > 2 * CHUNK_K:
            raise ValueError('at most {} chunks per dataset'.format(2 * CHUNK_K))
        entries = []
        for c in range(n_chunks):
            part = a[c * clen:(c + 1) * clen]
            if len(part) < clen:                               # edge chunks are stored whole
                part = np.concatenate([part, np.zeros(clen - len(part), dtype=a.dtype)])
            if compression == 'gzip':
                blob = zlib.compress(part.astype(a.dtype.newbyteorder('<')).tobytes(), 1)
            elif compression == 'vbz':
                if a.dtype != np.int16:
                    raise TypeError('VBZ is written for int16 samples')
                blob = vbz_encode(part)
            else:
                raise ValueError('compression ' + repr(compression))
            entries.append((len(blob), c * clen, f.put(blob)))
        node = b'TREE' + struct.pack('<BBHQQ', 1, 0, len(entries), UNDEF, UNDEF)
        for size, first, addr in entries:
            node += struct.pack('<IIQQ', size, 0, first, 0) + struct.pack('<Q', addr)
        node += struct.pack('<IIQQ', 0, 0, n_chunks * clen, 0)                   # the closing key
        node += b'\0' * (24 + 2 * CHUNK_K * 8 + (2 * CHUNK_K + 1) * 24 - len(node))
        btree = f.put(node)
        if compression == 'gzip':
            pipeline = struct.pack('<BB6x', 1, 1) + struct.pack('<HHHH', 1, 0, 0, 1) + struct.pack('<II', 1, 0)
        else:
            name = _pad8(b'vbz\0')
            pipeline = struct.pack('<BB6x', 1, 1) + struct.pack('<HHHH', 32020, len(name), 0, 4) + name + \
                struct.pack('<IIII', 1, 2, 1, 1)               # version, integer size, zig-zag, zstd level
        msgs += [_message(0x0005, struct.pack('<BBBB', 2, 3, 0, 0)), _message(0x000B, pipeline),
                 _message(0x0008, struct.pack('<BBBQII', 3, 2, 2, btree, clen, esz))]
    return f.put(_object_header(msgs + [_attribute(k, v) for k, v in attrs]))


def _read_groups(f, read_id, raw, calib, start_time, channel_number, run_id, sample_id, basecall,
                 compression, chunk, read_number=0):
    """(Raw-attribute group carrying Signal, channel_id, tracking_id, Analyses or None)."""
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    g_raw = _Group()
    g_raw.attrs = [('duration', np.uint32(len(raw))), ('start_time', np.uint64(start_time)),
                   ('read_id', read_id), ('read_number', np.int32(read_number))]
    g_raw.children['Signal'] = _dataset(f, raw, compression, chunk)
    g_ch = _Group()
    g_ch.attrs = [('channel_number', str(channel_number)), ('digitisation', np.float64(calib['digitisation'])),
                  ('offset', np.float64(calib['offset'])), ('range', np.float64(calib['range'])),
                  ('sampling_rate', np.float64(calib['sampling_rate']))]
    g_tr = _Group()
    g_tr.attrs = [('run_id', run_id), ('sample_id', sample_id)]
    analyses = None
    if basecall is not None:
        tpl = _Group()
        tpl.children['Fastq'] = _dataset(f, '@{}\n{}\n+\n{}\n'.format(read_id, basecall['sequence'], basecall['qstring']))
        tpl.children['Move'] = _dataset(f, np.asarray(basecall['move'], dtype=np.uint8))
        sm = _Group()
        sm.attrs = [('sequence_length', np.int32(basecall['sequence_length'])),
                    ('mean_qscore', np.float32(basecall['mean_qscore'])),
                    ('block_stride', np.int32(basecall.get('block_stride', 15)))]
        summary = _Group()
        summary.children['basecall_1d_template'] = sm.write(f)[0]
        bc = _Group()
        bc.children['BaseCalled_template'] = tpl.write(f)[0]
        bc.children['Summary'] = summary.write(f)[0]
        seg = _Group()
        seg.attrs = [('num_events_template', np.int32(basecall['num_events'])),
                     ('first_sample_template', np.int32(basecall['first_sample_template']))]
        seg_sum = _Group()
        seg_sum.children['segmentation'] = seg.write(f)[0]
        seg_top = _Group()
        seg_top.children['Summary'] = seg_sum.write(f)[0]
        analyses = _Group()
        analyses.children['Basecall_1D_000'] = bc.write(f)[0]
        analyses.children['Segmentation_000'] = seg_top.write(f)[0]
    return g_raw, g_ch, g_tr, analyses


def _finish(f, root, path):
    hdr, btree, heap = root.write(f)
    sb = b'\x89HDF\r\n\x1a\n' + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, len(f.buf), UNDEF)
    sb += struct.pack('<QQII', 0, hdr, 1, 0) + struct.pack('<QQ', btree, heap)
    assert len(sb) == 96
    f.buf[:96] = sb
    with open(path, 'wb') as fh:
        fh.write(f.buf)


class Fast5Writer:
    """Multi-read FAST5: one `read_<id>` group per read."""

    def __init__(self, path):
        self.path, self.f, self.root = path, _File(), _Group()

    def add_read(self, read_id, raw, calib, start_time=0, channel_number='1', run_id='run', sample_id='sample',
                 basecall=None, compression=None, chunk=None, read_number=0):
        g_raw, g_ch, g_tr, analyses = _read_groups(self.f, read_id, raw, calib, start_time, channel_number,
                                                    run_id, sample_id, basecall, compression, chunk, read_number)
        g = _Group()
        g.children['Raw'] = g_raw.write(self.f)[0]
        g.children['channel_id'] = g_ch.write(self.f)[0]
        g.children['tracking_id'] = g_tr.write(self.f)[0]
        if analyses is not None:
            g.children['Analyses'] = analyses.write(self.f)[0]
        self.root.children['read_' + read_id] = g.write(self.f)[0]

    def close(self):
        _finish(self.f, self.root, self.path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()


def write_single_read(path, read_id, raw, calib, start_time=0, channel_number='1', run_id='run',
                      sample_id='sample', basecall=None, compression=None, chunk=None, read_number=0):
    """Single-read FAST5: Raw/Reads/Read_<n>, UniqueGlobalKey/{channel_id,tracking_id}, Analyses."""
    f = _File()
    g_raw, g_ch, g_tr, analyses = _read_groups(f, read_id, raw, calib, start_time, channel_number, run_id,
                                                sample_id, basecall, compression, chunk, read_number)
    reads = _Group()
    reads.children['Read_{}'.format(read_number)] = g_raw.write(f)[0]
    rawtop = _Group()
    rawtop.children['Reads'] = reads.write(f)[0]
    ugk = _Group()
    ugk.children['channel_id'] = g_ch.write(f)[0]
    ugk.children['tracking_id'] = g_tr.write(f)[0]
    root = _Group()
    root.children['Raw'] = rawtop.write(f)[0]
    root.children['UniqueGlobalKey'] = ugk.write(f)[0]
    root.children['Analyses'] = (analyses if analyses is not None else _Group()).write(f)[0]
    _finish(f, root, path)


class H5Writer:
    """A plain HDF5 file of groups, 1-D datasets (numbers, 'S<n>', flat compound records) and
    scalar attributes, written whole on close -- the per-worker dump files of signal_analyzer.py
    (adapter-dumps/part-*.h5, events/part-*.h5).

        with H5Writer(path) as h5:
            h5.create_dataset('adapter/00000003/<read id>', float32_array)
            h5.create_dataset('catalog/adapter/00000003', records, attrs=[('n', np.int32(7))])
    """

    def __init__(self, path):
        self.path, self.f = path, _File()
        self.tree = {}                   # name -> sub-tree dict, or the address of a dataset header

    def _walk(self, parts):
        node = self.tree
        for name in parts:
            node = node.setdefault(name, {})
            if not isinstance(node, dict):
                raise ValueError('{!r} is a dataset'.format(name))
        return node

    def require_group(self, path):
        self._walk([p for p in path.split('/') if p])

    def __contains__(self, path):
        node = self.tree
        for name in [p for p in path.split('/') if p]:
            if not isinstance(node, dict) or name not in node:
                return False
            node = node[name]
        return True

    def create_dataset(self, path, data, attrs=()):
        parts = [p for p in path.split('/') if p]
        parent = self._walk(parts[:-1])
        if parts[-1] in parent:
            raise ValueError('name already exists: ' + path)
        parent[parts[-1]] = _dataset(self.f, data, attrs=attrs)

    def _write(self, node):
        g = _Group()
        for name, child in node.items():
            g.children[name] = self._write(child) if isinstance(child, dict) else child
        return g.write(self.f)[0]

    def close(self):
        root = _Group()
        for name, child in self.tree.items():
            root.children[name] = self._write(child) if isinstance(child, dict) else child
        _finish(self.f, root, self.path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
