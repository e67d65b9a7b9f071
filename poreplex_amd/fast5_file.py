"""Read sources for the hot path: FAST5 files and .pxr.npz bundles.

Mirrors what the per-read processor needs from the reference's
poreplex/fast5_file.py (`Fast5Reader`: metadata :97-120, raw samples :122-131,
basecall summary :133-164).  The pA conversion itself is NOT done here: the
reader hands int16 DAQ samples + calibration to the GPU (kernel a1).

FAST5 files are read WITHOUT h5py: csrc/pxg_h5.cpp (libpxghost.so, host only) parses the
subset of the HDF5 format FAST5 files are written in straight from a memory map and decodes
the signals of a whole batch on host threads into the staging arena the GPU copies from
(`Fast5Batch`); `Fast5Reader` / `get_read_ids` keep the reference's per-read surface on top of
the same library.  A file it declines (dense link / attribute storage of `libver latest`)
falls back to h5py where that is importable, and fails loudly where it is not.

A *bundle* (`*.pxr.npz`) is this build's array container for many reads (columnar metadata,
one sample arena); a batch of FAST5 reads becomes the same columns (`Fast5Batch.as_bundle`),
so everything behind the loader is shared.
"""
import json
import os
import threading
from collections import OrderedDict

import numpy as np

try:  # optional: only for files the native reader declines
    import h5py
except ImportError:  # pragma: no cover
    h5py = None

__all__ = ['get_read_ids', 'get_read_ids_many', 'open_read', 'ReadBundle', 'Fast5Reader', 'Fast5File', 'Fast5Batch', 'write_bundle']


TABLE_KINDS = ('', 'move', 'guppy_events', 'albacore', 'unsupported')   # '' = no event table
EVENT_NUMERIC = ('start', 'length', 'mean', 'stdv', 'move', 'p_model_state')
EVENT_COLUMNS = EVENT_NUMERIC + ('model_state',)

# columnar basecall summary of a bundle (version 2); ragged columns have offsets[n+1]
BASECALL_COLUMNS = ('bc_present', 'bc_sequence_length', 'bc_mean_qscore', 'bc_num_events',
                    'bc_first_sample', 'bc_block_stride', 'bc_table', 'bc_n_moves', 'bc_move_sum',
                    'seq_offsets', 'seq_arena', 'qual_arena', 'move_offsets', 'move_arena')


def basecall_columns(basecalls):
    """list of get_basecall()-style dicts (or None) -> the columnar form stored in a bundle.
    A per-event p_model_state column (Guppy `Events' tables) is kept as JSON text: rare."""
    n = len(basecalls)
    c = {'bc_present': np.zeros(n, dtype=bool), 'bc_sequence_length': np.zeros(n, dtype=np.int64),
         'bc_mean_qscore': np.zeros(n, dtype=np.float64), 'bc_num_events': np.zeros(n, dtype=np.int64),
         'bc_first_sample': np.zeros(n, dtype=np.int64), 'bc_block_stride': np.full(n, 15, dtype=np.int32),
         'bc_table': np.zeros(n, dtype=np.int8), 'bc_n_moves': np.full(n, -1, dtype=np.int64),
         'bc_move_sum': np.zeros(n, dtype=np.int64)}
    seqs, quals, moves, pms = [], [], [], []
    for i, bc in enumerate(basecalls):
        if not bc:
            seqs.append(b''); quals.append(b''); moves.append(np.zeros(0, np.uint8)); pms.append('')
            continue
        c['bc_present'][i] = True
        c['bc_sequence_length'][i] = bc['sequence_length']
        c['bc_mean_qscore'][i] = bc['mean_qscore']
        c['bc_num_events'][i] = bc['num_events']
        c['bc_first_sample'][i] = bc['first_sample_template']
        c['bc_block_stride'][i] = bc.get('block_stride', 15)
        move = bc.get('move')
        kind = bc.get('table', 'move' if move is not None else '') or ''
        c['bc_table'][i] = TABLE_KINDS.index(kind)
        mv = np.zeros(0, np.uint8) if move is None else np.asarray(move, dtype=np.uint8)
        if move is not None:
            c['bc_n_moves'][i], c['bc_move_sum'][i] = len(mv), int(mv.sum())
        seqs.append(bc['sequence'].encode('ascii'))
        quals.append(bc['qstring'].encode('ascii'))
        moves.append(mv)
        pms.append(json.dumps(bc['p_model_state']) if bc.get('p_model_state') is not None else '')
    for name, parts in (('seq', seqs), ('move', moves)):
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(p) for p in parts], out=off[1:])
        c[name + '_offsets'] = off
    # tables that bring their own events (albacore's 14-column Events): the columns the processor consumes,
    # ragged over the reads (float64 / text arenas + each read's own dtypes, so nothing of the file's typing
    # -- which decides how the reference's pandas arithmetic promotes -- is lost)
    tables = [bc.get('events') if bc else None for bc in basecalls]
    if any(t is not None for t in tables):
        eo = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(t['start']) if t is not None else 0 for t in tables], out=eo[1:])
        c['ev_offsets'] = eo
        for name in EVENT_NUMERIC:
            c['ev_' + name] = np.concatenate([np.asarray(t[name], dtype=np.float64) if t is not None and name in t
                                              else np.zeros(len(t['start']) if t is not None else 0)
                                              for t in tables]) if eo[-1] else np.zeros(0)
        c['ev_model_state'] = np.concatenate([np.asarray(t['model_state'], dtype='S8') if t is not None and 'model_state' in t
                                              else np.zeros(len(t['start']) if t is not None else 0, dtype='S8')
                                              for t in tables]) if eo[-1] else np.zeros(0, dtype='S8')
        c['ev_dtypes'] = np.array([json.dumps({k: np.asarray(v).dtype.str for k, v in t.items()}) if t is not None else ''
                                   for t in tables])
    c['seq_arena'] = np.frombuffer(b''.join(seqs), dtype=np.uint8).copy()
    c['qual_arena'] = np.frombuffer(b''.join(quals), dtype=np.uint8).copy()
    c['move_arena'] = np.concatenate(moves) if moves else np.zeros(0, np.uint8)
    if any(pms):
        c['bc_p_model_state'] = np.array(pms)
    return c


def write_bundle(path, arena, offsets, calib, filename, read_id, basecalls=None, compress=False, **columns):
    """Write a .pxr.npz read bundle (this build's array container for many reads: int16
    samples + per-read metadata + columnar basecall summaries).  Missing metadata columns get
    neutral defaults.  compress=True stores the samples as zig-zag delta bytes in independent
    1 024-sample chunks (include/pxg.h, pxg_z_*: ~1.2 bytes per sample) instead of the int16
    arena; the session then sends those bytes across PCIe and the GPU decodes them."""
    n = len(offsets) - 1
    arena = np.ascontiguousarray(arena, dtype=np.int16)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if compress:
        from . import native
        z, chunks, chunk_base = native.z_encode(arena, offsets)
        samples = {'arena_z': z, 'z_chunks': chunks, 'z_chunk_base': chunk_base}
    else:
        samples = {'arena': arena}
    d = {'offsets': offsets, 'calib': calib,
         'filename': np.asarray(filename), 'read_id': np.asarray(read_id),
         'duration': np.diff(offsets), 'start_time': np.zeros(n, dtype=np.int64),
         'channel_number': np.array(['0'] * n), 'run_id': np.array(['run'] * n),
         'sample_id': np.array(['sample'] * n), 'broken_files': np.array([], dtype='<U1'),
         'bundle_version': np.int64(3 if compress else 2)}
    d.update(samples)
    d.update(columns)
    d.update(basecall_columns(basecalls if basecalls is not None else [None] * n))
    np.savez(path, **d)


class ReadBundle:
    """Random access to the reads of one .pxr.npz file by (filename, read_id), and columnar
    access for the batch paths (signal_loader.ReadTable.extend_from_bundle)."""

    def __init__(self, path):
        with np.load(path, allow_pickle=False) as npz:
            self.d = {k: npz[k] for k in npz.files}
        d = self.d
        if 'bc_present' not in d:          # version 1: one JSON string per read -> columns
            js = d['basecall'] if 'basecall' in d else [''] * len(d['read_id'])
            d.update(basecall_columns([json.loads(str(j)) if str(j) else None for j in js]))
        self.filenames = [str(f) for f in d['filename']]
        self.read_ids = [str(r) for r in d['read_id']]
        self.keys = list(zip(self.filenames, self.read_ids))
        self.index = {key: i for i, key in enumerate(self.keys)}
        self.by_file = {}
        for i, f in enumerate(self.filenames):
            self.by_file.setdefault(f, []).append(i)
        if 'arena_z' in d:       # encoded samples: the records must describe the bytes (once, here)
            from . import native
            o, base = d['offsets'], d['z_chunk_base']
            if len(base) != len(o) or int(base[-1]) != len(d['z_chunks']) or \
                    not np.array_equal(np.diff(base), (np.diff(o) + native.Z_CHUNK - 1) // native.Z_CHUNK):
                raise native.PxgError('{}: chunk table does not match the read offsets'.format(path))
            native.z_validate(d['arena_z'], d['z_chunks'], int(o[-1]))
            # chunks never span reads: every read's first chunk starts at the read's first sample
            if len(d['z_chunks']) and not np.array_equal(d['z_chunks']['dst'][base[:-1][np.diff(base) > 0]],
                                                          o[:-1][np.diff(base) > 0]):
                raise native.PxgError('{}: a chunk spans two reads'.format(path))
        elif len(d['arena']) != int(d['offsets'][-1]) if len(d['offsets']) else False:
            raise ValueError('{}: sample arena does not match the read offsets'.format(path))
        # files that exist but cannot be opened (the corrupt-FAST5 case)
        self.broken = set(str(f) for f in d.get('broken_files', []))

    compressed = property(lambda self: 'arena_z' in self.d)

    def samples_run(self, i0, i1):
        """The samples of the consecutive reads [i0, i1): an int16 view of the arena, or -- in a
        compressed bundle -- the encoded bytes and chunk records of exactly those reads
        (native.EncodedSamples, views as well: chunks never span reads)."""
        d = self.d
        o = d['offsets']
        if not self.compressed:
            return d['arena'][o[i0]:o[i1]]
        from . import native
        base, ch = d['z_chunk_base'], d['z_chunks']
        c0, c1 = int(base[i0]), int(base[i1])
        b0 = int(ch['data_off'][c0]) if c0 < len(ch) else len(d['arena_z'])
        b1 = int(ch['data_off'][c1]) if c1 < len(ch) else len(d['arena_z'])
        return native.EncodedSamples(d['arena_z'][b0:b1], ch[c0:c1], b0, int(o[i0]), int(o[i1] - o[i0]))

    def samples(self, i):
        """int16 samples of read i (decoded on the host if the bundle is compressed)."""
        run = self.samples_run(i, i + 1)
        return run if isinstance(run, np.ndarray) else run.decode()

    def plain_run_columns(self, scaler_cfg):
        """What the worker call's short path needs of the bundle (signal_analyzer.process_plain_run, csrc/pxg_pyreport.c
        report_run), made once: the columns as contiguous arrays of the types the extension reads, and `ok` -- the
        reads a call may take that path with: long enough for the scaler (the gate of load_padded_signal_head,
        signal_loader.py:212-222) and either without a basecall or with a summary whose Guppy frame fits the raw signal
        (SignalAnalyzer.bulk_base_space's `regular', the part of it that does not depend on the GPU pass).  None if the
        bundle cannot take the path at all."""
        key = (scaler_cfg['length'], scaler_cfg['stride'], scaler_cfg['min_length'])
        cached = getattr(self, '_plain', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        d = self.d
        plain = None
        n = len(self.filenames)
        text = [np.ascontiguousarray(d[name]) for name in ('channel_number', 'run_id', 'sample_id')]
        if all(a.dtype.kind == 'U' and a.dtype.isnative and a.ndim == 1 and len(a) == n for a in text) \
                and (d['calib']['sampling_rate'] != 0).all():
            o = np.ascontiguousarray(d['offsets'], dtype=np.int64)
            n_raw = np.diff(o)
            usable = np.minimum(np.minimum(scaler_cfg['length'], d['duration']), n_raw)
            long_enough = usable - usable % scaler_cfg['stride'] >= scaler_cfg['min_length']
            first, stride, n_moves = d['bc_first_sample'], d['bc_block_stride'].astype(np.int64), d['bc_n_moves']
            covered = np.maximum(np.minimum(first + stride * n_moves, n_raw) - first, 0)
            kind = d['bc_table']
            regular = ((kind == 1) | (kind == 2)) & (n_moves >= 0) & (stride > 0) & \
                (-(-covered // np.maximum(stride, 1)) == n_moves)
            present = d['bc_present'].astype(bool)
            # the chimera scan (signal_loader.unsplit_frames, SignalAnalyzer.bulk_base_space): the Guppy block frame of
            # every regular read, the one block stride they share (None: none or several -- no short path with the
            # scan), and the reads whose Move table has the k-mer size the event frame needs
            frame = np.zeros((n, 3), dtype=np.int64)
            fits = present & regular
            frame[fits] = np.stack([first, n_moves, stride], axis=1)[fits]
            strides = np.unique(frame[frame[:, 1] > 0, 2])
            kmer = (d['seq_offsets'][1:] - d['seq_offsets'][:-1]) - d['bc_move_sum'] + 1
            plain = {'ok': long_enough & (~present | regular), 'offsets': o,
                     # (the two halves of `ok`: a read too short for the scaler may sit in a plain run -- its dict says so and
                     #  comes first, as the reference returns it --, an irregular basecall summary may not)
                     'long_enough': long_enough, 'regular': ~present | regular,
                     'frame_first': np.ascontiguousarray(frame[:, 0]), 'frame_blocks': np.ascontiguousarray(frame[:, 1]),
                     'frame_stride': int(strides[0]) if len(strides) == 1 else None,
                     'kmer_ok': ~fits | (kind == 2) | (kmer == 5) | (kmer == 1),
                     'filenames': self.filenames, 'read_ids': self.read_ids,
                     'channel_number': text[0], 'run_id': text[1], 'sample_id': text[2],
                     'calib': np.ascontiguousarray(d['calib'])}
            for name, dtype in (('start_time', np.int64), ('duration', np.int64), ('bc_present', np.bool_),
                                ('bc_sequence_length', np.int64), ('bc_mean_qscore', np.float64),
                                ('bc_num_events', np.int64), ('seq_offsets', np.int64), ('seq_arena', np.uint8),
                                ('qual_arena', np.uint8)):
                plain[name] = np.ascontiguousarray(d[name], dtype=dtype)
        self._plain = (key, plain)
        return plain

    def has_file(self, filename):
        return filename in self.by_file or filename in self.broken

    def read_ids_of(self, filename):
        return [(filename, self.read_ids[i]) for i in self.by_file.get(filename, [])]

    def reader(self, filename, read_id):
        key = (filename, read_id)
        if filename in self.broken:
            raise OSError('Unable to open file {!r} (file signature not found)'.format(filename))
        if key not in self.index:
            if filename in self.by_file:     # fast5_file.py:104-107
                raise ValueError('Unexpected read {} found in {}'.format(
                    self.read_ids[self.by_file[filename][0]], filename))
            raise FileNotFoundError(filename)
        return BundleReader(self, self.index[key])

    def sequence_text(self):
        """(sequences, quality strings) of all reads as two bytes objects (made once)."""
        if getattr(self, '_text', None) is None:
            self._text = (self.d['seq_arena'].tobytes(), self.d['qual_arena'].tobytes())
        return self._text

    def sequence_of(self, i):
        o = self.d['seq_offsets']
        return (self.d['seq_arena'][o[i]:o[i + 1]].tobytes().decode('ascii'),
                self.d['qual_arena'][o[i]:o[i + 1]].tobytes().decode('ascii'))

    def basecall_of(self, i, move_as_array=False):
        """get_basecall()-style dict of read i from the columns; None if not basecalled.  `move_as_array`: the Move
        column as a uint8 view of the bundle's arena instead of a list (the facade's per-read rules turn it into an
        array again: 4 000 blocks through a Python list cost more than the rules themselves)."""
        d = self.d
        if not d['bc_present'][i]:
            return None
        seq, qual = self.sequence_of(i)
        kind = TABLE_KINDS[int(d['bc_table'][i])] or None
        mo = d['move_offsets']
        move = d['move_arena'][mo[i]:mo[i + 1]] if d['bc_n_moves'][i] >= 0 else None
        if move is not None and not move_as_array:
            move = move.tolist()
        pms = None
        if 'bc_p_model_state' in d and str(d['bc_p_model_state'][i]):
            pms = json.loads(str(d['bc_p_model_state'][i]))
        out = {'sequence': seq, 'qstring': qual, 'block_stride': int(d['bc_block_stride'][i]),
               'sequence_length': int(d['bc_sequence_length'][i]),
               'mean_qscore': float(np.float32(d['bc_mean_qscore'][i])),
               'num_events': int(d['bc_num_events'][i]),
               'first_sample_template': int(d['bc_first_sample'][i]),
               'table': kind, 'move': move, 'p_model_state': pms}
        if 'ev_dtypes' in d and str(d['ev_dtypes'][i]):
            eo, dt = d['ev_offsets'], json.loads(str(d['ev_dtypes'][i]))
            out['events'] = {k: d['ev_' + k][eo[i]:eo[i + 1]].astype(np.dtype(dt[k])) for k in EVENT_COLUMNS if k in dt}
        return out


class _Fast5BatchBundle(ReadBundle):
    """ReadBundle over the columns of a Fast5Batch (nothing on disk)."""

    def __init__(self, d, batch):
        self.d, self.batch = d, batch
        self.filenames = batch.names
        self.read_ids = batch.read_ids if batch.read_ids is not None else [str(r) for r in d['read_id']]
        self.broken = set()

    # the lookup structures of a bundle on disk: built when somebody asks (the batch path does not)
    @property
    def keys(self):
        if getattr(self, '_keys', None) is None:
            self._keys = list(zip(self.filenames, self.read_ids))
        return self._keys

    @property
    def index(self):
        if getattr(self, '_index', None) is None:
            self._index = {key: i for i, key in enumerate(self.keys)}
        return self._index

    @property
    def by_file(self):
        if getattr(self, '_by_file', None) is None:
            self._by_file = {}
            for i, f in enumerate(self.filenames):
                self._by_file.setdefault(f, []).append(i)
        return self._by_file

    def text_lists(self, idx):
        """(channel numbers, run ids, sample ids) of ALL the bundle's reads as Python lists, when `idx` asks for all of
        them in order and the batch is stretches of multi-read files: slices of lists kept with the open files
        (FileRunColumns.texts) instead of three .tolist() of 10 000 fresh strings per batch.  None: the columns."""
        runs = getattr(self.batch, 'runs', None)
        n = len(self.filenames)
        if runs is None or not _RUN_COLUMNS_IN_BATCHES or len(idx) != n or (n and (idx[0] != 0 or idx[-1] != n - 1)):
            return None
        out = ([], [], [])
        for f, name, i0, count in runs:
            lists = file_run_columns(f, name).texts()
            for k in range(3):
                out[k].extend(lists[k][i0:i0 + count])
        return out

    def basecall_of(self, i, move_as_array=False):
        """The per-read path (chimera candidates, Events tables): straight from the file, so a
        p_model_state column comes along."""
        return self.batch.files[i].basecall(int(self.batch.index[i]))


class BundleReader:
    """Same attribute surface as the reference's Fast5Reader."""

    def __init__(self, bundle, i):
        d = bundle.d
        self.bundle, self.i, self.d = bundle, i, d
        self.path, self.read_id = bundle.filenames[i], bundle.read_ids[i]
        self.duration = int(d['duration'][i])
        self.start_time = int(d['start_time'][i])
        self.channel_number = str(d['channel_number'][i])
        cal = d['calib'][i]
        self.digitization = float(cal['digitisation'])
        self.offset = float(cal['offset'])
        self.range = float(cal['range'])
        self.sampling_rate = float(cal['sampling_rate'])
        self.run_id = str(d['run_id'][i])
        self.sample_id = str(d['sample_id'][i])

    def close(self):
        pass

    def get_raw_int16(self):
        return self.bundle.samples(self.i)

    def get_basecall(self, move_as_array=False):
        """Summary of Analyses/Basecall_1D_* (fast5_file.py:133-164); None if absent."""
        return self.bundle.basecall_of(self.i, move_as_array)


class Fast5Error(OSError):
    pass


# seconds spent in the native reader, by phase (walk = group walk at open + metadata columns, signals = copy / decode
# of the samples, text = basecall text and move tables): bench.py's ingest leg and the session print them
TIMING = {'walk_s': 0.0, 'signals_s': 0.0, 'text_s': 0.0}


def _timed(key, t0):
    import time
    TIMING[key] += time.perf_counter() - t0


class Fast5File:
    """One FAST5 file opened by the native reader: read ids, per-read metadata columns
    (pxg_h5_read_info) and handles for the batch loaders.  Cached per path (open_fast5)."""

    def __init__(self, path):
        from . import native
        import ctypes as C
        self.lib = native.load_text_library()
        self.path = path
        handle = C.c_void_p()
        import time
        t0 = time.perf_counter()
        rc = self.lib.pxg_h5_open_mt(os.fsencode(path), host_threads(), C.byref(handle))
        _timed('walk_s', t0)
        if rc:
            msg = (self.lib.pxg_h5_last_error() or b'').decode(errors='replace')
            err = Fast5Error(msg or 'Unable to open file {!r}'.format(path))
            err.code = rc
            raise err
        self.handle = handle
        self.n = int(self.lib.pxg_h5_n_reads(handle))
        self.multi = bool(self.lib.pxg_h5_is_multi(handle))
        self._ids = self._info = self._index = self._ids_array = None
        self._keys = {}
        self.size = 0
        self._lazy = threading.Lock()        # worker threads share an open file: its lazy columns are made once

    @classmethod
    def from_handle(cls, path, handle, n, multi, info=None, owner=None):
        """A Fast5File over a file that open_many has opened (`owner`: the OpenedFiles that closes it)."""
        from . import native
        import ctypes as C
        self = cls.__new__(cls)
        self.lib = native.load_text_library()
        self.path, self.handle, self.owner = path, C.c_void_p(int(handle)), owner
        self.n, self.multi = int(n), bool(multi)
        self._ids = self._index = self._ids_array = None
        self._info = info
        self._keys = {}
        self.size = 0
        self._lazy = threading.Lock()
        return self

    def close(self):
        handle, self.handle = getattr(self, 'handle', None), None
        if handle and getattr(self, 'owner', None) is None:
            self.lib.pxg_h5_close(handle)

    __del__ = close

    @property
    def read_ids(self):
        if self._ids is None:
            with self._lazy:
                if self._ids is None:
                    import ctypes as C
                    need = int(self.lib.pxg_h5_read_ids(self.handle, None, 0))
                    buf = C.create_string_buffer(max(need, 1))
                    if self.lib.pxg_h5_read_ids(self.handle, buf, need) != need:
                        raise Fast5Error('pxg_h5_read_ids failed')
                    ids = buf.raw[:need].decode(errors='replace').split('\n')[:-1] if need else []
                    self._index = {r: i for i, r in enumerate(ids)}
                    self._ids = ids
        return self._ids

    def index_of(self, read_id):
        self.read_ids
        return self._index.get(read_id, -1)

    @property
    def read_ids_array(self):
        """The read ids as a NumPy string column (built once per file)."""
        if self._ids_array is None:
            self._ids_array = np.asarray(self.read_ids) if self.n else np.zeros(0, dtype='<U1')
        return self._ids_array

    def keys_for(self, name):
        """[(name, read id), ...] of the file's reads in file order, `name` being what the caller calls the file: a
        worker batch that asks for a stretch of the file in this order is recognised by ONE list comparison."""
        keys = self._keys.get(name)
        if keys is None:
            ids = self.read_ids
            with self._lazy:
                keys = self._keys.get(name)
                if keys is None:
                    keys = self._keys[name] = list(zip([name] * self.n, ids))
        return keys

    @property
    def info(self):
        """pxg_h5_read_info of every read of the file (one native call, cached)."""
        if self._info is None:
            with self._lazy:
                if self._info is None:
                    from . import native
                    out = np.zeros(self.n, dtype=native.H5_INFO_DTYPE)
                    import time
                    t0 = time.perf_counter()
                    rc = self.lib.pxg_h5_info_mt(self.handle, 0, self.n, out.ctypes.data, host_threads())
                    _timed('walk_s', t0)
                    if rc:
                        raise Fast5Error('pxg_h5_info failed ({})'.format(rc))
                    self._info = out
        return self._info

    def signal(self, i):
        info = self.info[i]
        if info['status']:
            raise Fast5Error(info['error'].decode(errors='replace'))
        return load_signals([self], [i], [int(info['n_samples'])])[0]

    def basecall(self, i):
        """get_basecall()-style dict of read i (fast5_file.py:133-181), None if not basecalled."""
        import ctypes as C
        info = self.info[i]
        if info['status']:
            raise Fast5Error(info['error'].decode(errors='replace'))
        if not info['bc_present']:
            return None
        text = C.create_string_buffer(2 * int(info['bc_seq_len']) + 16)
        n_mv = max(int(info['bc_n_moves']), 0)
        move = np.zeros(max(n_mv, 1), dtype=np.uint8)
        pms = np.zeros(max(n_mv, 1), dtype=np.float64)
        has = C.c_int32(0)
        rc = self.lib.pxg_h5_basecall(self.handle, i, len(text), text, len(move), move.ctypes.data,
                                      pms.ctypes.data, C.byref(has))
        if rc:
            raise Fast5Error((self.lib.pxg_h5_last_error() or b'').decode(errors='replace'))
        seq, qual = text.value.decode('ascii').split('\n')        # (validated printable ASCII by the library)
        kind = TABLE_KINDS[int(info['bc_table'])] or None
        out = {'sequence': seq, 'qstring': qual, 'block_stride': int(info['bc_block_stride']),
               'sequence_length': int(info['bc_sequence_length']),
               'mean_qscore': float(info['bc_mean_qscore']), 'num_events': int(info['bc_num_events']),
               'first_sample_template': int(info['bc_first_sample']), 'table': kind,
               'move': move[:n_mv].tolist() if info['bc_n_moves'] >= 0 else None,
               'p_model_state': pms[:n_mv].tolist() if has.value else None}
        if kind in ('albacore', 'guppy_events'):       # tables whose own columns are consumed downstream
            out['events'] = self.events(i, n_mv)
        return out

    def events(self, i, n_rows):
        """The consumed columns of read i's Events table in the file's own dtypes (pxg_h5_events)."""
        n, width = max(int(n_rows), 1), 8
        info = np.zeros(21, dtype=np.int32)
        for _attempt in range(2):
            # sized from the move count and the usual 5-mer text first; the reader fills `info` and returns the row count
            # it needs BEFORE it converts anything, so a longer table or a wider model_state column costs one more call
            cols = {k: np.zeros(n, dtype=np.float64) for k in EVENT_NUMERIC}
            text = np.zeros(n, dtype='S{}'.format(width))
            got = self.lib.pxg_h5_events(self.handle, i, n, info.ctypes.data,
                                         *[cols[k].ctypes.data for k in EVENT_NUMERIC], text.ctypes.data, width)
            need_w = int(info[19]) if info[18] >= 0 else 0
            if got > n or need_w > width:
                n, width = max(int(got), n, 1), max(need_w, width)
                continue
            break
        if got < 0 or got > n:
            raise Fast5Error((self.lib.pxg_h5_last_error() or b'').decode(errors='replace') or 'Events table changed size')
        out = {}
        for k, name in enumerate(EVENT_NUMERIC):
            cls, size, signed = info[3 * k:3 * k + 3].tolist()
            if cls >= 0:
                out[name] = cols[name][:got].astype(np.dtype(('f' if cls == 1 else ('i' if signed else 'u')) + str(size)))
        if info[18] >= 0:
            out['model_state'] = text[:got].astype('S{}'.format(int(info[19])))
        return out


_OPEN, _OPEN_LOCK, _OPEN_MAX = OrderedDict(), threading.Lock(), 128
_OPEN_BYTES = 0                      # mapped bytes of the files in _OPEN
_OPENING = {}                        # key -> Event: files some thread is opening right now (open_fast5)
_OPEN_MAX_BYTES = int(os.environ.get('PXG_FAST5_CACHE_BYTES', 8 << 30))     # mapped file bytes kept open


def clear_open_cache():
    """Forget every cached Fast5File (tools that time a cold reader; files in use stay open through their references)."""
    global _OPEN_BYTES
    with _OPEN_LOCK:
        _OPEN.clear()
        _OPEN_BYTES = 0


def open_fast5(path):
    """Fast5File of `path` from a small LRU cache (a multi-read file serves thousands of reads;
    keyed by path + mtime + size so that a rewritten file is opened again)."""
    st = os.stat(path)
    key = (path, st.st_mtime_ns, st.st_size)
    global _OPEN_BYTES
    while True:
        with _OPEN_LOCK:
            f = _OPEN.get(key)
            if f is not None:
                _OPEN.move_to_end(key)
                return f
            busy = _OPENING.get(key)
            if busy is None:                  # this thread opens it; worker threads that reach the same file meanwhile wait
                _OPENING[key] = busy = threading.Event()      # (32 of them walking a 4 000-read file at once is 32 walks)
                break
        busy.wait()
    try:
        f = Fast5File(path)
        f.size = st.st_size
        with _OPEN_LOCK:
            if key not in _OPEN:
                _OPEN_BYTES += f.size
            _OPEN[key] = f
            # a run walks its files once: what stays open (= mapped) is bounded in files and in bytes,
            # the most recent ones first (a batch in flight keeps its own references)
            while len(_OPEN) > _OPEN_MAX or (_OPEN_BYTES > _OPEN_MAX_BYTES and len(_OPEN) > 2):
                _OPEN_BYTES -= _OPEN.popitem(last=False)[1].size
        return f
    finally:                                  # (an open that failed: the waiting threads try -- and fail -- themselves)
        with _OPEN_LOCK:
            _OPENING.pop(key, None)
        busy.set()


class OpenedFiles:
    """Many FAST5 files opened by ONE native call (pxg_h5_open_many: a directory of single-read files costs a worker
    batch thousands of opens): handles, per-file outcome and -- for files that hold one read in the single-read layout --
    that read's pxg_h5_read_info.  Closes what it opened when the last reference goes."""

    def __init__(self, paths, threads=None):
        from . import native
        import ctypes as C
        import time
        self.lib = native.load_text_library()
        self.paths = list(paths)
        n = self.n = len(self.paths)
        self.handles = np.zeros(n, dtype=np.uintp)
        self.rc = np.zeros(n, dtype=np.int32)
        self.n_reads = np.zeros(n, dtype=np.int64)
        self.multi = np.zeros(n, dtype=np.int32)
        self.info = np.zeros(n, dtype=native.H5_INFO_DTYPE)
        self.error = np.zeros(n, dtype='S160')
        encoded = [os.fsencode(p) for p in self.paths]
        argv = (C.c_char_p * max(n, 1))(*encoded)
        t0 = time.perf_counter()
        rc = self.lib.pxg_h5_open_many(n, argv, threads or host_threads(), self.handles.ctypes.data, self.rc.ctypes.data,
                                       self.n_reads.ctypes.data, self.multi.ctypes.data, self.info.ctypes.data,
                                       self.error.ctypes.data)
        _timed('walk_s', t0)
        if rc:
            self.close()
            raise Fast5Error('pxg_h5_open_many failed ({})'.format(rc))

    def file(self, k):
        """Fast5File of file k (it does not own the handle: this object does)."""
        one = self.info[k:k + 1] if (not self.multi[k] and self.n_reads[k] == 1) else None
        return Fast5File.from_handle(self.paths[k], self.handles[k], self.n_reads[k], self.multi[k], one, owner=self)

    def close(self):
        handles, self.handles = getattr(self, 'handles', None), None
        if handles is not None and len(handles):
            self.lib.pxg_h5_close_many(len(handles), handles.ctypes.data)

    __del__ = close


_HOST_CORES = None


def host_threads():
    """Threads for the batch decoders: the cores this process may use, capped by a cgroup quota."""
    global _HOST_CORES
    if _HOST_CORES is None:          # (asked once per process: a batch of single-read files used to ask twice per file)
        n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        try:
            with open('/sys/fs/cgroup/cpu.max') as fh:
                q, period = fh.read().split()
            if q != 'max':
                n = min(n, max(int(float(q) / float(period)), 1))
        except (OSError, ValueError):
            pass
        _HOST_CORES = n
    n = _HOST_CORES
    # several ranks on one host (torchrun sets LOCAL_WORLD_SIZE) share its cores: an equal share
    # each, so that N loader pools do not oversubscribe the sockets they were pinned to
    local = int(os.environ.get('LOCAL_WORLD_SIZE', 1) or 1)
    if local > 1:
        n = min(n, max((os.cpu_count() or n) // local, 1))
    return max(1, min(n, int(os.environ.get('PXG_HOST_THREADS', 32))))


def _handles(files):
    """What the batch decoders take as `files`: a list of Fast5File (one per read) or the uintp array of their handles."""
    if isinstance(files, np.ndarray):
        return files.ctypes.data
    import ctypes as C
    return (C.c_void_p * len(files))(*[f.handle.value for f in files])


def load_signals(files, index, n_samples, arena=None, dst_start=None, threads=None):
    """int16 samples of reads (files[k], index[k]) decoded on host threads.  Without `arena`:
    a list of arrays; with it: written at dst_start[k] (returns the per-read status array)."""
    from . import native
    lib = native.load_text_library()
    n = len(files)
    ns = np.ascontiguousarray(n_samples, dtype=np.int64)
    idx = np.ascontiguousarray(index, dtype=np.int64)
    own = arena is None
    if own:
        dst_start = np.zeros(n, dtype=np.int64)
        np.cumsum(ns[:-1], out=dst_start[1:])
        arena = np.empty(int(ns.sum()), dtype=np.int16)
    dst = np.ascontiguousarray(dst_start, dtype=np.int64)
    status = np.zeros(n, dtype=np.int32)
    if n:
        import time
        t0 = time.perf_counter()
        lib.pxg_h5_load_signals(n, _handles(files), idx.ctypes.data, dst.ctypes.data, ns.ctypes.data,
                                arena.ctypes.data, threads or host_threads(), status.ctypes.data)
        _timed('signals_s', t0)
    if own:
        if status.any():
            raise Fast5Error('signal of read {} cannot be decoded (code {})'.format(
                int(np.nonzero(status)[0][0]), int(status[status != 0][0])))
        return [arena[dst[k]:dst[k] + ns[k]] for k in range(n)]
    return status


def _text_column(col, encoding):
    """Fixed-width bytes column -> str column (a run's run id / sample id columns hold one
    value: decoded once)."""
    if len(col) and (col == col[0]).all():
        return np.full(len(col), col[0].decode(encoding, errors='replace'))
    if encoding == 'ascii':
        try:
            return col.astype('U{}'.format(max(col.dtype.itemsize, 1)))      # (raises on a byte above 127)
        except UnicodeDecodeError:
            pass
    return np.asarray([b.decode(encoding, errors='replace') for b in col.tolist()])


def decode_layout(p, threads=None):
    """Samples and basecall text of the reads of layout `p` (Fast5Batch.plan, FileRunColumns.layout) into its arenas, on
    host threads (two native calls)."""
    from . import native
    lib = native.load_text_library()
    if not p['n']:
        return
    import time
    threads = threads or host_threads()
    t0 = time.perf_counter()
    lib.pxg_h5_load_signals(p['n'], p['handles'].ctypes.data, p['index'].ctypes.data, p['dst'].ctypes.data,
                            p['n_samples'].ctypes.data, p['arena'].ctypes.data, threads, p['signal_status'].ctypes.data)
    _timed('signals_s', t0)
    t0 = time.perf_counter()
    lib.pxg_h5_basecall_many(p['n'], p['handles'].ctypes.data, p['index'].ctypes.data, p['seq_start'].ctypes.data,
                             p['seq_len'].ctypes.data, p['seq_arena'].ctypes.data, p['qual_arena'].ctypes.data,
                             p['move_start'].ctypes.data, p['n_moves'].ctypes.data, p['move_arena'].ctypes.data, threads,
                             p['basecall_status'].ctypes.data)
    _timed('text_s', t0)


class Fast5Batch:
    """Many FAST5 reads as COLUMNS: what ReadBundle holds for a bundle's reads, built for one
    worker batch from the files themselves -- metadata by one native call per file, signals and
    basecall text decoded on host threads into arenas (the sample arena may be a page-locked
    staging buffer).  `as_bundle()` is a ReadBundle, so the loader, the status rules and the
    result-dict builder take exactly the paths they take for bundle reads."""

    def __init__(self, files, index, names, read_ids=None):
        from . import native
        self._files, self.index, self.names = list(files), np.asarray(index, dtype=np.int64), list(names)
        self.read_ids = read_ids          # (known to the caller that looked the reads up by id)
        self.runs = self.handles = self.name_array = self.id_array = None
        info = np.zeros(len(self.files), dtype=native.H5_INFO_DTYPE)
        # reads of one file usually come as one run: a slice assignment per run
        k, n = 0, len(self.files)
        while k < n:
            f, e = self.files[k], k + 1
            while e < n and self.files[e] is f:
                e += 1
            info[k:e] = f.info[self.index[k:e]]
            k = e
        self._info = info

    @property
    def info(self):
        if self._info is None:
            self._info = np.concatenate([f.info[i0:i0 + count] for f, _, i0, count in self.runs])
        return self._info

    @classmethod
    def from_runs(cls, runs):
        """The batch of a request that is stretches of multi-read files in file order -- runs = [(Fast5File, name,
        first read, count)] -- which is what a worker batch of a run looks like: every per-read Python list of the
        general constructor becomes a slice or a repeat (10 000 reads: ~1 ms instead of ~8)."""
        self = cls.__new__(cls)
        self.runs, self._files = list(runs), None
        lens = [count for _, _, _, count in runs]
        self.index = np.concatenate([np.arange(i0, i0 + count, dtype=np.int64) for _, _, i0, count in runs])
        self._info = None                  # (info: the metadata records of the batch's reads, copied when somebody asks)
        self.names, self.read_ids = [], []
        for f, name, i0, count in runs:
            self.names += [name] * count
            self.read_ids += f.read_ids[i0:i0 + count]
        self.handles = np.repeat(np.array([f.handle.value for f, _, _, _ in runs], dtype=np.uintp), lens)
        self.name_array = self.id_array = None        # (run_arrays(): only a batch that describes itself needs them)
        return self

    def run_arrays(self):
        """File names and read ids of a batch of runs as NumPy string columns."""
        if self.runs is not None and self.name_array is None:
            lens = [count for _, _, _, count in self.runs]
            self.name_array = np.repeat(np.asarray([name for _, name, _, _ in self.runs]), lens)
            self.id_array = np.concatenate([f.read_ids_array[i0:i0 + count] for f, _, i0, count in self.runs])

    @classmethod
    def from_opened(cls, opened, which, names, read_ids):
        """The batch of the single-read files `which` of an OpenedFiles (one read each: read 0), without a Python
        object per file."""
        self = cls.__new__(cls)
        self.runs, self._files, self.opened, self.which = None, None, opened, np.asarray(which, dtype=np.int64)
        self.index = np.zeros(len(self.which), dtype=np.int64)
        self._info = opened.info[self.which]
        self.names, self.read_ids = list(names), list(read_ids)
        self.handles = np.ascontiguousarray(opened.handles[self.which])
        self.name_array = self.id_array = None
        return self

    @property
    def files(self):
        """The Fast5File of every read (kept alive by the batch either way)."""
        if self._files is None:
            if self.runs is None:
                self._files = [self.opened.file(int(k)) for k in self.which]
            else:
                self._files = [f for f, _, _, count in self.runs for _ in range(count)]
        return self._files

    def plan(self, reserve=None, arenas=True):
        """The layout of the batch's bundle -- where every read's samples, sequence, quality string and moves go, in
        arenas made here (`reserve(n_samples)` -> the int16 arena: a staging buffer) -- from the metadata alone, before
        anything is decoded.  decode() fills the arenas, bundle() is the ReadBundle over them.  arenas=False: the
        layout alone (empty arenas: FileRunColumns describes a whole file this way)."""
        n = len(self.index)
        handles = self.handles if self.handles is not None else \
            np.array([f.handle.value for f in self.files], dtype=np.uintp)
        if self.runs is not None and arenas and _RUN_COLUMNS_IN_BATCHES and len(self.runs) <= 64:
            # (stretches of multi-read files: the lengths are slices of what is kept with every open file)
            wholes = [(file_run_columns(f, name).whole, i0, count) for f, name, i0, count in self.runs]
            ns, present, seq_len, n_moves = (
                np.concatenate([w[key][i0:i0 + count] for w, i0, count in wholes]) if len(wholes) > 1 else
                np.ascontiguousarray(wholes[0][0][key][wholes[0][1]:wholes[0][1] + wholes[0][2]])
                for key in ('n_samples', 'present', 'seq_len', 'n_moves'))
        else:
            info = self.info
            ns = info['n_samples'].astype(np.int64)
            present = info['bc_present'] != 0
            seq_len = np.where(present, info['bc_seq_len'], 0).astype(np.int64)
            n_moves = np.where(present, np.maximum(info['bc_n_moves'], 0), 0).astype(np.int64)
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(ns, out=offsets[1:])
        if not arenas:
            arena = np.zeros(0, dtype=np.int16)
        else:
            arena = reserve(int(offsets[-1])) if reserve is not None else np.empty(int(offsets[-1]), dtype=np.int16)
        seq_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(seq_len, out=seq_off[1:])
        move_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(n_moves, out=move_off[1:])
        return {'n': n, 'handles': np.ascontiguousarray(handles), 'index': np.ascontiguousarray(self.index), 'n_samples': ns,
                'offsets': offsets, 'dst': np.ascontiguousarray(offsets[:-1]), 'arena': arena, 'present': present,
                'signal_status': np.zeros(n, dtype=np.int32), 'basecall_status': np.zeros(n, dtype=np.int32),
                'seq_len': seq_len, 'n_moves': n_moves, 'seq_off': seq_off, 'move_off': move_off,
                'seq_start': np.ascontiguousarray(seq_off[:-1]), 'move_start': np.ascontiguousarray(move_off[:-1]),
                'seq_arena': np.zeros(int(seq_off[-1]) if arenas else 0, dtype=np.uint8),
                'qual_arena': np.zeros(int(seq_off[-1]) if arenas else 0, dtype=np.uint8),
                'move_arena': np.zeros(int(move_off[-1]) if arenas else 0, dtype=np.uint8)}

    def decode(self, p, threads=None):
        """Samples and basecall text of the batch into the arenas of plan `p`, on host threads (two native calls)."""
        decode_layout(p, threads)

    def bundle(self, p):
        """The ReadBundle over plan `p` (its columns are made of the metadata; the arenas are p's, decoded or about to be)."""
        from . import native
        n, offsets = p['n'], p['offsets']
        present = p['present']
        if self.runs is not None and not p.get('whole_file') and _RUN_COLUMNS_IN_BATCHES and len(self.runs) <= 64:
            # a batch of stretches of multi-read files (the session's, a worker call's that crosses a file boundary): the
            # metadata columns are slices of what is kept with every open file (FileRunColumns: made once per file), not
            # twenty conversions of the batch's own -- 10 000 reads: ~4 ms of the loader thread's Python
            parts = [(file_run_columns(f, name).meta.d, i0, count) for f, name, i0, count in self.runs]
            d = {key: (parts[0][0][key][parts[0][1]:parts[0][1] + parts[0][2]] if len(parts) == 1 else
                       np.concatenate([m[key][i0:i0 + count] for m, i0, count in parts]))
                 for key in _FILE_COLUMNS}
            d.update({'arena': p['arena'][:offsets[-1]] if len(p['arena']) else p['arena'], 'offsets': offsets,
                      'broken_files': np.array([], dtype='<U1'), 'bundle_version': np.int64(2),
                      'seq_offsets': p['seq_off'], 'seq_arena': p['seq_arena'], 'qual_arena': p['qual_arena'],
                      'move_offsets': p['move_off'], 'move_arena': p['move_arena']})
            bundle = _Fast5BatchBundle(d, self)
            bundle.signal_status, bundle.basecall_status = p['signal_status'], p['basecall_status']
            return bundle
        info = self.info
        calib = np.zeros(n, dtype=native.CALIB_DTYPE)
        for name in ('range', 'digitisation', 'offset', 'sampling_rate'):
            calib[name] = info['calib'][name]
        self.run_arrays()
        d = {'arena': p['arena'][:offsets[-1]] if len(p['arena']) else p['arena'], 'offsets': offsets, 'calib': calib,
             'filename': self.name_array if self.name_array is not None else np.asarray(self.names),
             'read_id': self.id_array if self.id_array is not None else (
                 np.asarray(self.read_ids) if self.read_ids is not None else _text_column(info['read_id'], 'ascii')),
             'duration': info['duration'].astype(np.int64), 'start_time': info['start_time'].astype(np.int64),
             'channel_number': _text_column(info['channel_number'], 'ascii'),
             'run_id': _text_column(info['run_id'], 'ascii'),
             'sample_id': _text_column(info['sample_id'], 'utf-8'),
             'broken_files': np.array([], dtype='<U1'), 'bundle_version': np.int64(2),
             'bc_present': present, 'bc_sequence_length': info['bc_sequence_length'].astype(np.int64),
             'bc_mean_qscore': info['bc_mean_qscore'].astype(np.float64),
             'bc_num_events': info['bc_num_events'].astype(np.int64),
             'bc_first_sample': info['bc_first_sample'].astype(np.int64),
             'bc_block_stride': info['bc_block_stride'].astype(np.int32), 'bc_table': info['bc_table'].astype(np.int8),
             'bc_n_moves': np.where(present, info['bc_n_moves'], -1).astype(np.int64),
             'bc_move_sum': info['bc_move_sum'].astype(np.int64),
             'seq_offsets': p['seq_off'], 'seq_arena': p['seq_arena'], 'qual_arena': p['qual_arena'],
             'move_offsets': p['move_off'], 'move_arena': p['move_arena']}
        bundle = _Fast5BatchBundle(d, self)
        bundle.signal_status, bundle.basecall_status = p['signal_status'], p['basecall_status']
        return bundle

    def as_bundle(self, reserve=None, threads=None):
        """ReadBundle over the batch (reads whose info failed must have been left out by the
        caller).  `reserve(n_samples)` -> int16 arena to decode into (a staging buffer)."""
        p = self.plan(reserve)
        self.decode(p, threads)
        return self.bundle(p)


# the columns of a bundle that are functions of a read's metadata alone (Fast5Batch.bundle)
_FILE_COLUMNS = ('calib', 'filename', 'read_id', 'duration', 'start_time', 'channel_number', 'run_id', 'sample_id',
                 'bc_present', 'bc_sequence_length', 'bc_mean_qscore', 'bc_num_events', 'bc_first_sample', 'bc_block_stride',
                 'bc_table', 'bc_n_moves', 'bc_move_sum')
_RUN_COLUMNS_IN_BATCHES = os.environ.get('PXG_NO_RUN_COLUMNS_IN_BATCHES') is None        # (A/B, tests)


class FileRunColumns:
    """What worker calls that are runs of ONE multi-read file share, made once per (file, name under which it is asked
    for): the file's metadata as the columns of a read bundle over ALL its reads (no samples, no text: `meta`), hence
    its plain-run columns (ReadBundle.plain_run_columns), and the layout every call's arenas follow.  A call over reads
    [i0, i0 + n) then costs a dozen slices (layout) instead of the ~70 small NumPy operations that build a bundle of
    its own -- the interpreter lock is what bounds worker threads in reference-sized calls."""

    def __init__(self, f, name):
        batch = Fast5Batch.from_runs([(f, name, 0, f.n)])
        self.whole = batch.plan(arenas=False)
        self.whole['whole_file'] = True    # (bundle(): made from the metadata itself -- this IS what batches slice)
        self.meta = batch.bundle(self.whole)
        self.meta.batch = None             # (kept with the open file: no reference back to it, it closes when its last user lets go)
        self.handle = f.handle.value       # (a call holds the file itself for as long as it uses the handle: CallBundle.runs)

    def plain(self, scaler_cfg):
        return self.meta.plain_run_columns(scaler_cfg)

    def texts(self):
        """(channel numbers, run ids, sample ids) of the file's reads as Python lists (made once)."""
        lists = self.__dict__.get('_texts')
        if lists is None:
            d = self.meta.d
            lists = self._texts = tuple(d[name].tolist() for name in ('channel_number', 'run_id', 'sample_id'))
        return lists

    def layout(self, i0, n, reserve):
        """Fast5Batch.plan's dict for the reads [i0, i0 + n) of the file: arenas of the call's own, positions counted from
        its first read."""
        w, run = self.whole, slice(i0, i0 + n)
        offsets = w['offsets'][i0:i0 + n + 1] - w['offsets'][i0]
        seq_off = w['seq_off'][i0:i0 + n + 1] - w['seq_off'][i0]
        move_off = w['move_off'][i0:i0 + n + 1] - w['move_off'][i0]
        return {'n': n, 'handles': np.full(n, self.handle, dtype=np.uintp), 'index': np.arange(i0, i0 + n, dtype=np.int64),
                'n_samples': w['n_samples'][run], 'offsets': offsets, 'dst': offsets[:-1],
                'arena': reserve(int(offsets[-1])), 'present': w['present'][run],
                'signal_status': np.zeros(n, dtype=np.int32), 'basecall_status': np.zeros(n, dtype=np.int32),
                'seq_len': w['seq_len'][run], 'n_moves': w['n_moves'][run], 'seq_off': seq_off, 'move_off': move_off,
                'seq_start': seq_off[:-1], 'move_start': move_off[:-1],
                'seq_arena': np.zeros(int(seq_off[-1]), dtype=np.uint8), 'qual_arena': np.zeros(int(seq_off[-1]), dtype=np.uint8),
                'move_arena': np.zeros(int(move_off[-1]), dtype=np.uint8)}


def file_run_columns(f, name):
    """The FileRunColumns of multi-read file `f` asked for as `name` (kept with the open file)."""
    cache = f.__dict__.setdefault('_run_columns', {})
    cols = cache.get(name)
    if cols is None:
        f.read_ids, f.info                     # (each takes the file's lock itself)
        with f._lazy:
            cols = cache.get(name)
            if cols is None:
                cols = cache[name] = FileRunColumns(f, name)
    return cols


class Fast5Reader:
    """The reference's per-read reader surface (fast5_file.py:60-131) on the native file."""

    def __init__(self, path, read_id):
        self.path = path
        try:
            self.file = open_fast5(path)
        except Fast5Error as exc:
            if getattr(exc, 'code', 0) == -6 and h5py is not None:      # a layout the native reader declines
                self.__class__ = H5pyFast5Reader
                H5pyFast5Reader.__init__(self, path, read_id)
                return
            raise OSError(str(exc))
        f = self.file
        if f.multi:
            self.i = f.index_of(read_id)
            if self.i < 0:
                raise KeyError("Unable to open object (object 'read_{}' doesn't exist)".format(read_id))
        else:
            if not f.n:
                raise KeyError("Unable to open object (object 'Reads' doesn't exist)")
            self.i = 0
        info = f.info[self.i]
        if info['status']:
            if info['status'] == -6 and h5py is not None:
                self.__class__ = H5pyFast5Reader
                H5pyFast5Reader.__init__(self, path, read_id)
                return
            raise OSError(info['error'].decode(errors='replace'))
        self.duration, self.start_time = int(info['duration']), int(info['start_time'])
        file_read_id = info['read_id'].decode()
        self.read_id = file_read_id if read_id is None else read_id
        if file_read_id != self.read_id:
            raise ValueError('Unexpected read {} found in {}'.format(file_read_id, path))
        self.channel_number = info['channel_number'].decode()
        cal = info['calib']
        self.digitization, self.offset = float(cal['digitisation']), float(cal['offset'])
        self.range, self.sampling_rate = float(cal['range']), float(cal['sampling_rate'])
        self.run_id, self.sample_id = info['run_id'].decode(), info['sample_id'].decode(errors='replace')

    def close(self):
        self.file = None

    def get_raw_int16(self):
        return self.file.signal(self.i)

    def get_basecall(self, analysis_group='Basecall_1D'):
        return self.file.basecall(self.i)


class H5pyFast5Reader:
    """h5py-backed reader: only for files the native reader declines, where h5py exists."""

    def __init__(self, path, read_id):
        if h5py is None:
            raise RuntimeError('h5py is not installed and the native FAST5 reader declined this file')
        self.path, self.read_id = path, read_id
        self.handle = h5py.File(path, 'r')
        if 'UniqueGlobalKey' not in self.handle:          # multi-read (:71-75)
            base = 'read_{}/'.format(read_id)
            self.read_node, self.channel_node = base + 'Raw', base + 'channel_id'
            self.tracking_node, self.analyses_node = base + 'tracking_id', base + 'Analyses'
        else:                                             # single-read (:76-82)
            first = next(iter(self.handle['Raw/Reads'].keys()))
            self.read_node = 'Raw/Reads/' + first
            self.channel_node = 'UniqueGlobalKey/channel_id'
            self.tracking_node = 'UniqueGlobalKey/tracking_id'
            self.analyses_node = 'Analyses'
        s = lambda v: v.decode() if isinstance(v, bytes) else str(v)
        sig = self.handle[self.read_node].attrs
        self.duration = int(sig['duration'])
        self.start_time = int(sig['start_time'])
        file_read_id = s(sig['read_id'])
        if self.read_id is None:
            self.read_id = file_read_id
        elif file_read_id != self.read_id:
            raise ValueError('Unexpected read {} found in {}'.format(file_read_id, path))
        ch = self.handle[self.channel_node].attrs
        self.channel_number = s(ch['channel_number'])
        self.digitization = float(ch['digitisation'])
        self.offset = float(ch['offset'])
        self.range = float(ch['range'])
        self.sampling_rate = float(ch['sampling_rate'])
        tr = self.handle[self.tracking_node].attrs
        self.run_id, self.sample_id = s(tr['run_id']), s(tr['sample_id'])

    def close(self):
        handle, self.handle = self.handle, None
        if handle is not None:
            handle.close()

    def get_raw_int16(self):
        return np.asarray(self.handle[self.read_node + '/Signal'][()], dtype=np.int16)

    def get_basecall(self, analysis_group='Basecall_1D'):
        analnode = self.handle.get(self.analyses_node)
        groups = sorted(n for n in (analnode or ()) if n.startswith(analysis_group))
        if not groups:
            return None
        analyses = analnode[groups[-1]]            # highest-numbered group wins (:143)
        groupno = analyses.name.rsplit('_', 1)[-1]
        seg = analnode['Segmentation_{}/Summary/segmentation'.format(groupno)].attrs
        fq = analyses['BaseCalled_template/Fastq'][()]
        fq = (fq.decode() if isinstance(fq, bytes) else str(fq)).split('\n')
        sm = analyses['Summary/{}_template'.format(analysis_group.lower())].attrs
        # event mapping (fast5_file.py:166-181): `Events' (albacore, guppy < 2.3.7) wins over
        # `Move' (guppy >= 2.3.7); Guppy tables only say which blocks moved, the block means
        # are re-cut from the raw signal (on the GPU here)
        table, move, pms, events = None, None, None, None
        if 'BaseCalled_template/Events' in analyses:
            ev = analyses['BaseCalled_template/Events'][()]
            cols = ev.dtype.names or ()
            if len(cols) <= 3 and 'move' in cols:
                table = 'guppy_events'
            elif len(cols) == 14:
                table = 'albacore'
            else:
                table = 'unsupported'
            if 'move' in cols:
                move = ev['move'].tolist()
            if 'p_model_state' in cols:
                pms = ev['p_model_state'].astype(np.float64).tolist()
            if table in ('albacore', 'guppy_events'):
                events = {k: np.array(ev[k]) for k in EVENT_COLUMNS if k in cols}
        elif 'BaseCalled_template/Move' in analyses:
            table, move = 'move', analyses['BaseCalled_template/Move'][()].tolist()
        return {**({'events': events} if events is not None else {}), 'sequence': fq[1], 'qstring': fq[3],
                'block_stride': int(sm.get('block_stride', 15)),
                'sequence_length': int(sm['sequence_length']),
                'mean_qscore': float(sm['mean_qscore']),
                'num_events': int(seg['num_events_template']),
                'first_sample_template': int(seg['first_sample_template']),
                'table': table, 'move': move, 'p_model_state': pms}


_WARM = {'pid': None, 'pool': None}


def warm_file(f, name):
    """Have what worker calls need of multi-read file `f` (asked for as `name`) made in the background: its metadata
    columns (~6 us per read) and its run columns.  A 4 000-read file takes 15-40 ms to describe, and worker threads reach
    a new file together: whoever lists the input ahead of them -- the reference's scanner calls get_read_ids file by
    file (pipeline.py:321-324) -- starts this, and the first call on the file finds it done or nearly so.  One helper
    thread per process (a forked child starts its own)."""
    from concurrent.futures import ThreadPoolExecutor
    with _OPEN_LOCK:
        if _WARM['pid'] != os.getpid():
            _WARM['pid'], _WARM['pool'] = os.getpid(), ThreadPoolExecutor(1, thread_name_prefix='pxg-warm')
        pool = _WARM['pool']

    def job():
        try:
            f.info
            f.keys_for(name)
            file_run_columns(f, name)
        except Exception:             # noqa: BLE001  (the call that needs them reports what is wrong with the file)
            pass
    try:
        return pool.submit(job)
    except RuntimeError:              # (interpreter shutting down)
        return None


def get_read_ids(filename, basedir, bundle=None, warm=None):
    """(filename, read_id) pairs of one input file (fast5_file.py:37-58).  `warm`: start warm_file for a multi-read
    file; by default when this process runs worker calls itself (a thread pool in place of the reference's process pool:
    INTEGRATION.md section 6) -- in a parent of worker PROCESSES nobody would use what it makes."""
    if bundle is not None and bundle.has_file(filename):
        return bundle.read_ids_of(filename)
    path = os.path.join(basedir, filename) if basedir is not None else filename
    try:
        f = open_fast5(path)
        if not f.multi:
            return [(filename, rid) for rid in f.read_ids]
        if warm is None:
            import sys
            warm = f.n >= 64 and '__poreplex_amd_persistence' in sys.modules and not os.environ.get('PXG_NO_FILE_WARMUP')
        if warm:
            warm_file(f, filename)
        return list(f.keys_for(filename))      # (the tuples worker calls are compared with: fast5_runs)
    except Fast5Error as exc:
        if getattr(exc, 'code', 0) != -6 or h5py is None:
            raise OSError(str(exc))
    with h5py.File(path, 'r') as f5:
        if 'UniqueGlobalKey' in f5:
            try:
                first = next(iter(f5['Raw/Reads'].values()))
                rid = first.attrs['read_id']
                return [(filename, rid.decode() if isinstance(rid, bytes) else str(rid))]
            except KeyError:
                return []
        return [(filename, node[5:]) for node in f5 if node.startswith('read_')]


def get_read_ids_many(filenames, basedir, chunk=4096):
    """get_read_ids for many files, in their order: the files are opened `chunk` at a time by one native call
    (OpenedFiles) and a file that holds one read in the single-read layout answers from there -- a directory of
    single-read files is listed without a Python round per file; every other file (multi-read, unreadable, odd) goes
    through get_read_ids itself, which answers or raises as before."""
    out = []
    filenames = list(filenames)
    for lo in range(0, len(filenames), chunk):
        names = filenames[lo:lo + chunk]
        opened = OpenedFiles([os.path.join(basedir, n) if basedir is not None else n for n in names])
        try:
            single = (opened.rc == 0) & (opened.multi == 0) & (opened.n_reads == 1) & (opened.info['status'] == 0)
            ids = opened.info['read_id'].tolist()
            for k, name in enumerate(names):
                if single[k]:
                    try:
                        out.append((name, ids[k].decode('ascii')))
                        continue
                    except UnicodeDecodeError:
                        pass
                out.extend(get_read_ids(name, basedir, warm=False))      # (a listing of the whole run: nothing to warm yet)
        finally:
            opened.close()
    return out


def open_read(fullpath, filename, read_id, bundle=None):
    if bundle is not None and bundle.has_file(filename):
        return bundle.reader(filename, read_id)
    return Fast5Reader(fullpath, read_id)
