"""Read sources for the hot path: FAST5 (h5py, optional) and .pxr.npz bundles.

Mirrors what the per-read processor needs from the reference's
poreplex/fast5_file.py (`Fast5Reader`: metadata :97-120, raw samples :122-131,
basecall summary :133-164).  The pA conversion itself is NOT done here: the
reader hands int16 DAQ samples + calibration to the GPU (kernel a1).

A *bundle* (`*.pxr.npz`) is this build's array container for many reads: the
GPU boxes have no h5py, and the golden tests feed the same reads the reference
saw as FAST5 files (tools/make_golden.py writes both).
"""
import json
import os

import numpy as np

try:  # optional: only needed for real FAST5 input
    import h5py
except ImportError:  # pragma: no cover
    h5py = None

__all__ = ['get_read_ids', 'open_read', 'ReadBundle', 'Fast5Reader']


class ReadBundle:
    """Random access to the reads of one .pxr.npz file by (filename, read_id)."""

    def __init__(self, path):
        with np.load(path, allow_pickle=False) as npz:
            self.d = {k: npz[k] for k in npz.files}
        self.index = {(str(f), str(r)): i
                      for i, (f, r) in enumerate(zip(self.d['filename'], self.d['read_id']))}
        self.by_file = {}
        for (f, r), i in self.index.items():
            self.by_file.setdefault(f, []).append(i)
        # files that exist but cannot be opened (the corrupt-FAST5 case)
        self.broken = set(str(f) for f in self.d.get('broken_files', []))

    def has_file(self, filename):
        return filename in self.by_file or filename in self.broken

    def read_ids(self, filename):
        return [(filename, str(self.d['read_id'][i])) for i in self.by_file.get(filename, [])]

    def reader(self, filename, read_id):
        key = (filename, read_id)
        if filename in self.broken:
            raise OSError('Unable to open file {!r} (file signature not found)'.format(filename))
        if key not in self.index:
            if filename in self.by_file:     # fast5_file.py:104-107
                raise ValueError('Unexpected read {} found in {}'.format(
                    self.d['read_id'][self.by_file[filename][0]], filename))
            raise FileNotFoundError(filename)
        return BundleReader(self, self.index[key])


class BundleReader:
    """Same attribute surface as the reference's Fast5Reader."""

    def __init__(self, bundle, i):
        d = bundle.d
        self.path, self.i, self.d = str(d['filename'][i]), i, d
        self.read_id = str(d['read_id'][i])
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
        o = self.d['offsets']
        return self.d['arena'][o[self.i]:o[self.i + 1]]

    def get_basecall(self):
        """Summary of Analyses/Basecall_1D_* (fast5_file.py:133-164); None if absent."""
        js = str(self.d['basecall'][self.i]) if 'basecall' in self.d else ''
        if not js:
            return None
        bc = json.loads(js)
        return {'sequence': bc['sequence'], 'qstring': bc['qstring'],
                'block_stride': int(bc.get('block_stride', 15)),
                'sequence_length': int(bc['sequence_length']),
                'mean_qscore': float(np.float32(bc['mean_qscore'])),
                'num_events': int(bc['num_events']),
                'first_sample_template': int(bc['first_sample_template']),
                'table': bc.get('table', 'move' if bc.get('move') is not None else None),
                'move': bc.get('move'), 'p_model_state': bc.get('p_model_state')}


class Fast5Reader:
    """h5py-backed reader (single- and multi-read FAST5)."""

    def __init__(self, path, read_id):
        if h5py is None:
            raise RuntimeError('h5py is not installed: FAST5 input is unavailable here; '
                               'use a .pxr.npz read bundle (config["read_bundle"])')
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
        table, move, pms = None, None, None
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
        elif 'BaseCalled_template/Move' in analyses:
            table, move = 'move', analyses['BaseCalled_template/Move'][()].tolist()
        return {'sequence': fq[1], 'qstring': fq[3],
                'block_stride': int(sm.get('block_stride', 15)),
                'sequence_length': int(sm['sequence_length']),
                'mean_qscore': float(sm['mean_qscore']),
                'num_events': int(seg['num_events_template']),
                'first_sample_template': int(seg['first_sample_template']),
                'table': table, 'move': move, 'p_model_state': pms}


def get_read_ids(filename, basedir, bundle=None):
    """(filename, read_id) pairs of one input file (fast5_file.py:37-58)."""
    if bundle is not None and bundle.has_file(filename):
        return bundle.read_ids(filename)
    path = os.path.join(basedir, filename) if basedir is not None else filename
    if h5py is None:
        raise RuntimeError('h5py is not installed')
    with h5py.File(path, 'r') as f5:
        if 'UniqueGlobalKey' in f5:
            try:
                first = next(iter(f5['Raw/Reads'].values()))
                rid = first.attrs['read_id']
                return [(filename, rid.decode() if isinstance(rid, bytes) else str(rid))]
            except KeyError:
                return []
        return [(filename, node[5:]) for node in f5 if node.startswith('read_')]


def open_read(fullpath, filename, read_id, bundle=None):
    if bundle is not None and bundle.has_file(filename):
        return bundle.reader(filename, read_id)
    return Fast5Reader(fullpath, read_id)
