#!/opt/conda/bin/python3.9
"""One-off validation of poreplex_amd.fast5_file.Fast5Reader / get_read_ids against the
REAL reference reader on synthetic single- and multi-read FAST5 files (h5py only exists
under /opt/conda/bin/python3.9; the pytest suite runs where it does not, so this is a
script, not a test).  Prints OK lines; exits non-zero on any difference."""
import os
import sys
import tempfile

import h5py
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REPO)
TMP = tempfile.mkdtemp(prefix='pxg_f5_')

_attr_get = h5py.AttributeManager.__getitem__
h5py.AttributeManager.__getitem__ = lambda self, name: (
    lambda v: v.encode() if isinstance(v, str) else v)(_attr_get(self, name))   # h5py 2.x behaviour

pk = os.path.join(TMP, 'poreplex')
os.makedirs(pk)
for f in ('__init__.py', 'fast5_file.py', 'utils.py'):
    os.symlink(os.path.join(REF, 'poreplex', f), os.path.join(pk, f))
sys.path.insert(0, TMP)
from poreplex import fast5_file as RF  # noqa: E402
from poreplex_amd import fast5_file as MF  # noqa: E402

rng = np.random.default_rng(5)


def fill(node_raw, node_ch, node_tr, analyses, rid, raw, meta, bc):
    node_raw.attrs['duration'] = np.uint32(len(raw))
    node_raw.attrs['start_time'] = np.uint64(meta['start_time'])
    node_raw.attrs['read_id'] = rid.encode()
    node_raw.attrs['read_number'] = np.int32(meta['read_number'])
    node_raw.create_dataset('Signal', data=raw)
    node_ch.attrs['channel_number'] = str(meta['channel']).encode()
    node_ch.attrs['digitisation'] = 8192.0
    node_ch.attrs['offset'] = 12.0
    node_ch.attrs['range'] = 1201.5
    node_ch.attrs['sampling_rate'] = 3012.0
    node_tr.attrs['run_id'] = b'runabc'
    node_tr.attrs['sample_id'] = b'sampleX'
    if bc:
        g = analyses.create_group('Basecall_1D_000')
        t = g.create_group('BaseCalled_template')
        t.create_dataset('Fastq', data=np.string_('@{}\n{}\n+\n{}\n'.format(rid, bc['seq'], bc['q'])))
        t.create_dataset('Move', data=bc['move'])
        s = g.create_group('Summary/basecall_1d_template')
        s.attrs['sequence_length'] = np.int32(len(bc['seq']))
        s.attrs['mean_qscore'] = np.float32(9.25)
        s.attrs['block_stride'] = np.int32(15)
        sg = analyses.create_group('Segmentation_000/Summary/segmentation')
        sg.attrs['num_events_template'] = np.int32(len(bc['move']))
        sg.attrs['first_sample_template'] = np.int32(bc['first'])


def basecall(n_raw):
    nb = (n_raw - 20) // 15
    move = np.zeros(nb, np.uint8)
    move[np.sort(rng.choice(nb, nb // 8, replace=False))] = 1
    move[0] = 1
    n = int(move.sum())
    return {'seq': ''.join(rng.choice(list('ACGU'), n + 4)), 'q': ''.join(chr(40 + int(x)) for x in rng.integers(0, 20, n + 4)),
            'move': move, 'first': 20}


reads = []
for i in range(3):
    raw = rng.integers(300, 900, 20000 + 1000 * i).astype(np.int16)
    reads.append(('%08d-aaaa-bbbb-cccc-%012d' % (i, i), raw, {'start_time': 1000 * i + 7, 'read_number': 10 + i, 'channel': 100 + i},
                  basecall(len(raw)) if i != 1 else None))

single = []
for rid, raw, meta, bc in reads:
    fn = 'single_%d.fast5' % meta['read_number']
    with h5py.File(os.path.join(TMP, fn), 'w') as h5:
        fill(h5.create_group('Raw/Reads/Read_%d' % meta['read_number']), h5.create_group('UniqueGlobalKey/channel_id'),
             h5.create_group('UniqueGlobalKey/tracking_id'), h5.create_group('Analyses'), rid, raw, meta, bc)
    single.append(fn)
with h5py.File(os.path.join(TMP, 'multi.fast5'), 'w') as h5:
    for rid, raw, meta, bc in reads:
        base = h5.create_group('read_' + rid)
        fill(base.create_group('Raw'), base.create_group('channel_id'), base.create_group('tracking_id'),
             base.create_group('Analyses'), rid, raw, meta, bc)

bad = 0


def check(name, a, b):
    global bad
    ok = (np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b)
    if not ok:
        bad += 1
    print('%-34s %s' % (name, 'OK' if ok else 'DIFF %r != %r' % (a, b)))


for fn, rid in [(single[0], None), (single[1], None), (single[2], None)] + [('multi.fast5', r[0]) for r in reads]:
    ids_ref = RF.get_read_ids(fn, TMP)
    ids_me = MF.get_read_ids(fn, TMP)
    check(fn + ' get_read_ids', sorted(ids_ref), sorted(ids_me))
    use = rid if rid is not None else ids_ref[0][1]
    with RF.Fast5Reader(os.path.join(TMP, fn), use) as ref:
        me = MF.Fast5Reader(os.path.join(TMP, fn), use)
        for attr in ('read_id', 'duration', 'start_time', 'channel_number', 'digitization', 'offset', 'range',
                     'sampling_rate', 'run_id', 'sample_id'):
            check('%s %s' % (fn, attr), getattr(ref, attr), getattr(me, attr))
        pa_ref = ref.get_raw_data()
        raw_me = me.get_raw_int16()
        pa_me = np.array((raw_me + me.offset) * (me.range / me.digitization), dtype=np.float32)
        check(fn + ' raw->pA', pa_ref, pa_me)
        bref = ref.get_basecall()
        bme = me.get_basecall()
        if bref is None or bme is None:
            check(fn + ' basecall none', bref is None, bme is None)
        else:
            for k in ('sequence', 'qstring', 'sequence_length', 'num_events', 'first_sample_template', 'block_stride'):
                check('%s basecall %s' % (fn, k), bref[k] if k in bref else None, bme[k])
            check(fn + ' mean_qscore', float(np.float32(bref['mean_qscore'])), float(np.float32(bme['mean_qscore'])))
        me.close()
print('differences:', bad)
sys.exit(1 if bad else 0)
