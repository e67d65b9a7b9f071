"""Runs under an interpreter that has h5py (here: /opt/conda/bin/python3.9): the session
driver fed from real FAST5 FILES (single- and multi-read) must write exactly
what it writes for the same reads held in a .pxr.npz bundle.  The GPU context is the oracle
test double (no GPU in that interpreter either).  Prints 'OK <n reads>' on success."""
import os
import sys
import tempfile

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from poreplex_amd import native as N  # noqa: E402
from oracle_context import OracleBackedContext  # noqa: E402
N.NativeContext = OracleBackedContext
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.fast5_file import write_bundle  # noqa: E402
from poreplex_amd.session import GpuSession, enumerate_reads  # noqa: E402
from poreplex_amd.synth import synth_basecalls, synth_batch  # noqa: E402
from poreplex_amd.worker_persistence import WorkerPersistenceStorage  # noqa: E402


def fill(node_raw, node_ch, node_tr, analyses, rid, raw, cal, meta, bc, events_table=False):
    node_raw.attrs['duration'] = np.uint32(len(raw))
    node_raw.attrs['start_time'] = np.uint64(meta['start_time'])
    node_raw.attrs['read_id'] = rid.encode()
    node_raw.create_dataset('Signal', data=raw)
    node_ch.attrs['channel_number'] = str(meta['channel']).encode()
    node_ch.attrs['digitisation'] = float(cal['digitisation'])
    node_ch.attrs['offset'] = float(cal['offset'])
    node_ch.attrs['range'] = float(cal['range'])
    node_ch.attrs['sampling_rate'] = float(cal['sampling_rate'])
    node_tr.attrs['run_id'] = b'run0'
    node_tr.attrs['sample_id'] = b'sampleX'
    if bc is None:
        return
    g = analyses.create_group('Basecall_1D_000')
    t = g.create_group('BaseCalled_template')
    t.create_dataset('Fastq', data=np.string_('@{}\n{}\n+\n{}\n'.format(rid, bc['sequence'], bc['qstring'])))
    if events_table:        # guppy < 2.3.7 layout: an Events table instead of Move
        ev = np.zeros(len(bc['move']), dtype=[('model_state', 'S5'), ('move', 'u1'), ('p_model_state', '<f4')])
        ev['move'] = bc['move']
        ev['p_model_state'] = 0.9
        t.create_dataset('Events', data=ev)
    else:
        t.create_dataset('Move', data=np.asarray(bc['move'], dtype=np.uint8))
    s = g.create_group('Summary/basecall_1d_template')
    s.attrs['sequence_length'] = np.int32(bc['sequence_length'])
    s.attrs['mean_qscore'] = np.float32(bc['mean_qscore'])
    s.attrs['block_stride'] = np.int32(bc['block_stride'])
    sg = analyses.create_group('Segmentation_000/Summary/segmentation')
    sg.attrs['num_events_template'] = np.int32(bc['num_events'])
    sg.attrs['first_sample_template'] = np.int32(bc['first_sample_template'])


def main():
    n = 14
    sb = synth_batch(n, seed=77, samples_per_read=16000, jitter=0.3, short_fraction=0.15)
    bcs = synth_basecalls(sb, seed=3)
    bcs[2] = None                                   # a read that was never basecalled
    for b in bcs:
        if b is not None:
            b['mean_qscore'] = float(np.float32(b['mean_qscore']))
    top = tempfile.mkdtemp(prefix='pxg_f5sess_')
    os.makedirs(os.path.join(top, 'in', 'sub'))
    rids = ['%08d-1111-4222-8333-%012d' % (i, i) for i in range(n)]
    raws = [sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in range(n)]
    names = []
    # reads 0-7: single-read files (two of them in a sub-directory); 8-13: one multi-read file
    for i in range(8):
        rel = os.path.join('sub' if i >= 6 else '', 'single_%02d.fast5' % i)
        with h5py.File(os.path.join(top, 'in', rel), 'w') as h5:
            fill(h5.create_group('Raw/Reads/Read_%d' % i), h5.create_group('UniqueGlobalKey/channel_id'),
                 h5.create_group('UniqueGlobalKey/tracking_id'), h5.create_group('Analyses'), rids[i], raws[i],
                 sb['calib'][i], {'start_time': 1000 * i + 7, 'channel': 100 + i}, bcs[i], events_table=(i == 4))
        names.append(rel)
    with h5py.File(os.path.join(top, 'in', 'z_multi.fast5'), 'w') as h5:
        for i in range(8, n):
            base = h5.create_group('read_' + rids[i])
            fill(base.create_group('Raw'), base.create_group('channel_id'), base.create_group('tracking_id'),
                 base.create_group('Analyses'), rids[i], raws[i], sb['calib'][i],
                 {'start_time': 1000 * i + 7, 'channel': 100 + i}, bcs[i])
            names.append('z_multi.fast5')
    flags = dict(barcoding=True, measure_polya=True, filter_unsplit_reads=True, fastq_output=False)

    def run(outdir, **src):
        WorkerPersistenceStorage.reset()
        cfg = default_config(outputdir=outdir, **flags, **src)
        out = GpuSession(cfg, batch_reads=5).run()
        WorkerPersistenceStorage.reset()
        return out, open(os.path.join(outdir, 'sequencing_summary.txt')).read()

    files_out, files_txt = run(os.path.join(top, 'out_files'), inputdir=os.path.join(top, 'in'))
    # the same reads as a bundle, in the order the directory walk finds them
    found, _ = enumerate_reads(default_config(inputdir=os.path.join(top, 'in')))
    assert sorted(found) == sorted(zip(names, rids)) and len(found) == n, found
    order = [rids.index(rid) for _, rid in found]
    parts = [raws[i] for i in order]
    arena, off = N.pack_reads(parts)
    bpath = os.path.join(top, 'same.pxr.npz')
    write_bundle(bpath, arena, off, sb['calib'][order], [found[k][0] for k in range(n)],
                 [found[k][1] for k in range(n)], basecalls=[bcs[i] if i != 4 else dict(bcs[i], table='guppy_events', p_model_state=[0.9] * len(bcs[i]['move'])) for i in order],
                 start_time=np.array([1000 * i + 7 for i in order]), channel_number=np.array([str(100 + i) for i in order]),
                 run_id=np.array(['run0'] * n), sample_id=np.array(['sampleX'] * n))
    bundle_out, bundle_txt = run(os.path.join(top, 'out_bundle'), inputdir='/nonexistent', read_bundle=bpath)
    assert files_txt == bundle_txt, (files_txt, bundle_txt)
    assert files_out['labels'].tobytes() == bundle_out['labels'].tobytes()
    assert np.array_equal(files_out['counts'], bundle_out['counts'])
    assert files_txt.count('\n') >= 8, files_txt
    print('OK', n, 'reads;', files_txt.count('\n') - 1, 'summary rows')


if __name__ == '__main__':
    main()
