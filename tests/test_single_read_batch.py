"""A request of single-read FAST5 files -- the reference's classic input, one file per read -- opened by ONE native call
(fast5_file.OpenedFiles / pxg_h5_open_many, SignalLoader.prepare_single_read_files) against the per-file route it
replaces: same table, same result dicts, for readable files and for the ones the per-read path must report (vanished,
corrupt, holding another read, a multi-read file among them).  CPU only (the context is a double with made-up records)."""
import os
import sys

import numpy as np
import pytest

from poreplex_amd import native as N
from poreplex_amd import signal_analyzer as SA
from poreplex_amd.config import default_config
from poreplex_amd.fast5_file import OpenedFiles
from poreplex_amd.fast5_write import Fast5Writer, write_single_read
from poreplex_amd.signal_loader import SignalLoader
from poreplex_amd.synth import synth_basecalls, synth_batch
from poreplex_amd.worker_persistence import WorkerPersistenceStorage
from test_plain_run import CraftedRecords, same


def modes():
    try:
        from poreplex_amd.fast5_write import vbz_encode
        vbz_encode(np.zeros(4, np.int16))
        return (None, 'gzip', 'vbz')
    except OSError:
        return (None, 'gzip')


@pytest.fixture()
def directory(tmp_path):
    n = 40
    sb = synth_batch(n, seed=77, samples_per_read=12000)
    sb['calib']['offset'] = np.arange(n)                   # (CraftedRecords recognises a read by it)
    bcs = synth_basecalls(sb, seed=77)
    o = sb['offsets']
    names, ids = [], []
    for j in range(n):
        name = 'sub{}/read_{:03d}.fast5'.format(j % 3, j)
        os.makedirs(os.path.join(tmp_path, 'sub{}'.format(j % 3)), exist_ok=True)
        write_single_read(str(tmp_path / name), 'id-{:03d}'.format(j), sb['arena'][o[j]:o[j + 1]], sb['calib'][j],
                          start_time=100 + j, channel_number=str(1 + j), run_id='run', sample_id='s',
                          basecall=None if j % 7 == 3 else bcs[j], compression=modes()[j % len(modes())],
                          chunk=4000 if j % 2 else None)
        names.append(name)
        ids.append('id-{:03d}'.format(j))
    rec = np.zeros(n, dtype=N.RESULT_DTYPE)
    rec['seg_first'], rec['seg_last'] = -1, -1
    rec['bc_pushed'] = rec['bc_called'] = 1
    rec['bc_label'] = np.arange(n) % 4
    CraftedRecords.table = rec
    WorkerPersistenceStorage.reset()
    yield tmp_path, names, ids, rec
    WorkerPersistenceStorage.reset()
    CraftedRecords.table = None


def both_routes(monkeypatch, cfg, reads):
    taken = []
    real = SignalLoader.prepare_single_read_files

    def spy(self, *a, **kw):
        out = real(self, *a, **kw)
        taken.append(out is not None)
        return out
    monkeypatch.setattr(SignalLoader, 'prepare_single_read_files', spy)
    monkeypatch.setattr(SignalLoader, 'SINGLE_READ_BATCH_MIN', 8)
    batch = SA.process_batch(1, list(reads), cfg)
    used = list(taken)
    monkeypatch.setattr(SignalLoader, 'SINGLE_READ_BATCH_MIN', 10 ** 9)
    per_file = SA.process_batch(1, list(reads), cfg)
    return batch, per_file, used


def test_a_directory_of_single_read_files(directory, monkeypatch):
    tmp_path, names, ids, rec = directory
    monkeypatch.setattr(N, 'NativeContext', CraftedRecords)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), barcoding=True)
    assert isinstance(SA.process_batch(0, [(names[0], ids[0])], cfg), list)
    adapter = sys.modules[WorkerPersistenceStorage.STORAGE_NAME].storage['ctx'].state_names.index('adapter')
    rec['seg_first'][:, adapter], rec['seg_last'][:, adapter] = 30, 80
    reads = list(zip(names, ids))
    batch, per_file, used = both_routes(monkeypatch, cfg, reads)
    assert used == [True] and isinstance(batch, list)
    same(batch, per_file)
    assert [r['read_id'] for r in batch] == ids and {r['status'] for r in batch} == {'okay', 'not_basecalled'}
    assert [r['barcode'] for r in batch] == (np.arange(40) % 4).tolist()          # every read met ITS record
    assert [r['start_time'] for r in batch[:2]] == [round(100 / 3012.0, 3), round(101 / 3012.0, 3)]


def test_files_the_batch_open_leaves_to_the_per_read_path(directory, monkeypatch):
    tmp_path, names, ids, rec = directory
    monkeypatch.setattr(N, 'NativeContext', CraftedRecords)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), barcoding=True)
    assert isinstance(SA.process_batch(0, [(names[0], ids[0])], cfg), list)
    adapter = sys.modules[WorkerPersistenceStorage.STORAGE_NAME].storage['ctx'].state_names.index('adapter')
    rec['seg_first'][:, adapter], rec['seg_last'][:, adapter] = 30, 80
    reads = list(zip(names, ids))
    os.remove(str(tmp_path / names[5]))                                            # vanished
    with open(str(tmp_path / names[9]), 'wb') as fh:
        fh.write(b'this is not an HDF5 file, but it is long enough to be looked at ' * 4)
    with open(str(tmp_path / names[12]), 'r+b') as fh:                             # truncated inside its structures
        fh.truncate(700)
    reads[20] = (names[20], 'somebody-else')                                       # the file holds another read
    batch, per_file, used = both_routes(monkeypatch, cfg, reads)
    assert used == [True]
    assert len(batch) == len(per_file) == 40
    by_file = {r['filename']: r for r in batch}
    assert by_file[names[5]]['status'] == 'disappeared'
    assert by_file[names[9]]['status'] == by_file[names[12]]['status'] == by_file[names[20]]['status'] == 'unknown_error'
    for a, b in zip(batch, per_file):
        if a['status'] == 'unknown_error':                   # (same message up to the traceback's line numbers)
            assert b['status'] == 'unknown_error' and a['filename'] == b['filename'] and list(a) == list(b)
            assert a['error_message'].splitlines()[-1] == b['error_message'].splitlines()[-1]
        else:
            same(a, b)
    assert sum(r['status'] in ('okay', 'not_basecalled') for r in batch) == 36


def test_requests_the_batch_open_declines(directory, monkeypatch):
    tmp_path, names, ids, rec = directory
    monkeypatch.setattr(N, 'NativeContext', CraftedRecords)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), barcoding=True)
    reads = list(zip(names, ids))
    # a multi-read file among them; a file named twice; a short list
    with Fast5Writer(str(tmp_path / 'multi.fast5')) as w:
        sb = synth_batch(2, seed=5, samples_per_read=12000)
        for j in range(2):
            w.add_read('m-{}'.format(j), sb['arena'][sb['offsets'][j]:sb['offsets'][j + 1]], sb['calib'][j])
    for request in (reads[:10] + [('multi.fast5', 'm-1')], reads[:10] + reads[3:4], reads[:4]):
        batch, per_file, used = both_routes(monkeypatch, cfg, request)
        assert used in ([False], []) and isinstance(batch, list) and len(batch) == len(request)
        same(batch, per_file)


def test_opened_files_reports_each_file(directory):
    tmp_path, names, ids, rec = directory
    paths = [str(tmp_path / n) for n in names[:6]] + [str(tmp_path / 'nowhere.fast5')]
    got = OpenedFiles(paths, threads=3)
    assert got.rc[:6].tolist() == [0] * 6 and got.rc[6] != 0 and got.handles[6] == 0
    assert b'nowhere.fast5' in got.error[6] and got.n_reads[:6].tolist() == [1] * 6 and not got.multi.any()
    assert [r.decode() for r in got.info['read_id'][:6]] == ids[:6] and (got.info['status'][:6] == 0).all()
    f = got.file(2)
    assert f.n == 1 and not f.multi and f.info['n_samples'][0] == got.info['n_samples'][2] and f.read_ids == [ids[2]]
    samples = f.signal(0)
    assert samples.dtype == np.int16 and len(samples) == got.info['n_samples'][2]
    del f
    got.close()
    got.close()                                              # (idempotent)


def test_listing_a_directory(directory, tmp_path):
    """get_read_ids_many == get_read_ids file by file, for single-read files, a multi-read file among them and files it
    must leave to get_read_ids (which raises for what cannot be opened)."""
    from poreplex_amd.fast5_file import get_read_ids, get_read_ids_many
    top, names, ids, rec = directory
    with Fast5Writer(str(top / 'multi.fast5')) as w:
        sb = synth_batch(3, seed=5, samples_per_read=12000)
        for j in range(3):
            w.add_read('m-{}'.format(j), sb['arena'][sb['offsets'][j]:sb['offsets'][j + 1]], sb['calib'][j])
    files = names[:7] + ['multi.fast5'] + names[7:]
    want = [pair for f in files for pair in get_read_ids(f, str(top))]
    assert get_read_ids_many(files, str(top)) == want and len(want) == 43
    assert get_read_ids_many(files, str(top), chunk=5) == want
    assert get_read_ids_many([], str(top)) == []
    with open(str(top / names[4]), 'wb') as fh:
        fh.write(b'not an HDF5 file at all, whatever its name says ' * 8)
    with pytest.raises(OSError):
        get_read_ids(names[4], str(top))
    with pytest.raises(OSError):
        get_read_ids_many(files, str(top))
    from poreplex_amd.session import enumerate_reads
    os.remove(str(top / names[4]))
    listed, _ = enumerate_reads({'inputdir': str(top)})
    assert sorted(listed) == sorted(p for p in want if p[0] != names[4])


def test_a_batch_open_under_a_small_descriptor_limit(directory):
    """200 opens of 40 files under RLIMIT_NOFILE = 80: beyond a share of the limit a file keeps its mapping and gives its
    descriptor back, every one of them opens and decodes to the same samples."""
    import subprocess
    top, names, ids, rec = directory
    script = '''
import resource, sys
import numpy as np
sys.path.insert(0, {root!r})
from poreplex_amd import fast5_file as F5
names = {names!r}
want = F5.OpenedFiles([{top!r} + '/' + n for n in names])
ns = want.info['n_samples'].astype(np.int64)
ref = F5.load_signals(want.handles, np.zeros(len(names), np.int64), ns)
hard = resource.getrlimit(resource.RLIMIT_NOFILE)[1]
resource.setrlimit(resource.RLIMIT_NOFILE, (80, hard))
many = F5.OpenedFiles([{top!r} + '/' + n for n in names] * 5, threads=4)
assert (many.rc == 0).all(), many.rc
got = F5.load_signals(many.handles, np.zeros(len(names) * 5, np.int64), np.tile(ns, 5), threads=4)
for k, a in enumerate(got):
    assert np.array_equal(a, ref[k % len(names)]), k
print('ok', len(got))
'''.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), names=names, top=str(top))
    out = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == 'ok 200', (out.stdout, out.stderr[-2000:])
