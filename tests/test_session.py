"""Session driver (poreplex_amd/session.py): files -> shards -> double-buffered fat batches
-> facade -> sinks -> count all-reduce.  CPU legs drive the host logic with the oracle test
double of the GPU context (tests/oracle_context.py); the -m gpu leg streams the same bundles
through the real stage/swap path."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from poreplex_amd import native as N
from poreplex_amd import sinks as SINK
from poreplex_amd.config import default_config
from poreplex_amd.worker_persistence import WorkerPersistenceStorage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def session_config(outdir, bundle='chimera.pxr.npz', **kw):
    with open(os.path.join(GOLDEN, bundle.replace('.pxr.npz', '.results.json'))) as fh:
        flags = dict(json.load(fh)['config_flags'])
    flags.update(kw)
    return default_config(inputdir='/nonexistent-inputdir', outputdir=str(outdir),
                          read_bundle=os.path.join(GOLDEN, bundle), fastq_output=True, **flags)


def golden_reads(bundle):
    """The read list the reference was given for this bundle (includes a file that vanished
    and a corrupt one for batch0: they never enter the GPU pass but must be reported)."""
    with open(os.path.join(GOLDEN, bundle.replace('.pxr.npz', '.results.json'))) as fh:
        return [tuple(r) for r in json.load(fh)['reads']]


def run_session(outdir, bundle, batch_reads, **kw):
    from poreplex_amd.session import GpuSession
    WorkerPersistenceStorage.reset()
    try:
        return GpuSession(session_config(outdir, bundle, **kw),
                          batch_reads=batch_reads).run(golden_reads(bundle))
    finally:
        WorkerPersistenceStorage.reset()


@pytest.fixture()
def oracle_backed(monkeypatch):
    from oracle_context import OracleBackedContext
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', OracleBackedContext)
    yield
    WorkerPersistenceStorage.reset()


def reference_outputs(bundle):
    """What one process_batch call over the whole bundle + the sinks write (input order)."""
    from poreplex_amd.signal_analyzer import SignalAnalyzer
    cfg = session_config('/tmp', bundle)
    reads = golden_reads(bundle)
    with SignalAnalyzer(cfg, 0) as an:
        batch = an.prepare(reads)
        an.loader.fit_scalers(batch.table)
        return an.finish(batch, input_order=True)


@pytest.mark.parametrize('bundle', ['batch0.pxr.npz', 'chimera.pxr.npz'])
def test_session_outputs_do_not_depend_on_batch_size(oracle_backed, tmp_path, bundle):
    whole = run_session(tmp_path / 'a', bundle, batch_reads=10000)
    small = run_session(tmp_path / 'b', bundle, batch_reads=5)
    assert whole['batches'] == 1 and small['batches'] >= 3
    a = (tmp_path / 'a' / 'sequencing_summary.txt').read_bytes()
    assert a == (tmp_path / 'b' / 'sequencing_summary.txt').read_bytes() and a.count(b'\n') > 5
    assert np.array_equal(whole['counts'], small['counts'])
    assert whole['labels'].tobytes() == small['labels'].tobytes()
    # the same rows the facade + sinks give for one process_batch over everything
    results = reference_outputs(bundle)
    cfg = session_config(tmp_path / 'c', bundle)
    os.makedirs(tmp_path / 'c')
    labels, barcodes, layout = SINK.setup_output_name_mapping(cfg)
    w = SINK.SequencingSummaryWriter(dict(cfg, fast5_output=False), str(tmp_path / 'c'), labels, barcodes)
    w.write_results(results)
    w.close()
    assert a == (tmp_path / 'c' / 'sequencing_summary.txt').read_bytes()
    # rank 0's tracker was fed by the reduced count table == feeding the final result dicts
    direct = SINK.FinalSummaryTracker(labels, barcodes)
    direct.feed_results(results)
    assert dict(whole['tracker'].counts) == dict(direct.counts)
    # FASTQ routing: every sequence once, in the file of its (label, barcode)
    import gzip
    n_seq = 0
    for (label, bc), name in layout.items():
        path = tmp_path / 'b' / 'fastq' / (name + '.fastq.gz')
        ids = [ln[1:] for ln in gzip.open(path, 'rt').read().splitlines()[0::4]]
        want = [r['read_id'] for r in results if r.get('sequence') is not None
                and r.get('label') == label and r.get('barcode') == bc]
        assert ids == want, (label, bc)
        n_seq += len(ids)
    assert n_seq == sum(1 for r in results if r.get('sequence') is not None)
    assert not [f for f in os.listdir(tmp_path / 'b') if '.part' in f]


def test_session_dumps_one_file_per_batch(oracle_backed, tmp_path, monkeypatch):
    """--dump-adapter-signals / --dump-basecalls through the session driver: a dump file per
    batch (named by the batch's first read), and over all of them the same datasets and
    catalogue rows the REAL reference dumped for these reads (tests/golden/dumps0.npz)."""
    from poreplex_amd import fast5_write
    written, files = {}, []
    create, init = fast5_write.H5Writer.create_dataset, fast5_write.H5Writer.__init__

    events = {}

    def recording(self, path, data, attrs=()):
        if path.startswith('basecalled_events/'):
            events[path.split('/')[-1]] = np.array(data)
        else:
            written.setdefault(path.split('/')[-2 if path.startswith('adapter/') else -1], {})[path] = np.array(data)
        return create(self, path, data, attrs)

    def opened(self, path):
        files.append(os.path.basename(path))
        return init(self, path)
    monkeypatch.setattr(fast5_write.H5Writer, 'create_dataset', recording)
    monkeypatch.setattr(fast5_write.H5Writer, '__init__', opened)
    out = run_session(tmp_path / 'a', 'batch0.pxr.npz', batch_reads=7, dump_adapter_signals=True, dump_basecalls=True)
    assert out['batches'] == 5 and len(files) == 10 and len(set(files)) == 5        # (same names in two directories)
    assert sorted(os.listdir(tmp_path / 'a' / 'adapter-dumps')) == sorted(set(files))
    assert sorted(os.listdir(tmp_path / 'a' / 'events')) == sorted(set(files))
    want = np.load(os.path.join(GOLDEN, 'dumps0.npz'))
    rows, sigs = [], {}
    for batch in sorted(written):                          # batch ids = first read of the batch: input order
        for path, data in written[batch].items():
            if path.startswith('catalog/'):
                rows.append(data)
            else:
                sigs[path.split('/')[-1]] = data
    assert np.array_equal(np.concatenate(rows), want['adapter_catalog'])
    assert sorted(sigs) == want['adapter_ids'].tolist()
    for k, rid in enumerate(want['adapter_ids'].tolist()):
        assert np.array_equal(sigs[rid], want['adapter_values'][want['adapter_offsets'][k]:want['adapter_offsets'][k + 1]])
    # --dump-basecalls: the same event tables, whichever batch a read fell into
    assert sorted(events) == want['events_ids'].tolist()
    for k, rid in enumerate(want['events_ids'].tolist()):
        assert events[rid].tobytes() == want['events_rows'][want['events_offsets'][k]:want['events_offsets'][k + 1]].tobytes()
    # and the run's own outputs do not change
    plain = run_session(tmp_path / 'b', 'batch0.pxr.npz', batch_reads=7)
    assert (tmp_path / 'a' / 'sequencing_summary.txt').read_bytes() == (tmp_path / 'b' / 'sequencing_summary.txt').read_bytes()
    assert not os.path.exists(tmp_path / 'b' / 'adapter-dumps') and out['labels'].tobytes() == plain['labels'].tobytes()


def test_session_edge_cases_empty_run_and_no_gpu_rows(oracle_backed, tmp_path):
    """No reads at all; and a run in which NO read reaches the GPU (all too short, one file
    gone): the driver must neither launch nor wait for anything, and still write its files."""
    from poreplex_amd.fast5_file import write_bundle
    from poreplex_amd.session import GpuSession
    from poreplex_amd.synth import synth_batch
    sb = synth_batch(9, seed=4, samples_per_read=4000, jitter=0.2)        # < 9000 samples: too short
    names = ['s%02d.fast5' % i for i in range(9)]
    ids = ['%08d-0000-4000-8000-000000000000' % i for i in range(9)]
    path = str(tmp_path / 'short.pxr.npz')
    write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], names, ids)
    cfg = default_config(inputdir='/nonexistent', outputdir=str(tmp_path / 'o1'), read_bundle=path)
    WorkerPersistenceStorage.reset()
    out = GpuSession(cfg, batch_reads=4).run([])
    assert out['reads'] == 0 and out['batches'] == 0
    assert (tmp_path / 'o1' / 'sequencing_summary.txt').read_text().count('\n') == 1      # header only
    WorkerPersistenceStorage.reset()
    cfg = default_config(inputdir='/nonexistent', outputdir=str(tmp_path / 'o2'), read_bundle=path)
    out = GpuSession(cfg, batch_reads=4).run(list(zip(names, ids)) + [('gone.fast5', 'x')])
    assert out['reads'] == 10 and out['batches'] == 3
    from poreplex_amd import distributed as D
    fail = D.LABEL_NAMES.index('fail')
    assert out['counts'][fail, 0, N.STATUS_CODE['scaler_signal_too_short']] == 9
    assert out['counts'][fail, 0, N.STATUS_CODE['disappeared']] == 1
    assert (tmp_path / 'o2' / 'sequencing_summary.txt').read_text().count('\n') == 1      # no labelled read
    WorkerPersistenceStorage.reset()


def test_count_table_uses_final_labels(oracle_backed, tmp_path):
    """The all-reduced table must carry the statuses the facade decides AFTER the GPU pass
    (unsplit_read / artifact here), not the numeric-stage verdict."""
    out = run_session(tmp_path, 'chimera.pxr.npz', batch_reads=4)
    from poreplex_amd import distributed as D
    counts = out['counts']
    assert counts[D.LABEL_NAMES.index('artifact'), :, N.STATUS_CODE['unsplit_read']].sum() == 6
    assert counts.sum() == len(out['labels']) == 11


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
    import torch.distributed as dist
    from poreplex_amd import native as N
    from oracle_context import OracleBackedContext
    N.NativeContext = OracleBackedContext          # test double: no GPU in this container
    import test_session as TS
    from poreplex_amd.session import GpuSession
    dist.init_process_group('gloo')
    out = GpuSession(TS.session_config({out!r}, {bundle!r}), dist=dist,
                     batch_reads=3).run(TS.golden_reads({bundle!r}))
    assert out['reads'] == {n_reads} and out['world'] == 2 and 0 < out['reads_this_rank'] < {n_reads}
    assert out['labels']['read_index'].tolist() == list(range({n_reads}))
    dist.destroy_process_group()
    open(os.path.join({out!r}, 'rank%d.ok' % out['rank']), 'w').write('ok')
""")


@pytest.mark.parametrize('bundle,n_reads', [('chimera.pxr.npz', 11)])
def test_two_rank_session_writes_the_single_rank_files(oracle_backed, tmp_path, bundle, n_reads):
    single = run_session(tmp_path / 'one', bundle, batch_reads=4)
    assert single['reads'] == n_reads
    two = tmp_path / 'two'
    two.mkdir()
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, out=str(two), bundle=bundle, n_reads=n_reads))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (two / 'rank0.ok').exists() and (two / 'rank1.ok').exists()
    assert (two / 'sequencing_summary.txt').read_bytes() == \
        (tmp_path / 'one' / 'sequencing_summary.txt').read_bytes()
    import gzip
    for dirpath, _, files in os.walk(tmp_path / 'one' / 'fastq'):
        for f in files:
            rel = os.path.relpath(os.path.join(dirpath, f), tmp_path / 'one')
            assert gzip.open(two / rel).read() == gzip.open(tmp_path / 'one' / rel).read(), rel
    assert not [f for f in os.listdir(two) if '.part' in f]


@pytest.mark.gpu
@pytest.mark.parametrize('bundle', ['batch0.pxr.npz', 'chimera.pxr.npz'])
def test_session_on_gpu_streams_batches_through_stage_swap(tmp_path, bundle):
    """Real context: >= 3 double-buffered batches through pinned staging arenas; the files
    equal those of the single upload/run path (process over the whole bundle)."""
    small = run_session(tmp_path / 'b', bundle, batch_reads=4)
    assert small['batches'] >= 3
    WorkerPersistenceStorage.reset()
    results = reference_outputs(bundle)
    WorkerPersistenceStorage.reset()
    cfg = session_config(tmp_path / 'c', bundle)
    os.makedirs(tmp_path / 'c')
    labels, barcodes, _ = SINK.setup_output_name_mapping(cfg)
    w = SINK.SequencingSummaryWriter(dict(cfg, fast5_output=False), str(tmp_path / 'c'), labels, barcodes)
    w.write_results(results)
    w.close()
    assert (tmp_path / 'b' / 'sequencing_summary.txt').read_bytes() == \
        (tmp_path / 'c' / 'sequencing_summary.txt').read_bytes()


def test_session_degenerate_inputs(oracle_backed, tmp_path):
    """No reads at all, one read per batch, and a run whose reads all fail before the GPU pass
    (files that are not in the bundle): the driver still writes complete, consistent outputs."""
    from poreplex_amd.session import GpuSession
    cfg = session_config(tmp_path / 'none', 'chimera.pxr.npz')
    out = GpuSession(cfg, batch_reads=4).run([])
    assert out['reads'] == 0 and out['batches'] == 0 and len(out['labels']) == 0
    assert (tmp_path / 'none' / 'sequencing_summary.txt').read_text().count('\n') == 1      # header only
    WorkerPersistenceStorage.reset()
    reads = golden_reads('chimera.pxr.npz')
    one = run_session(tmp_path / 'one', 'chimera.pxr.npz', batch_reads=1)
    many = run_session(tmp_path / 'many', 'chimera.pxr.npz', batch_reads=64)
    assert one['batches'] == len(reads) and many['batches'] == 1
    assert (tmp_path / 'one' / 'sequencing_summary.txt').read_bytes() == \
        (tmp_path / 'many' / 'sequencing_summary.txt').read_bytes()
    assert one['labels'].tobytes() == many['labels'].tobytes()
    WorkerPersistenceStorage.reset()
    ghosts = [('no/such/file_%d.fast5' % i, 'ghost-%d' % i) for i in range(7)]
    out = GpuSession(session_config(tmp_path / 'ghosts', 'chimera.pxr.npz'), batch_reads=3).run(ghosts)
    assert out['reads'] == 7 and len(out['labels']) == 7
    assert int(out['counts'].sum()) == 7


# ---- failure path (pipeline.py:207-213,250-265: a failed batch / an early stop ends the run) ----
def _threads_alive():
    import threading
    return [t for t in threading.enumerate() if t is not threading.main_thread() and t.is_alive()]


def test_loader_exception_mid_run_unwinds_the_session(oracle_backed, tmp_path, monkeypatch):
    """An exception in the loader thread (batch 2 of 4) surfaces on the caller, after the
    thread is joined, the arenas released and the part files closed -- within seconds."""
    import time
    from poreplex_amd.session import GpuSession
    from poreplex_amd.signal_analyzer import SignalAnalyzer
    calls = {'n': 0}
    real = SignalAnalyzer.prepare

    def flaky(self, reads, table=None, reserve=None):
        calls['n'] += 1
        if calls['n'] == 3:
            raise OSError('input volume went away')
        return real(self, reads, table, reserve)
    monkeypatch.setattr(SignalAnalyzer, 'prepare', flaky)
    before = len(_threads_alive())
    sess = GpuSession(session_config(tmp_path, 'chimera.pxr.npz'), batch_reads=3)
    t0 = time.perf_counter()
    with pytest.raises(OSError, match='input volume'):
        sess.run(golden_reads('chimera.pxr.npz'))
    assert time.perf_counter() - t0 < 30
    assert len(_threads_alive()) == before                      # the loader thread is gone
    part = tmp_path / 'sequencing_summary.txt.part0000'
    assert part.exists() and part.read_text().count('\n') >= 1  # closed and flushed, header first
    # ... and the context is usable again: a second run on the same worker completes
    monkeypatch.setattr(SignalAnalyzer, 'prepare', real)
    out = GpuSession(session_config(tmp_path / 'again', 'chimera.pxr.npz'), batch_reads=3).run(
        golden_reads('chimera.pxr.npz'))
    assert out['reads'] == 11


def test_gpu_error_mid_run_unwinds_the_session(oracle_backed, tmp_path, monkeypatch):
    from oracle_context import OracleBackedContext
    from poreplex_amd.session import GpuSession
    runs = {'n': 0}
    real = OracleBackedContext.run

    def failing_run(self, mask=N.STAGE_ALL_DEMUX):
        runs['n'] += 1
        if runs['n'] == 2:
            raise N.PxgError('pxg_batch_run failed (-3): hipErrorLaunchFailure')
        return real(self, mask)
    monkeypatch.setattr(OracleBackedContext, 'run', failing_run)
    before = len(_threads_alive())
    with pytest.raises(N.PxgError, match='hipErrorLaunchFailure'):
        GpuSession(session_config(tmp_path, 'chimera.pxr.npz'), batch_reads=3).run(
            golden_reads('chimera.pxr.npz'))
    assert len(_threads_alive()) == before


def test_early_stop_is_a_clean_stop(oracle_backed, tmp_path, monkeypatch):
    """pipeline.py:250-260: reads keep arriving without basecalls -> one message, run over."""
    from poreplex_amd.session import GpuSession
    from poreplex_amd.fast5_file import ReadBundle
    real = ReadBundle.basecall_of
    monkeypatch.setattr(ReadBundle, 'basecall_of', lambda self, i: None)
    cfg = session_config(tmp_path, 'chimera.pxr.npz', nobasecall_stop_trigger=3)
    sess = GpuSession(cfg, batch_reads=4)
    sess.loader.bundle.d['bc_present'][:] = False
    before = len(_threads_alive())
    with pytest.raises(RuntimeError, match='Early stopping: '):
        sess.run(golden_reads('chimera.pxr.npz'))
    assert len(_threads_alive()) <= before


ABORT_WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
    import torch.distributed as dist
    from poreplex_amd import native as N
    from oracle_context import OracleBackedContext
    N.NativeContext = OracleBackedContext          # test double: no GPU in this container
    import test_session as TS
    from poreplex_amd.session import GpuSession, SessionAborted
    from poreplex_amd.signal_analyzer import SignalAnalyzer
    dist.init_process_group('gloo')
    rank = dist.get_rank()
    if rank == 1:                                   # rank 1 loses its input in its second batch
        real, calls = SignalAnalyzer.prepare, [0]
        def flaky(self, reads, table=None, reserve=None):
            calls[0] += 1
            if calls[0] == 2:
                raise OSError('rank 1 lost its input')
            return real(self, reads, table, reserve)
        SignalAnalyzer.prepare = flaky
    t0 = time.perf_counter()
    try:
        GpuSession(TS.session_config({out!r}, {bundle!r}), dist=dist, batch_reads=2).run(TS.golden_reads({bundle!r}))
        verdict = 'finished'
    except SessionAborted as exc:
        verdict = 'aborted: ' + str(exc)
    except OSError as exc:
        verdict = 'failed: ' + str(exc)
    open(os.path.join({out!r}, 'rank%d.txt' % rank), 'w').write('%s|%.1f' % (verdict, time.perf_counter() - t0))
    dist.destroy_process_group()
""")


def test_one_rank_failing_stops_every_rank(oracle_backed, tmp_path):
    """2 ranks over gloo, rank 1 fails in its second batch: rank 0 must not hang in the final
    collectives -- both leave the loop in the same round, within seconds."""
    script = tmp_path / 'worker.py'
    script.write_text(ABORT_WORKER.format(root=ROOT, out=str(tmp_path), bundle='chimera.pxr.npz'))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r0, r1 = [(tmp_path / ('rank%d.txt' % r)).read_text().split('|') for r in (0, 1)]
    assert r1[0] == 'failed: rank 1 lost its input'
    assert r0[0].startswith('aborted: rank 1 stopped the run')
    assert float(r0[1]) < 60 and float(r1[1]) < 60
