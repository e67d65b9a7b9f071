"""The reference's operator surface (process_batch / SignalAnalyzer /
SignalAnalysis / NanoporeRead.report) against what the REAL reference's
process_batch returned for the same reads (tests/golden/batch0.results.json).

CPU leg: the host logic (ordering, status/label rules, dict schema) is driven
with a test double of the GPU context that answers from the oracle -- test
infrastructure only, injected here, never reachable from the product.
GPU leg (-m gpu): the real context.
"""
import copy
import os

import numpy as np
import pytest

from poreplex_amd import native as N
from poreplex_amd.config import default_config
from poreplex_amd.worker_persistence import WorkerPersistenceStorage

from conftest import G  # noqa: E402  (golden file of the arithmetic the running test is in)

BUNDLE = G('batch0.pxr.npz')        # inputs: the same file for both arithmetics


def facade_config(ref_results, **kw):
    flags = dict(ref_results['config_flags'], outputdir='/tmp')
    flags.update(kw)
    return default_config(inputdir='/nonexistent-inputdir', read_bundle=BUNDLE, **flags)


def canon(o):
    """JSON-ify like tools/make_golden.py did for the reference's dicts."""
    if isinstance(o, dict):
        return {str(k): canon(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [canon(v) for v in o]
    if isinstance(o, np.floating):
        return float(o)
    if isinstance(o, np.integer):
        return int(o)
    return o


def compare_results(got, want, check_polya):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        g, w = canon(g), copy.deepcopy(w)
        if g['status'] == 'unknown_error':
            # same status, same keys; the traceback text names other files
            assert w['status'] == 'unknown_error' and set(g) == set(w)
            assert g['filename'] == w['filename'] and g['read_id'] == w['read_id']
            continue
        if not check_polya:
            g.pop('polya', None)
            w.pop('polya', None)
        assert list(g.keys()) == list(w.keys()), (g.get('read_id'), list(g), list(w))
        for k in w:
            assert g[k] == w[k], (g.get('read_id'), k, g[k], w[k])


from oracle_context import OneCallOracleContext, OracleBackedContext  # noqa: E402  (tests/oracle_context.py)


@pytest.fixture()
def oracle_backed(monkeypatch):
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', OracleBackedContext)
    yield
    WorkerPersistenceStorage.reset()


@pytest.fixture()
def one_call_double(monkeypatch):
    """The double that answers a worker batch in ONE call, as the library does: the host code then takes the branches
    it takes on the GPU (fit_scalers' one-call form; the plain-run path with the chimera scan on)."""
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', OneCallOracleContext)
    before = OneCallOracleContext.calls
    yield
    assert OneCallOracleContext.calls > before
    WorkerPersistenceStorage.reset()


def test_process_batch_host_logic_vs_reference(oracle_backed, ref_results):
    from poreplex_amd.signal_analyzer import process_batch
    reads = [tuple(r) for r in ref_results['reads']]
    got = process_batch(ref_results['batchid'], reads, facade_config(ref_results))
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref_results['results'], check_polya=True)
    # early failures first (encounter order), then loaded reads in input order
    assert [r['status'] for r in got[:3]] == ['unknown_error', 'disappeared',
                                             'scaler_signal_too_short']
    assert 'read_id' not in got[1]


def golden_batch_as_fast5(top, ref_results=None, bundle=None):
    """The golden batch's reads as the single-read FAST5 files the reference was given (written
    back from the committed bundle with the build's own writer, signal compression cycling),
    the corrupt file, and no file at all for the one that had vanished."""
    from poreplex_amd.fast5_file import ReadBundle
    from poreplex_amd.fast5_write import write_single_read
    b = ReadBundle(bundle or BUNDLE)
    d = b.d
    try:
        from poreplex_amd.fast5_write import vbz_encode
        vbz_encode(np.zeros(4, np.int16))
        modes = (None, 'gzip', 'vbz')
    except OSError:
        modes = (None, 'gzip')
    for i in range(len(d['read_id'])):
        bc = b.basecall_of(i)
        if bc is not None:
            bc['move'] = np.asarray(bc['move'], dtype=np.uint8)
        write_single_read(os.path.join(top, str(d['filename'][i])), str(d['read_id'][i]), b.samples(i), d['calib'][i],
                          start_time=int(d['start_time'][i]), channel_number=str(d['channel_number'][i]),
                          run_id=str(d['run_id'][i]), sample_id=str(d['sample_id'][i]), basecall=bc,
                          compression=modes[i % len(modes)], chunk=4000 if i % 2 else None, read_number=100 + i)
    with open(os.path.join(top, 'broken.fast5'), 'wb') as fh:
        fh.write(b'this is not an HDF5 file')


def test_process_batch_from_fast5_files_vs_reference(oracle_backed, ref_results, tmp_path):
    """north_star: "outputs match the reference CPU path on the same FAST5 inputs".  No bundle
    here: the reads are FAST5 files in inputdir, opened by the native reader a batch at a time,
    and the result list equals what the REAL reference returned for its FAST5 files."""
    from poreplex_amd.signal_analyzer import process_batch
    golden_batch_as_fast5(str(tmp_path), ref_results)
    flags = dict(ref_results['config_flags'])
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path / 'out'), **flags)
    assert not cfg.get('read_bundle')
    got = process_batch(ref_results['batchid'], [tuple(r) for r in ref_results['reads']], cfg)
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref_results['results'], check_polya=True)
    assert [r['status'] for r in got[:3]] == ['unknown_error', 'disappeared', 'scaler_signal_too_short']


def test_operator_surface(oracle_backed, ref_results):
    from poreplex_amd import signal_analyzer as SA
    assert SA.__all__ == ['SignalAnalyzer', 'SignalAnalysis', 'process_batch']
    cfg = facade_config(ref_results, measure_polya=False)
    with SA.SignalAnalyzer(cfg, 3) as an:
        for name in ('segmodel', 'unsplitmodel', 'kmersize', 'loader', 'demuxer', 'ctx'):
            assert hasattr(an, name)
        assert an.formatted_batchid == '00000003' and an.kmersize == 5
        res = an.process([tuple(r) for r in ref_results['reads'][:6]])
        assert all({'filename', 'status'} <= set(r) for r in res)
    for meth in ('process', 'detect_segments', 'push_barcode_signal', 'trim_adapter',
                 'load_events', 'is_stopped', 'set_error', 'clear_cache'):
        assert hasattr(SA.SignalAnalysis, meth)


def test_missing_gpu_is_fatal_tuple_not_fallback(ref_results, monkeypatch):
    """No library / no device -> the (-1, msg, tb) tuple (signal_analyzer.py:50-58);
    never a silent CPU path."""
    from poreplex_amd.signal_analyzer import process_batch
    WorkerPersistenceStorage.reset()

    def boom(config, device_id=0):
        raise N.PxgError('pxg_create failed (-2): no HIP device visible')
    monkeypatch.setattr(N, 'NativeContext', boom)
    out = process_batch(1, [tuple(r) for r in ref_results['reads'][:2]],
                        facade_config(ref_results))
    assert isinstance(out, tuple) and out[0] == -1 and 'PxgError' in out[1]
    WorkerPersistenceStorage.reset()


PY39 = '/opt/conda/bin/python3.9'
READ_DUMP = """
import glob, sys, h5py, numpy as np
(part,) = glob.glob(sys.argv[1] + '/adapter-dumps/part-*.h5')
with h5py.File(part, 'r') as h5:
    batches = list(h5['adapter'])
    assert batches == list(h5['catalog/adapter']) and len(batches) == 1
    ids = sorted(h5['adapter/' + batches[0]])
    sigs = [h5['adapter/' + batches[0] + '/' + k][:] for k in ids]
    assert all(x.dtype == np.float32 for x in sigs)
    cat = h5['catalog/adapter/' + batches[0]][:]
    cat = cat.astype([(name, cat.dtype[name].str) for name in cat.dtype.names])      # (h5py hangs metadata on the fields)
    np.savez(sys.argv[2], batch=np.array(batches[0]), catalog=cat, ids=np.array(ids),
             offsets=np.concatenate([[0], np.cumsum([len(x) for x in sigs])]).astype(np.int64), values=np.concatenate(sigs))
"""


def check_adapter_dump(outdir, tmp_path, written=None):
    """adapter-dumps/part-*.h5 of one batch-7 call against what the REAL reference dumped for
    the same reads (tests/golden/dumps0.npz, tools/make_golden.py --only dumps0): the recorded
    H5Writer calls, and -- where the image's python3.9 has h5py -- the file as the real HDF5
    library reads it back."""
    import subprocess
    want = np.load(G('dumps0.npz'))
    cat = want['adapter_catalog']
    if written is not None:
        assert written['catalog/adapter/00000007'].dtype == cat.dtype
        assert np.array_equal(written['catalog/adapter/00000007'], cat)
        for k, rid in enumerate(want['adapter_ids'].tolist()):
            sig = written['adapter/00000007/' + rid]
            assert sig.dtype == np.float32
            assert np.array_equal(sig, want['adapter_values'][want['adapter_offsets'][k]:want['adapter_offsets'][k + 1]]), rid
        assert len(written) == len(want['adapter_ids']) + 1
    if not os.path.exists(PY39) or subprocess.run([PY39, '-c', 'import h5py'], capture_output=True).returncode:
        return False
    back = str(tmp_path / 'readback.npz')
    out = subprocess.run([PY39, '-c', READ_DUMP, outdir, back], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = np.load(back)
    assert str(got['batch']) == '00000007' and got['catalog'].dtype == cat.dtype and np.array_equal(got['catalog'], cat)
    assert got['ids'].tolist() == want['adapter_ids'].tolist()
    assert np.array_equal(got['offsets'], want['adapter_offsets']) and np.array_equal(got['values'], want['adapter_values'])
    return True


READ_EVENTS = """
import glob, json, sys, h5py, numpy as np
(part,) = glob.glob(sys.argv[1] + '/events/part-*.h5')
with h5py.File(part, 'r') as h5:
    (batch,) = list(h5['basecalled_events'])
    ids = sorted(h5['basecalled_events/' + batch])
    tabs = [h5['basecalled_events/' + batch + '/' + k][:] for k in ids]
    rows = np.concatenate(tabs)
    rows = rows.astype([(name, rows.dtype[name].str) for name in rows.dtype.names])
    attrs = {}
    for k in ids:
        a = {}
        for name, v in h5['basecalled_events/' + batch + '/' + k].attrs.items():
            # (the text attribute is a fixed-width string here, a variable-width one in the
            # reference's file: both read back as bytes)
            a[name] = ['bytes', bytes(v).decode()] if isinstance(v, bytes) else [str(v.dtype), v.item()]
        attrs[k] = a
    np.savez(sys.argv[2], batch=np.array(batch), ids=np.array(ids), rows=rows, attrs=np.array(json.dumps(attrs)),
             offsets=np.concatenate([[0], np.cumsum([len(x) for x in tabs])]).astype(np.int64))
"""


def check_event_dump(outdir, tmp_path, written):
    """events/part-*.h5 of one batch-7 call against the REAL reference's --dump-basecalls
    output for the same reads (tests/golden/dumps0.npz): every row of every table bit for bit,
    every attribute with its type."""
    import json
    import subprocess
    want = np.load(G('dumps0.npz'))
    want_attrs = json.loads(str(want['events_attrs']))
    ids, off, rows = want['events_ids'].tolist(), want['events_offsets'], want['events_rows']
    assert sorted(written) == ['basecalled_events/00000007/' + k for k in ids]
    for k, rid in enumerate(ids):
        table, attrs = written['basecalled_events/00000007/' + rid]
        assert table.dtype == rows.dtype, rid
        assert table.tobytes() == rows[off[k]:off[k + 1]].tobytes(), rid
        got = {}
        for name, v in attrs:
            got[name] = ['bytes', v.decode()] if isinstance(v, bytes) else [str(v.dtype), v.item()]
        assert got == want_attrs[rid], (rid, got, want_attrs[rid])
    if not os.path.exists(PY39) or subprocess.run([PY39, '-c', 'import h5py'], capture_output=True).returncode:
        return False
    back = str(tmp_path / 'events_readback.npz')
    out = subprocess.run([PY39, '-c', READ_EVENTS, outdir, back], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = np.load(back)
    assert str(got['batch']) == '00000007' and got['ids'].tolist() == ids and np.array_equal(got['offsets'], off)
    assert got['rows'].dtype == rows.dtype and got['rows'].tobytes() == rows.tobytes()
    assert json.loads(str(got['attrs'])) == want_attrs
    return True


def run_with_event_dump(ref_results, tmp_path, monkeypatch):
    from poreplex_amd import fast5_write
    from poreplex_amd.signal_analyzer import process_batch
    written = {}
    create = fast5_write.H5Writer.create_dataset

    def recording(self, path, data, attrs=()):
        written[path] = (np.array(data), list(attrs))
        return create(self, path, data, attrs)
    monkeypatch.setattr(fast5_write.H5Writer, 'create_dataset', recording)
    cfg = facade_config(ref_results, dump_basecalls=True)
    cfg['outputdir'] = str(tmp_path)
    got = process_batch(ref_results['batchid'], [tuple(r) for r in ref_results['reads']], cfg)
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref_results['results'], check_polya=True)
    check_event_dump(str(tmp_path), tmp_path, written)


def test_event_dumps_equal_the_reference_dump(oracle_backed, ref_results, tmp_path, monkeypatch):
    """--dump-basecalls (signal_analyzer.py:165-197,259-263,288-309): the event table of every
    read that reaches load_events -- mean, start, stdv, length, model_state, move, pos, end,
    scaled_mean -- and its attributes; results unchanged."""
    run_with_event_dump(ref_results, tmp_path, monkeypatch)


def test_adapter_dumps_equal_the_reference_dump(oracle_backed, ref_results, tmp_path, monkeypatch):
    """--dump-adapter-signals (signal_analyzer.py:155-163,199-208,450-466): same reads, same
    pooled + scaled adapter stretches bit for bit, same catalogue rows in the same order, results
    unchanged."""
    from poreplex_amd import fast5_write
    from poreplex_amd.signal_analyzer import process_batch
    written = {}
    create = fast5_write.H5Writer.create_dataset

    def recording(self, path, data, attrs=()):
        written[path] = np.array(data)
        return create(self, path, data, attrs)
    monkeypatch.setattr(fast5_write.H5Writer, 'create_dataset', recording)
    cfg = facade_config(ref_results, dump_adapter_signals=True)
    cfg['outputdir'] = str(tmp_path)
    got = process_batch(ref_results['batchid'], [tuple(r) for r in ref_results['reads']], cfg)
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref_results['results'], check_polya=True)
    check_adapter_dump(str(tmp_path), tmp_path, written)


def test_barcoding_quality_filter_guard(oracle_backed, ref_results):
    from poreplex_amd.signal_analyzer import process_batch
    out = process_batch(1, [], facade_config(ref_results, barcoding_quality_filter=40))
    assert isinstance(out, tuple) and out[0] == -1 and 'barcoding-quality-filter' in out[1]


def _overlapping_calls(ref_results, workers=3, calls=7, **flags):
    """`calls` process_batch calls kept in flight `workers` at a time on ONE context, the way
    pipeline.py:96,204-205 keeps `parallel` worker calls in flight (threads instead of
    processes: one process per GPU).  Different slices per call, so a mix-up of two calls'
    resident batches cannot go unnoticed."""
    from concurrent.futures import ThreadPoolExecutor
    from poreplex_amd.signal_analyzer import process_batch
    cfg = facade_config(ref_results, **flags)
    reads = [tuple(r) for r in ref_results['reads']]
    slices = [reads[k % 5:len(reads) - (k % 3)] for k in range(calls)]
    serial = [process_batch(100 + k, sl, cfg) for k, sl in enumerate(slices)]
    with ThreadPoolExecutor(workers) as pool:
        together = list(pool.map(lambda a: process_batch(100 + a[0], a[1], cfg), enumerate(slices)))
    for one, two in zip(serial, together):
        assert not (isinstance(two, tuple) and two[0] == -1), two
        compare_results(two, [canon(r) for r in one], check_polya=True)
    compare_results(together[0], [r for r in ref_results['results']], check_polya=True)


def test_overlapping_process_batch_calls_share_one_context(oracle_backed, ref_results):
    _overlapping_calls(ref_results)


def _overlapping_calls_with_dumps(ref_results, tmp_path, workers, calls):
    """The same with both dump options on (the calls then take the three-step path under the
    loader's two locks): results unchanged, and every call leaves ITS dump files -- the one of a
    call over the whole golden batch equal to the reference's dump whichever calls ran beside it."""
    import glob
    import shutil
    _overlapping_calls(ref_results, workers=workers, calls=calls, outputdir=str(tmp_path), dump_adapter_signals=True,
                       dump_basecalls=True)
    for sub in ('adapter-dumps', 'events'):
        names = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / sub / 'part-*.h5')))
        assert len(names) == calls and [n.split('-')[-1] for n in names] == ['%08d.h5' % (100 + k) for k in range(calls)]


def test_overlapping_calls_with_dump_options(oracle_backed, ref_results, tmp_path):
    _overlapping_calls_with_dumps(ref_results, tmp_path, workers=3, calls=5)


@pytest.mark.gpu
def test_overlapping_process_batch_calls_on_the_gpu(ref_results):
    WorkerPersistenceStorage.reset()
    _overlapping_calls(ref_results, workers=3, calls=12)
    WorkerPersistenceStorage.reset()


@pytest.mark.gpu
def test_overlapping_calls_with_dump_options_on_the_gpu(ref_results, tmp_path):
    WorkerPersistenceStorage.reset()
    _overlapping_calls_with_dumps(ref_results, tmp_path, workers=3, calls=8)
    WorkerPersistenceStorage.reset()


@pytest.mark.gpu
def test_process_batch_from_fast5_files_gpu_vs_reference(ref_results, tmp_path):
    """The same on the GPU: FAST5 files -> native reader -> staging arena -> kernels -> the
    REAL reference's result dicts."""
    from poreplex_amd.signal_analyzer import process_batch
    WorkerPersistenceStorage.reset()
    golden_batch_as_fast5(str(tmp_path), ref_results)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path / 'out'), **dict(ref_results['config_flags']))
    got = process_batch(ref_results['batchid'], [tuple(r) for r in ref_results['reads']], cfg)
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref_results['results'], check_polya=True)
    WorkerPersistenceStorage.reset()


@pytest.mark.gpu
def test_process_batch_gpu_vs_reference(ref_results):
    from poreplex_amd.signal_analyzer import process_batch
    WorkerPersistenceStorage.reset()
    reads = [tuple(r) for r in ref_results['reads']]
    cfg = facade_config(ref_results)        # measure_polya on, as the reference ran it
    got = process_batch(ref_results['batchid'], reads, cfg)
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref_results['results'], check_polya=True)
    WorkerPersistenceStorage.reset()


@pytest.mark.gpu
def test_adapter_dumps_gpu_vs_reference(ref_results, tmp_path, monkeypatch):
    """--dump-adapter-signals on the GPU: the stretches come from pxg_batch_pooled_signal on the
    resident batch; file and catalogue equal the REAL reference's dump."""
    from poreplex_amd import fast5_write
    from poreplex_amd.signal_analyzer import process_batch
    WorkerPersistenceStorage.reset()
    written = {}
    create = fast5_write.H5Writer.create_dataset

    def recording(self, path, data, attrs=()):
        written[path] = np.array(data)
        return create(self, path, data, attrs)
    monkeypatch.setattr(fast5_write.H5Writer, 'create_dataset', recording)
    cfg = facade_config(ref_results, dump_adapter_signals=True)
    cfg['outputdir'] = str(tmp_path)
    got = process_batch(ref_results['batchid'], [tuple(r) for r in ref_results['reads']], cfg)
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref_results['results'], check_polya=True)
    check_adapter_dump(str(tmp_path), tmp_path, written)
    WorkerPersistenceStorage.reset()


@pytest.mark.gpu
def test_event_dumps_gpu_vs_reference(ref_results, tmp_path, monkeypatch):
    """--dump-basecalls on the GPU: mean / stdv / scaled mean from pxg_batch_event_table on the
    resident batch; tables and attributes equal the REAL reference's dump."""
    WorkerPersistenceStorage.reset()
    run_with_event_dump(ref_results, tmp_path, monkeypatch)
    WorkerPersistenceStorage.reset()


# ---- a18/a19: --filter-chimera against the real reference's process_batch ------
def chimera_bundle():
    return G('chimera.pxr.npz')


def chimera_case():
    import json
    with open(G('chimera.results.json')) as fh:
        ref = json.load(fh)
    cfg = default_config(inputdir='/nonexistent-inputdir', outputdir='/tmp',
                         read_bundle=chimera_bundle(), **ref['config_flags'])
    return ref, cfg


def check_chimera(got, ref):
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref['results'], check_polya=True)
    by_status = {}
    for r in got:
        by_status.setdefault(r['status'], []).append(r['label'])
    assert set(by_status) == {'okay', 'unsplit_read'}
    n_ref = sum(r['status'] == 'unsplit_read' for r in ref['results'])          # 6 in the f32 set, 5 in the q8 set
    assert set(by_status['unsplit_read']) == {'artifact'} and len(by_status['unsplit_read']) == n_ref >= 5


def test_filter_unsplit_reads_host_logic_vs_reference(oracle_backed, arith):
    from poreplex_amd.signal_analyzer import process_batch
    ref, cfg = chimera_case()
    assert cfg['filter_unsplit_reads']
    check_chimera(process_batch(ref['batchid'], [tuple(r) for r in ref['reads']], cfg), ref)


def test_filter_unsplit_reads_one_call_form_vs_reference(one_call_double, arith, monkeypatch):
    """The same batch the way the GPU context serves it -- records, spike rows and scan candidates from one call --:
    as one worker call (a plain run of bundle reads with candidates in some of them: a table of just those beside the C
    pass), in reference-sized slices, and with everything on the batch table."""
    from poreplex_amd import signal_analyzer as SA
    ref, cfg = chimera_case()
    reads = [tuple(r) for r in ref['reads']]
    before = SA.PLAIN_RUN_CALLS
    check_chimera(SA.process_batch(ref['batchid'], reads, cfg), ref)
    parts = []
    for lo in range(0, len(reads), 5):
        parts += SA.process_batch(ref['batchid'], reads[lo:lo + 5], cfg)
    check_chimera(parts, ref)
    if SA._PLAIN_RUN and N.load_pyhost() is not None:
        assert SA.PLAIN_RUN_CALLS > before
    monkeypatch.setattr(SA, '_PLAIN_RUN', False)
    check_chimera(SA.process_batch(ref['batchid'], reads, cfg), ref)
    monkeypatch.setattr(SA, '_BULK_UNSPLIT', False)
    check_chimera(SA.process_batch(ref['batchid'], reads, cfg), ref)


def test_golden_batch_one_call_form_vs_reference(one_call_double, ref_results, tmp_path):
    """The golden batch through the one-call double: from the bundle, and from FAST5 files."""
    from poreplex_amd.signal_analyzer import process_batch
    reads = [tuple(r) for r in ref_results['reads']]
    compare_results(process_batch(ref_results['batchid'], reads, facade_config(ref_results)), ref_results['results'],
                    check_polya=True)
    WorkerPersistenceStorage.reset()
    golden_batch_as_fast5(str(tmp_path), ref_results)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path / 'out'), **dict(ref_results['config_flags']))
    compare_results(process_batch(ref_results['batchid'], reads, cfg), ref_results['results'], check_polya=True)


def test_filter_unsplit_reads_from_multi_read_fast5_one_call_form(one_call_double, tmp_path, arith):
    """--filter-chimera with the chimera batch as ONE multi-read FAST5 file, served in the file's own read order in
    reference-sized calls: the plain-run path over per-call bundles with the scan on (what an unchanged caller of the
    GPU build gets), against the real reference's verdicts read by read."""
    from poreplex_amd import signal_analyzer as SA
    from poreplex_amd.fast5_file import ReadBundle, get_read_ids
    from poreplex_amd.fast5_write import Fast5Writer
    ref, _ = chimera_case()
    b = ReadBundle(chimera_bundle())
    with Fast5Writer(str(tmp_path / 'all.fast5')) as w:
        for i in range(len(b.read_ids)):
            bc = b.basecall_of(i)
            if bc is not None:
                bc['move'] = np.asarray(bc['move'], dtype=np.uint8)
            w.add_read(b.read_ids[i], b.samples(i), b.d['calib'][i], start_time=int(b.d['start_time'][i]),
                       channel_number=str(b.d['channel_number'][i]), run_id=str(b.d['run_id'][i]),
                       sample_id=str(b.d['sample_id'][i]), basecall=bc)
    keys = get_read_ids('all.fast5', str(tmp_path))
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path / 'out'), **ref['config_flags'])
    want = {r['read_id']: r for r in ref['results']}
    before = SA.PLAIN_RUN_CALLS
    got = []
    for lo in range(0, len(keys), 7):
        got += SA.process_batch(ref['batchid'], keys[lo:lo + 7], cfg)
    assert len(got) == len(want)
    compare_results(got, [dict(want[r['read_id']], filename='all.fast5') for r in got], check_polya=True)
    assert sum(r['status'] == 'unsplit_read' for r in got) >= 5
    if SA._PLAIN_RUN and N.load_pyhost() is not None:
        assert SA.PLAIN_RUN_CALLS > before


def chimera_from_fast5(tmp_path):
    """--filter-chimera with the reads as FAST5 files in inputdir (no bundle): the event frames
    come from the files' Move tables through the native reader."""
    from poreplex_amd.signal_analyzer import process_batch
    ref, cfg = chimera_case()
    golden_batch_as_fast5(str(tmp_path), bundle=chimera_bundle())
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path / 'out'), **ref['config_flags'])
    check_chimera(process_batch(ref['batchid'], [tuple(r) for r in ref['reads']], cfg), ref)


def test_filter_unsplit_reads_from_fast5_files_vs_reference(oracle_backed, tmp_path, arith):
    chimera_from_fast5(tmp_path)


@pytest.mark.gpu
def test_filter_unsplit_reads_from_fast5_files_gpu_vs_reference(tmp_path, arith):
    WorkerPersistenceStorage.reset()
    chimera_from_fast5(tmp_path)
    WorkerPersistenceStorage.reset()


def test_event_frame_base_space_columns_vs_reference(oracle_backed, arith):
    """pos / p_model_state of the Move-table frame equal the reference's load_events."""
    from poreplex_amd import signal_analyzer as SA
    ref, cfg = chimera_case()
    st = np.load(G('chimera.stages.npz'))
    eo = st['ev_offsets']
    with SA.SignalAnalyzer(cfg, 8) as an:
        for i, (fn, rid) in enumerate(ref['reads'][:4]):
            npread = an.loader.prepare_loading(fn, rid)
            npread.set_scaling_params(np.float32([1, 0]))
            ev = SA.SignalAnalysis(npread, an).load_events()
            assert np.array_equal(ev['pos'], st['ev_pos'][eo[i]:eo[i + 1]])
            # 10 ** x differs by <= 1 ulp between NumPy builds (py3.9 reference vs here);
            # the only consumer compares it with basecount_quality_limit = 0.4
            want = st['ev_pms'][eo[i]:eo[i + 1]]
            assert np.abs(ev['p_model_state'] - want).max() <= 2.3e-16
            assert np.array_equal(ev['p_model_state'] > 0.4, want > 0.4)
        an.loader.clear()


@pytest.mark.gpu
def test_filter_unsplit_reads_gpu_vs_reference():
    from poreplex_amd.signal_analyzer import process_batch
    WorkerPersistenceStorage.reset()
    ref, cfg = chimera_case()
    check_chimera(process_batch(ref['batchid'], [tuple(r) for r in ref['reads']], cfg), ref)
    WorkerPersistenceStorage.reset()


# ---- a18/a19 over albacore's 14-column Events tables (the reference's primary input: fast5_file.py:178-179) ----
def albacore_case(tmp_path):
    import json
    with open(G('albacore.results.json')) as fh:
        ref = json.load(fh)
    cfg = default_config(inputdir='/nonexistent-inputdir', outputdir=str(tmp_path), read_bundle=G('albacore.pxr.npz'),
                         **ref['config_flags'])
    return ref, cfg


def run_albacore(tmp_path, monkeypatch):
    """--filter-chimera + --dump-basecalls over reads whose basecall is an albacore Events table (variable-
    length events, their own `start' column) against what the REAL reference returned and dumped for the same
    tables (tests/golden/albacore.*, tools/make_golden.py): result dicts, the in-read adapter candidates of
    every read, every row and attribute of the dumped tables.  Two of the reads carry uint64 `start' columns,
    the dtype albacore itself writes: there the reference's own arithmetic fails (float64 `end', range() at
    signal_analyzer.py:385) and the read becomes unknown_error -- after its table was dumped."""
    import json
    from poreplex_amd import fast5_write, signal_loader
    from poreplex_amd.signal_analyzer import process_batch
    ref, cfg = albacore_case(tmp_path)
    written, cands = {}, {}
    create = fast5_write.H5Writer.create_dataset
    attach = signal_loader.SignalLoader.attach_unsplit

    def recording(self, path, data, attrs=()):
        written[path] = (np.array(data), list(attrs))
        return create(self, path, data, attrs)

    def spying(self, table, rows, sel, scanned):
        attach(self, table, rows, sel, scanned)
        for k in np.nonzero(sel)[0].tolist():
            cands[str(table.read_id[rows[k]])] = [list(map(int, iv)) for iv in (table.unsplit[rows[k]] or [])]
    monkeypatch.setattr(fast5_write.H5Writer, 'create_dataset', recording)
    monkeypatch.setattr(signal_loader.SignalLoader, 'attach_unsplit', spying)
    got = process_batch(ref['batchid'], [tuple(r) for r in ref['reads']], cfg)
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref['results'], check_polya=True)
    statuses = [r['status'] for r in got]
    assert statuses.count('unsplit_read') >= 2 and statuses.count('unknown_error') == 2 and statuses.count('okay') >= 2
    # candidates: the reference's excessive_adapters list of every read it scanned (append order)
    for (fn, rid), want in zip(ref['reads'], ref['candidates']):
        if rid in cands:
            assert cands[rid] == want, (rid, cands[rid], want)
        else:
            assert want == [], rid
    assert sum(1 for c in ref['candidates'] if c) >= 2
    # dumps
    st = np.load(G('albacore.stages.npz'))
    want_attrs = json.loads(str(st['events_attrs']))
    ids, off, rows = st['events_ids'].tolist(), st['events_offsets'], st['events_rows']
    assert sorted(written) == ['basecalled_events/00000009/' + k for k in ids]
    for k, rid in enumerate(ids):
        table, attrs = written['basecalled_events/00000009/' + rid]
        assert table.dtype == rows.dtype, rid
        assert table.tobytes() == rows[off[k]:off[k + 1]].tobytes(), rid
        gota = {}
        for name, v in attrs:
            gota[name] = ['bytes', v.decode()] if isinstance(v, bytes) else [str(v.dtype), v.item()]
        assert gota == want_attrs[rid], (rid, gota, want_attrs[rid])


def test_albacore_events_tables_host_logic_vs_reference(oracle_backed, arith, tmp_path, monkeypatch):
    run_albacore(tmp_path, monkeypatch)


@pytest.mark.gpu
def test_albacore_events_tables_gpu_vs_reference(arith, tmp_path, monkeypatch):
    WorkerPersistenceStorage.reset()
    run_albacore(tmp_path, monkeypatch)
    WorkerPersistenceStorage.reset()
