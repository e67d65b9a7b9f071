"""Result sinks against the REAL reference classes' outputs
(tests/golden/sinks.json, tools/make_golden_sinks.py)."""
import gzip
import io
import json
import os

import numpy as np
import pytest

from poreplex_amd import sinks as SINK

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def sinks():
    with open(os.path.join(GOLDEN, 'sinks.json')) as fh:
        return json.load(fh)


def results_of(case):
    if case.get('results') is not None:
        rs = case['results']
    else:
        with open(os.path.join(GOLDEN, case['source'])) as fh:
            rs = json.load(fh)['results']
    out = []
    for r in rs:
        r = dict(r)
        if 'sequence' in r:
            r['sequence'] = tuple(r['sequence'])
        out.append(r)
    return out


def as_key(k):
    return tuple(k)


@pytest.mark.parametrize('tag', ['batch0', 'chimera', 'mixed'])
def test_name_mapping_summary_and_final_table(sinks, tag, tmp_path):
    case = sinks[tag]
    cfg = case['config']
    labels, barcodes, layout = SINK.setup_output_name_mapping(cfg)
    assert labels == case['label_names']
    assert [[k, v] for k, v in barcodes.items()] == case['barcode_names']
    assert {k: v for k, v in layout.items()} == {as_key(k): v for k, v in case['layout']}
    results = results_of(case)
    w = SINK.SequencingSummaryWriter(cfg, str(tmp_path), labels, barcodes)
    w.write_results(results)
    w.close()
    assert (tmp_path / 'sequencing_summary.txt').read_text() == case['sequencing_summary']
    trk = SINK.FinalSummaryTracker(labels, barcodes)
    trk.feed_results(results)
    buf = io.StringIO()
    trk.print_results(buf)
    assert buf.getvalue() == case['final_summary']
    fq = SINK.FASTQWriter(str(tmp_path), layout)
    fq.write_sequences(results)
    fq.close()
    for name, want in case['fastq'].items():
        with gzip.open(fq.get_output_path(name), 'rt') as fh:
            assert fh.read() == want, name


@pytest.mark.parametrize('tag', ['batch0', 'chimera'])
def test_barcoding_off_variant(sinks, tag, tmp_path):
    case = sinks[tag]
    cfg = dict(case['config'], barcoding=False)
    nb = case['nobarcoding']
    labels, barcodes, layout = SINK.setup_output_name_mapping(cfg)
    assert labels == nb['label_names'] and [[k, v] for k, v in barcodes.items()] == nb['barcode_names']
    assert {k: v for k, v in layout.items()} == {as_key(k): v for k, v in nb['layout']}
    results = [{k: v for k, v in r.items() if not k.startswith('barcode')} for r in results_of(case)]
    w = SINK.SequencingSummaryWriter(cfg, str(tmp_path), labels, barcodes)
    w.write_results(results)
    w.close()
    assert (tmp_path / 'sequencing_summary.txt').read_text() == nb['sequencing_summary']
    trk = SINK.FinalSummaryTracker(labels, barcodes)
    trk.feed_results(results)
    buf = io.StringIO()
    trk.print_results(buf)
    assert buf.getvalue() == nb['final_summary']


def test_final_table_from_the_all_reduced_count_table(sinks):
    """feed_counts(count table) == feed_results(dicts): the table the N-GPU all-reduce
    produces carries everything the summary needs."""
    from poreplex_amd import native as N
    from poreplex_amd.distributed import LABEL_NAMES
    case = sinks['mixed']
    results = results_of(case)
    labels, barcodes, _ = SINK.setup_output_name_mapping(case['config'])
    a = SINK.FinalSummaryTracker(labels, barcodes)
    a.feed_results(results)
    tbl = np.zeros((len(LABEL_NAMES), 5, len(N.STATUS_NAMES)), dtype=np.int64)
    for r in results:
        bc = r.get('barcode')
        tbl[LABEL_NAMES.index(r.get('label', 'fail')), 0 if bc is None else bc + 1,
            N.STATUS_NAMES.index(r['status'])] += 1
    b = SINK.FinalSummaryTracker(labels, barcodes)
    b.feed_counts(tbl[:, :, :] * 1, label_order=LABEL_NAMES)
    b.feed_counts(tbl * 0, label_order=LABEL_NAMES)
    assert dict(a.counts) == dict(b.counts)
    buf = io.StringIO()
    b.print_results(buf)
    assert buf.getvalue().splitlines()[:2] == case['final_summary'].splitlines()[:2]


def test_process_batch_into_the_sinks_end_to_end(sinks, tmp_path, monkeypatch):
    """process_batch (host logic over the oracle-backed context double) -> the sinks: the
    sequencing summary and the final table of the real reference, byte for byte."""
    import test_facade as TF
    from poreplex_amd import native as N
    from poreplex_amd.worker_persistence import WorkerPersistenceStorage
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', TF.OracleBackedContext)
    try:
        with open(os.path.join(GOLDEN, 'batch0.results.json')) as fh:
            ref = json.load(fh)
        from poreplex_amd.signal_analyzer import process_batch
        got = process_batch(ref['batchid'], [tuple(r) for r in ref['reads']], TF.facade_config(ref))
        assert not (isinstance(got, tuple) and got[0] == -1), got
        got = [r for r in got if r['status'] != 'unknown_error']     # traceback text differs
        case = sinks['batch0']
        labels, barcodes, _ = SINK.setup_output_name_mapping(case['config'])
        w = SINK.SequencingSummaryWriter(case['config'], str(tmp_path), labels, barcodes)
        w.write_results(got)
        w.close()
        assert (tmp_path / 'sequencing_summary.txt').read_text() == case['sequencing_summary']
        trk = SINK.FinalSummaryTracker(labels, barcodes)
        trk.feed_results(got)
        assert sum(trk.counts.values()) == len(got)
    finally:
        WorkerPersistenceStorage.reset()


def _random_table(tmp_path, n, seed, unicode_name=False):
    """A ReadTable over a small bundle with every column the summary prints randomised."""
    from poreplex_amd import native as N
    from poreplex_amd.fast5_file import write_bundle, ReadBundle
    from poreplex_amd.signal_loader import ReadTable
    rng = np.random.default_rng(seed)
    lens = rng.integers(20, 60, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    arena = rng.integers(0, 1000, int(off[-1])).astype(np.int16)
    cal = np.zeros(n, dtype=N.CALIB_DTYPE)
    cal['range'], cal['digitisation'], cal['offset'] = 1400.0, 8192.0, 10.0
    cal['sampling_rate'] = rng.choice([3012.0, 4000.0, 3012.5], n)
    names = ['dir{}/r{:05d}.fast5'.format(i % 3, i) for i in range(n)]
    if unicode_name:
        names[n // 2] = 'dir0/résumé.fast5'
    ids = ['{:08x}-{:04x}'.format(int(rng.integers(1 << 31)), i) for i in range(n)]
    path = str(tmp_path / 'rand{}.pxr.npz'.format(seed))
    write_bundle(path, arena, off, cal, names, ids)
    b = ReadBundle(path)
    b.d['start_time'] = rng.choice([0, 1, 1506, 3012, 123456789, 10 ** 12], n).astype(np.int64) * rng.integers(1, 9, n)
    b.d['duration'] = rng.integers(0, 10 ** 7, n).astype(np.int64)
    b.d['channel_number'] = np.array([str(int(c)) for c in rng.integers(1, 513, n)])
    b.d['run_id'] = np.array([('run' + 'x' * int(c)) for c in rng.integers(0, 30, n)])
    b.d['sample_id'] = np.array([('s' * int(c)) for c in rng.integers(0, 5, n)])
    t = ReadTable()
    rows = t.extend_from_bundle(b, np.arange(n))
    t.label[rows] = rng.integers(0, 3, n)
    t.status[rows] = rng.integers(0, len(N.STATUS_NAMES), n)
    t.has_barcode[rows] = rng.random(n) < 0.6
    t.barcode[rows] = rng.integers(0, 4, n)
    t.barcode_phred[rows] = rng.integers(0, 30, n)
    t.has_summary[rows] = rng.random(n) < 0.8
    t.num_events[rows] = rng.integers(0, 10 ** 6, n)
    t.sequence_length[rows] = rng.integers(0, 10 ** 5, n)
    q = np.concatenate([rng.uniform(0, 40, n - 12).astype(np.float32).astype(np.float64),
                        [0.0, 1.0, 10.0, 100.0, 1e-5, 1.5e-4, 1e16, 1.25e17, 123456.0, 0.1, 1e15, 9.87]])
    t.mean_qscore[rows] = rng.permutation(q)
    t.polya_lazy[rows] = rng.random(n) < 0.5
    t.polya_dwell_time[rows] = np.where(rng.random(n) < 0.2, rng.integers(0, 3, n) * 0.00005, rng.uniform(0, 3, n))
    for i in rng.choice(n, 5, replace=False).tolist():       # a tail that came through set_polya_tail
        t.polya_lazy[i] = False
        t.polya[i] = {'begin': 1, 'end': 2, 'dwell_time': 0.12345, 'spikes': []}
    return t, rows


@pytest.mark.parametrize('barcoding,polya', [(True, True), (True, False), (False, True), (False, False)])
def test_native_summary_rows_equal_the_python_writer(tmp_path, barcoding, polya):
    """pxg_summary_rows (C, libpxg.so; no GPU involved) prints the bytes the Python writer
    prints: str(int), repr(float) in both layouts, round(x, 3), '%.4f', every name table."""
    import io
    from poreplex_amd.config import default_config
    from poreplex_amd.signal_loader import summary_columns
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), barcoding=barcoding,
                         measure_polya=polya, filter_unsplit_reads=True)
    labels, barcodes, _ = SINK.setup_output_name_mapping(cfg)
    for seed in range(4):
        t, rows = _random_table(tmp_path, 300, seed)
        w = SINK.SequencingSummaryWriter(dict(cfg, fast5_output=False), str(tmp_path), labels, barcodes)
        w.file.close()
        w.file = io.StringIO()
        assert w.write_table_rows(t, rows)
        native_text = w.file.getvalue()
        w.file = io.StringIO()
        w.write_columns(summary_columns(t, rows, barcoding, polya))
        assert native_text == w.file.getvalue()
        assert native_text.count('\n') == len(rows)


def test_native_summary_rows_decline_what_they_cannot_print(tmp_path):
    import io
    from poreplex_amd.config import default_config
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), barcoding=True, filter_unsplit_reads=True)
    labels, barcodes, _ = SINK.setup_output_name_mapping(cfg)
    t, rows = _random_table(tmp_path, 40, 9, unicode_name=True)
    w = SINK.SequencingSummaryWriter(dict(cfg, fast5_output=False), str(tmp_path), labels, barcodes)
    w.file.close()
    w.file = io.StringIO()
    assert w.write_table_rows(t, rows) is False and w.file.getvalue() == ''      # non-ASCII file name
    w2 = SINK.SequencingSummaryWriter(dict(cfg, fast5_output=True), str(tmp_path), labels, barcodes)
    assert w2.write_table_rows(t, rows[:3]) is False                               # fast5 layout
    w2.close()
