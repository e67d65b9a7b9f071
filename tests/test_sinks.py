"""Result sinks against the REAL reference classes' outputs
(tests/golden/sinks.json, tools/make_golden_sinks.py)."""
import gzip
import io
import json
import os

import numpy as np
import pytest

from poreplex_amd import io as SINK

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
