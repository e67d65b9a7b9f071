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

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
BUNDLE = os.path.join(GOLDEN, 'batch0.pxr.npz')


def facade_config(ref_results, **kw):
    flags = dict(ref_results['config_flags'])
    flags.update(kw)
    return default_config(inputdir='/nonexistent-inputdir', outputdir='/tmp',
                          read_bundle=BUNDLE, **flags)


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


from oracle_context import OracleBackedContext  # noqa: E402  (tests/oracle_context.py)


@pytest.fixture()
def oracle_backed(monkeypatch):
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', OracleBackedContext)
    yield
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


@pytest.mark.parametrize('option', ['dump_adapter_signals', 'dump_basecalls'])
def test_dump_options_are_fenced_before_any_read_is_touched(oracle_backed, ref_results, option):
    """The one place the operator surface says no (signal_analyzer.py:155-211,450-466: HDF5 debug
    dumps): the call fails as a WHOLE, with the reference's fatal-tuple convention and a message
    that names the option -- never a silent run without the dump."""
    from poreplex_amd.signal_analyzer import process_batch
    out = process_batch(1, [tuple(r) for r in ref_results['reads'][:4]], facade_config(ref_results, **{option: True}))
    assert isinstance(out, tuple) and out[0] == -1
    assert 'NotImplementedError' in out[1] and 'dump' in out[1]


def test_barcoding_quality_filter_guard(oracle_backed, ref_results):
    from poreplex_amd.signal_analyzer import process_batch
    out = process_batch(1, [], facade_config(ref_results, barcoding_quality_filter=40))
    assert isinstance(out, tuple) and out[0] == -1 and 'barcoding-quality-filter' in out[1]


def _overlapping_calls(ref_results, workers=3, calls=7):
    """`calls` process_batch calls kept in flight `workers` at a time on ONE context, the way
    pipeline.py:96,204-205 keeps `parallel` worker calls in flight (threads instead of
    processes: one process per GPU).  Different slices per call, so a mix-up of two calls'
    resident batches cannot go unnoticed."""
    from concurrent.futures import ThreadPoolExecutor
    from poreplex_amd.signal_analyzer import process_batch
    cfg = facade_config(ref_results)
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


@pytest.mark.gpu
def test_overlapping_process_batch_calls_on_the_gpu(ref_results):
    WorkerPersistenceStorage.reset()
    _overlapping_calls(ref_results, workers=3, calls=12)
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


# ---- a18/a19: --filter-chimera against the real reference's process_batch ------
CHIMERA_BUNDLE = os.path.join(GOLDEN, 'chimera.pxr.npz')


def chimera_case():
    import json
    with open(os.path.join(GOLDEN, 'chimera.results.json')) as fh:
        ref = json.load(fh)
    cfg = default_config(inputdir='/nonexistent-inputdir', outputdir='/tmp',
                         read_bundle=CHIMERA_BUNDLE, **ref['config_flags'])
    return ref, cfg


def check_chimera(got, ref):
    assert not (isinstance(got, tuple) and got[0] == -1), got
    compare_results(got, ref['results'], check_polya=True)
    by_status = {}
    for r in got:
        by_status.setdefault(r['status'], []).append(r['label'])
    assert set(by_status) == {'okay', 'unsplit_read'}
    assert set(by_status['unsplit_read']) == {'artifact'} and len(by_status['unsplit_read']) == 6


def test_filter_unsplit_reads_host_logic_vs_reference(oracle_backed):
    from poreplex_amd.signal_analyzer import process_batch
    ref, cfg = chimera_case()
    assert cfg['filter_unsplit_reads']
    check_chimera(process_batch(ref['batchid'], [tuple(r) for r in ref['reads']], cfg), ref)


def test_event_frame_base_space_columns_vs_reference(oracle_backed):
    """pos / p_model_state of the Move-table frame equal the reference's load_events."""
    from poreplex_amd import signal_analyzer as SA
    ref, cfg = chimera_case()
    st = np.load(os.path.join(GOLDEN, 'chimera.stages.npz'))
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
