"""`--trim-adapter` (signal_analyzer.py:328-344).  In the reference revision this repository follows the function returns
at once for every basecalled read (`if sequence is not None: return`): a no-op, reproduced by default and pinned by the
32-read golden batch (tests/test_facade.py, tests/test_sinks.py).  `trim_adapter_as_intended` (not a reference option)
runs what the function was written to do; this test computes the expected trimming lengths independently from the
golden batch's own basecall tables and segment boundaries.  CPU (oracle-backed context double)."""
import json
import os

import numpy as np
import pytest

import test_facade as TF
from poreplex_amd import native as N
from poreplex_amd.signal_analyzer import process_batch
from poreplex_amd.worker_persistence import WorkerPersistenceStorage

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture
def oracle_backed(monkeypatch):
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', TF.OracleBackedContext)
    yield
    WorkerPersistenceStorage.reset()


def _run(**flags):
    with open(os.path.join(GOLDEN, 'batch0.results.json')) as fh:
        ref = json.load(fh)
    got = process_batch(ref['batchid'], [tuple(r) for r in ref['reads']], TF.facade_config(ref, **flags))
    assert not (isinstance(got, tuple) and got[0] == -1), got
    return ref, got


def test_reference_behaviour_is_a_no_op(oracle_backed):
    ref, plain = _run(trim_adapter=False)
    _, trimmed = _run(trim_adapter=True)
    assert [TF.canon(r) for r in trimmed if r['status'] != 'unknown_error'] == \
           [TF.canon(r) for r in plain if r['status'] != 'unknown_error']
    assert all(r['sequence'][2] == 0 for r in trimmed if r.get('sequence'))


def test_intended_trimming_matches_an_independent_count(oracle_backed):
    ref, plain = _run(trim_adapter=False)
    _, got = _run(trim_adapter=True, trim_adapter_as_intended=True)
    bundle = dict(np.load(os.path.join(GOLDEN, 'batch0.pxr.npz')))
    stages = dict(np.load(os.path.join(GOLDEN, 'batch0.stages.npz')))
    index = {str(rid): i for i, rid in enumerate(bundle['read_id'])}
    adapter = 3                       # state order of the segmentation model: pre-leader, leader-low, leader-high, adapter, ...
    by_id = {r['read_id']: r for r in plain if 'read_id' in r}
    checked = trimmed = 0
    for r in got:
        if r['status'] == 'unknown_error' or not r.get('sequence'):
            continue
        i = index[r['read_id']]
        bc = json.loads(str(bundle['basecall'][i]))
        before = by_id[r['read_id']]
        assert r['status'] in ('okay', 'sequence_too_short', 'basecall_table_incomplete') or r['status'] == before['status']
        if not stages['has_seg'][i] or stages['seg_first'][i][adapter] < 0 or r['status'] == 'basecall_table_incomplete':
            continue
        start = bc['first_sample_template'] + bc['block_stride'] * np.arange(len(bc['move']))
        adapter_end = int(stages['seg_last'][i][adapter]) * 15
        inside = start <= adapter_end
        want = int(np.asarray(bc['move'])[inside].sum()) + 2 if inside.any() else 0        # k-mer size 5: two leading bases
        if want > len(bc['sequence']):
            continue
        assert r['sequence'][0] == before['sequence'][0]
        assert r['sequence'][2] == want, (r['read_id'], r['sequence'][2], want)
        checked += 1
        trimmed += want > 0
    assert checked >= 10 and trimmed >= 8
