"""csrc/pxg_pyreport.c (the CPython extension that builds the result dicts of a worker batch
from the batch table's columns) against ReadTable.report's Python loop -- the statement of
NanoporeRead.report (signal_loader.py:165-198): same keys, key order, value types and values
on randomised tables that hit every optional key."""
import json

import numpy as np
import pytest

from poreplex_amd import native
from poreplex_amd.signal_loader import LABELS, ReadTable


class _Source:
    def __init__(self, rng):
        self.start_time = int(rng.integers(0, 10**9))
        self.duration = int(rng.integers(1, 10**6))
        self.sampling_rate = float(rng.choice([3012.0, 4000.0, 3012.5, 7.0]))
        self.range, self.digitization, self.offset = 1200.0, 8192.0, 3.0
        self.channel_number = str(int(rng.integers(1, 513)))
        self.run_id, self.sample_id = 'run' + 'ab' * int(rng.integers(1, 6)), 'sample'

    def close(self):
        pass


def random_table(rng, n):
    t = ReadTable()
    for i in range(n):
        t.append('dir/f{:04d}.fast5'.format(i), 'id-{:06d}'.format(i), _Source(rng))
    t.status[:n] = rng.integers(0, len(native.STATUS_NAMES), n)
    t.label[:n] = rng.integers(-1, len(LABELS), n)
    t.has_summary[:n] = rng.random(n) < 0.7
    t.num_events[:n] = rng.integers(0, 10**5, n)
    t.sequence_length[:n] = rng.integers(0, 5000, n)
    t.mean_qscore[:n] = rng.uniform(3, 14, n).astype(np.float32)
    t.has_barcode[:n] = rng.random(n) < 0.6
    t.barcode[:n] = rng.integers(-1, 4, n)
    t.barcode_guess[:n] = rng.integers(-1, 4, n)
    t.barcode_phred[:n] = rng.integers(0, 30, n)
    t.gpu_row[:n] = rng.permutation(n)
    t.gpu_row[:n][rng.random(n) < 0.1] = -1
    per_record = rng.integers(0, 9, n)                      # spike rows of GPU record g (CSR)
    per_record[rng.integers(0, n)] = 300                    # a tail with hundreds of spikes
    t.spike_offsets = np.concatenate([[0], np.cumsum(per_record)]).astype(np.int64)
    t.spikes = rng.normal(size=(int(t.spike_offsets[-1]), 4)).astype(np.float32)
    for i in range(n):
        u = rng.random()
        if u < 0.3:
            t.sequence[i] = ('ACGU' * int(rng.integers(1, 50)), '!' * 7, int(rng.integers(0, 3)))
        if rng.random() < 0.2:
            t.error_message[i] = 'boom {}'.format(i)
        elif rng.random() < 0.1:
            t.error_message[i] = ''
        u = rng.random()
        if u < 0.25 and t.gpu_row[i] >= 0:        # a tail comes from a GPU record
            t.polya_lazy[i] = True
            t.polya_begin[i], t.polya_end[i] = int(rng.integers(0, 10**5)), int(rng.integers(10**5, 10**6))
            t.polya_dwell_time[i] = float(rng.uniform(0, 3))
            t.polya_spike_count[i] = per_record[t.gpu_row[i]]
        elif u < 0.35:
            t.polya[i] = {'begin': 1, 'end': 2, 'dwell_time': 0.5, 'spikes': []}
    return t


def both_ways(table, rows, monkeypatch):
    fast = native.load_pyhost()
    if fast is None:
        pytest.skip('csrc/_pxgpy was not built for this interpreter')
    got = table.report(rows)
    monkeypatch.setattr(native, '_pyhost', None)
    want = table.report(rows)
    monkeypatch.setattr(native, '_pyhost', fast)
    return got, want


def assert_same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert list(g) == list(w)                       # keys and their order
        for k in w:
            assert type(g[k]) is type(w[k]), (k, g[k], w[k])
            assert g[k] == w[k], (k, g[k], w[k])
        assert json.dumps(g) == json.dumps(w)


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_extension_builds_the_dicts_of_the_python_loop(seed, monkeypatch):
    rng = np.random.default_rng(seed)
    t = random_table(rng, 400)
    rows = rng.permutation(400)[:300]
    got, want = both_ways(t, rows, monkeypatch)
    assert_same(got, want)
    assert any('polya' in r and r['polya']['spikes'] for r in got)
    assert any('error_message' in r for r in got) and any('barcode' in r for r in got)
    assert any(r['mean_qscore'] == 0 and isinstance(r['mean_qscore'], int) for r in got)


def test_start_time_is_pythons_round(monkeypatch):
    """round(a / b, 3) is correctly rounded decimal (float.__round__), not floor(x*1000+.5)."""
    rng = np.random.default_rng(9)
    t = random_table(rng, 2000)
    t.start_time[:2000] = rng.integers(0, 2**40, 2000)
    t.start_time[:6] = [1, 5, 15, 25, 2**53 - 1, 0]
    t.sampling_rate[:6] = [2000.0, 10000.0, 10000.0, 10000.0, 3.0, 3012.0]
    got, want = both_ways(t, np.arange(2000), monkeypatch)
    assert [g['start_time'] for g in got] == [w['start_time'] for w in want]
    assert want[0]['start_time'] == round(1 / 2000.0, 3)


def test_bundle_sequences_are_read_lazily(monkeypatch, bundle):
    """Rows that point into a read bundle get their (sequence, qstring, 0) tuple from the
    bundle's text arenas."""
    import os
    from conftest import GOLDEN
    from poreplex_amd.fast5_file import ReadBundle
    b = ReadBundle(os.path.join(GOLDEN, 'batch0.pxr.npz'))
    t = ReadTable()
    rows = t.extend_from_bundle(b, np.arange(len(b.read_ids)))
    t.seq_lazy[rows] = b.d['bc_present']
    got, want = both_ways(t, rows, monkeypatch)
    assert_same(got, want)
    assert sum('sequence' in r for r in got) == int(b.d['bc_present'].sum()) > 0


def test_bad_rows_raise_instead_of_reading_past_a_column():
    fast = native.load_pyhost()
    if fast is None:
        pytest.skip('csrc/_pxgpy was not built for this interpreter')
    t = random_table(np.random.default_rng(4), 10)
    with pytest.raises(IndexError):
        fast.report(t._report_columns(), np.array([0, 64], dtype=np.int64))
    t.sampling_rate[3] = 0.0
    with pytest.raises(ZeroDivisionError):
        fast.report(t._report_columns(), np.array([3], dtype=np.int64))
