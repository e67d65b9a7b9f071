"""a18 (Guppy event table) and a19 (pseudo-fusion window scan): oracle vs the
REAL fast5_file.py / signal_analyzer.py outputs (tests/golden/chimera.*)."""
import json
import os

import numpy as np
import pytest

from poreplex_amd import native as N

@pytest.fixture(scope='module')
def chim(golden):
    b = dict(np.load(golden('chimera.pxr.npz')))
    st = dict(np.load(golden('chimera.stages.npz')))
    with open(golden('chimera.results.json')) as fh:
        res = json.load(fh)
    return b, st, res


def test_a18_event_means_bit_exact(oracle, chim):
    b, st, res = chim
    recs = oracle.process_batch(b['arena'], b['offsets'], b['calib'])
    eo = st['ev_offsets']
    for i in range(len(eo) - 1):
        bc = json.loads(str(b['basecall'][i]))
        raw = b['arena'][b['offsets'][i]:b['offsets'][i + 1]]
        mean, scaled = oracle.guppy_event_means(raw, b['calib'][i], bc['first_sample_template'],
                                                bc['num_events'], recs[i]['scale'], recs[i]['shift'])
        assert np.array_equal(mean, st['ev_mean'][eo[i]:eo[i + 1]], equal_nan=True), i
        assert np.array_equal(scaled, st['ev_scaled'][eo[i]:eo[i + 1]], equal_nan=True), i


def test_a19_window_scan_candidates(oracle, chim):
    b, st, res = chim
    recs = oracle.process_batch(b['arena'], b['offsets'], b['calib'])
    eo = st['ev_offsets']
    n_with = 0
    for i in range(len(eo) - 1):
        bc = json.loads(str(b['basecall'][i]))
        payload_start = (int(recs[i]['seg_last'][3]) + 1) * 15
        iv, n = oracle.unsplit_scan(st['ev_scaled'][eo[i]:eo[i + 1]], bc['first_sample_template'],
                                    payload_start, float(b['calib'][i]['sampling_rate']))
        want = res['candidates'][i]
        assert n == len(want), (i, b['tag'][i])
        assert iv.tolist() == want, (i, b['tag'][i])
        n_with += n > 0
    assert n_with >= 6


def test_event_count_mismatch_is_an_error(oracle, chim):
    b, st, res = chim
    raw = b['arena'][b['offsets'][0]:b['offsets'][1]]
    with pytest.raises(Exception, match='does not match'):
        oracle.guppy_event_means(raw[:3000], b['calib'][0], 10, 400, 1.0, 0.0)
    # a read that ends inside the last block: NaN-padded mean, like the reference
    mean, _ = oracle.guppy_event_means(raw[:3007], b['calib'][0], 10, 200, 1.0, 0.0)
    assert np.isnan(mean[-1]) and not np.isnan(mean[:-1]).any()
