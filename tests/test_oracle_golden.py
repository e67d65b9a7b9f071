"""ORACLE vs the reference's own outputs (tests/golden, made by
tools/make_golden.py running the real reference under py3.9).  CPU only."""
import json
import os

import numpy as np
import pytest

from poreplex_amd import native as N

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def reads_of(bundle):
    o = bundle['offsets']
    return [bundle['arena'][o[i]:o[i + 1]] for i in range(len(o) - 1)]


def test_a1_raw_to_pa_bit_exact(oracle, bundle, stages):
    # fast5_file.py:122-131 via the real Fast5Reader.get_raw_data(end=64)
    for i, raw in enumerate(reads_of(bundle)):
        got = oracle.raw_to_pa(raw[:64], bundle['calib'][i])
        assert np.array_equal(got, stages['pa64'][i, :len(got)])


def test_a2_head_pool_bit_exact(oracle, bundle, stages):
    # signal_loader.py:212-231 via the real NanoporeRead.load_padded_signal_head
    n_ok = 0
    for i, raw in enumerate(reads_of(bundle)):
        got, st = oracle.head_pool(raw, bundle['calib'][i])
        if stages['head_ok'][i]:
            assert st == 0
            assert np.array_equal(got, stages['head'][i])
            n_ok += 1
        else:
            assert st == N.STATUS_CODE['scaler_signal_too_short']
    assert n_ok >= 20


def test_a4_destandardise_and_qc(oracle, unit):
    # signal_loader.py:89-109 via the real SignalLoader.fit_scalers
    n_fail = 0
    for pred, ok, ss in zip(unit['xfrm_pred'], unit['xfrm_ok'], unit['xfrm_ss']):
        got, st = oracle.scaler_transform(pred)
        assert (st == 0) == bool(ok)
        if ok:
            assert np.array_equal(got, ss)
        n_fail += not ok
    assert n_fail > 10   # the sweep straddles all four QC bounds


def test_a5_pool_scale_bit_exact(oracle, bundle, stages, ref_results):
    # signal_loader.py:233-264 via the real NanoporeRead.load_signal(pool=15)
    po = stages['pooled_offsets']
    n = 0
    for i, raw in enumerate(reads_of(bundle)):
        want = stages['pooled_arena'][po[i]:po[i + 1]]
        if not len(want):
            continue
        # the scaling the reference used = oracle LSTM output de-standardised
        head, _ = oracle.head_pool(raw, bundle['calib'][i])
        ss, st = oracle.scaler_transform(oracle.scaler_forward(head))
        assert st == 0
        got = oracle.pool_scale(raw, bundle['calib'][i], ss[0], ss[1])
        assert np.array_equal(got, want)
        n += 1
    assert n >= 18


def test_a8_segments_match_reference_groupby(oracle, stages):
    # signal_analyzer.py:346-364 real detect_segments (run-length summary and
    # scan-limit truncation) around the stubbed viterbi
    po = stages['pooled_offsets']
    n = 0
    for i in range(len(po) - 1):
        if not stages['has_seg'][i]:
            continue
        sig = stages['pooled_arena'][po[i]:po[i + 1]]
        scan = oracle.cfg.segmentation_scan_limit // oracle.cfg.stride
        _, path = oracle.viterbi(sig[:scan])
        first, last = oracle.segments(path)
        assert np.array_equal(first, stages['seg_first'][i])
        assert np.array_equal(last, stages['seg_last'][i])
        n += 1
    assert n >= 18


def test_a9_a10_a11_barcode_window(oracle, stages):
    # signal_analyzer.py:445-448 + barcoding.py:83-101 real push()
    po = stages['pooled_offsets']
    A = oracle.cfg.segmentation_model.adapter_state
    seen_pad = seen_trim = seen_skip = 0
    for i in range(len(po) - 1):
        if not stages['has_seg'][i] or stages['seg_first'][i, A] < 0:
            continue
        sig = stages['pooled_arena'][po[i]:po[i + 1]]
        a0, a1 = stages['seg_first'][i, A], stages['seg_last'][i, A]
        got, pushed = oracle.barcode_window(sig[a0:a1 + 1])
        assert pushed == bool(stages['pushed'][i])
        if pushed:
            assert np.array_equal(got, stages['window'][i])
            seen_pad += (a1 - a0 + 1) < 300
            seen_trim += (a1 - a0 + 1) > 300
        else:
            seen_skip += 1
    assert seen_pad and seen_trim and seen_skip


def test_a11_normalize_signal_unit(oracle, unit):
    # barcoding.py:77-81 real normalize_signal incl. even/odd, mad == 0
    off = np.concatenate([[0], np.cumsum(unit['ns_len'])])
    for k in range(len(unit['ns_len'])):
        x = unit['ns_in'][off[k]:off[k + 1]]
        assert np.array_equal(oracle.normalize_signal(x), unit['ns_out'][off[k]:off[k + 1]])
        got, pushed = oracle.barcode_window(x)
        assert pushed == bool(unit['push_flag'][k])
        if pushed:
            assert np.array_equal(got, unit['push_out'][k])


def test_a13_phred_and_threshold(oracle, unit):
    # barcoding.py:72-75 real lookup_calibrated_phred_score; :110 threshold
    assert oracle.cfg.score_threshold == float(unit['score_threshold'])
    for s, p, t in zip(unit['phred_score'], unit['phred'], unit['thr_pass']):
        assert oracle.phred(s) == p
        assert (float(np.float32(s)) >= oracle.cfg.score_threshold) == bool(t)


def test_a15_events_vs_reference_extension(oracle, unit):
    # src/csupport.c:70-124 (real CPython extension) on assorted windows
    off = np.concatenate([[0], np.cumsum(unit['ev_len'])])
    eo = np.concatenate([[0], np.cumsum(unit['ev_cnt'])])
    for k in range(len(unit['ev_len'])):
        ev = oracle.detect_events(unit['ev_in'][off[k]:off[k + 1]])
        assert len(ev) == unit['ev_cnt'][k]
        for f in ('start', 'length', 'mean', 'stdv'):
            assert np.array_equal(ev[f], unit['ev_' + f][eo[k]:eo[k + 1]], equal_nan=True), (k, f)


def test_a15_events_vs_compiled_reference_so(oracle):
    # oracle/_ref/libscrappie_ref.so = the reference's event_detection.c
    from oracle.pxo import reference_detect_events, REF_LIB_PATH
    if not os.path.isfile(REF_LIB_PATH):
        pytest.skip('oracle/_ref not built (reference tree absent)')
    rng = np.random.default_rng(5)
    for k in range(40):
        n = int(rng.integers(5, 6000))
        lv = rng.normal(95, 15, n // 6 + 2)
        x = np.repeat(lv, rng.geometric(1 / 8., len(lv)))[:n]
        x = (x + rng.normal(0, rng.uniform(0.2, 3), len(x))).astype(np.float32)
        a, b = oracle.detect_events(x), reference_detect_events(x)
        assert len(a) == len(b)
        for f in ('start', 'length', 'mean', 'stdv', 'pos', 'state'):
            assert np.array_equal(a[f], b[f], equal_nan=True), (k, f)


def test_a16_interval_dp(oracle, unit):
    # polya.py:156-187 real find_best_polya_interval
    off = np.concatenate([[0], np.cumsum(unit['dp_n'])])
    n_found = 0
    for k in range(len(unit['dp_n'])):
        got = oracle.best_polya_interval(unit['dp_isp'][off[k]:off[k + 1]],
                                         unit['dp_len'][off[k]:off[k + 1]])
        want = tuple(unit['dp_res'][k])
        assert (got or (-1, -1)) == want
        n_found += got is not None
    assert n_found > 20


def test_medfilt_matches_scipy(oracle):
    scipy_signal = pytest.importorskip('scipy.signal')
    rng = np.random.default_rng(3)
    for n in (1, 2, 6, 7, 8, 100, 1001):
        x = rng.normal(100, 10, n).astype(np.float32)
        for k in (5, 7):
            assert np.array_equal(oracle.medfilt(x, k), scipy_signal.medfilt(x, k))


def test_full_path_matches_reference_process_batch(oracle, bundle, ref_results):
    """a14-a17, a20: statuses, segments-derived fields and poly(A) dicts of the
    real process_batch (measure_polya on) vs the oracle's whole-read path."""
    res, spikes = oracle.process_batch(bundle['arena'], bundle['offsets'], bundle['calib'],
                                       stage_mask=N.STAGE_ALL_DEMUX | N.STAGE_POLYA,
                                       want_spikes=True)
    by_id = {r.get('read_id'): r for r in ref_results['results'] if 'read_id' in r}
    n_polya = n_spike_reads = 0
    for i, rid in enumerate(bundle['read_id']):
        want, got = by_id[str(rid)], res[i]
        numeric_status = want['status']
        if numeric_status in ('not_basecalled', 'sequence_too_short', 'okay'):
            numeric_status = 'okay'      # decided after the numeric stages
        assert N.STATUS_NAMES[got['status']] == numeric_status, (i, want['status'])
        assert ('barcode' in want) == bool(got['bc_called'])
        if 'polya' in want:
            p = want['polya']
            assert got['polya_called'], (i, bundle['tag'][i])
            assert got['polya_begin'] == p['begin'] and got['polya_end'] == p['end'], \
                (i, bundle['tag'][i])
            rate = float(bundle['calib'][i]['sampling_rate'])
            assert got['polya_dwell_samples'] / rate == p['dwell_time']
            assert got['polya_n_spikes'] == len(p['spikes'])
            for k, sp in enumerate(p['spikes']):
                assert np.array_equal(np.float32(sp), spikes[i, k]), (i, k)
            n_polya += 1
            n_spike_reads += len(p['spikes']) > 0
        else:
            assert not got['polya_called'], (i, bundle['tag'][i])
    assert n_polya >= 15 and n_spike_reads >= 8
