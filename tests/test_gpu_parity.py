"""HIP path (through the C ABI, libpxg.so) vs the oracle and the reference
goldens.  Everything here needs a real MI355X: run with `-m gpu`.

Bar (north_star): bit-exact for integer/byte/index work -- and, because the
LSTM arithmetic is canonical, bit-exact for the float outputs too;
the 1e-4 softmax tolerance is kept as the documented fallback bound.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_spikes_equal

from poreplex_amd import native as N
from poreplex_amd.synth import synth_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def reads_of(bundle):
    o = bundle['offsets']
    return [bundle['arena'][o[i]:o[i + 1]] for i in range(len(o) - 1)]


def assert_records_equal(got, want, fields=None, ctxmsg=''):
    for f in fields or got.dtype.names:
        a, b = got[f], want[f]
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.nonzero([not np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)])[0]
            raise AssertionError('{} field {!r}: {} reads differ, first {}: got {} want {}'.format(
                ctxmsg, f, len(bad), bad[0], a[bad[0]], b[bad[0]]))


# ---- a1 / a2 / a5 -----------------------------------------------------------
def test_raw_to_pa_bit_exact(ctx, oracle, bundle, stages):
    for i, raw in enumerate(reads_of(bundle)[:8]):
        got = ctx.raw_to_pa(raw, bundle['calib'][i])
        assert np.array_equal(got, oracle.raw_to_pa(raw, bundle['calib'][i]))
        assert np.array_equal(got[:64], stages['pa64'][i, :min(64, len(got))])


def test_head_pool_vs_reference_golden(ctx, bundle, stages):
    head, status = ctx.head_pool(bundle['arena'], bundle['offsets'], bundle['calib'])
    for i in range(len(status)):
        if stages['head_ok'][i]:
            assert status[i] == 0
            assert np.array_equal(head[i], stages['head'][i]), i
        else:
            assert status[i] == N.STATUS_CODE['scaler_signal_too_short']


def test_pool_scale_vs_oracle(ctx, oracle, bundle):
    ss = bundle['true_scale_shift']
    out, poff = ctx.pool_scale(bundle['arena'], bundle['offsets'], bundle['calib'], ss)
    for i, raw in enumerate(reads_of(bundle)):
        want = oracle.pool_scale(raw, bundle['calib'][i], ss[i, 0], ss[i, 1])
        assert np.array_equal(out[poff[i]:poff[i + 1]], want), i


def test_empty_and_tiny_reads(ctx, oracle, config):
    # ragged edge cases: empty read, shorter than one stride, exactly the gate
    rng = np.random.default_rng(1)
    lens = [0, 7, 15, 8999, 9000, 9014, 9015, 30000, 30001]
    sigs = [rng.integers(300, 700, n).astype(np.int16) for n in lens]
    arena, off = N.pack_reads(sigs)
    calib = np.zeros(len(lens), N.CALIB_DTYPE)
    calib['range'], calib['digitisation'], calib['offset'], calib['sampling_rate'] = 1200, 8192, 7, 3012
    got = ctx.process_batch(arena, off, calib)
    want = oracle.process_batch(arena, off, calib)
    assert_records_equal(got, want, ctxmsg='edge')
    assert (got['status'][:4] == N.STATUS_CODE['scaler_signal_too_short']).all()


# ---- a4: scaler network ------------------------------------------------------
@pytest.mark.parametrize('n', [1, 16, 17, 70])
def test_scaler_lstm_bit_exact(ctx, oracle, stages, n):
    heads = stages['scaler_in']
    rng = np.random.default_rng(n)
    rows = np.stack([heads[i % len(heads)] for i in range(n)]).copy()
    rows[n // 2:] += rng.normal(0, 2, rows[n // 2:].shape).astype(np.float32)
    got = ctx.scaler_lstm(rows)
    want = np.stack([oracle.scaler_forward(r) for r in rows])
    assert np.array_equal(got, want), np.abs(got - want).max()


@pytest.mark.parametrize('n', [3, 4, 5, 127, 1025, 2047, 2048, 2049])
def test_scaler_lstm_latency_form_equals_tile_form(ctx, oracle, stages, n, arith):
    """Up to 8 x #CU reads K2 runs its latency form (4-read tiles, digit planes in the idle MFMA columns,
    k_lstm_q8_lat.hip): same bits as the 16-read-tile kernel (PXG_K2_LAT_MAX=0) and as the oracle."""
    heads = stages['scaler_in']
    rng = np.random.default_rng(100 + n)
    rows = np.stack([heads[i % len(heads)] for i in range(n)])
    rows = rows + rng.normal(0, 1.5, (n, 1)).astype(np.float32)
    rows[::7, :rng.integers(1, 1900)] = 0.0            # left zero padding of short reads
    got = ctx.scaler_lstm(rows)
    os.environ['PXG_K2_LAT_MAX'] = '0'
    try:
        tiled = ctx.scaler_lstm(rows)
    finally:
        del os.environ['PXG_K2_LAT_MAX']
    assert np.array_equal(got, tiled), np.abs(got - tiled).max()
    pick = rng.choice(n, min(n, 12), replace=False)
    want = np.stack([oracle.scaler_forward(rows[i]) for i in pick])
    assert np.array_equal(got[pick], want)


@pytest.mark.gpu
def test_scaler_lstm_latency_form_on_twelve_waves():
    """The 12-wave variant of K2's latency form (two gate tiles per wave, PXG_K2_LAT_WAVES=12; read once per process,
    hence a child process): same bits as the default 8-wave form."""
    code = ('import sys, numpy as np; sys.path.insert(0, %r)\n'
            'from poreplex_amd import native as N; from poreplex_amd.config import default_config\n'
            'rng = np.random.default_rng(5); rows = rng.normal(0, 1, (37, 2000)).astype(np.float32); rows[::3, :700] = 0\n'
            'c = N.NativeContext(default_config(), 0); print(c.scaler_lstm(rows).tobytes().hex())' % ROOT)
    outs = []
    for waves in ('8', '12'):
        p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, PXG_K2_LAT_WAVES=waves, PXG_LSTM_ARITH='q8'),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        outs.append(p.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 37 * 2 * 4 * 2


@pytest.mark.parametrize('n', [8200, 9999, 13000])
def test_scaler_lstm_time_sliced_equals_static(ctx, oracle, stages, n):
    """Between 1 and 2 read tiles per resident workgroup K2 runs time-sliced (tile
    states handed over through HBM between 256-step blocks, k_scaler_lstm_q): same
    bits as the static kernel and as the oracle."""
    heads = stages['scaler_in']
    rng = np.random.default_rng(n)
    rows = np.stack([heads[i % len(heads)] for i in range(n)])
    rows = rows + rng.normal(0, 1.5, (n, 1)).astype(np.float32)     # every read differs
    got = ctx.scaler_lstm(rows)
    os.environ['PXG_NO_TIMESLICE'] = '1'
    try:
        static = ctx.scaler_lstm(rows)
    finally:
        del os.environ['PXG_NO_TIMESLICE']
    assert np.array_equal(got, static)
    pick = rng.choice(n, 24, replace=False)
    want = np.stack([oracle.scaler_forward(rows[i]) for i in pick])
    assert np.array_equal(got[pick], want)


def test_scaler_transform_vs_reference_golden(ctx, unit):
    ss, status = ctx.scaler_transform(unit['xfrm_pred'])
    ok = status == 0
    assert np.array_equal(ok, unit['xfrm_ok'].astype(bool))
    assert np.array_equal(ss[ok], unit['xfrm_ss'][ok])


# ---- a7 / a8: Viterbi ----------------------------------------------------------
def test_viterbi_segments_vs_oracle(ctx, oracle, stages):
    po = stages['pooled_offsets']
    scan = oracle.cfg.segmentation_scan_limit // oracle.cfg.stride
    sigs, idx = [], []
    for i in range(len(po) - 1):
        if stages['has_seg'][i]:
            sigs.append(stages['pooled_arena'][po[i]:po[i + 1]][:scan])
            idx.append(i)
    first, last, paths, logp = ctx.viterbi(sigs, want_path=True)
    for k, i in enumerate(idx):
        assert np.array_equal(first[k], stages['seg_first'][i]), i
        assert np.array_equal(last[k], stages['seg_last'][i]), i
        olp, opath = oracle.viterbi(sigs[k])
        assert np.array_equal(paths[k], opath)
        assert abs(logp[k] - olp) <= 1e-9 * abs(olp)


def test_viterbi_short_sequences(ctx, oracle):
    rng = np.random.default_rng(4)
    sigs = [rng.choice([71.5, 102, 112, 80, 109, 95], n).astype(np.float32) +
            rng.normal(0, 2, n).astype(np.float32) for n in (1, 2, 3, 5, 9, 63, 64, 65, 129)]
    first, last, paths, logp = ctx.viterbi(sigs, want_path=True)
    for k, s in enumerate(sigs):
        olp, opath = oracle.viterbi(s)
        assert np.array_equal(paths[k], opath), k
        assert abs(logp[k] - olp) <= 1e-9 * max(1.0, abs(olp))


# ---- a9-a12 ------------------------------------------------------------------------
def test_barcode_window_vs_reference_golden(ctx, unit):
    off = np.concatenate([[0], np.cumsum(unit['ns_len'])])
    sigs = [unit['ns_in'][off[k]:off[k + 1]] for k in range(len(unit['ns_len']))]
    out, pushed = ctx.barcode_window(sigs)
    assert np.array_equal(pushed.astype(bool), unit['push_flag'].astype(bool))
    for k in range(len(sigs)):
        if pushed[k]:
            assert np.array_equal(out[k], unit['push_out'][k]), k


def test_resident_batch_windows_vs_reference_golden(ctx, bundle, stages):
    """pxg_batch_download_windows: the classifier input of every pushed read of a resident batch
    (what a training set is made of) equals the window the REAL reference queued for that read
    (BarcodeDemultiplexer.push, tests/golden/batch0.stages.npz; the scaling parameters of that run
    came from the canonical scaler arithmetic, which is the GPU's)."""
    ctx.upload(bundle['arena'], bundle['offsets'], bundle['calib'])
    ctx.run(N.STAGE_ALL_DEMUX)
    rec = ctx.download()
    win = ctx.download_windows(rec)
    pushed = stages['pushed'].astype(bool)
    assert np.array_equal(rec['bc_pushed'].astype(bool), pushed) and pushed.sum() >= 20
    assert np.array_equal(win[pushed], stages['window'][pushed])
    assert not win[~pushed].any()
    ctx.upload(bundle['arena'][:bundle['offsets'][2]], bundle['offsets'][:3], bundle['calib'][:2])
    ctx.run(N.STAGE_SCALER)
    with pytest.raises(N.PxgError):                              # no barcode stage in the last run
        ctx.download_windows()


@pytest.mark.parametrize('n', [1, 15, 33, 64])
def test_demux_lstm_bit_exact(ctx, oracle, stages, n):
    wins = stages['demux_in']
    rng = np.random.default_rng(n)
    rows = np.stack([wins[i % len(wins)] for i in range(n)]).copy()
    rows[n // 2:] = np.roll(rows[n // 2:], 3, axis=1)
    got = ctx.demux_lstm(rows)
    want = np.stack([oracle.demux_forward(r) for r in rows])
    assert np.abs(got - want).max() <= 1e-4          # north_star bound
    assert np.array_equal(got, want), np.abs(got - want).max()
    assert np.array_equal(got.argmax(1), want.argmax(1))


@pytest.mark.parametrize('n', [2, 4, 5, 130, 1025, 2047, 2048, 2049])
def test_demux_lstm_latency_form_equals_tile_form(ctx, oracle, stages, n, arith):
    """Up to 8 x #CU reads K5a / K5b run their latency forms (4-read tiles, k_lstm_q8_lat.hip): same bits as
    the 16-read-tile kernels (PXG_K5_LAT_MAX=0) and as the oracle."""
    wins = stages['demux_in']
    rng = np.random.default_rng(200 + n)
    rows = np.stack([wins[i % len(wins)] for i in range(n)])
    rows = rows + rng.normal(0, 0.05, (n, 1)).astype(np.float32)
    rows[::5] = np.roll(rows[::5], 7, axis=1)
    got = ctx.demux_lstm(rows)
    os.environ['PXG_K5_LAT_MAX'] = '0'
    try:
        tiled = ctx.demux_lstm(rows)
    finally:
        del os.environ['PXG_K5_LAT_MAX']
    assert np.array_equal(got, tiled), np.abs(got - tiled).max()
    pick = rng.choice(n, min(n, 12), replace=False)
    want = np.stack([oracle.demux_forward(rows[i]) for i in pick])
    assert np.array_equal(got[pick], want)


@pytest.mark.parametrize('n', [8200, 9898, 13000, 20001])
def test_demux_lstm_time_sliced_equals_static(ctx, oracle, stages, n):
    """With more read tiles than resident workgroups K5a / K5b run time-sliced
    (k_demux_bidir_q / k_demux_top_q: tile states handed over through HBM between step
    blocks whose number is chosen on the device): same bits as the static kernels and as
    the oracle."""
    wins = stages['demux_in']
    rng = np.random.default_rng(n)
    rows = np.stack([wins[i % len(wins)] for i in range(n)])
    rows = rows + rng.normal(0, 0.05, (n, 1)).astype(np.float32)     # every window differs
    rows[::7] = np.roll(rows[::7], 5, axis=1)
    got = ctx.demux_lstm(rows)
    os.environ['PXG_NO_DEMUX_TIMESLICE'] = '1'
    try:
        static = ctx.demux_lstm(rows)
    finally:
        del os.environ['PXG_NO_DEMUX_TIMESLICE']
    assert np.array_equal(got, static)
    pick = rng.choice(n, 24, replace=False)
    want = np.stack([oracle.demux_forward(rows[i]) for i in pick])
    assert np.array_equal(got[pick], want)


# ---- a14-a17: events + poly(A) ----------------------------------------------------
def test_detect_events_vs_reference_extension(ctx, oracle, unit):
    # src/csupport.c:70-124 outputs captured from the real CPython extension
    off = np.concatenate([[0], np.cumsum(unit['ev_len'])])
    eo = np.concatenate([[0], np.cumsum(unit['ev_cnt'])])
    sigs = [unit['ev_in'][off[k]:off[k + 1]] for k in range(len(unit['ev_len']))]
    evs, cnt = ctx.detect_events(sigs, max_events=int(unit['ev_cnt'].max()) + 4)
    assert np.array_equal(cnt, unit['ev_cnt'])
    for k, ev in enumerate(evs):
        for f in ('start', 'length', 'mean', 'stdv'):
            assert np.array_equal(ev[f], unit['ev_' + f][eo[k]:eo[k + 1]], equal_nan=True), (k, f)
        assert (ev['pos'] == -1).all() and (ev['state'] == -1).all()


def test_detect_events_many_windows_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(8)
    sigs = []
    for k in range(150):                      # > 2 waves of lanes, ragged lengths
        n = int(rng.integers(1, 4000))
        lv = rng.normal(95, 15, n // 6 + 2)
        x = np.repeat(lv, rng.geometric(1 / 8., len(lv)))[:n]
        sigs.append((x + rng.normal(0, rng.uniform(0.2, 3), len(x))).astype(np.float32))
    evs, cnt = ctx.detect_events(sigs, max_events=1200)
    for k, s in enumerate(sigs):
        want = oracle.detect_events(s)
        assert cnt[k] == len(want), k
        for f in ('start', 'length', 'mean', 'stdv'):
            assert np.array_equal(evs[k][f], want[f], equal_nan=True), (k, f)


def test_polya_golden_bundle_vs_reference(ctx, oracle, bundle, ref_results):
    """a14-a17 on the GPU vs the REAL polya.py results (begin/end/dwell/spikes)."""
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    ctx.upload(bundle['arena'], bundle['offsets'], bundle['calib'])
    ctx.run(mask)
    got, spikes = ctx.download(), ctx.download_spikes()
    want, wspk = oracle.process_batch(bundle['arena'], bundle['offsets'], bundle['calib'],
                                      stage_mask=mask, want_spikes=True)
    assert_records_equal(got, want, ctxmsg='polya golden')
    assert_spikes_equal(spikes, want, wspk)
    by_id = {r.get('read_id'): r for r in ref_results['results'] if 'read_id' in r}
    n = 0
    for i, rid in enumerate(bundle['read_id']):
        p = by_id[str(rid)].get('polya')
        assert bool(got['polya_called'][i]) == (p is not None), (i, bundle['tag'][i])
        if p is not None:
            assert (got['polya_begin'][i], got['polya_end'][i]) == (p['begin'], p['end'])
            assert got['polya_dwell_samples'][i] / float(bundle['calib'][i]['sampling_rate']) == p['dwell_time']
            assert got['polya_n_spikes'][i] == len(p['spikes'])
            rows = spikes[0][spikes[1][i]:spikes[1][i + 1]]
            assert len(rows) == len(p['spikes'])
            for k, sp in enumerate(p['spikes']):
                assert np.array_equal(np.float32(sp), rows[k]), (i, k)
            n += 1
    assert n >= 15


@pytest.mark.parametrize('seed,noise,dwell', [(927, 1.5, 9.0), (928, 0.6, 25.0), (929, 2.5, 5.0)])
def test_polya_synthetic_vs_oracle(ctx, oracle, seed, noise, dwell):
    b = synth_batch(200, seed=seed, samples_per_read=36000, jitter=0.3, sample_noise=noise,
                    mean_dwell=dwell)
    mask = N.STAGE_SEGMENT | N.STAGE_POLYA
    ctx.upload(b['arena'], b['offsets'], b['calib'], b['scale_shift'])
    ctx.run(mask)
    got, spikes = ctx.download(), ctx.download_spikes()
    want, wspk = oracle.process_batch(b['arena'], b['offsets'], b['calib'], b['scale_shift'],
                                      stage_mask=mask, want_spikes=True)
    assert_records_equal(got, want, ctxmsg='polya synthetic')
    assert_spikes_equal(spikes, want, wspk)
    assert got['polya_called'].sum() > 60


def test_polya_first_pass_in_work_order_equals_batch_order(ctx, oracle, monkeypatch):
    """K6's first pass takes the reads by poly(A) length class, longest first (k_polya_order_*, the `subset` path of
    k_polya); PXG_POLYA_INPUT_ORDER=1 takes them in batch order as until round 4.  Same records and spike rows either
    way, equal to the oracle's -- ragged reads, short reads, reads whose tail is not called."""
    mask = N.STAGE_SEGMENT | N.STAGE_POLYA
    b = synth_batch(333, seed=951, samples_per_read=30000, jitter=0.6, sample_noise=1.4, mean_dwell=9.0, short_fraction=0.08)
    want, wspk = oracle.process_batch(b['arena'], b['offsets'], b['calib'], b['scale_shift'], stage_mask=mask, want_spikes=True)
    for batch_order in (False, True):
        if batch_order:
            monkeypatch.setenv('PXG_POLYA_INPUT_ORDER', '1')
        else:
            monkeypatch.delenv('PXG_POLYA_INPUT_ORDER', raising=False)
        ctx.upload(b['arena'], b['offsets'], b['calib'], b['scale_shift'])
        ctx.run(mask)
        got, spikes = ctx.download(), ctx.download_spikes()
        assert_records_equal(got, want, ctxmsg='polya order %s' % batch_order)
        assert_spikes_equal(spikes, want, wspk)
    assert got['polya_called'].sum() > 100 and (got['polya_called'] == 0).sum() >= 5


# ---- whole path ------------------------------------------------------------------
def test_process_batch_golden_bundle(ctx, oracle, bundle):
    for inject in (None, bundle['true_scale_shift']):
        got = ctx.process_batch(bundle['arena'], bundle['offsets'], bundle['calib'], inject)
        want = oracle.process_batch(bundle['arena'], bundle['offsets'], bundle['calib'], inject)
        assert_records_equal(got, want, ctxmsg='golden inject=%s' % (inject is not None))
    assert set(N.STATUS_NAMES[s] for s in got['status']) >= {'okay', 'scaler_signal_too_short'}


def test_process_batch_reference_statuses(ctx, bundle, ref_results):
    """Statuses / barcode calls against what the REAL reference process_batch
    reported for the same reads (tests/golden/batch0.results.json)."""
    got = ctx.process_batch(bundle['arena'], bundle['offsets'], bundle['calib'])
    by_id = {r.get('read_id'): r for r in ref_results['results'] if 'read_id' in r}
    for i, rid in enumerate(bundle['read_id']):
        want = by_id[str(rid)]
        st = want['status']
        if st in ('not_basecalled', 'sequence_too_short', 'okay'):
            st = 'okay'
        assert N.STATUS_NAMES[got['status'][i]] == st, (i, want['status'])
        assert ('barcode' in want) == bool(got['bc_called'][i])


def test_process_batch_synthetic_300(ctx, oracle):
    b = synth_batch(300, seed=924, samples_per_read=40000, jitter=0.4, short_fraction=0.03)
    got = ctx.process_batch(b['arena'], b['offsets'], b['calib'])
    want = oracle.process_batch(b['arena'], b['offsets'], b['calib'])
    assert_records_equal(got, want, ctxmsg='synthetic')
    assert (got['status'] == 0).sum() > 250 and got['bc_pushed'].sum() > 200


def test_stage_masks(ctx, oracle):
    b = synth_batch(40, seed=925, samples_per_read=30000)
    for mask, inj in ((N.STAGE_SCALER, None), (N.STAGE_SEGMENT, b['scale_shift']),
                      (N.STAGE_SCALER | N.STAGE_SEGMENT, None),
                      (N.STAGE_SEGMENT | N.STAGE_BARCODE, b['scale_shift'])):
        got = ctx.process_batch(b['arena'], b['offsets'], b['calib'], inj, mask)
        want = oracle.process_batch(b['arena'], b['offsets'], b['calib'], inj, mask)
        assert_records_equal(got, want, ctxmsg='mask %d' % mask)


# ---- BASELINE-size properties (no oracle at this size) ------------------------------
def test_full_size_determinism_and_order_invariance(ctx):
    b = synth_batch(4000, seed=926, samples_per_read=60000, short_fraction=0.01)
    r1 = ctx.process_batch(b['arena'], b['offsets'], b['calib'])
    r2 = ctx.process_batch(b['arena'], b['offsets'], b['calib'])
    assert_records_equal(r1, r2, ctxmsg='rerun')
    # per-read results must not depend on batch composition / tile placement
    perm = np.random.default_rng(0).permutation(len(r1))[:1500]
    sigs = [b['arena'][b['offsets'][i]:b['offsets'][i + 1]] for i in perm]
    arena, off = N.pack_reads(sigs)
    r3 = ctx.process_batch(arena, off, b['calib'][perm])
    assert_records_equal(r3, r1[perm], ctxmsg='permuted')
    ok = r1['status'] == 0
    assert ok.mean() > 0.9
    A = 3
    assert (r1['seg_first'][ok, A] >= 0).all()
    assert (r1['seg_last'][ok] < r1['n_pooled'][ok, None]).all()
    p = r1['probs'][r1['bc_pushed'] == 1, :5]
    assert np.abs(p.sum(1) - 1).max() < 1e-5


def test_run_shaped_batch_length_order_and_pad_prefix_skip(ctx, oracle, config, monkeypatch):
    """Lengths as a sequencing run has them (log-normal, 15 % of the reads shorter than the scaler's 30 000
    samples, a tail up to 10^6): blocks of K3 take the reads in length order and K2 (q8) starts a tile behind
    the zero-pad prefix its 16 reads share, from the recorded zero-input trajectory.  Both are exact: the
    records equal the oracle's, and those of a context with both savings switched off."""
    b = synth_batch(700, seed=931, length_dist='lognormal', short_fraction=0.02)
    n_means = np.minimum(np.diff(b['offsets']), 30000) // 15
    assert ((n_means < 2000) & (n_means >= 600)).sum() > 60      # padded heads are in play
    got = ctx.process_batch(b['arena'], b['offsets'], b['calib'])
    want = oracle.process_batch(b['arena'], b['offsets'], b['calib'])
    assert_records_equal(got, want, ctxmsg='run-shaped batch vs oracle')
    monkeypatch.setenv('PXG_NO_LENGTH_ORDER', '1')
    monkeypatch.setenv('PXG_NO_PREFIX_SKIP', '1')
    plain = N.NativeContext(config, device_id=0)
    try:
        assert_records_equal(plain.process_batch(b['arena'], b['offsets'], b['calib']), got, ctxmsg='savings off')
    finally:
        plain.close()
    # every pad length of a tile's worth of reads, incl. heads of exactly 600 and 1 999 means: all 16 reads of a
    # tile padded alike (the skip is the whole prefix) and mixed with unpadded ones (no skip)
    lens = np.concatenate([np.repeat(np.arange(9000, 30015, 1500), 16), np.repeat([9000, 29985, 29999, 30000, 45000], 8)])
    rng = np.random.default_rng(5)
    pieces = [synth_batch(1, seed=4000 + i, samples_per_read=int(L), jitter=0.0) for i, L in enumerate(lens)]
    arena, off = N.pack_reads([p['arena'] for p in pieces])
    calib = np.concatenate([p['calib'] for p in pieces])
    for perm in (np.arange(len(lens)), rng.permutation(len(lens))):
        sig = [arena[off[i]:off[i + 1]] for i in perm]
        a2, o2 = N.pack_reads(sig)
        g = ctx.process_batch(a2, o2, calib[perm], None, N.STAGE_SCALER)
        w = oracle.process_batch(a2, o2, calib[perm], None, N.STAGE_SCALER)
        assert_records_equal(g, w, ctxmsg='pad ladder')


# ---- a18 / a19 ---------------------------------------------------------------
def _chimera():
    import json
    from conftest import G
    b = dict(np.load(G('chimera.pxr.npz')))
    st = dict(np.load(G('chimera.stages.npz')))
    with open(G('chimera.results.json')) as fh:
        res = json.load(fh)
    bc = [json.loads(str(x)) for x in b['basecall']]
    return b, st, res, bc


def test_guppy_event_means_vs_reference_golden(ctx):
    """a18: block means of the medfilt(5) pA signal and their scaled values,
    bit-exact against the REAL reference's event table."""
    b, st, res, bc = _chimera()
    ctx.upload(b['arena'], b['offsets'], b['calib'])
    ctx.run(N.STAGE_SCALER | N.STAGE_SEGMENT)
    recs = ctx.download()
    ss = np.stack([recs['scale'], recs['shift']], axis=1).astype(np.float32)
    first = [c['first_sample_template'] for c in bc]
    nb = [len(c['move']) for c in bc]
    mean, scaled, eo = ctx.guppy_event_means(b['arena'], b['offsets'], b['calib'], ss, first, nb)
    assert np.array_equal(eo, st['ev_offsets'])
    assert np.array_equal(mean, st['ev_mean'], equal_nan=True)
    assert np.array_equal(scaled, st['ev_scaled'], equal_nan=True)


def test_guppy_event_means_ragged_tail_vs_oracle(ctx, oracle):
    """Reads that end inside the last block (NaN-padded mean), first_sample 0 and
    a block count of 0."""
    b, st, res, bc = _chimera()
    raw = b['arena'][b['offsets'][0]:b['offsets'][1]]
    cases = [(raw[:3007], 10, 200), (raw[:3000], 0, 200), (raw[:45], 0, 3), (raw[:100], 7, 0),
             (raw[:64], 3, 5)]
    arena, off = N.pack_reads([c[0] for c in cases])
    calib = np.array([tuple(b['calib'][0])] * len(cases), dtype=N.CALIB_DTYPE)
    ss = np.float32([[0.97, -3.5]] * len(cases))
    mean, scaled, eo = ctx.guppy_event_means(arena, off, calib, ss, [c[1] for c in cases],
                                             [c[2] for c in cases])
    for i, (r, f, n) in enumerate(cases):
        wm, wsc = oracle.guppy_event_means(r, calib[i], f, n, ss[i, 0], ss[i, 1])
        assert np.array_equal(mean[eo[i]:eo[i + 1]], wm, equal_nan=True), i
        assert np.array_equal(scaled[eo[i]:eo[i + 1]], wsc, equal_nan=True), i
    assert np.isnan(mean[eo[0]:eo[1]][-1])


def test_batch_event_table_and_pooled_signal_vs_oracle(ctx, oracle):
    """The two downloads of the dump options on a resident batch: mean / stdv / scaled mean of
    every Guppy block (ragged tails, a NaN-padded last block, reads left out, block strides 15
    and 10) and one pooled + scaled stretch per read -- bit-identical to the oracle, whose
    event columns equal the REAL reference's dump (tests/golden/dumps0.npz, CPU suite)."""
    from poreplex_amd.synth import synth_batch
    sb = synth_batch(24, seed=77, samples_per_read=20000, jitter=0.6, short_fraction=0.0)
    n = 24
    reads = [sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in range(n)]
    ctx.upload(sb['arena'], sb['offsets'], sb['calib'])
    ctx.run(N.STAGE_SCALER | N.STAGE_SEGMENT)
    recs = ctx.download()
    rng = np.random.default_rng(9)
    for stride in (15, 10):
        first = rng.integers(0, 50, n)
        nb = np.array([-(-(len(r) - f) // stride) for r, f in zip(reads, first)])      # the last block may be cut
        nb[2], nb[7] = 0, 1
        nb[11] = max(nb[11] - 300, 1)
        mean, stdv, scaled, eo = ctx.event_table(first, nb, stride)
        assert eo[-1] == nb.sum()
        for i in range(n):
            if not nb[i]:
                continue
            wm, wsd, wsc = oracle.guppy_event_table(reads[i], sb['calib'][i], first[i], nb[i], recs[i]['scale'],
                                                    recs[i]['shift'], stride)
            assert np.array_equal(mean[eo[i]:eo[i + 1]], wm, equal_nan=True), (stride, i)
            assert np.array_equal(stdv[eo[i]:eo[i + 1]], wsd, equal_nan=True), (stride, i)
            assert np.array_equal(scaled[eo[i]:eo[i + 1]], wsc, equal_nan=True), (stride, i)
        assert np.isnan(stdv).sum() == np.isnan(mean).sum() > 0 and (stdv[~np.isnan(stdv)] >= 0).all()
    pooled = np.array([len(r) // 15 for r in reads])
    p0 = rng.integers(0, 200, n)
    cnt = np.minimum(rng.integers(0, 900, n), pooled - p0)
    cnt[4], p0[9], cnt[9] = 0, 0, pooled[9]                      # left out / the whole read
    values, off = ctx.pooled_signal(p0, cnt)
    for i in range(n):
        want = oracle.pool_scale(reads[i], sb['calib'][i], recs[i]['scale'], recs[i]['shift'])[p0[i]:p0[i] + cnt[i]]
        assert np.array_equal(values[off[i]:off[i + 1]], want), i
    with pytest.raises(N.PxgError):                             # a stretch that leaves its read
        ctx.pooled_signal(p0, np.where(np.arange(n) == 3, pooled[3] - p0[3] + 1, cnt))


def test_unsplit_scan_vs_reference_candidates(ctx):
    """a19: the candidate in-read adapters of every window equal the list the REAL
    reference handed to union_intervals."""
    b, st, res, bc = _chimera()
    ctx.upload(b['arena'], b['offsets'], b['calib'])
    ctx.run(N.STAGE_SCALER | N.STAGE_SEGMENT)
    iv, cnt, start = ctx.unsplit_scan([c['first_sample_template'] for c in bc],
                                      [len(c['move']) for c in bc])
    for i, want in enumerate(res['candidates']):
        assert cnt[i] == len(want), (i, str(b['tag'][i]))
        assert iv[start[i]:start[i + 1]].tolist() == want, (i, str(b['tag'][i]))
    assert (cnt > 0).sum() >= 6 and len(iv) == cnt.sum()


def test_unsplit_scan_synthetic_vs_oracle(ctx, oracle, config):
    """Random chimeras / plain reads of very different lengths in one wave: ragged
    window counts, reads without events, reads without an adapter."""
    from poreplex_amd.synth import synth_batch
    sb = synth_batch(40, seed=4141, samples_per_read=22000, jitter=0.5, fixed_calib=True,
                     scale_sigma=0.0, shift_sigma=0.0, short_fraction=0.1)
    parts = [sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in range(40)]
    reads = [np.concatenate([parts[i], parts[i + 1]]) for i in range(0, 16, 2)]
    reads += [np.concatenate([parts[16], parts[17], parts[18]])] + parts[19:40]
    rng = np.random.default_rng(5)
    first = rng.integers(0, 60, len(reads))
    nb = np.array([(len(r) - f) // 15 for r, f in zip(reads, first)])
    nb[3] = 0                                   # not basecalled
    nb[5] = max(nb[5] - 700, 1)                 # table ends long before the read does
    arena, off = N.pack_reads(reads)
    calib = np.array([tuple(sb['calib'][0])] * len(reads), dtype=N.CALIB_DTYPE)
    ctx.upload(arena, off, calib)
    ctx.run(N.STAGE_SCALER | N.STAGE_SEGMENT)
    recs = ctx.download()
    iv, cnt, start = ctx.unsplit_scan(first, nb)
    a = 3
    n_cand = 0
    for i, r in enumerate(reads):
        if nb[i] <= 0 or recs[i]['status'] != 0 or recs[i]['seg_first'][a] < 0:
            assert cnt[i] == 0
            continue
        _, scaled = oracle.guppy_event_means(r, calib[i], first[i], nb[i], recs[i]['scale'],
                                             recs[i]['shift'])
        want, c = oracle.unsplit_scan(scaled, first[i], (int(recs[i]['seg_last'][a]) + 1) * 15,
                                      3012.0)
        assert cnt[i] == c, i
        assert iv[start[i]:start[i + 1]].tolist() == want.tolist(), i
        n_cand += c
    assert n_cand >= 8
    # a read with an impossible event frame gets ITS OWN error code; the others are unchanged
    bad_first = first.copy()
    bad_first[7] = -5
    iv2, cnt2, start2 = ctx.unsplit_scan(bad_first, nb)
    assert cnt2[7] == N.UNSPLIT_E_GEOMETRY
    keep = np.arange(len(reads)) != 7
    assert np.array_equal(cnt2[keep], cnt[keep])
    # the 8-lanes-per-window kernel (models of more than six states take it; PXG_UNSPLIT_8_LANES sends every
    # model there) gives the same lists, and a batch without a single event is scanned without touching the
    # (empty) table
    os.environ['PXG_UNSPLIT_8_LANES'] = '1'
    try:
        iv3, cnt3, start3 = ctx.unsplit_scan(first, nb)
    finally:
        del os.environ['PXG_UNSPLIT_8_LANES']
    assert np.array_equal(cnt3, cnt) and np.array_equal(iv3, iv) and np.array_equal(start3, start)
    # the one-window-per-lane kernel is specialised for the shipped model's 16 edges (round 6); PXG_UNSPLIT_DENSE sends the
    # model through the dense form any other topology takes: same lists
    os.environ['PXG_UNSPLIT_DENSE'] = '1'
    try:
        iv5, cnt5, start5 = ctx.unsplit_scan(first, nb)
    finally:
        del os.environ['PXG_UNSPLIT_DENSE']
    assert np.array_equal(cnt5, cnt) and np.array_equal(iv5, iv) and np.array_equal(start5, start)
    _, cnt4, _ = ctx.unsplit_scan(first, np.zeros_like(nb))
    assert not cnt4.any()
    for i in np.nonzero(keep)[0]:
        assert iv2[start2[i]:start2[i + 1]].tolist() == iv[start[i]:start[i + 1]].tolist(), i


def test_label_all_gather_over_rccl_single_rank():
    """The N>1 bookkeeping collectives (RCCL all-gather of label records, all-reduce
    of the count table) driven on this one GPU: world 1, device tensors, nccl --
    in a fresh process that brings torch/RCCL up BEFORE libpxg.so, the order
    bench.py uses for N>1 (both then share one HIP runtime)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29617', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'tests', 'rccl_single_rank.py')],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'RCCL-SINGLE-RANK-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_rccl_with_two_ranks_on_one_device():
    """RCCL has only ever been driven with ONE rank here (no multi-GPU node).  Two ranks on the one
    device there is: when the runtime allows it, the label all-gather / count all-reduce of the N > 1 path run
    through a real 2-rank RCCL communicator; when it refuses (duplicate devices), the test records the
    runtime's own message and passes -- it cannot be done on this hardware, and the line says so."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29619', RANK=str(rank), WORLD_SIZE='2',
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.pop('PXG_LSTM_ARITH', None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'rccl_two_ranks_one_gpu.py')], env=env,
                                      cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate())
    text = ' | '.join(o[0].strip()[-500:] for o in outs)
    print('two RCCL ranks on one device:', text)
    ok = all('RCCL-TWO-RANKS-OK' in o[0] for o in outs)
    refused = any('RCCL-TWO-RANKS-REFUSED' in o[0] for o in outs) or any(p.returncode not in (0, None) for p in procs)
    assert ok or refused, text + ' || ' + ' | '.join(o[1][-1500:] for o in outs)
    if not ok:
        pytest.skip('RCCL does not form a 2-rank communicator on ONE device here: ' + text[:300])


def test_custom_left_to_right_hmm_generic_kernel(config, bundle):
    """A 7-state model with skip edges of span 3 and 4, a 3-component mixture and two
    start states: takes the generic K3 template (not the shipped-model fast path);
    whole batches and the pooled-signal hook against an oracle built from the same
    config."""
    import copy
    from oracle.pxo import Oracle
    cfg = copy.deepcopy(config)
    m = cfg['segmentation_model']
    names = [st['name'] for st in m]
    m.insert(3, {'name': 'stall', 'emission': [[90.0, 6.0, 0.5], [120.0, 9.0, 0.3], [60.0, 4.0, 0.2]],
                 'transition': [['stall', 0.9], ['adapter', 0.1]]})
    m[0]['start_prob'], m[1]['start_prob'] = 0.7, 0.3
    m[0]['transition'] = [['pre-leader', 0.9], ['leader-low', 0.05], ['stall', 0.03], ['adapter', 0.02]]
    m[2]['transition'] = [['leader-high', 0.98], ['stall', 0.01], ['adapter', 0.01]]
    assert [st['name'] for st in m] == names[:3] + ['stall'] + names[3:]
    orc = Oracle(cfg)
    c = N.NativeContext(cfg, device_id=0)
    try:
        want = orc.process_batch(bundle['arena'], bundle['offsets'], bundle['calib'])
        got = c.process_batch(bundle['arena'], bundle['offsets'], bundle['calib'])
        assert_records_equal(got, want)
        assert (got['status'] == 0).sum() >= 10
        sb = synth_batch(64, seed=515, samples_per_read=12000, jitter=0.4)
        want = orc.process_batch(sb['arena'], sb['offsets'], sb['calib'])
        got = c.process_batch(sb['arena'], sb['offsets'], sb['calib'])
        assert_records_equal(got, want)
        rng = np.random.default_rng(8)
        sigs = [rng.choice([71.5, 102, 112, 90, 80, 109, 95], n).astype(np.float32) +
                rng.normal(0, 3, n).astype(np.float32) for n in (1, 7, 40, 333, 2000)]
        first, last, paths, logp = c.viterbi(sigs, want_path=True)
        for k, sg in enumerate(sigs):
            olp, opath = orc.viterbi(sg)
            assert np.array_equal(paths[k], opath), k
    finally:
        c.close()


def test_mixture_emission_log_sum_exp_vs_libm(config):
    """The table-driven log(exp(d) + 1) of the kernels (pxg_common.h pxg_log1pexp) against the
    oracle's libm formula: a ONE-state HMM whose emission is a mixture makes the Viterbi
    log-probability of a 1-step sequence exactly log(1) + emission(x).  Sweeps the whole range
    of component gaps (identical components ... one component 40 log-units below the other),
    two- and three-component mixtures."""
    import copy
    from oracle.pxo import Oracle
    xs = np.concatenate([np.linspace(20, 180, 4001), np.random.default_rng(3).uniform(40, 140, 4000)]
                        ).astype(np.float32)
    for emission in ([[80.49, 7.41, 0.848], [65.30, 3.31, 0.152]],
                     [[81.9, 7.76, 0.49], [109.73, 12.73, 0.51]],
                     [[100.0, 2.0, 0.5], [100.0, 2.0, 0.5]],
                     [[90.0, 6.0, 0.5], [120.0, 9.0, 0.3], [60.0, 4.0, 0.2]]):
        cfg = copy.deepcopy(config)
        cfg['segmentation_model'] = [{'name': 'adapter', 'emission': emission, 'start_prob': 1.0,
                                      'transition': [['adapter', 1.0]]}]
        orc = Oracle(cfg)
        c = N.NativeContext(cfg, device_id=0)
        try:
            _, _, _, logp = c.viterbi([np.array([x], np.float32) for x in xs])
        finally:
            c.close()
        want = np.array([orc.emission(0, float(x)) for x in xs])
        err = np.abs(logp - want)
        # one ulp of the rounded sum exp(d) + 1 (2.2e-16) plus one ulp of the result
        assert err.max() <= 2.3e-16 + 2 * np.spacing(np.abs(want).max()), (emission, err.max())


def test_staged_double_buffered_batches(ctx, oracle):
    """pxg_batch_stage / pxg_batch_swap: batch i+1 is copied on the copy stream while
    batch i computes; every batch gets the records of a plain upload -> run."""
    batches = [synth_batch(n, seed=600 + n, samples_per_read=9000, jitter=0.3)
               for n in (96, 33, 150, 1, 64)]
    want = [oracle.process_batch(b['arena'], b['offsets'], b['calib']) for b in batches]
    batches[2]['arena'] = N.pinnable(batches[2]['arena'])       # (mapped pages of its own: never page-lock heap memory)
    pinned = ctx.pin(batches[2]['arena'])           # one of them page-locked
    try:
        ctx.upload(batches[0]['arena'], batches[0]['offsets'], batches[0]['calib'])
        for i in range(len(batches)):
            ctx.run(N.STAGE_ALL_DEMUX)
            if i + 1 < len(batches):
                nb = batches[i + 1]
                ctx.stage(nb['arena'], nb['offsets'], nb['calib'])
            got = ctx.download()
            assert_records_equal(got, want[i], ctxmsg='batch %d' % i)
            if i + 1 < len(batches):
                ctx.swap()
    finally:
        ctx.unpin(pinned)
    with pytest.raises(N.PxgError):
        ctx.swap()                                  # nothing staged


def test_structured_reads_around_every_length_threshold(ctx, oracle):
    """Real-looking reads cut at / around every length rule of the path: the scaler gate
    (9 000), the head window (30 000), the segmentation scan limit (100 000 samples =
    6 666 pooled), reads that end inside the adapter or the poly(A) tail, adapters shorter
    than minimum_dna_length (260 pooled) and longer than maximum_dna_length (3 000), all
    stages incl. poly(A), and the chimera scan on top."""
    sb = synth_batch(24, seed=8642, samples_per_read=130000, jitter=0.02)
    o, tr = sb['offsets'], sb['truth']
    reads, calib = [], []
    cuts = [8999, 9000, 9014, 29999, 30000, 30001, 99999, 100000, 100001, 100014, 100015, 129990]
    for i, c in enumerate(cuts):
        reads.append(sb['arena'][o[i]:o[i] + c])
        calib.append(sb['calib'][i])
    for i in range(12, 18):          # cut inside the adapter / the poly(A) tail / right after it
        a_beg, a_end, p_end = int(tr[i, 3, 0]), int(tr[i, 3, 1]), int(tr[i, 4, 1])
        for c in (a_beg + 260 * 15 - 15, a_beg + 260 * 15 + 15, a_end - 30, a_end + 7, p_end - 45, p_end + 15):
            reads.append(sb['arena'][o[i]:o[i] + max(c, 9100)])
            calib.append(sb['calib'][i])
    # a very long adapter (> 3 000 pooled): splice an adapter piece in several times
    i = 20
    a_beg, a_end = int(tr[i, 3, 0]), int(tr[i, 3, 1])
    piece = sb['arena'][o[i] + a_beg:o[i] + a_end]
    reads.append(np.concatenate([sb['arena'][o[i]:o[i] + a_beg]] + [piece] * 8 +
                                [sb['arena'][o[i] + a_end:o[i + 1]]]))
    calib.append(sb['calib'][i])
    arena, off = N.pack_reads(reads)
    calib = np.array(calib, dtype=N.CALIB_DTYPE)
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    want, wsp = oracle.process_batch(arena, off, calib, None, mask, want_spikes=True)
    ctx.upload(arena, off, calib)
    ctx.run(mask)
    got = ctx.download()
    assert_records_equal(got, want, ctxmsg='thresholds')
    assert_spikes_equal(ctx.download_spikes(), want, wsp)
    assert len(set(got['status'].tolist())) >= 2 and (got['bc_pushed'] == 0).any() and (got['bc_pushed'] == 1).any()
    first = np.zeros(len(reads), np.int64)
    nb = np.diff(off) // 15
    iv, cnt, start = ctx.unsplit_scan(first, nb)
    a = 3
    for r in range(len(reads)):
        if got[r]['status'] != 0 or got[r]['seg_first'][a] < 0:
            assert cnt[r] == 0
            continue
        _, sc = oracle.guppy_event_means(reads[r], calib[r], 0, int(nb[r]), got[r]['scale'], got[r]['shift'])
        wiv, wc = oracle.unsplit_scan(sc, 0, (int(got[r]['seg_last'][a]) + 1) * 15, 3012.0)
        assert cnt[r] == wc and iv[start[r]:start[r + 1]].tolist() == wiv.tolist(), r


def test_tiled_upload_equals_host_tiling(ctx):
    """pxg_batch_upload_tiled: read j of the resident batch is base read (phase + j) % K,
    replicated on the device -- records identical to uploading the same tiling from the host."""
    from poreplex_amd.synth import synth_batch
    base = synth_batch(37, seed=77, samples_per_read=14000, jitter=0.4, short_fraction=0.1)
    K, n, phase = 37, 150, 11
    order = (phase + np.arange(n)) % K
    parts = [base['arena'][base['offsets'][b]:base['offsets'][b + 1]] for b in order]
    arena, off = N.pack_reads(parts)
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    ctx.upload(arena, off, base['calib'][order])
    ctx.run(mask)
    want = ctx.download()
    ctx.upload_tiled(n, base['arena'], base['offsets'], base['calib'], phase=phase)
    ctx.run(mask)
    got = ctx.download()
    assert got.tobytes() == want.tobytes()
    # injected scaling travels with the tiling too
    ctx.upload_tiled(n, base['arena'], base['offsets'], base['calib'], base['scale_shift'], phase=phase)
    ctx.run(N.STAGE_SEGMENT)
    inj = ctx.download()
    assert np.array_equal(inj['scale'], base['scale_shift'][order, 0])
    assert np.array_equal(inj['n_pooled'], np.diff(off) // 15)


def test_polya_hook_equals_stage(ctx):
    """pxg_polya on caller-supplied scaling + segmentation == the poly(A) stage of a run."""
    from poreplex_amd.synth import synth_batch
    sb = synth_batch(64, seed=5150, samples_per_read=26000, jitter=0.3)
    ctx.upload(sb['arena'], sb['offsets'], sb['calib'])
    ctx.run(N.STAGE_ALL_DEMUX | N.STAGE_POLYA)
    want = ctx.download()
    wsp = ctx.download_spikes()
    ok = want['status'] == 0
    assert ok.sum() > 40 and want['polya_called'][ok].sum() > 10
    idx = np.nonzero(ok)[0]
    arena, off = N.pack_reads([sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in idx])
    ss = np.stack([want['scale'][idx], want['shift'][idx]], axis=1)
    got, gsp = ctx.polya(arena, off, sb['calib'][idx], ss, want['seg_first'][idx], want['seg_last'][idx])
    for f in ('polya_called', 'polya_n_spikes', 'polya_dwell_samples', 'polya_begin', 'polya_end'):
        assert np.array_equal(got[f], want[f][idx]), f
    rows = np.concatenate([wsp[0][wsp[1][i]:wsp[1][i + 1]] for i in idx])
    assert np.array_equal(gsp[0], rows, equal_nan=True) and gsp[1][-1] == len(rows)


def test_polya_window_larger_than_first_pass_scratch(ctx, oracle):
    """A featureless read whose segmentation calls 100 000 samples of poly(A): the open-ended
    extension grows the inspection window to the whole read and event detection returns more
    events than the first pass of K6 has rows for.  The read goes through the retry pass
    (pxg_polya_settle) and equals the oracle -- in a batch next to ordinary reads, and through
    the pxg_polya hook."""
    rng = np.random.default_rng(77)
    flat = (775 + rng.normal(0, 3, 130000)).astype(np.int16)
    sb = synth_batch(6, seed=5, samples_per_read=30000)
    parts = [sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in range(6)]
    parts.insert(2, flat); parts.append(flat[:120000].copy())
    cal = np.concatenate([sb['calib'][:2], sb['calib'][:1], sb['calib'][2:], sb['calib'][:1]])
    arena, off = N.pack_reads(parts)
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    want, wsp = oracle.process_batch(arena, off, cal, None, mask, want_spikes=True)
    assert want['polya_called'][2] == 1 and want['polya_end'][2] - want['polya_begin'][2] > 100000
    ctx.upload(arena, off, cal)
    ctx.run(mask)
    got = ctx.download()
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f], equal_nan=True), f
    assert_spikes_equal(ctx.download_spikes(), want, wsp)
    # spikes first, records second: either download settles the stage
    ctx.run(mask)
    assert_spikes_equal(ctx.download_spikes(), want, wsp)
    assert np.array_equal(ctx.download()['polya_end'], want['polya_end'])
    ok = np.nonzero(want['status'] == 0)[0]
    ss = np.stack([want['scale'], want['shift']], axis=1).astype(np.float32)
    sub_arena, sub_off = N.pack_reads([parts[i] for i in ok])
    res, spikes = ctx.polya(sub_arena, sub_off, cal[ok], ss[ok], want['seg_first'][ok], want['seg_last'][ok],
                            want_spikes=True)
    for f in ('polya_called', 'polya_n_spikes', 'polya_dwell_samples', 'polya_begin', 'polya_end'):
        assert np.array_equal(res[f], want[f][ok]), f
    assert_spikes_equal(spikes, want[ok], wsp[ok])


def test_prefix_staging_keeps_the_tail_of_long_reads_on_the_host(ctx, oracle, config):
    """pxg_batch_stage(_z)_prefix / pxg_process_batch: without poly(A) and the chimera scan no stage reads behind the
    segmentation's scan limit (signal_analyzer.py:347-349), so only that prefix of a long read crosses the link --
    records identical to the whole upload's (raw and encoded samples), whole-read stages refuse such a batch."""
    b = synth_batch(80, seed=933, length_dist='lognormal')
    lens = np.diff(b['offsets'])
    limit = ctx.prefix_limit_for(N.STAGE_ALL_DEMUX)
    assert limit == 100000 and (lens > limit).sum() >= 2 and ctx.prefix_limit_for(N.STAGE_ALL_DEMUX | N.STAGE_POLYA) == 0
    ctx.upload(b['arena'], b['offsets'], b['calib'])
    ctx.run(N.STAGE_ALL_DEMUX)
    want = ctx.download().copy()
    assert_records_equal(want, oracle.process_batch(b['arena'], b['offsets'], b['calib'], None, N.STAGE_ALL_DEMUX))
    # poison the device arena first: what is not copied must not matter
    junk = np.full(len(b['arena']), 12345, dtype=np.int16)
    z, chunks, _ = N.z_encode(b['arena'], b['offsets'])
    for form in ('raw', 'encoded'):
        ctx.stage(junk, b['offsets'], b['calib'])
        ctx.swap()
        ctx.stage(junk, b['offsets'], b['calib'])
        ctx.swap()                                   # both input slots hold junk now
        if form == 'raw':
            ctx.stage(b['arena'], b['offsets'], b['calib'], prefix_limit=limit)
        else:
            ctx.stage_z(N.EncodedSamples(z, chunks, 0, 0, len(b['arena'])), b['offsets'], b['calib'], prefix_limit=limit)
        ctx.swap()
        ctx.run(N.STAGE_ALL_DEMUX)
        assert_records_equal(ctx.download(), want, ctxmsg='prefix staging, ' + form)
        with pytest.raises(N.PxgError, match='prefix limit'):
            ctx.run(N.STAGE_ALL_DEMUX | N.STAGE_POLYA)
        with pytest.raises(N.PxgError, match='prefix limit'):
            ctx.download_samples(len(b['arena']))
    # the one-call form picks the limit from its mask
    assert_records_equal(ctx.process_batch(b['arena'], b['offsets'], b['calib'], None, N.STAGE_ALL_DEMUX), want)
    full = ctx.process_batch(b['arena'], b['offsets'], b['calib'], None, N.STAGE_ALL_DEMUX | N.STAGE_POLYA)
    assert_records_equal(full, oracle.process_batch(b['arena'], b['offsets'], b['calib'], None, N.STAGE_ALL_DEMUX | N.STAGE_POLYA))


def test_small_calls_are_merged_into_one_batch(ctx, oracle):
    """The reference's workers get 128 reads per call (commandline.py:402); calls that arrive from several threads
    while the pipeline is full run as ONE batch (include/pxg.h "small calls share a batch").  48 calls of 1 .. 300
    reads from 12 threads, two stage masks mixed in: every call gets exactly the records a call of its own gets,
    and calls were in fact merged."""
    import threading
    b = synth_batch(1500, seed=941, samples_per_read=20000, jitter=0.5, short_fraction=0.03)
    o = b['offsets']
    want = ctx.process_batch(b['arena'], o, b['calib'])           # one call on its own
    assert_records_equal(want[:200], oracle.process_batch(b['arena'][:o[200]], o[:201], b['calib'][:200]))
    rng = np.random.default_rng(3)
    cuts = np.sort(rng.choice(np.arange(1, 1500), 47, replace=False))
    spans = list(zip(np.r_[0, cuts], np.r_[cuts, 1500]))
    masks = [N.STAGE_ALL_DEMUX if k % 7 else (N.STAGE_SCALER | N.STAGE_SEGMENT) for k in range(len(spans))]
    got, errors = [None] * len(spans), []
    g0, c0 = ctx.merge_stats()

    z, chunks, cbase = N.z_encode(b['arena'], o)

    def worker(ks):
        try:
            for k in ks:
                lo, hi = spans[k]
                if k % 3 == 1:            # every third call brings its samples encoded (decoded into its stretch of the batch)
                    c0, c1 = int(cbase[lo]), int(cbase[hi])
                    b0 = int(chunks['data_off'][c0]) if c0 < len(chunks) else len(z)
                    b1 = int(chunks['data_off'][c1]) if c1 < len(chunks) else len(z)
                    enc = N.EncodedSamples(z[b0:b1], chunks[c0:c1], b0, int(o[lo]), int(o[hi] - o[lo]))
                    got[k] = ctx.process_batch_ex(enc, o[lo:hi + 1] - o[lo], b['calib'][lo:hi], masks[k])['records']
                    continue
                got[k] = ctx.process_batch(b['arena'][o[lo]:o[hi]], o[lo:hi + 1] - o[lo], b['calib'][lo:hi], None, masks[k])
        except BaseException as exc:      # noqa: B902 (reported below)
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(range(t, len(spans), 12),)) for t in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    seg_only = ctx.process_batch(b['arena'], o, b['calib'], None, N.STAGE_SCALER | N.STAGE_SEGMENT)
    for k, (lo, hi) in enumerate(spans):
        ref = want if masks[k] == N.STAGE_ALL_DEMUX else seg_only
        assert_records_equal(got[k], ref[lo:hi], ctxmsg='merged call %d' % k)
    g1, c1 = ctx.merge_stats()
    # fewer batches than calls: merging happened (calls that met a group of the OTHER stage mask went alone, uncounted)
    assert c1 - c0 >= 24 and g1 - g0 < c1 - c0, (g0, c0, g1, c1)


def test_spline_position_is_kept_below_one_for_tiny_negative_arguments(config, tmp_path, arith):
    """The position of a look-up argument inside its table segment is min(u - floor(u), largest float below 1)
    (oracle/pxo_core.c sig_position; on the device one v_fract_f32, checked for every float by
    tools/ubench/fract_check.hip).  A scaler network whose recurrent and input weights are zero has gate arguments that
    are exactly its biases: a ladder of tiny negative values (where u + 1 rounds to 1.0), -0.0, denormals, integers,
    values just below integers and both table ends in the biases of both layers -- the kernels must give the oracle's
    outputs bit for bit (they differ in the last bits if either side evaluates the cubic AT 1.0)."""
    import copy
    from oracle.pxo import Oracle
    from poreplex_amd.config import load_model_arrays
    m = load_model_arrays(config['signal_processing']['scaler_model'])
    ladder = np.array([-1e-9, -2.0 ** -26, -2.0 ** -27, -1e-12, -1e-30, -1e-40, -0.0, 0.0, 1e-40, -2.0 ** -25, -3e-8, -5.96e-8,
                       -1.0, -1.0000001, -0.99999994, 3.0, 2.9999998, -3.0000002, 40.0, -40.0, 31.999998, -32.0, 0.5, -0.5],
                      dtype=np.float32)
    rng = np.random.default_rng(17)
    for key in ('lstm1_bias', 'lstm2_bias'):
        b = ladder[rng.integers(0, len(ladder), 192)].astype(np.float32)
        b[:len(ladder)] = ladder
        b[48:48 + len(ladder)] = ladder[::-1]
        b[96:96 + len(ladder)] = ladder / np.float32(2.0)         # tanh columns: table units are 32 z
        b[144:144 + len(ladder)] = ladder
        m[key] = b / np.float32(16.0)                              # table units are 16 z: u = 16 b exactly
    for key in ('lstm1_kernel', 'lstm1_recurrent', 'lstm2_kernel', 'lstm2_recurrent'):
        m[key] = np.zeros_like(m[key])
    m['lstm2_kernel'] = (rng.choice([0.0, 2.0 ** -30, -2.0 ** -31], m['lstm2_kernel'].shape)).astype(np.float32)
    m['dense_kernel'] = rng.normal(0, 1, m['dense_kernel'].shape).astype(np.float32)
    path = str(tmp_path / 'edge_scaler.npz')
    np.savez(path, **m)
    cfg = copy.deepcopy(config)
    cfg['signal_processing']['scaler_model'] = path
    head = rng.normal(0, 1, (40, 2000)).astype(np.float32)
    orc = Oracle(cfg)
    c = N.NativeContext(cfg, device_id=0)
    try:
        got = c.scaler_lstm(head)
    finally:
        c.close()
    want = np.stack([orc.scaler_forward(h) for h in head])
    assert np.array_equal(got, want), np.abs(got - want).max()
