"""Size-independent properties at BASELINE.json's large-batch sizes (-m gpu)."""
import numpy as np
import pytest

from conftest import assert_spikes_equal

from poreplex_amd import native as N
from poreplex_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


def test_100k_read_batch_grid_limits_and_tiling(ctx, oracle):
    """configs[3] scale: 100 000 reads in ONE resident batch (more reads than
    gridDim.y allows, > 6000 LSTM tiles).  The batch is 2 000 distinct reads
    tiled 50 times, so every copy must produce byte-identical records, and the
    first copies are checked against the oracle."""
    base = synth_batch(2000, seed=31337, samples_per_read=9000, jitter=0.3, short_fraction=0.02)
    reps = 50
    n0 = 2000
    arena = np.tile(base['arena'], reps)
    lens = np.tile(np.diff(base['offsets']), reps)
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    calib = np.tile(base['calib'], reps)
    ctx.upload(arena, off, calib)
    ctx.run(N.STAGE_ALL_DEMUX | N.STAGE_POLYA)
    res = ctx.download()
    assert len(res) == n0 * reps
    first = res[:n0]
    for k in range(1, reps):
        assert res[k * n0:(k + 1) * n0].tobytes() == first.tobytes(), k
    want = oracle.process_batch(base['arena'][:base['offsets'][256]], base['offsets'][:257],
                                base['calib'][:256], None, N.STAGE_ALL_DEMUX | N.STAGE_POLYA)
    for f in first.dtype.names:
        assert np.array_equal(first[f][:256], want[f], equal_nan=True), f
    # the chimera scan at the same size
    nb = lens // 15
    iv, cnt, start = ctx.unsplit_scan(np.zeros(len(lens), np.int64), nb)
    assert np.array_equal(cnt[:n0], cnt[-n0:]) and (cnt >= 0).all()
    assert np.array_equal(iv[:start[n0]], iv[start[-1 - n0]:])


def test_configs3_full_shape_100k_reads_60k_samples(ctx, oracle):
    """BASELINE configs[3] at its stated shape: 100 000 reads x ~60 000 samples (12 GB of
    int16) resident on one GPU, all stages + the chimera scan.  The batch is 2 048 distinct
    reads tiled ON THE DEVICE (pxg_batch_upload_tiled), so (i) every copy of a read must give
    byte-identical records and candidates, (ii) the first 256 reads are checked against the
    oracle field by field."""
    K, n = 2048, 100000
    base = synth_batch(K, seed=925, samples_per_read=60000)
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    ctx.upload_tiled(n, base['arena'], base['offsets'], base['calib'])
    ctx.run(mask)
    res = ctx.download()
    spk = ctx.download_spikes()
    assert len(res) == n
    rows, soff = spk
    per_tile = int(soff[K])
    for k in range(1, n // K):
        assert res[k * K:(k + 1) * K].tobytes() == res[:K].tobytes(), k
        assert np.array_equal(soff[k * K:(k + 1) * K + 1] - soff[k * K], soff[:K + 1]), k
        assert rows[soff[k * K]:soff[(k + 1) * K]].tobytes() == rows[:per_tile].tobytes(), k
    tail = n - (n // K) * K
    assert res[-tail:].tobytes() == res[:tail].tobytes()
    assert (res['status'] == 0).mean() > 0.9 and res['polya_called'].mean() > 0.5
    want, wsp = oracle.process_batch(base['arena'][:base['offsets'][256]], base['offsets'][:257],
                                     base['calib'][:256], None, mask, want_spikes=True)
    for f in res.dtype.names:
        assert np.array_equal(res[f][:256], want[f], equal_nan=True), f
    assert_spikes_equal((rows[:soff[256]], soff[:257]), want, wsp)
    lens = np.diff(base['offsets'])[np.arange(n) % K]
    nb = lens // 15
    iv, cnt, start = ctx.unsplit_scan(np.zeros(n, np.int64), nb)
    assert (cnt >= 0).all()
    assert np.array_equal(cnt[:K], cnt[K:2 * K]) and np.array_equal(cnt[:tail], cnt[-tail:])
    assert np.array_equal(iv[:start[K]], iv[start[K]:start[2 * K]])
    a = 3
    for r in range(64):
        if res[r]['status'] != 0 or res[r]['seg_first'][a] < 0:
            assert cnt[r] == 0
            continue
        raw = base['arena'][base['offsets'][r]:base['offsets'][r + 1]]
        _, sc = oracle.guppy_event_means(raw, base['calib'][r], 0, int(nb[r]), res[r]['scale'], res[r]['shift'])
        wiv, wc = oracle.unsplit_scan(sc, 0, (int(res[r]['seg_last'][a]) + 1) * 15, 3012.0)
        assert cnt[r] == wc and iv[start[r]:start[r + 1]].tolist() == wiv.tolist(), r


def test_polya_retry_pass_in_several_launches(ctx, oracle):
    """20 000 copies of a featureless 130 000-sample read (5 GB of int16, tiled on the device) next
    to 24 ordinary reads: every copy outgrows the first-pass event scratch of K6, so the retry
    pass has more reads than one launch's scratch budget covers and runs in several launches.
    Every copy must give the oracle's record of that read."""
    rng = np.random.default_rng(77)
    flat = (775 + rng.normal(0, 3, 130000)).astype(np.int16)
    sb = synth_batch(24, seed=5, samples_per_read=30000)
    o = sb['offsets']
    parts = [flat] + [sb['arena'][o[i]:o[i + 1]] for i in range(24)]
    arena, off = N.pack_reads(parts)
    cal = np.concatenate([sb['calib'][:1], sb['calib']])
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    want = oracle.process_batch(arena, off, cal, None, mask)
    assert want['polya_called'][0] == 1 and want['polya_end'][0] - want['polya_begin'][0] > 100000
    n = 25 * 20000                       # read j = base read j % 25
    ctx.upload_tiled(n, arena, off, cal)
    ctx.run(mask)
    res = ctx.download()
    for f in res.dtype.names:
        got = res[f].reshape((20000, 25) + res[f].shape[1:])
        assert np.array_equal(got[0], want[f], equal_nan=True), f
        assert (got == got[0]).all() or f == 'probs' and np.array_equal(got, np.broadcast_to(got[0], got.shape), equal_nan=True), f
