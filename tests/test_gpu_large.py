"""Size-independent properties at BASELINE.json's large-batch sizes (-m gpu)."""
import numpy as np
import pytest

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
    iv, cnt = ctx.unsplit_scan(np.zeros(len(lens), np.int64), nb)
    assert np.array_equal(cnt[:n0], cnt[-n0:]) and np.array_equal(iv[:n0], iv[-n0:])
