"""Quick GPU check of the PXG_LSTM_Q8 kernels against the oracle (development aid; the suite's
-m gpu tests are the real gate)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from poreplex_amd.config import default_config
from poreplex_amd.native import NativeContext
from oracle.pxo import Oracle

cfg = default_config()
stages = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'batch0.stages.npz')))
heads = stages['scaler_in']
ctx = NativeContext(cfg, device_id=0)
orc = Oracle(cfg)
print('arith', ctx.ncfg.struct.lstm_arith if hasattr(ctx, 'ncfg') else '?', orc.cfg.lstm_arith)
which = sys.argv[1] if len(sys.argv) > 1 else 'scaler'
if which in ('scaler', 'all'):
    for n in (1, 16, 17, 70, 8200, 9999, 13000):
        rng = np.random.default_rng(n)
        rows = np.stack([heads[i % len(heads)] for i in range(n)]).copy()
        rows = rows + rng.normal(0, 1.5, (n, 1)).astype(np.float32)
        t0 = time.time(); got = ctx.scaler_lstm(rows); t1 = time.time()
        pick = rng.choice(n, min(n, 16), replace=False)
        want = np.stack([orc.scaler_forward(rows[i]) for i in pick])
        ok = np.array_equal(got[pick], want)
        print('scaler n=%d: %s  max|d|=%g  (%.1f ms incl. copies)' % (n, 'BIT-EXACT' if ok else 'MISMATCH', np.abs(got[pick] - want).max(), (t1 - t0) * 1e3), flush=True)
if which in ('demux', 'all'):
    wins = stages['demux_in'] if 'demux_in' in stages else None
    if wins is None:
        wins = np.random.default_rng(1).normal(0, 1, (64, 300)).astype(np.float32)
    for n in (1, 17, 70, 8200, 10000, 20001):
        rng = np.random.default_rng(n)
        rows = np.stack([wins[i % len(wins)] for i in range(n)]).copy()
        rows = rows + rng.normal(0, 0.05, rows.shape).astype(np.float32)
        if n > 20:
            rows[5, :40] = -1000.0
        t0 = time.time(); got = ctx.demux_lstm(rows); t1 = time.time()
        pick = rng.choice(n, min(n, 24), replace=False)
        want = np.stack([orc.demux_forward(rows[i]) for i in pick])
        ok = np.array_equal(got[pick][:, :want.shape[1]], want)
        print('demux n=%d: %s  max|d|=%g  (%.1f ms incl. copies)' % (n, 'BIT-EXACT' if ok else 'MISMATCH', np.abs(got[pick][:, :want.shape[1]] - want).max(), (t1 - t0) * 1e3), flush=True)
ctx.close()
