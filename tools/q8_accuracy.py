#!/usr/bin/env python
"""The two recurrent networks ALONE against the exact Keras equations in float64 (NumPy), per product arithmetic of
the oracle (CPU only; the kernels are bit-exact with the oracle in both: tests -m gpu).  Inputs: the scaler heads and
classifier windows of the golden batch, jittered.  usage: python tools/q8_accuracy.py [f32|q8] [n]
(round 5: one weight exponent per gate block -- profiles/r05/q8_accuracy_networks_alone.txt)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
arith = sys.argv[1] if len(sys.argv) > 1 else 'q8'
os.environ['PXG_LSTM_ARITH'] = arith
from oracle.pxo import Oracle  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from test_oracle_crosscheck import demux_f64, scaler_f64  # noqa: E402

np.seterr(over='ignore')
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
orc = Oracle(default_config())
st = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'batch0.stages.npz')))
heads, wins = st['scaler_in'], st['demux_in']
rng = np.random.default_rng(5)
es, ed = [], []
for i in range(n):
    h = (heads[i % len(heads)] + np.float32(rng.normal(0, 1.0))).astype(np.float32)
    es.append(np.abs(orc.scaler_forward(h).astype(np.float64) - scaler_f64(h)).max())
    w = (wins[i % len(wins)] + rng.normal(0, 0.05, wins.shape[1]).astype(np.float32)).astype(np.float32)
    ed.append(np.abs(orc.demux_forward(w)[:5].astype(np.float64) - demux_f64(w)).max())
print('{:3s} {:4d} heads / windows: scaler output |d| max {:.3g} mean {:.3g} | demux softmax |d| max {:.3g} mean {:.3g}'.format(
    arith, n, max(es), float(np.mean(es)), max(ed), float(np.mean(ed))))
