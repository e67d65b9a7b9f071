#!/usr/bin/env python
"""Synthetic barcode prototypes: one 300-step normalised adapter window per class that
the shipped demux network (demux-tetra-r4) classifies with high confidence.

No barcoded FAST5 exists near the reference, so the synthetic reads of tests/bench had
no barcode signal and every read came out "undetermined".  This script finds, by
gradient ascent through a PyTorch re-expression of the network (Bidirectional
LSTM(48) -> LSTM(64) -> Dense(5) softmax; weights from the converted Keras bundle),
an input window per class, and then VERIFIES each one with the oracle's exact
arithmetic.  Output: poreplex_amd/presets/MIN106-RNA001/synthetic-barcode-prototypes.npy
(5 x 300 float32, z-score units; row 0 = decoy class).  PyTorch is used for autograd
only; nothing in the product imports this.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pxo import Oracle  # noqa: E402
from poreplex_amd.config import default_config, load_model_arrays  # noqa: E402
from poreplex_amd.torch_models import DemuxNet  # noqa: E402

OUT = os.path.join(ROOT, 'poreplex_amd', 'presets', 'MIN106-RNA001',
                   'synthetic-barcode-prototypes.npy')


def standardise(x):
    """median 0, 1.4826*MAD 1 -- what BarcodeDemultiplexer.normalize_signal yields."""
    med = x.median(dim=1, keepdim=True).values
    mad = (x - med).abs().median(dim=1, keepdim=True).values * 1.4826
    return (x - med) / mad


def main():
    torch.manual_seed(922)
    cfg = default_config()
    w = load_model_arrays(cfg['demultiplexing']['demux_model'])
    net = DemuxNet(cfg['demultiplexing']['demux_model'])
    orc = Oracle(cfg)
    T = int(cfg['demultiplexing']['signal_trim_length'])
    n_cls = w['dense_kernel'].shape[1]
    protos = np.zeros((n_cls, T), dtype=np.float32)
    for cls in range(n_cls):
        best = None
        for trial in range(4):
            x = (torch.randn(8, T) * 0.8).requires_grad_(True)
            opt = torch.optim.Adam([x], lr=0.05)
            for it in range(250):
                opt.zero_grad()
                xs = standardise(x).clamp(-3.0, 3.0)
                lp = net(xs)[:, cls]
                # confident, but smooth enough to survive 15-sample pooling noise
                loss = -lp.mean() + 0.02 * (xs[:, 1:] - xs[:, :-1]).pow(2).mean()
                loss.backward()
                opt.step()
            with torch.no_grad():
                xs = standardise(x).clamp(-3.0, 3.0)
                lp = net(xs)[:, cls]
                k = int(lp.argmax())
                cand = xs[k].numpy().astype(np.float32)
            p = orc.demux_forward(orc.normalize_signal(cand * 7.4 + 80.5))   # exact arithmetic
            if best is None or p[cls] > best[0]:
                best = (float(p[cls]), cand)
            if best[0] > 0.995:
                break
        protos[cls] = best[1]
        print('class %d: oracle p = %.6f' % (cls, best[0]))
        assert best[0] > 0.95, 'no confident prototype for class %d' % cls
    np.save(OUT, protos)
    print('wrote', OUT, protos.shape, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
