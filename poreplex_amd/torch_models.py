"""The two Keras networks of the path re-expressed as PyTorch modules -- for weight
loading, autograd experiments (tools/make_prototypes.py) and as an independent fp32
statement of the forward passes in the tests.  NOT on the product path: the GPU
kernels (csrc/k_lstm.hip) read the same weight bundles through native.NativeConfig.

  ScalerNet  scaler-r3:       LSTM(48, return_sequences) -> LSTM(48) -> Dense(2)
                              (signal_loader.py:49-75, 96-97)
  DemuxNet   demux-tetra-r4:  Bidirectional(LSTM(48), concat) -> LSTM(64) -> Dense(5)
                              softmax (barcoding.py:37-47, 106-107)

Keras stores an LSTM as kernel [in, 4H], recurrent_kernel [H, 4H], bias [4H] with gate
order i, f, c, o -- torch.nn.LSTM uses the same order with transposed matrices.
"""
import numpy as np
import torch

from .config import load_model_arrays

__all__ = ['keras_lstm', 'ScalerNet', 'DemuxNet']


def keras_lstm(kernel, recurrent, bias):
    """torch.nn.LSTM (batch_first) carrying one Keras LSTM layer's weights."""
    m = torch.nn.LSTM(kernel.shape[0], recurrent.shape[0], batch_first=True)
    with torch.no_grad():
        m.weight_ih_l0.copy_(torch.from_numpy(np.ascontiguousarray(kernel.T)))
        m.weight_hh_l0.copy_(torch.from_numpy(np.ascontiguousarray(recurrent.T)))
        m.bias_ih_l0.copy_(torch.from_numpy(np.ascontiguousarray(bias)))
        m.bias_hh_l0.zero_()
    for p in m.parameters():
        p.requires_grad_(False)
    return m


class ScalerNet(torch.nn.Module):
    """x [B, 2000] (standardised, left zero-padded head) -> [B, 2] standardised (scale, shift)."""

    def __init__(self, bundle='MIN106-RNA001/scaler-r3.npz'):
        super().__init__()
        w = load_model_arrays(bundle)
        self.l1 = keras_lstm(w['lstm1_kernel'], w['lstm1_recurrent'], w['lstm1_bias'])
        self.l2 = keras_lstm(w['lstm2_kernel'], w['lstm2_recurrent'], w['lstm2_bias'])
        self.dk = torch.from_numpy(np.ascontiguousarray(w['dense_kernel']))
        self.db = torch.from_numpy(np.ascontiguousarray(w['dense_bias']))

    def forward(self, x):
        h, _ = self.l1(x.unsqueeze(-1))
        _, (hn, _) = self.l2(h)
        return hn[0] @ self.dk + self.db


class DemuxNet(torch.nn.Module):
    """x [B, 300] (normalised adapter window) -> log-probabilities [B, 5]."""

    def __init__(self, bundle='MIN106-RNA001/demux-tetra-r4.npz'):
        super().__init__()
        w = load_model_arrays(bundle)
        self.fwd = keras_lstm(w['fwd_kernel'], w['fwd_recurrent'], w['fwd_bias'])
        self.bwd = keras_lstm(w['bwd_kernel'], w['bwd_recurrent'], w['bwd_bias'])
        self.top = keras_lstm(w['top_kernel'], w['top_recurrent'], w['top_bias'])
        self.dk = torch.from_numpy(np.ascontiguousarray(w['dense_kernel']))
        self.db = torch.from_numpy(np.ascontiguousarray(w['dense_bias']))

    def forward(self, x):
        x = x.unsqueeze(-1)
        hf, _ = self.fwd(x)
        hb, _ = self.bwd(torch.flip(x, dims=[1]))          # Keras go_backwards ...
        h = torch.cat([hf, torch.flip(hb, dims=[1])], dim=-1)   # ... re-reversed, then concat
        _, (hn, _) = self.top(h)
        return torch.log_softmax(hn[0] @ self.dk + self.db, dim=-1)
