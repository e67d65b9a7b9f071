#!/opt/conda/bin/python3.9
"""Convert the reference's Keras model files into plain .npz weight bundles.

Run ONCE in the build container (needs h5py, which only /opt/conda/bin/python3.9
has here); the GPU box has no h5py, so the product loads the .npz bundles.

Source files (data, not code): /root/reference/poreplex/presets/MIN106-RNA001/
  scaler-r3.hdf5       -- Keras 2.2.4-tf Sequential[LSTM48(seq) -> LSTM48 -> Dense2]
  demux-tetra-r4.hdf5  -- Sequential[Bidirectional(LSTMCell48) -> LSTMCell64 -> Dense5]
The attrs the reference parses at load time (poreplex/signal_loader.py:52-60,
poreplex/barcoding.py:55-60) are carried over as arrays/JSON strings.
"""
import ast
import json
import sys

import h5py
import numpy as np


def _s(x):
    return x.decode() if isinstance(x, bytes) else str(x)


SRC = '/root/reference/poreplex/presets/MIN106-RNA001'
DST = sys.argv[1] if len(sys.argv) > 1 else 'poreplex_amd/presets/MIN106-RNA001'


def grab(h5, path):
    return np.ascontiguousarray(h5['model_weights/' + path][()], dtype=np.float32)


def scaler():
    with h5py.File(SRC + '/scaler-r3.hdf5', 'r') as h5:
        mw = h5['model_weights'].attrs
        input_defs = ast.literal_eval(_s(mw['input_defs']))
        xfrm = ast.literal_eval(_s(mw['output_transform']))
        out = {
            'lstm1_kernel': grab(h5, 'lstm_1/lstm_1/kernel:0'),
            'lstm1_recurrent': grab(h5, 'lstm_1/lstm_1/recurrent_kernel:0'),
            'lstm1_bias': grab(h5, 'lstm_1/lstm_1/bias:0'),
            'lstm2_kernel': grab(h5, 'lstm_2/lstm_2/kernel:0'),
            'lstm2_recurrent': grab(h5, 'lstm_2/lstm_2/recurrent_kernel:0'),
            'lstm2_bias': grab(h5, 'lstm_2/lstm_2/bias:0'),
            'dense_kernel': grab(h5, 'dense_2/dense_2/kernel:0'),
            'dense_bias': grab(h5, 'dense_2/dense_2/bias:0'),
            # float64, exactly as the literal in the file parses
            'output_transform': np.array([xfrm['scale_mean'], xfrm['scale_std'],
                                          xfrm['shift_mean'], xfrm['shift_std']],
                                         dtype=np.float64),
            'input_stride': np.int64(input_defs['stride']),
            'input_length': np.int64(input_defs['length']),
            'input_min_length': np.int64(input_defs['min_length']),
            'model_version': np.array(_s(mw['model_version'])),
            'meta': np.array(json.dumps({'input_defs': input_defs,
                                         'output_transform': xfrm})),
        }
    np.savez(DST + '/scaler-r3.npz', **out)
    return out


def demux():
    with h5py.File(SRC + '/demux-tetra-r4.hdf5', 'r') as h5:
        calib = h5['poreplex_params/calibration'][()]
        assert np.array_equal(calib['phred'], np.arange(len(calib)))
        b = 'bidirectional_2/bidirectional_2/'
        out = {
            'fwd_kernel': grab(h5, b + 'forward_rnn/kernel:0'),
            'fwd_recurrent': grab(h5, b + 'forward_rnn/recurrent_kernel:0'),
            'fwd_bias': grab(h5, b + 'forward_rnn/bias:0'),
            'bwd_kernel': grab(h5, b + 'backward_rnn/kernel:0'),
            'bwd_recurrent': grab(h5, b + 'backward_rnn/recurrent_kernel:0'),
            'bwd_bias': grab(h5, b + 'backward_rnn/bias:0'),
            'top_kernel': grab(h5, 'rnn_1/rnn_1/kernel:0'),
            'top_recurrent': grab(h5, 'rnn_1/rnn_1/recurrent_kernel:0'),
            'top_bias': grab(h5, 'rnn_1/rnn_1/bias:0'),
            'dense_kernel': grab(h5, 'dense_2/dense_2/kernel:0'),
            'dense_bias': grab(h5, 'dense_2/dense_2/bias:0'),
            'calibration': np.ascontiguousarray(calib['pred_score'], dtype=np.float64),
            'loss_weights': h5['poreplex_params/loss_weights'][()].astype(np.float32),
        }
    np.savez(DST + '/demux-tetra-r4.npz', **out)
    return out


if __name__ == '__main__':
    s = scaler()
    d = demux()
    n_s = sum(v.size for k, v in s.items() if v.dtype == np.float32)
    n_d = sum(v.size for k, v in d.items()
              if v.dtype == np.float32 and k != 'loss_weights')
    print('scaler params', n_s, 'demux params', n_d)
