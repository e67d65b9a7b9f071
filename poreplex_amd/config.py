"""Preset loading and assembly of the flat ``config`` dict the hot path reads.

Mirrors what the reference's CLI does before it hands ``config`` to the
workers (poreplex/commandline.py:60-76 preset lookup, :267-296 assembly); only
the keys the per-read processor consumes are produced.
"""
import os

import numpy as np
import yaml

PRESET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'presets')
DEFAULT_PRESET = 'rna-r941'

STATE_NAMES = ('pre-leader', 'leader-low', 'leader-high', 'adapter',
               'polya-tail', 'transcript')


def load_preset(name=DEFAULT_PRESET):
    """Resolve a preset by path or by name, like commandline.py:60-76."""
    candidates = [name, os.path.join(PRESET_DIR, name),
                  os.path.join(PRESET_DIR, name + '.cfg')]
    for path in candidates:
        if os.path.isfile(path):
            with open(path) as fh:
                return yaml.safe_load(fh)
    raise FileNotFoundError('Could not find a preset named {!r}'.format(name))


def default_config(preset=DEFAULT_PRESET, **overrides):
    """Build the worker ``config`` dict (commandline.py:267-296 image)."""
    config = load_preset(preset)
    config.update({
        'inputdir': '.',
        'outputdir': '.',
        'barcoding': True,
        'measure_polya': False,
        'trim_adapter': False,
        'trim_adapter_as_intended': False,      # NOT a reference option: see SignalAnalysis.trim_adapter
        'filter_unsplit_reads': False,
        'minimum_sequence_length': 10,
        'albacore_onthefly': False,
        'dump_adapter_signals': False,
        'dump_basecalls': False,
        'barcoding_quality_filter': 18,
        'batch_chunk_size': 128,
        'parallel': 1,
    })
    config.update(overrides)
    return config


def resolve_model_path(relpath):
    """Model files live under presets/ (signal_loader.py:50-51,
    barcoding.py:52-53).  The reference names Keras .hdf5 files; this build
    ships the same weights as .npz, so an .hdf5 name maps to its .npz twin."""
    path = relpath if os.path.isabs(relpath) else os.path.join(PRESET_DIR, relpath)
    if os.path.isfile(path) and not path.endswith(('.hdf5', '.h5')):
        return path
    twin = os.path.splitext(path)[0] + '.npz'
    if os.path.isfile(twin):
        return twin
    if os.path.isfile(path):
        return path
    raise FileNotFoundError('model file {!r} (or its .npz twin) not found'.format(relpath))


def load_model_arrays(relpath):
    """Return {name: ndarray} for a model bundle (.npz, or .hdf5 if h5py exists)."""
    path = resolve_model_path(relpath)
    if path.endswith('.npz'):
        with np.load(path, allow_pickle=False) as npz:
            return {k: npz[k] for k in npz.files}
    return _load_keras_hdf5(path)


def _load_keras_hdf5(path):  # pragma: no cover - needs h5py
    import ast
    import h5py
    out = {}
    with h5py.File(path, 'r') as h5:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                out[name] = np.ascontiguousarray(obj[()])
        h5['model_weights'].visititems(visit)
        attrs = h5['model_weights'].attrs
        _s = lambda x: x.decode() if isinstance(x, bytes) else str(x)
        keymap = {}
        for name in list(out):
            leaf = name.split('/')[-2:] if '/' in name else [name]
            keymap[name] = '/'.join(leaf)
        if 'input_defs' in attrs:   # scaler
            idefs = ast.literal_eval(_s(attrs['input_defs']))
            xfrm = ast.literal_eval(_s(attrs['output_transform']))
            res = {
                'lstm1_kernel': out['lstm_1/lstm_1/kernel:0'],
                'lstm1_recurrent': out['lstm_1/lstm_1/recurrent_kernel:0'],
                'lstm1_bias': out['lstm_1/lstm_1/bias:0'],
                'lstm2_kernel': out['lstm_2/lstm_2/kernel:0'],
                'lstm2_recurrent': out['lstm_2/lstm_2/recurrent_kernel:0'],
                'lstm2_bias': out['lstm_2/lstm_2/bias:0'],
                'dense_kernel': out['dense_2/dense_2/kernel:0'],
                'dense_bias': out['dense_2/dense_2/bias:0'],
                'output_transform': np.array([xfrm['scale_mean'], xfrm['scale_std'],
                                              xfrm['shift_mean'], xfrm['shift_std']]),
                'input_stride': np.int64(idefs['stride']),
                'input_length': np.int64(idefs['length']),
                'input_min_length': np.int64(idefs['min_length']),
            }
            return res
        b = 'bidirectional_2/bidirectional_2/'
        calib = h5['poreplex_params/calibration'][()]
        return {
            'fwd_kernel': out[b + 'forward_rnn/kernel:0'],
            'fwd_recurrent': out[b + 'forward_rnn/recurrent_kernel:0'],
            'fwd_bias': out[b + 'forward_rnn/bias:0'],
            'bwd_kernel': out[b + 'backward_rnn/kernel:0'],
            'bwd_recurrent': out[b + 'backward_rnn/recurrent_kernel:0'],
            'bwd_bias': out[b + 'backward_rnn/bias:0'],
            'top_kernel': out['rnn_1/rnn_1/kernel:0'],
            'top_recurrent': out['rnn_1/rnn_1/recurrent_kernel:0'],
            'top_bias': out['rnn_1/rnn_1/bias:0'],
            'dense_kernel': out['dense_2/dense_2/kernel:0'],
            'dense_bias': out['dense_2/dense_2/bias:0'],
            'calibration': np.ascontiguousarray(calib['pred_score'], dtype=np.float64),
        }


def norm_ppf(q, loc, scale):
    """scipy.stats.norm.ppf(q, loc, scale) (signal_loader.py:67-68)."""
    try:
        from scipy.stats import norm
        return float(norm.ppf(q, loc, scale))
    except ImportError:  # pragma: no cover
        from statistics import NormalDist
        return NormalDist(loc, scale).inv_cdf(q)
