"""Which third-party engines run under the reference when tests/golden/* is generated (tools/make_golden.py).

The reference delegates three computations to libraries this build image does not have: the two Keras models
(`model.predict`, signal_loader.py:96-97 / barcoding.py:106-107) to TensorFlow and the segmentation / unsplit
Viterbi (`model.viterbi`, signal_analyzer.py:352, worker_persistence.py:95-121) to pomegranate.  make_golden.py
therefore ran the reference's glue with stand-ins that call the ORACLE (oracle/libpxo.so) -- which pins
everything the reference itself computes and leaves rows a4 / a7 / a12 "parity unpinned" (DESIGN.md section 4).

This module is the switch that lets anybody who HAS the libraries close that gap with one command:

    /path/to/python3 tools/make_golden.py --engines real        # needs tensorflow + pomegranate importable

  --engines auto (default)  import tensorflow / pomegranate; each one that imports is used as it is, each one that
                            raises ImportError is replaced by its oracle stand-in
  --engines real            as auto, but a missing library is an error (nothing is written)
  --engines stub            always the oracle stand-ins (the committed bit-exact sets tests/golden/, tests/golden/q8/)

A set generated with at least one REAL engine is written to tests/golden/real/ and never replaces the bit-exact
sets: the product matches real TensorFlow / pomegranate within the tolerances north_star states (softmax 1e-4,
identical labels / boundaries), not bit for bit, and tests/test_real_engines.py checks exactly that -- or skips,
saying that the committed set was made with stand-ins.  Every set records what made it in `engines.json`.

No NumPy / h5py here: the module is imported by the CPU test suite (tests/test_real_engines.py) under any Python.
"""
import importlib
import json
import os
import sys

REAL, STUB = 'real', 'oracle-stub'
KERAS_MODULES = ('tensorflow', 'tensorflow.keras', 'tensorflow.keras.backend', 'tensorflow.keras.losses',
                 'tensorflow.keras.metrics')


def parse_mode(argv):
    mode = argv[argv.index('--engines') + 1] if '--engines' in argv else 'auto'
    if mode not in ('auto', 'real', 'stub'):
        raise SystemExit('--engines takes auto, real or stub')
    return mode


def _importable(name):
    try:
        importlib.import_module(name)
        return True
    except ImportError:
        return False


def select(mode, install_keras_stub, install_hmm_stub, wrap_real_keras=None):
    """Decide, per engine, real or stand-in; install the stand-ins that are needed (the two callables put fake
    modules into sys.modules) and, for a real Keras, let `wrap_real_keras(tensorflow)` hook the prediction log.
    Returns {'keras': 'real' | 'oracle-stub', 'hmm': ...}.  ImportError is the ONLY reason for a stand-in in
    auto mode; in real mode it is fatal."""
    engines = {}
    for key, probe, install in (('keras', 'tensorflow', install_keras_stub), ('hmm', 'pomegranate', install_hmm_stub)):
        have = mode != 'stub' and _importable(probe)
        if mode == 'real' and not have:
            raise SystemExit('--engines real: `import {}` failed in {} -- install it or use --engines auto'.format(
                probe, sys.executable))
        if have:
            engines[key] = REAL
            if key == 'keras' and wrap_real_keras is not None:
                wrap_real_keras(importlib.import_module('tensorflow'))
        else:
            engines[key] = STUB
            install()
    return engines


def any_real(engines):
    return REAL in engines.values()


def describe(engines, extra=None):
    doc = {'engines': dict(engines),
           'meaning': {'keras': 'Keras model.predict of scaler-r3 / demux-tetra-r4 (rows a4, a12)',
                       'hmm': 'pomegranate HiddenMarkovModel.viterbi (rows a6, a7, a19)'},
           'real': 'the third-party library itself ran under the reference',
           'oracle-stub': 'oracle/libpxo.so stood in (the library is not in the build image): '
                          'these rows are self-consistent, not pinned'}
    if extra:
        doc.update(extra)
    return doc


def write(outdir, engines, extra=None):
    with open(os.path.join(outdir, 'engines.json'), 'w') as fh:
        json.dump(describe(engines, extra), fh, indent=1, sort_keys=True)
        fh.write('\n')


def read(outdir):
    """engines of a golden set, or None when the set has no engines.json (does not exist)."""
    try:
        with open(os.path.join(outdir, 'engines.json')) as fh:
            return json.load(fh)['engines']
    except (OSError, KeyError, ValueError):
        return None
