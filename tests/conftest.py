import json
import os
import sys

# torch-CPU legs (training, DDP over gloo) share the box with forked 2-rank helpers and BLAS:
# without a cap every pool spins up one thread per core and the suite can stall for minutes
for _var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_var, '2')

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# the suite chooses the arithmetic of the recurrent networks itself (both, see `arith` below)
os.environ.pop('PXG_LSTM_ARITH', None)

# Both arithmetics of a4 / a12 (include/pxg.h pxg_lstm_arith) are product paths: everything that
# depends on the config runs once per arithmetic, each against the goldens the REAL reference glue
# produced with the oracle's stand-in networks in that arithmetic (tools/make_golden.py --arith):
# tests/golden/ holds the f32 set, tests/golden/q8/ the files of the q8 set that differ from it.
ARITHS = ('q8', 'f32')


def G(name):
    """golden file of the arithmetic the running test is in (see _arith_env)"""
    return golden_path(os.environ.get('PXG_LSTM_ARITH', 'f32'), name)


def golden_path(arith, name):
    p = os.path.join(GOLDEN, 'q8', name)
    return p if arith == 'q8' and os.path.exists(p) else os.path.join(GOLDEN, name)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session', params=ARITHS)
def arith(request):
    return request.param


@pytest.fixture(scope='session')
def config(arith):
    from poreplex_amd.config import default_config
    cfg = default_config()
    cfg['signal_processing']['lstm_arith'] = arith
    return cfg


@pytest.fixture(autouse=True)
def _arith_env(request, monkeypatch):
    """PXG_LSTM_ARITH overrides the config's lstm_arith (native.NativeConfig), for the product and the
    oracle alike: a test that uses an arithmetic-dependent fixture runs with every config it builds in
    THAT arithmetic; every other test (host logic against the f32 golden set) is pinned to f32."""
    monkeypatch.setenv('PXG_LSTM_ARITH', request.getfixturevalue('arith') if 'arith' in request.fixturenames else 'f32')


@pytest.fixture(scope='session')
def golden(arith):
    """path of a golden file of this arithmetic's set"""
    return lambda name: golden_path(arith, name)


@pytest.fixture(scope='session')
def oracle(config):
    from oracle.pxo import Oracle
    return Oracle(config)


@pytest.fixture(scope='session')
def bundle(golden):
    return dict(np.load(golden('batch0.pxr.npz')))


@pytest.fixture(scope='session')
def stages(golden):
    return dict(np.load(golden('batch0.stages.npz')))


@pytest.fixture(scope='session')
def unit(golden):
    return dict(np.load(golden('unit.npz')))


@pytest.fixture(scope='session')
def ref_results(golden):
    with open(golden('batch0.results.json')) as fh:
        return json.load(fh)


@pytest.fixture(scope='session')
def ctx(config):
    """One GPU context for the whole -m gpu session (fails loudly, no fallback)."""
    from poreplex_amd.native import NativeContext
    c = NativeContext(config, device_id=0)
    yield c
    c.close()


def assert_spikes_equal(got, want_records, want_dense, msg=None):
    """The product's spike rows (rows [total, 4], offsets [n + 1]: every spike of every tail)
    against the oracle's dense per-read table."""
    from oracle.pxo import spikes_csr
    rows, off = got
    wrows, woff = spikes_csr(want_records, want_dense)
    assert np.array_equal(off, woff), msg
    assert np.array_equal(rows, wrows, equal_nan=True), msg
