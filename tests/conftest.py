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


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def config():
    from poreplex_amd.config import default_config
    return default_config()


@pytest.fixture(scope='session')
def oracle(config):
    from oracle.pxo import Oracle
    return Oracle(config)


@pytest.fixture(scope='session')
def bundle():
    return dict(np.load(os.path.join(GOLDEN, 'batch0.pxr.npz')))


@pytest.fixture(scope='session')
def stages():
    return dict(np.load(os.path.join(GOLDEN, 'batch0.stages.npz')))


@pytest.fixture(scope='session')
def unit():
    return dict(np.load(os.path.join(GOLDEN, 'unit.npz')))


@pytest.fixture(scope='session')
def ref_results():
    with open(os.path.join(GOLDEN, 'batch0.results.json')) as fh:
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
