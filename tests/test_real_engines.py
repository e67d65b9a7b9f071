"""The switch that lets somebody WITH TensorFlow / pomegranate close the a4 / a7 / a12 pin.

tools/make_golden.py runs the reference's own glue; the two third-party engines under it are the real
libraries where they import and oracle stand-ins where they do not (tools/golden_engines.py).  Here:
  * the switch itself, with fake "real" modules in sys.modules (CPU);
  * every committed golden set says what made it (engines.json);
  * the tolerance tests that bind the oracle (CPU) and the HIP path (-m gpu) to a set made with REAL engines
    (tests/golden/real/, written by `make_golden.py --engines real`): north_star's bounds -- softmax within
    1e-4, identical argmax / labels / phred / segment boundaries.  Without such a set they SKIP and say why:
    the image this repository was built in has neither library, so rows a4 / a7 / a12 stay "parity unpinned".
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import golden_engines as GE  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REAL_SET = os.path.join(GOLDEN, 'real')
SOFTMAX_TOL = 1e-4          # north_star: "softmax scores within 1e-4"
SCALER_TOL = 1e-4           # the scaler net's two outputs (standardised scale / shift), same bound


# ---- the switch -------------------------------------------------------------------------------------------
class _Recorder:
    def __init__(self):
        self.calls = []

    def keras(self):
        self.calls.append('keras-stub')

    def hmm(self):
        self.calls.append('hmm-stub')

    def wrap(self, tf):
        self.calls.append(('wrap', tf.__name__))


@pytest.fixture
def no_engines(monkeypatch):
    """an interpreter in which neither library imports (the build image)"""
    for name in list(sys.modules):
        if name == 'tensorflow' or name.startswith('tensorflow.') or name == 'pomegranate':
            monkeypatch.delitem(sys.modules, name)
    real_import = importlib.import_module

    def guarded(name, *a, **kw):
        if name in ('tensorflow', 'pomegranate'):
            raise ImportError('No module named ' + name)
        return real_import(name, *a, **kw)
    monkeypatch.setattr(GE.importlib, 'import_module', guarded)


@pytest.fixture
def fake_real_engines(monkeypatch):
    """modules that IMPORT as tensorflow / pomegranate (stand-ins for an installation that has them)"""
    tf = types.ModuleType('tensorflow')
    tf.keras = types.ModuleType('tensorflow.keras')
    tf.keras.models = types.SimpleNamespace(load_model=lambda p, **kw: ('real-model', p))
    monkeypatch.setitem(sys.modules, 'tensorflow', tf)
    monkeypatch.setitem(sys.modules, 'pomegranate', types.ModuleType('pomegranate'))
    return tf


def test_auto_uses_what_imports_and_stubs_only_on_import_error(no_engines):
    rec = _Recorder()
    assert GE.select('auto', rec.keras, rec.hmm, rec.wrap) == {'keras': GE.STUB, 'hmm': GE.STUB}
    assert rec.calls == ['keras-stub', 'hmm-stub']


def test_auto_with_real_engines_installs_no_stub(fake_real_engines):
    rec = _Recorder()
    engines = GE.select('auto', rec.keras, rec.hmm, rec.wrap)
    assert engines == {'keras': GE.REAL, 'hmm': GE.REAL} and GE.any_real(engines)
    assert rec.calls == [('wrap', 'tensorflow')]            # the prediction log hooks the REAL loader, nothing is replaced
    assert sys.modules['tensorflow'] is fake_real_engines


def test_one_real_engine_is_enough_for_its_rows(fake_real_engines, monkeypatch):
    monkeypatch.delitem(sys.modules, 'pomegranate')
    real_import = importlib.import_module
    monkeypatch.setattr(GE.importlib, 'import_module',
                        lambda name, *a, **kw: (_ for _ in ()).throw(ImportError(name)) if name == 'pomegranate' else real_import(name, *a, **kw))
    rec = _Recorder()
    assert GE.select('auto', rec.keras, rec.hmm, rec.wrap) == {'keras': GE.REAL, 'hmm': GE.STUB}
    assert rec.calls == [('wrap', 'tensorflow'), 'hmm-stub']


def test_real_mode_refuses_to_fall_back(no_engines):
    rec = _Recorder()
    with pytest.raises(SystemExit) as e:
        GE.select('real', rec.keras, rec.hmm, rec.wrap)
    assert 'import tensorflow' in str(e.value) and rec.calls == []


def test_stub_mode_ignores_installed_engines(fake_real_engines):
    rec = _Recorder()
    assert GE.select('stub', rec.keras, rec.hmm, rec.wrap) == {'keras': GE.STUB, 'hmm': GE.STUB}
    assert rec.calls == ['keras-stub', 'hmm-stub']


def test_mode_flag_and_metadata_round_trip(tmp_path):
    assert GE.parse_mode(['x']) == 'auto' and GE.parse_mode(['x', '--engines', 'real']) == 'real'
    with pytest.raises(SystemExit):
        GE.parse_mode(['x', '--engines', 'maybe'])
    GE.write(str(tmp_path), {'keras': GE.REAL, 'hmm': GE.STUB}, {'arith': 'q8'})
    assert GE.read(str(tmp_path)) == {'keras': GE.REAL, 'hmm': GE.STUB}
    assert GE.read(str(tmp_path / 'nothing-here')) is None
    doc = json.load(open(tmp_path / 'engines.json'))
    assert doc['arith'] == 'q8' and set(doc['meaning']) == {'keras', 'hmm'}


def test_make_golden_goes_through_the_switch():
    """make_golden.py has no unconditional stand-ins left: both installers are reached through GE.select only."""
    src = open(os.path.join(ROOT, 'tools', 'make_golden.py')).read()
    assert 'GE.select(ENGINE_MODE, install_keras_stub, install_hmm_stub, wrap_real_keras)' in src
    assert src.count('install_keras_stub') == 2 and src.count('install_hmm_stub') == 2      # definition + the switch
    assert 'GE.write(OUT, ENGINES' in src


# ---- what made the committed sets -------------------------------------------------------------------------
@pytest.mark.parametrize('sub', ['', 'q8'])
def test_committed_sets_say_what_made_them(sub):
    engines = GE.read(os.path.join(GOLDEN, sub))
    assert engines is not None, 'engines.json missing: regenerate with tools/make_golden.py'
    assert set(engines) == {'keras', 'hmm'} and set(engines.values()) <= {GE.REAL, GE.STUB}
    # the bit-exact sets are stand-in sets by construction (a real-engine set lives in tests/golden/real/)
    assert engines == {'keras': GE.STUB, 'hmm': GE.STUB}


# ---- tolerance tests against a set made with real engines -------------------------------------------------
def _real(engine):
    """(directory of the real-engine set of the running arithmetic) or skip with the reason"""
    arith = os.environ.get('PXG_LSTM_ARITH', 'f32')
    d = os.path.join(REAL_SET, 'q8') if arith == 'q8' and os.path.isdir(os.path.join(REAL_SET, 'q8')) else REAL_SET
    engines = GE.read(d)
    if engines is None:
        pytest.skip('no tests/golden/real/: the committed goldens were generated with oracle stand-ins (TensorFlow / '
                    'pomegranate are not in the build image); run tools/make_golden.py --engines real where they are')
    if engines.get(engine) != GE.REAL:
        pytest.skip('tests/golden/real/ was generated with an oracle stand-in for `{}` ({})'.format(engine, engines))
    return d


def _stages(d):
    return dict(np.load(os.path.join(d, 'batch0.stages.npz')))


def test_oracle_scaler_net_within_tolerance_of_real_keras(oracle):
    s = _stages(_real('keras'))
    got = np.stack([oracle.scaler_forward(r) for r in s['scaler_in']])
    assert np.abs(got - s['scaler_out']).max() <= SCALER_TOL


def test_oracle_demux_net_within_tolerance_of_real_keras(oracle):
    s = _stages(_real('keras'))
    got = np.stack([oracle.demux_forward(r) for r in s['demux_in']])
    want = s['demux_out'][:, :got.shape[1]]
    assert np.abs(got - want).max() <= SOFTMAX_TOL
    assert np.array_equal(got.argmax(1), want.argmax(1))


def test_oracle_viterbi_segments_identical_to_real_pomegranate(oracle):
    s = _stages(_real('hmm'))
    po = s['pooled_offsets']
    scan = oracle.cfg.segmentation_scan_limit // oracle.cfg.stride
    n = 0
    for i in range(len(po) - 1):
        if not s['has_seg'][i]:
            continue
        _, path = oracle.viterbi(s['pooled_arena'][po[i]:po[i + 1]][:scan])
        first, last = oracle.segments(path)
        assert np.array_equal(first, s['seg_first'][i]) and np.array_equal(last, s['seg_last'][i]), i
        n += 1
    assert n >= 18


@pytest.mark.gpu
def test_hip_networks_within_tolerance_of_real_keras(ctx):
    s = _stages(_real('keras'))
    assert np.abs(ctx.scaler_lstm(s['scaler_in']) - s['scaler_out']).max() <= SCALER_TOL
    got = ctx.demux_lstm(s['demux_in'])
    want = s['demux_out'][:, :got.shape[1]]
    assert np.abs(got - want).max() <= SOFTMAX_TOL and np.array_equal(got.argmax(1), want.argmax(1))


@pytest.mark.gpu
def test_hip_segments_identical_to_real_pomegranate(ctx):
    s = _stages(_real('hmm'))
    po = s['pooled_offsets']
    scan = ctx.cfg.segmentation_scan_limit // ctx.cfg.stride
    keep = [i for i in range(len(po) - 1) if s['has_seg'][i]]
    segs = ctx.viterbi([s['pooled_arena'][po[i]:po[i + 1]][:scan] for i in keep])
    first, last = segs[0], segs[1]
    for k, i in enumerate(keep):
        assert np.array_equal(first[k][:len(s['seg_first'][i])], s['seg_first'][i][:len(first[k])]), i
        assert np.array_equal(last[k][:len(s['seg_last'][i])], s['seg_last'][i][:len(last[k])]), i


@pytest.mark.gpu
def test_hip_result_records_agree_with_real_engine_results(ctx, config):
    """labels / barcodes / statuses / phred identical, float fields within tolerance, on the 32-read golden batch."""
    d = _real('keras')
    _real('hmm')
    from poreplex_amd import native as N
    b = dict(np.load(os.path.join(d, 'batch0.pxr.npz')))
    with open(os.path.join(d, 'batch0.results.json')) as fh:
        ref = {r['read_id']: r for r in json.load(fh)['results'] if 'read_id' in r}
    got = ctx.process_batch(b['arena'], b['offsets'], b['calib'])
    ids = [str(x) for x in b['read_id']]
    checked = 0
    for i, rid in enumerate(ids):
        want = ref.get(rid)
        if want is None or want.get('status') != 'okay':
            continue
        assert N.STATUS_NAMES[int(got['status'][i])] == 'okay', rid
        if 'barcode_guess' in want:
            assert int(got['bc_label'][i]) == int(want['barcode_guess']), rid
            assert int(got['bc_phred'][i]) == int(want['barcode_score']), rid
            assert bool(got['bc_called'][i]) == (want.get('barcode') is not None), rid
        checked += 1
    assert checked >= 10


def test_the_tolerance_tests_run_on_a_set_that_says_real(tmp_path, monkeypatch, oracle, arith):
    """The tolerance tests above are not dead code: pointed at a copy of the committed stage fixtures that CLAIMS real
    engines (the oracle made it, so the tolerances are met with zero error), every CPU one of them runs to its end."""
    import shutil
    import tests.conftest as CT
    shutil.copy(CT.golden_path(arith, 'batch0.stages.npz'), tmp_path / 'batch0.stages.npz')
    GE.write(str(tmp_path), {'keras': GE.REAL, 'hmm': GE.REAL}, {'note': 'self-test copy'})
    monkeypatch.setattr(sys.modules[__name__], 'REAL_SET', str(tmp_path))
    test_oracle_scaler_net_within_tolerance_of_real_keras(oracle)
    test_oracle_demux_net_within_tolerance_of_real_keras(oracle)
    test_oracle_viterbi_segments_identical_to_real_pomegranate(oracle)
    GE.write(str(tmp_path), {'keras': GE.STUB, 'hmm': GE.REAL})
    with pytest.raises(pytest.skip.Exception, match='stand-in for `keras`'):
        test_oracle_scaler_net_within_tolerance_of_real_keras(oracle)
