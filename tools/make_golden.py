#!/opt/conda/bin/python3.9
"""Generate tests/golden/* by RUNNING THE REFERENCE in the build container.

Must run under /opt/conda/bin/python3.9 (NumPy 1.26 legacy casting is part of
the reference's numerics; h5py only exists there).  Nothing
here ships to the GPU box: only the arrays / JSON it writes do.

What is real reference code here:
  * poreplex/{signal_analyzer,signal_loader,barcoding,polya,fast5_file,
    worker_persistence,utils}.py imported from /root/reference through a
    shadow package of symlinks (the tree is read-only);
  * poreplex.csupport compiled from /root/reference/src (setup.py:34-37 flags).
What is NOT available in this image (no TensorFlow / pomegranate, no network):
the two third-party engines.  `--engines auto` (default) tries to import them
and replaces ONLY what raises ImportError by a stub that calls the ORACLE
restatement (oracle/libpxo.so) -- Keras `model.predict` and pomegranate
`viterbi`; `--engines real` insists on the libraries, `--engines stub` on the
stand-ins (tools/golden_engines.py).  A set made with a real engine goes to
tests/golden/real/ (tolerance tests: tests/test_real_engines.py); every set
records what made it in engines.json.  With the stand-ins
these goldens pin everything the reference itself computes (pA conversion,
pooling, padding, de-standardisation + QC, run-length summary, window rules,
robust z-score, thresholds, phred lookup, poly(A) logic, status/label logic,
result-dict schema and ordering) and leave a4/a12 forward passes and a7
Viterbi "parity unpinned" (DESIGN.md).
"""
import json
import os
import subprocess
import sys
import sysconfig
import tempfile
import types
import warnings

warnings.filterwarnings('ignore')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REPO)

import h5py  # noqa: E402
import numpy as np  # noqa: E402

assert np.__version__.startswith('1.'), 'run with /opt/conda/bin/python3.9 (NumPy 1.x)'

from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.synth import synth_batch  # noqa: E402
from oracle.pxo import Oracle, reference_detect_events  # noqa: E402
sys.path.insert(0, os.path.join(REPO, 'tools'))
import golden_engines as GE  # noqa: E402

# The reference targets h5py 2.x, where string attributes come back as bytes
# (it calls .decode() on them: fast5_file.py:102-120, signal_loader.py:55-58).
# h5py 3.x returns str for variable-length strings; emulate the 2.x behaviour.
_attr_get = h5py.AttributeManager.__getitem__


def _attr_get_bytes(self, name):
    v = _attr_get(self, name)
    return v.encode() if isinstance(v, str) else v


h5py.AttributeManager.__getitem__ = _attr_get_bytes

# --arith f32|q8: the arithmetic of the ORACLE's stand-in networks (include/pxg.h pxg_lstm_arith).  The
# reference's glue is the same either way; what changes is the last digits of (scale, shift) and of the
# softmax the stand-ins hand to it.  f32 -> tests/golden/ (rounds 1-3), q8 -> tests/golden/q8/.
ARITH = sys.argv[sys.argv.index('--arith') + 1] if '--arith' in sys.argv else 'f32'
assert ARITH in ('f32', 'q8')
os.environ['PXG_LSTM_ARITH'] = ARITH
OUT = os.path.join(REPO, 'tests', 'golden') if ARITH == 'f32' else os.path.join(REPO, 'tests', 'golden', 'q8')
ENGINE_MODE = GE.parse_mode(sys.argv)
ENGINES = None      # set by install_engines(): {'keras': 'real' | 'oracle-stub', 'hmm': ...}
# `make_golden.py --only batch0` rewrites just the fixtures whose name starts with that prefix.
# Every fixture is an (inputs, outputs-of-the-real-reference) pair that carries its own inputs,
# so sets made by different revisions of the synthetic generator can coexist: unit / polya /
# chimera date from before the generator planted barcode prototypes, batch0 was redone after
# (32 reads, six of them at the bench's ~60 000-sample shape).
ONLY = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else ''
if ONLY:
    _scratch = os.path.join(__import__('tempfile').mkdtemp(prefix='pxg_golden_'), 'all')
    os.makedirs(_scratch)
    _FINAL, OUT = OUT, _scratch
os.makedirs(OUT, exist_ok=True)
TMP = tempfile.mkdtemp(prefix='pxgold')


# --------------------------------------------------------------------------
# shadow package + compiled csupport
# --------------------------------------------------------------------------
def build_shadow():
    pk = os.path.join(TMP, 'poreplex')
    os.makedirs(pk)
    for f in os.listdir(REF + '/poreplex'):
        if f.endswith('.py'):
            os.symlink(os.path.join(REF, 'poreplex', f), os.path.join(pk, f))
    os.symlink(REF + '/poreplex/presets', os.path.join(pk, 'presets'))
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    subprocess.check_call([
        'gcc', '-std=c99', '-O2', '-fPIC', '-shared',
        '-I' + sysconfig.get_paths()['include'], '-I' + np.get_include(),
        '-I' + REF + '/src/contrib/scrappie',
        REF + '/src/csupport.c', REF + '/src/contrib/scrappie/event_detection.c',
        '-lm', '-o', os.path.join(pk, 'csupport' + ext)],
        stderr=subprocess.DEVNULL)
    sys.path.insert(0, TMP)


# --------------------------------------------------------------------------
# third-party stubs backed by the oracle
# --------------------------------------------------------------------------
ORACLE = None  # set in main()
PREDICT_LOG = []


class _State:
    def __init__(self, dist, name=None):
        self.distribution, self.name = dist, name


class _Normal:
    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma


class _GMM:
    def __init__(self, dists, weights=None):
        self.dists, self.weights = dists, weights


class _HMM:
    """pomegranate.HiddenMarkovModel facade: same construction calls as
    worker_persistence.py:95-121; viterbi() = the oracle's restatement."""

    def __init__(self, name):
        self.name = name
        self.start = _State(None, name + '-start')
        self.states, self.edges, self.starts = [], [], {}

    def add_state(self, s):
        self.states.append(s)

    def add_transition(self, a, b, p):
        if a is self.start:
            self.starts[b.name] = p
        else:
            self.edges.append((a.name, b.name, p))

    def bake(self):
        md = []
        for s in self.states:
            d = s.distribution
            if isinstance(d, _Normal):
                em = [[d.mu, d.sigma]]
            else:
                em = [[c.mu, c.sigma, float(w)] for c, w in zip(d.dists, d.weights)]
            row = {'name': s.name, 'emission': em,
                   'transition': [[b, p] for a, b, p in self.edges if a == s.name]}
            if s.name in self.starts:
                row['start_prob'] = self.starts[s.name]
            md.append(row)
        self.hmm = N.PxgHmm()
        N._fill_hmm(self.hmm, md)

    def viterbi(self, seq):
        import ctypes as C
        x = np.ascontiguousarray(np.asarray(seq), dtype=np.float32)
        path = np.zeros(len(x), dtype=np.int32)
        logp = ORACLE.L.pxo_viterbi(C.byref(self.hmm), x.ctypes.data_as(C.c_void_p),
                                    len(x), path.ctypes.data_as(C.c_void_p))
        return logp, [(len(self.states), self.start)] + \
            [(int(i), self.states[int(i)]) for i in path]


class _FakeKerasModel:
    def __init__(self, path):
        self.kind = 'scaler' if 'scaler' in os.path.basename(path) else 'demux'

    def predict(self, x, batch_size=None, verbose=0):
        x = np.asarray(x, dtype=np.float32)[:, :, 0]
        fn = ORACLE.scaler_forward if self.kind == 'scaler' else ORACLE.demux_forward
        out = np.stack([fn(row) for row in x])
        PREDICT_LOG.append((self.kind, x.copy(), out.copy()))
        return out


def install_hmm_stub():
    pom = types.ModuleType('pomegranate')
    pom.HiddenMarkovModel, pom.GeneralMixtureModel = _HMM, _GMM
    pom.State, pom.NormalDistribution = _State, _Normal
    sys.modules['pomegranate'] = pom


def install_keras_stub():
    tf = types.ModuleType('tensorflow')
    tf.get_logger = lambda: types.SimpleNamespace(setLevel=lambda *_: None)
    keras = types.ModuleType('tensorflow.keras')
    custom = {}
    keras.models = types.SimpleNamespace(load_model=lambda p, **kw: _FakeKerasModel(p))
    keras.utils = types.SimpleNamespace(get_custom_objects=lambda: custom)
    backend = types.ModuleType('tensorflow.keras.backend')
    losses = types.ModuleType('tensorflow.keras.losses')
    metrics = types.ModuleType('tensorflow.keras.metrics')
    losses.CategoricalCrossentropy = type('CategoricalCrossentropy', (), {})
    metrics.CategoricalAccuracy = type('CategoricalAccuracy', (), {})
    keras.backend, keras.losses, keras.metrics = backend, losses, metrics
    tf.keras = keras
    sys.modules.update({'tensorflow': tf, 'tensorflow.keras': keras,
                        'tensorflow.keras.backend': backend,
                        'tensorflow.keras.losses': losses,
                        'tensorflow.keras.metrics': metrics})


class _LoggedKerasModel:
    """A REAL Keras model whose predict() calls are logged like the stand-in's (the stage fixtures are the
    inputs / outputs of those calls)."""

    def __init__(self, model, path):
        self._model = model
        self.kind = 'scaler' if 'scaler' in os.path.basename(path) else 'demux'

    def __getattr__(self, name):
        return getattr(self._model, name)

    def predict(self, x, *a, **kw):
        out = np.asarray(self._model.predict(x, *a, **kw), dtype=np.float32)
        PREDICT_LOG.append((self.kind, np.asarray(x, dtype=np.float32)[:, :, 0].copy(), out.copy()))
        return out


def wrap_real_keras(tf):
    real_load = tf.keras.models.load_model
    tf.keras.models.load_model = lambda p, *a, **kw: _LoggedKerasModel(real_load(p, *a, **kw), p)


def install_engines():
    """tools/golden_engines.py: real TensorFlow / pomegranate where they import, oracle stand-ins where they do not."""
    global ENGINES, OUT
    ENGINES = GE.select(ENGINE_MODE, install_keras_stub, install_hmm_stub, wrap_real_keras)
    if GE.any_real(ENGINES):
        # a set with a real engine never replaces the bit-exact sets: tolerance tests read it from its own place
        if ONLY:
            raise SystemExit('--only rewrites fixtures of the bit-exact sets: use --engines stub with it')
        OUT = os.path.join(REPO, 'tests', 'golden', 'real') if ARITH == 'f32' else os.path.join(REPO, 'tests', 'golden', 'real', 'q8')
        os.makedirs(OUT, exist_ok=True)
    print('engines:', ENGINES, '->', OUT, file=sys.stderr)


# --------------------------------------------------------------------------
# synthetic FAST5 (SURVEY App. B layout)
# --------------------------------------------------------------------------
def write_fast5(path, read_id, raw, cal, meta, basecall=None):
    with h5py.File(path, 'w') as h5:
        rd = h5.create_group('Raw/Reads/Read_{}'.format(meta['read_number']))
        rd.attrs['duration'] = np.uint32(len(raw))
        rd.attrs['start_time'] = np.uint64(meta['start_time'])
        rd.attrs['read_id'] = read_id.encode()
        rd.attrs['read_number'] = np.int32(meta['read_number'])
        rd.create_dataset('Signal', data=np.asarray(raw, dtype=np.int16))
        ch = h5.create_group('UniqueGlobalKey/channel_id')
        ch.attrs['channel_number'] = str(meta['channel_number']).encode()
        ch.attrs['digitisation'] = float(cal['digitisation'])
        ch.attrs['offset'] = float(cal['offset'])
        ch.attrs['range'] = float(cal['range'])
        ch.attrs['sampling_rate'] = float(cal['sampling_rate'])
        tr = h5.create_group('UniqueGlobalKey/tracking_id')
        tr.attrs['run_id'] = meta['run_id'].encode()
        tr.attrs['sample_id'] = meta['sample_id'].encode()
        if basecall is not None:
            bc = h5.create_group('Analyses/Basecall_1D_000')
            tpl = bc.create_group('BaseCalled_template')
            fq = '@{}\n{}\n+\n{}\n'.format(read_id, basecall['sequence'], basecall['qstring'])
            tpl.create_dataset('Fastq', data=np.string_(fq))
            tpl.create_dataset('Move', data=np.asarray(basecall['move'], dtype=np.uint8))
            sm = bc.create_group('Summary/basecall_1d_template')
            sm.attrs['sequence_length'] = np.int32(basecall['sequence_length'])
            sm.attrs['mean_qscore'] = np.float32(basecall['mean_qscore'])
            sm.attrs['block_stride'] = np.int32(basecall['block_stride'])
            sg = h5.create_group('Analyses/Segmentation_000/Summary/segmentation')
            sg.attrs['num_events_template'] = np.int32(basecall['num_events'])
            sg.attrs['first_sample_template'] = np.int32(basecall['first_sample_template'])


def make_basecall(rng, n_raw, first_sample, seq_len=None):
    stride = 15
    n_blocks = (n_raw - first_sample) // stride
    if seq_len is None:
        seq_len = max(12, n_blocks // 9)
    seq_len = min(seq_len, n_blocks)
    move = np.zeros(n_blocks, dtype=np.uint8)
    move[np.sort(rng.choice(n_blocks, seq_len, replace=False))] = 1
    move[0] = 1 if move.sum() < seq_len or True else move[0]
    seq_len = int(move.sum())
    seq = ''.join(rng.choice(list('ACGU'), seq_len))
    qs = ''.join(chr(33 + int(q)) for q in rng.integers(3, 25, seq_len))
    return {'sequence': seq, 'qstring': qs, 'move': move.tolist(),
            'sequence_length': seq_len, 'mean_qscore': float(np.round(rng.uniform(7, 12), 3)),
            'block_stride': stride, 'num_events': int(n_blocks),
            'first_sample_template': int(first_sample)}


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, np.ndarray):
        return jsonable(o.tolist())
    if isinstance(o, bytes):
        return o.decode()
    return o


# --------------------------------------------------------------------------
def build_read_set(rng):
    """Reads engineered to hit every status / branch the reference has."""
    reads = []

    def add(batch, i, tag, basecall='auto', seq_len=None):
        a, o = batch['arena'], batch['offsets']
        reads.append({'tag': tag, 'raw': a[o[i]:o[i + 1]].copy(), 'cal': batch['calib'][i],
                      'ss': batch['scale_shift'][i], 'barcode': int(batch['barcode'][i]),
                      'basecall': basecall, 'seq_len': seq_len})

    b = synth_batch(10, seed=9220, samples_per_read=30000, jitter=0.3)
    for i in range(10):
        add(b, i, 'normal%d' % i, basecall='auto' if i % 3 else None)
    b = synth_batch(2, seed=9221, samples_per_read=62000, jitter=0.02)
    add(b, 0, 'long0')
    add(b, 1, 'long1_shortseq', seq_len=6)
    b = synth_batch(2, seed=9222, samples_per_read=26000, with_polya=False)
    add(b, 0, 'nopolya0')
    add(b, 1, 'nopolya1', basecall=None)
    b = synth_batch(2, seed=9223, samples_per_read=5000, jitter=0.0)
    add(b, 0, 'tooshort')
    # adapter shorter than the 260 gate and between 260 and 300 (left pad)
    from poreplex_amd import synth as S
    keep = dict(S._PIECE_LEN)
    S._PIECE_LEN['adapter'] = (3000, 3300)
    b = synth_batch(1, seed=9224, samples_per_read=24000)
    add(b, 0, 'adapter_lt260')
    S._PIECE_LEN['adapter'] = (4050, 4350)
    b = synth_batch(2, seed=9225, samples_per_read=24000)
    add(b, 0, 'adapter_pad0')
    add(b, 1, 'adapter_pad1', basecall=None)
    # long poly(A) (open-end extension) and shifted poly(A) level (recalibration)
    S._PIECE_LEN.update(keep)
    S._PIECE_LEN['polya-tail'] = (5000, 7000)
    b = synth_batch(2, seed=9226, samples_per_read=40000)
    add(b, 0, 'polya_long0')
    add(b, 1, 'polya_long1')
    S._PIECE_LEN.update(keep)
    emit = dict(S._EMIT)
    S._EMIT['polya-tail'] = [(114.9, 2.0, 1.0)]
    b = synth_batch(2, seed=9227, samples_per_read=30000)
    add(b, 0, 'polya_shift0')
    add(b, 1, 'polya_shift1')
    S._EMIT.update(emit)
    # no adapter at all: flat squiggles at a leader level; keep the first
    # candidate the (oracle-backed) pipeline reports as adapter_not_detected
    cal = np.zeros(1, N.CALIB_DTYPE)[0]
    cal['range'], cal['digitisation'], cal['offset'], cal['sampling_rate'] = 1200., 8192., 10., 3012.
    found = False
    for level in (102.07, 112.02, 71.5, 95.0):
        for sc, sh in ((0.95, -5.0), (0.85, 5.0), (1.05, -12.0), (0.75, 10.0)):
            flat = rng.normal(level, 2.0, 20000).astype(np.float32)
            raw = np.rint((flat - sh) / sc * 8192 / 1200 - 10).astype(np.int16)
            res = ORACLE.process_batch(raw, np.array([0, len(raw)]), np.array([tuple(cal)], N.CALIB_DTYPE))
            if res[0]['status'] == N.STATUS_CODE['adapter_not_detected']:
                reads.append({'tag': 'noadapter', 'raw': raw, 'cal': cal,
                              'ss': np.float32([sc, sh]), 'barcode': -1,
                              'basecall': None, 'seq_len': None})
                found = True
                break
        if found:
            break
    assert found, 'no adapter_not_detected candidate'
    # scaling QC failure candidates: wildly mis-scaled copies of a normal read
    base = reads[1]
    for k, (mul, add_) in enumerate([(2.6, 0), (0.25, 300), (1.0, 900)]):
        raw = np.clip(base['raw'].astype(np.float64) * mul + add_, -32768, 32767).astype(np.int16)
        reads.append({'tag': 'misscaled%d' % k, 'raw': raw, 'cal': base['cal'], 'ss': base['ss'],
                      'barcode': base['barcode'], 'basecall': None, 'seq_len': None})
    # configs[0] is "32 reads": six more reads at the bench's own shape (~60 000 samples)
    b = synth_batch(6, seed=9230, samples_per_read=61000, jitter=0.05)
    for i in range(6):
        add(b, i, 'bench60k_%d' % i, basecall='auto' if i != 4 else None)
    return reads


def main():
    global ORACLE
    build_shadow()
    install_engines()
    cfg_mine = default_config()
    ORACLE = Oracle(cfg_mine)

    import yaml
    import pandas as pd
    from poreplex import signal_analyzer as SA
    from poreplex import signal_loader as SL
    from poreplex import barcoding as BC
    from poreplex import polya as PA
    from poreplex import utils as UT
    from poreplex.csupport import detect_events as ref_detect_events

    rng = np.random.default_rng(20260928)
    inputdir = os.path.join(TMP, 'in')
    os.makedirs(inputdir)
    reads = build_read_set(rng)
    items = []
    rng_extra = np.random.default_rng(9230)     # reads added later draw from their own stream, so
    for i, r in enumerate(reads):               # every other fixture stays bit-identical
        rid = '%08x-0000-4000-8000-%012x' % (0x9220 + i, i)
        fn = 'r%03d.fast5' % i
        g = rng_extra if r['tag'].startswith('bench60k') else rng
        meta = {'read_number': 100 + i, 'start_time': int(g.integers(10**5, 10**8)),
                'channel_number': int(g.integers(1, 513)), 'run_id': 'run' + 'ab' * 19,
                'sample_id': 'synthetic'}
        bc = None
        if r['basecall'] == 'auto' and len(r['raw']) > 12000:
            first = int(g.integers(0, 40))
            bc = make_basecall(g, len(r['raw']), first, r['seq_len'])
        write_fast5(os.path.join(inputdir, fn), rid, r['raw'], r['cal'], meta, bc)
        items.append({**r, 'filename': fn, 'read_id': rid, 'meta': meta, 'basecall': bc})
    # a corrupt file and a vanished file
    with open(os.path.join(inputdir, 'broken.fast5'), 'wb') as fh:
        fh.write(b'this is not an HDF5 file')
    batch_reads = [(it['filename'], it['read_id']) for it in items]
    batch_reads.insert(5, ('broken.fast5', 'deadbeef-0000-4000-8000-000000000000'))
    batch_reads.insert(9, ('vanished.fast5', 'deadbeef-0000-4000-8000-000000000001'))

    kmer_path = os.path.join(TMP, 'kmer.model')
    with open(kmer_path, 'w') as fh:
        fh.write('kmer\tlevel_mean\tlevel_stdv\n')
        for k in ('AAAAA', 'AAAAC', 'AAAAG'):
            fh.write('%s\t80.0\t2.0\n' % k)

    refcfg = yaml.safe_load(open(REF + '/poreplex/presets/rna-r941.cfg'))
    refcfg.update({
        'inputdir': inputdir, 'outputdir': os.path.join(TMP, 'out'),
        'barcoding': True, 'measure_polya': True, 'trim_adapter': True,
        'filter_unsplit_reads': False, 'minimum_sequence_length': 10,
        'albacore_onthefly': False, 'dump_adapter_signals': False, 'dump_basecalls': False,
        'barcoding_quality_filter': 18, 'kmer_model': kmer_path,
    })

    # ---- instrument the reference to capture stage outputs ----------------
    cap = {'head': {}, 'pooled': {}, 'segments': {}, 'window': {}, 'pa64': {}, 'polya': {}}
    orig_head = SL.NanoporeRead.load_padded_signal_head
    orig_sig = SL.NanoporeRead.load_signal
    orig_seg = SA.SignalAnalysis.detect_segments
    orig_push = BC.BarcodeDemultiplexer.push

    def head(self, *a):
        out = orig_head(self, *a)
        cap['head'][self.read_id] = None if out is None else np.array(out)
        cap['pa64'][self.read_id] = np.array(self.fast5.get_raw_data(end=64))
        return out

    def sig(self, end=None, pool=None, pad=False, scale=True):
        out = orig_sig(self, end=end, pool=pool, pad=pad, scale=scale)
        if pool == 15 and scale:
            cap['pooled'][self.read_id] = np.array(out)
        return out

    def seg(self, signal, elspan):
        out = orig_seg(self, signal, elspan)
        cap['segments'][self.npread.read_id] = dict(out)
        return out

    def push(self, npread, signal):
        n0 = len(self.signals)
        orig_push(self, npread, signal)
        if len(self.signals) > n0:
            cap['window'][npread.read_id] = np.array(self.signals[-1])

    SL.NanoporeRead.load_padded_signal_head = head
    SL.NanoporeRead.load_signal = sig
    SA.SignalAnalysis.detect_segments = seg
    BC.BarcodeDemultiplexer.push = push

    sys.stderr, saved = open(os.devnull, 'w'), sys.stderr   # h5py traceback of broken.fast5
    try:
        results = SA.process_batch(7, batch_reads, refcfg)
    finally:
        sys.stderr = saved
    assert not (isinstance(results, tuple) and results[0] == -1), results
    SL.NanoporeRead.load_padded_signal_head = orig_head
    SL.NanoporeRead.load_signal = orig_sig
    SA.SignalAnalysis.detect_segments = orig_seg
    BC.BarcodeDemultiplexer.push = orig_push
    statuses = {}
    for r in results:
        statuses[r['status']] = statuses.get(r['status'], 0) + 1
    print('reference statuses:', statuses)
    print('labels:', {l: sum(1 for r in results if r.get('label') == l)
                      for l in ('pass', 'fail', 'artifact', None)})
    print('barcodes:', [r.get('barcode') for r in results])
    print('polya called:', sum(1 for r in results if 'polya' in r))

    # the same batch with the TRUE (scale, shift) injected instead of the
    # scaler network: pins stage-B results independently of the LSTM
    n = len(items)
    arena, offsets = N.pack_reads([it['raw'] for it in items])
    calib = np.array([tuple(it['cal']) for it in items], dtype=N.CALIB_DTYPE)

    # ---- save the read bundle (what the GPU-side tests feed the facade) ---
    np.savez_compressed(
        os.path.join(OUT, 'batch0.pxr.npz'),
        arena=arena, offsets=offsets, calib=calib,
        filename=np.array([it['filename'] for it in items]),
        read_id=np.array([it['read_id'] for it in items]),
        duration=np.array([len(it['raw']) for it in items], dtype=np.int64),
        start_time=np.array([it['meta']['start_time'] for it in items], dtype=np.int64),
        channel_number=np.array([str(it['meta']['channel_number']) for it in items]),
        run_id=np.array([it['meta']['run_id'] for it in items]),
        sample_id=np.array([it['meta']['sample_id'] for it in items]),
        basecall=np.array([json.dumps(it['basecall']) if it['basecall'] else '' for it in items]),
        true_scale_shift=np.array([it['ss'] for it in items], dtype=np.float32),
        true_barcode=np.array([it['barcode'] for it in items], dtype=np.int8),
        tag=np.array([it['tag'] for it in items]),
        broken_files=np.array(['broken.fast5']),   # present on disk but not HDF5
    )

    # ---- per-stage captures ------------------------------------------------
    nh = 2000
    head_arr = np.zeros((n, nh), np.float32)
    head_ok = np.zeros(n, np.int8)
    pa64 = np.zeros((n, 64), np.float32)
    win = np.zeros((n, 300), np.float32)
    pushed = np.zeros(n, np.int8)
    seg_first = -np.ones((n, 8), np.int32)
    seg_last = -np.ones((n, 8), np.int32)
    has_seg = np.zeros(n, np.int8)
    pooled_list = []
    names = ORACLE.state_names
    for i, it in enumerate(items):
        rid = it['read_id']
        if cap['head'].get(rid) is not None:
            head_arr[i], head_ok[i] = cap['head'][rid], 1
        if rid in cap['pa64']:
            pa64[i, :len(cap['pa64'][rid])] = cap['pa64'][rid]
        if rid in cap['window']:
            win[i], pushed[i] = cap['window'][rid], 1
        if rid in cap['segments']:
            has_seg[i] = 1
            for name, (a, b) in cap['segments'][rid].items():
                seg_first[i, names.index(name)] = a
                seg_last[i, names.index(name)] = b
        pooled_list.append(cap['pooled'].get(rid, np.zeros(0, np.float32)))
    parena, poff = np.concatenate(pooled_list), np.zeros(n + 1, np.int64)
    poff[1:] = np.cumsum([len(p) for p in pooled_list])
    scaler_in = [x for k, x, _ in PREDICT_LOG if k == 'scaler']
    scaler_out = [y for k, _, y in PREDICT_LOG if k == 'scaler']
    demux_in = [x for k, x, _ in PREDICT_LOG if k == 'demux']
    demux_out = [y for k, _, y in PREDICT_LOG if k == 'demux']
    np.savez_compressed(
        os.path.join(OUT, 'batch0.stages.npz'),
        head=head_arr, head_ok=head_ok, pa64=pa64, window=win, pushed=pushed,
        seg_first=seg_first, seg_last=seg_last, has_seg=has_seg,
        pooled_arena=parena.astype(np.float32), pooled_offsets=poff,
        scaler_in=np.concatenate(scaler_in), scaler_out=np.concatenate(scaler_out),
        demux_in=np.concatenate(demux_in) if demux_in else np.zeros((0, 300), np.float32),
        demux_out=np.concatenate(demux_out) if demux_out else np.zeros((0, 5), np.float32),
    )
    with open(os.path.join(OUT, 'batch0.results.json'), 'w') as fh:
        json.dump({'batchid': 7, 'reads': batch_reads,
                   'config_flags': {k: refcfg[k] for k in (
                       'barcoding', 'measure_polya', 'trim_adapter', 'filter_unsplit_reads',
                       'minimum_sequence_length', 'barcoding_quality_filter')},
                   'results': jsonable(results)}, fh, indent=1)

    # ---- the same batch with both HDF5 dump options on (signal_analyzer.py:155-211,450-466) ----
    dump_out = os.path.join(TMP, 'out_dumps')
    for sub in ('adapter-dumps', 'events'):
        os.makedirs(os.path.join(dump_out, sub))
    sys.stderr, saved = open(os.devnull, 'w'), sys.stderr
    try:
        dres = SA.process_batch(7, batch_reads, dict(refcfg, outputdir=dump_out, dump_adapter_signals=True,
                                                     dump_basecalls=True))
    finally:
        sys.stderr = saved
    assert [r['status'] for r in dres] == [r['status'] for r in results]
    import glob
    dumps = {}
    (adapter_part,) = glob.glob(os.path.join(dump_out, 'adapter-dumps', 'part-*.h5'))
    with h5py.File(adapter_part, 'r') as h5:
        assert list(h5['adapter']) == ['00000007'] and list(h5['catalog/adapter']) == ['00000007']
        dumps['adapter_catalog'] = h5['catalog/adapter/00000007'][:]
        ids = sorted(h5['adapter/00000007'])
        sigs = [h5['adapter/00000007/' + k][:] for k in ids]
        assert all(x.dtype == np.float32 for x in sigs)
        dumps['adapter_ids'] = np.array(ids)
        dumps['adapter_offsets'] = np.concatenate([[0], np.cumsum([len(x) for x in sigs])]).astype(np.int64)
        dumps['adapter_values'] = np.concatenate(sigs) if sigs else np.zeros(0, np.float32)
    (events_part,) = glob.glob(os.path.join(dump_out, 'events', 'part-*.h5'))
    with h5py.File(events_part, 'r') as h5:
        ids = sorted(h5['basecalled_events/00000007'])
        tabs = [h5['basecalled_events/00000007/' + k][:] for k in ids]
        dumps['events_ids'] = np.array(ids)
        dumps['events_offsets'] = np.concatenate([[0], np.cumsum([len(x) for x in tabs])]).astype(np.int64)
        dumps['events_rows'] = np.concatenate(tabs)
        attrs = {}
        for k in ids:
            a = {}
            for name, v in h5['basecalled_events/00000007/' + k].attrs.items():
                a[name] = [type(v).__name__ if not hasattr(v, 'dtype') else str(v.dtype),
                           v.decode() if isinstance(v, bytes) else (v.item() if hasattr(v, 'item') else v)]
            attrs[k] = a
        dumps['events_attrs'] = np.array(json.dumps(attrs))
    print('dumps: %d adapter signals, %d event tables' % (len(dumps['adapter_ids']), len(dumps['events_ids'])))
    for key in ('adapter_catalog', 'events_rows'):        # plain field types (h5py hangs metadata on them)
        a = dumps[key]
        dumps[key] = a.astype([(name, a.dtype[name].str) for name in a.dtype.names])
    np.savez_compressed(os.path.join(OUT, 'dumps0.npz'), **dumps)

    # ---- unit-level vectors straight from reference functions --------------
    unit = {}
    # a11 normalize_signal, a10 push rules
    demux = BC.BarcodeDemultiplexer(refcfg['demultiplexing'], 18)
    ns_in, ns_out = [], []
    for ln in (1, 2, 3, 7, 8, 259, 260, 299, 300, 301, 600, 3000, 3001):
        x = (rng.normal(80, 7, ln) + (rng.random(ln) < 0.1) * 25).astype(np.float32)
        if ln == 8:
            x[:] = 80.0          # mad == 0 -> divisor clamps to 0.01
        ns_in.append(x)
        ns_out.append(BC.BarcodeDemultiplexer.normalize_signal(x).astype(np.float32))

    class _R:
        pass
    push_out, push_flag = [], []
    for x in ns_in:
        demux.clear()
        demux.push(_R(), x)
        push_flag.append(len(demux.signals))
        push_out.append(np.array(demux.signals[0], np.float32) if demux.signals
                        else np.zeros(300, np.float32))
    unit['ns_len'] = np.array([len(x) for x in ns_in])
    unit['ns_in'] = np.concatenate(ns_in)
    unit['ns_out'] = np.concatenate(ns_out)
    unit['push_out'] = np.stack(push_out)
    unit['push_flag'] = np.array(push_flag, np.int8)
    # a13 phred lookup + threshold
    tbl = np.array(demux.calibration_table)
    sc = np.concatenate([np.float32([0, -0.5, 1e-9, 1.0, 0.5]), tbl.astype(np.float32),
                         np.nextafter(tbl.astype(np.float32), np.float32(2)),
                         np.nextafter(tbl.astype(np.float32), np.float32(-1)),
                         rng.random(200).astype(np.float32)]).astype(np.float32)
    unit['phred_score'] = sc
    unit['phred'] = np.array([demux.lookup_calibrated_phred_score(s) for s in sc], np.int32)
    unit['score_threshold'] = np.float64(demux.score_threshold)
    unit['thr_pass'] = np.array([bool(s >= demux.score_threshold) for s in sc], np.int8)
    # a4 de-standardise + QC (real fit_scalers with a stub predict)
    loader = SL.SignalLoader(refcfg['signal_processing'], inputdir)
    qs, qh = [0.6822922162926033, 1.2284099385049299], [-14.681989446640273, 25.67682983554034]
    sm, ss_, hm, hs = 0.9553510773987666, 0.13295630234669656, 5.497420194450036, 9.82564593783874
    pr = [rng.normal(0, 1.2, (300, 2))]
    for bound in qs:
        c = (bound - sm) / ss_
        pr.append(np.stack([c + np.arange(-40, 41) * 2e-8, np.zeros(81)], 1))
    for bound in qh:
        c = (bound - hm) / hs
        pr.append(np.stack([np.zeros(81), c + np.arange(-40, 41) * 2e-8], 1))
    pred = np.concatenate(pr).astype(np.float32)

    class _NR:
        def __init__(self):
            self.params, self.status = None, 'okay'

        def set_scaling_params(self, p):
            self.params = np.array(p)

        def set_status(self, s, stop=False):
            self.status = s
    loader.scaler_model = types.SimpleNamespace(predict=lambda x, bs: pred)
    loader.head_signals = [np.zeros(2000, np.float32)] * len(pred)
    loader.head_signal_assoc_reads = [_NR() for _ in pred]
    loader.fit_scalers()
    unit['xfrm_pred'] = pred
    unit['xfrm_ok'] = np.array([r.status == 'okay' for r in loader.head_signal_assoc_reads], np.int8)
    unit['xfrm_ss'] = np.array([r.params if r.params is not None else [np.nan, np.nan]
                                for r in loader.head_signal_assoc_reads], np.float32)
    # a15 event detector: real csupport (CPython ext) on assorted windows
    ev_in, ev_out, ev_cnt = [], [], []
    for k in range(12):
        ln = int(rng.integers(30, 5000))
        if k == 0:
            ln = 30       # < 2*window2 -> long detector silent
        if k == 1:
            ln = 13       # < 2*window1 -> single zero-length event
        lv = rng.normal(95, 15, ln // 8 + 2)
        x = np.repeat(lv, rng.geometric(1 / 9., len(lv)))[:ln]
        x = (x + rng.normal(0, 1.5, len(x))).astype(np.float32)
        ev = ref_detect_events(x, window_length1=7, window_length2=20, threshold1=3,
                               threshold2=8, peak_height=4)
        ev_in.append(x)
        ev_out.append(np.array(ev))
        ev_cnt.append(len(ev))
    unit['ev_len'] = np.array([len(x) for x in ev_in])
    unit['ev_in'] = np.concatenate(ev_in)
    evs = np.concatenate(ev_out)
    for f in ('start', 'length', 'mean', 'stdv'):
        unit['ev_' + f] = np.array(evs[f])
    unit['ev_cnt'] = np.array(ev_cnt)
    # a16 interval DP on random event tables (real find_best_polya_interval)
    pan = PA.PolyASignalAnalyzer(refcfg['polya_dwell'])
    dp_n, dp_isp, dp_len, dp_res = [], [], [], []
    for k in range(60):
        ne = int(rng.integers(1, 40))
        isp = rng.random(ne) < rng.uniform(0.2, 0.9)
        ln = rng.integers(1, 400, ne).astype(np.float32)
        evdf = pd.DataFrame({'is_polya': isp, 'length': ln})
        got = pan.find_best_polya_interval(evdf)
        dp_n.append(ne)
        dp_isp.append(isp)
        dp_len.append(ln)
        dp_res.append((-1, -1) if len(got) == 0 else (int(got.index[0]), int(got.index[-1])))
    unit['dp_n'] = np.array(dp_n)
    unit['dp_isp'] = np.concatenate(dp_isp).astype(np.uint8)
    unit['dp_len'] = np.concatenate(dp_len)
    unit['dp_res'] = np.array(dp_res, np.int32)
    # utils.union_intervals
    ui_in = [[[int(a), int(a + rng.integers(0, 30))] for a in rng.integers(0, 200, m)]
             for m in (0, 1, 2, 5, 12)]
    ui = {'in': ui_in, 'out': [UT.union_intervals([list(x) for x in s]) for s in ui_in]}
    np.savez_compressed(os.path.join(OUT, 'unit.npz'), **unit)
    with open(os.path.join(OUT, 'unit.json'), 'w') as fh:
        json.dump(jsonable(ui), fh)

    # ---- a18/a19: Guppy event table + pseudo-fusion filter -----------------
    # chimeric reads = two synthetic reads back to back with one DAQ setting
    # and one scaling; real process_batch with --filter-chimera
    cb = synth_batch(16, seed=9330, samples_per_read=26000, jitter=0.2, fixed_calib=True,
                     scale_sigma=0.0, shift_sigma=0.0)
    parts = [cb['arena'][cb['offsets'][i]:cb['offsets'][i + 1]] for i in range(16)]
    chim = []
    for k in range(6):                      # 6 chimeras (A|B), 4 plain reads
        chim.append(('chimera%d' % k, np.concatenate([parts[2 * k], parts[2 * k + 1]])))
    for k in range(12, 16):
        chim.append(('plain%d' % k, parts[k]))
    # a chimera whose second part is cut right after its adapter (few bases follow)
    cut = int(cb['truth'][1, 3, 1]) + 300
    chim.append(('chimera_short_tail', np.concatenate([parts[0], parts[1][:cut]])))
    cinput = os.path.join(TMP, 'in2')
    os.makedirs(cinput)
    citems = []
    for i, (tag, raw) in enumerate(chim):
        rid = '%08x-0000-4000-8000-%012x' % (0x9330 + i, i)
        fn = 'c%03d.fast5' % i
        meta = {'read_number': 500 + i, 'start_time': int(rng.integers(10**5, 10**8)),
                'channel_number': int(rng.integers(1, 513)), 'run_id': 'run' + 'cd' * 19,
                'sample_id': 'synthetic'}
        bc = make_basecall(rng, len(raw), int(rng.integers(0, 40)))
        write_fast5(os.path.join(cinput, fn), rid, raw, cb['calib'][0], meta, bc)
        citems.append({'tag': tag, 'raw': raw, 'filename': fn, 'read_id': rid, 'meta': meta,
                       'basecall': bc})
    ccfg = dict(refcfg)
    ccfg.update({'inputdir': cinput, 'measure_polya': False, 'filter_unsplit_reads': True,
                 'trim_adapter': False})
    ccap = {'events': {}, 'cands': {}, 'unsplit': {}}
    orig_load = SA.SignalAnalysis.load_events
    orig_det = SA.SignalAnalysis.detect_unsplit_read
    orig_union = SA.union_intervals
    current = {}

    def cload(self):
        ev = orig_load(self)
        ccap['events'][self.npread.read_id] = (np.array(ev['mean'], np.float32),
                                                np.array(ev['scaled_mean'], np.float32),
                                                np.array(ev['start'], np.int64),
                                                np.array(ev['pos'], np.int64),
                                                np.array(ev['p_model_state'], np.float64))
        return ev

    def cdet(self, events, segments, elspan):
        current['rid'] = self.npread.read_id
        ccap['cands'][current['rid']] = []
        out = orig_det(self, events, segments, elspan)
        ccap['unsplit'][current['rid']] = bool(out)
        return out

    def cunion(iset):
        ccap['cands'][current['rid']] = [list(map(int, x)) for x in iset]
        return orig_union(iset)

    SA.SignalAnalysis.load_events = cload
    SA.SignalAnalysis.detect_unsplit_read = cdet
    SA.union_intervals = cunion
    WPS = sys.modules.pop('__poreplex_persistence', None)     # new inputdir -> new loader
    cres = SA.process_batch(8, [(it['filename'], it['read_id']) for it in citems], ccfg)
    assert not (isinstance(cres, tuple) and cres[0] == -1), cres
    SA.SignalAnalysis.load_events = orig_load
    SA.SignalAnalysis.detect_unsplit_read = orig_det
    SA.union_intervals = orig_union
    print('chimera batch:', [(it['tag'], r['status'], r.get('label')) for it, r in zip(citems, cres)])
    carena, coff = N.pack_reads([it['raw'] for it in citems])
    nc = len(citems)
    np.savez_compressed(
        os.path.join(OUT, 'chimera.pxr.npz'),
        arena=carena, offsets=coff, calib=np.array([tuple(cb['calib'][0])] * nc, dtype=N.CALIB_DTYPE),
        filename=np.array([it['filename'] for it in citems]),
        read_id=np.array([it['read_id'] for it in citems]),
        duration=np.array([len(it['raw']) for it in citems], dtype=np.int64),
        start_time=np.array([it['meta']['start_time'] for it in citems], dtype=np.int64),
        channel_number=np.array([str(it['meta']['channel_number']) for it in citems]),
        run_id=np.array([it['meta']['run_id'] for it in citems]),
        sample_id=np.array([it['meta']['sample_id'] for it in citems]),
        basecall=np.array([json.dumps(it['basecall']) for it in citems]),
        tag=np.array([it['tag'] for it in citems]), broken_files=np.array([], dtype='U1'))
    ev_off = np.zeros(nc + 1, np.int64)
    ev_mean, ev_scaled, ev_pos, ev_pms, cand_list, unsplit = [], [], [], [], [], []
    for i, it in enumerate(citems):
        m, sm, st, ps, pm = ccap['events'][it['read_id']]
        ev_off[i + 1] = ev_off[i] + len(m)
        ev_mean.append(m); ev_scaled.append(sm); ev_pos.append(ps); ev_pms.append(pm)
        cand_list.append(ccap['cands'].get(it['read_id'], []))
        unsplit.append(ccap['unsplit'].get(it['read_id'], False))
    np.savez_compressed(os.path.join(OUT, 'chimera.stages.npz'), ev_offsets=ev_off,
                        ev_mean=np.concatenate(ev_mean), ev_scaled=np.concatenate(ev_scaled),
                        ev_pos=np.concatenate(ev_pos), ev_pms=np.concatenate(ev_pms),
                        unsplit=np.array(unsplit, np.int8))
    with open(os.path.join(OUT, 'chimera.results.json'), 'w') as fh:
        json.dump({'batchid': 8, 'reads': [(it['filename'], it['read_id']) for it in citems],
                   'config_flags': {k: ccfg[k] for k in (
                       'barcoding', 'measure_polya', 'trim_adapter', 'filter_unsplit_reads',
                       'minimum_sequence_length', 'barcoding_quality_filter')},
                   'candidates': cand_list, 'results': jsonable(cres)}, fh, indent=1)

    # ---- a18/a19 over albacore's 14-column Events tables (fast5_file.py:178-179 passes them through) ----
    # the chimera reads again, their events cut by the reference's own event detector (variable lengths),
    # run through the REAL process_batch with --filter-chimera and --dump-basecalls.  `start' is int64 for
    # most reads (the table the reference's arithmetic works on) and uint64 -- what albacore itself writes,
    # signal_analyzer.py:66-67 mirrors its dtypes -- for two: there `end' = start + duration promotes to
    # float64 and range() at :385 refuses it, so the reference turns those reads into unknown_error.
    ALB = [('mean', '<f4'), ('start', '<i8'), ('stdv', '<f4'), ('length', '<i8'), ('model_state', 'S5'),
           ('move', '<i4'), ('weights', '<f4'), ('p_model_state', '<f4'), ('mp_state', 'S5'),
           ('p_mp_state', '<f4'), ('p_A', '<f4'), ('p_C', '<f4'), ('p_G', '<f4'), ('p_T', '<f4')]

    def albacore_table(g, raw, cal, first, unsigned):
        pa = np.array(float(cal['range']) / float(cal['digitisation']) * (raw[first:].astype(np.float64) + float(cal['offset'])),
                      dtype=np.float32)
        ev = ref_detect_events(pa, window_length1=3, window_length2=6, threshold1=1.4, threshold2=9.0, peak_height=0.2)
        dt = [(n, t.replace('<i8', '<u8') if unsigned and n in ('start', 'length') else t) for n, t in ALB]
        tab = np.zeros(len(ev), dtype=dt)
        tab['mean'], tab['stdv'] = ev['mean'], ev['stdv']
        tab['start'] = np.asarray(ev['start'], dtype=np.int64) + first
        tab['length'] = np.asarray(ev['length'], dtype=np.int64)
        tab['move'] = g.choice([0, 1, 2], len(ev), p=[0.45, 0.5, 0.05])
        tab['move'][0] = 1
        kmers = np.array([''.join(k) for k in g.choice(list('ACGT'), (len(ev), 5))], dtype='S5')
        tab['model_state'], tab['mp_state'] = kmers, kmers
        tab['p_model_state'] = g.uniform(0.05, 0.99, len(ev)).astype(np.float32)
        tab['p_model_state'][g.random(len(ev)) < 0.02] = np.float32(0.4)      # exactly the quality limit
        tab['weights'], tab['p_mp_state'] = 1.0, tab['p_model_state']
        for c in ('p_A', 'p_C', 'p_G', 'p_T'):
            tab[c] = 0.25
        seq_len = int(tab['move'].sum())
        seq = ''.join(g.choice(list('ACGU'), seq_len))
        qs = ''.join(chr(33 + int(q)) for q in g.integers(3, 25, seq_len))
        return tab, {'sequence': seq, 'qstring': qs, 'sequence_length': seq_len,
                     'mean_qscore': float(np.round(g.uniform(7, 12), 3)), 'block_stride': 15,
                     'num_events': int(len(ev)), 'first_sample_template': int(first)}

    def write_albacore_fast5(path, read_id, raw, cal, meta, tab, bc):
        write_fast5(path, read_id, raw, cal, meta, None)
        with h5py.File(path, 'a') as h5:
            g = h5.create_group('Analyses/Basecall_1D_000')
            tpl = g.create_group('BaseCalled_template')
            tpl.create_dataset('Fastq', data=np.string_('@{}\n{}\n+\n{}\n'.format(read_id, bc['sequence'], bc['qstring'])))
            tpl.create_dataset('Events', data=tab)
            sm = g.create_group('Summary/basecall_1d_template')
            sm.attrs['sequence_length'] = np.int32(bc['sequence_length'])
            sm.attrs['mean_qscore'] = np.float32(bc['mean_qscore'])
            sg = h5.create_group('Analyses/Segmentation_000/Summary/segmentation')
            sg.attrs['num_events_template'] = np.int32(bc['num_events'])
            sg.attrs['first_sample_template'] = np.int32(bc['first_sample_template'])

    ag = np.random.default_rng(9440)
    ainput = os.path.join(TMP, 'in3')
    os.makedirs(ainput)
    aitems = []
    alb_reads = [chim[k] for k in (0, 1, 2, 6, 7, 10)] + [('chimera_u8', chim[0][1]), ('plain_u8', chim[7][1])]
    for i, (tag, raw) in enumerate(alb_reads):
        rid = '%08x-0000-4000-8000-%012x' % (0x9440 + i, i)
        fn = 'a%03d.fast5' % i
        meta = {'read_number': 700 + i, 'start_time': int(ag.integers(10**5, 10**8)),
                'channel_number': int(ag.integers(1, 513)), 'run_id': 'run' + 'ef' * 19, 'sample_id': 'synthetic'}
        tab, bc = albacore_table(ag, raw, cb['calib'][0], int(ag.integers(0, 40)), tag.endswith('_u8'))
        write_albacore_fast5(os.path.join(ainput, fn), rid, raw, cb['calib'][0], meta, tab, bc)
        bcd = dict(bc, table='albacore', move=tab['move'].tolist(), p_model_state=None,
                   events={k: np.array(tab[k]) for k in ('start', 'length', 'mean', 'stdv', 'move', 'p_model_state', 'model_state')})
        aitems.append({'tag': tag, 'raw': raw, 'filename': fn, 'read_id': rid, 'meta': meta, 'basecall': bcd})
    aout = os.path.join(TMP, 'out_alb')
    os.makedirs(os.path.join(aout, 'events'))
    acfg = dict(refcfg)
    acfg.update({'inputdir': ainput, 'outputdir': aout, 'measure_polya': False, 'filter_unsplit_reads': True,
                 'trim_adapter': False, 'dump_basecalls': True})
    acap = {'cands': {}, 'scaled': {}}
    acur = {}

    def aload(self):
        ev = orig_load(self)
        acap['scaled'][self.npread.read_id] = np.array(ev['scaled_mean'])
        return ev

    def adet(self, events, segments, elspan):
        acur['rid'] = self.npread.read_id
        acap['cands'][acur['rid']] = []
        return orig_det(self, events, segments, elspan)

    def aunion(iset):
        acap['cands'][acur['rid']] = [list(map(int, x)) for x in iset]
        return orig_union(iset)

    SA.SignalAnalysis.load_events, SA.SignalAnalysis.detect_unsplit_read, SA.union_intervals = aload, adet, aunion
    sys.modules.pop('__poreplex_persistence', None)
    sys.stderr, saved = open(os.devnull, 'w'), sys.stderr
    try:
        ares = SA.process_batch(9, [(it['filename'], it['read_id']) for it in aitems], acfg)
    finally:
        sys.stderr = saved
        SA.SignalAnalysis.load_events, SA.SignalAnalysis.detect_unsplit_read, SA.union_intervals = orig_load, orig_det, orig_union
    assert not (isinstance(ares, tuple) and ares[0] == -1), ares
    print('albacore batch:', [(it['tag'], r['status'], r.get('label')) for it, r in zip(aitems, ares)])
    import glob as _glob
    (apart,) = _glob.glob(os.path.join(aout, 'events', 'part-*.h5'))
    with h5py.File(apart, 'r') as h5:
        ids = sorted(h5['basecalled_events/00000009'])
        tabs = [h5['basecalled_events/00000009/' + k][:] for k in ids]
        aattrs = {}
        for k in ids:
            a = {}
            for name, v in h5['basecalled_events/00000009/' + k].attrs.items():
                a[name] = [type(v).__name__ if not hasattr(v, 'dtype') else str(v.dtype),
                           v.decode() if isinstance(v, bytes) else (v.item() if hasattr(v, 'item') else v)]
            aattrs[k] = a
    rows = np.concatenate(tabs)
    rows = rows.astype([(name, rows.dtype[name].str) for name in rows.dtype.names])
    from poreplex_amd.fast5_file import write_bundle
    aarena, aoff = N.pack_reads([it['raw'] for it in aitems])
    na = len(aitems)
    write_bundle(os.path.join(OUT, 'albacore.pxr.npz'), aarena, aoff,
                 np.array([tuple(cb['calib'][0])] * na, dtype=N.CALIB_DTYPE),
                 [it['filename'] for it in aitems], [it['read_id'] for it in aitems],
                 basecalls=[it['basecall'] for it in aitems],
                 start_time=np.array([it['meta']['start_time'] for it in aitems], dtype=np.int64),
                 channel_number=np.array([str(it['meta']['channel_number']) for it in aitems]),
                 run_id=np.array([it['meta']['run_id'] for it in aitems]),
                 sample_id=np.array([it['meta']['sample_id'] for it in aitems]),
                 tag=np.array([it['tag'] for it in aitems]))
    _b = dict(np.load(os.path.join(OUT, 'albacore.pxr.npz')))          # (the same arrays, deflated)
    np.savez_compressed(os.path.join(OUT, 'albacore.pxr.npz'), **_b)
    so = np.zeros(na + 1, np.int64)
    sc = []
    for i, it in enumerate(aitems):
        v = acap['scaled'].get(it['read_id'], np.zeros(0, np.float32))
        assert v.dtype == np.float32
        so[i + 1] = so[i] + len(v)
        sc.append(v)
    np.savez_compressed(os.path.join(OUT, 'albacore.stages.npz'), scaled_offsets=so, scaled=np.concatenate(sc),
                        events_ids=np.array(ids), events_rows=rows,
                        events_offsets=np.concatenate([[0], np.cumsum([len(x) for x in tabs])]).astype(np.int64),
                        events_attrs=np.array(json.dumps(aattrs)))
    with open(os.path.join(OUT, 'albacore.results.json'), 'w') as fh:
        json.dump({'batchid': 9, 'reads': [(it['filename'], it['read_id']) for it in aitems],
                   'config_flags': {k: acfg[k] for k in (
                       'barcoding', 'measure_polya', 'trim_adapter', 'filter_unsplit_reads', 'dump_basecalls',
                       'minimum_sequence_length', 'barcoding_quality_filter')},
                   'candidates': [acap['cands'].get(it['read_id'], []) for it in aitems],
                   'results': jsonable(ares)}, fh, indent=1)

    # ---- a14/a17 poly(A): real PolyASignalAnalyzer on full-resolution reads
    class _PRead:
        def __init__(self, sig, rate):
            self.sig, self.sampling_rate, self.polya = sig, rate, None

        def load_signal(self, pool=None, pad=False):
            return self.sig

        def set_polya_tail(self, info):
            self.polya = info
    pol = []
    for i, it in enumerate(items):
        rid = it['read_id']
        if rid not in cap['segments'] or 'adapter' not in cap['segments'][rid]:
            continue
        res = next(r for r in results if r.get('read_id') == rid)
        sp = np.float32([res.get('scale', 0), 0])
        pol.append({'index': i, 'read_id': rid, 'polya': res.get('polya')})
    with open(os.path.join(OUT, 'polya.json'), 'w') as fh:
        json.dump(jsonable(pol), fh, indent=1)
    GE.write(OUT, ENGINES, {'arith': ARITH, 'python': sys.version.split()[0], 'numpy': np.__version__})
    print('wrote goldens to', OUT, {f: os.path.getsize(os.path.join(OUT, f))
                                     for f in sorted(os.listdir(OUT))})


if __name__ == '__main__':
    main()
    if ONLY:
        import shutil
        for name in sorted(os.listdir(OUT)):
            if name.startswith(ONLY) or name == 'engines.json':
                shutil.copy(os.path.join(OUT, name), os.path.join(_FINAL, name))
                print('kept', name)
