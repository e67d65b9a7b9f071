"""The worker call's short path (SignalAnalyzer.process_plain_run + csrc/pxg_pyreport.c report_run) against the path
that defines it (ReadTable + SignalAnalyzer.judge + ReadTable.report): CPU only, no GPU involved -- both paths take
the SAME records, from a context double.

  * randomised bundles and crafted records (every status the rules can give, every barcode combination): the two
    paths return equal objects -- same keys in the same order, same value types, same values;
  * the calls the short path must decline (a short read in the run, an irregular basecall summary, a shuffled or
    foreign request, lists instead of tuples, options that walk a read) -- it returns None and the general path runs;
  * stretches of the golden batch, through the oracle-backed double: the short path's dicts equal the REAL
    reference's (tests/golden/batch0.results.json).
"""
import json
import math
import os
import sys

import numpy as np
import pytest

import oracle_context
from poreplex_amd import native as N
from poreplex_amd import signal_analyzer as SA
from poreplex_amd.config import default_config
from poreplex_amd.fast5_file import ReadBundle, write_bundle
from poreplex_amd.worker_persistence import WorkerPersistenceStorage

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

pytestmark = pytest.mark.skipif(N.load_pyhost() is None or not hasattr(N.load_pyhost(), 'report_run'),
                                reason='csrc/_pxgpy is not built for this interpreter')


def worker_objects():
    return sys.modules[WorkerPersistenceStorage.STORAGE_NAME].storage


def same(a, b, where='result'):
    """a and b are equal objects of equal types, recursively; dicts also agree in key order"""
    assert type(a) is type(b), (where, type(a), type(b))
    if isinstance(a, dict):
        assert list(a) == list(b), (where, list(a), list(b))
        for k in a:
            same(a[k], b[k], '{}[{!r}]'.format(where, k))
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), (where, len(a), len(b))
        for k, (x, y) in enumerate(zip(a, b)):
            same(x, y, '{}[{}]'.format(where, k))
    elif isinstance(a, float):
        assert (a == b and math.copysign(1, a) == math.copysign(1, b)) or (a != a and b != b), (where, a, b)
    else:
        assert a == b, (where, a, b)


class CraftedRecords(oracle_context.NativeEntryMixin, oracle_context.OracleBackedContext):
    """Context double whose GPU pass returns records made up per read (the read is recognised by its calibration
    offset, which the bundle below numbers)."""
    table = None
    candidate_every = 40

    def process_batch_ex(self, samples, offsets, calib, stage_mask=N.STAGE_ALL_DEMUX, scale_shift=None, unsplit=None,
                         want_spikes=False):
        which = np.asarray(calib)['offset'].astype(np.int64)
        assert len(which) == len(offsets) - 1
        rec = CraftedRecords.table[which].copy()
        out = {'records': rec}
        if want_spikes:                    # every spike row of the batch, CSR by record (pxg_batch_download_spikes)
            counts = np.where(rec['polya_called'] != 0, rec['polya_n_spikes'], 0)
            off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            rows = np.random.default_rng(int(which[0])).uniform(1, 90, (int(off[-1]), 4)).astype(np.float32)
            for k in np.nonzero(counts)[0].tolist():      # the rows of a read do not depend on the call it came in
                rows[off[k]:off[k + 1]] = np.random.default_rng(1000 + int(which[k])).uniform(1, 90, (int(counts[k]), 4))
            out['spikes'] = (rows, off)
        if unsplit is not None:            # in-read adapter candidates of the window scan: some calls have none at all
            first_sample, n_blocks, stride = unsplit
            assert len(first_sample) == len(which) and len(n_blocks) == len(which) and stride == 15
            cnt = np.where((np.asarray(n_blocks) > 0) & (which % CraftedRecords.candidate_every == 3 % CraftedRecords.candidate_every), 1 + which % 4, 0).astype(np.int32)
            start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
            iv = np.zeros((int(start[-1]), 2), dtype=np.int64)
            for k in np.nonzero(cnt)[0].tolist():
                # anywhere from before the payload to behind the read's last sample, overlapping, in no order
                g = np.random.default_rng(5000 + int(which[k]))
                begin = g.integers(0, 13000, int(cnt[k]))
                iv[start[k]:start[k + 1]] = np.stack([begin, begin + g.integers(1, 3000, int(cnt[k]))], axis=1)
            out['unsplit'] = (iv, cnt, start)
        return out


def crafted_bundle(tmp_path, n, seed):
    """A bundle of n reads (read k has calibration offset k) and one crafted record per read."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(9600, 12000, n)
    short = rng.random(n) < 0.04
    lens[short] = rng.integers(100, 8000, int(short.sum()))              # (the scaler needs 9 000 samples)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    arena = rng.integers(300, 900, int(off[-1])).astype(np.int16)
    cal = np.zeros(n, dtype=N.CALIB_DTYPE)
    cal['range'], cal['digitisation'] = 1400.0, 8192.0
    cal['offset'] = np.arange(n)
    cal['sampling_rate'] = rng.choice([3012.0, 4000.0, 3012.5], n)
    names = ['run{}/f{:03d}.fast5'.format(k // 50, k // 4) for k in range(n)]
    ids = ['{:08x}-0000-4000-8000-{:012x}'.format(seed, k) for k in range(n)]
    basecalls = []
    for k in range(n):
        u = rng.random()
        if u < 0.12:
            basecalls.append(None)                                       # not basecalled
            continue
        n_blocks = int(lens[k]) // 15
        n_bases = int(rng.integers(3, 9)) if u < 0.24 else int(rng.integers(12, 60))      # some below the minimum length
        move = np.zeros(n_blocks, dtype=np.uint8)
        move[rng.choice(n_blocks, size=min(n_bases - 4 if n_bases > 4 else 1, n_blocks), replace=False)] = 1
        n_bases = int(move.sum()) + 4
        bc = {'sequence': ''.join('ACGU'[i] for i in rng.integers(0, 4, n_bases)),
              'qstring': ''.join(chr(33 + q) for q in rng.integers(5, 25, n_bases)), 'block_stride': 15,
              'sequence_length': n_bases, 'mean_qscore': float(rng.uniform(5, 14)),      # (a float64 the table rounds to float32)
              'num_events': n_blocks, 'first_sample_template': 0, 'table': 'move', 'move': move}
        if 0.24 <= u < 0.27:
            bc['move'] = np.concatenate([move, np.zeros(5, dtype=np.uint8)])      # a frame longer than the raw signal: irregular
        basecalls.append(bc)
    path = str(tmp_path / 'crafted{}.pxr.npz'.format(seed))
    write_bundle(path, arena, off, cal, names, ids, basecalls=basecalls)
    rec = np.zeros(n, dtype=N.RESULT_DTYPE)
    rec['seg_first'], rec['seg_last'] = -1, -1
    adapter = 0                                                          # filled in by the caller (state order of the model)
    rec['status'] = np.where(rng.random(n) < 0.1, N.STATUS_CODE['scaling_qc_fail'], 0)
    rec['scale'], rec['shift'] = rng.uniform(0.7, 1.2, n), rng.uniform(-10, 20, n)
    rec['bc_pushed'] = rng.random(n) < 0.8
    rec['bc_called'] = rng.random(n) < 0.7
    rec['bc_label'] = rng.integers(0, 4, n)
    rec['bc_phred'] = rng.integers(0, 60, n)
    rec['polya_called'] = rng.random(n) < 0.6
    rec['polya_n_spikes'] = np.where(rng.random(n) < 0.5, 0, rng.integers(1, 5, n))
    rec['polya_dwell_samples'] = rng.integers(50, 9000, n)
    rec['polya_begin'] = rng.integers(100, 5000, n)
    rec['polya_end'] = rec['polya_begin'] + rng.integers(60, 4000, n)
    found = rng.random(n) >= 0.1
    return path, rec, found, short, adapter


@pytest.fixture()
def crafted(monkeypatch):
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', CraftedRecords)
    yield
    WorkerPersistenceStorage.reset()
    CraftedRecords.table = None


def spy_on_the_short_path(monkeypatch):
    taken = []
    real = SA.SignalAnalyzer.process_plain_run

    def spy(self, reads, phase):
        out = real(self, reads, phase)
        taken.append(out is not None)
        return out
    monkeypatch.setattr(SA.SignalAnalyzer, 'process_plain_run', spy)
    return taken


@pytest.mark.parametrize('barcoding,polya,chimera', [(True, False, False), (False, False, False), (True, True, False),
                                                     (False, True, False), (True, True, True), (True, False, True)])
@pytest.mark.parametrize('seed', [1, 2])
def test_short_path_equals_the_general_path(crafted, monkeypatch, tmp_path, seed, barcoding, polya, chimera):
    n = 400
    path, rec, found, short, _ = crafted_bundle(tmp_path, n, seed)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path, barcoding=barcoding,
                         measure_polya=polya, filter_unsplit_reads=chimera, minimum_sequence_length=10)
    # one call to make the worker's objects (the context knows where the model keeps its adapter state)
    keys = ReadBundle(path).keys
    CraftedRecords.table = rec
    first = SA.process_batch(0, keys[:1], cfg)
    assert isinstance(first, list), first
    adapter = worker_objects()['ctx'].state_names.index('adapter')
    plain = ReadBundle(path).plain_run_columns(worker_objects()['loader'].scaler_cfg)
    # (with the scan on, a Move table of another k-mer size also sends the call to the batch table; reads too short for
    #  the scaler ride along: their dicts come first, as on the general path)
    ok = plain['regular'] & (plain['kmer_ok'] | ~plain['long_enough']) if chimera else plain['regular']
    assert not plain['ok'].all() and not plain['regular'].all() and (plain['regular'] & ~plain['long_enough']).any()
    rec['seg_first'][:, adapter] = np.where(found, 40, -1)
    rec['seg_last'][:, adapter] = np.where(found, 90, -1)
    rng = np.random.default_rng(100 + seed)
    taken = spy_on_the_short_path(monkeypatch)
    fell_back = []
    real_finish = SA.SignalAnalyzer.finish_from_pass
    monkeypatch.setattr(SA.SignalAnalyzer, 'finish_from_pass',
                        lambda self, *a: fell_back.append(1) or real_finish(self, *a))
    few = []
    real_some = SA.SignalAnalyzer.finish_some_from_pass
    monkeypatch.setattr(SA.SignalAnalyzer, 'finish_some_from_pass',
                        lambda self, *a: few.append(len(a[2])) or real_some(self, *a))
    statuses, keysets, n_taken = set(), set(), 0
    windows = [(0, n)] + [(int(a), int(rng.integers(1, 70))) for a in rng.integers(0, n - 1, 60)]
    for lo, k in windows:
        reads = keys[lo:lo + k]
        monkeypatch.setattr(SA, '_PLAIN_RUN', True)
        del taken[:]
        fast = SA.process_batch(1, list(reads), cfg)
        took = taken == [True]
        monkeypatch.setattr(SA, '_PLAIN_RUN', False)
        general = SA.process_batch(1, list(reads), cfg)
        assert isinstance(general, list), general
        assert took == bool(ok[lo:lo + k].all()), (lo, k)
        if took:
            n_taken += 1
            same(fast, general, 'reads[{}:{}]'.format(lo, lo + k))
            statuses |= {r['status'] for r in fast}
            keysets |= {tuple(r) for r in fast}
        else:
            assert [r['read_id'] for r in fast if 'read_id' in r] == [r['read_id'] for r in general if 'read_id' in r]
    assert n_taken >= 15
    # (with the scan on, a read with candidates is judged over its event table, on the batch table the call falls back to)
    assert statuses - {'unsplit_read'} == {'okay', 'scaling_qc_fail', 'adapter_not_detected', 'not_basecalled',
                                           'sequence_too_short', 'scaler_signal_too_short'}
    assert ('unsplit_read' in statuses) == chimera
    # a few reads with candidates in a call: a table of just those beside the C pass; many (a quarter of a small call):
    # the whole call on the batch table with the pass it had made; none: the C pass alone
    assert (few and 0 < len(few) + len(fell_back) < n_taken) if chimera else not (few or fell_back)
    assert any('barcode' in ks for ks in keysets) == barcoding and any('sequence' not in ks for ks in keysets)
    assert any('polya' in ks for ks in keysets) == polya


def test_calls_the_short_path_declines(crafted, monkeypatch, tmp_path):
    path, rec, found, short, _ = crafted_bundle(tmp_path, 120, 7)
    CraftedRecords.table = rec
    b = ReadBundle(path)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path, barcoding=True)
    columns = b.plain_run_columns({'length': 30000, 'stride': 15, 'min_length': 9000})
    assert not columns['ok'][short].any() and columns['ok'].sum() > 60
    ok = columns['regular']              # (reads too short for the scaler ride along; an irregular basecall summary does not)
    assert not ok.all()
    # a stretch of plain reads
    run = max((j - i, i, j) for i in range(120) for j in range(i + 1, 121) if ok[i:j].all())
    lo, hi = run[1], run[2]
    assert hi - lo >= 5
    good = b.keys[lo:hi]
    taken = spy_on_the_short_path(monkeypatch)

    def call(reads, **kw):
        del taken[:]
        out = SA.process_batch(3, reads, dict(cfg, **kw))
        assert kw or isinstance(out, list), out          # (with other options on, the double's answers may not do)
        return taken[-1], out
    assert call(list(good))[0] is True
    assert call(list(good[:1]))[0] is True
    reference = call(list(good))[1]
    for reads in (list(reversed(good)),                         # not in bundle order
                  [list(k) for k in good],                      # lists: the general path takes any sequence of pairs
                  tuple(good),
                  list(good) + [('elsewhere/x.fast5', 'nobody')],
                  list(good[:2]) + list(good[3:])):             # a gap
        took, out = call(reads)
        assert took is False and len(out) == len(reads)
    took, out = call([list(k) for k in good])
    same(out, reference)
    for option in ({'dump_adapter_signals': True}, {'dump_basecalls': True},
                   {'trim_adapter': True, 'trim_adapter_as_intended': True}):
        WorkerPersistenceStorage.reset()
        assert call(list(good), **option)[0] is False, option
    WorkerPersistenceStorage.reset()
    bad = int(np.nonzero(~ok)[0][0])                            # a run with one read that is not plain in it
    took, out = call(b.keys[max(bad - 2, 0):bad + 3])
    assert took is False
    # a run with a read too short for the scaler in it is taken -- unless PXG_NO_SHORT_IN_RUN says otherwise
    tiny = next(i for i in np.nonzero(short)[0].tolist() if ok[max(i - 2, 0):i + 3].all())
    took, with_short = call(b.keys[max(tiny - 2, 0):tiny + 3])
    assert took is True and with_short[0]['status'] == 'scaler_signal_too_short' and with_short[0]['read_id'] == b.read_ids[tiny]
    monkeypatch.setattr(SA, '_SHORT_IN_RUN', False)
    took, table_says = call(b.keys[max(tiny - 2, 0):tiny + 3])
    assert took is False
    same(with_short, table_says)
    monkeypatch.setattr(SA, '_SHORT_IN_RUN', True)
    # nothing to do is not a run either
    assert call([])[1] == [] and taken == [False]


@pytest.mark.parametrize('encoded', [False, True])
def test_short_path_against_the_real_reference(monkeypatch, tmp_path, encoded):
    """Stretches of the golden batch that are plain runs, through the oracle-backed double: the REAL reference's
    dicts, poly(A) tails and their spike rows included -- from the bundle as committed and from the same bundle with
    its samples encoded (the call then hands the encoded bytes of its reads to the context)."""
    import test_facade as TF
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', TF.OracleBackedContext)
    try:
        with open(os.path.join(GOLDEN, 'batch0.results.json')) as fh:
            ref = json.load(fh)
        cfg = TF.facade_config(ref)
        bundle_path = TF.BUNDLE
        if encoded:
            d = dict(ReadBundle(TF.BUNDLE).d)
            z, chunks, chunk_base = N.z_encode(d.pop('arena'), d['offsets'])
            d.update(arena_z=z, z_chunks=chunks, z_chunk_base=chunk_base, bundle_version=np.int64(3))
            bundle_path = str(tmp_path / 'batch0_encoded.pxr.npz')
            np.savez(bundle_path, **d)
            cfg = dict(cfg, read_bundle=bundle_path)
        want = {(r['filename'], r.get('read_id')): r for r in ref['results']}
        b = ReadBundle(bundle_path)
        assert b.compressed == encoded
        ok = b.plain_run_columns({'length': 30000, 'stride': 15, 'min_length': 9000})['regular']
        reads = [tuple(r) for r in ref['reads']]
        at = [b.index.get(k, -1) for k in reads]
        taken = spy_on_the_short_path(monkeypatch)
        covered, pos = 0, 0
        while pos < len(reads):
            end = pos
            while end < len(reads) and at[end] >= 0 and ok[at[end]] and at[end] == at[pos] + (end - pos) and \
                    reads[end][0] not in b.broken:
                end += 1
            if end == pos:
                pos += 1
                continue
            got = SA.process_batch(ref['batchid'], reads[pos:end], cfg)
            assert taken[-1] is True and isinstance(got, list)
            expect = [want[k] for k in reads[pos:end]]
            # (what stopped before the pass comes first, in encounter order -- signal_analyzer.py:82-134 -- then the rest)
            expect = [r for r in expect if r['status'] == 'scaler_signal_too_short'] + \
                [r for r in expect if r['status'] != 'scaler_signal_too_short']
            TF.compare_results(got, expect, check_polya=True)
            covered += end - pos
            pos = end
        assert covered >= 24, covered
    finally:
        WorkerPersistenceStorage.reset()


def test_short_path_from_many_threads(crafted, monkeypatch, tmp_path):
    """Worker threads sharing the context (the reference keeps `parallel` calls in flight, pipeline.py:96): every call
    gets the dicts of its own reads."""
    from concurrent.futures import ThreadPoolExecutor
    path, rec, found, short, _ = crafted_bundle(tmp_path, 400, 11)
    CraftedRecords.table = rec
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path, barcoding=True)
    keys = ReadBundle(path).keys
    assert isinstance(SA.process_batch(0, keys[:1], cfg), list)
    adapter = worker_objects()['ctx'].state_names.index('adapter')
    rec['seg_first'][:, adapter] = np.where(found, 40, -1)
    ok = worker_objects()['loader'].bundle.plain_run_columns(worker_objects()['loader'].scaler_cfg)['regular']
    rng = np.random.default_rng(5)
    windows = [(int(a), int(rng.integers(1, 40))) for a in rng.integers(0, 399, 200)]
    monkeypatch.setattr(SA, '_PLAIN_RUN', False)
    want = [SA.process_batch(k, keys[lo:lo + n], cfg) for k, (lo, n) in enumerate(windows)]
    monkeypatch.setattr(SA, '_PLAIN_RUN', True)
    taken = spy_on_the_short_path(monkeypatch)
    with ThreadPoolExecutor(8) as pool:
        got = list(pool.map(lambda kw: SA.process_batch(kw[0], keys[kw[1][0]:kw[1][0] + kw[1][1]], cfg), enumerate(windows)))
    assert sum(taken) == sum(bool(ok[lo:lo + n].all()) for lo, n in windows) > 20
    for g, w, (lo, n) in zip(got, want, windows):
        if ok[lo:lo + n].all():
            same(g, w)
        else:
            assert isinstance(g, list) and len(g) == len(w)


def count_decodes(monkeypatch, decodes):
    """decodes gets a 1 for every pass of the native reader over a call's reads: separate calls (fast5_file.decode_layout,
    behind as_bundle too) or inside the fused call (SignalLoader.decode_and_run)."""
    from poreplex_amd import fast5_file as F5
    from poreplex_amd import signal_loader as SL
    real_decode, real_fused = F5.decode_layout, SL.SignalLoader.decode_and_run
    monkeypatch.setattr(F5, 'decode_layout', lambda *a, **kw: decodes.append(1) or real_decode(*a, **kw))
    monkeypatch.setattr(SA, 'decode_layout', F5.decode_layout)
    monkeypatch.setattr(SL.SignalLoader, 'decode_and_run', lambda self, *a, **kw: decodes.append(1) or real_fused(self, *a, **kw))


def crafted_fast5_files(tmp_path, bundle_path, reads_per_file=37):
    """The reads of a crafted bundle as multi-read FAST5 files under tmp_path/f5: [(filename, read_id)] in the order a
    batch maker lists them (file by file, each file's reads in the file's own order) and the bundle index of each."""
    from poreplex_amd.fast5_file import get_read_ids
    from poreplex_amd.fast5_write import Fast5Writer
    b = ReadBundle(bundle_path)
    top = tmp_path / 'f5'
    top.mkdir()
    n = len(b.keys)
    by_id = {}
    for lo in range(0, n, reads_per_file):
        with Fast5Writer(str(top / 'multi{:03d}.fast5'.format(lo // reads_per_file))) as w:
            for i in range(lo, min(lo + reads_per_file, n)):
                by_id[b.read_ids[i]] = i
                w.add_read(b.read_ids[i], b.samples(i), b.d['calib'][i], start_time=int(b.d['start_time'][i]),
                           channel_number=str(b.d['channel_number'][i]), run_id=str(b.d['run_id'][i]),
                           sample_id=str(b.d['sample_id'][i]), basecall=b.basecall_of(i))
    keys = []
    for lo in range(0, n, reads_per_file):
        keys += get_read_ids('multi{:03d}.fast5'.format(lo // reads_per_file), str(top))
    assert len(keys) == n
    return str(top), keys, np.array([by_id[r] for _, r in keys])


@pytest.mark.parametrize('barcoding,polya,chimera', [(True, False, False), (True, True, True), (False, True, False)])
def test_short_path_from_fast5_files_equals_the_general_path(crafted, monkeypatch, tmp_path, barcoding, polya, chimera):
    """Worker calls whose reads live in multi-read FAST5 files (the reference's own input, no read bundle): a call
    that is a stretch of the files' reads in file order -- across file boundaries too -- takes the short path over the
    per-call bundle the native reader makes of it, and returns what the batch table returns; everything else (a short
    or irregular read in the call, a shuffled call) declines AFTER the files were read, and the general path takes the
    bundle that was built instead of reading them again."""
    from poreplex_amd import fast5_file as F5
    n, seed = 300, 5
    path, rec, found, short, _ = crafted_bundle(tmp_path, n, seed)
    top, keys, which = crafted_fast5_files(tmp_path, path)
    cfg = default_config(inputdir=top, outputdir=str(tmp_path), barcoding=barcoding, measure_polya=polya,
                         filter_unsplit_reads=chimera, minimum_sequence_length=10)
    CraftedRecords.table = rec
    first = SA.process_batch(0, keys[:1], cfg)
    assert isinstance(first, list), first
    assert worker_objects()['loader'].bundle is None
    adapter = worker_objects()['ctx'].state_names.index('adapter')
    plain = ReadBundle(path).plain_run_columns(worker_objects()['loader'].scaler_cfg)
    ok = (plain['regular'] & (plain['kmer_ok'] | ~plain['long_enough']) if chimera else plain['regular'])[which]
    rec['seg_first'][:, adapter] = np.where(found, 40, -1)
    rec['seg_last'][:, adapter] = np.where(found, 90, -1)
    taken = spy_on_the_short_path(monkeypatch)
    decodes = []
    count_decodes(monkeypatch, decodes)
    rng = np.random.default_rng(100 + seed)
    windows = [(0, n), (30, 20)] + [(int(a), int(rng.integers(1, 90 if j < 15 else 25))) for j, a in enumerate(rng.integers(0, n - 1, 70))]
    statuses, n_taken, arenas = set(), 0, worker_objects()['loader'].call_arenas
    for lo, k in windows:
        reads = keys[lo:lo + k]
        monkeypatch.setattr(SA, '_PLAIN_RUN', True)
        del taken[:], decodes[:]
        fast = SA.process_batch(1, list(reads), cfg)
        took = taken == [True]
        assert decodes == [1], (lo, k, decodes)              # the files are read once, whichever path reports
        monkeypatch.setattr(SA, '_PLAIN_RUN', False)
        general = SA.process_batch(1, list(reads), cfg)
        assert isinstance(general, list), general
        assert took == bool(ok[lo:lo + k].all()), (lo, k)
        same(fast, general, 'reads[{}:{}]'.format(lo, lo + k))
        if took:
            n_taken += 1
            statuses |= {r['status'] for r in fast}
    assert n_taken >= 12
    assert statuses - {'unsplit_read'} == {'okay', 'scaling_qc_fail', 'adapter_not_detected', 'not_basecalled',
                                           'sequence_too_short', 'scaler_signal_too_short'}
    assert 1 <= len(arenas.free) <= 2                         # the calls took turns with the same sample arena
    # the taken calls made their GPU pass from inside csrc/pxg_pyreport.c decode_and_run (decode + pass behind one
    # release of the interpreter lock), through the double's C entry point; with PXG_NO_FUSED_CALL they decode first
    # and call the context afterwards -- same dicts
    assert CraftedRecords.native_calls >= n_taken
    before = CraftedRecords.native_calls
    monkeypatch.setattr(SA, '_PLAIN_RUN', True)
    monkeypatch.setattr(SA, '_FUSED_CALL', False)
    lo, k = next((lo, k) for lo, k in windows[1:] if ok[lo:lo + k].all() and k > 3)
    del taken[:]
    unfused = SA.process_batch(1, list(keys[lo:lo + k]), cfg)
    monkeypatch.setattr(SA, '_FUSED_CALL', True)
    fused = SA.process_batch(1, list(keys[lo:lo + k]), cfg)
    assert taken == [True, True] and CraftedRecords.native_calls == before + 1
    same(unfused, fused)
    # a shuffled call is not a run: the general path, as before
    monkeypatch.setattr(SA, '_PLAIN_RUN', True)
    del taken[:]
    back = SA.process_batch(2, list(reversed(keys[40:60])), cfg)
    monkeypatch.setattr(SA, '_PLAIN_RUN', False)
    same(back, SA.process_batch(2, list(reversed(keys[40:60])), cfg))
    assert taken == [False]
    # and with PXG_NO_PLAIN_RUN_FAST5 only bundle reads take the short path
    monkeypatch.setattr(SA, '_PLAIN_RUN', True)
    monkeypatch.setattr(SA, '_PLAIN_RUN_FAST5', False)
    run = next((lo, k) for lo, k in windows[1:] if ok[lo:lo + k].all())
    del taken[:]
    SA.process_batch(3, list(keys[run[0]:run[0] + run[1]]), cfg)
    assert taken == [False]


def test_fast5_call_with_an_undecodable_read_and_with_buffers_too_small(crafted, monkeypatch, tmp_path):
    """Two exits of the fused decode + pass (csrc/pxg_pyreport.c decode_and_run): a read whose samples cannot be decoded
    -- the pass is not made, the call goes to the batch table with the bundle it has, and that read alone is an
    'unknown_error' as on the general path --, and a variable-size output that outgrows the buffer of the prepared call
    (PXG_E_NOMEM with the totals set): the wrapper's loop makes the pass again."""
    import zlib
    from poreplex_amd.fast5_file import clear_open_cache, get_read_ids
    from poreplex_amd.fast5_write import Fast5Writer
    n = 24
    path, rec, found, short, _ = crafted_bundle(tmp_path, n, 21)
    b = ReadBundle(path)
    plain_ok = b.plain_run_columns({'length': 30000, 'stride': 15, 'min_length': 9000})['ok']      # (long enough and regular)
    good = [i for i in range(n) if plain_ok[i]][:12]
    assert len(good) == 12
    top = tmp_path / 'f5'
    top.mkdir()
    with Fast5Writer(str(top / 'z.fast5')) as w:
        for i in good:
            w.add_read(b.read_ids[i], b.samples(i), b.d['calib'][i], basecall=b.basecall_of(i), compression='gzip')
    keys = get_read_ids('z.fast5', str(top))
    victim = good[[r for _, r in keys].index(b.read_ids[good[5]])]
    CraftedRecords.table = rec
    rec['status'] = 0
    rec['polya_called'], rec['polya_n_spikes'] = 1, 3
    cfg = default_config(inputdir=str(top), outputdir=str(tmp_path), barcoding=True, measure_polya=True, minimum_sequence_length=10)
    assert isinstance(SA.process_batch(0, keys[:1], cfg), list)
    adapter = worker_objects()['ctx'].state_names.index('adapter')
    rec['seg_first'][:, adapter], rec['seg_last'][:, adapter] = 40, 90
    taken = spy_on_the_short_path(monkeypatch)
    whole = SA.process_batch(1, list(keys), cfg)
    assert taken == [True] and all(r['status'] != 'unknown_error' for r in whole)
    # --- buffers too small: the prepared call comes back with PXG_E_NOMEM, the wrapper's loop sizes them
    class Tight(N.BatchExCall):
        def __init__(self, *a):
            super().__init__(*a)
            self.spike_cap = self.x.spike_cap = 2
    monkeypatch.setattr(N, 'BatchExCall', Tight)
    before = CraftedRecords.native_calls
    del taken[:]
    again = SA.process_batch(1, list(keys), cfg)
    assert taken == [True] and CraftedRecords.native_calls == before + 1
    same(again, whole)
    monkeypatch.undo()
    monkeypatch.setattr(N, 'NativeContext', CraftedRecords)
    # --- one read's compressed samples damaged on disk (the deflate stream of its chunk)
    blob = zlib.compress(b.samples(victim).astype('<i2').tobytes(), 1)
    data = bytearray((top / 'z.fast5').read_bytes())
    at = data.find(blob)
    assert at > 0 and len(blob) > 64
    data[at + 20:at + 40] = bytes(20)
    (top / 'z.fast5').write_bytes(bytes(data))
    clear_open_cache()
    WorkerPersistenceStorage.reset()
    taken = spy_on_the_short_path(monkeypatch)
    decodes = []
    count_decodes(monkeypatch, decodes)
    before = CraftedRecords.native_calls
    hurt = SA.process_batch(2, list(keys), cfg)
    assert taken == [False] and decodes == [1] and CraftedRecords.native_calls == before      # no pass from the fused call
    monkeypatch.setattr(SA, '_PLAIN_RUN', False)
    same(hurt, SA.process_batch(2, list(keys), cfg))
    bad = [r for r in hurt if r['status'] == 'unknown_error']
    assert len(bad) == 1 and bad[0]['read_id'] == b.read_ids[victim] and 'cannot be decoded' in bad[0]['error_message']
    healthy = {r['read_id']: r for r in whole}
    for r in hurt:
        if r['status'] != 'unknown_error':
            same(r, healthy[r['read_id']])


def test_short_path_from_fast5_files_on_many_threads(crafted, monkeypatch, tmp_path):
    """Calls over FAST5 files from worker threads that share the loader: each gets the dicts of its own reads, and the
    sample arenas go round."""
    from concurrent.futures import ThreadPoolExecutor
    path, rec, found, short, _ = crafted_bundle(tmp_path, 300, 13)
    top, keys, which = crafted_fast5_files(tmp_path, path, reads_per_file=64)
    CraftedRecords.table = rec
    cfg = default_config(inputdir=top, outputdir=str(tmp_path), barcoding=True, measure_polya=True, minimum_sequence_length=10)
    assert isinstance(SA.process_batch(0, keys[:1], cfg), list)
    adapter = worker_objects()['ctx'].state_names.index('adapter')
    rec['seg_first'][:, adapter] = np.where(found, 40, -1)
    rng = np.random.default_rng(6)
    windows = [(int(a), int(rng.integers(1, 50))) for a in rng.integers(0, 299, 150)]
    monkeypatch.setattr(SA, '_PLAIN_RUN', False)
    want = [SA.process_batch(k, keys[lo:lo + k_], cfg) for k, (lo, k_) in enumerate(windows)]
    monkeypatch.setattr(SA, '_PLAIN_RUN', True)
    taken = spy_on_the_short_path(monkeypatch)
    with ThreadPoolExecutor(8) as pool:
        got = list(pool.map(lambda kw: SA.process_batch(kw[0], keys[kw[1][0]:kw[1][0] + kw[1][1]], cfg), enumerate(windows)))
    assert sum(taken) > 20
    for g, w in zip(got, want):
        same(g, w)
    assert 1 <= len(worker_objects()['loader'].call_arenas.free) <= 8


@pytest.mark.gpu
@pytest.mark.parametrize('polya,chimera', [(False, False), (True, False), (True, True)])
def test_short_path_equals_the_general_path_on_the_gpu(monkeypatch, tmp_path, polya, chimera):
    """The same comparison with the real context: a synthetic bundle, 128-read calls, from threads as well."""
    from concurrent.futures import ThreadPoolExecutor
    from poreplex_amd.synth import synth_basecalls, synth_batch
    n = 384
    sb = synth_batch(n, seed=31, samples_per_read=20000)
    # a few reads too short for the scaler in every call (and an empty one): they ride along in the plain run, the pass
    # gives them up at its own gate, the dicts say what the host's gate says and come first
    o = sb['offsets']
    cut = {5: 3000, 6: 0, 130: 8999, 255: 14, 256: 4000, 383: 1}
    parts = [sb['arena'][o[j]:o[j + 1]][:cut.get(j, 1 << 30)] for j in range(n)]
    sb = dict(sb, arena=np.concatenate(parts), offsets=np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64))
    names = ['a/r{:05d}.fast5'.format(i) for i in range(n)]
    ids = ['{:08x}-0000-4000-8000-{:012x}'.format(31, i) for i in range(n)]
    path = str(tmp_path / 'gpu.pxr.npz')
    write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], names, ids, basecalls=synth_basecalls(sb, seed=31))
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path, barcoding=True,
                         measure_polya=polya, filter_unsplit_reads=chimera)
    keys = list(zip(names, ids))
    WorkerPersistenceStorage.reset()
    try:
        monkeypatch.setattr(SA, '_PLAIN_RUN', False)
        want = [SA.process_batch(k, keys[lo:lo + 128], cfg) for k, lo in enumerate(range(0, n, 128))]
        assert all(isinstance(w, list) for w in want), want
        monkeypatch.setattr(SA, '_PLAIN_RUN', True)
        taken = spy_on_the_short_path(monkeypatch)
        got = [SA.process_batch(k, keys[lo:lo + 128], cfg) for k, lo in enumerate(range(0, n, 128))]
        assert taken == [True] * 3
        same(got, want)
        assert {r['status'] for w in want for r in w} >= {'okay', 'scaler_signal_too_short'}
        assert [r['read_id'] for r in want[0][:2]] == [ids[5], ids[6]] and want[2][0]['read_id'] == ids[256]
        assert any('polya' in r for w in want for r in w) == polya
        with ThreadPoolExecutor(6) as pool:
            again = list(pool.map(lambda lo: SA.process_batch(9, keys[lo:lo + 128], cfg), list(range(0, n, 128)) * 4))
        same(again, want * 4)
    finally:
        WorkerPersistenceStorage.reset()


@pytest.mark.gpu
@pytest.mark.parametrize('polya,chimera,compression', [(False, False, None), (True, True, None), (True, False, 'vbz')])
def test_short_path_from_fast5_files_on_the_gpu(monkeypatch, tmp_path, polya, chimera, compression):
    """128-read calls over multi-read FAST5 files with the real context (no read bundle): the short path over the
    per-call bundle, its samples in a recycled arena that is NOT page-locked (the library's bounded chunks carry them),
    against the batch table -- one call at a time and from threads -- and against the same reads served from a read
    bundle."""
    from concurrent.futures import ThreadPoolExecutor
    from poreplex_amd.fast5_file import get_read_ids
    from poreplex_amd.fast5_write import Fast5Writer, vbz_encode
    from poreplex_amd.synth import synth_basecalls, synth_batch
    if compression == 'vbz':
        try:
            vbz_encode(np.zeros(8, np.int16))
        except OSError as exc:
            pytest.skip('no libzstd on this host: {}'.format(exc))
    n = 384
    sb = synth_batch(n, seed=37, samples_per_read=20000)
    bcs = synth_basecalls(sb, seed=37)
    ids = ['{:08x}-0000-4000-8000-{:012x}'.format(37, i) for i in range(n)]
    top = tmp_path / 'f5'
    top.mkdir()
    o = sb['offsets']
    for lo in range(0, n, 150):                              # calls cross file boundaries
        with Fast5Writer(str(top / 'm{}.fast5'.format(lo // 150))) as w:
            for j in range(lo, min(lo + 150, n)):
                w.add_read(ids[j], sb['arena'][o[j]:o[j + 1]], sb['calib'][j], start_time=j, channel_number=str(1 + j % 512),
                           basecall=bcs[j], compression=compression)
    keys = []
    for lo in range(0, n, 150):
        keys += get_read_ids('m{}.fast5'.format(lo // 150), str(top))
    order = [ids.index(r) for _, r in keys]
    cfg = default_config(inputdir=str(top), outputdir=str(tmp_path), barcoding=True, measure_polya=polya,
                         filter_unsplit_reads=chimera)
    WorkerPersistenceStorage.reset()
    try:
        monkeypatch.setattr(SA, '_PLAIN_RUN', False)
        want = [SA.process_batch(k, keys[lo:lo + 128], cfg) for k, lo in enumerate(range(0, n, 128))]
        assert all(isinstance(w, list) for w in want), want
        monkeypatch.setattr(SA, '_PLAIN_RUN', True)
        taken = spy_on_the_short_path(monkeypatch)
        got = [SA.process_batch(k, keys[lo:lo + 128], cfg) for k, lo in enumerate(range(0, n, 128))]
        assert taken == [True] * 3
        same(got, want)
        assert {r['status'] for w in want for r in w} >= {'okay'}
        with ThreadPoolExecutor(6) as pool:
            again = list(pool.map(lambda lo: SA.process_batch(9, keys[lo:lo + 128], cfg), list(range(0, n, 128)) * 4))
        same(again, want * 4)
        assert 1 <= len(worker_objects()['loader'].call_arenas.free) <= 6
    finally:
        WorkerPersistenceStorage.reset()
    # the same reads from a read bundle: the records do not depend on where the samples came from
    path = str(tmp_path / 'same.pxr.npz')
    names = ['x/r{:05d}.fast5'.format(i) for i in range(n)]
    write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], names, ids, basecalls=bcs,
                 start_time=np.arange(n), channel_number=np.array([str(1 + j % 512) for j in range(n)]))
    try:
        from_bundle = SA.process_batch(0, list(zip(names, ids)), dict(cfg, inputdir=str(tmp_path), read_bundle=path))
        assert isinstance(from_bundle, list), from_bundle
        by_id = {r['read_id']: r for r in from_bundle}
        for w in want:
            for r in w:
                other = dict(by_id[r['read_id']], filename=r['filename'])
                assert {k: v for k, v in other.items() if k not in ('run_id', 'sample_id')} == \
                    {k: v for k, v in r.items() if k not in ('run_id', 'sample_id')}, r['read_id']
        assert order
    finally:
        WorkerPersistenceStorage.reset()


@pytest.mark.parametrize('seed', [3, 4])
def test_candidate_reads_in_bulk_equal_the_per_read_rule(crafted, monkeypatch, tmp_path, seed):
    """SignalAnalyzer.bulk_unsplit_rule (prefix sums over the Move column) against SignalAnalysis.detect_unsplit_read
    (the event table, read by read), on the batch table: every third read has candidates, anywhere in the read."""
    n = 300
    path, rec, found, short, _ = crafted_bundle(tmp_path, n, seed)
    rec['status'] = 0
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path, barcoding=True,
                         filter_unsplit_reads=True, minimum_sequence_length=10)
    keys = ReadBundle(path).keys
    CraftedRecords.table = rec
    monkeypatch.setattr(CraftedRecords, 'candidate_every', 3)
    assert isinstance(SA.process_batch(0, keys[:1], cfg), list)
    adapter = worker_objects()['ctx'].state_names.index('adapter')
    rng = np.random.default_rng(seed)
    rec['seg_first'][:, adapter] = 20
    rec['seg_last'][:, adapter] = rng.integers(30, 500, n)              # payload starts at 15 x (that + 1)
    monkeypatch.setattr(SA, '_PLAIN_RUN', False)
    settled_in_bulk = []
    real = SA.SignalAnalyzer.bulk_unsplit_rule

    def counting(self, *a):
        settled = real(self, *a)
        settled_in_bulk.append(int(settled.sum()))
        return settled
    monkeypatch.setattr(SA.SignalAnalyzer, 'bulk_unsplit_rule', counting)
    per_read = []
    real_checks = SA.SignalAnalyzer.base_space_checks
    monkeypatch.setattr(SA.SignalAnalyzer, 'base_space_checks',
                        lambda self, t, row, record: per_read.append(row) or real_checks(self, t, row, record))
    monkeypatch.setattr(SA, '_BULK_UNSPLIT', True)
    bulk = SA.process_batch(1, keys, cfg)
    in_bulk, left = sum(settled_in_bulk), len(per_read)
    monkeypatch.setattr(SA, '_BULK_UNSPLIT', False)
    del per_read[:]
    one_by_one = SA.process_batch(1, keys, cfg)
    assert isinstance(bulk, list) and isinstance(one_by_one, list)
    same(bulk, one_by_one)
    assert in_bulk >= 40 and len(per_read) == in_bulk + left
    verdicts = [r['status'] for r in bulk]
    assert verdicts.count('unsplit_read') >= 10 and verdicts.count('okay') >= 100


def test_without_the_extension_every_call_takes_the_batch_table(crafted, monkeypatch, tmp_path):
    """csrc/_pxgpy is optional (another interpreter version, PXG_NO_PYHOST=1): the short path then declines and the
    Python report loop returns the same dicts."""
    path, rec, found, short, _ = crafted_bundle(tmp_path, 120, 13)
    CraftedRecords.table = rec
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path, barcoding=True)
    b = ReadBundle(path)
    ok = b.plain_run_columns({'length': 30000, 'stride': 15, 'min_length': 9000})['regular']
    run = max((j - i, i, j) for i in range(120) for j in range(i + 1, 121) if ok[i:j].all())
    reads = b.keys[run[1]:run[2]]
    with_extension = SA.process_batch(0, list(reads), cfg)
    taken = spy_on_the_short_path(monkeypatch)
    monkeypatch.setattr(N, 'load_pyhost', lambda: None)
    without = SA.process_batch(0, list(reads), cfg)
    assert taken == [False]
    same(without, with_extension)


def test_call_arenas_pool():
    """SignalLoader's pool of sample arenas: the warmest arena that fits comes back, one that is too small makes room,
    at most `keep` wait; with a context the arenas are page-locked mappings of their own, never dropped, released once."""
    from poreplex_amd.signal_loader import CallArenas
    pool = CallArenas(keep=2)
    a = pool.take(1000)
    assert a.dtype == np.int16 and len(a) >= 1 << 20
    pool.give(a)
    assert pool.take(10) is a                      # the one given back last
    big = pool.take(3 << 20)
    assert len(big) >= 3 << 20 and big is not a
    pool.give(a)
    pool.give(big)
    third = np.empty(8, np.int16)
    pool.give(third)
    assert len(pool.free) == 2                     # (keep = 2)
    assert pool.take(2 << 20) is big and pool.take(5) is a
    pool.give(a)
    huge = pool.take(8 << 20)                      # nothing fits: the small one makes room
    assert len(huge) >= 8 << 20 and pool.free == []

    class Ctx:
        pinned, unpinned = [], []

        def pin(self, arr):
            self.pinned.append(arr.ctypes.data)

        def unpin(self, arr):
            self.unpinned.append(arr.ctypes.data)
    ctx = Ctx()
    locked = CallArenas(keep=1, ctx=ctx)
    x, y = locked.take(100), locked.take(100)
    assert x.ctypes.data % 4096 == 0 and len(ctx.pinned) == 2
    locked.give(x)
    locked.give(y)
    assert len(locked.free) == 2                   # a page-locked arena is never just dropped
    assert locked.take(50) is y
    locked.release()
    assert sorted(ctx.unpinned) == sorted(ctx.pinned) and locked.free == [] and locked.locked == []


def test_a_call_of_nothing_but_short_reads_makes_no_pass(crafted, monkeypatch, tmp_path):
    """Reads too short for the scaler only: the short path reports them without a GPU pass, like the batch table."""
    path, rec, found, short, _ = crafted_bundle(tmp_path, 200, 17)
    CraftedRecords.table = rec
    b = ReadBundle(path)
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path, barcoding=True)
    assert isinstance(SA.process_batch(0, b.keys[:1], cfg), list)
    i = int(np.nonzero(short)[0][0])
    taken = spy_on_the_short_path(monkeypatch)
    passes = []
    real = CraftedRecords.process_batch_ex
    monkeypatch.setattr(CraftedRecords, 'process_batch_ex', lambda self, *a, **kw: passes.append(1) or real(self, *a, **kw))
    got = SA.process_batch(1, b.keys[i:i + 1], cfg)
    assert taken == [True] and passes == [] and got[0]['status'] == 'scaler_signal_too_short'
    monkeypatch.setattr(SA, '_PLAIN_RUN', False)
    same(got, SA.process_batch(1, b.keys[i:i + 1], cfg))
    assert passes == []
