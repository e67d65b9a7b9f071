"""No fixed per-read output limits (missing 4): the reference lists EVERY spike of a
poly(A) tail (polya.py:109-121) and EVERY in-window adapter candidate of the pseudo-fusion scan
(signal_analyzer.py:384-418).  The GPU path used to keep 64 spike rows per read and 16 candidates
per window and turned anything beyond into that read's unknown_error; now the spike rows come
out of one arena that grows on demand (re-run of the reads that found it full) and the candidate
slots are sized from the config's duration cut-offs, so a window cannot overflow them.
-m gpu; the oracle has no limits."""
import copy
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import assert_spikes_equal

from poreplex_amd import native as N
from poreplex_amd import synth as SY
from poreplex_amd.config import default_config

pytestmark = pytest.mark.gpu


def spiky_reads(n=6, seed=41, period=90, width=8, bump=90):
    """Reads with a ~25 000-sample poly(A) piece carrying a short level bump every `period`
    samples: under the PRESET's spike tolerance the whole piece is one tail with hundreds of
    spike events."""
    old = SY._PIECE_LEN['polya-tail']
    SY._PIECE_LEN['polya-tail'] = (24000, 26000)
    try:
        sb = SY.synth_batch(n, seed=seed, samples_per_read=60000, jitter=0.05, mean_dwell=40.0,
                            sample_noise=1.0)
    finally:
        SY._PIECE_LEN['polya-tail'] = old
    arena = sb['arena'].copy()
    for i in range(n):
        o = sb['offsets'][i]
        a, b = sb['truth'][i, 4]
        for s in range(a + 200, b - 200, period):
            arena[o + s:o + s + width] += bump
    sb['arena'] = arena
    return sb


def leader_adapter_trains(n=4, seed=43, period_blocks=12):
    """Reads whose payload is a train of leader -> adapter stretches, 15 blocks apart."""
    sb = SY.synth_batch(n, seed=seed, samples_per_read=60000, jitter=0.05, fixed_calib=True,
                        scale_sigma=0.0, shift_sigma=0.0)
    arena = sb['arena'].copy()
    rng = np.random.default_rng(seed)
    k = float(sb['calib'][0]['digitisation'] / sb['calib'][0]['range'])
    off0 = float(sb['calib'][0]['offset'])
    scale, shift = sb['scale_shift'][0]

    def to_raw(level):
        return np.rint(((level - shift) / scale) * k - off0)
    for i in range(n):
        o = sb['offsets'][i]
        a = int(sb['truth'][i, 5, 0])
        L = int(sb['offsets'][i + 1] - o)
        s = a + 600
        while s + period_blocks * 15 < L - 300:
            half = period_blocks * 15 // 2
            arena[o + s:o + s + half // 2] = to_raw(102.0 + rng.normal(0, 1.0, half // 2))
            arena[o + s + half // 2:o + s + half] = to_raw(112.0 + rng.normal(0, 1.0, half - half // 2))
            arena[o + s + half:o + s + 2 * half] = to_raw(80.5 + rng.normal(0, 2.0, half))
            s += period_blocks * 15 + 45
    sb['arena'] = arena
    return sb


def test_tails_with_hundreds_of_spikes_equal_the_oracle(ctx, oracle):
    sb = spiky_reads()
    # mixed with ordinary reads, so that the batch has tails of every size
    plain = SY.synth_batch(40, seed=7, samples_per_read=30000)
    parts = [sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in range(6)] + \
        [plain['arena'][plain['offsets'][i]:plain['offsets'][i + 1]] for i in range(40)]
    order = np.random.default_rng(3).permutation(len(parts))
    arena, off = N.pack_reads([parts[i] for i in order])
    cal = np.concatenate([sb['calib'], plain['calib']])[order]
    ss = np.concatenate([sb['scale_shift'], plain['scale_shift']])[order]
    mask = N.STAGE_SEGMENT | N.STAGE_POLYA
    want, wsp = oracle.process_batch(arena, off, cal, ss, mask, want_spikes=True)
    assert (want['polya_n_spikes'] >= 200).sum() >= 3 and want['polya_n_spikes'].max() > 300
    ctx.upload(arena, off, cal, ss)
    ctx.run(mask)
    got = ctx.download()
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f], equal_nan=True), f
    spikes = ctx.download_spikes(got)
    assert_spikes_equal(spikes, want, wsp)
    assert len(spikes[0]) == int(want['polya_n_spikes'][want['polya_called'] != 0].sum()) > 1500
    # the standalone hook takes the same route
    ok = np.nonzero(want['status'] == 0)[0]
    sub_arena, sub_off = N.pack_reads([parts[order[i]] for i in ok])
    res, hook_spikes = ctx.polya(sub_arena, sub_off, cal[ok], ss[ok], want['seg_first'][ok],
                                 want['seg_last'][ok], want_spikes=True)
    assert np.array_equal(res['polya_n_spikes'], want['polya_n_spikes'][ok])
    assert_spikes_equal(hook_spikes, want[ok], wsp[ok])


def test_facade_reports_every_spike(oracle, tmp_path):
    """Through process_batch: the poly(A) dict of a 300-spike tail lists 300 rows (it used to be
    that read's unknown_error)."""
    from poreplex_amd.fast5_file import write_bundle
    from poreplex_amd.signal_analyzer import process_batch
    from poreplex_amd.worker_persistence import WorkerPersistenceStorage
    sb = spiky_reads(n=4, seed=45)
    n = 4
    names, ids = ['s/r%d.fast5' % i for i in range(n)], ['id-%d' % i for i in range(n)]
    path = str(tmp_path / 'spiky.pxr.npz')
    write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], names, ids,
                 basecalls=SY.synth_basecalls(sb, seed=1))
    cfg = default_config(inputdir=str(tmp_path), outputdir=str(tmp_path), read_bundle=path,
                         barcoding=True, measure_polya=True)
    WorkerPersistenceStorage.reset()
    try:
        res = process_batch(0, list(zip(names, ids)), cfg)
    finally:
        WorkerPersistenceStorage.reset()
    assert isinstance(res, list), res
    want, wsp = oracle.process_batch(sb['arena'], sb['offsets'], sb['calib'], None,
                                     N.STAGE_ALL_DEMUX | N.STAGE_POLYA, want_spikes=True)
    seen = 0
    for i, r in enumerate(res):
        assert r['status'] != 'unknown_error', r.get('error_message')
        if want['polya_called'][i]:
            rows = r['polya']['spikes']
            assert len(rows) == want['polya_n_spikes'][i]
            assert np.array_equal(np.float32(rows), wsp[i, :len(rows)], equal_nan=True)
            seen = max(seen, len(rows))
    assert seen >= 200


def many_candidate_config():
    cfg = copy.deepcopy(default_config())
    cfg['unsplit_read_detection'].update(strict_full_length=0.03, strict_dna_length=0.01,
                                         loosen_full_length=0.03, loosen_dna_length=0.01,
                                         window_size=6, window_step=3)
    return cfg


def test_windows_with_forty_candidates_equal_the_oracle():
    """Duration cut-offs of 30 ms let a 6 s window hold more than 40 leader -> adapter candidates; the
    slots follow the config (pxg_unsplit_cand_slots), every candidate comes back, in the
    reference's append order."""
    from oracle.pxo import Oracle
    cfg = many_candidate_config()
    orc = Oracle(cfg)
    sb = leader_adapter_trains()
    n = len(sb['offsets']) - 1
    mask = N.STAGE_SEGMENT
    want = orc.process_batch(sb['arena'], sb['offsets'], sb['calib'], sb['scale_shift'], mask)
    gpu = N.NativeContext(cfg, device_id=0)
    try:
        gpu.upload(sb['arena'], sb['offsets'], sb['calib'], sb['scale_shift'])
        gpu.run(mask)
        got = gpu.download()
        assert np.array_equal(got['seg_first'], want['seg_first'])
        nb = np.diff(sb['offsets']) // 15
        iv, cnt, start = gpu.unsplit_scan(np.zeros(n, np.int64), nb)
        most_in_a_window = 0
        for i in range(n):
            raw = sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]]
            _, sc = orc.guppy_event_means(raw, sb['calib'][i], 0, int(nb[i]), want['scale'][i], want['shift'][i])
            wiv, wc = orc.unsplit_scan(sc, 0, (int(want['seg_last'][i][3]) + 1) * 15,
                                       float(sb['calib'][i]['sampling_rate']))
            assert cnt[i] == wc, (i, cnt[i], wc)
            assert iv[start[i]:start[i + 1]].tolist() == wiv.tolist(), i
            if wc:      # candidates are appended window by window: a window = a run of increasing starts
                runs = np.split(wiv[:, 0], np.nonzero(np.diff(wiv[:, 0]) < 0)[0] + 1)
                most_in_a_window = max(most_in_a_window, max(len(r) for r in runs))
        assert most_in_a_window >= 40 and cnt.max() > 100
    finally:
        gpu.close()


def test_process_batch_ex_from_threads_equals_the_split_calls(ctx, oracle):
    """pxg_process_batch_ex (one call per worker batch, several host threads on one context,
    spikes + window scan in the same call) against upload / run / download / scan, on batches
    that differ per call; raw and encoded samples."""
    batches = []
    for k in range(6):
        sb = SY.synth_batch(48 + 16 * k, seed=200 + k, samples_per_read=26000, jitter=0.3,
                            short_fraction=0.05)
        batches.append(sb)
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    want = []
    for sb in batches:
        n = len(sb['offsets']) - 1
        ctx.upload(sb['arena'], sb['offsets'], sb['calib'])
        ctx.run(mask)
        rec = ctx.download()
        spk = ctx.download_spikes(rec)
        scan = ctx.unsplit_scan(np.zeros(n, np.int64), np.diff(sb['offsets']) // 15)
        want.append((rec, spk, scan))

    def call(k):
        sb = batches[k % len(batches)]
        n = len(sb['offsets']) - 1
        samples = sb['arena']
        if k % 2:
            z, chunks, _ = N.z_encode(sb['arena'], sb['offsets'])
            samples = N.EncodedSamples(z, chunks, 0, 0, len(sb['arena']))
        return ctx.process_batch_ex(samples, sb['offsets'], sb['calib'], mask, want_spikes=True,
                                    unsplit=(np.zeros(n, np.int64), np.diff(sb['offsets']) // 15, 15))
    with ThreadPoolExecutor(4) as pool:
        outs = list(pool.map(call, range(18)))
    for k, got in enumerate(outs):
        rec, spk, scan = want[k % len(batches)]
        assert got['records'].tobytes() == rec.tobytes(), k
        assert np.array_equal(got['spikes'][1], spk[1]) and np.array_equal(got['spikes'][0], spk[0], equal_nan=True), k
        assert np.array_equal(got['unsplit'][1], scan[1]) and np.array_equal(got['unsplit'][0], scan[0]), k
    # a too-small guess for the variable-size outputs is retried inside the binding
    ctx._spike_cap, ctx._unsplit_cap = 0, 0
    sb = spiky_reads(n=6, seed=49)
    got = ctx.process_batch_ex(sb['arena'], sb['offsets'], sb['calib'], N.STAGE_SEGMENT | N.STAGE_POLYA,
                               scale_shift=sb['scale_shift'], want_spikes=True)
    w, wsp = oracle.process_batch(sb['arena'], sb['offsets'], sb['calib'], sb['scale_shift'],
                                  N.STAGE_SEGMENT | N.STAGE_POLYA, want_spikes=True)
    assert_spikes_equal(got['spikes'], w, wsp)
    assert len(got['spikes'][0]) > 2 * 6 + 1024


def test_big_pageable_uploads_go_through_the_contexts_own_chunks(ctx):
    """pxg_h2d_big (profiles/r05/fault_hunt.md): a pageable sample arena of 512 KB or more never reaches the runtime's
    in-place page lock -- it travels through two 8 MB page-locked chunks of the context.  Arenas just below, at and
    above one and two chunks, from pageable memory, from a page-locked mapping and (round 4's form) handed to the
    runtime as they are: the same records."""
    base = SY.synth_batch(200, seed=31, samples_per_read=44000, jitter=0.05)
    lens = np.diff(base['offsets'])
    for total in ((8 << 20) // 2 - 3, (8 << 20) // 2, (8 << 20) // 2 + 1, (16 << 20) // 2 + 5):
        # whole reads up to `total` samples, the last one cut so that the arena has exactly that many
        k = int(np.searchsorted(np.cumsum(lens), total)) + 1
        off = np.concatenate([[0], np.cumsum(lens[:k])]).astype(np.int64)
        off[-1] = total                                           # (a short last read is just a short read)
        assert off[-1] > off[-2] and len(base['arena']) > total
        arena = np.array(base['arena'][:total])                   # pageable
        cal = base['calib'][:k]
        got = ctx.process_batch(arena, off, cal)
        locked = N.page_exclusive(total, np.int16)
        locked[:] = arena
        ctx.pin(locked)
        try:
            assert ctx.process_batch(locked, off, cal).tobytes() == got.tobytes(), total
        finally:
            ctx.unpin(locked)
        ctx.upload(arena, off, cal)
        ctx.run(N.STAGE_ALL_DEMUX)
        assert ctx.download().tobytes() == got.tobytes(), total
    assert (got['status'] == 0).sum() > 80
    # 64 MB and more: four host threads, each through its own pair of chunks
    reps = 5
    arena = np.tile(base['arena'], reps)[:len(base['arena']) * reps - 3]
    off = np.concatenate([[0]] + [base['offsets'][1:] + k * len(base['arena']) for k in range(reps)]).astype(np.int64)
    off[-1] = len(arena)
    cal = np.tile(base['calib'], reps)
    assert arena.nbytes > (64 << 20) + (8 << 20)
    got = ctx.process_batch(arena, off, cal)
    n0 = len(base['offsets']) - 1
    first = ctx.process_batch(np.array(base['arena']), base['offsets'], base['calib'])
    for k in range(reps - 1):                       # every copy of the base reads: the records of the base reads
        assert got[k * n0:(k + 1) * n0].tobytes() == first.tobytes(), k
    assert got[(reps - 1) * n0:-1].tobytes() == first[:-1].tobytes()
