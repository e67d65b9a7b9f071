#!/usr/bin/env python
"""Soak of the thread-safe per-batch call (pxg_process_batch_ex) on a GPU: `threads` host threads
issue `calls` calls over a pool of ragged batches (raw and encoded samples, with and without the
poly(A) stage and the window scan, batches that overflow the first-pass spike arena and event
scratch) and every result is compared with the one the split calls gave for that batch.
usage (GPU box): python tools/api_stress.py [threads] [calls] [seed]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.synth import synth_batch  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)
ctx = N.NativeContext(default_config(), device_id=0)
import test_gpu_limits as TL  # noqa: E402  (the many-spike read generator)

pool = []
for k in range(12):
    n = int(rng.choice([1, 5, 33, 64, 200, 700]))
    sb = synth_batch(n, seed=seed * 100 + k, samples_per_read=int(rng.choice([12000, 26000, 60000])), jitter=0.4,
                     short_fraction=0.05)
    if k % 4 == 0:                       # tails with hundreds of spikes: the spike arena has to grow
        sp = TL.spiky_reads(n=3, seed=seed * 100 + k)
        parts = [sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in range(n)] + \
            [sp['arena'][sp['offsets'][i]:sp['offsets'][i + 1]] for i in range(3)]
        arena, off = N.pack_reads(parts)
        sb = {'arena': arena, 'offsets': off, 'calib': np.concatenate([sb['calib'], sp['calib']])}
    if k == 5:                           # a featureless long read: event scratch retry
        flat = (775 + rng.normal(0, 3, 130000)).astype(np.int16)
        parts = [sb['arena'][sb['offsets'][i]:sb['offsets'][i + 1]] for i in range(len(sb['offsets']) - 1)] + [flat]
        arena, off = N.pack_reads(parts)
        sb = {'arena': arena, 'offsets': off, 'calib': np.concatenate([sb['calib'], sb['calib'][:1]])}
    pool.append(sb)
variants = []
for k, sb in enumerate(pool):
    n = len(sb['offsets']) - 1
    for mask, scan in ((N.STAGE_ALL_DEMUX, False), (N.STAGE_ALL_DEMUX | N.STAGE_POLYA, True)):
        ctx.upload(sb['arena'], sb['offsets'], sb['calib'])
        ctx.run(mask)
        rec = ctx.download().copy()
        spk = ctx.download_spikes(rec) if mask & N.STAGE_POLYA else None
        sc = ctx.unsplit_scan(np.zeros(n, np.int64), np.diff(sb['offsets']) // 15) if scan else None
        z, chunks, _ = N.z_encode(sb['arena'], sb['offsets'])
        variants.append((sb, mask, scan, rec, spk, sc, N.EncodedSamples(z, chunks, 0, 0, len(sb['arena']))))
print('reference results of', len(variants), 'batch variants ready', flush=True)
picks = rng.integers(0, len(variants), calls)
encoded = rng.random(calls) < 0.5


def one(i):
    sb, mask, scan, rec, spk, sc, enc = variants[picks[i]]
    n = len(sb['offsets']) - 1
    got = ctx.process_batch_ex(enc if encoded[i] else sb['arena'], sb['offsets'], sb['calib'], mask,
                               want_spikes=bool(mask & N.STAGE_POLYA),
                               unsplit=(np.zeros(n, np.int64), np.diff(sb['offsets']) // 15, 15) if scan else None)
    ok = all(np.array_equal(got['records'][f], rec[f], equal_nan=True) for f in rec.dtype.names)   # (not the padding bytes)
    if spk is not None:
        ok = ok and np.array_equal(got['spikes'][1], spk[1]) and np.array_equal(got['spikes'][0], spk[0], equal_nan=True)
    if sc is not None:
        ok = ok and np.array_equal(got['unsplit'][1], sc[1]) and np.array_equal(got['unsplit'][0], sc[0])
    if not ok:
        bad = [f for f in rec.dtype.names if not np.array_equal(got['records'][f], rec[f], equal_nan=True)]
        print('MISMATCH call', i, 'variant', int(picks[i]), 'encoded', bool(encoded[i]), 'fields', bad[:6], flush=True)
    return ok


t0 = time.perf_counter()
with ThreadPoolExecutor(threads) as ex:
    oks = list(ex.map(one, range(calls)))
print('{} calls on {} threads in {:.1f} s: {} mismatches'.format(calls, threads, time.perf_counter() - t0,
                                                                 calls - sum(oks)))
sys.exit(0 if all(oks) else 1)
