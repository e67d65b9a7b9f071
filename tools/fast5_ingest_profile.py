#!/usr/bin/env python
"""Where FAST5 ingestion spends its time on this host (no GPU involved): a multi-read file per
compression, then open / ids / info / signals / basecall text per thread count, into a
pre-touched arena (what the session's reused staging buffer is)."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poreplex_amd import fast5_file as F5  # noqa: E402
from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.fast5_write import Fast5Writer  # noqa: E402
from poreplex_amd.synth import synth_basecalls, synth_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
base = synth_batch(n, seed=1, samples_per_read=60000)
o = base['offsets']
bcs = synth_basecalls(base, seed=2)
work = tempfile.mkdtemp(prefix='pxg_f5prof_')
lib = N.load_text_library()


class _Cfg:
    stride, scaler_length, scaler_min_length = 15, 30000, 4500
    scaler_qc_scale, scaler_qc_shift = (0.0, 1e9), (-1e9, 1e9)


class _Ctx:
    cfg = _Cfg()


def loader_call(path, count, arena):
    """The session loader's own call for one batch (signal_loader.prepare_many: file lookup,
    metadata, signals into the staging arena, basecall text, table columns), cold file cache of
    the reader (the page cache stays warm)."""
    from poreplex_amd.config import default_config
    from poreplex_amd.signal_loader import ReadTable, SignalLoader
    top, name = os.path.split(path)
    loader = SignalLoader(default_config(inputdir=top, outputdir=top), top, _Ctx())
    reads = [(name, 'r%06d' % j) for j in range(count)]
    best = None
    for _ in range(3):
        F5.clear_open_cache()
        t0 = time.perf_counter()
        where = loader.prepare_many(reads, ReadTable(), reserve=lambda k: arena[:k])
        dt = time.perf_counter() - t0
        assert (where >= 0).all()
        best = dt if best is None else min(best, dt)
    return best


MODES = [None if m == 'none' else m for m in os.environ.get('PXG_PROF_MODES', 'none,vbz,gzip').split(',')]
for mode in MODES:
    count = n if mode != 'gzip' else min(n, 1000)
    path = os.path.join(work, str(mode) + '.fast5')
    with Fast5Writer(path) as w:
        for j in range(count):
            w.add_read('r%06d' % j, base['arena'][o[j]:o[j + 1]], base['calib'][j], basecall=bcs[j], compression=mode)
    for threads in (1, 4, 8, 16, 32):
        F5.clear_open_cache()
        t0 = time.perf_counter()
        f = F5.Fast5File(path)
        t1 = time.perf_counter()
        f.read_ids
        t2 = time.perf_counter()
        info = np.zeros(f.n, dtype=N.H5_INFO_DTYPE)
        lib.pxg_h5_info_mt(f.handle, 0, f.n, info.ctypes.data, threads)
        f._info = info
        t3 = time.perf_counter()
        ns = info['n_samples'].astype(np.int64)
        arena = np.empty(int(ns.sum()), dtype=np.int16)
        arena.fill(0)                                             # touched: a reused staging buffer (np.zeros is not)
        dst = np.concatenate([[0], np.cumsum(ns)[:-1]]).astype(np.int64)
        t4 = time.perf_counter()
        st = F5.load_signals([f] * f.n, np.arange(f.n), ns, arena, dst, threads)
        t5 = time.perf_counter()
        b = F5.Fast5Batch([f] * f.n, np.arange(f.n), ['x'] * f.n)
        t6 = time.perf_counter()
        b.as_bundle(reserve=lambda k: arena, threads=threads)
        t7 = time.perf_counter()
        assert not st.any()
        print('{:5s} {:5d} reads, {:2d} threads: open {:6.1f} ids {:5.1f} info {:6.1f} signals {:7.1f} ms ({:5.1f} GB/s) '
              'batch {:5.1f} as_bundle(all) {:7.1f} ms -> {:7.0f} reads/s'.format(
                  str(mode), f.n, threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3,
                  ns.sum() * 2 / (t5 - t4) / 1e9, (t6 - t5) * 1e3, (t7 - t6) * 1e3,
                  f.n / ((t3 - t0) + (t7 - t5))), flush=True)
    dt = loader_call(path, count, arena)
    print('{:5s} {:5d} reads, loader call (prepare_many, {} threads): {:7.1f} ms -> {:7.0f} reads/s'.format(
        str(mode), count, F5.host_threads(), dt * 1e3, count / dt), flush=True)
