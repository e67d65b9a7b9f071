#!/usr/bin/env python3
"""What the HOST side of reference-sized worker calls can sustain, without a GPU.

`process_batch` is called with 128 consecutive reads of a read bundle, one call at a time and from 8 / 32 / 64 threads
of one interpreter, over a stand-in context whose GPU pass is `time.sleep(--gpu-ms)` (interpreter lock released, like
the real call) and whose records are made up.  The GPU pass overlaps freely here, so the figure at 32 threads is what
the interpreter lock allows -- the ceiling the measured GPU figure (profiles/r06/api_128_read_calls.txt) sits under --
and the one-at-a-time figure minus the sleep is the Python cost of a call.

    python tools/dev/host_cap.py                  # the shipped path
    PXG_NO_PLAIN_RUN=1 python tools/dev/host_cap.py      # every call through the batch table (the path before)
"""
import argparse
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from poreplex_amd import native as N                                      # noqa: E402
from poreplex_amd import signal_analyzer as SA                            # noqa: E402
from poreplex_amd.config import default_config                            # noqa: E402
from poreplex_amd.fast5_file import write_bundle                          # noqa: E402
from poreplex_amd.synth import synth_basecalls, synth_batch               # noqa: E402
from poreplex_amd.worker_persistence import WorkerPersistenceStorage      # noqa: E402


class SleepingContext:
    """Stand-in for native.NativeContext: the one native call of a worker batch, answered after a sleep."""
    gpu_ms = 3.0
    records = None

    def __init__(self, config, device_id=0):
        self.ncfg = N.NativeConfig(config)
        self.cfg = self.ncfg.struct
        self.state_names = self.ncfg.state_names

    def pin(self, array):
        return array

    def unpin(self, array):
        pass

    def close(self):
        pass

    candidates_every = 0          # --full: every n-th read comes back from the chimera scan with a candidate

    # the NATIVE face of the one call (native.BatchExCall, csrc/pxg_pyreport.c decode_and_run): libc's usleep stands in
    # for pxg_process_batch_ex -- its first argument, the context handle, is the microseconds to sleep; the other seven
    # are ignored -- so the sleep happens without the interpreter lock and without any Python, like the real pass
    def batch_ex_call(self, n, stage_mask=N.STAGE_ALL_DEMUX, unsplit=None, want_spikes=False):
        import ctypes as C
        import types
        if not hasattr(self, 'lib'):
            self.lib = types.SimpleNamespace(pxg_process_batch_ex=C.CDLL(None).usleep)
        self.handle = C.c_void_p(max(int(self.gpu_ms * 1000), 1))
        call = N.BatchExCall(self, n, stage_mask, unsplit, want_spikes)
        call.records[:] = SleepingContext.records[:n]
        return call

    def _check(self, rc, what):
        assert rc == 0, (what, rc)

    def process_batch_ex(self, samples, offsets, calib, stage_mask=N.STAGE_ALL_DEMUX, scale_shift=None, unsplit=None,
                         want_spikes=False):
        time.sleep(self.gpu_ms * 1e-3)
        k = len(offsets) - 1
        out = {'records': SleepingContext.records[:k].copy()}
        if want_spikes:
            out['spikes'] = (np.zeros((0, 4), np.float32), np.zeros(k + 1, np.int64))
        if unsplit is not None:
            every = SleepingContext.candidates_every
            cnt = (np.arange(k) % every == 0).astype(np.int32) if every else np.zeros(k, np.int32)
            start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
            out['unsplit'] = (np.tile(np.array([[5000, 5200]], np.int64), (int(cnt.sum()), 1)), cnt, start)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=128)
    ap.add_argument('--gpu-ms', type=float, default=3.0)
    ap.add_argument('--calls', type=int, default=640)
    ap.add_argument('--repeats', type=int, default=4)
    ap.add_argument('--full', action='store_true',
                    help='configs[3]-shaped calls (poly(A) + chimera scan), one at a time, no sleep: Python per read')
    ap.add_argument('--candidates-every', type=int, default=0, help='--full: every n-th read has a chimera candidate')
    ap.add_argument('--files', type=int, default=1, help='--fast5: the reads in this many multi-read files, every timed pass '
                    'with the reader\'s file cache emptied first (a run meets new files all the time)')
    ap.add_argument('--single', action='store_true', help='--fast5: one single-read file per read (the classic input)')
    ap.add_argument('--fast5', choices=('none', 'vbz'), default=None,
                    help='the calls read a multi-read FAST5 file (the reference\'s real input) instead of a read bundle')
    ap.add_argument('--file-reads', type=int, default=4000, help='--fast5: reads in the file the calls walk through')
    ap.add_argument('--real', action='store_true',
                    help='--fast5: the real GPU context instead of the sleeping stand-in (on the GPU box: the figure itself)')
    args = ap.parse_args()
    if args.full:
        return full_calls(args)
    if args.fast5:
        return fast5_calls(args)
    n = args.reads
    sb = synth_batch(n, seed=924, samples_per_read=20000)
    work = tempfile.mkdtemp(prefix='pxg_hostcap_')
    names = ['a/r%07d.fast5' % i for i in range(n)]
    ids = ['%08x-0000-4000-8000-%012x' % (924, i) for i in range(n)]
    path = os.path.join(work, 'b.pxr.npz')
    write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], names, ids, basecalls=synth_basecalls(sb, seed=924))
    cfg = default_config(inputdir=work, outputdir=work, read_bundle=path, barcoding=True)
    SleepingContext.gpu_ms = args.gpu_ms
    N.NativeContext = SleepingContext
    WorkerPersistenceStorage.reset()
    rng = np.random.default_rng(924)
    rec = np.zeros(n, dtype=N.RESULT_DTYPE)                 # mostly passing reads with a barcode, like a good run
    adapter = N.NativeConfig(cfg).state_names.index('adapter')
    rec['seg_first'], rec['seg_last'] = -1, -1
    rec['seg_first'][:, adapter], rec['seg_last'][:, adapter] = np.where(rng.random(n) < 0.95, 30, -1), 80
    rec['scale'], rec['shift'] = 1.0, 0.0
    rec['bc_pushed'], rec['bc_called'] = rng.random(n) < 0.95, rng.random(n) < 0.8
    rec['bc_label'], rec['bc_phred'] = rng.integers(0, 4, n), rng.integers(10, 50, n)
    SleepingContext.records = rec
    reads = list(zip(names, ids))
    first = SA.process_batch(0, reads, cfg)
    assert isinstance(first, list) and len(first) == n, first
    print('path: %s' % ('batch table (PXG_NO_PLAIN_RUN)' if not SA._PLAIN_RUN else 'plain run'))
    t0 = time.perf_counter()
    for k in range(50):
        SA.process_batch(1 + k, reads, cfg)
    one = (time.perf_counter() - t0) / 50
    print('one call at a time: %.3f ms per call, %.1f ms of it the sleep -> %.3f ms of Python per %d-read call'
          % (one * 1e3, args.gpu_ms, one * 1e3 - args.gpu_ms, n))
    for threads in (8, 32, 64):
        rates = []
        for _ in range(args.repeats):          # (a shared development host is noisy: every repeat is printed)
            with ThreadPoolExecutor(threads) as pool:
                t0 = time.perf_counter()
                list(pool.map(lambda k: len(SA.process_batch(100 + k, reads, cfg)), range(args.calls)))
                rates.append(args.calls / (time.perf_counter() - t0))
        print('%2d threads: best %5.0f calls/s = %6.0f reads/s (%.3f ms of wall clock per call); all: %s'
              % (threads, max(rates), max(rates) * n, 1e3 / max(rates), ' '.join('%.0f' % r for r in rates)))
    WorkerPersistenceStorage.reset()


def fast5_calls(args):
    """Reference-sized calls over a multi-read FAST5 file: consecutive 128-read slices of its read list, as the
    reference's batch maker hands them out (commandline.py:398-402)."""
    from poreplex_amd.fast5_write import Fast5Writer
    n, total = args.reads, args.file_reads
    sb = synth_batch(total, seed=924, samples_per_read=20000)
    bcs = synth_basecalls(sb, seed=924)
    work = tempfile.mkdtemp(prefix='pxg_hostcap_')
    ids = ['%08x-0000-4000-8000-%012x' % (924, i) for i in range(total)]
    o = sb['offsets']
    if args.single:
        from poreplex_amd.fast5_write import write_single_read
        os.makedirs(os.path.join(work, 'd'))
        for j in range(total):
            write_single_read(os.path.join(work, 'd', 'r%06d.fast5' % j), ids[j], sb['arena'][o[j]:o[j + 1]], sb['calib'][j],
                              start_time=j, channel_number=str(1 + j % 512), basecall=bcs[j],
                              compression=None if args.fast5 == 'none' else args.fast5)
    else:
        per_file = -(-total // args.files // n) * n
        for lo in range(0, total, per_file):
            with Fast5Writer(os.path.join(work, 'run.fast5' if args.files == 1 else 'run%03d.fast5' % (lo // per_file))) as w:
                for j in range(lo, min(lo + per_file, total)):
                    w.add_read(ids[j], sb['arena'][o[j]:o[j + 1]], sb['calib'][j], start_time=j, channel_number=str(1 + j % 512),
                               basecall=bcs[j], compression=None if args.fast5 == 'none' else args.fast5)
    cfg = default_config(inputdir=work, outputdir=work, barcoding=True)
    SleepingContext.gpu_ms = args.gpu_ms
    if not args.real:
        N.NativeContext = SleepingContext
    WorkerPersistenceStorage.reset()
    rng = np.random.default_rng(924)
    rec = np.zeros(n, dtype=N.RESULT_DTYPE)
    adapter = N.NativeConfig(cfg).state_names.index('adapter')
    rec['seg_first'], rec['seg_last'] = -1, -1
    rec['seg_first'][:, adapter], rec['seg_last'][:, adapter] = np.where(rng.random(n) < 0.95, 30, -1), 80
    rec['scale'], rec['shift'] = 1.0, 0.0
    rec['bc_pushed'], rec['bc_called'] = rng.random(n) < 0.95, rng.random(n) < 0.8
    rec['bc_label'], rec['bc_phred'] = rng.integers(0, 4, n), rng.integers(10, 50, n)
    SleepingContext.records = rec
    calls = [[('run.fast5', r) for r in ids[k:k + n]] for k in range(0, total - n + 1, n)]
    if args.single:
        calls = [[('d/r%06d.fast5' % j, ids[j]) for j in range(k, k + n)] for k in range(0, total - n + 1, n)]
    elif args.files > 1:
        calls = [[('run%03d.fast5' % (k // per_file), r) for r in ids[k:k + n]] for k in range(0, total - n + 1, n)]
    first = SA.process_batch(0, calls[0], cfg)
    assert isinstance(first, list) and len(first) == n, first
    print('%s; FAST5 (%s), %d calls of %d reads; path: %s' % ('REAL context' if args.real else 'sleeping stand-in', args.fast5, len(calls), n,
          'batch table (PXG_NO_PLAIN_RUN)' if not SA._PLAIN_RUN else 'plain run where it applies'))
    best = 1e9
    for _ in range(max(args.repeats, 3)):
        t0 = time.perf_counter()
        for k, reads in enumerate(calls):
            SA.process_batch(1 + k, reads, cfg)
        best = min(best, (time.perf_counter() - t0) / len(calls))
    print('one call at a time: %.3f ms per call, %.1f ms of it the sleep -> %.3f ms of host work per %d-read call '
          '(plain-run calls so far: %d)' % (best * 1e3, args.gpu_ms, best * 1e3 - args.gpu_ms, n, SA.PLAIN_RUN_CALLS))
    for threads in (8, 32):
        rates = []
        for _ in range(args.repeats):
            if args.files > 1:
                from poreplex_amd.fast5_file import clear_open_cache
                clear_open_cache()
            with ThreadPoolExecutor(threads) as pool:
                t0 = time.perf_counter()
                if args.files > 1:
                    # as the reference's pipeline does it (pipeline.py:321-324): the scanner lists a file, then the calls
                    # over its reads are handed to the workers -- here threads of this process
                    from poreplex_amd.fast5_file import get_read_ids
                    futures = []
                    for name in sorted({c[0][0] for c in calls}):
                        listed = get_read_ids(name, work)
                        futures += [pool.submit(SA.process_batch, 100 + k, listed[k:k + n], cfg) for k in range(0, len(listed), n)]
                    assert all(len(f.result()) == n for f in futures) and len(futures) == len(calls)
                else:
                    list(pool.map(lambda k: len(SA.process_batch(100 + k, calls[k % len(calls)], cfg)), range(args.calls)))
                rates.append((len(calls) if args.files > 1 else args.calls) / (time.perf_counter() - t0))
        print('%2d threads: best %5.0f calls/s = %6.0f reads/s (%.3f ms of wall clock per call); all: %s'
              % (threads, max(rates), max(rates) * n, 1e3 / max(rates), ' '.join('%.0f' % r for r in rates)))
    WorkerPersistenceStorage.reset()


def full_calls(args):
    """Python per read of calls with the poly(A) stage and the chimera scan on, 2 000 reads each (no sleep)."""
    n = max(args.reads, 2000)
    sb = synth_batch(n, seed=924, samples_per_read=12000)
    work = tempfile.mkdtemp(prefix='pxg_hostcap_')
    names = ['a/r%07d.fast5' % i for i in range(n)]
    ids = ['%08x-0000-4000-8000-%012x' % (924, i) for i in range(n)]
    path = os.path.join(work, 'b.pxr.npz')
    write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], names, ids, basecalls=synth_basecalls(sb, seed=924))
    cfg = default_config(inputdir=work, outputdir=work, read_bundle=path, barcoding=True, measure_polya=True,
                         filter_unsplit_reads=True)
    SleepingContext.gpu_ms, SleepingContext.candidates_every = 0.0, args.candidates_every
    N.NativeContext = SleepingContext
    WorkerPersistenceStorage.reset()
    rec = np.zeros(n, dtype=N.RESULT_DTYPE)
    adapter = N.NativeConfig(cfg).state_names.index('adapter')
    rec['seg_first'], rec['seg_last'] = -1, -1
    rec['seg_first'][:, adapter], rec['seg_last'][:, adapter] = 30, 80
    rec['bc_pushed'] = rec['bc_called'] = 1
    rec['polya_called'], rec['polya_dwell_samples'] = 1, 500
    SleepingContext.records = rec
    reads = list(zip(names, ids))
    out = SA.process_batch(0, reads, cfg)
    assert isinstance(out, list), out
    statuses = {}
    for r in out:
        statuses[r['status']] = statuses.get(r['status'], 0) + 1
    best = 1e9
    for _ in range(max(args.repeats, 3)):
        t0 = time.perf_counter()
        SA.process_batch(1, reads, cfg)
        best = min(best, time.perf_counter() - t0)
    print('%s, %s, candidates in every %s read: %.2f ms per %d-read call = %.2f us of Python per read; statuses %s'
          % ('plain run' if SA._PLAIN_RUN else 'batch table (PXG_NO_PLAIN_RUN)',
             'candidates from the Move column' if SA._BULK_UNSPLIT else 'candidates through the event table (PXG_NO_BULK_UNSPLIT)',
             args.candidates_every or 'no', best * 1e3, n, best / n * 1e6, statuses))
    WorkerPersistenceStorage.reset()


if __name__ == '__main__':
    main()
