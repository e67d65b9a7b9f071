#!/usr/bin/env python3
"""A directory of SINGLE-read FAST5 files (the reference's classic input) through the loader, host only: N files are
written, then SignalLoader.prepare_many opens, describes and decodes them into one arena -- by the batch open
(pxg_h5_open_many: one native call on host threads) and by the per-file route it replaced (a Python round per file).

    python tools/dev/single_read_ingest.py [N=1024] [none|vbz|gzip]
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from poreplex_amd import fast5_file as F5                                  # noqa: E402
from poreplex_amd.config import default_config                             # noqa: E402
from poreplex_amd.fast5_write import write_single_read                     # noqa: E402
from poreplex_amd.signal_loader import ReadTable, SignalLoader            # noqa: E402
from poreplex_amd.synth import synth_basecalls, synth_batch                # noqa: E402


class _Cfg:
    stride, scaler_length, scaler_min_length = 15, 30000, 4500
    scaler_qc_scale, scaler_qc_shift = (0.0, 1e9), (-1e9, 1e9)


class _Ctx:
    cfg = _Cfg()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    mode = sys.argv[2] if len(sys.argv) > 2 else 'none'
    base = synth_batch(n, seed=1, samples_per_read=60000)
    o = base['offsets']
    bcs = synth_basecalls(base, seed=2)
    top = tempfile.mkdtemp(prefix='pxg_single_')
    reads = []
    t0 = time.perf_counter()
    for j in range(n):
        name = 'd{:02d}/r{:06d}.fast5'.format(j // 500, j)
        os.makedirs(os.path.join(top, 'd{:02d}'.format(j // 500)), exist_ok=True)
        write_single_read(os.path.join(top, name), 'r{:06d}'.format(j), base['arena'][o[j]:o[j + 1]], base['calib'][j],
                          basecall=bcs[j], compression=None if mode == 'none' else mode)
        reads.append((name, 'r{:06d}'.format(j)))
    print('{} single-read files ({}) written in {:.1f} s; {} host threads'.format(n, mode, time.perf_counter() - t0, F5.host_threads()))
    loader = SignalLoader(default_config(inputdir=top, outputdir=top), top, _Ctx())
    arena = np.zeros(int(o[-1]) + 16, dtype=np.int16)
    for label, least in (('per-file route', 10 ** 9), ('batch open', 8)):
        loader.SINGLE_READ_BATCH_MIN = least
        best = None
        for _ in range(3):
            F5.clear_open_cache()
            before = dict(F5.TIMING)
            t0 = time.perf_counter()
            where = loader.prepare_many(reads, ReadTable(), reserve=lambda k: arena[:k])
            dt = time.perf_counter() - t0
            assert (where >= 0).all()
            if best is None or dt < best[0]:
                best = (dt, {k: F5.TIMING[k] - before[k] for k in F5.TIMING})
        dt, phases = best
        print('{:15s}: {:7.1f} ms = {:6.1f} us per read ({:.0f} reads/s on this host); native: open + describe {:.1f} ms, '
              'signals {:.1f} ms, basecall text {:.1f} ms'.format(label, dt * 1e3, dt / n * 1e6, n / dt, phases['walk_s'] * 1e3,
                                                                 phases['signals_s'] * 1e3, phases['text_s'] * 1e3))


if __name__ == '__main__':
    main()
