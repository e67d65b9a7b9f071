#!/usr/bin/env python3
"""Mutated multi-read FAST5 files through the worker call (process_batch over FAST5 files: plain-run path, fused decode
+ pass, fall-backs) -- host only, the context a double whose pass is a C callback that fills plausible records.  Every
call must come back as a list of dicts (per-read errors as data) or the listing must refuse the file; nothing may crash.

    make -C poreplex_amd/csrc asan
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
        python tools/dev/fuzz_calls.py <seed> <trials>
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from poreplex_amd import native as N                                      # noqa: E402
ASAN_LIB = os.path.join(os.path.dirname(N.TEXT_LIB_PATH), '_obj', 'libpxghost_asan.so')
if os.path.isfile(ASAN_LIB) and 'asan' in os.environ.get('LD_PRELOAD', ''):
    N._text_lib = N.load_text_library(ASAN_LIB)
from poreplex_amd import fast5_file as F5                                 # noqa: E402
from poreplex_amd import signal_analyzer as SA                            # noqa: E402
from poreplex_amd.config import default_config                            # noqa: E402
from poreplex_amd.fast5_write import Fast5Writer                          # noqa: E402
from poreplex_amd.synth import synth_basecalls, synth_batch               # noqa: E402
from poreplex_amd.worker_persistence import WorkerPersistenceStorage      # noqa: E402
import oracle_context                                                      # noqa: E402


class Canned(oracle_context.NativeEntryMixin, oracle_context.OracleBackedContext):
    def process_batch_ex(self, samples, offsets, calib, stage_mask=N.STAGE_ALL_DEMUX, scale_shift=None, unsplit=None,
                         want_spikes=False):
        n = len(offsets) - 1
        rec = np.zeros(n, dtype=N.RESULT_DTYPE)
        rec['seg_first'], rec['seg_last'] = -1, -1
        rec['seg_first'][:, 0], rec['seg_last'][:, 0] = 10, 40
        rec['scale'] = 1.0
        out = {'records': rec}
        if want_spikes:
            out['spikes'] = (np.zeros((0, 4), np.float32), np.zeros(n + 1, np.int64))
        if unsplit is not None:
            out['unsplit'] = (np.zeros((0, 2), np.int64), np.zeros(n, np.int32), np.zeros(n + 1, np.int64))
        return out


def main():
    seed, trials = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300
    rng = np.random.default_rng(seed)
    work = tempfile.mkdtemp(prefix='pxg_fuzzcalls_')
    sb = synth_batch(48, seed=3, samples_per_read=12000)
    bcs = synth_basecalls(sb, seed=3)
    o = sb['offsets']
    sources = []
    for mode in (None, 'gzip', 'vbz'):
        path = os.path.join(work, 'src_{}.fast5'.format(mode))
        try:
            with Fast5Writer(path) as w:
                for j in range(48):
                    w.add_read('{:08x}-{:027d}'.format(3, j), sb['arena'][o[j]:o[j + 1]], sb['calib'][j], start_time=j,
                               basecall=bcs[j] if j % 9 else None, compression=mode)
            sources.append(open(path, 'rb').read())
        except OSError:
            pass
    N.NativeContext = Canned
    cfg = default_config(inputdir=work, outputdir=work, barcoding=True, filter_unsplit_reads=True, measure_polya=True)
    refused = listed = calls = errors = 0
    for trial in range(trials):
        blob = bytearray(sources[trial % len(sources)])
        kind = trial % 3
        if kind == 0:
            for pos in rng.integers(0, len(blob), rng.integers(1, 12)):
                blob[pos] = rng.integers(0, 256)
        elif kind == 1:
            blob = blob[:rng.integers(len(blob) // 2, len(blob))]
        else:
            for pos in rng.integers(8, len(blob) - 8, rng.integers(1, 8)):
                blob[pos:pos + 8] = rng.choice([b'\xff' * 8, b'\x00' * 8, (2 ** 63 - 1).to_bytes(8, 'little'),
                                                int(rng.integers(0, 2 ** 40)).to_bytes(8, 'little')])
        name = 'm{}.fast5'.format(trial % 4)
        with open(os.path.join(work, name), 'wb') as fh:
            fh.write(bytes(blob))
        F5.clear_open_cache()
        try:
            keys = F5.get_read_ids(name, work)
        except OSError:
            refused += 1
            continue
        listed += 1
        for lo in range(0, len(keys), 16):
            got = SA.process_batch(trial, keys[lo:lo + 16], cfg)
            calls += 1
            if isinstance(got, tuple):            # (the fatal tuple: a bug unless the double itself failed)
                raise SystemExit('fatal tuple at trial {}: {}'.format(trial, got[1]))
            assert len(got) == len(keys[lo:lo + 16])
            errors += sum(r['status'] == 'unknown_error' for r in got)
    WorkerPersistenceStorage.reset()
    print('trials {}: listing refused {}, listed {}, calls {} (plain-run {}), reads reported as unknown_error {}'.format(
        trials, refused, listed, calls, SA.PLAIN_RUN_CALLS, errors))


if __name__ == '__main__':
    main()
