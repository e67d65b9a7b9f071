#!/usr/bin/env python
"""cProfile of ONE process_batch call on the GPU (host side of the reference-shaped API).
usage (GPU box): python tools/api_profile.py [demux|full] [reads]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.fast5_file import write_bundle  # noqa: E402
from poreplex_amd.signal_analyzer import process_batch  # noqa: E402
from poreplex_amd.synth import synth_basecalls, synth_batch  # noqa: E402

full = len(sys.argv) > 1 and sys.argv[1] == 'full'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
sb = synth_batch(n, seed=924, samples_per_read=60000)
work = tempfile.mkdtemp(prefix='pxg_apiprof_')
names = ['a/r%07d.fast5' % i for i in range(n)]
ids = ['%08x-0000-4000-8000-%012x' % (924, i) for i in range(n)]
path = os.path.join(work, 'b.pxr.npz')
write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], names, ids, basecalls=synth_basecalls(sb, seed=924))
cfg = default_config(inputdir=work, outputdir=work, read_bundle=path, barcoding=True, measure_polya=full,
                     filter_unsplit_reads=full)
reads = list(zip(names, ids))
r = process_batch(0, reads, cfg)
assert isinstance(r, list), r
for k in range(2):
    t0 = time.perf_counter()
    r = process_batch(1 + k, reads, cfg)
    print('call %d: %.1f ms' % (k, (time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
r = process_batch(9, reads, cfg)
pr.disable()
from collections import Counter
print(Counter(x['status'] for x in r))
pstats.Stats(pr).sort_stats('cumulative').print_stats(32)
