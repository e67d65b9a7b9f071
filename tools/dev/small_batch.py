"""Stage times of small resident batches, K2's latency form on / off (development aid, GPU box):
python tools/dev/small_batch.py [n ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poreplex_amd import native as N          # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.synth import synth_batch      # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [128, 512, 940, 1024, 2048, 4096]
ctx = N.NativeContext(default_config(), device_id=0)
for n in sizes:
    b = synth_batch(n, seed=5, samples_per_read=60000)
    ctx.upload(b['arena'], b['offsets'], b['calib'])
    ref = None
    for mode in ('tile', 'lat', 'tile', 'lat'):
        if mode == 'tile':
            os.environ['PXG_K2_LAT_MAX'] = '0'
            os.environ['PXG_K5_LAT_MAX'] = '0'
        else:
            os.environ.pop('PXG_K2_LAT_MAX', None)
            os.environ.pop('PXG_K5_LAT_MAX', None)
        best = None
        for rep in range(4):
            ctx.run()
            ctx.sync()
            ms, _ = ctx.stage_times()
            if best is None or ms['total'] < best['total']:
                best = ms
        rec = ctx.download()
        if ref is None:
            ref = rec
        same = all(np.array_equal(rec[f], ref[f], equal_nan=True) for f in rec.dtype.names)
        print(n, mode, 'total %.3f' % best['total'], ' '.join('%s %.3f' % (k, v) for k, v in best.items() if v > 0.005 and k != 'total'),
              'same-as-first', same, flush=True)
ctx.close()
