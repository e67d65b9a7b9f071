#!/usr/bin/env python
"""Host cost of the reference-shaped call (process_batch -> result dicts) WITHOUT a GPU:
the numeric stages are answered from records the oracle computed once (test double), so
what is timed is only the Python / host-library work of poreplex_amd/signal_analyzer.py.
Development tool (imports oracle/ through tests/oracle_context.py): never part of the product.

    python tools/host_profile.py [--reads 10000] [--profile]
"""
import argparse
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.fast5_file import write_bundle  # noqa: E402
from poreplex_amd.synth import synth_basecalls, synth_batch  # noqa: E402


class ReplayContext:
    """NativeContext double: records of `distinct` reads computed once, replayed by tiling."""
    records = spikes = None
    distinct = 0

    def __init__(self, config, device_id=0):
        from oracle.pxo import Oracle
        o = Oracle(config)
        self.ncfg, self.cfg, self.state_names = o.ncfg, o.cfg, o.state_names
        self.n_resident = 0

    def device_info(self):
        return {'name': 'replay double', 'arch': 'none', 'compute_units': 0}

    def upload(self, arena, offsets, calib, scale_shift=None):
        self.n_resident = len(offsets) - 1

    def stage(self, arena, offsets, calib, scale_shift=None):
        self._n = len(offsets) - 1

    stage_z = lambda self, enc, offsets, calib, scale_shift=None: self.stage(None, offsets, calib)

    def swap(self):
        self.n_resident = self._n

    def pin(self, a):
        return a

    def unpin(self, a):
        pass

    def run(self, mask=7):
        pass

    def sync(self):
        pass

    def download(self, out=None):
        k = np.arange(self.n_resident) % self.distinct
        return ReplayContext.records[k]

    def download_spikes(self, records=None):
        from oracle.pxo import spikes_csr
        k = np.arange(self.n_resident) % self.distinct
        return spikes_csr(ReplayContext.records[k], ReplayContext.spikes[k])

    def unsplit_scan(self, first, nb, stride=15):
        n = self.n_resident
        return np.zeros((0, 2), np.int64), np.zeros(n, np.int32), np.zeros(n + 1, np.int64)

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=10000)
    ap.add_argument('--distinct', type=int, default=250)
    ap.add_argument('--samples', type=int, default=12000)
    ap.add_argument('--calls', type=int, default=3)
    ap.add_argument('--polya', action='store_true')
    ap.add_argument('--chimera', action='store_true')
    ap.add_argument('--profile', action='store_true')
    a = ap.parse_args()
    from oracle.pxo import Oracle
    config = default_config()
    base = synth_batch(a.distinct, seed=924, samples_per_read=a.samples)
    mask = N.STAGE_ALL_DEMUX | (N.STAGE_POLYA if a.polya else 0)
    rec, spk = Oracle(config).process_batch(base['arena'], base['offsets'], base['calib'],
                                            stage_mask=mask, want_spikes=True)
    ReplayContext.records, ReplayContext.spikes, ReplayContext.distinct = rec, spk, a.distinct
    which = np.arange(a.reads) % a.distinct
    o = base['offsets']
    arena, off = N.pack_reads([base['arena'][o[b]:o[b + 1]] for b in which])
    names = ['d/read{:07d}.fast5'.format(j) for j in range(a.reads)]
    ids = ['{:08x}-0000-4000-8000-{:012x}'.format(924, j) for j in range(a.reads)]
    work = tempfile.mkdtemp(prefix='pxg_hp_')
    path = os.path.join(work, 'b.pxr.npz')
    write_bundle(path, arena, off, base['calib'][which], names, ids,
                 basecalls=synth_basecalls({'offsets': off}, seed=1))
    N.NativeContext = ReplayContext
    from poreplex_amd import signal_analyzer as SA
    cfg = default_config(inputdir=work, outputdir=work, read_bundle=path, barcoding=True,
                         measure_polya=a.polya, filter_unsplit_reads=a.chimera)
    reads = list(zip(names, ids))
    SA.process_batch(0, reads[:64], cfg)           # context + bundle load
    prof = cProfile.Profile() if a.profile else None
    for k in range(a.calls):
        t0 = time.perf_counter()
        if prof:
            prof.enable()
        res = SA.process_batch(k, reads, cfg)
        if prof:
            prof.disable()
        dt = time.perf_counter() - t0
        assert isinstance(res, list) and len(res) == a.reads, res[:3]
        print('call {}: {:.1f} ms for {} reads ({:.2f} us/read), {} ok'.format(
            k, dt * 1e3, a.reads, dt / a.reads * 1e6, sum(r['status'] == 'okay' for r in res)))
    if prof:
        pstats.Stats(prof).sort_stats('cumulative').print_stats(28)


if __name__ == '__main__':
    main()
